"""SparseDiffAttn / SparseDiffMlp state machines on CPU (config C1 plumbing + FLUX- and Hunyuan-style schedules).

torch.ops.chipmunk.* gets CPU implementations from the ORACLE (tests/cpu_ops.py -- test infrastructure).  The same
oracle-backed ops were used to run the REFERENCE's modules when the fixtures were generated, so identical op sequences
give bit-identical outputs: the comparison pins the per-(step, layer) branch logic, the wrappers' padding contracts
and the storage plumbing.  The recorded op-call traces are compared too.
"""
import os

import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "module_runs.pt")


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


@pytest.fixture()
def cpu_chipmunk(fresh_config):
    import cpu_ops
    cpu_ops.register()
    fresh_config["offloading"]["global_disable_offloading"] = True
    fresh_config["steps"] = 50
    return fresh_config


def _seeded(shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(torch.bfloat16)


def seeded_linear(fin, fout, seed):
    lin = torch.nn.Linear(fin, fout)
    g = torch.Generator().manual_seed(seed)
    bound = 1.0 / fin ** 0.5
    with torch.no_grad():
        lin.weight.copy_((torch.rand(fout, fin, generator=g) * 2 - 1) * bound)
        lin.bias.copy_((torch.rand(fout, generator=g) * 2 - 1) * bound)
    return lin.bfloat16()


def _check(t, d, what):
    assert tuple(t.shape) == d["shape"], what
    assert torch.equal(t.detach().flatten()[::53][:8192], d["sample"]), what
    assert t.double().sum().item() == d["sum"] and t.double().abs().sum().item() == d["abs"], what


def _norm_calls(calls, drop=()):
    return [(n, a) for n, a in calls if n not in drop]


def test_flux_style_attention_schedule(cpu_chipmunk, gold):
    """unpadded q, in-place kernel, torch.topk indices (reference examples/flux/chipmunk-config.yml)."""
    import cpu_ops
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util.layer_counter import LayerCounter
    cpu_chipmunk["attn"].update(dict(top_keys=0.165, full_step_every=10, full_step_schedule=None,
                                     first_n_dense_layers=1, recompute_mask=False, should_compress_indices=False,
                                     counts_multiple_of=112, pad_qkv_before_kernel=False))
    H, N = 2, 1360
    counter = LayerCounter(2, 1)
    layers = [SparseDiffAttn(i, counter) for i in range(2)]
    i = 0
    with cpu_ops.recording() as calls:
        for step in range(12):
            for li, layer in enumerate(layers):
                q, k, v = [_seeded((1, H, N, 128), 1000 + 100 * step + 10 * li + j) for j in range(3)]
                _check(layer(q, k, v), gold["flux_attn_outs"][i], f"flux attn step {step} layer {li}")
                i += 1
        # same ops in the same order; q shapes differ only where this build skips the reference's pad-to-192 copy
        assert [n for n, _ in calls] == [n for n, _ in gold["flux_attn_calls"]]
        ours = [a for n, a in calls if n == "csp_attn"]
        assert ours == [a for n, a in gold["flux_attn_calls"] if n == "csp_attn"]


def test_hunyuan_style_attention_schedule(cpu_chipmunk, gold):
    """padded wrappers, bit-packed mask, mask_to_indices every step, static local mask (reference hunyuan config).
    The fused packed->indices path is off here so the op trace is the reference's."""
    import cpu_ops
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util.layer_counter import LayerCounter
    cpu_chipmunk["attn"].update(dict(top_keys=0.05, random_keys=0.01, local_voxels=0, first_n_dense_layers=1,
                                     recompute_mask=True, should_compress_indices=True, counts_multiple_of=128,
                                     pad_qkv_before_kernel=True, full_step_schedule={0, 1, 4},
                                     fused_packed_mask_to_indices=False))
    H, vid, txt = 2, (8, 12, 16), 40
    N = vid[0] * vid[1] * vid[2] + txt
    counter = LayerCounter(2, 1)
    layers = [SparseDiffAttn(i, counter) for i in range(2)]
    torch.manual_seed(123)  # the static mask draws torch.rand for its random keys
    layers[0].initialize_static_mask(vid, txt, H, torch.device("cpu"))
    i = 0
    with cpu_ops.recording() as calls:
        for step in range(6):
            for li, layer in enumerate(layers):
                q, k, v = [_seeded((1, H, N, 128), 5000 + 100 * step + 10 * li + j) for j in range(3)]
                torch.manual_seed(777 + step * 10 + li)
                _check(layer(q, k, v), gold["hunyuan_attn_outs"][i], f"hunyuan attn step {step} layer {li}")
                i += 1
    # same ops in the same order; shapes differ only where this build skips the reference's pad-to-192 copies
    assert [n for n, _ in calls] == [n for n, _ in gold["hunyuan_attn_calls"]]
    ours_m2i = [a for n, a in calls if n == "mask_to_indices"]
    ref_m2i = [a for n, a in gold["hunyuan_attn_calls"] if n == "mask_to_indices"]
    assert ours_m2i == ref_m2i


def test_sparse_mlp_schedule(cpu_chipmunk, gold):
    """C1 shapes (256 tokens, dim 1024, ffn 4096): dense layer, full steps, top-k/copy/mm1/mm2 sparse steps and the
    cached-mask branch (reference modules/mlp.py:30-120)."""
    import cpu_ops
    from chipmunk_amd.modules import SparseDiffMlp
    from chipmunk_amd.util.layer_counter import LayerCounter
    cpu_chipmunk["mlp"].update(dict(top_keys=0.3, random_keys=0.0, full_step_every=4, block_mask_cache=2,
                                    first_n_dense_layers=1, counts_multiple_of=256))
    counter = LayerCounter(2, 1)
    mlps = [SparseDiffMlp(i, counter, seeded_linear(1024, 4096, 4242 + 2 * i), torch.nn.GELU(approximate="tanh"),
                          seeded_linear(4096, 1024, 4243 + 2 * i), 6) for i in range(2)]
    i = 0
    with cpu_ops.recording() as calls, torch.no_grad():
        for step in range(13):
            for li, m in enumerate(mlps):
                x = (_seeded((1, 256, 1024), 9000 + li).float()
                     + 0.15 * _seeded((1, 256, 1024), 9100 + 10 * step + li).float()).to(torch.bfloat16)
                _check(m(x), gold["mlp_outs"][i], f"mlp step {step} layer {li}")
                i += 1
        assert _norm_calls(calls) == _norm_calls(gold["mlp_calls"])
    names = [n for n, _ in calls]
    assert names.count("csp_mlp_mm1") == names.count("csp_mlp_mm2_and_scatter_add") > 0
    assert names.count("topk_indices") < names.count("csp_mlp_mm1")  # the cached-mask branch was taken


def test_c1_dense_eager_path(cpu_chipmunk, gold):
    """BASELINE.json configs[0]: FLUX single block, 256 tokens, dim 1024, bf16, dense eager CPU path (is_enabled false)."""
    from chipmunk_amd.modules import SparseDiffAttn, SparseDiffMlp
    from chipmunk_amd.util.layer_counter import LayerCounter
    cpu_chipmunk["attn"]["is_enabled"] = False
    cpu_chipmunk["mlp"]["is_enabled"] = False
    q, k, v = [_seeded((1, 8, 256, 128), 70 + j) for j in range(3)]
    x = _seeded((1, 256, 1024), 73)
    counter = LayerCounter(1, 2)
    a = SparseDiffAttn(0, counter)
    m = SparseDiffMlp(0, counter, seeded_linear(1024, 4096, 74), torch.nn.GELU(approximate="tanh"),
                      seeded_linear(4096, 1024, 75), 6)
    with torch.no_grad():
        assert torch.equal(a(q, k, v)[:, :, ::4], gold["c1"]["attn_out"])
        assert torch.equal(m(x)[:, ::4], gold["c1"]["mlp_out"])


def test_fused_packed_path_gives_same_result_on_cpu_tensors(cpu_chipmunk):
    """With CPU tensors the module must take the unfused (bitunpack + mask_to_indices) route even when the flag is on."""
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util.layer_counter import LayerCounter
    cpu_chipmunk["attn"].update(dict(first_n_dense_layers=0, full_step_schedule={0, 1}, top_keys=0.1, random_keys=0.0))
    H, vid, txt = 1, (4, 6, 16), 0
    N = 384
    counter = LayerCounter(1, 1)
    layer = SparseDiffAttn(0, counter)
    layer.initialize_static_mask(vid, txt, H, torch.device("cpu"))
    for step in range(3):
        q, k, v = [_seeded((1, H, N, 128), 10 * step + j) for j in range(3)]
        torch.manual_seed(step)
        out = layer(q, k, v)
        assert out.shape == q.shape and torch.isfinite(out.float()).all()


def test_step_cache_follows_reference_schedule_and_counter(fresh_config):
    """StepCache (reference: inlined in examples/hunyuan/.../models.py:732-741,834-835 and examples/wan/.../model.py:
    580-593,628-630): skipped steps return the last computed state of the same invocation and advance the shared counter
    by exactly one model invocation; computed steps leave the counter to the modules."""
    from chipmunk_amd.util import GLOBAL_CONFIG, LayerCounter, StepCache
    GLOBAL_CONFIG["steps"] = 12
    GLOBAL_CONFIG["step_caching"] = {"is_enabled": True, "skip_step_schedule": {3, 4, 7}}
    for n_inv in (1, 2):
        GLOBAL_CONFIG["num_model_invocations_per_inference_step"] = n_inv
        counter = LayerCounter(num_layers=3, num_sparse_submodules_per_layer=1)
        cache = StepCache(counter)
        seen = []
        for step in range(9):
            for inv in range(n_inv):
                assert counter.cur_inference_step == step and counter.cur_model_invocation_per_step == inv
                if cache.should_skip(step):
                    seen.append((step, inv, "skip", float(cache.skip()[0])))
                else:
                    for _ in range(3):          # the three blocks tick the odometer themselves
                        counter.increment()
                    cache.store(torch.full((2,), 10.0 * step + inv))
                    seen.append((step, inv, "run", 10.0 * step + inv))
        for step, inv, kind, val in seen:
            if kind == "skip":
                last_run = max(s for s, i, k, _ in seen if k == "run" and i == inv and s < step)
                assert val == 10.0 * last_run + inv
    GLOBAL_CONFIG["step_caching"]["is_enabled"] = False
    assert not StepCache(counter).should_skip(3)
    cache2 = StepCache(LayerCounter(1, 1))
    GLOBAL_CONFIG["step_caching"]["is_enabled"] = True
    with pytest.raises(RuntimeError):
        cache2.skip()
