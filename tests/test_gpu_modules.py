"""The module state machines on the GPU: every fused path of this build (GLOBAL_CONFIG keys in util/config.py:
AMD_EXTRA_KEYS) against the reference's op sequence on the same device, over a schedule that covers dense layers, full
steps, mask-reuse steps and sparse steps.  Fusions must not change a single bit."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_flux_schedule(fused: bool, steps: int = 13):
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util import layer_counter as lc
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.modules import SparseDiffAttn, SparseDiffMlp
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)
    cfg.load_from_file(os.path.join(ROOT, "configs", "flux_c2.yml"))
    g = cfg.GLOBAL_CONFIG
    g["steps"] = 50
    g["mlp"]["first_n_dense_layers"] = g["attn"]["first_n_dense_layers"] = 1
    for sec, key in (("attn", "fused_residual"), ("mlp", "fused_scatter"), ("mlp", "fused_topk_delta")):
        g[sec][key] = fused
    dev = torch.device("cuda:0")
    chipmunk_amd.ops.manual_seed(1)   # the 5 % random MLP keys: same seed + same launch order = same columns in both runs
    H, N, HID, FFN, n_layers = 3, 1152, 512, 2048, 3
    gen = torch.Generator(device=dev).manual_seed(77)
    layers = []
    for _ in range(n_layers):
        num, counter = LayerCounter.build_for_layer(is_mlp_sparse=True, is_attn_sparse=True)
        fc1 = torch.nn.Linear(HID, FFN, device=dev, dtype=torch.bfloat16)
        fc2 = torch.nn.Linear(FFN, HID, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            for prm in (fc1.weight, fc1.bias, fc2.weight, fc2.bias):
                prm.copy_((torch.randn(prm.shape, device=dev, generator=gen) * 0.05).to(torch.bfloat16))
        layers.append((SparseDiffAttn(num, counter), SparseDiffMlp(num, counter, fc1, torch.nn.GELU(approximate="tanh"), fc2, 6)))
    q0, k0, v0, dq = [torch.randn(1, H, N, 128, device=dev, generator=gen) for _ in range(4)]
    x0, dx = [torch.randn(1, N, HID, device=dev, generator=gen) for _ in range(2)]
    outs = []
    with torch.no_grad():
        for step in range(steps):
            a = 0.05 * step
            q = (q0 + a * dq).to(torch.bfloat16)
            x = (x0 + a * dx).to(torch.bfloat16)
            for attn, mlp in layers:
                outs.append(attn(q, k0.to(torch.bfloat16), v0.to(torch.bfloat16)).clone())
                outs.append(mlp(x).clone())
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)
    return outs


def test_fused_paths_do_not_change_a_bit_over_a_flux_schedule():
    """13 steps x 3 blocks (block 0 dense): attention full at steps 0, 1, 10, MLP full at 0, 10, MLP mask recomputed every
    other step -- with fused_residual / fused_scatter / fused_topk_delta on vs off."""
    fused = _run_flux_schedule(True)
    plain = _run_flux_schedule(False)
    assert len(fused) == len(plain) == 13 * 3 * 2
    if any(not torch.equal(a, b) for a, b in zip(fused, plain)):
        # the dense layers go through torch's GEMM backend: if THAT is not run-to-run deterministic on this box, a bit
        # comparison across runs says nothing about the fusions
        again = _run_flux_schedule(False)
        if any(not torch.equal(a, b) for a, b in zip(plain, again)):
            pytest.skip("the reference op sequence itself is not run-to-run deterministic here (torch GEMM backend)")
    for i, (a, b) in enumerate(zip(fused, plain)):
        assert a.shape == b.shape and torch.isfinite(a.float()).all(), i
        assert torch.equal(a.view(torch.int16), b.view(torch.int16)), f"output {i} (step {i // 6}, block {(i % 6) // 2}, {'mlp' if i % 2 else 'attn'})"
    # and the sparse steps actually differ from step to step (the schedule is exercised, not a constant)
    assert not torch.equal(fused[6 * 2 + 3], fused[6 * 3 + 3])


def test_ops_are_hipgraph_capturable():
    """Every launch goes to the stream it is given and nothing allocates / synchronises / reads back on the host once the
    per-stream scratch has its size: a sparse-step op sequence (gathered attention with residual, GEMM1 + scatter, GEMM2,
    fused top-k, packed mask -> indices, dense attention with its key-split tail) is captured into a graph and replayed."""
    import math
    import chipmunk_amd  # noqa: F401
    from helpers import randn_bf16, random_index_sets
    dev = torch.device("cuda:0")
    H, n, count = 2, 1152, 384
    G = math.ceil(n / 192)
    q, k, v, base = [randn_bf16(1, H, n, 128, seed=s).to(dev) for s in (1, 2, 3, 4)]
    inds, counts = [t.to(dev) for t in random_index_sets(1, H, G, n, count, n, seed=5)]
    M, K, F = 256, 256, 1024
    x, w1 = randn_bf16(M, K, seed=6, scale=0.5).to(dev), randn_bf16(F, K, seed=7, scale=0.1).to(dev)
    b1, w2t = randn_bf16(F, seed=8, scale=0.1).to(dev), randn_bf16(F, K, seed=9, scale=0.1).to(dev)
    cache0, out0 = randn_bf16(F, M, seed=10, scale=0.3).to(dev), randn_bf16(M, K, seed=11).to(dev)
    bm, bmc0 = randn_bf16(1, M // 128, F, seed=12).to(dev), randn_bf16(1, M // 128, F, seed=13).to(dev)
    mask = (torch.rand(1, H, G, n, generator=torch.Generator().manual_seed(14)) < 0.2).to(dev)
    packed = torch.ops.chipmunk.bitpack(mask)

    def step(cache, out, bmc):
        minds = torch.empty(1, M // 128, F, dtype=torch.int32, device=dev)
        mcnt = torch.empty(1, M // 128, dtype=torch.int32, device=dev)
        torch.ops.chipmunk.topk_delta_indices(bm, bmc, minds, mcnt, 0.7, 256, 0.0)
        c = torch.empty(M, F, dtype=torch.bfloat16, device=dev)
        torch.ops.chipmunk.csp_mlp_mm1_scatter(x, w1, c, b1, cache, minds[0], mcnt[0])
        torch.ops.chipmunk.csp_mlp_mm2(c, w2t, minds[0], mcnt[0], out)
        i2, c2 = torch.ops.chipmunk.packed_mask_to_indices(packed, list(mask.shape), 128, 192)
        o = torch.ops.chipmunk.csp_attn_out(q, k, v, base, inds, counts, 1)
        o2 = torch.ops.chipmunk.csp_128_attn(q, k, v, i2, c2)
        od, l = torch.ops.chipmunk.dense_attn(q, k, v)
        return o, o2, od, l

    eager_state = [cache0.clone(), out0.clone(), bmc0.clone()]
    eager = step(*eager_state)          # also brings the scratch to its final size
    torch.cuda.synchronize()
    graph_state = [cache0.clone(), out0.clone(), bmc0.clone()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(cache0.clone(), out0.clone(), bmc0.clone())   # scratch of the capture stream
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        captured = step(*graph_state)
    for t, src in zip(graph_state, (cache0, out0, bmc0)):
        t.copy_(src)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, captured):
        assert torch.equal(a, b)
    for a, b in zip(eager_state, graph_state):
        assert torch.equal(a, b)


def test_step_cache_and_sparse_attention_over_the_shipped_skip_schedule(fresh_config):
    """The transformer loop of the reference's HunyuanVideo model (examples/hunyuan/hyvideo/modules/models.py:732-741 skip check,
    :796-835 block loop + store) on the GPU modules: 60 SparseDiffAttn layers (HIP kernels, tiny sequence) + StepCache over
    the shipped schedule (full steps {0, 1, 10, 40}; skipped {7, 11, 13, ...}) for inference steps 0..24.
    * every computed (step, layer) sees exactly the counter coordinates and full-step flag of the reference's odometer
      (tests/golden/layer_counter.pt, generated from the imported reference);
    * a skipped step advances the odometer by one model invocation without touching a layer and returns the stored state;
    * sparse steps on unchanged q, k, v reproduce the dense output (cache + delta) to bf16 precision."""
    import os
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util import StepCache
    from chipmunk_amd.util import config as cfgmod
    from chipmunk_amd.util.layer_counter import LayerCounter
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgmod.load_from_file(os.path.join(root, "configs", "hunyuan_c3.yml"))
    cfg = fresh_config
    cfg["steps"] = 50
    assert cfg["step_caching"]["is_enabled"]
    skip = cfg["step_caching"]["skip_step_schedule"]
    gold = torch.load(os.path.join(root, "tests", "golden", "layer_counter.pt"))["hunyuan"]   # rows: (step, layer, sub, inv', full_attn, full_mlp)
    dev = torch.device("cuda:0")
    L, H, vid, txt = 60, 2, (4, 6, 16), 64
    N = vid[0] * vid[1] * vid[2] + txt
    g = torch.Generator().manual_seed(11)
    q, k, v = [torch.randn(1, H, N, 128, generator=g).to(torch.bfloat16).to(dev) for _ in range(3)]
    layers = []
    for _ in range(L):
        layer_num, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
        layers.append(SparseDiffAttn(layer_num, counter))
    layers[0].initialize_static_mask(vid, txt, H, dev)
    cache = StepCache(counter)
    dense = torch.ops.chipmunk.dense_attn(q, k, v)[0]
    computed, skipped, last_hidden = [], [], None
    for step in range(25):
        assert (counter.cur_inference_step, counter.cur_layer, counter.cur_layer_submodule) == (step, 0, 0)
        if cache.should_skip(step):
            assert step in skip
            out = cache.skip()
            assert torch.equal(out, last_hidden) and out.data_ptr() == stored_ptr, "a skipped step returns the stored state"
            skipped.append(step)
            continue
        assert step not in skip
        hidden = None
        for li, layer in enumerate(layers):
            row = gold[step * L + li]
            assert (counter.cur_inference_step, counter.cur_layer, counter.cur_layer_submodule) == tuple(int(x) for x in row[:3])
            assert int(counter.should_do_full_attn_step()) == int(row[4])
            if step > 0 or li > 0:
                layer.storage.load_async_wait()
            layers[(li + 1) % L].storage.load_async()
            hidden = layer(q, k, v)
            assert counter.cur_model_invocation_per_step == int(row[3])
            if li in (0, 1, 2, 31, 59):
                atol = 2e-2 if li < 2 or row[4] else 6e-2     # dense layers / full steps are the dense kernel; sparse = cache + delta
                assert (hidden.float() - dense.float()).abs().max().item() < atol, (step, li)
        cache.store(hidden)
        last_hidden, stored_ptr = hidden.clone(), cache._cache[0].data_ptr()
        computed.append(step)
    torch.cuda.synchronize()
    assert skipped == sorted(s for s in skip if s < 25) and len(computed) + len(skipped) == 25
    assert set(computed) >= {0, 1, 10}
