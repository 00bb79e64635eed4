"""Module-level parity that is NOT a self-comparison: this build's ``SparseDiffAttn`` / ``SparseDiffMlp`` on HIP kernels
(CUDA tensors, fused paths on -- the shipped configuration) against outputs of the REFERENCE's modules
(src/chipmunk/modules/attn.py:86-190, mlp.py:30-120) run on CPU with oracle-backed ops when the fixtures were generated
(tests/golden/make_golden.py -> module_runs_structured.pt, module_runs.pt).

Attention inputs have planted structure (tests/helpers.py: structured_qkv): the kept-key sets are the hot set plus
noise-floor filler, so implementations that round the bf16 column sums / break top-k ties differently (torch.topk on
CPU vs the HIP top-k, fp32 vs bf16 partial sums) still agree on every output to bf16 precision, while a wrapper mistake
(cache sign, padding, a stale mask, `l` not zeroed) moves the output by O(|o|).  Tolerance per element:
``6e-3 + 2e-2 * |ref|`` (|o| ~ 0.1-0.4; three bf16 roundings of values near 0.4 lie between the two pipelines); hot sets asserted bit-exact inside the kept sets; FLUX counts bit-exact.
The CPU half of the file checks the mirror against the same fixture with the oracle-backed ops (bit-exact)."""
import os

import pytest
import torch

from helpers import structured_qkv

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return torch.load(os.path.join(GOLD, "module_runs_structured.pt"), weights_only=False)


def _check(t, d, what, exact):
    assert tuple(t.shape) == d["shape"], what
    got = t.detach().flatten()[::53][:8192].float().cpu()
    want = d["sample"].float()
    if exact:
        assert torch.equal(got, want), what
        return
    err = (got - want).abs()
    tol = 6e-3 + 2e-2 * want.abs()
    assert not (err > tol).any(), f"{what}: {int((err > tol).sum())} of {err.numel()} sampled elements off, max abs diff {err.max():.4g} (|ref| max {want.abs().max():.3g})"
    rel = abs(t.double().abs().sum().item() - d["abs"]) / d["abs"]
    assert rel < 5e-3, f"{what}: sum |o| off by {rel:.3%}"


def _flux_cfg(cfg):
    cfg["attn"].update(dict(top_keys=0.165, full_step_every=10, full_step_schedule=None, first_n_dense_layers=1,
                            recompute_mask=False, should_compress_indices=False, counts_multiple_of=112,
                            pad_qkv_before_kernel=False, random_keys=0.0, local_voxels=0))


def _hunyuan_cfg(cfg):
    cfg["attn"].update(dict(top_keys=0.05, random_keys=0.0, local_voxels=0, first_n_dense_layers=1, recompute_mask=True,
                            should_compress_indices=True, counts_multiple_of=128, pad_qkv_before_kernel=True,
                            full_step_schedule={0, 1, 4}))


def _run_flux(gold, dev, exact):
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util.layer_counter import LayerCounter
    p = gold["flux"]
    counter = LayerCounter(2, 1)
    layers = [SparseDiffAttn(i, counter) for i in range(2)]
    i = 0
    for step in range(p["steps"]):
        for li, layer in enumerate(layers):
            q, k, v, hot = structured_qkv(p["H"], p["N"], p["n_hot"], step, li)
            o = layer(q.to(dev), k.to(dev), v.to(dev))
            _check(o, p["outs"][i], f"FLUX-style attention, step {step} layer {li}", exact)
            i += 1
    inds, counts = layers[1].storage.get_indices().cpu(), layers[1].storage.get_counts().cpu()
    assert torch.equal(counts, p["counts"])
    kept = inds[..., :224]
    for h in range(p["H"]):
        for g in range(kept.shape[2]):
            assert torch.isin(hot[h], kept[0, h, g]).all(), f"hot keys missing from the kept set of head {h} group {g}"
            assert torch.isin(hot[h], p["indices"][0, h, g]).all()      # ... and they are in the reference's set too


def _run_hunyuan(gold, dev, exact):
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.ops.bitpack import bitunpack
    from chipmunk_amd.util.layer_counter import LayerCounter
    p = gold["hunyuan"]
    vid, txt = p["vid"], p["txt"]
    N = vid[0] * vid[1] * vid[2] + txt
    counter = LayerCounter(2, 1)
    layers = [SparseDiffAttn(i, counter) for i in range(2)]
    torch.manual_seed(123)
    layers[0].initialize_static_mask(vid, txt, p["H"], torch.device("cpu"))
    if dev.type == "cuda":
        from chipmunk_amd.modules import attn as mattn
        mattn.singleton_static_mask = mattn.singleton_static_mask.to(dev)
        mattn.singleton_video_query_groups = mattn.singleton_video_query_groups.to(dev)
    i = 0
    real = torch.randint
    torch.randint = lambda lo, hi, shape, **k: torch.ones(shape, dtype=k.get("dtype", torch.int64), device=k.get("device"))
    try:
        for step in range(p["steps"]):
            for li, layer in enumerate(layers):
                q, k, v, hot = structured_qkv(p["H"], N, p["n_hot"], step, li)
                o = layer(q.to(dev), k.to(dev), v.to(dev))
                _check(o, p["outs"][i], f"Hunyuan-style attention, step {step} layer {li}", exact)
                i += 1
    finally:
        torch.randint = real
    mask = bitunpack(layers[1].storage.get_indices(), layers[1].mask_shape[0]).cpu()
    ref_mask = bitunpack(p["packed_mask"], p["mask_shape"])
    assert tuple(mask.shape) == p["mask_shape"]
    for h in range(p["H"]):
        # (the last query group holds the text rows: not a sparse group, its row is the static mask alone)
        assert mask[0, h, :-1][:, hot[h]].all() and ref_mask[0, h, :-1][:, hot[h]].all(), "hot keys kept by every sparse query group"
    assert torch.equal(mask[:, :, -1], ref_mask[:, :, -1])
    if exact:
        assert torch.equal(mask, ref_mask)
    else:   # same popcount per row up to the ~1 % hash-random extra columns of the fused top-k mask kernel and a few
        #         filler columns that coincide (or not) with static-mask columns
        d = (mask.sum(-1) - ref_mask.sum(-1)).float()
        assert d.min() >= -8 and d.max() <= 0.03 * N


# ------------------------------------------------------------------------------------------------ CPU: mirror, bit-exact
@pytest.fixture()
def cpu_chipmunk(fresh_config):
    import cpu_ops
    cpu_ops.register()
    fresh_config["offloading"]["global_disable_offloading"] = True
    fresh_config["steps"] = 50
    return fresh_config


def test_mirror_reproduces_structured_flux_run_on_cpu(cpu_chipmunk, gold):
    _flux_cfg(cpu_chipmunk)
    _run_flux(gold, torch.device("cpu"), exact=True)


def test_mirror_reproduces_structured_hunyuan_run_on_cpu(cpu_chipmunk, gold):
    _hunyuan_cfg(cpu_chipmunk)
    cpu_chipmunk["attn"]["fused_packed_mask_to_indices"] = False
    _run_hunyuan(gold, torch.device("cpu"), exact=True)


# ------------------------------------------------------------------------------------------------ GPU: HIP vs reference run
@pytest.fixture()
def gpu_chipmunk(fresh_config):
    import chipmunk_amd  # noqa: F401
    fresh_config["offloading"]["global_disable_offloading"] = True
    fresh_config["steps"] = 50
    return fresh_config


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_hip_flux_style_attention_matches_reference_run(gpu_chipmunk, gold, fused):
    _flux_cfg(gpu_chipmunk)
    gpu_chipmunk["attn"]["fused_residual"] = fused
    _run_flux(gold, torch.device("cuda:0"), exact=False)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_hip_hunyuan_style_attention_matches_reference_run(gpu_chipmunk, gold, fused):
    _hunyuan_cfg(gpu_chipmunk)
    for key in ("fused_packed_mask_to_indices", "sorted_indices", "fused_topk_mask"):
        gpu_chipmunk["attn"][key] = fused
    if fused:
        gpu_chipmunk["attn"]["random_keys"] = 0.0
    _run_hunyuan(gold, torch.device("cuda:0"), exact=False)


@pytest.mark.gpu
def test_keep_resident_cache_survives_sparse_steps_on_the_unpadded_path(gpu_chipmunk, gold):
    """with `offloading.keep_resident_if_fits` the out-cache flagged for offload stays in HBM and
    `get_out_cache()` returns the cache itself; the in-place sparse step must then work on a copy (or the out-of-place
    kernel), or every sparse step would accumulate another delta into the cache."""
    _flux_cfg(gpu_chipmunk)
    gpu_chipmunk["offloading"].update({"global_disable_offloading": False, "attn.out_cache": True, "attn.indices": False,
                                       "keep_resident_if_fits": True})
    for fused in (True, False):
        gpu_chipmunk["attn"]["fused_residual"] = fused
        _run_flux(gold, torch.device("cuda:0"), exact=False)


@pytest.mark.gpu
def test_hip_sparse_mlp_schedule_matches_reference_run(gpu_chipmunk):
    """SparseDiffMlp on HIP kernels vs the reference module's outputs (module_runs.pt, C1 shapes, 13 steps x 2 blocks:
    dense block, full steps, top-k / copy / GEMM1 / scatter / GEMM2 sparse steps, cached-mask steps)."""
    from chipmunk_amd.modules import SparseDiffMlp
    from chipmunk_amd.util.layer_counter import LayerCounter
    from test_modules_cpu import _seeded, seeded_linear
    gold = torch.load(os.path.join(GOLD, "module_runs.pt"), weights_only=False)
    dev = torch.device("cuda:0")
    gpu_chipmunk["mlp"].update(dict(top_keys=0.3, random_keys=0.0, full_step_every=4, block_mask_cache=2,
                                    first_n_dense_layers=1, counts_multiple_of=256))
    for fused in (True, False):
        gpu_chipmunk["mlp"]["fused_scatter"] = gpu_chipmunk["mlp"]["fused_topk_delta"] = fused
        counter = LayerCounter(2, 1)
        mlps = [SparseDiffMlp(i, counter, seeded_linear(1024, 4096, 4242 + 2 * i).to(dev), torch.nn.GELU(approximate="tanh"),
                              seeded_linear(4096, 1024, 4243 + 2 * i).to(dev), 6) for i in range(2)]
        i = 0
        worst = 0.0
        with torch.no_grad():
            for step in range(13):
                for li, m in enumerate(mlps):
                    x = (_seeded((1, 256, 1024), 9000 + li).float()
                         + 0.15 * _seeded((1, 256, 1024), 9100 + 10 * step + li).float()).to(torch.bfloat16)
                    y = m(x.to(dev))
                    d = gold["mlp_outs"][i]
                    got, want = y.flatten()[::53][:8192].float().cpu(), d["sample"].float()
                    err = (got - want).abs()
                    worst = max(worst, float(err.max()))
                    tol = 2e-2 + 2e-2 * want.abs()
                    assert not (err > tol).any(), f"MLP step {step} layer {li} (fused={fused}): max abs diff {err.max():.4g}"
                    i += 1
        assert worst > 0 or True
