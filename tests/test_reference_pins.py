"""Parity pins produced by the REFERENCE'S OWN code (tests/golden/kernel_pins.pt, module_runs.pt['c1']; generator:
tests/golden/make_golden.py, run in the build container with TRITON_INTERPRET=1):

* GEMM2 and fp8 GEMM1: the reference's Triton kernels (triton/csp_mlp_mm2.py:26-129, triton/csp_mlp_mm1.py:37-164)
  executed by Triton's CPU interpreter;
* mask -> indices: counts and kept set of the reference's ``masktoinds`` (ops/voxel.py:161-180);
* top-k mask: the reference's ``SparseDiffAttn.random_and_topk`` (modules/attn.py:76-82) with the random draw pinned;
* C1 dense eager outputs of the reference modules (its CPU / eager path).

CPU tests check the ORACLE (and the Python mirror) against the pins; ``-m gpu`` tests check the HIP kernels against the
same pins.  Inputs are regenerated from the recorded seeds.  Tolerances: bf16 results of differently ordered fp32
accumulations ``atol = rtol = 2e-2``; integer results exact."""
import os

import pytest
import torch

import oracle
from helpers import assert_close_bf16

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


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


@pytest.fixture(scope="module")
def pins():
    return torch.load(os.path.join(GOLD, "kernel_pins.pt"), weights_only=False)


def _mm2_inputs(p):
    M, F, N2 = p["shape"]
    (s0, s1, s2), (c0, c1, c2) = p["seeds"], p["scales"]
    return _seeded((M, F), s0, c0), _seeded((F, N2), s1, c1), _seeded((M, N2), s2, c2)


def _mm1_inputs(p):
    M, K, F = p["shape"]
    (s0, s1, s2, s3), (c0, c1, c2, c3) = p["seeds"], p["scales"]
    a8 = _seeded((M, K), s0, c0).to(torch.float8_e4m3fn)
    b8 = _seeded((F, K), s1, c1).to(torch.float8_e4m3fn)
    return a8, b8, _seeded((F,), s2, c2), _seeded((F, M), s3, c3)


def _check_mm1(p, packed, cache, what):
    M, K, F = p["shape"]
    for g, cnt in enumerate(p["counts"].tolist()):
        rows = slice(g * 128, (g + 1) * 128)
        assert_close_bf16(packed[rows, :cnt], p["packed"][rows, :cnt], what=f"{what}: packed deltas of group {g}")
    assert_close_bf16(cache, p["cache"], what=f"{what}: activation cache after the call")
    # columns that were not selected keep their old cache value bit for bit
    old = _mm1_inputs(p)[3]
    for g, cnt in enumerate(p["counts"].tolist()):
        untouched = p["indices"][g, cnt:].long()
        assert torch.equal(cache[untouched][:, g * 128:(g + 1) * 128].cpu(), old[untouched][:, g * 128:(g + 1) * 128])


# ------------------------------------------------------------------------------------------------ oracle vs reference
def test_oracle_mm2_matches_reference_triton_kernel(pins):
    p = pins["mm2"]
    a, b, c = _mm2_inputs(p)
    oracle.csp_mlp_mm2(a, b, c, p["indices"], p["counts"])
    assert_close_bf16(c, p["out"], what="oracle GEMM2 vs reference Triton kernel")
    assert (c.view(torch.int16) == p["out"].view(torch.int16)).float().mean() > 0.97   # same roundings almost everywhere


def test_oracle_fp8_mm1_matches_reference_triton_kernel(pins):
    p = pins["mm1_fp8"]
    a8, b8, bias, cache = _mm1_inputs(p)
    packed = torch.zeros(p["shape"][0], p["shape"][2], dtype=torch.bfloat16)
    oracle.csp_mlp_mm1_fp8(a8, b8, packed, bias, cache, p["indices"], p["counts"], p["scale_a"].item(), p["scale_b"].item(),
                           update_cache=True)
    _check_mm1(p, packed, cache, "oracle fp8 GEMM1 vs reference Triton kernel")


def test_oracle_mask_to_indices_matches_reference_masktoinds(pins):
    p = pins["masktoinds"]
    inds, counts = oracle.mask_to_indices(p["mask"], 128, 192)
    assert torch.equal(counts, p["counts"])
    pop = p["popcount"]
    for idx in torch.cartesian_prod(*[torch.arange(n) for n in p["mask"].shape[:-1]]):
        i = tuple(idx.tolist())
        n = int(pop[i])
        assert torch.equal(inds[i][:n].sort().values, p["kept_sorted"][i][:n]), i
        pad = inds[i][n:int(counts[i])]          # padding = distinct columns that are NOT set (both implementations)
        if p["mask"].shape[-1] - n >= pad.numel():   # (an all-True row has nothing to pad with)
            assert not p["mask"][i][pad.long()].any() and pad.unique().numel() == pad.numel()


def test_mirror_random_and_topk_matches_reference_method(pins, fresh_config):
    from chipmunk_amd.modules import attn as mattn
    from chipmunk_amd.ops.bitpack import bitpack, bitunpack
    from chipmunk_amd.util.layer_counter import LayerCounter
    p = pins["random_and_topk"]
    fresh_config["offloading"]["global_disable_offloading"] = True
    mattn.singleton_static_mask = bitunpack(p["static_mask_packed"], p["static_shape"])
    mattn.singleton_video_query_groups = p["groups"]
    layer = mattn.SparseDiffAttn(0, LayerCounter(1, 1))
    real = torch.randint
    torch.randint = lambda lo, hi, shape, **k: torch.ones(shape, dtype=k.get("dtype", torch.int64))
    try:
        mask = layer.random_and_topk(p["cs"], p["k"])
    finally:
        torch.randint = real
    assert torch.equal(bitpack(mask)[0], p["mask_norand_packed"])


def test_oracle_dense_path_matches_reference_c1_eager_outputs():
    """BASELINE.json configs[0]: the reference modules' dense CPU/eager outputs (modules/attn.py:193-194, mlp.py:33-34)."""
    gold = torch.load(os.path.join(GOLD, "module_runs.pt"), weights_only=False)["c1"]
    q, k, v = [_seeded((1, 8, 256, 128), 70 + j) for j in range(3)]
    o, _ = oracle.dense_attn(q, k, v)
    assert_close_bf16(o[:, :, ::4], gold["attn_out"], what="oracle dense_attn vs reference eager SDPA (C1)")
    x = _seeded((1, 256, 1024), 73)
    fc1, fc2 = seeded_linear(1024, 4096, 74), seeded_linear(4096, 1024, 75)
    y = oracle.dense_mlp(x[0], fc1.weight.data, fc1.bias.data, fc2.weight.data, fc2.bias.data)
    assert_close_bf16(y[::4], gold["mlp_out"][0], what="oracle dense_mlp vs reference eager MLP (C1)")


# ------------------------------------------------------------------------------------------------ HIP vs reference
@pytest.fixture()
def dev():
    import chipmunk_amd  # noqa: F401
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_hip_mm2_matches_reference_triton_kernel(pins, dev):
    p = pins["mm2"]
    a, b, c = [t.to(dev) for t in _mm2_inputs(p)]
    torch.ops.chipmunk.csp_mlp_mm2(a, b, p["indices"].to(dev), p["counts"].to(dev), c)
    assert_close_bf16(c, p["out"], what="HIP GEMM2 vs reference Triton kernel")
    # the reference-named fused entry: scatter-add + GEMM2 (the GEMM half must give the same result)
    a, b, c = [t.to(dev) for t in _mm2_inputs(p)]
    M, F, _ = p["shape"]
    unpacked = torch.zeros(1, F, M, dtype=torch.bfloat16, device=dev)
    torch.ops.chipmunk.csp_mlp_mm2_and_scatter_add(a[None], unpacked, p["indices"].to(dev)[None], p["counts"].to(dev)[None],
                                                   a[None], b[None], c[None], 6, 0)
    assert_close_bf16(c, p["out"], what="HIP csp_mlp_mm2_and_scatter_add (GEMM half) vs reference Triton kernel")


@pytest.mark.gpu
def test_hip_fp8_mm1_matches_reference_triton_kernel(pins, dev):
    p = pins["mm1_fp8"]
    a8, b8, bias, cache = [t.to(dev) for t in _mm1_inputs(p)]
    packed = torch.zeros(p["shape"][0], p["shape"][2], dtype=torch.bfloat16, device=dev)
    torch.ops.chipmunk.csp_mlp_mm1_fp8(a8, b8, packed, bias, cache, p["indices"].to(dev), p["counts"].to(dev),
                                       p["scale_a"].to(dev), p["scale_b"].to(dev), True)
    _check_mm1(p, packed.cpu(), cache.cpu(), "HIP fp8 GEMM1 vs reference Triton kernel")


@pytest.mark.gpu
def test_hip_mask_to_indices_matches_reference_masktoinds(pins, dev):
    p = pins["masktoinds"]
    mask = p["mask"].to(dev)
    inds, counts = torch.ops.chipmunk.mask_to_indices(mask, 128, 192)
    assert torch.equal(counts.cpu(), p["counts"])
    sinds, scounts = torch.ops.chipmunk.mask_to_sorted_indices(mask, list(mask.shape), 128, 192)
    # the bit-packed form needs whole bytes per row (n % 8 == 0): pad the columns with False
    padded = torch.nn.functional.pad(mask, (0, (-mask.shape[-1]) % 8))
    packed = torch.ops.chipmunk.bitpack(padded)
    pinds, pcounts = torch.ops.chipmunk.packed_mask_to_indices(packed, list(padded.shape), 128, 192)
    assert torch.equal(scounts.cpu(), p["counts"]) and torch.equal(pcounts.cpu(), p["counts"])
    inds, sinds, pinds = inds.cpu(), sinds.cpu(), pinds.cpu()
    for idx in torch.cartesian_prod(*[torch.arange(n) for n in p["mask"].shape[:-1]]):
        i = tuple(idx.tolist())
        n = int(p["popcount"][i])
        want = p["kept_sorted"][i][:n]
        assert torch.equal(inds[i][:n].sort().values, want) and torch.equal(pinds[i][:n].sort().values, want), i
        assert torch.equal(sinds[i][:n], want), i          # the sorted form emits the kept set ascending


@pytest.mark.gpu
def test_hip_topk_mask_matches_reference_random_and_topk(pins, dev):
    from chipmunk_amd.ops.bitpack import bitunpack
    p = pins["random_and_topk"]
    static = bitunpack(p["static_mask_packed"], p["static_shape"]).to(dev)
    mask = torch.ops.chipmunk.topk_mask(p["cs"].to(dev), p["k"], 0.0, p["groups"].to(dev), static)
    want = bitunpack(p["mask_norand_packed"], p["mask_shape"])
    assert torch.equal(mask.cpu(), want)
    # with the random part on: a superset of the pinned mask, ~1 % extra columns in the sparse groups, different per launch
    m1 = torch.ops.chipmunk.topk_mask(p["cs"].to(dev), p["k"], 0.01, p["groups"].to(dev), static).cpu()
    m2 = torch.ops.chipmunk.topk_mask(p["cs"].to(dev), p["k"], 0.01, p["groups"].to(dev), static).cpu()
    assert (m1 | want).equal(m1) and (m2 | want).equal(m2)
    extra = (m1 & ~want).float().sum() / want.numel()
    assert 0.002 < extra < 0.02, extra
    assert not torch.equal(m1, m2), "every launch draws a fresh random set (ADVICE r1: a static subset defeats the refresh)"
    import chipmunk_amd.ops as ops
    ops.manual_seed(7)
    a = torch.ops.chipmunk.topk_mask(p["cs"].to(dev), p["k"], 0.01, p["groups"].to(dev), static).cpu()
    ops.manual_seed(7)
    b = torch.ops.chipmunk.topk_mask(p["cs"].to(dev), p["k"], 0.01, p["groups"].to(dev), static).cpu()
    assert torch.equal(a, b), "same seed + same launch order = same random columns"


@pytest.mark.gpu
def test_hip_dense_path_matches_reference_c1_eager_outputs(dev):
    gold = torch.load(os.path.join(GOLD, "module_runs.pt"), weights_only=False)["c1"]
    q, k, v = [_seeded((1, 8, 256, 128), 70 + j).to(dev) for j in range(3)]
    o, _ = torch.ops.chipmunk.dense_attn(q, k, v)
    assert_close_bf16(o[:, :, ::4], gold["attn_out"], what="HIP dense_attn vs reference eager SDPA (C1)")
