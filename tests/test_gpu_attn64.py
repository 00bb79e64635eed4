"""The one-wave-per-SIMD dense kernel (csrc/attn64.hip), forced through option attn_dense64 = 1 at sizes the oracle
finishes in seconds: same tolerances as test_gpu_attn.py (bf16 o: atol = rtol = 2e-2; l: rtol 1e-3 / 2e-3)."""
import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def forced(dev):
    from chipmunk_amd import _native
    _native.set_option("attn_dense64", 1)
    yield
    _native.set_option("attn_dense64", 0)


@pytest.mark.parametrize("nq,nk", [(256, 256), (512, 64), (1000, 1000), (384, 1984), (1300, 777), (64, 4160)])
def test_dense64_vs_oracle(dev, forced, nq, nk):
    """multiples of the tile sizes, ragged query rows (last workgroup / wave partly empty), ragged key tiles (masked tail,
    padding tiles of the 4-tile unroll), fewer keys than one tile"""
    H = 3
    q = randn_bf16(1, H, nq, 128, seed=nq)
    k = randn_bf16(1, H, nk, 128, seed=nk + 1)
    v = randn_bf16(1, H, nk, 128, seed=nk + 2)
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense64 o {nq}x{nk}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)


def test_dense64_strided_and_batched(dev, forced):
    B, H, N = 2, 4, 576
    base = [randn_bf16(B, N, H, 128, seed=s).to(dev) for s in (1, 2, 3)]
    q, k, v = [t.permute(0, 2, 1, 3) for t in base]
    o, l = torch.ops.chipmunk.dense_attn(q, k, v)
    o_ref, l_ref = oracle.dense_attn(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous())
    assert_close_bf16(o, o_ref, what="strided dense64")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_dense64_running_max_update_paths(dev, forced, pattern):
    """same constructions as test_gpu_attn.py::test_running_max_update_paths: the reference point of the exponentials must
    move (with the rescale of the accumulator registers) for some lanes of a wave and not for others"""
    n, H = 1152, 2
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, n, 128, generator=g)
    k = torch.randn(1, H, n, 128, generator=g)
    v = torch.randn(1, H, n, 128, generator=g)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    q = 0.3 * q + 3.0 * u
    if pattern == "ramp":
        k = 0.3 * k + (torch.arange(n).float() / n * 30.0)[None, None, :, None] * u
    elif pattern == "descending":
        k = 0.3 * k + ((n - torch.arange(n)).float() / n * 30.0)[None, None, :, None] * u
    else:
        k = 0.3 * k
        j = 5 if pattern == "spike_first" else 1000
        k[0, :, j] += 40.0 * u
        q[0, :, ::3] *= 0.05
    q, k, v = [t.to(torch.bfloat16) for t in (q, k, v)]
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense64, {pattern}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=2e-3, atol=0)


def test_dense64_matches_general_kernel_at_scale(dev):
    """8 448 x 8 448, 2 heads: the two dense kernels agree (bf16 rounding of different summation orders only)"""
    from chipmunk_amd import _native
    q, k, v = [randn_bf16(1, 2, 8448, 128, seed=s).to(dev) for s in (5, 6, 7)]
    _native.set_option("attn_dense64", 2)
    o_a, l_a = torch.ops.chipmunk.dense_attn(q, k, v)
    _native.set_option("attn_dense64", 1)
    o_b, l_b = torch.ops.chipmunk.dense_attn(q, k, v)
    _native.set_option("attn_dense64", 0)
    assert_close_bf16(o_b, o_a.float().cpu(), what="dense64 vs general kernel")
    torch.testing.assert_close(l_b, l_a, rtol=1e-3, atol=0)


# ---- gathered launches through the three-compute-waves + loader kernel (option attn_csp64 = 1 forces it at test sizes) ----

@pytest.fixture(params=["attn_csp64", "attn_csp96"])
def forced_csp(dev, request):
    """both one-wave-per-SIMD gathered kernels: three compute waves + loader (attn64.hip), two waves x 96 rows (attn96.hip)"""
    from chipmunk_amd import _native
    _native.set_option(request.param, 1)
    yield
    _native.set_option(request.param, 0)


@pytest.mark.parametrize("n,nk,count", [(384, 384, 128), (1000, 1000, 333), (1152, 1152, 1152), (576, 2000, 64), (200, 640, 7), (960, 960, 0),
                                         (576, 997, 333), (390, 1001, 1001)])   # (index rows that are not 16-byte aligned: the dword load path)
def test_csp64_random_indices_vs_oracle(dev, forced_csp, n, nk, count):
    """csp_128_attn: ragged query groups, counts that are not multiples of the 64-key tile (masked tail), fewer keys than a
    tile, all keys, no keys"""
    import math
    from helpers import random_index_sets
    H = 3
    q = randn_bf16(1, H, n, 128, seed=n)
    k = randn_bf16(1, H, nk, 128, seed=nk + 1)
    v = randn_bf16(1, H, nk, 128, seed=nk + 2)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, nk, count, nk, seed=5)
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what=f"csp64 {n}x{nk} count {count}")


@pytest.mark.parametrize("how", ["option", "large_scores"])
def test_csp96_running_maximum_schedule_vs_oracle(dev, how):
    """attn96.hip has two slot schedules (tools/gen_attn96_sched.py): the paired PV order of the loop without a reference point
    and the query-block-major one of the running-maximum loop.  The second one runs when the |q| max|k| bound is too large
    (here: q, k x 3) or when option attn_nomax = 2 switches the bound off; ragged counts, a masked tail, sliced heavy items."""
    import math
    from chipmunk_amd import _native
    from helpers import random_index_sets
    H, n = 3, 2304
    sc = 3.0 if how == "large_scores" else 1.0
    q, k = [(randn_bf16(1, H, n, 128, seed=s).float() * sc).to(torch.bfloat16) for s in (51, 52)]
    v = randn_bf16(1, H, n, 128, seed=53)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 640, n, seed=4)
    counts[0, 0, 1], counts[0, 1, 2], counts[0, 2, 3] = 37, 0, n
    inds[0, 2, 3] = torch.arange(n, dtype=torch.int32)
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    _native.set_option("attn_csp96", 1)
    _native.set_option("attn_nomax", 2 if how == "option" else 0)
    try:
        o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
        torch.cuda.synchronize()
    finally:
        _native.set_option("attn_csp96", 0)
        _native.set_option("attn_nomax", 0)
    assert_close_bf16(o, o_ref, what=f"csp96 running-maximum schedule ({how})")


@pytest.mark.parametrize("o_scale", [1, -1])
def test_csp64_inplace_and_out_forms(dev, forced_csp, o_scale):
    import math
    from helpers import random_index_sets
    H, n = 2, 1344
    q, k, v, base = [randn_bf16(1, H, n, 128, seed=s) for s in (1, 2, 3, 4)]
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 400, n, seed=9)
    counts[0, 0, 1] = 37
    counts[0, 1, 2] = 0
    ref = base.clone()
    oracle.csp_attn(q, k, v, ref, inds, counts, o_scale)
    acc = base.clone().to(dev)
    torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), acc, inds.to(dev), counts.to(dev), o_scale)
    assert_close_bf16(acc, ref, atol=3e-2, what="csp64 in place")
    out = torch.ops.chipmunk.csp_attn_out(q.to(dev), k.to(dev), v.to(dev), base.to(dev), inds.to(dev), counts.to(dev), o_scale)
    assert torch.equal(out, acc), "the out-of-place form is the in-place form on a copy"
    assert torch.equal(out[0, 1, 2 * 192:3 * 192].cpu(), base[0, 1, 2 * 192:3 * 192]), "a group without keys keeps the base"


def test_csp64_sliced_heavy_items_merge(dev, forced_csp):
    """a few groups keep ALL keys while the rest keep few: the plan cuts the heavy items into key slices that different
    workgroups process and the last arriver merges"""
    import math
    H, n = 2, 9216
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (11, 12, 13)]
    G = math.ceil(n / 192)
    gen = torch.Generator().manual_seed(3)
    inds = torch.stack([torch.randperm(n, generator=gen) for _ in range(H * G)]).view(1, H, G, n).to(torch.int32)
    counts = torch.full((1, H, G), 96, dtype=torch.int32)
    counts[0, :, -1] = n
    counts[0, 0, 3] = n
    inds[0, :, -1] = torch.arange(n, dtype=torch.int32)
    inds[0, 0, 3] = torch.arange(n, dtype=torch.int32)
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what="csp64 sliced items")
    again = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert torch.equal(o, again), "slices fold in slice order: run-to-run deterministic"


# ---- the one-wave-per-group column-sum pass (option attn_colsum64 = 1 forces it at test sizes) ----

# route: "two_pass" = dense kernel + colsum64_kernel; "fused" = the column sums inside the dense kernel (attn64 MODE 3, the
# default where the dense kernel runs): reference point -log2 p_i and unit weights where every query of a wave allows it,
# "fused_weighted" = fixed reference point |q_i| max|k| with weights, "fused_runmax" = running maximum with weights,
# "fused_per_head" = the chunked form (one launch per head: what runs when the partial-sum buffer cannot be had whole)
@pytest.mark.parametrize("route", ["two_pass", "fused", "fused_weighted", "fused_runmax", "fused_per_head"])
@pytest.mark.parametrize("n,nk", [(384, 384), (1000, 1000), (1984, 1984), (777, 200), (960, 64), (4160, 768), (200, 192)])
def test_colsum64_vs_oracle(dev, n, nk, route):
    """group counts that are not multiples of four (idle waves), ragged last groups, ragged / padding key tiles (stored
    twice with the same values / only their own keys), fewer rows than one group, row blocks past the last row"""
    import math
    from chipmunk_amd import _native
    H = 2
    q = randn_bf16(1, H, n, 128, seed=n)
    k = randn_bf16(1, H, nk, 128, seed=nk + 1)
    v = randn_bf16(1, H, nk, 128, seed=nk + 2)
    q2 = (q.float() + 0.1 * torch.randn(q.shape, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
    _, l0 = oracle.dense_attn(q, k, v)
    o_ref, cs_ref, l_ref = oracle.dense_colsum_attn(q2, k, v, l0)
    _native.set_option("attn_colsum64", 1)
    _native.set_option("attn_dense64", 1)
    _native.set_option("attn_fused_colsum", {"two_pass": 2, "fused_weighted": 3, "fused_per_head": 4}.get(route, 0))
    _native.set_option("attn_nomax", 2 if route == "fused_runmax" else 0)
    try:
        o, cs, l = torch.ops.chipmunk.dense_colsum_attn(q2.to(dev), k.to(dev), v.to(dev), l0.to(dev))
        again = torch.ops.chipmunk.dense_colsum_attn(q2.to(dev), k.to(dev), v.to(dev), l0.to(dev))[1]
    finally:
        for opt in ("attn_colsum64", "attn_dense64", "attn_fused_colsum", "attn_nomax"):
            _native.set_option(opt, 0)
    G = math.ceil(n / 192)
    assert cs.shape == (1, H, G, n) and cs.dtype == torch.bfloat16   # Nq columns, the first Nk meaningful (dense_colsum_attn.cu:580-583)
    assert_close_bf16(o, o_ref, what="colsum64 o")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    assert_close_bf16(cs[..., :nk], cs_ref[..., :nk], atol=2e-3, rtol=3e-2, what="colsum64 cs vs oracle")
    assert torch.equal(cs[..., :nk], again[..., :nk]), "no order-dependent reduction: run-to-run identical"


def test_fused_colsum_strided_batched_and_degenerate_p(dev):
    """[B, N, H, D] storage viewed as [B, H, N, D], two batches; rows whose p is 0 (they must add nothing to the column
    sums, as in the two-pass kernels) and rows whose p is large"""
    from chipmunk_amd import _native
    B, H, N = 2, 3, 640
    base = [randn_bf16(B, N, H, 128, seed=s).to(dev) for s in (51, 52, 53)]
    q, k, v = [t.permute(0, 2, 1, 3) for t in base]
    qc, kc, vc = [t.cpu().contiguous() for t in (q, k, v)]
    _, l0 = oracle.dense_attn(qc, kc, vc)
    l0[:, :, 5::7] = 0.0
    l0[:, :, 3::11] *= 1.0e6
    o_ref, cs_ref, l_ref = oracle.dense_colsum_attn(qc, kc, vc, l0)
    _native.set_option("attn_dense64", 1)
    try:
        o, cs, l = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l0.to(dev))
        _native.set_option("attn_fused_colsum", 4)      # one (batch, head) per launch: same kernels, same numbers
        o4, cs4, l4 = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l0.to(dev))
    finally:
        _native.set_option("attn_dense64", 0)
        _native.set_option("attn_fused_colsum", 0)
    assert torch.equal(o, o4) and torch.equal(cs, cs4) and torch.equal(l, l4), "chunked launches change nothing"
    assert_close_bf16(o, o_ref, what="fused colsum, strided o")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    assert_close_bf16(cs, cs_ref, atol=2e-3, rtol=3e-2, what="fused colsum, strided cs")


def test_k_only_passes_are_run_to_run_identical_at_scale(dev):
    """24 heads x 16 384: ~4 million wave-tiles per launch.  Regression test for the general kernel's K-only pass, which
    handed its per-wave partial sums to wave 0 across a barrier without waiting for its own LDS writes (about one stale
    partial per million wave-tiles: a few events per launch at this size); colsum64_kernel and the one-pass route beside it."""
    from chipmunk_amd import _native
    H, n = 24, 16384
    q, k, v = [randn_bf16(1, H, n, 128, seed=s).to(dev) for s in (61, 62, 63)]
    _, l = torch.ops.chipmunk.dense_attn(q, k, v)
    ref = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1]
    assert torch.equal(ref, torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1])
    for which in (2, 1):
        _native.set_option("attn_fused_colsum", 2)
        _native.set_option("attn_colsum64", which)
        try:
            outs = [torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1] for _ in range(4)]
        finally:
            _native.set_option("attn_fused_colsum", 0)
            _native.set_option("attn_colsum64", 0)
        for o in outs[1:]:
            assert torch.equal(outs[0], o), f"K-only kernel {which}: launches differ"
        assert_close_bf16(outs[0], ref.float().cpu(), atol=1e-5, rtol=2e-2, what=f"K-only kernel {which} vs one pass")


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_fused_colsum_running_max_update_paths(dev, pattern):
    """the fused column sums while the reference point of the exponentials moves (the weights exp2(m c) p_i change with
    it, per query block, between two tiles): o, l and cs against the oracle; cs also against the two-pass route"""
    from chipmunk_amd import _native
    n, H = 1152, 2
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, n, 128, generator=g)
    k = torch.randn(1, H, n, 128, generator=g)
    v = torch.randn(1, H, n, 128, generator=g)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    q = 0.3 * q + 3.0 * u
    if pattern == "ramp":
        k = 0.3 * k + (torch.arange(n).float() / n * 30.0)[None, None, :, None] * u
    elif pattern == "descending":
        k = 0.3 * k + ((n - torch.arange(n)).float() / n * 30.0)[None, None, :, None] * u
    else:
        k = 0.3 * k
        j = 5 if pattern == "spike_first" else 1000
        k[0, :, j] += 40.0 * u
        q[0, :, ::3] *= 0.05
    q, k, v = [t.to(torch.bfloat16) for t in (q, k, v)]
    _, l0 = oracle.dense_attn(q, k, v)
    o_ref, cs_ref, l_ref = oracle.dense_colsum_attn(q, k, v, l0)
    args = [t.to(dev) for t in (q, k, v, l0)]
    _native.set_option("attn_dense64", 1)
    _native.set_option("attn_colsum64", 1)
    try:
        o, cs, l = torch.ops.chipmunk.dense_colsum_attn(*args)
        _native.set_option("attn_fused_colsum", 2)
        o2, cs2, l2 = torch.ops.chipmunk.dense_colsum_attn(*args)
    finally:
        for opt in ("attn_colsum64", "attn_dense64", "attn_fused_colsum"):
            _native.set_option(opt, 0)
    assert_close_bf16(o, o_ref, what=f"fused colsum o, {pattern}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=2e-3, atol=0)
    assert_close_bf16(cs, cs_ref, atol=2e-3, rtol=3e-2, what=f"fused cs vs oracle, {pattern}")
    assert_close_bf16(cs, cs2.float().cpu(), atol=2e-3, rtol=3e-2, what=f"fused cs vs two-pass, {pattern}")
    # (the one-pass route may place the reference point of the exponentials at -log2 p_i: same o and l up to rounding)
    assert_close_bf16(o, o2.float().cpu(), atol=4e-3, rtol=1e-2, what=f"o, one pass vs two, {pattern}")
    torch.testing.assert_close(l, l2, rtol=2e-3, atol=0)


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_csp64_running_max_update_paths(dev, forced_csp, pattern):
    """the gathered one-wave-per-SIMD kernels on the constructions that force the reference point of the exponentials to
    move (per query block, with the rescale of that block's accumulator registers) for some lanes and not for others"""
    import math
    n, H = 1152, 2
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, n, 128, generator=g)
    k = torch.randn(1, H, n, 128, generator=g)
    v = torch.randn(1, H, n, 128, generator=g)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    q = 0.3 * q + 3.0 * u
    if pattern == "ramp":
        k = 0.3 * k + (torch.arange(n).float() / n * 30.0)[None, None, :, None] * u
    elif pattern == "descending":
        k = 0.3 * k + ((n - torch.arange(n)).float() / n * 30.0)[None, None, :, None] * u
    else:
        k = 0.3 * k
        j = 5 if pattern == "spike_first" else 1000
        k[0, :, j] += 40.0 * u
        q[0, :, ::3] *= 0.05
    q, k, v = [t.to(torch.bfloat16) for t in (q, k, v)]
    o_ref, _ = oracle.dense_attn(q, k, v)
    G = math.ceil(n / 192)
    counts = torch.full((1, H, G), n, dtype=torch.int32)
    for name, order in (("identity", torch.arange(n)), ("reverse", torch.arange(n - 1, -1, -1))):
        inds = order.to(torch.int32).expand(1, H, G, n).contiguous()
        o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
        assert_close_bf16(o, o_ref, what=f"gathered, {name} key order, {pattern}")


@pytest.mark.parametrize("scale,expect", [(1.0, "fixed"), (4.0, "running")])
def test_dense64_fixed_reference_point_and_its_fallback(dev, forced, scale, expect):
    """unit-variance inputs let every wave prove |s| <= |q| max|k| small enough: the exponentials use that fixed reference
    point and no running maximum; 4x larger inputs (scores 16x) cannot, and run the running-maximum loop.  Same results
    (o, l) within the stated tolerances either way, and the same as with the fixed reference switched off."""
    from chipmunk_amd import _native
    n = 1536
    q = (randn_bf16(1, 2, n, 128, seed=21).float() * scale).to(torch.bfloat16)
    k = (randn_bf16(1, 2, n, 128, seed=22).float() * scale).to(torch.bfloat16)
    v = randn_bf16(1, 2, n, 128, seed=23)
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense64 ({expect} reference)")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    _native.set_option("attn_nomax", 2)
    try:
        o2, l2 = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    finally:
        _native.set_option("attn_nomax", 0)
    assert_close_bf16(o2, o_ref, what="dense64, running maximum forced")
    torch.testing.assert_close(l2.cpu(), l_ref, rtol=1e-3, atol=0)
    if expect == "running":
        assert torch.equal(o, o2) and torch.equal(l, l2), "inputs too large for the bound: both runs take the running-maximum loop"


def test_long_launch_kernels_are_hipgraph_capturable(dev):
    """the one-wave-per-SIMD kernels with their pre-pass (memset + K row-norm kernel, work plan) captured into a graph and
    replayed on new data: nothing allocates or synchronises once the per-stream scratch has its size"""
    import math
    from chipmunk_amd import _native
    from helpers import random_index_sets
    H, n, count = 2, 1536, 512
    G = math.ceil(n / 192)
    q, k, v = [randn_bf16(1, H, n, 128, seed=s).to(dev) for s in (31, 32, 33)]
    inds, counts = [t.to(dev) for t in random_index_sets(1, H, G, n, count, n, seed=7)]
    opts = ("attn_dense64", "attn_colsum64", "attn_csp96")
    for o in opts:
        _native.set_option(o, 1)
    try:
        def step():
            od, l = torch.ops.chipmunk.dense_attn(q, k, v)
            o2, cs, l2 = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)
            os_ = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
            return od, l, cs, os_
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                       # scratch of the capture stream reaches its size
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            captured = step()
        q.copy_(randn_bf16(1, H, n, 128, seed=41).to(dev))   # new data, same buffers
        k.copy_(randn_bf16(1, H, n, 128, seed=42).to(dev))
        g.replay()
        torch.cuda.synchronize()
        eager = step()
        torch.cuda.synchronize()
        for a, b in zip(eager, captured):
            assert torch.equal(a, b)
    finally:
        for o in opts:
            _native.set_option(o, 0)


@pytest.mark.parametrize("n,nk,heads", [(1536, 1536, 4), (1000, 1000, 3), (2304, 2304, 2), (960, 2304, 2)])
def test_fused_mask_step_equals_colsum_then_topk_mask(dev, n, nk, heads):
    """chipmunk.dense_colsum_topk_mask (the mask-recompute step without the column-sum tensor: the mask kernel adds the three
    bf16 partial rows of a group itself) against dense_colsum_attn followed by topk_mask: o, l and the mask bit for bit --
    ragged last groups, fewer query rows than keys (query-group sharding), static mask and group flags; and both against the
    oracle's column sums."""
    import math
    from chipmunk_amd import _native
    from chipmunk_amd import ops
    G = math.ceil(n / 192)
    q = randn_bf16(1, heads, n, 128, seed=n + 1).to(dev)
    k = randn_bf16(1, heads, nk, 128, seed=nk + 2).to(dev)
    v = randn_bf16(1, heads, nk, 128, seed=nk + 3).to(dev)
    gen = torch.Generator().manual_seed(7)
    static = (torch.rand(1, heads, G, nk, generator=gen) < 0.02).to(dev)
    groups = (torch.rand(1, heads, G, 1, generator=gen) < 0.8).to(dev)
    ktop = 128
    for opt in ("attn_dense64", "attn_colsum64"):
        _native.set_option(opt, 1)
    try:
        _, l0 = ops.dense_attn(q, k, v)
        o_a, cs, l_a = ops.dense_colsum_attn(q, k, v, l0)
        cs = cs[..., :G, :nk]
        m_a = ops.topk_mask(cs, ktop, 0.0, groups, static)
        o_b, m_b, l_b = ops.dense_colsum_topk_mask(q, k, v, l0, ktop, 0.0, groups, static)
        torch.cuda.synchronize()
    finally:
        for opt in ("attn_dense64", "attn_colsum64"):
            _native.set_option(opt, 0)
    assert torch.equal(o_a, o_b) and torch.equal(l_a, l_b)
    assert m_b.shape == (1, heads, G, nk) and torch.equal(m_a, m_b)
    assert (m_b.sum(-1)[groups[..., 0]] >= ktop).all()
    if n == nk:   # (the oracle's cs has Nq columns)
        cs_ref = oracle.dense_colsum_attn(q.cpu(), k.cpu(), v.cpu(), l0[..., :n, :].cpu())[1]
        torch.testing.assert_close(cs.float().cpu(), cs_ref.float()[..., :G, :nk], rtol=3e-2, atol=2e-3)
