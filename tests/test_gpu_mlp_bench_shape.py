"""GEMM1 (csp_mlp_mm1 / csp_mlp_mm1_scatter / csp_mlp_mm1_fp8) at the FLUX bench's launch shape, where the PERSISTENT tile loop and the
TAIL SPLIT of `mm1_kernel` (csrc/mlp.hip: plan_tiles / tile_at) actually run: 34 groups x 32 live column tiles = 1088 tiles = 136 per XCD
on 64 resident slots per XCD -> every workgroup walks two whole tiles, then the XCD's last 8 tiles are handed out as 32 quarter-size
sub-tiles (64 x 64 outputs).  Reference behaviour: csrc/mlp/csp_mlp_mm1.cu:207-247 (tile walk, tiles past counts[g] skipped),
:354-390 (epilogue).  Every group is compared with fp32 torch math on the device; groups chosen to cover each tile class (first-round whole
tile, second / third persistent iteration, tail sub-tiles on several XCDs) are compared with the CPU oracle as well.

The host mirror of the kernel's plan below is pinned to the kernel itself: with `mm1_probe = 3` the kernel skips exactly its sub-tiles, so the
sentinel pattern it leaves must be the one the mirror predicts."""
import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16

pytestmark = pytest.mark.gpu

# The library ships ONE GEMM1 and ONE GEMM2 form.  The measured-and-not-shipped forms live in tools/probes/mm1_forms (a build of the
# same sources with -DCHIPMUNK_MM1_PROBES); tests/test_gpu_mlp_forms.py re-runs this file against that library with CHIPMUNK_MM1_FORMS=1.
import os as _os
_FORMS = _os.environ.get("CHIPMUNK_MM1_FORMS") == "1"
MM1_FORMS = [0, 20, 21] if _FORMS else [0]
MM1_FORMS_10 = [0, 10, 20, 21] if _FORMS else [0]
MM2_FORMS = [0, 10, 14] if _FORMS else [0]

BM, BN, NR, NSUB, WPS = 128, 128, 4, 4, 2     # shipped GEMM1 variant <128, 64, 2, 2>: NSUB = 2 * (BN / 64)
M, F = 4352, 12288                            # FLUX.1-dev 1280x768: 4096 image + 256 text tokens, mlp hidden 12288
G = M // BM


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def bench_like_counts(top=4096):
    """Module-like ragged counts: multiples of 256 around 0.3 * F, one group at 0; the groups the tail split lands on (32, 33 with
    NR = 4) keep enough columns for their sub-tiles to be live on most XCDs."""
    base = [3328, 3584, 3840, 4096]
    c = [min(top, base[(7 * g + 3) % 4]) for g in range(G)]
    c[5] = 0
    c[32], c[33] = top, top - 256
    return c


def plan(counts, cus, F_=F):
    """Host mirror of plan_tiles + tile_at + the persistent walk: a list of dicts, one per work item of the launch."""
    ntmax = (F_ + BN - 1) // BN
    ntl = min((max(counts) + BN - 1) // BN, ntmax)
    total = ntl * G
    slots_per_xcd = WPS * cus // 8
    tiles_per_xcd = (G * ntmax + 7) // 8
    stride = min(tiles_per_xcd, slots_per_xcd)
    q, r = total >> 3, total & 7
    items = []
    info = []
    for xcd in range(8):
        mine = q + (1 if xcd < r else 0)
        full = mine
        rem = mine % slots_per_xcd
        if mine > slots_per_xcd and rem > 0 and rem * NSUB <= slots_per_xcd:
            full = mine - rem
        base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
        info.append(dict(mine=mine, full=full, rem=rem, slots_per_xcd=slots_per_xcd))
        for slot in range(full + (mine - full) * NSUB):
            sub, s = -1, slot
            if slot >= full:
                k = slot - full
                sub, s = k % NSUB, full + k // NSUB
            t = base + s
            per = G * NR
            nb, rm = divmod(t, per)
            nr = min(NR, ntl - nb * NR)
            g = rm // nr
            nt = nb * NR + rm - g * nr
            n0 = nt * BN + (0 if sub < 0 else (sub >> 1) * 64)
            items.append(dict(xcd=xcd, slot=slot, it=slot // stride, g=g, nt=nt, sub=sub, n0=n0,
                              m_off=0 if sub < 0 else (sub & 1) * 64, live=n0 < counts[g]))
    return items, info


def index_rows(counts, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.stack([torch.randperm(F, device=dev, generator=g) for _ in range(G)]).to(torch.int32)


def covering_groups(items):
    """Groups that between them own: a live first-round whole tile, a second- and a third-or-later-iteration whole tile, and live
    sub-tiles on at least two XCDs.  Returns {class: [up to two groups]}, class = ("whole", iteration) or ("sub", xcd)."""
    want = {}
    for it in items:
        if not it["live"]:
            continue
        key = ("sub", it["xcd"]) if it["sub"] >= 0 else ("whole", min(it["it"], 2))
        gs = want.setdefault(key, [])
        if it["g"] not in gs and len(gs) < 2:
            gs.append(it["g"])
    return want


def oracle_groups(cover, limit=8):
    """Up to `limit` distinct groups, every tile class represented: {group: [classes]}."""
    by_group = {}
    for rank in (0, 1):
        for key, gs in sorted(cover.items()):
            if rank < len(gs) and (gs[rank] in by_group or len(by_group) < limit):
                by_group.setdefault(gs[rank], []).append(key)
    return by_group


def test_plan_mirror_matches_kernel_and_split_is_taken(dev):
    """The launch takes the split (mine > slots_per_xcd, rem * NSUB <= slots) and the kernel's sub-tiles are where the mirror says: with
    mm1_probe = 3 the kernel skips its sub-tiles and nothing else."""
    from chipmunk_amd import _native
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    counts = bench_like_counts()
    items, info = plan(counts, cus)
    for x in info:
        assert x["mine"] > x["slots_per_xcd"] and 0 < x["rem"] and x["rem"] * NSUB <= x["slots_per_xcd"], info
        assert x["full"] == x["mine"] - x["rem"]
    assert max(i["it"] for i in items if i["sub"] < 0) >= 1, "whole tiles beyond the first persistent iteration"
    assert len({i["xcd"] for i in items if i["sub"] >= 0 and i["live"]}) >= 2, "live sub-tiles on at least two XCDs"
    K = 256
    a, b = randn_bf16(M, K, seed=1, scale=0.5, device=dev), randn_bf16(F, K, seed=2, scale=0.1, device=dev)
    bias, cache = randn_bf16(F, seed=3, scale=0.2, device=dev), randn_bf16(F, M, seed=4, scale=0.3, device=dev)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    inds = index_rows(counts, 5, dev)
    c = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    _native.set_option("mm1_probe", 3)
    try:
        torch.ops.chipmunk.csp_mlp_mm1(a, b, c, bias, cache, inds, cnt)
        torch.cuda.synchronize()
    finally:
        _native.set_option("mm1_probe", 0)
    untouched = (c == 7.0)
    expect = torch.zeros(M, F, dtype=torch.bool, device=dev)
    for g in range(G):
        expect[g * BM:(g + 1) * BM, counts[g]:] = True
    for it in items:
        if it["sub"] >= 0 and it["live"]:
            r0 = it["g"] * BM + it["m_off"]
            expect[r0:r0 + 64, it["n0"]:min(it["n0"] + 64, counts[it["g"]])] = True
    assert torch.equal(untouched, expect), "the kernel's sub-tile map differs from the host mirror"


@pytest.mark.parametrize("variant", MM1_FORMS)   # 20 / 21: the producer / consumer forms (own tile walks: 128 x 256 tiles, 128 x 128 with the DMA
@pytest.mark.parametrize("top", [4096, F])   # stream across tiles).  4096: the bench's plan (2 iterations + split tail); F: one group keeps every column
def test_mm1_bench_shape_vs_torch_and_oracle(dev, top, variant, request):  # (96 live column tiles, 408 tiles per XCD = 7 persistent iterations, no split)
    from chipmunk_amd import _native
    _native.set_option("mm1_variant", variant)
    request.addfinalizer(lambda: _native.set_option("mm1_variant", 0))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    K = 3072
    counts = bench_like_counts(4096)
    if top == F:
        counts[17] = F
    items, info = plan(counts, cus)
    cover = covering_groups(items)
    if top == 4096:
        assert ("whole", 0) in cover and ("whole", 1) in cover and sum(k[0] == "sub" for k in cover) >= 2, cover
    else:
        assert ("whole", 2) in cover and not any(k[0] == "sub" for k in cover), cover
    a, b = randn_bf16(M, K, seed=1, scale=0.5, device=dev), randn_bf16(F, K, seed=2, scale=0.05, device=dev)
    bias, cache = randn_bf16(F, seed=3, scale=0.2, device=dev), randn_bf16(F, M, seed=4, scale=0.3, device=dev)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    inds = index_rows(counts, 5, dev)
    c = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    cache0 = cache.clone()
    torch.ops.chipmunk.csp_mlp_mm1(a, b, c, bias, cache, inds, cnt)
    assert torch.equal(cache, cache0), "csp_mlp_mm1 does not write the cache"
    # the fused-scatter form: same packed deltas bit for bit, cache == cache + delta in bf16 (scatter_add.cu:43-98)
    c2 = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    cache2 = cache0.clone()
    torch.ops.chipmunk.csp_mlp_mm1_scatter(a, b, c2, bias, cache2, inds, cnt)
    assert torch.equal(c2.view(torch.int16), c.view(torch.int16)), "mm1_scatter's deltas differ from mm1's"
    for g in range(G):                               # every group vs fp32 torch
        rows = slice(g * BM, (g + 1) * BM)
        n = counts[g]
        assert (c[rows, n:] == 7.0).all(), f"group {g}: columns past the count written"
        if n == 0:
            continue
        cols = inds[g, :n].long()
        act = torch.nn.functional.gelu(a[rows].float() @ b[cols].float().T + bias[cols].float(), approximate="tanh")
        want = act - cache0[cols][:, rows].float().T
        assert_close_bf16(c[rows, :n], want, what=f"mm1 group {g} vs fp32 torch")
        new = (cache0[cols][:, rows].float() + c[rows, :n].float().T).to(torch.bfloat16)
        assert torch.equal(cache2[cols][:, rows], new), f"group {g}: fused scatter-add != bf16(cache + delta)"
        rest = inds[g, n:].long()
        assert torch.equal(cache2[rest][:, rows], cache0[rest][:, rows]), f"group {g}: unselected cache columns changed"
    # the covering groups vs the CPU oracle (csp_mlp_mm1.cu restated in C)
    ac, bc, biasc, cachec, indc = a.cpu(), b.cpu(), bias.cpu(), cache0.cpu(), inds.cpu()
    picked = oracle_groups(cover)
    assert {k for ks in picked.values() for k in ks} >= set(cover), "a tile class has no oracle-checked group"
    assert len(picked) >= (6 if top == 4096 else 2), picked
    for g, key in sorted(picked.items()):
        rows = slice(g * BM, (g + 1) * BM)
        c_ref = torch.full((BM, F), 7.0, dtype=torch.bfloat16)
        oracle.csp_mlp_mm1(ac[rows].contiguous(), bc, c_ref, biasc, cachec[:, rows].contiguous(), indc[g:g + 1].contiguous(),
                           torch.tensor(counts[g:g + 1], dtype=torch.int32))
        assert_close_bf16(c[rows], c_ref, what=f"mm1 group {g} (covers {key}) vs oracle")


@pytest.mark.parametrize("variant", MM1_FORMS)
def test_mm1_fp8_split_plan_vs_torch(dev, variant, request):
    from chipmunk_amd import _native
    _native.set_option("mm1_variant", variant)
    request.addfinalizer(lambda: _native.set_option("mm1_variant", 0))
    _fp8_split_plan_vs_torch(dev)


def _fp8_split_plan_vs_torch(dev):
    """The fp8 template on a launch whose plan takes the split (same 34 x 32 live tiles; Wan2.1's K and F): every group vs fp32 torch on
    the dequantised operands, as tests/test_gpu_fullsize.py::test_c5_wan_fp8_gemm1_full_size does for the (split-free) Wan launch."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    K, F8 = 1536, 8960
    counts = bench_like_counts(4096)
    items, info = plan(counts, cus, F8)
    assert all(x["full"] < x["mine"] for x in info) and max(i["it"] for i in items if i["sub"] < 0) >= 1
    g_ = torch.Generator(device=dev).manual_seed(51)
    x = torch.randn(M, K, device=dev, generator=g_)
    w = torch.randn(F8, K, device=dev, generator=g_) * 0.05
    sa, sb = 448.0 / x.abs().max(), 448.0 / w.abs().max()
    a8, b8 = (x * sa).to(torch.float8_e4m3fn), (w * sb).to(torch.float8_e4m3fn)
    bias = (torch.randn(F8, device=dev, generator=g_) * 0.2).to(torch.bfloat16)
    cache0 = (torch.randn(F8, M, device=dev, generator=g_) * 0.3).to(torch.bfloat16)
    inds = torch.stack([torch.randperm(F8, device=dev, generator=g_) for _ in range(G)]).to(torch.int32)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    ra, rb = (1.0 / sa).reshape(1).float(), (1.0 / sb).reshape(1).float()
    for update in (False, True):
        cache = cache0.clone()
        packed = torch.full((M, F8), 7.0, dtype=torch.bfloat16, device=dev)
        torch.ops.chipmunk.csp_mlp_mm1_fp8(a8, b8, packed, bias, cache, inds, cnt, ra, rb, update)
        for g in range(G):
            rows = slice(g * BM, (g + 1) * BM)
            n = counts[g]
            assert (packed[rows, n:] == 7.0).all()
            if n == 0:
                continue
            cols = inds[g, :n].long()
            acc = (a8[rows].float() @ b8[cols].float().T) * ra * rb + bias[cols].float()
            act = torch.nn.functional.gelu(acc, approximate="tanh").to(torch.bfloat16)
            want = (act.float() - cache0[cols][:, rows].float().T).to(torch.bfloat16)
            assert_close_bf16(packed[rows, :n], want, atol=3e-2, rtol=3e-2, what=f"fp8 GEMM1 group {g}")
            if update:
                assert_close_bf16(cache[cols][:, rows], act.T, atol=3e-2, rtol=3e-2, what=f"fp8 cache update group {g}")
        if not update:
            assert torch.equal(cache, cache0)


@pytest.mark.parametrize("fp8", [False, True])
def test_producer_consumer_forms_equal_the_shipped_kernel_bit_for_bit_and_run_to_run(dev, fp8):
    """Race screen for the two producer / consumer forms (their synchronisation is hand-counted vmcnt / lgkmcnt across raw barriers): at the
    bench's launch shape, ten launches each give the SAME bits -- and the bits of the shipped kernel, whose k order, accumulator seeding
    and epilogue arithmetic they share.  Packed deltas and the scattered cache are compared."""
    from chipmunk_amd import _native
    K = 1536 if fp8 else 3072
    Fw = 8960 if fp8 else F
    counts = bench_like_counts(4096)
    g_ = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(M, K, device=dev, generator=g_)
    w = torch.randn(Fw, K, device=dev, generator=g_) * 0.05
    bias = (torch.randn(Fw, device=dev, generator=g_) * 0.2).to(torch.bfloat16)
    cache0 = (torch.randn(Fw, M, device=dev, generator=g_) * 0.3).to(torch.bfloat16)
    inds = torch.stack([torch.randperm(Fw, device=dev, generator=g_) for _ in range(G)]).to(torch.int32)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    if fp8:
        sa, sb = 448.0 / x.abs().max(), 448.0 / w.abs().max()
        a, b = (x * sa).to(torch.float8_e4m3fn), (w * sb).to(torch.float8_e4m3fn)
        ra, rb = (1.0 / sa).reshape(1).float(), (1.0 / sb).reshape(1).float()
    else:
        a, b = x.to(torch.bfloat16), w.to(torch.bfloat16)

    def run(variant):
        _native.set_option("mm1_variant", variant)
        try:
            c = torch.full((M, Fw), 7.0, dtype=torch.bfloat16, device=dev)
            cache = cache0.clone()
            if fp8:
                torch.ops.chipmunk.csp_mlp_mm1_fp8_scatter(a, b, c, bias, cache, inds, cnt, ra, rb)
            else:
                torch.ops.chipmunk.csp_mlp_mm1_scatter(a, b, c, bias, cache, inds, cnt)
            torch.cuda.synchronize()
            return c, cache
        finally:
            _native.set_option("mm1_variant", 0)

    c0, cache_ref = run(0)
    for variant in MM1_FORMS[1:]:
        for rep in range(10):
            c, cache = run(variant)
            assert torch.equal(c.view(torch.int16), c0.view(torch.int16)), f"variant {variant}, launch {rep}: packed deltas differ from the shipped kernel's"
            assert torch.equal(cache.view(torch.int16), cache_ref.view(torch.int16)), f"variant {variant}, launch {rep}: cache differs"
