"""Parity on the launches bench.py actually makes (round-2 verdict, weak 1d / next 8): HunyuanVideo C3 size, ALL 24 heads, the
ragged key counts that the module's own mask pipeline produces (dense_colsum_attn -> top-k + 1 % random + static text columns ->
bit-packed mask -> sorted indices), HIP `csp_128_attn` / `csp_attn_out` against the C oracle on sampled (head, group) items --
text groups with all 119 056 keys included -- and the same launch with the running-maximum loop forced."""
import math

import pytest
import torch

import oracle
from helpers import assert_close_bf16

pytestmark = pytest.mark.gpu

N_IMG, N_TXT, H = 33 * 45 * 80, 256, 24
N = N_IMG + N_TXT


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def bench_launch(dev):
    """q, k, v, the module-generated (indices, counts) and the dense output of one C3 layer, exactly as bench.py builds them."""
    import os
    from chipmunk_amd.util import config as cfg
    from chipmunk_amd.util import layer_counter as lc
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util.layer_counter import LayerCounter
    import chipmunk_amd.ops as ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)
    cfg.load_from_file(os.path.join(root, "configs", "hunyuan_c3.yml"))
    cfg.GLOBAL_CONFIG["step_caching"]["is_enabled"] = False
    cfg.GLOBAL_CONFIG["attn"]["first_n_dense_layers"] = 0
    g = torch.Generator(device=dev).manual_seed(1234)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    layer_num, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
    attn = SparseDiffAttn(layer_num, counter)
    attn.initialize_static_mask((33, 45, 80), N_TXT, H, dev)
    with torch.no_grad():
        attn(q, k, v)                      # step 0: dense, remembers l
        o_dense = attn(q, k, v)            # step 1: column sums -> mask -> cache = dense - sparse
        inds, counts = ops.mask_to_sorted_indices(attn.storage.get_indices(), attn.mask_shape[0], 128, 192)
        cache = attn.storage.get_out_cache()
    torch.cuda.synchronize()
    yield {"q": q, "k": k, "v": v, "inds": inds, "counts": counts, "o_dense": o_dense, "cache": cache, "attn": attn}
    cfg.reset_to_base()
    lc.singleton.__init__(0, 0)


def _sample_items(counts, n_random=28):
    """>= 32 (head, group) items: the longest (text / tail groups keep all keys), the shortest, the ragged last group, random ones."""
    G = counts.shape[-1]
    c = counts[0].cpu()
    gen = torch.Generator().manual_seed(0)
    items = {(int(h), int(gi)) for h, gi in zip(torch.randint(0, H, (n_random,), generator=gen), torch.randint(0, G, (n_random,), generator=gen))}
    flat = c.flatten()
    for idx in (int(flat.argmax()), int(flat.argmin())):
        items.add((idx // G, idx % G))
    items |= {(0, G - 1), (H - 1, G - 1), (5, G - 2), (11, 0)}
    return sorted(items)


def _oracle_item(launch, h, gi):
    """Oracle csp_128_attn of one (head, group): that group's query rows against the head's K, V with the group's index row."""
    rows = slice(gi * 192, min((gi + 1) * 192, N))
    qg = launch["q"][:, h:h + 1, rows].cpu()
    kh, vh = launch["k"][:, h:h + 1].cpu(), launch["v"][:, h:h + 1].cpu()
    ind = launch["inds"][:, h:h + 1, gi:gi + 1].cpu().contiguous()
    cnt = launch["counts"][:, h:h + 1, gi:gi + 1].cpu().contiguous()
    return oracle.csp_128_attn(qg, kh, vh, ind, cnt), rows


def test_bench_mask_is_ragged_and_has_full_text_groups(bench_launch):
    c = bench_launch["counts"][0].float()
    assert c.max().item() >= N - 192 and c.min().item() < 0.2 * N, "the bench's launch mixes 119 k-key text / tail groups with ~9 k-key ones"
    assert 0.90 < 1.0 - c.mean().item() / N < 0.95      # the configuration BASELINE.json calls "93 % sparsity"


@pytest.mark.parametrize("form", ["csp_128_attn", "csp_attn_out"])
def test_c3_all_heads_bench_launch_vs_oracle(dev, bench_launch, form):
    """the very launch of the bench's timed region (24 heads, ragged counts, attn96.hip + plan slicing)"""
    L = bench_launch
    if form == "csp_128_attn":
        o = torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"])
    else:
        o = torch.ops.chipmunk.csp_attn_out(L["q"], L["k"], L["v"], L["cache"], L["inds"], L["counts"], 1)
    torch.cuda.synchronize()
    items = _sample_items(L["counts"])
    assert len(items) >= 32
    for h, gi in items:
        ref, rows = _oracle_item(L, h, gi)
        if form == "csp_128_attn":
            assert_close_bf16(o[:, h:h + 1, rows], ref, what=f"bench launch, head {h} group {gi} ({int(L['counts'][0, h, gi])} keys)")
        else:
            # sparse step = cache + sparse = dense (the cache was built as dense - sparse from the same q, k, v): bf16 roundings only
            want = L["cache"][:, h:h + 1, rows].float().cpu() + ref.float()
            assert_close_bf16(o[:, h:h + 1, rows], want, atol=3e-2, what=f"bench launch (cache + delta), head {h} group {gi}")
    assert_close_bf16(torch.ops.chipmunk.csp_attn_out(L["q"], L["k"], L["v"], L["cache"], L["inds"], L["counts"], 1)[:, :2],
                      L["o_dense"][:, :2].float().cpu(), atol=6e-2, what="sparse step reproduces the dense step it was cached from")


@pytest.mark.parametrize("how", ["option", "q_times_4"])
def test_c3_running_maximum_loop_vs_oracle(dev, bench_launch, how):
    """the data-dependent slow path at full size: the gathered kernel's running-maximum loop, forced by the option (same data)
    and by inputs whose |q| max|k| bound is too large for the loop without a reference point (q x 4, as bench.py's
    running_max_fallback_leg)"""
    from chipmunk_amd import _native
    L = bench_launch
    hs = slice(3, 4)                        # one head at C3 size (a head-parallel rank's launch), all its 621 groups
    q = L["q"][:, hs].contiguous()
    if how == "q_times_4":
        q = (q.float() * 4).to(torch.bfloat16)
    k, v = L["k"][:, hs].contiguous(), L["v"][:, hs].contiguous()
    inds, counts = L["inds"][:, hs].contiguous(), L["counts"][:, hs].contiguous()
    _native.set_option("attn_csp96", 1)
    _native.set_option("attn_nomax", 2 if how == "option" else 0)
    try:
        o = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
        torch.cuda.synchronize()
    finally:
        _native.set_option("attn_csp96", 0)
        _native.set_option("attn_nomax", 0)
    G = counts.shape[-1]
    kc, vc = k.cpu(), v.cpu()
    for gi in (0, 7, 200, 333, G - 2, G - 1, int(counts[0, 0].argmax()), int(counts[0, 0].argmin())):
        rows = slice(gi * 192, min((gi + 1) * 192, N))
        ref = oracle.csp_128_attn(q[:, :, rows].cpu(), kc, vc, inds[:, :, gi:gi + 1].cpu().contiguous(), counts[:, :, gi:gi + 1].cpu().contiguous())
        assert_close_bf16(o[:, :, rows], ref, what=f"running-maximum loop ({how}), group {gi}")


def test_bench_launch_is_run_to_run_identical(dev, bench_launch):
    """32 launches of the bench's 24-head gathered launch, both output forms and the running-maximum loop: bit-identical every
    time.  (No float atomics, slices merged in slice order -- and, since round 3, no register copied while its LDS read is in
    flight: one launch in ~15 used to differ in one 32x32 block of one item; tools/probes/race96.py, tools/audit_async_lds.py.)"""
    from chipmunk_amd import _native
    L = bench_launch
    ref = torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"])
    ref_out = torch.ops.chipmunk.csp_attn_out(L["q"], L["k"], L["v"], L["cache"], L["inds"], L["counts"], 1)
    for i in range(32):
        assert torch.equal(torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"]), ref), f"launch {i}"
    for i in range(8):
        assert torch.equal(torch.ops.chipmunk.csp_attn_out(L["q"], L["k"], L["v"], L["cache"], L["inds"], L["counts"], 1), ref_out), f"out form, launch {i}"
    _native.set_option("attn_nomax", 2)
    try:
        ref_rm = torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"])
        for i in range(8):
            assert torch.equal(torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"]), ref_rm), f"running maximum, launch {i}"
    finally:
        _native.set_option("attn_nomax", 0)


def test_bench_launch_two_kernels_agree_everywhere(dev, bench_launch):
    """All 24 x 621 items of the bench's launch through attn96.hip (the shipped selection) and through the general kernel of
    attn.hip (option attn_csp96 = 2): two independent implementations -- different tiling, MFMA shape, softmax form (exponent of
    the folded score vs lagging running maximum), slice plans -- must agree to bf16 rounding on EVERY row.  With outputs that are
    averages over ~9 000 keys (|o| ~ 0.03) that is a 1.5e-3 absolute test: the oracle samples 34 items, this covers the rest
    (a one-tile error in one 32x32 block is ~5e-3)."""
    from chipmunk_amd import _native
    L = bench_launch
    a = torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"])
    _native.set_option("attn_csp96", 2)
    try:
        b = torch.ops.chipmunk.csp_128_attn(L["q"], L["k"], L["v"], L["inds"], L["counts"])
    finally:
        _native.set_option("attn_csp96", 0)
    d = (a.float() - b.float()).abs()
    tol = 1.5e-3 + 1e-2 * b.float().abs()
    bad = d > tol
    assert not bad.any(), f"{int(bad.sum())} elements differ, worst {float(d.max()):.4g} at {[int(x) for x in (d == d.max()).nonzero()[0]]}"


def test_bench_launch_ragged_rows_and_token_major_cache(dev, bench_launch):
    """The shipped sparse step at the bench's size: `csp_attn_out_ragged` over the kept ragged index rows (0.56 GB instead of the 7 GB
    padded tensor) with a token-major cache, against `csp_attn_out` over the padded rows with the contiguous cache -- bit for bit,
    all 24 x 621 items (text groups with 119 056 keys and their key slices included)."""
    import chipmunk_amd.ops as ops
    L = bench_launch
    g = torch.Generator(device=dev).manual_seed(5)
    cache = torch.randn(L["q"].shape, device=dev, dtype=torch.bfloat16, generator=g)
    ref = ops.csp_attn_out(L["q"], L["k"], L["v"], cache, L["inds"], L["counts"], 1)
    flat, offsets = ops.compact_indices(L["inds"], L["counts"])
    assert flat.numel() * 4 < 0.1 * L["inds"].numel() * 4, "the kept keys are a small part of the padded index tensor"
    B, H, N, D = cache.shape
    cache_tm = torch.empty(B, N, H, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    cache_tm.copy_(cache)
    got = ops.csp_attn_out_ragged(L["q"], L["k"], L["v"], cache_tm, flat, offsets, L["counts"], 1)
    assert got.stride() == cache_tm.stride() and got.permute(0, 2, 1, 3).is_contiguous()
    assert torch.equal(got, ref)
