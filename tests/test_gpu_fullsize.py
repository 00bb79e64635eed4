"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle is too slow there), plus the
domain's edge cases: empty / ragged groups, duplicate (colliding) indices, all-true / all-false mask rows."""
import math

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


def test_c3_sparse_attention_with_all_keys_equals_dense(dev):
    """HunyuanVideo C3 sequence (119 056 tokens), 1 head: identity index lists (every key kept) must reproduce the
    dense kernel, and the dense kernel must agree with SDPA on a slice of query rows."""
    N, H = 119056, 1
    g = torch.Generator(device=dev).manual_seed(1)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    G = math.ceil(N / 192)
    o_dense, l = torch.ops.chipmunk.dense_attn(q, k, v)
    # identity indices for a subset of groups only (the full index tensor would be 119056*621*4 B = 296 MB: fine)
    inds = torch.arange(N, dtype=torch.int32, device=dev).expand(1, H, G, N).contiguous()
    counts = torch.full((1, H, G), N, dtype=torch.int32, device=dev)
    o_sparse = torch.zeros_like(q)
    torch.ops.chipmunk.csp_attn(q, k, v, o_sparse, inds, counts, 1)
    assert_close_bf16(o_sparse, o_dense, atol=1e-2, rtol=1e-2, what="C3 identity sparse vs dense")
    rows = slice(50000, 50384)
    ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, rows].float(), k.float(), v.float())
    assert_close_bf16(o_dense[:, :, rows], ref, atol=1e-2, rtol=2e-2, what="C3 dense vs SDPA slice")
    lref = 1.0 / torch.exp((q[:, :, rows].float() @ k.float().transpose(-1, -2)) / math.sqrt(128)).sum(-1, keepdim=True)
    torch.testing.assert_close(l[:, :, rows], lref, rtol=2e-3, atol=0)


def test_c3_fused_column_sums_are_a_partition_of_unity(dev):
    """HunyuanVideo C3 sequence, 2 heads, the one-pass dense_colsum_attn (the route the bench takes).  With p = this very
    call's own l, the summand exp(s_ij) p_i is the softmax probability: every query row contributes exactly 1 to the sum of
    its group's column sums, so sum_j cs[g, j] = rows of group g (192; 16 in the last).  Size-independent, exercises all
    1 861 key tiles x 1 861 wave rows x the combine; plus: same o / l as dense_attn, cs equal to the two-pass route, and
    one group checked against an fp32 reference."""
    from chipmunk_amd import _native
    N, H = 119056, 2
    g = torch.Generator(device=dev).manual_seed(7)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    G = math.ceil(N / 192)
    o_d, l = torch.ops.chipmunk.dense_attn(q, k, v)
    o, cs, l2 = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)
    assert cs.shape == (1, H, G, N)
    assert_close_bf16(o, o_d.float().cpu(), atol=4e-3, rtol=1e-2, what="C3 one-pass o vs dense_attn")
    torch.testing.assert_close(l2, l, rtol=2e-3, atol=0)
    tot = cs.float().sum(-1)
    rows = torch.full((G,), 192.0, device=dev)
    rows[-1] = N - 192 * (G - 1)
    torch.testing.assert_close(tot, rows.expand(1, H, G), rtol=5e-3, atol=0)
    # both two-pass routes (dense + colsum64_kernel, dense + the general kernel's K-only pass) agree with it everywhere: this
    # comparison is how the general pass's missing lgkmcnt(0) before its barrier was found (one wave's partial sums of an
    # odd tile read four tiles stale, about once per launch at this size)
    for which in (1, 2):
        _native.set_option("attn_fused_colsum", 2)
        _native.set_option("attn_colsum64", which)
        try:
            for _ in range(2):
                two = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1]
                assert_close_bf16(cs, two.float().cpu(), atol=1e-5, rtol=2e-2, what=f"C3 column sums, one pass vs two (K-only kernel {which})")
        finally:
            _native.set_option("attn_fused_colsum", 0)
            _native.set_option("attn_colsum64", 0)
    again = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1]
    assert torch.equal(cs, again), "one pass: run-to-run identical"
    for hh, gi in ((1, 300), (0, 0), (0, G - 2), (1, 77), (0, 411), (1, 555)):
        qs = q[0, hh, gi * 192:(gi + 1) * 192].float()
        p_ref = torch.exp(qs @ k[0, hh].float().T / math.sqrt(128)) * l[0, hh, gi * 192:(gi + 1) * 192]
        assert_close_bf16(cs[0, hh, gi], p_ref.sum(0).cpu(), atol=1e-5, rtol=2e-2, what=f"C3 column sums of group {gi} vs fp32")


def test_c3_cache_plus_delta_identity(dev):
    """The identity the method rests on, at C3 size: o_cache = dense - sparse; o_cache + sparse == dense."""
    N, H, count = 119056, 1, 7296
    g = torch.Generator(device=dev).manual_seed(2)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    G = math.ceil(N / 192)
    inds = torch.zeros(1, H, G, G * 192, dtype=torch.int32, device=dev)
    for g0 in range(0, G, 64):
        r = torch.rand(min(64, G - g0), N, device=dev, generator=g)
        inds[0, 0, g0:g0 + r.shape[0], :count] = r.topk(count, dim=-1).indices.to(torch.int32)
    counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
    import chipmunk_amd
    o, _ = chipmunk_amd.ops.dense_attn(q, k, v)
    sparse = chipmunk_amd.ops.csp_attn(q, k, v, inds, counts)
    assert sparse.shape == q.shape
    back = (o - sparse) + sparse
    assert_close_bf16(back, o, atol=2e-2, rtol=2e-2, what="C3 cache + delta")
    # the sparse result is a convex combination of the selected V rows: bounded by their extrema per group
    sel = v[0, 0, inds[0, 0, 5, :count].long()].float()
    grp = sparse[0, 0, 5 * 192:6 * 192].float()
    assert (grp <= sel.max(0).values + 2e-2).all() and (grp >= sel.min(0).values - 2e-2).all()


def test_c3_mask_to_indices_properties_and_fused_path(dev):
    """C3 mask shape (2 of 24 heads): counts = popcount rounded up to 128; per row the index list is the reference's
    class-interleaved order (strictly increasing inside each residue class mod 32, classes ascending); the fused
    packed-bits path returns identical tensors."""
    import chipmunk_amd
    H, G, N = 2, 621, 119232
    g = torch.Generator(device=dev).manual_seed(3)
    mask = torch.rand(1, H, G, N, device=dev, generator=g) < 0.06
    inds, counts = torch.ops.chipmunk.mask_to_indices(mask, 128, 192)
    pop = mask.sum(-1).to(torch.int32)
    assert torch.equal(counts, ((pop + 127) // 128) * 128)
    row_i = inds[0, 1, 300, :pop[0, 1, 300]].cpu()
    cls = row_i % 32
    assert (cls[1:] >= cls[:-1]).all()
    same = cls[1:] == cls[:-1]
    assert (row_i[1:][same] > row_i[:-1][same]).all()
    assert torch.equal(torch.sort(row_i).values, torch.nonzero(mask[0, 1, 300]).flatten().cpu().to(torch.int32))
    pad = inds[0, 1, 300, pop[0, 1, 300]:counts[0, 1, 300]].cpu()
    assert torch.equal(pad, torch.nonzero(~mask[0, 1, 300]).flatten()[:pad.numel()].cpu().to(torch.int32))
    packed, shp = chipmunk_amd.ops.bitpack(mask)
    i2, c2 = chipmunk_amd.ops.packed_mask_to_indices(packed, shp, 128, 192)
    assert torch.equal(c2, counts)
    live = torch.arange(inds.shape[-1], device=dev)[None, None, None, :] < counts[..., None]
    assert torch.equal(i2[live], inds[live])


def test_c2_mlp_full_size_linearity(dev):
    """FLUX C2 MLP shape (M 4352, K 3072, F 12288): GEMM2 is linear in the packed activations and in-place additive:
    mm2(2*p) - C0 == 2 * (mm2(p) - C0) up to bf16 rounding; scatter-add twice == adding 2*p once."""
    M, K, F, keep = 4352, 3072, 12288, 4096
    g = torch.Generator(device=dev).manual_seed(4)
    packed = (torch.randn(M, F, device=dev, generator=g) * 0.125).to(torch.bfloat16)
    w2t = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    inds = torch.stack([torch.randperm(F, device=dev, generator=g) for _ in range(M // 128)]).to(torch.int32)
    counts = torch.full((M // 128,), keep, dtype=torch.int32, device=dev)
    c1, c2 = torch.zeros(M, K, device=dev, dtype=torch.bfloat16), torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
    torch.ops.chipmunk.csp_mlp_mm2(packed, w2t, inds, counts, c1)
    torch.ops.chipmunk.csp_mlp_mm2(packed * 2, w2t, inds, counts, c2)
    assert_close_bf16(c2, c1.float() * 2, atol=2e-2, rtol=2e-2, what="mm2 linearity")
    rows = slice(128 * 7, 128 * 8)
    ref = packed[rows, :keep].float() @ w2t[inds[7, :keep].long()].float()
    assert_close_bf16(c1[rows], ref, atol=3e-2, rtol=2e-2, what="mm2 group 7 vs torch")
    cache = torch.zeros(F, M, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        torch.ops.chipmunk.csp_scatter_add(packed[None], cache[None], inds[None], counts[None], 6)
    once = torch.zeros(F, M, device=dev, dtype=torch.bfloat16)
    torch.ops.chipmunk.csp_scatter_add((packed * 2)[None], once[None], inds[None], counts[None], 6)
    assert torch.equal(cache, once)          # x + x == 2x exactly in bf16
    assert cache[inds[3, keep:].long(), 3 * 128:4 * 128].abs().sum() == 0   # unselected columns untouched


# ------------------------------------------------------------------------------------------------ edge cases
def test_empty_and_ragged_groups(dev):
    n, H = 576, 2
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (1, 2, 3)]
    inds = torch.stack([torch.randperm(n, generator=torch.Generator().manual_seed(i)) for i in range(H * 3)])
    inds = inds.view(1, H, 3, n).to(torch.int32)
    counts = torch.tensor([[[0, 16, 576], [48, 0, 32]]], dtype=torch.int32)   # empty groups, tiny, everything
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what="ragged counts")
    assert (o[0, 0, :192] == 0).all() and (o[0, 1, 192:384] == 0).all()       # empty group -> zeros (documented)
    o0 = randn_bf16(1, H, n, 128, seed=9)
    oi = o0.clone().to(dev)
    torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), oi, inds.to(dev), counts.to(dev), -1)
    assert torch.equal(oi[0, 0, :192].cpu(), o0[0, 0, :192])                   # in place: empty group leaves o alone


def test_duplicate_indices_are_not_deduplicated(dev):
    """Collisions: the reference gathers whatever the list says: a key listed twice counts twice."""
    n, H, count = 384, 1, 64
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (4, 5, 6)]
    base = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:32]
    inds = torch.zeros(1, H, 2, n, dtype=torch.int32)
    inds[0, 0, :, :count] = torch.cat([base, base]).to(torch.int32)          # every key twice
    counts = torch.full((1, H, 2), count, dtype=torch.int32)
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what="duplicates vs oracle")
    inds1 = inds.clone(); inds1[0, 0, :, 32:] = 0
    o_single = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds1.to(dev),
                                              torch.full((1, H, 2), 32, dtype=torch.int32, device=dev))
    assert_close_bf16(o, o_single, what="softmax is invariant to duplicating every key")


def test_mask_rows_all_true_all_false_and_tiny(dev):
    n = 200   # not a multiple of 32, 64 or 192
    mask = torch.zeros(1, 1, 3, n, dtype=torch.bool)
    mask[0, 0, 0] = True
    mask[0, 0, 2, [0, 199]] = True
    ref_i, ref_c = oracle.mask_to_indices(mask, 128, 192)
    i, c = torch.ops.chipmunk.mask_to_indices(mask.to(dev), 128, 192)
    assert torch.equal(c.cpu(), ref_c) and c.cpu().tolist() == [[[256, 0, 128]]]
    assert torch.equal(i[0, 0, 0, :200].cpu(), ref_i[0, 0, 0, :200])           # all True: 200 written, count 256
    assert torch.equal(i[0, 0, 2, :128].cpu(), ref_i[0, 0, 2, :128])


@pytest.mark.gpu
def test_key_split_tail_matches_unsplit_at_scale():
    """The key-split tail of the attention launches (partials through library scratch, last-arriver merge) against the
    same launch with the split disabled, at a size with a real tail: 3 heads x 180 query groups = 540 workgroups on
    512 slots -> 28 tail items x 8 slices of ~135 key tiles each.  Inputs are different on every repetition so a stale
    partial from an earlier launch cannot hide."""
    from chipmunk_amd import _native
    dev = torch.device("cuda:0")
    H, N = 3, 192 * 180
    for rep in range(3):
        g = torch.Generator(device=dev).manual_seed(100 + rep)
        q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
        o_split, l_split = torch.ops.chipmunk.dense_attn(q, k, v)
        _native.set_option("attn_no_split", 1)
        try:
            o_ref, l_ref = torch.ops.chipmunk.dense_attn(q, k, v)
        finally:
            _native.set_option("attn_no_split", 0)
        # same tiles, same per-tile arithmetic; only the fp32 merge order of 8 slices differs
        torch.testing.assert_close(l_split, l_ref, rtol=1e-5, atol=0)
        d = (o_split.float() - o_ref.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * o_ref.float().abs().max().item(), d.max().item()  # one bf16 ulp
        # the 28 split items round p = exp2(s - m_run) to bf16 against a different running max than the unsplit
        # launch does, so about half of THEIR elements move by one last bit; everything else is untouched
        assert (d > 0).float().mean().item() < 0.06
        assert d.mean().item() < 2.0 ** -11 * o_ref.float().abs().mean().item()


@pytest.mark.gpu
@pytest.mark.parametrize("H", [1, 3])
def test_sliced_heavy_items_match_unsliced_and_oracle(H):
    """Ragged key counts with few heads (the head-parallel C4 case: 3 heads per rank): items far above a slot's share are
    cut into slices over their key tiles by the device-built work plan (attn_plan_kernel) and merged by the last arriver.
    Compared with the same launch without the plan (option attn_no_order) and, on the heavy groups, with the oracle; the
    in-place / out-of-place accumulate forms go through the same plan."""
    from chipmunk_amd import _native
    dev = torch.device("cuda:0")
    N = 33000                                   # >= 32768 keys: the plan is on; 172 query groups, ragged last one
    G = (N + 191) // 192
    g = torch.Generator().manual_seed(5 + H)
    q, k, v = [randn_bf16(1, H, N, 128, seed=40 + s + H) for s in range(3)]
    inds = torch.empty(1, H, G, G * 192, dtype=torch.int32)
    counts = torch.full((1, H, G), 512, dtype=torch.int32)
    for h in range(H):
        for gi in range(G):
            inds[0, h, gi, :N] = torch.randperm(N, generator=g).to(torch.int32)
            inds[0, h, gi, N:] = 0
    counts[0, :, 5] = N                          # a text-like group that keeps every key
    counts[0, :, G - 1] = 20000                  # the ragged tail group keeps most
    counts[0, 0, 7] = 0                          # and an empty one
    counts[0, H - 1, 9] = 33                     # not a multiple of the tile
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    # (a) the general kernel (attn.hip) with and without the plan: unsliced items must not change by a bit (the plan's tail cut --
    #     the last `blocks mod slots` items as slices, so that the final round is full -- is switched off for this comparison)
    _native.set_option("attn_csp96", 2)
    _native.set_option("attn_no_tail", 1)
    try:
        for rep in range(2):
            o_gen = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
            _native.set_option("attn_no_order", 1)
            try:
                o_plain = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
            finally:
                _native.set_option("attn_no_order", 0)
            d = (o_gen.float() - o_plain.float()).abs()
            assert d.max().item() <= 2.0 ** -7 * max(1e-3, o_plain.float().abs().max().item()), d.max().item()
            light = torch.ones(G, dtype=torch.bool)
            light[[5, G - 1]] = False
            rows = light.repeat_interleave(192)[:N].to(dev)
            assert torch.equal(o_gen[:, :, rows], o_plain[:, :, rows]), "unsliced items are computed exactly as before"
        # (a') with the tail cut: the same values to bf16 rounding (a cut item is a merge of fp32 partials), run to run identical
        _native.set_option("attn_no_tail", 0)
        o_tail = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
        assert torch.equal(o_tail, torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd))
        d = (o_tail.float() - o_plain.float()).abs()
        assert d.max().item() <= 2.0 ** -7 * max(1e-3, o_plain.float().abs().max().item()), d.max().item()
    finally:
        _native.set_option("attn_csp96", 0)
        _native.set_option("attn_no_tail", 0)
    # (b) the shipped selection for this launch (attn96.hip over the same plan): same results to bf16 precision, run to run identical
    o = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
    assert torch.equal(o, torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd))
    assert_close_bf16(o, o_gen.float().cpu(), atol=1e-2, rtol=1e-2, what="attn96 vs the general kernel over the same plan")
    # heavy groups against the oracle (only those: the oracle is a scalar restatement)
    for gi in (5, G - 1, 7, 9):
        r0, r1 = gi * 192, min(N, gi * 192 + 192)
        o_ref = oracle.csp_128_attn(q[:, :, r0:r1].contiguous(), k, v, inds[:, :, gi:gi + 1, :N].contiguous(), counts[:, :, gi:gi + 1].contiguous())
        assert_close_bf16(o[:, :, r0:r1], o_ref, what=f"group {gi} vs oracle")
    # accumulate forms
    base = randn_bf16(1, H, N, 128, seed=77)
    out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base.to(dev), indd, cntd, -1)
    want = (base.float() - o.float().cpu().to(torch.bfloat16).float()).to(torch.bfloat16)
    assert_close_bf16(out, want, atol=1e-2, rtol=1e-2, what="csp_attn_out with sliced items")
    inpl = base.clone().to(dev)
    torch.ops.chipmunk.csp_attn(qd, kd, vd, inpl, indd, cntd, -1)
    assert torch.equal(inpl, out), "in-place and out-of-place accumulate agree bit for bit"


# ------------------------------------------------------------------------------------------------ config-sized cases
def test_c2_flux_attention_full_size_all_heads(dev):
    """BASELINE configs[1] at its real size: 24 heads, 4352 tokens, 672 kept keys per 192-query group, in-place delta
    kernel.  Heads 0, 11 and 23 against the oracle; every head through the cache identity (o - sparse) + sparse == o up to
    the two bf16 roundings; and `csp_attn_out` == clone + in-place, bit for bit."""
    from helpers import random_index_sets
    H, N, count = 24, 4352, 672
    G = math.ceil(N / 192)
    q, k, v = [randn_bf16(1, H, N, 128, seed=s) for s in (31, 32, 33)]
    inds, counts = random_index_sets(1, H, G, N, count, N, seed=34, multiple_of=112)
    base = randn_bf16(1, H, N, 128, seed=35)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    o = base.clone().to(dev)
    torch.ops.chipmunk.csp_attn(qd, kd, vd, o, indd, cntd, -1)
    for h in (0, 11, 23):
        ref = base[:, h:h + 1].clone()
        oracle.csp_attn(q[:, h:h + 1].contiguous(), k[:, h:h + 1].contiguous(), v[:, h:h + 1].contiguous(), ref,
                        inds[:, h:h + 1].contiguous(), counts[:, h:h + 1].contiguous(), -1)
        assert_close_bf16(o[:, h:h + 1], ref, what=f"C2 csp_attn head {h} vs oracle")
    out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base.to(dev), indd, cntd, -1)
    assert torch.equal(out, o)
    back = o.clone()
    torch.ops.chipmunk.csp_attn(qd, kd, vd, back, indd, cntd, 1)
    assert_close_bf16(back, base, atol=2e-2, rtol=2e-2, what="C2 (o - sparse) + sparse")


def test_c5_wan_fp8_gemm1_full_size(dev):
    """BASELINE configs[4] shapes: Wan2.1-1.3B MLP, M = 32 768 rows, K = 1536, F = 8960, 30 % of the columns kept per
    128-row group, fp8 e4m3 operands.  Four row groups are checked against fp32 torch math on the dequantised operands
    (gelu_tanh((a8 . b8[idx]) * sa * sb + bias) - cache); columns past the count and un-selected cache columns untouched."""
    M, K, F = 32768, 1536, 8960
    keep = 2816                                         # 0.3 * 8960 rounded up to a multiple of 256
    g = torch.Generator(device=dev).manual_seed(51)
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(F, K, device=dev, generator=g) * 0.05
    sa, sb = 448.0 / x.abs().max(), 448.0 / w.abs().max()
    a8, b8 = (x * sa).to(torch.float8_e4m3fn), (w * sb).to(torch.float8_e4m3fn)
    bias = (torch.randn(F, device=dev, generator=g) * 0.2).to(torch.bfloat16)
    cache = (torch.randn(F, M, device=dev, generator=g) * 0.3).to(torch.bfloat16)
    cache0 = cache.clone()
    Gm = M // 128
    inds = torch.stack([torch.randperm(F, device=dev, generator=g) for _ in range(Gm)]).to(torch.int32)
    counts = torch.full((Gm,), keep, dtype=torch.int32, device=dev)
    counts[7] = 0
    counts[100] = 256
    packed = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    ra, rb = (1.0 / sa).reshape(1).float(), (1.0 / sb).reshape(1).float()
    torch.ops.chipmunk.csp_mlp_mm1_fp8(a8, b8, packed, bias, cache, inds, counts, ra, rb, True)
    for gi in (0, 7, 100, Gm - 1):
        rows = slice(gi * 128, (gi + 1) * 128)
        n = int(counts[gi])
        cols = inds[gi, :n].long()
        acc = (a8[rows].float() @ b8[cols].float().T) * ra * rb + bias[cols].float()
        act = torch.nn.functional.gelu(acc, approximate="tanh").to(torch.bfloat16)
        want = (act.float() - cache0[cols][:, rows].float().T).to(torch.bfloat16)
        assert_close_bf16(packed[rows, :n], want, atol=3e-2, rtol=3e-2, what=f"C5 fp8 GEMM1 group {gi}")
        assert (packed[rows, n:] == 7.0).all(), "columns past the count are not written"
        assert_close_bf16(cache[cols][:, rows], act.T, atol=3e-2, rtol=3e-2, what=f"C5 cache update group {gi}")
        rest = inds[gi, n:].long()
        assert torch.equal(cache[rest][:, rows], cache0[rest][:, rows]), "unselected cache columns keep their bits"


def test_c5_wan_attention_shapes(dev):
    """Wan2.1 832x480x81: 12 heads, 32 760 tokens (171 query groups, the last one ragged).  Dense vs SDPA on two row
    slices; sparse with identity lists == dense; sparse with random 10 % lists: one group per head against the oracle."""
    H, N = 12, 32760
    G = math.ceil(N / 192)
    g = torch.Generator(device=dev).manual_seed(61)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    o_dense, l = torch.ops.chipmunk.dense_attn(q, k, v)
    for rows in (slice(0, 384), slice(N - 300, N)):
        ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, rows].float(), k.float(), v.float())
        assert_close_bf16(o_dense[:, :, rows], ref, atol=1e-2, rtol=2e-2, what="Wan dense vs SDPA")
    count = 3328
    inds = torch.zeros(1, H, G, G * 192, dtype=torch.int32, device=dev)
    for h in range(H):
        r = torch.rand(G, N, device=dev, generator=g)
        inds[0, h, :, :count] = r.topk(count, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
    counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
    o = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
    qc, kc, vc, ic = q.cpu(), k.cpu(), v.cpu(), inds.cpu()
    for h, gi in ((0, 0), (5, 77), (11, G - 1)):
        r0, r1 = gi * 192, min(N, gi * 192 + 192)
        ref = oracle.csp_128_attn(qc[:, h:h + 1, r0:r1].contiguous(), kc[:, h:h + 1].contiguous(), vc[:, h:h + 1].contiguous(),
                                  ic[:, h:h + 1, gi:gi + 1, :N].contiguous(), counts[:, h:h + 1, gi:gi + 1].cpu().contiguous())
        assert_close_bf16(o[:, h:h + 1, r0:r1], ref, what=f"Wan sparse head {h} group {gi} vs oracle")


@pytest.mark.gpu
def test_plan_tail_cut_items_match_uncut_and_oracle():
    """The work plan's tail: 3 heads x 172 groups = 516 equal items on 512 resident slots leave 4 items for a round of their own;
    the plan cuts each of them into 16 slices (8 key tiles per slice), merged by the last arriver.  The other 512 items must not
    change by a bit against the plan without the cut (option attn_no_tail), the 4 cut ones agree with it to bf16 rounding and with
    the oracle, and the launch is run-to-run identical; both kernels that read the plan (attn96.hip, the general kernel)."""
    from chipmunk_amd import _native
    dev = torch.device("cuda:0")
    H, N, keep = 3, 33000, 4096     # (4 096 keys: below 1.5 x the plan's slice size, so no item is cut for length)
    G = (N + 191) // 192
    assert (H * G) % 512 == 4
    g = torch.Generator().manual_seed(91)
    q, k, v = [randn_bf16(1, H, N, 128, seed=60 + s) for s in range(3)]
    inds = torch.empty(1, H, G, N, dtype=torch.int32)
    for h in range(H):
        for gi in range(G):
            inds[0, h, gi] = torch.randperm(N, generator=g).to(torch.int32)
    inds[..., :keep] = inds[..., :keep].sort(-1).values
    counts = torch.full((1, H, G), keep, dtype=torch.int32)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    tail_rows = torch.zeros(H, G, dtype=torch.bool)
    tail_rows[H - 1, G - 4:] = True                     # the last four items of the (head, group) order
    for opt in (0, 2):                                   # attn96.hip (the shipped selection at this size), then the general kernel
        _native.set_option("attn_csp96", opt)
        try:
            o_cut = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
            assert torch.equal(o_cut, torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)), "run to run identical"
            _native.set_option("attn_no_tail", 1)
            try:
                o_plain = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
            finally:
                _native.set_option("attn_no_tail", 0)
        finally:
            _native.set_option("attn_csp96", 0)
        same = (o_cut == o_plain).all(dim=-1)[0].cpu()                                  # [H, N] rows that did not change
        row_is_tail = tail_rows.repeat_interleave(192, dim=1)[:, :N]
        assert bool(same[~row_is_tail].all()), "items outside the tail are computed exactly as without the cut"
        assert not bool(same[row_is_tail].all()), "the four tail items went through the slice merge"
        d = (o_cut.float() - o_plain.float()).abs().max().item()
        assert d <= 2.0 ** -7 * max(1e-3, o_plain.float().abs().max().item()), d
        for gi in (G - 4, G - 1):
            r0, r1 = gi * 192, min(N, gi * 192 + 192)
            o_ref = oracle.csp_128_attn(q[:, H - 1:, r0:r1].contiguous(), k[:, H - 1:], v[:, H - 1:], inds[:, H - 1:, gi:gi + 1].contiguous(),
                                        counts[:, H - 1:, gi:gi + 1].contiguous())
            assert_close_bf16(o_cut[:, H - 1:, r0:r1], o_ref, what=f"cut item (head {H - 1}, group {gi}) vs oracle, attn_csp96={opt}")


def test_c5_wan_cross_attention_shape(dev):
    """Wan2.1's cross-attention over the 512 text tokens as the Wan workload calls it (tools/wan_workload.py): q = head views of a
    [32 768, 12 * 128] projection output, k / v = head views of one [512, 2, 12, 128] projection output, output token-major.
    chipmunk.dense_attn_layout vs fp32 SDPA on row slices, and the token-major view in front of the output projection is free."""
    H, M, T, D = 12, 32768, 512, 128
    g = torch.Generator(device=dev).manual_seed(71)
    xq = torch.randn(M, H * D, device=dev, dtype=torch.bfloat16, generator=g)
    xkv = torch.randn(1, T, 2, H, D, device=dev, dtype=torch.bfloat16, generator=g)
    q = xq.view(1, M, H, D).transpose(1, 2)
    k, v = xkv[:, :, 0].transpose(1, 2), xkv[:, :, 1].transpose(1, 2)
    assert not q.is_contiguous() and not k.is_contiguous()
    o, l = torch.ops.chipmunk.dense_attn_layout(q, k, v, True)
    assert o.shape == (1, H, M, D) and l.shape == (1, H, M, 1)
    flat = o.transpose(1, 2).reshape(M, H * D)
    assert flat.data_ptr() == o.data_ptr(), "token-major output: the head merge is a view"
    for rows in (slice(0, 256), slice(17000, 17300), slice(M - 192, M)):
        ref = torch.nn.functional.scaled_dot_product_attention(q[:, :, rows].float(), k.float(), v.float())
        assert_close_bf16(o[:, :, rows], ref, atol=1e-2, rtol=2e-2, what="Wan cross-attention vs SDPA")
    # the head-major form gives the same values
    o2, _ = torch.ops.chipmunk.dense_attn(q, k, v)
    assert torch.equal(o2, o)
