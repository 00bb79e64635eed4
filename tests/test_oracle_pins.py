"""Pins the CPU oracle against every check the reference's own tests hold for the hot path:
formulas, not stored vectors -- regenerated here against torch on CPU.  The reference's tests need an H100; these run
anywhere.  Ops without any reference test (mask_to_indices, topk_indices, copy_indices, scatter_add, mm2) are pinned
only by the restatement of the cited kernel source ("parity unpinned by the reference") plus the properties below.
"""
import math

import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16, random_index_sets


def _sdpa(q, k, v):
    return torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())


@pytest.mark.parametrize("n", [512, 700])
def test_dense_attn_matches_sdpa_and_l_definition(n):
    """reference tests/test_dense_attn.py:29-36; l = 1/sum_j exp(s_ij/sqrt(D)) (dense_attn.cu:225-227)."""
    q, k, v = [randn_bf16(1, 2, n, 128, seed=s) for s in (1, 2, 3)]
    o, l = oracle.dense_attn(q, k, v)
    assert_close_bf16(o, _sdpa(q, k, v), what="oracle dense_attn vs SDPA")
    lref = 1.0 / torch.exp((q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128)).sum(-1, keepdim=True)
    torch.testing.assert_close(l, lref, rtol=1e-4, atol=0)


@pytest.mark.parametrize("n,tile", [(448, 112), (560, 112), (512, 128)])
def test_csp_attn_identity_indices_is_sdpa(n, tile):
    """reference tests/test_csp_attn.py:30-38: identity indices + counts = n into zero o == SDPA (kv tile 112 / 128)."""
    H = 2
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (4, 5, 6)]
    G = math.ceil(n / 192)
    inds = torch.arange(n, dtype=torch.int32).expand(1, H, G, n).contiguous()
    counts = torch.full((1, H, G), n, dtype=torch.int32)
    o = torch.zeros_like(q)
    oracle.csp_attn(q, k, v, o, inds, counts, 1, kv_tile=tile)
    assert_close_bf16(o, _sdpa(q, k, v), what="oracle csp_attn identity")
    if n % 192 == 0:
        assert_close_bf16(oracle.csp_128_attn(q, k, v, inds, counts), _sdpa(q, k, v), what="oracle csp_128 identity")


def test_csp_attn_subset_equals_masked_softmax():
    """definition check: attention restricted to the listed keys; -1 scale subtracts; bf16 accumulate into o."""
    n, H, count = 384, 2, 128
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (7, 8, 9)]
    inds, counts = random_index_sets(1, H, 2, n, count, n, seed=1)
    logits = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128)
    mask = torch.zeros(1, H, n, n, dtype=torch.bool)
    for h in range(H):
        for g in range(2):
            mask[0, h, g * 192:(g + 1) * 192, inds[0, h, g, :count].long()] = True
    ref = torch.softmax(logits.masked_fill(~mask, float("-inf")), -1) @ v.float()
    o0 = randn_bf16(1, H, n, 128, seed=10)
    o = o0.clone()
    oracle.csp_attn(q, k, v, o, inds, counts, -1)
    assert_close_bf16(o, o0.float() - ref, atol=3e-2, what="oracle csp_attn -1")


def test_dense_colsum_matches_fp32_formula():
    """reference tests/test_dense_colsum_attn.py:13-36: p = exp(logits - rowmax) scaled by last step's constants,
    rows padded to 192, summed per 192-row group."""
    n, H = 576, 2
    q, k, v = [randn_bf16(1, H, n, 128, seed=s) for s in (11, 12, 13)]
    _, l = oracle.dense_attn(q, k, v)
    o, cs, l2 = oracle.dense_colsum_attn(q, k, v, l)
    assert_close_bf16(o, _sdpa(q, k, v), what="colsum o")
    torch.testing.assert_close(l2, l, rtol=1e-5, atol=0)
    p = torch.softmax((q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128), -1)  # exp(logits) * l == softmax
    ref = p.view(1, H, n // 192, 192, n).sum(3)
    assert cs.shape == (1, H, 3, n)
    assert_close_bf16(cs, ref, atol=2e-3, rtol=3e-2, what="colsum vs fp32 formula")
    assert torch.allclose(cs.float().sum(-1), torch.full((1, H, 3), 192.0), rtol=2e-2)  # each row of P sums to 1


def test_mm1_known_answer_recipe():
    """reference csrc/mlp/csp_mlp_mm1.cu:401-424,458-486: reversed identity indices, U(-0.5, 0.5) data, tolerance 0.1."""
    M, K, F = 128, 256, 512
    g = torch.Generator().manual_seed(42)
    a, b = [(torch.rand(s, generator=g) - 0.5).to(torch.bfloat16) for s in ((M, K), (F, K))]
    bias, cache = [(torch.rand(s, generator=g) - 0.5).to(torch.bfloat16) for s in ((F,), (F, M))]
    inds = torch.arange(F - 1, -1, -1, dtype=torch.int32).view(1, F).contiguous()
    counts = torch.tensor([F], dtype=torch.int32)
    c = torch.zeros(M, F, dtype=torch.bfloat16)
    oracle.csp_mlp_mm1(a, b, c, bias, cache, inds, counts)
    ref = torch.nn.functional.gelu(a.float() @ b.float().T + bias.float(), approximate="tanh") - cache.float().T
    assert (c.float() - ref.flip(1)).abs().max() < 0.1
    assert_close_bf16(c, ref.flip(1), what="oracle mm1")


def test_mm2_and_scatter_add_definitions():
    M, F, N2 = 256, 512, 128
    packed, unp, w2t, out = randn_bf16(M, F, seed=1, scale=0.3), randn_bf16(F, M, seed=2), randn_bf16(F, N2, seed=3, scale=0.2), randn_bf16(M, N2, seed=4)
    counts = torch.tensor([64, 200], dtype=torch.int32)
    inds = torch.stack([torch.randperm(F, generator=torch.Generator().manual_seed(i)).int() for i in range(2)])
    unp_ref, out_ref = unp.float().clone(), out.float().clone()
    for g in range(2):
        rows = slice(g * 128, (g + 1) * 128)
        cols = inds[g, :counts[g]].long()
        unp_ref[cols, rows] += packed[rows, :counts[g]].float().T
        out_ref[rows] += (packed[rows, :counts[g]].float() @ w2t[cols].float()).to(torch.bfloat16).float()
    oracle.csp_mlp_mm2_and_scatter_add(packed, unp, inds, counts, packed, w2t, out)
    assert torch.equal(unp, unp_ref.to(torch.bfloat16))      # one bf16 add per element: exact
    assert_close_bf16(out, out_ref, what="oracle mm2")


def test_mask_to_indices_order_and_padding():
    """lane-interleaved order (mask_to_indices.cu:49-68) and first-False padding (:71-86) on a hand-checkable row."""
    n = 70
    mask = torch.zeros(1, 1, 1, n, dtype=torch.bool)
    true_cols = [0, 1, 5, 32, 33, 37, 64, 69]
    mask[0, 0, 0, true_cols] = True
    inds, counts = oracle.mask_to_indices(mask, 16, 192)
    assert inds.shape == (1, 1, 1, 192) and counts.item() == 16
    # class 0: 0,32,64; class 1: 1,33; class 5: 5,37,69 ; then first False columns 2,3,4,6,7,8,9,10
    assert inds[0, 0, 0, :16].tolist() == [0, 32, 64, 1, 33, 5, 37, 69, 2, 3, 4, 6, 7, 8, 9, 10]


def test_topk_indices_threshold_rule_and_edges():
    C = 2048
    g = torch.Generator().manual_seed(3)
    act = torch.rand(1, 2, C, generator=g)
    inds = torch.full((1, 2, C), -5, dtype=torch.int32)
    counts = torch.zeros(1, 2, dtype=torch.int32)
    oracle.topk_indices(act, inds, counts, 0.75, 64, 0.0)
    for r in range(2):
        thr = act[0, r, :1024].sort().values[768]                     # rank int(1024*0.75) of the first 1024 only
        kept = torch.nonzero(act[0, r] >= thr).flatten().tolist()
        n = counts[0, r].item()
        assert n % 64 == 0 and n >= len(kept)
        assert inds[0, r, :len(kept)].tolist() == kept                # ascending canonical order
        pad = inds[0, r, len(kept):n].tolist()
        assert len(set(pad)) == len(pad) and all(act[0, r, c] < thr for c in pad)
    oracle.topk_indices(act, inds, counts, 0.0, 64, 0.0)
    assert counts.tolist() == [[C, C]] and inds[0, 0].tolist() == list(range(C))
    oracle.topk_indices(act, inds, counts, 1.0, 64, 0.0)
    assert counts.tolist() == [[0, 0]] and (inds == -1).all()


def test_copy_indices_and_bitpack_roundtrip():
    src, dst = torch.randn(1, 4, 64), torch.zeros(1, 4, 64)
    inds = torch.stack([torch.randperm(64, generator=torch.Generator().manual_seed(i)).int() for i in range(2)])[None]
    counts = torch.tensor([[3, 64]], dtype=torch.int32)
    oracle.copy_indices(src, dst, inds, counts)   # R = 2 rows share each index row
    for row in range(4):
        cols = inds[0, row // 2, :counts[0, row // 2]].long()
        assert torch.equal(dst[0, row, cols], src[0, row, cols])
        assert dst[0, row].count_nonzero() == len(cols)
    m = torch.rand(5, 13, generator=torch.Generator().manual_seed(1)) < 0.5
    packed, shape = oracle.bitpack(m)
    assert packed.numel() == math.ceil(65 / 8)
    assert torch.equal(oracle.bitunpack(packed, shape), m)
    assert packed[0].item() == sum(int(b) << i for i, b in enumerate(m.flatten()[:8].tolist()))


def test_fp8_mm1_oracle_matches_torch_dequant_math():
    """fp8 GEMM1 restatement vs torch: decode e4m3fn exactly, scale, bias, tanh-GeLU, bf16 round, bf16 subtract."""
    M, K, F = 128, 256, 256
    g = torch.Generator().manual_seed(11)
    a8 = (torch.randn(M, K, generator=g) * 20).to(torch.float8_e4m3fn)
    b8 = (torch.randn(F, K, generator=g) * 20).to(torch.float8_e4m3fn)
    bias, cache = randn_bf16(F, seed=1, scale=0.3), randn_bf16(F, M, seed=2)
    inds = torch.randperm(F, generator=g).int().view(1, F).contiguous()
    cnt = torch.tensor([F], dtype=torch.int32)
    c, cache2 = torch.zeros(M, F, dtype=torch.bfloat16), cache.clone()
    oracle.csp_mlp_mm1_fp8(a8, b8, c, bias, cache2, inds, cnt, 0.01, 0.02, update_cache=True)
    x = torch.nn.functional.gelu((a8.float() @ b8.float().T) * 0.01 * 0.02 + bias.float(), approximate="tanh")
    x = x.to(torch.bfloat16)
    cols = inds[0].long()
    assert_close_bf16(c, (x[:, cols] - cache.T[:, cols]), what="fp8 packed")
    assert_close_bf16(cache2.T, x, what="fp8 cache update")       # every column selected -> whole cache rewritten


def test_f8linear_scale_arithmetic():
    """reference modules/mlp_fp8.py:172-195: amax_to_scale / saturating cast / reciprocal scales."""
    from chipmunk_amd.modules.mlp_fp8 import F8Linear, amax_to_scale, to_fp8_saturated
    assert amax_to_scale(torch.tensor(2.0), 448.0).item() == 224.0
    assert amax_to_scale(torch.tensor(0.0), 448.0).item() == 448.0       # clamp(max=max_val) with amax -> 1e-12
    x = torch.tensor([-1000.0, -1.0, 0.5, 3.0])
    assert to_fp8_saturated(x, torch.tensor(224.0), 448.0).tolist() == [-448.0, -224.0, 112.0, 448.0]
    lin = torch.nn.Linear(64, 32).bfloat16()
    w = lin.weight.data.clone()
    f8 = F8Linear.from_linear(lin, input_float8_dtype=torch.float8_e4m3fn)
    assert f8.weight.dtype == torch.float8_e4m3fn and f8.weight_initialized
    torch.testing.assert_close(f8.scale * f8.scale_reciprocal, torch.tensor(1.0), rtol=1e-6, atol=0)
    deq = f8.weight.float() * f8.scale_reciprocal
    assert (deq - w.float()).abs().max() <= w.float().abs().max() / 16 + 1e-6      # 3 mantissa bits
    xin = torch.randn(4, 64)
    q1 = f8.quantize_input(xin)
    assert q1.dtype == torch.float8_e4m3fn and f8.trial_index == 1 and not f8.input_scale_initialized
    assert torch.isclose(f8.input_scale, 448.0 / xin.abs().max())
    for _ in range(f8.num_scale_trials):
        f8.quantize_input(xin * 0.5)
    assert f8.input_scale_initialized                                   # frozen after the calibration trials
