"""GPU parity of the four attention ops against the CPU oracle (calls go torch.ops.chipmunk.* -> C ABI -> HIP).

Tolerances (stated per SURVEY 8c): bf16 outputs atol = rtol = 2e-2 vs the fp32-accumulating oracle; l (fp32) rtol 1e-3;
cs (bf16 sums of up to 192 bf16 terms, reduction order free) rtol 3e-2 + atol 2e-3.
"""
import math

import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16, random_index_sets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401  (loads the HIP library and registers torch.ops.chipmunk)
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _qkv(B, H, Nq, Nk, seed):
    return (randn_bf16(B, H, Nq, 128, seed=seed), randn_bf16(B, H, Nk, 128, seed=seed + 1),
            randn_bf16(B, H, Nk, 128, seed=seed + 2))


@pytest.mark.parametrize("n", [384, 512, 1000, 1984])
def test_dense_attn_vs_oracle_and_sdpa(dev, n):
    """reference tests/test_dense_attn.py:29-36 (vs SDPA) + the oracle; ragged n exercises the masked last group/tile."""
    q, k, v = _qkv(1, 3, n, n, seed=n)
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert o.shape == q.shape and l.shape == (1, 3, n, 1) and l.dtype == torch.float32
    assert_close_bf16(o, o_ref, what="dense_attn o")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    sdpa = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    assert_close_bf16(o, sdpa, what="dense_attn vs SDPA")


def test_dense_attn_strided_inputs(dev):
    """reference tests/test_dense_attn.py:18-27: [B,N,H,D] storage viewed as [B,H,N,D]."""
    B, H, N = 1, 4, 576
    base = [randn_bf16(B, N, H, 128, seed=s).to(dev) for s in (1, 2, 3)]
    q, k, v = [t.permute(0, 2, 1, 3) for t in base]
    assert not q.is_contiguous()
    o, l = torch.ops.chipmunk.dense_attn(q, k, v)
    o_ref, l_ref = oracle.dense_attn(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous())
    assert_close_bf16(o, o_ref, what="strided dense_attn")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)


@pytest.mark.parametrize("n", [4480, 4592])
def test_csp_attn_identity_indices_is_sdpa(dev, n):
    """reference tests/test_csp_attn.py:30-38: identity indices, counts = n, into a zero o => SDPA."""
    H = 2
    q, k, v = _qkv(1, H, n, n, seed=7)
    G = math.ceil(n / 192)
    inds = torch.arange(n, dtype=torch.int32).expand(1, H, G, n).contiguous()
    counts = torch.full((1, H, G), n, dtype=torch.int32)
    o = torch.zeros_like(q).to(dev)
    torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), o, inds.to(dev), counts.to(dev), 1)
    sdpa = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    assert_close_bf16(o, sdpa, what="csp_attn identity")


@pytest.mark.parametrize("o_scale", [1, -1])
@pytest.mark.parametrize("n,count", [(960, 224), (1100, 336)])
def test_csp_attn_inplace_random_indices(dev, n, count, o_scale):
    """in-place accumulate with non-identity index sets and both signs (no reference test covers this)."""
    H = 2
    q, k, v = _qkv(1, H, n, n, seed=11)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=5)
    o0 = randn_bf16(1, H, n, 128, seed=99)
    o_ref = o0.clone()
    oracle.csp_attn(q, k, v, o_ref, inds, counts, o_scale)
    o = o0.clone().to(dev)
    torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), o, inds.to(dev), counts.to(dev), o_scale)
    assert_close_bf16(o, o_ref, atol=3e-2, what="csp_attn in place")


@pytest.mark.parametrize("o_scale", [1, -1])
def test_csp_attn_out_equals_clone_plus_inplace(dev, o_scale):
    """csp_attn_out == `o = base.clone(); csp_attn(q, k, v, o, ...)` bit for bit (same kernel, different base
    pointer), leaves the base untouched, and copies the base for a group that keeps no key."""
    H, n, count = 2, 1100, 336
    q, k, v = _qkv(1, H, n, n, seed=13)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=6)
    counts[0, 1, 2] = 0
    base = randn_bf16(1, H, n, 128, seed=98).to(dev)
    keep = base.clone()
    qd, kd, vd, indd, cntd = q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev)
    ref = base.clone()
    torch.ops.chipmunk.csp_attn(qd, kd, vd, ref, indd, cntd, o_scale)
    out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, o_scale)
    assert torch.equal(base, keep)
    assert torch.equal(out, ref)
    assert torch.equal(out[0, 1, 2 * 192:3 * 192], base[0, 1, 2 * 192:3 * 192])


def test_csp_attn_key_split_forced(dev):
    """The key-split tail is off for gathered launches by default (it does not pay at FLUX sizes); forced on, the in-place
    gathered kernel must still match the oracle -- including ragged counts, a group with no keys and slices of 0 tiles."""
    from chipmunk_amd import _native
    H, n = 2, 1100
    q, k, v = _qkv(1, H, n, n, seed=17)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 672, n, seed=8)
    counts[0, 0, 1] = 0
    counts[0, 1, 3] = 96
    o0 = randn_bf16(1, H, n, 128, seed=97)
    o_ref = o0.clone()
    oracle.csp_attn(q, k, v, o_ref, inds, counts, 1)
    o = o0.clone().to(dev)
    _native.set_option("attn_split_gather", 1)
    try:
        torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), o, inds.to(dev), counts.to(dev), 1)
    finally:
        _native.set_option("attn_split_gather", 0)
    assert_close_bf16(o, o_ref, atol=3e-2, what="csp_attn with forced key split")


def test_csp_attn_strided_qkv(dev):
    n, H, count = 768, 3, 224
    base = [randn_bf16(1, n, H, 128, seed=s) for s in (21, 22, 23)]
    q, k, v = [t.permute(0, 2, 1, 3) for t in base]
    G = n // 192
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=6)
    o_ref = torch.zeros(1, H, n, 128, dtype=torch.bfloat16)
    oracle.csp_attn(q, k, v, o_ref, inds, counts, 1)
    o = torch.zeros(1, n, H, 128, dtype=torch.bfloat16, device=dev).permute(0, 2, 1, 3)
    torch.ops.chipmunk.csp_attn(*[t.to(dev).permute(0, 2, 1, 3) for t in base], o, inds.to(dev), counts.to(dev), 1)
    assert_close_bf16(o, o_ref, what="strided csp_attn")


@pytest.mark.parametrize("n,nk,count", [(768, 768, 256), (1152, 1100, 384), (960, 960, 64)])
def test_csp_128_attn_random_indices(dev, n, nk, count):
    H = 2
    q, k, v = _qkv(1, H, n, nk, seed=31)
    G = n // 192
    inds, counts = random_index_sets(1, H, G, nk, count, n, seed=8)
    # ragged per-group counts, multiples of 16 (the kernel accepts any multiple; the reference wants 128)
    counts[0, 0, 0] = max(16, count - 48)
    counts[0, 1, G - 1] = max(16, count - 16)
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what="csp_128_attn")


def test_csp_attn_full_minus_sparse_roundtrip(dev):
    """Size-independent property used by the module: o_cache = o - sparse; o_cache + sparse == o (bf16 tolerance)."""
    n, H, count = 4352, 4, 672  # FLUX C2 shape per head
    q, k, v = _qkv(1, H, n, n, seed=41)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=9)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    o, _ = torch.ops.chipmunk.dense_attn(qd, kd, vd)
    cache = o.clone()
    torch.ops.chipmunk.csp_attn(qd, kd, vd, cache, indd, cntd, -1)
    back = cache.clone()
    torch.ops.chipmunk.csp_attn(qd, kd, vd, back, indd, cntd, 1)
    assert_close_bf16(back, o, atol=3e-2, what="cache roundtrip")


@pytest.mark.parametrize("n", [576, 1000])
def test_dense_colsum_attn(dev, n):
    """reference tests/test_dense_colsum_attn.py:13-36 (fp32 column-sum formula) + the oracle."""
    H = 2
    q, k, v = _qkv(1, H, n, n, seed=51)
    # a second, correlated step (q' = q + 0.1 eps) as in the real pipeline
    q2 = (q.float() + 0.1 * torch.randn(q.shape, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
    _, l0 = oracle.dense_attn(q, k, v)
    o_ref, cs_ref, l_ref = oracle.dense_colsum_attn(q2, k, v, l0)
    o, cs, l = torch.ops.chipmunk.dense_colsum_attn(q2.to(dev), k.to(dev), v.to(dev), l0.to(dev))
    G = math.ceil(n / 192)
    assert cs.shape == (1, H, G, n) and cs.dtype == torch.bfloat16
    assert_close_bf16(o, o_ref, what="colsum o")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    assert_close_bf16(cs, cs_ref, atol=2e-3, rtol=3e-2, what="colsum cs vs oracle")
    # the reference test's fp32 formula: p = exp(logits) * l_prev, summed over each 192-row group
    logits = (q2.float() @ k.float().transpose(-1, -2)) / math.sqrt(128)
    p = torch.exp(logits) * l0
    pad = G * 192 - n
    p = torch.nn.functional.pad(p, (0, 0, 0, pad)).view(1, H, G, 192, n).sum(3)
    assert_close_bf16(cs, p, atol=4e-3, rtol=4e-2, what="colsum cs vs fp32 formula")


def test_batched_inputs_all_attention_ops(dev):
    """B = 2 (the reference kernels pin the batch index to 0, csp_128_attn.cu:82; the C ABI takes B): every attention op
    against the oracle, which loops over the batch."""
    B, H, n, count = 2, 2, 768, 256
    q, k, v = _qkv(B, H, n, n, seed=61)
    G = n // 192
    inds, counts = random_index_sets(B, H, G, n, count, n, seed=12)
    counts[1, 0, 1] = 128
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(qd, kd, vd)
    assert_close_bf16(o, o_ref, what="dense B=2")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)
    assert_close_bf16(torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd), oracle.csp_128_attn(q, k, v, inds, counts),
                      what="csp_128 B=2")
    base = randn_bf16(B, H, n, 128, seed=62)
    ref = base.clone()
    oracle.csp_attn(q, k, v, ref, inds, counts, -1)
    got = base.clone().to(dev)
    torch.ops.chipmunk.csp_attn(qd, kd, vd, got, indd, cntd, -1)
    assert_close_bf16(got, ref, atol=3e-2, what="csp_attn B=2")
    assert torch.equal(torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base.to(dev), indd, cntd, -1), got)
    o2_ref, cs_ref, l2_ref = oracle.dense_colsum_attn(q, k, v, l_ref)
    o2, cs, l2 = torch.ops.chipmunk.dense_colsum_attn(qd, kd, vd, l)
    assert_close_bf16(o2, o2_ref, what="colsum o B=2")
    assert_close_bf16(cs, cs_ref, atol=2e-3, rtol=3e-2, what="colsum cs B=2")


def test_packed_positions_past_the_key_count_are_masked(dev):
    """right_fill (csp_128_attn.cu:314): with more queries than keys, packed positions >= N_k are masked even when
    counts says otherwise -- the kernel clamps counts to N_k like the oracle does."""
    H, n, nk = 1, 768, 384
    q, k, v = _qkv(1, H, n, nk, seed=71)
    G = n // 192
    inds = torch.arange(n, dtype=torch.int32).remainder(nk).view(1, 1, 1, n).expand(1, H, G, n).contiguous()
    counts = torch.full((1, H, G), 640, dtype=torch.int32)  # > nk: only the first nk packed positions count
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    o = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o, o_ref, what="right_fill")
    sdpa = torch.nn.functional.scaled_dot_product_attention(q.float(), k.float(), v.float())
    assert_close_bf16(o, sdpa, what="right_fill == attention over all nk keys")


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_running_max_update_paths(dev, pattern):
    """The kernel's exponentials are taken against a reference point that lags the running row maximum by up to 2^4 and
    is raised (with a rescale of O and l once the pending PV has landed) only when some query column outgrows the lag.
    Bounded random data almost never takes that branch, so force it: scores that grow tile after tile ("ramp"), one
    key far above everything late in the list for SOME query rows only ("spike"), the same in the very first tile, and
    scores that only shrink ("descending": the branch must never be needed).  Dense and gathered kernels vs the oracle,
    `l` included (it carries the reference point)."""
    n, H = 1152, 2
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, n, 128, generator=g)
    k = torch.randn(1, H, n, 128, generator=g)
    v = torch.randn(1, H, n, 128, generator=g)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    q = 0.3 * q + 3.0 * u                      # every query has a component along u ...
    if pattern == "ramp":                      # ... and key j has a growing one: scores rise by ~0.25 nats per 32-key tile
        k = 0.3 * k + (torch.arange(n).float() / n * 30.0)[None, None, :, None] * u
    elif pattern == "descending":
        k = 0.3 * k + ((n - torch.arange(n)).float() / n * 30.0)[None, None, :, None] * u
    else:
        k = 0.3 * k
        j = 5 if pattern == "spike_first" else 1000
        k[0, :, j] += 40.0 * u                 # q.k/sqrt(128) ~ +10 nats for the rows below
        q[0, :, ::3] *= 0.05                   # two thirds of the rows see it, the others barely: lanes of one wave disagree
    q, k, v = [t.to(torch.bfloat16) for t in (q, k, v)]
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense, {pattern}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=2e-3, atol=0)
    G = math.ceil(n / 192)
    inds = torch.arange(n, dtype=torch.int32).expand(1, H, G, n).contiguous()
    counts = torch.full((1, H, G), n, dtype=torch.int32)
    o2 = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), inds.to(dev), counts.to(dev))
    assert_close_bf16(o2, o_ref, what=f"gathered with identity lists, {pattern}")
    # a gathered list that visits the keys in DEscending score order for the ramp = only the first tile sets the reference
    rev = torch.arange(n - 1, -1, -1, dtype=torch.int32).expand(1, H, G, n).contiguous()
    o3 = torch.ops.chipmunk.csp_128_attn(q.to(dev), k.to(dev), v.to(dev), rev.to(dev), counts.to(dev))
    assert_close_bf16(o3, o_ref, what=f"gathered in reverse key order, {pattern}")
    base = randn_bf16(1, H, n, 128, seed=3)
    acc = base.clone().to(dev)
    torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), acc, inds.to(dev), counts.to(dev), 1)
    assert_close_bf16(acc, (base.float() + o_ref.float()).to(torch.bfloat16), what=f"in-place accumulate, {pattern}")
