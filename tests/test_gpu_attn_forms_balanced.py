"""The work-balanced launch of the gathered attention kernel (csrc/attn.hip, template flag BAL; reference ops
csrc/attn/csp_attn.cu:315-423 and csp_128_attn.cu:355-461): the launch's key tiles are one line that the resident workgroups
share equally, an item cut between two neighbours is CONTINUED by the second from the (O, m, l) state the first published.

Same tolerances as tests/test_gpu_attn.py (bf16 outputs atol = rtol = 2e-2 vs the oracle, 3e-2 for the accumulate form);
run-to-run the launch is bit-stable (the cuts are a function of the counts and the slot count, not of arrival order).
"""
import math

import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16, random_index_sets

import os

# The balanced launch is one of the measured-and-not-shipped forms (tools/probes/mm1_forms/build.sh compiles it in): these tests run when the
# library under test is that build -- tests/test_gpu_mlp_forms.py starts them in a subprocess bound to it.
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("CHIPMUNK_MM1_FORMS") != "1", reason="needs the probe-forms library (attn_balanced)")]


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class balanced:
    """attn_balanced: 1 = always, 3 = by shape, 0 / 2 = never"""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        from chipmunk_amd import _native
        _native.set_option("attn_balanced", self.v)

    def __exit__(self, *a):
        from chipmunk_amd import _native
        _native.set_option("attn_balanced", 0)


def _qkv(B, H, Nq, Nk, seed):
    return (randn_bf16(B, H, Nq, 128, seed=seed), randn_bf16(B, H, Nk, 128, seed=seed + 1),
            randn_bf16(B, H, Nk, 128, seed=seed + 2))


def host_cuts(counts, nk, nwg, min_tiles=3):
    """Mirror of the kernel's share arithmetic: the (item, first tile of the right part) of every cut item."""
    nt = [max(1, (max(0, min(int(c), nk)) + 31) // 32) for c in counts.flatten().tolist()]
    pre = [0]
    for t in nt:
        pre.append(pre[-1] + t)
    W, cuts = pre[-1], []
    for L in range(1, nwg):
        a = W * L // nwg
        j = max(i for i in range(len(nt)) if pre[i] <= a)
        off = a - pre[j]
        if off < min_tiles or nt[j] - off < min_tiles:
            continue
        cuts.append((j, off))
    return cuts


@pytest.mark.parametrize("o_scale", [1, -1])
def test_balanced_forced_small_ragged_counts(dev, o_scale):
    """12 items on 12 workgroups with ragged counts (0, 96, 672 ...): shares are cut inside items; in place, both signs."""
    H, n = 2, 1100
    q, k, v = _qkv(1, H, n, n, seed=17)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 672, n, seed=8)
    counts[0, 0, 1] = 0
    counts[0, 1, 3] = 96
    counts[0, 1, 0] = 333
    assert len(host_cuts(counts, n, H * G)) >= 3
    o0 = randn_bf16(1, H, n, 128, seed=97)
    o_ref = o0.clone()
    oracle.csp_attn(q, k, v, o_ref, inds, counts, o_scale)
    o = o0.clone().to(dev)
    with balanced(1):
        torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), o, inds.to(dev), counts.to(dev), o_scale)
    assert_close_bf16(o, o_ref, atol=3e-2, what="balanced csp_attn")
    assert torch.equal(o[0, 0, 192:384].cpu(), o0[0, 0, 192:384])   # the group without keys is left as it was


def test_balanced_out_of_place_forms(dev):
    """csp_128_attn (plain output) and csp_attn_out (base + result into a new tensor, a keyless group copies its base)."""
    H, n, nk = 2, 1152, 1100
    q, k, v = _qkv(1, H, n, nk, seed=31)
    G = n // 192
    inds, counts = random_index_sets(1, H, G, nk, 384, n, seed=8)
    counts[0, 0, 0] = 336
    counts[0, 1, G - 1] = 368
    o_ref = oracle.csp_128_attn(q, k, v, inds, counts)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    with balanced(1):
        o = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
    assert_close_bf16(o, o_ref, what="balanced csp_128_attn")
    counts2 = counts.clone()
    counts2[0, 1, 2] = 0
    base = randn_bf16(1, H, n, 128, seed=98).to(dev)
    ref = base.clone()
    with balanced(2):
        torch.ops.chipmunk.csp_attn(qd, kd, vd, ref, indd, counts2.to(dev), -1)
    with balanced(1):
        out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, counts2.to(dev), -1)
    assert_close_bf16(out, ref, atol=3e-2, what="balanced csp_attn_out vs the unbalanced in-place launch")
    assert torch.equal(out[0, 1, 2 * 192:3 * 192], base[0, 1, 2 * 192:3 * 192])


def test_balanced_item_longer_than_a_share_is_a_chain(dev):
    """One item with every key beside five short ones: it spans several shares, its middle parts consume AND publish."""
    H, n = 1, 1100
    q, k, v = _qkv(1, H, n, n, seed=23)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 1100, n, seed=3)
    for g in (0, 1, 3, 4, 5):
        counts[0, 0, g] = 32
    cuts = host_cuts(counts, n, G)
    assert sum(1 for j, _ in cuts if j == 2) >= 3, cuts
    o_ref = torch.zeros_like(q)
    oracle.csp_attn(q, k, v, o_ref, inds, counts, 1)
    o = torch.zeros_like(q).to(dev)
    with balanced(1):
        torch.ops.chipmunk.csp_attn(q.to(dev), k.to(dev), v.to(dev), o, inds.to(dev), counts.to(dev), 1)
    assert_close_bf16(o, o_ref, atol=3e-2, what="balanced chain")


def test_balanced_flux_launch_by_shape_named_cuts_and_bit_stability(dev, request):
    """The FLUX C2 launch (24 heads x 23 groups x 672 keys at n = 4352) takes the balanced form by shape: every cut item named by
    the host mirror is compared with the oracle, the whole output with the unbalanced launch, and 20 launches are bit-identical."""
    from chipmunk_amd import _native
    H, n, count = 24, 4352, 672
    q, k, v = _qkv(1, H, n, n, seed=41)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=9)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    base = randn_bf16(1, H, n, 128, seed=5).to(dev)
    nwg = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    cuts = host_cuts(counts, n, nwg)
    assert len(cuts) >= nwg // 2, "most shares end inside an item at this shape"
    _native.set_option("attn_row_split", 2)  # the plain launch: neither balanced nor with a row-split tail
    request.addfinalizer(lambda: _native.set_option("attn_row_split", 0))
    with balanced(2):
        plain = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    _native.set_option("attn_balanced", 3)   # by shape: this launch qualifies (552 items on 512 slots)
    request.addfinalizer(lambda: _native.set_option("attn_balanced", 0))
    out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    assert torch.equal(out, plain), "a continued item is the uncut item's own arithmetic: same bits"
    assert_close_bf16(out, plain, atol=3e-2, what="balanced vs unbalanced FLUX launch")
    # oracle on cut items: first, last, and a spread of 14 between (one head = 23 groups per call keeps it in seconds)
    picks = sorted({cuts[0][0], cuts[-1][0]} | {cuts[i][0] for i in range(0, len(cuts), max(1, len(cuts) // 14))})
    for item in picks:
        h, g = divmod(item, G)
        rows = slice(g * 192, min(n, (g + 1) * 192))
        o_ref = base[:, h:h + 1, rows].cpu().clone()
        oracle.csp_attn(q[:, h:h + 1, rows].contiguous(), k[:, h:h + 1], v[:, h:h + 1], o_ref,
                        inds[:, h:h + 1, g:g + 1].contiguous(), counts[:, h:h + 1, g:g + 1].contiguous(), 1)
        assert_close_bf16(out[:, h:h + 1, rows], o_ref, atol=3e-2, what=f"cut item (head {h}, group {g})")
    for _ in range(20):
        again = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
        assert torch.equal(again, out)
    # the library's counters are back at zero: an in-place launch right after, and one on another stream
    o2 = base.clone()
    torch.ops.chipmunk.csp_attn(qd, kd, vd, o2, indd, cntd, 1)
    assert torch.equal(o2, out)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        o3 = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    s.synchronize()
    assert torch.equal(o3, out)


def test_balanced_ragged_index_rows(dev):
    """csp_attn_out_ragged (kept keys as ragged rows) through the balanced launch == the padded form, bit for bit."""
    import chipmunk_amd
    H, n = 2, 1100
    q, k, v = _qkv(1, H, n, n, seed=19)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 640, n, seed=4)
    counts[0, 0, 2] = 40
    counts[0, 1, 1] = 0
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    base = randn_bf16(1, H, n, 128, seed=6).to(dev)
    flat, offs = chipmunk_amd.ops.compact_indices(indd, cntd)
    with balanced(1):
        a = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
        b = torch.ops.chipmunk.csp_attn_out_ragged(qd, kd, vd, base, flat, offs, cntd, 1)
    assert torch.equal(a, b)
