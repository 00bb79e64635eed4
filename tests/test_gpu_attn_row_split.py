"""The row-split tail of gathered attention launches (csrc/attn.hip, template flag MIX; reference ops csrc/attn/csp_attn.cu:315-423 and
csp_128_attn.cu:355-461): the last `items mod slots` items of a launch run as three 64-row workgroups with one query block per wave.

Same tolerances as tests/test_gpu_attn.py (bf16 outputs atol = rtol = 2e-2 vs the oracle, 3e-2 for the accumulate form); run to run bit-stable.
"""
import math

import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16, random_index_sets

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _qkv(B, H, Nq, Nk, seed):
    return (randn_bf16(B, H, Nq, 128, seed=seed), randn_bf16(B, H, Nk, 128, seed=seed + 1),
            randn_bf16(B, H, Nk, 128, seed=seed + 2))


# ---------------------------------------------------------------------------------------------------------------------------------
# Row-split tail (attn.hip, template flag MIX): the last `items mod slots` items of a gathered launch run as three 64-row workgroups
# with one 16-row query block per wave.  Rows are independent -- no partial state, no merge --, but the reference point of the
# exponentials moves per WAVE (when any of its query columns outgrows the lag), and a third's waves hold 16 rows instead of 48: the
# output agrees with the unsplit launch to bf16 rounding, not bit for bit; run to run it is bit-stable.
class row_split:
    """attn_row_split: 0 = by shape, 1 = always (the last min(items, slots / 3) items), 2 = never"""

    def __init__(self, v):
        self.v = v

    def __enter__(self):
        from chipmunk_amd import _native
        _native.set_option("attn_row_split", self.v)

    def __exit__(self, *a):
        from chipmunk_amd import _native
        _native.set_option("attn_row_split", 0)


@pytest.mark.parametrize("o_scale", [1, -1])
def test_row_split_forced_all_forms_bit_equal_and_vs_oracle(dev, o_scale):
    """n = 1100: the last group has 140 of its 192 rows (the third thirds hold rows past Nq); counts ragged incl. 0 and a ragged last tile."""
    H, n = 2, 1100
    q, k, v = _qkv(1, H, n, n, seed=29)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, 672, n, seed=12)
    counts[0, 0, 1] = 0
    counts[0, 1, 3] = 100
    counts[0, 1, 5] = 333
    o0 = randn_bf16(1, H, n, 128, seed=97)
    o_ref = o0.clone()
    oracle.csp_attn(q, k, v, o_ref, inds, counts, o_scale)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    outs = {}
    for opt in (2, 1):
        with row_split(opt):
            o = o0.clone().to(dev)
            torch.ops.chipmunk.csp_attn(qd, kd, vd, o, indd, cntd, o_scale)
            out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, o0.to(dev), indd, cntd, o_scale)
            plain = torch.ops.chipmunk.csp_128_attn(qd, kd, vd, indd, cntd)
            outs[opt] = (o, out, plain)
    for a, b in zip(outs[1], outs[2]):
        assert_close_bf16(a, b, what="row-split vs unsplit launch")
    assert_close_bf16(outs[1][0], o_ref, atol=3e-2, what="row-split csp_attn")
    assert torch.equal(outs[1][0][0, 0, 192:384].cpu(), o0[0, 0, 192:384])     # the group without keys is left as it was


def test_row_split_ragged_index_rows(dev):
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
    with row_split(2):
        ref = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    with row_split(1):
        a = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
        b = torch.ops.chipmunk.csp_attn_out_ragged(qd, kd, vd, base, flat, offs, cntd, 1)
    assert torch.equal(a, b), "ragged and padded index rows through the same row-split launch: same bits"
    assert_close_bf16(b, ref, what="row-split (ragged rows) vs unsplit launch")


def test_row_split_flux_launch_by_shape(dev):
    """The FLUX C2 launch (552 items on 2 x CUs slots): the last 552 mod slots items are split by shape; the items in front of them keep
    their bits, the whole output agrees with the unsplit launch, the split items with the oracle, 10 launches are bit-identical."""
    H, n, count = 24, 4352, 672
    q, k, v = _qkv(1, H, n, n, seed=41)
    G = math.ceil(n / 192)
    inds, counts = random_index_sets(1, H, G, n, count, n, seed=9)
    qd, kd, vd, indd, cntd = [t.to(dev) for t in (q, k, v, inds, counts)]
    base = randn_bf16(1, H, n, 128, seed=5).to(dev)
    slots = 2 * torch.cuda.get_device_properties(0).multi_processor_count
    rem = (H * G) % slots if H * G > slots else 0
    if not (0 < rem and 6 * rem <= slots):
        pytest.skip(f"{H * G} items on {slots} slots: no row-split tail at this CU count")
    with row_split(2):
        plain = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    out = torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1)
    first = (H * G - rem) // G, (H * G - rem) % G          # (head, group) of the first row-split item
    assert torch.equal(out[:, :first[0]], plain[:, :first[0]]), "items in front of the tail run the unsplit body: same bits"
    assert_close_bf16(out, plain, what="row-split tail vs unsplit launch")
    for item in (H * G - rem, H * G - 1):
        h, g = divmod(item, G)
        rows = slice(g * 192, min(n, (g + 1) * 192))
        o_ref = base[:, h:h + 1, rows].cpu().clone()
        oracle.csp_attn(q[:, h:h + 1, rows].contiguous(), k[:, h:h + 1], v[:, h:h + 1], o_ref,
                        inds[:, h:h + 1, g:g + 1].contiguous(), counts[:, h:h + 1, g:g + 1].contiguous(), 1)
        assert_close_bf16(out[:, h:h + 1, rows], o_ref, atol=3e-2, what=f"row-split item (head {h}, group {g})")
    for _ in range(10):
        assert torch.equal(torch.ops.chipmunk.csp_attn_out(qd, kd, vd, base, indd, cntd, 1), out)
