"""Token-reorder ops on the GPU (one gather kernel with a cached permutation, chipmunk_gather_rows) against the fixtures
produced by the reference's own patchify / voxel code (tests/golden/layout_ops.pt): bit-exact, every
dtype width, the HunyuanVideo grid at full size as a round trip, and the step-level callers' shapes."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "layout_ops.pt")


@pytest.fixture()
def dev(fresh_config):
    import chipmunk_amd  # noqa: F401
    return torch.device("cuda:0")


def test_patchify_family_matches_reference_fixtures(dev):
    from chipmunk_amd import ops
    gold = torch.load(GOLD, weights_only=False)
    for h, w in ((16, 16), (48, 80)):
        x = torch.arange(2 * h * w, dtype=torch.int32).view(2, h, w)
        y = ops.patchify(x.to(dev))
        assert torch.equal(y.cpu(), gold[f"patchify_{h}x{w}"])
        assert torch.equal(ops.unpatchify(y, x.shape).cpu(), x)
        for dt in (torch.bfloat16, torch.float32, torch.uint8):          # 2-, 4- and 1-byte elements
            xf = (x % 251).to(dt)
            assert torch.equal(ops.patchify(xf.to(dev)).cpu(), ops.patchify(xf))
    out = ops.patchify_rope((1, 256), gold["patchify_rope_in"].clone().to(dev), 16, 16)
    assert torch.equal(out.cpu(), gold["patchify_rope_out"])


def test_voxel_reorder_matches_reference_fixtures(dev):
    from chipmunk_amd.ops import voxel
    gold = torch.load(GOLD, weights_only=False)
    for shape, vox in (((4, 6, 9), (4, 4, 4)), ((33, 45, 10), (4, 6, 8)), ((5, 13, 17), (4, 6, 8))):
        t, h, w = shape
        x = torch.arange(t * h * w, dtype=torch.int32).view(1, 1, t, h, w, 1)
        y = voxel.voxel_chunk_no_padding(x.to(dev), vox)
        assert torch.equal(y.flatten().cpu(), gold[f"voxel_{t}x{h}x{w}_{vox[0]}{vox[1]}{vox[2]}"])
        assert torch.equal(voxel.reverse_voxel_chunk_no_padding(y, x.shape, vox).cpu(), x)
        # rows of a real hidden size, two batches x two "heads": same permutation applied to every row
        g = torch.Generator().manual_seed(t)
        xr = torch.randn(2, 2, t, h, w, 96, generator=g).to(torch.bfloat16)
        yr = voxel.voxel_chunk_no_padding(xr.to(dev), vox)
        assert torch.equal(yr.cpu(), voxel.voxel_chunk_no_padding(xr, vox))
        assert torch.equal(voxel.reverse_voxel_chunk_no_padding(yr, xr.shape, vox).cpu(), xr)


def test_hunyuan_full_size_voxel_round_trip_and_locality(dev):
    """BASELINE configs[2] grid 33 x 45 x 80 with (4, 6, 8) voxels, hidden 3072 (the model's voxel_in / voxel_out,
    reference examples/hunyuan/hyvideo/modules/models.py:675-702): round trip exact; every full voxel's 192 tokens are
    contiguous in the new order (that is what makes 192-query groups spatially local)."""
    from chipmunk_amd.ops import voxel
    t, h, w, d = 33, 45, 80, 3072
    x = torch.randn(1, 1, t, h, w, d, device=dev, dtype=torch.bfloat16)
    y = voxel.voxel_chunk_no_padding(x, (4, 6, 8))
    assert y.shape == (1, 1, t * h * w, d)
    assert torch.equal(voxel.reverse_voxel_chunk_no_padding(y, x.shape, (4, 6, 8)), x)
    idx = torch.arange(t * h * w, dtype=torch.int32, device=dev).view(1, 1, t, h, w, 1)
    order = voxel.voxel_chunk_no_padding(idx, (4, 6, 8)).flatten().cpu().long()
    first = order[:192]
    tt, hh, ww = first // (h * w), (first // w) % h, first % w
    assert tt.max() < 4 and hh.max() < 6 and ww.max() < 8
    assert torch.equal(order.sort().values, torch.arange(t * h * w))


def test_gather_rows_is_traceable(dev):
    """The op has a fake (meta) kernel, so torch.compile / fake-tensor tracing goes through it without a graph break."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        src = torch.empty(2, 3, 50, 64, dtype=torch.bfloat16, device="cuda")
        m = torch.empty(40, dtype=torch.int32, device="cuda")
        out = torch.ops.chipmunk.gather_rows(src, m)
        assert out.shape == (2, 3, 40, 64) and out.dtype == torch.bfloat16


def test_ops_trace_under_dynamo_without_graph_breaks(dev):
    """Reference ops have no fake kernels, so torch.compile graph-breaks at every one; here a sparse
    attention step traces as ONE graph (fullgraph=True; backend "eager": tracing and fake-tensor propagation only)."""
    import math
    from helpers import randn_bf16, random_index_sets
    H, n, count = 2, 768, 256
    q, k, v, base = [randn_bf16(1, H, n, 128, seed=s).to(dev) for s in (1, 2, 3, 4)]
    inds, counts = [t.to(dev) for t in random_index_sets(1, H, math.ceil(n / 192), n, count, n, seed=5)]

    def step(q, k, v, base, inds, counts):
        o = torch.ops.chipmunk.csp_attn_out(q, k, v, base, inds, counts, 1)
        d, l = torch.ops.chipmunk.dense_attn(q, k, v)
        return o + d, l

    eager = step(q, k, v, base, inds, counts)
    compiled = torch.compile(step, backend="eager", fullgraph=True)(q, k, v, base, inds, counts)
    assert torch.equal(eager[0], compiled[0]) and torch.equal(eager[1], compiled[1])


@pytest.mark.parametrize("n,heads,extra,weights,rope", [(1000, 24, 0, True, 0), (4352, 24, 12288, True, 4096), (37, 3, 64, False, 37), (1, 1, 0, True, 0),
                                                        (1200, 4, 0, True, 1000)])
def test_qkv_split_norm_matches_the_reference_sequence(n, heads, extra, weights, rope):
    """chipmunk.qkv_split_norm = the caller's rearrange("B L (K H D) -> K B L H D") + RMSNorm(head_dim) on q and k
    (reference hyvideo/modules/models.py:188-193, norm_layers.py:43-58) + the transposes to [B, H, L, D]; the input may be the
    front part of a wider projection (single-stream blocks: linear1's 3*hidden + mlp columns).  v is a pure copy: bit-exact;
    q, k: the sum of squares is added in another order than torch's, so a normalised value can land one bf16 step away, and the
    weight product can round that into a second step (seen: 1 element in 3 million)."""
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.ops.qkv import qkv_split_norm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n)
    full = (torch.randn(n, 3 * heads * 128 + extra, generator=g) * 1.7).to(torch.bfloat16)
    qw = (1 + 0.1 * torch.randn(128, generator=g)).to(torch.bfloat16) if weights else None
    kw = (1 + 0.1 * torch.randn(128, generator=g)).to(torch.bfloat16) if weights else None
    fc = fs = None
    if rope:   # rotary embedding of the first `rope` tokens (the image tokens), (cos, sin) form of posemb_layers.py:133-172
        ang = torch.rand(rope, 64, generator=g) * 6.28
        fc, fs = ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1)
    ref = qkv_split_norm(full, qw, kw, heads, 1e-6, fc, fs)        # CPU: the reference's op sequence
    got = qkv_split_norm(full.to(dev), None if qw is None else qw.to(dev), None if kw is None else kw.to(dev), heads, 1e-6,
                         None if fc is None else fc.to(dev), None if fs is None else fs.to(dev))
    torch.cuda.synchronize()
    for name, r, o in zip("qkv", ref, got):
        assert o.shape == (1, heads, n, 128) and o.is_contiguous()
        if name == "v":
            assert torch.equal(o.cpu(), r)
        else:
            torch.testing.assert_close(o.cpu().float(), r.float(), rtol=1.6e-2, atol=2e-2 if rope else 1e-6)   # two bf16 steps (rotated
            #                                     values are sums of two products: a step of the larger product can exceed 1.6 % of a small sum)
            assert (o.cpu() == r).float().mean() > 0.99


@pytest.mark.parametrize("rows,cols,residual", [(1000, 3072, True), (777, 3072, False), (333, 1536, True), (65, 1024, True), (5, 8, False),
                                                (40, 5120, True), (3, 8192, True)])
def test_residual_ln_modulate_matches_the_reference_sequence(rows, cols, residual):
    """chipmunk.residual_ln_modulate = the block's torch sequence x = addcmul(x, gate, y); modulate(LayerNorm(x)) (reference
    hyvideo/modules/models.py:184-186, 262-275) in one pass.  The residual is bit-exact (one fp32 fma, one rounding, as addcmul); the
    normalised value is a fp32 quotient whose statistics are summed in another order than torch's, so it can land one bf16 step
    away (rarely), which the modulation carries through: compared against the fp32 evaluation of the same formula with a two-step
    tolerance, and element-for-element with torch's bf16 sequence on > 99 % of the entries."""
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.ops.qkv import residual_ln_modulate
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2.0 + 0.3).to(torch.bfloat16)
    y = torch.randn(rows, cols, generator=g).to(torch.bfloat16) if residual else None
    gate = (0.5 * torch.randn(cols, generator=g)).to(torch.bfloat16) if residual else None
    shift = (0.2 * torch.randn(cols, generator=g)).to(torch.bfloat16)
    scale = (0.2 * torch.randn(cols, generator=g)).to(torch.bfloat16)
    rx, rxm = residual_ln_modulate(x, y, gate, shift, scale, 1e-6)                      # CPU: the reference's op sequence
    to = lambda t: None if t is None else t.to(dev)
    ox, oxm = residual_ln_modulate(to(x), to(y), to(gate), to(shift), to(scale), 1e-6)
    torch.cuda.synchronize()
    assert torch.equal(ox.cpu(), rx), "x + gate * y, rounded once"
    xf = rx.float()
    xn = ((xf - xf.mean(-1, keepdim=True)) * torch.rsqrt(xf.var(-1, unbiased=False, keepdim=True) + 1e-6)).to(torch.bfloat16).float()
    want = shift.float() + xn * (1 + scale).float()           # (1 + scale is a bf16 tensor in the reference)
    torch.testing.assert_close(oxm.cpu().float(), want, rtol=1.6e-2, atol=2e-2)
    assert (oxm.cpu() == rxm).float().mean() > 0.99
    if not residual:
        assert ox.data_ptr() == to(x).data_ptr() or torch.equal(ox.cpu(), x)
