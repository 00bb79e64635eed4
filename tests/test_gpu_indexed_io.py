"""GPU parity of the indexed-IO ops against the CPU oracle: integer work, bit-exact."""
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _rand_mask(shape, density, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) < density


def _check_m2i(dev, mask, multiple_of, pad_to):
    ref_i, ref_c = oracle.mask_to_indices(mask, multiple_of, pad_to)
    got_i, got_c = torch.ops.chipmunk.mask_to_indices(mask.to(dev), multiple_of, pad_to)
    assert got_i.shape == ref_i.shape and got_i.dtype == torch.int32
    assert torch.equal(got_c.cpu(), ref_c)
    n_written = ref_c.clamp(max=mask.shape[-1])  # a row can run out of False columns for padding
    live = torch.arange(ref_i.shape[-1])[None, None, None, :] < n_written[..., None]
    live &= ref_i >= 0
    assert torch.equal(got_i.cpu()[live], ref_i[live]), "index order must match the reference's lane-interleaved order"
    return got_i, got_c


@pytest.mark.parametrize("shape,density", [((1, 2, 3, 4352), 0.15), ((1, 3, 5, 4000), 0.06), ((2, 2, 4, 1000), 0.5),
                                           ((1, 1, 2, 77), 0.3), ((1, 2, 2, 7488), 0.07)])
def test_mask_to_indices(dev, shape, density):
    _check_m2i(dev, _rand_mask(shape, density, seed=shape[-1]), 128, 192)


def test_mask_to_indices_edge_rows(dev):
    n = 1536
    mask = torch.zeros(1, 1, 4, n, dtype=torch.bool)
    mask[0, 0, 1] = True                 # everything kept: no padding possible
    mask[0, 0, 2, ::3] = True            # count 512 -> already a multiple of 128
    mask[0, 0, 3, 5] = True              # one True -> 127 padding columns
    _check_m2i(dev, mask, 128, 192)
    _check_m2i(dev, mask, 112, 192)


def test_packed_mask_to_indices_equals_unpack_then_m2i(dev):
    import chipmunk_amd
    shape = (1, 3, 4, 4352)
    mask = _rand_mask(shape, 0.07, seed=3).to(dev)
    packed, shp = chipmunk_amd.ops.bitpack(mask)
    ref_p, _ = oracle.bitpack(mask.cpu())
    assert torch.equal(packed.cpu(), ref_p)
    assert torch.equal(chipmunk_amd.ops.bitunpack(packed, shp).cpu(), mask.cpu())
    i1, c1 = torch.ops.chipmunk.mask_to_indices(mask, 128, 192)
    i2, c2 = chipmunk_amd.ops.packed_mask_to_indices(packed, shp, 128, 192)
    assert torch.equal(c1, c2)
    live = torch.arange(i1.shape[-1], device=dev)[None, None, None, :] < c1[..., None]
    assert torch.equal(i1[live], i2[live])


def test_bitpack_unaligned_length(dev):
    import chipmunk_amd
    mask = _rand_mask((3, 37), 0.4, seed=9).to(dev)  # 111 bits: not a multiple of 8
    packed, shp = chipmunk_amd.ops.bitpack(mask)
    ref_p, _ = oracle.bitpack(mask.cpu())
    assert torch.equal(packed.cpu(), ref_p)
    assert torch.equal(chipmunk_amd.ops.bitunpack(packed, shp).cpu(), mask.cpu())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("sparsity,multiple_of", [(0.7, 256), (0.9, 112), (0.0, 256), (1.0, 256)])
def test_topk_indices(dev, dtype, sparsity, multiple_of):
    B, R, C = 1, 5, 12288  # FLUX: [1, 30, 12288]
    g = torch.Generator().manual_seed(17)
    act = torch.randn(B, R, C, generator=g).abs().to(dtype)
    ref_i = torch.full((B, R, C), -7, dtype=torch.int32)
    ref_c = torch.zeros(B, R, dtype=torch.int32)
    oracle.topk_indices(act, ref_i, ref_c, sparsity, multiple_of, 0.0)
    got_i = torch.full((B, R, C), -7, dtype=torch.int32, device=dev)
    got_c = torch.zeros(B, R, dtype=torch.int32, device=dev)
    torch.ops.chipmunk.topk_indices(act.to(dev), got_i, got_c, sparsity, multiple_of, 0.0)
    assert torch.equal(got_c.cpu(), ref_c)
    assert torch.equal(got_i.cpu(), ref_i), "kept set, canonical order and padding columns must be bit-exact"
    if 0 < sparsity < 1:  # the kept set is exactly {x >= threshold}; threshold from the first 1024 columns
        thr = act[0, 0, :1024].float().sort().values[int(1024 * torch.tensor(sparsity, dtype=torch.float32).item())]
        kept = set(torch.nonzero(act[0, 0].float() >= thr).flatten().tolist())
        assert kept.issubset(set(got_i[0, 0, :got_c[0, 0]].cpu().tolist()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_copy_indices(dev, dtype):
    B, M, R, F = 2, 3, 2, 512
    src = torch.randn(B, M * R, F).to(dtype)
    dst = torch.randn(B, M * R, F).to(dtype)
    counts = torch.tensor([[16, 100, 512], [0, 7, 256]], dtype=torch.int32)
    inds = torch.stack([torch.stack([torch.randperm(F, generator=torch.Generator().manual_seed(b * 10 + m)).int()
                                     for m in range(M)]) for b in range(B)])
    ref = dst.clone()
    oracle.copy_indices(src, ref, inds, counts)
    out = dst.clone().to(dev)
    torch.ops.chipmunk.copy_indices(src.to(dev), out, inds.to(dev), counts.to(dev))
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("kind", ["normal", "ties", "tiny", "zeros"])
def test_topk_delta_threshold_shortcut_is_bit_identical(dev, dtype, kind):
    """The fused kernel decides the low (always-zero) key bits of a 16-bit |delta| in one bisection round; the unfused
    topk_indices on the eager |delta| runs all 32.  Same threshold -> same indices, including ties at the threshold,
    subnormal deltas and an all-zero sample."""
    B, R, C = 1, 5, 4096
    g = torch.Generator().manual_seed(7)
    b = torch.randn(B, R, C, generator=g)
    if kind == "ties":
        d = torch.randint(0, 4, (B, R, C), generator=g).float() * 0.25
    elif kind == "tiny":
        d = torch.randn(B, R, C, generator=g) * 1e-7
    elif kind == "zeros":
        d = torch.zeros(B, R, C); d[..., 2048:] = torch.randn(B, R, 2048, generator=g)
    else:
        d = torch.randn(B, R, C, generator=g) * 0.3
    b = b.to(dtype).to(dev)
    cache0 = (b.float().cpu() + d).to(dtype).to(dev)
    mdiff = (b - cache0).abs()
    i1 = torch.full((B, R, C), -9, dtype=torch.int32, device=dev); c1 = torch.zeros(B, R, dtype=torch.int32, device=dev)
    i2 = torch.full_like(i1, -9); c2 = torch.zeros_like(c1)
    torch.ops.chipmunk.topk_indices(mdiff, i1, c1, 0.7, 64, 0.0)
    torch.ops.chipmunk.topk_delta_indices(b, cache0.clone(), i2, c2, 0.7, 64, 0.0)
    assert torch.equal(c1, c2) and torch.equal(i1, i2)
    ri = torch.full((B, R, C), -9, dtype=torch.int32); rc = torch.zeros(B, R, dtype=torch.int32)
    oracle.topk_indices(mdiff.cpu(), ri, rc, 0.7, 64, 0.0)
    assert torch.equal(rc, c2.cpu()) and torch.equal(ri, i2.cpu())


@pytest.mark.parametrize("sparsity,multiple_of,rk", [(0.7, 256, 0.0), (0.85, 112, 0.0), (0.7, 256, 0.05)])
def test_topk_delta_indices_equals_unfused_sequence(dev, sparsity, multiple_of, rk):
    """fused |b - cache| -> topk_indices -> copy_indices == the three separate ops (reference modules/mlp.py:70-85)."""
    B, R, C = 1, 6, 12288
    g = torch.Generator().manual_seed(23)
    b = torch.randn(B, R, C, generator=g).to(torch.bfloat16).to(dev)
    cache0 = (b.float().cpu() + 0.3 * torch.randn(B, R, C, generator=g)).to(torch.bfloat16).to(dev)
    # unfused
    mdiff = (b - cache0).abs()
    i1 = torch.full((B, R, C), -9, dtype=torch.int32, device=dev)
    c1 = torch.zeros(B, R, dtype=torch.int32, device=dev)
    cache1 = cache0.clone()
    import chipmunk_amd.ops as cm_ops
    cm_ops.manual_seed(99)      # random keys: every launch draws a fresh set; same seed + same launch order = same set
    torch.ops.chipmunk.topk_indices(mdiff, i1, c1, sparsity, multiple_of, rk)
    torch.ops.chipmunk.copy_indices(b, cache1, i1, c1)
    # fused
    i2 = torch.full((B, R, C), -9, dtype=torch.int32, device=dev)
    c2 = torch.zeros(B, R, dtype=torch.int32, device=dev)
    cache2 = cache0.clone()
    cm_ops.manual_seed(99)
    torch.ops.chipmunk.topk_delta_indices(b, cache2, i2, c2, sparsity, multiple_of, rk)
    assert torch.equal(c1, c2) and torch.equal(i1, i2)
    assert torch.equal(cache1.view(torch.int16), cache2.view(torch.int16))
    if rk > 0.0:    # a second launch draws other random columns (the random keys exist to refresh stale cache columns)
        i3, c3 = torch.full_like(i2, -9), torch.zeros_like(c2)
        torch.ops.chipmunk.topk_indices(mdiff, i3, c3, sparsity, multiple_of, rk)
        assert not torch.equal(i3, i1)
    if rk == 0.0:   # and against the oracle on the eager |delta|
        ri = torch.full((B, R, C), -9, dtype=torch.int32)
        rc = torch.zeros(B, R, dtype=torch.int32)
        oracle.topk_indices(mdiff.cpu(), ri, rc, sparsity, multiple_of, 0.0)
        assert torch.equal(i2.cpu(), ri) and torch.equal(c2.cpu(), rc)


@pytest.mark.parametrize("shape", [(1, 3840, 12288), (2, 100, 72), (1, 4352, 1024)])
def test_transpose_last2(dev, shape):
    x = torch.randn(*shape).to(torch.bfloat16).to(dev)
    assert torch.equal(torch.ops.chipmunk.transpose_last2(x), x.transpose(-1, -2).contiguous())


@pytest.mark.parametrize("shape,density", [((1, 2, 3, 4352), 0.15), ((1, 1, 4, 7488), 0.07), ((1, 1, 2, 200), 0.4)])
def test_mask_to_sorted_indices_same_set_ascending(dev, shape, density):
    import chipmunk_amd
    mask = _rand_mask(shape, density, seed=11).to(dev)
    ri, rc = torch.ops.chipmunk.mask_to_indices(mask, 128, 192)
    si, sc = chipmunk_amd.ops.mask_to_sorted_indices(mask, mask.shape, 128, 192)
    assert torch.equal(rc, sc) and si.shape == ri.shape
    pop = mask.sum(-1)
    for idx in [(0, 0, 0), (0, shape[1] - 1, shape[2] - 1)]:
        n_true, n_all = int(pop[idx]), int(min(sc[idx], shape[-1]))
        assert torch.equal(si[idx][:n_true], torch.nonzero(mask[idx]).flatten().to(torch.int32))   # ascending kept set
        assert torch.equal(si[idx][n_true:n_all], ri[idx][n_true:n_all])                            # same padding
    if shape[-1] % 8 == 0:
        packed, shp = chipmunk_amd.ops.bitpack(mask)
        pi, pc = chipmunk_amd.ops.mask_to_sorted_indices(packed, shp, 128, 192)
        live = torch.arange(si.shape[-1], device=dev)[None, None, None, :] < sc.clamp(max=shape[-1])[..., None]
        assert torch.equal(pc, sc) and torch.equal(pi[live], si[live])


def _distinct_bf16_rows(rows, n, seed):
    """rows x n bf16 values, all distinct within a row (positive normal bf16 bit patterns, permuted): no ties."""
    g = torch.Generator().manual_seed(seed)
    base = torch.arange(0x3000, 0x3000 + n, dtype=torch.int32)  # 0x3000.. : positive normals well below inf
    assert 0x3000 + n < 0x7f80
    out = torch.stack([base[torch.randperm(n, generator=g)] for _ in range(rows)])
    return out.to(torch.int16).view(torch.bfloat16)


@pytest.mark.parametrize("H,G,n,k", [(2, 3, 4352, 672), (1, 2, 20000, 1000), (1, 2, 10003, 517)])
def test_topk_mask_equals_reference_chain_without_ties(dev, H, G, n, k):
    """topk_mask with random_amount = 0 == the reference's `random_and_topk` chain (modules/attn.py:76-82) minus its
    randint: scatter_(topk) & groups | static -- exact on tie-free rows (aligned and unaligned row lengths)."""
    cs = _distinct_bf16_rows(H * G, n, seed=3).view(1, H, G, n).to(dev)
    g = torch.Generator().manual_seed(4)
    static = (torch.rand(1, H, G, n, generator=g) < 0.02).to(dev)
    groups = torch.tensor([True, False, True, True, False, True][:H * G]).view(1, H, G, 1).to(dev)
    ref = torch.zeros(1, H, G, n, dtype=torch.bool, device=dev)
    ref.scatter_(-1, cs.float().topk(k=k, dim=-1).indices, True)
    ref = (ref & groups) | static
    out = torch.ops.chipmunk.topk_mask(cs, k, 0.0, groups, static)
    assert out.dtype == torch.bool and out.shape == ref.shape
    assert torch.equal(out, ref)
    # no optional inputs
    out2 = torch.ops.chipmunk.topk_mask(cs, k, 0.0, None, None)
    ref2 = torch.zeros_like(ref).scatter_(-1, cs.float().topk(k=k, dim=-1).indices, True)
    assert torch.equal(out2, ref2)


def test_topk_mask_ties_and_random(dev):
    """With ties at the k-th value exactly k columns are taken, everything above the threshold is kept and nothing below;
    the random part adds about random_amount of the remaining columns, only in active groups."""
    H, G, n, k = 2, 2, 8192, 1000
    g = torch.Generator().manual_seed(5)
    cs = (torch.randint(0, 40, (1, H, G, n), generator=g).float() / 8).to(torch.bfloat16).to(dev)  # heavy ties
    out = torch.ops.chipmunk.topk_mask(cs, k, 0.0, None, None)
    assert (out.sum(-1) == k).all()
    v = cs.float()
    kth = v.topk(k, dim=-1).values[..., -1:]
    assert out[v > kth].all() and not out[v < kth].any()
    groups = torch.tensor([True, False, True, False]).view(1, H, G, 1).to(dev)
    out_r = torch.ops.chipmunk.topk_mask(cs, k, 0.01, groups, None)
    extra = out_r.sum(-1) - k * groups.squeeze(-1).long()
    assert (extra[~groups.squeeze(-1)] == 0).all()
    assert ((extra[groups.squeeze(-1)] > 30) & (extra[groups.squeeze(-1)] < 130)).all()  # ~1 % of 7192
    assert (out_r | ~out)[groups.expand_as(out)].all()  # the random part only adds


def test_c_abi_called_directly_without_the_torch_registry(dev):
    """The boundary is the C ABI: call two entry points through ctypes (device pointers + sizes + the raw stream handle)
    and compare with the oracle -- no torch types cross the call."""
    import ctypes
    from chipmunk_amd import _native
    lib = _native.lib()
    g = torch.Generator().manual_seed(21)
    mask = torch.rand(3, 700, generator=g) < 0.3
    rows, n, pad_n = 3, 700, 768
    md = mask.to(dev)
    inds = torch.full((rows, pad_n), -7, dtype=torch.int32, device=dev)
    counts = torch.zeros(rows, dtype=torch.int32, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.chipmunk_mask_to_indices(ctypes.c_void_p(md.data_ptr()), ctypes.c_void_p(inds.data_ptr()),
                                      ctypes.c_void_p(counts.data_ptr()), ctypes.c_int64(rows), ctypes.c_int(n),
                                      ctypes.c_int(pad_n), ctypes.c_int(128), stream)
    assert rc == 0, _native.last_error()
    ref_i, ref_c = oracle.mask_to_indices(mask.view(1, 1, rows, n), 128, 192)   # pads 700 -> 768 columns, like pad_n
    torch.cuda.synchronize()
    assert torch.equal(counts.cpu(), ref_c.view(-1))
    for r in range(rows):
        c = int(ref_c.view(-1)[r])
        assert torch.equal(inds[r, :c].cpu(), ref_i.view(rows, -1)[r, :c])
    packed = torch.zeros((rows * n + 7) // 8, dtype=torch.uint8, device=dev)
    rc = lib.chipmunk_bitpack(ctypes.c_void_p(md.data_ptr()), ctypes.c_void_p(packed.data_ptr()), ctypes.c_int64(rows * n), stream)
    assert rc == 0, _native.last_error()
    torch.cuda.synchronize()
    assert torch.equal(packed.cpu(), oracle.bitpack(mask)[0])
    # a bad argument comes back as an error code with a message, not an exception or an exit
    rc = lib.chipmunk_mask_to_indices(None, ctypes.c_void_p(inds.data_ptr()), ctypes.c_void_p(counts.data_ptr()),
                                      ctypes.c_int64(rows), ctypes.c_int(n), ctypes.c_int(pad_n), ctypes.c_int(128), stream)
    assert rc != 0 and _native.last_error()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,mbm", [((1, 4352, 3072), 128), ((2, 256, 1000), 128), ((1, 384, 8), 64), ((1, 3840, 1536), 128)])
def test_block_mean_kernel_vs_torch(dev, shape, mbm):
    """chipmunk.block_mean == x.reshape(b, n/mbm, mbm, c).mean(2) (reference modules/mlp.py:11-16): both are fp32 sums rounded once,
    in different summation orders -- equal within one bf16 unit in the last place, and exactly equal on integer-valued rows."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(shape, generator=g).to(torch.bfloat16).to(dev)
    got = torch.ops.chipmunk.block_mean(x, mbm)
    want = x.reshape(shape[0], shape[1] // mbm, mbm, shape[2]).float().mean(dim=2)
    assert got.shape == want.shape and got.dtype == torch.bfloat16
    err = (got.float() - want).abs()
    assert (err <= want.abs() * 2.0 ** -8 + 1e-6).all(), float(err.max())
    xi = torch.randint(-8, 9, shape, generator=g).to(torch.bfloat16).to(dev)     # exact in fp32 in any order
    assert torch.equal(torch.ops.chipmunk.block_mean(xi, mbm), xi.reshape(shape[0], shape[1] // mbm, mbm, shape[2]).mean(dim=2))
