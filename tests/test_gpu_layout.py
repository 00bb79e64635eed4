"""Two additions around the attention operators that change no value: outputs laid out token-major ([B, H, N, D] view of
[B, N, H, D] storage, ``attn.token_major_output``) and the unpacked (indices, counts) kept beside a resident bit-packed mask
(``attn.keep_unpacked_indices``).  Every comparison here is bit for bit."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _qkv(H, N, seed, dev):
    g = torch.Generator(device=dev).manual_seed(seed)
    return [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]


def _is_token_major(o):
    B, H, N, D = o.shape
    return o.stride() == (N * H * D, D, H * D, 1)


@pytest.mark.parametrize("H,N", [(3, 1000), (4, 20480), (2, 33000)])
def test_dense_operators_token_major(dev, H, N):
    """General kernel (small) and attn64.hip (long launches): same bits in either layout, and the model's
    `b h s d -> b s (h d)` of the token-major result is a view of it."""
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd import ops
    q, k, v = _qkv(H, N, 5, dev)
    o0, l0 = torch.ops.chipmunk.dense_attn(q, k, v)
    o1, l1 = torch.ops.chipmunk.dense_attn_layout(q, k, v, True)
    assert o0.is_contiguous() and _is_token_major(o1) and o1.shape == o0.shape
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    flat = o1.permute(0, 2, 1, 3).reshape(1, N, H * 128)
    assert flat.data_ptr() == o1.data_ptr() and flat.is_contiguous()
    c0 = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l0)
    c1 = torch.ops.chipmunk.dense_colsum_attn_layout(q, k, v, l0, True)
    assert _is_token_major(c1[0])
    for a, b in zip(c0, c1):
        assert torch.equal(a, b)
    G = (N + 191) // 192
    st = torch.rand(1, H, G, N, device=dev) < 0.01
    gr = torch.ones(1, H, G, 1, dtype=torch.bool, device=dev)
    ops.manual_seed(3)       # the 1 % random keys come from a counter-based hash: same seed, same launch order, same bits
    m0 = ops.dense_colsum_topk_mask(q, k, v, l0, 128 * max(1, N // 2560), 0.01, gr, st)
    ops.manual_seed(3)
    m1 = ops.dense_colsum_topk_mask(q, k, v, l0, 128 * max(1, N // 2560), 0.01, gr, st, True)
    assert _is_token_major(m1[0])
    for a, b in zip(m0, m1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("H,N,keep", [(2, 1344, 256), (4, 16512, 2048)])
def test_sparse_step_keeps_the_cache_layout(dev, H, N, keep):
    """csp_attn_out writes o_in +/- sparse attention in o_in's layout (general kernel; attn96.hip at the larger size) and
    csp_attn accumulates into a token-major tensor in place."""
    import chipmunk_amd  # noqa: F401
    q, k, v = _qkv(H, N, 6, dev)
    G = (N + 191) // 192
    g = torch.Generator(device=dev).manual_seed(8)
    inds = torch.rand(1, H, G, N, device=dev, generator=g).topk(keep, dim=-1).indices.sort(-1).values.to(torch.int32).contiguous()
    counts = torch.full((1, H, G), keep, dtype=torch.int32, device=dev)
    cache = torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g)
    cache_tm = torch.empty(1, N, H, 128, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    cache_tm.copy_(cache)
    for scale in (1, -1):
        a = torch.ops.chipmunk.csp_attn_out(q, k, v, cache, inds, counts, scale)
        b = torch.ops.chipmunk.csp_attn_out(q, k, v, cache_tm, inds, counts, scale)
        assert a.is_contiguous() and _is_token_major(b)
        assert torch.equal(a, b)
        assert torch.equal(cache, cache_tm), "o_in is only read"
    acc, acc_tm = cache.clone(), cache_tm.clone()
    assert _is_token_major(acc_tm)
    torch.ops.chipmunk.csp_attn(q, k, v, acc, inds, counts, 1)
    torch.ops.chipmunk.csp_attn(q, k, v, acc_tm, inds, counts, 1)
    assert torch.equal(acc, acc_tm)


@pytest.mark.parametrize("H,N,keep", [(2, 1344, 256), (4, 16512, 2048)])
def test_ragged_index_rows(dev, H, N, keep):
    """compact_indices + csp_attn_out_ragged against the padded form: counts from 0 to every key (the text groups of the video
    models keep them all), rows of any width back to back; general kernel and attn96.hip."""
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd import ops
    q, k, v = _qkv(H, N, 16, dev)
    G = (N + 191) // 192
    g = torch.Generator(device=dev).manual_seed(18)
    inds = torch.rand(1, H, G, N, device=dev, generator=g).argsort(dim=-1).to(torch.int32).contiguous()
    counts = (torch.randint(keep // 64, keep // 32 + 1, (1, H, G), device=dev, generator=g) * 32).to(torch.int32)
    counts[0, :, -1] = N                      # a group that keeps everything
    counts[0, 0, 0] = 0                       # and one that keeps nothing
    counts[0, -1, 1] = 40                     # not a multiple of the key tile
    valid = torch.arange(N, device=dev)[None, None, None, :] < counts[..., None]
    inds = torch.where(valid, inds, torch.full_like(inds, -1))
    flat, offsets = ops.compact_indices(inds, counts)
    off = offsets.cpu()
    assert off[0] == 0 and bool(((off[1:] - off[:-1]) % 32 == 0).all()) and flat.numel() == int(off[-1]) + 64
    assert bool(((off[1:] - off[:-1]) >= counts.flatten().cpu()).all())
    r = 1 * G + 3                              # head 1, group 3
    c = int(counts.flatten()[r])
    assert torch.equal(flat[int(off[r]):int(off[r]) + c], inds.view(-1, N)[r, :c]) and int(flat[int(off[r]) + c:int(off[r + 1])].abs().sum()) == 0
    cache = torch.randn(1, N, H, 128, device=dev, dtype=torch.bfloat16, generator=g).permute(0, 2, 1, 3)
    for scale in (1, -1):
        a = ops.csp_attn_out(q, k, v, cache, inds, counts, scale)
        b = ops.csp_attn_out_ragged(q, k, v, cache, flat, offsets, counts, scale)
        assert _is_token_major(b) and torch.equal(a, b)


def _run_hunyuan_schedule(dev, token_major, keep_unpacked, resident, steps=13, keep_offloaded=True):
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util import config as cfgmod
    from chipmunk_amd.util import layer_counter as lc
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.util.storage import offloaded_tensor as ot
    cfgmod.reset_to_base()
    lc.singleton.__init__(0, 0)
    cfgmod.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
    cfg = cfgmod.GLOBAL_CONFIG
    cfg["steps"] = 50
    cfg["step_caching"]["is_enabled"] = False
    cfg["attn"]["token_major_output"] = token_major
    cfg["attn"]["keep_unpacked_indices"] = keep_unpacked
    cfg["attn"]["keep_unpacked_indices_offloaded"] = keep_offloaded
    cfg["offloading"]["keep_resident_if_fits"] = resident
    ot.gpu_tensors.clear()
    chipmunk_amd.ops.manual_seed(9)     # the 1 % random keys of the mask step (counter-based hash)
    torch.manual_seed(9)                # ... and of the static mask (torch's generator, reference ops/voxel.py)
    booked = ot._resident_bytes
    L, H, vid, txt = 5, 2, (4, 12, 16), 64
    N = vid[0] * vid[1] * vid[2] + txt
    g = torch.Generator(device=dev).manual_seed(21)
    q0, k0, v0, dq = [torch.randn(1, H, N, 128, device=dev, generator=g) for _ in range(4)]
    layers = []
    for _ in range(L):
        num, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
        layers.append(SparseDiffAttn(num, counter))
    layers[0].initialize_static_mask(vid, txt, H, dev)
    outs, kept = [], 0
    with torch.no_grad():
        for step in range(steps):
            q = (q0 + 0.03 * step * dq).to(torch.bfloat16)
            k, v = k0.to(torch.bfloat16), v0.to(torch.bfloat16)
            for li, layer in enumerate(layers):
                if step > 0 or li > 0:
                    layer.storage.load_async_wait()
                layers[(li + 1) % L].storage.load_async()
                o = layer(q, k, v)
                assert _is_token_major(o) == token_major, (step, li)
                outs.append(o.contiguous().clone())
                layer.storage.complete_cur_layer()
        kept = sum(1 for layer in layers if layer._unpacked[0] is not None)
        width = [(u[0].numel(), int(((u[2].flatten().long() + 31) // 32 * 32).sum())) for u in (layer._unpacked[0] for layer in layers) if u is not None]
    torch.cuda.synchronize()
    booked = ot._resident_bytes - booked
    cfgmod.reset_to_base()
    lc.singleton.__init__(0, 0)
    return outs, kept, width, N, booked


def test_layout_and_kept_indices_change_nothing_over_a_schedule(dev):
    """13 inference steps x 5 layers (2 dense) of the HunyuanVideo configuration -- full steps 0, 1 (mask) and 10 (mask recompute),
    sparse steps between -- in four set-ups: the reference's sequence (contiguous outputs, indices unpacked from the bits every
    step), both additions on with the caches resident, and both with the caches going through pinned host memory (where the
    unpacked indices are NOT kept and a token-major cache must come back token-major)."""
    ref, kept, _, N, _ = _run_hunyuan_schedule(dev, False, False, True)
    assert kept == 0
    new, kept, width, N, booked = _run_hunyuan_schedule(dev, True, True, True)
    assert kept == 3 and all(n == need + 64 for n, need in width), "three sparse layers keep ragged index rows: the kept keys, no more"
    assert booked > 4 * sum(n for n, _ in width), "the kept rows are booked against the HBM budget"
    off, kept_off, _, _, _ = _run_hunyuan_schedule(dev, True, True, False, keep_offloaded=False)
    assert kept_off == 0, "attn.keep_unpacked_indices_offloaded off: a mask that travels to the host is unpacked where it lands"
    off_kept, kept_off2, _, _, _ = _run_hunyuan_schedule(dev, True, True, False)
    assert kept_off2 == 3, "... on (the default): the index rows stay in HBM while the mask and the output cache go through the host"
    off_ref, _, _, _, _ = _run_hunyuan_schedule(dev, False, True, False)
    assert len(ref) == len(new) == len(off) == len(off_kept) == 65
    for i, (a, b, c, d, e) in enumerate(zip(ref, new, off, off_ref, off_kept)):
        assert torch.equal(a, b), f"resident, layer call {i}"
        assert torch.equal(a, c), f"offloaded, layer call {i}"
        assert torch.equal(a, d), f"offloaded contiguous, layer call {i}"
        assert torch.equal(a, e), f"offloaded, index rows kept, layer call {i}"


def test_suppressed_mask_load_is_fetched_on_demand(dev):
    """attn.keep_unpacked_indices_offloaded keeps the index rows in HBM and does not bring the offloaded mask back; if the kept rows cannot serve a
    sparse step after all (here: attn.fused_residual switched off in between), the module fetches the mask's host copy on the spot and the step
    gives the same bits as a run that never kept the rows."""
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd.modules import SparseDiffAttn
    from chipmunk_amd.util import config as cfgmod
    from chipmunk_amd.util import layer_counter as lc
    from chipmunk_amd.util.layer_counter import LayerCounter
    from chipmunk_amd.util.storage import offloaded_tensor as ot

    def run(keep_offloaded, switch_off_at):
        cfgmod.reset_to_base()
        lc.singleton.__init__(0, 0)
        cfgmod.load_from_file(os.path.join(ROOT, "configs", "hunyuan_c3.yml"))
        cfg = cfgmod.GLOBAL_CONFIG
        cfg["steps"] = 50
        cfg["step_caching"]["is_enabled"] = False
        cfg["attn"]["first_n_dense_layers"] = 0
        cfg["attn"]["keep_unpacked_indices_offloaded"] = keep_offloaded
        cfg["offloading"]["keep_resident_if_fits"] = False
        ot.gpu_tensors.clear()
        chipmunk_amd.ops.manual_seed(9)
        torch.manual_seed(9)
        H, vid, txt = 2, (4, 12, 16), 64
        N = vid[0] * vid[1] * vid[2] + txt
        g = torch.Generator(device=dev).manual_seed(21)
        q0, k0, v0, dq = [torch.randn(1, H, N, 128, device=dev, generator=g) for _ in range(4)]
        num, counter = LayerCounter.build_for_layer(is_attn_sparse=True)
        layer = SparseDiffAttn(num, counter)
        layer.initialize_static_mask(vid, txt, H, dev)
        outs, suppressed = [], []
        with torch.no_grad():
            for step in range(5):
                if step == switch_off_at:
                    cfg["attn"]["fused_residual"] = False
                if step > 0:
                    layer.storage.load_async()
                    layer.storage.load_async_wait()
                suppressed.append(bool(layer.storage.indices.suppress_load[0]))
                outs.append(layer((q0 + 0.03 * step * dq).to(torch.bfloat16), k0.to(torch.bfloat16), v0.to(torch.bfloat16)).contiguous().clone())
                layer.storage.complete_cur_layer()
        torch.cuda.synchronize()
        cfgmod.reset_to_base()
        lc.singleton.__init__(0, 0)
        return outs, suppressed

    ref, sup_ref = run(False, 3)
    got, sup = run(True, 3)
    assert not any(sup_ref) and sup[2] and sup[3], (sup_ref, sup)      # steps 2, 3 are sparse: the mask's load was suppressed ...
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"step {i}"                          # ... and step 3 (no fused residual any more) fetched it on demand
