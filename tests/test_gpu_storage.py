"""Pinned-host offload pool on the GPU: D2H on the offload stream, H2D prefetch into the 2-slot device pipeline, the
compute stream waiting on both (reference util/storage/offloaded_tensor.py:91-178), plus the residency policy."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def cfg(fresh_config):
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    from chipmunk_amd.util.storage import offloaded_tensor as ot
    ot.gpu_tensors.clear()
    ot._resident_bytes = 0
    return fresh_config


def test_offload_roundtrip_through_pinned_host(cfg):
    from chipmunk_amd.util.storage import AttnStorage
    from chipmunk_amd.util.storage import offloaded_tensor as ot
    cfg["offloading"]["global_disable_offloading"] = False
    cfg["offloading"]["attn.out_cache"] = True
    dev = torch.device("cuda:0")
    layers = [AttnStorage(i, init_names=["out_cache"]) for i in range(4)]
    data = [torch.randn(1, 4, 384, 128, device=dev).to(torch.bfloat16) for _ in range(4)]
    for st, t in zip(layers, data):
        assert st.out_cache.is_offload_enabled
        st.set_out_cache(t.clone())
    for st in layers:
        assert st.out_cache.cpu_buf[0].is_pinned() and st.out_cache.cpu_buf[0].numel() == data[0].numel()
    # the model loop: wait(this layer), prefetch(next layer)  (reference hunyuan/models.py:796-801)
    layers[0].load_async()
    for i, st in enumerate(layers):
        st.load_async_wait()
        if i + 1 < len(layers):
            layers[i + 1].load_async()
        got = st.get_out_cache()
        assert got.device.type == "cuda" and torch.equal(got, data[i])
        assert got.data_ptr() == ot.gpu_tensors["attn.out_cache"][i % ot.PIPELINE_DEPTH].data_ptr()
    # only PIPELINE_DEPTH device slots exist for the 4 layers
    assert len({t.data_ptr() for t in ot.gpu_tensors["attn.out_cache"]}) == ot.PIPELINE_DEPTH


def test_offload_cur_value_and_shape_change(cfg):
    from chipmunk_amd.util.storage import MaybeOffloadedTensor
    cfg["offloading"]["global_disable_offloading"] = False
    cfg["offloading"]["mlp.out_cache"] = True
    dev = torch.device("cuda:0")
    t = MaybeOffloadedTensor("mlp.out_cache", 0, torch.bfloat16, dev)
    a = torch.randn(1, 3840, 64, device=dev).to(torch.bfloat16)
    t.offload(a)
    t.load_async(); t.load_async_wait()
    slot = t.get_loaded_value()
    slot += 1                       # in-place update of the device slot ...
    t.offload_cur_value()           # ... written back to the host copy
    b = torch.randn(1, 4352, 64, device=dev).to(torch.bfloat16)   # double -> single block shape switch
    t2 = MaybeOffloadedTensor("mlp.out_cache", 2, torch.bfloat16, dev)
    t2.offload(b)
    t2.load_async(); t2.load_async_wait()
    assert torch.equal(t2.get_loaded_value(), b)
    t.load_async(); t.load_async_wait()
    assert torch.equal(t.get_loaded_value(), a + 1)


def test_disabled_offload_keeps_device_tensor(cfg):
    from chipmunk_amd.util.storage import MlpStorage
    cfg["offloading"]["global_disable_offloading"] = True
    st = MlpStorage(0)
    x = torch.randn(8, 8, device="cuda:0")
    st.set_indices(x)
    assert st.get_indices() is x
    st.load_async(); st.load_async_wait()
    assert st.get_counts() is None


def test_keep_resident_if_fits_policy(cfg):
    from chipmunk_amd.util.storage import MaybeOffloadedTensor
    cfg["offloading"].update({"global_disable_offloading": False, "attn.out_cache": True,
                              "keep_resident_if_fits": True, "hbm_budget_gb": 0.001})  # ~1 MB budget
    dev = torch.device("cuda:0")
    small = MaybeOffloadedTensor("attn.out_cache", 0, torch.bfloat16, dev)
    big = MaybeOffloadedTensor("attn.out_cache", 1, torch.bfloat16, dev)
    a = torch.randn(256, 128, device=dev).to(torch.bfloat16)          # 64 KB: stays in HBM
    b = torch.randn(4096, 1024, device=dev).to(torch.bfloat16)        # 8 MB: over budget -> host
    small.offload(a)
    big.offload(b)
    assert small.get_loaded_value() is a and small.cpu_buf[0] is None
    assert big.cpu_buf[0] is not None and big.cpu_buf[0].is_pinned()
    big.load_async(); big.load_async_wait()
    assert torch.equal(big.get_loaded_value(), b)


def test_chunk_modules_of_one_layer_have_their_own_load_slots(cfg):
    """a sequence-parallel rank builds several SparseDiffAttn modules with the SAME layer number (head
    chunks, distributed.chunk_counters).  With the caches on the host each chunk's load must land in its own device slot;
    with one shared slot every chunk read the LAST chunk's cache."""
    from chipmunk_amd.util.storage import AttnStorage
    from chipmunk_amd.util.storage import offloaded_tensor as ot
    cfg["offloading"]["global_disable_offloading"] = False
    cfg["offloading"]["attn.out_cache"] = True
    cfg["offloading"]["keep_resident_if_fits"] = False
    dev = torch.device("cuda:0")
    n_layers, n_chunks = 4, 3
    store = [[AttnStorage(l, init_names=["out_cache"], slot=c) for c in range(n_chunks)] for l in range(n_layers)]
    data = [[torch.randn(1, 1 + c, 192, 128, device=dev).to(torch.bfloat16) for c in range(n_chunks)] for _ in range(n_layers)]
    for l in range(n_layers):
        for c in range(n_chunks):
            store[l][c].set_out_cache(data[l][c].clone())
            assert not store[l][c].out_cache.is_resident()
    # the bench's loop: wait for this layer's chunks, prefetch the next layer's, then every chunk reads its cache
    for c in range(n_chunks):
        store[0][c].load_async()
    for l in range(n_layers):
        for c in range(n_chunks):
            store[l][c].load_async_wait()
        for c in range(n_chunks):
            store[(l + 1) % n_layers][c].load_async()
        got = [store[l][c].get_out_cache() for c in range(n_chunks)]
        torch.cuda.synchronize()
        for c in range(n_chunks):
            assert got[c].shape == data[l][c].shape and torch.equal(got[c], data[l][c]), (l, c)
        assert len({g.data_ptr() for g in got}) == n_chunks
    assert set(ot.gpu_tensors) == {"attn.out_cache", "attn.out_cache#1", "attn.out_cache#2"}


def test_two_compute_streams_each_see_their_loaded_values(cfg):
    """The offload -> load dependency is an event the LOAD stream waits on, not a process-wide flag one compute stream clears: two
    compute streams that each offload, update and reload their own tensors (large enough for the copies to be in flight when the
    other stream asks) both read back what they stored."""
    from chipmunk_amd.util.storage import MaybeOffloadedTensor
    cfg["offloading"]["global_disable_offloading"] = False
    cfg["offloading"]["mlp.out_cache"] = True
    cfg["offloading"]["attn.out_cache"] = True
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    names = ["mlp.out_cache", "attn.out_cache"]
    holders = [MaybeOffloadedTensor(n, 0, torch.bfloat16, dev) for n in names]
    base = [torch.randn(64 << 20, device=dev).to(torch.bfloat16) for _ in names]      # 128 MiB each
    torch.cuda.synchronize()
    for rnd in range(3):
        for s, h, b in zip(streams, holders, base):
            with torch.cuda.stream(s):
                h.offload(b + rnd)               # device-to-host on the offload stream, behind stream s
        for s, h, b in zip(reversed(streams), reversed(holders), reversed(base)):
            with torch.cuda.stream(s):
                h.load_async(); h.load_async_wait()
                got = h.get_loaded_value()
                assert torch.equal(got, b + rnd), f"round {rnd}: {h.name} read back stale data on its own compute stream"


def test_native_pinned_pool_is_the_default_and_torch_pinned_tensors_the_fallback(cfg):
    """Round 6: the pinned side of the offload is the LIBRARY's pool (chipmunk_host_alloc = hipHostMalloc, chipmunk_copy_*_async =
    hipMemcpyAsync on the side streams; north_star wording, reference offloaded_tensor.py:42-44,71,104-118).  Same round trip with the
    key on (default) and off (torch pinned tensors); a dense permuted (token-major) tensor travels as its storage either way."""
    from chipmunk_amd import _native
    from chipmunk_amd.util.storage import MaybeOffloadedTensor
    cfg["offloading"]["global_disable_offloading"] = False
    cfg["offloading"]["attn.out_cache"] = True
    dev = torch.device("cuda:0")
    base = torch.randn(1, 640, 6, 128, device=dev).to(torch.bfloat16)
    tm = base.permute(0, 2, 1, 3)                       # [B, H, N, D] view of [B, N, H, D] storage
    before = _native.host_bytes()
    for native in (True, False):
        cfg["offloading"]["native_host_pool"] = native
        t = MaybeOffloadedTensor("attn.out_cache", 0, torch.bfloat16, dev)
        t.offload(tm)
        buf = t.cpu_buf[0]
        assert isinstance(buf, _native.HostBuffer) == native and buf.is_pinned() and buf.numel() == tm.numel()
        t.load_async(); t.load_async_wait()
        got = t.get_loaded_value()
        assert got.stride() == tm.stride() and torch.equal(got, tm)
        if native:
            assert _native.host_bytes() == before + tm.numel() * 2
            t.cpu_buf[0] = None
            del buf
            import gc
            gc.collect()
            assert _native.host_bytes() == before, "a dropped HostBuffer gives its pages back"


def test_host_pool_c_abi_called_directly():
    """chipmunk_host_alloc / chipmunk_copy_d2h_async / chipmunk_copy_h2d_async / chipmunk_host_free through ctypes, no torch types in the
    signatures: device -> pinned host -> device round trip on a side stream, bit for bit; errors come back as return codes."""
    import ctypes
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd import _native
    lib = _native.lib()
    dev = torch.device("cuda:0")
    src = torch.randn(1 << 20, device=dev)
    dst = torch.zeros_like(src)
    nbytes = src.numel() * 4
    p = ctypes.c_void_p()
    assert lib.chipmunk_host_alloc(ctypes.c_size_t(nbytes), ctypes.byref(p)) == 0 and p.value
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    st = ctypes.c_void_p(s.cuda_stream)
    assert lib.chipmunk_copy_d2h_async(p, ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(nbytes), st) == 0
    assert lib.chipmunk_copy_h2d_async(ctypes.c_void_p(dst.data_ptr()), p, ctypes.c_size_t(nbytes), st) == 0
    s.synchronize()
    assert torch.equal(src, dst)
    host = (ctypes.c_float * 4).from_address(p.value)
    assert list(host) == src[:4].cpu().tolist()
    assert lib.chipmunk_host_free(p) == 0
    assert lib.chipmunk_host_free(ctypes.c_void_p(0x1000)) != 0 and b"not allocated" in lib.chipmunk_last_error()
    assert lib.chipmunk_host_alloc(ctypes.c_size_t(0), ctypes.byref(p)) != 0
