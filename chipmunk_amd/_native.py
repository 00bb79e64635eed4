"""ctypes view of the C ABI (``include/chipmunk_hip.h``) -- used by the ABI tests and by code that wants to call the
kernels without going through ``torch.ops``.  Loading fails loudly: there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHIPMUNK_HIP_LIB: another build of the same library (tools/probes/mm1_forms/build.sh: the measured-and-not-shipped GEMM forms); set
# LD_LIBRARY_PATH to its directory as well so that the torch registry binds to the same file
LIB_PATH = os.environ.get("CHIPMUNK_HIP_LIB") or os.path.join(_HERE, "lib", "libchipmunk_hip.so")

# every symbol include/chipmunk_hip.h declares (tests/test_abi.py checks this list against the header)
SYMBOLS = [
    "chipmunk_last_error", "chipmunk_abi_version", "chipmunk_set_option", "chipmunk_set_random_seed",
    "chipmunk_csp_attn", "chipmunk_csp_attn_out", "chipmunk_csp_128_attn", "chipmunk_dense_attn", "chipmunk_dense_colsum_attn",
    "chipmunk_csp_mlp_mm1", "chipmunk_csp_mlp_mm1_scatter", "chipmunk_csp_mlp_mm1_fp8", "chipmunk_csp_mlp_mm2_and_scatter_add", "chipmunk_csp_scatter_add", "chipmunk_csp_mlp_mm2",
    "chipmunk_topk_indices", "chipmunk_topk_delta_indices", "chipmunk_topk_mask", "chipmunk_mask_to_indices", "chipmunk_mask_to_sorted_indices", "chipmunk_packed_mask_to_indices", "chipmunk_copy_indices",
    "chipmunk_bitpack", "chipmunk_bitunpack", "chipmunk_transpose16", "chipmunk_block_mean", "chipmunk_quantize_fp8", "chipmunk_gather_rows", "chipmunk_qkv_split_norm", "chipmunk_dense_colsum_topk_mask", "chipmunk_dense_attn_strided", "chipmunk_csp_attn_out_ragged", "chipmunk_compact_indices", "chipmunk_residual_ln_modulate", "chipmunk_dense_colsum_attn_strided", "chipmunk_dense_colsum_topk_mask_strided", "chipmunk_release_scratch", "chipmunk_big_scratch_fallbacks",
    "chipmunk_host_alloc", "chipmunk_host_free", "chipmunk_copy_d2h_async", "chipmunk_copy_h2d_async", "chipmunk_host_bytes",
]

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m chipmunk_amd.build` (hipcc --offload-arch=gfx950). "
                "chipmunk_amd has no CPU fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.chipmunk_last_error.restype = ctypes.c_char_p
        _lib.chipmunk_abi_version.restype = ctypes.c_int
        # A/B experiments without code edits: CHIPMUNK_AMD_OPTIONS="mm2_variant=12,mm1_nr=8" (tuning knobs only;
        # unknown names raise)
        for item in filter(None, os.environ.get("CHIPMUNK_AMD_OPTIONS", "").split(",")):
            name, _, val = item.partition("=")
            if _lib.chipmunk_set_option(name.strip().encode(), int(val)) != 0:
                raise ValueError(f"CHIPMUNK_AMD_OPTIONS: {_lib.chipmunk_last_error().decode()}")
    return _lib


def last_error() -> str:
    return lib().chipmunk_last_error().decode()


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what}: {last_error()}")


def set_option(name: str, value: int) -> None:
    check(lib().chipmunk_set_option(name.encode(), int(value)), "chipmunk_set_option")


def manual_seed(seed: int) -> None:
    """Restart the random-key sequence of topk_indices / topk_mask (`random_amount` > 0): same seed + same launch order
    = same random columns.  Without a call the sequence starts from a fixed default seed."""
    check(lib().chipmunk_set_random_seed(ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF)), "chipmunk_set_random_seed")


class HostBuffer:
    """Page-locked host memory from the library's pool (``chipmunk_host_alloc`` = hipHostMalloc): the pinned side of the cache offload
    (reference ``util/storage/offloaded_tensor.py:42-44,71`` keeps ``torch.empty(..., pin_memory=True)`` tensors).  Exposes the few
    tensor-like accessors the storage code and the reports use."""

    def __init__(self, numel: int, dtype) -> None:
        import torch
        self.dtype = dtype
        self._numel = int(numel)
        self._itemsize = torch.empty(0, dtype=dtype).element_size()
        self.nbytes = self._numel * self._itemsize
        p = ctypes.c_void_p()
        check(lib().chipmunk_host_alloc(ctypes.c_size_t(max(self.nbytes, 1)), ctypes.byref(p)), "chipmunk_host_alloc")
        self.ptr = p.value

    def numel(self) -> int:
        return self._numel

    def element_size(self) -> int:
        return self._itemsize

    def is_pinned(self) -> bool:
        return True

    def free(self) -> None:
        if getattr(self, "ptr", None):
            lib().chipmunk_host_free(ctypes.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass


def copy_d2h_async(host_ptr: int, dev_ptr: int, nbytes: int, stream: int) -> None:
    check(lib().chipmunk_copy_d2h_async(ctypes.c_void_p(host_ptr), ctypes.c_void_p(dev_ptr), ctypes.c_size_t(nbytes), ctypes.c_void_p(stream)),
          "chipmunk_copy_d2h_async")


def copy_h2d_async(dev_ptr: int, host_ptr: int, nbytes: int, stream: int) -> None:
    check(lib().chipmunk_copy_h2d_async(ctypes.c_void_p(dev_ptr), ctypes.c_void_p(host_ptr), ctypes.c_size_t(nbytes), ctypes.c_void_p(stream)),
          "chipmunk_copy_h2d_async")


def host_bytes() -> int:
    f = lib().chipmunk_host_bytes
    f.restype = ctypes.c_size_t
    return int(f())
