"""CPU ORACLE -- test infrastructure only.

ctypes front-end for ``oracle/liboracle.so`` (built from ``oracle/chipmunk_oracle.c`` by ``oracle/Makefile``),
a plain-C restatement of the reference's hot-path algorithms with the reference file:line cited per function in
the C source.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; ``chipmunk_amd`` (the product) never does.

Function names and argument order mirror ``torch.ops.chipmunk.*`` (reference ``csrc/chipmunk.cpp:45-60``) so the
parity tests read like the reference's own tests.  All arguments are CPU torch tensors.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a few seconds)."""
    src = os.path.join(_HERE, "chipmunk_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == "cpu"
    return ctypes.c_void_p(t.data_ptr())


def _strides3(t: torch.Tensor):
    assert t.dim() == 4 and t.stride(3) == 1 and t.dtype == torch.bfloat16
    return (ctypes.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2))


def _i32(t: torch.Tensor) -> torch.Tensor:
    assert t.dtype == torch.int32
    return t.contiguous()


# --------------------------------------------------------------------------- attention
def csp_attn(q, k, v, o, indices, indices_counts, o_scale: int, kv_tile: int = 112) -> None:
    """In place ``o += o_scale * attn(q, K[idx], V[idx])`` (reference csrc/attn/csp_attn.cu:315-423)."""
    assert o_scale in (1, -1)
    B, H, Nq, D = q.shape
    assert D == 128 and o.dtype == torch.bfloat16
    indices, indices_counts = _i32(indices), _i32(indices_counts)
    lib().oracle_csp_attn(_p(q), _p(k), _p(v), _p(o), _strides3(q), _strides3(k), _strides3(v), _strides3(o),
                          _p(indices), _p(indices_counts), B, H, Nq, k.shape[2], indices.shape[-1], kv_tile, 1,
                          int(o_scale))


def csp_128_attn(q, k, v, indices, indices_counts, kv_tile: int = 128) -> torch.Tensor:
    """Out of place column-sparse attention (reference csrc/attn/csp_128_attn.cu:355-461)."""
    B, H, Nq, D = q.shape
    assert D == 128
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    indices, indices_counts = _i32(indices), _i32(indices_counts)
    o = torch.zeros_like(q)
    lib().oracle_csp_attn(_p(q), _p(k), _p(v), _p(o), _strides3(q), _strides3(k), _strides3(v), _strides3(o),
                          _p(indices), _p(indices_counts), B, H, Nq, k.shape[2], indices.shape[-1], kv_tile, 0, 1)
    return o


def dense_attn(q, k, v) -> List[torch.Tensor]:
    """``[o, l]`` with ``l = 1 / sum_j exp(q_i.k_j / sqrt(D))`` fp32 ``[B,H,Nq,1]`` (reference dense_attn.cu:225-233)."""
    B, H, Nq, D = q.shape
    assert D == 128
    o = torch.empty((B, H, Nq, D), dtype=torch.bfloat16)
    l = torch.empty((B, H, Nq, 1), dtype=torch.float32)
    lib().oracle_dense_attn(_p(q), _p(k), _p(v), _p(o), _p(l), _strides3(q), _strides3(k), _strides3(v), B, H, Nq,
                            k.shape[2], _p(None), _p(None), 0)
    return [o, l]


def dense_colsum_attn(q, k, v, p) -> List[torch.Tensor]:
    """``[o, cs, l]``; ``cs`` bf16 ``[B,H,ceil(Nq/192),Nq]`` (reference dense_colsum_attn.cu:267-277,580-583)."""
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    assert D == 128 and p.dtype == torch.float32
    p = p.contiguous()
    G = (Nq + 191) // 192
    assert Nk <= Nq, "cs has Nq columns in the reference (dense_colsum_attn.cu:580-583)"
    cs_stride = Nq
    o = torch.empty((B, H, Nq, D), dtype=torch.bfloat16)
    l = torch.empty((B, H, Nq, 1), dtype=torch.float32)
    cs = torch.zeros((B, H, G, cs_stride), dtype=torch.bfloat16)
    lib().oracle_dense_attn(_p(q), _p(k), _p(v), _p(o), _p(l), _strides3(q), _strides3(k), _strides3(v), B, H, Nq, Nk,
                            _p(p), _p(cs), cs_stride)
    return [o, cs, l]


# --------------------------------------------------------------------------- MLP
def csp_mlp_mm1(a, b_colmajor, c, bias, pa_cache_colmajor, indices, indices_counts) -> None:
    M, K = a.shape
    F = b_colmajor.shape[0]
    for t in (a, b_colmajor, c, bias, pa_cache_colmajor):
        assert t.dtype == torch.bfloat16 and t.is_contiguous()
    lib().oracle_csp_mlp_mm1(_p(a), _p(b_colmajor), _p(c), _p(bias), _p(pa_cache_colmajor), _p(_i32(indices)),
                             _p(_i32(indices_counts)), M, K, F)


def csp_mlp_mm1_fp8(a, b, c, bias, pa_cache_colmajor, indices, indices_counts, scale_a, scale_b,
                    update_cache: bool = False) -> None:
    """fp8 e4m3fn GEMM1 (reference triton/csp_mlp_mm1.py:37-164); a [M,K], b [F,K] float8_e4m3fn tensors."""
    M, K = a.shape
    F = b.shape[0]
    assert a.dtype == torch.float8_e4m3fn and b.dtype == torch.float8_e4m3fn
    a8, b8 = a.contiguous().view(torch.uint8), b.contiguous().view(torch.uint8)
    lib().oracle_csp_mlp_mm1_fp8(_p(a8), _p(b8), _p(c), _p(bias), _p(pa_cache_colmajor), _p(_i32(indices)),
                                 _p(_i32(indices_counts)), ctypes.c_float(float(scale_a)), ctypes.c_float(float(scale_b)),
                                 M, K, F, int(bool(update_cache)))


def csp_scatter_add(packed, unpacked_colmajor, sp_inds, sp_counts, num_sms: int = 0) -> None:
    """Accepts the reference's ``[1,M,F]`` / ``[1,F,M]`` shapes (scatter_add.cu:58-59 hard-wires B=1) or 2-D."""
    packed2 = packed[0] if packed.dim() == 3 else packed
    unp2 = unpacked_colmajor[0] if unpacked_colmajor.dim() == 3 else unpacked_colmajor
    M, F = packed2.shape
    assert packed2.is_contiguous() and unp2.is_contiguous()
    lib().oracle_csp_scatter_add(_p(packed2), _p(unp2), _p(_i32(sp_inds)), _p(_i32(sp_counts)), M, F)


def csp_mlp_mm2(packed, w2t, out, indices, counts) -> None:
    packed2 = packed[0] if packed.dim() == 3 else packed
    w2 = w2t[0] if w2t.dim() == 3 else w2t
    out2 = out[0] if out.dim() == 3 else out
    M, F = packed2.shape
    N2 = w2.shape[1]
    assert packed2.is_contiguous() and w2.is_contiguous() and out2.is_contiguous()
    lib().oracle_csp_mlp_mm2(_p(packed2), _p(w2), _p(out2), _p(_i32(indices)), _p(_i32(counts)), M, F, N2)


def csp_mlp_mm2_and_scatter_add(packed, unpacked_colmajor, sp_inds, sp_counts, mma_a, mma_b, mma_c,
                                num_sms_scatter_add: int = 0, matmul_kernel: int = 0) -> None:
    """(i) scatter-add then (ii) GEMM2 ``mma_c += mma_a @ mma_b[idx]`` (csp_mlp_mm2_and_scatter_add.cu:96-259)."""
    csp_scatter_add(packed, unpacked_colmajor, sp_inds, sp_counts)
    csp_mlp_mm2(mma_a, mma_b, mma_c, sp_inds, sp_counts)


def dense_mlp(x, w1, b1, w2, b2) -> torch.Tensor:
    """The reference's dense eager path ``fc2(gelu_tanh(fc1(x)))`` (modules/mlp.py:33-34) -- cpu_baseline leg."""
    M, K = x.shape
    F = w1.shape[0]
    N2 = w2.shape[0]
    out = torch.empty((M, N2), dtype=torch.bfloat16)
    lib().oracle_dense_mlp(_p(x.contiguous()), _p(w1.contiguous()), _p(b1.contiguous()), _p(w2.contiguous()),
                           _p(b2.contiguous()), _p(out), M, K, F, N2)
    return out


# --------------------------------------------------------------------------- indexed IO
def topk_indices(activation, indices, counts, sparsity_amount: float, multiple_of: int, random_amount: float) -> None:
    assert random_amount == 0, "the reference's random keep is RNG-order dependent; parity is defined for 0 only"
    B, R, C = activation.shape
    assert C >= 1024 and indices.dtype == torch.int32 and counts.dtype == torch.int32
    assert indices.is_contiguous() and counts.is_contiguous()
    act = activation.to(torch.float32).contiguous()
    lib().oracle_topk_indices(_p(act), _p(indices), _p(counts), B * R, C, ctypes.c_double(sparsity_amount),
                              int(multiple_of))


def mask_to_indices(mask, multiple_of: int, pad_to_multiple_of: int) -> List[torch.Tensor]:
    assert mask.dim() == 4 and mask.dtype == torch.bool
    b, h, m, n = mask.shape
    pad_n = ((n + pad_to_multiple_of - 1) // pad_to_multiple_of) * pad_to_multiple_of
    mask = mask.contiguous()
    indices = torch.full((b, h, m, pad_n), -1, dtype=torch.int32)
    counts = torch.empty((b, h, m), dtype=torch.int32)
    lib().oracle_mask_to_indices(_p(mask.view(torch.uint8)), _p(indices), _p(counts), ctypes.c_int64(b * h * m), n,
                                 pad_n, int(multiple_of))
    return [indices, counts]


def copy_indices(bmfc1, bm_mid_cache, sp_inds, sp_counts) -> None:
    B, MR, F = bmfc1.shape
    M = sp_counts.shape[1]
    R = MR // M
    assert bmfc1.is_contiguous() and bm_mid_cache.is_contiguous() and bmfc1.dtype == bm_mid_cache.dtype
    lib().oracle_copy_indices(_p(bmfc1), _p(bm_mid_cache), _p(_i32(sp_inds)), _p(_i32(sp_counts)), B, M, R, F,
                              bmfc1.element_size())


def bitpack(mask) -> Tuple[torch.Tensor, torch.Size]:
    flat = mask.contiguous().view(torch.uint8).flatten()
    n = flat.numel()
    packed = torch.empty(((n + 7) // 8,), dtype=torch.uint8)
    lib().oracle_bitpack(_p(flat), _p(packed), ctypes.c_int64(n))
    return packed, mask.shape


def bitunpack(packed, original_shape) -> torch.Tensor:
    n = 1
    for d in original_shape:
        n *= d
    out = torch.empty((n,), dtype=torch.uint8)
    lib().oracle_bitunpack(_p(packed.contiguous()), _p(out), ctypes.c_int64(n))
    return out.view(torch.bool).view(*original_shape)
