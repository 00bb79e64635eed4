"""Sparse-MLP op wrappers (mirror of reference ``src/chipmunk/ops/mlp.py:7-92``).

``run_e2e`` (exported as ``chipmunk_amd.ops.mlp``) = packed GEMM1 with fused bias+GeLU+cache-subtract, then
scatter-add of the deltas into the column-major activation cache and GEMM2 accumulating into the output cache.
Both GEMMs are native HIP kernels here; the reference's GEMM2 is a Triton kernel whose CUfunction pointer is passed
through the op as an int (``ops/mlp.py:42``) -- the argument is kept (value 0) and ignored.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..util.config import GLOBAL_CONFIG, amd_key
from .indexed_io import scatter_add

USE_FUSED_MLP_MATMUL_2 = True
csp_mlp_mm2_function_ptr = 0  # placeholder for the reference's Triton kernel pointer (triton/csp_mlp_mm2.py:131-138)


def mm1(x: torch.Tensor, fc1w: torch.Tensor, sparse_act_packed: torch.Tensor, fc1b: torch.Tensor,
        sparse_act_T: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
        scale_a: Optional[torch.Tensor] = None, scale_b: Optional[torch.Tensor] = None) -> None:
    # (the reference asserts x is bf16 even on its fp8 branch, one of the reasons that branch cannot run as shipped)
    assert x.dtype == (torch.float8_e4m3fn if fc1w.dtype == torch.float8_e4m3fn else torch.bfloat16)
    assert sparse_act_packed.dtype == torch.bfloat16
    assert sparse_act_T.dtype == torch.bfloat16
    if fc1w.dtype == torch.bfloat16:
        torch.ops.chipmunk.csp_mlp_mm1(x, fc1w, sparse_act_packed, fc1b, sparse_act_T, indices, counts)
    elif fc1w.dtype == torch.float8_e4m3fn:
        # reference: triton csp_mlp_mm1_fp8(x, fc1w.T, ...) (ops/mlp.py:22-23).  That kernel also stores the new
        # activation into the cache (triton/csp_mlp_mm1.py:140) and the scatter-add then adds the delta again; here
        # the cache is left to the scatter-add like on the bf16 path (set FP8_MM1_UPDATES_CACHE for the literal form).
        csp_mlp_mm1_fp8(x, fc1w, fc1b, indices, counts, sparse_act_T, sparse_act_packed, scale_a, scale_b)
    else:
        raise ValueError(f"Unsupported dtype: {fc1w.dtype}")


def mm1_scatter(x: torch.Tensor, fc1w: torch.Tensor, sparse_act_packed: torch.Tensor, fc1b: torch.Tensor,
                sparse_act_T: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor) -> None:
    """GEMM1 + ``csp_scatter_add`` of its output into ``sparse_act_T`` in one kernel (bf16 only)."""
    assert x.dtype == torch.bfloat16 and sparse_act_packed.dtype == torch.bfloat16 and sparse_act_T.dtype == torch.bfloat16
    torch.ops.chipmunk.csp_mlp_mm1_scatter(x, fc1w, sparse_act_packed, fc1b, sparse_act_T, indices, counts)


def mm1_fp8_scatter(x: torch.Tensor, fc1w: torch.Tensor, sparse_act_packed: torch.Tensor, fc1b: torch.Tensor,
                    sparse_act_T: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor, scale_a: torch.Tensor,
                    scale_b: torch.Tensor) -> None:
    """fp8 GEMM1 + ``csp_scatter_add`` of its output into ``sparse_act_T`` in one kernel (the fp8 counterpart of ``mm1_scatter``;
    bit-identical in the packed deltas and in the cache to ``csp_mlp_mm1_fp8`` followed by the scatter-add)."""
    assert x.dtype == torch.float8_e4m3fn and fc1w.dtype == torch.float8_e4m3fn
    torch.ops.chipmunk.csp_mlp_mm1_fp8_scatter(x, fc1w.contiguous(), sparse_act_packed, fc1b, sparse_act_T, indices, counts,
                                               scale_a.reshape(1).float(), scale_b.reshape(1).float())


FP8_MM1_UPDATES_CACHE = False


def csp_mlp_mm1_fp8(a: torch.Tensor, b: torch.Tensor, fc1b: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
                    sparse_act_unpacked_inout: torch.Tensor, sparse_act_packed_out: torch.Tensor,
                    scale_a: torch.Tensor, scale_b: torch.Tensor) -> None:
    """Native counterpart of the reference's Triton ``csp_mlp_mm1_fp8`` (same argument order,
    triton/csp_mlp_mm1.py:143).  ``a`` fp8 ``[M,K]``; ``b`` fp8 ``[F,K]`` (the reference passes ``fc1w.T``, a view of
    the same storage); scales are the RECIPROCAL quantisation scales (modules/mlp.py:98-99)."""
    if b.shape[0] == a.shape[1] and b.shape[1] != a.shape[1]:
        b = b.T  # accept the reference's fc1w.T view
    torch.ops.chipmunk.csp_mlp_mm1_fp8(a, b.contiguous(), sparse_act_packed_out, fc1b, sparse_act_unpacked_inout,
                                       indices, counts, scale_a.reshape(1).float(), scale_b.reshape(1).float(),
                                       FP8_MM1_UPDATES_CACHE)


def mm2_fused(packed: torch.Tensor, unpacked_colmajor: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
              sparse_act_packed: torch.Tensor, fc2wT: torch.Tensor, cached_out: torch.Tensor,
              num_sms_scatter_add: int) -> None:
    assert sparse_act_packed.dtype == torch.bfloat16
    assert fc2wT.dtype == torch.bfloat16
    assert cached_out.dtype == torch.bfloat16
    torch.ops.chipmunk.csp_mlp_mm2_and_scatter_add(
        packed.unsqueeze(0), unpacked_colmajor.unsqueeze(0), indices.unsqueeze(0), counts.unsqueeze(0),
        sparse_act_packed.unsqueeze(0), fc2wT.unsqueeze(0), cached_out.unsqueeze(0), num_sms_scatter_add,
        csp_mlp_mm2_function_ptr)


def csp_mlp_mm2(sparse_act_packed: torch.Tensor, fc2wT: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
                cached_out: torch.Tensor, num_sms: int = 0) -> None:
    """Native counterpart of the reference's Triton ``csp_mlp_mm2`` (triton/csp_mlp_mm2.py:104-129)."""
    torch.ops.chipmunk.csp_mlp_mm2(sparse_act_packed, fc2wT, indices, counts, cached_out)


def mm2_unfused(sparse_act_packed: torch.Tensor, fc2wT: torch.Tensor, cached_out: torch.Tensor,
                unpacked_colmajor: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
                num_sms_scatter_add: int) -> None:
    assert sparse_act_packed.dtype == torch.bfloat16
    assert fc2wT.dtype == torch.bfloat16
    assert cached_out.dtype == torch.bfloat16
    scatter_add(sparse_act_packed, unpacked_colmajor, indices, counts, num_sms_scatter_add)
    csp_mlp_mm2(sparse_act_packed, fc2wT, indices, counts, cached_out, 132 - num_sms_scatter_add)


@torch.compiler.disable
def run_e2e(x: torch.Tensor, fc1w: torch.Tensor, fc1b: torch.Tensor, fc2w_T: torch.Tensor, indices: torch.Tensor,
            counts: torch.Tensor, sparse_act_T: torch.Tensor, cached_out: torch.Tensor, num_sms_scatter_add: int,
            mm1_scale_a: Optional[torch.Tensor] = None, mm1_scale_b: Optional[torch.Tensor] = None) -> None:
    M, K1 = x.shape
    K2, K1_ = fc1w.shape
    assert K1 == K1_, "K1 must match"
    K2_, _N = fc2w_T.shape
    assert K2 == K2_, "K2 must match"
    sparse_act_packed = torch.empty((M, K2), device=x.device, dtype=sparse_act_T.dtype)  # bf16 also when x is fp8
    if (x.is_cuda and fc1w.dtype == torch.bfloat16
            and amd_key("mlp", "fused_scatter")):
        # GEMM1 applies the scatter-add of its own deltas (same bits as the two-kernel form), GEMM2 runs alone
        mm1_scatter(x, fc1w, sparse_act_packed, fc1b, sparse_act_T, indices, counts)
        csp_mlp_mm2(sparse_act_packed, fc2w_T, indices, counts, cached_out)
        return
    if (x.is_cuda and fc1w.dtype == torch.float8_e4m3fn and x.dtype == torch.float8_e4m3fn and amd_key("mlp", "fused_scatter")
            and not FP8_MM1_UPDATES_CACHE and x.shape[1] % 128 == 0):
        mm1_fp8_scatter(x, fc1w, sparse_act_packed, fc1b, sparse_act_T, indices, counts, mm1_scale_a, mm1_scale_b)
        csp_mlp_mm2(sparse_act_packed, fc2w_T, indices, counts, cached_out)
        return
    mm1(x, fc1w, sparse_act_packed, fc1b, sparse_act_T, indices, counts, mm1_scale_a, mm1_scale_b)
    if USE_FUSED_MLP_MATMUL_2:
        mm2_fused(sparse_act_packed, sparse_act_T, indices, counts, sparse_act_packed, fc2w_T, cached_out,
                  num_sms_scatter_add)
    else:
        mm2_unfused(sparse_act_packed, fc2w_T, cached_out, sparse_act_T, indices, counts, num_sms_scatter_add)


__all__ = ["mm1", "mm1_scatter", "mm1_fp8_scatter", "mm2_fused", "mm2_unfused", "run_e2e", "csp_mlp_mm2", "csp_mlp_mm1_fp8"]
