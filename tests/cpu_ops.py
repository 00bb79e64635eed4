"""TEST INFRASTRUCTURE: registers CPU-key implementations of torch.ops.chipmunk.* backed by the oracle.

The product registers the CUDA (HIP) key only and has no CPU path.  The CPU tests need one to drive the Python state
machines (SparseDiffAttn / SparseDiffMlp) end to end without a GPU -- config C1 "plumbing" -- and to record op-call
traces that are compared with the traces of the reference's own modules (tests/golden/make_golden.py).
Nothing outside tests/ imports this file.
"""
from typing import List

import torch

import oracle

CALLS: List[tuple] = []   # (op name, [arg shapes / scalars]) appended on every call when recording is on
RECORD = {"on": False}
_lib = None


def _note(name, *args):
    if RECORD["on"]:
        CALLS.append((name, [tuple(a.shape) if isinstance(a, torch.Tensor) else a for a in args]))


def _csp_attn(q, k, v, o, indices, counts, o_scale):
    _note("csp_attn", q, k, v, o, indices, counts, int(o_scale))
    oracle.csp_attn(q, k, v, o, indices, counts, int(o_scale))


def _csp_128_attn(q, k, v, indices, counts):
    _note("csp_128_attn", q, k, v, indices, counts)
    return oracle.csp_128_attn(q, k, v, indices, counts)


def _dense_attn(q, k, v):
    _note("dense_attn", q, k, v)
    return oracle.dense_attn(q, k, v)


def _dense_colsum_attn(q, k, v, p):
    _note("dense_colsum_attn", q, k, v, p)
    return oracle.dense_colsum_attn(q, k, v, p)


def _mm1(a, b, c, bias, cache, indices, counts):
    _note("csp_mlp_mm1", a, b, c, bias, cache, indices, counts)
    oracle.csp_mlp_mm1(a, b, c, bias, cache, indices, counts)


def _mm2_sa(packed, unpacked, inds, counts, mma_a, mma_b, mma_c, num_sms, kernel):
    _note("csp_mlp_mm2_and_scatter_add", packed, unpacked, inds, counts, mma_a, mma_b, mma_c, int(num_sms))
    oracle.csp_mlp_mm2_and_scatter_add(packed, unpacked, inds[0], counts[0], mma_a, mma_b, mma_c)


def _scatter_add(packed, unpacked, inds, counts, num_sms):
    _note("csp_scatter_add", packed, unpacked, inds, counts, int(num_sms))
    oracle.csp_scatter_add(packed, unpacked, inds[0], counts[0])


def _copy_indices(src, dst, inds, counts):
    _note("copy_indices", src, dst, inds, counts)
    oracle.copy_indices(src, dst, inds, counts)


def _topk_indices(act, indices, counts, sparsity, multiple_of, random_amount):
    _note("topk_indices", act, indices, counts, float(sparsity), int(multiple_of), float(random_amount))
    oracle.topk_indices(act, indices, counts, float(sparsity), int(multiple_of), 0.0)


def _mask_to_indices(mask, multiple_of, pad_to):
    _note("mask_to_indices", mask, int(multiple_of), int(pad_to))
    return oracle.mask_to_indices(mask, int(multiple_of), int(pad_to))


def register() -> None:
    """Idempotent.  Requires the `chipmunk` library (schemas) to be defined, i.e. `import chipmunk_amd` first."""
    global _lib
    if _lib is not None:
        return
    import chipmunk_amd  # noqa: F401
    _lib = torch.library.Library("chipmunk", "IMPL")
    for name, fn in (("csp_attn", _csp_attn), ("csp_128_attn", _csp_128_attn), ("dense_attn", _dense_attn),
                     ("dense_colsum_attn", _dense_colsum_attn), ("csp_mlp_mm1", _mm1),
                     ("csp_mlp_mm2_and_scatter_add", _mm2_sa), ("csp_scatter_add", _scatter_add),
                     ("copy_indices", _copy_indices), ("topk_indices", _topk_indices),
                     ("mask_to_indices", _mask_to_indices)):
        _lib.impl(name, fn, "CPU")


class recording:
    def __enter__(self):
        CALLS.clear()
        RECORD["on"] = True
        return CALLS

    def __exit__(self, *exc):
        RECORD["on"] = False
