"""Attention op wrappers (mirror of reference ``src/chipmunk/ops/attn.py:42-169``: same names, arguments, return
shapes).  The reference pads q (and the index rows) to a multiple of 192 with extra copies because its kernels demand
it; the gfx950 kernels mask the ragged last group themselves, so no q copy is made here -- only the documented return
contract is kept: ``l`` comes back padded to a multiple of 192 with zeros past ``n`` ("leave l padded to pass back in",
reference ``:77``) and ``cs`` is cropped to ``[..., ceil(Nk/192), Nk]`` (reference ``:118-126``).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

PM = 192  # query rows per group


def _pad_len(n: int) -> int:
    return ((n + PM - 1) // PM) * PM


def _last_dim_contiguous(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def dense_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, token_major_o: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense attention; returns ``(o [..., n, d], l [..., pad192(n), 1] fp32)`` with ``l[n:] = 0``.  ``token_major_o`` (an
    addition): ``o`` is the ``[B, H, N, D]`` view of ``[B, N, H, D]`` storage, so the model's ``b h s d -> b s (h d)`` is a view;
    ``csp_attn_out`` on such a tensor keeps the layout."""
    n = q.shape[-2]
    q, k, v = _last_dim_contiguous(q), _last_dim_contiguous(k), _last_dim_contiguous(v)
    o, l = torch.ops.chipmunk.dense_attn_layout(q, k, v, True) if token_major_o else torch.ops.chipmunk.dense_attn(q, k, v)
    padded = _pad_len(n)
    if padded != n:
        l = F.pad(l, (0, 0, 0, padded - n))
    return o, l


def dense_colsum_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, p: torch.Tensor, token_major_o: bool = False
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Dense attention + 192-row column sums of the probabilities normalised by last step's ``p``.

    ``p`` is the (padded) ``l`` returned by the previous ``dense_attn`` / ``dense_colsum_attn`` call.
    Returns ``(o, cs [..., ceil(Nk/192), Nk] bf16, l padded)``.
    """
    n = q.shape[-2]
    padded = _pad_len(n)
    assert p.shape[-2] in (n, padded), "p must be the l vector of the previous full step"
    p_rows = p[..., :n, :].contiguous()
    q, k, v = _last_dim_contiguous(q), _last_dim_contiguous(k), _last_dim_contiguous(v)
    if token_major_o:
        o, cs, l = torch.ops.chipmunk.dense_colsum_attn_layout(q, k, v, p_rows, True)
    else:
        o, cs, l = torch.ops.chipmunk.dense_colsum_attn(q, k, v, p_rows)
    if padded != n:
        l = F.pad(l, (0, 0, 0, padded - n))
    kseq = k.shape[-2]
    kgroups = (kseq + PM - 1) // PM
    if cs.shape[-2] != kgroups or cs.shape[-1] != kseq:
        cs = cs[..., :kgroups, :kseq]
    return o, cs, l


def dense_colsum_topk_mask(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, p: torch.Tensor, k_top: int, random_amount: float,
                           groups, static_mask, token_major_o: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``dense_colsum_attn`` followed by ``topk_mask`` on its column sums, without the ``[..., ceil(N/192), N]`` tensor between
    them (reference ``modules/attn.py:131-141``: 3.55 GB per HunyuanVideo layer).  Returns ``(o, mask, l padded)`` -- the same
    bits as the two calls.  GPU only (CPU tensors take the reference's op sequence in the module)."""
    n = q.shape[-2]
    padded = _pad_len(n)
    assert p.shape[-2] in (n, padded), "p must be the l vector of the previous full step"
    o, mask, l = torch.ops.chipmunk.dense_colsum_topk_mask(_last_dim_contiguous(q), _last_dim_contiguous(k), _last_dim_contiguous(v),
                                                           p[..., :n, :].contiguous(), k_top, random_amount, groups, static_mask,
                                                           token_major_o)
    if padded != n:
        l = F.pad(l, (0, 0, 0, padded - n))
    return o, mask, l


def csp_attn(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, indices: torch.Tensor,
             indices_counts: torch.Tensor) -> torch.Tensor:
    """Out-of-place column-sparse attention (reference ``ops/attn.py:134-169`` -> ``csp_128_attn``)."""
    return torch.ops.chipmunk.csp_128_attn(q.contiguous(), k.contiguous(), v.contiguous(), indices.contiguous(),
                                           indices_counts.contiguous())


def csp_attn_inplace(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, indices: torch.Tensor,
                     indices_counts: torch.Tensor, o_scale: int) -> None:
    """``o += o_scale * sparse_attention`` in place (the reference's modules call ``torch.ops.chipmunk.csp_attn``
    directly, ``modules/attn.py:168,189``; this named entry exists so callers can be instrumented)."""
    torch.ops.chipmunk.csp_attn(q, k, v, o, indices, indices_counts, o_scale)


def csp_attn_out(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o_in: torch.Tensor, indices: torch.Tensor,
                 indices_counts: torch.Tensor, o_scale: int) -> torch.Tensor:
    """``o_in + o_scale * sparse_attention`` into a new tensor: the reference's ``o = cache.clone(); csp_attn(..., o, ...)``
    pair (``modules/attn.py:186-188``) as one kernel, without the copy."""
    return torch.ops.chipmunk.csp_attn_out(q, k, v, o_in, indices, indices_counts, o_scale)


def compact_indices(indices: torch.Tensor, indices_counts: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Padded index rows ``[B, H, G, W]`` -> ``(flat, offsets)``: the first ``counts`` entries of every row back to back (rows
    rounded up to 32 entries), ``offsets`` int64 ``[B*H*G + 1]``.  HunyuanVideo: 0.56 GB instead of the 7 GB the padded tensor
    takes because the text groups keep every key.  One host sync (the total)."""
    return tuple(torch.ops.chipmunk.compact_indices(indices.contiguous(), indices_counts.contiguous()))


def csp_attn_out_ragged(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o_in: torch.Tensor, indices: torch.Tensor,
                        offsets: torch.Tensor, indices_counts: torch.Tensor, o_scale: int) -> torch.Tensor:
    """``csp_attn_out`` reading the ragged rows of ``compact_indices``; the same bits as the padded form."""
    return torch.ops.chipmunk.csp_attn_out_ragged(q, k, v, o_in, indices, offsets, indices_counts, o_scale)


__all__ = ["csp_attn", "csp_attn_inplace", "csp_attn_out", "csp_attn_out_ragged", "compact_indices", "dense_attn", "dense_colsum_attn",
           "dense_colsum_topk_mask"]
