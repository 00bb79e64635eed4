"""Indexed-IO op wrappers (mirror of reference ``src/chipmunk/ops/indexed_io.py:1-37``)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def copy_indices(bm_fc1: torch.Tensor, bm_mid_cache: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor) -> None:
    torch.ops.chipmunk.copy_indices(bm_fc1, bm_mid_cache, indices, counts)


def topk_indices(activations: torch.Tensor, indices_out: torch.Tensor, counts_out: torch.Tensor,
                 sparsity_amount: float, multiple_of: int, rk: float) -> None:
    torch.ops.chipmunk.topk_indices(activations, indices_out, counts_out, sparsity_amount, multiple_of, rk)


def scatter_add(packed: torch.Tensor, unpacked: torch.Tensor, indices: torch.Tensor, counts: torch.Tensor,
                num_sms: int) -> None:
    torch.ops.chipmunk.csp_scatter_add(packed.unsqueeze(0), unpacked.unsqueeze(0), indices.unsqueeze(0),
                                       counts.unsqueeze(0), num_sms)


def mask_to_indices(mask: torch.Tensor, multiple_of: int, pad_to_multiple_of: int) -> List[torch.Tensor]:
    return torch.ops.chipmunk.mask_to_indices(mask, multiple_of, pad_to_multiple_of)


def packed_mask_to_indices(packed: torch.Tensor, shape: Sequence[int], multiple_of: int,
                           pad_to_multiple_of: int) -> List[torch.Tensor]:
    """``mask_to_indices(bitunpack(packed, shape), ...)`` in one kernel (not in the reference)."""
    return torch.ops.chipmunk.packed_mask_to_indices(packed, list(shape), multiple_of, pad_to_multiple_of)


def mask_to_sorted_indices(mask: torch.Tensor, shape: Sequence[int], multiple_of: int,
                           pad_to_multiple_of: int) -> List[torch.Tensor]:
    """Same kept set / counts / padding as ``mask_to_indices`` (``mask`` bool) or ``packed_mask_to_indices`` (``mask``
    uint8 bit-packed, ``shape`` = original mask shape) with ASCENDING columns: sequential DRAM pages for the K/V gather."""
    return torch.ops.chipmunk.mask_to_sorted_indices(mask, list(shape), multiple_of, pad_to_multiple_of)


def topk_mask(cs: torch.Tensor, k: int, random_amount: float = 0.0, groups: Optional[torch.Tensor] = None,
              static_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``((top-k of cs | random) & groups) | static_mask`` as one kernel: the reference's ``random_and_topk``
    (``modules/attn.py:76-82``: randint + topk + scatter_ + two mask combines).  Exactly ``k`` columns per active row
    come from the top-k part (ties at the k-th value broken deterministically); the random part is a counter-based
    hash, so only ``random_amount = 0`` is comparable bit for bit with the torch chain."""
    return torch.ops.chipmunk.topk_mask(cs, k, random_amount, groups, static_mask)


def manual_seed(seed: int) -> None:
    """Seed of the random-key hash used when ``random_amount`` / ``rk`` > 0 (see ``include/chipmunk_hip.h``:
    ``chipmunk_set_random_seed``).  Every launch draws a different set; the same seed and launch order reproduce a run."""
    from .._native import manual_seed as _seed
    _seed(seed)
