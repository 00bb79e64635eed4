"""Token-reorder permutations as cached index maps + one gather kernel.

The reference's reorders (``ops/patch.py``, ``ops/voxel.py``) are fixed permutations of the token axis for a given
shape.  On GPU tensors they run as ``torch.ops.chipmunk.gather_rows(src, map)`` -- one pass at HBM rate -- with ``map``
computed ONCE per (shape, parameters, device) by pushing ``arange`` through the reference-order index math (the torch
implementations in ``patch.py`` / ``voxel.py``, which stay the CPU path and the definition of the order)."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch

_MAPS: Dict[Tuple, torch.Tensor] = {}


def index_map(key: Tuple, n: int, device: torch.device, reorder: Callable[[torch.Tensor], torch.Tensor],
              inverse: bool = False) -> torch.Tensor:
    """int32 map with ``out[i] = in[map[i]]`` for the permutation ``reorder`` applies to a flat arange(n) (or for its
    inverse).  ``reorder`` is called once, on CPU int32 indices."""
    k = (key, n, str(device), inverse)
    m = _MAPS.get(k)
    if m is None:
        fwd = reorder(torch.arange(n, dtype=torch.int32)).reshape(-1)
        assert fwd.numel() == n
        if inverse:
            inv = torch.empty(n, dtype=torch.int32)
            inv[fwd.long()] = torch.arange(n, dtype=torch.int32)
            fwd = inv
        m = fwd.contiguous().to(device)
        _MAPS[k] = m
    return m


def gather_rows(src: torch.Tensor, map_: torch.Tensor) -> torch.Tensor:
    return torch.ops.chipmunk.gather_rows(src, map_)
