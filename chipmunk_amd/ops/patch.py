"""FLUX token reorder: two-level (8 then 4) patch order so that 64-token and 192-token groups are spatially local
(mirror of reference ``src/chipmunk/ops/patch.py:7-80``: ``patchify``, ``unpatchify``, ``patchify_rope``).

Pure index permutations, written as one reshape/permute each instead of the reference's einops chains.  The chunk
sizes are read from ``GLOBAL_CONFIG['patchify']`` at call time (the reference freezes them at import, ``:4-5``).
"""
from __future__ import annotations

from typing import Sequence

import torch

from ..util.config import GLOBAL_CONFIG
from . import _reorder


def _chunks():
    cfg = GLOBAL_CONFIG["patchify"]
    return int(cfg["chunk_size_1"]), int(cfg["chunk_size_2"])


def _patchify_torch(x: torch.Tensor, c1: int, c2: int) -> torch.Tensor:
    b, h, w = x.shape
    s = c1 // c2
    x7 = x.reshape(b, h // c1, s, c2, w // c1, s, c2)
    return x7.permute(0, 1, 4, 2, 5, 3, 6).reshape(b, h * w)


def patchify(x: torch.Tensor) -> torch.Tensor:
    """[b, h, w] -> [b, h*w]; output order = (patch row, patch col, sub-patch row, sub-patch col, row, col)."""
    assert x.ndim == 3, "Input tensor must have 3 dimensions (b, h, w)."
    c1, c2 = _chunks()
    b, h, w = x.shape
    assert h % c1 == 0, "Height must be divisible by chunk_size."
    assert w % c1 == 0, "Width must be divisible by chunk_size."
    assert h % c2 == 0, "Height must be divisible by chunk_size_2."
    assert w % c2 == 0, "Width must be divisible by chunk_size_2."
    assert c1 % c2 == 0, "chunk_size_1 must be divisible by chunk_size_2."
    if x.is_cuda:   # one gather with the cached permutation
        m = _reorder.index_map(("patchify", h, w, c1, c2), h * w, x.device, lambda i: _patchify_torch(i.view(1, h, w), c1, c2))
        return _reorder.gather_rows(x.reshape(b, h * w, 1), m).reshape(b, h * w)
    # row = (ph, sh, r), col = (pw, sw, c)
    return _patchify_torch(x, c1, c2)


def unpatchify(x_chunk_flat: torch.Tensor, original_shape: Sequence[int]) -> torch.Tensor:
    """Inverse of :func:`patchify`: [b, h*w] -> [b, h, w]."""
    c1, c2 = _chunks()
    b, h, w = original_shape
    s = c1 // c2
    if x_chunk_flat.is_cuda:
        m = _reorder.index_map(("patchify", h, w, c1, c2), h * w, x_chunk_flat.device,
                               lambda i: _patchify_torch(i.view(1, h, w), c1, c2), inverse=True)
        return _reorder.gather_rows(x_chunk_flat.reshape(b, h * w, 1), m).reshape(b, h, w)
    x7 = x_chunk_flat.reshape(b, h // c1, w // c1, s, s, c2, c2)
    return x7.permute(0, 1, 3, 5, 2, 4, 6).reshape(b, h, w)


def patchify_rope(x_shape: Sequence[int], pe: torch.Tensor, width_rope: int, height_rope: int) -> torch.Tensor:
    """Apply the patch order to the image-token rows of the rotary table ``pe [a, b, tokens, d, e, 2]`` in place."""
    img_tokens = x_shape[1]
    for comp in (0, 1):  # cos, sin
        table = pe[:, :, -img_tokens:, :, :, comp]           # [a, b, hw, d, e]
        a, bb, _, d, e = table.shape
        grid = table.permute(0, 1, 3, 4, 2).reshape(a * bb * d * e, height_rope, width_rope)
        reordered = patchify(grid).reshape(a, bb, d, e, img_tokens).permute(0, 1, 4, 2, 3)
        pe[:, :, -img_tokens:, :, :, comp] = reordered
    return pe
