"""Bool mask <-> bit-packed uint8 (mirror of reference ``src/chipmunk/ops/bitpack.py:4-69``): 8 mask values per byte,
little-endian within the byte, over the FLATTENED mask, zero-padded to a whole byte.  GPU tensors go through the
single-pass HIP kernels; CPU tensors use the same arithmetic in torch (the reference's implementation is device
agnostic torch code under ``torch.compile``)."""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def bitpack(mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Size]:
    shape = mask.shape
    if mask.is_cuda:
        return torch.ops.chipmunk.bitpack(mask), shape
    flat = mask.reshape(-1).to(torch.uint8)
    pad = (-flat.numel()) % 8
    if pad:
        flat = torch.cat([flat, flat.new_zeros(pad)])
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8)
    packed = (flat.view(-1, 8) * weights).sum(dim=1, dtype=torch.uint8).contiguous()
    return packed, shape


def bitunpack(packed: torch.Tensor, original_shape: Sequence[int]) -> torch.Tensor:
    if packed.is_cuda:
        return torch.ops.chipmunk.bitunpack(packed, list(original_shape))
    total = 1
    for d in original_shape:
        total *= int(d)
    weights = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8)
    bits = (packed.view(-1, 1) & weights) != 0
    return bits.reshape(-1)[:total].view(*original_shape)
