"""Projection output -> attention operands (caller-side glue of the attention ops, SURVEY 8f "callers either side of the path").

The reference's blocks do this in model code (``examples/hunyuan/hyvideo/modules/models.py:188-193, 376-381``):
``rearrange(qkv, "B L (K H D) -> K B L H D")``, ``RMSNorm(head_dim)`` on q and k (``norm_layers.py:43-58``), then the transposes
to the ``[B, H, L, D]`` operands of ``chipmunk.*`` attention.  ``qkv_split_norm`` is that sequence as one HBM pass on the GPU
(``chipmunk_qkv_split_norm``); on CPU tensors it is the reference's op sequence itself.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def _rms_norm_reference(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """norm_layers.py:43-58: normalise in fp32, cast back, multiply by the weight in the tensor's dtype."""
    xf = x.float()
    out = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return out if weight is None else out * weight


def qkv_split_norm(qkv: torch.Tensor, q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor], heads: int,
                   eps: float = 1e-6) -> List[torch.Tensor]:
    """``qkv [n, >= 3*heads*128]`` (one batch element's projection rows) -> ``[q, k, v]``, each ``[1, heads, n, 128]``, q and k
    RMS-normalised over the head dimension."""
    if qkv.is_cuda:
        return torch.ops.chipmunk.qkv_split_norm(qkv, q_weight, k_weight, heads, eps)
    n = qkv.shape[0]
    q, k, v = qkv[:, :3 * heads * 128].reshape(n, 3, heads, 128).permute(1, 2, 0, 3).unsqueeze(1)   # [3][1, H, n, 128]
    return [_rms_norm_reference(q, q_weight, eps).contiguous(), _rms_norm_reference(k, k_weight, eps).contiguous(), v.contiguous()]


__all__ = ["qkv_split_norm"]
