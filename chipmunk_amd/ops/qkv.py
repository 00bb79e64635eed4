"""Projection output -> attention operands (caller-side glue of the attention ops, SURVEY 8f "callers either side of the path").

The reference's blocks do this in model code (``examples/hunyuan/hyvideo/modules/models.py:188-193, 376-381``):
``rearrange(qkv, "B L (K H D) -> K B L H D")``, ``RMSNorm(head_dim)`` on q and k (``norm_layers.py:43-58``), the rotary embedding of
the image tokens (``posemb_layers.py:133-172``), then the transposes to the ``[B, H, L, D]`` operands of ``chipmunk.*`` attention.  ``qkv_split_norm`` is that sequence as one HBM pass on the GPU
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


def _rotary_reference(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """posemb_layers.py:133-172, (cos, sin) form, on ``x [1, H, rows, 128]`` with ``cos, sin [rows, 128]``."""
    xf = x.float()
    re, im = xf.reshape(*xf.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-im, re], dim=-1).flatten(-2)
    return (xf * cos + rot * sin).type_as(x)


def qkv_split_norm(qkv: torch.Tensor, q_weight: Optional[torch.Tensor], k_weight: Optional[torch.Tensor], heads: int,
                   eps: float = 1e-6, freqs_cos: Optional[torch.Tensor] = None,
                   freqs_sin: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
    """``qkv [n, >= 3*heads*128]`` (one batch element's projection rows) -> ``[q, k, v]``, each ``[1, heads, n, 128]``, q and k
    RMS-normalised over the head dimension and -- with ``freqs_cos / freqs_sin`` fp32 ``[rows, 128]`` -- rotated (rotary
    embedding of the first ``rows`` tokens: the image tokens; the text tokens behind them carry none)."""
    if qkv.is_cuda:
        return torch.ops.chipmunk.qkv_split_norm(qkv, q_weight, k_weight, heads, eps, freqs_cos, freqs_sin)
    n = qkv.shape[0]
    q, k, v = qkv[:, :3 * heads * 128].reshape(n, 3, heads, 128).permute(1, 2, 0, 3).unsqueeze(1)   # [3][1, H, n, 128]
    q, k = _rms_norm_reference(q, q_weight, eps).contiguous(), _rms_norm_reference(k, k_weight, eps).contiguous()
    if freqs_cos is not None:
        r = freqs_cos.shape[0]
        q[:, :, :r] = _rotary_reference(q[:, :, :r], freqs_cos, freqs_sin)
        k[:, :, :r] = _rotary_reference(k[:, :, :r], freqs_cos, freqs_sin)
    return [q, k, v.contiguous()]


__all__ = ["qkv_split_norm"]


def residual_ln_modulate(x: torch.Tensor, y: Optional[torch.Tensor], gate: Optional[torch.Tensor], shift: torch.Tensor,
                         scale: torch.Tensor, eps: float = 1e-6):
    """The block's row-wise chain between two GEMMs as one HBM pass (``chipmunk_residual_ln_modulate``): with ``y`` and ``gate``
    ``x = x + gate * y`` first (``models.py:262-275, 431``), then ``xm = LayerNorm(x) * (1 + scale) + shift`` (``modulate(norm(x))``,
    ``models.py:184-186``; LayerNorm without affine).  Returns ``(x, xm)``.  On CPU tensors: the reference's torch sequence."""
    if x.is_cuda:
        out = torch.ops.chipmunk.residual_ln_modulate(x, y, gate, shift, scale, eps)
        return (out[0], out[1]) if y is not None else (x, out[0])      # (the operator never returns its own input)
    if y is not None:
        x = torch.addcmul(x, gate, y)
    xn = torch.nn.functional.layer_norm(x, (x.shape[-1],), eps=eps)
    return x, torch.addcmul(shift, xn, 1 + scale)
