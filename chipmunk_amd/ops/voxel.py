"""Video token reorder into (vt, vh, vw) voxels and the static local-voxel attention mask (mirror of reference
``src/chipmunk/ops/voxel.py:9-304``; same function names, arguments and results).

Everything here is integer index math.  The reference builds the neighbour table with Python loops over every voxel
(``get_local_voxel_indices``, ``:115-160``); here the per-axis windows are computed once per axis and combined by
broadcasting, which gives the identical table.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


# ----------------------------------------------------------------------------------------------- token reorder
def _split(x: torch.Tensor, voxel_shape: Sequence[int]):
    b, ah, t, h, w, d = x.shape
    vt, vh, vw = voxel_shape
    return (t // vt) * vt, (h // vh) * vh, (w // vw) * vw


def voxel_chunk_no_padding(x: torch.Tensor, voxel_shape: Sequence[int] = (4, 4, 4)) -> torch.Tensor:
    """[b, ah, t, h, w, d] -> [b, ah, t*h*w, d]: full voxels first (voxel raster order, raster inside each voxel),
    then the t-tail, the h-tail and the w-tail in raster order."""
    if x.is_cuda:   # one gather with the cached permutation
        from . import _reorder
        b, ah, t, h, w, d = x.shape
        vs = tuple(int(v) for v in voxel_shape)
        m = _reorder.index_map(("voxel", t, h, w, vs), t * h * w, x.device,
                               lambda i: _voxel_chunk_torch(i.view(1, 1, t, h, w, 1), vs))
        return _reorder.gather_rows(x.reshape(b, ah, t * h * w, d), m)
    return _voxel_chunk_torch(x, voxel_shape)


def _voxel_chunk_torch(x: torch.Tensor, voxel_shape: Sequence[int]) -> torch.Tensor:
    b, ah, t, h, w, d = x.shape
    vt, vh, vw = voxel_shape
    tf, hf, wf = _split(x, voxel_shape)
    main = x[:, :, :tf, :hf, :wf, :].reshape(b, ah, tf // vt, vt, hf // vh, vh, wf // vw, vw, d)
    main = main.permute(0, 1, 2, 4, 6, 3, 5, 7, 8).reshape(b, ah, tf * hf * wf, d)
    tails = [
        x[:, :, tf:, :, :, :].reshape(b, ah, -1, d),
        x[:, :, :tf, hf:, :, :].reshape(b, ah, -1, d),
        x[:, :, :tf, :hf, wf:, :].reshape(b, ah, -1, d),
    ]
    out = torch.cat([main] + tails, dim=2).contiguous()
    assert out.shape[2] == t * h * w
    return out


def reverse_voxel_chunk_no_padding(x_chunk_flat: torch.Tensor, original_shape: Sequence[int],
                                   voxel_shape: Sequence[int] = (4, 4, 4)) -> torch.Tensor:
    """Inverse of :func:`voxel_chunk_no_padding`."""
    b, ah, t, h, w, d = original_shape
    if x_chunk_flat.is_cuda:
        from . import _reorder
        vs = tuple(int(v) for v in voxel_shape)
        m = _reorder.index_map(("voxel", t, h, w, vs), t * h * w, x_chunk_flat.device,
                               lambda i: _voxel_chunk_torch(i.view(1, 1, t, h, w, 1), vs), inverse=True)
        return _reorder.gather_rows(x_chunk_flat.reshape(b, ah, t * h * w, d), m).reshape(b, ah, t, h, w, d)
    vt, vh, vw = voxel_shape
    tf, hf, wf = (t // vt) * vt, (h // vh) * vh, (w // vw) * vw
    out = torch.zeros(tuple(original_shape), dtype=x_chunk_flat.dtype, device=x_chunk_flat.device)
    n_main = tf * hf * wf
    main = x_chunk_flat[:, :, :n_main].reshape(b, ah, tf // vt, hf // vh, wf // vw, vt, vh, vw, d)
    out[:, :, :tf, :hf, :wf, :] = main.permute(0, 1, 2, 5, 3, 6, 4, 7, 8).reshape(b, ah, tf, hf, wf, d)
    cursor = n_main
    for region, shape in (
        ((slice(tf, t), slice(0, h), slice(0, w)), (t - tf, h, w)),
        ((slice(0, tf), slice(hf, h), slice(0, w)), (tf, h - hf, w)),
        ((slice(0, tf), slice(0, hf), slice(wf, w)), (tf, hf, w - wf)),
    ):
        count = shape[0] * shape[1] * shape[2]
        if count > 0:
            piece = x_chunk_flat[:, :, cursor:cursor + count].reshape(b, ah, *shape, d)
            out[:, :, region[0], region[1], region[2], :] = piece
            cursor += count
    return out


# ----------------------------------------------------------------------------------------------- static local mask
def offsets(base_coord: int, full_size: int, offset_range: int) -> List[int]:
    """Window of relative offsets around ``base_coord``: ``offset_range`` on each side, shifted inwards at the borders
    so it always has ``2*offset_range + 1`` entries (reference ``voxel.py:101-113``)."""
    left = [-i for i in range(1, offset_range + 1) if base_coord - i >= 0]
    right = [i for i in range(1, offset_range + 1) if base_coord + i < full_size]
    if len(left) < offset_range:
        for _ in range(offset_range - len(left)):
            right.append(right[-1] + 1)
    elif len(right) < offset_range:
        for _ in range(offset_range - len(right)):
            left.append(left[-1] - 1)
    return sorted(left + [0] + right)


def get_local_voxel_indices(full_shape: Sequence[int], local_shape: Sequence[int]) -> torch.Tensor:
    """For every voxel of a (t, h, w) voxel grid the flat indices of its local (lt, lh, lw) neighbourhood.

    Returns int64 ``[t*h*w, (lt+1)*(lh+1)*(lw+1)]``; slots not covered by the window stay 0 exactly as in the reference
    (odd window sizes leave trailing zero slots, ``voxel.py:130-158``)."""
    t, h, w = full_shape
    lt, lh, lw = local_shape
    width = (lt + 1) * (lh + 1) * (lw + 1)
    inds = torch.zeros((t * h * w, width), dtype=torch.int64)
    if lt == 0 or lh == 0 or lw == 0:
        return inds
    # absolute neighbour coordinates per axis: [size, window]
    at = torch.tensor([[bt + o for o in offsets(bt, t, lt // 2)] for bt in range(t)], dtype=torch.int64)
    ah_ = torch.tensor([[bh + o for o in offsets(bh, h, lh // 2)] for bh in range(h)], dtype=torch.int64)
    aw = torch.tensor([[bw + o for o in offsets(bw, w, lw // 2)] for bw in range(w)], dtype=torch.int64)
    nt, nh, nw = at.shape[1], ah_.shape[1], aw.shape[1]
    flat = (at[:, None, None, :, None, None] * (h * w) + ah_[None, :, None, None, :, None] * w
            + aw[None, None, :, None, None, :])                                     # [t, h, w, nt, nh, nw]
    flat = flat.expand(t, h, w, nt, nh, nw)
    slot = (torch.arange(nt)[:, None, None] * ((lh + 1) * (lw + 1)) + torch.arange(nh)[None, :, None] * (lw + 1)
            + torch.arange(nw)[None, None, :]).reshape(-1)
    inds[:, slot] = flat.reshape(t * h * w, nt * nh * nw)
    return inds


def masktoinds(mask: torch.Tensor, multiple: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-row True columns first (``inds``) and their count rounded up to ``multiple`` (reference ``voxel.py:163-183``)."""
    counts = mask.sum(dim=-1).to(torch.int32)
    if multiple is not None:
        counts = ((counts + multiple - 1) // multiple) * multiple
    inds = mask.to(torch.int8).argsort(dim=-1, descending=True, stable=True)
    return inds.contiguous().to(torch.int32), counts.contiguous().to(torch.int32)


def merge_indices(a: torch.Tensor, b: torch.Tensor, full_shape) -> Tuple[torch.Tensor, torch.Tensor]:
    """Union of two index sets per row (reference ``voxel.py:185-204``)."""
    assert a.shape[:-1] == b.shape[:-1]
    shape = tuple(full_shape.shape) if isinstance(full_shape, torch.Tensor) else tuple(full_shape)
    mask = torch.zeros(shape, device=a.device, dtype=torch.bool)
    mask.scatter_(dim=-1, index=a, value=True)
    mask.scatter_(dim=-1, index=b, value=True)
    return masktoinds(mask)


def get_local_indices_with_text(vid_shape: Sequence[int], txt_len: int, voxel_shape: Sequence[int],
                                local_shape: Sequence[int], full_tail_from_attn: bool = False,
                                full_tail_to_attn: bool = False, rk: float = 0, kv_tile_size: int = 128,
                                device: torch.device = torch.device("cuda")):
    """Static mask ``[n_voxel_groups, vid+txt]``: every query group sees the text tokens and its local voxel cube;
    the trailing text groups see (almost) everything (reference ``voxel.py:206-304``)."""
    tt, th, tw = vid_shape
    lt, lh, lw = local_shape
    vt, vh, vw = voxel_shape
    vid_len = tt * th * tw
    seq_len = vid_len + txt_len
    voxel_size = vt * vh * vw
    n_groups = (seq_len + voxel_size - 1) // voxel_size

    mask = torch.zeros((n_groups, seq_len), device=device, dtype=torch.bool)
    mask[:, vid_len:] = True  # everyone attends to text

    gt, gh, gw = tt // vt, th // vh, tw // vw
    n_img_voxels = gt * gh * gw
    neighbours = get_local_voxel_indices((gt, gh, gw), (lt, lh, lw)).to(device)
    voxel_mask = torch.zeros((n_img_voxels, n_img_voxels), device=device, dtype=torch.bool)
    voxel_mask.scatter_(-1, neighbours, True)
    local = voxel_mask.repeat_interleave(voxel_size, dim=1)[:n_groups, :seq_len]

    pad_rows = n_groups - neighbours.shape[0]
    if pad_rows > 0:
        local = torch.cat([local, torch.zeros((pad_rows, local.shape[1]), device=device, dtype=torch.bool)], dim=0)
    pad_cols = seq_len - local.shape[1]
    if pad_cols > 0:
        filler = torch.ones if full_tail_to_attn else torch.zeros
        local = torch.cat([local, filler((local.shape[0], pad_cols), device=device, dtype=torch.bool)], dim=1)
    local_size = voxel_size * lt * lh * lw
    if local_size > 0:
        # groups past the full voxels (video tail + text) see a 1-D window at the end of the sequence
        local[local.shape[0] - pad_rows:, -local_size:] = True
    mask = mask | local
    tile_aligned = (seq_len // kv_tile_size) * kv_tile_size
    mask[-(txt_len // voxel_size + 1):, -tile_aligned:] = True
    if full_tail_from_attn and pad_rows > 0:
        mask[-pad_rows:, -tile_aligned:] = True
    if rk > 0:
        rand = torch.rand(mask.shape, device=device) < rk
        if full_tail_from_attn and pad_rows > 0:
            rand[-pad_rows:, :] = False
        rand[-(txt_len // voxel_size + 1):, :] = False
        mask = mask | rand
    inds, counts = masktoinds(mask, multiple=kv_tile_size)
    return mask, inds, counts
