"""Sparse-delta attention layer state machine (mirror of reference ``src/chipmunk/modules/attn.py:16-204``).

Per (inference step, layer) one of four things happens (reference ``_fast_attention``, ``:86-190``):

* layer < ``first_n_dense_layers``               -> dense attention;
* full step, step 0                               -> dense attention, remember ``l`` (softmax denominators);
* full step, step 1 or ``recompute_mask``         -> dense attention with column sums -> pick the keys every 192-query
                                                     group keeps (top-k of the column sums [+ 1% random + static local
                                                     mask]) -> ``o_cache = o_dense - sparse(q, k, v)``;
* sparse step                                     -> ``o = o_cache + sparse(q, k, v)`` over the kept keys only.

Two index styles, selected by the config exactly as in the reference:
``should_compress_indices`` (HunyuanVideo/Wan): a bit-packed bool mask is stored and turned into (indices, counts)
every step; otherwise (FLUX) ``torch.topk`` indices are stored directly and the in-place kernel is used.
With ``attn.fused_packed_mask_to_indices`` (on in BASE_CONFIG, not in the reference) the bit-packed mask goes straight to the
indices kernel instead of through a materialised bool mask -- same indices, 8x less traffic.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .. import ops
from ..ops.voxel import get_local_indices_with_text
from ..util.config import GLOBAL_CONFIG, amd_key
from ..util.layer_counter import LayerCounter
from ..util.storage import AttnStorage
from ..util.storage.offloaded_tensor import release_kept_offloaded, release_resident, reserve_kept_offloaded, reserve_resident

# shared by all layers, initialised from the sequence shape (reference attn.py:12-14)
singleton_static_mask: Optional[Tensor] = None
singleton_video_query_groups: Optional[Tensor] = None


def _cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


def _dense_attn_raw(q: Tensor, k: Tensor, v: Tensor, token_major_o: bool):
    """The operator itself (no padding of l), as the reference's unpadded path calls it (modules/attn.py:126)."""
    return torch.ops.chipmunk.dense_attn_layout(q, k, v, True) if token_major_o else torch.ops.chipmunk.dense_attn(q, k, v)


class SparseDiffAttn(nn.Module):
    def __init__(self, layer_num: int, layer_counter: LayerCounter, storage_slot: int = 0, query_group_offset: int = 0):
        """``storage_slot``: several modules may serve ONE layer (head chunks of a sequence-parallel rank,
        ``distributed.chunk_counters``); each needs its own device slot in the offload pipeline.  ``query_group_offset``:
        first 192-row query group this module sees (query-group sharding: q holds a slice of the sequence's rows, k / v
        the whole sequence); selects the rows of the shared static mask."""
        super().__init__()
        self.layer_num = layer_num
        self.layer_counter = layer_counter
        self.query_group_offset = query_group_offset
        self.storage = AttnStorage(layer_num, init_names=["indices", "out_cache"], slot=storage_slot)
        self.mask_shape = [None] * GLOBAL_CONFIG["num_model_invocations_per_inference_step"]
        self._unpacked = [None] * GLOBAL_CONFIG["num_model_invocations_per_inference_step"]   # (indices, offsets, counts, bytes booked, booked in offload mode)

    # ------------------------------------------------------------------------------------------ static mask
    def initialize_static_mask(self, seq_shape: Tuple, txt_len: int, local_heads_num: int, device: torch.device):
        if len(seq_shape) == 2:
            raise NotImplementedError("Not yet implemented for 2D sequences")
        tt, th, tw = seq_shape
        cfg = GLOBAL_CONFIG["attn"]
        n_video = tt * th * tw
        topk = int(cfg["top_keys"] * n_video)
        lv = cfg["local_voxels"]
        mask, _, _ = get_local_indices_with_text(vid_shape=(tt, th, tw), txt_len=txt_len, voxel_shape=(4, 6, 8),
                                                 local_shape=(lv, lv, lv), rk=cfg["random_keys"], device=device)
        if cfg["local_1d_window"] > 0:
            window = int(cfg["local_1d_window"] * n_video)
            for qg in range(n_video // 192):
                centre = qg * 192 + 96
                mask[qg, max(0, centre - window // 2):min(n_video, centre + window // 2)] = True
        mask = mask[None, None, :, :].expand(1, local_heads_num, -1, -1).contiguous()
        sparse_groups = (mask.sum(dim=-1, keepdim=True) + topk) < (n_video + txt_len)
        global singleton_static_mask, singleton_video_query_groups
        singleton_static_mask = mask
        singleton_video_query_groups = sparse_groups

    def _static(self, heads: int, qg: int, n: int):
        """(static mask, sparse-group flags) for this module's heads and query groups.  The shared tensors hold
        ``local_heads_num`` identical head planes (reference attn.py:67); a module serving fewer heads takes a prefix."""
        g0 = self.query_group_offset
        assert heads <= singleton_static_mask.shape[1] or singleton_static_mask.shape[1] == 1, (
            f"this module serves {heads} heads but initialize_static_mask() built {singleton_static_mask.shape[1]} head planes "
            "(local_heads_num): build the static mask for the largest head chunk")
        h = min(heads, singleton_static_mask.shape[1])      # (one plane broadcasts over the heads)
        return (singleton_static_mask[:, :h, g0:g0 + qg, :n], singleton_video_query_groups[:, :h, g0:g0 + qg, :])

    def random_and_topk(self, cs: Tensor, topk: int) -> Tensor:
        """1% random keys + top-k column sums, limited to the groups that are sparse at all, plus the static mask."""
        cfg = GLOBAL_CONFIG["attn"]
        qg, n = cs.shape[-2], cs.shape[-1]
        static, groups = self._static(cs.shape[1], qg, n)
        if cs.is_cuda and amd_key("attn", "fused_topk_mask") and cs.dtype == torch.bfloat16 and n <= 122880:
            # one kernel for the whole chain below (the random 1 % comes from a counter-based hash, not torch's RNG)
            return ops.topk_mask(cs, topk, 0.01, groups, static)
        mask = torch.randint(0, 100, cs.shape, device=cs.device, dtype=torch.uint8) == 0
        mask.scatter_(-1, cs.topk(k=topk, dim=-1).indices, True)
        # (mask * groups) | static of the reference (modules/attn.py:76-82) as in-place logical ops on the fresh mask: same
        # booleans, no bool x bool product kernel and no two 1.8 GB temporaries at HunyuanVideo size (16 -> 3 ms)
        mask.logical_and_(groups)
        mask.logical_or_(static)
        return mask

    # ------------------------------------------------------------------------------------------ helpers
    def release_kept_indices(self) -> None:
        """Give the HBM booked for the kept index rows back to the residency budget (a module that is dropped or rebuilt -- new model,
        new resolution -- would otherwise leave its share booked for the life of the process)."""
        for inv, old in enumerate(self._unpacked):
            if old is not None:
                self._release_kept(old)
                self._unpacked[inv] = None

    @staticmethod
    def _release_kept(kept) -> None:
        release_resident(kept[3])
        if kept[4]:
            release_kept_offloaded(kept[3])

    def __del__(self):
        try:
            self.release_kept_indices()
        except Exception:       # noqa: BLE001  (interpreter shutdown: the budget module may be gone)
            pass

    @torch.compiler.disable
    def _remember_indices(self, inds: Tensor, counts: Tensor) -> None:
        """``attn.keep_unpacked_indices``: keep what the bit-packed mask just stored unpacks to -- while that mask itself stays in
        HBM, or (``attn.keep_unpacked_indices_offloaded``) also when it travels to pinned host memory -- as ragged rows (``ops.compact_indices``: the kept keys back to back; the padded ``[B, H, G, N]`` int32 tensor is 7 GB per
        HunyuanVideo layer because the text groups keep every key).  One host sync per mask recompute for the total."""
        inv = self.layer_counter.cur_model_invocation_per_step
        old, self._unpacked[inv] = self._unpacked[inv], None
        if old is not None:
            self._release_kept(old)
        self.storage.indices.suppress_current(False)
        if not (inds.is_cuda and amd_key("attn", "keep_unpacked_indices") and amd_key("attn", "fused_residual")
                and (self.storage.indices.is_resident() or amd_key("attn", "keep_unpacked_indices_offloaded"))):
            return
        if not reserve_resident(0):
            return
        flat, offsets = ops.compact_indices(inds, counts)
        nbytes = 4 * flat.numel() + 8 * offsets.numel() + 4 * counts.numel()
        # rows whose mask goes to the host also count against the small dedicated budget (attn.kept_indices_offloaded_budget_gb)
        off_mode = not self.storage.indices.is_resident()
        if off_mode and not reserve_kept_offloaded(nbytes):
            return
        if not reserve_resident(nbytes):
            if off_mode:
                release_kept_offloaded(nbytes)
            return
        self._unpacked[inv] = (flat, offsets, counts, nbytes, off_mode)
        # the sparse steps read these rows: a mask that went to the host need not come back for them
        # (without recompute_mask the full steps unpack the stored mask themselves: it has to come back then)
        self.storage.indices.suppress_current(off_mode and bool(GLOBAL_CONFIG["attn"]["recompute_mask"]))

    def _kept_indices(self):
        """(flat indices, offsets, counts) of the current model invocation if they were kept and their mask is still resident."""
        kept = self._unpacked[self.layer_counter.cur_model_invocation_per_step]
        if kept is not None and GLOBAL_CONFIG["attn"]["should_compress_indices"] and (
                self.storage.indices.is_resident() or amd_key("attn", "keep_unpacked_indices_offloaded")):
            return kept[:3]
        return None

    def _stored_indices(self, multiple_of: int, bm: int):
        cfg = GLOBAL_CONFIG["attn"]
        if cfg["should_compress_indices"]:
            if self.storage.indices.is_suppressed():
                # the kept index rows were expected to serve this step and the mask's host copy was not brought back (configuration changed in
                # between): fetch it now, on the spot
                self.storage.indices.suppress_current(False)
                self.storage.indices.load_async()
                self.storage.indices.load_async_wait()
            packed = self.storage.get_indices()
            shape = self.mask_shape[self.layer_counter.cur_model_invocation_per_step]
            if amd_key("attn", "fused_packed_mask_to_indices") and packed.is_cuda and shape[-1] % 8 == 0:
                if amd_key("attn", "sorted_indices"):
                    return ops.mask_to_sorted_indices(packed, shape, multiple_of, bm)
                return ops.packed_mask_to_indices(packed, shape, multiple_of, bm)
            return ops.mask_to_indices(ops.bitunpack(packed, shape), multiple_of, bm)
        return self.storage.get_indices(), self.storage.get_counts()

    # ------------------------------------------------------------------------------------------ state machine
    def _fast_attention(self, q: Tensor, k: Tensor, v: Tensor, inference_step: int, do_full_step: bool) -> Tensor:
        cfg = GLOBAL_CONFIG["attn"]
        bm = cfg["mbm"]
        assert bm == 192, "The kernel was written for BM=192. You may need to change the kernel."
        do_padding = cfg["pad_qkv_before_kernel"]
        multiple_of = 128 if do_padding else cfg["counts_multiple_of"]

        tm = bool(q.is_cuda and amd_key("attn", "token_major_output"))     # output layout of the dense calls; the sparse ones follow the cache
        if self.layer_num < cfg["first_n_dense_layers"]:
            o, _ = ops.dense_attn(q, k, v, tm)
            return o

        if do_full_step:
            if inference_step == 0:
                o, lse = ops.dense_attn(q, k, v, tm) if do_padding else _dense_attn_raw(q, k, v, tm)
                lse[..., k.shape[-2]:, :] = 0
                self.storage.set_lse_constants(lse)
                return o

            if inference_step == 1 or cfg["recompute_mask"]:
                prev_lse = self.storage.get_lse_constants()
                tk = int(multiple_of * round((cfg["top_keys"] * k.shape[-2]) / multiple_of))
                mask = bs = None
                if (q.is_cuda and cfg["should_compress_indices"] and tk > 0 and amd_key("attn", "fused_colsum_topk")
                        and amd_key("attn", "fused_topk_mask") and k.shape[-2] <= 122880):
                    # dense attention -> column sums -> mask without the column-sum tensor in between (same bits as the two steps)
                    static, groups = self._static(q.shape[1], _cdiv(q.shape[-2], bm), k.shape[-2])
                    o, mask, lse = ops.dense_colsum_topk_mask(q, k, v, prev_lse, tk, 0.01, groups, static, tm)
                elif do_padding:
                    o, bs, lse = ops.dense_colsum_attn(q, k, v, prev_lse, tm)
                else:
                    o, bs, lse = (torch.ops.chipmunk.dense_colsum_attn_layout(q, k, v, prev_lse, True) if tm else
                                  torch.ops.chipmunk.dense_colsum_attn(q, k, v, prev_lse))
                lse[..., k.shape[-2]:, :] = 0
                self.storage.set_lse_constants(lse)
                if cfg["should_compress_indices"]:
                    if mask is not None:
                        pass
                    elif tk > 0:
                        mask = self.random_and_topk(bs, tk)
                    else:
                        mask = self._static(bs.shape[1], bs.shape[-2], bs.shape[-1])[0]
                    packed, mask_shape = ops.bitpack(mask)
                    self.mask_shape[self.layer_counter.cur_model_invocation_per_step] = mask_shape
                    self.storage.set_indices(packed)
                    if mask.is_cuda and amd_key("attn", "fused_packed_mask_to_indices") and amd_key("attn", "sorted_indices"):
                        inds, counts = ops.mask_to_sorted_indices(mask, mask.shape, multiple_of, bm)
                    else:
                        inds, counts = ops.mask_to_indices(mask, multiple_of, bm)
                    self._remember_indices(inds, counts)
                else:
                    kseq = k.shape[-2]
                    bs = bs[..., :_cdiv(kseq, bm), :kseq]
                    inds = torch.topk(bs, k=tk, dim=-1).indices
                    counts = torch.full((q.shape[0], q.shape[1], _cdiv(q.shape[-2], bm)), tk, device=q.device,
                                        dtype=torch.int32)
                    pad = torch.empty((*counts.shape, q.shape[-2] - tk), device=q.device, dtype=torch.int32)
                    inds = torch.cat([inds, pad], dim=-1).to(torch.int32)
                    self.storage.set_indices(inds)
                    self.storage.set_counts(counts)
            else:
                o, _ = ops.dense_attn(q, k, v, tm)

            if not cfg["recompute_mask"]:
                inds, counts = self._stored_indices(multiple_of, bm)

            if o.is_cuda and amd_key("attn", "fused_residual"):
                # dense - sparse in the attention kernel's epilogue (bf16(o - bf16(sparse)), the same two roundings as the
                # reference's `o - csp_attn(...)`), no 731 MB intermediate at HunyuanVideo size
                o_cache = ops.csp_attn_out(q, k, v, o, inds, counts, -1)
            elif do_padding:
                o_cache = o - ops.csp_attn(q, k, v, inds, counts)
            else:
                o_cache = o.clone()
                ops.csp_attn_inplace(q, k, v, o_cache, inds, counts, -1)
            self.storage.set_out_cache(o_cache)
            return o

        # sparse step
        o = self.storage.get_out_cache()
        kept = self._kept_indices() if (o.is_cuda and amd_key("attn", "fused_residual")) else None
        if kept is not None:
            return ops.csp_attn_out_ragged(q, k, v, o, kept[0], kept[1], kept[2], 1)    # cache + delta, index rows as kept
        inds, counts = self._stored_indices(multiple_of, bm)
        if do_padding:
            if o.is_cuda and amd_key("attn", "fused_residual"):
                return ops.csp_attn_out(q, k, v, o, inds, counts, 1)   # cache + delta in one kernel; the cache is only read
            return o + ops.csp_attn(q, k, v, inds, counts)
        # Is `o` the persistent cache itself, or a pipeline slot that the next load overwrites from the host copy?  The
        # reference decides on the config flag (attn.py:186-188) because there a flagged tensor always lives on the host;
        # here a flagged tensor may stay resident (offloading.keep_resident_if_fits), so ask the storage.
        persistent = self.storage.out_cache.is_resident()
        if o.is_cuda and amd_key("attn", "fused_residual") and persistent:
            return ops.csp_attn_out(q, k, v, o, inds, counts, 1)  # cache + delta in one kernel, cache untouched
        if persistent:
            o = o.clone()  # the kernel accumulates in place and the cache must survive (reference attn.py:186-188)
        ops.csp_attn_inplace(q, k, v, o, inds, counts, 1)
        return o

    def forward(self, q: Tensor, k: Tensor, v: Tensor) -> Tensor:
        if not GLOBAL_CONFIG["attn"]["is_enabled"]:
            return F.scaled_dot_product_attention(q, k, v)
        do_full_step = self.layer_counter.should_do_full_attn_step()
        out = self._fast_attention(q, k, v, self.layer_counter.cur_inference_step, do_full_step)
        self.layer_counter.increment()
        return out

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)
