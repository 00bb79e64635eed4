"""Sparse-delta MLP layer state machine (mirror of reference ``src/chipmunk/modules/mlp.py:11-123``).

Full step: dense ``fc2(act(fc1(x)))``; cache the transposed activations (column-major ``[F, M]``), the output and the
128-row block means of the fc1 output.  Sparse step: pick, per 128-row group, the fc1 columns whose block mean moved
most since they were last computed (``topk_indices`` on |delta|), recompute only those columns (``csp_mlp_mm1``),
scatter the activation deltas into the cache and add ``delta @ fc2^T`` onto the cached output
(``csp_mlp_mm2_and_scatter_add``).  The fc2 bias is already inside the cached output.
"""
from __future__ import annotations

import torch

from .. import ops
from ..util.config import GLOBAL_CONFIG, amd_key
from ..util.layer_counter import LayerCounter
from ..util.storage import MlpStorage


def block_mean(x: torch.Tensor, mbm: int) -> torch.Tensor:
    """[b, n, c] -> [b, n/mbm, c] mean over consecutive row blocks (reference modules/mlp.py:11-16).  bf16 GPU tensors take the
    one-pass kernel (``mlp.fused_block_mean``: fp32 sums, one rounding -- torch's reduction in another summation order)."""
    b, n, c = x.shape
    if (x.is_cuda and x.dtype == torch.bfloat16 and mbm % 4 == 0 and c % 8 == 0 and n % mbm == 0
            and b * (n // mbm) < 65536       # the kernel's row blocks ride on grid.y
            and amd_key("mlp", "fused_block_mean")):
        return torch.ops.chipmunk.block_mean(x, mbm)
    return x.reshape(b, n // mbm, mbm, c).mean(dim=2)


def _transposed(x: torch.Tensor) -> torch.Tensor:
    """``x.transpose(-1, -2).contiguous()``; one HBM-rate kernel for 16-bit GPU tensors."""
    if x.is_cuda and x.element_size() == 2:
        return torch.ops.chipmunk.transpose_last2(x)
    return x.transpose(-1, -2).contiguous()


class SparseDiffMlp:
    def __init__(self, layer_num: int, layer_counter: LayerCounter, fc1: torch.nn.Linear,
                 activation: torch.nn.Module, fc2: torch.nn.Linear, heuristic_sms_scatter_add: int = 6):
        # lists keep the Linear modules out of any parent nn.Module's parameter registry (reference mlp.py:21-23)
        self.fc1 = [fc1]
        self.fc2 = [fc2]
        self.fc2w_T = [fc2.weight.data.transpose(0, 1).contiguous()]
        self.layer_counter = layer_counter
        self.activation = activation
        self.storage = MlpStorage(layer_num)
        self.num_sms_scatter_add = heuristic_sms_scatter_add

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        fc1, fc2 = self.fc1[0], self.fc2[0]
        cfg = GLOBAL_CONFIG["mlp"]
        if not cfg["is_enabled"]:
            return fc2(self.activation(fc1(x)))

        do_full = self.layer_counter.should_do_full_mlp_step()
        inference_step, layer, _submodule = self.layer_counter.increment()
        assert x.ndim == 3 and x.shape[0] == 1, "x must be (1, N, C)"
        mbm, bm = cfg["mbm"], cfg["bm"]

        if layer < cfg["first_n_dense_layers"]:
            return fc2(self.activation(fc1(x)))

        if do_full:
            mid = fc1(x)
            act = self.activation(mid)
            out = fc2(act)
            self.storage.set_sparse_act_T(_transposed(act))
            self.storage.set_out_cache(out)
            self.storage.set_blockmean_mid_cache(block_mean(mid, mbm))
            return out

        reuse_mask = (inference_step % cfg["block_mask_cache"] != 0 and self.storage.get_indices() is not None
                      and inference_step >= 10)
        if not reuse_mask:
            bmfc1 = fc1(block_mean(x, mbm))
            r = bm // mbm
            cache = self.storage.get_blockmean_mid_cache()
            if r == 1 and bmfc1.is_cuda and amd_key("mlp", "fused_topk_delta") and bmfc1.is_contiguous():
                # one kernel for |bmfc1 - cache| -> top-k indices -> copy of the selected columns into the cache
                inds = torch.empty_like(bmfc1, dtype=torch.int32)
                counts = torch.empty((bmfc1.size(0), bmfc1.size(1)), dtype=torch.int32, device=x.device)
                torch.ops.chipmunk.topk_delta_indices(bmfc1, cache, inds, counts, 1 - cfg["top_keys"],
                                                      cfg["counts_multiple_of"], cfg["random_keys"])
            else:
                mdiff = (bmfc1 - cache).abs()
                b, rows, f = mdiff.shape
                mdiff = mdiff.reshape(b, rows // r, r, f).sum(dim=2)
                inds = torch.empty_like(mdiff, dtype=torch.int32, device=x.device)
                counts = torch.empty((mdiff.size(0), mdiff.size(1)), dtype=torch.int32, device=x.device)
                ops.topk_indices(mdiff, inds, counts, 1 - cfg["top_keys"], cfg["counts_multiple_of"],
                                 cfg["random_keys"])
                ops.copy_indices(bmfc1, cache, inds, counts)
            self.storage.set_indices(inds)
            self.storage.set_counts(counts)

        indices = self.storage.get_indices()[0]
        counts = self.storage.get_counts()[0]
        out_cache = self.storage.get_out_cache()[0]
        sparse_act_T = self.storage.get_sparse_act_T()[0]

        scale_a = scale_b = None
        if fc1.weight.dtype == torch.float8_e4m3fn:
            x = fc1.quantize_input(x)
            scale_a, scale_b = fc1.input_scale_reciprocal, fc1.scale_reciprocal

        ops.mlp(x=x[0], fc1w=fc1.weight.data, fc1b=fc1.bias.data, fc2w_T=self.fc2w_T[0], indices=indices,
                counts=counts, sparse_act_T=sparse_act_T, cached_out=out_cache,
                num_sms_scatter_add=self.num_sms_scatter_add, mm1_scale_a=scale_a, mm1_scale_b=scale_b)

        out_cache = out_cache.unsqueeze(0)
        self.storage.set_out_cache(out_cache)
        return out_cache

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)
