"""Operator wrappers with the reference's names (``src/chipmunk/ops/__init__.py:1-7``)."""
from .mlp import run_e2e as mlp
from .indexed_io import (copy_indices, topk_indices, mask_to_indices, scatter_add, packed_mask_to_indices,
                         mask_to_sorted_indices, topk_mask, manual_seed)
from .attn import compact_indices, csp_attn, csp_attn_inplace, csp_attn_out, csp_attn_out_ragged, dense_attn, dense_colsum_attn, dense_colsum_topk_mask
from .patch import patchify, unpatchify, patchify_rope
from .bitpack import bitpack, bitunpack
from . import voxel
from .qkv import qkv_split_norm, residual_ln_modulate

__all__ = ["mlp", "copy_indices", "topk_indices", "mask_to_indices", "scatter_add", "csp_attn", "dense_attn",
           "dense_colsum_attn", "patchify", "unpatchify", "patchify_rope", "bitpack", "bitunpack",
           "packed_mask_to_indices", "mask_to_sorted_indices", "csp_attn_inplace", "csp_attn_out", "topk_mask", "voxel", "manual_seed", "qkv_split_norm", "dense_colsum_topk_mask", "compact_indices", "csp_attn_out_ragged", "residual_ln_modulate"]

from . import _fake  # noqa: E402,F401  shape-only ("fake") kernels so torch.compile can trace through the ops
