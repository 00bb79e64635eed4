"""`import chipmunk` works and the import lines of the reference's model code run unchanged (SURVEY.md 8b "Callers":
examples/flux/src/flux/{model.py:5-7,sampling.py:15-16,util.py:15-16,modules/layers.py:9-10},
examples/hunyuan/hyvideo/modules/models.py:31-33, examples/wan/wan/modules/model.py:11-14)."""
import torch

CALLER_IMPORTS = """
import chipmunk
from chipmunk.util import GLOBAL_CONFIG, LayerCounter
from chipmunk.util.config import load_from_file
from chipmunk.util.config import GLOBAL_CONFIG as G2
from chipmunk.modules import SparseDiffMlp, SparseDiffAttn
from chipmunk.modules import quantize_fp8
from chipmunk.ops import patchify, unpatchify, patchify_rope
from chipmunk.ops.voxel import voxel_chunk_no_padding, reverse_voxel_chunk_no_padding
from chipmunk.util.storage.offloaded_tensor import PIPELINE_DEPTH
from chipmunk.util.storage import MaybeOffloadedTensor, AttnStorage, MlpStorage
from chipmunk.triton import csp_mlp_mm2, csp_mlp_mm1_fp8, csp_mlp_mm2_function_ptr
import chipmunk.cuda
import chipmunk.ops
"""


def test_reference_import_lines_run_unchanged():
    ns = {}
    exec(CALLER_IMPORTS, ns)
    import chipmunk_amd
    assert ns["SparseDiffAttn"] is chipmunk_amd.modules.SparseDiffAttn
    assert ns["GLOBAL_CONFIG"] is chipmunk_amd.util.config.GLOBAL_CONFIG is ns["G2"]
    assert ns["PIPELINE_DEPTH"] == 2
    assert ns["chipmunk"].cuda is chipmunk_amd.cuda
    # importing the package registered the operator library, as `import chipmunk` does in the reference
    for name in ("csp_attn", "csp_128_attn", "dense_attn", "dense_colsum_attn", "csp_mlp_mm1", "csp_mlp_mm2_and_scatter_add",
                 "csp_scatter_add", "topk_indices", "mask_to_indices", "copy_indices"):
        assert hasattr(torch.ops.chipmunk, name)
    # one state, not two copies: a config change through the alias is seen by the implementation modules
    ns["GLOBAL_CONFIG"]["attn"]["top_keys"] = 0.123
    assert chipmunk_amd.util.config.GLOBAL_CONFIG["attn"]["top_keys"] == 0.123
    chipmunk_amd.util.config.reset_to_base()
