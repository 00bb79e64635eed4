from .attn import SparseDiffAttn
from .mlp import SparseDiffMlp
from .mlp_fp8 import F8Linear, quantize_fp8, recursive_swap_linears

__all__ = ["SparseDiffAttn", "SparseDiffMlp", "F8Linear", "quantize_fp8", "recursive_swap_linears"]
