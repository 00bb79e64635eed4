from .attn import SparseDiffAttn
from .mlp import SparseDiffMlp

__all__ = ["SparseDiffAttn", "SparseDiffMlp"]
