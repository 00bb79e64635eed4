"""chipmunk_amd -- MI355X (gfx950) native column-sparse DiT hot path behind Chipmunk's operator surface.

Importing the package loads ``lib/libchipmunk_hip.so`` (HIP kernels + C ABI) and the ``cuda`` extension module, whose
static initialisers register ``torch.ops.chipmunk.*`` with the reference's schemas (reference
``src/chipmunk/__init__.py:3`` does ``from . import cuda, triton`` for the same purpose).  Both loads fail loudly.
"""
from . import _native

_native.lib()          # the C ABI: raises ImportError if the HIP library has not been built
from . import cuda     # noqa: E402,F401  TORCH_LIBRARY(chipmunk) registration (chipmunk_amd/csrc/torch_registry.cpp)
from . import util, ops, modules  # noqa: E402,F401

__all__ = ["util", "ops", "modules", "cuda"]
