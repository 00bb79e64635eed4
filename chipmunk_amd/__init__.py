"""chipmunk_amd -- MI355X (gfx950) native column-sparse DiT hot path behind Chipmunk's operator surface.

Importing the package loads ``lib/libchipmunk_hip.so`` (HIP kernels + C ABI) and the ``cuda`` extension module, whose
static initialisers register ``torch.ops.chipmunk.*`` with the reference's schemas (reference
``src/chipmunk/__init__.py:3`` does ``from . import cuda, triton`` for the same purpose).  Both loads fail loudly.
"""
import os as _os
import sys as _sys

# `python -m chipmunk_amd.build` (and __graft_entry__.build(), which sets the variable) must be able to run before the
# native artefacts exist or while they are stale; everything else gets the loud failure.
_BUILDING = _os.environ.get("CHIPMUNK_AMD_BUILDING") == "1" or "chipmunk_amd.build" in getattr(_sys, "orig_argv", [])

if not _BUILDING:
    from . import _native

    _native.lib()          # the C ABI: raises ImportError if the HIP library has not been built
    from . import cuda     # noqa: E402,F401  TORCH_LIBRARY(chipmunk) registration (chipmunk_amd/csrc/torch_registry.cpp)
    from . import util, ops, modules  # noqa: E402,F401

    __all__ = ["util", "ops", "modules", "cuda"]
