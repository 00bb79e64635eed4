"""``import chipmunk`` -- the reference's package name over the MI355X build.

Model code written against the reference (``examples/flux/src/flux/{model,sampling,util}.py``,
``examples/flux/src/flux/modules/layers.py:9-10``, ``examples/hunyuan/hyvideo/modules/models.py:31-33``,
``examples/wan/wan/modules/model.py:11-14``) imports ``chipmunk.modules``, ``chipmunk.util``, ``chipmunk.ops`` ... and
relies on ``import chipmunk`` registering ``torch.ops.chipmunk.*`` (reference ``src/chipmunk/__init__.py:3``).  This
package aliases every ``chipmunk_amd`` module under the same dotted path, so those import lines run unchanged:
``chipmunk.cuda`` is the operator-registry extension, ``chipmunk.triton`` exposes the native counterparts of the two
Triton entry points the reference's wrappers import (``csp_mlp_mm2``, ``csp_mlp_mm1_fp8``).
Loading fails loudly when the HIP library is missing (there is no CPU fallback).
"""
import importlib as _importlib
import pkgutil as _pkgutil
import sys as _sys
import types as _types

import chipmunk_amd as _impl

_SKIP = {"chipmunk_amd.build"}            # the build script is a program, not part of the runtime surface


def _alias_all() -> None:
    names = ["chipmunk_amd"] + [m.name for m in _pkgutil.walk_packages(_impl.__path__, "chipmunk_amd.") if m.name not in _SKIP]
    for real in names:
        mod = _sys.modules.get(real) or _importlib.import_module(real)
        alias = "chipmunk" + real[len("chipmunk_amd"):]
        if alias != "chipmunk":
            _sys.modules[alias] = mod
    # chipmunk.triton: the reference's wrappers do `from chipmunk.triton import csp_mlp_mm2, csp_mlp_mm1_fp8,
    # csp_mlp_mm2_function_ptr` (src/chipmunk/ops/mlp.py:2-3); the pointer is the CUfunction the reference smuggles into
    # csp_mlp_mm2_and_scatter_add as an int -- accepted and ignored by this build
    tri = _types.ModuleType("chipmunk.triton")
    mlp_ops = _sys.modules["chipmunk_amd.ops.mlp"]     # (`chipmunk_amd.ops.mlp` the attribute is run_e2e, as in the reference)
    tri.csp_mlp_mm2 = mlp_ops.csp_mlp_mm2
    tri.csp_mlp_mm1_fp8 = mlp_ops.csp_mlp_mm1_fp8
    tri.csp_mlp_mm2_function_ptr = 0
    tri.__all__ = ["csp_mlp_mm2", "csp_mlp_mm1_fp8", "csp_mlp_mm2_function_ptr"]
    _sys.modules["chipmunk.triton"] = tri
    globals()["triton"] = tri


_alias_all()
from chipmunk_amd import cuda, modules, ops, util  # noqa: E402,F401

__path__ = []   # every submodule is already in sys.modules; nothing is looked up on disk under this name
__all__ = ["cuda", "triton", "modules", "ops", "util"]
