"""In-tree build of the two native artefacts (no JIT cache, so the built files travel with the tree):

* ``chipmunk_amd/lib/libchipmunk_hip.so`` -- the HIP kernels + C ABI (``include/chipmunk_hip.h``), hipcc, gfx950 only;
* ``chipmunk_amd/cuda*.so``               -- the PyTorch operator registry over that ABI (module ``chipmunk_amd.cuda``).

The reference builds one ``CUDAExtension('chipmunk.cuda')`` for sm_90a (reference setup.py:101-143).
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
LIBDIR = os.path.join(ROOT, "lib")
HIP_LIB = os.path.join(LIBDIR, "libchipmunk_hip.so")
TORCH_EXT = os.path.join(ROOT, "cuda.so")
HIP_SOURCES = ["attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "rowwise.hip", "capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _includes(path: str):
    """Local headers a source includes (one level is all this tree has)."""
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith('#include "'):
                h = line.split('"')[1]
                for base in (CSRC, os.path.join(ROOT, "..", "include")):
                    hp = os.path.join(base, h)
                    if os.path.exists(hp):
                        out.append(hp)
    return out


def build_hip_lib(force: bool = False, verbose: bool = False) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header it includes changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    common = [os.path.join(CSRC, "common.h"), os.path.join(ROOT, "..", "include", "chipmunk_hip.h")]
    jobs, objs = [], []
    for s in HIP_SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + common + _includes(src)):
            jobs.append([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", obj, src])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(run, jobs))
    if jobs or _newer(HIP_LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", HIP_LIB] + objs)
    return HIP_LIB


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension

    src = os.path.join(CSRC, "torch_registry.cpp")
    if not (force or _newer(TORCH_EXT, [src, os.path.join(ROOT, "..", "include", "chipmunk_hip.h")])):
        return TORCH_EXT
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    incs = cpp_extension.include_paths("cuda") + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{i}" for i in incs]
    cmd += [src, "-o", TORCH_EXT, f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip",
            f"-L{LIBDIR}", "-lchipmunk_hip", "-Wl,-rpath,$ORIGIN/lib", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TORCH_EXT


def build_all(force: bool = False, verbose: bool = False) -> None:
    build_hip_lib(force, verbose)
    build_torch_ext(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv, verbose=True)
