"""The C-ABI library loads on a CPU-only box and exports every symbol include/chipmunk_hip.h declares; argument errors
come back as codes + messages (no compute is launched here).  The torch registry exposes the reference's schemas."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "chipmunk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(chipmunk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from chipmunk_amd import _native
    lib = _native.lib()
    declared = _declared_symbols()
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/chipmunk_hip.h but not exported"
    assert sorted(_native.SYMBOLS) == declared
    assert lib.chipmunk_abi_version() == 1


def test_invalid_arguments_return_codes_not_crashes():
    from chipmunk_amd import _native
    lib = _native.lib()
    null = ctypes.c_void_p(0)
    rc = lib.chipmunk_csp_128_attn(null, null, null, null, null, null, 1, 1, 192, 192, 192, null)
    assert rc == 1 and "null" in _native.last_error()
    rc = lib.chipmunk_csp_mlp_mm1(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                  ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 100, 64, 256, null)
    assert rc == 1 and "multiple of 128" in _native.last_error()
    rc = lib.chipmunk_topk_indices(ctypes.c_void_p(16), 0, ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 512,
                                   ctypes.c_double(0.5), 256, ctypes.c_double(0.0), null)
    assert rc == 1 and "1024" in _native.last_error()
    assert lib.chipmunk_set_option(b"no_such_option", 1) == 1


REFERENCE_SCHEMAS = {  # reference csrc/chipmunk.cpp:47-60, verbatim
    "csp_mlp_mm1": "chipmunk::csp_mlp_mm1(Tensor a, Tensor b_colmajor, Tensor(c!) c, Tensor bias, Tensor pa_cache_colmajor, Tensor indices, Tensor indices_counts) -> ()",
    "csp_mlp_mm2_and_scatter_add": "chipmunk::csp_mlp_mm2_and_scatter_add(Tensor packed, Tensor(unpacked_colmajor!) unpacked_colmajor, Tensor sp_inds, Tensor sp_counts, Tensor mma_a, Tensor mma_b, Tensor mma_c, int num_sms_scatter_add, int matmul_kernel) -> ()",
    "csp_attn": "chipmunk::csp_attn(Tensor q, Tensor k, Tensor v, Tensor o, Tensor indices, Tensor indices_counts, int o_scale) -> ()",
    "csp_128_attn": "chipmunk::csp_128_attn(Tensor q, Tensor k, Tensor v, Tensor indices, Tensor indices_counts) -> Tensor",
    "dense_attn": "chipmunk::dense_attn(Tensor q, Tensor k, Tensor v) -> Tensor[]",
    "dense_colsum_attn": "chipmunk::dense_colsum_attn(Tensor q, Tensor k, Tensor v, Tensor p) -> Tensor[]",
    "copy_indices": "chipmunk::copy_indices(Tensor bmfc1, Tensor(bm_mid_cache!) bm_mid_cache, Tensor sp_inds, Tensor sp_counts) -> ()",
    "topk_indices": "chipmunk::topk_indices(Tensor activation, Tensor(indices!) indices, Tensor counts, float sparsity_amount, int multiple_of, float random_amount) -> ()",
    "csp_scatter_add": "chipmunk::csp_scatter_add(Tensor packed, Tensor(unpacked_colmajor!) unpacked_colmajor, Tensor sp_inds, Tensor sp_counts, int num_sms) -> ()",
    "mask_to_indices": "chipmunk::mask_to_indices(Tensor mask, int multiple_of, int pad_to_multiple_of) -> Tensor[]",
}


@pytest.mark.parametrize("name", sorted(REFERENCE_SCHEMAS))
def test_registered_schema_is_the_references(name):
    import chipmunk_amd  # noqa: F401
    schema = str(getattr(torch.ops.chipmunk, name).default._schema)
    assert schema == REFERENCE_SCHEMAS[name]


def test_gpu_only_ops_fail_loudly_on_cpu_tensors():
    """No CPU fallback in the product: a CPU tensor has no kernel registered for it."""
    import subprocess
    import sys
    code = ("import torch, chipmunk_amd\n"
            "q = torch.zeros(1, 1, 192, 128, dtype=torch.bfloat16)\n"
            "try:\n    torch.ops.chipmunk.dense_attn(q, q, q)\nexcept (NotImplementedError, RuntimeError) as e:\n"
            "    print('RAISED', type(e).__name__)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert "RAISED" in out.stdout, out.stdout + out.stderr


def test_fake_kernels_give_shapes_without_a_gpu():
    """torch.compile / FakeTensor tracing: shape-only kernels for the returning ops (chipmunk_amd/ops/_fake.py)."""
    import chipmunk_amd  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        q = torch.empty(1, 2, 400, 128, dtype=torch.bfloat16)
        o, l = torch.ops.chipmunk.dense_attn(q, q, q)
        assert o.shape == q.shape and l.shape == (1, 2, 400, 1) and l.dtype == torch.float32
        _, cs, _ = torch.ops.chipmunk.dense_colsum_attn(q, q, q, l)
        assert cs.shape == (1, 2, 3, 400)
        m = torch.empty(1, 2, 3, 400, dtype=torch.bool)
        i, c = torch.ops.chipmunk.mask_to_indices(m, 128, 192)
        assert i.shape == (1, 2, 3, 576) and c.shape == (1, 2, 3) and i.dtype == torch.int32
        assert torch.ops.chipmunk.bitpack(m).shape == (300,)
