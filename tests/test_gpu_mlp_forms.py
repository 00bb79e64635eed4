"""The forms that were measured against the shipped kernels and live outside the product library (tools/probes/mm1_forms: GEMM tile
shapes, the two producer / consumer GEMM1 forms, the work-balanced gathered attention launch) stay parity-tested: when
tools/bin/forms/libchipmunk_hip.so has been built (tools/probes/mm1_forms/build.sh), the MLP parity files and the balanced-launch tests run in a
subprocess bound to that library with every form enabled.  Skipped when the forms library is absent (it is not part of build())."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORMS = os.path.join(ROOT, "tools", "bin", "forms", "libchipmunk_hip.so")


@pytest.mark.skipif(not os.path.exists(FORMS), reason="tools/probes/mm1_forms/build.sh has not been run")
def test_probe_forms_pass_the_mlp_parity_suite():
    env = dict(os.environ, CHIPMUNK_MM1_FORMS="1", CHIPMUNK_HIP_LIB=FORMS,
               LD_LIBRARY_PATH=os.path.dirname(FORMS) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "tests/test_gpu_mlp.py", "tests/test_gpu_mlp_bench_shape.py",
                        "tests/test_gpu_attn_forms_balanced.py"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_product_library_ships_one_form_of_each_gemm():
    """Unknown variant numbers are not an error (tuning knob), they simply select the one shipped form: same bits as variant 0."""
    import torch
    import chipmunk_amd  # noqa: F401
    from chipmunk_amd import _native
    if os.environ.get("CHIPMUNK_HIP_LIB"):
        pytest.skip("running against the forms library")
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    M, K, F = 256, 256, 512
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = (torch.randn(F, K, device=dev, generator=g) * 0.1).to(torch.bfloat16)
    bias = torch.zeros(F, device=dev, dtype=torch.bfloat16)
    cache = torch.randn(F, M, device=dev, generator=g).to(torch.bfloat16)
    inds = torch.stack([torch.randperm(F, device=dev, generator=g) for _ in range(M // 128)]).to(torch.int32)
    counts = torch.tensor([256, 384], dtype=torch.int32, device=dev)
    outs = []
    for variant in (0, 20):
        _native.set_option("mm1_variant", variant)
        try:
            c = torch.zeros(M, F, dtype=torch.bfloat16, device=dev)
            torch.ops.chipmunk.csp_mlp_mm1(a, b, c, bias, cache, inds, counts)
            outs.append(c)
        finally:
            _native.set_option("mm1_variant", 0)
    assert torch.equal(outs[0], outs[1])
