"""The one-wave-per-SIMD dense kernel (csrc/attn64.hip), forced through option attn_dense64 = 1 at sizes the oracle
finishes in seconds: same tolerances as test_gpu_attn.py (bf16 o: atol = rtol = 2e-2; l: rtol 1e-3 / 2e-3)."""
import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def forced(dev):
    from chipmunk_amd import _native
    _native.set_option("attn_dense64", 1)
    yield
    _native.set_option("attn_dense64", 0)


@pytest.mark.parametrize("nq,nk", [(256, 256), (512, 64), (1000, 1000), (384, 1984), (1300, 777), (64, 4160)])
def test_dense64_vs_oracle(dev, forced, nq, nk):
    """multiples of the tile sizes, ragged query rows (last workgroup / wave partly empty), ragged key tiles (masked tail,
    padding tiles of the 4-tile unroll), fewer keys than one tile"""
    H = 3
    q = randn_bf16(1, H, nq, 128, seed=nq)
    k = randn_bf16(1, H, nk, 128, seed=nk + 1)
    v = randn_bf16(1, H, nk, 128, seed=nk + 2)
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense64 o {nq}x{nk}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)


def test_dense64_strided_and_batched(dev, forced):
    B, H, N = 2, 4, 576
    base = [randn_bf16(B, N, H, 128, seed=s).to(dev) for s in (1, 2, 3)]
    q, k, v = [t.permute(0, 2, 1, 3) for t in base]
    o, l = torch.ops.chipmunk.dense_attn(q, k, v)
    o_ref, l_ref = oracle.dense_attn(q.cpu().contiguous(), k.cpu().contiguous(), v.cpu().contiguous())
    assert_close_bf16(o, o_ref, what="strided dense64")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=1e-3, atol=0)


@pytest.mark.parametrize("pattern", ["ramp", "spike", "spike_first", "descending"])
def test_dense64_running_max_update_paths(dev, forced, pattern):
    """same constructions as test_gpu_attn.py::test_running_max_update_paths: the reference point of the exponentials must
    move (with the rescale of the accumulator registers) for some lanes of a wave and not for others"""
    n, H = 1152, 2
    g = torch.Generator().manual_seed(11)
    q = torch.randn(1, H, n, 128, generator=g)
    k = torch.randn(1, H, n, 128, generator=g)
    v = torch.randn(1, H, n, 128, generator=g)
    u = torch.randn(128, generator=g)
    u = u / u.norm()
    q = 0.3 * q + 3.0 * u
    if pattern == "ramp":
        k = 0.3 * k + (torch.arange(n).float() / n * 30.0)[None, None, :, None] * u
    elif pattern == "descending":
        k = 0.3 * k + ((n - torch.arange(n)).float() / n * 30.0)[None, None, :, None] * u
    else:
        k = 0.3 * k
        j = 5 if pattern == "spike_first" else 1000
        k[0, :, j] += 40.0 * u
        q[0, :, ::3] *= 0.05
    q, k, v = [t.to(torch.bfloat16) for t in (q, k, v)]
    o_ref, l_ref = oracle.dense_attn(q, k, v)
    o, l = torch.ops.chipmunk.dense_attn(q.to(dev), k.to(dev), v.to(dev))
    assert_close_bf16(o, o_ref, what=f"dense64, {pattern}")
    torch.testing.assert_close(l.cpu(), l_ref, rtol=2e-3, atol=0)


def test_dense64_matches_general_kernel_at_scale(dev):
    """8 448 x 8 448, 2 heads: the two dense kernels agree (bf16 rounding of different summation orders only)"""
    from chipmunk_amd import _native
    q, k, v = [randn_bf16(1, 2, 8448, 128, seed=s).to(dev) for s in (5, 6, 7)]
    _native.set_option("attn_dense64", 2)
    o_a, l_a = torch.ops.chipmunk.dense_attn(q, k, v)
    _native.set_option("attn_dense64", 1)
    o_b, l_b = torch.ops.chipmunk.dense_attn(q, k, v)
    _native.set_option("attn_dense64", 0)
    assert_close_bf16(o_b, o_a.float().cpu(), what="dense64 vs general kernel")
    torch.testing.assert_close(l_b, l_a, rtol=1e-3, atol=0)
