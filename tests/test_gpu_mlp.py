"""GPU parity of the column-sparse MLP ops against the CPU oracle (torch.ops.chipmunk.* -> C ABI -> HIP)."""
import pytest
import torch

import oracle
from helpers import assert_close_bf16, randn_bf16

pytestmark = pytest.mark.gpu

# The library ships ONE GEMM1 and ONE GEMM2 form.  The measured-and-not-shipped forms live in tools/probes/mm1_forms (a build of the
# same sources with -DCHIPMUNK_MM1_PROBES); tests/test_gpu_mlp_forms.py re-runs this file against that library with CHIPMUNK_MM1_FORMS=1.
import os as _os
_FORMS = _os.environ.get("CHIPMUNK_MM1_FORMS") == "1"
MM1_FORMS = [0, 20, 21] if _FORMS else [0]
MM1_FORMS_10 = [0, 10, 20, 21] if _FORMS else [0]
MM2_FORMS = [0, 10, 14] if _FORMS else [0]


@pytest.fixture(scope="module")
def dev():
    import chipmunk_amd  # noqa: F401
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _index_rows(G, F, counts, seed):
    g = torch.Generator().manual_seed(seed)
    inds = torch.full((G, F), -1, dtype=torch.int32)
    for i in range(G):
        inds[i, :counts[i]] = torch.randperm(F, generator=g)[:counts[i]].to(torch.int32)
    return inds


def _uniform_bf16(*shape, seed, lo=-0.5, hi=0.5):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * (hi - lo) + lo).to(torch.bfloat16)


def test_mm1_known_answer_reversed_identity(dev):
    """The reference's own mm1 check (csp_mlp_mm1.cu:458-486,590-602): reversed identity indices, U(-0.5,0.5) data,
    abs tolerance 0.1 -- at a reduced shape so the CPU side finishes in seconds."""
    M, K, F = 256, 512, 1024
    a, b = _uniform_bf16(M, K, seed=42), _uniform_bf16(F, K, seed=43)
    bias, cache = _uniform_bf16(F, seed=44), _uniform_bf16(F, M, seed=45)
    inds = torch.arange(F - 1, -1, -1, dtype=torch.int32).expand(M // 128, F).contiguous()
    counts = torch.full((M // 128,), F, dtype=torch.int32)
    c_ref = torch.zeros(M, F, dtype=torch.bfloat16)
    oracle.csp_mlp_mm1(a, b, c_ref, bias, cache, inds, counts)
    c = torch.zeros(M, F, dtype=torch.bfloat16, device=dev)
    torch.ops.chipmunk.csp_mlp_mm1(a.to(dev), b.to(dev), c, bias.to(dev), cache.to(dev), inds.to(dev), counts.to(dev))
    assert (c.float().cpu() - c_ref.float()).abs().max() < 0.1
    assert_close_bf16(c, c_ref, what="mm1 known answer")
    # the in-tree CPU formula directly (csp_mlp_mm1.cu:411-424)
    ref = torch.nn.functional.gelu(a.float() @ b.float().T + bias.float(), approximate="tanh") - cache.float().T
    assert_close_bf16(c, ref.flip(1), what="mm1 vs torch formula")


@pytest.mark.parametrize("variant", MM1_FORMS)
@pytest.mark.parametrize("M,K,F,counts", [(256, 256, 1024, [256, 512]), (384, 320, 768, [16, 272, 768]), (640, 512, 1024, [0, 1024, 136, 512, 8])])
def test_mm1_random_indices_and_ragged_counts(dev, M, K, F, counts, variant, request):
    from chipmunk_amd import _native
    _native.set_option("mm1_variant", variant)
    request.addfinalizer(lambda: _native.set_option("mm1_variant", 0))
    a, b = randn_bf16(M, K, seed=1, scale=0.5), randn_bf16(F, K, seed=2, scale=0.1)
    bias, cache = randn_bf16(F, seed=3, scale=0.2), randn_bf16(F, M, seed=4, scale=0.3)
    cnt = torch.tensor(counts, dtype=torch.int32)
    inds = _index_rows(M // 128, F, counts, seed=5)
    sentinel = 7.0
    c_ref = torch.full((M, F), sentinel, dtype=torch.bfloat16)
    oracle.csp_mlp_mm1(a, b, c_ref, bias, cache, inds, cnt)
    c = torch.full((M, F), sentinel, dtype=torch.bfloat16, device=dev)
    torch.ops.chipmunk.csp_mlp_mm1(a.to(dev), b.to(dev), c, bias.to(dev), cache.to(dev), inds.to(dev), cnt.to(dev))
    assert_close_bf16(c, c_ref, what="mm1")
    for g, n in enumerate(counts):  # columns past counts[g] are untouched
        assert (c[g * 128:(g + 1) * 128, n:].float() == sentinel).all()


@pytest.mark.parametrize("variant", [0, 4, 10, 20, 21] if _FORMS else [0])  # 21: producer / consumer form, DMA stream across tiles; 0: staged epilogue (fused in-kernel), 4: falls back to the scatter kernel, 10: 8 waves on 128 x 256 tiles, 20: producer / consumer form
@pytest.mark.parametrize("M,K,F,counts", [(256, 256, 1024, [256, 512]), (384, 320, 768, [16, 272, 768]),
                                          (4352, 128, 1024, [1024 - 16 * (g % 5) for g in range(34)]),
                                          (1280, 448, 2048, [2048 - 24 * (g % 7) for g in range(10)])])
def test_mm1_scatter_equals_mm1_then_scatter_add(dev, M, K, F, counts, variant):
    """csp_mlp_mm1_scatter == csp_mlp_mm1 followed by csp_scatter_add, bit for bit, in c AND in the cache.  (The last
    shape has 34 groups x 8 column tiles = 272 tiles on 512 resident slots: ONE tile per workgroup -- the persistent loop's later iterations and
    the tail split are covered by tests/test_gpu_mlp_bench_shape.py at the FLUX launch shape.)"""
    from chipmunk_amd import _native
    a, b = randn_bf16(M, K, seed=1, scale=0.5), randn_bf16(F, K, seed=2, scale=0.1)
    bias, cache = randn_bf16(F, seed=3, scale=0.2), randn_bf16(F, M, seed=4, scale=0.3)
    cnt = torch.tensor(counts, dtype=torch.int32)
    inds = _index_rows(M // 128, F, counts, seed=5)
    ad, bd, biasd, indd, cntd = a.to(dev), b.to(dev), bias.to(dev), inds.to(dev), cnt.to(dev)
    c_ref = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    cache_ref = cache.clone().to(dev)
    torch.ops.chipmunk.csp_mlp_mm1(ad, bd, c_ref, biasd, cache_ref, indd, cntd)
    torch.ops.chipmunk.csp_scatter_add(c_ref[None], cache_ref[None], indd[None], cntd[None], 6)
    c = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev)
    cache_new = cache.clone().to(dev)
    _native.set_option("mm1_variant", variant)
    try:
        torch.ops.chipmunk.csp_mlp_mm1_scatter(ad, bd, c, biasd, cache_new, indd, cntd)
    finally:
        _native.set_option("mm1_variant", 0)
    assert torch.equal(c.view(torch.int16), c_ref.view(torch.int16))
    assert torch.equal(cache_new.view(torch.int16), cache_ref.view(torch.int16))


@pytest.mark.parametrize("M,F,counts", [(256, 512, [64, 192]), (384, 1024, [1024, 8, 320])])
def test_scatter_add(dev, M, F, counts):
    packed = randn_bf16(M, F, seed=11)
    unpacked = randn_bf16(F, M, seed=12)
    cnt = torch.tensor(counts, dtype=torch.int32)
    inds = _index_rows(M // 128, F, counts, seed=13)
    ref = unpacked.clone()
    oracle.csp_scatter_add(packed, ref, inds, cnt)
    out = unpacked.clone().to(dev)
    torch.ops.chipmunk.csp_scatter_add(packed.to(dev)[None], out[None], inds.to(dev)[None], cnt.to(dev)[None], 6)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16)), "scatter_add must be bit-exact"


@pytest.mark.parametrize("variant", MM2_FORMS)  # 0: shipped (8 waves, k step 32); 10: 8 waves, k step 64; 14: 4 waves
@pytest.mark.parametrize("M,F,N2,counts", [(256, 512, 256, [64, 192]), (384, 1024, 384, [512, 8, 328]),
                                           (128, 512, 768, [512])])
def test_mm2_and_scatter_add(dev, M, F, N2, counts, variant):
    from chipmunk_amd import _native
    _native.set_option("mm2_variant", variant)
    try:
        _mm2_and_scatter_add(dev, M, F, N2, counts)
    finally:
        _native.set_option("mm2_variant", 0)


def _mm2_and_scatter_add(dev, M, F, N2, counts):
    packed = randn_bf16(M, F, seed=21, scale=0.2)
    # columns past the count hold garbage in the real pipeline (torch.empty): make sure they are never read
    for g, n in enumerate(counts):
        packed[g * 128:(g + 1) * 128, n:] = float("nan")
    unpacked = randn_bf16(F, M, seed=22)
    w2t = randn_bf16(F, N2, seed=23, scale=0.1)
    out0 = randn_bf16(M, N2, seed=24)
    cnt = torch.tensor(counts, dtype=torch.int32)
    inds = _index_rows(M // 128, F, counts, seed=25)
    unp_ref, out_ref = unpacked.clone(), out0.clone()
    oracle.csp_mlp_mm2_and_scatter_add(packed, unp_ref, inds, cnt, packed, w2t, out_ref)
    unp, out = unpacked.clone().to(dev), out0.clone().to(dev)
    pk = packed.to(dev)
    torch.ops.chipmunk.csp_mlp_mm2_and_scatter_add(pk[None], unp[None], inds.to(dev)[None], cnt.to(dev)[None], pk[None],
                                                   w2t.to(dev)[None], out[None], 6, 0)
    assert torch.equal(unp.cpu().view(torch.int16), unp_ref.view(torch.int16))
    assert_close_bf16(out, out_ref, what="mm2")


def test_run_e2e_matches_dense_delta(dev):
    """ops.mlp end to end: with ALL columns selected the sparse step must reproduce the dense MLP on the new input
    (cache + delta == fresh activations), the identity the whole method rests on (modules/mlp.py:51-116)."""
    import chipmunk_amd
    M, K, F = 256, 256, 1024
    x0, x1 = randn_bf16(M, K, seed=31), randn_bf16(M, K, seed=32)
    w1, b1 = randn_bf16(F, K, seed=33, scale=0.06), randn_bf16(F, seed=34, scale=0.1)
    w2, b2 = randn_bf16(K, F, seed=35, scale=0.03), randn_bf16(K, seed=36, scale=0.1)
    x0d, x1d, w1d, b1d, w2d, b2d = [t.to(dev) for t in (x0, x1, w1, b1, w2, b2)]
    act0 = torch.nn.functional.gelu(x0d @ w1d.T + b1d, approximate="tanh")
    out_cache = (act0 @ w2d.T + b2d).contiguous()
    act_T = act0.T.contiguous()
    inds = torch.arange(F, dtype=torch.int32, device=dev).expand(M // 128, F).contiguous()
    cnt = torch.full((M // 128,), F, dtype=torch.int32, device=dev)
    chipmunk_amd.ops.mlp(x1d, w1d, b1d, w2d.T.contiguous(), inds, cnt, act_T, out_cache, 6)
    act1 = torch.nn.functional.gelu(x1d @ w1d.T + b1d, approximate="tanh")
    ref = act1 @ w2d.T + b2d
    assert_close_bf16(act_T.T, act1, atol=3e-2, what="activation cache after scatter-add")
    assert_close_bf16(out_cache, ref, atol=6e-2, rtol=3e-2, what="sparse step output")


@pytest.mark.parametrize("variant", MM1_FORMS_10)   # 10: 8 waves on 128 x 256 tiles, 20 / 21: producer / consumer forms
@pytest.mark.parametrize("update_cache", [False, True])
def test_mm1_fp8_vs_oracle(dev, update_cache, variant, request):
    """BASELINE config C5: fp8 e4m3fn GEMM1 (reference triton/csp_mlp_mm1.py:37-164), Wan-like K = 1536."""
    from chipmunk_amd import _native
    _native.set_option("mm1_variant", variant)
    request.addfinalizer(lambda: _native.set_option("mm1_variant", 0))
    M, K, F = 256, 1536, 1024
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(F, K, generator=g) * 0.05
    sa, sb = 448.0 / x.abs().max(), 448.0 / w.abs().max()
    a8, b8 = (x * sa).to(torch.float8_e4m3fn), (w * sb).to(torch.float8_e4m3fn)
    bias, cache = randn_bf16(F, seed=3, scale=0.2), randn_bf16(F, M, seed=4, scale=0.3)
    counts = [512, 272]
    cnt = torch.tensor(counts, dtype=torch.int32)
    inds = _index_rows(M // 128, F, counts, seed=5)
    ra, rb = torch.tensor([1.0 / sa]), torch.tensor([1.0 / sb])
    c_ref, cache_ref = torch.full((M, F), 7.0, dtype=torch.bfloat16), cache.clone()
    oracle.csp_mlp_mm1_fp8(a8, b8, c_ref, bias, cache_ref, inds, cnt, ra.item(), rb.item(), update_cache)
    c, cache_d = torch.full((M, F), 7.0, dtype=torch.bfloat16, device=dev), cache.clone().to(dev)
    torch.ops.chipmunk.csp_mlp_mm1_fp8(a8.to(dev), b8.to(dev), c, bias.to(dev), cache_d, inds.to(dev), cnt.to(dev),
                                       ra.to(dev), rb.to(dev), update_cache)
    assert_close_bf16(c, c_ref, what="fp8 mm1 packed")
    assert_close_bf16(cache_d, cache_ref, what="fp8 mm1 cache")
    if not update_cache:
        assert torch.equal(cache_d.cpu().view(torch.int16), cache.view(torch.int16))
    # sanity against the unquantised math: the fp8 result approximates gelu(x @ w.T + b) - cache
    ref = torch.nn.functional.gelu(x @ w.T + bias.float(), approximate="tanh") - cache.float().T
    got = c.float().cpu()
    for gidx, n in enumerate(counts):
        rows = slice(gidx * 128, (gidx + 1) * 128)
        cols = inds[gidx, :n].long()
        assert (got[rows, :n] - ref[rows][:, cols]).abs().mean() < 0.05


def test_sparse_mlp_module_fp8_path(dev, fresh_config):
    """SparseDiffMlp with an fp8 fc1 (BASELINE C5 shapes scaled down): full step through F8Linear / torch._scaled_mm,
    sparse step through csp_mlp_mm1_fp8 + scatter-add + GEMM2.  With every column selected the sparse step must track
    the dense fp8 MLP on the new input."""
    import chipmunk_amd
    from chipmunk_amd.modules import SparseDiffMlp, F8Linear
    from chipmunk_amd.util.layer_counter import LayerCounter
    cfg = fresh_config
    cfg["offloading"]["global_disable_offloading"] = True
    cfg["mlp"].update(dict(top_keys=1.0, random_keys=0.0, full_step_every=4, block_mask_cache=2, first_n_dense_layers=0,
                           counts_multiple_of=256))
    torch.manual_seed(0)
    K, F, M = 1536, 2048, 256
    fc1 = F8Linear.from_linear(torch.nn.Linear(K, F, device=dev, dtype=torch.bfloat16),
                               input_float8_dtype=torch.float8_e4m3fn)
    fc2 = torch.nn.Linear(F, K, device=dev, dtype=torch.bfloat16)
    act = torch.nn.GELU(approximate="tanh")
    counter = LayerCounter(1, 1)
    mlp = SparseDiffMlp(0, counter, fc1, act, fc2, 6)
    x0 = torch.randn(1, M, K, device=dev, dtype=torch.bfloat16)
    x1 = x0 + 0.1 * torch.randn_like(x0)
    with torch.no_grad():
        try:
            out0 = mlp(x0)                       # full step (step 0)
        except (RuntimeError, NotImplementedError) as e:   # torch._scaled_mm support varies with the ROCm build
            pytest.skip(f"torch._scaled_mm fp8 unavailable here: {e}")
        out1 = mlp(x1).clone()                   # sparse step (step 1), all columns selected
        ref1 = fc2(act(fc1(x1)))
    assert out0.shape == (1, M, K)
    assert_close_bf16(out1, ref1, atol=8e-2, rtol=5e-2, what="fp8 sparse step vs dense fp8 MLP")


@pytest.mark.parametrize("M,K,F,counts", [(256, 256, 1024, [256, 512]), (384, 128, 768, [0, 768, 256]), (512, 1536, 1024, [512, 256, 1024, 768])])
@pytest.mark.parametrize("variant", MM1_FORMS_10)
def test_fp8_mm1_scatter_equals_fp8_mm1_then_scatter_add(dev, M, K, F, counts, variant, request):
    """csp_mlp_mm1_fp8_scatter == csp_mlp_mm1_fp8 (update_cache off) followed by csp_scatter_add: bit for bit in the packed deltas
    AND in the activation cache (the fp8 counterpart of the bf16 fusion test above)."""
    from chipmunk_amd import _native
    _native.set_option("mm1_variant", variant)
    request.addfinalizer(lambda: _native.set_option("mm1_variant", 0))
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(F, K, generator=g) * 0.06
    sa, sb = 448.0 / x.abs().max(), 448.0 / w.abs().max()
    a8, b8 = (x * sa).to(torch.float8_e4m3fn).to(dev), (w * sb).to(torch.float8_e4m3fn).to(dev)
    ra, rb = (1.0 / sa).reshape(1).float().to(dev), (1.0 / sb).reshape(1).float().to(dev)
    bias = (torch.randn(F, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    cache0 = torch.randn(F, M, generator=g).to(torch.bfloat16).to(dev)
    inds = torch.stack([torch.randperm(F, generator=g) for _ in range(M // 128)]).to(torch.int32).to(dev)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    c_a, cache_a = torch.full((M, F), 3.0, dtype=torch.bfloat16, device=dev), cache0.clone()
    torch.ops.chipmunk.csp_mlp_mm1_fp8(a8, b8, c_a, bias, cache_a, inds, cnt, ra, rb, False)
    torch.ops.chipmunk.csp_scatter_add(c_a.unsqueeze(0), cache_a.unsqueeze(0), inds.unsqueeze(0), cnt.unsqueeze(0), 6)
    c_b, cache_b = torch.full((M, F), 3.0, dtype=torch.bfloat16, device=dev), cache0.clone()
    torch.ops.chipmunk.csp_mlp_mm1_fp8_scatter(a8, b8, c_b, bias, cache_b, inds, cnt, ra, rb)
    torch.cuda.synchronize()
    assert torch.equal(c_a, c_b), "packed deltas"
    assert torch.equal(cache_a, cache_b), "activation cache"
    assert not torch.equal(cache_b, cache0) or sum(counts) == 0


def test_quantize_fp8_matches_the_torch_chain(dev):
    """chipmunk.quantize_fp8 == F8Linear.quantize_input's (x * scale).clamp(-448, 448).to(float8_e4m3fn), bit for bit: normal values, values
    that saturate, ties between fp8 neighbours, the fp8 subnormal range, zeros and a NaN."""
    g = torch.Generator(device=dev).manual_seed(9)
    x = torch.randn(4096, 1536, device=dev, generator=g)
    x[0, :64] = torch.linspace(-3e-3, 3e-3, 64, device=dev)            # fp8 subnormals after scaling by ~40
    x[1, :64] = torch.linspace(-20, 20, 64, device=dev)                 # saturates
    x[2, :8] = torch.tensor([0.0, -0.0, 1.0, 1.0625, 1.1875, 0.40625, float("nan"), 448.0], device=dev)
    x = x.to(torch.bfloat16)
    for scale in (torch.tensor([40.7], device=dev), torch.tensor([1.0], device=dev), torch.tensor([448.0 / 5.3], device=dev)):
        want = (x * scale[0]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
        got = torch.ops.chipmunk.quantize_fp8(x, scale, 448.0)
        assert got.dtype == torch.float8_e4m3fn and got.shape == x.shape
        assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), (
            f"scale {scale.item()}: {(got.view(torch.uint8) != want.view(torch.uint8)).sum().item()} bytes differ")
    # through the module: the fused path and the elementwise path give the same fp8 tensor
    from chipmunk_amd.modules.mlp_fp8 import F8Linear
    from chipmunk_amd.util import config as cfg
    lin = F8Linear.from_linear(torch.nn.Linear(1536, 64, device=dev, dtype=torch.bfloat16), input_float8_dtype=torch.float8_e4m3fn)
    xin = x[3:].contiguous()
    a = lin.quantize_input(xin)
    lin.trial_index, lin.input_scale_initialized = 0, False
    lin.input_amax_trials.zero_()
    cfg.GLOBAL_CONFIG["mlp"]["fused_fp8_quantize"] = False
    try:
        b = lin.quantize_input(xin)
    finally:
        cfg.GLOBAL_CONFIG["mlp"].pop("fused_fp8_quantize", None)
    assert torch.equal(a.view(torch.uint8), b.view(torch.uint8))
