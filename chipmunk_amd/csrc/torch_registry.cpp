// PyTorch operator registry over the C ABI (include/chipmunk_hip.h).
//
// Re-creates the reference's drop-in boundary: library `chipmunk` with byte-identical schemas and an importable
// extension module `cuda` whose load runs the static initialisers (reference csrc/chipmunk.cpp:9-25,45-80).
// ROCm PyTorch dispatches HIP tensors under the `CUDA` key, so FLUX / HunyuanVideo / Wan code written against
// `torch.ops.chipmunk.*` runs unchanged.  Every op enqueues on the CURRENT stream (the reference launches several
// ops on the legacy stream 0: csp_attn.cu:411, csp_mlp_mm1.cu:701, topk_indices.cu, mask_to_indices.cu:133).
// Argument checks mirror the reference's TORCH_CHECKs (file:line cited per op); failures raise c10::Error.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <Python.h>
#include <c10/core/DeviceGuard.h>
#include <torch/library.h>

#include <vector>

#include "../../include/chipmunk_hip.h"

extern "C" {
PyObject *PyInit_cuda(void) {
    static struct PyModuleDef module_def = {PyModuleDef_HEAD_INIT, "cuda", NULL, -1, NULL};
    return PyModule_Create(&module_def);
}
}

namespace chipmunk {
namespace {

void *cur_stream(const at::Tensor &t) {
    return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}
void check(int rc, const char *op) { TORCH_CHECK(rc == 0, "chipmunk::", op, ": ", chipmunk_last_error()); }

#define CHECK_DEV(x) TORCH_CHECK((x).is_cuda(), #x " must be a GPU tensor")
#define CHECK_BF16(x) TORCH_CHECK((x).scalar_type() == at::kBFloat16, #x " must be bfloat16")
#define CHECK_I32(x) TORCH_CHECK((x).scalar_type() == at::kInt, #x " must be a 32-bit integer tensor")
#define CHECK_CONTIG(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")

struct Strides3 {
    int64_t s[3];
};
Strides3 strides_of(const at::Tensor &t, const char *name) {
    TORCH_CHECK(t.dim() == 4, name, " must be a 4D tensor [B,H,N,D]");
    TORCH_CHECK(t.size(3) == 128, "Head dimension must be 128");  // csp_attn.cu:381-383, dense_attn.cu:319-321
    TORCH_CHECK(t.stride(3) == 1, name, " must be contiguous in the head dimension");
    return {{t.stride(0), t.stride(1), t.stride(2)}};
}

void check_attn_shapes(const at::Tensor &q, const at::Tensor &k, const at::Tensor &v) {
    CHECK_DEV(q); CHECK_DEV(k); CHECK_DEV(v);
    CHECK_BF16(q); CHECK_BF16(k); CHECK_BF16(v);
    TORCH_CHECK(q.dim() == 4 && k.dim() == 4 && v.dim() == 4, "q, k, v must be 4D tensors");
    TORCH_CHECK(k.size(0) == q.size(0) && v.size(0) == q.size(0), "batch dimension - idx 0 - must match for all inputs");
    TORCH_CHECK(k.size(1) == q.size(1) && v.size(1) == q.size(1), "QO heads must be equal to KV heads");
    TORCH_CHECK(v.size(2) == k.size(2), "K/V sequence length dimension - idx 2 - must match");
}

void check_indices(const at::Tensor &q, const at::Tensor &indices, const at::Tensor &counts, int64_t groups) {
    CHECK_DEV(indices); CHECK_DEV(counts);
    CHECK_CONTIG(indices); CHECK_CONTIG(counts);
    TORCH_CHECK(counts.dim() == 3, "Indices counts must be a 3D tensor");
    TORCH_CHECK(indices.dim() == 4, "Indices must be a 4D tensor");
    CHECK_I32(indices); CHECK_I32(counts);
    TORCH_CHECK(indices.size(0) == q.size(0) && counts.size(0) == q.size(0), "Indices batch dimension - idx 0 - must match for all inputs");
    TORCH_CHECK(indices.size(1) == q.size(1) && counts.size(1) == q.size(1), "Indices QO head dimension - idx 1 - must match for all inputs");
    TORCH_CHECK(indices.size(2) == groups && counts.size(2) == groups, "Indices query group dimension - idx 2 - must match for all inputs");
}

// ---------------------------------------------------------------------------------- attention
// reference csrc/attn/csp_attn.cu:315-423
void csp_attn(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor o, at::Tensor indices, at::Tensor indices_counts,
              int64_t o_scale) {
    check_attn_shapes(q, k, v);
    CHECK_DEV(o); CHECK_BF16(o);
    TORCH_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");
    TORCH_CHECK(o.sizes() == q.sizes(), "O must have the shape of Q");
    const int64_t groups = (q.size(2) + 191) / 192;
    check_indices(q, indices, indices_counts, groups);
    c10::DeviceGuard guard(q.device());
    auto qs = strides_of(q, "Q"), ks = strides_of(k, "K"), vs = strides_of(v, "V"), os = strides_of(o, "O");
    check(chipmunk_csp_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), qs.s, ks.s, vs.s, os.s,
                            indices.data_ptr<int>(), indices_counts.data_ptr<int>(), (int)q.size(0), (int)q.size(1),
                            (int)q.size(2), (int)k.size(2), (int)indices.size(3), (int)o_scale, cur_stream(q)),
          "csp_attn");
}

// addition: o_in + o_scale * sparse attention into a fresh tensor (the clone + in-place pair of modules/attn.py:186-188)
at::Tensor csp_attn_out(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor o_in, at::Tensor indices,
                        at::Tensor indices_counts, int64_t o_scale) {
    check_attn_shapes(q, k, v);
    CHECK_DEV(o_in); CHECK_BF16(o_in);
    TORCH_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");
    TORCH_CHECK(o_in.sizes() == q.sizes(), "O must have the shape of Q");
    const int64_t groups = (q.size(2) + 191) / 192;
    check_indices(q, indices, indices_counts, groups);
    c10::DeviceGuard guard(q.device());
    // the result takes o_in's layout when that has contiguous 128-element rows (e.g. the token-major cache of a dense call with
    // token_major_o): o_in and o share one set of strides in the C entry
    const bool keep = o_in.stride(3) == 1 && o_in.is_non_overlapping_and_dense();
    at::Tensor oi = keep ? o_in : o_in.contiguous();
    at::Tensor o = at::empty_strided(oi.sizes(), oi.strides(), oi.options());
    auto qs = strides_of(q, "Q"), ks = strides_of(k, "K"), vs = strides_of(v, "V"), os = strides_of(o, "O");
    check(chipmunk_csp_attn_out(q.data_ptr(), k.data_ptr(), v.data_ptr(), oi.data_ptr(), o.data_ptr(), qs.s, ks.s, vs.s,
                                os.s, indices.data_ptr<int>(), indices_counts.data_ptr<int>(), (int)q.size(0),
                                (int)q.size(1), (int)q.size(2), (int)k.size(2), (int)indices.size(3), (int)o_scale,
                                cur_stream(q)),
          "csp_attn_out");
    return o;
}

// addition: padded index rows [B,H,G,W] + counts -> (flat int32 rows back to back, int64 offsets [B*H*G + 1]); every row is
// rounded up to 32 entries (zero filled) so that whole key tiles can be read, 64 spare entries at the end
std::vector<at::Tensor> compact_indices(at::Tensor indices, at::Tensor counts) {
    CHECK_DEV(indices); CHECK_DEV(counts); CHECK_I32(indices); CHECK_I32(counts); CHECK_CONTIG(indices); CHECK_CONTIG(counts);
    TORCH_CHECK(indices.dim() == 4 && counts.dim() == 3 && indices.size(0) == counts.size(0) && indices.size(1) == counts.size(1) &&
                indices.size(2) == counts.size(2), "indices must be [B,H,G,W] and counts [B,H,G]");
    c10::DeviceGuard guard(indices.device());
    const int64_t rows = counts.numel(), W = indices.size(3);
    at::Tensor len = counts.flatten().clamp(0, W).to(at::kLong).add_(31).div_(32, "floor").mul_(32);
    at::Tensor offsets = at::zeros({rows + 1}, len.options());
    if (rows) offsets.narrow(0, 1, rows).copy_(len.cumsum(0));
    const int64_t total = rows ? offsets[rows].item<int64_t>() : 0;      // one host sync (a mask-recompute step, not the sparse step)
    at::Tensor flat = at::empty({total + 64}, indices.options());
    flat.narrow(0, total, 64).zero_();
    check(chipmunk_compact_indices(indices.data_ptr<int>(), W, counts.data_ptr<int>(), offsets.data_ptr<int64_t>(), flat.data_ptr<int>(),
                                   rows, cur_stream(indices)),
          "compact_indices");
    return {flat, offsets};
}

// addition: csp_attn_out over the ragged rows of compact_indices
at::Tensor csp_attn_out_ragged(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor o_in, at::Tensor indices, at::Tensor offsets,
                               at::Tensor indices_counts, int64_t o_scale) {
    check_attn_shapes(q, k, v);
    CHECK_DEV(o_in); CHECK_BF16(o_in); CHECK_DEV(indices); CHECK_DEV(offsets); CHECK_DEV(indices_counts);
    CHECK_I32(indices); CHECK_I32(indices_counts); CHECK_CONTIG(indices); CHECK_CONTIG(offsets); CHECK_CONTIG(indices_counts);
    TORCH_CHECK(offsets.scalar_type() == at::kLong, "offsets must be int64");
    TORCH_CHECK(o_scale == 1 || o_scale == -1, "o_scale must be 1 or -1");
    TORCH_CHECK(o_in.sizes() == q.sizes(), "O must have the shape of Q");
    const int64_t groups = (q.size(2) + 191) / 192;
    TORCH_CHECK(indices_counts.dim() == 3 && indices_counts.size(0) == q.size(0) && indices_counts.size(1) == q.size(1) &&
                indices_counts.size(2) == groups, "counts must be [B, H, ceil(N/192)]");
    TORCH_CHECK(offsets.numel() == indices_counts.numel() + 1, "offsets must have one entry per (batch, head, group) + 1");
    c10::DeviceGuard guard(q.device());
    const bool keep = o_in.stride(3) == 1 && o_in.is_non_overlapping_and_dense();
    at::Tensor oi = keep ? o_in : o_in.contiguous();
    at::Tensor o = at::empty_strided(oi.sizes(), oi.strides(), oi.options());
    auto qs = strides_of(q, "Q"), ks = strides_of(k, "K"), vs = strides_of(v, "V"), os = strides_of(o, "O");
    check(chipmunk_csp_attn_out_ragged(q.data_ptr(), k.data_ptr(), v.data_ptr(), oi.data_ptr(), o.data_ptr(), qs.s, ks.s, vs.s, os.s,
                                       indices.data_ptr<int>(), offsets.data_ptr<int64_t>(), indices_counts.data_ptr<int>(),
                                       (int)q.size(0), (int)q.size(1), (int)q.size(2), (int)k.size(2), (int)o_scale, cur_stream(q)),
          "csp_attn_out_ragged");
    return o;
}

// addition (model code in the reference): x + gate * y -> LayerNorm -> * (1 + scale) + shift in one pass; returns {x_out, xm};
// without y / gate: {x, xm}
std::vector<at::Tensor> residual_ln_modulate(at::Tensor x, const c10::optional<at::Tensor> &y, const c10::optional<at::Tensor> &gate,
                                             at::Tensor shift, at::Tensor scale, double eps) {
    CHECK_DEV(x); CHECK_BF16(x); CHECK_CONTIG(x); CHECK_DEV(shift); CHECK_BF16(shift); CHECK_DEV(scale); CHECK_BF16(scale);
    TORCH_CHECK(x.dim() >= 2, "x must be [..., rows, cols]");
    const int64_t C = x.size(-1), rows = x.numel() / C;
    const bool res = y.has_value() && y->defined();
    TORCH_CHECK(res == (gate.has_value() && gate->defined()), "y and gate come together");
    at::Tensor sh = shift.contiguous(), sc = scale.contiguous(), yc, gc, xo;
    TORCH_CHECK(sh.numel() == C && sc.numel() == C, "shift / scale must have one entry per column");
    if (res) {
        yc = *y; gc = gate->contiguous();
        CHECK_DEV(yc); CHECK_BF16(yc); CHECK_CONTIG(yc); CHECK_DEV(gc); CHECK_BF16(gc);
        TORCH_CHECK(yc.sizes() == x.sizes() && gc.numel() == C, "y must have the shape of x, gate one entry per column");
        xo = at::empty_like(x);
    }
    c10::DeviceGuard guard(x.device());
    at::Tensor xm = at::empty_like(x);
    check(chipmunk_residual_ln_modulate(x.data_ptr(), res ? yc.data_ptr() : nullptr, res ? gc.data_ptr() : nullptr, sh.data_ptr(),
                                        sc.data_ptr(), res ? xo.data_ptr() : nullptr, xm.data_ptr(), rows, (int)C, eps, cur_stream(x)),
          "residual_ln_modulate");
    // without a residual the input is not handed back: an operator's outputs must not alias its inputs (the schema declares fresh
    // tensors; functionalisation / torch.compile rely on it) -- the Python wrapper returns the caller's x beside xm
    if (res) return {xo, xm};
    return {xm};
}

// reference csrc/attn/csp_128_attn.cu:355-461
at::Tensor csp_128_attn(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor indices, at::Tensor indices_counts) {
    check_attn_shapes(q, k, v);
    TORCH_CHECK(q.is_contiguous(), "Q must be contiguous");
    TORCH_CHECK(k.is_contiguous(), "K must be contiguous");
    TORCH_CHECK(v.is_contiguous(), "V must be contiguous");
    TORCH_CHECK(q.size(3) == 128, "Head dimension must be 128");
    // The reference additionally demands seq_len % 192 == 0 and indices.size(3) == seq_len (csp_128_attn.cu:419,429)
    // and its Python wrapper pads q / indices to get there (ops/attn.py:146-161).  The gfx950 kernel masks the ragged
    // last group itself, so both are accepted as-is (a superset of the reference's valid inputs).
    const int64_t groups = (q.size(2) + 191) / 192;
    check_indices(q, indices, indices_counts, groups);
    c10::DeviceGuard guard(q.device());
    at::Tensor o = at::empty(q.sizes(), v.options());
    check(chipmunk_csp_128_attn(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), indices.data_ptr<int>(),
                                indices_counts.data_ptr<int>(), (int)q.size(0), (int)q.size(1), (int)q.size(2),
                                (int)k.size(2), (int)indices.size(3), cur_stream(q)),
          "csp_128_attn");
    return o;
}

// [B, H, N, 128] output of the dense operators: contiguous as the reference's, or (token_major_o, an addition) the permuted view
// of [B, N, H, 128] storage, which `o.permute(0, 2, 1, 3).reshape(B, N, H * 128)` turns into the next GEMM's operand for free
static at::Tensor alloc_o(const at::Tensor &q, const at::Tensor &v, bool token_major) {
    if (!token_major) return at::empty(q.sizes(), v.options().memory_format(at::MemoryFormat::Contiguous));
    return at::empty({q.size(0), q.size(2), q.size(1), q.size(3)}, v.options()).permute({0, 2, 1, 3});
}

// reference csrc/attn/dense_attn.cu:246-372
std::vector<at::Tensor> dense_attn_layout(at::Tensor q, at::Tensor k, at::Tensor v, bool token_major_o) {
    check_attn_shapes(q, k, v);
    c10::DeviceGuard guard(q.device());
    auto qs = strides_of(q, "Q"), ks = strides_of(k, "K"), vs = strides_of(v, "V");
    at::Tensor o = alloc_o(q, v, token_major_o);
    auto os = strides_of(o, "O");
    at::Tensor l = at::empty({q.size(0), q.size(1), q.size(2), 1}, q.options().dtype(at::kFloat));
    check(chipmunk_dense_attn_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), qs.s, ks.s, vs.s, o.data_ptr(), os.s,
                                      l.data_ptr<float>(), (int)q.size(0), (int)q.size(1), (int)q.size(2), (int)k.size(2),
                                      cur_stream(q)),
          "dense_attn");
    return {o, l};
}

std::vector<at::Tensor> dense_attn(at::Tensor q, at::Tensor k, at::Tensor v) { return dense_attn_layout(q, k, v, false); }

// reference csrc/attn/dense_colsum_attn.cu:521-668
std::vector<at::Tensor> dense_colsum_attn_layout(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor p, bool token_major_o) {
    check_attn_shapes(q, k, v);
    CHECK_DEV(p);
    TORCH_CHECK(p.scalar_type() == at::kFloat, "p must be float32");
    TORCH_CHECK(p.is_contiguous(), "p must be contiguous");
    TORCH_CHECK(p.numel() == q.size(0) * q.size(1) * q.size(2), "p must have one entry per query row [B,H,N,1]");
    c10::DeviceGuard guard(q.device());
    auto qs = strides_of(q, "Q"), ks = strides_of(k, "K"), vs = strides_of(v, "V");
    const int64_t groups = (q.size(2) + 191) / 192;
    // the reference's cs has one column per (padded) query position (:580-583: it only ever sees Nk <= Nq); a rank that holds a
    // slice of the query rows against the whole sequence's keys (query-group sharding) gets one column per key
    const int64_t width = std::max<int64_t>(q.size(2), k.size(2));
    at::Tensor o = alloc_o(q, v, token_major_o);
    auto os = strides_of(o, "O");
    at::Tensor cs = at::empty({q.size(0), q.size(1), groups, width}, v.options());
    at::Tensor l = at::empty({q.size(0), q.size(1), q.size(2), 1}, q.options().dtype(at::kFloat));
    check(chipmunk_dense_colsum_attn_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), qs.s, ks.s, vs.s, p.data_ptr<float>(),
                                             o.data_ptr(), os.s, cs.data_ptr(), l.data_ptr<float>(), (int)q.size(0),
                                             (int)q.size(1), (int)q.size(2), (int)k.size(2), (int)width, cur_stream(q)),
          "dense_colsum_attn");
    return {o, cs, l};
}

std::vector<at::Tensor> dense_colsum_attn(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor p) {
    return dense_colsum_attn_layout(q, k, v, p, false);
}

// ---------------------------------------------------------------------------------- MLP
// reference csrc/mlp/csp_mlp_mm1.cu:625-702
void csp_mlp_mm1(at::Tensor a, at::Tensor b_colmajor, at::Tensor c, at::Tensor bias, at::Tensor pa_cache_colmajor,
                 at::Tensor indices, at::Tensor indices_counts) {
    CHECK_DEV(a); CHECK_DEV(b_colmajor); CHECK_DEV(c); CHECK_DEV(bias); CHECK_DEV(pa_cache_colmajor);
    CHECK_DEV(indices); CHECK_DEV(indices_counts);
    CHECK_BF16(a); CHECK_BF16(b_colmajor); CHECK_BF16(c); CHECK_BF16(bias); CHECK_BF16(pa_cache_colmajor);
    CHECK_I32(indices); CHECK_I32(indices_counts);
    CHECK_CONTIG(a); CHECK_CONTIG(b_colmajor); CHECK_CONTIG(c); CHECK_CONTIG(bias); CHECK_CONTIG(pa_cache_colmajor);
    CHECK_CONTIG(indices); CHECK_CONTIG(indices_counts);
    TORCH_CHECK(a.dim() == 2 && b_colmajor.dim() == 2 && c.dim() == 2, "a, b_colmajor, c must be 2D");
    const int64_t M = a.size(0), K = a.size(1), F = b_colmajor.size(0);
    TORCH_CHECK(b_colmajor.size(1) == K, "a and b_colmajor must share the K dimension");
    TORCH_CHECK(c.size(0) == M && c.size(1) == F, "c must be [M, F]");
    TORCH_CHECK(bias.numel() == F, "bias must have F entries");
    TORCH_CHECK(pa_cache_colmajor.numel() == F * M, "pa_cache_colmajor must be [F, M]");
    TORCH_CHECK(indices.numel() == (M / 128) * F && indices_counts.numel() == M / 128, "indices must be [M/128, F], counts [M/128]");
    c10::DeviceGuard guard(a.device());
    check(chipmunk_csp_mlp_mm1(a.data_ptr(), b_colmajor.data_ptr(), c.data_ptr(), bias.data_ptr(),
                               pa_cache_colmajor.data_ptr(), indices.data_ptr<int>(), indices_counts.data_ptr<int>(),
                               (int)M, (int)K, (int)F, cur_stream(a)),
          "csp_mlp_mm1");
}

// addition: GEMM1 + the scatter-add of its output into the activation cache, one kernel
void csp_mlp_mm1_scatter(at::Tensor a, at::Tensor b_colmajor, at::Tensor c, at::Tensor bias, at::Tensor pa_cache_colmajor,
                         at::Tensor indices, at::Tensor indices_counts) {
    CHECK_DEV(a); CHECK_DEV(b_colmajor); CHECK_DEV(c); CHECK_DEV(bias); CHECK_DEV(pa_cache_colmajor);
    CHECK_DEV(indices); CHECK_DEV(indices_counts);
    CHECK_BF16(a); CHECK_BF16(b_colmajor); CHECK_BF16(c); CHECK_BF16(bias); CHECK_BF16(pa_cache_colmajor);
    CHECK_I32(indices); CHECK_I32(indices_counts);
    CHECK_CONTIG(a); CHECK_CONTIG(b_colmajor); CHECK_CONTIG(c); CHECK_CONTIG(bias); CHECK_CONTIG(pa_cache_colmajor);
    CHECK_CONTIG(indices); CHECK_CONTIG(indices_counts);
    TORCH_CHECK(a.dim() == 2 && b_colmajor.dim() == 2 && c.dim() == 2, "a, b_colmajor, c must be 2D");
    const int64_t M = a.size(0), K = a.size(1), F = b_colmajor.size(0);
    TORCH_CHECK(b_colmajor.size(1) == K, "a and b_colmajor must share the K dimension");
    TORCH_CHECK(c.size(0) == M && c.size(1) == F, "c must be [M, F]");
    TORCH_CHECK(bias.numel() == F, "bias must have F entries");
    TORCH_CHECK(pa_cache_colmajor.numel() == F * M, "pa_cache_colmajor must be [F, M]");
    TORCH_CHECK(indices.numel() == (M / 128) * F && indices_counts.numel() == M / 128, "indices must be [M/128, F], counts [M/128]");
    c10::DeviceGuard guard(a.device());
    check(chipmunk_csp_mlp_mm1_scatter(a.data_ptr(), b_colmajor.data_ptr(), c.data_ptr(), bias.data_ptr(),
                                       pa_cache_colmajor.data_ptr(), indices.data_ptr<int>(),
                                       indices_counts.data_ptr<int>(), (int)M, (int)K, (int)F, cur_stream(a)),
          "csp_mlp_mm1_scatter");
}

// native counterpart of the reference's Triton csp_mlp_mm1_fp8 (src/chipmunk/triton/csp_mlp_mm1.py:143-164)
static void mm1_fp8_impl(at::Tensor a, at::Tensor b, at::Tensor c, at::Tensor bias, at::Tensor pa_cache_colmajor,
                         at::Tensor indices, at::Tensor indices_counts, at::Tensor scale_a, at::Tensor scale_b,
                         int update_cache) {
    CHECK_DEV(a); CHECK_DEV(b); CHECK_DEV(c); CHECK_DEV(bias); CHECK_DEV(pa_cache_colmajor);
    CHECK_DEV(indices); CHECK_DEV(indices_counts); CHECK_DEV(scale_a); CHECK_DEV(scale_b);
    TORCH_CHECK(a.scalar_type() == at::kFloat8_e4m3fn && b.scalar_type() == at::kFloat8_e4m3fn,
                "a and b must be float8_e4m3fn (OCP; gfx950 has no fnuz)");
    CHECK_BF16(c); CHECK_BF16(bias); CHECK_BF16(pa_cache_colmajor);
    CHECK_I32(indices); CHECK_I32(indices_counts);
    CHECK_CONTIG(a); CHECK_CONTIG(b); CHECK_CONTIG(c); CHECK_CONTIG(bias); CHECK_CONTIG(pa_cache_colmajor);
    CHECK_CONTIG(indices); CHECK_CONTIG(indices_counts);
    TORCH_CHECK(scale_a.scalar_type() == at::kFloat && scale_b.scalar_type() == at::kFloat && scale_a.numel() == 1 &&
                scale_b.numel() == 1, "scale_a and scale_b must be one-element float32 tensors");
    TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && c.dim() == 2, "a, b, c must be 2D");
    const int64_t M = a.size(0), K = a.size(1), F = b.size(0);
    TORCH_CHECK(b.size(1) == K && c.size(0) == M && c.size(1) == F, "shape mismatch");
    TORCH_CHECK(bias.numel() == F && pa_cache_colmajor.numel() == F * M, "bias must be [F], pa_cache_colmajor [F, M]");
    c10::DeviceGuard guard(a.device());
    check(chipmunk_csp_mlp_mm1_fp8(a.data_ptr(), b.data_ptr(), c.data_ptr(), bias.data_ptr(),
                                   pa_cache_colmajor.data_ptr(), indices.data_ptr<int>(),
                                   indices_counts.data_ptr<int>(), scale_a.data_ptr<float>(), scale_b.data_ptr<float>(),
                                   (int)M, (int)K, (int)F, update_cache, cur_stream(a)),
          "csp_mlp_mm1_fp8");
}

void csp_mlp_mm1_fp8(at::Tensor a, at::Tensor b, at::Tensor c, at::Tensor bias, at::Tensor pa_cache_colmajor,
                     at::Tensor indices, at::Tensor indices_counts, at::Tensor scale_a, at::Tensor scale_b,
                     bool update_cache) {
    mm1_fp8_impl(a, b, c, bias, pa_cache_colmajor, indices, indices_counts, scale_a, scale_b, update_cache ? 1 : 0);
}
// fp8 GEMM1 that applies the scatter-add of its own deltas to the cache (the fp8 counterpart of csp_mlp_mm1_scatter)
void csp_mlp_mm1_fp8_scatter(at::Tensor a, at::Tensor b, at::Tensor c, at::Tensor bias, at::Tensor pa_cache_colmajor,
                             at::Tensor indices, at::Tensor indices_counts, at::Tensor scale_a, at::Tensor scale_b) {
    mm1_fp8_impl(a, b, c, bias, pa_cache_colmajor, indices, indices_counts, scale_a, scale_b, 2);
}

void check_scatter_args(const at::Tensor &packed, const at::Tensor &unpacked, const at::Tensor &inds,
                        const at::Tensor &counts) {
    // reference csrc/indexed_io/scatter_add.cu:111-142 (B is hard-wired to 1, :58-59,138)
    CHECK_DEV(packed); CHECK_DEV(unpacked); CHECK_DEV(inds); CHECK_DEV(counts);
    CHECK_CONTIG(packed); CHECK_CONTIG(unpacked); CHECK_CONTIG(inds); CHECK_CONTIG(counts);
    TORCH_CHECK(packed.dim() == 3, "packed must be a 3D tensor");
    TORCH_CHECK(unpacked.dim() == 3, "unpacked_colmajor must be a 3D tensor");
    TORCH_CHECK(inds.dim() == 3, "sp_inds must be a 3D tensor");
    TORCH_CHECK(counts.dim() == 2, "sp_counts must be a 2D tensor");
    CHECK_BF16(packed); CHECK_BF16(unpacked);
    CHECK_I32(inds); CHECK_I32(counts);
    TORCH_CHECK(packed.size(0) == 1, "batch size must be 1");
    TORCH_CHECK(unpacked.size(1) == packed.size(2) && unpacked.size(2) == packed.size(1), "unpacked_colmajor must be [1, F, M]");
    TORCH_CHECK(inds.size(1) == packed.size(1) / 128 && inds.size(2) == packed.size(2), "sp_inds must be [1, M/128, F]");
    TORCH_CHECK(counts.size(1) == packed.size(1) / 128, "sp_counts must be [1, M/128]");
}

// reference csrc/indexed_io/scatter_add.cu:102-181
void csp_scatter_add(at::Tensor packed, at::Tensor unpacked_colmajor, at::Tensor sp_inds, at::Tensor sp_counts,
                     int64_t num_sms) {
    check_scatter_args(packed, unpacked_colmajor, sp_inds, sp_counts);
    c10::DeviceGuard guard(packed.device());
    check(chipmunk_csp_scatter_add(packed.data_ptr(), unpacked_colmajor.data_ptr(), sp_inds.data_ptr<int>(),
                                   sp_counts.data_ptr<int>(), (int)packed.size(1), (int)packed.size(2), (int)num_sms,
                                   cur_stream(packed)),
          "csp_scatter_add");
}

// reference csrc/mlp/csp_mlp_mm2_and_scatter_add.cu:96-259.  `matmul_kernel` is the reference's Triton CUfunction
// smuggled as an int (:170); the GEMM is native here so the value is accepted and ignored.
void csp_mlp_mm2_and_scatter_add(at::Tensor packed, at::Tensor unpacked_colmajor, at::Tensor sp_inds,
                                 at::Tensor sp_counts, at::Tensor mma_a, at::Tensor mma_b, at::Tensor mma_c,
                                 int64_t num_sms_scatter_add, int64_t matmul_kernel) {
    (void)matmul_kernel;
    check_scatter_args(packed, unpacked_colmajor, sp_inds, sp_counts);
    CHECK_DEV(mma_a); CHECK_DEV(mma_b); CHECK_DEV(mma_c);
    CHECK_BF16(mma_a); CHECK_BF16(mma_b); CHECK_BF16(mma_c);
    CHECK_CONTIG(mma_a); CHECK_CONTIG(mma_b); CHECK_CONTIG(mma_c);
    TORCH_CHECK(mma_a.dim() == 3 && mma_b.dim() == 3 && mma_c.dim() == 3, "mma_a, mma_b, mma_c must be 3D tensors");
    const int64_t M = packed.size(1), F = packed.size(2), N2 = mma_b.size(2);
    TORCH_CHECK(mma_a.size(1) == M && mma_a.size(2) == F, "mma_a must be [1, M, F]");
    TORCH_CHECK(mma_b.size(1) == F, "mma_b must be [1, F, N]");
    TORCH_CHECK(mma_c.size(1) == M && mma_c.size(2) == N2, "mma_c must be [1, M, N]");
    c10::DeviceGuard guard(packed.device());
    check(chipmunk_csp_mlp_mm2_and_scatter_add(packed.data_ptr(), unpacked_colmajor.data_ptr(),
                                               sp_inds.data_ptr<int>(), sp_counts.data_ptr<int>(), mma_a.data_ptr(),
                                               mma_b.data_ptr(), mma_c.data_ptr(), (int)M, (int)F, (int)N2,
                                               (int)num_sms_scatter_add, cur_stream(packed)),
          "csp_mlp_mm2_and_scatter_add");
}

// native counterpart of the reference's Triton csp_mlp_mm2 (src/chipmunk/triton/csp_mlp_mm2.py:104-129)
void csp_mlp_mm2(at::Tensor mma_a, at::Tensor mma_b, at::Tensor indices, at::Tensor counts, at::Tensor mma_c) {
    CHECK_DEV(mma_a); CHECK_DEV(mma_b); CHECK_DEV(mma_c); CHECK_DEV(indices); CHECK_DEV(counts);
    CHECK_BF16(mma_a); CHECK_BF16(mma_b); CHECK_BF16(mma_c);
    CHECK_I32(indices); CHECK_I32(counts);
    CHECK_CONTIG(mma_a); CHECK_CONTIG(mma_b); CHECK_CONTIG(mma_c); CHECK_CONTIG(indices); CHECK_CONTIG(counts);
    TORCH_CHECK(mma_a.dim() == 2 && mma_b.dim() == 2 && mma_c.dim() == 2, "mma_a, mma_b, mma_c must be 2D tensors");
    const int64_t M = mma_a.size(0), F = mma_a.size(1), N2 = mma_b.size(1);
    TORCH_CHECK(mma_b.size(0) == F && mma_c.size(0) == M && mma_c.size(1) == N2, "shape mismatch");
    c10::DeviceGuard guard(mma_a.device());
    check(chipmunk_csp_mlp_mm2(mma_a.data_ptr(), mma_b.data_ptr(), mma_c.data_ptr(), indices.data_ptr<int>(),
                               counts.data_ptr<int>(), (int)M, (int)F, (int)N2, cur_stream(mma_a)),
          "csp_mlp_mm2");
}

// ---------------------------------------------------------------------------------- indexed IO
// reference csrc/indexed_io/copy_indices.cu:82-154
void copy_indices(at::Tensor bmfc1, at::Tensor bm_mid_cache, at::Tensor sp_inds, at::Tensor sp_counts) {
    CHECK_DEV(bmfc1); CHECK_DEV(bm_mid_cache); CHECK_DEV(sp_inds); CHECK_DEV(sp_counts);
    CHECK_CONTIG(bmfc1); CHECK_CONTIG(bm_mid_cache); CHECK_CONTIG(sp_inds); CHECK_CONTIG(sp_counts);
    TORCH_CHECK(bmfc1.dim() == 3 && bm_mid_cache.dim() == 3, "bmfc1 and bm_mid_cache must be 3D [B, M*R, F]");
    TORCH_CHECK(sp_inds.dim() == 3 && sp_counts.dim() == 2, "sp_inds must be [B, M, F] and sp_counts [B, M]");
    CHECK_I32(sp_inds); CHECK_I32(sp_counts);
    TORCH_CHECK(bmfc1.scalar_type() == bm_mid_cache.scalar_type(), "bmfc1 and bm_mid_cache must have the same dtype");
    TORCH_CHECK(bmfc1.sizes() == bm_mid_cache.sizes(), "bmfc1 and bm_mid_cache must have the same shape");
    const auto st = bmfc1.scalar_type();
    TORCH_CHECK(st == at::kBFloat16 || st == at::kHalf || st == at::kFloat, "Unsupported dtype for copy_indices");
    const int64_t B = bmfc1.size(0), MR = bmfc1.size(1), F = bmfc1.size(2), M = sp_counts.size(1);
    TORCH_CHECK(M > 0 && MR % M == 0 && sp_inds.size(1) == M && sp_inds.size(2) == F, "inconsistent shapes");
    c10::DeviceGuard guard(bmfc1.device());
    check(chipmunk_copy_indices(bmfc1.data_ptr(), bm_mid_cache.data_ptr(), sp_inds.data_ptr<int>(),
                                sp_counts.data_ptr<int>(), (int)B, (int)M, (int)(MR / M), (int)F,
                                (int)bmfc1.element_size(), cur_stream(bmfc1)),
          "copy_indices");
}

// reference csrc/indexed_io/topk_indices.cu:145-218
void topk_indices(at::Tensor activation, at::Tensor indices, at::Tensor counts, double sparsity_amount,
                  int64_t multiple_of, double random_amount) {
    CHECK_DEV(activation); CHECK_DEV(indices); CHECK_DEV(counts);
    CHECK_CONTIG(activation); CHECK_CONTIG(indices); CHECK_CONTIG(counts);
    TORCH_CHECK(activation.dim() == 3, "activation must be [batch, rows, cols]");
    CHECK_I32(indices); CHECK_I32(counts);
    TORCH_CHECK(indices.sizes() == activation.sizes(), "indices must have the shape of activation");
    TORCH_CHECK(counts.numel() == activation.size(0) * activation.size(1), "counts must be [batch, rows]");
    int dtype;
    switch (activation.scalar_type()) {
        case at::kBFloat16: dtype = CHIPMUNK_DTYPE_BF16; break;
        case at::kHalf: dtype = CHIPMUNK_DTYPE_FP16; break;
        case at::kFloat: dtype = CHIPMUNK_DTYPE_FP32; break;
        default: TORCH_CHECK(false, "Unsupported dtype for topk_indices");
    }
    c10::DeviceGuard guard(activation.device());
    check(chipmunk_topk_indices(activation.data_ptr(), dtype, indices.data_ptr<int>(), counts.data_ptr<int>(),
                                (int)(activation.size(0) * activation.size(1)), (int)activation.size(2),
                                sparsity_amount, (int)multiple_of, random_amount, cur_stream(activation)),
          "topk_indices");
}

// fused |activation - cache| -> topk_indices -> copy_indices (reference modules/mlp.py:70-85 with bm == mbm)
void topk_delta_indices(at::Tensor activation, at::Tensor cache, at::Tensor indices, at::Tensor counts,
                        double sparsity_amount, int64_t multiple_of, double random_amount) {
    CHECK_DEV(activation); CHECK_DEV(cache); CHECK_DEV(indices); CHECK_DEV(counts);
    CHECK_CONTIG(activation); CHECK_CONTIG(cache); CHECK_CONTIG(indices); CHECK_CONTIG(counts);
    TORCH_CHECK(activation.dim() == 3 && cache.sizes() == activation.sizes(), "activation and cache must be [batch, rows, cols]");
    TORCH_CHECK(cache.scalar_type() == activation.scalar_type(), "activation and cache must have the same dtype");
    CHECK_I32(indices); CHECK_I32(counts);
    TORCH_CHECK(indices.sizes() == activation.sizes(), "indices must have the shape of activation");
    TORCH_CHECK(counts.numel() == activation.size(0) * activation.size(1), "counts must be [batch, rows]");
    int dtype;
    switch (activation.scalar_type()) {
        case at::kBFloat16: dtype = CHIPMUNK_DTYPE_BF16; break;
        case at::kHalf: dtype = CHIPMUNK_DTYPE_FP16; break;
        case at::kFloat: dtype = CHIPMUNK_DTYPE_FP32; break;
        default: TORCH_CHECK(false, "Unsupported dtype for topk_delta_indices");
    }
    c10::DeviceGuard guard(activation.device());
    check(chipmunk_topk_delta_indices(activation.data_ptr(), cache.data_ptr(), dtype, indices.data_ptr<int>(),
                                      counts.data_ptr<int>(), (int)(activation.size(0) * activation.size(1)),
                                      (int)activation.size(2), sparsity_amount, (int)multiple_of, random_amount,
                                      cur_stream(activation)),
          "topk_delta_indices");
}

// reference csrc/indexed_io/mask_to_indices.cu:92-143
std::vector<at::Tensor> mask_to_indices(at::Tensor mask, int64_t multiple_of, int64_t pad_to_multiple_of) {
    TORCH_CHECK(mask.dim() == 4, "mask must be 4-dimensional [b, h, m, n]");
    TORCH_CHECK(mask.scalar_type() == at::kBool, "mask must be bool type");
    CHECK_DEV(mask);
    TORCH_CHECK(multiple_of > 0 && pad_to_multiple_of > 0, "multiple_of and pad_to_multiple_of must be positive");
    mask = mask.contiguous();
    const int64_t b = mask.size(0), h = mask.size(1), m = mask.size(2), n = mask.size(3);
    const int64_t pad_n = ((n + pad_to_multiple_of - 1) / pad_to_multiple_of) * pad_to_multiple_of;
    c10::DeviceGuard guard(mask.device());
    at::Tensor indices = at::empty({b, h, m, pad_n}, mask.options().dtype(at::kInt));
    at::Tensor counts = at::empty({b, h, m}, mask.options().dtype(at::kInt));
    check(chipmunk_mask_to_indices(mask.data_ptr(), indices.data_ptr<int>(), counts.data_ptr<int>(), b * h * m, (int)n,
                                   (int)pad_n, (int)multiple_of, cur_stream(mask)),
          "mask_to_indices");
    return {indices, counts};
}

// fused bitunpack + mask_to_indices: `packed` is ops.bitpack's output for a [b,h,m,n] mask
std::vector<at::Tensor> packed_mask_to_indices(at::Tensor packed, at::IntArrayRef shape, int64_t multiple_of,
                                               int64_t pad_to_multiple_of) {
    TORCH_CHECK(packed.scalar_type() == at::kByte && packed.is_contiguous(), "packed must be a contiguous uint8 tensor");
    CHECK_DEV(packed);
    TORCH_CHECK(shape.size() == 4, "shape must be [b, h, m, n]");
    const int64_t b = shape[0], h = shape[1], m = shape[2], n = shape[3];
    TORCH_CHECK(n % 8 == 0, "n must be a multiple of 8");
    TORCH_CHECK(packed.numel() * 8 >= b * h * m * n, "packed is too short for the given shape");
    const int64_t pad_n = ((n + pad_to_multiple_of - 1) / pad_to_multiple_of) * pad_to_multiple_of;
    c10::DeviceGuard guard(packed.device());
    at::Tensor indices = at::empty({b, h, m, pad_n}, packed.options().dtype(at::kInt));
    at::Tensor counts = at::empty({b, h, m}, packed.options().dtype(at::kInt));
    check(chipmunk_packed_mask_to_indices(packed.data_ptr(), indices.data_ptr<int>(), counts.data_ptr<int>(),
                                          b * h * m, (int)n, (int)pad_n, (int)multiple_of, cur_stream(packed)),
          "packed_mask_to_indices");
    return {indices, counts};
}

// x.transpose(-1, -2).contiguous() for 16-bit dtypes in one HBM-rate kernel (reference modules/mlp.py:56)
at::Tensor transpose_last2(at::Tensor x) {
    CHECK_DEV(x);
    TORCH_CHECK(x.dim() >= 2 && x.element_size() == 2, "transpose_last2: need a >=2-D tensor of a 16-bit dtype");
    x = x.contiguous();
    const int64_t R = x.size(-2), C = x.size(-1), B = x.numel() / (R * C);
    auto sizes = x.sizes().vec();
    std::swap(sizes[sizes.size() - 1], sizes[sizes.size() - 2]);
    c10::DeviceGuard guard(x.device());
    at::Tensor out = at::empty(sizes, x.options());
    check(chipmunk_transpose16(x.data_ptr(), out.data_ptr(), (int)B, (int)R, (int)C, cur_stream(x)), "transpose_last2");
    return out;
}

// [b, n, c] -> [b, n / mbm, c] mean over consecutive row blocks (reference modules/mlp.py:11-16) in one HBM-rate kernel, bf16
at::Tensor block_mean(at::Tensor x, int64_t mbm) {
    CHECK_DEV(x);
    TORCH_CHECK(x.dim() == 3 && x.scalar_type() == at::kBFloat16, "block_mean: need a [b, n, c] bfloat16 tensor");
    TORCH_CHECK(mbm > 0 && x.size(1) % mbm == 0, "block_mean: n must be a multiple of mbm");
    x = x.contiguous();
    c10::DeviceGuard guard(x.device());
    at::Tensor out = at::empty({x.size(0), x.size(1) / mbm, x.size(2)}, x.options());
    check(chipmunk_block_mean(x.data_ptr(), out.data_ptr(), x.size(0) * x.size(1), (int)x.size(2), (int)mbm, cur_stream(x)), "block_mean");
    return out;
}

// F8Linear.quantize_input as one kernel: (x * scale).clamp(-max, max).to(float8_e4m3fn), same roundings
at::Tensor quantize_fp8(at::Tensor x, at::Tensor scale, double max_value) {
    CHECK_DEV(x); CHECK_DEV(scale);
    TORCH_CHECK(x.scalar_type() == at::kBFloat16, "quantize_fp8: x must be bfloat16");
    TORCH_CHECK(scale.scalar_type() == at::kFloat && scale.numel() == 1, "quantize_fp8: scale must be a one-element float32 tensor");
    TORCH_CHECK(x.numel() % 8 == 0, "quantize_fp8: the element count must be a multiple of 8");
    x = x.contiguous();
    c10::DeviceGuard guard(x.device());
    at::Tensor out = at::empty(x.sizes(), x.options().dtype(at::kFloat8_e4m3fn));
    check(chipmunk_quantize_fp8(x.data_ptr(), scale.data_ptr<float>(), out.data_ptr(), x.numel(), (float)max_value, cur_stream(x)), "quantize_fp8");
    return out;
}

// ascending-order variant of (packed_)mask_to_indices (same set / counts / padding; see chipmunk_hip.h)
std::vector<at::Tensor> mask_to_sorted_indices(at::Tensor mask, at::IntArrayRef shape, int64_t multiple_of,
                                               int64_t pad_to_multiple_of) {
    CHECK_DEV(mask);
    const bool packed = mask.scalar_type() == at::kByte;
    TORCH_CHECK(packed || mask.scalar_type() == at::kBool, "mask must be bool, or uint8 bit-packed with a shape");
    mask = mask.contiguous();
    std::vector<int64_t> shp = packed ? shape.vec() : mask.sizes().vec();
    TORCH_CHECK(shp.size() == 4, "shape must be [b, h, m, n]");
    const int64_t b = shp[0], h = shp[1], m = shp[2], n = shp[3];
    if (packed) TORCH_CHECK(n % 8 == 0 && mask.numel() * 8 >= b * h * m * n, "bad packed mask");
    const int64_t pad_n = ((n + pad_to_multiple_of - 1) / pad_to_multiple_of) * pad_to_multiple_of;
    c10::DeviceGuard guard(mask.device());
    at::Tensor indices = at::empty({b, h, m, pad_n}, mask.options().dtype(at::kInt));
    at::Tensor counts = at::empty({b, h, m}, mask.options().dtype(at::kInt));
    check(chipmunk_mask_to_sorted_indices(mask.data_ptr(), packed ? 1 : 0, indices.data_ptr<int>(),
                                          counts.data_ptr<int>(), b * h * m, (int)n, (int)pad_n, (int)multiple_of,
                                          cur_stream(mask)),
          "mask_to_sorted_indices");
    return {indices, counts};
}

// addition: randint + topk + scatter_ + the two mask combines of modules/attn.py:76-82 in one kernel
// dense_colsum_attn + topk_mask without the cs tensor between them (see chipmunk_dense_colsum_topk_mask); falls back to the two
// operators when the fused entry does not apply to the launch
std::vector<at::Tensor> dense_colsum_attn_layout(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor p, bool token_major_o);
at::Tensor topk_mask(at::Tensor cs, int64_t k, double random_amount, const c10::optional<at::Tensor> &groups,
                     const c10::optional<at::Tensor> &static_mask);
std::vector<at::Tensor> dense_colsum_topk_mask(at::Tensor q, at::Tensor k, at::Tensor v, at::Tensor p, int64_t k_top, double random_amount,
                                               const c10::optional<at::Tensor> &groups, const c10::optional<at::Tensor> &static_mask,
                                               bool token_major_o) {
    CHECK_DEV(q); CHECK_DEV(k); CHECK_DEV(v); CHECK_DEV(p);
    CHECK_BF16(q); CHECK_BF16(k); CHECK_BF16(v);
    const bool shapes_ok = q.dim() == 4 && k.dim() == 4 && v.dim() == 4 && q.size(3) == 128 && q.stride(3) == 1 && k.stride(3) == 1 &&
                           v.stride(3) == 1 && p.scalar_type() == at::kFloat && p.is_contiguous();
    if (shapes_ok) {
        const int64_t B = q.size(0), H = q.size(1), Nq = q.size(2), Nk = k.size(2), G = (Nq + 191) / 192;
        const void *st = nullptr, *gf = nullptr;
        int64_t st_stride = 0, st_rows = 1;
        at::Tensor stc, gfc;
        bool ok = p.numel() == B * H * Nq;
        if (ok && static_mask.has_value() && static_mask->defined()) {
            stc = *static_mask;
            ok = stc.is_cuda() && stc.scalar_type() == at::kBool && stc.dim() == 4 && stc.size(3) == Nk && stc.size(2) == G && stc.size(1) == H &&
                 (stc.size(0) == 1 || stc.size(0) == B) && stc.stride(3) == 1 && stc.stride(1) == G * stc.stride(2) &&
                 (stc.size(0) == 1 || stc.stride(0) == H * stc.stride(1));
            if (ok) st = stc.data_ptr(), st_stride = stc.stride(2), st_rows = stc.size(0) * H * G;
        }
        if (ok && groups.has_value() && groups->defined()) {
            gfc = *groups;
            ok = gfc.is_cuda() && gfc.scalar_type() == at::kBool && gfc.dim() == 4 && gfc.size(3) == 1 && gfc.size(2) == G && gfc.size(1) == H &&
                 (gfc.size(0) == 1 || gfc.size(0) == B);
            if (ok) {
                gfc = gfc.expand({B, H, G, 1}).contiguous();
                gf = gfc.data_ptr();
            }
        }
        if (ok) {
            c10::DeviceGuard guard(q.device());
            at::Tensor o = alloc_o(q, q, token_major_o);
            const int64_t os[3] = {o.stride(0), o.stride(1), o.stride(2)};
            at::Tensor l = at::empty({B, H, Nq, 1}, q.options().dtype(at::kFloat));
            at::Tensor mask = at::empty({B, H, G, Nk}, q.options().dtype(at::kBool));
            const int64_t qs[3] = {q.stride(0), q.stride(1), q.stride(2)}, ks[3] = {k.stride(0), k.stride(1), k.stride(2)},
                          vs[3] = {v.stride(0), v.stride(1), v.stride(2)};
            const int rc = chipmunk_dense_colsum_topk_mask_strided(q.data_ptr(), k.data_ptr(), v.data_ptr(), qs, ks, vs, p.data_ptr<float>(), o.data_ptr(),
                                                           os, l.data_ptr<float>(), (int)B, (int)H, (int)Nq, (int)Nk, st, st_stride, (int)st_rows, gf,
                                                           mask.data_ptr(), (int)k_top, random_amount, cur_stream(q));
            if (rc == CHIPMUNK_OK) return {o, mask, l};
            if (rc != CHIPMUNK_ERR_UNSUPPORTED) check(rc, "dense_colsum_topk_mask");
        }
    }
    auto ocl = dense_colsum_attn_layout(q, k, v, p, token_major_o);
    at::Tensor cs = ocl[1];
    const int64_t Nk = k.size(2), G = (q.size(2) + 191) / 192;
    if (cs.size(-1) != Nk || cs.size(-2) != G) cs = cs.slice(-2, 0, G).slice(-1, 0, Nk);
    return {ocl[0], topk_mask(cs, k_top, random_amount, groups, static_mask), ocl[2]};
}

at::Tensor topk_mask(at::Tensor cs, int64_t k, double random_amount, const c10::optional<at::Tensor> &groups,
                     const c10::optional<at::Tensor> &static_mask) {
    CHECK_DEV(cs); CHECK_BF16(cs);
    TORCH_CHECK(cs.dim() == 4 && cs.stride(3) == 1, "cs must be [B, H, G, N] with a contiguous last dim");
    const int64_t B = cs.size(0), H = cs.size(1), G = cs.size(2), N = cs.size(3);
    // rows must be addressable with one stride: collapse (B, H, G) or copy
    if (!(cs.stride(1) == G * cs.stride(2) && (B == 1 || cs.stride(0) == H * cs.stride(1)))) cs = cs.contiguous();
    const int64_t rows = B * H * G;
    c10::DeviceGuard guard(cs.device());
    at::Tensor mask = at::empty({B, H, G, N}, cs.options().dtype(at::kBool));
    const void *st = nullptr, *gf = nullptr;
    int64_t st_stride = 0, st_rows = 1;
    at::Tensor stc, gfc;
    if (static_mask.has_value() && static_mask->defined()) {
        stc = *static_mask;
        CHECK_DEV(stc);
        TORCH_CHECK(stc.scalar_type() == at::kBool && stc.dim() == 4 && stc.size(3) == N && stc.size(2) == G &&
                    stc.size(1) == H && (stc.size(0) == 1 || stc.size(0) == B), "static_mask must be bool [1|B, H, G, N]");
        if (!(stc.stride(3) == 1 && stc.stride(1) == G * stc.stride(2) && (stc.size(0) == 1 || stc.stride(0) == H * stc.stride(1))))
            stc = stc.contiguous();
        st = stc.data_ptr(), st_stride = stc.stride(2), st_rows = stc.size(0) * H * G;
    }
    if (groups.has_value() && groups->defined()) {
        gfc = *groups;
        CHECK_DEV(gfc);
        TORCH_CHECK(gfc.scalar_type() == at::kBool && gfc.dim() == 4 && gfc.size(3) == 1 && gfc.size(2) == G &&
                    gfc.size(1) == H && (gfc.size(0) == 1 || gfc.size(0) == B), "groups must be bool [1|B, H, G, 1]");
        gfc = gfc.expand({B, H, G, 1}).contiguous();
        gf = gfc.data_ptr();
    }
    check(chipmunk_topk_mask(cs.data_ptr(), cs.stride(2), st, st_stride, (int)st_rows, gf, mask.data_ptr(), (int)rows, (int)N,
                             (int)k, random_amount, cur_stream(cs)),
          "topk_mask");
    return mask;
}

// reference src/chipmunk/ops/bitpack.py:4-69 as single kernels
at::Tensor bitpack(at::Tensor mask) {
    TORCH_CHECK(mask.scalar_type() == at::kBool, "mask must be bool type");
    CHECK_DEV(mask);
    mask = mask.contiguous();
    c10::DeviceGuard guard(mask.device());
    at::Tensor packed = at::empty({(mask.numel() + 7) / 8}, mask.options().dtype(at::kByte));
    check(chipmunk_bitpack(mask.data_ptr(), packed.data_ptr(), mask.numel(), cur_stream(mask)), "bitpack");
    return packed;
}
at::Tensor bitunpack(at::Tensor packed, at::IntArrayRef shape) {
    TORCH_CHECK(packed.scalar_type() == at::kByte && packed.is_contiguous(), "packed must be a contiguous uint8 tensor");
    CHECK_DEV(packed);
    int64_t n = 1;
    for (auto d : shape) n *= d;
    TORCH_CHECK(packed.numel() * 8 >= n, "packed is too short for the given shape");
    c10::DeviceGuard guard(packed.device());
    at::Tensor mask = at::empty(shape, packed.options().dtype(at::kBool));
    check(chipmunk_bitunpack(packed.data_ptr(), mask.data_ptr(), n, cur_stream(packed)), "bitunpack");
    return mask;
}

// [n, >= 3*heads*128] projection output -> q, k, v [1, heads, n, 128] with q/k RMSNorm (see chipmunk_qkv_split_norm)
std::vector<at::Tensor> qkv_split_norm(at::Tensor qkv, const c10::optional<at::Tensor> &q_weight, const c10::optional<at::Tensor> &k_weight,
                                       int64_t heads, double eps, const c10::optional<at::Tensor> &freqs_cos,
                                       const c10::optional<at::Tensor> &freqs_sin) {
    CHECK_DEV(qkv); CHECK_BF16(qkv);
    TORCH_CHECK(qkv.dim() == 2 && qkv.stride(1) == 1 && qkv.size(1) >= 3 * heads * 128, "qkv_split_norm: qkv must be [n, >= 3*heads*128] with a contiguous last dim");
    const void *qw = nullptr, *kw = nullptr;
    at::Tensor qwc, kwc;
    if (q_weight.has_value() && q_weight->defined()) { qwc = q_weight->contiguous(); CHECK_DEV(qwc); CHECK_BF16(qwc); TORCH_CHECK(qwc.numel() == 128); qw = qwc.data_ptr(); }
    if (k_weight.has_value() && k_weight->defined()) { kwc = k_weight->contiguous(); CHECK_DEV(kwc); CHECK_BF16(kwc); TORCH_CHECK(kwc.numel() == 128); kw = kwc.data_ptr(); }
    const int64_t n = qkv.size(0);
    const float *fc = nullptr, *fs = nullptr;
    int64_t rope_rows = 0;
    at::Tensor fcc, fsc;
    if (freqs_cos.has_value() && freqs_cos->defined()) {
        TORCH_CHECK(freqs_sin.has_value() && freqs_sin->defined(), "qkv_split_norm: freqs_cos without freqs_sin");
        fcc = freqs_cos->contiguous(), fsc = freqs_sin->contiguous();
        CHECK_DEV(fcc); CHECK_DEV(fsc);
        TORCH_CHECK(fcc.scalar_type() == at::kFloat && fsc.scalar_type() == at::kFloat && fcc.dim() == 2 && fcc.size(1) == 128 &&
                    fsc.sizes() == fcc.sizes() && fcc.size(0) <= n, "qkv_split_norm: freqs must be float32 [rows <= n, 128]");
        fc = fcc.data_ptr<float>(), fs = fsc.data_ptr<float>(), rope_rows = fcc.size(0);
    }
    c10::DeviceGuard guard(qkv.device());
    at::Tensor out = at::empty({3, 1, heads, n, 128}, qkv.options());
    check(chipmunk_qkv_split_norm(qkv.data_ptr(), qkv.stride(0), qw, kw, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), n,
                                  (int)heads, (float)eps, fc, fs, rope_rows, cur_stream(qkv)),
          "qkv_split_norm");
    return {out[0], out[1], out[2]};
}

// dst[..., i, :] = src[..., map[i], :] over the second-to-last axis (token reorder; see chipmunk_gather_rows)
at::Tensor gather_rows(at::Tensor src, at::Tensor map) {
    CHECK_DEV(src);
    CHECK_DEV(map);
    TORCH_CHECK(src.dim() >= 2, "gather_rows: src must be [..., n, d]");
    TORCH_CHECK(map.scalar_type() == at::kInt && map.dim() == 1 && map.is_contiguous(), "gather_rows: map must be a contiguous int32 vector");
    src = src.contiguous();
    const int64_t n_src = src.size(-2), d = src.size(-1), n_out = map.numel();
    const int64_t outer = n_src * d == 0 ? 0 : src.numel() / (n_src * d);
    auto sizes = src.sizes().vec();
    sizes[sizes.size() - 2] = n_out;
    c10::DeviceGuard guard(src.device());
    at::Tensor dst = at::empty(sizes, src.options());
    if (dst.numel() == 0) return dst;
    check(chipmunk_gather_rows(src.data_ptr(), dst.data_ptr(), map.data_ptr<int>(), outer, n_src, n_out,
                               d * (int64_t)src.element_size(), cur_stream(src)),
          "gather_rows");
    return dst;
}

}  // namespace

// schemas: byte-identical to reference csrc/chipmunk.cpp:47-60 (including its un-annotated mutations of `o`,
// `mma_c` and `counts`), followed by this build's additions.
TORCH_LIBRARY(chipmunk, m) {
    m.def("csp_mlp_mm1(Tensor a, Tensor b_colmajor, Tensor(c!) c, Tensor bias, Tensor pa_cache_colmajor, Tensor indices, Tensor indices_counts) -> ()");
    m.def("csp_mlp_mm2_and_scatter_add(Tensor packed, Tensor(unpacked_colmajor!) unpacked_colmajor, Tensor sp_inds, Tensor sp_counts, Tensor mma_a, Tensor mma_b, Tensor mma_c, int num_sms_scatter_add, int matmul_kernel) -> ()");

    m.def("csp_attn(Tensor q, Tensor k, Tensor v, Tensor o, Tensor indices, Tensor indices_counts, int o_scale) -> ()");
    m.def("csp_128_attn(Tensor q, Tensor k, Tensor v, Tensor indices, Tensor indices_counts) -> Tensor");
    m.def("dense_attn(Tensor q, Tensor k, Tensor v) -> Tensor[]");
    m.def("dense_colsum_attn(Tensor q, Tensor k, Tensor v, Tensor p) -> Tensor[]");

    m.def("copy_indices(Tensor bmfc1, Tensor(bm_mid_cache!) bm_mid_cache, Tensor sp_inds, Tensor sp_counts) -> ()");
    m.def("topk_indices(Tensor activation, Tensor(indices!) indices, Tensor counts, float sparsity_amount, int multiple_of, float random_amount) -> ()");
    m.def("csp_scatter_add(Tensor packed, Tensor(unpacked_colmajor!) unpacked_colmajor, Tensor sp_inds, Tensor sp_counts, int num_sms) -> ()");
    m.def("mask_to_indices(Tensor mask, int multiple_of, int pad_to_multiple_of) -> Tensor[]");

    // additions (not in the reference): native GEMM2 entry, fused packed-mask path, single-kernel bit packing
    m.def("csp_attn_out(Tensor q, Tensor k, Tensor v, Tensor o_in, Tensor indices, Tensor indices_counts, int o_scale) -> Tensor");
    m.def("csp_mlp_mm1_scatter(Tensor a, Tensor b_colmajor, Tensor(c!) c, Tensor bias, Tensor(pa_cache_colmajor!) pa_cache_colmajor, Tensor indices, Tensor indices_counts) -> ()");
    m.def("csp_mlp_mm2(Tensor mma_a, Tensor mma_b, Tensor indices, Tensor counts, Tensor(mma_c!) mma_c) -> ()");
    m.def("csp_mlp_mm1_fp8(Tensor a, Tensor b, Tensor(c!) c, Tensor bias, Tensor(pa_cache_colmajor!) pa_cache_colmajor, Tensor indices, Tensor indices_counts, Tensor scale_a, Tensor scale_b, bool update_cache) -> ()");
    m.def("csp_mlp_mm1_fp8_scatter(Tensor a, Tensor b, Tensor(c!) c, Tensor bias, Tensor(pa_cache_colmajor!) pa_cache_colmajor, Tensor indices, Tensor indices_counts, Tensor scale_a, Tensor scale_b) -> ()");
    m.def("topk_delta_indices(Tensor activation, Tensor(cache!) cache, Tensor(indices!) indices, Tensor(counts!) counts, float sparsity_amount, int multiple_of, float random_amount) -> ()");
    m.def("packed_mask_to_indices(Tensor packed, int[] shape, int multiple_of, int pad_to_multiple_of) -> Tensor[]");
    m.def("mask_to_sorted_indices(Tensor mask, int[] shape, int multiple_of, int pad_to_multiple_of) -> Tensor[]");
    m.def("topk_mask(Tensor cs, int k, float random_amount, Tensor? groups, Tensor? static_mask) -> Tensor");
    m.def("dense_colsum_topk_mask(Tensor q, Tensor k, Tensor v, Tensor p, int k_top, float random_amount, Tensor? groups, Tensor? static_mask, bool token_major_o=False) -> Tensor[]");
    // the two dense operators with a choice of output layout (the reference's schemas above stay as they are)
    m.def("dense_attn_layout(Tensor q, Tensor k, Tensor v, bool token_major_o) -> Tensor[]");
    m.def("dense_colsum_attn_layout(Tensor q, Tensor k, Tensor v, Tensor p, bool token_major_o) -> Tensor[]");
    m.def("compact_indices(Tensor indices, Tensor counts) -> Tensor[]");
    m.def("csp_attn_out_ragged(Tensor q, Tensor k, Tensor v, Tensor o_in, Tensor indices, Tensor offsets, Tensor indices_counts, int o_scale) -> Tensor");
    m.def("residual_ln_modulate(Tensor x, Tensor? y, Tensor? gate, Tensor shift, Tensor scale, float eps) -> Tensor[]");
    m.def("transpose_last2(Tensor x) -> Tensor");
    m.def("block_mean(Tensor x, int mbm) -> Tensor");
    m.def("quantize_fp8(Tensor x, Tensor scale, float max_value) -> Tensor");
    m.def("bitpack(Tensor mask) -> Tensor");
    m.def("bitunpack(Tensor packed, int[] shape) -> Tensor");
    m.def("gather_rows(Tensor src, Tensor map) -> Tensor");
    m.def("qkv_split_norm(Tensor qkv, Tensor? q_weight, Tensor? k_weight, int heads, float eps, Tensor? freqs_cos=None, Tensor? freqs_sin=None) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(chipmunk, CUDA, m) {
    m.impl("qkv_split_norm", &qkv_split_norm);
    m.impl("csp_mlp_mm1_fp8_scatter", &csp_mlp_mm1_fp8_scatter);
    m.impl("dense_colsum_topk_mask", &dense_colsum_topk_mask);
    m.impl("csp_mlp_mm1", &csp_mlp_mm1);
    m.impl("csp_mlp_mm2_and_scatter_add", &csp_mlp_mm2_and_scatter_add);
    m.impl("copy_indices", &copy_indices);
    m.impl("topk_indices", &topk_indices);
    m.impl("csp_scatter_add", &csp_scatter_add);
    m.impl("mask_to_indices", &mask_to_indices);
    m.impl("csp_attn", &csp_attn);
    m.impl("csp_128_attn", &csp_128_attn);
    m.impl("dense_attn", &dense_attn);
    m.impl("dense_colsum_attn", &dense_colsum_attn);
    m.impl("compact_indices", &compact_indices);
    m.impl("residual_ln_modulate", &residual_ln_modulate);
    m.impl("csp_attn_out_ragged", &csp_attn_out_ragged);
    m.impl("dense_attn_layout", &dense_attn_layout);
    m.impl("dense_colsum_attn_layout", &dense_colsum_attn_layout);
    m.impl("csp_attn_out", &csp_attn_out);
    m.impl("csp_mlp_mm1_scatter", &csp_mlp_mm1_scatter);
    m.impl("csp_mlp_mm2", &csp_mlp_mm2);
    m.impl("csp_mlp_mm1_fp8", &csp_mlp_mm1_fp8);
    m.impl("topk_delta_indices", &topk_delta_indices);
    m.impl("packed_mask_to_indices", &packed_mask_to_indices);
    m.impl("mask_to_sorted_indices", &mask_to_sorted_indices);
    m.impl("topk_mask", &topk_mask);
    m.impl("transpose_last2", &transpose_last2);
    m.impl("block_mean", &block_mean);
    m.impl("quantize_fp8", &quantize_fp8);
    m.impl("bitpack", &bitpack);
    m.impl("bitunpack", &bitunpack);
    m.impl("gather_rows", &gather_rows);
}

}  // namespace chipmunk
