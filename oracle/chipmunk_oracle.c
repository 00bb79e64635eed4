/*
 * chipmunk_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the algorithms of the reference's column-sparse DiT hot path
 * (sandyresearch/chipmunk @ 2025-05-23).  Every function cites the reference file:line it
 * follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (chipmunk_amd/) never links, imports or calls it.
 *
 * The reference's own kernels cannot be compiled in this environment (they need nvcc,
 * sm_90a and the un-vendored ThunderKittens submodule: setup.py:78-103), so there is no
 * oracle/_ref build.  This oracle is pinned instead against the formulas the reference's own
 * tests use (tests/test_oracle_pins.py): SDPA under identity indices
 * (src/chipmunk/tests/test_csp_attn.py:30-38, test_dense_attn.py:29-36), the fp32 column-sum
 * formula (test_dense_colsum_attn.py:13-36) and the mm1 known-answer recipe
 * (csrc/mlp/csp_mlp_mm1.cu:401-424,458-486).  Ops with no reference test at all
 * (mask_to_indices, topk_indices, copy_indices, scatter_add, mm2) are "parity unpinned by the
 * reference": the restatement of the cited kernel source is the only pin.
 *
 * All bf16 tensors are passed as uint16_t bit patterns.  Arithmetic: fp32 with explicit bf16
 * rounding (round-to-nearest-even) at the points where the reference rounds.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ bf16 helpers */
static inline float bf2f(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u); /* NaN */
    uint32_t lsb = (u >> 16) & 1u;
    u += 0x7fffu + lsb;
    return (uint16_t)(u >> 16);
}
static inline float rbf(float f) { return bf2f(f2bf(f)); }

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* softmax temperature in the exp2 domain: 1/sqrt(128) * log2(e)
 * (csrc/attn/csp_128_attn.cu:307, csp_attn.cu:264, dense_attn.cu:160) */
#define TEMPERATURE_SCALE (0.08838834764f * 1.44269504089f)
#define HEAD_DIM 128
#define QGROUP 192

typedef struct {
    const uint16_t *q, *k, *v;
    int64_t qs[3], ks[3], vs[3]; /* element strides of dims (b, h, n); last dim contiguous */
    int B, H, Nq, Nk;
} attn_in_t;

/*
 * One 192-row query group of the online-softmax loop shared by all four attention kernels.
 *   csp_128_attn.cu:306-341 / csp_attn.cu:264-302 (gathered keys, kv tile 128 / 112)
 *   dense_attn.cu:156-233 (all keys, kv tile 128, exports l)
 *   dense_colsum_attn.cu:219-277 (adds the 192-row column sums)
 * idx == NULL means identity (dense).  Packed positions >= Nk are masked to -inf exactly like
 * `right_fill(att_block, K.rows - iter*kv_tile)` (csp_128_attn.cu:314).
 * out_o: fp32 [rows][128] = o_reg / norm_vec;  out_l: 1 / (exp2(m*c) * norm)  (dense_attn.cu:225-227)
 * prev_l / cs: colsum inputs/outputs (NULL when unused); cs has `count` entries (bf16 bits).
 */
static void attn_group(const attn_in_t *in, int b, int h, int row0, int rows, const int32_t *idx, int count,
                       int kv_tile, float *out_o, float *out_l, const float *prev_l, uint16_t *cs) {
    const float c = TEMPERATURE_SCALE;
    float *m = (float *)malloc(sizeof(float) * rows);
    float *l = (float *)malloc(sizeof(float) * rows);
    float *s = (float *)malloc(sizeof(float) * kv_tile);
    float *colacc = cs ? (float *)malloc(sizeof(float) * 12 * kv_tile) : NULL; /* 12 consumer warps x tile */
    for (int r = 0; r < rows; ++r) {
        m[r] = -INFINITY;
        l[r] = 0.f;
        for (int d = 0; d < HEAD_DIM; ++d) out_o[(size_t)r * HEAD_DIM + d] = 0.f;
    }
    const uint16_t *kb = in->k + (int64_t)b * in->ks[0] + (int64_t)h * in->ks[1];
    const uint16_t *vb = in->v + (int64_t)b * in->vs[0] + (int64_t)h * in->vs[1];
    const uint16_t *qb = in->q + (int64_t)b * in->qs[0] + (int64_t)h * in->qs[1];
    int ntiles = (count + kv_tile - 1) / kv_tile;
    for (int t = 0; t < ntiles; ++t) {
        int j0 = t * kv_tile;
        int jn = kv_tile;
        if (colacc) memset(colacc, 0, sizeof(float) * 12 * kv_tile);
        for (int r = 0; r < rows; ++r) {
            int qi = row0 + r;
            int q_valid = qi < in->Nq;
            const uint16_t *qrow = qb + (int64_t)(q_valid ? qi : 0) * in->qs[2];
            float qf[HEAD_DIM];
            for (int d = 0; d < HEAD_DIM; ++d) qf[d] = q_valid ? bf2f(qrow[d]) : 0.f; /* TMA zero-fills OOB rows */
            float mx = m[r];
            for (int j = 0; j < jn; ++j) {
                int p = j0 + j; /* packed position */
                if (p >= count || p >= in->Nk) {
                    s[j] = -INFINITY;
                    continue;
                }
                int key = idx ? idx[p] : p;
                const uint16_t *krow = kb + (int64_t)key * in->ks[2];
                float acc = 0.f;
                for (int d = 0; d < HEAD_DIM; ++d) acc += qf[d] * bf2f(krow[d]);
                s[j] = acc;
                if (acc > mx) mx = acc;
            }
            float m_old_scaled = m[r] * c;
            float m_scaled = mx * c;
            float alpha = exp2f(m_old_scaled - m_scaled); /* exp2(-inf) = 0 on the first tile */
            if (mx == -INFINITY) alpha = 1.f;             /* fully masked so far: keep zeros */
            float rowsum = 0.f;
            float *orow = out_o + (size_t)r * HEAD_DIM;
            for (int d = 0; d < HEAD_DIM; ++d) orow[d] *= alpha;
            /* colsum row factor: bf16(exp2(m_run*c) * prev_l_i)  (dense_colsum_attn.cu:268-271) */
            float rowfac = 0.f;
            if (cs) rowfac = rbf(exp2f(m_scaled) * prev_l[r]);
            for (int j = 0; j < jn; ++j) {
                if (s[j] == -INFINITY) continue;
                float pj = exp2f(s[j] * c - m_scaled);
                rowsum += pj;
                float pb = rbf(pj); /* copy(att_block_mma, att_block): P -> bf16 before P.V */
                int p = j0 + j;
                int key = idx ? idx[p] : p;
                const uint16_t *vrow = vb + (int64_t)key * in->vs[2];
                for (int d = 0; d < HEAD_DIM; ++d) orow[d] += pb * bf2f(vrow[d]);
                if (cs) {
                    /* mul_row in bf16, then the 16-row per-warp col_sum (dense_colsum_attn.cu:272-274) */
                    float prod = rbf(pb * rowfac);
                    colacc[(r / 16) * kv_tile + j] += prod;
                }
            }
            l[r] = l[r] * alpha + rowsum;
            m[r] = mx;
        }
        if (cs) {
            /* cross-warp reduction by bf16 shared-memory atomics (store_add, dense_colsum_attn.cu:275-277):
             * restated as warp-ordered bf16 adds of bf16-rounded per-warp partials. */
            for (int j = 0; j < jn; ++j) {
                int p = j0 + j;
                if (p >= count) break;
                float acc = 0.f;
                for (int w = 0; w < 12; ++w) acc = rbf(acc + rbf(colacc[w * kv_tile + j]));
                cs[p] = f2bf(acc);
            }
        }
    }
    for (int r = 0; r < rows; ++r) {
        float *orow = out_o + (size_t)r * HEAD_DIM;
        for (int d = 0; d < HEAD_DIM; ++d) orow[d] = orow[d] / l[r];
        if (out_l) out_l[r] = 1.0f / (exp2f(m[r] * c) * l[r]);
    }
    free(m);
    free(l);
    free(s);
    if (colacc) free(colacc);
}

/*
 * csp_attn (in place, o += o_scale * result; csrc/attn/csp_attn.cu:294-300,315-423; kv tile 112) and
 * csp_128_attn (out of place; csrc/attn/csp_128_attn.cu:355-461; kv tile 128).
 * in_place != 0: o_new = bf16(o_old + bf16(o_scale * result))   (bf16 store then bf16 reduce-add, :294-300)
 * Groups: G = ceil(Nq/192); indices [B,H,G,idx_stride] int32; counts [B,H,G].
 * Deviation, documented in DESIGN.md: a group with count == 0 contributes zeros (the reference divides 0/0).
 */
void oracle_csp_attn(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *o, const int64_t *qs,
                     const int64_t *ks, const int64_t *vs, const int64_t *os, const int32_t *indices,
                     const int32_t *counts, int B, int H, int Nq, int Nk, int idx_stride, int kv_tile, int in_place,
                     int o_scale) {
    attn_in_t in = {q, k, v, {qs[0], qs[1], qs[2]}, {ks[0], ks[1], ks[2]}, {vs[0], vs[1], vs[2]}, B, H, Nq, Nk};
    int G = (Nq + QGROUP - 1) / QGROUP;
#pragma omp parallel for collapse(3) schedule(dynamic)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int g = 0; g < G; ++g) {
                int row0 = g * QGROUP;
                int rows = Nq - row0 < QGROUP ? Nq - row0 : QGROUP;
                float *buf = (float *)malloc(sizeof(float) * QGROUP * HEAD_DIM);
                int64_t gi = ((int64_t)b * H + h) * G + g;
                int count = counts[gi];
                if (count > 0)
                    attn_group(&in, b, h, row0, rows, indices + gi * idx_stride, count, kv_tile, buf, NULL, NULL, NULL);
                else
                    memset(buf, 0, sizeof(float) * QGROUP * HEAD_DIM);
                for (int r = 0; r < rows; ++r) {
                    uint16_t *orow = o + (int64_t)b * os[0] + (int64_t)h * os[1] + (int64_t)(row0 + r) * os[2];
                    for (int d = 0; d < HEAD_DIM; ++d) {
                        float res = buf[(size_t)r * HEAD_DIM + d];
                        if (in_place) {
                            float add = rbf(res * (float)o_scale);
                            orow[d] = f2bf(bf2f(orow[d]) + add);
                        } else {
                            orow[d] = f2bf(res);
                        }
                    }
                }
                free(buf);
            }
}

/*
 * dense_attn (csrc/attn/dense_attn.cu:246-372): o bf16 [B,H,Nq,128], l fp32 [B,H,Nq] = 1/sum_j exp(s_ij/sqrt(D)).
 * dense_colsum_attn (csrc/attn/dense_colsum_attn.cu:521-668): also cs bf16 [B,H,ceil(Nq/192),cs_stride]
 * (cs_stride = Nq in the reference, :580-583); only columns < Nk of each row are written.
 * prev_l: fp32 [B,H,Nq] (last step's l); NULL for plain dense_attn.
 */
void oracle_dense_attn(const uint16_t *q, const uint16_t *k, const uint16_t *v, uint16_t *o, float *l_out,
                       const int64_t *qs, const int64_t *ks, const int64_t *vs, int B, int H, int Nq, int Nk,
                       const float *prev_l, uint16_t *cs, int cs_stride) {
    attn_in_t in = {q, k, v, {qs[0], qs[1], qs[2]}, {ks[0], ks[1], ks[2]}, {vs[0], vs[1], vs[2]}, B, H, Nq, Nk};
    int G = (Nq + QGROUP - 1) / QGROUP;
#pragma omp parallel for collapse(3) schedule(dynamic)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int g = 0; g < G; ++g) {
                int row0 = g * QGROUP;
                int rows = Nq - row0 < QGROUP ? Nq - row0 : QGROUP;
                float *buf = (float *)malloc(sizeof(float) * QGROUP * HEAD_DIM);
                float lbuf[QGROUP];
                float pbuf[QGROUP];
                int64_t bh = (int64_t)b * H + h;
                if (prev_l)
                    for (int r = 0; r < QGROUP; ++r) pbuf[r] = r < rows ? prev_l[bh * Nq + row0 + r] : 0.f;
                attn_group(&in, b, h, row0, rows, NULL, Nk, 128, buf, lbuf, prev_l ? pbuf : NULL,
                           cs ? cs + (bh * G + g) * (int64_t)cs_stride : NULL);
                for (int r = 0; r < rows; ++r) {
                    uint16_t *orow = o + (bh * Nq + row0 + r) * (int64_t)HEAD_DIM;
                    for (int d = 0; d < HEAD_DIM; ++d) orow[d] = f2bf(buf[(size_t)r * HEAD_DIM + d]);
                    l_out[bh * Nq + row0 + r] = lbuf[r];
                }
                free(buf);
            }
}

/* ------------------------------------------------------------------ MLP */
/* tanh-GeLU: x*0.5*(1+tanh(0.79788456*x*(1+0.044715x^2)))  (csrc/common/elementwise/gelu.cuh:26-30,
 * csrc/mlp/csp_mlp_mm1.cu:401-409 -- the in-tree CPU reference uses tanhf) */
static inline float gelu_tanh(float x) {
    const float sqrt_2_over_pi = 0.7978845608028654f, coef = 0.044715f;
    return x * (0.5f * (1.0f + tanhf(sqrt_2_over_pi * (x + coef * x * x * x))));
}

/*
 * csp_mlp_mm1 (csrc/mlp/csp_mlp_mm1.cu:345-390,411-424,625-702):
 *   for 128-row group g and packed column j < counts[g]:
 *     C[m, j] = bf16( gelu(A[m,:].B[idx[g,j],:] + bias[idx[g,j]]) - pa_cache[idx[g,j], m] )
 * a [M,K], b [F,K] (= fc1.weight), c [M,F] packed, bias [F], pa_cache [F,M] (col-major activations),
 * indices [M/128, F], counts [M/128].  Columns j >= counts[g] of C are left untouched.
 */
void oracle_csp_mlp_mm1(const uint16_t *a, const uint16_t *bw, uint16_t *c, const uint16_t *bias,
                        const uint16_t *pa_cache, const int32_t *indices, const int32_t *counts, int M, int K, int F) {
    int G = M / 128;
#pragma omp parallel for schedule(dynamic)
    for (int gm = 0; gm < G * 128; ++gm) {
        int g = gm / 128, m = gm;
        int cnt = counts[g];
        float *af = (float *)malloc(sizeof(float) * K);
        for (int kk = 0; kk < K; ++kk) af[kk] = bf2f(a[(size_t)m * K + kk]);
        for (int j = 0; j < cnt; ++j) {
            int col = indices[(size_t)g * F + j];
            const uint16_t *brow = bw + (size_t)col * K;
            float acc = bf2f(bias[col]); /* bias seeds the fp32 accumulator (:347-350) */
            for (int kk = 0; kk < K; ++kk) acc += af[kk] * bf2f(brow[kk]);
            acc = gelu_tanh(acc);
            acc -= bf2f(pa_cache[(size_t)col * M + m]);
            c[(size_t)m * F + j] = f2bf(acc);
        }
        free(af);
    }
}

/*
 * csp_scatter_add (csrc/indexed_io/scatter_add.cu:43-98):
 *   unpacked_colmajor[idx[g,c], g*128 + r] += packed[g*128 + r, c]   for c < counts[g], r < 128 (bf16 add)
 */
void oracle_csp_scatter_add(const uint16_t *packed, uint16_t *unpacked_colmajor, const int32_t *indices,
                            const int32_t *counts, int M, int F) {
    int G = M / 128;
#pragma omp parallel for schedule(dynamic)
    for (int g = 0; g < G; ++g) {
        int cnt = counts[g];
        for (int cc = 0; cc < cnt; ++cc) {
            int col = indices[(size_t)g * F + cc];
            for (int r = 0; r < 128; ++r) {
                size_t off = (size_t)col * M + (size_t)g * 128 + r;
                unpacked_colmajor[off] = f2bf(bf2f(unpacked_colmajor[off]) + bf2f(packed[((size_t)g * 128 + r) * F + cc]));
            }
        }
    }
}

/*
 * GEMM2 of csp_mlp_mm2_and_scatter_add (src/chipmunk/triton/csp_mlp_mm2.py:68-110):
 *   out[m,:] = bf16(sum_{c<count_g} packed[m,c] * w2t[idx[g,c],:]) + out[m,:]    (bf16 add, :100-101)
 * packed [M,F], w2t [F,N2] (= fc2.weight.T contiguous), out [M,N2].
 */
void oracle_csp_mlp_mm2(const uint16_t *packed, const uint16_t *w2t, uint16_t *out, const int32_t *indices,
                        const int32_t *counts, int M, int F, int N2) {
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < M; ++m) {
        int g = m / 128;
        int cnt = counts[g];
        float *acc = (float *)calloc(N2, sizeof(float));
        for (int cc = 0; cc < cnt; ++cc) {
            int col = indices[(size_t)g * F + cc];
            float pa = bf2f(packed[(size_t)m * F + cc]);
            const uint16_t *wrow = w2t + (size_t)col * N2;
            for (int n = 0; n < N2; ++n) acc[n] += pa * bf2f(wrow[n]);
        }
        for (int n = 0; n < N2; ++n) out[(size_t)m * N2 + n] = f2bf(rbf(acc[n]) + bf2f(out[(size_t)m * N2 + n]));
        free(acc);
    }
}

/* OCP fp8 e4m3fn -> float (1-4-3, bias 7, no infinities, S.1111.111 = NaN, max 448) */
static inline float fp8_e4m3fn_to_f(uint8_t v) {
    int sign = v >> 7, e = (v >> 3) & 0xf, m = v & 7;
    float f;
    if (e == 0xf && m == 7) return NAN;
    if (e == 0) f = ldexpf((float)m, -9);            /* subnormal: m/8 * 2^-6 */
    else f = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return sign ? -f : f;
}

/*
 * fp8 GEMM1 (src/chipmunk/triton/csp_mlp_mm1.py:37-164):
 *   acc = (a_fp8 . b_fp8[idx]) * scale_a * scale_b (:121-122);  x = bf16(gelu(acc + bias[idx])) (:124-130);
 *   c[m,j] = bf16(x - pa_cache[idx, m]) (:133, bf16 subtract);  if update_cache: pa_cache[idx, m] = x (:140).
 */
void oracle_csp_mlp_mm1_fp8(const uint8_t *a, const uint8_t *bw, uint16_t *c, const uint16_t *bias, uint16_t *pa_cache,
                            const int32_t *indices, const int32_t *counts, float scale_a, float scale_b, int M, int K,
                            int F, int update_cache) {
    int G = M / 128;
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < G * 128; ++m) {
        int g = m / 128;
        int cnt = counts[g];
        float *af = (float *)malloc(sizeof(float) * K);
        for (int kk = 0; kk < K; ++kk) af[kk] = fp8_e4m3fn_to_f(a[(size_t)m * K + kk]);
        for (int j = 0; j < cnt; ++j) {
            int col = indices[(size_t)g * F + j];
            const uint8_t *brow = bw + (size_t)col * K;
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) acc += af[kk] * fp8_e4m3fn_to_f(brow[kk]);
            acc = acc * scale_a;
            acc = acc * scale_b;
            float x = rbf(gelu_tanh(acc + bf2f(bias[col])));
            size_t coff = (size_t)col * M + m;
            c[(size_t)m * F + j] = f2bf(x - bf2f(pa_cache[coff]));
            if (update_cache) pa_cache[coff] = f2bf(x);
        }
        free(af);
    }
}

/* dense eager MLP of the reference's CPU path: fc2(gelu_tanh(fc1(x)))  (src/chipmunk/modules/mlp.py:33-34,51-53).
 * x [M,K], w1 [F,K], b1 [F], w2 [N2,F], b2 [N2]; mid/pa/out rounded to bf16 like nn.Linear in bf16. */
void oracle_dense_mlp(const uint16_t *x, const uint16_t *w1, const uint16_t *b1, const uint16_t *w2,
                      const uint16_t *b2, uint16_t *out, int M, int K, int F, int N2) {
#pragma omp parallel for schedule(dynamic)
    for (int m = 0; m < M; ++m) {
        float *pa = (float *)malloc(sizeof(float) * F);
        for (int f = 0; f < F; ++f) {
            float acc = 0.f;
            for (int kk = 0; kk < K; ++kk) acc += bf2f(x[(size_t)m * K + kk]) * bf2f(w1[(size_t)f * K + kk]);
            float mid = rbf(acc + bf2f(b1[f]));
            pa[f] = rbf(gelu_tanh(mid));
        }
        for (int n = 0; n < N2; ++n) {
            float acc = 0.f;
            for (int f = 0; f < F; ++f) acc += pa[f] * bf2f(w2[(size_t)n * F + f]);
            out[(size_t)m * N2 + n] = f2bf(acc + bf2f(b2[n]));
        }
        free(pa);
    }
}

/* ------------------------------------------------------------------ indexed IO */
static int cmp_float(const void *x, const void *y) {
    float a = *(const float *)x, b = *(const float *)y;
    return (a > b) - (a < b);
}

/*
 * topk_indices (csrc/indexed_io/topk_indices.cu:26-141), random_amount == 0 only.
 *   threshold = ascending-sorted first 1024 values of the row, element int(1024*sparsity) (:9-12,94-101);
 *   keep x >= threshold (:107-111); sparsity 0 -> all columns, 1 -> none (:51-69);
 *   count padded up to multiple_of with rejected columns (:126-140).
 * The reference's order inside the kept set and the choice of padding columns depend on atomics.  The
 * canonical order restated here is one the reference can produce: kept columns ascending, then the
 * padding candidates "last rejected column of thread t" (t = col % 1024, :118-120,135-139) for ascending t.
 * act is fp32 (the caller converts bf16/fp16 exactly).  indices [rows, cols] (entries beyond the count are
 * left untouched except the sparsity==1 case which fills -1 like the reference), counts [rows].
 */
void oracle_topk_indices(const float *act, int32_t *indices, int32_t *counts, int rows, int cols, double sparsity,
                         int multiple_of) {
    float quantile = (float)sparsity;
#pragma omp parallel for schedule(dynamic)
    for (int r = 0; r < rows; ++r) {
        const float *x = act + (size_t)r * cols;
        int32_t *out = indices + (size_t)r * cols;
        if (quantile == 0.f) {
            for (int ccol = 0; ccol < cols; ++ccol) out[ccol] = ccol;
            counts[r] = cols;
            continue;
        }
        if (quantile == 1.f) {
            for (int ccol = 0; ccol < cols; ++ccol) out[ccol] = -1;
            counts[r] = 0;
            continue;
        }
        float sample[1024];
        for (int i = 0; i < 1024; ++i) sample[i] = x[i];
        qsort(sample, 1024, sizeof(float), cmp_float);
        float thr = sample[(int)(1024 * quantile)];
        int cnt = 0;
        int last_invalid[1024];
        for (int t = 0; t < 1024; ++t) last_invalid[t] = -1;
        for (int ccol = 0; ccol < cols; ++ccol) {
            if (x[ccol] >= thr)
                out[cnt++] = ccol;
            else
                last_invalid[ccol % 1024] = ccol;
        }
        int mod = cnt % multiple_of;
        int pad = mod == 0 ? 0 : multiple_of - mod;
        counts[r] = cnt + pad;
        for (int t = 0; t < 1024 && pad > 0; ++t)
            if (last_invalid[t] != -1) {
                out[cnt++] = last_invalid[t];
                --pad;
            }
    }
}

/*
 * mask_to_indices (csrc/indexed_io/mask_to_indices.cu:21-88,92-143).  Integer-only; order is exact:
 * one 32-lane warp per row, lane t emits its True columns t, t+32, ... ascending, lanes concatenated (:49-68);
 * count rounded up to multiple_of, padded by lane 0 with the first False columns ascending (:71-86).
 * mask [rows, n] bytes, indices [rows, pad_n] (pad_n = n rounded up to pad_to_multiple_of, :107), counts [rows].
 */
void oracle_mask_to_indices(const uint8_t *mask, int32_t *indices, int32_t *counts, int64_t rows, int n, int pad_n,
                            int multiple_of) {
#pragma omp parallel for schedule(dynamic)
    for (int64_t r = 0; r < rows; ++r) {
        const uint8_t *mrow = mask + r * n;
        int32_t *out = indices + r * pad_n;
        int off = 0;
        for (int t = 0; t < 32; ++t)
            for (int ccol = t; ccol < n; ccol += 32)
                if (mrow[ccol]) out[off++] = ccol;
        int total = off;
        int padded = ((total + multiple_of - 1) / multiple_of) * multiple_of;
        if (padded > total)
            for (int ccol = 0; ccol < n; ++ccol)
                if (!mrow[ccol]) {
                    out[off++] = ccol;
                    if (off == padded) break;
                }
        counts[r] = padded;
    }
}

/*
 * copy_indices (csrc/indexed_io/copy_indices.cu:35-78): cache[b, row, idx] = src[b, row, idx] for the first
 * counts[b, row / R] indices of sp_inds[b, row / R, :].  elem_size in bytes (2 or 4).
 */
void oracle_copy_indices(const uint8_t *src, uint8_t *dst, const int32_t *inds, const int32_t *counts, int B, int M,
                         int R, int F, int elem_size) {
    for (int b = 0; b < B; ++b)
        for (int row = 0; row < M * R; ++row) {
            int base_m = row / R;
            int cnt = counts[b * M + base_m];
            for (int cc = 0; cc < cnt; ++cc) {
                int col = inds[((size_t)b * M + base_m) * F + cc];
                size_t off = (((size_t)b * M * R + row) * F + col) * elem_size;
                memcpy(dst + off, src + off, elem_size);
            }
        }
}

/* bitpack / bitunpack (src/chipmunk/ops/bitpack.py:4-69): little-endian, 8 bools -> 1 byte, flat. */
void oracle_bitpack(const uint8_t *mask, uint8_t *packed, int64_t n) {
    int64_t nb = (n + 7) / 8;
    for (int64_t i = 0; i < nb; ++i) {
        uint8_t byte = 0;
        for (int j = 0; j < 8; ++j)
            if (i * 8 + j < n && mask[i * 8 + j]) byte |= (uint8_t)(1u << j);
        packed[i] = byte;
    }
}
void oracle_bitunpack(const uint8_t *packed, uint8_t *mask, int64_t n) {
    for (int64_t i = 0; i < n; ++i) mask[i] = (packed[i / 8] >> (i % 8)) & 1u;
}
