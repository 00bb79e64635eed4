// Row-wise passes of the transformer block around the attention / MLP operators (model code in the reference, torch ops there):
// gated residual + LayerNorm + modulate in one pass over the hidden state.
//
// The blocks of the DiT models the reference patches do, between any two GEMMs,
//     x  = x + gate * y                      (torch.addcmul; examples/hunyuan/hyvideo/modules/models.py:262-275, 431)
//     xm = LayerNorm(x) * (1 + scale) + shift   (modulate(norm(x)), models.py:184-186, 265-268, 371-372; no affine, eps 1e-6)
// as three elementwise / normalisation kernels: 8 passes over the [rows, C] bf16 hidden state (read x, y, write x; read x, write
// xn; read xn, write xm) where 4 suffice (read x, y; write x, xm).  HBM-bound: 4 * rows * C * 2 bytes per call.
//
// One wave per row (C = 3 072: 6 x 16 bytes per lane), the row in registers between the two reductions (mean, then the centred
// sum of squares: the two-pass form, no E[x^2] - E[x]^2 cancellation), 64-lane sums by DPP + lane swaps, no LDS.  A wave keeps
// its gate / shift / scale vectors in registers and walks rows with the grid's stride.  Rounding points are torch's: bf16 after
// the residual, bf16 after the normalisation, bf16(1 + scale), bf16 after the modulation.
#include "common.h"

namespace {

template <int NV, bool RESIDUAL>   // NV 16-byte vectors per lane: C <= 512 * NV
__global__ __launch_bounds__(256) void residual_ln_modulate_kernel(const uint16_t *x, const uint16_t *y, const uint16_t *gate,
                                                                   const uint16_t *shift, const uint16_t *scale, uint16_t *x_out,
                                                                   uint16_t *xm, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const float inv_c = 1.0f / (float)C;
    u32x4 g[NV], sh[NV], sc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (lane + 64 * j) * 8;
        const u32x4 z = {0u, 0u, 0u, 0u};
        g[j] = (RESIDUAL && c < C) ? *(const u32x4 *)(gate + c) : z;
        sh[j] = c < C ? *(const u32x4 *)(shift + c) : z;
        sc[j] = c < C ? *(const u32x4 *)(scale + c) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // bf16(1 + scale), once per wave
            float a = 1.0f + __uint_as_float(sc[j][e] << 16), b = 1.0f + __uint_as_float(sc[j][e] & 0xffff0000u);
            sc[j][e] = pack_bf16x2(a, b);
        }
    }
    for (int64_t r = wave; r < rows; r += nwaves) {
        float f[NV][8];
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (lane + 64 * j) * 8;
            u32x4 xv = {0u, 0u, 0u, 0u}, yv = xv;
            if (c < C) {
                xv = *(const u32x4 *)(x + r * C + c);
                if constexpr (RESIDUAL) yv = *(const u32x4 *)(y + r * C + c);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = __uint_as_float(xv[e] << 16), b = __uint_as_float(xv[e] & 0xffff0000u);
                if constexpr (RESIDUAL) {
                    a += __uint_as_float(g[j][e] << 16) * __uint_as_float(yv[e] << 16);
                    b += __uint_as_float(g[j][e] & 0xffff0000u) * __uint_as_float(yv[e] & 0xffff0000u);
                    round_bf16_pair(a, b);
                    xv[e] = pack_bf16x2(a, b);
                }
                f[j][2 * e] = a, f[j][2 * e + 1] = b;
                sum += a + b;
            }
            if constexpr (RESIDUAL)
                if (c < C) *(u32x4 *)(x_out + r * C + c) = xv;
        }
        const float mean = sum_across_rows(row16_sum(sum)) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const bool in = (lane + 64 * j) * 8 < C;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                f[j][e] -= mean;
                ss = in ? __builtin_fmaf(f[j][e], f[j][e], ss) : ss;
            }
        }
        const float var = sum_across_rows(row16_sum(ss)) * inv_c;
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int c = (lane + 64 * j) * 8;
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = f[j][2 * e] * rstd, b = f[j][2 * e + 1] * rstd;
                round_bf16_pair(a, b);
                a = __builtin_fmaf(a, __uint_as_float(sc[j][e] << 16), __uint_as_float(sh[j][e] << 16));
                b = __builtin_fmaf(b, __uint_as_float(sc[j][e] & 0xffff0000u), __uint_as_float(sh[j][e] & 0xffff0000u));
                o[e] = pack_bf16x2(a, b);
            }
            if (c < C) *(u32x4 *)(xm + r * C + c) = o;
        }
    }
}

template <int NV>
void launch_rlm(const void *x, const void *y, const void *gate, const void *shift, const void *scale, void *x_out, void *xm, int64_t rows,
                int C, float eps, hipStream_t s) {
    // 4 rows per workgroup at a time; enough workgroups for 8 per CU, the rest of the rows by stride
    const int64_t want = (rows + 3) / 4;
    const unsigned grid = (unsigned)(want < 256 * 8 ? want : 256 * 8);
    if (y)
        hipLaunchKernelGGL((residual_ln_modulate_kernel<NV, true>), dim3(grid), dim3(256), 0, s, (const uint16_t *)x, (const uint16_t *)y,
                           (const uint16_t *)gate, (const uint16_t *)shift, (const uint16_t *)scale, (uint16_t *)x_out, (uint16_t *)xm, rows,
                           C, eps);
    else
        hipLaunchKernelGGL((residual_ln_modulate_kernel<NV, false>), dim3(grid), dim3(256), 0, s, (const uint16_t *)x, nullptr, nullptr,
                           (const uint16_t *)shift, (const uint16_t *)scale, nullptr, (uint16_t *)xm, rows, C, eps);
}


// Mean over consecutive blocks of `mbm` rows: [R, C] -> [R / mbm, C] bf16 (reference modules/mlp.py:11-16, `block_mean`, the first
// operation of every sparse MLP step; torch's reshape + mean kernel reads the 27 MB of a FLUX layer's input in 17 us).  One workgroup
// = (block, 512 columns): the four waves take a quarter of the block's rows each, a lane 8 columns (16 bytes) of every row; fp32
// sums in row order, the four partial sums added in wave order, one rounding to bf16.  HBM-bound: R * C * 2 bytes.
template <int NW, int VEC>   // NW waves per workgroup, each takes mbm / NW rows of the block (all requested before the first is added);
__global__ __launch_bounds__(NW * 64) void block_mean_kernel(const uint16_t *x, uint16_t *out, int C, int mbm) {   // VEC = columns per lane (8 or 4)
    __shared__ float part[NW - 1][64][VEC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = (blockIdx.x * 64 + lane) * VEC;
    const int64_t blk = blockIdx.y;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (c < C) {
        const int per = mbm / NW;
        const uint16_t *src = x + (blk * mbm + (int64_t)w * per) * C + c;
#pragma unroll 16
        for (int r = 0; r < per; ++r) {
            uint32_t v[VEC / 2];
            if constexpr (VEC == 8) {
                const u32x4 t = *(const u32x4 *)(src + (int64_t)r * C);
                v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
            } else {
                const u32x2 t = *(const u32x2 *)(src + (int64_t)r * C);
                v[0] = t[0], v[1] = t[1];
            }
#pragma unroll
            for (int e = 0; e < VEC / 2; ++e) acc[2 * e] += __uint_as_float(v[e] << 16), acc[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
        }
    }
    if (w > 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) part[w - 1][lane][e] = acc[e];
    }
    __syncthreads();
    if (w == 0 && c < C) {
        const float inv = 1.0f / (float)mbm;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int ww = 0; ww < NW - 1; ++ww) acc[e] += part[ww][lane][e];      // fixed order: wave 0's rows, then waves 1 .. NW-1
        uint32_t o[VEC / 2];
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) o[e] = pack_bf16x2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
        if constexpr (VEC == 8) *(u32x4 *)(out + blk * C + c) = (u32x4){o[0], o[1], o[2], o[3]};
        else *(u32x2 *)(out + blk * C + c) = (u32x2){o[0], o[1]};
    }
}
}  // namespace

extern "C" int chipmunk_residual_ln_modulate(const void *x, const void *y, const void *gate, const void *shift, const void *scale,
                                             void *x_out, void *xm, int64_t rows, int cols, double eps, void *stream) {
    CM_CHECK(x && shift && scale && xm, "residual_ln_modulate: null pointer");
    CM_CHECK((y == nullptr) == (gate == nullptr) && (y == nullptr) == (x_out == nullptr),
             "residual_ln_modulate: y, gate and x_out come together (all set: residual form; all null: LayerNorm + modulate only)");
    CM_CHECK(rows >= 0 && cols > 0 && cols % 8 == 0 && cols <= 512 * 16, "residual_ln_modulate: cols must be a multiple of 8, at most 8192 (got %d)",
             cols);
    CM_CHECK(((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gate | (uintptr_t)shift | (uintptr_t)scale | (uintptr_t)x_out | (uintptr_t)xm)) & 15) == 0,
             "residual_ln_modulate: pointers must be 16-byte aligned");
    if (rows == 0) return CHIPMUNK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nv = (cols + 511) / 512;
    const float e = (float)eps;
    if (nv <= 2) launch_rlm<2>(x, y, gate, shift, scale, x_out, xm, rows, cols, e, s);
    else if (nv <= 3) launch_rlm<3>(x, y, gate, shift, scale, x_out, xm, rows, cols, e, s);
    else if (nv <= 6) launch_rlm<6>(x, y, gate, shift, scale, x_out, xm, rows, cols, e, s);
    else if (nv <= 10) launch_rlm<10>(x, y, gate, shift, scale, x_out, xm, rows, cols, e, s);
    else launch_rlm<16>(x, y, gate, shift, scale, x_out, xm, rows, cols, e, s);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_block_mean(const void *x, void *out, int64_t rows, int C, int mbm, void *stream) {
    CM_CHECK(x && out, "block_mean: null tensor pointer");
    CM_CHECK(mbm > 0 && mbm % 4 == 0 && rows > 0 && rows % mbm == 0, "block_mean: rows (%lld) must be a positive multiple of mbm (%d), mbm a multiple of 4",
             (long long)rows, mbm);
    CM_CHECK(C > 0 && C % 8 == 0 && rows / mbm < 65536, "block_mean: C must be a positive multiple of 8 and rows / mbm < 65536");
    // 8 waves x mbm / 8 rows x FOUR columns per lane when the block divides (FLUX / Wan: 128 rows): 408 workgroups instead of 204 and 16 loads in
    // flight per lane -- 27 MB of cold rows in 6.5 us (4.2 TB/s); 4 waves x 8 columns measured 17.3 us, 8 x 8: 10.9, 16 x 4: 7.3, 8 x 2: 6.5
    if (mbm % 8 == 0)
        hipLaunchKernelGGL((block_mean_kernel<8, 4>), dim3((C + 255) / 256, (unsigned)(rows / mbm)), dim3(512), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (uint16_t *)out, C, mbm);
    else
        hipLaunchKernelGGL((block_mean_kernel<4, 8>), dim3((C + 511) / 512, (unsigned)(rows / mbm)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t *)x, (uint16_t *)out, C, mbm);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

// ---------------------------------------------------------------------------------------------- fp8 input quantisation
// F8Linear.quantize_input (reference src/chipmunk/modules/mlp_fp8.py / flux fp8 linear: `(x * scale).clamp(-max, max).to(float8_e4m3fn)`)
// as ONE pass: torch runs it as three elementwise kernels (140 us at Wan2.1's 32 768 x 1536 rows, next to a 350 us GEMM1).  Same arithmetic,
// bit for bit: the scale is rounded to bf16 and the product formed in fp32 and rounded to bf16 (torch's bf16 tensor x 0-dim fp32 tensor), clamped in bf16 (NaN stays NaN), and
// converted to OCP e4m3 with round-to-nearest-even (v_cvt_pk_fp8_f32; the clamp keeps every value in range).
namespace {
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const uint16_t *x, const float *scale, uint8_t *out, int64_t n8, float maxv) {
    // torch multiplies a bf16 tensor by a 0-dim fp32 tensor in the tensor's dtype: the scale is rounded to bf16 first
    const float sc = bf16_bits_to_f32(f32_to_bf16_bits(scale[0]));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const u32x4 v = *(const u32x4 *)(x + i * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = __uint_as_float(v[e] << 16) * sc;
            f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u) * sc;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float r = bf16_bits_to_f32(f32_to_bf16_bits(f[e]));      // the bf16 product torch materialises
            r = r != r ? r : fminf(fmaxf(r, -maxv), maxv);           // clamp; NaN propagates as in torch
            f[e] = r;
        }
        u32x2 o;
        int w0 = 0, w1 = 0;
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
        w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
        w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
        o[0] = (uint32_t)w0, o[1] = (uint32_t)w1;
#pragma unroll
        for (int e = 0; e < 8; ++e)              // NaN: torch's cast gives 0x7f with the input's sign
            if (f[e] != f[e]) {
                const uint32_t byte = 0x7fu | ((__float_as_uint(f[e]) >> 24) & 0x80u);
                o[e >> 2] = (o[e >> 2] & ~(0xffu << ((e & 3) * 8))) | (byte << ((e & 3) * 8));
            }
        *(u32x2 *)(out + i * 8) = o;
    }
}
}  // namespace

extern "C" int chipmunk_quantize_fp8(const void *x, const float *scale, void *out, int64_t n, float max_value, void *stream) {
    CM_CHECK(x && scale && out, "quantize_fp8: null pointer");
    CM_CHECK(n >= 0 && n % 8 == 0, "quantize_fp8: the element count must be a multiple of 8 (got %lld)", (long long)n);
    CM_CHECK((((uintptr_t)x & 15) | ((uintptr_t)out & 7)) == 0, "quantize_fp8: x must be 16-byte and out 8-byte aligned");
    if (n == 0) return CHIPMUNK_OK;
    const int64_t n8 = n / 8;
    const int64_t blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(quantize_fp8_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)x, scale, (uint8_t *)out, n8, max_value);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}
