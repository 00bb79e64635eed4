// Column-sparse ("delta") MLP GEMMs for gfx950.
// Replaces reference csrc/mlp/csp_mlp_mm1.cu, csrc/mlp/csp_mlp_mm2_and_scatter_add.cu (+ the Triton GEMM
// src/chipmunk/triton/csp_mlp_mm2.py) and csrc/indexed_io/scatter_add.cu.
//
// Both GEMMs: workgroup tile 128 rows (one sparsity group, reference bm = 128) x 256 packed columns, K step 64,
// 4 waves (2x2), each wave a 64x128 accumulator of v_mfma_f32_32x32x16_bf16 tiles (128 fp32 VGPRs/lane), one wave
// per SIMD, one workgroup per CU, 2-deep LDS ring filled by LDS-DMA (global_load_lds_dwordx4).  The gather is the
// per-lane source address of the DMA: fc1 rows (mm1) are 2*K contiguous bytes, fc2^T rows (mm2) are 2*N2 contiguous
// bytes, so every gathered piece is a full 128-byte line.  XOR swizzles are applied on the source chunk index so the
// lane-linear LDS image is conflict-free for ds_read_b128 (k-contiguous operands) and ds_read_b64_tr_b16 (fc2^T,
// which is n-contiguous in memory and must be fed k-contiguous to the MFMA).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 256, BK = 64;
constexpr int A_TILE = BM * BK * 2;   // 16 KiB
constexpr int B1_TILE = BN * BK * 2;  // 32 KiB (mm1: 256 gathered rows x 64 k)
constexpr int B2_TILE = BK * BN * 2;  // 32 KiB (mm2: 64 gathered rows x 256 n)
constexpr int MLP_LDS = 2 * (A_TILE + B1_TILE);

__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// tanh-GeLU (reference csrc/common/elementwise/gelu.cuh:26-30): x*0.5*(1+tanh(u)) == x*(1 - 1/(1+exp(2u)))
__device__ __forceinline__ float gelu_tanh(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float e = __builtin_amdgcn_exp2f(u * (2.0f * 1.44269504089f));
    return x - x * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------------------------------------ mm1
struct Mm1Params {
    const uint16_t *a, *b, *bias, *cache;
    uint16_t *c;
    const int32_t *indices, *counts;
    int M, K, F, NT;
};

// k-contiguous [rows][64] bf16 tile image, row stride 128 B, 16-byte chunk c of row r stored at chunk c ^ ((r>>1)&7)
__device__ __forceinline__ bf16x8 read_kfrag(const unsigned char *tile, int row, int kk, int lane) {
    const int c = kk * 2 + (lane >> 5);
    return *(const bf16x8 *)(tile + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
}

__global__ __launch_bounds__(256, 1) void mm1_kernel(const Mm1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Al = smem;               // [2][A_TILE]
    unsigned char *Bl = smem + 2 * A_TILE;  // [2][B1_TILE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int g = wid / p.NT, nt = wid - g * p.NT;
    const int cnt = p.counts[g];
    const int n0 = nt * BN;
    if (n0 >= cnt) return;  // tiles past counts[g] are skipped (csp_mlp_mm1.cu:233-243)
    const int32_t *idxg = p.indices + (int64_t)g * p.F;

    // per-lane DMA sources: lane -> (row = inst*8 + lane/8, stored chunk = lane%8), source chunk = stored ^ swizzle(row)
    int aoff[4], boff[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (lane >> 3);
        aoff[i] = (g * BM + row) * p.K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (w * 8 + i) * 8 + (lane >> 3);
        const int j = n0 + row;
        const int key = idxg[j < cnt ? j : n0];  // rows past the count re-read a live row and are never stored
        boff[i] = key * p.K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    auto issue = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.a + aoff[i] + kb * BK, Al + buf * A_TILE + (w * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) glds16(p.b + boff[i] + kb * BK, Bl + buf * B1_TILE + (w * 8 + i) * 1024);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][n4][r] = 0.f;

    const int nkb = p.K / BK;
    issue(0, 0);
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < nkb) issue(kb + 1, buf ^ 1);
        const unsigned char *At = Al + buf * A_TILE;
        const unsigned char *Bt = Bl + buf * B1_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 af[2], bfr[4];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) af[mt] = read_kfrag(At, wm * 64 + mt * 32 + (lane & 31), kk, lane);
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) bfr[n4] = read_kfrag(Bt, wn * 128 + n4 * 32 + (lane & 31), kk, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int n4 = 0; n4 < 4; ++n4) acc[mt][n4] = mfma32(af[mt], bfr[n4], acc[mt][n4]);
        }
    }

    // ---- epilogue: lane owns packed column j = lane&31 of each 32x32 tile and rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    //      C[m,j] = bf16(gelu(acc + bias[idx]) - cache[idx, m])     (csp_mlp_mm1.cu:354-390)
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4) {
        const int j = n0 + wn * 128 + n4 * 32 + (lane & 31);
        const bool live = j < cnt;
        const int col = live ? idxg[j] : 0;
        const float bia = bf16_bits_to_f32(p.bias[col]);
        const uint16_t *crow = p.cache + (int64_t)col * p.M;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int m = g * BM + wm * 64 + mt * 32 + q4 * 8 + (lane >> 5) * 4;
                const u32x2 cv = *(const u32x2 *)(crow + m);
                const float c0 = __uint_as_float(cv[0] << 16), c1 = __uint_as_float(cv[0] & 0xffff0000u);
                const float c2 = __uint_as_float(cv[1] << 16), c3 = __uint_as_float(cv[1] & 0xffff0000u);
                const float x0 = gelu_tanh(acc[mt][n4][q4 * 4 + 0] + bia) - c0;
                const float x1 = gelu_tanh(acc[mt][n4][q4 * 4 + 1] + bia) - c1;
                const float x2 = gelu_tanh(acc[mt][n4][q4 * 4 + 2] + bia) - c2;
                const float x3 = gelu_tanh(acc[mt][n4][q4 * 4 + 3] + bia) - c3;
                if (live) {
                    uint16_t *cp = p.c + (int64_t)m * p.F + j;
                    cp[0] = f32_to_bf16_bits(x0);
                    cp[(int64_t)p.F] = f32_to_bf16_bits(x1);
                    cp[2 * (int64_t)p.F] = f32_to_bf16_bits(x2);
                    cp[3 * (int64_t)p.F] = f32_to_bf16_bits(x3);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ mm2
struct Mm2Params {
    const uint16_t *a, *b;  // a = packed [M,F], b = fc2^T [F,N2]
    uint16_t *c;            // [M,N2], accumulated in place
    const int32_t *indices, *counts;
    int M, F, N2, NT;
};

__global__ __launch_bounds__(256, 1) void mm2_kernel(const Mm2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *Al = smem;               // [2][A_TILE]  packed activations, k-contiguous
    unsigned char *Bl = smem + 2 * A_TILE;  // [2][B2_TILE] gathered fc2^T rows [64 k][256 n], row stride 512 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int g = wid / p.NT, nt = wid - g * p.NT;
    const int cnt = p.counts[g];
    const int n0 = nt * BN;
    const int ncols = min(BN, p.N2 - n0);  // N2 is a multiple of 8 (checked on the host)
    const int32_t *idxg = p.indices + (int64_t)g * p.F;
    const int nkb = (cnt + BK - 1) / BK;
    if (nkb == 0) return;

    int aoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (lane >> 3);
        aoff[i] = (g * BM + row) * p.F + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    // B rows: one DMA instruction = 2 gathered rows of 512 B (32 chunks); stored chunk c of row r <- source chunk c ^ ((r&3)<<2)
    int keys[8];
    auto load_keys = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kb * BK + (w * 8 + i) * 2 + (lane >> 5);
            keys[i] = idxg[k < cnt ? k : 0];
        }
    };
    auto issue = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(p.a + aoff[i] + kb * BK, Al + buf * A_TILE + (w * 4 + i) * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = (w * 8 + i) * 2 + (lane >> 5);
            int chunk = (lane & 31) ^ ((r & 3) << 2);
            chunk = chunk * 8 < ncols ? chunk : 0;  // partial last column tile: stay inside the row
            glds16(p.b + (int64_t)keys[i] * p.N2 + n0 + chunk * 8, Bl + buf * B2_TILE + (w * 8 + i) * 1024);
        }
    };

    f32x16 acc[4][2];  // [n tile][m tile]: MFMA rows = n (fc2^T via transpose reads), MFMA cols = m
#pragma unroll
    for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n4][mt][r] = 0.f;

    load_keys(0);
    issue(0, 0);
    if (nkb > 1) load_keys(1);
    const int li = lane & 15, grp = lane >> 4;
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < nkb) {
            issue(kb + 1, buf ^ 1);
            if (kb + 2 < nkb) load_keys(kb + 2);
        }
        const unsigned char *At = Al + buf * A_TILE;
        const unsigned char *Bt = Bl + buf * B2_TILE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 pf[2], wf[4];
            const bool kdead = kb * BK + kk * 16 + (lane >> 5) * 8 >= cnt;  // counts are multiples of 8
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bf16x8 z = {};
                const bf16x8 v = read_kfrag(At, wm * 64 + mt * 32 + (lane & 31), kk, lane);
                pf[mt] = kdead ? z : v;  // packed columns past the count hold garbage
            }
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4) {
                // lane group grp: n half = grp&1, k half = grp>>1; lane li addresses block row li>>2, cols (li&3)*4
                const int row = kk * 16 + (grp >> 1) * 8 + (li >> 2);
                const int chunk = (wn * 16 + n4 * 4 + (grp & 1) * 2 + ((li & 3) >> 1)) ^ ((row & 3) << 2);
                const unsigned char *ba = Bt + row * 512 + chunk * 16 + (li & 1) * 8;
                const s16x4 lo = lds_read_tr16_b64(ba);
                const s16x4 hi = lds_read_tr16_b64(ba + 4 * 512);
                wf[n4] = __builtin_bit_cast(
                    bf16x8, (__attribute__((ext_vector_type(8))) short){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
#pragma unroll
            for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[n4][mt] = mfma32(wf[n4], pf[mt], acc[n4][mt]);
        }
    }

    // ---- epilogue: lane owns row m = lane&31 of each tile and 4 consecutive n per accumulator quad
    //      C = bf16(acc) + C in bf16  (triton/csp_mlp_mm2.py:100-101)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = g * BM + wm * 64 + mt * 32 + (lane & 31);
        uint16_t *crow = p.c + (int64_t)m * p.N2;
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int n = n0 + wn * 128 + n4 * 32 + q4 * 8 + (lane >> 5) * 4;
                if (n >= p.N2) continue;
                const u32x2 old = *(const u32x2 *)(crow + n);
                const float a0 = round_bf16(acc[n4][mt][q4 * 4 + 0]), a1 = round_bf16(acc[n4][mt][q4 * 4 + 1]);
                const float a2 = round_bf16(acc[n4][mt][q4 * 4 + 2]), a3 = round_bf16(acc[n4][mt][q4 * 4 + 3]);
                u32x2 out;
                out[0] = pack_bf16x2(a0 + __uint_as_float(old[0] << 16), a1 + __uint_as_float(old[0] & 0xffff0000u));
                out[1] = pack_bf16x2(a2 + __uint_as_float(old[1] << 16), a3 + __uint_as_float(old[1] & 0xffff0000u));
                *(u32x2 *)(crow + n) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ scatter-add
// unpacked[idx[g,c], g*128 + r] += packed[g*128 + r, c]   (scatter_add.cu:43-98).  One workgroup = (group, 64 packed
// columns): the 128x64 packed tile is transposed through LDS so both the read (128 B per row) and the read-modify-write
// of the column-major cache (256 B per column) are full-line accesses.  Every (group, column) pair is owned by exactly
// one workgroup, so no atomics are needed (the reference needs TMA reduce-add only because of its thread mapping).
constexpr int SC_COLS = 64;
constexpr int SC_LD = 136;  // padded row length of the transposed tile (bf16 elements)

__global__ __launch_bounds__(256) void scatter_add_kernel(const uint16_t *packed, uint16_t *unpacked,
                                                          const int32_t *indices, const int32_t *counts, int M, int F) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[SC_COLS * SC_LD];
    const int g = blockIdx.y, c0 = blockIdx.x * SC_COLS;
    const int cnt = counts[g];
    if (c0 >= cnt) return;
    const int tid = threadIdx.x;
    // load: 128 rows x 8 chunks of 16 B
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = i * 256 + tid;
        const int r = item >> 3, ch = item & 7;
        const u32x4 v = *(const u32x4 *)(packed + (int64_t)(g * 128 + r) * F + c0 + ch * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            tile[(ch * 8 + 2 * e) * SC_LD + r] = (uint16_t)(v[e] & 0xffffu);
            tile[(ch * 8 + 2 * e + 1) * SC_LD + r] = (uint16_t)(v[e] >> 16);
        }
    }
    __syncthreads();
    // accumulate: 64 columns x 16 chunks of 8 rows
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = i * 256 + tid;
        const int c = item >> 4, ch = item & 15;
        if (c0 + c >= cnt) continue;
        const int col = indices[(int64_t)g * F + c0 + c];
        uint16_t *dst = unpacked + (int64_t)col * M + g * 128 + ch * 8;
        const u32x4 old = *(const u32x4 *)dst;
        const u32x4 add = *(const u32x4 *)(tile + c * SC_LD + ch * 8);
        u32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[e] = pack_bf16x2(__uint_as_float(old[e] << 16) + __uint_as_float(add[e] << 16),
                                 __uint_as_float(old[e] & 0xffff0000u) + __uint_as_float(add[e] & 0xffff0000u));
        *(u32x4 *)dst = out;
    }
}

int check_mlp_common(int M, int F, const int32_t *indices, const int32_t *counts) {
    CM_CHECK(indices && counts, "mlp: indices / counts missing");
    CM_CHECK(M > 0 && M % BM == 0, "mlp: M must be a positive multiple of 128 (got %d)", M);
    CM_CHECK(F > 0 && F % 64 == 0, "mlp: F must be a positive multiple of 64 (got %d)", F);
    return CHIPMUNK_OK;
}

int launch_scatter_add(const void *packed, void *unpacked, const int32_t *indices, const int32_t *counts, int M, int F,
                       hipStream_t s) {
    hipLaunchKernelGGL(scatter_add_kernel, dim3(F / SC_COLS, M / BM), dim3(256), 0, s, (const uint16_t *)packed,
                       (uint16_t *)unpacked, indices, counts, M, F);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

int launch_mm2(const void *a, const void *b, void *c, const int32_t *indices, const int32_t *counts, int M, int F, int N2,
               hipStream_t s) {
    CM_CHECK(N2 > 0 && N2 % 8 == 0, "mm2: N2 must be a positive multiple of 8 (got %d)", N2);
    CM_CHECK((int64_t)M * F < (1ll << 31), "mm2: M*F too large for 32-bit offsets");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)mm2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
        attr_set = true;
    }
    Mm2Params p = {(const uint16_t *)a, (const uint16_t *)b, (uint16_t *)c, indices, counts, M, F, N2, (N2 + BN - 1) / BN};
    hipLaunchKernelGGL(mm2_kernel, dim3((M / BM) * p.NT), dim3(256), MLP_LDS, s, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

}  // namespace

extern "C" int chipmunk_csp_mlp_mm1(const void *a, const void *b, void *c, const void *bias, const void *pa_cache,
                                    const int32_t *indices, const int32_t *counts, int M, int K, int F, void *stream) {
    CM_CHECK(a && b && c && bias && pa_cache, "csp_mlp_mm1: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    CM_CHECK(K > 0 && K % BK == 0, "csp_mlp_mm1: K must be a positive multiple of 64 (got %d)", K);
    CM_CHECK((int64_t)F * K < (1ll << 31) && (int64_t)M * K < (1ll << 31), "csp_mlp_mm1: operand too large for 32-bit offsets");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)mm1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_LDS);
        attr_set = true;
    }
    Mm1Params p = {(const uint16_t *)a, (const uint16_t *)b, (const uint16_t *)bias, (const uint16_t *)pa_cache,
                   (uint16_t *)c, indices, counts, M, K, F, (F + BN - 1) / BN};
    hipLaunchKernelGGL(mm1_kernel, dim3((M / BM) * p.NT), dim3(256), MLP_LDS, (hipStream_t)stream, p);
    CM_LAUNCH_CHECK();
    return CHIPMUNK_OK;
}

extern "C" int chipmunk_csp_scatter_add(const void *packed, void *unpacked_colmajor, const int32_t *indices,
                                        const int32_t *counts, int M, int F, int num_sms, void *stream) {
    (void)num_sms;
    CM_CHECK(packed && unpacked_colmajor, "csp_scatter_add: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    return launch_scatter_add(packed, unpacked_colmajor, indices, counts, M, F, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_mlp_mm2(const void *mma_a, const void *mma_b, void *mma_c, const int32_t *indices,
                                    const int32_t *counts, int M, int F, int N2, void *stream) {
    CM_CHECK(mma_a && mma_b && mma_c, "csp_mlp_mm2: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    return launch_mm2(mma_a, mma_b, mma_c, indices, counts, M, F, N2, (hipStream_t)stream);
}

extern "C" int chipmunk_csp_mlp_mm2_and_scatter_add(const void *packed, void *unpacked_colmajor,
                                                    const int32_t *indices, const int32_t *counts, const void *mma_a,
                                                    const void *mma_b, void *mma_c, int M, int F, int N2,
                                                    int num_sms_scatter_add, void *stream) {
    (void)num_sms_scatter_add;
    CM_CHECK(packed && unpacked_colmajor && mma_a && mma_b && mma_c, "csp_mlp_mm2_and_scatter_add: null tensor pointer");
    if (int e = check_mlp_common(M, F, indices, counts)) return e;
    if (int e = launch_scatter_add(packed, unpacked_colmajor, indices, counts, M, F, (hipStream_t)stream)) return e;
    return launch_mm2(mma_a, mma_b, mma_c, indices, counts, M, F, N2, (hipStream_t)stream);
}
