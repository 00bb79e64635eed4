// Probe: how fast can a GEMM-shaped workgroup stream [256 rows x 128 B] k-slices from L2 into LDS on gfx950, and what
// does the streaming cost the MFMA pipe?  Same access pattern as csp_mlp_mm1 (128 contiguous rows + 128 gathered rows
// of a row-major bf16 matrix with 6 KiB rows, 48 k-steps).
//   mode 0: global_load_lds_dwordx4 (LDS-DMA, 64-bit per-lane addresses)
//   mode 1: buffer_load_dwordx4 ... lds (LDS-DMA, SGPR base + 32-bit per-lane offset)
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128 (one step of register prefetch)
//   mode 3: buffer_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 4: buffer_load_dwordx4 -> VGPR only (the L1 -> register path alone)
// usage: fill_rate [mfma_per_step]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

constexpr int K = 3072, ROWB = K * 2, NKB = 48, STAGE = 32768;

template <int MODE, int NST, int NMFMA, int RD = 0>
__global__ __launch_bounds__(256, 2) void fill(const unsigned char *mat, const int *idx, int nrows, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = blockIdx.x;
    int off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = (w * 8 + i) * 8 + (lane >> 3);  // 0..255
        const int row = r < 128 ? (wg % (nrows / 128)) * 128 + r : idx[(wg * 128 + (r - 128)) % (nrows)];
        off[i] = row * ROWB + (lane & 7) * 16;
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)mat, 0, 0x7fffffff, 0x00020000);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 fa = {}, fb = {};
    fa[0] = (__bf16)(float)lane;
    fb[1] = (__bf16)1.0f;

    if constexpr (MODE < 2) {
        auto issue = [&](int kb, int buf) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                unsigned char *dst = smem + buf * STAGE + (w * 8 + i) * 1024;
                if constexpr (MODE == 0)
                    __builtin_amdgcn_global_load_lds(GLB_PTR(mat + off[i] + kb * 128), LDS_PTR(dst), 16, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(dst), 16, off[i], kb * 128, 0, 0);
            }
        };
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue(s, s);
        int buf = 0, nbuf = NST - 1;
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + NST - 1 <= NKB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * 8) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kb + NST - 1 < NKB) issue(kb + NST - 1, nbuf);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RD == 0) {
#pragma unroll
                for (int j = 0; j < NMFMA; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
            } else {
                // the GEMM's operand traffic: per 16-wide k slice 2 A + 2 B fragments (ds_read_b128), 4 MFMAs;
                // RD 1: fragments of slice kk+1 requested before the MFMAs of slice kk; RD 2: all 16 reads up front
                const unsigned char *st = smem + buf * STAGE;
                auto frag = [&](int which, int kk) {
                    const int row = (which & 1) * 32 + (w >> 1) * 64 + (lane & 31) + (which >> 1) * 128;
                    const int c = kk * 2 + (lane >> 5);
                    return *(const bf16x8 *)(st + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
                };
                bf16x8 f[RD == 2 ? 4 : 2][4];
                if constexpr (RD == 2) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int q = 0; q < 4; ++q) f[kk][q] = frag(q, kk);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) f[0][q] = frag(q, 0);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if constexpr (RD == 1) {
                        if (kk + 1 < 4) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) f[(kk + 1) & 1][q] = frag(q, kk + 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int s = RD == 2 ? kk : (kk & 1);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s][0], f[s][2], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s][0], f[s][3], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s][1], f[s][2], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[s][1], f[s][3], acc[3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            buf = buf + 1 == NST ? 0 : buf + 1;
            nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
        }
    } else {
        u32x4 regs[8];
        auto gload = [&](int kb) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 2) regs[i] = *(const u32x4 *)(mat + off[i] + kb * 128);
                else regs[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off[i], kb * 128, 0));
            }
        };
        auto lwrite = [&](int buf) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (MODE == 4) { if (regs[i][0] == 0x12345u && regs[i][3] == 7u) sink[1] = 1.f; }   // mode 4: loads only, no LDS write
                else *(u32x4 *)(smem + buf * STAGE + (w * 8 + i) * 1024 + lane * 16) = regs[i];
            }
        };
        gload(0);
        lwrite(0);
        gload(1);
        for (int kb = 0; kb < NKB; ++kb) {
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NMFMA; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[j & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kb + 1 < NKB) lwrite((kb + 1) & 1);
            if (kb + 2 < NKB) gload(kb + 2);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) t += acc[j][0] + acc[j][5];
    t += *(const float *)(smem + tid * 4);
    if (t == 123.456f) sink[0] = t;
}

template <int MODE, int NST, int NMFMA, int RD = 0>
void run(const char *name, const unsigned char *mat, const int *idx, int nrows, float *sink, int nwg) {
    auto k = fill<MODE, NST, NMFMA, RD>;
    const int lds = (MODE < 2 ? NST : 2) * STAGE;
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, mat, idx, nrows, sink);
    hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, mat, idx, nrows, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)nwg * NKB * STAGE;
    const double flops = (double)nwg * NKB * 4 * NMFMA * 32768.0;
    printf("%-44s lds %3d KB  %8.1f us  %6.2f TB/s into LDS  %7.1f TFLOP/s dummy MFMA\n", name, lds >> 10, ms * 1e3,
           bytes / ms / 1e9, flops / ms / 1e9);
}

int main(int argc, char **argv) {
    const int nrows = 16384;  // 96 MiB matrix: L2-missing, Infinity-Cache resident like fc1 + activations
    unsigned char *mat;
    int *idx;
    float *sink;
    hipMalloc(&mat, (size_t)nrows * ROWB);
    {   // random bf16 in (-1, 1): all-zero operands would let the MFMA pipe run at unrealistically high clocks
        const size_t n = (size_t)nrows * K;
        uint16_t *hm = (uint16_t *)malloc(n * 2);
        uint32_t st = 12345u;
        for (size_t i = 0; i < n; ++i) {
            st = st * 1664525u + 1013904223u;
            const float f = ((st >> 8) & 0xffff) / 32768.0f - 1.0f;
            uint32_t bits;
            memcpy(&bits, &f, 4);
            hm[i] = (uint16_t)(bits >> 16);
        }
        hipMemcpy(mat, hm, n * 2, hipMemcpyHostToDevice);
        free(hm);
    }
    hipMalloc(&idx, nrows * 4);
    hipMalloc(&sink, 8);
    int *h = (int *)malloc(nrows * 4);
    srand(1);
    for (int i = 0; i < nrows; ++i) h[i] = rand() % nrows;
    hipMemcpy(idx, h, nrows * 4, hipMemcpyHostToDevice);
    const int nwg = 1024;
#define RUN(M, N, F) run<M, N, F>("mode " #M " NST " #N " mfma/step " #F, mat, idx, nrows, sink, nwg)
    RUN(0, 2, 0); RUN(1, 2, 0); RUN(2, 2, 0); RUN(3, 2, 0); RUN(4, 2, 0);
    RUN(0, 3, 0); RUN(1, 3, 0);
    RUN(0, 2, 16); RUN(1, 2, 16); RUN(2, 2, 16); RUN(3, 2, 16);
    RUN(0, 3, 16); RUN(1, 3, 16);
    RUN(0, 2, 32); RUN(1, 2, 32); RUN(2, 2, 32); RUN(3, 2, 32);
#define RUNR(M, N, R) run<M, N, 16, R>("mode " #M " NST " #N " 16 mfma fed by ds_read, RD " #R, mat, idx, nrows, sink, nwg)
    RUNR(0, 2, 1); RUNR(1, 2, 1); RUNR(0, 2, 2); RUNR(1, 2, 2);
    return 0;
}
