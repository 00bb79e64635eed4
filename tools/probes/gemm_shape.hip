// Probe: ceiling of the GEMM main loop (LDS-DMA stream + ds_read_b128 operand fetch + 32x32x16 bf16 MFMA) for several
// workgroup shapes on gfx950, with the column-sparse GEMM1 access pattern (AROWS contiguous rows + BROWS gathered rows
// of a row-major bf16 matrix with 6 KiB rows).  No epilogue, no correctness: it answers "which loop shape is worth
// building" before building it.  Data is random (all-zero operands run at unrealistic clocks).
//   template <AROWS, BROWS, BKB (bytes of K per row per step), NW (waves), WMG x WNG (wave grid), NST, WPC (WGs / CU)>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

constexpr int K = 3072, ROWB = K * 2;

template <int AROWS, int BROWS, int BKB, int NW, int WMG, int WNG, int NST, int WPC>
__device__ __forceinline__ void gemm_loop_body(const unsigned char *mat, const int *idx, int nrows, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ROWS = AROWS + BROWS, STAGE = ROWS * BKB;
    constexpr int RPI = 1024 / BKB;                 // rows per 1 KiB DMA instruction
    constexpr int INST = ROWS / RPI / NW;           // DMA instructions per wave per step
    constexpr int CPR = BKB / 16;                   // 16-byte chunks per row
    constexpr int KK = BKB / 32;                    // 16-element k slices per step
    constexpr int MT = AROWS / WMG / 32, NT = BROWS / WNG / 32;
    constexpr int NKB = ROWB / BKB;
    static_assert(INST >= 1 && WMG * WNG == NW, "shape");
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / WNG, wn = w % WNG;
    const int wg = blockIdx.x;
    uint32_t off[INST];
#pragma unroll
    for (int i = 0; i < INST; ++i) {
        const int r = (w * INST + i) * RPI + lane / CPR;
        const int row = r < AROWS ? (wg % (nrows / AROWS)) * AROWS + r : idx[(wg * BROWS + (r - AROWS)) % nrows];
        off[i] = row * ROWB + (((lane % CPR) ^ ((r / (256 / BKB)) & (CPR - 1))) << 4);
    }
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)mat, 0, 0xffffffff, 0x00020000);
    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto issue = [&](int kb, int buf) {
#pragma unroll
        for (int i = 0; i < INST; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(smem + buf * STAGE + (w * INST + i) * 1024), 16, off[i], kb * BKB, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) issue(s, s);
    int buf = 0, nbuf = NST - 1;
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + NST - 1 <= NKB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * INST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kb + NST - 1 < NKB) issue(kb + NST - 1, nbuf);
        const unsigned char *st = smem + buf * STAGE;
        auto frag = [&](int row, int kk) {
            const int c = kk * 2 + (lane >> 5);
            return *(const bf16x8 *)(st + row * BKB + ((c ^ ((row / (256 / BKB)) & (CPR - 1))) << 4));
        };
        bf16x8 fa[2][MT], fb[2][NT];
        auto load = [&](int kk, int set) {
#pragma unroll
            for (int a = 0; a < MT; ++a) fa[set][a] = frag(wm * (AROWS / WMG) + a * 32 + (lane & 31), kk);
#pragma unroll
            for (int b = 0; b < NT; ++b) fb[set][b] = frag(AROWS + wn * (BROWS / WNG) + b * 32 + (lane & 31), kk);
        };
        load(0, 0);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            if (kk + 1 < KK) load(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][a], fb[kk & 1][b], acc[a][b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = buf + 1 == NST ? 0 : buf + 1;
        nbuf = nbuf + 1 == NST ? 0 : nbuf + 1;
    }
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) t += acc[a][b][r];
    if (t == 123.456f) sink[0] = t;
}

template <int AROWS, int BROWS, int BKB, int WMG, int WNG, int NST, int WPC>
__global__ __launch_bounds__(256, WPC) void gemm_loop4(const unsigned char *mat, const int *idx, int nrows, float *sink) {
    gemm_loop_body<AROWS, BROWS, BKB, 4, WMG, WNG, NST, WPC>(mat, idx, nrows, sink);
}
template <int AROWS, int BROWS, int BKB, int WMG, int WNG, int NST, int WPC>
__global__ __launch_bounds__(512, WPC) void gemm_loop8(const unsigned char *mat, const int *idx, int nrows, float *sink) {
    gemm_loop_body<AROWS, BROWS, BKB, 8, WMG, WNG, NST, WPC>(mat, idx, nrows, sink);
}

template <int AROWS, int BROWS, int BKB, int NW, int WMG, int WNG, int NST, int WPC>
void run(const char *name, const unsigned char *mat, const int *idx, int nrows, float *sink) {
    void (*k)(const unsigned char *, const int *, int, float *);
    if constexpr (NW == 8) k = gemm_loop8<AROWS, BROWS, BKB, WMG, WNG, NST, WPC>;
    else k = gemm_loop4<AROWS, BROWS, BKB, WMG, WNG, NST, WPC>;
    const int lds = NST * (AROWS + BROWS) * BKB;
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    // two exact rounds of resident workgroups: steady-state throughput without tail quantisation
    int occ0 = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, k, NW * 64, lds);
    const int nwg = 2 * 256 * (occ0 > 0 ? occ0 : 1);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, 0, mat, idx, nrows, sink);
    (void)hipEventRecord(a);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(nwg), dim3(NW * 64), lds, 0, mat, idx, nrows, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double flops = 2.0 * AROWS * BROWS * 3072.0 * nwg;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, NW * 64, lds);
    printf("%-52s lds %3d KB  WG/CU %d  WGs %4d  %8.1f us  %7.1f TFLOP/s\n", name, lds >> 10, occ, nwg, ms * 1e3, flops / ms / 1e9);
}

int main() {
    const int nrows = 16384;
    unsigned char *mat;
    int *idx;
    float *sink;
    (void)hipMalloc(&mat, (size_t)nrows * ROWB);
    {
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
        (void)hipMemcpy(mat, hm, n * 2, hipMemcpyHostToDevice);
        free(hm);
    }
    (void)hipMalloc(&idx, nrows * 4);
    (void)hipMalloc(&sink, 4);
    int *h = (int *)malloc(nrows * 4);
    srand(1);
    for (int i = 0; i < nrows; ++i) h[i] = rand() % nrows;
    (void)hipMemcpy(idx, h, nrows * 4, hipMemcpyHostToDevice);
#define RUN(...) run<__VA_ARGS__>(#__VA_ARGS__, mat, idx, nrows, sink)
    //  AROWS BROWS BKB NW WMG WNG NST WPC
    RUN(128, 128, 128, 4, 2, 2, 2, 2);   // shipped GEMM1 shape
    RUN(128, 128, 128, 4, 2, 2, 3, 1);
    RUN(128, 128, 64, 4, 2, 2, 4, 2);
    RUN(128, 128, 64, 4, 2, 2, 3, 3);
    RUN(128, 256, 64, 4, 2, 2, 3, 2);    // shipped GEMM2 tile shape
    RUN(128, 256, 64, 4, 2, 2, 2, 3);
    RUN(128, 256, 128, 4, 2, 2, 2, 1);
    RUN(128, 256, 128, 4, 2, 2, 3, 1);
    RUN(128, 256, 128, 8, 2, 4, 3, 1);   // 8 waves, each 64x64
    RUN(128, 256, 64, 8, 2, 4, 3, 2);
    RUN(128, 256, 64, 8, 2, 4, 4, 2);
    RUN(128, 512, 64, 8, 2, 4, 3, 1);    // 8 waves, each 64x128
    RUN(128, 512, 64, 8, 2, 4, 2, 1);
    RUN(128, 512, 32, 8, 2, 4, 4, 1);
    RUN(128, 384, 64, 4, 2, 2, 3, 1);    // 4 waves, each 64x192
    return 0;
}
