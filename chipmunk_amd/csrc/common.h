// Shared device helpers for the gfx950 (CDNA4) kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/chipmunk_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// 16-byte async global -> LDS copy.  LDS destination = wave-uniform base + lane*16 (lane-linear);
// the global source address is per lane, which is what makes it a gather engine.
__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

// The same copy in buffer form: SGPR resource {base, 4 GiB range} + 32-bit per-lane byte offset + wave-uniform SGPR byte
// offset.  No 64-bit per-lane address arithmetic, and measurably cheaper to issue next to MFMAs than the global form
// (tools/probes/fill_rate.hip: GEMM-shaped stream + 16 MFMA per step, 107 -> 95 us).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xffffffff, 0x00020000);
}
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off_bytes, uint32_t wave_off_bytes,
                                       void *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_wave_base), 16, (int)lane_off_bytes, (int)wave_off_bytes, 0, 0);
}

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even float -> bf16 bits (matches torch / the oracle's f2bf for finite values)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float round_bf16(float f) { return bf16_bits_to_f32(f32_to_bf16_bits(f)); }
// round two floats to bf16 with the hardware converter (v_cvt_pk_bf16_f32, RNE) and widen them back
__device__ __forceinline__ void round_bf16_pair(float &a, float &b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    a = __uint_as_float(u << 16);
    b = __uint_as_float(u & 0xffff0000u);
}
// round-to-nearest-even to bf16 precision without the NaN special case (3 integer VALU ops); inf stays inf
__device__ __forceinline__ float round_bf16_fast(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ float hw_round_bf16(float x) {
    const __bf16 h = (__bf16)x;
    return __uint_as_float(((uint32_t)__builtin_bit_cast(uint16_t, h)) << 16);
}
// pack two floats to a dword of two bf16 (lo in bits 0-15)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
}

// hardware transpose read: each 16-lane group reads a 4x16 b16 block (lane i supplies the address of
// 4 contiguous elements: block row i/4, columns (i%4)*4..+3) and lane i receives column i (4 rows).
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const void *lds_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds_addr));
}

// 16-lane-row rotate (DPP row_ror:n) -- cross-lane without LDS traffic.
template <int N>
__device__ __forceinline__ float dpp_row_ror(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + N, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row; every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float x) {
    x += dpp_row_ror<8>(x);
    x += dpp_row_ror<4>(x);
    x += dpp_row_ror<2>(x);
    x += dpp_row_ror<1>(x);
    return x;
}

// max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with two gfx950 lane-swap VALU ops instead of
// two ds_bpermute round trips: permlane32_swap(x, x) yields {lo,lo} and {hi,hi}; permlane16_swap pairs rows 0/1, 2/3.
// The swaps are issued through inline asm: with the builtin, hipcc (ROCm 7.2) folds the two results of
// __builtin_amdgcn_permlane{16,32}_swap into one when both operands carry the same value and silently drops the
// reduction (seen in the .s: the fmax of the two results disappears).  Two "+v" operands force two registers; the
// leading s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see inside the string.
__device__ __forceinline__ void lane_swap32(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap16(float &a, float &b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float max_across_rows(float x) {
    float a = x, b = x;
    lane_swap32(a, b);   // a = {lo, lo}, b = {hi, hi}
    float c = fmaxf(a, b), d = c;
    lane_swap16(c, d);   // c = {r0, r0, r2, r2}, d = {r1, r1, r3, r3}
    return fmaxf(c, d);
}
__device__ __forceinline__ float sum_across_rows(float x) {
    float a = x, b = x;
    lane_swap32(a, b);
    float c = a + b, d = c;
    lane_swap16(c, d);
    return c + d;
}

// XCD-aware remap of a 1-D grid (8 XCDs, block b is dispatched to XCD b % 8): returns an id such that every
// XCD works on a contiguous chunk of the logical index space (L2 affinity only, never correctness).  Bijective.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int NX = 8;
    int xcd = bid % NX, slot = bid / NX;
    int q = nblocks / NX, r = nblocks % NX;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// compute units of the current device (all devices of a node are the same part); 256 on MI355X
inline int device_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

// Dynamic LDS above 64 KiB has to be requested per kernel AND per device; `done` is the caller's static per-kernel bit set
// (bit = device ordinal), so a process that drives several devices asks once on each.
inline void ensure_dynamic_lds(const void *kernel, int bytes, uint64_t &done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 0;
    if (!(done >> dev & 1)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        done |= uint64_t(1) << dev;
    }
}

// ---- host-side error plumbing (thread-local message, see chipmunk_last_error) ----
void chipmunk_set_error(const char *fmt, ...);
int chipmunk_get_option(const char *name);
uint32_t chipmunk_next_random_salt();  // per-launch salt of the random-key hash (capi.hip)
// zero-initialised, grow-only device scratch owned by the library, one per (device, stream); nullptr on failure
void *chipmunk_scratch(hipStream_t stream, size_t bytes);
void *chipmunk_big_scratch(hipStream_t stream, size_t bytes);   // separate multi-GB buffer, not zeroed; nullptr if unavailable
// attn64.hip: dense attention, one wave per SIMD (see there); strides in elements, [batch, head, row]
int chipmunk_dense64_launch(const void *q, const void *k, const void *v, void *o, float *l, const int64_t qs[3],
                            const int64_t ks[3], const int64_t vs[3], const int64_t os[3], int B, int H, int Nq, int Nk,
                            hipStream_t stream);
#define CM_CHECK(cond, ...)                    \
    do {                                       \
        if (!(cond)) {                         \
            chipmunk_set_error(__VA_ARGS__);   \
            return CHIPMUNK_ERR_INVALID;       \
        }                                      \
    } while (0)
#define CM_LAUNCH_CHECK()                                                          \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            chipmunk_set_error("HIP launch failed: %s", hipGetErrorString(e__));   \
            return CHIPMUNK_ERR_LAUNCH;                                            \
        }                                                                          \
    } while (0)
