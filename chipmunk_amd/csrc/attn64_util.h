// Small device helpers shared by the one-wave-per-SIMD kernels (attn64.hip, attn96.hip).
#pragma once
#include <type_traits>
#include "common.h"

namespace {

template <int V>
using ic = std::integral_constant<int, V>;
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(ic<I>{});
        static_for<I + 1, N>(f);
    }
}

template <typename T>
__device__ __forceinline__ void pin(T &x) {
    asm volatile("" : "+v"(x));
}
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int BASE>
__device__ __forceinline__ void acc_write4(const u32x4 &v) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3\n\ts_nop 1"
                 ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
}
template <int BASE>
__device__ __forceinline__ f32x4 acc_read4() {
    float a, b, c, d;
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "i"(BASE), "i"(BASE + 1), "i"(BASE + 2), "i"(BASE + 3));
    return (f32x4){a, b, c, d};
}


}  // namespace
