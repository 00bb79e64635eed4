// Probe: semantics of v_permlane32_swap / v_permlane16_swap as exposed by the hipcc builtins (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    unsigned x = threadIdx.x;
    auto a = __builtin_amdgcn_permlane32_swap(x, x + 100, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(x, x + 100, false, false);
    out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64 * 16);
    k<<<1, 64>>>(d);
    unsigned h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("lane: p32.out0 p32.out1 | p16.out0 p16.out1   (inputs: vdst = lane, src = lane + 100)\n");
    for (int l = 0; l < 64; l += 1) printf("%2d: %3u %3u | %3u %3u\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
