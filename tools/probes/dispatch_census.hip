// Which workgroups of a one-round launch end up ALONE on their CU?  408 workgroups of 512 threads with 72 KiB of LDS each (GEMM2's launch shape:
// two fit a CU, 512 slots) spin for a fixed time and record (XCC id, HW id, start); the host prints, per XCD, the local dispatch indices
// (blockIdx >> 3) of the workgroups that had a CU to themselves.  Speed heuristic only -- HIP promises nothing about placement.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/dispatch_census tools/probes/dispatch_census.hip ; run: /tmp/dispatch_census [nblocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(512) void census(unsigned *out, long long spin) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        out[blockIdx.x * 4 + 0] = xcc & 0xf;
        out[blockIdx.x * 4 + 1] = hw;
        out[blockIdx.x * 4 + 2] = (unsigned)t0;
        smem[0] = 1;
        while (__builtin_amdgcn_s_memrealtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}
int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 408;
    unsigned *d;
    hipMalloc(&d, n * 16);
    hipFuncSetAttribute((const void *)census, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    std::vector<unsigned> h(n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(census, dim3(n), dim3(512), 72 * 1024, 0, d, 3000LL);   // 30 us at 100 MHz
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<int>> percu;   // (xcc, se, cu) -> blocks
        int xcd_mismatch = 0;
        for (int b = 0; b < n; ++b) {
            const unsigned xcc = h[b * 4], hw = h[b * 4 + 1];
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            percu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
            xcd_mismatch += (xcc != (unsigned)(b & 7));
        }
        printf("launch %d: %d workgroups on %zu distinct CUs; block b on XCD b %% 8 violated %d times\n", rep, n, percu.size(), xcd_mismatch);
        std::map<int, std::vector<int>> lone;   // xcd -> local indices of lone workgroups
        std::map<int, std::vector<std::pair<int, int>>> pairs;
        for (auto &kv : percu) {
            if (kv.second.size() == 1) lone[kv.second[0] & 7].push_back(kv.second[0] >> 3);
            else if (kv.second.size() == 2) pairs[kv.second[0] & 7].push_back({kv.second[0] >> 3, kv.second[1] >> 3});
        }
        for (int x = 0; x < 2; ++x) {
            printf("  XCD %d lone local indices:", x);
            for (int l : lone[x]) printf(" %d", l);
            printf("\n  XCD %d pairs:", x);
            for (auto &pr : pairs[x]) printf(" (%d,%d)", pr.first, pr.second);
            printf("\n");
        }
    }
    return 0;
}
