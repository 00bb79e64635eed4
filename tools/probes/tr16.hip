// Probe: prints the lane/element mapping of ds_read_b64_tr_b16 and of the 16x16x32 bf16 MFMA C layout on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(int* out, float* cout) {
    __shared__ __attribute__((aligned(16))) short lds[4 * 64];   // 4 rows x 64 cols (values = row*1000 + col)
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)((i / 64) * 1000 + (i % 64));
    __syncthreads();
    int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    // lane li of group g addresses block row (li>>2), cols g*16 + (li&3)*4
    const short* addr = &lds[(li >> 2) * 64 + g * 16 + (li & 3) * 4];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
    // MFMA C layout: A[m][k] = (m==3 && k==5) ? 1 : 0 ; B[k][n] = (k==5) ? n+1 : 0  => C[3][n] = n+1
    bf16x8 a = {}, b = {};
    for (int j = 0; j < 8; ++j) {
        int kk = g * 8 + j;
        a[j] = (__bf16)((li == 3 && kk == 5) ? 1.0f : 0.0f);
        b[j] = (__bf16)((kk == 5) ? (float)(li + 1) : 0.0f);
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) cout[lane * 4 + j] = c[j];
}
int main() {
    int* d; float* c; hipMalloc(&d, 256 * 4); hipMalloc(&c, 256 * 4);
    k<<<1, 64>>>(d, c);
    int h[256]; float hc[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    printf("tr16: lane -> 4 values (row*1000+col); expect lane li of group g: rows 0..3 of col g*16+li\n");
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d: %5d %5d %5d %5d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
        for (int j = 0; j < 4; ++j) if (h[l*4+j] != j * 1000 + (l >> 4) * 16 + (l & 15)) ok = 0;
    }
    printf("TR16_EXPECTED_MAPPING=%d\n", ok);
    int okc = 1;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int row = (l >> 4) * 4 + j, col = l & 15;
        float want = row == 3 ? (float)(col + 1) : 0.f;
        if (hc[l*4+j] != want) okc = 0;
    }
    printf("MFMA16_C_LAYOUT_EXPECTED=%d\n", okc);
    return 0;
}
