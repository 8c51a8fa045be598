// calibration: sustained v_mfma_f32_16x16x4_f32 rate and shader clock on this box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out, int iters, unsigned long long *clk) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
        a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
        a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
    float *out; unsigned long long *clk, h[2];
    for (int wpc : {4, 8}) {
        int blocks = 256 * wpc / 4, threads = 256;
        hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, 16);
        for (int iters : {2000, 20000}) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            k<<<blocks, threads>>>(out, iters, clk); hipDeviceSynchronize();
            hipEventRecord(a); k<<<blocks, threads>>>(out, iters, clk); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double flop = 2048.0 * 8 * iters * blocks * (threads / 64);
            printf("waves/CU %d iters %d: %.3f ms  %.1f TFLOP/s  shader clock %.2f GHz (cycles %llu, cyc/MFMA/SIMD %.1f)\n", wpc, iters,
                   ms, flop / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0) , h[0], (double)h[0] / (8.0 * iters) / (wpc / 4.0));
        }
    }
    return 0;
}
