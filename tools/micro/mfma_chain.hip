// microbenchmark: issue-to-issue time of DEPENDENT v_mfma_f32_16x16x32_bf16 (same accumulator) vs independent ones
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
    const f32x4 a4 = {a, b, a, b}, b4 = {b, a, b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 12; ++m)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m % CHAINS]) : "v"(a4), "v"(b4));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0;
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[256 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(float *d, int waves) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS>), dim3(256), dim3(waves * 64), 0, 0, d, 100);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<CHAINS>), dim3(256), dim3(waves * 64), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chains=%d waves/SIMD=%d: %.1f ns per MFMA per SIMD\n", CHAINS, waves / 4, ms * 1e6 / (iters * 12.0) / (waves / 4.0));
}
int main() {
    float *d;
    (void)hipMalloc(&d, (256 + 256 * 512) * sizeof(float));
    (void)hipMemset(d, 0, (256 + 256 * 512) * sizeof(float));
    for (int waves : {4, 8}) { run<1>(d, waves); run<2>(d, waves); run<3>(d, waves); run<4>(d, waves); }
    return 0;
}
