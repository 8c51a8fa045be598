// microbenchmark: how many independent VALU instructions issue in the shadow of one v_mfma_f32_16x16x4_f32 (8 passes,
// 32 cycles), with 1 and 2 waves per SIMD?  Each loop iteration = 4 MFMAs (independent accumulators), each followed by
// J independent v_fma_f32 (8 chains), order pinned with sched_group_barrier.  Prints shader cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int J, bool LDS, int KIND>
__global__ __launch_bounds__(512) void k(float *out, long long *cyc, int iters) {
    __shared__ float sm[4096];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sm[i] = i;
    __syncthreads();
    const float a = out[threadIdx.x & 63], b = out[64 + (threadIdx.x & 63)];
    const float c1 = out[128], c2 = out[129];
    const f32x4 a4 = {a, b, a, b}, b4 = {b, a, b, a};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 a2 = {a, b}, b2 = {b, a};
    const unsigned addr = (threadIdx.x & 63) * 4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // (same accumulator every 4th MFMA: three 8-pass MFMAs in between, no software wait states needed)
            if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a4), "v"(b4));
            if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a2), "v"(b2));
            if (KIND == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a4), "v"(b4));
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int q = (m * J + j) & 7;
                if (LDS) {
                    float t;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(t) : "v"(addr + 256 * q));
                    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                    x[q] = t;
                } else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q]) : "v"(c1), "v"(c2));
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0;
    for (int j = 0; j < 8; ++j) s += x[j];
    for (int m = 0; m < 4; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
    out[256 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int J, bool LDS, int KIND = 0>
void run(float *d, long long *c, int waves) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<J, LDS, KIND>), dim3(256), dim3(waves * 64), 0, 0, d, c, 100);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<J, LDS, KIND>), dim3(256), dim3(waves * 64), 0, 0, d, c, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h; (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    const double per_mfma_ns = ms * 1e6 / (iters * 4.0) / (waves / 4.0);   // per MFMA per SIMD (all its waves)
    static const char *kinds[] = {"f32 16x16x4", "bf16 16x16x32", "bf16 16x16x16", "f16 16x16x32"};
    printf("%-14s %s J=%2d waves/SIMD=%d: %.3f ms  %.1f ns per MFMA per SIMD   clock64 ticks per MFMA (wave 0) %.1f\n",
           kinds[KIND], LDS ? "ds_read" : "v_fma  ", J, waves / 4, ms, per_mfma_ns, (double)h / (iters * 4.0));
}

int main() {
    float *d; long long *c;
    (void)hipMalloc(&d, (256 + 256 * 512) * sizeof(float));
    (void)hipMemset(d, 0, (256 + 256 * 512) * sizeof(float));
    (void)hipMalloc(&c, 8);
    for (int waves : {4, 8}) {
        run<0, false>(d, c, waves); run<1, false>(d, c, waves); run<2, false>(d, c, waves); run<4, false>(d, c, waves);
        run<6, false>(d, c, waves); run<8, false>(d, c, waves); run<12, false>(d, c, waves); run<16, false>(d, c, waves);
        run<1, true>(d, c, waves); run<2, true>(d, c, waves); run<4, true>(d, c, waves);
        run<0, false, 1>(d, c, waves); run<2, false, 1>(d, c, waves); run<4, false, 1>(d, c, waves); run<6, false, 1>(d, c, waves);
        run<8, false, 1>(d, c, waves); run<12, false, 1>(d, c, waves); run<2, true, 1>(d, c, waves);
        run<0, false, 2>(d, c, waves); run<2, false, 2>(d, c, waves); run<4, false, 2>(d, c, waves); run<8, false, 2>(d, c, waves);
        run<0, false, 3>(d, c, waves); run<4, false, 3>(d, c, waves); run<8, false, 3>(d, c, waves);
    }
    return 0;
}
