// probe: fp32 GEMM tile C(16x16) = A(16xK) B(Kx16) on (a) v_mfma_f32_16x16x4_f32 and (b) three
// v_mfma_f32_16x16x16_bf16 per K=16 with the operands split x = hi + lo in bf16 (hi*hi + hi*lo + lo*hi, fp32
// accumulate).  Reports the error of both against an fp64 reference and the sustained rates.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// one wave per tile; A row-major [16][K], B row-major [K][16]
__global__ void gemm_f32(const float *A, const float *B, int K, float *C) {
    const int l = threadIdx.x, i = l & 15, kq = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + 4 * s + kq], B[(4 * s + kq) * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(kq * 4 + r) * 16 + i] = acc[r];
}
__global__ void gemm_bf16x3(const float *A, const float *B, int K, float *C) {
    const int l = threadIdx.x, i = l & 15, kq = l >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 16; ++s) {
        s16x4 ah, al, bh, bl;
        for (int j = 0; j < 4; ++j) {
            const int k = 16 * s + 4 * kq + j;
            const float a = A[i * K + k], b = B[k * 16 + i];
            const unsigned short a_h = bf16_rne(a), b_h = bf16_rne(b);
            ah[j] = (short)a_h; al[j] = (short)bf16_rne(a - bf16_f(a_h));
            bh[j] = (short)b_h; bl[j] = (short)bf16_rne(b - bf16_f(b_h));
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[(kq * 4 + r) * 16 + i] = acc[r];
}
// rate: 8 independent accumulators
__global__ void rate_bf16(float *out, int iters) {
    f32x4 a[8];
    for (int q = 0; q < 8; ++q) a[q] = f32x4{0, 0, 0, 0};
    s16x4 x = {(short)threadIdx.x, 1, 2, 3}, y = {(short)blockIdx.x, 3, 2, 1};
    for (int i = 0; i < iters; ++i)
        for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, a[q], 0, 0, 0);
    f32x4 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ void rate_f32(float *out, int iters) {
    f32x4 a[8];
    for (int q = 0; q < 8; ++q) a[q] = f32x4{0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i)
        for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[q], 0, 0, 0);
    f32x4 s = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    const int K = 112;                               // ~ 9 taps x 12 channels, multiple of 16
    std::vector<float> A(16 * K), B(K * 16), C(256);
    srand(1);
    for (auto &v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;       // weights ~ U(-0.3, 0.3)
    for (auto &v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;        // activations ~ U(-2, 2)
    std::vector<double> R(256, 0.0), Rabs(256, 0.0);
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j)
            for (int k = 0; k < K; ++k) {
                R[i * 16 + j] += (double)A[i * K + k] * B[k * 16 + j];
                Rabs[i * 16 + j] += fabs((double)A[i * K + k] * B[k * 16 + j]);
            }
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) gemm_f32<<<1, 64>>>(dA, dB, K, dC); else gemm_bf16x3<<<1, 64>>>(dA, dB, K, dC);
        hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost);
        double e_abs = 0, e_rel = 0;
        for (int q = 0; q < 256; ++q) {
            e_abs = fmax(e_abs, fabs(C[q] - R[q]));
            e_rel = fmax(e_rel, fabs(C[q] - R[q]) / Rabs[q]);               // relative to sum |a b| (conditioning-free)
        }
        printf("%s: max abs err %.3e, max err / sum|ab| %.3e\n", mode ? "bf16x3 (3 MFMA per K=16)" : "fp32 16x16x4        ", e_abs, e_rel);
    }
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        const int blocks = 512, threads = 256, iters = 20000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        if (mode) rate_bf16<<<blocks, threads>>>(out, iters); else rate_f32<<<blocks, threads>>>(out, iters);
        hipDeviceSynchronize();
        hipEventRecord(a);
        if (mode) rate_bf16<<<blocks, threads>>>(out, iters); else rate_f32<<<blocks, threads>>>(out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double n_mfma = 8.0 * iters * blocks * (threads / 64);
        const double kflop = mode ? 16.0 * 16 * 16 * 2 : 16.0 * 16 * 4 * 2;
        printf("%s: %.3f ms, %.1f TFLOP/s raw, %.2f ns per MFMA per SIMD -> fp32-equivalent GEMM rate %.1f TFLOP/s\n",
               mode ? "bf16 16x16x16" : "fp32 16x16x4 ", ms, n_mfma * kflop / ms / 1e9, ms * 1e6 / (n_mfma / 1024.0),
               mode ? n_mfma * kflop / 3.0 / ms / 1e9 : n_mfma * kflop / ms / 1e9);
    }
    return 0;
}
