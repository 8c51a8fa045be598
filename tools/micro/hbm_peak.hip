// hbm_peak: what a plain streaming kernel reaches on this GPU -- write-only (float4 stores), read-only (float4 loads summed)
// and copy -- on a 512 MiB buffer (beyond the 256 MiB Infinity Cache): the practical ceilings behind the `frac_hbm` figures
// of adj_build (write-bound) and spatial_agg (read-bound).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_peak tools/micro/hbm_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void fill(float4 *p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ __launch_bounds__(256) void rsum(const float4 *p, size_t n, float *out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { const float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void copy(const float4 *a, float4 *b, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
int main() {
    const size_t bytes = 512ull << 20, n = bytes / 16;
    float4 *a, *b; float *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 32768}) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0.f;
            for (int w = 0; w < 2; ++w) {
                CK(hipEventRecord(e0, 0));
                for (int r = 0; r < 10; ++r) {
                    if (k == 0) hipLaunchKernelGGL(fill, dim3(grid), dim3(256), 0, 0, a, n);
                    if (k == 1) hipLaunchKernelGGL(rsum, dim3(grid), dim3(256), 0, 0, a, n, o);
                    if (k == 2) hipLaunchKernelGGL(copy, dim3(grid), dim3(256), 0, 0, a, b, n);
                }
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double gb = (k == 2 ? 2.0 : 1.0) * bytes * 10 / 1e9;
            printf("grid %6d  %-5s %7.1f us per launch  %6.0f GB/s\n", grid, k == 0 ? "fill" : k == 1 ? "read" : "copy", ms * 100, gb / (ms * 1e-3));
        }
    }
    return 0;
}
