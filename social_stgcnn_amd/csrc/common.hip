// Error reporting + small utility kernels (SGD step, MFMA self-test).
#include "common.hpp"
#include <cstdlib>

namespace stg {

static thread_local char g_err[512] = "";

char *last_error_buf() { return g_err; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#ifdef STG_DIAG
int diag_env(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return static_cast<int>(e);
}

// p -= lr * g  (train.py:197 SGD without momentum / weight decay)
__global__ void sgd_kernel(float *__restrict__ p, const float *__restrict__ g, int64_t n, float lr) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] - lr * g[i];
}

// clip_grad_norm_ (train.py:71-73) + SGD (train.py:197) over the flat buffers in ONE single-workgroup launch:
// total = ||g||_2, coef = min(1, max_norm / (total + 1e-6)), g *= coef (in place, as torch does), p -= lr g.
// lr comes from device memory when lr_dev != NULL, so a captured hipGraph follows the StepLR schedule.
__global__ __launch_bounds__(1024) void optim_step_kernel(float *__restrict__ p, float *__restrict__ g, int64_t n,
                                                          const float *__restrict__ lr_dev, float lr_host,
                                                          float max_norm, float *__restrict__ norm_out) {
    __shared__ float red[16];
    const int tid = threadIdx.x;
    const float lr = lr_dev ? lr_dev[0] : lr_host;
    float coef = 1.f;
    if (max_norm > 0.f || norm_out) {
        float acc = 0.f;
        for (int64_t i = tid; i < n; i += blockDim.x) acc = fmaf(g[i], g[i], acc);
        acc = wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
        const float nrm = sqrtf(tot);
        if (norm_out && tid == 0) norm_out[0] = nrm;
        if (max_norm > 0.f) {
            coef = max_norm / (nrm + 1e-6f);
            coef = coef > 1.f ? 1.f : coef;
        }
    }
    for (int64_t i = tid; i < n; i += blockDim.x) {
        float gi = g[i];
        if (max_norm > 0.f) {
            gi *= coef;
            g[i] = gi;
        }
        p[i] = p[i] - lr * gi;
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave: C(16x16) = A(16xK) B(Kx16).  A operand: lane l holds A[l&15][4s + (l>>4)];
// B operand: lane l holds B[4s + (l>>4)][l&15]; D: lane l, reg r -> C[(l>>4)*4 + r][l&15].
__global__ void mfma_probe_kernel(const float *__restrict__ a, const float *__restrict__ b, int K,
                                  float *__restrict__ c) {
    const int l = threadIdx.x;
    const int i = l & 15, kq = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < K / 4; ++s) {
        float av = a[i * K + 4 * s + kq];
        float bv = b[(4 * s + kq) * 16 + i];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) c[(kq * 4 + r) * 16 + i] = acc[r];
}

}  // namespace stg

extern "C" {

int stg_abi_version(void) { return STG_ABI_VERSION; }

const char *stg_last_error(void) { return stg::last_error_buf(); }

int stg_sgd_step(float *params, const float *grads, int64_t count, float lr, void *stream) {
    STG_REQUIRE(params && grads && count >= 0, STG_EINVAL, "stg_sgd_step: null pointer or negative count");
    if (count == 0) return STG_OK;
    const int threads = 256;
    const int64_t blocks = (count + threads - 1) / threads;
    hipLaunchKernelGGL(stg::sgd_kernel, dim3((unsigned)blocks), dim3(threads), 0, stg::as_stream(stream),
                       params, grads, count, lr);
    STG_LAUNCH_CHECK("stg_sgd_step");
    return STG_OK;
}

int stg_optim_step(float *params, float *grads, int64_t count, const float *lr_dev, float lr, float max_norm,
                   float *grad_norm, void *stream) {
    STG_REQUIRE(params && grads && count >= 0, STG_EINVAL, "stg_optim_step: null pointer or negative count");
    STG_REQUIRE(count <= (1ll << 22), STG_EUNSUPPORTED,
                "stg_optim_step: %lld parameters exceed the single-workgroup step (use stg_sgd_step)", (long long)count);
    if (count == 0) return STG_OK;
    hipLaunchKernelGGL(stg::optim_step_kernel, dim3(1), dim3(1024), 0, stg::as_stream(stream), params, grads, count,
                       lr_dev, lr, max_norm, grad_norm);
    STG_LAUNCH_CHECK("stg_optim_step");
    return STG_OK;
}

int stg_selftest_mfma(const float *a, const float *b, int K, float *c, void *stream) {
    STG_REQUIRE(a && b && c, STG_EINVAL, "stg_selftest_mfma: null pointer");
    STG_REQUIRE(K > 0 && K % 4 == 0, STG_EINVAL, "stg_selftest_mfma: K=%d must be a positive multiple of 4", K);
    hipLaunchKernelGGL(stg::mfma_probe_kernel, dim3(1), dim3(64), 0, stg::as_stream(stream), a, b, K, c);
    STG_LAUNCH_CHECK("stg_selftest_mfma");
    return STG_OK;
}

}  // extern "C"
