// Error reporting + small utility kernels (SGD step, MFMA self-test).
#include "common.hpp"
#include "tail_parts.hpp"
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

// clip_grad_norm_ + SGD over the flat buffers in ONE single-workgroup launch (body: tail_parts.hpp)
__global__ __launch_bounds__(1024) void optim_step_kernel(float *__restrict__ p, float *__restrict__ g, int64_t n,
                                                          const float *__restrict__ lr_dev, float lr_host,
                                                          float max_norm, float *__restrict__ norm_out) {
    __shared__ float red[16];
    optim_step_body(p, g, n, lr_dev, lr_host, max_norm, norm_out, red);
}

// ---- data-parallel step: ONE collective carries the gradient and the BatchNorm fold ---------------
// pack = [ gradient (n_p) | per rank r: acc_r (n_b), n_r ].  acc_r = after - keep^{n_r} before is what rank r's n_r
// non-empty scenes added to its copy of the running statistics (after = keep^{n_r} before + acc_r); only the own slot
// is non-zero, so the all-reduce(sum) of the pack is at the same time the gradient sum and an all-gather of the slots.
__global__ __launch_bounds__(1024) void dp_pack_kernel(const float *__restrict__ grad, const float *__restrict__ before,
                                                       const float *__restrict__ after,
                                                       const int32_t *__restrict__ num_peds, int N, float momentum,
                                                       int rank, int world, int n_p, int n_b, float *__restrict__ pack) {
    __shared__ int cnt[16];
    const int tid = threadIdx.x;
    int c = 0;
    for (int i = tid; i < N; i += blockDim.x) c += (!num_peds || num_peds[i] > 0) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((tid & 63) == 0) cnt[tid >> 6] = c;
    __syncthreads();
    int n_r = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) n_r += cnt[w];
    const float own = (float)pow(1.0 - (double)momentum, (double)n_r);
    for (int i = tid; i < n_p; i += blockDim.x) pack[i] = grad[i];
    const int slot = n_b + 1;
    for (int i = tid; i < world * slot; i += blockDim.x) {
        const int r = i / slot, k = i - r * slot;
        float v = 0.f;
        if (r == rank) v = k < n_b ? after[k] - own * before[k] : (float)n_r;
        pack[n_p + i] = v;
    }
}

// after the all-reduce: running statistics of ONE process that saw rank 0's scenes, then rank 1's, ...:
//   before * keep^{sum n} + sum_r acc_r * keep^{sum_{j>r} n_j}
// nbt (optional): the num_batches_tracked counters, which the forward's own fold has advanced by THIS rank's scenes, take
// the other ranks' scene counts from the same pack -- every rank ends with the count of the one process
__global__ __launch_bounds__(256) void dp_fold_kernel(const float *__restrict__ pack, const float *__restrict__ before,
                                                      float momentum, int rank, int world, int n_p, int n_b,
                                                      float *__restrict__ buffers, NbtPtrs nbt) {
    const int slot = n_b + 1;
    const double keep = 1.0 - (double)momentum;
    if ((int)threadIdx.x < nbt.n) {
        int64_t others = 0;
        for (int r = 0; r < world; ++r)
            if (r != rank) others += (int64_t)(pack[n_p + (int64_t)r * slot + n_b] + 0.5f);
        *nbt.p[threadIdx.x] += others;
    }
    for (int k = threadIdx.x; k < n_b; k += blockDim.x) {
        double later = 0.0, acc = 0.0;               // later = scenes of the ranks behind r
        for (int r = world - 1; r >= 0; --r) {
            const float *s = pack + n_p + (int64_t)r * slot;
            acc += (double)s[k] * pow(keep, later);
            later += (double)s[n_b];
        }
        buffers[k] = (float)((double)before[k] * pow(keep, later) + acc);
    }
}

__global__ __launch_bounds__(1024) void weighted_sum_kernel(const float *__restrict__ v, const float *__restrict__ w,
                                                            int N, float *__restrict__ out) {
    __shared__ float red[16];
    weighted_sum_body(v, w, N, out, red);
}

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

int stg_dp_pack(const float *grads, const float *bn_before, const float *bn_after, const int32_t *num_peds, int N,
                float momentum, int rank, int world, int n_params, int n_buffers, float *pack, void *stream) {
    STG_REQUIRE(grads && bn_before && bn_after && pack, STG_EINVAL, "stg_dp_pack: null pointer");
    STG_REQUIRE(N >= 0 && world >= 1 && rank >= 0 && rank < world && n_params >= 0 && n_buffers >= 0, STG_EINVAL,
                "stg_dp_pack: bad sizes N=%d rank=%d world=%d", N, rank, world);
    hipLaunchKernelGGL(stg::dp_pack_kernel, dim3(1), dim3(1024), 0, stg::as_stream(stream), grads, bn_before, bn_after,
                       num_peds, N, momentum, rank, world, n_params, n_buffers, pack);
    STG_LAUNCH_CHECK("stg_dp_pack");
    return STG_OK;
}

int stg_dp_fold(const float *pack, const float *bn_before, float momentum, int rank, int world, int n_params,
                int n_buffers, float *buffers, int64_t *const *nbt, int n_bn, void *stream) {
    STG_REQUIRE(pack && bn_before && buffers, STG_EINVAL, "stg_dp_fold: null pointer");
    STG_REQUIRE(world >= 1 && rank >= 0 && rank < world && n_params >= 0 && n_buffers >= 0, STG_EINVAL,
                "stg_dp_fold: bad sizes rank=%d world=%d", rank, world);
    STG_REQUIRE(n_bn >= 0 && n_bn <= 3 * STG_MAX_BLOCKS && (n_bn == 0 || nbt), STG_EINVAL, "stg_dp_fold: bad nbt / n_bn=%d", n_bn);
    stg::NbtPtrs np{};
    np.n = nbt ? n_bn : 0;
    for (int k = 0; k < np.n; ++k) np.p[k] = nbt[k];
    hipLaunchKernelGGL(stg::dp_fold_kernel, dim3(1), dim3(256), 0, stg::as_stream(stream), pack, bn_before, momentum,
                       rank, world, n_params, n_buffers, buffers, np);
    STG_LAUNCH_CHECK("stg_dp_fold");
    return STG_OK;
}

int stg_weighted_sum(const float *values, const float *weights, int N, float *out, void *stream) {
    STG_REQUIRE(values && out && N >= 0, STG_EINVAL, "stg_weighted_sum: null pointer or negative N");
    hipLaunchKernelGGL(stg::weighted_sum_kernel, dim3(1), dim3(1024), 0, stg::as_stream(stream), values, weights, N,
                       out);
    STG_LAUNCH_CHECK("stg_weighted_sum");
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
