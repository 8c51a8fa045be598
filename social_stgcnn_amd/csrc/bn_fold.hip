// bn_fold: running_mean / running_var update of the BatchNorm2d layers (model.py:114,123,140) for a
// whole batch at once, equal to N successive per-scene momentum updates
//     r <- (1-m) r + m s_n ,  n = 0..N-1
// (the reference forwards one scene at a time, train.py:173-177).  One workgroup per statistic:
// 256 threads fold contiguous chunks of scenes, thread 0 chains the chunk results in order.
#include "model_common.hpp"

namespace stg {

struct NbtPtrs {
    int64_t *p[3 * STG_MAX_BLOCKS];
    int n;
};

__global__ __launch_bounds__(256) void bn_fold_kernel(const float *__restrict__ stats, const int32_t *__restrict__ num_peds,
                                                      int N, int stat_floats, float momentum, float *__restrict__ buffers,
                                                      NbtPtrs nbt) {
    __shared__ float acc_s[256], dec_s[256];
    __shared__ int cnt_s[256];
    const int i = blockIdx.x, tid = threadIdx.x;
    const int chunk = (N + 255) / 256;
    const int lo = tid * chunk, hi = (lo + chunk) < N ? (lo + chunk) : N;
    const float keep = 1.0f - momentum;
    float acc = 0.f, dec = 1.f;
    int cnt = 0;
    for (int n = lo; n < hi; ++n) {
        if (num_peds && num_peds[n] <= 0) continue;
        acc = fmaf(acc, keep, momentum * stats[(int64_t)n * stat_floats + i]);
        dec *= keep;
        ++cnt;
    }
    acc_s[tid] = acc;
    dec_s[tid] = dec;
    cnt_s[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        float r = buffers[i];
        int total = 0;
        for (int j = 0; j < 256; ++j) {
            r = fmaf(r, dec_s[j], acc_s[j]);
            total += cnt_s[j];
        }
        buffers[i] = r;
        // one counter per BatchNorm: bumped by the block that owns the layer's first statistic
        if (i % (2 * Cfg::C) == 0) {
            const int k = i / (2 * Cfg::C);
            if (k < nbt.n && nbt.p[k]) *nbt.p[k] += total;
        }
    }
}

}  // namespace stg

extern "C" int stg_bn_fold(const stg_model_desc *d, const float *stats, const int32_t *num_peds, int N, float *buffers,
                           int64_t *const *nbt, int n_bn, void *stream) {
    using namespace stg;
    ModelLayout l;
    const int rc = make_layout(d, &l);
    if (rc != STG_OK) return rc;
    STG_REQUIRE(N >= 0, STG_EINVAL, "stg_bn_fold: N=%d", N);
    if (N == 0) return STG_OK;
    STG_REQUIRE(stats && buffers, STG_EINVAL, "stg_bn_fold: null pointer");
    STG_REQUIRE(n_bn >= 0 && n_bn <= 3 * STG_MAX_BLOCKS, STG_EINVAL, "stg_bn_fold: n_bn=%d", n_bn);
    STG_REQUIRE(n_bn <= l.n_buffers, STG_EINVAL, "stg_bn_fold: n_bn=%d > statistics %d", n_bn, l.n_buffers);
    if (N == 0 || l.n_buffers == 0) return STG_OK;
    NbtPtrs p{};
    p.n = nbt ? n_bn : 0;
    for (int k = 0; k < p.n; ++k) p.p[k] = nbt[k];
    hipLaunchKernelGGL(bn_fold_kernel, dim3(l.n_buffers), dim3(256), 0, as_stream(stream), stats, num_peds, N,
                       l.stat_floats, d->bn_momentum, buffers, p);
    STG_LAUNCH_CHECK("stg_bn_fold");
    return STG_OK;
}
