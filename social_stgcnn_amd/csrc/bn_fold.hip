// bn_fold: running_mean / running_var update of the BatchNorm2d layers (model.py:114,123,140) for a
// whole batch at once, equal to N successive per-scene momentum updates
//     r <- (1-m) r + m s_n ,  n = 0..N-1
// (the reference forwards one scene at a time, train.py:173-177).  One workgroup per statistic:
// 256 threads fold contiguous chunks of scenes; the chunk results are composed in order by a tree.
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
    for (int n0 = lo; n0 < hi; n0 += 8) {            // eight loads in flight, folded in scene order
        float sv[8];
        bool live[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = n0 + u;
            live[u] = n < hi && !(num_peds && num_peds[n] <= 0);
            sv[u] = live[u] ? stats[(int64_t)n * stat_floats + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (live[u]) {
                acc = fmaf(acc, keep, momentum * sv[u]);
                dec *= keep;
                ++cnt;
            }
    }
    acc_s[tid] = acc;
    dec_s[tid] = dec;
    cnt_s[tid] = cnt;
    __syncthreads();
    // ordered tree over the 256 chunk results: (dec, acc) pairs compose associatively,
    // (d1, a1) then (d2, a2) = (d1 d2, a1 d2 + a2) -- eight steps instead of a 256-long serial chain
    for (int off = 1; off < 256; off <<= 1) {
        if ((tid & (2 * off - 1)) == 0) {
            const float d2 = dec_s[tid + off], a2 = acc_s[tid + off];
            acc_s[tid] = fmaf(acc_s[tid], d2, a2);
            dec_s[tid] *= d2;
            cnt_s[tid] += cnt_s[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float r = fmaf(buffers[i], dec_s[0], acc_s[0]);
        const int total = cnt_s[0];
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
