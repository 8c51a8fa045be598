// bn_fold: running_mean / running_var update of the BatchNorm2d layers (model.py:114,123,140) for a
// whole batch at once, equal to N successive per-scene momentum updates
//     r <- (1-m) r + m s_n ,  n = 0..N-1
// (the reference forwards one scene at a time, train.py:173-177).  One workgroup per statistic:
// 256 threads fold contiguous chunks of scenes; the chunk results are composed in order by a tree.
#include "model_common.hpp"
#include "tail_parts.hpp"

namespace stg {

__global__ __launch_bounds__(256) void bn_fold_kernel(const float *__restrict__ stats, const int32_t *__restrict__ num_peds,
                                                      int N, int stat_floats, float momentum, float *__restrict__ buffers,
                                                      NbtPtrs nbt) {
    __shared__ float acc_s[256], dec_s[256];
    __shared__ int cnt_s[256];
    bn_fold_body<256>(blockIdx.x, stats, num_peds, N, stat_floats, momentum, buffers, nbt, acc_s, dec_s, cnt_s);
}

// The tail of a training step in ONE launch (different blocks do different jobs; they do not depend on each other):
//   blocks 0 .. n_buffers-1 : BatchNorm running-statistics fold of the forward just done (stg_bn_fold)
//   block  n_buffers        : reported loss  total = sum_n weights[n] * losses[n]   (train.py:58-67,76)
//   block  n_buffers + 1    : clip_grad_norm_ + SGD update of the flat parameters  (train.py:71-74,197)
__global__ __launch_bounds__(1024) void train_tail_kernel(const float *__restrict__ stats, const int32_t *__restrict__ num_peds,
                                                         int N, int stat_floats, float momentum, float *__restrict__ buffers,
                                                         NbtPtrs nbt, int n_buffers, const float *__restrict__ losses,
                                                         const float *__restrict__ weights, float *__restrict__ total,
                                                         float *__restrict__ params, float *__restrict__ grads,
                                                         int64_t count, const float *__restrict__ lr_dev, float lr,
                                                         float max_norm, float *__restrict__ grad_norm) {
    __shared__ float acc_s[1024], dec_s[1024];
    __shared__ int cnt_s[1024];
    const int b = blockIdx.x;
    if (b < n_buffers) {
        if (stats) bn_fold_body<1024>(b, stats, num_peds, N, stat_floats, momentum, buffers, nbt, acc_s, dec_s, cnt_s);
    } else if (b == n_buffers) {
        if (losses && total) weighted_sum_body(losses, weights, N, total, acc_s);
    } else {
        optim_step_body(params, grads, count, lr_dev, lr, max_norm, grad_norm, acc_s);
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

extern "C" int stg_train_tail(const stg_model_desc *d, const float *stats, const int32_t *num_peds, int N, float *buffers,
                              int64_t *const *nbt, int n_bn, const float *losses, const float *weights, float *total,
                              float *params, float *grads, int64_t count, const float *lr_dev, float lr, float max_norm,
                              float *grad_norm, void *stream) {
    using namespace stg;
    ModelLayout l;
    const int rc = make_layout(d, &l);
    if (rc != STG_OK) return rc;
    STG_REQUIRE(N >= 0 && count >= 0, STG_EINVAL, "stg_train_tail: N=%d count=%lld", N, (long long)count);
    STG_REQUIRE(params && grads, STG_EINVAL, "stg_train_tail: null parameter / gradient buffer");
    STG_REQUIRE(!stats || buffers, STG_EINVAL, "stg_train_tail: statistics without a buffer to fold them into");
    STG_REQUIRE(n_bn >= 0 && n_bn <= 3 * STG_MAX_BLOCKS && n_bn <= l.n_buffers, STG_EINVAL, "stg_train_tail: n_bn=%d", n_bn);
    NbtPtrs p{};
    p.n = nbt ? n_bn : 0;
    for (int k = 0; k < p.n; ++k) p.p[k] = nbt[k];
    const int nb = (stats && N > 0) ? l.n_buffers : 0;
    hipLaunchKernelGGL(train_tail_kernel, dim3(nb + 2), dim3(1024), 0, as_stream(stream), stats, num_peds, N, l.stat_floats,
                       d->bn_momentum, buffers, p, nb, losses, weights, total, params, grads, count, lr_dev, lr, max_norm,
                       grad_norm);
    STG_LAUNCH_CHECK("stg_train_tail");
    return STG_OK;
}
