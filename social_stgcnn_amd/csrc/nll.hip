// nll: metrics.bivariate_loss (metrics.py:84-113), batched and fused with its gradient.
//   loss[n] = mean_{p<P, v<V_n} -log(max(pdf, 1e-20)),
//   pdf = exp(-z / (2(1-rho^2))) / (2 pi sx sy sqrt(1-rho^2)),  z as metrics.py:97.
// One workgroup per scene; the forward value follows the reference's operation order; the gradient
// is the closed form of d(-log pdf)/d(pred) and is zero where the clamp is active.
#include "common.hpp"
#include "nll_elem.hpp"

namespace stg {

__global__ __launch_bounds__(256) void nll_fwd_kernel(
    const float *__restrict__ pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv,
    const float *__restrict__ target, const int32_t *__restrict__ num_peds, const float *__restrict__ gscale,
    int P, int V, float *__restrict__ loss, float *__restrict__ grad, int fast) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    (void)fast;                                        // (diagnostic builds: STG_FAST_NLL=1 measures nll_elem_t<true>)
    const float *pn = pred + n * p_sn;
    const float *tn = target + (int64_t)n * P * V * 2;
    float *gn = grad ? grad + (int64_t)n * 5 * P * V : nullptr;
    const float inv_cnt = vi > 0 ? 1.0f / (float)(P * vi) : 0.f;
    const float gs = inv_cnt * (gscale ? gscale[n] : 1.f);      // d(mean)/d(elem) times the caller's per-scene weight
    float acc = 0.f;
    for (int e = tid; e < P * V; e += blockDim.x) {
        const int p = e / V, v = e - p * V;
        float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (v < vi) {
            const float *q = pn + p * p_sp + v * p_sv;
            const float2 tg = *reinterpret_cast<const float2 *>(tn + (int64_t)e * 2);
#ifdef STG_DIAG
            if (fast) acc += nll_elem_t<true>(q[0], q[p_sf], q[2 * p_sf], q[3 * p_sf], q[4 * p_sf], tg.x, tg.y, gn != nullptr, g);
            else
#endif
            acc += nll_elem(q[0], q[p_sf], q[2 * p_sf], q[3 * p_sf], q[4 * p_sf], tg.x, tg.y, gn != nullptr, g);
        }
        if (gn) {
            const int64_t pv = (int64_t)P * V;
            gn[e] = g[0] * gs;
            gn[pv + e] = g[1] * gs;
            gn[2 * pv + e] = g[2] * gs;
            gn[3 * pv + e] = g[3] * gs;
            gn[4 * pv + e] = g[4] * gs;
        }
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) loss[n] = (red[0] + red[1] + red[2] + red[3]) * inv_cnt;
}

__global__ void nll_bwd_kernel(const float *grad, const float *__restrict__ gloss, int64_t per_scene,
                               int64_t total, float *out) {      // out may alias grad
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < total) out[i] = grad[i] * gloss[i / per_scene];
}

}  // namespace stg

extern "C" {

int stg_nll_fwd(const float *pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv, const float *target,
                const int32_t *num_peds, const float *grad_scale, int N, int P, int V, float *loss, float *grad,
                void *stream) {
    STG_REQUIRE(N >= 0 && P > 0 && V > 0, STG_EINVAL, "stg_nll_fwd: bad sizes N=%d P=%d V=%d", N, P, V);
    if (N == 0) return STG_OK;
    STG_REQUIRE(pred && target && loss, STG_EINVAL, "stg_nll_fwd: null pointer");
    hipLaunchKernelGGL(stg::nll_fwd_kernel, dim3(N), dim3(256), 0, stg::as_stream(stream), pred, p_sn, p_sf, p_sp,
                       p_sv, target, num_peds, grad_scale, P, V, loss, grad, stg::diag_env("STG_FAST_NLL", 0));
    STG_LAUNCH_CHECK("stg_nll_fwd");
    return STG_OK;
}

int stg_nll_bwd(const float *grad, const float *gloss, int N, int P, int V, float *out, void *stream) {
    STG_REQUIRE(grad && gloss && out, STG_EINVAL, "stg_nll_bwd: null pointer");
    STG_REQUIRE(N >= 0 && P > 0 && V > 0, STG_EINVAL, "stg_nll_bwd: bad sizes");
    if (N == 0) return STG_OK;
    const int64_t per_scene = (int64_t)5 * P * V, total = per_scene * N;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(stg::nll_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stg::as_stream(stream), grad, gloss,
                       per_scene, total, out);
    STG_LAUNCH_CHECK("stg_nll_bwd");
    return STG_OK;
}

}  // extern "C"
