// bestofk: the evaluation tail of test.test (test.py:59-123) with metrics.ade / fde / nodes_rel_to_nodes_abs
// (metrics.py:21-75) as ONE device op: per pedestrian, draw K trajectories from the predicted bivariate
// Gaussians (MultivariateNormal(mean, cov).sample() = mean + chol(cov) eps, test.py:59-71), integrate the
// displacements from the last observed position (test.py:89-91), and keep the smallest average / final
// displacement error against the ground truth (test.py:104-117).
//
// One lane per (scene, pedestrian); lanes run along v, so every load of V_pred, the targets and the noise is
// coalesced.  The whole K x P loop stays in registers: HBM traffic is V_pred + targets once (L2 serves the K
// re-reads) + the optional noise tensor.  Noise: either caller-provided standard normals (exact parity with a
// CPU sampler fed the same numbers) or an in-kernel Philox4x32-10 stream keyed by (seed; scene, ped, k, t).
#include "common.hpp"

namespace stg {

namespace {

__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0,
                                             uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
}

// two independent standard normals for counter (lane, draw) under `seed` (Philox4x32-10 + Box-Muller)
__device__ __forceinline__ float2 philox_normal2(uint64_t seed, uint64_t lane, uint32_t draw) {
    uint32_t c0 = (uint32_t)lane, c1 = (uint32_t)(lane >> 32), c2 = draw, c3 = 0x5354474Eu;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);       // (0,1)
    const float u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u0));
    float s, c;
    sincosf(6.28318530717958647692f * u1, &s, &c);
    return make_float2(r * c, r * s);
}

}  // namespace

__global__ __launch_bounds__(256) void bestofk_kernel(
    const float *__restrict__ pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv,
    const float *__restrict__ target_rel, const float *__restrict__ obs_last,
    const int32_t *__restrict__ num_peds, const float *__restrict__ noise, uint64_t seed, int N, int P, int V,
    int K, float *__restrict__ ade, float *__restrict__ fde) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * V) return;
    const int n = (int)(idx / V), v = (int)(idx - (int64_t)n * V);
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    if (v >= vi) {                         // padded slot
        ade[idx] = 0.f;
        fde[idx] = 0.f;
        return;
    }
    const float *pn = pred + n * p_sn + v * p_sv;
    const float *tn = target_rel + ((int64_t)n * P * V + v) * 2;
    float ox = 0.f, oy = 0.f;
    if (obs_last) {
        ox = obs_last[idx * 2];
        oy = obs_last[idx * 2 + 1];
    }
    float best_a = INFINITY, best_f = INFINITY;
    const float inv_p = 1.0f / (float)P;
    for (int k = 0; k < K; ++k) {
        float cx = 0.f, cy = 0.f, gx = 0.f, gy = 0.f, acc = 0.f, last = 0.f;
        for (int t = 0; t < P; ++t) {
            const float *q = pn + t * p_sp;
            const float mx = q[0], my = q[p_sf];
            const float sx = expf(q[2 * p_sf]), sy = expf(q[3 * p_sf]), rho = tanhf(q[4 * p_sf]);
            // chol([[sx^2, rho sx sy], [rho sx sy, sy^2]]) in the order torch.linalg.cholesky evaluates it
            const float c01 = rho * sx * sy;
            const float l00 = sqrtf(sx * sx);
            const float l10 = c01 / l00;
            const float l11 = sqrtf(sy * sy - l10 * l10);
            float2 e;
            if (noise)
                e = *reinterpret_cast<const float2 *>(noise + ((((int64_t)k * N + n) * P + t) * V + v) * 2);
            else
                e = philox_normal2(seed, (uint64_t)idx, (uint32_t)(k * P + t));
            cx += mx + l00 * e.x;                       // cumulative sum of sampled displacements (metrics.py:70-73)
            cy += my + (l10 * e.x + l11 * e.y);
            const float2 tg = *reinterpret_cast<const float2 *>(tn + (int64_t)t * V * 2);
            gx += tg.x;
            gy += tg.y;
            const float dx = (cx + ox) - (gx + ox), dy = (cy + oy) - (gy + oy);
            last = sqrtf(dx * dx + dy * dy);
            acc += last;
        }
        best_a = fminf(best_a, acc * inv_p);
        best_f = fminf(best_f, last);
    }
    ade[idx] = K > 0 ? best_a : 0.f;
    fde[idx] = K > 0 ? best_f : 0.f;
}

}  // namespace stg

extern "C" {

int stg_bestofk_eval(const float *pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv,
                     const float *target_rel, const float *obs_last, const int32_t *num_peds, const float *noise,
                     uint64_t seed, int N, int P, int V, int K, float *ade, float *fde, void *stream) {
    STG_REQUIRE(N >= 0 && P > 0 && V > 0 && K >= 0, STG_EINVAL, "stg_bestofk_eval: bad sizes N=%d P=%d V=%d K=%d", N,
                P, V, K);
    if (N == 0) return STG_OK;
    STG_REQUIRE(pred && target_rel && ade && fde, STG_EINVAL, "stg_bestofk_eval: null pointer");
    const int64_t total = (int64_t)N * V;
    STG_REQUIRE(total < (1ll << 31) * 256, STG_EINVAL, "stg_bestofk_eval: N*V too large");
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(stg::bestofk_kernel, dim3((unsigned)blocks), dim3(256), 0, stg::as_stream(stream), pred, p_sn,
                       p_sf, p_sp, p_sv, target_rel, obs_last, num_peds, noise, seed, N, P, V, K, ade, fde);
    STG_LAUNCH_CHECK("stg_bestofk_eval");
    return STG_OK;
}

}  // extern "C"
