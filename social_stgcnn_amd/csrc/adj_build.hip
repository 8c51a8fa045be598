// adj_build: utils.seq_to_graph (utils.py:29-53) + anorm (utils.py:23-27) + the networkx
// normalized_laplacian_matrix call (utils.py:48-50) as one HBM-write-bound kernel.
//
// One workgroup per scene.  The (V,2,T) displacement vectors sit in LDS; pass 1 gives every row its
// degree d_h = 1 + sum_{k!=h} a_hk (fp64 accumulation of fp32 weights, one lane per (t,h) row),
// pass 2 streams the T V x V tiles to HBM with coalesced 16-byte stores:
//   L_hh = (d_h - 1)/d_h,   L_hk = -a_hk / sqrt(d_h d_k),   a_hk = 1/||p_h - p_k||  (0 if equal).
// Diagonal and off-diagonal use ONE formula, w * (dinv_h * dinv_k) with w = d_h-1 resp. -a_hk, so that the
// exact cancellations of the reference's fp64 result (two-pedestrian scenes: L = c [[1,-1],[-1,1]], which
// makes the st_gcn BatchNorm mean exactly 0 and puts PReLU inputs exactly on the kink) survive in fp32.
// Algorithmic bytes per scene-window: 64*V read + 32*V*V (+64*V nodes) written.
#include "common.hpp"

namespace stg {

__device__ __forceinline__ float inv_dist(float ax, float ay, float bx, float by) {
    // the reference subtracts, squares and adds in fp32 (0-dim tensors), then 1/sqrt; the exact
    // "== 0 -> 0" rule of utils.anorm is kept, the reciprocal square root is v_rsq_f32 (1 ulp)
    const float dx = ax - bx, dy = ay - by;
    const float s = dx * dx + dy * dy;
    return s == 0.f ? 0.f : rsqrtf(s);
}

// One workgroup per scene: the (V,2,T) displacements are read once (coalesced) into LDS; thread
// (t,h) sums the degree of row h at time t in fp64; the T V x V tiles leave as 16-byte stores.
template <bool VEC4>
__global__ __launch_bounds__(256) void adj_build_kernel(
    const float *__restrict__ rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
    const int32_t *__restrict__ num_peds, int V, int T, int normalize,
    float *__restrict__ nodes, float *__restrict__ adj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *px = sm;                // [T][V]
    float *py = px + T * V;        // [T][V]
    float *dinv = py + T * V;      // [T][V]  1/sqrt(d)
    float *diag = dinv + T * V;    // [T][V]  d-1 (sum of the off-diagonal weights)
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *r = rel + n * rel_sn;
    const bool dense = rel_st == 1 && rel_sc == T && rel_sv == 2 * T;      // reference layout (V,2,T)
    for (int e = tid; e < V * 2 * T; e += blockDim.x) {
        // e enumerates (h, c, t) in the reference's memory order so the dense case is one coalesced sweep
        const int t = e % T, hc = e / T, c = hc & 1, h = hc >> 1;
        float v = 0.f;
        if (h < vi) v = dense ? r[e] : r[h * rel_sv + c * rel_sc + t * rel_st];
        (c ? py : px)[t * V + h] = v;
    }
    __syncthreads();
    if (nodes) {
        float2 *o = reinterpret_cast<float2 *>(nodes + (int64_t)n * T * V * 2);
        for (int e = tid; e < T * V; e += blockDim.x) o[e] = make_float2(px[e], py[e]);
    }
    if (normalize) {
        for (int e = tid; e < T * V; e += blockDim.x) {
            const int t = e / V, h = e - t * V;
            if (h < vi) {
                const float *qx = px + t * V, *qy = py + t * V;
                const float hx = qx[h], hy = qy[h];
                double acc = 1.0;
                for (int k = 0; k < vi; ++k)
                    if (k != h) acc += (double)inv_dist(hx, hy, qx[k], qy[k]);
                dinv[e] = (float)(1.0 / sqrt(acc));
                diag[e] = (float)(acc - 1.0);
            }
        }
        __syncthreads();
    }
    float *out = adj + (int64_t)n * T * V * V;
    if (VEC4) {
        const int v4 = V >> 2;
        for (int e = tid; e < T * V * v4; e += blockDim.x) {
            const int th = e / v4, k0 = (e - th * v4) << 2;
            const int t = th / V, h = th - t * V;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < vi) {
                const float *qx = px + t * V, *qy = py + t * V, *qd = dinv + t * V;
                const float hx = qx[h], hy = qy[h];
                const float dh = normalize ? qd[h] : 1.f;
                float vals[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    float v = 0.f;
                    if (k < vi) {
                        if (k == h)
                            v = normalize ? diag[th] * (dh * dh) : 1.f;     // same formula as off-diagonal:
                        else {                                              // V=2 rows cancel exactly like the reference's
                            const float a = inv_dist(hx, hy, qx[k], qy[k]);
                            v = normalize ? -(a * (dh * qd[k])) : a;   // dh*dk commutes: L is bitwise symmetric
                        }
                    }
                    vals[j] = v;
                }
                o = make_float4(vals[0], vals[1], vals[2], vals[3]);
            }
            reinterpret_cast<float4 *>(out)[e] = o;
        }
    } else {
        for (int e = tid; e < T * V * V; e += blockDim.x) {
            const int th = e / V, k = e - th * V;
            const int t = th / V, h = th - t * V;
            float v = 0.f;
            if (h < vi && k < vi) {
                if (k == h)
                    v = normalize ? diag[th] * (dinv[th] * dinv[th]) : 1.f;
                else {
                    const float a = inv_dist(px[th], py[th], px[t * V + k], py[t * V + k]);
                    v = normalize ? -(a * (dinv[th] * dinv[t * V + k])) : a;
                }
            }
            out[e] = v;
        }
    }
}

}  // namespace stg

extern "C" int stg_adj_build(const float *rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc,
                             int64_t rel_st, const int32_t *num_peds, int N, int V, int T,
                             int normalize, float *nodes, float *adj, void *stream) {
    STG_REQUIRE(N >= 0 && V > 0 && T > 0, STG_EINVAL, "stg_adj_build: bad sizes N=%d V=%d T=%d", N, V, T);
    STG_REQUIRE((int64_t)N * T < (1ll << 31), STG_EINVAL, "stg_adj_build: N*T too large");
    if (N == 0) return STG_OK;
    STG_REQUIRE(rel && adj, STG_EINVAL, "stg_adj_build: null rel/adj pointer");
    const size_t lds = (size_t)4 * T * V * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_adj_build: V=%d exceeds the LDS budget", V);
    const dim3 grid((unsigned)N), block(256);
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(adj) & 15) == 0);
    if (vec4)
        hipLaunchKernelGGL(stg::adj_build_kernel<true>, grid, block, lds, stg::as_stream(stream), rel, rel_sn,
                           rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
    else
        hipLaunchKernelGGL(stg::adj_build_kernel<false>, grid, block, lds, stg::as_stream(stream), rel, rel_sn,
                           rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
    STG_LAUNCH_CHECK("stg_adj_build");
    return STG_OK;
}
