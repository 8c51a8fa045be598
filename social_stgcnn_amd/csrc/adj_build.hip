// adj_build: utils.seq_to_graph (utils.py:29-53) + anorm (utils.py:23-27) + the networkx
// normalized_laplacian_matrix call (utils.py:48-50) as one HBM-write-bound kernel.
//
// One workgroup per (scene n, timestep t).  The V displacement vectors sit in LDS; pass 1 gives
// every row its degree d_h = 1 + sum_{k!=h} a_hk (fp64 accumulation of fp32 weights, one wave per
// row, wave-shuffle reduction), pass 2 streams the V x V tile to HBM with coalesced 16-byte stores:
//   L_hh = (d_h - 1)/d_h,   L_hk = -a_hk / sqrt(d_h d_k),   a_hk = 1/||p_h - p_k||  (0 if equal).
// Algorithmic bytes per scene-window: 64*V read + 32*V*V (+64*V nodes) written.
#include "common.hpp"

namespace stg {

__device__ __forceinline__ float inv_dist(float ax, float ay, float bx, float by) {
    // the reference subtracts, squares and adds in fp32 (0-dim tensors), then sqrt / divide
    const float dx = ax - bx, dy = ay - by;
    const float s = dx * dx + dy * dy;
    return s == 0.f ? 0.f : 1.0f / sqrtf(s);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void adj_build_kernel(
    const float *__restrict__ rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
    const int32_t *__restrict__ num_peds, int V, int T, int normalize,
    float *__restrict__ nodes, float *__restrict__ adj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *px = sm;            // [V]
    float *py = sm + V;        // [V]
    float *dinv = sm + 2 * V;  // [V]  1/sqrt(d)
    float *diag = sm + 3 * V;  // [V]  (d-1)/d
    const int n = blockIdx.x / T, t = blockIdx.x % T;
    const int tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);

    const float *r = rel + n * rel_sn + t * rel_st;
    for (int h = tid; h < V; h += blockDim.x) {
        float x = 0.f, y = 0.f;
        if (h < vi) {
            x = r[h * rel_sv];
            y = r[h * rel_sv + rel_sc];
        }
        px[h] = x;
        py[h] = y;
        if (nodes) {
            float2 *o = reinterpret_cast<float2 *>(nodes + ((int64_t)(n * T + t) * V + h) * 2);
            *o = make_float2(x, y);
        }
    }
    __syncthreads();

    if (normalize) {
        const int wave = tid >> 6, lane = tid & 63, nwaves = blockDim.x >> 6;
        for (int h = wave; h < vi; h += nwaves) {
            const float hx = px[h], hy = py[h];
            double acc = 0.0;
            for (int k = lane; k < vi; k += 64)
                if (k != h) acc += (double)inv_dist(hx, hy, px[k], py[k]);
            acc = wave_sum(acc);
            if (lane == 0) {
                const double d = 1.0 + acc;
                dinv[h] = (float)(1.0 / sqrt(d));
                diag[h] = (float)((d - 1.0) / d);
            }
        }
        __syncthreads();
    }

    float *out = adj + (int64_t)(n * T + t) * V * V;
    if (VEC4) {
        const int v4 = V >> 2;
        for (int e = tid; e < V * v4; e += blockDim.x) {
            const int h = e / v4, k0 = (e - h * v4) << 2;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < vi) {
                const float hx = px[h], hy = py[h];
                const float dh = normalize ? dinv[h] : 1.f;
                float vals[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    float v = 0.f;
                    if (k < vi) {
                        if (k == h)
                            v = normalize ? diag[h] : 1.f;
                        else {
                            const float a = inv_dist(hx, hy, px[k], py[k]);
                            v = normalize ? -(a * (dh * dinv[k])) : a;   // dh*dk commutes: L is bitwise symmetric
                        }
                    }
                    vals[j] = v;
                }
                o = make_float4(vals[0], vals[1], vals[2], vals[3]);
            }
            reinterpret_cast<float4 *>(out)[e] = o;
        }
    } else {
        for (int e = tid; e < V * V; e += blockDim.x) {
            const int h = e / V, k = e - h * V;
            float v = 0.f;
            if (h < vi && k < vi) {
                if (k == h)
                    v = normalize ? diag[h] : 1.f;
                else {
                    const float a = inv_dist(px[h], py[h], px[k], py[k]);
                    v = normalize ? -(a * (dinv[h] * dinv[k])) : a;
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
    STG_REQUIRE(rel && adj, STG_EINVAL, "stg_adj_build: null rel/adj pointer");
    STG_REQUIRE(N >= 0 && V > 0 && T > 0, STG_EINVAL, "stg_adj_build: bad sizes N=%d V=%d T=%d", N, V, T);
    STG_REQUIRE((int64_t)N * T < (1ll << 31), STG_EINVAL, "stg_adj_build: N*T too large");
    if (N == 0) return STG_OK;
    const size_t lds = (size_t)4 * V * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_adj_build: V=%d exceeds the LDS budget", V);
    const dim3 grid((unsigned)(N * T)), block(256);
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
