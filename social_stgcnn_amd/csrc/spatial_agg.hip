// spatial_agg: the einsum of ConvTemporalGraphical.forward (model.py:67)
//     y[n,c,t,w] = sum_v x[n,c,t,v] A[n,t,v,w]              forward
//     dx[n,c,t,v] = sum_w dy[n,c,t,w] A[n,t,v,w]            backward (no dA: A is data)
// HBM-bound (A is read once: 32*V*V bytes per scene-window against 80*V*V flops).
//
// forward : one workgroup per scene; x[n] (C*T*V floats) is staged in LDS, every lane owns a
//           (t, w..w+VEC-1) output strip and streams column strips of A[n,t] from HBM with coalesced
//           VEC*4-byte loads along w (lanes consecutive in w, then in t).
// backward: one workgroup per (scene, t); the V x V tile A[n,t] is staged through LDS in row chunks
//           (coalesced global reads, odd row stride -> conflict-free column reads), lanes own (c, v).
#include "common.hpp"

namespace stg {

constexpr int kAggMaxC = 8;  // channels per pass (register accumulators)

template <int VEC>
__global__ __launch_bounds__(256) void spatial_agg_fwd_kernel(
    const float *__restrict__ x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
    const float *__restrict__ adj, int64_t a_sn, const int32_t *__restrict__ num_peds,
    int C, int T, int V, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [C][T][V]
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *xn = x + n * x_sn;
    for (int e = tid; e < C * T * V; e += blockDim.x) {
        const int v = e % V, ct = e / V, t = ct % T, c = ct / T;
        xs[e] = v < vi ? xn[c * x_sc + t * x_st + v * x_sv] : 0.f;
    }
    __syncthreads();
    const float *an = adj + n * a_sn;
    float *yn = y + (int64_t)n * C * T * V;
    const int vq = V / VEC;
    for (int c0 = 0; c0 < C; c0 += kAggMaxC) {
        const int cn = (C - c0) < kAggMaxC ? (C - c0) : kAggMaxC;
        for (int q = tid; q < T * vq; q += blockDim.x) {
            const int t = q / vq, w0 = (q - t * vq) * VEC;
            float acc[kAggMaxC][VEC];
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c)
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[c][j] = 0.f;
            const float *at = an + (int64_t)t * V * V + w0;
            const float *xt = xs + (c0 * T + t) * V;
#pragma unroll 4
            for (int v = 0; v < vi; ++v) {
                float a[VEC];
                if (VEC == 4) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(at + (int64_t)v * V);
                    a[0] = a4.x; a[1 % VEC] = a4.y; a[2 % VEC] = a4.z; a[3 % VEC] = a4.w;
                } else {
                    a[0] = at[(int64_t)v * V];
                }
#pragma unroll
                for (int c = 0; c < kAggMaxC; ++c) {
                    if (c < cn) {
                        const float xv = xt[c * T * V + v];
#pragma unroll
                        for (int j = 0; j < VEC; ++j) acc[c][j] = fmaf(xv, a[j], acc[c][j]);
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c) {
                if (c < cn) {
                    float *o = yn + ((int64_t)(c0 + c) * T + t) * V + w0;
                    if (VEC == 4) {
                        float4 r;
                        r.x = (w0 + 0 < vi) ? acc[c][0] : 0.f;
                        r.y = (w0 + 1 < vi) ? acc[c][1 % VEC] : 0.f;
                        r.z = (w0 + 2 < vi) ? acc[c][2 % VEC] : 0.f;
                        r.w = (w0 + 3 < vi) ? acc[c][3 % VEC] : 0.f;
                        *reinterpret_cast<float4 *>(o) = r;
                    } else {
                        o[0] = (w0 < vi) ? acc[c][0] : 0.f;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void spatial_agg_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ adj, int64_t a_sn,
    const int32_t *__restrict__ num_peds, int C, int T, int V, int rows_per_chunk,
    float *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = V | 1;                 // odd row stride
    float *dys = sm;                      // [C][V]
    float *as = sm + C * V;               // [rows_per_chunk][ld]
    const int n = blockIdx.x / T, t = blockIdx.x % T, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *dyn = dy + (int64_t)n * C * T * V;
    float *dxn = dx + (int64_t)n * C * T * V;
    for (int e = tid; e < C * V; e += blockDim.x) {
        const int c = e / V, w = e - c * V;
        dys[e] = w < vi ? dyn[((int64_t)c * T + t) * V + w] : 0.f;
    }
    const float *at = adj + n * a_sn + (int64_t)t * V * V;
    for (int r0 = 0; r0 < V; r0 += rows_per_chunk) {
        const int rn = (V - r0) < rows_per_chunk ? (V - r0) : rows_per_chunk;
        __syncthreads();
        for (int e = tid; e < rn * V; e += blockDim.x) {
            const int r = e / V, w = e - r * V;
            as[r * ld + w] = at[(int64_t)(r0 + r) * V + w];
        }
        __syncthreads();
        for (int e = tid; e < C * rn; e += blockDim.x) {
            const int c = e / rn, r = e - c * rn;
            const int v = r0 + r;
            float acc = 0.f;
            if (v < vi) {
                const float *row = as + r * ld;
                const float *d = dys + c * V;
                for (int w = 0; w < vi; ++w) acc = fmaf(d[w], row[w], acc);
            }
            dxn[((int64_t)c * T + t) * V + v] = acc;
        }
    }
}

}  // namespace stg

extern "C" {

int stg_spatial_agg_fwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                        const float *adj, int64_t a_sn, const int32_t *num_peds, int N, int C, int T,
                        int V, float *y, void *stream) {
    STG_REQUIRE(x && adj && y, STG_EINVAL, "stg_spatial_agg_fwd: null pointer");
    STG_REQUIRE(N >= 0 && C > 0 && T > 0 && V > 0, STG_EINVAL, "stg_spatial_agg_fwd: bad sizes");
    if (N == 0) return STG_OK;
    const size_t lds = (size_t)C * T * V * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_spatial_agg_fwd: C*T*V=%d floats exceed LDS", C * T * V);
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(adj) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && (a_sn % 4 == 0);
    if (vec4)
        hipLaunchKernelGGL(stg::spatial_agg_fwd_kernel<4>, dim3(N), dim3(256), lds, stg::as_stream(stream), x,
                           x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, C, T, V, y);
    else
        hipLaunchKernelGGL(stg::spatial_agg_fwd_kernel<1>, dim3(N), dim3(256), lds, stg::as_stream(stream), x,
                           x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, C, T, V, y);
    STG_LAUNCH_CHECK("stg_spatial_agg_fwd");
    return STG_OK;
}

int stg_spatial_agg_bwd(const float *dy, const float *adj, int64_t a_sn, const int32_t *num_peds, int N,
                        int C, int T, int V, float *dx, void *stream) {
    STG_REQUIRE(dy && adj && dx, STG_EINVAL, "stg_spatial_agg_bwd: null pointer");
    STG_REQUIRE(N >= 0 && C > 0 && T > 0 && V > 0, STG_EINVAL, "stg_spatial_agg_bwd: bad sizes");
    STG_REQUIRE((int64_t)N * T < (1ll << 31), STG_EINVAL, "stg_spatial_agg_bwd: N*T too large");
    if (N == 0) return STG_OK;
    const int ld = V | 1;
    int rows = (64 * 1024 / 4 - C * V) / ld;
    if (rows > V) rows = V;
    STG_REQUIRE(rows >= 1, STG_ELDS, "stg_spatial_agg_bwd: V=%d C=%d exceed the LDS budget", V, C);
    const size_t lds = ((size_t)C * V + (size_t)rows * ld) * sizeof(float);
    hipLaunchKernelGGL(stg::spatial_agg_bwd_kernel, dim3((unsigned)(N * T)), dim3(256), lds,
                       stg::as_stream(stream), dy, adj, a_sn, num_peds, C, T, V, rows, dx);
    STG_LAUNCH_CHECK("stg_spatial_agg_bwd");
    return STG_OK;
}

}  // extern "C"
