// spatial_agg: the einsum of ConvTemporalGraphical.forward (model.py:67)
//     y[n,c,t,w] = sum_v x[n,c,t,v] A[n,t,v,w]              forward
//     dx[n,c,t,v] = sum_w dy[n,c,t,w] A[n,t,v,w]            backward (no dA: A is data)
// HBM-bound (A is read once: 32*V*V bytes per scene-window against 80*V*V flops).
//
// forward : one workgroup per scene; x[n] (C*T*V floats) is staged in LDS, every lane owns a
//           (t, w..w+VEC-1) output strip and streams column strips of A[n,t] from HBM with coalesced
//           VEC*4-byte loads along w (lanes consecutive in w, then in t).
// backward: one workgroup per scene; dy[n] is staged in LDS, every lane owns a ROW (t, v) of the adjacency, streams it
//           with 16-byte loads and keeps the C channel sums of its dx[., t, v] in registers.
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

// forward, all 256 lanes streaming, for V = 4 * LPR (used at V = 32): a lane owns a 16-byte piece of the OUTPUT
// row (t, w0..w0+3) for a QUARTER of the adjacency rows v -- its LPR loads all in flight -- and the four quarters of a piece sit
// four lanes apart in a 16-lane DPP row, so their partial sums meet in two rotate-adds (pure VALU).  (The strip form above
// leaves three of four waves idle at V = 32: T * V / 4 = 64 strips.)
template <int LPR>
__global__ __launch_bounds__(256) void spatial_agg_fwd_quarters_kernel(
    const float *__restrict__ x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
    const float *__restrict__ adj, int64_t a_sn, const int32_t *__restrict__ num_peds,
    int C, int T, float *__restrict__ y) {
    constexpr int V = 4 * LPR, RPL = V / 4;                          // rows per lane: a quarter of the tile
    constexpr int TL = 64 / (4 * LPR);                               // time steps per wave and pass
    extern __shared__ __attribute__((aligned(16))) float xs[];       // [C][T][V], padded rows zeroed
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *xn = x + n * x_sn;
    for (int e = tid; e < C * T * V; e += 256) {
        const int v = e % V, ct = e / V, t = ct % T, c = ct / T;
        xs[e] = v < vi ? xn[c * x_sc + t * x_st + v * x_sv] : 0.f;
    }
    __syncthreads();
    // lane = [piece low 2 bits | quarter (2 bits) | piece high bits | time step of the wave]
    const int plo = lane & 3, q = (lane >> 2) & 3, rest = lane >> 4;
    constexpr int PHB = LPR / 4;                                     // values of the piece's high part
    const int piece = plo | ((rest % PHB) << 2), tl = rest / PHB, w0 = 4 * piece;
    const float *an = adj + n * a_sn;
    float *yn = y + (int64_t)n * C * T * V;
    for (int t0 = 0; t0 < T; t0 += 4 * TL) {
        const int t = t0 + wave * TL + tl;
        const bool live = t < T;
        float4 a[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int v = q * RPL + i;
            a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live && v < vi) a[i] = *reinterpret_cast<const float4 *>(an + ((int64_t)t * V + v) * V + w0);
        }
        for (int c0 = 0; c0 < C; c0 += kAggMaxC) {
            const int cn = (C - c0) < kAggMaxC ? (C - c0) : kAggMaxC;
            float4 acc[kAggMaxC];
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *xt = xs + ((c0 * T) + (live ? t : 0)) * V + q * RPL;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
#pragma unroll
                for (int c = 0; c < kAggMaxC; ++c) {
                    if (c < cn) {
                        const float xv = xt[c * T * V + i];
                        acc[c].x = fmaf(xv, a[i].x, acc[c].x);
                        acc[c].y = fmaf(xv, a[i].y, acc[c].y);
                        acc[c].z = fmaf(xv, a[i].z, acc[c].z);
                        acc[c].w = fmaf(xv, a[i].w, acc[c].w);
                    }
                }
            }
#define STG_ROR_ADD(v_, ctrl) v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), (ctrl), 0xf, 0xf, false))
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c) {
                if (c < cn) {
                    // the four quarters: lanes l, l + 4, l + 8, l + 12 of the 16-lane row (row_ror:4, then row_ror:8)
                    STG_ROR_ADD(acc[c].x, 0x124); STG_ROR_ADD(acc[c].y, 0x124); STG_ROR_ADD(acc[c].z, 0x124); STG_ROR_ADD(acc[c].w, 0x124);
                    STG_ROR_ADD(acc[c].x, 0x128); STG_ROR_ADD(acc[c].y, 0x128); STG_ROR_ADD(acc[c].z, 0x128); STG_ROR_ADD(acc[c].w, 0x128);
                    if (live && q == 0) {
                        float4 r;
                        r.x = (w0 + 0 < vi) ? acc[c].x : 0.f;
                        r.y = (w0 + 1 < vi) ? acc[c].y : 0.f;
                        r.z = (w0 + 2 < vi) ? acc[c].z : 0.f;
                        r.w = (w0 + 3 < vi) ? acc[c].w : 0.f;
                        *reinterpret_cast<float4 *>(yn + ((int64_t)(c0 + c) * T + t) * V + w0) = r;
                    }
                }
            }
#undef STG_ROR_ADD
        }
    }
}

// backward: one workgroup per scene.  dy[n] (C*T*V floats) is staged in LDS; a lane owns one ROW (t, v) of the adjacency and
// streams it from HBM with 16-byte loads -- a wave reads 64 consecutive rows, every 128-byte line is consumed by consecutive
// instructions of the same lane (L1 hits) -- while dy[c][t][w..w+3] comes from LDS as a broadcast read (all lanes of a
// time step read the same address): one adjacency element is loaded once and used for all C channels from registers.
// (Round 2 staged the tile through LDS with scalar loads and let 160 of 256 lanes run 32-long dot products out of LDS: two
// LDS reads per multiply-add, 1.97 TB/s against the forward's 4.1 on the same bytes.)
template <int VEC>
__global__ __launch_bounds__(256) void spatial_agg_bwd_kernel(
    const float *__restrict__ dy, const float *__restrict__ adj, int64_t a_sn,
    const int32_t *__restrict__ num_peds, int C, int T, int V, float *__restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) float dys[];      // [C][T][V]
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *dyn = dy + (int64_t)n * C * T * V;
    float *dxn = dx + (int64_t)n * C * T * V;
    for (int e = tid; e < C * T * V; e += blockDim.x) {
        const int w = e % V;
        dys[e] = w < vi ? dyn[e] : 0.f;
    }
    __syncthreads();
    const float *an = adj + n * a_sn;
    const int wq = (vi + VEC - 1) / VEC;                            // column strips that hold live pedestrians
    for (int c0 = 0; c0 < C; c0 += kAggMaxC) {
        const int cn = (C - c0) < kAggMaxC ? (C - c0) : kAggMaxC;
        for (int q = tid; q < T * V; q += blockDim.x) {
            const int t = q / V, v = q - t * V;
            float acc[kAggMaxC];
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c) acc[c] = 0.f;
            if (v < vi) {
                const float *row = an + ((int64_t)t * V + v) * V;
                const float *d = dys + (c0 * T + t) * V;
                constexpr int U = 8;                                // loads in flight per lane
                for (int s0 = 0; s0 < wq; s0 += U) {
                    float a[U][VEC];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int w0 = (s0 + u) * VEC;
                        if (s0 + u < wq) {
                            if (VEC == 4) {
                                // (padded columns of A are "ignored on input": never let them in, whatever they hold)
                                const float4 a4 = *reinterpret_cast<const float4 *>(row + w0);
                                a[u][0] = a4.x;
                                a[u][1 % VEC] = w0 + 1 < vi ? a4.y : 0.f;
                                a[u][2 % VEC] = w0 + 2 < vi ? a4.z : 0.f;
                                a[u][3 % VEC] = w0 + 3 < vi ? a4.w : 0.f;
                            } else {
                                a[u][0] = row[w0];
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < VEC; ++j) a[u][j] = 0.f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int w0 = (s0 + u) * VEC;
                        if (s0 + u < wq) {
#pragma unroll
                            for (int c = 0; c < kAggMaxC; ++c) {
                                if (c < cn) {
                                    // (columns >= vi of dy are zero in LDS: a padded column of A never contributes)
                                    if (VEC == 4) {
                                        const float4 d4 = *reinterpret_cast<const float4 *>(d + c * T * V + w0);
                                        acc[c] = fmaf(d4.x, a[u][0], acc[c]);
                                        acc[c] = fmaf(d4.y, a[u][1 % VEC], acc[c]);
                                        acc[c] = fmaf(d4.z, a[u][2 % VEC], acc[c]);
                                        acc[c] = fmaf(d4.w, a[u][3 % VEC], acc[c]);
                                    } else {
                                        acc[c] = fmaf(d[c * T * V + w0], a[u][0], acc[c]);
                                    }
                                }
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < kAggMaxC; ++c)
                if (c < cn) dxn[((int64_t)(c0 + c) * T + t) * V + v] = acc[c];
        }
    }
}

// backward, coalesced form for V = 4 * 2^k (k = 1..4): LPR = V / 4 lanes share a row of the adjacency (one 16-byte piece
// each), a wave-instruction reads 64 / LPR CONSECUTIVE rows -- 1 KiB of contiguous memory, like the forward -- and the C
// partial sums of a row meet through log2(LPR) DPP adds (pure VALU, no LDS crossbar).  The pieces of TB time steps are in
// flight together; dy[c][t][w0..w0+3] of the lane's fixed column piece is fetched per (c, t) (L1 / L2 hits: every row of a
// time step reads the same 128 bytes).
template <int LPR>
__device__ __forceinline__ float row_group_sum(float v) {
#define STG_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), 0xf, 0xf, false))
    if (LPR >= 2) STG_DPP_ADD(0xB1);       // quad_perm [1,0,3,2]
    if (LPR >= 4) STG_DPP_ADD(0x4E);       // quad_perm [2,3,0,1]
    if (LPR >= 8) STG_DPP_ADD(0x141);      // row_half_mirror: every lane holds the sum of its 8 lanes
    if (LPR >= 16) STG_DPP_ADD(0x140);     // row_mirror: ... of its 16 lanes
#undef STG_DPP_ADD
    return v;
}

template <int LPR>
__global__ __launch_bounds__(256) void spatial_agg_bwd_rows_kernel(
    const float *__restrict__ dy, const float *__restrict__ adj, int64_t a_sn,
    const int32_t *__restrict__ num_peds, int C, int T, float *__restrict__ dx) {
    constexpr int V = 4 * LPR, RPI = 256 / LPR;                      // rows per workgroup pass
    constexpr int G = (V + RPI - 1) / RPI;                           // row groups of a tile (1 up to V = 32, 4 at V = 64)
    constexpr int TB = G == 1 ? 4 : 1;                               // time steps in flight together
    extern __shared__ __attribute__((aligned(16))) float dys[];      // [C][T][V]: dy[n], padded columns zeroed
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *dyn = dy + (int64_t)n * C * T * V;
    float *dxn = dx + (int64_t)n * C * T * V;
    const float *an = adj + n * a_sn;
    const int piece = tid & (LPR - 1), rsub = tid / LPR, w0 = 4 * piece;
    const bool m1 = w0 + 1 < vi, m2 = w0 + 2 < vi, m3 = w0 + 3 < vi;
    // dy through LDS: every row of a time step needs the same 4 V bytes per channel -- as global loads (L1 hits) they cost
    // the CU's address path five wave-instructions per adjacency instruction; as LDS reads nothing
    for (int e = tid; e < C * T * (V / 4); e += 256) {
        float4 q = reinterpret_cast<const float4 *>(dyn)[e];
        const int w = (e % (V / 4)) * 4;
        q.x = w < vi ? q.x : 0.f;
        q.y = w + 1 < vi ? q.y : 0.f;
        q.z = w + 2 < vi ? q.z : 0.f;
        q.w = w + 3 < vi ? q.w : 0.f;
        reinterpret_cast<float4 *>(dys)[e] = q;
    }
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += TB) {
        float4 a[TB][G];
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int v = g * RPI + rsub, t = t0 + tb;
                a[tb][g] = (t < T && v < vi && w0 < vi)
                               ? *reinterpret_cast<const float4 *>(an + ((int64_t)t * V + v) * V + w0)
                               : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int tb = 0; tb < TB; ++tb) {
            const int t = t0 + tb;
            if (t >= T) break;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // (padded columns of A are "ignored on input": never let them in, whatever they hold)
                a[tb][g].y = m1 ? a[tb][g].y : 0.f;
                a[tb][g].z = m2 ? a[tb][g].z : 0.f;
                a[tb][g].w = m3 ? a[tb][g].w : 0.f;
            }
            for (int c = 0; c < C; ++c) {
                const float4 d = *reinterpret_cast<const float4 *>(dys + (c * T + t) * V + w0);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int v = g * RPI + rsub;
                    float acc = fmaf(d.x, a[tb][g].x, fmaf(d.y, a[tb][g].y, fmaf(d.z, a[tb][g].z, d.w * a[tb][g].w)));
                    acc = row_group_sum<LPR>(acc);
                    if (piece == 0 && v < V) dxn[((int64_t)c * T + t) * V + v] = v < vi ? acc : 0.f;
                }
            }
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
    if (vec4 && V == 32 && lds <= 64 * 1024) {
        // (same box, 537 MB working set: 4.07 -> 4.82 TB/s; at V = 64 the quarters form -- 16 loads in flight per lane -- loses to
        // the strip form, 3.85 against 4.15, at V = 16 they are equal: only instantiated for 32)
        hipLaunchKernelGGL(stg::spatial_agg_fwd_quarters_kernel<8>, dim3(N), dim3(256), lds, stg::as_stream(stream), x, x_sn,
                           x_sc, x_st, x_sv, adj, a_sn, num_peds, C, T, y);
    } else if (vec4)
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
    const size_t lds = (size_t)C * T * V * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_spatial_agg_bwd: C*T*V=%d floats exceed LDS", C * T * V);
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(adj) & 15) == 0) && (a_sn % 4 == 0);
    const bool aligned = vec4 && ((reinterpret_cast<uintptr_t>(dy) & 15) == 0);
    if (aligned && (V == 8 || V == 16 || V == 32 || V == 64) && lds <= 64 * 1024) {
        // the coalesced form: rows shared by V / 4 lanes (measured at V = 32 on a 537 MB working set: see DESIGN 5)
        switch (V) {
            case 8: hipLaunchKernelGGL(stg::spatial_agg_bwd_rows_kernel<2>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj, a_sn, num_peds, C, T, dx); break;
            case 16: hipLaunchKernelGGL(stg::spatial_agg_bwd_rows_kernel<4>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj, a_sn, num_peds, C, T, dx); break;
            case 32: hipLaunchKernelGGL(stg::spatial_agg_bwd_rows_kernel<8>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj, a_sn, num_peds, C, T, dx); break;
            default: hipLaunchKernelGGL(stg::spatial_agg_bwd_rows_kernel<16>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj, a_sn, num_peds, C, T, dx); break;
        }
    } else if (vec4) {
        if (lds > 64 * 1024) {
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&stg::spatial_agg_bwd_kernel<4>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e_ != hipSuccess) return stg::hip_fail(e_, "stg_spatial_agg_bwd: hipFuncSetAttribute");
        }
        hipLaunchKernelGGL(stg::spatial_agg_bwd_kernel<4>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj,
                           a_sn, num_peds, C, T, V, dx);
    } else {
        if (lds > 64 * 1024) {
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&stg::spatial_agg_bwd_kernel<1>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e_ != hipSuccess) return stg::hip_fail(e_, "stg_spatial_agg_bwd: hipFuncSetAttribute");
        }
        hipLaunchKernelGGL(stg::spatial_agg_bwd_kernel<1>, dim3((unsigned)N), dim3(256), lds, stg::as_stream(stream), dy, adj,
                           a_sn, num_peds, C, T, V, dx);
    }
    STG_LAUNCH_CHECK("stg_spatial_agg_bwd");
    return STG_OK;
}

}  // extern "C"
