// stgcn_agg: the HBM-bound head of the first st_gcn block -- the adjacency-weighted aggregation of the CIN input
// channels and the column sums of A that the re-associated 1x1 graph convolution needs (model.py:66-67):
//     ax[n][ci][t][w] = sum_v x[n][ci][t][v] A[n][t][v][w]          cs[n][t][w] = sum_v A[n][t][v][w]
// so that g = Wg ax + bg cs (2.5x fewer flops than conv-then-aggregate, and A is read exactly ONCE per step: the
// backward works from the saved ax / cs).  This is the only kernel of the forward that touches A (32 V^2 bytes per
// scene-window): a pure streaming kernel at full occupancy, so that the scene-resident kernels behind it start from
// 96 V bytes per scene instead of paying A's HBM latency inside their dependent phases.
//
// One 256-thread workgroup per scene-window.  x[n] (strided: the caller's permute(0,3,1,2) view) is staged in LDS;
// threads own single (t, w) columns (16 row loads in flight each: at V = 32 all 256 threads of the workgroup stream,
// 32 waves per CU keep HBM busy) or, for dense crowds (V a multiple of 4 with at least 256 strips), (t, 4 consecutive
// w) strips with 16-byte loads, 8 rows in flight.  Outputs are compact ([.][T][V_n]): the layout of the saved arrays
// the block kernels index.
#include "model_common.hpp"
#include "txp_conv_bf16.hpp"
#include "scene_order.hpp"

namespace stg {

namespace {

constexpr int T = Cfg::T;

template <int CIN, int VEC>
__global__ __launch_bounds__(256) void stgcn_agg_kernel(const float *__restrict__ x, int64_t x_sn, int64_t x_sc,
                                                        int64_t x_st, int64_t x_sv, const float *__restrict__ adj,
                                                        int64_t a_sn, const int32_t *__restrict__ num_peds, int N, int V,
                                                        float *__restrict__ out, int64_t out_stride, int64_t ax_off,
                                                        int64_t cs_off, AggPrep prep) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    int blk = (int)blockIdx.x;
    if (prep.order) {
        // ragged batch: workgroup 0 -- dispatched first, so it runs beside the whole launch -- sorts the scenes by crowd
        // size (no launch of its own)
        if (blk == 0) {
            __shared__ int wave_tot[4];
            scene_order_body<4>(num_peds, N, V, prep.order, prep.key_start, prep.order_peds, reinterpret_cast<int *>(sm),
                                wave_tot);
            return;
        }
        --blk;
    }
    if (blk >= N) {
        int b = blk - N;
        // training: the workgroups behind the scenes prepare the A operands of the backward's exact-bf16
        // input-gradient GEMMs (one 16-byte vector per lane and block; txp_conv_bf16.hpp) -- no launch of their own
        const bool fwd = b >= prep.n_layers * cv::kWpVecs;
        if (fwd) b -= prep.n_layers * cv::kWpVecs;
        const int l = b / cv::kWpVecs, v = b - l * cv::kWpVecs;
        if (tid < 64) {
            if (fwd) {
                if (prep.wp_fwd)
                    cv::prep_fwd_vector(prep.params + prep.w_off[l], l == 0 ? Cfg::T : Cfg::P, v, tid,
                                        prep.wp_fwd + (int64_t)l * cv::kWpDwords);
            } else if (prep.wp) {
                cv::prep_dgrad_vector(prep.params + prep.w_off[l], l == 0 ? Cfg::T : Cfg::P, v, tid,
                                      prep.wp + (int64_t)l * cv::kWpDwords);
            }
        }
        return;
    }
    const int n = blk;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    if (vi == 0) return;
    float *xs = sm;                                   // [CIN][T][vi]
    const float *xn = x + n * x_sn;
    for (int e = tid; e < CIN * T * vi; e += 256) {
        const int v = e % vi, ct = e / vi, t = ct % T, c = ct / T;
        xs[e] = xn[c * x_sc + t * x_st + v * x_sv];
    }
    __syncthreads();
    const float *an = adj + n * a_sn;
    float *axo = out + n * out_stride + ax_off, *cso = out + n * out_stride + cs_off;
    const int vq = (vi + VEC - 1) / VEC;
    for (int q = tid; q < T * vq; q += 256) {
        const int t = q / vq, w0 = (q - t * vq) * VEC;
        float ax[CIN][VEC], cs[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            cs[j] = 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) ax[ci][j] = 0.f;
        }
        const float *at = an + (int64_t)t * V * V + w0;
        const float *xt = xs + t * vi;
        constexpr int U = VEC == 4 ? 8 : 16;          // rows in flight per lane
        for (int v0 = 0; v0 < vi; v0 += U) {
            float av[U][VEC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (v0 + u < vi) {
                    if (VEC == 4) {
                        const float4 a4 = *reinterpret_cast<const float4 *>(at + (int64_t)(v0 + u) * V);
                        av[u][0] = a4.x; av[u][1 % VEC] = a4.y; av[u][2 % VEC] = a4.z; av[u][3 % VEC] = a4.w;
                    } else {
                        av[u][0] = at[(int64_t)(v0 + u) * V];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) av[u][j] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (v0 + u < vi) {
                    float xv[CIN];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) xv[ci] = xt[ci * T * vi + v0 + u];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        cs[j] += av[u][j];
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) ax[ci][j] = fmaf(xv[ci], av[u][j], ax[ci][j]);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (w0 + j < vi) {
                cso[t * vi + w0 + j] = cs[j];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) axo[(ci * T + t) * vi + w0 + j] = ax[ci][j];
            }
        }
    }
}

// V = 32, every lane streaming 16 bytes at a time: a lane owns a 16-byte piece (t, w0..w0+3) of the outputs for a QUARTER of the
// adjacency rows -- its eight loads in flight -- and the four quarters of a piece sit four lanes apart in a 16-lane DPP row: two
// rotate-adds bring their partial sums together (the form of spatial_agg_fwd_quarters_kernel; the column form above moves the
// same bytes with four times the load instructions).
template <int CIN>
__global__ __launch_bounds__(256) void stgcn_agg_q32_kernel(const float *__restrict__ x, int64_t x_sn, int64_t x_sc,
                                                            int64_t x_st, int64_t x_sv, const float *__restrict__ adj,
                                                            int64_t a_sn, const int32_t *__restrict__ num_peds, int N,
                                                            float *__restrict__ out, int64_t out_stride, int64_t ax_off,
                                                            int64_t cs_off, AggPrep prep) {
    constexpr int V = 32, LPR = 8, RPL = 8, TL = 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    int blk = (int)blockIdx.x;
    if (prep.order) {
        if (blk == 0) {
            __shared__ int wave_tot[4];
            scene_order_body<4>(num_peds, N, V, prep.order, prep.key_start, prep.order_peds, reinterpret_cast<int *>(sm),
                                wave_tot);
            return;
        }
        --blk;
    }
    if (blk >= N) {
        int b = blk - N;
        const bool fwd = b >= prep.n_layers * cv::kWpVecs;
        if (fwd) b -= prep.n_layers * cv::kWpVecs;
        const int l = b / cv::kWpVecs, v = b - l * cv::kWpVecs;
        if (tid < 64) {
            if (fwd) {
                if (prep.wp_fwd)
                    cv::prep_fwd_vector(prep.params + prep.w_off[l], l == 0 ? Cfg::T : Cfg::P, v, tid,
                                        prep.wp_fwd + (int64_t)l * cv::kWpDwords);
            } else if (prep.wp) {
                cv::prep_dgrad_vector(prep.params + prep.w_off[l], l == 0 ? Cfg::T : Cfg::P, v, tid,
                                      prep.wp + (int64_t)l * cv::kWpDwords);
            }
        }
        return;
    }
    const int n = blk;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    if (vi == 0) return;
    const int lane = tid & 63, wave = tid >> 6;
    const int plo = lane & 3, q = (lane >> 2) & 3, rest = lane >> 4;     // lane = [piece low | quarter | piece high | time step]
    const int piece = plo | ((rest & 1) << 2), tl = rest >> 1, w0 = 4 * piece;
    const float *an = adj + n * a_sn;
    // the adjacency does not wait for x: all of this lane's rows of its first time step are requested before x is staged
    float4 a[T / (4 * TL)][RPL];
#pragma unroll
    for (int pass = 0; pass < T / (4 * TL); ++pass) {
        const int t = pass * 4 * TL + wave * TL + tl;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const int v = q * RPL + i;
            a[pass][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < vi) a[pass][i] = *reinterpret_cast<const float4 *>(an + ((int64_t)t * V + v) * V + w0);
        }
    }
    float *xs = sm;                                   // [CIN][T][V], rows past the crowd zero
    const float *xn = x + n * x_sn;
    for (int e = tid; e < CIN * T * V; e += 256) {
        const int v = e % V, ct = e / V, t = ct % T, c = ct / T;
        xs[e] = v < vi ? xn[c * x_sc + t * x_st + v * x_sv] : 0.f;
    }
    __syncthreads();
    float *axo = out + n * out_stride + ax_off, *cso = out + n * out_stride + cs_off;
#define STG_ROR_ADD(v_, ctrl) v_ += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v_), (ctrl), 0xf, 0xf, false))
#pragma unroll
    for (int pass = 0; pass < T / (4 * TL); ++pass) {
        const int t = pass * 4 * TL + wave * TL + tl;
        float4 acc[CIN + 1];
#pragma unroll
        for (int c = 0; c <= CIN; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *xt = xs + t * V + q * RPL;
#pragma unroll
        for (int i = 0; i < RPL; ++i) {
            const float4 av = a[pass][i];
            acc[CIN].x += av.x; acc[CIN].y += av.y; acc[CIN].z += av.z; acc[CIN].w += av.w;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float xv = xt[c * T * V + i];
                acc[c].x = fmaf(xv, av.x, acc[c].x);
                acc[c].y = fmaf(xv, av.y, acc[c].y);
                acc[c].z = fmaf(xv, av.z, acc[c].z);
                acc[c].w = fmaf(xv, av.w, acc[c].w);
            }
        }
#pragma unroll
        for (int c = 0; c <= CIN; ++c) {
            STG_ROR_ADD(acc[c].x, 0x124); STG_ROR_ADD(acc[c].y, 0x124); STG_ROR_ADD(acc[c].z, 0x124); STG_ROR_ADD(acc[c].w, 0x124);
            STG_ROR_ADD(acc[c].x, 0x128); STG_ROR_ADD(acc[c].y, 0x128); STG_ROR_ADD(acc[c].z, 0x128); STG_ROR_ADD(acc[c].w, 0x128);
        }
        if (q == 0) {
            // outputs are compact ([.][T][vi]): 16-byte stores when the scene fills its 32 slots, scalar ones otherwise
#pragma unroll
            for (int c = 0; c <= CIN; ++c) {
                float *dst = c < CIN ? axo + (c * T + t) * vi + w0 : cso + t * vi + w0;
                if (vi == V) {
                    *reinterpret_cast<float4 *>(dst) = acc[c];
                } else {
                    if (w0 + 0 < vi) dst[0] = acc[c].x;
                    if (w0 + 1 < vi) dst[1] = acc[c].y;
                    if (w0 + 2 < vi) dst[2] = acc[c].z;
                    if (w0 + 3 < vi) dst[3] = acc[c].w;
                }
            }
        }
    }
#undef STG_ROR_ADD
}

}  // namespace

// out + n * out_stride + ax_off : ax [cin][T][V_n];  out + n * out_stride + cs_off : cs [T][V_n]
int launch_stgcn_agg(int cin, const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj,
                     int64_t a_sn, const int32_t *num_peds, int N, int V, float *out, int64_t out_stride, int64_t ax_off,
                     int64_t cs_off, hipStream_t st, const AggPrep *prep_in) {
    if (N == 0) return STG_OK;
    AggPrep prep{};
    if (prep_in) prep = *prep_in;
    const dim3 grid(N + (prep.order ? 1 : 0) + ((prep.wp || prep.wp_fwd) ? 2 * prep.n_layers * cv::kWpVecs : 0)), block(256);
    size_t lds = (size_t)cin * T * V * sizeof(float);
    if (prep.order && lds < (size_t)(V + 1) * 4 * sizeof(int)) lds = (size_t)(V + 1) * 4 * sizeof(int);   // the sort's histogram
    STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "stgcn_agg: V=%d needs %zu bytes of LDS", V, lds);
    // 16-byte loads need 16-byte aligned rows: V a multiple of 4 and a 16-byte aligned base / batch stride
    const int min_strips = diag_env("STG_AGG_VEC_STRIPS", 256);
    const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(adj) & 15) == 0) && (a_sn % 4 == 0) &&
                     T * (V / 4) >= min_strips;
    const bool can_q32 = ((reinterpret_cast<uintptr_t>(adj) & 15) == 0) && (a_sn % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (out_stride % 4 == 0) && (ax_off % 4 == 0) &&
                         (cs_off % 4 == 0);
#define STG_AGG(CI, VE)                                                                                          \
    do {                                                                                                         \
        if (lds > 64 * 1024) {                                                                                   \
            hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&stgcn_agg_kernel<CI, VE>),       \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            if (e_ != hipSuccess) return hip_fail(e_, "stgcn_agg: hipFuncSetAttribute");                        \
        }                                                                                                        \
        hipLaunchKernelGGL((stgcn_agg_kernel<CI, VE>), grid, block, lds, st, x, x_sn, x_sc, x_st, x_sv, adj, a_sn, \
                           num_peds, N, V, out, out_stride, ax_off, cs_off, prep);                               \
    } while (0)
    // V = 32 with 16-byte aligned rows and outputs: the quarters form
    const bool q32 = V == 32 && can_q32 && !diag_env("STG_AGG_COLS", 0);
    if (q32) {
        const size_t lds32 = lds > (size_t)cin * T * 32 * sizeof(float) ? lds : (size_t)cin * T * 32 * sizeof(float);
        if (cin == Cfg::CIN0)
            hipLaunchKernelGGL((stgcn_agg_q32_kernel<Cfg::CIN0>), grid, block, lds32, st, x, x_sn, x_sc, x_st, x_sv, adj, a_sn,
                               num_peds, N, out, out_stride, ax_off, cs_off, prep);
        else
            hipLaunchKernelGGL((stgcn_agg_q32_kernel<Cfg::C>), grid, block, lds32, st, x, x_sn, x_sc, x_st, x_sv, adj, a_sn,
                               num_peds, N, out, out_stride, ax_off, cs_off, prep);
        STG_LAUNCH_CHECK("stgcn_agg");
        return STG_OK;
    }
    if (cin == Cfg::CIN0) {
        if (vec) STG_AGG(Cfg::CIN0, 4); else STG_AGG(Cfg::CIN0, 1);
    } else {
        if (vec) STG_AGG(Cfg::C, 4); else STG_AGG(Cfg::C, 1);
    }
#undef STG_AGG
    STG_LAUNCH_CHECK("stgcn_agg");
    return STG_OK;
}

}  // namespace stg
