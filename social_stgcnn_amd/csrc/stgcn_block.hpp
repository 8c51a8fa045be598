// st_gcn block (model.py:92-155) forward and backward for ONE scene resident in LDS -- device code shared by the
// workgroup-per-scene kernels (model_fwd.hip / model_bwd.hip: WAVES wave64 cooperate on a scene, __syncthreads
// between phases) and the wave-per-scene kernels (txp_wave.hip: WAVES = 0, one wave owns the scene, no workgroup
// barrier anywhere -- the other waves of the workgroup are busy with their own scenes).
//
// VALU work, 3 % of the model's flops: lanes own (t, w) columns; the einsum is re-associated to aggregate the CIN
// input channels first (g = Wg (x A) + bg colsum(A)); BatchNorm statistics are PER SCENE (the reference trains with
// N = 1, train.py:173-177) through DPP wave reductions (+ LDS across waves in workgroup mode).
#pragma once
#include "model_common.hpp"
#include "scene_team.hpp"

namespace stg {

constexpr int kRedMax = 32;   // widest block reduction (values)

// Cooperation scope of a scene: WAVES >= 1 waves of a workgroup, or (WAVES == 0) the calling wave alone.
template <int WAVES>
struct Scope {
    static constexpr int NT = WAVES ? WAVES * 64 : 64;
    static __device__ __forceinline__ int tid() { return WAVES ? (int)threadIdx.x : (int)(threadIdx.x & 63); }
    static __device__ __forceinline__ void sync() {
        if (WAVES) __syncthreads();
        else __builtin_amdgcn_wave_barrier();     // one wave: LDS operations complete in program order
    }
};

// Sum K per-thread values over the scope; every thread holds the totals on return.
template <int K, int WAVES>
__device__ __forceinline__ void block_sum(float (&v)[K], float *red) {
    wave_sum_n<K>(v);
    if (WAVES > 1) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) s += red[w * K + k];
            v[k] = s;
        }
    }
}

// Sum K per-thread values over the scope; totals land in tot[0..K) (LDS), visible to every thread on return.
template <int K, int WAVES>
__device__ __forceinline__ void block_reduce(float (&v)[K], float *red, float *tot) {
    static_assert(K <= kRedMax, "reduction too wide");
    using S = Scope<WAVES>;
    const int tid = S::tid();
    wave_sum_n<K>(v);
    if (WAVES == 0) {
        S::sync();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) tot[k] = v[k];
        }
        S::sync();
        return;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
    }
    __syncthreads();
    for (int k = tid; k < K; k += S::NT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < (WAVES ? WAVES : 1); ++w) s += red[w * K + k];
        tot[k] = s;
    }
    __syncthreads();
}

// (t, w) of column q of a scene with vi pedestrians: from the caller's position table (entry = t << 8 | w, built once
// per scene by the wave kernels) when there is one -- no integer division by the runtime vi per column and pass
typedef unsigned short ptab_t;
__device__ __forceinline__ void col_of(const ptab_t *qtab, int vi, int q, int &t, int &w) {
    if (qtab) {
        const unsigned hw = qtab[q];
        t = (int)(hw >> 8);
        w = (int)(hw & 0xffu);
    } else {
        t = q / vi;
        w = q - t * vi;
    }
}

// diagnostic build: per-phase stamps of the block inside the wave kernels (slot k of the scene's 16 stamps)
#ifdef STG_DIAG
#define STG_BLK_STAMP(k)                                                                                              \
    do {                                                                                                              \
        if constexpr (WAVES == 0) {                                                                                   \
            if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(int64_t)n * 16 + (k)] = __builtin_amdgcn_s_memtime();  \
        }                                                                                                             \
    } while (0)
#else
#define STG_BLK_STAMP(k) do { } while (0)
#endif

// What the block code needs from its caller's argument block (FwdArgs / BwdArgs / TxpFwdArgs / TxpBwdArgs all carry
// these members): lay, V, adj, a_sn, x, x_sn, x_sc, x_st, x_sv.

// ------------------------------------------------------------------------------------------
// forward.  X [CIN][T][vi] block input (LDS), G / H [C][T][vi] scratch (LDS).
//   pre_ax / pre_cs : the aggregated input x A ([CIN][T][vi]) and colsum(A) ([T][vi]) of THIS scene computed by
//                     stgcn_agg_kernel (block 0), or null -> the block streams A itself (stacked blocks, whose input
//                     only exists in LDS) and saves both for the backward.
//   workgroup mode  : on return the block output s is in H (same layout) and, when `to_txp`, also scattered into the
//                     zero-bordered TXP plane `plane` (a region of its own).
//   wave mode       : `plane` may overlap X / G / H: the outputs are formed in registers, the whole plane image
//                     (plane_zero_f4 16-byte vectors from plane_base) is zeroed, then the outputs are scattered.
// ------------------------------------------------------------------------------------------
constexpr int kWaveMaxCols = 9;      // columns per lane in wave mode: ceil(T * 68 / 64)

template <int CIN, int WAVES, typename Args>
__device__ __forceinline__ void stgcn_block_fwd(const Args &a, const float *__restrict__ P_, const float *__restrict__ B_,
                                const BlockLayout &b, int n, int vi, const float *X, float *G, float *H, float *red,
                                float *wsn, float *statn, const float *pre_ax, const float *pre_cs, bool to_txp,
                                float *plane, int plane_sc, float *plane_base, int plane_zero_f4, float *yblock,
                                bool save_s, const ptab_t *qtab = nullptr) {
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT;
    using S = Scope<WAVES>;
    constexpr int NT = S::NT;
    const int tid = S::tid(), V = a.V;
    const int cnt = T * vi;
    const bool train = a.lay.bn_mode == 1;
    const float eps = a.lay.eps;
    float *wsa = wsn ? wsn + a.lay.ws_hdr_floats : nullptr;    // saved arrays sit behind the header

    // ---- P1: aggregation of the CIN input channels + 1x1 conv (model.py:66-67) ------------------
    float s1[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s1[c] = 0.f;
    if (pre_ax) {
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float ax[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) ax[ci] = pre_ax[(ci * T + t) * vi + w];
            const float csum = pre_cs[q];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float g = P_[b.gcn_b + c] * csum;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) g = fmaf(P_[b.gcn_w + c * CIN + ci], ax[ci], g);
                G[(c * T + t) * vi + w] = g;
                if (wsn) wsa[(int64_t)b.ws_g * V + (c * T + t) * vi + w] = g;
                s1[c] += g;
            }
        }
    } else {
        const float *an = a.adj + n * a.a_sn;
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float ax[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) ax[ci] = 0.f;
            float csum = 0.f;
            const float *at = an + (int64_t)t * V * V + w;
            const float *xt = X + t * vi;
            // column w of A[n,t]: 16 row loads in flight per lane (the loop is HBM-latency-bound otherwise)
            constexpr int UA = 16;
            for (int v0 = 0; v0 < vi; v0 += UA) {
                float av[UA];
#pragma unroll
                for (int u = 0; u < UA; ++u) av[u] = (v0 + u) < vi ? at[(int64_t)(v0 + u) * V] : 0.f;
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    if (v0 + u < vi) {
                        csum += av[u];
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) ax[ci] = fmaf(xt[ci * T * vi + v0 + u], av[u], ax[ci]);
                    }
                }
            }
            if (wsn) {
                wsa[(int64_t)b.ws_cs * V + q] = csum;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) wsa[(int64_t)b.ws_ax * V + (ci * T + t) * vi + w] = ax[ci];
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float g = P_[b.gcn_b + c] * csum;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) g = fmaf(P_[b.gcn_w + c * CIN + ci], ax[ci], g);
                G[(c * T + t) * vi + w] = g;
                if (wsn) wsa[(int64_t)b.ws_g * V + (c * T + t) * vi + w] = g;
                s1[c] += g;
            }
        }
    }
    STG_BLK_STAMP(10);
    // ---- BatchNorm tcn.0 statistics (model.py:114) ------------------------------------------------
    float m1[C], r1[C];
    if (train) {
        block_sum<C, WAVES>(s1, red);
        float s2[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { m1[c] = s1[c] / (float)cnt; s2[c] = 0.f; }
        S::sync();   // G complete
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = G[(c * T + t) * vi + w] - m1[c];
                s2[c] = fmaf(d, d, s2[c]);
            }
        }
        block_sum<C, WAVES>(s2, red);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            r1[c] = 1.0f / sqrtf(s2[c] / (float)cnt + eps);
            if (statn && tid == 0) {
                statn[b.stat + c] = m1[c];
                statn[b.stat + C + c] = s2[c] / (float)(cnt - 1);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m1[c] = B_[b.buf + c];
            r1[c] = 1.0f / sqrtf(B_[b.buf + C + c] + eps);
        }
        S::sync();
    }
    if (wsn && tid == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wsn[b.ws_hdr + c] = m1[c];
            wsn[b.ws_hdr + C + c] = r1[c];
        }
    }
    STG_BLK_STAMP(11);
    // ---- P3: BN + PReLU in place (tcn.0, tcn.1) --------------------------------------------------
    {
        const float al = P_[b.prelu1];
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float v = fmaf((G[i] - m1[c]) * r1[c], P_[b.bn1_g + c], P_[b.bn1_b + c]);
                G[i] = v > 0.f ? v : al * v;
            }
        }
    }
    S::sync();
    STG_BLK_STAMP(12);
    // ---- P4: temporal conv (tcn.2) + residual 1x1 conv statistics -------------------------------
    float s2r[2 * C];
#pragma unroll
    for (int c = 0; c < 2 * C; ++c) s2r[c] = 0.f;
    float tw[C * C * KT], tb[C];        // temporal conv weights: registers for the pass
#pragma unroll
    for (int k = 0; k < C * C * KT; ++k) tw[k] = P_[b.tcn_w + k];
#pragma unroll
    for (int c = 0; c < C; ++c) tb[c] = P_[b.tcn_b + c];
    for (int q = tid; q < cnt; q += NT) {
        int t, w;
            col_of(qtab, vi, q, t, w);
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = tb[c];
#pragma unroll
        for (int dt = 0; dt < KT; ++dt) {
            const int ti = t + dt - (KT - 1) / 2;
            if (ti < 0 || ti >= T) continue;
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const float hv = G[(ci * T + ti) * vi + w];
#pragma unroll
                for (int c = 0; c < C; ++c) h[c] = fmaf(tw[(c * C + ci) * KT + dt], hv, h[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            H[(c * T + t) * vi + w] = h[c];
            if (wsn) wsa[(int64_t)b.ws_h2 * V + (c * T + t) * vi + w] = h[c];
            s2r[c] += h[c];
        }
        if (b.residual == 2) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float r = P_[b.res_b + c];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) r = fmaf(P_[b.res_w + c * CIN + ci], X[(ci * T + t) * vi + w], r);
                s2r[C + c] += r;
            }
        }
    }
    STG_BLK_STAMP(13);
    float m2[C], r2[C], mr[C], rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { mr[c] = 0.f; rr[c] = 0.f; }
    if (train) {
        block_sum<2 * C, WAVES>(s2r, red);
        float v2r[2 * C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m2[c] = s2r[c] / (float)cnt;
            mr[c] = s2r[C + c] / (float)cnt;
            v2r[c] = 0.f;
            v2r[C + c] = 0.f;
        }
        S::sync();   // H complete
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float d = H[(c * T + t) * vi + w] - m2[c];
                v2r[c] = fmaf(d, d, v2r[c]);
            }
            if (b.residual == 2) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float r = P_[b.res_b + c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
                        r = fmaf(P_[b.res_w + c * CIN + ci], X[(ci * T + t) * vi + w], r);
                    const float d = r - mr[c];
                    v2r[C + c] = fmaf(d, d, v2r[C + c]);
                }
            }
        }
        block_sum<2 * C, WAVES>(v2r, red);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            r2[c] = 1.0f / sqrtf(v2r[c] / (float)cnt + eps);
            rr[c] = 1.0f / sqrtf(v2r[C + c] / (float)cnt + eps);
            if (statn && tid == 0) {
                statn[b.stat + 2 * C + c] = m2[c];
                statn[b.stat + 3 * C + c] = v2r[c] / (float)(cnt - 1);
                if (b.residual == 2) {
                    statn[b.stat + 4 * C + c] = mr[c];
                    statn[b.stat + 5 * C + c] = v2r[C + c] / (float)(cnt - 1);
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m2[c] = B_[b.buf + 2 * C + c];
            r2[c] = 1.0f / sqrtf(B_[b.buf + 3 * C + c] + eps);
            if (b.residual == 2) {
                mr[c] = B_[b.buf + 4 * C + c];
                rr[c] = 1.0f / sqrtf(B_[b.buf + 5 * C + c] + eps);
            }
        }
        S::sync();
    }
    if (wsn && tid == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wsn[b.ws_hdr + 2 * C + c] = m2[c];
            wsn[b.ws_hdr + 3 * C + c] = r2[c];
            wsn[b.ws_hdr + 4 * C + c] = mr[c];
            wsn[b.ws_hdr + 5 * C + c] = rr[c];
        }
    }
    STG_BLK_STAMP(14);
    // ---- P6: BN (tcn.3) + residual + PReLU (model.py:150-153) ------------------------------------
    const float ao = P_[b.prelu_o];
    const int SW = txp_sw(vi), SC = plane_sc;
    auto out_value = [&](int c, int t, int w) -> float {
        const int i = (c * T + t) * vi + w;
        float u = fmaf((H[i] - m2[c]) * r2[c], P_[b.bn2_g + c], P_[b.bn2_b + c]);
        if (b.residual == 2) {
            float r = P_[b.res_b + c];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) r = fmaf(P_[b.res_w + c * CIN + ci], X[(ci * T + t) * vi + w], r);
            u += fmaf((r - mr[c]) * rr[c], P_[b.bnr_g + c], P_[b.bnr_b + c]);
        } else if (b.residual == 1) {
            if (CIN == C) u += X[i];
        }
        return (a.lay.use_mdn || u > 0.f) ? u : ao * u;
    };
    if (WAVES == 0) {
        // wave mode: the plane image overlaps X / G / H.  Every lane forms the outputs of its columns in registers,
        // then the wave zeroes the whole image and scatters them.
        float sv[kWaveMaxCols][C];
#pragma unroll
        for (int k = 0; k < kWaveMaxCols; ++k) {
            const int q = tid + 64 * k;
            if (q < cnt) {
                int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
                for (int c = 0; c < C; ++c) sv[k][c] = out_value(c, t, w);
            }
        }
        S::sync();
        if (to_txp) {
            float4 *z4 = reinterpret_cast<float4 *>(plane_base);
            for (int e = tid; e < plane_zero_f4; e += 64) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            S::sync();
        }
#pragma unroll
        for (int k = 0; k < kWaveMaxCols; ++k) {
            const int q = tid + 64 * k;
            if (q < cnt) {
                int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float s = sv[k][c];
                    if (wsn && save_s) wsa[(int64_t)b.ws_s * V + (c * T + t) * vi + w] = s;
                    if (to_txp) {
                        const int f = c * T + t, ch = f / C, row = f - ch * C;
                        plane[ch * SC + (row + 1) * SW + (w + 1)] = s;
                    } else {
                        H[(c * T + t) * vi + w] = s;
                    }
                    if (yblock) yblock[(int64_t)(c * T + t) * V + w] = s;
                }
            }
        }
    } else {
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float s = out_value(c, t, w);
                H[i] = s;
                if (wsn && save_s) wsa[(int64_t)b.ws_s * V + i] = s;
                if (to_txp) {
                    // v.view(N, T, C, V) (model.py:187): flat plane index f = c*T+t -> (f / C, f % C)
                    const int f = c * T + t, ch = f / C, row = f - ch * C;
                    plane[ch * SC + (row + 1) * SW + (w + 1)] = s;
                }
                if (yblock) yblock[(int64_t)(c * T + t) * V + w] = s;
            }
        }
    }
    S::sync();
}

// ------------------------------------------------------------------------------------------
// forward, COLUMN mode (wave kernels, V_n <= 64): lane w owns pedestrian w and keeps all T time steps of every
// channel in registers -- the temporal convolution, the residual branch and both BatchNorm applications never touch
// LDS, and the whole block is ~2k straight-line VALU instructions instead of seven LDS round trips over runtime-trip
// loops.  Cross-lane traffic: the 30 DPP wave sums of the per-scene BatchNorm statistics, nothing else.
// The outputs go straight into the zero-bordered TXP plane image (zeroed here first; nothing of the block lives in
// LDS).  P_ / B_: the block's parameters / running statistics (the workgroup's LDS copy).  First-block shape only:
// CIN0 input channels, residual = conv + BN (2) or none (0).
// ------------------------------------------------------------------------------------------
// HALF (V_n <= 32, register hand-over only): lane = (pedestrian, time half) -- lanes 0..31 own t = 0..3, lanes 32..63
// t = 4..7; the temporal conv's taps across the half boundary come from the partner lane (lane ^ 32), and `regs_out`
// receives the 20 outputs f = 20*half .. 20*half+19 of the pedestrian (plane channels 4*half .. 4*half+3 of its 5 rows).
// CK (scene_team.hpp): SoloScene -- the wave owns the scene -- or TeamScene: the wave owns the column chunk
// [ck.w0(), ck.w0() + ck.wc()) of a scene shared by ck.nch() waves (HALF only; the scene-wide BatchNorm sums are
// exchanged through LDS, `plane_base` is the team's shared image and is zeroed by all of them, `qtab` is the wave's own
// table of its chunk's positions).
template <bool HALF, typename Args, typename CK>
__device__ __forceinline__ void stgcn_block_fwd_cols(const Args &a, const float *P_, const float *B_, const BlockLayout &b,
                                                     int n, int vi, float *wsn, float *statn, const float *pre_ax,
                                                     const float *pre_cs, float *plane, int plane_sc, float *plane_base,
                                                     int plane_zero_f4, ptab_t *qtab, float *regs_out, const CK &ck) {
    // regs_out (a register array of C*T floats in the caller): the outputs of pedestrian w, flat index f = c*T+t, are
    // handed back instead of written to `plane` (the exact-bf16 forward splits and stores them itself)
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT, CIN = Cfg::CIN0;
    [[maybe_unused]] constexpr int WAVES = 0;          // (diagnostic stamps)
    constexpr int TL = HALF ? T / 2 : T;               // time steps of a lane
    static_assert(HALF || !CK::kTeam, "a team of waves shares a scene in (pedestrian, time half) mode only");
    const int lane = threadIdx.x & 63, V = a.V;
    const int wl = HALF ? lane & 31 : lane, hh = HALF ? lane >> 5 : 0, toff = TL * hh;
    const int w = ck.w0() + wl;                        // the lane's pedestrian
    const bool act = wl < ck.wc(), lane0 = lane == 0 && ck.lead();
    const float fact = act ? 1.f : 0.f;
    const int cnt = T * vi;
    const bool train = a.lay.bn_mode == 1;
    const float eps = a.lay.eps, inv_cnt = 1.0f / (float)cnt;
    float *wsa = wsn ? wsn + a.lay.ws_hdr_floats : nullptr;
    // ---- inputs of pedestrian w: block input x (strided view), aggregated input ax, colsum cs ----------------
    float x[CIN][TL], g[C][TL];
    {
        float ax[CIN][TL], cs[TL];
        const float *xn = a.x + n * a.x_sn + w * a.x_sv;
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const int tg = toff + t;                   // the lane's time step t is step tg of the scene
            cs[t] = act ? pre_cs[tg * vi + w] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                ax[ci][t] = act ? pre_ax[(ci * T + tg) * vi + w] : 0.f;
                x[ci][t] = act ? xn[ci * a.x_sc + tg * a.x_st] : 0.f;
            }
        }
        // (while those loads are in flight) the TXP plane image: zeros everywhere but the interior this block writes
        // at the end; the scene's position table
        {
            float4 *z4 = reinterpret_cast<float4 *>(plane_base);
            for (int e = lane + 64 * ck.ci(); e < plane_zero_f4; e += 64 * ck.nch()) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int wcw = ck.wc();
            for (int p = lane; p < T * wcw; p += 64) {
                const int h = p / wcw;
                qtab[p] = (ptab_t)((h << 8) | (ck.w0() + p - h * wcw));
            }
        }
        // ---- gcn 1x1 conv on the aggregated input (model.py:66-67) --------------------------------------
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float bg = P_[b.gcn_b + c];
            float wg[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) wg[ci] = P_[b.gcn_w + c * CIN + ci];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                float v = bg * cs[t];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) v = fmaf(wg[ci], ax[ci][t], v);
                g[c][t] = v;
                if (wsa && act) wsa[(int64_t)b.ws_g * V + (c * T + toff + t) * vi + w] = v;
            }
        }
    }
    STG_BLK_STAMP(9);
    // ---- BatchNorm tcn.0 (per-scene statistics in training) + PReLU ---------------------------------------
    float m1[C], r1[C];
    if (train) {
        float s1[C], s2[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            s1[c] = 0.f;
#pragma unroll
            for (int t = 0; t < TL; ++t) s1[c] += g[c][t];            // (inactive lanes hold zeros)
        }
        ck.template sum<C>(s1, kXrF_s1);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m1[c] = s1[c] * inv_cnt;
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float d = g[c][t] - m1[c];
                s = fmaf(d, d, s);
            }
            s2[c] = s * fact;
        }
        ck.template sum<C>(s2, kXrF_s2);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            r1[c] = 1.0f / sqrtf(s2[c] * inv_cnt + eps);
            if (statn && lane0) {
                statn[b.stat + c] = m1[c];
                statn[b.stat + C + c] = s2[c] / (float)(cnt - 1);
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m1[c] = B_[b.buf + c];
            r1[c] = 1.0f / sqrtf(B_[b.buf + C + c] + eps);
        }
    }
    STG_BLK_STAMP(10);
    {
        const float al = P_[b.prelu1];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float ga = P_[b.bn1_g + c], be = P_[b.bn1_b + c];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float v = fmaf((g[c][t] - m1[c]) * r1[c], ga, be);
                g[c][t] = v > 0.f ? v : al * v;                    // h1
            }
        }
    }
    STG_BLK_STAMP(11);
    // ---- temporal conv tcn.2 (all taps in this lane's registers) ----------------------------------------
    // h1 with its two neighbours in time: zero outside the scene, the partner lane's edge value across the half boundary
    float he[C][TL + 2];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
#pragma unroll
        for (int t = 0; t < TL; ++t) he[ci][t + 1] = g[ci][t];
        float lo = 0.f, hi = 0.f;
        if (HALF) {
            const float p_last = __shfl_xor(g[ci][TL - 1], 32, 64), p_first = __shfl_xor(g[ci][0], 32, 64);
            lo = hh == 1 ? p_last : 0.f;
            hi = hh == 0 ? p_first : 0.f;
        }
        he[ci][0] = lo;
        he[ci][TL + 1] = hi;
    }
    float h2[C][TL];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float tb = P_[b.tcn_b + c];
#pragma unroll
        for (int t = 0; t < TL; ++t) h2[c][t] = tb;
#pragma unroll
        for (int ci = 0; ci < C; ++ci)
#pragma unroll
            for (int dt = 0; dt < KT; ++dt) {
                const float wv = P_[b.tcn_w + (c * C + ci) * KT + dt];
#pragma unroll
                for (int t = 0; t < TL; ++t) {
                    const int ti = t + dt - (KT - 1) / 2;           // h1 time step of this tap
                    if (!HALF && (ti < 0 || ti >= T)) continue;      // a known zero
                    h2[c][t] = fmaf(wv, he[ci][ti + 1], h2[c][t]);
                }
            }
        if (wsa && act) {
#pragma unroll
            for (int t = 0; t < TL; ++t) wsa[(int64_t)b.ws_h2 * V + (c * T + toff + t) * vi + w] = h2[c][t];
        }
    }
    STG_BLK_STAMP(12);
    // ---- BatchNorm tcn.3 / residual.1 statistics ---------------------------------------------------------
    auto res_val = [&](int c, int t, float rb, const float (&rw)[CIN]) -> float {
        float r = rb;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) r = fmaf(rw[ci], x[ci][t], r);
        return r;
    };
    float m2[C], r2[C], mr[C], rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { mr[c] = 0.f; rr[c] = 0.f; }
    if (train) {
        float sm2[2 * C], sv2[2 * C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float s = 0.f, sr = 0.f;
            float rw[CIN];
            const float rb = b.residual == 2 ? P_[b.res_b + c] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rw[ci] = b.residual == 2 ? P_[b.res_w + c * CIN + ci] : 0.f;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                s += h2[c][t];
                if (b.residual == 2) sr += res_val(c, t, rb, rw);
            }
            sm2[c] = s * fact;
            sm2[C + c] = sr * fact;
        }
        ck.template sum<2 * C>(sm2, kXrF_sm2);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m2[c] = sm2[c] * inv_cnt;
            mr[c] = b.residual == 2 ? sm2[C + c] * inv_cnt : 0.f;
            float rw[CIN];
            const float rb = b.residual == 2 ? P_[b.res_b + c] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rw[ci] = b.residual == 2 ? P_[b.res_w + c * CIN + ci] : 0.f;
            float v = 0.f, vr = 0.f;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float d = h2[c][t] - m2[c];
                v = fmaf(d, d, v);
                if (b.residual == 2) {
                    const float dr = res_val(c, t, rb, rw) - mr[c];
                    vr = fmaf(dr, dr, vr);
                }
            }
            sv2[c] = v * fact;
            sv2[C + c] = vr * fact;
        }
        ck.template sum<2 * C>(sv2, kXrF_sv2);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            r2[c] = 1.0f / sqrtf(sv2[c] * inv_cnt + eps);
            rr[c] = 1.0f / sqrtf((b.residual == 2 ? sv2[C + c] * inv_cnt : 0.f) + eps);
            if (statn && lane0) {
                statn[b.stat + 2 * C + c] = m2[c];
                statn[b.stat + 3 * C + c] = sv2[c] / (float)(cnt - 1);
                if (b.residual == 2) {
                    statn[b.stat + 4 * C + c] = mr[c];
                    statn[b.stat + 5 * C + c] = sv2[C + c] / (float)(cnt - 1);
                }
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            m2[c] = B_[b.buf + 2 * C + c];
            r2[c] = 1.0f / sqrtf(B_[b.buf + 3 * C + c] + eps);
            if (b.residual == 2) {
                mr[c] = B_[b.buf + 4 * C + c];
                rr[c] = 1.0f / sqrtf(B_[b.buf + 5 * C + c] + eps);
            }
        }
    }
    STG_BLK_STAMP(13);
    if (wsn && lane0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wsn[b.ws_hdr + c] = m1[c];
            wsn[b.ws_hdr + C + c] = r1[c];
            wsn[b.ws_hdr + 2 * C + c] = m2[c];
            wsn[b.ws_hdr + 3 * C + c] = r2[c];
            wsn[b.ws_hdr + 4 * C + c] = mr[c];
            wsn[b.ws_hdr + 5 * C + c] = rr[c];
        }
    }
    STG_BLK_STAMP(14);
    // ---- BN (tcn.3) + residual + PReLU (model.py:150-153) -> the TXP plane -------------------------------
    ck.sync();                                        // (the image is zero before its interior is written)
    const float ao = P_[b.prelu_o];
    const int SW = txp_sw(vi);
    float *pw = plane + (w + 1);
    [[maybe_unused]] float so[C][TL];                  // (HALF) the lane's outputs before the hand-over
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g2 = P_[b.bn2_g + c], b2 = P_[b.bn2_b + c];
        float rw[CIN];
        const float rb = b.residual == 2 ? P_[b.res_b + c] : 0.f;
        const float gr = b.residual == 2 ? P_[b.bnr_g + c] : 0.f, br = b.residual == 2 ? P_[b.bnr_b + c] : 0.f;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) rw[ci] = b.residual == 2 ? P_[b.res_w + c * CIN + ci] : 0.f;
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            float u = fmaf((h2[c][t] - m2[c]) * r2[c], g2, b2);
            if (b.residual == 2) u += fmaf((res_val(c, t, rb, rw) - mr[c]) * rr[c], gr, br);
            const float s = (a.lay.use_mdn || u > 0.f) ? u : ao * u;
            if constexpr (HALF) {
                so[c][t] = act ? s : 0.f;
            } else {
                // v.view(N, T, C, V) (model.py:187): flat plane index f = c*T+t -> (f / C, f % C), static here
                const int f = c * T + t, ch = f / C, row = f - ch * C;
                if (regs_out) regs_out[f] = act ? s : 0.f;
                else if (act) pw[ch * plane_sc + (row + 1) * SW] = s;
            }
        }
    }
    if constexpr (HALF) {
        // The pedestrian's 40 outputs f = c*T + t; half 0 hands over f = 0..19 (c = 0, 1 and t < 4 of c = 2), half 1
        // f = 20..39 (t >= 4 of c = 2 and c = 3, 4): each lane is short of two channels' other time half, which the
        // partner lane holds.
        static_assert(!HALF || (C == 5 && T == 8), "the hand-over below is written for the 5 x 8 block output");
        float rcv[2][TL];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < TL; ++k) rcv[i][k] = __shfl_xor(hh == 0 ? so[3 + i][k] : so[i][k], 32, 64);
#pragma unroll
        for (int k = 0; k < TL; ++k) {
            regs_out[k] = hh == 0 ? so[0][k] : so[2][k];
            regs_out[4 + k] = rcv[0][k];
            regs_out[8 + k] = hh == 0 ? so[1][k] : so[3][k];
            regs_out[12 + k] = rcv[1][k];
            regs_out[16 + k] = hh == 0 ? so[2][k] : so[4][k];
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// the wave owns the scene
template <bool HALF = false, typename Args>
__device__ __forceinline__ void stgcn_block_fwd_cols(const Args &a, const float *P_, const float *B_, const BlockLayout &b,
                                                     int n, int vi, float *wsn, float *statn, const float *pre_ax,
                                                     const float *pre_cs, float *plane, int plane_sc, float *plane_base,
                                                     int plane_zero_f4, ptab_t *qtab, float *regs_out = nullptr) {
    stgcn_block_fwd_cols<HALF>(a, P_, B_, b, n, vi, wsn, statn, pre_ax, pre_cs, plane, plane_sc, plane_base, plane_zero_f4,
                               qtab, regs_out, SoloScene{vi});
}

// ------------------------------------------------------------------------------------------
// backward, column mode (wave-per-scene kernels, vi <= 64, first-block shape, no input gradient): lane = pedestrian, the
// 8 time steps of every array in registers -- the mirror of stgcn_block_fwd_cols.  Nothing of the block lives in LDS
// except ds ([C][T][vi] in `D`, read once); the saved arrays g, h2, ax, colsum and the block input come straight from
// HBM (all ~130 loads of a lane in flight at once); cross-lane traffic: the DPP wave sums of the 142 parameter
// gradients and BatchNorm reductions, in five batches.  Gradients leave as the scene's own row (`row[b.* + k]`, every
// entry written exactly once, by lane 0).
// ------------------------------------------------------------------------------------------
// HALF (vi <= 32): lane = (pedestrian, time half) -- lanes 0..31 own t = 0..3, lanes 32..63 t = 4..7 -- so that a crowd of
// <= 32 uses every lane and the per-lane arrays halve; the temporal conv's two taps across the half boundary come from
// the partner lane (lane ^ 32).
// CK: SoloScene, or TeamScene -- the wave owns one column chunk of a scene shared by several waves (HALF only): the two
// reductions the BatchNorm backward goes on with are scene-wide sums (exchanged through LDS), the parameter gradients are
// parked per wave (scene_team.hpp) -- the caller adds the parked rows into the scene's row afterwards.
template <bool HALF, typename Args, typename CK>
__device__ __forceinline__ void stgcn_block_bwd_cols(const Args &a, const float *P_, const BlockLayout &b, int n, int vi,
                                                     const float *D, float *row, const float *wsn, const CK &ck) {
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT, CIN = Cfg::CIN0;
    constexpr int TL = HALF ? T / 2 : T;               // time steps of a lane
    static_assert(HALF || !CK::kTeam, "a team of waves shares a scene in (pedestrian, time half) mode only");
    const int lane = threadIdx.x & 63, V = a.V;
    const int wl = HALF ? lane & 31 : lane, hh = HALF ? lane >> 5 : 0, toff = TL * hh;
    const int w = ck.w0() + wl;                        // the lane's pedestrian
    const bool act = wl < ck.wc(), lane0 = lane == 0 && ck.lead();
    // where the gradients go: the scene's row (a wave that owns its scene), or this wave's parked row in LDS (a team: the
    // leading wave adds the rows at the end of the scene).  Totals of scene-wide sums (ck.sum) are written by lane0,
    // totals of per-wave sums (ck.reduce) by ck.writer().
    row = ck.row(row);
    const bool wr = ck.writer();
    const bool train = a.lay.bn_mode == 1, res2 = b.residual == 2;
    const float inv_cnt = 1.0f / (float)(T * vi);
    const float *wsa = wsn + a.lay.ws_hdr_floats;
    const float *w_ax = wsa + (int64_t)b.ws_ax * V, *w_cs = wsa + (int64_t)b.ws_cs * V;
    const float *w_g = wsa + (int64_t)b.ws_g * V, *w_h2 = wsa + (int64_t)b.ws_h2 * V;
    const float *hdr = wsn + b.ws_hdr;
    float m1[C], r1[C], m2[C], r2[C], mr[C], rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        m1[c] = hdr[c]; r1[c] = hdr[C + c]; m2[c] = hdr[2 * C + c];
        r2[c] = hdr[3 * C + c]; mr[c] = hdr[4 * C + c]; rr[c] = hdr[5 * C + c];
    }
    // ---- loads: everything this pedestrian needs, in flight together ------------------------------------------------
    float ds[C][TL], h2[C][TL], g[C][TL], ax[CIN][TL], cs[TL], x[CIN][TL];
    {
        const float *xn = a.x + n * a.x_sn + w * a.x_sv;
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const int tg = toff + t;                   // the lane's time step t is step tg of the scene
            cs[t] = act ? w_cs[tg * vi + w] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                ax[ci][t] = act ? w_ax[(ci * T + tg) * vi + w] : 0.f;
                x[ci][t] = (act && b.residual != 0) ? xn[ci * a.x_sc + tg * a.x_st] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + tg) * vi + w;
                h2[c][t] = act ? w_h2[i] : 0.f;
                g[c][t] = act ? w_g[i] : 0.f;
                ds[c][t] = act ? D[i] : 0.f;
            }
        }
    }
    // ---- B1: du = ds * prelu'(u); BatchNorm tcn.3 / residual.1 reductions ------------------------------------------------
    const float ao = P_[b.prelu_o], a1 = P_[b.prelu1];
    float s1[3 * C + 1];
#pragma unroll
    for (int k = 0; k < 3 * C + 1; ++k) s1[k] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g2 = P_[b.bn2_g + c], b2 = P_[b.bn2_b + c];
        const float gr = res2 ? P_[b.bnr_g + c] : 0.f, br = res2 ? P_[b.bnr_b + c] : 0.f, rb = res2 ? P_[b.res_b + c] : 0.f;
        float rw[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) rw[ci] = res2 ? P_[b.res_w + c * CIN + ci] : 0.f;
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const float x2 = (h2[c][t] - m2[c]) * r2[c];
            float u = fmaf(x2, g2, b2), xr = 0.f;
            if (res2) {
                float r = rb;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) r = fmaf(rw[ci], x[ci][t], r);
                xr = (r - mr[c]) * rr[c];
                u += fmaf(xr, gr, br);
            }
            const float d = ds[c][t];
            float du = d;
            if (!a.lay.use_mdn && !(u > 0.f)) {
                du = ao * d;
                s1[3 * C] = fmaf(d, u, s1[3 * C]);
            }
            ds[c][t] = du;                             // (ds now holds du)
            h2[c][t] = x2;                             // (h2 now holds xhat2)
            s1[c] += du;
            s1[C + c] = fmaf(du, x2, s1[C + c]);
            s1[2 * C + c] = fmaf(du, xr, s1[2 * C + c]);
        }
    }
    ck.template sum<3 * C + 1>(s1, kXrB_s1);
    if (lane0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            row[b.bn2_b + c] = s1[c];
            row[b.bn2_g + c] = s1[C + c];
            if (res2) {
                row[b.bnr_b + c] = s1[c];
                row[b.bnr_g + c] = s1[2 * C + c];
            }
        }
        row[b.prelu_o] = s1[3 * C];
    }
    // ---- B2: dh2, dr; residual 1x1 conv gradients --------------------------------------------------------------------
    {
        float s2[C * CIN + C];
#pragma unroll
        for (int k = 0; k < C * CIN + C; ++k) s2[k] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float mdu = train ? s1[c] * inv_cnt : 0.f, mdx2 = train ? s1[C + c] * inv_cnt : 0.f;
            const float mdxr = train ? s1[2 * C + c] * inv_cnt : 0.f;
            const float k2 = P_[b.bn2_g + c] * r2[c];
            const float gr = res2 ? P_[b.bnr_g + c] : 0.f, rb = res2 ? P_[b.res_b + c] : 0.f;
            float rw[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rw[ci] = res2 ? P_[b.res_w + c * CIN + ci] : 0.f;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float du = ds[c][t];
                if (res2) {
                    float r = rb;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) r = fmaf(rw[ci], x[ci][t], r);
                    const float xr = (r - mr[c]) * rr[c];
                    const float dr = act ? gr * rr[c] * (du - mdu - xr * mdxr) : 0.f;
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) s2[c * CIN + ci] = fmaf(dr, x[ci][t], s2[c * CIN + ci]);
                    s2[C * CIN + c] += dr;
                }
                h2[c][t] = act ? k2 * (du - mdu - h2[c][t] * mdx2) : 0.f;      // (h2 now holds dh2)
            }
        }
        if (res2) {
            ck.template reduce<C * CIN + C>(s2);
            if (wr) {
#pragma unroll
                for (int k = 0; k < C * CIN; ++k) row[b.res_w + k] = s2[k];
#pragma unroll
                for (int c = 0; c < C; ++c) row[b.res_b + c] = s2[C * CIN + c];
            }
        }
    }
    // ---- B3a: h1 = prelu(bn1(g)) (into the ds registers); temporal conv weight gradients, one tap at a time -------------
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float g1 = P_[b.bn1_g + c], b1p = P_[b.bn1_b + c];
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const float b1 = fmaf((g[c][t] - m1[c]) * r1[c], g1, b1p);
            ds[c][t] = b1 > 0.f ? b1 : a1 * b1;
        }
    }
    // h1 with its two neighbours in time: zero outside the scene, the partner lane's edge value across the half boundary
    float he[C][TL + 2];
#pragma unroll
    for (int ci = 0; ci < C; ++ci) {
#pragma unroll
        for (int t = 0; t < TL; ++t) he[ci][t + 1] = ds[ci][t];
        float lo = 0.f, hi = 0.f;
        if (HALF) {
            const float p_last = __shfl_xor(ds[ci][TL - 1], 32, 64), p_first = __shfl_xor(ds[ci][0], 32, 64);
            lo = hh == 1 ? p_last : 0.f;
            hi = hh == 0 ? p_first : 0.f;
        }
        he[ci][0] = lo;
        he[ci][TL + 1] = hi;
    }
#pragma unroll
    for (int dt = 0; dt < KT; ++dt) {
        float sw[C * C];
#pragma unroll
        for (int k = 0; k < C * C; ++k) sw[k] = 0.f;
#pragma unroll
        for (int t = 0; t < TL; ++t) {                 // h1 at time t + dt - 1
            if (!HALF && (t + dt - 1 < 0 || t + dt - 1 >= T)) continue;   // a known zero
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int ci = 0; ci < C; ++ci) sw[c * C + ci] = fmaf(h2[c][t], he[ci][t + dt], sw[c * C + ci]);
        }
        ck.template reduce<C * C>(sw);
        if (wr) {
#pragma unroll
            for (int k = 0; k < C * C; ++k) row[b.tcn_w + k * KT + dt] = sw[k];
        }
    }
    // ---- B3b: dh1 -> db1, conv bias gradient, BatchNorm tcn.0 reductions, PReLU slope -------------------------------------
    float s3[3 * C + 1];
#pragma unroll
    for (int k = 0; k < 3 * C + 1; ++k) s3[k] = 0.f;
    {
        float dh1[C][TL];
#pragma unroll
        for (int ci = 0; ci < C; ++ci)
#pragma unroll
            for (int t = 0; t < TL; ++t) dh1[ci][t] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            // dh2 with its two neighbours in time (as h1 above)
            float de[TL + 2];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                de[t + 1] = h2[c][t];
                s3[c] += h2[c][t];
            }
            float lo = 0.f, hi = 0.f;
            if (HALF) {
                const float p_last = __shfl_xor(h2[c][TL - 1], 32, 64), p_first = __shfl_xor(h2[c][0], 32, 64);
                lo = hh == 1 ? p_last : 0.f;
                hi = hh == 0 ? p_first : 0.f;
            }
            de[0] = lo;
            de[TL + 1] = hi;
#pragma unroll
            for (int ci = 0; ci < C; ++ci)
#pragma unroll
                for (int dt = 0; dt < KT; ++dt) {
                    const float wv = P_[b.tcn_w + (c * C + ci) * KT + dt];
                    // dh1[ci][t] += W[c][ci][dt] dh2[c][t - dt + 1]
#pragma unroll
                    for (int t = 0; t < TL; ++t) {
                        if (!HALF && (t + 1 - dt < 0 || t + 1 - dt >= T)) continue;   // a known zero
                        dh1[ci][t] = fmaf(wv, de[t + 2 - dt], dh1[ci][t]);
                    }
                }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float g1 = P_[b.bn1_g + c], b1p = P_[b.bn1_b + c];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float x1 = (g[c][t] - m1[c]) * r1[c];
                const float b1 = fmaf(x1, g1, b1p);
                float db = dh1[c][t];
                if (!(b1 > 0.f)) {
                    db = a1 * dh1[c][t];
                    s3[3 * C] = fmaf(dh1[c][t], b1, s3[3 * C]);
                }
                ds[c][t] = db;                         // (ds now holds db1)
                g[c][t] = x1;                          // (g now holds xhat1)
                s3[C + c] += db;
                s3[2 * C + c] = fmaf(db, x1, s3[2 * C + c]);
            }
        }
    }
    ck.template sum<3 * C + 1>(s3, kXrB_s3);
    if (lane0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            row[b.tcn_b + c] = s3[c];
            row[b.bn1_b + c] = s3[C + c];
            row[b.bn1_g + c] = s3[2 * C + c];
        }
        row[b.prelu1] = s3[3 * C];
    }
    // ---- B4: dg; gcn 1x1 conv gradients ------------------------------------------------------------------------------
    {
        float s4[C * CIN + C];
#pragma unroll
        for (int k = 0; k < C * CIN + C; ++k) s4[k] = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float mdb = train ? s3[C + c] * inv_cnt : 0.f, mdbx = train ? s3[2 * C + c] * inv_cnt : 0.f;
            const float k1 = P_[b.bn1_g + c] * r1[c];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const float dg = act ? k1 * (ds[c][t] - mdb - g[c][t] * mdbx) : 0.f;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) s4[c * CIN + ci] = fmaf(dg, ax[ci][t], s4[c * CIN + ci]);
                s4[C * CIN + c] = fmaf(dg, cs[t], s4[C * CIN + c]);
            }
        }
        ck.template reduce<C * CIN + C>(s4);
        if (wr) {
#pragma unroll
            for (int k = 0; k < C * CIN; ++k) row[b.gcn_w + k] = s4[k];
#pragma unroll
            for (int c = 0; c < C; ++c) row[b.gcn_b + c] = s4[C * CIN + c];
        }
    }
}
// the wave owns the scene
template <bool HALF, typename Args>
__device__ __forceinline__ void stgcn_block_bwd_cols(const Args &a, const float *P_, const BlockLayout &b, int n, int vi,
                                                     const float *D, float *row, const float *wsn) {
    stgcn_block_bwd_cols<HALF>(a, P_, b, n, vi, D, row, wsn, SoloScene{vi});
}

// ------------------------------------------------------------------------------------------
// backward.  ds (gradient w.r.t. the block output, [C][T][vi]) is in `D` (LDS) and is consumed in place.  If `dxs`
// != nullptr the gradient w.r.t. the block input is written there ([CIN][T][vi], LDS; may alias D) -- needed for
// stacked blocks; `dxg` is the optional global dx.  Small-parameter gradients go to `gsm`: ACCUM = true adds (an
// LDS accumulator that lives for the whole launch), false stores (a per-scene row in HBM; every entry of the block's
// parameters is written exactly once per scene).
// ------------------------------------------------------------------------------------------
template <int CIN, int WAVES, bool ACCUM, typename Args>
__device__ __forceinline__ void stgcn_block_bwd(const Args &a, const float *P_, const BlockLayout &b, int n, int vi,
                                float *D, float *H1,
                                float *DH2, float *DB1, float *red, float *tot, float *gsm, const float *wsn,
                                const float *xin_ws /* block input saved by the previous block, or null */,
                                float *dxs, float *dxg, const float *lds_saved /* staged [ax|cs|g|h2] or null */,
                                const ptab_t *qtab = nullptr) {
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT, TP = T + 2;
    using S = Scope<WAVES>;
    constexpr int NT = S::NT;
    const int tid = S::tid(), V = a.V, cnt = T * vi;
    const bool train = a.lay.bn_mode == 1;
    const float inv_cnt = 1.0f / (float)cnt;
    auto put = [&](int idx, float v) {
        if (ACCUM) gsm[idx] += v;
        else gsm[idx] = v;
    };
    const float *wsa = wsn + a.lay.ws_hdr_floats;     // saved arrays sit behind the header
    const float *w_ax = wsa + (int64_t)b.ws_ax * V, *w_cs = wsa + (int64_t)b.ws_cs * V;
    const float *w_g = wsa + (int64_t)b.ws_g * V, *w_h2 = wsa + (int64_t)b.ws_h2 * V;
    if (lds_saved) {      // the kernel staged the four arrays into LDS with one DMA burst (compact, 4-float padded)
        const int n_ax = (CIN * T * vi + 3) & ~3, n_cs = (T * vi + 3) & ~3, n_g = (C * T * vi + 3) & ~3;
        w_ax = lds_saved;
        w_cs = w_ax + n_ax;
        w_g = w_cs + n_cs;
        w_h2 = w_g + n_g;
    }
    const float *hdr = wsn + b.ws_hdr;
    float m1[C], r1[C], m2[C], r2[C], mr[C], rr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        m1[c] = hdr[c]; r1[c] = hdr[C + c]; m2[c] = hdr[2 * C + c];
        r2[c] = hdr[3 * C + c]; mr[c] = hdr[4 * C + c]; rr[c] = hdr[5 * C + c];
    }
    const float *xn = a.x + n * a.x_sn;
    auto load_x = [&](int ci, int t, int w) -> float {
        return xin_ws ? xin_ws[(ci * T + t) * vi + w] : xn[ci * a.x_sc + t * a.x_st + w * a.x_sv];
    };

    // ---- B1: du = ds * prelu'(u); BatchNorm tcn.3 / residual.1 reductions; h1 = prelu(bn1(g)) ----
    {
        float s[3 * C + 1];
#pragma unroll
        for (int k = 0; k < 3 * C + 1; ++k) s[k] = 0.f;
        const float ao = P_[b.prelu_o], a1 = P_[b.prelu1];
        // loop-invariant parameters into registers once per pass (the compiler cannot hoist them itself:
        // params may alias the kernel's stores)
        float g2[C], b2[C], gr[C], br[C], g1[C], b1p[C], rb[C], rw[C * CIN];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            g2[c] = P_[b.bn2_g + c]; b2[c] = P_[b.bn2_b + c]; g1[c] = P_[b.bn1_g + c]; b1p[c] = P_[b.bn1_b + c];
            gr[c] = b.residual == 2 ? P_[b.bnr_g + c] : 0.f;
            br[c] = b.residual == 2 ? P_[b.bnr_b + c] : 0.f;
            rb[c] = b.residual == 2 ? P_[b.res_b + c] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rw[c * CIN + ci] = b.residual == 2 ? P_[b.res_w + c * CIN + ci] : 0.f;
        }
        // zero rows of the t-padded h1 plane
        for (int e = tid; e < C * vi; e += NT) {
            const int c = e / vi, w = e - c * vi;
            H1[(c * TP) * vi + w] = 0.f;
            H1[(c * TP + T + 1) * vi + w] = 0.f;
        }
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float xv[CIN];
            if (b.residual != 0) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) xv[ci] = load_x(ci, t, w);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float x2 = (w_h2[i] - m2[c]) * r2[c];
                float u = fmaf(x2, g2[c], b2[c]);
                float xr = 0.f;
                if (b.residual == 2) {
                    float r = rb[c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) r = fmaf(rw[c * CIN + ci], xv[ci], r);
                    xr = (r - mr[c]) * rr[c];
                    u += fmaf(xr, gr[c], br[c]);
                } else if (b.residual == 1) {
                    if (CIN == C) u += xv[c % CIN];
                }
                const float ds = D[i];
                float du = ds;
                if (!a.lay.use_mdn && !(u > 0.f)) {
                    du = ao * ds;
                    s[3 * C] = fmaf(ds, u, s[3 * C]);
                }
                D[i] = du;
                s[c] += du;
                s[C + c] = fmaf(du, x2, s[C + c]);
                s[2 * C + c] = fmaf(du, xr, s[2 * C + c]);
                // h1 for the temporal-conv weight gradient
                const float b1 = fmaf((w_g[i] - m1[c]) * r1[c], g1[c], b1p[c]);
                H1[(c * TP + t + 1) * vi + w] = b1 > 0.f ? b1 : a1 * b1;
            }
        }
        block_reduce<3 * C + 1, WAVES>(s, red, tot);
        for (int k = tid; k < 3 * C + 1; k += NT) {
            const float v = tot[k];
            if (k < C) {
                put(b.bn2_b + k, v);
                if (b.residual == 2) put(b.bnr_b + k, v);
            } else if (k < 2 * C) {
                put(b.bn2_g + k - C, v);
            } else if (k < 3 * C) {
                if (b.residual == 2) put(b.bnr_g + k - 2 * C, v);
            } else {
                put(b.prelu_o, v);
            }
        }
    }
    float mdu[C], mdx2[C], mdxr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        mdu[c] = train ? tot[c] * inv_cnt : 0.f;
        mdx2[c] = train ? tot[C + c] * inv_cnt : 0.f;
        mdxr[c] = train ? tot[2 * C + c] * inv_cnt : 0.f;
    }
    // ---- B2: dh2, dr; residual 1x1 conv gradients ------------------------------------------------
    {
        constexpr int K2 = C * CIN + C;
        float s[K2];
#pragma unroll
        for (int k = 0; k < K2; ++k) s[k] = 0.f;
        float g2[C], gr[C], rb[C], rw[C * CIN];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            g2[c] = P_[b.bn2_g + c];
            gr[c] = b.residual == 2 ? P_[b.bnr_g + c] : 0.f;
            rb[c] = b.residual == 2 ? P_[b.res_b + c] : 0.f;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) rw[c * CIN + ci] = b.residual == 2 ? P_[b.res_w + c * CIN + ci] : 0.f;
        }
        S::sync();       // (wave mode: the totals above were read from LDS before the next reduction overwrites them)
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float xv[CIN];
            if (b.residual == 2) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) xv[ci] = load_x(ci, t, w);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float du = D[i];
                const float x2 = (w_h2[i] - m2[c]) * r2[c];
                DH2[(c * TP + t + 1) * vi + w] = g2[c] * r2[c] * (du - mdu[c] - x2 * mdx2[c]);
                if (b.residual == 2) {
                    float r = rb[c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) r = fmaf(rw[c * CIN + ci], xv[ci], r);
                    const float xr = (r - mr[c]) * rr[c];
                    const float dr = gr[c] * rr[c] * (du - mdu[c] - xr * mdxr[c]);
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) s[c * CIN + ci] = fmaf(dr, xv[ci], s[c * CIN + ci]);
                    s[C * CIN + c] += dr;
                }
            }
        }
        for (int e = tid; e < C * vi; e += NT) {
            const int c = e / vi, w = e - c * vi;
            DH2[(c * TP) * vi + w] = 0.f;
            DH2[(c * TP + T + 1) * vi + w] = 0.f;
        }
        if (b.residual == 2) {
            block_reduce<K2, WAVES>(s, red, tot);
            for (int k = tid; k < K2; k += NT) {
                if (k < C * CIN) put(b.res_w + k, tot[k]);
                else put(b.res_b + k - C * CIN, tot[k]);
            }
        } else {
            S::sync();
        }
    }
    // ---- B3a: temporal conv weight gradients, one temporal tap at a time (25 accumulators, not 75) -------
    for (int dt = 0; dt < KT; ++dt) {
        float s[C * C];
#pragma unroll
        for (int k = 0; k < C * C; ++k) s[k] = 0.f;
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float dh[C];
#pragma unroll
            for (int c = 0; c < C; ++c) dh[c] = DH2[(c * TP + t + 1) * vi + w];
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                // h1 at t + dt - 1 (t-padded plane, rows 0 and T+1 are zero)
                const float hv = H1[(ci * TP + t + dt) * vi + w];
#pragma unroll
                for (int c = 0; c < C; ++c) s[c * C + ci] = fmaf(dh[c], hv, s[c * C + ci]);
            }
        }
        block_reduce<C * C, WAVES>(s, red, tot);
        for (int k = tid; k < C * C; k += NT) put(b.tcn_w + k * KT + dt, tot[k]);
        if (WAVES == 0) S::sync();
    }
    // ---- B3b: dh1 -> db1, conv bias gradient, BatchNorm tcn.0 reductions, PReLU slope ---------------------
    {
        constexpr int K3 = 3 * C + 1;                  // conv bias, sum db1, sum db1*xhat1, prelu slope
        float s[K3];
#pragma unroll
        for (int k = 0; k < K3; ++k) s[k] = 0.f;
        const float a1 = P_[b.prelu1];
        float tw[C * C * KT], g1[C], b1p[C];
#pragma unroll
        for (int k = 0; k < C * C * KT; ++k) tw[k] = P_[b.tcn_w + k];
#pragma unroll
        for (int c = 0; c < C; ++c) { g1[c] = P_[b.bn1_g + c]; b1p[c] = P_[b.bn1_b + c]; }
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float dh1[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                s[c] += DH2[(c * TP + t + 1) * vi + w];
                dh1[c] = 0.f;
            }
            // input gradient: dh1[ci][t] = sum_{c,dt} Wt[c][ci][dt] dh2[c][t - dt + 1]
#pragma unroll
            for (int dt = 0; dt < KT; ++dt)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float dv = DH2[(c * TP + t - dt + 2) * vi + w];
#pragma unroll
                    for (int ci = 0; ci < C; ++ci) dh1[ci] = fmaf(tw[(c * C + ci) * KT + dt], dv, dh1[ci]);
                }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float x1 = (w_g[i] - m1[c]) * r1[c];
                const float b1 = fmaf(x1, g1[c], b1p[c]);
                float db = dh1[c];
                if (!(b1 > 0.f)) {
                    db = a1 * dh1[c];
                    s[3 * C] = fmaf(dh1[c], b1, s[3 * C]);
                }
                DB1[i] = db;
                s[C + c] += db;
                s[2 * C + c] = fmaf(db, x1, s[2 * C + c]);
            }
        }
        block_reduce<K3, WAVES>(s, red, tot);
        for (int k = tid; k < K3; k += NT) {
            const float v = tot[k];
            if (k < C) put(b.tcn_b + k, v);
            else if (k < 2 * C) put(b.bn1_b + k - C, v);
            else if (k < 3 * C) put(b.bn1_g + k - 2 * C, v);
            else put(b.prelu1, v);
        }
    }
    float mdb[C], mdbx[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        mdb[c] = train ? tot[C + c] * inv_cnt : 0.f;
        mdbx[c] = train ? tot[2 * C + c] * inv_cnt : 0.f;
    }
    // ---- B4: dg; gcn 1x1 conv gradients; d(aggregated input) -------------------------------------
    {
        constexpr int K4 = C * CIN + C;
        float s[K4];
#pragma unroll
        for (int k = 0; k < K4; ++k) s[k] = 0.f;
        const bool want_dx = dxs != nullptr || dxg != nullptr;
        float g1[C], gw[C * CIN];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            g1[c] = P_[b.bn1_g + c];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) gw[c * CIN + ci] = P_[b.gcn_w + c * CIN + ci];
        }
        S::sync();       // (wave mode: mdb / mdbx were read from LDS before the next reduction overwrites them)
        for (int q = tid; q < cnt; q += NT) {
            int t, w;
            col_of(qtab, vi, q, t, w);
            float axv[CIN], dax[CIN];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) {
                axv[ci] = w_ax[(ci * T + t) * vi + w];
                dax[ci] = 0.f;
            }
            const float csum = w_cs[q];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float x1 = (w_g[i] - m1[c]) * r1[c];
                const float dg = g1[c] * r1[c] * (DB1[i] - mdb[c] - x1 * mdbx[c]);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    s[c * CIN + ci] = fmaf(dg, axv[ci], s[c * CIN + ci]);
                    dax[ci] = fmaf(gw[c * CIN + ci], dg, dax[ci]);
                }
                s[C * CIN + c] = fmaf(dg, csum, s[C * CIN + c]);
            }
            if (want_dx) {
                // stash d(ax) in the (now free) h1 plane: [CIN][T][vi]
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) H1[(ci * T + t) * vi + w] = dax[ci];
            }
        }
        block_reduce<K4, WAVES>(s, red, tot);
        for (int k = tid; k < K4; k += NT) {
            if (k < C * CIN) put(b.gcn_w + k, tot[k]);
            else put(b.gcn_b + k - C * CIN, tot[k]);
        }
        if (want_dx) {
            // ---- B5: dx[ci][t][v] = sum_w dax[ci][t][w] A[t][v][w] + residual path -----------------
            const float *an = a.adj + n * a.a_sn;
            for (int q = tid; q < cnt; q += NT) {
                int t, v;
                col_of(qtab, vi, q, t, v);
                float acc[CIN];
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
                const float *arow = an + ((int64_t)t * V + v) * V;
                for (int w = 0; w < vi; ++w) {
                    const float av = arow[w];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(H1[(ci * T + t) * vi + w], av, acc[ci]);
                }
                if (b.residual == 2) {
                    float xv[CIN];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) xv[ci] = load_x(ci, t, v);
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int i = (c * T + t) * vi + v;
                        float r = P_[b.res_b + c];
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) r = fmaf(P_[b.res_w + c * CIN + ci], xv[ci], r);
                        const float xr = (r - mr[c]) * rr[c];
                        const float dr = P_[b.bnr_g + c] * rr[c] * (D[i] - mdu[c] - xr * mdxr[c]);
#pragma unroll
                        for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(P_[b.res_w + c * CIN + ci], dr, acc[ci]);
                    }
                } else if (b.residual == 1) {
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
                        if (CIN == C) acc[ci] += D[(ci * T + t) * vi + v];
                }
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    if (dxg) dxg[(int64_t)(ci * T + t) * V + v] = acc[ci];
                    if (dxs) DB1[(ci * T + t) * vi + v] = acc[ci];      // staged; copied to D after the barrier
                }
            }
            S::sync();
            if (dxs)
                for (int e = tid; e < CIN * cnt; e += NT) dxs[e] = DB1[e];
            S::sync();
        }
    }
}

}  // namespace stg
