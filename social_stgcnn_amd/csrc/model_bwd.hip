// model_bwd: backward of social_stgcnn.forward (model.py:182-198) / st_gcn.forward (model.py:145-155)
// as ONE scene-resident kernel with persistent workgroups.
//
// A workgroup walks scenes n = blockIdx.x, +gridDim.x, ...; its parameter-gradient accumulator
// (all 7,563 floats) lives in LDS for the whole launch and leaves once as a slab; a second tiny
// kernel sums the slabs (no global float atomics; the TXP weight-gradient tiles of the waves of a
// workgroup meet in LDS through ds_add_f32, so the last bits depend on arrival order).
//
// Per scene, from the activations the forward saved (a_l, z_l of the TXP-CNN; ax, colsum, g, h2 and
// the BatchNorm statistics of each st_gcn block):
//   TXP-CNN, output conv first then hidden layers in reverse:
//     dz      = d(out) * prelu'(z)                       (VALU; also the PReLU slope gradient)
//     dW, db += dz (x) im2col(a_l)                       MFMA 16x16x4 f32: M = 12 out-channels,
//                                                        N = (tap, in-channel) columns + a ones column
//                                                        (bias gradient), K = scene positions
//     d(a_l)  = conv_transpose(dz, W) [+ d(out)]         MFMA, same implicit GEMM as the forward with
//                                                        flipped taps and W^T
//   st_gcn block: BatchNorm backward with PER-SCENE statistics, PReLU, temporal conv, 1x1 convs --
//   VALU with wave-shuffle + LDS block reductions.
#include "model_common.hpp"

namespace stg {

struct BwdArgs {
    ModelLayout lay;
    const float *params, *buffers, *x;
    int64_t x_sn, x_sc, x_st, x_sv;
    const float *adj;
    int64_t a_sn;
    const int32_t *num_peds;
    int N, V;
    const float *dy, *ws;
    int64_t ws_stride;
    float *slabs, *dx;
    int debug_skip;   // timing-only diagnostic (STG_DEBUG_SKIP): 1 wgrad, 2 dgrad, 4 st_gcn -- wrong results
};

constexpr int kRedMax = 96;   // widest block reduction (values)

// Sum K per-thread values over the workgroup; totals land in tot[0..K) (LDS), visible to every
// thread on return.
template <int K, int WAVES>
__device__ __forceinline__ void block_reduce(float (&v)[K], float *red, float *tot) {
    static_assert(K <= kRedMax, "reduction too wide");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) red[wave * K + k] = v[k];
    }
    __syncthreads();
    for (int k = tid; k < K; k += WAVES * 64) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[w * K + k];
        tot[k] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// TXP-CNN backward pieces
// ------------------------------------------------------------------------------------------
// dgrad weights: A operand lane (i = ci, kq) of K-step (tap', j) holds W[co = 4j+kq][ci][8 - tap'].
template <int CINL>
__device__ __forceinline__ void txp_load_weights_t(const float *__restrict__ W, float (&wreg)[27]) {
    const int lane = threadIdx.x & 63, ci = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int co = 4 * j + kq;
            wreg[tap * 3 + j] = ci < CINL ? W[(co * CINL + ci) * 9 + (8 - tap)] : 0.f;
        }
}

// d(a_l)[ci][h][w] = sum_{co,kh,kw} W[co][ci][kh][kw] dz[co][h-kh+1][w-kw+1]  (+ dcur if accumulate)
template <int CINL, int WAVES>
__device__ void txp_dgrad(const float *__restrict__ W, const float *dzb, float *dcur, int vi, bool accumulate) {
    constexpr int C = Cfg::C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = lane & 15, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi);
    const int npos = C * vi, ntiles = (npos + 15) >> 4;
    float wreg[27];
    txp_load_weights_t<CINL>(W, wreg);
    for (int tile0 = wave * 2; tile0 < ntiles; tile0 += WAVES * 2) {
        int hh[2], ww[2], base[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = (tile0 + u) * 16 + nq;
            ok[u] = pos < npos;
            const int pc = ok[u] ? pos : 0;
            hh[u] = pc / vi;
            ww[u] = pc - hh[u] * vi;
            base[u] = kq * SC + hh[u] * SW + ww[u];
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * SW + (tap % 3);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float b0 = dzb[base[0] + 4 * j * SC + toff];
                const float b1 = dzb[base[1] + 4 * j * SC + toff];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * 3 + j], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * 3 + j], b1, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u]) continue;
            const f32x4 acc = u == 0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = 4 * kq + r;
                if (ci < CINL) {
                    const int i = (ci * C + hh[u]) * vi + ww[u];
                    dcur[i] = accumulate ? dcur[i] + acc[r] : acc[r];
                }
            }
        }
    }
}

// dW[co][ci][tap] += sum_pos dz[co][pos] a[ci][pos + tap], db[co] += sum_pos dz[co][pos]
// (K = positions, split over the waves; partial tiles are added into the LDS accumulator gsm).
template <int CINL, int WAVES>
__device__ void txp_wgrad(const float *dzb, const float *ain, const int *pt, float *gsm, int w_off, int b_off,
                          int vi) {
    constexpr int C = Cfg::C, P = Cfg::P;
    constexpr int NCOL = 9 * CINL + 1, NTILE = (NCOL + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = lane & 15, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi);
    const int npos = C * vi, nsteps = (npos + 3) >> 2;
    const int co_a = nq < P ? nq : P - 1;     // rows 12..15 of the tile are never read back
    int boff[NTILE];
    bool bone[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) {
        int col = tl * 16 + nq;
        bone[tl] = col == NCOL - 1;
        if (col > NCOL - 2) col = NCOL - 2;
        const int tap = col / CINL, ci = col - tap * CINL;
        boff[tl] = ci * SC + (tap / 3 - 1) * SW + (tap % 3 - 1);
    }
    f32x4 acc[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = wave; s < nsteps; s += WAVES) {
        const int p = 4 * s + kq;
        const bool ok = p < npos;
        const int off = pt[ok ? p : 0];
        const float av = ok ? dzb[co_a * SC + off] : 0.f;
        const int offb = ok ? off : SW + 1;
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) {
            const float bv = bone[tl] ? 1.f : ain[boff[tl] + offb];
            acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[tl], 0, 0, 0);
        }
    }
    if (kq < 3) {
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) {
            const int col = tl * 16 + nq;
            if (col < NCOL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * kq + r;
                    int idx;
                    if (col == NCOL - 1) {
                        idx = b_off + co;
                    } else {
                        const int tap = col / CINL, ci = col - tap * CINL;
                        idx = w_off + (co * CINL + ci) * 9 + tap;
                    }
                    atomicAdd(&gsm[idx], acc[tl][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// st_gcn block backward.  ds (gradient w.r.t. the block output, [C][T][vi]) is in `D` (LDS) and is
// consumed in place.  If `dxs` != nullptr the gradient w.r.t. the block input is written there
// ([CIN][T][vi], LDS; may alias D) -- needed for stacked blocks; `dxg` is the optional global dx.
// ------------------------------------------------------------------------------------------
template <int CIN, int WAVES>
__device__ void stgcn_block_bwd(const BwdArgs &a, const BlockLayout &b, int n, int vi, float *D, float *H1,
                                float *DH2, float *DB1, float *red, float *tot, float *gsm, const float *wsn,
                                const float *xin_ws /* block input saved by the previous block, or null */,
                                float *dxs, float *dxg) {
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT, NT = WAVES * 64, TP = T + 2;
    const int tid = threadIdx.x, V = a.V, cnt = T * vi;
    const float *P_ = a.params;
    const bool train = a.lay.bn_mode == 1;
    const float inv_cnt = 1.0f / (float)cnt;
    const float *wsa = wsn + a.lay.ws_hdr_floats;     // saved arrays sit behind the header
    const float *w_ax = wsa + (int64_t)b.ws_ax * V, *w_cs = wsa + (int64_t)b.ws_cs * V;
    const float *w_g = wsa + (int64_t)b.ws_g * V, *w_h2 = wsa + (int64_t)b.ws_h2 * V;
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
        // zero rows of the t-padded h1 plane
        for (int e = tid; e < C * vi; e += NT) {
            const int c = e / vi, w = e - c * vi;
            H1[(c * TP) * vi + w] = 0.f;
            H1[(c * TP + T + 1) * vi + w] = 0.f;
        }
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
            float xv[CIN];
            if (b.residual != 0) {
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) xv[ci] = load_x(ci, t, w);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float x2 = (w_h2[i] - m2[c]) * r2[c];
                float u = fmaf(x2, P_[b.bn2_g + c], P_[b.bn2_b + c]);
                float xr = 0.f;
                if (b.residual == 2) {
                    float r = P_[b.res_b + c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) r = fmaf(P_[b.res_w + c * CIN + ci], xv[ci], r);
                    xr = (r - mr[c]) * rr[c];
                    u += fmaf(xr, P_[b.bnr_g + c], P_[b.bnr_b + c]);
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
                const float b1 = fmaf((w_g[i] - m1[c]) * r1[c], P_[b.bn1_g + c], P_[b.bn1_b + c]);
                H1[(c * TP + t + 1) * vi + w] = b1 > 0.f ? b1 : a1 * b1;
            }
        }
        block_reduce<3 * C + 1, WAVES>(s, red, tot);
        for (int k = tid; k < 3 * C + 1; k += NT) {
            const float v = tot[k];
            if (k < C) {
                gsm[b.bn2_b + k] += v;
                if (b.residual == 2) gsm[b.bnr_b + k] += v;
            } else if (k < 2 * C) {
                gsm[b.bn2_g + k - C] += v;
            } else if (k < 3 * C) {
                if (b.residual == 2) gsm[b.bnr_g + k - 2 * C] += v;
            } else {
                gsm[b.prelu_o] += v;
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
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
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
                DH2[(c * TP + t + 1) * vi + w] = P_[b.bn2_g + c] * r2[c] * (du - mdu[c] - x2 * mdx2[c]);
                if (b.residual == 2) {
                    float r = P_[b.res_b + c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci) r = fmaf(P_[b.res_w + c * CIN + ci], xv[ci], r);
                    const float xr = (r - mr[c]) * rr[c];
                    const float dr = P_[b.bnr_g + c] * rr[c] * (du - mdu[c] - xr * mdxr[c]);
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
                if (k < C * CIN) gsm[b.res_w + k] += tot[k];
                else gsm[b.res_b + k - C * CIN] += tot[k];
            }
        } else {
            __syncthreads();
        }
    }
    // ---- B3: temporal conv gradients, dh1 -> db1, BatchNorm tcn.0 reductions ------------------------
    {
        constexpr int KW = C * C * KT;                 // 75 weight gradients
        constexpr int K3 = KW + C + 2 * C + 1;         // + conv bias, sum db1, sum db1*xhat1, prelu slope
        float s[K3];
#pragma unroll
        for (int k = 0; k < K3; ++k) s[k] = 0.f;
        const float a1 = P_[b.prelu1];
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
            float dh[C], dh1[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                dh[c] = DH2[(c * TP + t + 1) * vi + w];
                s[KW + c] += dh[c];
                dh1[c] = 0.f;
            }
#pragma unroll
            for (int dt = 0; dt < KT; ++dt) {
#pragma unroll
                for (int ci = 0; ci < C; ++ci) {
                    // weight gradient: h1 at t + dt - 1 (t-padded plane, rows 0 and T+1 are zero)
                    const float hv = H1[(ci * TP + t + dt) * vi + w];
#pragma unroll
                    for (int c = 0; c < C; ++c) s[(c * C + ci) * KT + dt] = fmaf(dh[c], hv, s[(c * C + ci) * KT + dt]);
                }
                // input gradient: dh1[ci][t] = sum_{c,dt} Wt[c][ci][dt] dh2[c][t - dt + 1]
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float dv = DH2[(c * TP + t - dt + 2) * vi + w];
#pragma unroll
                    for (int ci = 0; ci < C; ++ci) dh1[ci] = fmaf(P_[b.tcn_w + (c * C + ci) * KT + dt], dv, dh1[ci]);
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float x1 = (w_g[i] - m1[c]) * r1[c];
                const float b1 = fmaf(x1, P_[b.bn1_g + c], P_[b.bn1_b + c]);
                float db = dh1[c];
                if (!(b1 > 0.f)) {
                    db = a1 * dh1[c];
                    s[KW + 3 * C] = fmaf(dh1[c], b1, s[KW + 3 * C]);
                }
                DB1[i] = db;
                s[KW + C + c] += db;
                s[KW + 2 * C + c] = fmaf(db, x1, s[KW + 2 * C + c]);
            }
        }
        block_reduce<K3, WAVES>(s, red, tot);
        for (int k = tid; k < K3; k += NT) {
            const float v = tot[k];
            if (k < KW) gsm[b.tcn_w + k] += v;
            else if (k < KW + C) gsm[b.tcn_b + k - KW] += v;
            else if (k < KW + 2 * C) gsm[b.bn1_b + k - KW - C] += v;
            else if (k < KW + 3 * C) gsm[b.bn1_g + k - KW - 2 * C] += v;
            else gsm[b.prelu1] += v;
        }
    }
    float mdb[C], mdbx[C];
    {
        constexpr int KW = C * C * KT;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            mdb[c] = train ? tot[KW + C + c] * inv_cnt : 0.f;
            mdbx[c] = train ? tot[KW + 2 * C + c] * inv_cnt : 0.f;
        }
    }
    // ---- B4: dg; gcn 1x1 conv gradients; d(aggregated input) -------------------------------------
    {
        constexpr int K4 = C * CIN + C;
        float s[K4];
#pragma unroll
        for (int k = 0; k < K4; ++k) s[k] = 0.f;
        const bool want_dx = dxs != nullptr || dxg != nullptr;
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
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
                const float dg = P_[b.bn1_g + c] * r1[c] * (DB1[i] - mdb[c] - x1 * mdbx[c]);
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) {
                    s[c * CIN + ci] = fmaf(dg, axv[ci], s[c * CIN + ci]);
                    dax[ci] = fmaf(P_[b.gcn_w + c * CIN + ci], dg, dax[ci]);
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
            if (k < C * CIN) gsm[b.gcn_w + k] += tot[k];
            else gsm[b.gcn_b + k - C * CIN] += tot[k];
        }
        if (want_dx) {
            // ---- B5: dx[ci][t][v] = sum_w dax[ci][t][w] A[t][v][w] + residual path -----------------
            const float *an = a.adj + n * a.a_sn;
            for (int q = tid; q < cnt; q += NT) {
                const int t = q / vi, v = q - t * vi;
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
            __syncthreads();
            if (dxs)
                for (int e = tid; e < CIN * cnt; e += NT) dxs[e] = DB1[e];
            __syncthreads();
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void model_bwd_kernel(const BwdArgs a) {
    constexpr int C = Cfg::C, T = Cfg::T, P = Cfg::P, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int V = a.V, tid = threadIdx.x;
    const ModelLayout &L = a.lay;
    const int scmax = txp_sc(V);
    const int plane_floats = P * scmax;
    // st_gcn phase needs 3 planes: h1 [C][T+2][V], dh2 [C][T+2][V], db1 [C][T][V]
    const int st_floats = (2 * C * (T + 2) + C * T) * V;
    const int two_planes = 2 * plane_floats > st_floats ? 2 * plane_floats : st_floats;
    float *gsm = sm;                                  // [n_params]
    float *ain = gsm + ((L.n_params + 3) & ~3);       // [P][SC]   a_l, zero-bordered
    float *dzb = ain + plane_floats;                  // [P][SC]   dz_l, zero-bordered
    float *dcur = ain + two_planes;                   // [P*C*V]   gradient w.r.t. the layer output
    float *red = dcur + P * C * V;                    // [WAVES*kRedMax]
    float *tot = red + WAVES * kRedMax;               // [kRedMax]
    int *pt = reinterpret_cast<int *>(tot + kRedMax); // [C*V] position -> plane offset
    const float *Pm = a.params;

    for (int e = tid; e < L.n_params; e += NT) gsm[e] = 0.f;

    for (int n = blockIdx.x; n < a.N; n += gridDim.x) {
        int vi = a.num_peds ? a.num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        float *dxn = a.dx ? a.dx + (int64_t)n * L.blk[0].cin * T * V : nullptr;
        if (dxn && vi < V)
            for (int e = tid; e < L.blk[0].cin * T * (V - vi); e += NT) {
                const int r = e / (V - vi), w = vi + (e - r * (V - vi));
                dxn[(int64_t)r * V + w] = 0.f;
            }
        if (vi == 0) continue;
        const float *wsn = a.ws + n * a.ws_stride;
        const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi;
        const int out_rows = L.n_txp > 0 ? C * P : C * T;
        const float *dyn = a.dy + (int64_t)n * out_rows * V;
        __syncthreads();
        if (L.n_txp > 0) {
            // ---- TXP-CNN backward ----------------------------------------------------------------
            for (int e = tid; e < P * SC; e += NT) {
                ain[e] = 0.f;
                dzb[e] = 0.f;
            }
            for (int e = tid; e < npos; e += NT) {
                const int h = e / vi, w = e - h * vi;
                pt[e] = (h + 1) * SW + (w + 1);
            }
            __syncthreads();
            for (int l = L.L; l >= 0; --l) {
                const bool is_out = l == L.L;
                const int cin_l = l == 0 ? T : P;
                // stage a_l (zero-bordered) and dz_l
                const float *al = wsn + L.ws_hdr_floats + (int64_t)L.ws_a[l] * V;
                for (int e = tid; e < cin_l * npos; e += NT) {
                    const int ch = e / npos, p = e - ch * npos;
                    ain[ch * SC + pt[p]] = al[e];
                }
                if (is_out) {
                    for (int e = tid; e < P * npos; e += NT) {
                        const int ch = e / npos, p = e - ch * npos, h = p / vi, w = p - h * vi;
                        dzb[ch * SC + pt[p]] = dyn[(int64_t)(ch * C + h) * V + w];
                    }
                    __syncthreads();
                } else {
                    const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
                    const float alpha = Pm[L.prelus + l];
                    float s[1] = {0.f};
                    for (int e = tid; e < P * npos; e += NT) {
                        const int ch = e / npos, p = e - ch * npos;
                        const float z = zl[e], d = dcur[e];
                        float dz = d;
                        if (!(z > 0.f)) {
                            dz = alpha * d;
                            s[0] = fmaf(d, z, s[0]);
                        }
                        dzb[ch * SC + pt[p]] = dz;
                    }
                    block_reduce<1, WAVES>(s, red, tot);
                    if (tid == 0) gsm[L.prelus + l] += tot[0];
                }
                const int w_off = is_out ? L.out_w : L.txp_w[l];
                const int b_off = is_out ? L.out_b : L.txp_b[l];
                if (l == 0) {
                    if (!(a.debug_skip & 1)) txp_wgrad<Cfg::T, WAVES>(dzb, ain, pt, gsm, w_off, b_off, vi);
                    if (!(a.debug_skip & 2)) txp_dgrad<Cfg::T, WAVES>(Pm + w_off, dzb, dcur, vi, false);
                } else {
                    if (!(a.debug_skip & 1)) txp_wgrad<Cfg::P, WAVES>(dzb, ain, pt, gsm, w_off, b_off, vi);
                    if (!(a.debug_skip & 2)) txp_dgrad<Cfg::P, WAVES>(Pm + w_off, dzb, dcur, vi, !is_out);
                }
                __syncthreads();
            }
        } else {
            for (int e = tid; e < C * T * vi; e += NT) {
                const int r = e / vi, w = e - r * vi;
                dcur[e] = dyn[(int64_t)r * V + w];
            }
            __syncthreads();
        }
        // ---- st_gcn blocks, last to first ------------------------------------------------------
        float *H1 = ain, *DH2 = ain + C * (T + 2) * V, *DB1 = DH2 + C * (T + 2) * V;
        for (int j = L.n_blocks - 1; j >= 0 && !(a.debug_skip & 4); --j) {
            const float *xin = j > 0 ? wsn + L.ws_hdr_floats + (int64_t)L.blk[j - 1].ws_s * V : nullptr;
            float *dxs = j > 0 ? dcur : nullptr;
            float *dxg = j == 0 ? dxn : nullptr;
            if (L.blk[j].cin == Cfg::CIN0)
                stgcn_block_bwd<Cfg::CIN0, WAVES>(a, L.blk[j], n, vi, dcur, H1, DH2, DB1, red, tot, gsm, wsn, xin,
                                                  dxs, dxg);
            else
                stgcn_block_bwd<Cfg::C, WAVES>(a, L.blk[j], n, vi, dcur, H1, DH2, DB1, red, tot, gsm, wsn, xin, dxs,
                                               dxg);
        }
    }
    __syncthreads();
    float *slab = a.slabs + (int64_t)blockIdx.x * L.n_params;
    for (int e = tid; e < L.n_params; e += NT) slab[e] = gsm[e];
}

// grad[p] = sum over slabs.  One workgroup owns 32 consecutive parameters (one 128-byte line per slab
// row); its 8 lane-groups stride over the slabs, partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float *__restrict__ slabs, int n_slabs, int n_params,
                                                           float *__restrict__ grad) {
    __shared__ float part[8][32];
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + col;
    float s = 0.f;
    if (p < n_params)
        for (int k = grp; k < n_slabs; k += 8) s += slabs[(int64_t)k * n_params + p];
    part[grp][col] = s;
    __syncthreads();
    if (grp == 0 && p < n_params) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) t += part[g][col];
        grad[p] = t;
    }
}

static size_t bwd_lds_bytes(const ModelLayout &L, int V, int waves) {
    const int plane = Cfg::P * txp_sc(V);
    const int st = (2 * Cfg::C * (Cfg::T + 2) + Cfg::C * Cfg::T) * V;
    const int two = 2 * plane > st ? 2 * plane : st;
    const size_t fl = ((L.n_params + 3) & ~3) + (size_t)two + (size_t)Cfg::P * Cfg::C * V + (size_t)waves * kRedMax +
                      kRedMax + (size_t)Cfg::C * V;
    return fl * sizeof(float);
}

static int bwd_waves(int V) {
    int waves = V <= 12 ? 1 : (V <= 40 ? 2 : (V <= 80 ? 4 : 8));
    if (const char *e = getenv("STG_BWD_WAVES")) {
        const int w = atoi(e);
        if (w == 1 || w == 2 || w == 4 || w == 8) waves = w;
    }
    return waves;
}

static int bwd_grid(const ModelLayout &L, int N, int V) {
    const int waves = bwd_waves(V);
    const size_t lds = bwd_lds_bytes(L, V, waves);
    if (lds > (size_t)kLdsBytes) return -1;
    int per_cu = (int)(kLdsBytes / lds);
    const int by_waves = 16 / waves;          // keep <= 16 waves per CU resident
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    int grid = kNumCU * per_cu;
    if (const char *e = getenv("STG_BWD_GRID")) {
        const int g = atoi(e);
        if (g > 0) grid = g;
    }
    return grid < N ? grid : N;
}

}  // namespace stg

extern "C" {

int64_t stg_model_bwd_slabs(const stg_model_desc *d, int N, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (N <= 0 || V <= 0) return stg::fail(STG_EINVAL, "stg_model_bwd_slabs: N=%d V=%d", N, V);
    const int g = stg::bwd_grid(l, N, V);
    if (g < 0) return stg::fail(STG_ELDS, "stg_model_bwd: V=%d does not fit the LDS of one CU", V);
    return g;
}

int stg_model_bwd(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                  int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                  int N, int V, const float *dy, const float *ws, float *slabs, float *grad_params, float *dx,
                  void *stream) {
    using namespace stg;
    BwdArgs a{};
    const int rc = make_layout(d, &a.lay);
    if (rc != STG_OK) return rc;
    STG_REQUIRE(params && buffers && x && adj && dy && ws && slabs && grad_params, STG_EINVAL,
                "stg_model_bwd: null pointer");
    STG_REQUIRE(N >= 0 && V > 0, STG_EINVAL, "stg_model_bwd: bad sizes N=%d V=%d", N, V);
    hipStream_t st = as_stream(stream);
    if (N == 0) {
        hipError_t e = hipMemsetAsync(grad_params, 0, sizeof(float) * a.lay.n_params, st);
        if (e != hipSuccess) return hip_fail(e, "stg_model_bwd: memset");
        return STG_OK;
    }
    a.params = params; a.buffers = buffers; a.x = x;
    a.x_sn = x_sn; a.x_sc = x_sc; a.x_st = x_st; a.x_sv = x_sv;
    a.adj = adj; a.a_sn = a_sn; a.num_peds = num_peds; a.N = N; a.V = V;
    a.dy = dy; a.ws = ws; a.ws_stride = ws_floats_per_scene(a.lay, V); a.slabs = slabs; a.dx = dx;
    if (const char *e = getenv("STG_DEBUG_SKIP")) a.debug_skip = atoi(e);
    const int waves = bwd_waves(V);
    const size_t lds = bwd_lds_bytes(a.lay, V, waves);
    STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "stg_model_bwd: V=%d needs %zu bytes of LDS (> %d)", V, lds,
                kLdsBytes);
    const int grid = bwd_grid(a.lay, N, V);
#define STG_LAUNCH_BWD(W)                                                                                    \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&model_bwd_kernel<W>),            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e_ != hipSuccess) return hip_fail(e_, "stg_model_bwd: hipFuncSetAttribute");                     \
        hipLaunchKernelGGL(model_bwd_kernel<W>, dim3(grid), dim3(W * 64), lds, st, a);                       \
    } while (0)
    switch (waves) {
        case 1: STG_LAUNCH_BWD(1); break;
        case 2: STG_LAUNCH_BWD(2); break;
        case 4: STG_LAUNCH_BWD(4); break;
        default: STG_LAUNCH_BWD(8); break;
    }
#undef STG_LAUNCH_BWD
    STG_LAUNCH_CHECK("stg_model_bwd");
    const int np = a.lay.n_params;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((np + 31) / 32), dim3(256), 0, st, slabs, grid, np, grad_params);
    STG_LAUNCH_CHECK("stg_model_bwd: reduce_slabs");
    return STG_OK;
}

}  // extern "C"
