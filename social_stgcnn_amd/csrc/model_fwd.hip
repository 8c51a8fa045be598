// model_fwd: st_gcn.forward (model.py:145-155) + social_stgcnn.forward (model.py:182-198) as ONE
// scene-resident kernel.  A workgroup of WAVES wave64 owns one scene-window at a time: the scene's
// activations never leave LDS between the adjacency-weighted aggregation, the temporal
// Conv+BN+PReLU block and the TXP-CNN; HBM sees x and A once (coalesced along w, then t) and the
// (5,12,V) output once (plus, in training, the activations the backward needs).
//
//  st_gcn block (VALU, 3% of the flops): lanes own (t,w) columns; the einsum is re-associated to
//    aggregate the Cin input channels first (g = Wg (x A) + bg colsum(A)), BatchNorm statistics are
//    PER SCENE (the reference trains with N=1, train.py:173-177) via wave-shuffle + LDS reductions.
//  TXP-CNN (97% of the flops): every 3x3 conv over the (5, V) plane is an implicit GEMM on
//    v_mfma_f32_16x16x4_f32 (exact fp32): M = 12 output channels (padded to 16), N = 16 positions
//    of the scene, K = (tap, 4 input channels).  Weights live in VGPRs (one per K-step), the im2col
//    B operand is one ds_read_b32 per MFMA from a zero-bordered LDS plane whose channel stride is
//    == 16 (mod 32) dwords (conflict-free for the four K-lane groups).
//
// On the fast path (one st_gcn block, V <= 68) this kernel runs the block only (F1) and hands a_0 to the
// wave-per-scene TXP kernel of txp_wave.hip (F2) in that kernel's in-place plane layout.
#include "model_common.hpp"
#include "txp_wave.hpp"

namespace stg {

struct FwdArgs {
    ModelLayout lay;
    const float *params, *buffers, *x;
    int64_t x_sn, x_sc, x_st, x_sv;
    const float *adj;
    int64_t a_sn;
    const int32_t *num_peds;
    int N, V;
    float *y, *ws;
    int64_t ws_stride;
    float *stats;
    const int32_t *order;   // non-null: scenes sorted by crowd size (descending); block b takes order[b]
    float *a0g;       // non-null: blocks only -- the a_0 plane goes to a0g[n] for the wave-per-scene TXP kernel
    int debug_skip;   // timing-only diagnostic (STG_DEBUG_SKIP): 16 TXP-CNN, 32 st_gcn -- wrong results
};

// floats of the kernel's TXP plane buffer: two-plane layout [P][txp_sc(V)], or the in-place hand-off layout
// [T][txp_sci(V)] of the wave path -- whichever is larger (the paddings to == 16 (mod 32) differ)
__host__ __device__ inline int fwd_plane_floats(int V) {
    const int a = Cfg::P * txp_sc(V), b = Cfg::T * txp_sci(V);
    return a > b ? a : b;
}

template <int K, int WAVES>
__device__ __forceinline__ void block_sum(float (&v)[K], float *red) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
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

// ------------------------------------------------------------------------------------------
// one st_gcn block for the scene resident in LDS
//   X [CIN][T][vi] input, G / H [C][T][vi] scratch; on return the block output s is in H (same
//   layout), and, when `to_txp`, also scattered into the zero-bordered TXP plane `plane`.
// ------------------------------------------------------------------------------------------
template <int CIN, int WAVES>
__device__ void stgcn_block_fwd(const FwdArgs &a, const float *__restrict__ P_, const float *__restrict__ B_,
                                const BlockLayout &b, int n, int vi, const float *X, float *G,
                                float *H, float *cs, float *red, float *wsn, float *statn, bool to_txp,
                                float *plane, float *yblock, bool save_s) {
    constexpr int C = Cfg::C, T = Cfg::T, KT = Cfg::KT, NT = WAVES * 64;
    const int tid = threadIdx.x, V = a.V;
    const int cnt = T * vi;
    const bool train = a.lay.bn_mode == 1;
    const float eps = a.lay.eps;
    const float *an = a.adj + n * a.a_sn;
    float *wsa = wsn ? wsn + a.lay.ws_hdr_floats : nullptr;    // saved arrays sit behind the header

    // ---- P1: aggregation of the CIN input channels + 1x1 conv (model.py:66-67) ------------------
    float s1[C];
#pragma unroll
    for (int c = 0; c < C; ++c) s1[c] = 0.f;
    for (int q = tid; q < cnt; q += NT) {
        const int t = q / vi, w = q - t * vi;
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
        cs[q] = csum;
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
    // ---- BatchNorm tcn.0 statistics (model.py:114) ------------------------------------------------
    float m1[C], r1[C];
    if (train) {
        block_sum<C, WAVES>(s1, red);
        float s2[C];
#pragma unroll
        for (int c = 0; c < C; ++c) { m1[c] = s1[c] / (float)cnt; s2[c] = 0.f; }
        __syncthreads();   // G complete
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
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
        __syncthreads();
    }
    if (wsn && tid == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            wsn[b.ws_hdr + c] = m1[c];
            wsn[b.ws_hdr + C + c] = r1[c];
        }
    }
    // ---- P3: BN + PReLU in place (tcn.0, tcn.1) --------------------------------------------------
    {
        const float al = P_[b.prelu1];
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                const float v = fmaf((G[i] - m1[c]) * r1[c], P_[b.bn1_g + c], P_[b.bn1_b + c]);
                G[i] = v > 0.f ? v : al * v;
            }
        }
    }
    __syncthreads();
    // ---- P4: temporal conv (tcn.2) + residual 1x1 conv statistics -------------------------------
    float s2r[2 * C];
#pragma unroll
    for (int c = 0; c < 2 * C; ++c) s2r[c] = 0.f;
    for (int q = tid; q < cnt; q += NT) {
        const int t = q / vi, w = q - t * vi;
        float h[C];
#pragma unroll
        for (int c = 0; c < C; ++c) h[c] = P_[b.tcn_b + c];
#pragma unroll
        for (int dt = 0; dt < KT; ++dt) {
            const int ti = t + dt - (KT - 1) / 2;
            if (ti < 0 || ti >= T) continue;
#pragma unroll
            for (int ci = 0; ci < C; ++ci) {
                const float hv = G[(ci * T + ti) * vi + w];
#pragma unroll
                for (int c = 0; c < C; ++c) h[c] = fmaf(P_[b.tcn_w + (c * C + ci) * KT + dt], hv, h[c]);
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
        __syncthreads();   // H complete
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
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
        __syncthreads();
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
    // ---- P6: BN (tcn.3) + residual + PReLU (model.py:150-153) ------------------------------------
    {
        const float ao = P_[b.prelu_o];
        // plane geometry: the wave-per-scene TXP kernel takes a_0 in its in-place layout (channel stride txp_sci,
        // `plane` already points at padded row 0 = row slot 2), the in-kernel TXP path in the two-plane layout
        const int SW = txp_sw(vi), SC = a.a0g ? txp_sci(vi) : txp_sc(vi);
        for (int q = tid; q < cnt; q += NT) {
            const int t = q / vi, w = q - t * vi;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int i = (c * T + t) * vi + w;
                float u = fmaf((H[i] - m2[c]) * r2[c], P_[b.bn2_g + c], P_[b.bn2_b + c]);
                if (b.residual == 2) {
                    float r = P_[b.res_b + c];
#pragma unroll
                    for (int ci = 0; ci < CIN; ++ci)
                        r = fmaf(P_[b.res_w + c * CIN + ci], X[(ci * T + t) * vi + w], r);
                    u += fmaf((r - mr[c]) * rr[c], P_[b.bnr_g + c], P_[b.bnr_b + c]);
                } else if (b.residual == 1) {
                    if (CIN == C) u += X[i];
                }
                const float s = (a.lay.use_mdn || u > 0.f) ? u : ao * u;
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
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// TXP-CNN layer on MFMA
// ------------------------------------------------------------------------------------------
template <int CINL>
__device__ __forceinline__ void txp_load_weights(const float *__restrict__ W, float (&wreg)[CINL * 9 / 4]) {
    const int lane = threadIdx.x & 63, co = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < CINL / 4; ++j) {
            const int ci = 4 * j + kq;
            wreg[tap * (CINL / 4) + j] = co < Cfg::P ? W[(co * CINL + ci) * 9 + tap] : 0.f;
        }
}

// kind: 0 = hidden layer without residual (layer 0), 1 = hidden layer with residual, 2 = output conv
template <int CINL, int WAVES>
__device__ void txp_layer_fwd(const float *__restrict__ W, const float *__restrict__ bias, float alpha, int kind,
                              const float *in, float *out, int vi, int V, float *zsave, float *yout) {
    constexpr int C = Cfg::C, P = Cfg::P, KS = CINL * 9 / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = lane & 15, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi);
    const int npos = C * vi, ntiles = (npos + 15) >> 4;
    float wreg[KS];
    txp_load_weights<CINL>(W, wreg);
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = (4 * kq + r) < P ? bias[4 * kq + r] : 0.f;

    for (int tile0 = wave * 2; tile0 < ntiles; tile0 += WAVES * 2) {
        int pos[2], hh[2], ww[2], base[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pos[u] = (tile0 + u) * 16 + nq;
            ok[u] = pos[u] < npos;
            const int pc = ok[u] ? pos[u] : 0;
            hh[u] = pc / vi;
            ww[u] = pc - hh[u] * vi;
            base[u] = kq * SC + hh[u] * SW + ww[u];
        }
        f32x4 acc0 = binit, acc1 = binit;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * SW + (tap % 3);
#pragma unroll
            for (int j = 0; j < CINL / 4; ++j) {
                const float b0 = in[base[0] + 4 * j * SC + toff];
                const float b1 = in[base[1] + 4 * j * SC + toff];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * (CINL / 4) + j], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * (CINL / 4) + j], b1, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u] || kq == 3) continue;
            const f32x4 acc = u == 0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * kq + r;
                const float z = acc[r];
                const int flat = (co * C + hh[u]) * vi + ww[u];
                if (kind == 2) {
                    // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
                    yout[(int64_t)(co * C + hh[u]) * V + ww[u]] = z;
                } else {
                    const int li = co * SC + (hh[u] + 1) * SW + (ww[u] + 1);
                    float av = z > 0.f ? z : alpha * z;
                    if (kind == 1) av += in[li];
                    out[li] = av;
                    if (zsave) zsave[flat] = z;
                }
            }
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void model_fwd_kernel(const FwdArgs a, const float *__restrict__ params,
                                                               const float *__restrict__ buffers) {
    // params / buffers are separate __restrict__ kernel arguments on purpose: only then can the compiler prove
    // that the kernel's own stores never clobber them and fetch the (wave-uniform) weights with scalar loads
    // into SGPRs instead of per-lane vector loads + s_waitcnt (340 vector loads per scene otherwise)
    constexpr int C = Cfg::C, T = Cfg::T, P = Cfg::P, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int V = a.V, tid = threadIdx.x;
    // blocks-only mode (wave path): the plane is just the [T][txp_sci] hand-off image and X holds c_in channels --
    // 24 KB instead of 28 KB at V = 32: six resident workgroups per CU instead of five
    const bool slim = a.a0g != nullptr;
    const int xin_floats = (slim ? a.lay.blk[0].cin : C) * T * V;
    const int plane_floats = slim ? T * txp_sci(V) : fwd_plane_floats(V);
    const int reg_floats = slim ? xin_floats + 2 * C * T * V
                                : (plane_floats > 3 * C * T * V ? plane_floats : 3 * C * T * V);
    float *bufA = sm;
    float *reg = bufA + plane_floats;
    float *cs = reg + reg_floats;
    float *red = cs + T * V;
    const ModelLayout &L = a.lay;
    const int c_in = L.blk[0].cin;
    const int out_rows = L.n_txp > 0 ? C * P : C * T;

    for (int it = blockIdx.x; it < a.N; it += gridDim.x) {
        const int n = a.order ? a.order[it] : it;       // largest crowds are dispatched first
        int vi = a.num_peds ? a.num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        float *yn = a.y + (int64_t)n * out_rows * V;
        // padded pedestrian slots of the output are zeros
        if (vi < V)
            for (int e = tid; e < out_rows * (V - vi); e += NT) {
                const int r = e / (V - vi), w = vi + (e - r * (V - vi));
                yn[(int64_t)r * V + w] = 0.f;
            }
        if (vi == 0) continue;
        float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;
        float *statn = a.stats ? a.stats + (int64_t)n * L.stat_floats : nullptr;
        const int SW = txp_sw(vi), SC = txp_sc(vi);
        __syncthreads();
        float *X = reg, *G = reg + xin_floats, *H = G + C * T * V;
        // stage x[n] (strided: the caller's permute(0,3,1,2) view, train.py:48)
        {
            const float *xn = a.x + n * a.x_sn;
            for (int e = tid; e < c_in * T * vi; e += NT) {
                const int v = e % vi, ct = e / vi, t = ct % T, c = ct / T;
                X[e] = xn[c * a.x_sc + t * a.x_st + v * a.x_sv];
            }
            if (L.n_txp > 0) {                       // (both sizes are multiples of 16 floats)
                float4 *z4 = reinterpret_cast<float4 *>(bufA);
                for (int e = tid; e < ((a.a0g ? T * txp_sci(vi) : P * SC) >> 2); e += NT)
                    z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        for (int j = 0; j < L.n_blocks && !STG_SKIP(a, 32); ++j) {
            const bool last = j == L.n_blocks - 1;
            float *yb = (last && L.n_txp == 0) ? yn : nullptr;
            if (L.blk[j].cin == Cfg::CIN0)
                stgcn_block_fwd<Cfg::CIN0, WAVES>(a, params, buffers, L.blk[j], n, vi, X, G, H, cs, red, wsn, statn,
                                                  last && L.n_txp > 0, a.a0g ? bufA + 2 * SW : bufA, yb, !last);
            else
                stgcn_block_fwd<Cfg::C, WAVES>(a, params, buffers, L.blk[j], n, vi, X, G, H, cs, red, wsn, statn,
                                               last && L.n_txp > 0, a.a0g ? bufA + 2 * SW : bufA, yb, !last);
            float *tmp = X; X = H; H = tmp;      // block output becomes the next block's input
        }
        if (L.n_txp == 0 || STG_SKIP(a, 16)) continue;
        if (a.a0g) {
            // hand the zero-bordered channel-major a_0 plane to txp_fwd_wave_kernel (linear 16-byte copy) and,
            // in training, save it position-major for the weight-gradient GEMM
            // (T channels in the in-place layout of txp_wave: padded row r at row slot r + 2, zeros in the spare slots;
            // the block wrote the plane in that layout, so this is a linear copy)
            const int SCI = txp_sci(vi);
            {
                float4 *dst = reinterpret_cast<float4 *>(a.a0g + (int64_t)n * a0_slot(V));
                const float4 *src = reinterpret_cast<const float4 *>(bufA);
                for (int e = tid; e < (T * SCI) >> 2; e += NT) dst[e] = src[e];
            }
            if (wsn) {        // saved planes: the C interior ROWS with their two border columns, [C*SW][P]
                float *d2 = wsn + ws_plane_off(L, V, 0);
                for (int h = 0; h < C; ++h)
                    for (int e = tid; e < SW * 3; e += NT) {              // (position, channel quad): 16-byte stores
                        const int col = e / 3, q = e - col * 3;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (q < T / 4) {
                            const float *src = bufA + (4 * q) * SCI + (h + 3) * SW + col;
                            v = make_float4(src[0], src[SCI], src[2 * SCI], src[3 * SCI]);
                        }
                        *reinterpret_cast<float4 *>(d2 + ((h * SW + col) * P + 4 * q)) = v;
                    }
            }
            continue;
        }
        // ---- TXP-CNN (model.py:187-195) ------------------------------------------------------
        float *bufB = reg;
        for (int e = tid; e < P * SC; e += NT) bufB[e] = 0.f;
        __syncthreads();
        const float *__restrict__ Pm = params;
        float *in = bufA, *out = bufB;
        // a_l leaves as a whole zero-bordered plane, transposed to position-major [(C+2)*SW][P] with
        // coalesced stores: the layout the weight-gradient GEMM (K = positions, 16 lanes = 16 channels/taps)
        // stages back with a linear LDS-DMA copy and reads without bank conflicts
        auto save_plane = [&](const float *pl, int idx) {
            if (!wsn) return;
            float *dst = wsn + ws_plane_off(L, V, idx);     // the C interior rows with their border columns, [C*SW][P]
            for (int h = 0; h < C; ++h)
                for (int e = tid; e < SW * P; e += NT) {
                    const int col = e / P, ch = e - col * P;
                    dst[(h * SW) * P + e] = pl[ch * SC + (h + 1) * SW + col];
                }
        };
        save_plane(in, 0);
        for (int l = 0; l < L.L; ++l) {
            float *zs = wsn ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr;
            const float alpha = Pm[L.prelus + l];
            if (l == 0)
                txp_layer_fwd<Cfg::T, WAVES>(Pm + L.txp_w[0], Pm + L.txp_b[0], alpha, 0, in, out, vi, V, zs, nullptr);
            else
                txp_layer_fwd<Cfg::P, WAVES>(Pm + L.txp_w[l], Pm + L.txp_b[l], alpha, 1, in, out, vi, V, zs, nullptr);
            __syncthreads();
            float *tmp = in; in = out; out = tmp;
            save_plane(in, l + 1);
        }
        txp_layer_fwd<Cfg::P, WAVES>(Pm + L.out_w, Pm + L.out_b, 0.f, 2, in, out, vi, V, nullptr, yn);
    }
}

static size_t fwd_lds_bytes(int V, int waves, bool slim = false, int cin = Cfg::C) {
    const int plane = slim ? Cfg::T * txp_sci(V) : fwd_plane_floats(V);
    const int reg = slim ? (cin + 2 * Cfg::C) * Cfg::T * V
                         : (plane > 3 * Cfg::C * Cfg::T * V ? plane : 3 * Cfg::C * Cfg::T * V);
    return (size_t)(plane + reg + Cfg::T * V + waves * 16 + 16) * sizeof(float);
}

}  // namespace stg

extern "C" int64_t stg_model_fwd_scratch_floats(const stg_model_desc *d, int N, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (N < 0 || V <= 0) return stg::fail(STG_EINVAL, "stg_model_fwd_scratch_floats: N=%d V=%d", N, V);
    const bool stamps = stg::diag_env("STG_STAMPS", 0) != 0;
    return stg::txp_wave_fits(l, V)
               ? (((int64_t)N * stg::a0_slot(V) + 3) & ~(int64_t)3) + 4 + stg::order_floats(N, V) + (stamps ? (int64_t)N * 32 : 0)
               : 0;
}

extern "C" int stg_model_fwd(const stg_model_desc *d, const float *params, const float *buffers, const float *x,
                             int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj,
                             int64_t a_sn, const int32_t *num_peds, int N, int V, float *y, float *ws,
                             float *stats, float *scratch, void **events, int n_events, void *stream) {
    using namespace stg;
    FwdArgs a{};
    const int rc = make_layout(d, &a.lay);
    if (rc != STG_OK) return rc;
    STG_REQUIRE(N >= 0 && V > 0, STG_EINVAL, "stg_model_fwd: bad sizes N=%d V=%d", N, V);
    if (N == 0) return STG_OK;                        // empty batch: nothing to read or write
    STG_REQUIRE(params && buffers && x && adj && y, STG_EINVAL, "stg_model_fwd: null pointer");
    if (N == 0) return STG_OK;
    a.params = params; a.buffers = buffers; a.x = x;
    a.x_sn = x_sn; a.x_sc = x_sc; a.x_st = x_st; a.x_sv = x_sv;
    a.adj = adj; a.a_sn = a_sn; a.num_peds = num_peds; a.N = N; a.V = V;
    a.y = y; a.ws = ws; a.ws_stride = ws_floats_per_scene(a.lay, V); a.stats = stats;
    const bool wave_path = txp_wave_fits(a.lay, V);
    STG_REQUIRE(!wave_path || scratch, STG_EINVAL, "stg_model_fwd: scratch (stg_model_fwd_scratch_floats) is null");
    STG_REQUIRE(!ws || (reinterpret_cast<uintptr_t>(ws) & 15) == 0, STG_EINVAL, "stg_model_fwd: ws must be 16-byte aligned");
    a.a0g = wave_path ? scratch : nullptr;
    hipStream_t st = as_stream(stream);
    EventList evl{events, events ? n_events : 0, 0, st};
    evl.mark();
    if (wave_path) {     // ragged batch: sorted scene list behind the a_0 planes
        int32_t *order = reinterpret_cast<int32_t *>(scratch + (((int64_t)N * a0_slot(V) + 3) & ~(int64_t)3) + 4);
        a.order = launch_scene_order(num_peds, N, V, order, order + N, st) ? order : nullptr;
    }
    a.debug_skip = diag_env("STG_DEBUG_SKIP", 0);
    int waves = V <= 12 ? 1 : (V <= 40 ? 2 : (V <= 80 ? 4 : 8));
    if (wave_path) waves = V <= 4 ? 1 : 2;     // blocks only (measured at V=32: 2 waves 74 us, 4: 79, 1: 96, 8: 172)
    if (a.lay.wg_waves) waves = a.lay.wg_waves;
    const size_t lds = fwd_lds_bytes(V, waves, wave_path, a.lay.blk[0].cin);
    STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "stg_model_fwd: V=%d needs %zu bytes of LDS (> %d)", V, lds,
                kLdsBytes);
    const dim3 grid((unsigned)N);
#define STG_LAUNCH_FWD(W)                                                                                    \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&model_fwd_kernel<W>),            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e_ != hipSuccess) return hip_fail(e_, "stg_model_fwd: hipFuncSetAttribute");                     \
        hipLaunchKernelGGL(model_fwd_kernel<W>, grid, dim3(W * 64), lds, st, a, params, buffers);                             \
    } while (0)
    switch (waves) {
        case 1: STG_LAUNCH_FWD(1); break;
        case 2: STG_LAUNCH_FWD(2); break;
        case 4: STG_LAUNCH_FWD(4); break;
        default: STG_LAUNCH_FWD(8); break;
    }
#undef STG_LAUNCH_FWD
    STG_LAUNCH_CHECK("stg_model_fwd");
    evl.mark();
    if (wave_path && !STG_SKIP(a, 16)) {
        TxpFwdArgs t{};
        t.lay = a.lay; t.params = params; t.num_peds = num_peds; t.N = N; t.V = V;
        t.a0g = scratch; t.y = y; t.ws = ws; t.ws_stride = a.ws_stride;
        t.stamps = diag_env("STG_STAMPS", 0) ? reinterpret_cast<unsigned long long *>(scratch + (((int64_t)N * a0_slot(V) + 3) & ~(int64_t)3) + 4 + order_floats(N, V)) : nullptr;
        const int serp = diag_env("STG_WALK", 1);
        // one launch for the whole (sorted) batch: V-tiers in separate launches were measured slower -- the few
        // large scenes of a real batch take one wave tens of microseconds each and need the small ones to overlap
        t.tier = SceneTier{a.order, a.order ? a.order + N : nullptr, -1, V, serp};
        t.Vl = V;
        const int rcw = launch_txp_fwd_wave(t, st);
        evl.mark();
        return rcw;
    }
    return STG_OK;
}
