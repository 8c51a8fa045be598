// model_bwd: backward of social_stgcnn.forward (model.py:182-198) / st_gcn.forward (model.py:145-155).
//
// Two kernels + a slab reduction, all fed by what the forward saved per scene (zero-bordered planes
// a_0..a_L and pre-activations z_l of the TXP-CNN; ax, colsum, g, h2 and the BatchNorm statistics of
// each st_gcn block):
//
//  K1 model_bwd_kernel<W>   scene-resident, persistent workgroups of W wave64.  Per scene, output conv
//     first then the hidden TXP layers in reverse:
//        dz_l   = d(a_{l+1}) * prelu'(z_l)        VALU (+ PReLU slope gradient); dz_l also goes to HBM for K2
//        d(a_l) = conv_transpose(dz_l, W_l) [+ d(a_{l+1})]   MFMA 16x16x4 f32: the forward's implicit
//                                                           GEMM with W^T and flipped taps
//     then the st_gcn blocks: BatchNorm backward with PER-SCENE statistics, PReLU, temporal conv,
//     1x1 convs -- VALU with wave-shuffle + LDS block reductions.  Small-parameter gradients
//     (block parameters, PReLU slopes) accumulate in LDS for the whole launch and leave as one slab row.
//  K2 txp_wgrad_kernel<W>   the TXP weight/bias gradients as ONE skinny GEMM per layer over ALL scenes:
//        dW_l[co][ci][tap] = sum_{scene,pos} dz_l[co][pos] a_l[ci][pos+tap],  db_l = sum dz_l
//     M = 12 out-channels (16-row tile), N = 9*c_in columns + a ones column (bias), K = every position of
//     every scene.  Each wave owns work items (scene, or <= 32-column chunk of a larger scene) round-robin with a
//     private LDS image (plane a_l + dz_l) and keeps the N/16 accumulator tiles in VGPRs for the whole launch --
//     no barriers, no atomics.
//  reduce_slabs_kernel      sums the slab rows of K1 and K2 into the flat gradient in a fixed order.
//
// On the fast path (one st_gcn block, V <= 68) the TXP input-gradient chain runs wave-per-scene in
// txp_wave.hip (B2) and K1 is launched in its lean form (B3: st_gcn blocks only, d(a_0) handed over through HBM);
// with num_peds the scenes are sorted by crowd size first (model_common.hpp) and the small ones run in their own,
// denser K1 launch.
#include "model_common.hpp"
#include "stgcn_block.hpp"
#include "txp_wave.hpp"
#include "txp_wgrad.hpp"
#include "tail_parts.hpp"
#include "nll_elem.hpp"

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
    // fused loss (stg_model_bwd_nll / _step): dy is V_pred (N, 5, P, V) and the kernel computes dV_pred itself
    const float *nll_target, *nll_weights;
    float *nll_losses;
    int64_t ws_stride;
    float *slab1;     // K1 slab rows: [gridDim.x][n_blk_params + n_txp]
    float *dzg;       // dz_l of the hidden TXP layers for K2: [N][L][P*C*V]
    float *dx;
    SceneTier tier;   // which scenes this launch serves (ragged batches: sorted, walked boustrophedon)
    int Vl;           // LDS geometry of the launch: >= every V_n of the tier; 0 = V
    int debug_skip;   // timing-only diagnostic (STG_DEBUG_SKIP): 1 wgrad, 2 dgrad, 4 st_gcn -- wrong results
};


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

// Fused loss of the workgroup-per-scene backward (stg_model_bwd_nll / _step): `yn` is the scene's V_pred (5, P, V); a
// thread takes (prediction step p, pedestrian w) and writes its five gradients -- rows f * P + p of the (C*P) x V array
// the input-gradient chain starts from -- into the dz plane and, position-major, into the weight-gradient GEMM's copy.
// Out of line: inlined, its live range pushed the kernel to 260 registers and one workgroup per SIMD pair.
__device__ __noinline__ void bwd_nll_stage(const float *yn, const float *tn, float weight, int vi, int V, int SC, int SW,
                                           float *dzb, float *dzo, float *nll_part, float *loss_out) {
    constexpr int C = Cfg::C, P = Cfg::P;
    const int tid = threadIdx.x, nt = blockDim.x;
    const float inv_cnt = 1.0f / (float)(P * vi);
    const float gs = inv_cnt * weight;
    float lacc = 0.f;
    for (int e = tid; e < P * vi; e += nt) {
        const int p = e / vi, w = e - p * vi;
        const float *q = yn + (int64_t)p * V + w;
        const float2 tg = *reinterpret_cast<const float2 *>(tn + ((int64_t)p * V + w) * 2);
        float g[5];
        lacc += nll_elem(q[0], q[(int64_t)P * V], q[(int64_t)2 * P * V], q[(int64_t)3 * P * V], q[(int64_t)4 * P * V], tg.x,
                         tg.y, true, g);
#pragma unroll
        for (int f = 0; f < C; ++f) {
            const int rc = f * P + p, ch = rc / C, h = rc - ch * C;
            const float dv = g[f] * gs;
            dzb[ch * SC + (h + 1) * SW + (w + 1)] = dv;
            dzo[(h * vi + w) * P + ch] = dv;
        }
    }
    lacc = wave_sum(lacc);
    if ((tid & 63) == 0) nll_part[tid >> 6] = lacc;
    __syncthreads();
    if (tid == 0) {                                    // fixed order: the same loss bits on every run
        float s = 0.f;
        for (int k = 0; k < nt >> 6; ++k) s += nll_part[k];
        *loss_out = s * inv_cnt;
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void model_bwd_kernel(const BwdArgs a, const float *params) {
    // (unlike the forward kernel, params is NOT __restrict__ here: scalar-loading the weights into SGPRs pushed
    // this kernel's SGPR spills up and measured 81 vs 69 us)
    constexpr int C = Cfg::C, T = Cfg::T, P = Cfg::P, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int V = a.V, tid = threadIdx.x;
    const int Vl = a.Vl > 0 ? a.Vl : V;               // LDS image sized for the largest scene of this launch's tier
    const ModelLayout &L = a.lay;
    const int scmax = txp_sc(Vl);
    const int plane_floats = P * scmax;
    // st_gcn phase needs 3 planes: h1 [C][T+2][V], dh2 [C][T+2][V], db1 [C][T][V]
    const int st_floats = (2 * C * (T + 2) + C * T) * Vl;
    const int reg_floats = plane_floats > st_floats ? plane_floats : st_floats;
    const int dcur_floats = P * C * Vl;
    const int n_small = L.n_blk_params + L.n_txp;
    float *gsm = sm;                                  // [n_small] block parameters, then the PReLU slopes
    float *dzb = gsm + ((n_small + 3) & ~3);          // [P][SC]   dz_l, zero-bordered (aliases the st_gcn planes)
    float *dcur = dzb + reg_floats;                   // [P*C*V]   gradient w.r.t. the layer output
    float *red = dcur + dcur_floats;                  // [WAVES*kRedMax]
    float *tot = red + WAVES * kRedMax;               // [kRedMax]
    const float *Pm = params;
    __shared__ float nll_part[WAVES];                 // (fused loss) per-wave partial sums of the scene's loss

    for (int e = tid; e < n_small; e += NT) gsm[e] = 0.f;

    int t_begin, t_end;
    tier_range(a.tier, a.N, V, t_begin, t_end);
    const int M = t_end - t_begin;
    for (int r = 0; r * (int)gridDim.x < M; ++r) {
        const int it = walk_item(r, blockIdx.x, gridDim.x, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        const int n = a.tier.order ? a.tier.order[t_begin + it] : it;
        int vi = a.num_peds ? a.num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        float *dxn = a.dx ? a.dx + (int64_t)n * L.blk[0].cin * T * V : nullptr;
        if (dxn && vi < V)
            for (int e = tid; e < L.blk[0].cin * T * (V - vi); e += NT) {
                const int r = e / (V - vi), w = vi + (e - r * (V - vi));
                dxn[(int64_t)r * V + w] = 0.f;
            }
        if (vi == 0) {
            if (a.nll_target && tid == 0) a.nll_losses[n] = 0.f;
            continue;
        }
        const float *wsn = a.ws + n * a.ws_stride;
        const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi;
        const int out_rows = L.n_txp > 0 ? C * P : C * T;
        const float *dyn = a.dy + (int64_t)n * out_rows * V;
        __syncthreads();
        if (L.n_txp > 0) {
            // ---- TXP-CNN backward (input-gradient chain) -------------------------------------------
            for (int e = tid; e < P * SC; e += NT) dzb[e] = 0.f;
            __syncthreads();
            for (int l = L.L; l >= 0; --l) {
                const bool is_out = l == L.L;
                if (is_out) {
                    // (dz of the output conv is dy; it also leaves position-major for the weight-gradient GEMM)
                    float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + L.L) * dz_slot(V);
                    constexpr int U = 4;
                    if (a.nll_target) {
                        bwd_nll_stage(dyn, a.nll_target + (int64_t)n * P * V * 2, a.nll_weights ? a.nll_weights[n] : 1.f,
                                      vi, V, SC, SW, dzb, dzo, nll_part, a.nll_losses + n);
                    } else
                    for (int e0 = tid; e0 < P * npos; e0 += NT * U) {
                        float dv[U];
                        int li[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + u * NT;
                            const int ec = e < P * npos ? e : 0;
                            const int ch = ec / npos, p = ec - ch * npos, h = p / vi, w = p - h * vi;
                            li[u] = ch * SC + (h + 1) * SW + (w + 1);
                            dv[u] = dyn[(int64_t)(ch * C + h) * V + w];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u)
                            if (e0 + u * NT < P * npos) {
                                const int e = e0 + u * NT, ch = e / npos;
                                dzb[li[u]] = dv[u];
                                dzo[(e - ch * npos) * P + ch] = dv[u];
                            }
                    }
                    __syncthreads();
                } else {
                    const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
                    float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + l) * dz_slot(V);
                    const float alpha = Pm[L.prelus + l];
                    float s[1] = {0.f};
                    constexpr int U = 4;                       // z loads in flight per lane
                    for (int e0 = tid; e0 < P * npos; e0 += NT * U) {
                        float zv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + u * NT;
                            zv[u] = e < P * npos ? zl[e] : 1.f;
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int e = e0 + u * NT;
                            if (e < P * npos) {
                                const int ch = e / npos, p = e - ch * npos, h = p / vi, w = p - h * vi;
                                const float z = zv[u], d = dcur[e];
                                float dz = d;
                                if (!(z > 0.f)) {
                                    dz = alpha * d;
                                    s[0] = fmaf(d, z, s[0]);
                                }
                                dzb[ch * SC + (h + 1) * SW + (w + 1)] = dz;
                                dzo[p * P + ch] = dz;            // position-major hand-off [pos][12]
                            }
                        }
                    }
                    block_reduce<1, WAVES>(s, red, tot);
                    if (tid == 0) gsm[L.n_blk_params + l] += tot[0];
                }
                const int w_off = is_out ? L.out_w : L.txp_w[l];
                if (!STG_SKIP(a, 2)) {
                    if (l == 0)
                        txp_dgrad<Cfg::T, WAVES>(Pm + w_off, dzb, dcur, vi, false);
                    else
                        txp_dgrad<Cfg::P, WAVES>(Pm + w_off, dzb, dcur, vi, !is_out);
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
        float *H1 = dzb, *DH2 = dzb + C * (T + 2) * Vl, *DB1 = DH2 + C * (T + 2) * Vl;
        for (int j = L.n_blocks - 1; j >= 0 && !STG_SKIP(a, 4); --j) {
            const float *xin = j > 0 ? wsn + L.ws_hdr_floats + (int64_t)L.blk[j - 1].ws_s * V : nullptr;
            float *dxs = j > 0 ? dcur : nullptr;
            float *dxg = j == 0 ? dxn : nullptr;
            if (L.blk[j].cin == Cfg::CIN0)
                stgcn_block_bwd<Cfg::CIN0, WAVES, true>(a, params, L.blk[j], n, vi, dcur, H1, DH2, DB1, red, tot, gsm, wsn,
                                                        xin, dxs, dxg, nullptr);
            else
                stgcn_block_bwd<Cfg::C, WAVES, true>(a, params, L.blk[j], n, vi, dcur, H1, DH2, DB1, red, tot, gsm, wsn, xin,
                                                     dxs, dxg, nullptr);
        }
    }
    __syncthreads();
    float *slab = a.slab1 + (int64_t)blockIdx.x * n_small;
    for (int e = tid; e < n_small; e += NT) slab[e] = gsm[e];
}

// ------------------------------------------------------------------------------------------
// slab reduction: grad[p] = sum of the slab rows that hold parameter p, in row order
// ------------------------------------------------------------------------------------------
struct ReduceSeg {
    int p0, len, rows, row_stride;
    int64_t base;          // offset of (row 0, parameter p0) in the slab buffer
};
struct ReduceArgs {
    ReduceSeg seg[kMaxTxp + 3];
    int n_seg;
    const float *slabs;
    float *grad;
    int n_params;
};

// One workgroup owns 8 consecutive parameters; its 32 lane-groups stride over the slab rows (up to 2048 of
// them), 4 independent row loads in flight each; the 32 partial sums meet in LDS in a fixed order.
// The step tail (stg_model_bwd_step) rides in the same launch: the reducing workgroups also apply SGD to their eight
// parameters, workgroups behind them fold the BatchNorm statistics (one each) and sum the reported loss.
struct TailArgs {
    float *params;              // null: plain reduction
    const float *lr_dev;
    float lr;
    const float *stats, *losses, *weights;
    const int32_t *num_peds;
    float *buffers, *total;
    NbtPtrs nbt;
    int n_reduce, n_buffers, N, stat_floats;
    float momentum;
};

__global__ __launch_bounds__(256) void reduce_slabs_kernel(const ReduceArgs a, const TailArgs t_) {
    __shared__ float part[32][8];
    if ((int)blockIdx.x >= t_.n_reduce) {
        float *acc_s = &part[0][0];                      // 256 floats
        __shared__ float dec_s[256];
        __shared__ int cnt_s[256];
        const int b = (int)blockIdx.x - t_.n_reduce;
        if (b < t_.n_buffers)
            bn_fold_body<256>(b, t_.stats, t_.num_peds, t_.N, t_.stat_floats, t_.momentum, t_.buffers, t_.nbt, acc_s, dec_s,
                              cnt_s);
        else if (t_.total)
            weighted_sum_body(t_.losses, t_.weights, t_.N, t_.total, acc_s);
        return;
    }
    const int col = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int p = blockIdx.x * 8 + col;
    float s = 0.f;
    if (p < a.n_params) {
        for (int q = 0; q < a.n_seg; ++q) {
            const ReduceSeg g = a.seg[q];
            if (p >= g.p0 && p < g.p0 + g.len) {
                const float *src = a.slabs + g.base + (p - g.p0);
                int k = grp;
                for (; k + 992 < g.rows; k += 1024) {       // 32 rows in flight per lane (2048 slab rows: 2 trips)
                    float v[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) v[u] = src[(int64_t)(k + 32 * u) * g.row_stride];
#pragma unroll
                    for (int u = 0; u < 32; ++u) s += v[u];
                }
                for (; k + 480 < g.rows; k += 512) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = src[(int64_t)(k + 32 * u) * g.row_stride];
#pragma unroll
                    for (int u = 0; u < 16; ++u) s += v[u];
                }
                for (; k + 96 < g.rows; k += 128) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(k + 32 * u) * g.row_stride];
#pragma unroll
                    for (int u = 0; u < 4; ++u) s += v[u];
                }
                for (; k < g.rows; k += 32) s += src[(int64_t)k * g.row_stride];
            }
        }
    }
    part[grp][col] = s;
    __syncthreads();
    if (grp == 0 && p < a.n_params) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += part[g][col];
        a.grad[p] = t;
        if (t_.params) t_.params[p] -= (t_.lr_dev ? t_.lr_dev[0] : t_.lr) * t;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int env_waves(const char *name, int dflt) {     // (diagnostic builds only; the product returns dflt)
    const int w = diag_env(name, dflt);
    return (w == 1 || w == 2 || w == 4 || w == 8) ? w : dflt;
}

static size_t bwd_lds_bytes(const ModelLayout &L, int V, int waves) {
    const int plane = Cfg::P * txp_sc(V);
    const int st = (2 * Cfg::C * (Cfg::T + 2) + Cfg::C * Cfg::T) * V;
    const int reg = plane > st ? plane : st;
    const int dcur = Cfg::P * Cfg::C * V;
    const int n_small = L.n_blk_params + L.n_txp;
    const size_t fl = ((n_small + 3) & ~3) + (size_t)reg + (size_t)dcur + (size_t)waves * kRedMax + kRedMax;
    return fl * sizeof(float);
}

// measured (profiles/): the st_gcn backward is fastest with ONE wave per scene up to V ~ 40 (no cross-wave
// barriers in its ~15 block reductions), more waves only when a scene's rows no longer fit one wave's registers
static int bwd_waves(const ModelLayout &L, int N, int V) {
    int w = 0;
    use_wave_path(L, N, V, &w);          // (small batches: more waves per scene, see use_wave_path)
    return w ? w : (V <= 40 ? 1 : (V <= 80 ? 4 : 8));
}

static int bwd_grid_w(const ModelLayout &L, int N, int V, int waves) {
    const size_t lds = bwd_lds_bytes(L, V, waves);
    if (lds > (size_t)kLdsBytes) return -1;
    int per_cu = (int)(kLdsBytes / lds);
    const int by_waves = 8 / waves > 0 ? 8 / waves : 1;   // 256-VGPR kernel: 2 waves per SIMD
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    int grid = kNumCU * per_cu;
    if (const int g = diag_env("STG_BWD_GRID", 0)) grid = g > 0 ? g : grid;
    return grid < N ? grid : N;
}
static int bwd_grid(const ModelLayout &L, int N, int V) { return bwd_grid_w(L, N, V, bwd_waves(L, N, V)); }
// ragged batches padded beyond kBwdTierV: the scenes up to kBwdTierV pedestrians (most of a real batch) run in a
// second launch with ONE wave per scene and the LDS image of V = kBwdTierV; slab rows of both launches are stacked
constexpr int kBwdTierV = 32;
static int bwd_grid_small(const ModelLayout &L, int N, int V) {
    return V > kBwdTierV ? bwd_grid_w(L, N, kBwdTierV, 1) : 0;
}

// scratch carve of stg_model_bwd (offsets in floats):
//   rows   small-parameter gradient rows [n_rows][n_blk_params + n_txp]: one per persistent workgroup of K1 (both
//          launches of a two-tier ragged run), or one per SCENE on the wave-per-scene path
//   slab2  weight-gradient slab rows of K2          dzg  dz_l hand-off [N][L+1][dz_slot(V)]
//   order  sorted scene list + tier offsets (int32)
struct BwdCarve {
    int64_t rows, slab2, dzg, order, total;
};
static bool bwd_carve(const ModelLayout &L, int N, int V, BwdCarve *c, WgradGeom *wg) {
    const int g1 = bwd_grid(L, N, V);
    if (g1 < 0) return false;
    int gs = bwd_grid_small(L, N, V);
    if (gs < 0) gs = 0;
    int64_t n_rows = g1 + gs;
    if (use_wave_path(L, N, V, nullptr) && N > n_rows) n_rows = N;
    c->rows = 0;
    int64_t fl = (n_rows * (L.n_blk_params + L.n_txp) + 3) & ~(int64_t)3;
    c->slab2 = fl;
    c->dzg = fl;
    if (L.n_txp > 0) {
        if (!wgrad_geom(L, N, V, wg)) return false;
        fl = (fl + wgrad_slab_base(L.L + 1, wg->rows) + 3) & ~(int64_t)3;
        c->dzg = fl;
        fl = (fl + (int64_t)N * (L.L + 1) * dz_slot(V) + 3) & ~(int64_t)3;
    }
    c->order = fl;
    c->total = fl + order_floats(N, V);
    return true;
}

}  // namespace stg

extern "C" {

int64_t stg_model_bwd_scratch_floats(const stg_model_desc *d, int N, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (N <= 0 || V <= 0) return stg::fail(STG_EINVAL, "stg_model_bwd_scratch_floats: N=%d V=%d", N, V);
    stg::BwdCarve c;
    stg::WgradGeom wg{};
    if (!stg::bwd_carve(l, N, V, &c, &wg)) return stg::fail(STG_ELDS, "stg_model_bwd: V=%d does not fit the LDS of one CU", V);
    return c.total;
}

// dy = dV_pred, or (nll_target != null) V_pred itself: the wave-per-scene backward then computes the loss gradient in
// its input stage and writes the per-scene losses (stg_model_bwd_nll)
static int model_bwd_impl(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                          int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn,
                          const int32_t *num_peds, int N, int V, const float *dy, const float *nll_target,
                          const float *nll_weights, float *nll_losses, const stg_step_tail *tail, const float *ws,
                          float *scratch, float *grad_params, float *dx, void **events, int n_events, void *stream) {
    using namespace stg;
    BwdArgs a{};
    const int rc = make_layout(d, &a.lay);
    if (rc != STG_OK) return rc;
    const ModelLayout &L = a.lay;
    STG_REQUIRE(N >= 0 && V > 0, STG_EINVAL, "stg_model_bwd: bad sizes N=%d V=%d", N, V);
    STG_REQUIRE(grad_params, STG_EINVAL, "stg_model_bwd: null grad_params");
    STG_REQUIRE(N == 0 || (params && buffers && x && adj && dy && ws && scratch), STG_EINVAL,
                "stg_model_bwd: null pointer");
    STG_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
                STG_EINVAL, "stg_model_bwd: ws / scratch must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    EventList evl{events, events ? n_events : 0, 0, st};
    evl.mark();
    if (N == 0) {
        hipError_t e = hipMemsetAsync(grad_params, 0, sizeof(float) * L.n_params, st);
        if (e != hipSuccess) return hip_fail(e, "stg_model_bwd: memset");
        return STG_OK;
    }
    BwdCarve cv;
    WgradGeom wg{};
    STG_REQUIRE(bwd_carve(L, N, V, &cv, &wg), STG_ELDS, "stg_model_bwd: V=%d does not fit the LDS of one CU", V);
    const int n_small = L.n_blk_params + L.n_txp;
    float *rows = scratch + cv.rows, *slab2 = scratch + cv.slab2;
    float *dzg = L.n_txp > 0 ? scratch + cv.dzg : nullptr;
    // the wave-per-scene kernels compute no input gradient (the reference never needs one: x is data), and their
    // saved pre-activations are laid out for themselves: a caller that wants dx runs BOTH passes on the workgroup
    // kernels (STG_OPT_WG_PATH in the descriptor)
    int auto_waves = 0;
    const bool wave_path = use_wave_path(L, N, V, &auto_waves);
    (void)auto_waves;
    STG_REQUIRE(wave_path || !(L.flags & STG_OPT_BF16_STORE), STG_EUNSUPPORTED,
                "stg_model_bwd: bf16 storage (STG_OPT_BF16_STORE) is built for the wave-per-scene kernels only");
    STG_REQUIRE(!((L.flags & STG_OPT_BF16_STORE) && (L.flags & STG_OPT_SPLIT_BF16)), STG_EUNSUPPORTED,
                "stg_model_bwd: STG_OPT_BF16_STORE and STG_OPT_SPLIT_BF16 cannot be combined");
    if (nll_target && ((L.flags & STG_OPT_SPLIT_BF16) || L.n_txp == 0))
        return STG_EUNSUPPORTED;        // (no message: the caller falls back to stg_nll_fwd + stg_model_bwd)
    STG_REQUIRE(!(wave_path && dx), STG_EUNSUPPORTED,
                "stg_model_bwd: dx is only computed by the workgroup-per-scene kernels: set STG_OPT_WG_PATH in the "
                "descriptor of the forward and the backward call");
    a.params = params; a.buffers = buffers; a.x = x;
    a.x_sn = x_sn; a.x_sc = x_sc; a.x_st = x_st; a.x_sv = x_sv;
    a.adj = adj; a.a_sn = a_sn; a.num_peds = num_peds; a.N = N; a.V = V;
    a.nll_target = nll_target; a.nll_weights = nll_weights; a.nll_losses = nll_losses;
    a.dy = dy; a.ws = ws; a.ws_stride = ws_floats_per_scene(L, V); a.slab1 = rows; a.dzg = dzg; a.dx = dx;
    // ragged batch: sorted scene list at the tail of the scratch buffer
    int32_t *order = reinterpret_cast<int32_t *>(scratch + cv.order);
    bool sorted;
    if (wave_path && scene_order_applies(num_peds, N, V)) {
        // the wave-per-scene forward left its order in the workspace's batch tail (stg_model_ws_tail_floats)
        order = reinterpret_cast<int32_t *>(const_cast<float *>(ws) + (int64_t)N * ws_floats_per_scene(L, V) +
                                            ws_tail_wp_floats(L, V));
        sorted = true;
    } else {
        sorted = launch_scene_order(num_peds, N, V, order, order + N, st, order + N + V + 2);
    }
    int32_t *order_peds = order + N + V + 2;
    const int serp = diag_env("STG_WALK", 1);
    a.tier = SceneTier{sorted ? order : nullptr, sorted ? order + N : nullptr, -1, V, serp};
    a.debug_skip = diag_env("STG_DEBUG_SKIP", 0);
    int slab_rows;
    if (wave_path) {
        // K1 (wave per scene): TXP input-gradient chain + st_gcn block backward; one small-gradient row per scene
        TxpBwdArgs t{};
        t.lay = L; t.params = params; t.num_peds = num_peds; t.N = N; t.V = V; t.dy = dy; t.ws = ws;
        t.nll_target = nll_target; t.nll_weights = nll_weights; t.nll_losses = nll_losses;
        t.x = x; t.x_sn = x_sn; t.x_sc = x_sc; t.x_st = x_st; t.x_sv = x_sv; t.adj = adj; t.a_sn = a_sn;
        t.ws_stride = a.ws_stride; t.dzg = dzg; t.rows = rows;
        t.tier = a.tier;
        t.Vl = V;
        t.debug_skip = a.debug_skip;
        t.split_bf16 = (L.flags & STG_OPT_SPLIT_BF16) ? 1 : 0;
        t.stagger = diag_env("STG_STAGGER_B", 0);
        // the prepared A operands of the exact-bf16 chain sit in the batch tail of the workspace (written by the forward)
        if (txp_bwd_x6_fits(L, V))
            t.wp = reinterpret_cast<const unsigned *>(ws + (int64_t)N * a.ws_stride);
        const int rcw = launch_txp_bwd_wave(t, st);
        if (rcw != STG_OK) return rcw;
        slab_rows = N;
    } else {
        const int waves = bwd_waves(L, N, V);
        const size_t lds = bwd_lds_bytes(L, V, waves);
        STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "stg_model_bwd: V=%d needs %zu bytes of LDS (> %d)", V, lds,
                    kLdsBytes);
        const int grid = bwd_grid(L, N, V);
#define STG_LAUNCH_BWD(W)                                                                                    \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&model_bwd_kernel<W>),            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e_ != hipSuccess) return hip_fail(e_, "stg_model_bwd: hipFuncSetAttribute");                     \
        hipLaunchKernelGGL(model_bwd_kernel<W>, dim3(grid), dim3(W * 64), lds, st, a, params);               \
    } while (0)
        slab_rows = grid;
        if (a.tier.order && bwd_grid_small(L, N, V) > 0) {
            // large scenes: `waves` per scene, slab rows [0, grid); small scenes: one wave per scene with the LDS image
            // (and so the residency) of V = kBwdTierV, slab rows behind them
            a.tier.v_lo = kBwdTierV;
            switch (waves) {
                case 1: STG_LAUNCH_BWD(1); break;
                case 2: STG_LAUNCH_BWD(2); break;
                case 4: STG_LAUNCH_BWD(4); break;
                default: STG_LAUNCH_BWD(8); break;
            }
            STG_LAUNCH_CHECK("stg_model_bwd: K1 (large scenes)");
            const int grid_hi = grid;
            {
                const int grid = bwd_grid_small(L, N, V);
                const size_t lds = bwd_lds_bytes(L, kBwdTierV, 1);
                a.Vl = kBwdTierV;
                a.tier.v_lo = -1; a.tier.v_hi = kBwdTierV;
                a.slab1 = rows + (int64_t)grid_hi * n_small;
                STG_LAUNCH_BWD(1);
                slab_rows = grid_hi + grid;
            }
            a.slab1 = rows;
            a.Vl = 0;
            a.tier.v_lo = -1; a.tier.v_hi = V;
        } else {
            switch (waves) {
                case 1: STG_LAUNCH_BWD(1); break;
                case 2: STG_LAUNCH_BWD(2); break;
                case 4: STG_LAUNCH_BWD(4); break;
                default: STG_LAUNCH_BWD(8); break;
            }
        }
#undef STG_LAUNCH_BWD
        STG_LAUNCH_CHECK("stg_model_bwd: K1");
    }
    evl.mark();

    ReduceArgs r{};
    r.slabs = scratch;
    r.grad = grad_params;
    r.n_params = L.n_params;
    r.seg[r.n_seg++] = ReduceSeg{0, L.n_blk_params, slab_rows, n_small, cv.rows};
    if (L.n_txp > 0) {
        r.seg[r.n_seg++] = ReduceSeg{L.prelus, L.n_txp, slab_rows, n_small, cv.rows + (int64_t)L.n_blk_params};
        WgradArgs w{};
        w.lay = L; w.num_peds = num_peds; w.order = a.tier.order; w.order_peds = sorted ? order_peds : nullptr;
        w.key_start = a.tier.key_start;
        w.serpentine = a.tier.serpentine; w.N = N; w.V = V;
        w.ws = ws; w.dzg = dzg; w.ws_stride = a.ws_stride; w.slab2 = slab2; w.rows = wg.rows; w.debug_skip = a.debug_skip;
        for (int l = 0; l <= L.L + 1; ++l) w.wg_begin[l] = wg.wg_begin[l];
        if (!STG_SKIP(a, 1)) {
            const int rck = launch_txp_wgrad(w, wg, st);
            if (rck != STG_OK) return rck;
        }
        evl.mark();
        for (int l = 0; l <= L.L; ++l) {
            const int p0 = l == L.L ? L.out_w : L.txp_w[l];
            // (only the rows the layer's workgroups wrote: no memset of the slab needed)
            r.seg[r.n_seg++] = ReduceSeg{p0, wgrad_row_len(l), STG_SKIP(a, 1) ? 0 : wg.wg_begin[l + 1] - wg.wg_begin[l],
                                         wgrad_row_len(l), cv.slab2 + wgrad_slab_base(l, wg.rows)};
        }
    }
    TailArgs ta{};
    ta.n_reduce = (L.n_params + 7) / 8;
    int extra = 0;
    if (tail) {
        ta.params = tail->params; ta.lr_dev = tail->lr_dev; ta.lr = tail->lr;
        ta.stats = tail->stats; ta.buffers = tail->buffers; ta.total = tail->total;
        ta.losses = nll_losses; ta.weights = nll_weights; ta.num_peds = num_peds;
        ta.N = N; ta.stat_floats = L.stat_floats; ta.momentum = d->bn_momentum;
        ta.n_buffers = (tail->stats && N > 0) ? L.n_buffers : 0;
        ta.nbt.n = tail->nbt ? tail->n_bn : 0;
        for (int k = 0; k < ta.nbt.n; ++k) ta.nbt.p[k] = tail->nbt[k];
        extra = ta.n_buffers + 1;
    }
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ta.n_reduce + extra), dim3(256), 0, st, r, ta);
    STG_LAUNCH_CHECK("stg_model_bwd: reduce_slabs");
    evl.mark();
    evl.finish();
    return STG_OK;
}

int stg_model_bwd(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                  int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                  int N, int V, const float *dy, const float *ws, float *scratch, float *grad_params, float *dx,
                  void **events, int n_events, void *stream) {
    return model_bwd_impl(d, params, buffers, x, x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, N, V, dy, nullptr, nullptr,
                          nullptr, nullptr, ws, scratch, grad_params, dx, events, n_events, stream);
}

int stg_model_bwd_nll(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                      int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                      int N, int V, const float *y, const float *target, const float *weights, float *losses,
                      const float *ws, float *scratch, float *grad_params, void **events, int n_events, void *stream) {
    if (!target || !losses || (N > 0 && !y)) return stg::fail(STG_EINVAL, "stg_model_bwd_nll: null pointer");
    return model_bwd_impl(d, params, buffers, x, x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, N, V, y, target, weights,
                          losses, nullptr, ws, scratch, grad_params, nullptr, events, n_events, stream);
}

int stg_model_bwd_step(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                       int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                       int N, int V, const float *y, const float *target, const float *weights, float *losses,
                       const float *ws, float *scratch, float *grad_params, const stg_step_tail *tail, void **events,
                       int n_events, void *stream) {
    if (!target || !losses || (N > 0 && !y)) return stg::fail(STG_EINVAL, "stg_model_bwd_step: null pointer");
    if (!tail || !tail->params || (tail->stats && !tail->buffers) || tail->n_bn < 0 || tail->n_bn > 3 * STG_MAX_BLOCKS)
        return stg::fail(STG_EINVAL, "stg_model_bwd_step: bad tail");
    if (N == 0) return STG_EUNSUPPORTED;               // (an empty batch has no backward launch to ride on)
    return model_bwd_impl(d, params, buffers, x, x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, N, V, y, target, weights,
                          losses, tail, ws, scratch, grad_params, nullptr, events, n_events, stream);
}

}  // extern "C"
