// txp_wave: the TXP-CNN (model.py:187-195) forward and its input-gradient chain as WAVE-PER-SCENE
// kernels -- the fast path for scenes whose two activation planes fit a wave-private LDS image
// (V <= ~57).  One wave64 owns one scene-window: no workgroup barriers, no cross-wave imbalance; six to
// eight independent waves per CU keep the four matrix pipes fed while each wave alternates between
// its MFMA stream and its own LDS / HBM traffic.
//
// Every 3x3 conv over the (5, V) plane is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32):
//   forward   out[co][pos] = b[co] + sum_{tap,ci} W[co][ci][tap] in[ci][pos+tap]
//             M = 12 out-channels (16-row tile), N = 16 positions, K = (tap, 4 channels)
//   backward  d in[ci][pos] = sum_{tap,co} W[co][ci][tap] dz[co][pos-tap]   (same GEMM, W^T, flipped taps)
// The B operand is one ds_read_b32 per MFMA from a zero-bordered channel-major plane whose channel
// stride is == 16 (mod 32) dwords; weights sit in VGPRs for the whole layer (the next layer's are
// prefetched while the current layer computes).  Epilogues write position-major [pos][12] images with
// 16-byte stores: the layout the weight-gradient GEMM (model_bwd.hip, K2) stages back with LDS-DMA.
#include "model_common.hpp"
#include "txp_wave.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P, T = Cfg::T;

// forward A operand: lane (co = l&15, kq = l>>4) of K-step (tap, j) holds W[co][4j+kq][tap]
template <int CINL>
__device__ __forceinline__ void load_w_fwd(const float *__restrict__ W, float (&wreg)[CINL * 9 / 4]) {
    const int lane = threadIdx.x & 63, co = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < CINL / 4; ++j)
            wreg[tap * (CINL / 4) + j] = co < P ? W[(co * CINL + 4 * j + kq) * 9 + tap] : 0.f;
}

// input-gradient A operand: lane (ci = l&15, kq) of K-step (tap', j) holds W[4j+kq][ci][8 - tap']
template <int CINL>
__device__ __forceinline__ void load_w_bwd(const float *__restrict__ W, float (&wreg)[27]) {
    const int lane = threadIdx.x & 63, ci = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            wreg[tap * 3 + j] = ci < CINL ? W[((4 * j + kq) * CINL + ci) * 9 + (8 - tap)] : 0.f;
}

// one pair of 16-position tiles: acc[u][r] = sum_k wreg[k] * plane[base[u] + koff(k)]
template <int KJ>
__device__ __forceinline__ void tile_pair(const float (&wreg)[9 * KJ], const float *plane, const int (&base)[2],
                                          int SW, int SC, f32x4 &acc0, f32x4 &acc1) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int toff = (tap / 3) * SW + (tap % 3);
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const float b0 = plane[base[0] + 4 * j * SC + toff];
            const float b1 = plane[base[1] + 4 * j * SC + toff];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * KJ + j], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * KJ + j], b1, acc1, 0, 0, 0);
        }
    }
}

struct TileGeom {
    int hh[2], ww[2], base[2], pos[2];
    bool ok[2];
};
__device__ __forceinline__ TileGeom tile_geom(int tile0, int vi, int npos, int SW, int SC) {
    const int lane = threadIdx.x & 63, nq = lane & 15, kq = lane >> 4;
    TileGeom g;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = (tile0 + u) * 16 + nq;
        g.ok[u] = p < npos;
        g.pos[u] = g.ok[u] ? p : 0;
        g.hh[u] = g.pos[u] / vi;
        g.ww[u] = g.pos[u] - g.hh[u] * vi;
        g.base[u] = kq * SC + g.hh[u] * SW + g.ww[u];
    }
    return g;
}

// zero a wave-private plane (borders must be 0; interiors are rewritten every layer)
__device__ __forceinline__ void wave_zero(float *p, int n4) {
    const int lane = threadIdx.x & 63;
    float4 *q = reinterpret_cast<float4 *>(p);
    for (int e = lane; e < n4; e += 64) q[e] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// linear global -> LDS copy of nvec 16-byte vectors with LDS-DMA (caller waits vmcnt(0))
__device__ __forceinline__ void wave_dma(const float *__restrict__ src, float *lds_dst, int nvec) {
    const int lane = threadIdx.x & 63;
    for (int i = 0; i * 64 < nvec; ++i) {
        const int e = i * 64 + lane;
        if (e < nvec)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * e),
                                             (__attribute__((address_space(3))) void *)(lds_dst + 256 * i), 16, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// KIND 0: layer 0 (PReLU), 1: hidden layer with residual, 2: output conv (writes y)
template <int CINL, int KIND>
__device__ __forceinline__ void fwd_layer(const float (&wreg)[CINL * 9 / 4], const float *__restrict__ bias, float alpha,
                                          const float *in, float *out, int vi, int V, float *zsave, float *psave,
                                          float *yout) {
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi, ntiles = (npos + 15) >> 4;
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = kq < 3 ? bias[4 * kq + r] : 0.f;
    for (int tile0 = 0; tile0 < ntiles; tile0 += 2) {
        const TileGeom g = tile_geom(tile0, vi, npos, SW, SC);
        f32x4 acc0 = binit, acc1 = binit;
        tile_pair<CINL / 4>(wreg, in, g.base, SW, SC, acc0, acc1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!g.ok[u] || kq == 3) continue;
            const f32x4 z = u == 0 ? acc0 : acc1;
            if (KIND == 2) {
                // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
#pragma unroll
                for (int r = 0; r < 4; ++r) yout[(int64_t)((4 * kq + r) * C + g.hh[u]) * V + g.ww[u]] = z[r];
            } else {
                const int pp = (g.hh[u] + 1) * SW + (g.ww[u] + 1);
                f32x4 av;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int li = (4 * kq + r) * SC + pp;
                    float v = z[r] > 0.f ? z[r] : alpha * z[r];
                    if (KIND == 1) v += in[li];
                    out[li] = v;
                    av[r] = v;
                }
                if (zsave) {   // position-major [pos][12]: one 16-byte store per lane
                    *reinterpret_cast<f32x4 *>(zsave + g.pos[u] * P + 4 * kq) = z;
                    *reinterpret_cast<f32x4 *>(psave + pp * P + 4 * kq) = av;
                }
            }
        }
    }
}

// zero the border positions of a saved position-major plane [(C+2)*SW][P] (interiors come from the epilogue)
__device__ __forceinline__ void zero_saved_borders(float *psave, int vi) {
    const int lane = threadIdx.x & 63, SW = txp_sw(vi);
    const int nb = 2 * SW + 2 * C;                  // top row, bottom row, left/right of the C inner rows
    for (int e = lane; e < nb * 3; e += 64) {
        const int b = e / 3, q = e - b * 3;
        int pos;
        if (b < SW) pos = b;
        else if (b < 2 * SW) pos = (C + 1) * SW + (b - SW);
        else {
            const int k = b - 2 * SW, row = 1 + (k >> 1);
            pos = row * SW + ((k & 1) ? SW - 1 : 0);
        }
        *reinterpret_cast<float4 *>(psave + pos * P + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void txp_fwd_wave_kernel(const TxpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ModelLayout &L = a.lay;
    const int V = a.V, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = plane_slot(V);
    float *pa = sm + wave * 2 * slot, *pb = pa + slot;
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
    if (n >= a.N) return;
    int vi = a.num_peds ? a.num_peds[n] : V;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    if (vi == 0) return;                               // (the block kernel already zero-filled y)
    const int SC = txp_sc(vi);
    const float *Pm = a.params;
    float *yn = a.y + (int64_t)n * (C * P) * V;
    float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;

    wave_dma(a.a0g + (int64_t)n * slot, pa, (P * SC) >> 2);
    wave_zero(pb, (P * SC) >> 2);
    float w0[T * 9 / 4];
    load_w_fwd<T>(Pm + L.txp_w[0], w0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    float *in = pa, *out = pb;
    float wa[27], wb[27];          // two weight register sets: layer l computes from one while l+1 loads
    auto w_of = [&](int l) { return Pm + (l < L.L ? L.txp_w[l] : L.out_w); };
    auto zs_of = [&](int l) { return wsn ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr; };
    auto ps_of = [&](int l) {
        float *ps = wsn ? wsn + ws_plane_off(L, V, l + 1) : nullptr;
        if (ps) zero_saved_borders(ps, vi);
        return ps;
    };
    // layer 0 (weights w0 already resident); layer 1's weights load meanwhile
    load_w_fwd<P>(w_of(1), wa);
    {
        float *zs = zs_of(0), *ps = ps_of(0);
        fwd_layer<T, 0>(w0, Pm + L.txp_b[0], Pm[L.prelus], in, out, vi, V, zs, ps, nullptr);
        float *t = in; in = out; out = t;
    }
    int l = 1;
    bool in_a = true;              // which register set holds layer l's weights
    for (; l < L.L; ++l) {
        float *zs = zs_of(l), *ps = ps_of(l);
        __builtin_amdgcn_wave_barrier();
        if (in_a) {
            load_w_fwd<P>(w_of(l + 1), wb);
            fwd_layer<P, 1>(wa, Pm + L.txp_b[l], Pm[L.prelus + l], in, out, vi, V, zs, ps, nullptr);
        } else {
            load_w_fwd<P>(w_of(l + 1), wa);
            fwd_layer<P, 1>(wb, Pm + L.txp_b[l], Pm[L.prelus + l], in, out, vi, V, zs, ps, nullptr);
        }
        in_a = !in_a;
        float *t = in; in = out; out = t;
    }
    __builtin_amdgcn_wave_barrier();
    if (in_a)
        fwd_layer<P, 2>(wa, Pm + L.out_b, 0.f, in, out, vi, V, nullptr, nullptr, yn);
    else
        fwd_layer<P, 2>(wb, Pm + L.out_b, 0.f, in, out, vi, V, nullptr, nullptr, yn);
    (void)lane;
}

// ------------------------------------------------------------------------------------------
// backward: input-gradient chain
// ------------------------------------------------------------------------------------------
template <int CINL>
__device__ __forceinline__ void dgrad_layer(const float (&wreg)[27], const float *dzb, float *dcur, int vi,
                                            bool accumulate) {
    const int lane = threadIdx.x & 63, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi, ntiles = (npos + 15) >> 4;
    for (int tile0 = 0; tile0 < ntiles; tile0 += 2) {
        const TileGeom g = tile_geom(tile0, vi, npos, SW, SC);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        tile_pair<3>(wreg, dzb, g.base, SW, SC, acc0, acc1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!g.ok[u]) continue;
            const f32x4 acc = u == 0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = 4 * kq + r;
                if (ci < CINL) {
                    const int i = ci * npos + g.pos[u];
                    dcur[i] = accumulate ? dcur[i] + acc[r] : acc[r];
                }
            }
        }
    }
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64) void txp_bwd_wave_kernel(const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const ModelLayout &L = a.lay;
    const int V = a.V, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = plane_slot(V);
    float *dzb = sm + wave * (slot + P * C * V), *dcur = dzb + slot;
    const int n = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave);
    if (n >= a.N) return;
    int vi = a.num_peds ? a.num_peds[n] : V;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    float *slope_row = a.slopes + (int64_t)n * L.n_txp;
    if (vi == 0) {
        for (int e = lane; e < L.n_txp; e += 64) slope_row[e] = 0.f;
        return;
    }
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi, ld = dz_stride(vi);
    const float *Pm = a.params;
    const float *wsn = a.ws + n * a.ws_stride;
    const float *dyn = a.dy + (int64_t)n * (C * P) * V;
    wave_zero(dzb, (P * SC) >> 2);
    float wr[27];
    load_w_bwd<P>(Pm + L.out_w, wr);
    __builtin_amdgcn_wave_barrier();
    for (int l = L.L; l >= 0; --l) {
        const bool is_out = l == L.L;
        constexpr int U = 4;
        float slope_acc = 0.f;
        if (is_out) {
            // dz of the output conv is dy: (C*P) rows of V floats -> plane interior
            for (int e0 = lane; e0 < P * npos; e0 += 64 * U) {
                float dv[U];
                int li[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + 64 * u, ec = e < P * npos ? e : 0;
                    const int row = ec / vi, w = ec - row * vi, ch = row / C, h = row - ch * C;
                    li[u] = ch * SC + (h + 1) * SW + (w + 1);
                    dv[u] = dyn[(int64_t)row * V + w];
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (e0 + 64 * u < P * npos) dzb[li[u]] = dv[u];
            }
        } else {
            // dz_l = d(a_{l+1}) * prelu'(z_l); z is position-major [pos][12]; dz also leaves for the
            // weight-gradient GEMM as [P][ld]
            const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
            float *dzo = a.dzg + ((int64_t)n * L.L + l) * dz_slot(V);
            const float alpha = Pm[L.prelus + l];
            for (int e0 = lane; e0 < P * npos; e0 += 64 * U) {
                float zv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + 64 * u;
                    zv[u] = e < P * npos ? zl[e] : 1.f;            // e = pos*12 + ch (contiguous read)
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + 64 * u;
                    if (e < P * npos) {
                        const int p = e / P, ch = e - p * P, h = p / vi, w = p - h * vi;
                        const float z = zv[u], d = dcur[ch * npos + p];
                        float dz = d;
                        if (!(z > 0.f)) {
                            dz = alpha * d;
                            slope_acc = fmaf(d, z, slope_acc);
                        }
                        dzb[ch * SC + (h + 1) * SW + (w + 1)] = dz;
                        dzo[ch * ld + p] = dz;
                    }
                }
            }
            slope_acc = wave_sum(slope_acc);
            if (lane == 0) slope_row[l] = slope_acc;
        }
        __builtin_amdgcn_wave_barrier();
        if (l == 0) {
            float w8[27];
            load_w_bwd<T>(Pm + L.txp_w[0], w8);
            dgrad_layer<T>(w8, dzb, dcur, vi, false);
        } else {
            dgrad_layer<P>(wr, dzb, dcur, vi, !is_out);
            if (l > 1) load_w_bwd<P>(Pm + L.txp_w[l - 1], wr);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // dead slopes (layers >= L) and the hand-off of d(a_0) = d(block output) [C][T][vi] to the st_gcn backward
    for (int e = L.L + lane; e < L.n_txp; e += 64) slope_row[e] = 0.f;
    float *dout = a.da0 + (int64_t)n * (C * T * V);
    for (int e = lane; e < C * T * vi; e += 64) dout[e] = dcur[e];
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int wave_wpb(size_t per_wave) {
    int wpb = 2;
    if (const char *e = getenv("STG_TXP_WPB")) {
        const int w = atoi(e);
        if (w == 1 || w == 2 || w == 4) wpb = w;
    }
    while (wpb > 1 && per_wave * wpb > (size_t)kLdsBytes) wpb >>= 1;
    return wpb;
}

bool txp_wave_fits(const ModelLayout &L, int V) {
    if (L.n_txp < 1 || L.n_blocks != 1) return false;
    if (const char *e = getenv("STG_NO_WAVE_PATH"))
        if (atoi(e)) return false;
    const size_t fwd = (size_t)2 * plane_slot(V) * sizeof(float);
    return fwd <= 48 * 1024;        // at least three waves per CU
}

int launch_txp_fwd_wave(const TxpFwdArgs &a, hipStream_t st) {
    const size_t per_wave = (size_t)2 * plane_slot(a.V) * sizeof(float);
    const int wpb = wave_wpb(per_wave);
    const size_t lds = per_wave * wpb;
    const dim3 grid((a.N + wpb - 1) / wpb);
#define STG_L(W)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_fwd_wave_kernel<W>),          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_wave: hipFuncSetAttribute");                       \
        hipLaunchKernelGGL(txp_fwd_wave_kernel<W>, grid, dim3(W * 64), lds, st, a);                           \
    } while (0)
    if (wpb == 4) STG_L(4); else if (wpb == 2) STG_L(2); else STG_L(1);
#undef STG_L
    STG_LAUNCH_CHECK("txp_fwd_wave");
    return STG_OK;
}

int launch_txp_bwd_wave(const TxpBwdArgs &a, hipStream_t st) {
    const size_t per_wave = ((size_t)plane_slot(a.V) + (size_t)P * C * a.V) * sizeof(float);
    const int wpb = wave_wpb(per_wave);
    const size_t lds = per_wave * wpb;
    const dim3 grid((a.N + wpb - 1) / wpb);
#define STG_L(W)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_bwd_wave_kernel<W>),          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_wave: hipFuncSetAttribute");                       \
        hipLaunchKernelGGL(txp_bwd_wave_kernel<W>, grid, dim3(W * 64), lds, st, a);                           \
    } while (0)
    if (wpb == 4) STG_L(4); else if (wpb == 2) STG_L(2); else STG_L(1);
#undef STG_L
    STG_LAUNCH_CHECK("txp_bwd_wave");
    return STG_OK;
}

}  // namespace stg
