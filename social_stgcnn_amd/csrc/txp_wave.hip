// txp_wave: the TXP-CNN (model.py:187-195) forward and its input-gradient chain as WAVE-PER-SCENE
// kernels -- the fast path for scenes whose activation plane(s) fit a wave-private LDS image (V <= 68).
// One wave64 owns one scene-window: no workgroup barriers, no cross-wave imbalance; eight independent waves per CU
// (two per SIMD) keep the four matrix pipes fed while each wave alternates between its MFMA stream and its own
// LDS / HBM traffic.  The forward keeps ONE plane and updates it in place (txp_sci, model_common.hpp); the
// input-gradient chain holds a dz plane and the running input gradient.
//
// Every 3x3 conv over the (5, V) plane is an implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32):
//   forward   out[co][pos] = b[co] + sum_{tap,ci} W[co][ci][tap] in[ci][pos+tap]
//             M = 12 out-channels (16-row tile), N = 16 positions, K = (tap, 4 channels)
//   backward  d in[ci][pos] = sum_{tap,co} W[co][ci][tap] dz[co][pos-tap]   (same GEMM, W^T, flipped taps)
// The B operand is one ds_read_b32 per MFMA from a zero-bordered channel-major plane whose channel
// stride is == 16 (mod 32) dwords; weights sit in VGPRs for the whole layer (the next layer's are
// prefetched while the current layer computes).  Epilogues write position-major [pos][12] images with
// 16-byte stores: the layout the weight-gradient GEMM (model_bwd.hip, K2) stages back with LDS-DMA.
#include "txp_scene_common.hpp"
#include "nll_elem.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P, T = Cfg::T;

// A lane's weights of one (co, ci) pair are 9 consecutive floats (the taps): two 16-byte loads + one dword per pair
// instead of nine scattered dwords (every lane reads a different cache line, so the request count is what costs)
struct __attribute__((packed, aligned(4))) F4U {
    float v[4];
};
__device__ __forceinline__ void load_taps(const float *__restrict__ p, float (&t)[9]) {
    const F4U a = *reinterpret_cast<const F4U *>(p), b = *reinterpret_cast<const F4U *>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        t[i] = a.v[i];
        t[4 + i] = b.v[i];
    }
    t[8] = p[8];
}

// forward A operand: lane (co = l&15, kq = l>>4) of K-step (tap, j) holds W[co][4j+kq][tap]
template <int CINL>
__device__ __forceinline__ void load_w_fwd(const float *__restrict__ W, float (&wreg)[CINL * 9 / 4]) {
    const int lane = threadIdx.x & 63, co = lane & 15, kq = lane >> 4;
    const int cc = co < P ? co : 0;                   // rows 12..15 of the tile are zero
#pragma unroll
    for (int j = 0; j < CINL / 4; ++j) {
        float t[9];
        load_taps(W + (cc * CINL + 4 * j + kq) * 9, t);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wreg[tap * (CINL / 4) + j] = co < P ? t[tap] : 0.f;
    }
}

// input-gradient A operand: lane (ci = l&15, kq) of K-step (tap', j) holds W[4j+kq][ci][8 - tap']
template <int CINL>
__device__ __forceinline__ void load_w_bwd(const float *__restrict__ W, float (&wreg)[27]) {
    const int lane = threadIdx.x & 63, ci = lane & 15, kq = lane >> 4;
    const int cc = ci < CINL ? ci : 0;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float t[9];
        load_taps(W + ((4 * j + kq) * CINL + cc) * 9, t);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) wreg[tap * 3 + j] = ci < CINL ? t[8 - tap] : 0.f;
    }
}

// ---- tile loop ------------------------------------------------------------------------------------
// A scene's positions are walked in pairs of 16-position tiles: all 18*KJ im2col reads of the pair are
// issued (kw offsets as ds_read immediates), then the 18*KJ MFMAs on two independent accumulators.
struct TileGeom {
    int hh[2], ww[2], pos[2];
    bool ok[2];
};

__device__ __forceinline__ TileGeom tile_geom(int tile0, const ptab_t *ptab, int npos) {
    const int nq = threadIdx.x & 15;
    TileGeom g;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int p = (tile0 + u) * 16 + nq;
        g.ok[u] = p < npos;
        g.pos[u] = g.ok[u] ? p : 0;
        const unsigned hw = ptab[g.pos[u]];
        g.hh[u] = (int)(hw >> 8);
        g.ww[u] = (int)(hw & 0xffu);
    }
    return g;
}

template <int KJ>
struct BRegs {
    float v[2][9 * KJ];
};

template <int KJ>
__device__ __forceinline__ void load_b(const float *plane, const TileGeom &g, int SW, int SC,
                                       BRegs<KJ> &b) {
    const int kq = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float *q0 = plane + kq * SC + g.hh[u] * SW + g.ww[u];
#pragma unroll
        for (int j = 0; j < KJ; ++j)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float *q = q0 + 4 * j * SC + kh * SW;      // kw = 0,1,2 become immediate offsets
                b.v[u][(kh * 3 + 0) * KJ + j] = q[0];
                b.v[u][(kh * 3 + 1) * KJ + j] = q[1];
                b.v[u][(kh * 3 + 2) * KJ + j] = q[2];
            }
    }
}

template <int KJ>
// Two accumulators per tile (even / odd K steps, added at the end): four independent MFMA chains per pair, so a
// dependent MFMA never issues back to back behind its producer whatever order the scheduler picks.
__device__ __forceinline__ void mma_pair(const float (&wreg)[9 * KJ], const BRegs<KJ> &b, f32x4 &acc0, f32x4 &acc1) {
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9 * KJ; ++k) {
        if (k & 1) {
            o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[k], b.v[0][k], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[k], b.v[1][k], o1, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[k], b.v[0][k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[k], b.v[1][k], acc1, 0, 0, 0);
        }
    }
    acc0 += o0;
    acc1 += o1;
}

// epi(geom, u, acc) finishes tile u of a pair.  (A software-pipelined form -- next pair's operands fetched
// ahead, previous pair's epilogue deferred -- measured no faster and cost 100 VGPRs: tools/micro/conv_tile_bench.)
// REV walks the tile pairs from the last position down and finishes the upper tile of a pair first (the in-place
// forward's odd layers: their writes land two rows ABOVE the rows still to be read).
template <int KJ, bool REV = false, typename Epi>
__device__ __forceinline__ void conv_tiles(const float (&wreg)[9 * KJ], const f32x4 binit,
                                           const float *plane, const ptab_t *ptab, int npos, int SW,
                                           int SC, Epi epi) {
    const int ntiles = (npos + 15) >> 4;
    const int npairs = (ntiles + 1) >> 1;
    for (int pr = 0; pr < npairs; ++pr) {
        const int tile0 = 2 * (REV ? npairs - 1 - pr : pr);
        const TileGeom g = tile_geom(tile0, ptab, npos);
        BRegs<KJ> b;
        load_b<KJ>(plane, g, SW, SC, b);
        f32x4 c0 = binit, c1 = binit;
        mma_pair<KJ>(wreg, b, c0, c1);
        if (REV) {
            epi(g, 1, c1);
            epi(g, 0, c0);
        } else {
            epi(g, 0, c0);
            epi(g, 1, c1);
        }
    }
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
// `in` / `out` point at padded row 0 of the layer's input / output inside the ONE in-place plane (channel stride
// SC = txp_sci(vi)); they differ by two row slots and alias, hence no __restrict__.
template <int CINL, int KIND, bool REV>
__device__ __forceinline__ void fwd_layer(const float (&wreg)[CINL * 9 / 4], const float *__restrict__ bias, float alpha,
                                          const float *in, float *out, const ptab_t *ptab,
                                          int vi, int V, float *zsave, float *psave, float *yout, bool bf16 = false) {
    const int kq = (threadIdx.x & 63) >> 4;
    const int SW = txp_sw(vi), SC = txp_sci(vi), npos = C * vi;
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = kq < 3 ? bias[4 * kq + r] : 0.f;
    conv_tiles<CINL / 4, REV>(wreg, binit, in, ptab, npos, SW, SC, [&](const TileGeom &g, int u, const f32x4 &z) {
        if (!g.ok[u] || kq == 3) return;
        if (KIND == 2) {
            // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
#pragma unroll
            for (int r = 0; r < 4; ++r) yout[(int64_t)((4 * kq + r) * C + g.hh[u]) * V + g.ww[u]] = z[r];
        } else {
            const int pp = (g.hh[u] + 1) * SW + (g.ww[u] + 1);
            f32x4 av;
            // the residual inputs first, all four in flight: `in` and `out` are one buffer (two row slots apart), so
            // the compiler would otherwise order every read behind the previous write (read, wait, write, read, ...)
            float res[4] = {0.f, 0.f, 0.f, 0.f};
            if (KIND == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) res[r] = in[(4 * kq + r) * SC + pp];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = (z[r] > 0.f ? z[r] : alpha * z[r]) + res[r];
                av[r] = v;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(4 * kq + r) * SC + pp] = av[r];
            if (zsave) {   // position-major [pos][12]: one 16-byte store per lane
                store_vec4(zsave, g.pos[u] * 3 + kq, z, bf16);
                store_vec4(psave, (g.hh[u] * save_sw(vi, bf16) + g.ww[u] + 1) * 3 + kq, av, bf16);   // rows 1..C of the padded plane
            }
        }
    });
}

// Scenes are dealt round-robin to the persistent waves (scene = wave id, + number of waves, ...).  A
// device-scope work queue (one atomicAdd per scene) was tried and was SLOWER: ~12k dequeues per launch
// saturate one word (~88 dequeues/us, MI355X_MICROARCH price list) and put 1-3 us of latency in front of
// every scene.

// diagnostic stamps (never read by the kernel; only with STG_STAMPS=1)
#ifdef STG_DIAG
#define STG_STAMP(k)                                                                         \
    do {                                                                                     \
        if (a.stamps && (threadIdx.x & 63) == 0) a.stamps[(int64_t)n * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STG_STAMP(k) do { } while (0)
#endif

// zero one row slot (SW floats) of every channel of the in-place plane
__device__ __forceinline__ void zero_row_slot(float *buf, int slot_row, int SW, int SC) {
    const int lane = threadIdx.x & 63;
    float *row = buf + slot_row * SW;
#pragma unroll
    for (int ch = 0; ch < P; ++ch)
        for (int c = lane; c < SW; c += 64) row[ch * SC + c] = 0.f;
}

// One scene-window, whole model.  `buf` = the wave's LDS region: the in-place TXP plane image [P][txp_sci(vi)], which
// during the st_gcn block phase holds the block's arrays instead -- G [C][T][vi] at the start, the block input X
// [CIN0][T][vi] behind it, H [C][T][vi] at the END of the image (96 vi <= P txp_sci(vi) - 40 vi always).  The block
// forms its outputs in registers, zeroes the image and scatters a_0 into it (stgcn_block_fwd, wave mode): the block
// output never visits HBM on its way to the TXP-CNN.
__device__ __forceinline__ void txp_fwd_scene(const TxpFwdArgs &a, const float *__restrict__ params,
                                              const float *blk_params, const float *blk_buffers, int n, float *buf,
                                              ptab_t *ptab) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63;
    STG_STAMP(0);
    int vi = a.num_peds ? a.num_peds[n] : V;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    float *yn = a.y + (int64_t)n * (C * P) * V;
    if (vi < V)                                        // padded pedestrian slots of the output are zeros
        for (int e = lane; e < C * P * (V - vi); e += 64) {
            const int r = e / (V - vi), w = vi + (e - r * (V - vi));
            yn[(int64_t)r * V + w] = 0.f;
        }
    if (vi == 0) return;
    const int SW = txp_sw(vi), SC = txp_sci(vi);
    const float *Pm = params;
    const bool bf16 = (L.flags & STG_OPT_BF16_STORE) != 0;        // bf16 storage of the saved planes / pre-activations
    float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;
    float *statn = a.stats ? a.stats + (int64_t)n * L.stat_floats : nullptr;

    // ---- st_gcn block (model.py:145-155) ------------------------------------------------------------
    {
        const float *agn = a.agg + n * a.agg_stride;
        if (vi <= 64 && !STG_SKIP(a, 256)) {
            // column mode: lane = pedestrian, everything in registers, outputs straight into the plane image
            stgcn_block_fwd_cols(a, blk_params, blk_buffers, L.blk[0], n, vi, wsn, statn, agn + a.agg_ax, agn + a.agg_cs,
                                 buf + 2 * SW, SC, buf, (P * SC) >> 2, ptab);
        } else {
            build_ptab(ptab, vi);
            float *G = buf, *X = buf + C * T * vi, *H = buf + P * SC - C * T * vi;
            const float *xn = a.x + n * a.x_sn;
            for (int e = lane; e < Cfg::CIN0 * T * vi; e += 64) {      // strided: the caller's permute(0,3,1,2) view
                const int v = e % vi, ct = e / vi, t = ct % T, c = ct / T;
                X[e] = xn[c * a.x_sc + t * a.x_st + v * a.x_sv];
            }
            __builtin_amdgcn_wave_barrier();
            STG_STAMP(9);
            stgcn_block_fwd<Cfg::CIN0, 0>(a, blk_params, blk_buffers, L.blk[0], n, vi, X, G, H, nullptr, wsn, statn,
                                          agn + a.agg_ax, agn + a.agg_cs, true, buf + 2 * SW, SC, buf, (P * SC) >> 2,
                                          nullptr, false, ptab);
        }
    }
    STG_STAMP(1);
    // a_0 now sits in the in-place layout (T channels, rows at slot offset 2, zeros elsewhere)
    float w0[T * 9 / 4];
    load_w_fwd<T>(Pm + L.txp_w[0], w0);
    if (wsn) {
        // training: a_0 is saved position-major for the weight-gradient GEMM -- the C interior ROWS with their two
        // border columns, [C*SW][P] (16-byte stores); the saved planes a_1 .. a_L get their zero border columns here
        float *d2 = wsn + ws_plane_off(L, V, 0);
        const int SWs = save_sw(vi, bf16);                         // row stride of the saved planes
        for (int h = 0; h < C; ++h)
            for (int e = lane; e < SW * 3; e += 64) {              // (position, channel quad)
                const int col = e / 3, q = e - col * 3;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < T / 4) {
                    const float *src = buf + (4 * q) * SC + (h + 3) * SW + col;
                    v = make_float4(src[0], src[SC], src[2 * SC], src[3 * SC]);
                }
                store_vec4(d2, (h * SWs + col) * 3 + q, f32x4{v.x, v.y, v.z, v.w}, bf16);
            }
        if (lane < 2 * C * 3) {
            const int b = lane / 3, q = lane - b * 3, pos = (b >> 1) * SWs + ((b & 1) ? SW - 1 : 0);
            for (int l = 0; l < L.L; ++l)
                store_vec4(wsn + ws_plane_off(L, V, l + 1), pos * 3 + q, f32x4{0.f, 0.f, 0.f, 0.f}, bf16);
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- TXP-CNN (model.py:187-195) -------------------------------------------------------------------
    float *hi = buf + 2 * SW, *lo = buf;            // padded row 0 of the two positions of the plane
    float wa[27], wb[27];          // two weight register sets: layer l computes from one while l+1 loads
    auto w_of = [&](int l) { return Pm + (l < L.L ? L.txp_w[l] : L.out_w); };
    auto zs_of = [&](int l) { return wsn ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr; };
    auto ps_of = [&](int l) { return wsn ? wsn + ws_plane_off(L, V, l + 1) : nullptr; };
    // layer 0 (weights w0 already resident): hi -> lo, upwards.  The next layer's weights are fetched AFTER a
    // layer's tile loop, not during it: 27 more live registers made the compiler serialise the loop's LDS reads
    // (one `s_waitcnt lgkmcnt(0)` per read) instead of batching them
    {
        float *zs = zs_of(0), *ps = ps_of(0);
        fwd_layer<T, 0, false>(w0, Pm + L.txp_b[0], Pm[L.prelus], hi, lo, ptab, vi, V, zs, ps, nullptr, bf16);
        load_w_fwd<P>(w_of(1), wa);
        __builtin_amdgcn_wave_barrier();
        zero_row_slot(buf, C + 1, SW, SC);           // lo's bottom border held hi's padded row C - 1
    }
    STG_STAMP(2);
    int l = 1;
    bool in_a = true;              // which register set holds layer l's weights
    for (; l < L.L; ++l) {
        float *zs = zs_of(l), *ps = ps_of(l);
        const bool odd = l & 1;    // odd layers: lo -> hi, downwards; even layers: hi -> lo, upwards
        __builtin_amdgcn_wave_barrier();
        auto run = [&](const float (&wr)[27]) {      // (register arrays: selected statically, never by reference)
            if (odd)
                fwd_layer<P, 1, true>(wr, Pm + L.txp_b[l], Pm[L.prelus + l], lo, hi, ptab, vi, V, zs, ps, nullptr, bf16);
            else
                fwd_layer<P, 1, false>(wr, Pm + L.txp_b[l], Pm[L.prelus + l], hi, lo, ptab, vi, V, zs, ps, nullptr, bf16);
        };
        if (l == 1) STG_STAMP(12);
        if (in_a) {
            run(wa);
            if (l == 1) STG_STAMP(9);
            load_w_fwd<P>(w_of(l + 1), wb);
        } else {
            run(wb);
            if (l == 1) STG_STAMP(9);
            load_w_fwd<P>(w_of(l + 1), wa);
        }
        if (l == 1) STG_STAMP(10);
        __builtin_amdgcn_wave_barrier();
        zero_row_slot(buf, odd ? 2 : C + 1, SW, SC);   // hi's top border held lo's padded row 2 / see layer 0
        if (l == 1) STG_STAMP(11);
        in_a = !in_a;
        STG_STAMP(2 + l);
    }
    __builtin_amdgcn_wave_barrier();
    if (in_a)
        fwd_layer<P, 2, false>(wa, Pm + L.out_b, 0.f, (l & 1) ? lo : hi, nullptr, ptab, vi, V, nullptr, nullptr, yn);
    else
        fwd_layer<P, 2, false>(wb, Pm + L.out_b, 0.f, (l & 1) ? lo : hi, nullptr, ptab, vi, V, nullptr, nullptr, yn);
    STG_STAMP(8);
}

template <int WPB>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_wave_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    // params / buffers are separate __restrict__ kernel arguments on purpose: only then can the compiler prove that
    // the kernel's own stores never clobber them and fetch the (wave-uniform) st_gcn weights with scalar loads
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int slot = P * txp_sci(Vl);
    const int per_wave = slot + ptab_floats(Vl);
    float *pa = sm + wave * per_wave;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(pa + slot);
    float *blk_p = sm + WPB * per_wave, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, WPB * 64);
    stagger_start(wave, WPB, a.stagger);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? a.tier.order[begin + it] : it);
        txp_fwd_scene(a, params, blk_p, blk_b, n, pa, ptab);
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// backward: input-gradient chain
// ------------------------------------------------------------------------------------------
template <int CINL>
__device__ __forceinline__ void dgrad_layer(const float (&wreg)[27], const float *__restrict__ dzb,
                                            float *__restrict__ dcur, const ptab_t *ptab, int vi, bool accumulate) {
    const int kq = (threadIdx.x & 63) >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    conv_tiles<3>(wreg, zero, dzb, ptab, npos, SW, SC, [&](const TileGeom &g, int u, const f32x4 &acc) {
        if (!g.ok[u] || 4 * kq >= CINL) return;
        // the running input gradient is position-major [pos][P]: a lane's four channels are one 16-byte LDS access
        f32x4 *slot = reinterpret_cast<f32x4 *>(dcur + g.pos[u] * P + 4 * kq);
        f32x4 v = acc;
        if (accumulate) v += *slot;
        *slot = v;
    });
    (void)npos;
}

// Tail of the per-scene backward: the st_gcn block (model.py:145-155 backwards) in wave mode.  The input-gradient
// chain left d(a_0) position-major in `dcur` ([pos][P], channels 0..T-1); the dz plane is dead, so the block's three
// LDS arrays are carved from the start of the wave's region: D = d(block output) [C][T][vi] | h1 [C][T+2][vi] | dh2
// [C][T+2][vi] (140 vi floats <= plane_slot + 60 V); db1 reuses D.  Small-parameter gradients leave as the scene's own
// row (stores, no atomics): reduce_slabs_kernel sums the rows in a fixed order.
__device__ __forceinline__ void txp_bwd_block_tail(const TxpBwdArgs &a, const float *blk_params, int n, int vi,
                                                   float *dzb, float *dcur, ptab_t *ptab, float *tot, float *slope_row,
                                                   bool d_ready = false) {
    const ModelLayout &L = a.lay;
    const int lane = threadIdx.x & 63;
    for (int e = L.L + lane; e < L.n_txp; e += 64) slope_row[e] = 0.f;       // dead slopes (layers >= L)
    if (STG_SKIP(a, 4)) return;
    float *D = dzb, *H1 = dzb + C * T * vi, *DH2 = H1 + C * (T + 2) * vi;
    // v.view(N, T, C, V) (model.py:187) backwards: plane (ch, row) is flat f = ch*C + row = c*T + t of the block output
    if (!d_ready)
        for (int f = 0; f < C * T; ++f) {
            const int ch = f / C, row = f - ch * C;
            for (int w = lane; w < vi; w += 64) D[f * vi + w] = dcur[(row * vi + w) * P + ch];
        }
    __builtin_amdgcn_wave_barrier();
    float *row = slope_row - L.n_blk_params;
    if (vi <= 64 && !STG_SKIP(a, 8192)) {
        // column mode: lane = pedestrian, all 8 time steps in registers (the mirror of the forward's column mode)
        if (vi <= 32) stgcn_block_bwd_cols<true>(a, blk_params, L.blk[0], n, vi, D, row, a.ws + n * a.ws_stride);
        else stgcn_block_bwd_cols<false>(a, blk_params, L.blk[0], n, vi, D, row, a.ws + n * a.ws_stride);
        return;
    }
    stgcn_block_bwd<Cfg::CIN0, 0, false>(a, blk_params, L.blk[0], n, vi, D, H1, DH2, D, nullptr, tot, row,
                                             a.ws + n * a.ws_stride, nullptr, nullptr, nullptr, nullptr, ptab);
}

__device__ __forceinline__ void txp_bwd_scene(const TxpBwdArgs &a, const float *blk_params, int n, float *dzb, float *dcur,
                                              ptab_t *ptab, float *tot) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63;
    int vi = a.num_peds ? a.num_peds[n] : V;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    float *slope_row = a.rows + (int64_t)n * (L.n_blk_params + L.n_txp) + L.n_blk_params;
    if (vi == 0) {                                     // empty scene: its row of small-parameter gradients is zero
        for (int e = lane; e < L.n_blk_params + L.n_txp; e += 64) slope_row[e - L.n_blk_params] = 0.f;
        if (a.nll_target && lane == 0) a.nll_losses[n] = 0.f;
        return;
    }
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi;
    const float *Pm = a.params;
    const bool bf16 = (L.flags & STG_OPT_BF16_STORE) != 0;        // bf16 storage of z_l (read) and dz_l (written)
    const float *wsn = a.ws + n * a.ws_stride;
    const float *dyn = a.dy + (int64_t)n * (C * P) * V;
    wave_zero(dzb, (P * SC) >> 2);
    build_ptab(ptab, vi);
    float wr[27];
    load_w_bwd<P>(Pm + L.out_w, wr);
    __builtin_amdgcn_wave_barrier();
    for (int l = L.L; l >= 0; --l) {
        const bool is_out = l == L.L;
        constexpr int U = 4;
        float slope_acc = 0.f;
        if (STG_SKIP(a, 512)) {
        } else if (is_out) {
            float *dzo_out = a.dzg + ((int64_t)n * (L.L + 1) + L.L) * dz_slot(V);
            // dz of the output conv is dy: (C*P) rows of V floats, vi valid -> plane interior.  Lanes are laid over
            // (sub-row, w) with the row length rounded up to a power of two: no division by the runtime vi
            // (it cost ~50 instructions per element, 30 elements per lane)
            const int vp = vi <= 1 ? 1 : (vi <= 2 ? 2 : (vi <= 4 ? 4 : (vi <= 8 ? 8 : (vi <= 16 ? 16 : (vi <= 32 ? 32 : 64)))));
            const int sh = __builtin_ctz(vp), rpi = 64 >> sh;       // rows per 64-lane pass
            const int sub = lane >> sh, w0 = lane & (vp - 1);
            if (a.nll_target) {
                // fused loss: a.dy is V_pred (N, 5, P, V); lanes are laid over (prediction step p, pedestrian w); the
                // five gradients of (p, w) are rows f * P + p of the (C*P) x V array the chain starts from
                const float *tn = a.nll_target + (int64_t)n * P * V * 2;
                const float inv_cnt = 1.0f / (float)(P * vi);
                const float gs = inv_cnt * (a.nll_weights ? a.nll_weights[n] : 1.f);
                float lacc = 0.f;
                for (int wb = 0; wb < vi; wb += 64) {
                    const int w = wb + w0;
                    const bool okw = w < vi;
                    for (int p0 = 0; p0 < P; p0 += rpi) {
                        const int p = p0 + sub;
                        if (okw && p < P) {
                            const float *q = dyn + (int64_t)p * V + w;
                            const float2 tg = *reinterpret_cast<const float2 *>(tn + ((int64_t)p * V + w) * 2);
                            float g[5];
                            lacc += nll_elem(q[0], q[(int64_t)P * V], q[(int64_t)2 * P * V], q[(int64_t)3 * P * V],
                                             q[(int64_t)4 * P * V], tg.x, tg.y, true, g);
#pragma unroll
                            for (int f = 0; f < C; ++f) {
                                const int rc = f * P + p, ch = rc / C, h = rc - ch * C;
                                dzb[ch * SC + (h + 1) * SW + (w + 1)] = g[f] * gs;
                            }
                        }
                    }
                }
                lacc = wave_sum(lacc);
                if (lane == 0) a.nll_losses[n] = lacc * inv_cnt;
            } else
            for (int wb = 0; wb < vi; wb += 64) {                   // (vi > 64: a second column block)
                const int w = wb + w0;
                const bool okw = w < vi;
                for (int r0 = 0; r0 < C * P; r0 += rpi * U) {
                    float dv[U];
                    int li[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        const bool ok = okw && row < C * P;
                        const int rc = ok ? row : 0, ch = rc / C, h = rc - ch * C;
                        li[u] = ok ? ch * SC + (h + 1) * SW + (w + 1) : -1;
                        dv[u] = ok ? dyn[(int64_t)rc * V + w] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (li[u] >= 0) dzb[li[u]] = dv[u];
                }
            }
            // position-major copy of this dz (= dy) for the weight-gradient GEMM: one 16-byte store per (position,
            // channel quad), gathered from the plane just built
            __builtin_amdgcn_wave_barrier();
            for (int vv = lane; vv < (P * npos) >> 2; vv += 64) {
                const int p = vv / 3, q = vv - p * 3;
                const unsigned hw = ptab[p];
                const float *src = dzb + (4 * q) * SC + ((int)(hw >> 8) + 1) * SW + ((int)(hw & 0xffu) + 1);
                store_vec4(dzo_out, ((int)(hw >> 8) * save_vw(vi, bf16) + (int)(hw & 0xffu)) * 3 + q,
                           f32x4{src[0], src[SC], src[2 * SC], src[3 * SC]}, bf16);
            }
        } else {
            // dz_l = d(a_{l+1}) * prelu'(z_l); z is position-major [pos][12]; dz also leaves for the
            // weight-gradient GEMM, position-major as well (one 16-byte store per lane)
            const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
            float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + l) * dz_slot(V);
            const float alpha = Pm[L.prelus + l];
            // z_l is position-major [pos][12]: 3 x 16 bytes per position; 8 vectors in flight per lane so the
            // HBM latency is paid once per layer, not once per element batch
            constexpr int UV = 8;
            const int nvec = (P * npos) >> 2;                       // P*npos is a multiple of 4
            for (int v0 = lane; v0 < nvec; v0 += 64 * UV) {
                f32x4 zv[UV];
#pragma unroll
                for (int u = 0; u < UV; ++u) {
                    const int vv = v0 + 64 * u;
                    zv[u] = vv < nvec ? load_vec4(zl, vv, bf16) : f32x4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int u = 0; u < UV; ++u) {
                    const int vv = v0 + 64 * u;
                    if (vv < nvec) {
                        const int p = vv / 3, q = vv - p * 3;       // position, channel quad
                        const unsigned hw = ptab[p];
                        const int h = (int)(hw >> 8), w = (int)(hw & 0xffu);
                        f32x4 dzv;
                        const f32x4 dv = reinterpret_cast<const f32x4 *>(dcur)[vv];   // [pos][P]: vector vv = (p, q)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ch = 4 * q + r;
                            const float z = zv[u][r], d = dv[r];
                            float dz = d;
                            if (!(z > 0.f)) {
                                dz = alpha * d;
                                slope_acc = fmaf(d, z, slope_acc);
                            }
                            dzb[ch * SC + (h + 1) * SW + (w + 1)] = dz;
                            dzv[r] = dz;
                        }
                        store_vec4(dzo, (h * save_vw(vi, bf16) + w) * 3 + q, dzv, bf16);   // (= vv in fp32: rows of vi)
                    }
                }
            }
            slope_acc = wave_sum(slope_acc);
            if (lane == 0) slope_row[l] = slope_acc;
        }
        __builtin_amdgcn_wave_barrier();
        if (STG_SKIP(a, 1024)) {
        } else if (l == 0) {
            float w8[27];
            load_w_bwd<T>(Pm + L.txp_w[0], w8);
            dgrad_layer<T>(w8, dzb, dcur, ptab, vi, false);
        } else {
            dgrad_layer<P>(wr, dzb, dcur, ptab, vi, !is_out);
            if (l > 1) load_w_bwd<P>(Pm + L.txp_w[l - 1], wr);
        }
        __builtin_amdgcn_wave_barrier();
    }
    txp_bwd_block_tail(a, blk_params, n, vi, dzb, dcur, ptab, tot, slope_row);
}

// ------------------------------------------------------------------------------------------
// backward, split-bf16 variant: the input-gradient GEMMs on v_mfma_f32_16x16x16_bf16
// ------------------------------------------------------------------------------------------
// Every fp32 operand is split x = hi + lo (two bf16, round-to-nearest-even) and a product is taken as
// hi*hi + hi*lo + lo*hi with fp32 accumulation: error ~1e-6 of sum|ab| (tools/micro/bf16x3_probe.hip) at 2.6x the
// rate of the fp32 MFMA.  K = 16 is ONE tap x 12 output channels (+ 4 zero lanes).  The dz plane is position-major:
// [padded position][12] bf16 for hi, the same for lo behind it -- exactly the bytes of the fp32 plane -- so a lane's
// B operand (four consecutive channels at one position) is one 8-byte LDS read per part.
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_val(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split4(const float (&x)[4], s16x4 &hi, s16x4 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned short h = bf16_rne(x[j]);
        hi[j] = (short)h;
        lo[j] = (short)bf16_rne(x[j] - bf16_val(h));
    }
}

// A operand of tap' for lane (ci = l&15, kq): W[4kq+j][ci][8 - tap'], j = 0..3, split into hi / lo
template <int CINL>
__device__ __forceinline__ void load_w_bwd_bf16(const float *__restrict__ W, s16x4 (&whi)[9], s16x4 (&wlo)[9]) {
    const int lane = threadIdx.x & 63, ci = lane & 15, kq = lane >> 4;
    const bool live = ci < CINL && kq < 3;
    const int cc = ci < CINL ? ci : 0, kc = kq < 3 ? kq : 0;
    float t[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j) load_taps(W + ((4 * kc + j) * CINL + cc) * 9, t[j]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = live ? t[j][8 - tap] : 0.f;
        split4(x, whi[tap], wlo[tap]);
    }
}

template <int CINL>
__device__ __forceinline__ void dgrad_layer_bf16(const s16x4 (&whi)[9], const s16x4 (&wlo)[9], const char *hi,
                                                 const char *lo, float *__restrict__ dcur, const ptab_t *ptab, int vi,
                                                 bool accumulate) {
    const int kq = (threadIdx.x & 63) >> 4, kc = kq < 3 ? kq : 2;       // lanes kq = 3: finite data x zero weights
    const int SW = txp_sw(vi), npos = C * vi;
    const int ntiles = (npos + 15) >> 4;
    for (int tile0 = 0; tile0 < ntiles; tile0 += 2) {
        const TileGeom g = tile_geom(tile0, ptab, npos);
        s16x4 bh[2][9], bl[2][9];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int base = (g.hh[u] * SW + g.ww[u]) * 24 + kc * 8;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int off = base + (kh * SW + kw) * 24;
                    bh[u][kh * 3 + kw] = *reinterpret_cast<const s16x4 *>(hi + off);
                    bl[u][kh * 3 + kw] = *reinterpret_cast<const s16x4 *>(lo + off);
                }
        }
        f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo[tap], bh[0][tap], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wlo[tap], bh[1][tap], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi[tap], bl[0][tap], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi[tap], bl[1][tap], c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi[tap], bh[0][tap], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(whi[tap], bh[1][tap], c1, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!g.ok[u] || 4 * kq >= CINL) continue;
            f32x4 *slot = reinterpret_cast<f32x4 *>(dcur + g.pos[u] * P + 4 * kq);
            f32x4 v = u ? c1 : c0;
            if (accumulate) v += *slot;
            *slot = v;
        }
    }
}

__device__ __forceinline__ void txp_bwd_scene_bf16(const TxpBwdArgs &a, const float *blk_params, int n, float *dzb,
                                                   float *dcur, ptab_t *ptab, float *tot) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63;
    int vi = a.num_peds ? a.num_peds[n] : V;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    float *slope_row = a.rows + (int64_t)n * (L.n_blk_params + L.n_txp) + L.n_blk_params;
    if (vi == 0) {                                     // empty scene: its row of small-parameter gradients is zero
        for (int e = lane; e < L.n_blk_params + L.n_txp; e += 64) slope_row[e - L.n_blk_params] = 0.f;
        return;
    }
    const int SW = txp_sw(vi), SC = txp_sc(vi), npos = C * vi, npad = (C + 2) * SW;
    const float *Pm = a.params;
    const float *wsn = a.ws + n * a.ws_stride;
    const float *dyn = a.dy + (int64_t)n * (C * P) * V;
    char *hi = reinterpret_cast<char *>(dzb), *lo = hi + npad * 24;          // 2 * npad * 24 B <= P * SC * 4 B
    wave_zero(dzb, (P * SC) >> 2);
    build_ptab(ptab, vi);
    s16x4 whi[9], wlo[9];
    load_w_bwd_bf16<P>(Pm + L.out_w, whi, wlo);
    __builtin_amdgcn_wave_barrier();
    for (int l = L.L; l >= 0; --l) {
        const bool is_out = l == L.L;
        float slope_acc = 0.f;
        if (is_out) {
            float *dzo_out = a.dzg + ((int64_t)n * (L.L + 1) + L.L) * dz_slot(V);
            // dz of the output conv is dy: (C*P) rows of V floats, vi valid -> split, position-major plane interior
            constexpr int U = 4;
            const int vp = vi <= 1 ? 1 : (vi <= 2 ? 2 : (vi <= 4 ? 4 : (vi <= 8 ? 8 : (vi <= 16 ? 16 : (vi <= 32 ? 32 : 64)))));
            const int sh = __builtin_ctz(vp), rpi = 64 >> sh;
            const int sub = lane >> sh, w0 = lane & (vp - 1);
            for (int wb = 0; wb < vi; wb += 64) {
                const int w = wb + w0;
                const bool okw = w < vi;
                for (int r0 = 0; r0 < C * P; r0 += rpi * U) {
                    float dv[U];
                    int bo[U], gi[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        const bool ok = okw && row < C * P;
                        const int rc = ok ? row : 0, ch = rc / C, h = rc - ch * C;
                        bo[u] = ok ? ((h + 1) * SW + (w + 1)) * 24 + ch * 2 : -1;
                        gi[u] = (h * vi + w) * P + ch;
                        dv[u] = ok ? dyn[(int64_t)rc * V + w] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (bo[u] >= 0) {
                            dzo_out[gi[u]] = dv[u];          // position-major copy for the weight-gradient GEMM
                            const unsigned short h16 = bf16_rne(dv[u]);
                            *reinterpret_cast<unsigned short *>(hi + bo[u]) = h16;
                            *reinterpret_cast<unsigned short *>(lo + bo[u]) = bf16_rne(dv[u] - bf16_val(h16));
                        }
                }
            }
        } else {
            const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
            float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + l) * dz_slot(V);
            const float alpha = Pm[L.prelus + l];
            constexpr int UV = 8;
            const int nvec = (P * npos) >> 2;
            const f32x4 *zl4 = reinterpret_cast<const f32x4 *>(zl);
            for (int v0 = lane; v0 < nvec; v0 += 64 * UV) {
                f32x4 zv[UV];
#pragma unroll
                for (int u = 0; u < UV; ++u) {
                    const int vv = v0 + 64 * u;
                    zv[u] = vv < nvec ? zl4[vv] : f32x4{1.f, 1.f, 1.f, 1.f};
                }
#pragma unroll
                for (int u = 0; u < UV; ++u) {
                    const int vv = v0 + 64 * u;
                    if (vv < nvec) {
                        const int p = vv / 3, q = vv - p * 3;
                        const unsigned hw = ptab[p];
                        const int h = (int)(hw >> 8), w = (int)(hw & 0xffu);
                        const f32x4 dv = reinterpret_cast<const f32x4 *>(dcur)[vv];
                        f32x4 dzv;
                        float x[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float z = zv[u][r], d = dv[r];
                            float dz = d;
                            if (!(z > 0.f)) {
                                dz = alpha * d;
                                slope_acc = fmaf(d, z, slope_acc);
                            }
                            dzv[r] = dz;
                            x[r] = dz;
                        }
                        s16x4 h4, l4;
                        split4(x, h4, l4);
                        const int bo = ((h + 1) * SW + (w + 1)) * 24 + q * 8;
                        *reinterpret_cast<s16x4 *>(hi + bo) = h4;
                        *reinterpret_cast<s16x4 *>(lo + bo) = l4;
                        reinterpret_cast<f32x4 *>(dzo)[vv] = dzv;
                    }
                }
            }
            slope_acc = wave_sum(slope_acc);
            if (lane == 0) slope_row[l] = slope_acc;
        }
        __builtin_amdgcn_wave_barrier();
        if (l == 0) {
            s16x4 w8h[9], w8l[9];
            load_w_bwd_bf16<T>(Pm + L.txp_w[0], w8h, w8l);
            dgrad_layer_bf16<T>(w8h, w8l, hi, lo, dcur, ptab, vi, false);
        } else {
            dgrad_layer_bf16<P>(whi, wlo, hi, lo, dcur, ptab, vi, !is_out);
            if (l > 1) load_w_bwd_bf16<P>(Pm + L.txp_w[l - 1], whi, wlo);
        }
        __builtin_amdgcn_wave_barrier();
    }
    txp_bwd_block_tail(a, blk_params, n, vi, dzb, dcur, ptab, tot, slope_row);
}

template <int WPB, bool BF16>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) void txp_bwd_wave_kernel(const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int slot = plane_slot(Vl);
    const int per_wave = slot + P * C * Vl + bwd_ptab_floats(Vl);
    float *dzb = sm + wave * per_wave, *dcur = dzb + slot;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(dcur + P * C * Vl);
    float *tot = dcur + P * C * Vl + ptab_floats(Vl);
    float *blk_p = sm + WPB * per_wave;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, WPB * 64);
    stagger_start(wave, WPB, a.stagger);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? a.tier.order[begin + it] : it);
        if (BF16) txp_bwd_scene_bf16(a, blk_p, n, dzb, dcur, ptab, tot);      // (separate instantiations: the variants
        else txp_bwd_scene(a, blk_p, n, dzb, dcur, ptab, tot);               //  do not share a register budget)
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// mixed-V launch: class assignment of a workgroup (see MixGeom)
// ------------------------------------------------------------------------------------------
struct MixSlot {
    int begin, end;        // this class's range of the sorted scene list
    int worker, nworkers;  // this wave's place among the class's active waves
    int region;            // LDS floats of this wave's private region
    int vc;                // largest crowd of the class (LDS geometry of its scenes)
    bool active;
};

__device__ __forceinline__ MixSlot mix_assign(const SceneTier &t, const MixGeom &g, int N, int V) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *ks = t.key_start;
    // summed crowd size (+ a per-scene constant) of the three classes, keys spread over the lanes
    float w0 = 0.f, w1 = 0.f, w2 = 0.f;
    for (int k = lane; k <= V; k += 64) {
        const int v = V - k;
        const float w = (float)(ks[k + 1] - ks[k]) * (float)(v + 4);
        if (v <= g.v_small) w0 += w;
        else if (v <= g.v_mid) w1 += w;
        else w2 += w;
    }
    w0 = wave_sum(w0);
    w1 = wave_sum(w1);
    w2 = wave_sum(w2);
    const int s0 = ks[V - g.v_small];                       // first scene with V_n <= v_small
    const int s1 = ks[V - (g.v_mid < V ? g.v_mid : V)];     // first scene with V_n <= v_mid
    // workgroup-time of a class = work / active waves per workgroup (4, 2, 1)
    const float t0 = w0 * 0.25f, t1 = w1 * 0.5f, t2 = w2;
    const float tt = t0 + t1 + t2;
    const int G = gridDim.x;
    int g2 = s1 > 0 ? (int)((float)G * t2 / tt) : 0;
    int g1 = s0 > s1 ? (int)((float)G * t1 / tt) : 0;
    if (s1 > 0 && g2 < 1) g2 = 1;
    if (s0 > s1 && g1 < 1) g1 = 1;
    if (g2 > s1) g2 = s1;                                   // never more workgroups than scenes
    if (g1 * 2 > s0 - s1 + 1) g1 = (s0 - s1 + 1) >> 1;
    int g0 = G - g1 - g2;
    while (g0 < 1 && s0 < N) {                              // the small class must keep a workgroup (G >= 4)
        if (g2 > 1 && g2 >= g1) --g2;
        else if (g1 > 1) --g1;
        else break;
        g0 = G - g1 - g2;
    }
    const int b = blockIdx.x;
    MixSlot m;
    int aw, j, gc;
    if (b < g2) { aw = 1; j = b; gc = g2; m.begin = 0; m.end = s1; m.vc = V; }
    else if (b < g2 + g1) { aw = 2; j = b - g2; gc = g1; m.begin = s1; m.end = s0; m.vc = g.v_mid < V ? g.v_mid : V; }
    else { aw = 4; j = b - g2 - g1; gc = g0; m.begin = s0; m.end = N; m.vc = g.v_small; }
    m.active = wave < aw;
    m.worker = j * aw + wave;
    m.nworkers = gc * aw;
    m.region = g.block_floats / aw;
    return m;
}

__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_wave_mixed_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *blk_p = sm + a.mix.block_floats, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, 256);
    const MixSlot m = mix_assign(a.tier, a.mix, a.N, a.V);
    if (!m.active) return;
    const int wave = threadIdx.x >> 6;
    const int slot = P * txp_sci(m.vc);                 // region = [in-place plane | ptab] of the class's largest scene
    float *pa = sm + wave * m.region;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(pa + slot);
    const int M = m.end - m.begin;
    for (int r = 0; r * m.nworkers < M; ++r) {
        const int it = walk_item(r, m.worker, m.nworkers, M, a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order[m.begin + it]);
        txp_fwd_scene(a, params, blk_p, blk_b, n, pa, ptab);
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool BF16>
__global__ __launch_bounds__(256, 2) void txp_bwd_wave_mixed_kernel(const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *blk_p = sm + a.mix.block_floats;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, 256);
    const MixSlot m = mix_assign(a.tier, a.mix, a.N, a.V);
    if (!m.active) return;
    const int wave = threadIdx.x >> 6;
    const int slot = plane_slot(m.vc);                  // region = [dz plane | dcur | ptab | totals] of the class's largest scene
    float *dzb = sm + wave * m.region, *dcur = dzb + slot;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(dcur + P * C * m.vc);
    float *tot = dcur + P * C * m.vc + ptab_floats(m.vc);
    const int M = m.end - m.begin;
    for (int r = 0; r * m.nworkers < M; ++r) {
        const int it = walk_item(r, m.worker, m.nworkers, M, a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order[m.begin + it]);
        if (BF16) txp_bwd_scene_bf16(a, blk_p, n, dzb, dcur, ptab, tot);      // (separate instantiations: the variants
        else txp_bwd_scene(a, blk_p, n, dzb, dcur, ptab, tot);               //  do not share a register budget)
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
bool txp_wave_fits(const ModelLayout &L, int V) {
    if (L.n_txp < 1 || L.n_blocks != 1 || L.blk[0].cin != Cfg::CIN0) return false;
    if (L.flags & STG_OPT_WG_PATH) return false;
    if (txp_fwd_x6_fits(L, V) && txp_bwd_x6_fits(L, V)) return true;      // the exact-bf16 kernels (teams of waves beyond 32)
    const size_t fwd = (size_t)2 * plane_slot(V) * sizeof(float);
    return fwd <= 48 * 1024;        // the fp32-MFMA wave kernels: at least three waves per CU
}

constexpr int kSmallBatch = 288;      // measured: 256 scenes 1.89 (workgroup kernels) vs 1.79 M/s (a wave per scene), 320 scenes 2.10 vs 2.19

bool use_wave_path(const ModelLayout &L, int N, int V, int *wg_waves) {
    if (wg_waves) *wg_waves = L.wg_waves;
    if (!txp_wave_fits(L, V)) return false;
    if (txp_fwd_x6_fits(L, V) && txp_bwd_x6_fits(L, V)) return true;     // (small batches: finer teams, see team_geom)
    if (N >= kSmallBatch || L.wg_waves != 0 || V > 40 || (L.flags & (STG_OPT_WAVE_PATH | STG_OPT_BF16_STORE))) return true;
    // small batch of small scenes: 2048 resident wave slots / N scenes, at most the 8 waves a scene's tiles can use
    if (wg_waves) *wg_waves = N <= 192 ? 8 : 4;
    return false;
}

static size_t fwd_per_wave_floats(int v) { return (size_t)P * txp_sci(v) + ptab_floats(v); }
static size_t bwd_per_wave_floats(int v) { return (size_t)plane_slot(v) + (size_t)P * C * v + bwd_ptab_floats(v); }
constexpr int kMixSmallV = 32;

// geometry of the mixed-V launch for per-wave footprint `pw(v)`; false when the padded V does not call for it
template <typename F>
static bool mix_geom(F pw, int V, bool sorted, MixGeom *g) {
    g->on = 0;
    if (diag_env("STG_NO_MIX", 0)) return false;
    if (!sorted || V <= kMixSmallV) return false;
    const size_t block = 4 * pw(kMixSmallV);
    if (pw(V) > block || block * sizeof(float) > (size_t)kLdsBytes) return false;
    int v_mid = kMixSmallV;
    while (v_mid < V && 2 * pw(v_mid + 1) <= block) ++v_mid;
    g->on = 1; g->v_small = kMixSmallV; g->v_mid = v_mid; g->block_floats = (int)block;
    return true;
}
static int mix_grid(size_t lds_bytes, int N) {
    int per_cu = (int)(kLdsBytes / lds_bytes);
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    int g = kNumCU * per_cu;
    const int need = (N + 3) / 4 + 2;
    if (g > need) g = need;
    return g < 4 ? 4 : g;
}

int launch_txp_fwd_wave(const TxpFwdArgs &a0, hipStream_t st) {
    TxpFwdArgs a = a0;
    if (a.wpf && txp_fwd_x6_fits(a.lay, a.V)) return launch_txp_fwd_x6(a, st);     // txp_x6.hip: one wave per scene, or teams
    if (mix_geom(fwd_per_wave_floats, a.V, a.tier.order && a.tier.key_start, &a.mix)) {
        const size_t lds = ((size_t)a.mix.block_floats + wave_param_floats(a.lay)) * sizeof(float);
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_fwd_wave_mixed_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_wave_mixed: hipFuncSetAttribute");
        hipLaunchKernelGGL(txp_fwd_wave_mixed_kernel, dim3(mix_grid(lds, a.N)), dim3(256), lds, st, a, a.params, a.buffers);
        STG_LAUNCH_CHECK("txp_fwd_wave_mixed");
        return STG_OK;
    }
    const size_t per_wave = fwd_per_wave_floats(a.Vl) * sizeof(float);
    const int wpb = wave_wpb(per_wave);
    const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
    const dim3 grid(wave_grid(lds, wpb, a.N));
#define STG_L(W)                                                                                              \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_fwd_wave_kernel<W>),          \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_wave: hipFuncSetAttribute");                       \
        hipLaunchKernelGGL(txp_fwd_wave_kernel<W>, grid, dim3(W * 64), lds, st, a, a.params, a.buffers);       \
    } while (0)
    if (wpb == 8) STG_L(8); else if (wpb == 4) STG_L(4); else if (wpb == 2) STG_L(2); else STG_L(1);
#undef STG_L
    STG_LAUNCH_CHECK("txp_fwd_wave");
    return STG_OK;
}

int launch_txp_bwd_wave(const TxpBwdArgs &a0, hipStream_t st) {
    TxpBwdArgs a = a0;
    if (a.wp && txp_bwd_x6_fits(a.lay, a.V)) return launch_txp_bwd_x6(a, st);      // txp_x6.hip
    if (mix_geom(bwd_per_wave_floats, a.V, a.tier.order && a.tier.key_start, &a.mix)) {
        const size_t lds = ((size_t)a.mix.block_floats + wave_param_floats(a.lay)) * sizeof(float);
        const void *fn = a.split_bf16 ? reinterpret_cast<const void *>(&txp_bwd_wave_mixed_kernel<true>)
                                      : reinterpret_cast<const void *>(&txp_bwd_wave_mixed_kernel<false>);
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_wave_mixed: hipFuncSetAttribute");
        if (a.split_bf16)
            hipLaunchKernelGGL(txp_bwd_wave_mixed_kernel<true>, dim3(mix_grid(lds, a.N)), dim3(256), lds, st, a);
        else
            hipLaunchKernelGGL(txp_bwd_wave_mixed_kernel<false>, dim3(mix_grid(lds, a.N)), dim3(256), lds, st, a);
        STG_LAUNCH_CHECK("txp_bwd_wave_mixed");
        return STG_OK;
    }
    const size_t per_wave = bwd_per_wave_floats(a.Vl) * sizeof(float);
    const int wpb = wave_wpb(per_wave);
    const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
    const dim3 grid(wave_grid(lds, wpb, a.N));
#define STG_L2(W, B)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_bwd_wave_kernel<W, B>),       \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_wave: hipFuncSetAttribute");                       \
        hipLaunchKernelGGL((txp_bwd_wave_kernel<W, B>), grid, dim3(W * 64), lds, st, a);                      \
    } while (0)
#define STG_L(W)                                                                                              \
    do {                                                                                                      \
        if (a.split_bf16) STG_L2(W, true); else STG_L2(W, false);                                             \
    } while (0)
    if (wpb == 8) STG_L(8); else if (wpb == 4) STG_L(4); else if (wpb == 2) STG_L(2); else STG_L(1);
#undef STG_L
#undef STG_L2
    STG_LAUNCH_CHECK("txp_bwd_wave");
    return STG_OK;
}

}  // namespace stg
