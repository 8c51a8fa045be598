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
#include "model_common.hpp"
#include "stgcn_block.hpp"
#include "txp_wave.hpp"
#include "nll_elem.hpp"
#include "txp_conv_bf16.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P, T = Cfg::T;

// LDS floats of one wave's position table (16-bit entries, T * v of them) [+ the st_gcn tail's 32 reduction totals]
__host__ __device__ inline int ptab_floats(int v) { return ((T * v + 1) / 2 + 3) & ~3; }
__host__ __device__ inline int bwd_ptab_floats(int v) { return ptab_floats(v) + kRedMax; }
// LDS floats of the workgroup's copy of the st_gcn block parameters and BatchNorm running statistics: the block code
// of the wave kernels reads them with broadcast LDS reads (no SGPR pressure, no scalar-load waits inside its passes)
__host__ __device__ inline int wave_param_floats(const ModelLayout &L) {
    return ((L.n_blk_params + 3) & ~3) + ((L.n_buffers + 3) & ~3);
}

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

// position -> (h << 8 | w) table of the scene, T * vi entries of 16 bits: p < C * vi are the positions of the TXP
// plane, q < T * vi the (t, w) columns of the st_gcn block -- no integer divisions per tile / column
__device__ __forceinline__ void build_ptab(ptab_t *ptab, int vi) {
    const int lane = threadIdx.x & 63;
    for (int p = lane; p < T * vi; p += 64) {
        const int h = p / vi;
        ptab[p] = (ptab_t)((h << 8) | (p - h * vi));
    }
}
// the same for a wave that owns the column chunk [w0, w0 + wc) of a scene: position p of the chunk is (row p / wc, column
// w0 + p % wc) of the scene
__device__ __forceinline__ void build_ptab(ptab_t *ptab, int w0, int wc) {
    const int lane = threadIdx.x & 63;
    if (wc <= 0 && lane == 0) ptab[0] = 0;            // (an empty chunk: entry 0 is still read, never used)
    for (int p = lane; p < T * wc; p += 64) {
        const int h = p / wc;
        ptab[p] = (ptab_t)((h << 8) | (w0 + p - h * wc));
    }
}


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

// "Every vector-memory operation issued so far has completed", stated where the compiler can see it (an S_WAITCNT it
// models).  vmcnt counts loads AND stores in order; wherever a register MAY still be waiting for a load on some path of
// the control-flow graph (a tile loop whose iterations are guarded by runtime tile counts is enough), the compiler puts
// s_waitcnt vmcnt(0) in front of its use -- which also waits for the acknowledgement of every store issued since: one
// HBM round trip per tile.  Draining once, right after the loads and before the guarded code, leaves nothing pending,
// and the tiles' stores are fire-and-forget again.  (Found in the ISA, not in a counter: DESIGN 5.2.)
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt free

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

// Workgroup prologue of the wave kernels: the st_gcn block's parameters (and running statistics) into LDS, once.
__device__ __forceinline__ void stage_block_params(const ModelLayout &L, const float *__restrict__ params,
                                                   const float *__restrict__ buffers, float *blk_p, float *blk_b, int nt) {
    for (int e = threadIdx.x; e < L.n_blk_params; e += nt) blk_p[e] = params[e];
    if (buffers)
        for (int e = threadIdx.x; e < L.n_buffers; e += nt) blk_b[e] = buffers[e];
    __syncthreads();
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
// forward, exact-bf16 variant (x6 = six bf16 products per fp32 product): the convs on v_mfma_f32_16x16x32_bf16
// ------------------------------------------------------------------------------------------
// txp_conv_bf16.hpp.  a_l lives in LDS as three position-major bf16 piece images (7 row slots: no in-place ring is
// needed, because the layer's outputs stay in REGISTERS until every tile has read its inputs); the same registers are
// the residual input of the next layer -- a lane owns the same (position, channel quad) of every tile in every layer
// (<= 10 tiles for V_n <= 32).  The st_gcn block (column mode) hands its outputs over in registers as well.
constexpr int kF6Tiles = 10, kF6Slots = 7;

__host__ __device__ inline int fwd6_region_floats(int v) { return (cv::image_bytes(v, kF6Slots) / 4 + 3) & ~3; }

// CK (scene_team.hpp): SoloScene -- this wave owns the scene, `region` / `ptab` are its own -- or TeamScene: the wave owns
// the column chunk [ck.w0(), ck.w0() + ck.wc()) of a scene that ck.nch() waves share (`region` = the team's image, `ptab` =
// this wave's table of its chunk's positions); `vi` = pedestrians of the scene.
template <bool BF, typename CK>    // BF: bf16 storage of the saved planes / pre-activations (STG_OPT_BF16_STORE)
__device__ __forceinline__ void txp_fwd_scene_x6(const TxpFwdArgs &a, const float *__restrict__ params,
                                                 const float *blk_params, const float *blk_buffers, int n, int vi,
                                                 float *region, ptab_t *ptab, const CK &ck) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63, nq = lane & 15, kq = lane >> 4;
    float *yn = a.y + (int64_t)n * (C * P) * V;
    if (vi < V)                                        // padded pedestrian slots of the output are zeros
        for (int e = lane + 64 * ck.ci(); e < C * P * (V - vi); e += 64 * ck.nch()) {
            const int r = e / (V - vi), w = vi + (e - r * (V - vi));
            yn[(int64_t)r * V + w] = 0.f;
        }
    if (vi == 0) return;
    const int npos = C * ck.wc(), ntiles = (npos + 15) >> 4;      // this wave's positions: (row, its columns)
    const float *Pm = params;
    float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;
    float *statn = a.stats ? a.stats + (int64_t)n * L.stat_floats : nullptr;
    unsigned char *img = reinterpret_cast<unsigned char *>(region);
    const cv::LaneGeom lg = cv::lane_geom(vi, kF6Slots);
    constexpr bool bf16 = BF;
    const int SWs = save_sw(vi, bf16), VWs = save_vw(vi, bf16);   // row strides (positions) of the saved arrays

    // ---- st_gcn block (model.py:145-155), column mode: lane = pedestrian; zeroes the image, builds the position table
    {
        // lane = (pedestrian, time half); a lane receives plane channels 4*half .. 4*half+3 of its pedestrian's C rows
        float sv[C * T / 2];
        const float *agn = a.agg + n * a.agg_stride;
        stgcn_block_fwd_cols<true>(a, blk_params, blk_buffers, L.blk[0], n, vi, wsn, statn, agn + a.agg_ax, agn + a.agg_cs,
                                   nullptr, 0, region, cv::image_bytes(vi, kF6Slots) >> 4, ptab, sv, ck);
        // (a team: the block's own ck.sync() has made the whole image zero before anyone writes its interior)
        // v.view(N, T, C, V) (model.py:187): flat f = c*T+t of the block output is plane channel f / C, row f % C.  Per
        // row a pedestrian's eight channels are two record quads (one per lane of the pair); the third quad (channels
        // 8..11) stays zero.
        float *d2 = wsn ? wsn + ws_plane_off(L, V, 0) : nullptr;
        const int pl = lane & 31, pw = ck.w0() + pl, q = lane >> 5;
        if (pl < ck.wc()) {
#pragma unroll
            for (int row = 0; row < C; ++row) {
                const f32x4 v4 = {sv[row], sv[C + row], sv[2 * C + row], sv[3 * C + row]};
                cv::put4(img, (unsigned)(cv::pos_off(vi, 1 + row, pw) + 8 * q), lg.PL, v4);
                if (d2) {
                    store_vec4(d2, (row * SWs + pw + 1) * 3 + q, v4, bf16);
                    if (q == 0) store_vec4(d2, (row * SWs + pw + 1) * 3 + 2, f32x4{0.f, 0.f, 0.f, 0.f}, bf16);
                }
            }
        }
        if (wsn && lane < 2 * C * 3 && ck.lead()) {
            // zero border columns of the saved planes a_0 .. a_L (the weight-gradient GEMM reads them)
            const int b = lane / 3, q = lane - b * 3, pos = (b >> 1) * SWs + ((b & 1) ? vi + 1 : 0);
            for (int l = 0; l <= L.L; ++l)
                store_vec4(wsn + ws_plane_off(L, V, l), pos * 3 + q, f32x4{0.f, 0.f, 0.f, 0.f}, bf16);
        }
    }
    ck.sync();

    // ---- TXP-CNN (model.py:187-195) -------------------------------------------------------------------
    if (STG_SKIP(a, 16)) return;                       // (diagnostic build: time the block alone)
    const unsigned lds_base = (unsigned)(uintptr_t)img;
    f32x4 av[kF6Tiles];
#pragma unroll
    for (int t = 0; t < kF6Tiles; ++t) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l <= L.L; ++l) {
        const bool is_out = l == L.L;
        cv::u32x4 w[cv::kWpVecs];
        cv::load_wp(a.wpf + (int64_t)l * cv::kWpDwords, w);
        const float *bias = Pm + (is_out ? L.out_b : L.txp_b[l]);
        f32x4 binit;
#pragma unroll
        for (int r = 0; r < 4; ++r) binit[r] = kq < 3 ? bias[4 * kq + r] : 0.f;
        const float alpha = is_out ? 0.f : Pm[L.prelus + l];
        float *zs = (wsn && !is_out) ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr;
        float *ps = (wsn && !is_out) ? wsn + ws_plane_off(L, V, l + 1) : nullptr;
        // the layer's operands and bias have landed before the first guarded tile: no vmcnt wait inside the tile loop
        asm volatile("" ::"v"(w[0]), "v"(w[cv::kWpVecs - 1]), "v"(binit), "v"(alpha) : "memory");
        vm_drain();
        unsigned code = cv::tile_code(0, ptab, npos);
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) {
            if (t < ntiles) {
                const cv::Tile tl = cv::tile_from<1>(t, code, npos, lg, vi);
                if (t + 1 < kF6Tiles) code = cv::tile_code(t + 1, ptab, npos);      // (in flight behind this tile's reads)
                cv::BHalf b;
                f32x4 z = binit;
                cv::load_b_half<0>(lds_base, tl, b);
                cv::mma_half<0>(w, b, z);
                cv::load_b_half<1>(lds_base, tl, b);
                cv::mma_half<1>(w, b, z);
                if (tl.ok && kq < 3) {
                    if (is_out) {
                        // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
#pragma unroll
                        for (int r = 0; r < 4; ++r) yn[(int64_t)((4 * kq + r) * C + tl.h) * V + tl.w] = z[r];
                    } else {
                        f32x4 v4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = (z[r] > 0.f ? z[r] : alpha * z[r]) + av[t][r];   // (av = 0 at l = 0)
                        av[t] = v4;
                        if (zs) {
                            store_vec4(zs, (tl.h * VWs + tl.w) * 3 + kq, z, bf16);
                            store_vec4(ps, (tl.h * SWs + tl.w + 1) * 3 + kq, v4, bf16);
                        }
                    }
                }
            }
        }
        if (is_out) break;
        // every tile (of every wave of a team) has read a_l: a_{l+1} replaces it in the image
        ck.sync();
        unsigned codes[kF6Tiles];                      // (all ten table reads in flight: the weight registers are dead here)
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) codes[t] = cv::tile_code(t, ptab, npos);
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) {
            const int p = 16 * t + nq;
            if (p < npos && kq < 3) {
                const unsigned hw = codes[t];
                cv::put4(img, (unsigned)(cv::pos_off(vi, 1 + (int)(hw >> 8), (int)(hw & 0xffu)) + 8 * kq), lg.PL, av[t]);
            }
        }
        ck.sync();
    }
}

template <int WPB, bool BF>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_x6_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int per_wave = fwd6_region_floats(Vl) + ptab_floats(Vl);
    float *region = sm + wave * per_wave;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(region + fwd6_region_floats(Vl));
    float *blk_p = sm + WPB * per_wave, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, WPB * 64);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? a.tier.order[begin + it] : it);
        int vi = a.num_peds ? a.num_peds[n] : a.V;
        vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > a.V ? a.V : vi));
        txp_fwd_scene_x6<BF>(a, params, blk_p, blk_b, n, vi, region, ptab, SoloScene{vi});
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
// backward, exact-bf16 variant (x6): the input-gradient GEMMs on v_mfma_f32_16x16x32_bf16 with three-piece operands
// ------------------------------------------------------------------------------------------
// txp_conv_bf16.hpp: dz_l lives in LDS as three position-major bf16 piece images (x = x_h + x_m + x_l exactly), the six
// products that reach 2^-24 are accumulated in fp32 -- the same accuracy class as the fp32 MFMA, but 24 MFMAs of 16
// matrix-pipe cycles per tile that run BESIDE the VALU instead of 27 fp32 MFMAs that hold the SIMD's VALU port for 32
// cycles each (tools/micro/mfma_valu_overlap.hip).  The running input gradient d(a_l) stays in REGISTERS: a lane owns
// the same (position, channel quad) of every tile in every layer (<= 10 tiles for V_n <= 32), which is also the quad
// structure of the saved z_l / dz_l arrays -- no LDS copy of it, no position table in the dz construction.
constexpr int kX6Tiles = 10;                          // 16-position tiles of a scene of <= 32 pedestrians
constexpr int kX6Slots = 7;                           // row slots of the dz image: borders + C interior rows
__host__ __device__ inline int bwd6_region_floats(int v) {
    const int img = cv::image_bytes(v, kX6Slots) / 4, tail = (2 * C * (T + 2) + C * T) * v;    // (the block tail's arrays)
    return ((img > tail ? img : tail) + 3) & ~3;
}

// CK (scene_team.hpp): SoloScene -- this wave owns the scene -- or TeamScene: the wave owns the column chunk [ck.w0(),
// ck.w0() + ck.wc()) of a scene shared by ck.nch() waves (`region` = the team's image, `ptab` = this wave's table of its
// chunk's positions); `vi` = pedestrians of the scene.  Per-scene sums are exchanged through LDS (ck.sum), the team's
// leading wave writes the scene's loss and its row of small-parameter gradients.
template <bool BF, typename CK>    // BF: bf16 storage of z_l (read) and dz_l (written)
__device__ __forceinline__ void txp_bwd_scene_x6(const TxpBwdArgs &a, const float *blk_params, int n, int vi, float *region,
                                                 ptab_t *ptab, const CK &ck) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63, nq = lane & 15, kq = lane >> 4;
    float *slope_row = a.rows + (int64_t)n * (L.n_blk_params + L.n_txp) + L.n_blk_params;
    if (vi == 0) {                                     // empty scene: its row of small-parameter gradients is zero
        for (int e = lane; e < L.n_blk_params + L.n_txp; e += 64) slope_row[e - L.n_blk_params] = 0.f;
        if (a.nll_target && lane == 0) a.nll_losses[n] = 0.f;
        return;
    }
    const int npos = C * ck.wc(), ntiles = (npos + 15) >> 4;      // this wave's positions: (row, its columns)
    const float *Pm = a.params;
    const float *wsn = a.ws + n * a.ws_stride;
    const float *dyn = a.dy + (int64_t)n * (C * P) * V;
    // sums that only leave the kernel (PReLU slope gradients, the loss, the block's parameter gradients): a solo wave writes
    // them to the scene's row; the waves of a team park theirs in LDS rows (zeroed here) that the leading wave adds at the end
    float *prow = ck.row(slope_row - L.n_blk_params);
    if constexpr (CK::kTeam) {
        for (int e = lane; e < kTeamRow; e += 64) prow[e] = 0.f;
    }
    unsigned char *img = reinterpret_cast<unsigned char *>(region);
    const cv::LaneGeom lg = cv::lane_geom(vi, kX6Slots);
    constexpr bool bf16 = BF;
    const int VWs = save_vw(vi, bf16);                             // row stride (positions) of the saved z_l / dz_l
    // From ONE read of the position table: the vector index of the lane's quad of tile t in those arrays (fp32: rows of vi
    // positions, i.e. (16 t + n) * 3 + kq) and the byte offset of its record quad in the image (interior row 0 = slot 1;
    // -1 past the scene's last position)
    auto tile_slots = [&](int t, int &rec, int &quad) {
        const int p = 16 * t + nq;
        const unsigned hw = ptab[p < npos ? p : 0];
        const int h = (int)(hw >> 8), w = (int)(hw & 0xffu);
        quad = (h * VWs + w) * 3 + kq;
        rec = (p < npos && kq < 3) ? cv::pos_off(vi, 1 + h, w) + 8 * kq : -1;
    };
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(img);
        for (int e = lane + 64 * ck.ci(); e < cv::image_bytes(vi, kX6Slots) >> 4; e += 64 * ck.nch()) z4[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    build_ptab(ptab, ck.w0(), ck.wc());
    ck.sync();
    f32x4 dcur[kX6Tiles];
#pragma unroll
    for (int t = 0; t < kX6Tiles; ++t) dcur[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int l = L.L; l >= 0; --l) {
        // ---- dz_l -> the piece images (and position-major fp32 to HBM for the weight-gradient GEMM) -----------------
        float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + l) * dz_slot(V);
        if (STG_SKIP(a, 512) || (l == L.L && STG_SKIP(a, 2048)) || (l != L.L && STG_SKIP(a, 4096))) {
        } else if (l == L.L) {
            // dz of the output conv is dV_pred: row rc = f * P + p of the (C*P) x V array is channel rc / C, plane row
            // rc % C.  Lanes are laid over (row or prediction step, pedestrian) with the row length rounded up to a power
            // of two.
            const int wcw = ck.wc();
            const int vp = wcw <= 1 ? 1 : (wcw <= 2 ? 2 : (wcw <= 4 ? 4 : (wcw <= 8 ? 8 : (wcw <= 16 ? 16 : 32))));
            const int sh = __builtin_ctz(vp), rpi = 64 >> sh;
            const int sub = lane >> sh, wl = lane & (vp - 1), w = ck.w0() + wl;      // the lane's pedestrian
            const bool okw = wl < wcw;
            // the values pass through an fp32 staging array S [C*P rows][vi] laid over the (still empty) m / l piece
            // images: coalesced along the pedestrians here, read back as record quads below
            float *S = reinterpret_cast<float *>(img + lg.PL);
            auto put1 = [&](int rc, float g) { S[rc * vi + w] = g; };
            if (a.nll_target) {
                const float *tn = a.nll_target + (int64_t)n * P * V * 2;
                const float inv_cnt = 1.0f / (float)(P * vi);
                const float gs = inv_cnt * (a.nll_weights ? a.nll_weights[n] : 1.f);
                float lacc = 0.f;
                for (int p0 = 0; p0 < P; p0 += rpi) {
                    const int p = p0 + sub;
                    if (okw && p < P) {
                        const float *q = dyn + (int64_t)p * V + w;
                        const float2 tg = *reinterpret_cast<const float2 *>(tn + ((int64_t)p * V + w) * 2);
                        float g[5];
                        lacc += nll_elem(q[0], q[(int64_t)P * V], q[(int64_t)2 * P * V], q[(int64_t)3 * P * V],
                                         q[(int64_t)4 * P * V], tg.x, tg.y, true, g);
#pragma unroll
                        for (int f = 0; f < C; ++f) put1(f * P + p, g[f] * gs);
                    }
                }
                float ls[1] = {lacc};
                ck.template reduce<1>(ls);
                if (ck.writer()) {
                    if constexpr (CK::kTeam) prow[kTeamRowLoss] = ls[0];
                    else a.nll_losses[n] = ls[0] * inv_cnt;
                }
            } else {
                constexpr int U = 4;
                for (int r0 = 0; r0 < C * P; r0 += rpi * U) {
                    float dv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        dv[u] = (okw && row < C * P) ? dyn[(int64_t)row * V + w] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        if (okw && row < C * P) put1(row, dv[u]);
                    }
                }
            }
            // quads: the lane's (position, channel quad) of every tile from S (plane (ch, row) = row ch*C + row of the
            // array), held in registers while S is wiped (the m / l images must be zero outside the interior), then
            // split into the three images and stored position-major for the weight-gradient GEMM (16 bytes per lane: a
            // scattered 4-byte store and three 2-byte LDS stores per value cost 20 us)
            // (the registers of the running input gradient are free here: the output conv's chain starts from zero)
            __builtin_amdgcn_wave_barrier();
            vm_drain();                                 // (V_pred / target are consumed: nothing pending past here)
            f32x4 (&qd)[kX6Tiles] = dcur;
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                const int p = 16 * t + nq;
                const unsigned hw = ptab[p < npos ? p : 0];
                const float *sq = S + ((4 * (kq < 3 ? kq : 0)) * C + (int)(hw >> 8)) * vi + (int)(hw & 0xffu);
                qd[t] = f32x4{sq[0], sq[C * vi], sq[2 * C * vi], sq[3 * C * vi]};
            }
            ck.sync();                                  // (every wave of a team has its quads: S may be wiped)
            {
                uint4 *z4 = reinterpret_cast<uint4 *>(img + lg.PL);
                for (int e = lane + 64 * ck.ci(); e < (2 * lg.PL + 128) >> 4; e += 64 * ck.nch()) z4[e] = make_uint4(0u, 0u, 0u, 0u);
            }
            ck.sync();
            int rec[kX6Tiles], qv[kX6Tiles];           // (the table reads of all ten tiles in flight together)
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) tile_slots(t, rec[t], qv[t]);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (rec[t] >= 0) {
                    cv::put4(img, (unsigned)rec[t], lg.PL, qd[t]);
                    store_vec4(dzo, qv[t], qd[t], bf16);
                }
            }
        } else {
            // dz_l = d(a_{l+1}) * prelu'(z_l): z_l and dz_l are position-major [pos][12] in HBM, the lane's quad of tile t
            // is vector (16 t + n) * 3 + kq
            const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
            const float alpha = Pm[L.prelus + l];
            float slope_acc = 0.f;
            // all ten quads of z_l in flight at once (the weight registers are dead here): one HBM latency per layer.
            // (Requesting them BEFORE the previous layer's MFMAs needs 40 more live registers there and spilled; a
            // never-awaited "touch" load into a dead register is not an option either -- the register is reused
            // while the load is in flight and the late write-back corrupts its new owner.)
            f32x4 zv[kX6Tiles];
            int rec[kX6Tiles], qv[kX6Tiles];
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) tile_slots(t, rec[t], qv[t]);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t)
                zv[t] = rec[t] >= 0 ? load_vec4_raw(zl, qv[t], bf16) : f32x4{1.f, 1.f, 1.f, 1.f};
            // all ten have landed before the first guarded tile: the tiles' dz stores are not waited for (vm_drain)
            asm volatile("" ::"v"(zv[0]), "v"(zv[1]), "v"(zv[2]), "v"(zv[3]), "v"(zv[4]), "v"(zv[5]), "v"(zv[6]), "v"(zv[7]),
                         "v"(zv[8]), "v"(zv[9]), "v"(alpha)
                         : "memory");
            vm_drain();
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (rec[t] >= 0) {
                    f32x4 dzv;
                    const f32x4 zq = finish_vec4(zv[t], bf16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = zq[r], d = dcur[t][r];
                        float dz = d;
                        if (!(z > 0.f)) {
                            dz = alpha * d;
                            slope_acc = fmaf(d, z, slope_acc);
                        }
                        dzv[r] = dz;
                    }
                    cv::put4(img, (unsigned)rec[t], lg.PL, dzv);
                    store_vec4(dzo, qv[t], dzv, bf16);
                }
            }
            float ss[1] = {slope_acc};
            ck.template reduce<1>(ss);
            if (ck.writer()) prow[L.n_blk_params + l] = ss[0];
        }
        ck.sync();
        // ---- d(a_l) = conv_transpose(dz_l, W_l) [+ d(a_{l+1}) through the residual of the hidden layers] ----------------
        if (!STG_SKIP(a, 1024)) {
            cv::u32x4 w[cv::kWpVecs];
            cv::load_wp(a.wp + (int64_t)l * cv::kWpDwords, w);
            asm volatile("" ::"v"(w[0]), "v"(w[cv::kWpVecs - 1]) : "memory");
            vm_drain();                                 // (no per-operand waits in front of every guarded tile)
            const unsigned lds_base = (unsigned)(uintptr_t)img;
            const bool keep = l != L.L && l != 0;       // d(a_l) += d(a_{l+1}) (a_{l+1} = prelu(z_l) + a_l for 1 <= l < L)
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            unsigned code = cv::tile_code(0, ptab, npos);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (t < ntiles) {
                    const cv::Tile tl = cv::tile_from<1>(t, code, npos, lg, vi);
                    if (t + 1 < kX6Tiles) code = cv::tile_code(t + 1, ptab, npos);  // (in flight behind this tile's reads)
                    // two half-tiles through ONE 32-register operand set (96 weight + 40 gradient registers are live);
                    // each half's reads and their wait are one asm statement, the SIMD's other wave covers the latency
                    cv::BHalf b;
                    f32x4 acc = keep ? dcur[t] : zero;
                    cv::load_b_half<0>(lds_base, tl, b);
                    cv::mma_half<0>(w, b, acc);
                    cv::load_b_half<1>(lds_base, tl, b);
                    cv::mma_half<1>(w, b, acc);
                    dcur[t] = acc;
                }
            }
        }
        ck.sync();
    }
    // ---- d(a_0) (channels 0..T-1) -> D [C][T][vi] at the start of the region: v.view(N, T, C, V) (model.py:187)
    // backwards, plane (ch, row) is flat f = ch*C + row = c*T + t of the block output.  The dz image is dead.
    float *D = region;
#pragma unroll
    for (int t = 0; t < kX6Tiles; ++t) {
        const int p = 16 * t + nq;
        if (p < npos && kq < T / 4) {
            const unsigned hw = ptab[p];
            const int h = (int)(hw >> 8), ww = (int)(hw & 0xffu);
#pragma unroll
            for (int r = 0; r < 4; ++r) D[((4 * kq + r) * C + h) * vi + ww] = dcur[t][r];
        }
    }
    __builtin_amdgcn_wave_barrier();                  // (a lane reads back its own wave's columns of D)
    // ---- the st_gcn block (model.py:145-155 backwards), column mode: lane = (pedestrian, time half) ------------------------
    if constexpr (!CK::kTeam) {
        for (int e = L.L + lane; e < L.n_txp; e += 64) slope_row[e] = 0.f;       // dead slopes (layers >= L)
    }
    if (!STG_SKIP(a, 4))
        stgcn_block_bwd_cols<true>(a, blk_params, L.blk[0], n, vi, D, slope_row - L.n_blk_params, a.ws + n * a.ws_stride, ck);
    if constexpr (CK::kTeam) {
        // the team's parked rows -> the scene's row of small-parameter gradients and its loss, added in chunk order
        ck.sync();
        if (ck.lead()) {
            float *row = slope_row - L.n_blk_params;
            const int nrow = L.n_blk_params + L.n_txp;
            for (int e0 = 0; e0 < kTeamRow; e0 += 64) {
                const int e = e0 + lane < kTeamRow ? e0 + lane : kTeamRow - 2;      // (a spare place, never stored)
                float t = ck.row_of(0)[e];
                for (int c = 1; c < ck.nch(); ++c) t += ck.row_of(c)[e];
                if (e < nrow) row[e] = t;
                if (e == kTeamRowLoss && a.nll_target) a.nll_losses[n] = t * (1.0f / (float)(P * vi));
            }
        }
    }
}

template <int WPB, bool BF>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_bwd_x6_kernel(
    const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int per_wave = bwd6_region_floats(Vl) + bwd_ptab_floats(Vl);
    float *region = sm + wave * per_wave;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(region + bwd6_region_floats(Vl));
    float *blk_p = sm + WPB * per_wave;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, WPB * 64);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? a.tier.order[begin + it] : it);
        int vi = a.num_peds ? a.num_peds[n] : a.V;
        vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > a.V ? a.V : vi));
        txp_bwd_scene_x6<BF>(a, blk_p, n, vi, region, ptab, SoloScene{vi});
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// team launch: scenes of 1 .. kTeamMaxV pedestrians in ONE launch, one / two / four waves per scene (scene_team.hpp)
// ------------------------------------------------------------------------------------------
// Work units of a launch, in list order: one per four-wave scene, one per PAIR of two-wave scenes, one per four solo scenes
// (the sorted list is descending, so the units come heaviest first).  A workgroup takes the units of a fixed walk; all four
// waves of a workgroup are always in the same kind of unit, so the workgroup barriers inside the scene code match.
struct TeamCount {
    int n4, n2, n1;        // scenes per class (sorted list: [0, n4) | [n4, n4 + n2) | the rest)
    int u4, u2, units;
    bool sorted;
};
__device__ __forceinline__ TeamCount team_count(const SceneTier &t, const TeamGeom &g, const int32_t *num_peds, int N, int V) {
    TeamCount c;
    c.sorted = t.order && t.key_start;
    if (c.sorted) {
        // key_start[k] = number of scenes with more than V - k pedestrians
        const int a4 = g.v2 < V ? t.key_start[V - g.v2] : 0, a2 = g.v1 < V ? t.key_start[V - g.v1] : 0;
        c.n4 = a4;
        c.n2 = a2 - a4;
        c.u2 = (c.n2 + 1) >> 1;
    } else {
        // no sorted list: every scene in the class of the padded V.  With num_peds (a single scene, a batch beyond the sort's
        // limits) no pairs of two-wave scenes -- an EMPTY scene may only leave the barrier sequence together with its whole
        // workgroup
        const bool pairs = !num_peds && V > g.v1 && V <= g.v2;
        c.n4 = (V > g.v1 && !pairs) ? N : 0;
        c.n2 = pairs ? N : 0;
        c.u2 = (c.n2 + 1) >> 1;
    }
    c.n1 = N - c.n4 - c.n2;
    c.u4 = c.n4;
    c.units = c.u4 + c.u2 + ((c.n1 + 3) >> 2);
    return c;
}
struct TeamUnit {
    int n, vi;             // scene (n < 0: this wave idles this round) and its pedestrians
    int nch, ci, slot;     // waves on the scene, this wave's place among them, which of the round's 4 / nch scenes
    int w0, wc;            // this wave's column chunk
};
__device__ __forceinline__ TeamUnit team_unit(const SceneTier &t, const TeamCount &c, const int32_t *__restrict__ num_peds,
                                              int N, int V, int u) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TeamUnit q;
    int idx;
    if (u < c.u4) {
        q.nch = 4; q.slot = 0; q.ci = wave; idx = u;
    } else if (u < c.u4 + c.u2) {
        const int i0 = c.n4 + 2 * (u - c.u4);
        if (i0 + 1 < c.n4 + c.n2) { q.nch = 2; q.slot = wave >> 1; q.ci = wave & 1; idx = i0 + q.slot; }
        else { q.nch = 4; q.slot = 0; q.ci = wave; idx = i0; }           // the odd one out takes the whole workgroup
    } else {
        q.nch = 1; q.slot = wave; q.ci = 0; idx = c.n4 + c.n2 + 4 * (u - c.u4 - c.u2) + wave;
    }
    q.n = idx < N ? (c.sorted ? t.order[idx] : idx) : -1;
    q.n = __builtin_amdgcn_readfirstlane(q.n);
    int vi = q.n >= 0 ? (num_peds ? num_peds[q.n] : V) : 0;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    q.vi = vi;
    const int wc = (vi + q.nch - 1) / q.nch;          // equal chunks; the last one may be short (or empty)
    q.w0 = q.ci * wc;
    q.wc = vi - q.w0 < wc ? (vi - q.w0 > 0 ? vi - q.w0 : 0) : wc;
    return q;
}
// equal shares: the fewest rounds the grid can do, then just enough workgroups for them
__device__ __forceinline__ int team_workers(int units, int grid) {
    const int rounds = (units + grid - 1) / grid;
    return rounds > 0 ? (units + rounds - 1) / rounds : 1;
}
__host__ __device__ inline int team_ptab_floats() { return ptab_floats(32); }

template <bool BF>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_team_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = a.team.region_floats;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(sm + R + wave * team_ptab_floats());
    float *xr = sm + R + 4 * team_ptab_floats();
    float *blk_p = xr + 2 * kXrFwd, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, 256);
    const TeamCount tc = team_count(a.tier, a.team, a.num_peds, a.N, a.V);
    const int G = team_workers(tc.units, gridDim.x);
    int prev = 1;
    for (int r = 0; r * G < tc.units; ++r) {
        const int u = walk_item(r, blockIdx.x, G, tc.units, a.tier.serpentine != 0);
        if (u < 0 || (int)blockIdx.x >= G) continue;
        const TeamUnit q = team_unit(a.tier, tc, a.num_peds, a.N, a.V, u);
        if (q.nch > 1 || prev > 1) team_barrier();     // (the previous round is over before a region changes hands)
        prev = q.nch;
        if (q.n >= 0) {
            const TeamScene ck{q.vi, q.w0, q.wc, q.nch, q.ci, xr + (q.slot & 1) * kXrFwd, nullptr, q.nch > 1 && !STG_SKIP(a, 1 << 20)};
            float *region = sm + q.slot * ((R >> 2) * q.nch);
            txp_fwd_scene_x6<BF>(a, params, blk_p, blk_b, q.n, q.vi, region, ptab, ck);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool BF>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_bwd_team_kernel(const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = a.team.region_floats;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(sm + R + wave * team_ptab_floats());
    float *xr = sm + R + 4 * team_ptab_floats();
    float *blk_p = xr + kXrBwdWg;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, 256);
    const TeamCount tc = team_count(a.tier, a.team, a.num_peds, a.N, a.V);
    const int G = team_workers(tc.units, gridDim.x);
    int prev = 1;
    for (int r = 0; r * G < tc.units; ++r) {
        const int u = walk_item(r, blockIdx.x, G, tc.units, a.tier.serpentine != 0);
        if (u < 0 || (int)blockIdx.x >= G) continue;
        const TeamUnit q = team_unit(a.tier, tc, a.num_peds, a.N, a.V, u);
        if (q.nch > 1 || prev > 1) team_barrier();
        prev = q.nch;
        if (q.n >= 0) {
            // (a team of nch waves is waves slot * nch .. slot * nch + nch - 1 of the workgroup: their parked rows are adjacent)
            const TeamScene ck{q.vi, q.w0, q.wc, q.nch, q.ci, xr + (q.slot & 1) * kXrRows,
                               xr + 2 * kXrRows + q.slot * q.nch * kTeamRow, q.nch > 1 && !STG_SKIP(a, 1 << 20)};
            float *region = sm + q.slot * ((R >> 2) * q.nch);
            txp_bwd_scene_x6<BF>(a, blk_p, q.n, q.vi, region, ptab, ck);
        }
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
static int wave_wpb(size_t per_wave) {
    // 4 waves per workgroup: the LDS footprint then admits either one workgroup (forward: 4 waves per CU,
    // one per SIMD) or two (backward: 8 per CU, two per SIMD) -- BALANCED over the four SIMDs.  Odd
    // residencies (6 waves per CU) measured 1.5x slower per wave (tools/micro/conv_tile_bench.hip).
    int wpb = diag_env("STG_TXP_WPB", 4);
    if (wpb != 1 && wpb != 2 && wpb != 8) wpb = 4;
    while (wpb > 1 && per_wave * wpb > (size_t)kLdsBytes) wpb >>= 1;
    return wpb;
}

// persistent grid: as many workgroups as the chip holds at once (LDS-limited, 2 waves per SIMD)
static int wave_grid(size_t lds, int wpb, int N) {
    int per_cu = (int)(kLdsBytes / lds);
    if (per_cu * wpb > 8) per_cu = 8 / wpb;
    if (per_cu < 1) per_cu = 1;
    const int need = (N + wpb - 1) / wpb;
    const int g = kNumCU * per_cu;
    if (g >= need) return need;
    // equal shares: the smallest number of rounds that fits, then just enough workgroups for it
    const int rounds = (need + g - 1) / g;
    return (need + rounds - 1) / rounds;
}

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

bool txp_fwd_x6_fits(const ModelLayout &L, int V) {
    return L.n_txp > 0 && V <= kTeamMaxV && !(L.flags & STG_OPT_F32_MFMA) && L.n_blocks == 1 &&
           L.blk[0].cin == Cfg::CIN0 && !diag_env("STG_FWD_F32", 0);
}

// Team launch geometry.  Class bounds: a scene of up to v1 pedestrians belongs to one wave, up to v2 to two, beyond to four
// (a wave's chunk is at most 32 columns = 10 tiles).  Large batches fill the chip with whole scenes (32 / 64); a small
// batch is latency-bound -- every scene's dependency chain IS the step -- so its scenes are cut finer.
// A small batch is latency-bound -- with fewer scene-waves than the chip has SIMDs every scene's dependency chain IS the
// step -- so its scenes are cut finer (measured, synthetic V = 32 / eth-train histogram, M scene-windows/s: N = 256
// 1.88 -> 2.17 at (8, 16); N = 512 3.27 -> 3.71, eth/train x 512 3.07 -> 3.57 at (16, 32); N = 1024 5.52 -> 5.74 / 5.57 ->
// 6.33 at (16, 32); profiles/r03_team_bounds.log).  `uniform` (no num_peds): every scene has V pedestrians, the number of
// scene-waves is known -- never cut so fine that they no longer fit the chip's 2048 wave slots at once.
constexpr int kTeamFineBatch = 1536, kTeamFinestBatch = 384;
static int team_waves(int v, int v1, int v2) { return v <= v1 ? 1 : (v <= v2 ? 2 : 4); }
static bool team_geom(int N, int V, bool uniform, TeamGeom *g) {
    g->on = 0;
    if (V > kTeamMaxV) return false;
    int v1 = 32, v2 = 64;
    if (N < kTeamFinestBatch) { v1 = 8; v2 = 16; }
    else if (N < kTeamFineBatch) { v1 = 16; v2 = 32; }
    if (uniform)
        while (v1 < 32 && (int64_t)N * team_waves(V, v1, v2) > 2048) { v1 *= 2; v2 *= 2; }
    if (const int e = diag_env("STG_TEAM_V1", 0)) v1 = e;
    if (const int e = diag_env("STG_TEAM_V2", 0)) v2 = e;
    if (v1 > 32) v1 = 32;
    if (v2 > 64) v2 = 64;
    if (v2 < v1) v2 = v1;
    // (the backward's column-mode block needs D = [C][T][v] of the dead image, not the LDS arrays of bwd6_region_floats)
    auto region = [&](int v) { return fwd6_region_floats(v); };
    int r = 4 * region(v1 < V ? v1 : V);
    if (2 * region(v2 < V ? v2 : V) > r) r = 2 * region(v2 < V ? v2 : V);
    if (region(V) > r) r = region(V);
    g->on = 1; g->v1 = v1; g->v2 = v2; g->region_floats = (r + 15) & ~15;
    return true;
}
// beyond 32 pedestrians always; up to 32 when the batch is small enough for finer teams to pay
static bool team_wanted(int N, int V, bool uniform, TeamGeom *g) {
    if (!team_geom(N, V, uniform, g)) return false;
    return V > 16 * kF6Tiles / C || g->v1 < V || diag_env("STG_TEAM", 0) != 0;
}
static int team_grid(size_t lds_bytes, int N) {
    int per_cu = (int)(kLdsBytes / lds_bytes);
    if (per_cu > 2) per_cu = 2;                        // 256-register kernels: two waves per SIMD
    if (per_cu < 1) per_cu = 1;
    const int g = kNumCU * per_cu;
    return g < N ? g : N;                              // (at most one unit per scene)
}

int launch_txp_fwd_wave(const TxpFwdArgs &a0, hipStream_t st) {
    TxpFwdArgs a = a0;
    if (a.wpf && txp_fwd_x6_fits(a.lay, a.V) && team_wanted(a.N, a.V, a.num_peds == nullptr, &a.team)) {
        const size_t lds = ((size_t)a.team.region_floats + 4 * team_ptab_floats() + 2 * kXrFwd + wave_param_floats(a.lay)) * sizeof(float);
        STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "txp_fwd_team: V=%d needs %zu bytes of LDS", a.V, lds);
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
        const void *fn = bf ? reinterpret_cast<const void *>(&txp_fwd_team_kernel<true>)
                            : reinterpret_cast<const void *>(&txp_fwd_team_kernel<false>);
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_team: hipFuncSetAttribute");
        const dim3 grid(team_grid(lds, a.N));
        if (bf) hipLaunchKernelGGL(txp_fwd_team_kernel<true>, grid, dim3(256), lds, st, a, a.params, a.buffers);
        else hipLaunchKernelGGL(txp_fwd_team_kernel<false>, grid, dim3(256), lds, st, a, a.params, a.buffers);
        STG_LAUNCH_CHECK("txp_fwd_team");
        return STG_OK;
    }
    if (a.wpf && txp_fwd_x6_fits(a.lay, a.V) && a.V <= 16 * kF6Tiles / C) {
        const size_t per_wave = (size_t)(fwd6_region_floats(a.Vl) + ptab_floats(a.Vl)) * sizeof(float);
        const int wpb = wave_wpb(per_wave) == 8 ? 8 : 4;     // (the 18 KB images of V <= 32 always fit four waves)
        const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
        const dim3 grid(wave_grid(lds, wpb, a.N));
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
#define STG_LX(W, B)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_fwd_x6_kernel<W, B>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_x6: hipFuncSetAttribute");                         \
        hipLaunchKernelGGL((txp_fwd_x6_kernel<W, B>), grid, dim3(W * 64), lds, st, a, a.params, a.buffers);   \
    } while (0)
        if (wpb == 8) { if (bf) STG_LX(8, true); else STG_LX(8, false); }
        else { if (bf) STG_LX(4, true); else STG_LX(4, false); }
#undef STG_LX
        STG_LAUNCH_CHECK("txp_fwd_x6");
        return STG_OK;
    }
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

bool txp_bwd_x6_fits(const ModelLayout &L, int V) {
    return L.n_txp > 0 && V <= kTeamMaxV && L.n_blocks == 1 && L.blk[0].cin == Cfg::CIN0 &&
           !(L.flags & (STG_OPT_SPLIT_BF16 | STG_OPT_F32_MFMA)) && !diag_env("STG_BWD_F32", 0);
}
int64_t txp_bwd_x6_wp_floats(const ModelLayout &L) { return (int64_t)(L.L + 1) * cv::kWpDwords; }
int launch_txp_bwd_wave(const TxpBwdArgs &a0, hipStream_t st) {
    TxpBwdArgs a = a0;
    if (a.wp && txp_bwd_x6_fits(a.lay, a.V) && team_wanted(a.N, a.V, a.num_peds == nullptr, &a.team)) {
        const size_t lds = ((size_t)a.team.region_floats + 4 * team_ptab_floats() + kXrBwdWg + wave_param_floats(a.lay)) * sizeof(float);
        STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "txp_bwd_team: V=%d needs %zu bytes of LDS", a.V, lds);
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
        const void *fn = bf ? reinterpret_cast<const void *>(&txp_bwd_team_kernel<true>)
                            : reinterpret_cast<const void *>(&txp_bwd_team_kernel<false>);
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_team: hipFuncSetAttribute");
        const dim3 grid(team_grid(lds, a.N));
        if (bf) hipLaunchKernelGGL(txp_bwd_team_kernel<true>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(txp_bwd_team_kernel<false>, grid, dim3(256), lds, st, a);
        STG_LAUNCH_CHECK("txp_bwd_team");
        return STG_OK;
    }
    if (a.wp && txp_bwd_x6_fits(a.lay, a.V) && a.V <= 16 * kX6Tiles / C) {
        const size_t per_wave = (size_t)(bwd6_region_floats(a.Vl) + bwd_ptab_floats(a.Vl)) * sizeof(float);
        const int wpb = wave_wpb(per_wave) == 8 ? 8 : 4;
        const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
        const dim3 grid(wave_grid(lds, wpb, a.N));
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
#define STG_LX(W, B)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_bwd_x6_kernel<W, B>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_x6: hipFuncSetAttribute");                         \
        hipLaunchKernelGGL((txp_bwd_x6_kernel<W, B>), grid, dim3(W * 64), lds, st, a);                        \
    } while (0)
        if (wpb == 8) { if (bf) STG_LX(8, true); else STG_LX(8, false); }
        else { if (bf) STG_LX(4, true); else STG_LX(4, false); }
#undef STG_LX
        STG_LAUNCH_CHECK("txp_bwd_x6");
        return STG_OK;
    }
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
