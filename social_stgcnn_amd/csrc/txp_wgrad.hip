// txp_wgrad (K2): the TXP-CNN weight / bias gradients as ONE skinny GEMM per layer over ALL scenes,
//        dW_l[co][ci][tap] = sum_{scene,pos} dz_l[co][pos] a_l[ci][pos+tap],   db_l = sum dz_l
// M = 12 out-channels (16-row tile), N = 9*c_in (tap, channel) columns + a ones column (bias), K = every position of
// every scene, on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// A workgroup of 8 waves owns a layer and walks its share of the work items (scene, or <= 32-column chunk of a larger
// scene) TOGETHER, the K-steps of an item dealt round-robin over the waves: every wave keeps the layer's 7 (5 for
// layer 0) accumulator tiles in VGPRs for the whole launch, adds its K-steps of every item, and the eight partial sums
// meet in LDS once at the end -- no atomics.  An item's image (plane a_l + dz_l, position-major: what the forward /
// input-gradient kernels wrote, 19 KB) is staged by LDS-DMA into a ring of NBUF buffers TWO items ahead of the one
// being computed (global_load_lds_dwordx4, pieces dealt over the waves, counted s_waitcnt vmcnt so the newer items'
// pieces stay in flight across the one s_barrier per item): the HBM / Infinity-Cache latency of the staging, which
// the wave-per-item form of this kernel paid in full between every two items (staging alone 48 us, MFMA alone 64 us,
// together 98 us: no overlap, every wave stage -> compute -> stage in lockstep), hides behind the MFMAs of two items.
// Four waves per SIMD (two workgroups per CU) keep the matrix pipe fed between a wave's own LDS reads.
#include "txp_wgrad.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P;
constexpr int kWaves = 8;

// s_waitcnt vmcnt(n) for a runtime (wave-uniform) n: leaves the wave's n youngest vector-memory operations in flight
__device__ __forceinline__ void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;    // (waits for more than needed: safe)
    }
}

// one 1 KiB piece of a linear global -> LDS copy: vectors [64 * piece, 64 * piece + 64) of `nvec` 16-byte vectors.
// Inline assembly on purpose: hipcc treats the builtin as a pending LDS write and drains vmcnt(0) in front of every
// later LDS read that might alias it -- which would retire the two items in flight before each K-step.  The asm form
// is invisible to that bookkeeping; completion is counted by hand (wait_vmcnt) and published by the item barrier.
// M0 carries the wave-uniform LDS destination and is saved / restored inside the statement (compiler-reserved).
__device__ __forceinline__ void dma_piece(const float *__restrict__ src, float *lds_dst, int nvec, int piece) {
    const int lane = threadIdx.x & 63;
    const int e = piece * 64 + lane;
    const unsigned dst = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)(lds_dst + 256 * piece));
    if (e < nvec) {
        const float *g = src + 4 * e;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(g), "s"(dst)
                     : "memory");
    }
}

struct Item {
    int vi, w0, vc;        // pedestrians of the scene, first column and width of the chunk (vc == vi: the whole scene)
    bool valid;
    const float *pl, *dz;  // the scene's saved plane a_l and dz_l of this layer
};

template <int CINL, int NBUF, bool BF>
__device__ __forceinline__ void wgrad_layer(const WgradArgs &a, const int32_t *__restrict__ order,
                                            const int32_t *__restrict__ order_peds, const int32_t *__restrict__ num_peds,
                                            int layer, float *sm, int wg, int nwg) {
    constexpr int NCOL = 9 * CINL + 1, NTILE = (NCOL + 15) / 16;
    const ModelLayout &L = a.lay;
    const int V = a.V, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nq = lane & 15, kq = lane >> 4;
    const int co_a = nq < P ? nq : P - 1;          // rows 12..15 of the tile are never written back
    constexpr int ES = BF ? 2 : 4;                 // bytes per stored element (STG_OPT_BF16_STORE: bf16 planes and dz)
    const int image = wgrad_image_floats(V, BF), pslot = wgrad_plane_floats(V, BF);
    auto ld = [&](const float *base, int idx) -> float {          // element idx of a staged array
        if (BF) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(base)[idx] << 16);
        return base[idx];
    };
    const int nch = wgrad_chunks(V), items = a.N * nch;
    const int64_t plane_off = ws_plane_off(L, V, layer), dzs_floats = dz_slot(V);
    int tapr[NTILE], tapc[NTILE], cic[NTILE];      // column (tap, ci) of this lane in every tile
    bool bone[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) {
        int col = tl * 16 + nq;
        bone[tl] = col == NCOL - 1;
        if (col > NCOL - 2) col = NCOL - 2;
        const int tap = col / CINL;
        cic[tl] = col - tap * CINL;
        tapr[tl] = tap / 3 - 1;
        tapc[tl] = tap % 3 - 1;
    }
    f32x4 acc[NTILE];
#pragma unroll
    for (int tl = 0; tl < NTILE; ++tl) acc[tl] = f32x4{0.f, 0.f, 0.f, 0.f};

    // work item of round r, in two steps so that no scalar-load latency ever sits between two items: fetch() issues
    // the (independent) loads of the scene index and its pedestrian count and is called TWO rounds before the item is
    // staged; finish() does the arithmetic when the values are needed.  All wave-uniform scalar work.
    struct Raw { int at, n, v; };
    auto fetch = [&](int r) -> Raw {
        Raw w{-1, 0, 0};
        w.at = walk_item(r, wg, nwg, items, order != nullptr && a.serpentine);
        if (w.at >= 0) {
            const int si = nch > 1 ? w.at / nch : w.at;
            w.n = order ? order[si] : si;
            w.v = order ? order_peds[si] : (num_peds ? num_peds[si] : V);
        }
        return w;
    };
    auto finish = [&](const Raw &w) -> Item {
        Item it{0, 0, 0, false, nullptr, nullptr};
        if (w.at < 0) return it;
        const int chunk = nch > 1 ? w.at - (w.at / nch) * nch : 0;
        const int vfull = w.v < 0 ? 0 : (w.v > V ? V : w.v);
        const int nc = wgrad_chunks(vfull);
        if (vfull == 0 || chunk >= nc) return it;
        int wc = vfull;
        if (nc > 1) wc = (vfull + nc - 1) / nc;
        if (BF && nc > 1) wc = (wc + 1) & ~1;          // (chunks start on even columns: 16-byte aligned rows)
        it.pl = a.ws + w.n * a.ws_stride + plane_off;
        it.dz = a.dzg + ((int64_t)w.n * (L.L + 1) + layer) * dzs_floats;
        it.vi = vfull;
        it.w0 = chunk * wc;
        it.vc = (vfull - it.w0) < wc ? (vfull - it.w0) : wc;
        it.valid = true;
        return it;
    };
    // stage item `it` into buffer `buf`: this wave's share of the LDS-DMA pieces; returns how many it issued
    auto stage = [&](const Item &it, float *buf) -> int {
        if (!it.valid || STG_SKIP(a, 64)) return 0;
        const int vi = it.vi, vc = it.vc, w0 = it.w0;
        const int SWf = save_sw(vi, BF), VWf = save_vw(vi, BF);       // row strides of the saved arrays (positions)
        const int SWc = save_sw(vc, BF), VWc = save_vw(vc, BF);       // row strides of the staged image
        // top and bottom border rows of the plane image are zeros (the saved plane holds the C interior rows)
        if (tid < 2 * SWc * 3) {
            const int b = tid / 3, q = tid - b * 3;
            const int pos = b < SWc ? b : (C + 1) * SWc + (b - SWc);
            store_vec4(buf, pos * 3 + q, f32x4{0.f, 0.f, 0.f, 0.f}, BF);
        }
        const float *pl = it.pl, *dz = it.dz;
        int issued = 0;
        if (vc == vi) {            // whole scene: two linear copies
            const int nv0 = (C * SWf * P * ES + 15) >> 4, nv1 = (C * VWf * P * ES + 15) >> 4;
            const int np0 = (nv0 + 63) >> 6, np1 = (nv1 + 63) >> 6;
            for (int q = wave; q < np0 + np1; q += kWaves) {
                if (q < np0) dma_piece(pl, buf + (SWc * P * ES) / 4, nv0, q);
                else dma_piece(dz, buf + pslot, nv1, q - np0);
                ++issued;
            }
        } else {                   // column chunk: per row, the plane with its two halo columns and the dz columns
            const int nvp = (SWc * P * ES + 15) >> 4, nvd = (vc * P * ES + 15) >> 4;
            const int npp = (nvp + 63) >> 6, npd = (nvd + 63) >> 6;
            for (int q = wave; q < C * (npp + npd); q += kWaves) {
                if (q < C * npp) {
                    const int h = q / npp, pc = q - h * npp;
                    dma_piece(pl + ((int64_t)(h * SWf + w0) * P * ES) / 4, buf + ((h + 1) * SWc * P * ES) / 4, nvp, pc);
                } else {
                    const int q2 = q - C * npp, h = q2 / npd, pc = q2 - h * npd;
                    dma_piece(dz + ((int64_t)(h * VWf + w0) * P * ES) / 4, buf + pslot + (h * VWc * P * ES) / 4, nvd, pc);
                }
                ++issued;
            }
        }
        return issued;
    };
    // this wave's K-steps of the item staged in `buf`
    auto compute = [&](const Item &it, const float *buf) {
        if (!it.valid || STG_SKIP(a, 128)) return;
        const float *plane = buf, *dzs = buf + pslot;
        const int vc = it.vc, SW = save_sw(vc, BF), VW = save_vw(vc, BF), npos = C * vc, nsteps = (npos + 3) >> 2;
        // p / vc == (p * inv) >> 16 for p < 409 with inv = ceil(65536 / vc): from the float reciprocal, fixed up exactly
        unsigned inv = (unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)vc));
        while (inv * (unsigned)vc < 65536u) ++inv;
        while ((inv - 1u) * (unsigned)vc >= 65536u) --inv;
        int boff[NTILE];
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) boff[tl] = (tapr[tl] * SW + tapc[tl]) * P + cic[tl];
        for (int s = wave; s < nsteps; s += kWaves) {
            const int p = 4 * s + kq;
            const bool ok = p < npos;
            const int pc = ok ? p : 0;
            const int hh = (int)(((unsigned)pc * inv) >> 16), ww = pc - hh * vc;
            const float av = ok ? ld(dzs, (hh * VW + ww) * P + co_a) : 0.f;
            const int offb = ((hh + 1) * SW + (ww + 1)) * P;
            float raw[NTILE];
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl) raw[tl] = ld(plane, boff[tl] + offb);
#pragma unroll
            for (int tl = 0; tl < NTILE; ++tl)
                acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bone[tl] ? 1.f : raw[tl], acc[tl], 0, 0, 0);
        }
    };

    // ---- the pipeline: item r is computed from buffer r % NBUF while items r+1 .. r+D-1 are in flight ---------------
    constexpr int D = NBUF - 1;                    // staging runs D items ahead
    const int rounds = (items + nwg - 1) / nwg;    // (the same for every wave of the workgroup: barriers are uniform)
    Item q[D + 1];                                 // q[j] = item of round r + j (finished: pointers and geometry)
    Raw nxt;                                       // raw loads of round r + D + 1 (issued one round before they are used)
    int ahead[D];                                  // ahead[j] = pieces this wave has in flight for item r + 1 + j
#pragma unroll
    for (int j = 0; j < D + 1; ++j) q[j] = finish(fetch(j));
    nxt = fetch(D + 1);
#pragma unroll
    for (int j = 0; j < D; ++j) {                  // prologue: items 0 .. D-1
        const int c = stage(q[j], sm + (j % NBUF) * image);
        if (j > 0) ahead[j - 1] = c;
    }
    ahead[D - 1] = 0;
    for (int r = 0; r < rounds; ++r) {
        // item r's pieces have landed once only the pieces of the newer items are still outstanding
        int newer = 0;
#pragma unroll
        for (int j = 0; j < D - 1; ++j) newer += ahead[j];
        wait_vmcnt(newer);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's border zeros of item r
        __builtin_amdgcn_s_barrier();                                // every wave's pieces of item r; item r-1 consumed
        // refill the buffer item r-1 just left with item r + D
        ahead[D - 1] = stage(q[D], sm + ((r + D) % NBUF) * image);
        compute(q[0], sm + (r % NBUF) * image);
#pragma unroll
        for (int j = 0; j < D; ++j) q[j] = q[j + 1];
        q[D] = finish(nxt);
        nxt = fetch(r + D + 2);
#pragma unroll
        for (int j = 0; j < D - 1; ++j) ahead[j] = ahead[j + 1];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // each wave parks its accumulator tiles as one row [P][CINL][9] weights + [P] biases (the parameters' own order)
    // in LDS; the workgroup sums its eight rows into ONE slab row in a fixed order
    constexpr int ROW = (P * CINL * 9 + P + 3) & ~3;
    float *row = sm + wave * ROW;
    if (kq < 3) {
#pragma unroll
        for (int tl = 0; tl < NTILE; ++tl) {
            const int col = tl * 16 + nq;
            if (col < NCOL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * kq + r;
                    if (col == NCOL - 1) {
                        row[P * CINL * 9 + co] = acc[tl][r];
                    } else {
                        const int tap = col / CINL, ci = col - tap * CINL;
                        row[(co * CINL + ci) * 9 + tap] = acc[tl][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    const int len = wgrad_row_len(layer);
    float *dst = a.slab2 + wgrad_slab_base(layer, a.rows) + (int64_t)wg * len;
    for (int e = tid; e < len; e += kWaves * 64) {
        float t = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) t += sm[w2 * ROW + e];
        dst[e] = t;
    }
}

template <int NBUF, bool BF>
__global__ __launch_bounds__(kWaves * 64, 4) void txp_wgrad_kernel(const WgradArgs a, const int32_t *__restrict__ order,
                                                                   const int32_t *__restrict__ order_peds,
                                                                   const int32_t *__restrict__ num_peds) {
    // (order / num_peds are separate __restrict__ arguments so that the item descriptors are fetched with SCALAR loads:
    // a vector load in the pipelined loop would make hipcc drain vmcnt(0), i.e. the staged items in flight)
    extern __shared__ __attribute__((aligned(16))) float sm[];
    // blockIdx.x -> (layer, workgroup within the layer): layer l owns blocks [wg_begin[l], wg_begin[l+1])
    int layer = 0;
    while (layer < a.lay.L && (int)blockIdx.x >= a.wg_begin[layer + 1]) ++layer;
    const int wg = (int)blockIdx.x - a.wg_begin[layer];
    const int nwg = a.wg_begin[layer + 1] - a.wg_begin[layer];
    if (layer == 0)
        wgrad_layer<Cfg::T, NBUF, BF>(a, order, order_peds, num_peds, layer, sm, wg, nwg);
    else
        wgrad_layer<Cfg::P, NBUF, BF>(a, order, order_peds, num_peds, layer, sm, wg, nwg);
}

}  // namespace

// launch geometry: persistent 8-wave workgroups, two per CU (four waves per SIMD), each with a ring of NBUF item
// images; the chip's workgroup slots are split over the layers in proportion to their MFMA work (layer 0 has 5 column
// tiles, the others 7)
bool wgrad_geom(const ModelLayout &L, int N, int V, WgradGeom *g) {
    const bool bf16 = (L.flags & STG_OPT_BF16_STORE) != 0;
    const size_t image = (size_t)wgrad_image_floats(V, bf16) * sizeof(float);
    const size_t row = (size_t)((Cfg::P * Cfg::P * 9 + Cfg::P + 3) & ~3) * sizeof(float) * kWaves;   // final reduction
    int nbuf = diag_env("STG_WGRAD_NBUF", 4);
    if (nbuf != 3) nbuf = 4;
    int per_cu = 2;
    while (nbuf > 3 && image * nbuf * per_cu > (size_t)kLdsBytes) --nbuf;
    if (image * nbuf * per_cu > (size_t)kLdsBytes) per_cu = 1;
    size_t lds = image * nbuf;
    if (lds < row) lds = row;
    if (lds > (size_t)kLdsBytes) return false;
    g->waves = kWaves;
    g->nbuf = nbuf;
    g->lds = lds;
    g->bf16mma = wgrad_bf16_fits(L, V) && !diag_env("STG_WGRAD_F32", 0);
    if (g->bf16mma) {
        wgrad_bf16_geom(g, L, V);
        per_cu = (int)((size_t)kLdsBytes / g->lds);
    }
    int total = kNumCU * per_cu;                       // resident workgroups on the chip
    if (const int v = diag_env("STG_WGRAD_GRID", 0)) total = v > 0 ? v : total;
    const int nl = L.L + 1;
    if (total < nl) total = nl;
    const int need = N * wgrad_chunks(V);              // never more workgroups per layer than work items
    // per item: the same staging, 5 (layer 0) vs 7 column tiles of MFMAs (bf16 pipe: the staging weighs more; 15 : 18
    // measured best of 12..18 : 18, by 1 us at the headline shape, 3 at V = 64)
    const int w0 = g->bf16mma ? 15 : 6, w1 = g->bf16mma ? 18 : 7, wsum = w0 + w1 * (nl - 1);
    int begin = 0, maxw = 0;
    for (int l = 0; l < nl; ++l) {
        int cnt = (int)((int64_t)total * (l == 0 ? w0 : w1) / wsum);
        if (cnt < 1) cnt = 1;
        if (cnt > need) cnt = need;
        g->wg_begin[l] = begin;
        begin += cnt;
        if (cnt > maxw) maxw = cnt;
    }
    g->wg_begin[nl] = begin;
    g->grid = begin;
    g->rows = maxw;                                    // one slab row per workgroup
    return true;
}

int launch_txp_wgrad(const WgradArgs &w, const WgradGeom &g, hipStream_t st) {
    if (g.bf16mma) return launch_txp_wgrad_bf16(w, g, st);
    const dim3 grid(g.grid), block(kWaves * 64);
#define STG_LAUNCH_WG(NB, BF)                                                                                \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_wgrad_kernel<NB, BF>),       \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);         \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_wgrad: hipFuncSetAttribute");                         \
        hipLaunchKernelGGL((txp_wgrad_kernel<NB, BF>), grid, block, g.lds, st, w, w.order, w.order_peds, w.num_peds); \
    } while (0)
    const bool bf16 = (w.lay.flags & STG_OPT_BF16_STORE) != 0;
    if (g.nbuf == 3) {
        if (bf16) STG_LAUNCH_WG(3, true); else STG_LAUNCH_WG(3, false);
    } else {
        if (bf16) STG_LAUNCH_WG(4, true); else STG_LAUNCH_WG(4, false);
    }
#undef STG_LAUNCH_WG
    STG_LAUNCH_CHECK("txp_wgrad");
    return STG_OK;
}

}  // namespace stg
