// txp_wgrad_bf16 (K2): the TXP-CNN weight / bias gradients
//        dW_l[co][ci][tap] = sum_{scene,pos} dz_l[co][pos] a_l[ci][pos+tap],   db_l = sum dz_l
// on v_mfma_f32_16x16x32_bf16 with exact three-piece operands (txp_conv_bf16.hpp), six products per fp32 product --
// fp32-accurate, not bit-exact (the m.l, l.m and l.l products are dropped).  On gfx950 the fp32 MFMA (what txp_wgrad.hip
// issues) holds the SIMD's VALU port for all of its 32 cycles; the bf16 MFMAs take 16 cycles of the matrix pipe for 8x
// the K and run beside the VALU / LDS instructions of the other waves (tools/micro/mfma_valu_overlap.hip).
// GEMM shape: M = 12 out-channels (dz, one 16-row tile), N = (tap, input channel) columns, K = 32 positions per MFMA.  Both
// operands want 8 consecutive positions of one channel per lane while the saved arrays are position-major ([pos][12], what
// the forward / input-gradient kernels write with 16-byte stores): the LDS image keeps 96-byte position records
// [3 pieces][12 channels + pad] and the operands are fetched with ds_read_b64_tr_b16, the transposing LDS read
// (tools/micro/tr_read_probe.hip).  A workgroup owns a layer and walks its share of the work items (a scene, or a
// <= 32-column chunk of a larger one): fp32 quads fetched one item ahead, split, written to an LDS image, one barrier,
// MFMAs; partial sums meet in LDS once at the end (fixed order, no atomics).
// Second form of this kernel (round 3; the first: nine tap tiles + a bias tile, per-item image geometry, 71 us at the
// headline shape -- 27 us of it with staging and MFMAs switched off; tools/micro/k2_bench.hip is the A/B harness):
//  * COLUMN-PACKED GEMM: the N dimension is the 9 * c_in (tap, input channel) pairs + one bias column, packed densely --
//    7 tiles of 16 columns (5 for layer 0) instead of nine tap tiles + a bias tile: 42 MFMAs per K-step instead of 57.  A
//    lane of the transposing LDS read supplies ITS OWN record address, so the four channel quads of a tile may belong to
//    different taps; the bias column's operand is a constant record of ones.  (Two of a tile's four quads then share a
//    quad slot of the 96-byte record -- a two-way bank conflict on those reads; nine conflict-free tap tiles with the
//    bias in the record's pad quad were measured too: 55.6 us against 47.4, the MFMA count decides.)
//  * FIXED IMAGE GEOMETRY: the LDS image of a work item always has rows of 34 (plane) / 32 (dz) records, whatever the
//    crowd size: every LDS address of the staging tasks and the border rows are loop constants, written / computed once.
//  * Two workgroup shapes: 10 waves (two share a K-step of 32 positions and split its column tiles) with two images, two
//    per CU, one barrier per item -- fp32 storage; 5 waves (a wave owns a K-step and ALL column tiles: the dz operand is
//    fetched once per K-step) with ONE image, four per CU, two barriers per item -- bf16 storage.  The MFMA phase runs at
//    raised wave priority (s_setprio): the waves past the barrier win the issue slots over the ones still splitting.
//  * The six MFMAs of a tile accumulate IN PLACE (one asm statement with a read-write accumulator).  Reading tile t+1's
//    operand in front of tile t's MFMAs (a second operand set) was measured: it spills at the 80 registers two 10-wave
//    workgroups per CU allow (54.7 us against 47.4).
#include "txp_conv_bf16.hpp"
#include "txp_wgrad.hpp"
#include "scene_team.hpp"
#include <type_traits>

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P;
constexpr int kRec = 96;                        // bytes of a position record: [h | m | l][12 bf16 + 4 pad]
constexpr int kPiece = 32;
constexpr int SWI = kWgradChunkV + 2, VWI = kWgradChunkV;      // image rows: plane / dz records
constexpr int kARecs = (C + 2) * SWI;           // plane image incl. its zero rows 0 and C + 1
constexpr int kZRecs = C * VWI;
constexpr int kZeroRec = kARecs + kZRecs;       // dz operand of a K slot past the item's last position
constexpr int kOnesRec = kZeroRec + 1;          // plane operand of the bias column
constexpr int kImageBytes = (kOnesRec + 1) * kRec;             // 38,400
constexpr int kATasks = C * SWI * 3, kZTasks = C * VWI * 3;    // 16-byte quads of a full-width item: 510 + 480

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Item {
    int vc, w0;            // columns of the item (a whole scene, or a <= 32-column chunk of a larger one) and its first column
    int swf, vwf;          // row strides (positions) of the scene's SAVED plane / dz arrays
    bool valid;
    const float *pl, *dz;
};

struct Op {                // one operand: 8 K slots of the lane's channel, three pieces
    u32x2 h[2], m[2], l[2];
};
__device__ __forceinline__ u32x4 cat(const u32x2 (&p)[2]) { return u32x4{p[0].x, p[0].y, p[1].x, p[1].y}; }

// the transposing reads of an operand AND their wait are one asm statement (no register with data in flight is ever visible
// to the register allocator); OFF: byte offset of the record from the lane's base address (an instruction immediate)
template <int OFF>
__device__ __forceinline__ void read_op(unsigned a0, unsigned a1, Op &o) {
    asm volatile("ds_read_b64_tr_b16 %0, %6 offset:%8\n\tds_read_b64_tr_b16 %1, %7 offset:%8\n\t"
                 "ds_read_b64_tr_b16 %2, %6 offset:%9\n\tds_read_b64_tr_b16 %3, %7 offset:%9\n\t"
                 "ds_read_b64_tr_b16 %4, %6 offset:%10\n\tds_read_b64_tr_b16 %5, %7 offset:%10\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o.h[0]), "=&v"(o.h[1]), "=&v"(o.m[0]), "=&v"(o.m[1]), "=&v"(o.l[0]), "=&v"(o.l[1])
                 : "v"(a0), "v"(a1), "n"(OFF), "n"(OFF + kPiece), "n"(OFF + 2 * kPiece)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void read_op_h(unsigned a0, unsigned a1, Op &o) {
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%4\n\tds_read_b64_tr_b16 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o.h[0]), "=&v"(o.h[1])
                 : "v"(a0), "v"(a1), "n"(OFF)
                 : "memory");
}
__device__ __forceinline__ f32x4 mma(const u32x4 &a, const u32x4 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cv::bf16x8, a), __builtin_bit_cast(cv::bf16x8, b), c, 0,
                                                   0, 0);
}
// the six products of one column tile, accumulated IN PLACE: as an asm statement with a read-write accumulator (through the
// builtin hipcc renamed the accumulators from tile to tile inside the item loop -- copies at the loop edges, 21 spilled
// registers).  hipcc's hazard recognizer cannot see inside the statement, so it carries its own wait states: s_nop 3 in
// front (a VALU write of an accumulator / operand register needs wait states before an MFMA reads it: two were enough in
// every arrangement measured, four are issued -- found the hard
// way: a peeled first iteration put an accumulator's zero-initialising v_mov right in front of the statement and the sums
// were garbage) and s_nop 7 behind (MFMA result -> VALU / LDS read: 4 passes + 3).
__device__ __forceinline__ void mma6(const u32x4 &zh, const u32x4 &zm, const u32x4 &zl, const Op &a, f32x4 &acc) {
    const u32x4 ah = cat(a.h), am = cat(a.m), al = cat(a.l);
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %2, %4, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %1, %5, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %2, %5, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %1, %6, %0\n\t"
                 "s_nop 7"
                 : "+v"(acc)
                 : "v"(zh), "v"(zm), "v"(zl), "v"(ah), "v"(am), "v"(al));
}
__device__ __forceinline__ void mma1(const u32x4 &zh, const u32x4 &ah, f32x4 &acc) {
    asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 7" : "+v"(acc) : "v"(zh), "v"(ah));
}
template <bool BF> struct StageType { typedef f32x4 type; };
template <> struct StageType<true> { typedef u32x2 type; };

// NW waves per workgroup (5: a wave owns a K-step and every column tile; 10: two waves share a K-step's tiles);
// NIMG LDS images per workgroup (1: two barriers per item; 2: one)
template <int CINL, bool BF, bool CH, int NW, int NIMG>
__device__ __forceinline__ void layer(const WgradArgs &a, const int32_t *__restrict__ order,
                                      const int32_t *__restrict__ order_peds, const int32_t *__restrict__ num_peds,
                                      const int32_t *__restrict__ key_start, int layer, unsigned char *sm, int wg, int nwg) {
    constexpr int NTH = NW * 64;
    constexpr int HS = NW / 5;                       // waves sharing a K-step
    constexpr int QPT = CINL / 4;                    // channel quads of a tap
    constexpr int NQ = 9 * QPT;                      // quads before the bias quad
    constexpr int NT = (9 * CINL + 1 + 15) / 16;     // column tiles: 7 | 5
    constexpr int T0 = HS == 1 ? NT : (NT + 1) / 2;  // tiles of the first wave of a K-step
    constexpr int TA = (kATasks + NTH - 1) / NTH, TZ = (kZTasks + NTH - 1) / NTH;
    const ModelLayout &L = a.lay;
    const int V = a.V, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave / HS, hf = wave - ks * HS;
    const int kg = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;   // tr reads: K sub-chunk, position and channel quad
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sm;

    // ---- work items (as in txp_wgrad_bf16.hip): chunk c of a scene exists when the scene has more than 32 c pedestrians;
    // with the sorted scene list the items are compact [chunk 0 of all scenes | chunk 1 of those with more than 32 | ...]
    const int nch = CH ? wgrad_chunks(V) : 1;
    const bool compact = CH && order && key_start && nch > 1;
    int items = a.N * nch;
    if (compact) {
        items = a.N;
        for (int c = 1; c < nch; ++c) {
            const int more = key_start[V - kWgradChunkV * c];
            items += more < 0 ? 0 : (more > a.N ? a.N : more);
        }
    }
    const int64_t plane_off = ws_plane_off(L, V, layer), dzs_floats = dz_slot(V);
    const bool bf = BF;
    auto fetch = [&](int r) -> Item {
        Item it{0, 0, 0, 0, false, nullptr, nullptr};
        const int at = walk_item(r, wg, nwg, items, order != nullptr && a.serpentine);
        if (at < 0) return it;
        int si = at, chunk = 0;
        if (compact) {
            int cnt = a.N;
            while (si >= cnt) {
                si -= cnt;
                ++chunk;
                cnt = key_start[V - kWgradChunkV * chunk];
                cnt = cnt < 0 ? 0 : (cnt > a.N ? a.N : cnt);
                if (chunk >= nch - 1) break;
            }
            if (si >= a.N) si = a.N - 1;
        } else if (nch > 1) {
            si = at / nch;
            chunk = at - si * nch;
        }
        const int n = order ? scene_index(order[si], a.N) : si;
        int vfull = order ? order_peds[si] : (num_peds ? num_peds[si] : V);
        vfull = vfull < 0 ? 0 : (vfull > V ? V : vfull);
        const int nc = CH ? wgrad_chunks(vfull) : 1;
        if (vfull == 0 || chunk >= nc) return it;
        const int wc = nc > 1 ? (vfull + nc - 1) / nc : vfull;
        it.pl = a.ws + n * a.ws_stride + plane_off;
        it.dz = a.dzg + ((int64_t)n * (L.L + 1) + layer) * dzs_floats;
        it.w0 = chunk * wc;
        it.vc = (vfull - it.w0) < wc ? (vfull - it.w0) : wc;
        it.swf = save_sw(vfull, bf);
        it.vwf = save_vw(vfull, bf);
        it.valid = true;
        return it;
    };

    // ---- staging tasks: slot u of a thread is one 16-byte quad (4 channels of a position) of the image, always the same
    // one.  Plane slots and dz slots are separate, so that a slot's global base is wave-uniform.
    constexpr int TF = BF ? 2 : 4;                   // floats per task
    using StageV = typename StageType<BF>::type;
    struct Stage { StageV a[TA], z[TZ]; };
    unsigned dsta[TA], dstz[TZ];                     // LDS byte offsets (within an image)
    int rowa[TA], cola[TA], rowz[TZ], colz[TZ];      // row * 3 TF | (column * 3 + quad) * TF of the slot (column < 0: no task)
#pragma unroll
    for (int u = 0; u < TA; ++u) {
        const int e = tid + u * NTH, rec = e / 3, q = e - rec * 3, row = rec / SWI, col = rec - row * SWI;
        dsta[u] = (unsigned)((SWI + rec) * kRec + 8 * q);
        rowa[u] = row * 3 * TF;
        cola[u] = e < kATasks ? ((col * 3 + q) * TF) | (col << 16) : -1;
    }
#pragma unroll
    for (int u = 0; u < TZ; ++u) {
        const int e = tid + u * NTH, rec = e / 3, q = e - rec * 3, row = rec / VWI, col = rec - row * VWI;
        dstz[u] = (unsigned)((kARecs + rec) * kRec + 8 * q);
        rowz[u] = row * 3 * TF;
        colz[u] = e < kZTasks ? ((col * 3 + q) * TF) | (col << 16) : -1;
    }
    auto load = [&](const Item &it, Stage &s) {
        const bool live = it.valid && !STG_SKIP(a, 64);
        const int lima = live ? it.vc + 2 : 0, limz = live ? it.vc : 0;
        const float *pa = (live ? it.pl : a.ws) + (live ? it.w0 * 3 * TF : 0);
        const float *pz = (live ? it.dz : a.ws) + (live ? it.w0 * 3 * TF : 0);
#pragma unroll
        for (int u = 0; u < TA; ++u) {
            const bool used = cola[u] >= 0 && (cola[u] >> 16) < lima;
            const int off = used ? rowa[u] * it.swf + (cola[u] & 0xffff) : 0;
            s.a[u] = *reinterpret_cast<const StageV *>(pa + off);      // (no task: the item's first bytes again)
        }
#pragma unroll
        for (int u = 0; u < TZ; ++u) {
            const bool used = colz[u] >= 0 && (colz[u] >> 16) < limz;
            const int off = used ? rowz[u] * it.vwf + (colz[u] & 0xffff) : 0;
            s.z[u] = *reinterpret_cast<const StageV *>(pz + off);
        }
    };
    auto put = [&](unsigned char *dst, const StageV &v) {
        if constexpr (BF) {
            *reinterpret_cast<uint2 *>(dst) = make_uint2(v.x, v.y);
        } else {
            uint2 ph, pm, pl;
            cv::split_pack4(v, ph, pm, pl);
            *reinterpret_cast<uint2 *>(dst) = ph;
            *reinterpret_cast<uint2 *>(dst + kPiece) = pm;
            *reinterpret_cast<uint2 *>(dst + 2 * kPiece) = pl;
        }
    };
    auto convert = [&](const Item &it, const Stage &s, unsigned char *img) {
        if (!it.valid || STG_SKIP(a, 64)) return;
#pragma unroll
        for (int u = 0; u < TA; ++u)
            if (cola[u] >= 0 && (cola[u] >> 16) < it.vc + 2) put(img + dsta[u], s.a[u]);
#pragma unroll
        for (int u = 0; u < TZ; ++u)
            if (colz[u] >= 0 && (colz[u] >> 16) < it.vc) put(img + dstz[u], s.z[u]);
    };

    // ---- operands ---------------------------------------------------------------------------------------------------
    // column tile k of this wave: the lane supplies channel quad cp of the tile = quad Q = 4 t + cp of the packed columns:
    // tap Q / QPT, channels 4 (Q % QPT) ..; Q >= NQ: the bias quad (and the unused quads behind it): the ones record
    int toff[T0];
    bool ones_last = false;
#pragma unroll
    for (int k = 0; k < T0; ++k) {
        const int t = hf * T0 + k, Q = 4 * t + cp;
        const int tap = Q / QPT, cq = Q - tap * QPT;
        toff[k] = ((tap / 3) * SWI + (tap % 3)) * kRec + 8 * cq;       // from the K slot's top-left neighbour
        if (Q >= NQ) {
            toff[k] = 0;
            if (t == NT - 1) ones_last = true;
        }
    }
    f32x4 acc[T0];
#pragma unroll
    for (int k = 0; k < T0; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // the K slots of this wave's step for an item of vc columns (recomputed only when vc changes)
    int vc_cur = -1;
    unsigned za[2] = {0u, 0u}, aa[2] = {0u, 0u};       // byte offsets within the image
    auto kslots = [&](int vc) {
        vc_cur = vc;
        const int npos = C * vc;
        const float inv = __builtin_amdgcn_rcpf((float)vc);
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            // lane group kg holds positions 4kg..4kg+3 and 16+4kg..16+4kg+3 of the step: the two groups of a 32-lane half
            // read 8 CONSECUTIVE 96-byte records, whose 32-byte piece rows tile the 64 banks exactly
            const int p = 32 * ks + 16 * rd + 4 * kg + rq;
            const bool ok = p < npos;
            const int hh = (int)(((float)p + 0.5f) * inv), ww = p - hh * vc;
            za[rd] = (unsigned)((ok ? kARecs + hh * VWI + ww : kZeroRec) * kRec + 8 * cp);
            aa[rd] = (unsigned)((ok ? hh * SWI + ww : 0) * kRec);               // plane record (row hh - 1, column ww - 1)
        }
    };
    auto addr = [&](int k, int rd, unsigned base) -> unsigned {
        const unsigned rel = base + aa[rd] + (unsigned)toff[k];
        // (only the last column tile holds the bias quad)
        return (hf * T0 + k == NT - 1 && ones_last) ? base + (unsigned)(kOnesRec * kRec + 8 * cp) : rel;
    };
    auto compute = [&](const Item &it, unsigned base) {
        if (!it.valid || STG_SKIP(a, 128)) return;
        if (32 * ks >= C * it.vc) return;
        if (it.vc != vc_cur) kslots(it.vc);
        __builtin_amdgcn_s_setprio(2);
        Op z, x0;
        u32x4 zh, zm = u32x4{0u, 0u, 0u, 0u}, zl = zm;
        if constexpr (BF) {
            read_op_h<0>(base + za[0], base + za[1], z);
            zh = cat(z.h);
        } else {
            read_op<0>(base + za[0], base + za[1], z);
            zh = cat(z.h); zm = cat(z.m); zl = cat(z.l);
        }
#pragma unroll
        for (int k = 0; k < T0; ++k) {
            if (hf * T0 + k < NT) {
                if constexpr (BF) {
                    read_op_h<0>(addr(k, 0, base), addr(k, 1, base), x0);
                    mma1(zh, cat(x0.h), acc[k]);
                } else {
                    read_op<0>(addr(k, 0, base), addr(k, 1, base), x0);
                    mma6(zh, zm, zl, x0, acc[k]);
                }
            }
        }
        (void)zm; (void)zl;
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- constants of the images: everything zero (the plane's border rows stay so), the ones record --------------------
    for (int i = 0; i < NIMG; ++i) {
        unsigned char *img = sm + i * kImageBytes;
        constexpr int U = kRec / 8;                    // 8-byte units of a record
        for (int e = tid; e < (kOnesRec + 1) * U; e += NTH)
            *reinterpret_cast<uint2 *>(img + e * 8) =
                (e >= kOnesRec * U && e < kOnesRec * U + kPiece / 8) ? make_uint2(0x3f803f80u, 0x3f803f80u) : make_uint2(0u, 0u);
    }
    team_barrier();                                    // (the first item's records land on top of these zeros)

    // ---- the pipeline -------------------------------------------------------------------------------------------------
    const int rounds = (items + nwg - 1) / nwg;        // (uniform over the workgroup: the barriers match)
    Item cur = fetch(0), nxt = fetch(1);
    Stage s;
#pragma unroll
    for (int u = 0; u < TA; ++u) s.a[u] = StageV{};
#pragma unroll
    for (int u = 0; u < TZ; ++u) s.z[u] = StageV{};
    load(cur, s);
    for (int r = 0; r < rounds; ++r) {
        const int ib = NIMG == 1 ? 0 : (r & 1) * kImageBytes;
        if (NIMG == 1 && r > 0) team_barrier();        // every wave has read the previous item
        convert(cur, s, sm + ib);
        load(nxt, s);
        team_barrier();
        compute(cur, lds0 + (unsigned)ib);
        cur = nxt;
        nxt = fetch(r + 2);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // each wave parks its tiles in its own LDS row ([P][CINL][9] weights + [P] biases: the parameters' own order); an
    // entry is the sum over the five K-step waves that own its tile
    constexpr int ROW = (P * CINL * 9 + P + 3) & ~3;
    float *rowsm = reinterpret_cast<float *>(sm);
    float *row = rowsm + wave * ROW;
    const int nq = lane & 15, kq = lane >> 4;          // accumulator: column within the tile, row quad (out-channels 4kq..)
    if (kq < 3) {
#pragma unroll
        for (int k = 0; k < T0; ++k) {
            const int t = hf * T0 + k;
            if (t < NT) {
                const int jj = 16 * t + nq;
                const int tap = jj / CINL, ci = jj - tap * CINL;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = 4 * kq + r;
                    if (jj < 9 * CINL) row[(co * CINL + ci) * 9 + tap] = acc[k][r];
                    else if (jj == 9 * CINL) row[P * CINL * 9 + co] = acc[k][r];
                }
            }
        }
    }
    __syncthreads();
    const int len = wgrad_row_len(layer);
    float *dst = a.slab2 + wgrad_slab_base(layer, a.rows) + (int64_t)wg * len;
    for (int e = tid; e < len; e += NTH) {
        int jj = 9 * CINL;
        if (e < P * CINL * 9) {
            const int tap = e % 9, ci = (e / 9) % CINL;
            jj = tap * CINL + ci;
        }
        const int h = (jj >> 4) >= T0 ? 1 : 0;
        float t = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < 5; ++sidx) t += rowsm[(sidx * HS + h) * ROW + e];
        dst[e] = t;
    }
}

template <bool BF, bool CH, int NW, int NIMG>
__global__ __launch_bounds__(NW * 64, NW == 5 ? 5 : 6) void txp_wgrad_bf16_kernel(const WgradArgs a, const int32_t *__restrict__ order,
                                                                   const int32_t *__restrict__ order_peds,
                                                                   const int32_t *__restrict__ num_peds,
                                                                   const int32_t *__restrict__ key_start) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    int l = 0;
    while (l < a.lay.L && (int)blockIdx.x >= a.wg_begin[l + 1]) ++l;
    const int wg = (int)blockIdx.x - a.wg_begin[l];
    const int nwg = a.wg_begin[l + 1] - a.wg_begin[l];
    if (l == 0)
        layer<Cfg::T, BF, CH, NW, NIMG>(a, order, order_peds, num_peds, key_start, l, smb, wg, nwg);
    else
        layer<Cfg::P, BF, CH, NW, NIMG>(a, order, order_peds, num_peds, key_start, l, smb, wg, nwg);
}


}  // namespace

bool wgrad_bf16_fits(const ModelLayout &L, int V) {
    (void)V;                                           // (every V: larger scenes are cut into <= 32-column chunks)
    return !(L.flags & STG_OPT_F32_MFMA);
}

// workgroup shape: fp32 storage: 10 waves (two share a K-step's column tiles) and two images, two workgroups per CU;
// bf16 storage (one piece, one product: 58 registers): 5 waves (a wave owns a K-step and all column tiles) and one
// image, four per CU -- measured at the headline shape 35.0 against 32.0 us (fp32 storage the other way round: the
// 5-wave form needs 96 registers and spills, 68 against 48.6 us)
void wgrad_bf16_geom(WgradGeom *g, const ModelLayout &L, int V) {
    (void)V;
    const bool bf = (L.flags & STG_OPT_BF16_STORE) != 0;
    g->waves = bf ? 5 : 10;
    g->nbuf = bf ? 1 : 2;
    const size_t row = (size_t)((Cfg::P * Cfg::P * 9 + Cfg::P + 3) & ~3) * sizeof(float) * g->waves;
    size_t lds = (size_t)g->nbuf * kImageBytes;
    if (lds < row) lds = row;
    g->lds = lds;
}

int launch_txp_wgrad_bf16(const WgradArgs &w, const WgradGeom &g, hipStream_t st) {
    const dim3 grid(g.grid), block(g.waves * 64);
    const bool bf = (w.lay.flags & STG_OPT_BF16_STORE) != 0, ch = w.V > kWgradChunkV;
#define STG_LW3(B, H, NW, NI)                                                                                         \
    do {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_wgrad_bf16_kernel<B, H, NW, NI>),                 \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);                   \
        if (e != hipSuccess) return hip_fail(e, "txp_wgrad_bf16: hipFuncSetAttribute");                                \
        hipLaunchKernelGGL((txp_wgrad_bf16_kernel<B, H, NW, NI>), grid, block, g.lds, st, w, w.order, w.order_peds, w.num_peds,  \
                           w.key_start);                                                                              \
    } while (0)
#define STG_LW3V(NW, NI)                                                                                              \
    do {                                                                                                              \
        if (bf) { if (ch) STG_LW3(true, true, NW, NI); else STG_LW3(true, false, NW, NI); }                           \
        else { if (ch) STG_LW3(false, true, NW, NI); else STG_LW3(false, false, NW, NI); }                            \
    } while (0)
#ifdef STG_K2_ALL_SHAPES                               // (tools/micro/k2_bench.hip: every storage mode in both workgroup shapes)
    if (g.waves == 5 && g.nbuf == 1) STG_LW3V(5, 1);
    else if (g.waves == 10 && g.nbuf == 2) STG_LW3V(10, 2);
    else return fail(STG_EINVAL, "txp_wgrad_bf16: no kernel for %d waves / %d images", g.waves, g.nbuf);
#else
    if (bf && g.waves == 5 && g.nbuf == 1) { if (ch) STG_LW3(true, true, 5, 1); else STG_LW3(true, false, 5, 1); }
    else if (!bf && g.waves == 10 && g.nbuf == 2) { if (ch) STG_LW3(false, true, 10, 2); else STG_LW3(false, false, 10, 2); }
    else return fail(STG_EINVAL, "txp_wgrad_bf16: no kernel for %d waves / %d images", g.waves, g.nbuf);
#endif
#undef STG_LW3V
#undef STG_LW3
    STG_LAUNCH_CHECK("txp_wgrad_bf16");
    return STG_OK;
}

}  // namespace stg
