// txp_wgrad_bf16 (K2, whole-scene fp32 items): the TXP-CNN weight / bias gradients
//        dW_l[co][ci][tap] = sum_{scene,pos} dz_l[co][pos] a_l[ci][pos+tap],   db_l = sum dz_l
// on v_mfma_f32_16x16x32_bf16 with fp32-exact operands: every fp32 value is split into three bf16 pieces (txp_conv_bf16.hpp:
// x = x_h + x_m + x_l exactly) and the six products that reach 2^-24 are accumulated in fp32.  On gfx950 the fp32 MFMA
// (v_mfma_f32_16x16x4_f32, what txp_wgrad.hip issues) holds the SIMD's VALU port for all of its 32 cycles; the bf16 MFMAs
// take 16 cycles of the matrix pipe for 8x the K and run beside the VALU / LDS instructions of the other waves
// (tools/micro/mfma_valu_overlap.hip).  Per 32 positions: 9 taps x 6 products x 16 = 864 pipe cycles instead of
// 8 x 7 x 32 = 1792 VALU-blocking ones.
//
// GEMM shape: M = 12 out-channels (dz, 16-row tile), N = 16 input channels of ONE tap (12 real; nine tap tiles + a
// ones tile for the bias), K = 32 positions per MFMA.  Both operands want 8 consecutive positions of one channel per
// lane, the saved arrays are position-major ([pos][12], what the forward / input-gradient kernels write with 16-byte
// stores): the LDS image keeps them position-major -- 96-byte records [3 pieces][12 channels + pad] -- and the operands are
// fetched with ds_read_b64_tr_b16, the transposing LDS read (4 positions x 16 channels -> lane = channel, 4 positions;
// tools/micro/tr_read_probe.hip), so a tap shift is a whole number of records and every read is 8-byte aligned.
//
// A workgroup of 10 waves owns a layer and walks its share of the scenes: wave w takes K-step w >> 1 (32 positions;
// a 32-pedestrian scene has 5) and half of the tap tiles (w & 1: taps 0..4 | taps 5..8 + bias), accumulates in VGPRs
// for the whole launch, and the partial sums meet in LDS once at the end (fixed order, no atomics; same slab rows as
// txp_wgrad.hip).  The fp32 arrays of a scene are fetched into registers TWO scenes ahead (plain global loads, 16
// bytes per lane), split and written into one of two LDS images one scene ahead of its MFMAs: one s_barrier per scene.
#include "txp_conv_bf16.hpp"
#include "txp_wgrad.hpp"
#include "scene_team.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P;
constexpr int kWavesB = 10;
constexpr int kRec = 96;                       // bytes of a position record: [h | m | l][12 bf16 + 4 pad]
constexpr int kPiece = 32;                     // bytes between the pieces of a record
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// bytes of one LDS image for batch width V: plane (C+2)(V+2) records, dz C*V records + one zero record
// (bf16 storage, STG_OPT_BF16_STORE: the saved rows are padded to an even number of positions -- save_sw / save_vw)
__host__ __device__ inline int image_a_recs(int V, bool bf) { return (C + 2) * save_sw(V, bf); }
__host__ __device__ inline int image_bytes(int V, bool bf) {
    return ((image_a_recs(V, bf) + C * save_vw(V, bf) + 1) * kRec + 15) & ~15;
}

template <bool BF> struct StageType { typedef f32x4 type; };
template <> struct StageType<true> { typedef unsigned type __attribute__((ext_vector_type(2))); };

struct Item {
    int vi, w0, vc;        // pedestrians of the scene; first column and width of the chunk (vc == vi: the whole scene)
    bool valid;
    const float *pl, *dz;
};

struct Op3 {            // one operand chunk (8 positions of the lane's channel), three pieces
    u32x2 h[2], m[2], l[2];
};
// The transposing reads of an operand AND their s_waitcnt are ONE asm statement: the compiler never sees a register
// whose data is still in flight (a copy inserted between a read and a separate wait statement would pick up stale
// data).  The other waves of the SIMD cover the LDS latency.  (offset immediates: the three pieces of a record)
__device__ __forceinline__ void read_op(unsigned a0, unsigned a1, Op3 &o) {
    asm volatile("ds_read_b64_tr_b16 %0, %6\n\tds_read_b64_tr_b16 %1, %7\n\t"
                 "ds_read_b64_tr_b16 %2, %6 offset:32\n\tds_read_b64_tr_b16 %3, %7 offset:32\n\t"
                 "ds_read_b64_tr_b16 %4, %6 offset:64\n\tds_read_b64_tr_b16 %5, %7 offset:64\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o.h[0]), "=&v"(o.h[1]), "=&v"(o.m[0]), "=&v"(o.m[1]), "=&v"(o.l[0]), "=&v"(o.l[1])
                 : "v"(a0), "v"(a1)
                 : "memory");
}
// bf16 storage: the h piece only
__device__ __forceinline__ void read_op_h(unsigned a0, unsigned a1, Op3 &o) {
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o.h[0]), "=&v"(o.h[1])
                 : "v"(a0), "v"(a1)
                 : "memory");
}
__device__ __forceinline__ f32x4 mma(const u32x2 (&a)[2], const u32x2 (&b)[2], const f32x4 &c) {
    const u32x4 av = {a[0].x, a[0].y, a[1].x, a[1].y}, bv = {b[0].x, b[0].y, b[1].x, b[1].y};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cv::bf16x8, av), __builtin_bit_cast(cv::bf16x8, bv), c,
                                                   0, 0, 0);
}

// image rows of a COLUMN CHUNK (a work item that is part of a larger scene): fixed widths, whatever the chunk's own --
// the staging index arithmetic then divides by constants only
constexpr int kChunkSW = kWgradChunkV + 2, kChunkVW = kWgradChunkV;

// CH: the batch is padded beyond kWgradChunkV pedestrians, scenes may be cut into column chunks (the instantiation for
// batches of whole scenes carries none of the chunk arithmetic: the kernel sits at its 80-register budget)
template <int CINL, bool BF, bool CH>
__device__ __forceinline__ void wgrad_bf16_layer(const WgradArgs &a, const int32_t *__restrict__ order,
                                                 const int32_t *__restrict__ order_peds,
                                                 const int32_t *__restrict__ num_peds,
                                                 const int32_t *__restrict__ key_start, int layer, unsigned char *sm, int wg,
                                                 int nwg) {
    const ModelLayout &L = a.lay;
    const int V = a.V, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ks = wave >> 1, hf = wave & 1;                    // K-step and tap half of this wave
    const int kg = lane >> 4, rq = (lane & 15) >> 2, cp = lane & 3;   // operand chunk, row and column quad of the tr reads
    const int nq = lane & 15, kq = lane >> 4;                   // accumulator: column (input channel), row quad
    const int Vc = wgrad_image_v(V);                   // widest work item: a scene, or a <= 32-column chunk of a larger one
    const int img = image_bytes(Vc, BF);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)sm;
    const int nch = CH ? wgrad_chunks(V) : 1;
    // Work items: chunk c of a scene exists when the scene has more than 32 c pedestrians.  With the sorted scene list the
    // items are COMPACT -- [chunk 0 of all N scenes | chunk 1 of the scenes with more than 32 | chunk 2 of those with more
    // than 64 | ...], the counts come from the list's tier offsets -- so that a ragged batch padded to 57 pays for the
    // second chunk of its few large scenes only; without the list every scene has nch item slots.
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

    f32x4 acc[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Raw { int at, n, v, chunk; };
    auto fetch = [&](int r) -> Raw {
        Raw w{-1, 0, 0, 0};
        w.at = walk_item(r, wg, nwg, items, order != nullptr && a.serpentine);
        if (w.at >= 0) {
            int si = w.at;
            if (compact) {
                int cnt = a.N;
                while (si >= cnt) {                    // (at most nch - 1 steps; the counts are wave-uniform scalar loads)
                    si -= cnt;
                    ++w.chunk;
                    cnt = key_start[V - kWgradChunkV * w.chunk];
                    cnt = cnt < 0 ? 0 : (cnt > a.N ? a.N : cnt);
                    if (w.chunk >= nch - 1) break;
                }
                if (si >= a.N) si = a.N - 1;           // (only with a tail that is not this batch's: stay inside the list)
            } else if (nch > 1) {
                si = w.at / nch;
                w.chunk = w.at - si * nch;
            }
            w.n = order ? scene_index(order[si], a.N) : si;
            w.v = order ? order_peds[si] : (num_peds ? num_peds[si] : V);
        }
        return w;
    };
    auto finish = [&](const Raw &w) -> Item {
        Item it{0, 0, 0, false, nullptr, nullptr};
        if (w.at < 0) return it;
        const int chunk = w.chunk;
        const int vfull = w.v < 0 ? 0 : (w.v > V ? V : w.v);
        const int nc = CH ? wgrad_chunks(vfull) : 1;   // a scene of more than 32 pedestrians is cut into equal column chunks
        if (vfull == 0 || chunk >= nc) return it;
        const int wc = nc > 1 ? (vfull + nc - 1) / nc : vfull;
        it.pl = a.ws + w.n * a.ws_stride + plane_off;
        it.dz = a.dzg + ((int64_t)w.n * (L.L + 1) + layer) * dzs_floats;
        it.vi = vfull;
        it.w0 = chunk * wc;
        it.vc = (vfull - it.w0) < wc ? (vfull - it.w0) : wc;
        it.valid = true;
        return it;
    };
    // ---- staging: task e of a scene = one 16-byte quad (4 channels of a position) of the plane rows or of dz ----------
    // plane: C rows x (vi + 2) columns x 3 quads, saved [h][col][12] fp32; dz: C*vi positions x 3 quads.  At most
    // 2 tasks per thread (kWgradChunkV = 32 pedestrians: 990 tasks, 640 threads).
    // The loads are plain C++ loads the compiler tracks: it waits for a set's registers with COUNTED s_waitcnt vmcnt(N) right
    // where convert() first reads them, so the other set's loads stay in flight -- also across the scene's barrier, because
    // that barrier is team_barrier() (s_waitcnt lgkmcnt(0) + s_barrier as one asm statement: hipcc drains vmcnt(0) in front
    // of every s_barrier it can see, which would retire the two scenes in flight at each scene's barrier).  (Round 2 issued
    // these loads from inline assembly with a hand-counted wait: a register with data in flight was then visible to the
    // register allocator, and nothing but a "+v" constraint kept it from being copied early.  Same speed, no hazard.)
    // Lanes without a task, and rounds past the last scene, re-read the first bytes of the workspace.
    // bf16 storage (BF): the saved arrays ARE the h pieces (24-byte positions, 8-byte quads); m = l = 0 and only the
    // h x h product is issued -- a task moves 8 bytes, nothing is split.
    using StageV = typename StageType<BF>::type;       // what one task fetches: 16 bytes (fp32 quad) or 8 (bf16 quad)
    struct Stage { StageV v[2]; };
    auto load = [&](const Item &it, Stage &s) {
        const bool live = it.valid && !STG_SKIP(a, 64);
        const bool whole = !CH || it.vc == it.vi;
        // image rows: the whole scene's saved rows as they are, or the chunk's vc + 2 plane columns / vc dz columns in rows of
        // the fixed chunk widths
        const int SWi = whole ? save_sw(it.vi, BF) : kChunkSW, VWi = whole ? save_vw(it.vi, BF) : kChunkVW;
        const int SWf = save_sw(it.vi, BF), VWf = save_vw(it.vi, BF);
        const int na = live ? C * SWi * 3 : 0, nz = live ? C * VWi * 3 : 0;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * kWavesB * 64;
            constexpr int TF = BF ? 2 : 4;             // floats per task
            int ea = e, ez = e - na;                   // quad index in the saved plane / dz array
            bool used = true;
            if (!whole) {
                // chunk: (row, column, quad) of the image -> the scene's saved row, shifted by the chunk's first column
                const int ra = e / 3, qa = e - ra * 3, ha = ra / kChunkSW, ca = ra - ha * kChunkSW;
                ea = (ha * SWf + it.w0 + ca) * 3 + qa;
                const int rz = ez / 3, qz = ez - rz * 3, hz = rz / kChunkVW, cz = rz - hz * kChunkVW;
                ez = (hz * VWf + it.w0 + cz) * 3 + qz;
                used = e < na ? ca < it.vc + 2 : cz < it.vc;          // (columns past the chunk's own are never read)
            }
            const float *src = !used ? a.ws : (e < na ? it.pl + TF * ea : (e < na + nz ? it.dz + TF * ez : a.ws));
            s.v[u] = *reinterpret_cast<const StageV *>(src);
        }
    };
    auto convert = [&](const Item &it, const Stage &s, unsigned char *buf) {
        if (!it.valid || STG_SKIP(a, 64)) return;
        const bool whole = !CH || it.vc == it.vi;
        const int SWa = whole ? save_sw(it.vi, BF) : kChunkSW, VWz = whole ? save_vw(it.vi, BF) : kChunkVW;
        const int na = C * SWa * 3, nz = C * VWz * 3;
        unsigned char *dzimg = buf + image_a_recs(Vc, BF) * kRec;
        // zero border rows of the plane image (rows 0 and C + 1) and the zero record behind dz: 8-byte stores
        constexpr int U = kRec / 8;
        for (int e = tid; e < 2 * SWa * U + U; e += kWavesB * 64) {
            unsigned char *dst;
            if (e < SWa * U) dst = buf + e * 8;
            else if (e < 2 * SWa * U) dst = buf + (C + 1) * SWa * kRec + (e - SWa * U) * 8;
            else dst = dzimg + C * VWz * kRec + (e - 2 * SWa * U) * 8;
            *reinterpret_cast<uint2 *>(dst) = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = tid + u * kWavesB * 64;
            if (e < na + nz) {
                const bool isz = e >= na;
                const int f = isz ? e - na : e, rec = f / 3, q = f - rec * 3;
                // plane rows h = 0..C-1 land in image rows 1..C: record index + SWa
                unsigned char *dst = (isz ? dzimg + rec * kRec : buf + (rec + SWa) * kRec) + 8 * q;
                if constexpr (BF) {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(s.v[u].x, s.v[u].y);
                } else {
                    uint2 ph, pm, pl;
                    cv::split_pack4(s.v[u], ph, pm, pl);
                    *reinterpret_cast<uint2 *>(dst) = ph;
                    *reinterpret_cast<uint2 *>(dst + kPiece) = pm;
                    *reinterpret_cast<uint2 *>(dst + 2 * kPiece) = pl;
                }
            }
        }
    };
    // ---- this wave's K-step of the scene staged in `buf` ------------------------------------------------------------
    auto compute = [&](const Item &it, unsigned buf_off) {
        if (!it.valid || STG_SKIP(a, 128)) return;
        const bool whole = !CH || it.vc == it.vi;
        const int vi = it.vc, npos = C * vi;           // (the K loop of a chunk is that of a scene of vc pedestrians)
        const int SWa = whole ? save_sw(vi, BF) : kChunkSW, VWz = whole ? save_vw(vi, BF) : kChunkVW;
        if (32 * ks >= npos) return;
        unsigned inv = (unsigned)(65536.0f * __builtin_amdgcn_rcpf((float)vi));
        while (inv * (unsigned)vi < 65536u) ++inv;
        while ((inv - 1u) * (unsigned)vi >= 65536u) --inv;
        const unsigned abase = lds0 + buf_off, zbase = abase + image_a_recs(Vc, BF) * kRec;
        unsigned za[2], aa[2];
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            // K slots: lane group kg holds positions 4kg..4kg+3 and 16+4kg..16+4kg+3 of the step, so that the two groups of
            // a 32-lane half read 8 CONSECUTIVE records: with 96-byte records (24 dwords = 3 x 8) their 32-byte rows
            // tile the 64 banks exactly
            const int p = 32 * ks + 16 * rd + 4 * kg + rq;
            const bool ok = p < npos;
            const int pc = ok ? p : 0;
            const int hh = (int)(((unsigned)pc * inv) >> 16), ww = pc - hh * vi;
            za[rd] = zbase + (ok ? hh * VWz + ww : C * VWz) * kRec + 8 * cp;          // past the end: the zero record
            aa[rd] = abase + (ok ? ((hh + 1) * SWa + (ww + 1)) * kRec : SWa * kRec + kRec) + 8 * cp;
        }
        // one operand register set: the LDS latency of a tap's six reads is covered by the other four waves of the SIMD
        // (five resident waves need <= 96 VGPRs; a second operand set spilled)
        const int ntap = hf == 0 ? 5 : 4;                  // tap tiles of this half (the bias tile needs no plane operand)
        Op3 dzo, ao;
        if (BF) {
            // bf16 storage: one piece, one product per tap
            read_op_h(za[0], za[1], dzo);
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (k < ntap) {
                    const int tap = hf * 5 + k;
                    const int shift = ((tap / 3 - 1) * SWa + (tap % 3 - 1)) * kRec;
                    read_op_h(aa[0] + shift, aa[1] + shift, ao);
                    acc[k] = mma(dzo.h, ao.h, acc[k]);
                } else {
                    const u32x2 one[2] = {u32x2{0x3f803f80u, 0x3f803f80u}, u32x2{0x3f803f80u, 0x3f803f80u}};
                    acc[k] = mma(dzo.h, one, acc[k]);
                }
            }
            return;
        }
        read_op(za[0], za[1], dzo);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            if (k < ntap) {
                const int tap = hf * 5 + k;
                const int shift = ((tap / 3 - 1) * SWa + (tap % 3 - 1)) * kRec;
                read_op(aa[0] + shift, aa[1] + shift, ao);
                acc[k] = mma(dzo.h, ao.h, acc[k]);
                acc[k] = mma(dzo.m, ao.h, acc[k]);
                acc[k] = mma(dzo.h, ao.m, acc[k]);
                acc[k] = mma(dzo.m, ao.m, acc[k]);
                acc[k] = mma(dzo.l, ao.h, acc[k]);
                acc[k] = mma(dzo.h, ao.l, acc[k]);
            } else {
                // bias: B = ones, every column of the tile becomes sum_pos dz[co][pos]
                const u32x2 one[2] = {u32x2{0x3f803f80u, 0x3f803f80u}, u32x2{0x3f803f80u, 0x3f803f80u}};
                acc[k] = mma(dzo.h, one, acc[k]);
                acc[k] = mma(dzo.m, one, acc[k]);
                acc[k] = mma(dzo.l, one, acc[k]);
            }
        }
    };

    // ---- the pipeline ---------------------------------------------------------------------------------------------
    const int rounds = (items + nwg - 1) / nwg;        // (uniform over the workgroup: the barriers match)
    Item q0 = finish(fetch(0)), q1 = finish(fetch(1)), q2 = finish(fetch(2));
    Raw nxt = fetch(3);
    Stage sa, sb;                                      // sa: the scene converted next, sb: the one after it
    sa.v[0] = sa.v[1] = sb.v[0] = sb.v[1] = StageV{};
    load(q0, sa);
    load(q1, sb);
    for (int r = 0; r < rounds; r += 2) {
        // round r: scene q0 from `sa` into image 0; refill sa with scene r + 2
        convert(q0, sa, sm);
        load(q2, sa);
        team_barrier();
        compute(q0, 0u);
        q0 = q1; q1 = q2; q2 = finish(nxt); nxt = fetch(r + 4);
        if (r + 1 >= rounds) break;
        // round r + 1: scene (now q0) from `sb` into image 1; refill sb with scene r + 3
        convert(q0, sb, sm + img);
        load(q2, sb);
        team_barrier();
        compute(q0, (unsigned)img);
        q0 = q1; q1 = q2; q2 = finish(nxt); nxt = fetch(r + 5);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // each wave parks its tap tiles in its own LDS row ([P][CINL][9] weights + [P] biases: the parameters' own order);
    // an entry is the sum over the five waves that own its tap half
    constexpr int ROW = (P * CINL * 9 + P + 3) & ~3;
    float *rowsm = reinterpret_cast<float *>(sm);
    float *row = rowsm + wave * ROW;
    if (kq < 3) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * kq + r;
                if (hf == 1 && k == 4) {
                    if (nq == 0) row[P * CINL * 9 + co] = acc[k][r];
                } else if (nq < CINL) {
                    row[(co * CINL + nq) * 9 + hf * 5 + k] = acc[k][r];
                }
            }
        }
    }
    __syncthreads();
    const int len = wgrad_row_len(layer);
    float *dst = a.slab2 + wgrad_slab_base(layer, a.rows) + (int64_t)wg * len;
    for (int e = tid; e < len; e += kWavesB * 64) {
        const int tap = e < P * CINL * 9 ? e % 9 : 9;
        const int h = tap >= 5 ? 1 : 0;
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 5; ++s) t += rowsm[(2 * s + h) * ROW + e];
        dst[e] = t;
    }
}

template <bool BF, bool CH>
__global__ __launch_bounds__(kWavesB * 64, 6) void txp_wgrad_bf16_kernel(const WgradArgs a, const int32_t *__restrict__ order,
                                                                         const int32_t *__restrict__ order_peds,
                                                                         const int32_t *__restrict__ num_peds,
                                                                         const int32_t *__restrict__ key_start) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    int layer = 0;
    while (layer < a.lay.L && (int)blockIdx.x >= a.wg_begin[layer + 1]) ++layer;
    const int wg = (int)blockIdx.x - a.wg_begin[layer];
    const int nwg = a.wg_begin[layer + 1] - a.wg_begin[layer];
    if (layer == 0)
        wgrad_bf16_layer<Cfg::T, BF, CH>(a, order, order_peds, num_peds, key_start, layer, smb, wg, nwg);
    else
        wgrad_bf16_layer<Cfg::P, BF, CH>(a, order, order_peds, num_peds, key_start, layer, smb, wg, nwg);
}

}  // namespace

// every V: a scene of more than kWgradChunkV pedestrians is cut into equal column chunks (as in txp_wgrad.hip)
bool wgrad_bf16_fits(const ModelLayout &L, int V) {
    const bool bf = (L.flags & STG_OPT_BF16_STORE) != 0;
    (void)V;
    return !(L.flags & STG_OPT_F32_MFMA) && 2 * (size_t)image_bytes(kWgradChunkV, bf) * 2 <= (size_t)kLdsBytes;
}

void wgrad_bf16_geom(WgradGeom *g, const ModelLayout &L, int V) {
    const size_t row = (size_t)((Cfg::P * Cfg::P * 9 + Cfg::P + 3) & ~3) * sizeof(float) * kWavesB;
    size_t lds = 2 * (size_t)image_bytes(wgrad_image_v(V), (L.flags & STG_OPT_BF16_STORE) != 0);
    if (lds < row) lds = row;
    g->waves = kWavesB;
    g->nbuf = 2;
    g->lds = lds;
}

int launch_txp_wgrad_bf16(const WgradArgs &w, const WgradGeom &g, hipStream_t st) {
    const dim3 grid(g.grid), block(kWavesB * 64);
    const bool bf = (w.lay.flags & STG_OPT_BF16_STORE) != 0, ch = w.V > kWgradChunkV;
#define STG_LW(B, H)                                                                                                  \
    do {                                                                                                              \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_wgrad_bf16_kernel<B, H>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);                   \
        if (e != hipSuccess) return hip_fail(e, "txp_wgrad_bf16: hipFuncSetAttribute");                              \
        hipLaunchKernelGGL((txp_wgrad_bf16_kernel<B, H>), grid, block, g.lds, st, w, w.order, w.order_peds, w.num_peds, \
                           w.key_start);                                                                              \
    } while (0)
    if (bf) { if (ch) STG_LW(true, true); else STG_LW(true, false); }
    else { if (ch) STG_LW(false, true); else STG_LW(false, false); }
#undef STG_LW
    STG_LAUNCH_CHECK("txp_wgrad_bf16");
    return STG_OK;
}

}  // namespace stg
