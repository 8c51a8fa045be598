// txp_conv_bf16: the 3x3 conv core of the wave-per-scene TXP kernels on the bf16 matrix pipe with fp32-exact operands.
//
// Why not v_mfma_f32_16x16x4_f32: on gfx950 the fp32 MFMA and the VALU exclude each other on a SIMD -- the time of a
// wave pair is (fp32 MFMA cycles) + (VALU cycles), measured in tools/micro/mfma_valu_overlap.hip (every v_fma beside an
// fp32 MFMA costs its full 4 cycles, with one or two waves per SIMD) -- whereas the bf16 MFMAs run beside the VALU
// (J <= 2 fillers free, and a VALU-bound stream hides them completely).  So the convs run on
// v_mfma_f32_16x16x32_bf16 with every fp32 operand split EXACTLY into three bf16 pieces
//     x = x_h + x_m + x_l        (8 + 8 + 8 significand bits; truncation splits, no rounding anywhere)
// and the six products that reach 2^-24:  W_h x_h + W_h x_m + W_m x_h + W_m x_m + W_h x_l + W_l x_h
// (dropped: W_m x_l, W_l x_m, W_l x_l <= 2^-23 |W x|, the size of one fp32 rounding; bf16 x bf16 products are exact
// in the fp32 accumulator).  Per K = 32 that is 6 x 16 = 96 matrix-pipe cycles instead of 8 x 32 = 256, and they
// overlap the epilogue's VALU work.
//
// Data layout (one wave's LDS image): POSITION-major records of 12 channels, one image per piece
//     piece s, row slot r (0..7, a ring), column c (-1..vi-1):   byte  off_s + ((r * SW + c + 1) * 12 + ch) * 2
//     SW = vi + 1: the zero column c = -1 of row r+1 doubles as the right border of row r
//     PL == 128 (mod 256) bytes (pieces at 0, PL, 2 PL + 128): the two pieces a 32-lane group reads in one ds_read_b64
//     sit on disjoint bank halves
// so that the (kw, ci) taps of one kernel row are 36 CONTIGUOUS bf16 of the image: K = (kh, [kw, ci] padded to 40)
// = 15 chunks of 8, fetched as 8-byte-aligned 16-byte pieces (no im2col copies, no gathers).
// The forward updates the image in place (model_common.hpp, txp_sci): a layer reads at slot offset 3 and writes at
// slot offset 1 or the reverse; with a ring of 8 slots the bottom border of the high image is slot 8 = slot 0, the
// permanently zero top border of the low one.
//
// K slots of the 8 MFMA groups of a 16-position tile (lane = (n = l & 15, kg = l >> 4); kg holds K values 8kg..8kg+7):
//     group g: lanes kg 0,1 take chunk 2g, lanes kg 2,3 chunk 2g+1 (chunk 15 does not exist: zero weights)
//     MFMA 1: B = [x_h | x_m] (even | odd kg)   A = [W_m | W_h]
//     MFMA 2: B = the same registers            A = [W_h | W_m]
//     MFMA 3: B = [x_h | x_l]                   A = [W_l | W_h]
// A operands come pre-arranged from txp_weight_prep (24 x 16 bytes per lane and layer).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stg {
namespace cv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kCh = 12;                 // channels per position record
constexpr int kPosBytes = 2 * kCh;      // 24
constexpr int kSlots = 8;               // row slots of the ring
constexpr int kGroups = 8;              // MFMA groups per tile
constexpr int kWpVecs = 3 * kGroups;    // 16-byte A-operand vectors per lane and layer
constexpr int kWpDwords = kWpVecs * 64 * 4;   // dwords of one prepared layer

__host__ __device__ inline int sw(int vi) { return vi + 1; }
__host__ __device__ inline int row_bytes(int vi) { return sw(vi) * kPosBytes; }
// bytes of one piece image: 8 row slots + the leading border position + 3 positions of read overrun, == 128 (mod 256)
__host__ __device__ inline int plane_bytes(int vi, int slots = kSlots) {
    const int r = (slots * sw(vi) + 4) * kPosBytes;
    return r + ((128 - r % 256) + 256) % 256;
}
// byte offset of the l piece image: 128 bytes past 2 PL, so that it too is == 128 (mod 256) away from the h image (the read
// set [x_h | x_l] pairs those two in one ds_read_b64: at 2 PL == 0 (mod 256) its two lane groups collided on every bank)
__host__ __device__ inline int l_off(int PL) { return 2 * PL + 128; }
__host__ __device__ inline int image_bytes(int vi, int slots = kSlots) { return 3 * plane_bytes(vi, slots) + 128; }
// byte offset (within a piece image) of the record of (row slot, column)
__host__ __device__ inline int pos_off(int vi, int slot, int col) { return (slot * sw(vi) + col + 1) * kPosBytes; }

// ---- exact three-way split ------------------------------------------------------------------------------
// pieces as fp32 values whose low 16 bits are zero (the bf16 pattern is the top half)
__host__ __device__ inline void split3(float x, float &h, float &m, float &l) {
    union { float f; unsigned u; } a, b;
    a.f = x;
    a.u &= 0xffff0000u;
    h = a.f;
    const float r = x - h;
    b.f = r;
    b.u &= 0xffff0000u;
    m = b.f;
    l = r - m;
}
__host__ __device__ inline unsigned hi16(float x) {
    union { float f; unsigned u; } a;
    a.f = x;
    return a.u >> 16;
}
// (hi.top16 << 16) | lo.top16 : one v_perm_b32
__device__ __forceinline__ unsigned pack_top(float lo, float hi) {
    return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
// four fp32 values -> three 8-byte records (pieces h, m, l of channels 4q..4q+3)
// (the two subtractions of split3 on float PAIRS: v_pk_add_f32 does two per instruction -- the same IEEE operations, the
// same bits: 18 instead of 22 vector instructions per quad)
__device__ __forceinline__ void split_pack4(const f32x4 &v, uint2 &ph, uint2 &pm, uint2 &pl) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    f32x2 h[2], m[2], l[2];
    // a - b on a float pair, as the instruction (hipcc packs only one of the two subtractions by itself)
    auto pk_sub = [](const f32x2 &a_, const f32x2 &b_) -> f32x2 {
        f32x2 d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a_), "v"(b_));
        return d;
    };
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const f32x2 x = {v[2 * r], v[2 * r + 1]};
        h[r] = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, x) & 0xffff0000u);
        const f32x2 rem = pk_sub(x, h[r]);
        m[r] = __builtin_bit_cast(f32x2, __builtin_bit_cast(u32x2, rem) & 0xffff0000u);
        l[r] = pk_sub(rem, m[r]);
    }
    ph = make_uint2(pack_top(h[0][0], h[0][1]), pack_top(h[1][0], h[1][1]));
    pm = make_uint2(pack_top(m[0][0], m[0][1]), pack_top(m[1][0], m[1][1]));
    pl = make_uint2(pack_top(l[0][0], l[0][1]), pack_top(l[1][0], l[1][1]));
}
// the inverse: exact (l + m fits 16 bits, + h fits 24)
__device__ __forceinline__ f32x4 unsplit4(const uint2 &ph, const uint2 &pm, const uint2 &pl) {
    auto lo = [](unsigned u) { return __uint_as_float(u << 16); };
    auto hi = [](unsigned u) { return __uint_as_float(u & 0xffff0000u); };
    return f32x4{(lo(pl.x) + lo(pm.x)) + lo(ph.x), (hi(pl.x) + hi(pm.x)) + hi(ph.x), (lo(pl.y) + lo(pm.y)) + lo(ph.y),
                 (hi(pl.y) + hi(pm.y)) + hi(ph.y)};
}

// ---- A operand: the K-slot map ----------------------------------------------------------------------------
// value i (0..7) of lane `lane` in MFMA k (0..2) of group g: W(m, ch, kh, kw) is the fp32 weight that multiplies input
// channel ch at tap (kh, kw) for output row m (forward: W[co=m][ci=ch][kh][kw]; input gradient:
// W[co=ch][ci=m][2-kh][2-kw]); rows / channels outside the layer are zero.
template <typename WF>
__host__ __device__ inline unsigned short wp_value(WF W, int g, int k, int lane, int i) {
    const int m = lane & 15, kg = lane >> 4;
    const int c = 2 * g + (kg >> 1);
    if (c >= 15) return 0;
    const int kh = c / 5, j = c - 5 * kh, e = 8 * j + i;
    if (e >= 36) return 0;
    const int kw = e / kCh, ch = e - kCh * kw;
    const float w = W(m, ch, kh, kw);
    float h, mm, l;
    split3(w, h, mm, l);
    const bool odd = kg & 1;
    const int piece = k == 0 ? (odd ? 0 : 1) : (k == 1 ? (odd ? 1 : 0) : (odd ? 0 : 2));
    return (unsigned short)hi16(piece == 0 ? h : (piece == 1 ? mm : l));
}

#ifdef __HIPCC__
// One 16-byte A-operand vector (vector v of layer-local table `wp`) of the INPUT-GRADIENT GEMM of a 3x3 conv with weights
// W [12][cinl][3][3]: d in[ci][pos] = sum W[co][ci][2-kh][2-kw] dz[co][pos+tap].  64 lanes.
__device__ __forceinline__ void prep_dgrad_vector(const float *__restrict__ W, int cinl, int v, int lane, unsigned *__restrict__ wp) {
    auto wf = [&](int m, int ch, int kh, int kw) -> float {
        return m < cinl ? W[(ch * cinl + m) * 9 + (2 - kh) * 3 + (2 - kw)] : 0.f;
    };
    unsigned d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        d[q] = (unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q) | ((unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q + 1) << 16);
    reinterpret_cast<u32x4 *>(wp)[v * 64 + lane] = u32x4{d[0], d[1], d[2], d[3]};
}
#endif

#ifdef __HIPCC__
// the same for the FORWARD conv: out[co][pos] = sum W[co][ci][kh][kw] in[ci][pos+tap]
__device__ __forceinline__ void prep_fwd_vector(const float *__restrict__ W, int cinl, int v, int lane, unsigned *__restrict__ wp) {
    auto wf = [&](int m, int ch, int kh, int kw) -> float {
        return (m < kCh && ch < cinl) ? W[(m * cinl + ch) * 9 + kh * 3 + kw] : 0.f;
    };
    unsigned d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        d[q] = (unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q) | ((unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q + 1) << 16);
    reinterpret_cast<u32x4 *>(wp)[v * 64 + lane] = u32x4{d[0], d[1], d[2], d[3]};
}
#endif

#ifdef __HIPCC__
typedef uint16_t ptab_t;

__device__ __forceinline__ void load_wp(const unsigned *__restrict__ wp, u32x4 (&w)[kWpVecs]) {
    const int lane = threadIdx.x & 63;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(wp) + lane;
#pragma unroll
    for (int v = 0; v < kWpVecs; ++v) w[v] = p[v * 64];
}

// per-lane constants of a scene
struct LaneGeom {
    unsigned c1, d2;     // byte offsets: read set 1 = base + c1 (piece + half-chunk), read set 2 = set 1 + d2
    int PL, RB;
};
__device__ __forceinline__ LaneGeom lane_geom(int vi, int slots = kSlots) {
    const int kg = (threadIdx.x & 63) >> 4;
    LaneGeom g;
    g.PL = plane_bytes(vi, slots);
    g.RB = row_bytes(vi);
    g.c1 = ((kg & 1) ? g.PL : 0) + (kg >> 1) * 16;
    g.d2 = (kg & 1) ? l_off(g.PL) - g.PL : 0;
    return g;
}

struct Tile {
    unsigned r1[4], r2[4];   // read bases of kernel rows 0, 1, 2 and of the mixed group (chunks 4 | 5)
    int h, w, pos;
    bool ok;
};

// IN0 = row slot of interior row 0 of the INPUT image (3: high image, 1: low image)
// the (row << 8 | column) code of the lane's position of a tile (position 0 past the scene's end), and the tile built from
// a code fetched earlier: the tile loops fetch tile t+1's code before tile t's operand reads, so that the table's LDS
// latency is not paid in front of every tile's address arithmetic
__device__ __forceinline__ unsigned tile_code(int tile, const ptab_t *ptab, int npos) {
    const int p = tile * 16 + (int)(threadIdx.x & 15);
    return ptab[p < npos ? p : 0];
}
template <int IN0>
__device__ __forceinline__ Tile tile_from(int tile, unsigned hw, int npos, const LaneGeom &lg, int vi);
template <int IN0>
__device__ __forceinline__ Tile tile_of(int tile, const ptab_t *ptab, int npos, const LaneGeom &lg, int vi) {
    return tile_from<IN0>(tile, tile_code(tile, ptab, npos), npos, lg, vi);
}
template <int IN0>
__device__ __forceinline__ Tile tile_from(int tile, unsigned hw, int npos, const LaneGeom &lg, int vi) {
    const int n = threadIdx.x & 15, kg = (threadIdx.x & 63) >> 4;
    Tile t;
    const int p = tile * 16 + n;
    t.ok = p < npos;
    t.pos = t.ok ? p : 0;
    t.h = (int)(hw >> 8);
    t.w = (int)(hw & 0xffu);
    const int SW = sw(vi);
    const unsigned b0 = (unsigned)(((IN0 - 1 + t.h) * SW + t.w) * kPosBytes) + lg.c1;
    t.r1[0] = b0;
    t.r1[1] = b0 + lg.RB;
    // kernel row 2 of the last interior row of the high image: row slot 8 = slot 0 of the ring
    t.r1[2] = b0 + 2 * lg.RB - ((IN0 == 3 && t.h == kSlots - IN0 - 1) ? kSlots * lg.RB : 0);
    t.r1[3] = kg < 2 ? b0 + 64 : t.r1[1] - 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) t.r2[i] = t.r1[i] + lg.d2;
    return t;
}

// B operand registers of one tile: two read sets per group, each 16 bytes fetched as two ds_read_b64 (the records are
// 8-byte aligned).  The reads are inline asm: left to the compiler, neighbouring reads are fused into ds_read2_b64, which
// moves half the bytes per clock (MI355X_MICROARCH.md, LDS table), and volatile C++ reads are waited on one by one.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
struct BRegs {
    u32x2 s1[kGroups][2], s2[kGroups][2];
};

template <int IMM>
__device__ __forceinline__ void read16(unsigned addr, u32x2 &lo, u32x2 &hi) {
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(IMM), "n"(IMM + 8)
                 : "memory");
}

// issue the 32 reads of a tile (LDS returns in order: wait_groups<N>() below makes the first N groups usable)
__device__ __forceinline__ void load_b(unsigned lds_base, const Tile &t, BRegs &b) {
#define STG_CV_G(g, bi, imm)                                       \
    read16<imm>(lds_base + t.r1[bi], b.s1[g][0], b.s1[g][1]);     \
    read16<imm>(lds_base + t.r2[bi], b.s2[g][0], b.s2[g][1]);
    STG_CV_G(0, 0, 0)
    STG_CV_G(1, 0, 32)
    STG_CV_G(2, 3, 0)
    STG_CV_G(3, 1, 16)
    STG_CV_G(4, 1, 48)
    STG_CV_G(5, 2, 0)
    STG_CV_G(6, 2, 32)
    STG_CV_G(7, 2, 64)
#undef STG_CV_G
}
// the record quad (channels 4q..4q+3, three pieces) of one position: the residual input of the tile's epilogue, fetched
// with the tile's operand reads
struct Quad {
    u32x2 h, m, l;
};
__device__ __forceinline__ void read_quad(unsigned addr, int PL, Quad &q) {
    asm volatile("ds_read_b64 %0, %1" : "=v"(q.h) : "v"(addr) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(q.m) : "v"(addr + PL) : "memory");
    asm volatile("ds_read_b64 %0, %1" : "=v"(q.l) : "v"(addr + l_off(PL)) : "memory");
}
// s_waitcnt lgkmcnt(CNT) tied to groups [G0, G0 + 4) (and to the residual quad): their registers are only read after it
template <int CNT, int G0>
__device__ __forceinline__ void wait_groups(BRegs &b, Quad &q) {
    asm volatile("s_waitcnt lgkmcnt(%19)"
                 : "+v"(b.s1[G0][0]), "+v"(b.s1[G0][1]), "+v"(b.s2[G0][0]), "+v"(b.s2[G0][1]), "+v"(b.s1[G0 + 1][0]),
                   "+v"(b.s1[G0 + 1][1]), "+v"(b.s2[G0 + 1][0]), "+v"(b.s2[G0 + 1][1]), "+v"(b.s1[G0 + 2][0]),
                   "+v"(b.s1[G0 + 2][1]), "+v"(b.s2[G0 + 2][0]), "+v"(b.s2[G0 + 2][1]), "+v"(b.s1[G0 + 3][0]),
                   "+v"(b.s1[G0 + 3][1]), "+v"(b.s2[G0 + 3][0]), "+v"(b.s2[G0 + 3][1]), "+v"(q.h), "+v"(q.m), "+v"(q.l)
                 : "n"(CNT)
                 : "memory");
}

__device__ __forceinline__ f32x4 mma(const u32x4 &a, const u32x2 &b0, const u32x2 &b1, const f32x4 &c) {
    const u32x4 b = {b0.x, b0.y, b1.x, b1.y};
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// 12 MFMAs of groups [G0, G0 + 4), dealt round-robin to NC accumulator chains (a dependent 4-pass MFMA issues back to
// back at full rate -- tools/micro/mfma_chain.hip -- so NC only trades registers against the final adds)
template <int G0, int NC>
__device__ __forceinline__ void mma_groups(const u32x4 (&w)[kWpVecs], const BRegs &b, f32x4 (&acc)[NC]) {
#pragma unroll
    for (int g = G0; g < G0 + 4; ++g) {
        acc[(3 * g + 0) % NC] = mma(w[3 * g + 0], b.s1[g][0], b.s1[g][1], acc[(3 * g + 0) % NC]);
        acc[(3 * g + 1) % NC] = mma(w[3 * g + 1], b.s1[g][0], b.s1[g][1], acc[(3 * g + 1) % NC]);
        acc[(3 * g + 2) % NC] = mma(w[3 * g + 2], b.s2[g][0], b.s2[g][1], acc[(3 * g + 2) % NC]);
    }
}

// ---- half-tile form (32 operand registers instead of 64): groups 4H .. 4H+3 at a time ------------------------------
struct BHalf {
    u32x2 s1[4][2], s2[4][2];
};
// The sixteen reads of a half-tile AND their s_waitcnt are ONE asm statement: the compiler never sees a register whose
// data is still in flight, so no register-allocation decision (a copy between the read and a separate wait statement)
// can pick up stale data.  The wave stalls at the wait either way; the SIMD's other wave covers the LDS latency.
template <int H>
__device__ __forceinline__ void load_b_half(unsigned lds_base, const Tile &t, BHalf &b) {
    // groups 4H .. 4H+3: (base index, immediate) = (0,0) (0,32) (3,0) (1,16) | (1,48) (2,0) (2,32) (2,64)
    const unsigned a0 = lds_base + t.r1[H == 0 ? 0 : 1], a1 = lds_base + t.r1[H == 0 ? 3 : 2], a2 = lds_base + t.r1[H == 0 ? 1 : 2];
    const unsigned c0 = lds_base + t.r2[H == 0 ? 0 : 1], c1 = lds_base + t.r2[H == 0 ? 3 : 2], c2 = lds_base + t.r2[H == 0 ? 1 : 2];
    // H == 0: g0 = a0+0, g1 = a0+32, g2 = a1+0, g3 = a2+16;   H == 1: g0 = a0+48, g1 = a1+0 (kernel row 2), g2 = a1+32, g3 = a1+64
    if (H == 0) {
        asm volatile(
            "ds_read_b64 %0, %16\n\tds_read_b64 %1, %16 offset:8\n\tds_read_b64 %2, %19\n\tds_read_b64 %3, %19 offset:8\n\t"
            "ds_read_b64 %4, %16 offset:32\n\tds_read_b64 %5, %16 offset:40\n\tds_read_b64 %6, %19 offset:32\n\tds_read_b64 %7, %19 offset:40\n\t"
            "ds_read_b64 %8, %17\n\tds_read_b64 %9, %17 offset:8\n\tds_read_b64 %10, %20\n\tds_read_b64 %11, %20 offset:8\n\t"
            "ds_read_b64 %12, %18 offset:16\n\tds_read_b64 %13, %18 offset:24\n\tds_read_b64 %14, %21 offset:16\n\tds_read_b64 %15, %21 offset:24\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(b.s1[0][0]), "=&v"(b.s1[0][1]), "=&v"(b.s2[0][0]), "=&v"(b.s2[0][1]), "=&v"(b.s1[1][0]), "=&v"(b.s1[1][1]),
              "=&v"(b.s2[1][0]), "=&v"(b.s2[1][1]), "=&v"(b.s1[2][0]), "=&v"(b.s1[2][1]), "=&v"(b.s2[2][0]), "=&v"(b.s2[2][1]),
              "=&v"(b.s1[3][0]), "=&v"(b.s1[3][1]), "=&v"(b.s2[3][0]), "=&v"(b.s2[3][1])
            : "v"(a0), "v"(a1), "v"(a2), "v"(c0), "v"(c1), "v"(c2)
            : "memory");
    } else {
        asm volatile(
            "ds_read_b64 %0, %16 offset:48\n\tds_read_b64 %1, %16 offset:56\n\tds_read_b64 %2, %19 offset:48\n\tds_read_b64 %3, %19 offset:56\n\t"
            "ds_read_b64 %4, %17\n\tds_read_b64 %5, %17 offset:8\n\tds_read_b64 %6, %20\n\tds_read_b64 %7, %20 offset:8\n\t"
            "ds_read_b64 %8, %17 offset:32\n\tds_read_b64 %9, %17 offset:40\n\tds_read_b64 %10, %20 offset:32\n\tds_read_b64 %11, %20 offset:40\n\t"
            "ds_read_b64 %12, %18 offset:64\n\tds_read_b64 %13, %18 offset:72\n\tds_read_b64 %14, %21 offset:64\n\tds_read_b64 %15, %21 offset:72\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(b.s1[0][0]), "=&v"(b.s1[0][1]), "=&v"(b.s2[0][0]), "=&v"(b.s2[0][1]), "=&v"(b.s1[1][0]), "=&v"(b.s1[1][1]),
              "=&v"(b.s2[1][0]), "=&v"(b.s2[1][1]), "=&v"(b.s1[2][0]), "=&v"(b.s1[2][1]), "=&v"(b.s2[2][0]), "=&v"(b.s2[2][1]),
              "=&v"(b.s1[3][0]), "=&v"(b.s1[3][1]), "=&v"(b.s2[3][0]), "=&v"(b.s2[3][1])
            : "v"(a0), "v"(a1), "v"(a2), "v"(c0), "v"(c1), "v"(c2)
            : "memory");
    }
}
template <int H>
__device__ __forceinline__ void mma_half(const u32x4 (&w)[kWpVecs], const BHalf &b, f32x4 &acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = 4 * H + i;
        acc = mma(w[3 * g + 0], b.s1[i][0], b.s1[i][1], acc);
        acc = mma(w[3 * g + 1], b.s1[i][0], b.s1[i][1], acc);
        acc = mma(w[3 * g + 2], b.s2[i][0], b.s2[i][1], acc);
    }
}

// Tile loop over the 16-position tiles of a scene; epi(tile geometry, accumulator, residual quad) finishes one tile.
// Software-pipelined by one tile: the operand reads of tile i are issued, THEN the epilogue of tile i-1 runs (its VALU
// work covers the LDS latency), then tile i's MFMAs.  The in-place forward tolerates the late writes: a tile's output
// rows are rows no later tile reads.  RES: fetch the tile's own records of the INPUT image (slot offset IN0) for the
// epilogue's residual.  REV walks the tiles from the last position down (the in-place forward's low -> high layers).
template <int IN0, bool REV, bool RES, int DBG = 0, typename Epi>
__device__ __forceinline__ void conv_tiles(const u32x4 (&w)[kWpVecs], const f32x4 binit, const unsigned char *lds,
                                           const ptab_t *ptab, int npos, int vi, const LaneGeom &lg, Epi epi,
                                           long long *stamps = nullptr) {
#define STG_CV_STAMP(k) do { if ((DBG & 32) && stamps && threadIdx.x == 0) stamps[i * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
    const int ntiles = (npos + 15) >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;      // (LDS addresses are 32-bit: the low half of the generic one)
    const int kq = (threadIdx.x & 63) >> 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    Tile t = tile_of<IN0>(REV ? ntiles - 1 : 0, ptab, npos, lg, vi), tp = t;
    f32x4 cp = zero;
    Quad qp = {};
    const bool late = (DBG & 16) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256;
    if ((DBG & 16) && late) __builtin_amdgcn_s_barrier();
    for (int i = 0; i < ntiles; ++i) {
        BRegs b;
        Quad q = {};
        STG_CV_STAMP(0);
        if (DBG & 2) {
#pragma unroll
            for (int g = 0; g < kGroups; ++g) b.s1[g][0] = b.s1[g][1] = b.s2[g][0] = b.s2[g][1] = u32x2{(unsigned)t.r1[g & 3], (unsigned)t.r2[g & 3]};
        } else {
            load_b(lds_base, t, b);
            if (RES) read_quad(lds_base + pos_off(vi, IN0 + t.h, t.w) + 8 * (kq < 3 ? kq : 0), lg.PL, q);
        }
        STG_CV_STAMP(1);
        if (i > 0) epi(tp, cp, qp);
        STG_CV_STAMP(2);
        Tile tn = t;
        if (i + 1 < ntiles) tn = tile_of<IN0>(REV ? ntiles - 2 - i : i + 1, ptab, npos, lg, vi);
        f32x4 acc[4] = {binit, zero, zero, zero};
        STG_CV_STAMP(3);
        if (DBG & 16) __builtin_amdgcn_s_barrier();
        wait_groups<15, 0>(b, q);       // reads issued in order, <= 15 outstanding: groups 0..3 have landed
        STG_CV_STAMP(4);
        if (!(DBG & 4)) mma_groups<0, 4>(w, b, acc);
        STG_CV_STAMP(5);
        wait_groups<0, 4>(b, q);
        if (!(DBG & 4)) mma_groups<4, 4>(w, b, acc);
        if (DBG & 4) acc[1] = f32x4{__uint_as_float(b.s1[0][0].x ^ b.s2[7][1].y), __uint_as_float(b.s1[3][0].x), 0.f, __uint_as_float(w[5].x ^ w[23].y)};
        cp = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        STG_CV_STAMP(6);
        if (DBG & 16) __builtin_amdgcn_s_barrier();
        tp = t;
        qp = q;
        t = tn;
    }
    if (ntiles > 0) epi(tp, cp, qp);
    if ((DBG & 16) && !late) __builtin_amdgcn_s_barrier();
#undef STG_CV_STAMP
}
__device__ __forceinline__ f32x4 unsplit4(const Quad &q) {
    return unsplit4(make_uint2(q.h.x, q.h.y), make_uint2(q.m.x, q.m.y), make_uint2(q.l.x, q.l.y));
}

// write the four channels 4q..4q+3 of a position record (all three pieces)
__device__ __forceinline__ void put4(unsigned char *lds, unsigned off, int PL, const f32x4 &v) {
    uint2 ph, pm, pl;
    split_pack4(v, ph, pm, pl);
    *reinterpret_cast<uint2 *>(lds + off) = ph;
    *reinterpret_cast<uint2 *>(lds + off + PL) = pm;
    *reinterpret_cast<uint2 *>(lds + off + l_off(PL)) = pl;
}
__device__ __forceinline__ f32x4 get4(const unsigned char *lds, unsigned off, int PL) {
    const uint2 ph = *reinterpret_cast<const uint2 *>(lds + off), pm = *reinterpret_cast<const uint2 *>(lds + off + PL),
                pl = *reinterpret_cast<const uint2 *>(lds + off + l_off(PL));
    return unsplit4(ph, pm, pl);
}
#endif  // __HIPCC__

}  // namespace cv
}  // namespace stg
