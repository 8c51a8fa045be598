// Shared by the wave-per-scene kernel families: the fp32-MFMA generation (txp_wave.hip) and the exact-bf16 kernels with
// their team forms (txp_x6.hip) -- position tables, the workgroup's LDS copy of the st_gcn parameters, persistent-grid sizes.
#pragma once
#include "model_common.hpp"
#include "stgcn_block.hpp"
#include "txp_wave.hpp"

namespace stg {

// LDS floats of one wave's position table (16-bit entries, Cfg::T * v of them) [+ the st_gcn tail's 32 reduction totals]
__host__ __device__ inline int ptab_floats(int v) { return ((Cfg::T * v + 1) / 2 + 3) & ~3; }
__host__ __device__ inline int bwd_ptab_floats(int v) { return ptab_floats(v) + kRedMax; }
// LDS floats of the workgroup's copy of the st_gcn block parameters and BatchNorm running statistics: the block code
// of the wave kernels reads them with broadcast LDS reads (no SGPR pressure, no scalar-load waits inside its passes)
__host__ __device__ inline int wave_param_floats(const ModelLayout &L) {
    return ((L.n_blk_params + 3) & ~3) + ((L.n_buffers + 3) & ~3);
}

// position -> (h << 8 | w) table of the scene, T * vi entries of 16 bits: p < C * vi are the positions of the TXP
// plane, q < T * vi the (t, w) columns of the st_gcn block -- no integer divisions per tile / column
__device__ __forceinline__ void build_ptab(ptab_t *ptab, int vi) {
    const int lane = threadIdx.x & 63;
    for (int p = lane; p < Cfg::T * vi; p += 64) {
        const int h = p / vi;
        ptab[p] = (ptab_t)((h << 8) | (p - h * vi));
    }
}
// the same for a wave that owns the column chunk [w0, w0 + wc) of a scene: position p of the chunk is (row p / wc, column
// w0 + p % wc) of the scene
__device__ __forceinline__ void build_ptab(ptab_t *ptab, int w0, int wc) {
    const int lane = threadIdx.x & 63;
    if (wc <= 0 && lane == 0) ptab[0] = 0;            // (an empty chunk: entry 0 is still read, never used)
    for (int p = lane; p < Cfg::T * wc; p += 64) {
        const int h = p / wc;
        ptab[p] = (ptab_t)((h << 8) | (w0 + p - h * wc));
    }
}


// "Every vector-memory operation issued so far has completed", stated where the compiler can see it (an S_WAITCNT it
// models).  vmcnt counts loads AND stores in order; wherever a register MAY still be waiting for a load on some path of
// the control-flow graph (a tile loop whose iterations are guarded by runtime tile counts is enough), the compiler puts
// s_waitcnt vmcnt(0) in front of its use -- which also waits for the acknowledgement of every store issued since: one
// HBM round trip per tile.  Draining once, right after the loads and before the guarded code, leaves nothing pending,
// and the tiles' stores are fire-and-forget again.  (Found in the ISA, not in a counter: DESIGN 5.2.)
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }   // vmcnt(0), expcnt / lgkmcnt free

// Workgroup prologue of the wave kernels: the st_gcn block's parameters (and running statistics) into LDS, once.
__device__ __forceinline__ void stage_block_params(const ModelLayout &L, const float *__restrict__ params,
                                                   const float *__restrict__ buffers, float *blk_p, float *blk_b, int nt) {
    for (int e = threadIdx.x; e < L.n_blk_params; e += nt) blk_p[e] = params[e];
    if (buffers)
        for (int e = threadIdx.x; e < L.n_buffers; e += nt) blk_b[e] = buffers[e];
    __syncthreads();
}

// ---- persistent grids (host) ----------------------------------------------------------------------------
inline int wave_wpb(size_t per_wave) {
    // 4 waves per workgroup: the LDS footprint then admits either one workgroup (forward: 4 waves per CU,
    // one per SIMD) or two (backward: 8 per CU, two per SIMD) -- BALANCED over the four SIMDs.  Odd
    // residencies (6 waves per CU) measured 1.5x slower per wave (tools/micro/conv_tile_bench.hip).
    int wpb = diag_env("STG_TXP_WPB", 4);
    if (wpb != 1 && wpb != 2 && wpb != 8) wpb = 4;
    while (wpb > 1 && per_wave * wpb > (size_t)kLdsBytes) wpb >>= 1;
    return wpb;
}

// persistent grid: as many workgroups as the chip holds at once (LDS-limited, 2 waves per SIMD)
inline int wave_grid(size_t lds, int wpb, int N) {
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

}  // namespace stg
