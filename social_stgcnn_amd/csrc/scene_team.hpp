// scene_team: who works on a scene-window inside the bf16-pipe scene kernels (txp_wave.hip, stgcn_block.hpp column mode).
//
// A scene of up to 32 pedestrians belongs to ONE wave (SoloScene: every member is a compile-time constant, the code is
// what it was before teams existed).  A larger scene -- or, in a small batch, any scene whose latency chain would be the
// step -- is cut into `nch` (2 or 4) equal COLUMN CHUNKS of at most 32 pedestrians, one wave each (TeamScene): the waves
// share the scene's LDS image (a chunk's halo columns are simply its neighbours' columns of the same image), keep their
// own tiles' outputs in registers like the solo wave does, and meet at workgroup barriers where the solo wave has wave
// barriers.  Per-scene sums the computation goes on with (BatchNorm statistics) are wave sums exchanged through a small
// LDS area (`sum`: one barrier each); sums that only leave the kernel (parameter gradients, the loss) are parked per wave in
// an LDS row (`reduce` + `writer` + `row`) and added by the team's leading wave at the end of the scene, in chunk order:
// deterministic, no atomics.
//
// Every wave of a workgroup executes the same barrier sequence (the scene code's control flow depends on the model,
// never on the crowd size), so two 2-wave teams may share a workgroup.
#pragma once
#include "common.hpp"

namespace stg {

// LDS operations of this wave complete, then the workgroup meets.  Global stores are NOT drained (hipcc's
// __syncthreads() is the same on gfx950 unless an LDS-DMA is in flight; spelled out so that no fence is ever added).
__device__ __forceinline__ void team_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct SoloScene {
    static constexpr bool kTeam = false;
    int vi;                                            // pedestrians of the scene
    __device__ __forceinline__ int w0() const { return 0; }
    __device__ __forceinline__ int wc() const { return vi; }
    __device__ __forceinline__ int nch() const { return 1; }
    __device__ __forceinline__ int ci() const { return 0; }
    __device__ __forceinline__ bool lead() const { return true; }
    __device__ __forceinline__ void sync() const { __builtin_amdgcn_wave_barrier(); }
    // K per-lane values summed over the scene, the totals in every lane
    template <int K>
    __device__ __forceinline__ void sum(float (&v)[K], int) const { wave_sum_n<K>(v); }
    // K per-lane values summed over this wave's part of the scene; the totals are valid in the lanes where writer() holds,
    // which store them to row(the scene's row of small-parameter gradients).  Lane 63 stores them where the DPP stages leave
    // them: broadcast to every lane (v_readlane) they became SCALARS -- 142 readlanes, a v_mov per stored value and enough
    // SGPR pressure in the backward's block tail for 516 spill reloads (v_readlane from a spill VGPR) in the ISA.
    template <int K>
    __device__ __forceinline__ void reduce(float (&v)[K]) const { wave_sum_to_last<K>(v); }
    __device__ __forceinline__ bool writer() const { return (threadIdx.x & 63) == 63; }
    __device__ __forceinline__ float *row(float *scene_row) const { return scene_row; }
};

// exchange-area slots (floats): forward st_gcn block | backward (two broadcast sums, then the four parked rows)
constexpr int kXrF_s1 = 0, kXrF_s2 = 20, kXrF_sm2 = 40, kXrF_sv2 = 80, kXrFwd = 120;
constexpr int kXrB_s1 = 0, kXrB_s3 = 64, kXrRows = 128;
constexpr int kTeamRow = 160;                          // floats of a wave's parked row (>= block parameters + PReLU slopes + loss)
constexpr int kTeamRowLoss = kTeamRow - 1;             // where a wave parks its part of the scene's loss sum

struct TeamScene {
    static constexpr bool kTeam = true;
    int vi;                                            // pedestrians of the scene (row strides of every array)
    int w0_, wc_;                                      // first column and width of this wave's chunk
    int nch_, ci_;                                     // waves on the scene, this wave's place among them
    float *xr;                                         // LDS exchange area of the scene's waves (kXrFwd / kXrRows floats)
    float *rows;                                       // LDS: the team's parked rows (backward), kTeamRow floats per wave
    bool bar;                                          // nch_ > 1 (diagnostic builds can switch the barriers off for timing)
    __device__ __forceinline__ int w0() const { return w0_; }
    __device__ __forceinline__ int wc() const { return wc_; }
    __device__ __forceinline__ int nch() const { return nch_; }
    __device__ __forceinline__ int ci() const { return ci_; }
    __device__ __forceinline__ bool lead() const { return ci_ == 0; }
    __device__ __forceinline__ void sync() const {
        if (bar) team_barrier();
        else __builtin_amdgcn_wave_barrier();
    }
    // `slot`: offset of this call site's 4 * K floats in the exchange area (a slot is reused only behind a later barrier of
    // the team).  Every wave's lane 63 parks its totals (a two-wave team also zeroes the two unused places), one barrier,
    // then everybody reads all four places: no per-element branches, all 4 K reads in flight at once.
    template <int K>
    __device__ __forceinline__ void sum(float (&v)[K], int slot) const {
        if (nch_ == 1) {
            wave_sum_n<K>(v);
            return;
        }
        wave_sum_to_last<K>(v);
        float *x = xr + slot;
        if ((threadIdx.x & 63) == 63) {
#pragma unroll
            for (int k = 0; k < K; ++k) x[ci_ * K + k] = v[k];
            if (nch_ == 2) {
#pragma unroll
                for (int k = 0; k < K; ++k) x[(ci_ + 2) * K + k] = 0.f;
            }
        }
        if (bar) team_barrier();
        float p[4][K];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int k = 0; k < K; ++k) p[c][k] = x[c * K + k];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = (p[0][k] + p[1][k]) + (p[2][k] + p[3][k]);
    }
    template <int K>
    __device__ __forceinline__ void reduce(float (&v)[K]) const { wave_sum_to_last<K>(v); }
    __device__ __forceinline__ bool writer() const { return (threadIdx.x & 63) == 63; }
    // this wave's parked row (LDS); `c`: the row of wave c of the team
    __device__ __forceinline__ float *row(float *) const { return rows + ci_ * kTeamRow; }
    __device__ __forceinline__ const float *row_of(int c) const { return rows + c * kTeamRow; }
};

// backward exchange area of a WORKGROUP: two teams' broadcast sums, then one parked row per wave
constexpr int kXrBwdWg = 2 * kXrRows + 4 * kTeamRow;

}  // namespace stg
