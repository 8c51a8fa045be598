// txp_wgrad (K2): the TXP-CNN weight / bias gradients -- argument block, launch geometry, launcher.
#pragma once
#include "model_common.hpp"

namespace stg {

// K2 stages at most kWgradChunkV pedestrian columns of a scene at a time: a larger scene is cut into
// ceil(V_n / kWgradChunkV) equal column chunks (each with its two halo columns of the plane), every chunk one work item
// whose LDS image -- and K loop -- is that of a small scene.  The weight gradient is a sum over positions, so chunks add.
constexpr int kWgradChunkV = 32;
__host__ __device__ inline int wgrad_chunks(int V) { return (V + kWgradChunkV - 1) / kWgradChunkV; }
__host__ __device__ inline int wgrad_image_v(int V) { return V < kWgradChunkV ? V : kWgradChunkV; }
// LDS floats of one staged work item: the zero-bordered position-major plane a_l, then dz_l (bf16 storage: half the
// bytes, rows padded to an even number of positions -- save_sw / save_vw)
__host__ __device__ inline int wgrad_plane_floats(int V0, bool bf16) {
    const int V = wgrad_image_v(V0);
    if (!bf16) return plane_slot(V);
    return (((Cfg::C + 2) * save_sw(V, true) * Cfg::P) / 2 + 3) & ~3;
}
__host__ __device__ inline int wgrad_image_floats(int V0, bool bf16 = false) {
    const int V = wgrad_image_v(V0);
    if (!bf16) return plane_slot(V) + dz_slot(V);
    return wgrad_plane_floats(V0, true) + (((Cfg::C * save_vw(V, true) * Cfg::P) / 2 + 3 + 4) & ~3);
}
// slab geometry: layer 0 has c_in = T, layers 1..L (L = output conv) have c_in = P; one row = [P][c_in][9] + [P]
__host__ __device__ inline int wgrad_row_len(int layer) {
    return Cfg::P * (layer == 0 ? Cfg::T : Cfg::P) * 9 + Cfg::P;
}
__host__ __device__ inline int64_t wgrad_slab_base(int layer, int rows) {
    return layer == 0 ? 0 : (int64_t)rows * (wgrad_row_len(0) + (int64_t)(layer - 1) * wgrad_row_len(1));
}

struct WgradGeom {
    int waves, nbuf, grid, rows;    // rows = slab rows per layer (= the largest workgroup count of a layer)
    bool bf16mma;                   // txp_wgrad_bf16.hip (bf16 matrix pipe) instead of txp_wgrad.hip
    int wg_begin[kMaxTxp + 2];      // layer l owns workgroups [wg_begin[l], wg_begin[l+1])
    size_t lds;
};
bool wgrad_geom(const ModelLayout &L, int N, int V, WgradGeom *g);

struct WgradArgs {
    ModelLayout lay;
    const int32_t *num_peds;
    const int32_t *order;  // non-null: scenes sorted by crowd size (descending)
    const int32_t *order_peds;   // with `order`: order_peds[i] = pedestrians of scene order[i] (clamped to [0, V])
    const int32_t *key_start;    // with `order`: key_start[k] = scenes with more than V - k pedestrians (scene_order_kernel), or null
    int serpentine;        // walk the sorted list boustrophedon (1) or with a plain stride (0)
    int N, V;
    const float *ws;       // saved planes a_0 .. a_L (position-major, interior rows with their border columns)
    const float *dzg;      // [N][L+1][dz_slot(V)]  dz_l of every layer (l = L: the output conv's, i.e. dy), position-major
    int64_t ws_stride;
    float *slab2;          // [layer 0..L][rows][row_len(layer)] packed, see wgrad_slab_base()
    int rows;
    int wg_begin[kMaxTxp + 2];
    int debug_skip;        // diagnostic builds only: 64 skip staging, 128 skip the MFMA loop
};
int launch_txp_wgrad(const WgradArgs &w, const WgradGeom &g, hipStream_t st);
// txp_wgrad_bf16.hip: the same GEMM on the bf16 matrix pipe with exact three-piece operands
bool wgrad_bf16_fits(const ModelLayout &L, int V);
void wgrad_bf16_geom(WgradGeom *g, const ModelLayout &L, int V);
int launch_txp_wgrad_bf16(const WgradArgs &w, const WgradGeom &g, hipStream_t st);

}  // namespace stg
