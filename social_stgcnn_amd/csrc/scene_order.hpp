// scene_order: the ragged-batch schedule -- scene indices sorted by pedestrian count (descending, stable) with the tier
// offsets of the sorted list -- as device code of ONE workgroup of NW waves.  Runs as its own launch (scene_order_kernel,
// 16 waves: stg_scene_order and the entry points without an aggregation launch in front) or inside an extra workgroup of the
// forward's aggregation launch (stgcn_agg.hip, 4 waves: no launch of its own on the training path).
#pragma once
#include "model_common.hpp"

namespace stg {

// Stable counting sort of the scenes by pedestrian count, descending, in ONE workgroup of NW waves.  Wave w owns
// the contiguous index range [w*per_wave, (w+1)*per_wave) and the column w of the LDS histogram hist[key][wave]
// (key k = 0 is the LARGEST crowd).  Pass 1 counts (integer LDS adds).  A workgroup scan over (key-major,
// wave-minor) turns the counts into list offsets; pass 2 ranks the 64 scenes of a batch among their equal-key
// lanes (one distinct key peeled per iteration: readlane + ballot, no memory) and places scene i at offset + rank.
// The result is deterministic and stable.
// key_start[k] (k = 0..V+1, optional) = number of scenes with more than V-k pedestrians, i.e. the scenes with at
// most x pedestrians are order[key_start[V - x] .. N): the V-tiers of the entry points.
template <int NW>
__device__ __forceinline__ void scene_order_body(const int32_t *__restrict__ num_peds, int N, int V,
                                                 int32_t *__restrict__ order, int32_t *__restrict__ key_start,
                                                 int32_t *__restrict__ order_peds, int *hist /* [K][NW] */,
                                                 int *wave_tot /* [NW] */) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, K = V + 1;
    const int per_wave = (((N + NW - 1) / NW) + 63) & ~63;
    const int w_lo = wv * per_wave, w_hi = (w_lo + per_wave) < N ? (w_lo + per_wave) : N;
    const int total = K * NW;
    for (int e = tid; e < total; e += NW * 64) hist[e] = 0;
    __syncthreads();
    for (int b = w_lo; b < w_hi; b += 64) {
        const int i = b + lane;
        int k = -1;
        if (i < w_hi) {
            const int v = num_peds[i];
            k = V - (v < 0 ? 0 : (v > V ? V : v));
        }
        if (k >= 0) atomicAdd(&hist[k * NW + wv], 1);       // integer LDS add, no return: order-independent
    }
    __syncthreads();
    // exclusive scan over the K*16 counts: E consecutive entries per thread, then a scan of the thread sums
    const int E = (total + NW * 64 - 1) / (NW * 64);
    const int lo = tid * E, hi = (lo + E) < total ? (lo + E) : total;
    int c = 0;
    for (int e = lo; e < hi; ++e) c += hist[e];
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int run = incl - c;
    for (int w = 0; w < wv; ++w) run += wave_tot[w];
    for (int e = lo; e < hi; ++e) {
        const int t = hist[e];
        hist[e] = run;
        run += t;
    }
    __syncthreads();
    if (key_start) {
        for (int k = tid; k < K; k += NW * 64) key_start[k] = hist[k * NW];     // first list position of key V-k
        if (tid == 0) key_start[K] = N;
    }
    __syncthreads();
    for (int b = w_lo; b < w_hi; b += 64) {
        const int i = b + lane;
        int k = -1;
        if (i < w_hi) {
            const int v = num_peds[i];
            k = V - (v < 0 ? 0 : (v > V ? V : v));
        }
        // rank of every lane among the equal-key lanes below it: peel one distinct key per iteration (ALU only)
        unsigned long long todo = __ballot(k >= 0);
        const unsigned long long below = (1ull << lane) - 1ull;
        int rank = 0;
        bool last = false;
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            const int k0 = __builtin_amdgcn_readlane(k, src);
            const unsigned long long m = __ballot(k == k0);
            if (k == k0) {
                rank = __popcll(m & below);
                last = (m >> lane) == 1ull;              // highest lane holding this key
            }
            todo &= ~m;
        }
        if (k >= 0) {
            const int base = hist[k * NW + wv];
            order[base + rank] = i;
            if (order_peds) order_peds[base + rank] = V - k;      // the sorted (clamped) pedestrian counts themselves
            __builtin_amdgcn_wave_barrier();
            if (last) hist[k * NW + wv] = base + rank + 1;
        }
    }
}

}  // namespace stg
