// gather: the collation step of the reference's DataLoader over TrajectoryDataset (utils.py:121-193 builds the
// per-window tensors, the loader picks windows) as a device op -- the whole windowed dataset stays resident in HBM
// (ragged concatenation of the windows' relative trajectories, 160 bytes per pedestrian-window: eth/train is 4.8 MB)
// and a batch is gathered BY INDEX, zero-padded to V pedestrian slots, in the layouts the kernels behind it consume:
//     obs_rel (N, V, 2, T_obs)   -> stg_adj_build (utils.seq_to_graph)
//     target  (N, T_pred, V, 2)  -> stg_nll_fwd   (V_tr of train.py:48-56)
//     num_peds (N)
// The index array lives on the device (a shuffled epoch order, refreshed in place), so a captured training step
// gather -> adj_build -> forward -> loss -> backward -> update replays with no host->device traffic at all.
// One workgroup per scene-window; pure data movement (HBM-bound, 160 V bytes read, 160 V written per window).
#include "common.hpp"

namespace stg {

__global__ __launch_bounds__(256) void gather_windows_kernel(const float *__restrict__ rel_all,
                                                             const int32_t *__restrict__ win_start,
                                                             const int32_t *__restrict__ index, int n_windows, int V,
                                                             int T_obs, int T_pred, float *__restrict__ obs_rel,
                                                             float *__restrict__ target, int32_t *__restrict__ num_peds) {
    const int n = blockIdx.x, tid = threadIdx.x, T_all = T_obs + T_pred;
    int w = index ? index[n] : n;
    w = w < 0 ? 0 : (w >= n_windows ? n_windows - 1 : w);
    const int s = win_start[w];
    int c = win_start[w + 1] - s;
    c = c < 0 ? 0 : (c > V ? V : c);
    if (tid == 0) num_peds[n] = c;
    const float *src = rel_all + (int64_t)s * 2 * T_all;            // (c, 2, T_all)
    float *o = obs_rel + (int64_t)n * V * 2 * T_obs;                // (V, 2, T_obs)
    for (int e = tid; e < V * 2 * T_obs; e += blockDim.x) {
        const int t = e % T_obs, vc = e / T_obs, ch = vc & 1, v = vc >> 1;
        o[e] = v < c ? src[((int64_t)v * 2 + ch) * T_all + t] : 0.f;
    }
    float *g = target + (int64_t)n * T_pred * V * 2;                // (T_pred, V, 2)
    for (int e = tid; e < T_pred * V * 2; e += blockDim.x) {
        const int ch = e & 1, v = (e >> 1) % V, t = (e >> 1) / V;
        g[e] = v < c ? src[((int64_t)v * 2 + ch) * T_all + T_obs + t] : 0.f;
    }
}

}  // namespace stg

extern "C" int stg_gather_windows(const float *seq_rel_all, const int32_t *win_start, const int32_t *index,
                                  int n_windows, int N, int V, int T_obs, int T_pred, float *obs_rel, float *target,
                                  int32_t *num_peds, void *stream) {
    STG_REQUIRE(N >= 0 && V > 0 && T_obs > 0 && T_pred >= 0 && n_windows > 0, STG_EINVAL,
                "stg_gather_windows: bad sizes N=%d V=%d T=%d+%d windows=%d", N, V, T_obs, T_pred, n_windows);
    if (N == 0) return STG_OK;
    STG_REQUIRE(seq_rel_all && win_start && obs_rel && target && num_peds, STG_EINVAL, "stg_gather_windows: null pointer");
    hipLaunchKernelGGL(stg::gather_windows_kernel, dim3(N), dim3(256), 0, stg::as_stream(stream), seq_rel_all, win_start,
                       index, n_windows, V, T_obs, T_pred, obs_rel, target, num_peds);
    STG_LAUNCH_CHECK("stg_gather_windows");
    return STG_OK;
}
