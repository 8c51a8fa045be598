// Wave-per-scene TXP-CNN kernels (txp_wave.hip): argument blocks and launchers.
#pragma once
#include "model_common.hpp"

namespace stg {

// Mixed-V launch (ragged batches padded beyond V = 32, sorted scene list available): ONE launch of 4-wave
// workgroups whose LDS is sized for four scenes of up to 32 pedestrians.  The workgroups split themselves (on the
// device, from the tier offsets of the sorted list -- no host sync) into three classes: small scenes run four to a
// workgroup, scenes up to `v_mid` two to a workgroup (two waves idle), larger ones one to a workgroup; the class
// sizes follow the summed crowd sizes, large classes take the lowest block indices (dispatched first).
struct MixGeom {
    int on;               // 0: uniform launch (Vl / tier as given)
    int v_small, v_mid;   // class bounds: V_n <= v_small | <= v_mid | <= V
    int block_floats;     // LDS floats of one workgroup = 4 * per-wave floats at v_small
};

// Team launch of the exact-bf16 kernels (batches padded beyond 32 pedestrians; small batches): ONE launch of 4-wave
// workgroups in which a scene-window is worked on by one, two or four waves (scene_team.hpp) according to its crowd:
// V_n <= v1 one wave (four scenes per workgroup round), V_n <= v2 two waves (two scenes per round), larger ones four.
// The workgroups read the class sizes from the tier offsets of the sorted scene list on the device (no host sync).
constexpr int kTeamMaxV = 128;       // four chunks of 32 columns
struct TeamGeom {
    int on;
    int v1, v2;           // class bounds (v1 <= 32, v2 <= 64)
    int region_floats;    // LDS floats of a workgroup's image region: one four-wave scene, two two-wave scenes, four solo scenes
};

// forward: the WHOLE model per scene -- st_gcn block (from the aggregated input stgcn_agg_kernel left) + TXP-CNN
struct TxpFwdArgs {
    ModelLayout lay;
    const float *params, *buffers;
    const int32_t *num_peds;
    SceneTier tier;        // which scenes this launch serves (ragged batches: sorted, walked boustrophedon)
    int Vl;                // LDS geometry of the launch: >= every V_n of the tier (<= V)
    MixGeom mix;
    TeamGeom team;
    int N, V;
    const float *x;        // (N, c_in, T, V) strided block input (residual branch)
    int64_t x_sn, x_sc, x_st, x_sv;
    const float *adj;      // (unused by the wave kernels: A was consumed by stgcn_agg_kernel)
    int64_t a_sn;
    const float *agg;      // per scene [agg_stride]: ax at agg_ax ([c_in][T][V_n]), cs at agg_cs ([T][V_n])
    int64_t agg_stride, agg_ax, agg_cs;
    float *y;              // (N, C, P, V)
    const unsigned *wpf;   // prepared forward A operands (txp_conv_bf16.hpp), [L+1][cv::kWpDwords], or null
    float *ws;             // per-scene workspace or null (inference)
    int64_t ws_stride;
    float *stats;          // (N, stat_floats) per-scene BatchNorm statistics (bn_mode 1) or null
    unsigned long long *stamps;   // diagnostic build only (STG_STAMPS=1): [N][16] s_memtime stamps, else null
    int debug_skip;        // diagnostic builds only
    int stagger;           // start delay of the second half of every workgroup's waves (stagger_start units)
};

// backward: TXP-CNN input-gradient chain + the st_gcn block backward per scene (everything but the TXP weight gradients)
struct TxpBwdArgs {
    ModelLayout lay;
    const float *params;
    const int32_t *num_peds;
    SceneTier tier;        // which scenes this launch serves (ragged batches: sorted, walked boustrophedon)
    int Vl;                // LDS geometry of the launch: >= every V_n of the tier (<= V)
    MixGeom mix;
    TeamGeom team;
    int N, V;
    const float *x;        // (N, c_in, T, V) strided block input (residual branch)
    int64_t x_sn, x_sc, x_st, x_sv;
    const float *adj;      // (unused: no dx on this path, A is not needed)
    int64_t a_sn;
    const float *dy;       // (N, C, P, V): dV_pred -- or, with nll_target, V_pred itself
    // fused loss (stg_model_bwd_nll): the input stage computes d(sum_n w_n loss_n)/dV_pred from V_pred and the
    // target instead of reading it, and writes the per-scene losses
    const float *nll_target;   // (N, P, V, 2) or null
    const float *nll_weights;  // (N) or null (all ones)
    float *nll_losses;         // (N)
    const float *ws;
    int64_t ws_stride;
    float *dzg;            // [N][L][dz_slot(V)]   dz_l of the hidden layers for the weight-gradient GEMM
    const unsigned *wp;    // prepared input-gradient A operands (txp_conv_bf16.hpp), [L+1][cv::kWpDwords] -- the batch
                           // tail of the workspace, written by the forward's aggregation launch -- or null
    float *rows;           // [N][n_blk_params + n_txp]  per-scene small-parameter gradients: st_gcn block, PReLU slopes
    int debug_skip;        // timing-only diagnostic (STG_DEBUG_SKIP): 512 dz build, 1024 dgrad tile loops -- wrong results
    int split_bf16;        // 1: the input-gradient GEMMs run on bf16 MFMAs with hi/lo-split operands (see txp_wave.hip)
    int stagger;           // start delay of the second half of every workgroup's waves (stagger_start units)
};

// true when the wave-per-scene path serves this model / V (else the workgroup-per-scene kernels run)
bool txp_wave_fits(const ModelLayout &L, int V);
// Path of a batch of N scenes: the wave-per-scene kernels when they fit AND the batch fills the chip with one wave
// per scene; a small batch (fewer scenes than resident wave slots: every scene's latency chain is the step) runs the
// workgroup-per-scene kernels with `*wg_waves` waves per scene instead (measured: N = 512 x 4 waves 2.9 vs 2.5 M
// scene-windows/s, N = 128 x 8 waves 0.95 vs 0.70).  Forward and backward make the same choice from (L, N, V).
bool use_wave_path(const ModelLayout &L, int N, int V, int *wg_waves);
int64_t ws_tail_wp_floats(const ModelLayout &l, int V);      // model_fwd.hip: floats of the operand part of the workspace's batch tail
int launch_txp_fwd_wave(const TxpFwdArgs &a, hipStream_t st);
int launch_txp_bwd_wave(const TxpBwdArgs &a, hipStream_t st);
// txp_x6.hip: the exact-bf16 kernels (one wave per scene, or the team launch) -- what the two launchers above run whenever
// txp_fwd_x6_fits / txp_bwd_x6_fits hold and the prepared operands are there
int launch_txp_fwd_x6(const TxpFwdArgs &a, hipStream_t st);
int launch_txp_bwd_x6(const TxpBwdArgs &a, hipStream_t st);
// the exact-bf16 kernels (txp_fwd_x6 / txp_bwd_x6 and their team forms): V <= kTeamMaxV
bool txp_bwd_x6_fits(const ModelLayout &L, int V);
bool txp_fwd_x6_fits(const ModelLayout &L, int V);
int64_t txp_bwd_x6_wp_floats(const ModelLayout &L);

}  // namespace stg
