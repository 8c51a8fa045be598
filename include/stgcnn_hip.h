/*
 * stgcnn_hip.h -- C ABI of libstgcnn_hip.so: the MI355X (gfx950) Social-STGCNN hot path.
 *
 * The reference (GRatTWCU/Social-STGCNN) has no FFI: its hot path is Python on torch
 * (model.py, utils.py, metrics.py).  This library is what a binding for that path binds
 * (ctypes stub: social_stgcnn_amd/_lib.py).  Each entry point cites the
 * reference code it replaces (file:line under /root/reference).
 *
 * Conventions
 *  - plain pointers (DEVICE memory unless stated) and sizes; fp32 data, int32 counts.
 *  - every function returns 0 on success, a negative STG_E* code for invalid arguments, or a
 *    positive hipError_t; it never aborts, allocates nothing, does not synchronise the host,
 *    reads no environment variable and keeps no mutable global state (re-entrant across
 *    streams; the only per-thread state is the text behind stg_last_error()).  Every choice a
 *    caller can make -- kernel path, waves per scene, storage type -- is an explicit argument
 *    (stg_model_desc.flags / .wg_waves).  Work is enqueued on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream).
 *  - a batch holds N scene-windows padded to V pedestrian slots; `num_peds` (int32[N], may be
 *    NULL = all V valid) gives the real count V_i of each scene.  Slots >= V_i are ignored on
 *    input and written as zeros on output.
 *  - tensors are contiguous in the layouts named below unless explicit strides (in elements)
 *    are part of the signature.
 */
#ifndef STGCNN_HIP_H
#define STGCNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STG_OK 0
#define STG_EINVAL (-1)      /* bad size / null pointer / inconsistent arguments          */
#define STG_EUNSUPPORTED (-2) /* configuration outside what the kernels are built for      */
#define STG_ELDS (-3)        /* scene too large for the 160 KiB LDS of one CU             */

#define STG_ABI_VERSION 7
#define STG_MAX_BLOCKS 4     /* st_gcn blocks in one fused model                          */

int stg_abi_version(void);
/* Human-readable description of the last error raised on the calling thread. */
const char *stg_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * R1/R2  utils.anorm + utils.seq_to_graph (utils.py:23-53) incl. networkx
 *        normalized_laplacian_matrix (call site utils.py:48-50).
 * rel:   relative displacements, element (n, v, c, t) at rel[n*rel_sn + v*rel_sv + c*rel_sc + t*rel_st]
 *        (the reference layout (V,2,T) is rel_sv=2T, rel_sc=T, rel_st=1).
 * nodes: out (N,T,V,2)  node features  V[s,h,:] = rel[h,:,s]            (may be NULL)
 * adj:   out (N,T,V,V)  A[s,h,k] = 1/||rel_h - rel_k|| (0 if equal), A[s,h,h] = 1;
 *        if normalize != 0 the symmetric normalised Laplacian D^-1/2 (D - A) D^-1/2.
 */
int stg_adj_build(const float *rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
                  const int32_t *num_peds, int N, int V, int T, int normalize,
                  float *nodes, float *adj, void *stream);

/* ---------------------------------------------------------------------------------------------
 * R3  the einsum of ConvTemporalGraphical.forward (model.py:67):
 *        y[n,c,t,w] = sum_v x[n,c,t,v] * A[n,t,v,w]         ('nctv,ntvw->nctw')
 * x (N,C,T,V) with strides; A (N,T,V,V) contiguous per scene with batch stride a_sn
 * (a_sn = 0 shares one (T,V,V) adjacency over the batch: the reference's 'nctv,tvw->nctw').
 * y (N,C,T,V) contiguous.  Backward: dx[n,c,t,v] = sum_w dy[n,c,t,w] * A[n,t,v,w] (A is data:
 * no dA).
 */
int stg_spatial_agg_fwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                        const float *adj, int64_t a_sn, const int32_t *num_peds,
                        int N, int C, int T, int V, float *y, void *stream);
int stg_spatial_agg_bwd(const float *dy, const float *adj, int64_t a_sn, const int32_t *num_peds,
                        int N, int C, int T, int V, float *dx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * R3/R4  nn.Conv2d(Cin, Cout, (kt,1), padding=(pad,0)) as built at model.py:55-62 and
 *        model.py:116-122,135-139 (stride 1, dilation 1).
 * x (N,Cin,T,V) strided, w (Cout,Cin,kt,1), b (Cout) or NULL, y (N,Cout,To,V), To = T+2*pad-kt+1.
 * bwd: dx (N,Cin,T,V) (may be NULL), dw/db are ACCUMULATED into (+=) and must be zeroed by the
 * caller (db may be NULL).
 */
int stg_conv_t_fwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                   const float *w, const float *b, const int32_t *num_peds,
                   int N, int Cin, int Cout, int T, int V, int kt, int pad, float *y, void *stream);
int stg_conv_t_bwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                   const float *w, const float *dy, const int32_t *num_peds,
                   int N, int Cin, int Cout, int T, int V, int kt, int pad,
                   float *dx, float *dw, float *db, void *stream);

/* ---------------------------------------------------------------------------------------------
 * R4/R5  st_gcn.forward (model.py:145-155) and social_stgcnn.forward (model.py:182-198) as ONE
 *        scene-resident kernel per direction.
 *
 * Parameters live in one flat fp32 buffer in the reference's named_parameters() order
 * (st_gcns.j: gcn.conv.weight, gcn.conv.bias, tcn.0.{weight,bias}, tcn.1.weight,
 * tcn.2.{weight,bias}, tcn.3.{weight,bias}, [residual.0.{weight,bias}, residual.1.{weight,bias}],
 * prelu.weight; tpcnns.k.{weight,bias}; tpcnn_ouput.{weight,bias}; prelus.k.weight), BatchNorm
 * running statistics in a second flat buffer (per block: tcn.0 mean,var; tcn.3 mean,var;
 * [residual.1 mean,var]).  stg_model_param_count()/stg_model_buffer_count() give the sizes.
 */
typedef struct {
    int32_t n_stgcnn;      /* number of st_gcn blocks (model.py:163-166), 1..STG_MAX_BLOCKS      */
    int32_t n_txpcnn;      /* 0 = no TXP-CNN (stand-alone st_gcn module); else model.py:168-172 */
    int32_t c_in;          /* input_feat                                                        */
    int32_t c_out;         /* output_feat                                                       */
    int32_t t_obs;         /* seq_len                                                           */
    int32_t t_pred;        /* pred_seq_len                                                      */
    int32_t kt;            /* temporal kernel size (odd)                                        */
    int32_t residual0;     /* first block: 0 none (residual=False), 1 identity, 2 conv+BN       */
    int32_t use_mdn;       /* skip the block-final PReLU (model.py:152)                         */
    int32_t bn_mode;       /* 0 eval (running stats), 1 train with per-scene statistics (the     */
                           /* reference's N=1 training loop, train.py:36-77)                    */
    float bn_eps;          /* 1e-5                                                              */
    float bn_momentum;     /* 0.1                                                               */
    int32_t flags;         /* STG_OPT_* bit set, 0 = defaults.  Part of the descriptor because it decides  */
                           /* the workspace / scratch layouts the size queries report                      */
    int32_t wg_waves;      /* 0 = auto; 1, 2, 4 or 8: waves per scene of the workgroup-per-scene kernels   */
} stg_model_desc;

#define STG_OPT_WG_PATH 1     /* run the workgroup-per-scene kernels even where the wave-per-scene path fits */
#define STG_OPT_SPLIT_BF16 2  /* TXP input-gradient GEMMs on bf16 MFMAs with hi/lo-split operands (fp32 in/out) */
#define STG_OPT_BF16_STORE 8  /* bf16 STORAGE of what the forward saves for the backward and of the hand-offs between the  */
                              /* backward's kernels (TXP planes a_l, pre-activations z_l, dz_l): half the bytes; compute,    */
                              /* accumulation, parameters, inputs and V_pred stay fp32 (the forward result is unchanged).    */
                              /* Wave-per-scene path only (one st_gcn block, V <= 68).                                       */
#define STG_OPT_F32_MFMA 16   /* run the TXP convolutions and the weight-gradient GEMM on v_mfma_f32_16x16x4_f32 (the round-1 kernels)
                               * where the default uses v_mfma_f32_16x16x32_bf16 with exact three-piece operands (V <= 32, fp32
                               * storage): same accuracy class, for A/B measurements                                            */
#define STG_OPT_WAVE_PATH 4   /* keep the wave-per-scene kernels for small batches too (default: batches of fewer than  */
                              /* 288 scenes of <= 40 pedestrians run the workgroup kernels, several waves per scene)     */

int64_t stg_model_param_count(const stg_model_desc *d);
int64_t stg_model_buffer_count(const stg_model_desc *d);
/* Per-scene activation workspace (floats) the forward writes for the backward when save != 0. */
int64_t stg_model_ws_floats(const stg_model_desc *d, int V);
/* floats of the BATCH tail of the training workspace, behind the N per-scene blocks: the forward leaves there what the
 * backward needs once per batch (the prepared bf16 operands of the input-gradient GEMMs, written by extra workgroups
 * of the forward's first launch; the scene order of a ragged batch, so that the backward does not sort again -- it
 * must be given the forward's num_peds).  ws holds N * stg_model_ws_floats + stg_model_ws_tail_floats floats.      */
int64_t stg_model_ws_tail_floats(const stg_model_desc *d, int N, int V);
/* Per-scene batch statistics the forward emits in bn_mode 1: (N, stat_floats) =
 * per block, per BatchNorm: mean[C], unbiased var[C].                                           */
int64_t stg_model_stat_floats(const stg_model_desc *d);
/* Scratch floats stg_model_fwd needs (hand-off between its block kernel and its TXP-CNN kernel; 0 when
 * one kernel does both). */
int64_t stg_model_fwd_scratch_floats(const stg_model_desc *d, int N, int V);
/* Scratch floats stg_model_bwd needs (partial-gradient slab rows of its kernels + the dz hand-off
 * between the input-gradient kernel and the weight-gradient kernel). */
int64_t stg_model_bwd_scratch_floats(const stg_model_desc *d, int N, int V);

/* x (N,c_in,t_obs,V) strided; adj (N,t_obs,V,V), batch stride a_sn (0 = shared);
 * y: (N,c_out,t_pred,V) when n_txpcnn>0, else the block output (N,c_out,t_obs,V).
 * ws: N * stg_model_ws_floats + stg_model_ws_tail_floats floats (16-byte aligned) or NULL (inference).  stats: N * stg_model_stat_floats
 * or NULL.  scratch: stg_model_fwd_scratch_floats floats, 16-byte aligned (may be NULL when that is 0).     */
int stg_model_fwd(const stg_model_desc *d, const float *params, const float *buffers,
                  const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                  const float *adj, int64_t a_sn, const int32_t *num_peds, int N, int V,
                  float *y, float *ws, float *stats, float *scratch, void **events, int n_events, void *stream);
/* events (may be NULL): n_events hipEvent_t handles for per-kernel device timing -- the entry point records
 * events[0] on `stream` before its first kernel and events[k] after its k-th kernel, as far as n_events reaches
 * (stg_model_fwd / stg_model_bwd kernel order: see DESIGN.md section 5).                                      */
/* dy like y.  grad_params (param_count) is OVERWRITTEN with the gradient summed over the batch;
 * dx (N,c_in,t_obs,V) may be NULL (the reference never needs it: x is data); a non-NULL dx needs STG_OPT_WG_PATH in
 * the descriptor of BOTH passes (STG_EUNSUPPORTED otherwise: the wave-per-scene kernels compute no input gradient).  scratch: stg_model_bwd_scratch_floats floats, 16-byte aligned
 * (ws must be 16-byte aligned too).                                                               */
int stg_model_bwd(const stg_model_desc *d, const float *params, const float *buffers,
                  const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
                  const float *adj, int64_t a_sn, const int32_t *num_peds, int N, int V,
                  const float *dy, const float *ws, float *scratch, float *grad_params, float *dx,
                  void **events, int n_events, void *stream);

/* stg_model_bwd fused with the loss (train.py:52-74: l = graph_loss(V_pred, V_tr); loss += l; loss.backward()): the
 * backward starts from V_pred itself -- y, the (N, 5, pred, V) output of stg_model_fwd -- and the (N, pred, V, 2)
 * target, computes d(sum_n weights[n] * loss_n)/dV_pred in its input stage (metrics.py:84-113, as stg_nll_fwd does)
 * and writes losses[n] = bivariate_loss of scene n.  weights may be NULL (all ones).  Served by the wave-per-scene
 * and the workgroup-per-scene kernels alike; STG_EUNSUPPORTED (nothing launched, no error message) with
 * STG_OPT_SPLIT_BF16 or without a TXP-CNN -- call stg_nll_fwd + stg_model_bwd then.  No input gradient.            */
int stg_model_bwd_nll(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                      int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                      int N, int V, const float *y, const float *target, const float *weights, float *losses,
                      const float *ws, float *scratch, float *grad_params, void **events, int n_events, void *stream);

/* The whole tail of a single-rank training step behind the fused loss + backward (train.py:58-76,197 + the
 * running-statistics updates of model.py:114,123,140 for the scenes just forwarded): stg_model_bwd_nll whose last launch
 * -- the fixed-order reduction of the partial gradients -- also applies SGD without clipping, p -= lr * grad (grad is
 * still written), folds the forward's per-scene BatchNorm statistics into the running ones (= stg_bn_fold) and writes
 * total[0] = sum_n weights[n] * losses[n] (= stg_weighted_sum), in extra workgroups of the same grid.  With
 * clip_grad_norm_ the update depends on the norm of the complete gradient: use stg_model_bwd_nll + stg_train_tail.   */
typedef struct stg_step_tail {
    float *params;          /* the flat parameters (the same buffer as `params`), updated in place               */
    const float *lr_dev;    /* learning rate in device memory (captured graphs follow StepLR), or NULL: use lr   */
    float lr;
    const float *stats;     /* per-scene batch statistics written by stg_model_fwd, or NULL: no fold             */
    float *buffers;         /* running statistics, updated in place                                              */
    int64_t *const *nbt;    /* num_batches_tracked counters (device pointers), n_bn of them, or NULL             */
    int n_bn;
    float *total;           /* out, 1 float, or NULL                                                             */
} stg_step_tail;
int stg_model_bwd_step(const stg_model_desc *d, const float *params, const float *buffers, const float *x, int64_t x_sn,
                       int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj, int64_t a_sn, const int32_t *num_peds,
                       int N, int V, const float *y, const float *target, const float *weights, float *losses,
                       const float *ws, float *scratch, float *grad_params, const stg_step_tail *tail, void **events,
                       int n_events, void *stream);
/* Sequential-fold update of the BatchNorm running statistics with the per-scene statistics of a
 * batch, exactly as N successive reference forwards would (momentum update per scene,
 * model.py:114,123,140; SURVEY 7 'BatchNorm semantics').  Scenes with num_peds[n] == 0 are skipped.
 * nbt: HOST array of n_bn DEVICE pointers to the int64 num_batches_tracked counters (may be NULL);
 * each is incremented by the number of non-empty scenes.                                          */
int stg_bn_fold(const stg_model_desc *d, const float *stats, const int32_t *num_peds, int N,
                float *buffers, int64_t *const *nbt, int n_bn, void *stream);

/* ---------------------------------------------------------------------------------------------
 * R6  metrics.bivariate_loss (metrics.py:84-113), batched: loss[n] = mean over (P, V_n) of
 *     -log(clamp(pdf, 1e-20)).  pred element (n,f,p,v) at pred[n*p_sn + f*p_sf + p*p_sp + v*p_sv]
 *     (f = mux,muy,log sx,log sy,atanh rho; the model output (N,5,P,V) is p_sf=P*V, p_sp=V, p_sv=1);
 *     target (N,P,V,2) contiguous.  grad (N,5,P,V) contiguous receives grad_scale[n] * d loss[n] / d pred
 *     (grad may be NULL; grad_scale float[N] may be NULL = 1: the caller's per-scene loss weights, so that
 *     grad is directly d(sum_n w_n loss_n)/d pred).
 */
int stg_nll_fwd(const float *pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv,
                const float *target, const int32_t *num_peds, const float *grad_scale, int N, int P, int V,
                float *loss, float *grad, void *stream);
/* out[n,f,p,v] = grad[n,f,p,v] * gloss[n]  (chain rule with the upstream gradient of loss[n]). */
int stg_nll_bwd(const float *grad, const float *gloss, int N, int P, int V, float *out, void *stream);

/* N3  optim.SGD(lr) step without momentum / weight decay (train.py:197): p -= lr * g.          */
int stg_sgd_step(float *params, const float *grads, int64_t count, float lr, void *stream);

/* Ragged batches (the reference's DataLoader yields scenes of 2..57 pedestrians, utils.py:121-193): order[0..N) =
 * scene indices sorted by clamp(num_peds[n], 0, V) descending, stable.  stg_model_fwd / stg_model_bwd run this
 * themselves when num_peds is given (into their scratch buffers) and deal the sorted scenes to their persistent
 * waves boustrophedon, so a padded ragged batch is load-balanced; exported for callers that schedule their own
 * work the same way.  key_start (V+2 ints, may be NULL): key_start[k] = number of scenes with more than V-k
 * pedestrians, so the scenes with at most x pedestrians are order[key_start[V-x] .. N).
 * 2 <= N <= 65536, V <= 1023 (one workgroup, per-wave key histograms in LDS).                                                   */
int stg_scene_order(const int32_t *num_peds, int N, int V, int32_t *order, int32_t *key_start, void *stream);

/* N3  torch.nn.utils.clip_grad_norm_ (train.py:71-73) + optim.SGD step (train.py:197) + the StepLR-scheduled
 *     learning rate (train.py:200) over the flat parameter / gradient buffers in one launch:
 *       total = ||grads||_2;  if max_norm > 0: grads *= min(1, max_norm / (total + 1e-6)) (in place);
 *       params -= lr * grads.   lr is read from device memory when lr_dev != NULL (so that a captured hipGraph
 *     follows the schedule), else the host value `lr`.  grad_norm (1 float, may be NULL) receives `total`.
 *     count <= 2^22 (single-workgroup kernel; the model has 7,563 parameters).                              */
int stg_optim_step(float *params, float *grads, int64_t count, const float *lr_dev, float lr, float max_norm,
                   float *grad_norm, void *stream);

/* ---------------------------------------------------------------------------------------------
 * N1  TrajectoryDataset / DataLoader collation (utils.py:121-193, train.py:167-177) with the windowed dataset resident
 *     in HBM: seq_rel_all = ragged concatenation of every window's relative trajectories (total_peds, 2, T_obs+T_pred);
 *     win_start int32[n_windows+1] = pedestrian offsets of the windows; index int32[N] (DEVICE memory, NULL = windows
 *     0..N-1) = the windows of this batch.  Writes obs_rel (N,V,2,T_obs) -- the input of stg_adj_build --, target
 *     (N,T_pred,V,2) and num_peds (N), zero-padded to V slots (a window with more than V pedestrians is truncated:
 *     pad to the dataset's largest crowd).  No host synchronisation: the whole training step can be captured with the
 *     index refreshed in place.                                                                                     */
int stg_gather_windows(const float *seq_rel_all, const int32_t *win_start, const int32_t *index, int n_windows, int N,
                       int V, int T_obs, int T_pred, float *obs_rel, float *target, int32_t *num_peds, void *stream);

/* Data-parallel training over scene-windows (the reference has no multi-GPU path): ONE all-reduce(sum) per
 * optimizer step carries the flat gradient AND the exact sequential fold of the BatchNorm running statistics over the
 * ranks ("R ranks x B scenes == one rank on the concatenated batch").
 *   stg_dp_pack : pack = [ grads (n_params) | world slots of (n_buffers + 1) floats ]; the slot of `rank` receives
 *                 acc = bn_after - (1-momentum)^{n_r} * bn_before and n_r = the rank's non-empty scenes (num_peds[i] > 0;
 *                 num_peds NULL = N), every other slot 0.  bn_before / bn_after: the flat running statistics before /
 *                 after this rank's stg_bn_fold.  pack holds n_params + world * (n_buffers + 1) floats.
 *   (the caller all-reduces `pack` with SUM over the ranks: RCCL on MI355X)
 *   stg_dp_fold : buffers = bn_before * keep^{sum_r n_r} + sum_r acc_r * keep^{sum_{j>r} n_j}, keep = 1 - momentum;
 *                 the first n_params floats of `pack` are the summed gradient (feed them to stg_optim_step).
 *                 nbt (HOST array of n_bn DEVICE pointers to the int64 num_batches_tracked counters, may be NULL): each
 *                 counter, already advanced by this rank's own scenes (stg_bn_fold), also receives the scene counts of
 *                 the OTHER ranks from the pack -- afterwards every rank holds the single-process value.              */
int stg_dp_pack(const float *grads, const float *bn_before, const float *bn_after, const int32_t *num_peds, int N,
                float momentum, int rank, int world, int n_params, int n_buffers, float *pack, void *stream);
int stg_dp_fold(const float *pack, const float *bn_before, float momentum, int rank, int world, int n_params,
                int n_buffers, float *buffers, int64_t *const *nbt, int n_bn, void *stream);
/* out[0] = sum_n weights[n] * values[n] (weights NULL: plain sum), fixed summation order: the reported group loss of
 * train.train (train.py:58-67,76) from the per-scene losses of stg_nll_fwd.                                          */
int stg_weighted_sum(const float *values, const float *weights, int N, float *out, void *stream);

/* The tail of a single-rank training step in ONE launch (the three jobs do not depend on each other; they run in
 * different workgroups): the BatchNorm running-statistics fold of the forward just done (= stg_bn_fold; stats NULL
 * skips it), total[0] = sum_n weights[n] * losses[n] (= stg_weighted_sum; losses or total NULL skips it) and
 * clip_grad_norm_ + SGD on the flat parameters (= stg_optim_step).  Replaces train.py:58-76 + the per-forward
 * running-statistics updates of model.py:114,123,140 for one group of scenes.                                        */
int stg_train_tail(const stg_model_desc *d, const float *stats, const int32_t *num_peds, int N, float *buffers,
                   int64_t *const *nbt, int n_bn, const float *losses, const float *weights, float *total,
                   float *params, float *grads, int64_t count, const float *lr_dev, float lr, float max_norm,
                   float *grad_norm, void *stream);

/* ---------------------------------------------------------------------------------------------
 * N2  evaluation tail of test.test (test.py:59-123) + metrics.ade/fde/nodes_rel_to_nodes_abs
 *     (metrics.py:21-75): per pedestrian the best-of-K average / final displacement error of K trajectories
 *     sampled from the predicted bivariate Gaussians.  pred as in stg_nll_fwd; target_rel (N,P,V,2) ground-truth
 *     displacements; obs_last (N,V,2) last observed absolute position (may be NULL: the errors are translation
 *     invariant up to fp32 rounding); noise: K*N*P*V*2 standard normals laid out (K,N,P,V,2), or NULL to draw
 *     them in the kernel (Philox4x32-10 keyed by `seed`, counter = (scene*V + ped, k*P + t)).
 *     sample = mean + chol(cov) * eps, trajectory = cumsum_t(sample) + obs_last; ade/fde (N,V) receive
 *     min_k mean_t |traj - truth| and min_k |traj_P - truth_P| (0 for padded pedestrians).
 */
int stg_bestofk_eval(const float *pred, int64_t p_sn, int64_t p_sf, int64_t p_sp, int64_t p_sv,
                     const float *target_rel, const float *obs_last, const int32_t *num_peds, const float *noise,
                     uint64_t seed, int N, int P, int V, int K, float *ade, float *fde, void *stream);

/* Self-test helper: C(16x16) = A(16xK) * B(Kx16) through v_mfma_f32_16x16x4_f32 with the operand
 * maps the TXP-CNN kernels rely on (K multiple of 4).                                           */
int stg_selftest_mfma(const float *a, const float *b, int K, float *c, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* STGCNN_HIP_H */
