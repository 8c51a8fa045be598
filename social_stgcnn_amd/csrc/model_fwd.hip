// model_fwd: st_gcn.forward (model.py:145-155) + social_stgcnn.forward (model.py:182-198) as ONE
// scene-resident kernel.  A workgroup of WAVES wave64 owns one scene-window at a time: the scene's
// activations never leave LDS between the adjacency-weighted aggregation, the temporal
// Conv+BN+PReLU block and the TXP-CNN; HBM sees x and A once (coalesced along w, then t) and the
// (5,12,V) output once (plus, in training, the activations the backward needs).
//
//  st_gcn block (VALU, 3% of the flops): lanes own (t,w) columns; the einsum is re-associated to
//    aggregate the Cin input channels first (g = Wg (x A) + bg colsum(A)), BatchNorm statistics are
//    PER SCENE (the reference trains with N=1, train.py:173-177) via wave-shuffle + LDS reductions.
//  TXP-CNN (97% of the flops): every 3x3 conv over the (5, V) plane is an implicit GEMM on
//    v_mfma_f32_16x16x4_f32 (exact fp32): M = 12 output channels (padded to 16), N = 16 positions
//    of the scene, K = (tap, 4 input channels).  Weights live in VGPRs (one per K-step), the im2col
//    B operand is one ds_read_b32 per MFMA from a zero-bordered LDS plane whose channel stride is
//    == 16 (mod 32) dwords (conflict-free for the four K-lane groups).
//
// On the fast path (one st_gcn block, V <= 68) this kernel runs the block only (F1) and hands a_0 to the
// wave-per-scene TXP kernel of txp_wave.hip (F2) in that kernel's in-place plane layout.
#include "model_common.hpp"
#include "stgcn_block.hpp"
#include "txp_wave.hpp"

namespace stg {

struct FwdArgs {
    ModelLayout lay;
    const float *params, *buffers, *x;
    int64_t x_sn, x_sc, x_st, x_sv;
    const float *adj;
    int64_t a_sn;
    const int32_t *num_peds;
    int N, V;
    float *y, *ws;
    int64_t ws_stride;
    float *stats;
    const int32_t *order;   // non-null: scenes sorted by crowd size (descending); block b takes order[b]
    const float *agg;       // per scene [agg_stride]: ax at agg_ax, cs at agg_cs (stgcn_agg_kernel, block 0)
    int64_t agg_stride, agg_ax, agg_cs;
    int debug_skip;   // diagnostic builds only (STG_DEBUG_SKIP): 16 TXP-CNN, 32 st_gcn -- wrong results
};

// floats of the kernel's TXP plane buffer: two-plane layout [P][txp_sc(V)]
__host__ __device__ inline int fwd_plane_floats(int V) { return Cfg::P * txp_sc(V); }

// ------------------------------------------------------------------------------------------
// TXP-CNN layer on MFMA
// ------------------------------------------------------------------------------------------
template <int CINL>
__device__ __forceinline__ void txp_load_weights(const float *__restrict__ W, float (&wreg)[CINL * 9 / 4]) {
    const int lane = threadIdx.x & 63, co = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int j = 0; j < CINL / 4; ++j) {
            const int ci = 4 * j + kq;
            wreg[tap * (CINL / 4) + j] = co < Cfg::P ? W[(co * CINL + ci) * 9 + tap] : 0.f;
        }
}

// kind: 0 = hidden layer without residual (layer 0), 1 = hidden layer with residual, 2 = output conv
template <int CINL, int WAVES>
__device__ void txp_layer_fwd(const float *__restrict__ W, const float *__restrict__ bias, float alpha, int kind,
                              const float *in, float *out, int vi, int V, float *zsave, float *yout) {
    constexpr int C = Cfg::C, P = Cfg::P, KS = CINL * 9 / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = lane & 15, kq = lane >> 4;
    const int SW = txp_sw(vi), SC = txp_sc(vi);
    const int npos = C * vi, ntiles = (npos + 15) >> 4;
    float wreg[KS];
    txp_load_weights<CINL>(W, wreg);
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) binit[r] = (4 * kq + r) < P ? bias[4 * kq + r] : 0.f;

    for (int tile0 = wave * 2; tile0 < ntiles; tile0 += WAVES * 2) {
        int pos[2], hh[2], ww[2], base[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            pos[u] = (tile0 + u) * 16 + nq;
            ok[u] = pos[u] < npos;
            const int pc = ok[u] ? pos[u] : 0;
            hh[u] = pc / vi;
            ww[u] = pc - hh[u] * vi;
            base[u] = kq * SC + hh[u] * SW + ww[u];
        }
        f32x4 acc0 = binit, acc1 = binit;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * SW + (tap % 3);
#pragma unroll
            for (int j = 0; j < CINL / 4; ++j) {
                const float b0 = in[base[0] + 4 * j * SC + toff];
                const float b1 = in[base[1] + 4 * j * SC + toff];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * (CINL / 4) + j], b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap * (CINL / 4) + j], b1, acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!ok[u] || kq == 3) continue;
            const f32x4 acc = u == 0 ? acc0 : acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 4 * kq + r;
                const float z = acc[r];
                const int flat = (co * C + hh[u]) * vi + ww[u];
                if (kind == 2) {
                    // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
                    yout[(int64_t)(co * C + hh[u]) * V + ww[u]] = z;
                } else {
                    const int li = co * SC + (hh[u] + 1) * SW + (ww[u] + 1);
                    float av = z > 0.f ? z : alpha * z;
                    if (kind == 1) av += in[li];
                    out[li] = av;
                    if (zsave) zsave[flat] = z;
                }
            }
        }
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void model_fwd_kernel(const FwdArgs a, const float *__restrict__ params,
                                                               const float *__restrict__ buffers) {
    // params / buffers are separate __restrict__ kernel arguments on purpose: only then can the compiler prove
    // that the kernel's own stores never clobber them and fetch the (wave-uniform) weights with scalar loads
    // into SGPRs instead of per-lane vector loads + s_waitcnt (340 vector loads per scene otherwise)
    constexpr int C = Cfg::C, T = Cfg::T, P = Cfg::P, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int V = a.V, tid = threadIdx.x;
    const int xin_floats = C * T * V;
    const int plane_floats = fwd_plane_floats(V);
    const int reg_floats = plane_floats > 3 * C * T * V ? plane_floats : 3 * C * T * V;
    float *bufA = sm;
    float *reg = bufA + plane_floats;
    float *red = reg + reg_floats;
    const ModelLayout &L = a.lay;
    const int c_in = L.blk[0].cin;
    const int out_rows = L.n_txp > 0 ? C * P : C * T;

    for (int it = blockIdx.x; it < a.N; it += gridDim.x) {
        const int n = a.order ? a.order[it] : it;       // largest crowds are dispatched first
        int vi = a.num_peds ? a.num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        float *yn = a.y + (int64_t)n * out_rows * V;
        // padded pedestrian slots of the output are zeros
        if (vi < V)
            for (int e = tid; e < out_rows * (V - vi); e += NT) {
                const int r = e / (V - vi), w = vi + (e - r * (V - vi));
                yn[(int64_t)r * V + w] = 0.f;
            }
        if (vi == 0) continue;
        float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;
        float *statn = a.stats ? a.stats + (int64_t)n * L.stat_floats : nullptr;
        const int SW = txp_sw(vi), SC = txp_sc(vi);
        __syncthreads();
        float *X = reg, *G = reg + xin_floats, *H = G + C * T * V;
        // stage x[n] (strided: the caller's permute(0,3,1,2) view, train.py:48)
        {
            const float *xn = a.x + n * a.x_sn;
            for (int e = tid; e < c_in * T * vi; e += NT) {
                const int v = e % vi, ct = e / vi, t = ct % T, c = ct / T;
                X[e] = xn[c * a.x_sc + t * a.x_st + v * a.x_sv];
            }
            if (L.n_txp > 0) {                       // (both sizes are multiples of 16 floats)
                float4 *z4 = reinterpret_cast<float4 *>(bufA);
                for (int e = tid; e < ((P * SC) >> 2); e += NT) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        for (int j = 0; j < L.n_blocks && !STG_SKIP(a, 32); ++j) {
            const bool last = j == L.n_blocks - 1;
            float *yb = (last && L.n_txp == 0) ? yn : nullptr;
            // block 0 starts from the aggregated input stgcn_agg_kernel left (training: in the workspace itself)
            const float *pax = j == 0 ? a.agg + n * a.agg_stride + a.agg_ax : nullptr;
            const float *pcs = j == 0 ? a.agg + n * a.agg_stride + a.agg_cs : nullptr;
            if (L.blk[j].cin == Cfg::CIN0)
                stgcn_block_fwd<Cfg::CIN0, WAVES>(a, params, buffers, L.blk[j], n, vi, X, G, H, red, wsn, statn, pax, pcs,
                                                  last && L.n_txp > 0, bufA, SC, bufA, 0, yb, !last);
            else
                stgcn_block_fwd<Cfg::C, WAVES>(a, params, buffers, L.blk[j], n, vi, X, G, H, red, wsn, statn, pax, pcs,
                                               last && L.n_txp > 0, bufA, SC, bufA, 0, yb, !last);
            float *tmp = X; X = H; H = tmp;      // block output becomes the next block's input
        }
        if (L.n_txp == 0 || STG_SKIP(a, 16)) continue;
        // ---- TXP-CNN (model.py:187-195) ------------------------------------------------------
        float *bufB = reg;
        for (int e = tid; e < P * SC; e += NT) bufB[e] = 0.f;
        __syncthreads();
        const float *__restrict__ Pm = params;
        float *in = bufA, *out = bufB;
        // a_l leaves as a whole zero-bordered plane, transposed to position-major [(C+2)*SW][P] with
        // coalesced stores: the layout the weight-gradient GEMM (K = positions, 16 lanes = 16 channels/taps)
        // stages back with a linear LDS-DMA copy and reads without bank conflicts
        auto save_plane = [&](const float *pl, int idx) {
            if (!wsn) return;
            float *dst = wsn + ws_plane_off(L, V, idx);     // the C interior rows with their border columns, [C*SW][P]
            for (int h = 0; h < C; ++h)
                for (int e = tid; e < SW * P; e += NT) {
                    const int col = e / P, ch = e - col * P;
                    dst[(h * SW) * P + e] = pl[ch * SC + (h + 1) * SW + col];
                }
        };
        save_plane(in, 0);
        for (int l = 0; l < L.L; ++l) {
            float *zs = wsn ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr;
            const float alpha = Pm[L.prelus + l];
            if (l == 0)
                txp_layer_fwd<Cfg::T, WAVES>(Pm + L.txp_w[0], Pm + L.txp_b[0], alpha, 0, in, out, vi, V, zs, nullptr);
            else
                txp_layer_fwd<Cfg::P, WAVES>(Pm + L.txp_w[l], Pm + L.txp_b[l], alpha, 1, in, out, vi, V, zs, nullptr);
            __syncthreads();
            float *tmp = in; in = out; out = tmp;
            save_plane(in, l + 1);
        }
        txp_layer_fwd<Cfg::P, WAVES>(Pm + L.out_w, Pm + L.out_b, 0.f, 2, in, out, vi, V, nullptr, yn);
    }
}

static size_t fwd_lds_bytes(int V, int waves) {
    const int plane = fwd_plane_floats(V);
    const int reg = plane > 3 * Cfg::C * Cfg::T * V ? plane : 3 * Cfg::C * Cfg::T * V;
    return (size_t)(plane + reg + waves * 16 + 16) * sizeof(float);
}

// scratch carve of stg_model_fwd: [aggregated input of block 0: N x (cin + 1) T V | scene order | (diag) stamps]
static int64_t fwd_agg_floats(const ModelLayout &l, int N, int V) {
    return (((int64_t)N * (l.blk[0].cin + 1) * Cfg::T * V + 3) & ~(int64_t)3) + 4;
}

// ... | prepared forward operands of the exact-bf16 convs (16-byte vectors)]
static int64_t fwd_wp_off(const ModelLayout &l, int N, int V, bool stamps) {
    return (fwd_agg_floats(l, N, V) + order_floats(N, V) + (stamps ? (int64_t)N * 32 : 0) + 3) & ~(int64_t)3;
}

}  // namespace stg

// batch tail of the training workspace: [prepared operands of the exact-bf16 input-gradient chain | scene order]
namespace stg {
int64_t ws_tail_wp_floats(const ModelLayout &l, int V) { return txp_bwd_x6_fits(l, V) ? txp_bwd_x6_wp_floats(l) : 0; }
}
extern "C" int64_t stg_model_ws_tail_floats(const stg_model_desc *d, int N, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (N < 0 || V <= 0) return stg::fail(STG_EINVAL, "stg_model_ws_tail_floats: N=%d V=%d", N, V);
    return stg::ws_tail_wp_floats(l, V) + stg::order_floats(N, V);
}

extern "C" int64_t stg_model_fwd_scratch_floats(const stg_model_desc *d, int N, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (N < 0 || V <= 0) return stg::fail(STG_EINVAL, "stg_model_fwd_scratch_floats: N=%d V=%d", N, V);
    const bool stamps = stg::diag_env("STG_STAMPS", 0) != 0;
    return stg::fwd_wp_off(l, N, V, stamps) + (stg::txp_fwd_x6_fits(l, V) ? stg::txp_bwd_x6_wp_floats(l) : 0);
}

extern "C" int stg_model_fwd(const stg_model_desc *d, const float *params, const float *buffers, const float *x,
                             int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj,
                             int64_t a_sn, const int32_t *num_peds, int N, int V, float *y, float *ws,
                             float *stats, float *scratch, void **events, int n_events, void *stream) {
    using namespace stg;
    FwdArgs a{};
    const int rc = make_layout(d, &a.lay);
    if (rc != STG_OK) return rc;
    STG_REQUIRE(N >= 0 && V > 0, STG_EINVAL, "stg_model_fwd: bad sizes N=%d V=%d", N, V);
    if (N == 0) return STG_OK;                        // empty batch: nothing to read or write
    STG_REQUIRE(params && buffers && x && adj && y, STG_EINVAL, "stg_model_fwd: null pointer");
    STG_REQUIRE(scratch && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0, STG_EINVAL,
                "stg_model_fwd: scratch (stg_model_fwd_scratch_floats floats, 16-byte aligned) is missing");
    STG_REQUIRE(!ws || (reinterpret_cast<uintptr_t>(ws) & 15) == 0, STG_EINVAL, "stg_model_fwd: ws must be 16-byte aligned");
    const ModelLayout &L = a.lay;
    a.params = params; a.buffers = buffers; a.x = x;
    a.x_sn = x_sn; a.x_sc = x_sc; a.x_st = x_st; a.x_sv = x_sv;
    a.adj = adj; a.a_sn = a_sn; a.num_peds = num_peds; a.N = N; a.V = V;
    a.y = y; a.ws = ws; a.ws_stride = ws_floats_per_scene(L, V); a.stats = stats;
    int auto_waves = 0;
    const bool wave_path = use_wave_path(L, N, V, &auto_waves);
    STG_REQUIRE(wave_path || !(L.flags & STG_OPT_BF16_STORE), STG_EUNSUPPORTED,
                "stg_model_fwd: bf16 storage (STG_OPT_BF16_STORE) is built for the wave-per-scene kernels only "
                "(one st_gcn block, input_feat 2, V <= 68, no STG_OPT_WG_PATH)");
    hipStream_t st = as_stream(stream);
    EventList evl{events, events ? n_events : 0, 0, st};
    evl.mark();
    // K0: x A and colsum(A) of the first block -- the only read of A.  Training: straight into the saved arrays of
    // the workspace (the backward needs exactly these); inference: into the scratch buffer.
    const BlockLayout &b0 = L.blk[0];
    if (ws) {
        a.agg = ws; a.agg_stride = a.ws_stride;
        a.agg_ax = L.ws_hdr_floats + (int64_t)b0.ws_ax * V;
        a.agg_cs = L.ws_hdr_floats + (int64_t)b0.ws_cs * V;
    } else {
        a.agg = scratch; a.agg_stride = (int64_t)(b0.cin + 1) * Cfg::T * V;
        a.agg_ax = 0;
        a.agg_cs = (int64_t)b0.cin * Cfg::T * V;
    }
    a.debug_skip = diag_env("STG_DEBUG_SKIP", 0);
    bool sorted_in_agg = false;
    {
        // training on the wave-per-scene path: the launch also prepares the backward's exact-bf16 A operands into the
        // batch tail of the workspace (stg_model_ws_tail_floats behind the N per-scene blocks)
        AggPrep prep{};
        prep.params = params;
        prep.n_layers = L.L + 1;
        for (int l = 0; l <= L.L; ++l) prep.w_off[l] = l < L.L ? L.txp_w[l] : L.out_w;
        if (ws && wave_path && txp_bwd_x6_fits(L, V)) prep.wp = reinterpret_cast<unsigned *>(ws + (int64_t)N * a.ws_stride);
        // ... and the forward's own (this launch's successor reads them from the scratch buffer)
        if (wave_path && txp_fwd_x6_fits(L, V))
            prep.wp_fwd = reinterpret_cast<unsigned *>(scratch + fwd_wp_off(L, N, V, diag_env("STG_STAMPS", 0) != 0));
        // ... and, for a ragged batch on the wave-per-scene path, the scene order: one more workgroup of this launch
        // (training: into the workspace's batch tail, where the backward finds it -- no second sort)
        int32_t *order0 = reinterpret_cast<int32_t *>(scratch + fwd_agg_floats(L, N, V));
        if (ws && wave_path) order0 = reinterpret_cast<int32_t *>(ws + (int64_t)N * a.ws_stride + ws_tail_wp_floats(L, V));
        sorted_in_agg = wave_path && agg_sorts(num_peds, N, V);
        if (sorted_in_agg) { prep.order = order0; prep.key_start = order0 + N; prep.order_peds = order0 + N + V + 2; }
        const int rca = launch_stgcn_agg(b0.cin, x, x_sn, x_sc, x_st, x_sv, adj, a_sn, num_peds, N, V,
                                         const_cast<float *>(a.agg), a.agg_stride, a.agg_ax, a.agg_cs, st,
                                         (prep.wp || prep.wp_fwd || prep.order) ? &prep : nullptr);
        if (rca != STG_OK) return rca;
    }
    evl.mark();
    int32_t *order = reinterpret_cast<int32_t *>(scratch + fwd_agg_floats(L, N, V));
    if (wave_path) {
        // K1: ONE wave-per-scene kernel for the whole model: st_gcn block (VALU) + TXP-CNN (MFMA), the a_0 plane never
        // leaves LDS.  Ragged batch: sorted scene list, walked boustrophedon.  Training: the order (with its tier offsets
        // and the sorted counts) goes to the workspace's batch tail, where the backward finds it -- no second sort.
        if (ws) order = reinterpret_cast<int32_t *>(ws + (int64_t)N * a.ws_stride + ws_tail_wp_floats(L, V));
        bool sorted = sorted_in_agg;                   // (sorted by a workgroup of the aggregation launch above ...)
        if (!sorted) {                                  // (... or, for a large batch, by the 16-wave kernel)
            sorted = launch_scene_order(num_peds, N, V, order, order + N, st, order + N + V + 2);
            if (!sorted && scene_order_applies(num_peds, N, V)) return hip_fail(hipErrorLaunchFailure, "stg_model_fwd: scene_order launch");
        }
        TxpFwdArgs t{};
        t.lay = L; t.params = params; t.buffers = buffers; t.num_peds = num_peds; t.N = N; t.V = V;
        t.x = x; t.x_sn = x_sn; t.x_sc = x_sc; t.x_st = x_st; t.x_sv = x_sv;
        t.adj = adj; t.a_sn = a_sn;
        t.agg = a.agg; t.agg_stride = a.agg_stride; t.agg_ax = a.agg_ax; t.agg_cs = a.agg_cs;
        t.y = y; t.ws = ws; t.ws_stride = a.ws_stride; t.stats = stats;
        if (txp_fwd_x6_fits(L, V))
            t.wpf = reinterpret_cast<const unsigned *>(scratch + fwd_wp_off(L, N, V, diag_env("STG_STAMPS", 0) != 0));
        t.stamps = diag_env("STG_STAMPS", 0) ? reinterpret_cast<unsigned long long *>(scratch + fwd_agg_floats(L, N, V) + order_floats(N, V)) : nullptr;
        const int serp = diag_env("STG_WALK", 1);
        // one launch for the whole (sorted) batch: V-tiers in separate launches were measured slower -- the few
        // large scenes of a real batch take one wave tens of microseconds each and need the small ones to overlap
        t.tier = SceneTier{sorted ? order : nullptr, sorted ? order + N : nullptr, -1, V, serp};
        t.Vl = V;
        t.debug_skip = a.debug_skip;
        t.stagger = diag_env("STG_STAGGER_F", 0);
        const int rcw = launch_txp_fwd_wave(t, st);
        evl.mark();
        evl.finish();
        return rcw;
    }
    int waves = V <= 12 ? 1 : (V <= 40 ? 2 : (V <= 80 ? 4 : 8));
    if (auto_waves) waves = auto_waves;
    const size_t lds = fwd_lds_bytes(V, waves);
    STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "stg_model_fwd: V=%d needs %zu bytes of LDS (> %d)", V, lds,
                kLdsBytes);
    const dim3 grid((unsigned)N);
#define STG_LAUNCH_FWD(W)                                                                                    \
    do {                                                                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&model_fwd_kernel<W>),            \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
        if (e_ != hipSuccess) return hip_fail(e_, "stg_model_fwd: hipFuncSetAttribute");                     \
        hipLaunchKernelGGL(model_fwd_kernel<W>, grid, dim3(W * 64), lds, st, a, params, buffers);                             \
    } while (0)
    switch (waves) {
        case 1: STG_LAUNCH_FWD(1); break;
        case 2: STG_LAUNCH_FWD(2); break;
        case 4: STG_LAUNCH_FWD(4); break;
        default: STG_LAUNCH_FWD(8); break;
    }
#undef STG_LAUNCH_FWD
    STG_LAUNCH_CHECK("stg_model_fwd");
    evl.mark();
    evl.finish();
    return STG_OK;
}
