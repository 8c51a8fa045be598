// Layout of the flat parameter / buffer / workspace arrays shared by the fused scene-resident
// kernels (model_fwd.hip, model_bwd.hip) and their host entry points.
#pragma once
#include "common.hpp"

namespace stg {

constexpr int kMaxTxp = 8;       // n_txpcnn supported by the fused kernels
constexpr int kWsHdrPerBlock = 64;  // per-scene, per-block saved BatchNorm statistics (6*C <= 64)

// Compile-time model family the fused kernels are instantiated for (reference defaults,
// train.py:127-136): input_feat 2 -> output_feat 5, obs 8 -> pred 12, temporal kernel 3.
struct Cfg {
    static constexpr int CIN0 = 2, C = 5, T = 8, P = 12, KT = 3;
};

struct BlockLayout {
    int32_t cin, residual;      // residual: 0 none, 1 identity, 2 conv+BN
    // offsets (floats) into the flat parameter buffer, reference named_parameters() order
    int32_t gcn_w, gcn_b, bn1_g, bn1_b, prelu1, tcn_w, tcn_b, bn2_g, bn2_b, res_w, res_b, bnr_g, bnr_b, prelu_o;
    int32_t buf;                // running stats: bn1 mean,var | bn2 mean,var | [bnr mean,var], C floats each
    int32_t stat;               // offset into the per-scene batch-statistics row (same order: mean, unbiased var)
    int32_t n_bn;               // 2 or 3
    // per-scene workspace: header slot (floats) and arrays in units of V floats
    int32_t ws_hdr;             // mean1,rstd1,mean2,rstd2,meanr,rstdr (C each)
    int32_t ws_ax, ws_cs, ws_g, ws_h2, ws_s;
};

struct ModelLayout {
    int32_t n_blocks, n_txp, L;     // L = hidden TXP layers actually used = max(1, n_txp-1), 0 if n_txp==0
    BlockLayout blk[STG_MAX_BLOCKS];
    int32_t txp_w[kMaxTxp], txp_b[kMaxTxp], out_w, out_b, prelus;
    int32_t n_params, n_buffers, stat_floats;
    int32_t ws_hdr_floats;          // fixed part of the per-scene workspace
    int32_t ws_z[kMaxTxp];          // z_0 .. z_{L-1} (pre-activation, [P][C][vi]), units of V floats
    int32_t ws_units;               // total units of V floats (block arrays + z)
    int32_t n_planes;               // a_0 .. a_L saved as zero-bordered planes [P][txp_sc(V)] behind the units
    // small-parameter map of the backward: block parameters [0, n_blk_params) and the PReLU slopes
    int32_t n_blk_params;
    int32_t use_mdn, bn_mode;
    float eps, momentum;
    int32_t flags, wg_waves;        // stg_model_desc.flags / .wg_waves
};

// Fills `lay` from the public descriptor; returns STG_OK or an error code (message in last_error).
int make_layout(const stg_model_desc *d, ModelLayout *lay);

typedef float f32x4 __attribute__((ext_vector_type(4)));

// bf16 storage (STG_OPT_BF16_STORE): round-to-nearest-even fp32 -> bf16, four values packed into 8 bytes
__device__ __forceinline__ unsigned bf16_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4 &v) {
    return make_uint2(bf16_bits(v[0]) | (bf16_bits(v[1]) << 16), bf16_bits(v[2]) | (bf16_bits(v[3]) << 16));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(const uint2 &u) {
    return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                 __uint_as_float(u.y & 0xffff0000u)};
}
// Row strides (in positions) of the SAVED position-major arrays.  fp32: the natural ones (vi + 2 with the border
// columns for a plane, vi for dz).  bf16 storage: rounded up to even, so that every row of 24-byte positions starts
// 16-byte aligned -- the weight-gradient kernel stages rows and column chunks with 16-byte LDS-DMA.
__host__ __device__ inline int save_sw(int vi, bool bf16) { return bf16 ? (vi + 3) & ~1 : vi + 2; }
__host__ __device__ inline int save_vw(int vi, bool bf16) { return bf16 ? (vi + 1) & ~1 : vi; }

// a 4-channel vector of a position-major [pos][12] array: 16 bytes in fp32, 8 bytes in bf16 storage
__device__ __forceinline__ void store_vec4(float *base, int vec, const f32x4 &v, bool bf16) {
    if (bf16) reinterpret_cast<uint2 *>(base)[vec] = pack_bf16x4(v);
    else reinterpret_cast<f32x4 *>(base)[vec] = v;
}
__device__ __forceinline__ f32x4 load_vec4(const float *base, int vec, bool bf16) {
    if (bf16) return unpack_bf16x4(reinterpret_cast<const uint2 *>(base)[vec]);
    return reinterpret_cast<const f32x4 *>(base)[vec];
}
// The same in two steps, for several loads in flight: the raw bits now (bf16 storage: the 8 bytes in lanes 0 / 1 of the
// vector), the conversion after they have all landed -- converting right behind each load makes the loads wait for
// one another.
__device__ __forceinline__ f32x4 load_vec4_raw(const float *base, int vec, bool bf16) {
    if (bf16) {
        const uint2 r = reinterpret_cast<const uint2 *>(base)[vec];
        return f32x4{__uint_as_float(r.x), __uint_as_float(r.y), 0.f, 0.f};
    }
    return reinterpret_cast<const f32x4 *>(base)[vec];
}
__device__ __forceinline__ f32x4 finish_vec4(const f32x4 &raw, bool bf16) {
    return bf16 ? unpack_bf16x4(make_uint2(__float_as_uint(raw[0]), __float_as_uint(raw[1]))) : raw;
}

// LDS geometry of one padded TXP plane set: rows = C + 2, row stride SW = vi + 2, channel stride SC
// chosen == 16 (mod 32) so the four K-lanes groups of a 16x16x4 B-operand read hit disjoint banks.
__host__ __device__ inline int txp_sw(int vi) { return vi + 2; }
__host__ __device__ inline int txp_sc(int vi) {
    const int raw = (Cfg::C + 2) * (vi + 2);
    return raw + ((16 - (raw & 31)) & 31);
}
// floats of one saved plane slot (sized for the padded batch V).  Planes are SAVED position-major,
// [(C+2)*(vi+2) padded positions][P channels] (what the weight-gradient GEMM reads conflict-free); the
// slot is sized by the channel-major LDS form P*txp_sc(V) >= P*(C+2)*(V+2).
__host__ __device__ inline int plane_slot(int V) { return Cfg::P * txp_sc(V); }
// The wave-per-scene TXP FORWARD keeps ONE plane per scene and updates it in place: C + 4 row slots; a layer reads
// its input at slot offset 2 (even layers) or 0 (odd layers) and writes row r of its output two slots away from
// where it read row r -- towards the rows it has already consumed (even layers walk the positions upwards, odd
// layers downwards), so no pending read ever sees a new value.  Channel stride == 16 (mod 32) as for txp_sc.
__host__ __device__ inline int txp_sci(int vi) {
    const int raw = (Cfg::C + 4) * (vi + 2);
    return raw + ((16 - (raw & 31)) & 31);
}
// floats of one dz_l hand-off slot: position-major [C*V positions][P channels] (padded batch V)
__host__ __device__ inline int dz_slot(int V) { return Cfg::P * Cfg::C * V; }

// offset (floats, from the scene's workspace base, 16-byte aligned) of saved plane a_l
__host__ __device__ inline int64_t ws_plane_off(const ModelLayout &l, int V, int idx) {
    const int64_t arrays = ((int64_t)l.ws_hdr_floats + (int64_t)l.ws_units * V + 3) & ~(int64_t)3;
    return arrays + (int64_t)idx * plane_slot(V);
}
__host__ __device__ inline int64_t ws_floats_per_scene(const ModelLayout &l, int V) {
    return ws_plane_off(l, V, l.n_planes);
}

// ---- ragged batches: balanced static schedules ------------------------------------------------
// The persistent kernels give worker w of G (a wave or a workgroup) the scenes of a fixed walk.  With per-scene
// work ~ V_n a plain stride walk is as unbalanced as the batch is ragged, so when num_peds is given the entry
// points first sort the scenes by V_n (descending, stable: `launch_scene_order`) and the workers walk the sorted
// list boustrophedon: round r hands worker w item r*G + w (r even) or r*G + G-1-w (r odd).
// walk_item() returns the list position of (round r, worker w), or -1 when that slot is past the end.
__device__ __forceinline__ int walk_item(int r, int w, int G, int N, bool serpentine) {
    const int i = r * G + ((serpentine && (r & 1)) ? G - 1 - w : w);
    return i < N ? i : -1;
}
// a scene index read from the sorted list, forced into [0, N): the backward takes the list the forward left in the
// workspace's batch tail on trust (include/stgcnn_hip.h: the same descriptor and num_peds for both passes)
__host__ __device__ inline int scene_index(int i, int N) { return i < 0 ? 0 : (i >= N ? N - 1 : i); }
// V-tier of a launch: the scenes with v_lo < V_n <= v_hi, a contiguous range of the sorted list
struct SceneTier {
    const int32_t *order;      // sorted scene list or null (then the tier is the whole batch in batch order)
    const int32_t *key_start;  // tier offsets of the sorted list (see scene_order_kernel) or null
    int v_lo, v_hi;
    int serpentine;            // walk the sorted list boustrophedon (1) or with a plain stride (0)
};
__device__ __forceinline__ void tier_range(const SceneTier &t, int N, int V, int &begin, int &end) {
    begin = 0;
    end = N;
    if (t.order && t.key_start) {
        begin = t.key_start[V - (t.v_hi > V ? V : t.v_hi)];
        end = t.key_start[V - t.v_lo];          // v_lo = -1 -> key_start[V + 1] = N
        begin = begin < 0 ? 0 : (begin > N ? N : begin);
        end = end < begin ? begin : (end > N ? N : end);
    }
}
constexpr int kOrderMaxN = 65536;      // scene_order_kernel is ONE workgroup: N/1024 scenes per lane
constexpr int kOrderMaxV = 1023;
// floats reserved in the scratch buffers for the scene order: N int32 + the V+2 tier offsets key_start[]
// (+ the sorted pedestrian counts behind them: N more)
__host__ __device__ inline int64_t order_floats(int N, int V) { return ((int64_t)2 * N + V + 2 + 3) & ~(int64_t)3; }
// Fills order[0..N) with the scene indices sorted by clamp(num_peds[n], 0, V) descending (stable) on `st`;
// returns false (order untouched) when the batch is outside the kernel's limits or num_peds is null.
// order_peds (optional): order_peds[i] = clamp(num_peds[order[i]], 0, V), the sorted counts themselves.
bool launch_scene_order(const int32_t *num_peds, int N, int V, int32_t *order, int32_t *key_start, hipStream_t st,
                        int32_t *order_peds = nullptr);
// whether launch_scene_order sorts such a batch at all (the training forward leaves its order in the workspace's batch
// tail and the backward picks it up instead of sorting again: both sides decide with this)
inline bool scene_order_applies(const int32_t *num_peds, int N, int V) {
    return num_peds && N >= 2 && N <= kOrderMaxN && V <= kOrderMaxV;
}

// stgcn_agg.hip: ax = x A ([cin][T][V_n]) and cs = colsum(A) ([T][V_n]) of every scene -- the one read of A in a
// step -- written to out + n * out_stride + ax_off / cs_off.
// prep (optional): the launch also prepares the A operands of the backward's exact-bf16 input-gradient GEMMs
// (txp_conv_bf16.hpp) into `wp`, [n_layers][cv::kWpDwords], from the conv weights at params + w_off[l].
struct AggPrep {
    const float *params;
    unsigned *wp;          // input-gradient operands (training: the workspace's batch tail) or null
    unsigned *wp_fwd;      // forward operands (the forward's scratch) or null
    int n_layers;
    int32_t w_off[kMaxTxp + 1];
    // ragged batch: ONE more workgroup of the launch sorts the scenes by crowd size (scene_order.hpp) -- the schedule of
    // the kernels behind it -- instead of a launch of its own; null = no sort here
    int32_t *order, *key_start, *order_peds;
};
// whether the aggregation launch carries the sort: its workgroups have 4 waves (N / 4 scenes per wave), so beyond 1024
// scenes the sorting workgroup outlasts the aggregation itself (measured at 2048: 30 us against 21) and the 16-wave
// scene_order_kernel as a launch of its own is the cheaper way (5 us + a kernel boundary)
inline bool agg_sorts(const int32_t *num_peds, int N, int V) { return scene_order_applies(num_peds, N, V) && N <= 1024; }
int launch_stgcn_agg(int cin, const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *adj,
                     int64_t a_sn, const int32_t *num_peds, int N, int V, float *out, int64_t out_stride, int64_t ax_off,
                     int64_t cs_off, hipStream_t st, const AggPrep *prep = nullptr);

}  // namespace stg
