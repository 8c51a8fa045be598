// txp_x6: the exact-bf16 ("x6": six bf16 products per fp32 product) scene kernels of the wave-per-scene path -- the whole
// model forward (st_gcn block in column mode + TXP-CNN convs) and the backward's loss stage, input-gradient chain and
// block backward, on v_mfma_f32_16x16x32_bf16 with exact three-piece operands (txp_conv_bf16.hpp).
//   one wave per scene   txp_fwd_x6_kernel / txp_bwd_x6_kernel: batches of up to 32 pedestrians per scene
//   teams of waves       txp_fwd_team_kernel / txp_bwd_team_kernel: ONE launch in which a scene-window is worked on by one,
//                        two or four waves (scene_team.hpp) -- batches padded beyond 32 pedestrians (up to kTeamMaxV = 128),
//                        and small batches, whose scenes are cut finer because every scene's latency chain is the step.
// The scene code is ONE template, instantiated for SoloScene (every team quantity a compile-time constant) and TeamScene.
#include "txp_scene_common.hpp"
#include "nll_elem.hpp"
#include "txp_conv_bf16.hpp"

namespace stg {

namespace {

constexpr int C = Cfg::C, P = Cfg::P, T = Cfg::T;

// ------------------------------------------------------------------------------------------
// forward, exact-bf16 variant (x6 = six bf16 products per fp32 product): the convs on v_mfma_f32_16x16x32_bf16
// ------------------------------------------------------------------------------------------
// txp_conv_bf16.hpp.  a_l lives in LDS as three position-major bf16 piece images (7 row slots: no in-place ring is
// needed, because the layer's outputs stay in REGISTERS until every tile has read its inputs); the same registers are
// the residual input of the next layer -- a lane owns the same (position, channel quad) of every tile in every layer
// (<= 10 tiles for V_n <= 32).  The st_gcn block (column mode) hands its outputs over in registers as well.
constexpr int kF6Tiles = 10, kF6Slots = 7;

__host__ __device__ inline int fwd6_region_floats(int v) { return (cv::image_bytes(v, kF6Slots) / 4 + 3) & ~3; }

// CK (scene_team.hpp): SoloScene -- this wave owns the scene, `region` / `ptab` are its own -- or TeamScene: the wave owns
// the column chunk [ck.w0(), ck.w0() + ck.wc()) of a scene that ck.nch() waves share (`region` = the team's image, `ptab` =
// this wave's table of its chunk's positions); `vi` = pedestrians of the scene.
template <bool BF, typename CK>    // BF: bf16 storage of the saved planes / pre-activations (STG_OPT_BF16_STORE)
__device__ __forceinline__ void txp_fwd_scene_x6(const TxpFwdArgs &a, const float *__restrict__ params,
                                                 const float *blk_params, const float *blk_buffers, int n, int vi,
                                                 float *region, ptab_t *ptab, const CK &ck) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63, nq = lane & 15, kq = lane >> 4;
    float *yn = a.y + (int64_t)n * (C * P) * V;
    if (vi < V)                                        // padded pedestrian slots of the output are zeros
        for (int e = lane + 64 * ck.ci(); e < C * P * (V - vi); e += 64 * ck.nch()) {
            const int r = e / (V - vi), w = vi + (e - r * (V - vi));
            yn[(int64_t)r * V + w] = 0.f;
        }
    if (vi == 0) return;
    const int npos = C * ck.wc(), ntiles = (npos + 15) >> 4;      // this wave's positions: (row, its columns)
    const float *Pm = params;
    float *wsn = a.ws ? a.ws + n * a.ws_stride : nullptr;
    float *statn = a.stats ? a.stats + (int64_t)n * L.stat_floats : nullptr;
    unsigned char *img = reinterpret_cast<unsigned char *>(region);
    const cv::LaneGeom lg = cv::lane_geom(vi, kF6Slots);
    constexpr bool bf16 = BF;
    const int SWs = save_sw(vi, bf16), VWs = save_vw(vi, bf16);   // row strides (positions) of the saved arrays

    // ---- st_gcn block (model.py:145-155), column mode: lane = pedestrian; zeroes the image, builds the position table
    {
        // lane = (pedestrian, time half); a lane receives plane channels 4*half .. 4*half+3 of its pedestrian's C rows
        float sv[C * T / 2];
        const float *agn = a.agg + n * a.agg_stride;
        stgcn_block_fwd_cols<true>(a, blk_params, blk_buffers, L.blk[0], n, vi, wsn, statn, agn + a.agg_ax, agn + a.agg_cs,
                                   nullptr, 0, region, cv::image_bytes(vi, kF6Slots) >> 4, ptab, sv, ck);
        // (a team: the block's own ck.sync() has made the whole image zero before anyone writes its interior)
        // v.view(N, T, C, V) (model.py:187): flat f = c*T+t of the block output is plane channel f / C, row f % C.  Per
        // row a pedestrian's eight channels are two record quads (one per lane of the pair); the third quad (channels
        // 8..11) stays zero.
        float *d2 = wsn ? wsn + ws_plane_off(L, V, 0) : nullptr;
        const int pl = lane & 31, pw = ck.w0() + pl, q = lane >> 5;
        if (pl < ck.wc()) {
#pragma unroll
            for (int row = 0; row < C; ++row) {
                const f32x4 v4 = {sv[row], sv[C + row], sv[2 * C + row], sv[3 * C + row]};
                cv::put4(img, (unsigned)(cv::pos_off(vi, 1 + row, pw) + 8 * q), lg.PL, v4);
                if (d2) {
                    store_vec4(d2, (row * SWs + pw + 1) * 3 + q, v4, bf16);
                    if (q == 0) store_vec4(d2, (row * SWs + pw + 1) * 3 + 2, f32x4{0.f, 0.f, 0.f, 0.f}, bf16);
                }
            }
        }
        if (wsn && lane < 2 * C * 3 && ck.lead()) {
            // zero border columns of the saved planes a_0 .. a_L (the weight-gradient GEMM reads them)
            const int b = lane / 3, q = lane - b * 3, pos = (b >> 1) * SWs + ((b & 1) ? vi + 1 : 0);
            for (int l = 0; l <= L.L; ++l)
                store_vec4(wsn + ws_plane_off(L, V, l), pos * 3 + q, f32x4{0.f, 0.f, 0.f, 0.f}, bf16);
        }
    }
    ck.sync();

    // ---- TXP-CNN (model.py:187-195) -------------------------------------------------------------------
    if (STG_SKIP(a, 16)) return;                       // (diagnostic build: time the block alone)
    const unsigned lds_base = (unsigned)(uintptr_t)img;
    f32x4 av[kF6Tiles];
#pragma unroll
    for (int t = 0; t < kF6Tiles; ++t) av[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l <= L.L; ++l) {
        const bool is_out = l == L.L;
        cv::u32x4 w[cv::kWpVecs];
        cv::load_wp(a.wpf + (int64_t)l * cv::kWpDwords, w);
        const float *bias = Pm + (is_out ? L.out_b : L.txp_b[l]);
        f32x4 binit;
#pragma unroll
        for (int r = 0; r < 4; ++r) binit[r] = kq < 3 ? bias[4 * kq + r] : 0.f;
        const float alpha = is_out ? 0.f : Pm[L.prelus + l];
        float *zs = (wsn && !is_out) ? wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V : nullptr;
        float *ps = (wsn && !is_out) ? wsn + ws_plane_off(L, V, l + 1) : nullptr;
        // the layer's operands and bias have landed before the first guarded tile: no vmcnt wait inside the tile loop
        asm volatile("" ::"v"(w[0]), "v"(w[cv::kWpVecs - 1]), "v"(binit), "v"(alpha) : "memory");
        vm_drain();
        // the tile loop -- LDS reads, MFMAs -- wins the issue slots over the SIMD's other wave while that one is in a VALU phase
        // (same-box A/B, three rounds: teams gain -- F 147.9 -> 144.8 us at V = 64, 532 -> 517 at V = 128 x 4096; the solo kernel
        // at V = 32 and the backward's tile loops: no difference)
        __builtin_amdgcn_s_setprio(1);
        unsigned code = cv::tile_code(0, ptab, npos);
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) {
            if (t < ntiles) {
                const cv::Tile tl = cv::tile_from<1>(t, code, npos, lg, vi);
                if (t + 1 < kF6Tiles) code = cv::tile_code(t + 1, ptab, npos);      // (in flight behind this tile's reads)
                cv::BHalf b;
                f32x4 z = binit;
                cv::load_b_half<0>(lds_base, tl, b);
                cv::mma_half<0>(w, b, z);
                cv::load_b_half<1>(lds_base, tl, b);
                cv::mma_half<1>(w, b, z);
                if (tl.ok && kq < 3) {
                    if (is_out) {
                        // v.view(N, C, P, V) (model.py:195): the (P, C, V) conv output IS the (C, P, V) tensor
#pragma unroll
                        for (int r = 0; r < 4; ++r) yn[(int64_t)((4 * kq + r) * C + tl.h) * V + tl.w] = z[r];
                    } else {
                        f32x4 v4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v4[r] = (z[r] > 0.f ? z[r] : alpha * z[r]) + av[t][r];   // (av = 0 at l = 0)
                        av[t] = v4;
                        if (zs) {
                            store_vec4(zs, (tl.h * VWs + tl.w) * 3 + kq, z, bf16);
                            store_vec4(ps, (tl.h * SWs + tl.w + 1) * 3 + kq, v4, bf16);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        if (is_out) break;
        // every tile (of every wave of a team) has read a_l: a_{l+1} replaces it in the image
        ck.sync();
        unsigned codes[kF6Tiles];                      // (all ten table reads in flight: the weight registers are dead here)
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) codes[t] = cv::tile_code(t, ptab, npos);
#pragma unroll
        for (int t = 0; t < kF6Tiles; ++t) {
            const int p = 16 * t + nq;
            if (p < npos && kq < 3) {
                const unsigned hw = codes[t];
                cv::put4(img, (unsigned)(cv::pos_off(vi, 1 + (int)(hw >> 8), (int)(hw & 0xffu)) + 8 * kq), lg.PL, av[t]);
            }
        }
        ck.sync();
    }
}

template <int WPB, bool BF>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_x6_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int per_wave = fwd6_region_floats(Vl) + ptab_floats(Vl);
    float *region = sm + wave * per_wave;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(region + fwd6_region_floats(Vl));
    float *blk_p = sm + WPB * per_wave, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, WPB * 64);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        // (clamped: a workspace tail that does not hold THIS batch's order -- a caller's contract breach -- must not turn
        // into an out-of-bounds scene index)
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? scene_index(a.tier.order[begin + it], a.N) : it);
        int vi = a.num_peds ? a.num_peds[n] : a.V;
        vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > a.V ? a.V : vi));
        txp_fwd_scene_x6<BF>(a, params, blk_p, blk_b, n, vi, region, ptab, SoloScene{vi});
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// backward, exact-bf16 variant (x6): the input-gradient GEMMs on v_mfma_f32_16x16x32_bf16 with three-piece operands
// ------------------------------------------------------------------------------------------
// txp_conv_bf16.hpp: dz_l lives in LDS as three position-major bf16 piece images (x = x_h + x_m + x_l exactly), the six
// products that reach 2^-24 are accumulated in fp32 -- the same accuracy class as the fp32 MFMA, but 24 MFMAs of 16
// matrix-pipe cycles per tile that run BESIDE the VALU instead of 27 fp32 MFMAs that hold the SIMD's VALU port for 32
// cycles each (tools/micro/mfma_valu_overlap.hip).  The running input gradient d(a_l) stays in REGISTERS: a lane owns
// the same (position, channel quad) of every tile in every layer (<= 10 tiles for V_n <= 32), which is also the quad
// structure of the saved z_l / dz_l arrays -- no LDS copy of it, no position table in the dz construction.
constexpr int kX6Tiles = 10;                          // 16-position tiles of a scene of <= 32 pedestrians
constexpr int kX6Slots = 7;                           // row slots of the dz image: borders + C interior rows
__host__ __device__ inline int bwd6_region_floats(int v) {
    const int img = cv::image_bytes(v, kX6Slots) / 4, tail = (2 * C * (T + 2) + C * T) * v;    // (the block tail's arrays)
    return ((img > tail ? img : tail) + 3) & ~3;
}

// CK (scene_team.hpp): SoloScene -- this wave owns the scene -- or TeamScene: the wave owns the column chunk [ck.w0(),
// ck.w0() + ck.wc()) of a scene shared by ck.nch() waves (`region` = the team's image, `ptab` = this wave's table of its
// chunk's positions); `vi` = pedestrians of the scene.  Per-scene sums are exchanged through LDS (ck.sum), the team's
// leading wave writes the scene's loss and its row of small-parameter gradients.
template <bool BF, typename CK>    // BF: bf16 storage of z_l (read) and dz_l (written)
__device__ __forceinline__ void txp_bwd_scene_x6(const TxpBwdArgs &a, const float *blk_params, int n, int vi, float *region,
                                                 ptab_t *ptab, const CK &ck) {
    const ModelLayout &L = a.lay;
    const int V = a.V, lane = threadIdx.x & 63, nq = lane & 15, kq = lane >> 4;
    float *slope_row = a.rows + (int64_t)n * (L.n_blk_params + L.n_txp) + L.n_blk_params;
    if (vi == 0) {                                     // empty scene: its row of small-parameter gradients is zero
        for (int e = lane; e < L.n_blk_params + L.n_txp; e += 64) slope_row[e - L.n_blk_params] = 0.f;
        if (a.nll_target && lane == 0) a.nll_losses[n] = 0.f;
        return;
    }
    const int npos = C * ck.wc(), ntiles = (npos + 15) >> 4;      // this wave's positions: (row, its columns)
    const float *Pm = a.params;
    const float *wsn = a.ws + n * a.ws_stride;
    const float *dyn = a.dy + (int64_t)n * (C * P) * V;
    // sums that only leave the kernel (PReLU slope gradients, the loss, the block's parameter gradients): a solo wave writes
    // them to the scene's row; the waves of a team park theirs in LDS rows (zeroed here) that the leading wave adds at the end
    float *prow = ck.row(slope_row - L.n_blk_params);
    if constexpr (CK::kTeam) {
        for (int e = lane; e < kTeamRow; e += 64) prow[e] = 0.f;
    }
    unsigned char *img = reinterpret_cast<unsigned char *>(region);
    const cv::LaneGeom lg = cv::lane_geom(vi, kX6Slots);
    constexpr bool bf16 = BF;
    const int VWs = save_vw(vi, bf16);                             // row stride (positions) of the saved z_l / dz_l
    // From ONE read of the position table: the vector index of the lane's quad of tile t in those arrays (fp32: rows of vi
    // positions, i.e. (16 t + n) * 3 + kq) and the byte offset of its record quad in the image (interior row 0 = slot 1;
    // -1 past the scene's last position)
    auto tile_slots = [&](int t, int &rec, int &quad) {
        const int p = 16 * t + nq;
        const unsigned hw = ptab[p < npos ? p : 0];
        const int h = (int)(hw >> 8), w = (int)(hw & 0xffu);
        quad = (h * VWs + w) * 3 + kq;
        rec = (p < npos && kq < 3) ? cv::pos_off(vi, 1 + h, w) + 8 * kq : -1;
    };
    {
        uint4 *z4 = reinterpret_cast<uint4 *>(img);
        for (int e = lane + 64 * ck.ci(); e < cv::image_bytes(vi, kX6Slots) >> 4; e += 64 * ck.nch()) z4[e] = make_uint4(0u, 0u, 0u, 0u);
    }
    build_ptab(ptab, ck.w0(), ck.wc());
    ck.sync();
    f32x4 dcur[kX6Tiles];
#pragma unroll
    for (int t = 0; t < kX6Tiles; ++t) dcur[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int l = L.L; l >= 0; --l) {
        // ---- dz_l -> the piece images (and position-major fp32 to HBM for the weight-gradient GEMM) -----------------
        float *dzo = a.dzg + ((int64_t)n * (L.L + 1) + l) * dz_slot(V);
        if (STG_SKIP(a, 512) || (l == L.L && STG_SKIP(a, 2048)) || (l != L.L && STG_SKIP(a, 4096))) {
        } else if (l == L.L) {
            // dz of the output conv is dV_pred: row rc = f * P + p of the (C*P) x V array is channel rc / C, plane row
            // rc % C.  Lanes are laid over (row or prediction step, pedestrian) with the row length rounded up to a power
            // of two.
            const int wcw = ck.wc();
            const int vp = wcw <= 1 ? 1 : (wcw <= 2 ? 2 : (wcw <= 4 ? 4 : (wcw <= 8 ? 8 : (wcw <= 16 ? 16 : 32))));
            const int sh = __builtin_ctz(vp), rpi = 64 >> sh;
            const int sub = lane >> sh, wl = lane & (vp - 1), w = ck.w0() + wl;      // the lane's pedestrian
            const bool okw = wl < wcw;
            // the values pass through an fp32 staging array S [C*P rows][vi] laid over the (still empty) m / l piece
            // images: coalesced along the pedestrians here, read back as record quads below
            float *S = reinterpret_cast<float *>(img + lg.PL);
            auto put1 = [&](int rc, float g) { S[rc * vi + w] = g; };
            if (a.nll_target) {
                const float *tn = a.nll_target + (int64_t)n * P * V * 2;
                const float inv_cnt = 1.0f / (float)(P * vi);
                const float gs = inv_cnt * (a.nll_weights ? a.nll_weights[n] : 1.f);
                float lacc = 0.f;
                for (int p0 = 0; p0 < P; p0 += rpi) {
                    const int p = p0 + sub;
                    if (okw && p < P) {
                        const float *q = dyn + (int64_t)p * V + w;
                        const float2 tg = *reinterpret_cast<const float2 *>(tn + ((int64_t)p * V + w) * 2);
                        float g[5];
#ifdef STG_DIAG
                        if (STG_SKIP(a, 1 << 21))       // (diagnostic builds: time the hardware-transcendental form)
                            lacc += nll_elem_t<true>(q[0], q[(int64_t)P * V], q[(int64_t)2 * P * V], q[(int64_t)3 * P * V],
                                                     q[(int64_t)4 * P * V], tg.x, tg.y, true, g);
                        else
#endif
                        lacc += nll_elem(q[0], q[(int64_t)P * V], q[(int64_t)2 * P * V], q[(int64_t)3 * P * V],
                                         q[(int64_t)4 * P * V], tg.x, tg.y, true, g);
#pragma unroll
                        for (int f = 0; f < C; ++f) put1(f * P + p, g[f] * gs);
                    }
                }
                float ls[1] = {lacc};
                ck.template reduce<1>(ls);
                if (ck.writer()) {
                    if constexpr (CK::kTeam) prow[kTeamRowLoss] = ls[0];
                    else a.nll_losses[n] = ls[0] * inv_cnt;
                }
            } else {
                constexpr int U = 4;
                for (int r0 = 0; r0 < C * P; r0 += rpi * U) {
                    float dv[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        dv[u] = (okw && row < C * P) ? dyn[(int64_t)row * V + w] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int row = r0 + u * rpi + sub;
                        if (okw && row < C * P) put1(row, dv[u]);
                    }
                }
            }
            // quads: the lane's (position, channel quad) of every tile from S (plane (ch, row) = row ch*C + row of the
            // array), held in registers while S is wiped (the m / l images must be zero outside the interior), then
            // split into the three images and stored position-major for the weight-gradient GEMM (16 bytes per lane: a
            // scattered 4-byte store and three 2-byte LDS stores per value cost 20 us)
            // (the registers of the running input gradient are free here: the output conv's chain starts from zero)
            __builtin_amdgcn_wave_barrier();
            vm_drain();                                 // (V_pred / target are consumed: nothing pending past here)
            f32x4 (&qd)[kX6Tiles] = dcur;
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                const int p = 16 * t + nq;
                const unsigned hw = ptab[p < npos ? p : 0];
                const float *sq = S + ((4 * (kq < 3 ? kq : 0)) * C + (int)(hw >> 8)) * vi + (int)(hw & 0xffu);
                qd[t] = f32x4{sq[0], sq[C * vi], sq[2 * C * vi], sq[3 * C * vi]};
            }
            ck.sync();                                  // (every wave of a team has its quads: S may be wiped)
            {
                uint4 *z4 = reinterpret_cast<uint4 *>(img + lg.PL);
                for (int e = lane + 64 * ck.ci(); e < (2 * lg.PL + 128) >> 4; e += 64 * ck.nch()) z4[e] = make_uint4(0u, 0u, 0u, 0u);
            }
            ck.sync();
            int rec[kX6Tiles], qv[kX6Tiles];           // (the table reads of all ten tiles in flight together)
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) tile_slots(t, rec[t], qv[t]);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (rec[t] >= 0) {
                    cv::put4(img, (unsigned)rec[t], lg.PL, qd[t]);
                    store_vec4(dzo, qv[t], qd[t], bf16);
                }
            }
        } else {
            // dz_l = d(a_{l+1}) * prelu'(z_l): z_l and dz_l are position-major [pos][12] in HBM, the lane's quad of tile t
            // is vector (16 t + n) * 3 + kq
            const float *zl = wsn + L.ws_hdr_floats + (int64_t)L.ws_z[l] * V;
            const float alpha = Pm[L.prelus + l];
            float slope_acc = 0.f;
            // all ten quads of z_l in flight at once (the weight registers are dead here): one HBM latency per layer.
            // (Requesting them BEFORE the previous layer's MFMAs needs 40 more live registers there and spilled; a
            // never-awaited "touch" load into a dead register is not an option either -- the register is reused
            // while the load is in flight and the late write-back corrupts its new owner.)
            f32x4 zv[kX6Tiles];
            int rec[kX6Tiles], qv[kX6Tiles];
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) tile_slots(t, rec[t], qv[t]);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t)
                zv[t] = rec[t] >= 0 ? load_vec4_raw(zl, qv[t], bf16) : f32x4{1.f, 1.f, 1.f, 1.f};
            // all ten have landed before the first guarded tile: the tiles' dz stores are not waited for (vm_drain)
            asm volatile("" ::"v"(zv[0]), "v"(zv[1]), "v"(zv[2]), "v"(zv[3]), "v"(zv[4]), "v"(zv[5]), "v"(zv[6]), "v"(zv[7]),
                         "v"(zv[8]), "v"(zv[9]), "v"(alpha)
                         : "memory");
            vm_drain();
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (rec[t] >= 0) {
                    f32x4 dzv;
                    const f32x4 zq = finish_vec4(zv[t], bf16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = zq[r], d = dcur[t][r];
                        float dz = d;
                        if (!(z > 0.f)) {
                            dz = alpha * d;
                            slope_acc = fmaf(d, z, slope_acc);
                        }
                        dzv[r] = dz;
                    }
                    cv::put4(img, (unsigned)rec[t], lg.PL, dzv);
                    store_vec4(dzo, qv[t], dzv, bf16);
                }
            }
            float ss[1] = {slope_acc};
            ck.template reduce<1>(ss);
            if (ck.writer()) prow[L.n_blk_params + l] = ss[0];
        }
        ck.sync();
        // ---- d(a_l) = conv_transpose(dz_l, W_l) [+ d(a_{l+1}) through the residual of the hidden layers] ----------------
        if (!STG_SKIP(a, 1024)) {
            cv::u32x4 w[cv::kWpVecs];
            cv::load_wp(a.wp + (int64_t)l * cv::kWpDwords, w);
            asm volatile("" ::"v"(w[0]), "v"(w[cv::kWpVecs - 1]) : "memory");
            vm_drain();                                 // (no per-operand waits in front of every guarded tile)
            const unsigned lds_base = (unsigned)(uintptr_t)img;
            const bool keep = l != L.L && l != 0;       // d(a_l) += d(a_{l+1}) (a_{l+1} = prelu(z_l) + a_l for 1 <= l < L)
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            unsigned code = cv::tile_code(0, ptab, npos);
#pragma unroll
            for (int t = 0; t < kX6Tiles; ++t) {
                if (t < ntiles) {
                    const cv::Tile tl = cv::tile_from<1>(t, code, npos, lg, vi);
                    if (t + 1 < kX6Tiles) code = cv::tile_code(t + 1, ptab, npos);  // (in flight behind this tile's reads)
                    // two half-tiles through ONE 32-register operand set (96 weight + 40 gradient registers are live);
                    // each half's reads and their wait are one asm statement, the SIMD's other wave covers the latency
                    cv::BHalf b;
                    f32x4 acc = keep ? dcur[t] : zero;
                    cv::load_b_half<0>(lds_base, tl, b);
                    cv::mma_half<0>(w, b, acc);
                    cv::load_b_half<1>(lds_base, tl, b);
                    cv::mma_half<1>(w, b, acc);
                    dcur[t] = acc;
                }
            }
        }
        ck.sync();
    }
    // ---- d(a_0) (channels 0..T-1) -> D [C][T][vi] at the start of the region: v.view(N, T, C, V) (model.py:187)
    // backwards, plane (ch, row) is flat f = ch*C + row = c*T + t of the block output.  The dz image is dead.
    float *D = region;
#pragma unroll
    for (int t = 0; t < kX6Tiles; ++t) {
        const int p = 16 * t + nq;
        if (p < npos && kq < T / 4) {
            const unsigned hw = ptab[p];
            const int h = (int)(hw >> 8), ww = (int)(hw & 0xffu);
#pragma unroll
            for (int r = 0; r < 4; ++r) D[((4 * kq + r) * C + h) * vi + ww] = dcur[t][r];
        }
    }
    __builtin_amdgcn_wave_barrier();                  // (a lane reads back its own wave's columns of D)
    // ---- the st_gcn block (model.py:145-155 backwards), column mode: lane = (pedestrian, time half) ------------------------
    if constexpr (!CK::kTeam) {
        for (int e = L.L + lane; e < L.n_txp; e += 64) slope_row[e] = 0.f;       // dead slopes (layers >= L)
    }
    if (!STG_SKIP(a, 4))
        stgcn_block_bwd_cols<true>(a, blk_params, L.blk[0], n, vi, D, slope_row - L.n_blk_params, a.ws + n * a.ws_stride, ck);
    if constexpr (CK::kTeam) {
        // the team's parked rows -> the scene's row of small-parameter gradients and its loss, added in chunk order
        ck.sync();
        if (ck.lead()) {
            float *row = slope_row - L.n_blk_params;
            const int nrow = L.n_blk_params + L.n_txp;
            for (int e0 = 0; e0 < kTeamRow; e0 += 64) {
                const int e = e0 + lane < kTeamRow ? e0 + lane : kTeamRow - 2;      // (a spare place, never stored)
                float t = ck.row_of(0)[e];
                for (int c = 1; c < ck.nch(); ++c) t += ck.row_of(c)[e];
                if (e < nrow) row[e] = t;
                if (e == kTeamRowLoss && a.nll_target) a.nll_losses[n] = t * (1.0f / (float)(P * vi));
            }
        }
    }
}

template <int WPB, bool BF>
__global__ __launch_bounds__(WPB * 64, WPB == 8 ? 1 : 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_bwd_x6_kernel(
    const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Vl = a.Vl, wave = threadIdx.x >> 6;
    const int per_wave = bwd6_region_floats(Vl) + bwd_ptab_floats(Vl);
    float *region = sm + wave * per_wave;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(region + bwd6_region_floats(Vl));
    float *blk_p = sm + WPB * per_wave;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, WPB * 64);
    const int gw = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + wave), nw = gridDim.x * WPB;
    int begin, end;
    tier_range(a.tier, a.N, a.V, begin, end);
    const int M = end - begin;
    for (int r = 0; r * nw < M; ++r) {
        const int it = walk_item(r, gw, nw, M, a.tier.order != nullptr && a.tier.serpentine);
        if (it < 0) continue;
        // (clamped: a workspace tail that does not hold THIS batch's order -- a caller's contract breach -- must not turn
        // into an out-of-bounds scene index)
        const int n = __builtin_amdgcn_readfirstlane(a.tier.order ? scene_index(a.tier.order[begin + it], a.N) : it);
        int vi = a.num_peds ? a.num_peds[n] : a.V;
        vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > a.V ? a.V : vi));
        txp_bwd_scene_x6<BF>(a, blk_p, n, vi, region, ptab, SoloScene{vi});
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------
// team launch: scenes of 1 .. kTeamMaxV pedestrians in ONE launch, one / two / four waves per scene (scene_team.hpp)
// ------------------------------------------------------------------------------------------
// Work units of a launch, in list order: one per four-wave scene, one per PAIR of two-wave scenes, one per four solo scenes
// (the sorted list is descending, so the units come heaviest first).  A workgroup takes the units of a fixed walk; all four
// waves of a workgroup are always in the same kind of unit, so the workgroup barriers inside the scene code match.
struct TeamCount {
    int n4, n2, n1;        // scenes per class (sorted list: [0, n4) | [n4, n4 + n2) | the rest)
    int u4, u2, units;
    bool sorted;
};
__device__ __forceinline__ TeamCount team_count(const SceneTier &t, const TeamGeom &g, const int32_t *num_peds, int N, int V) {
    TeamCount c;
    c.sorted = t.order && t.key_start;
    if (c.sorted) {
        // key_start[k] = number of scenes with more than V - k pedestrians
        const int a4 = g.v2 < V ? t.key_start[V - g.v2] : 0, a2 = g.v1 < V ? t.key_start[V - g.v1] : 0;
        c.n4 = a4;
        c.n2 = a2 - a4;
        c.u2 = (c.n2 + 1) >> 1;
    } else {
        // no sorted list: every scene in the class of the padded V.  With num_peds (a single scene, a batch beyond the sort's
        // limits) no pairs of two-wave scenes -- an EMPTY scene may only leave the barrier sequence together with its whole
        // workgroup
        const bool pairs = !num_peds && V > g.v1 && V <= g.v2;
        c.n4 = (V > g.v1 && !pairs) ? N : 0;
        c.n2 = pairs ? N : 0;
        c.u2 = (c.n2 + 1) >> 1;
    }
    c.n1 = N - c.n4 - c.n2;
    c.u4 = c.n4;
    c.units = c.u4 + c.u2 + ((c.n1 + 3) >> 2);
    return c;
}
struct TeamUnit {
    int n, vi;             // scene (n < 0: this wave idles this round) and its pedestrians
    int nch, ci, slot;     // waves on the scene, this wave's place among them, which of the round's 4 / nch scenes
    int w0, wc;            // this wave's column chunk
};
__device__ __forceinline__ TeamUnit team_unit(const SceneTier &t, const TeamCount &c, const int32_t *__restrict__ num_peds,
                                              int N, int V, int u) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    TeamUnit q;
    int idx;
    if (u < c.u4) {
        q.nch = 4; q.slot = 0; q.ci = wave; idx = u;
    } else if (u < c.u4 + c.u2) {
        const int i0 = c.n4 + 2 * (u - c.u4);
        if (i0 + 1 < c.n4 + c.n2) { q.nch = 2; q.slot = wave >> 1; q.ci = wave & 1; idx = i0 + q.slot; }
        else { q.nch = 4; q.slot = 0; q.ci = wave; idx = i0; }           // the odd one out takes the whole workgroup
    } else {
        q.nch = 1; q.slot = wave; q.ci = 0; idx = c.n4 + c.n2 + 4 * (u - c.u4 - c.u2) + wave;
    }
    q.n = idx < N ? (c.sorted ? scene_index(t.order[idx], N) : idx) : -1;
    q.n = __builtin_amdgcn_readfirstlane(q.n);
    int vi = q.n >= 0 ? (num_peds ? num_peds[q.n] : V) : 0;
    vi = __builtin_amdgcn_readfirstlane(vi < 0 ? 0 : (vi > V ? V : vi));
    q.vi = vi;
    const int wc = (vi + q.nch - 1) / q.nch;          // equal chunks; the last one may be short (or empty)
    q.w0 = q.ci * wc;
    q.wc = vi - q.w0 < wc ? (vi - q.w0 > 0 ? vi - q.w0 : 0) : wc;
    return q;
}
// equal shares: the fewest rounds the grid can do, then just enough workgroups for them
__device__ __forceinline__ int team_workers(int units, int grid) {
    const int rounds = (units + grid - 1) / grid;
    return rounds > 0 ? (units + rounds - 1) / rounds : 1;
}
__host__ __device__ inline int team_ptab_floats() { return ptab_floats(32); }

template <bool BF>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_fwd_team_kernel(
    const TxpFwdArgs a, const float *__restrict__ params, const float *__restrict__ buffers) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = a.team.region_floats;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(sm + R + wave * team_ptab_floats());
    float *xr = sm + R + 4 * team_ptab_floats();
    float *blk_p = xr + 2 * kXrFwd, *blk_b = blk_p + ((a.lay.n_blk_params + 3) & ~3);
    stage_block_params(a.lay, params, buffers, blk_p, blk_b, 256);
    const TeamCount tc = team_count(a.tier, a.team, a.num_peds, a.N, a.V);
    const int G = team_workers(tc.units, gridDim.x);
    int prev = 1;
    for (int r = 0; r * G < tc.units; ++r) {
        const int u = walk_item(r, blockIdx.x, G, tc.units, a.tier.serpentine != 0);
        if (u < 0 || (int)blockIdx.x >= G) continue;
        const TeamUnit q = team_unit(a.tier, tc, a.num_peds, a.N, a.V, u);
        if (q.nch > 1 || prev > 1) team_barrier();     // (the previous round is over before a region changes hands)
        prev = q.nch;
        if (q.n >= 0) {
            const TeamScene ck{q.vi, q.w0, q.wc, q.nch, q.ci, xr + (q.slot & 1) * kXrFwd, nullptr, q.nch > 1 && !STG_SKIP(a, 1 << 20)};
            float *region = sm + q.slot * ((R >> 2) * q.nch);
            txp_fwd_scene_x6<BF>(a, params, blk_p, blk_b, q.n, q.vi, region, ptab, ck);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <bool BF>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void txp_bwd_team_kernel(const TxpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int R = a.team.region_floats;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(sm + R + wave * team_ptab_floats());
    float *xr = sm + R + 4 * team_ptab_floats();
    float *blk_p = xr + kXrBwdWg;
    stage_block_params(a.lay, a.params, nullptr, blk_p, nullptr, 256);
    const TeamCount tc = team_count(a.tier, a.team, a.num_peds, a.N, a.V);
    const int G = team_workers(tc.units, gridDim.x);
    int prev = 1;
    for (int r = 0; r * G < tc.units; ++r) {
        const int u = walk_item(r, blockIdx.x, G, tc.units, a.tier.serpentine != 0);
        if (u < 0 || (int)blockIdx.x >= G) continue;
        const TeamUnit q = team_unit(a.tier, tc, a.num_peds, a.N, a.V, u);
        if (q.nch > 1 || prev > 1) team_barrier();
        prev = q.nch;
        if (q.n >= 0) {
            // (a team of nch waves is waves slot * nch .. slot * nch + nch - 1 of the workgroup: their parked rows are adjacent)
            const TeamScene ck{q.vi, q.w0, q.wc, q.nch, q.ci, xr + (q.slot & 1) * kXrRows,
                               xr + 2 * kXrRows + q.slot * q.nch * kTeamRow, q.nch > 1 && !STG_SKIP(a, 1 << 20)};
            float *region = sm + q.slot * ((R >> 2) * q.nch);
            txp_bwd_scene_x6<BF>(a, blk_p, q.n, q.vi, region, ptab, ck);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
bool txp_fwd_x6_fits(const ModelLayout &L, int V) {
    return L.n_txp > 0 && V <= kTeamMaxV && !(L.flags & STG_OPT_F32_MFMA) && L.n_blocks == 1 &&
           L.blk[0].cin == Cfg::CIN0 && !diag_env("STG_FWD_F32", 0);
}

// Team launch geometry.  Class bounds: a scene of up to v1 pedestrians belongs to one wave, up to v2 to two, beyond to four
// (a wave's chunk is at most 32 columns = 10 tiles).  Large batches fill the chip with whole scenes (32 / 64); a small
// batch is latency-bound -- every scene's dependency chain IS the step -- so its scenes are cut finer.
// A small batch is latency-bound -- with fewer scene-waves than the chip has SIMDs every scene's dependency chain IS the
// step -- so its scenes are cut finer (measured, synthetic V = 32 / eth-train histogram, M scene-windows/s: N = 256
// 1.88 -> 2.17 at (8, 16); N = 512 3.27 -> 3.71, eth/train x 512 3.07 -> 3.57 at (16, 32); N = 1024 5.52 -> 5.74 / 5.57 ->
// 6.33 at (16, 32); profiles/r03_team_bounds.log).  `uniform` (no num_peds): every scene has V pedestrians, the number of
// scene-waves is known -- never cut so fine that they no longer fit the chip's 2048 wave slots at once.
constexpr int kTeamFineBatch = 1536, kTeamFinestBatch = 384;
static int team_waves(int v, int v1, int v2) { return v <= v1 ? 1 : (v <= v2 ? 2 : 4); }
static bool team_geom(int N, int V, bool uniform, TeamGeom *g) {
    g->on = 0;
    if (V > kTeamMaxV) return false;
    int v1 = 32, v2 = 64;
    if (N < kTeamFinestBatch) { v1 = 8; v2 = 16; }
    else if (N < kTeamFineBatch) { v1 = 16; v2 = 32; }
    if (uniform)
        while (v1 < 32 && (int64_t)N * team_waves(V, v1, v2) > 2048) { v1 *= 2; v2 *= 2; }
    if (const int e = diag_env("STG_TEAM_V1", 0)) v1 = e;
    if (const int e = diag_env("STG_TEAM_V2", 0)) v2 = e;
    if (v1 > 32) v1 = 32;
    if (v2 > 64) v2 = 64;
    if (v2 < v1) v2 = v1;
    // (the backward's column-mode block needs D = [C][T][v] of the dead image, not the LDS arrays of bwd6_region_floats)
    auto region = [&](int v) { return fwd6_region_floats(v); };
    int r = 4 * region(v1 < V ? v1 : V);
    if (2 * region(v2 < V ? v2 : V) > r) r = 2 * region(v2 < V ? v2 : V);
    if (region(V) > r) r = region(V);
    g->on = 1; g->v1 = v1; g->v2 = v2; g->region_floats = (r + 15) & ~15;
    return true;
}
// beyond 32 pedestrians always; up to 32 when the batch is small enough for finer teams to pay
static bool team_wanted(int N, int V, bool uniform, TeamGeom *g) {
    if (!team_geom(N, V, uniform, g)) return false;
    return V > 16 * kF6Tiles / C || g->v1 < V || diag_env("STG_TEAM", 0) != 0;
}
static int team_grid(size_t lds_bytes, int N) {
    int per_cu = (int)(kLdsBytes / lds_bytes);
    if (per_cu > 2) per_cu = 2;                        // 256-register kernels: two waves per SIMD
    if (per_cu < 1) per_cu = 1;
    const int g = kNumCU * per_cu;
    return g < N ? g : N;                              // (at most one unit per scene)
}

int launch_txp_fwd_x6(const TxpFwdArgs &a0, hipStream_t st) {
    TxpFwdArgs a = a0;
    if (a.wpf && txp_fwd_x6_fits(a.lay, a.V) && team_wanted(a.N, a.V, a.num_peds == nullptr, &a.team)) {
        const size_t lds = ((size_t)a.team.region_floats + 4 * team_ptab_floats() + 2 * kXrFwd + wave_param_floats(a.lay)) * sizeof(float);
        STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "txp_fwd_team: V=%d needs %zu bytes of LDS", a.V, lds);
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
        const void *fn = bf ? reinterpret_cast<const void *>(&txp_fwd_team_kernel<true>)
                            : reinterpret_cast<const void *>(&txp_fwd_team_kernel<false>);
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_team: hipFuncSetAttribute");
        const dim3 grid(team_grid(lds, a.N));
        if (bf) hipLaunchKernelGGL(txp_fwd_team_kernel<true>, grid, dim3(256), lds, st, a, a.params, a.buffers);
        else hipLaunchKernelGGL(txp_fwd_team_kernel<false>, grid, dim3(256), lds, st, a, a.params, a.buffers);
        STG_LAUNCH_CHECK("txp_fwd_team");
        return STG_OK;
    }
    if (a.wpf && txp_fwd_x6_fits(a.lay, a.V) && a.V <= 16 * kF6Tiles / C) {
        const size_t per_wave = (size_t)(fwd6_region_floats(a.Vl) + ptab_floats(a.Vl)) * sizeof(float);
        const int wpb = wave_wpb(per_wave) == 8 ? 8 : 4;     // (the 18 KB images of V <= 32 always fit four waves)
        const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
        const dim3 grid(wave_grid(lds, wpb, a.N));
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
#define STG_LX(W, B)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_fwd_x6_kernel<W, B>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_fwd_x6: hipFuncSetAttribute");                         \
        hipLaunchKernelGGL((txp_fwd_x6_kernel<W, B>), grid, dim3(W * 64), lds, st, a, a.params, a.buffers);   \
    } while (0)
        if (wpb == 8) { if (bf) STG_LX(8, true); else STG_LX(8, false); }
        else { if (bf) STG_LX(4, true); else STG_LX(4, false); }
#undef STG_LX
        STG_LAUNCH_CHECK("txp_fwd_x6");
        return STG_OK;
    }
    return fail(STG_EUNSUPPORTED, "txp_fwd_x6: V=%d outside the exact-bf16 kernels", a.V);
}

bool txp_bwd_x6_fits(const ModelLayout &L, int V) {
    return L.n_txp > 0 && V <= kTeamMaxV && L.n_blocks == 1 && L.blk[0].cin == Cfg::CIN0 &&
           !(L.flags & (STG_OPT_SPLIT_BF16 | STG_OPT_F32_MFMA)) && !diag_env("STG_BWD_F32", 0);
}
int64_t txp_bwd_x6_wp_floats(const ModelLayout &L) { return (int64_t)(L.L + 1) * cv::kWpDwords; }
int launch_txp_bwd_x6(const TxpBwdArgs &a0, hipStream_t st) {
    TxpBwdArgs a = a0;
    if (a.wp && txp_bwd_x6_fits(a.lay, a.V) && team_wanted(a.N, a.V, a.num_peds == nullptr, &a.team)) {
        const size_t lds = ((size_t)a.team.region_floats + 4 * team_ptab_floats() + kXrBwdWg + wave_param_floats(a.lay)) * sizeof(float);
        STG_REQUIRE(lds <= (size_t)kLdsBytes, STG_ELDS, "txp_bwd_team: V=%d needs %zu bytes of LDS", a.V, lds);
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
        const void *fn = bf ? reinterpret_cast<const void *>(&txp_bwd_team_kernel<true>)
                            : reinterpret_cast<const void *>(&txp_bwd_team_kernel<false>);
        hipError_t e_ = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_team: hipFuncSetAttribute");
        const dim3 grid(team_grid(lds, a.N));
        if (bf) hipLaunchKernelGGL(txp_bwd_team_kernel<true>, grid, dim3(256), lds, st, a);
        else hipLaunchKernelGGL(txp_bwd_team_kernel<false>, grid, dim3(256), lds, st, a);
        STG_LAUNCH_CHECK("txp_bwd_team");
        return STG_OK;
    }
    if (a.wp && txp_bwd_x6_fits(a.lay, a.V) && a.V <= 16 * kX6Tiles / C) {
        const size_t per_wave = (size_t)(bwd6_region_floats(a.Vl) + bwd_ptab_floats(a.Vl)) * sizeof(float);
        const int wpb = wave_wpb(per_wave) == 8 ? 8 : 4;
        const size_t lds = per_wave * wpb + wave_param_floats(a.lay) * sizeof(float);
        const dim3 grid(wave_grid(lds, wpb, a.N));
        const bool bf = (a.lay.flags & STG_OPT_BF16_STORE) != 0;
#define STG_LX(W, B)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&txp_bwd_x6_kernel<W, B>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e_ != hipSuccess) return hip_fail(e_, "txp_bwd_x6: hipFuncSetAttribute");                         \
        hipLaunchKernelGGL((txp_bwd_x6_kernel<W, B>), grid, dim3(W * 64), lds, st, a);                        \
    } while (0)
        if (wpb == 8) { if (bf) STG_LX(8, true); else STG_LX(8, false); }
        else { if (bf) STG_LX(4, true); else STG_LX(4, false); }
#undef STG_LX
        STG_LAUNCH_CHECK("txp_bwd_x6");
        return STG_OK;
    }
    return fail(STG_EUNSUPPORTED, "txp_bwd_x6: V=%d outside the exact-bf16 kernels", a.V);
}

}  // namespace stg
