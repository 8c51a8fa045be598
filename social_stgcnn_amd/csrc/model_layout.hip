// Host-side layout arithmetic for the fused model kernels + the public size queries.
#include "model_common.hpp"
#include "scene_order.hpp"

namespace stg {

int make_layout(const stg_model_desc *d, ModelLayout *lay) {
    STG_REQUIRE(d && lay, STG_EINVAL, "model descriptor is null");
    STG_REQUIRE(d->n_stgcnn >= 1 && d->n_stgcnn <= STG_MAX_BLOCKS, STG_EUNSUPPORTED,
                "n_stgcnn=%d outside 1..%d", d->n_stgcnn, STG_MAX_BLOCKS);
    STG_REQUIRE(d->n_txpcnn >= 0 && d->n_txpcnn <= kMaxTxp, STG_EUNSUPPORTED, "n_txpcnn=%d outside 0..%d",
                d->n_txpcnn, kMaxTxp);
    STG_REQUIRE(d->c_out == Cfg::C && d->t_obs == Cfg::T && d->kt == Cfg::KT, STG_EUNSUPPORTED,
                "fused kernels are built for output_feat=%d seq_len=%d kernel_size=%d (got %d,%d,%d)", Cfg::C,
                Cfg::T, Cfg::KT, d->c_out, d->t_obs, d->kt);
    STG_REQUIRE(d->c_in == Cfg::CIN0 || d->c_in == Cfg::C, STG_EUNSUPPORTED,
                "fused kernels are built for input_feat %d or %d (got %d)", Cfg::CIN0, Cfg::C, d->c_in);
    STG_REQUIRE(d->n_txpcnn == 0 || d->t_pred == Cfg::P, STG_EUNSUPPORTED,
                "fused kernels are built for pred_seq_len=%d (got %d)", Cfg::P, d->t_pred);
    STG_REQUIRE(d->bn_mode == 0 || d->bn_mode == 1, STG_EINVAL, "bn_mode=%d (0 eval, 1 per-scene train)", d->bn_mode);
    STG_REQUIRE(d->residual0 >= 0 && d->residual0 <= 2, STG_EINVAL, "residual0=%d", d->residual0);
    STG_REQUIRE(d->residual0 != 1 || d->c_in == d->c_out, STG_EINVAL, "identity residual needs c_in == c_out");
    const int C = Cfg::C, T = Cfg::T, P = Cfg::P, KT = Cfg::KT;
    ModelLayout &l = *lay;
    l = ModelLayout{};
    l.n_blocks = d->n_stgcnn;
    l.n_txp = d->n_txpcnn;
    l.L = d->n_txpcnn == 0 ? 0 : (d->n_txpcnn - 1 > 1 ? d->n_txpcnn - 1 : 1);
    l.use_mdn = d->use_mdn;
    l.bn_mode = d->bn_mode;
    l.eps = d->bn_eps;
    l.momentum = d->bn_momentum;
    STG_REQUIRE((d->flags & ~(STG_OPT_WG_PATH | STG_OPT_SPLIT_BF16 | STG_OPT_WAVE_PATH | STG_OPT_BF16_STORE | STG_OPT_F32_MFMA)) == 0, STG_EINVAL,
                "unknown flags 0x%x", d->flags);
    STG_REQUIRE(d->wg_waves == 0 || d->wg_waves == 1 || d->wg_waves == 2 || d->wg_waves == 4 || d->wg_waves == 8,
                STG_EINVAL, "wg_waves=%d (0 auto, 1, 2, 4, 8)", d->wg_waves);
    l.flags = d->flags;
    l.wg_waves = d->wg_waves;
    int p = 0, b = 0, s = 0, hdr = 0, u = 0;
    for (int j = 0; j < l.n_blocks; ++j) {
        BlockLayout &k = l.blk[j];
        k.cin = j == 0 ? d->c_in : C;
        // social_stgcnn builds every block with residual=True (model.py:164-166): conv+BN when the
        // channel count changes, identity otherwise (model.py:127-141).
        k.residual = j == 0 ? d->residual0 : 1;
        k.gcn_w = p; p += C * k.cin;
        k.gcn_b = p; p += C;
        k.bn1_g = p; p += C;
        k.bn1_b = p; p += C;
        k.prelu1 = p; p += 1;
        k.tcn_w = p; p += C * C * KT;
        k.tcn_b = p; p += C;
        k.bn2_g = p; p += C;
        k.bn2_b = p; p += C;
        if (k.residual == 2) {
            k.res_w = p; p += C * k.cin;
            k.res_b = p; p += C;
            k.bnr_g = p; p += C;
            k.bnr_b = p; p += C;
        } else {
            k.res_w = k.res_b = k.bnr_g = k.bnr_b = -1;
        }
        k.prelu_o = p; p += 1;
        k.n_bn = k.residual == 2 ? 3 : 2;
        k.buf = b; b += 2 * C * k.n_bn;
        k.stat = s; s += 2 * C * k.n_bn;
        k.ws_hdr = hdr; hdr += kWsHdrPerBlock;
        k.ws_ax = u; u += k.cin * T;
        k.ws_cs = u; u += T;
        k.ws_g = u; u += C * T;
        k.ws_h2 = u; u += C * T;
        k.ws_s = u; u += C * T;
    }
    for (int q = 0; q < l.n_txp; ++q) {
        const int cin = q == 0 ? T : P;
        l.txp_w[q] = p; p += P * cin * 9;
        l.txp_b[q] = p; p += P;
    }
    if (l.n_txp > 0) {
        l.out_w = p; p += P * P * 9;
        l.out_b = p; p += P;
        l.prelus = p; p += l.n_txp;
        for (int q = 0; q < l.L; ++q) { l.ws_z[q] = u; u += P * C; }
        l.n_planes = l.L + 1;                        // a_0 (the last block's output, model.py:187) .. a_L
    }
    l.n_blk_params = l.n_txp > 0 ? l.txp_w[0] : p;
    l.n_params = p;
    l.n_buffers = b;
    l.stat_floats = s;
    l.ws_hdr_floats = hdr;
    l.ws_units = u;
    return STG_OK;
}

__global__ __launch_bounds__(1024) void scene_order_kernel(const int32_t *__restrict__ num_peds, int N, int V,
                                                           int32_t *__restrict__ order,
                                                           int32_t *__restrict__ key_start,
                                                           int32_t *__restrict__ order_peds) {
    extern __shared__ int hist[];      // [K][16]
    __shared__ int wave_tot[16];
    scene_order_body<16>(num_peds, N, V, order, key_start, order_peds, hist, wave_tot);
}

bool launch_scene_order(const int32_t *num_peds, int N, int V, int32_t *order, int32_t *key_start, hipStream_t st,
                        int32_t *order_peds) {
    if (!order || !scene_order_applies(num_peds, N, V)) return false;
    const size_t lds = (size_t)(V + 1) * 16 * sizeof(int);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(&scene_order_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return false;
    hipLaunchKernelGGL(scene_order_kernel, dim3(1), dim3(1024), lds, st, num_peds, N, V, order, key_start, order_peds);
    return hipGetLastError() == hipSuccess;
}

}  // namespace stg

extern "C" {

int64_t stg_model_param_count(const stg_model_desc *d) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    return rc == STG_OK ? l.n_params : rc;
}
int64_t stg_model_buffer_count(const stg_model_desc *d) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    return rc == STG_OK ? l.n_buffers : rc;
}
int64_t stg_model_ws_floats(const stg_model_desc *d, int V) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    if (rc != STG_OK) return rc;
    if (V <= 0) return stg::fail(STG_EINVAL, "stg_model_ws_floats: V=%d", V);
    return stg::ws_floats_per_scene(l, V);
}
int stg_scene_order(const int32_t *num_peds, int N, int V, int32_t *order, int32_t *key_start, void *stream) {
    STG_REQUIRE(N >= 0 && V > 0, STG_EINVAL, "stg_scene_order: bad sizes N=%d V=%d", N, V);
    if (N == 0) return STG_OK;
    STG_REQUIRE(num_peds && order, STG_EINVAL, "stg_scene_order: null pointer");
    STG_REQUIRE(N >= 2, STG_EUNSUPPORTED, "stg_scene_order: a single scene needs no order");
    STG_REQUIRE(N <= stg::kOrderMaxN && V <= stg::kOrderMaxV, STG_EUNSUPPORTED,
                "stg_scene_order: N=%d V=%d outside the single-workgroup sort (N <= %d, V <= %d)", N, V,
                stg::kOrderMaxN, stg::kOrderMaxV);
    STG_REQUIRE(stg::launch_scene_order(num_peds, N, V, order, key_start, stg::as_stream(stream)), STG_EINVAL,
                "stg_scene_order: launch failed");
    return STG_OK;
}

int64_t stg_model_stat_floats(const stg_model_desc *d) {
    stg::ModelLayout l;
    const int rc = stg::make_layout(d, &l);
    return rc == STG_OK ? l.stat_floats : rc;
}

}  // extern "C"
