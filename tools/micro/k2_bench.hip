// k2_bench: the weight-gradient kernel (K2) alone, on synthetic saved arrays -- A/B of the shipped launcher
// (stg::launch_txp_wgrad, from libstgcnn_hip.so) against the workgroup shapes of csrc/txp_wgrad_bf16.hip as it stands in the tree, checked against each other
// and (small batches) against an fp64 host sum.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I social_stgcnn_amd/csrc -I include tools/micro/k2_bench.hip \
//          -L social_stgcnn_amd/csrc -lstgcnn_hip -Wl,-rpath,'$ORIGIN/../../social_stgcnn_amd/csrc' -o tools/micro/k2_bench
//   run:   tools/micro/k2_bench [N=2048] [V=32] [ragged=0] [bf16=0]
// the kernel source under development is compiled INTO this harness under other entry-point names; "shipped" is the library's
#define STG_K2_ALL_SHAPES
#define wgrad_bf16_fits dev_wgrad_bf16_fits
#define wgrad_bf16_geom dev_wgrad_bf16_geom
#define launch_txp_wgrad_bf16 dev_launch_txp_wgrad_bf16
#include "../../social_stgcnn_amd/csrc/txp_wgrad_bf16.hip"
#undef wgrad_bf16_fits
#undef wgrad_bf16_geom
#undef launch_txp_wgrad_bf16
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cmath>

using namespace stg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static std::vector<double> sum_rows(const std::vector<float> &slab, const WgradGeom &g, int nl) {
    std::vector<double> out;
    for (int l = 0; l < nl; ++l) {
        const int len = wgrad_row_len(l), rows = g.wg_begin[l + 1] - g.wg_begin[l];
        const int64_t base = wgrad_slab_base(l, g.rows);
        for (int e = 0; e < len; ++e) {
            double s = 0;
            for (int r = 0; r < rows; ++r) s += slab[base + (int64_t)r * len + e];
            out.push_back(s);
        }
    }
    return out;
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2048, V = argc > 2 ? atoi(argv[2]) : 32;
    const int ragged = argc > 3 ? atoi(argv[3]) : 0, bf16 = argc > 4 ? atoi(argv[4]) : 0;
    stg_model_desc d{};
    d.n_stgcnn = 1; d.n_txpcnn = 5; d.c_in = 2; d.c_out = 5; d.t_obs = 8; d.t_pred = 12; d.kt = 3; d.residual0 = 2;
    d.use_mdn = 0; d.bn_mode = 1; d.bn_eps = 1e-5f; d.bn_momentum = 0.1f; d.flags = bf16 ? STG_OPT_BF16_STORE : 0; d.wg_waves = 0;
    ModelLayout L;
    if (make_layout(&d, &L) != STG_OK) { printf("make_layout failed: %s\n", stg_last_error()); return 1; }
    const int nl = L.L + 1;
    const int64_t stride = ws_floats_per_scene(L, V), dzs = dz_slot(V);
    printf("N=%d V=%d ragged=%d bf16=%d  layers=%d ws_stride=%lld floats\n", N, V, ragged, bf16, nl, (long long)stride);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> ws((size_t)N * stride), dz((size_t)N * nl * dzs);
    std::vector<int32_t> peds(N, V);
    if (ragged) for (auto &p : peds) p = 1 + (int)(rng() % V);
    if (const char *cs = getenv("K2_COUNTS")) {              // explicit crowd sizes, comma separated, repeated over the batch
        std::vector<int> c;
        for (const char *q = cs; *q;) { c.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
        for (int n = 0; n < N; ++n) peds[n] = c[n % c.size()] > V ? V : c[n % c.size()];
    }
    const bool bf = bf16 != 0;
    const bool garbage = getenv("K2_GARBAGE") != nullptr;      // nonzero values in the unused channels 8..11 of a_0
    // saved arrays as the forward / the input-gradient kernel leave them: position-major, the scene's own row strides
    auto put = [&](float *base, int64_t pos, int ch, float v) {
        if (!bf) base[pos * 12 + ch] = v;
        else {
            union { float f; unsigned u; } cvt; cvt.f = v; const unsigned u = cvt.u;
            reinterpret_cast<unsigned short *>(base)[pos * 12 + ch] = (unsigned short)(u >> 16);
        }
    };
    auto get = [&](const float *base, int64_t pos, int ch) -> double {
        if (!bf) return base[pos * 12 + ch];
        unsigned u = (unsigned)reinterpret_cast<const unsigned short *>(base)[pos * 12 + ch] << 16;
        union { float f; unsigned u; } cvt; cvt.u = u;
        return cvt.f;
    };
    for (int n = 0; n < N; ++n) {
        const int vi = peds[n], sw = save_sw(vi, bf), vw = save_vw(vi, bf);
        for (int l = 0; l < nl; ++l) {
            float *pl = ws.data() + n * stride + ws_plane_off(L, V, l);
            float *z = dz.data() + ((int64_t)n * nl + l) * dzs;
            const int cin = l == 0 ? Cfg::T : Cfg::P;
            for (int h = 0; h < Cfg::C; ++h) {
                for (int c = 0; c < vi + 2; ++c)
                    for (int ch = 0; ch < 12; ++ch)
                        put(pl, (int64_t)h * sw + c, ch, (c == 0 || c == vi + 1) ? 0.f : (ch >= cin ? (garbage ? 3.f + U(rng) : 0.f) : U(rng)));
                for (int c = 0; c < vi; ++c)
                    for (int ch = 0; ch < 12; ++ch) put(z, (int64_t)h * vw + c, ch, U(rng) * 0.01f);
            }
        }
    }
    float *d_ws, *d_dz, *d_slab;
    int32_t *d_peds;
    CK(hipMalloc(&d_ws, ws.size() * 4)); CK(hipMalloc(&d_dz, dz.size() * 4)); CK(hipMalloc(&d_peds, N * 4));
    CK(hipMemcpy(d_ws, ws.data(), ws.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dz, dz.data(), dz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_peds, peds.data(), N * 4, hipMemcpyHostToDevice));
    const size_t slab_floats = (size_t)8192 * (wgrad_row_len(0) + (size_t)L.L * wgrad_row_len(1));
    CK(hipMalloc(&d_slab, slab_floats * 4));
    std::vector<float> slab(slab_floats);

    auto args = [&](const WgradGeom &g) {
        WgradArgs w{};
        w.lay = L; w.num_peds = ragged ? d_peds : nullptr; w.order = nullptr; w.order_peds = nullptr; w.key_start = nullptr;
        w.serpentine = 1; w.N = N; w.V = V; w.ws = d_ws; w.dzg = d_dz; w.ws_stride = stride; w.slab2 = d_slab; w.rows = g.rows;
        w.debug_skip = getenv("K2_SKIP") ? atoi(getenv("K2_SKIP")) : 0;
        for (int l = 0; l <= L.L + 1; ++l) w.wg_begin[l] = g.wg_begin[l];
        return w;
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto &&launch, const WgradGeom &g) -> std::vector<double> {
        CK(hipMemset(d_slab, 0, slab_floats * 4));
        if (launch() != STG_OK) { printf("%s: launch failed: %s\n", name, stg_last_error()); exit(1); }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(slab.data(), d_slab, slab_floats * 4, hipMemcpyDeviceToHost));
        auto sums = sum_rows(slab, g, nl);
        for (int i = 0; i < 5; ++i) launch();
        CK(hipDeviceSynchronize());
        const int reps = 40;
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s grid %4d x %2d waves, lds %6zu : %7.1f us per launch\n", name, g.grid, g.waves, g.lds, ms * 1e3 / reps);
        return sums;
    };
    WgradGeom g0{};
    if (!wgrad_geom(L, N, V, &g0)) { printf("wgrad_geom failed\n"); return 1; }
    const WgradArgs w0 = args(g0);
    const auto ref = run("shipped (launch_txp_wgrad)", [&] { return launch_txp_wgrad(w0, g0, 0); }, g0);
    double refmax = 0;
    for (double v : ref) refmax = std::fmax(refmax, std::fabs(v));
    const int nvar = 2;
    const int nws[nvar] = {5, 10}, nis[nvar] = {1, 2};
    const char *names[nvar] = {"tree 5w 1img", "tree 10w 2img"};
    for (int v = 0; v < nvar; ++v) {
        if (getenv("K2_ONLY") && atoi(getenv("K2_ONLY")) != v) continue;
        WgradGeom g{};
        if (!wgrad_geom(L, N, V, &g)) return 1;         // the library's split of the chip over the layers ...
        g.waves = nws[v]; g.nbuf = nis[v];              // ... for this workgroup shape
        g.lds = (size_t)nis[v] * kImageBytes;
        {
            const int w0 = getenv("K2_W0") ? atoi(getenv("K2_W0")) : 15, w1 = 18;      // split of the chip over the layers (halves of a tile)
            const int per_cu = getenv("K2_PERCU") ? atoi(getenv("K2_PERCU")) : (int)((size_t)kLdsBytes / g.lds);
            const int total = kNumCU * per_cu, wsum = w0 + w1 * (nl - 1);
            int begin = 0, maxw = 0;
            for (int l = 0; l < nl; ++l) {
                int cnt = total * (l == 0 ? w0 : w1) / wsum;
                if (cnt > N * wgrad_chunks(V)) cnt = N * wgrad_chunks(V);
                g.wg_begin[l] = begin; begin += cnt; if (cnt > maxw) maxw = cnt;
            }
            g.wg_begin[nl] = begin; g.grid = begin; g.rows = maxw;
        }
        const WgradArgs w = args(g);
        const auto got = run(names[v], [&] { return dev_launch_txp_wgrad_bf16(w, g, 0); }, g);
        double worst = 0;
        size_t wi = 0;
        for (size_t i = 0; i < ref.size(); ++i) {
            const double e = std::fabs(got[i] - ref[i]);
            if (!(e <= worst)) { worst = e; wi = i; }
        }
        printf("    vs shipped: max |diff| %.3e of max |ref| %.3e  (rel %.2e, at %zu: %.6e vs %.6e)\n", worst, refmax,
               worst / refmax, wi, got[wi], ref[wi]);
    }
    if (N <= 64) {
        // fp64 host sum of every layer (layer 0: c_in = 8, the others 12)
        for (int l = 0; l < nl; ++l) {
            const int cin = l == 0 ? Cfg::T : Cfg::P;
            std::vector<double> W((size_t)12 * cin * 9 + 12, 0.0);
            for (int n = 0; n < N; ++n) {
                const int vi = peds[n], sw = save_sw(vi, bf), vw = save_vw(vi, bf);
                const float *pl = ws.data() + n * stride + ws_plane_off(L, V, l);
                const float *z = dz.data() + ((int64_t)n * nl + l) * dzs;
                for (int h = 0; h < Cfg::C; ++h)
                    for (int c = 0; c < vi; ++c)
                        for (int co = 0; co < 12; ++co) {
                            const double g = get(z, (int64_t)h * vw + c, co);
                            W[(size_t)12 * cin * 9 + co] += g;
                            for (int ci = 0; ci < cin; ++ci)
                                for (int kh = 0; kh < 3; ++kh)
                                    for (int kw = 0; kw < 3; ++kw) {
                                        const int r = h + kh - 1, cc = c + kw;      // plane column = pedestrian + 1 + (kw - 1)
                                        if (r < 0 || r >= Cfg::C) continue;
                                        W[((size_t)co * cin + ci) * 9 + kh * 3 + kw] += g * get(pl, (int64_t)r * sw + cc, ci);
                                    }
                        }
            }
            size_t off = 0;
            for (int k = 0; k < l; ++k) off += wgrad_row_len(k);
            double worst = 0, mx = 0;
            for (size_t i = 0; i < W.size(); ++i) { worst = std::fmax(worst, std::fabs(W[i] - ref[off + i])); mx = std::fmax(mx, std::fabs(W[i])); }
            printf("layer %d: shipped vs fp64 host: max |diff| %.3e of %.3e\n", l, worst, mx);
        }
    }
    return 0;
}
