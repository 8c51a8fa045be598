// microbenchmark + numerics check of the bf16-split conv core (social_stgcnn_amd/csrc/txp_conv_bf16.hpp): the TXP-CNN
// stack of one scene per wave (layer 0: 8 -> 12 channels + PReLU, 4 hidden 12 -> 12 with PReLU + residual, output
// conv), in-place ring image, optional global saves, against a double-precision CPU conv.
#include "../../social_stgcnn_amd/csrc/txp_conv_bf16.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace stg::cv;
constexpr int C = 5, P = 12, T = 8, NL = 6;

__global__ void prep_kernel(const float *W, unsigned *wp) {   // W [NL][12][12][3][3] (layer 0: ci < 8 real)
    const int l = blockIdx.x, lane = threadIdx.x;
    const float *Wl = W + l * 12 * 12 * 9;
    auto wf = [&](int m, int ch, int kh, int kw) -> float { return m < 12 ? Wl[((m * 12 + ch) * 3 + kh) * 3 + kw] : 0.f; };
    for (int v = 0; v < kWpVecs; ++v) {
        unsigned d[4];
        for (int q = 0; q < 4; ++q)
            d[q] = (unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q) | ((unsigned)wp_value(wf, v / 3, v % 3, lane, 2 * q + 1) << 16);
        reinterpret_cast<u32x4 *>(wp + l * kWpDwords)[v * 64 + lane] = u32x4{d[0], d[1], d[2], d[3]};
    }
}

__device__ __forceinline__ void zero_slot(unsigned char *lds, int slot, int vi, int PL) {
    const int lane = threadIdx.x & 63;
    const int n8 = row_bytes(vi) / 8;
    for (int s = 0; s < 3; ++s) {
        uint2 *p = reinterpret_cast<uint2 *>(lds + (s == 2 ? l_off(PL) : s * PL) + slot * row_bytes(vi) + kPosBytes);   // columns 0..vi-1 and the next border
        for (int e = lane; e < n8; e += 64) p[e] = make_uint2(0u, 0u);
    }
}

template <int KIND, int IN0, bool REV, int DBG>
__device__ __forceinline__ void layer(const u32x4 (&w)[kWpVecs], const float *bias, float alpha, unsigned char *lds,
                                      const ptab_t *ptab, int vi, const LaneGeom &lg, float *zsave, float *asave, float *y, int V, long long *stamps = nullptr) {
    const int kq = (threadIdx.x & 63) >> 4;
    constexpr int OUT0 = IN0 == 3 ? 1 : 3;
    f32x4 binit;
    for (int r = 0; r < 4; ++r) binit[r] = kq < 3 ? bias[4 * kq + r] : 0.f;
    conv_tiles<IN0, REV, KIND == 1, DBG>(w, binit, lds, ptab, C * vi, vi, lg, [&](const Tile &t, const f32x4 &z, const Quad &rq) {
        if (!t.ok || kq == 3) return;
        if (KIND == 2) {
            for (int r = 0; r < 4; ++r) y[(size_t)((4 * kq + r) * C + t.h) * V + t.w] = z[r];
            return;
        }
        f32x4 res = {0.f, 0.f, 0.f, 0.f};
        if (DBG & 8) { if (z[0] == 123.f) y[0] = z[1] + z[2] + z[3]; return; }
        if (KIND == 1) res = unsplit4(rq);
        f32x4 av;
        for (int r = 0; r < 4; ++r) av[r] = (z[r] > 0.f ? z[r] : alpha * z[r]) + res[r];
        put4(lds, pos_off(vi, OUT0 + t.h, t.w) + 8 * kq, lg.PL, av);
        if (zsave) {
            reinterpret_cast<f32x4 *>(zsave)[t.pos * 3 + kq] = z;
            reinterpret_cast<f32x4 *>(asave)[(t.h * (vi + 2) + t.w + 1) * 3 + kq] = av;
        }
    }, stamps);
}

template <int DBG>
__global__ __launch_bounds__(512, 1) void bench(const float *a0, const unsigned *wp, const float *bias, const float *alpha,
                                                float *y, float *saves, int N, int V, int vi, int save, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4;
    const int per = image_bytes(V) + ((T * V * 2 + 15) & ~15);
    unsigned char *lds = sm + wave * per;
    ptab_t *ptab = reinterpret_cast<ptab_t *>(lds + image_bytes(V));
    const int n = blockIdx.x * (blockDim.x >> 6) + wave;
    if (n >= N) return;
    if (mode & 128) return;
    const LaneGeom lg = lane_geom(vi);
    if ((mode & 16) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
    if ((mode & 32) && __builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(3);
    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256)
        for (int i = 0; i < (mode >> 8); ++i) __builtin_amdgcn_s_sleep(4);
    for (int e = lane; e < image_bytes(vi) / 16; e += 64) reinterpret_cast<uint4 *>(lds)[e] = make_uint4(0, 0, 0, 0);
    for (int p = lane; p < C * vi; p += 64) ptab[p] = (ptab_t)(((p / vi) << 8) | (p % vi));
    __builtin_amdgcn_wave_barrier();
    // a_0 [8][C][vi] -> high image (interior row h at slot 3 + h)
    const float *an = a0 + (size_t)n * T * C * V;
    for (int p = lane; p < C * vi; p += 64) {
        const int h = p / vi, w = p % vi;
        for (int q = 0; q < 2; ++q) {
            f32x4 v;
            for (int r = 0; r < 4; ++r) v[r] = an[((4 * q + r) * C + h) * V + w];
            put4(lds, pos_off(vi, 3 + h, w) + 8 * q, lg.PL, v);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (mode & 64) return;
    float *sv = save ? saves + (size_t)n * 5 * 2 * (C * (V + 2) * P) : nullptr;
    auto zs = [&](int l) { return sv ? sv + (size_t)(2 * l) * (C * (V + 2) * P) : nullptr; };
    auto as = [&](int l) { return sv ? sv + (size_t)(2 * l + 1) * (C * (V + 2) * P) : nullptr; };
    u32x4 w[kWpVecs];
    load_wp(wp, w);
    layer<0, 3, false, DBG>(w, bias, alpha[0], lds, ptab, vi, lg, zs(0), as(0), nullptr, V);
    __builtin_amdgcn_wave_barrier();
    zero_slot(lds, 6, vi, lg.PL);
    for (int l = 1; l < 5; ++l) {
        if (!(mode & 1)) load_wp(wp + l * kWpDwords, w);
        __builtin_amdgcn_wave_barrier();
        if (l & 1) layer<1, 1, true, DBG>(w, bias + 12 * l, alpha[l], lds, ptab, vi, lg, zs(l), as(l), nullptr, V);
        else layer<1, 3, false, DBG>(w, bias + 12 * l, alpha[l], lds, ptab, vi, lg, zs(l), as(l), nullptr, V, (l == 2 && blockIdx.x == 0) ? reinterpret_cast<long long *>(saves) : nullptr);
        __builtin_amdgcn_wave_barrier();
        zero_slot(lds, (l & 1) ? 2 : 6, vi, lg.PL);
    }
    if (!(mode & 1)) load_wp(wp + 5 * kWpDwords, w);
    __builtin_amdgcn_wave_barrier();
    layer<2, 1, false, DBG>(w, bias + 60, 0.f, lds, ptab, vi, lg, nullptr, nullptr, y + (size_t)n * P * C * V, V);   // after 5 layers the data is in the low image
    (void)kq;
}

static void cpu_ref(const float *a0, const float *W, const float *bias, const float *alpha, int V, int vi, std::vector<double> &y) {
    std::vector<double> cur(12 * C * vi, 0.0), nxt(12 * C * vi);
    for (int ch = 0; ch < T; ++ch)
        for (int h = 0; h < C; ++h)
            for (int w = 0; w < vi; ++w) cur[(ch * C + h) * vi + w] = a0[(ch * C + h) * V + w];
    for (int l = 0; l < NL; ++l) {
        for (int co = 0; co < 12; ++co)
            for (int h = 0; h < C; ++h)
                for (int w = 0; w < vi; ++w) {
                    double s = bias[12 * l + co];
                    for (int ci = 0; ci < 12; ++ci)
                        for (int kh = 0; kh < 3; ++kh)
                            for (int kw = 0; kw < 3; ++kw) {
                                const int hh = h + kh - 1, ww = w + kw - 1;
                                if (hh < 0 || hh >= C || ww < 0 || ww >= vi) continue;
                                s += (double)W[(((l * 12 + co) * 12 + ci) * 3 + kh) * 3 + kw] * cur[(ci * C + hh) * vi + ww];
                            }
                    if (l < 5) {
                        double a = s > 0 ? s : (double)alpha[l] * s;
                        if (l > 0) a += cur[(co * C + h) * vi + w];
                        nxt[(co * C + h) * vi + w] = a;
                    } else {
                        nxt[(co * C + h) * vi + w] = s;
                    }
                }
        cur = nxt;
    }
    y = cur;
}

int main(int argc, char **argv) {
    const int N = 2048, V = 32;
    const int save = argc > 1 ? atoi(argv[1]) : 1;
    std::vector<float> W(NL * 12 * 12 * 9), bias(NL * 12), alpha(5), a0((size_t)N * T * C * V);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &x : W) x = rnd() * 0.15f;
    for (int co = 0; co < 12; ++co)
        for (int ci = 8; ci < 12; ++ci)
            for (int t = 0; t < 9; ++t) W[((0 * 12 + co) * 12 + ci) * 9 + t] = 0.f;   // layer 0 has 8 input channels
    for (auto &x : bias) x = rnd() * 0.1f;
    for (auto &x : alpha) x = 0.25f + rnd() * 0.1f;
    for (auto &x : a0) x = rnd();
    float *dW, *db, *dal, *da0, *dy, *dsv;
    unsigned *dwp;
    hipMalloc(&dW, W.size() * 4); hipMalloc(&db, bias.size() * 4); hipMalloc(&dal, 20); hipMalloc(&da0, a0.size() * 4);
    hipMalloc(&dy, (size_t)N * P * C * V * 4); hipMalloc(&dwp, NL * kWpDwords * 4);
    hipMalloc(&dsv, (size_t)N * 10 * (C * (V + 2) * P) * 4);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dal, alpha.data(), 20, hipMemcpyHostToDevice); hipMemcpy(da0, a0.data(), a0.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(prep_kernel, dim3(NL), dim3(64), 0, 0, dW, dwp);
    const int per = image_bytes(V) + ((T * V * 2 + 15) & ~15);
    const size_t lds = (size_t)8 * per;
    printf("LDS per wave %d B, per workgroup %zu B\n", per, lds);
    hipFuncSetAttribute(reinterpret_cast<const void *>(bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int vi : {32, 5, 17, 1, 31}) {
        hipMemset(dy, 0, (size_t)N * P * C * V * 4);
        hipLaunchKernelGGL(bench<0>, dim3(N / 8), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, N, V, vi, save, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        std::vector<float> y((size_t)P * C * V);
        double worst = 0, scale = 0;
        for (int n : {0, 77, 2047}) {
            hipMemcpy(y.data(), dy + (size_t)n * P * C * V, y.size() * 4, hipMemcpyDeviceToHost);
            std::vector<double> ref;
            cpu_ref(a0.data() + (size_t)n * T * C * V, W.data(), bias.data(), alpha.data(), V, vi, ref);
            for (int co = 0; co < 12; ++co)
                for (int h = 0; h < C; ++h)
                    for (int w = 0; w < vi; ++w) {
                        const double r = ref[(co * C + h) * vi + w], g = y[(co * C + h) * V + w];
                        worst = fmax(worst, fabs(r - g));
                        scale = fmax(scale, fabs(r));
                    }
        }
        printf("vi=%2d: max |err| %.3e (max |y| %.3f)\n", vi, worst, scale);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto kern, const char *name, int mode = 0) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(N / 8), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, N, V, 32, 0, mode);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(kern, dim3(N / 8), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, N, V, 32, 0, mode);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f us per launch\n", name, ms * 1000 / 20);
    };
    time(bench<0>, "full");
    time(bench<16>, "full, ping-pong barriers");
    {
        hipFuncSetAttribute(reinterpret_cast<const void *>(bench<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(bench<32>, dim3(N / 8), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, N, V, 32, 0, 0);
        hipDeviceSynchronize();
        long long st[80];
        hipMemcpy(st, dsv, sizeof(st), hipMemcpyDeviceToHost);
        printf("stamps (cycles since tile start): issue | epilogue | next tile | wait1 | mma1 | wait2+mma2 ; tile total\n");
        for (int i = 0; i < 10; ++i) {
            printf(" tile %d:", i);
            for (int k = 1; k < 7; ++k) printf(" %5lld", st[i * 8 + k] - st[i * 8 + k - 1]);
            printf("  ; %5lld\n", i < 9 ? st[(i + 1) * 8] - st[i * 8] : 0);
        }
    }
    {
        hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), lds, 0, da0, dwp, db, dal, dy, dsv, 1024, V, 32, 0, 0);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), lds, 0, da0, dwp, db, dal, dy, dsv, 1024, V, 32, 0, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f us per launch\n", "1 wave/SIMD (1024 scenes)", ms * 1000 / 20);
        hipLaunchKernelGGL(bench<0>, dim3(256), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, 1024, V, 32, 0, 0);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(bench<0>, dim3(128), dim3(512), lds, 0, da0, dwp, db, dal, dy, dsv, 1024, V, 32, 0, 0);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %.1f us per launch\n", "2 waves/SIMD on half the CUs", ms * 1000 / 20);
    }
    time(bench<0>, "full, waves 4-7 +256 cyc", 1 << 8);
    time(bench<0>, "full, waves 4-7 +512 cyc", 2 << 8);
    time(bench<0>, "full, waves 4-7 +768 cyc", 3 << 8);
    time(bench<0>, "full, waves 4-7 +1024 cyc", 4 << 8);
    time(bench<0>, "full, waves 4-7 +1536 cyc", 6 << 8);
    time(bench<0>, "full, waves 4-7 +2048 cyc", 8 << 8);
    time(bench<0>, "full, waves 4-7 +4096 cyc", 16 << 8);
    time(bench<2>, "no B reads");
    time(bench<4>, "no MFMA");
    time(bench<8>, "no epilogue");
    time(bench<6>, "no reads, no MFMA");
    time(bench<12>, "no MFMA, no epilogue");
    time(bench<10>, "no reads, no epilogue");
    time(bench<14>, "skeleton");
    time(bench<14>, "skeleton, weights loaded once", 1);
    time(bench<14>, "return at once", 128);
    time(bench<14>, "zero + ptab + a0 only", 64);
    time(bench<0>, "full, weights loaded once", 1);
    return 0;
}
