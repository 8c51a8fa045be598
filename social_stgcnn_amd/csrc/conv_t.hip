// conv_t: nn.Conv2d(Cin, Cout, (kt,1), padding=(pad,0)), stride 1, dilation 1 -- the temporal /
// 1x1 convolutions the reference builds at model.py:55-62 (gcn.conv), model.py:116-122 (tcn) and
// model.py:135-139 (residual).  Stand-alone kernels for the ConvTemporalGraphical module surface;
// the training hot path uses the fused scene-resident kernels in model_fwd.hip / model_bwd.hip.
//
// One workgroup walks scenes n = blockIdx.x, blockIdx.x + gridDim.x, ...; x[n] and the weights sit
// in LDS.  Weight gradients are summed in registers over all scenes of the workgroup (thread per
// weight) and leave with one float atomic per weight per workgroup.
#include "common.hpp"

namespace stg {

__global__ __launch_bounds__(256) void conv_t_fwd_kernel(
    const float *__restrict__ x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
    const float *__restrict__ w, const float *__restrict__ b, const int32_t *__restrict__ num_peds,
    int N, int Cin, int Cout, int T, int V, int kt, int pad, float *__restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *ws = sm;                       // [Cout][Cin][kt]
    float *xs = sm + Cout * Cin * kt;     // [Cin][T][V]
    const int tid = threadIdx.x;
    const int To = T + 2 * pad - kt + 1;
    for (int e = tid; e < Cout * Cin * kt; e += blockDim.x) ws[e] = w[e];
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int vi = num_peds ? num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        __syncthreads();
        const float *xn = x + n * x_sn;
        for (int e = tid; e < Cin * T * V; e += blockDim.x) {
            const int v = e % V, ct = e / V, t = ct % T, c = ct / T;
            xs[e] = v < vi ? xn[c * x_sc + t * x_st + v * x_sv] : 0.f;
        }
        __syncthreads();
        float *yn = y + (int64_t)n * Cout * To * V;
        for (int e = tid; e < Cout * To * V; e += blockDim.x) {
            const int v = e % V, ct = e / V, to = ct % To, co = ct / To;
            float acc = 0.f;
            if (v < vi) {
                acc = b ? b[co] : 0.f;
                for (int ci = 0; ci < Cin; ++ci)
                    for (int dt = 0; dt < kt; ++dt) {
                        const int ti = to + dt - pad;
                        if (ti >= 0 && ti < T)
                            acc = fmaf(ws[(co * Cin + ci) * kt + dt], xs[(ci * T + ti) * V + v], acc);
                    }
            }
            yn[e] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void conv_t_bwd_kernel(
    const float *__restrict__ x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv,
    const float *__restrict__ w, const float *__restrict__ dy, const int32_t *__restrict__ num_peds,
    int N, int Cin, int Cout, int T, int V, int kt, int pad,
    float *__restrict__ dx, float *__restrict__ dw, float *__restrict__ db) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int To = T + 2 * pad - kt + 1;
    const int nw = Cout * Cin * kt;
    float *ws = sm;                      // [Cout][Cin][kt]
    float *xs = ws + nw;                 // [Cin][T][V]
    float *ds = xs + Cin * T * V;        // [Cout][To][V]
    const int tid = threadIdx.x;
    for (int e = tid; e < nw; e += blockDim.x) ws[e] = w[e];
    // thread-owned gradient accumulators: weights tid, tid+256, ... (<= 4 per thread) and biases
    float gacc[4] = {0.f, 0.f, 0.f, 0.f};
    float bacc = 0.f;
    for (int n = blockIdx.x; n < N; n += gridDim.x) {
        int vi = num_peds ? num_peds[n] : V;
        vi = vi < 0 ? 0 : (vi > V ? V : vi);
        __syncthreads();
        const float *xn = x + n * x_sn;
        for (int e = tid; e < Cin * T * V; e += blockDim.x) {
            const int v = e % V, ct = e / V, t = ct % T, c = ct / T;
            xs[e] = v < vi ? xn[c * x_sc + t * x_st + v * x_sv] : 0.f;
        }
        const float *dyn = dy + (int64_t)n * Cout * To * V;
        for (int e = tid; e < Cout * To * V; e += blockDim.x) ds[e] = (e % V) < vi ? dyn[e] : 0.f;
        __syncthreads();
        if (dx) {
            float *dxn = dx + (int64_t)n * Cin * T * V;
            for (int e = tid; e < Cin * T * V; e += blockDim.x) {
                const int v = e % V, ct = e / V, t = ct % T, ci = ct / T;
                float acc = 0.f;
                if (v < vi)
                    for (int co = 0; co < Cout; ++co)
                        for (int dt = 0; dt < kt; ++dt) {
                            const int to = t - dt + pad;
                            if (to >= 0 && to < To)
                                acc = fmaf(ws[(co * Cin + ci) * kt + dt], ds[(co * To + to) * V + v], acc);
                        }
                dxn[e] = acc;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * 256;
            if (e < nw) {
                const int dt = e % kt, cc = e / kt, ci = cc % Cin, co = cc / Cin;
                float acc = 0.f;
                for (int to = 0; to < To; ++to) {
                    const int ti = to + dt - pad;
                    if (ti < 0 || ti >= T) continue;
                    const float *dr = ds + (co * To + to) * V;
                    const float *xr = xs + (ci * T + ti) * V;
                    for (int v = 0; v < vi; ++v) acc = fmaf(dr[v], xr[v], acc);
                }
                gacc[j] += acc;
            }
        }
        if (db && tid < Cout) {
            float acc = 0.f;
            const float *dr = ds + tid * To * V;
            for (int e = 0; e < To * V; ++e) acc += dr[e];
            bacc += acc;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + j * 256;
        if (e < nw) atomicAdd(dw + e, gacc[j]);
    }
    if (db && tid < Cout) atomicAdd(db + tid, bacc);
}

}  // namespace stg

extern "C" {

int stg_conv_t_fwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *w,
                   const float *b, const int32_t *num_peds, int N, int Cin, int Cout, int T, int V, int kt,
                   int pad, float *y, void *stream) {
    STG_REQUIRE(x && w && y, STG_EINVAL, "stg_conv_t_fwd: null pointer");
    STG_REQUIRE(N >= 0 && Cin > 0 && Cout > 0 && T > 0 && V > 0 && kt > 0 && pad >= 0, STG_EINVAL,
                "stg_conv_t_fwd: bad sizes");
    STG_REQUIRE(T + 2 * pad - kt + 1 > 0, STG_EINVAL, "stg_conv_t_fwd: empty output (T=%d kt=%d pad=%d)", T, kt, pad);
    if (N == 0) return STG_OK;
    const size_t lds = ((size_t)Cout * Cin * kt + (size_t)Cin * T * V) * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_conv_t_fwd: scene does not fit LDS (%zu bytes)", lds);
    const int grid = N < 2048 ? N : 2048;
    hipLaunchKernelGGL(stg::conv_t_fwd_kernel, dim3(grid), dim3(256), lds, stg::as_stream(stream), x, x_sn, x_sc,
                       x_st, x_sv, w, b, num_peds, N, Cin, Cout, T, V, kt, pad, y);
    STG_LAUNCH_CHECK("stg_conv_t_fwd");
    return STG_OK;
}

int stg_conv_t_bwd(const float *x, int64_t x_sn, int64_t x_sc, int64_t x_st, int64_t x_sv, const float *w,
                   const float *dy, const int32_t *num_peds, int N, int Cin, int Cout, int T, int V, int kt,
                   int pad, float *dx, float *dw, float *db, void *stream) {
    STG_REQUIRE(x && w && dy && dw, STG_EINVAL, "stg_conv_t_bwd: null pointer");
    STG_REQUIRE(N >= 0 && Cin > 0 && Cout > 0 && T > 0 && V > 0 && kt > 0 && pad >= 0, STG_EINVAL,
                "stg_conv_t_bwd: bad sizes");
    const int To = T + 2 * pad - kt + 1;
    STG_REQUIRE(To > 0, STG_EINVAL, "stg_conv_t_bwd: empty output");
    STG_REQUIRE(Cout * Cin * kt <= 1024 && Cout <= 256, STG_EUNSUPPORTED,
                "stg_conv_t_bwd: more than 1024 weights (%d) not supported", Cout * Cin * kt);
    if (N == 0) return STG_OK;
    const size_t lds = ((size_t)Cout * Cin * kt + (size_t)Cin * T * V + (size_t)Cout * To * V) * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_conv_t_bwd: scene does not fit LDS (%zu bytes)", lds);
    const int grid = N < 512 ? N : 512;
    hipLaunchKernelGGL(stg::conv_t_bwd_kernel, dim3(grid), dim3(256), lds, stg::as_stream(stream), x, x_sn, x_sc,
                       x_st, x_sv, w, dy, num_peds, N, Cin, Cout, T, V, kt, pad, dx, dw, db);
    STG_LAUNCH_CHECK("stg_conv_t_bwd");
    return STG_OK;
}

}  // extern "C"
