// Device bodies of the small per-step kernels, shared by their stand-alone launches (stg_bn_fold, stg_optim_step,
// stg_weighted_sum) and by the fused step tail (stg_train_tail: all three in ONE launch, different blocks).
#pragma once
#include "model_common.hpp"

namespace stg {

struct NbtPtrs {
    int64_t *p[3 * STG_MAX_BLOCKS];
    int n;
};

// running_mean / running_var update of statistic `i` for a whole batch at once, equal to N successive per-scene
// momentum updates r <- (1-m) r + m s_n (model.py:114,123,140; the reference forwards one scene at a time).  One
// workgroup of NT threads (a power of two; scratch arrays of NT entries): threads fold contiguous chunks of scenes,
// the chunk results compose in order by a tree.
template <int NT>
__device__ __forceinline__ void bn_fold_body(int i, const float *__restrict__ stats, const int32_t *__restrict__ num_peds,
                                             int N, int stat_floats, float momentum, float *__restrict__ buffers,
                                             const NbtPtrs &nbt, float *acc_s, float *dec_s, int *cnt_s) {
    const int tid = threadIdx.x;
    const int chunk = (N + NT - 1) / NT;
    const int lo = tid * chunk, hi = (lo + chunk) < N ? (lo + chunk) : N;
    const float keep = 1.0f - momentum;
    float acc = 0.f, dec = 1.f;
    int cnt = 0;
    for (int n0 = lo; n0 < hi; n0 += 8) {            // eight loads in flight, folded in scene order
        float sv[8];
        bool live[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int n = n0 + u;
            live[u] = n < hi && !(num_peds && num_peds[n] <= 0);
            sv[u] = live[u] ? stats[(int64_t)n * stat_floats + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (live[u]) {
                acc = fmaf(acc, keep, momentum * sv[u]);
                dec *= keep;
                ++cnt;
            }
    }
    acc_s[tid] = acc;
    dec_s[tid] = dec;
    cnt_s[tid] = cnt;
    __syncthreads();
    // ordered tree over the NT chunk results: (dec, acc) pairs compose associatively,
    // (d1, a1) then (d2, a2) = (d1 d2, a1 d2 + a2) -- log2(NT) steps instead of an NT-long serial chain
    for (int off = 1; off < NT; off <<= 1) {
        if ((tid & (2 * off - 1)) == 0) {
            const float d2 = dec_s[tid + off], a2 = acc_s[tid + off];
            acc_s[tid] = fmaf(acc_s[tid], d2, a2);
            dec_s[tid] *= d2;
            cnt_s[tid] += cnt_s[tid + off];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float r = fmaf(buffers[i], dec_s[0], acc_s[0]);
        const int total = cnt_s[0];
        buffers[i] = r;
        // one counter per BatchNorm: bumped by the block that owns the layer's first statistic
        if (i % (2 * Cfg::C) == 0) {
            const int k = i / (2 * Cfg::C);
            if (k < nbt.n && nbt.p[k]) *nbt.p[k] += total;
        }
    }
}

// clip_grad_norm_ (train.py:71-73) + SGD (train.py:197) over the flat buffers by ONE workgroup:
// total = ||g||_2, coef = min(1, max_norm / (total + 1e-6)), g *= coef (in place, as torch does), p -= lr g.
// lr comes from device memory when lr_dev != NULL, so a captured hipGraph follows the StepLR schedule.
__device__ __forceinline__ void optim_step_body(float *__restrict__ p, float *__restrict__ g, int64_t n,
                                                const float *__restrict__ lr_dev, float lr_host, float max_norm,
                                                float *__restrict__ norm_out, float *red /* [16] */) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const float lr = lr_dev ? lr_dev[0] : lr_host;
    const bool need_norm = max_norm > 0.f || norm_out;
    constexpr int R = 8;                             // values a thread keeps in registers (7,563 floats / 1024 threads)
    if (n <= (int64_t)R * nt) {
        // one trip to memory: gradients and parameters are loaded once, the norm is reduced while they sit in registers
        float gv[R], pv[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t i = tid + (int64_t)k * nt;
            gv[k] = i < n ? g[i] : 0.f;
            pv[k] = i < n ? p[i] : 0.f;
        }
        float coef = 1.f;
        if (need_norm) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < R; ++k) acc = fmaf(gv[k], gv[k], acc);
            acc = wave_sum(acc);
            if ((tid & 63) == 0) red[tid >> 6] = acc;
            __syncthreads();
            float tot = 0.f;
            for (int w = 0; w < (nt >> 6); ++w) tot += red[w];
            const float nrm = sqrtf(tot);
            if (norm_out && tid == 0) norm_out[0] = nrm;
            if (max_norm > 0.f) {
                coef = max_norm / (nrm + 1e-6f);
                coef = coef > 1.f ? 1.f : coef;
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int64_t i = tid + (int64_t)k * nt;
            if (i < n) {
                float gi = gv[k];
                if (max_norm > 0.f) {
                    gi *= coef;
                    g[i] = gi;
                }
                p[i] = pv[k] - lr * gi;
            }
        }
        return;
    }
    float coef = 1.f;
    if (need_norm) {
        float acc = 0.f;
        for (int64_t i = tid; i < n; i += nt) acc = fmaf(g[i], g[i], acc);
        acc = wave_sum(acc);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (nt >> 6); ++w) tot += red[w];
        const float nrm = sqrtf(tot);
        if (norm_out && tid == 0) norm_out[0] = nrm;
        if (max_norm > 0.f) {
            coef = max_norm / (nrm + 1e-6f);
            coef = coef > 1.f ? 1.f : coef;
        }
    }
    for (int64_t i = tid; i < n; i += nt) {
        float gi = g[i];
        if (max_norm > 0.f) {
            gi *= coef;
            g[i] = gi;
        }
        p[i] = p[i] - lr * gi;
    }
}

// out[0] = sum_n w[n] * v[n] (w null: plain sum), one workgroup, fixed summation order
__device__ __forceinline__ void weighted_sum_body(const float *__restrict__ v, const float *__restrict__ w, int N,
                                                  float *__restrict__ out, float *red /* [16] */) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    for (int i = tid; i < N; i += blockDim.x) acc = w ? fmaf(v[i], w[i], acc) : acc + v[i];
    acc = wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
        out[0] = t;
    }
}

}  // namespace stg
