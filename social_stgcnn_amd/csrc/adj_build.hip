// adj_build: utils.seq_to_graph (utils.py:29-53) + anorm (utils.py:23-27) + the networkx
// normalized_laplacian_matrix call (utils.py:48-50) as one HBM-write-bound kernel.
//
// One workgroup per scene.  The (V,2,T) displacement vectors sit in LDS; pass 1 gives every row its
// degree d_h = 1 + sum_{k!=h} a_hk (fp64 accumulation of fp32 weights, one lane per (t,h) row),
// pass 2 streams the T V x V tiles to HBM with coalesced 16-byte stores:
//   L_hh = (d_h - 1)/d_h,   L_hk = -a_hk / sqrt(d_h d_k),   a_hk = 1/||p_h - p_k||  (0 if equal).
// Diagonal and off-diagonal use ONE formula, w * (dinv_h * dinv_k) with w = d_h-1 resp. -a_hk, so that the
// exact cancellations of the reference's fp64 result (two-pedestrian scenes: L = c [[1,-1],[-1,1]], which
// makes the st_gcn BatchNorm mean exactly 0 and puts PReLU inputs exactly on the kink) survive in fp32.
// Algorithmic bytes per scene-window: 64*V read + 32*V*V (+64*V nodes) written.
#include "common.hpp"

namespace stg {

__device__ __forceinline__ float inv_dist(float ax, float ay, float bx, float by) {
    // the reference subtracts, squares and adds in fp32 (0-dim tensors), then 1/sqrt; the exact
    // "== 0 -> 0" rule of utils.anorm is kept, the reciprocal square root is v_rsq_f32 (1 ulp)
    const float dx = ax - bx, dy = ay - by;
    const float s = dx * dx + dy * dy;
    return s == 0.f ? 0.f : rsqrtf(s);
}

// One workgroup per scene: the (V,2,T) displacements are read once (coalesced) into LDS; thread
// (t,h) sums the degree of row h at time t in fp64; the T V x V tiles leave as 16-byte stores.
template <bool VEC4>
__global__ __launch_bounds__(256) void adj_build_kernel(
    const float *__restrict__ rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
    const int32_t *__restrict__ num_peds, int V, int T, int normalize,
    float *__restrict__ nodes, float *__restrict__ adj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *px = sm;                // [T][V]
    float *py = px + T * V;        // [T][V]
    float *dinv = py + T * V;      // [T][V]  1/sqrt(d)
    float *diag = dinv + T * V;    // [T][V]  d-1 (sum of the off-diagonal weights)
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *r = rel + n * rel_sn;
    const bool dense = rel_st == 1 && rel_sc == T && rel_sv == 2 * T;      // reference layout (V,2,T)
    for (int e = tid; e < V * 2 * T; e += blockDim.x) {
        // e enumerates (h, c, t) in the reference's memory order so the dense case is one coalesced sweep
        const int t = e % T, hc = e / T, c = hc & 1, h = hc >> 1;
        float v = 0.f;
        if (h < vi) v = dense ? r[e] : r[h * rel_sv + c * rel_sc + t * rel_st];
        (c ? py : px)[t * V + h] = v;
    }
    __syncthreads();
    if (nodes) {
        float2 *o = reinterpret_cast<float2 *>(nodes + (int64_t)n * T * V * 2);
        for (int e = tid; e < T * V; e += blockDim.x) o[e] = make_float2(px[e], py[e]);
    }
    if (normalize) {
        for (int e = tid; e < T * V; e += blockDim.x) {
            const int t = e / V, h = e - t * V;
            if (h < vi) {
                const float *qx = px + t * V, *qy = py + t * V;
                const float hx = qx[h], hy = qy[h];
                double acc = 1.0;
                int k = 0;
                if (VEC4) {     // rows are 16-byte aligned: four neighbours per LDS read (padded slots hold (0,0))
                    for (; k + 4 <= vi; k += 4) {
                        const float4 x4 = *reinterpret_cast<const float4 *>(qx + k);
                        const float4 y4 = *reinterpret_cast<const float4 *>(qy + k);
                        const float a0 = inv_dist(hx, hy, x4.x, y4.x), a1 = inv_dist(hx, hy, x4.y, y4.y);
                        const float a2 = inv_dist(hx, hy, x4.z, y4.z), a3 = inv_dist(hx, hy, x4.w, y4.w);
                        acc += (double)(k + 0 != h ? a0 : 0.f);
                        acc += (double)(k + 1 != h ? a1 : 0.f);
                        acc += (double)(k + 2 != h ? a2 : 0.f);
                        acc += (double)(k + 3 != h ? a3 : 0.f);
                    }
                }
                for (; k < vi; ++k)
                    if (k != h) acc += (double)inv_dist(hx, hy, qx[k], qy[k]);
                dinv[e] = (float)(1.0 / sqrt(acc));
                diag[e] = (float)(acc - 1.0);
            }
        }
        __syncthreads();
    }
    float *out = adj + (int64_t)n * T * V * V;
    if (VEC4) {
        // lanes = (row within a group of rows, 16-byte column chunk), the chunk count rounded up to a power of two: no
        // integer division per element (rows of 57 -> 60 pedestrians: 15 chunks on 16 lanes), (t, h) advance incrementally
        const int v4 = V >> 2;
        int cw = 1;
        while (cw < v4) cw <<= 1;
        if (cw > (int)blockDim.x) cw = blockDim.x;
        const int rows_per_pass = blockDim.x / cw, r0 = tid / cw, c0 = tid - r0 * cw;
        int t = r0 / V, h = r0 - t * V;
        for (int th = r0; th < T * V; th += rows_per_pass) {
          for (int c = c0; c < v4; c += cw) {
            const int k0 = c << 2, e = th * v4 + c;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h < vi) {
                const float *qx = px + t * V, *qy = py + t * V, *qd = dinv + t * V;
                const float hx = qx[h], hy = qy[h];
                const float dh = normalize ? qd[h] : 1.f;
                float vals[4];
                const float4 x4 = *reinterpret_cast<const float4 *>(qx + k0);
                const float4 y4 = *reinterpret_cast<const float4 *>(qy + k0);
                const float4 d4 = normalize ? *reinterpret_cast<const float4 *>(qd + k0) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
                const float ds[4] = {d4.x, d4.y, d4.z, d4.w};
                const float dg = (normalize && h >= k0 && h < k0 + 4) ? diag[th] : 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + j;
                    float v = 0.f;
                    if (k < vi) {
                        if (k == h)
                            v = normalize ? dg * (dh * dh) : 1.f;           // same formula as off-diagonal:
                        else {                                              // V=2 rows cancel exactly like the reference's
                            const float a = inv_dist(hx, hy, xs[j], ys[j]);
                            v = normalize ? -(a * (dh * ds[j])) : a;   // dh*dk commutes: L is bitwise symmetric
                        }
                    }
                    vals[j] = v;
                }
                o = make_float4(vals[0], vals[1], vals[2], vals[3]);
            }
            reinterpret_cast<float4 *>(out)[e] = o;
          }
          h += rows_per_pass;
          while (h >= V) { h -= V; ++t; }
        }
    } else {
        for (int e = tid; e < T * V * V; e += blockDim.x) {
            const int th = e / V, k = e - th * V;
            const int t = th / V, h = th - t * V;
            float v = 0.f;
            if (h < vi && k < vi) {
                if (k == h)
                    v = normalize ? diag[th] * (dinv[th] * dinv[th]) : 1.f;
                else {
                    const float a = inv_dist(px[th], py[th], px[t * V + k], py[t * V + k]);
                    v = normalize ? -(a * (dinv[th] * dinv[t * V + k])) : a;
                }
            }
            out[e] = v;
        }
    }
}

// One workgroup per (scene, time step): the form for SMALL batches (a real ETH/UCY group is 512 scenes of 2..57 pedestrians
// padded to 60: 512 workgroups of the per-scene kernel are two per CU, and each walks its 8 x 60 rows alone -- 29 us for
// 59 MB, the latency of one workgroup).  Same operations in the same order as adj_build_kernel: bitwise identical output.
template <bool VEC4>
__global__ __launch_bounds__(256) void adj_build_tile_kernel(
    const float *__restrict__ rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
    const int32_t *__restrict__ num_peds, int V, int T, int normalize,
    float *__restrict__ nodes, float *__restrict__ adj) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *px = sm, *py = px + V, *dinv = py + V, *diag = dinv + V;      // [V] each (this time step)
    const int n = blockIdx.x / T, t = blockIdx.x - n * T, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *r = rel + n * rel_sn + t * rel_st;
    for (int e = tid; e < 2 * V; e += blockDim.x) {
        const int c = e & 1, h = e >> 1;
        (c ? py : px)[h] = h < vi ? r[h * rel_sv + c * rel_sc] : 0.f;
    }
    __syncthreads();
    if (nodes) {
        float2 *o = reinterpret_cast<float2 *>(nodes + ((int64_t)n * T + t) * V * 2);
        for (int e = tid; e < V; e += blockDim.x) o[e] = make_float2(px[e], py[e]);
    }
    if (normalize) {
        for (int h = tid; h < vi; h += blockDim.x) {
            const float hx = px[h], hy = py[h];
            double acc = 1.0;
            int k = 0;
            if (VEC4) {
                for (; k + 4 <= vi; k += 4) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(px + k);
                    const float4 y4 = *reinterpret_cast<const float4 *>(py + k);
                    const float a0 = inv_dist(hx, hy, x4.x, y4.x), a1 = inv_dist(hx, hy, x4.y, y4.y);
                    const float a2 = inv_dist(hx, hy, x4.z, y4.z), a3 = inv_dist(hx, hy, x4.w, y4.w);
                    acc += (double)(k + 0 != h ? a0 : 0.f);
                    acc += (double)(k + 1 != h ? a1 : 0.f);
                    acc += (double)(k + 2 != h ? a2 : 0.f);
                    acc += (double)(k + 3 != h ? a3 : 0.f);
                }
            }
            for (; k < vi; ++k)
                if (k != h) acc += (double)inv_dist(hx, hy, px[k], py[k]);
            dinv[h] = (float)(1.0 / sqrt(acc));
            diag[h] = (float)(acc - 1.0);
        }
        __syncthreads();
    }
    float *out = adj + ((int64_t)n * T + t) * V * V;
    if (VEC4) {
        const int v4 = V >> 2;
        int cw = 1;
        while (cw < v4) cw <<= 1;
        if (cw > (int)blockDim.x) cw = blockDim.x;
        const int rows_per_pass = blockDim.x / cw, r0 = tid / cw, c0 = tid - r0 * cw;
        for (int h = r0; h < V; h += rows_per_pass) {
            for (int c = c0; c < v4; c += cw) {
                const int k0 = c << 2;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (h < vi) {
                    const float hx = px[h], hy = py[h];
                    const float dh = normalize ? dinv[h] : 1.f;
                    float vals[4];
                    const float4 x4 = *reinterpret_cast<const float4 *>(px + k0);
                    const float4 y4 = *reinterpret_cast<const float4 *>(py + k0);
                    const float4 d4 = normalize ? *reinterpret_cast<const float4 *>(dinv + k0) : make_float4(1.f, 1.f, 1.f, 1.f);
                    const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
                    const float ds[4] = {d4.x, d4.y, d4.z, d4.w};
                    const float dg = (normalize && h >= k0 && h < k0 + 4) ? diag[h] : 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + j;
                        float v = 0.f;
                        if (k < vi) {
                            if (k == h)
                                v = normalize ? dg * (dh * dh) : 1.f;
                            else {
                                const float a = inv_dist(hx, hy, xs[j], ys[j]);
                                v = normalize ? -(a * (dh * ds[j])) : a;
                            }
                        }
                        vals[j] = v;
                    }
                    o = make_float4(vals[0], vals[1], vals[2], vals[3]);
                }
                reinterpret_cast<float4 *>(out)[h * v4 + c] = o;
            }
        }
    } else {
        for (int e = tid; e < V * V; e += blockDim.x) {
            const int h = e / V, k = e - h * V;
            float v = 0.f;
            if (h < vi && k < vi) {
                if (k == h)
                    v = normalize ? diag[h] * (dinv[h] * dinv[h]) : 1.f;
                else {
                    const float a = inv_dist(px[h], py[h], px[k], py[k]);
                    v = normalize ? -(a * (dinv[h] * dinv[k])) : a;
                }
            }
            out[e] = v;
        }
    }
}

// V == 32 fast path (the north-star crowd size): thread (t, h) owns ROW h of tile t.  It computes the row's 32
// weights once, keeps them in registers across the degree barrier (the generic kernel recomputes them), scales them
// and parks the row in an LDS tile (16-byte chunks XOR-swizzled by the row index: conflict-free although every lane
// writes a different row); the workgroup then streams the T tiles to HBM with coalesced 16-byte stores.
// Same operations in the same order as adj_build_kernel: bitwise identical output.
// ROWS: rows of the LDS tile = rows streamed out per phase.  256: the whole pass in one phase (35 KB of LDS, four workgroups per CU);
// 128: two phases (19 KB, eight per CU) -- the better choice once the batch no longer fits the chip at four per CU (same
// box, 537 MB of adjacency: 3.35 -> 3.78 TB/s; at 2048 scenes, where every workgroup is resident either way, 2 % slower).
template <int ROWS>
__global__ __launch_bounds__(256) void adj_build_rows32_kernel(
    const float *__restrict__ rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc, int64_t rel_st,
    const int32_t *__restrict__ num_peds, int T, int normalize, float *__restrict__ nodes,
    float *__restrict__ adj) {
    constexpr int V = 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *px = sm;                // [T][V]
    float *py = px + T * V;        // [T][V]
    float *dinv = py + T * V;      // [T][V]  1/sqrt(d)
    float *tile = dinv + T * V;    // [ROWS rows][V], chunk j of row e at chunk (j ^ (e & 7))
    const int n = blockIdx.x, tid = threadIdx.x;
    int vi = num_peds ? num_peds[n] : V;
    vi = vi < 0 ? 0 : (vi > V ? V : vi);
    const float *r = rel + n * rel_sn;
    const bool dense = rel_st == 1 && rel_sc == T && rel_sv == 2 * T;
    for (int e = tid; e < V * 2 * T; e += blockDim.x) {
        const int t = e % T, hc = e / T, c = hc & 1, h = hc >> 1;
        float v = 0.f;
        if (h < vi) v = dense ? r[e] : r[h * rel_sv + c * rel_sc + t * rel_st];
        (c ? py : px)[t * V + h] = v;
    }
    __syncthreads();
    if (nodes) {
        float2 *o = reinterpret_cast<float2 *>(nodes + (int64_t)n * T * V * 2);
        for (int e = tid; e < T * V; e += blockDim.x) o[e] = make_float2(px[e], py[e]);
    }
    for (int e0 = 0; e0 < T * V; e0 += blockDim.x) {       // (T = 8: one pass)
        const int e = e0 + tid;
        const bool live_row = e < T * V;
        const int ec = live_row ? e : 0;
        const int t = ec / V, h = ec - t * V;
        const float *qx = px + t * V, *qy = py + t * V;
        const float hx = qx[h], hy = qy[h];
        float a[V];
        double acc = 1.0;
#pragma unroll
        for (int k4 = 0; k4 < V / 4; ++k4) {
            const float4 x4 = *reinterpret_cast<const float4 *>(qx + 4 * k4);
            const float4 y4 = *reinterpret_cast<const float4 *>(qy + 4 * k4);
            const float xs[4] = {x4.x, x4.y, x4.z, x4.w}, ys[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * k4 + j;
                const float w = (k < vi && k != h) ? inv_dist(hx, hy, xs[j], ys[j]) : 0.f;
                a[k] = w;
                acc += (double)w;                  // (+0.0 for the skipped entries: same sum as the generic kernel)
            }
        }
        const float dh = normalize ? (float)(1.0 / sqrt(acc)) : 1.f;
        const float dg = (float)(acc - 1.0);
        if (normalize && live_row) dinv[ec] = dh;
        __syncthreads();
        // the scaled rows leave through the LDS tile, ROWS at a time: the threads of a phase park their rows, the whole
        // workgroup streams them out
        const float *qd = dinv + t * V;
        float4 *out4 = reinterpret_cast<float4 *>(adj + (int64_t)n * T * V * V);
#pragma unroll
        for (int half = 0; half < 256 / ROWS; ++half) {
            const int r0 = e0 + ROWS * half;                                   // first row of this phase
            if (live_row && tid / ROWS == half) {
                float4 *trow = reinterpret_cast<float4 *>(tile + (tid & (ROWS - 1)) * V);
#pragma unroll
                for (int k4 = 0; k4 < V / 4; ++k4) {
                    float vals[4];
                    const float4 d4 = normalize ? *reinterpret_cast<const float4 *>(qd + 4 * k4) : make_float4(1.f, 1.f, 1.f, 1.f);
                    const float ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 4 * k4 + j;
                        float v = 0.f;
                        if (h < vi && k < vi) {
                            if (k == h) v = normalize ? dg * (dh * dh) : 1.f;
                            else v = normalize ? -(a[k] * (dh * ds[j])) : a[k];
                        }
                        vals[j] = v;
                    }
                    trow[k4 ^ (ec & 7)] = make_float4(vals[0], vals[1], vals[2], vals[3]);
                }
            }
            __syncthreads();
            // coalesced copy-out of the rows [r0, r0 + ROWS): 8 chunks per row, un-swizzled on the way
            const int rows = (T * V - r0) < ROWS ? (T * V - r0) : ROWS;
            for (int f = tid; f < rows * (V / 4); f += blockDim.x) {
                const int lr = f >> 3, row = r0 + lr, p = f & 7;
                out4[row * (V / 4) + (p ^ (row & 7))] = reinterpret_cast<const float4 *>(tile + lr * V)[p];
            }
            __syncthreads();
        }
    }
}

}  // namespace stg

extern "C" int stg_adj_build(const float *rel, int64_t rel_sn, int64_t rel_sv, int64_t rel_sc,
                             int64_t rel_st, const int32_t *num_peds, int N, int V, int T,
                             int normalize, float *nodes, float *adj, void *stream) {
    STG_REQUIRE(N >= 0 && V > 0 && T > 0, STG_EINVAL, "stg_adj_build: bad sizes N=%d V=%d T=%d", N, V, T);
    STG_REQUIRE((int64_t)N * T < (1ll << 31), STG_EINVAL, "stg_adj_build: N*T too large");
    if (N == 0) return STG_OK;
    STG_REQUIRE(rel && adj, STG_EINVAL, "stg_adj_build: null rel/adj pointer");
    const size_t lds = (size_t)4 * T * V * sizeof(float);
    STG_REQUIRE(lds <= stg::kLdsBytes, STG_ELDS, "stg_adj_build: V=%d exceeds the LDS budget", V);
    const dim3 grid((unsigned)N), block(256);
    const bool vec4 = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(adj) & 15) == 0);
    if (vec4 && V == 32 && T * V <= 1024 && !stg::diag_env("STG_ADJ_TILE", 0)) {
        const int rows = N > 2048 ? 128 : 256;              // (rows of the LDS tile: see the kernel)
        const size_t lds32 = ((size_t)3 * T * V + (size_t)rows * V) * sizeof(float);
        if (lds32 <= (size_t)stg::kLdsBytes) {
            if (rows == 128) {
                hipLaunchKernelGGL(stg::adj_build_rows32_kernel<128>, grid, block, lds32, stg::as_stream(stream), rel, rel_sn,
                                   rel_sv, rel_sc, rel_st, num_peds, T, normalize, nodes, adj);
            } else {
                if (lds32 > 48 * 1024) {
                    hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&stg::adj_build_rows32_kernel<256>),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32);
                    if (e_ != hipSuccess) return stg::hip_fail(e_, "stg_adj_build: hipFuncSetAttribute");
                }
                hipLaunchKernelGGL(stg::adj_build_rows32_kernel<256>, grid, block, lds32, stg::as_stream(stream), rel, rel_sn,
                                   rel_sv, rel_sc, rel_st, num_peds, T, normalize, nodes, adj);
            }
            STG_LAUNCH_CHECK("stg_adj_build");
            return STG_OK;
        }
    }
    // small batches: a workgroup per (scene, time step) -- T times the workgroups -- while the per-scene form would leave most of
    // the chip's workgroup slots empty (measured on a real eth/train group, 512 scenes padded to 60: 29 us per-scene)
    if (N < 1024 || stg::diag_env("STG_ADJ_TILE", 0)) {
        const dim3 tgrid((unsigned)(N * T));
        const size_t tlds = (size_t)4 * V * sizeof(float);
        if (vec4)
            hipLaunchKernelGGL(stg::adj_build_tile_kernel<true>, tgrid, block, tlds, stg::as_stream(stream), rel, rel_sn,
                               rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
        else
            hipLaunchKernelGGL(stg::adj_build_tile_kernel<false>, tgrid, block, tlds, stg::as_stream(stream), rel, rel_sn,
                               rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
        STG_LAUNCH_CHECK("stg_adj_build");
        return STG_OK;
    }
    if (vec4)
        hipLaunchKernelGGL(stg::adj_build_kernel<true>, grid, block, lds, stg::as_stream(stream), rel, rel_sn,
                           rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
    else
        hipLaunchKernelGGL(stg::adj_build_kernel<false>, grid, block, lds, stg::as_stream(stream), rel, rel_sn,
                           rel_sv, rel_sc, rel_st, num_peds, V, T, normalize, nodes, adj);
    STG_LAUNCH_CHECK("stg_adj_build");
    return STG_OK;
}
