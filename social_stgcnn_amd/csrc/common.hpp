// Shared host/device helpers for libstgcnn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "../../include/stgcnn_hip.h"

namespace stg {

// ---- error reporting -------------------------------------------------------------------
char *last_error_buf();
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define STG_REQUIRE(cond, code, ...)            \
    do {                                        \
        if (!(cond)) return stg::fail((code), __VA_ARGS__); \
    } while (0)

#define STG_LAUNCH_CHECK(what)                           \
    do {                                                 \
        hipError_t e_ = hipGetLastError();               \
        if (e_ != hipSuccess) return stg::hip_fail(e_, what); \
    } while (0)

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// The product library reads NO environment variable.  A diagnostic build (make DIAG=1 -> -DSTG_DIAG,
// libstgcnn_hip_diag.so, used by tools/ only) can override tuning constants and skip kernel phases for timing.
#ifdef STG_DIAG
int diag_env(const char *name, int dflt);
#define STG_SKIP(args, bit) (((args).debug_skip & (bit)) != 0)
#else
static inline int diag_env(const char *, int dflt) { return dflt; }
#define STG_SKIP(args, bit) false
#endif

// Desynchronise the two waves that share a SIMD: every wave of a persistent wave-per-item kernel walks the same
// phases (stage / VALU, then MFMA) on equally sized items, so SIMD partners that start together stay in lockstep --
// both stage, then both fight for the matrix pipe -- and nothing overlaps.  The second half of a workgroup's waves
// (wave >= waves/2 shares a SIMD with wave - waves/2: waves are dealt to the SIMDs cyclically) starts `units` x 64 x 127
// cycles late, about half a phase period, and from then on one partner stages while the other computes.
__device__ __forceinline__ void stagger_start(int wave, int waves, int units) {
    // units < 0: no delay, the second half of the waves gets issue priority instead (-1 -> prio 1, ... -3 -> prio 3)
    if (waves >= 2 && wave >= waves / 2) {
        if (units == -1) __builtin_amdgcn_s_setprio(1);
        else if (units == -2) __builtin_amdgcn_s_setprio(2);
        else if (units <= -3) __builtin_amdgcn_s_setprio(3);
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// per-kernel device timing requested by the caller (stg_model_fwd / stg_model_bwd `events`)
struct EventList {
    void **ev;
    int n, next;
    hipStream_t st;
    void mark() {
        if (ev && next < n) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(ev[next]), st);
        ++next;
    }
    // record the events the entry point did not reach, so that every handle the caller passed can be read back
    void finish() {
        while (ev && next < n) mark();
    }
};

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kLdsBytes = 160 * 1024;
constexpr int kNumCU = 256;

// ---- device helpers --------------------------------------------------------------------
// Sum over the 64 lanes of the wave, result in every lane.  Six DPP-modified v_add (quad swaps, row
// mirrors, row broadcasts -- pure VALU, no LDS crossbar, no s_waitcnt) leave the total in lane 63;
// v_readlane broadcasts it.  Needs all 64 lanes active (call from wave-uniform control flow).
__device__ __forceinline__ float wave_sum(float v) {
#define STG_DPP_ADD(ctrl, row_mask)                                                                              \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), (ctrl), (row_mask), 0xf, false))
    STG_DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
    STG_DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
    STG_DPP_ADD(0x141, 0xf);   // row_half_mirror
    STG_DPP_ADD(0x140, 0xf);   // row_mirror            -> every lane holds its 16-lane row sum
    STG_DPP_ADD(0x142, 0xa);   // row_bcast15 into rows 1, 3
    STG_DPP_ADD(0x143, 0xc);   // row_bcast31 into rows 2, 3 -> lane 63 holds the wave sum
#undef STG_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// K independent wave sums at once: every DPP stage is applied to all K values before the next stage, so the
// dependent-DPP wait states of one chain are filled by the other chains (a lone wave_sum costs ~350 cycles of
// nops and readlane hazards at two waves per SIMD; K at once ~30 cycles per value).
// the DPP stages alone: lane 63 ends up with the K wave totals (the other lanes hold partial sums)
template <int K>
__device__ __forceinline__ void wave_sum_to_last(float (&v)[K]) {
#define STG_DPP_STAGE(ctrl, row_mask)                                                                                \
    _Pragma("unroll") for (int k = 0; k < K; ++k)                                                                    \
        v[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), (ctrl), (row_mask), 0xf, false))
    STG_DPP_STAGE(0xB1, 0xf);
    STG_DPP_STAGE(0x4E, 0xf);
    STG_DPP_STAGE(0x141, 0xf);
    STG_DPP_STAGE(0x140, 0xf);
    STG_DPP_STAGE(0x142, 0xa);
    STG_DPP_STAGE(0x143, 0xc);
#undef STG_DPP_STAGE
}
template <int K>
__device__ __forceinline__ void wave_sum_n(float (&v)[K]) {
    wave_sum_to_last<K>(v);
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace stg
