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

constexpr int kWave = 64;           // gfx950 wavefront
constexpr int kLdsBytes = 160 * 1024;
constexpr int kNumCU = 256;

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

}  // namespace stg
