// One element of metrics.bivariate_loss (metrics.py:84-113) and its gradient: shared by the stand-alone loss kernel
// (nll.hip) and the input stage of the wave-per-scene backward (txp_wave.hip), which computes dV_pred in place of
// reading it.
#pragma once
#include "common.hpp"

namespace stg {

// prediction (mx, my, log sx, log sy, atanh-ish corr) against the target (tx, ty): returns -log(max(pdf, 1e-20)) and
// the five partial derivatives (zero where the clamp is active).  The forward value follows the reference's operation
// order.  torch.clamp(min=eps) passes the gradient iff x >= eps.  A NaN pdf (tanh saturated to rho = +-1: 1 - rho^2 =
// 0, 0/0) is NOT clamped by torch: the loss and the element's five gradients become NaN there, so a diverged run shows
// up instead of training on a silent finite 46.05.
// FAST: the transcendental functions and reciprocals on the hardware's own instructions (v_exp_f32 / v_log_f32 /
// v_rcp_f32, 1 ulp each) instead of libm-accurate sequences and IEEE divisions: ~3x fewer vector instructions per element.
// Measured against the reference fixtures (tools/nll_fast_check.sh, diagnostic build): see DESIGN 5.3.
template <bool FAST>
__device__ __forceinline__ float nll_elem_t(float mx, float my, float a, float b, float c, float tx, float ty, bool want_grad,
                                            float (&g)[5]);

template <>
__device__ __forceinline__ float nll_elem_t<true>(float mx, float my, float a, float b, float c, float tx, float ty,
                                                  bool want_grad, float (&g)[5]) {
#pragma clang fp contract(off)
    const float dx = tx - mx, dy = ty - my;
    const float sx = __expf(a), sy = __expf(b);
    const float rho = 1.f - 2.f * __builtin_amdgcn_rcpf(__expf(2.f * c) + 1.f);       // tanh; saturates to +-1 like tanhf
    const float isx = __builtin_amdgcn_rcpf(sx), isy = __builtin_amdgcn_rcpf(sy);
    const float sxsy = sx * sy;
    const float ux = dx * isx, uy = dy * isy;
    const float cross = (rho * dx * dy) * (isx * isy);
    const float z = ux * ux + uy * uy - 2.f * cross;
    const float om = 1.f - rho * rho;
    const float iom = __builtin_amdgcn_rcpf(om);
    const float num = __expf(-z * (0.5f * iom));
    const float den = 2.f * 3.14159265358979323846f * (sxsy * __builtin_amdgcn_sqrtf(om));
    const float pdf = num * __builtin_amdgcn_rcpf(den);
    const bool nan = pdf != pdf;
    const bool live = pdf >= 1e-20f;
    g[0] = g[1] = g[2] = g[3] = g[4] = 0.f;
    if (want_grad && nan) {
        g[0] = g[1] = g[2] = g[3] = g[4] = pdf;
    } else if (want_grad && live) {
        const float qq = ux * uy;
        g[0] = -((ux - rho * uy) * isx) * iom;
        g[1] = -((uy - rho * ux) * isy) * iom;
        g[2] = 1.f - (ux * ux - rho * qq) * iom;
        g[3] = 1.f - (uy * uy - rho * qq) * iom;
        g[4] = -qq + (z * rho) * iom - rho;
    }
    return nan ? pdf : -__logf(live ? pdf : 1e-20f);
}

template <>
__device__ __forceinline__ float nll_elem_t<false>(float mx, float my, float a, float b, float c, float tx, float ty,
                                                   bool want_grad, float (&g)[5]) {
    // no fused multiply-adds here: every call site (the loss kernel, the two backward input stages) then rounds the same
    // way whatever surrounds it, and the forward value keeps the reference's operation order
#pragma clang fp contract(off)
    const float dx = tx - mx, dy = ty - my;
    const float sx = expf(a), sy = expf(b), rho = tanhf(c);
    const float sxsy = sx * sy;
    const float ux = dx / sx, uy = dy / sy;
    const float cross = (rho * dx * dy) / sxsy;
    const float z = ux * ux + uy * uy - 2.f * cross;
    const float om = 1.f - rho * rho;
    const float num = expf(-z / (2.f * om));
    const float den = 2.f * 3.14159265358979323846f * (sxsy * sqrtf(om));
    const float pdf = num / den;
    const bool nan = pdf != pdf;
    const bool live = pdf >= 1e-20f;
    g[0] = g[1] = g[2] = g[3] = g[4] = 0.f;
    if (want_grad && nan) {
        g[0] = g[1] = g[2] = g[3] = g[4] = pdf;
    } else if (want_grad && live) {
        // three correctly rounded reciprocals instead of ten divisions (a division is ~10 instructions); the quotients
        // ux = dx/sx, uy = dy/sy of the forward value are reused
        const float isx = 1.f / sx, isy = 1.f / sy, iom = 1.f / om;
        const float qq = ux * uy;
        g[0] = -((ux - rho * uy) * isx) * iom;
        g[1] = -((uy - rho * ux) * isy) * iom;
        g[2] = 1.f - (ux * ux - rho * qq) * iom;
        g[3] = 1.f - (uy * uy - rho * qq) * iom;
        g[4] = -qq + (z * rho) * iom - rho;
    }
    return nan ? pdf : -logf(live ? pdf : 1e-20f);
}

__device__ __forceinline__ float nll_elem(float mx, float my, float a, float b, float c, float tx, float ty, bool want_grad,
                                          float (&g)[5]) {
    return nll_elem_t<false>(mx, my, a, b, c, tx, ty, want_grad, g);
}

}  // namespace stg
