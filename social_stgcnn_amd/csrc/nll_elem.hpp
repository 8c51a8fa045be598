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
__device__ __forceinline__ float nll_elem(float mx, float my, float a, float b, float c, float tx, float ty, bool want_grad,
                                          float (&g)[5]) {
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
        const float qq = (dx * dy) / sxsy;
        g[0] = -(dx / (sx * sx) - rho * dy / sxsy) / om;
        g[1] = -(dy / (sy * sy) - rho * dx / sxsy) / om;
        g[2] = 1.f - (ux * ux - rho * qq) / om;
        g[3] = 1.f - (uy * uy - rho * qq) / om;
        g[4] = -qq + z * rho / om - rho;
    }
    return nan ? pdf : -logf(live ? pdf : 1e-20f);
}

}  // namespace stg
