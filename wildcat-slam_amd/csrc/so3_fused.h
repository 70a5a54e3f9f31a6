// so3_fused.h - device-only fused forms of the SO(3) helpers of dmath.h, used by the factor kernels of window.hip.
// Same values as the dmath.h functions to a few ulp (pinned by wc_selftest_so3_fused / tests/test_kat_gpu.py), a fraction
// of their transcendental calls.
#pragma once
#include "dmath.h"

namespace wc {

__device__ __forceinline__ double rsqrt_nr(double a) {
  double inv = __builtin_amdgcn_rsq(a);  // ~2^-26 relative; one Newton step squares that, the second is insurance the
  const double h = 0.5 * inv;             // pivot chain cannot afford (every dependent fp64 op costs ~25-30 clk here)
  return fma(h, fma(-a * inv, inv, 1.0), inv);  // inv + inv/2 (1 - a inv^2)
}

// sin and cos of a half angle.  The corrections the window solves for are milliradians: for |h| below 0.5 the two Taylor
// polynomials (to h^15 / h^16: truncation below 1e-19) take the place of the library's sincos - ~20 fused multiply-adds instead of
// ~150 instructions of argument reduction, polynomial selection and sign handling, in a kernel whose phase A is bound by fp64
// instruction issue.  The choice is PER LANE (a wave-uniform `__all` made a factor's bits depend on which other factors
// shared its wavefront - record order, piece cuts, sharded against unsharded builds); lanes with larger angles take the library
// call in a divergent branch that a wavefront of small angles skips.
__device__ __forceinline__ void sincos_half(double h, double *s, double *c) {
  if (fabs(h) < 0.5) {
    const double z = h * h;
    double ps = fma(z, -1.0 / 1307674368000.0, 1.0 / 6227020800.0);
    ps = fma(z, ps, -1.0 / 39916800.0);
    ps = fma(z, ps, 1.0 / 362880.0);
    ps = fma(z, ps, -1.0 / 5040.0);
    ps = fma(z, ps, 1.0 / 120.0);
    ps = fma(z, ps, -1.0 / 6.0);
    *s = fma(h * z, ps, h);
    double pc = fma(z, 1.0 / 20922789888000.0, -1.0 / 87178291200.0);
    pc = fma(z, pc, 1.0 / 479001600.0);
    pc = fma(z, pc, -1.0 / 3628800.0);
    pc = fma(z, pc, 1.0 / 40320.0);
    pc = fma(z, pc, -1.0 / 720.0);
    pc = fma(z, pc, 1.0 / 24.0);
    pc = fma(z, pc, -0.5);
    *c = fma(z, pc, 1.0);
  } else {
    sincos(h, s, c);
  }
}

// The IMU factor evaluates Exp of the same two rotation vectors five times, Jr of them three times and Jr^-1 of two
// logarithms (cost_functor.h:286-321, :446-448): twenty fp64 sin / cos calls and three atan2 in ONE thread's dependent
// chain (44 k clocks per factor, `-DWC_PROF`-style clocks).  Here: Exp and Jr of a vector from one sincos of the half angle
// (as surfel_side does), and Jr^-1 of a logarithm from the quaternion itself - |log q| = 2 |atan2(n, w)|,
// cos(|log q| / 2) = |w|, sin(|log q| / 2) = n - so the factor costs two sincos and three atan2.
struct ExpJr {
  Q4 E;
  M3 Jr;
};
__device__ __forceinline__ ExpJr exp_jr(V3 r) {
  ExpJr o;
  const double th2 = dot(r, r);
  if (th2 < 1e-10 * 1e-10) {  // so3.hpp:705-712, utils.h:47
    o.E = so3_exp(r);
    o.Jr = m3_identity();
    return o;
  }
  const double ith = rsqrt_nr(th2), th = th2 * ith;
  double sh, ch;
  sincos_half(0.5 * th, &sh, &ch);
  const double imag = sh * ith;
  o.E = {ch, imag * r.x, imag * r.y, imag * r.z};
  const double s2 = (sh + sh) * ith, s = s2 * ch, omc = s2 * sh;  // sin th / th, (1 - cos th) / th
  const V3 a = ith * r;
  o.Jr = s * m3_identity() + (1 - s) * outer(a, a) + (-omc) * hat(a);  // Jr(r) = Jl(-r), utils.h:46-58
  return o;
}
// so3_log (so3.hpp:264-311) and, if Jri != null, so3_Jr_inv of the result (utils.h:32-43)
__device__ __forceinline__ V3 log_jr_inv(Q4 q, M3 *Jri) {
  const double nn = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q = {q.w / nn, q.x / nn, q.y / nn, q.z / nn};
  const double sq = q.x * q.x + q.y * q.y + q.z * q.z, w = q.w;
  if (sq < 1e-10 * 1e-10) {
    const double k = 2.0 / w - (2.0 / 3.0) * sq / (w * (w * w));
    const V3 lg = mk3(k * q.x, k * q.y, k * q.z);
    if (Jri) *Jri = so3_Jr_inv(lg);
    return lg;
  }
  const double n = sqrt(sq);
  const double at = (w < 0) ? atan2(-n, -w) : atan2(n, w);
  const double k = 2.0 * at / n;
  const V3 lg = mk3(k * q.x, k * q.y, k * q.z);
  if (Jri) {
    const double th = 2.0 * fabs(at);
    if (th > 1e-10) {
      const M3 H = hat(-lg);
      const double kk = 1 - th * fabs(w) / 2 / n;  // 1 - th cos(th / 2) / 2 / sin(th / 2)
      *Jri = m3_identity() + (-0.5) * H + (kk / (th * th)) * (H * H);
    } else {
      *Jri = m3_identity();
    }
  }
  return lg;
}
}  // namespace wc
