// The transcendental functions of the proposal draw (shared by
// nb_kernels.hip -- the draw kernel -- and nb_eval_fast.hip -- the acceptance
// kernel that draws its own proposals).
#pragma once
#include "nb_common.h"

namespace {

// ---- the two transcendental functions of a Box-Muller pair, for the
// arguments the draw actually has: u = (w + 1/2) / 2^32, w a 32-bit word.
// The library's log / sincospi handle every double (zeros, subnormals,
// infinities, huge arguments) at twice the instructions; these are the
// classic polynomial kernels (fdlibm: e_log.c, k_sin.c, k_cos.c) on the
// reduced ranges, < 1 ulp against the exact values (checked against mpmath
// over 2 x 10^6 words, the corner words included).

// a / b for b in [1.7, 2.5]: reciprocal estimate + two Newton steps + residual
__device__ __forceinline__ double draw_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// log(u), u normal in (0, 1)
__device__ __forceinline__ double draw_log(double u) {
  double m = __builtin_amdgcn_frexp_mant(u);          // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(u);
  const bool low = m < 0.70710678118654752;
  m = low ? 2.0 * m : m;                              // [sqrt(1/2), sqrt(2))
  const double k = (double)(low ? e - 1 : e);
  const double f = m - 1.0;
  const double s = draw_div(f, 2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 =
      w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01),
              3.999999999940941908e-01);
  const double t2 =
      z * fma(w, fma(w, fma(w, 1.479819860511658591e-01,
                            1.818357216161805012e-01),
                     2.857142874366239149e-01),
              6.666666666666735130e-01);
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  return k * 6.93147180369123816490e-01 -
         ((hfsq - fma(s, hfsq + R, k * 1.90821492927058770002e-10)) - f);
}

// sin(2 pi u), cos(2 pi u) for u in (0, 1): quadrant q = rint(4 u), the rest
// y = (4 u - q) pi / 2 in [-pi/4, pi/4] (4 u - q is exact)
__device__ __forceinline__ void draw_sincos(double u, double& sn, double& cs) {
  const double a = 4.0 * u, q = __builtin_rint(a);
  const double y = (a - q) * 1.5707963267948966;
  const double z = y * y;
  const double r = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10,
                                            -2.50507602534068634195e-08),
                                     2.75573137070700676789e-06),
                              -1.98412698298579493134e-04),
                       8.33333333332248946124e-03);
  const double s0 = fma(y * z, fma(z, r, -1.66666666666666324348e-01), y);
  const double rc =
      z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11,
                                          2.08757232129817482790e-09),
                                   -2.75573143513906633035e-07),
                            2.48015872894767294178e-05),
                     -1.38888888888741095749e-03),
              4.16666666666666019037e-02);
  const double c0 = 1.0 - (0.5 * z - z * rc);
  const int qi = (int)q & 3;
  const double sb = (qi & 1) ? c0 : s0, cb = (qi & 1) ? s0 : c0;
  sn = (qi & 2) ? -sb : sb;
  cs = ((qi + 1) & 2) ? -cb : cb;
}

// One Box-Muller pair from two 32-bit words (RNG contract, DESIGN.md section
// 3: uniforms (w + 1/2) / 2^32)
__device__ __forceinline__ void draw_normal_pair(uint32_t w0, uint32_t w1,
                                                 double& z0, double& z1) {
  const double r = sqrt(-2.0 * draw_log(nb_unit32(w0)));
  double sn, cs;
  draw_sincos(nb_unit32(w1), sn, cs);
  z0 = r * cs;
  z1 = r * sn;
}

}  // namespace
