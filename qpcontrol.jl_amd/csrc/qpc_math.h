// qpc_math.h -- the two elementary functions of the tick as straight-line fp64 code: no private (scratch) memory, a handful of
// registers, ~40-60 instructions each (the library versions need several hundred instructions and, inlined into the 256-VGPR tick,
// push its register allocation into scratch: the kernel with the StandingController laws fused in spilled 512 bytes per lane around
// the library atan2).  Pure functions of their arguments; tests/offline/math_check.cpp compares them with libm on the host.
#pragma once
#ifndef QPC_MATH_FN
#define QPC_MATH_FN __device__ __forceinline__
#endif

// sin and cos of a joint angle together: three-term Cody-Waite reduction by pi/2 (exact products up to |x| ~ 1.6e6 rad)
// and the fdlibm kernel polynomials on [-pi/4, pi/4] -- about 40 straight-line fp64 instructions, < 1 ulp-ish (1e-16
// absolute), against several hundred for the library sincos, whose Payne-Hanek path for huge arguments also needs
// private (scratch) memory.  Beyond ~1e6 rad the reduction loses accuracy gracefully (no joint angle lives there).
QPC_MATH_FN void qpc_sincos(double x, double &s, double &c) {
  const double kf = rint(x * 6.36619772367581382433e-01); // x * 2/pi
  double r = fma(-kf, 1.57079632673412561417e+00, x);     // pi/2 in three 33-bit pieces (fdlibm pio2_1..3)
  r = fma(-kf, 6.07710050630396597660e-11, r);
  r = fma(-kf, 2.02226624871116645580e-21, r);
  const int q = (int)kf & 3;
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                          -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double sr = fma(r * z, ps, r);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}

// atan2(y, x) for finite arguments: fdlibm's atan (argument reduction against 7/16 and 11/16 -- the ratio is folded into [0, 1]
// first -- and the odd polynomial of degree 23) written branch-free, the two divisions as reciprocal seed + two Newton steps.  < 2 ulp against libm on the host check;
// atan2(0, 0) = 0, atan2(0, x < 0) = pi as in libm (signed zeros of y are not distinguished: the callers pass y >= 0 norms).
QPC_MATH_FN double qpc_rcp_newton(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
#else
  return 1.0 / x;
#endif
}
QPC_MATH_FN double qpc_atan_01(double x) { // atan(x) for 0 <= x <= 1 (the ranges of fdlibm's reduction that such x can reach)
  const bool r0 = x < 0.4375, r1 = x < 0.6875;
  const double num = r1 ? 2.0 * x - 1.0 : x - 1.0, den = r1 ? 2.0 + x : x + 1.0; // t = (2x - 1) / (2 + x)  |  (x - 1) / (x + 1)
  const double t = r0 ? x : num * qpc_rcp_newton(den);
  const double hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : 7.85398163397448278999e-01); // atan(0.5), atan(1)
  const double lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : 3.06161699786838301793e-17);
  const double z = t * t, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02), 6.66107313738753120669e-02),
                                              9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                                      -1.11111104054623557880e-01), -1.99999999998764832476e-01);
  return hi - ((t * (s1 + s2) - lo) - t);
}
QPC_MATH_FN double qpc_atan2(double y, double x) {
  const double ay = y < 0.0 ? -y : y, ax = x < 0.0 ? -x : x;
  // atan of the smaller ratio where the division is safe: ay / ax when ax >= ay, else pi/2 - atan(ax / ay)
  const bool swap = ay > ax;
  const double nu = swap ? ax : ay, de = swap ? ay : ax;
  const double q = de > 0.0 ? nu * qpc_rcp_newton(de) : 0.0;
  double a = qpc_atan_01(q);
  a = swap ? 1.57079632679489655800e+00 - a + 6.12323399573676603587e-17 : a;
  a = x < 0.0 ? 3.14159265358979311600e+00 - a + 1.22464679914735317720e-16 : a;
  return y < 0.0 ? -a : a;
}

// ---- fp32 variant of the tick (QPC_F32): the same two functions in float.  sin / cos: two-term Cody-Waite reduction by pi/2 (joint
// angles are a few radians: exact products up to |x| ~ 1e3 rad, then a graceful loss) and the fdlibm float kernel polynomials on
// [-pi/4, pi/4], ~1 ulp; ~20 straight-line instructions.  (Written with float literals throughout: this header is also compiled
// without the single-precision-constant option.)
QPC_MATH_FN void qpc_sincosf(float x, float &s, float &c) {
  const float kf = rintf(x * 6.3661977236758134e-01f);
  float r = fmaf(-kf, 1.5707855224609375e+00f, x);      // pi/2 = 1.5707855 (12 bits) + 1.0804334e-05 + 6.0770999e-11
  r = fmaf(-kf, 1.0804334124e-05f, r);
  r = fmaf(-kf, 6.0770999344e-11f, r);
  const int q = (int)kf & 3;
  const float z = r * r;
  const float ps = fmaf(z, fmaf(z, fmaf(z, 2.7557314297e-06f, -1.9841270114e-04f), 8.3333337680e-03f), -1.6666667163e-01f);
  const float pc = fmaf(z, fmaf(z, fmaf(z, -2.7557314297e-07f, 2.4801587642e-05f), -1.3888889225e-03f), 4.1666667908e-02f);
  const float sr = fmaf(r * z, ps, r);
  const float cr = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const float s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  s = (q & 2) ? -s0 : s0;
  c = ((q + 1) & 2) ? -c0 : c0;
}
// (atan2 in float -- only the orientation / SE(3) laws use it -- is the library atan2f: straight-line code, no private memory)
