// host check of qpcontrol.jl_amd/csrc/qpc_math.h against libm (the device versions use the same formulas with the
// hardware reciprocal seed + two Newton steps in place of the divisions):
//   g++ -O2 -std=c++17 -o /tmp/math_check tests/offline/math_check.cpp && /tmp/math_check
#include <cmath>
#include <cstdio>
#include <random>
#define QPC_MATH_FN static inline
#include "../../qpcontrol.jl_amd/csrc/qpc_math.h"

static double ulps(double a, double b) {
  if (a == b) return 0.0;
  const double u = std::nextafter(std::fabs(b), INFINITY) - std::fabs(b);
  return std::fabs(a - b) / u;
}

int main() {
  std::mt19937_64 g(12345);
  std::uniform_real_distribution<double> U(-1.0, 1.0), A(-40.0, 40.0);
  double worst = 0.0, worst_abs = 0.0, ws = 0.0, wc = 0.0;
  const long n = 20000000;
  for (long i = 0; i < n; i++) {
    // magnitudes over ~12 decades, all four quadrants, plus the axes
    double y = U(g) * std::exp2(A(g) * 0.5), x = U(g) * std::exp2(A(g) * 0.5);
    if (i % 1000 == 0) x = 0.0;
    if (i % 1000 == 1) y = 0.0;
    if (i % 1000 == 2) y = x;
    const double r = qpc_atan2(y, x), ref = std::atan2(y, x);
    const double e = ulps(r, ref);
    if (e > worst) worst = e;
    if (std::fabs(r - ref) > worst_abs) worst_abs = std::fabs(r - ref);
    double s, c;
    const double t = U(g) * 1.0e4;
    qpc_sincos(t, s, c);
    if (std::fabs(s - std::sin(t)) > ws) ws = std::fabs(s - std::sin(t));
    if (std::fabs(c - std::cos(t)) > wc) wc = std::fabs(c - std::cos(t));
  }
  std::printf("atan2: worst %.2f ulp, worst |error| %.3e over %ld points; sincos |error| %.3e / %.3e (|x| < 1e4)\n", worst, worst_abs, n, ws, wc);
  return (worst <= 2.5 && ws < 5e-16 && wc < 5e-16) ? 0 : 1;
}
