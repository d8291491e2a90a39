// qpc_real.h -- `real`, the arithmetic type qpc_tick.h is written against.
//   default   : double (fp64 tick, the product's parity path)
//   QPC_F32   : a float wrapped in a class, so that the SAME source -- written with plain literals (0.5, 1e-9, ...) -- runs in
//               single precision throughout: a literal meets a `real` only through the converting constructor, i.e. it is narrowed
//               at compile time and the operation itself is a float operation.  (A plain `typedef float real` would promote every
//               `x * 0.5` to double arithmetic; clang's -cl-single-precision-constant does not survive template instantiation.)
//               Conversions OUT of a real are explicit, so a forgotten fp64 detour does not compile.
// Shared by the device build (qpc_device.h) and the CPU emulation (tests/emul/emul.cpp).
#pragma once

#if defined(__HIPCC__) || defined(__HIPCC_RTC__) || defined(__HIP_DEVICE_COMPILE__)
#define QPC_REAL_FN __host__ __device__ __forceinline__
#else
#define QPC_REAL_FN inline
#endif

#ifdef QPC_F32
struct real {
  float v;
  real() = default;
  QPC_REAL_FN constexpr real(float x) : v(x) {}
  QPC_REAL_FN constexpr real(double x) : v((float)x) {}
  QPC_REAL_FN constexpr real(int x) : v((float)x) {}
  QPC_REAL_FN explicit constexpr operator float() const { return v; }
  QPC_REAL_FN explicit constexpr operator double() const { return (double)v; }
  QPC_REAL_FN explicit constexpr operator int() const { return (int)v; }
  QPC_REAL_FN real &operator+=(real o) { v += o.v; return *this; }
  QPC_REAL_FN real &operator-=(real o) { v -= o.v; return *this; }
  QPC_REAL_FN real &operator*=(real o) { v *= o.v; return *this; }
  QPC_REAL_FN real &operator/=(real o) { v /= o.v; return *this; }
};
QPC_REAL_FN constexpr real operator+(real a, real b) { return real(a.v + b.v); }
QPC_REAL_FN constexpr real operator-(real a, real b) { return real(a.v - b.v); }
QPC_REAL_FN constexpr real operator*(real a, real b) { return real(a.v * b.v); }
QPC_REAL_FN constexpr real operator/(real a, real b) { return real(a.v / b.v); }
QPC_REAL_FN constexpr real operator-(real a) { return real(-a.v); }
QPC_REAL_FN constexpr real operator+(real a) { return a; }
QPC_REAL_FN constexpr bool operator<(real a, real b) { return a.v < b.v; }
QPC_REAL_FN constexpr bool operator>(real a, real b) { return a.v > b.v; }
QPC_REAL_FN constexpr bool operator<=(real a, real b) { return a.v <= b.v; }
QPC_REAL_FN constexpr bool operator>=(real a, real b) { return a.v >= b.v; }
QPC_REAL_FN constexpr bool operator==(real a, real b) { return a.v == b.v; }
QPC_REAL_FN constexpr bool operator!=(real a, real b) { return a.v != b.v; }
// the tables in HBM, read as `real`: the device's fp32 module gets a float copy of them (QPC_DTAB_F32: no conversion per read, half
// the bytes); the CPU emulation reads the double tables and narrows -- the same values either way
struct qpc_dtab {
#ifdef QPC_DTAB_F32
  const float *p;
  QPC_REAL_FN real operator[](int i) const { return real(p[i]); }
#else
  const double *p;
  QPC_REAL_FN real operator[](int i) const { return real((float)p[i]); }
#endif
  QPC_REAL_FN qpc_dtab operator+(int o) const { return qpc_dtab{p + o}; }
};
#define QPC_TO_DOUBLE(x) ((double)(x))
#else
typedef double real;
typedef const double *qpc_dtab;
#define QPC_TO_DOUBLE(x) (x)
#endif
