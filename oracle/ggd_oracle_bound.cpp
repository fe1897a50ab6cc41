/*
 * ggd_oracle_bound.cpp -- TEST INFRASTRUCTURE (CPU oracle): running error analysis of the per-Gaussian stages.
 *
 * The same restatement as the fp32 checker and its fp64 twin (ggd_oracle_impl.inc, included a third time) evaluated
 * over the number type ER = {v, e}: v is the value in double, e a first-order bound, in units of the fp32 unit
 * roundoff u = 2^-24, on the absolute error an fp32 evaluation OF THE SAME OPERATIONS accumulates:
 *     z = x + y : e_z = e_x + e_y + |z|            z = x * y : e_z = |x| e_y + |y| e_x + |z|
 *     z = x / y : e_z = (e_x + |z| e_y) / |y| + |z|      sqrt, exp, fma, ceil likewise
 * Literal constants that are not fp32 numbers start with e = |c| (the fp32 algorithm uses the rounded constant).
 * Inputs arrive as (v, e) pairs: e = 0 for the fp32 scene data, e = the conditioning S of the a10 sums (see
 * ggo_render_backward_ref64) for the incoming gradients of stage a11.  The outputs' e are then the per-element error
 * budget of the GPU backward test:  |gpu - v| <= ATOL + kappa * u * e  (tests/test_raster_backward_gpu.py).
 * Unlike a column-wise |J| . S this sees cancellation INSIDE the map (denom - a c, dM = 2 M G, R^T dM ...).
 * PARITY UNPINNED, like the rest of oracle/ (see the header of ggd_oracle_impl.inc).
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "ggd_oracle_types.h"

struct ER {
  double v, e;
  ER() : v(0), e(0) {}
  ER(double x) : v(x), e((double)(float)x == x ? 0.0 : std::fabs(x)) {}
  ER(double x, double err) : v(x), e(err) {}
  explicit operator double() const { return v; }
  explicit operator float() const { return (float)v; }
  explicit operator int() const { return (int)v; }
  ER operator-() const { return ER(-v, e); }
  ER& operator+=(const ER& o) { *this = ER(v + o.v, e + o.e + std::fabs(v + o.v)); return *this; }
  ER& operator-=(const ER& o) { *this = ER(v - o.v, e + o.e + std::fabs(v - o.v)); return *this; }
  ER& operator*=(const ER& o) { const double z = v * o.v; *this = ER(z, std::fabs(v) * o.e + std::fabs(o.v) * e + std::fabs(z)); return *this; }
};
static inline ER operator+(ER a, const ER& b) { a += b; return a; }
static inline ER operator-(ER a, const ER& b) { a -= b; return a; }
static inline ER operator*(ER a, const ER& b) { a *= b; return a; }
static inline ER operator/(const ER& a, const ER& b) {
  const double z = a.v / b.v, ib = 1.0 / std::fabs(b.v);
  return ER(z, (a.e + std::fabs(z) * b.e) * ib + std::fabs(z));
}
static inline bool operator<(const ER& a, const ER& b) { return a.v < b.v; }
static inline bool operator>(const ER& a, const ER& b) { return a.v > b.v; }
static inline bool operator<=(const ER& a, const ER& b) { return a.v <= b.v; }
static inline bool operator>=(const ER& a, const ER& b) { return a.v >= b.v; }
static inline bool operator==(const ER& a, const ER& b) { return a.v == b.v; }
static inline bool operator!=(const ER& a, const ER& b) { return a.v != b.v; }
static inline ER er_sqrt(const ER& a) { const double z = std::sqrt(a.v); return ER(z, (z > 0 ? a.e / (2.0 * z) : 0.0) + z); }
static inline ER er_ceil(const ER& a) { return ER(std::ceil(a.v), 0.0); }   /* a discrete decision: taken on v */
static inline ER er_exp(const ER& a) { const double z = std::exp(a.v); return ER(z, z * a.e + 2.0 * z); }
static inline ER er_fma(const ER& a, const ER& b, const ER& c) {
  const double z = std::fma(a.v, b.v, c.v);
  return ER(z, std::fabs(a.v) * b.e + std::fabs(b.v) * a.e + c.e + std::fabs(z));
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define REAL ER
#define SUF(x) CAT(x, _err)
#define R_SQRT er_sqrt
#define R_CEIL er_ceil
#define R_EXP er_exp
#define R_FMA er_fma
extern "C" {
#include "ggd_oracle_impl.inc"

/* Sigma = R S S R^T of EVERY Gaussian (stage a4 only fills it for the ones that pass its culling tests, and those
 * tests can fall the other way in double for a point that sits on a threshold in fp32). */
void ggo_cov3d_all_err(int P, const ER* scales, double mod, const ER* rotations, ER* cov3D) {
  for (int i = 0; i < P; ++i) cov3d_from_scale_rot_err(scales + 3 * i, ER(mod), rotations + 4 * i, cov3D + 6 * i);
}
}
