/*
 * ggd_oracle.c -- TEST INFRASTRUCTURE: CPU oracle for the 3DGS rasterizer hot path (not shipped, not measured
 * as the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it).
 *
 * PARITY UNPINNED (see the header of ggd_oracle_impl.inc): the reference's native rasterizer is an empty,
 * unpinned submodule (/root/reference/.gitmodules:4-6) and the reference holds no tests for it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: plain IEEE arithmetic, no FMA contraction, so that the
 * integer outputs radii / tiles_touched / depth bits are reproducible bit-for-bit by the HIP kernels, which are
 * compiled with the same contraction setting and the same operation order).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ggd_oracle_types.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define SUF(x) CAT(x, _f32)
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_EXP expf
#define R_FMA fmaf
#include "ggd_oracle_impl.inc"
#undef REAL
#undef SUF
#undef R_SQRT
#undef R_CEIL
#undef R_EXP
#undef R_FMA
#undef SH_C0
#undef SH_C1

#define REAL double
#define SUF(x) CAT(x, _f64)
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_EXP exp
#define R_FMA fma
#include "ggd_oracle_impl.inc"
#undef REAL
#undef SUF

/* ---- integer stages (type-independent) ------------------------------------------------------------------ */

/* a5: inclusive prefix sum; returns the total R ("num_rendered"). */
int64_t ggo_scan(int P, const uint32_t* tiles_touched, uint32_t* offsets) {
  uint64_t acc = 0;
  for (int i = 0; i < P; ++i) {
    acc += tiles_touched[i];
    offsets[i] = (uint32_t)acc;
  }
  return (int64_t)acc;
}

/* Upstream's "getHigherMsb": the smallest k with (n >> k) == 0, found by bisection starting at 16.
 * 11 for 1024 tiles, 13 for 4096. */
uint32_t ggo_higher_msb(uint32_t n) {
  uint32_t msb = 16, step = 16;
  while (step > 1) {
    step /= 2;
    if (n >> msb) msb += step; else msb -= step;
  }
  if (n >> msb) msb++;
  return msb;
}

typedef struct { uint64_t key; uint32_t val; uint32_t pos; } ggo_pair;
static int ggo_pair_cmp(const void* a, const void* b) {
  const ggo_pair* x = (const ggo_pair*)a; const ggo_pair* y = (const ggo_pair*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos ? 1 : 0);
}
/* a7: STABLE ascending sort on the low `nbits` bits of the key (ties keep emission order). */
int ggo_sort_pairs(int64_t R, const uint64_t* keys_in, const uint32_t* vals_in, uint64_t* keys_out,
                   uint32_t* vals_out, int nbits) {
  if (R <= 0) return 0;
  ggo_pair* tmp = (ggo_pair*)malloc((size_t)R * sizeof(ggo_pair));
  if (!tmp) return -3;
  const uint64_t mask = nbits >= 64 ? ~0ull : ((1ull << nbits) - 1ull);
  for (int64_t i = 0; i < R; ++i) { tmp[i].key = keys_in[i] & mask; tmp[i].val = vals_in[i]; tmp[i].pos = (uint32_t)i; }
  qsort(tmp, (size_t)R, sizeof(ggo_pair), ggo_pair_cmp);
  for (int64_t i = 0; i < R; ++i) { keys_out[i] = keys_in[tmp[i].pos]; vals_out[i] = tmp[i].val; }
  free(tmp);
  return 0;
}

/* a8: identifyTileRanges.  ranges is uint32[2*T], zero-initialised here. */
void ggo_tile_ranges(int64_t R, const uint64_t* keys_sorted, int T, uint32_t* ranges) {
  memset(ranges, 0, (size_t)T * 2 * sizeof(uint32_t));
  for (int64_t i = 0; i < R; ++i) {
    const uint32_t tile = (uint32_t)(keys_sorted[i] >> 32);
    if (i == 0) ranges[2 * tile] = 0;
    else {
      const uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32);
      if (prev != tile) { ranges[2 * prev + 1] = (uint32_t)i; ranges[2 * tile] = (uint32_t)i; }
    }
    if (i == R - 1) ranges[2 * tile + 1] = (uint32_t)R;
  }
}

/* a12: mark_visible -- passes the z > 0.2 frustum test. */
void ggo_mark_visible(int P, const float* means3D, const float* view, uint8_t* present) {
  for (int i = 0; i < P; ++i) {
    float t[3];
    xform43_f32(view, means3D + 3 * i, t);
    present[i] = (uint8_t)(t[2] > 0.2f);
  }
}

/* ---- a10 reference for the GPU parity tests: fp32 DECISIONS, fp64 VALUES, plus the conditioning of every sum ------
 * Input = the fp32 state a forward pass saved (xy, conic_opacity, rgb per Gaussian; final_T, n_contrib per pixel;
 * list / ranges) -- what upstream's backward consumes.  Which (pixel, Gaussian) pairs contribute is decided exactly as
 * the fp32 forward decides it (gauss_power_f32, expf, the 1/255 floor in float), so the contributor set is the
 * forward's; every term is then evaluated in double from those fp32 inputs and accumulated in double.  `S_*` receive
 * the same accumulations with every product of every term taken in absolute value (the conditioning of the sum AND of
 * the differences inside a term): |fp32 result - this| <= kappa * eps32 * S is the bound a correct fp32
 * implementation (any summation order) can be held to; each term carries a weight = the number of fp32 roundings its
 * factors have been through (see relT below), so kappa is a plain safety factor of order 1.  `fragile[id]` counts pairs whose
 * alpha lies within 1e-6 (relative) of the 1/255 floor: an implementation whose exp differs by an ulp may decide those
 * the other way, which changes the sums discontinuously; the tests exclude such Gaussians and bound their number. */
void ggo_render_backward_ref64(const ggo_params* prm, const float* bg, const uint32_t* ranges, const uint32_t* list,
                               const float* xy, const float* conic_opacity, const float* rgb, const float* final_T,
                               const uint32_t* n_contrib, const float* dL_dpix,
                               double* dL_dmean2D /*[P,2]*/, double* dL_dconic /*[P,3]*/, double* dL_dopacity /*[P]*/,
                               double* dL_dcolors /*[P,3]*/,
                               double* S_mean2D, double* S_conic, double* S_opacity, double* S_colors,
                               uint32_t* fragile /*[P]*/) {
  const int W = prm->W, H = prm->H, gx = (W + 15) / 16;
  const float alpha_floor = 1.0f / 255.0f;
  const double ddelx_dx = 0.5 * (double)W, ddely_dy = 0.5 * (double)H;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const int tile = (py / 16) * gx + (px / 16);
      const uint32_t lo = ranges[2 * tile];
      const size_t pix = (size_t)py * W + px;
      const double T_final = (double)final_T[pix];
      double T = T_final;
      const uint32_t last_contributor = n_contrib[pix];
      double accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
      double g[3];
      for (int ch = 0; ch < 3; ++ch) g[ch] = (double)dL_dpix[(size_t)ch * H * W + pix];
      const double bg_dot_dpixel = ((double)bg[0] * g[0] + (double)bg[1] * g[1]) + (double)bg[2] * g[2];
      /* relT: relative uncertainty (in units of eps32) an fp32 evaluation carries in the running state of this pixel:
       * T is rebuilt from final_T by one division per contributor (1 rounding) by (1 - alpha), whose own relative
       * error is that of alpha (relG + 1, see below) times alpha / (1 - alpha); the blended colour behind (accum_rec)
       * takes one more rounded step per contributor.  Every |term| below is weighted by 4 + relT + relG. */
      double relT = 0;
      for (uint32_t j = lo + last_contributor; j-- > lo;) {
        const uint32_t id = list[j];
        const float* co = conic_opacity + 4 * id;
        /* decisions: the fp32 forward's */
        const float dxf = xy[2 * id] - (float)px, dyf = xy[2 * id + 1] - (float)py;
        const float power32 = gauss_power_f32(co[0], co[1], co[2], dxf, dyf);
        if (power32 > 0.0f) continue;
        const float alpha32 = rmin_f32(0.99f, co[3] * expf(power32));
        if (fabs((double)alpha32 * 255.0 - 1.0) <= 1e-6) fragile[id] += 1;
        if (alpha32 < alpha_floor) continue;
        /* values: double, from the same fp32 inputs (dx, dy are exact differences of floats when rounded once in
         * float as the forward does; the double difference is the exact value) */
        const double dx = (double)xy[2 * id] - (double)px, dy = (double)xy[2 * id + 1] - (double)py;
        const double A = co[0], B = co[1], C = co[2], o = co[3];
        const double power = gauss_power_f64(A, B, C, dx, dy);
        const double G = exp(power);
        const double alpha = rmin_f64((double)0.99f, o * G);   /* the fp32 algorithm's cap is the float 0.99f */
        T = T / (1.0 - alpha);
        /* power is a cancelling sum of three products of rounded factors (dx, dy are themselves rounded differences):
         * its absolute error, i.e. the RELATIVE error of G = exp(power), is ~2 roundings per product */
        const double relG = 2.0 + 2.0 * ((fabs(A) * dx * dx + fabs(C) * dy * dy) + 2.0 * fabs(B * dx * dy));
        const double rel_alpha = (o * G > (double)0.99f) ? 0.0 : relG + 1.0;   /* the capped value is exact */
        relT += 2.0 + rel_alpha * alpha / (1.0 - alpha);
        const double wgt = 4.0 + relT + relG;
        const double dchannel_dcolor = alpha * T;
        double dL_dalpha = 0, mag = 0;   /* mag: the same expression with every product taken in absolute value */
        for (int ch = 0; ch < 3; ++ch) {
          const double c = rgb[3 * id + ch];
          accum_rec[ch] = last_alpha * last_color[ch] + (1.0 - last_alpha) * accum_rec[ch];
          last_color[ch] = c;
          dL_dalpha += (c - accum_rec[ch]) * g[ch];
          mag += (fabs(c) + fabs(accum_rec[ch])) * fabs(g[ch]);
          const double t = dchannel_dcolor * g[ch];
          dL_dcolors[3 * (size_t)id + ch] += t; S_colors[3 * (size_t)id + ch] += wgt * fabs(t);
        }
        dL_dalpha *= T; mag *= T;
        last_alpha = alpha;
        const double bgw = T_final / (1.0 - alpha);
        dL_dalpha += -bgw * bg_dot_dpixel;
        mag += bgw * ((fabs((double)bg[0] * g[0]) + fabs((double)bg[1] * g[1])) + fabs((double)bg[2] * g[2]));
        const double dL_dG = o * dL_dalpha, mG = wgt * fabs(o) * mag;
        const double gdx = G * dx, gdy = G * dy;
        const double dG_ddelx = -gdx * A - gdy * B;
        const double dG_ddely = -gdy * C - gdx * B;
        dL_dmean2D[2 * (size_t)id + 0] += dL_dG * dG_ddelx * ddelx_dx;
        S_mean2D[2 * (size_t)id + 0] += mG * (fabs(gdx * A) + fabs(gdy * B)) * ddelx_dx;
        dL_dmean2D[2 * (size_t)id + 1] += dL_dG * dG_ddely * ddely_dy;
        S_mean2D[2 * (size_t)id + 1] += mG * (fabs(gdy * C) + fabs(gdx * B)) * ddely_dy;
        dL_dconic[3 * (size_t)id + 0] += -0.5 * gdx * dx * dL_dG; S_conic[3 * (size_t)id + 0] += fabs(0.5 * gdx * dx) * mG;
        dL_dconic[3 * (size_t)id + 1] += -0.5 * gdx * dy * dL_dG; S_conic[3 * (size_t)id + 1] += fabs(0.5 * gdx * dy) * mG;
        dL_dconic[3 * (size_t)id + 2] += -0.5 * gdy * dy * dL_dG; S_conic[3 * (size_t)id + 2] += fabs(0.5 * gdy * dy) * mG;
        dL_dopacity[id] += G * dL_dalpha; S_opacity[id] += wgt * G * mag;
      }
    }
}

/* Pixels whose fp32 blend contains a DECISION that one ulp of exp() can flip: a (pixel, Gaussian) pair whose alpha sits
 * within `window` (relative) of the 1/255 floor, or a transmittance test T (1 - alpha) within 4 `window` of the 1e-4 stop.
 * Walks every pixel front to back exactly as the fp32 forward (ggo_render_f32).  A flip in the middle of a pixel's list
 * changes the transmittance of everything behind it by a factor (1 - 1/255): both outcomes are correct fp32 results, but
 * they differ by far more than rounding, in the pixel's colour and in the gradient of every Gaussian behind the flipped
 * one.  The parity tests therefore compare colours on the other pixels and give these pixels zero upstream gradient (for
 * the HIP backward and for the reference alike), and bound their number. */
int64_t ggo_fragile_pixels(const ggo_params* prm, const uint32_t* ranges, const uint32_t* list, const float* xy,
                           const float* conic_opacity, double window, uint8_t* mask /*[H*W]*/) {
  const int W = prm->W, H = prm->H, gx = (W + 15) / 16;
  const float alpha_floor = 1.0f / 255.0f;
  int64_t count = 0;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const int tile = (py / 16) * gx + (px / 16);
      const uint32_t lo = ranges[2 * tile], hi = ranges[2 * tile + 1];
      float T = 1.0f;
      uint8_t frag = 0;
      for (uint32_t j = lo; j < hi; ++j) {
        const uint32_t id = list[j];
        const float dx = xy[2 * id] - (float)px, dy = xy[2 * id + 1] - (float)py;
        const float* co = conic_opacity + 4 * id;
        const float power = gauss_power_f32(co[0], co[1], co[2], dx, dy);
        if (power > 0.0f) continue;
        const float alpha = rmin_f32(0.99f, co[3] * expf(power));
        if (fabs((double)alpha * 255.0 - 1.0) <= window) frag = 1;
        if (alpha < alpha_floor) continue;
        const float test_T = T * (1.0f - alpha);
        if (fabs((double)test_T * 1e4 - 1.0) <= 4.0 * window) frag = 1;   /* T carries the exp() ulps of every contributor in front */
        if (test_T < 0.0001f) break;
        T = test_T;
      }
      mask[(size_t)py * W + px] = frag;
      count += frag;
    }
  return count;
}

/* ---- unit-test entry points (let tests/golden pin the pieces that DO have an in-tree Python twin) ----------- */
void ggo_test_sh_to_rgb(int deg, const float* sh /*[M][3]*/, const float* p, const float* campos, float* rgb,
                        uint8_t* clamped) {
  sh_to_rgb_f32(deg, sh, p, campos, rgb, clamped);
}
void ggo_test_cov3d(const float* scale3, float mod, const float* quat4, float* cov6) {
  cov3d_from_scale_rot_f32(scale3, mod, quat4, cov6);
}
