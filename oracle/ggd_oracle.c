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

typedef struct ggo_params {
  int32_t P, M, D, W, H;
  int32_t prefiltered;
  double tanfovx, tanfovy, scale_modifier;
} ggo_params;

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

/* ---- unit-test entry points (let tests/golden pin the pieces that DO have an in-tree Python twin) ----------- */
void ggo_test_sh_to_rgb(int deg, const float* sh /*[M][3]*/, const float* p, const float* campos, float* rgb,
                        uint8_t* clamped) {
  sh_to_rgb_f32(deg, sh, p, campos, rgb, clamped);
}
void ggo_test_cov3d(const float* scale3, float mod, const float* quat4, float* cov6) {
  cov3d_from_scale_rot_f32(scale3, mod, quat4, cov6);
}
