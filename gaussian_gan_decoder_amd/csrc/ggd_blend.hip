// ggd_blend.hip -- stages a9 (forward alpha blend) and a10 (backward blend) for gfx950.
//
// Replaces renderCUDA fwd/bwd inside `_C.rasterize_gaussians{,_backward}` (reached from
// gaussian_splatting/gaussian_renderer/__init__.py:167-175; algorithm restated in SURVEY.md 9.4 / 9.5).
//
// wave64 design (not the 16x16-thread block of the CUDA original):
//   * A 64-lane wave owns a 16x16 tile (4 horizontally adjacent pixels per lane) or a 16x8 half of it (2 pixels per
//     lane; template PXL) -- two waves per tile give finer culling, earlier exits and more waves per SIMD.  The pixels
//     of a lane are handled as packed pairs (v_pk_*_f32: two pixels per VALU issue), a tile row is written as
//     contiguous bytes, and the LDS cost of a record (3 broadcast reads) is amortised over all the lane's pixels.
//   * The tile's sorted list is consumed 64 records per round: lane j gathers record j (one 48 B ggd_splat, three
//     16 B loads; the next round's gather is in flight while the current one is blended), tests the record's
//     alpha >= 1/255 box against the wave's pixel rectangle, and only the survivors are staged, compacted, in LDS.
//   * Per staged record: a wave-level cull in the power domain before any exp, then a branch-free packed update.
//   * No workgroup barrier, no __syncthreads_count: "wave finished" is one wave-uniform ballot.
//   * Backward: the 9 per-Gaussian partial gradients are first summed over the lane's pixels, then over the wave
//     with DPP row shifts (no LDS, no atomics), parked in LDS per staged record, and flushed with ONE global float
//     atomic per (wave, Gaussian, component) instead of one per (pixel, Gaussian, component).
#include "ggd_common.h"

namespace {

constexpr float ALPHA_FLOOR = 1.0f / 255.0f;

struct TileGeom {
  int tile, px0, py;
  uint32_t lo, hi;
};

__device__ __forceinline__ TileGeom tile_geom(int gx, const uint32_t* __restrict__ ranges) {
  TileGeom g;
  g.tile = blockIdx.x;
  const int tx = g.tile % gx, ty = g.tile / gx;
  const int lane = threadIdx.x;
  g.px0 = tx * 16 + (lane & 3) * 4;
  g.py = ty * 16 + (lane >> 2);
  const uint2 r = reinterpret_cast<const uint2*>(ranges)[g.tile];
  g.lo = r.x; g.hi = r.y;
  return g;
}

typedef float f2 __attribute__((ext_vector_type(2)));  // -> v_pk_{add,mul,fma}_f32: two pixels per VALU issue

// power = -1/2 (A dx^2 + C dy^2) - B dx dy for a pixel PAIR, in the project's fixed operation order (same as the
// oracle's gauss_power):  fma( fma(-A/2, dx, -B*dy), dx, ((-C/2)*dy)*dy ).  hA = -A/2; nBdy, hCdy2 are per-lane.
__device__ __forceinline__ f2 gauss_power2(float hA, float nBdy, float hCdy2, f2 dx) {
  const f2 inner = __builtin_elementwise_fma((f2){hA, hA}, dx, (f2){nBdy, nBdy});
  return __builtin_elementwise_fma(inner, dx, (f2){hCdy2, hCdy2});
}

// exp() variants for the blend (GGD_OPT_EXP_MODE).
template <int MODE>
__device__ __forceinline__ float blend_exp(float x) {
  if (MODE == 1) return __expf(x);  // v_exp_f32(x * log2e): ~3 ulp on [-6, 0]
  if (MODE == 2) {                  // 2^(hi) * (1 + lo*ln2): hi = fl(x*log2e), lo = exact product residual + low bits
    const float t = x * 1.44269502162933349609375f;
    float lo = __builtin_fmaf(x, 1.44269502162933349609375f, -t);
    lo = __builtin_fmaf(x, 1.925963033500011e-08f, lo);
    const float e = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(e, lo * 0.693147182464599609375f, e);
  }
  return expf(x);                   // ocml, <= 1 ulp
}

// Two exps per call; MODE 2 is the same arithmetic as blend_exp<2> per component, with the multiplies / FMAs packed.
template <int MODE>
__device__ __forceinline__ f2 blend_exp2v(f2 x) {
  if (MODE == 2) {
    const f2 L2E = {1.44269502162933349609375f, 1.44269502162933349609375f};
    const f2 t = x * L2E;
    f2 lo = __builtin_elementwise_fma(x, L2E, -t);
    lo = __builtin_elementwise_fma(x, (f2){1.925963033500011e-08f, 1.925963033500011e-08f}, lo);
    const f2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return __builtin_elementwise_fma(e, lo * 0.693147182464599609375f, e);
  }
  return (f2){blend_exp<MODE>(x.x), blend_exp<MODE>(x.y)};
}

// Record-level pre-cull, done ONCE per staged record by the lane that gathers it (the per-pixel cull below costs every
// lane of the wave ~15 instructions per record): the set {power >= thr} is the ellipse d^T Q d <= tau2 = -2 thr,
// Q = [[A, B], [B, C]]; its axis-aligned extents are sqrt(tau2 * C / det), sqrt(tau2 * A / det).  If that box misses the
// wave's pixel rectangle no pixel can pass the `power >= thr` test, so the record is never written to LDS.  The extents
// are inflated by eps = 1e-3 + 4e-6 * trace^2 / det (covers the fp32 rounding of the in-loop power evaluation, whose
// relative error grows with the anisotropy of Q) + 0.01 px; indefinite / NaN conics are kept.
__device__ __forceinline__ bool record_box_hits(float x, float y, float A, float B, float C, float thr, float wx0,
                                                float wx1, float wy0, float wy1) {
  const float det = A * C - B * B;
  const float tau2 = -2.0f * thr;
  float ex = __builtin_huge_valf(), ey = __builtin_huge_valf();
  if (det > 0.0f) {
    const float s = tau2 / det;
    const float tr = A + C;
    const float infl = 1.001f + 4e-6f * (tr * tr) / det;
    ex = __builtin_sqrtf(fmaxf(s * C, 0.0f)) * infl + 0.01f;
    ey = __builtin_sqrtf(fmaxf(s * A, 0.0f)) * infl + 0.01f;
  }
  return !(tau2 < 0.0f) && !(x + ex < wx0 || x - ex > wx1 || y + ey < wy0 || y - ey > wy1);
}

// Wave <-> pixel mapping of the blend kernels.  PXL = pixels per lane (horizontally adjacent):
//   PXL = 4: one wave per 16x16 tile (lane = 4x1 pixels, 4 lanes per row, 16 rows);
//   PXL = 2: two waves per tile, each a 16x8 half (lane = 2x1 pixels, 8 lanes per row, 8 rows) -- finer culling and
//            earlier "all pixels done" exits, twice the waves (better latency hiding on small images).
// The halves of one tile are T blocks apart so that both land on the same XCD (block b -> XCD b % 8) and share L2.
template <int PXL>
struct WaveGeom {
  int px0, py;
  uint32_t lo, hi;
  __device__ __forceinline__ WaveGeom(int gx, int T, const uint32_t* __restrict__ ranges, uint32_t capacity = 0xffffffffu) {
    constexpr int LPR = 16 / PXL, ROWS = 64 / LPR;
    const int tile = (int)blockIdx.x % T, sub = (int)blockIdx.x / T;
    const int tx = tile % gx, ty = tile / gx;
    const int lane = threadIdx.x;
    px0 = tx * 16 + (lane % LPR) * PXL;
    py = ty * 16 + sub * ROWS + lane / LPR;
    const uint2 r = reinterpret_cast<const uint2*>(ranges)[tile];
    lo = min(r.x, capacity); hi = min(r.y, capacity);  // capacity < R only in a speculative forward that is retried
  }
};

// Forward blend.  Per staged record:
//   (1) the `power` values of the lane's pixels with packed fp32 math (operation order == the published form),
//   (2) CULL: if no pixel of the wave can reach alpha >= 1/255 -- tested in the power domain against a per-record
//       threshold ln(1/(255*opacity)) lowered by a safety margin, so the decision is exact w.r.t. the float alpha
//       test that follows -- the whole wave skips the record with one ballot, before any exp,
//   (3) the exact per-pixel tests and the blend update.
// A finished pixel gets x = +inf: its power becomes -inf/NaN and it drops out in (2) with no extra instructions.
template <int EXP_MODE, bool CULL, int PXL>
__global__ __launch_bounds__(64) void blend_forward_kernel(int W, int H, int gx, int T,
                                                           const ggd_splat* __restrict__ splat,
                                                           const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ ranges, uint32_t capacity,
                                                           const float* __restrict__ bg,
                                                           float* __restrict__ out_color,
                                                           float* __restrict__ final_T,
                                                           uint32_t* __restrict__ n_contrib,
                                                           unsigned long long* __restrict__ stats) {
  constexpr int NP = PXL / 2;  // pixel pairs per lane
  __shared__ float4 s_rec[64 * 3];
  const int lane = threadIdx.x;
  const WaveGeom<PXL> g(gx, T, ranges, capacity);
  const bool row_in = g.py < H;
  const float INF = __builtin_huge_valf();
  uint32_t st_visited = 0, st_culled = 0, st_lanes = 0, st_pixels = 0;  // wave-uniform debug counters (GGD stats)
  f2 Tr[NP], C[NP][3];  // per pixel PAIR: transmittance, accumulated colour
  uint32_t last[PXL];
  f2 px[NP];  // pixel x coordinates; +inf once the pixel is finished / outside the image
  int alive = 0;
  const float pyf = (float)g.py;
#pragma unroll
  for (int k = 0; k < PXL; ++k) {
    if (k & 1) { Tr[k >> 1] = (f2){1.0f, 1.0f}; C[k >> 1][0] = C[k >> 1][1] = C[k >> 1][2] = (f2){0.0f, 0.0f}; }
    last[k] = 0;
    const bool in = row_in && (g.px0 + k) < W;
    alive += in ? 1 : 0;
    const float x = in ? (float)(g.px0 + k) : INF;
    if (k & 1) px[k >> 1].y = x; else px[k >> 1].x = x;
  }

  // the wave's pixel rectangle (pixel centres), for the record-level pre-cull
  constexpr int LPR = 16 / PXL, ROWS = 64 / LPR;
  const float wx0 = (float)(g.px0 - (lane % LPR) * PXL), wx1 = wx0 + 15.0f;
  const float wy0 = (float)(g.py - lane / LPR), wy1 = wy0 + (float)(ROWS - 1);
  const uint64_t lt_mask = (1ull << lane) - 1ull;

  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
  bool keep = false;
  auto fetch = [&](uint32_t pos) {
    keep = false;
    if (pos < g.hi) {
      const float4* p = reinterpret_cast<const float4*>(splat + list[pos]);
      r0 = p[0]; r1 = p[1]; r2 = p[2];
      // record slot 2 is re-used for the blend: {b, power threshold, contributor index (1-based list position), -}
      const float L = logf(1.0f / (255.0f * r1.y));
      r2.y = CULL ? L - (2e-5f + 1e-6f * fabsf(L)) : -INF;
      r2.z = __uint_as_float(pos - g.lo + 1u);
      keep = CULL ? record_box_hits(r0.x, r0.y, r0.z, r0.w, r1.x, r2.y, wx0, wx1, wy0, wy1) : true;
    }
  };
  fetch(g.lo + lane);
  bool finished = __ballot(alive != 0) == 0ull;
  for (uint32_t base = g.lo; base < g.hi && !finished; base += 64) {
    __syncthreads();  // single-wave block: orders the previous round's LDS reads before this round's writes
    const uint64_t kept = __ballot(keep);
    if (keep) {  // compacted: only records whose box reaches this wave's pixels are staged
      const int slot = __popcll(kept & lt_mask);
      s_rec[slot * 3 + 0] = r0; s_rec[slot * 3 + 1] = r1; s_rec[slot * 3 + 2] = r2;
    }
    st_visited += min(64u, g.hi - base);
    st_culled += min(64u, g.hi - base) - (uint32_t)__popcll(kept);
    fetch(base + 64 + lane);  // next round's gather is in flight while this round is blended
    __syncthreads();
    const int n = __popcll(kept);
    for (int j = 0; j < n; ++j) {
      if ((j & 7) == 0 && __ballot(alive != 0) == 0ull) { finished = true; break; }
      const float4 a = s_rec[j * 3 + 0];  // x, y, conA, conB
      const float4 b = s_rec[j * 3 + 1];  // conC, opacity, r, g
      const float4 c = s_rec[j * 3 + 2];  // b, power threshold, contributor index
      const float dy = a.y - pyf;
      const float hA = -0.5f * a.z, nBdy = (-a.w) * dy, hCdy2 = ((-0.5f * b.x) * dy) * dy;
      const f2 gxx = {a.x, a.x};
      f2 pw[NP];
      bool need[PXL];
      bool lane_need = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const f2 dx = gxx - px[p];
        pw[p] = gauss_power2(hA, nBdy, hCdy2, dx);
        need[2 * p] = pw[p].x >= c.y; need[2 * p + 1] = pw[p].y >= c.y;
        lane_need = lane_need || need[2 * p] || need[2 * p + 1];
      }
      const uint64_t need_lanes = __ballot(lane_need);
      if (need_lanes == 0ull) { st_culled += 1; continue; }
      if (stats) {
        st_lanes += (uint32_t)__popcll(need_lanes);
#pragma unroll
        for (int k = 0; k < PXL; ++k) st_pixels += (uint32_t)__popcll(__ballot(need[k]));
      }
      const uint32_t contributor = __float_as_uint(c.z);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        // branch-free, packed update of a pixel pair.  alpha, the thresholds and T <- T (1 - alpha) are evaluated
        // exactly as published; the colour is accumulated as fma(col, alpha * T, C) (one rounding fewer than
        // (col * alpha) * T + C -- inside the 1e-5 tolerance, three packed FMAs for the pair instead of 18 scalar ops).
        const f2 G = blend_exp2v<EXP_MODE>(pw[p]);
        const f2 av = G * b.y;
        const f2 alpha = {fminf(0.99f, av.x), fminf(0.99f, av.y)};
        const bool live0 = need[2 * p] && !(pw[p].x > 0.0f) && !(alpha.x < ALPHA_FLOOR);
        const bool live1 = need[2 * p + 1] && !(pw[p].y > 0.0f) && !(alpha.y < ALPHA_FLOOR);
        const f2 test_T = Tr[p] * ((f2){1.0f, 1.0f} - alpha);
        const bool low0 = test_T.x < 0.0001f, low1 = test_T.y < 0.0001f;
        const bool stop0 = live0 && low0, stop1 = live1 && low1;
        const bool upd0 = live0 && !low0, upd1 = live1 && !low1;
        const f2 aT = alpha * Tr[p];
        const f2 w = {upd0 ? aT.x : 0.0f, upd1 ? aT.y : 0.0f};
        C[p][0] = __builtin_elementwise_fma((f2){b.z, b.z}, w, C[p][0]);
        C[p][1] = __builtin_elementwise_fma((f2){b.w, b.w}, w, C[p][1]);
        C[p][2] = __builtin_elementwise_fma((f2){c.x, c.x}, w, C[p][2]);
        Tr[p] = (f2){upd0 ? test_T.x : Tr[p].x, upd1 ? test_T.y : Tr[p].y};
        last[2 * p] = upd0 ? contributor : last[2 * p];
        last[2 * p + 1] = upd1 ? contributor : last[2 * p + 1];
        px[p] = (f2){stop0 ? INF : px[p].x, stop1 ? INF : px[p].y};
        alive -= (stop0 ? 1 : 0) + (stop1 ? 1 : 0);
      }
    }
  }

  if (stats && lane == 0) {
    atomicAdd(stats + 0, (unsigned long long)st_visited);
    atomicAdd(stats + 1, (unsigned long long)st_culled);
    atomicAdd(stats + 2, (unsigned long long)st_lanes);
    atomicAdd(stats + 3, (unsigned long long)st_pixels);
    atomicAdd(stats + 4, (unsigned long long)(g.hi - g.lo));
  }
  if (!row_in) return;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)g.py * W + g.px0;
  if (g.px0 + PXL - 1 < W && (W & 3) == 0) {
    if constexpr (PXL == 4) {
      *reinterpret_cast<float4*>(final_T + pix0) = make_float4(Tr[0].x, Tr[0].y, Tr[NP - 1].x, Tr[NP - 1].y);
      *reinterpret_cast<uint4*>(n_contrib + pix0) = make_uint4(last[0], last[1], last[2], last[3]);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float bgc = ch == 0 ? bg0 : (ch == 1 ? bg1 : bg2);
        *reinterpret_cast<float4*>(out_color + ch * HW + pix0) = make_float4(
            C[0][ch].x + Tr[0].x * bgc, C[0][ch].y + Tr[0].y * bgc, C[NP - 1][ch].x + Tr[NP - 1].x * bgc,
            C[NP - 1][ch].y + Tr[NP - 1].y * bgc);
      }
    } else {
      *reinterpret_cast<float2*>(final_T + pix0) = make_float2(Tr[0].x, Tr[0].y);
      *reinterpret_cast<uint2*>(n_contrib + pix0) = make_uint2(last[0], last[1]);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float bgc = ch == 0 ? bg0 : (ch == 1 ? bg1 : bg2);
        *reinterpret_cast<float2*>(out_color + ch * HW + pix0) =
            make_float2(C[0][ch].x + Tr[0].x * bgc, C[0][ch].y + Tr[0].y * bgc);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < PXL; ++k) {
      if (g.px0 + k < W) {
        const float Tk = (k & 1) ? Tr[k >> 1].y : Tr[k >> 1].x;
        const float c0 = (k & 1) ? C[k >> 1][0].y : C[k >> 1][0].x, c1 = (k & 1) ? C[k >> 1][1].y : C[k >> 1][1].x,
                    c2 = (k & 1) ? C[k >> 1][2].y : C[k >> 1][2].x;
        final_T[pix0 + k] = Tk;
        n_contrib[pix0 + k] = last[k];
        out_color[pix0 + k] = c0 + Tk * bg0;
        out_color[HW + pix0 + k] = c1 + Tk * bg1;
        out_color[2 * HW + pix0 + k] = c2 + Tk * bg2;
      }
    }
  }
}

// Backward blend.  Same wave/tile mapping as the forward; records are visited back-to-front from the last position
// any pixel of the tile contributed to.  The pixel body is SELECT-FREE and packed (two pixels per VALU issue):
// a pixel that does not see the record (culled, beyond its n_contrib, alpha < 1/255) gets alpha = G = 0, which makes
// every state update an exact no-op (T/(1-0) = T; the pending (last_alpha, last_color) pair is folded into the
// running colour one record early and then applied with weight 0), so no per-pixel branches or selects on the
// 8 words of recurrence state are needed.  1/(1-alpha) is one v_rcp + one Newton step, shared by both divisions.
template <int EXP_MODE, bool CULL, int PXL>
__global__ __launch_bounds__(64) void blend_backward_kernel(
    int W, int H, int gx, int T_tiles, const ggd_splat* __restrict__ splat, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ ranges, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ grad_acc) {
  __shared__ float4 s_rec[64 * 3];
  __shared__ float s_sum[64 * 9];  // [record slot][component], written by lane 63 only
  const int lane = threadIdx.x;
  constexpr int NP = PXL / 2;  // pixel pairs per lane
  const WaveGeom<PXL> g(gx, T_tiles, ranges);
  const bool row_in = g.py < H;
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)g.py * W + g.px0;

  // per-pixel state, packed as pairs: [p] = pixels (2p, 2p+1) of the lane
  f2 T[NP], nTfin[NP], la[NP], bgdot[NP], acc[NP][3], lastc[NP][3], gpx[NP][3];
  f2 px[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) px[p] = (f2){(float)(g.px0 + 2 * p), (float)(g.px0 + 2 * p + 1)};
  uint32_t lastn[PXL];
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float pyf = (float)g.py;
  uint32_t maxn = 0;
#pragma unroll
  for (int k = 0; k < PXL; ++k) {
    const bool in = row_in && (g.px0 + k) < W;
    const float tf = in ? final_T[pix0 + k] : 0.0f;
    lastn[k] = in ? n_contrib[pix0 + k] : 0u;
    maxn = max(maxn, lastn[k]);
    float gg[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) gg[ch] = in ? dL_dpix[ch * HW + pix0 + k] : 0.0f;
    const float bd = (bg0 * gg[0] + bg1 * gg[1]) + bg2 * gg[2];
    const int p = k >> 1;
    if (k & 1) {
      T[p].y = tf; nTfin[p].y = -tf; bgdot[p].y = bd;
      gpx[p][0].y = gg[0]; gpx[p][1].y = gg[1]; gpx[p][2].y = gg[2];
    } else {
      T[p].x = tf; nTfin[p].x = -tf; bgdot[p].x = bd;
      gpx[p][0].x = gg[0]; gpx[p][1].x = gg[1]; gpx[p][2].x = gg[2];
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    la[p] = (f2){0.0f, 0.0f};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { acc[p][ch] = (f2){0.0f, 0.0f}; lastc[p][ch] = (f2){0.0f, 0.0f}; }
  }
  // wave-uniform number of list positions anyone in the tile contributed to
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxn = max(maxn, (uint32_t)__shfl_xor((int)maxn, d, 64));
  if (maxn == 0) return;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  constexpr int LPR = 16 / PXL, ROWS = 64 / LPR;
  const float twx0 = (float)(g.px0 - (lane % LPR) * PXL), twy0 = (float)(g.py - lane / LPR);  // the wave's pixel rectangle

  uint32_t cend = g.lo + maxn;  // one past the last position that matters
  while (cend > g.lo) {
    const uint32_t cstart = (cend - g.lo > 64u) ? cend - 64u : g.lo;
    const int n = (int)(cend - cstart);
    __syncthreads();
    bool keep = false;
    float4 q0 = make_float4(0, 0, 0, 0), q1 = q0, q2 = q0;
    if (lane < n) {
      const uint32_t my_id = list[cstart + lane];
      const float4* p = reinterpret_cast<const float4*>(splat + my_id);
      q0 = p[0]; q1 = p[1]; q2 = p[2];
      const float L = logf(1.0f / (255.0f * q1.y));  // same conservative power-domain threshold as the forward
      // record slot 2: {b, power threshold, Gaussian id, 0-based list position}
      q2.y = CULL ? L - (2e-5f + 1e-6f * fabsf(L)) : -__builtin_huge_valf();
      q2.z = __uint_as_float(my_id);
      q2.w = __uint_as_float((cstart - g.lo) + (uint32_t)lane);
      keep = CULL ? record_box_hits(q0.x, q0.y, q0.z, q0.w, q1.x, q2.y, twx0, twx0 + 15.0f, twy0, twy0 + (float)(ROWS - 1)) : true;
    }
    const uint64_t kept = __ballot(keep);
    const int nk = __popcll(kept);
    if (keep) {  // compacted, order preserved
      const int slot = __popcll(kept & ((1ull << lane) - 1ull));
      s_rec[slot * 3 + 0] = q0; s_rec[slot * 3 + 1] = q1; s_rec[slot * 3 + 2] = q2;
    }
    __syncthreads();
    uint64_t touched = 0;
    for (int j = nk - 1; j >= 0; --j) {
      const float4 a = s_rec[j * 3 + 0];                                         // x, y, conA, conB
      const float4 b = s_rec[j * 3 + 1];                                         // conC, opacity, r, g
      const float4 c4 = s_rec[j * 3 + 2];                                        // b, power threshold, id, position
      const float2 c2 = make_float2(c4.x, c4.y);
      const uint32_t pos0 = __float_as_uint(c4.w);  // 0-based position in the tile's list
      const float col[3] = {b.z, b.w, c2.x};
      const float dy = a.y - pyf;
      const float hA = -0.5f * a.z, nBdy = (-a.w) * dy, hCdy2 = ((-0.5f * b.x) * dy) * dy;
      const f2 gxx = {a.x, a.x};
      f2 dx[NP], pw[NP];
      bool need[PXL];
      bool lane_need = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        dx[p] = gxx - px[p];
        pw[p] = gauss_power2(hA, nBdy, hCdy2, dx[p]);
        need[2 * p] = (pos0 < lastn[2 * p]) && (pw[p].x >= c2.y);
        need[2 * p + 1] = (pos0 < lastn[2 * p + 1]) && (pw[p].y >= c2.y);
        lane_need = lane_need || need[2 * p] || need[2 * p + 1];
      }
      if (__ballot(lane_need) == 0ull) continue;  // nobody in the wave saw this Gaussian
      f2 sc[3] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, sop = {0.f, 0.f}, scA = {0.f, 0.f}, scB = {0.f, 0.f},
         scC = {0.f, 0.f}, smx = {0.f, 0.f}, smy = {0.f, 0.f};
      bool any = false;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        // exact per-pixel visibility test, then alpha = G = 0 for pixels that do not see the record
        f2 G, alpha;
        {
          const float g0 = blend_exp<EXP_MODE>(pw[p].x), g1 = blend_exp<EXP_MODE>(pw[p].y);
          const float a0 = fminf(0.99f, b.y * g0), a1 = fminf(0.99f, b.y * g1);
          const bool l0 = need[2 * p] && !(pw[p].x > 0.0f) && !(a0 < ALPHA_FLOOR);
          const bool l1 = need[2 * p + 1] && !(pw[p].y > 0.0f) && !(a1 < ALPHA_FLOOR);
          any = any || l0 || l1;
          G = (f2){l0 ? g0 : 0.0f, l1 ? g1 : 0.0f};
          alpha = (f2){l0 ? a0 : 0.0f, l1 ? a1 : 0.0f};
        }
        const f2 om = 1.0f - alpha;
        f2 inv;
        {
          float i0 = __builtin_amdgcn_rcpf(om.x), i1 = __builtin_amdgcn_rcpf(om.y);
          i0 = __builtin_fmaf(i0, __builtin_fmaf(-om.x, i0, 1.0f), i0);
          i1 = __builtin_fmaf(i1, __builtin_fmaf(-om.y, i1, 1.0f), i1);
          inv = (f2){i0, i1};
        }
        // Explicit packed FMAs (the build runs with -ffp-contract=off); the per-record constants -- opacity,
        // -1/2, the 0.5 W / 0.5 H of the pixel-to-NDC map and the minus sign of dG/d(delta) -- are applied once to the
        // reduced sums below instead of to every pixel.
        T[p] = T[p] * inv;
        const f2 dchannel_dcolor = alpha * T[p];
        const f2 oml = 1.0f - la[p];
        f2 dL_dalpha = {0.0f, 0.0f};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          acc[p][ch] = __builtin_elementwise_fma(la[p], lastc[p][ch], oml * acc[p][ch]);
          lastc[p][ch] = (f2){col[ch], col[ch]};
          dL_dalpha = __builtin_elementwise_fma(col[ch] - acc[p][ch], gpx[p][ch], dL_dalpha);
          sc[ch] = __builtin_elementwise_fma(dchannel_dcolor, gpx[p][ch], sc[ch]);
        }
        la[p] = alpha;
        dL_dalpha = __builtin_elementwise_fma(nTfin[p] * inv, bgdot[p], dL_dalpha * T[p]);
        const f2 gdx = G * dx[p], gdy = G * dy;
        const f2 ex = __builtin_elementwise_fma(gdy, (f2){a.w, a.w}, gdx * a.z);   // -dG/d(delta x)
        const f2 ey = __builtin_elementwise_fma(gdx, (f2){a.w, a.w}, gdy * b.x);   // -dG/d(delta y)
        smx = __builtin_elementwise_fma(dL_dalpha, ex, smx);
        smy = __builtin_elementwise_fma(dL_dalpha, ey, smy);
        const f2 wx = gdx * dL_dalpha, wy = gdy * dL_dalpha;
        scA = __builtin_elementwise_fma(wx, dx[p], scA);
        scB = __builtin_elementwise_fma(wx, (f2){dy, dy}, scB);
        scC = __builtin_elementwise_fma(wy, (f2){dy, dy}, scC);
        sop = __builtin_elementwise_fma(G, dL_dalpha, sop);
      }
      if (__ballot(any) != 0ull) {  // wave-uniform: somebody in the tile saw this Gaussian
        touched |= 1ull << j;
        float t[9] = {sc[0].x + sc[0].y, sc[1].x + sc[1].y, sc[2].x + sc[2].y, sop.x + sop.y, scA.x + scA.y,
                      scB.x + scB.y, scC.x + scC.y, smx.x + smx.y, smy.x + smy.y};
        ggd_wave_sum9_to63(t);
        const float nho = -0.5f * b.y;   // d alpha / d G = opacity; d G / d conic = -1/2 G d d^T
        const float t0 = t[0], t1 = t[1], t2 = t[2], t3 = t[3], t4 = nho * t[4], t5 = nho * t[5], t6 = nho * t[6],
                    t7 = (-b.y * ddelx_dx) * t[7], t8 = (-b.y * ddely_dy) * t[8];
        if (lane == 63) {
          float* o = s_sum + j * 9;
          o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = t5; o[6] = t6; o[7] = t7; o[8] = t8;
        }
      }
    }
    __syncthreads();
    if (lane < nk && ((touched >> lane) & 1ull)) {
      const float* o = s_sum + lane * 9;
      const size_t id = __float_as_uint(s_rec[lane * 3 + 2].z);
      // one 48-byte accumulator record per Gaussian (GGD_ACC_*): the 9 atomics of a record land in one cache line
      float* a = grad_acc + GGD_ACC_FLOATS * id;
      atomicAdd(a + GGD_ACC_COLOR + 0, o[0]);
      atomicAdd(a + GGD_ACC_COLOR + 1, o[1]);
      atomicAdd(a + GGD_ACC_COLOR + 2, o[2]);
      atomicAdd(a + GGD_ACC_OPACITY, o[3]);
      atomicAdd(a + GGD_ACC_CONIC + 0, o[4]);
      atomicAdd(a + GGD_ACC_CONIC + 1, o[5]);
      atomicAdd(a + GGD_ACC_CONIC + 2, o[6]);
      atomicAdd(a + GGD_ACC_MEAN2D + 0, o[7]);
      atomicAdd(a + GGD_ACC_MEAN2D + 1, o[8]);
    }
    cend = cstart;
  }
}

}  // namespace

int ggd_launch_blend(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                     const uint32_t* list, const uint32_t* ranges, uint32_t capacity, float* out_color, float* final_T,
                     uint32_t* n_contrib) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE];
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
  const int T = gx * gy;
  const int split = ctx->opt[GGD_OPT_BLEND_SPLIT];
  const bool two = split == 2 || (split == 1 && true);  // auto: two half-tile waves per tile
#define GGD_LAUNCH_FWD(EM, CU)                                                                                      \
  do {                                                                                                              \
    if (two)                                                                                                        \
      hipLaunchKernelGGL((blend_forward_kernel<EM, CU, 2>), dim3(2 * T), dim3(64), 0, s, prm.width, prm.height, gx, \
                         T, splat, list, ranges, capacity, prm.bg, out_color, final_T, n_contrib, ctx->blend_stats); \
    else                                                                                                            \
      hipLaunchKernelGGL((blend_forward_kernel<EM, CU, 4>), dim3(T), dim3(64), 0, s, prm.width, prm.height, gx, T,  \
                         splat, list, ranges, capacity, prm.bg, out_color, final_T, n_contrib, ctx->blend_stats);  \
  } while (0)
  if (cull) {
    if (em == 0) GGD_LAUNCH_FWD(0, true); else if (em == 1) GGD_LAUNCH_FWD(1, true); else GGD_LAUNCH_FWD(2, true);
  } else {
    if (em == 0) GGD_LAUNCH_FWD(0, false); else if (em == 1) GGD_LAUNCH_FWD(1, false); else GGD_LAUNCH_FWD(2, false);
  }
#undef GGD_LAUNCH_FWD
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// auto: below this many tiles the backward runs two waves per tile (measured: see DESIGN.md)
constexpr int GGD_BWD_SPLIT_MAX_TILES = 4096;

int ggd_launch_blend_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                              const uint32_t* list, const uint32_t* ranges, const float* final_T,
                              const uint32_t* n_contrib, const float* dL_dpix, float* grad_acc) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE];
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
  // two waves per tile (16x8 halves) when there are too few tiles to give every SIMD a couple of waves
  const int T = gx * gy;
  const int split = ctx->opt[GGD_OPT_BLEND_SPLIT];
  const bool two = split == 2 || (split == 1 && T < GGD_BWD_SPLIT_MAX_TILES);
#define GGD_LAUNCH_BWD(EM, CU)                                                                                          \
  do {                                                                                                                  \
    if (two)                                                                                                            \
      hipLaunchKernelGGL((blend_backward_kernel<EM, CU, 2>), dim3(2 * T), dim3(64), 0, s, prm.width, prm.height, gx, T, \
                         splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, grad_acc);                           \
    else                                                                                                                \
      hipLaunchKernelGGL((blend_backward_kernel<EM, CU, 4>), dim3(T), dim3(64), 0, s, prm.width, prm.height, gx, T,     \
                         splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, grad_acc);                           \
  } while (0)
  if (cull) {
    if (em == 0) GGD_LAUNCH_BWD(0, true); else if (em == 1) GGD_LAUNCH_BWD(1, true); else GGD_LAUNCH_BWD(2, true);
  } else {
    if (em == 0) GGD_LAUNCH_BWD(0, false); else if (em == 1) GGD_LAUNCH_BWD(1, false); else GGD_LAUNCH_BWD(2, false);
  }
#undef GGD_LAUNCH_BWD
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
