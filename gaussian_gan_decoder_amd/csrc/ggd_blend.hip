// ggd_blend.hip -- stages a9 (forward alpha blend) and a10 (backward blend) for gfx950.
//
// Replaces renderCUDA fwd/bwd inside `_C.rasterize_gaussians{,_backward}` (reached from
// gaussian_splatting/gaussian_renderer/__init__.py:167-175; algorithm restated in SURVEY.md 9.4 / 9.5).
//
// wave64 design (not the 16x16-thread block of the CUDA original):
//   * ONE 64-lane wave owns one 16x16 tile; lane l owns the 4 horizontally adjacent pixels
//     x = 16*tx + 4*(l&3) .. +3 of row y = 16*ty + (l>>2).  4 independent pixel chains per lane hide the
//     exp / LDS latency, a row of the tile is written as 64 contiguous bytes, and the LDS cost of a record
//     (3 x ds_read_b128 broadcast = 12 LDS cycles) is amortised over 4x the VALU work, which keeps the loop
//     VALU-bound instead of LDS-bound.
//   * The tile's sorted list is staged 64 records per round into LDS (lane j gathers record j: one 48 B
//     ggd_splat, three 16 B loads), the next round's gather is issued before the current round is blended.
//   * No workgroup barrier, no __syncthreads_count: "tile finished" is one wave-uniform ballot.
//   * Backward: the 9 per-Gaussian partial gradients are first summed over the lane's 4 pixels, then over the
//     wave with DPP row shifts (no LDS, no atomics), parked in LDS per staged record, and flushed with ONE
//     global float atomic per (tile, Gaussian, component) instead of one per (pixel, Gaussian, component).
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

typedef float f2 __attribute__((ext_vector_type(2)));  // -> v_pk_{add,mul}_f32: two pixels per VALU issue

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

// One wave = one 16x16 tile, 4 horizontally adjacent pixels per lane.  Per staged record:
//   (1) the four `power` values with packed fp32 math (operation order == the algorithm's published form),
//   (2) CULL: if no pixel of the tile can reach alpha >= 1/255 -- tested in the power domain against a per-record
//       threshold ln(1/(255*opacity)) lowered by a safety margin, so the decision is exact w.r.t. the float alpha
//       test that follows -- the whole wave skips the record with one ballot, before any exp,
//   (3) the exact per-pixel tests and the blend update.
// A finished pixel gets x = +inf: its power becomes -inf/NaN and it drops out in (2) with no extra instructions.
template <int EXP_MODE, bool CULL>
__global__ __launch_bounds__(64) void blend_forward_kernel(int W, int H, int gx, const ggd_splat* __restrict__ splat,
                                                           const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ ranges,
                                                           const float* __restrict__ bg,
                                                           float* __restrict__ out_color,
                                                           float* __restrict__ final_T,
                                                           uint32_t* __restrict__ n_contrib) {
  __shared__ float4 s_rec[64 * 3];
  const int lane = threadIdx.x;
  const TileGeom g = tile_geom(gx, ranges);
  const bool row_in = g.py < H;
  const float INF = __builtin_huge_valf();
  float T[4], C[4][3];
  uint32_t last[4];
  f2 pxA, pxB;  // pixel x coordinates (0,1) and (2,3); +inf once the pixel is finished / outside the image
  int alive = 0;
  const float pyf = (float)g.py;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    T[k] = 1.0f; C[k][0] = C[k][1] = C[k][2] = 0.0f;
    last[k] = 0;
    const bool in = row_in && (g.px0 + k) < W;
    alive += in ? 1 : 0;
    const float x = in ? (float)(g.px0 + k) : INF;
    if (k == 0) pxA.x = x; else if (k == 1) pxA.y = x; else if (k == 2) pxB.x = x; else pxB.y = x;
  }

  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
  auto fetch = [&](uint32_t pos) {
    if (pos < g.hi) {
      const float4* p = reinterpret_cast<const float4*>(splat + list[pos]);
      r0 = p[0]; r1 = p[1]; r2 = p[2];
      // record slot 2 is re-used for the blend: {b, power threshold, -, -}
      const float L = logf(1.0f / (255.0f * r1.y));
      r2.y = CULL ? L - (2e-5f + 1e-6f * fabsf(L)) : -INF;
    }
  };
  fetch(g.lo + lane);
  bool finished = __ballot(alive != 0) == 0ull;
  for (uint32_t base = g.lo; base < g.hi && !finished; base += 64) {
    __syncthreads();  // single-wave block: orders the previous round's LDS reads before this round's writes
    s_rec[lane * 3 + 0] = r0; s_rec[lane * 3 + 1] = r1; s_rec[lane * 3 + 2] = r2;
    fetch(base + 64 + lane);  // next round's gather is in flight while this round is blended
    __syncthreads();
    const int n = (int)min(64u, g.hi - base);
    const uint32_t cbase = base - g.lo;
    for (int j = 0; j < n; ++j) {
      if ((j & 7) == 0 && __ballot(alive != 0) == 0ull) { finished = true; break; }
      const float4 a = s_rec[j * 3 + 0];  // x, y, conA, conB
      const float4 b = s_rec[j * 3 + 1];  // conC, opacity, r, g
      const float2 c = *reinterpret_cast<const float2*>(&s_rec[j * 3 + 2]);  // b, power threshold
      const float dy = a.y - pyf;
      const float cdy2 = b.x * dy * dy;
      const f2 gxx = {a.x, a.x};
      const f2 dxA = gxx - pxA, dxB = gxx - pxB;
      const f2 powA = -0.5f * (a.z * dxA * dxA + cdy2) - a.w * dxA * dy;
      const f2 powB = -0.5f * (a.z * dxB * dxB + cdy2) - a.w * dxB * dy;
      const bool n0 = powA.x >= c.y, n1 = powA.y >= c.y, n2 = powB.x >= c.y, n3 = powB.y >= c.y;
      if (__ballot(n0 || n1 || n2 || n3) == 0ull) continue;
      const uint32_t contributor = cbase + (uint32_t)j + 1u;
#define GGD_PIXEL(k, POWER, NEED, PX)                                                        \
      {                                                                                      \
        const float power = POWER;                                                           \
        const float alpha = fminf(0.99f, b.y * blend_exp<EXP_MODE>(power));                  \
        const bool live = (NEED) && !(power > 0.0f) && !(alpha < ALPHA_FLOOR);               \
        const float test_T = T[k] * (1.0f - alpha);                                          \
        if (live) {                                                                          \
          if (test_T < 0.0001f) {                                                            \
            PX = INF;                                                                        \
            alive -= 1;                                                                      \
          } else {                                                                           \
            C[k][0] += b.z * alpha * T[k];                                                   \
            C[k][1] += b.w * alpha * T[k];                                                   \
            C[k][2] += c.x * alpha * T[k];                                                   \
            T[k] = test_T;                                                                   \
            last[k] = contributor;                                                           \
          }                                                                                  \
        }                                                                                    \
      }
      GGD_PIXEL(0, powA.x, n0, pxA.x)
      GGD_PIXEL(1, powA.y, n1, pxA.y)
      GGD_PIXEL(2, powB.x, n2, pxB.x)
      GGD_PIXEL(3, powB.y, n3, pxB.y)
#undef GGD_PIXEL
    }
  }

  if (!row_in) return;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)g.py * W + g.px0;
  if (g.px0 + 3 < W && (W & 3) == 0) {
    *reinterpret_cast<float4*>(final_T + pix0) = make_float4(T[0], T[1], T[2], T[3]);
    *reinterpret_cast<uint4*>(n_contrib + pix0) = make_uint4(last[0], last[1], last[2], last[3]);
    *reinterpret_cast<float4*>(out_color + pix0) =
        make_float4(C[0][0] + T[0] * bg0, C[1][0] + T[1] * bg0, C[2][0] + T[2] * bg0, C[3][0] + T[3] * bg0);
    *reinterpret_cast<float4*>(out_color + HW + pix0) =
        make_float4(C[0][1] + T[0] * bg1, C[1][1] + T[1] * bg1, C[2][1] + T[2] * bg1, C[3][1] + T[3] * bg1);
    *reinterpret_cast<float4*>(out_color + 2 * HW + pix0) =
        make_float4(C[0][2] + T[0] * bg2, C[1][2] + T[1] * bg2, C[2][2] + T[2] * bg2, C[3][2] + T[3] * bg2);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (g.px0 + k < W) {
        final_T[pix0 + k] = T[k];
        n_contrib[pix0 + k] = last[k];
        out_color[pix0 + k] = C[k][0] + T[k] * bg0;
        out_color[HW + pix0 + k] = C[k][1] + T[k] * bg1;
        out_color[2 * HW + pix0 + k] = C[k][2] + T[k] * bg2;
      }
    }
  }
}

template <int EXP_MODE, bool CULL>
__global__ __launch_bounds__(64) void blend_backward_kernel(
    int W, int H, int gx, const ggd_splat* __restrict__ splat, const uint32_t* __restrict__ list,
    const uint32_t* __restrict__ ranges, const float* __restrict__ bg, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D,
    float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors) {
  __shared__ float4 s_rec[64 * 3];
  __shared__ float s_sum[64 * 9];  // [record slot][component], written by lane 63 only
  const int lane = threadIdx.x;
  const TileGeom g = tile_geom(gx, ranges);
  const bool row_in = g.py < H;
  const size_t HW = (size_t)H * W;
  const size_t pix0 = (size_t)g.py * W + g.px0;

  float T[4], Tfin[4], gpx[4][3], acc[4][3], lastc[4][3], last_alpha[4], bgdot[4];
  const f2 pxA = {(float)g.px0, (float)(g.px0 + 1)}, pxB = {(float)(g.px0 + 2), (float)(g.px0 + 3)};
  uint32_t lastn[4];
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float pyf = (float)g.py;
  uint32_t maxn = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = row_in && (g.px0 + k) < W;
    Tfin[k] = in ? final_T[pix0 + k] : 0.0f;
    T[k] = Tfin[k];
    lastn[k] = in ? n_contrib[pix0 + k] : 0u;
    maxn = max(maxn, lastn[k]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      gpx[k][ch] = in ? dL_dpix[ch * HW + pix0 + k] : 0.0f;
      acc[k][ch] = 0.0f; lastc[k][ch] = 0.0f;
    }
    last_alpha[k] = 0.0f;
    bgdot[k] = (bg0 * gpx[k][0] + bg1 * gpx[k][1]) + bg2 * gpx[k][2];
  }
  // wave-uniform number of list positions anyone in the tile contributed to
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) maxn = max(maxn, (uint32_t)__shfl_xor((int)maxn, d, 64));
  if (maxn == 0) return;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

  uint32_t cend = g.lo + maxn;  // one past the last position that matters
  while (cend > g.lo) {
    const uint32_t cstart = (cend - g.lo > 64u) ? cend - 64u : g.lo;
    const int n = (int)(cend - cstart);
    __syncthreads();
    uint32_t my_id = 0;
    if (lane < n) {
      my_id = list[cstart + lane];
      const float4* p = reinterpret_cast<const float4*>(splat + my_id);
      const float4 q1 = p[1];
      float4 q2 = p[2];
      const float L = logf(1.0f / (255.0f * q1.y));  // same conservative power-domain threshold as the forward
      q2.y = CULL ? L - (2e-5f + 1e-6f * fabsf(L)) : -__builtin_huge_valf();
      s_rec[lane * 3 + 0] = p[0]; s_rec[lane * 3 + 1] = q1; s_rec[lane * 3 + 2] = q2;
    }
    __syncthreads();
    uint64_t touched = 0;
    for (int j = n - 1; j >= 0; --j) {
      const uint32_t pos0 = (cstart - g.lo) + (uint32_t)j;  // 0-based position in the tile's list
      const float4 a = s_rec[j * 3 + 0];
      const float4 b = s_rec[j * 3 + 1];
      const float2 c2 = *reinterpret_cast<const float2*>(&s_rec[j * 3 + 2]);  // b, power threshold
      const float col[3] = {b.z, b.w, c2.x};
      const float dy = a.y - pyf;
      const float cdy2 = b.x * dy * dy;
      const f2 gxx = {a.x, a.x};
      const f2 dxA = gxx - pxA, dxB = gxx - pxB;
      const f2 powA = -0.5f * (a.z * dxA * dxA + cdy2) - a.w * dxA * dy;
      const f2 powB = -0.5f * (a.z * dxB * dxB + cdy2) - a.w * dxB * dy;
      const float dxs[4] = {dxA.x, dxA.y, dxB.x, dxB.y};
      const float pows[4] = {powA.x, powA.y, powB.x, powB.y};
      bool need[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) need[k] = (pos0 < lastn[k]) && (pows[k] >= c2.y);
      if (__ballot(need[0] || need[1] || need[2] || need[3]) == 0ull) continue;  // nobody in the tile saw it
      float s_col[3] = {0.f, 0.f, 0.f}, s_op = 0.f, s_cA = 0.f, s_cB = 0.f, s_cC = 0.f, s_mx = 0.f, s_my = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dx = dxs[k];
        const float power = pows[k];
        const float G = blend_exp<EXP_MODE>(power);
        const float alpha = fminf(0.99f, b.y * G);
        const bool live = need[k] && !(power > 0.0f) && !(alpha < ALPHA_FLOOR);
        if (live) {
          any = true;
          T[k] = T[k] / (1.0f - alpha);
          const float dchannel_dcolor = alpha * T[k];
          float dL_dalpha = 0.0f;
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            acc[k][ch] = last_alpha[k] * lastc[k][ch] + (1.0f - last_alpha[k]) * acc[k][ch];
            lastc[k][ch] = col[ch];
            dL_dalpha += (col[ch] - acc[k][ch]) * gpx[k][ch];
            s_col[ch] += dchannel_dcolor * gpx[k][ch];
          }
          dL_dalpha *= T[k];
          last_alpha[k] = alpha;
          dL_dalpha += (-Tfin[k] / (1.0f - alpha)) * bgdot[k];
          const float dL_dG = b.y * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * a.z - gdy * a.w;
          const float dG_ddely = -gdy * b.x - gdx * a.w;
          s_mx += dL_dG * dG_ddelx * ddelx_dx;
          s_my += dL_dG * dG_ddely * ddely_dy;
          s_cA += -0.5f * gdx * dx * dL_dG;
          s_cB += -0.5f * gdx * dy * dL_dG;
          s_cC += -0.5f * gdy * dy * dL_dG;
          s_op += G * dL_dalpha;
        }
      }
      if (__ballot(any) != 0ull) {  // wave-uniform: somebody in the tile saw this Gaussian
        touched |= 1ull << j;
        const float t0 = ggd_wave_sum_to63(s_col[0]), t1 = ggd_wave_sum_to63(s_col[1]),
                    t2 = ggd_wave_sum_to63(s_col[2]), t3 = ggd_wave_sum_to63(s_op),
                    t4 = ggd_wave_sum_to63(s_cA), t5 = ggd_wave_sum_to63(s_cB), t6 = ggd_wave_sum_to63(s_cC),
                    t7 = ggd_wave_sum_to63(s_mx), t8 = ggd_wave_sum_to63(s_my);
        if (lane == 63) {
          float* o = s_sum + j * 9;
          o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = t5; o[6] = t6; o[7] = t7; o[8] = t8;
        }
      }
    }
    __syncthreads();
    if (lane < n && ((touched >> lane) & 1ull)) {
      const float* o = s_sum + lane * 9;
      const size_t id = my_id;
      atomicAdd(dL_dcolors + 3 * id + 0, o[0]);
      atomicAdd(dL_dcolors + 3 * id + 1, o[1]);
      atomicAdd(dL_dcolors + 3 * id + 2, o[2]);
      atomicAdd(dL_dopacity + id, o[3]);
      atomicAdd(dL_dconic + 4 * id + 0, o[4]);
      atomicAdd(dL_dconic + 4 * id + 1, o[5]);
      atomicAdd(dL_dconic + 4 * id + 2, o[6]);
      atomicAdd(dL_dmean2D + 3 * id + 0, o[7]);
      atomicAdd(dL_dmean2D + 3 * id + 1, o[8]);
    }
    cend = cstart;
  }
}

}  // namespace

int ggd_launch_blend(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                     const uint32_t* list, const uint32_t* ranges, float* out_color, float* final_T,
                     uint32_t* n_contrib) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE];
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
#define GGD_LAUNCH_FWD(EM, CU)                                                                                   \
  hipLaunchKernelGGL((blend_forward_kernel<EM, CU>), dim3(gx * gy), dim3(64), 0, s, prm.width, prm.height, gx,  \
                     splat, list, ranges, prm.bg, out_color, final_T, n_contrib)
  if (cull) {
    if (em == 0) GGD_LAUNCH_FWD(0, true); else if (em == 1) GGD_LAUNCH_FWD(1, true); else GGD_LAUNCH_FWD(2, true);
  } else {
    if (em == 0) GGD_LAUNCH_FWD(0, false); else if (em == 1) GGD_LAUNCH_FWD(1, false); else GGD_LAUNCH_FWD(2, false);
  }
#undef GGD_LAUNCH_FWD
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int ggd_launch_blend_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                              const uint32_t* list, const uint32_t* ranges, const float* final_T,
                              const uint32_t* n_contrib, const float* dL_dpix, float* dL_dmean2D,
                              float* dL_dconic, float* dL_dopacity, float* dL_dcolors) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  const int em = ctx->opt[GGD_OPT_EXP_MODE];
  const bool cull = ctx->opt[GGD_OPT_BLEND_CULL] != 0;
#define GGD_LAUNCH_BWD(EM, CU)                                                                                    \
  hipLaunchKernelGGL((blend_backward_kernel<EM, CU>), dim3(gx * gy), dim3(64), 0, s, prm.width, prm.height, gx,  \
                     splat, list, ranges, prm.bg, final_T, n_contrib, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, \
                     dL_dcolors)
  if (cull) {
    if (em == 0) GGD_LAUNCH_BWD(0, true); else if (em == 1) GGD_LAUNCH_BWD(1, true); else GGD_LAUNCH_BWD(2, true);
  } else {
    if (em == 0) GGD_LAUNCH_BWD(0, false); else if (em == 1) GGD_LAUNCH_BWD(1, false); else GGD_LAUNCH_BWD(2, false);
  }
#undef GGD_LAUNCH_BWD
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
