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
  float pxf[4], T[4], C[4][3];
  uint32_t last[4];
  bool done[4];
  const float pyf = (float)g.py;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pxf[k] = (float)(g.px0 + k);
    T[k] = 1.0f; C[k][0] = C[k][1] = C[k][2] = 0.0f;
    last[k] = 0;
    done[k] = !(row_in && (g.px0 + k) < W);
  }

  float4 r0 = make_float4(0, 0, 0, 0), r1 = r0, r2 = r0;
  auto fetch = [&](uint32_t pos) {
    if (pos < g.hi) {
      const float4* p = reinterpret_cast<const float4*>(splat + list[pos]);
      r0 = p[0]; r1 = p[1]; r2 = p[2];
    }
  };
  fetch(g.lo + lane);
  bool finished = false;
  for (uint32_t base = g.lo; base < g.hi && !finished; base += 64) {
    __syncthreads();  // single-wave block: orders the previous round's LDS reads before this round's writes
    s_rec[lane * 3 + 0] = r0; s_rec[lane * 3 + 1] = r1; s_rec[lane * 3 + 2] = r2;
    fetch(base + 64 + lane);  // next round's gather is in flight while this round is blended
    __syncthreads();
    const int n = (int)min(64u, g.hi - base);
    const uint32_t cbase = base - g.lo;
    for (int j = 0; j < n; ++j) {
      if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) { finished = true; break; }
      const float4 a = s_rec[j * 3 + 0];  // x, y, conA, conB
      const float4 b = s_rec[j * 3 + 1];  // conC, opacity, r, g
      const float cb = s_rec[j * 3 + 2].x;
      const uint32_t contributor = cbase + (uint32_t)j + 1u;
      const float dy = a.y - pyf;
      const float cdy2 = b.x * dy * dy;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dx = a.x - pxf[k];
        const float power = -0.5f * (a.z * dx * dx + cdy2) - a.w * dx * dy;
        const float alpha = fminf(0.99f, b.y * expf(power));
        const bool live = !done[k] && !(power > 0.0f) && !(alpha < ALPHA_FLOOR);
        const float test_T = T[k] * (1.0f - alpha);
        const bool stop = live && (test_T < 0.0001f);
        done[k] = done[k] || stop;
        if (live && !stop) {
          C[k][0] += b.z * alpha * T[k];
          C[k][1] += b.w * alpha * T[k];
          C[k][2] += cb * alpha * T[k];
          T[k] = test_T;
          last[k] = contributor;
        }
      }
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

  float pxf[4], T[4], Tfin[4], gpx[4][3], acc[4][3], lastc[4][3], last_alpha[4], bgdot[4];
  uint32_t lastn[4];
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const float pyf = (float)g.py;
  uint32_t maxn = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool in = row_in && (g.px0 + k) < W;
    pxf[k] = (float)(g.px0 + k);
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
      s_rec[lane * 3 + 0] = p[0]; s_rec[lane * 3 + 1] = p[1]; s_rec[lane * 3 + 2] = p[2];
    }
    __syncthreads();
    uint64_t touched = 0;
    for (int j = n - 1; j >= 0; --j) {
      const uint32_t pos0 = (cstart - g.lo) + (uint32_t)j;  // 0-based position in the tile's list
      const float4 a = s_rec[j * 3 + 0];
      const float4 b = s_rec[j * 3 + 1];
      const float cb = s_rec[j * 3 + 2].x;
      const float col[3] = {b.z, b.w, cb};
      const float dy = a.y - pyf;
      const float cdy2 = b.x * dy * dy;
      float s_col[3] = {0.f, 0.f, 0.f}, s_op = 0.f, s_cA = 0.f, s_cB = 0.f, s_cC = 0.f, s_mx = 0.f, s_my = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dx = a.x - pxf[k];
        const float power = -0.5f * (a.z * dx * dx + cdy2) - a.w * dx * dy;
        const float G = expf(power);
        const float alpha = fminf(0.99f, b.y * G);
        const bool live = (pos0 < lastn[k]) && !(power > 0.0f) && !(alpha < ALPHA_FLOOR);
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
  hipLaunchKernelGGL(blend_forward_kernel, dim3(gx * gy), dim3(64), 0, s, prm.width, prm.height, gx, splat, list,
                     ranges, prm.bg, out_color, final_T, n_contrib);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int ggd_launch_blend_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const ggd_splat* splat,
                              const uint32_t* list, const uint32_t* ranges, const float* final_T,
                              const uint32_t* n_contrib, const float* dL_dpix, float* dL_dmean2D,
                              float* dL_dconic, float* dL_dopacity, float* dL_dcolors) {
  const int gx = (prm.width + 15) / 16, gy = (prm.height + 15) / 16;
  if (gx * gy == 0) return GGD_OK;
  hipLaunchKernelGGL(blend_backward_kernel, dim3(gx * gy), dim3(64), 0, s, prm.width, prm.height, gx, splat, list,
                     ranges, prm.bg, final_T, n_contrib, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolors);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
