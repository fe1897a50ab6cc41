// ggd_preprocess.hip -- stage a4 (per-Gaussian forward) and a12 (mark_visible) for gfx950.
//
// Replaces the per-Gaussian half of `_C.rasterize_gaussians` that the reference reaches through
// gaussian_splatting/gaussian_renderer/__init__.py:167-175 (source not in the reference tree; algorithm restated
// in SURVEY.md section 9.2).  One lane per Gaussian, 256-lane workgroups; every input array is read exactly once
// with per-lane vector loads over contiguous addresses (a wave covers 64 consecutive Gaussians = one contiguous
// 768 B / 1 KiB span per array), and the outputs the blend needs are packed into ONE 48-byte record per Gaussian
// (ggd_splat) so that the per-tile gather later touches a single 64 B-aligned-ish record instead of four arrays.
// HBM-bound: 56 B in (degree 0) + 56 B out per visible Gaussian, 56 B in + 8 B out per culled one.
#include "ggd_math.h"

namespace {
using namespace ggdm;

// SHVEC (more than the band-0 coefficient per channel, rows a multiple of 16 bytes: M = 4, 8, 12, 16): a visible Gaussian's
// 3 M coefficients are fetched as 3 M / 4 dwordx4 loads into registers instead of 3 M lone words -- each such load
// instruction touches 64 different rows whatever its width (1 M Gaussians, M = 16: preprocess 76 -> see DESIGN.md).
// FOLD (ggd_fold, single-call forward on the tile-binning path): the workgroup also (a) clears its share of the OTHER
// control block for the next frame, (b) adds the four digit counts of its kept depth keys to replica blockIdx % REPS of the
// depth sort's histograms (LDS histogram first; bins that stayed empty cost nothing), (c) stores {sum of tiles_touched, kept
// keys} of its 256 points for the offsets scan -- the sort's histogram launch and the scan's first step disappear.
template <bool SHVEC, bool FOLD>
__global__ __launch_bounds__(256) void preprocess_kernel(
    int P, int M, int deg, int W, int H, float tanfovx, float tanfovy, float fx, float fy, float mod, int prefiltered, int raw,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos_p,
    const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ opacities, const float* __restrict__ scales, const float* __restrict__ rotations,
    const float* __restrict__ cov3D_precomp, ggd_splat* __restrict__ splat, uint32_t* __restrict__ tiles_touched,
    uint8_t* __restrict__ clamped, int32_t* __restrict__ radii, uint32_t* __restrict__ depth_keys,
    uint2* __restrict__ rect, uint32_t* __restrict__ trap_flag, uint32_t* __restrict__ zero_ptr, int zero_words,
    ggd_fold fold) {
  __shared__ uint32_t s_hist[FOLD ? GGD_FOLD_REP_STRIDE : 1];
  __shared__ uint32_t s_red[FOLD ? 17 : 1];   // per wave: sum of tiles, kept keys, ~min key, max key; [16]: keys outside the window
  __shared__ int s_rowdiff[FOLD ? 65 : 1];
  if constexpr (FOLD) {
    for (uint32_t z = blockIdx.x * 256 + threadIdx.x; z < fold.clear_words; z += gridDim.x * 256) fold.clear[z] = 0u;
    for (int b = threadIdx.x; b < GGD_FOLD_REP_STRIDE; b += 256) s_hist[b] = 0u;
    if (threadIdx.x < 65) s_rowdiff[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_red[16] = 0u;
    __syncthreads();
  } else {
    // first kernel of a frame: its first workgroups also clear the depth sort's control block (no memset launch there)
    const int zb = min(8, (int)gridDim.x);
    if ((int)blockIdx.x < zb)
      for (int z = blockIdx.x * 256 + threadIdx.x; z < zero_words; z += zb * 256) zero_ptr[z] = 0u;
  }
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool in_range = i < P;
  if (!FOLD && !in_range) return;
  int irad = 0;
  uint32_t ntiles = 0, rect_rows = 0;   // rect_rows = miny | maxy << 16 of a visible Gaussian
  bool visible = false;
  float depth = 0.0f;
  if (in_range) {
  const Mat16 V = load_mat(view);
  const Mat16 PV = load_mat(proj);

  // Every per-Gaussian input is requested HERE, in one go, whatever the culling tests below decide: behind the tests (position
  // -> depth test -> scale / rotation -> rect test -> colour -> opacity) a wave paid four dependent trips to memory and lived
  // 10 us, 60 % of it waiting.  The price is 16 B of colour + opacity for a Gaussian whose rect turns out empty.  Inline asm,
  // because the compiler sinks plain loads back behind the tests (their only uses); an array this call does not have is
  // replaced by the 64-byte view matrix so that no load sits behind a branch.  The hardware completes loads in order, but
  // stores issued earlier (the control-block clear) share the counter: one wait for everything.
  typedef float f3v __attribute__((ext_vector_type(3)));
  typedef float f4v __attribute__((ext_vector_type(4)));
  f3v p_v, s_v, c_v;
  f4v q_v;
  float opac_in;
  {
    const float* pp = means3D + 3 * (size_t)i;
    const float* sp = cov3D_precomp ? view : scales + 3 * (size_t)i;
    const float* qp = cov3D_precomp ? view : rotations + 4 * (size_t)i;
    const float* cp = colors_precomp ? colors_precomp + 3 * (size_t)i : (shs ? shs + (size_t)i * M * 3 : view);   // colour, or SH band 0
    const float* op = opacities + i;
    asm volatile("global_load_dwordx3 %0, %5, off\n\t"
                 "global_load_dwordx3 %1, %6, off\n\t"
                 "global_load_dwordx4 %2, %7, off\n\t"
                 "global_load_dwordx3 %3, %8, off\n\t"
                 "global_load_dword %4, %9, off\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(p_v), "=&v"(s_v), "=&v"(q_v), "=&v"(c_v), "=&v"(opac_in)
                 : "v"(pp), "v"(sp), "v"(qp), "v"(cp), "v"(op)
                 : "memory");
  }
  const float p[3] = {p_v.x, p_v.y, p_v.z};
  float s3[3] = {s_v.x, s_v.y, s_v.z};
  float4 q = make_float4(q_v.x, q_v.y, q_v.z, q_v.w);
  const float rgb_in[3] = {c_v.x, c_v.y, c_v.z};
  float t[3];
  t[0] = V.m[0] * p[0] + V.m[4] * p[1] + V.m[8] * p[2] + V.m[12];
  t[1] = V.m[1] * p[0] + V.m[5] * p[1] + V.m[9] * p[2] + V.m[13];
  t[2] = V.m[2] * p[0] + V.m[6] * p[1] + V.m[10] * p[2] + V.m[14];
  depth = t[2];

  ggd_splat out;
  uint32_t clamp_bits = 0;
  uint2 rect_out = make_uint2(0u, 0u);

  if (t[2] > 0.2f) {
    float h[4];
    h[0] = PV.m[0] * p[0] + PV.m[4] * p[1] + PV.m[8] * p[2] + PV.m[12];
    h[1] = PV.m[1] * p[0] + PV.m[5] * p[1] + PV.m[9] * p[2] + PV.m[13];
    h[3] = PV.m[3] * p[0] + PV.m[7] * p[1] + PV.m[11] * p[2] + PV.m[15];
    const float pw = 1.0f / (h[3] + 0.0000001f);
    const float ndcx = h[0] * pw, ndcy = h[1] * pw;

    float c6[6];
    if (cov3D_precomp) {
#pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * (size_t)i + k];
    } else {
      if (raw) {
        float nrm;
        s3[0] = expf(s3[0]); s3[1] = expf(s3[1]); s3[2] = expf(s3[2]);
        q = act_normalize(q, nrm);
      }
      cov3d_from_scale_rot(s3, mod, q, c6);
    }
    float abc[3], Tm[2][3], tcl[3];
    bool clx, cly;
    ewa_cov2d(t, fx, fy, tanfovx, tanfovy, c6, V, abc, Tm, tcl, clx, cly);
    const float a = abc[0] + 0.3f, b = abc[1], c = abc[2] + 0.3f;
    const float det = a * c - b * b;
    if (det != 0.0f) {
      const float det_inv = 1.0f / det;
      const float mid = 0.5f * (a + c);
      const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
      const float lambda1 = mid + disc, lambda2 = mid - disc;
      const float my_radius = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
      const float px = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
      const float py = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
      const int r_i = (int)my_radius;
      int minx, miny, maxx, maxy;
      const int gx = (W + 15) / 16, gy = (H + 15) / 16;
      const int area = ggd_tile_rect(px, py, r_i, gx, gy, minx, miny, maxx, maxy);
      if (area != 0) {
        visible = true;
        irad = r_i;
        ntiles = (uint32_t)area;
        float rgb[3];
        if (colors_precomp) {
          rgb[0] = rgb_in[0]; rgb[1] = rgb_in[1]; rgb[2] = rgb_in[2];
        } else if (deg == 0) {   // band 0 only (what the decoder's renderer passes): the coefficients arrived with the rest
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float res = SH_C0 * rgb_in[c] + 0.5f;
            if (res < 0.0f) clamp_bits |= (1u << c);
            rgb[c] = fmaxf(res, 0.0f);
          }
        } else {
          const float campos[3] = {campos_p[0], campos_p[1], campos_p[2]};
          if constexpr (SHVEC) {
            float shr[48];
            const float4* src = reinterpret_cast<const float4*>(shs + (size_t)i * M * 3);
            const int nq = (3 * M) >> 2;
#pragma unroll
            for (int qd = 0; qd < 12; ++qd) {
              float4 v = make_float4(0, 0, 0, 0);
              if (qd < nq) v = src[qd];
              shr[4 * qd] = v.x; shr[4 * qd + 1] = v.y; shr[4 * qd + 2] = v.z; shr[4 * qd + 3] = v.w;
            }
            sh_to_rgb(deg, shr, p, campos, rgb, clamp_bits);
          } else {
            sh_to_rgb(deg, shs + (size_t)i * M * 3, p, campos, rgb, clamp_bits);
          }
        }
        rect_out = make_uint2((uint32_t)minx | ((uint32_t)maxx << 16), (uint32_t)miny | ((uint32_t)maxy << 16));
        const float conA = c * det_inv, conB = -b * det_inv, conC = a * det_inv;   // the published conic
        const float opac = raw ? act_sigmoid(opac_in) : opac_in;
        out.x = px; out.y = py;
        out.hA = -0.5f * conA; out.nB = -conB; out.hC = -0.5f * conC;   // exact rescalings (see ggd_raster.h)
        out.opacity = opac;
        out.r = rgb[0]; out.g = rgb[1]; out.b = rgb[2];
        // Blend-side culling data, once per Gaussian (the blend kernels used to derive it once per (tile, Gaussian)):
        // alpha = opacity * exp(power) >= 1/255  <=>  power >= L = ln(1 / (255 opacity)); thr sits a safety margin below L so
        // that the decision is exact w.r.t. the float alpha test that follows.  {power >= thr} is the ellipse
        // d^T Q d <= tau2 = -2 thr, Q = [[A, B], [B, C]]; its axis-aligned half extents are sqrt(tau2 C / det),
        // sqrt(tau2 A / det), inflated by 1.001 + 4e-6 trace^2 / det (the fp32 rounding of the in-loop power evaluation
        // grows with the anisotropy of Q) + 0.01 px.  Indefinite / NaN conics get +inf (never culled by the box);
        // thr > 0 (opacity < 1/255) can never be reached by power <= 0: the box is empty (extents -inf).
        // (hardware log / reciprocal / square root here, not the correctly rounded forms the bit-exact outputs above need: this
        // block only has to be conservative, its 1-2 ulp are three orders below the margins -- 28.2 -> 25.9 us at 1 M points)
        const float L = -__logf(255.0f * opac);
        float thr = L - (2e-5f + 1e-6f * fabsf(L));
        const float cdet = conA * conC - conB * conB;
        const float tau2 = -2.0f * thr;
        float ex = __builtin_huge_valf(), ey = __builtin_huge_valf();
        if (cdet > 0.0f) {
          const float rc = __builtin_amdgcn_rcpf(cdet);
          const float sdet = tau2 * rc;
          const float tr = conA + conC;
          const float infl = 1.001f + 4e-6f * (tr * tr) * rc;
          ex = __builtin_amdgcn_sqrtf(fmaxf(sdet * conC, 0.0f)) * infl + 0.01f;
          ey = __builtin_amdgcn_sqrtf(fmaxf(sdet * conA, 0.0f)) * infl + 0.01f;
        }
        if (tau2 < 0.0f) { ex = -__builtin_huge_valf(); ey = -__builtin_huge_valf(); }
        // opacity <= 0 (or NaN): L is +inf / NaN and thr = inf - inf = NaN.  alpha = opacity * G <= 0 < 1/255 for every
        // pixel, so the record can never contribute: say so explicitly (threshold +inf, empty box) instead of relying
        // on how the three blend kernels' comparisons treat a NaN threshold
        if (!(opac > 0.0f) || !(L < __builtin_huge_valf())) {
          thr = __builtin_huge_valf(); ex = -__builtin_huge_valf(); ey = -__builtin_huge_valf();
        }
        out.thr = thr; out.ex = ex; out.ey = ey;
      }
    }
  } else if (prefiltered) {
    atomicOr(trap_flag, 1u);  // upstream traps here; we report GGD_E_PREFILTER instead
  }

  radii[i] = irad;
  tiles_touched[i] = ntiles;
  depth_keys[i] = visible ? __float_as_uint(t[2]) : 0xFFFFFFFFu;
  rect[i] = rect_out;
  rect_rows = rect_out.y;
  if (clamped) clamped[i] = (uint8_t)clamp_bits;
  if (visible) {
    float4* dst = reinterpret_cast<float4*>(splat + i);
    const float4* src = reinterpret_cast<const float4*>(&out);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
  }
  }  // in_range
  if constexpr (FOLD) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t key = __float_as_uint(depth);
    const uint64_t act = __ballot(visible);
    if (act != 0ull) {
      // (as sort_global_hist_kernel: the two high bytes -- sign / exponent / leading mantissa bits of a depth -- are usually
      // shared by the whole wave: one lane adds the count instead of 64 conflicting LDS atomics)
      const int leader = __builtin_ctzll(act);
      if (fold.msd) {   // two-launch sort: the bucket inside the key window (never shared by a wave) and the top byte (almost always)
        if (visible) {
          const uint32_t bkt = (key - fold.msd_lo) >> fold.msd_shift;     // (a key below the window wraps to a huge value)
          atomicAdd(&s_hist[min(bkt, (uint32_t)(GGD_MSD_BINS - 1))], 1u);
          if (bkt > (uint32_t)(GGD_MSD_BINS - 1)) atomicAdd(&s_red[16], 1u);   // outside: the frame will be rendered again
        }
        const uint32_t d = key >> 24;
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
        if (__ballot(visible && d != d0) == 0ull) {
          if (lane == leader) atomicAdd(&s_hist[GGD_MSD_BINS + d0], (uint32_t)__popcll(act));
        } else if (visible) {
          atomicAdd(&s_hist[GGD_MSD_BINS + d], 1u);
        }
      } else
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const uint32_t d = (key >> (8 * pass)) & 0xffu;
        if (pass < 2) {
          if (visible) atomicAdd(&s_hist[pass * 256 + d], 1u);
          continue;
        }
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, leader);
        if (__ballot(visible && d != d0) == 0ull) {
          if (lane == leader) atomicAdd(&s_hist[pass * 256 + d0], (uint32_t)__popcll(act));
        } else if (visible) {
          atomicAdd(&s_hist[pass * 256 + d], 1u);
        }
      }
    }
    // (d) grids of <= 64 tile rows: entries per row for the row binning's first level -- difference array over the rows the
    //     Gaussian's rect covers, prefix over the lanes after the barrier
    if (fold.rows && visible) {
      atomicAdd(&s_rowdiff[rect_rows & 0xffffu], 1);
      atomicAdd(&s_rowdiff[rect_rows >> 16], -1);
    }
    uint32_t tsum = ntiles;
    // kept-key range of the frame (for the NEXT frames' two-launch-sort window): ~min and max, so that both reduce -- and
    // accumulate in the zeroed control block -- as maxima
    uint32_t nmin = visible ? ~key : 0u, kmax = visible ? key : 0u;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
      tsum += __shfl_xor(tsum, sh, 64);
      nmin = max(nmin, (uint32_t)__shfl_xor((int)nmin, sh, 64));
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, sh, 64));
    }
    if (lane == 0) { s_red[wv] = tsum; s_red[4 + wv] = (uint32_t)__popcll(act); s_red[8 + wv] = nmin; s_red[12 + wv] = kmax; }
    __syncthreads();
    uint32_t* hist = fold.ctl + (blockIdx.x % GGD_FOLD_REPS) * GGD_FOLD_REP_STRIDE;
    const int used = fold.msd ? GGD_FOLD_REP_STRIDE : 4 * 256;
    for (int b = threadIdx.x; b < used; b += 256) {
      const uint32_t c = s_hist[b];
      if (c) atomicAdd(&hist[b], c);
    }
    if (fold.rows && threadIdx.x < 64) {
      int c = s_rowdiff[lane];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(c, d, 64); if (lane >= d) c += o; }
      if (c) atomicAdd(&fold.ctl[GGD_FOLD_ROWTOT + (blockIdx.x % GGD_FOLD_REPS) * 64 + lane], (uint32_t)c);
    }
    if (threadIdx.x == 0) {
      const uint32_t kept = s_red[4] + s_red[5] + s_red[6] + s_red[7];
      fold.wg_info[blockIdx.x] = make_uint4(s_red[0] + s_red[1] + s_red[2] + s_red[3], kept,
                                            max(max(s_red[8], s_red[9]), max(s_red[10], s_red[11])),
                                            max(max(s_red[12], s_red[13]), max(s_red[14], s_red[15])));
      if (s_red[16]) atomicAdd(&fold.ctl[GGD_FOLD_OUTSIDE], s_red[16]);
    }
  }
}

__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* __restrict__ means3D,
                                                           const float* __restrict__ view,
                                                           uint8_t* __restrict__ present) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float x = means3D[3 * (size_t)i], y = means3D[3 * (size_t)i + 1], z = means3D[3 * (size_t)i + 2];
  const float tz = view[2] * x + view[6] * y + view[10] * z + view[14];
  present[i] = (uint8_t)(tz > 0.2f);
}

}  // namespace

int ggd_launch_preprocess(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const float* means3D,
                          const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, const float* rotations, const float* cov3D_precomp,
                          ggd_splat* splat, uint32_t* tiles_touched, uint8_t* clamped, int32_t* radii,
                          uint32_t* depth_keys, uint2* rect, uint32_t* trap_flag, uint32_t* zero_ptr, int zero_words,
                          const ggd_fold* fold) {
  if (prm.P == 0) return GGD_OK;
  const int grid = (prm.P + 255) / 256;
  const bool shvec = !colors_precomp && prm.M > 1 && prm.M <= 16 && ((3 * prm.M) & 3) == 0;
  const ggd_fold f = fold ? *fold : ggd_fold{};
  // focal lengths: wave-uniform correctly rounded divisions, done once here (same fp32 expression, same result)
  const float fx = (float)prm.width / (2.0f * prm.tanfovx), fy = (float)prm.height / (2.0f * prm.tanfovy);
#define GGD_PREPROCESS(SHV, FLD)                                                                                            \
  hipLaunchKernelGGL((preprocess_kernel<SHV, FLD>), dim3(grid), dim3(256), 0, s, prm.P, prm.M, prm.sh_degree, prm.width,     \
                     prm.height, prm.tanfovx, prm.tanfovy, fx, fy, prm.scale_modifier, prm.prefiltered, prm.raw_attributes,          \
                     prm.viewmatrix, prm.projmatrix, prm.campos, means3D, shs, colors_precomp, opacities, scales, rotations, \
                     cov3D_precomp, splat, tiles_touched, clamped, radii, depth_keys, rect, trap_flag, zero_ptr,             \
                     zero_ptr ? zero_words : 0, f)
  if (fold) { if (shvec) GGD_PREPROCESS(true, true); else GGD_PREPROCESS(false, true); }
  else { if (shvec) GGD_PREPROCESS(true, false); else GGD_PREPROCESS(false, false); }
#undef GGD_PREPROCESS
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int ggd_launch_mark_visible(ggd_ctx* ctx, hipStream_t s, int P, const float* means3D, const float* view,
                            uint8_t* present) {
  if (P == 0) return GGD_OK;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
