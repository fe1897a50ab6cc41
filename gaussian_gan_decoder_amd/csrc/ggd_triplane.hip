// ggd_triplane.hip -- tri-plane feature gather for the per-point decoder (the caller on the input side of the raster
// hot path; SURVEY.md section 8f row 1 "optionally fuse the tri-plane gather").
//
// Computes what the reference does with three torch ops (main/decoder_models/sequential_decoder_reverse.py:42-57 ->
// eg3d/training/volumetric_rendering/renderer.py:40-65 `sample_from_planes`, then `triplane_features.mean(0)` in
// main/decoder_models/base_decoder.py:22):   out[n, :] = (1/3) * sum_p bilinear(plane_p, proj_p(pos_n))
// with grid_sample semantics (bilinear, zero padding, align_corners = False) and the EG3D plane axes
// (plane 0 -> (x, y), plane 1 -> (x, z), plane 2 -> (z, x)).
//
// Layout: planes are CHANNEL-LAST [3][H][W][C] so that one texel's C channels are contiguous (C = 32 -> one 128-byte
// line): lane = channel, 64/C points per wave; every texel access is one coalesced line read, and the backward's
// scatter-add is one line-wide group of float atomics per texel (torch's NCHW grid_sampler_2d_backward issues one
// scattered atomic per (point, channel): 9.6 ms per 5e5 points measured on MI355X; this kernel is HBM/L2 bound).
#include "ggd_common.h"

namespace {

struct Tap { int idx[4]; float w[4]; };  // up to 4 texels (idx < 0: outside -> zero padding)

__device__ __forceinline__ Tap bilinear_taps(float u, float v, int H, int W) {
  // grid_sample, align_corners = False: pixel = ((g + 1) * size - 1) / 2
  const float ix = ((u + 1.0f) * (float)W - 1.0f) * 0.5f;
  const float iy = ((v + 1.0f) * (float)H - 1.0f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float ax = ix - fx, ay = iy - fy;
  Tap t;
  t.w[0] = (1.0f - ax) * (1.0f - ay); t.w[1] = ax * (1.0f - ay); t.w[2] = (1.0f - ax) * ay; t.w[3] = ax * ay;
  const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
  t.idx[0] = (vx0 && vy0) ? y0 * W + x0 : -1;
  t.idx[1] = (vx1 && vy0) ? y0 * W + x1 : -1;
  t.idx[2] = (vx0 && vy1) ? y1 * W + x0 : -1;
  t.idx[3] = (vx1 && vy1) ? y1 * W + x1 : -1;
  return t;
}

__device__ __forceinline__ void plane_uv(int p, float x, float y, float z, float& u, float& v) {
  if (p == 0) { u = x; v = y; } else if (p == 1) { u = x; v = z; } else { u = z; v = x; }
}

template <int C, bool BACKWARD>
__global__ __launch_bounds__(256) void triplane_kernel(const float* __restrict__ planes_cl, float* __restrict__ dplanes_cl,
                                                       int H, int W, const float* __restrict__ pos, int N, float scale,
                                                       const float* __restrict__ dout, float* __restrict__ out,
                                                       const float* __restrict__ mod) {
  constexpr int PPW = 64 / C;  // points per wave
  const int lane = threadIdx.x & 63;
  const int c = lane % C;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n = wave * PPW + lane / C;
  if (n >= N) return;
  const float x = scale * pos[3 * n], y = scale * pos[3 * n + 1], z = scale * pos[3 * n + 2];
  const size_t plane_stride = (size_t)H * W * C;
  float acc = 0.0f;
  const float m = mod ? mod[c] : 1.0f;   // per-channel modulation of the planes (texel * m, rounded, is what is sampled)
  const float g = BACKWARD ? dout[n * C + c] * (1.0f / 3.0f) * m : 0.0f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float u, v;
    plane_uv(p, x, y, z, u, v);
    const Tap t = bilinear_taps(u, v, H, W);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (t.idx[k] >= 0) {
        const size_t off = p * plane_stride + (size_t)t.idx[k] * C + c;
        if (BACKWARD) atomicAdd(dplanes_cl + off, t.w[k] * g);
        else acc += t.w[k] * (planes_cl[off] * m);
      }
    }
  }
  if (!BACKWARD) out[n * C + c] = acc * (1.0f / 3.0f);
}

// ---- PanoHead tri-grid: every "plane" is a C x D grid sampled with a 3-D grid_sample (trilinear, zero padding,
// align_corners = False) at the point's three projected coordinates (PanoHead/training/volumetric_rendering/
// renderer.py:47-58; selected at main/decoder_models/sequential_decoder_reverse.py:42-50).  (u, v, w) index (W, H, D).
// axes: 0 = EG3D plane axes (plane 2 -> (z, x, y)), 1 = PanoHead (plane 2 -> (y, z, x)); plane 0 -> (x, y, z),
// plane 1 -> (x, z, y) in both.  Grids are channel-last [3][D][H][W][C]: lane = channel, one coalesced line per texel.
__device__ __forceinline__ void grid_uvw(int axes, int p, float x, float y, float z, float& u, float& v, float& w) {
  if (p == 0) { u = x; v = y; w = z; }
  else if (p == 1) { u = x; v = z; w = y; }
  else if (axes == 0) { u = z; v = x; w = y; }
  else { u = y; v = z; w = x; }
}

template <int C, bool BACKWARD>
__global__ __launch_bounds__(256) void trigrid_kernel(const float* __restrict__ grids_cl, float* __restrict__ dgrids_cl,
                                                      int D, int H, int W, int axes, const float* __restrict__ pos, int N,
                                                      float scale, const float* __restrict__ dout, float* __restrict__ out,
                                                      const float* __restrict__ mod /*[D][C] or null*/) {
  constexpr int PPW = 64 / C;  // points per wave
  const int lane = threadIdx.x & 63;
  const int c = lane % C;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n = wave * PPW + lane / C;
  if (n >= N) return;
  const float x = scale * pos[3 * n], y = scale * pos[3 * n + 1], z = scale * pos[3 * n + 2];
  const size_t grid_stride = (size_t)D * H * W * C;
  float acc = 0.0f;
  const float g = BACKWARD ? dout[n * C + c] * (1.0f / 3.0f) : 0.0f;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float u, v, w;
    grid_uvw(axes, p, x, y, z, u, v, w);
    const float ix = ((u + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float iy = ((v + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float iz = ((w + 1.0f) * (float)D - 1.0f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float ax = ix - fx, ay = iy - fy, az = iz - fz;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int xx = x0 + (k & 1), yy = y0 + ((k >> 1) & 1), zz = z0 + (k >> 2);
      if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D) {
        // weights in grid_sampler_3d's order: (x) * (y) * (z)
        const float wgt = ((k & 1) ? ax : 1.0f - ax) * ((k & 2) ? ay : 1.0f - ay) * ((k & 4) ? az : 1.0f - az);
        const size_t off = p * grid_stride + (((size_t)zz * H + yy) * W + xx) * C + c;
        const float m = mod ? mod[zz * C + c] : 1.0f;
        if (BACKWARD) atomicAdd(dgrids_cl + off, wgt * g * m);
        else acc += wgt * (grids_cl[off] * m);
      }
    }
  }
  if (!BACKWARD) out[n * C + c] = acc * (1.0f / 3.0f);
}

// ---- the gather, four channels per lane (C a multiple of 4, C / 4 a power of two <= 16) -----------------------------------
// With lane = channel (the kernels above) the 64 / C points of a wave repeat the whole coordinate -> tap arithmetic in every
// one of their C lanes (~175 VALU instructions per point), and a 128-byte texel line costs a wave instruction per tap and
// point pair.  Here a point owns C / 4 lanes, each loading 16 bytes of a texel line: a wave holds 256 / C points (8 for
// C = 32), one global_load_dwordx4 instruction fetches one tap of all of them, and the arithmetic is repeated C / 4 times
// instead of C times.  Same operations per channel in the same order as the kernels above: bit-identical features.
// (Measured before the rewrite: visiting the points block by block of a 32^3 grid -- every line an L2 hit after its first
// use -- bought 35 of 336 us and cost a 50 us sort: the gather was bound by its own instruction stream, not by the fabric.)
template <int C, bool G3>
__global__ __launch_bounds__(256) void gather4_kernel(const float* __restrict__ grids_cl, int D, int H, int W, int axes,
                                                      const float* __restrict__ pos, int N, float scale,
                                                      float* __restrict__ out, const float* __restrict__ mod) {
  constexpr int LPP = C / 4;            // lanes per point
  constexpr int PPW = 64 / LPP;         // points per wave
  const int lane = threadIdx.x & 63;
  const int q = lane % LPP;             // which 16 bytes of a texel line
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n = wave * PPW + lane / LPP;
  if (n >= N) return;
  const float x = scale * pos[3 * n], y = scale * pos[3 * n + 1], z = scale * pos[3 * n + 2];
  const int Dd = G3 ? D : 1;
  const size_t grid_stride = (size_t)Dd * H * W * C;
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    float u, v, w = 0.0f;
    if (G3) grid_uvw(axes, p, x, y, z, u, v, w); else plane_uv(p, x, y, z, u, v);
    const float ix = ((u + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float iy = ((v + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float iz = G3 ? ((w + 1.0f) * (float)D - 1.0f) * 0.5f : 0.0f;
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float ax = ix - fx, ay = iy - fy, az = iz - fz;
    // all taps of the plane are requested before the first is used
    float4 t[G3 ? 8 : 4];
    float wg[G3 ? 8 : 4];
    int zs[G3 ? 8 : 4];
#pragma unroll
    for (int k = 0; k < (G3 ? 8 : 4); ++k) {
      const int xx = x0 + (k & 1), yy = y0 + ((k >> 1) & 1), zz = G3 ? z0 + (k >> 2) : 0;
      const bool in = xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < Dd;
      // weights in grid_sampler's order: (x) * (y) [* (z)]
      float wk = ((k & 1) ? ax : 1.0f - ax) * ((k & 2) ? ay : 1.0f - ay);
      if (G3) wk = wk * ((k & 4) ? az : 1.0f - az);
      wg[k] = wk;
      zs[k] = in ? zz : -1;
      t[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (in) t[k] = *reinterpret_cast<const float4*>(grids_cl + p * grid_stride + (((size_t)zz * H + yy) * W + xx) * C + 4 * q);
    }
#pragma unroll
    for (int k = 0; k < (G3 ? 8 : 4); ++k) {
      if (zs[k] >= 0) {   // (an outside tap adds nothing: skipped as the kernels above skip it, so that -0 / NaN texels cannot differ)
        float4 m = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (mod) m = *reinterpret_cast<const float4*>(mod + zs[k] * C + 4 * q);
        acc.x += wg[k] * (t[k].x * m.x); acc.y += wg[k] * (t[k].y * m.y);
        acc.z += wg[k] * (t[k].z * m.z); acc.w += wg[k] * (t[k].w * m.w);
      }
    }
  }
  const float third = 1.0f / 3.0f;
  *reinterpret_cast<float4*>(out + n * C + 4 * q) = make_float4(acc.x * third, acc.y * third, acc.z * third, acc.w * third);
}

template <int C>
int launch_gather4(ggd_ctx* ctx, hipStream_t s, const float* grids_cl, int D, int H, int W, int axes, const float* pos, int N,
                   float box_warp, float* out, const float* mod) {
  constexpr int PPW = 64 / (C / 4);
  const int64_t waves = ((int64_t)N + PPW - 1) / PPW;
  const float scale = 2.0f / box_warp;
  if (D > 0)
    hipLaunchKernelGGL((gather4_kernel<C, true>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, grids_cl, D, H, W, axes,
                       pos, N, scale, out, mod);
  else
    hipLaunchKernelGGL((gather4_kernel<C, false>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, grids_cl, D, H, W, axes,
                       pos, N, scale, out, mod);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

template <bool BACKWARD>
int launch_trigrid(ggd_ctx* ctx, hipStream_t s, const float* grids_cl, float* dgrids_cl, int C, int D, int H, int W, int axes,
                   const float* pos, int N, float box_warp, const float* dout, float* out, const float* mod) {
  if (N <= 0) return GGD_OK;
  const float scale = 2.0f / box_warp;
#define GGD_TG(CC)                                                                                                  \
  case CC: {                                                                                                        \
    const int64_t waves = ((int64_t)N + (64 / CC) - 1) / (64 / CC);                                                 \
    hipLaunchKernelGGL((trigrid_kernel<CC, BACKWARD>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,          \
                       grids_cl, dgrids_cl, D, H, W, axes, pos, N, scale, dout, out, mod);                          \
  } break;
  switch (C) {
    GGD_TG(1) GGD_TG(2) GGD_TG(4) GGD_TG(8) GGD_TG(16) GGD_TG(32) GGD_TG(64)
    default: return ggd_fail(ctx, GGD_E_INVALID, "trigrid: channel count must be a power of two <= 64");
  }
#undef GGD_TG
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

template <bool BACKWARD>
int launch(ggd_ctx* ctx, hipStream_t s, const float* planes_cl, float* dplanes_cl, int C, int H, int W, const float* pos,
           int N, float box_warp, const float* dout, float* out, const float* mod) {
  if (N <= 0) return GGD_OK;
  const float scale = 2.0f / box_warp;
#define GGD_TP(CC)                                                                                                  \
  case CC: {                                                                                                        \
    const int64_t waves = ((int64_t)N + (64 / CC) - 1) / (64 / CC);                                                 \
    hipLaunchKernelGGL((triplane_kernel<CC, BACKWARD>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,         \
                       planes_cl, dplanes_cl, H, W, pos, N, scale, dout, out, mod);                                 \
  } break;
  switch (C) {
    GGD_TP(1) GGD_TP(2) GGD_TP(4) GGD_TP(8) GGD_TP(16) GGD_TP(32) GGD_TP(64)
    default: return ggd_fail(ctx, GGD_E_INVALID, "triplane: channel count must be a power of two <= 64");
  }
#undef GGD_TP
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Sorted-run backward (tri-planes AND tri-grids).  A training scene puts 5e5 surface points on a few 1e4 cells per plane:
// dozens of (point, plane) items per cell.  The items are sorted by cell (the library's 32-bit onesweep sort; items with
// no tap inside the grid carry the key ~0 and are dropped by its first pass), and a wave then walks a contiguous piece of
// the sorted list with the cell's 4 (8) tap sums in REGISTERS -- lane = channel, the 64 / C lane groups of a wave walk
// 64 / C disjoint sub-pieces -- and issues the taps' atomics once per run of equal cells.  No LDS accumulation tile (the
// 8 x 8-texel LDS tiles of the earlier design cost one dependent LDS read-add-write per item and tap and do not extend to
// a tri-grid: 9 x 9 x D x C floats per wave), global float atomics ~ cells x taps instead of items x taps.
constexpr int SR_CHUNK = 256;          // sorted items per wave
constexpr int SR_DEPTH = 8;            // gradient rows in flight per lane group
constexpr int SR_MOD_LDS = 1024;       // floats of the modulation table kept in LDS (D * C <= this; else read from memory)
constexpr int SR_MIN_POINTS = 49152;   // below: the plain scatter-add (the sort's launches cost more than they save)

struct SrGeom { int W, H, D, xb, yb, zb; };   // D = 0: 2-D planes; bit widths of the (x0 + 1), (y0 + 1), (z0 + 1) key fields
static inline int sr_bits(int vmax) { int b = 1; while ((1 << b) <= vmax) ++b; return b; }   // bits to hold 0 .. vmax

// cell key of item (point, plane p) and its interpolation fractions; ~0 when no tap lies inside the grid
template <bool G3>
__device__ __forceinline__ uint32_t sr_item(const SrGeom& g, int axes, int p, float x, float y, float z, float& ax, float& ay,
                                            float& az) {
  float u, v, w = 0.0f;
  if (G3) grid_uvw(axes, p, x, y, z, u, v, w); else plane_uv(p, x, y, z, u, v);
  const float ix = ((u + 1.0f) * (float)g.W - 1.0f) * 0.5f, iy = ((v + 1.0f) * (float)g.H - 1.0f) * 0.5f;
  const float iz = G3 ? ((w + 1.0f) * (float)g.D - 1.0f) * 0.5f : 0.0f;
  const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
  ax = ix - fx; ay = iy - fy; az = iz - fz;
  bool ok = fx >= -1.0f && fx <= (float)(g.W - 1) && fy >= -1.0f && fy <= (float)(g.H - 1);   // false for NaN
  if (G3) ok = ok && fz >= -1.0f && fz <= (float)(g.D - 1);
  if (!ok) return ~0u;
  uint32_t key = (uint32_t)p;
  if (G3) key = (key << g.zb) | (uint32_t)((int)fz + 1);
  key = (key << g.yb) | (uint32_t)((int)fy + 1);
  key = (key << g.xb) | (uint32_t)((int)fx + 1);
  return key;
}

template <bool G3>
__global__ __launch_bounds__(256) void sr_keys_kernel(const float* __restrict__ pos, int N, float scale, SrGeom g, int axes,
                                                      uint32_t* __restrict__ keys /*[3][N]*/) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float x = scale * pos[3 * n], y = scale * pos[3 * n + 1], z = scale * pos[3 * n + 2];
  float ax, ay, az;
#pragma unroll
  for (int p = 0; p < 3; ++p) keys[(size_t)p * N + n] = sr_item<G3>(g, axes, p, x, y, z, ax, ay, az);
}

template <int C, bool G3>
__global__ __launch_bounds__(256) void sr_accumulate_kernel(
    const uint32_t* __restrict__ keys_a, const uint32_t* __restrict__ vals_a, const uint32_t* __restrict__ keys_b,
    const uint32_t* __restrict__ vals_b, const uint32_t* __restrict__ flat_flag, const uint32_t* __restrict__ n_valid_ptr,
    const float* __restrict__ pos, int N, float scale, SrGeom g, int axes, const float* __restrict__ dout,
    const float* __restrict__ mod, float* __restrict__ dgrid) {
  constexpr int G = 64 / C;            // lane groups = independent streams of one wave
  constexpr int T = G3 ? 8 : 4;
  constexpr int SL = SR_CHUNK / G;     // items per stream
  __shared__ uint4 s_rec[4][64];       // per item of the batch: point | run-start bit, ax, ay, az
  __shared__ uint32_t s_key[4][64];
  __shared__ float s_mod[SR_MOD_LDS];  // the modulation table: a flush must not wait on a global load (it would wait for
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;   // every gradient row in flight with it)
  const int nmod = (G3 ? g.D : 1) * C;
  const bool mod_lds = mod != nullptr && nmod <= SR_MOD_LDS;
  if (mod_lds) {
    for (int i = threadIdx.x; i < nmod; i += 256) s_mod[i] = mod[i];
    __syncthreads();
  }
  const uint32_t nv = *n_valid_ptr;
  const uint32_t wbase = (blockIdx.x * 4u + (uint32_t)wv) * SR_CHUNK;
  if (wbase >= nv) return;
  const bool flat = *flat_flag != 0u;  // the sort's last pass found a constant digit and left the result in (b)
  const uint32_t* keys = flat ? keys_b : keys_a;
  const uint32_t* vals = flat ? vals_b : vals_a;
  const int s = lane / C, c = lane % C;
  const uint32_t sbeg = wbase + (uint32_t)s * SL, send = min(nv, sbeg + SL);
  const int pshift = g.xb + g.yb + g.zb;
  float acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = 0.0f;
  uint32_t cur = ~0u;                  // the cell whose sums the lane group holds (~0: none)

  auto flush = [&](uint32_t key) {
    const int x0 = (int)(key & ((1u << g.xb) - 1u)) - 1, y0 = (int)((key >> g.xb) & ((1u << g.yb) - 1u)) - 1;
    const int z0 = G3 ? (int)((key >> (g.xb + g.yb)) & ((1u << g.zb) - 1u)) - 1 : 0;
    const int p = (int)(key >> pshift);
    const int Dd = G3 ? g.D : 1;
    float* base = dgrid + (size_t)p * Dd * g.H * g.W * C + c;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int xx = x0 + (t & 1), yy = y0 + ((t >> 1) & 1), zz = z0 + (t >> 2);
      if (xx >= 0 && xx < g.W && yy >= 0 && yy < g.H && zz >= 0 && zz < Dd) {
        float v = acc[t] * (1.0f / 3.0f);
        if (mod_lds) v *= s_mod[zz * C + c];
        else if (mod) v *= mod[zz * C + c];
        atomicAdd(base + (((size_t)zz * g.H + yy) * g.W + xx) * C, v);
      }
    }
  };

  for (uint32_t b0 = 0; b0 < (uint32_t)SL; b0 += C) {
    if (wbase + b0 >= nv) break;       // stream 0 is the earliest: nothing left for any stream
    {  // lane (s, c) prepares item c of its stream's batch
      const uint32_t i = sbeg + b0 + (uint32_t)c;
      uint32_t key = ~0u, rec0 = 0x80000000u;
      float ax = 0.0f, ay = 0.0f, az = 0.0f;
      if (i < send) {
        key = keys[i];
        const uint32_t prev = i > sbeg ? keys[i - 1] : ~0u;
        const uint32_t p = key >> pshift;
        const uint32_t n = vals[i] - p * (uint32_t)N;
        const float x = scale * pos[3 * (size_t)n], y = scale * pos[3 * (size_t)n + 1], z = scale * pos[3 * (size_t)n + 2];
        (void)sr_item<G3>(g, axes, (int)p, x, y, z, ax, ay, az);
        rec0 = n | ((i == sbeg || prev != key) ? 0x80000000u : 0u);
      }
      s_rec[wv][lane] = make_uint4(rec0, __builtin_bit_cast(uint32_t, ax), __builtin_bit_cast(uint32_t, ay),
                                   __builtin_bit_cast(uint32_t, az));
      s_key[wv][lane] = key;
    }
    __builtin_amdgcn_wave_barrier();
    for (int j0 = 0; j0 < C; j0 += SR_DEPTH) {
      uint4 r[SR_DEPTH];
      float gr[SR_DEPTH];
#pragma unroll
      for (int k = 0; k < SR_DEPTH; ++k) r[k] = s_rec[wv][s * C + j0 + k];
#pragma unroll
      for (int k = 0; k < SR_DEPTH; ++k) gr[k] = dout[(size_t)(r[k].x & 0x7fffffffu) * C + c];
#pragma unroll
      for (int k = 0; k < SR_DEPTH; ++k) {
        if (r[k].x >> 31) {            // first item of a run (uniform over the lane group): hand the finished cell over
          if (cur != ~0u) flush(cur);
          cur = s_key[wv][s * C + j0 + k];
#pragma unroll
          for (int t = 0; t < T; ++t) acc[t] = 0.0f;
        }
        const float ax = __builtin_bit_cast(float, r[k].y), ay = __builtin_bit_cast(float, r[k].z);
        const float bx = 1.0f - ax, by = 1.0f - ay;
        // weights in grid_sampler's order: (x) * (y) [* (z)]
        const float w00 = bx * by, w10 = ax * by, w01 = bx * ay, w11 = ax * ay;
        if constexpr (G3) {
          const float az = __builtin_bit_cast(float, r[k].w), bz = 1.0f - az;
          acc[0] = __builtin_fmaf(w00 * bz, gr[k], acc[0]); acc[1] = __builtin_fmaf(w10 * bz, gr[k], acc[1]);
          acc[2] = __builtin_fmaf(w01 * bz, gr[k], acc[2]); acc[3] = __builtin_fmaf(w11 * bz, gr[k], acc[3]);
          acc[4] = __builtin_fmaf(w00 * az, gr[k], acc[4]); acc[5] = __builtin_fmaf(w10 * az, gr[k], acc[5]);
          acc[6] = __builtin_fmaf(w01 * az, gr[k], acc[6]); acc[7] = __builtin_fmaf(w11 * az, gr[k], acc[7]);
        } else {
          acc[0] = __builtin_fmaf(w00, gr[k], acc[0]); acc[1] = __builtin_fmaf(w10, gr[k], acc[1]);
          acc[2] = __builtin_fmaf(w01, gr[k], acc[2]); acc[3] = __builtin_fmaf(w11, gr[k], acc[3]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (cur != ~0u) flush(cur);
}

size_t sr_tmp_bytes(int N) {
  const size_t arr = ggd_align((size_t)3 * N * sizeof(uint32_t));
  return 5 * arr + ggd_sort32_tmp_bytes((int64_t)3 * N);
}

bool sr_supported(int C, int D, int H, int W, int N) {
  if (!(C == 16 || C == 32 || C == 64) || N < SR_MIN_POINTS || (int64_t)3 * N >= (1ll << 31)) return false;
  return 2 + sr_bits(W) + sr_bits(H) + (D > 0 ? sr_bits(D) : 0) <= 32;
}

// D = 0: the 2-D tri-plane form
template <int C>
int launch_sorted_backward(ggd_ctx* ctx, hipStream_t s, int D, int H, int W, int axes, const float* pos, int N, float box_warp,
                           const float* dout, const float* mod, float* dgrid) {
  int rc = ggd_reserve_scratch(ctx, sr_tmp_bytes(N), s);
  if (rc != GGD_OK) return rc;
  const float scale = 2.0f / box_warp;
  const SrGeom g{W, H, D, sr_bits(W), sr_bits(H), D > 0 ? sr_bits(D) : 0};
  const int64_t M = (int64_t)3 * N;
  const size_t arr = ggd_align((size_t)M * sizeof(uint32_t));
  char* p = static_cast<char*>(ctx->scratch);
  uint32_t* keys_src = reinterpret_cast<uint32_t*>(p); p += arr;
  uint32_t* keys_a = reinterpret_cast<uint32_t*>(p); p += arr;
  uint32_t* vals_a = reinterpret_cast<uint32_t*>(p); p += arr;
  uint32_t* keys_b = reinterpret_cast<uint32_t*>(p); p += arr;
  uint32_t* vals_b = reinterpret_cast<uint32_t*>(p); p += arr;
  void* sort_tmp = p;
  const size_t sort_bytes = ggd_sort32_tmp_bytes(M);
  if (D > 0) hipLaunchKernelGGL((sr_keys_kernel<true>), dim3((N + 255) / 256), dim3(256), 0, s, pos, N, scale, g, axes, keys_src);
  else hipLaunchKernelGGL((sr_keys_kernel<false>), dim3((N + 255) / 256), dim3(256), 0, s, pos, N, scale, g, axes, keys_src);
  rc = ggd_launch_sort32_iota(ctx, s, keys_src, keys_a, vals_a, keys_b, vals_b, M, 32, sort_tmp, sort_bytes, nullptr, nullptr,
                              /*flag_flat_last=*/true);
  if (rc != GGD_OK) return rc;
  const uint32_t* n_valid = ggd_sort32_nvalid_ptr(sort_tmp);
  const uint32_t* flat = ggd_sort32_flat_ptr(sort_tmp);
  const unsigned blocks = (unsigned)((M + 4 * SR_CHUNK - 1) / (4 * SR_CHUNK));
  if (D > 0)
    hipLaunchKernelGGL((sr_accumulate_kernel<C, true>), dim3(blocks), dim3(256), 0, s, keys_a, vals_a, keys_b, vals_b, flat,
                       n_valid, pos, N, scale, g, axes, dout, mod, dgrid);
  else
    hipLaunchKernelGGL((sr_accumulate_kernel<C, false>), dim3(blocks), dim3(256), 0, s, keys_a, vals_a, keys_b, vals_b, flat,
                       n_valid, pos, N, scale, g, axes, dout, mod, dgrid);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

int sorted_backward(ggd_ctx* ctx, hipStream_t s, int C, int D, int H, int W, int axes, const float* pos, int N, float box_warp,
                    const float* dout, const float* mod, float* dgrid) {
  switch (C) {
    case 16: return launch_sorted_backward<16>(ctx, s, D, H, W, axes, pos, N, box_warp, dout, mod, dgrid);
    case 32: return launch_sorted_backward<32>(ctx, s, D, H, W, axes, pos, N, box_warp, dout, mod, dgrid);
    default: return launch_sorted_backward<64>(ctx, s, D, H, W, axes, pos, N, box_warp, dout, mod, dgrid);
  }
}

}  // namespace

extern "C" int ggd_planes_gather(ggd_ctx* ctx, void* stream, const float* grids_cl, int32_t C, int32_t D, int32_t H,
                                 int32_t W, int32_t axes, const float* mod, const float* pos, int32_t N, float box_warp,
                                 float* out) {
  if (!ctx) return GGD_E_INVALID;
  if (axes < 0 || axes > 1 || D < 0 || (D == 0 && axes != 0))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_planes_gather: bad axes / depth (the 2-D form has the EG3D axes only)");
  if (N > 0 && (!grids_cl || !pos || !out || H <= 0 || W <= 0 || box_warp == 0.0f))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_planes_gather: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (N <= 0) return GGD_OK;
  // four channels per lane where the channel count allows it (the decoder's 32); lane = channel otherwise
  if (C == 16) return launch_gather4<16>(ctx, s, grids_cl, D, H, W, axes, pos, N, box_warp, out, mod);
  if (C == 32) return launch_gather4<32>(ctx, s, grids_cl, D, H, W, axes, pos, N, box_warp, out, mod);
  if (C == 64) return launch_gather4<64>(ctx, s, grids_cl, D, H, W, axes, pos, N, box_warp, out, mod);
  if (D == 0) return launch<false>(ctx, s, grids_cl, nullptr, C, H, W, pos, N, box_warp, nullptr, out, mod);
  return launch_trigrid<false>(ctx, s, grids_cl, nullptr, C, D, H, W, axes, pos, N, box_warp, nullptr, out, mod);
}

extern "C" int ggd_planes_scatter(ggd_ctx* ctx, void* stream, int32_t C, int32_t D, int32_t H, int32_t W, int32_t axes,
                                  const float* mod, const float* pos, int32_t N, float box_warp, const float* dout,
                                  float* dgrids_cl, int32_t accumulate) {
  if (!ctx) return GGD_E_INVALID;
  if (!dgrids_cl || H <= 0 || W <= 0 || C <= 0 || D < 0 || axes < 0 || axes > 1 || (D == 0 && axes != 0))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_planes_scatter: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Dd = D > 0 ? D : 1;
  if (!accumulate) GGD_HIP(hipMemsetAsync(dgrids_cl, 0, (size_t)3 * Dd * H * W * C * sizeof(float), s));
  if (N <= 0) return GGD_OK;
  if (!pos || !dout || box_warp == 0.0f) return ggd_fail(ctx, GGD_E_INVALID, "ggd_planes_scatter: bad argument");
  // many points per cell: sort the (point, plane) items by cell and sum runs in registers; few: plain scatter-add
  if (sr_supported(C, D, H, W, N)) return sorted_backward(ctx, s, C, D, H, W, axes, pos, N, box_warp, dout, mod, dgrids_cl);
  if (D == 0) return launch<true>(ctx, s, nullptr, dgrids_cl, C, H, W, pos, N, box_warp, dout, nullptr, mod);
  return launch_trigrid<true>(ctx, s, nullptr, dgrids_cl, C, D, H, W, axes, pos, N, box_warp, dout, nullptr, mod);
}

extern "C" int ggd_triplane_forward(ggd_ctx* ctx, void* stream, const float* planes_cl, int32_t C, int32_t H, int32_t W,
                                    const float* pos, int32_t N, float box_warp, float* out) {
  return ggd_planes_gather(ctx, stream, planes_cl, C, 0, H, W, 0, nullptr, pos, N, box_warp, out);
}

extern "C" int ggd_triplane_backward(ggd_ctx* ctx, void* stream, int32_t C, int32_t H, int32_t W, const float* pos,
                                     int32_t N, float box_warp, const float* dout, float* dplanes_cl) {
  return ggd_planes_scatter(ctx, stream, C, 0, H, W, 0, nullptr, pos, N, box_warp, dout, dplanes_cl, 0);
}

extern "C" int ggd_trigrid_forward(ggd_ctx* ctx, void* stream, const float* grids_cl, int32_t C, int32_t D, int32_t H,
                                   int32_t W, int32_t axes, const float* pos, int32_t N, float box_warp, float* out) {
  if (ctx && D <= 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_trigrid_forward: bad axes / depth");
  return ggd_planes_gather(ctx, stream, grids_cl, C, D, H, W, axes, nullptr, pos, N, box_warp, out);
}

extern "C" int ggd_trigrid_backward(ggd_ctx* ctx, void* stream, int32_t C, int32_t D, int32_t H, int32_t W, int32_t axes,
                                    const float* pos, int32_t N, float box_warp, const float* dout, float* dgrids_cl) {
  if (ctx && D <= 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_trigrid_backward: bad argument");
  return ggd_planes_scatter(ctx, stream, C, D, H, W, axes, nullptr, pos, N, box_warp, dout, dgrids_cl, 0);
}
