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
                                                       const float* __restrict__ dout, float* __restrict__ out) {
  constexpr int PPW = 64 / C;  // points per wave
  const int lane = threadIdx.x & 63;
  const int c = lane % C;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t n = wave * PPW + lane / C;
  if (n >= N) return;
  const float x = scale * pos[3 * n], y = scale * pos[3 * n + 1], z = scale * pos[3 * n + 2];
  const size_t plane_stride = (size_t)H * W * C;
  float acc = 0.0f;
  const float g = BACKWARD ? dout[n * C + c] * (1.0f / 3.0f) : 0.0f;
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
        else acc += t.w[k] * planes_cl[off];
      }
    }
  }
  if (!BACKWARD) out[n * C + c] = acc * (1.0f / 3.0f);
}

template <bool BACKWARD>
int launch(ggd_ctx* ctx, hipStream_t s, const float* planes_cl, float* dplanes_cl, int C, int H, int W, const float* pos,
           int N, float box_warp, const float* dout, float* out) {
  if (N <= 0) return GGD_OK;
  const float scale = 2.0f / box_warp;
#define GGD_TP(CC)                                                                                                  \
  case CC: {                                                                                                        \
    const int64_t waves = ((int64_t)N + (64 / CC) - 1) / (64 / CC);                                                 \
    hipLaunchKernelGGL((triplane_kernel<CC, BACKWARD>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,         \
                       planes_cl, dplanes_cl, H, W, pos, N, scale, dout, out);                                      \
  } break;
  switch (C) {
    GGD_TP(1) GGD_TP(2) GGD_TP(4) GGD_TP(8) GGD_TP(16) GGD_TP(32) GGD_TP(64)
    default: return ggd_fail(ctx, GGD_E_INVALID, "triplane: channel count must be a power of two <= 64");
  }
#undef GGD_TP
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}

}  // namespace

extern "C" int ggd_triplane_forward(ggd_ctx* ctx, void* stream, const float* planes_cl, int32_t C, int32_t H, int32_t W,
                                    const float* pos, int32_t N, float box_warp, float* out) {
  if (!ctx) return GGD_E_INVALID;
  if (N > 0 && (!planes_cl || !pos || !out || H <= 0 || W <= 0 || box_warp == 0.0f))
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_triplane_forward: bad argument");
  return launch<false>(ctx, static_cast<hipStream_t>(stream), planes_cl, nullptr, C, H, W, pos, N, box_warp, nullptr, out);
}

extern "C" int ggd_triplane_backward(ggd_ctx* ctx, void* stream, int32_t C, int32_t H, int32_t W, const float* pos,
                                     int32_t N, float box_warp, const float* dout, float* dplanes_cl) {
  if (!ctx) return GGD_E_INVALID;
  if (!dplanes_cl || H <= 0 || W <= 0 || C <= 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_triplane_backward: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  GGD_HIP(hipMemsetAsync(dplanes_cl, 0, (size_t)3 * H * W * C * sizeof(float), s));
  if (N > 0 && (!pos || !dout || box_warp == 0.0f)) return ggd_fail(ctx, GGD_E_INVALID, "ggd_triplane_backward: bad argument");
  return launch<true>(ctx, s, nullptr, dplanes_cl, C, H, W, pos, N, box_warp, dout, nullptr);
}
