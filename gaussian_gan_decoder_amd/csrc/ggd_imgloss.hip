// ggd_imgloss.hip -- fused image losses of the decoder training step.
//
// Replaces, for one rendered image [3,H,W] against its target, the PyTorch graph of
//   l1_loss / l2_loss / ssim   gaussian_splatting/utils/loss_utils.py:17-63  (11x11 Gaussian window, sigma 1.5, zero
//                              padding, C1 = 0.01^2, C2 = 0.03^2, mean over 3*H*W)
//   sobel_loss                 main/loss_utils/sobel_loss.py:19-30            (3x3 Sobel x / y cross-correlation SUMMED
//                              over the three channels, squared difference, mean over H*W)
// as used in main/train_pano2gaussian_decoder.py:246-261:
//   loss = w_l1 * L1 + w_l2 * L2 + w_ssim * (1 - SSIM) + w_sobel * Sobel
// -- about 60 small kernels forward + backward in PyTorch -- by three launches that produce the four loss terms AND
// dloss/dimage in one go (the loss is a scalar with known weights, so its gradient is formed directly):
//   A  ssim_stats_kernel  per 32x32 tile and channel: separable 11-tap blur of x, y, x^2, y^2, xy through LDS, the SSIM
//                         map, its partial derivatives w.r.t. the blurred moments (maps a, b, c), block-reduced sums of
//                         SSIM, |x-y|, (x-y)^2
//   B  sobel_kernel       channel-summed difference image -> Sobel responses gx, gy (maps) and the sum of gx^2 + gy^2
//   C  grad_kernel        dL/dx = -w_ssim/(3HW) [G*a + 2x G*b + y G*c] + L1 / L2 terms + the transposed Sobel stencil on
//                         (gx, gy); thread 0 also finalises the loss terms.
// All stencil work is HBM-trivial (a 512x512 image is 3 MB); what matters is the launch count on the step's critical
// path.  SSIM algebra: S = A1 A2 / (B1 B2), A1 = 2 mu1 mu2 + C1, A2 = 2 s12 + C2, B1 = mu1^2 + mu2^2 + C1,
// B2 = s11 + s22 + C2 with s11 = G*x^2 - mu1^2, s12 = G*xy - mu1 mu2;  c = dS/ds12 = 2 A1/(B1 B2),
// b = dS/ds11 = -S/B2,  a = dS/dmu1 (total) = 2 mu2 A2/(B1 B2) - 2 mu1 S/B1 - 2 mu1 b - mu2 c.
#include "ggd_common.h"

namespace {

constexpr int TILE = 32;
constexpr int HALO = 5;
constexpr int EXT = TILE + 2 * HALO;  // 42

struct Win { float g[11]; };

__device__ __forceinline__ float block_sum(float v, float* red) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float s = 0.f;
  if (tid == 0) s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(256) void ssim_stats_kernel(int W, int H, Win win, const float* __restrict__ img,
                                                         const float* __restrict__ tgt, float* __restrict__ abc,
                                                         float* __restrict__ sums) {
  __shared__ float sx[EXT][EXT + 1], sy[EXT][EXT + 1];
  __shared__ float hb[5][EXT][TILE + 1];
  __shared__ float red[4];
  const int tid = threadIdx.x, ch = blockIdx.z;
  const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
  const size_t plane = (size_t)H * W;
  const float* X = img + ch * plane;
  const float* Y = tgt + ch * plane;
  for (int i = tid; i < EXT * EXT; i += 256) {
    const int r = i / EXT, c = i % EXT;
    const int gy = y0 + r - HALO, gx = x0 + c - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    sx[r][c] = in ? X[(size_t)gy * W + gx] : 0.f;
    sy[r][c] = in ? Y[(size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < EXT * TILE; i += 256) {
    const int r = i / TILE, c = i % TILE;
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float a = sx[r][c + k], b = sy[r][c + k], g = win.g[k];
      m1 += g * a; m2 += g * b; e11 += g * (a * a); e22 += g * (b * b); e12 += g * (a * b);
    }
    hb[0][r][c] = m1; hb[1][r][c] = m2; hb[2][r][c] = e11; hb[3][r][c] = e22; hb[4][r][c] = e12;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
  float s_ssim = 0, s_l1 = 0, s_l2 = 0;
  const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int gy = y0 + r, gx = x0 + tx;
    if (gy >= H || gx >= W) continue;
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = win.g[k];
      m1 += g * hb[0][r + k][tx]; m2 += g * hb[1][r + k][tx]; e11 += g * hb[2][r + k][tx];
      e22 += g * hb[3][r + k][tx]; e12 += g * hb[4][r + k][tx];
    }
    const float s11 = e11 - m1 * m1, s22 = e22 - m2 * m2, s12 = e12 - m1 * m2;
    const float A1 = 2.f * m1 * m2 + C1, A2 = 2.f * s12 + C2, B1 = m1 * m1 + m2 * m2 + C1, B2 = s11 + s22 + C2;
    const float inv = 1.0f / (B1 * B2);
    const float S = A1 * A2 * inv;
    const float c = 2.f * A1 * inv;
    const float b = -S / B2;
    const float a = 2.f * m2 * A2 * inv - 2.f * m1 * S / B1 - 2.f * m1 * b - m2 * c;
    const size_t p = (size_t)gy * W + gx;
    abc[(ch * 3 + 0) * plane + p] = a;
    abc[(ch * 3 + 1) * plane + p] = b;
    abc[(ch * 3 + 2) * plane + p] = c;
    const float d = sx[r + HALO][tx + HALO] - sy[r + HALO][tx + HALO];
    s_ssim += S; s_l1 += fabsf(d); s_l2 += d * d;
  }
  const float t0 = block_sum(s_ssim, red), t1 = block_sum(s_l1, red), t2 = block_sum(s_l2, red);
  if (tid == 0) { atomicAdd(sums + 0, t1); atomicAdd(sums + 1, t2); atomicAdd(sums + 2, t0); }
}

__global__ __launch_bounds__(256) void sobel_kernel(int W, int H, const float* __restrict__ img,
                                                    const float* __restrict__ tgt, float* __restrict__ gxy,
                                                    float* __restrict__ sums) {
  __shared__ float red[4];
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const size_t plane = (size_t)H * W;
  float v = 0.f;
  if (x < W && y < H) {
    float d[3][3];
#pragma unroll
    for (int u = -1; u <= 1; ++u)
#pragma unroll
      for (int w = -1; w <= 1; ++w) {
        const int yy = y + u, xx = x + w;
        float s = 0.f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
          const size_t p = (size_t)yy * W + xx;
          s = (img[p] - tgt[p]) + (img[plane + p] - tgt[plane + p]) + (img[2 * plane + p] - tgt[2 * plane + p]);
        }
        d[u + 1][w + 1] = s;
      }
    // sobel_x = [[1,0,-1],[2,0,-2],[1,0,-1]], sobel_y = [[1,2,1],[0,0,0],[-1,-2,-1]] (cross-correlation)
    const float gx = (d[0][0] - d[0][2]) + 2.f * (d[1][0] - d[1][2]) + (d[2][0] - d[2][2]);
    const float gy = (d[0][0] + 2.f * d[0][1] + d[0][2]) - (d[2][0] + 2.f * d[2][1] + d[2][2]);
    const size_t p = (size_t)y * W + x;
    gxy[p] = gx; gxy[plane + p] = gy;
    v = gx * gx + gy * gy;
  }
  const float t = block_sum(v, red);
  if (threadIdx.x == 0) atomicAdd(sums + 3, t);
}

__global__ __launch_bounds__(256) void grad_kernel(int W, int H, Win win, const float* __restrict__ img,
                                                   const float* __restrict__ tgt, const float* __restrict__ abc,
                                                   const float* __restrict__ gxy, const float* __restrict__ sums,
                                                   float w_l1, float w_l2, float w_ssim, float w_sobel,
                                                   float* __restrict__ grad, float* __restrict__ terms) {
  __shared__ float sm[3][EXT][EXT + 1];
  __shared__ float hb[3][EXT][TILE + 1];
  const int tid = threadIdx.x, ch = blockIdx.z;
  const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
  const size_t plane = (size_t)H * W;
  const float n3 = 3.0f * (float)H * (float)W, n1 = (float)H * (float)W;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && ch == 0) {
    const float l1 = sums[0] / n3, l2 = sums[1] / n3, ds = 1.0f - sums[2] / n3, sb = sums[3] / n1;
    terms[0] = l1; terms[1] = l2; terms[2] = ds; terms[3] = sb;
    terms[4] = w_l1 * l1 + w_l2 * l2 + w_ssim * ds + w_sobel * sb;
  }
  for (int i = tid; i < EXT * EXT; i += 256) {
    const int r = i / EXT, c = i % EXT;
    const int gy = y0 + r - HALO, gx = x0 + c - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t p = in ? (size_t)gy * W + gx : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) sm[k][r][c] = in ? abc[(ch * 3 + k) * plane + p] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < EXT * TILE; i += 256) {
    const int r = i / TILE, c = i % TILE;
    float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = win.g[k];
      s0 += g * sm[0][r][c + k]; s1 += g * sm[1][r][c + k]; s2 += g * sm[2][r][c + k];
    }
    hb[0][r][c] = s0; hb[1][r][c] = s1; hb[2][r][c] = s2;
  }
  __syncthreads();
  const int tx = tid & 31, ty = tid >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    const int gy = y0 + r, gx = x0 + tx;
    if (gy >= H || gx >= W) continue;
    float Ga = 0, Gb = 0, Gc = 0;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = win.g[k];
      Ga += g * hb[0][r + k][tx]; Gb += g * hb[1][r + k][tx]; Gc += g * hb[2][r + k][tx];
    }
    const size_t p = (size_t)gy * W + gx;
    const float x = img[ch * plane + p], y = tgt[ch * plane + p];
    const float d = x - y;
    float gsum = -w_ssim / n3 * (Ga + 2.f * x * Gb + y * Gc);
    gsum += w_l1 / n3 * (float)((d > 0.f) - (d < 0.f));
    gsum += w_l2 * 2.f * d / n3;
    // transposed Sobel: dL/dd(p) = 2/N sum_{u,v} Kx[u][v] gx(p - (u,v)) + Ky[u][v] gy(p - (u,v)), responses outside = 0
    float sob = 0.f;
#pragma unroll
    for (int u = -1; u <= 1; ++u)
#pragma unroll
      for (int w = -1; w <= 1; ++w) {
        const int yy = gy - u, xx = gx - w;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float kx = (w == 0) ? 0.f : ((w < 0 ? 1.f : -1.f) * (u == 0 ? 2.f : 1.f));
        const float ky = (u == 0) ? 0.f : ((u < 0 ? 1.f : -1.f) * (w == 0 ? 2.f : 1.f));
        const size_t pp = (size_t)yy * W + xx;
        sob += kx * gxy[pp] + ky * gxy[plane + pp];
      }
    gsum += w_sobel * 2.f / n1 * sob;
    grad[ch * plane + p] = gsum;
  }
}

}  // namespace

extern "C" size_t ggd_image_loss_tmp_bytes(int32_t W, int32_t H) {
  if (W <= 0 || H <= 0) return 0;
  return ggd_align(64) + ggd_align((size_t)11 * H * W * sizeof(float));   // sums | 9 abc maps + 2 Sobel maps
}

extern "C" int ggd_image_loss(ggd_ctx* ctx, void* stream, int32_t W, int32_t H, const float* image,
                              const float* target, const float* weights4, float* terms5, float* grad_image,
                              void* tmp, size_t tmp_bytes) {
  if (!ctx) return GGD_E_INVALID;
  if (W <= 0 || H <= 0) return ggd_fail(ctx, GGD_E_INVALID, "ggd_image_loss: empty image");
  if (!image || !target || !weights4 || !terms5 || !grad_image || !tmp)
    return ggd_fail(ctx, GGD_E_INVALID, "ggd_image_loss: NULL pointer");
  if (tmp_bytes < ggd_image_loss_tmp_bytes(W, H)) return ggd_fail(ctx, GGD_E_INVALID, "ggd_image_loss: tmp too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  Win win;
  {  // loss_utils.py:23-25: exp(-(x - 5)^2 / (2 sigma^2)) as float32, normalised by its float32 sum
    float g[11], sum = 0.f;
    for (int i = 0; i < 11; ++i) { g[i] = (float)std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += g[i]; }
    for (int i = 0; i < 11; ++i) win.g[i] = g[i] / sum;
  }
  float* sums = static_cast<float*>(tmp);
  float* abc = reinterpret_cast<float*>(static_cast<char*>(tmp) + ggd_align(64));
  float* gxy = abc + (size_t)9 * H * W;
  GGD_HIP(hipMemsetAsync(sums, 0, 64, s));
  const dim3 tiles((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, 3);
  hipLaunchKernelGGL(ssim_stats_kernel, tiles, dim3(256), 0, s, W, H, win, image, target, abc, sums);
  hipLaunchKernelGGL(sobel_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, s, W, H, image, target, gxy, sums);
  hipLaunchKernelGGL(grad_kernel, tiles, dim3(256), 0, s, W, H, win, image, target, abc, gxy, sums, weights4[0],
                     weights4[1], weights4[2], weights4[3], grad_image, terms5);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
