// ggd_math.h -- per-Gaussian device math shared by the forward and backward preprocess kernels.
// fp32, evaluated exactly as written (-ffp-contract=off); the operation order is part of the parity contract
// (radii / tiles_touched / depth bits are bit-exact anchors), see SURVEY.md section 9.2.
#pragma once
#include "ggd_common.h"

namespace ggdm {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Mat16 { float m[16]; };

// Camera matrices / bg / campos are small read-only device arrays written before the launch: read them through
// the constant address space so the wave-uniform loads become s_load (SGPRs, scalar cache) instead of 16 VMEM
// loads per lane.
typedef const __attribute__((address_space(4))) float* ggd_cptr;
__device__ __forceinline__ ggd_cptr as_const(const float* p) { return (ggd_cptr)(unsigned long long)p; }

__device__ __forceinline__ Mat16 load_mat(const float* __restrict__ p) {
  Mat16 r;
  ggd_cptr c = as_const(p);
#pragma unroll
  for (int i = 0; i < 16; ++i) r.m[i] = c[i];
  return r;
}

// Activation prologue of the reference's GaussianModel getters (gaussian_model.py:100-121), fused on request.
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float4 act_normalize(float4 q, float& norm) {
  norm = sqrtf((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
  const float d = fmaxf(norm, 1e-12f);  // torch.nn.functional.normalize: v / max(||v||, eps)
  return make_float4(q.x / d, q.y / d, q.z / d, q.w / d);
}

// Sigma = R S S R^T, stored (S00,S01,S02,S11,S12,S22); quaternion (w,x,y,z) used as given (not re-normalised).
__device__ __forceinline__ void cov3d_from_scale_rot(const float s3[3], float mod, const float4 q, float cov6[6]) {
  const float r = q.x, x = q.y, y = q.z, z = q.w;
  float R[3][3];
  R[0][0] = 1.0f - 2.0f * (y * y + z * z);
  R[0][1] = 2.0f * (x * y - r * z);
  R[0][2] = 2.0f * (x * z + r * y);
  R[1][0] = 2.0f * (x * y + r * z);
  R[1][1] = 1.0f - 2.0f * (x * x + z * z);
  R[1][2] = 2.0f * (y * z - r * x);
  R[2][0] = 2.0f * (x * z - r * y);
  R[2][1] = 2.0f * (y * z + r * x);
  R[2][2] = 1.0f - 2.0f * (x * x + y * y);
  const float s[3] = {mod * s3[0], mod * s3[1], mod * s3[2]};
  float M[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[k][j] = s[k] * R[j][k];
  auto sig = [&](int i, int j) { return (M[0][i] * M[0][j] + M[1][i] * M[1][j]) + M[2][i] * M[2][j]; };
  cov6[0] = sig(0, 0); cov6[1] = sig(0, 1); cov6[2] = sig(0, 2);
  cov6[3] = sig(1, 1); cov6[4] = sig(1, 2); cov6[5] = sig(2, 2);
}

// cov2D = (J W) Sigma (J W)^T  ->  (a, b, c) before the 0.3 low-pass.
__device__ __forceinline__ void ewa_cov2d(const float t[3], float fx, float fy, float tanfovx, float tanfovy,
                                          const float cov6[6], const Mat16& V, float abc[3], float T[2][3],
                                          float tclamped[3], bool& clampx, bool& clampy) {
  const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  const float tz = t[2];
  const float txtz = t[0] / tz, tytz = t[1] / tz;
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
  clampx = (txtz < -limx) || (txtz > limx);
  clampy = (tytz < -limy) || (tytz > limy);
  tclamped[0] = tx; tclamped[1] = ty; tclamped[2] = tz;
  const float J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
  const float J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
#pragma unroll
  for (int j = 0; j < 3; ++j) {  // W[r][c] = V.m[4c + r]
    T[0][j] = J00 * V.m[4 * j + 0] + J02 * V.m[4 * j + 2];
    T[1][j] = J11 * V.m[4 * j + 1] + J12 * V.m[4 * j + 2];
  }
  const float S[3][3] = {{cov6[0], cov6[1], cov6[2]}, {cov6[1], cov6[3], cov6[4]}, {cov6[2], cov6[4], cov6[5]}};
  float U[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) U[i][j] = (T[i][0] * S[0][j] + T[i][1] * S[1][j]) + T[i][2] * S[2][j];
  abc[0] = (U[0][0] * T[0][0] + U[0][1] * T[0][1]) + U[0][2] * T[0][2];
  abc[1] = (U[0][0] * T[1][0] + U[0][1] * T[1][1]) + U[0][2] * T[1][2];
  abc[2] = (U[1][0] * T[1][0] + U[1][1] * T[1][1]) + U[1][2] * T[1][2];
}

__device__ __forceinline__ void sh_to_rgb(int deg, const float* __restrict__ sh, const float p[3],
                                          const float campos[3], float rgb[3], uint32_t& clamp_bits) {
  float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx / len, y = dy / len, z = dz / len;
  clamp_bits = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#define SHK(k) sh[(k) * 3 + c]
    float res = SH_C0 * SHK(0);
    if (deg > 0) {
      res = res - SH_C1 * y * SHK(1) + SH_C1 * z * SHK(2) - SH_C1 * x * SHK(3);
      if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * SHK(4) + SH_C2[1] * yz * SHK(5) + SH_C2[2] * (2.0f * zz - xx - yy) * SHK(6) +
              SH_C2[3] * xz * SHK(7) + SH_C2[4] * (xx - yy) * SHK(8);
        if (deg > 2) {
          res = res + SH_C3[0] * y * (3.0f * xx - yy) * SHK(9) + SH_C3[1] * xy * z * SHK(10) +
                SH_C3[2] * y * (4.0f * zz - xx - yy) * SHK(11) +
                SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SHK(12) +
                SH_C3[4] * x * (4.0f * zz - xx - yy) * SHK(13) + SH_C3[5] * z * (xx - yy) * SHK(14) +
                SH_C3[6] * x * (xx - 3.0f * yy) * SHK(15);
        }
      }
    }
#undef SHK
    res = res + 0.5f;
    if (res < 0.0f) clamp_bits |= (1u << c);
    rgb[c] = fmaxf(res, 0.0f);
  }
}


}  // namespace ggdm
