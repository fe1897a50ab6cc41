// ggd_preprocess_bwd.hip -- stage a11: per-Gaussian backward (conic -> cov2D -> cov3D / view point; screen
// position through the perspective divide; colour -> SH (+ view direction); cov3D -> scale, quaternion).
//
// Replaces computeCov2D-backward + preprocess-backward inside `_C.rasterize_gaussians_backward`, which the
// reference reaches through autograd from main/train_pano2gaussian_decoder.py:263 (loss.backward()).  Algorithm:
// SURVEY.md section 9.6.  One lane per Gaussian, fused into ONE streaming pass (upstream runs two kernels and
// round-trips dL_dcov3D through memory when scales/rotations are given); cov3D is recomputed from scale/rotation
// instead of being stored by the forward pass (saves 24 B/Gaussian written + read).
// Conventions kept from the published algorithm because they change values beyond the 1e-5 tolerance otherwise:
// 1/(det^2 + 1e-7) in the conic inverse, dL_dconic slot 1 = half the true d/dB, and the clamped view-space x,y
// treated as independent of z.
#include "ggd_math.h"

namespace {
using namespace ggdm;

// The per-Gaussian work of the kernels below.  dsh_stage: where this Gaussian's 3 M SH gradients go instead of
// dL_dsh[i] (the staged kernel's LDS row), or nullptr.
// SHVEC: the Gaussian's SH row is read with dwordx4 loads into registers (see preprocess_kernel); DREG: its gradients are
// collected in registers too and leave as 3 M / 4 dwordx4 stores
template <bool SHVEC, bool DREG = false>
__device__ __forceinline__ void preprocess_backward_body(
    int i, float* dsh_stage,
    int P, int M, int deg, int W, int H, float tanfovx, float tanfovy, float mod, int raw,
    const float* __restrict__ opacities_raw, float* __restrict__ dL_dopacity,
    const float* __restrict__ view, const float* __restrict__ proj, const float* __restrict__ campos_p,
    const float* __restrict__ means3D, const float* __restrict__ shs, const float* __restrict__ colors_precomp,
    const float* __restrict__ scales, const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp,
    const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
    const float* __restrict__ grad_acc, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dcolors,
    float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
    float* __restrict__ dL_dscales, float* __restrict__ dL_drots) {
  const size_t ii = (size_t)i;
  if (radii[i] <= 0) {  // culled: every gradient of this Gaussian is zero (the caller does not pre-fill the arrays)
    dL_dmean2D[3 * ii] = 0.0f; dL_dmean2D[3 * ii + 1] = 0.0f; dL_dmean2D[3 * ii + 2] = 0.0f;
    dL_dcolors[3 * ii] = 0.0f; dL_dcolors[3 * ii + 1] = 0.0f; dL_dcolors[3 * ii + 2] = 0.0f;
    dL_dopacity[i] = 0.0f;
    dL_dmeans3D[3 * ii] = 0.0f; dL_dmeans3D[3 * ii + 1] = 0.0f; dL_dmeans3D[3 * ii + 2] = 0.0f;
#pragma unroll
    for (int k = 0; k < 6; ++k) dL_dcov3D[6 * ii + k] = 0.0f;
    if (!colors_precomp) {
      if (DREG) {
        float4* z4 = reinterpret_cast<float4*>(dL_dsh + ii * M * 3);
        for (int qd = 0; qd < ((3 * M) >> 2); ++qd) z4[qd] = make_float4(0, 0, 0, 0);
      } else {
        float* z = dsh_stage ? dsh_stage : dL_dsh + ii * M * 3;
        for (int k = 0; k < 3 * M; ++k) z[k] = 0.0f;
      }
    }
    if (!cov3D_precomp) {
      dL_dscales[3 * ii] = 0.0f; dL_dscales[3 * ii + 1] = 0.0f; dL_dscales[3 * ii + 2] = 0.0f;
      reinterpret_cast<float4*>(dL_drots)[i] = make_float4(0, 0, 0, 0);
    }
    return;
  }
  const float4 acc0 = reinterpret_cast<const float4*>(grad_acc)[3 * ii];      // conic A, B, C | opacity
  const float4 acc1 = reinterpret_cast<const float4*>(grad_acc)[3 * ii + 1];  // mean2D x, y | colour r, g
  const float4 acc2 = reinterpret_cast<const float4*>(grad_acc)[3 * ii + 2];  // colour b
  const float gcol3[3] = {acc1.z, acc1.w, acc2.x};
  dL_dmean2D[3 * ii] = acc1.x; dL_dmean2D[3 * ii + 1] = acc1.y; dL_dmean2D[3 * ii + 2] = 0.0f;
  dL_dcolors[3 * ii] = gcol3[0]; dL_dcolors[3 * ii + 1] = gcol3[1]; dL_dcolors[3 * ii + 2] = gcol3[2];
  const Mat16 V = load_mat(view);
  const Mat16 PV = load_mat(proj);
  const float p[3] = {means3D[3 * ii], means3D[3 * ii + 1], means3D[3 * ii + 2]};

  float c6[6];
  float4 q = make_float4(0, 0, 0, 0), q_raw = make_float4(0, 0, 0, 0);
  float q_norm = 1.0f;
  float s3[3] = {0, 0, 0};
  if (cov3D_precomp) {
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = cov3D_precomp[6 * ii + k];
  } else {
    s3[0] = scales[3 * ii]; s3[1] = scales[3 * ii + 1]; s3[2] = scales[3 * ii + 2];
    q = reinterpret_cast<const float4*>(rotations)[i];
    if (raw) {
      q_raw = q;
      s3[0] = expf(s3[0]); s3[1] = expf(s3[1]); s3[2] = expf(s3[2]);
      q = act_normalize(q, q_norm);
    }
    cov3d_from_scale_rot(s3, mod, q, c6);
  }

  float dmean[3];
  float dc[6] = {0, 0, 0, 0, 0, 0};
  // (1) conic -> cov2D -> cov3D, view-space point
  {
    float t[3];
    t[0] = V.m[0] * p[0] + V.m[4] * p[1] + V.m[8] * p[2] + V.m[12];
    t[1] = V.m[1] * p[0] + V.m[5] * p[1] + V.m[9] * p[2] + V.m[13];
    t[2] = V.m[2] * p[0] + V.m[6] * p[1] + V.m[10] * p[2] + V.m[14];
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    float abc[3], T[2][3], tc[3];
    bool clx, cly;
    ewa_cov2d(t, fx, fy, tanfovx, tanfovy, c6, V, abc, T, tc, clx, cly);
    const float x_grad_mul = clx ? 0.0f : 1.0f, y_grad_mul = cly ? 0.0f : 1.0f;
    const float a = abc[0] + 0.3f, b = abc[1], c = abc[2] + 0.3f;
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / (denom * denom + 0.0000001f);
    const float gA = acc0.x, gB = acc0.y, gC = acc0.z;
    float dL_da = 0.0f, dL_db = 0.0f, dL_dc = 0.0f;
    if (denom2inv != 0.0f) {
      dL_da = denom2inv * (-c * c * gA + 2.0f * b * c * gB + (denom - a * c) * gC);
      dL_dc = denom2inv * (-a * a * gC + 2.0f * a * b * gB + (denom - a * c) * gA);
      dL_db = denom2inv * 2.0f * (b * c * gA - (denom + 2.0f * b * b) * gB + a * b * gC);
      dc[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
      dc[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
      dc[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
      dc[1] = 2.0f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db +
              2.0f * T[1][0] * T[1][1] * dL_dc;
      dc[2] = 2.0f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db +
              2.0f * T[1][0] * T[1][2] * dL_dc;
      dc[4] = 2.0f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db +
              2.0f * T[1][1] * T[1][2] * dL_dc;
    }
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float dT[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float s0 = T[0][0] * S[0][j] + T[0][1] * S[1][j] + T[0][2] * S[2][j];
      const float s1 = T[1][0] * S[0][j] + T[1][1] * S[1][j] + T[1][2] * S[2][j];
      dT[0][j] = 2.0f * s0 * dL_da + s1 * dL_db;
      dT[1][j] = 2.0f * s1 * dL_dc + s0 * dL_db;
    }
    // W[r][c] = V.m[4c + r]
    const float dJ00 = V.m[0] * dT[0][0] + V.m[4] * dT[0][1] + V.m[8] * dT[0][2];
    const float dJ02 = V.m[2] * dT[0][0] + V.m[6] * dT[0][1] + V.m[10] * dT[0][2];
    const float dJ11 = V.m[1] * dT[1][0] + V.m[5] * dT[1][1] + V.m[9] * dT[1][2];
    const float dJ12 = V.m[2] * dT[1][0] + V.m[6] * dT[1][1] + V.m[10] * dT[1][2];
    const float tz = 1.0f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = x_grad_mul * -fx * tz2 * dJ02;
    const float dty = y_grad_mul * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.0f * fx * tc[0]) * tz3 * dJ02 +
                      (2.0f * fy * tc[1]) * tz3 * dJ12;
    dmean[0] = V.m[0] * dtx + V.m[1] * dty + V.m[2] * dtz;
    dmean[1] = V.m[4] * dtx + V.m[5] * dty + V.m[6] * dtz;
    dmean[2] = V.m[8] * dtx + V.m[9] * dty + V.m[10] * dtz;
  }
  // (2) screen position through the perspective divide
  {
    const float h0 = PV.m[0] * p[0] + PV.m[4] * p[1] + PV.m[8] * p[2] + PV.m[12];
    const float h1 = PV.m[1] * p[0] + PV.m[5] * p[1] + PV.m[9] * p[2] + PV.m[13];
    const float h3 = PV.m[3] * p[0] + PV.m[7] * p[1] + PV.m[11] * p[2] + PV.m[15];
    const float m_w = 1.0f / (h3 + 0.0000001f);
    const float mul1 = h0 * m_w * m_w, mul2 = h1 * m_w * m_w;
    const float g0 = acc1.x, g1 = acc1.y;
    dmean[0] += (PV.m[0] * m_w - PV.m[3] * mul1) * g0 + (PV.m[1] * m_w - PV.m[3] * mul2) * g1;
    dmean[1] += (PV.m[4] * m_w - PV.m[7] * mul1) * g0 + (PV.m[5] * m_w - PV.m[7] * mul2) * g1;
    dmean[2] += (PV.m[8] * m_w - PV.m[11] * mul1) * g0 + (PV.m[9] * m_w - PV.m[11] * mul2) * g1;
  }
  // (3) colour -> SH coefficients (+ position through the view direction when deg > 0)
  if (!colors_precomp) {
    float shr[SHVEC ? 48 : 1];
    const float* sh;
    if constexpr (SHVEC) {
      const float4* src = reinterpret_cast<const float4*>(shs + ii * M * 3);
      const int nq = deg > 0 ? (3 * M) >> 2 : 0;   // (degree 0 reads no coefficient)
#pragma unroll
      for (int qd = 0; qd < 12; ++qd) {
        float4 v = make_float4(0, 0, 0, 0);
        if (qd < nq) v = src[qd];
        shr[4 * qd] = v.x; shr[4 * qd + 1] = v.y; shr[4 * qd + 2] = v.z; shr[4 * qd + 3] = v.w;
      }
      sh = shr;
    } else {
      shr[0] = 0.0f;
      sh = shs + ii * M * 3;
    }
    // SHVEC without a staging row: the gradients are collected in registers and leave as 3 M / 4 dwordx4 stores
    float dshr[DREG ? 48 : 1];
    float* dsh;
    if constexpr (DREG) {
#pragma unroll
      for (int k = 0; k < 48; ++k) dshr[k] = 0.0f;
      dsh = dshr;
    } else {
      dshr[0] = 0.0f;
      dsh = dsh_stage ? dsh_stage : dL_dsh + ii * M * 3;
    }
    const uint32_t cl = clamped[i];
    const float v0 = p[0] - campos_p[0], v1 = p[1] - campos_p[1], v2 = p[2] - campos_p[2];
    const float len = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
    const float x = v0 / len, y = v1 / len, z = v2 / len;
    float ddir[3] = {0.0f, 0.0f, 0.0f};
    // coefficients above the active degree (M > (deg+1)^2, e.g. max_sh_degree 3 with active degree 1) take no part
    // in the colour: their gradient is zero and is written here (the caller does not pre-fill the arrays)
    if constexpr (!DREG)
      for (int k = 3 * (deg + 1) * (deg + 1); k < 3 * M; ++k) dsh[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float gcol = ((cl >> c) & 1u) ? 0.0f : gcol3[c];
#define SHK(k) sh[(k) * 3 + c]
#define DSH(k) dsh[(k) * 3 + c]
      float dx = 0.0f, dy = 0.0f, dz = 0.0f;
      DSH(0) = SH_C0 * gcol;
      if (deg > 0) {
        DSH(1) = -SH_C1 * y * gcol; DSH(2) = SH_C1 * z * gcol; DSH(3) = -SH_C1 * x * gcol;
        dx = -SH_C1 * SHK(3); dy = -SH_C1 * SHK(1); dz = SH_C1 * SHK(2);
        if (deg > 1) {
          const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
          DSH(4) = SH_C2[0] * xy * gcol; DSH(5) = SH_C2[1] * yz * gcol;
          DSH(6) = SH_C2[2] * (2.0f * zz - xx - yy) * gcol;
          DSH(7) = SH_C2[3] * xz * gcol; DSH(8) = SH_C2[4] * (xx - yy) * gcol;
          dx += SH_C2[0] * y * SHK(4) + SH_C2[2] * 2.0f * -x * SHK(6) + SH_C2[3] * z * SHK(7) +
                SH_C2[4] * 2.0f * x * SHK(8);
          dy += SH_C2[0] * x * SHK(4) + SH_C2[1] * z * SHK(5) + SH_C2[2] * 2.0f * -y * SHK(6) +
                SH_C2[4] * 2.0f * -y * SHK(8);
          dz += SH_C2[1] * y * SHK(5) + SH_C2[2] * 4.0f * z * SHK(6) + SH_C2[3] * x * SHK(7);
          if (deg > 2) {
            DSH(9) = SH_C3[0] * y * (3.0f * xx - yy) * gcol;
            DSH(10) = SH_C3[1] * xy * z * gcol;
            DSH(11) = SH_C3[2] * y * (4.0f * zz - xx - yy) * gcol;
            DSH(12) = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * gcol;
            DSH(13) = SH_C3[4] * x * (4.0f * zz - xx - yy) * gcol;
            DSH(14) = SH_C3[5] * z * (xx - yy) * gcol;
            DSH(15) = SH_C3[6] * x * (xx - 3.0f * yy) * gcol;
            dx += SH_C3[0] * SHK(9) * 6.0f * xy + SH_C3[1] * SHK(10) * yz + SH_C3[2] * SHK(11) * -2.0f * xy +
                  SH_C3[3] * SHK(12) * -6.0f * xz + SH_C3[4] * SHK(13) * (4.0f * zz - 3.0f * xx - yy) +
                  SH_C3[5] * SHK(14) * 2.0f * xz + SH_C3[6] * SHK(15) * 3.0f * (xx - yy);
            dy += SH_C3[0] * SHK(9) * 3.0f * (xx - yy) + SH_C3[1] * SHK(10) * xz +
                  SH_C3[2] * SHK(11) * (4.0f * zz - xx - 3.0f * yy) + SH_C3[3] * SHK(12) * -6.0f * yz +
                  SH_C3[4] * SHK(13) * -2.0f * xy + SH_C3[5] * SHK(14) * -2.0f * yz +
                  SH_C3[6] * SHK(15) * -6.0f * xy;
            dz += SH_C3[1] * SHK(10) * xy + SH_C3[2] * SHK(11) * 8.0f * yz +
                  SH_C3[3] * SHK(12) * 3.0f * (2.0f * zz - xx - yy) + SH_C3[4] * SHK(13) * 8.0f * xz +
                  SH_C3[5] * SHK(14) * (xx - yy);
          }
        }
      }
#undef SHK
#undef DSH
      ddir[0] += dx * gcol; ddir[1] += dy * gcol; ddir[2] += dz * gcol;
    }
    if constexpr (DREG) {
      {
        float4* d4 = reinterpret_cast<float4*>(dL_dsh + ii * M * 3);
        const int nq = (3 * M) >> 2;
#pragma unroll
        for (int qd = 0; qd < 12; ++qd)
          if (qd < nq) d4[qd] = make_float4(dshr[4 * qd], dshr[4 * qd + 1], dshr[4 * qd + 2], dshr[4 * qd + 3]);
      }
    }
    if (deg > 0) {
      const float sum2 = v0 * v0 + v1 * v1 + v2 * v2;
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      dmean[0] += ((sum2 - v0 * v0) * ddir[0] - v1 * v0 * ddir[1] - v2 * v0 * ddir[2]) * invsum32;
      dmean[1] += (-v0 * v1 * ddir[0] + (sum2 - v1 * v1) * ddir[1] - v2 * v1 * ddir[2]) * invsum32;
      dmean[2] += (-v0 * v2 * ddir[0] - v1 * v2 * ddir[1] + (sum2 - v2 * v2) * ddir[2]) * invsum32;
    }
  }
  dL_dmeans3D[3 * ii] = dmean[0]; dL_dmeans3D[3 * ii + 1] = dmean[1]; dL_dmeans3D[3 * ii + 2] = dmean[2];
#pragma unroll
  for (int k = 0; k < 6; ++k) dL_dcov3D[6 * ii + k] = dc[k];

  // (4) cov3D -> scale, quaternion
  if (!cov3D_precomp) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z); R[0][1] = 2.0f * (x * y - r * z); R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z); R[1][1] = 1.0f - 2.0f * (x * x + z * z); R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y); R[2][1] = 2.0f * (y * z + r * x); R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    const float s[3] = {mod * s3[0], mod * s3[1], mod * s3[2]};
    float Mm[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) Mm[k][j] = s[k] * R[j][k];
    const float Gs[3][3] = {{dc[0], 0.5f * dc[1], 0.5f * dc[2]},
                            {0.5f * dc[1], dc[3], 0.5f * dc[4]},
                            {0.5f * dc[2], 0.5f * dc[4], dc[5]}};
    float dM[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        dM[k][j] = 2.0f * (Mm[k][0] * Gs[0][j] + Mm[k][1] * Gs[1][j] + Mm[k][2] * Gs[2][j]);
    float Q[3][3];
    float dscale[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      dscale[k] = mod * (R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2]);
#pragma unroll
      for (int j = 0; j < 3; ++j) Q[j][k] = dM[k][j] * s[k];
    }
    if (raw) {  // d exp(x) = exp(x)
      dscale[0] *= s3[0]; dscale[1] *= s3[1]; dscale[2] *= s3[2];
    }
    dL_dscales[3 * ii] = dscale[0]; dL_dscales[3 * ii + 1] = dscale[1]; dL_dscales[3 * ii + 2] = dscale[2];
    float4 dq;
    dq.x = 2.0f * (-z * Q[0][1] + y * Q[0][2] + z * Q[1][0] - x * Q[1][2] - y * Q[2][0] + x * Q[2][1]);
    dq.y = 2.0f * (y * Q[0][1] + z * Q[0][2] + y * Q[1][0] - 2.0f * x * Q[1][1] - r * Q[1][2] + z * Q[2][0] +
                   r * Q[2][1] - 2.0f * x * Q[2][2]);
    dq.z = 2.0f * (-2.0f * y * Q[0][0] + x * Q[0][1] + r * Q[0][2] + x * Q[1][0] + z * Q[1][2] - r * Q[2][0] +
                   z * Q[2][1] - 2.0f * y * Q[2][2]);
    dq.w = 2.0f * (-2.0f * z * Q[0][0] - r * Q[0][1] + x * Q[0][2] + r * Q[1][0] - 2.0f * z * Q[1][1] +
                   y * Q[1][2] + x * Q[2][0] + y * Q[2][1]);
    if (raw) {  // through q_hat = q / max(||q||, eps): (g - q_hat (q_hat . g)) / max(||q||, eps)
      const float dotg = (q.x * dq.x + q.y * dq.y) + (q.z * dq.z + q.w * dq.w);
      const float dn = fmaxf(q_norm, 1e-12f);
      dq = make_float4((dq.x - q.x * dotg) / dn, (dq.y - q.y * dotg) / dn, (dq.z - q.z * dotg) / dn,
                       (dq.w - q.w * dotg) / dn);
      (void)q_raw;
    }
    reinterpret_cast<float4*>(dL_drots)[i] = dq;
  }
  if (raw) {  // sigmoid'(x) = s (1 - s) applied to the blend's dL/d(opacity)
    const float sg = act_sigmoid(opacities_raw[i]);
    dL_dopacity[i] = acc0.w * (sg * (1.0f - sg));
  } else {
    dL_dopacity[i] = acc0.w;
  }
}

#define GGD_PPB_PARAMS                                                                                                  \
    int P, int M, int deg, int W, int H, float tanfovx, float tanfovy, float mod, int raw,                              \
    const float* __restrict__ opacities_raw, float* __restrict__ dL_dopacity, const float* __restrict__ view,           \
    const float* __restrict__ proj, const float* __restrict__ campos_p, const float* __restrict__ means3D,              \
    const float* __restrict__ shs, const float* __restrict__ colors_precomp, const float* __restrict__ scales,          \
    const float* __restrict__ rotations, const float* __restrict__ cov3D_precomp, const int32_t* __restrict__ radii,    \
    const uint8_t* __restrict__ clamped, const float* __restrict__ grad_acc, float* __restrict__ dL_dmean2D,            \
    float* __restrict__ dL_dcolors, float* __restrict__ dL_dmeans3D, float* __restrict__ dL_dcov3D,                     \
    float* __restrict__ dL_dsh, float* __restrict__ dL_dscales, float* __restrict__ dL_drots
#define GGD_PPB_ARGS                                                                                                    \
    P, M, deg, W, H, tanfovx, tanfovy, mod, raw, opacities_raw, dL_dopacity, view, proj, campos_p, means3D, shs,        \
    colors_precomp, scales, rotations, cov3D_precomp, radii, clamped, grad_acc, dL_dmean2D, dL_dcolors, dL_dmeans3D,    \
    dL_dcov3D, dL_dsh, dL_dscales, dL_drots

__global__ __launch_bounds__(256) void preprocess_backward_kernel(GGD_PPB_PARAMS) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  preprocess_backward_body<false>(i, nullptr, GGD_PPB_ARGS);
}

__global__ __launch_bounds__(256) void preprocess_backward_vec_kernel(GGD_PPB_PARAMS) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  preprocess_backward_body<true, true>(i, nullptr, GGD_PPB_ARGS);
}

// SH degree > 0 (M > 1 coefficients per channel): a Gaussian's 3 M gradients are 12 M bytes apart from its neighbour's, so
// written by their owner lane they leave the wave as 3 M store instructions of 64 lone words each (1 M Gaussians, M = 16:
// 494 us for this kernel against 88 us at M = 1).  Here every lane parks its row in LDS and the wave writes its 64 rows --
// one contiguous 768 M-byte span of dL_dsh -- with lane-consecutive (16-byte where 3 M allows) stores.  (1 M Gaussians,
// M = 16: 247 -> 146 us; preprocess_backward_vec_kernel above does as well without LDS where the rows allow dwordx4.)
template <bool SHVEC>
__global__ __launch_bounds__(256) void preprocess_backward_staged_kernel(GGD_PPB_PARAMS) {
  extern __shared__ float s_dsh[];   // [256][3 M + 1]
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rl = 3 * M, rowlen = rl + 1;
  if (i < P) preprocess_backward_body<SHVEC>(i, s_dsh + (size_t)threadIdx.x * rowlen, GGD_PPB_ARGS);
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const int i0 = blockIdx.x * 256 + wv * 64;
  const int nrows = min(64, P - i0);
  if (nrows <= 0) return;
  const float* rows = s_dsh + (size_t)wv * 64 * rowlen;
  float* dst = dL_dsh + (size_t)i0 * rl;
  const int total = nrows * rl;
  if ((rl & 3) == 0) {
    for (int e = 4 * lane; e < total; e += 256) {
      const int r = e / rl, k = e - r * rl;
      const float* src = rows + r * rowlen + k;
      *reinterpret_cast<float4*>(dst + e) = make_float4(src[0], src[1], src[2], src[3]);
    }
  } else {
    for (int e = lane; e < total; e += 64) {
      const int r = e / rl, k = e - r * rl;
      dst[e] = rows[r * rowlen + k];
    }
  }
}

}  // namespace

int ggd_launch_preprocess_backward(ggd_ctx* ctx, hipStream_t s, const ggd_params& prm, const float* means3D,
                                   const float* shs, const float* colors_precomp, const float* opacities,
                                   float* dL_dopacity, const float* scales,
                                   const float* rotations, const float* cov3D_precomp, const int32_t* radii,
                                   const uint8_t* clamped, const float* grad_acc, float* dL_dmean2D,
                                   float* dL_dcolors, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh,
                                   float* dL_dscales, float* dL_drots) {
  if (prm.P == 0) return GGD_OK;
  // the staged form whenever there is more than the band-0 coefficient per channel (and room: 256 rows of 3 M + 1 floats)
  const bool staged = !colors_precomp && prm.M > 1 && (size_t)256 * (3 * prm.M + 1) * sizeof(float) <= 64 * 1024;
  const bool shvec = staged && prm.M <= 16 && ((3 * prm.M) & 3) == 0;
  // rows that are a multiple of 16 bytes (M = 4, 8, 12, 16): read and written from registers with dwordx4 accesses (145 us
  // at M = 16, 1 M Gaussians; the LDS-staged form measures 152 with the same loads and is kept for the other M)
  if (shvec)
    hipLaunchKernelGGL(preprocess_backward_vec_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, s, prm.P, prm.M,
                       prm.sh_degree, prm.width, prm.height, prm.tanfovx, prm.tanfovy, prm.scale_modifier,
                       prm.raw_attributes, opacities, dL_dopacity, prm.viewmatrix, prm.projmatrix, prm.campos, means3D, shs, colors_precomp, scales, rotations,
                       cov3D_precomp, radii, clamped, grad_acc, dL_dmean2D, dL_dcolors, dL_dmeans3D, dL_dcov3D,
                       dL_dsh, dL_dscales, dL_drots);
  else if (staged)
    hipLaunchKernelGGL(preprocess_backward_staged_kernel<false>, dim3((prm.P + 255) / 256), dim3(256),
                       (size_t)256 * (3 * prm.M + 1) * sizeof(float), s, prm.P, prm.M,
                       prm.sh_degree, prm.width, prm.height, prm.tanfovx, prm.tanfovy, prm.scale_modifier,
                       prm.raw_attributes, opacities, dL_dopacity, prm.viewmatrix, prm.projmatrix, prm.campos, means3D, shs, colors_precomp, scales, rotations,
                       cov3D_precomp, radii, clamped, grad_acc, dL_dmean2D, dL_dcolors, dL_dmeans3D, dL_dcov3D,
                       dL_dsh, dL_dscales, dL_drots);
  else
  hipLaunchKernelGGL(preprocess_backward_kernel, dim3((prm.P + 255) / 256), dim3(256), 0, s, prm.P, prm.M,
                     prm.sh_degree, prm.width, prm.height, prm.tanfovx, prm.tanfovy, prm.scale_modifier,
                     prm.raw_attributes, opacities, dL_dopacity, prm.viewmatrix, prm.projmatrix, prm.campos, means3D, shs, colors_precomp, scales, rotations,
                     cov3D_precomp, radii, clamped, grad_acc, dL_dmean2D, dL_dcolors, dL_dmeans3D, dL_dcov3D,
                     dL_dsh, dL_dscales, dL_drots);
  GGD_HIP(hipGetLastError());
  return GGD_OK;
}
