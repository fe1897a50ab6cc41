// Probe: do v_mfma_f32_16x16x32_bf16 and plain VALU work overlap on one gfx950 SIMD -- (a) issued by DIFFERENT waves of the
// SIMD, (b) interleaved inside ONE wave's instruction stream?  Reports SIMD cycles per loop iteration for: MFMA only, VALU
// only, both from different waves, both interleaved in every wave -- with 1 and 2 waves per SIMD.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_valu_overlap_probe.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// per iteration: NM MFMAs on 4 independent accumulators and / or NV v_fma_f32 (or v_pk_fma_f32) on 8 independent chains
// MODE 0: MFMA only   1: VALU only   2: odd waves MFMA, even waves VALU   3: every wave both, interleaved 1 MFMA : R VALU
// MODE 4: like 2, but the role is taken from the wave's slot ON ITS SIMD (HW_ID.wave_id parity), and the kernel records
//         (simd, role) of every wave so that the co-residency of one MFMA and one VALU wave per SIMD is checked, not assumed
template <int MODE, int R, bool PK>
__global__ __launch_bounds__(1024) void probe(float* out, int iters, float seed) {
  const int wv = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
  f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v0 = seed + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  const float k = 0.999f, c = 1e-3f;
  unsigned hwid = 0;
  if (MODE == 4) asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const int slot = hwid & 15, simd = (hwid >> 4) & 3;
  const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && (wv & 1)) || (MODE == 4 && (slot & 1));
  const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && !(wv & 1)) || (MODE == 4 && !(slot & 1));
  if (MODE == 4 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) { out[512 + wv] = (float)(simd * 100 + slot * 2 + (do_m ? 1 : 0)); }
#define MF(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#define VF4A asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" \
                          : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(k), "v"(c));
#define VF4B asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" \
                          : "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(k), "v"(c));
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7};
  const f2 k2 = {k, k}, c2v = {c, c};
#define PF4 asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" \
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k2), "v"(c2v));
  for (int i = 0; i < iters; ++i) {
    if (MODE == 3) {
      // 16 MFMAs, each followed by R VALU instructions x 4 (R = 1 -> 4 VALU per MFMA ... the MFMA occupies its pipe 16
      // cycles = 4 issue slots)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        MF(c0) if (R >= 1) { if (PK) { PF4 } else { VF4A } } if (R >= 2) { if (PK) { PF4 } else { VF4B } }
        MF(c1) if (R >= 1) { if (PK) { PF4 } else { VF4A } } if (R >= 2) { if (PK) { PF4 } else { VF4B } }
        MF(c2) if (R >= 1) { if (PK) { PF4 } else { VF4A } } if (R >= 2) { if (PK) { PF4 } else { VF4B } }
        MF(c3) if (R >= 1) { if (PK) { PF4 } else { VF4A } } if (R >= 2) { if (PK) { PF4 } else { VF4B } }
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { MF(c0) MF(c1) MF(c2) MF(c3) }
      }
      if (do_v) {
#pragma unroll
        for (int j = 0; j < 16 * R; ++j) { if (PK) { PF4 } else { if (j & 1) { VF4B } else { VF4A } } }
      }
    }
  }
  float r = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + p0.x + p1.y + p2.x + p3.y;
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int R, bool PK>
static void run(const char* name, int waves_per_simd, float* d) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // one workgroup per CU up to 4 waves per SIMD, two workgroups of 16 waves above
  const int wgs = waves_per_simd > 4 ? 2 : 1;
  const int threads = 256 * waves_per_simd / wgs;
  hipLaunchKernelGGL((probe<MODE, R, PK>), dim3(256 * wgs), dim3(threads), 0, 0, d, 100, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<MODE, R, PK>), dim3(256 * wgs), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || hipGetLastError() != hipSuccess) printf("launch failed: ");
  // what a SIMD was asked for per iteration
  const int mf = (MODE == 1 ? 0 : 16), vf = (MODE == 0 ? 0 : 64 * R);
  if (MODE == 4) { float h[16]; hipMemcpy(h, d + 512, sizeof(h), hipMemcpyDeviceToHost); printf("   wave -> simd*100 + slot*2 + is_mfma:"); for (int w = 0; w < threads / 64 && w < 16; ++w) printf(" %d", (int)h[w]); printf("\n"); }
  double waves_m = MODE == 2 || MODE == 4 ? waves_per_simd / 2.0 : (MODE == 1 ? 0 : waves_per_simd);
  double waves_v = MODE == 2 || MODE == 4 ? waves_per_simd / 2.0 : (MODE == 0 ? 0 : waves_per_simd);
  const double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("%-44s waves/SIMD=%d  %8.1f cycles/iter   (asked per SIMD and iter: %4.0f MFMA = %5.0f cyc at 16, %5.0f VALU = %5.0f cyc at 4%s)\n",
         name, waves_per_simd, cyc, mf * waves_m, mf * waves_m * 16, vf * waves_v, vf * waves_v * (PK ? 8 : 4), PK ? "x2 (pk)" : "");
}

int main() {
  float* d;
  hipMalloc(&d, 8192);
  for (int w : {1, 2, 4, 8}) {
    run<0, 1, false>("MFMA only", w, d);
    run<1, 1, false>("VALU only (64 v_fma)", w, d);
    run<1, 2, false>("VALU only (128 v_fma)", w, d);
    if (w >= 2) {
      run<2, 1, false>("odd waves MFMA, even waves 64 v_fma", w, d);
      run<2, 2, false>("odd waves MFMA, even waves 128 v_fma", w, d);
      run<4, 1, false>("same SIMD: odd slots MFMA, even slots 64 v_fma", w, d);
      run<4, 2, false>("same SIMD: odd slots MFMA, even slots 128 v_fma", w, d);
      run<4, 1, true>("same SIMD: odd slots MFMA, even slots 64 v_pk_fma", w, d);
    }
    run<3, 1, false>("one wave: 1 MFMA : 4 v_fma interleaved", w, d);
    run<3, 2, false>("one wave: 1 MFMA : 8 v_fma interleaved", w, d);
    run<1, 1, true>("VALU only (64 v_pk_fma)", w, d);
    run<3, 1, true>("one wave: 1 MFMA : 4 v_pk_fma interleaved", w, d);
    if (w >= 2) run<2, 1, true>("odd waves MFMA, even waves 64 v_pk_fma", w, d);
  }
  return 0;
}
