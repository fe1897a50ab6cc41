// Probe: issue cost (SIMD cycles per wave64 instruction) of the VALU instruction classes the blend kernels use.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/probes/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float k = 0.999f, c = 1e-3f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // v_fma_f32, 8 independent chains
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c));)
    } else if (KIND == 1) {  // v_pk_fma_f32, 4 independent chains (8 floats)
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((f2){k, k}), "v"((f2){c, c}));)
    } else if (KIND == 2) {  // v_exp_f32
      REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                         "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 3) {  // v_cndmask_b32 with vcc
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");)
    } else if (KIND == 4) {  // v_cmp_ge_f32 -> sgpr pair (VOP3)
      REP16(asm volatile("v_cmp_ge_f32 s[20:21], %0, %1\n v_cmp_ge_f32 s[22:23], %1, %2\n v_cmp_ge_f32 s[24:25], %2, %3\n v_cmp_ge_f32 s[26:27], %3, %4\n"
                         "v_cmp_ge_f32 s[20:21], %4, %5\n v_cmp_ge_f32 s[22:23], %5, %6\n v_cmp_ge_f32 s[24:25], %6, %7\n v_cmp_ge_f32 s[26:27], %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    } else if (KIND == 5) {  // v_pk_mul_f32
      REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((f2){k, k}));)
    } else if (KIND == 6) {  // v_mul_f32 (VOP2)
      REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 7) {  // ds_read_b128 broadcast (same address for every lane)
      __shared__ float4 s[64];
      if (i == 0) { s[threadIdx.x & 63] = make_float4(a0, a1, a2, a3); __syncthreads(); }
      float4 r0, r1, r2, r3;
      REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4 offset:48\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(0));)
      a0 += r0.x + r1.y + r2.z + r3.w;
    } else if (KIND == 8) {  // v_rcp_f32
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                         "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 9) {  // v_add_f32 dpp row_shr
      REP16(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                         "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 10) {  // v_cndmask_b32_e64 with an SGPR-pair mask (what hipcc emits)
      REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                         "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "s20", "s21");)
    } else if (KIND == 11) {  // v_and_b32
      REP16(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n"
                         "v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 12) {  // v_max_f32
      REP16(asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n"
                         "v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 13) {  // v_mov_b32
      REP16(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                         "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 14) {  // v_cmp_ge_f32 -> vcc (VOP2 / VOPC encoding)
      REP16(asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %2\n v_cmp_ge_f32 vcc, %2, %3\n v_cmp_ge_f32 vcc, %3, %4\n"
                         "v_cmp_ge_f32 vcc, %4, %5\n v_cmp_ge_f32 vcc, %5, %6\n v_cmp_ge_f32 vcc, %6, %7\n v_cmp_ge_f32 vcc, %7, %0"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
    } else if (KIND == 15) {  // s_and_b64 (SALU)
      REP16(asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]\n"
                         "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[24:25], s[24:25], s[22:23]\n s_and_b64 s[26:27], s[26:27], s[22:23]\n s_and_b64 s[28:29], s[28:29], s[22:23]"
                         : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc");)
    } else if (KIND == 16) {  // v_add_u32
      REP16(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 17) {  // v_cndmask_b32 (VOP2, vcc) after VCC was written by an SALU op first
      asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k) : "vcc");)
    } else if (KIND == 18) {  // v_bfi_b32 (3-operand bit select: a VGPR lane mask instead of an SGPR condition)
      REP16(asm volatile("v_bfi_b32 %0, %8, %0, %8\n v_bfi_b32 %1, %8, %1, %8\n v_bfi_b32 %2, %8, %2, %8\n v_bfi_b32 %3, %8, %3, %8\n"
                         "v_bfi_b32 %4, %8, %4, %8\n v_bfi_b32 %5, %8, %5, %8\n v_bfi_b32 %6, %8, %6, %8\n v_bfi_b32 %7, %8, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));)
    } else if (KIND == 19) {  // ds_read_b128, per-lane addresses (48-byte stride, random-ish record order)
      __shared__ float4 s2[64 * 3];
      if (i == 0) { for (int q = 0; q < 3; ++q) s2[(threadIdx.x & 63) * 3 + q] = make_float4(a0, a1, a2, a3); __syncthreads(); }
      float4 r0, r1, r2, r3;
      const int addr = (int)((threadIdx.x * 37u) & 63u) * 48;
      REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n ds_read_b128 %2, %4 offset:32\n ds_read_b128 %3, %4\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr));)
      a0 += r0.x + r1.y + r2.z + r3.w;
    } else if (KIND == 20) {  // v_cmp_class / v_cmpx? plain v_cmp_lt_f32 e64 against an inline constant
      REP16(asm volatile("v_cmp_lt_f32 s[20:21], 0, %0\n v_cmp_lt_f32 s[22:23], 0, %1\n v_cmp_lt_f32 s[24:25], 0, %2\n v_cmp_lt_f32 s[26:27], 0, %3\n"
                         "v_cmp_lt_f32 s[20:21], 0, %4\n v_cmp_lt_f32 s[22:23], 0, %5\n v_cmp_lt_f32 s[24:25], 0, %6\n v_cmp_lt_f32 s[26:27], 0, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int KIND>
static void run(const char* name, int wavesPerSimd) {
  const int CUS = 256, blocks = CUS * wavesPerSimd;  // 256-thread blocks = 4 waves = 1 per SIMD
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_wave = (double)iters * 16 * 8 * ((KIND == 7 || KIND == 19) ? 0.5 : 1.0);
  const double insts_per_simd = insts_per_wave * wavesPerSimd;
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per instr per SIMD = %.2f cycles @2.4GHz\n", name, wavesPerSimd, ms,
         ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 8}) {
    run<0>("v_fma_f32", w); run<6>("v_mul_f32 (VOP2)", w); run<1>("v_pk_fma_f32", w); run<5>("v_pk_mul_f32", w);
    run<2>("v_exp_f32", w); run<8>("v_rcp_f32", w); run<3>("v_cndmask_b32", w); run<4>("v_cmp_ge_f32 -> sgpr", w);
    run<9>("v_add_f32_dpp row_shr", w); run<7>("ds_read_b128 broadcast", w);
    run<10>("v_cndmask_b32_e64 sgpr-pair", w); run<17>("v_cndmask_b32 vcc (vcc set)", w); run<11>("v_and_b32", w);
    run<12>("v_max_f32", w); run<13>("v_mov_b32", w); run<14>("v_cmp_ge_f32 -> vcc", w); run<15>("s_and_b64", w);
    run<16>("v_add_u32", w); run<18>("v_bfi_b32", w); run<19>("ds_read_b128 per-lane 48B stride", w);
    run<20>("v_cmp_lt_f32 e64 imm", w);
  }
  return 0;
}
