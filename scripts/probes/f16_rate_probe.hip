// Probe: issue cost (SIMD cycles per wave64 instruction) of the packed-f16 VALU classes the f16 form of the decoder's GELU
// uses, beside v_fma_f32 / v_pk_fma_f32 as the yardsticks.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/probes/f16_rate_probe.hip -o /tmp/f16_probe && /tmp/f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
#define OP8_3(op) op " %0, %0, %8, %9\n" op " %1, %1, %8, %9\n" op " %2, %2, %8, %9\n" op " %3, %3, %8, %9\n" \
                  op " %4, %4, %8, %9\n" op " %5, %5, %8, %9\n" op " %6, %6, %8, %9\n" op " %7, %7, %8, %9"
#define OP8_2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" \
                  op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8"
#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
template <int KIND>
__global__ __launch_bounds__(256) void probe(uint32_t* out, int iters, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {1.0f, 2.0f}, p1 = {3.0f, 4.0f}, p2 = {5.0f, 6.0f}, p3 = {7.0f, 8.0f};
  const uint32_t k = 0x3bff3bffu, c = 0x14001400u;   // packed halves ~0.9995, small
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { REP16(asm volatile(OP8_3("v_fma_f32") : REGS : "v"(0.999f), "v"(1e-3f));) }
    else if (KIND == 1) {
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"((f2){0.999f, 0.999f}), "v"((f2){1e-3f, 1e-3f}));)
    }
    else if (KIND == 2) { REP16(asm volatile(OP8_3("v_pk_fma_f16") : REGS : "v"(k), "v"(c));) }
    else if (KIND == 3) { REP16(asm volatile(OP8_2("v_pk_mul_f16") : REGS : "v"(k));) }
    else if (KIND == 4) { REP16(asm volatile(OP8_2("v_pk_max_f16") : REGS : "v"(k));) }
    else if (KIND == 5) { REP16(asm volatile(OP8_2("v_pk_min_f16") : REGS : "v"(k));) }
    else if (KIND == 6) { REP16(asm volatile(OP8_2("v_pk_add_f16") : REGS : "v"(k));) }
    else if (KIND == 7) { REP16(asm volatile(OP8_2("v_cvt_pk_f16_f32") : REGS : "v"(k));) }
    else if (KIND == 8) { REP16(asm volatile(OP8_2("v_cvt_pkrtz_f16_f32") : REGS : "v"(k));) }
    else if (KIND == 9) { REP16(asm volatile(OP8_2("v_cvt_pk_bf16_f32") : REGS : "v"(k));) }
    else if (KIND == 10) { REP16(asm volatile(OP8_2("v_and_b32") : REGS : "v"(k));) }
    else if (KIND == 11) {  // v_pk_fma_f16 with an SGPR-held constant pair (literal coefficients of a polynomial)
      REP16(asm volatile(OP8_3("v_pk_fma_f16") : REGS : "v"(k), "s"(c));)
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y);
}
template <int KIND>
static void run(const char* name, int wavesPerSimd) {
  const int CUS = 256, blocks = CUS * wavesPerSimd;  // 256-thread blocks = 4 waves = 1 per SIMD
  uint32_t* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<KIND><<<blocks, 256>>>(out, 10, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<KIND><<<blocks, 256>>>(out, iters, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 16 * 8 * wavesPerSimd;
  printf("%-34s waves/SIMD=%d  %.3f ms  -> %.2f cycles per instr per SIMD @2.4GHz\n", name, wavesPerSimd, ms,
         ms * 1e6 / insts_per_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 8}) {
    run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_fma_f16", w); run<11>("v_pk_fma_f16 (sgpr const)", w);
    run<3>("v_pk_mul_f16", w); run<4>("v_pk_max_f16", w); run<5>("v_pk_min_f16", w); run<6>("v_pk_add_f16", w);
    run<7>("v_cvt_pk_f16_f32", w); run<8>("v_cvt_pkrtz_f16_f32", w); run<9>("v_cvt_pk_bf16_f32", w); run<10>("v_and_b32", w);
  }
  return 0;
}
