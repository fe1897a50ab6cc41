// Probe of v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) through inline asm: prints what each lane holds.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float a = (float)l, b = (float)(100 + l);
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  out[l] = a; out[64 + l] = b;
  float c = (float)l, d = (float)(100 + l);
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));
  out[128 + l] = c; out[192 + l] = d;
  // the cross-row stage of the transposed reduction: x0, x1 hold per-row partial sums; lanes 0-31 want the 4-row sum of x0,
  // lanes 32-63 the 4-row sum of x1
  float x0 = (float)(1000 + (l >> 4)), x1 = (float)(2000 + 10 * (l >> 4));   // row sums: x0 rows 1000..1003 -> 4006, x1 -> 8060
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x0), "+v"(x1));
  float y = x0 + x1, t = y;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(y), "+v"(t));
  out[256 + l] = y + t;
}
int main() {
  float* d; hipMalloc(&d, 320 * 4); k<<<1, 64>>>(d); float h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"swap32: a (was l)", "swap32: b (was 100+l)", "swap16: c (was l)", "swap16: d (was 100+l)", "cross-row sum (expect 4006 x32, 8060 x32)"};
  for (int s = 0; s < 5; ++s) { printf("%s\n", names[s]); for (int l = 0; l < 64; ++l) printf("%g%s", h[64 * s + l], (l & 15) == 15 ? "\n" : " "); }
  return 0;
}
