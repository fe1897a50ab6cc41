// Probe of the transposed wave reduction used by blend_backward_wg_kernel: prints, per lane, what each step produced.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = (float)((i + 1) * 1000) + (float)l * 0.0f + 1.0f;   // every lane contributes 1 + 1000(i+1)
  // variant with lane-dependent weights to identify lanes: v_i[l] = 2^i ... use small integers: v_i[l] = (i+1)
  for (int i = 0; i < 8; ++i) v[i] = (float)((i + 1) * 1000 + l);   // totals: 64000 (i + 1) + 2016
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  out[l] = v[0]; out[64 + l] = v[2];
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %4, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  out[128 + l] = v[0];
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  out[192 + l] = v[0];
  float x0 = v[0], x1 = v[4];
  x0 += __shfl_xor(x0, 16, 64); x1 += __shfl_xor(x1, 16, 64);
  x0 += __shfl_xor(x0, 32, 64); x1 += __shfl_xor(x1, 32, 64);
  const float z = (l & 32) ? x1 : x0;
  out[256 + l] = z;
}
int main() {
  float* d; hipMalloc(&d, 320 * 4); k<<<1, 64>>>(d); float h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"after xor8: v0", "after xor8: v2", "after xor4: v0", "after quads: v0", "final"};
  for (int s = 0; s < 5; ++s) { printf("%s\n", names[s]); for (int l = 0; l < 64; ++l) printf("%g%s", h[64 * s + l], (l & 15) == 15 ? "\n" : " "); }
  printf("expected final: lanes 0-31 banks {v0,v2,v1,v3} x64 = {64,192,128,256}; lanes 32-63 {v4,v6,v5,v7} x64 = {320,448,384,512}\n");
  return 0;
}
