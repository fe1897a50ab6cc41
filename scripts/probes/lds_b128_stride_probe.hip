// Probe: cycles per ds_read_b128 for the fused decoder's weight-fragment access pattern -- lane (i = lane & 15,
// g = lane >> 4) reads 16 bytes at  i * STRIDE + g * 16  (+ a per-iteration offset) -- as a function of the row stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(int stride, int iters, float* out, long long* cyc) {
  extern __shared__ unsigned char lds[];
  const int lane = threadIdx.x & 63;
  for (int b = threadIdx.x; b < 16384; b += blockDim.x) reinterpret_cast<float*>(lds)[b] = (float)b;
  __syncthreads();
  const int i = lane & 15, g = lane >> 4;
  const unsigned char* p = lds + i * stride + g * 16;
  f4 acc = {0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f4 v = *reinterpret_cast<const f4*>(p + u * 64 + (it & 3) * 4096);
      acc += v;
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
  const int strides[] = {256, 272, 288, 304, 320, 264, 280, 296, 336, 384, 528, 544};
  for (int waves : {1, 8}) {
    for (int st : strides) {
      const int iters = 2000;
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 65536, 0, st, iters, out, cyc);
      hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 65536, 0, st, iters, out, cyc);
      hipDeviceSynchronize();
      std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double m = 0; for (auto v : h) m += v; m /= 256;
      printf("waves/CU-block %d  stride %4d B: %.1f cycles per ds_read_b128 per wave (block of %d waves)\n", waves, st, m / (iters * 8.0), waves);
    }
  }
  return 0;
}
