// Probe: what a global atomicAdd costs as a function of how many of them hit the same word.
//   every thread of G workgroups x 256 threads issues K non-returning atomicAdd(u32), addresses spread over A words
//   (word = (global thread id * 2654435761) % A): A = 1 ... all on one word, A >= threads ... every atomic its own word.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_rate_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_atomics(uint32_t* w, uint32_t A, int K, int lanes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if ((int)(threadIdx.x & 63) >= lanes) return;
  for (int k = 0; k < K; ++k) atomicAdd(&w[(uint32_t)((t + (uint32_t)k * 7919u) * 2654435761u) % A], 1u);
}
__global__ __launch_bounds__(256) void k_atomic_max(int* w, uint32_t A, int K, int lanes) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if ((int)(threadIdx.x & 63) >= lanes) return;
  for (int k = 0; k < K; ++k) atomicMax(&w[(uint32_t)((t + (uint32_t)k * 7919u) * 2654435761u) % A], (int)(t & 31));
}
// the backward blend's flush: 63 lanes = 7 records x 9 consecutive floats of a 48-byte accumulator row, rows at random
template <int STRIDE>
__global__ __launch_bounds__(256) void k_spans(float* w, uint32_t rows, int K) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  if (lane >= 63) return;
  const uint32_t rec = lane / 9, comp = lane % 9;
  for (int k = 0; k < K; ++k) {
    const uint32_t row = (uint32_t)(((t / 64) * 7 + rec + (uint32_t)k * 7919u) * 2654435761u) % rows;
    atomicAdd(&w[(size_t)row * STRIDE + comp], 1.0f);
  }
}
template <typename F>
static float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  uint32_t* w; (void)hipMalloc(&w, 64u << 20); (void)hipMemset(w, 0, 64u << 20);
  const uint32_t As[] = {1u, 16u, 1024u, 65536u, 16u << 20};
  printf("%10s %8s %6s %6s %12s %10s %14s %16s\n", "words", "WGs", "K", "lanes", "atomics", "us", "G atomics/s", "ns per same-word");
  for (uint32_t A : As)
    for (int G : {256, 1024, 4096})
      for (int lanes : {1, 64}) {
        const int K = 4;
        const double n = (double)G * 4 * lanes * K;
        const float ms = timeit([&] { k_atomics<<<G, 256>>>(w, A, K, lanes); });
        printf("%10u %8d %6d %6d %12.0f %10.1f %14.2f %16.1f\n", A, G, K, lanes, n, ms * 1e3, n / (ms * 1e-3) / 1e9, ms * 1e6 / (n / A));
      }
  printf("float spans of 9 words on random 48-byte rows (7 per wave instruction), K = 8 per thread\n");
  for (uint32_t rows : {4096u, 350000u, 1000000u})
    for (int G : {1024, 4096, 16384}) {
      const double n = (double)G * 4 * 63 * 8;
      const float ms = timeit([&] { k_spans<12><<<G, 256>>>((float*)w, rows, 8); });
      const float ms16 = timeit([&] { k_spans<16><<<G, 256>>>((float*)w, rows, 8); });
      printf("rows %8u WGs %6d lane-atomics %12.0f | 48-byte rows %8.1f us %7.2f G spans/s | 64-byte rows (no span straddles a line) %8.1f us %7.2f G spans/s\n",
             rows, G, n, ms * 1e3, n / 9 / (ms * 1e-3) / 1e9, ms16 * 1e3, n / 9 / (ms16 * 1e-3) / 1e9);
    }
  printf("atomicMax (value rarely changes), one word\n");
  for (int G : {256, 1024, 4096}) {
    const double n = (double)G * 4 * 4;
    const float ms = timeit([&] { k_atomic_max<<<G, 256>>>((int*)w, 1u, 4, 1); });
    printf("%10u %8d %6d %6d %12.0f %10.1f %14.2f %16.1f\n", 1u, G, 4, 1, n, ms * 1e3, n / (ms * 1e-3) / 1e9, ms * 1e6 / n);
  }
  return 0;
}
