#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds + threadIdx.x * 4));
  *(bf16x4*)(out + threadIdx.x * 4) = v;
}
int main() {
  unsigned short h[1024], o[256]; for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned short *d, *e; hipMalloc(&d, 2048); hipMalloc(&e, 512); hipMemcpy(d, h, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, e); hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) if (l < 4 || l % 16 == 0 || l == 17) printf("lane %d: %d %d %d %d\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3]);
}
