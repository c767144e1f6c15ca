// probe: lane semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  // every lane reads 8 bytes at lds[4*lane .. 4*lane+3]  (M[lane][e] = 4*lane + e)
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * threadIdx.x));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)r[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 256 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) {
      int expect = 4 * (16 * (l >> 4) + 4 * j + ((l & 15) >> 2)) + (l & 3);
      printf(" %4d%s", h[l * 4 + j], h[l * 4 + j] == expect ? "" : "!");
      bad += h[l * 4 + j] != expect;
    }
    printf("\n");
  }
  printf("hypothesis R[l][j] = M[16*(l>>4) + 4j + ((l&15)>>2)][l&3]: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  return 0;
}
