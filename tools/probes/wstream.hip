// wstream.hip - probe: how fast can every CU stream the SAME fragment-packed weight matrix L2 -> VGPR (global_load_dwordx4, 1 KB per
// wave instruction, fully contiguous) while feeding v_mfma_f32_32x32x16_bf16 whose other operand is stationary?  This sizes the
// "x-stationary" GEMM family (a 32-row activation slab per workgroup stays in LDS / registers, only W moves).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wstream.hip -o build/wstream && build/wstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// W: nfrag fragments of 1 KB; workgroup = NW waves; wave w takes fragments w, w + NW, ... ; UNROLL loads in flight per wave
template <int NW, int UNROLL, bool MFMA>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const u32x4* __restrict__ W, int nfrag, float* out, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc[2] = {};
  unsigned keep = 0;
  bf16x8 xb;
  for (int e = 0; e < 8; ++e) xb[e] = (__bf16)(0.001f * (lane + e));
  for (int r = 0; r < reps; ++r) {
    for (int f0 = wave; f0 < nfrag; f0 += NW * UNROLL) {
      u32x4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int f = f0 + u * NW;
        v[u] = f < nfrag ? W[(long)f * 64 + lane] : u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (MFMA) {
          bf16x8 a = __builtin_bit_cast(bf16x8, v[u]);
          acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb, acc[u & 1], 0, 0, 0);
        } else {
          keep ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
        }
      }
    }
  }
  float s = (float)keep;
  for (int i = 0; i < 16; ++i) s += acc[0][i] + acc[1][i];
  if (s == 12345.678f) out[blockIdx.x] = s;
}

template <int NW, int UNROLL, bool MFMA>
static void run(const u32x4* W, size_t bytes, float* out, int nwg, const char* what) {
  const int nfrag = (int)(bytes / 1024), reps = 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((stream_kernel<NW, UNROLL, MFMA>), dim3(nwg), dim3(NW * 64), 0, 0, W, nfrag, out, reps);
  hipEventRecord(e0);
  const int n = 10;
  for (int it = 0; it < n; ++it) hipLaunchKernelGGL((stream_kernel<NW, UNROLL, MFMA>), dim3(nwg), dim3(NW * 64), 0, 0, W, nfrag, out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double t = ms * 1e-3 / n;
  const double per_cu = (double)bytes * reps / t;      // every workgroup (one per CU when nwg = 256) streams all of W `reps` times
  printf("%-46s W %5.2f MB  wgs %4d  %8.1f us per pass  %6.1f GB/s per WG = %5.1f B/clk @2.4GHz  chip %6.2f TB/s%s\n", what, bytes / 1e6, nwg,
         t / reps * 1e6, per_cu / 1e9, per_cu / 2.4e9, per_cu * nwg / 1e12,
         MFMA ? "" : "  (loads only)");
}

int main() {
  const size_t maxb = 8u << 20;
  u32x4* W;
  float* out;
  hipMalloc(&W, maxb);
  hipMalloc(&out, 4096 * 4);
  std::vector<unsigned short> h(maxb / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (unsigned short)(rand() & 0xff);
  hipMemcpy(W, h.data(), maxb, hipMemcpyHostToDevice);
  for (size_t bytes : {(size_t)128 << 10, (size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20}) {
    run<8, 4, false>(W, bytes, out, 256, "8 waves x4 in flight, loads only");
    run<8, 4, true>(W, bytes, out, 256, "8 waves x4 in flight, + MFMA 32x32x16");
    run<8, 8, true>(W, bytes, out, 256, "8 waves x8 in flight, + MFMA");
    run<16, 4, true>(W, bytes, out, 256, "16 waves x4 in flight, + MFMA");
    run<16, 8, true>(W, bytes, out, 256, "16 waves x8 in flight, + MFMA");
    run<4, 8, true>(W, bytes, out, 256, "4 waves x8 in flight, + MFMA");
    run<8, 8, true>(W, bytes, out, 64, "8 waves x8, 64 workgroups (decoder: 1 per clip)");
    run<16, 8, true>(W, bytes, out, 64, "16 waves x8, 64 workgroups");
    run<8, 8, true>(W, bytes, out, 512, "8 waves x8, 512 workgroups (2 per CU)");
  }
  return 0;
}
