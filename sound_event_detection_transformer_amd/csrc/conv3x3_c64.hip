// conv3x3_c64.hip - direct 3x3 convolution (stride 1, pad 1) over NHWC bf16 tokens with 64 input and 64 output channels and
// a 16-pixel-wide map: conv2 of the three layer1 Bottlenecks of the SEDT backbone (torchvision resnet50 layer1 at 125 x 16),
// forward and - with the taps flipped and the dgrad weight layout - its input gradient.
//
// Why not the implicit GEMM (igemm3.hip): with Cin = 64 a K tile of the GEMM is exactly one tap, so every input pixel travels
// L2 -> LDS nine times (147 MB per launch at B = 64: the kernel sat at 25 us forward / 39 us dgrad, L2-bound, for a 9.4 GFLOP
// problem whose HBM floor is 5 us).  Here a workgroup stages a 16-row x 16-column output tile's 18 x 18 halo ONCE (one tile
// ahead, through registers) and keeps all nine 64 x 64 weight taps (72 KB) in LDS for its whole life (persistent workgroups),
// so a pixel is read 1.27 times and the nine taps are nine LDS offsets of the same fragment address.
//
// Workgroup = 8 waves; wave w owns output rows 2w, 2w+1 of the tile (32 pixels) x 64 channels: per k16 step one A fragment and
// two B fragments (ds_read_b128, 144-byte pixel / weight-row pitch: conflict-free) feed two v_mfma_f32_32x32x16_bf16.
// Epilogue through LDS: FrozenBN scale / bias + ReLU (forward) or the ReLU mask of the consumer (dgrad), 16-byte stores.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace sedt {

constexpr int C3_W = 16, C3_C = 64, C3_TR = 16;               // map width, channels, output rows per tile
constexpr int C3_HW = C3_W + 2, C3_HR = C3_TR + 2;            // halo tile 18 x 18 pixels
constexpr int C3_PP = C3_C * 2 + 16;                          // 144-byte pitch of a pixel / weight row in LDS
constexpr int C3_HALO = C3_HR * C3_HW * C3_PP;                // 46656 B
constexpr int C3_WL = 9 * C3_C * C3_PP;                       // 82944 B
constexpr int C3_CHUNKS = C3_HR * C3_HW * 8;                  // 2592 16-byte chunks per halo tile
constexpr int C3_PRE = (C3_CHUNKS + 511) / 512;               // 6 per thread

typedef int c3_i4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int c3_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(512) void conv3x3_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, int flip,
                                                          const float* __restrict__ scale, const float* __restrict__ bias, int relu,
                                                          const bf16_t* __restrict__ mask, bf16_t* __restrict__ y, int H, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;                          // [9 taps][64 out][64 in] bf16, 144-byte rows
  unsigned char* halo = smem + C3_WL;                // [18][18] pixels x 64 channels, 144-byte pixels; reused as the output stage
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, hf = lane >> 5, n = lane & 31;
  // ---- the nine weight taps, once per workgroup: w is [64 out][9 taps][64 in]; LDS tap t holds source tap (flip ? 8 - t : t)
  for (int i = tid; i < 9 * C3_C * 8; i += 512) {
    const int c8 = i & 7, row = i >> 3;              // row = tap * 64 + out channel
    const int tap = row >> 6, co = row & 63;
    const int st = flip ? 8 - tap : tap;
    *reinterpret_cast<uint4*>(wl + row * C3_PP + c8 * 16) = *reinterpret_cast<const uint4*>(w + ((long)co * 9 + st) * C3_C + c8 * 8);
  }
  const int tpc = (H + C3_TR - 1) / C3_TR;
  uint4 pre[C3_PRE];
  auto fetch = [&](int t) {
    const int b = t / tpc, row0 = (t - b * tpc) * C3_TR;
    const bf16_t* xb = x + (long)b * H * C3_W * C3_C;
#pragma unroll
    for (int q = 0; q < C3_PRE; ++q) {
      const int id = tid + 512 * q;
      const int c8 = id & 7, px = id >> 3;
      const int hr = px / C3_HW, hc = px - hr * C3_HW;
      const int r = row0 - 1 + hr, c = hc - 1;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (id < C3_CHUNKS && (unsigned)r < (unsigned)H && (unsigned)c < (unsigned)C3_W)
        v = *reinterpret_cast<const uint4*>(xb + ((long)r * C3_W + c) * C3_C + c8 * 8);
      pre[q] = v;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  float sc[2] = {1.f, 1.f}, bi[2] = {0.f, 0.f};
  if (scale) { sc[0] = scale[n]; sc[1] = scale[32 + n]; }
  if (bias) { bi[0] = bias[n]; bi[1] = bias[32 + n]; }
  // this lane's fragment bases: A = pixel (row 2 wv + (n >> 4), column n & 15) of the tile, B = output channel n (+ 32)
  const unsigned char* abase = halo + ((2 * wv + (n >> 4)) * C3_HW + (n & 15)) * C3_PP + hf * 16;
  const unsigned char* bbase = wl + n * C3_PP + hf * 16;
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tpc, row0 = (t - b * tpc) * C3_TR;
    __syncthreads();                                  // the previous tile's output stage is stored (and the weights are in place)
#pragma unroll
    for (int q = 0; q < C3_PRE; ++q) {
      const int id = tid + 512 * q;
      if (id < C3_CHUNKS) *reinterpret_cast<uint4*>(halo + (id >> 3) * C3_PP + (id & 7) * 16) = pre[q];
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    // The 16 lanes of a DPP row are the 16 pixels of one image row, so the fragments of the left / right taps (kw = 0, 2) are
    // the centre tap's fragment shifted by one lane (row_shr:1 / row_shl:1; the lane shifted in from outside the row reads
    // zero = the convolution's zero padding): one LDS read per (kh, k16 step) instead of three.
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const bf16x8 ac = *reinterpret_cast<const bf16x8*>(abase + (kh * C3_HW + 1) * C3_PP + ks * 32);
        const c3_i4 ci = __builtin_bit_cast(c3_i4, ac);
        c3_i4 li, ri;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          li[q] = __builtin_amdgcn_update_dpp(0, ci[q], 0x111, 0xf, 0xf, true);       // row_shr:1 -> pixel column c - 1
          ri[q] = __builtin_amdgcn_update_dpp(0, ci[q], 0x101, 0xf, 0xf, true);       // row_shl:1 -> pixel column c + 1
        }
        const bf16x8 av[3] = {__builtin_bit_cast(bf16x8, li), ac, __builtin_bit_cast(bf16x8, ri)};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = kh * 3 + kw;
          const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(bbase + (tap * C3_C) * C3_PP + ks * 32);
          const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(bbase + (tap * C3_C + 32) * C3_PP + ks * 32);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kw], b0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[kw], b1, acc1, 0, 0, 0);
        }
      }
    }
    __syncthreads();                                  // every wave is done with the halo: it becomes the output stage
    unsigned char* stage = halo + wv * 32 * C3_PP;    // this wave's 32 pixels x 64 channels
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int px = c3_crow(i, hf);
      float v0 = acc0[i] * sc[0] + bi[0], v1 = acc1[i] * sc[1] + bi[1];
      if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
      *reinterpret_cast<bf16_t*>(stage + px * C3_PP + n * 2) = (bf16_t)v0;
      *reinterpret_cast<bf16_t*>(stage + px * C3_PP + (32 + n) * 2) = (bf16_t)v1;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's stage is written (it alone reads it back)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // 32 pixels x 8 chunks = 256 chunks, 64 lanes
      const int id = lane + 64 * q, px = id >> 3, c8 = id & 7;
      const int r = row0 + 2 * wv + (px >> 4);
      if (r < H) {
        const long o = (((long)b * H + r) * C3_W + (px & 15)) * C3_C + c8 * 8;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(stage + px * C3_PP + c8 * 16);
        if (mask) {
          const bf16x8 m = *reinterpret_cast<const bf16x8*>(mask + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((float)m[e] > 0.f) ? v[e] : (bf16_t)0.f;
        }
        *reinterpret_cast<bf16x8*>(y + o) = v;
      }
    }
  }
}

// (Measured and dropped: the same kernel with all 72 weight fragments in registers - 288 of the unified 512 - and 4 waves per
//  workgroup: no LDS traffic for B at all, but one wave per SIMD cannot hide the A-fragment and epilogue latencies: 24.0 us
//  forward / 29.6 us dgrad against 17.7 / 21.5 us for this version.)

}  // namespace sedt

extern "C" int sedt_conv3x3_c64(const void* x, const void* w, int flip, const float* scale, const float* bias, int relu,
                                const void* mask, void* y, int B, int H, int W, int C, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(x && w && y, "conv3x3_c64: null pointer");
  SEDT_REQUIRE(W == C3_W && C == C3_C && H >= 1 && B >= 1, "conv3x3_c64: built for 64 channels on a 16-wide map (got C=%d W=%d)", C, W);
  SEDT_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) |
                 reinterpret_cast<uintptr_t>(mask)) & 15) == 0, "conv3x3_c64: operands must be 16-byte aligned");
  const long nt = (long)B * ((H + C3_TR - 1) / C3_TR);
  SEDT_REQUIRE(nt < (1L << 30), "conv3x3_c64: too many tiles");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       C3_WL + C3_HALO);
    if (e != hipSuccess) {
      set_error("conv3x3_c64: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 1;
    }
    attr = true;
  }
  static int gmax = -1;
  if (gmax < 0) {
    const char* e = sedt::dev_getenv("SEDT_C3_GRID");
    gmax = std::max(e ? atoi(e) : 256, 1);          // (a tuning override; never a zero-size grid)
  }
  const int grid = (int)std::min<long>(nt, gmax);
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(512), C3_WL + C3_HALO, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(w), flip, scale, bias, relu,
                     reinterpret_cast<const bf16_t*>(mask), reinterpret_cast<bf16_t*>(y), H, (int)nt);
  return check_launch("conv3x3_c64");
}
