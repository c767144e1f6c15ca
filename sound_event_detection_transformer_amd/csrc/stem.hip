// stem.hip - the ResNet stem of the SEDT backbone as two direct kernels (bf16 mode, 64 mel bands):
//
//   forward   conv0 (1 -> 3, 1x1, bias) o conv1 (7x7 stride 2 pad 3) o FrozenBN o ReLU o max-pool 3x3 stride 2 pad 1
//             (reference sedt/backbone.py:98-111 + torchvision's stem) from the f32 spectrogram straight to the pooled
//             activation + argmax bytes: the im2col matrix (131 MB at B = 64) and the un-pooled activation (65 MB) of the
//             im2col -> GEMM -> pool chain never exist.
//   backward  the weight gradient of the folded 7x7 convolution (all conv0 needs: conv1 is frozen) from the POOLED
//             gradient: the max-pool backward (argmax routing + ReLU mask from the pooled output) is evaluated per element
//             while the MFMA A-fragments are built, the patch matrix is rebuilt from the staged input rows.
//
// conv0 is folded into conv1 exactly as stem_prep_kernel (misc.hip) lays it out: wcat[co][tap] for the 49 taps of x and
// wcat[co][64 + tap] for 49 "tap in bounds" indicator columns (conv1 zero-pads conv0's OUTPUT, so conv0's bias only reaches
// in-bounds taps).  v_mfma_f32_32x32x16_bf16; one output row of 32 pixels per wave.
#include <stdlib.h>
#include <algorithm>
#include "common.h"

namespace sedt {

constexpr int XP = 72;        // pitch of a staged input row in floats: 3 zero columns, 64 bands, 3 zero columns, 2 spare
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
constexpr int S1P = 144;      // bytes per pixel of the un-pooled LDS tile (64 channels bf16 + 16 B: conflict-free 16-B reads)

__device__ __forceinline__ int stem_crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ bf16x8 stem_pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
  return o;
}

// ---------------------------------------------------------------------------------------------------------------- forward
// Workgroup = 5 waves = the 5 un-pooled rows (2 hp0 - 1 ... 2 hp0 + 3) under the pooled rows hp0, hp0 + 1 of one clip; the row
// shared with the next tile is recomputed (compute is free here, HBM writes are what the kernel is made of).
// K order of the MFMA (any order works as long as A and B agree): k-step j = 0..3, lane half h, element e <-> tap
// (kh = 2j + h, kw = e): the A fragment of a lane is then 8 CONSECUTIVE floats of one staged input row, and the slots with
// kh = 7 or kw = 7 carry zero weights.  k-steps 4..7: the indicator columns in the same order.
__global__ __launch_bounds__(320) void stem_pool_fwd_kernel(const float* __restrict__ x, const bf16_t* __restrict__ wcat,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ pool, uint8_t* __restrict__ idx,
                                                            bf16_t* __restrict__ s1_out, int H, int Ho, int Hp, int ntiles, int dbg) {
  __shared__ float xin[16 * XP];
  __shared__ __attribute__((aligned(16))) unsigned char s1t[5 * 32 * S1P];
  __shared__ bf16x8 wfr[2 * 4 * 2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, n = lane & 31;
  // the folded weights once per workgroup: coalesced into LDS (the un-pooled tile's space, pitch 130 elements against bank
  // conflicts), then re-laid as ready-made B fragments wfr[part][j][h][co] (16 B each, 16 KB): a fragment is one ds_read_b128
  // in the loop instead of 64 live registers (128 scattered 2-byte global loads per lane were the whole first version)
  {
    bf16_t* wl = reinterpret_cast<bf16_t*>(s1t);
    for (int i = tid; i < 64 * 16; i += 320) {
      const int co = i >> 4, c8 = i & 15;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(wcat + co * 128 + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) wl[co * 130 + c8 * 8 + e] = v[e];
    }
    __syncthreads();
    for (int f = tid; f < 2 * 4 * 2 * 64; f += 320) {
      const int co = f & 63, hh = (f >> 6) & 1, j = (f >> 7) & 3, part = f >> 9;
      const int kh = 2 * j + hh;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (kh < 7 && e < 7) ? wl[co * 130 + part * 64 + kh * 7 + e] : (bf16_t)0.f;
      wfr[f] = v;
    }
  }
  bf16x8 cfrag, zfrag;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    cfrag[e] = (bf16_t)(((unsigned)(2 * n - 3 + e) < 64u) ? 1.f : 0.f);
    zfrag[e] = (bf16_t)0.f;
  }
  const float sc0 = scale[n], sc1 = scale[32 + n], bi0 = bias[n], bi1 = bias[32 + n];
  const int tpc = (Hp + 1) / 2;
  // the 16 input rows of a tile = 1152 floats = up to 4 per thread, fetched into registers one tile ahead (the loads of tile
  // t + grid fly while tile t is multiplied and pooled: with a plain load -> barrier -> compute loop the kernel sat on the latency)
  float pre[4];
  auto fetch = [&](int t) {
    const int b = t / tpc, hp0 = 2 * (t - b * tpc);
    const int in0 = 2 * (2 * hp0 - 1) - 3;
    const float* xb = x + (long)b * H * 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = tid + 320 * q;
      const int rr = i / XP, j = i - rr * XP;
      const int hi = in0 + rr, wi = j - 3;
      pre[q] = (i < 16 * XP && (unsigned)hi < (unsigned)H && (unsigned)wi < 64u) ? xb[hi * 64 + wi] : 0.f;
    }
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  __syncthreads();                                     // fragment image complete; the weight image in s1t is dead
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int b = t / tpc, hp0 = 2 * (t - b * tpc);
    const int r0 = 2 * hp0 - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (tid + 320 * q < 16 * XP) xin[tid + 320 * q] = pre[q];
    __syncthreads();                                   // xin complete; the previous tile's pooling pass is done with s1t
    if (t + (int)gridDim.x < ntiles && !(dbg & 4)) fetch(t + gridDim.x);
    const int r = r0 + w;
    if ((unsigned)r < (unsigned)Ho && !(dbg & 1)) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float2* src = reinterpret_cast<const float2*>(xin + (2 * w + 2 * j + h) * XP + 2 * n);
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float2 f = src[q];
          v[2 * q] = f.x;
          v[2 * q + 1] = f.y;
        }
        const bf16x8 a = stem_pack8(v);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfr[(j * 2 + h) * 64 + n], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfr[(j * 2 + h) * 64 + 32 + n], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool rv = (unsigned)(2 * r - 3 + 2 * j + h) < (unsigned)H;
        const bf16x8 a = rv ? cfrag : zfrag;
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfr[((4 + j) * 2 + h) * 64 + n], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wfr[((4 + j) * 2 + h) * 64 + 32 + n], acc1, 0, 0, 0);
      }
      unsigned char* row = s1t + (w * 32) * S1P;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int px = stem_crow(i, h);
        // ReLU as an integer maximum of the bit pattern: negative floats AND -0 become +0 (the pooling pass orders the
        // activation as unsigned 16-bit integers)
        const int v0 = max(__builtin_bit_cast(int, acc0[i] * sc0 + bi0), 0), v1 = max(__builtin_bit_cast(int, acc1[i] * sc1 + bi1), 0);
        *reinterpret_cast<bf16_t*>(row + px * S1P + n * 2) = (bf16_t)__builtin_bit_cast(float, v0);
        *reinterpret_cast<bf16_t*>(row + px * S1P + (32 + n) * 2) = (bf16_t)__builtin_bit_cast(float, v1);
      }
    }
    __syncthreads();
    if (s1_out && (w < 4 || t - b * tpc == tpc - 1) && (unsigned)r < (unsigned)Ho) {   // optional un-pooled activation (tests)
      bf16_t* dst = s1_out + ((long)b * Ho + r) * 32 * 64;
      for (int i = lane; i < 32 * 8; i += 64) {
        const int px = i >> 3, c8 = i & 7;
        *reinterpret_cast<uint4*>(dst + px * 64 + c8 * 8) = *reinterpret_cast<const uint4*>(s1t + (w * 32 + px) * S1P + c8 * 16);
      }
    }
    if (tid < 256 && !(dbg & 2)) {
      const int ph = tid >> 7, wp = (tid >> 3) & 15, c8 = tid & 7;
      const int hp = hp0 + ph;
      if (hp < Hp) {
        // The activation is >= +0 (the epilogue clamps through the integer maximum, so no -0), hence bf16 order = unsigned
        // 16-bit order: packed v_pk_max_u16 over the window, two channels per instruction.  Argmax (training only): per tap
        // cand = tap | (value != max) << 4, running packed minimum -> the FIRST maximal tap in (kh, kw) scan order, as torch
        // and maxpool_fwd_kernel choose.  (The float compare/select form was 330 VALU instructions per thread: the kernel's bound.)
        u16x2 m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = (u16x2)(0);
        uint4 tapv[9];
        bool tapok[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int rr = 2 * hp - 1 + kh;
          const int lr = 2 * ph + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int px = 2 * wp - 1 + kw;
            const bool ok = (unsigned)rr < (unsigned)Ho && (unsigned)px < 32u;
            tapok[kh * 3 + kw] = ok;
            uint4 xv = make_uint4(0, 0, 0, 0);
            if (ok) xv = *reinterpret_cast<const uint4*>(s1t + (lr * 32 + px) * S1P + c8 * 16);
            tapv[kh * 3 + kw] = xv;
            m[0] = __builtin_elementwise_max(m[0], __builtin_bit_cast(u16x2, xv.x));
            m[1] = __builtin_elementwise_max(m[1], __builtin_bit_cast(u16x2, xv.y));
            m[2] = __builtin_elementwise_max(m[2], __builtin_bit_cast(u16x2, xv.z));
            m[3] = __builtin_elementwise_max(m[3], __builtin_bit_cast(u16x2, xv.w));
          }
        }
        const long o = (((long)b * Hp + hp) * 16 + wp) * 64 + c8 * 8;
        *reinterpret_cast<uint4*>(pool + o) = make_uint4(__builtin_bit_cast(uint32_t, m[0]), __builtin_bit_cast(uint32_t, m[1]),
                                                         __builtin_bit_cast(uint32_t, m[2]), __builtin_bit_cast(uint32_t, m[3]));
        if (idx) {
          u16x2 best[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) best[q] = (u16x2)(31);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            if (!tapok[tap]) continue;
            const uint32_t xs[4] = {tapv[tap].x, tapv[tap].y, tapv[tap].z, tapv[tap].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const u16x2 d = m[q] - __builtin_bit_cast(u16x2, xs[q]);
              const u16x2 cand = (__builtin_elementwise_min(d, (u16x2)(1)) << (u16x2)(4)) | (u16x2)(tap);
              best[q] = __builtin_elementwise_min(best[q], cand);
            }
          }
          uint2 io;
          io.x = (uint32_t)best[0].x | ((uint32_t)best[0].y << 8) | ((uint32_t)best[1].x << 16) | ((uint32_t)best[1].y << 24);
          io.y = (uint32_t)best[2].x | ((uint32_t)best[2].y << 8) | ((uint32_t)best[3].x << 16) | ((uint32_t)best[3].y << 24);
          *reinterpret_cast<uint2*>(idx + o) = io;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------------------------------- backward
// G[co][k] = sum over pixels of gs[pixel][co] * patch[pixel][k]: M = co (2 tiles), N = the 128 columns of wcat's layout
// (4 tiles; lanes <-> columns, so the slab is written in that layout directly), K = the 32 pixels of an un-pooled row (2
// steps).  gs = max-pool backward of the pooled gradient through the stem ReLU, rounded to bf16 as the unfused chain does.
// Workgroup = 8 waves = 4 consecutive un-pooled rows x 2 co tiles (each wave: one row, 32 output channels, all 128 columns:
// 64 accumulator registers - the first version held 128 and ran one wave per SIMD, latency-bound); persistent over tiles;
// one f32 slab [64][128] per workgroup.
__global__ __launch_bounds__(512, 4) void stem_pool_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ g,
                                                              const uint8_t* __restrict__ idx, const bf16_t* __restrict__ pool,
                                                              float* __restrict__ slab, int H, int Ho, int Hp, int ntiles) {
  __shared__ float xin[13 * XP];
  __shared__ __attribute__((aligned(16))) bf16_t gm[3 * 16 * 64];
  __shared__ __attribute__((aligned(16))) uint8_t ix[3 * 16 * 64];
  __shared__ __attribute__((aligned(16))) float red[64 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, h = lane >> 5, n = lane & 31;
  const int w = wid & 3, mt = wid >> 2;
  // per-lane column constants.  Columns 49..63 / 113..127 of the layout are zero columns: their lanes multiply finite
  // garbage (tap 0) and are zeroed when the slab is written.
  int ckh[2], ckw[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int tap = 32 * nt + n;
    ckh[nt] = tap < 49 ? tap / 7 : 0;
    ckw[nt] = tap < 49 ? tap - ckh[nt] * 7 : 0;
  }
  // "tap in bounds along the mel axis" indicator fragments: depend on the lane's kw and the pixel only, not on the row
  bf16x8 cfr[2][2], zfrag;
#pragma unroll
  for (int e = 0; e < 8; ++e) zfrag[e] = (bf16_t)0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
      for (int e = 0; e < 8; ++e) cfr[nt][ps][e] = (bf16_t)(((unsigned)(2 * (16 * ps + 8 * h + e) - 3 + ckw[nt]) < 64u) ? 1.f : 0.f);
  f32x16 acc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
  const int tpc = (Ho + 3) / 4;
  // one tile ahead in registers (see the forward kernel): 13 input rows = 936 floats (2 per thread) and 3 pooled rows of
  // gradient / output / argmax = 384 eight-channel items (threads 0..383)
  float pre[2];
  bf16x8 pg;
  uint2 pi;
  auto fetch = [&](int t) {
    const int b = t / tpc, r0 = 4 * (t - b * tpc);
    const int hpb = r0 / 2, in0 = 2 * r0 - 3;
    const float* xb = x + (long)b * H * 64;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int i = tid + 512 * q;
      const int rr = i / XP, j = i - rr * XP;
      const int hi = in0 + rr, wi = j - 3;
      pre[q] = (i < 13 * XP && (unsigned)hi < (unsigned)H && (unsigned)wi < 64u) ? xb[hi * 64 + wi] : 0.f;
    }
    const int lhp = tid >> 7, rest = tid & 127;
    const int hp = hpb + lhp;
    bf16x8 gv;
#pragma unroll
    for (int e = 0; e < 8; ++e) gv[e] = (bf16_t)0.f;
    uint2 iv = make_uint2(0xffffffffu, 0xffffffffu);
    if (tid < 384 && hp < Hp) {
      const long o = ((long)b * Hp + hp) * 16 * 64 + rest * 8;
      const bf16x8 gg = *reinterpret_cast<const bf16x8*>(g + o);
      const bf16x8 pv = *reinterpret_cast<const bf16x8*>(pool + o);
      iv = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = ((float)pv[e] > 0.f) ? gg[e] : (bf16_t)0.f;
    }
    pg = gv;
    pi = iv;
  };
  if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int r0 = 4 * (t - (t / tpc) * tpc);
    const int hpb = r0 / 2;
    __syncthreads();                                   // the previous tile's fragments are built: xin / gm / ix are free
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (tid + 512 * q < 13 * XP) xin[tid + 512 * q] = pre[q];
    if (tid < 384) {
      *reinterpret_cast<bf16x8*>(gm + tid * 8) = pg;
      *reinterpret_cast<uint2*>(ix + tid * 8) = pi;
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    const int r = r0 + w;
    if (r < Ho) {
      // gs[pixel][co] = max-pool backward at un-pooled pixel (r, wo): the pooled windows that can have selected it are
      // hp = (r+1)/2 - a (a = 0, 1), wp = (wo+1)/2 - d (d = 0, 1) - summed a-major, d-minor like maxpool_bwd_kernel.  Branch-free:
      // the 5 pooled columns under 8 consecutive pixels are read once per row candidate, then selected by argmax code.
      bf16x8 A[2];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int co = 32 * mt + n, wb = 8 * ps + 4 * h;             // first pooled column under pixels [16 ps + 8 h, +8)
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int hp = (r + 1) / 2 - a;
          const int kh = r - (2 * hp - 1);
          if (hp < 0 || hp >= Hp || kh < 0 || kh > 2) continue;     // wave-uniform
          const int base = (hp - hpb) * 1024 + co;
          float gv[5];
          int cv[5];
#pragma unroll
          for (int c = 0; c < 5; ++c) {
            const int wp = wb + c;
            const int o = base + (wp < 16 ? wp : 15) * 64;
            cv[c] = wp < 16 ? (int)ix[o] - kh * 3 : -1;                 // argmax code relative to this kernel row: kw if selected here
            gv[c] = (float)gm[o];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if ((e & 1) == 0) {
              s[e] += cv[e >> 1] == 1 ? gv[e >> 1] : 0.f;               // wo even: wp = wo / 2, kw = 1
            } else {
              s[e] += cv[(e + 1) >> 1] == 0 ? gv[(e + 1) >> 1] : 0.f;   // d = 0: wp = (wo + 1) / 2, kw = 0
              s[e] += cv[(e - 1) >> 1] == 2 ? gv[(e - 1) >> 1] : 0.f;   // d = 1: wp = (wo - 1) / 2, kw = 2
            }
          }
        }
        A[ps] = stem_pack8(s);
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float v[8];
          const float* src = xin + (2 * w + ckh[nt]) * XP + 2 * (16 * ps + 8 * h) + ckw[nt];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = src[2 * e];
          acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ps], stem_pack8(v), acc[nt], 0, 0, 0);
          const bool rv = (unsigned)(2 * r - 3 + ckh[nt]) < (unsigned)H;
          acc[2 + nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ps], rv ? cfr[nt][ps] : zfrag, acc[2 + nt], 0, 0, 0);
        }
      }
    }
  }
  // the four row-waves' partial sums of each co tile, added in wave order (fixed order: bit-reproducible); zero columns zeroed
  for (int ww = 0; ww < 4; ++ww) {
    __syncthreads();
    if (w == ww) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bool live = ((32 * nt + n) & 63) < 49;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* p = red + (32 * mt + stem_crow(i, h)) * 128 + 32 * nt + n;
          const float v = (ww == 0 ? 0.f : *p) + acc[nt][i];
          *p = live ? v : 0.f;
        }
      }
    }
  }
  __syncthreads();
  float4* dst = reinterpret_cast<float4*>(slab + (long)blockIdx.x * 64 * 128);
  for (int i = tid; i < 64 * 128 / 4; i += 512) dst[i] = reinterpret_cast<const float4*>(red)[i];
}

}  // namespace sedt

extern "C" int sedt_stem_pool_fwd(const float* x, const void* wcat, const float* scale, const float* bias, void* pool, uint8_t* idx,
                                  void* s1_out, int B, int H, int W, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(x && wcat && scale && bias && pool, "stem_pool_fwd: null pointer");
  SEDT_REQUIRE(W == 64 && H >= 1 && B >= 1, "stem_pool_fwd: needs 64 mel bands (got W=%d, H=%d, B=%d)", W, H, B);
  const int Ho = (H - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1;
  const long nt = (long)B * ((Hp + 1) / 2);
  SEDT_REQUIRE(nt < (1L << 30) && (long)B * H * 64 < (1L << 40), "stem_pool_fwd: too many tiles");
  static int gmax = -1, dbg = 0;
  if (gmax < 0) {
    const char* e = sedt::dev_getenv("SEDT_STEM_GRID");
    gmax = std::max(e ? atoi(e) : 512, 1);        // measured: 512 persistent workgroups 37.6 us, 768 40.2, 1024 44.0 (C2 shape)
#ifdef SEDT_DEV                      // ablation switch of tools/dev/time_stem.py (skips the MFMAs / the prefetch: WRONG results);
    e = sedt::dev_getenv("SEDT_STEM_DBG");     // compiled only into developer builds (hipcc -DSEDT_DEV), never into the product library
    dbg = e ? atoi(e) : 0;
#endif
  }
  const int grid = (int)std::min<long>(nt, gmax);
  hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3(grid), dim3(320), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<const bf16_t*>(wcat), scale, bias, reinterpret_cast<bf16_t*>(pool), idx,
                     reinterpret_cast<bf16_t*>(s1_out), H, Ho, Hp, (int)nt, dbg);
  return check_launch("stem_pool_fwd");
}

extern "C" int sedt_stem_pool_wgrad_slabs(int B, int H) {
  const int Ho = (H - 1) / 2 + 1;
  const long nt = (long)B * ((Ho + 3) / 4);
  return (int)std::min<long>(nt, 512);
}

extern "C" int sedt_stem_pool_wgrad(const float* x, const void* g, const uint8_t* idx, const void* pool, float* slab, int nslab,
                                    int B, int H, int W, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(x && g && idx && pool && slab, "stem_pool_wgrad: null pointer");
  SEDT_REQUIRE(W == 64 && H >= 1 && B >= 1, "stem_pool_wgrad: needs 64 mel bands (got W=%d, H=%d, B=%d)", W, H, B);
  SEDT_REQUIRE(nslab == sedt_stem_pool_wgrad_slabs(B, H), "stem_pool_wgrad: slab count %d != sedt_stem_pool_wgrad_slabs() = %d", nslab,
               sedt_stem_pool_wgrad_slabs(B, H));
  const int Ho = (H - 1) / 2 + 1, Hp = (Ho - 1) / 2 + 1;
  const long nt = (long)B * ((Ho + 3) / 4);
  hipLaunchKernelGGL(stem_pool_wgrad_kernel, dim3(nslab), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), x,
                     reinterpret_cast<const bf16_t*>(g), idx, reinterpret_cast<const bf16_t*>(pool), slab, H, Ho, Hp, (int)nt);
  return check_launch("stem_pool_wgrad");
}
