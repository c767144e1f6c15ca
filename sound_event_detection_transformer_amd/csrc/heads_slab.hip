// heads_slab.hip - the prediction heads on the stacked decoder output (reference sedt/sedt.py:88-95, 398-409) in ONE launch each way.
//
// hs [L*B*Qp][256]: class_embed (256 -> C1) on every row, bbox_embed (256 -> 256 -> 256 -> 2, ReLU, sigmoid) on every row and
// weak_class_embed (256 -> CA, sigmoid) on query 0 of the last layer.  The per-op path is 5 launches forward (3 skinny kernels + 2
// GEMMs, ~49 us at C2) and 8 + 2 backward (~80 us) for 0.6 GFLOP: pure launch latency.  Here a workgroup owns a slab of 32 rows
// (csrc/slab.h): the two 256 x 256 layers of the box MLP stream their fragment-major weights, the three narrow heads read their f32
// MASTER weights from LDS (as skinny.hip does), and the backward produces the input gradient, the two hidden-layer gradients the
// weight-gradient GEMMs read, and per-slab partial sums of the narrow heads' weight / bias gradients (summed by the layer's reduce
// launch in a fixed order).  Rounding points are those of the per-op chain (h1, h2, gradients in bf16; head outputs f32).
// Envelope: bf16, d = 256, C1, CA <= 16.
#include "slab.h"

namespace sedt {

using slab::u32x4;
using slab::XP;

constexpr int HS_D = 256, HS_WP = 260;       // f32 weight rows in LDS: 1040-byte pitch
constexpr int HS_MAXN = 16;

struct HeadsFwdArgs {
  const bf16_t* x;                                     // [rows][256]
  const float* wc; const float* bc;                    // class head [C1][256]
  const u32x4* w1; const float* b1; const u32x4* w2; const float* b2;      // box MLP layers 0, 1 (fragment-major)
  const float* w3; const float* b3;                    // box MLP layer 2 [2][256]
  const float* wa; const float* ba;                    // audio-tag head [CA][256] or null
  float* cls; float* box; float* at;                   // [rows][C1], [rows][2], [B][CA]
  bf16_t* h1; bf16_t* h2;                              // [rows][256] (training) or null
  int rows, L, B, Qp, C1, CA;
};

// the row of `at` a hs row feeds (query 0 of the last layer), or -1
__device__ __forceinline__ int at_row(int row, int L, int B, int Qp) {
  const int q = row % Qp, lb = row / Qp;
  return (q == 0 && lb / B == L - 1) ? lb % B : -1;
}

__global__ __launch_bounds__(512) void heads_fwd_kernel(const HeadsFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* X = reinterpret_cast<bf16_t*>(smem);                      // [32][XP]
  bf16_t* H1 = X + 32 * XP;
  bf16_t* H2 = H1 + 32 * XP;
  float* WS = reinterpret_cast<float*>(H2 + 32 * XP);               // [(C1 + CA + 2)][HS_WP]: wc | wa | w3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
  const long row0 = (long)blockIdx.x * 32;
  const int nvalid = min(32, a.rows - (int)row0);
  const long ts256 = 64L * 16;
  slab::u32x4 wa_[8], wb_[8];
  slab::load_chunk<1>(wa_, a.w1 + (long)wave * ts256, 0, 0, lane);
  slab::load_chunk<1>(wb_, a.w1 + (long)wave * ts256, 0, 8, lane);
  slab::issue_fence();
  {
    uint4 xr[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      xr[q] = r < nvalid ? *reinterpret_cast<const uint4*>(a.x + (row0 + r) * HS_D + c) : make_uint4(0, 0, 0, 0);
    }
    const int NW = a.C1 + a.CA + 2;
    for (int i = tid; i < NW * 64; i += 512) {                      // 64 float4 per weight row
      const int wn = i >> 6, c4 = (i & 63) * 4;
      const float* src = wn < a.C1 ? a.wc + wn * HS_D : wn < a.C1 + a.CA ? a.wa + (wn - a.C1) * HS_D : a.w3 + (wn - a.C1 - a.CA) * HS_D;
      *reinterpret_cast<float4*>(WS + wn * HS_WP + c4) = *reinterpret_cast<const float4*>(src + c4);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      *reinterpret_cast<uint4*>(X + r * XP + c) = xr[q];
    }
  }
  __syncthreads();
  // dot products of a slab row with an f32 weight row in LDS
  auto dot = [&](const bf16_t* tile, int r, const float* w) {
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < HS_D; k += 8) {
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(tile + r * XP + k);
      const float4 w0 = *reinterpret_cast<const float4*>(w + k), w1 = *reinterpret_cast<const float4*>(w + k + 4);
      s += (float)xv[0] * w0.x + (float)xv[1] * w0.y + (float)xv[2] * w0.z + (float)xv[3] * w0.w + (float)xv[4] * w1.x + (float)xv[5] * w1.y +
           (float)xv[6] * w1.z + (float)xv[7] * w1.w;
    }
    return s;
  };
  // ---- box MLP layer 0 (MFMA) ...
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    float4 bb[4];
    slab::load_feat4(bb, a.b1, wave, hf);
    slab::issue_fence();
    const slab::u32x4* w2p = a.w2 + (long)wave * ts256;
    slab::wave_gemm_small(acc, X, XP, lane, wa_, wb_, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w2p, 0, 0, lane); },
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w2p, 0, 8, lane); });
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      const float bv[4] = {bb[g4].x, bb[g4].y, bb[g4].z, bb[g4].w};
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(n < nvalid ? fmaxf(acc[0][4 * g4 + e] + bv[e], 0.f) : 0.f);
      *reinterpret_cast<VecT<bf16_t, 4>*>(H1 + n * XP + f) = o;
      if (a.h1 && n < nvalid) *reinterpret_cast<VecT<bf16_t, 4>*>(a.h1 + (row0 + n) * HS_D + f) = o;
    }
  }
  // ---- ... while the narrow heads on x are plain dot products: class logits, audio tags
  for (int i = tid; i < 32 * (a.C1 + a.CA); i += 512) {
    const int r = i & 31, wn = i >> 5;
    if (r >= nvalid) continue;
    if (wn < a.C1) {
      a.cls[(row0 + r) * a.C1 + wn] = dot(X, r, WS + wn * HS_WP) + a.bc[wn];
    } else {
      const int ar = at_row((int)(row0 + r), a.L, a.B, a.Qp);
      if (ar >= 0) {
        const int c = wn - a.C1;
        a.at[(long)ar * a.CA + c] = 1.f / (1.f + __expf(-(dot(X, r, WS + wn * HS_WP) + a.ba[c])));
      }
    }
  }
  __syncthreads();
  // ---- box MLP layer 1
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    float4 bb[4];
    slab::load_feat4(bb, a.b2, wave, hf);
    slab::issue_fence();
    slab::wave_gemm_small(acc, H1, XP, lane, wa_, wb_, [&](slab::u32x4(&)[8]) {}, [&](slab::u32x4(&)[8]) {});
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      const float bv[4] = {bb[g4].x, bb[g4].y, bb[g4].z, bb[g4].w};
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(n < nvalid ? fmaxf(acc[0][4 * g4 + e] + bv[e], 0.f) : 0.f);
      *reinterpret_cast<VecT<bf16_t, 4>*>(H2 + n * XP + f) = o;
      if (a.h2 && n < nvalid) *reinterpret_cast<VecT<bf16_t, 4>*>(a.h2 + (row0 + n) * HS_D + f) = o;
    }
  }
  __syncthreads();
  // ---- box MLP layer 2 + sigmoid
  if (tid < 64) {
    const int r = tid & 31, j = tid >> 5;
    if (r < nvalid) a.box[(row0 + r) * 2 + j] = 1.f / (1.f + __expf(-(dot(H2, r, WS + (a.C1 + a.CA + j) * HS_WP) + a.b3[j])));
  }
}

struct HeadsBwdArgs {
  const bf16_t* x; const bf16_t* h1; const bf16_t* h2;  // [rows][256]
  const float* box; const float* at;                   // saved outputs [rows][2], [B][CA]
  const float* g_cls; const float* g_box; const float* g_at;      // [rows][C1], [rows][2], [B][CA] (g_at may be null)
  const float* wc; const float* w3; const float* wa;   // f32 masters
  const u32x4* w2t; const u32x4* w1t;                  // fragment-major W2^T, W1^T
  bf16_t* dhs; bf16_t* g_h1; bf16_t* g_h2;             // [rows][256]
  float* part;                                         // [slabs][NG * 257], NG = C1 + CA + 2: [NG][256] weight sums, then [NG] bias sums
  int rows, L, B, Qp, C1, CA;
};

__global__ __launch_bounds__(512) void heads_bwd_kernel(const HeadsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* X = reinterpret_cast<bf16_t*>(smem);
  bf16_t* H1 = X + 32 * XP;
  bf16_t* H2 = H1 + 32 * XP;
  bf16_t* G2 = H2 + 32 * XP;
  bf16_t* G1 = G2 + 32 * XP;
  float* WS = reinterpret_cast<float*>(G1 + 32 * XP);               // [(C1 + CA + 2)][HS_WP]: wc | wa | w3
  float* GS = WS + (HS_MAXN * 2 + 2) * HS_WP;                       // [32][36]: per row g_cls (C1) | g_at' (CA) | g_box' (2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
  const long row0 = (long)blockIdx.x * 32;
  const int nvalid = min(32, a.rows - (int)row0);
  const int C1 = a.C1, CA = a.CA, NG = C1 + CA + 2;
  const long ts256 = 64L * 16;
  slab::u32x4 wa_[8], wb_[8];
  slab::load_chunk<1>(wa_, a.w2t + (long)wave * ts256, 0, 0, lane);
  slab::load_chunk<1>(wb_, a.w2t + (long)wave * ts256, 0, 8, lane);
  slab::issue_fence();
  {
    uint4 xr[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int u = tid + (q % 2) * 512, r = u >> 5, c = (u & 31) * 8;
      const bf16_t* src = q < 2 ? a.x : q < 4 ? a.h1 : a.h2;
      xr[q] = r < nvalid ? *reinterpret_cast<const uint4*>(src + (row0 + r) * HS_D + c) : make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < NG * 64; i += 512) {
      const int wn = i >> 6, c4 = (i & 63) * 4;
      const float* src = wn < C1 ? a.wc + wn * HS_D : wn < C1 + CA ? a.wa + (wn - C1) * HS_D : a.w3 + (wn - C1 - CA) * HS_D;
      *reinterpret_cast<float4*>(WS + wn * HS_WP + c4) = *reinterpret_cast<const float4*>(src + c4);
    }
    // the gradients entering the three narrow heads, activation derivatives folded in
    for (int i = tid; i < 32 * NG; i += 512) {
      const int r = i / NG, c = i - r * NG;
      float g = 0.f;
      if (r < nvalid) {
        const long row = row0 + r;
        if (c < C1) g = a.g_cls[row * C1 + c];
        else if (c < C1 + CA) {
          const int ar = at_row((int)row, a.L, a.B, a.Qp);
          if (ar >= 0 && a.g_at) { const float y = a.at[(long)ar * CA + (c - C1)]; g = a.g_at[(long)ar * CA + (c - C1)] * y * (1.f - y); }
        } else {
          const float y = a.box[row * 2 + (c - C1 - CA)];
          g = a.g_box[row * 2 + (c - C1 - CA)] * y * (1.f - y);
        }
      }
      GS[r * 36 + c] = g;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int u = tid + (q % 2) * 512, r = u >> 5, c = (u & 31) * 8;
      bf16_t* dst = q < 2 ? X : q < 4 ? H1 : H2;
      *reinterpret_cast<uint4*>(dst + r * XP + c) = xr[q];
    }
  }
  __syncthreads();
  // ---- g_h2 = (g_box' w3) [h2 > 0]
  for (int u = tid; u < 32 * 32; u += 512) {
    const int r = u >> 5, c = (u & 31) * 8;
    const float g0 = GS[r * 36 + C1 + CA], g1 = GS[r * 36 + C1 + CA + 1];
    const float* w30 = WS + (C1 + CA) * HS_WP + c;
    const float* w31 = w30 + HS_WP;
    const bf16x8 hv = *reinterpret_cast<const bf16x8*>(H2 + r * XP + c);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)hv[e] > 0.f ? g0 * w30[e] + g1 * w31[e] : 0.f);
    *reinterpret_cast<bf16x8*>(G2 + r * XP + c) = o;
    if (r < nvalid) *reinterpret_cast<bf16x8*>(a.g_h2 + (row0 + r) * HS_D + c) = o;
  }
  __syncthreads();
  // ---- g_h1 = (g_h2 W2) [h1 > 0]
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    const slab::u32x4* w1p = a.w1t + (long)wave * ts256;
    slab::wave_gemm_small(acc, G2, XP, lane, wa_, wb_, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w1p, 0, 0, lane); },
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w1p, 0, 8, lane); });
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      const VecT<bf16_t, 4> hv = *reinterpret_cast<const VecT<bf16_t, 4>*>(H1 + n * XP + f);
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((float)hv.v[e] > 0.f ? acc[0][4 * g4 + e] : 0.f);
      *reinterpret_cast<VecT<bf16_t, 4>*>(G1 + n * XP + f) = o;
      if (n < nvalid) *reinterpret_cast<VecT<bf16_t, 4>*>(a.g_h1 + (row0 + n) * HS_D + f) = o;
    }
  }
  __syncthreads();
  // ---- dhs = g_h1 W1 + g_cls wc + g_at' wa
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    slab::wave_gemm_small(acc, G1, XP, lane, wa_, wb_, [&](slab::u32x4(&)[8]) {}, [&](slab::u32x4(&)[8]) {});
    if (n < nvalid) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int f = wave * 32 + 8 * g4 + 4 * hf;
        float v4[4] = {acc[0][4 * g4 + 0], acc[0][4 * g4 + 1], acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]};
        for (int c = 0; c < C1 + CA; ++c) {
          const float g = GS[n * 36 + c];
          const float4 w = *reinterpret_cast<const float4*>(WS + c * HS_WP + f);
          v4[0] += g * w.x; v4[1] += g * w.y; v4[2] += g * w.z; v4[3] += g * w.w;
        }
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)v4[e];
        *reinterpret_cast<VecT<bf16_t, 4>*>(a.dhs + (row0 + n) * HS_D + f) = o;
      }
    }
  }
  // ---- per-slab partial sums of the narrow heads' weight / bias gradients: head row c of GS against X (class, audio tag) or H2 (box)
  float* part = a.part + (long)blockIdx.x * NG * 257;
  for (int i = tid; i < NG * 257; i += 512) {                      // [NG][256] weight sums, then [NG] bias sums
    const bool isw = i < NG * 256;
    const int c = isw ? i >> 8 : i - NG * 256, k = isw ? i & 255 : 256;
    const bf16_t* tile = c < C1 + CA ? X : H2;
    float s = 0.f;
    if (k < 256) {
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s += GS[r * 36 + c] * (float)tile[r * XP + k];
    } else {
#pragma unroll 8
      for (int r = 0; r < 32; ++r) s += GS[r * 36 + c];
    }
    part[i] = s;
  }
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_heads_slab_ok(int D, int C1, int CA, int dtype) {
  return dtype == SEDT_BF16 && D == HS_D && C1 >= 1 && C1 <= HS_MAXN && CA >= 0 && CA <= HS_MAXN;
}

static size_t heads_lds(bool bwd) {
  return (size_t)(bwd ? 5 : 3) * 32 * XP * sizeof(bf16_t) + (size_t)(HS_MAXN * 2 + 2) * HS_WP * sizeof(float) + (bwd ? 32 * 36 * sizeof(float) : 0);
}

extern "C" int sedt_heads_fwd(const void* x, const float* wc, const float* bc, const void* w1_frag, const float* b1, const void* w2_frag,
                              const float* b2, const float* w3, const float* b3, const float* wa, const float* ba, float* cls, float* box,
                              float* at, void* h1, void* h2, int L, int B, int Qp, int C1, int CA, void* stream) {
  SEDT_REQUIRE(x && wc && bc && w1_frag && b1 && w2_frag && b2 && w3 && b3 && cls && box, "heads_fwd: null pointer");
  SEDT_REQUIRE(sedt_heads_slab_ok(HS_D, C1, CA, SEDT_BF16) && L >= 1 && B >= 1 && Qp >= 1, "heads_fwd: C1 = %d / CA = %d outside the envelope", C1, CA);
  SEDT_REQUIRE(CA == 0 || (wa && ba && at), "heads_fwd: the audio-tag head needs wa, ba, at");
  SEDT_REQUIRE((h1 == nullptr) == (h2 == nullptr), "heads_fwd: h1 / h2 come both or not at all");
  HeadsFwdArgs a{(const bf16_t*)x, wc, bc, (const u32x4*)w1_frag, b1, (const u32x4*)w2_frag, b2, w3, b3, wa, ba, cls, box, at,
                 (bf16_t*)h1, (bf16_t*)h2, L * B * Qp, L, B, Qp, C1, CA};
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)heads_lds(false));
    if (e != hipSuccess) { set_error("heads_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 1; }
    attr = true;
  }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((a.rows + 31) / 32), dim3(512), heads_lds(false), reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("heads_fwd");
}

extern "C" size_t sedt_heads_bwd_part_floats(int L, int B, int Qp, int C1, int CA) {
  return (size_t)((L * B * Qp + 31) / 32) * (C1 + CA + 2) * 257;
}

extern "C" int sedt_heads_bwd(const void* x, const void* h1, const void* h2, const float* box, const float* at, const float* g_cls,
                              const float* g_box, const float* g_at, const float* wc, const float* w3, const float* wa, const void* w2t_frag,
                              const void* w1t_frag, void* dhs, void* g_h1, void* g_h2, float* part, int L, int B, int Qp, int C1, int CA,
                              void* stream) {
  SEDT_REQUIRE(x && h1 && h2 && box && g_cls && g_box && wc && w3 && w2t_frag && w1t_frag && dhs && g_h1 && g_h2 && part, "heads_bwd: null pointer");
  SEDT_REQUIRE(sedt_heads_slab_ok(HS_D, C1, CA, SEDT_BF16) && L >= 1 && B >= 1 && Qp >= 1, "heads_bwd: C1 = %d / CA = %d outside the envelope", C1, CA);
  SEDT_REQUIRE(CA == 0 || (wa && at), "heads_bwd: the audio-tag head needs wa, at");
  HeadsBwdArgs a{(const bf16_t*)x, (const bf16_t*)h1, (const bf16_t*)h2, box, at, g_cls, g_box, g_at, wc, w3, wa, (const u32x4*)w2t_frag,
                 (const u32x4*)w1t_frag, (bf16_t*)dhs, (bf16_t*)g_h1, (bf16_t*)g_h2, part, L * B * Qp, L, B, Qp, C1, CA};
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(heads_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)heads_lds(true));
    if (e != hipSuccess) { set_error("heads_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 1; }
    attr = true;
  }
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((a.rows + 31) / 32), dim3(512), heads_lds(true), reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("heads_bwd");
}
