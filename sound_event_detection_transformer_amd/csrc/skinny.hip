// skinny.hip - linear layers with a handful of output features (the SEDT heads: class 256->11, box 256->2, audio tag 256->10;
// reference sedt/sedt.py:36-38, 90-95, 398-409) as three direct kernels.
//
// An MFMA tile is 32 wide: a GEMM with N = 2..11 wastes it, and the generic kernel's unaligned-shape path takes 10-20 us per
// launch for work that is a few hundred KFLOP.  Here the weight matrix (N <= 16 rows of K floats, read from the f32 MASTER
// parameter - no packing) sits in LDS and every thread owns whole outputs:
//   forward   y[m][n] = act(x[m].w[n] + b[n])                       thread = (row, n)
//   d input   gx[m][k] = sum_n g'[m][n] w[n][k]                     thread = (row, 8 k)
//   d weight  dW[n][k] = sum_m g'[m][n] x[m][k], db[n] = sum_m g'   thread = (k, 4 n) x 16 row groups, LDS tree, deterministic
// with g' = g * y * (1 - y) folded in for the sigmoid heads (no separate sigmoid-gradient / cast launches); g arrives in f32
// exactly as the loss kernel writes it.
#include <algorithm>
#include "common.h"

namespace sedt {

constexpr int SK_MAXN = 16;

template <typename T>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ w,
                                                         const float* __restrict__ bias, void* __restrict__ y, long ldy, int M,
                                                         int N, int K, int act, int out_f32) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int P = K + 4;                  // row pitch: 16-byte aligned rows, float4 reads of 16 different rows hit 64 distinct banks
  float* ws = sm;                       // [N][P]
  float* xs = sm + N * P;               // [16][P]
  const int t = threadIdx.x;
  const int r0 = blockIdx.x * 16;
  // 16-byte global loads when the operands allow it (K % 8 == 0, aligned rows): the tile is 16 x K activations + N x K weights,
  // and with one element per load a thread issued 27 dependent-latency loads before the first multiply
  const bool vec = (K & 7) == 0 && (ldx & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 31) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
  if (vec) {
    const int k4 = K >> 2;
    for (int i = t; i < N * k4; i += 256) {
      const int n = i / k4, q = i - n * k4;
      *reinterpret_cast<float4*>(ws + n * P + 4 * q) = *reinterpret_cast<const float4*>(w + (long)n * K + 4 * q);
    }
    const int k8 = K >> 3;
    for (int i = t; i < 16 * k8; i += 256) {
      const int r = i / k8, q = i - r * k8;
      float v[8];
      if (r0 + r < M) {
        const VecT<T, 8> xv = *reinterpret_cast<const VecT<T, 8>*>(x + (long)(r0 + r) * ldx + 8 * q);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)xv.v[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
      *reinterpret_cast<float4*>(xs + r * P + 8 * q) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(xs + r * P + 8 * q + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
    for (int i = t; i < N * K; i += 256) ws[(i / K) * P + i % K] = w[i];
    for (int i = t; i < 16 * K; i += 256) {
      const int r = i / K, k = i - r * K;
      xs[r * P + k] = r0 + r < M ? (float)x[(long)(r0 + r) * ldx + k] : 0.f;
    }
  }
  __syncthreads();
  const int r = t >> 4, n = t & 15;
  if (n >= N || r0 + r >= M) return;
  const float4* xr = reinterpret_cast<const float4*>(xs + r * P);
  const float4* wr = reinterpret_cast<const float4*>(ws + n * P);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int k = 0; k < K / 4; ++k) {
    const float4 xv = xr[k], wv = wr[k];
    a0 = fmaf(xv.x, wv.x, a0); a1 = fmaf(xv.y, wv.y, a1); a2 = fmaf(xv.z, wv.z, a2); a3 = fmaf(xv.w, wv.w, a3);
  }
  float acc = (a0 + a1) + (a2 + a3);
  if (bias) acc += bias[n];
  if (act == SEDT_ACT_RELU) acc = fmaxf(acc, 0.f);
  else if (act == SEDT_ACT_SIGMOID) acc = 1.f / (1.f + __expf(-acc));
  const long o = (long)(r0 + r) * ldy + n;
  if (out_f32) reinterpret_cast<float*>(y)[o] = acc;
  else reinterpret_cast<T*>(y)[o] = (T)acc;
}

// g' of one (row, n): upstream gradient times the activation derivative
__device__ __forceinline__ float skinny_gprime(const float* g, const float* ysaved, long idx, int act) {
  float v = g[idx];
  if (act == SEDT_ACT_SIGMOID) { const float s = ysaved[idx]; v *= s * (1.f - s); }
  else if (act == SEDT_ACT_RELU) { if (!(ysaved[idx] > 0.f)) v = 0.f; }
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ ysaved, long ldg,
                                                           const float* __restrict__ w, const T* __restrict__ mask, long ldm,
                                                           T* __restrict__ gx, long ldo, int M, int N, int K, int act, int accumulate) {
  extern __shared__ float sm[];
  float* ws = sm;                       // [N][K]
  const int t = threadIdx.x;
  for (int i = t; i < N * K; i += 256) ws[i] = w[i];
  __syncthreads();
  const int cpr = K / 8;                // 8-column chunks per row
  const long u = (long)blockIdx.x * 256 + t;
  if (u >= (long)M * cpr) return;
  const int m = (int)(u / cpr), k0 = (int)(u % cpr) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int n = 0; n < N; ++n) {
    const float gv = skinny_gprime(g, ysaved, (long)m * ldg + n, act);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(gv, ws[n * K + k0 + e], acc[e]);
  }
  if (accumulate) {                     // gx += ...: this head shares its input with others (the running input gradient)
    const VecT<T, 8> old = *reinterpret_cast<const VecT<T, 8>*>(gx + (long)m * ldo + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += (float)old.v[e];
  }
  VecT<T, 8> out;
  if (mask) {
    const VecT<T, 8> mv = *reinterpret_cast<const VecT<T, 8>*>(mask + (long)m * ldm + k0);
#pragma unroll
    for (int e = 0; e < 8; ++e) out.v[e] = ((float)mv.v[e] > 0.f) ? (T)acc[e] : (T)0.f;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) out.v[e] = (T)acc[e];
  }
  *reinterpret_cast<VecT<T, 8>*>(gx + (long)m * ldo + k0) = out;
}

// weight gradient in two launches.  Partial: grid (K / 64, SK_SPLIT row slices); block 256 = 64 columns x 4 row sub-groups.
// The slice's g' rows go to LDS once (activation derivative applied), every thread then walks its rows with one coalesced x
// load and N LDS broadcasts per row.  Final: sums the SK_SPLIT partials in fixed order (deterministic).
constexpr int SK_SPLIT = 16;

template <typename T>
__global__ __launch_bounds__(256) void skinny_wgrad_partial_kernel(const float* __restrict__ g, const float* __restrict__ ysaved,
                                                                   long ldg, const T* __restrict__ x, long ldx,
                                                                   float* __restrict__ part, int M, int N, int K, int act) {
  extern __shared__ __attribute__((aligned(16))) float sm[];     // gs[rows][16], then red[4][16][64]
  const int rows_per = (M + SK_SPLIT - 1) / SK_SPLIT;
  const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per), nr = max(m1 - m0, 0);
  float* gs = sm;
  const int t = threadIdx.x;
  for (int i = t; i < nr * 16; i += 256) {
    const int r = i >> 4, n = i & 15;
    gs[i] = n < N ? skinny_gprime(g, ysaved, (long)(m0 + r) * ldg + n, act) : 0.f;
  }
  __syncthreads();
  const int tx = t & 63, sub = t >> 6;
  const int k = blockIdx.x * 64 + tx;
  float acc[16], bsum[16];
#pragma unroll
  for (int n = 0; n < 16; ++n) { acc[n] = 0.f; bsum[n] = 0.f; }
  for (int r = sub; r < nr; r += 4) {
    const float xv = (float)x[(long)(m0 + r) * ldx + k];
    const float4* gr = reinterpret_cast<const float4*>(gs + r * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 gv = gr[q];
      acc[4 * q] = fmaf(gv.x, xv, acc[4 * q]); acc[4 * q + 1] = fmaf(gv.y, xv, acc[4 * q + 1]);
      acc[4 * q + 2] = fmaf(gv.z, xv, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(gv.w, xv, acc[4 * q + 3]);
      bsum[4 * q] += gv.x; bsum[4 * q + 1] += gv.y; bsum[4 * q + 2] += gv.z; bsum[4 * q + 3] += gv.w;
    }
  }
  __syncthreads();                      // gs is dead: reuse LDS for the cross-sub-group sum
  float* red = sm;                      // [4][17][64]: rows 0..15 = dW columns, row 16 = bias sums (lane tx == n)
#pragma unroll
  for (int n = 0; n < 16; ++n) red[(sub * 17 + n) * 64 + tx] = acc[n];
  if (tx < 16) red[(sub * 17 + 16) * 64 + tx] = bsum[tx];
  __syncthreads();
  // part layout: [split][17][K]: 16 rows of dW partials + one row whose first 16 entries are the bias partials
  float* po = part + (long)blockIdx.y * 17 * K;
  for (int i = t; i < 17 * 64; i += 256) {
    const int n = i >> 6, c = i & 63;
    const float s = red[(0 * 17 + n) * 64 + c] + red[(1 * 17 + n) * 64 + c] + red[(2 * 17 + n) * 64 + c] + red[(3 * 17 + n) * 64 + c];
    if (n < 16) po[(long)n * K + blockIdx.x * 64 + c] = s;
    else if (blockIdx.x == 0 && c < 16) po[(long)16 * K + c] = s;
  }
}

__global__ void skinny_wgrad_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int N,
                                          int K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * K) {
    float s = 0.f;
    for (int z = 0; z < SK_SPLIT; ++z) s += part[(long)z * 17 * K + i];
    dw[i] = s;
  } else if (db && i - N * K < N) {
    const int n = i - N * K;
    float s = 0.f;
    for (int z = 0; z < SK_SPLIT; ++z) s += part[(long)z * 17 * K + 16 * K + n];
    db[n] = s;
  }
}

static inline hipStream_t SS(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_skinny_linear_fwd(const void* x, int64_t ldx, const float* w, const float* bias, void* y, int64_t ldy, int M,
                                      int N, int K, int act, int out_f32, int dtype, void* stream) {
  SEDT_REQUIRE(x && w && y && M > 0 && N >= 1 && N <= SK_MAXN && K >= 8 && K <= 1024, "skinny_linear_fwd: bad arguments (N <= 16, K <= 1024)");
  const size_t lds = (size_t)(N + 16) * (K + 4) * sizeof(float);
  dim3 grid((M + 15) / 16), block(256);
  if (dtype == SEDT_F32)
    hipLaunchKernelGGL(skinny_fwd_kernel<float>, grid, block, lds, SS(stream), (const float*)x, (long)ldx, w, bias, y, (long)ldy, M, N, K,
                       act, out_f32);
  else if (dtype == SEDT_BF16)
    hipLaunchKernelGGL(skinny_fwd_kernel<bf16_t>, grid, block, lds, SS(stream), (const bf16_t*)x, (long)ldx, w, bias, y, (long)ldy, M, N,
                       K, act, out_f32);
  else { set_error("skinny_linear_fwd: unsupported dtype %d", dtype); return 1; }
  return check_launch("skinny_linear_fwd");
}

extern "C" size_t sedt_skinny_linear_bwd_scratch(int K) { return (size_t)SK_SPLIT * 17 * K * sizeof(float); }

extern "C" int sedt_skinny_linear_bwd(const float* g, const float* ysaved, int64_t ldg, const float* w, const void* x, int64_t ldx,
                                      const void* mask, int64_t ldm, void* gx, int64_t ldo, float* dw, float* db, float* scratch,
                                      int M, int N, int K, int act, int accumulate_gx, int dtype, void* stream) {
  SEDT_REQUIRE(g && w && x && M > 0 && N >= 1 && N <= SK_MAXN && K >= 64 && K % 64 == 0 && K <= 1024,
               "skinny_linear_bwd: bad arguments (N <= 16, K a multiple of 64, <= 1024)");
  SEDT_REQUIRE(act == SEDT_ACT_NONE || ysaved, "skinny_linear_bwd: the activation derivative needs the saved output");
  const size_t lds = (size_t)N * K * sizeof(float);
  if (gx) {
    const long nu = (long)M * (K / 8);
    dim3 grid((unsigned)((nu + 255) / 256)), block(256);
    if (dtype == SEDT_F32)
      hipLaunchKernelGGL(skinny_dgrad_kernel<float>, grid, block, lds, SS(stream), g, ysaved, (long)ldg, w, (const float*)mask, (long)ldm,
                         (float*)gx, (long)ldo, M, N, K, act, accumulate_gx);
    else if (dtype == SEDT_BF16)
      hipLaunchKernelGGL(skinny_dgrad_kernel<bf16_t>, grid, block, lds, SS(stream), g, ysaved, (long)ldg, w, (const bf16_t*)mask, (long)ldm,
                         (bf16_t*)gx, (long)ldo, M, N, K, act, accumulate_gx);
    else { set_error("skinny_linear_bwd: unsupported dtype %d", dtype); return 1; }
  }
  if (dw || scratch) {                  // scratch without dw: the partial sums only (the caller reduces the SK_SPLIT slabs itself,
                                        // e.g. as one more column-sum job of a layer's sedt_multi_wgrad_reduce launch)
    SEDT_REQUIRE(scratch != nullptr, "skinny_linear_bwd: dw needs the scratch buffer (sedt_skinny_linear_bwd_scratch bytes)");
    const int rows_per = (M + SK_SPLIT - 1) / SK_SPLIT;
    const size_t lds2 = std::max((size_t)rows_per * 16, (size_t)4 * 17 * 64) * sizeof(float);
    SEDT_REQUIRE(lds2 <= 64 * 1024, "skinny_linear_bwd: M = %d too large for the row-slice LDS buffer", M);
    dim3 grid(K / 64, SK_SPLIT), block(256);
    if (dtype == SEDT_F32)
      hipLaunchKernelGGL(skinny_wgrad_partial_kernel<float>, grid, block, lds2, SS(stream), g, ysaved, (long)ldg, (const float*)x,
                         (long)ldx, scratch, M, N, K, act);
    else if (dtype == SEDT_BF16)
      hipLaunchKernelGGL(skinny_wgrad_partial_kernel<bf16_t>, grid, block, lds2, SS(stream), g, ysaved, (long)ldg, (const bf16_t*)x,
                         (long)ldx, scratch, M, N, K, act);
    else { set_error("skinny_linear_bwd: unsupported dtype %d", dtype); return 1; }
    if (dw) {
      const int nout = N * K + (db ? N : 0);
      hipLaunchKernelGGL(skinny_wgrad_final_kernel, dim3((nout + 255) / 256), dim3(256), 0, SS(stream), scratch, dw, db, N, K);
    }
  }
  return check_launch("skinny_linear_bwd");
}
