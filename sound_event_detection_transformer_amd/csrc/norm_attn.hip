// norm_attn.hip - wave64-reduced LayerNorm (fwd/bwd) and multi-head attention (fwd/bwd) for the
// SEDT transformer (S <= 128 encoder tokens, Q <= 21 decoder queries, head dim 32).
//
// LayerNorm: one wave per row, each lane owns D/64 contiguous features, reductions are
// __shfl_xor butterflies; the backward adds the residual gradient and produces per-block
// gamma/beta partials that a second tiny kernel sums in a fixed order (deterministic).
//
// Attention: the whole K/V (and, backward, Q/dO) of one (batch, head) lives in LDS as f32 with
// pitch 33 (conflict-free row-per-lane reads); a wave owns a query row (forward / dQ pass) or a
// key row (dK,dV pass), lanes run over the other sequence axis (row operands of the owner are held in
// registers, the per-lane rows are read as float4 = ds_read_b128), softmax statistics are wave reductions.  The probabilities are never written to HBM: backward recomputes them from the
// saved log-sum-exp and regenerates the dropout mask from (seed, element index).
#include <stdlib.h>
#include "common.h"

namespace sedt {

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ============================================================================ LayerNorm
template <typename T, int VPT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const T* __restrict__ add,
                                                     T* __restrict__ y, T* __restrict__ y2, float* __restrict__ mean,
                                                     float* __restrict__ rstd, int rows) {
  constexpr int D = VPT * 64;
  typedef VecT<T, VPT> V;                       // one 8/16-byte access per lane and tensor (scalar bf16 accesses halve the rate)
  typedef VecT<float, VPT> VF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const long base = (long)row * D + lane * VPT;
  const V vx = *reinterpret_cast<const V*>(x + base);
  V va;
  if (y2) va = *reinterpret_cast<const V*>(add + base);
  const VF vg = *reinterpret_cast<const VF*>(gamma + lane * VPT), vb = *reinterpret_cast<const VF*>(beta + lane * VPT);
  float v[VPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { v[i] = (float)vx.v[i]; s += v[i]; }
  const float mu = wave_sum(s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) { float d = v[i] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) * (1.f / D) + 1e-5f);
  V vy, vy2;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    float o = (v[i] - mu) * rs * vg.v[i] + vb.v[i];
    vy.v[i] = (T)o;
    if (y2) vy2.v[i] = (T)(o + (float)va.v[i]);
  }
  *reinterpret_cast<V*>(y + base) = vy;
  if (y2) *reinterpret_cast<V*>(y2 + base) = vy2;
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

template <typename T, int VPT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2,
                                                     const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const T* __restrict__ dres, const T* __restrict__ dres2, T* __restrict__ dx,
                                                     float* __restrict__ partial, int rows, T* __restrict__ dx_drop,
                                                     uint32_t thresh, float inv_keep, uint32_t seed,
                                                     const uint32_t* __restrict__ seed_ptr) {
  constexpr int D = VPT * 64;
  typedef VecT<T, VPT> V;                       // one 8/16-byte access per lane and tensor
  __shared__ float red[4][2 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[VPT], dg[VPT], db[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) { g[i] = gamma[lane * VPT + i]; dg[i] = 0.f; db[i] = 0.f; }
  for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
    const long base = (long)row * D + lane * VPT;
    const float mu = mean[row], rs = rstd[row];
    const V vdy = *reinterpret_cast<const V*>(dy + base);
    const V vx = *reinterpret_cast<const V*>(x + base);
    V vdy2, vres, vres2;
    if (dy2) vdy2 = *reinterpret_cast<const V*>(dy2 + base);
    if (dres) vres = *reinterpret_cast<const V*>(dres + base);
    if (dres2) vres2 = *reinterpret_cast<const V*>(dres2 + base);
    float xh[VPT], dyt[VPT];
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      float d = (float)vdy.v[i];
      if (dy2) d += (float)vdy2.v[i];
      dyt[i] = d;
      xh[i] = ((float)vx.v[i] - mu) * rs;
      float dgv = d * g[i];
      c1 += dgv;
      c2 += dgv * xh[i];
      dg[i] += d * xh[i];
      db[i] += d;
    }
    c1 = wave_sum(c1) * (1.f / D);
    c2 = wave_sum(c2) * (1.f / D);
    V out;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      float o = rs * (dyt[i] * g[i] - c1 - xh[i] * c2);
      if (dres) o += (float)vres.v[i];
      if (dres2) o += (float)vres2.v[i];
      out.v[i] = (T)o;
    }
    *reinterpret_cast<V*>(dx + base) = out;
    if (dx_drop) {
      // second output: dx through the dropout mask of the sub-layer whose output fed this LayerNorm (element index =
      // row * D + column, the convention of the GEMM epilogue that dropped it and of dropout_grad_kernel)
      const uint32_t sd = eff_seed(seed, seed_ptr);
      V od;
#pragma unroll
      for (int i = 0; i < VPT; ++i)
        od.v[i] = drop_keep(sd, (uint64_t)(base + i), thresh) ? (T)((float)out.v[i] * inv_keep) : (T)0.f;
      *reinterpret_cast<V*>(dx_drop + base) = od;
    }
  }
#pragma unroll
  for (int i = 0; i < VPT; ++i) { red[wave][lane * VPT + i] = dg[i]; red[wave][D + lane * VPT + i] = db[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256)
    partial[(long)blockIdx.x * 2 * D + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

__global__ __launch_bounds__(1024) void ln_bwd_final_kernel(const float* __restrict__ partial, int nblocks, int D, float* dgamma,
                                                            float* dbeta) {
  __shared__ float red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 columns x 16 row groups
  const int c = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (c < 2 * D) {
    // four independent partial sums: keeps four loads in flight per thread (the loop is latency-, not bandwidth-bound)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = ty;
    for (; b + 48 < nblocks; b += 64) {
      s0 += partial[(long)b * 2 * D + c];
      s1 += partial[(long)(b + 16) * 2 * D + c];
      s2 += partial[(long)(b + 32) * 2 * D + c];
      s3 += partial[(long)(b + 48) * 2 * D + c];
    }
    for (; b < nblocks; b += 16) s0 += partial[(long)b * 2 * D + c];
    s = (s0 + s1) + (s2 + s3);
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < 2 * D) {
    s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    if (c < D) { if (dgamma) dgamma[c] = s; }
    else if (dbeta) dbeta[c - D] = s;
  }
}

static int ln_bwd_blocks(int rows) {
  int b = (rows + 3) / 4;
  return b < 1 ? 1 : (b > 512 ? 512 : b);
}

// ============================================================================ attention
constexpr int DH = 32;      // head dim
constexpr int KP = DH + 4;  // LDS pitch: 16-B aligned rows; 36*r mod 64 hits 16 distinct 16-B slots for 16 consecutive rows
constexpr int MAXKPL = 8;   // keys per lane -> L <= 512

__device__ __forceinline__ float dot32(const float (&a)[DH], const float* __restrict__ row) {
  float acc = 0.f;
#pragma unroll
  for (int e = 0; e < DH; e += 4) {
    const float4 r = *reinterpret_cast<const float4*>(row + e);
    acc += a[e] * r.x + a[e + 1] * r.y + a[e + 2] * r.z + a[e + 3] * r.w;
  }
  return acc;
}

template <typename T>
__device__ __forceinline__ void load_rows_lds(float* dst, const T* src, long ld, int L, int tid, int nthreads) {
  for (int e = tid; e < L * DH; e += nthreads) {
    int r = e / DH, d = e - r * DH;
    dst[r * KP + d] = (float)src[(long)r * ld + d];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk,
                                                       const T* __restrict__ v, long ldv, T* __restrict__ o, long ldo,
                                                       float* __restrict__ lse, const uint8_t* __restrict__ kpm,
                                                       const float* __restrict__ amask, int H, int Lq, int Lk,
                                                       float scale, uint32_t thresh, float inv_keep, uint32_t seed,
                                                       const uint32_t* seed_ptr) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* Ks = lds_f;              // [Lk][33]
  float* Vs = Ks + Lk * KP;       // [Lk][33]
  float* Ps = Vs + Lk * KP;       // [4][LkPad]
  const int LkPad = (Lk + 63) & ~63;
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_rows_lds(Ks, k + (long)b * Lk * ldk + h * DH, ldk, Lk, threadIdx.x, 256);
  load_rows_lds(Vs, v + (long)b * Lk * ldv + h * DH, ldv, Lk, threadIdx.x, 256);
  __syncthreads();
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const int q0 = blockIdx.y * 64;
  float* P = Ps + wave * LkPad;
  for (int it = 0; it < 16; ++it) {      // 64 query rows per block, interleaved over the 4 waves; uniform trip count for the barriers
    const int i = q0 + it * 4 + wave;
    const bool live = i < Lq;
    float qr[DH];
    if (live) {
      const T* qp = q + ((long)b * Lq + i) * ldq + h * DH;
#pragma unroll
      for (int d = 0; d < DH; ++d) qr[d] = (float)qp[d] * scale;
    }
    float s[MAXKPL];
    float m = -INFINITY;
    if (live) {
#pragma unroll
      for (int c = 0; c < MAXKPL; ++c) {
        const int j = lane + c * 64;
        s[c] = -INFINITY;
        if (j < Lk) {
          float a = dot32(qr, Ks + j * KP);
          if (amask) a += amask[(long)i * Lk + j];
          if (kpm && kpm[(long)b * Lk + j]) a = -INFINITY;
          s[c] = a;
        }
        m = fmaxf(m, s[c]);
      }
      m = wave_max(m);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < MAXKPL; ++c) {
        const int j = lane + c * 64;
        float e = (j < Lk) ? __expf(s[c] - m) : 0.f;
        s[c] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      if (lane == 0) lse[((long)b * H + h) * Lq + i] = m + __logf(sum);
#pragma unroll
      for (int c = 0; c < MAXKPL; ++c) {
        const int j = lane + c * 64;
        if (j < Lk) {
          float pv = s[c] * inv;
          if (thresh) pv = drop_keep(sd, ((uint64_t)bh * Lq + i) * Lk + j, thresh) ? pv * inv_keep : 0.f;
          P[j] = pv;
        }
      }
    }
    __syncthreads();
    if (live) {
      const int d = lane & 31, half = lane >> 5;
      float acc = 0.f;
#pragma unroll 8
      for (int j = half; j < Lk; j += 2) acc += P[j] * Vs[j * KP + d];
      acc += __shfl_xor(acc, 32, 64);
      if (lane < 32) o[((long)b * Lq + i) * ldo + h * DH + d] = (T)acc;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ q, long ldq, const T* __restrict__ k, long ldk,
                                                       const T* __restrict__ v, long ldv, const T* __restrict__ o, long ldo,
                                                       const T* __restrict__ dout, long lddo, const float* __restrict__ lse,
                                                       const uint8_t* __restrict__ kpm, const float* __restrict__ amask,
                                                       T* __restrict__ dq, long lddq, T* __restrict__ dk, long lddk,
                                                       T* __restrict__ dv, long lddv, int H, int Lq, int Lk, float scale,
                                                       uint32_t thresh, float inv_keep, uint32_t seed,
                                                       const uint32_t* seed_ptr) {
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  const int LmPad = (max(Lq, Lk) + 63) & ~63;
  float* Qs = lds_f;               // [Lq][33]
  float* Ks = Qs + Lq * KP;        // [Lk][33]
  float* Vs = Ks + Lk * KP;        // [Lk][33]
  float* Ds = Vs + Lk * KP;        // dO [Lq][33]
  float* Ls = Ds + Lq * KP;        // lse [Lq]
  float* De = Ls + Lq;             // delta [Lq]
  float* W1 = De + Lq;             // [4][LmPad]  dS scratch
  float* W2 = W1 + 4 * LmPad;      // [4][LmPad]  dropped-P scratch
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  load_rows_lds(Qs, q + (long)b * Lq * ldq + h * DH, ldq, Lq, threadIdx.x, 256);
  load_rows_lds(Ks, k + (long)b * Lk * ldk + h * DH, ldk, Lk, threadIdx.x, 256);
  load_rows_lds(Vs, v + (long)b * Lk * ldv + h * DH, ldv, Lk, threadIdx.x, 256);
  load_rows_lds(Ds, dout + (long)b * Lq * lddo + h * DH, lddo, Lq, threadIdx.x, 256);
  for (int i = threadIdx.x; i < Lq; i += 256) Ls[i] = lse[((long)b * H + h) * Lq + i];
  __syncthreads();
  for (int i = threadIdx.x; i < Lq; i += 256) {
    const T* op = o + ((long)b * Lq + i) * ldo + h * DH;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) a += Ds[i * KP + d] * (float)op[d];
    De[i] = a;
  }
  __syncthreads();
  const uint32_t sd = eff_seed(seed, seed_ptr);
  float* dSw = W1 + wave * LmPad;
  float* Pdw = W2 + wave * LmPad;
  const int d = lane & 31, half = lane >> 5;

  // -------- pass 1 (blockIdx.y == 0): a wave owns query row i, lanes run over keys -> dQ
  const int nqi = blockIdx.y == 0 ? (Lq + 3) / 4 : 0;
  for (int it = 0; it < nqi; ++it) {
    const int i = it * 4 + wave;
    const bool live = i < Lq;
    if (live) {
      const float li = Ls[i], di = De[i];
      float qi[DH], doi[DH];
#pragma unroll
      for (int e = 0; e < DH; ++e) { qi[e] = Qs[i * KP + e]; doi[e] = Ds[i * KP + e]; }
      for (int j = lane; j < Lk; j += 64) {
        float s = dot32(qi, Ks + j * KP), dp = dot32(doi, Vs + j * KP);
        s *= scale;
        if (amask) s += amask[(long)i * Lk + j];
        if (kpm && kpm[(long)b * Lk + j]) s = -INFINITY;
        const float p = __expf(s - li);
        if (thresh) dp = drop_keep(sd, ((uint64_t)bh * Lq + i) * Lk + j, thresh) ? dp * inv_keep : 0.f;
        dSw[j] = p * (dp - di);
      }
    }
    __syncthreads();
    if (live) {
      float acc = 0.f;
#pragma unroll 8
      for (int j = half; j < Lk; j += 2) acc += dSw[j] * Ks[j * KP + d];
      acc += __shfl_xor(acc, 32, 64);
      if (lane < 32) dq[((long)b * Lq + i) * lddq + h * DH + d] = (T)(acc * scale);
    }
    __syncthreads();
  }

  // -------- pass 2 (blockIdx.y == 1): a wave owns key row j, lanes run over queries -> dK, dV
  const int nkj = blockIdx.y == 1 ? (Lk + 3) / 4 : 0;
  for (int it = 0; it < nkj; ++it) {
    const int j = it * 4 + wave;
    const bool live = j < Lk;
    if (live) {
      const bool padded = kpm && kpm[(long)b * Lk + j];
      float kj[DH], vj[DH];
#pragma unroll
      for (int e = 0; e < DH; ++e) { kj[e] = Ks[j * KP + e]; vj[e] = Vs[j * KP + e]; }
      for (int i = lane; i < Lq; i += 64) {
        float s = dot32(kj, Qs + i * KP), dp = dot32(vj, Ds + i * KP);
        s *= scale;
        if (amask) s += amask[(long)i * Lk + j];
        if (padded) s = -INFINITY;
        const float p = __expf(s - Ls[i]);
        float pd = p;
        if (thresh) {
          const bool keep = drop_keep(sd, ((uint64_t)bh * Lq + i) * Lk + j, thresh);
          dp = keep ? dp * inv_keep : 0.f;
          pd = keep ? p * inv_keep : 0.f;
        }
        dSw[i] = p * (dp - De[i]);
        Pdw[i] = pd;
      }
    }
    __syncthreads();
    if (live) {
      float ak = 0.f, av = 0.f;
#pragma unroll 8
      for (int i = half; i < Lq; i += 2) {
        ak += dSw[i] * Qs[i * KP + d];
        av += Pdw[i] * Ds[i * KP + d];
      }
      ak += __shfl_xor(ak, 32, 64);
      av += __shfl_xor(av, 32, 64);
      if (lane < 32) {
        dk[((long)b * Lk + j) * lddk + h * DH + d] = (T)(ak * scale);
        dv[((long)b * Lk + j) * lddv + h * DH + d] = (T)av;
      }
    }
    __syncthreads();
  }
}

template <typename K>
static int set_lds_attr(K kern, size_t bytes, const char* what) {
  static bool done = false;   // one static per kernel instantiation: never called again (e.g. during graph capture)
  if (done) return 0;
  done = true;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%zu B LDS) failed: %s", what, bytes, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

}  // namespace sedt

namespace sedt {
int attn_fwd_mfma_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                      float* lse, const uint8_t* kpm, const float* amask, int B, int H, int Lq, int Lk, float drop_p,
                      uint32_t seed, const uint32_t* seed_ptr, hipStream_t st);
int attn_bwd_mfma_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                      int64_t ldo, const void* dout, int64_t lddo, const float* lse, const uint8_t* kpm, const float* amask,
                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int B, int H, int Lq, int Lk,
                      float drop_p, uint32_t seed, const uint32_t* seed_ptr, hipStream_t st);
int attn_f32_fwd_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                     const uint8_t* kpm, const float* amask, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                     const uint32_t* seed_ptr, hipStream_t st);
int attn_f32_bwd_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                     const void* dout, int64_t lddo, const float* lse, const uint8_t* kpm, const float* amask, void* dq, int64_t lddq,
                     void* dk, int64_t lddk, void* dv, int64_t lddv, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                     const uint32_t* seed_ptr, hipStream_t st);
static bool use_attn_mfma() {
  static int v = -1;
  if (v < 0) {
    const char* e = sedt::dev_getenv("SEDT_ATTN_MFMA");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}
}  // namespace sedt

using namespace sedt;

extern "C" int sedt_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* add, void* y, void* y2,
                                  float* mean, float* rstd, int rows, int D, int dtype, void* stream) {
  SEDT_REQUIRE(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  SEDT_REQUIRE((y2 == nullptr) == (add == nullptr), "layernorm_fwd: y2 and add go together");
  dim3 grid((rows + 3) / 4), block(256);
#define A_(T) grid, block, 0, S(stream), (const T*)x, gamma, beta, (const T*)add, (T*)y, (T*)y2, mean, rstd, rows
  if (dtype == SEDT_F32 && D == 256) hipLaunchKernelGGL((ln_fwd_kernel<float, 4>), A_(float));
  else if (dtype == SEDT_F32 && D == 512) hipLaunchKernelGGL((ln_fwd_kernel<float, 8>), A_(float));
  else if (dtype == SEDT_BF16 && D == 256) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 4>), A_(bf16_t));
  else if (dtype == SEDT_BF16 && D == 512) hipLaunchKernelGGL((ln_fwd_kernel<bf16_t, 8>), A_(bf16_t));
  else { set_error("layernorm: unsupported dtype %d / width %d (256 or 512)", dtype, D); return 1; }
#undef A_
  return check_launch("layernorm_fwd");
}

extern "C" size_t sedt_layernorm_bwd_scratch(int rows, int D) { return (size_t)ln_bwd_blocks(rows) * 2 * D * sizeof(float); }

extern "C" int sedt_layernorm_bwd_drop(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, const void* dres, const void* dres2, void* dx, float* dgamma, float* dbeta,
                                       float* scratch, size_t scratch_bytes, int rows, int D, void* dx_drop, float drop_p,
                                       uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream);

extern "C" int sedt_layernorm_bwd(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean,
                                  const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* scratch,
                                  size_t scratch_bytes, int rows, int D, int dtype, void* stream) {
  return sedt_layernorm_bwd_drop(dy, dy2, x, gamma, mean, rstd, dres, nullptr, dx, dgamma, dbeta, scratch, scratch_bytes, rows, D,
                                 nullptr, 0.f, 0u, nullptr, dtype, stream);
}

extern "C" int sedt_layernorm_bwd_drop(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, const void* dres, const void* dres2, void* dx, float* dgamma, float* dbeta,
                                       float* scratch, size_t scratch_bytes, int rows, int D, void* dx_drop, float drop_p,
                                       uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream) {
  SEDT_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "layernorm_bwd: drop_p out of range");
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  SEDT_REQUIRE(scratch && scratch_bytes >= sedt_layernorm_bwd_scratch(rows, D), "layernorm_bwd: scratch too small");
  int nb = ln_bwd_blocks(rows);
  dim3 grid(nb), block(256);
#define A_(T) grid, block, 0, S(stream), (const T*)dy, (const T*)dy2, (const T*)x, gamma, mean, rstd, (const T*)dres, (const T*)dres2, (T*)dx, scratch, rows, (T*)dx_drop, th, ik, seed, seed_ptr
  if (dtype == SEDT_F32 && D == 256) hipLaunchKernelGGL((ln_bwd_kernel<float, 4>), A_(float));
  else if (dtype == SEDT_F32 && D == 512) hipLaunchKernelGGL((ln_bwd_kernel<float, 8>), A_(float));
  else if (dtype == SEDT_BF16 && D == 256) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 4>), A_(bf16_t));
  else if (dtype == SEDT_BF16 && D == 512) hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 8>), A_(bf16_t));
  else { set_error("layernorm: unsupported dtype %d / width %d (256 or 512)", dtype, D); return 1; }
#undef A_
  if (dgamma || dbeta)
    hipLaunchKernelGGL(ln_bwd_final_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, S(stream), scratch, nb, D, dgamma, dbeta);
  return check_launch("layernorm_bwd");
}

extern "C" int sedt_layernorm_bwd_final(const float* scratch, int rows, int D, float* dgamma, float* dbeta, void* stream) {
  SEDT_REQUIRE(scratch && (dgamma || dbeta) && rows > 0 && D > 0, "layernorm_bwd_final: bad arguments");
  hipLaunchKernelGGL(ln_bwd_final_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, S(stream), scratch, ln_bwd_blocks(rows), D, dgamma,
                     dbeta);
  return check_launch("layernorm_bwd_final");
}

extern "C" int sedt_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                                  int64_t ldo, float* lse, const uint8_t* kpm, const float* amask, int B, int H, int Lq, int Lk,
                                  float drop_p, uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream) {
  SEDT_REQUIRE(q && k && v && o && lse, "attention_fwd: null pointer");
  SEDT_REQUIRE(Lk >= 1 && Lk <= 64 * MAXKPL && Lq >= 1, "attention_fwd: Lk=%d out of range (1..%d)", Lk, 64 * MAXKPL);
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_fwd: drop_p out of range");
  if (dtype == SEDT_BF16 && use_attn_mfma()) {          // MFMA kernels (attn_mfma.hip) when the problem fits
    int r = attn_fwd_mfma_try(q, ldq, k, ldk, v, ldv, o, ldo, lse, kpm, amask, B, H, Lq, Lk, drop_p, seed, seed_ptr, S(stream));
    if (r >= 0) return r;
  }
  if (dtype == SEDT_F32) {                               // exact-f32 MFMA kernels (attn_f32_mfma.hip): Lq, Lk <= 128
    int r = attn_f32_fwd_try(q, ldq, k, ldk, v, ldv, o, ldo, lse, kpm, amask, B, H, Lq, Lk, drop_p, seed, seed_ptr, S(stream));
    if (r >= 0) return r;
  }
  const int LkPad = (Lk + 63) & ~63;
  size_t lds = ((size_t)2 * Lk * KP + 4 * LkPad) * sizeof(float);
  SEDT_REQUIRE(lds <= 160 * 1024, "attention_fwd: Lk=%d needs %zu B of LDS", Lk, lds);
  const float scale = 1.f / sqrtf((float)DH);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H, (Lq + 63) / 64), block(256);
  if (dtype == SEDT_F32) {
    if (set_lds_attr(attn_fwd_kernel<float>, 160 * 1024, "attention_fwd")) return 1;
    hipLaunchKernelGGL(attn_fwd_kernel<float>, grid, block, lds, S(stream), (const float*)q, (long)ldq, (const float*)k, (long)ldk,
                       (const float*)v, (long)ldv, (float*)o, (long)ldo, lse, kpm, amask, H, Lq, Lk, scale, th, ik, seed, seed_ptr);
  } else if (dtype == SEDT_BF16) {
    if (set_lds_attr(attn_fwd_kernel<bf16_t>, 160 * 1024, "attention_fwd")) return 1;
    hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, grid, block, lds, S(stream), (const bf16_t*)q, (long)ldq, (const bf16_t*)k,
                       (long)ldk, (const bf16_t*)v, (long)ldv, (bf16_t*)o, (long)ldo, lse, kpm, amask, H, Lq, Lk, scale, th, ik,
                       seed, seed_ptr);
  } else { set_error("attention_fwd: unsupported dtype %d", dtype); return 1; }
  return check_launch("attention_fwd");
}

extern "C" int sedt_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                  const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                                  const uint8_t* kpm, const float* amask, void* dq, int64_t lddq, void* dk, int64_t lddk,
                                  void* dv, int64_t lddv, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                                  const uint32_t* seed_ptr, int dtype, void* stream) {
  SEDT_REQUIRE(q && k && v && o && dout && lse && dq && dk && dv, "attention_bwd: null pointer");
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "attention_bwd: drop_p out of range");
  if (dtype == SEDT_BF16 && use_attn_mfma()) {
    int r = attn_bwd_mfma_try(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, kpm, amask, dq, lddq, dk, lddk, dv, lddv, B, H, Lq,
                              Lk, drop_p, seed, seed_ptr, S(stream));
    if (r >= 0) return r;
  }
  if (dtype == SEDT_F32) {
    int r = attn_f32_bwd_try(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, kpm, amask, dq, lddq, dk, lddk, dv, lddv, B, H, Lq, Lk,
                             drop_p, seed, seed_ptr, S(stream));
    if (r >= 0) return r;
  }
  const int LmPad = (std::max(Lq, Lk) + 63) & ~63;
  size_t lds = ((size_t)2 * Lq * KP + 2 * Lk * KP + 2 * Lq + 8 * LmPad) * sizeof(float);
  SEDT_REQUIRE(lds <= 160 * 1024, "attention_bwd: Lq=%d Lk=%d need %zu B of LDS", Lq, Lk, lds);
  const float scale = 1.f / sqrtf((float)DH);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H, 2), block(256);
  if (dtype == SEDT_F32) {
    if (set_lds_attr(attn_bwd_kernel<float>, 160 * 1024, "attention_bwd")) return 1;
    hipLaunchKernelGGL(attn_bwd_kernel<float>, grid, block, lds, S(stream), (const float*)q, (long)ldq, (const float*)k, (long)ldk,
                       (const float*)v, (long)ldv, (const float*)o, (long)ldo, (const float*)dout, (long)lddo, lse, kpm, amask,
                       (float*)dq, (long)lddq, (float*)dk, (long)lddk, (float*)dv, (long)lddv, H, Lq, Lk, scale, th, ik, seed,
                       seed_ptr);
  } else if (dtype == SEDT_BF16) {
    if (set_lds_attr(attn_bwd_kernel<bf16_t>, 160 * 1024, "attention_bwd")) return 1;
    hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, grid, block, lds, S(stream), (const bf16_t*)q, (long)ldq, (const bf16_t*)k,
                       (long)ldk, (const bf16_t*)v, (long)ldv, (const bf16_t*)o, (long)ldo, (const bf16_t*)dout, (long)lddo, lse,
                       kpm, amask, (bf16_t*)dq, (long)lddq, (bf16_t*)dk, (long)lddk, (bf16_t*)dv, (long)lddv, H, Lq, Lk, scale, th,
                       ik, seed, seed_ptr);
  } else { set_error("attention_bwd: unsupported dtype %d", dtype); return 1; }
  return check_launch("attention_bwd");
}
