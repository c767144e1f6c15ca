// misc.hip - HBM-bound support kernels of the SEDT path: weight packing, FrozenBN fold, stem
// im2col, max/avg pooling, position encoding, mask resize, column sums (bias grads), dropout
// gradient, split-K wgrad reduction, global grad-norm + fused clip/AdamW.
// All are coalesced along the channel (fastest) axis of the NHWC / [rows][cols] layouts.
#include <stdarg.h>
#include <string.h>
#include <type_traits>
#include <algorithm>
#include "common.h"

namespace sedt {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline unsigned nblk(long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------ wgrad split-K reduce

struct ReduceJobs {
  int n;
  SedtReduceJob j[SEDT_MAX_REDUCE_JOBS];
  // what the NEXT launch streams (SedtPrefetch: the next encoder layer's backward weights): touched by the first 256 blocks, one
  // 128-byte line per load, so that every XCD's L2 holds them when that launch's workgroups stream them in lockstep
  const uint32_t* pf[3]; int pf_lines[3];
};

// blocks of one job: taps == 1 (and 4 | Ci): 1024 consecutive elements per block, float4 per thread;
// taps > 1 and 64 | Ci: one (row, 64-channel chunk) with all its taps per block - the [tap][c] -> [c][tap] transposition
// goes through LDS so that slab reads and gradient writes are both contiguous runs; otherwise 256 scalar elements.
// (taps == 1 with many slices - a tiny output under a very long K, e.g. the stem: mode 3 = 256 elements per block, the
// slices dealt to 4 thread groups.)  The wide modes need 16-byte aligned slabs / outputs.
__host__ __device__ inline int reduce_job_mode(const SedtReduceJob& J) {
  if ((reinterpret_cast<uintptr_t>(J.slab) | reinterpret_cast<uintptr_t>(J.out)) & 15) return 2;
  if (J.taps == 1 && (J.Ci & 3) == 0) return J.splitk >= 16 ? 3 : 0;
  if (J.taps > 1 && J.taps <= 9 && (J.Ci & 63) == 0) return 1;
  return 2;
}
__host__ __device__ inline int reduce_job_blocks(const SedtReduceJob& J) {
  const long per = (long)J.R * J.taps * J.Ci;
  const int mode = reduce_job_mode(J);
  if (mode == 0) return (int)((per + 1023) / 1024);
  if (mode == 1) return J.R * (J.Ci / 64);
  return (int)((per + 255) / 256);       // modes 2 and 3
}

__global__ __launch_bounds__(256) void multi_wgrad_reduce_kernel(const ReduceJobs jobs) {
  __shared__ float tile[9][65];
  if (blockIdx.x < 256 && jobs.pf_lines[0] + jobs.pf_lines[1] + jobs.pf_lines[2] > 0) {
    // (blocks go round-robin over the 8 XCDs: 32 blocks per XCD share the lines; the values are never used)
    const int slot = blockIdx.x >> 3, nslots = min(32, max(1, (int)(gridDim.x >> 3)));
    uint32_t acc = 0;
    for (int r = 0; r < 3; ++r)
      for (int j = slot * 256 + threadIdx.x; j < jobs.pf_lines[r]; j += nslots * 256) acc ^= jobs.pf[r][(long)j * 32];
    if (acc == 0x9e3779b9u && jobs.n < 0) tile[0][0] = 1.f;
  }
  int lo = 0, hi = jobs.n - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs.j[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SedtReduceJob& J = jobs.j[lo];
  const int blk = blockIdx.x - J.blk0;
  const long per = (long)J.R * J.taps * J.Ci;
  if (J.colsum_slab) {
    // column sums over the split slices: the fused bias gradient of a wgrad, or (per == 0, splitk = partial rows) the
    // gamma/beta reduction of a LayerNorm backward riding in the same launch.  The first ceil(R / 64) blocks of the job take
    // 64 columns each; the slices are dealt to 4 thread groups x 4 independent sums (a LayerNorm job has up to 512 slices:
    // one thread walking them alone was the tail of the whole launch), combined in a fixed order.
    __shared__ float csum[4][64];
    const int col = blk * 64 + (threadIdx.x & 63), zg = threadIdx.x >> 6;
    if (blk * 64 < J.R) {
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      if (col < J.R) {
        const float* cp = J.colsum_slab + col;
        const int csk = J.cs_splitk > 0 ? J.cs_splitk : J.splitk;      // (the column sums may have fewer slices than the slabs: bf16x3)
        int z = zg;
        for (; z + 12 < csk; z += 16) {
          b0 += cp[(long)z * J.R];
          b1 += cp[(long)(z + 4) * J.R];
          b2 += cp[(long)(z + 8) * J.R];
          b3 += cp[(long)(z + 12) * J.R];
        }
        for (; z < csk; z += 4) b0 += cp[(long)z * J.R];
      }
      csum[zg][threadIdx.x & 63] = (b0 + b1) + (b2 + b3);
      __syncthreads();
      if (zg == 0 && col < J.R) J.bias_out[col] = (csum[0][col & 63] + csum[1][col & 63]) + (csum[2][col & 63] + csum[3][col & 63]);
    }
  }
  const int mode = reduce_job_mode(J);
  if (mode == 3) {
    __shared__ float4 zsum[4][64];
    const int q = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const long e = ((long)blk * 64 + q) * 4;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (e < per) {
      const float* sp = J.slab + e;
      int z = zg;
      for (; z + 4 < J.splitk; z += 8) {
        const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
        const float4 b = *reinterpret_cast<const float4*>(sp + (long)(z + 4) * per);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      }
      if (z < J.splitk) {
        const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
        s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      }
    }
    zsum[zg][q] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (zg == 0 && e < per) {
      const float4 a = zsum[0][q], b = zsum[1][q], c = zsum[2][q], d = zsum[3][q];
      float4 s = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
      if (J.rowscale) {
        const float sc = J.rowscale[e / J.Ci];
        s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
      }
      *reinterpret_cast<float4*>(J.out + e) = s;
    }
  } else if (mode == 0) {
    const long e = ((long)blk * 256 + threadIdx.x) * 4;
    if (e >= per) return;
    // four independent partial sums: the slices are read as parallel streams, not as one dependent chain of loads
    const float* sp = J.slab + e;
    float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, s3 = s1;
    int z = 1;
    for (; z + 4 <= J.splitk; z += 4) {
      const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
      const float4 b = *reinterpret_cast<const float4*>(sp + (long)(z + 1) * per);
      const float4 c = *reinterpret_cast<const float4*>(sp + (long)(z + 2) * per);
      const float4 d = *reinterpret_cast<const float4*>(sp + (long)(z + 3) * per);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
      s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; z < J.splitk; ++z) {
      const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    float4 s = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                           (s0.w + s1.w) + (s2.w + s3.w));
    if (J.rowscale) {
      const float sc = J.rowscale[e / J.Ci];
      s.x *= sc; s.y *= sc; s.z *= sc; s.w *= sc;
    }
    *reinterpret_cast<float4*>(J.out + e) = s;
  } else if (mode == 1) {
    const int chunks = J.Ci / 64;
    const int r = blk / chunks, c0 = (blk - r * chunks) * 64;
    if (r >= J.R) return;                       // surplus blocks that only carried bias sums
    const int n = J.taps * 64;
    const float sc = J.rowscale ? J.rowscale[r] : 1.f;
    const long rbase = (long)r * J.taps * J.Ci;
    for (int i = threadIdx.x; i < n / 4; i += 256) {          // float4 along the channel run of one tap, two slice streams
      const int tap = i >> 4, c = (i & 15) * 4;
      const float* sp = J.slab + rbase + (long)tap * J.Ci + c0 + c;
      float4 v = *reinterpret_cast<const float4*>(sp), u = make_float4(0.f, 0.f, 0.f, 0.f);
      int z = 1;
      for (; z + 2 <= J.splitk; z += 2) {
        const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
        const float4 b = *reinterpret_cast<const float4*>(sp + (long)(z + 1) * per);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        u.x += b.x; u.y += b.y; u.z += b.z; u.w += b.w;
      }
      if (z < J.splitk) {
        const float4 a = *reinterpret_cast<const float4*>(sp + (long)z * per);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      tile[tap][c] = (v.x + u.x) * sc; tile[tap][c + 1] = (v.y + u.y) * sc;
      tile[tap][c + 2] = (v.z + u.z) * sc; tile[tap][c + 3] = (v.w + u.w) * sc;
    }
    __syncthreads();
    float* o = J.out + ((long)r * J.Ci + c0) * J.taps;      // 64 * taps contiguous floats, 16-byte aligned (64 * taps * 4 bytes per chunk)
    if (J.taps == 9) {                               // compile-time divisor for the 3x3 case
      for (int i = threadIdx.x; i < n / 4; i += 256) {
        float4 q;
        { const int k = 4 * i, c = k / 9, tap = k - c * 9; q.x = tile[tap][c]; }
        { const int k = 4 * i + 1, c = k / 9, tap = k - c * 9; q.y = tile[tap][c]; }
        { const int k = 4 * i + 2, c = k / 9, tap = k - c * 9; q.z = tile[tap][c]; }
        { const int k = 4 * i + 3, c = k / 9, tap = k - c * 9; q.w = tile[tap][c]; }
        *reinterpret_cast<float4*>(o + 4 * i) = q;
      }
    } else {
      for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / J.taps, tap = i - c * J.taps;
        o[i] = tile[tap][c];
      }
    }
  } else {
    const long e = (long)blk * 256 + threadIdx.x;
    if (e >= per) return;
    const int c = (int)(e % J.Ci);
    const long rt = e / J.Ci;
    const int tap = (int)(rt % J.taps);
    const int r = (int)(rt / J.taps);
    float s = 0.f;
    for (int z = 0; z < J.splitk; ++z) s += J.slab[(long)z * per + e];
    if (J.rowscale) s *= J.rowscale[r];
    J.out[((long)r * J.Ci + c) * J.taps + tap] = s;
  }
}

// ------------------------------------------------------------------ column sums
template <typename TI>
__global__ void colsum_kernel(const TI* __restrict__ in, long ld, int rows, int cols, int rows_per_chunk,
                              float* __restrict__ out) {
  __shared__ float red[4][64];
  int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int col = blockIdx.x * 64 + tx;
  int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float s = 0.f;
  if (col < cols)
    for (int r = r0 + ty; r < r1; r += 4) s += (float)in[(long)r * ld + col];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && col < cols) out[(long)blockIdx.y * cols + col] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

static int colsum_chunks(int rows) {
  int c = (rows + 255) / 256;
  return c < 1 ? 1 : (c > 256 ? 256 : c);
}

// ------------------------------------------------------------------ elementwise
template <typename T>
__global__ void dropout_grad_kernel(const T* __restrict__ in, long ldi, T* __restrict__ out, long ldo, int rows, int cols,
                                    uint32_t thresh, float inv_keep, uint32_t seed, const uint32_t* seed_ptr) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  int r = (int)(e / cols), c = (int)(e % cols);
  uint32_t sd = eff_seed(seed, seed_ptr);
  float v = (float)in[(long)r * ldi + c];
  out[(long)r * ldo + c] = (T)(drop_keep(sd, (uint64_t)e, thresh) ? v * inv_keep : 0.f);
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int rows, int cols,
                           int b_mod) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  int r = (int)(e / cols), c = (int)(e % cols);
  int rb = b_mod > 0 ? r % b_mod : r;
  out[e] = (T)((float)a[e] + (float)b[(long)rb * cols + c]);
}

// out = sum of up to 8 equally shaped tensors (fixed order): the gradient of a tensor with several consumers in ONE launch (the
// decoder's query position embedding feeds two attention blocks in each of its layers; autograd would add them pair by pair)
struct AddN { const void* p[8]; int n; };
template <typename T>
__global__ void add_n_kernel(const AddN a, T* __restrict__ out, long n8) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n8) return;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int j = 0; j < a.n; ++j) {
    const VecT<T, 8> v = reinterpret_cast<const VecT<T, 8>*>(a.p[j])[e];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += (float)v.v[i];
  }
  VecT<T, 8> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o.v[i] = (T)acc[i];
  reinterpret_cast<VecT<T, 8>*>(out)[e] = o;
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ in, TO* __restrict__ out, long n) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = (TO)(float)in[e];
}

template <typename T>
__global__ void relu_mask_kernel(const T* __restrict__ g, const T* __restrict__ y, T* __restrict__ out, long n) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = ((float)y[e] > 0.f) ? g[e] : (T)0.f;
}

// GELU (erf form: torch's F.gelu default, reference transformer.py:423-431 "gelu") followed by the FFN's dropout:
//   a = drop(gelu(h));  backward g_h = drop'(g_a) * (Phi(h) + h * phi(h)).  The keep decision is the epilogue dropout's
// (seed, row * cols + col) hash, so forward and backward agree without a stored mask.  8 elements per thread.
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ h, T* __restrict__ a, long n8, uint32_t thresh, float inv_keep, uint32_t seed,
                                const uint32_t* seed_ptr) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n8) return;
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const VecT<T, 8> v = reinterpret_cast<const VecT<T, 8>*>(h)[e];
  VecT<T, 8> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = (float)v.v[i];
    const float y = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    o.v[i] = (T)((thresh == 0u || drop_keep(sd, (uint64_t)(e * 8 + i), thresh)) ? y * inv_keep : 0.f);
  }
  reinterpret_cast<VecT<T, 8>*>(a)[e] = o;
}

template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ h, T* __restrict__ out, long n8, uint32_t thresh,
                                float inv_keep, uint32_t seed, const uint32_t* seed_ptr) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n8) return;
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const VecT<T, 8> gv = reinterpret_cast<const VecT<T, 8>*>(g)[e];
  const VecT<T, 8> hv = reinterpret_cast<const VecT<T, 8>*>(h)[e];
  VecT<T, 8> o;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = (float)hv.v[i];
    const float d = 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.39894228040143268f * expf(-0.5f * x * x);
    o.v[i] = (T)((thresh == 0u || drop_keep(sd, (uint64_t)(e * 8 + i), thresh)) ? (float)gv.v[i] * d * inv_keep : 0.f);
  }
  reinterpret_cast<VecT<T, 8>*>(out)[e] = o;
}

__global__ void sigmoid_grad_kernel(const float* g, const float* s, float* out, long n) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = g[e] * s[e] * (1.f - s[e]);
}

// ------------------------------------------------------------------ strided 2-D copies, several per launch
// job: `outer` items of `inner` bytes (a multiple of 4), items src_stride / dst_stride bytes apart.  The mean-teacher step runs ONE student
// forward over labelled + unlabelled clips and two criterion calls on the clip ranges of its stacked head outputs [L][B][Q][C]
// (engine.GraphedSemiStep): the six range copies are one launch, and so is the merge of the six gradient parts in the backward (as
// autograd slices: ~20 torch copy / fill / add launches per step).
constexpr int COPY_MAXJ = 8;
struct CopyJobs {
  int n;
  SedtCopyJob j[COPY_MAXJ];
};
__global__ __launch_bounds__(256) void copy2d_kernel(const CopyJobs jobs) {
  int i = 0;
  while (i + 1 < jobs.n && (int)blockIdx.x >= jobs.j[i + 1].blk0) ++i;
  const SedtCopyJob& J = jobs.j[i];
  const long w = (long)(blockIdx.x - J.blk0) * 256 + threadIdx.x;          // 4-byte word of the job
  const long per = J.inner / 4;
  if (w >= per * J.outer) return;
  const long o = w / per, k = w - o * per;
  reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(J.dst) + o * J.dst_stride)[k] =
      reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(J.src) + o * J.src_stride)[k];
}

// ------------------------------------------------------------------ SP-SEDT decoder input (reference sedt/spsedt.py:48-69)
// training: dec_in[b][q] = 2 * query[q] + keep(q, b) * patch[b][q / qpp]     (spsedt.py:65-67: decoder_input += patches * mask + decoder_input)
// eval:     dec_in[b][q] = query[q] + patch[b][q / qpp]
// The reference builds it with repeat / flatten / permute / contiguous copies, a torch.rand draw, a compare and three elementwise ops
// (~10 launches and their autograd nodes); here one launch each way.  Token-major [B*Q][D] output (the decoder's query-position rows),
// keep(q, b) either given (f32 [Q][B]: tests inject the reference's draw) or drawn from the counter hash of (seed, q * B + b) with
// probability 1 - ratio and written to keep_out for the backward.  D = 256: one thread per 8 columns, 32 threads per row.
template <typename T>
__global__ void spsedt_dec_in_kernel(const T* __restrict__ patch, const float* __restrict__ query, const float* __restrict__ keep_in,
                                     float* __restrict__ keep_out, T* __restrict__ out, int B, int Q, int P, int qpp, int D, int train,
                                     uint32_t thresh, uint32_t seed, const uint32_t* seed_ptr) {
  const int row = blockIdx.x * (blockDim.x / 32) + threadIdx.x / 32;          // b * Q + q
  if (row >= B * Q) return;
  const int b = row / Q, q = row - b * Q, c0 = (threadIdx.x & 31) * 8;
  float k = 1.f;
  if (train) {
    if (keep_in) k = keep_in[(long)q * B + b];
    else k = (thresh == 0u || drop_keep(eff_seed(seed, seed_ptr), (uint64_t)q * B + b, thresh)) ? 1.f : 0.f;
    if (keep_out && c0 == 0) keep_out[(long)q * B + b] = k;
  }
  const float qs = train ? 2.f : 1.f;
  for (int c = c0; c < D; c += 256) {
    const VecT<T, 8> pv = *reinterpret_cast<const VecT<T, 8>*>(patch + ((long)b * P + q / qpp) * D + c);
    const float4 qa = *reinterpret_cast<const float4*>(query + (long)q * D + c), qb = *reinterpret_cast<const float4*>(query + (long)q * D + c + 4);
    const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
    VecT<T, 8> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = (T)(qs * qv[i] + k * (float)pv.v[i]);
    *reinterpret_cast<VecT<T, 8>*>(out + (long)row * D + c) = o;
  }
}

// backward: blocks [0, B*P): d_patch[b][p] = sum_{r < qpp} keep(p*qpp + r, b) * g[b][p*qpp + r]  (compute dtype);
//           blocks [B*P, B*P + Q): d_query[q] = (train ? 2 : 1) * sum_b g[b][q]  (f32).  The clip sum is a fixed-order tree: the block's
//           four thread groups own consecutive quarters of the clips and add them in clip order - eight loads in flight per thread, the
//           first version's one dependent load per clip took 48 us at B = 200 -, then the four partials are added in group order
template <typename T>
__global__ void spsedt_dec_in_bwd_kernel(const T* __restrict__ g, const float* __restrict__ keep, T* __restrict__ d_patch,
                                         float* __restrict__ d_query, int B, int Q, int P, int qpp, int D, int train) {
  __shared__ float part[3][1024 / 4];
  const int blk = blockIdx.x;
  const int c = threadIdx.x % D, grp = threadIdx.x / D;          // blockDim = 4 * D (D <= 256)
  if (blk < B * P) {
    if (!d_patch || grp) return;
    const int b = blk / P, p = blk - b * P;
    float acc = 0.f;
    for (int r = 0; r < qpp; ++r) {
      const int q = p * qpp + r;
      if (q >= Q) break;
      const float k = (train && keep) ? keep[(long)q * B + b] : 1.f;
      acc += k * (float)g[((long)b * Q + q) * D + c];
    }
    d_patch[(long)blk * D + c] = (T)acc;
    return;
  }
  const int q = blk - B * P;
  const int per = (B + 3) / 4, b0 = grp * per, b1 = min(B, b0 + per);
  float acc = 0.f;
  int b = b0;
  for (; b + 8 <= b1; b += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)g[((long)(b + i) * Q + q) * D + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  for (; b < b1; ++b) acc += (float)g[((long)b * Q + q) * D + c];
  if (grp) part[grp - 1][c] = acc;
  __syncthreads();
  if (!grp) d_query[(long)q * D + c] = (train ? 2.f : 1.f) * (((acc + part[0][c]) + part[1][c]) + part[2][c]);
}

// ------------------------------------------------------------------ backbone support
__global__ void bn_fold_kernel(const float* w, const float* b, const float* rm, const float* rv, float* scale,
                               float* bias, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sc = w[i] * rsqrtf(rv[i] + 1e-5f);
  scale[i] = sc;
  bias[i] = b[i] - rm[i] * sc;
}

template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, const float* bnscale,
                                 T* __restrict__ wf, T* __restrict__ wb) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long n = (long)Cout * Cin * taps;
  if (e >= n) return;
  int tap = (int)(e % taps);
  long r = e / taps;
  int ci = (int)(r % Cin), co = (int)(r / Cin);
  float v = w[e];
  if (wf) wf[((long)co * taps + tap) * Cin + ci] = (T)v;
  if (wb) wb[((long)ci * taps + tap) * Cout + co] = (T)(bnscale ? v * bnscale[co] : v);
}

template <typename T>
__global__ void stem_prep_kernel(const float* w0, const float* b0, const float* w1, T* wcat) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;  // 64*128
  if (e >= 64 * 128) return;
  int co = e >> 7, k = e & 127;
  float v = 0.f;
  int tap = k & 63;
  if (tap < 49) {
    const float* src = (k < 64) ? w0 : b0;
    for (int c = 0; c < 3; ++c) v += w1[(co * 3 + c) * 49 + tap] * src[c];
  }
  wcat[e] = (T)v;
}

// thread = one output pixel x 8 consecutive columns (one 16/32-byte store); 32-bit index arithmetic, the tap -> (kh, kw)
// split has a compile-time divisor.  Columns: [0,49) taps of x, [64,113) in-bounds indicators, the rest zero.
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int H, int W,
                                                          int Ho, int Wo) {
  // grid (ceil(Wo / 16), Ho, B).  The seven input rows of this output row go to LDS first (coalesced, zero-padded by 3 on
  // every side): the 4-byte tap gathers straight from global memory were what bounded the kernel (one scattered load
  // instruction per tap), the column stores are 16/32 bytes per thread.
  extern __shared__ float rows[];                     // [7][W + 6]
  const int P = W + 6;
  const int ho = blockIdx.y, b = blockIdx.z;
  const float* xb = x + (long)b * H * W;
  for (int i = threadIdx.x; i < 7 * P; i += 256) {
    const int kh = i / P, j = i - kh * P;
    const int hi = 2 * ho - 3 + kh, wi = j - 3;
    rows[i] = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? xb[hi * W + wi] : 0.f;
  }
  __syncthreads();
  const int wo = blockIdx.x * 16 + (threadIdx.x >> 4), chunk = threadIdx.x & 15;
  if (wo >= Wo) return;
  const int pix = (b * Ho + ho) * Wo + wo;
  const int k0 = chunk * 8;
  const bool ind = k0 >= 64;
  VecT<T, 8> out;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int tap = (k0 & 63) + e;
    float v = 0.f;
    if (tap < 49) {
      const int kh = tap / 7, kw = tap - kh * 7;
      const int hi = 2 * ho - 3 + kh, wi = 2 * wo - 3 + kw;
      if (ind) v = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? 1.f : 0.f;
      else v = rows[kh * P + 2 * wo + kw];
    }
    out.v[e] = (T)v;
  }
  *reinterpret_cast<VecT<T, 8>*>(col + (long)pix * 128 + k0) = out;
}

__global__ void stem_conv0_grad_kernel(const float* G, const float* w1, float* dw0, float* db0) {
  __shared__ float red[6][256];
  int t = threadIdx.x;
  float acc[6] = {0, 0, 0, 0, 0, 0};
  for (int e = t; e < 64 * 49; e += 256) {
    int co = e / 49, tap = e - co * 49;
    float gx = G[co * 128 + tap], gb = G[co * 128 + 64 + tap];
    for (int c = 0; c < 3; ++c) {
      float w = w1[(co * 3 + c) * 49 + tap];
      acc[c] += w * gx;
      acc[3 + c] += w * gb;
    }
  }
  for (int i = 0; i < 6; ++i) red[i][t] = acc[i];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s)
      for (int i = 0; i < 6; ++i) red[i][t] += red[i][t + s];
    __syncthreads();
  }
  if (t < 3) dw0[t] = red[t][0];
  else if (t < 6) db0[t - 3] = red[t][0];
}

// thread = one output pixel x VEC consecutive channels: 16-byte loads of the (up to) nine window taps, one 16-byte store of
// the maxima and VEC argmax bytes; 32-bit index arithmetic.  First maximum wins (scan order kh, kw), as torch does.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
  // grid (ceil(Wo * C/VEC / 256), Ho, B): one division (by the channel-chunk count) per thread instead of six
  const int CV = C / VEC;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Wo * CV) return;
  const int wo = e / CV, c = (e - wo * CV) * VEC;
  const int ho = blockIdx.y, b = blockIdx.z;
  const int pix = (b * Ho + ho) * Wo + wo;
  float best[VEC];
  uint8_t bi[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { best[v] = -INFINITY; bi[v] = 0; }
  bool first = true;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = 2 * ho - 1 + kh;
    if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = 2 * wo - 1 + kw;
      if ((unsigned)wi >= (unsigned)W) continue;
      const VecT<T, VEC> xv = *reinterpret_cast<const VecT<T, VEC>*>(x + (((long)b * H + hi) * W + wi) * C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float f = (float)xv.v[v];
        if (first || f > best[v]) { best[v] = f; bi[v] = (uint8_t)(kh * 3 + kw); }
      }
      first = false;
    }
  }
  VecT<T, VEC> yo;
  VecT<uint8_t, VEC> io;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { yo.v[v] = (T)best[v]; io.v[v] = bi[v]; }
  *reinterpret_cast<VecT<T, VEC>*>(y + (long)pix * C + c) = yo;
  *reinterpret_cast<VecT<uint8_t, VEC>*>(idx + (long)pix * C + c) = io;
}

// thread = one input pixel x VEC consecutive channels (16 bytes of dy / relu_src / dx at a time when C % VEC == 0)
template <typename T, int VEC>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, const T* __restrict__ relu_src,
                                   const T* __restrict__ y, T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
  // grid (ceil(W * C/VEC / 256), H, B): row and clip from the block index (the kernel was VALU-bound on its divisions)
  const int CV = C / VEC;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= W * CV) return;
  const int wi = e / CV, c = (e - wi * CV) * VEC;
  const int hi = blockIdx.y, b = blockIdx.z;
  float s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) s[v] = 0.f;
  for (int a = 0; a < 2; ++a) {
    const int ho = (hi + 1) / 2 - a;
    const int kh = hi - (2 * ho - 1);
    if (ho < 0 || ho >= Ho || kh < 0 || kh > 2) continue;
    for (int d = 0; d < 2; ++d) {
      const int wo = (wi + 1) / 2 - d;
      const int kw = wi - (2 * wo - 1);
      if (wo < 0 || wo >= Wo || kw < 0 || kw > 2) continue;
      const long o = (((long)b * Ho + ho) * Wo + wo) * C + c;
      T gv[VEC];
      uint8_t iv[VEC];
      *reinterpret_cast<typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(gv) =
          *reinterpret_cast<const typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(dy + o);
      *reinterpret_cast<typename std::conditional<VEC == 8, uint2, uint32_t>::type*>(iv) =
          *reinterpret_cast<const typename std::conditional<VEC == 8, uint2, uint32_t>::type*>(idx + o);
      if (y) {
        // ReLU mask of the pooled tensor's producer, taken from the pooled OUTPUT: the selected element is the window
        // maximum, so it passed the ReLU exactly when y > 0 (a quarter of the bytes of the full-resolution mask source)
        T yv[VEC];
        *reinterpret_cast<typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(yv) =
            *reinterpret_cast<const typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(y + o);
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (iv[v] == kh * 3 + kw && (float)yv[v] > 0.f) s[v] += (float)gv[v];
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (iv[v] == kh * 3 + kw) s[v] += (float)gv[v];
      }
    }
  }
  const long xo = (((long)b * H + hi) * W + wi) * C + c;
  T ov[VEC];
  if (relu_src) {
    T rv[VEC];
    *reinterpret_cast<typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(rv) =
        *reinterpret_cast<const typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(relu_src + xo);
#pragma unroll
    for (int v = 0; v < VEC; ++v) ov[v] = ((float)rv[v] > 0.f) ? (T)s[v] : (T)0.f;
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) ov[v] = (T)s[v];
  }
  *reinterpret_cast<typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(dx + xo) =
      *reinterpret_cast<typename std::conditional<sizeof(T) * VEC == 16, uint4, uint2>::type*>(ov);
}

template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int P, int C) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * C) return;
  int c = (int)(e % C), b = (int)(e / C);
  float s = 0.f;
  for (int q = 0; q < P; ++q) s += (float)x[((long)b * P + q) * C + c];
  out[e] = s / (float)P;
}

// the same mean with 8 channels per thread (16-byte loads of bf16) and four positions in flight: the SP-SEDT patch pool reads 262 MB
// (2000 patches x 32 positions x 2048 channels) - 125 us at one 2-byte load per thread and position, HBM-bound here.  Same summation
// order per channel (positions ascending), so the result is bit-identical to avgpool_kernel.
template <typename T>
__global__ __launch_bounds__(256) void avgpool8_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int P, int C8) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * C8) return;
  const int c8 = (int)(e % C8), b = (int)(e / C8);
  const VecT<T, 8>* src = reinterpret_cast<const VecT<T, 8>*>(x) + (long)b * P * C8 + c8;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  int q = 0;
  for (; q + 4 <= P; q += 4) {
    const VecT<T, 8> v0 = src[(long)q * C8], v1 = src[(long)(q + 1) * C8], v2 = src[(long)(q + 2) * C8], v3 = src[(long)(q + 3) * C8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = (((s[i] + (float)v0.v[i]) + (float)v1.v[i]) + (float)v2.v[i]) + (float)v3.v[i];
  }
  for (; q < P; ++q) {
    const VecT<T, 8> v0 = src[(long)q * C8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += (float)v0.v[i];
  }
  float* o = out + (long)b * C8 * 8 + c8 * 8;
  const float inv = (float)P;
  *reinterpret_cast<float4*>(o) = make_float4(s[0] / inv, s[1] / inv, s[2] / inv, s[3] / inv);
  *reinterpret_cast<float4*>(o + 4) = make_float4(s[4] / inv, s[5] / inv, s[6] / inv, s[7] / inv);
}

// thread = one pixel x 8 consecutive channels (4 sin/cos pairs): the cumulative count of unmasked rows is computed once per
// thread instead of once per channel
template <typename T>
__global__ __launch_bounds__(256) void posenc_kernel(const uint8_t* __restrict__ mask, T* __restrict__ pos, int B, int H, int W,
                                                     int D) {
  const int DV = D / 8;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)B * H * W * DV) return;
  const int c0 = (int)(e % DV) * 8;
  const int pix = (int)(e / DV);
  const int w = pix % W, r = pix / W;
  const int h = r % H, b = r / H;
  const uint8_t* m = mask + (long)b * H * W + w;
  float cum = 0.f, tot = 0.f;
  for (int i = 0; i < H; ++i) {
    const float nm = m[(long)i * W] ? 0.f : 1.f;
    tot += nm;
    if (i <= h) cum += nm;
  }
  const float y = cum / (tot + 1e-6f) * 6.283185307179586f;
  VecT<T, 8> out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float dim_t = powf(10000.f, (float)(c0 + 2 * q) / (float)D);
    const float a = y / dim_t;
    out.v[2 * q] = (T)sinf(a);
    out.v[2 * q + 1] = (T)cosf(a);
  }
  *reinterpret_cast<VecT<T, 8>*>(pos + (long)pix * D + c0) = out;
}

__global__ void mask_resize_kernel(const uint8_t* in, uint8_t* out, int B, int Hin, int Win, int Hout, int Wout) {
  long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * Hout * Wout) return;
  int w = (int)(e % Wout);
  long r = e / Wout;
  int h = (int)(r % Hout), b = (int)(r / Hout);
  float sh = (float)Hin / (float)Hout, sw = (float)Win / (float)Wout;
  int hi = min((int)floorf(h * sh), Hin - 1), wi = min((int)floorf(w * sw), Win - 1);
  out[e] = in[((long)b * Hin + hi) * Win + wi];
}

// ------------------------------------------------------------------ optimizer
__global__ void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// step_ptr / guard (multi-tensor optimizer only, may be null): a non-finite squared norm raises the guard word; the device-side
// Adam step count advances here unless the guard is up - the update kernels that follow skip themselves on a raised guard, so a
// NaN / inf loss or gradient never reaches parameters, moments, step count or the EMA teacher (the reference aborts before the
// backward on such a loss, engine.py:70-73 / 167-169; a captured step cannot, its host only polls the word now and then)
__global__ void sumsq_final_kernel(const float* part, int nparts, float* out, int accumulate, int32_t* step_ptr = nullptr,
                                   int32_t* guard = nullptr, uint32_t* seed_word = nullptr) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = red[0] + red[1] + red[2] + red[3];
    tot = accumulate ? out[0] + tot : tot;
    out[0] = tot;
    if (guard && !(tot <= 3.0e38f)) *guard = 1;
    if (step_ptr && !(guard && *guard)) step_ptr[0] += 1;
    if (seed_word) seed_word[0] += 1;          // the device-side dropout seed word: the NEXT step draws other masks
  }
}
static int sumsq_parts(long n) {
  long b = (n + 256 * 8 - 1) / (256 * 8);
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

__global__ void adamw_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, long n, const float* sumsq, float max_norm, float lr, float b1,
                                  float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  float gr = g[i] * coef;
  float pi = p[i] * (1.f - lr * wd);
  float mi = m[i] + (gr - m[i]) * (1.f - b1);
  float vi = v[i] * b2 + gr * gr * (1.f - b2);
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - (lr / bc1) * (mi / denom);
  m[i] = mi;
  v[i] = vi;
}

// ---- multi-tensor variants: one launch walks a device table of <= 65536-element chunks of many tensors
__global__ void multi_sumsq_kernel(const SedtChunk* __restrict__ table, float* __restrict__ partial) {
  __shared__ float red[4];
  const SedtChunk c = table[blockIdx.x];
  float s = 0.f;
  if (c.gflags & 1) {                                        // gradients in the bf16 flat buffer of the data-parallel step
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(c.g);
    int i0 = 0;
    if ((reinterpret_cast<uintptr_t>(gb) & 15) == 0) {
      const int n8 = c.n >> 3;
      for (int i = threadIdx.x; i < n8; i += blockDim.x) {
        const VecT<bf16_t, 8> a = reinterpret_cast<const VecT<bf16_t, 8>*>(gb)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = (float)a.v[e]; s += x * x; }
      }
      i0 = n8 << 3;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += blockDim.x) { const float a = (float)gb[i]; s += a * a; }
  } else {
    const float* g = reinterpret_cast<const float*>(c.g);
    int i0 = 0;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {          // 16-byte loads over the aligned body, four in flight per lane
      const int n4 = c.n >> 2;
      const float4* g4 = reinterpret_cast<const float4*>(g);
      const int bd = blockDim.x;
      int i = threadIdx.x;
      float s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (; i + 3 * bd < n4; i += 4 * bd) {
        const float4 a = g4[i], b = g4[i + bd], cc = g4[i + 2 * bd], d = g4[i + 3 * bd];
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
        s1 += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
        s2 += cc.x * cc.x + cc.y * cc.y + cc.z * cc.z + cc.w * cc.w;
        s3 += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      }
      for (; i < n4; i += bd) {
        const float4 a = g4[i];
        s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
      }
      s = (s + s1) + (s2 + s3);
      i0 = n4 << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += blockDim.x) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <bool NT>
__global__ void multi_adamw_kernel(const SedtChunk* __restrict__ table, const float* sumsq, float max_norm, float b1,
                                   float b2, float eps, const int32_t* __restrict__ step_ptr, const int32_t* __restrict__ guard) {
  if (guard && *guard) return;                // non-finite loss / gradient norm upstream: leave every tensor as it is
  const SedtChunk c = table[blockIdx.x];
  const float stepf = (float)step_ptr[0];
  const float bc1 = 1.f - powf(b1, stepf);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, stepf));
  float* p = reinterpret_cast<float*>(c.p);
  const float* g = reinterpret_cast<const float*>(c.g);
  float* m = reinterpret_cast<float*>(c.m);
  float* v = reinterpret_cast<float*>(c.v);
  float coef = 1.f;
  if (max_norm > 0.f) {
    coef = max_norm / (sqrtf(sumsq[0]) + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  const float decay = 1.f - c.lr * c.wd, slr = c.lr / bc1;
  auto upd = [&](float& pi, float gi, float& mi, float& vi) {
    const float gr = gi * coef;
    pi *= decay;
    mi = mi + (gr - mi) * (1.f - b1);
    vi = vi * b2 + gr * gr * (1.f - b2);
    pi -= slr * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  };
  if (c.gflags & 1) {                                        // bf16 flat gradients (data-parallel step with bf16 buckets)
    const bf16_t* gb = reinterpret_cast<const bf16_t*>(c.g);
    int i0 = 0;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(gb) & 7) == 0) {
      const int n4 = c.n >> 2;
      float4* p4 = reinterpret_cast<float4*>(p);
      float4* m4 = reinterpret_cast<float4*>(m);
      float4* v4 = reinterpret_cast<float4*>(v);
      for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const VecT<bf16_t, 4> gg = reinterpret_cast<const VecT<bf16_t, 4>*>(gb)[i];
        upd(pp.x, (float)gg.v[0], mm.x, vv.x);
        upd(pp.y, (float)gg.v[1], mm.y, vv.y);
        upd(pp.z, (float)gg.v[2], mm.z, vv.z);
        upd(pp.w, (float)gg.v[3], mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
      }
      i0 = n4 << 2;
    }
    for (int i = i0 + threadIdx.x; i < c.n; i += blockDim.x) {
      float pi = p[i], mi = m[i], vi = v[i];
      upd(pi, (float)gb[i], mi, vi);
      p[i] = pi; m[i] = mi; v[i] = vi;
    }
    return;
  }
  int i0 = 0;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
        reinterpret_cast<uintptr_t>(v)) & 15) == 0) {          // 16-byte accesses over the aligned body
    const int n4 = c.n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    // the gradient is dead after this pass and the two moments are next touched one step later: non-temporal accesses keep the
    // 390 MB of them from evicting the freshly written weights (which the next forward's packing kernel reads) from the MALL
    typedef float __attribute__((ext_vector_type(4))) f4v;
    const f4v* g4n = reinterpret_cast<const f4v*>(g);
    f4v* m4n = reinterpret_cast<f4v*>(m);
    f4v* v4n = reinterpret_cast<f4v*>(v);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      float4 pp = p4[i];
      f4v mm = NT ? __builtin_nontemporal_load(m4n + i) : m4n[i];
      f4v vv = NT ? __builtin_nontemporal_load(v4n + i) : v4n[i];
      const f4v gg = NT ? __builtin_nontemporal_load(g4n + i) : g4n[i];
      float m0 = mm.x, m1 = mm.y, m2 = mm.z, m3 = mm.w, v0 = vv.x, v1 = vv.y, v2 = vv.z, v3 = vv.w;
      upd(pp.x, gg.x, m0, v0);
      upd(pp.y, gg.y, m1, v1);
      upd(pp.z, gg.z, m2, v2);
      upd(pp.w, gg.w, m3, v3);
      p4[i] = pp;
      const f4v mo = {m0, m1, m2, m3}, vo = {v0, v1, v2, v3};
      if (NT) { __builtin_nontemporal_store(mo, m4n + i); __builtin_nontemporal_store(vo, v4n + i); }
      else { m4n[i] = mo; v4n[i] = vo; }
    }
    i0 = n4 << 2;
  }
  for (int i = i0 + threadIdx.x; i < c.n; i += blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// mean-teacher weights (reference utilities/utils.py:62-67): shadow = (1 - decay) * p + decay * shadow, chunk.m = shadow
__global__ void multi_ema_kernel(const SedtChunk* __restrict__ table, float decay, const int32_t* __restrict__ guard) {
  if (guard && *guard) return;
  const SedtChunk c = table[blockIdx.x];
  const float* p = reinterpret_cast<const float*>(c.p);
  float* sh = reinterpret_cast<float*>(c.m);
  const float om = 1.f - decay;
  int i0 = 0;
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(sh)) & 15) == 0) {
    const int n4 = c.n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float4* s4 = reinterpret_cast<float4*>(sh);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      const float4 a = p4[i];
      float4 b = s4[i];
      b.x = om * a.x + decay * b.x; b.y = om * a.y + decay * b.y; b.z = om * a.z + decay * b.z; b.w = om * a.w + decay * b.w;
      s4[i] = b;
    }
    i0 = n4 << 2;
  }
  for (int i = i0 + threadIdx.x; i < c.n; i += blockDim.x) sh[i] = om * p[i] + decay * sh[i];
}

// ---- all FrozenBN folds / all weight packs of a model in one launch each (device job tables)
__global__ void multi_bn_fold_kernel(const SedtBnJob* __restrict__ jobs) {
  const SedtBnJob j = jobs[blockIdx.x];
  for (int i = threadIdx.x; i < j.n; i += blockDim.x) {
    float sc = j.w[i] * rsqrtf(j.rv[i] + 1e-5f);
    j.scale[i] = sc;
    j.bias[i] = j.b[i] - j.rm[i] * sc;
  }
}

// one workgroup = one TS(co) x TS(ci) x taps tile of one weight tensor (TS = 64 for 1x1 kernels / linears, 32 for 3x3),
// staged through LDS so that the f32 source rows ([co][ci][tap], tap fastest) are read in 128/256-byte runs and both packed
// layouts are written in 64/128-byte runs.  PackPlan (packing.py) counts blocks with the same rule.
template <typename T, int TAPS>      // TAPS = 0: run-time tap count
__device__ __forceinline__ void pack_tile(const SedtPackJob& j, int tb, float* tile) {
  const int taps = TAPS ? TAPS : j.taps;          // compile-time in the common cases: no integer divisions in the loops
  const int sh = taps == 1 ? 6 : 5, TS = 1 << sh;
  const int tci = (j.Cin + TS - 1) >> sh;
  T* wf = reinterpret_cast<T*>(j.wf);
  T* wb = reinterpret_cast<T*>(j.wb);
  const int rowlen = TS * taps, pitch = rowlen + 1;
  const int co0 = (tb / tci) << sh, ci0 = (tb % tci) << sh;
  const int nco = min(TS, j.Cout - co0), nci = min(TS, j.Cin - ci0);
  const int valid = nci * taps;
  for (int idx = threadIdx.x; idx < TS * rowlen; idx += 256) {
    const int co = idx / rowlen, off = idx - co * rowlen;
    float v = 0.f;
    if (co < nco && off < valid) v = j.w[((long)(co0 + co) * j.Cin + ci0) * taps + off];
    tile[co * pitch + off] = v;
  }
  __syncthreads();
  if (wf) {
    for (int idx = threadIdx.x; idx < TS * rowlen; idx += 256) {     // (co, tap, ci) with ci fastest
      const int ci = idx & (TS - 1), r = idx >> sh;
      const int co = r / taps, tap = r - co * taps;
      if (co < nco && ci < nci)
        wf[((long)(co0 + co) * taps + tap) * j.Cin + ci0 + ci] = (T)tile[co * pitch + ci * taps + tap];
    }
  }
  if (wb) {
    for (int idx = threadIdx.x; idx < TS * rowlen; idx += 256) {     // (ci, tap, co) with co fastest
      const int co = idx & (TS - 1), r = idx >> sh;
      const int ci = r / taps, tap = r - ci * taps;
      if (co < nco && ci < nci) {
        float v = tile[co * pitch + ci * taps + tap];
        if (j.bnscale) v *= j.bnscale[co0 + co];
        wb[((long)(ci0 + ci) * taps + tap) * j.Cout + co0 + co] = (T)v;
      }
    }
  }
}

// the same tile with 16-byte global accesses (float4 source loads, 8 packed elements per store): the scalar form above spends
// its time issuing 2-byte stores.  Needs Cin % 8 == 0, Cout % 8 == 0 and 16-byte aligned tensors.
template <typename T, int TAPS, int SH>
__device__ __forceinline__ void pack_tile_vec(const SedtPackJob& j, int tb, float* tile) {
  constexpr int TS = 1 << SH, ROWLEN = TS * TAPS, PITCH = ROWLEN + 1, C8 = TS / 8;
  const int tci = (j.Cin + TS - 1) >> SH;
  T* wf = reinterpret_cast<T*>(j.wf);
  T* wb = reinterpret_cast<T*>(j.wb);
  const int co0 = (tb / tci) << SH, ci0 = (tb % tci) << SH;
  const int nco = min(TS, j.Cout - co0), nci = min(TS, j.Cin - ci0);
  const int valid = nci * TAPS;                       // multiple of 4 (nci % 8 == 0)
  for (int idx = threadIdx.x; idx < TS * (ROWLEN / 4); idx += 256) {
    const int co = idx / (ROWLEN / 4), off = (idx - co * (ROWLEN / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (co < nco && off < valid) v = *reinterpret_cast<const float4*>(j.w + ((long)(co0 + co) * j.Cin + ci0) * TAPS + off);
    float* d = tile + co * PITCH + off;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  if (wf) {
    for (int idx = threadIdx.x; idx < TS * TAPS * C8; idx += 256) {      // (co, tap, 8 consecutive ci)
      const int c8 = idx % C8, r = idx / C8;
      const int co = r / TAPS, tap = r - co * TAPS;
      if (co < nco && c8 * 8 < nci) {
        VecT<T, 8> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = (T)tile[co * PITCH + (c8 * 8 + e) * TAPS + tap];
        *reinterpret_cast<VecT<T, 8>*>(wf + ((long)(co0 + co) * TAPS + tap) * j.Cin + ci0 + c8 * 8) = o;
      }
    }
  }
  if (wb) {
    for (int idx = threadIdx.x; idx < TS * TAPS * C8; idx += 256) {      // (ci, tap, 8 consecutive co)
      const int c8 = idx % C8, r = idx / C8;
      const int ci = r / TAPS, tap = r - ci * TAPS;
      if (ci < nci && c8 * 8 < nco) {
        VecT<T, 8> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = tile[(c8 * 8 + e) * PITCH + ci * TAPS + tap];
          if (j.bnscale) v *= j.bnscale[co0 + c8 * 8 + e];
          o.v[e] = (T)v;
        }
        *reinterpret_cast<VecT<T, 8>*>(wb + ((long)(ci0 + ci) * TAPS + tap) * j.Cout + co0 + c8 * 8) = o;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void multi_pack_kernel(const SedtPackJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32 * (32 * 9 + 1)];         // >= 64 * 65 as well
  // find the tensor this block belongs to: jobs[].e0 holds the first block index of each tensor (ascending)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].e0 <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SedtPackJob j = jobs[lo];
  const int tb = (int)((long)blockIdx.x - j.e0);
  const bool vec = ((j.Cin | j.Cout) & 7) == 0 &&
                   ((reinterpret_cast<uintptr_t>(j.w) | reinterpret_cast<uintptr_t>(j.wf) | reinterpret_cast<uintptr_t>(j.wb)) & 15) == 0;
  if (j.taps == 1) {
    if (vec) pack_tile_vec<T, 1, 6>(j, tb, tile);
    else pack_tile<T, 1>(j, tb, tile);
  } else if (j.taps == 9) {
    if (vec) pack_tile_vec<T, 9, 5>(j, tb, tile);
    else pack_tile<T, 9>(j, tb, tile);
  } else {
    pack_tile<T, 0>(j, tb, tile);
  }
}

// chunk.p[i] (=|+=) chunk.g[i]: ACC adds into the flat buffer (gradient accumulation over micro-batches, reference
// engine.py:76, 174), BF stores the flat buffer as bf16 (half the all-reduce bytes of the data-parallel step; the source gradients
// and the optimizer's arithmetic stay f32)
template <bool ACC, bool BF>
__global__ void multi_gather_kernel(const SedtChunk* __restrict__ table) {
  // grid (chunks, GATHER_SPLIT): a chunk (<= 65536 elements) is shared by GATHER_SPLIT workgroups - one workgroup per chunk walked
  // 64 dependent 16-byte loads per lane, and the small segments of the data-parallel schedule (19 chunks for the 4.8 MB tail)
  // took as long as the large ones
  const SedtChunk c = table[blockIdx.x];
  const float* src = reinterpret_cast<const float*>(c.g);
  const int per = (((c.n + gridDim.y - 1) / gridDim.y) + 7) & ~7;          // elements per workgroup, a multiple of 8
  const int e0 = blockIdx.y * per, e1 = min(c.n, e0 + per);
  if (e0 >= e1) return;
  if (BF) {
    bf16_t* dst = reinterpret_cast<bf16_t*>(c.p);
    int i0 = e0;
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {      // 8 elements per lane and access
      const int n8 = (e1 - e0) >> 3;
      for (int i = threadIdx.x; i < n8; i += blockDim.x) {
        const float4 a = reinterpret_cast<const float4*>(src + e0)[2 * i], b2 = reinterpret_cast<const float4*>(src + e0)[2 * i + 1];
        float v[8] = {a.x, a.y, a.z, a.w, b2.x, b2.y, b2.z, b2.w};
        VecT<bf16_t, 8> o;
        if (ACC) {
          const VecT<bf16_t, 8> old = reinterpret_cast<const VecT<bf16_t, 8>*>(dst + e0)[i];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)old.v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = (bf16_t)v[e];
        reinterpret_cast<VecT<bf16_t, 8>*>(dst + e0)[i] = o;
      }
      i0 = e0 + (n8 << 3);
    }
    for (int i = i0 + threadIdx.x; i < e1; i += blockDim.x) dst[i] = (bf16_t)(ACC ? (float)dst[i] + src[i] : src[i]);
    return;
  }
  float* dst = reinterpret_cast<float*>(c.p);
  int i0 = e0;
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0) {
    const int n4 = (e1 - e0) >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst + e0);
    const float4* s4 = reinterpret_cast<const float4*>(src + e0);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
      float4 v = s4[i];
      if (ACC) { const float4 o = d4[i]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      d4[i] = v;
    }
    i0 = e0 + (n4 << 2);
  }
  for (int i = i0 + threadIdx.x; i < e1; i += blockDim.x) dst[i] = ACC ? dst[i] + src[i] : src[i];
}

}  // namespace sedt

using namespace sedt;

extern "C" const char* sedt_last_error(void) { return g_err; }
extern "C" int sedt_version(void) { return 1; }

#define BY_DTYPE(dtype, CALL_F32, CALL_BF16)                   \
  if ((dtype) == SEDT_F32) { CALL_F32; }                       \
  else if ((dtype) == SEDT_BF16) { CALL_BF16; }                \
  else { set_error("unsupported dtype %d", (int)(dtype)); return 1; }



extern "C" int sedt_wgrad_reduce_bias(const float* slab, int splitk, int R, int taps, int Ci, const float* rowscale,
                                      float* out, const float* colsum_slab, float* bias_out, void* stream) {
  SEDT_REQUIRE(slab && out && splitk >= 1, "wgrad_reduce: bad args");
  SEDT_REQUIRE((colsum_slab == nullptr) == (bias_out == nullptr), "wgrad_reduce: colsum_slab and bias_out go together");
  SedtReduceJob j = {};
  j.slab = slab; j.out = out; j.rowscale = rowscale; j.colsum_slab = colsum_slab; j.bias_out = bias_out;
  j.splitk = splitk; j.R = R; j.taps = taps; j.Ci = Ci;
  return sedt_multi_wgrad_reduce(&j, 1, nullptr, stream);      // one job of the grouped kernel (vector / LDS-transposing paths)
}

extern "C" int sedt_wgrad_reduce(const float* slab, int splitk, int R, int taps, int Ci, const float* rowscale,
                                 float* out, void* stream) {
  return sedt_wgrad_reduce_bias(slab, splitk, R, taps, Ci, rowscale, out, nullptr, nullptr, stream);
}

extern "C" size_t sedt_colsum_scratch(int rows, int cols) { return (size_t)colsum_chunks(rows) * cols * sizeof(float); }

extern "C" int sedt_colsum(const void* in, int64_t ld, int rows, int cols, int in_f32, int dtype, float* out,
                           float* scratch, size_t scratch_bytes, void* stream) {
  int chunks = colsum_chunks(rows);
  SEDT_REQUIRE(in && out, "colsum: null pointer");
  SEDT_REQUIRE(chunks == 1 || (scratch && scratch_bytes >= sedt_colsum_scratch(rows, cols)), "colsum: scratch too small");
  int rpc = (rows + chunks - 1) / chunks;
  dim3 grid((cols + 63) / 64, chunks);
  float* stage1 = chunks == 1 ? out : scratch;
  if (in_f32 || dtype == SEDT_F32)
    hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, S(stream), (const float*)in, (long)ld, rows, cols, rpc, stage1);
  else
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, S(stream), (const bf16_t*)in, (long)ld, rows, cols, rpc, stage1);
  if (chunks > 1)
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((cols + 63) / 64, 1), dim3(256), 0, S(stream), (const float*)scratch,
                       (long)cols, chunks, cols, chunks, out);
  return check_launch("colsum");
}

extern "C" int sedt_dropout_grad(const void* in, int64_t ldi, void* out, int64_t ldo, int rows, int cols, float p,
                                 uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream) {
  SEDT_REQUIRE(p >= 0.f && p < 1.f, "dropout_grad: p out of range");
  long n = (long)rows * cols;
  uint32_t th = drop_threshold(p);
  float ik = 1.f / (1.f - p);
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(dropout_grad_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), (const float*)in, (long)ldi,
                              (float*)out, (long)ldo, rows, cols, th, ik, seed, seed_ptr),
           hipLaunchKernelGGL(dropout_grad_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), (const bf16_t*)in,
                              (long)ldi, (bf16_t*)out, (long)ldo, rows, cols, th, ik, seed, seed_ptr));
  return check_launch("dropout_grad");
}

extern "C" int sedt_add(const void* a, const void* b, void* out, int rows, int cols, int b_mod, int dtype, void* stream) {
  long n = (long)rows * cols;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(add_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), (const float*)a, (const float*)b,
                              (float*)out, rows, cols, b_mod),
           hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), (const bf16_t*)a,
                              (const bf16_t*)b, (bf16_t*)out, rows, cols, b_mod));
  return check_launch("add");
}

extern "C" int sedt_add_n(const void* const* srcs, int n, void* out, int64_t numel, int dtype, void* stream) {
  SEDT_REQUIRE(srcs && out && n >= 1 && n <= 8 && numel % 8 == 0, "add_n: 1..8 sources, element count a multiple of 8 (got %d, %ld)", n,
               (long)numel);
  AddN a;
  for (int i = 0; i < 8; ++i) a.p[i] = i < n ? srcs[i] : nullptr;
  a.n = n;
  for (int i = 0; i < n; ++i) SEDT_REQUIRE(a.p[i] != nullptr, "add_n: null source %d", i);
  const long n8 = numel / 8;
  BY_DTYPE(dtype, hipLaunchKernelGGL(add_n_kernel<float>, dim3(nblk(n8)), dim3(256), 0, S(stream), a, (float*)out, n8),
           hipLaunchKernelGGL(add_n_kernel<bf16_t>, dim3(nblk(n8)), dim3(256), 0, S(stream), a, (bf16_t*)out, n8));
  return check_launch("add_n");
}

extern "C" int sedt_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, void* stream) {
  dim3 g(nblk(n)), b(256);
  if (in_dtype == SEDT_F32 && out_dtype == SEDT_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, S(stream), (const float*)in, (bf16_t*)out, (long)n);
  else if (in_dtype == SEDT_BF16 && out_dtype == SEDT_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, S(stream), (const bf16_t*)in, (float*)out, (long)n);
  else if (in_dtype == SEDT_F32 && out_dtype == SEDT_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, S(stream), (const float*)in, (float*)out, (long)n);
  else if (in_dtype == SEDT_BF16 && out_dtype == SEDT_BF16)
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, S(stream), (const bf16_t*)in, (bf16_t*)out, (long)n);
  else { set_error("cast: bad dtypes %d -> %d", in_dtype, out_dtype); return 1; }
  return check_launch("cast");
}

extern "C" int sedt_copy2d(const SedtCopyJob* jobs, int njobs, void* stream) {
  SEDT_REQUIRE(jobs && njobs >= 1 && njobs <= COPY_MAXJ, "copy2d: 1..%d jobs", COPY_MAXJ);
  CopyJobs a;
  a.n = njobs;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    const SedtCopyJob& j = jobs[i];
    SEDT_REQUIRE(j.src && j.dst && j.outer >= 1 && j.inner >= 4 && (j.inner & 3) == 0 && (j.src_stride & 3) == 0 && (j.dst_stride & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 3) == 0,
                 "copy2d: job %d: outer %d inner %d (bytes, multiples of 4; 4-byte aligned pointers and strides)", i, j.outer, j.inner);
    a.j[i] = j;
    a.j[i].blk0 = blk;
    blk += (int)(((long)j.outer * (j.inner / 4) + 255) / 256);
  }
  hipLaunchKernelGGL(copy2d_kernel, dim3(blk), dim3(256), 0, S(stream), a);
  return check_launch("copy2d");
}

extern "C" int sedt_spsedt_dec_in(const void* patch, const float* query, const float* keep_in, float* keep_out, void* out, int B, int Q,
                                  int P, int qpp, int D, int train, float ratio, uint32_t seed, const uint32_t* seed_ptr, int dtype,
                                  void* stream) {
  SEDT_REQUIRE(patch && query && out && B > 0 && Q > 0 && P > 0 && qpp > 0 && (Q + qpp - 1) / qpp <= P, "spsedt_dec_in: bad arguments");
  SEDT_REQUIRE(D % 8 == 0 && ((uintptr_t)patch & 15) == 0 && ((uintptr_t)query & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "spsedt_dec_in: D must be a multiple of 8 and the pointers 16-byte aligned");
  const uint32_t th = (train && !keep_in && ratio > 0.f) ? drop_threshold(ratio < 1.f ? ratio : 0.99999f) : 0u;
  const int rows = B * Q;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(spsedt_dec_in_kernel<float>, dim3((rows + 7) / 8), dim3(256), 0, S(stream), (const float*)patch, query, keep_in,
                              keep_out, (float*)out, B, Q, P, qpp, D, train, th, seed, seed_ptr),
           hipLaunchKernelGGL(spsedt_dec_in_kernel<bf16_t>, dim3((rows + 7) / 8), dim3(256), 0, S(stream), (const bf16_t*)patch, query,
                              keep_in, keep_out, (bf16_t*)out, B, Q, P, qpp, D, train, th, seed, seed_ptr));
  return check_launch("spsedt_dec_in");
}

extern "C" int sedt_spsedt_dec_in_bwd(const void* g, const float* keep, void* d_patch, float* d_query, int B, int Q, int P, int qpp, int D,
                                      int train, int dtype, void* stream) {
  SEDT_REQUIRE(g && d_query && B > 0 && Q > 0 && P > 0 && qpp > 0 && D > 0 && D <= 256 && D % 64 == 0, "spsedt_dec_in_bwd: D a multiple of 64, <= 256");
  const int thr = 4 * D;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(spsedt_dec_in_bwd_kernel<float>, dim3(B * P + Q), dim3(thr), 0, S(stream), (const float*)g, keep, (float*)d_patch,
                              d_query, B, Q, P, qpp, D, train),
           hipLaunchKernelGGL(spsedt_dec_in_bwd_kernel<bf16_t>, dim3(B * P + Q), dim3(thr), 0, S(stream), (const bf16_t*)g, keep,
                              (bf16_t*)d_patch, d_query, B, Q, P, qpp, D, train));
  return check_launch("spsedt_dec_in_bwd");
}

extern "C" int sedt_gelu_fwd(const void* h, void* a, int64_t n, float p, uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream) {
  SEDT_REQUIRE(p >= 0.f && p < 1.f, "gelu_fwd: p out of range");
  SEDT_REQUIRE(n % 8 == 0 && ((uintptr_t)h & 15) == 0 && ((uintptr_t)a & 15) == 0, "gelu_fwd: numel must be a multiple of 8, pointers 16-byte aligned");
  const uint32_t th = p > 0.f ? drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  const long n8 = n / 8;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(gelu_fwd_kernel<float>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const float*)h, (float*)a, n8, th, ik, seed, seed_ptr),
           hipLaunchKernelGGL(gelu_fwd_kernel<bf16_t>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const bf16_t*)h, (bf16_t*)a, n8, th, ik, seed, seed_ptr));
  return check_launch("gelu_fwd");
}

extern "C" int sedt_gelu_bwd(const void* g, const void* h, void* out, int64_t n, float p, uint32_t seed, const uint32_t* seed_ptr, int dtype,
                             void* stream) {
  SEDT_REQUIRE(p >= 0.f && p < 1.f, "gelu_bwd: p out of range");
  SEDT_REQUIRE(n % 8 == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)h & 15) == 0 && ((uintptr_t)out & 15) == 0,
               "gelu_bwd: numel must be a multiple of 8, pointers 16-byte aligned");
  const uint32_t th = p > 0.f ? drop_threshold(p) : 0u;
  const float ik = 1.f / (1.f - p);
  const long n8 = n / 8;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const float*)g, (const float*)h, (float*)out, n8, th, ik, seed, seed_ptr),
           hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const bf16_t*)g, (const bf16_t*)h, (bf16_t*)out, n8, th, ik, seed, seed_ptr));
  return check_launch("gelu_bwd");
}

extern "C" int sedt_relu_mask(const void* g, const void* y, void* out, int64_t n, int dtype, void* stream) {
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(relu_mask_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), (const float*)g, (const float*)y,
                              (float*)out, (long)n),
           hipLaunchKernelGGL(relu_mask_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), (const bf16_t*)g,
                              (const bf16_t*)y, (bf16_t*)out, (long)n));
  return check_launch("relu_mask");
}

extern "C" int sedt_sigmoid_grad(const float* g, const float* s, float* out, int64_t n, void* stream) {
  hipLaunchKernelGGL(sigmoid_grad_kernel, dim3(nblk(n)), dim3(256), 0, S(stream), g, s, out, (long)n);
  return check_launch("sigmoid_grad");
}

extern "C" int sedt_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* bias,
                            int n, void* stream) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(nblk(n)), dim3(256), 0, S(stream), w, b, rm, rv, scale, bias, n);
  return check_launch("bn_fold");
}

extern "C" int sedt_pack_conv(const float* w, int Cout, int Cin, int taps, const float* bnscale, void* wf, void* wb,
                              int dtype, void* stream) {
  long n = (long)Cout * Cin * taps;
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(pack_conv_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), w, Cout, Cin, taps, bnscale,
                              (float*)wf, (float*)wb),
           hipLaunchKernelGGL(pack_conv_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), w, Cout, Cin, taps, bnscale,
                              (bf16_t*)wf, (bf16_t*)wb));
  return check_launch("pack_conv");
}

extern "C" int sedt_stem_prep(const float* w0, const float* b0, const float* w1, void* wcat, int dtype, void* stream) {
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(stem_prep_kernel<float>, dim3(32), dim3(256), 0, S(stream), w0, b0, w1, (float*)wcat),
           hipLaunchKernelGGL(stem_prep_kernel<bf16_t>, dim3(32), dim3(256), 0, S(stream), w0, b0, w1, (bf16_t*)wcat));
  return check_launch("stem_prep");
}

extern "C" int sedt_stem_im2col(const float* x, void* col, int B, int H, int W, int dtype, void* stream) {
  int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long npix = (long)B * Ho * Wo;
  SEDT_REQUIRE(npix * 128 < (1L << 31) * 16 && npix < (1L << 31) - 16, "stem_im2col: too many pixels for 32-bit indexing");
  SEDT_REQUIRE(Ho <= 65535 && B <= 65535, "stem_im2col: Ho / B exceed the grid limits");
  SEDT_REQUIRE(W <= 2042, "stem_im2col: W = %d too wide for the row staging buffer", W);
  const dim3 grid((unsigned)((Wo + 15) / 16), (unsigned)Ho, (unsigned)B);
  const size_t lds = (size_t)7 * (W + 6) * sizeof(float);
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(stem_im2col_kernel<float>, grid, dim3(256), lds, S(stream), x, (float*)col, B, H, W, Ho, Wo),
           hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, grid, dim3(256), lds, S(stream), x, (bf16_t*)col, B, H, W, Ho, Wo));
  return check_launch("stem_im2col");
}

extern "C" int sedt_stem_conv0_grad(const float* G, const float* w1, float* dw0, float* db0, void* stream) {
  hipLaunchKernelGGL(stem_conv0_grad_kernel, dim3(1), dim3(256), 0, S(stream), G, w1, dw0, db0);
  return check_launch("stem_conv0_grad");
}

extern "C" int sedt_maxpool_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int dtype, void* stream) {
  int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  auto al = [](const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
  SEDT_REQUIRE((long)B * H * W < (1L << 31) && H <= 65535 && B <= 65535, "maxpool_fwd: too many pixels for 32-bit indexing / the grid");
  if (dtype == SEDT_BF16) {
    SEDT_REQUIRE(C % 8 == 0 && al(x, 16) && al(y, 16) && al(idx, 8), "maxpool_fwd: needs C %% 8 == 0 and 16-byte aligned tensors");
    const int bs = std::min(256, (Wo * (C / 8) + 63) / 64 * 64);        // a row of 16 pixels x 8 chunks fills only two waves
    const dim3 grid(nblk((long)Wo * (C / 8), bs), (unsigned)Ho, (unsigned)B);
    hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t, 8>), grid, dim3(bs), 0, S(stream), (const bf16_t*)x, (bf16_t*)y, idx,
                       B, H, W, C, Ho, Wo);
  } else if (dtype == SEDT_F32) {
    SEDT_REQUIRE(C % 4 == 0 && al(x, 16) && al(y, 16) && al(idx, 4), "maxpool_fwd: needs C %% 4 == 0 and 16-byte aligned tensors");
    const int bs = std::min(256, (Wo * (C / 4) + 63) / 64 * 64);        // a row of 16 pixels x 8 chunks fills only two waves
    const dim3 grid(nblk((long)Wo * (C / 4), bs), (unsigned)Ho, (unsigned)B);
    hipLaunchKernelGGL((maxpool_fwd_kernel<float, 4>), grid, dim3(bs), 0, S(stream), (const float*)x, (float*)y, idx, B,
                       H, W, C, Ho, Wo);
  } else {
    set_error("maxpool_fwd: unsupported dtype %d", dtype);
    return 1;
  }
  return check_launch("maxpool_fwd");
}

extern "C" int sedt_maxpool_bwd_y(const void* dy, const uint8_t* idx, const void* relu_src, const void* y, void* dx, int B, int H,
                                  int W, int C, int dtype, void* stream);

extern "C" int sedt_maxpool_bwd(const void* dy, const uint8_t* idx, const void* relu_src, void* dx, int B, int H, int W,
                                int C, int dtype, void* stream) {
  return sedt_maxpool_bwd_y(dy, idx, relu_src, nullptr, dx, B, H, W, C, dtype, stream);
}

extern "C" int sedt_maxpool_bwd_y(const void* dy, const uint8_t* idx, const void* relu_src, const void* y, void* dx, int B, int H,
                                  int W, int C, int dtype, void* stream) {
  int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  SEDT_REQUIRE((long)B * H * W < (1L << 31) && H <= 65535 && B <= 65535, "maxpool_bwd: too many pixels for 32-bit indexing / the grid");
  if (dtype == SEDT_BF16) {
    SEDT_REQUIRE(C % 8 == 0 && al(dy) && al(dx) && (!relu_src || al(relu_src)) && (reinterpret_cast<uintptr_t>(idx) & 7) == 0,
                 "maxpool_bwd: needs C %% 8 == 0 and 16-byte aligned tensors");
    const int bs = std::min(256, (W * (C / 8) + 63) / 64 * 64);
    const dim3 grid(nblk((long)W * (C / 8), bs), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t, 8>), grid, dim3(bs), 0, S(stream), (const bf16_t*)dy, idx,
                       (const bf16_t*)relu_src, (const bf16_t*)y, (bf16_t*)dx, B, H, W, C, Ho, Wo);
  } else if (dtype == SEDT_F32) {
    SEDT_REQUIRE(C % 4 == 0 && al(dy) && al(dx) && (!relu_src || al(relu_src)) && (reinterpret_cast<uintptr_t>(idx) & 3) == 0,
                 "maxpool_bwd: needs C %% 4 == 0 and 16-byte aligned tensors");
    const int bs = std::min(256, (W * (C / 4) + 63) / 64 * 64);
    const dim3 grid(nblk((long)W * (C / 4), bs), (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL((maxpool_bwd_kernel<float, 4>), grid, dim3(bs), 0, S(stream), (const float*)dy, idx,
                       (const float*)relu_src, (const float*)y, (float*)dx, B, H, W, C, Ho, Wo);
  } else {
    set_error("unsupported dtype %d", dtype);
    return 1;
  }
  return check_launch("maxpool_bwd");
}

extern "C" int sedt_avgpool(const void* x, float* out, int B, int P, int C, int dtype, void* stream) {
  long n = (long)B * C;
  if ((C & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 31) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const long n8 = n / 8;
    BY_DTYPE(dtype,
             hipLaunchKernelGGL(avgpool8_kernel<float>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const float*)x, out, B, P, C / 8),
             hipLaunchKernelGGL(avgpool8_kernel<bf16_t>, dim3(nblk(n8)), dim3(256), 0, S(stream), (const bf16_t*)x, out, B, P, C / 8));
    return check_launch("avgpool");
  }
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(avgpool_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), (const float*)x, out, B, P, C),
           hipLaunchKernelGGL(avgpool_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), (const bf16_t*)x, out, B, P, C));
  return check_launch("avgpool");
}

extern "C" int sedt_posenc(const uint8_t* mask, void* pos, int B, int H, int W, int D, int dtype, void* stream) {
  SEDT_REQUIRE(D % 8 == 0 && (reinterpret_cast<uintptr_t>(pos) & 31) == 0 && (long)B * H * W < (1L << 31),
               "posenc: needs D %% 8 == 0, a 32-byte aligned output and < 2^31 pixels");
  long n = (long)B * H * W * (D / 8);
  BY_DTYPE(dtype,
           hipLaunchKernelGGL(posenc_kernel<float>, dim3(nblk(n)), dim3(256), 0, S(stream), mask, (float*)pos, B, H, W, D),
           hipLaunchKernelGGL(posenc_kernel<bf16_t>, dim3(nblk(n)), dim3(256), 0, S(stream), mask, (bf16_t*)pos, B, H, W, D));
  return check_launch("posenc");
}

extern "C" int sedt_mask_resize(const uint8_t* in, uint8_t* out, int B, int Hin, int Win, int Hout, int Wout, void* stream) {
  long n = (long)B * Hout * Wout;
  hipLaunchKernelGGL(mask_resize_kernel, dim3(nblk(n)), dim3(256), 0, S(stream), in, out, B, Hin, Win, Hout, Wout);
  return check_launch("mask_resize");
}

extern "C" size_t sedt_sumsq_scratch(int64_t n) { return (size_t)sumsq_parts(n) * sizeof(float); }

extern "C" int sedt_sumsq(const float* g, int64_t n, float* sumsq, float* scratch, size_t scratch_bytes, int accumulate,
                          void* stream) {
  int parts = sumsq_parts(n);
  SEDT_REQUIRE(scratch && scratch_bytes >= (size_t)parts * sizeof(float), "sumsq: scratch too small");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(parts), dim3(256), 0, S(stream), g, (long)n, scratch);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, S(stream), scratch, parts, sumsq, accumulate);
  return check_launch("sumsq");
}

extern "C" int sedt_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                               float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream) {
  SEDT_REQUIRE(step >= 1, "adamw: step must be >= 1");
  float bc1 = 1.f - powf(beta1, (float)step);
  float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_clip_kernel, dim3(nblk(n)), dim3(256), 0, S(stream), p, g, m, v, (long)n, sumsq, max_norm, lr, beta1,
                     beta2, eps, weight_decay, bc1, bc2s);
  return check_launch("adamw_clip");
}

extern "C" int sedt_multi_sumsq(const SedtChunk* table, int nchunks, float* partial, float* sumsq, int32_t* step_ptr,
                                int32_t* guard, uint32_t* seed_word, void* stream) {
  SEDT_REQUIRE(table && partial && sumsq && nchunks > 0, "multi_sumsq: bad arguments");
  hipLaunchKernelGGL(multi_sumsq_kernel, dim3(nchunks), dim3(256), 0, S(stream), table, partial);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, S(stream), partial, nchunks, sumsq, 0, step_ptr, guard, seed_word);
  return check_launch("multi_sumsq");
}

extern "C" int sedt_multi_adamw(const SedtChunk* table, int nchunks, const float* sumsq, float max_norm, float beta1,
                                float beta2, float eps, const int32_t* step_ptr, const int32_t* guard, void* stream) {
  SEDT_REQUIRE(table && nchunks > 0 && step_ptr, "multi_adamw: bad arguments");
  SEDT_REQUIRE(max_norm <= 0.f || sumsq, "multi_adamw: clipping needs sumsq");
  static int nt = -1;
  if (nt < 0) {
    const char* e = sedt::dev_getenv("SEDT_ADAMW_NT");          // developer A/B switch (default on)
    nt = (e && e[0] == '0') ? 0 : 1;
  }
  if (nt) hipLaunchKernelGGL(multi_adamw_kernel<true>, dim3(nchunks), dim3(256), 0, S(stream), table, sumsq, max_norm, beta1, beta2, eps,
                             step_ptr, guard);
  else hipLaunchKernelGGL(multi_adamw_kernel<false>, dim3(nchunks), dim3(256), 0, S(stream), table, sumsq, max_norm, beta1, beta2, eps,
                          step_ptr, guard);
  return check_launch("multi_adamw");
}

extern "C" int sedt_multi_ema(const SedtChunk* table, int nchunks, float decay, const int32_t* guard, void* stream) {
  SEDT_REQUIRE(table && nchunks > 0 && decay >= 0.f && decay <= 1.f, "multi_ema: bad arguments");
  hipLaunchKernelGGL(multi_ema_kernel, dim3(nchunks), dim3(256), 0, S(stream), table, decay, guard);
  return check_launch("multi_ema");
}

extern "C" int sedt_multi_gather(const SedtChunk* table, int nchunks, int mode, void* stream) {
  SEDT_REQUIRE(table && nchunks > 0, "multi_gather: bad arguments");
  SEDT_REQUIRE(mode >= 0 && mode <= 3, "multi_gather: mode %d (bit 0: accumulate, bit 1: bf16 destination)", mode);
  switch (mode) {
    case 0: hipLaunchKernelGGL((multi_gather_kernel<false, false>), dim3(nchunks, 8), dim3(256), 0, S(stream), table); break;
    case 1: hipLaunchKernelGGL((multi_gather_kernel<true, false>), dim3(nchunks, 8), dim3(256), 0, S(stream), table); break;
    case 2: hipLaunchKernelGGL((multi_gather_kernel<false, true>), dim3(nchunks, 8), dim3(256), 0, S(stream), table); break;
    default: hipLaunchKernelGGL((multi_gather_kernel<true, true>), dim3(nchunks, 8), dim3(256), 0, S(stream), table); break;
  }
  return check_launch("multi_gather");
}

extern "C" int sedt_multi_bn_fold(const SedtBnJob* jobs, int njobs, void* stream) {
  SEDT_REQUIRE(jobs && njobs > 0, "multi_bn_fold: bad arguments");
  hipLaunchKernelGGL(multi_bn_fold_kernel, dim3(njobs), dim3(256), 0, S(stream), jobs);
  return check_launch("multi_bn_fold");
}

extern "C" int sedt_multi_pack(const SedtPackJob* jobs, int njobs, int nblocks, int dtype, void* stream) {
  SEDT_REQUIRE(jobs && njobs > 0 && nblocks > 0, "multi_pack: bad arguments");
  BY_DTYPE(dtype, hipLaunchKernelGGL(multi_pack_kernel<float>, dim3(nblocks), dim3(256), 0, S(stream), jobs, njobs),
           hipLaunchKernelGGL(multi_pack_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, S(stream), jobs, njobs));
  return check_launch("multi_pack");
}

extern "C" int sedt_multi_wgrad_reduce(const SedtReduceJob* jobs, int njobs, const SedtPrefetch* pf, void* stream) {
  SEDT_REQUIRE(jobs && njobs > 0 && njobs <= SEDT_MAX_REDUCE_JOBS, "multi_wgrad_reduce: 1..%d jobs", SEDT_MAX_REDUCE_JOBS);
  ReduceJobs a;
  a.n = njobs;
  for (int r = 0; r < 3; ++r) {                                  // what the NEXT launch streams: touched by this one
    a.pf[r] = pf ? (const uint32_t*)pf->ptr[r] : nullptr;
    a.pf_lines[r] = (pf && pf->ptr[r]) ? (int)(pf->bytes[r] / 128) : 0;
  }
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    a.j[i] = jobs[i];
    const bool sums_only = (long)jobs[i].R * jobs[i].taps * jobs[i].Ci == 0;      // column sums only (LayerNorm gamma/beta)
    SEDT_REQUIRE((sums_only ? jobs[i].colsum_slab != nullptr : (jobs[i].slab && jobs[i].out)) && jobs[i].splitk >= 1,
                 "multi_wgrad_reduce: bad job %d", i);
    SEDT_REQUIRE((jobs[i].colsum_slab == nullptr) == (jobs[i].bias_out == nullptr), "multi_wgrad_reduce: job %d colsum/bias", i);
    a.j[i].blk0 = blk;
    int nb = reduce_job_blocks(jobs[i]);
    if (jobs[i].colsum_slab) nb = std::max(nb, (jobs[i].R + 63) / 64);   // the bias sums ride on the first blocks
    blk += nb;
  }
  hipLaunchKernelGGL(multi_wgrad_reduce_kernel, dim3(blk), dim3(256), 0, S(stream), a);
  return check_launch("multi_wgrad_reduce");
}
