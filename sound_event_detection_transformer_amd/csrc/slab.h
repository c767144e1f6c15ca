// slab.h - the "x-stationary" GEMM building block of the transformer-side kernels (enc_slab.hip, dec_slab.hip).
//
// At B = 64 every GEMM of the transformer has M = 8192 (encoder) or M = 704 (decoder) rows: one or two 64x128 tiles per CU in the
// LDS-DMA GEMM family, whose tiles move BOTH operands through LDS at ~30 B/clk/CU, launch by launch.  Here a workgroup owns a SLAB of
// 32 token rows for a whole chain of layers: the activations stay in LDS (bf16 [32][K] tiles) as the B operand of
// v_mfma_f32_32x32x16_bf16 (lane <-> token), and only the WEIGHTS move - straight from L2 into registers as A operands
// (lane <-> output feature), in a fragment-major packing (sedt_pack_frag) in which one wave instruction reads one fully contiguous
// 1 KB block.  Measured (tools/probes/wstream.hip, 256 workgroups streaming the same matrix): 36-44 B/clk/CU = 88-106 GB/s per CU,
// 23-27 TB/s over the chip, against ~30 B/clk/CU for the LDS-DMA ring - and nothing but W moves.
//
// Fragment-major packing of W [N][K] (N outputs, K inputs; N % 32 == 0, K % 16 == 0): block (ft = n / 32, ks = k / 16) of 1 KB at
// byte ((ft * (K / 16) + ks) * 1024); inside, lane l = 32 * ((k % 16) / 8) + n % 32 owns the 8 consecutive k of its half = 16 bytes.
// That is exactly the A operand of the 32x32x16 MFMA (row = lane % 32, k-slots 8 * (lane / 32) ..+7).
#pragma once
#include "common.h"

namespace sedt {
namespace slab {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

constexpr int SR = 32;             // token rows of a slab
constexpr int XP = 256 + 8;        // element pitch of a [32][256] bf16 LDS tile: 528 B rows (16-byte fragment reads conflict-free)

// C-layout row (= output feature inside its tile of 32) of accumulator register r in lane half hf
__device__ __forceinline__ int crow(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// A weight CHUNK = 8 fragments (8 KB per wave): T output-feature tiles x U = 8 / T k-steps of 16.
//   W : the fragment-major weight, pointing at the block of (first tile, first k-step); tstride = 16-byte units between the same
//       k-step of consecutive tiles (= 64 * K / 16)
template <int T>
__device__ __forceinline__ void load_chunk(u32x4 (&dst)[8], const u32x4* __restrict__ W, long tstride, int ks0, int lane) {
  constexpr int U = 8 / T;
  const u32x4* wl = W + lane;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int u = 0; u < U; ++u) dst[t * U + u] = wl[t * tstride + (long)(ks0 + u) * 64];
}

// the seed-only half of the dropout hash for element indices below 2^33 (every tensor of this path): hoisted out of the epilogues'
// inner loops by the compiler once it is a pure function of a kernel-constant seed (drop_keep4 recomputes it for larger indices)
__device__ __forceinline__ uint32_t inner0(uint32_t seed) { return drop_inner(seed, 0u); }

// "everything issued above stays above": keeps a block of loads ahead of the work that hides their latency
__device__ __forceinline__ void issue_fence() { __builtin_amdgcn_sched_barrier(0); }

struct NoNext {
  __device__ __forceinline__ void operator()(u32x4 (&)[8]) const {}
};

// T output-feature tiles x the slab's 32 tokens over NKS k-steps of 16; xs = the activation tile in LDS (element pitch xp, the k-steps
// start at its column 0).  On entry `cur` holds chunk 0 (load_chunk<T>(cur, W, tstride, 0, lane), issued by the caller as early as
// it likes - weights do not depend on activations); chunk c + 1 is in flight while chunk c is multiplied, and while the LAST chunk
// is multiplied `next(cur)` has the first chunk of the FOLLOWING GEMM in flight (the number of chunks is even, so it lands in
// `cur` again): a wave's weight stream never drains across the barriers and epilogues between the GEMMs of a chain - each
// restart exposed a full L2 round trip (measured on the FFN pair: 48 -> 2x us).
template <int T, int NKS, class Next>
__device__ __forceinline__ void wave_gemm(f32x16 (&acc)[T], const bf16_t* xs, int xp, const u32x4* __restrict__ W, long tstride,
                                          int lane, u32x4 (&cur)[8], u32x4 (&alt)[8], Next next) {
  constexpr int U = 8 / T, NCH = NKS / U;
  static_assert(NKS % U == 0 && NCH % 2 == 0, "an even number of chunks");
  const bf16_t* xrow = xs + (lane & 31) * xp + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4(&src)[8] = (c & 1) ? alt : cur;
    u32x4(&dst)[8] = (c & 1) ? cur : alt;
    if (c + 1 < NCH) load_chunk<T>(dst, W, tstride, (c + 1) * U, lane);
    else next(dst);
    // the machine scheduler otherwise sinks every load to just above its MFMA (fewer live registers): global_load -> s_waitcnt
    // vmcnt(0) -> v_mfma, one exposed L2 round trip per fragment (seen in the ISA of the first version: the FFN pair at 18 B/clk)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 xb[U];                                              // the chunk's activation fragments: all LDS reads ahead of the MFMAs
#pragma unroll
    for (int u = 0; u < U; ++u) xb[u] = *reinterpret_cast<const bf16x8*>(xrow + (c * U + u) * 16);
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[t * U + u]), xb[u], acc[t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the 4 x 4 per-feature values (bias, ...) a lane needs for the epilogue of output tile `tile`: feature tile * 32 + 8 * g4 + 4 * hf + e.
// Issued BEFORE the GEMM whose epilogue uses them: loads return in order, so a load issued in the epilogue would wait behind the
// weight chunks already prefetched for the next GEMM - the stream would drain once per epilogue.
__device__ __forceinline__ void load_feat4(float4 (&dst)[4], const float* __restrict__ p, int tile, int hf) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) dst[g4] = *reinterpret_cast<const float4*>(p + tile * 32 + 8 * g4 + 4 * hf);
}

// a SMALL GEMM: one output tile over 16 k-steps = two chunks, BOTH preloaded (cur = chunk 0, alt = chunk 1: load_chunk<1>(cur, W, 0, 0,
// lane); load_chunk<1>(alt, W, 0, 8, lane)).  As soon as a chunk is multiplied its registers take a chunk of the FOLLOWING GEMM
// (next0(cur) / next1(alt)): between two GEMMs of a chain of 256 x 256 projections (the decoder layer: LayerNorms, attention
// cores, barriers in between) the whole next weight tile is in flight, so the GEMM itself costs its 16 MFMAs and no memory round trip.
template <class Next0, class Next1>
__device__ __forceinline__ void wave_gemm_small(f32x16 (&acc)[1], const bf16_t* xs, int xp, int lane, u32x4 (&cur)[8], u32x4 (&alt)[8],
                                                Next0 next0, Next1 next1) {
  const bf16_t* xrow = xs + (lane & 31) * xp + 8 * (lane >> 5);
  bf16x8 xb[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) xb[u] = *reinterpret_cast<const bf16x8*>(xrow + u * 16);
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[u]), xb[u], acc[0], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  next0(cur);
#pragma unroll
  for (int u = 0; u < 8; ++u) xb[u] = *reinterpret_cast<const bf16x8*>(xrow + (8 + u) * 16);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, alt[u]), xb[u], acc[0], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  next1(alt);
  __builtin_amdgcn_sched_barrier(0);
}

// ONE output-feature tile x FOUR row slabs (a 128-row block in LDS, rows s * 32 + (lane & 31) of slab s) over NKS k-steps: every
// weight fragment feeds four MFMAs, so a 128-row workgroup moves a quarter of the weight bytes per row of the 32-row scheme.  Chunks
// of 8 k-steps alternate between `cur` and `alt` as in wave_gemm; `next(cur)` issues the first chunk of the following GEMM.
template <int NKS, class Next>
__device__ __forceinline__ void wave_gemm_r4(f32x16 (&acc)[4], const bf16_t* xs, int xp, const u32x4* __restrict__ W, int lane,
                                             u32x4 (&cur)[8], u32x4 (&alt)[8], Next next) {
  constexpr int NCH = NKS / 8;
  static_assert(NKS % 8 == 0 && NCH % 2 == 0, "an even number of chunks");
  const bf16_t* xrow = xs + (lane & 31) * xp + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4(&src)[8] = (c & 1) ? alt : cur;
    u32x4(&dst)[8] = (c & 1) ? cur : alt;
    if (c + 1 < NCH) load_chunk<1>(dst, W, 0, (c + 1) * 8, lane);
    else next(dst);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bf16x8 xb[4];
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) xb[sl] = *reinterpret_cast<const bf16x8*>(xrow + sl * 32 * xp + (c * 8 + u) * 16);
#pragma unroll
      for (int sl = 0; sl < 4; ++sl)
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[u]), xb[sl], acc[sl], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// 16 bytes stored WRITE-THROUGH (sc1): the line leaves this XCD's L2 at once, so publishing it needs no release fence (a
// release = buffer_wbl2 writes back EVERYTHING dirty in the XCD's L2 - with 32 workgroups per XCD each leaving 128-256 KB of fresh
// partial sums and activations that made the first version of the split FFN take 58 us instead of 14)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
__device__ __forceinline__ void store_sc1(float* p, f32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// In-launch reduction over the workgroups that share an output block (the split-K seam of cdna_hip_programming.md section 5 /
// Guideline 16, counter form with write-through payload stores): call after the workgroup's partial result has been stored with
// store_sc1.  Returns true in the workgroup that arrived LAST (all `nparts` partials are then visible to it: one agent-scope
// acquire drops its CU's stale L1 lines); that workgroup also re-arms the counter.  `flag` = one LDS word.  The counter must be
// zero before the first launch that uses it.
__device__ __forceinline__ bool arrive_last(unsigned* cnt, unsigned nparts, unsigned* flag, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // every storing wave drains its (write-through) stores
  __syncthreads();
  if (tid == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = ticket == nparts - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-armed for the next launch
    }
    *flag = last ? 1u : 0u;
  }
  __syncthreads();
  return *flag != 0u;
}

template <int T>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[T]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
}

// LayerNorm of the slab rows [4 * wave, 4 * wave + 4) over 256 features: a lane owns 4 consecutive features of a row.
// src(row) -> pointer to the row's 256 bf16 values (global or LDS); rows >= nvalid give zeros.  f(row, lane, y[4], mean, rstd).
template <class Src, class Out>
__device__ __forceinline__ void slab_layernorm(int wave, int lane, int nvalid, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, Src src, Out out) {
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4), b4 = *reinterpret_cast<const float4*>(beta + lane * 4);
  const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
  VecT<bf16_t, 4> xin[4];                                     // all four rows on their way before the first reduction
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 4 + i;
    xin[i] = row < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(src(row) + lane * 4) : VecT<bf16_t, 4>{};
  }
  issue_fence();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 4 + i;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (float)xin[i].v[e];
    const float mu = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256.f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = v[e] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.f / 256.f) + 1e-5f);
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = row < nvalid ? (v[e] - mu) * rs * g[e] + bt[e] : 0.f;
    out(row, y, mu, rs);
  }
}

// cooperative copy of an LDS tile [rows][cols] (bf16, element pitch lp) to global rows of stride ld (16 bytes per thread and step)
__device__ __forceinline__ void tile_to_global(const bf16_t* tile, int lp, bf16_t* __restrict__ dst, long ld, int rows, int cols, int tid,
                                               int nthr) {
  const int cpr = cols >> 3;
  for (int u = tid; u < rows * cpr; u += nthr) {
    const int r = u / cpr, c = (u - r * cpr) * 8;
    *reinterpret_cast<uint4*>(dst + (long)r * ld + c) = *reinterpret_cast<const uint4*>(tile + r * lp + c);
  }
}

}  // namespace slab
}  // namespace sedt
