// common.h - shared device helpers for the gfx950 SEDT kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/sedt_hip.h"

namespace sedt {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ---- tuning switches.  Every host-side dispatch rule of the library (tile per shape, ring depth, split-K target, grid caps, which
// kernel generation takes a problem) has its MEASURED default written at its site; the environment can override a default only in a
// developer build (hipcc -DSEDT_DEV: `SEDT_DEV_BUILD=1 python -m sound_event_detection_transformer_amd._build` writes
// build/dev/libsedt_hip_dev.so for the sweeps under tools/).  The product library never reads the environment: dev_getenv() is a
// constant nullptr there, so its behaviour cannot depend on variables leaked into a training job.
inline const char* dev_getenv(const char* name) {
#ifdef SEDT_DEV
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// ---- describe mode (sedt_igemm_describe): while `describe.on`, every GEMM launcher writes the name of the kernel instance it WOULD
// launch for the problem (as a profiler prints it) and returns 0 without launching - bench.py uses it to attach algorithmic flops and
// bytes to the per-kernel times of the measured step (roofline.families).
struct Describe {
  bool on;
  char name[96];
};
extern thread_local Describe describe;
#define SEDT_DESCRIBE(...)                                                   \
  do {                                                                       \
    if (sedt::describe.on) {                                                 \
      snprintf(sedt::describe.name, sizeof(sedt::describe.name), __VA_ARGS__); \
      return 0;                                                              \
    }                                                                        \
  } while (0)

// ---- error plumbing (thread-local message, integer status across the C ABI)
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define SEDT_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      sedt::set_error(__VA_ARGS__);    \
      return 1;                        \
    }                                  \
  } while (0)

// ---- scalar load/store in the compute dtype
template <typename T>
__device__ __forceinline__ float ldf(const T* p, long i) { return (float)p[i]; }
template <typename T>
__device__ __forceinline__ void stf(T* p, long i, float v) { p[i] = (T)v; }

// ---- counter-based hash RNG for dropout: one 32-bit draw per (seed, element index).
// Any kernel can regenerate the keep-mask of element idx without stored masks.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t rng32(uint32_t seed, uint64_t idx) {
  uint32_t hi = (uint32_t)(idx >> 32), lo = (uint32_t)idx;
  return mix32(lo ^ mix32(seed ^ (hi * 0x9E3779B9U) ^ 0x85ebca6bU));
}
// keep-decision of element idx: 16 bits of the hash of (seed, idx/2) - two neighbouring elements share one hash, which
// halves the VALU cost in the GEMM epilogues (8 consecutive columns per lane).  Resolution of p: 1/65536.
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
  double t = (double)p * 65536.0 + 0.5;
  return t >= 65535.0 ? 0xffffU : (uint32_t)t;
}
__device__ __forceinline__ bool drop_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  const uint32_t h = rng32(seed, idx >> 1);
  const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return bits >= thresh;
}
// The same decisions as drop_keep with the seed-only half of the hash taken out of the inner loops (the attention kernels
// draw one decision per probability and were VALU-bound on this hash: 100 VALU per MFMA by PMC).
// drop_inner: the (seed, high index word) part; valid for every element whose hash index idx >> 1 has that high word.
__device__ __forceinline__ uint32_t drop_inner(uint32_t seed, uint32_t hi) { return mix32(seed ^ (hi * 0x9E3779B9U) ^ 0x85ebca6bU); }
__device__ __forceinline__ bool drop_keep_in(uint32_t inner, uint32_t inner_hi, uint32_t seed, uint64_t idx, uint32_t thresh) {
  const uint64_t hidx = idx >> 1;
  const uint32_t hi = (uint32_t)(hidx >> 32);
  const uint32_t in = hi == inner_hi ? inner : drop_inner(seed, hi);          // (never taken below 2^33 elements)
  const uint32_t h = mix32((uint32_t)hidx ^ in);
  const uint32_t bits = (idx & 1) ? (h >> 16) : (h & 0xffffu);
  return bits >= thresh;
}
// four consecutive elements idx0 .. idx0+3: bit e of the result = keep(idx0 + e).  Two hashes when idx0 is even.
__device__ __forceinline__ uint32_t drop_keep4(uint32_t inner, uint32_t inner_hi, uint32_t seed, uint64_t idx0, uint32_t thresh) {
  if (idx0 & 1) {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) m |= (drop_keep_in(inner, inner_hi, seed, idx0 + e, thresh) ? 1u : 0u) << e;
    return m;
  }
  const uint64_t h0 = idx0 >> 1, h1 = h0 + 1;
  const uint32_t in0 = (uint32_t)(h0 >> 32) == inner_hi ? inner : drop_inner(seed, (uint32_t)(h0 >> 32));
  const uint32_t in1 = (uint32_t)(h1 >> 32) == inner_hi ? inner : drop_inner(seed, (uint32_t)(h1 >> 32));
  const uint32_t ha = mix32((uint32_t)h0 ^ in0), hb = mix32((uint32_t)h1 ^ in1);
  return ((ha & 0xffffu) >= thresh ? 1u : 0u) | ((ha >> 16) >= thresh ? 2u : 0u) | ((hb & 0xffffu) >= thresh ? 4u : 0u) |
         ((hb >> 16) >= thresh ? 8u : 0u);
}
// keep-decisions of 8 consecutive elements base .. base+7 (base % 8 == 0) as a bit mask - the same decisions as drop_keep, with
// the part of the hash that depends only on (seed, high index word) computed once instead of four times
__device__ __forceinline__ uint32_t drop_keep8(uint32_t seed, uint64_t base, uint32_t thresh) {
  const uint64_t h0 = base >> 1;                       // hash index of elements base, base+1; +1, +2, +3 for the next pairs
  const uint32_t hi = (uint32_t)(h0 >> 32), lo = (uint32_t)h0;      // lo % 4 == 0: lo + 3 does not carry into hi
  const uint32_t inner = mix32(seed ^ (hi * 0x9E3779B9U) ^ 0x85ebca6bU);
  uint32_t m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t h = mix32((lo + q) ^ inner);
    m |= ((h & 0xffffu) >= thresh ? 1u : 0u) << (2 * q);
    m |= ((h >> 16) >= thresh ? 1u : 0u) << (2 * q + 1);
  }
  return m;
}
__device__ __forceinline__ uint32_t eff_seed(uint32_t seed, const uint32_t* seed_ptr) {
  return seed + (seed_ptr ? *seed_ptr : 0u);
}

// ---- N elements moved as one 8/16/32-byte access
template <typename T, int N>
struct alignas(sizeof(T) * N) VecT { T v[N]; };

// ---- wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace sedt
