// bneck_common.h - what the fused-Bottleneck kernels (bneck.hip, bneck3.hip) share: the slab GEMM over weight chunks in registers and the
// packed-math epilogue helpers.
#pragma once
#include "slab.h"

namespace sedt {

using slab::u32x4;

// NSW pixel slabs x one output tile over NKS k-steps (chunks of 8 fragments alternating cur / alt as in slab::wave_gemm)
template <int NSW, int NKS, class Next>
__device__ __forceinline__ void gemm_slabs(f32x16 (&acc)[NSW], const bf16_t* xs, int xp, const u32x4* __restrict__ W, int lane, u32x4 (&cur)[8],
                                           u32x4 (&alt)[8], Next next) {
  constexpr int NCH = NKS / 8;
  static_assert(NKS % 8 == 0 && NCH % 2 == 0, "an even number of chunks");
  const bf16_t* xrow = xs + (lane & 31) * xp + 8 * (lane >> 5);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4(&src)[8] = (c & 1) ? alt : cur;
    u32x4(&dst)[8] = (c & 1) ? cur : alt;
    if (c + 1 < NCH) slab::load_chunk<1>(dst, W, 0, (c + 1) * 8, lane);
    else next(dst);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      bf16x8 xb[NSW];
#pragma unroll
      for (int s = 0; s < NSW; ++s) xb[s] = *reinterpret_cast<const bf16x8*>(xrow + s * 32 * xp + (c * 8 + u) * 16);
#pragma unroll
      for (int s = 0; s < NSW; ++s)
        acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[u]), xb[s], acc[s], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// four fragments (a 4-k-step unit of one tile) into dst[o .. o + 3]
template <int NR>
__device__ __forceinline__ void load4(u32x4 (&dst)[NR], int o, const u32x4* __restrict__ W, int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) dst[o + u] = W[u * 64 + lane];
}

// the value again, opaque to the optimiser: keeps the strip loop's address arithmetic INSIDE the loop (hoisted, the invariant piece
// offsets of every copy loop below filled the register file and spilled: 155 VGPRs to scratch in the first build)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---- epilogue arithmetic, two elements per instruction where the ISA has it (v_pk_fma_f32 / v_pk_add_f32 / v_cvt_pk_bf16_f32 /
// v_pk_min_u16): with one workgroup per CU the epilogues of a strip (50 k elements) are VALU time nothing else hides
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 widen2(unsigned w) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
__device__ __forceinline__ f32x2 relu2(f32x2 v) { return f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}; }
// 0xffff in the half whose mask bit (bits pos, pos + 1 of m) is set
__device__ __forceinline__ unsigned keep2(unsigned m, int pos) {
  const unsigned m0 = (unsigned)__builtin_amdgcn_sbfe((int)m, pos, 1), m1 = (unsigned)__builtin_amdgcn_sbfe((int)m, pos + 1, 1);
  return (m0 & 0xffffu) | (m1 & 0xffff0000u);
}
// sign nibble of four NON-NEGATIVE bf16 values (two packed words): bit e <-> element e != 0
__device__ __forceinline__ unsigned nibble4(unsigned w0, unsigned w1) {
  unsigned r0, r1;                                        // (the vector builtin expands to compares and selects per half)
  const unsigned one = 0x00010001u;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r0) : "v"(w0), "v"(one));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(r1) : "v"(w1), "v"(one));
  const unsigned t = r0 | (r1 << 2);                      // b0 | b2 << 2 | b1 << 16 | b3 << 18
  return (t | (t >> 15)) & 0xfu;
}

}  // namespace sedt
