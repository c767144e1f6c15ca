// attn_frag.h - LDS image / MFMA fragment helpers of the bf16 attention kernels (head dim 32), shared by attn_mfma.hip and the
// slab kernels (enc_slab.hip, dec_slab.hip).  A head's Q / K / V tile is a bf16 [L][32] LDS image with 64-byte rows.
#pragma once
#include "common.h"

namespace sedt {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

constexpr int AD = 32;          // head dim
constexpr int AROW = AD * 2;    // bytes per image row
constexpr int AMAXT = 8;        // key / query tiles of 32 -> L <= 256

__device__ __forceinline__ int crow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }   // C-layout row of register r

// stage rows [0, L) of a [*, ld] bf16 matrix (32 columns from `src`) into a zero-padded [Lpad][32] LDS image
__device__ __forceinline__ void stage_image(unsigned char* img, const bf16_t* src, long ld, int L, int Lpad, int tid, int nthr) {
  for (int u = tid; u < Lpad * 4; u += nthr) {     // 4 x 16-byte chunks per row
    const int r = u >> 2, c = u & 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < L) v = *reinterpret_cast<const uint4*>(src + (long)r * ld + c * 8);
    *reinterpret_cast<uint4*>(img + r * AROW + c * 16) = v;
  }
}

// A/B fragment whose 32 "rows" are image rows row0..row0+31 and whose k-slots are the 16 dims [16*ks, +16)
__device__ __forceinline__ bf16x8 frag_rows(const unsigned char* img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + (row0 + (lane & 31)) * AROW + (ks * 16 + 8 * (lane >> 5)) * 2);
}

// fragment whose 32 "rows"/"cols" are the 32 dims and whose k-slots are image rows: slot s of half h <-> row
// base16 + (s&3) + 8*(s>>2) + 4*h  (the order in which a C-layout accumulator holds its rows).  Two transposing reads.
__device__ __forceinline__ bf16x8 frag_cols_tr(const unsigned char* img, int base16, int lane) {
  const int h = lane >> 5, dgrp = (lane >> 4) & 1, s16 = lane & 15;
  const unsigned char* p = img + (base16 + 4 * h + (s16 >> 2)) * AROW + (16 * dgrp + 4 * (s16 & 3)) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 8 * AROW));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);   // register concatenation, no ALU
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
  return o;
}

// additive key bias of one (batch) row into LDS: 0 for live keys, -inf for padded keys (kpm) and for the zero-padded tail
__device__ __forceinline__ void stage_key_bias(float* kb, const uint8_t* kpm_row, int Lk, int LkP, int tid, int nthr) {
  for (int i = tid; i < LkP; i += nthr) kb[i] = (i >= Lk || (kpm_row && kpm_row[i])) ? -INFINITY : 0.f;
}

}  // namespace sedt
