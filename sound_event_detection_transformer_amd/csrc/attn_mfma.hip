// attn_mfma.hip - bf16 multi-head attention forward/backward on MFMA for gfx950 (head dim 32, L <= 256).
//
// One workgroup per (batch, head); Q/K/V (and dO) tiles of that head are staged once in LDS as bf16 [L][32] images
// (64-B rows).  All four contractions run on v_mfma_f32_32x32x16_bf16:
//
//   forward, wave = 32 queries:   S^T = K Q^T   (lane <-> query, registers <-> keys: softmax statistics are in-lane
//                                                reductions + one cross-half shuffle)
//                                 O  += P V      P straight from the S^T accumulators (k-slot s of half h <-> key
//                                                16u + (s&3) + 8(s>>2) + 4h); the matching V^T fragment is read with
//                                                ds_read_b64_tr_b16 (same k-slot order), so no cross-lane movement.
//   backward pass A (32 queries): S^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta)  ->  dQ += dS K   (K^T via tr reads)
//   backward pass B (32 keys):    S = Q K^T, dP = dO V^T (lane <-> key, registers <-> queries)
//                                 dV^T += dO^T Pd,  dK^T += Q^T dS            (dO^T, Q^T via tr reads)
//
// Probabilities never touch HBM: backward recomputes them from the saved log-sum-exp; the dropout mask is the counter
// hash of ((b*H+h)*Lq + q)*Lk + k, identical in forward and backward (and to the f32 reference kernels).
#include <stdlib.h>
#include "common.h"
#include "attn_frag.h"

namespace sedt {

// ============================================================================================ forward
template <int NT, bool AMASK>   // key tiles; additive attention mask present
__global__ __launch_bounds__(256, 2) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ k,
                                                               long ldk, const bf16_t* __restrict__ v, long ldv,
                                                               bf16_t* __restrict__ o, long ldo, float* __restrict__ lse,
                                                               const uint8_t* __restrict__ kpm, const float* __restrict__ amask,
                                                               int H, int Lq, int Lk, float scale, uint32_t thresh, float inv_keep,
                                                               uint32_t seed, const uint32_t* seed_ptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int LkP = NT * 32, LqP = (Lq + 31) & ~31;
  unsigned char* Ki = smem;
  unsigned char* Vi = Ki + LkP * AROW;
  unsigned char* Qi = Vi + LkP * AROW;
  float* Kb = reinterpret_cast<float*>(Qi + LqP * AROW);
  float* Rs = Kb + LkP;                                // [waves][32] 1 / row-sum strips
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  stage_image(Ki, k + (long)b * Lk * ldk + h * AD, ldk, Lk, LkP, tid, nthr);
  stage_image(Vi, v + (long)b * Lk * ldv + h * AD, ldv, Lk, LkP, tid, nthr);
  stage_image(Qi, q + (long)b * Lq * ldq + h * AD, ldq, Lq, LqP, tid, nthr);
  stage_key_bias(Kb, kpm ? kpm + (long)b * Lk : nullptr, Lk, LkP, tid, nthr);
  __syncthreads();
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const int hf = lane >> 5;
  for (int q0 = wave * 32; q0 < Lq; q0 += nwave * 32) {
    const int qi = q0 + (lane & 31);                   // this lane's query
    const bf16x8 qf0 = frag_rows(Qi, q0, 0, lane), qf1 = frag_rows(Qi, q0, 1, lane);
    // one score tile (32 keys x this wave's 32 queries): S^T = K Q^T, scaled and masked.  Register r of half hf <-> key
    // 8*(r>>2) + 4*hf + (r&3): four consecutive keys per register group
    auto score_tile = [&](int kt, f32x16& st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ki, kt * 32, 0, lane), qf0, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ki, kt * 32, 1, lane), qf1, st, 0, 0, 0);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int key0 = kt * 32 + 8 * g4 + 4 * hf;
        const float4 kb = *reinterpret_cast<const float4*>(Kb + key0);
        const float kbv[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sc = st[4 * g4 + e] * scale + kbv[e];
          if (AMASK) { if (qi < Lq && key0 + e < Lk) sc += amask[(long)qi * Lk + key0 + e]; }
          st[4 * g4 + e] = sc;
        }
      }
    };
    // pass 1: the row maximum only (no exponentials: v_exp_f32 is quarter rate, and the first version of this kernel paid
    // it twice per score - once for the row sum here, once for the probabilities below)
    float m = -INFINITY;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 st;
      score_tile(kt, st);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float ms = m > -INFINITY ? m : 0.f;             // a fully masked row: every exp(-inf - 0) = 0, sum 0, output NaN as torch
    const uint64_t rowbase = ((uint64_t)bh * Lq + qi) * Lk;
    const uint32_t d_hi = (uint32_t)(rowbase >> 33), d_inner = drop_inner(sd, d_hi);      // seed half of the dropout hash, once per row
    // pass 2: O = dropout(exp(S - max)) V with the row sum gathered on the way (before dropout), scores recomputed tile by
    // tile (two MFMAs per tile - cheaper than keeping them); the 1 / sum goes onto the output rows at the end
    float sum = 0.f;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NT; ++kt) {
      f32x16 st;
      score_tile(kt, st);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float pv[8];
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {                  // a lane's keys come in runs of four: two hashes per run
          uint32_t keep = 0xfu;
          if (thresh) keep = drop_keep4(d_inner, d_hi, sd, rowbase + (kt * 32 + crow(8 * u + 4 * s4, hf)), thresh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __expf(st[8 * u + 4 * s4 + e] - ms);
            sum += p;
            pv[4 * s4 + e] = (keep >> e & 1u) ? (thresh ? p * inv_keep : p) : 0.f;
          }
        }
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(pv), frag_cols_tr(Vi, kt * 32 + 16 * u, lane), oacc, 0, 0, 0);
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    if (hf == 0 && qi < Lq) lse[((long)b * H + h) * Lq + qi] = m + __logf(sum);
    // 1 / sum is per QUERY = per lane here, but per accumulator ROW in oacc (lane <-> dim): through this wave's 32-float LDS strip
    float* strip = Rs + wave * 32;
    if (hf == 0) strip[lane] = 1.f / sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): the strip is written (same wave reads it back)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 iv = *reinterpret_cast<const float4*>(strip + 8 * g4 + 4 * hf);
      oacc[4 * g4 + 0] *= iv.x; oacc[4 * g4 + 1] *= iv.y; oacc[4 * g4 + 2] *= iv.z; oacc[4 * g4 + 3] *= iv.w;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = q0 + crow(r, hf);
      if (qr < Lq) o[((long)b * Lq + qr) * ldo + h * AD + (lane & 31)] = (bf16_t)oacc[r];
    }
  }
}

// ============================================================================================ backward
template <int NTQ, int NTK, bool AMASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ k,
                                                               long ldk, const bf16_t* __restrict__ v, long ldv,
                                                               const bf16_t* __restrict__ o, long ldo, const bf16_t* __restrict__ dout,
                                                               long lddo, const float* __restrict__ lse, const uint8_t* __restrict__ kpm,
                                                               const float* __restrict__ amask, bf16_t* __restrict__ dq, long lddq,
                                                               bf16_t* __restrict__ dk, long lddk, bf16_t* __restrict__ dv, long lddv,
                                                               int H, int Lq, int Lk, float scale, uint32_t thresh, float inv_keep,
                                                               uint32_t seed, const uint32_t* seed_ptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LqP = NTQ * 32, LkP = NTK * 32;
  unsigned char* Qi = smem;
  unsigned char* Ki = Qi + LqP * AROW;
  unsigned char* Vi = Ki + LkP * AROW;
  unsigned char* Di = Vi + LkP * AROW;                  // dO image
  float* Ls = reinterpret_cast<float*>(Di + LqP * AROW);   // lse   [LqP]
  float* De = Ls + LqP;                                 // delta [LqP]
  float* Kb = De + LqP;                                 // key bias [LkP]
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwave = nthr >> 6;
  stage_image(Qi, q + (long)b * Lq * ldq + h * AD, ldq, Lq, LqP, tid, nthr);
  stage_image(Ki, k + (long)b * Lk * ldk + h * AD, ldk, Lk, LkP, tid, nthr);
  stage_image(Vi, v + (long)b * Lk * ldv + h * AD, ldv, Lk, LkP, tid, nthr);
  stage_image(Di, dout + (long)b * Lq * lddo + h * AD, lddo, Lq, LqP, tid, nthr);
  stage_key_bias(Kb, kpm ? kpm + (long)b * Lk : nullptr, Lk, LkP, tid, nthr);
  for (int i = tid; i < LqP; i += nthr) {
    float dl = 0.f, l = 0.f;
    if (i < Lq) {
      l = lse[((long)b * H + h) * Lq + i];
      const uint4* op = reinterpret_cast<const uint4*>(o + ((long)b * Lq + i) * ldo + h * AD);
      const uint4* dp = reinterpret_cast<const uint4*>(dout + ((long)b * Lq + i) * lddo + h * AD);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, op[c]), g = __builtin_bit_cast(bf16x8, dp[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)a[e] * (float)g[e];
      }
    }
    Ls[i] = l;
    De[i] = dl;
  }
  __syncthreads();
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const int hf = lane >> 5;

  // ---------------- pass A: this wave's 32 queries, all keys -> dQ
  for (int q0 = wave * 32; q0 < Lq; q0 += nwave * 32) {
    const int qi = q0 + (lane & 31);
    const float lq = Ls[qi], dlq = De[qi];
    const uint64_t rowbase = ((uint64_t)bh * Lq + qi) * Lk;
    const uint32_t d_hi = (uint32_t)(rowbase >> 33), d_inner = drop_inner(sd, d_hi);
    f32x16 dqa;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqa[r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < NTK; ++kt) {
      f32x16 st, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 qf = frag_rows(Qi, q0, ks, lane), df = frag_rows(Di, q0, ks, lane);
        st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ki, kt * 32, ks, lane), qf, st, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vi, kt * 32, ks, lane), df, dp, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float ds[8];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int key0 = kt * 32 + 16 * u + 8 * g2 + 4 * hf;
          const float4 kb = *reinterpret_cast<const float4*>(Kb + key0);
          const float kbv[4] = {kb.x, kb.y, kb.z, kb.w};
          uint32_t keep = 0xfu;
          if (thresh) keep = drop_keep4(d_inner, d_hi, sd, rowbase + key0, thresh);      // four consecutive keys: two hashes
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 8 * u + 4 * g2 + e, key = key0 + e;
            float sc = st[r] * scale + kbv[e];
            if (AMASK) { if (qi < Lq && key < Lk) sc += amask[(long)qi * Lk + key]; }
            const float p = __expf(sc - lq);
            float g = dp[r];
            if (thresh) g = (keep >> e & 1u) ? g * inv_keep : 0.f;
            ds[4 * g2 + e] = (qi < Lq) ? p * (g - dlq) : 0.f;
          }
        }
        dqa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(ds), frag_cols_tr(Ki, kt * 32 + 16 * u, lane), dqa, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qr = q0 + crow(r, hf);
      if (qr < Lq) dq[((long)b * Lq + qr) * lddq + h * AD + (lane & 31)] = (bf16_t)(dqa[r] * scale);
    }
  }

  // ---------------- pass B: this wave's 32 keys, all queries -> dK, dV (accumulated transposed: rows = dims, cols = keys)
  for (int k0 = wave * 32; k0 < Lk; k0 += nwave * 32) {
    const int kj = k0 + (lane & 31);                    // this lane's key
    const float kbj = Kb[kj];
    const uint32_t k_hi = (uint32_t)((((uint64_t)bh * Lq) * Lk) >> 33), k_inner = drop_inner(sd, k_hi);   // seed half of the dropout hash
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }
#pragma unroll 1
    for (int qt = 0; qt < NTQ; ++qt) {
      f32x16 sa, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = frag_rows(Ki, k0, ks, lane), vf = frag_rows(Vi, k0, ks, lane);
        sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qi, qt * 32, ks, lane), kf, sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Di, qt * 32, ks, lane), vf, dp, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float pd[8], ds[8];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int qr0 = qt * 32 + 16 * u + 8 * g2 + 4 * hf;
          const float4 l4 = *reinterpret_cast<const float4*>(Ls + qr0), d4 = *reinterpret_cast<const float4*>(De + qr0);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 8 * u + 4 * g2 + e, qr = qr0 + e;
            float sc = sa[r] * scale + kbj;
            if (AMASK) { if (qr < Lq && kj < Lk) sc += amask[(long)qr * Lk + kj]; }
            float p = (qr < Lq) ? __expf(sc - lv[e]) : 0.f;
            float g = dp[r], pk = p;
            if (thresh) {
              const bool keep = drop_keep_in(k_inner, k_hi, sd, ((uint64_t)bh * Lq + qr) * Lk + kj, thresh);
              g = keep ? g * inv_keep : 0.f;
              pk = keep ? p * inv_keep : 0.f;
            }
            pd[4 * g2 + e] = pk;
            ds[4 * g2 + e] = p * (g - dv4[e]);
          }
        }
        dvt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Di, qt * 32 + 16 * u, lane), pack8(pd), dvt, 0, 0, 0);
        dkt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_cols_tr(Qi, qt * 32 + 16 * u, lane), pack8(ds), dkt, 0, 0, 0);
      }
    }
    // dvt/dkt: register r <-> dim crow(r, hf), lane <-> key: four 4-dim (8-byte) stores per row
    if (kj < Lk) {
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 a, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (bf16_t)(dkt[4 * g4 + e] * scale); c[e] = (bf16_t)dvt[4 * g4 + e]; }
        const int d0 = 8 * g4 + 4 * hf;
        *reinterpret_cast<bf16x4*>(dk + ((long)b * Lk + kj) * lddk + h * AD + d0) = a;
        *reinterpret_cast<bf16x4*>(dv + ((long)b * Lk + kj) * lddv + h * AD + d0) = c;
      }
    }
  }
}

template <typename K>
static int set_attr_once(K kern, bool& done, size_t bytes, const char* what) {
  if (done) return 0;
  done = true;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

static bool aligned_ok(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 7) == 0; }

// returns -1 if outside the envelope (caller falls back to the generic kernels)
int attn_fwd_mfma_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                      float* lse, const uint8_t* kpm, const float* amask, int B, int H, int Lq, int Lk, float drop_p,
                      uint32_t seed, const uint32_t* seed_ptr, hipStream_t st) {
  if (Lk > 32 * AMAXT || Lq > 32 * AMAXT) return -1;
  if (!aligned_ok(q, ldq) || !aligned_ok(k, ldk) || !aligned_ok(v, ldv)) return -1;
  const int nt = (Lk + 31) / 32, LqP = (Lq + 31) & ~31;
  const int nwave = 4;     // all four waves stage K/V/Q; waves beyond the query tiles then idle
  const size_t lds = (size_t)(2 * nt * 32 + LqP) * AROW + (size_t)(nt * 32 + nwave * 32) * sizeof(float);
  const float scale = 1.f / sqrtf((float)AD);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H), block(64 * nwave);
#define SEDT_AF1(NT_, AM_)                                                                                                 \
  {                                                                                                                        \
    static bool done = false;                                                                                              \
    if (set_attr_once(attn_fwd_mfma_kernel<NT_, AM_>, done, 64 * 1024, "attention_fwd")) return 1;                         \
    hipLaunchKernelGGL((attn_fwd_mfma_kernel<NT_, AM_>), grid, block, lds, st, (const bf16_t*)q, (long)ldq, (const bf16_t*)k, \
                       (long)ldk, (const bf16_t*)v, (long)ldv, (bf16_t*)o, (long)ldo, lse, kpm, amask, H, Lq, Lk, scale,   \
                       th, ik, seed, seed_ptr);                                                                            \
  }
#define SEDT_AF(NT_)                                                                                                       \
  case NT_:                                                                                                                \
    if (amask) SEDT_AF1(NT_, true) else SEDT_AF1(NT_, false)                                                               \
    break;
  switch (nt) {
    SEDT_AF(1) SEDT_AF(2) SEDT_AF(3) SEDT_AF(4) SEDT_AF(5) SEDT_AF(6) SEDT_AF(7) SEDT_AF(8)
    default: return -1;
  }
#undef SEDT_AF
#undef SEDT_AF1
  return check_launch("attention_fwd_mfma");
}

template <int NTQ>
static int launch_bwd_q(int ntk, dim3 grid, dim3 block, size_t lds, hipStream_t st, const void* q, int64_t ldq, const void* k,
                        int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo, const void* dout, int64_t lddo,
                        const float* lse, const uint8_t* kpm, const float* amask, void* dq, int64_t lddq, void* dk, int64_t lddk,
                        void* dv, int64_t lddv, int H, int Lq, int Lk, float scale, uint32_t th, float ik, uint32_t seed,
                        const uint32_t* seed_ptr) {
#define SEDT_AB1(NTK_, AM_)                                                                                                \
  {                                                                                                                        \
    static bool done = false;                                                                                              \
    if (set_attr_once(attn_bwd_mfma_kernel<NTQ, NTK_, AM_>, done, 96 * 1024, "attention_bwd")) return 1;                   \
    hipLaunchKernelGGL((attn_bwd_mfma_kernel<NTQ, NTK_, AM_>), grid, block, lds, st, (const bf16_t*)q, (long)ldq,          \
                       (const bf16_t*)k, (long)ldk, (const bf16_t*)v, (long)ldv, (const bf16_t*)o, (long)ldo,              \
                       (const bf16_t*)dout, (long)lddo, lse, kpm, amask, (bf16_t*)dq, (long)lddq, (bf16_t*)dk, (long)lddk, \
                       (bf16_t*)dv, (long)lddv, H, Lq, Lk, scale, th, ik, seed, seed_ptr);                                 \
  }
#define SEDT_AB(NTK_)                                                                                                      \
  case NTK_:                                                                                                               \
    if (amask) SEDT_AB1(NTK_, true) else SEDT_AB1(NTK_, false)                                                             \
    break;
  switch (ntk) {
    SEDT_AB(1) SEDT_AB(2) SEDT_AB(3) SEDT_AB(4)
    default: return -1;
  }
#undef SEDT_AB
#undef SEDT_AB1
  return check_launch("attention_bwd_mfma");
}

int attn_bwd_mfma_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o,
                      int64_t ldo, const void* dout, int64_t lddo, const float* lse, const uint8_t* kpm, const float* amask,
                      void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv, int64_t lddv, int B, int H, int Lq, int Lk,
                      float drop_p, uint32_t seed, const uint32_t* seed_ptr, hipStream_t st) {
  if (Lk > 128 || Lq > 128) return -1;                    // template matrix kept small: 4 x 4 tile counts
  if (!aligned_ok(q, ldq) || !aligned_ok(k, ldk) || !aligned_ok(v, ldv) || !aligned_ok(dout, lddo)) return -1;
  if ((lddk & 3) || (lddv & 3) || (reinterpret_cast<uintptr_t>(dk) & 7) || (reinterpret_cast<uintptr_t>(dv) & 7)) return -1;
  const int ntq = (Lq + 31) / 32, ntk = (Lk + 31) / 32;
  const size_t lds = (size_t)(2 * ntq * 32 + 2 * ntk * 32) * AROW + (size_t)(2 * ntq * 32 + ntk * 32) * sizeof(float);
  const int nwave = std::min(4, std::max(ntq, ntk));
  const float scale = 1.f / sqrtf((float)AD);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H), block(64 * nwave);
#define SEDT_ARGS ntk, grid, block, lds, st, q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, lse, kpm, amask, dq, lddq, dk, lddk, dv, \
                  lddv, H, Lq, Lk, scale, th, ik, seed, seed_ptr
  switch (ntq) {
    case 1: return launch_bwd_q<1>(SEDT_ARGS);
    case 2: return launch_bwd_q<2>(SEDT_ARGS);
    case 3: return launch_bwd_q<3>(SEDT_ARGS);
    case 4: return launch_bwd_q<4>(SEDT_ARGS);
    default: return -1;
  }
#undef SEDT_ARGS
}

}  // namespace sedt

