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

// ============================================================================================ fused encoder block head
// Pre-norm encoder self-attention up to the context (reference sedt/transformer.py:196-199):
//     xn = LayerNorm1(x);  q = k = (xn + pos) Wqk^T + b;  v = xn Wv^T + b;  ctx = softmax(q k^T / sqrt(32) + key mask) dropout . v
// ONE launch instead of LayerNorm, the grouped Q|K / V projection GEMMs and the attention core: a workgroup = (clip, pair of
// heads), 8 waves = 4 slabs of 32 tokens x 2 heads.  A wave keeps its 32 x 256 slab of x in registers IN MFMA A-FRAGMENT
// LAYOUT (lane = token row, 16 chunks of 8 channels: 64 VGPRs), normalises it there (row statistics: in-lane sums + one
// cross-half shuffle), streams the 3 x 32 weight rows of its head from L2 as B fragments (16 bytes per lane per k-step) and
// accumulates the Q, K, V tiles (3 x 16 MFMAs).  The tiles go to the [token][32] bf16 LDS images of the attention core
// (the same code as attn_fwd_mfma_kernel from there on), so Q, K, V and the normalised activations never travel through HBM
// in a no-grad forward.  TRAIN additionally writes what the (unfused) backward kernels read: xn, xn + pos, the LayerNorm row
// statistics, q | k and v (coalesced 16-byte copies out of the LDS images).
// Envelope: d_model 256, 8 heads of 32, S <= 128 tokens, bf16.
// sum over the 32 lanes of each half wave, returned in every lane: DPP row shifts + row_bcast:15 (VALU rate; five
// ds_bpermute round trips per sum - what __shfl_xor compiles to - made the LayerNorm statistics the longest part of the kernel)
__device__ __forceinline__ float halfwave_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));   // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));   // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));   // row_shr:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));   // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true));   // row_bcast:15 into rows 1, 3
  const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
  const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
  return (threadIdx.x & 32) ? hi : lo;
}

constexpr int EF_D = 256, EF_S = 128, EF_IMG = EF_S * AROW;
constexpr int EF_AP = 128 * 2 + 16;          // row pitch of the staged half images (272 B: conflict-free 16-byte fragment reads)

template <bool TRAIN>
__global__ __launch_bounds__(512) void enc_attn_fused_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ pos,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const bf16_t* __restrict__ w_in, const float* __restrict__ b_in,
                                                             bf16_t* __restrict__ ctx, float* __restrict__ lse,
                                                             bf16_t* __restrict__ xn_out, bf16_t* __restrict__ xnp_out,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             bf16_t* __restrict__ qk_out, bf16_t* __restrict__ v_out,
                                                             const uint8_t* __restrict__ kpm, int S, float scale, uint32_t thresh,
                                                             float inv_keep, uint32_t seed, const uint32_t* seed_ptr, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // projection phase: An | Ap (normalised activation halves, [128 tokens][128 channels], 272-byte rows) | Wl (the head pair's
  // 192 weight rows, same shape) - all filled with fully coalesced 16-byte accesses and read back as MFMA fragments with
  // ds_read_b128 (the first version loaded fragment-shaped operands - 32 rows x 32 B per wave instruction - straight from
  // L2 and was bound by the texture-address path: 20 of its 38 us).  The attention images alias that area afterwards.
  unsigned char* An = smem;
  unsigned char* Ap = An + EF_S * EF_AP;
  unsigned char* Wl = Ap + EF_S * EF_AP;
  float* Kb = reinterpret_cast<float*>(Wl + 192 * EF_AP);      // [128] additive key bias (lives through both phases)
  float* Rs = Kb + EF_S;                              // [8 waves][32] 1 / row-sum strips
  unsigned char* Kimg = smem;                         // [2 heads][128][32] bf16
  unsigned char* Vimg = Kimg + 2 * EF_IMG;
  unsigned char* Qimg = Vimg + 2 * EF_IMG;
  const int b = blockIdx.x >> 2, hp = blockIdx.x & 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = wave >> 2, slab = wave & 3, head = hp * 2 + hh, hf = lane >> 5;
  stage_key_bias(Kb, kpm ? kpm + (long)b * S : nullptr, S, EF_S, tid, 512);
  // ---- the clip's x (and pos) tile in registers: thread <-> column chunk cc (8 channels) of rows rb + 16 i
  const int cc = tid & 31, rb = tid >> 5;
  bf16x8 xv[8], pz[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = rb + 16 * i;
    const long o = ((long)b * S + row) * EF_D + cc * 8;
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    xv[i] = row < S ? *reinterpret_cast<const bf16x8*>(x + o) : z;
    pz[i] = row < S ? *reinterpret_cast<const bf16x8*>(pos + o) : z;
  }
  // first half of the weight rows (global row of staged row r: q / k / v block r >> 6, rows hp*64 + (r & 63))
  auto wsrc = [&](int q, int half) {
    const int id = tid + 512 * q, r = id >> 4, wc = id & 15;
    return w_in + ((long)(r >> 6) * EF_D + hp * 64 + (r & 63)) * EF_D + half * 128 + wc * 8;
  };
  uint4 wreg[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) wreg[q] = *reinterpret_cast<const uint4*>(wsrc(q, 0));
  float gv[8], ev[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { gv[e] = gamma[cc * 8 + e]; ev[e] = beta[cc * 8 + e]; }
  // ---- LayerNorm in registers: a row's 32 chunks sit in the 32 lanes of a half wave (two-pass mean / variance)
  const bool writer = TRAIN && hp == 0;                 // one workgroup of the clip writes the shared by-products
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = rb + 16 * i;
    float s1 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s1 += (float)xv[i][e];
    s1 = halfwave_sum(s1);
    const float mu = s1 * (1.f / EF_D);
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = (float)xv[i][e] - mu; s2 += d * d; }
    s2 = halfwave_sum(s2);
    const float rs = rsqrtf(s2 * (1.f / EF_D) + 1e-5f);
    bf16x8 an, ap;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float o = ((float)xv[i][e] - mu) * rs * gv[e] + ev[e];
      an[e] = row < S ? (bf16_t)o : (bf16_t)0.f;
      ap[e] = row < S ? (bf16_t)(o + (float)pz[i][e]) : (bf16_t)0.f;
    }
    xv[i] = an;
    pz[i] = ap;
    if (writer && row < S) {
      const long g = (long)b * S + row;
      *reinterpret_cast<bf16x8*>(xn_out + g * EF_D + cc * 8) = an;
      *reinterpret_cast<bf16x8*>(xnp_out + g * EF_D + cc * 8) = ap;
      if (cc == 0) { mean_out[g] = mu; rstd_out[g] = rs; }
    }
  }
  // ---- projections: Q, K from xn + pos, V from xn, in two K halves of 128 channels
  f32x16 aq, ak, av;
#pragma unroll
  for (int r = 0; r < 16; ++r) { aq[r] = 0.f; ak[r] = 0.f; av[r] = 0.f; }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                          // the first half's fragments are read
    if ((cc >> 4) == half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        *reinterpret_cast<bf16x8*>(An + (rb + 16 * i) * EF_AP + (cc & 15) * 16) = xv[i];
        *reinterpret_cast<bf16x8*>(Ap + (rb + 16 * i) * EF_AP + (cc & 15) * 16) = pz[i];
      }
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int id = tid + 512 * q;
      *reinterpret_cast<uint4*>(Wl + (id >> 4) * EF_AP + (id & 15) * 16) = wreg[q];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) wreg[q] = *reinterpret_cast<const uint4*>(wsrc(q, 1));      // in flight under the MFMAs
    }
    if (!(dbg & 1)) {
      const unsigned char* arow = Ap + (slab * 32 + (lane & 31)) * EF_AP + 16 * hf;
      const unsigned char* nrow = An + (slab * 32 + (lane & 31)) * EF_AP + 16 * hf;
      const unsigned char* wrow = Wl + (hh * 32 + (lane & 31)) * EF_AP + 16 * hf;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 apf = *reinterpret_cast<const bf16x8*>(arow + 32 * ks);
        const bf16x8 anf = *reinterpret_cast<const bf16x8*>(nrow + 32 * ks);
        const bf16x8 bq = *reinterpret_cast<const bf16x8*>(wrow + 32 * ks);
        const bf16x8 bk = *reinterpret_cast<const bf16x8*>(wrow + 64 * EF_AP + 32 * ks);
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wrow + 128 * EF_AP + 32 * ks);
        aq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apf, bq, aq, 0, 0, 0);
        ak = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apf, bk, ak, 0, 0, 0);
        av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(anf, bv, av, 0, 0, 0);
      }
    }
  }
  __syncthreads();                                      // every fragment is read: the staging area becomes the Q / K / V images
  // ---- bias, bf16, into the LDS images of this head: accumulator register r of half hf <-> token slab*32 + crow(r, hf)
  {
    const float biq = b_in[head * AD + (lane & 31)], bik = b_in[EF_D + head * AD + (lane & 31)],
                biv = b_in[2 * EF_D + head * AD + (lane & 31)];
    bf16_t* Qh = reinterpret_cast<bf16_t*>(Qimg + hh * EF_IMG);
    bf16_t* Kh = reinterpret_cast<bf16_t*>(Kimg + hh * EF_IMG);
    bf16_t* Vh = reinterpret_cast<bf16_t*>(Vimg + hh * EF_IMG);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = slab * 32 + crow(r, hf);
      Qh[t * AD + (lane & 31)] = (bf16_t)(aq[r] + biq);
      Kh[t * AD + (lane & 31)] = (bf16_t)(ak[r] + bik);
      Vh[t * AD + (lane & 31)] = (bf16_t)(av[r] + biv);
    }
  }
  __syncthreads();
  if (TRAIN) {
    // q | k -> qk_out [B*S][512], v -> v_out [B*S][256]: one 16-byte chunk per thread and image (128 rows x 4 chunks)
    const int r = tid >> 2, c = tid & 3;
    if (r < S) {
      const long g = (long)b * S + r;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int hd = hp * 2 + h2;
        *reinterpret_cast<uint4*>(qk_out + g * (2 * EF_D) + hd * AD + c * 8) = *reinterpret_cast<const uint4*>(Qimg + h2 * EF_IMG + r * AROW + c * 16);
        *reinterpret_cast<uint4*>(qk_out + g * (2 * EF_D) + EF_D + hd * AD + c * 8) = *reinterpret_cast<const uint4*>(Kimg + h2 * EF_IMG + r * AROW + c * 16);
        *reinterpret_cast<uint4*>(v_out + g * EF_D + hd * AD + c * 8) = *reinterpret_cast<const uint4*>(Vimg + h2 * EF_IMG + r * AROW + c * 16);
      }
    }
  }
  // ---- attention core of (head, query slab): identical to attn_fwd_mfma_kernel<4, false>
  const unsigned char* Ki = Kimg + hh * EF_IMG;
  const unsigned char* Vi = Vimg + hh * EF_IMG;
  const unsigned char* Qi = Qimg + hh * EF_IMG;
  const uint32_t sd = eff_seed(seed, seed_ptr);
  const int q0 = slab * 32, qi = q0 + (lane & 31), bh = b * 8 + head;
  if (q0 >= S || (dbg & 2)) return;
  const bf16x8 qf0 = frag_rows(Qi, q0, 0, lane), qf1 = frag_rows(Qi, q0, 1, lane);
  auto score_tile = [&](int kt, f32x16& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ki, kt * 32, 0, lane), qf0, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ki, kt * 32, 1, lane), qf1, st, 0, 0, 0);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 kb = *reinterpret_cast<const float4*>(Kb + kt * 32 + 8 * g4 + 4 * hf);
      st[4 * g4 + 0] = st[4 * g4 + 0] * scale + kb.x;
      st[4 * g4 + 1] = st[4 * g4 + 1] * scale + kb.y;
      st[4 * g4 + 2] = st[4 * g4 + 2] * scale + kb.z;
      st[4 * g4 + 3] = st[4 * g4 + 3] * scale + kb.w;
    }
  };
  // (as attn_fwd_mfma_kernel: row maximum first, probabilities + row sum in one pass, 1 / sum on the output rows at the end)
  float m = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < 4; ++kt) {
    f32x16 st;
    score_tile(kt, st);
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, st[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float ms = m > -INFINITY ? m : 0.f;
  const uint64_t rowbase = ((uint64_t)bh * S + qi) * S;
  const uint32_t d_hi = (uint32_t)(rowbase >> 33), d_inner = drop_inner(sd, d_hi);
  float ssum = 0.f;
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < 4; ++kt) {
    f32x16 st;
    score_tile(kt, st);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float pv[8];
#pragma unroll
      for (int s4 = 0; s4 < 2; ++s4) {
        uint32_t keep = 0xfu;
        if (thresh) keep = drop_keep4(d_inner, d_hi, sd, rowbase + (kt * 32 + crow(8 * u + 4 * s4, hf)), thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __expf(st[8 * u + 4 * s4 + e] - ms);
          ssum += p;
          pv[4 * s4 + e] = (keep >> e & 1u) ? (thresh ? p * inv_keep : p) : 0.f;
        }
      }
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(pv), frag_cols_tr(Vi, kt * 32 + 16 * u, lane), oacc, 0, 0, 0);
    }
  }
  ssum += __shfl_xor(ssum, 32, 64);
  if (hf == 0 && qi < S) lse[((long)b * 8 + head) * S + qi] = m + __logf(ssum);
  float* strip = Rs + wave * 32;
  if (hf == 0) strip[lane] = 1.f / ssum;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 iv = *reinterpret_cast<const float4*>(strip + 8 * g4 + 4 * hf);
    oacc[4 * g4 + 0] *= iv.x; oacc[4 * g4 + 1] *= iv.y; oacc[4 * g4 + 2] *= iv.z; oacc[4 * g4 + 3] *= iv.w;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int qr = q0 + crow(r, hf);
    if (qr < S) ctx[((long)b * S + qr) * EF_D + head * AD + (lane & 31)] = (bf16_t)oacc[r];
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

int enc_attn_fused_launch(const void* x, const void* pos, const float* gamma, const float* beta, const void* w_in, const float* b_in,
                          void* ctx, float* lse, void* xn, void* xnp, float* mean, float* rstd, void* qk, void* v, const uint8_t* kpm,
                          int B, int S, float drop_p, uint32_t seed, const uint32_t* seed_ptr, hipStream_t st) {
  const size_t lds = (size_t)(2 * EF_S + 192) * EF_AP + (size_t)(EF_S + 8 * 32) * sizeof(float);       // 123392 B: one workgroup per CU
  const float scale = 1.f / sqrtf((float)AD);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  const bool train = xn != nullptr;
#ifdef SEDT_DEV                      // ablation switch (1: no projections, 2: no attention - WRONG results): developer builds only
  static const int dbg = sedt::dev_getenv("SEDT_ENC_DBG") ? atoi(sedt::dev_getenv("SEDT_ENC_DBG")) : 0;
#else
  const int dbg = 0;
#endif
  dim3 grid(B * 4), block(512);
  if (train) {
    static bool done = false;
    if (set_attr_once(enc_attn_fused_kernel<true>, done, 128 * 1024, "enc_attn_fused")) return 1;
    hipLaunchKernelGGL(enc_attn_fused_kernel<true>, grid, block, lds, st, (const bf16_t*)x, (const bf16_t*)pos, gamma, beta,
                       (const bf16_t*)w_in, b_in, (bf16_t*)ctx, lse, (bf16_t*)xn, (bf16_t*)xnp, mean, rstd, (bf16_t*)qk, (bf16_t*)v, kpm,
                       S, scale, th, ik, seed, seed_ptr, dbg);
  } else {
    static bool done = false;
    if (set_attr_once(enc_attn_fused_kernel<false>, done, 128 * 1024, "enc_attn_fused")) return 1;
    hipLaunchKernelGGL(enc_attn_fused_kernel<false>, grid, block, lds, st, (const bf16_t*)x, (const bf16_t*)pos, gamma, beta,
                       (const bf16_t*)w_in, b_in, (bf16_t*)ctx, lse, (bf16_t*)nullptr, (bf16_t*)nullptr, (float*)nullptr, (float*)nullptr,
                       (bf16_t*)nullptr, (bf16_t*)nullptr, kpm, S, scale, th, ik, seed, seed_ptr, dbg);
  }
  return check_launch("enc_attn_fused");
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

// LayerNorm1 + (Q | K | V) projections + attention core of a pre-norm encoder layer in one launch (bf16, d_model 256, 8 heads,
// S <= 128).  xn .. v may be null together (no-grad forward: nothing but ctx / lse is written).
extern "C" int sedt_encoder_attn_fwd(const void* x, const void* pos, const float* gamma, const float* beta, const void* w_in,
                                     const float* b_in, void* ctx, float* lse, void* xn, void* xnp, float* mean, float* rstd,
                                     void* qk, void* v, const uint8_t* kpm, int B, int S, int D, int H, float drop_p, uint32_t seed,
                                     const uint32_t* seed_ptr, int dtype, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(x && pos && gamma && beta && w_in && b_in && ctx && lse, "encoder_attn_fwd: null pointer");
  SEDT_REQUIRE(dtype == SEDT_BF16 && D == EF_D && H == 8 && S >= 1 && S <= EF_S && B >= 1,
               "encoder_attn_fwd: envelope is bf16, d_model 256, 8 heads, S <= 128 (got dtype %d D %d H %d S %d)", dtype, D, H, S);
  const bool any = xn || xnp || mean || rstd || qk || v, all = xn && xnp && mean && rstd && qk && v;
  SEDT_REQUIRE(any == all, "encoder_attn_fwd: the training by-products (xn, xnp, mean, rstd, qk, v) go together");
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "encoder_attn_fwd: drop_p out of range");
  SEDT_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(w_in) |
                 reinterpret_cast<uintptr_t>(ctx)) & 15) == 0, "encoder_attn_fwd: 16-byte aligned tensors required");
  return enc_attn_fused_launch(x, pos, gamma, beta, w_in, b_in, ctx, lse, xn, xnp, mean, rstd, qk, v, kpm, B, S, drop_p, seed, seed_ptr,
                               reinterpret_cast<hipStream_t>(stream));
}
