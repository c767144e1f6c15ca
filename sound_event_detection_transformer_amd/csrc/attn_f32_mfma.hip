// attn_f32_mfma.hip - the f32 (parity mode / bf16x3 mode) multi-head attention core on the exact-f32 matrix instruction.
//
// The f32 attention of csrc/norm_attn.hip is a VALU kernel: every lane walks a key row in LDS per score (one ds_read per FMA) - 90 us
// per encoder layer forward and 250 us backward at B = 64, 1.4 ms of the bf16x3 step.  v_mfma_f32_32x32x2_f32 does exact f32 FMA chains
// (the same arithmetic, another summation order) with 64 operand values feeding 2048 FMAs, so the same LDS images feed it 16 x fewer reads.
// One workgroup per (clip, head), four waves; head dim 32, Lq, Lk <= 128 (everything on the SEDT path: S = 128 / 124, Q = 11 / 21).
//
// Orientation trick (as in attn_mfma.hip): a wave computes S^T = K Q^T for ITS 32 queries - C[row = key][col = query] - so a lane owns one
// query column and the softmax statistics are in-lane (+ one exchange between the two half waves).  The probabilities then ARE the B operand
// of the next product: v_mfma_f32_32x32x2_f32 contracts two k values per instruction, k = 0 supplied by lanes 0-31 and k = 1 by lanes 32-63,
// and accumulator register r of a lane holds key row (r & 3) + 8 (r >> 2) + 4 half - so step r of O^T = V^T P^T takes the probability
// register r as B and the V row of that same key as A: no shuffle, no LDS round trip for P.  The backward uses the same scheme twice:
// query-major (dQ) and key-major (dK, dV), blockIdx.y selects.
// Dropout decisions are the counter hashes of the other attention kernels (same (seed, (bh * Lq + i) * Lk + j)): identical masks.
#include <math.h>
#include "common.h"

namespace sedt {

namespace af {
constexpr int DH = 32, PT = 33;          // head dim; LDS row pitch in floats (odd: 32 rows x one k conflict-free)
__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// rows [0, L) of a [B*L][ld] tensor's head slice -> dst[row][PT] as f32, rows [L, LP) zero
__device__ __forceinline__ void stage(float* dst, const float* src, long ld, int L, int LP, int tid) {
  for (int e = tid; e < LP * 8; e += 256) {
    const int r = e >> 3, c = (e & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < L) v = *reinterpret_cast<const float4*>(src + (long)r * ld + c);
    float* d = dst + r * PT + c;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// C[row][col] += sum_k A[arow0 + row][k] * B[brow0 + col][k] over the 32 head channels (16 MFMAs): both operands as LDS images
__device__ __forceinline__ f32x16 mm32(const float* A, int arow0, const float* B, int brow0, int l31, int half) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = A + (arow0 + l31) * PT + half;
  const float* bp = B + (brow0 + l31) * PT + half;
#pragma unroll
  for (int kk = 0; kk < DH; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
  return acc;
}
}  // namespace af

template <int NTK, bool AMASK, bool DROP>
__global__ __launch_bounds__(256) void attn_f32_fwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                                           const float* __restrict__ v, long ldv, float* __restrict__ o, long ldo,
                                                           float* __restrict__ lse, const uint8_t* __restrict__ kpm,
                                                           const float* __restrict__ amask, int H, int Lq, int Lk, float scale,
                                                           uint32_t thresh, float inv_keep, uint32_t seed, const uint32_t* seed_ptr) {
  using namespace af;
  extern __shared__ __attribute__((aligned(16))) float lds_af[];
  const int LqP = (Lq + 31) & ~31, LkP = NTK * 32;
  float* Qs = lds_af;
  float* Ks = Qs + LqP * PT;
  float* Vs = Ks + LkP * PT;
  float* Kb = Vs + LkP * PT;                         // [LkP] additive key bias: -inf for padded keys and keys beyond Lk
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  stage(Qs, q + (long)b * Lq * ldq + h * DH, ldq, Lq, LqP, tid);
  stage(Ks, k + (long)b * Lk * ldk + h * DH, ldk, Lk, LkP, tid);
  stage(Vs, v + (long)b * Lk * ldv + h * DH, ldv, Lk, LkP, tid);
  for (int j = tid; j < LkP; j += 256) Kb[j] = (j >= Lk || (kpm && kpm[(long)b * Lk + j])) ? -INFINITY : 0.f;
  __syncthreads();
  if (wave * 32 >= Lq) return;                       // (no barrier below)
  const int i = wave * 32 + l31;                     // this lane's query
  const bool qok = i < Lq;
  const uint32_t sd = eff_seed(seed, seed_ptr);
  f32x16 s[NTK];
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NTK; ++t) {
    s[t] = mm32(Ks, 32 * t, Qs, 32 * wave, l31, half);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * t + rowmap(r, half);
      float a = s[t][r] * scale + Kb[j];
      if (AMASK && qok && j < Lk) a += amask[(long)i * Lk + j];
      s[t][r] = a;
      m = fmaxf(m, a);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NTK; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(s[t][r] - m);
      s[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;
  if (qok && half == 0) lse[((long)b * H + h) * Lq + i] = m + __logf(sum);
  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  // (all sixteen V rows of a tile are read BEFORE the element loop and the loop is branch-free: with a load or a branch per element the
  // compiler waits lgkmcnt(0) in front of every MFMA - one exposed LDS round trip per instruction)
  const uint64_t ibase = ((uint64_t)bh * Lq + (qok ? i : 0)) * Lk;
#pragma unroll
  for (int t = 0; t < NTK; ++t) {
    float va[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) va[r] = Vs[(32 * t + rowmap(r, half)) * PT + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pv = s[t][r] * inv;
      if (DROP) {
        const int j = 32 * t + rowmap(r, half);
        const bool keep = drop_keep(sd, ibase + (uint64_t)min(j, Lk - 1), thresh);
        pv = keep ? pv * inv_keep : 0.f;
      }
      s[t][r] = pv;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
      oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(va[r], s[t][r], oacc, 0, 0, 0);            // O^T[d][query] += V[j][d] P[query][j]
  }
  if (qok) {
    float* op = o + ((long)b * Lq + i) * ldo + h * DH;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<float4*>(op + 8 * g4 + 4 * half) = make_float4(oacc[4 * g4], oacc[4 * g4 + 1], oacc[4 * g4 + 2], oacc[4 * g4 + 3]);
  }
}

// blockIdx.y == 0: dQ (a wave = 32 queries against every key tile); blockIdx.y == 1: dK, dV (a wave = 32 keys against every query tile)
template <int NTQ, int NTK, bool AMASK, bool DROP>
__global__ __launch_bounds__(256) void attn_f32_bwd_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                                           const float* __restrict__ v, long ldv, const float* __restrict__ o, long ldo,
                                                           const float* __restrict__ dout, long lddo, const float* __restrict__ lse,
                                                           const uint8_t* __restrict__ kpm, const float* __restrict__ amask,
                                                           float* __restrict__ dq, long lddq, float* __restrict__ dk, long lddk,
                                                           float* __restrict__ dv, long lddv, int H, int Lq, int Lk, float scale,
                                                           uint32_t thresh, float inv_keep, uint32_t seed, const uint32_t* seed_ptr) {
  using namespace af;
  extern __shared__ __attribute__((aligned(16))) float lds_af[];
  constexpr int LqP = NTQ * 32, LkP = NTK * 32;
  float* Qs = lds_af;
  float* Ks = Qs + LqP * PT;
  float* Vs = Ks + LkP * PT;
  float* Ds = Vs + LkP * PT;
  float* Ls = Ds + LqP * PT;       // [LqP] log-sum-exp
  float* De = Ls + LqP;            // [LqP] delta = dO . O
  float* Kb = De + LqP;            // [LkP] additive key bias: -inf for padded keys and keys beyond Lk
  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  stage(Qs, q + (long)b * Lq * ldq + h * DH, ldq, Lq, LqP, tid);
  stage(Ks, k + (long)b * Lk * ldk + h * DH, ldk, Lk, LkP, tid);
  stage(Vs, v + (long)b * Lk * ldv + h * DH, ldv, Lk, LkP, tid);
  stage(Ds, dout + (long)b * Lq * lddo + h * DH, lddo, Lq, LqP, tid);
  __syncthreads();
  for (int j = tid; j < LkP; j += 256) Kb[j] = (j >= Lk || (kpm && kpm[(long)b * Lk + j])) ? -INFINITY : 0.f;
  for (int i = tid; i < LqP; i += 256) {
    float a = 0.f, l = 0.f;
    if (i < Lq) {
      const float4* op = reinterpret_cast<const float4*>(o + ((long)b * Lq + i) * ldo + h * DH);
      float4 ov[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) ov[d] = op[d];
#pragma unroll
      for (int d = 0; d < 8; ++d)
        a += Ds[i * PT + 4 * d] * ov[d].x + Ds[i * PT + 4 * d + 1] * ov[d].y + Ds[i * PT + 4 * d + 2] * ov[d].z + Ds[i * PT + 4 * d + 3] * ov[d].w;
      l = lse[((long)b * H + h) * Lq + i];
    }
    De[i] = a;
    Ls[i] = l;
  }
  __syncthreads();
  const uint32_t sd = eff_seed(seed, seed_ptr);
  if (blockIdx.y == 0) {
    // ---------------------------------------------------------------- dQ: lane = query column, registers = key rows
    if (wave >= NTQ || wave * 32 >= Lq) return;
    const int i = wave * 32 + l31;
    const bool qok = i < Lq;
    const float li = Ls[i], di = De[i];
    const uint64_t ibase = ((uint64_t)bh * Lq + (qok ? i : 0)) * Lk;
    f32x16 gq;
#pragma unroll
    for (int r = 0; r < 16; ++r) gq[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NTK; ++t) {
      if (32 * t >= Lk) break;
      const f32x16 st = mm32(Ks, 32 * t, Qs, 32 * wave, l31, half);      // S^T  [key][query]
      const f32x16 pt = mm32(Vs, 32 * t, Ds, 32 * wave, l31, half);      // dP^T [key][query]
      float ka[16], kb[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ka[r] = Ks[(32 * t + rowmap(r, half)) * PT + l31];
        kb[r] = Kb[32 * t + rowmap(r, half)];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + rowmap(r, half);
        float a = st[r] * scale + kb[r];
        if (AMASK && qok && j < Lk) a += amask[(long)i * Lk + j];
        const float p = qok ? __expf(a - li) : 0.f;
        float dp = pt[r];
        if (DROP) dp = drop_keep(sd, ibase + (uint64_t)min(j, Lk - 1), thresh) ? dp * inv_keep : 0.f;
        ds[r] = p * (dp - di);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r)
        gq = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[r], ds[r], gq, 0, 0, 0);               // dQ^T[d][query] += K[j][d] dS[query][j]
    }
    if (qok) {
      float* gp = dq + ((long)b * Lq + i) * lddq + h * DH;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<float4*>(gp + 8 * g4 + 4 * half) =
            make_float4(gq[4 * g4] * scale, gq[4 * g4 + 1] * scale, gq[4 * g4 + 2] * scale, gq[4 * g4 + 3] * scale);
    }
    return;
  }
  // ------------------------------------------------------------------ dK, dV: lane = key column, registers = query rows
  if (wave >= NTK || wave * 32 >= Lk) return;
  const int j = wave * 32 + l31;
  const bool kok = j < Lk;
  const int jc = kok ? j : 0;
  const float kbias = Kb[j];
  f32x16 gk, gv;
#pragma unroll
  for (int r = 0; r < 16; ++r) { gk[r] = 0.f; gv[r] = 0.f; }
#pragma unroll
  for (int u = 0; u < NTQ; ++u) {
    if (32 * u >= Lq) break;
    const f32x16 su = mm32(Qs, 32 * u, Ks, 32 * wave, l31, half);        // S  [query][key]
    const f32x16 pu = mm32(Ds, 32 * u, Vs, 32 * wave, l31, half);        // dP [query][key]
    float qa[16], da[16], ls[16], de[16], ds[16], pd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * u + rowmap(r, half);
      qa[r] = Qs[i * PT + l31];
      da[r] = Ds[i * PT + l31];
      ls[r] = Ls[i];
      de[r] = De[i];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = 32 * u + rowmap(r, half);
      const bool live = i < Lq && kok;
      float a = su[r] * scale + kbias;
      if (AMASK && live) a += amask[(long)i * Lk + j];
      const float p = live ? __expf(a - ls[r]) : 0.f;
      float dp = pu[r];
      pd[r] = p;
      if (DROP) {
        const bool keep = drop_keep(sd, ((uint64_t)bh * Lq + min(i, Lq - 1)) * Lk + jc, thresh);
        dp = keep ? dp * inv_keep : 0.f;
        pd[r] = keep ? p * inv_keep : 0.f;
      }
      ds[r] = p * (dp - de[r]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      gk = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[r], ds[r], gk, 0, 0, 0);                 // dK^T[d][key] += Q[i][d] dS[i][key]
      gv = __builtin_amdgcn_mfma_f32_32x32x2f32(da[r], pd[r], gv, 0, 0, 0);                 // dV^T[d][key] += dO[i][d] Pd[i][key]
    }
  }
  if (kok) {
    float* kp = dk + ((long)b * Lk + j) * lddk + h * DH;
    float* vp = dv + ((long)b * Lk + j) * lddv + h * DH;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      *reinterpret_cast<float4*>(kp + 8 * g4 + 4 * half) =
          make_float4(gk[4 * g4] * scale, gk[4 * g4 + 1] * scale, gk[4 * g4 + 2] * scale, gk[4 * g4 + 3] * scale);
      *reinterpret_cast<float4*>(vp + 8 * g4 + 4 * half) = make_float4(gv[4 * g4], gv[4 * g4 + 1], gv[4 * g4 + 2], gv[4 * g4 + 3]);
    }
  }
}

template <typename K>
static int af_set_lds(K kern, bool& done, const char* what) {
  if (done) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    return 1;
  }
  done = true;
  return 0;
}

static bool af_aligned(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0; }

static bool af_on() {
  static int v = -1;
  if (v < 0) {
    const char* e = dev_getenv("SEDT_ATTN_F32_MFMA");      // developer A/B switch: 0 = the VALU kernels of norm_attn.hip
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// -1: outside the envelope (head dim 32 is implied by the callers; Lq, Lk <= 128; 16-byte aligned rows) - the caller uses the VALU kernel
int attn_f32_fwd_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo, float* lse,
                     const uint8_t* kpm, const float* amask, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                     const uint32_t* seed_ptr, hipStream_t st) {
  if (!af_on() || Lq > 128 || Lk > 128 || Lq < 1 || Lk < 1) return -1;
  if (!af_aligned(q, ldq) || !af_aligned(k, ldk) || !af_aligned(v, ldv) || !af_aligned(o, ldo)) return -1;
  const int ntk = Lk <= 32 ? 1 : 4, LqP = (Lq + 31) & ~31;             // instances: one key tile (the decoder's self-attention) or four
  const size_t lds = ((size_t)(LqP + 2 * ntk * 32) * af::PT + ntk * 32) * sizeof(float);
  const float scale = 1.f / sqrtf((float)af::DH);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H), block(256);
#define SEDT_AFF1(NT_, AM_, DR_)                                                                                                          \
  {                                                                                                                                       \
    static bool done = false;                                                                                                             \
    if (af_set_lds(attn_f32_fwd_kernel<NT_, AM_, DR_>, done, "attention_fwd (f32 mfma)")) return 1;                                       \
    hipLaunchKernelGGL((attn_f32_fwd_kernel<NT_, AM_, DR_>), grid, block, lds, st, (const float*)q, (long)ldq, (const float*)k, (long)ldk, \
                       (const float*)v, (long)ldv, (float*)o, (long)ldo, lse, kpm, amask, H, Lq, Lk, scale, th, ik, seed, seed_ptr);        \
  }
#define SEDT_AFF(NT_)                                                        \
  if (ntk == NT_) {                                                          \
    if (amask) { if (th) SEDT_AFF1(NT_, true, true) else SEDT_AFF1(NT_, true, false) } \
    else { if (th) SEDT_AFF1(NT_, false, true) else SEDT_AFF1(NT_, false, false) }     \
  }
  SEDT_AFF(1) SEDT_AFF(4)
#undef SEDT_AFF
#undef SEDT_AFF1
  return check_launch("attention_fwd_f32_mfma");
}

int attn_f32_bwd_try(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* o, int64_t ldo,
                     const void* dout, int64_t lddo, const float* lse, const uint8_t* kpm, const float* amask, void* dq, int64_t lddq,
                     void* dk, int64_t lddk, void* dv, int64_t lddv, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                     const uint32_t* seed_ptr, hipStream_t st) {
  if (!af_on() || Lq > 128 || Lk > 128 || Lq < 1 || Lk < 1) return -1;
  if (!af_aligned(q, ldq) || !af_aligned(k, ldk) || !af_aligned(v, ldv) || !af_aligned(o, ldo) || !af_aligned(dout, lddo) ||
      !af_aligned(dq, lddq) || !af_aligned(dk, lddk) || !af_aligned(dv, lddv))
    return -1;
  const int ntq = (Lq + 31) / 32, ntk = (Lk + 31) / 32;
  // instances: the query side 1 tile (the decoder's Q = 11 / 21 queries) or 4 (encoder), the key side 1 or 4
  const int NQ = ntq <= 1 ? 1 : 4, NK = ntk <= 1 ? 1 : 4;
  const size_t lds = ((size_t)(2 * NQ * 32 + 2 * NK * 32) * af::PT + 2 * NQ * 32 + NK * 32) * sizeof(float);
  const float scale = 1.f / sqrtf((float)af::DH);
  const uint32_t th = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  const float ik = 1.f / (1.f - drop_p);
  dim3 grid(B * H, 2), block(256);
#define SEDT_AFB1(NQ_, NK_, AM_, DR_)                                                                                                    \
  {                                                                                                                                      \
    static bool done = false;                                                                                                            \
    if (af_set_lds(attn_f32_bwd_kernel<NQ_, NK_, AM_, DR_>, done, "attention_bwd (f32 mfma)")) return 1;                                 \
    hipLaunchKernelGGL((attn_f32_bwd_kernel<NQ_, NK_, AM_, DR_>), grid, block, lds, st, (const float*)q, (long)ldq, (const float*)k,      \
                       (long)ldk, (const float*)v, (long)ldv, (const float*)o, (long)ldo, (const float*)dout, (long)lddo, lse, kpm,       \
                       amask, (float*)dq, (long)lddq, (float*)dk, (long)lddk, (float*)dv, (long)lddv, H, Lq, Lk, scale, th, ik, seed,     \
                       seed_ptr);                                                                                                         \
  }
#define SEDT_AFB(NQ_, NK_)                                                                       \
  if (NQ == NQ_ && NK == NK_) {                                                                  \
    if (amask) { if (th) SEDT_AFB1(NQ_, NK_, true, true) else SEDT_AFB1(NQ_, NK_, true, false) } \
    else { if (th) SEDT_AFB1(NQ_, NK_, false, true) else SEDT_AFB1(NQ_, NK_, false, false) }     \
  }
  SEDT_AFB(1, 1) SEDT_AFB(1, 4) SEDT_AFB(4, 1) SEDT_AFB(4, 4)
#undef SEDT_AFB
#undef SEDT_AFB1
  return check_launch("attention_bwd_f32_mfma");
}

}  // namespace sedt
