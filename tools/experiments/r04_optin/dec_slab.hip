// dec_slab.hip - one pre-norm decoder layer (reference sedt/transformer.py:263-284) in ONE launch, a workgroup per clip.
//
// At B = 64 the decoder works on M = B * Q = 704 rows: every launch of the per-op chain (3 LayerNorms, 2 grouped projection launches,
// 2 attention cores, 2 out-projections, 2 FFN GEMMs = 11 launches, ~91 us per layer) is launch latency, not work.  Here the Q <= 32
// query rows of a clip are ONE slab (csrc/slab.h): they stay in LDS from the first LayerNorm to the FFN output while the layer's
// weights stream L2 -> registers (2.8 MB per clip), and the 8 waves are the 8 heads in both attention cores.  What stays outside is
// the K | V projection of the encoder memory (M = B * S rows: a proper GEMM; sedt_igemm_group), whose output the kernel reads as
// kc / vc.
//
//   tn = LN1(tgt); q|k = (tn + qpos) Wqk^T + b; v = tn Wv^T + b; t1 = tgt + drop(softmax(q k^T / sqrt(32) + tgt_mask) drop . v  Wo^T + bo)
//   t1n = LN2(t1); qc = (t1n + qpos) Wq^T + b;                    t2 = t1 + drop(softmax(qc kc^T / sqrt(32) + key padding) drop . vc  Wo'^T + bo')
//   t2n = LN3(t2);                                                out = t2 + drop(drop(relu(t2n W1^T + b1)) W2^T + b2)
//
// Rounding points and dropout decisions are those of the per-op chain (bf16 at every tensor it materialises; counter hashes of
// (seed, element index)), so the per-op backward kernels consume the by-products (TRAIN).
// Envelope: bf16, d_model 256, 8 heads of 32, Q <= 32, S <= 128, dim_feedforward a multiple of 512.
#include "slab.h"
#include "attn_frag.h"

namespace sedt {

using slab::u32x4;
using slab::XP;

constexpr int DS_D = 256, DS_H = 8, DS_LK = 128;
constexpr int DS_HP = 512 + 8;                        // element pitch of a hidden chunk tile
constexpr int DS_IMGK = DS_LK * 64 + 32;              // bytes of one head's [128][32] image of the memory keys / values
constexpr int DS_IMGQ = 32 * 64 + 32;                 // bytes of one head's [32][32] image of the clip's own q / k / v
// LDS map (bytes)
constexpr int DS_R0 = 2 * DS_H * DS_IMGK;             // 131584: cross K | V images; before that phase: the self-attention working set
constexpr int DS_KB = DS_R0, DS_RS = DS_KB + DS_LK * 4, DS_R2 = DS_RS + 8 * 32 * 4;       // key bias, 1/sum strips, q images / context tile
constexpr int DS_TILE = 32 * XP * 2;                  // 16896: one [32][256] bf16 tile
constexpr int DS_BIAS = DS_R2 + DS_TILE;              // 150016: the layer's small bias vectors (self in_proj 768 | self out 256 | cross q 256 | cross out 256)
constexpr int DS_LDS = DS_BIAS + 1536 * 4;            // 156160
constexpr int DS_TN = 0, DS_TNP = DS_TILE, DS_QI = 2 * DS_TILE, DS_KI = DS_QI + DS_H * DS_IMGQ, DS_VI = DS_KI + DS_H * DS_IMGQ,
              DS_T1 = DS_VI + DS_H * DS_IMGQ;        // ... DS_T1 + DS_TILE = 100608 <= DS_R0
constexpr int DS_T2 = 0, DS_T2N = DS_TILE, DS_HT = 2 * DS_TILE;                          // FFN phase: + 2 * 32 * DS_HP * 2 = 100352 <= DS_R0
static_assert(DS_T1 + DS_TILE <= DS_R0 && DS_HT + 2 * 32 * DS_HP * 2 <= DS_R0 && DS_H * DS_IMGQ <= DS_TILE, "LDS map");

struct DecLayerArgs {
  const bf16_t* tgt; const bf16_t* qpos;             // [B*Q][256]
  const bf16_t* kc; long ldk; const bf16_t* vc; long ldv;     // projected memory keys / values [B*S][>= 256]
  const uint8_t* kpm;                                // [B][S] or null
  const float* amask;                                // additive self-attention mask [Q][Q] or null
  const u32x4* s_win; const float* s_bin; const u32x4* s_wo; const float* s_bo;      // self-attention (fragment-major weights)
  const u32x4* c_wq; const float* c_bq; const u32x4* c_wo; const float* c_bo;        // cross-attention: query rows of in_proj, out_proj
  const u32x4* w1; const float* b1; const u32x4* w2; const float* b2;
  const float* g1; const float* be1; const float* g2; const float* be2; const float* g3; const float* be3;
  bf16_t* out;                                       // layer output [B*Q][256]
  bf16_t* t1;                                        // always written (the cross out-proj re-reads it as its residual)
  // training by-products (all or none)
  bf16_t* tn; bf16_t* tnp; float* m1; float* r1; bf16_t* qk_s; bf16_t* v_s; bf16_t* ctx_s; float* lse_s;
  bf16_t* t1np; float* m2; float* r2; bf16_t* q_c; bf16_t* ctx_c; float* lse_c;
  bf16_t* t2; float* m3; float* r3; bf16_t* t2n; bf16_t* h;
  int B, Q, S, FF;
  float scale, drop_p;
  uint32_t thresh, seed[6];                          // self attention, self out-proj, cross attention, cross out-proj, hidden, FFN output
  const uint32_t* seed_ptr;
  int dbg;                                           // developer builds: phase ablation (WRONG results); 0 in the product library
};

// attention of the slab's queries (lane & 31 <-> query) against NT key tiles of one head; Q / K / V as [rows][32] bf16 images with
// 64-byte rows.  Returns the normalised context in C layout (lane & 31 <-> head dim, register r <-> query crow(r, hf)) and the
// log-sum-exp of this lane's query.  The body of attn_fwd_mfma_kernel.
template <bool AMASK>
__device__ __forceinline__ void slab_attention(f32x16& oacc, float& lse_out, const unsigned char* Qh, const unsigned char* Kh,
                                               const unsigned char* Vh, const float* Kb, float* strip, int NT, int Lq, int Lk,
                                               const float* __restrict__ amask, float scale, uint32_t thresh, float inv_keep,
                                               uint32_t sd, uint64_t rowbase, int lane) {
  const int n = lane & 31, hf = lane >> 5;
  const bf16x8 qf0 = frag_rows(Qh, 0, 0, lane), qf1 = frag_rows(Qh, 0, 1, lane);
  auto score_tile = [&](int kt, f32x16& st) {
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kh, kt * 32, 0, lane), qf0, st, 0, 0, 0);
    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kh, kt * 32, 1, lane), qf1, st, 0, 0, 0);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int key0 = kt * 32 + 8 * g4 + 4 * hf;
      const float4 kb = *reinterpret_cast<const float4*>(Kb + key0);
      const float kbv[4] = {kb.x, kb.y, kb.z, kb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float sc = st[4 * g4 + e] * scale + kbv[e];
        if (AMASK) { if (n < Lq && key0 + e < Lk) sc += amask[(long)n * Lk + key0 + e]; }
        st[4 * g4 + e] = sc;
      }
    }
  };
  float m = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < NT; ++kt) {
    f32x16 st;
    score_tile(kt, st);
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, st[r]);
  }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  const float ms = m > -INFINITY ? m : 0.f;
  const uint32_t d_hi = (uint32_t)(rowbase >> 33), d_inner = drop_inner(sd, d_hi);
  float sum = 0.f;
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
      for (int s4 = 0; s4 < 2; ++s4) {
        uint32_t keep = 0xfu;
        if (thresh) keep = drop_keep4(d_inner, d_hi, sd, rowbase + (kt * 32 + crow(8 * u + 4 * s4, hf)), thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p = __expf(st[8 * u + 4 * s4 + e] - ms);
          sum += p;
          pv[4 * s4 + e] = (keep >> e & 1u) ? (thresh ? p * inv_keep : p) : 0.f;
        }
      }
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(pv), frag_cols_tr(Vh, kt * 32 + 16 * u, lane), oacc, 0, 0, 0);
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  lse_out = m + __logf(sum);
  if (hf == 0) strip[lane] = 1.f / sum;
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const float4 iv = *reinterpret_cast<const float4*>(strip + 8 * g4 + 4 * hf);
    oacc[4 * g4 + 0] *= iv.x; oacc[4 * g4 + 1] *= iv.y; oacc[4 * g4 + 2] *= iv.z; oacc[4 * g4 + 3] *= iv.w;
  }
  __builtin_amdgcn_wave_barrier();
}

// epilogue helper: the 4 x 4 values of output tile `tile` a lane holds -> (+ bias) -> f(g4, feature of element 0, v[4])
template <class F>
__device__ __forceinline__ void tile_epilogue(const f32x16& acc, const float4 (&bias)[4], int tile, int hf, F f) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    float v4[4] = {acc[4 * g4 + 0] + bias[g4].x, acc[4 * g4 + 1] + bias[g4].y, acc[4 * g4 + 2] + bias[g4].z, acc[4 * g4 + 3] + bias[g4].w};
    f(g4, tile * 32 + 8 * g4 + 4 * hf, v4);
  }
}

template <bool TRAIN, bool AMASK>
__global__ __launch_bounds__(512) void dec_layer_kernel(const DecLayerArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* TN = reinterpret_cast<bf16_t*>(smem + DS_TN);
  bf16_t* TNP = reinterpret_cast<bf16_t*>(smem + DS_TNP);
  unsigned char* QIs = smem + DS_QI;
  unsigned char* KIs = smem + DS_KI;
  unsigned char* VIs = smem + DS_VI;
  bf16_t* T1 = reinterpret_cast<bf16_t*>(smem + DS_T1);
  float* Kb = reinterpret_cast<float*>(smem + DS_KB);
  float* Rs = reinterpret_cast<float*>(smem + DS_RS);
  bf16_t* CTX = reinterpret_cast<bf16_t*>(smem + DS_R2);            // context tile; the cross-attention q images alias it
  unsigned char* QIc = smem + DS_R2;
  unsigned char* KIc = smem;
  unsigned char* VIc = smem + DS_H * DS_IMGK;
  bf16_t* T2 = reinterpret_cast<bf16_t*>(smem + DS_T2);
  bf16_t* T2N = reinterpret_cast<bf16_t*>(smem + DS_T2N);
  bf16_t* HT = reinterpret_cast<bf16_t*>(smem + DS_HT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const int b = blockIdx.x, Q = a.Q, S = a.S, FF = a.FF;
  const int nvalid = Q;
  const long row0 = (long)b * Q;
  const float inv_keep = a.thresh ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t sd_off = a.seed_ptr ? *a.seed_ptr : 0u;
  const long ts256 = 64L * (DS_D / 16), ts_ff = 64L * (FF / 16);
#ifdef SEDT_DEV
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int nchunk = (dbg & 2) ? 0 : FF / 512;
  slab::u32x4 wa[8], wb[8];
  // the weight stream starts now: self-attention q | k tiles w, w + 8 (= head w)
  slab::load_chunk<2>(wa, a.s_win + (long)wave * ts256, 8 * ts256, 0, lane);
  slab::issue_fence();

  // ---- P1: LayerNorm1 (+ query position) -> TN, TNP; key bias of the self-attention (keys >= Q are padding); the projection
  // biases into LDS (an epilogue that loaded them from memory would wait behind the weight chunks prefetched for the next GEMM)
  float* BIAS = reinterpret_cast<float*>(smem + DS_BIAS);
  if (tid < 32) Kb[tid] = tid < Q ? 0.f : -INFINITY;
  for (int i = tid; i < 1536; i += 512)
    BIAS[i] = i < 768 ? a.s_bin[i] : i < 1024 ? a.s_bo[i - 768] : i < 1280 ? a.c_bq[i - 1024] : a.c_bo[i - 1280];
  VecT<bf16_t, 4> pin[4];                                           // this wave's rows of the query position embedding (LayerNorm1 and 2)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wave * 4 + i;
      pin[i] = r < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(a.qpos + (row0 + r) * DS_D + lane * 4) : VecT<bf16_t, 4>{};
    }
    slab::issue_fence();
    slab::slab_layernorm(
        wave, lane, nvalid, a.g1, a.be1, [&](int r) { return a.tgt + (row0 + r) * DS_D; },
        [&](int r, const float* y, float mu, float rs) {
          VecT<bf16_t, 4> o, op;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o.v[e] = (bf16_t)y[e];
            op.v[e] = (bf16_t)(r < nvalid ? y[e] + (float)pin[r & 3].v[e] : 0.f);
          }
          *reinterpret_cast<VecT<bf16_t, 4>*>(TN + r * XP + lane * 4) = o;
          *reinterpret_cast<VecT<bf16_t, 4>*>(TNP + r * XP + lane * 4) = op;
          if (TRAIN && r < nvalid) {
            *reinterpret_cast<VecT<bf16_t, 4>*>(a.tn + (row0 + r) * DS_D + lane * 4) = o;
            *reinterpret_cast<VecT<bf16_t, 4>*>(a.tnp + (row0 + r) * DS_D + lane * 4) = op;
            if (lane == 0) { a.m1[row0 + r] = mu; a.r1[row0 + r] = rs; }
          }
        });
  }
  __syncthreads();

  // ---- P2: self-attention projections of head `wave` straight into its [32][32] images
  {
    f32x16 acc[2];
    slab::zero_acc(acc);
    const slab::u32x4* wv = a.s_win + (long)(16 + wave) * ts256;
    slab::wave_gemm<2, 16>(acc, TNP, XP, a.s_win + (long)wave * ts256, 8 * ts256, lane, wa, wb,
                           [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wv, 0, 0, lane); });
    auto to_image = [&](unsigned char* img, const f32x16& c, int btile, bf16_t* save, long lds, int col0) {
      float4 bias[4];
      slab::load_feat4(bias, BIAS, btile, hf);
      tile_epilogue(c, bias, 0, hf, [&](int g4, int f, const float* v4) {
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(n < nvalid ? v4[e] : 0.f);
        *reinterpret_cast<VecT<bf16_t, 4>*>(img + wave * DS_IMGQ + n * AROW + f * 2) = o;
        if (TRAIN && n < nvalid) *reinterpret_cast<VecT<bf16_t, 4>*>(save + (row0 + n) * lds + col0 + wave * 32 + f) = o;
      });
    };
    to_image(QIs, acc[0], wave, a.qk_s, 512, 0);
    to_image(KIs, acc[1], wave + 8, a.qk_s, 512, 256);
    slab::load_chunk<1>(wb, wv, 0, 8, lane);                          // (the q | k GEMM left v's chunk 0 in flight in `wa`)
    slab::issue_fence();
    f32x16 accv[1];
    slab::zero_acc(accv);
    const slab::u32x4* wo = a.s_wo + (long)wave * ts256;
    slab::wave_gemm_small(accv, TN, XP, lane, wa, wb, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wo, 0, 0, lane); },
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wo, 0, 8, lane); });
    to_image(VIs, accv[0], wave + 16, a.v_s, 256, 0);
  }
  // ---- P3: self-attention of head `wave` (its own images only: no workgroup barrier needed before)
  {
    f32x16 oacc = {};
    float lse = 0.f;
    const uint64_t rowbase = ((uint64_t)(b * DS_H + wave) * Q + n) * Q;
    if (!(dbg & 1))
    slab_attention<AMASK>(oacc, lse, QIs + wave * DS_IMGQ, KIs + wave * DS_IMGQ, VIs + wave * DS_IMGQ, Kb, Rs + wave * 32, 1, Q, Q, a.amask,
                          a.scale, a.thresh, inv_keep, a.seed[0] + sd_off, rowbase, lane);
    if (TRAIN && hf == 0 && n < nvalid) a.lse_s[((long)b * DS_H + wave) * Q + n] = lse;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = crow(r, hf);
      const bf16_t o = (bf16_t)(q < nvalid ? oacc[r] : 0.f);
      CTX[q * XP + wave * AD + n] = o;
      if (TRAIN && q < nvalid) a.ctx_s[(row0 + q) * DS_D + wave * AD + n] = o;
    }
  }
  __syncthreads();

  // ---- P4: t1 = tgt + dropout(ctx Wo^T + bo)
  {
    VecT<bf16_t, 4> res[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      res[g4] = n < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(a.tgt + (row0 + n) * DS_D + wave * 32 + 8 * g4 + 4 * hf) : VecT<bf16_t, 4>{};
    slab::issue_fence();
    f32x16 acc[1];
    slab::zero_acc(acc);
    const slab::u32x4* wq = a.c_wq + (long)wave * ts256;
    slab::wave_gemm_small(acc, CTX, XP, lane, wa, wb, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wq, 0, 0, lane); },
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wq, 0, 8, lane); });
    const uint32_t sd = a.seed[1] + sd_off;
    float4 bo4[4];
    slab::load_feat4(bo4, BIAS + 768, wave, hf);
    tile_epilogue(acc[0], bo4, wave, hf, [&](int g4, int f, const float* v4) {
      VecT<bf16_t, 4> o{};
      if (n < nvalid) {
        const uint64_t idx = (uint64_t)(row0 + n) * DS_D + f;
        uint32_t keep = 0xfu;
        if (a.thresh) keep = drop_keep4(slab::inner0(sd), 0u, sd, idx, a.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(((keep >> e & 1u) ? v4[e] * inv_keep : 0.f) + (float)res[g4].v[e]);
        *reinterpret_cast<VecT<bf16_t, 4>*>(a.t1 + (row0 + n) * DS_D + f) = o;
      }
      *reinterpret_cast<VecT<bf16_t, 4>*>(T1 + n * XP + f) = o;
    });
  }
  __syncthreads();

  // ---- P5: t1n = LayerNorm2(t1); the cross-attention query input t1n + qpos -> TNP
  {
    slab::slab_layernorm(
        wave, lane, nvalid, a.g2, a.be2, [&](int r) { return T1 + r * XP; },
        [&](int r, const float* y, float mu, float rs) {
          VecT<bf16_t, 4> op;
#pragma unroll
          for (int e = 0; e < 4; ++e) op.v[e] = (bf16_t)(r < nvalid ? y[e] + (float)pin[r & 3].v[e] : 0.f);
          *reinterpret_cast<VecT<bf16_t, 4>*>(TNP + r * XP + lane * 4) = op;
          if (TRAIN && r < nvalid) {
            *reinterpret_cast<VecT<bf16_t, 4>*>(a.t1np + (row0 + r) * DS_D + lane * 4) = op;
            if (lane == 0) { a.m2[row0 + r] = mu; a.r2[row0 + r] = rs; }
          }
        });
  }
  __syncthreads();

  // ---- P6: cross-attention query of head `wave` into its image (region R2: the self-attention context is consumed).  The memory's
  // K / V rows are requested first (registers) and written into the images once every wave is done with the tiles they overwrite.
  uint4 kreg[8], vreg[8];
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {                                   // 128 rows x 32 chunks of 16 bytes
    const int u = tid + q8 * 512, r = u >> 5, c = u & 31;
    kreg[q8] = r < S ? *reinterpret_cast<const uint4*>(a.kc + ((long)b * S + r) * a.ldk + c * 8) : make_uint4(0, 0, 0, 0);
  }
  slab::issue_fence();
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    const slab::u32x4* wo = a.c_wo + (long)wave * ts256;
    slab::wave_gemm_small(acc, TNP, XP, lane, wa, wb, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wo, 0, 0, lane); },
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wo, 0, 8, lane); });
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) {                                 // (the V rows only now: registers)
      const int u = tid + q8 * 512, r = u >> 5, c = u & 31;
      vreg[q8] = r < S ? *reinterpret_cast<const uint4*>(a.vc + ((long)b * S + r) * a.ldv + c * 8) : make_uint4(0, 0, 0, 0);
    }
    slab::issue_fence();
    __syncthreads();                                                 // every wave is done with TNP / T1 / CTX: the images may overwrite them
    float4 bq[4];
    slab::load_feat4(bq, BIAS + 1024, wave, hf);
    tile_epilogue(acc[0], bq, 0, hf, [&](int g4, int f, const float* v4) {
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(n < nvalid ? v4[e] : 0.f);
      *reinterpret_cast<VecT<bf16_t, 4>*>(QIc + wave * DS_IMGQ + n * AROW + f * 2) = o;
      if (TRAIN && n < nvalid) *reinterpret_cast<VecT<bf16_t, 4>*>(a.q_c + (row0 + n) * DS_D + wave * 32 + f) = o;
    });
    if (!(dbg & 4))
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) {
      const int u = tid + q8 * 512, r = u >> 5, c = u & 31, h = c >> 2, cc = c & 3;
      *reinterpret_cast<uint4*>(KIc + h * DS_IMGK + r * AROW + cc * 16) = kreg[q8];
      *reinterpret_cast<uint4*>(VIc + h * DS_IMGK + r * AROW + cc * 16) = vreg[q8];
    }
    stage_key_bias(Kb, a.kpm ? a.kpm + (long)b * S : nullptr, S, DS_LK, tid, 512);
  }
  __syncthreads();

  // ---- P7: cross-attention of head `wave` over the S memory tokens
  {
    f32x16 oacc = {};
    float lse = 0.f;
    const uint64_t rowbase = ((uint64_t)(b * DS_H + wave) * Q + n) * S;
    // (the context tile aliases the q images: every head must be done with its image before any wave writes its context
    // columns - the barrier below)
    if (!(dbg & 1))
    slab_attention<false>(oacc, lse, QIc + wave * DS_IMGQ, KIc + wave * DS_IMGK, VIc + wave * DS_IMGK, Kb, Rs + wave * 32, (S + 31) >> 5, Q, S,
                          nullptr, a.scale, a.thresh, inv_keep, a.seed[2] + sd_off, rowbase, lane);
    if (TRAIN && hf == 0 && n < nvalid) a.lse_c[((long)b * DS_H + wave) * Q + n] = lse;
    __syncthreads();                                                 // all heads have read their q images
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = crow(r, hf);
      const bf16_t o = (bf16_t)(q < nvalid ? oacc[r] : 0.f);
      CTX[q * XP + wave * AD + n] = o;
      if (TRAIN && q < nvalid) a.ctx_c[(row0 + q) * DS_D + wave * AD + n] = o;
    }
  }
  __syncthreads();                                                   // K / V images dead from here on

  // ---- P8: t2 = t1 + dropout(ctx Wo'^T + bo')
  {
    VecT<bf16_t, 4> res[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      res[g4] = n < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(a.t1 + (row0 + n) * DS_D + wave * 32 + 8 * g4 + 4 * hf) : VecT<bf16_t, 4>{};
    slab::issue_fence();
    f32x16 acc[1];
    slab::zero_acc(acc);
    slab::wave_gemm_small(acc, CTX, XP, lane, wa, wb,
                          [&](slab::u32x4(&d)[8]) { slab::load_chunk<2>(d, a.w1 + (long)(2 * wave) * ts256, ts256, 0, lane); },
                          [&](slab::u32x4(&)[8]) {});
    const uint32_t sd = a.seed[3] + sd_off;
    float4 bo4[4];
    slab::load_feat4(bo4, BIAS + 1280, wave, hf);
    tile_epilogue(acc[0], bo4, wave, hf, [&](int g4, int f, const float* v4) {
      VecT<bf16_t, 4> o{};
      if (n < nvalid) {
        const uint64_t idx = (uint64_t)(row0 + n) * DS_D + f;
        uint32_t keep = 0xfu;
        if (a.thresh) keep = drop_keep4(slab::inner0(sd), 0u, sd, idx, a.thresh);
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(((keep >> e & 1u) ? v4[e] * inv_keep : 0.f) + (float)res[g4].v[e]);
        if (TRAIN) *reinterpret_cast<VecT<bf16_t, 4>*>(a.t2 + (row0 + n) * DS_D + f) = o;
      }
      *reinterpret_cast<VecT<bf16_t, 4>*>(T2 + n * XP + f) = o;
    });
  }
  __syncthreads();

  // ---- P9: t2n = LayerNorm3(t2)
  slab::slab_layernorm(
      wave, lane, nvalid, a.g3, a.be3, [&](int r) { return T2 + r * XP; },
      [&](int r, const float* y, float mu, float rs) {
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)y[e];
        *reinterpret_cast<VecT<bf16_t, 4>*>(T2N + r * XP + lane * 4) = o;
        if (TRAIN && r < nvalid) {
          *reinterpret_cast<VecT<bf16_t, 4>*>(a.t2n + (row0 + r) * DS_D + lane * 4) = o;
          if (lane == 0) { a.m3[row0 + r] = mu; a.r3[row0 + r] = rs; }
        }
      });
  __syncthreads();

  // ---- P10: the FFN pair, hidden in chunks of 512 through LDS (double-buffered); wave w owns output tile w of linear2
  f32x16 acc2[1];
  slab::zero_acc(acc2);
  const uint32_t sd_h = a.seed[4] + sd_off, sd_f = a.seed[5] + sd_off;
  for (int c = 0; c < nchunk; ++c) {
    bf16_t* Hc = HT + (c & 1) * 32 * DS_HP;
    {
      f32x16 acc[2];
      slab::zero_acc(acc);
      const int t0 = c * 16 + 2 * wave;
      const slab::u32x4* w2c = a.w2 + (long)wave * ts_ff + (long)c * 32 * 64;
      float4 b1r[2][4];
      slab::load_feat4(b1r[0], a.b1, t0, hf);
      slab::load_feat4(b1r[1], a.b1, t0 + 1, hf);
      slab::issue_fence();
      slab::wave_gemm<2, 16>(acc, T2N, XP, a.w1 + (long)t0 * ts256, ts256, lane, wa, wb,
                             [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w2c, 0, 0, lane); });
#pragma unroll
      for (int t = 0; t < 2; ++t)
        tile_epilogue(acc[t], b1r[t], t0 + t, hf, [&](int g4, int f, const float* v4) {
          const uint64_t idx = (uint64_t)(row0 + n) * FF + f;
          uint32_t keep = 0xfu;
          if (a.thresh) keep = drop_keep4(slab::inner0(sd_h), 0u, sd_h, idx, a.thresh);
          VecT<bf16_t, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((n < nvalid && (keep >> e & 1u)) ? fmaxf(v4[e], 0.f) * inv_keep : 0.f);
          *reinterpret_cast<VecT<bf16_t, 4>*>(Hc + n * DS_HP + (f - c * 512)) = o;
        });
    }
    __syncthreads();
    if (TRAIN) slab::tile_to_global(Hc, DS_HP, a.h + row0 * FF + c * 512, FF, nvalid, 512, tid, 512);
    slab::wave_gemm<1, 32>(acc2, Hc, DS_HP, a.w2 + (long)wave * ts_ff + (long)c * 32 * 64, 0, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (c + 1 < nchunk) slab::load_chunk<2>(d, a.w1 + (long)((c + 1) * 16 + 2 * wave) * ts256, ts256, 0, lane);
    });
  }
  // ---- out = t2 + dropout(acc2 + b2): straight from the lanes (32 rows: no staging needed; the weight stream has ended, so the
  // bias load waits behind nothing)
  float4 b2r[4];
  slab::load_feat4(b2r, a.b2, wave, hf);
  tile_epilogue(acc2[0], b2r, wave, hf, [&](int g4, int f, const float* v4) {
    if (n >= nvalid) return;
    const uint64_t idx = (uint64_t)(row0 + n) * DS_D + f;
    uint32_t keep = 0xfu;
    if (a.thresh) keep = drop_keep4(slab::inner0(sd_f), 0u, sd_f, idx, a.thresh);
    const VecT<bf16_t, 4> xr = *reinterpret_cast<const VecT<bf16_t, 4>*>(T2 + n * XP + f);
    VecT<bf16_t, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(((keep >> e & 1u) ? v4[e] * inv_keep : 0.f) + (float)xr.v[e]);
    *reinterpret_cast<VecT<bf16_t, 4>*>(a.out + (row0 + n) * DS_D + f) = o;
  });
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_decoder_slab_ok(int D, int H, int Q, int S, int FF, int dtype) {
  return dtype == SEDT_BF16 && D == DS_D && H == DS_H && Q >= 1 && Q <= 32 && S >= 1 && S <= DS_LK && FF >= 512 && FF % 512 == 0;
}

extern "C" int sedt_decoder_layer_fwd(const SedtDecLayer* p, void* stream) {
  SEDT_REQUIRE(p != nullptr, "decoder_layer_fwd: null arguments");
  SEDT_REQUIRE(sedt_decoder_slab_ok(DS_D, DS_H, p->Q, p->S, p->FF, SEDT_BF16) && p->B >= 1, "decoder_layer_fwd: Q = %d / S = %d / FF = %d outside the envelope",
               p->Q, p->S, p->FF);
  SEDT_REQUIRE(p->tgt && p->qpos && p->kc && p->vc && p->s_win && p->s_bin && p->s_wo && p->s_bo && p->c_wq && p->c_bq && p->c_wo && p->c_bo &&
                   p->w1 && p->b1 && p->w2 && p->b2 && p->g1 && p->be1 && p->g2 && p->be2 && p->g3 && p->be3 && p->out && p->t1,
               "decoder_layer_fwd: null pointer");
  SEDT_REQUIRE(p->ldk >= 256 && p->ldv >= 256 && p->ldk % 8 == 0 && p->ldv % 8 == 0, "decoder_layer_fwd: bad K / V row strides");
  SEDT_REQUIRE(p->drop_p >= 0.f && p->drop_p < 1.f, "decoder_layer_fwd: drop_p out of range");
  const bool train = p->tn != nullptr;
  SEDT_REQUIRE(!train || (p->tnp && p->m1 && p->r1 && p->qk_s && p->v_s && p->ctx_s && p->lse_s && p->t1np && p->m2 && p->r2 && p->q_c &&
                          p->ctx_c && p->lse_c && p->t2 && p->m3 && p->r3 && p->t2n && p->h),
               "decoder_layer_fwd: the training by-products come all or none");
  DecLayerArgs a;
  a.tgt = (const bf16_t*)p->tgt; a.qpos = (const bf16_t*)p->qpos;
  a.kc = (const bf16_t*)p->kc; a.ldk = p->ldk; a.vc = (const bf16_t*)p->vc; a.ldv = p->ldv;
  a.kpm = p->kpm; a.amask = p->amask;
  a.s_win = (const u32x4*)p->s_win; a.s_bin = p->s_bin; a.s_wo = (const u32x4*)p->s_wo; a.s_bo = p->s_bo;
  a.c_wq = (const u32x4*)p->c_wq; a.c_bq = p->c_bq; a.c_wo = (const u32x4*)p->c_wo; a.c_bo = p->c_bo;
  a.w1 = (const u32x4*)p->w1; a.b1 = p->b1; a.w2 = (const u32x4*)p->w2; a.b2 = p->b2;
  a.g1 = p->g1; a.be1 = p->be1; a.g2 = p->g2; a.be2 = p->be2; a.g3 = p->g3; a.be3 = p->be3;
  a.out = (bf16_t*)p->out; a.t1 = (bf16_t*)p->t1;
  a.tn = (bf16_t*)p->tn; a.tnp = (bf16_t*)p->tnp; a.m1 = p->m1; a.r1 = p->r1; a.qk_s = (bf16_t*)p->qk_s; a.v_s = (bf16_t*)p->v_s;
  a.ctx_s = (bf16_t*)p->ctx_s; a.lse_s = p->lse_s; a.t1np = (bf16_t*)p->t1np; a.m2 = p->m2; a.r2 = p->r2; a.q_c = (bf16_t*)p->q_c;
  a.ctx_c = (bf16_t*)p->ctx_c; a.lse_c = p->lse_c; a.t2 = (bf16_t*)p->t2; a.m3 = p->m3; a.r3 = p->r3; a.t2n = (bf16_t*)p->t2n; a.h = (bf16_t*)p->h;
  a.B = p->B; a.Q = p->Q; a.S = p->S; a.FF = p->FF;
  a.scale = 0.17677669529663687f;
  a.drop_p = p->drop_p;
  a.thresh = p->drop_p > 0.f ? drop_threshold(p->drop_p) : 0u;
  for (int i = 0; i < 6; ++i) a.seed[i] = p->seed[i];
  a.seed_ptr = p->seed_ptr;
  static const int dbg_env = dev_getenv("SEDT_SLAB_DBG") ? atoi(dev_getenv("SEDT_SLAB_DBG")) : 0;
  a.dbg = dbg_env;
  static bool attr = false;
  if (!attr) {
    const void* ks[4] = {reinterpret_cast<const void*>(dec_layer_kernel<true, true>), reinterpret_cast<const void*>(dec_layer_kernel<true, false>),
                         reinterpret_cast<const void*>(dec_layer_kernel<false, true>), reinterpret_cast<const void*>(dec_layer_kernel<false, false>)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS);
      if (e != hipSuccess) {
        set_error("decoder_layer_fwd: hipFuncSetAttribute(%d B LDS) failed: %s", DS_LDS, hipGetErrorString(e));
        return 1;
      }
    }
    attr = true;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool am = p->amask != nullptr;
  if (train && am) hipLaunchKernelGGL((dec_layer_kernel<true, true>), dim3(p->B), dim3(512), DS_LDS, st, a);
  else if (train) hipLaunchKernelGGL((dec_layer_kernel<true, false>), dim3(p->B), dim3(512), DS_LDS, st, a);
  else if (am) hipLaunchKernelGGL((dec_layer_kernel<false, true>), dim3(p->B), dim3(512), DS_LDS, st, a);
  else hipLaunchKernelGGL((dec_layer_kernel<false, false>), dim3(p->B), dim3(512), DS_LDS, st, a);
  return check_launch("decoder_layer_fwd");
}
