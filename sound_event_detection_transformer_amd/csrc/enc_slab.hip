// enc_slab.hip - the pre-norm encoder layer (reference sedt/transformer.py:192-204) in TWO launches on the x-stationary slab scheme
// (slab.h): a workgroup owns 32 tokens of one clip, its activations stay in LDS, only weights stream (L2 -> registers).
//
//   enc_qkv_kernel       xn = LayerNorm1(x); q | k = (xn + pos) Wqk^T + b; v = xn Wv^T + b               (393 KB of weights per slab)
//   enc_attn_ffn_kernel  ctx = softmax(q k^T / sqrt(32) + key padding) dropout . v  over the clip's keys (one head per wave);
//                        x1 = x + dropout(ctx Wo^T + bo); x1n = LayerNorm2(x1);
//                        x2 = x1 + dropout(dropout(relu(x1n W1^T + b1)) W2^T + b2)                       (2.23 MB of weights per slab)
//
// against LayerNorm | grouped Q|K / V GEMM | attention core | out-proj GEMM | LayerNorm | FFN GEMM | FFN GEMM = seven launches in
// which every GEMM moves both operands through LDS.  The hidden activation h (8192 x 2048 at B = 64) is written once (training: the
// weight-gradient GEMMs read it) and never read back in the forward.
// Rounding points are those of the unfused chain (bf16 at every tensor the chain materialises, f32 accumulation inside), the
// dropout decisions are the same counter hashes of (seed, element index): the unfused backward kernels consume the by-products.
// Envelope: bf16, d_model 256, 8 heads of 32, S <= 128, dim_feedforward a multiple of 512.
#include "slab.h"
#include "attn_frag.h"

namespace sedt {

using slab::u32x4;
using slab::XP;

constexpr int ES_D = 256, ES_H = 8, ES_LK = 128;
constexpr int ES_IMG = ES_LK * 64 + 32;              // bytes of one head's [128][32] bf16 image: the 32 spare bytes shift every head's rows
                                                     // by 8 banks (eight heads at stride 8192 would all start on bank 0: 8-way write conflicts)
constexpr int ES_HP = 512 + 8;                       // element pitch of a hidden chunk tile [32][512]
constexpr int ES_QP = 768 + 8;                       // element pitch of the q | k | v staging tile

struct EncQkvArgs {
  const bf16_t* x; const bf16_t* pos;
  const float* gamma; const float* beta;
  const u32x4* w_in;                                 // fragment-major in_proj_weight [768][256]
  const float* b_in;
  bf16_t* qk; bf16_t* v;                             // [B*S][512], [B*S][256]
  bf16_t* xn; bf16_t* xnp; float* mean; float* rstd; // training by-products (all or none)
  int B, S;
  // what the NEXT launch streams (sedt_encoder_attn_ffn_fwd's weights): touched here, one 128-byte line per load, so that every XCD's
  // L2 holds them when its 32 workgroups start streaming in lockstep (cold, each chunk was an HBM-latency miss for all of them at once)
  const uint32_t* pf[3]; int pf_lines[3];
};

// ---------------------------------------------------------------------------------------------------------------- enc_qkv
template <bool TRAIN>
__global__ __launch_bounds__(512) void enc_qkv_kernel(const EncQkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XN = reinterpret_cast<bf16_t*>(smem);                  // [32][XP]
  bf16_t* XNP = XN + 32 * XP;                                    // [32][XP]
  bf16_t* OUT = XNP + 32 * XP;                                   // [32][ES_QP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int SL = (a.S + 31) >> 5;
  const int b = blockIdx.x / SL, s0 = (blockIdx.x - b * SL) * 32;
  const int nvalid = min(32, a.S - s0);
  const long row0 = (long)b * a.S + s0;
  // the weight stream starts before anything else: chunk 0 of this wave's q|k tiles (tiles w and w + 8)
  const long tstride = 64L * (ES_D / 16);
  slab::u32x4 wa[8], wb[8];
  slab::load_chunk<2>(wa, a.w_in + (long)wave * tstride, 8 * tstride, 0, lane);
  slab::issue_fence();
  // ---- LayerNorm1 (+ pos) into the two operand tiles
  VecT<bf16_t, 4> pin[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 4 + i;
    pin[i] = r < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(a.pos + (row0 + r) * ES_D + lane * 4) : VecT<bf16_t, 4>{};
  }
  slab::issue_fence();
  slab::slab_layernorm(
      wave, lane, nvalid, a.gamma, a.beta, [&](int r) { return a.x + (row0 + r) * ES_D; },
      [&](int r, const float* y, float mu, float rs) {
        VecT<bf16_t, 4> o, op;
        float pv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) pv[e] = (float)pin[r & 3].v[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o.v[e] = (bf16_t)y[e];
          op.v[e] = (bf16_t)(r < nvalid ? y[e] + pv[e] : 0.f);      // (the sum is formed in f32 from the unrounded LayerNorm output, as ln_fwd_kernel)
        }
        *reinterpret_cast<VecT<bf16_t, 4>*>(XN + r * XP + lane * 4) = o;
        *reinterpret_cast<VecT<bf16_t, 4>*>(XNP + r * XP + lane * 4) = op;
        if (TRAIN && r < nvalid) {
          *reinterpret_cast<VecT<bf16_t, 4>*>(a.xn + (row0 + r) * ES_D + lane * 4) = o;
          *reinterpret_cast<VecT<bf16_t, 4>*>(a.xnp + (row0 + r) * ES_D + lane * 4) = op;
          if (lane == 0) { a.mean[row0 + r] = mu; a.rstd[row0 + r] = rs; }
        }
      });
  __syncthreads();
  // ---- 24 output tiles of 32 features: wave w takes q|k tiles w, w + 8 (operand xn + pos) and v tile 16 + w (operand xn)
  const int n = lane & 31, hf = lane >> 5;
  float4 bq[4], bk[4], bv[4];
  slab::load_feat4(bq, a.b_in, wave, hf);
  slab::load_feat4(bk, a.b_in, wave + 8, hf);
  slab::load_feat4(bv, a.b_in, wave + 16, hf);
  slab::issue_fence();
  auto store = [&](int tile, const f32x16& acc, const float4 (&bias)[4]) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = tile * 32 + 8 * g4 + 4 * hf;
      const float4 bb = bias[g4];
      VecT<bf16_t, 4> o;
      o.v[0] = (bf16_t)(acc[4 * g4 + 0] + bb.x); o.v[1] = (bf16_t)(acc[4 * g4 + 1] + bb.y);
      o.v[2] = (bf16_t)(acc[4 * g4 + 2] + bb.z); o.v[3] = (bf16_t)(acc[4 * g4 + 3] + bb.w);
      *reinterpret_cast<VecT<bf16_t, 4>*>(OUT + n * ES_QP + f) = o;
    }
  };
  uint32_t pf_acc = 0;
  {
    // workgroups go round-robin over the 8 XCDs: the workgroups of one XCD (blockIdx % 8 equal) share the lines between them
    const int slot = blockIdx.x >> 3, nslots = max(1, (int)(gridDim.x >> 3));
#pragma unroll
    for (int r = 0; r < 3; ++r)
      for (int j = slot * 512 + tid; j < a.pf_lines[r]; j += nslots * 512) pf_acc ^= a.pf[r][(long)j * 32];
  }
  {
    f32x16 acc[2];
    slab::zero_acc(acc);
    // tiles w and w + 8 (tile distance 8); the v tile's first chunk is in flight while the last q|k chunk is multiplied
    const slab::u32x4* wv = a.w_in + (long)(16 + wave) * tstride;
    slab::wave_gemm<2, 16>(acc, XNP, XP, a.w_in + (long)wave * tstride, 8 * tstride, lane, wa, wb,
                           [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, wv, 0, 0, lane); });
    store(wave, acc[0], bq);
    store(wave + 8, acc[1], bk);
    f32x16 accv[1];
    slab::zero_acc(accv);
    slab::wave_gemm<1, 16>(accv, XN, XP, wv, 0, lane, wa, wb, slab::NoNext());
    store(16 + wave, accv[0], bv);
  }
  __syncthreads();
  // ---- coalesced stores: q | k [32][512], v [32][256]
  slab::tile_to_global(OUT, ES_QP, a.qk + row0 * 512, 512, nvalid, 512, tid, 512);
  slab::tile_to_global(OUT + 512, ES_QP, a.v + row0 * 256, 256, nvalid, 256, tid, 512);
  if (pf_acc == 0x9e3779b9u && a.B < 0) a.qk[0] = (bf16_t)1.f;     // (never: keeps the touching loads alive)
}

// ---------------------------------------------------------------------------------------------------------------- enc_attn_ffn
struct EncAttnFfnArgs {
  const bf16_t* x;                                   // layer input (residual) [B*S][256]
  const bf16_t* qk; const bf16_t* v;                 // [B*S][512], [B*S][256]
  const uint8_t* kpm;                                // [B][S] or null
  const u32x4* w_o; const float* b_o;                // fragment-major out_proj [256][256]
  const float* gamma2; const float* beta2;
  const u32x4* w1; const float* b1;                  // fragment-major linear1 [FF][256]
  const u32x4* w2; const float* b2;                  // fragment-major linear2 [256][FF]
  bf16_t* x2;                                        // layer output [B*S][256]
  bf16_t* ctx; float* lse; bf16_t* x1; float* mean2; float* rstd2; bf16_t* x1n; bf16_t* h;       // training by-products
  int B, S, FF;
  float scale, drop_p;
  uint32_t thresh, seed_attn, seed_o, seed_h, seed_f;
  const uint32_t* seed_ptr;
  int dbg;                                           // developer builds: phase ablation (WRONG results); always 0 in the product library
};

template <bool TRAIN>
__global__ __launch_bounds__(512) void enc_attn_ffn_kernel(const EncAttnFfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // phase A / B (attention): K and V images of the clip's 8 heads, key bias, 1/sum strips, the context tile
  unsigned char* Ki = smem;                                        // 8 x [128][32] bf16
  unsigned char* Vi = Ki + ES_H * ES_IMG;
  float* Kb = reinterpret_cast<float*>(Vi + ES_H * ES_IMG);        // [128]
  float* Rs = Kb + ES_LK;                                          // [8][32]
  bf16_t* CTX = reinterpret_cast<bf16_t*>(Rs + 8 * 32);            // [32][XP]
  // phases C.. (K / V images dead): x1, x1n, two hidden chunk tiles
  bf16_t* X1 = reinterpret_cast<bf16_t*>(smem);                    // [32][XP]
  bf16_t* X1N = X1 + 32 * XP;                                      // [32][XP]
  bf16_t* HT = X1N + 32 * XP;                                      // 2 x [32][ES_HP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const int S = a.S;
  const int SL = (S + 31) >> 5;
  const int b = blockIdx.x / SL, s0 = (blockIdx.x - b * SL) * 32;
  const int nvalid = min(32, S - s0);
  const long row0 = (long)b * S + s0;
  const float inv_keep = a.thresh ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t sd_off = a.seed_ptr ? *a.seed_ptr : 0u;

#ifdef SEDT_DEV
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
  // ---- phase A: stage the clip's K and V as per-head images (rows >= S zero)
  // out-proj weights: chunk 0 of this wave's tile on its way before anything else (held across the attention phase)
  const long ts256 = 64L * (ES_D / 16);
  slab::u32x4 wa[8], wb[8];
  slab::load_chunk<1>(wa, a.w_o + (long)wave * ts256, 0, 0, lane);
  slab::issue_fence();
  if (!(dbg & 16)) {
    // all 16 loads of a thread in flight before the first LDS write (one at a time they cost 8 exposed L2 / HBM round trips)
    uint4 kreg[8], vreg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                  // 128 rows x 32 chunks of 16 bytes
      const int u = tid + q * 512, r = u >> 5, c = u & 31;
      kreg[q] = vreg[q] = make_uint4(0, 0, 0, 0);
      if (r < S) {
        kreg[q] = *reinterpret_cast<const uint4*>(a.qk + ((long)b * S + r) * 512 + 256 + c * 8);
        vreg[q] = *reinterpret_cast<const uint4*>(a.v + ((long)b * S + r) * 256 + c * 8);
      }
    }
    slab::issue_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = u & 31, h = c >> 2, cc = c & 3;
      *reinterpret_cast<uint4*>(Ki + h * ES_IMG + r * AROW + cc * 16) = kreg[q];
      *reinterpret_cast<uint4*>(Vi + h * ES_IMG + r * AROW + cc * 16) = vreg[q];
    }
  }
  stage_key_bias(Kb, a.kpm ? a.kpm + (long)b * S : nullptr, S, ES_LK, tid, 512);
  __syncthreads();

  // ---- phase B: attention of the slab's 32 queries, one head per wave (the body of attn_fwd_mfma_kernel)
  if (!(dbg & 1)) {
    const int h = wave;
    const unsigned char* Kh = Ki + h * ES_IMG;
    const unsigned char* Vh = Vi + h * ES_IMG;
    const int qi = s0 + n;                                         // this lane's query (token index in the clip)
    bf16x8 qf0, qf1;
    {
      uint4 z = make_uint4(0, 0, 0, 0), q0v = z, q1v = z;
      if (n < nvalid) {
        const bf16_t* qp = a.qk + (row0 + n) * 512 + h * AD + 8 * hf;
        q0v = *reinterpret_cast<const uint4*>(qp);
        q1v = *reinterpret_cast<const uint4*>(qp + 16);
      }
      qf0 = __builtin_bit_cast(bf16x8, q0v);
      qf1 = __builtin_bit_cast(bf16x8, q1v);
    }
    auto score_tile = [&](int kt, f32x16& st) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kh, kt * 32, 0, lane), qf0, st, 0, 0, 0);
      st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Kh, kt * 32, 1, lane), qf1, st, 0, 0, 0);
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 kb = *reinterpret_cast<const float4*>(Kb + kt * 32 + 8 * g4 + 4 * hf);
        st[4 * g4 + 0] = st[4 * g4 + 0] * a.scale + kb.x;
        st[4 * g4 + 1] = st[4 * g4 + 1] * a.scale + kb.y;
        st[4 * g4 + 2] = st[4 * g4 + 2] * a.scale + kb.z;
        st[4 * g4 + 3] = st[4 * g4 + 3] * a.scale + kb.w;
      }
    };
    float m = -INFINITY;
#pragma unroll 1
    for (int kt = 0; kt < ES_LK / 32; ++kt) {
      f32x16 st;
      score_tile(kt, st);
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, st[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float ms = m > -INFINITY ? m : 0.f;
    const uint32_t sd = a.seed_attn + sd_off;
    const uint64_t rowbase = ((uint64_t)(b * ES_H + h) * S + qi) * S;
    const uint32_t d_hi = (uint32_t)(rowbase >> 33), d_inner = drop_inner(sd, d_hi);
    float sum = 0.f;
    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
#pragma unroll 1
    for (int kt = 0; kt < ES_LK / 32; ++kt) {
      f32x16 st;
      score_tile(kt, st);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        float pv[8];
#pragma unroll
        for (int s4 = 0; s4 < 2; ++s4) {
          uint32_t keep = 0xfu;
          if (a.thresh) keep = drop_keep4(d_inner, d_hi, sd, rowbase + (kt * 32 + crow(8 * u + 4 * s4, hf)), a.thresh);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float p = __expf(st[8 * u + 4 * s4 + e] - ms);
            sum += p;
            pv[4 * s4 + e] = (keep >> e & 1u) ? (a.thresh ? p * inv_keep : p) : 0.f;
          }
        }
        oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(pv), frag_cols_tr(Vh, kt * 32 + 16 * u, lane), oacc, 0, 0, 0);
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    if (TRAIN && hf == 0 && n < nvalid) a.lse[((long)b * ES_H + h) * S + qi] = m + __logf(sum);
    float* strip = Rs + wave * 32;
    if (hf == 0) strip[lane] = 1.f / sum;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 iv = *reinterpret_cast<const float4*>(strip + 8 * g4 + 4 * hf);
      oacc[4 * g4 + 0] *= iv.x; oacc[4 * g4 + 1] *= iv.y; oacc[4 * g4 + 2] *= iv.z; oacc[4 * g4 + 3] *= iv.w;
    }
    __builtin_amdgcn_wave_barrier();
    // oacc: lane <-> head dim n, register r <-> query crow(r, hf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = crow(r, hf);
      CTX[q * XP + h * AD + n] = (bf16_t)(q < nvalid ? oacc[r] : 0.f);
    }
  }
  __syncthreads();                                                 // all heads done: K / V images are dead from here on
  if (TRAIN && !(dbg & 4)) slab::tile_to_global(CTX, XP, a.ctx + row0 * ES_D, ES_D, nvalid, ES_D, tid, 512);

  // ---- phase C: x1 = x + dropout(ctx Wo^T + bo): wave w computes feature tile w; linear1's first chunk follows in the stream
  const int FF = a.FF, nchunk = (dbg & 2) ? 0 : FF / 512;
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    float4 bo4[4];
    VecT<bf16_t, 4> xres[4];
    slab::load_feat4(bo4, a.b_o, wave, hf);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      xres[g4] = n < nvalid ? *reinterpret_cast<const VecT<bf16_t, 4>*>(a.x + (row0 + n) * ES_D + wave * 32 + 8 * g4 + 4 * hf) : VecT<bf16_t, 4>{};
    slab::issue_fence();
    slab::wave_gemm<1, 16>(acc, CTX, XP, a.w_o + (long)wave * ts256, 0, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (nchunk) slab::load_chunk<2>(d, a.w1 + (long)(2 * wave) * ts256, ts256, 0, lane);
    });
    const uint32_t sd = a.seed_o + sd_off;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      const float4 bb = bo4[g4];
      float v4[4] = {acc[0][4 * g4 + 0] + bb.x, acc[0][4 * g4 + 1] + bb.y, acc[0][4 * g4 + 2] + bb.z, acc[0][4 * g4 + 3] + bb.w};
      VecT<bf16_t, 4> o;
      if (n < nvalid) {
        const uint64_t idx = (uint64_t)(row0 + n) * ES_D + f;
        uint32_t keep = 0xfu;
        if (a.thresh) keep = drop_keep4(slab::inner0(sd), 0u, sd, idx, a.thresh);
        const VecT<bf16_t, 4> xr = xres[g4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(((keep >> e & 1u) ? v4[e] * inv_keep : 0.f) + (float)xr.v[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)0.f;
      }
      *reinterpret_cast<VecT<bf16_t, 4>*>(X1 + n * XP + f) = o;          // (X1 aliases the K images: dead since the barrier above)
    }
  }
  __syncthreads();
  if (TRAIN && !(dbg & 4)) slab::tile_to_global(X1, XP, a.x1 + row0 * ES_D, ES_D, nvalid, ES_D, tid, 512);

  // ---- phase D: x1n = LayerNorm2(x1)
  slab::slab_layernorm(
      wave, lane, nvalid, a.gamma2, a.beta2, [&](int r) { return X1 + r * XP; },
      [&](int r, const float* y, float mu, float rs) {
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)y[e];
        *reinterpret_cast<VecT<bf16_t, 4>*>(X1N + r * XP + lane * 4) = o;
        if (TRAIN && r < nvalid) {
          *reinterpret_cast<VecT<bf16_t, 4>*>(a.x1n + (row0 + r) * ES_D + lane * 4) = o;
          if (TRAIN && lane == 0) { a.mean2[row0 + r] = mu; a.rstd2[row0 + r] = rs; }
        }
      });
  __syncthreads();

  // ---- phase E: the FFN pair, hidden in chunks of 512 through LDS (double-buffered); wave w owns output tile w of linear2
  f32x16 acc2[1];
  slab::zero_acc(acc2);
  float4 b2r[4];
  slab::load_feat4(b2r, a.b2, wave, hf);
  slab::issue_fence();
  const long ts_ff = 64L * (FF / 16);                               // tile stride of linear2 [256][FF]
  const uint32_t sd_h = a.seed_h + sd_off, sd_f = a.seed_f + sd_off;
  for (int c = 0; c < nchunk; ++c) {
    bf16_t* Hc = HT + (c & 1) * 32 * ES_HP;
    {   // linear1 + ReLU + dropout: hidden tiles c * 16 + 2w, + 1
      f32x16 acc[2];
      slab::zero_acc(acc);
      const int t0 = c * 16 + 2 * wave;
      const slab::u32x4* w2c = a.w2 + (long)wave * ts_ff + (long)c * 32 * 64;
      float4 b1r[2][4];
      slab::load_feat4(b1r[0], a.b1, t0, hf);
      slab::load_feat4(b1r[1], a.b1, t0 + 1, hf);
      slab::issue_fence();
      slab::wave_gemm<2, 16>(acc, X1N, XP, a.w1 + (long)t0 * ts256, ts256, lane, wa, wb,
                             [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w2c, 0, 0, lane); });
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int f = (t0 + t) * 32 + 8 * g4 + 4 * hf;           // hidden feature
          const float4 bb = b1r[t][g4];
          float v4[4] = {acc[t][4 * g4 + 0] + bb.x, acc[t][4 * g4 + 1] + bb.y, acc[t][4 * g4 + 2] + bb.z, acc[t][4 * g4 + 3] + bb.w};
          const uint64_t idx = (uint64_t)(row0 + n) * FF + f;
          uint32_t keep = 0xfu;
          if (a.thresh) keep = drop_keep4(slab::inner0(sd_h), 0u, sd_h, idx, a.thresh);
          VecT<bf16_t, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((n < nvalid && (keep >> e & 1u)) ? fmaxf(v4[e], 0.f) * inv_keep : 0.f);
          *reinterpret_cast<VecT<bf16_t, 4>*>(Hc + n * ES_HP + (f - c * 512)) = o;
        }
    }
    __syncthreads();
    if (TRAIN && !(dbg & 4)) slab::tile_to_global(Hc, ES_HP, a.h + row0 * FF + c * 512, FF, nvalid, 512, tid, 512);
    if (dbg & 8) continue;                                         // (ablation: linear1 only)
    // linear2 partial sum over this chunk's 512 hidden features; the next chunk's linear1 weights follow in the stream
    slab::wave_gemm<1, 32>(acc2, Hc, ES_HP, a.w2 + (long)wave * ts_ff + (long)c * 32 * 64, 0, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (c + 1 < nchunk) slab::load_chunk<2>(d, a.w1 + (long)((c + 1) * 16 + 2 * wave) * ts256, ts256, 0, lane);
    });
  }
  // ---- x2 = x1 + dropout(acc2 + b2)
  __syncthreads();                                                 // (every wave is done reading X1N / the hidden tiles: X1N becomes the output tile)
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int f = wave * 32 + 8 * g4 + 4 * hf;
    const float4 bb = b2r[g4];
    float v4[4] = {acc2[0][4 * g4 + 0] + bb.x, acc2[0][4 * g4 + 1] + bb.y, acc2[0][4 * g4 + 2] + bb.z, acc2[0][4 * g4 + 3] + bb.w};
    const uint64_t idx = (uint64_t)(row0 + n) * ES_D + f;
    uint32_t keep = 0xfu;
    if (a.thresh) keep = drop_keep4(slab::inner0(sd_f), 0u, sd_f, idx, a.thresh);
    const VecT<bf16_t, 4> xr = *reinterpret_cast<const VecT<bf16_t, 4>*>(X1 + n * XP + f);
    VecT<bf16_t, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)(((keep >> e & 1u) ? v4[e] * inv_keep : 0.f) + (float)xr.v[e]);
    *reinterpret_cast<VecT<bf16_t, 4>*>(X1N + n * XP + f) = o;
  }
  __syncthreads();
  slab::tile_to_global(X1N, XP, a.x2 + row0 * ES_D, ES_D, nvalid, ES_D, tid, 512);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// LayerNorm backward of the slab rows [4 * wave, +4) (the arithmetic of ln_bwd_kernel): dy(row) -> the row's 256 bf16 gradient values
// in LDS, x / dres rows in global memory; out(row, lane, dx[4]) receives dx = rstd * (dy g - c1 - xhat c2) + dres.  The gamma / beta
// partial sums of the slab (sum over its rows of dy * xhat and of dy) are reduced over the 8 waves through `red` [8][512] and
// written to part[512] - one row of the [slabs][512] table the layer's reduce launch sums (ops.ReduceBatch.add_colsum).
template <class Dy, class Out>
__device__ __forceinline__ void slab_layernorm_bwd(int wave, int lane, int tid, int nvalid, long row0, const float* __restrict__ gamma,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const bf16_t* __restrict__ x, const bf16_t* __restrict__ dres, Dy dy, Out out,
                                                   float* red, float* __restrict__ part) {
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + lane * 4);
  const float g[4] = {g4.x, g4.y, g4.z, g4.w};
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 4 + i;
    if (row >= nvalid) continue;                               // (wave-uniform)
    const long base = (row0 + row) * 256 + lane * 4;
    const float mu = mean[row0 + row], rs = rstd[row0 + row];
    const VecT<bf16_t, 4> vdy = *reinterpret_cast<const VecT<bf16_t, 4>*>(dy(row) + lane * 4);
    const VecT<bf16_t, 4> vx = *reinterpret_cast<const VecT<bf16_t, 4>*>(x + base);
    const VecT<bf16_t, 4> vr = *reinterpret_cast<const VecT<bf16_t, 4>*>(dres + base);
    float xh[4], d[4], c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      d[e] = (float)vdy.v[e];
      xh[e] = ((float)vx.v[e] - mu) * rs;
      const float dgv = d[e] * g[e];
      c1 += dgv;
      c2 += dgv * xh[e];
      dg[e] += d[e] * xh[e];
      db[e] += d[e];
    }
    c1 = wave_sum(c1) * (1.f / 256.f);
    c2 = wave_sum(c2) * (1.f / 256.f);
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = rs * (d[e] * g[e] - c1 - xh[e] * c2) + (float)vr.v[e];
    out(row, o);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[wave * 512 + lane * 4 + e] = dg[e]; red[wave * 512 + 256 + lane * 4 + e] = db[e]; }
  __syncthreads();
  {
    const int c = tid;                                         // 512 threads <-> 512 columns
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[w * 512 + c];
    part[c] = s;
  }
}

struct EncFfnBwdArgs {
  const bf16_t* gx2;                                  // gradient wrt the layer output [B*S][256]
  const bf16_t* h;                                    // saved dropped ReLU output [B*S][FF]
  const bf16_t* x1; const float* mean2; const float* rstd2; const float* gamma2;
  const u32x4* w2t;                                   // fragment-major linear2^T: features = hidden, contraction 256
  const u32x4* w1t;                                   // fragment-major linear1^T: features = 256, contraction FF
  const u32x4* wot;                                   // fragment-major out_proj^T
  bf16_t* g2; bf16_t* gh; bf16_t* gx1; bf16_t* g1; bf16_t* gctx;
  float* ln_part;                                     // [slabs][512]
  int B, S, FF;
  float drop_p;
  uint32_t thresh, seed_f, seed_o;
  const uint32_t* seed_ptr;
};

// FFN backward + LayerNorm2 backward + out-proj input gradient of a 32-token slab:
//   g2 = dropout'(gx2); gh = (g2 W2) [h > 0] / (1 - p); g_x1n = gh W1; gx1 = LN2'(g_x1n) + gx2; g1 = dropout'(gx1); gctx = g1 Wo
// g2, gh, g1 are also what the weight-gradient GEMMs of linear2, linear1 and out_proj read; gctx feeds sedt_attention_bwd.
__global__ __launch_bounds__(512) void enc_ffn_bwd_kernel(const EncFfnBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* G2 = reinterpret_cast<bf16_t*>(smem);                    // [32][XP]   g2, later g1
  bf16_t* HM = G2 + 32 * XP;                                       // [32][ES_HP] h chunk, later g_x1n / the output staging tile
  bf16_t* GH = HM + 32 * ES_HP;                                    // 2 x [32][ES_HP]
  float* RED = reinterpret_cast<float*>(GH);                       // [8][512] (after the chunk loop)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const int S = a.S, FF = a.FF;
  const int SL = (S + 31) >> 5;
  const int b = blockIdx.x / SL, s0 = (blockIdx.x - b * SL) * 32;
  const int nvalid = min(32, S - s0);
  const long row0 = (long)b * S + s0;
  const float inv_keep = a.thresh ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t sd_off = a.seed_ptr ? *a.seed_ptr : 0u;
  const long ts256 = 64L * (ES_D / 16), ts_ff = 64L * (FF / 16);
  slab::u32x4 wa[8], wb[8];                                        // the weight stream starts now: linear2^T tiles 2w, 2w + 1, chunk 0
  slab::load_chunk<2>(wa, a.w2t + (long)(2 * wave) * ts256, ts256, 0, lane);
  slab::issue_fence();
  // ---- g2 = gx2 through the FFN-output dropout mask
  {
    const uint32_t sd = a.seed_f + sd_off;
    bf16x8 gin[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) gin[q][e] = (bf16_t)0.f;
      if (r < nvalid) gin[q] = *reinterpret_cast<const bf16x8*>(a.gx2 + (row0 + r) * ES_D + c);
    }
    slab::issue_fence();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)0.f;
      if (r < nvalid) {
        const long base = (row0 + r) * ES_D + c;
        const bf16x8 gv = gin[q];
        const uint32_t keep = a.thresh ? drop_keep8(sd, (uint64_t)base, a.thresh) : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (keep >> e & 1u) ? (a.thresh ? (bf16_t)((float)gv[e] * inv_keep) : gv[e]) : (bf16_t)0.f;
        if (a.g2) *reinterpret_cast<bf16x8*>(a.g2 + base) = o;
      }
      *reinterpret_cast<bf16x8*>(G2 + r * XP + c) = o;
    }
  }
  __syncthreads();
  f32x16 acc1[1];
  slab::zero_acc(acc1);
  const int nchunk = FF / 512;
  for (int c = 0; c < nchunk; ++c) {
    // this chunk of h on its way (coalesced) while the GEMM runs
    uint4 hreg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = tid + q * 512, r = u >> 6, cc = (u & 63) * 8;
      hreg[q] = r < nvalid ? *reinterpret_cast<const uint4*>(a.h + (row0 + r) * FF + c * 512 + cc) : make_uint4(0, 0, 0, 0);
    }
    slab::issue_fence();
    f32x16 acc[2];
    slab::zero_acc(acc);
    const int t0 = c * 16 + 2 * wave;
    const slab::u32x4* w1c = a.w1t + (long)wave * ts_ff + (long)c * 32 * 64;
    slab::wave_gemm<2, 16>(acc, G2, XP, a.w2t + (long)t0 * ts256, ts256, lane, wa, wb,
                           [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w1c, 0, 0, lane); });
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u = tid + q * 512, r = u >> 6, cc = (u & 63) * 8;
      *reinterpret_cast<uint4*>(HM + r * ES_HP + cc) = hreg[q];
    }
    __syncthreads();
    bf16_t* Gc = GH + (c & 1) * 32 * ES_HP;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int fl = (2 * wave + t) * 32 + 8 * g4 + 4 * hf;      // hidden feature inside the chunk
        const VecT<bf16_t, 4> hv = *reinterpret_cast<const VecT<bf16_t, 4>*>(HM + n * ES_HP + fl);
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((float)hv.v[e] > 0.f ? acc[t][4 * g4 + e] * inv_keep : 0.f);
        *reinterpret_cast<VecT<bf16_t, 4>*>(Gc + n * ES_HP + fl) = o;
      }
    __syncthreads();
    slab::tile_to_global(Gc, ES_HP, a.gh + row0 * FF + c * 512, FF, nvalid, 512, tid, 512);
    slab::wave_gemm<1, 32>(acc1, Gc, ES_HP, w1c, 0, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (c + 1 < nchunk) slab::load_chunk<2>(d, a.w2t + (long)((c + 1) * 16 + 2 * wave) * ts256, ts256, 0, lane);
      else slab::load_chunk<1>(d, a.wot + (long)wave * ts256, 0, 0, lane);          // the out-proj^T tile of the last phase
    });
  }
  // ---- g_x1n (bf16, as the per-op chain rounds it) into the HM tile
  bf16_t* GX = HM;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int f = wave * 32 + 8 * g4 + 4 * hf;
    VecT<bf16_t, 4> o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)acc1[0][4 * g4 + e];
    *reinterpret_cast<VecT<bf16_t, 4>*>(GX + n * XP + f) = o;
  }
  __syncthreads();                                                 // (also: every wave is done with the GH tiles -> RED may alias them)
  // ---- LayerNorm2 backward + residual -> gx1; g1 = gx1 through the out-proj dropout mask (tile G2, dead since the chunk loop)
  {
    const uint32_t sd = a.seed_o + sd_off;
    slab_layernorm_bwd(
        wave, lane, tid, nvalid, row0, a.gamma2, a.mean2, a.rstd2, a.x1, a.gx2, [&](int r) { return GX + r * XP; },
        [&](int r, const float* o) {
          const long base = (row0 + r) * ES_D + lane * 4;
          VecT<bf16_t, 4> ov, od;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.v[e] = (bf16_t)o[e];
          *reinterpret_cast<VecT<bf16_t, 4>*>(a.gx1 + base) = ov;
          uint32_t keep = 0xfu;
          if (a.thresh) keep = drop_keep4(slab::inner0(sd), 0u, sd, (uint64_t)base, a.thresh);
#pragma unroll
          for (int e = 0; e < 4; ++e) od.v[e] = (keep >> e & 1u) ? (a.thresh ? (bf16_t)((float)ov.v[e] * inv_keep) : ov.v[e]) : (bf16_t)0.f;
          if (a.g1) *reinterpret_cast<VecT<bf16_t, 4>*>(a.g1 + base) = od;
          *reinterpret_cast<VecT<bf16_t, 4>*>(G2 + r * XP + lane * 4) = od;
        },
        RED, a.ln_part + (long)blockIdx.x * 512);
    // rows beyond the clip: zero operand rows for the GEMM below
    for (int r = nvalid + wave; r < 32; r += 8) *reinterpret_cast<VecT<bf16_t, 4>*>(G2 + r * XP + lane * 4) = VecT<bf16_t, 4>{};
  }
  __syncthreads();
  // ---- gctx = g1 Wo
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    slab::wave_gemm<1, 16>(acc, G2, XP, a.wot + (long)wave * ts256, 0, lane, wa, wb, slab::NoNext());
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)acc[0][4 * g4 + e];
      *reinterpret_cast<VecT<bf16_t, 4>*>(GX + n * XP + f) = o;      // (GX: its last readers passed the barrier inside slab_layernorm_bwd)
    }
  }
  __syncthreads();
  slab::tile_to_global(GX, XP, a.gctx + row0 * ES_D, ES_D, nvalid, ES_D, tid, 512);
}

struct EncQkvBwdArgs {
  const bf16_t* dqk; const bf16_t* dv;                // [B*S][512], [B*S][256]
  const bf16_t* x; const float* mean1; const float* rstd1; const float* gamma1;
  const bf16_t* gx1;                                  // residual gradient
  const u32x4* wint;                                  // fragment-major in_proj^T: features = 256, contraction 768
  bf16_t* gx;
  float* ln_part;
  int B, S;
};

// in-projection input gradient + LayerNorm1 backward of a slab: g_xn = dq|dk Wqk + dv Wv (one K = 768 contraction: both land on xn,
// the position encoding is a constant); gx = LN1'(g_xn) + gx1
__global__ __launch_bounds__(512) void enc_qkv_bwd_kernel(const EncQkvBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* DQ = reinterpret_cast<bf16_t*>(smem);                    // [32][ES_QP]
  bf16_t* GX = DQ + 32 * ES_QP;                                    // [32][XP]
  float* RED = reinterpret_cast<float*>(DQ);                       // [8][512] (after the GEMM)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, hf = lane >> 5;
  const int S = a.S;
  const int SL = (S + 31) >> 5;
  const int b = blockIdx.x / SL, s0 = (blockIdx.x - b * SL) * 32;
  const int nvalid = min(32, S - s0);
  const long row0 = (long)b * S + s0;
  slab::u32x4 wa[8], wb[8];
  slab::load_chunk<1>(wa, a.wint + (long)wave * 64L * 48, 0, 0, lane);
  slab::issue_fence();
  {
    uint4 v[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {                                  // 96 chunks of 8 per row: dq | dk | dv
      const int u = tid + q * 512, r = u / 96, c = (u - r * 96) * 8;
      v[q] = make_uint4(0, 0, 0, 0);
      if (r < nvalid) v[q] = c < 512 ? *reinterpret_cast<const uint4*>(a.dqk + (row0 + r) * 512 + c)
                                     : *reinterpret_cast<const uint4*>(a.dv + (row0 + r) * 256 + (c - 512));
    }
    slab::issue_fence();
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const int u = tid + q * 512, r = u / 96, c = (u - r * 96) * 8;
      *reinterpret_cast<uint4*>(DQ + r * ES_QP + c) = v[q];
    }
  }
  __syncthreads();
  {
    f32x16 acc[1];
    slab::zero_acc(acc);
    slab::wave_gemm<1, 48>(acc, DQ, ES_QP, a.wint + (long)wave * 64L * 48, 0, lane, wa, wb, slab::NoNext());
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int f = wave * 32 + 8 * g4 + 4 * hf;
      VecT<bf16_t, 4> o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)acc[0][4 * g4 + e];
      *reinterpret_cast<VecT<bf16_t, 4>*>(GX + n * XP + f) = o;
    }
  }
  __syncthreads();
  slab_layernorm_bwd(
      wave, lane, tid, nvalid, row0, a.gamma1, a.mean1, a.rstd1, a.x, a.gx1, [&](int r) { return GX + r * XP; },
      [&](int r, const float* o) {
        VecT<bf16_t, 4> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov.v[e] = (bf16_t)o[e];
        *reinterpret_cast<VecT<bf16_t, 4>*>(a.gx + (row0 + r) * ES_D + lane * 4) = ov;
      },
      RED, a.ln_part + (long)blockIdx.x * 512);
}

// ---------------------------------------------------------------------------------------------------------------- weight packing
__global__ __launch_bounds__(256) void pack_frag_kernel(const SedtFragJob* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SedtFragJob j = jobs[lo];
  const int tb = (int)blockIdx.x - j.blk0;
  const int kt = j.K / 32, n0 = (tb / kt) * 32, k0 = (tb % kt) * 32;
  const int tid = threadIdx.x;
  if (j.src_bf16) {                                               // the source is a bf16 matrix already (a packed conv operand)
    const bf16_t* src = reinterpret_cast<const bf16_t*>(j.w);
    for (int u = tid; u < 32 * 8; u += 256) {
      const int r = u >> 3, c = (u & 7) * 4;
      const VecT<bf16_t, 4> v = *reinterpret_cast<const VecT<bf16_t, 4>*>(src + (long)(n0 + r) * j.K + k0 + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[r][c + e] = (float)v.v[e];
    }
  } else {
    for (int u = tid; u < 32 * 8; u += 256) {                     // 32 rows x 8 float4
      const int r = u >> 3, c = (u & 7) * 4;
      const float4 v = *reinterpret_cast<const float4*>(j.w + (long)(n0 + r) * j.K + k0 + c);
      tile[r][c] = v.x; tile[r][c + 1] = v.y; tile[r][c + 2] = v.z; tile[r][c + 3] = v.w;
    }
  }
  __syncthreads();
  const int which = tid >> 7, t = tid & 127, sub = t >> 6, l = t & 63, m = l & 31, hf = l >> 5;
  bf16x8 o;
  if (which == 0) {
    if (!j.wf) return;
    // forward: feature = n0 + m, k = k0 + sub * 16 + hf * 8 + e
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)tile[m][sub * 16 + hf * 8 + e];
    const long blk = (long)(n0 / 32) * (j.K / 16) + (k0 / 16 + sub);
    reinterpret_cast<bf16x8*>(j.wf)[blk * 64 + l] = o;
  } else {
    if (!j.wb) return;
    // transposed: feature = k0 + m, contraction index = n0 + sub * 16 + hf * 8 + e
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)tile[sub * 16 + hf * 8 + e][m];
    const long blk = (long)(k0 / 32) * (j.N / 16) + (n0 / 16 + sub);
    reinterpret_cast<bf16x8*>(j.wb)[blk * 64 + l] = o;
  }
}

template <typename K>
static int set_lds(K kern, size_t lds, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute(%zu B LDS) failed: %s", what, lds, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_pack_frag(const SedtFragJob* jobs, int njobs, int nblocks, void* stream) {
  SEDT_REQUIRE(jobs && njobs > 0 && nblocks > 0, "pack_frag: bad arguments");
  hipLaunchKernelGGL(pack_frag_kernel, dim3(nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), jobs, njobs);
  return check_launch("pack_frag");
}

extern "C" int sedt_encoder_slab_ok(int D, int H, int S, int FF, int dtype) {
  return dtype == SEDT_BF16 && D == ES_D && H == ES_H && S >= 1 && S <= ES_LK && FF >= 512 && FF % 512 == 0;
}

extern "C" int sedt_encoder_qkv_fwd(const void* x, const void* pos, const float* gamma, const float* beta, const void* w_in_frag,
                                    const float* b_in, void* qk, void* v, void* xn, void* xnp, float* mean, float* rstd, int B, int S,
                                    const SedtPrefetch* pf, void* stream) {
  SEDT_REQUIRE(x && pos && gamma && beta && w_in_frag && b_in && qk && v, "encoder_qkv_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && S >= 1 && S <= ES_LK, "encoder_qkv_fwd: S = %d outside 1..%d", S, ES_LK);
  const bool train = xn != nullptr;
  SEDT_REQUIRE(!train || (xnp && mean && rstd), "encoder_qkv_fwd: the training by-products come all or none");
  EncQkvArgs a{(const bf16_t*)x, (const bf16_t*)pos, gamma, beta, (const u32x4*)w_in_frag, b_in, (bf16_t*)qk, (bf16_t*)v,
               (bf16_t*)xn, (bf16_t*)xnp, mean, rstd, B, S, {nullptr, nullptr, nullptr}, {0, 0, 0}};
  for (int r = 0; r < 3 && pf; ++r) {                            // the weights the NEXT launch streams: touched by this one
    a.pf[r] = (const uint32_t*)pf->ptr[r];
    a.pf_lines[r] = pf->ptr[r] ? (int)(pf->bytes[r] / 128) : 0;
  }
  constexpr size_t lds = (size_t)(2 * 32 * XP + 32 * ES_QP) * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    if (set_lds(enc_qkv_kernel<true>, lds, "encoder_qkv_fwd") || set_lds(enc_qkv_kernel<false>, lds, "encoder_qkv_fwd")) return 1;
    attr = true;
  }
  const int grid = B * ((S + 31) / 32);
  if (train) hipLaunchKernelGGL(enc_qkv_kernel<true>, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(enc_qkv_kernel<false>, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("encoder_qkv_fwd");
}

static int enc_attn_ffn_launch(const void* x, const void* qk, const void* v, const uint8_t* kpm, const void* w_o_frag,
                               const float* b_o, const float* gamma2, const float* beta2, const void* w1_frag,
                               const float* b1, const void* w2_frag, const float* b2, void* x2, void* ctx, float* lse,
                               void* x1, float* mean2, float* rstd2, void* x1n, void* h, int B, int S, int FF,
                               float drop_p, uint32_t seed_attn, uint32_t seed_o, uint32_t seed_h, uint32_t seed_f,
                               const uint32_t* seed_ptr, void* stream);

extern "C" int sedt_encoder_attn_ffn_fwd(const void* x, const void* qk, const void* v, const uint8_t* kpm, const void* w_o_frag,
                                         const float* b_o, const float* gamma2, const float* beta2, const void* w1_frag,
                                         const float* b1, const void* w2_frag, const float* b2, void* x2, void* ctx, float* lse,
                                         void* x1, float* mean2, float* rstd2, void* x1n, void* h, int B, int S, int FF,
                                         float drop_p, uint32_t seed_attn, uint32_t seed_o, uint32_t seed_h, uint32_t seed_f,
                                         const uint32_t* seed_ptr, void* stream) {
  SEDT_REQUIRE(w1_frag && b1 && w2_frag && b2 && x2, "encoder_attn_ffn_fwd: null pointer");
  return enc_attn_ffn_launch(x, qk, v, kpm, w_o_frag, b_o, gamma2, beta2, w1_frag, b1, w2_frag, b2, x2, ctx, lse, x1, mean2, rstd2, x1n, h, B, S,
                             FF, drop_p, seed_attn, seed_o, seed_h, seed_f, seed_ptr, stream);
}

static int enc_attn_ffn_launch(const void* x, const void* qk, const void* v, const uint8_t* kpm, const void* w_o_frag,
                               const float* b_o, const float* gamma2, const float* beta2, const void* w1_frag,
                               const float* b1, const void* w2_frag, const float* b2, void* x2, void* ctx, float* lse,
                               void* x1, float* mean2, float* rstd2, void* x1n, void* h, int B, int S, int FF,
                               float drop_p, uint32_t seed_attn, uint32_t seed_o, uint32_t seed_h, uint32_t seed_f,
                               const uint32_t* seed_ptr, void* stream) {
  SEDT_REQUIRE(x && qk && v && w_o_frag && b_o && gamma2 && beta2, "encoder_attn_ffn_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && S >= 1 && S <= ES_LK && FF >= 512 && FF % 512 == 0, "encoder_attn_ffn_fwd: S = %d / FF = %d outside the envelope", S, FF);
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "encoder_attn_ffn_fwd: drop_p out of range");
  const bool train = ctx != nullptr;
  SEDT_REQUIRE(!train || (lse && x1 && mean2 && rstd2 && x1n && h), "encoder_attn_ffn_fwd: the training by-products come all or none");
  EncAttnFfnArgs a;
  a.x = (const bf16_t*)x; a.qk = (const bf16_t*)qk; a.v = (const bf16_t*)v; a.kpm = kpm;
  a.w_o = (const u32x4*)w_o_frag; a.b_o = b_o; a.gamma2 = gamma2; a.beta2 = beta2;
  a.w1 = (const u32x4*)w1_frag; a.b1 = b1; a.w2 = (const u32x4*)w2_frag; a.b2 = b2;
  a.x2 = (bf16_t*)x2; a.ctx = (bf16_t*)ctx; a.lse = lse; a.x1 = (bf16_t*)x1; a.mean2 = mean2; a.rstd2 = rstd2;
  a.x1n = (bf16_t*)x1n; a.h = (bf16_t*)h;
  a.B = B; a.S = S; a.FF = FF;
  a.scale = 0.17677669529663687f;                                  // 1 / sqrt(32)
  a.drop_p = drop_p;
  a.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  a.seed_attn = seed_attn; a.seed_o = seed_o; a.seed_h = seed_h; a.seed_f = seed_f; a.seed_ptr = seed_ptr;
  static const int dbg_env = dev_getenv("SEDT_SLAB_DBG") ? atoi(dev_getenv("SEDT_SLAB_DBG")) : 0;
  a.dbg = dbg_env;
  constexpr size_t lds_a = (size_t)2 * ES_H * ES_IMG + (ES_LK + 8 * 32) * sizeof(float) + (size_t)32 * XP * sizeof(bf16_t);
  constexpr size_t lds_c = (size_t)(2 * 32 * XP + 2 * 32 * ES_HP) * sizeof(bf16_t);
  constexpr size_t lds = lds_a > lds_c ? lds_a : lds_c;
  static bool attr = false;
  if (!attr) {
    if (set_lds(enc_attn_ffn_kernel<true>, lds, "encoder_attn_ffn_fwd") || set_lds(enc_attn_ffn_kernel<false>, lds, "encoder_attn_ffn_fwd"))
      return 1;
    attr = true;
  }
  const int grid = B * ((S + 31) / 32);
  if (train) hipLaunchKernelGGL(enc_attn_ffn_kernel<true>, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(enc_attn_ffn_kernel<false>, dim3(grid), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("encoder_attn_ffn_fwd");
}

extern "C" int sedt_encoder_ffn_bwd(const void* gx2, const void* h, const void* x1, const float* mean2, const float* rstd2,
                                    const float* gamma2, const void* w2t_frag, const void* w1t_frag, const void* wot_frag, void* g2,
                                    void* gh, void* gx1, void* g1, void* gctx, float* ln_part, int B, int S, int FF, float drop_p,
                                    uint32_t seed_f, uint32_t seed_o, const uint32_t* seed_ptr, void* stream) {
  SEDT_REQUIRE(gx2 && h && x1 && mean2 && rstd2 && gamma2 && w2t_frag && w1t_frag && wot_frag && gh && gx1 && gctx && ln_part,
               "encoder_ffn_bwd: null pointer");
  SEDT_REQUIRE(B >= 1 && S >= 1 && S <= ES_LK && FF >= 512 && FF % 512 == 0, "encoder_ffn_bwd: S = %d / FF = %d outside the envelope", S, FF);
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "encoder_ffn_bwd: drop_p out of range");
  EncFfnBwdArgs a;
  a.gx2 = (const bf16_t*)gx2; a.h = (const bf16_t*)h; a.x1 = (const bf16_t*)x1; a.mean2 = mean2; a.rstd2 = rstd2; a.gamma2 = gamma2;
  a.w2t = (const u32x4*)w2t_frag; a.w1t = (const u32x4*)w1t_frag; a.wot = (const u32x4*)wot_frag;
  a.g2 = (bf16_t*)g2; a.gh = (bf16_t*)gh; a.gx1 = (bf16_t*)gx1; a.g1 = (bf16_t*)g1; a.gctx = (bf16_t*)gctx; a.ln_part = ln_part;
  a.B = B; a.S = S; a.FF = FF; a.drop_p = drop_p;
  a.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  a.seed_f = seed_f; a.seed_o = seed_o; a.seed_ptr = seed_ptr;
  constexpr size_t lds = (size_t)(32 * XP + 3 * 32 * ES_HP) * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    if (set_lds(enc_ffn_bwd_kernel, lds, "encoder_ffn_bwd")) return 1;
    attr = true;
  }
  hipLaunchKernelGGL(enc_ffn_bwd_kernel, dim3(B * ((S + 31) / 32)), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("encoder_ffn_bwd");
}

extern "C" int sedt_encoder_qkv_bwd(const void* dqk, const void* dv, const void* x, const float* mean1, const float* rstd1,
                                    const float* gamma1, const void* gx1, const void* wint_frag, void* gx, float* ln_part, int B, int S,
                                    void* stream) {
  SEDT_REQUIRE(dqk && dv && x && mean1 && rstd1 && gamma1 && gx1 && wint_frag && gx && ln_part, "encoder_qkv_bwd: null pointer");
  SEDT_REQUIRE(B >= 1 && S >= 1 && S <= ES_LK, "encoder_qkv_bwd: S = %d outside 1..%d", S, ES_LK);
  EncQkvBwdArgs a{(const bf16_t*)dqk, (const bf16_t*)dv, (const bf16_t*)x, mean1, rstd1, gamma1, (const bf16_t*)gx1,
                  (const u32x4*)wint_frag, (bf16_t*)gx, ln_part, B, S};
  constexpr size_t lds = (size_t)(32 * ES_QP + 32 * XP) * sizeof(bf16_t);
  static bool attr = false;
  if (!attr) {
    if (set_lds(enc_qkv_bwd_kernel, lds, "encoder_qkv_bwd")) return 1;
    attr = true;
  }
  hipLaunchKernelGGL(enc_qkv_bwd_kernel, dim3(B * ((S + 31) / 32)), dim3(512), lds, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("encoder_qkv_bwd");
}

