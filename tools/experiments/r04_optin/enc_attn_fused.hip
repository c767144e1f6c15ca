// enc_attn_fused.hip - ARCHIVED EXPERIMENT (rounds 2-4), cut out of csrc/attn_mfma.hip in round 5; not part of the product library.
// LayerNorm1 + Q|K|V projections + attention core of a pre-norm encoder layer in ONE launch (workgroup = (clip, head pair)).
// Correct (it was tested against the 3-launch chain) and level with it, not ahead: 30.6 us no-grad / 32.2 us training form against
// 32.7 us; used for the decoder's self-attention (S = Q) it was slower (C2 5.49 -> 5.57 ms).  Phase ablation and the reasons:
// DESIGN.md appendix "dead ends", profiles/r04_slab_phase_ablation.txt.  Last commit that built and tested it: d3bd146ce59d.
// The three pieces below sat (1) between the forward and backward attention kernels, (2) among the host launchers, (3) at the end of
// csrc/attn_mfma.hip and use that file's helpers (AROW, AD, crow, set_attr_once ...).

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


// ---- host launcher
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


// ---- C ABI entry point
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
