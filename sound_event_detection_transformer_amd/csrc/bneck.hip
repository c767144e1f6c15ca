// bneck.hip - an identity Bottleneck of ResNet layer1 / layer2 (torchvision v1.5 block behind reference sedt/backbone.py:97-113: 1x1
// C -> C/4, 3x3, 1x1 C/4 -> C, FrozenBatchNorm after each, residual + ReLU) in ONE launch, its input-gradient chain in one more, and
// (end of file) the forward of layer1's first block with its projection skip.  The numbers below are layer1's (C = 256).
//
// The per-op path moves every intermediate through HBM: at B = 64 (128,000 pixels of a 125 x 16 map) a block reads x (65 MB) twice,
// writes and re-reads a and b (16 MB each) and writes y (65 MB) in three launches, ~70 us forward and ~79 us for the input gradients,
// each launch HBM-bound on its own.  Here a workgroup owns a strip of R image rows of one clip (16 columns wide = R/2 slabs of 32
// pixels, csrc/slab.h): the x tile with one halo row above and below sits in LDS once - it is the B operand of the first 1x1 AND the
// residual of the last - the two 64-channel intermediates never leave the CU (LDS tiles; the 3x3 reads its nine shifted views of a
// zero-padded tile), and only the weights stream from L2 (fragment-major, sedt_pack_frag).  HBM traffic per pixel: 512 B in (+ 2/R
// halo, mostly L2 hits) + 512 B out (+ 48 B of sign bits when a backward will follow; + 256 B of a and b only for a trainable block)
// against 2,080 B.
//
// The input-gradient chain has the same shape with the weights transposed (gy -> 1x1 256 -> 64 masked by [b > 0] -> 3x3 with the taps
// mirrored, masked by [a > 0] -> 1x1 64 -> 256 + gy, masked by the sign bits of the block input), so ONE kernel template serves both;
// the FrozenBN scales are folded into the transposed weights exactly as for the per-op dgrad kernels.  layer1 is frozen in the
// reference (backbone.py:60-62): no weight gradients are needed there; a trainable block (layer2) takes the two intermediate
// gradients out of the chain (gb_out / ga_out) for its weight-gradient GEMMs.
// Rounding points are those of the per-op chain: a, b (their gradients) rounded to bf16, accumulation in f32.
#include "bneck_common.h"

namespace sedt {

// Geometry of an identity Bottleneck: C block channels, P = C / 4 planes, W map columns; strips of R = 8 image rows.
// layer1: C 256, P 64, W 16 (128-pixel strips); layer2: C 512, P 128, W 8 (64-pixel strips) - the same bytes per image row, so every
// tile of the two has the same size in bytes and the same number of 16-byte pieces.
template <int C_, int P_, int W_>
struct BG {
  static constexpr int C = C_, P = P_, W = W_, R = 8;
  static constexpr int NP1 = (R + 2) * W, NP = R * W;                // pixels of the halo tile / of the strip
  static constexpr int NS1 = (NP1 + 31) / 32, NS = NP / 32;          // their 32-pixel slabs (the last halo slab may be partial)
  static constexpr int XP = C + 8, AP = P + 8, AW = W + 2;           // LDS pitches (elements): 16-byte fragment reads conflict-free
  static constexpr int T12 = P / 32, T3 = C / 32;                    // output tiles of stages 1, 2 / of stage 3
  static constexpr int KS1 = C / 16, KT = P / 16, KS3 = P / 16;      // k-steps of stage 1 / per tap of stage 2 / of stage 3
  static constexpr int G1 = 4 / T12;                                 // slab groups of the four compute waves (tile = wave % T12, group = wave / T12)
  static constexpr int S1 = 3, S2 = NS / G1;                         // slabs per compute wave in stage 1 / stage 2
  static constexpr int T3W = T3 / 8;                                 // stage-3 tiles per wave (all eight waves)
  static constexpr int NU = 9 * KT / 4;                              // stage 2 as a stream of 4-fragment units (a tap = KT / 4 units)
  static constexpr int CP = C / 8, PP = P / 8;                       // 16-byte pieces (= sign-bit bytes) per pixel of a C- / P-channel row
  static constexpr int NX = NP1 * CP / 256;                          // pieces of the in tile per moving thread
  static_assert(T12 * G1 == 4 && G1 * S1 >= NS1 && S2 * G1 == NS && T3W * 8 == T3 && KS1 % 16 == 0 && KT % 4 == 0, "wave assignment");
  static_assert(NX == 20 && NP1 * PP == 80 * 16 && NP * PP == 64 * 16 && NP * CP == 256 * 16 && W * CP == 512, "tile pieces");
};

struct BneckArgs {
  const bf16_t* in;                  // x (forward) / gy (backward) [B*H*W][C]
  bf16_t* out;                       // y / gx
  const u32x4* wA;                   // [P][C]    conv1 (fwd) / (s3 . conv3)^T (bwd), fragment-major
  const u32x4* wB;                   // [P][9*P]  conv2, k = tap * P + channel
  const u32x4* wC;                   // [C][P]    conv3 (fwd) / (s1 . conv1)^T (bwd)
  const float* sA; const float* bA; const float* sB; const float* bB; const float* sC; const float* bC;   // folded BN (fwd)
  bf16_t* a_out; bf16_t* b_out;      // [M][P] or null: fwd the two intermediates a, b; bwd their gradients ga (stage 2) and gb (stage 1) - what
                                     // the weight-gradient GEMMs of a trainable block read (a_out <- stage 1, b_out <- stage 2)
  uint8_t* abits_out; uint8_t* bbits_out;      // fwd: sign bits of a, b [M][P/8] or null (ALL the fused backward needs of them)
  uint8_t* bits_out;                 // fwd: sign bits of y [M][C/8] or null
  const uint8_t* abits_in; const uint8_t* bbits_in;      // bwd: sign bits of the saved a, b
  const uint8_t* bits_in;            // bwd: sign bits of the block input [M][C/8], or null (no mask)
  int B, H, spw;                     // spw: consecutive strips per workgroup
  int dbg;                           // developer build: phase ablation (1 stage 1, 2 stage 2, 4 stage 3, 8 stores, 16 tile loads)
};

template <class G, bool BWD>
struct BneckLds {
  static constexpr size_t XT = 0;
  static constexpr size_t AT = XT + (size_t)G::NP1 * G::XP * 2;
  static constexpr size_t BT = AT + (size_t)(G::R + 2) * G::AW * G::AP * 2;
  static constexpr size_t BITS = BT + (size_t)G::NP * G::AP * 2;                // [NP][C/8]
  static constexpr size_t MH = BITS + (size_t)G::NP * G::CP;                    // [NP1][P/8] sign bits on the halo tile: of a (fwd, out) / b (bwd, in)
  static constexpr size_t MA = MH + (size_t)G::NP1 * G::PP;                     // [NP][P/8] sign bits on the strip: of b (fwd, out) / a (bwd, in)
  static constexpr size_t SB = MA + (size_t)G::NP * G::PP;
  static constexpr size_t TOTAL = SB + (BWD ? 0 : (4 * G::P + 2 * G::C) * 4);
};

// The workgroup walks `spw` consecutive strips, one workgroup per CU (the tile is 84 KB of LDS).  What the measurements of the earlier
// versions say (profiles/r04_bneck_ablation.txt): a strip is bound by the L2 -> CU weight stream (~40 B/clk/CU, csrc/slab.h), not by HBM
// and not by the MFMAs - so a weight fragment has to serve as many pixels as the accumulators allow (2-3 slabs per fragment; a 4-row
// version with two tile buffers streamed twice the bytes per pixel and took 8.6 us per 64 pixels) - and HBM latency must never sit
// inside a stage.  Waves 0..3 COMPUTE stages 1 and 2 (one per SIMD; the weight stream never stops: the next chunk / tap / strip is
// always in flight) and issue no stores (loads and stores share one in-order counter: a store ahead of a weight fragment would hold the
// MFMA that waits for the fragment until HBM has acknowledged the store); waves 4..7 MOVE: they hold the NEXT strip's tile in registers
// for a whole strip (issued right after the current tile went to LDS) and write the first intermediate out; stage 3 and the output
// stores are everybody's.
//   A(s) .. B(s): compute stage 1 (reads XT, writes AT)             | move: issue the loads of strip s + 1 (registers)
//   B(s) .. C(s): compute stage 2 (reads AT, writes BT)             | move: store the stage-1 result of s (AT)
//   C(s) .. D(s): all: stage 3 (reads BT, XT in place, BITS)
//   D(s) .. E(s): all: store out / stage-2 result / sign bits of s (XT, BT, BITS, MA)
//   E(s) .. A(s+1): move: registers -> XT (+ the sign bits of strip s + 1; bwd)
template <class G, bool BWD, bool SKIP3 = false>      // SKIP3 (bwd): stop after stage 2 - the caller finishes the chain
__global__ __launch_bounds__(512) void bneck_kernel(const BneckArgs a) {
  using LD = BneckLds<G, BWD>;
  constexpr int NP1 = G::NP1, NP = G::NP, NS1 = G::NS1, NS = G::NS, R = G::R, W = G::W, C = G::C, P = G::P;
  constexpr int XP = G::XP, AP = G::AP, AW = G::AW, CP = G::CP, PP = G::PP, NX = G::NX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XT = reinterpret_cast<bf16_t*>(smem + LD::XT);        // [NP1][XP]: in tile, halo row first; becomes the out tile
  bf16_t* AT = reinterpret_cast<bf16_t*>(smem + LD::AT);        // [(R + 2)][W + 2][AP]: 3x3 input, zero border
  bf16_t* BT = reinterpret_cast<bf16_t*>(smem + LD::BT);        // [NP][AP]: 3x3 output
  uint8_t* BITS = smem + LD::BITS;                              // [NP][C/8]: fwd sign bits of y; bwd sign bits of the block input
  uint8_t* MH = smem + LD::MH;                                  // [NP1][P/8]
  uint8_t* MA = smem + LD::MA;                                  // [NP][P/8]
  float* SB = reinterpret_cast<float*>(smem + LD::SB);          // fwd: sA bA sB bB (P each) sC bC (C each)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int strips = (a.H + R - 1) / R, nst = a.B * strips;
  const int first = blockIdx.x * a.spw, last = min(first + a.spw, nst);
  if (first >= last) return;
  const bool comp = wave < 4;
  const int tl = wave % G::T12, grp = (wave / G::T12) % G::G1;  // compute waves: output tile of stages 1 / 2, slab group

  // ONE register pool for both roles (a wave computes or moves for its whole life): the compute waves' weight chunks cur = pool[0..7],
  // alt = pool[8..15]; the moving waves' tile in flight xr = pool[0..19] and (bwd) the three sign-bit pieces pool[20..22]
  constexpr int NW3 = G::T3W * G::KS3;                          // stage-3 fragments of a wave: 4 (kept for every strip) or 16 (loaded per strip)
  constexpr bool W3_KEPT = NW3 <= 4;
  u32x4 pool[BWD ? 23 : 20], w3r[NW3];
  u32x4(&cur)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[0]);
  u32x4(&alt)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[8]);
  u32x4(&xr)[20] = *reinterpret_cast<u32x4(*)[20]>(&pool[0]);
  const u32x4* w1p = a.wA + (long)tl * G::KS1 * 64;
  const u32x4* w2p = a.wB + (long)tl * (9 * G::KT) * 64;
  const u32x4* w3p = a.wC + (long)(wave * G::T3W) * G::KS3 * 64;            // stage 3: tiles wave * T3W + [0, T3W), KS3 fragments each, contiguous
  auto load_w3 = [&](int t, int lane) {                         // tile t of this wave into half t & 1 of w3r
#pragma unroll
    for (int u = 0; u < G::KS3; ++u) w3r[(NW3 > G::KS3 ? (t & 1) * G::KS3 : 0) + u] = w3p[(t * G::KS3 + u) * 64 + lane];
  };
  if (W3_KEPT && !SKIP3) load_w3(0, tid & 63);
  if (comp) slab::load_chunk<1>(cur, w1p, 0, 0, tid & 63);
  slab::issue_fence();

  auto geom = [&](int s, int& r0, long& pix0) {
    const int clip = s / strips;
    r0 = (s % strips) * R;
    pix0 = ((long)clip * a.H + r0) * W;
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u}, ones4 = {~0u, ~0u, ~0u, ~0u};
  // the in tile of strip s into the moving threads' registers / from there into LDS
  auto fetch = [&](int s) {
    int r0; long pix0;
    geom(s, r0, pix0);
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int u = mt + q * 256, p = u / CP, c = (u % CP) * 8, gr = r0 - 1 + p / W;
      xr[q] = (gr >= 0 && gr < a.H) ? *reinterpret_cast<const u32x4*>(a.in + (pix0 + p - W) * C + c) : zero4;
    }
    if (BWD) {          // sign bits: b on the halo tile, a, the block input (an image row is 8 / 8 / 32 pieces of 16 bytes in both geometries)
      const int grh = r0 - 1 + (mt >> 3);
      pool[BWD ? 20 : 0] = (mt < 80 && grh >= 0 && grh < a.H) ? reinterpret_cast<const u32x4*>(a.bbits_in + (pix0 - W) * PP)[mt] : zero4;
      pool[BWD ? 21 : 0] = (mt < 64 && r0 + (mt >> 3) < a.H) ? reinterpret_cast<const u32x4*>(a.abits_in + pix0 * PP)[mt] : zero4;
      pool[BWD ? 22 : 0] = !a.bits_in ? ones4 : r0 + (mt >> 5) < a.H ? reinterpret_cast<const u32x4*>(a.bits_in + pix0 * CP)[mt] : zero4;
    }
  };
  auto put = [&]() {
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int u = mt + q * 256, p = u / CP, c = (u % CP) * 8;
      *reinterpret_cast<u32x4*>(XT + p * XP + c) = xr[q];
    }
    if (BWD) {
      if (mt < 80) reinterpret_cast<u32x4*>(MH)[mt] = pool[BWD ? 20 : 0];
      if (mt < 64) reinterpret_cast<u32x4*>(MA)[mt] = pool[BWD ? 21 : 0];
      reinterpret_cast<u32x4*>(BITS)[mt] = pool[BWD ? 22 : 0];
    }
  };

  // ---- prologue: constants, the zero border of the 3x3 input tile, the first strip's tiles
  for (int u = tid; u < (R + 2) * AW * AP / 8; u += 512) reinterpret_cast<uint4*>(AT)[u] = make_uint4(0, 0, 0, 0);
  if (!BWD)
    for (int u = tid; u < 4 * P + 2 * C; u += 512) {
      const float* src = u < P ? a.sA + u : u < 2 * P ? a.bA + (u - P) : u < 3 * P ? a.sB + (u - 2 * P) : u < 4 * P ? a.bB + (u - 3 * P)
                         : u < 4 * P + C ? a.sC + (u - 4 * P) : a.bC + (u - 4 * P - C);
      SB[u] = *src;
    }
  if (!comp) {
    fetch(first);
    put();
  }
  __syncthreads();

  for (int s = first; s < last; ++s) {
    int r0; long pix0;
    geom(s, r0, pix0);
    const int rows_in = min(R, a.H - r0);                        // interior rows inside the image
    const bool more = s + 1 < last;
    const int lane = opaque(tid) & 63, n = lane & 31, hf = lane >> 5, mt = opaque(tid) - 256;     // (mt: index of a tile-moving thread)

    // ---- A(s) .. B(s)
    if (comp) {
     if (!(a.dbg & 1)) {
      // stage 1: 1x1, C -> P on the halo tile: tile tl, slabs 3 * grp + {0, 1, 2} (pixels past the tile: computed on whatever follows it
      // in LDS and dropped)
      f32x16 acc[3];
      slab::zero_acc<3>(acc);
      gemm_slabs<3, G::KS1>(acc, XT + (3 * grp) * 32 * XP, XP, w1p, lane, cur, alt, [&](u32x4(&d)[8]) {
        load4(d, 0, w2p, lane);                                  // units 0, 1 of the 3x3 (land in cur)
        load4(d, 4, w2p + 4 * 64, lane);
      });
      load4(alt, 0, w2p + 8 * 64, lane);                         // units 2, 3
      load4(alt, 4, w2p + 12 * 64, lane);
      slab::issue_fence();
      float4 sc[4], bi[4];
      if (!BWD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4*>(SB + tl * 32 + 8 * g4 + 4 * hf);
          bi[g4] = *reinterpret_cast<const float4*>(SB + P + tl * 32 + 8 * g4 + 4 * hf);
        }
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int sl = 3 * grp + s3;
        if (sl < NS1) {
          const int p = sl * 32 + n, trow = p / W, pc = p % W, gr = r0 - 1 + trow;
          const bool intile = NP1 % 32 == 0 || p < NP1;          // (the last slab of the halo tile may be partial)
          const bool inimg = gr >= 0 && gr < a.H;
          bf16_t* dst = AT + (trow * AW + pc + 1) * AP + tl * 32 + 4 * hf;
          unsigned m4 = 0, nb = 0;
          if (BWD && intile) m4 = *reinterpret_cast<const unsigned*>(MH + p * PP + tl * 4) >> (4 * hf);       // bytes g4 = 0..3 of this tile
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x2 lo = {acc[s3][4 * g4], acc[s3][4 * g4 + 1]}, hi = {acc[s3][4 * g4 + 2], acc[s3][4 * g4 + 3]};
            uint2 o;
            if (BWD) {
              o.x = pack2(lo) & keep2(m4, 8 * g4);
              o.y = pack2(hi) & keep2(m4, 8 * g4 + 2);
            } else {
              o.x = pack2(relu2(lo * f32x2{sc[g4].x, sc[g4].y} + f32x2{bi[g4].x, bi[g4].y}));
              o.y = pack2(relu2(hi * f32x2{sc[g4].z, sc[g4].w} + f32x2{bi[g4].z, bi[g4].w}));
              if (!inimg) o = make_uint2(0, 0);
              nb |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
            }
            if (intile) *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
          }
          if (!BWD && a.abits_out) {                             // the two lane halves hold the two nibbles of every byte
            const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
            if (hf == 0 && intile) *reinterpret_cast<unsigned*>(MH + p * PP + tl * 4) = nb | other;
          }
        }
      }
     }
    } else if (more && !(a.dbg & 16)) {
      fetch(s + 1);                                              // (a whole strip to land)
    }
    __syncthreads();

    // ---- B(s) .. C(s)
    if (comp) {
     if (!(a.dbg & 2)) {
      // stage 2: 3x3, P -> P: tile tl, slabs S2 * grp + [0, S2)
      constexpr int S2 = G::S2, UPT = G::KT / 4;                 // (units per tap)
      f32x16 acc[S2];
      slab::zero_acc<S2>(acc);
      const bf16_t* ctr[S2];                                     // lane n <-> pixel: centre of its 3x3 neighbourhood in the padded tile
#pragma unroll
      for (int s2 = 0; s2 < S2; ++s2) {
        const int q = (grp * S2 + s2) * 32 + n;
        ctr[s2] = AT + ((q / W + 1) * AW + q % W + 1) * AP + 8 * hf;
      }
#pragma unroll
      for (int u = 0; u < G::NU; ++u) {
        // unit u sits in quarter u % 4 of (cur | alt); three units are in flight behind it
        u32x4(&src)[8] = (u & 2) ? alt : cur;
        const int so = (u & 1) * 4;
        const int tap = u / UPT, kk0 = (u % UPT) * 4;
        const int dr = tap / 3 - 1, dc = tap % 3 - 1;
        const int off = (BWD ? -(dr * AW + dc) : (dr * AW + dc)) * AP + kk0 * 16;       // the input gradient mirrors the taps
        bf16x8 xb[4][S2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int s2 = 0; s2 < S2; ++s2) xb[kk][s2] = *reinterpret_cast<const bf16x8*>(ctr[s2] + off + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int s2 = 0; s2 < S2; ++s2)
            acc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[so + kk]), xb[kk][s2], acc[s2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < G::NU) load4(src, so, w2p + (long)(u + 4) * 4 * 64, lane);
        if (u == G::NU - 1) slab::load_chunk<1>(cur, w1p, 0, 0, lane);  // the next strip's first chunk (cur is free from here)
        __builtin_amdgcn_sched_barrier(0);
      }
      float4 sc[4], bi[4];
      if (!BWD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4*>(SB + 2 * P + tl * 32 + 8 * g4 + 4 * hf);
          bi[g4] = *reinterpret_cast<const float4*>(SB + 3 * P + tl * 32 + 8 * g4 + 4 * hf);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < S2; ++s2) {
        const int q = (grp * S2 + s2) * 32 + n;
        unsigned m4 = 0, nb = 0;
        if (BWD) m4 = *reinterpret_cast<const unsigned*>(MA + q * PP + tl * 4) >> (4 * hf);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int ch = tl * 32 + 8 * g4 + 4 * hf;
          const f32x2 lo = {acc[s2][4 * g4], acc[s2][4 * g4 + 1]}, hi = {acc[s2][4 * g4 + 2], acc[s2][4 * g4 + 3]};
          uint2 o;
          if (BWD) {
            o.x = pack2(lo) & keep2(m4, 8 * g4);
            o.y = pack2(hi) & keep2(m4, 8 * g4 + 2);
          } else {
            o.x = pack2(relu2(lo * f32x2{sc[g4].x, sc[g4].y} + f32x2{bi[g4].x, bi[g4].y}));
            o.y = pack2(relu2(hi * f32x2{sc[g4].z, sc[g4].w} + f32x2{bi[g4].z, bi[g4].w}));
            nb |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
          }
          *reinterpret_cast<uint2*>(BT + q * AP + ch) = o;
        }
        if (!BWD && a.bbits_out) {
          const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
          if (hf == 0) *reinterpret_cast<unsigned*>(MA + q * PP + tl * 4) = nb | other;
        }
      }
     }
    } else if (!(a.dbg & 8)) {
      if (a.a_out)                                               // the stage-1 result on the strip's own rows
        for (int u = mt; u < NP * PP; u += 256) {
          const int q = u / PP, c = (u % PP) * 8;
          if (q / W < rows_in)
            *reinterpret_cast<uint4*>(a.a_out + (pix0 + q) * P + c) = *reinterpret_cast<const uint4*>(AT + ((q / W + 1) * AW + q % W + 1) * AP + c);
        }
      if (!BWD && a.abits_out && mt < 64 && (mt >> 3) < rows_in)         // (the strip's own rows of the halo tile)
        reinterpret_cast<uint4*>(a.abits_out + pix0 * PP)[mt] = reinterpret_cast<const uint4*>(MH + W * PP)[mt];
    }
    if (!W3_KEPT && !SKIP3) load_w3(0, lane);                  // (two tiles per wave: loaded per strip, the first in flight across the barrier)
    __syncthreads();

    // ---- C(s) .. D(s): stage 3 (all waves, tiles wave * T3W + t): 1x1, P -> C, + the in tile (residual), ReLU / sign-bit mask, in place
    if (!(a.dbg & 4) && !SKIP3) {
      constexpr int T3W = G::T3W;
      const bf16_t* xrow = BT + n * AP + 8 * hf;
#pragma unroll
      for (int t = 0; t < T3W; ++t) {
        const int tile = wave * T3W + t;
        if (!W3_KEPT && t + 1 < T3W) load_w3(t + 1, lane);        // (the next tile's fragments land while this one is multiplied)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[NS];
        slab::zero_acc<NS>(acc);
#pragma unroll
        for (int kk = 0; kk < G::KS3; ++kk) {
          bf16x8 xb[NS];
#pragma unroll
          for (int s3 = 0; s3 < NS; ++s3) xb[s3] = *reinterpret_cast<const bf16x8*>(xrow + s3 * 32 * AP + kk * 16);
#pragma unroll
          for (int s3 = 0; s3 < NS; ++s3)
            acc[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[(NW3 > G::KS3 ? (t & 1) * G::KS3 : 0) + kk]), xb[s3],
                                                              acc[s3], 0, 0, 0);
        }
        unsigned nibs[NS];                                       // fwd: this lane's sign nibbles of slab s3, nibble g4 at bits 8 * g4 (+ 4 * hf)
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3) {
          const int q = s3 * 32 + n;
          bf16_t* xp = XT + (q + W) * XP + tile * 32 + 4 * hf;
          unsigned m4 = 0;
          if (BWD) m4 = *reinterpret_cast<const unsigned*>(BITS + q * CP + tile * 4) >> (4 * hf);
          nibs[s3] = 0;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const uint2 xw = *reinterpret_cast<const uint2*>(xp + 8 * g4);
            const f32x2 lo = {acc[s3][4 * g4], acc[s3][4 * g4 + 1]}, hi = {acc[s3][4 * g4 + 2], acc[s3][4 * g4 + 3]};
            uint2 o;
            if (BWD) {
              o.x = pack2(lo + widen2(xw.x)) & keep2(m4, 8 * g4);
              o.y = pack2(hi + widen2(xw.y)) & keep2(m4, 8 * g4 + 2);
            } else {
              // (scale / bias read per use: eight float4 held across the slab loop pushed the 512-channel geometry into scratch)
              const float4 sc = *reinterpret_cast<const float4*>(SB + 4 * P + tile * 32 + 8 * g4 + 4 * hf);
              const float4 bi = *reinterpret_cast<const float4*>(SB + 4 * P + C + tile * 32 + 8 * g4 + 4 * hf);
              o.x = pack2(relu2(lo * f32x2{sc.x, sc.y} + f32x2{bi.x, bi.y} + widen2(xw.x)));
              o.y = pack2(relu2(hi * f32x2{sc.z, sc.w} + f32x2{bi.z, bi.w} + widen2(xw.y)));
              nibs[s3] |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
            }
            *reinterpret_cast<uint2*>(xp + 8 * g4) = o;
          }
        }
        if (!BWD && a.bits_out) {
          // the two lane halves hold the two nibbles of every byte: one exchange per slab, one 4-byte store by the lower half
#pragma unroll
          for (int s3 = 0; s3 < NS; ++s3) {
            const unsigned other = (unsigned)__shfl_xor((int)nibs[s3], 32);
            if (hf == 0) *reinterpret_cast<unsigned*>(BITS + (s3 * 32 + n) * CP + tile * 4) = nibs[s3] | other;
          }
        }
      }
    }
    __syncthreads();

    // ---- D(s) .. E(s): out (all waves)
    if (!(a.dbg & 8)) {
      const int t = opaque(tid);
      if (!SKIP3) {
#pragma unroll 4
        for (int u = t; u < NP * CP; u += 512) {
          const int q = u / CP, c = (u % CP) * 8;
          if (q / W < rows_in) *reinterpret_cast<uint4*>(a.out + (pix0 + q) * C + c) = *reinterpret_cast<const uint4*>(XT + (q + W) * XP + c);
        }
      }
      if (a.b_out)                                               // the stage-2 result
        for (int u = t; u < NP * PP; u += 512) {
          const int q = u / PP, c = (u % PP) * 8;
          if (q / W < rows_in) *reinterpret_cast<uint4*>(a.b_out + (pix0 + q) * P + c) = *reinterpret_cast<const uint4*>(BT + q * AP + c);
        }
      if (!BWD) {
        if (a.bits_out && t < 256 && (t >> 5) < rows_in) reinterpret_cast<uint4*>(a.bits_out + pix0 * CP)[t] = reinterpret_cast<const uint4*>(BITS)[t];
        if (a.bbits_out && t >= 256 && t < 320 && ((t - 256) >> 3) < rows_in)
          reinterpret_cast<uint4*>(a.bbits_out + pix0 * PP)[t - 256] = reinterpret_cast<const uint4*>(MA)[t - 256];
      }
    }
    if (more) {
      __syncthreads();
      // ---- E(s) .. A(s+1)
      if (!comp && !(a.dbg & 16)) put();
      __syncthreads();
    }
  }
}

template <class G, bool BWD, bool SKIP3 = false>
static int bneck_launch(BneckArgs& a, hipStream_t s, const char* what) {
  using LD = BneckLds<G, BWD>;
  static_assert(LD::TOTAL <= 160 * 1024, "LDS");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck_kernel<G, BWD, SKIP3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LD::TOTAL);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute(%zu B LDS) failed: %s", what, (size_t)LD::TOTAL, hipGetErrorString(e));
      return 1;
    }
    attr = true;
  }
  // one workgroup per CU: 256 workgroups walk ceil(strips / 256) consecutive strips each
  const int nst = a.B * ((a.H + G::R - 1) / G::R);
  static int spw_env = [] { const char* e = dev_getenv("SEDT_BNECK_SPW"); return e ? atoi(e) : 0; }();
  a.spw = spw_env > 0 ? spw_env : (nst + 255) / 256;
  static int dbg_env = [] { const char* e = dev_getenv("SEDT_BNECK_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg_env;
  hipLaunchKernelGGL((bneck_kernel<G, BWD, SKIP3>), dim3((nst + a.spw - 1) / a.spw), dim3(512), LD::TOTAL, s, a);
  return check_launch(what);
}

// ---------------------------------------------------------------------------------------------------------------- layer1, block 0
// The first Bottleneck of layer1 (64 -> 64 -> 64 -> 256 with the 1x1 projection 64 -> 256 on the skip path, stride 1) in ONE forward
// launch: x (the pooled stem output, 128 B per pixel) in, y out; the per-op chain of four launches moves 2.3 KB per pixel for the 640 B
// this one does (C4 runs the block over 2,200 clip-equivalents per step).  Same strip walk and wave roles as bneck_kernel; the
// differences: the in tile has 64 channels (so the out tile is its own LDS buffer), stage 1 has K = 64, and stage 3 is TWO K = 64
// products per output tile - conv3 over b and the projection over x - with separate FrozenBN affines, the projection rounded to bf16
// before the sum exactly where the per-op chain stores it.  W1 / W3 / Wd fragments live in registers for the whole launch; only the
// 3x3 streams.  The backward of this block stays per-op (it reads x, a, b, y: written here when training).
struct Bneck0Args {
  const bf16_t* in;                  // x [M][64]
  bf16_t* out;                       // y [M][256]
  const u32x4* w1; const u32x4* w2; const u32x4* w3; const u32x4* wd;      // [64][64], [64][576], [256][64], [256][64] fragment-major
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3; const float* sd; const float* bd;
  bf16_t* a_out; bf16_t* b_out;      // [M][64] or null
  uint8_t* abits_out; uint8_t* bbits_out;      // sign bits of a, b [M][8] or null (what sedt_bneck_bwd reads)
  uint8_t* bits_out;                 // sign bits of y [M][32] or null
  int B, H, spw;
};

struct B0 {
  static constexpr int CI = 64, P = 64, C = 256, W = 16, R = 8, NP1 = (R + 2) * W, NP = R * W, NS1 = NP1 / 32, NS = NP / 32;
  static constexpr int XP = CI + 8, AP = P + 8, AW = W + 2, YP = C + 8;
  static constexpr size_t XT = 0;
  static constexpr size_t AT = XT + (size_t)NP1 * XP * 2;
  static constexpr size_t BT = AT + (size_t)(R + 2) * AW * AP * 2;
  static constexpr size_t YT = BT + (size_t)NP * AP * 2;
  static constexpr size_t BITS = YT + (size_t)NP * YP * 2;
  static constexpr size_t MH = BITS + (size_t)NP * 32;           // sign bits of a on the halo tile [NP1][8]
  static constexpr size_t MA = MH + (size_t)NP1 * 8;             // sign bits of b [NP][8]
  static constexpr size_t SB = MA + (size_t)NP * 8;
  static constexpr size_t TOTAL = SB + (4 * 64 + 4 * 256) * 4;
};

__global__ __launch_bounds__(512) void bneck0_fwd_kernel(const Bneck0Args a) {
  constexpr int NP1 = B0::NP1, NP = B0::NP, NS1 = B0::NS1, R = B0::R, W = B0::W, XP = B0::XP, AP = B0::AP, AW = B0::AW, YP = B0::YP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XT = reinterpret_cast<bf16_t*>(smem + B0::XT);        // [NP1][72]: x on the halo tile
  bf16_t* AT = reinterpret_cast<bf16_t*>(smem + B0::AT);        // [10][18][72]: 3x3 input, zero border
  bf16_t* BT = reinterpret_cast<bf16_t*>(smem + B0::BT);        // [NP][72]: 3x3 output
  bf16_t* YT = reinterpret_cast<bf16_t*>(smem + B0::YT);        // [NP][264]: out tile
  uint8_t* BITS = smem + B0::BITS;                              // [NP][32]
  uint8_t* MH = smem + B0::MH;
  uint8_t* MA = smem + B0::MA;
  float* SB = reinterpret_cast<float*>(smem + B0::SB);          // s1 b1 s2 b2 (64 each) s3 b3 sd bd (256 each)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int strips = (a.H + R - 1) / R, nst = a.B * strips;
  const int first = blockIdx.x * a.spw, last = min(first + a.spw, nst);
  if (first >= last) return;
  const bool comp = wave < 4;
  const int tl = wave & 1, grp = (wave >> 1) & 1;

  // one register pool for both roles: the compute waves' 3x3 units (cur | alt), the moving waves' next in tile (5 pieces)
  u32x4 pool[16], w1r[4], w3r[4], wdr[4];
  u32x4(&cur)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[0]);
  u32x4(&alt)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[8]);
  const u32x4* w2p = a.w2 + (long)tl * 36 * 64;
  load4(w3r, 0, a.w3 + (long)wave * 4 * 64, tid & 63);
  load4(wdr, 0, a.wd + (long)wave * 4 * 64, tid & 63);
  if (comp) {
    load4(w1r, 0, a.w1 + (long)tl * 4 * 64, tid & 63);
    load4(cur, 0, w2p, tid & 63);                                // units 0..3 of the 3x3 (reloaded at the end of every stage 2 for the next strip)
    load4(cur, 4, w2p + 4 * 64, tid & 63);
    load4(alt, 0, w2p + 8 * 64, tid & 63);
    load4(alt, 4, w2p + 12 * 64, tid & 63);
  }
  slab::issue_fence();

  auto geom = [&](int s, int& r0, long& pix0) {
    const int clip = s / strips;
    r0 = (s % strips) * R;
    pix0 = ((long)clip * a.H + r0) * W;
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto fetch = [&](int s) {
    int r0; long pix0;
    geom(s, r0, pix0);
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int u = mt + q * 256, p = u >> 3, c = (u & 7) * 8, gr = r0 - 1 + (p >> 4);
      pool[q] = (gr >= 0 && gr < a.H) ? *reinterpret_cast<const u32x4*>(a.in + (pix0 + p - W) * 64 + c) : zero4;
    }
  };
  auto put = [&]() {
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const int u = mt + q * 256, p = u >> 3, c = (u & 7) * 8;
      *reinterpret_cast<u32x4*>(XT + p * XP + c) = pool[q];
    }
  };

  for (int u = tid; u < (R + 2) * AW * AP / 8; u += 512) reinterpret_cast<uint4*>(AT)[u] = make_uint4(0, 0, 0, 0);
  for (int u = tid; u < 1280; u += 512) {
    const float* src = u < 64 ? a.s1 + u : u < 128 ? a.b1 + (u - 64) : u < 192 ? a.s2 + (u - 128) : u < 256 ? a.b2 + (u - 192)
                       : u < 512 ? a.s3 + (u - 256) : u < 768 ? a.b3 + (u - 512) : u < 1024 ? a.sd + (u - 768) : a.bd + (u - 1024);
    SB[u] = *src;
  }
  if (!comp) {
    fetch(first);
    put();
  }
  __syncthreads();

  for (int s = first; s < last; ++s) {
    int r0; long pix0;
    geom(s, r0, pix0);
    const int rows_in = min(R, a.H - r0);
    const bool more = s + 1 < last;
    const int lane = opaque(tid) & 63, n = lane & 31, hf = lane >> 5, mt = opaque(tid) - 256;

    // ---- A .. B: stage 1 (1x1, 64 -> 64 on the halo tile: tile tl, slabs 3 * grp + {0, 1, 2}) | move: the next in tile into registers
    if (comp) {
      f32x16 acc[3];
      slab::zero_acc<3>(acc);
      const bf16_t* xrow = XT + ((3 * grp) * 32 + n) * XP + 8 * hf;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 xb[3];
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) xb[s3] = *reinterpret_cast<const bf16x8*>(xrow + s3 * 32 * XP + kk * 16);
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) acc[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1r[kk]), xb[s3], acc[s3], 0, 0, 0);
      }
      float4 sc[4], bi[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4*>(SB + tl * 32 + 8 * g4 + 4 * hf);
        bi[g4] = *reinterpret_cast<const float4*>(SB + 64 + tl * 32 + 8 * g4 + 4 * hf);
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int sl = 3 * grp + s3;
        if (sl < NS1) {
          const int p = sl * 32 + n, trow = p >> 4, pc = p & 15, gr = r0 - 1 + trow;
          const bool inimg = gr >= 0 && gr < a.H;
          bf16_t* dst = AT + (trow * AW + pc + 1) * AP + tl * 32 + 4 * hf;
          unsigned nb = 0;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x2 lo = {acc[s3][4 * g4], acc[s3][4 * g4 + 1]}, hi = {acc[s3][4 * g4 + 2], acc[s3][4 * g4 + 3]};
            uint2 o;
            o.x = pack2(relu2(lo * f32x2{sc[g4].x, sc[g4].y} + f32x2{bi[g4].x, bi[g4].y}));
            o.y = pack2(relu2(hi * f32x2{sc[g4].z, sc[g4].w} + f32x2{bi[g4].z, bi[g4].w}));
            if (!inimg) o = make_uint2(0, 0);
            nb |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
            *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
          }
          if (a.abits_out) {
            const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
            if (hf == 0) *reinterpret_cast<unsigned*>(MH + p * 8 + tl * 4) = nb | other;
          }
        }
      }
    } else if (more) {
      fetch(s + 1);
    }
    __syncthreads();

    // ---- B .. C: stage 2 (3x3: tile tl, slabs 2 * grp + {0, 1}) | move: the first intermediate out
    if (comp) {
      f32x16 acc[2];
      slab::zero_acc<2>(acc);
      const bf16_t* ctr[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q = (grp * 2 + s2) * 32 + n;
        ctr[s2] = AT + (((q >> 4) + 1) * AW + (q & 15) + 1) * AP + 8 * hf;
      }
#pragma unroll
      for (int u = 0; u < 9; ++u) {
        u32x4(&src)[8] = (u & 2) ? alt : cur;
        const int so = (u & 1) * 4;
        const int off = ((u / 3 - 1) * AW + (u % 3 - 1)) * AP;
        bf16x8 xb[4][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) xb[kk][s2] = *reinterpret_cast<const bf16x8*>(ctr[s2] + off + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
            acc[s2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[so + kk]), xb[kk][s2], acc[s2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < 9) load4(src, so, w2p + (long)(u + 4) * 4 * 64, lane);        // the ring: unit u + 4 into the quarter just used
        __builtin_amdgcn_sched_barrier(0);
      }
      load4(cur, 0, w2p, lane);                                  // the next strip's units 0..3 (unit u lives in quarter u % 4): in flight
      load4(cur, 4, w2p + 4 * 64, lane);                         // through stage 3, the stores and stage 1
      load4(alt, 0, w2p + 8 * 64, lane);
      load4(alt, 4, w2p + 12 * 64, lane);
      slab::issue_fence();
      float4 sc[4], bi[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        sc[g4] = *reinterpret_cast<const float4*>(SB + 128 + tl * 32 + 8 * g4 + 4 * hf);
        bi[g4] = *reinterpret_cast<const float4*>(SB + 192 + tl * 32 + 8 * g4 + 4 * hf);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q = (grp * 2 + s2) * 32 + n;
        unsigned nb = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x2 lo = {acc[s2][4 * g4], acc[s2][4 * g4 + 1]}, hi = {acc[s2][4 * g4 + 2], acc[s2][4 * g4 + 3]};
          uint2 o;
          o.x = pack2(relu2(lo * f32x2{sc[g4].x, sc[g4].y} + f32x2{bi[g4].x, bi[g4].y}));
          o.y = pack2(relu2(hi * f32x2{sc[g4].z, sc[g4].w} + f32x2{bi[g4].z, bi[g4].w}));
          nb |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
          *reinterpret_cast<uint2*>(BT + q * AP + tl * 32 + 8 * g4 + 4 * hf) = o;
        }
        if (a.bbits_out) {
          const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
          if (hf == 0) *reinterpret_cast<unsigned*>(MA + q * 8 + tl * 4) = nb | other;
        }
      }
    } else {
      if (a.a_out)
        for (int u = mt; u < NP * 8; u += 256) {
          const int q = u >> 3, c = (u & 7) * 8;
          if ((q >> 4) < rows_in)
            *reinterpret_cast<uint4*>(a.a_out + (pix0 + q) * 64 + c) = *reinterpret_cast<const uint4*>(AT + (((q >> 4) + 1) * AW + (q & 15) + 1) * AP + c);
        }
      if (a.abits_out && mt < 64 && (mt >> 3) < rows_in) reinterpret_cast<uint4*>(a.abits_out + pix0 * 8)[mt] = reinterpret_cast<const uint4*>(MH + W * 8)[mt];
    }
    __syncthreads();

    // ---- C .. D: stage 3 (all waves, tile = wave): conv3 over b and the projection over x, two slabs at a time
    {
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        f32x16 acc3[2], accd[2];
        slab::zero_acc<2>(acc3);
        slab::zero_acc<2>(accd);
        const bf16_t* brow = BT + ((2 * sp) * 32 + n) * AP + 8 * hf;
        const bf16_t* xrow = XT + ((2 * sp) * 32 + n + W) * XP + 8 * hf;        // (the strip's own rows of the halo tile)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          bf16x8 bb[2], xb[2];
#pragma unroll
          for (int s3 = 0; s3 < 2; ++s3) {
            bb[s3] = *reinterpret_cast<const bf16x8*>(brow + s3 * 32 * AP + kk * 16);
            xb[s3] = *reinterpret_cast<const bf16x8*>(xrow + s3 * 32 * XP + kk * 16);
          }
#pragma unroll
          for (int s3 = 0; s3 < 2; ++s3) {
            acc3[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[kk]), bb[s3], acc3[s3], 0, 0, 0);
            accd[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wdr[kk]), xb[s3], accd[s3], 0, 0, 0);
          }
        }
#pragma unroll
        for (int s3 = 0; s3 < 2; ++s3) {
          const int q = (2 * sp + s3) * 32 + n;
          unsigned nibs = 0;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float* sbp = SB + wave * 32 + 8 * g4 + 4 * hf;      // (read per use: sixteen float4 held across the slab loop spilled)
            const float4 s3v_ = *reinterpret_cast<const float4*>(sbp + 256), b3v_ = *reinterpret_cast<const float4*>(sbp + 512);
            const float4 sdv_ = *reinterpret_cast<const float4*>(sbp + 768), bdv_ = *reinterpret_cast<const float4*>(sbp + 1024);
            const f32x2 lo = {acc3[s3][4 * g4], acc3[s3][4 * g4 + 1]}, hi = {acc3[s3][4 * g4 + 2], acc3[s3][4 * g4 + 3]};
            const f32x2 dlo = {accd[s3][4 * g4], accd[s3][4 * g4 + 1]}, dhi = {accd[s3][4 * g4 + 2], accd[s3][4 * g4 + 3]};
            // the skip path as the per-op chain stores it: bf16(sd * acc + bd)
            const unsigned ilo = pack2(dlo * f32x2{sdv_.x, sdv_.y} + f32x2{bdv_.x, bdv_.y});
            const unsigned ihi = pack2(dhi * f32x2{sdv_.z, sdv_.w} + f32x2{bdv_.z, bdv_.w});
            uint2 o;
            o.x = pack2(relu2(lo * f32x2{s3v_.x, s3v_.y} + f32x2{b3v_.x, b3v_.y} + widen2(ilo)));
            o.y = pack2(relu2(hi * f32x2{s3v_.z, s3v_.w} + f32x2{b3v_.z, b3v_.w} + widen2(ihi)));
            nibs |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
            *reinterpret_cast<uint2*>(YT + q * YP + wave * 32 + 8 * g4 + 4 * hf) = o;
          }
          if (a.bits_out) {
            const unsigned other = (unsigned)__shfl_xor((int)nibs, 32);
            if (hf == 0) *reinterpret_cast<unsigned*>(BITS + q * 32 + wave * 4) = nibs | other;
          }
        }
      }
    }
    __syncthreads();

    // ---- D .. E: out (all waves); the moving waves then put the next in tile into LDS (XT's last readers were stage 3)
    {
      const int t = opaque(tid);
#pragma unroll 4
      for (int u = t; u < NP * 32; u += 512) {
        const int q = u >> 5, c = (u & 31) * 8;
        if ((q >> 4) < rows_in) *reinterpret_cast<uint4*>(a.out + (pix0 + q) * 256 + c) = *reinterpret_cast<const uint4*>(YT + q * YP + c);
      }
      if (a.b_out)
        for (int u = t; u < NP * 8; u += 512) {
          const int q = u >> 3, c = (u & 7) * 8;
          if ((q >> 4) < rows_in) *reinterpret_cast<uint4*>(a.b_out + (pix0 + q) * 64 + c) = *reinterpret_cast<const uint4*>(BT + q * AP + c);
        }
      if (a.bits_out && t < 256 && (t >> 5) < rows_in) reinterpret_cast<uint4*>(a.bits_out + pix0 * 32)[t] = reinterpret_cast<const uint4*>(BITS)[t];
      if (a.bbits_out && t >= 256 && t < 320 && ((t - 256) >> 3) < rows_in)
        reinterpret_cast<uint4*>(a.bbits_out + pix0 * 8)[t - 256] = reinterpret_cast<const uint4*>(MA)[t - 256];
      if (more && !comp) put();
    }
    __syncthreads();
  }
}

static int bneck0_launch(Bneck0Args& a, hipStream_t s) {
  static_assert(B0::TOTAL <= 160 * 1024, "LDS");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck0_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B0::TOTAL);
    if (e != hipSuccess) {
      set_error("bneck0_fwd: hipFuncSetAttribute(%zu B LDS) failed: %s", (size_t)B0::TOTAL, hipGetErrorString(e));
      return 1;
    }
    attr = true;
  }
  const int nst = a.B * ((a.H + B0::R - 1) / B0::R);
  a.spw = (nst + 255) / 256;
  hipLaunchKernelGGL(bneck0_fwd_kernel, dim3((nst + a.spw - 1) / a.spw), dim3(512), B0::TOTAL, s, a);
  return check_launch("bneck0_fwd");
}

// ---------------------------------------------------------------------------------------------------------------- layer2, block 0
// The first Bottleneck of layer2 (256 -> 128 at full resolution, 3x3 STRIDE 2 -> half resolution, 128 -> 512, with the stride-2 1x1
// projection 256 -> 512 on the skip path) in ONE forward launch.  A workgroup walks strips of 4 OUTPUT rows (8 columns: one slab of 32
// pixels); the in tile is the 9 input rows x 16 columns under them (144 pixels x 256 channels, 76 KB), the first intermediate a
// zero-bordered 9 x 18 tile whose stride-2 views are the taps, and the out tile re-uses that tile's memory once the 3x3 is done.
// Roles, barriers and the register pool as in bneck_kernel; stage 3 streams its weights (conv3 + the projection: 24 fragments per output
// tile, two tiles per wave) in chunks of 8 through the same cur / alt registers.  The backward of this block stays per-op.
struct Bneck2Args {
  const bf16_t* in;                  // x [B*H*16][256]
  bf16_t* out;                       // y [B*H2*8][512], H2 = (H - 1) / 2 + 1
  const u32x4* w1; const u32x4* w2; const u32x4* w3; const u32x4* wd;      // [128][256], [128][1152], [512][128], [512][256] fragment-major
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3; const float* sd; const float* bd;
  bf16_t* a_out;                     // [B*H*16][128] or null
  bf16_t* b_out;                     // [B*H2*8][128] or null
  uint8_t* bits_out;                 // sign bits of y [B*H2*8][64] or null
  int B, H, H2, spw;
};

struct B2 {
  static constexpr int CI = 256, P = 128, C = 512, WI = 16, WO = 8, RO = 4, RI = 2 * RO + 1;
  static constexpr int NPI = RI * WI, NPO = RO * WO, NS1 = (NPI + 31) / 32;                   // 144 in pixels (5 slabs, the last half), 32 out
  static constexpr int XP = CI + 8, AP = P + 8, AW = WI + 2, YP = C + 8;
  static constexpr size_t XT = 0;
  static constexpr size_t AT = XT + (size_t)NPI * XP * 2;                                    // [9][18][136]; the out tile [32][520] later
  static constexpr size_t BT = AT + (size_t)RI * AW * AP * 2;
  static constexpr size_t BITS = BT + (size_t)NPO * AP * 2;
  static constexpr size_t SB = BITS + (size_t)NPO * (C / 8);
  static constexpr size_t TOTAL = SB + (4 * P + 4 * C) * 4;
  static_assert((size_t)NPO * YP * 2 <= (size_t)RI * AW * AP * 2, "the out tile fits in the 3x3 input tile");
};

__global__ __launch_bounds__(512) void bneck2_fwd_kernel(const Bneck2Args a) {
  constexpr int NPI = B2::NPI, NPO = B2::NPO, NS1 = B2::NS1, RO = B2::RO, XP = B2::XP, AP = B2::AP, AW = B2::AW, YP = B2::YP, P = B2::P, C = B2::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XT = reinterpret_cast<bf16_t*>(smem + B2::XT);
  bf16_t* AT = reinterpret_cast<bf16_t*>(smem + B2::AT);
  bf16_t* YT = AT;                                              // (stage 3 on: the 3x3 input tile is dead)
  bf16_t* BT = reinterpret_cast<bf16_t*>(smem + B2::BT);
  uint8_t* BITS = smem + B2::BITS;                              // [32][64]
  float* SB = reinterpret_cast<float*>(smem + B2::SB);          // s1 b1 s2 b2 (128 each) s3 b3 sd bd (512 each)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int strips = (a.H2 + RO - 1) / RO, nst = a.B * strips;
  const int first = blockIdx.x * a.spw, last = min(first + a.spw, nst);
  if (first >= last) return;
  const bool comp = wave < 4;

  // registers: cur = pool[0..7], alt = pool[8..15] (everybody: stage 3 streams through them too); pool[16..35] is the moving waves' next
  // in tile (18 pieces) AND the compute waves' five stage-1 accumulators - a wave is one or the other for life
  u32x4 pool[36];
  f32x16(&acc1)[5] = *reinterpret_cast<f32x16(*)[5]>(&pool[16]);
  u32x4(&cur)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[0]);
  u32x4(&alt)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[8]);
  const u32x4* w1p = a.w1 + (long)wave * 16 * 64;               // compute waves: stage 1 / 2 tile = wave
  const u32x4* w2p = a.w2 + (long)wave * 72 * 64;
  if (comp) slab::load_chunk<1>(cur, w1p, 0, 0, tid & 63);
  slab::issue_fence();

  auto geom = [&](int s, int& clip, int& r0o) {
    clip = s / strips;
    r0o = (s % strips) * RO;
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  auto fetch = [&](int s) {
    int clip, r0o;
    geom(s, clip, r0o);
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int u = mt + q * 256, p = u >> 5, c = (u & 31) * 8, gr = 2 * r0o - 1 + (p >> 4);
      pool[16 + q] = (gr >= 0 && gr < a.H) ? *reinterpret_cast<const u32x4*>(a.in + (((long)clip * a.H + gr) * 16 + (p & 15)) * 256 + c) : zero4;
    }
  };
  auto put = [&]() {
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      const int u = mt + q * 256, p = u >> 5, c = (u & 31) * 8;
      *reinterpret_cast<u32x4*>(XT + p * XP + c) = pool[16 + q];
    }
  };

  for (int u = tid; u < B2::RI * AW * AP / 8; u += 512) reinterpret_cast<uint4*>(AT)[u] = make_uint4(0, 0, 0, 0);
  for (int u = tid; u < 4 * P + 4 * C; u += 512) {
    const float* src = u < P ? a.s1 + u : u < 2 * P ? a.b1 + (u - P) : u < 3 * P ? a.s2 + (u - 2 * P) : u < 4 * P ? a.b2 + (u - 3 * P)
                       : u < 4 * P + C ? a.s3 + (u - 4 * P) : u < 4 * P + 2 * C ? a.b3 + (u - 4 * P - C)
                       : u < 4 * P + 3 * C ? a.sd + (u - 4 * P - 2 * C) : a.bd + (u - 4 * P - 3 * C);
    SB[u] = *src;
  }
  if (!comp) {
    fetch(first);
    put();
  }
  __syncthreads();

  for (int s = first; s < last; ++s) {
    int clip, r0o;
    geom(s, clip, r0o);
    const int rows_out = min(RO, a.H2 - r0o);
    const long opix0 = ((long)clip * a.H2 + r0o) * 8;            // first out pixel
    const bool more = s + 1 < last;
    const int lane = opaque(tid) & 63, n = lane & 31, hf = lane >> 5, mt = opaque(tid) - 256;

    // ---- A .. B: stage 1 (1x1, 256 -> 128 on the 144 in pixels: tile = wave, all five slabs) | move: the next in tile into registers
    if (comp) {
      f32x16(&acc)[5] = acc1;
      slab::zero_acc<5>(acc);
      gemm_slabs<5, 16>(acc, XT, XP, w1p, lane, cur, alt, [&](u32x4(&d)[8]) {
        load4(d, 0, w2p, lane);                                  // units 0, 1 of the 3x3 (land in cur)
        load4(d, 4, w2p + 4 * 64, lane);
      });
      load4(alt, 0, w2p + 8 * 64, lane);                         // units 2, 3
      load4(alt, 4, w2p + 12 * 64, lane);
      slab::issue_fence();
#pragma unroll
      for (int s3 = 0; s3 < 5; ++s3) {
        const int p = s3 * 32 + n, trow = p >> 4, pc = p & 15, gr = 2 * r0o - 1 + trow;
        const bool inimg = gr >= 0 && gr < a.H;
        bf16_t* dst = AT + (trow * AW + pc + 1) * AP + wave * 32 + 4 * hf;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const f32x2 lo = {acc[s3][4 * g4], acc[s3][4 * g4 + 1]}, hi = {acc[s3][4 * g4 + 2], acc[s3][4 * g4 + 3]};
          const float4 sc = *reinterpret_cast<const float4*>(SB + wave * 32 + 8 * g4 + 4 * hf);
          const float4 bi = *reinterpret_cast<const float4*>(SB + P + wave * 32 + 8 * g4 + 4 * hf);
          uint2 o;
          o.x = pack2(relu2(lo * f32x2{sc.x, sc.y} + f32x2{bi.x, bi.y}));
          o.y = pack2(relu2(hi * f32x2{sc.z, sc.w} + f32x2{bi.z, bi.w}));
          if (!inimg) o = make_uint2(0, 0);
          if (p < NPI) *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
        }
      }
    } else if (more) {
      fetch(s + 1);
    }
    __syncthreads();

    // ---- B .. C: stage 2 (3x3 stride 2, 128 -> 128: tile = wave, the strip's one slab) | move: the first intermediate out
    if (comp) {
      f32x16 acc[1];
      slab::zero_acc<1>(acc);
      const bf16_t* ctr = AT + ((2 * (n >> 3) + 1) * AW + 2 * (n & 7) + 1) * AP + 8 * hf;      // out pixel n <-> in (2 ro, 2 co), padded
#pragma unroll
      for (int u = 0; u < 18; ++u) {
        u32x4(&src)[8] = (u & 2) ? alt : cur;
        const int so = (u & 1) * 4;
        const int tap = u >> 1, kk0 = (u & 1) * 4;
        const int off = ((tap / 3 - 1) * AW + (tap % 3 - 1)) * AP + kk0 * 16;
        bf16x8 xb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xb[kk] = *reinterpret_cast<const bf16x8*>(ctr + off + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[so + kk]), xb[kk], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (u + 4 < 18) load4(src, so, w2p + (long)(u + 4) * 4 * 64, lane);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 sc = *reinterpret_cast<const float4*>(SB + 2 * P + wave * 32 + 8 * g4 + 4 * hf);
        const float4 bi = *reinterpret_cast<const float4*>(SB + 3 * P + wave * 32 + 8 * g4 + 4 * hf);
        const f32x2 lo = {acc[0][4 * g4], acc[0][4 * g4 + 1]}, hi = {acc[0][4 * g4 + 2], acc[0][4 * g4 + 3]};
        uint2 o;
        o.x = pack2(relu2(lo * f32x2{sc.x, sc.y} + f32x2{bi.x, bi.y}));
        o.y = pack2(relu2(hi * f32x2{sc.z, sc.w} + f32x2{bi.z, bi.w}));
        *reinterpret_cast<uint2*>(BT + n * AP + wave * 32 + 8 * g4 + 4 * hf) = o;
      }
    } else if (a.a_out) {
      // in rows 2 r0o .. 2 r0o + 7 = tile rows 1 .. 8 (tile row 0 is the previous strip's row 8): every image row exactly once
      for (int u = mt; u < 128 * 16; u += 256) {
        const int p = (u >> 4) + 16, c = (u & 15) * 8, gr = 2 * r0o - 1 + (p >> 4);
        if (gr < a.H && (gr >> 1) < a.H2 && ((p >> 4) - 1) < 2 * rows_out)
          *reinterpret_cast<uint4*>(a.a_out + (((long)clip * a.H + gr) * 16 + (p & 15)) * P + c) =
              *reinterpret_cast<const uint4*>(AT + ((p >> 4) * AW + (p & 15) + 1) * AP + c);
      }
    }
    // stage 3's first chunk (conv3 of this wave's first tile) in flight across the barrier
    slab::load_chunk<1>(cur, a.w3 + (long)(2 * wave) * 8 * 64, 0, 0, lane);
    __syncthreads();

    // ---- C .. D: stage 3 (all waves, tiles 2 wave + {0, 1}): conv3 over b + the stride-2 projection over x, ReLU; out tile in AT's memory
    {
      const bf16_t* brow = BT + n * AP + 8 * hf;
      const bf16_t* xrow = XT + ((2 * (n >> 3) + 1) * 16 + 2 * (n & 7)) * XP + 8 * hf;      // x at (2 ro, 2 co)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tile = 2 * wave + t;
        f32x16 acc3[1], accd[1];
        slab::zero_acc<1>(acc3);
        slab::zero_acc<1>(accd);
        // chunks of this tile: conv3 (8 k-steps) in cur/alt[t & 1 ? alt : cur], projection k-steps 0..7, 8..15 in the other two turns
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int turn = t * 3 + c;                            // 0..5: even turns sit in cur, odd in alt
          u32x4(&src)[8] = (turn & 1) ? alt : cur;
          u32x4(&dst)[8] = (turn & 1) ? cur : alt;
          // the next turn's chunk: conv3 of the next tile after this tile's projection; the next strip's first stage-1 chunk at the very end
          if (c < 2) slab::load_chunk<1>(dst, a.wd + (long)tile * 16 * 64, 0, c * 8, lane);
          else if (t == 0) slab::load_chunk<1>(dst, a.w3 + (long)(tile + 1) * 8 * 64, 0, 0, lane);
          else if (comp) slab::load_chunk<1>(dst, w1p, 0, 0, lane);      // (turn 5 sits in alt: cur takes the next strip's first chunk)
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (c == 0) {
              const bf16x8 xb = *reinterpret_cast<const bf16x8*>(brow + u * 16);
              acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[u]), xb, acc3[0], 0, 0, 0);
            } else {
              const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xrow + ((c - 1) * 8 + u) * 16);
              accd[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, src[u]), xb, accd[0], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        unsigned nibs = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const float* sbp = SB + 4 * P + tile * 32 + 8 * g4 + 4 * hf;
          const float4 s3v = *reinterpret_cast<const float4*>(sbp), b3v = *reinterpret_cast<const float4*>(sbp + C);
          const float4 sdv = *reinterpret_cast<const float4*>(sbp + 2 * C), bdv = *reinterpret_cast<const float4*>(sbp + 3 * C);
          const f32x2 lo = {acc3[0][4 * g4], acc3[0][4 * g4 + 1]}, hi = {acc3[0][4 * g4 + 2], acc3[0][4 * g4 + 3]};
          const f32x2 dlo = {accd[0][4 * g4], accd[0][4 * g4 + 1]}, dhi = {accd[0][4 * g4 + 2], accd[0][4 * g4 + 3]};
          const unsigned ilo = pack2(dlo * f32x2{sdv.x, sdv.y} + f32x2{bdv.x, bdv.y});      // the skip path as the per-op chain stores it
          const unsigned ihi = pack2(dhi * f32x2{sdv.z, sdv.w} + f32x2{bdv.z, bdv.w});
          uint2 o;
          o.x = pack2(relu2(lo * f32x2{s3v.x, s3v.y} + f32x2{b3v.x, b3v.y} + widen2(ilo)));
          o.y = pack2(relu2(hi * f32x2{s3v.z, s3v.w} + f32x2{b3v.z, b3v.w} + widen2(ihi)));
          nibs |= nibble4(o.x, o.y) << (8 * g4 + 4 * hf);
          *reinterpret_cast<uint2*>(YT + n * YP + tile * 32 + 8 * g4 + 4 * hf) = o;
        }
        if (a.bits_out) {
          const unsigned other = (unsigned)__shfl_xor((int)nibs, 32);
          if (hf == 0) *reinterpret_cast<unsigned*>(BITS + n * 64 + tile * 4) = nibs | other;
        }
      }
    }
    __syncthreads();

    // ---- D .. E: out (all waves)
    {
      const int t = opaque(tid);
      for (int u = t; u < NPO * 64; u += 512) {
        const int q = u >> 6, c = (u & 63) * 8;
        if ((q >> 3) < rows_out) *reinterpret_cast<uint4*>(a.out + (opix0 + q) * C + c) = *reinterpret_cast<const uint4*>(YT + q * YP + c);
      }
      if (a.b_out) {
        const int q = t >> 4, c = (t & 15) * 8;                  // 32 pixels x 16 pieces = 512 threads
        if ((q >> 3) < rows_out) *reinterpret_cast<uint4*>(a.b_out + (opix0 + q) * P + c) = *reinterpret_cast<const uint4*>(BT + q * AP + c);
      }
      if (a.bits_out && t < 128 && (t >> 5) < rows_out) reinterpret_cast<uint4*>(a.bits_out + opix0 * 64)[t] = reinterpret_cast<const uint4*>(BITS)[t];
    }
    if (more) {
      __syncthreads();
      // ---- E .. A: the out tile is gone: the zero border of the 3x3 input tile back (its interior is rewritten by the next stage 1),
      // the next in tile into LDS
      for (int u = tid; u < B2::RI * 2 * (AP / 8); u += 512) {
        const int r = u / (2 * (AP / 8)), w = u % (2 * (AP / 8)), side = w / (AP / 8), c = (w % (AP / 8)) * 8;
        *reinterpret_cast<uint4*>(AT + (r * AW + side * (AW - 1)) * AP + c) = make_uint4(0, 0, 0, 0);
      }
      if (!comp) put();
      __syncthreads();
    }
  }
}

static int bneck2_launch(Bneck2Args& a, hipStream_t s) {
  static_assert(B2::TOTAL <= 160 * 1024, "LDS");
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck2_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2::TOTAL);
    if (e != hipSuccess) {
      set_error("bneck2_fwd: hipFuncSetAttribute(%zu B LDS) failed: %s", (size_t)B2::TOTAL, hipGetErrorString(e));
      return 1;
    }
    attr = true;
  }
  const int nst = a.B * ((a.H2 + B2::RO - 1) / B2::RO);
  a.spw = (nst + 255) / 256;
  hipLaunchKernelGGL(bneck2_fwd_kernel, dim3((nst + a.spw - 1) / a.spw), dim3(512), B2::TOTAL, s, a);
  return check_launch("bneck2_fwd");
}

typedef BG<256, 64, 16> BG1;          // layer1
typedef BG<512, 128, 8> BG2;          // layer2

static int bneck_geom(int cin, int planes, int W) { return (cin == 256 && planes == 64 && W == 16) ? 1 : (cin == 512 && planes == 128 && W == 8) ? 2 : 0; }

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_bneck_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype) {
  return dtype == SEDT_BF16 && bneck_geom(cin, planes, W) != 0 && stride == 1 && dil == 1 && !has_downsample;
}

extern "C" int sedt_bneck_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const float* s1,
                              const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, void* a_out, void* b_out,
                              uint8_t* abits_out, uint8_t* bbits_out, uint8_t* bits_out, int cin, int planes, int W, int B, int H, void* stream) {
  SEDT_REQUIRE(x && y && w1_frag && w2_frag && w3_frag && s1 && b1 && s2 && b2 && s3 && b3, "bneck_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck_fwd: B = %d, H = %d", B, H);
  SEDT_REQUIRE((a_out == nullptr) == (b_out == nullptr) && (abits_out == nullptr) == (bbits_out == nullptr),
               "bneck_fwd: the two intermediates (their sign bits) come both or not at all");
  const int g = bneck_geom(cin, planes, W);
  SEDT_REQUIRE(g != 0, "bneck_fwd: cin %d, planes %d, W %d outside the envelope (256/64/16, 512/128/8)", cin, planes, W);
  BneckArgs a{};
  a.in = (const bf16_t*)x; a.out = (bf16_t*)y;
  a.wA = (const u32x4*)w1_frag; a.wB = (const u32x4*)w2_frag; a.wC = (const u32x4*)w3_frag;
  a.sA = s1; a.bA = b1; a.sB = s2; a.bB = b2; a.sC = s3; a.bC = b3;
  a.a_out = (bf16_t*)a_out; a.b_out = (bf16_t*)b_out; a.abits_out = abits_out; a.bbits_out = bbits_out; a.bits_out = bits_out;
  a.B = B; a.H = H;
  return g == 1 ? bneck_launch<BG1, false>(a, reinterpret_cast<hipStream_t>(stream), "bneck_fwd")
                : bneck_launch<BG2, false>(a, reinterpret_cast<hipStream_t>(stream), "bneck_fwd");
}

extern "C" int sedt_bneck_bwd(const void* gy, void* gx, const void* w3t_frag, const void* w2t_frag, const void* w1t_frag, const uint8_t* abits,
                              const uint8_t* bbits, const uint8_t* xbits, void* gb_out, void* ga_out, int cin, int planes, int W, int B, int H,
                              void* stream) {
  const bool skip3 = gx == nullptr;                              // the chain stops at ga: the caller's block has another first convolution
  SEDT_REQUIRE(gy && w3t_frag && w2t_frag && abits && bbits && (skip3 || w1t_frag), "bneck_bwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck_bwd: B = %d, H = %d", B, H);
  SEDT_REQUIRE(skip3 ? ga_out != nullptr : (gb_out == nullptr) == (ga_out == nullptr),
               "bneck_bwd: the two intermediate gradients come both or not at all (ga alone when gx is null)");
  const int g = bneck_geom(cin, planes, W);
  SEDT_REQUIRE(g != 0, "bneck_bwd: cin %d, planes %d, W %d outside the envelope (256/64/16, 512/128/8)", cin, planes, W);
  BneckArgs a{};
  a.in = (const bf16_t*)gy; a.out = (bf16_t*)gx;
  a.wA = (const u32x4*)w3t_frag; a.wB = (const u32x4*)w2t_frag; a.wC = (const u32x4*)w1t_frag;
  a.abits_in = abits; a.bbits_in = bbits; a.bits_in = xbits;
  a.a_out = (bf16_t*)gb_out; a.b_out = (bf16_t*)ga_out;          // (stage 1 of the chain produces gb, stage 2 ga)
  a.B = B; a.H = H;
  if (skip3) {
    SEDT_REQUIRE(g == 1, "bneck_bwd: the chain-only form exists for the layer1 geometry");
    a.bits_in = nullptr;
    return bneck_launch<BG1, true, true>(a, reinterpret_cast<hipStream_t>(stream), "bneck_bwd");
  }
  return g == 1 ? bneck_launch<BG1, true>(a, reinterpret_cast<hipStream_t>(stream), "bneck_bwd")
                : bneck_launch<BG2, true>(a, reinterpret_cast<hipStream_t>(stream), "bneck_bwd");
}

extern "C" int sedt_bneck0_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype) {
  return dtype == SEDT_BF16 && cin == 64 && planes == 64 && W == 16 && stride == 1 && dil == 1 && has_downsample;
}

extern "C" int sedt_bneck0_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const void* wd_frag,
                               const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* sd,
                               const float* bd, void* a_out, void* b_out, uint8_t* abits_out, uint8_t* bbits_out, uint8_t* bits_out, int B, int H,
                               void* stream) {
  SEDT_REQUIRE(x && y && w1_frag && w2_frag && w3_frag && wd_frag && s1 && b1 && s2 && b2 && s3 && b3 && sd && bd, "bneck0_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck0_fwd: B = %d, H = %d", B, H);
  SEDT_REQUIRE((a_out == nullptr) == (b_out == nullptr) && (abits_out == nullptr) == (bbits_out == nullptr),
               "bneck0_fwd: the two intermediates (their sign bits) come both or not at all");
  Bneck0Args a{};
  a.in = (const bf16_t*)x; a.out = (bf16_t*)y;
  a.w1 = (const u32x4*)w1_frag; a.w2 = (const u32x4*)w2_frag; a.w3 = (const u32x4*)w3_frag; a.wd = (const u32x4*)wd_frag;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.sd = sd; a.bd = bd;
  a.a_out = (bf16_t*)a_out; a.b_out = (bf16_t*)b_out; a.abits_out = abits_out; a.bbits_out = bbits_out; a.bits_out = bits_out;
  a.B = B; a.H = H;
  return bneck0_launch(a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sedt_bneck2_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype) {
  return dtype == SEDT_BF16 && cin == 256 && planes == 128 && W == 16 && stride == 2 && dil == 1 && has_downsample;
}

extern "C" int sedt_bneck2_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const void* wd_frag,
                               const float* s1, const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* sd,
                               const float* bd, void* a_out, void* b_out, uint8_t* bits_out, int B, int H, void* stream) {
  SEDT_REQUIRE(x && y && w1_frag && w2_frag && w3_frag && wd_frag && s1 && b1 && s2 && b2 && s3 && b3 && sd && bd, "bneck2_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck2_fwd: B = %d, H = %d", B, H);
  SEDT_REQUIRE((a_out == nullptr) == (b_out == nullptr), "bneck2_fwd: the two intermediates come both or not at all");
  Bneck2Args a{};
  a.in = (const bf16_t*)x; a.out = (bf16_t*)y;
  a.w1 = (const u32x4*)w1_frag; a.w2 = (const u32x4*)w2_frag; a.w3 = (const u32x4*)w3_frag; a.wd = (const u32x4*)wd_frag;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.sd = sd; a.bd = bd;
  a.a_out = (bf16_t*)a_out; a.b_out = (bf16_t*)b_out; a.bits_out = bits_out;
  a.B = B; a.H = H; a.H2 = (H - 1) / 2 + 1;
  return bneck2_launch(a, reinterpret_cast<hipStream_t>(stream));
}
