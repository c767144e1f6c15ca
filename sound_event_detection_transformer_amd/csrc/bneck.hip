// bneck.hip - an identity Bottleneck of ResNet layer1 (torchvision v1.5 block behind reference sedt/backbone.py:97-113: 1x1 256 -> 64,
// 3x3 64 -> 64, 1x1 64 -> 256, FrozenBatchNorm after each, residual + ReLU) in ONE launch, and its input-gradient chain in one more.
//
// The per-op path moves every intermediate through HBM: at B = 64 (128,000 pixels of a 125 x 16 map) a block reads x (65 MB) twice,
// writes and re-reads a and b (16 MB each) and writes y (65 MB) in three launches, ~70 us forward and ~79 us for the input gradients,
// each launch HBM-bound on its own.  Here a workgroup owns a strip of R image rows of one clip (16 columns wide = R/2 slabs of 32
// pixels, csrc/slab.h): the x tile with one halo row above and below sits in LDS once - it is the B operand of the first 1x1 AND the
// residual of the last - the two 64-channel intermediates never leave the CU (LDS tiles; the 3x3 reads its nine shifted views of a
// zero-padded tile), and only the weights stream from L2 (fragment-major, sedt_pack_frag).  HBM traffic per pixel: 512 B in (+ 2/R
// halo) + 512 B out (+ 256 B of a, b and 32 B of sign bits when the backward will need them) against 1,792 B.
//
// The input-gradient chain has the same shape with the weights transposed (gy -> 1x1 256 -> 64 masked by [b > 0] -> 3x3 with the taps
// mirrored, masked by [a > 0] -> 1x1 64 -> 256 + gy, masked by the sign bits of the block input), so ONE kernel template serves both;
// the FrozenBN scales are folded into the transposed weights exactly as for the per-op dgrad kernels.  layer1 is frozen in the
// reference (backbone.py:60-62): no weight gradients are needed there; a trainable block keeps the per-op backward.
// Rounding points are those of the per-op chain: a, b (their gradients) rounded to bf16, accumulation in f32.
#include "slab.h"

namespace sedt {

using slab::u32x4;
using slab::XP;

constexpr int BN_W = 16;                  // map width
constexpr int BN_C = 256, BN_P = 64;      // block channels, bottleneck planes
constexpr int BN_AP = BN_P + 8;           // element pitch of the 64-channel tiles: 144 B, conflict-free 16-byte fragment reads
constexpr int BN_AW = BN_W + 2;           // a row of the zero-padded 3x3 input tile

struct BneckArgs {
  const bf16_t* in;                  // x (forward) / gy (backward) [B*H*16][256]
  bf16_t* out;                       // y / gx
  const u32x4* wA;                   // [64][256]  conv1 (fwd) / (s3 . conv3)^T (bwd), fragment-major
  const u32x4* wB;                   // [64][9*64] conv2, k = tap * 64 + channel
  const u32x4* wC;                   // [256][64]  conv3 (fwd) / (s1 . conv1)^T (bwd)
  const float* sA; const float* bA; const float* sB; const float* bB; const float* sC; const float* bC;   // folded BN (fwd)
  bf16_t* a_out; bf16_t* b_out;      // fwd by-products [M][64] or null (what a per-op backward with weight gradients reads)
  uint8_t* abits_out; uint8_t* bbits_out;      // fwd: sign bits of a, b [M][8] or null (ALL the fused backward needs of them)
  uint8_t* bits_out;                 // fwd: sign bits of y [M][32] or null
  const uint8_t* abits_in; const uint8_t* bbits_in;      // bwd: sign bits of the saved a, b
  const uint8_t* bits_in;            // bwd: sign bits of the block input [M][32], or null (no mask)
  int B, H, spw;                     // spw: consecutive strips per workgroup
  int dbg;                           // developer build: phase ablation (1 stage 1, 2 stage 2, 4 stage 3, 8 stores, 16 tile loads)
};

constexpr int BN_R = 8;                                           // image rows of a strip
constexpr int BN_NP1 = (BN_R + 2) * BN_W, BN_NP = BN_R * BN_W;    // pixels of the halo tile / of the strip: 160 / 128
constexpr int BN_NS1 = BN_NP1 / 32, BN_NS = BN_NP / 32;           // their 32-pixel slabs: 5 / 4

template <bool BWD>
struct BneckLds {
  static constexpr size_t XT = 0;
  static constexpr size_t AT = XT + (size_t)BN_NP1 * XP * 2;
  static constexpr size_t BT = AT + (size_t)(BN_R + 2) * BN_AW * BN_AP * 2;
  static constexpr size_t BITS = BT + (size_t)BN_NP * BN_AP * 2;                // [NP][32]
  static constexpr size_t MH = BITS + (size_t)BN_NP * 32;                       // [NP1][8] sign bits on the halo tile: of a (fwd, out) / b (bwd, in)
  static constexpr size_t MA = MH + (size_t)BN_NP1 * 8;                         // [NP][8] sign bits on the strip: of b (fwd, out) / a (bwd, in)
  static constexpr size_t SB = MA + (size_t)BN_NP * 8;
  static constexpr size_t TOTAL = SB + (BWD ? 0 : 768 * 4);
};

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

// four fragments (one tap of the 3x3, or a 64-deep 1x1) into dst[o .. o + 3]
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

// sign bits of 8 bf16 values (a 16-byte piece): bit e <-> element e > 0
__device__ __forceinline__ unsigned sign_byte(u32x4 v) {
  const unsigned w[4] = {v[0], v[1], v[2], v[3]};
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned lo = w[i] & 0xffffu, hi = w[i] >> 16;
    m |= ((lo != 0u && lo < 0x8000u) ? 1u : 0u) << (2 * i);
    m |= ((hi != 0u && hi < 0x8000u) ? 1u : 0u) << (2 * i + 1);
  }
  return m;
}

// The workgroup walks `spw` consecutive strips, one workgroup per CU (the tile is 84 KB of LDS).  What the measurements of the earlier
// versions say (profiles/r04_bneck_ablation.txt): a strip is bound by the L2 -> CU weight stream (~40 B/clk/CU, csrc/slab.h), not by HBM
// and not by the MFMAs - so a weight fragment has to serve as many pixels as the accumulators allow (8-row strips: 2-3 slabs per
// fragment; the 4-row version with two tile buffers streamed twice the bytes per pixel and took 8.6 us per 64 pixels) - and HBM latency
// must never sit inside a stage.  Waves 0..3 COMPUTE stages 1 and 2 (one per SIMD; the weight stream never stops: the next chunk / tap /
// strip is always in flight) and issue no stores (loads and stores share one in-order counter: a store ahead of a weight fragment would
// hold the MFMA that waits for the fragment until HBM has acknowledged the store); waves 4..7 MOVE: they hold the NEXT strip's tile in
// registers for a whole strip (issued right after the current tile went to LDS) and write the first intermediate out; stage 3 and the
// output stores are everybody's.
//   A(s) .. B(s): compute stage 1 (reads XT, writes AT)             | move: issue the loads of strip s + 1 (registers)
//   B(s) .. C(s): compute stage 2 (reads AT, writes BT)             | move: store a of s (AT)
//   C(s) .. D(s): all: stage 3 (reads BT, XT in place, BITS)
//   D(s) .. E(s): all: store y / b / bits of s (XT, BT, BITS)
//   E(s) .. A(s+1): move: registers -> XT (+ the sign bits of a, b of strip s + 1; bwd)
template <bool BWD>
__global__ __launch_bounds__(512) void bneck_kernel(const BneckArgs a) {
  using LD = BneckLds<BWD>;
  constexpr int NP1 = BN_NP1, NP = BN_NP, NS1 = BN_NS1, NS = BN_NS, R = BN_R;
  static_assert(NS1 == 5 && NS == 4, "the wave assignment below is written for 8-row strips");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* XT = reinterpret_cast<bf16_t*>(smem + LD::XT);        // [NP1][XP]: in tile, halo row first; becomes the out tile
  bf16_t* AT = reinterpret_cast<bf16_t*>(smem + LD::AT);        // [(R + 2)][18][AP]: 3x3 input, zero border
  bf16_t* BT = reinterpret_cast<bf16_t*>(smem + LD::BT);        // [NP][AP]: 3x3 output
  uint8_t* BITS = smem + LD::BITS;                              // [NP][32]: fwd sign bits of y; bwd sign bits of the block input
  uint8_t* MH = smem + LD::MH;                                  // [NP1][8]
  uint8_t* MA = smem + LD::MA;                                  // [NP][8]
  float* SB = reinterpret_cast<float*>(smem + LD::SB);          // fwd: sA bA sB bB (64 each) sC bC (256 each)
  const int tid = threadIdx.x, wave = tid >> 6;
  const int strips = (a.H + R - 1) / R, nst = a.B * strips;
  const int first = blockIdx.x * a.spw, last = min(first + a.spw, nst);
  if (first >= last) return;
  const bool comp = wave < 4;
  const int tl = wave & 1, grp = (wave >> 1) & 1;               // compute waves: output tile of stages 1 / 2, slab group
  constexpr int NX = NP1 * 32 / 256;                            // 16-byte pieces of the in tile per moving thread: 20
  static_assert(NX == 20 && NP1 * 8 == 80 * 16 && NP * 8 == 64 * 16 && NP * 2 == 256, "tile pieces");

  // ONE register pool for both roles (a wave computes or moves for its whole life): the compute waves' weight chunks cur = pool[0..7],
  // alt = pool[8..15]; the moving waves' tile in flight xr = pool[0..19] and (bwd) the three sign-bit pieces pool[20..22]
  u32x4 pool[BWD ? 23 : 20], w3r[4];
  u32x4(&cur)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[0]);
  u32x4(&alt)[8] = *reinterpret_cast<u32x4(*)[8]>(&pool[8]);
  u32x4(&xr)[20] = *reinterpret_cast<u32x4(*)[20]>(&pool[0]);
  const u32x4* w1p = a.wA + (long)tl * 16 * 64;
  const u32x4* w2p = a.wB + (long)tl * 36 * 64;
  load4(w3r, 0, a.wC + (long)wave * 4 * 64, tid & 63);           // stage 3: tile = wave - the same four fragments for every strip
  if (comp) slab::load_chunk<1>(cur, w1p, 0, 0, tid & 63);
  slab::issue_fence();

  auto geom = [&](int s, int& r0, long& pix0) {
    const int clip = s / strips;
    r0 = (s % strips) * R;
    pix0 = ((long)clip * a.H + r0) * BN_W;
  };
  const u32x4 zero4 = {0u, 0u, 0u, 0u}, ones4 = {~0u, ~0u, ~0u, ~0u};
  // the in tile of strip s into the moving threads' registers / from there into LDS
  auto fetch = [&](int s) {
    int r0; long pix0;
    geom(s, r0, pix0);
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int u = mt + q * 256, p = u >> 5, c = (u & 31) * 8, gr = r0 - 1 + (p >> 4);
      xr[q] = (gr >= 0 && gr < a.H) ? *reinterpret_cast<const u32x4*>(a.in + (pix0 + p - BN_W) * BN_C + c) : zero4;
    }
    if (BWD) {                                                   // sign bits: b on the halo tile, a, the block input (16-byte pieces = 2 pixels)
      const int grh = r0 - 1 + (mt >> 3);
      pool[BWD ? 20 : 0] = (mt < 80 && grh >= 0 && grh < a.H) ? reinterpret_cast<const u32x4*>(a.bbits_in + (pix0 - BN_W) * 8)[mt] : zero4;
      pool[BWD ? 21 : 0] = (mt < 64 && r0 + (mt >> 3) < a.H) ? reinterpret_cast<const u32x4*>(a.abits_in + pix0 * 8)[mt] : zero4;
      pool[BWD ? 22 : 0] = !a.bits_in ? ones4 : r0 + (mt >> 5) < a.H ? reinterpret_cast<const u32x4*>(a.bits_in + pix0 * 32)[mt] : zero4;
    }
  };
  auto put = [&]() {
    const int mt = opaque(tid) - 256;
#pragma unroll
    for (int q = 0; q < NX; ++q) {
      const int u = mt + q * 256, p = u >> 5, c = (u & 31) * 8;
      *reinterpret_cast<u32x4*>(XT + p * XP + c) = xr[q];
    }
    if (BWD) {
      if (mt < 80) reinterpret_cast<u32x4*>(MH)[mt] = pool[BWD ? 20 : 0];
      if (mt < 64) reinterpret_cast<u32x4*>(MA)[mt] = pool[BWD ? 21 : 0];
      reinterpret_cast<u32x4*>(BITS)[mt] = pool[BWD ? 22 : 0];
    }
  };

  // ---- prologue: constants, the zero border of the 3x3 input tile, the first strip's tiles
  for (int u = tid; u < (R + 2) * BN_AW * BN_AP / 8; u += 512) reinterpret_cast<uint4*>(AT)[u] = make_uint4(0, 0, 0, 0);
  if (!BWD)
    for (int u = tid; u < 768; u += 512) {
      const float* src = u < 64 ? a.sA + u : u < 128 ? a.bA + (u - 64) : u < 192 ? a.sB + (u - 128) : u < 256 ? a.bB + (u - 192)
                         : u < 512 ? a.sC + (u - 256) : a.bC + (u - 512);
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
      // stage 1: 1x1, 256 -> 64 on the halo tile: tile tl, slabs 3 * grp + {0, 1, 2} (slab 5 does not exist: computed on whatever follows
      // the tile in LDS and dropped)
      f32x16 acc[3];
      slab::zero_acc<3>(acc);
      gemm_slabs<3, 16>(acc, XT + (3 * grp) * 32 * XP, XP, w1p, lane, cur, alt, [&](u32x4(&d)[8]) {
        load4(d, 0, w2p, lane);                                  // taps 0, 1 of the 3x3 (land in cur)
        load4(d, 4, w2p + 4 * 64, lane);
      });
      load4(alt, 0, w2p + 8 * 64, lane);                         // taps 2, 3
      load4(alt, 4, w2p + 12 * 64, lane);
      slab::issue_fence();
      float4 sc[4], bi[4];
      if (!BWD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4*>(SB + tl * 32 + 8 * g4 + 4 * hf);
          bi[g4] = *reinterpret_cast<const float4*>(SB + 64 + tl * 32 + 8 * g4 + 4 * hf);
        }
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const int sl = 3 * grp + s3;
        if (sl < NS1) {
          const int p = sl * 32 + n, trow = p >> 4, pc = p & 15, gr = r0 - 1 + trow;
          const bool inimg = gr >= 0 && gr < a.H;
          bf16_t* dst = AT + (trow * BN_AW + pc + 1) * BN_AP + tl * 32 + 4 * hf;
          unsigned m4 = 0, nb = 0;
          if (BWD) m4 = *reinterpret_cast<const unsigned*>(MH + p * 8 + tl * 4) >> (4 * hf);       // bytes g4 = 0..3 of this tile
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
            *reinterpret_cast<uint2*>(dst + 8 * g4) = o;
          }
          if (!BWD && a.abits_out) {                             // the two lane halves hold the two nibbles of every byte
            const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
            if (hf == 0) *reinterpret_cast<unsigned*>(MH + p * 8 + tl * 4) = nb | other;
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
      // stage 2: 3x3, 64 -> 64: tile tl, slabs 2 * grp + {0, 1}
      f32x16 acc[2];
      slab::zero_acc<2>(acc);
      const bf16_t* ctr[2];                                      // lane n <-> pixel: centre of its 3x3 neighbourhood in the padded tile
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q = (grp * 2 + s2) * 32 + n;
        ctr[s2] = AT + (((q >> 4) + 1) * BN_AW + (q & 15) + 1) * BN_AP + 8 * hf;
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // tap t sits in quarter t % 4 of (cur | alt); three taps are in flight behind it
        u32x4(&src)[8] = (tap & 2) ? alt : cur;
        const int so = (tap & 1) * 4;
        const int dr = tap / 3 - 1, dc = tap % 3 - 1;
        const int off = (BWD ? -(dr * BN_AW + dc) : (dr * BN_AW + dc)) * BN_AP;     // the input gradient mirrors the taps
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
        if (tap + 4 < 9) load4(src, so, w2p + (long)(tap + 4) * 4 * 64, lane);
        if (tap == 8) slab::load_chunk<1>(cur, w1p, 0, 0, lane);  // the next strip's first chunk (cur is free from here)
        __builtin_amdgcn_sched_barrier(0);
      }
      float4 sc[4], bi[4];
      if (!BWD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4*>(SB + 128 + tl * 32 + 8 * g4 + 4 * hf);
          bi[g4] = *reinterpret_cast<const float4*>(SB + 192 + tl * 32 + 8 * g4 + 4 * hf);
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int q = (grp * 2 + s2) * 32 + n;
        unsigned m4 = 0, nb = 0;
        if (BWD) m4 = *reinterpret_cast<const unsigned*>(MA + q * 8 + tl * 4) >> (4 * hf);
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
          *reinterpret_cast<uint2*>(BT + q * BN_AP + ch) = o;
        }
        if (!BWD && a.bbits_out) {
          const unsigned other = (unsigned)__shfl_xor((int)nb, 32);
          if (hf == 0) *reinterpret_cast<unsigned*>(MA + q * 8 + tl * 4) = nb | other;
        }
      }
     }
    } else if (!BWD && !(a.dbg & 8)) {
      if (a.a_out)
        for (int u = mt; u < NP * 8; u += 256) {
          const int q = u >> 3, c = (u & 7) * 8;
          if ((q >> 4) < rows_in)
            *reinterpret_cast<uint4*>(a.a_out + (pix0 + q) * BN_P + c) =
                *reinterpret_cast<const uint4*>(AT + (((q >> 4) + 1) * BN_AW + (q & 15) + 1) * BN_AP + c);
        }
      if (a.abits_out && mt < 64 && (mt >> 3) < rows_in)         // (interior rows of the halo tile: 2 pixels per 16-byte piece)
        reinterpret_cast<uint4*>(a.abits_out + pix0 * 8)[mt] = reinterpret_cast<const uint4*>(MH + BN_W * 8)[mt];
    }
    __syncthreads();

    // ---- C(s) .. D(s): stage 3 (all waves, tile = wave): 1x1, 64 -> 256, + the in tile (residual), ReLU / sign-bit mask, in place
    if (!(a.dbg & 4)) {
      f32x16 acc[NS];
      slab::zero_acc<NS>(acc);
      const bf16_t* xrow = BT + n * BN_AP + 8 * hf;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 xb[NS];
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3) xb[s3] = *reinterpret_cast<const bf16x8*>(xrow + s3 * 32 * BN_AP + kk * 16);
#pragma unroll
        for (int s3 = 0; s3 < NS; ++s3)
          acc[s3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w3r[kk]), xb[s3], acc[s3], 0, 0, 0);
      }
      float4 sc[4], bi[4];
      if (!BWD) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          sc[g4] = *reinterpret_cast<const float4*>(SB + 256 + wave * 32 + 8 * g4 + 4 * hf);
          bi[g4] = *reinterpret_cast<const float4*>(SB + 512 + wave * 32 + 8 * g4 + 4 * hf);
        }
      }
      unsigned nibs[NS];                                         // fwd: this lane's sign nibbles of slab s3, nibble g4 at bits 8 * g4 (+ 4 * hf)
#pragma unroll
      for (int s3 = 0; s3 < NS; ++s3) {
        const int q = s3 * 32 + n;
        bf16_t* xp = XT + (q + BN_W) * XP + wave * 32 + 4 * hf;
        unsigned m4 = 0;
        if (BWD) m4 = *reinterpret_cast<const unsigned*>(BITS + q * 32 + wave * 4) >> (4 * hf);
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
            o.x = pack2(relu2(lo * f32x2{sc[g4].x, sc[g4].y} + f32x2{bi[g4].x, bi[g4].y} + widen2(xw.x)));
            o.y = pack2(relu2(hi * f32x2{sc[g4].z, sc[g4].w} + f32x2{bi[g4].z, bi[g4].w} + widen2(xw.y)));
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
          if (hf == 0) *reinterpret_cast<unsigned*>(BITS + (s3 * 32 + n) * 32 + wave * 4) = nibs[s3] | other;
        }
      }
    }
    __syncthreads();

    // ---- D(s) .. E(s): out (all waves)
    if (!(a.dbg & 8)) {
      const int t = opaque(tid);
#pragma unroll 4
      for (int u = t; u < NP * 32; u += 512) {
        const int q = u >> 5, c = (u & 31) * 8;
        if ((q >> 4) < rows_in) *reinterpret_cast<uint4*>(a.out + (pix0 + q) * BN_C + c) = *reinterpret_cast<const uint4*>(XT + (q + BN_W) * XP + c);
      }
      if (!BWD) {
        if (a.b_out)
          for (int u = t; u < NP * 8; u += 512) {
            const int q = u >> 3, c = (u & 7) * 8;
            if ((q >> 4) < rows_in) *reinterpret_cast<uint4*>(a.b_out + (pix0 + q) * BN_P + c) = *reinterpret_cast<const uint4*>(BT + q * BN_AP + c);
          }
        if (a.bits_out && t < NP * 2 && (t >> 5) < rows_in) reinterpret_cast<uint4*>(a.bits_out + pix0 * 32)[t] = reinterpret_cast<const uint4*>(BITS)[t];
        if (a.bbits_out && t >= 256 && t < 320 && ((t - 256) >> 3) < rows_in)
          reinterpret_cast<uint4*>(a.bbits_out + pix0 * 8)[t - 256] = reinterpret_cast<const uint4*>(MA)[t - 256];
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

template <bool BWD>
static int bneck_launch(BneckArgs& a, hipStream_t s, const char* what) {
  using LD = BneckLds<BWD>;
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bneck_kernel<BWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LD::TOTAL);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute(%zu B LDS) failed: %s", what, (size_t)LD::TOTAL, hipGetErrorString(e));
      return 1;
    }
    attr = true;
  }
  // one workgroup per CU: 256 workgroups walk ceil(strips / 256) consecutive strips each
  const int nst = a.B * ((a.H + BN_R - 1) / BN_R);
  static int spw_env = [] { const char* e = dev_getenv("SEDT_BNECK_SPW"); return e ? atoi(e) : 0; }();
  a.spw = spw_env > 0 ? spw_env : (nst + 255) / 256;
  static int dbg_env = [] { const char* e = dev_getenv("SEDT_BNECK_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg_env;
  hipLaunchKernelGGL((bneck_kernel<BWD>), dim3((nst + a.spw - 1) / a.spw), dim3(512), LD::TOTAL, s, a);
  return check_launch(what);
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_bneck_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype) {
  return dtype == SEDT_BF16 && cin == BN_C && planes == BN_P && W == BN_W && stride == 1 && dil == 1 && !has_downsample;
}

extern "C" int sedt_bneck_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const float* s1,
                              const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, void* a_out, void* b_out,
                              uint8_t* abits_out, uint8_t* bbits_out, uint8_t* bits_out, int B, int H, void* stream) {
  SEDT_REQUIRE(x && y && w1_frag && w2_frag && w3_frag && s1 && b1 && s2 && b2 && s3 && b3, "bneck_fwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck_fwd: B = %d, H = %d", B, H);
  SEDT_REQUIRE((a_out == nullptr) == (b_out == nullptr) && (abits_out == nullptr) == (bbits_out == nullptr),
               "bneck_fwd: the two intermediates (their sign bits) come both or not at all");
  BneckArgs a{};
  a.in = (const bf16_t*)x; a.out = (bf16_t*)y;
  a.wA = (const u32x4*)w1_frag; a.wB = (const u32x4*)w2_frag; a.wC = (const u32x4*)w3_frag;
  a.sA = s1; a.bA = b1; a.sB = s2; a.bB = b2; a.sC = s3; a.bC = b3;
  a.a_out = (bf16_t*)a_out; a.b_out = (bf16_t*)b_out; a.abits_out = abits_out; a.bbits_out = bbits_out; a.bits_out = bits_out;
  a.B = B; a.H = H;
  return bneck_launch<false>(a, reinterpret_cast<hipStream_t>(stream), "bneck_fwd");
}

extern "C" int sedt_bneck_bwd(const void* gy, void* gx, const void* w3t_frag, const void* w2t_frag, const void* w1t_frag, const uint8_t* abits,
                              const uint8_t* bbits, const uint8_t* xbits, int B, int H, void* stream) {
  SEDT_REQUIRE(gy && gx && w3t_frag && w2t_frag && w1t_frag && abits && bbits, "bneck_bwd: null pointer");
  SEDT_REQUIRE(B >= 1 && H >= 1, "bneck_bwd: B = %d, H = %d", B, H);
  BneckArgs a{};
  a.in = (const bf16_t*)gy; a.out = (bf16_t*)gx;
  a.wA = (const u32x4*)w3t_frag; a.wB = (const u32x4*)w2t_frag; a.wC = (const u32x4*)w1t_frag;
  a.abits_in = abits; a.bbits_in = bbits; a.bits_in = xbits;
  a.B = B; a.H = H;
  return bneck_launch<true>(a, reinterpret_cast<hipStream_t>(stream), "bneck_bwd");
}
