// igemm3.hip - the lean-issue version of the bf16 LDS-DMA implicit GEMM (igemm2.hip) for the common cases.
//
// Hardware counters on the layer-4 3x3 convolution showed igemm2 to be INSTRUCTION-ISSUE bound, not memory bound: 12 VALU +
// 14 SALU instructions per MFMA (per-DMA gather arithmetic with 64-bit multiplies, an integer division per K tile to find
// the tap, M0 set-up from a non-uniform wave id, fragment addresses recomputed per k-step), zero LDS bank conflicts and a
// 90 % L2 hit rate.  This kernel keeps the data movement identical (LDS-DMA into an XOR-swizzled 2-stage ring, counted
// waits, raw barriers, LDS-staged epilogue) and strips the issue stream:
//   * per DMA row, ONE 32-bit byte offset of the un-shifted pixel and a bit mask "tap t lands inside the image" are
//     computed before the loop; a K tile adds one wave-uniform (tap, channel) delta held in SGPRs:
//     offset = row_off + delta, valid = mask >> tap & 1  ->  4 VALU per A row, 1 per B row;
//   * (tap, kh, kw, c0) advance incrementally on the scalar unit - no division in the loop;
//   * the wave id is made uniform once (readfirstlane), so every LDS-DMA destination / M0 value is scalar arithmetic;
//   * fragment read offsets are per-thread constants; the loop is unrolled by the ring depth so the stage base is an
//     instruction immediate.
// Envelope: trans == 0, bf16, K % 64 == 0, forward or dgrad gather (square stride) with <= 32 taps; everything else
// (K tails, f32 output, sigmoid) stays on igemm2 / igemm.
#include <stdlib.h>
#include <type_traits>
#include "wgrad3_body.h"

namespace sedt {

__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// The own-code prefetch of igemm3_impl reads up to 32 KB past a point inside a kernel.  Every kernel of this file is followed by other code
// except the last one of .text: this zero-initialised tail (a later segment of the same loaded image, which the loader allocates as ONE
// region) keeps that read inside the image whatever the order of the kernels is.
__device__ __attribute__((used)) unsigned char g_code_tail_pad[64 * 1024];

#ifdef SEDT_DEV
// developer build only: shader-clock stamps of a workgroup's phases (start, K loop entered, K loop left, stores issued) and the constant
// 100 MHz clock at its start, for the first 4096 workgroups of the most recent igemm3 launch (tools/dev/r06_phase_ts.py)
__device__ unsigned long long g_phase_ts[5 * 4096];
__device__ int g_ts_filter[3];      // (M, N, K) of the only problem that records, or M = 0: every launch records (sedt_dev_ts_filter)
// (the filter is read ONCE, before the first stamp: a reload per stamp would charge an L2 miss to every phase of a launch inside a step)
#define SEDT_TS_INIT() \
  const bool ts_on_ = threadIdx.x == 0 && bx < 4096 && (g_ts_filter[0] == 0 || (g_ts_filter[0] == p.M && g_ts_filter[1] == p.N && g_ts_filter[2] == p.K))
#define SEDT_TS(i_, v_)                              \
  do {                                               \
    if (ts_on_) g_phase_ts[bx * 5 + (i_)] = (v_);    \
  } while (0)
#else
#define SEDT_TS_INIT() do {} while (0)
#define SEDT_TS(i_, v_) do {} while (0)
#endif

// S = ring depth (stages); bx = index of this workgroup among the problem's tiles; NW = waves per workgroup: 4, or 8 = two
// groups of four that each own the full output tile but only half of the k16 steps of every K tile (their accumulators are
// added through LDS in the epilogue).  The 8-wave form gives a 128x128 tile - 64 flop per byte fetched from L2 instead of
// 43 for 64x128 - enough waves to hide latency when a problem has only one workgroup per CU (M = 8192, N = 512).
// CONV: gathered A operand (tap walk); false = plain row-major A, whose per-K-tile bookkeeping is ONE scalar add.
// PP (8 waves only): "ping-pong" - the second wave group runs its MFMAs half a stage late, i.e. while the first group reads
// its fragments and issues the next LDS-DMA, and reads its own fragments while the first group's MFMAs run: with one
// workgroup per CU (128x128 tiles at M = 8192, N = 512) nothing else overlaps the LDS phase with the MFMA phase.
// KH = 2 (8 waves, ping-pong only): TWO such 8-wave teams in one 1024-thread workgroup, each with its own ring, each over one half
// of K; the four wave groups' partial sums meet in the epilogue.  For the problems with exactly one 64x128 tile per CU
// (M = 8192, N = 256: layer3, the FFN's second linear) it doubles the waves per SIMD without shrinking the tile.
// BR = 1 (8 waves, ping-pong, BN = 128; round 6, review item 6b): the B operand (a weight) does not pass through LDS at all - every wave
// owns a 32-column strip of the tile over all BM rows and reads its B fragments straight from L2 into registers out of the FRAGMENT-MAJOR
// image of the operand (SedtIgemm.bfrag, the layout of sedt_pack_frag: one contiguous KB per (32 rows, k16 step)), S tiles ahead like the
// ring; the ring holds A only.  LDS-DMA then fills a third (64x128) or half (128x128) of the bytes per K tile.
template <int BM, int BN, int S, int NW, bool CONV, int PP = 0, int KH = 1, int BR = 0>
__device__ __forceinline__ void igemm3_impl(const SedtIgemm& p, const unsigned a_bytes_flag, const unsigned b_bytes_flag, const int bx) {
  const unsigned b_bytes = b_bytes_flag & 0x7fffffffu;      // (bit 31: cooperative weight prefetch on, see below)
  const unsigned a_bytes = a_bytes_flag & 0x7fffffffu;      // (bit 31: the first workgroups prefetch this program's own code, see below)
  constexpr int WM = BR ? BM : BM / 2, WN = BR ? BN / 4 : BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int STAGE_BYTES = (BM + (BR ? 0 : BN)) * ROWB;
  constexpr int GA = BM / (8 * NW), GB = BR ? 0 : BN / (8 * NW);
  constexpr int NT = NW * KH * 64;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(KH == 1 || (KH == 2 && NW == 8 && PP), "two K halves: the ping-pong 8-wave form only");
  static_assert(GA >= 1 && (BR || GB >= 1), "tile too small for the wave count");
  static_assert(!BR || (NW == 8 && PP && KH == 1 && BN == 128), "register-streamed B: the 8-wave ping-pong form, 128 columns");
  constexpr unsigned OOB = 0x80000000u;          // >= 2^31 > num_records; stays out of range after adding any K offset
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  SEDT_TS_INIT();
  SEDT_TS(0, __builtin_amdgcn_s_memtime());
  SEDT_TS(4, __builtin_amdgcn_s_memrealtime());

  const int t = threadIdx.x, lane = t & 63;
  const int wave_all = uniform_i32(t >> 6);
  const int khalf = KH == 2 ? wave_all >> 3 : 0;       // which half of K (and which ring) this wave's team works on
  const int wave = KH == 2 ? (wave_all & 7) : wave_all;
  const int kgrp = wave >> 2;                          // 0, or 0/1 with 8 waves: which half of the k16 steps
  const int wm = BR ? 0 : ((wave & 3) >> 1) * WM, wn = BR ? (wave & 3) * WN : (wave & 1) * WN;
  unsigned char* const ring = smem + khalf * (S * STAGE_BYTES);

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  int vid;
  {
    const int b = bx, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int m0 = (vid / ntn) * BM, n0 = (vid % ntn) * BN;
  const int nkb = p.K / BK2 / KH;                       // K tiles of this team
  const int kb0 = khalf * nkb;                         // its first K tile

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsF0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BR ? p.bfrag : p.B), 0,
                                                                  BR ? (unsigned)((long)p.N * p.ldb * 2) : b_bytes, 0x00020000);

  // ---- cooperative weight prefetch.  Inside a training step the weight operand is cold in the Infinity Cache (packed a whole forward earlier)
  //      while the activations - just written - are not: the K loop of a launch whose workgroups all start together then waits for B k-slice by
  //      k-slice, one HBM round trip after the other (measured with the phase stamps: K loop 35.6 k clocks with B from HBM against 25.8 k with B
  //      resident, 38.2 k inside the step: profiles/r06_ab_breg.txt).  So the first workgroups each TOUCH one slice of the whole B operand before
  //      anything else - one LDS-DMA dword per 128-byte line, 64 lines per instruction - and all of B is on its way from HBM at once.  The
  //      dwords land in the LDS bytes of this wave's own first A piece, which the same wave's real piece - a later load, and loads of a wave
  //      return in order - overwrites.
  // ---- the program's own code.  Inside a step every launch starts with its code cold (197 other kernels and 13 GB of traffic since its last
  //      run): the straight-line prologue and epilogue are fetched from HBM one instruction-cache miss after the other (a 128x128 launch whose
  //      256 workgroups all start together: prologue 3.3 k -> 10.1 k clocks, epilogue 6.7 k -> 13.1 k with cold code and warm data,
  //      tools/dev/r06_phase_ts_step.py).  The first workgroups touch the CODE_KB KB that follow this point - this specialisation's prologue,
  //      loop and epilogue - so that the lines are on their way from HBM together (same LDS-DMA trick as the weight prefetch below).
  constexpr int CODE_KB = BM * BN >= 128 * 128 ? 32 : 24;
  if ((a_bytes_flag >> 31) && wave_all == 0 && bx < CODE_KB / 8) {
    const uint64_t pc = __builtin_amdgcn_s_getpc() & ~(uint64_t)127;
    __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(pc), 0, CODE_KB * 1024u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (lds_void*)(ring), 4, (unsigned)bx * 8192u + (unsigned)lane * 128u, 0, 0, 0);
  }
  if ((b_bytes_flag >> 31) && !p.f32ep && wave_all == 0) {
    const int npf = nwg < 256 ? nwg : 256;
    if (bx < npf) {
      const unsigned total = BR ? (unsigned)((long)p.N * p.ldb * 2) : b_bytes;
      const unsigned chunk = ((total + npf - 1) / npf + 127u) & ~127u;
      const unsigned lo = (unsigned)bx * chunk, hi = lo + chunk < total ? lo + chunk : total;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned off = lo + (unsigned)i * 8192u + (unsigned)lane * 128u;
        if (lo + (unsigned)i * 8192u < hi)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(BR ? rsF0 : rsB, (lds_void*)(ring), 4, off < hi ? off : OOB, 0, 0, 0);
      }
    }
  }

  // ---- epilogue operands (residual, ReLU mask of the consumer) are fetched NOW into registers: the small-K problems of
  //      layer1/layer2 are HBM streams whose per-workgroup time is a chain of memory round trips (A tile -> residual ->
  //      store); issuing the residual/mask loads first overlaps them with the whole K loop.  They are older than every
  //      LDS-DMA, so the counted vmcnt waits of the ring stay valid.
  constexpr int CPR = BN / 8;                          // 16-byte chunks per tile row
  constexpr int NCH = (BM * CPR + NT - 1) / NT;        // chunks per thread
  constexpr bool PARTIAL = BM * CPR % NT != 0;         // more threads than chunks (64x64 tile, 1024 threads): the upper threads idle
  static_assert(!PARTIAL || NCH == 1, "epilogue chunks must divide evenly, or be fewer than the threads");
  // f32ep (the bf16x3 mode, see sedt_split3): C, res and a non-bit mask are f32 tensors; they are read in the epilogue itself (no
  // register prefetch: eight floats per chunk would double this kernel's epilogue registers for a mode that is MFMA-bound anyway)
  const bool f32ep = p.f32ep != 0;
  const bf16_t* resT = f32ep ? nullptr : reinterpret_cast<const bf16_t*>(p.res);
  const bool mbits = p.mask_bits != 0;                 // the mask is a 1-bit image: one BYTE per 8-column chunk
  const bf16_t* maskT = (mbits || f32ep) ? nullptr : reinterpret_cast<const bf16_t*>(p.mask);
  const float* resF = f32ep ? reinterpret_cast<const float*>(p.res) : nullptr;
  const float* maskF = (f32ep && !mbits) ? reinterpret_cast<const float*>(p.mask) : nullptr;
  const uint8_t* maskB = mbits ? reinterpret_cast<const uint8_t*>(p.mask) : nullptr;
  // LATE (256-row tiles): 128 accumulators + fragments leave no room for the prefetched chunks of a 256-row epilogue (the first attempt
  // spilled 293 VGPRs): the epilogue operands are read where they are used, as the f32 epilogue does
  constexpr bool LATE = BM > 128;
  bf16x8 res_pf[LATE ? 1 : NCH], mask_pf[LATE ? 1 : NCH];
  uint32_t mbit_pf[LATE ? 1 : NCH];
  // omap (conv problems only): GEMM row (n, ho, wo) of the problem's Ho x Wo output grid lands on pixel (ho * o_sh + o_h0, wo * o_sw + o_w0)
  // of an o_Hi x o_Wi image - C, res, mask and bits_out are all indexed by that pixel.  One parity class of a stride-2 input
  // gradient writes every second row / column of dx this way (ops.conv_dgrad).
  auto out_row = [&](const int row) -> long {
    if (!p.omap) return row;
    const int hw = p.Ho * p.Wo;
    const int n = row / hw, rem = row - n * hw;
    const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
    return ((long)n * p.o_Hi + ho * p.o_sh + p.o_h0) * p.o_Wi + wo * p.o_sw + p.o_w0;
  };
  if constexpr (!LATE) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int u = t + c * NT;
    const int row = m0 + u / CPR, col = n0 + (u % CPR) * 8;
    if (row < p.M && col < p.N && (!PARTIAL || u < BM * CPR)) {
      const long orow = out_row(row);
      if (resT) {
        long rrow = orow;
        if (p.res_mod > 0) rrow = orow % p.res_mod;          // uniform branch: the common case pays no integer division
        res_pf[c] = *reinterpret_cast<const bf16x8*>(resT + rrow * p.ldr + col);
      }
      if (maskT) mask_pf[c] = *reinterpret_cast<const bf16x8*>(maskT + orow * p.ldm + col);
      if (maskB) mbit_pf[c] = maskB[orow * p.ldm + (col >> 3)];
    }
  }
  }

  // the per-column operands (folded FrozenBN scale / bias): read in the epilogue they cost it one exposed L2 round trip (≈ 450 of its ≈ 4300
  // clocks on the 64x128 tile, tools/dev/r06_phase_ts.py).  The 8-wave kernels have the 16 registers to hold them across the K loop; on the 4-wave
  // 64x64 kernel they would cost a wave of occupancy per SIMD (108 -> 148 VGPRs), so it keeps reading them late
  static_assert(NT % CPR == 0, "column of a thread's chunks must not depend on the chunk");
  constexpr bool EARLY_AFFINE = NW == 8 && !LATE;
  const int ccol = n0 + (t % CPR) * 8;
  const bool col_ok = ccol < p.N;
  float sc8[8], bi8[8];
  auto load_affine = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc8[e] = 1.f; bi8[e] = 0.f; }
    if (col_ok && p.scale) {
      const float4 s0 = *reinterpret_cast<const float4*>(p.scale + ccol), s1 = *reinterpret_cast<const float4*>(p.scale + ccol + 4);
      sc8[0] = s0.x; sc8[1] = s0.y; sc8[2] = s0.z; sc8[3] = s0.w; sc8[4] = s1.x; sc8[5] = s1.y; sc8[6] = s1.z; sc8[7] = s1.w;
    }
    if (col_ok && p.bias) {
      const float4 s0 = *reinterpret_cast<const float4*>(p.bias + ccol), s1 = *reinterpret_cast<const float4*>(p.bias + ccol + 4);
      bi8[0] = s0.x; bi8[1] = s0.y; bi8[2] = s0.z; bi8[3] = s0.w; bi8[4] = s1.x; bi8[5] = s1.y; bi8[6] = s1.z; bi8[7] = s1.w;
    }
  };
  if constexpr (EARLY_AFFINE) load_affine();

  // ---- per-lane DMA rows
  const int lrow = lane >> 3, pc = lane & 7;
  const int taps = CONV ? p.KH * p.KW : 1;
  const int sg = (CONV && p.transposed) ? -1 : 1;         // dgrad: the tap shift is subtracted
  const long tsub = (CONV && p.transposed) ? p.lda / p.sh : p.lda;     // elements per (sub-)pixel step, see below
  unsigned a_off[GA], a_mask[GA];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int trow = (i * NW + wave) * 8 + lrow;
    const int row = m0 + trow;
    const int swz = (pc ^ ((trow >> 1) & 7)) * 8;
    unsigned mask = 0;
    long off;
    if (CONV) {
      const int HoWo = p.Ho * p.Wo;
      const int n = row / HoWo, rem = row - n * HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int hb = p.transposed ? ho + p.ph : ho * p.sh - p.ph;
      const int wb = p.transposed ? wo + p.pw : wo * p.sw - p.pw;
      // forward: pixel (hb + kh*dh, wb + kw*dw).  dgrad (transposed): pixel ((hb - kh*dh)/s, (wb - kw*dw)/s) where both
      // divide; its byte offset is [(hb - kh*dh)*Wi + (wb - kw*dw)] * (pitch/s) - the division cancels against the pixel
      // pitch - so "lane base + wave-uniform tap delta" holds for strided dgrads too (tsub = pitch/s in elements).
      off = ((long)n * p.Hi * p.Wi) * p.lda + ((long)hb * p.Wi + wb) * tsub + swz;
      if (row < p.M)
        for (int tp = 0, kh = 0, kw = 0; tp < taps; ++tp, ++kw) {      // (kh, kw) walk without a division per tap
          if (kw == p.KW) { kw = 0; ++kh; }
          int hi = hb + sg * kh * p.dh, wi = wb + sg * kw * p.dw;
          bool ok = hi >= 0 && wi >= 0;
          if (p.transposed) {
            ok = ok && (hi % p.sh) == 0 && (wi % p.sw) == 0;
            hi /= p.sh;
            wi /= p.sw;
          }
          if (ok && hi < p.Hi && wi < p.Wi) mask |= 1u << tp;
        }
    } else {
      off = (long)row * p.lda + swz;
      mask = row < p.M ? 1u : 0u;
    }
    a_off[i] = (unsigned)(off * 2);
    if (!CONV && !mask) a_off[i] = OOB;        // plain rows beyond M: the offset itself is out of range, no per-tile test
    a_mask[i] = mask;
  }
  unsigned b_off[GB ? GB : 1];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int trow = (i * NW + wave) * 8 + lrow;
    const int row = n0 + trow;
    const int swz = (pc ^ ((trow >> 1) & 7)) * 8;
    b_off[i] = row < p.N ? (unsigned)(((long)row * p.ldb + swz) * 2) : OOB;
  }
  // BR: byte offset of this lane's 16 bytes inside the wave's first fragment block of a K tile: block (n / 32, k / 16) of the image is the KB at
  // ((n / 32) * (ldb / 16) + k / 16) * 1024; the wave group's two k16 steps of a tile are neighbours
  __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(BR ? p.bfrag : p.B), 0,
                                                                 BR ? (unsigned)((long)p.N * p.ldb * 2) : b_bytes, 0x00020000);
  const unsigned f_off = (BR && n0 + wn < p.N) ? ((unsigned)((n0 + wn) >> 5) * (unsigned)(p.ldb >> 4) + 2u * (unsigned)kgrp) * 1024u + (unsigned)lane * 16u
                                               : OOB;
  u32x4 fbr[BR ? S : 1][2];

  // ---- wave-uniform K position: channel offset inside the tap, tap index, its pixel shift in bytes
  int c0 = 0, tap = 0, kh = 0, kw = 0, k0 = kb0 * BK2;
  const int rowbytes = (int)(tsub * 2);
  if (CONV && KH == 2) {                                // the second team starts in the middle of the (tap, channel) walk
    tap = k0 / p.Ci;
    c0 = k0 - tap * p.Ci;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }
  // CONV: per A row, the offset of the CURRENT tap with its validity folded in (out of range when the tap misses the image);
  // it changes only when the tap does, so a K tile costs one add per row
  unsigned a_cur[GA];
  {
    const unsigned tapdelta0 = CONV ? (unsigned)(sg * (kh * p.dh * p.Wi + kw * p.dw) * rowbytes) : 0u;
#pragma unroll
    for (int i = 0; i < GA; ++i) a_cur[i] = (!CONV || (a_mask[i] & (1u << tap))) ? a_off[i] + tapdelta0 : OOB;
  }
  // btap (conv problems only): the B rows hold MORE taps than this problem walks - tap t of the walk reads the channel run that starts
  // at element btap[t] of a B row (a parity class of a stride-2 input gradient uses 1, 2 or 4 of the nine tap blocks of the packed
  // dgrad weight, a column half of layer4's dilated 3x3 six,
  // dgrad weight, in place).  cur_bt = that start for the current tap (wave-uniform)
  int cur_bt = (CONV && p.btap_on) ? p.btap[tap] : 0;
  auto advance = [&]() {
    k0 += BK2;
    if (!CONV) return;
    c0 += BK2;
    if (c0 >= p.Ci) {
      // (the empty volatile asm keeps this rare block a real branch: if-converted, its instructions - two scalar
      // multiplies among them - would run on every K tile)
      asm volatile("" ::: "memory");
      c0 = 0;
      ++tap;
      if (++kw == p.KW) { kw = 0; ++kh; }
      const unsigned tapdelta = (unsigned)(sg * (kh * p.dh * p.Wi + kw * p.dw) * rowbytes);
      const unsigned tapbit = 1u << tap;
#pragma unroll
      for (int i = 0; i < GA; ++i) a_cur[i] = (a_mask[i] & tapbit) ? a_off[i] + tapdelta : OOB;
      if (p.btap_on) cur_bt = p.btap[tap & 7];
    }
  };

  // one LDS-DMA piece of the tile at the current K position: pieces 0..GA-1 are A rows, GA..GA+GB-1 are B rows
  auto issue_piece = [&](const int stage, const int i) {      // literal stage / piece only: folds to immediates
    unsigned char* st = ring + stage * STAGE_BYTES;
    const unsigned kb2 = (CONV && p.btap_on) ? (unsigned)((cur_bt + c0) * 2) : (unsigned)(k0 * 2);
    if (i < GA) {
      unsigned koff = CONV ? (unsigned)(c0 * 2) : (unsigned)(k0 * 2);    // (an out-of-range a_cur stays out of range: c0 * 2 < 2^31)
      // awrap (bf16x3): the A rows hold [hi | lo] (2 awrap channels) and the walk over 3 awrap reads hi again for its last third
      if (p.awrap && (CONV ? c0 : k0) >= 2 * p.awrap) koff -= 4u * (unsigned)p.awrap;
      unsigned voff = a_cur[i] + koff;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + ((i * NW + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
    } else {
      // (a local, not the expression, as the builtin argument: clang's host pass otherwise drops the kernel stub)
      unsigned bv = b_off[i - GA] + kb2;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + (BM + ((i - GA) * NW + wave) * 8) * ROWB), 16, bv, 0, 0, 0);
    }
  };
  auto issue = [&](const int stage) {
#pragma unroll
    for (int i = 0; i < GA + GB; ++i) issue_piece(stage, i);
    if constexpr (BR) {            // the tile's two B fragments of this wave: wave-uniform K position as the scalar offset (64 bytes per k)
      const int kpos = (CONV && p.btap_on) ? cur_bt + c0 : k0;
      fbr[stage][0] = __builtin_amdgcn_raw_buffer_load_b128(rsF, f_off, kpos * 64, 0);
      fbr[stage][1] = __builtin_amdgcn_raw_buffer_load_b128(rsF, f_off + 1024u, kpos * 64, 0);
    }
    advance();
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets inside a stage (constants of the thread)
  const int frow = lane & 31, fhalf = lane >> 5;
  constexpr int NKQ = NW == 8 ? 2 : 4;                  // k16 steps of a K tile handled by this wave
  int a_rd[NKQ][MI], b_rd[NKQ][NI];
#pragma unroll
  for (int kq = 0; kq < NKQ; ++kq) {
    const int ks = NW == 8 ? 2 * kgrp + kq : kq;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int row = wm + i * 32 + frow;
      a_rd[kq][i] = row * ROWB + (((ks * 2 + fhalf) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int row = wn + j * 32 + frow;
      b_rd[kq][j] = (BM + row) * ROWB + (((ks * 2 + fhalf) ^ ((row >> 1) & 7)) * 16);
    }
  }
  // A stage is consumed in two steps so that the LDS-DMA issue of the NEXT tile sits between them: fragment reads first
  // (all k16 steps of the stage), then - after the DMA bookkeeping has overlapped the LDS latency - the MFMAs back to back.
  // The waves of a workgroup run in lockstep between barriers, so nothing else hides that latency.
  bf16x8 fa[NKQ][MI], fb[NKQ][NI];
  auto load_frags = [&](const int stage) {
    const unsigned char* st = ring + stage * STAGE_BYTES;
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[kq][i] = *reinterpret_cast<const bf16x8*>(st + a_rd[kq][i]);
      if constexpr (!BR) {
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[kq][j] = *reinterpret_cast<const bf16x8*>(st + b_rd[kq][j]);
      }
    }
  };
  auto mfma_all = [&](const int stage) {
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (BR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kq][i], __builtin_bit_cast(bf16x8, fbr[stage][kq]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kq][i], fb[kq][j], acc[i][j], 0, 0, 0);
        }
  };
  auto compute = [&](const int stage) {
    load_frags(stage);
    mfma_all(stage);
  };
  // ---- S-stage pipeline, unrolled by the ring depth (stage indices are literals after unrolling).  A wave waits only for
  //      its own oldest tile: (S-2) later tiles x G DMA instructions may stay in flight.
  SEDT_TS(1, __builtin_amdgcn_s_memtime());
  constexpr int G = GA + GB + (BR ? 2 : 0);            // vector-memory operations of one tile and wave: LDS-DMA pieces + register loads, in issue order
  if constexpr (PP) {
    // Ping-pong: every stage is two barrier-delimited segments - "load" (fragment reads of tile s, LDS-DMA issue of tile
    // s+S-1) and "math" (the MFMAs of tile s) - and the second wave group runs ONE BARRIER behind the first, so on every
    // SIMD one wave is in its math segment while the other is in its load segment.  Same instruction stream for both
    // groups (the skew is one extra barrier before the loop for group 1, after it for group 0).  Ordering rules:
    //   RAW  tile s+1 is waited for (own DMA pieces, counted vmcnt) at the END of the load segment of tile s: that is
    //        before the barrier preceding the first group's reads of tile s+1 for both groups;
    // (Issuing the LDS-DMA pieces between the MFMAs of the math segment instead was measured 3-15 % slower.)
    //   WAR  the buffer of tile s-1 is restaged in the load segment of tile s; the last reads of it (group 1, load segment
    //        of s-1) retired before the barrier ending that segment (lgkmcnt(0)), which precedes the first group's issue.
    static_assert(NW == 8 && S >= 2, "ping-pong needs two wave groups and a ring");
#pragma unroll
    for (int s0 = 0; s0 < S - 1; ++s0)
      if (s0 < nkb) issue(s0);
    if (nkb >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (kgrp == 1) lds_barrier();
    int it = 0;
    for (; it + 2 * S - 1 <= nkb; it += S) {
#pragma unroll
      for (int ph = 0; ph < S; ++ph) {
        lds_barrier();
        load_frags(ph);
        issue((ph + S - 1) % S);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");      // tile s+1 complete (own pieces)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfma_all(ph);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    for (; it < nkb; it += S) {
#pragma unroll
      for (int ph = 0; ph < S; ++ph) {
        if (it + ph < nkb) {
          const bool more = it + ph + S - 1 < nkb;
          lds_barrier();
          load_frags(ph);
          if (more) issue((ph + S - 1) % S);
          if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          lds_barrier();
          __builtin_amdgcn_s_setprio(1);
          mfma_all(ph);
          __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    if (kgrp == 0) lds_barrier();
  } else {
  if constexpr (S == 1) {
    // single K tile (K = 64: the 64 -> 256 convolutions of layer1): no ring, half the LDS, twice the resident workgroups -
    // these problems are pure HBM streams and only occupancy hides their latency
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    compute(0);
  }
#pragma unroll
  for (int s0 = 0; s0 < S - 1; ++s0)
    if (s0 < nkb) issue(s0);
  int it = 0;
  // steady state: whole rounds in which every phase still has a tile to prefetch - no per-phase tests
  for (; S > 1 && it + 2 * S - 1 <= nkb; it += S) {
#pragma unroll
    for (int ph = 0; ph < S; ++ph) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S >= 2 ? S - 2 : 0) * G) : "memory");
      lds_barrier();
      load_frags(ph);
      issue((ph + S - 1) % S);
      mfma_all(ph);
    }
  }
  for (; S > 1 && it + S <= nkb; it += S) {
#pragma unroll
    for (int ph = 0; ph < S; ++ph) {
      const bool more = it + ph + S - 1 < nkb;
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S >= 2 ? S - 2 : 0) * G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      load_frags(ph);
      if (more) issue((ph + S - 1) % S);
      mfma_all(ph);
    }
  }
#pragma unroll
  for (int ph = 0; ph < S - 1; ++ph)
    if (it + ph < nkb) {                     // tail: already issued, nothing left to prefetch
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      compute(ph);
    }
  }
  lds_barrier();
  SEDT_TS(2, __builtin_amdgcn_s_memtime());

  // ---- epilogue through LDS (identical to igemm2)
  constexpr int CP = BN + 4;
  // 8 waves: the two groups hold the two halves of the K sum.  When the ring has room for two staging tiles each group
  // writes its own and the chunk loop adds them (one phase, one barrier); otherwise the second group adds into the first
  // group's tile in a second phase.
  constexpr bool TWO_CS = NW == 8 && (size_t)KH * S * STAGE_BYTES >= (size_t)2 * KH * BM * CP * sizeof(float);
  static_assert(KH == 1 || TWO_CS, "two K halves: the rings must hold the four staging tiles");
  float* Cs = reinterpret_cast<float*>(smem);
  float* Cs2 = Cs + BM * CP;
  if (NW == 4 || kgrp == 0 || TWO_CS) {
    float* dst = Cs + (TWO_CS ? (khalf * 2 + kgrp) * (BM * CP) : 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          dst[row * CP + wn + j * 32 + frow] = acc[i][j][r];
        }
  }
  __syncthreads();
  if (NW == 8 && !TWO_CS) {                     // second wave group: add its half of the K sum
    if (kgrp == 1) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
            Cs[row * CP + wn + j * 32 + frow] += acc[i][j][r];
          }
    }
    __syncthreads();
  }

  const uint32_t seed = eff_seed(p.seed, p.seed_ptr);
  const uint32_t thresh = drop_threshold(p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  bf16_t* outT = reinterpret_cast<bf16_t*>(p.C);
  // a thread's chunks all sit in the same 8 columns (NT is a multiple of the chunks per row): per-column operands once.
  // The K <= 256 problems are instruction-issue bound in this epilogue (PMC: 17 VALU per MFMA), so it is kept lean.
  if constexpr (!EARLY_AFFINE) load_affine();
  const bool affine = p.scale != nullptr || p.bias != nullptr;
  const bool relu_pre = !p.act_post_res && p.act == SEDT_ACT_RELU, relu_post = p.act_post_res && p.act == SEDT_ACT_RELU;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int u = t + c * NT;
    const int trow = u / CPR, cc = (u % CPR) * 8;
    const int row = m0 + trow, col = n0 + cc;
    if (row >= p.M || !col_ok || (PARTIAL && u >= BM * CPR)) continue;
    const long orow = out_row(row);
    float v[8];
    {
      const float4 x0 = *reinterpret_cast<const float4*>(Cs + trow * CP + cc);
      const float4 x1 = *reinterpret_cast<const float4*>(Cs + trow * CP + cc + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      if (TWO_CS) {
#pragma unroll
        for (int gq = 1; gq < 2 * KH; ++gq) {           // the other wave groups' shares of the K sum, in a fixed order
          const float4 y0 = *reinterpret_cast<const float4*>(Cs2 + (gq - 1) * (BM * CP) + trow * CP + cc);
          const float4 y1 = *reinterpret_cast<const float4*>(Cs2 + (gq - 1) * (BM * CP) + trow * CP + cc + 4);
          v[0] += y0.x; v[1] += y0.y; v[2] += y0.z; v[3] += y0.w; v[4] += y1.x; v[5] += y1.y; v[6] += y1.z; v[7] += y1.w;
        }
      }
    }
    if (affine) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc8[e], bi8[e]);
    }
    if (relu_pre) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.drop_p > 0.f) {
      // drop_keep of 8 consecutive elements: the (seed, high word) part of the hash once, one outer hash per pair
      const uint64_t h0 = ((uint64_t)orow * (uint64_t)p.N + col) >> 1;
      const uint32_t lo = (uint32_t)h0;
      const uint32_t inner = mix32(seed ^ ((uint32_t)(h0 >> 32) * 0x9E3779B9U) ^ 0x85ebca6bU);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t h = mix32((lo + q) ^ inner);
        v[2 * q] = (h & 0xffffu) >= thresh ? v[2 * q] * inv_keep : 0.f;
        v[2 * q + 1] = (h >> 16) >= thresh ? v[2 * q + 1] * inv_keep : 0.f;
      }
    }
    if (resT) {
      bf16x8 rv;
      if constexpr (LATE) {
        long rrow = orow;
        if (p.res_mod > 0) rrow = orow % p.res_mod;
        rv = *reinterpret_cast<const bf16x8*>(resT + rrow * p.ldr + col);
      } else {
        rv = res_pf[c];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
    }
    if (resF) {
      long rrow = orow;
      if (p.res_mod > 0) rrow = orow % p.res_mod;
      const float4 r0 = *reinterpret_cast<const float4*>(resF + rrow * p.ldr + col);
      const float4 r1 = *reinterpret_cast<const float4*>(resF + rrow * p.ldr + col + 4);
      v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
    }
    if (relu_post) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (maskT) {
      bf16x8 mv;
      if constexpr (LATE) mv = *reinterpret_cast<const bf16x8*>(maskT + orow * p.ldm + col);
      else mv = mask_pf[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)mv[e] > 0.f) ? v[e] : 0.f;
    }
    if (maskF) {
      const float4 q0 = *reinterpret_cast<const float4*>(maskF + orow * p.ldm + col);
      const float4 q1 = *reinterpret_cast<const float4*>(maskF + orow * p.ldm + col + 4);
      const float mv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
    }
    if (maskB) {
      uint32_t mb;
      if constexpr (LATE) mb = maskB[orow * p.ldm + (col >> 3)];
      else mb = mbit_pf[c];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
    }
    if (p.alpha != 1.f) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    }
    if (f32ep) {
      float* outF = reinterpret_cast<float*>(p.C) + orow * p.ldc + col;
      *reinterpret_cast<float4*>(outF) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(outF + 4) = make_float4(v[4], v[5], v[6], v[7]);
      if (p.bits_out) {
        uint32_t ob = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob |= (v[e] > 0.f ? 1u : 0u) << e;
        p.bits_out[orow * p.ldbits + (col >> 3)] = (uint8_t)ob;
      }
      if (p.split_out) {        // the [hi | lo] operand image of what was stored (sedt_split3, pattern 0) for the GEMMs that consume it
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          hi[e] = (bf16_t)v[e];
          lo[e] = (bf16_t)(v[e] - (float)hi[e]);
        }
        bf16_t* sp = reinterpret_cast<bf16_t*>(p.split_out) + orow * (2L * p.N) + col;
        *reinterpret_cast<bf16x8*>(sp) = hi;
        *reinterpret_cast<bf16x8*>(sp + p.N) = lo;
      }
      continue;
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(outT + orow * p.ldc + col) = o;
    if (p.bits_out) {                               // sign bits of what was stored: the backward's ReLU mask at 1/16 of the bytes
      uint32_t ob = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) ob |= ((float)o[e] > 0.f ? 1u : 0u) << e;
      p.bits_out[orow * p.ldbits + (col >> 3)] = (uint8_t)ob;
    }
  }
  SEDT_TS(3, __builtin_amdgcn_s_memtime());
}

template <int BM, int BN, int S, int NW = 4, int PP = 0, int KH = 1, int BR = 0>
__device__ __forceinline__ void igemm3_body(const SedtIgemm& p, const unsigned a_bytes, const unsigned b_bytes, const int bx) {
  if (p.conv) igemm3_impl<BM, BN, S, NW, true, PP, KH, BR>(p, a_bytes, b_bytes, bx);      // uniform branch: two specialised programs
  else igemm3_impl<BM, BN, S, NW, false, PP, KH, BR>(p, a_bytes, b_bytes, bx);
}

template <int BM, int BN, int S>
__global__ __launch_bounds__(256) void igemm3_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes) {
  igemm3_body<BM, BN, S>(p, a_bytes, b_bytes, blockIdx.x);
}

template <int BM, int BN, int S, int PP = 0>
__global__ __launch_bounds__(512) void igemm3_w8_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes) {
  igemm3_body<BM, BN, S, 8, PP>(p, a_bytes, b_bytes, blockIdx.x);
}

#ifdef SEDT_DEV            // (the register-streamed weight operand is a measured developer variant, profiles/r06_ab_breg.txt: not compiled into the product)
// 8 waves, ping-pong, B through registers out of the fragment-major image (BR = 1)
template <int BM, int BN, int S>
__global__ __launch_bounds__(512) void igemm3_br_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes) {
  igemm3_body<BM, BN, S, 8, 1, 1, 1>(p, a_bytes, b_bytes, blockIdx.x);
}
#endif

// 16 waves: two 8-wave ping-pong teams, each over one half of K (KH = 2)
template <int BM, int BN, int S>
__global__ __launch_bounds__(1024) void igemm3_w16_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes) {
  igemm3_body<BM, BN, S, 8, 1, 2>(p, a_bytes, b_bytes, blockIdx.x);
}

template <int BM, int BN, int S>
static int launch3_w16(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  SEDT_DESCRIBE("igemm3_w16_kernel<%d, %d, %d>", BM, BN, S);
  constexpr size_t ring = (size_t)2 * S * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)4 * BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_w16_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 w16: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(1024), lds, st, p, a_bytes, b_bytes);
  return check_launch("igemm3_w16");
}

template <int BM, int BN, int S, int PP = 0>
static int launch3_w8(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  SEDT_DESCRIBE("igemm3_w8_kernel<%d, %d, %d, %d>", BM, BN, S, PP);
  constexpr size_t ring = (size_t)S * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_w8_kernel<BM, BN, S, PP>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 w8: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, p, a_bytes, b_bytes);
  return check_launch("igemm3_w8");
}

#ifdef SEDT_DEV
template <int BM, int BN, int S>
static int launch3_br(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  SEDT_DESCRIBE("igemm3_br_kernel<%d, %d, %d>", BM, BN, S);
  constexpr size_t ring = (size_t)S * BM * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_br_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 br: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), lds, st, p, a_bytes, b_bytes);
  return check_launch("igemm3_br");
}
#endif

// Co-scheduled launch: the first nwg_main workgroups run the forward / dgrad GEMM p, the rest run pending weight-gradient
// problems.  At batch 64 most GEMMs of the backward chain occupy 2 of the ~5 workgroup slots of a CU; weight gradients
// are needed only by the optimizer, so their tiles ride in the spare slots of the chain's launches instead of taking
// launches (and tails) of their own.  Measured with two streams: dgrad + wgrad of one layer4 conv 162 us -> 139 us.
template <int BM, int BN, int S>
__global__ __launch_bounds__(256) void igemm3_co_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes,
                                                        const int nwg_main, const WgradGroup g) {
  if ((int)blockIdx.x < nwg_main) igemm3_body<BM, BN, S>(p, a_bytes, b_bytes, blockIdx.x);
  else wgrad_group_run(g, (int)blockIdx.x - nwg_main);
}

// set by sedt_igemm_co (igemm.hip) around its sedt_igemm call: the weight-gradient group the next igemm3 launch takes along
thread_local const WgradGroup* co_group = nullptr;
thread_local bool co_taken = false;

// dry run: igemm3_try resolves the kernel configuration of a problem without launching (sedt_igemm_group uses it)
struct Igemm3Plan {
  bool on, ok;
  int bm, bn, s;
  unsigned a_bytes, b_bytes;
};
thread_local Igemm3Plan plan3 = {false, false, 0, 0, 0, 0u, 0u};

// several independent forward / dgrad problems of the 64x64 configuration (all with the same ring depth) in ONE launch (the q / k / v projections
// of an attention block and their three dgrads are independent GEMMs of a few microseconds each)
constexpr int IG_MAXG = 8;
struct IgemmGroup {
  int n;
  int blk0[IG_MAXG + 1];
  unsigned a_bytes[IG_MAXG], b_bytes[IG_MAXG];
  SedtIgemm p[IG_MAXG];
};

template <int S>
__global__ __launch_bounds__(256) void igemm3_group_kernel(const IgemmGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk0[i + 1]) ++i;
  igemm3_body<64, 64, S>(g.p[i], g.a_bytes[i], g.b_bytes[i], (int)blockIdx.x - g.blk0[i]);
}

// two or more problems of ONE ping-pong configuration in one launch (the column halves of layer4's dilated 3x3 convolutions: 128 tiles of
// 128x128 each - launched alone, either half would leave half the chip idle)
template <int BM, int BN, int S>
__global__ __launch_bounds__(512) void igemm3_w8_group_kernel(const IgemmGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk0[i + 1]) ++i;
  igemm3_body<BM, BN, S, 8, 1>(g.p[i], g.a_bytes[i], g.b_bytes[i], (int)blockIdx.x - g.blk0[i]);
}

#ifdef SEDT_DEV
template <int BM, int BN, int S>
__global__ __launch_bounds__(512) void igemm3_br_group_kernel(const IgemmGroup g) {
  int i = 0;
  while (i + 1 < g.n && (int)blockIdx.x >= g.blk0[i + 1]) ++i;
  igemm3_body<BM, BN, S, 8, 1, 1, 1>(g.p[i], g.a_bytes[i], g.b_bytes[i], (int)blockIdx.x - g.blk0[i]);
}

template <int BM, int BN, int S>
static int launch3_br_group(const IgemmGroup& g, int nblk, hipStream_t st) {
  constexpr size_t ring = (size_t)S * BM * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_br_group_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 br group: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, st, g);
  return check_launch("igemm3_br_group");
}
#endif

template <int BM, int BN, int S>
static int launch3_w8_group(const IgemmGroup& g, int nblk, hipStream_t st) {
  constexpr size_t ring = (size_t)S * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_w8_group_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 w8 group: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, st, g);
  return check_launch("igemm3_w8_group");
}

template <int BM, int BN, int S>
static int launch3_co(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st, const WgradGroup& g) {
  constexpr size_t ring = (size_t)S * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t wg = (size_t)2 * (64 * ROWB + 64 * 64 * 2);
  constexpr size_t lds0 = ring > ctile ? ring : ctile;
  constexpr size_t lds = lds0 > wg ? lds0 : wg;
  static bool attr_set = false;
  auto kern = igemm3_co_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3 co: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg + g.blk0[g.n]), dim3(256), lds, st, p, a_bytes, b_bytes, nwg, g);
  co_taken = true;
  return check_launch("igemm3_co");
}

template <int BM, int BN, int S>
static int launch3(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  if (!plan3.on) SEDT_DESCRIBE("igemm3_kernel<%d, %d, %d>", BM, BN, S);
  if (plan3.on) {
    plan3.ok = true;
    plan3.bm = BM; plan3.bn = BN; plan3.s = S;
    plan3.a_bytes = a_bytes; plan3.b_bytes = b_bytes;
    return 0;
  }
  // riders inherit the launch's LDS allocation: only the 32 KB configuration keeps their occupancy (5 workgroups per CU)
  if (co_group != nullptr && !co_taken && BM == 64 && BN == 64 && S == 2) return launch3_co<BM, BN, S>(p, a_bytes, b_bytes, st, *co_group);
  constexpr size_t ring = (size_t)S * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm3_kernel<BM, BN, S>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm3: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, p, a_bytes, b_bytes);
  return check_launch("igemm3");
}

int igemm_lds_try(const SedtIgemm& p, hipStream_t st);   // below: envelope checks + tile choice, then igemm3_try
bool igemm3_planning() { return plan3.on; }

// 0 = launched as one grouped kernel, -1 = some problem does not resolve to the 64x64 2-stage kernel (nothing launched)
int igemm3_group_try(const SedtIgemm* jobs, int njobs, hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM_GROUP");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  if (!on || njobs < 2 || njobs > IG_MAXG) return -1;
  IgemmGroup g;
  g.n = njobs;
  int blk = 0, S = 0, big = 0;
  for (int i = 0; i < njobs; ++i) {
    const SedtIgemm& p = jobs[i];
    if (p.trans) return -1;
    plan3 = {true, false, 0, 0, 0, 0u, 0u};
    const int r = igemm_lds_try(p, st);
    const Igemm3Plan got = plan3;
    plan3.on = false;
    if (r != 0 || !got.ok) return -1;
    const bool b128 = got.bm == 128 && got.bn == 128 && got.s == 3;           // (requested by the caller's tile hint)
    if (!b128 && (got.bm != 64 || got.bn != 64 || (got.s != 2 && got.s != 3))) return -1;
    if (i == 0) big = b128;
    if ((int)b128 != big) return -1;
    if (i == 0) S = got.s;
    if (got.s < S) S = got.s;               // mixed depths: the shallower ring serves every K
    g.p[i] = p;
    g.a_bytes[i] = got.a_bytes;
    g.b_bytes[i] = got.b_bytes;
    g.blk0[i] = blk;
    blk += big ? ((p.N + 127) / 128) * ((p.M + 127) / 128) : ((p.N + 63) / 64) * ((p.M + 63) / 64);
  }
  g.blk0[njobs] = blk;
#ifdef SEDT_DEV
  if (big) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM_BREG");
    bool br = e && atoi(e) != 0;
    for (int i = 0; i < njobs && br; ++i) {
      const SedtIgemm& p = jobs[i];
      br = p.bfrag != nullptr && !p.f32ep && !p.awrap && (p.N % 32) == 0 && (p.ldb % 64) == 0 && (reinterpret_cast<uintptr_t>(p.bfrag) & 15) == 0 &&
           (long)p.N * p.ldb * 2 < (1L << 31);
    }
    if (br) {
      SEDT_DESCRIBE("igemm3_br_group_kernel<128, 128, 3>");
      return launch3_br_group<128, 128, 3>(g, blk, st);
    }
  }
#endif
  if (big) SEDT_DESCRIBE("igemm3_w8_group_kernel<128, 128, 3>");
  else SEDT_DESCRIBE("igemm3_group_kernel<%d>", S == 3 ? 3 : 2);
  if (big) return launch3_w8_group<128, 128, 3>(g, blk, st);
  if (S == 3) {
    constexpr size_t lds = (size_t)3 * (64 + 64) * ROWB;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm3_group_kernel<3>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("igemm3 group: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return 1;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(igemm3_group_kernel<3>, dim3(blk), dim3(256), lds, st, g);
  } else {
    constexpr size_t lds = (size_t)2 * (64 + 64) * ROWB;      // ring 32 KB >= the 17 KB epilogue tile
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm3_group_kernel<2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        set_error("igemm3 group: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return 1;
      }
      attr_set = true;
    }
    hipLaunchKernelGGL(igemm3_group_kernel<2>, dim3(blk), dim3(256), lds, st, g);
  }
  return check_launch("igemm3_group");
}

// -1 = outside the envelope
int igemm3_try(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, int bm, int bn, hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM_V3");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  if (!on) return -1;
  if (p.K % BK2) return -1;
  if (p.conv && (p.KH * p.KW > 32 || (p.transposed && (p.sh != p.sw || (p.lda % p.sh) != 0)))) return -1;
  if ((reinterpret_cast<uintptr_t>(p.scale) & 15) || (reinterpret_cast<uintptr_t>(p.bias) & 15)) return -1;
  static int stages = -1;
  if (stages < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_STAGES");
    stages = e ? atoi(e) : 0;
  }
  int S = stages;
  if (S == 0) {
    // re-tuned after the 8-wave / lean-bookkeeping changes (full step, SEDT_IGEMM3_STAGES sweep: 2 everywhere 8881,
    // old rule "3 when K >= 1024 and <= 3 workgroups per CU" 9447, 3 everywhere 9590, 4 9457 clips/s): two tiles in flight
    // pay at every depth that has them
    // (per-shape sweep of the step's launches, bench.py --dump-igemm: K = 256 problems and the N = 64, K = 576 layer1 3x3
    // convolution - few K tiles, fixed per-workgroup costs dominate - prefer the occupancy of the 2-stage ring)
    S = (p.K >= 5 * BK2 && !(p.N <= 64 && p.K <= 9 * BK2)) ? 3 : 2;
  }
  static int nw_env = -1;
  if (nw_env < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_NW");
    nw_env = e ? atoi(e) : 0;
  }
  // measured (tools/tune_igemm.py, SEDT_IGEMM3_NW): the 64x128 tile runs 2-14 % faster with 8 waves (two groups splitting the
  // k16 steps) at every shape that selects it; 128x128 / 128x64 with 8 waves are experiment-only (SEDT_IGEMM3_NW=8)
  static int pp_env = -1;
  if (pp_env < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_PP");
    pp_env = e ? atoi(e) : 1;
  }
  static int w4k = -1;          // experiment: below this K the 64x128 tile runs on the 4-wave kernel (32 MFMAs per wave and tile)
  if (w4k < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_W4_BELOW_K");
    w4k = e ? atoi(e) : 0;
  }
  const bool force4 = p.K < w4k;
  // exactly one 64x128 tile per CU and a long K: two 8-wave teams per workgroup, half of K each (SEDT_IGEMM3_W16=1)
  static int w16 = -1, w16_mink = 0, w16_tiles = 320;
  if (w16 < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_W16");        // same-box A/B on the C2 step: 5.688 -> 5.657 ms (29 launches, ~ -10 % each)
    w16 = (e && e[0] == '0') ? 0 : 1;
    e = sedt::dev_getenv("SEDT_IGEMM3_W16_TILES");
    w16_tiles = e ? atoi(e) : 320;
    e = sedt::dev_getenv("SEDT_IGEMM3_W16_MINK");
    w16_mink = e ? atoi(e) : 512;            // (512 vs 1024: C2 5.61 vs 5.65 ms, C3 4.56 vs 4.58 - within the run-to-run spread)
  }
  if (w16 && !plan3.on && co_group == nullptr && bm == 64 && bn == 128 && S >= 3 && p.K >= w16_mink && (p.K / BK2) % 2 == 0) {
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 127) / 128);
    if (tiles <= w16_tiles) return launch3_w16<64, 128, 3>(p, a_bytes, b_bytes, st);
  }
  // the B = 32 configurations (M = 3968 rows) run on 64x64 tiles, 248 of them at N = 256 - one 4-wave workgroup per CU: the same
  // two-team form on that tile (same-box A/B: C3 4.655 -> 4.568 ms, C5 8.34 -> 8.27; one 8-wave team instead: no change)
  static int small16 = -1;
  if (small16 < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM3_SMALL16");
    small16 = (e && e[0] == '0') ? 0 : 1;
  }
  if (small16 && !plan3.on && co_group == nullptr && bm == 64 && bn == 64 && S >= 3 && p.K >= w16_mink && (p.K / BK2) % 2 == 0) {
    const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (tiles <= 320) return launch3_w16<64, 64, 3>(p, a_bytes, b_bytes, st);
  }
#ifdef SEDT_DEV
  // B through registers (SedtIgemm.bfrag given; SEDT_IGEMM_BREG=1 in the developer build): the 64x128 / 128x128 ping-pong problems
  static int breg = -1;
  if (breg < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM_BREG");
    breg = e ? atoi(e) : 0;
  }
  if (breg && p.bfrag != nullptr && !plan3.on && co_group == nullptr && bn == 128 && (bm == 64 || bm == 128) && !p.f32ep && !p.awrap &&
      (p.N % 32) == 0 && (p.ldb % 64) == 0 && (reinterpret_cast<uintptr_t>(p.bfrag) & 15) == 0 && (long)p.N * p.ldb * 2 < (1L << 31)) {
    // (the ring holds A only and the epilogue's staging tile sets the LDS size, so a fourth / fifth stage is free: measured, 4 stages -4 % on
    // these launches, 5 no better, profiles/r06_ab_breg.txt - the instances were removed again)
    if (bm == 64) return S >= 3 ? launch3_br<64, 128, 3>(p, a_bytes, b_bytes, st) : launch3_br<64, 128, 2>(p, a_bytes, b_bytes, st);
    return S >= 3 ? launch3_br<128, 128, 3>(p, a_bytes, b_bytes, st) : launch3_br<128, 128, 2>(p, a_bytes, b_bytes, st);
  }
#endif
  if (!plan3.on && co_group == nullptr && nw_env != 4 && pp_env && !force4) {
#define SEDT_PP(BM_, BN_)                                                                                   \
  if (bm == BM_ && bn == BN_) {                                                                             \
    return S >= 3 ? launch3_w8<BM_, BN_, 3, 1>(p, a_bytes, b_bytes, st) : launch3_w8<BM_, BN_, 2, 1>(p, a_bytes, b_bytes, st); \
  }
    SEDT_PP(64, 128)
    SEDT_PP(128, 128)
    SEDT_PP(128, 64)
    if (bm == 256 && bn == 128 && S >= 3) return launch3_w8<256, 128, 3, 1>(p, a_bytes, b_bytes, st);
#undef SEDT_PP
  }
  if (!plan3.on && co_group == nullptr && nw_env != 4 && !force4) {
    if (bm == 64 && bn == 128) return S >= 3 ? launch3_w8<64, 128, 3>(p, a_bytes, b_bytes, st) : launch3_w8<64, 128, 2>(p, a_bytes, b_bytes, st);
    if (nw_env == 8 && bm == 128 && bn == 128)
      return S >= 3 ? launch3_w8<128, 128, 3>(p, a_bytes, b_bytes, st) : launch3_w8<128, 128, 2>(p, a_bytes, b_bytes, st);
    if (nw_env == 8 && bm == 128 && bn == 64)
      return S >= 3 ? launch3_w8<128, 64, 3>(p, a_bytes, b_bytes, st) : launch3_w8<128, 64, 2>(p, a_bytes, b_bytes, st);
  }
  if (bm == 64 && bn == 64 && p.K == BK2 && !plan3.on && co_group == nullptr) return launch3<64, 64, 1>(p, a_bytes, b_bytes, st);
#define SEDT_L3(BM_, BN_)                                                      \
  if (bm == BM_ && bn == BN_) {                                                \
    if (S == 3) return launch3<BM_, BN_, 3>(p, a_bytes, b_bytes, st);          \
    if (S == 4 && BM_ + BN_ <= 192) return launch3<BM_, BN_, 4>(p, a_bytes, b_bytes, st); \
    return launch3<BM_, BN_, 2>(p, a_bytes, b_bytes, st);                      \
  }
  SEDT_L3(64, 64)
  SEDT_L3(64, 128)
  SEDT_L3(128, 64)
  SEDT_L3(128, 128)
#undef SEDT_L3
  return -1;
}

int wgrad_lds_try(const SedtIgemm& p, hipStream_t st);   // wgrad3.hip

// Envelope checks + tile rule of the LDS-DMA GEMM family, then the lean-issue kernel (igemm3_try).
// Returns -1 when the problem is outside the envelope (the caller - sedt_igemm - then uses the general kernel of igemm.hip).
int igemm_lds_try(const SedtIgemm& p, hipStream_t st) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.trans) {
    return wgrad_lds_try(p, st);
  }
  if ((p.out_f32 != 0) != (p.f32ep != 0) || p.splitk > 1 || p.act == SEDT_ACT_SIGMOID) return -1;      // f32 C only with the f32 epilogue
  if (p.f32ep && ((p.ldr & 3) || (p.ldm & 3) || (reinterpret_cast<uintptr_t>(p.res) & 15) ||
                  (!p.mask_bits && (reinterpret_cast<uintptr_t>(p.mask) & 15)))) return -1;
  if ((p.K & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 7)) return -1;
  if (!al16(p.A) || !al16(p.B) || !al16(p.C)) return -1;
  if (!p.f32ep && p.res && (!al16(p.res) || (p.ldr & 7))) return -1;
  if (!p.f32ep && p.mask && !p.mask_bits && (!al16(p.mask) || (p.ldm & 7))) return -1;
  if (p.conv && (p.Ci % BK2)) return -1;
  // bytes addressable through the A descriptor: every gathered pixel row + one K tile past its start
  long a_rows = p.conv ? (long)((p.M + (long)p.Ho * p.Wo - 1) / ((long)p.Ho * p.Wo)) * p.Hi * p.Wi : (long)p.M;
  long a_bytes = ((a_rows - 1) * p.lda + (p.awrap ? 2 * p.awrap : (p.conv ? p.Ci : p.K))) * 2;
  if (p.awrap && 3 * p.awrap != (p.conv ? p.Ci : p.K)) return -1;
  long b_bytes = ((long)(p.N - 1) * p.ldb + p.K) * 2;
  if (p.btap_on) {
    if (!p.conv || p.KH * p.KW > 8) return -1;
    b_bytes = (long)p.N * p.ldb * 2;                 // (the tap table points anywhere inside a B row)
  }
  if (p.omap && !p.conv) return -1;
  if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31) || a_bytes <= 0 || b_bytes <= 0) return -1;
  int bm = p.tile_m, bn = p.tile_n;
  // measured (tools/tune_igemm.py on MI355X): occupancy beats prefetch depth at every SEDT shape - the 64x64 tile with a
  // 2-stage ring (32 KB LDS, 5 workgroups per CU) wins or ties; SEDT_IGEMM_STAGES / tile_m override for experiments
  if (bm == 0 || bn == 0) {
    bm = 64; bn = 64;
    // measured with the lean-issue kernel (tools/tune_igemm.py): the wider N tile (one A fragment feeds two MFMAs) wins
    // once K is deep and enough tiles remain to fill the chip; everything else prefers the 64x64 tile's occupancy
    // (with the 8-wave form of that tile - igemm3.hip - the tile-count condition of the 4-wave kernel no longer applies)
    // (a grouped launch - igemm3_planning() - runs on the 64x64 program: worth it for the launch-bound decoder-sized problems)
    static int mink = -1;
    if (mink < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_BN128_MINK");
      mink = e ? atoi(e) : 512;
    }
    static int bn128t = -1;
    if (bn128t < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_BN128_MINTILES");
      bn128t = e ? atoi(e) : 250;
    }
    // below 250 tiles of 64x128 the 64x64 tile covers more of the chip: the B = 32 configurations have M = 3968 rows, i.e. 124
    // tiles of 64x128 at N = 256 (measured, tools/dev/sweep_c3.sh: C3 4.97 -> 4.88 ms, C5 8.71 -> 8.61 ms, C2 - exactly 256 tiles -
    // unchanged; a threshold of 300 costs C2 0.2 %)
    const long t128 = (long)((p.M + 63) / 64) * (p.N / 128);
    if ((p.N % 128) == 0 && p.K >= mink && !(igemm3_planning() && p.M <= 1024) && (t128 >= bn128t || p.M <= 1024)) bn = 128;
    // the ping-pong 8-wave kernel makes the 128x128 tile (one workgroup per CU) pay where the K loop is long enough to
    // amortise its exposed prologue / epilogue and the tiles still cover the chip: layer4 conv1 fwd / conv2 / conv3 dgrad
    static int bm128k = -1, bm128t = -1;
    if (bm128k < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_BM128_MINK");
      bm128k = e ? atoi(e) : 2048;
      e = sedt::dev_getenv("SEDT_IGEMM_BM128_MINTILES");
      bm128t = e ? atoi(e) : 256;
    }
    if (bn == 128 && p.K >= bm128k && (long)((p.M + 127) / 128) * (p.N / 128) >= bm128t && !igemm3_planning()) bm = 128;
    // the wide outputs of layer4 (conv3 and the projection: N = 2048) with a short K (512 / 1024 = 8 / 16 K tiles) are prologue- and
    // epilogue-dominated on 64x128 tiles (2048 of them at M = 8192): a 256x128 tile (85 flop per staged byte instead of 43, 512 tiles =
    // two per CU) with the epilogue operands read late (SEDT_IGEMM_BM256=0/1 in the developer build)
    static int bm256 = -1;
    if (bm256 < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_BM256");
      bm256 = e ? atoi(e) : 0;
    }
    if (bm256 && bn == 128 && bm == 64 && p.K >= 512 && p.K <= 1024 && (p.M % 256) == 0 && (long)(p.M / 256) * (p.N / 128) >= 512 &&
        !igemm3_planning() && !p.f32ep)
      bm = 256;
  }
  {   // the lean-issue kernel takes the common cases
    // bit 31 of the B descriptor size = "the first workgroups prefetch the whole B operand" (igemm3_impl): on for every forward / dgrad problem -
    // B is a weight there (same-box A/B in profiles/r06_ab_bpf.txt; SEDT_IGEMM_BPF=0 in the developer build)
    static int bpf = -1;
    if (bpf < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_BPF");
      bpf = e ? atoi(e) : 1;
    }
    // bit 31 of the A descriptor size = "the first workgroups prefetch this program's own code" (igemm3_impl; SEDT_IGEMM_CPF=0 in the developer build;
    // same-box A/B in profiles/r06_ab_bpf.txt)
    static int cpf = -1;
    if (cpf < 0) {
      const char* e = sedt::dev_getenv("SEDT_IGEMM_CPF");
      cpf = e ? atoi(e) : 1;
    }
    int r3 = igemm3_try(p, (unsigned)a_bytes | (cpf ? 0x80000000u : 0u), (unsigned)b_bytes | (bpf ? 0x80000000u : 0u), bm, bn, st);
    if (r3 >= 0) return r3;
    if (igemm3_planning()) return -1;      // dry run (sedt_igemm_group): never launch from here
  }
  return -1;                             // outside the lean-issue envelope: the general kernel (igemm.hip) takes it
}

}  // namespace sedt

#ifdef SEDT_DEV
extern "C" int sedt_dev_ts_filter(int M, int N, int K) {
  const int v[3] = {M, N, K};
  return hipMemcpyToSymbol(HIP_SYMBOL(sedt::g_ts_filter), v, sizeof(v)) == hipSuccess ? 0 : 1;
}
extern "C" int sedt_dev_phase_ts(unsigned long long* out, int nwg) {
  if (nwg > 4096) nwg = 4096;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(sedt::g_phase_ts), sizeof(unsigned long long) * 5 * nwg) == hipSuccess ? 0 : 1;
}
#endif
