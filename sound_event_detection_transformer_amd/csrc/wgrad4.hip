// wgrad4.hip - the 128x128, 8-wave "ping-pong" weight-gradient GEMM for the large problems (layer3/4, FFN: Cout, taps*Cin >= 256).
//
// Same data path as wgrad3 ([pixel][channel] images of dY and X by LDS-DMA into an XOR-swizzled ring, k-contiguous MFMA
// fragments by ds_read_b64_tr_b16, split-K slabs), with the two things that made the forward GEMM (igemm3.hip, PP) faster
// at these sizes:
//   * a 128x128 output tile: 64 flop per byte moved L2 -> LDS instead of 32 for 64x64 (the LDS-DMA path, ~27 B/clk/CU,
//     is what bounds these kernels);
//   * two groups of four waves that each own the whole tile but only two of the four k16 steps of a 64-pixel K tile, and
//     run ONE BARRIER apart: on every SIMD one wave is in its "load" segment (transposing fragment reads + the LDS-DMA
//     issue of a later tile) while the other is in its "math" segment (8 MFMAs).  Ordering rules as in igemm3.hip:
//     a tile is waited for (counted vmcnt) at the end of the load segment of the tile before it; a buffer is restaged
//     one stage after its last reads, which retired (lgkmcnt(0)) before the barrier that ended their segment.
// Envelope: bf16, M % 128 == 0, N % 128 == 0, M >= 256, N >= 256, wgrad3's conv condition; fused bias sums opt-in.
#include <stdlib.h>
#include <algorithm>
#include "wgrad3_body.h"

namespace sedt {

// XOR swizzle of the 16-byte chunk index (4 bits: 16 chunks per 256-byte image row).  A transposing read serves 32 lanes per
// LDS cycle: four consecutive pixel rows x one 64-byte quarter row each - and with 256-byte rows every row starts at bank 0,
// so the four rows must sit in four different quarters: chunk bits 3:2 ^= row bits 1:0.  (The first version re-used wgrad3's
// 128-byte-row swizzle on the low three chunk bits: rows r and r+1 shared a quarter, 49 % of the LDS cycles were bank
// conflicts by SQ_LDS_BANK_CONFLICT.)
__device__ __forceinline__ int w4_swz(int row) { return (row & 3) << 2; }

template <bool CONV, int S, int NA>
__device__ __forceinline__ void wgrad4_impl(const SedtIgemm& p, const unsigned a_bytes, const unsigned b_bytes, const int nmajor,
                                            const int bx, const int by) {
  // NA = 1: 128 x 128 tile; NA = 2: 256 x 128 (two dY images per stage, 85 flop per staged byte instead of 64)
  constexpr int BM = 128 * NA, BN = 128, BKP = 64, NW = 8;
  constexpr int IMG_ROWB = 256;                       // image row: 128 channels of bf16
  constexpr int IMG_BYTES = BKP * IMG_ROWB;           // 16 KB per image per stage
  constexpr int STAGE_BYTES = (NA + 1) * IMG_BYTES;   // NA images of dY, then one of X
  constexpr int B_IMG = NA * IMG_BYTES;
  constexpr int CPR = 16, RPI = 4;                    // 16-byte chunks per image row, rows per DMA instruction
  constexpr int GB = BKP / RPI / NW, GA = NA * GB;    // 2 DMA instructions per wave per image and tile
  constexpr int G = GA + GB;
  constexpr int MI = 2 * NA;                          // 32-row blocks of a wave's sub-tile (64 x 64 or 128 x 64)
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = w3_uniform(t >> 6);
  const int kgrp = wave >> 2;
  const int wm = ((wave & 3) >> 1) * (32 * MI), wn = (wave & 1) * 64;

  const int ntn = p.N / BN, ntm = p.M / BM;
  const int nwg = ntn * ntm;
  int vid;
  {
    const int b = bx, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  int m0, n0;
  if (CONV && (nmajor & 4)) {
    // channel-block-major order (3x3 problems, Ci % 128 == 0): consecutive tiles - one XCD's share - are the NINE TAPS of one 128-channel
    // block of X against ONE dY tile, so that XCD's L2 fetches one channel block of X (the taps re-read it shifted by a few pixel rows)
    // and one dY tile instead of 4-5 column tiles of different blocks and every dY tile
    const int taps = p.KH * p.KW, ncb = p.Ci / BN;
    const int cb = vid / (ntm * taps), rem = vid - cb * (ntm * taps);
    const int mt = rem / taps, tap = rem - mt * taps;
    m0 = mt * BM;
    n0 = (tap * ncb + cb) * BN;
  } else if (nmajor & 1) { n0 = (vid / ntm) * BN; m0 = (vid % ntm) * BM; }
  else { m0 = (vid / ntn) * BM; n0 = (vid % ntn) * BN; }

  const int nkb_total = (p.K + BKP - 1) / BKP;
  int kb_begin = 0, kb_end = nkb_total;
  if (p.splitk > 1) {
    const int per = (nkb_total + p.splitk - 1) / p.splitk;
    kb_begin = by * per;
    kb_end = min(nkb_total, kb_begin + per);
  }
  const int nkb = max(0, kb_end - kb_begin);

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);

  // ---- DMA lanes: instruction `instr` covers image rows [instr*4, +4) x 16 chunks; the physical chunk a lane writes holds
  //      logical chunk lchunk = pchunk ^ w4_swz(row)
  const int drow = lane / CPR, pchunk = lane % CPR;
  unsigned a_poff[GA];
  int a_trow[GA];
  const unsigned a_step = (unsigned)(BKP * p.lda * 2);
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int trow = ((i % GB) * NW + wave) * RPI + drow;
    const int lchunk = pchunk ^ w4_swz(trow);
    a_trow[i] = trow;
    a_poff[i] = (unsigned)((((long)kb_begin * BKP + trow) * p.lda + m0 + (i / GB) * 128 + lchunk * 8) * 2);
  }
  unsigned b_poff[GB];
  int b_ho[GB], b_hoff[GB], b_trow[GB];
  bool b_wok[GB];
  const int step_h = CONV ? BKP / p.Wo : 0;
  const unsigned b_step = CONV ? (unsigned)((long)step_h * p.sh * p.Wi * p.ldb * 2) : (unsigned)(BKP * p.ldb * 2);
  const unsigned b_wrap = CONV ? (unsigned)(((long)p.Hi * p.Wi - (long)p.Ho * p.sh * p.Wi) * p.ldb * 2) : 0u;
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int trow = (i * NW + wave) * RPI + drow;
    const int lchunk = pchunk ^ w4_swz(trow);
    const int j = n0 + lchunk * 8;
    b_trow[i] = trow;
    bool ok = true;
    long off;
    if (CONV) {
      const int tap = j / p.Ci, c = j - tap * p.Ci;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const int pix = kb_begin * BKP + trow;
      const int HoWo = p.Ho * p.Wo;
      const int n = pix / HoWo, rem = pix - n * HoWo;
      const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
      const int wi = wo * p.sw - p.pw + kw * p.dw;
      b_ho[i] = ho;
      b_hoff[i] = kh * p.dh - p.ph;
      ok = (unsigned)wi < (unsigned)p.Wi;
      off = ((((long)n * p.Hi + ho * p.sh + b_hoff[i]) * p.Wi + wi) * p.ldb + c) * 2;
    } else {
      b_ho[i] = 0; b_hoff[i] = 0;
      off = (((long)kb_begin * BKP + trow) * p.ldb + j) * 2;
    }
    b_wok[i] = ok;
    b_poff[i] = (unsigned)off;
  }

  int kbase = kb_begin * BKP;      // first pixel of the tile about to be issued (uniform)
  auto issue = [&](const int stage) {
    unsigned char* st = smem + stage * STAGE_BYTES;
    const int left = p.K - kbase;
    const bool full = left >= BKP;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      unsigned voff = (full || a_trow[i] < left) ? a_poff[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + (i / GB) * IMG_BYTES + (((i % GB) * NW + wave) * RPI) * IMG_ROWB), 16,
                                               voff, 0, 0, 0);
      a_poff[i] += a_step;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned voff = OOB;
      bool ok = b_wok[i] && (full || b_trow[i] < left);
      if (CONV) ok = ok && (unsigned)(b_ho[i] * p.sh + b_hoff[i]) < (unsigned)p.Hi;
      if (ok) voff = b_poff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + B_IMG + ((i * NW + wave) * RPI) * IMG_ROWB), 16, voff, 0, 0, 0);
      b_poff[i] += b_step;
      if (CONV) {
        b_ho[i] += step_h;
        if (b_ho[i] >= p.Ho) { b_ho[i] -= p.Ho; b_poff[i] += b_wrap; }
      }
    }
    kbase += BKP;
  };

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposing fragment reads of this wave's two k16 steps: per-thread constant offsets inside a stage
  const int grp = lane >> 4, s16 = lane & 15;
  const int src_pix = (grp >> 1) * 8 + (s16 >> 2);
  int a_rd[2][MI][2], b_rd[2][2][2];           // [k16 step][32-wide sub-tile][half]
#pragma unroll
  for (int kq = 0; kq < 2; ++kq)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int pixrow = (2 * kgrp + kq) * 16 + src_pix + 4 * h2;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int a_ch = wm + i * 32 + (grp & 1) * 16 + (s16 & 3) * 4;       // channel inside the 128 * NA rows of the tile
        const int aphys = ((a_ch & 127) >> 3) ^ w4_swz(pixrow);
        a_rd[kq][i][h2] = (a_ch >> 7) * IMG_BYTES + pixrow * IMG_ROWB + aphys * 16 + ((a_ch >> 2) & 1) * 8;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int b_ch = wn + j * 32 + (grp & 1) * 16 + (s16 & 3) * 4;
        const int bphys = (b_ch >> 3) ^ w4_swz(pixrow);
        b_rd[kq][j][h2] = B_IMG + pixrow * IMG_ROWB + bphys * 16 + ((b_ch >> 2) & 1) * 8;
      }
    }
  auto tr = [&](const unsigned char* ptr) -> w3_s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)ptr); };
  w3_s16x8 fa[2][MI], fb[2][2];
  auto load_frags = [&](const int stage) {
    const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const w3_s16x4 a0 = tr(st + a_rd[kq][i][0]), a1 = tr(st + a_rd[kq][i][1]);
        fa[kq][i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const w3_s16x4 b0 = tr(st + b_rd[kq][j][0]), b1 = tr(st + b_rd[kq][j][1]);
        fb[kq][j] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  };
  auto mfma_all = [&]() {
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[kq][i]), __builtin_bit_cast(bf16x8, fb[kq][j]),
                                                              acc[i][j], 0, 0, 0);
  };

  // optional bias gradient (column sums of dY) by the column-tile-0 workgroups: thread = (channel, pixel quarter), summed
  // from the dY image in the load segment of either group (all 512 threads take part, each in its own group's segment)
  const bool do_colsum = p.colsum_out != nullptr && n0 == 0;
  float bsum = 0.f;
  constexpr int CS_PARTS = NW * 64 / BM, CS_ROWS = BKP / CS_PARTS;       // pixel ranges per channel: 4 x 16 or 2 x 32 rows
  const int cs_ch = t % BM, cs_q = t / BM;
  auto colsum_tile = [&](const int stage) {
    const unsigned char* st = smem + stage * STAGE_BYTES + (cs_ch >> 7) * IMG_BYTES;
    const int c8 = (cs_ch & 127) >> 3;
#pragma unroll
    for (int r = 0; r < CS_ROWS; ++r) {
      const int pixrow = cs_q * CS_ROWS + r;
      const int phys = c8 ^ w4_swz(pixrow);
      bsum += (float)*reinterpret_cast<const bf16_t*>(st + pixrow * IMG_ROWB + phys * 16 + (cs_ch & 7) * 2);
    }
  };

  // ---- ping-pong ring (see igemm3.hip): load segment | barrier | math segment, group 1 one barrier behind group 0
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
      if (do_colsum) colsum_tile(ph);
      issue((ph + S - 1) % S);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      lds_barrier();
      __builtin_amdgcn_s_setprio(1);
      mfma_all();
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
        if (do_colsum) colsum_tile(ph);
        if (more) issue((ph + S - 1) % S);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfma_all();
        __builtin_amdgcn_s_setprio(0);
      }
    }
  }
  if (kgrp == 0) lds_barrier();
  lds_barrier();

  if (do_colsum) {
    float* red = reinterpret_cast<float*>(smem);
    red[t] = bsum;
    __syncthreads();
    if (t < BM) {
      float sum = red[t];
#pragma unroll
      for (int q = 1; q < CS_PARTS; ++q) sum += red[t + q * BM];
      p.colsum_out[(long)(p.splitk > 1 ? by : 0) * p.M + m0 + t] = sum;
    }
    __syncthreads();
  }

  // ---- epilogue: the two groups' halves of the K sum meet in LDS, then 16-byte rows go to the slab / gradient
  constexpr int CP = BN + 4;
  static_assert((size_t)BM * CP * 4 <= (size_t)S * STAGE_BYTES, "the epilogue tile re-uses the ring");
  float* Cs = reinterpret_cast<float*>(smem);
  const int frow = lane & 31, fhalf = lane >> 5;
  if (kgrp == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          Cs[row * CP + wn + j * 32 + frow] = acc[i][j][r];
        }
  }
  __syncthreads();
  if (kgrp == 1) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
          Cs[row * CP + wn + j * 32 + frow] += acc[i][j][r];
        }
  }
  __syncthreads();
  float* out = p.splitk > 1 ? p.slab + (long)by * p.M * p.N : reinterpret_cast<float*>(p.C);
  const long ldo = p.splitk > 1 ? p.N : p.ldc;
#pragma unroll
  for (int c = 0; c < BM * (BN / 4) / (NW * 64); ++c) {
    const int u = t + c * NW * 64;
    const int trow = u / (BN / 4), cc = (u % (BN / 4)) * 4;
    *reinterpret_cast<float4*>(out + (long)(m0 + trow) * ldo + n0 + cc) = *reinterpret_cast<const float4*>(Cs + trow * CP + cc);
  }
}

template <int S>
__device__ __forceinline__ void wgrad4_body(const SedtIgemm& p, const unsigned a_bytes, const unsigned b_bytes, const int nmajor,
                                            const int bx, const int by) {
  if (nmajor & 2) {                      // 256 x 128 tiles (wg4_wide)
    if (p.conv) wgrad4_impl<true, 3, 2>(p, a_bytes, b_bytes, nmajor, bx, by);
    else wgrad4_impl<false, 3, 2>(p, a_bytes, b_bytes, nmajor, bx, by);
    return;
  }
  if (p.conv) wgrad4_impl<true, S, 1>(p, a_bytes, b_bytes, nmajor, bx, by);
  else wgrad4_impl<false, S, 1>(p, a_bytes, b_bytes, nmajor, bx, by);
}

template <int S>
__global__ __launch_bounds__(512) void wgrad4_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes,
                                                     const int nmajor) {
  wgrad4_body<S>(p, a_bytes, b_bytes, nmajor, blockIdx.x, blockIdx.y);
}

template <int S>
__global__ __launch_bounds__(512) void wgrad4_group_kernel(const WgradGroup g) {
  const int b = blockIdx.x;
  int i = 0;
  while (i + 1 < g.n && b >= g.blk0[i + 1]) ++i;
  const int local = b - g.blk0[i];
  const int nwg = g.nwg[i], sk = g.p[i].splitk > 1 ? g.p[i].splitk : 1;
  if (local >= nwg * sk) return;                       // padding workgroups (problem ranges start on multiples of 8)
  wgrad4_body<S>(g.p[i], g.a_bytes[i], g.b_bytes[i], g.nmajor[i], local % nwg, local / nwg);
}

// ring of S stages x 32 KB (96 / 128 KB) >= the 128 x 132 f32 epilogue tile (66 KB).  SEDT_WGRAD4_STAGES picks the depth.
static int wg4_stages() {
  static int s = -1;
  if (s < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD4_STAGES");
    s = (e && atoi(e) == 3) ? 3 : (e && atoi(e) == 4) ? 4 : 3;
  }
  return s;
}
// rows of a tile for a problem with M output rows (Cout): the 256 x 128 tile where M allows it (SEDT_WGRAD4_BM=128: never).
// Same-box A/B on the C2 step (tools/dev/ab_wgrad4.sh, ms/step): 128-row tiles 5.85-5.86 at every split target; 256-row tiles
// with split-K target 32 / 48 / 64 / 80 / 128 / 192 tiles: 5.79 / 5.71 / 5.67 / 5.74 / 5.81 / 5.96
int wgrad4_tile_m(int M, int N) {
  static int bm = -1, min_tiles = 1;
  if (bm < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD4_BM");
    bm = (e && atoi(e) == 128) ? 128 : 256;
    e = sedt::dev_getenv("SEDT_WGRAD4_WIDE_MIN");               // experiment: the 256-row tile only for problems with at least this many of them
    min_tiles = e ? atoi(e) : 1;
  }
  return (bm == 256 && M % 256 == 0 && (long)(M / 256) * (N / 128) >= min_tiles) ? 256 : 128;
}
static size_t wg4_lds() {
  const size_t narrow = (size_t)wg4_stages() * 2 * 64 * 256, wide = (size_t)3 * 3 * 64 * 256;
  return wgrad4_tile_m(256, 1 << 20) == 256 ? std::max(narrow, wide) : narrow;
}

// shape part of the envelope (sedt_igemm_splitk sizes the split for the 128x128 tiling when this holds)
bool wgrad4_shape_ok(int M, int N) {
  static int mn = -1;
  if (mn < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD4_MIN");
    mn = e ? atoi(e) : 256;     // full-step sweep: 512 -> 6.32, 256 -> 6.27, 128 -> 6.30 ms
  }
  return M >= mn && N >= mn && (M % 128) == 0 && (N % 128) == 0;
}

bool wgrad4_ok(const SedtIgemm& p) {
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD_V4");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  // problems with a fused bias gradient are the transformer linears: measured on the full step they are better off in the
  // 64x64 grouped launch with the rest of their layer (6.05 vs 6.10-6.13 ms); SEDT_WGRAD4_BIAS=1 sends them here (tests do)
  static int with_bias = -1;
  if (with_bias < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD4_BIAS");
    with_bias = (e && e[0] == '1') ? 1 : 0;
  }
  return on && p.trans && wgrad4_shape_ok(p.M, p.N) && (p.colsum_out == nullptr || with_bias) && p.out_f32 &&
         (p.splitk > 1 ? p.slab != nullptr : ((p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0)) &&
         (reinterpret_cast<uintptr_t>(p.slab) & 15) == 0;
}

// channel-block-major tile order for the 3x3 problems (see wgrad4_impl); SEDT_WGRAD4_CBMAJOR=0/1 in the developer build
static bool wg4_cbmajor(const SedtIgemm& p) {
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD4_CBMAJOR");
    // same-box A/B on the C2 step (profiles/r06_ab_wgrad4_cbmajor.txt): FETCH_SIZE of the wgrad4 launches 2.38 -> 2.08 GB (x2-corrected),
    // their L2 hit rate 0.44 -> 0.50, step 5.142-5.173 -> 5.134-5.168 ms: a seventh less fabric traffic and no measurable time - the
    // launches are not bound by it.  On by default because it is never slower
    on = e ? atoi(e) : 1;
  }
  return on && p.conv && p.KH * p.KW > 1 && (p.Ci % 128) == 0 && p.N == p.KH * p.KW * p.Ci;
}

template <typename K>
static int wg4_attr(K kern, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wg4_lds());
  if (e != hipSuccess) {
    set_error("%s: hipFuncSetAttribute failed: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

int launch_wgrad4(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  SEDT_DESCRIBE("wgrad4_kernel<%d>", wg4_stages());
  static bool attr_set = false;
  if (!attr_set) {
    if (wg4_attr(wgrad4_kernel<3>, "wgrad4") || wg4_attr(wgrad4_kernel<4>, "wgrad4")) return 1;
    attr_set = true;
  }
  const int bm = wgrad4_tile_m(p.M, p.N);
  const int nwg = (p.N / 128) * (p.M / bm);
  const int nmajor = (p.N > p.M ? 1 : 0) | (bm == 256 ? 2 : 0);
  const dim3 grid(nwg, p.splitk > 1 ? p.splitk : 1);
  if (wg4_stages() == 4) hipLaunchKernelGGL(wgrad4_kernel<4>, grid, dim3(512), wg4_lds(), st, p, a_bytes, b_bytes, nmajor);
  else hipLaunchKernelGGL(wgrad4_kernel<3>, grid, dim3(512), wg4_lds(), st, p, a_bytes, b_bytes, nmajor);
  return check_launch("wgrad4");
}

// launches the (already validated, all wgrad4_ok) problems of g as one grouped kernel; nwg / blk0 are filled here
int launch_wgrad4_group(WgradGroup& g, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    if (wg4_attr(wgrad4_group_kernel<3>, "wgrad4 group") || wg4_attr(wgrad4_group_kernel<4>, "wgrad4 group")) return 1;
    attr_set = true;
  }
  int blk = 0;
  for (int i = 0; i < g.n; ++i) {
    const SedtIgemm& p = g.p[i];
    const int bm = wgrad4_tile_m(p.M, p.N);
    g.nwg[i] = (p.N / 128) * (p.M / bm);
    g.nmajor[i] = (p.N > p.M ? 1 : 0) | (bm == 256 ? 2 : 0) | (wg4_cbmajor(p) ? 4 : 0);
    g.blk0[i] = blk;
    blk += (g.nwg[i] * (p.splitk > 1 ? p.splitk : 1) + 7) / 8 * 8;
  }
  g.blk0[g.n] = blk;
  if (wg4_stages() == 4) hipLaunchKernelGGL(wgrad4_group_kernel<4>, dim3(blk), dim3(512), wg4_lds(), st, g);
  else hipLaunchKernelGGL(wgrad4_group_kernel<3>, dim3(blk), dim3(512), wg4_lds(), st, g);
  return check_launch("wgrad4_group");
}

}  // namespace sedt
