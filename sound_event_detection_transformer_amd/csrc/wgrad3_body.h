// wgrad3_body.h - the workgroup program of the lean-issue bf16 weight-gradient GEMM (see wgrad3.hip for the design notes)
// as a device function, so that it can run as its own kernel (wgrad3.hip), as one problem of a grouped launch, or in the
// spare workgroups of a forward / dgrad GEMM launch (igemm3.hip: co-scheduling).
#pragma once
#include "lds_gemm_common.h"

namespace sedt {

typedef __attribute__((ext_vector_type(4))) short w3_s16x4;
typedef __attribute__((ext_vector_type(8))) short w3_s16x8;
typedef __attribute__((address_space(3))) w3_s16x4 w3_lds_s16x4;

__device__ __forceinline__ int w3_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// XOR swizzle of the 16-byte chunk index of image row `row`.  A transposing read (ds_read_b64_tr_b16) serves lanes 0-31 in
// one LDS cycle and they touch rows r, r+1, r+2, r+3 x 64 bytes: rows r and r+2 start in the same 32 banks, so their
// swizzles must differ in chunk bit 2 (the other 64-byte half of the row) - the bit-reversed row pair index does that
// (the plain (row>>1)&7 of the b128-read kernels gave 2-way conflicts here: 50 % of the LDS cycles by SQ_LDS_BANK_CONFLICT).
__device__ __forceinline__ int w3_swz(int row) {
  const int v = (row >> 1) & 7;
  return ((v & 1) << 2) | (v & 2) | (v >> 2);
}

// the workgroup program; bx = tile index of this workgroup inside its problem, by = its split-K slice
// nmajor: 0 = m-major tile order, 1 = n-major, both over XCD-contiguous tile ranges; 2/3 = the same orders with bx used as
// the tile id directly (K-slice mapping: the caller already placed all tiles of one K slice on one XCD)
template <int BN, bool CONV>   // BM = 64 output channels, BN = 64 or 128 columns of (tap, cin); CONV: gathered (tap) B operand
__device__ __forceinline__ void wgrad3_impl(const SedtIgemm& p, const unsigned a_bytes, const unsigned b_bytes, const int nmajor,
                                            const int bx, const int by) {
  constexpr int BM = 64, BKP = 64;
  constexpr int NI = BN / 64;                         // 32-wide column tiles per wave (wave tile 32 x BN/2)
  constexpr int A_BYTES = BKP * ROWB;                 // dY image: 64 pixels x 64 channels
  constexpr int B_ROWB = BN * 2;                      // X image row: BN channels
  constexpr int STAGE_BYTES = A_BYTES + BKP * B_ROWB;
  constexpr int GA = 2, GB = BN / 32;                 // DMA instructions per wave per tile (A: 8 pixel rows each; B: see below)
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = w3_uniform(t >> 6);
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * (BN / 2);

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  int vid = bx;
  if (nmajor < 2) {
    const int b = bx, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  // an XCD owns a contiguous run of tile ids.  When X (the N side: taps*Cin columns) is the larger operand, give each XCD
  // a few column tiles x all channel tiles (n-major) so X is streamed from HBM once in total instead of once per XCD.
  int m0, n0;
  if (nmajor & 1) { n0 = (vid / ntm) * BN; m0 = (vid % ntm) * BM; }
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

  // ---- A (dY) DMA lanes: instruction i covers pixel rows [(i*4+wave)*8, +8) x 8 chunks of 8 channels
  const int lrow = lane >> 3, pc = lane & 7;
  unsigned a_poff[GA];
  int a_trow[GA];
  bool a_ok[GA];
  const unsigned a_step = (unsigned)(BKP * p.lda * 2);
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int trow = (i * 4 + wave) * 8 + lrow;
    const int col = m0 + (pc ^ w3_swz(trow)) * 8;
    a_trow[i] = trow;
    a_ok[i] = col < p.M;
    a_poff[i] = (unsigned)((((long)kb_begin * BKP + trow) * p.lda + col) * 2);
  }
  // ---- B (X, gathered) DMA lanes.  Image rows are BN channels wide: BN = 64 -> 8 chunks per row, 8 rows per instruction;
  //      BN = 128 -> 16 chunks per row, 4 rows per instruction.  Swizzle on the low 3 chunk bits only (within 128-B halves).
  constexpr int B_CPR = BN / 8, B_RPI = 64 / B_CPR;   // chunks per row, rows per instruction
  constexpr int B_INSTR = BKP / B_RPI;                // instructions per tile (8 or 16) -> per wave GB = B_INSTR / 4
  static_assert(B_INSTR / 4 == GB, "B DMA split");
  unsigned b_poff[GB];
  int b_ho[GB], b_hoff[GB], b_trow[GB];
  bool b_wok[GB];
  const int step_h = CONV ? BKP / p.Wo : 0;
  const unsigned b_step = CONV ? (unsigned)((long)step_h * p.sh * p.Wi * p.ldb * 2) : (unsigned)(BKP * p.ldb * 2);
  const unsigned b_wrap = CONV ? (unsigned)(((long)p.Hi * p.Wi - (long)p.Ho * p.sh * p.Wi) * p.ldb * 2) : 0u;
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int instr = i * 4 + wave;
    const int trow = instr * B_RPI + lane / B_CPR;
    const int pchunk = lane % B_CPR;
    const int lchunk = (pchunk & ~7) | ((pchunk & 7) ^ w3_swz(trow));
    const int j = n0 + lchunk * 8;
    b_trow[i] = trow;
    bool ok = j < p.N;
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
      ok = ok && (unsigned)wi < (unsigned)p.Wi;
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
    const int left = p.K - kbase;                    // pixels of this tile inside K
    const bool full = left >= BKP;                   // uniform: every tile but the last skips the per-row bound checks
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      unsigned voff = OOB;
      if (a_ok[i] && (full || a_trow[i] < left)) voff = a_poff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + ((i * 4 + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
      a_poff[i] += a_step;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned voff = OOB;
      bool ok = b_wok[i] && (full || b_trow[i] < left);
      if (CONV) ok = ok && (unsigned)(b_ho[i] * p.sh + b_hoff[i]) < (unsigned)p.Hi;
      if (ok) voff = b_poff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + A_BYTES + ((i * 4 + wave) * B_RPI) * B_ROWB), 16, voff, 0, 0, 0);
      b_poff[i] += b_step;
      if (CONV) {
        b_ho[i] += step_h;
        if (b_ho[i] >= p.Ho) { b_ho[i] -= p.Ho; b_poff[i] += b_wrap; }
      }
    }
    kbase += BKP;
  };

  f32x16 acc[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- transposing fragment reads: per-thread constant offsets inside a stage
  const int grp = lane >> 4, s16 = lane & 15;
  const int src_pix = (grp >> 1) * 8 + (s16 >> 2);
  const int a_ch = wm + (grp & 1) * 16 + (s16 & 3) * 4;
  int a_rd[4][2], b_rd[4][NI][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int pixrow = ks * 16 + src_pix + 4 * h2;
      a_rd[ks][h2] = pixrow * ROWB + (((a_ch >> 3) ^ w3_swz(pixrow)) * 16) + ((a_ch >> 2) & 1) * 8;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int b_ch = wn + j * 32 + (grp & 1) * 16 + (s16 & 3) * 4;
        const int ch8 = b_ch >> 3;
        const int phys = (ch8 & ~7) | ((ch8 & 7) ^ w3_swz(pixrow));
        b_rd[ks][j][h2] = A_BYTES + pixrow * B_ROWB + phys * 16 + ((b_ch >> 2) & 1) * 8;
      }
    }
  auto tr = [&](const unsigned char* ptr) -> w3_s16x4 { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_s16x4*)ptr); };
  auto compute = [&](const int stage) {
    const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const w3_s16x4 a0 = tr(st + a_rd[ks][0]), a1 = tr(st + a_rd[ks][1]);
      const w3_s16x8 av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);   // register concatenation, no ALU
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const w3_s16x4 b0 = tr(st + b_rd[ks][j][0]), b1 = tr(st + b_rd[ks][j][1]);
        const w3_s16x8 bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[j], 0, 0, 0);
      }
    }
  };

  // optional bias gradient (column sums of dY) by the n-tile-0 workgroups
  const bool do_colsum = p.colsum_out != nullptr && n0 == 0;
  float bsum = 0.f;
  const int cs_ch = t & 63, cs_q = t >> 6;
  auto colsum_tile = [&](const int stage) {
    const unsigned char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pixrow = cs_q * 16 + r;
      const int phys = (cs_ch >> 3) ^ w3_swz(pixrow);
      bsum += (float)*reinterpret_cast<const bf16_t*>(st + pixrow * ROWB + phys * 16 + (cs_ch & 7) * 2);
    }
  };

  // ---- WS-stage ring, unrolled by its depth (stage indices are literals); a wave waits only for its own oldest tile:
  //      (WS-2) later tiles x (GA+GB) DMA instructions stay in flight across the barrier
  constexpr int WS = 2;     // measured: a 3-stage ring (48 KB, 3 workgroups per CU) is 2.5 % slower on the full step
  constexpr int G = GA + GB;
#pragma unroll
  for (int s0 = 0; s0 < WS - 1; ++s0)
    if (s0 < nkb) issue(s0);
  int it = 0;
  for (; it + WS <= nkb; it += WS) {
#pragma unroll
    for (int ph = 0; ph < WS; ++ph) {
      const bool more = it + ph + WS - 1 < nkb;
      if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WS - 2) * G) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (more) issue((ph + WS - 1) % WS);
      compute(ph);
      if (do_colsum) colsum_tile(ph);
    }
  }
#pragma unroll
  for (int ph = 0; ph < WS - 1; ++ph)
    if (it + ph < nkb) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      compute(ph);
      if (do_colsum) colsum_tile(ph);
    }
  if (do_colsum) {
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem);
    red[t] = bsum;
    __syncthreads();
    if (t < 64 && m0 + t < p.M)
      p.colsum_out[(long)(p.splitk > 1 ? by : 0) * p.M + m0 + t] = red[t] + red[t + 64] + red[t + 128] + red[t + 192];
  }

  float* out = p.splitk > 1 ? p.slab + (long)by * p.M * p.N : reinterpret_cast<float*>(p.C);
  const long ldo = p.splitk > 1 ? p.N : p.ldc;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    if (col < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < p.M) out[(long)row * ldo + col] = acc[j][r];
      }
    }
  }
}

template <int BN>
__device__ __forceinline__ void wgrad3_body(const SedtIgemm& p, const unsigned a_bytes, const unsigned b_bytes, const int nmajor,
                                            const int bx, const int by) {
  if (p.conv) wgrad3_impl<BN, true>(p, a_bytes, b_bytes, nmajor, bx, by);      // uniform branch: two specialised programs
  else wgrad3_impl<BN, false>(p, a_bytes, b_bytes, nmajor, bx, by);
}

// several independent weight-gradient problems in ONE launch: the wgrads of a layer are small (a ResNet block: 3-4
// problems of 20-60 us, a transformer layer: 6-9 problems of < 10 us) and only needed by the optimizer, so they are
// collected and issued together - fewer graph nodes, and the chip is filled by the union of their tiles.
constexpr int WG_MAXG = 10;
struct WgradGroup {
  int n;
  int blk0[WG_MAXG + 1];                 // first workgroup of each problem
  int nwg[WG_MAXG];                      // tiles per problem (x split-K slices = workgroups)
  int nmajor[WG_MAXG];
  unsigned a_bytes[WG_MAXG], b_bytes[WG_MAXG];
  SedtIgemm p[WG_MAXG];
};

static_assert(sizeof(WgradGroup) <= 3900, "WgradGroup must fit the 4 KB kernel-argument segment");

// which problem of the group does workgroup `b` (counted from the group's first workgroup) belong to
__device__ __forceinline__ void wgrad_group_run(const WgradGroup& g, const int b) {
  int i = 0;
  while (i + 1 < g.n && b >= g.blk0[i + 1]) ++i;
  const int local = b - g.blk0[i];
  const int nwg = g.nwg[i], sk = g.p[i].splitk > 1 ? g.p[i].splitk : 1;
  if (local >= nwg * sk) return;                       // padding workgroups (problem ranges start on multiples of 8)
  if (g.nmajor[i] >= 2) {
    // K-slice mapping: workgroup ids are dealt to the 8 XCDs round-robin and blk0 is a multiple of 8, so (local & 7) is
    // the XCD; it works through K slices xcd, xcd + 8, ... - all tiles of a slice before the next
    const int j = local >> 3;
    wgrad3_body<64>(g.p[i], g.a_bytes[i], g.b_bytes[i], g.nmajor[i], j % nwg, (local & 7) + 8 * (j / nwg));
  } else {
    wgrad3_body<64>(g.p[i], g.a_bytes[i], g.b_bytes[i], g.nmajor[i], local % nwg, local / nwg);
  }
}

}  // namespace sedt
