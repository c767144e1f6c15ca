// wgrad2.hip - bf16 weight-gradient GEMM v2 for gfx950: LDS-DMA + hardware transposing LDS reads.
//
// trans == 1 problems of sedt_igemm:  D[co][j] = sum_pix dY[pix][co] * X[g(pix, tap_j)][c_j],  j = (tap, c).
// Both operands have the reduction index (pixels) as the slow axis in memory, so their natural [pixel][channel] tiles
// are what the LDS-DMA lands (64 pixels x 64 channels, 128-B rows, the same XOR swizzle as igemm2.hip).  The
// k-contiguous MFMA fragments come from ds_read_b64_tr_b16: within a 16-lane group, result lane l / element j is
// element (l&3) of source lane 4j + (l&15)/4 (probed on gfx950: tools/probes/tr_read.hip).  So source lane s reads the
// 4 channels [4(s&3), +4) of pixel (s>>2), and result lane l holds channel (l&15) of pixels 0..3; two such reads give
// the 8 pixels one v_mfma_f32_32x32x16_bf16 operand needs - no register transposes, no VGPR staging.
// Split-K partial tiles go straight from the accumulators to the f32 slab (sedt_wgrad_reduce finishes them).
#include <stdlib.h>
#include "igemm2_common.h"

namespace sedt {

int wgrad3_try(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st);   // wgrad3.hip

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int STAGES>
__global__ __launch_bounds__(256) void wgrad2_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes) {
  constexpr int BM = 64, BN = 64, BKP = 64;          // output tile (channels x channels), pixels per K tile
  constexpr int STAGE_BYTES = 2 * BKP * ROWB;        // A image [64 pix][64 ch] + B image [64 pix][64 ch]
  constexpr int GA = 2, GB = 2, G = GA + GB;         // DMA instructions per wave per tile (8 pixel rows each, 4 waves)
  constexpr unsigned OOB = 0xFFFFFF00u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  int vid;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int m0 = (vid / ntn) * BM, n0 = (vid % ntn) * BN;

  const int nkb_total = (p.K + BKP - 1) / BKP;
  int kb_begin = 0, kb_end = nkb_total;
  if (p.splitk > 1) {
    const int per = (nkb_total + p.splitk - 1) / p.splitk;
    kb_begin = blockIdx.y * per;
    kb_end = min(nkb_total, kb_begin + per);
  }
  const int nkb = max(0, kb_end - kb_begin);

  Geom2 g{p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.KH, p.KW, p.sh, p.sw, p.ph, p.pw, p.dh, p.dw, 0};
  const int HoWo = p.conv ? p.Ho * p.Wo : 1;
  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);

  // DMA lane roles: instruction i of this wave covers pixel rows [(i*4+wave)*8, +8) of the K tile; lane -> (row, phys chunk)
  const int lrow = lane >> 3, pc = lane & 7;
  int a_col[GA], b_col[GB], b_kh[GB], b_kw[GB], b_n[GB], b_ho[GB], b_wo[GB];
  bool a_ok[GA], b_ok[GB];
  const int step_h = p.conv ? BKP / p.Wo : 0, step_w = p.conv ? BKP % p.Wo : 0;
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int trow = (i * 4 + wave) * 8 + lrow;
    const int lc = (pc ^ ((trow >> 1) & 7)) * 8;      // logical channel offset inside the tile
    a_col[i] = m0 + lc;
    a_ok[i] = a_col[i] < p.M;
    const int j = n0 + lc;
    b_ok[i] = j < p.N;
    if (p.conv) {
      const int tap = j / p.Ci;
      b_col[i] = j - tap * p.Ci;
      b_kh[i] = tap / p.KW;
      b_kw[i] = tap - b_kh[i] * p.KW;
      const int pix = kb_begin * BKP + trow;
      const int n = pix / HoWo, rem = pix - n * HoWo;
      b_n[i] = n;
      b_ho[i] = rem / p.Wo;
      b_wo[i] = rem - b_ho[i] * p.Wo;
    } else {
      b_col[i] = j; b_kh[i] = 0; b_kw[i] = 0; b_n[i] = 0; b_ho[i] = 0; b_wo[i] = 0;
    }
  }

  // tiles must be issued in increasing order: the gathered operand's (n, ho, wo) advance incrementally by BKP pixels
  auto issue = [&](int kb) {
    unsigned char* st = smem + ((kb - kb_begin) % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int trow = (i * 4 + wave) * 8 + lrow;
      const int pix = kb * BKP + trow;
      unsigned voff = OOB;
      if (a_ok[i] && pix < p.K) voff = (unsigned)(((long)pix * p.lda + a_col[i]) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + ((i * 4 + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int trow = (i * 4 + wave) * 8 + lrow;
      const int pix = kb * BKP + trow;
      unsigned voff = OOB;
      if (b_ok[i] && pix < p.K) {
        long gp = p.conv ? gather_pix2(g, b_n[i], b_ho[i], b_wo[i], b_kh[i], b_kw[i]) : (long)pix;
        if (gp >= 0) voff = (unsigned)((gp * p.ldb + b_col[i]) * 2);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + (BKP + (i * 4 + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
      if (p.conv) {                                   // advance this lane's pixel by BKP for the next tile
        b_wo[i] += step_w;
        if (b_wo[i] >= p.Wo) { b_wo[i] -= p.Wo; b_ho[i] += 1; }
        b_ho[i] += step_h;
        while (b_ho[i] >= p.Ho) { b_ho[i] -= p.Ho; b_n[i] += 1; }
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // optional bias gradient: the n-tile-0 workgroups also sum their dY tile over the pixels (thread -> channel t&63,
  // pixel quarter t>>6; 16 two-byte LDS reads per K tile)
  const bool do_colsum = p.colsum_out != nullptr && n0 == 0;
  float bsum = 0.f;
  const int cs_ch = t & 63, cs_q = t >> 6;
  auto colsum_tile = [&](int kb) {
    const unsigned char* st = smem + ((kb - kb_begin) % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pixrow = cs_q * 16 + r;
      const int phys = (cs_ch >> 3) ^ ((pixrow >> 1) & 7);
      bsum += (float)*reinterpret_cast<const bf16_t*>(st + pixrow * ROWB + phys * 16 + (cs_ch & 7) * 2);
    }
  };

  // transposing fragment reads: lane -> source role (pixel s>>2, channel quad s&3) inside its 16-lane group
  const int grp = lane >> 4, s16 = lane & 15;
  const int src_pix = (grp >> 1) * 8 + (s16 >> 2);   // + 4*h + 16*ks
  const int a_ch = wm + (grp & 1) * 16 + (s16 & 3) * 4;
  const int b_ch = wn + (grp & 1) * 16 + (s16 & 3) * 4;
  auto frag = [&](const unsigned char* img, int ch, int pixrow) -> s16x4 {
    const int phys = ((ch >> 3) ^ ((pixrow >> 1) & 7));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + pixrow * ROWB + phys * 16 + ((ch >> 2) & 1) * 8));
  };
  auto compute = [&](int kb) {
    const unsigned char* st = smem + ((kb - kb_begin) % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BKP / 16; ++ks) {
      const int p0 = ks * 16 + src_pix;
      const s16x4 a0 = frag(st, a_ch, p0), a1 = frag(st, a_ch, p0 + 4);
      const s16x4 b0 = frag(st + BKP * ROWB, b_ch, p0), b1 = frag(st + BKP * ROWB, b_ch, p0 + 4);
      s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
    }
  };

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nkb) issue(kb_begin + s);
  for (int it = 0; it < nkb; ++it) {
    if (it + STAGES - 2 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (it + STAGES - 1 < nkb) issue(kb_begin + it + STAGES - 1);
    compute(kb_begin + it);
    if (do_colsum) colsum_tile(kb_begin + it);
  }
  if (do_colsum) {
    lds_barrier();                      // ring no longer read: reuse it for the 4-way reduction
    float* red = reinterpret_cast<float*>(smem);
    red[t] = bsum;
    __syncthreads();
    if (t < 64 && m0 + t < p.M)
      p.colsum_out[(long)(p.splitk > 1 ? blockIdx.y : 0) * p.M + m0 + t] = red[t] + red[t + 64] + red[t + 128] + red[t + 192];
  }

  float* out = p.splitk > 1 ? p.slab + (long)blockIdx.y * p.M * p.N : reinterpret_cast<float*>(p.C);
  const long ldo = p.splitk > 1 ? p.N : p.ldc;
  const int col = n0 + wn + (lane & 31);
  if (col < p.N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < p.M) out[(long)row * ldo + col] = acc[r];
    }
  }
}

template <int STAGES>
static int launch_wgrad2(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  constexpr size_t lds = (size_t)STAGES * 2 * 64 * ROWB;
  static bool attr_set = false;
  auto kern = wgrad2_kernel<STAGES>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("wgrad2: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + 63) / 64) * ((p.M + 63) / 64);
  hipLaunchKernelGGL(kern, dim3(nwg, p.splitk > 1 ? p.splitk : 1), dim3(256), lds, st, p, a_bytes, b_bytes);
  return check_launch("wgrad2");
}

// 0 when the problem fits the LDS-DMA weight-gradient kernels (v2 / v3); fills the buffer-descriptor sizes
int wgrad2_envelope(const SedtIgemm& p, long* a_bytes_out, long* b_bytes_out) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.scale || p.bias || p.res || p.mask || p.act != SEDT_ACT_NONE || p.drop_p > 0.f || p.alpha != 1.f) return -1;
  if (p.splitk <= 1 && !p.out_f32) return -1;
  if (p.splitk > 1 && !p.slab) return -1;
  if ((p.M & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7)) return -1;
  if (!al16(p.A) || !al16(p.B)) return -1;
  if (p.conv && ((p.Ci & 7) || p.transposed)) return -1;
  long a_bytes = ((long)(p.K - 1) * p.lda + p.M) * 2;
  long b_rows = p.conv ? (long)((p.K + (long)p.Ho * p.Wo - 1) / ((long)p.Ho * p.Wo)) * p.Hi * p.Wi : (long)p.K;
  long b_bytes = ((b_rows - 1) * p.ldb + (p.conv ? p.Ci : p.N)) * 2;
  if (a_bytes >= (1L << 31) || b_bytes >= (1L << 31) || a_bytes <= 0 || b_bytes <= 0) return -1;
  *a_bytes_out = a_bytes;
  *b_bytes_out = b_bytes;
  return 0;
}

// returns -1 when the problem is outside the envelope (the caller then uses the general v1 kernel)
int wgrad2_try(const SedtIgemm& p, hipStream_t st) {
  long a_bytes, b_bytes;
  if (wgrad2_envelope(p, &a_bytes, &b_bytes) != 0) return -1;
  {   // the lean-issue kernel takes the common cases
    int r3 = wgrad3_try(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    if (r3 >= 0) return r3;
  }
  static int stages = -1;
  if (stages < 0) {
    const char* e = getenv("SEDT_WGRAD_STAGES");
    stages = e ? atoi(e) : 0;
  }
  if (stages == 3) return launch_wgrad2<3>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  if (stages == 4) return launch_wgrad2<4>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  return launch_wgrad2<2>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
}

}  // namespace sedt
