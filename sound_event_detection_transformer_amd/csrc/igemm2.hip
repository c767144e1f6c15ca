// igemm2.hip - bf16 implicit GEMM v2 for gfx950: LDS-DMA multi-stage pipeline.
//
// Same contract as igemm.hip (trans == 0 problems: linear / conv forward / conv dgrad with the fused epilogue), but the
// operand staging no longer goes through VGPRs:
//   * global -> LDS by `buffer_load_dwordx4 ... lds` (LDS-DMA): one wave instruction lands 8 tile rows x 128 B.
//     Out-of-image conv taps, rows >= M and the K tail are redirected to an out-of-range buffer offset, for which the
//     hardware writes zeros - no zero page, no branches.
//   * an S-stage LDS ring ([BM+BN] rows x 64 k per stage); tile t+S-1 is issued while tile t is multiplied.  Each wave
//     waits only for ITS OWN oldest tile with a counted `s_waitcnt vmcnt(N)` and one raw `s_barrier` per K tile makes
//     every wave's rows visible (never `__syncthreads()`, which would drain the DMA queue).
//   * LDS rows are unpadded (the DMA image is lane-linear), so the 16-B chunk index is XOR-swizzled with (row>>1)&7 on
//     the SOURCE address and on the ds_read_b128 address: the 16 rows of a read group hit 16 distinct 16-B slots.
//   * epilogue: accumulators -> LDS (f32, padded pitch) -> each lane finishes 8 consecutive columns of a row and
//     stores 16 B (residual / ReLU-mask loads are 16 B too) instead of 2-byte strided stores.
//   * workgroup order is XCD-aware: consecutive tiles of an M panel run on the same XCD and share its L2.
#include <stdlib.h>
#include "igemm2_common.h"

namespace sedt {

int wgrad2_try(const SedtIgemm& p, hipStream_t st);   // wgrad2.hip
bool igemm3_planning();                                 // igemm3.hip
int igemm3_try(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, int bm, int bn, hipStream_t st);   // igemm3.hip

template <int BM, int BN, int STAGES>
__global__ __launch_bounds__(256) void igemm2_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes,
                                                     const int nmajor) {
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  constexpr int GA = BM / 32, GB = BN / 32;     // DMA instructions per wave per tile for A / B (8 rows each, 4 waves)
  constexpr int G = GA + GB;
  constexpr unsigned OOB = 0xFFFFFF00u;         // beyond any num_records (< 2^31): the DMA writes zeros
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;

  // ---- XCD-aware tile order: hardware places block b on XCD b % 8; give every XCD a contiguous run of tiles
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int nwg = ntn * ntm;
  int vid;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  // each XCD owns a contiguous run of tile ids (its own L2).  m-major runs (n fastest) keep an A panel resident and re-read
  // all of B per M tile - right while B (weights) fits the 4 MB L2; when it does not (layer4 3x3: 4.7 MB), n-major runs
  // keep a B panel resident and stream A instead.
  int m0, n0;
  if (nmajor) { n0 = (vid / ntm) * BN; m0 = (vid % ntm) * BM; }
  else { m0 = (vid / ntn) * BM; n0 = (vid % ntn) * BN; }

  const int nkb = (p.K + BK2 - 1) / BK2;
  Geom2 g{p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.KH, p.KW, p.sh, p.sw, p.ph, p.pw, p.dh, p.dw, p.transposed};
  const int HoWo = p.conv ? p.Ho * p.Wo : 1;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);

  // ---- per-lane DMA rows: instruction i of this wave covers tile rows [ (i*4 + wave)*8, +8 ); lane -> (row, phys chunk)
  const int lrow = lane >> 3, pc = lane & 7;
  int a_n[GA], a_ho[GA], a_wo[GA];
  unsigned a_swz[GA];
  bool a_ok[GA];
#pragma unroll
  for (int i = 0; i < GA; ++i) {
    const int trow = (i * 4 + wave) * 8 + lrow;
    const int row = m0 + trow;
    a_ok[i] = row < p.M;
    a_swz[i] = (unsigned)(pc ^ ((trow >> 1) & 7)) * 8u;   // logical k offset (elements) this lane fetches
    if (p.conv) {
      const int n = row / HoWo, rem = row - n * HoWo;
      a_n[i] = n;
      a_ho[i] = rem / p.Wo;
      a_wo[i] = rem - a_ho[i] * p.Wo;
    } else {
      a_n[i] = row; a_ho[i] = 0; a_wo[i] = 0;
    }
  }
  unsigned b_base[GB], b_swz[GB];
  bool b_ok[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int trow = (i * 4 + wave) * 8 + lrow;
    const int row = n0 + trow;
    b_ok[i] = row < p.N;
    b_swz[i] = (unsigned)(pc ^ ((trow >> 1) & 7)) * 8u;
    b_base[i] = (unsigned)((long)row * p.ldb * 2);
  }

  auto issue_a = [&](int kb) {
    const int k0 = kb * BK2;
    unsigned char* st = smem + (kb % STAGES) * STAGE_BYTES;
    int kh = 0, kw = 0, c0 = k0;
    if (p.conv) {
      const int tap = k0 / p.Ci;
      c0 = k0 - tap * p.Ci;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
    }
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      long pix = -1;
      if (a_ok[i]) pix = p.conv ? gather_pix2(g, a_n[i], a_ho[i], a_wo[i], kh, kw) : (long)a_n[i];
      unsigned voff = OOB;
      if (pix >= 0 && (k0 + (int)a_swz[i]) < p.K) voff = (unsigned)((pix * p.lda + c0 + a_swz[i]) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(st + ((i * 4 + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
    }
  };
  auto issue_b = [&](int kb) {
    const int k0 = kb * BK2;
    unsigned char* st = smem + (kb % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      unsigned voff = OOB;
      if (b_ok[i] && (k0 + (int)b_swz[i]) < p.K) voff = b_base[i] + (unsigned)((k0 + b_swz[i]) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + (BM + (i * 4 + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addressing: row = w? + i*32 + (lane&31); logical chunk = kk/8 + (lane>>5); physical = chunk ^ ((row>>1)&7)
  const int frow = lane & 31, fhalf = lane >> 5;
  // multiply tile kb; the DMA of tile `nxt` (>= 0) is issued between the k-steps so its address math overlaps the MFMAs
  auto compute = [&](int kb, int nxt) {
    const unsigned char* st = smem + (kb % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK2 / 16; ++ks) {
      bf16x8 a[MI], b[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm + i * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> 1) & 7);
        a[i] = *reinterpret_cast<const bf16x8*>(st + row * ROWB + ch * 16);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn + j * 32 + frow;
        const int ch = (ks * 2 + fhalf) ^ ((row >> 1) & 7);
        b[j] = *reinterpret_cast<const bf16x8*>(st + (BM + row) * ROWB + ch * 16);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (nxt >= 0) {
        if (ks == 0) issue_a(nxt);
        if (ks == 1) issue_b(nxt);
      }
    }
  };

  // ---- pipeline
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nkb) { issue_a(s); issue_b(s); }
  for (int it = 0; it < nkb; ++it) {
    // tiles issued after `it` and still allowed in flight: min(STAGES-2, nkb-1-it)
    if (it + STAGES - 2 < nkb) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    compute(it, (it + STAGES - 1 < nkb) ? it + STAGES - 1 : -1);
  }
  lds_barrier();   // everyone is done reading the ring: reuse it for the C tile

  // ---- epilogue through LDS: f32 tile [BM][BN + 4]
  constexpr int CP = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        Cs[row * CP + wn + j * 32 + frow] = acc[i][j][r];
      }
  __syncthreads();

  const uint32_t seed = eff_seed(p.seed, p.seed_ptr);
  const uint32_t thresh = drop_threshold(p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const bf16_t* resT = reinterpret_cast<const bf16_t*>(p.res);
  const bf16_t* maskT = reinterpret_cast<const bf16_t*>(p.mask);
  bf16_t* outT = reinterpret_cast<bf16_t*>(p.C);
  constexpr int CPR = BN / 8;                   // 8-column chunks per tile row
  for (int u = t; u < BM * CPR; u += 256) {
    const int trow = u / CPR, cc = (u % CPR) * 8;
    const int row = m0 + trow, col = n0 + cc;
    if (row >= p.M || col >= p.N) continue;     // N % 8 == 0 is required by the launcher: chunks never straddle N
    float v[8];
    {
      const float4 x0 = *reinterpret_cast<const float4*>(Cs + trow * CP + cc);
      const float4 x1 = *reinterpret_cast<const float4*>(Cs + trow * CP + cc + 4);
      v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
    }
    if (p.scale) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.scale[col + e];
    }
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += p.bias[col + e];
    }
    if (!p.act_post_res && p.act == SEDT_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.drop_p > 0.f) {
      const uint64_t base = (uint64_t)row * (uint64_t)p.N + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = drop_keep(seed, base + e, thresh) ? v[e] * inv_keep : 0.f;
    }
    if (resT) {
      const long rr = p.res_mod > 0 ? (row % p.res_mod) : row;
      const bf16x8 rv = *reinterpret_cast<const bf16x8*>(resT + rr * p.ldr + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
    }
    if (p.act_post_res && p.act == SEDT_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (maskT) {
      const bf16x8 mv = *reinterpret_cast<const bf16x8*>(maskT + (long)row * p.ldm + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ((float)mv[e] > 0.f) ? v[e] : 0.f;
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] * p.alpha);
    *reinterpret_cast<bf16x8*>(outT + (long)row * p.ldc + col) = o;
  }
}

template <int BM, int BN, int STAGES>
static int launch2(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * ROWB;
  constexpr size_t ctile = (size_t)BM * (BN + 4) * sizeof(float);
  constexpr size_t lds = ring > ctile ? ring : ctile;
  static bool attr_set = false;
  auto kern = igemm2_kernel<BM, BN, STAGES>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm2: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  static int force = -2;
  if (force == -2) {
    const char* e = getenv("SEDT_IGEMM_NMAJOR");
    force = e ? atoi(e) : -1;
  }
  const int nmajor = force > 0 ? 1 : 0;   // measured on the full step: m-major wins even for the 4.7 MB layer4 3x3 weights
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, st, p, a_bytes, b_bytes, nmajor);
  return check_launch("igemm2");
}

// returns -1 when the problem is outside v2's envelope (the caller then uses the general v1 kernel)
int igemm2_try(const SedtIgemm& p, hipStream_t st) {
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if (p.trans) {
    static int wg2 = -1;
    if (wg2 < 0) {
      const char* e = getenv("SEDT_WGRAD_V2");
      wg2 = (e && e[0] == '0') ? 0 : 1;
    }
    return wg2 ? wgrad2_try(p, st) : -1;
  }
  if (p.out_f32 || p.splitk > 1 || p.act == SEDT_ACT_SIGMOID) return -1;
  if ((p.K & 7) || (p.N & 7) || (p.lda & 7) || (p.ldb & 7) || (p.ldc & 7)) return -1;
  if (!al16(p.A) || !al16(p.B) || !al16(p.C)) return -1;
  if (p.res && (!al16(p.res) || (p.ldr & 7))) return -1;
  if (p.mask && (!al16(p.mask) || (p.ldm & 7))) return -1;
  if (p.conv && (p.Ci % BK2)) return -1;
  // bytes addressable through the A descriptor: every gathered pixel row + one K tile past its start
  long a_rows = p.conv ? (long)((p.M + (long)p.Ho * p.Wo - 1) / ((long)p.Ho * p.Wo)) * p.Hi * p.Wi : (long)p.M;
  long a_bytes = ((a_rows - 1) * p.lda + (p.conv ? p.Ci : p.K)) * 2;
  long b_bytes = ((long)(p.N - 1) * p.ldb + p.K) * 2;
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
      const char* e = getenv("SEDT_IGEMM_BN128_MINK");
      mink = e ? atoi(e) : 512;
    }
    static int bn128t = -1;
    if (bn128t < 0) {
      const char* e = getenv("SEDT_IGEMM_BN128_MINTILES");
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
      const char* e = getenv("SEDT_IGEMM_BM128_MINK");
      bm128k = e ? atoi(e) : 2048;
      e = getenv("SEDT_IGEMM_BM128_MINTILES");
      bm128t = e ? atoi(e) : 256;
    }
    if (bn == 128 && p.K >= bm128k && (long)((p.M + 127) / 128) * (p.N / 128) >= bm128t && !igemm3_planning()) bm = 128;
  }
  {   // the lean-issue kernel takes the common cases
    int r3 = igemm3_try(p, (unsigned)a_bytes, (unsigned)b_bytes, bm, bn, st);
    if (r3 >= 0) return r3;
    if (igemm3_planning()) return -1;      // dry run (sedt_igemm_group): never launch from here
  }
  static int stages = -1;
  if (stages < 0) {
    const char* e = getenv("SEDT_IGEMM_STAGES");
    stages = e ? atoi(e) : 0;
  }
  if (bm == 128 && bn == 128) {
    if (stages == 3) return launch2<128, 128, 3>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    return launch2<128, 128, 2>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  }
  if (bm == 128 && bn == 64) {
    if (stages == 3) return launch2<128, 64, 3>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    return launch2<128, 64, 2>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  }
  if (bm == 64 && bn == 128) {
    return launch2<64, 128, 2>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  }
  if (bm == 64 && bn == 64) {
    if (stages == 4) return launch2<64, 64, 4>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    if (stages == 3) return launch2<64, 64, 3>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    return launch2<64, 64, 2>(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
  }
  return -1;
}

}  // namespace sedt
