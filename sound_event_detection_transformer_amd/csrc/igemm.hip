// igemm.hip - implicit GEMM on MFMA for gfx950 (MI355X).
//
// One kernel template serves every contraction on the SEDT hot path:
//   * nn.Linear / 1x1 conv forward and dgrad            (trans=0, conv=0)
//   * 3x3 / strided / dilated conv forward              (trans=0, conv=1)
//   * conv dgrad as a gather over output pixels         (trans=0, conv=1, transposed=1)
//   * wgrad of all of the above (reduction over pixels)  (trans=1, split-K slabs)
// with a fused epilogue (FrozenBN scale/bias, bias, ReLU/sigmoid, dropout, residual add,
// ReLU-mask of the consumer, alpha).  Activations are NHWC so a conv is a GEMM whose A rows
// are gathered pixels; weights arrive packed [N][taps][Ci] (see pack.hip).
//
// Tiling: 256 threads = 4 waves (2x2), workgroup tile BMxBN in {128,64}^2, each wave owns
// (BM/2)x(BN/2) as 32x32 MFMA tiles.  Operands are staged global -> VGPR -> LDS ([row][k],
// k contiguous, padded pitch) with the next tile's global loads issued before the current
// tile's MFMAs and written to the other LDS buffer afterwards: one barrier per K tile.
//   bf16: BK=64, pitch 72 (144 B: ds_read_b128 of 16 distinct rows hits 16 distinct 16-B slots),
//         v_mfma_f32_32x32x16_bf16, 8 k per lane fragment read as one ds_read_b128.
//   f32 : BK=32, pitch 33 (odd: 32 rows x fixed k conflict-free for ds_read_b32),
//         v_mfma_f32_32x32x2_f32 (exact f32 FMA chain - the parity mode).
//   x3  : (T = float, X3) the "bf16x3" mode: f32 tensors in memory, every f32 operand element split ONCE - when its tile is
//         staged - into hi = bf16(x) and lo = bf16(x - hi), kept as two bf16 LDS images ([row][32 k], pitch 40), and each
//         k16 block of a product computed as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with f32 accumulation: the
//         dropped lo.lo term and the rounding of lo bound the per-product error near 2^-16 relative (f32: 2^-24, plain bf16:
//         2^-8), at 3 bf16 MFMAs per 16 k instead of 8 f32 MFMAs - north_star's 1e-3 tolerance at MFMA rate.
// trans=1 sources have the reduction index as the slow axis; bf16 transposes 4(k)x8(r) blocks
// in registers before the LDS write so the fragment reads stay ds_read_b128.
#include <stdlib.h>
#include "common.h"

namespace sedt {

template <typename T> struct Cfg;
template <> struct Cfg<float> {
  static constexpr int BK = 32, PITCH = 33, EPC = 4, KSTEP = 2, UK = 1;
};
template <> struct Cfg<bf16_t> {
  static constexpr int BK = 64, PITCH = 72, EPC = 8, KSTEP = 16, UK = 4;
};

struct GatherGeom {
  int Hi, Wi, Ci, Ho, Wo, KH, KW, sh, sw, ph, pw, dh, dw, transposed;
};

// pixel index (in the Hi x Wi grid of image n) gathered for row-grid position (ho,wo) and tap (kh,kw); -1 = zero
__device__ __forceinline__ long gather_pix(const GatherGeom& g, int n, int ho, int wo, int kh, int kw) {
  int hi, wi;
  if (!g.transposed) {
    hi = ho * g.sh - g.ph + kh * g.dh;
    wi = wo * g.sw - g.pw + kw * g.dw;
    if ((unsigned)hi >= (unsigned)g.Hi || (unsigned)wi >= (unsigned)g.Wi) return -1;
  } else {
    int th = ho + g.ph - kh * g.dh, tw = wo + g.pw - kw * g.dw;
    if (th < 0 || tw < 0) return -1;
    hi = th / g.sh;
    wi = tw / g.sw;
    if (hi * g.sh != th || wi * g.sw != tw || hi >= g.Hi || wi >= g.Wi) return -1;
  }
  return ((long)n * g.Hi + hi) * g.Wi + wi;
}

template <typename T>
__device__ __forceinline__ uint4 load_chunk(const T* base, long off, int valid_elems, bool vec) {
  // loads EPC consecutive elements starting at base[off]; elements >= valid_elems read as 0
  constexpr int EPC = Cfg<T>::EPC;
  uint4 r = make_uint4(0, 0, 0, 0);
  if (valid_elems <= 0) return r;
  if (vec && valid_elems >= EPC) return *reinterpret_cast<const uint4*>(base + off);
  T tmp[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) tmp[e] = e < valid_elems ? base[off + e] : (T)0.f;
  r = *reinterpret_cast<uint4*>(tmp);
  return r;
}

constexpr int X3BK = 32;     // K block of the x3 mode (64 - 12 MFMAs per wave between two barriers, 74 KB of LDS, two workgroups per CU - measured
                             // slower: 29.3 against 27.3 ms per C2 step)
constexpr int X3P = X3BK + 8; // bf16 element pitch of the hi / lo images of the x3 mode (80-byte rows: 16-byte fragment reads of 16 rows conflict-free)

// hi / lo split of 4 consecutive f32 values -> 4 + 4 bf16
__device__ __forceinline__ void split4(const float* s, VecT<bf16_t, 4>& hi, VecT<bf16_t, 4>& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi.v[e] = (bf16_t)s[e];
    lo.v[e] = (bf16_t)(s[e] - (float)hi.v[e]);
  }
}

// FAST: every chunk is a whole, 16-byte aligned chunk inside the problem (M % BM == N % BN == K % BK == 0, aligned operands) - the loads
// are plain predicated 16-byte loads.  The general loader's per-chunk bounds / alignment branches made the staging code of an iteration
// several thousand instructions long (round 4: that, not the memory system and not the matrix pipe, bounded the f32 and bf16x3 modes).
template <typename T, int BM, int BN, bool TRANS, bool X3 = false, bool FAST = false>
__global__ __launch_bounds__(256) void igemm_kernel(const SedtIgemm p, const int vecA, const int vecB) {
  static_assert(!X3 || sizeof(T) == 4, "the x3 mode splits f32 operands");
  constexpr int BK = X3 ? X3BK : Cfg<T>::BK, PITCH = Cfg<T>::PITCH, EPC = Cfg<T>::EPC, KSTEP = Cfg<T>::KSTEP;
  constexpr int WM = BM / 2, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* sA = reinterpret_cast<T*>(smem_raw);  // [2][BM*PITCH]
  T* sB = sA + 2 * BM * PITCH;             // [2][BN*PITCH]
  // x3: per buffer and operand a hi image and a lo image [rows][X3P] bf16
  bf16_t* xA = reinterpret_cast<bf16_t*>(smem_raw);          // [2][2][BM * X3P]
  bf16_t* xB = xA + 2 * 2 * BM * X3P;                        // [2][2][BN * X3P]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const T* __restrict__ Ag = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ Bg = reinterpret_cast<const T*>(p.B);

  const int nkb_total = (p.K + BK - 1) / BK;
  int kb_begin = 0, kb_end = nkb_total;
  if (p.splitk > 1) {
    int per = (nkb_total + p.splitk - 1) / p.splitk;
    kb_begin = blockIdx.z * per;
    kb_end = min(nkb_total, kb_begin + per);
  }
  const int nkb = max(0, kb_end - kb_begin);

  GatherGeom g{p.Hi, p.Wi, p.Ci, p.Ho, p.Wo, p.KH, p.KW, p.sh, p.sw, p.ph, p.pw, p.dh, p.dw, p.transposed};
  const int HoWo = p.conv ? p.Ho * p.Wo : 1;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ======================================================================== staging
  // ---- trans == 0: rows are output rows, k contiguous in memory
  constexpr int CPR = BK / EPC;  // 16-byte chunks per tile row (8)
  constexpr int RPP = 256 / CPR; // rows per pass (32)
  constexpr int APASS = TRANS ? 1 : BM / RPP, BPASS = TRANS ? 1 : BN / RPP;
  // ---- trans == 1: unit = UK k-rows x EPC r-elements
  constexpr int UK = Cfg<T>::UK;
  constexpr int AU = (BK / UK) * (BM / EPC), BU = (BK / UK) * (BN / EPC);
  constexpr int TAP = TRANS ? (AU + 255) / 256 : 1, TBP = TRANS ? (BU + 255) / 256 : 1;
  constexpr int NRA = TRANS ? TAP * UK : APASS, NRB = TRANS ? TBP * UK : BPASS;
  // TWO register stages: tile it + 2 is loaded while tile it is multiplied and tile it + 1 waits in the other set - a global load has two
  // iterations to land instead of one (round 4: with one stage every iteration exposed most of a memory round trip; the f32 / bf16x3 steps
  // spend 90 % of their time in this kernel)
  uint4 ra0[NRA], rb0[NRB], ra1[NRA], rb1[NRB];

  // per-thread row state (trans == 0)
  const int chunk = t % CPR, lrow = t / CPR;
  int a_n[APASS], a_ho[APASS], a_wo[APASS];
  bool a_ok[APASS];
  if constexpr (!TRANS) {
#pragma unroll
    for (int ps = 0; ps < APASS; ++ps) {
      int row = m0 + lrow + ps * RPP;
      a_ok[ps] = row < p.M;
      if (p.conv) {
        int n = row / HoWo, rem = row - n * HoWo;
        a_n[ps] = n;
        a_ho[ps] = rem / p.Wo;
        a_wo[ps] = rem - a_ho[ps] * p.Wo;
      } else {
        a_n[ps] = row; a_ho[ps] = 0; a_wo[ps] = 0;
      }
    }
  }
  // per-thread column state for the gathered operand of trans == 1
  int b_tap_kh[TBP], b_tap_kw[TBP], b_c[TBP];
  if constexpr (TRANS) {
#pragma unroll
    for (int ps = 0; ps < TBP; ++ps) {
      int u = t + ps * 256;
      int ju = u % (BN / EPC);
      int j = n0 + ju * EPC;
      if (p.conv) {
        int tap = j / p.Ci;
        b_c[ps] = j - tap * p.Ci;
        b_tap_kh[ps] = tap / p.KW;
        b_tap_kw[ps] = tap - b_tap_kh[ps] * p.KW;
      } else {
        b_c[ps] = j; b_tap_kh[ps] = 0; b_tap_kw[ps] = 0;
      }
    }
  }

  // trans == 0: element offsets of this thread's A rows for the tap the K walk is in (-1: padding) and of its B rows - recomputed when the
  // walk enters a new tap, not per iteration (the gather arithmetic, integer divisions included, was most of an iteration's instructions)
  long a_off[APASS], b_off[BPASS];
  int w_tap = -1, w_c0 = 0, w_k0 = kb_begin * BK;               // the K walk: load_tiles is called for kb_begin, kb_begin + 1, ... in order
  if constexpr (!TRANS) {
    if (p.conv) { w_tap = w_k0 / p.Ci; w_c0 = w_k0 - w_tap * p.Ci; w_tap = -1 - w_tap; }     // (negative: offsets not computed yet)
    else w_c0 = w_k0;
#pragma unroll
    for (int ps = 0; ps < BPASS; ++ps) b_off[ps] = (long)(n0 + lrow + ps * RPP) * p.ldb + chunk * EPC;
    if (!p.conv) {
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) a_off[ps] = a_ok[ps] ? (long)a_n[ps] * p.lda + chunk * EPC : -1;
    }
  }
  auto ld = [&](const T* base, long off, int valid, int vec) -> uint4 {
    if constexpr (FAST) return *reinterpret_cast<const uint4*>(base + off);
    else return load_chunk<T>(base, off, valid, vec);
  };
  auto load_tiles = [&](int kb, uint4(&ra)[NRA], uint4(&rb)[NRB]) {
    const int k0 = kb * BK;
    if constexpr (!TRANS) {
      // A: gathered rows
      if (p.conv && w_tap < 0) {                                  // a new tap: this thread's row offsets
        w_tap = -1 - w_tap;
        const int kh = w_tap / p.KW, kw = w_tap - kh * p.KW;
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
          const long pix = a_ok[ps] ? gather_pix(g, a_n[ps], a_ho[ps], a_wo[ps], kh, kw) : -1;
          a_off[ps] = pix >= 0 ? pix * p.lda + chunk * EPC : -1;
        }
      }
      const int kvalid = p.K - (k0 + chunk * EPC);  // elements of this chunk inside K
#pragma unroll
      for (int ps = 0; ps < APASS; ++ps) ra[ps] = (a_off[ps] >= 0) ? ld(Ag, a_off[ps] + w_c0, kvalid, vecA) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int ps = 0; ps < BPASS; ++ps) {
        int row = n0 + lrow + ps * RPP;
        rb[ps] = (FAST || row < p.N) ? ld(Bg, b_off[ps] + k0, kvalid, vecB) : make_uint4(0, 0, 0, 0);
      }
      w_c0 += BK;                                                 // (a K block never straddles a tap: Ci % BK == 0)
      if (p.conv && w_c0 >= p.Ci) { w_c0 = 0; w_tap = -2 - w_tap; }
    } else {
      // A: dY[k][m], plain
#pragma unroll
      for (int ps = 0; ps < TAP; ++ps) {
        int u = t + ps * 256;
        int ku = u / (BM / EPC), iu = u % (BM / EPC);
        int i = m0 + iu * EPC;
#pragma unroll
        for (int kk = 0; kk < UK; ++kk) {
          int k = k0 + ku * UK + kk;
          bool ok = (u < AU) && (FAST || k < p.K);
          ra[ps * UK + kk] = ok ? ld(Ag, (long)k * p.lda + i, p.M - i, vecA) : make_uint4(0, 0, 0, 0);
        }
      }
      // B: X gathered by pixel k and this thread's tap
#pragma unroll
      for (int ps = 0; ps < TBP; ++ps) {
        int u = t + ps * 256;
        int ku = u / (BN / EPC), ju = u % (BN / EPC);
        int j = n0 + ju * EPC;
#pragma unroll
        for (int kk = 0; kk < UK; ++kk) {
          int k = k0 + ku * UK + kk;
          bool ok = (u < BU) && (FAST || ((k < p.K) && (j < p.N)));
          long pix = -1;
          if (ok) {
            if (p.conv) {
              int n = k / HoWo, rem = k - n * HoWo;
              int ho = rem / p.Wo, wo = rem - ho * p.Wo;
              pix = gather_pix(g, n, ho, wo, b_tap_kh[ps], b_tap_kw[ps]);
            } else {
              pix = k;
            }
          }
          // a chunk never straddles a tap (Ci % EPC == 0 is required when conv); plain: clip at N
          int valid = p.conv ? EPC : p.N - j;
          rb[ps * UK + kk] = (pix >= 0) ? ld(Bg, pix * p.ldb + b_c[ps], valid, vecB) : make_uint4(0, 0, 0, 0);
        }
      }
    }
  };

  auto store_tiles = [&](int buf, const uint4(&ra)[NRA], const uint4(&rb)[NRB]) {
    T* dA = sA + buf * BM * PITCH;
    T* dB = sB + buf * BN * PITCH;
    if constexpr (X3) {
      bf16_t* hA = xA + buf * 2 * BM * X3P;
      bf16_t* lA = hA + BM * X3P;
      bf16_t* hB = xB + buf * 2 * BN * X3P;
      bf16_t* lB = hB + BN * X3P;
      if constexpr (!TRANS) {
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
          VecT<bf16_t, 4> hi, lo;
          split4(reinterpret_cast<const float*>(&ra[ps]), hi, lo);
          *reinterpret_cast<VecT<bf16_t, 4>*>(hA + (lrow + ps * RPP) * X3P + chunk * EPC) = hi;
          *reinterpret_cast<VecT<bf16_t, 4>*>(lA + (lrow + ps * RPP) * X3P + chunk * EPC) = lo;
        }
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps) {
          VecT<bf16_t, 4> hi, lo;
          split4(reinterpret_cast<const float*>(&rb[ps]), hi, lo);
          *reinterpret_cast<VecT<bf16_t, 4>*>(hB + (lrow + ps * RPP) * X3P + chunk * EPC) = hi;
          *reinterpret_cast<VecT<bf16_t, 4>*>(lB + (lrow + ps * RPP) * X3P + chunk * EPC) = lo;
        }
      } else {
        // registers hold 1 k-row x 4 r: element (r, k) of the [r][k] images
#pragma unroll
        for (int ps = 0; ps < TAP; ++ps) {
          const int u = t + ps * 256;
          if (u < AU) {
            const int ku = u / (BM / EPC), iu = u % (BM / EPC);
            VecT<bf16_t, 4> hi, lo;
            split4(reinterpret_cast<const float*>(&ra[ps]), hi, lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) { hA[(iu * 4 + e) * X3P + ku] = hi.v[e]; lA[(iu * 4 + e) * X3P + ku] = lo.v[e]; }
          }
        }
#pragma unroll
        for (int ps = 0; ps < TBP; ++ps) {
          const int u = t + ps * 256;
          if (u < BU) {
            const int ku = u / (BN / EPC), ju = u % (BN / EPC);
            VecT<bf16_t, 4> hi, lo;
            split4(reinterpret_cast<const float*>(&rb[ps]), hi, lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) { hB[(ju * 4 + e) * X3P + ku] = hi.v[e]; lB[(ju * 4 + e) * X3P + ku] = lo.v[e]; }
          }
        }
      }
      return;
    }
    if constexpr (!TRANS) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps)
          *reinterpret_cast<uint4*>(dA + (lrow + ps * RPP) * PITCH + chunk * EPC) = ra[ps];
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps)
          *reinterpret_cast<uint4*>(dB + (lrow + ps * RPP) * PITCH + chunk * EPC) = rb[ps];
      } else {
#pragma unroll
        for (int ps = 0; ps < APASS; ++ps) {
          float* d = reinterpret_cast<float*>(dA) + (lrow + ps * RPP) * PITCH + chunk * EPC;
          const float* s = reinterpret_cast<const float*>(&ra[ps]);
          d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
        }
#pragma unroll
        for (int ps = 0; ps < BPASS; ++ps) {
          float* d = reinterpret_cast<float*>(dB) + (lrow + ps * RPP) * PITCH + chunk * EPC;
          const float* s = reinterpret_cast<const float*>(&rb[ps]);
          d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
        }
      }
    } else {
      if constexpr (sizeof(T) == 2) {
        // registers hold 4 k-rows x 8 r (bf16 pairs per dword); emit for each r the 4 k values (8 bytes)
        auto tr_store = [&](const uint4* r4, T* dst, int ru, int ku) {
          const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&r4[0]);
          const uint32_t* w1 = reinterpret_cast<const uint32_t*>(&r4[1]);
          const uint32_t* w2 = reinterpret_cast<const uint32_t*>(&r4[2]);
          const uint32_t* w3 = reinterpret_cast<const uint32_t*>(&r4[3]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            uint2 ev, od;
            ev.x = (w0[q] & 0xffffu) | (w1[q] << 16);
            ev.y = (w2[q] & 0xffffu) | (w3[q] << 16);
            od.x = (w0[q] >> 16) | (w1[q] & 0xffff0000u);
            od.y = (w2[q] >> 16) | (w3[q] & 0xffff0000u);
            *reinterpret_cast<uint2*>(dst + (ru * 8 + 2 * q) * PITCH + ku * 4) = ev;
            *reinterpret_cast<uint2*>(dst + (ru * 8 + 2 * q + 1) * PITCH + ku * 4) = od;
          }
        };
#pragma unroll
        for (int ps = 0; ps < TAP; ++ps) {
          int u = t + ps * 256;
          if (u < AU) tr_store(&ra[ps * UK], dA, u % (BM / EPC), u / (BM / EPC));
        }
#pragma unroll
        for (int ps = 0; ps < TBP; ++ps) {
          int u = t + ps * 256;
          if (u < BU) tr_store(&rb[ps * UK], dB, u % (BN / EPC), u / (BN / EPC));
        }
      } else {
#pragma unroll
        for (int ps = 0; ps < TAP; ++ps) {
          int u = t + ps * 256;
          if (u < AU) {
            int ku = u / (BM / EPC), iu = u % (BM / EPC);
            float* d = reinterpret_cast<float*>(dA) + (iu * 4) * PITCH + ku;
            const float* s = reinterpret_cast<const float*>(&ra[ps]);
            d[0] = s[0]; d[PITCH] = s[1]; d[2 * PITCH] = s[2]; d[3 * PITCH] = s[3];
          }
        }
#pragma unroll
        for (int ps = 0; ps < TBP; ++ps) {
          int u = t + ps * 256;
          if (u < BU) {
            int ku = u / (BN / EPC), ju = u % (BN / EPC);
            float* d = reinterpret_cast<float*>(dB) + (ju * 4) * PITCH + ku;
            const float* s = reinterpret_cast<const float*>(&rb[ps]);
            d[0] = s[0]; d[PITCH] = s[1]; d[2 * PITCH] = s[2]; d[3 * PITCH] = s[3];
          }
        }
      }
    }
  };

  auto compute = [&](int buf) {
    if constexpr (X3) {
      const bf16_t* hA = xA + buf * 2 * BM * X3P + (wm + (lane & 31)) * X3P + 8 * (lane >> 5);
      const bf16_t* lA = hA + BM * X3P;
      const bf16_t* hB = xB + buf * 2 * BN * X3P + (wn + (lane & 31)) * X3P + 8 * (lane >> 5);
      const bf16_t* lB = hB + BN * X3P;
#pragma unroll
      for (int kk = 0; kk < BK; kk += 16) {
        bf16x8 ah[MI], al[MI], bh[NI], bl[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8*>(hA + i * 32 * X3P + kk);
          al[i] = *reinterpret_cast<const bf16x8*>(lA + i * 32 * X3P + kk);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8*>(hB + j * 32 * X3P + kk);
          bl[j] = *reinterpret_cast<const bf16x8*>(lB + j * 32 * X3P + kk);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            // small terms first: their sum is formed before it meets the large one
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      return;
    }
    const T* cA = sA + buf * BM * PITCH + (wm + (lane & 31)) * PITCH;
    const T* cB = sB + buf * BN * PITCH + (wn + (lane & 31)) * PITCH;
    if constexpr (sizeof(T) == 2) {
      const int kofs = 8 * (lane >> 5);
#pragma unroll
      for (int kk = 0; kk < BK; kk += KSTEP) {
        bf16x8 a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = *reinterpret_cast<const bf16x8*>(cA + i * 32 * PITCH + kk + kofs);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = *reinterpret_cast<const bf16x8*>(cB + j * 32 * PITCH + kk + kofs);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int kofs = lane >> 5;
#pragma unroll 4
      for (int kk = 0; kk < BK; kk += KSTEP) {
        float a[MI], b[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a[i] = cA[i * 32 * PITCH + kk + kofs];
#pragma unroll
        for (int j = 0; j < NI; ++j) b[j] = cB[j * 32 * PITCH + kk + kofs];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ======================================================================== main loop
  if (nkb > 0) {
    load_tiles(kb_begin, ra0, rb0);
    if (nkb > 1) load_tiles(kb_begin + 1, ra1, rb1);
    store_tiles(0, ra0, rb0);
    __syncthreads();
    // tile `it` is in LDS buffer it & 1, tile it + 1 in (or on its way to) the register set `nxt`, the set `fre` is free
    auto body = [&](int it, uint4(&fa)[NRA], uint4(&fb)[NRB], const uint4(&na)[NRA], const uint4(&nb)[NRB]) {
      if (it + 2 < nkb) load_tiles(kb_begin + it + 2, fa, fb);
      compute(it & 1);
      if (it + 1 < nkb) store_tiles((it & 1) ^ 1, na, nb);
      __syncthreads();
    };
    for (int it = 0; it < nkb; it += 2) {
      body(it, ra0, rb0, ra1, rb1);
      if (it + 1 < nkb) body(it + 1, ra1, rb1, ra0, rb0);
    }
  }

  // ======================================================================== epilogue
  const bool slab = p.splitk > 1;
  const uint32_t seed = eff_seed(p.seed, p.seed_ptr);
  const uint32_t thresh = drop_threshold(p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const T* resT = reinterpret_cast<const T*>(p.res);
  const T* maskT = p.mask_bits ? nullptr : reinterpret_cast<const T*>(p.mask);
  const uint8_t* maskB = p.mask_bits ? reinterpret_cast<const uint8_t*>(p.mask) : nullptr;
  // one 32x32 accumulator tile; called with compile-time (i, j) so acc stays in registers
  auto epilogue_tile = [&](const f32x16& a, const int i, const int j) {
    const int col = n0 + wn + j * 32 + (lane & 31);
    const bool colok = col < p.N;
    float sc = 1.f, bi = 0.f;
    if (!slab && colok) {
      if (p.scale) sc = p.scale[col];
      if (p.bias) bi = p.bias[col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (p.bits_out && !slab) {
        // (uniform branch) every lane takes part in the ballot below: lanes outside the matrix contribute a 0 bit
        const bool ok = row < p.M && colok;
        float v = 0.f;
        if (ok) {
          v = a[r] * sc + bi;
          if (!p.act_post_res) {
            if (p.act == SEDT_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == SEDT_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          }
          if (p.drop_p > 0.f) v = drop_keep(seed, (uint64_t)row * (uint64_t)p.N + col, thresh) ? v * inv_keep : 0.f;
          if (resT) {
            long rr = p.res_mod > 0 ? (row % p.res_mod) : row;
            v += (float)resT[rr * p.ldr + col];
          }
          if (p.act_post_res) {
            if (p.act == SEDT_ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == SEDT_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
          }
          if (maskT) v = ((float)maskT[(long)row * p.ldm + col] > 0.f) ? v : 0.f;
          if (maskB) v = ((maskB[(long)row * p.ldm + (col >> 3)] >> (col & 7)) & 1) ? v : 0.f;
          v *= p.alpha;
          if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
          else { const T tv = (T)v; reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = tv; v = (float)tv; }
        }
        const unsigned long long pos = __ballot(ok && v > 0.f);       // lanes 0..31: this row's 32 columns, 32..63: row + 4
        if (ok && (lane & 7) == 0) p.bits_out[(long)row * p.ldbits + (col >> 3)] = (uint8_t)(pos >> (lane & ~7));
        continue;
      }
      if (!(row < p.M && colok)) continue;
      float v = a[r];
      if (slab) {
        p.slab[((long)blockIdx.z * p.M + row) * p.N + col] = v;
        continue;
      }
      v = v * sc + bi;
      if (!p.act_post_res) {
        if (p.act == SEDT_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == SEDT_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
      }
      if (p.drop_p > 0.f) v = drop_keep(seed, (uint64_t)row * (uint64_t)p.N + col, thresh) ? v * inv_keep : 0.f;
      if (resT) {
        long rr = p.res_mod > 0 ? (row % p.res_mod) : row;
        v += (float)resT[rr * p.ldr + col];
      }
      if (p.act_post_res) {
        if (p.act == SEDT_ACT_RELU) v = fmaxf(v, 0.f);
        else if (p.act == SEDT_ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
      }
      if (maskT) v = ((float)maskT[(long)row * p.ldm + col] > 0.f) ? v : 0.f;
      if (maskB) v = ((maskB[(long)row * p.ldm + (col >> 3)] >> (col & 7)) & 1) ? v : 0.f;
      v *= p.alpha;
      if (p.out_f32) reinterpret_cast<float*>(p.C)[(long)row * p.ldc + col] = v;
      else reinterpret_cast<T*>(p.C)[(long)row * p.ldc + col] = (T)v;
    }
  };
  epilogue_tile(acc[0][0], 0, 0);
  if constexpr (NI > 1) epilogue_tile(acc[0][1], 0, 1);
  if constexpr (MI > 1) epilogue_tile(acc[1][0], 1, 0);
  if constexpr (MI > 1 && NI > 1) epilogue_tile(acc[1][1], 1, 1);
}

// ---------------------------------------------------------------------------- host side
template <typename T, int BM, int BN, bool TRANS, bool X3 = false, bool FAST = false>
static int launch_cfg(const SedtIgemm& p, int vecA, int vecB, hipStream_t st) {
  if constexpr (!FAST && sizeof(T) == 4) {
    // whole, aligned tiles: the branch-free loader (f32 / bf16x3 only: the bf16 mode lives in the LDS-DMA family)
    if (vecA && vecB && p.M % BM == 0 && p.N % BN == 0 && p.K % (X3 ? X3BK : Cfg<T>::BK) == 0 && (!p.conv || p.trans || p.Ci % (X3 ? X3BK : Cfg<T>::BK) == 0))
      return launch_cfg<T, BM, BN, TRANS, X3, true>(p, vecA, vecB, st);
  }
  SEDT_DESCRIBE("igemm_kernel<%s, %d, %d, %s%s%s>", sizeof(T) == 4 ? "float" : "__bf16", BM, BN, TRANS ? "true" : "false", X3 ? ", true" : "",
                FAST ? (X3 ? ", true" : ", false, true") : "");
  constexpr size_t lds = X3 ? (size_t)(BM + BN) * X3P * sizeof(bf16_t) * 2 * 2 : (size_t)(BM + BN) * Cfg<T>::PITCH * sizeof(T) * 2;
  static bool attr_set = false;  // idempotent; a benign race sets it twice
  auto kern = igemm_kernel<T, BM, BN, TRANS, X3, FAST>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      set_error("igemm: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splitk > 1 ? p.splitk : 1);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, vecA, vecB);
  return check_launch("igemm");
}

template <typename T, bool X3 = false>
static int launch_typed(const SedtIgemm& p, hipStream_t st) {
  constexpr int EPC = Cfg<T>::EPC;
  const size_t es = sizeof(T);
  auto aligned = [&](const void* ptr, int64_t ld) {
    return ((reinterpret_cast<uintptr_t>(ptr) % 16) == 0) && ((ld * (int64_t)es) % 16 == 0);
  };
  int vecA = aligned(p.A, p.lda), vecB = aligned(p.B, p.ldb);
  if (p.conv) {
    SEDT_REQUIRE(p.Ci % EPC == 0, "igemm: conv needs Ci %% %d == 0 (got %d)", EPC, p.Ci);
    if (!p.trans) SEDT_REQUIRE(p.Ci % (X3 ? X3BK : Cfg<T>::BK) == 0, "igemm: conv needs Ci %% BK == 0 (Ci=%d, BK=%d)", p.Ci, X3 ? X3BK : Cfg<T>::BK);
  }
  if (!p.trans) {
    // non-trans vector chunks run along k: k offsets are multiples of EPC, so K must be too (else scalar path)
    if (p.K % EPC != 0) { vecA = 0; vecB = 0; }
  }
  int bm = p.tile_m, bn = p.tile_n;
  if (bm == 0 || bn == 0) {
    // measured on MI355X (tools/tune_igemm.py, profiles/): this register-staged pipeline is latency-bound, so the
    // 64x64 tile (4 workgroups / CU resident) beats the larger tiles at every shape of the SEDT step
    bm = 64; bn = 64;
    // (larger tiles - 24 instead of 6 MFMAs per wave between two barriers - measured SLOWER, before and after the loader clean-up of
    // round 4 (tools/dev/ab_gen_tile.sh, C2 step: 64x64 everywhere 26.1 ms bf16x3 / 28.1 ms f32; 128x64 above 512 tiles 29.9 / 30.2;
    // 128x128 above 512 tiles 31.8 / 30.0; above 256 tiles 34.3-34.5 / 31.2-31.9): what the pipeline needs is resident workgroups)
    static int gen_tile = [] { const char* e = dev_getenv("SEDT_GEN_TILE"); return e ? atoi(e) : 0; }();      // developer A/B: 128064 / 128128
    static int gen_min = [] { const char* e = dev_getenv("SEDT_GEN_TILE_MIN"); return e ? atoi(e) : 512; }();
    if (gen_tile && sizeof(T) == 4) {
      const int tm = gen_tile / 1000 >= 128 ? 128 : 64, tn = gen_tile % 1000 >= 128 ? 128 : 64;
      if (p.M % tm == 0 && p.N % tn == 0 && (long)(p.M / tm) * (p.N / tn) * (p.splitk > 1 ? p.splitk : 1) >= gen_min) { bm = tm; bn = tn; }
    }
  }
  if (p.splitk > 1) SEDT_REQUIRE(p.slab != nullptr, "igemm: splitk > 1 needs a slab");
#define SEDT_DISPATCH(BM_, BN_)                                                         \
  if (bm == BM_ && bn == BN_) {                                                         \
    return p.trans ? launch_cfg<T, BM_, BN_, true, X3>(p, vecA, vecB, st)               \
                   : launch_cfg<T, BM_, BN_, false, X3>(p, vecA, vecB, st);             \
  }
  SEDT_DISPATCH(128, 128)
  SEDT_DISPATCH(128, 64)
  SEDT_DISPATCH(64, 64)
#undef SEDT_DISPATCH
  set_error("igemm: unsupported tile %dx%d", bm, bn);
  return 1;
}

}  // namespace sedt

namespace sedt { int igemm_lds_try(const SedtIgemm& p, hipStream_t st); }   // igemm3.hip: the LDS-DMA GEMM family's dispatcher

static bool use_lds_family() {
  static int v = -1;
  if (v < 0) {
    const char* e = sedt::dev_getenv("SEDT_IGEMM_LDS");       // developer A/B switch: 0 = the general kernel of this file for everything
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

extern "C" int sedt_igemm(const SedtIgemm* args, int dtype, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr, "igemm: null args");
  SEDT_REQUIRE(args->M > 0 && args->N > 0 && args->K > 0, "igemm: bad dims M=%d N=%d K=%d", args->M, args->N, args->K);
  SEDT_REQUIRE(args->A && args->B && (args->C || args->splitk > 1), "igemm: null operand");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype != SEDT_BF16)
    SEDT_REQUIRE(!args->omap && !args->btap_on && !args->f32ep && !args->awrap, "igemm: omap / btap / f32ep / awrap are features of the bf16 LDS-DMA kernels");
  if (dtype == SEDT_F32) {
    SEDT_REQUIRE(args->colsum_out == nullptr, "igemm: colsum_out is a bf16-only feature");
    return launch_typed<float>(*args, st);
  }
  if (dtype == SEDT_BF16X3) {           // f32 tensors, split-bf16 products (see the header of this file)
    SEDT_REQUIRE(args->colsum_out == nullptr, "igemm: colsum_out is a bf16-only feature");
    return launch_typed<float, true>(*args, st);
  }
  if (dtype == SEDT_BF16) {
    if (use_lds_family()) {               // LDS-DMA kernels (igemm3 / wgrad3 / wgrad4) when the problem fits their envelope
      int r = igemm_lds_try(*args, st);
      if (r >= 0) return r;
    }
    SEDT_REQUIRE(args->colsum_out == nullptr, "igemm: colsum_out needs the bf16 LDS-DMA wgrad kernel (M,N,lda,ldb %% 8 == 0)");
    SEDT_REQUIRE(!args->omap && !args->btap_on && !args->f32ep && !args->awrap,
                 "igemm: omap / btap / f32ep / awrap problem outside the LDS-DMA envelope (M %d N %d K %d)", args->M, args->N, args->K);
    return launch_typed<bf16_t>(*args, st);
  }
  set_error("igemm: unsupported dtype %d", dtype);
  return 1;
}

namespace sedt {
thread_local Describe describe = {false, {0}};
bool wgrad4_ok(const SedtIgemm& p);
int wgrad_lds_envelope(const SedtIgemm& p, long* a_bytes, long* b_bytes);
}

// name of the kernel instance sedt_igemm (grouped == 0), sedt_igemm_group (1: forward / dgrad) or sedt_wgrad_group (1: trans) runs
// the problem on, as rocprofv3 / the torch profiler print it; nothing is launched.  A measurement aid (bench.py roofline.families).
extern "C" int sedt_igemm_describe(const SedtIgemm* args, int dtype, int grouped, char* out, int cap) {
  using namespace sedt;
  SEDT_REQUIRE(args && out && cap > 0, "igemm_describe: bad arguments");
  out[0] = 0;
  if (grouped && dtype == SEDT_BF16 && args->trans) {       // sedt_wgrad_group: the 256/128x128 group or the 64x64 group
    long a, b;
    if (wgrad_lds_envelope(*args, &a, &b) == 0) {
      snprintf(out, cap, "%s", wgrad4_ok(*args) ? "wgrad4_group_kernel" : "wgrad3_group_kernel");
      return 0;
    }
  }
  describe.on = true;
  describe.name[0] = 0;
  const int r = sedt_igemm(args, dtype, nullptr);
  describe.on = false;
  if (r == 0) snprintf(out, cap, "%s", describe.name);
  return r;
}

namespace sedt {
int wgrad3_group_try(const SedtIgemm* jobs, int njobs, hipStream_t st);
bool wgrad4_shape_ok(int M, int N);
int wgrad4_tile_m(int M, int N);
int igemm3_group_try(const SedtIgemm* jobs, int njobs, hipStream_t st);
}

extern "C" int sedt_igemm_group(const SedtIgemm* jobs, int njobs, int dtype, void* stream) {
  SEDT_REQUIRE(jobs && njobs >= 1, "igemm_group: bad arguments");
  if (dtype == SEDT_BF16 && njobs >= 2) {
    const int r = sedt::igemm3_group_try(jobs, njobs, reinterpret_cast<hipStream_t>(stream));
    if (r >= 0) return r;
  }
  for (int i = 0; i < njobs; ++i)
    if (int r = sedt_igemm(&jobs[i], dtype, stream)) return r;
  return 0;
}

// the kernel instance sedt_igemm_group would run the problems on ("" when they fall back to one launch each); nothing is launched
extern "C" int sedt_igemm_group_describe(const SedtIgemm* jobs, int njobs, int dtype, char* out, int cap) {
  using namespace sedt;
  SEDT_REQUIRE(jobs && njobs >= 1 && out && cap > 0, "igemm_group_describe: bad arguments");
  out[0] = 0;
  if (dtype != SEDT_BF16 || njobs < 2) return 0;
  describe.on = true;
  describe.name[0] = 0;
  const int r = igemm3_group_try(jobs, njobs, nullptr);
  describe.on = false;
  if (r == 0) snprintf(out, cap, "%s", describe.name);
  return r > 0 ? r : 0;
}

extern "C" int sedt_wgrad_group(const SedtIgemm* jobs, int njobs, int dtype, void* stream) {
  SEDT_REQUIRE(jobs && njobs >= 1, "wgrad_group: bad arguments");
  if (dtype == SEDT_BF16) {
    const int r = sedt::wgrad3_group_try(jobs, njobs, reinterpret_cast<hipStream_t>(stream));
    if (r >= 0) return r;
  }
  for (int i = 0; i < njobs; ++i)          // problems the grouped kernel cannot take (f32 mode, odd shapes): one by one
    if (int r = sedt_igemm(&jobs[i], dtype, stream)) return r;
  return 0;
}

// forward / dgrad GEMM `main` with up to 10 weight-gradient problems riding in the same launch (igemm3_co_kernel).  Falls
// back to separate launches whenever either side is outside its fast kernel's envelope - the results are the same.
#include "wgrad3_body.h"
namespace sedt {
int wgrad3_group_build(const SedtIgemm* jobs, int njobs, WgradGroup* g);
extern thread_local const WgradGroup* co_group;
extern thread_local bool co_taken;
}

extern "C" int sedt_igemm_co(const SedtIgemm* main, const SedtIgemm* wjobs, int nw, int dtype, void* stream, int* taken) {
  using namespace sedt;
  SEDT_REQUIRE(main && (nw == 0 || wjobs) && taken, "igemm_co: bad arguments");
  *taken = 0;
  if (nw == 0) return sedt_igemm(main, dtype, stream);
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_COSCHEDULE");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  WgradGroup g;
  bool grouped = on && dtype == SEDT_BF16 && main->trans == 0 && wgrad3_group_build(wjobs, nw, &g) == 0;
  co_taken = false;
  co_group = grouped ? &g : nullptr;
  const int r = sedt_igemm(main, dtype, stream);
  co_group = nullptr;
  *taken = co_taken ? 1 : 0;          // 0: the main GEMM ran on a kernel that cannot carry riders; the caller keeps them
  return r;
}

extern "C" int sedt_igemm_splitk(int M, int N, int K, int dtype) {
  // wgrad outputs are small (Cout x taps*Cin) while K (pixels) is long: split K until ~TARGET workgroups of 64x64,
  // keeping at least 4 K tiles per slice.  Every split costs a 64x64 f32 partial tile written and re-read, so the target
  // balances occupancy against slab traffic (measured on the full step: SEDT_SPLITK_TARGET sweep)
  static int target_env = -2;
  if (target_env == -2) {
    const char* e = sedt::dev_getenv("SEDT_SPLITK_TARGET");
    target_env = e ? atoi(e) : -1;
  }
  // bf16: wgrads of a layer are issued as one grouped launch, so the union of their tiles fills the chip and each problem
  // needs less splitting (re-tuned: 192 -> 9407, 384 -> 9489, 512 -> 9457, 768 -> 9356, 1024 -> 9250 clips/s);
  // f32 parity mode launches them one by one on the register-staged kernel and keeps the larger target
  const int target = target_env > 0 ? target_env : (dtype == SEDT_BF16 ? 384 : 768);
  int bk = dtype == SEDT_BF16 ? 64 : 32;
  long tiles = (long)((M + 63) / 64) * ((N + 63) / 64);
  int nkb = (K + bk - 1) / bk;
  // the large problems run on 256x128 / 128x128 tiles (wgrad4.hip), one 8-wave workgroup per CU
  int tgt = target;
  if (dtype == SEDT_BF16 && sedt::wgrad4_shape_ok(M, N)) {
    static int wide_env = -2;
    if (wide_env == -2) {
      const char* e = sedt::dev_getenv("SEDT_SPLITK_TARGET_WIDE");
      wide_env = e ? atoi(e) : -1;
    }
    tiles = (long)(M / sedt::wgrad4_tile_m(M, N)) * (N / 128);
    tgt = wide_env > 0 ? wide_env : (sedt::wgrad4_tile_m(M, N) == 256 ? 64 : 128);
  }
  int s = (int)((tgt + tiles - 1) / tiles);
  int maxs = nkb / 4 > 1 ? nkb / 4 : 1;
  if (s > maxs) s = maxs;
  if (s > 128) s = 128;
  // one or two output tiles under a very long K (the stem: 64 x 128 over 512 k pixels) is a pure HBM stream: only many
  // resident workgroups hide its latency, and its slabs are tiny
  if (tiles <= 2) s = maxs < 512 ? maxs : 512;
  if (s < 1) s = 1;
  // K-slice mapping (bf16 grouped launches, csrc/wgrad3_body.h): with a split factor that is a multiple of 8, XCD x works
  // on K slice x of EVERY output tile, so each private 4 MB L2 fetches 1/8 of both operands instead of one operand in
  // full (PMC: the wgrad launches fetched 8x their algorithmic bytes).  The price is more f32 slabs; take it when the
  // estimated fabric traffic drops by a fifth: operands A = K*M*2 B (dY), B = K*Cin*2 B (X; 3x3 kernels re-use a pixel for
  // 9 taps), slabs 2 * s * M*N*4 B (written, then read by the reduction).
  static int kslice = -1;
  if (kslice < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD_KSLICE");
    kslice = (e && e[0] == '1') ? 1 : 0;      // opt-in: measured -33 % fetched bytes for the wgrad launches, same run time
  }
  if (kslice && dtype == SEDT_BF16) {
    const int taps = (N % 9 == 0 && (N / 9) % 64 == 0) ? 9 : 1;
    const double A = 2.0 * K * M, B = 2.0 * K * (N / taps);
    const double cur = (N > M ? 8.0 * A + B : A + 8.0 * B) + 2.0 * s * 4.0 * M * N;
    int s8 = ((s + 7) / 8) * 8;
    if (s8 < 8) s8 = 8;
    if (s8 <= maxs && s8 <= 128) {
      const double ks = A + B + 2.0 * s8 * 4.0 * M * N;
      if (ks < 0.8 * cur) s = s8;
    }
  }
  return s;
}
