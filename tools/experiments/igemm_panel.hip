// EXPERIMENT (round 3), not part of the build.  Correct (the linear-epilogue / 1-bit-mask tests pass on it) and SLOWER than the
// tiled kernel it was meant to replace - measured on MI355X, 50 launches in a HIP graph:
//   8192 x 2048 x 256: 27.6 us vs 24.2 us (64x64 tiles);  8192 x 1024 x 256: 15.2 vs 11.8;  8192 x 512 x 256: 9.3 vs 7.3.
// Where the 27.6 us go (parts switched off one by one): K loops + prologue 12.4 us (16 steps of ~0.6 us - the same ~1450 clk per
// 128x128x64 step as the ping-pong kernel of igemm3.hip on its best shape - plus ~2.3 us for 256 workgroups fetching their A
// panels at once); LDS staging of the two wave groups' K halves 5.8 us; global stores 3.3 us; barriers / conversion / the
// vmcnt drain 4.8 us.  With ONE workgroup per CU every epilogue is serial work; the tiled kernel hides its epilogues behind
// the K loops of the other four workgroups of the CU.  Even with a 1 us epilogue the ceiling is ~16 us, i.e. ~1 % of the step.
// (tools/experiments/igemm5.hip is the round-2 attempt at the same shapes with A in registers: same conclusion.)
//
// igemm_panel.hip - the "row panel" bf16 GEMM for short-K, wide-N problems (K = 128 / 256, N >= 512, plain row-major A):
// FFN1 forward / FFN2 input gradient (8192 x 2048 x 256), layer3 conv3 forward / conv1 input gradient (8192 x 1024 x 256).
//
// Why: on the tiled kernels (igemm3.hip) these problems are bound by the LDS fill path, not by the matrix cores or HBM.  With K = 256
// a 64x64 tile stages 64 KB (A and W tile, 4 K tiles each) for 2.1 MFLOP - 32 flop per staged byte - and A is staged N/64 times:
// 8192 x 2048 x 256 moved 268 MB through the LDS-DMA path for 5 MB of operands and ran at 11 % of the matrix peak (30 us);
// larger tiles cut the traffic but expose their prologue / epilogue (one workgroup per CU, 4 K tiles between them).
// Here a workgroup keeps its 128 rows of A - ALL of K, 64 KB - resident in LDS and walks over several 128-column tiles of the
// output: only W streams (a 3-stage ring of 16 KB K tiles, 128 flop per staged byte), the ring runs ahead across tile
// boundaries, so a tile's epilogue overlaps the next tile's first W tiles, and the fixed per-workgroup costs are paid once
// per 4 tiles.  Schedule inside a tile: the ping-pong of igemm3.hip (two groups of four waves, each group owns the whole
// 128x128 tile and half of the k16 steps, one barrier apart: one wave per SIMD reads fragments / issues DMA while the other
// multiplies).  Epilogue: the two groups' halves of the K sum meet in a 64-row f32 staging tile (two passes per tile), then
// the same per-chunk epilogue as igemm3.hip (scale / bias / ReLU / dropout / residual / mask / 1-bit mask / sign bits).
//
// Counted waits and the epilogue: stores and loads share the vector-memory counter, so after an epilogue's stores the first
// load segment of the next tile drains the counter (vmcnt(0)) BEFORE it issues anything new; the next tile's epilogue
// operands (residual, mask, per-column scale / bias) are requested right after that, behind the first W tile, and the
// second step's wait allows exactly those requests to stay in flight.
#include <stdlib.h>
#include <algorithm>
#include "lds_gemm_common.h"

namespace sedt {

constexpr int PN_BM = 128, PN_BN = 128, PN_NW = 8, PN_S = 3;
constexpr int PN_IMG = PN_BM * ROWB;                 // 16 KB: one K tile of the A panel = one ring stage of W
constexpr int PN_CP = PN_BN + 4;
constexpr int PN_STAGING = 64 * PN_CP * 4;           // f32 staging of 64 output rows
constexpr int PN_G = 2;                              // DMA instructions per wave per W stage

__device__ __forceinline__ int pn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

#define PN_WAIT_CASE(n) \
  case n: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory"); break;

// wait until at most `n` vector-memory operations of this wave are outstanding (n is wave-uniform, 2 <= n <= 18 here)
__device__ __forceinline__ void pn_wait_vm(const int n) {
  switch (n) {
    PN_WAIT_CASE(2) PN_WAIT_CASE(3) PN_WAIT_CASE(4) PN_WAIT_CASE(5) PN_WAIT_CASE(6) PN_WAIT_CASE(7) PN_WAIT_CASE(8) PN_WAIT_CASE(9)
    PN_WAIT_CASE(10) PN_WAIT_CASE(11) PN_WAIT_CASE(12) PN_WAIT_CASE(13) PN_WAIT_CASE(14) PN_WAIT_CASE(15) PN_WAIT_CASE(16)
    PN_WAIT_CASE(17) PN_WAIT_CASE(18)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// grid: panels x nsplit workgroups; workgroup (panel, part) computes rows [128 panel, +128) x column tiles [jt0, jt1)
// epi_bytes: bytes addressable through the residual (x) and bit-mask (y) descriptors
__global__ __launch_bounds__(512) void igemm_panel_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes,
                                                          const int nsplit, const uint2 epi_bytes) {
  constexpr int BM = PN_BM, NW = PN_NW, S = PN_S, G = PN_G;
  constexpr int CPR = PN_BN / 8;                       // 16-byte chunks per output tile row
  constexpr int NT = NW * 64;
  constexpr int NCH = 4;                               // chunks per thread and tile: 2 per 64-row half
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = pn_uniform(t >> 6);
  const int kgrp = wave >> 2;
  const int wm = ((wave & 3) >> 1) * 64, wn = (wave & 1) * 64;

  const int npanel = (p.M + BM - 1) / BM, ntn = p.N / PN_BN;
  const int nwg = npanel * nsplit;
  int vid;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    vid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  // the nsplit workgroups of one panel are neighbours in vid, i.e. on the same XCD: the panel's A rows miss its L2 once
  const int panel = vid / nsplit, part = vid - panel * nsplit;
  const int per = (ntn + nsplit - 1) / nsplit;
  const int jt0 = part * per, jt1 = min(ntn, jt0 + per);
  if (jt0 >= jt1) return;
  const int m0 = panel * BM;
  const int nkb = p.K / BK2;                           // 2 .. 4 K tiles
  const int T = (jt1 - jt0) * nkb;                     // W stages this workgroup consumes

  unsigned char* const Apanel = smem;
  unsigned char* const ring = smem + nkb * PN_IMG;
  float* const Cs = reinterpret_cast<float*>(ring + S * PN_IMG);

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);

  // ---- epilogue operands of one tile, requested ahead of its K loop
  const bf16_t* resT = reinterpret_cast<const bf16_t*>(p.res);
  const bool mbits = p.mask_bits != 0;
  const bf16_t* maskT = mbits ? nullptr : reinterpret_cast<const bf16_t*>(p.mask);
  const uint8_t* maskB = mbits ? reinterpret_cast<const uint8_t*>(p.mask) : nullptr;
  const int n_epi = (resT ? NCH : 0) + (maskB ? NCH : 0) + (p.scale ? 2 : 0) + (p.bias ? 2 : 0);   // requests per tile
  bf16x8 res_pf[NCH];
  uint32_t mbit_pf[NCH];
  float sc8[8], bi8[8];                                // per-column affine of the current tile (this thread's 8 columns)
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc8[e] = 1.f; bi8[e] = 0.f; }
  const int crow = t / CPR, ccol8 = (t % CPR) * 8;     // this thread's chunk: rows crow + 32 c (c = 0, 1) of a half, columns ccol8..+8
  // Per-thread byte offsets of the four chunks, computed ONCE: a tile's requests are then "uniform base (advanced per tile on
  // the scalar unit) + constant 32-bit lane offset" - no address arithmetic in vector registers next to in-flight loads (the
  // first version recomputed 64-bit addresses per tile, the allocator reused destination registers of pending loads for
  // them and the compiler serialised the requests with vmcnt(0) waits).  Rows past M read row M-1 and are never used.
  uint32_t res_voff[NCH], mb_voff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int row = min(m0 + (c >> 1) * 64 + (c & 1) * 32 + crow, p.M - 1);
    const int rrow = (resT && p.res_mod > 0) ? row % p.res_mod : row;
    res_voff[c] = (uint32_t)(((long)rrow * p.ldr + ccol8) * 2);
    mb_voff[c] = (uint32_t)((long)row * p.ldm + (ccol8 >> 3));
  }
  const uint32_t col_voff = (uint32_t)(ccol8 * 4);
  // Requests go through buffer descriptors: scalar base + the constant lane offset + a scalar per-tile offset, destination
  // registers that nothing else writes.  Every request is issued unconditionally and fenced against compiler reordering: the
  // counted wait of a tile's second step relies on exactly n_epi requests sitting behind the first W tile.
  typedef __attribute__((ext_vector_type(4))) float pn_f32x4;
  __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(resT), 0, resT ? epi_bytes.x : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(maskB), 0, maskB ? epi_bytes.y : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.scale), 0, p.scale ? p.N * 4 : 0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);
  auto prefetch_epi = [&](const int n0) {
    asm volatile("" ::: "memory");
    if (p.scale) {
      // (whole-vector bit casts: __builtin_bit_cast of a vector ELEMENT compiled to a one-dword load of element 0, hipcc 7.2)
      const pn_f32x4 s0 = __builtin_bit_cast(pn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, col_voff, n0 * 4, 0));
      const pn_f32x4 s1 = __builtin_bit_cast(pn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, col_voff + 16, n0 * 4, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { sc8[e] = s0[e]; sc8[4 + e] = s1[e]; }
    }
    if (p.bias) {
      const pn_f32x4 s0 = __builtin_bit_cast(pn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBi, col_voff, n0 * 4, 0));
      const pn_f32x4 s1 = __builtin_bit_cast(pn_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsBi, col_voff + 16, n0 * 4, 0));
#pragma unroll
      for (int e = 0; e < 4; ++e) { bi8[e] = s0[e]; bi8[4 + e] = s1[e]; }
    }
    if (resT) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) res_pf[c] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsR, res_voff[c], n0 * 2, 0));
    }
    if (maskB) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) mbit_pf[c] = __builtin_amdgcn_raw_buffer_load_b8(rsM, mb_voff[c], n0 >> 3, 0);
    }
    asm volatile("" ::: "memory");
  };

  // ---- DMA lanes: a piece = 8 rows x 128 B; 16 pieces per 128-row image, two per wave
  const int lrow = lane >> 3, pc = lane & 7;
  unsigned a_off[G], b_off[G];
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const int trow = (i * NW + wave) * 8 + lrow;
    const int swz = (pc ^ ((trow >> 1) & 7)) * 8;
    const int row = m0 + trow;
    a_off[i] = row < p.M ? (unsigned)(((long)row * p.lda + swz) * 2) : OOB;
    b_off[i] = (unsigned)(((long)(jt0 * PN_BN + trow) * p.ldb + swz) * 2);
  }
  const unsigned b_tile_step = (unsigned)((long)PN_BN * p.ldb * 2);
  auto issue_a = [&](const int kb) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      unsigned voff = a_off[i] + (unsigned)(kb * ROWB);        // (an out-of-range offset stays out of range)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void*)(Apanel + kb * PN_IMG + ((i * NW + wave) * 8) * ROWB), 16, voff, 0, 0, 0);
    }
  };
  int w_kb = 0;                                        // K tile of the W stage issued next (wave-uniform)
  auto issue_w = [&](const int stage) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      unsigned bv = b_off[i] + (unsigned)(w_kb * ROWB);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(ring + stage * PN_IMG + ((i * NW + wave) * 8) * ROWB), 16, bv, 0, 0, 0);
    }
    if (++w_kb == nkb) {
      w_kb = 0;
#pragma unroll
      for (int i = 0; i < G; ++i) b_off[i] += b_tile_step;
    }
  };

  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  zero_acc();

  // ---- fragment read offsets inside an image (constants of the thread): this wave's two k16 steps of a K tile
  const int frow = lane & 31, fhalf = lane >> 5;
  int a_rd[2][2], b_rd[2][2];
#pragma unroll
  for (int kq = 0; kq < 2; ++kq) {
    const int ks = 2 * kgrp + kq;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm + i * 32 + frow;
      a_rd[kq][i] = row * ROWB + (((ks * 2 + fhalf) ^ ((row >> 1) & 7)) * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn + j * 32 + frow;
      b_rd[kq][j] = row * ROWB + (((ks * 2 + fhalf) ^ ((row >> 1) & 7)) * 16);
    }
  }
  bf16x8 fa[2][2], fb[2][2];
  auto load_frags = [&](const int stage, const int kb) {
    const unsigned char* ai = Apanel + kb * PN_IMG;
    const unsigned char* wi = ring + stage * PN_IMG;
#pragma unroll
    for (int kq = 0; kq < 2; ++kq) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[kq][i] = *reinterpret_cast<const bf16x8*>(ai + a_rd[kq][i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[kq][j] = *reinterpret_cast<const bf16x8*>(wi + b_rd[kq][j]);
    }
  };
  auto mfma_all = [&]() {
#pragma unroll
    for (int kq = 0; kq < 2; ++kq)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kq][i], fb[kq][j], acc[i][j], 0, 0, 0);
  };

  // ---- epilogue of the tile whose first column is n0c
  const uint32_t seed = eff_seed(p.seed, p.seed_ptr);
  const uint32_t thresh = drop_threshold(p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  bf16_t* outT = reinterpret_cast<bf16_t*>(p.C);
  const bool affine = p.scale != nullptr || p.bias != nullptr;
  const bool relu_pre = !p.act_post_res && p.act == SEDT_ACT_RELU, relu_post = p.act_post_res && p.act == SEDT_ACT_RELU;
  int n0c = jt0 * PN_BN;
  int q = 0, kb = 0, tile = 0;                         // step, K tile of the step, tiles finished (all wave-uniform)
  auto epilogue = [&]() {
    if (kgrp == 0) lds_barrier();                      // the second group is one barrier behind: let it finish its MFMAs
    // the tile's epilogue operands have landed by now; saying so with the BUILTIN (which the compiler's wait-count pass
    // models, unlike inline asm) keeps it from waiting vmcnt(0) again before every chunk - i.e. for the previous chunk's stores
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0), expcnt / lgkmcnt untouched
    const int col = n0c + ccol8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool mine = wm == h * 64;                  // this wave holds rows of half h
      if (mine && kgrp == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
              Cs[row * PN_CP + wn + j * 32 + frow] = acc[i][j][r];
            }
      }
      lds_barrier();
      if (mine && kgrp == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
              Cs[row * PN_CP + wn + j * 32 + frow] += acc[i][j][r];
            }
      }
      lds_barrier();
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int c = h * 2 + c2;
        const int trow = c2 * 32 + crow;               // row inside the half
        const int row = m0 + h * 64 + trow;
        if (row >= p.M) continue;
        float v[8];
        {
          const float4 x0 = *reinterpret_cast<const float4*>(Cs + trow * PN_CP + ccol8);
          const float4 x1 = *reinterpret_cast<const float4*>(Cs + trow * PN_CP + ccol8 + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
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
          const uint64_t h0 = ((uint64_t)row * (uint64_t)p.N + col) >> 1;
          const uint32_t lo = (uint32_t)h0;
          const uint32_t inner = mix32(seed ^ ((uint32_t)(h0 >> 32) * 0x9E3779B9U) ^ 0x85ebca6bU);
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const uint32_t hh = mix32((lo + qq) ^ inner);
            v[2 * qq] = (hh & 0xffffu) >= thresh ? v[2 * qq] * inv_keep : 0.f;
            v[2 * qq + 1] = (hh >> 16) >= thresh ? v[2 * qq + 1] * inv_keep : 0.f;
          }
        }
        if (resT) {
          const bf16x8 rv = res_pf[c];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
        }
        if (relu_post) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (maskT) {                                     // (bf16 masks are the rare form: read here, not ahead of the tile)
          const bf16x8 mv = *reinterpret_cast<const bf16x8*>(maskT + (long)row * p.ldm + col);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((float)mv[e] > 0.f) ? v[e] : 0.f;
        }
        if (maskB) {
          const uint32_t mb = mbit_pf[c];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
        }
        if (p.alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
        *reinterpret_cast<bf16x8*>(outT + (long)row * p.ldc + col) = o;
        if (p.bits_out) {
          uint32_t ob = 0;
#pragma unroll
          for (int e = 0; e < 8; ++e) ob |= ((float)o[e] > 0.f ? 1u : 0u) << e;
          p.bits_out[(long)row * p.ldbits + (col >> 3)] = (uint8_t)ob;
        }
      }
      if (h == 0) lds_barrier();                       // the staging tile is rewritten by the other half
    }
    zero_acc();
    n0c += PN_BN;
    ++tile;
    if (q < T && kgrp == 1) lds_barrier();             // back to the ping-pong skew for the next tile
  };

  // ---- prologue: epilogue operands of the first tile, the whole A panel, the first S-1 W stages
  prefetch_epi(n0c);
  for (int k = 0; k < nkb; ++k) issue_a(k);
#pragma unroll
  for (int s0 = 0; s0 < S - 1; ++s0)
    if (s0 < T) issue_w(s0);
  if (T >= S - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (kgrp == 1) lds_barrier();

  // ---- the ring, unrolled by its depth (stage indices are literals); q % S == ph at every executed phase
  while (q < T) {
#pragma unroll
    for (int ph = 0; ph < S; ++ph) {
      if (q < T) {
        const bool more = q + S - 1 < T;
        lds_barrier();
        load_frags(ph, kb);
        if (tile > 0 && kb == 0) {
          // first step after an epilogue: its stores (and everything older) retire before counted waits mean loads again;
          // the W tile of the next step was issued two steps ago - it is covered by this drain
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (more) issue_w((ph + S - 1) % S);
          prefetch_epi(n0c);
        } else if (tile > 0 && kb == 1) {
          // second step: behind the W tile it waits for sit this tile's epilogue requests and the stage issued just now
          if (more) {
            issue_w((ph + S - 1) % S);
            pn_wait_vm(n_epi + (S - 2) * G);
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        } else {
          if (more) {
            issue_w((ph + S - 1) % S);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((S - 2) * G) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();
        __builtin_amdgcn_s_setprio(1);
        mfma_all();
        __builtin_amdgcn_s_setprio(0);
        ++q;
        if (++kb == nkb) {
          kb = 0;
          epilogue();
        }
      }
    }
  }
}

constexpr size_t pn_lds_bytes(int nkb) { return (size_t)nkb * PN_IMG + (size_t)PN_S * PN_IMG + PN_STAGING; }

// -1: outside the envelope (the caller goes on to the tiled kernels).  The caller (igemm_lds_try) has already checked bf16,
// trans == 0, the alignment of every operand and the 31-bit descriptor ranges.
int igemm_panel_try(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  static int on = -1, min_m = 0, min_k = 0;
  if (on < 0) {
    const char* e = getenv("SEDT_IGEMM_PANEL");
    on = (e && e[0] == '0') ? 0 : 1;
    e = getenv("SEDT_IGEMM_PANEL_MIN_M");
    min_m = e ? atoi(e) : 4096;
    e = getenv("SEDT_IGEMM_PANEL_MIN_K");
    min_k = e ? atoi(e) : 256;
  }
  if (!on || p.conv || p.trans || p.out_f32 || p.splitk > 1 || p.act == SEDT_ACT_SIGMOID) return -1;
  if (p.K % BK2 || p.K > 256 || p.K < min_k || p.N % PN_BN || p.N < 512 || p.M < min_m) return -1;
  if ((reinterpret_cast<uintptr_t>(p.scale) & 15) || (reinterpret_cast<uintptr_t>(p.bias) & 15)) return -1;
  const long res_rows = p.res_mod > 0 ? std::min(p.M, p.res_mod) : p.M;
  const long res_bytes = p.res ? ((res_rows - 1) * p.ldr + p.N) * 2 : 0;
  const long mask_bytes = (p.mask && p.mask_bits) ? (long)(p.M - 1) * p.ldm + p.N / 8 : 0;
  if (res_bytes >= (1L << 31) || mask_bytes >= (1L << 31)) return -1;
  const int npanel = (p.M + PN_BM - 1) / PN_BM, ntn = p.N / PN_BN;
  // enough workgroups for every CU; beyond that, more column tiles per workgroup (the A panel and the fixed costs amortise)
  int nsplit = std::min(ntn, std::max(1, (256 + npanel - 1) / npanel));
  const int nkb = p.K / BK2;
  const size_t lds = pn_lds_bytes(nkb);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)pn_lds_bytes(4));
    if (e != hipSuccess) {
      set_error("igemm_panel: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(igemm_panel_kernel, dim3(npanel * nsplit), dim3(512), lds, st, p, a_bytes, b_bytes, nsplit,
                     make_uint2((unsigned)res_bytes, (unsigned)mask_bytes));
  return check_launch("igemm_panel");
}

}  // namespace sedt
