// igemm5.hip - short-K plain GEMM with the activation rows held in REGISTERS and the weights streamed through LDS.
//
// The K <= 256 problems of the step (FFN linear1, the 256 -> 1024 convolutions of layer3 and their dgrads, layer1/2 1x1s) run
// on igemm3's 64x64 tile at 11-13 % of the MFMA peak: with 4 K tiles per output tile a workgroup moves 64 KB through the
// LDS-DMA path for 2.1 MFLOP (32 flop/B), and that path (~27 B/clk/CU) is the bound.  Here a workgroup owns 128 rows and
// walks over several 64-column tiles of the output:
//   * each wave keeps its 32 rows x K of the A operand as MFMA fragments in VGPRs (K = 256: 64 registers), loaded once;
//   * only the weight tile (64 columns x K: 32 KB) goes through LDS per output tile, double-buffered - 128 flop per byte
//     of LDS-DMA traffic, and the fragment reads from LDS halve as well (no A reads);
//   * the f32 staging tile of the epilogue aliases the weight buffer just consumed, so a workgroup needs 64 KB of LDS and
//     two of them share a CU: one's epilogue overlaps the other's MFMA phase.
// Envelope: bf16 in / bf16 out, plain A (1x1 stride-1 convolutions and linears), K in {64, 128, 192, 256}, N % 64 == 0.
#include <stdlib.h>
#include "igemm2_common.h"

namespace sedt {

template <int NK>      // K = NK * 64
__global__ __launch_bounds__(256, 2) void igemm5_kernel(const SedtIgemm p, const unsigned b_bytes, const int nsplit,
                                                        const int tiles_per) {
  constexpr int BM = 128, BN = 64, NT = 256;
  constexpr int SUB = BN * ROWB;                   // one [64 columns][64 k] sub-image of the weight tile: 8 KB
  constexpr int BUF = NK * SUB;                    // weight tile
  constexpr int GB = NK * 2;                       // LDS-DMA pieces per wave per weight tile
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // 2 weight buffers (>= 32 KB each: the Cs alias)
  constexpr int BUFSZ = BUF > BM * BN * 4 ? BUF : BM * BN * 4;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ntn = p.N / BN;
  const int mt = blockIdx.x / nsplit, sp = blockIdx.x - mt * nsplit;
  const int m0 = mt * BM;
  const int j0 = sp * tiles_per, j1 = min(ntn, j0 + tiles_per);
  if (j0 >= j1) return;

  // ---- A fragments: lane (frow, fhalf) holds row m0 + 32*wave + frow, k = ks*16 + fhalf*8 .. +7 for every k16 step
  const int frow = lane & 31, fhalf = lane >> 5;
  bf16x8 fa[NK * 4];
  {
    const int row = m0 + wave * 32 + frow;
    const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.A) + (long)row * p.lda + fhalf * 8;
#pragma unroll
    for (int ks = 0; ks < NK * 4; ++ks) {
      if (row < p.M) fa[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 16);
      else
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[ks][e] = (bf16_t)0.f;
    }
  }

  // ---- weight tile DMA: sub-image kb = columns' k range [kb*64, +64); rows = output columns; layout / swizzle of igemm3's B stage
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, b_bytes, 0x00020000);
  const int lrow = lane >> 3, pc = lane & 7;
  unsigned b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int trow = (i * 4 + wave) * 8 + lrow;
    const int swz = (pc ^ ((trow >> 1) & 7)) * 8;
    b_off[i] = (unsigned)(((long)trow * p.ldb + swz) * 2);
  }
  auto issue = [&](const int buf, const int j) {        // weight tile j -> buffer buf
    unsigned char* st = smem + buf * BUFSZ;
    const unsigned base = (unsigned)((long)j * BN * p.ldb * 2);
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        unsigned bv = base + b_off[i] + (unsigned)(kb * BK2 * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void*)(st + kb * SUB + ((i * 4 + wave) * 8) * ROWB), 16, bv, 0, 0, 0);
      }
  };
  int b_rd[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 32 + frow;
      b_rd[ks][j] = row * ROWB + (((ks * 2 + fhalf) ^ ((row >> 1) & 7)) * 16);
    }

  issue(0, j0);
  if (j0 + 1 < j1) issue(1, j0 + 1);

  const uint32_t seed = eff_seed(p.seed, p.seed_ptr);
  const uint32_t thresh = drop_threshold(p.drop_p);
  const float inv_keep = p.drop_p > 0.f ? 1.f / (1.f - p.drop_p) : 1.f;
  const bf16_t* resT = reinterpret_cast<const bf16_t*>(p.res);
  const bf16_t* maskT = reinterpret_cast<const bf16_t*>(p.mask);
  bf16_t* outT = reinterpret_cast<bf16_t*>(p.C);
  constexpr int CPR = BN / 8, NCH = BM * CPR / NT;       // 16-byte output chunks per row / per thread

  for (int j = j0; j < j1; ++j) {
    const int buf = (j - j0) & 1;
    const int n0 = j * BN;
    // own pieces of tile j landed (a later tile may stay in flight), then everybody's
    if (j + 1 < j1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // epilogue operands of this tile: issued now, consumed after the MFMAs
    bf16x8 res_pf[NCH], mask_pf[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int u = t + c * NT;
      const int row = m0 + u / CPR, col = n0 + (u % CPR) * 8;
      if (row < p.M) {
        if (resT) res_pf[c] = *reinterpret_cast<const bf16x8*>(resT + (long)(p.res_mod > 0 ? (row % p.res_mod) : row) * p.ldr + col);
        if (maskT) mask_pf[c] = *reinterpret_cast<const bf16x8*>(maskT + (long)row * p.ldm + col);
      }
    }

    f32x16 acc[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jj][r] = 0.f;
    const unsigned char* st = smem + buf * BUFSZ;
#pragma unroll
    for (int kb = 0; kb < NK; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x8 fb = *reinterpret_cast<const bf16x8*>(st + kb * SUB + b_rd[ks][jj]);
          acc[jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kb * 4 + ks], fb, acc[jj], 0, 0, 0);
        }
    __builtin_amdgcn_s_barrier();                    // every wave is done reading the weight tile: the buffer becomes Cs

    // ---- epilogue through LDS ([128][64] f32, column bit 5 flipped on rows with bit 2 set: the two half-waves of an
    //      accumulator store hit different banks)
    float* Cs = reinterpret_cast<float*>(smem + buf * BUFSZ);
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhalf;
        const int col = (jj * 32 + frow) ^ (((row >> 2) & 1) << 5);
        Cs[row * BN + col] = acc[jj][r];
      }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int u = t + c * NT;
      const int trow = u / CPR, cc = (u % CPR) * 8;
      const int row = m0 + trow, col = n0 + cc;
      if (row >= p.M) continue;
      float v[8];
      {
        const int pcol = cc ^ (((trow >> 2) & 1) << 5);
        const float4 x0 = *reinterpret_cast<const float4*>(Cs + trow * BN + pcol);
        const float4 x1 = *reinterpret_cast<const float4*>(Cs + trow * BN + pcol + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
      }
      if (p.scale) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.scale + col), s1 = *reinterpret_cast<const float4*>(p.scale + col + 4);
        v[0] *= s0.x; v[1] *= s0.y; v[2] *= s0.z; v[3] *= s0.w; v[4] *= s1.x; v[5] *= s1.y; v[6] *= s1.z; v[7] *= s1.w;
      }
      if (p.bias) {
        const float4 s0 = *reinterpret_cast<const float4*>(p.bias + col), s1 = *reinterpret_cast<const float4*>(p.bias + col + 4);
        v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
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
        const bf16x8 rv = res_pf[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
      }
      if (p.act_post_res && p.act == SEDT_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (maskT) {
        const bf16x8 mv = mask_pf[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ((float)mv[e] > 0.f) ? v[e] : 0.f;
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] * p.alpha);
      *reinterpret_cast<bf16x8*>(outT + (long)row * p.ldc + col) = o;
    }
    if (j + 2 < j1) {
      __syncthreads();                               // Cs has been read: the buffer takes the tile after next
      issue(buf, j + 2);
    }
  }
}

template <int NK>
static int launch5(const SedtIgemm& p, unsigned b_bytes, hipStream_t st) {
  constexpr size_t buf = (size_t)NK * 64 * ROWB > (size_t)128 * 64 * 4 ? (size_t)NK * 64 * ROWB : (size_t)128 * 64 * 4;
  constexpr size_t lds = 2 * buf;
  static bool attr_set = false;
  auto kern = igemm5_kernel<NK>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("igemm5: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int ntm = (p.M + 127) / 128, ntn = p.N / 64;
  // enough workgroups for two per CU; a workgroup re-uses its A rows over tiles_per column tiles
  int nsplit = (512 + ntm - 1) / ntm;
  if (nsplit > ntn) nsplit = ntn;
  if (nsplit < 1) nsplit = 1;
  const int tiles_per = (ntn + nsplit - 1) / nsplit;
  nsplit = (ntn + tiles_per - 1) / tiles_per;
  hipLaunchKernelGGL(kern, dim3(ntm * nsplit), dim3(256), lds, st, p, b_bytes, nsplit, tiles_per);
  return check_launch("igemm5");
}

// -1 = outside the envelope
int igemm5_try(const SedtIgemm& p, unsigned b_bytes, hipStream_t st) {
  static int on = -1, min_m = 0;
  if (on < 0) {
    const char* e = getenv("SEDT_IGEMM_V5");
    on = (e && e[0] == '0') ? 0 : 1;
    const char* m = getenv("SEDT_IGEMM5_MIN_M");
    min_m = m ? atoi(m) : 4096;
  }
  if (!on || p.conv || p.trans || p.out_f32 || p.splitk > 1 || p.act == SEDT_ACT_SIGMOID) return -1;
  if (p.K % 64 || p.K > 256 || p.N % 64 || p.N < 128 || p.M < min_m) return -1;
  if ((p.lda & 7) || (reinterpret_cast<uintptr_t>(p.A) & 15)) return -1;
  if ((reinterpret_cast<uintptr_t>(p.scale) & 15) || (reinterpret_cast<uintptr_t>(p.bias) & 15)) return -1;
  switch (p.K / 64) {
    case 1: return launch5<1>(p, b_bytes, st);
    case 2: return launch5<2>(p, b_bytes, st);
    case 3: return launch5<3>(p, b_bytes, st);
    default: return launch5<4>(p, b_bytes, st);
  }
}

}  // namespace sedt
