// ffn_split.hip - the transformer FFN pair  y = x1 + drop(drop(relu(x W1^T + b1)) W2^T + b2)  and its input-gradient chain as ONE
// launch each, tiled in TWO dimensions: a workgroup = (block of 128 rows) x (quarter of the hidden features).
//
// Why: with one 32-row slab per workgroup (enc_slab.hip) every CU streams ALL of W1 and W2 (2 MB at FF = 2048) for its rows - the
// FFN's time is that stream (~24 us at the ~40 B/clk a CU gets from L2, measured 35 us), whatever M is.  Here a workgroup streams a
// QUARTER of the weights (512 KB) and every weight fragment feeds four MFMAs (four row slabs): at M = 8192 the 256 workgroups move
// 131 MB instead of 538 MB through L2, and the MFMA time (2048 per workgroup) is level with the stream.  The price is a reduction
// over the four hidden quarters of a row block: each workgroup leaves its [128][256] f32 partial sum in a scratch buffer, takes a
// ticket, and the LAST of the four adds the partials in a fixed order (bit-reproducible) and applies the epilogue - the split-K seam
// of cdna_hip_programming.md Guideline 16 (agent-scope release / acquire, placement independent).
// Rounding points / dropout hashes as the per-op chain.  Envelope: bf16, d = 256, FF a multiple of 1024.
#include "slab.h"

namespace sedt {

using slab::u32x4;
using slab::XP;

constexpr int FS_D = 256, FS_RB = 128, FS_NQ = 4;

struct FfnSplitArgs {
  const bf16_t* xin;                                  // forward: x1n (the FFN input); backward: gx2 (gradient wrt the FFN output)
  const bf16_t* res;                                  // forward: x1 (residual); backward: unused
  const u32x4* wa;                                    // forward: W1 [FF][256]; backward: W2^T (features = hidden, contraction 256)
  const u32x4* wb;                                    // forward: W2 [256][FF]; backward: W1^T (features = 256, contraction FF)
  const float* b1; const float* b2;                   // forward only
  bf16_t* h;                                          // forward: the dropped ReLU output [M][FF] (written when training); backward: read
  bf16_t* out;                                        // forward: x2 [M][256]; backward: g_x1n [M][256]
  bf16_t* aux;                                        // backward: g2 = dropout'(gx2) [M][256] (or null when drop_p == 0); forward: unused
  bf16_t* gh;                                         // backward: the hidden gradient [M][FF]
  float* part;                                        // [4][Mpad][256] f32 partial sums
  unsigned* cnt;                                      // [row blocks] arrival counters (zero before the first launch; re-armed here)
  int M, FF;
  float drop_p;
  uint32_t thresh, seed_h, seed_f;
  const uint32_t* seed_ptr;
  int dbg;                                            // developer builds: phase ablation (WRONG results)
};

// the reduction of a row block by its last workgroup: sum of the four partials (fixed order) -> f(row, col0, v[8])
template <class F>
__device__ __forceinline__ void reduce_block(const float* __restrict__ part, long Mpad, long row0, int nvalid, int tid, F f) {
#pragma unroll 2
  for (int u = tid; u < FS_RB * 32; u += 512) {
    const int r = u >> 5, c = (u & 31) * 8;
    if (r >= nvalid) continue;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < FS_NQ; ++q) {
      const float* p = part + ((long)q * Mpad + row0 + r) * FS_D + c;
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
      v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
    f(r, c, v);
  }
}

// partial accumulators of output tile `wave` (4 slabs) -> part[q][row][col] as 16-byte WRITE-THROUGH stores (slab::store_sc1)
__device__ __forceinline__ void store_partial(const f32x16 (&acc)[4], float* __restrict__ part, long Mpad, int q, long row0, int nvalid, int wave,
                                              int lane) {
  const int n = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int r = sl * 32 + n;
    if (r >= nvalid) continue;
    float* p = part + ((long)q * Mpad + row0 + r) * FS_D + wave * 32 + 4 * hf;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      slab::store_sc1(p + 8 * g4, slab::f32x4_t{acc[sl][4 * g4 + 0], acc[sl][4 * g4 + 1], acc[sl][4 * g4 + 2], acc[sl][4 * g4 + 3]});
  }
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <bool TRAIN>
__global__ __launch_bounds__(512) void ffn_split_fwd_kernel(const FfnSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* X = reinterpret_cast<bf16_t*>(smem);                     // [128][XP] the block's rows of x1n
  bf16_t* H = X + FS_RB * XP;                                      // [128][XP] 256 hidden features of the block
  unsigned* FLAG = reinterpret_cast<unsigned*>(H + FS_RB * XP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
  const int hq = blockIdx.x & 3, rb = blockIdx.x >> 2;
  const long row0 = (long)rb * FS_RB;
  const int nvalid = min(FS_RB, a.M - (int)row0);
  const long Mpad = (long)((a.M + FS_RB - 1) / FS_RB) * FS_RB;
  const int FF = a.FF, HQ = FF / FS_NQ, nsub = HQ / 256;           // hidden features of this workgroup: [hq * HQ, +HQ) in sub-chunks of 256
  const float inv_keep = a.thresh ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t sd_off = a.seed_ptr ? *a.seed_ptr : 0u;
  const uint32_t sd_h = a.seed_h + sd_off, sd_f = a.seed_f + sd_off;
  const long ts256 = 64L * 16, ts_ff = 64L * (FF / 16);
#ifdef SEDT_DEV
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
  slab::u32x4 wa[8], wb[8];
  const int t1_0 = (hq * HQ) / 32 + wave;                          // this wave's linear1 tile of sub-chunk 0
  slab::load_chunk<1>(wa, a.wa + (long)t1_0 * ts256, 0, 0, lane);
  slab::issue_fence();
  {
    uint4 xr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      xr[q] = r < nvalid ? *reinterpret_cast<const uint4*>(a.xin + (row0 + r) * FS_D + c) : make_uint4(0, 0, 0, 0);
    }
    slab::issue_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      *reinterpret_cast<uint4*>(X + r * XP + c) = xr[q];
    }
  }
  __syncthreads();
  f32x16 acc2[4];
  slab::zero_acc(acc2);
  for (int sc = 0; sc < nsub; ++sc) {
    const int t1 = (hq * HQ + sc * 256) / 32 + wave;               // linear1 tile (32 hidden features) of this wave
    const slab::u32x4* w2c = a.wb + (long)wave * ts_ff + (long)((hq * HQ + sc * 256) / 16) * 64;       // linear2 tile `wave`, this sub-chunk's k-steps
    {
      f32x16 acc[4];
      slab::zero_acc(acc);
      float4 bb[4];
      slab::load_feat4(bb, a.b1, t1, hf);
      slab::issue_fence();
      if (!(dbg & 1))
      slab::wave_gemm_r4<16>(acc, X, XP, a.wa + (long)t1 * ts256, lane, wa, wb, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w2c, 0, 0, lane); });
      if (!(dbg & 16))
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) {
        const int r = sl * 32 + n;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int fl = wave * 32 + 8 * g4 + 4 * hf;                // column inside the sub-chunk
          const int f = hq * HQ + sc * 256 + fl;                     // hidden feature
          const float bv[4] = {bb[g4].x, bb[g4].y, bb[g4].z, bb[g4].w};
          const uint64_t idx = (uint64_t)(row0 + r) * FF + f;
          uint32_t keep = 0xfu;
          if (a.thresh) keep = drop_keep4(slab::inner0(sd_h), 0u, sd_h, idx, a.thresh);
          VecT<bf16_t, 4> o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((r < nvalid && (keep >> e & 1u)) ? fmaxf(acc[sl][4 * g4 + e] + bv[e], 0.f) * inv_keep : 0.f);
          *reinterpret_cast<VecT<bf16_t, 4>*>(H + r * XP + fl) = o;
        }
      }
    }
    __syncthreads();
    if (TRAIN && !(dbg & 8)) slab::tile_to_global(H, XP, a.h + row0 * FF + hq * HQ + sc * 256, FF, nvalid, 256, tid, 512);
    if (!(dbg & 2))
    slab::wave_gemm_r4<16>(acc2, H, XP, w2c, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (sc + 1 < nsub) slab::load_chunk<1>(d, a.wa + (long)(t1 + 8) * ts256, 0, 0, lane);
    });
    __syncthreads();                                               // (the next sub-chunk overwrites H)
  }
  if (dbg & 4) return;
  store_partial(acc2, a.part, Mpad, hq, row0, nvalid, wave, lane);
  if (!slab::arrive_last(a.cnt + rb, FS_NQ, FLAG, tid)) return;
  // ---- the last of the four: x2 = x1 + dropout(sum of the partials + b2)
  reduce_block(a.part, Mpad, row0, nvalid, tid, [&](int r, int c, const float* v) {
    const long base = (row0 + r) * FS_D + c;
    const float4 b0 = *reinterpret_cast<const float4*>(a.b2 + c), b1v = *reinterpret_cast<const float4*>(a.b2 + c + 4);
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1v.x, b1v.y, b1v.z, b1v.w};
    const bf16x8 xr = *reinterpret_cast<const bf16x8*>(a.res + base);
    const uint32_t keep = a.thresh ? drop_keep8(sd_f, (uint64_t)base, a.thresh) : 0xffu;
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((keep >> e & 1u) ? (v[e] + bv[e]) * inv_keep : 0.f) + (float)xr[e]);
    *reinterpret_cast<bf16x8*>(a.out + base) = o;
  });
}

// ---------------------------------------------------------------------------------------------------------------- backward
// g2 = dropout'(gx2); gh = (g2 W2) [h > 0] / (1 - p)  (this workgroup's quarter of the hidden features); g_x1n = gh W1 summed over
// the quarters by the last workgroup of the row block
__global__ __launch_bounds__(512) void ffn_split_bwd_kernel(const FfnSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* G2 = reinterpret_cast<bf16_t*>(smem);                    // [128][XP]
  bf16_t* GH = G2 + FS_RB * XP;                                    // [128][XP]: first the h sub-chunk, then gh in place
  unsigned* FLAG = reinterpret_cast<unsigned*>(GH + FS_RB * XP);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, hf = lane >> 5;
  const int hq = blockIdx.x & 3, rb = blockIdx.x >> 2;
  const long row0 = (long)rb * FS_RB;
  const int nvalid = min(FS_RB, a.M - (int)row0);
  const long Mpad = (long)((a.M + FS_RB - 1) / FS_RB) * FS_RB;
  const int FF = a.FF, HQ = FF / FS_NQ, nsub = HQ / 256;
  const float inv_keep = a.thresh ? 1.f / (1.f - a.drop_p) : 1.f;
  const uint32_t sd_off = a.seed_ptr ? *a.seed_ptr : 0u;
  const uint32_t sd_f = a.seed_f + sd_off;
  const long ts256 = 64L * 16, ts_ff = 64L * (FF / 16);
  slab::u32x4 wa[8], wb[8];
  slab::load_chunk<1>(wa, a.wa + (long)((hq * HQ) / 32 + wave) * ts256, 0, 0, lane);
  slab::issue_fence();
  {
    bf16x8 gin[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) gin[q][e] = (bf16_t)0.f;
      if (r < nvalid) gin[q] = *reinterpret_cast<const bf16x8*>(a.xin + (row0 + r) * FS_D + c);
    }
    slab::issue_fence();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
      bf16x8 o = gin[q];
      if (a.thresh && r < nvalid) {
        const long base = (row0 + r) * FS_D + c;
        const uint32_t keep = drop_keep8(sd_f, (uint64_t)base, a.thresh);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (keep >> e & 1u) ? (bf16_t)((float)gin[q][e] * inv_keep) : (bf16_t)0.f;
        if (a.aux && hq == 0) *reinterpret_cast<bf16x8*>(a.aux + base) = o;
      }
      *reinterpret_cast<bf16x8*>(G2 + r * XP + c) = o;
    }
  }
  __syncthreads();
  f32x16 acc1[4];
  slab::zero_acc(acc1);
  for (int sc = 0; sc < nsub; ++sc) {
    const int t2 = (hq * HQ + sc * 256) / 32 + wave;               // W2^T tile (32 hidden features) of this wave
    const slab::u32x4* w1c = a.wb + (long)wave * ts_ff + (long)((hq * HQ + sc * 256) / 16) * 64;       // W1^T tile `wave`, this sub-chunk's k-steps
    f32x16 acc[4];
    slab::zero_acc(acc);
    slab::wave_gemm_r4<16>(acc, G2, XP, a.wa + (long)t2 * ts256, lane, wa, wb, [&](slab::u32x4(&d)[8]) { slab::load_chunk<1>(d, w1c, 0, 0, lane); });
    {   // this sub-chunk of h into the GH tile (its sign is the ReLU / dropout mask)
      uint4 hr[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
        hr[q] = r < nvalid ? *reinterpret_cast<const uint4*>(a.h + (row0 + r) * FF + hq * HQ + sc * 256 + c) : make_uint4(0, 0, 0, 0);
      }
      slab::issue_fence();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int u = tid + q * 512, r = u >> 5, c = (u & 31) * 8;
        *reinterpret_cast<uint4*>(GH + r * XP + c) = hr[q];
      }
    }
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const int r = sl * 32 + n;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int fl = wave * 32 + 8 * g4 + 4 * hf;
        const VecT<bf16_t, 4> hv = *reinterpret_cast<const VecT<bf16_t, 4>*>(GH + r * XP + fl);
        VecT<bf16_t, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = (bf16_t)((float)hv.v[e] > 0.f ? acc[sl][4 * g4 + e] * inv_keep : 0.f);
        *reinterpret_cast<VecT<bf16_t, 4>*>(GH + r * XP + fl) = o;     // (in place: this lane alone touches these four elements)
      }
    }
    __syncthreads();
    slab::tile_to_global(GH, XP, a.gh + row0 * FF + hq * HQ + sc * 256, FF, nvalid, 256, tid, 512);
    slab::wave_gemm_r4<16>(acc1, GH, XP, w1c, lane, wa, wb, [&](slab::u32x4(&d)[8]) {
      if (sc + 1 < nsub) slab::load_chunk<1>(d, a.wa + (long)(t2 + 8) * ts256, 0, 0, lane);
    });
    __syncthreads();
  }
  store_partial(acc1, a.part, Mpad, hq, row0, nvalid, wave, lane);
  if (!slab::arrive_last(a.cnt + rb, FS_NQ, FLAG, tid)) return;
  reduce_block(a.part, Mpad, row0, nvalid, tid, [&](int r, int c, const float* v) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
    *reinterpret_cast<bf16x8*>(a.out + (row0 + r) * FS_D + c) = o;
  });
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_ffn_split_ok(int D, int FF, int dtype) { return dtype == SEDT_BF16 && D == FS_D && FF >= 1024 && FF % 1024 == 0; }

extern "C" size_t sedt_ffn_split_part_floats(int M) { return (size_t)FS_NQ * ((size_t)(M + FS_RB - 1) / FS_RB) * FS_RB * FS_D; }
extern "C" int sedt_ffn_split_blocks(int M) { return (M + FS_RB - 1) / FS_RB; }

static int ffn_split_launch(bool bwd, bool train, const FfnSplitArgs& a, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * FS_RB * XP * sizeof(bf16_t) + 16;
  static bool attr = false;
  if (!attr) {
    const void* ks[3] = {reinterpret_cast<const void*>(ffn_split_fwd_kernel<true>), reinterpret_cast<const void*>(ffn_split_fwd_kernel<false>),
                         reinterpret_cast<const void*>(ffn_split_bwd_kernel)};
    for (const void* k : ks) {
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { set_error("ffn_split: hipFuncSetAttribute(%zu B LDS) failed: %s", lds, hipGetErrorString(e)); return 1; }
    }
    attr = true;
  }
  const int grid = ((a.M + FS_RB - 1) / FS_RB) * FS_NQ;
  if (bwd) hipLaunchKernelGGL(ffn_split_bwd_kernel, dim3(grid), dim3(512), lds, st, a);
  else if (train) hipLaunchKernelGGL(ffn_split_fwd_kernel<true>, dim3(grid), dim3(512), lds, st, a);
  else hipLaunchKernelGGL(ffn_split_fwd_kernel<false>, dim3(grid), dim3(512), lds, st, a);
  return check_launch(bwd ? "ffn_split_bwd" : "ffn_split_fwd");
}

extern "C" int sedt_ffn_split_fwd(const void* x1n, const void* x1, const void* w1_frag, const float* b1, const void* w2_frag, const float* b2,
                                  void* h, void* x2, float* part, uint32_t* cnt, int M, int FF, float drop_p, uint32_t seed_h,
                                  uint32_t seed_f, const uint32_t* seed_ptr, void* stream) {
  SEDT_REQUIRE(x1n && x1 && w1_frag && b1 && w2_frag && b2 && x2 && part && cnt, "ffn_split_fwd: null pointer");
  SEDT_REQUIRE(M >= 1 && sedt_ffn_split_ok(FS_D, FF, SEDT_BF16), "ffn_split_fwd: FF = %d outside the envelope", FF);
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "ffn_split_fwd: drop_p out of range");
  FfnSplitArgs a{};
  a.xin = (const bf16_t*)x1n; a.res = (const bf16_t*)x1; a.wa = (const u32x4*)w1_frag; a.wb = (const u32x4*)w2_frag; a.b1 = b1; a.b2 = b2;
  a.h = (bf16_t*)h; a.out = (bf16_t*)x2; a.part = part; a.cnt = cnt; a.M = M; a.FF = FF; a.drop_p = drop_p;
  a.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  a.seed_h = seed_h; a.seed_f = seed_f; a.seed_ptr = seed_ptr;
  static const int dbg_env = dev_getenv("SEDT_SLAB_DBG") ? atoi(dev_getenv("SEDT_SLAB_DBG")) : 0;
  a.dbg = dbg_env;
  return ffn_split_launch(false, h != nullptr, a, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int sedt_ffn_split_bwd(const void* gx2, const void* h, const void* w2t_frag, const void* w1t_frag, void* g2, void* gh, void* g_x1n,
                                  float* part, uint32_t* cnt, int M, int FF, float drop_p, uint32_t seed_f, const uint32_t* seed_ptr,
                                  void* stream) {
  SEDT_REQUIRE(gx2 && h && w2t_frag && w1t_frag && gh && g_x1n && part && cnt, "ffn_split_bwd: null pointer");
  SEDT_REQUIRE(M >= 1 && sedt_ffn_split_ok(FS_D, FF, SEDT_BF16), "ffn_split_bwd: FF = %d outside the envelope", FF);
  SEDT_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || g2), "ffn_split_bwd: drop_p / g2");
  FfnSplitArgs a{};
  a.xin = (const bf16_t*)gx2; a.wa = (const u32x4*)w2t_frag; a.wb = (const u32x4*)w1t_frag; a.h = (bf16_t*)const_cast<void*>(h);
  a.out = (bf16_t*)g_x1n; a.aux = (bf16_t*)g2; a.gh = (bf16_t*)gh; a.part = part; a.cnt = cnt; a.M = M; a.FF = FF; a.drop_p = drop_p;
  a.thresh = drop_p > 0.f ? drop_threshold(drop_p) : 0u;
  a.seed_f = seed_f; a.seed_ptr = seed_ptr;
  return ffn_split_launch(true, true, a, reinterpret_cast<hipStream_t>(stream));
}
