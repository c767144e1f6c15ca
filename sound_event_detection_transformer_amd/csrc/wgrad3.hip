// wgrad3.hip - lean-issue version of the bf16 LDS-DMA weight-gradient GEMM (wgrad2.hip).
//
// Same data path as wgrad2 ([pixel][channel] tiles by LDS-DMA into an XOR-swizzled 2-stage ring, k-contiguous MFMA
// fragments by ds_read_b64_tr_b16, split-K slabs), with the per-K-tile instruction stream stripped the way igemm3 does it:
//   * every DMA lane keeps ONE running 32-bit byte offset; a K tile (64 pixels) advances it by a wave-uniform step, and
//     a conv row wrap (ho >= Ho -> next image) adds a second uniform constant.  Requires 64 % Wo == 0 and Ho*Wo >= 64, so
//     a lane's wo - and with it the horizontal tap validity - never changes; vertical validity is one unsigned compare;
//   * the 4 x 4 transposing-read offsets are per-thread constants, the ring is unrolled so the stage base is an immediate;
//   * optional 64x128 output tile: one dY fragment feeds two MFMAs.
// Problems outside the envelope fall back to wgrad2 (generic gather) and then to the register-staged kernel.
#include <stdlib.h>
#include <algorithm>
#include "wgrad3_body.h"

namespace sedt {

template <int BN>
__global__ __launch_bounds__(256) void wgrad3_kernel(const SedtIgemm p, const unsigned a_bytes, const unsigned b_bytes,
                                                     const int nmajor) {
  wgrad3_body<BN>(p, a_bytes, b_bytes, nmajor, blockIdx.x, blockIdx.y);
}

__global__ __launch_bounds__(256) void wgrad3_group_kernel(const WgradGroup g) { wgrad_group_run(g, blockIdx.x); }

template <int BN>
static int launch_wgrad3(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  SEDT_DESCRIBE("wgrad3_kernel<%d>", BN);
  constexpr size_t lds = (size_t)2 * (64 * ROWB + 64 * BN * 2);
  static bool attr_set = false;
  auto kern = wgrad3_kernel<BN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      set_error("wgrad3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  const int nwg = ((p.N + BN - 1) / BN) * ((p.M + 63) / 64);
  static int force = -2;
  if (force == -2) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD_NMAJOR");
    force = e ? atoi(e) : -1;
  }
  const int nmajor = force >= 0 ? force : (p.N > p.M ? 1 : 0);
  hipLaunchKernelGGL(kern, dim3(nwg, p.splitk > 1 ? p.splitk : 1), dim3(256), lds, st, p, a_bytes, b_bytes, nmajor);
  return check_launch("wgrad3");
}

static bool wgrad3_conv_ok(const SedtIgemm& p) {
  return !(p.conv && ((64 % p.Wo) != 0 || p.Ho * p.Wo < 64 || p.Ho < 64 / p.Wo));
}

int wgrad_lds_envelope(const SedtIgemm& p, long* a_bytes, long* b_bytes);   // below
bool wgrad4_ok(const SedtIgemm& p);                                        // wgrad4.hip: 128x128 ping-pong kernel
int launch_wgrad4(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st);
int launch_wgrad4_group(WgradGroup& g, hipStream_t st);

// 0 = launched, -1 = some problem is outside the lean kernel's envelope (caller launches them one by one)
int wgrad3_group_try(const SedtIgemm* jobs, int njobs, hipStream_t st) {
  static int on = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD_GROUP");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  if (!on || njobs < 1) return -1;
  long ab[WG_MAXG], bb[WG_MAXG];
  for (int base = 0; base < njobs; base += WG_MAXG) {       // validate everything before launching anything
    const int n = std::min(WG_MAXG, njobs - base);
    for (int i = 0; i < n; ++i) {
      const SedtIgemm& p = jobs[base + i];
      if (!p.trans || wgrad_lds_envelope(p, &ab[i], &bb[i]) != 0 || !wgrad3_conv_ok(p)) return -1;
    }
  }
  static bool attr_set = false;
  constexpr size_t lds = (size_t)2 * (64 * ROWB + 64 * 64 * 2);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad3_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) {
      set_error("wgrad3 group: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 1;
    }
    attr_set = true;
  }
  for (int base = 0; base < njobs; base += WG_MAXG) {
    WgradGroup g, gw;                 // the 64x64 program, and the large problems that take the 128x128 ping-pong kernel
    g.n = gw.n = 0;
    const int n = std::min(WG_MAXG, njobs - base);
    int blk = 0;
    for (int j = 0; j < n; ++j) {
      const SedtIgemm& p = jobs[base + j];
      long a, b;
      wgrad_lds_envelope(p, &a, &b);
      if (wgrad4_ok(p)) {
        const int i = gw.n++;
        gw.p[i] = p;
        gw.a_bytes[i] = (unsigned)a;
        gw.b_bytes[i] = (unsigned)b;
        continue;
      }
      const int i = g.n++;
      g.p[i] = p;
      g.a_bytes[i] = (unsigned)a;
      g.b_bytes[i] = (unsigned)b;
      g.nwg[i] = ((p.N + 63) / 64) * ((p.M + 63) / 64);
      g.nmajor[i] = (p.N > p.M ? 1 : 0) + ((p.splitk >= 8 && p.splitk % 8 == 0) ? 2 : 0);
      g.blk0[i] = blk;
      blk += (g.nwg[i] * (p.splitk > 1 ? p.splitk : 1) + 7) / 8 * 8;      // ranges start on multiples of 8 (XCD = id & 7)
    }
    if (gw.n > 0)
      if (int r = launch_wgrad4_group(gw, st)) return r;
    if (g.n > 0) {
      g.blk0[g.n] = blk;
      hipLaunchKernelGGL(wgrad3_group_kernel, dim3(blk), dim3(256), lds, st, g);
      if (int r = check_launch("wgrad3_group")) return r;
    }
  }
  return 0;
}

// fills g from up to WG_MAXG problems; -1 if one of them is outside the lean kernel's envelope
int wgrad3_group_build(const SedtIgemm* jobs, int njobs, WgradGroup* g) {
  if (njobs < 1 || njobs > WG_MAXG) return -1;
  g->n = njobs;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    const SedtIgemm& p = jobs[i];
    long ab, bb;
    if (!p.trans || wgrad_lds_envelope(p, &ab, &bb) != 0 || !wgrad3_conv_ok(p)) return -1;
    g->p[i] = p;
    g->a_bytes[i] = (unsigned)ab;
    g->b_bytes[i] = (unsigned)bb;
    g->nwg[i] = ((p.N + 63) / 64) * ((p.M + 63) / 64);
    g->nmajor[i] = p.N > p.M ? 1 : 0;            // (riders of a co-scheduled launch do not start on an XCD boundary: no K-slice map)
    g->blk0[i] = blk;
    blk += g->nwg[i] * (p.splitk > 1 ? p.splitk : 1);
  }
  g->blk0[njobs] = blk;
  return 0;
}

// -1 = outside the envelope (caller continues with wgrad2)
int wgrad3_try(const SedtIgemm& p, unsigned a_bytes, unsigned b_bytes, hipStream_t st) {
  static int on = -1, wide = -1;
  if (on < 0) {
    const char* e = sedt::dev_getenv("SEDT_WGRAD_V3");
    on = (e && e[0] == '0') ? 0 : 1;
    const char* w = sedt::dev_getenv("SEDT_WGRAD_WIDE");
    wide = w ? atoi(w) : -1;
  }
  if (!on) return -1;
  if (!wgrad3_conv_ok(p)) return -1;
  if (wgrad4_ok(p)) return launch_wgrad4(p, a_bytes, b_bytes, st);
  int bn = 64;
  if (wide == 1) {   // measured on the full step: the wide tile does not pay (fewer, longer workgroups); opt-in only
    // a 16-byte chunk never straddles a tap (Ci % 8 == 0), so the wide tile needs nothing beyond N % 128 == 0
    if ((p.N % 128) == 0) bn = 128;
  }
  if (wide == 0) bn = 64;
  return bn == 128 ? launch_wgrad3<128>(p, a_bytes, b_bytes, st) : launch_wgrad3<64>(p, a_bytes, b_bytes, st);
}

// 0 when the problem fits the LDS-DMA weight-gradient kernels (wgrad3 / wgrad4); fills the buffer-descriptor sizes
int wgrad_lds_envelope(const SedtIgemm& p, long* a_bytes_out, long* b_bytes_out) {
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
int wgrad_lds_try(const SedtIgemm& p, hipStream_t st) {
  long a_bytes, b_bytes;
  if (wgrad_lds_envelope(p, &a_bytes, &b_bytes) != 0) return -1;
  {   // the lean-issue kernel takes the common cases
    int r3 = wgrad3_try(p, (unsigned)a_bytes, (unsigned)b_bytes, st);
    if (r3 >= 0) return r3;
  }
  return -1;
}

}  // namespace sedt
