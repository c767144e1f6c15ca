// lds_gemm_common.h - definitions shared by the LDS-DMA kernels (igemm3.hip, wgrad3.hip, wgrad4.hip)
#pragma once
#include "common.h"

namespace sedt {

constexpr int BK2 = 64;           // bf16 elements per K tile = 128 B per row
constexpr int ROWB = BK2 * 2;     // bytes per LDS row

struct Geom2 {
  int Hi, Wi, Ci, Ho, Wo, KH, KW, sh, sw, ph, pw, dh, dw, transposed;
};

__device__ __forceinline__ long gather_pix2(const Geom2& g, int n, int ho, int wo, int kh, int kw) {
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

typedef __attribute__((address_space(3))) void lds_void;

// Workgroup barrier that first retires this wave's outstanding LDS operations.  A bare s_barrier orders nothing in LDS: the
// compiler may schedule it ABOVE the lgkmcnt wait of the last fragment reads (their consumers are MFMAs, not memory
// operations), and then another wave that passed the barrier can overwrite the buffer - the f32 staging tile of the epilogue
// aliases the ring - before those reads have executed.  Seen on the single-stage K = 64 kernel: sporadic wrong elements, found
// by the bit-reproducibility test of the full-size step.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

}  // namespace sedt
