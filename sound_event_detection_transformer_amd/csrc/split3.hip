// split3.hip - operand preparation of the fast bf16x3 mode ("parity-grade" GEMMs at bf16 MFMA rate).
//
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) carries 16 mantissa bits of an f32 value; a product x * w is then
// hi_x hi_w + lo_x hi_w + hi_x lo_w to ~2^-16 relative (the lo lo term is dropped).  Instead of splitting fragment by fragment inside
// the GEMM main loop (csrc/igemm.hip's X3 form: ~350 VALU per 6 MFMAs, issue-bound at the f32 mode's pace), the split is done ONCE per
// operand here and the three partial products become ONE plain bf16 GEMM with a three times longer contraction:
//     activations   [rows][C]  f32  ->  [rows][2C] bf16 = [ hi | lo ]               (the bytes of the f32 tensor)
//     weights       [rows][C]  f32  ->  [rows][3C] bf16 = [ hi | hi | lo ]          (rows = Cout * taps, C = Cin: per-tap runs)
// and the GEMM walks the activation's channels as hi, lo, hi (SedtIgemm.awrap: the last third of the walk wraps back onto hi), so that
// sum_k a'[k] w'[k] over 3C = hi hi + lo hi + hi lo.  The LDS-DMA kernels (igemm3 / wgrad3 / wgrad4) take these operands otherwise
// unchanged - a convolution simply walks Ci' = 3 Ci channels per pixel - and accumulate in f32; their epilogue writes f32 (SedtIgemm.f32ep).
// A weight gradient dW = dY^T X needs hi hi + lo hi + hi lo over the PIXEL axis: two problems on column views of the same images,
// dY'[:, 0:2Co] x X'[:, 0:Ci] (rows 0..Co-1 = hi hi, rows Co..2Co-1 = lo hi) and dY'[:, 0:Co] x X'[:, Ci:2Ci] (hi lo), whose slabs the
// split-K reduction adds (ops.wgrad).
// HBM-bound: 4 bytes read, 4 (activations) or 6 (weights) written per element.
#include "common.h"

namespace sedt {

constexpr int SPLIT_MAXJ = 8;
struct SplitJobs {
  int n;
  SedtSplitJob j[SPLIT_MAXJ];
};

__global__ __launch_bounds__(256) void split3_kernel(const SplitJobs jobs) {
  int i = 0;
  while (i + 1 < jobs.n && (int)blockIdx.x >= jobs.j[i + 1].blk0) ++i;
  const SedtSplitJob& J = jobs.j[i];
  const long q = ((long)(blockIdx.x - J.blk0) * 256 + threadIdx.x) * 4;      // first of this thread's four elements
  if (q >= (long)J.rows * J.cols) return;
  const long row = q / J.cols;
  const int c = (int)(q - row * J.cols);
  const float4 x = *reinterpret_cast<const float4*>(J.src + row * J.ld + c);
  const float xv[4] = {x.x, x.y, x.z, x.w};
  VecT<bf16_t, 4> hi, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi.v[e] = (bf16_t)xv[e];
    lo.v[e] = (bf16_t)(xv[e] - (float)hi.v[e]);
  }
  if (J.pattern) {                                  // weight: [hi | hi | lo]
    bf16_t* d = reinterpret_cast<bf16_t*>(J.dst) + row * (3L * J.cols) + c;
    *reinterpret_cast<VecT<bf16_t, 4>*>(d) = hi;
    *reinterpret_cast<VecT<bf16_t, 4>*>(d + J.cols) = hi;
    *reinterpret_cast<VecT<bf16_t, 4>*>(d + 2L * J.cols) = lo;
  } else {                                          // activation / gradient: [hi | lo] (the GEMM's walk wraps back onto hi)
    bf16_t* d = reinterpret_cast<bf16_t*>(J.dst) + row * (2L * J.cols) + c;
    *reinterpret_cast<VecT<bf16_t, 4>*>(d) = hi;
    *reinterpret_cast<VecT<bf16_t, 4>*>(d + J.cols) = lo;
  }
}

}  // namespace sedt

using namespace sedt;

extern "C" int sedt_split3(const SedtSplitJob* jobs, int njobs, void* stream) {
  SEDT_REQUIRE(jobs && njobs >= 1 && njobs <= SPLIT_MAXJ, "split3: 1..%d jobs", SPLIT_MAXJ);
  SplitJobs a;
  a.n = njobs;
  int blk = 0;
  for (int i = 0; i < njobs; ++i) {
    const SedtSplitJob& j = jobs[i];
    SEDT_REQUIRE(j.src && j.dst && j.rows >= 1 && j.cols >= 4 && (j.cols & 3) == 0 && j.ld >= j.cols && (j.ld & 3) == 0,
                 "split3: job %d: rows %d cols %d ld %ld (cols, ld multiples of 4)", i, j.rows, j.cols, (long)j.ld);
    SEDT_REQUIRE(((reinterpret_cast<uintptr_t>(j.src) & 15) | (reinterpret_cast<uintptr_t>(j.dst) & 7)) == 0, "split3: job %d: alignment", i);
    SEDT_REQUIRE(j.pattern == 0 || j.pattern == 1, "split3: pattern 0 ([hi|lo], activations) or 1 ([hi|hi|lo], weights)");
    a.j[i] = j;
    a.j[i].blk0 = blk;
    blk += (int)(((long)j.rows * j.cols / 4 + 255) / 256);
  }
  hipLaunchKernelGGL(split3_kernel, dim3(blk), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("split3");
}
