// criterion.hip - the device half of SetCriterion (reference sedt/sedt.py:161-283) as ONE kernel.
//
// The host half (SetCriterion.prepare: cost matrices, one D2H copy, batched Hungarian, one H2D copy) leaves dense,
// fixed-shape targets on the device: for every (decoder layer, strong clip, query) a target class, a CE weight, a box
// weight (0 = unmatched) and a target box.  Given those, every loss of the step - weighted cross-entropy, L1 and GIoU
// on (centre, length) intervals, the audio-tag BCE, the cardinality / class-error logs - and ALL their gradients
// w.r.t. the model outputs are a few thousand independent rows: one 1024-thread workgroup computes everything, instead
// of ~190 elementwise launches of loss math + autograd.  The gradients are written per loss term and UNWEIGHTED
// (d loss_ce_d / d logits, d loss_bbox_d / d boxes, d loss_giou_d / d boxes, d loss_weak / d at); a second one-launch
// kernel combines them with whatever gradient arrives at the loss vector (the weighted total and/or single entries),
// so `sum(loss_dict[k] * weight_dict[k])` in a caller's own train loop differentiates exactly as with the reference.
#include "common.h"

namespace sedt {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

__global__ __launch_bounds__(1024) void set_criterion_kernel(const SedtCriterion a) {
  __shared__ float red[16][SEDT_CRIT_MAXOUT];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int L = a.L, B = a.B, ns = a.ns, Q = a.Q, C1 = a.C + 1, C = a.C;
  const float nb = a.num_boxes[0];
  const float inv_nb = 1.f / nb;
  float acc[SEDT_CRIT_MAXOUT];
#pragma unroll
  for (int i = 0; i < SEDT_CRIT_MAXOUT; ++i) acc[i] = 0.f;
  // output slots: [4*d + 0..3] = ce, bbox, giou, cardinality of dense layer d; then class_error hits, matched count, weak
  const int SLOT_HIT = 4 * L, SLOT_CNT = 4 * L + 1, SLOT_WEAK = 4 * L + 2;

  // ---------------- classification + boxes: one row = (dense layer d, clip b, query q)
  const int nrows = L * B * Q;
  for (int r = t; r < nrows; r += 1024) {
    const int q = r % Q, b = (r / Q) % B, d = r / (Q * B);
    const int ml = a.layer_of[d];                                 // which slice of the model's stacked outputs
    const float* x = a.logits + (((long)ml * B + b) * Q + q) * C1;
    float* gx = a.dlogits + (((long)ml * B + b) * Q + q) * C1;
    float* gbx = a.dboxes + (((long)ml * B + b) * Q + q) * 2;
    float* gbx2 = a.dboxes2 + (((long)ml * B + b) * Q + q) * 2;
    if (b >= ns) {                                                // not strongly labelled: no CE / box loss, zero grads
      for (int c = 0; c < C1; ++c) gx[c] = 0.f;
      gbx[0] = 0.f; gbx[1] = 0.f;
      gbx2[0] = 0.f; gbx2[1] = 0.f;
      continue;
    }
    const long di = ((long)d * ns + b) * Q + q;
    const int tc = (int)a.tc[di];
    const float coef = a.coef[di], wb = a.wbox[di];
    float m = -INFINITY;
    int amax = 0;
    for (int c = 0; c < C1; ++c)
      if (x[c] > m) { m = x[c]; amax = c; }
    float se = 0.f;
    for (int c = 0; c < C1; ++c) se += __expf(x[c] - m);
    const float lse = m + __logf(se);
    const float w = a.empty_weight[tc];
    acc[4 * d + 0] += w * (lse - x[tc]) * coef * inv_nb;
    const float gscale = coef * w * inv_nb;
    for (int c = 0; c < C1; ++c) gx[c] = gscale * (__expf(x[c] - lse) - (c == tc ? 1.f : 0.f));
    if (d == 0 && wb > 0.f) {
      acc[SLOT_CNT] += 1.f;
      if (amax == tc) acc[SLOT_HIT] += 1.f;
    }
    // boxes (centre, length) -> interval [s, e]
    const float* bx = a.boxes + (((long)ml * B + b) * Q + q) * 2;
    float gc = 0.f, gl = 0.f, gc2 = 0.f, gl2 = 0.f;
    if (wb > 0.f) {
      const float s1 = bx[0] - 0.5f * bx[1], e1 = bx[0] + 0.5f * bx[1];
      const float tcn = a.tbox[2 * di], tln = a.tbox[2 * di + 1];
      const float s2 = tcn - 0.5f * tln, e2 = tcn + 0.5f * tln;
      // L1 on the fake boxes [s,0,e,1]: |s1-s2| + |e1-e2|
      acc[4 * d + 1] += (fabsf(s1 - s2) + fabsf(e1 - e2)) * wb * inv_nb;
      float gs = wb * inv_nb * sgn(s1 - s2), ge = wb * inv_nb * sgn(e1 - e2);
      gc = gs + ge;
      gl = 0.5f * (ge - gs);
      // GIoU
      const float lo = fmaxf(s1, s2), hi = fminf(e1, e2);
      const float inter = fmaxf(hi - lo, 0.f);
      const float di_e = (hi - lo > 0.f && e1 < e2) ? 1.f : 0.f;     // d inter / d e1
      const float di_s = (hi - lo > 0.f && s1 > s2) ? -1.f : 0.f;    // d inter / d s1
      const float uni = (e1 - s1) + (e2 - s2) - inter;
      const float du_e = 1.f - di_e, du_s = -1.f - di_s;
      const float hull = fmaxf(fmaxf(e1, e2) - fminf(s1, s2), 0.f);
      const float dh_e = e1 > e2 ? 1.f : 0.f, dh_s = s1 < s2 ? -1.f : 0.f;
      const float giou = inter / uni - (hull - uni) / hull;
      acc[4 * d + 2] += (1.f - giou) * wb * inv_nb;
      // d giou = d(inter/uni) + d(uni/hull)
      const float dg_e = (di_e * uni - inter * du_e) / (uni * uni) + (du_e * hull - uni * dh_e) / (hull * hull);
      const float dg_s = (di_s * uni - inter * du_s) / (uni * uni) + (du_s * hull - uni * dh_s) / (hull * hull);
      const float k = -wb * inv_nb;
      gc2 = k * (dg_s + dg_e);
      gl2 = 0.5f * k * (dg_e - dg_s);
    }
    gbx[0] = gc;
    gbx[1] = gl;
    gbx2[0] = gc2;
    gbx2[1] = gl2;
  }

  // ---------------- cardinality: one (dense layer, clip) per thread iteration, over ALL clips
  for (int r = t; r < L * B; r += 1024) {
    const int b = r % B, d = r / B;
    const int ml = a.layer_of[d];
    int cnt = 0;
    for (int q = 0; q < Q; ++q) {
      const float* x = a.logits + (((long)ml * B + b) * Q + q) * C1;
      float m = x[0];
      int am = 0;
      for (int c = 1; c < C1; ++c)
        if (x[c] > m) { m = x[c]; am = c; }
      cnt += am != C;
    }
    acc[4 * d + 3] += fabsf((float)cnt - a.tgt_len[b]) / (float)B;
  }

  // ---------------- audio-tag BCE (mean over n_lab x C), torch semantics: log clamped at -100, grad denominator >= 1e-12
  if (a.at) {
    const int n = a.n_lab * C;
    for (int r = t; r < a.Bat * C; r += 1024) {
      float g = 0.f;
      if (r < n) {
        const float p = a.at[r], y = a.gt_weak[r];
        acc[SLOT_WEAK] += -(y * fmaxf(__logf(p), -100.f) + (1.f - y) * fmaxf(__logf(1.f - p), -100.f)) / (float)n;
        g = (p - y) / fmaxf(p * (1.f - p), 1e-12f) / (float)n;
      }
      a.dat[r] = g;
    }
  }

  // ---------------- block reduction of the scalar outputs
  const int nout = 4 * L + 3;
  for (int i = 0; i < nout; ++i) {
    const float v = wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (t < nout) {
    float v = 0.f;
    for (int w = 0; w < 16; ++w) v += red[w][t];
    a.out[t] = v;
  }
  __syncthreads();
  if (t == 0) {
    float total = 0.f;
    for (int d = 0; d < L; ++d)
      total += a.w_ce[d] * a.out[4 * d] + a.w_bbox[d] * a.out[4 * d + 1] + a.w_giou[d] * a.out[4 * d + 2];
    if (a.at) total += a.w_weak * a.out[SLOT_WEAK];
    a.out[nout] = total;                                                               // weighted total
    a.out[nout + 1] = 100.f - 100.f * a.out[SLOT_HIT] / fmaxf(a.out[SLOT_CNT], 1.f);   // class_error
  }
}

// grads of the model outputs from the gradient g[4L+5] that reached the loss vector
__global__ __launch_bounds__(256) void set_criterion_bwd_kernel(const SedtCriterion a, const float* __restrict__ g,
                                                                float* __restrict__ glogits, float* __restrict__ gboxes,
                                                                float* __restrict__ gat) {
  const int L = a.L, B = a.B, Q = a.Q, C1 = a.C + 1;
  const float gtot = g[4 * L + 3];
  const int nrows = L * B * Q;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) {
    const int d = r / (Q * B);
    const int ml = a.layer_of[d];
    const long row = (long)ml * B * Q + (r - d * Q * B);
    const float kce = g[4 * d] + gtot * a.w_ce[d], kl1 = g[4 * d + 1] + gtot * a.w_bbox[d],
                kgi = g[4 * d + 2] + gtot * a.w_giou[d];
    for (int c = 0; c < C1; ++c) glogits[row * C1 + c] = kce * a.dlogits[row * C1 + c];
    gboxes[row * 2] = kl1 * a.dboxes[row * 2] + kgi * a.dboxes2[row * 2];
    gboxes[row * 2 + 1] = kl1 * a.dboxes[row * 2 + 1] + kgi * a.dboxes2[row * 2 + 1];
  }
  if (gat) {
    const float kw = g[4 * L + 2] + gtot * a.w_weak;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < a.Bat * a.C; r += gridDim.x * 256) gat[r] = kw * a.dat[r];
  }
}

}  // namespace sedt

extern "C" int sedt_set_criterion(const SedtCriterion* args, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr, "set_criterion: null args");
  const SedtCriterion& a = *args;
  SEDT_REQUIRE(a.L >= 1 && a.L <= SEDT_CRIT_MAXL && a.C >= 1 && a.C <= 63, "set_criterion: L=%d (1..%d), C=%d", a.L, SEDT_CRIT_MAXL, a.C);
  SEDT_REQUIRE(a.logits && a.boxes && a.dlogits && a.dboxes && a.dboxes2 && a.tc && a.coef && a.wbox && a.tbox && a.tgt_len && a.num_boxes &&
                   a.empty_weight && a.out,
               "set_criterion: null pointer");
  SEDT_REQUIRE((a.at == nullptr) == (a.dat == nullptr), "set_criterion: at and dat go together");
  hipLaunchKernelGGL(set_criterion_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("set_criterion");
}

extern "C" int sedt_set_criterion_bwd(const SedtCriterion* args, const float* g, float* glogits, float* gboxes, float* gat,
                                      void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr && g && glogits && gboxes, "set_criterion_bwd: null pointer");
  const SedtCriterion& a = *args;
  SEDT_REQUIRE(a.L >= 1 && a.L <= SEDT_CRIT_MAXL, "set_criterion_bwd: L=%d", a.L);
  SEDT_REQUIRE(a.dlogits && a.dboxes && a.dboxes2 && ((gat == nullptr) || a.dat), "set_criterion_bwd: null gradient buffers");
  const int rows = a.L * a.B * a.Q;
  hipLaunchKernelGGL(set_criterion_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, g,
                     glogits, gboxes, gat);
  return check_launch("set_criterion_bwd");
}
