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
#include <algorithm>
#include "common.h"

namespace sedt {

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }

// deterministic sum of one value per thread over the 1024-thread workgroup (result valid in every thread)
__device__ __forceinline__ float block_sum_1024(float v, float* red /* [17] */) {
  const int t = threadIdx.x;
  v = wave_sum(v);
  __syncthreads();                 // red may still be read from the previous call
  if ((t & 63) == 0) red[t >> 6] = v;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    red[16] = s;
  }
  __syncthreads();
  return red[16];
}

__global__ __launch_bounds__(1024) void set_criterion_kernel(const SedtCriterion a, float* __restrict__ rowvals) {
  // Phases (ONE workgroup; the whole problem is a few thousand rows, so what counts is the number of dependent steps, not the
  // width): (1) every (dense layer, clip, query) row - all layers at once - computes its loss terms and per-term gradients and
  // leaves {ce, l1, giou, hit, matched} in rowvals[row][5]; (2) one wave per (layer, quantity) adds its column of rowvals in a
  // fixed order; (3) the audio-tag BCE and the totals.  (The first version looped over the layers with five block-wide
  // reductions each: 38 us at L = 3, three times what the arithmetic needs.)
  __shared__ float red[17];
  __shared__ int card[SEDT_CRIT_MAXCARD];      // predicted-event count of every (dense layer, clip): integer LDS atomics
  __shared__ float sums[SEDT_CRIT_MAXL][6];    // per dense layer: ce, l1, giou, hits, matched, cardinality error
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int L = a.L, B = a.B, ns = a.ns, Q = a.Q, C1 = a.C + 1, C = a.C;
  // strong / labelled clip counts as DATA (mix-up moves clips across the strong | weak boundary per batch): a.ns / a.n_lab
  // stay the strides and capacities of the dense tables
  const int ns_eff = a.split ? min(a.split[0], ns) : ns;
  const int n_lab_eff = a.split ? min(a.split[1], a.n_lab) : a.n_lab;
  float nb;
  if (a.num_boxes) nb = a.num_boxes[0];
  else {                                        // num_boxes = sum of the box weights of the final layer (sedt.py:330)
    float sv = 0.f;
    for (int i = t; i < ns * Q; i += 1024) sv += a.wbox[i];
    nb = block_sum_1024(sv, red);
  }
  const float inv_nb = 1.f / nb;
  // output slots: [4*d + 0..3] = ce, bbox, giou, cardinality of dense layer d; then class_error hits, matched count, weak
  const int SLOT_HIT = 4 * L, SLOT_CNT = 4 * L + 1, SLOT_WEAK = 4 * L + 2;
  for (int i = t; i < L * B; i += 1024) card[i] = 0;
  __syncthreads();

  // ---------------- phase 1: all rows of all dense layers
  const int BQ = B * Q;
  for (int r = t; r < L * BQ; r += 1024) {
    const int d = r / BQ, rr = r - d * BQ;
    const int ml = a.layer_of[d];               // which slice of the model's stacked outputs
    const int b = rr / Q, q = rr - b * Q;
    const long mrow = ((long)ml * B + b) * Q + q;                       // row of the (compact) per-term gradient buffers
    const long xrow = ((long)ml * B + b) * a.Qs + a.q0 + q;             // row of the model's head outputs (Qs queries per clip)
    const float* x = a.logits + xrow * C1;
    float* gx = a.dlogits + mrow * C1;
    float* gbx = a.dboxes + mrow * 2;
    float* gbx2 = a.dboxes2 + mrow * 2;
    float* rv = rowvals + (long)r * 5;
    // the row's logits: all loads issued before the first use (a `for c < C1` loop with the load inside serialises C1 global
    // latencies per pass - two thirds of this kernel's time in its first form); C + 1 <= 16 in every configuration, wider rows
    // take the loop
    constexpr int XR = 16;
    float xr[XR];
    const bool small = C1 <= XR;
    if (small) {
#pragma unroll
      for (int c = 0; c < XR; ++c) xr[c] = c < C1 ? x[c] : -INFINITY;
    }
    float m = -INFINITY;
    int amax = 0;
    if (small) {
#pragma unroll
      for (int c = 0; c < XR; ++c)
        if (xr[c] > m) { m = xr[c]; amax = c; }
    } else {
      for (int c = 0; c < C1; ++c)
        if (x[c] > m) { m = x[c]; amax = c; }
    }
    if (amax != C) atomicAdd(&card[d * B + b], 1);
    if (b >= ns_eff) {                        // not strongly labelled: no CE / box loss, zero grads
      for (int c = 0; c < C1; ++c) gx[c] = 0.f;
      gbx[0] = 0.f; gbx[1] = 0.f;
      gbx2[0] = 0.f; gbx2[1] = 0.f;
      rv[0] = 0.f; rv[1] = 0.f; rv[2] = 0.f; rv[3] = 0.f; rv[4] = 0.f;
      continue;
    }
    const long di = ((long)d * ns + b) * Q + q;
    const int tc = (int)a.tc[di];
    const float coef = a.coef[di], wb = a.wbox[di];
    float l_ce = 0.f, l_bb = 0.f, l_gi = 0.f, n_hit = 0.f, n_cnt = 0.f;
    if (a.fl) {
      // sigmoid focal loss over the C+1 logits, one-hot target at tc (sedt.py:211-218, 412-422)
      float row = 0.f;
      const float ks = coef * inv_nb;
      for (int c = 0; c < C1; ++c) {
        const float xv = x[c], p = 1.f / (1.f + __expf(-xv));
        const float sp_pos = fmaxf(xv, 0.f) + log1pf(__expf(-fabsf(xv)));     // softplus(x)  = -log(1 - p)
        const float sp_neg = sp_pos - xv;                                     // softplus(-x) = -log p
        float ce, mod, dce, dmod, at;
        if (c == tc) {
          const float w = a.empty_weight[c];
          ce = w * sp_neg; dce = -w * (1.f - p); mod = 1.f - p; dmod = -p * (1.f - p); at = a.alpha_fl;
        } else {
          ce = sp_pos; dce = p; mod = p; dmod = p * (1.f - p); at = 1.f - a.alpha_fl;
        }
        if (a.alpha_fl < 0.f) at = 1.f;
        const float mg = a.gamma_fl == 1.f ? mod : powf(mod, a.gamma_fl);
        const float dmg = a.gamma_fl == 1.f ? 1.f : a.gamma_fl * powf(mod, a.gamma_fl - 1.f);
        row += at * ce * mg;
        gx[c] = ks * at * (dce * mg + ce * dmg * dmod);
      }
      l_ce = row * ks;
    } else if (small) {
      float se = 0.f, xt = 0.f;
#pragma unroll
      for (int c = 0; c < XR; ++c) {
        se += __expf(xr[c] - m);               // (-inf pads add 0)
        xt = c == tc ? xr[c] : xt;
      }
      const float lse = m + __logf(se);
      const float w = a.empty_weight[tc];
      l_ce = w * (lse - xt) * coef * inv_nb;
      const float gscale = coef * w * inv_nb;
#pragma unroll
      for (int c = 0; c < XR; ++c)
        if (c < C1) gx[c] = gscale * (__expf(xr[c] - lse) - (c == tc ? 1.f : 0.f));
    } else {
      float se = 0.f;
      for (int c = 0; c < C1; ++c) se += __expf(x[c] - m);
      const float lse = m + __logf(se);
      const float w = a.empty_weight[tc];
      l_ce = w * (lse - x[tc]) * coef * inv_nb;
      const float gscale = coef * w * inv_nb;
      for (int c = 0; c < C1; ++c) gx[c] = gscale * (__expf(x[c] - lse) - (c == tc ? 1.f : 0.f));
    }
    if (d == 0 && wb > 0.f) {
      n_cnt = 1.f;
      if (amax == tc) n_hit = 1.f;
    }
    // boxes (centre, length) -> interval [s, e]
    const float* bx = a.boxes + xrow * 2;
    float gc = 0.f, gl = 0.f, gc2 = 0.f, gl2 = 0.f;
    if (wb > 0.f) {
      const float s1 = bx[0] - 0.5f * bx[1], e1 = bx[0] + 0.5f * bx[1];
      const float tcn = a.tbox[2 * di], tln = a.tbox[2 * di + 1];
      const float s2 = tcn - 0.5f * tln, e2 = tcn + 0.5f * tln;
      // L1 on the fake boxes [s,0,e,1]: |s1-s2| + |e1-e2|
      l_bb = (fabsf(s1 - s2) + fabsf(e1 - e2)) * wb * inv_nb;
      const float gs = wb * inv_nb * sgn(s1 - s2), ge = wb * inv_nb * sgn(e1 - e2);
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
      l_gi = (1.f - giou) * wb * inv_nb;
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
    rv[0] = l_ce; rv[1] = l_bb; rv[2] = l_gi; rv[3] = n_hit; rv[4] = n_cnt;
  }
  __syncthreads();                              // (block scope: the rowvals written above are visible to every wave)

  // ---------------- phase 2: one wave per (dense layer, quantity) column; fixed summation order
  for (int job = wave; job < L * 6; job += 16) {
    const int d = job / 6, k = job - d * 6;
    float v = 0.f;
    if (k < 5) {
      const float* col = rowvals + (long)d * BQ * 5 + k;
      for (int i = lane; i < BQ; i += 64) v += col[(long)i * 5];
    } else {                                    // |#predicted events - #targets| averaged over ALL clips
      for (int bb = lane; bb < B; bb += 64) v += fabsf((float)card[d * B + bb] - a.tgt_len[bb]) / (float)B;
    }
    v = wave_sum(v);
    if (lane == 0) sums[d][k] = v;
  }
  __syncthreads();
  float total = 0.f;
  for (int d = 0; d < L; ++d) total += a.w_ce[d] * sums[d][0] + a.w_bbox[d] * sums[d][1] + a.w_giou[d] * sums[d][2];
  if (t < L) {
    a.out[4 * t] = sums[t][0];
    a.out[4 * t + 1] = sums[t][1];
    a.out[4 * t + 2] = sums[t][2];
    a.out[4 * t + 3] = sums[t][5];
  }
  if (t == 0) {
    const float n_hit = sums[0][3], n_cnt = sums[0][4];
    a.out[SLOT_HIT] = n_hit;
    a.out[SLOT_CNT] = n_cnt;
    a.out[4 * L + 4] = 100.f - 100.f * n_hit / fmaxf(n_cnt, 1.f);     // class_error
  }

  // ---------------- audio-tag BCE (mean over n_lab x C), torch semantics: log clamped at -100, grad denominator >= 1e-12
  float weak = 0.f;
  if (a.at) {
    const int n = n_lab_eff * C;
    for (int r = t; r < a.Bat * C; r += 1024) {
      float g = 0.f;
      if (r < n) {
        const float p = a.at[r], y = a.gt_weak[r];
        const float ce = -(y * fmaxf(__logf(p), -100.f) + (1.f - y) * fmaxf(__logf(1.f - p), -100.f));
        const float dce = (p - y) / fmaxf(p * (1.f - p), 1e-12f);
        if (a.fl) {                             // weak_focal_loss (sedt.py:425-433): sum over classes, mean over clips
          const float mod = 1.f - (p * y + (1.f - p) * (1.f - y)), dmod = 1.f - 2.f * y;
          const float at = a.alpha_fl < 0.f ? 1.f : a.alpha_fl * y + (1.f - a.alpha_fl) * (1.f - y);
          const float mg = a.gamma_fl == 1.f ? mod : powf(mod, a.gamma_fl);
          const float dmg = a.gamma_fl == 1.f ? 1.f : a.gamma_fl * powf(mod, a.gamma_fl - 1.f);
          weak += at * ce * mg / (float)n_lab_eff;
          g = at * (dce * mg + ce * dmg * dmod) / (float)n_lab_eff;
        } else {
          weak += ce / (float)n;
          g = dce / (float)n;
        }
      }
      a.dat[r] = g;
    }
    weak = block_sum_1024(weak, red);
    total += a.w_weak * weak;
  }
  // ---------------- BCE of the pooled clip-level probabilities (--pooling, sedt.py:182-185) on the weak clips; weak_mask None
  //                  (wp_all) = every labelled clip.  An empty range divides 0 by 0 like the reference's mean over nothing.
  float weak_p = 0.f;
  if (a.at_p) {
    const int r0 = a.wp_all ? 0 : ns_eff, r1 = n_lab_eff;
    const int n = max(r1 - r0, 0) * C;
    for (int r = t; r < a.Bp * C; r += 1024) {
      float g = 0.f;
      const int b = r / C;
      if (b >= r0 && b < r1) {
        const float p = a.at_p[r], y = a.gt_weak[r];
        weak_p += -(y * fmaxf(__logf(p), -100.f) + (1.f - y) * fmaxf(__logf(1.f - p), -100.f));
        g = (p - y) / fmaxf(p * (1.f - p), 1e-12f) / (float)n;
      }
      a.dat_p[r] = g;
    }
    weak_p = block_sum_1024(weak_p, red) / (float)n;
    total += a.w_weak_p * weak_p;
  }
  if (t == 0) {
    a.out[SLOT_WEAK] = weak;
    a.out[4 * L + 5] = weak_p;
    a.out[4 * L + 3] = total;                   // weighted total
    if (a.total) a.total[0] = total;
    if (a.nonfinite && !(fabsf(total) <= 3.0e38f)) *a.nonfinite = 1;      // NaN or inf (engine.py:70-73)
  }
}

// grads of the model outputs from the gradient g[4L+5] that reached the loss vector and / or the gradient gtotal[1] that reached the
// separately returned total (either may be null).  glogits / gboxes have the layout of the head outputs ([L][B][Qs] rows): the
// rows of queries outside [q0, q0 + Q) - the audio-tag query of dec_at models - get zeros.
__global__ __launch_bounds__(256) void set_criterion_bwd_kernel(const SedtCriterion a, const float* __restrict__ g,
                                                                const float* __restrict__ gtotal, float* __restrict__ glogits,
                                                                float* __restrict__ gboxes, float* __restrict__ gat,
                                                                float* __restrict__ gat_p) {
  const int L = a.L, B = a.B, Q = a.Q, Qs = a.Qs, C1 = a.C + 1;
  const float gtot = (g ? g[4 * L + 3] : 0.f) + (gtotal ? gtotal[0] : 0.f);
  const int nrows = L * B * Qs;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < nrows; r += gridDim.x * 256) {
    const int d = r / (Qs * B);
    const int rem = r - d * Qs * B, b = rem / Qs, qs = rem - b * Qs, q = qs - a.q0;
    const int ml = a.layer_of[d];
    const long xrow = ((long)ml * B + b) * Qs + qs;
    if (q < 0 || q >= Q) {
      for (int c = 0; c < C1; ++c) glogits[xrow * C1 + c] = 0.f;
      gboxes[xrow * 2] = 0.f;
      gboxes[xrow * 2 + 1] = 0.f;
      continue;
    }
    const long row = ((long)ml * B + b) * Q + q;
    const float kce = (g ? g[4 * d] : 0.f) + gtot * a.w_ce[d], kl1 = (g ? g[4 * d + 1] : 0.f) + gtot * a.w_bbox[d],
                kgi = (g ? g[4 * d + 2] : 0.f) + gtot * a.w_giou[d];
    for (int c = 0; c < C1; ++c) glogits[xrow * C1 + c] = kce * a.dlogits[row * C1 + c];
    gboxes[xrow * 2] = kl1 * a.dboxes[row * 2] + kgi * a.dboxes2[row * 2];
    gboxes[xrow * 2 + 1] = kl1 * a.dboxes[row * 2 + 1] + kgi * a.dboxes2[row * 2 + 1];
  }
  if (gat) {
    const float kw = (g ? g[4 * L + 2] : 0.f) + gtot * a.w_weak;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < a.Bat * a.C; r += gridDim.x * 256) gat[r] = kw * a.dat[r];
  }
  if (gat_p) {
    const float kw = (g ? g[4 * L + 5] : 0.f) + gtot * a.w_weak_p;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < a.Bp * a.C; r += gridDim.x * 256) gat_p[r] = kw * a.dat_p[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Device-side Hungarian matching (reference sedt/matcher.py:60-95 + the target bookkeeping of sedt.py:161-283).
// One wave per problem (decoder layer, strong clip): the cost matrix (Q queries x n targets, both <= 63) is built in
// LDS, then the shortest-augmenting-path algorithm with potentials runs with ONE LANE PER COLUMN: the column scan of
// each step is a single wave-wide min/argmin (ties -> lowest column, as the serial scan of csrc/host.cpp), potentials in
// double precision like scipy.  The assignment is turned into the dense targets the loss kernel reads, so a training
// step needs no device->host copy at all and the whole step can live in one HIP graph.
// ---- wave-uniform helpers of the assignment solver: everything below runs in ONE wave with one lane per column, and every
// index it broadcasts (the column j0 being expanded, its row i0) is the same in all lanes - so a value of "lane j0" is a
// v_readlane (a few cycles) rather than a cross-lane permute through the LDS crossbar, and the minimum over the lanes is a DPP
// prefix-minimum inside each row of 16 lanes plus four readlanes (the first versions - a 6-level butterfly of three permutes per
// level in double precision, then a serial scan over LDS - spent 35 us on the 192 small problems of a C2 step).
__device__ __forceinline__ int rl_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double rl_d(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <int CTRL>
__device__ __forceinline__ double dpp_min_step(double v) {     // min(v, v of the lane CTRL names); a lane without source keeps v
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return fmin(v, __hiloint2double(hi2, lo2));
}
__device__ __forceinline__ double wave_min_d(double v) {       // the minimum over all 64 lanes, in every lane
  v = dpp_min_step<0x111>(v);      // row_shr:1
  v = dpp_min_step<0x112>(v);      // row_shr:2
  v = dpp_min_step<0x114>(v);      // row_shr:4
  v = dpp_min_step<0x118>(v);      // row_shr:8  -> lane 15 of every row holds the row's minimum
  return fmin(fmin(rl_d(v, 15), rl_d(v, 31)), fmin(rl_d(v, 47), rl_d(v, 63)));
}

// rows (n) <= cols (m) <= 63; cost(i, j) for 0-based row i, column j.  Returns in lane j (1..m) the 1-based row assigned
// to column j (0 = none).  Every lane of the wave must call this.
template <typename F>
__device__ int wave_lsa(int n, int m, F cost) {
  const int lane = threadIdx.x & 63;
  const double INF = 1e300;
  double u = 0.0, v = 0.0;        // lane r: potential of row r (1-based); lane j: potential of column j
  int p = 0, way = 0;             // lane j: row matched to column j; predecessor column on the alternating path
  for (int i = 1; i <= n; ++i) {
    if (lane == 0) p = i;
    int j0 = 0, guard = 0;
    double minv = INF;
    bool used = false, in_rows = false;
    do {
      if (lane == j0) used = true;
      const int i0 = rl_i(p, j0);
      if (lane == i0) in_rows = true;
      const double u0 = rl_d(u, i0);
      const bool cand = lane >= 1 && lane <= m && !used;
      if (cand) {
        const double cur = (double)cost(i0 - 1, lane - 1) - u0 - v;
        if (cur < minv) { minv = cur; way = j0; }
      }
      // delta = min over candidate columns, j1 = lowest column attaining it (ties -> lowest column, as the serial scan of host.cpp)
      const double mine = cand ? minv : INF;
      const double delta = wave_min_d(mine);
      const unsigned long long at_min = __ballot(cand && mine == delta);
      const int j1 = at_min ? (int)__builtin_ctzll(at_min) : 64;
      if (in_rows) u += delta;
      if (used) v -= delta; else minv -= delta;
      j0 = j1;
    } while (j0 < 64 && rl_i(p, j0) != 0 && ++guard <= m + 1);
    if (j0 >= 64) j0 = 0;         // (unreachable with finite costs)
    guard = 0;
    do {                          // augment along the path
      const int j1 = rl_i(way, j0);
      const int pj1 = rl_i(p, j1);
      if (lane == j0) p = pj1;
      j0 = j1;
    } while (j0 && ++guard <= m + 1);
  }
  return p;
}

__global__ __launch_bounds__(64) void match_targets_kernel(const SedtMatch a) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  const int L = a.L, B = a.B, ns = a.ns, Q = a.Q, C = a.C, C1 = a.C + 1;
  const int ns_eff = a.split ? min(a.split[0], ns) : ns;          // (see set_criterion_kernel: the split as data)
  const int n_lab_eff = a.split ? min(a.split[1], a.n_lab) : a.n_lab;
  if ((int)blockIdx.x == L * ns) {
    // ---- bookkeeping block: per-clip target counts (cardinality) and the clip-level tag targets (sedt.py:199-209)
    float* row = lds + lane * (C + 1);          // this lane's tag row lives in LDS until it is complete
    for (int b = lane; b < B; b += 64) {
      const int o = a.lab_off[b], n = a.lab_off[b + 1] - o;
      a.tgt_len[b] = (float)n;
      if (a.gt_weak && b < a.n_lab) {
        for (int c = 0; c < C; ++c) row[c] = 0.f;
        if (b < n_lab_eff)
          for (int j = 0; j < n; ++j) row[a.lab_cat[o + j]] += a.ratio_cat ? a.ratio_cat[o + j] : 1.f;
        float* g = a.gt_weak + (long)b * C;
        for (int c = 0; c < C; ++c) g[c] = fminf(fmaxf(row[c], 0.f), 1.f);
      }
    }
    return;
  }
  const int d = blockIdx.x / ns, b = blockIdx.x % ns;
  const int ml = a.layer_of[d];
  if (b >= ns_eff) {                       // beyond this batch's strong part: "no target" rows, zero box weight
    if (lane < Q) {
      const long di = ((long)d * ns + b) * Q + lane;
      a.tc[di] = (float)C; a.coef[di] = 1.f; a.wbox[di] = 0.f; a.tbox[2 * di] = 0.5f; a.tbox[2 * di + 1] = 0.5f; a.tidx[di] = 0.f;
      if (a.assign) a.assign[di] = -1;
    }
    return;
  }
  const int bo = a.box_off[b], n = a.box_off[b + 1] - bo, lo = a.lab_off[b];
  float* prob = lds;                       // [Q][C1]: class part of the matching cost per (query, class)
  float* cst = lds + Q * C1;               // [Q][n]
  float* loc = cst + Q * a.max_targets;    // [Q][n] localisation-only cost (fine_tune)
  const bool main_ft = a.fine_tune && d == 0;
  if (lane < Q) {
    const float* x = a.logits + (((long)ml * B + b) * a.Qs + a.q0 + lane) * C1;
    if (a.fl) {                            // matcher.py:73-78: focal cost on sigmoid probabilities
      for (int c = 0; c < C1; ++c) {
        const float p = 1.f / (1.f + expf(-x[c]));
        const float neg = (1.f - a.alpha_fl) * powf(p, a.gamma_fl) * (-logf(1.f - p + 1e-8f));
        const float pos = a.alpha_fl * powf(1.f - p, a.gamma_fl) * (-logf(p + 1e-8f));
        prob[lane * C1 + c] = pos - neg;
      }
    } else if (C1 <= 16) {                 // -softmax probability; the row's logits loaded up front (see set_criterion_kernel)
      float xr[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) xr[c] = c < C1 ? x[c] : -INFINITY;
      float m = -INFINITY;
#pragma unroll
      for (int c = 0; c < 16; ++c) m = fmaxf(m, xr[c]);
      float se = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) se += expf(xr[c] - m);
#pragma unroll
      for (int c = 0; c < 16; ++c)
        if (c < C1) prob[lane * C1 + c] = -(expf(xr[c] - m) / se);
    } else {
      float m = -INFINITY;
      for (int c = 0; c < C1; ++c) m = fmaxf(m, x[c]);
      float se = 0.f;
      for (int c = 0; c < C1; ++c) se += expf(x[c] - m);
      for (int c = 0; c < C1; ++c) prob[lane * C1 + c] = -(expf(x[c] - m) / se);
    }
  }
  __syncthreads();
  for (int i = lane; i < Q * n; i += 64) {
    const int q = i / n, t = i % n;
    const float* bx = a.boxes + (((long)ml * B + b) * a.Qs + a.q0 + q) * 2;
    const float c1 = bx[0], l1 = bx[1], c2 = a.box_cat[2 * (bo + t)], l2 = a.box_cat[2 * (bo + t) + 1];
    const float s1 = c1 - l1 / 2, e1 = c1 + l1 / 2, s2 = c2 - l2 / 2, e2 = c2 + l2 / 2;
    const float cost_bbox = fabsf(s1 - s2) + fabsf(e1 - e2);
    const float inter = fmaxf(fminf(e1, e2) - fmaxf(s1, s2), 0.f);
    const float uni = (e1 - s1) + (e2 - s2) - inter;
    const float hull = fmaxf(fmaxf(e1, e2) - fminf(s1, s2), 0.f);
    const float giou = inter / uni - (hull - uni) / hull;
    const float cost_class = prob[q * C1 + (int)a.lab_cat[lo + t]];
    // a non-finite cost (diverged model outputs) must not stall the augmenting-path search: it becomes a huge finite cost,
    // the losses computed afterwards are NaN/inf anyway and raise the criterion's non-finite flag
    const float cv = a.w_bbox * cost_bbox + a.w_class * cost_class - a.w_giou * giou;
    cst[q * n + t] = fabsf(cv) <= 1e30f ? cv : 1e30f;
    if (main_ft) {
      const float lv = a.w_bbox * cost_bbox - a.w_giou * giou;
      loc[q * n + t] = fabsf(lv) <= 1e30f ? lv : 1e30f;
    }
  }
  __syncthreads();
  // assignment: target index matched to query `lane`, or -1
  int asg = -1;
  if (n > 0) {
    if (n <= Q) {          // every target gets a query: rows = targets, columns = queries
      const int p = wave_lsa(n, Q, [&](int t, int q) { return cst[q * n + t]; });
      if (lane >= 1 && lane <= Q && p > 0) {
        // lane j holds column j = query j-1; move the result to the lane of the query
        asg = p - 1;
      }
      asg = __shfl(asg, lane + 1 < 64 ? lane + 1 : 63, 64);
      if (lane >= Q) asg = -1;
    } else {               // more targets than queries: rows = queries, columns = targets
      const int p = wave_lsa(Q, n, [&](int q, int t) { return cst[q * n + t]; });
      // lane j (1..n) holds the query (1-based) matched to target j-1: scatter through LDS
      int* tmp = reinterpret_cast<int*>(lds);
      __syncthreads();
      if (lane < Q) tmp[lane] = -1;
      __syncthreads();
      if (lane >= 1 && lane <= n && p > 0) tmp[p - 1] = lane - 1;
      __syncthreads();
      if (lane < Q) asg = tmp[lane];
    }
  }
  // ---- coefficients (matcher.py:124-132) and the fine-tune re-matching (matcher.py:99-121; dense layer 0 only)
  float cf = 1.f;
  if (main_ft && n > 0) {
    float nc = INFINITY;
    int nt = 0;
    if (lane < Q)
      for (int t = 0; t < n; ++t) {
        const float v = loc[lane * n + t];
        if (v < nc) { nc = v; nt = t; }                       // first minimum, as torch.min on the CPU
      }
    const bool hung = lane < Q && asg >= 0;
    const bool close = lane < Q && nc < a.epsilon;
    const int n_gt = __popcll(__ballot(hung));
    const bool extra = close && !hung;
    const unsigned long long em = __ballot(extra);
    const int k = __popcll(em & ((1ull << lane) - 1ull));
    float u = 0.f;
    if (extra) {
      if (a.ft_rand) u = a.ft_rand[b * Q + k];
      else u = (float)(rng32(eff_seed(a.ft_seed, a.seed_ptr), (uint64_t)b * 64 + k) >> 8) * (1.f / 16777216.f);
    }
    const float keep_p = (float)((double)a.alpha * (double)n_gt / (double)Q);
    asg = (hung && close) ? asg : ((extra && !(u > keep_p)) ? nt : -1);
    if (a.normalize) {
      int cnt = 0;
      for (int j = 0; j < Q; ++j) cnt += (__shfl(asg, j, 64) == asg) ? 1 : 0;
      cf = 1.f / (float)max(cnt, 1);
    }
  } else if (!(a.normalize && d == 0) && a.ratio_cat) {       // (normalize reaches the final layer only: sedt.py:320 vs :340)
    // POSITIONAL: the k-th matched query of the clip (ascending query index) takes ratio[k]
    const unsigned long long mm = __ballot(lane < Q && asg >= 0);
    cf = a.ratio_cat[lo + __popcll(mm & ((1ull << lane) - 1ull))];
  }
  if (lane < Q) {
    const long di = ((long)d * ns + b) * Q + lane;
    const bool hit = asg >= 0;
    const int t = hit ? asg : 0;
    const float ratio = hit ? cf : 1.f;
    a.tc[di] = hit ? (float)a.lab_cat[lo + t] : (float)C;
    a.coef[di] = ratio;
    a.wbox[di] = hit ? ratio : 0.f;
    a.tbox[2 * di] = hit ? a.box_cat[2 * (bo + t)] : 0.5f;
    a.tbox[2 * di + 1] = hit ? a.box_cat[2 * (bo + t) + 1] : 0.5f;
    a.tidx[di] = (float)t;
    if (a.assign) a.assign[di] = asg;
  }
}

}  // namespace sedt

extern "C" size_t sedt_set_criterion_scratch(int L, int B, int Q) { return (size_t)L * B * Q * 5 * sizeof(float); }

extern "C" int sedt_set_criterion(const SedtCriterion* args, float* scratch, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr, "set_criterion: null args");
  const SedtCriterion& a = *args;
  SEDT_REQUIRE(a.L >= 1 && a.L <= SEDT_CRIT_MAXL && a.C >= 1 && a.C <= 63, "set_criterion: L=%d (1..%d), C=%d", a.L, SEDT_CRIT_MAXL, a.C);
  SEDT_REQUIRE(scratch != nullptr, "set_criterion: scratch (sedt_set_criterion_scratch bytes) is required");
  SEDT_REQUIRE(a.logits && a.boxes && a.dlogits && a.dboxes && a.dboxes2 && a.tc && a.coef && a.wbox && a.tbox && a.tgt_len &&
                   a.empty_weight && a.out,
               "set_criterion: null pointer");
  SEDT_REQUIRE((a.at == nullptr) == (a.dat == nullptr), "set_criterion: at and dat go together");
  SEDT_REQUIRE((a.at_p == nullptr) == (a.dat_p == nullptr), "set_criterion: at_p and dat_p go together");
  SEDT_REQUIRE(a.at_p == nullptr || (a.at != nullptr && a.gt_weak != nullptr && a.Bp >= a.n_lab),
               "set_criterion: at_p needs the audio-tag targets (at / gt_weak) and at least n_lab=%d rows (Bp=%d)", a.n_lab, a.Bp);
  SEDT_REQUIRE(a.L * a.B <= SEDT_CRIT_MAXCARD, "set_criterion: L*B = %d exceeds %d", a.L * a.B, SEDT_CRIT_MAXCARD);
  SEDT_REQUIRE(a.q0 >= 0 && a.Qs >= a.q0 + a.Q, "set_criterion: query window q0=%d Q=%d outside Qs=%d", a.q0, a.Q, a.Qs);
  hipLaunchKernelGGL(set_criterion_kernel, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), a, scratch);
  return check_launch("set_criterion");
}

extern "C" int sedt_set_criterion_bwd(const SedtCriterion* args, const float* g, const float* gtotal, float* glogits, float* gboxes,
                                      float* gat, float* gat_p, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr && (g || gtotal) && glogits && gboxes, "set_criterion_bwd: null pointer");
  const SedtCriterion& a = *args;
  SEDT_REQUIRE(a.L >= 1 && a.L <= SEDT_CRIT_MAXL, "set_criterion_bwd: L=%d", a.L);
  SEDT_REQUIRE(a.dlogits && a.dboxes && a.dboxes2 && ((gat == nullptr) || a.dat) && ((gat_p == nullptr) || a.dat_p),
               "set_criterion_bwd: null gradient buffers");
  const int rows = a.L * a.B * a.Qs;
  hipLaunchKernelGGL(set_criterion_bwd_kernel, dim3((rows + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, g,
                     gtotal, glogits, gboxes, gat, gat_p);
  return check_launch("set_criterion_bwd");
}

extern "C" int sedt_match_targets(const SedtMatch* args, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr, "match_targets: null args");
  const SedtMatch& a = *args;
  SEDT_REQUIRE(a.L >= 1 && a.L <= SEDT_CRIT_MAXL && a.Q >= 1 && a.Q <= 63 && a.C >= 1 && a.C <= 63 && a.ns >= 0 && a.ns <= a.B,
               "match_targets: L=%d Q=%d (<=63) C=%d ns=%d B=%d", a.L, a.Q, a.C, a.ns, a.B);
  SEDT_REQUIRE(a.max_targets >= 1 && a.max_targets <= 63, "match_targets: max_targets=%d (1..63 per clip)", a.max_targets);
  SEDT_REQUIRE(a.q0 >= 0 && a.Qs >= a.q0 + a.Q, "match_targets: query window q0=%d Q=%d outside Qs=%d", a.q0, a.Q, a.Qs);
  SEDT_REQUIRE(a.logits && a.boxes && a.lab_cat && a.lab_off && a.box_cat && a.box_off && a.tc && a.coef && a.wbox && a.tbox &&
                   a.tidx && a.tgt_len,
               "match_targets: null pointer");
  SEDT_REQUIRE(!(a.fine_tune && a.ratio_cat && !a.normalize), "match_targets: fine_tune with mixup ratios is undefined in the reference (matcher.py:130)");
  const size_t lds = std::max((size_t)a.Q * (a.C + 1 + 2 * a.max_targets), (size_t)64 * (a.C + 1)) * sizeof(float);
  hipLaunchKernelGGL(match_targets_kernel, dim3(a.L * a.ns + 1), dim3(64), lds, reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("match_targets");
}
