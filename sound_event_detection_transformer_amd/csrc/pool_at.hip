// pool_at.hip - the --pooling variants of SEDT (reference sedt/sedt.py:47-61 construction, :96-119 forward): the clip-level
// probabilities at_p[b][c] pooled over the event queries' class probabilities y[b][q][c] = softmax(logits[b][q])[c], c < C:
//   max           at_p = max_q y                      (AdaptiveMaxPool2d((1, None)); gradient to the FIRST maximal query)
//   avg           at_p = mean_q y                     (AdaptiveAvgPool2d((1, None)))
//   attn          s = clamp(softmax_c(attn[b][q]), 1e-7, 1),  at_p = sum_q s y / sum_q s          (:53-60)
//   weighted_sum  at_p = clip(sum_q y * length[b][q], 0, 1),  length = pred_boxes[..., 1]         (:98-100)
// One workgroup per clip: the Q x (C+1) class probabilities (and the attention weights) live in LDS, phase 1 is one thread per
// query row (row softmax), phase 2 one thread per class (the pooling over the queries, in query order).  The backward launch
// recomputes the probabilities, forms d at_p / d y (and d / d s, d / d length) per class and finishes with the row-softmax
// backward per query; rows of the logits outside the event-query window [q0, q0 + Q) - the audio-tag query - get zeros.
#include "common.h"

namespace sedt {

constexpr int POOL_THREADS = 128;

// row softmax of x[0..n) into p[0..n) exactly as torch does it: exp(x - max) / sum
__device__ __forceinline__ void row_softmax(const float* __restrict__ x, float* p, int n) {
  float m = -INFINITY;
  for (int c = 0; c < n; ++c) m = fmaxf(m, x[c]);
  float s = 0.f;
  for (int c = 0; c < n; ++c) {
    const float e = expf(x[c] - m);
    p[c] = e;
    s += e;
  }
  for (int c = 0; c < n; ++c) p[c] = p[c] / s;
}

// LDS: prob [Q][C+1], then (attn) t [Q][C] = the un-clamped softmax of the attention logits
__device__ __forceinline__ void pool_load(const SedtPoolAt& a, int b, float* prob, float* tat) {
  const int C1 = a.C + 1;
  for (int q = threadIdx.x; q < a.Q; q += POOL_THREADS) {
    row_softmax(a.logits + ((long)b * a.Qs + a.q0 + q) * C1, prob + q * C1, C1);
    if (a.mode == SEDT_POOL_ATTN) row_softmax(a.attn + ((long)b * a.Q + q) * a.C, tat + q * a.C, a.C);
  }
  __syncthreads();
}

__device__ __forceinline__ float clamp_attn(float t) { return fminf(fmaxf(t, 1e-7f), 1.f); }

__global__ __launch_bounds__(POOL_THREADS) void pool_at_kernel(const SedtPoolAt a, float* __restrict__ at_p) {
  extern __shared__ float lds[];
  const int b = blockIdx.x, C = a.C, C1 = C + 1, Q = a.Q;
  float* prob = lds;
  float* tat = lds + Q * C1;
  pool_load(a, b, prob, tat);
  for (int c = threadIdx.x; c < C; c += POOL_THREADS) {
    float v;
    if (a.mode == SEDT_POOL_MAX) {
      v = prob[c];
      for (int q = 1; q < Q; ++q) v = fmaxf(v, prob[q * C1 + c]);
    } else if (a.mode == SEDT_POOL_AVG) {
      v = 0.f;
      for (int q = 0; q < Q; ++q) v += prob[q * C1 + c];
      v = v / (float)Q;
    } else if (a.mode == SEDT_POOL_ATTN) {
      float num = 0.f, den = 0.f;
      for (int q = 0; q < Q; ++q) {
        const float s = clamp_attn(tat[q * C + c]);
        num += s * prob[q * C1 + c];
        den += s;
      }
      v = num / den;
    } else {
      v = 0.f;
      for (int q = 0; q < Q; ++q) v += prob[q * C1 + c] * a.boxes[((long)b * a.Qs + a.q0 + q) * 2 + 1];
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    at_p[(long)b * C + c] = v;
  }
}

__global__ __launch_bounds__(POOL_THREADS) void pool_at_bwd_kernel(const SedtPoolAt a, const float* __restrict__ g,
                                                                   float* __restrict__ glogits, float* __restrict__ gboxes,
                                                                   float* __restrict__ gattn) {
  extern __shared__ float lds[];
  const int b = blockIdx.x, C = a.C, C1 = C + 1, Q = a.Q;
  float* prob = lds;                       // [Q][C1]
  float* tat = prob + Q * C1;              // [Q][C]  (attn)
  float* gy = tat + (a.mode == SEDT_POOL_ATTN ? Q * C : 0);   // [Q][C]  d total / d y
  float* gs = gy + Q * C;                  // [Q][C]  (attn) d total / d s;  (weighted_sum) [C] g * inside-the-clip mask
  pool_load(a, b, prob, tat);
  // ---- per class: d at_p[c] / d y[q][c] (times the incoming gradient)
  for (int c = threadIdx.x; c < C; c += POOL_THREADS) {
    const float gc = g[(long)b * C + c];
    if (a.mode == SEDT_POOL_MAX) {
      int arg = 0;
      float v = prob[c];
      for (int q = 1; q < Q; ++q)
        if (prob[q * C1 + c] > v) { v = prob[q * C1 + c]; arg = q; }
      for (int q = 0; q < Q; ++q) gy[q * C + c] = q == arg ? gc : 0.f;
    } else if (a.mode == SEDT_POOL_AVG) {
      for (int q = 0; q < Q; ++q) gy[q * C + c] = gc / (float)Q;
    } else if (a.mode == SEDT_POOL_ATTN) {
      float num = 0.f, den = 0.f;
      for (int q = 0; q < Q; ++q) {
        const float s = clamp_attn(tat[q * C + c]);
        num += s * prob[q * C1 + c];
        den += s;
      }
      const float res = num / den;
      for (int q = 0; q < Q; ++q) {
        const float t = tat[q * C + c];
        gy[q * C + c] = gc * clamp_attn(t) / den;
        // clamp passes the gradient where the value lies inside [min, max] (bounds included, as torch.clamp)
        gs[q * C + c] = (t >= 1e-7f && t <= 1.f) ? gc * (prob[q * C1 + c] - res) / den : 0.f;
      }
    } else {
      float v = 0.f;
      for (int q = 0; q < Q; ++q) v += prob[q * C1 + c] * a.boxes[((long)b * a.Qs + a.q0 + q) * 2 + 1];
      const float gin = (v >= 0.f && v <= 1.f) ? gc : 0.f;
      gs[c] = gin;
      for (int q = 0; q < Q; ++q) gy[q * C + c] = gin * a.boxes[((long)b * a.Qs + a.q0 + q) * 2 + 1];
    }
  }
  __syncthreads();
  // ---- per query row: softmax backward (the no-event column C carries no gradient of its own)
  for (int qs = threadIdx.x; qs < a.Qs; qs += POOL_THREADS) {
    float* gl = glogits + ((long)b * a.Qs + qs) * C1;
    const int q = qs - a.q0;
    if (q < 0 || q >= Q) {
      for (int j = 0; j < C1; ++j) gl[j] = 0.f;
      if (gboxes) { gboxes[((long)b * a.Qs + qs) * 2] = 0.f; gboxes[((long)b * a.Qs + qs) * 2 + 1] = 0.f; }
      continue;
    }
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += gy[q * C + c] * prob[q * C1 + c];
    for (int j = 0; j < C1; ++j) gl[j] = prob[q * C1 + j] * ((j < C ? gy[q * C + j] : 0.f) - dot);
    if (a.mode == SEDT_POOL_ATTN) {
      float d2 = 0.f;
      for (int c = 0; c < C; ++c) d2 += gs[q * C + c] * tat[q * C + c];
      for (int c = 0; c < C; ++c) gattn[((long)b * Q + q) * C + c] = tat[q * C + c] * (gs[q * C + c] - d2);
    }
    if (gboxes) {
      float gw = 0.f;
      if (a.mode == SEDT_POOL_WSUM)
        for (int c = 0; c < C; ++c) gw += gs[c] * prob[q * C1 + c];
      gboxes[((long)b * a.Qs + qs) * 2] = 0.f;
      gboxes[((long)b * a.Qs + qs) * 2 + 1] = gw;
    }
  }
}

static int pool_check(const SedtPoolAt& a, const char* what) {
  SEDT_REQUIRE(a.mode >= SEDT_POOL_MAX && a.mode <= SEDT_POOL_WSUM, "%s: mode %d", what, a.mode);
  SEDT_REQUIRE(a.logits != nullptr && a.B >= 1 && a.Q >= 1 && a.C >= 1 && a.q0 >= 0 && a.Qs >= a.q0 + a.Q,
               "%s: B=%d, query window q0=%d Q=%d of Qs=%d, C=%d", what, a.B, a.q0, a.Q, a.Qs, a.C);
  SEDT_REQUIRE((long)a.Q * (a.C + 1) <= 4096, "%s: Q*(C+1) = %ld exceeds 4096 (LDS sizing)", what, (long)a.Q * (a.C + 1));
  SEDT_REQUIRE(a.mode != SEDT_POOL_ATTN || a.attn != nullptr, "%s: attn pooling needs the attention logits", what);
  SEDT_REQUIRE(a.mode != SEDT_POOL_WSUM || a.boxes != nullptr, "%s: weighted_sum pooling needs the boxes", what);
  return 0;
}

}  // namespace sedt

extern "C" int sedt_pool_at(const SedtPoolAt* args, float* at_p, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr && at_p != nullptr, "pool_at: null pointer");
  const SedtPoolAt& a = *args;
  if (int r = pool_check(a, "pool_at")) return r;
  const size_t lds = sizeof(float) * ((size_t)a.Q * (a.C + 1) + (a.mode == SEDT_POOL_ATTN ? (size_t)a.Q * a.C : 0));
  hipLaunchKernelGGL(pool_at_kernel, dim3(a.B), dim3(POOL_THREADS), lds, reinterpret_cast<hipStream_t>(stream), a, at_p);
  return check_launch("pool_at");
}

extern "C" int sedt_pool_at_bwd(const SedtPoolAt* args, const float* g, float* glogits, float* gboxes, float* gattn, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(args != nullptr && g != nullptr && glogits != nullptr, "pool_at_bwd: null pointer");
  const SedtPoolAt& a = *args;
  if (int r = pool_check(a, "pool_at_bwd")) return r;
  SEDT_REQUIRE(a.mode != SEDT_POOL_ATTN || gattn != nullptr, "pool_at_bwd: attn pooling returns a gradient for the attention logits");
  SEDT_REQUIRE(a.mode != SEDT_POOL_WSUM || gboxes != nullptr, "pool_at_bwd: weighted_sum pooling returns a gradient for the boxes");
  const size_t lds = sizeof(float) * ((size_t)a.Q * (a.C + 1) + (a.mode == SEDT_POOL_ATTN ? 3 : 2) * (size_t)a.Q * a.C);
  hipLaunchKernelGGL(pool_at_bwd_kernel, dim3(a.B), dim3(POOL_THREADS), lds, reinterpret_cast<hipStream_t>(stream), a, g, glogits,
                     gboxes, gattn);
  return check_launch("pool_at_bwd");
}
