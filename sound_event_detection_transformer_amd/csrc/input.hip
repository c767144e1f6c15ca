// input.hip - the input side of the training step on the device:
//   * box_transform_kernel  the per-clip feature transforms of reference utilities/BoxTransforms.py composed as
//                           get_transforms does (:454-490): ApplyLog (librosa.amplitude_to_db, :55-67) -> PadOrTrunc (:70-117)
//                           -> TimeMask (:363-395) -> FreqMask(fill "mean" or constant, :398-425) -> FreqShift (:428-451)
//                           -> ToTensor (add the channel axis, f32) -> Normalize (Scaler.normalize, Scaler.py:102-108)
//   * mixup_kernel          the feature half of utilities/mixup.py:13-196 (lam * x1 + (1 - lam) * x2, or one of the two)
// One workgroup per clip: the clip (frames x 64 mel, <= 127 KB as f32) lives in LDS between the passes, so HBM sees one read
// of the raw amplitudes and one write of the normalised features.  Random parameters are drawn on the HOST exactly as the
// reference draws them (np.random, same order) and arrive as integers: the kernel is deterministic.
#include <algorithm>
#include "common.h"

namespace sedt {

struct ClipAug {               // mirrors utilities/transforms.py:_AUG (8 x int32 per clip)
  int32_t nframes_raw;         // frames of the raw clip (rows of amp actually present)
  int32_t tm_t, tm_t0;         // time mask: rows [t0, t0 + t) are zeroed (t = 0: off)
  int32_t fm_f, fm_f0;         // frequency mask: mel bands [f0, f0 + f) are overwritten (f = 0 with fm_on = 0: off)
  int32_t fm_on;               // 1 = apply (f may be 0: numpy then writes nothing)
  int32_t fs_shift;            // frequency shift in bands (0 = off)
  int32_t pad_;
};

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, red[w]);
  return m;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];          // fixed order
  return s;
}

// amp [B][raw_stride][F] f32 mel amplitudes (rows >= nframes_raw are ignored), out [B][1][frames][F] f32
__global__ __launch_bounds__(1024) void box_transform_kernel(const float* __restrict__ amp, long raw_stride, const ClipAug* __restrict__ aug,
                                                             const double* __restrict__ mean, const double* __restrict__ stdv,
                                                             int frames, int F, int apply_log, int fill_mean, float fill_const,
                                                             float* __restrict__ out) {
  extern __shared__ float lds[];
  float* clip = lds;                              // [frames][F]
  float* red = lds + (long)frames * F;            // [16]
  const int b = blockIdx.x, t = threadIdx.x;
  const ClipAug a = aug[b];
  const float* src = amp + (long)b * raw_stride * F;
  const int nraw = a.nframes_raw;
  // ---- pass 1: amplitude -> dB (10 log10(max(1e-10, x^2)), reference value 1.0), clip maximum over ALL raw frames
  float mx = -INFINITY;
  for (long i = t; i < (long)nraw * F; i += 1024) {
    const float x = src[i];
    const float v = apply_log ? 10.f * log10f(fmaxf(1e-10f, x * x)) : x;
    mx = fmaxf(mx, v);
    if (i < (long)frames * F) clip[i] = v;
  }
  for (long i = (long)nraw * F + t; i < (long)frames * F; i += 1024) clip[i] = 0.f;        // PadOrTrunc: zero rows appended
  mx = block_max(mx, red);
  __syncthreads();
  const float floor_db = mx - 80.f;               // top_db = 80
  const int keep = min(nraw, frames);
  // ---- pass 2 (LDS): top_db clamp on the real rows, then the time mask
  for (long i = t; i < (long)frames * F; i += 1024) {
    const int r = (int)(i / F);
    float v = clip[i];
    if (apply_log && r < keep) v = fmaxf(v, floor_db);
    if (r >= a.tm_t0 && r < a.tm_t0 + a.tm_t) v = 0.f;
    clip[i] = v;
  }
  __syncthreads();
  // ---- frequency mask: the bands [f0, f0 + f) take their mean over the whole (padded, time-masked) clip
  if (a.fm_on && a.fm_f > 0) {
    float fill = fill_const;
    if (fill_mean) {
      float s = 0.f;
      for (long i = t; i < (long)frames * a.fm_f; i += 1024) {
        const int r = (int)(i / a.fm_f), c = a.fm_f0 + (int)(i - (long)r * a.fm_f);
        s += clip[(long)r * F + c];
      }
      s = block_sum(s, red);
      fill = s / (float)((long)frames * a.fm_f);
      __syncthreads();
    }
    for (long i = t; i < (long)frames * a.fm_f; i += 1024) {
      const int r = (int)(i / a.fm_f), c = a.fm_f0 + (int)(i - (long)r * a.fm_f);
      clip[(long)r * F + c] = fill;
    }
    __syncthreads();
  }
  // ---- frequency shift (np.roll along mel, wrapped bands zeroed) + normalisation (float64 like Scaler.normalize), store
  float* dst = out + (long)b * frames * F;
  const int sh = a.fs_shift;
  for (long i = t; i < (long)frames * F; i += 1024) {
    const int r = (int)(i / F), c = (int)(i - (long)r * F);
    const int cs = c - sh;                         // out[c] = in[c - shift]
    const float v = (cs >= 0 && cs < F) ? clip[(long)r * F + cs] : 0.f;
    dst[i] = mean ? (float)(((double)v - mean[c]) / stdv[c]) : v;
  }
}

struct MixJob {                // mirrors utilities/mixup.py (4 x int32 + 1 float per output clip)
  int32_t src1, src2;          // rows of x1 / x2
  int32_t mode;                // 0: lam * x1[src1] + (1 - lam) * x2[src2], 1: x1[src1], 2: x2[src2]
  float lam;
};

__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                    const MixJob* __restrict__ jobs, long clip4, float* __restrict__ out) {
  const MixJob j = jobs[blockIdx.y];
  const float4* a = reinterpret_cast<const float4*>(x1) + (long)j.src1 * clip4;
  const float4* b = reinterpret_cast<const float4*>(x2) + (long)j.src2 * clip4;
  float4* o = reinterpret_cast<float4*>(out) + (long)blockIdx.y * clip4;
  const float l = j.lam, m = 1.f - j.lam;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < clip4; i += (long)gridDim.x * 256) {
    float4 v;
    if (j.mode == 1) v = a[i];
    else if (j.mode == 2) v = b[i];
    else {
      const float4 p = a[i], q = b[i];
      v.x = l * p.x + m * q.x; v.y = l * p.y + m * q.y; v.z = l * p.z + m * q.z; v.w = l * p.w + m * q.w;
    }
    o[i] = v;
  }
}

// ---- label half of mixup_label_unlabel (reference utilities/mixup.py:129-196) on the device.  In the mean-teacher step the
// pseudo labels of the unlabelled clips are produced ON the device between the teacher and the student forward
// (pseudo_labels_kernel), so the merge with the labelled targets cannot be planned on the host without a device->host copy in
// the middle of the step.  One workgroup, one wave per unlabelled clip i (round robin):
//   i < mix_num:  n1 + n2 > max_events -> the pseudo target (if it has events) else the labelled one;
//                 else candidate = labels1 ++ labels2, boxes1 ++ boxes2, ratio = lam x len(labels1) ++ (1-lam) x len(labels2);
//                 two events of one class overlapping in time (sorted by onset: end[k] < onset[k+1] violated; evaluated pair-wise
//                 in the reference's f32 arithmetic) -> abandoned: the labelled clip and its target replace the unlabelled one;
//   i >= mix_num: the pseudo target unchanged.
// Outputs: the merged targets in the flat layout match_targets_kernel reads (every clip counts as strong), and the MixJob records
// that make mixup_kernel produce the matching features (mode 0 mix, 1 labelled clip, 2 unlabelled clip).
struct MixTargets {
  const int64_t* lab1; const int32_t* lab_off1; const float* box1; const int32_t* box_off1; const float* ratio1; const int32_t* split1;
  const int64_t* lab2; const int32_t* lab_off2; const float* box2; const int32_t* box_off2;
  const float* lam;              // device {lam, 1 - lam} as the host rounded them to f32
  int64_t* lab_out; int32_t* lab_off_out; float* box_out; int32_t* box_off_out; float* ratio_out;
  MixJob* jobs;
  int32_t B1, ns1, B2, mix_num, max_events, cap;
};

__global__ __launch_bounds__(1024) void mixup_targets_kernel(const MixTargets a) {
  extern __shared__ int sm[];                 // [B2] decision, [B2+1] label offsets, [B2+1] box offsets
  int* dec = sm;
  int* lo = sm + a.B2;
  int* bo = lo + a.B2 + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int ns1 = a.split1 ? min(a.split1[0], a.ns1) : a.ns1;
  for (int i = wave; i < a.B2; i += nw) {
    const int l2o = a.lab_off2[i], nl2 = a.lab_off2[i + 1] - l2o, b2o = a.box_off2[i], nb2 = a.box_off2[i + 1] - b2o;
    int d = 2, nl = nl2, nb = nb2;
    if (i < a.mix_num && i < a.B1) {
      const int l1o = a.lab_off1[i], nl1 = a.lab_off1[i + 1] - l1o;
      const int b1o = i < ns1 ? a.box_off1[i] : 0, nb1 = i < ns1 ? a.box_off1[i + 1] - b1o : 0;
      if (nb1 + nb2 > a.max_events) {
        if (nb2 == 0) { d = 1; nl = nl1; nb = nb1; }
      } else {
        const int n = nb1 + nb2;                     // candidate boxes; box j carries label j of the concatenated LABEL list
        bool clash = false;
        for (int j = lane; j < n; j += 64) {
          const float cj = j < nb1 ? a.box1[2 * (b1o + j)] : a.box2[2 * (b2o + j - nb1)];
          const float lj = j < nb1 ? a.box1[2 * (b1o + j) + 1] : a.box2[2 * (b2o + j - nb1) + 1];
          const long ej = j < nl1 ? a.lab1[l1o + j] : a.lab2[l2o + j - nl1];
          const float sj = cj - lj / 2, tj = cj + lj / 2;
          for (int k = 0; k < j; ++k) {
            const long ek = k < nl1 ? a.lab1[l1o + k] : a.lab2[l2o + k - nl1];
            if (ek != ej) continue;
            const float ck = k < nb1 ? a.box1[2 * (b1o + k)] : a.box2[2 * (b2o + k - nb1)];
            const float lk = k < nb1 ? a.box1[2 * (b1o + k) + 1] : a.box2[2 * (b2o + k - nb1) + 1];
            const float sk = ck - lk / 2, tk = ck + lk / 2;
            if (!(tj < sk) && !(tk < sj)) clash = true;
          }
        }
        if (__any(clash)) { d = 1; nl = nl1; nb = nb1; }
        else { d = 0; nl = nl1 + nl2; nb = n; }
      }
    }
    if (lane == 0) { dec[i] = d; lo[i + 1] = nl; bo[i + 1] = nb; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo[0] = 0; bo[0] = 0;
    for (int i = 0; i < a.B2; ++i) { lo[i + 1] += lo[i]; bo[i + 1] += bo[i]; }
  }
  __syncthreads();
  const bool fits = lo[a.B2] <= a.cap && bo[a.B2] <= a.cap;      // (capacity = B2 * max_targets >= max_events per clip: always)
  for (int i = threadIdx.x; i <= a.B2; i += blockDim.x) {
    a.lab_off_out[i] = fits ? lo[i] : 0;
    a.box_off_out[i] = fits ? bo[i] : 0;
  }
  const float lam = a.lam[0], lam1 = a.lam[1];
  for (int i = wave; i < a.B2; i += nw) {
    const int d = dec[i];
    if (lane == 0) a.jobs[i] = MixJob{i, i, d, d == 0 ? lam : 0.f};
    if (!fits) continue;
    const int l2o = a.lab_off2[i], nl2 = a.lab_off2[i + 1] - l2o, b2o = a.box_off2[i], nb2 = a.box_off2[i + 1] - b2o;
    int l1o = 0, nl1 = 0, b1o = 0, nb1 = 0;
    if (d != 2) {
      l1o = a.lab_off1[i]; nl1 = a.lab_off1[i + 1] - l1o;
      if (i < ns1) { b1o = a.box_off1[i]; nb1 = a.box_off1[i + 1] - b1o; }
    }
    const int take1 = d == 2 ? 0 : nl1, take2 = d == 1 ? 0 : nl2;
    for (int j = lane; j < take1 + take2; j += 64) {
      a.lab_out[lo[i] + j] = j < take1 ? a.lab1[l1o + j] : a.lab2[l2o + j - take1];
      if (a.ratio_out) a.ratio_out[lo[i] + j] = d == 0 ? (j < take1 ? lam : lam1) : (d == 1 && a.ratio1 ? a.ratio1[l1o + j] : 1.f);
    }
    const int tb1 = d == 2 ? 0 : nb1, tb2 = d == 1 ? 0 : nb2;
    for (int j = lane; j < 2 * (tb1 + tb2); j += 64)
      a.box_out[2 * bo[i] + j] = j < 2 * tb1 ? a.box1[2 * b1o + j] : a.box2[2 * b2o + j - 2 * tb1];
  }
}

// ---- SP-SEDT query patches (reference utilities/BoxTransforms.py:315-360, Query.transform_label): crop rows [s, e) of a
// transformed clip, min-max normalise to [0, 1], quantise to 8 bits (torchvision ToPILImage: x 255, truncated), resize to 128
// rows with PIL's bilinear resampling, back to float (/ 255) and de-normalise.  Only the vertical pass of Pillow's
// ImagingResample runs (the mel axis keeps its 64 bands): triangle filter of support 1 stretched by the scale when shrinking,
// double-precision weights normalised per output row, rounded to 22-bit fixed point, int32 accumulation with a half-unit
// bias, clipped to [0, 255] (Resample.c).  The float steps use explicitly rounded intrinsics so nothing is contracted into
// an FMA: the result is bit-identical to the reference pipeline.  One workgroup per patch, the 8-bit crop lives in LDS.
struct PatchJob {              // mirrors utilities/transforms.py (4 x int32 per patch)
  int32_t clip, s_idx, e_idx, pad_;
};

__global__ __launch_bounds__(1024) void query_patch_kernel(const float* __restrict__ data, int T, int F, const PatchJob* __restrict__ jobs,
                                                           int fixed, float* __restrict__ out) {
  // HIP's __fmul_rn / __fadd_rn are inline functions compiled with contraction allowed: after inlining the backend still fuses
  // them into one FMA (an ulp off the reference's separately rounded mul and add).  Plain operators under this pragma carry no
  // contract flag.
#pragma clang fp contract(off)
  extern __shared__ unsigned char code[];            // [h][F] uint8, then 32 floats of reduction scratch (4-byte aligned)
  const PatchJob j = jobs[blockIdx.x];
  const int t = threadIdx.x, h = j.e_idx - j.s_idx;
  const float* src = data + ((long)j.clip * T + j.s_idx) * F;
  float* dst = out + (long)blockIdx.x * 128 * F;
  if (fixed) {                                       // fixed_patch_size: the 128 rows as they are
    for (int i = t; i < 128 * F; i += 1024) dst[i] = src[i];
    return;
  }
  float* red = reinterpret_cast<float*>(code + (((long)h * F + 15) & ~15L));
  float mn = INFINITY, mx = -INFINITY;
  for (int i = t; i < h * F; i += 1024) {
    const float v = src[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  mx = block_max(mx, red);
  mn = -block_max(-mn, red);
  const float range = mx - mn;
  for (int i = t; i < h * F; i += 1024)
    code[i] = (unsigned char)(int)(((src[i] - mn) / range) * 255.f);
  __syncthreads();
  const double scale = (double)h / 128.0;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale, ss = 1.0 / filterscale;
  for (int i = t; i < 128 * F; i += 1024) {
    const int yy = i / F, col = i - yy * F;
    int q;
    if (h == 128) {
      q = code[i];                                   // same size: Pillow copies
    } else {
      const double center = (yy + 0.5) * scale;
      int ymin = (int)(center - support + 0.5);
      if (ymin < 0) ymin = 0;
      int ymax = (int)(center + support + 0.5);
      if (ymax > h) ymax = h;
      ymax -= ymin;
      double ww = 0.0;
      for (int y = 0; y < ymax; ++y) {
        double x = ((double)(y + ymin) - center + 0.5) * ss;
        x = x < 0.0 ? -x : x;
        ww += x < 1.0 ? 1.0 - x : 0.0;
      }
      int acc = 1 << 21;
      for (int y = 0; y < ymax; ++y) {
        double x = ((double)(y + ymin) - center + 0.5) * ss;
        x = x < 0.0 ? -x : x;
        double w = x < 1.0 ? 1.0 - x : 0.0;
        if (ww != 0.0) w = w / ww;
        const double kq = w * 4194304.0;
        acc += (int)code[(ymin + y) * F + col] * (int)(0.5 + kq);
      }
      q = acc >> 22;
      q = q < 0 ? 0 : (q > 255 ? 255 : q);
    }
    const float back = (float)q / 255.f;
    const float scaled = back * range;
    dst[i] = scaled + mn;
  }
}

}  // namespace sedt

extern "C" int sedt_query_patches(const float* data, int B, int T, int F, const void* jobs, int n_patches, int fixed, float* out,
                                  void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(data && jobs && out, "query_patches: null pointer");
  SEDT_REQUIRE(B >= 1 && T >= 1 && F >= 1 && n_patches >= 0, "query_patches: B=%d T=%d F=%d n=%d", B, T, F, n_patches);
  const size_t lds = (((size_t)T * F + 15) & ~(size_t)15) + 32 * sizeof(float);
  SEDT_REQUIRE(lds <= 160 * 1024, "query_patches: a crop of up to %d x %d bytes does not fit the 160 KB LDS", T, F);
  if (n_patches == 0) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(query_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(query_patch_kernel, dim3(n_patches), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), data, T, F,
                     reinterpret_cast<const PatchJob*>(jobs), fixed, out);
  return check_launch("query_patches");
}

extern "C" int sedt_box_transform(const float* amp, int64_t raw_stride, const void* aug, const double* mean, const double* stdv,
                                  int B, int frames, int F, int apply_log, int fill_mean, float fill_const, float* out, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(amp && aug && out, "box_transform: null pointer");
  SEDT_REQUIRE((mean == nullptr) == (stdv == nullptr), "box_transform: mean and std go together");
  SEDT_REQUIRE(B >= 0 && frames >= 1 && F >= 1 && raw_stride >= 1, "box_transform: B=%d frames=%d F=%d", B, frames, F);
  const size_t lds = ((size_t)frames * F + 16) * sizeof(float);
  SEDT_REQUIRE(lds <= 160 * 1024, "box_transform: a clip of %d x %d f32 does not fit the 160 KB LDS", frames, F);
  if (B == 0) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(box_transform_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(box_transform_kernel, dim3(B), dim3(1024), lds, reinterpret_cast<hipStream_t>(stream), amp, (long)raw_stride,
                     reinterpret_cast<const ClipAug*>(aug), mean, stdv, frames, F, apply_log, fill_mean, fill_const, out);
  return check_launch("box_transform");
}

extern "C" int sedt_mixup(const float* x1, const float* x2, const void* jobs, int n_out, int64_t clip_elems, float* out, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(x1 && x2 && jobs && out, "mixup: null pointer");
  SEDT_REQUIRE(clip_elems % 4 == 0 && n_out >= 0, "mixup: clip size %ld must be a multiple of 4", (long)clip_elems);
  if (n_out == 0) return 0;
  const long c4 = clip_elems / 4;
  const unsigned gx = (unsigned)std::min<long>((c4 + 255) / 256, 64);
  hipLaunchKernelGGL(mixup_kernel, dim3(gx, n_out), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x1, x2,
                     reinterpret_cast<const MixJob*>(jobs), c4, out);
  return check_launch("mixup");
}

extern "C" int sedt_mixup_targets(const int64_t* lab1, const int32_t* lab_off1, const float* box1, const int32_t* box_off1,
                                  const float* ratio1, const int32_t* split1, int B1, int ns1, const int64_t* lab2,
                                  const int32_t* lab_off2, const float* box2, const int32_t* box_off2, int B2, const float* lam,
                                  int mix_num, int max_events, int64_t* lab_out, int32_t* lab_off_out, float* box_out,
                                  int32_t* box_off_out, float* ratio_out, int cap, void* jobs, void* stream) {
  using namespace sedt;
  SEDT_REQUIRE(lab1 && lab_off1 && box1 && box_off1 && lab2 && lab_off2 && box2 && box_off2 && lam && lab_out && lab_off_out &&
                   box_out && box_off_out && jobs,
               "mixup_targets: null pointer");
  SEDT_REQUIRE(B1 >= 0 && B2 >= 1 && B2 <= 4096 && ns1 >= 0 && ns1 <= B1 && mix_num >= 0 && mix_num <= B2 && mix_num <= B1 &&
                   max_events >= 0 && cap >= 1,
               "mixup_targets: B1=%d ns1=%d B2=%d mix_num=%d", B1, ns1, B2, mix_num);
  MixTargets a{lab1, lab_off1, box1, box_off1, ratio1, split1, lab2, lab_off2, box2, box_off2, lam, lab_out, lab_off_out, box_out,
               box_off_out, ratio_out, reinterpret_cast<MixJob*>(jobs), B1, ns1, B2, mix_num, max_events, cap};
  hipLaunchKernelGGL(mixup_targets_kernel, dim3(1), dim3(1024), (3 * (size_t)B2 + 2) * sizeof(int),
                     reinterpret_cast<hipStream_t>(stream), a);
  return check_launch("mixup_targets");
}
