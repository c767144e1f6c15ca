/* sedt_hip.h - C ABI of libsedt_hip.so, the MI355X (gfx950) implementation of the
 * SEDT forward/backward hot path.
 *
 * Every entry point takes plain device pointers, sizes and a hipStream_t (passed as
 * void*); no torch types cross this boundary.  All buffers (inputs, parameters,
 * gradients, outputs, saved activations, workspaces) are owned by the caller; the
 * library never allocates or frees device memory and keeps no pointer across calls
 * (the weight-pack cache in a plan handle stores host-side copies of pointer VALUES
 * only, to detect changes).  Return value: 0 = ok, non-zero = error;
 * sedt_last_error() returns the message (thread-local).  No exception crosses the ABI.
 *
 * dtype codes: SEDT_F32 = parity mode (f32 operands, v_mfma_f32_32x32x2_f32, exact f32
 * FMA chains); SEDT_BF16 = throughput mode (bf16 operands, f32 accumulate,
 * v_mfma_f32_32x32x16_bf16).  Parameters and gradients are always f32.
 *
 * Binding: the library is built with plain `hipcc -shared` and bound with ctypes (sound_event_detection_transformer_amd/
 * lib.py mirrors every prototype below); no torch headers are involved - torch only supplies device pointers and streams.
 *
 * Reference interface each group replaces (file:line under the reference repo):
 *   sedt_igemm / sedt_igemm_group / sedt_wgrad_group / sedt_multi_wgrad_reduce / sedt_skinny_linear_*
 *        torch.nn.Conv2d / Linear forward + autograd: sedt/transformer.py:160-165, 183-204, 220-233, 248-284,
 *        sedt/sedt.py:36, 88-92, 398-409, torchvision Bottleneck convs behind sedt/backbone.py:98-100
 *   sedt_bn_fold / sedt_multi_bn_fold / sedt_pack_conv / sedt_multi_pack      FrozenBatchNorm2d     sedt/backbone.py:17-53
 *   sedt_stem_* / sedt_maxpool_* / sedt_avgpool / sedt_mask_resize            Backbone.forward      sedt/backbone.py:56-113
 *   sedt_layernorm_* / sedt_attention_* / sedt_add / sedt_dropout_grad        TransformerEncoder/DecoderLayer  sedt/transformer.py:155-297
 *   sedt_posenc                                                               PositionEmbeddingSine sedt/position_encoding.py:27-47
 *   sedt_match_targets / sedt_hungarian_batch                                 HungarianMatcher      sedt/matcher.py:41-133
 *   sedt_set_criterion(_bwd) / sedt_feature_loss                              SetCriterion          sedt/sedt.py:134-352
 *   sedt_postprocess / sedt_pseudo_labels                                     PostProcess, get_pseudo_labels  sedt/sedt.py:355-396, engine.py:300-348
 *   sedt_multi_sumsq / sedt_multi_adamw / sedt_adamw_clip                     clip_grad_norm_ + AdamW.step  engine.py:77-80
 *   sedt_multi_ema                                                            EMA.update            utilities/utils.py:62-67
 *   sedt_multi_gather                                                         DDP gradient buckets  train_spsedt.py:157-158
 */
#ifndef SEDT_HIP_H
#define SEDT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEDT_F32 0
#define SEDT_BF16 1
/* GEMM entry points only (sedt_igemm, sedt_igemm_group, sedt_wgrad_group, sedt_igemm_splitk): f32 tensors exactly as SEDT_F32, but
 * every product computed from bf16 hi / lo splits of both operands - hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16, f32
 * accumulation - instead of v_mfma_f32_32x32x2_f32: ~2^-16 relative error per product, about five times the matrix rate of the
 * exact-f32 mode.  The "bf16x3" compute mode (runtime.set_compute_dtype('bf16x3')): f32 activations and gradients everywhere,
 * this code for the contractions; meets the 1e-3 parity tolerance (tests/test_x3_gpu.py). */
#define SEDT_BF16X3 2

#define SEDT_ACT_NONE 0
#define SEDT_ACT_RELU 1
#define SEDT_ACT_SIGMOID 2

const char* sedt_last_error(void);
int sedt_version(void);

/* ------------------------------------------------------------------ implicit GEMM
 * C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )
 *
 * trans == 0 (forward / dgrad):
 *   A(m,k): conv == 0: A[m*lda + k]
 *           conv == 1: m = (img, ho, wo) over the Ho x Wo "row grid", k = (tap, c) with c < Ci,
 *                      element = A[pix*lda + c], pix = gathered pixel of the Hi x Wi grid:
 *                        transposed == 0: hi = ho*sh - ph + kh*dh        (convolution forward)
 *                        transposed == 1: hi = (ho + ph - kh*dh)/sh      (dgrad: rows are input pixels)
 *                      out-of-range / non-divisible taps read as 0.
 *   B(n,k): B[n*ldb + k]    (weights packed [N][taps][Ci])
 * trans == 1 (wgrad; the reduction runs over pixels):
 *   A(m,k): A[k*lda + m]                      (dY: k = output pixel, m = output channel)
 *   B(n,k): conv == 0: B[k*ldb + n]
 *           conv == 1: n = (tap, c), element = B[pix(k,tap)*ldb + c] (forward gather)
 *   splitk > 1: partial sums go to slab[z][M][N] (f32) and the epilogue is skipped.
 *
 * epilogue, in order: v = acc*scale[n] + bias[n]; if(!act_post_res) v = act(v);
 *   dropout(v); v += res[(res_mod ? m % res_mod : m)*ldr + n]; if(act_post_res) v = act(v);
 *   v = mask[m*ldm+n] > 0 ? v : 0; v *= alpha; store as f32 (out_f32) or the compute dtype.
 * 1-bit ReLU masks (round 3): a backward pass needs only the SIGN of a saved post-ReLU activation.  mask_bits != 0: `mask` is a
 *   uint8 bit image - bit (n & 7) of mask[m*ldm + (n >> 3)], ldm in BYTES - instead of the activation itself (1/16 of the bytes of a
 *   bf16 tensor); bits_out != null: the same image of the value stored to C (bit = v > 0) is written to bits_out[m*ldbits + (n >> 3)]
 *   (N must be a multiple of 8).  trans == 0 only.
 */
typedef struct SedtIgemm {
  int32_t M, N, K;
  const void* A;
  const void* B;
  int64_t lda, ldb;
  int32_t trans, conv, transposed;
  int32_t Hi, Wi, Ci, Ho, Wo;
  int32_t KH, KW, sh, sw, ph, pw, dh, dw;
  void* C;
  int64_t ldc;
  int32_t out_f32;
  const float* scale;
  const float* bias;
  const void* res;
  int64_t ldr;
  int32_t res_mod;
  const void* mask;
  int64_t ldm;
  int32_t act, act_post_res;
  float alpha;
  float drop_p;
  uint32_t seed;            /* effective seed = seed + (seed_ptr ? *seed_ptr : 0): the device word */
  const uint32_t* seed_ptr; /* lets a captured hipGraph draw a fresh mask on every replay */
  int32_t splitk;
  float* slab;
  int32_t tile_m, tile_n; /* 0 = choose automatically; else 64 or 128 */
  float* colsum_out;      /* trans == 1, bf16 LDS-DMA kernel only: if non-null, per-split column sums of A over the
                             reduction axis (= the bias gradient sum_pix dY[pix][m]) are written to colsum_out[z][M];
                             sedt_wgrad_reduce adds them up.  sedt_igemm fails if the fast kernel cannot take the problem. */
  uint8_t* bits_out;      /* or null: sign bits of the stored output (see "1-bit ReLU masks") */
  int64_t ldbits;         /* bytes per row of bits_out */
  int32_t mask_bits;      /* != 0: `mask` is a bit image, ldm in bytes */
  int32_t f32ep;          /* bf16 operands with an f32 epilogue (the fast bf16x3 mode, sedt_split3): C (out_f32 must be set), res and a
                             non-bit mask are f32 tensors (ldc / ldr / ldm in f32 elements); trans == 0, LDS-DMA kernels only */
  /* conv problems on the LDS-DMA kernels only (a parity class of a stride-2 input gradient, ops.conv_dgrad): */
  int32_t omap;           /* != 0: GEMM row (n, ho, wo) of the Ho x Wo grid is pixel (ho * o_sh + o_h0, wo * o_sw + o_w0) of an o_Hi x o_Wi
                             image: C, res, mask, bits_out are indexed by that pixel */
  int32_t o_Hi, o_Wi, o_sh, o_sw, o_h0, o_w0;
  int32_t btap_on;        /* != 0: tap t of the (<= 8-tap) walk reads the Ci channels starting at element btap[t] of a B row (ldb spans
                             all the taps the packed weight holds) */
  int32_t btap[8];
  void* split_out;        /* f32ep only, or null: bf16 [M][2 N] = the [hi | lo] operand image (sedt_split3, pattern 0) of the stored
                             output, written by the epilogue - the GEMMs that consume this output then need no split pass */
  int32_t awrap;          /* bf16x3, trans == 0: the A rows hold [hi | lo] = 2 awrap channels while the contraction walks 3 awrap per pixel
                             (K = 3 awrap, a convolution: Ci = 3 awrap): the last third re-reads hi.  0 = off */
  int32_t pad2_;
  const void* bfrag;      /* or null: the fragment-major image (sedt_pack_frag) of the WHOLE B operand [N][ldb] (trans == 0, bf16): kernels
                             that stream B from L2 into registers read it instead of B (developer build only so far) */
} SedtIgemm;

int sedt_igemm(const SedtIgemm* args, int dtype, void* stream);
/* sizeof of an argument struct as THIS library was compiled (0 SedtIgemm, 1 SedtReduceJob, 2 SedtSplitJob, 3 SedtPrefetch, 4 SedtCriterion,
 * 5 SedtMatch, 6 SedtChunk, 7 SedtBnJob, 8 SedtPackJob, 9 SedtFragJob, 10 SedtPoolAt, 11 SedtCopyJob; -1 otherwise): lets a binding verify its mirror of the structs */
int sedt_sizeof(int which);
/* njobs (<= 8 per launch) independent trans == 0 problems - e.g. the q / k / v projections of an attention block; one
 * launch when every problem resolves to the 64x64 2-stage bf16 kernel, otherwise njobs sedt_igemm calls.  HOST array. */
int sedt_igemm_group(const SedtIgemm* jobs, int njobs, int dtype, void* stream);
/* njobs independent trans == 1 (weight-gradient) problems; in bf16 they run as ONE launch (grouped workgroup ranges) when
 * every problem fits the LDS-DMA kernel, otherwise this is njobs sedt_igemm calls.  `jobs` is a HOST array. */
int sedt_wgrad_group(const SedtIgemm* jobs, int njobs, int dtype, void* stream);
/* one forward / dgrad problem (`main`, trans == 0) plus nw <= 10 weight-gradient problems in ONE launch: the wgrad tiles run
 * in workgroup slots the main problem leaves idle (co-scheduling).  *taken = 1: semantically sedt_igemm(main) +
 * sedt_wgrad_group(wjobs); *taken = 0: only `main` was launched (its kernel configuration cannot carry riders, or a rider
 * is outside the grouped kernel's envelope) and the caller still owns the weight-gradient problems. */
int sedt_igemm_co(const SedtIgemm* main, const SedtIgemm* wjobs, int nw, int dtype, void* stream, int* taken);
/* recommended split-K factor and slab bytes for a trans==1 problem */
int sedt_igemm_splitk(int M, int N, int K, int dtype);
/* measurement aid: the name of the kernel instance the problem runs on - as rocprofv3 prints it, e.g. "igemm3_w16_kernel<64, 128, 3>" -
 * through sedt_igemm (grouped = 0) or through sedt_wgrad_group (grouped = 1, trans problems).  Nothing is launched.  bench.py joins
 * these names with the per-kernel times of the measured step to price every kernel family against its roofline. */
int sedt_igemm_describe(const SedtIgemm* args, int dtype, int grouped, char* out, int cap);
/* the same for sedt_igemm_group: the ONE kernel instance the njobs problems run on as a group ("igemm3_group_kernel<2>",
 * "igemm3_w8_group_kernel<128, 128, 3>"), or "" when the dispatcher would launch them one by one */
int sedt_igemm_group_describe(const SedtIgemm* jobs, int njobs, int dtype, char* out, int cap);

/* out[r][c...] = rowscale[r] * sum_z slab[z][r][tap][c], written in (R, Ci, taps) order
 * (the torch (Cout, Cin, KH, KW) parameter layout) as f32.  taps == 1: plain [R][Ci]. */
int sedt_wgrad_reduce(const float* slab, int splitk, int R, int taps, int Ci, const float* rowscale,
                      float* out, void* stream);
/* same, plus bias_out[r] = sum_z colsum_slab[z][r] (no rowscale) when colsum_slab is non-null */
int sedt_wgrad_reduce_bias(const float* slab, int splitk, int R, int taps, int Ci, const float* rowscale, float* out,
                           const float* colsum_slab, float* bias_out, void* stream);

/* A prefetch hint, passed BY THE LAUNCH THAT CONSUMES IT (the library keeps no pointers between calls): up to three regions (null /
 * bytes 0 = none) that the launch AFTER this one will stream - fragment-major weights.  The launch touches one 128-byte line of them
 * per load so that every XCD's L2 holds them when the streaming launch starts.  Taken by sedt_encoder_qkv_fwd (the weights of
 * sedt_encoder_attn_ffn_fwd), sedt_bneck3_fwd / sedt_bneck3_bwd (the next block's operands) and sedt_multi_wgrad_reduce (what the next
 * layer's backward streams first); `pf` may be null. */
typedef struct SedtPrefetch {
  const void* ptr[3];
  size_t bytes[3];
} SedtPrefetch;

/* Operand preparation of the fast bf16x3 mode (csrc/split3.hip): src f32 [rows][cols] (row stride ld) -> bf16, hi = bf16(x), lo = bf16(x - hi):
 *   pattern 0 (an activation / gradient operand): dst [rows][2 * cols] = [hi | lo] - the bytes of the f32 tensor;
 *   pattern 1 (a weight operand; rows = Cout * taps, cols = Cin): dst [rows][3 * cols] = [hi | hi | lo].
 * A bf16 GEMM whose contraction walks hi, lo, hi of the activation (SedtIgemm.awrap: the last third wraps back onto hi) against
 * [hi | hi | lo] of the weight yields hi hi + lo hi + hi lo in its f32 accumulator: an f32 product to ~2^-16 at bf16 MFMA rate (with
 * SedtIgemm.f32ep for the epilogue).  Up to 8 jobs per launch (HOST array, copied into the kernel arguments).  cols and ld multiples of 4,
 * src 16-byte aligned. */
typedef struct SedtSplitJob {
  const float* src;
  int64_t ld;
  void* dst;
  int32_t rows, cols, pattern;
  int32_t blk0;           /* filled by the library */
} SedtSplitJob;
int sedt_split3(const SedtSplitJob* jobs, int njobs, void* stream);

/* up to SEDT_MAX_REDUCE_JOBS split-K reductions in ONE launch (jobs are copied into the kernel arguments, so a captured
 * graph holds them by value).  `jobs` is a HOST array; fields as the arguments of sedt_wgrad_reduce_bias. */
#define SEDT_MAX_REDUCE_JOBS 40
typedef struct SedtReduceJob {
  const float* slab;
  const float* rowscale;
  float* out;
  const float* colsum_slab;
  float* bias_out;
  int32_t splitk, R, taps, Ci;
  int32_t blk0;           /* filled by the library */
  int32_t cs_splitk;      /* slices of colsum_slab when they differ from splitk (0 = splitk) */
} SedtReduceJob;
int sedt_multi_wgrad_reduce(const SedtReduceJob* jobs, int njobs, const SedtPrefetch* pf, void* stream);

/* ------------------------------------------------------------------ linear layers with N <= 16 outputs (the SEDT heads,
 * sedt/sedt.py:36-38, 90-95, 398-409): direct kernels on the f32 MASTER weight w[N][K] (no packing).
 * fwd: y[m][n] = act(x[m] . w[n] + bias[n]); y is f32 (out_f32) or the compute dtype.
 * bwd: g[M][N] f32 is the gradient of y; with act != NONE the derivative is folded in from ysaved (= y, f32, same ld as g).
 *      gx (compute dtype, optional; masked by mask > 0 when given) = g' w - or gx += g' w when accumulate_gx != 0 (a head that
 *      shares its input with other heads adds to the running input gradient; rows ldo apart, so a strided row subset works);
 *      dw[N][K], db[N] (optional) = g'^T x, column sums.  With scratch given and dw NULL only the partial sums are left in
 *      scratch as [16 slabs][17][K] (rows 0..15 dW, row 16 = the bias sums in its first 16 entries): the caller adds the slabs. */
int sedt_skinny_linear_fwd(const void* x, int64_t ldx, const float* w, const float* bias, void* y, int64_t ldy, int M, int N,
                           int K, int act, int out_f32, int dtype, void* stream);
size_t sedt_skinny_linear_bwd_scratch(int K); /* bytes of `scratch` (row-slice partial sums) when dw is requested */
int sedt_skinny_linear_bwd(const float* g, const float* ysaved, int64_t ldg, const float* w, const void* x, int64_t ldx,
                           const void* mask, int64_t ldm, void* gx, int64_t ldo, float* dw, float* db, float* scratch, int M, int N,
                           int K, int act, int accumulate_gx, int dtype, void* stream);

/* ------------------------------------------------------------------ elementwise / reductions */
/* out[c] = sum_r in[r*ld + c]   (in: compute dtype or f32 if in_f32), out f32 */
int sedt_colsum(const void* in, int64_t ld, int rows, int cols, int in_f32, int dtype, float* out,
                float* scratch, size_t scratch_bytes, void* stream);
size_t sedt_colsum_scratch(int rows, int cols);
/* out = in * keep(seed, r*cols+c) / (1-p)   - the gradient side of the epilogue dropout */
int sedt_dropout_grad(const void* in, int64_t ldi, void* out, int64_t ldo, int rows, int cols, float p,
                      uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream);
/* out[r][c] = a[r][c] + b[(b_mod ? r % b_mod : r)][c] */
int sedt_add(const void* a, const void* b, void* out, int rows, int cols, int b_mod, int dtype, void* stream);
/* out = srcs[0] + ... + srcs[n-1] (n <= 8 equally shaped tensors of the compute dtype, numel a multiple of 8; srcs is a HOST array of
 * device pointers): the gradient of a tensor with several consumers in one launch */
int sedt_add_n(const void* const* srcs, int n, void* out, int64_t numel, int dtype, void* stream);
/* dtype conversion f32 <-> compute dtype, elementwise over n */
int sedt_cast(const void* in, int in_dtype, void* out, int out_dtype, int64_t n, void* stream);
/* out = y > 0 ? g : 0 (ReLU backward), compute dtype, elementwise over n */
int sedt_relu_mask(const void* g, const void* y, void* out, int64_t n, int dtype, void* stream);
/* up to 8 strided 2-D copies in one launch (jobs: HOST array, copied into the kernel arguments): `outer` items of `inner` bytes (multiple of 4),
 * src_stride / dst_stride bytes apart.  The clip-range split of the stacked head outputs [L][B][Q][C] between the two criterion calls of the
 * mean-teacher step (reference engine.py:134-165 computes a supervised and an unsupervised loss on two forwards; here ONE student forward
 * covers both clip sets) and the merge of their gradients are one launch each. */
typedef struct SedtCopyJob {
  const void* src;
  void* dst;
  int64_t src_stride, dst_stride;
  int32_t outer, inner;
  int32_t blk0;           /* filled by the library */
  int32_t pad_;
} SedtCopyJob;
int sedt_copy2d(const SedtCopyJob* jobs, int njobs, void* stream);
/* SP-SEDT decoder input, reference sedt/spsedt.py:48-69 in one launch each way.  patch [B*P][D] (compute dtype: patch2query of the pooled patch
 * features), query f32 [Q][D] (query_embed.weight rows from `start`), out [B*Q][D] token-major (compute dtype):
 *   train: out[b][q] = 2 * query[q] + keep(q, b) * patch[b][q / qpp]     eval: out[b][q] = query[q] + patch[b][q / qpp]
 * keep_in f32 [Q][B] (may be NULL): the Bernoulli(1 - ratio) query-patch mask of spsedt.py:65; NULL = drawn from the counter hash of
 * (seed (+ *seed_ptr), q * B + b), ratio <= 0 keeps every patch.  keep_out f32 [Q][B] (may be NULL) receives the mask used.
 * Backward: d_patch [B*P][D] (compute dtype, may be NULL) = sum over the qpp queries of a patch of keep * g; d_query f32 [Q][D] =
 * (train ? 2 : 1) * sum_b g[b][q], a fixed-order sum (four consecutive clip ranges added in clip order, then the four partials in range
 * order).  D a multiple of 8 (forward); a multiple of 64, <= 256 (backward). */
int sedt_spsedt_dec_in(const void* patch, const float* query, const float* keep_in, float* keep_out, void* out, int B, int Q, int P, int qpp,
                       int D, int train, float ratio, uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream);
int sedt_spsedt_dec_in_bwd(const void* g, const float* keep, void* d_patch, float* d_query, int B, int Q, int P, int qpp, int D, int train,
                           int dtype, void* stream);
/* the FFN activation "gelu" of reference sedt/transformer.py:423-431 (F.gelu, erf form) with the FFN's dropout behind it
 * (transformer.py:187 / :203: dropout(activation(linear1(x)))):  a = keep(seed, e) * gelu(h) / (1-p);
 * backward: out = keep(seed, e) * g * gelu'(h) / (1-p).  h is the saved PRE-activation.  Elementwise over n (a multiple of 8) in the
 * compute dtype; the keep decision is the GEMM epilogue's (seed, element index) hash; p = 0: no dropout */
int sedt_gelu_fwd(const void* h, void* a, int64_t n, float p, uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream);
int sedt_gelu_bwd(const void* g, const void* h, void* out, int64_t n, float p, uint32_t seed, const uint32_t* seed_ptr, int dtype,
                  void* stream);
/* y = g * s * (1 - s) (sigmoid backward), all f32 */
int sedt_sigmoid_grad(const float* g, const float* s, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------ LayerNorm (width D = 256)
 * y = (x-mean)*rstd*gamma+beta ; y2 = y + add (optional) ; saves mean/rstd.  eps = 1e-5. */
int sedt_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* add, void* y, void* y2,
                       float* mean, float* rstd, int rows, int D, int dtype, void* stream);
/* dx = dres + LN'(dy (+dy2)); dgamma/dbeta: f32[D] (deterministic two-stage reduction) */
int sedt_layernorm_bwd(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean,
                       const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta, float* scratch,
                       size_t scratch_bytes, int rows, int D, int dtype, void* stream);
size_t sedt_layernorm_bwd_scratch(int rows, int D);
/* same, with a second output dx_drop = dropout-backward of dx under (drop_p, seed): the gradient entering the sub-layer
 * whose dropped output fed this LayerNorm (post-norm layers, reference transformer.py:186-189) - saves the separate
 * sedt_dropout_grad launch.  dx_drop may be NULL.  dres2 (may be NULL): a second gradient added to dx - the share of another
 * consumer of x (a decoder layer's output also feeds the shared final LayerNorm, transformer.py:134-147). */
int sedt_layernorm_bwd_drop(const void* dy, const void* dy2, const void* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dres, const void* dres2, void* dx, float* dgamma, float* dbeta,
                            float* scratch, size_t scratch_bytes, int rows, int D, void* dx_drop, float drop_p, uint32_t seed,
                            const uint32_t* seed_ptr, int dtype, void* stream);
/* second stage of sedt_layernorm_bwd on its own (call sedt_layernorm_bwd with dgamma = dbeta = NULL first): reduces the
 * per-workgroup partial sums in `scratch` to dgamma / dbeta.  Lets a caller move the parameter-gradient half off the
 * critical path of the backward pass (another stream). */
int sedt_layernorm_bwd_final(const float* scratch, int rows, int D, float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------ attention (head dim 32)
 * rows are batch-first: q row = b*Lq + i, k/v row = b*Lk + j, head h occupies columns [h*32, h*32+32).
 * kpm: uint8 [B][Lk] (1 = padded key), amask: f32 [Lq][Lk] additive (may hold -inf), both optional.
 * probabilities are dropped with (drop_p, seed); lse [B][H][Lq] f32 saved for backward. */
int sedt_attention_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       void* o, int64_t ldo, float* lse, const uint8_t* kpm, const float* amask, int B, int H,
                       int Lq, int Lk, float drop_p, uint32_t seed, const uint32_t* seed_ptr, int dtype, void* stream);
int sedt_attention_bwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                       const void* o, int64_t ldo, const void* dout, int64_t lddo, const float* lse,
                       const uint8_t* kpm, const float* amask, void* dq, int64_t lddq, void* dk, int64_t lddk,
                       void* dv, int64_t lddv, int B, int H, int Lq, int Lk, float drop_p, uint32_t seed,
                       const uint32_t* seed_ptr, int dtype, void* stream);

/* ------------------------------------------------------------------ position encoding
 * pos[b][h*W+w][c] for the (B,H,W) uint8 mask (1 = padded): sine over the time axis H only,
 * normalised, scale 2*pi, temperature 10000, eps 1e-6, D features interleaved sin/cos. */
int sedt_posenc(const uint8_t* mask, void* pos, int B, int H, int W, int D, int dtype, void* stream);
/* nearest resize of the (B,Hin,Win) padding mask to (B,Hout,Wout): F.interpolate default */
int sedt_mask_resize(const uint8_t* in, uint8_t* out, int B, int Hin, int Win, int Hout, int Wout, void* stream);

/* ------------------------------------------------------------------ backbone pieces */
/* FrozenBatchNorm fold: scale = w*rsqrt(rv+1e-5), bias = b - rm*scale, n channels */
int sedt_bn_fold(const float* w, const float* b, const float* rm, const float* rv, float* scale, float* bias, int n,
                 void* stream);
/* conv weight (Cout,Cin,KH,KW) f32 -> fwd pack [Cout][taps][Cin] and (optional) dgrad pack
 * [Cin][taps][Cout] * bnscale[Cout], both in the compute dtype */
int sedt_pack_conv(const float* w, int Cout, int Cin, int taps, const float* bnscale, void* wf, void* wb, int dtype,
                   void* stream);
/* stem: conv0 (1->3, 1x1, bias) folded into conv1 (3->64, 7x7, s2, p3):
 * prep builds Wcat[64][128] = [Weff(49) | 0(15) | Beff(49) | 0(15)]; im2col builds
 * col[B*Ho*Wo][128] = [x taps | 0 | in-bounds indicators | 0]. */
int sedt_stem_prep(const float* w0, const float* b0, const float* w1, void* wcat, int dtype, void* stream);
int sedt_stem_im2col(const float* x, void* col, int B, int H, int W, int dtype, void* stream);
/* Direct 3x3 convolution, stride 1, pad 1, over NHWC bf16 tokens x [B*H*W][C] with C = 64 channels in and out on a W = 16 wide
 * map (conv2 of the layer1 Bottlenecks, torchvision resnet50 behind sedt/backbone.py:98-111): y = act(conv(x, w) * scale + bias),
 * or, with flip = 1 and w = the dgrad pack [Cin][taps][Cout] (BN scale folded in), the input gradient masked by (mask > 0).
 * w: bf16 [64 out][9 taps][64 in]; scale / bias / mask may be null; relu: 0 / 1. */
int sedt_conv3x3_c64(const void* x, const void* w, int flip, const float* scale, const float* bias, int relu, const void* mask, void* y,
                     int B, int H, int W, int C, void* stream);
/* The whole stem in one launch (bf16, W = 64 mel bands): conv0 o conv1 7x7/s2 o FrozenBN (scale, bias) o ReLU o max-pool 3x3/s2
 * (sedt/backbone.py:98-111 + torchvision stem) from x f32 [B][H][64] to pool bf16 [B*Hp*16][64] + argmax bytes idx (may be
 * null); wcat = sedt_stem_prep's bf16 [64][128]; s1_out (optional, tests): the un-pooled activation [B*Ho*32][64]. */
int sedt_stem_pool_fwd(const float* x, const void* wcat, const float* scale, const float* bias, void* pool, uint8_t* idx,
                       void* s1_out, int B, int H, int W, void* stream);
/* weight gradient of the folded stem convolution from the POOLED gradient g (max-pool backward through the ReLU evaluated on
 * the fly): slab f32 [nslab][64][128] partial sums in wcat's layout, nslab = sedt_stem_pool_wgrad_slabs(B, H); reduce with
 * sedt_wgrad_reduce_bias(slab, nslab, 64, 1, 128, bn_scale, G, ...) and feed G to sedt_stem_conv0_grad */
int sedt_stem_pool_wgrad_slabs(int B, int H);
int sedt_stem_pool_wgrad(const float* x, const void* g, const uint8_t* idx, const void* pool, float* slab, int nslab, int B, int H,
                         int W, void* stream);
/* dw0[c] = sum w1[co,c,tap]*G[co][tap], db0[c] = sum w1[co,c,tap]*G[co][64+tap]; G f32 [64][128] */
int sedt_stem_conv0_grad(const float* G, const float* w1, float* dw0, float* db0, void* stream);
/* 3x3 stride-2 pad-1 max pooling over NHWC; idx (uint8 argmax tap) saved for backward */
int sedt_maxpool_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int dtype, void* stream);
/* dx = (sum of dy routed by idx) * (relu_src > 0 if relu_src) */
int sedt_maxpool_bwd(const void* dy, const uint8_t* idx, const void* relu_src, void* dx, int B, int H, int W, int C,
                     int dtype, void* stream);
/* same with the ReLU mask taken from the pooled output y (NHWC at the pooled resolution; the selected element passed the
 * ReLU exactly when y > 0): pass relu_src = NULL.  Reads a quarter of the bytes and the un-pooled activation need not be kept. */
int sedt_maxpool_bwd_y(const void* dy, const uint8_t* idx, const void* relu_src, const void* y, void* dx, int B, int H, int W,
                       int C, int dtype, void* stream);
/* global average pool over NHWC pixels: out[b][c] = mean_p x[b][p][c] (f32 out) */
int sedt_avgpool(const void* x, float* out, int B, int P, int C, int dtype, void* stream);

/* ------------------------------------------------------------------ optimizer (engine.py:77-80)
 * Flat f32 buffers.  sumsq: f32[1] workspace holding the squared global grad norm (computed by
 * sedt_sumsq), clip_coef = min(1, max_norm/(sqrt(sumsq)+1e-6)) as torch.clip_grad_norm_. */
int sedt_sumsq(const float* g, int64_t n, float* sumsq, float* scratch, size_t scratch_bytes, int accumulate,
               void* stream);
size_t sedt_sumsq_scratch(int64_t n);
int sedt_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, void* stream);

/* multi-tensor form: `table` is a DEVICE array of chunks (<= 65536 elements each) covering every parameter tensor;
 * one launch updates them all.  sedt_multi_sumsq writes the squared global gradient norm to sumsq[0]
 * (partial: f32[nchunks] scratch); sedt_multi_adamw applies clip_grad_norm_(max_norm) + AdamW per chunk with the
 * chunk's own lr / weight decay (the reference's two parameter groups, train_sedt.py:234-240). */
typedef struct SedtChunk {
  void* p;
  const void* g;
  void* m;
  void* v;
  int32_t n;
  float lr;
  float wd;
  int32_t gflags; /* bit 0: g addresses bf16 elements (sedt_multi_sumsq / sedt_multi_adamw reading a bf16 flat gradient buffer) */
} SedtChunk;
/* table-driven forms of sedt_bn_fold / sedt_pack_conv: `jobs` is a DEVICE array; one launch serves every FrozenBN layer /
 * every weight tensor of the model.  Pack: one job per tensor (taps <= 9), e0 = index of the tensor's first workgroup
 * (ascending; a tensor takes ceil(Cout/32)*ceil(Cin/32) workgroups, each an LDS-staged 32x32xtaps tile); nblocks = total */
typedef struct SedtBnJob {
  const float* w;
  const float* b;
  const float* rm;
  const float* rv;
  float* scale;
  float* bias;
  int32_t n;
  int32_t pad_;
} SedtBnJob;
typedef struct SedtPackJob {
  const float* w;
  const float* bnscale; /* may be null */
  void* wf;             /* [Cout][taps][Cin] or null */
  void* wb;             /* [Cin][taps][Cout] (* bnscale) or null */
  int64_t e0;
  int32_t ne, Cout, Cin, taps;
} SedtPackJob;
int sedt_multi_bn_fold(const SedtBnJob* jobs, int njobs, void* stream);
int sedt_multi_pack(const SedtPackJob* jobs, int njobs, int nblocks, int dtype, void* stream);

/* ------------------------------------------------------------------ x-stationary ("slab") transformer kernels (csrc/slab.h)
 * A workgroup owns a slab of 32 token rows for a whole chain of sub-layers: activations stay in LDS as the B operand of
 * v_mfma_f32_32x32x16_bf16, and only the WEIGHTS move - from L2 straight into registers as A operands - in a fragment-major
 * packing: W [N][K] (nn.Linear layout; N % 32 == 0, K % 32 == 0) as 1 KB blocks, block (n / 32, k / 16) at byte
 * ((n / 32) * (K / 16) + k / 16) * 1024, inside it lane l = 32 * ((k % 16) / 8) + n % 32 owns 8 consecutive k (16 bytes).
 * sedt_pack_frag packs any number of f32 master weights into that form (wf) and / or the same form of W^T (wb: the operand of the
 * input-gradient chain) in one launch; nblocks = sum over the jobs of (N / 32) * (K / 32), blk0 = first block of a job. */
typedef struct SedtFragJob {
  const float* w; /* f32 master [N][K]; with src_bf16 a bf16 matrix [N][K] (a packed convolution operand of sedt_multi_pack) */
  void* wf;       /* fragment-major W, bf16, or null */
  void* wb;       /* fragment-major W^T, bf16, or null */
  int32_t N, K;
  int32_t blk0, src_bf16;
} SedtFragJob;
int sedt_pack_frag(const SedtFragJob* jobs, int njobs, int nblocks, void* stream);
/* The pre-norm encoder layer (sedt/transformer.py:192-204; nn.MultiheadAttention arithmetic as in sedt_attention_fwd) in TWO launches:
 *   sedt_encoder_qkv_fwd:      xn = LayerNorm1(x); q | k = (xn + pos) Wqk^T + b; v = xn Wv^T + b
 *   sedt_encoder_attn_ffn_fwd: ctx = dropout(softmax(q k^T / sqrt(32) + key padding)) v over the clip's S keys;
 *                              x1 = x + dropout(ctx Wo^T + bo); x1n = LayerNorm2(x1);
 *                              x2 = x1 + dropout(dropout(relu(x1n W1^T + b1)) W2^T + b2)
 * x, pos, x2 [B*S][256] bf16 contiguous; qk [B*S][512], v [B*S][256] bf16; weights fragment-major (sedt_pack_frag), biases and
 * LayerNorm parameters f32; kpm [B][S] (1 = padding) or null.  Training by-products - what sedt_layernorm_bwd, sedt_attention_bwd and
 * the weight-gradient GEMMs read; all or none (null: a no-grad forward): xn, xnp [B*S][256], mean, rstd [B*S]; ctx, x1, x1n
 * [B*S][256], lse [B][8][S], mean2, rstd2 [B*S], h [B*S][FF] (the dropped ReLU output).  Dropout decisions are the counter hashes of
 * the unfused chain: attention ((b*8 + h)*S + q)*S + k under seed_attn, out-proj row*256 + col under seed_o, hidden row*FF + col
 * under seed_h, FFN output row*256 + col under seed_f (each + *seed_ptr).  Envelope (sedt_encoder_slab_ok): bf16, d_model 256,
 * 8 heads, S <= 128, FF a multiple of 512. */
/* (SedtPrefetch: declared with sedt_multi_wgrad_reduce above) */
int sedt_encoder_slab_ok(int D, int H, int S, int FF, int dtype);
/* The input-gradient chain of the same layer, two launches around sedt_attention_bwd (weights as fragment-major W^T, the `wb` of
 * sedt_pack_frag):
 *   sedt_encoder_ffn_bwd:  g2 = dropout'(gx2); gh = (g2 W2) [h > 0] / (1 - p); g_x1n = gh W1; gx1 = LayerNorm2'(g_x1n) + gx2;
 *                          g1 = dropout'(gx1); gctx = g1 Wo          (g2 may be null when drop_p == 0: it equals gx2; g1 likewise)
 *   sedt_encoder_qkv_bwd:  gx = LayerNorm1'(dq|dk Wqk + dv Wv) + gx1
 * g2, gh, g1 are the left operands of the weight-gradient GEMMs of linear2, linear1, out_proj; ln_part [slabs][512] = per-slab
 * sums of (dy * xhat | dy) whose column sums are the LayerNorm gamma / beta gradients (slabs = B * ceil(S / 32)). */

/* An identity Bottleneck of ResNet layer1 / layer2 (torchvision v1.5 block behind sedt/backbone.py:97-113; cin -> planes -> planes -> cin,
 * stride 1, no downsample branch) in ONE launch, and its input-gradient chain in one more (csrc/bneck.hip; a workgroup per strip of 8
 * image rows, the planes-channel intermediates never leave the CU).  M = B*H*W pixels, NHWC.
 *   sedt_bneck_fwd: a = relu(s1 (x W1^T) + b1); b = relu(s2 conv3x3(a, W2) + b2); y = relu(s3 (b W3^T) + b3 + x)
 *     x, y [M][cin] bf16; w1/w2/w3_frag = sedt_pack_frag (src_bf16) of the forward operands of sedt_multi_pack ([Cout][taps][Cin]);
 *     s*, b* the folded FrozenBN scale / bias.  Training by-products, each group may be null: abits_out, bbits_out [M][planes/8] = sign
 *     bits of a, b (bit c % 8 of byte c / 8) - all sedt_bneck_bwd needs of them, 1/16 of the bytes; a_out, b_out [M][planes] - what the
 *     weight-gradient GEMMs of a trainable block read; bits_out [M][cin/8] = sign bits of y.
 *   sedt_bneck_bwd: gb = (gy W3s) [b > 0]; ga = conv3x3^T(gb, W2s) [a > 0]; gx = (ga W1s + gy) [x > 0]
 *     w*t_frag = sedt_pack_frag (src_bf16) of the dgrad operands of sedt_multi_pack ([Cin][taps][Cout], BN scale folded in); gy is the
 *     gradient w.r.t. y already masked by [y > 0]; abits / bbits from the forward; xbits = sign bits of the block input, or null (no mask).
 *     gb_out, ga_out [M][planes] (both or none): the two intermediate gradients, left operands of the weight-gradient GEMMs of a
 *     trainable block (layer2; layer1 is frozen in the reference, backbone.py:60-62, and needs neither).  gx == null: the chain stops
 *     at ga (ga_out required, w1t_frag / xbits ignored) - for a block whose first convolution has another shape (layer1's block 0: the
 *     caller finishes with its own two input-gradient GEMMs).
 * Envelope (sedt_bneck_ok): bf16, stride 1, dilation 1, no downsample, and (cin, planes, W) = (256, 64, 16) or (512, 128, 8). */
int sedt_bneck_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype);
int sedt_bneck_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const float* s1, const float* b1,
                   const float* s2, const float* b2, const float* s3, const float* b3, void* a_out, void* b_out, uint8_t* abits_out,
                   uint8_t* bbits_out, uint8_t* bits_out, int cin, int planes, int W, int B, int H, void* stream);
int sedt_bneck_bwd(const void* gy, void* gx, const void* w3t_frag, const void* w2t_frag, const void* w1t_frag, const uint8_t* abits,
                   const uint8_t* bbits, const uint8_t* xbits, void* gb_out, void* ga_out, int cin, int planes, int W, int B, int H,
                   void* stream);

/* The same pair for an identity Bottleneck of layer3 (1024 -> 256 -> 256 -> 1024 on a map 4 columns wide; csrc/bneck3.hip): one workgroup
 * per strip of 8 rows x 4 columns = 32 pixels, every workgroup streams all 2.2 MB of the block's weights from L2 while its activations
 * stay in LDS - a ~25 us floor set by the L2 -> CU stream, which beats the three per-op launches only while the strips cover the chip
 * about once: sedt_bneck3_ok additionally wants 192 <= B * ceil(H / 8) <= 512.  Arguments as for sedt_bneck_fwd / sedt_bneck_bwd (cin,
 * planes, W implied); abits / bbits [M][32], bits [M][128]. */
int sedt_bneck3_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int B, int H, int dtype);
int sedt_bneck3_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const float* s1, const float* b1,
                    const float* s2, const float* b2, const float* s3, const float* b3, void* a_out, void* b_out, uint8_t* abits_out,
                    uint8_t* bbits_out, uint8_t* bits_out, int B, int H, const SedtPrefetch* pf, void* stream);
int sedt_bneck3_bwd(const void* gy, void* gx, const void* w3t_frag, const void* w2t_frag, const void* w1t_frag, const uint8_t* abits,
                    const uint8_t* bbits, const uint8_t* xbits, void* gb_out, void* ga_out, int B, int H, const SedtPrefetch* pf,
                    void* stream);

/* The first Bottleneck of layer1 (64 -> 64 -> 64 -> 256, 1x1 projection 64 -> 256 on the skip path, stride 1, map 16 columns wide) in ONE
 * forward launch (csrc/bneck.hip: bneck0_fwd_kernel): a = relu(s1 (x W1^T) + b1); b = relu(s2 conv3x3(a, W2) + b2);
 * y = relu(s3 (b W3^T) + b3 + bf16(sd (x Wd^T) + bd)).  x [M][64], y [M][256] bf16 NHWC (M = B*H*16); w*_frag as for sedt_bneck_fwd (wd = the
 * projection); training by-products a_out / b_out [M][64], abits_out / bbits_out [M][8] (their sign bits; each pair both or none) and
 * bits_out [M][32] = sign bits of y, each may be null.  The block's backward: sedt_bneck_bwd with gx == null (gy -> gb -> ga in one
 * launch), then the two input-gradient GEMMs of conv1 and the projection.  Envelope (sedt_bneck0_ok): bf16, cin 64, planes 64, W 16,
 * stride 1, dilation 1, WITH the downsample branch. */
int sedt_bneck0_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype);
int sedt_bneck0_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const void* wd_frag, const float* s1,
                    const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* sd, const float* bd,
                    void* a_out, void* b_out, uint8_t* abits_out, uint8_t* bbits_out, uint8_t* bits_out, int B, int H, void* stream);

/* The first Bottleneck of layer2 (256 -> 128 at the input resolution, 3x3 stride 2, 128 -> 512, stride-2 1x1 projection 256 -> 512 on the
 * skip path; input map 16 columns wide, output 8) in ONE forward launch (csrc/bneck.hip: bneck2_fwd_kernel).  x [B*H*16][256], y
 * [B*H2*8][512] with H2 = (H - 1) / 2 + 1; w*_frag / s* / b* as for sedt_bneck0_fwd.  Training by-products, each may be null: a_out
 * [B*H*16][128] and b_out [B*H2*8][128] (both or none; what the per-op backward of the block reads), bits_out [B*H2*8][64] = sign bits
 * of y.  Envelope (sedt_bneck2_ok): bf16, cin 256, planes 128, W 16, stride 2, dilation 1, WITH the downsample branch. */
int sedt_bneck2_ok(int cin, int planes, int W, int stride, int dil, int has_downsample, int dtype);
int sedt_bneck2_fwd(const void* x, void* y, const void* w1_frag, const void* w2_frag, const void* w3_frag, const void* wd_frag, const float* s1,
                    const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* sd, const float* bd,
                    void* a_out, void* b_out, uint8_t* bits_out, int B, int H, void* stream);

/* The prediction heads on the stacked decoder output hs [L*B*Qp][256] bf16 (sedt/sedt.py:88-95, 398-409) in ONE launch each way
 * (csrc/heads_slab.hip; a workgroup per 32 rows): class logits cls [rows][C1] = hs wc^T + bc, boxes [rows][2] = sigmoid(W3 relu(W2
 * relu(W1 hs + b1) + b2) + b3), audio tags at [B][CA] = sigmoid(wa hs + ba) on query 0 of the last layer (CA = 0: no such head).
 * wc, w3, wa: the f32 master weights; W1, W2 fragment-major (sedt_pack_frag).  h1 / h2 [rows][256] bf16: the hidden activations, kept
 * for the backward (both or none).  sedt_heads_bwd: from the gradients of the three outputs (g_at may be null) the input gradient dhs,
 * the hidden-layer gradients g_h1 / g_h2 (left operands of the W1 / W2 weight-gradient GEMMs) and part [slabs][NG * 257], NG = C1 +
 * CA + 2: per 32-row slab the [NG][256] weight-gradient sums of the head outputs (class rows, then audio-tag rows, then the 2 box
 * rows) followed by their NG bias sums; the sums over the slabs are the gradients of wc | wa | w3 and bc | ba | b3.  Envelope: bf16, d = 256, C1, CA <= 16. */
int sedt_heads_slab_ok(int D, int C1, int CA, int dtype);
int sedt_heads_fwd(const void* x, const float* wc, const float* bc, const void* w1_frag, const float* b1, const void* w2_frag,
                   const float* b2, const float* w3, const float* b3, const float* wa, const float* ba, float* cls, float* box, float* at,
                   void* h1, void* h2, int L, int B, int Qp, int C1, int CA, void* stream);
size_t sedt_heads_bwd_part_floats(int L, int B, int Qp, int C1, int CA);
int sedt_heads_bwd(const void* x, const void* h1, const void* h2, const float* box, const float* at, const float* g_cls,
                   const float* g_box, const float* g_at, const float* wc, const float* w3, const float* wa, const void* w2t_frag,
                   const void* w1t_frag, void* dhs, void* g_h1, void* g_h2, float* part, int L, int B, int Qp, int C1, int CA, void* stream);

int sedt_encoder_ffn_bwd(const void* gx2, const void* h, const void* x1, const float* mean2, const float* rstd2,
                         const float* gamma2, const void* w2t_frag, const void* w1t_frag, const void* wot_frag, void* g2, void* gh,
                         void* gx1, void* g1, void* gctx, float* ln_part, int B, int S, int FF, float drop_p, uint32_t seed_f,
                         uint32_t seed_o, const uint32_t* seed_ptr, void* stream);
int sedt_encoder_qkv_bwd(const void* dqk, const void* dv, const void* x, const float* mean1, const float* rstd1,
                         const float* gamma1, const void* gx1, const void* wint_frag, void* gx, float* ln_part, int B, int S,
                         void* stream);
int sedt_encoder_qkv_fwd(const void* x, const void* pos, const float* gamma, const float* beta, const void* w_in_frag,
                         const float* b_in, void* qk, void* v, void* xn, void* xnp, float* mean, float* rstd, int B, int S,
                         const SedtPrefetch* pf, void* stream);
int sedt_encoder_attn_ffn_fwd(const void* x, const void* qk, const void* v, const uint8_t* kpm, const void* w_o_frag,
                              const float* b_o, const float* gamma2, const float* beta2, const void* w1_frag, const float* b1,
                              const void* w2_frag, const float* b2, void* x2, void* ctx, float* lse, void* x1, float* mean2,
                              float* rstd2, void* x1n, void* h, int B, int S, int FF, float drop_p, uint32_t seed_attn,
                              uint32_t seed_o, uint32_t seed_h, uint32_t seed_f, const uint32_t* seed_ptr, void* stream);

/* chunk.p[i] = chunk.g[i] for every chunk: packs all gradient tensors into one flat buffer (one launch) ahead of the RCCL
 * all-reduce of the data-parallel step.  mode bit 0: chunk.p[i] += chunk.g[i] instead (gradient accumulation over micro-batches,
 * engine.py:76, 174); bit 1: the flat buffer is bf16 (chunk.p addresses 2-byte elements; half the all-reduce bytes). */
int sedt_multi_gather(const SedtChunk* table, int nchunks, int mode, void* stream);
/* mean-teacher update of every tensor in one launch (utilities/utils.py:62-67, EMA.update):
 * chunk.m[i] = (1 - decay) * chunk.p[i] + decay * chunk.m[i]   (p = student parameter, m = shadow)
 * Non-finite guard (the reference aborts BEFORE the backward when the loss is NaN / inf: engine.py:70-73, 167-169; a captured
 * step cannot stop itself, its host only polls a flag now and then): `guard` is an optional device int32 word, the same one
 * SedtCriterion.nonfinite / sedt_feature_loss raise.  sedt_multi_sumsq also raises it when the squared gradient norm is not
 * finite and - step_ptr given - advances the device-side Adam step count unless the guard is up; sedt_multi_adamw and
 * sedt_multi_ema return without touching parameters, moments or the shadow weights while it is up.  So a bad batch leaves the
 * whole training state (student, optimizer, teacher) at its last good value until the host sees the flag and raises. */
int sedt_multi_ema(const SedtChunk* table, int nchunks, float decay, const int32_t* guard /* or null */, void* stream);
int sedt_multi_sumsq(const SedtChunk* table, int nchunks, float* partial, float* sumsq, int32_t* step_ptr /* or null: += 1 */,
                     int32_t* guard /* or null */, uint32_t* seed_word /* or null: the device dropout-seed word, += 1 */, void* stream);
int sedt_multi_adamw(const SedtChunk* table, int nchunks, const float* sumsq, float max_norm, float beta1, float beta2,
                     float eps, const int32_t* step_ptr /* device: 1-based step count */, const int32_t* guard /* or null */,
                     void* stream);

/* ------------------------------------------------------------------ device half of SetCriterion (sedt/sedt.py:161-283)
 * After the host matching the targets are dense: for dense layer d (0 = final decoder layer, d>=1 = aux layer d-1),
 * strong clip b < ns, query q: tc = target class (as f32), coef = CE weight multiplier, wbox = box-loss weight (0 =
 * unmatched), tbox = target (centre, length).  logits [L][B][Q][C+1] / boxes [L][B][Q][2] are the model's stacked head
 * outputs; layer_of[d] is the slice that dense layer d reads.  One launch writes
 *   out[4d+0..3] = loss_ce, loss_bbox, loss_giou, cardinality_error of dense layer d (unweighted, as the reference logs)
 *   out[4L] = class-error hits, out[4L+1] = matched count, out[4L+2] = loss_weak,
 *   out[4L+3] = weighted total (sum_k weight_k * loss_k), out[4L+4] = class_error, out[4L+5] = loss_weak_p
 * and the UNWEIGHTED per-term gradients: dlogits = d loss_ce_d / d logits, dboxes = d loss_bbox_d / d boxes, dboxes2 =
 * d loss_giou_d / d boxes (each layer slice holds the gradient of its own layer's loss), dat = d loss_weak / d at.
 * at / gt_weak / dat may be null together (model without audio-tag head).
 * --pooling models (sedt.py:96-119, sedt_pool_at below) add at_p [Bp][C], the pooled clip-level probabilities: loss_weak_p =
 * BCE(at_p[rows], gt_weak[rows]) (sedt.py:182-185) over the weak clips rows = [ns, n_lab), or - wp_all != 0, the reference's
 * weak_mask None - over all labelled clips [0, n_lab); an empty row range gives NaN, as the reference's mean over nothing.
 * dat_p [Bp][C] = d loss_weak_p / d at_p.  at_p / dat_p may be null together; at_p needs at (the reference forms gt there).
 * sedt_set_criterion_bwd combines them with the gradient g[4L+6] that reached `out`:
 *   glogits = (g[4d] + g[4L+3] w_ce[d]) dlogits,  gboxes = (g[4d+1] + g[4L+3] w_bbox[d]) dboxes + (g[4d+2] + ...) dboxes2,
 *   gat = (g[4L+2] + g[4L+3] w_weak) dat,  gat_p = (g[4L+5] + g[4L+3] w_weak_p) dat_p. */
#define SEDT_CRIT_MAXL 8
#define SEDT_CRIT_MAXOUT (4 * SEDT_CRIT_MAXL + 6)
#define SEDT_CRIT_MAXCARD 8192 /* limit on L * B */
typedef struct SedtCriterion {
  const float* logits;
  const float* boxes;
  const float* at;       /* [Bat][C] probabilities, rows < n_lab are labelled */
  const float* tc;       /* [L][ns][Q] */
  const float* coef;     /* [L][ns][Q] */
  const float* wbox;     /* [L][ns][Q] */
  const float* tbox;     /* [L][ns][Q][2] */
  const float* gt_weak;  /* [n_lab][C] */
  const float* tgt_len;  /* [B] */
  const float* num_boxes;    /* [1], or null: computed as sum(wbox[0]) */
  const float* empty_weight; /* [C+1] */
  float* dlogits;
  float* dboxes;
  float* dboxes2;
  float* dat;
  float* out; /* [4L+6] */
  int32_t L, B, ns, Q, C, n_lab, Bat;
  int32_t layer_of[SEDT_CRIT_MAXL];
  float w_ce[SEDT_CRIT_MAXL], w_bbox[SEDT_CRIT_MAXL], w_giou[SEDT_CRIT_MAXL];
  float w_weak;
  /* focal-loss variant (sedt/sedt.py:176, 211-218, 412-433; train_ss_sedt.py --focal_loss): fl != 0 replaces the weighted
   * cross-entropy by sigmoid_focal_loss over the C+1 logits (pos_weight = empty_weight) and the audio-tag BCE by
   * weak_focal_loss (sum over classes, mean over clips); alpha_fl < 0 disables the alpha_t factor (config.py:71-72). */
  int32_t fl;
  float alpha_fl, gamma_fl;
  /* optional device word: set to 1 (never cleared here) when the weighted total is not finite - the reference aborts on
   * such a loss (engine.py:70-73, 167-169); a graphed step polls this word instead of synchronising every step */
  int32_t* nonfinite;
  /* optional device words {ns, n_lab} that override the two fields above as the NUMBER of strong / labelled clips of this batch
   * (utilities/mixup.py:13-127 returns a new strong | weak split for every batch); ns / n_lab then are the capacities = the
   * strides of the dense tables.  Lets a captured step take batches of any split. */
  const int32_t* split;
  /* layout of logits / boxes (and of the gradients sedt_set_criterion_bwd returns): [L][B][Qs] rows, of which the Q queries
   * q0 .. q0 + Q - 1 take part in the losses (dec_at models run their heads over the audio-tag query 0 too: Qs = Q + 1, q0 = 1;
   * otherwise Qs = Q, q0 = 0).  The per-term gradient buffers dlogits / dboxes / dboxes2 stay compact ([L][B][Q]). */
  int32_t Qs, q0;
  float* total; /* optional: receives out[4L+3] as a separate scalar (its own autograd output: no select/backward-of-select) */
  const float* at_p; /* [Bp][C] pooled probabilities (--pooling), or null */
  float* dat_p;      /* [Bp][C] */
  float w_weak_p;
  int32_t wp_all, Bp;
} SedtCriterion;
size_t sedt_set_criterion_scratch(int L, int B, int Q); /* bytes of `scratch`: per-row loss terms, summed per layer in a fixed order */
int sedt_set_criterion(const SedtCriterion* args, float* scratch, void* stream);
/* g: gradient that reached out[4L+6] (or null), gtotal: gradient that reached the separate total scalar (or null) */
int sedt_set_criterion_bwd(const SedtCriterion* args, const float* g, const float* gtotal, float* glogits, float* gboxes,
                           float* gat, float* gat_p /* or null */, void* stream);

/* ------------------------------------------------------------------ --pooling variants of SEDT (sedt/sedt.py:47-61, 96-119)
 * at_p[b][c], c < C: the clip-level probability pooled over the Q event queries' class probabilities
 * y[b][q][c] = softmax(logits[b][q0 + q])[c] of the FINAL decoder layer (replaces nn.AdaptiveMaxPool2d / AdaptiveAvgPool2d((1, None)),
 * the attn_pooling closure and the weighted sum of sedt.py:98-100):
 *   SEDT_POOL_MAX  max_q y        SEDT_POOL_AVG  mean_q y
 *   SEDT_POOL_ATTN s = clamp(softmax_c(attn[b][q]), 1e-7, 1); sum_q s y / sum_q s   (attn = attn_dense_softmax(hs[-1]) [B][Q][C])
 *   SEDT_POOL_WSUM clip(sum_q y * boxes[b][q0 + q][1], 0, 1)                        (the predicted event length as weight)
 * logits [B][Qs][C+1] / boxes [B][Qs][2] are the head outputs over ALL Qs query rows (dec_at models: q0 = 1 skips the
 * audio-tag query).  sedt_pool_at_bwd: g [B][C] -> glogits [B][Qs][C+1] (zero rows outside the window), gboxes [B][Qs][2]
 * (WSUM; may be null otherwise), gattn [B][Q][C] (ATTN; may be null otherwise); max sends the gradient to the first maximal
 * query, clamp / clip pass it where the value lies inside the closed interval (torch semantics).  Q * (C+1) <= 4096. */
enum { SEDT_POOL_MAX = 0, SEDT_POOL_AVG = 1, SEDT_POOL_ATTN = 2, SEDT_POOL_WSUM = 3 };
typedef struct SedtPoolAt {
  const float* logits;
  const float* boxes; /* WSUM, else may be null */
  const float* attn;  /* ATTN, else may be null */
  int32_t B, Qs, q0, Q, C, mode;
} SedtPoolAt;
int sedt_pool_at(const SedtPoolAt* args, float* at_p /* [B][C] */, void* stream);
int sedt_pool_at_bwd(const SedtPoolAt* args, const float* g, float* glogits, float* gboxes, float* gattn, void* stream);

/* ------------------------------------------------------------------ SP-SEDT feature-reconstruction loss (sedt/sedt.py:263-283)
 * For dense layer d, strong clip b, query q matched to patch tidx (wbox > 0):
 *   row = sum_f (normalize(pred[layer_of[d]][b][q])[f] - normalize(gt[b*P + tidx])[f])^2,   normalize = x / max(|x|_2, 1e-12)
 *   out[d] = sum over matched rows / num_boxes.  dpred receives d out[d] / d pred (unweighted; zero rows where unmatched).
 * pred / dpred [L][B][Q][F] f32, gt [B*P][F] f32 (treated as a constant: the SP-SEDT backbone is frozen,
 * train_spsedt.py:50), wbox / tidx [L][ns][Q] as written by sedt_match_targets, num_boxes [1] device scalar.
 * rowloss [L*ns*Q] scratch.  Deterministic (fixed summation order).  sedt_scale_layers: x[l][...] *= g[d] + gtot[0]*w[d] with d = idx[l] (idx = inverse of layer_of: dpred is in the model's
 * layer order, the loss vector in dense order). */
int sedt_feature_loss(const float* pred, const float* gt, const float* wbox, const float* tidx, const float* num_boxes,
                      const int32_t* layer_of /* host [L] */, const float* w /* device [L] or null */, int L, int B, int ns,
                      int Q, int P, int F, float* rowloss, float* out /* [L+1]: out[L] = sum_d w[d] out[d] */, float* dpred,
                      int32_t* nonfinite /* or null: set to 1 when out[L] is NaN / inf (see SedtCriterion.nonfinite) */,
                      const float* base /* or null */, float* total_out /* or null: *total_out = out[L] + base[0] - the step's weighted total
                      with SetCriterion's (sedt_set_criterion's `total`) folded in, so the sum costs no launch of its own */, void* stream);
int sedt_scale_layers(float* x, const float* g, const float* gtot, const float* w, const int32_t* idx /* host [L] or null */,
                      int L, int64_t per_layer, void* stream);
/* out[0] = sum_i x[i] (one workgroup, fixed order): num_boxes = sum of the final layer's box weights (sedt.py:322-324) */
int sedt_sum_f32(const float* x, int n, float* out, void* stream);

/* ------------------------------------------------------------------ device-side matching (sedt/matcher.py:41-133)
 * The Hungarian assignment of every (decoder layer, strong clip) and the dense targets sedt_set_criterion reads, in one
 * launch and without leaving the device (one wave per problem; Q <= 63 queries, <= 63 targets per clip).
 * Targets arrive concatenated over the batch: lab_cat[lab_off[b] .. lab_off[b+1]) = labels of clip b (all B clips; for a
 * strong clip the first n_b labels belong to its n_b boxes), box_cat[box_off[b] .. box_off[b+1]) = (centre, length) of
 * the events of strong clip b < ns, ratio_cat (optional) aligned with lab_cat.  Cost = w_bbox*L1 + w_class*cost_class
 * - w_giou*GIoU with cost_class = -softmax(logits)[class] (fl == 0) or the focal matching cost of matcher.py:73-78
 * (fl != 0); same optimum and tie-breaking as sedt_hungarian_batch.
 * Coefficients (matcher.py:124-132): normalize != 0: 1 / (number of queries matched to the same target); else with
 * ratio_cat: the k-th matched query of a clip (ascending query index) takes ratio[k] - POSITIONAL, as the reference
 * assigns them; else 1.
 * fine_tune != 0 (matcher.py:99-121; dense layer 0 only, aux layers keep the plain assignment as sedt.py:340 does): with
 * the localisation cost w_bbox*L1 - w_giou*GIoU, a Hungarian pair survives only if the query's closest target is nearer
 * than epsilon; every other query whose closest target is nearer than epsilon is matched to that target unless
 * u > alpha * n_gt / Q, u = ft_rand[b*Q + k] for the k-th such query of clip b (ascending query index; the reference
 * draws torch.rand in that order) or, when ft_rand is null, a counter hash of (ft_seed + *seed_ptr, b, k).
 * Outputs: tc/coef/wbox/tidx [L][ns][Q], tbox [L][ns][Q][2], tgt_len [B], gt_weak [n_lab][C] (may be null),
 * assign [L][ns][Q] int32 (may be null): index of the matched target within its clip or -1.
 * max_targets: capacity per clip the caller guarantees (LDS sizing; box_off differences must not exceed it). */
typedef struct SedtMatch {
  const float* logits;   /* [L][B][Q][C+1] */
  const float* boxes;    /* [L][B][Q][2] */
  const int64_t* lab_cat;
  const int32_t* lab_off; /* [B+1] */
  const float* box_cat;
  const int32_t* box_off; /* [ns+1] */
  const float* ratio_cat; /* or null */
  float* tc;
  float* coef;
  float* wbox;
  float* tbox;
  float* tidx;
  float* tgt_len;
  float* gt_weak;
  int32_t* assign;
  int32_t L, B, ns, Q, C, n_lab, max_targets;
  int32_t layer_of[SEDT_CRIT_MAXL];
  float w_class, w_bbox, w_giou;
  int32_t fl, fine_tune, normalize;
  float alpha_fl, gamma_fl, epsilon, alpha;
  const float* ft_rand;     /* [ns][Q] or null */
  uint32_t ft_seed;
  const uint32_t* seed_ptr; /* or null */
  const int32_t* split;     /* or null: device {ns, n_lab} of this batch, see SedtCriterion.split (box_off then has ns + 1
                               entries for the capacity ns; clips >= split[0] get "no target" rows) */
  int32_t Qs, q0;           /* logits / boxes rows are [L][B][Qs]; queries q0 .. q0 + Q - 1 are matched (see SedtCriterion) */
} SedtMatch;
int sedt_match_targets(const SedtMatch* args, void* stream);

/* ------------------------------------------------------------------ PostProcess + pseudo labels on the device
 * sedt_postprocess (sedt/sedt.py:355-396): softmax over the C+1 logits of every query; optional fusion with clip-level
 * tags (tags [B][C] as f32 0/1, at_m 1/2/3 exactly as the reference: at_m 2/3 lift the best query of every class to
 * `threshold`, at_m 1/2 multiply the class scores by the tags); scores [B][Q] = max class score, labels [B][Q] (int64) =
 * its class (first maximum), boxes_out [B][Q][2] = (onset, offset) * sizes[b], or the (centre, length) input when is_semi.
 *
 * sedt_pseudo_labels (engine.py:300-348, get_pseudo_labels): tags = at >= thr[class]; PostProcess(at_m = 1, is_semi); an
 * event survives when score >= thr[label] and length > min_len; per clip the survivors are ordered by descending score
 * (ties: lower query first) and, with del_overlap, an event is dropped when an already kept event of the same class
 * overlaps it in time.  Results are written in the flat layout sedt_match_targets reads: lab_cat / box_cat compacted
 * over the clips in kept order, lab_off / box_off [B+1], counter[C] += kept events per class (int32; pseudo_labels_counter
 * of the reference).  cap = capacity of lab_cat / box_cat in events.  at may be null (no tag gating). */
int sedt_postprocess(const float* logits, const float* boxes, const float* tags, const float* sizes, int B, int Q, int C,
                     int at_m, float threshold, int is_semi, float* scores, int64_t* labels, float* boxes_out, void* stream);
int sedt_pseudo_labels(const float* logits, const float* boxes, const float* at, const float* thr, float min_len, int B,
                       int Q, int C, int del_overlap, int64_t* lab_cat, float* box_cat, int32_t* lab_off, int32_t* box_off,
                       int32_t* counter, int cap, void* stream);

/* ------------------------------------------------------------------ input side on the device (utilities/BoxTransforms.py,
 * utilities/mixup.py)
 * sedt_box_transform: the per-clip feature transforms composed as get_transforms does (BoxTransforms.py:454-490):
 *   ApplyLog (librosa.amplitude_to_db: 10 log10(max(1e-10, x^2)), clamped at clip-max - 80 dB; apply_log = 0 skips it)
 *   -> PadOrTrunc to `frames` rows (zero rows appended) -> TimeMask (rows [tm_t0, tm_t0 + tm_t) zeroed)
 *   -> FreqMask (bands [fm_f0, fm_f0 + fm_f) := their mean over the clip when fill_mean, else fill_const; only if fm_on)
 *   -> FreqShift (np.roll by fs_shift bands, wrapped bands zeroed) -> (x - mean[band]) / std[band] in float64 (Scaler.normalize;
 *   mean/std may be null: no normalisation).  The random parameters are drawn by the caller exactly as the reference's
 *   transform classes draw them; aug = B records of 8 int32 {nframes_raw, tm_t, tm_t0, fm_f, fm_f0, fm_on, fs_shift, 0}.
 *   amp [B][raw_stride][F] f32 (rows >= nframes_raw ignored), out [B][1][frames][F] f32.  One workgroup per clip, the clip
 *   stays in LDS between the passes (frames * F * 4 bytes <= 160 KB).
 * sedt_mixup: out[i] = lam * x1[src1] + (1 - lam) * x2[src2] (mode 0), x1[src1] (1) or x2[src2] (2); jobs = n_out records
 *   {int32 src1, src2, mode; float lam} (mixup.py:35, 142: the label bookkeeping stays on the host). */
int sedt_box_transform(const float* amp, int64_t raw_stride, const void* aug, const double* mean, const double* stdv, int B,
                       int frames, int F, int apply_log, int fill_mean, float fill_const, float* out, void* stream);
int sedt_mixup(const float* x1, const float* x2, const void* jobs, int n_out, int64_t clip_elems, float* out, void* stream);
/* sedt_mixup_targets: the LABEL half of mixup_label_unlabel (utilities/mixup.py:129-196; call site engine.py:150-153, between the
 * teacher and the student forward of semi_train) without leaving the device.  Set 1 = the labelled targets (flat tables as
 * sedt_match_targets reads them: lab1/lab_off1 [B1+1], box1/box_off1 [ns1+1], optional ratio1 aligned with lab1, optional split1 =
 * device {ns, n_lab}), set 2 = the pseudo targets sedt_pseudo_labels wrote (all B2 clips with boxes).  For clip i < mix_num:
 * more than max_events events together -> the pseudo target if it has events, else the labelled one; otherwise labels / boxes
 * concatenated (labelled first) with ratio lam[0] for the labelled and lam[1] (= 1 - lam as the host rounds it) for the pseudo
 * labels, unless two events of one class overlap in time -> the labelled target (and clip) replaces the unlabelled one.
 * Clips >= mix_num keep their pseudo target.  Outputs: merged tables (capacity cap events), ratio_out (1 where the reference has no
 * 'ratio'), jobs = B2 sedt_mixup records {i, i, mode, lam} producing the matching features from (x1 = labelled, x2 = unlabelled). */
int sedt_mixup_targets(const int64_t* lab1, const int32_t* lab_off1, const float* box1, const int32_t* box_off1, const float* ratio1,
                       const int32_t* split1, int B1, int ns1, const int64_t* lab2, const int32_t* lab_off2, const float* box2,
                       const int32_t* box_off2, int B2, const float* lam, int mix_num, int max_events, int64_t* lab_out,
                       int32_t* lab_off_out, float* box_out, int32_t* box_off_out, float* ratio_out, int cap, void* jobs,
                       void* stream);

/* SP-SEDT query patches (utilities/BoxTransforms.py:315-360, Query.transform_label; boxes from DataLoad.py:57-77 are turned into
 * row ranges on the host): for each of n_patches jobs {clip, s_idx, e_idx, 0} (int32 x 4, device memory) crop rows [s_idx, e_idx)
 * of data f32 [B][T][F], min-max normalise, quantise to 8 bits, resize to 128 rows with Pillow's bilinear resampling arithmetic,
 * de-normalise -> out f32 [n_patches][128][F], bit-identical to the reference pipeline; fixed != 0: copy the 128 rows as they are
 * (fixed_patch_size). */
int sedt_query_patches(const float* data, int B, int T, int F, const void* jobs, int n_patches, int fixed, float* out, void* stream);

/* ------------------------------------------------------------------ host-side matching (sedt/matcher.py:95)
 * HOST pointers.  cost [nlayers][nclips][Q][Nt] f32; clip b owns columns [col_off[b], col_off[b]+ncols[b]).
 * assign [nlayers][nclips][Q]: index (within the clip) of the target matched to query q, or -1.
 * Same optimum as scipy.optimize.linear_sum_assignment (min(Q, n_b) pairs per problem). */
int sedt_hungarian_batch(const float* cost, int nlayers, int nclips, int Q, int Nt, const int32_t* col_off,
                         const int32_t* ncols, int32_t* assign);

#ifdef __cplusplus
}
#endif
#endif
