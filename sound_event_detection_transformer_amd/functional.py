"""torch.autograd.Function wrappers that run the SEDT blocks on the HIP kernels.

Layouts: activations are NHWC "token" matrices [B*H*W, C] in the compute dtype (f32 parity mode
or bf16 throughput mode); transformer tokens are batch-first [B*S, d].  Parameters stay ordinary
f32 nn.Parameters (read through data_ptr() on every call, so EMA swaps / load_state_dict /
optimizer updates are always seen); gradients are returned as f32 tensors, so DDP hooks,
clip_grad_norm_ and AdamW work unchanged.  Every Function saves only what its backward needs and
recomputes dropout masks from (seed, element index).
"""
import torch
from torch.autograd import Function

from . import ops, packing, runtime
from .lib import F32, BF16
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID, ConvGeom


def _dt_of(t):
    return F32 if t.dtype == torch.float32 else BF16


def _as(t, dt):
    """cast an incoming activation/gradient to the compute dtype (contiguous)"""
    want = torch.float32 if dt == F32 else torch.bfloat16
    t = t.contiguous()
    return t if t.dtype == want else ops.cast(t, dt)


def _wf(dt, w):
    """forward operand [N][K] of a linear weight"""
    return w if dt == F32 else ops.cast(w, BF16)


def _wb(dt, w):
    """dgrad operand [K][N] (= W^T) of a linear weight"""
    return ops.pack_conv(dt, w, want_fwd=False)[1]


def _prep_linear(dt, w, need_b):
    """(forward operand [N][K], dgrad operand [K][N] or None): from the model's PackPlan when one is active (two launches
    for ALL weights), else packed on the spot"""
    hit = packing.lookup(w)
    if hit is not None:
        return hit[0], hit[1]
    return _wf(dt, w), (_wb(dt, w) if need_b else None)


def _prep_conv(dt, w, bn):
    """(wf, wb, scale, bias) of one conv + FrozenBN"""
    hit = packing.lookup(w)
    if hit is not None and hit[2] is not None:
        return hit
    sc, bi = ops.bn_fold(bn[0], bn[1], bn[2], bn[3])
    wf, wb = ops.pack_conv(dt, w, bnscale=sc)
    return wf, wb, sc, bi


# ======================================================================================= generic
class LinearFn(Function):
    """y = act(x @ W^T + b); act in {none, relu, sigmoid}; optional f32 output (heads)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, out_f32, dt, relu_input=False, x_bits=None):
        """relu_input: x is a post-ReLU tensor whose producer expects a gradient already masked by x > 0 (fused into the
        dgrad epilogue here instead of a separate pass in the producer's backward); x_bits: that mask as the producer's 1-bit
        image (uint8 [rows, cols / 8]) - read instead of x itself in the backward"""
        x = _as(x, dt)
        ctx.relu_input = relu_input
        ctx.x_bits = x_bits if relu_input else None
        ctx.dt, ctx.act, ctx.out_f32 = dt, act, out_f32
        ctx.has_bias = bias is not None
        # heads with a handful of outputs (class 11, box 2, audio tag 10): direct kernels on the master weight
        ctx.skinny = ops.skinny_ok(weight.shape[0], weight.shape[1]) and out_f32 and weight.is_contiguous()
        if ctx.skinny:
            y = ops.skinny_linear_fwd(dt, x, weight, bias, act, True)
            ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
            return y
        wf, ctx.wb = _prep_linear(dt, weight, ctx.needs_input_grad[0])
        y = ops.linear(dt, x, wf, bias=bias, act=act, out_f32=out_f32)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        dt = ctx.dt
        gy = gy.contiguous()
        if ctx.skinny:
            gx, gw, gb = ops.skinny_linear_bwd(dt, gy.float() if gy.dtype != torch.float32 else gy, y, weight, x, ctx.act,
                                               mask=x if ctx.relu_input else None, need_gx=ctx.needs_input_grad[0],
                                               need_gw=ctx.needs_input_grad[1], need_gb=ctx.has_bias and ctx.needs_input_grad[2])
            return gx, gw, gb, None, None, None, None, None
        if ctx.act == ACT_SIGMOID:
            gy = ops.sigmoid_grad(gy.float() if gy.dtype != torch.float32 else gy, y.float() if y.dtype != torch.float32 else y)
        elif ctx.act == ACT_RELU:
            gy = ops.relu_mask(_dt_of(gy), gy, y)
        g = _as(gy, dt)
        gx = None
        if ctx.needs_input_grad[0]:
            if ctx.relu_input and ctx.x_bits is not None:
                gx = ops.linear(dt, g, ctx.wb, mask=ctx.x_bits, ldm=ctx.x_bits.stride(0), mask_bits=True)
            else:
                gx = ops.linear(dt, g, ctx.wb, mask=x, ldm=x.stride(0)) if ctx.relu_input else ops.linear(dt, g, ctx.wb)
        gb = torch.empty((g.shape[1],), device=g.device, dtype=torch.float32) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if ctx.needs_input_grad[1]:
            gw = ops.linear_wgrad(dt, g, x, bias_out=gb)
        else:
            gw = None
            if gb is not None:
                ops.colsum(dt, g, out=gb)
        return gx, gw, gb, None, None, None, None, None


class PoolAtFn(Function):
    """the --pooling variants (reference sedt.py:96-119) as one launch each way (csrc/pool_at.hip): the final decoder layer's class
    logits [B,Qp,C+1] (event queries q0 .. q0+Q-1) -> clip-level probabilities at_p [B,C]; ``boxes`` [B,Qp,2] feeds mode
    'weighted_sum', ``attn`` [B,Q,C] (attn_dense_softmax of the event queries) mode 'attn'."""

    @staticmethod
    def forward(ctx, logits, boxes, attn, mode, q0, Q):
        f32c = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()
        lg = f32c(logits)
        bx = f32c(boxes) if mode == 'weighted_sum' else None
        at = f32c(attn) if mode == 'attn' else None
        ctx.cfg = (mode, q0, Q)
        ctx.dts = (logits.dtype, None if boxes is None else boxes.dtype, None if attn is None else attn.dtype)
        ctx.save_for_backward(lg, bx, at)
        return ops.pool_at(lg, bx, at, mode, q0, Q)

    @staticmethod
    def backward(ctx, g):
        lg, bx, at = ctx.saved_tensors
        mode, q0, Q = ctx.cfg
        gl, gb, ga = ops.pool_at_bwd(lg, bx, at, mode, q0, Q, g)
        return (gl.to(ctx.dts[0]), None if gb is None else gb.to(ctx.dts[1]), None if ga is None else ga.to(ctx.dts[2]),
                None, None, None)


class HeadsFn(Function):
    """the prediction heads on the stacked decoder output hs [L,B,Qp,d] (reference sedt.py:88-95): class_embed on every row,
    bbox_embed (3-layer MLP + sigmoid) on every row, and - dec_at - weak_class_embed + sigmoid on query 0 of the last layer,
    as ONE autograd node.  hs has three consumers; as separate nodes their input gradients are summed by autograd (two
    elementwise launches) and the slices hs[-1, :, 0] / outputs[:, :, 1:] cost copy / fill launches in both directions.  Here the
    audio-tag head reads its rows strided, the outputs keep all Qp rows (the criterion kernels take the query window as (q0, Q)),
    and the backward builds ONE dhs: class head writes it, the box MLP's last dgrad adds through the GEMM epilogue's residual
    operand, the audio-tag head accumulates into its rows.
    Returns (class logits [L,B,Qp,C+1] f32, boxes [L,B,Qp,2] f32, at [B,C] f32 or None)."""

    @staticmethod
    def forward(ctx, hs, wc, bc, w1, b1, w2, b2, w3, b3, wa, ba, dt):
        Lh, B, Qp, d = hs.shape
        x = _as(hs, dt).view(Lh * B * Qp, d)
        tr = any(ctx.needs_input_grad)
        ctx.slab = None
        if ops.heads_slab_ok(dt, d, wc.shape[0], 0 if wa is None else wa.shape[0], hs.numel() // d) and wc.is_contiguous() and w3.is_contiguous():
            fr = [packing.lookup_frag(w) for w in (w1, w2)]
            if all(f is not None for f in fr):
                # all heads in ONE launch (csrc/heads_slab.hip)
                cls, box, at, hh = ops.heads_fwd(x, wc, bc, fr[0][0], b1, fr[1][0], b2, w3, b3, wa, ba, Lh, B, Qp, train=tr)
                ctx.dt, ctx.dims, ctx.has_at = dt, (Lh, B, Qp, d), wa is not None
                if tr:
                    ctx.slab = (fr[0][1], fr[1][1])
                    ctx.save_for_backward(x, hh[0], hh[1], box, at, wc, w3, wa)
                cls, box = cls.view(Lh, B, Qp, -1), box.view(Lh, B, Qp, 2)
                return (cls, box, at) if at is not None else (cls, box)
        cls = ops.skinny_linear_fwd(dt, x, wc, bc, ACT_NONE, True)
        w1f, w1b = _prep_linear(dt, w1, tr)
        w2f, w2b = _prep_linear(dt, w2, tr)
        h1 = ops.linear(dt, x, w1f, bias=b1, act=ACT_RELU)
        h2 = ops.linear(dt, h1, w2f, bias=b2, act=ACT_RELU)
        box = ops.skinny_linear_fwd(dt, h2, w3, b3, ACT_SIGMOID, True)
        at = None
        if wa is not None:
            xa = x.view(Lh, B, Qp, d)[Lh - 1, :, 0, :]                       # [B, d] rows Qp*d apart: no copy
            at = ops.skinny_linear_fwd(dt, xa, wa, ba, ACT_SIGMOID, True)
        ctx.dt, ctx.dims, ctx.wb = dt, (Lh, B, Qp, d), (w1b, w2b)
        ctx.has_at = wa is not None
        ctx.save_for_backward(x, h1, h2, box, at, wc, w3, wa)
        cls, box = cls.view(Lh, B, Qp, -1), box.view(Lh, B, Qp, 2)
        return (cls, box, at) if at is not None else (cls, box)

    @staticmethod
    def backward(ctx, g_cls, g_box, g_at=None):
        x, h1, h2, box, at, wc, w3, wa = ctx.saved_tensors
        dt = ctx.dt
        Lh, B, Qp, d = ctx.dims
        M = Lh * B * Qp
        if ctx.slab is not None:
            f32c = lambda t: None if t is None else (t.contiguous() if t.dtype == torch.float32 else t.contiguous().float())
            C1, CA = wc.shape[0], 0 if wa is None else wa.shape[0]
            w1t, w2t = ctx.slab
            dhs, g_h1, g_h2, part = ops.heads_bwd(x, h1, h2, box.view(M, 2), at, f32c(g_cls).view(M, -1), f32c(g_box).view(M, 2),
                                                  f32c(g_at) if ctx.has_at else None, wc, w3, wa, w2t, w1t, Lh, B, Qp)
            rb = ops.ReduceBatch()
            d_b2 = torch.empty((d,), device=x.device, dtype=torch.float32)
            d_w2 = ops.linear_wgrad(dt, g_h2, h1, bias_out=d_b2, batch=rb)
            d_b1 = torch.empty((d,), device=x.device, dtype=torch.float32)
            d_w1 = ops.linear_wgrad(dt, g_h1, x, bias_out=d_b1, batch=rb)
            NG = C1 + CA + 2
            tot = torch.empty((NG * 257,), device=x.device, dtype=torch.float32)
            rb.add_colsum(part, part.shape[0], NG * 257, tot)
            rb.flush()
            tw, tb = tot[:NG * 256].view(NG, 256), tot[NG * 256:]
            d_wc, d_bc = tw[:C1], tb[:C1]
            d_w3, d_b3 = tw[C1 + CA:], tb[C1 + CA:]
            d_wa = d_ba = None
            if ctx.has_at:
                d_wa, d_ba = tw[C1:C1 + CA], tb[C1:C1 + CA]
            return (dhs.view(Lh, B, Qp, d), d_wc, d_bc, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_wa, d_ba, None)
        w1b, w2b = ctx.wb
        f32 = lambda t: t.contiguous() if t.dtype == torch.float32 else t.contiguous().float()
        rb = ops.ReduceBatch()
        # class head: writes the running input gradient
        dhs, d_wc, d_bc = ops.skinny_linear_bwd(dt, f32(g_cls).view(M, -1), None, wc, x, ACT_NONE, batch=rb)
        # box MLP: sigmoid' and the ReLU masks ride in the kernels; the last dgrad adds the running gradient (epilogue residual)
        g_h2, d_w3, d_b3 = ops.skinny_linear_bwd(dt, f32(g_box).view(M, 2), box.view(M, 2), w3, h2, ACT_SIGMOID, mask=h2, batch=rb)
        d_b2 = torch.empty((g_h2.shape[1],), device=x.device, dtype=torch.float32)
        d_w2 = ops.linear_wgrad(dt, g_h2, h1, bias_out=d_b2, batch=rb)
        g_h1 = ops.linear(dt, g_h2, w2b, mask=h1, ldm=h1.stride(0))
        d_b1 = torch.empty((g_h1.shape[1],), device=x.device, dtype=torch.float32)
        d_w1 = ops.linear_wgrad(dt, g_h1, x, bias_out=d_b1, batch=rb)
        dhs = ops.linear(dt, g_h1, w1b, res=dhs, ldr=dhs.stride(0))
        d_wa = d_ba = None
        if ctx.has_at:
            xa = x.view(Lh, B, Qp, d)[Lh - 1, :, 0, :]
            if g_at is None:
                d_wa, d_ba = torch.zeros_like(wa), torch.zeros(wa.shape[0], device=wa.device)
            else:
                ga = f32(g_at).view(B, -1)
                _, d_wa, d_ba = ops.skinny_linear_bwd(dt, ga, at.view(B, -1), wa, xa, ACT_SIGMOID,
                                                      gx_acc=dhs.view(Lh, B, Qp, d)[Lh - 1, :, 0, :], batch=rb)
        rb.flush()
        return (dhs.view(Lh, B, Qp, d), d_wc, d_bc, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_wa, d_ba, None)


class SplitClipsFn(Function):
    """clip ranges [0, n) and [n, B) of tensors laid out [..., B, ...] - the stacked head outputs [L, B, Q, C] and the audio-tag output
    [B, C] of ONE student forward over labelled + unlabelled clips, handed to the two criterion calls of the mean-teacher step
    (reference engine.py:134-165) - as contiguous tensors from ONE launch (sedt_copy2d); the backward writes the gradient parts into
    full-size tensors with ONE launch (every element is covered by exactly one part: no zero fill, no adds).  As autograd slices this
    is ~20 torch copy / fill / add launches per step.  Inputs: (n, batch_dim_of_each, *tensors); returns (first parts..., second parts...)"""

    @staticmethod
    def forward(ctx, n, dims, *ts):
        ctx.set_materialize_grads(False)
        ts = [t.contiguous() for t in ts]
        metas, outs_a, outs_b, jobs = [], [], [], []
        for t, d in zip(ts, dims):
            B = t.shape[d]
            outer = 1
            for s_ in t.shape[:d]:
                outer *= s_
            per = t.numel() // (outer * B) * t.element_size()             # bytes of one clip inside one outer item
            a = torch.empty(t.shape[:d] + (n,) + t.shape[d + 1:], device=t.device, dtype=t.dtype)
            b = torch.empty(t.shape[:d] + (B - n,) + t.shape[d + 1:], device=t.device, dtype=t.dtype)
            jobs.append((t.data_ptr(), a.data_ptr(), outer, n * per, B * per, n * per))
            jobs.append((t.data_ptr() + n * per, b.data_ptr(), outer, (B - n) * per, B * per, (B - n) * per))
            metas.append((tuple(t.shape), t.dtype, outer, per, B))
            outs_a.append(a)
            outs_b.append(b)
        ops.copy2d(jobs)
        ctx.n, ctx.metas = n, metas
        return tuple(outs_a + outs_b)

    @staticmethod
    def backward(ctx, *gs):
        k = len(ctx.metas)
        grads, jobs = [], []
        for i, (shape, dtype, outer, per, B) in enumerate(ctx.metas):
            ga, gb = gs[i], gs[k + i]
            if ga is None and gb is None:
                grads.append(None)
                continue
            n = ctx.n
            full = torch.empty(shape, device=(ga if ga is not None else gb).device, dtype=dtype)
            if ga is None or gb is None:
                full.zero_()
            if ga is not None:
                ga = ga.contiguous().to(dtype)
                jobs.append((ga.data_ptr(), full.data_ptr(), outer, n * per, n * per, B * per))
            if gb is not None:
                gb = gb.contiguous().to(dtype)
                jobs.append((gb.data_ptr(), full.data_ptr() + n * per, outer, (B - n) * per, (B - n) * per, B * per))
            grads.append(full)
            ctx.keep = getattr(ctx, 'keep', []) + [ga, gb]
        if jobs:
            ops.copy2d(jobs)
        ctx.keep = None
        return (None, None) + tuple(grads)


class FanoutFn(Function):
    """n handles on one tensor for n independent consumers (SP-SEDT's stacked decoder output feeds the class head, the box MLP and the
    feature-alignment MLP: spsedt.py:79-83).  As plain autograd the consumers' input gradients are added pair by pair (n - 1 elementwise
    launches of torch); here the backward is ONE sedt_add_n over the gradients that arrived."""

    @staticmethod
    def forward(ctx, x, n, dt):
        ctx.set_materialize_grads(False)
        ctx.dt = dt
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_as(g, ctx.dt) for g in gs if g is not None]
        if not gs:
            return None, None, None
        if len(gs) == 1 or gs[0].numel() % 8:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None, None
        return ops.add_n(ctx.dt, gs), None, None


class SpDecInFn(Function):
    """SP-SEDT decoder input (reference sedt/spsedt.py:48-69) as one launch each way: patch queries [B*P, D] (compute dtype) + query
    embedding rows [Q, D] (f32 parameter view) -> token-major [B*Q, D]; the Bernoulli query-patch mask is injected (``keep`` [Q, B]) or
    drawn inside the launch (a counter hash, fresh per graph replay through the device seed word)"""

    @staticmethod
    def forward(ctx, patch, query, keep, B, Q, P, qpp, train, ratio, dt):
        patch = _as(patch, dt)
        q = query.detach().float().contiguous()
        sp = runtime.seed_ptr(patch.device) if (train and keep is None and ratio > 0) else None
        out, keep_used = ops.spsedt_dec_in(dt, patch, q, B, Q, P, qpp, train, ratio, keep, runtime.next_seed() if sp is not None else 0, sp)
        ctx.cfg = (B, Q, P, qpp, train, dt, patch.dtype)
        ctx.save_for_backward(keep_used)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Q, P, qpp, train, dt, _ = ctx.cfg
        keep, = ctx.saved_tensors
        d_patch, d_query = ops.spsedt_dec_in_bwd(dt, _as(g, dt), keep, B, Q, P, qpp, train, need_patch=ctx.needs_input_grad[0])
        return d_patch, d_query, None, None, None, None, None, None, None, None


class GradAccumulator(object):
    """side channel for a tensor that several autograd nodes of one chain consume (the decoder layers all read the encoder
    memory and the query position embedding): instead of each node returning its share - which autograd then adds pair by pair,
    one elementwise launch per pair - the nodes hand their shares over here as they run (backward order is the reverse of the
    forward order, so it is deterministic); every node but the LAST to run returns None for that input, and the last one returns
    the total: folded through GEMM epilogues (``pending`` = a running sum the next producing GEMM takes as its residual operand)
    or summed in one launch (``parts``)."""

    def __init__(self, n_nodes):
        self.n, self.seen, self.pending, self.parts = n_nodes, 0, None, []

    def last(self):
        """call once per node, in its backward; True for the node that must return the total"""
        self.seen += 1
        done = self.seen == self.n
        if done:
            self.seen = 0
        return done

    def take_parts(self):
        p, self.parts = self.parts, []
        return p

    def take_pending(self):
        p, self.pending = self.pending, None
        return p


class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, dt):
        x = _as(x, dt)
        y, _, mean, rstd = ops.layernorm_fwd(dt, x, gamma, beta)
        ctx.dt = dt
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ops.layernorm_bwd(ctx.dt, _as(gy, ctx.dt), x, gamma, mean, rstd)
        return dx, dg, db, None


class StackViewFn(Function):
    """the decoder layers write their outputs straight into consecutive row ranges of ONE buffer (DecoderLayerFn, cfg['out']);
    this node hands that buffer to the shared LayerNorm as a single [n*R, D] tensor - the torch.cat without the copy.
    Backward: the row ranges of the incoming gradient (views)."""

    @staticmethod
    def forward(ctx, buf, share, *outs):
        """share (dict or None): when given, the gradient slices of all layers but the last are handed to the NEXT layer's
        backward through share['stack_g'] (it adds them inside its first-LayerNorm backward kernel) instead of being returned -
        a layer's output feeds the next layer AND this stack, and autograd would add the two gradients in a launch of its own"""
        R = outs[0].shape[0]
        for i, o in enumerate(outs):
            if o.data_ptr() != buf.data_ptr() + i * R * buf.shape[1] * buf.element_size() or o.shape != outs[0].shape:
                raise RuntimeError('StackViewFn: the layer outputs are not the row ranges of the stacking buffer')
        ctx.R, ctx.n, ctx.share = R, len(outs), share
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        parts = [g[i * ctx.R:(i + 1) * ctx.R] for i in range(ctx.n)]
        if ctx.share is not None:
            ctx.share['stack_g'] = parts
            return (None, None) + (None,) * (ctx.n - 1) + (parts[-1],)
        return (None, None) + tuple(parts)


class BroadcastRowsFn(Function):
    """out[b*Q + q] = table[q] in the compute dtype (the decoder's query position embedding for every clip); backward = the
    column sums of the gradient seen as [B, Q*C] (one launch instead of an expand/reshape copy forward and a strided
    reduction backward)"""

    @staticmethod
    def forward(ctx, table, zeros, B, dt):
        ctx.dt, ctx.B, ctx.shape = dt, B, tuple(table.shape)
        return ops.add(dt, zeros, _as(table.detach(), dt), table.shape[0])

    @staticmethod
    def backward(ctx, g):
        Q, C = ctx.shape
        g = _as(g, ctx.dt)
        return ops.colsum(ctx.dt, g.view(ctx.B, Q * C)).view(Q, C), None, None, None


class CastFn(Function):
    """x in the compute dtype; the gradient is cast back to x's dtype (one conversion however many consumers x has)"""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.src_dtype = x.dtype
        return _as(x, dt)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.src_dtype), None


class AddFn(Function):
    """out = a + b[(r % b_mod)] ; gradient flows to a only (b is the constant position encoding)."""

    @staticmethod
    def forward(ctx, a, b, b_mod, dt):
        ctx.set_materialize_grads(False)       # (no consumer returned a gradient: pass None on, not a zero-filled tensor)
        return ops.add(dt, _as(a, dt), _as(b, dt), b_mod)

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None


# ======================================================================================= attention + FFN sub-blocks
def _mha_fwd(dt, q_in, k_in, v_in, same_qk, w_in, b_in, w_out, b_out, res, B, H, Lq, Lk, kpm, amask, p, seeds, train=True):
    """res + drop(out_proj(attention(q_in Wq, k_in Wk, v_in Wv)));  returns (out, saved)."""
    E = w_in.shape[1]
    wf, wb_in = _prep_linear(dt, w_in, train)
    wf_o, wb_o = _prep_linear(dt, w_out, train)
    if same_qk:                       # q_in is k_in: one GEMM for Q|K (N = 2E); the V projection rides in the same launch
        qk, v = ops.linear_group(dt, [(q_in, wf[:2 * E], dict(bias=b_in[:2 * E])), (v_in, wf[2 * E:], dict(bias=b_in[2 * E:]))])
        q, k = qk[:, :E], qk[:, E:]
    else:
        q, k, v = ops.linear_group(dt, [(q_in, wf[:E], dict(bias=b_in[:E])), (k_in, wf[E:2 * E], dict(bias=b_in[E:2 * E])),
                                        (v_in, wf[2 * E:], dict(bias=b_in[2 * E:]))])
        qk = None
    sp = runtime.seed_ptr(q_in.device) if p > 0 else None
    ctxv, lse = ops.attention_fwd(dt, q, k, v, B, H, Lq, Lk, kpm, amask, p, seeds[0], sp)
    out = ops.linear(dt, ctxv, wf_o, bias=b_out, drop_p=p, seed=seeds[1], seed_ptr=sp, res=res, ldr=res.stride(0))
    saved = dict(wb_in=wb_in, wb_o=wb_o, q_in=q_in, k_in=k_in, v_in=v_in, same_qk=same_qk, qk=qk, q=q, k=k, v=v, ctxv=ctxv, lse=lse,
                 dims=(B, H, Lq, Lk), kpm=kpm, amask=amask, p=p, seeds=seeds)
    return out, saved


def _drop_args(s, device):
    """(p, seed, seed_ptr) of the dropout on a block's output, for ops.layernorm_bwd(drop=...)"""
    p = s['p']
    return (p, s['seeds'][1], runtime.seed_ptr(device) if p > 0 else None)


def _mha_bwd(dt, s, g_out, w_in, w_out, need_q=True, need_k=True, need_v=True, batch=None, g_dropped=None, kv_fused=False,
             res_kv=None):
    """g_out = grad wrt the block output (the residual branch is the caller's business); g_dropped = the same gradient
    already through the output dropout (ops.layernorm_bwd(drop=...)).
    returns g_q_in, g_k_in, g_v_in, d_in_proj_weight, d_in_proj_bias, d_out_w, d_out_b"""
    B, H, Lq, Lk = s['dims']
    E = w_in.shape[1]
    p = s['p']
    sp = runtime.seed_ptr(g_out.device) if p > 0 else None
    if g_dropped is not None:
        g1 = g_dropped
    else:
        g1 = ops.dropout_grad(dt, g_out, p, s['seeds'][1], sp) if p > 0 else g_out
    d_bo = torch.empty((E,), device=g_out.device, dtype=torch.float32)
    d_wo = ops.linear_wgrad(dt, g1, s['ctxv'], bias_out=d_bo, batch=batch, param=w_out)
    g_ctx = ops.linear(dt, g1, s['wb_o'])
    td = g_out.dtype
    if s['same_qk']:
        dqk = torch.empty((B * Lq, 2 * E), device=g_out.device, dtype=td)
        dq, dk = dqk[:, :E], dqk[:, E:]
        dv = torch.empty((B * Lk, E), device=g_out.device, dtype=td)
    elif kv_fused:
        # key and value inputs are the same tensor up to a constant: dK | dV side by side, one dgrad GEMM over K = 2E
        dq = torch.empty((B * Lq, E), device=g_out.device, dtype=td)
        dkv = torch.empty((B * Lk, 2 * E), device=g_out.device, dtype=td)
        dk, dv = dkv[:, :E], dkv[:, E:]
    else:
        dq = torch.empty((B * Lq, E), device=g_out.device, dtype=td)
        dk = torch.empty((B * Lk, E), device=g_out.device, dtype=td)
        dv = torch.empty((B * Lk, E), device=g_out.device, dtype=td)
    ops.attention_bwd(dt, s['q'], s['k'], s['v'], s['ctxv'], g_ctx, s['lse'], B, H, Lq, Lk, dq, dk, dv, s['kpm'], s['amask'],
                      p, s['seeds'][0], sp)
    d_win = ops._sink(w_in, (3 * E, E))                     # (the data-parallel steppers' flat gradient buffer, ops.grad_sink)
    if d_win is None:
        d_win = torch.empty((3 * E, E), device=g_out.device, dtype=torch.float32)
    d_bin = torch.empty((3 * E,), device=g_out.device, dtype=torch.float32)
    wb = s['wb_in']                                         # [E][3E]
    g_q = g_k = g_v = None
    dgrads = []                                             # independent input gradients: one grouped launch
    if s['same_qk']:
        ops.linear_wgrad(dt, dqk, s['q_in'], out=d_win[:2 * E], bias_out=d_bin[:2 * E], batch=batch)
        if need_q or need_k:
            dgrads.append(('q', dqk, wb[:, :2 * E]))        # grad wrt the shared q/k input
    else:
        ops.linear_wgrad(dt, dq, s['q_in'], out=d_win[:E], bias_out=d_bin[:E], batch=batch)
        ops.linear_wgrad(dt, dk, s['k_in'], out=d_win[E:2 * E], bias_out=d_bin[E:2 * E], batch=batch)
        if need_q:
            dgrads.append(('q', dq, wb[:, :E]))
        if need_k and not kv_fused:
            dgrads.append(('k', dk, wb[:, E:2 * E]))
    ops.linear_wgrad(dt, dv, s['v_in'], out=d_win[2 * E:], bias_out=d_bin[2 * E:], batch=batch)
    if kv_fused and not s['same_qk']:
        # grad wrt the shared key/value source, returned in the value slot; res_kv: the running sum of the other consumers'
        # shares of that source's gradient, added in the GEMM epilogue
        dgrads.append(('v', dkv, wb[:, E:], {} if res_kv is None else dict(res=res_kv, ldr=res_kv.stride(0))))
    elif need_v:
        dgrads.append(('v', dv, wb[:, 2 * E:]))
    if dgrads:
        got = dict(zip([d[0] for d in dgrads], ops.linear_group(dt, [(d[1], d[2], dict(d[3]) if len(d) > 3 else {}) for d in dgrads])))
        g_q, g_k, g_v = got.get('q'), got.get('k'), got.get('v')
    return g_q, g_k, g_v, d_win, d_bin, d_wo, d_bo


def _ffn_fwd(dt, x_in, w1, b1, w2, b2, res, p, seeds, train=True, out=None, act='relu'):
    """res + drop(linear2(drop(act(linear1(x_in)))));  act 'relu' rides in linear1's epilogue with its dropout (the saved h = drop(relu(.))
    is its own backward mask); 'gelu' (transformer.py:423-431) keeps the PRE-activation and runs gelu + dropout as one launch"""
    sp = runtime.seed_ptr(x_in.device) if p > 0 else None
    wf1, wb1 = _prep_linear(dt, w1, train)
    wf2, wb2 = _prep_linear(dt, w2, train)
    if act == 'gelu':
        h_pre = ops.linear(dt, x_in, wf1, bias=b1)
        h = ops.gelu_fwd(dt, h_pre, p, seeds[0], sp)
        out = ops.linear(dt, h, wf2, out, bias=b2, drop_p=p, seed=seeds[1], seed_ptr=sp, res=res, ldr=res.stride(0))
        return out, dict(x_in=x_in, h=h, h_pre=h_pre, p=p, seeds=seeds, wb1=wb1, wb2=wb2)
    h = ops.linear(dt, x_in, wf1, bias=b1, act=ACT_RELU, drop_p=p, seed=seeds[0], seed_ptr=sp)
    out = ops.linear(dt, h, wf2, out, bias=b2, drop_p=p, seed=seeds[1], seed_ptr=sp, res=res, ldr=res.stride(0))
    return out, dict(x_in=x_in, h=h, p=p, seeds=seeds, wb1=wb1, wb2=wb2)


def _ffn_bwd(dt, s, g_out, w1, w2, res_for_gx=None, batch=None, g_dropped=None):
    """returns g_x_in (+ res_for_gx), dW1, db1, dW2, db2"""
    p = s['p']
    if g_dropped is not None:
        g2 = g_dropped
    else:
        g2 = ops.dropout_grad(dt, g_out, p, s['seeds'][1], runtime.seed_ptr(g_out.device)) if p > 0 else g_out
    d_b2 = torch.empty((g2.shape[1],), device=g2.device, dtype=torch.float32)
    d_w2 = ops.linear_wgrad(dt, g2, s['h'], bias_out=d_b2, batch=batch, param=w2)
    if s.get('h_pre') is not None:          # gelu: d_hidden = drop'(g2 @ W2) * gelu'(h_pre)
        gh = ops.gelu_bwd(dt, ops.linear(dt, g2, s['wb2']), s['h_pre'], p, s['seeds'][0], runtime.seed_ptr(g2.device) if p > 0 else None)
    else:
        # d_hidden = (g2 @ W2) * [h > 0] / (1-p): h = drop(relu(.)) is positive exactly where kept and active
        gh = ops.linear(dt, g2, s['wb2'], mask=s['h'], ldm=s['h'].stride(0), alpha=1.0 / (1.0 - p) if p > 0 else 1.0)
    d_b1 = torch.empty((gh.shape[1],), device=gh.device, dtype=torch.float32)
    d_w1 = ops.linear_wgrad(dt, gh, s['x_in'], bias_out=d_b1, batch=batch, param=w1)
    if res_for_gx is not None:
        gx = ops.linear(dt, gh, s['wb1'], res=res_for_gx, ldr=res_for_gx.stride(0))
    else:
        gx = ops.linear(dt, gh, s['wb1'])
    return gx, d_w1, d_b1, d_w2, d_b2


# ======================================================================================= encoder layer
class EncoderLayerFn(Function):
    """reference sedt/transformer.py:155-212 (forward_pre :192-204, forward_post :177-190), batch-first tokens.
    params: in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias, linear1.w, linear1.b, linear2.w, linear2.b,
            norm1.w, norm1.b, norm2.w, norm2.b"""

    @staticmethod
    def forward(ctx, x, pos, kpm, amask, cfg, *P):
        dt = cfg['dt']
        (w_in, b_in, w_o, b_o, w1, b1, w2, b2, g1, be1, g2, be2) = P
        B, S, H = cfg['B'], cfg['S'], cfg['H']
        p = cfg['dropout'] if cfg['training'] else 0.0
        seeds = [runtime.next_seed() for _ in range(4)]
        x = _as(x, dt)
        pos = _as(pos, dt)
        sv = {}
        tr = any(ctx.needs_input_grad)
        fr = None
        act = cfg.get('act', 'relu')
        if cfg['pre_norm'] and act == 'relu' and ops.encoder_slab_ok(dt, x.shape[1], H, S, w1.shape[0], amask, B):
            fr = [packing.lookup_frag(w) for w in (w_in, w_o, w1, w2)]
            fr = fr if all(f is not None for f in fr) else None
        if fr is not None:
            # the whole layer in TWO launches on the x-stationary slab kernels (csrc/enc_slab.hip); by-products for the per-op backward
            # kernels are written only when a backward will follow
            x, pos = x.contiguous(), pos.contiguous()
            sp = runtime.seed_ptr(x.device) if p > 0 else None
            _, wb_in = _prep_linear(dt, w_in, tr)
            _, wb_o = _prep_linear(dt, w_o, tr)
            _, wb1 = _prep_linear(dt, w1, tr)
            _, wb2 = _prep_linear(dt, w2, tr)
            qk, v, by1 = ops.encoder_qkv_fwd(x, pos, g1, be1, fr[0][0], b_in, B, S, train=tr, prefetch=(fr[1][0], fr[2][0], fr[3][0]))
            x2, by2 = ops.encoder_attn_ffn_fwd(x, qk, v, kpm, fr[1][0], b_o, g2, be2, fr[2][0], b1, fr[3][0], b2, B, S, w1.shape[0],
                                               p, seeds, sp, train=tr)
            if tr:
                xn, xnp, m1, r1 = by1
                ctxv, lse, x1, m2, r2, x1n, h = by2
                E = x.shape[1]
                sv['mha'] = dict(wb_in=wb_in, wb_o=wb_o, q_in=xnp, k_in=xnp, v_in=xn, same_qk=True, qk=qk, q=qk[:, :E], k=qk[:, E:], v=v,
                                 ctxv=ctxv, lse=lse, dims=(B, H, S, S), kpm=kpm, amask=None, p=p, seeds=seeds[0:2])
                sv['ffn'] = dict(x_in=x1n, h=h, p=p, seeds=seeds[2:4], wb1=wb1, wb2=wb2)
                sv.update(x=x, x1=x1, m1=m1, r1=r1, m2=m2, r2=r2, slab=(fr[0][1], fr[1][1], fr[2][1], fr[3][1]))
                # the backward of the layer BELOW runs right after this layer's: its first operands get touched by this layer's closing
                # reduce launch.  This layer leaves (W2^T, W1^T, Wo^T) - what its own backward streams first - for the layer above
                chain = cfg.get('chain')
                if chain is not None:
                    sv['bwd_next'], chain.top = chain.top, (fr[3][1], fr[2][1], fr[1][1])
            ctx.sv, ctx.cfg, ctx.P = sv, cfg, P
            return x2
        if cfg['pre_norm']:
            xn, xnp, m1, r1 = ops.layernorm_fwd(dt, x, g1, be1, add_t=pos)
            x1, sv['mha'] = _mha_fwd(dt, xnp, xnp, xn, True, w_in, b_in, w_o, b_o, x, B, H, S, S, kpm, amask, p, seeds[0:2], tr)
            x1n, _, m2, r2 = ops.layernorm_fwd(dt, x1, g2, be2)
            x2, sv['ffn'] = _ffn_fwd(dt, x1n, w1, b1, w2, b2, x1, p, seeds[2:4], tr, act=act)
            sv.update(x=x, x1=x1, m1=m1, r1=r1, m2=m2, r2=r2)
        else:
            xp = ops.add(dt, x, pos)
            t, sv['mha'] = _mha_fwd(dt, xp, xp, x, True, w_in, b_in, w_o, b_o, x, B, H, S, S, kpm, amask, p, seeds[0:2], tr)
            x1, _, m1, r1 = ops.layernorm_fwd(dt, t, g1, be1)
            t2, sv['ffn'] = _ffn_fwd(dt, x1, w1, b1, w2, b2, x1, p, seeds[2:4], tr, act=act)
            x2, _, m2, r2 = ops.layernorm_fwd(dt, t2, g2, be2)
            sv.update(t=t, t2=t2, m1=m1, r1=r1, m2=m2, r2=r2)
        ctx.sv, ctx.cfg, ctx.P = sv, cfg, P
        return x2

    @staticmethod
    def backward(ctx, gx2):
        sv, cfg, P = ctx.sv, ctx.cfg, ctx.P
        dt = cfg['dt']
        (w_in, b_in, w_o, b_o, w1, b1, w2, b2, g1, be1, g2, be2) = P
        gx2 = _as(gx2, dt)
        # weight gradients: this layer's own grouped launch + reduce launch, or - per-op path inside a captured stepper - the stack's
        # shared batch, launched once when the backward leaves the stack (ops.WgradShare).  The slab path keeps its per-layer reduce
        # launch: that launch also touches the weights the layer below streams first
        share = cfg.get('wg_share') if not (sv.get('slab') is not None and ops.SLAB_ENC_BWD) else None
        rb = share.batch() if share is not None else ops.ReduceBatch()
        if sv.get('slab') is not None and ops.SLAB_ENC_BWD:
            # the input-gradient chain on the slab kernels: [FFN + LayerNorm2 + out-proj] | attention core | [in-proj + LayerNorm1];
            # the weight gradients stay the layer's one grouped GEMM launch + one reduce launch
            B, S, H = cfg['B'], cfg['S'], cfg['H']
            wint, wot, w1t, w2t = sv['slab']
            ff, mh = sv['ffn'], sv['mha']
            p = ff['p']
            dev = gx2.device
            sp = runtime.seed_ptr(dev) if p > 0 else None
            E = gx2.shape[1]
            g2d, gh, gx1, g1d, gctx, part2 = ops.encoder_ffn_bwd(gx2.contiguous(), ff['h'], sv['x1'], sv['m2'], sv['r2'], g2, w2t, w1t, wot,
                                                                 B, S, p, (ff['seeds'][1], mh['seeds'][1]), sp)
            d_b2 = torch.empty((E,), device=dev, dtype=torch.float32)
            d_w2 = ops.linear_wgrad(dt, g2d, ff['h'], bias_out=d_b2, batch=rb, param=w2)
            d_b1 = torch.empty((gh.shape[1],), device=dev, dtype=torch.float32)
            d_w1 = ops.linear_wgrad(dt, gh, ff['x_in'], bias_out=d_b1, batch=rb, param=w1)
            d_bo = torch.empty((E,), device=dev, dtype=torch.float32)
            d_wo = ops.linear_wgrad(dt, g1d, mh['ctxv'], bias_out=d_bo, batch=rb, param=w_o)
            dgb2 = torch.empty((2 * E,), device=dev, dtype=torch.float32)
            rb.add_colsum(part2, part2.shape[0], 2 * E, dgb2)
            dqk = torch.empty((B * S, 2 * E), device=dev, dtype=gx2.dtype)
            dv = torch.empty((B * S, E), device=dev, dtype=gx2.dtype)
            ops.attention_bwd(dt, mh['q'], mh['k'], mh['v'], mh['ctxv'], gctx, mh['lse'], B, H, S, S, dqk[:, :E], dqk[:, E:], dv, mh['kpm'],
                              None, p, mh['seeds'][0], sp)
            d_win = ops._sink(w_in, (3 * E, E))
            if d_win is None:
                d_win = torch.empty((3 * E, E), device=dev, dtype=torch.float32)
            d_bin = torch.empty((3 * E,), device=dev, dtype=torch.float32)
            ops.linear_wgrad(dt, dqk, mh['q_in'], out=d_win[:2 * E], bias_out=d_bin[:2 * E], batch=rb)
            ops.linear_wgrad(dt, dv, mh['v_in'], out=d_win[2 * E:], bias_out=d_bin[2 * E:], batch=rb)
            gx, part1 = ops.encoder_qkv_bwd(dqk, dv, sv['x'], sv['m1'], sv['r1'], g1, gx1, wint, B, S)
            dgb1 = torch.empty((2 * E,), device=dev, dtype=torch.float32)
            rb.add_colsum(part1, part1.shape[0], 2 * E, dgb1)
            # the reduce launch that closes this layer touches the weights the NEXT layer's backward (the one below) streams first
            rb.flush(prefetch=sv.get('bwd_next'))
            ctx.sv = None
            return (gx, None, None, None, None, d_win, d_bin, d_wo, d_bo, d_w1, d_b1, d_w2, d_b2, dgb1[:E], dgb1[E:], dgb2[:E], dgb2[E:])
        if cfg['pre_norm']:
            g_x1n, d_w1, d_b1, d_w2, d_b2 = _ffn_bwd(dt, sv['ffn'], gx2, w1, w2, batch=rb)
            gx1, d_g2, d_be2, gx1d = ops.layernorm_bwd(dt, g_x1n, sv['x1'], g2, sv['m2'], sv['r2'], dres=gx2, batch=rb,
                                                       drop=_drop_args(sv['mha'], gx2.device))
            g_qk, _, g_v, d_win, d_bin, d_wo, d_bo = _mha_bwd(dt, sv['mha'], gx1, w_in, w_o, batch=rb, g_dropped=gx1d)
            # LN1 fed xn (to V) and xn+pos (to Q,K): both gradients land on xn
            gx, d_g1, d_be1 = ops.layernorm_bwd(dt, g_v, sv['x'], g1, sv['m1'], sv['r1'], dy2=g_qk, dres=gx1, batch=rb)
        else:
            dev = gx2.device
            g_t2, d_g2, d_be2, g_t2d = ops.layernorm_bwd(dt, gx2, sv['t2'], g2, sv['m2'], sv['r2'], batch=rb,
                                                         drop=_drop_args(sv['ffn'], dev))
            g_x1, d_w1, d_b1, d_w2, d_b2 = _ffn_bwd(dt, sv['ffn'], g_t2, w1, w2, res_for_gx=g_t2, batch=rb, g_dropped=g_t2d)
            g_t, d_g1, d_be1, g_td = ops.layernorm_bwd(dt, g_x1, sv['t'], g1, sv['m1'], sv['r1'], batch=rb,
                                                       drop=_drop_args(sv['mha'], dev))
            g_qk, _, g_v, d_win, d_bin, d_wo, d_bo = _mha_bwd(dt, sv['mha'], g_t, w_in, w_o, batch=rb, g_dropped=g_td)
            gx = ops.add(dt, ops.add(dt, g_qk, g_v), g_t)
        if share is not None:
            share.layer_done(last=cfg.get('layer_idx', 0) == 0)
        else:
            rb.flush()
        ctx.sv = None
        return (gx, None, None, None, None, d_win, d_bin, d_wo, d_bo, d_w1, d_b1, d_w2, d_b2, d_g1, d_be1, d_g2, d_be2)


# ======================================================================================= decoder layer
class DecoderLayerFn(Function):
    """reference sedt/transformer.py:215-297.  Inputs: tgt [B*Q,d], memory [B*S,d], memory+pos [B*S,d],
    query_pos [B*Q,d].  params: self_attn(in_w,in_b,out_w,out_b), multihead_attn(4), linear1(2), linear2(2),
    norm1(2), norm2(2), norm3(2)"""

    @staticmethod
    def forward(ctx, tgt, mem, mem_pos, qpos, kpm, tgt_mask, cfg, *P):
        dt = cfg['dt']
        (sw_in, sb_in, sw_o, sb_o, cw_in, cb_in, cw_o, cb_o, w1, b1, w2, b2, g1, be1, g2, be2, g3, be3) = P
        B, S, Q, H = cfg['B'], cfg['S'], cfg['Q'], cfg['H']
        p = cfg['dropout'] if cfg['training'] else 0.0
        seeds = [runtime.next_seed() for _ in range(6)]
        tgt, mem, mem_pos, qpos = _as(tgt, dt), _as(mem, dt), _as(mem_pos, dt), _as(qpos, dt)
        sv = {}
        tr = any(ctx.needs_input_grad)
        if cfg['pre_norm']:
            tn, tnp, m1, r1 = ops.layernorm_fwd(dt, tgt, g1, be1, add_t=qpos)
            t1, sv['sa'] = _mha_fwd(dt, tnp, tnp, tn, True, sw_in, sb_in, sw_o, sb_o, tgt, B, H, Q, Q, None, tgt_mask, p, seeds[0:2], tr)
            t1n, t1np, m2, r2 = ops.layernorm_fwd(dt, t1, g2, be2, add_t=qpos)
            t2, sv['ca'] = _mha_fwd(dt, t1np, mem_pos, mem, False, cw_in, cb_in, cw_o, cb_o, t1, B, H, Q, S, kpm, None, p, seeds[2:4], tr)
            t2n, _, m3, r3 = ops.layernorm_fwd(dt, t2, g3, be3)
            t3, sv['ffn'] = _ffn_fwd(dt, t2n, w1, b1, w2, b2, t2, p, seeds[4:6], tr, out=cfg.get('out'), act=cfg.get('act', 'relu'))
            sv.update(tgt=tgt, t1=t1, t2=t2, m1=m1, r1=r1, m2=m2, r2=r2, m3=m3, r3=r3)
        else:
            tp = ops.add(dt, tgt, qpos)
            a, sv['sa'] = _mha_fwd(dt, tp, tp, tgt, True, sw_in, sb_in, sw_o, sb_o, tgt, B, H, Q, Q, None, tgt_mask, p, seeds[0:2], tr)
            t1, _, m1, r1 = ops.layernorm_fwd(dt, a, g1, be1)
            t1p = ops.add(dt, t1, qpos)
            c, sv['ca'] = _mha_fwd(dt, t1p, mem_pos, mem, False, cw_in, cb_in, cw_o, cb_o, t1, B, H, Q, S, kpm, None, p, seeds[2:4], tr)
            t2, _, m2, r2 = ops.layernorm_fwd(dt, c, g2, be2)
            f, sv['ffn'] = _ffn_fwd(dt, t2, w1, b1, w2, b2, t2, p, seeds[4:6], tr, act=cfg.get('act', 'relu'))
            t3, _, m3, r3 = ops.layernorm_fwd(dt, f, g3, be3)
            sv.update(a=a, c=c, f=f, m1=m1, r1=r1, m2=m2, r2=r2, m3=m3, r3=r3)
        if tr and cfg.get('chain') is not None and cfg.get('layer_idx', 0) == 0:
            sv['bwd_next'] = cfg['chain'].top       # (the top encoder layer's backward operands: ops.BackwardChain)
        ctx.sv, ctx.cfg, ctx.P = sv, cfg, P
        return t3

    @staticmethod
    def backward(ctx, g3_):
        sv, cfg, P = ctx.sv, ctx.cfg, ctx.P
        dt = cfg['dt']
        (sw_in, sb_in, sw_o, sb_o, cw_in, cb_in, cw_o, cb_o, w1, b1, w2, b2, g1, be1, g2, be2, g3, be3) = P
        gt3 = _as(g3_, dt)
        share = cfg.get('wg_share')
        rb = share.batch() if share is not None else ops.ReduceBatch()
        if cfg['pre_norm']:
            g_t2n, d_w1, d_b1, d_w2, d_b2 = _ffn_bwd(dt, sv['ffn'], gt3, w1, w2, batch=rb)
            gt2, d_g3, d_be3, gt2d = ops.layernorm_bwd(dt, g_t2n, sv['t2'], g3, sv['m3'], sv['r3'], dres=gt3, batch=rb,
                                                       drop=_drop_args(sv['ca'], gt3.device))
            kvf = cfg.get('kv_fused', False)
            acc_m = cfg.get('acc_mem') if kvf else None
            acc_q = cfg.get('acc_qpos')
            g_q, g_k, g_v, d_cwin, d_cbin, d_cwo, d_cbo = _mha_bwd(dt, sv['ca'], gt2, cw_in, cw_o, batch=rb, g_dropped=gt2d, kv_fused=kvf,
                                                                   res_kv=None if acc_m is None else acc_m.take_pending())
            # LN2 outputs: t1n (unused on its own) and t1n + qpos (cross-attn query)
            gt1, d_g2, d_be2, gt1d = ops.layernorm_bwd(dt, g_q, sv['t1'], g2, sv['m2'], sv['r2'], dres=gt2, batch=rb,
                                                       drop=_drop_args(sv['sa'], gt3.device))
            g_qpos = g_q
            g_mem_pos, g_mem = g_k, g_v
            g_qk, _, g_vs, d_swin, d_sbin, d_swo, d_sbo = _mha_bwd(dt, sv['sa'], gt1, sw_in, sw_o, batch=rb, g_dropped=gt1d)
            # this layer's input is the previous layer's output, which also feeds the shared final LayerNorm: that share of its
            # gradient (StackViewFn) is added here, in the kernel that produces the gradient
            sg = cfg.get('share', {}).get('stack_g') if cfg.get('share') is not None else None
            extra = sg[cfg['layer_idx'] - 1] if (sg is not None and cfg.get('layer_idx', 0) > 0) else None
            gtgt, d_g1, d_be1 = ops.layernorm_bwd(dt, g_vs, sv['tgt'], g1, sv['m1'], sv['r1'], dy2=g_qk, dres=gt1, batch=rb, dres2=extra)
            if acc_m is not None and not acc_m.last():       # memory's gradient: handed on as the next layer's GEMM residual
                acc_m.pending, g_mem = g_mem, None
            if acc_q is not None:                            # qpos' gradient: two shares per layer, ONE sum at the end
                acc_q.parts += [g_q, g_qk]
                g_qpos = ops.add_n(dt, acc_q.take_parts()) if acc_q.last() else None
            else:
                g_qpos = ops.add(dt, g_qpos, g_qk)
        else:
            dev = gt3.device
            g_f, d_g3, d_be3, g_fd = ops.layernorm_bwd(dt, gt3, sv['f'], g3, sv['m3'], sv['r3'], batch=rb,
                                                       drop=_drop_args(sv['ffn'], dev))
            g_t2, d_w1, d_b1, d_w2, d_b2 = _ffn_bwd(dt, sv['ffn'], g_f, w1, w2, res_for_gx=g_f, batch=rb, g_dropped=g_fd)
            g_c, d_g2, d_be2, g_cd = ops.layernorm_bwd(dt, g_t2, sv['c'], g2, sv['m2'], sv['r2'], batch=rb,
                                                       drop=_drop_args(sv['ca'], dev))
            g_q, g_k, g_v, d_cwin, d_cbin, d_cwo, d_cbo = _mha_bwd(dt, sv['ca'], g_c, cw_in, cw_o, batch=rb, g_dropped=g_cd,
                                                                   kv_fused=cfg.get('kv_fused', False))
            g_t1 = ops.add(dt, g_q, g_c)                     # query path + residual
            g_qpos = g_q
            g_mem_pos, g_mem = g_k, g_v
            g_a, d_g1, d_be1, g_ad = ops.layernorm_bwd(dt, g_t1, sv['a'], g1, sv['m1'], sv['r1'], batch=rb,
                                                       drop=_drop_args(sv['sa'], dev))
            g_qk, _, g_vs, d_swin, d_sbin, d_swo, d_sbo = _mha_bwd(dt, sv['sa'], g_a, sw_in, sw_o, batch=rb, g_dropped=g_ad)
            gtgt = ops.add(dt, ops.add(dt, g_qk, g_vs), g_a)
            g_qpos = ops.add(dt, g_qpos, g_qk)
        # the first decoder layer's backward is the last thing before the encoder's: its reduce launch touches the weights the top encoder
        # layer's backward streams first (only LayerNorm launches follow it)
        if share is not None:
            share.layer_done(last=cfg.get('layer_idx', 0) == 0, prefetch=sv.get('bwd_next'))
        else:
            rb.flush(prefetch=sv.get('bwd_next'))
        ctx.sv = None
        return (gtgt, g_mem, g_mem_pos, g_qpos, None, None, None,
                d_swin, d_sbin, d_swo, d_sbo, d_cwin, d_cbin, d_cwo, d_cbo, d_w1, d_b1, d_w2, d_b2,
                d_g1, d_be1, d_g2, d_be2, d_g3, d_be3)


# ======================================================================================= backbone
def _bn(bn):
    """fold the four FrozenBatchNorm buffers (backbone.py:43-53) into per-channel scale / bias"""
    return ops.bn_fold(bn[0], bn[1], bn[2], bn[3])


class StemFn(Function):
    """conv0 (1->3, 1x1, bias) + conv1 (7x7 s2 p3) + FrozenBN + ReLU + maxpool 3x3 s2 p1 (backbone.py:102 + torchvision
    stem).  conv0 is folded into an effective 1-channel 7x7 conv with a border-aware bias (extra indicator columns in the
    im2col matrix), so the 3-channel intermediate never exists.  Only conv0 can be trainable (conv1 is frozen)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, bnw, bnb, bnrm, bnrv, dt):
        B, _, H, W = x.shape
        x = x.contiguous().float()
        wcat = ops.stem_prep(dt, w0, b0, w1)
        _, _, sc, bi = _prep_conv(dt, w1, (bnw, bnb, bnrm, bnrv))
        ctx.direct = ops.stem_pool_ok(dt, W)
        if ctx.direct:
            # ONE launch from the spectrogram to the pooled activation (csrc/stem.hip): neither the im2col matrix nor the
            # un-pooled activation exists; the backward needs x, the pooled output and the argmax bytes only
            pool, idx, Hp, Wp = ops.stem_pool_fwd(x, wcat, sc, bi, B, H, W, want_idx=any(ctx.needs_input_grad))
            ctx.dt, ctx.dims = dt, (B, H, W)
            if idx is not None:
                ctx.save_for_backward(x, pool, idx, sc, w1)
            ctx.out_hw = (Hp, Wp)
            return pool
        col, Ho, Wo = ops.stem_im2col(dt, x, B, H, W)
        s1 = ops.linear(dt, col, wcat, scale=sc, bias=bi, act=ACT_RELU)
        pool, idx, Hp, Wp = ops.maxpool_fwd(dt, s1, B, Ho, Wo, 64)
        ctx.dt, ctx.dims = dt, (B, Ho, Wo)
        ctx.save_for_backward(col, pool, idx, sc, w1)          # (the un-pooled activation s1 is not kept: see backward)
        ctx.mark_non_differentiable(idx)
        ctx.out_hw = (Hp, Wp)
        return pool

    @staticmethod
    def backward(ctx, g):
        col, pool, idx, sc, w1 = ctx.saved_tensors
        dt = ctx.dt
        if ctx.direct:
            B, H, W = ctx.dims
            G = ops.stem_pool_wgrad(col, _as(g, dt).contiguous(), idx, pool, sc, B, H, W)       # (col = x here)
            dw0, db0 = ops.stem_conv0_grad(G, w1)
            return None, dw0, db0, None, None, None, None, None, None
        B, Ho, Wo = ctx.dims
        # routed by argmax, masked by the stem ReLU: the selected element is the window maximum = the pooled value
        gs = ops.maxpool_bwd(dt, _as(g, dt), idx, None, B, Ho, Wo, 64, y=pool)
        G = ops.wgrad(dt, gs, col, gs.shape[0], ConvGeom(1, 1, 128, 64), rowscale=sc).view(64, 128)
        dw0, db0 = ops.stem_conv0_grad(G, w1)
        return None, dw0, db0, None, None, None, None, None, None


class BlockCfg(object):
    """one torchvision Bottleneck: inplanes -> planes (1x1) -> planes (3x3, stride, dilation) -> 4*planes (1x1)"""
    __slots__ = ('cin', 'planes', 'stride', 'dil', 'ds')

    def __init__(self, cin, planes, stride, dil, ds):
        self.cin, self.planes, self.stride, self.dil, self.ds = cin, planes, stride, dil, ds


class StageFn(Function):
    """one ResNet stage (layer1..layer4) of Bottleneck blocks on NHWC tokens.
    per-block tensors: conv1.w, bn1(4), conv2.w, bn2(4), conv3.w, bn3(4) [, ds.conv.w, ds.bn(4)].
    backward convention inside the stage: gradients handed from block to block are already masked by the ReLU of the
    tensor they belong to (the mask is fused into the producing dgrad epilogue)."""

    @staticmethod
    def forward(ctx, x, meta, *T):
        dt, B, H, W, blocks = meta['dt'], meta['B'], meta['H'], meta['W'], meta['blocks']
        x = _as(x, dt)
        saved = []
        i = 0
        # 1-bit ReLU masks: the backward needs only the SIGN of a block's (post-ReLU) output x - for the mask of the gradient that
        # flows into it - and reading the bf16 tensor for that is 16x the bytes (65 MB per layer1 block at B = 64).  Every block's
        # last convolution also writes the sign bits of its output; the next block's backward (or the next stage's, or
        # input_proj's: meta['holder']) reads those.
        tr = any(ctx.needs_input_grad) and ops.RELU_BITS
        xbits = meta.get('x_bits')
        for blk in blocks:
            n = 20 if blk.ds else 15
            t = T[i:i + n]
            i += n
            cin, pl = blk.cin, blk.planes
            w1f, w1b, s1, b1 = _prep_conv(dt, t[0], t[1:5])
            w2f, w2b, s2, b2 = _prep_conv(dt, t[5], t[6:10])
            w3f, w3b, s3, b3 = _prep_conv(dt, t[10], t[11:15])
            g1 = ConvGeom(H, W, cin, pl, 1)
            g2 = ConvGeom(H, W, pl, pl, 3, blk.stride, blk.dil, blk.dil)
            g3 = ConvGeom(g2.Ho, g2.Wo, pl, 4 * pl, 1)
            if ops.bneck_ok(dt, blk, W, B, H) and x.is_contiguous():
                # the whole block in one launch, its intermediates never in HBM (csrc/bneck.hip)
                cf = [packing.lookup_conv_frag(t[k]) for k in (0, 5, 10)]
                if all(c is not None for c in cf):
                    # a frozen block (layer1, backbone.py:60-62) keeps only the sign bits of its intermediates: the fused backward
                    # needs nothing else; a trainable one (layer2) keeps a and b as well, for its weight gradients
                    frozen = not (t[0].requires_grad or t[5].requires_grad or t[10].requires_grad)
                    # the fused input-gradient chain needs the sign bits of the block input whenever that gradient is masked; without
                    # them (RELU_BITS off) the backward is the per-op chain, which reads a and b: keep them then
                    first_ = blk is blocks[0]
                    mask_x_ = (not first_) or meta['mask_input']
                    if mask_x_ and xbits is None:
                        frozen = False
                    # (layer3: the launch also touches the NEXT block's operands, so that they are L2-resident when that block streams them)
                    nx = [packing.lookup_conv_frag(T[i + k]) for k in (0, 5, 10)] if (cin == 1024 and i + 15 <= len(T)) else None
                    nx = [c[0] for c in nx] if nx and all(c is not None for c in nx) else None
                    y, a, b, ybits, abits, bbits = ops.bneck_fwd(x, B, H, W, [c[0] for c in cf], ((s1, b1), (s2, b2), (s3, b3)),
                                                                 train=any(ctx.needs_input_grad), want_bits=bool(tr), want_ab=not frozen, nxt=nx)
                    saved.append(dict(blk=blk, x=x, a=a, b=b, g1=g1, g2=g2, g3=g3, s=(s1, s2, s3), wb=(w1b, w2b, w3b), H=H, W=W, y=y,
                                      xbits=xbits, fused=[c[1] for c in cf], ab_bits=(abits, bbits)))
                    x, xbits = y, ybits
                    continue
            if blk.ds and ops.bneck2_ok(dt, blk, W) and x.is_contiguous():
                # layer2's first block (stride 2), forward in one launch; the per-op backward below finds x, a, b, y as usual
                wdf, wdb, sd, bd = _prep_conv(dt, t[15], t[16:20])
                cf = [packing.lookup_conv_frag(t[k]) for k in (0, 5, 10, 15)]
                if all(c is not None for c in cf):
                    y, a, b, ybits = ops.bneck2_fwd(x, B, H, [c[0] for c in cf], ((s1, b1), (s2, b2), (s3, b3), (sd, bd)),
                                                    train=any(ctx.needs_input_grad), want_bits=bool(tr))
                    saved.append(dict(blk=blk, x=x, a=a, b=b, g1=g1, g2=g2, g3=g3, s=(s1, s2, s3), wb=(w1b, w2b, w3b), H=H, W=W, y=y,
                                      xbits=xbits, gd=ConvGeom(H, W, cin, 4 * pl, 1, blk.stride), sd=sd, wdb=wdb))
                    x, H, W, xbits = y, g2.Ho, g2.Wo, ybits
                    continue
            if blk.ds and ops.bneck0_ok(dt, blk, W) and x.is_contiguous():
                # layer1's first block, forward in one launch (the backward below is the per-op one: it finds x, a, b, y as usual)
                wdf, wdb, sd, bd = _prep_conv(dt, t[15], t[16:20])
                cf = [packing.lookup_conv_frag(t[k]) for k in (0, 5, 10, 15)]
                if all(c is not None for c in cf):
                    frozen = not any(t[k].requires_grad for k in (0, 5, 10, 15))
                    y, a, b, ybits, abits, bbits = ops.bneck0_fwd(x, B, H, [c[0] for c in cf], ((s1, b1), (s2, b2), (s3, b3), (sd, bd)),
                                                                  train=any(ctx.needs_input_grad), want_bits=bool(tr), want_ab=not frozen)
                    saved.append(dict(blk=blk, x=x, a=a, b=b, g1=g1, g2=g2, g3=g3, s=(s1, s2, s3), wb=(w1b, w2b, w3b), H=H, W=W, y=y,
                                      xbits=xbits, gd=ConvGeom(H, W, cin, 4 * pl, 1, blk.stride), sd=sd, wdb=wdb,
                                      chain=[c[1] for c in cf] if frozen else None, ab_bits=(abits, bbits)))
                    x, xbits = y, ybits
                    continue
            a = ops.conv_fwd(dt, x, B, g1, w1f, scale=s1, bias=b1, act=ACT_RELU)
            b = ops.conv_fwd(dt, a, B, g2, w2f, scale=s2, bias=b2, act=ACT_RELU)
            rec = dict(blk=blk, x=x, a=a, b=b, g1=g1, g2=g2, g3=g3, s=(s1, s2, s3), wb=(w1b, w2b, w3b), H=H, W=W)
            if blk.ds:
                wdf, wdb, sd, bd = _prep_conv(dt, t[15], t[16:20])
                gd = ConvGeom(H, W, cin, 4 * pl, 1, blk.stride)
                idn = ops.conv_fwd(dt, x, B, gd, wdf, scale=sd, bias=bd)
                rec.update(gd=gd, sd=sd, wdb=wdb)
            else:
                idn = x
            ybits = torch.empty((B * g3.Ho * g3.Wo, g3.Co // 8), device=x.device, dtype=torch.uint8) if (tr and g3.Co % 8 == 0) else None
            y = ops.conv_fwd(dt, b, B, g3, w3f, scale=s3, bias=b3, res=idn, ldr=idn.stride(0), act=ACT_RELU, act_post_res=1,
                             **({} if ybits is None else dict(bits_out=ybits)))
            rec['y'], rec['xbits'] = y, xbits
            saved.append(rec)
            x, H, W, xbits = y, g2.Ho, g2.Wo, ybits
        if meta.get('holder') is not None:
            meta['holder']['bits'] = xbits
        ctx.bwd_next = None
        chain = meta.get('chain')
        if tr and chain is not None:
            ctx.bwd_next = chain.top                    # (the stage below: its backward follows this stage's, ops.BackwardChain)
            if saved and saved[-1].get('fused') is not None:
                chain.top = tuple(saved[-1]['fused'][:3])      # what this stage's backward streams first: its last block's operands
        ctx.saved, ctx.meta, ctx.T = saved, meta, T
        ctx.out_hw = (H, W)
        return x

    @staticmethod
    def backward(ctx, gy):
        saved, meta, T = ctx.saved, ctx.meta, ctx.T
        dt, B = meta['dt'], meta['B']
        grads = [None] * len(T)
        # incoming gradient is w.r.t. the post-ReLU stage output: mask it (idempotent if the consumer already did)
        gp = _as(gy, dt) if meta.get('grad_premasked') else ops.relu_mask(dt, _as(gy, dt), saved[-1]['y'])
        rb = ops.ReduceBatch()
        i_end = len(T)
        need_x_grad = ctx.needs_input_grad[0]
        for bi_ in range(len(saved) - 1, -1, -1):
            r = saved[bi_]
            blk = r['blk']
            n = 20 if blk.ds else 15
            base = i_end - n
            i_end = base
            t = T[base:base + n]
            s1, s2, s3 = r['s']
            w1b, w2b, w3b = r['wb']
            first = bi_ == 0
            want_gx = (not first) or need_x_grad
            if not want_gx and not any(t[k].requires_grad for k in range(0, n, 5)):
                gp = None               # a frozen first block whose input needs no gradient: nothing to compute (and a fused forward
                continue                # kept no intermediates for the per-op code below)
            if r.get('chain') is not None and want_gx:
                # layer1's block 0, frozen: gy -> gb -> ga in one launch (sign bits of b, a from the fused forward), then the input-gradient
                # GEMMs of the projection and of conv1 (its residual operand); the stem's output needs no mask here (StemFn masks by its own
                # pooled output)
                _, _, ga = ops.bneck_bwd(gp, B, r['H'], r['W'], r['chain'], *r['ab_bits'], None, chain_only=True)
                side = ops.conv_dgrad(dt, gp, B, r['gd'], r['wdb'])
                mask_x = (not first) or meta['mask_input']
                ep = (dict(mask=r['x'], ldm=r['x'].stride(0)) if mask_x else {})
                gp = ops.conv_dgrad(dt, ga, B, r['g1'], w1b, res=side, ldr=side.stride(0), **ep)
                continue
            if r.get('fused') is not None and want_gx:
                mask_x = (not first) or meta['mask_input']
                if not mask_x or r.get('xbits') is not None:
                    # the input-gradient chain in one launch; a trainable block (layer2) also takes the two intermediate gradients out
                    # of it for its three weight-gradient GEMMs (a frozen one - layer1, backbone.py:60-62 - needs neither)
                    train_w = t[0].requires_grad or t[5].requires_grad or t[10].requires_grad
                    prev = saved[bi_ - 1] if bi_ > 0 else None          # (the block the backward visits next: its operands, layer3 only)
                    nx = prev['fused'] if (prev is not None and prev.get('fused') is not None and blk.cin == 1024) else None
                    gx, gb, ga = ops.bneck_bwd(gp, B, r['H'], r['W'], r['fused'], *r['ab_bits'], r['xbits'] if mask_x else None, want_g=train_w,
                                               nxt=nx)
                    if t[10].requires_grad:
                        grads[base + 10] = ops.wgrad(dt, gp, r['b'], B, r['g3'], rowscale=s3, batch=rb, param=t[10])
                    if t[5].requires_grad:
                        grads[base + 5] = ops.wgrad(dt, gb, r['a'], B, r['g2'], rowscale=s2, batch=rb, param=t[5])
                    if t[0].requires_grad:
                        grads[base + 0] = ops.wgrad(dt, ga, r['x'], B, r['g1'], rowscale=s1, batch=rb, param=t[0])
                    gp = gx
                    continue
            # conv3 (1x1): wgrad, dgrad masked by relu(b)
            if t[10].requires_grad:
                grads[base + 10] = ops.wgrad(dt, gp, r['b'], B, r['g3'], rowscale=s3, batch=rb, param=t[10])
            gb = ops.conv_dgrad(dt, gp, B, r['g3'], w3b, mask=r['b'], ldm=r['b'].stride(0))
            # conv2 (3x3): wgrad, dgrad masked by relu(a)
            if t[5].requires_grad:
                grads[base + 5] = ops.wgrad(dt, gb, r['a'], B, r['g2'], rowscale=s2, batch=rb, param=t[5])
            ga = ops.conv_dgrad(dt, gb, B, r['g2'], w2b, mask=r['a'], ldm=r['a'].stride(0))
            # conv1 (1x1) wgrad
            if t[0].requires_grad:
                grads[base + 0] = ops.wgrad(dt, ga, r['x'], B, r['g1'], rowscale=s1, batch=rb, param=t[0])
            if blk.ds and t[15].requires_grad:
                grads[base + 15] = ops.wgrad(dt, gp, r['x'], B, r['gd'], rowscale=r['sd'], batch=rb, param=t[15])
            if want_gx:
                side = ops.conv_dgrad(dt, gp, B, r['gd'], r['wdb']) if blk.ds else gp
                mask_x = (not first) or meta['mask_input']
                if mask_x and r.get('xbits') is not None:
                    ep = dict(mask=r['xbits'], ldm=r['xbits'].stride(0), mask_bits=True)
                else:
                    ep = dict(mask=r['x'], ldm=r['x'].stride(0)) if mask_x else {}
                gp = ops.conv_dgrad(dt, ga, B, r['g1'], w1b, res=side, ldr=side.stride(0), **ep)
            else:
                gp = None
        # the reduce launch that closes this stage touches the weights the stage BELOW streams first in its backward
        rb.flush(prefetch=ctx.bwd_next)
        ctx.saved = None
        return (gp, None) + tuple(grads)
