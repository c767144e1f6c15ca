"""SEDT model on the HIP path - counterpart of reference sedt/sedt.py:17-131, 398-409.

Same constructor, parameter names and output dict as the reference.  ``input_proj``, ``class_embed``,
``bbox_embed`` and ``weak_class_embed`` are nn.Conv2d / nn.Linear subclasses whose forward runs the MFMA GEMM
kernel (bias / ReLU / sigmoid fused in its epilogue)."""
import torch
from torch import nn

from .. import functional as Fn
from .. import ops, packing, runtime
from ..ops import ACT_NONE, ACT_RELU, ACT_SIGMOID
from ..utilities.utils import NestedTensor, nested_tensor_from_tensor_list


class HipLinear(nn.Linear):
    """nn.Linear whose forward is the HIP GEMM; accepts (..., in_features); returns f32 (model outputs) or the compute dtype"""

    def forward(self, x, act=ACT_NONE, out_f32=False):
        lead = x.shape[:-1]
        y = Fn.LinearFn.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias, act, out_f32, runtime.compute_dtype())
        return y.view(*lead, self.out_features)


class HipConv1x1(nn.Conv2d):
    """1x1 nn.Conv2d (input_proj, sedt.py:36) as a GEMM over NHWC tokens; returns (B,Cout,H,W) with channels-last strides"""

    relu_input = False      # SEDT sets it: the input is the backbone's post-ReLU feature map (see ResNet50Body.forward)
    input_bits = None       # ... and this: a callable returning the 1-bit image of [input > 0] the backbone left (or None)

    def forward(self, x):
        B, C, H, W = x.shape
        tok = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
        bits = self.input_bits() if (self.relu_input and self.input_bits is not None) else None
        if bits is not None and tuple(bits.shape) != (B * H * W, C // 8):
            bits = None
        y = Fn.LinearFn.apply(tok, self.weight.view(self.out_channels, self.in_channels), self.bias, ACT_NONE, False,
                              runtime.compute_dtype(), self.relu_input, bits)
        return y.view(B, H, W, self.out_channels).permute(0, 3, 1, 2)


class MLP(nn.Module):
    """reference sedt.py:398-409; ReLU fused into the GEMM epilogues, optional fused final activation"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(HipLinear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x, final_act=ACT_NONE, out_f32=False):
        for i, layer in enumerate(self.layers):
            last = i == self.num_layers - 1
            x = layer(x, act=final_act if last else ACT_RELU, out_f32=out_f32 and last)
        return x


class SEDT(nn.Module):
    """reference sedt.py:17-131"""

    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, dec_at=False, pooling=None):
        super().__init__()
        self.pool_mode = None
        if pooling is not None:                  # sedt.py:47-61; --pooling takes exactly these four names (train_sedt.py:72)
            if pooling not in ('max', 'avg', 'attn', 'weighted_sum') or (pooling == 'weighted_sum' and not dec_at):
                raise ValueError(f"pooling={pooling!r}: the reference defines max / avg / attn, and weighted_sum for dec_at models only "
                                 "(sedt.py:47-61, 98-100: without dec_at it fails there with a missing pooling_func)")
            self.pool_mode = pooling
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = HipLinear(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 2, 3)
        self.input_proj = HipConv1x1(backbone.num_channels, hidden_dim, kernel_size=1)
        self.backbone = backbone
        body = getattr(backbone[0], 'body', None) if isinstance(backbone, nn.Sequential) else None
        if body is not None and hasattr(body, 'premasked_consumer'):
            # contract between the two modules: input_proj's dgrad epilogue applies the ReLU mask of layer4's output,
            # so layer4's backward skips its own masking pass
            self.input_proj.relu_input = True
            self.input_proj.input_bits = lambda: getattr(body, 'out_bits', None)
            body.premasked_consumer = True
        self.aux_loss = aux_loss
        self.dec_at = dec_at
        self.pooling = pooling
        if self.dec_at:
            self.query_embed = nn.Embedding(num_queries + 1, hidden_dim)
            self.weak_class_embed = HipLinear(hidden_dim, num_classes)
        else:
            self.query_embed = nn.Embedding(num_queries, hidden_dim)
        if self.pool_mode == 'attn':                             # sedt.py:52-53
            self.attn_dense_softmax = HipLinear(hidden_dim, num_classes)

    def pack_plan(self):
        """the model's weight-preparation plans (two launches prepare every weight / FrozenBN of a forward), cached per
        dtype/device; one plan per parameter-pointer set (packing.PlanSet), so student / EMA-teacher forwards - eager or
        captured in HIP graphs - never share job tables"""
        dt, dev = runtime.compute_dtype(), self.query_embed.weight.device
        key = (dt, str(dev))
        plans = self.__dict__.setdefault('_plans', {})
        if key not in plans:
            def factory():
                body = self.backbone[0].body
                convs = []
                l34 = set(id(p_) for layer in (body.layer3, body.layer4) for p_ in layer.parameters())
                bn_only = [(body.conv1.weight, body.bn1.tensors())]        # 7x7 stem: only its FrozenBN fold is needed
                for layer in (body.layer1, body.layer2, body.layer3, body.layer4):
                    for b in layer:
                        convs += [(b.conv1.weight, b.bn1.tensors()), (b.conv2.weight, b.bn2.tensors()), (b.conv3.weight, b.bn3.tensors())]
                        if b.downsample is not None:
                            convs.append((b.downsample[0].weight, b.downsample[1].tensors()))
                lin = [self.input_proj.weight]          # (Co, Ci, 1, 1): the Parameter itself, so pointer swaps are seen
                frags = []                              # weights the slab kernels stream fragment-major (csrc/slab.h)
                for l in self.transformer.encoder.layers:
                    lin += [l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.linear1.weight, l.linear2.weight]
                    frags += [l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.linear1.weight, l.linear2.weight]
                for l in self.transformer.decoder.layers:
                    lin += [l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.multihead_attn.in_proj_weight,
                            l.multihead_attn.out_proj.weight, l.linear1.weight, l.linear2.weight]
                    frags += [l.self_attn.in_proj_weight, l.self_attn.out_proj.weight, l.multihead_attn.in_proj_weight,
                              l.multihead_attn.out_proj.weight, l.linear1.weight, l.linear2.weight]
                lin += [self.class_embed.weight] + [m.weight for m in self.bbox_embed.layers]
                frags += [m.weight for m in self.bbox_embed.layers[:2]]
                for name in ('weak_class_embed', 'patch2query', 'attn_dense_softmax'):
                    if hasattr(self, name):
                        lin.append(getattr(self, name).weight)
                if hasattr(self, 'feature_align'):
                    lin += [m.weight for m in self.feature_align.layers]
                # identity Bottlenecks of layer1 / layer2 / layer3 run as one fused kernel each way (csrc/bneck.hip, bneck3.hip): their
                # packed operands fragment-major too
                cfr = [w for layer in (body.layer1, body.layer2, body.layer3) for b in layer if b.downsample is None
                       for w in (b.conv1.weight, b.conv2.weight, b.conv3.weight)]
                for b0 in (body.layer1[0], body.layer2[0]):     # the two projection blocks: fused forwards (bneck0 / bneck2_fwd_kernel)
                    cfr += [b0.conv1.weight, b0.conv2.weight, b0.conv3.weight, b0.downsample[0].weight]
                if ops.IGEMM_BREG:                              # developer A/B: B through registers in the LDS-DMA GEMMs of layer3 block 0 / layer4
                    have = set(id(w) for w in cfr)
                    cfr += [w for w, _ in convs if id(w) not in have and id(w) in l34]
                return packing.PackPlan(dt, dev, convs, lin, bn_only, frags, cfr)
            plans[key] = packing.PlanSet(factory)
        return plans[key]

    def forward(self, samples):
        """samples: NestedTensor | list of (1,T,F) tensors | (B,1,T,F) tensor.  Returns pred_logits (B,Q,C+1),
        pred_boxes (B,Q,2) = (centre, length) in [0,1], at (B,C) when dec_at, aux_outputs per decoder layer."""
        with self.pack_plan():
            return self._forward(samples)

    def _forward(self, samples):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        assert mask is not None
        out = {}
        hs, memory = self.transformer(self.input_proj(src), mask, self.query_embed.weight, pos[-1], enc_at_embed=None)
        # all heads as ONE autograd node over the stacked decoder output (functional.HeadsFn): they run over ALL query rows - with
        # dec_at query 0 is the audio-tag query - and the event outputs are views of rows q0..; the criterion kernels take the
        # full tensors plus the window (q0, Q), so no slice is ever materialised in either direction
        q0 = 1 if self.dec_at else 0
        b = self.bbox_embed.layers
        wa = (self.weak_class_embed.weight, self.weak_class_embed.bias) if self.dec_at else (None, None)
        res = Fn.HeadsFn.apply(hs, self.class_embed.weight, self.class_embed.bias, b[0].weight, b[0].bias, b[1].weight, b[1].bias,
                               b[2].weight, b[2].bias, wa[0], wa[1], runtime.compute_dtype())
        cls_full, box_full = res[0], res[1]
        outputs_class, outputs_coord = cls_full[:, :, q0:, :], box_full[:, :, q0:, :]
        if self.dec_at:
            out['at'] = res[2].squeeze()
        out['pred_logits'] = outputs_class[-1]
        out['pred_boxes'] = outputs_coord[-1]
        if self.pool_mode is not None:                           # sedt.py:96-106 / 112-119: one launch (csrc/pool_at.hip)
            mode, Q = self.pool_mode, cls_full.shape[2] - q0
            attn = self.attn_dense_softmax(hs[-1][:, q0:, :].contiguous(), out_f32=True) if mode == 'attn' else None
            at_p = Fn.PoolAtFn.apply(cls_full[-1], box_full[-1] if mode == 'weighted_sum' else None, attn, mode, q0, Q)
            out['at_p'] = at_p.squeeze() if mode in ('max', 'avg') else at_p
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(outputs_class, outputs_coord)
            out['_stacked'] = (cls_full, box_full)          # all decoder layers and all query rows, for the fused criterion kernels
            out['_q0'] = q0
        return out

    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]


# ----------------------------------------------------------------------------------------------------------------
# host-side loss path (north star: Hungarian matching and SetCriterion stay on the host, in PyTorch)
# ----------------------------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from ..utilities import box_ops  # noqa: E402


class TargetTables(object):
    """The targets of one batch as the flat device tables the matching kernel reads (include/sedt_hip.h, SedtMatch):
    labels / boxes / ratios concatenated over the clips plus int32 offset tables.  Buffers are allocated once with a
    fixed capacity (max_targets per clip), so a captured HIP graph can keep reading them while ``load`` refreshes their
    contents for every batch.  All tables are views of ONE device buffer: host-resident targets (what a loader yields) are
    laid out in a pinned staging buffer and travel in ONE asynchronous host->device copy per batch; device-resident targets
    are copied table by table."""

    def __init__(self, batch, ns, n_lab, device, max_targets=32, with_ratio=False, slots=4, dynamic_split=False,
                 weak_mask_none=False):
        """ns / n_lab: number of strongly labelled / labelled clips.  dynamic_split=True: they are only the FIRST batch's values -
        every ``load`` may bring another split (mix-up moves clips across the strong | weak boundary, utilities/mixup.py:13-127);
        the tables then have room for ``batch`` strong clips and the current split travels to the kernels as two device words."""
        if not 1 <= max_targets <= 63:
            raise ValueError('max_targets must be in 1..63 (one wave lane per target)')
        self.dynamic = bool(dynamic_split)
        # the reference's criterion called with weak_mask=None (a fully strong batch): loss_weak_p of --pooling models then runs
        # over every labelled clip instead of the weak ones (sedt.py:184)
        self.weak_mask_none = bool(weak_mask_none)
        self.B, self.max_targets, self.dev = batch, max_targets, device
        self.ns, self.n_lab = (batch, batch) if self.dynamic else (ns, max(n_lab, ns))          # capacities = strides
        self.cur_ns, self.cur_n_lab = ns, max(n_lab, ns)
        ns = self.ns
        n_off = batch + ns + 4                                                                   # lab_off | box_off | split
        n_lab_e, n_box_e = batch * max_targets, max(ns, 1) * max_targets
        o_lab = (4 * n_off + 7) // 8 * 8
        o_box = o_lab + 8 * n_lab_e
        o_rat = o_box + 8 * n_box_e
        total = o_rat + (4 * n_lab_e if with_ratio else 0)
        self._lay = (n_off, o_lab, n_lab_e, o_box, n_box_e, o_rat, total)
        self._blob = torch.zeros(total, dtype=torch.uint8, device=device)
        self.off = self._blob[:4 * n_off].view(torch.int32)
        self.lab_cat = self._blob[o_lab:o_box].view(torch.int64)
        self.box_cat = self._blob[o_box:o_rat].view(torch.float32).view(n_box_e, 2)
        self.ratio_cat = self._blob[o_rat:total].view(torch.float32) if with_ratio else None
        if with_ratio:
            self.ratio_cat.fill_(1.0)
        self.split = self.off[batch + ns + 2:] if self.dynamic else None
        self._pin = [torch.zeros(total, dtype=torch.uint8).pin_memory() for _ in range(slots)] if device.type == 'cuda' else None
        self._ev = [None] * slots
        self._slot = 0

    def as_dict(self):
        return {'lab_cat': self.lab_cat, 'box_cat': self.box_cat, 'ratio_cat': self.ratio_cat,
                'lab_off': self.off[:self.B + 1], 'box_off': self.off[self.B + 1:self.B + self.ns + 2]}

    @torch.no_grad()
    def load(self, targets, ns=None, n_lab=None):
        """ns / n_lab: this batch's split (dynamic_split tables only; default: the previous one)"""
        import numpy as np
        B = self.B
        if len(targets) != B:
            raise ValueError(f'expected {B} clips, got {len(targets)}')
        if ns is not None or n_lab is not None:
            if not self.dynamic:
                if (ns, n_lab) != (self.cur_ns, self.cur_n_lab):
                    raise ValueError('these tables were built for a fixed strong/weak split: TargetTables(dynamic_split=True)')
            else:
                ns = self.cur_ns if ns is None else ns
                self.cur_ns, self.cur_n_lab = ns, max(ns if n_lab is None else n_lab, ns)
                if not 0 <= self.cur_ns <= self.cur_n_lab <= B:
                    raise ValueError(f'split {self.cur_ns} | {self.cur_n_lab} outside 0..{B}')
        ns = self.cur_ns
        nlab = [int(t['labels'].shape[0]) for t in targets]
        nbox = [int(targets[b]['boxes'].reshape(-1, 2).shape[0]) for b in range(ns)]
        if max(nlab + nbox + [0]) > self.max_targets:
            raise ValueError(f'a clip has more than max_targets={self.max_targets} events')
        if any(nb > nl for nb, nl in zip(nbox, nlab)):
            raise ValueError('a strong clip has more boxes than labels')
        if self.ratio_cat is None and any('ratio' in t for t in targets):
            raise ValueError('targets carry pseudo-label ratios: build TargetTables(with_ratio=True)')
        off = [0]
        for n in nlab:
            off.append(off[-1] + n)
        off.append(0)
        for n in nbox:
            off.append(off[-1] + n)
        off += [off[-1]] * (self.ns - ns)                     # (dynamic split: clips beyond the strong part own no boxes)
        off += [self.cur_ns, self.cur_n_lab]
        nl, nb = sum(nlab), sum(nbox)
        n_off, o_lab, n_lab_e, o_box, n_box_e, o_rat, total = self._lay
        host = self._pin is not None and all(not t['labels'].is_cuda and not t['boxes'].is_cuda for t in targets)
        if self._pin is None or host:
            # ---- host-resident targets: lay the whole blob out on the host, ONE copy
            k = self._slot
            self._slot = (k + 1) % len(self._ev)
            if self._pin is not None:
                if self._ev[k] is not None:
                    self._ev[k].synchronize()                 # the copy that last used this pinned slot has long finished
                buf = self._pin[k].numpy()
            else:
                buf = np.zeros(total, np.uint8)
            buf[:4 * n_off].view(np.int32)[:len(off)] = off
            if nl:
                buf[o_lab:o_lab + 8 * nl].view(np.int64)[:] = torch.cat([t['labels'].reshape(-1) for t in targets]).numpy()
            if nb:
                buf[o_box:o_box + 8 * nb].view(np.float32)[:] = torch.cat([targets[b]['boxes'].reshape(-1, 2).float() for b in range(ns)]).numpy().reshape(-1)
            if self.ratio_cat is not None and nl:
                buf[o_rat:o_rat + 4 * nl].view(np.float32)[:] = np.concatenate(
                    [t['ratio'].detach().float().reshape(-1).cpu().numpy() if 'ratio' in t else np.ones(n, np.float32) for t, n in zip(targets, nlab)])
            if self._pin is not None:
                self._blob.copy_(self._pin[k], non_blocking=True)
                self._ev[k] = torch.cuda.Event()
                self._ev[k].record()
            else:
                self._blob.copy_(torch.from_numpy(buf))
            return self
        # ---- device-resident targets: table by table
        k = self._slot
        self._slot = (k + 1) % len(self._ev)
        if self._ev[k] is not None:
            self._ev[k].synchronize()
        pin_off = self._pin[k][:4 * n_off].view(torch.int32)
        pin_off[:len(off)].copy_(torch.tensor(off, dtype=torch.int32))
        self.off.copy_(pin_off, non_blocking=True)
        self._ev[k] = torch.cuda.Event()
        self._ev[k].record()
        if nl:
            self.lab_cat[:nl].copy_(torch.cat([t['labels'].reshape(-1) for t in targets]), non_blocking=True)
        if nb:
            self.box_cat[:nb].copy_(torch.cat([targets[b]['boxes'].reshape(-1, 2).float() for b in range(ns)]), non_blocking=True)
        if self.ratio_cat is not None and nl:
            src = targets[0]['labels'].device
            self.ratio_cat[:nl].copy_(torch.cat([t['ratio'].detach().float().reshape(-1).to(src) if 'ratio' in t else
                                                 torch.ones(n, device=src) for t, n in zip(targets, nlab)]), non_blocking=True)
        return self


ALPHA_FL, GAMMA_FL = 0.5, 1.0          # reference config.py:71-72 (focal-loss constants)


class _CriterionFn(torch.autograd.Function):
    """autograd node around ops.set_criterion / set_criterion_bwd: forward computes the loss vector, the weighted total as a
    scalar of its own and every per-term gradient; backward combines them with the gradients that reached the vector (single
    entries, e.g. a caller's own ``sum(loss_dict[k] * weight_dict[k])``) and / or the total."""

    @staticmethod
    def forward(ctx, logits_all, boxes_all, at, dense, empty_weight, layer_of, w_ce, w_bbox, w_giou, w_weak, fl, nonfinite, q0,
                at_p=None, w_weak_p=0.0):
        from .. import ops
        ctx.set_materialize_grads(False)           # usually only the total carries a gradient: no zero-filled g for the vector
        f32c = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()
        out, total, ctx.state = ops.set_criterion(f32c(logits_all), f32c(boxes_all), None if at is None else f32c(at), dense,
                                                  empty_weight, layer_of, w_ce, w_bbox, w_giou, w_weak, fl=fl, alpha_fl=ALPHA_FL,
                                                  gamma_fl=GAMMA_FL, nonfinite=nonfinite, q0=q0,
                                                  at_p=None if at_p is None else f32c(at_p), w_weak_p=w_weak_p,
                                                  wp_all=bool(dense.get('wp_all', False)))
        ctx.dts = (logits_all.dtype, boxes_all.dtype, None if at is None else at.dtype, None if at_p is None else at_p.dtype)
        return out, total

    @staticmethod
    def backward(ctx, g, gtotal):
        from .. import ops
        if g is None and gtotal is None:
            return (None,) * 15
        gl, gb, gat, gat_p = ops.set_criterion_bwd(ctx.state, g, gtotal)
        return (gl.to(ctx.dts[0]), gb.to(ctx.dts[1]), None if gat is None else gat.to(ctx.dts[2]),
                None, None, None, None, None, None, None, None, None, None,
                None if gat_p is None else gat_p.to(ctx.dts[3]), None)


class _FeatureLossFn(torch.autograd.Function):
    """SP-SEDT feature reconstruction (sedt.py:263-283) of all decoder layers: out[:L] = loss_feature per dense layer,
    out[L] = their weighted sum.  The forward launch also leaves the unweighted gradient; backward scales it per layer."""

    @staticmethod
    def forward(ctx, pred_all, gt, dense, layer_of, num_boxes, wvec, nonfinite=None, base_total=None):
        """base_total (optional device scalar with its own autograd history: SetCriterion's weighted total): the second result is then
        base_total + sum_d w[d] loss[d], added inside the launch (a torch add, and the zero-fill + copy of indexing the vector in the
        backward, are three launches otherwise)"""
        from .. import ops
        ctx.set_materialize_grads(False)
        res = ops.feature_loss(pred_all.detach().float().contiguous(), gt.detach().float().contiguous(), dense,
                               layer_of, num_boxes, wvec, nonfinite, None if base_total is None else base_total.detach().float())
        out, ctx.dpred = res[0], res[1]
        ctx.wvec, ctx.dt = wvec, pred_all.dtype
        inv = [0] * len(layer_of)
        for d, ml in enumerate(layer_of):
            inv[ml] = d
        ctx.inv = inv                                  # dpred is in the model's layer order, the loss vector in dense order
        ctx.has_base = base_total is not None
        return (out, res[2]) if ctx.has_base else out

    @staticmethod
    def backward(ctx, g, gtotal=None):
        from .. import ops
        if g is None and gtotal is None:
            return (None,) * 8
        L = len(ctx.inv)
        gl = gt_ = None
        if g is not None:
            g = g.contiguous().float()
            gl, gt_ = g[:L], g[L:]
        if gtotal is not None:
            gtotal = gtotal.reshape(1).float()
            gt_ = gtotal if gt_ is None else gt_ + gtotal
        d = ops.scale_layers(ctx.dpred, gl, gt_, ctx.wvec, ctx.inv)
        ctx.dpred = None
        return d.to(ctx.dt), None, None, None, None, None, None, (gtotal.reshape(()) if (ctx.has_base and gtotal is not None) else None)


class SetCriterion(nn.Module):
    """reference sedt.py:134-352: same loss names, weights, normalisation, fine_tune / normalize / fl switches.

    ``forward`` keeps the reference's host-side API (list-of-dict targets; north star: matching on the host) but is
    organised for the GPU it feeds: ``prepare`` builds the matching costs of ALL decoder layers on the device, brings them
    to the host in ONE copy, solves every assignment in one C++ call (sedt_hungarian_batch) and uploads dense, fixed-shape
    target tensors in ONE copy; ``compute`` is then one fused launch for every loss and gradient.  ``prepare_device`` is
    the same preparation without leaving the device (sedt_match_targets), which lets a whole step live in one HIP graph."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer('empty_weight', empty_weight)
        self.last_total = None
        self._wvec = {}
        self.nonfinite = None                     # optional int32 device word (engine: polled instead of a sync per step)
        self.host_compute = None                  # test hook: a callable (criterion, outputs, dense, fl) for CPU tensors (tests/host_criterion.py)

    # ------------------------------------------------------------------ host part
    @torch.no_grad()
    def prepare(self, outputs, targets, weak_mask=None, strong_mask=None, normalize=False, fine_tune=False, fl=False,
                ft_rand=None):
        """ft_rand: optional list (one entry per strong clip) of the uniforms the fine-tune branch consumes, in the order the
        reference draws them (matcher.py:116); default: torch.rand."""
        import numpy as np
        from .. import lib as L_
        if strong_mask is None or strong_mask.start not in (None, 0) or strong_mask.step not in (None, 1):
            raise NotImplementedError('strong_mask must be slice(0, n): strongly labelled clips come first in every driver')
        layers = [outputs] + list(outputs.get('aux_outputs', []))
        L = len(layers)
        dev = outputs['pred_logits'].device
        ns = len(targets[strong_mask])
        st = targets[:ns]
        logits = torch.stack([o['pred_logits'][:ns] for o in layers]).detach().float()
        boxes = torch.stack([o['pred_boxes'][:ns] for o in layers]).detach().float()
        Q = logits.shape[2]
        sizes = [int(len(t['boxes'])) for t in st]
        Nt = sum(sizes)
        n_lab = (weak_mask.stop if weak_mask is not None else strong_mask.stop) if 'at' in outputs else ns
        n_lab = max(n_lab, ns)
        lab_sizes = [int(len(targets[i]['labels'])) for i in range(n_lab)]
        has_ratio = any('ratio' in t for t in targets[:n_lab])
        if fine_tune and has_ratio and not normalize:
            raise ValueError('fine_tune with mixup ratios is undefined in the reference (matcher.py:130: shapes mismatch)')
        if fine_tune and (Nt == 0 or min(sizes) == 0):
            raise ValueError('fine_tune needs at least one event in every strong clip (matcher.py:103 takes a min over them)')
        parts = []
        if Nt > 0:
            tgt_ids = torch.cat([t['labels'][:len(t['boxes'])] for t in st]).to(dev)
            tgt_bbox = torch.cat([t['boxes'].reshape(-1, 2) for t in st]).to(dev).float()
            cost, loc = self.matcher.cost_matrices(logits, boxes, tgt_ids, tgt_bbox, fl=fl, with_loc=fine_tune)
            parts += [cost.flatten(), tgt_bbox.flatten()]
            if fine_tune:
                parts.append(loc[0].flatten())
        lab_all = torch.cat([targets[i]['labels'] for i in range(n_lab)]).to(dev).float() if sum(lab_sizes) else None
        if lab_all is not None:
            parts.append(lab_all)
        if has_ratio:
            parts.append(torch.cat([targets[i]['ratio'].detach().float().to(dev) if 'ratio' in targets[i] else
                                    torch.ones(lab_sizes[i], device=dev) for i in range(n_lab)]))
        host = torch.cat(parts).cpu().numpy() if parts else np.zeros(0, np.float32)   # the ONE device->host copy
        o = 0
        assign = -np.ones((L, ns, Q), np.int32)
        tb = np.zeros((Nt, 2), np.float32)
        off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32) if ns else np.zeros(0, np.int32)
        extra0 = np.zeros((ns, Q), bool)              # fine_tune: queries added beyond the Hungarian pairs (final layer)
        if Nt > 0:
            cost_h = np.ascontiguousarray(host[o:o + L * ns * Q * Nt]); o += L * ns * Q * Nt
            tb = host[o:o + 2 * Nt].reshape(Nt, 2); o += 2 * Nt
            nc = np.asarray(sizes, np.int32)
            L_.check(L_.load().sedt_hungarian_batch(cost_h.ctypes.data, L, ns, Q, Nt, off.ctypes.data, nc.ctypes.data,
                                                    assign.ctypes.data), 'hungarian_batch')
            if fine_tune:                             # matcher.py:99-121 on the final layer's assignment
                loc_h = host[o:o + ns * Q * Nt].reshape(ns, Q, Nt); o += ns * Q * Nt
                m = self.matcher
                for b, n in enumerate(sizes):
                    cl = loc_h[b][:, off[b]:off[b] + n]
                    near_t, near_c = cl.argmin(1), cl.min(1)
                    hung = assign[0, b] >= 0
                    close = near_c < np.float32(m.epsilon)
                    extra = np.nonzero(close & ~hung)[0]
                    u = (np.asarray(ft_rand[b], np.float32)[:len(extra)] if ft_rand is not None
                         else torch.rand(len(extra)).numpy())
                    add = extra[~(u > np.float32(m.alpha * int(hung.sum()) / Q))]
                    new = np.where(hung & close, assign[0, b], -1)
                    new[add] = near_t[add]
                    assign[0, b] = new
                    extra0[b, add] = True
        nl = sum(lab_sizes)
        lab_h = host[o:o + nl].astype(np.int64); o += nl
        ratio_h = host[o:o + nl] if has_ratio else np.ones(nl, np.float32)
        lab_off = np.concatenate([[0], np.cumsum(lab_sizes)]).astype(np.int64)
        # per-clip padded target tables (strong clips: the first len(boxes) labels belong to the boxes)
        nmax = max(max(sizes) if sizes else 0, 1)
        lab_pad = np.full((ns, nmax), self.num_classes, np.int64)
        box_pad = np.full((ns, nmax, 2), 0.5, np.float32)
        rat_pos = np.ones((ns, max(Q, 1)), np.float32)       # ratio[k] for the k-th matched query (POSITIONAL, matcher.py:130)
        bo = 0
        for b, n in enumerate(sizes):
            lab_pad[b, :n] = lab_h[lab_off[b]:lab_off[b] + n]
            box_pad[b, :n] = tb[bo:bo + n]
            r = ratio_h[lab_off[b]:lab_off[b + 1]][:Q]
            rat_pos[b, :len(r)] = r
            bo += n
        matched = assign >= 0
        a = np.clip(assign, 0, None)
        bi = np.arange(ns)[None, :, None]
        kth = np.clip(np.cumsum(matched, axis=2) - 1, 0, None)
        cf = rat_pos[bi, kth] if has_ratio else np.ones((L, ns, Q), np.float32)
        if normalize:                                         # final layer only (sedt.py:320 vs :340): 1 / #queries per target
            cnt = (a[0][:, :, None] == a[0][:, None, :]) & matched[0][:, :, None] & matched[0][:, None, :]
            cf = cf.copy()
            cf[0] = 1.0 / np.maximum(cnt.sum(2), 1)
        tc = np.where(matched, lab_pad[bi, a], self.num_classes).astype(np.float32)
        coef = np.where(matched, cf, 1.0).astype(np.float32)          # CE weight of every query
        wbox = np.where(matched, cf, 0.0).astype(np.float32)          # box-loss weight (0 = unmatched)
        tbox = np.where(matched[..., None], box_pad[bi, a], 0.5).astype(np.float32)
        num_boxes = float(wbox[0].sum())
        C = self.num_classes
        gt_weak = np.zeros((n_lab, C), np.float32)
        if nl and 'at' in outputs:
            clip_of = np.repeat(np.arange(n_lab), lab_sizes)
            np.add.at(gt_weak, (clip_of, lab_h), ratio_h)
            gt_weak = np.clip(gt_weak, 0, 1)
        tgt_len = np.asarray([len(t['labels']) for t in targets], np.float32)
        pack = np.concatenate([tc.ravel(), coef.ravel(), wbox.ravel(), tbox.ravel(), a.astype(np.float32).ravel(),
                               gt_weak.ravel(), tgt_len, np.asarray([num_boxes], np.float32)])
        d = torch.from_numpy(pack).to(dev, non_blocking=True)                       # the ONE host->device copy
        dense = self.dense_views(d, (L, ns, Q, n_lab, C, len(targets)))
        dense['wp_all'] = weak_mask is None           # loss_weak_p indexes with weak_mask: None = every labelled clip (sedt.py:184)
        idx0 = []
        for b in range(ns):                           # reference order: surviving Hungarian pairs, then the added queries
            first = np.nonzero(matched[0, b] & ~extra0[b])[0]
            second = np.nonzero(extra0[b])[0]
            qs = np.concatenate([first, second]).astype(np.int64)
            idx0.append((torch.from_numpy(qs), torch.from_numpy(assign[0, b][qs].astype(np.int64))))
        return dense, idx0

    @staticmethod
    def dense_views(d, meta):
        """views into the packed dense-target buffer (layout fixed by the batch composition, so a captured graph can keep
        one static buffer and only its contents change).  All entries are f32 VIEWS: integer conversions happen inside
        ``compute`` so that they are part of a captured graph."""
        L, ns, Q, n_lab, C, nt = meta
        n3 = L * ns * Q
        return {'tc': d[0:n3].view(L, ns, Q), 'coef': d[n3:2 * n3].view(L, ns, Q),
                'wbox': d[2 * n3:3 * n3].view(L, ns, Q), 'tbox': d[3 * n3:5 * n3].view(L, ns, Q, 2),
                'tidx': d[5 * n3:6 * n3].view(L, ns, Q),
                'gt_weak': d[6 * n3:6 * n3 + n_lab * C].view(n_lab, C),
                'tgt_len': d[6 * n3 + n_lab * C:6 * n3 + n_lab * C + nt],
                'num_boxes': d[-1:], 'ns': ns, 'n_lab': n_lab, 'L': L, '_pack': d, '_meta': meta}

    # ------------------------------------------------------------------ device matching (no host round trip)
    def prepare_device(self, outputs, tables, pack=None, assign=None, normalize=False, fine_tune=False, fl=False, ft_rand=None):
        """same result as ``prepare`` (dense targets of all decoder layers), but the assignment problems are solved ON the
        device (ops.match_targets: one wave per problem) from the flat target tables of ``TargetTables`` - no device->host
        copy, no host work that depends on model outputs, so the whole train step can be captured in ONE HIP graph.
        ft_rand: optional f32 [ns, Q] device tensor of injected uniforms (default: a counter hash, fresh per replay)."""
        from .. import ops
        if '_stacked' not in outputs:
            raise RuntimeError('prepare_device needs the stacked head outputs (model built with aux_loss=True)')
        logits_all, boxes_all = outputs['_stacked']
        q0 = outputs.get('_q0', 0)
        L, B, Qs, C1 = logits_all.shape
        Q = Qs - q0
        meta = (L, tables.ns, Q, tables.n_lab if 'at' in outputs else tables.ns, C1 - 1, B)
        if pack is None:                                  # (every entry is written by the matching kernel)
            pack = torch.empty(self.dense_numel(meta), device=logits_all.device, dtype=torch.float32)
        dense = self.dense_views(pack, meta)
        dense['split'] = getattr(tables, 'split', None)      # {ns, n_lab} as device words: the split is data, not graph structure
        dense['wp_all'] = bool(getattr(tables, 'weak_mask_none', False))
        m = self.matcher
        seed_ptr = runtime.seed_ptr(logits_all.device) if (fine_tune and ft_rand is None) else None
        f32c = lambda t: t.detach() if (t.dtype == torch.float32 and t.is_contiguous()) else t.detach().float().contiguous()
        ops.match_targets(f32c(logits_all), f32c(boxes_all), tables.as_dict(),
                          dense, [L - 1] + list(range(L - 1)), float(m.cost_class), float(m.cost_bbox), float(m.cost_giou),
                          tables.max_targets, assign=assign, fl=fl, fine_tune=fine_tune, normalize=normalize,
                          epsilon=float(m.epsilon), alpha=float(m.alpha), alpha_fl=ALPHA_FL, gamma_fl=GAMMA_FL, ft_rand=ft_rand,
                          ft_seed=runtime.next_seed() if (fine_tune and ft_rand is None) else 0, seed_ptr=seed_ptr, q0=q0)
        if 'feature' in self.losses:                      # sum of the final layer's coefficients (the criterion kernel sums it itself)
            dense['num_boxes'] = ops.sum_f32(dense['wbox'][0], out=dense['num_boxes'])
        else:
            dense['num_boxes'] = None
        return dense

    @staticmethod
    def dense_numel(meta):
        L, ns, Q, n_lab, C, nt = meta
        return 6 * L * ns * Q + n_lab * C + nt + 1

    # ------------------------------------------------------------------ device part (fixed shapes)
    def _weights(self, name, L):
        return [float(self.weight_dict.get(name if d == 0 else f'{name}_{d - 1}', 0.0)) for d in range(L)]

    def _dev_const(self, key, values, device):
        """small f32 constant on the device, uploaded once and then reused (graph-capture safe); values: list or callable"""
        k = (key, str(device))
        if k not in self._wvec:
            v = values() if callable(values) else values
            self._wvec[k] = torch.as_tensor(v, dtype=torch.float32).to(device)
        return self._wvec[k]

    def _compute_fused(self, outputs, dense, fl=False):
        """all losses + their gradients in ONE launch (csrc/criterion.hip) - plus one for the SP-SEDT feature loss; the dict
        returned has the same keys / values as the reference's."""
        q0 = 0
        if '_stacked' in outputs:
            logits_all, boxes_all = outputs['_stacked']
            q0 = outputs.get('_q0', 0)
        else:                                          # a hand-made output dict: stack [final, aux_0, ...] -> model order
            layers = list(outputs.get('aux_outputs', [])) + [outputs]
            logits_all = torch.stack([o['pred_logits'] for o in layers])
            boxes_all = torch.stack([o['pred_boxes'] for o in layers])
        L = dense['L']
        layer_of = [L - 1] + list(range(L - 1))
        at = outputs['at'] if ('weak' in self.losses and 'at' in outputs) else None
        if at is not None and at.dim() == 1:
            at = at[None]
        at_p = self._pooled(outputs, at)
        zero = [0.0] * L
        dev = logits_all.device
        ew = self._dev_const('ew', lambda: self.empty_weight.detach().float().cpu(), dev)
        vec, total = _CriterionFn.apply(
            logits_all, boxes_all, at, dense, ew, layer_of,
            self._weights('loss_ce', L) if 'labels' in self.losses else zero,
            self._weights('loss_bbox', L) if 'boxes' in self.losses else zero,
            self._weights('loss_giou', L) if 'boxes' in self.losses else zero,
            float(self.weight_dict.get('loss_weak', 0.0)) if at is not None else 0.0, fl, self.nonfinite, q0,
            at_p, float(self.weight_dict.get('loss_weak_p', 0.0)) if at_p is not None else 0.0)
        out = {}
        names = []
        if 'labels' in self.losses:
            names.append((0, 'loss_ce'))
            out['class_error'] = vec[4 * L + 4].detach()
        if 'boxes' in self.losses:
            names += [(1, 'loss_bbox'), (2, 'loss_giou')]
        if 'cardinality' in self.losses:
            names.append((3, 'cardinality_error'))
        for slot, k in names:
            for d in range(L):
                v = vec[4 * d + slot]
                out[k if d == 0 else f'{k}_{d - 1}'] = v.detach() if slot == 3 else v
        if at is not None:
            out['loss_weak'] = vec[4 * L + 2]
        if at_p is not None:
            out['loss_weak_p'] = vec[4 * L + 5]
        if 'feature' in self.losses:
            if '_stacked_feature' in outputs:
                feats = outputs['_stacked_feature']
            else:
                feats = torch.stack([o['pred_feature'] for o in list(outputs.get('aux_outputs', [])) + [outputs]])
            wv = self._dev_const(('wfeat', L), self._weights('loss_feature', L), dev)
            nb = dense['num_boxes'] if dense.get('num_boxes') is not None else None
            if nb is None:
                from .. import ops
                nb = ops.sum_f32(dense['wbox'][0])
            fv, total = _FeatureLossFn.apply(feats, outputs['gt_feature'], dense, layer_of, nb, wv, self.nonfinite, total)
            for d in range(L):
                out['loss_feature' if d == 0 else f'loss_feature_{d - 1}'] = fv[d]
        self.last_total = total
        return out

    def _pooled(self, outputs, at):
        """outputs['at_p'] [B,C] when loss_weak_p applies (sedt.py:182-185: inside loss_weak, i.e. only with 'weak' among the losses)"""
        if 'weak' not in self.losses or 'at_p' not in outputs:
            return None
        if at is None:
            raise ValueError("loss_weak_p needs outputs['at']: the reference builds its targets from it (sedt.py:164-175)")
        return outputs['at_p'].reshape(-1, self.num_classes)

    def compute(self, outputs, dense, fl=False):
        if outputs['pred_logits'].is_cuda:
            L, B = dense['L'], outputs['pred_logits'].shape[0]
            if L > 8 or L * B > 8192:
                raise NotImplementedError(f'fused criterion handles up to 8 decoder layers and L*B <= 8192 (got {L}, {B})')
            return self._compute_fused(outputs, dense, fl)
        if self.host_compute is None:
            raise RuntimeError('SetCriterion computes its losses on the MI355X HIP path only (sedt_set_criterion): the outputs are not GPU '
                               'tensors.  (The CPU tests of the matching / target logic install tests/host_criterion.py through '
                               '`criterion.host_compute`; the product has no CPU implementation.)')
        return self.host_compute(self, outputs, dense, fl)

    def forward(self, outputs, targets, weak_mask=None, strong_mask=None, fine_tune=False, normalize=False, fl=False,
                ft_rand=None):
        dense, idx0 = self.prepare(outputs, targets, weak_mask, strong_mask, normalize, fine_tune, fl, ft_rand)
        return self.compute(outputs, dense, fl), idx0


class PostProcess(nn.Module):
    """reference sedt.py:355-396: logits/boxes -> per-clip scores, labels, (onset, offset) in seconds - one launch
    (sedt_postprocess) for the whole batch and every fusion mode; the list of per-clip dicts the reference returns is a
    list of views into the batched results."""

    @torch.no_grad()
    def batched(self, outputs, target_sizes, audio_tags=None, at_m=2, is_semi=False, threshold=0.5):
        from .. import ops
        return ops.postprocess(outputs['pred_logits'], outputs['pred_boxes'], None if is_semi else target_sizes, audio_tags, at_m,
                               is_semi, threshold)

    @torch.no_grad()
    def forward(self, outputs, target_sizes, audio_tags=None, at_m=2, is_semi=False, threshold=0.5):
        scores, labels, boxes = self.batched(outputs, target_sizes, audio_tags, at_m, is_semi, threshold)
        return [{'scores': s, 'labels': l, 'boxes': b} for s, l, b in zip(scores, labels, boxes)]
