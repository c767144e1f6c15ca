"""SEDT model on the HIP path - counterpart of reference sedt/sedt.py:17-131, 398-409.

Same constructor, parameter names and output dict as the reference.  ``input_proj``, ``class_embed``,
``bbox_embed`` and ``weak_class_embed`` are nn.Conv2d / nn.Linear subclasses whose forward runs the MFMA GEMM
kernel (bias / ReLU / sigmoid fused in its epilogue)."""
import torch
from torch import nn

from .. import functional as Fn
from .. import runtime
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

    def forward(self, x):
        B, C, H, W = x.shape
        tok = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
        y = Fn.LinearFn.apply(tok, self.weight.view(self.out_channels, self.in_channels), self.bias, ACT_NONE, False,
                              runtime.compute_dtype())
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
        if pooling is not None:
            raise NotImplementedError('--pooling variants (sedt.py:47-61) are off in every supported config')
        self.num_queries = num_queries
        self.transformer = transformer
        hidden_dim = transformer.d_model
        self.class_embed = HipLinear(hidden_dim, num_classes + 1)
        self.bbox_embed = MLP(hidden_dim, hidden_dim, 2, 3)
        self.input_proj = HipConv1x1(backbone.num_channels, hidden_dim, kernel_size=1)
        self.backbone = backbone
        self.aux_loss = aux_loss
        self.dec_at = dec_at
        self.pooling = pooling
        if self.dec_at:
            self.query_embed = nn.Embedding(num_queries + 1, hidden_dim)
            self.weak_class_embed = HipLinear(hidden_dim, num_classes)
        else:
            self.query_embed = nn.Embedding(num_queries, hidden_dim)

    def forward(self, samples):
        """samples: NestedTensor | list of (1,T,F) tensors | (B,1,T,F) tensor.  Returns pred_logits (B,Q,C+1),
        pred_boxes (B,Q,2) = (centre, length) in [0,1], at (B,C) when dec_at, aux_outputs per decoder layer."""
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        assert mask is not None
        out = {}
        hs, memory = self.transformer(self.input_proj(src), mask, self.query_embed.weight, pos[-1], enc_at_embed=None)
        if self.dec_at:
            ev = hs[:, :, 1:, :]
            outputs_class = self.class_embed(ev, out_f32=True)
            outputs_coord = self.bbox_embed(ev, final_act=ACT_SIGMOID, out_f32=True)
            at = self.weak_class_embed(hs[-1, :, 0, :], act=ACT_SIGMOID, out_f32=True).squeeze()
            out['at'] = at
        else:
            outputs_class = self.class_embed(hs, out_f32=True)
            outputs_coord = self.bbox_embed(hs, final_act=ACT_SIGMOID, out_f32=True)
        out['pred_logits'] = outputs_class[-1]
        out['pred_boxes'] = outputs_coord[-1]
        if self.aux_loss:
            out['aux_outputs'] = self._set_aux_loss(outputs_class, outputs_coord)
        return out

    def _set_aux_loss(self, outputs_class, outputs_coord):
        return [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]


# ----------------------------------------------------------------------------------------------------------------
# host-side loss path (north star: Hungarian matching and SetCriterion stay on the host, in PyTorch)
# ----------------------------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from ..utilities import box_ops  # noqa: E402


class SetCriterion(nn.Module):
    """reference sedt.py:134-352 (fl=False, fine_tune=False).  Same loss names, weights and normalisation; all decoder
    layers are matched with one device->host copy and every loss is vectorised over the batch."""

    def __init__(self, num_classes, matcher, weight_dict, eos_coef, losses):
        super().__init__()
        self.num_classes, self.matcher, self.weight_dict = num_classes, matcher, weight_dict
        self.eos_coef, self.losses = eos_coef, losses
        empty_weight = torch.ones(self.num_classes + 1)
        empty_weight[-1] = self.eos_coef
        self.register_buffer('empty_weight', empty_weight)

    # ---- individual losses, given flat match index tensors on the device
    def _labels(self, logits, m, num_boxes, log):
        Bs, Q = logits.shape[:2]
        tc = torch.full((Bs, Q), self.num_classes, dtype=torch.int64, device=logits.device)
        cb = torch.ones((Bs, Q), dtype=torch.float32, device=logits.device)
        tc[m['b'], m['s']] = m['labels']
        cb[m['b'], m['s']] = m['coef']
        ce = F.cross_entropy(logits.transpose(1, 2), tc, self.empty_weight.to(logits.device), reduction='none')
        out = {'loss_ce': (ce * cb).sum() / num_boxes}
        if log:
            if m['labels'].numel() == 0:
                out['class_error'] = torch.zeros([], device=logits.device) + 100.0
            else:
                acc = (logits[m['b'], m['s']].argmax(-1) == m['labels']).float().mean() * 100
                out['class_error'] = 100 - acc
        return out

    def _boxes(self, boxes, m, num_boxes):
        src = boxes[m['b'], m['s']]
        tgt = m['boxes']
        s1, e1, s2, e2 = src[:, 0] - src[:, 1] / 2, src[:, 0] + src[:, 1] / 2, tgt[:, 0] - tgt[:, 1] / 2, tgt[:, 0] + tgt[:, 1] / 2
        l1 = (s1 - s2).abs() + (e1 - e2).abs()
        giou = 1 - box_ops.interval_giou_diag(src[:, 0], src[:, 1], tgt[:, 0], tgt[:, 1])
        return {'loss_bbox': (l1 * m['coef']).sum() / num_boxes, 'loss_giou': (giou * m['coef']).sum() / num_boxes}

    @torch.no_grad()
    def _cardinality(self, logits, tgt_lengths):
        card = (logits.argmax(-1) != logits.shape[-1] - 1).sum(1)
        return {'cardinality_error': F.l1_loss(card.float(), tgt_lengths.float())}

    def _weak(self, outputs, targets, strong_mask, weak_mask):
        if 'at' not in outputs:
            return {}
        lm = slice(weak_mask.stop) if weak_mask is not None else slice(strong_mask.stop)
        pred = outputs['at'][lm]
        gt = torch.zeros(pred.shape, dtype=torch.float32)
        for i in range(pred.shape[0]):
            lab = targets[i]["labels"].cpu()
            w = targets[i]['ratio'].detach().cpu().float() if 'ratio' in targets[i] else torch.ones(len(lab))
            gt[i].index_add_(0, lab, w)
        gt = gt.clamp(0, 1).to(pred.device)
        return {'loss_weak': F.binary_cross_entropy(pred, gt)}

    def _feature(self, outputs, m, n_clips, num_boxes):
        tf = outputs['gt_feature']
        tf = tf.view(n_clips, tf.shape[0] // n_clips, -1)[m['b'], m['t']]
        sf = outputs['pred_feature'][m['b'], m['s']]
        sf, tf = F.normalize(sf.float(), dim=1), F.normalize(tf.float(), dim=1)
        return {'loss_feature': F.mse_loss(sf, tf, reduction='none').sum() / num_boxes}

    @staticmethod
    def _flat_match(idx, coef, targets, device):
        b = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(idx)])
        s = torch.cat([s for s, _ in idx])
        t = torch.cat([t for _, t in idx])
        labels = torch.cat([tg["labels"].cpu()[J] for tg, (_, J) in zip(targets, idx)])
        boxes = torch.cat([tg["boxes"].cpu()[J].reshape(-1, 2) for tg, (_, J) in zip(targets, idx)]).float()
        pack = torch.cat([b[:, None].float(), s[:, None].float(), t[:, None].float(), labels[:, None].float(),
                          torch.cat(coef)[:, None], boxes], dim=1).to(device, non_blocking=True)   # one host->device copy
        return {'b': pack[:, 0].long(), 's': pack[:, 1].long(), 't': pack[:, 2].long(), 'labels': pack[:, 3].long(),
                'coef': pack[:, 4], 'boxes': pack[:, 5:7]}

    def forward(self, outputs, targets, weak_mask=None, strong_mask=None, fine_tune=False, normalize=False, fl=False):
        if fine_tune or fl:
            raise NotImplementedError('fine_tune / focal-loss branches are not built')
        if strong_mask is None:
            raise NotImplementedError('strong_mask=None (no strongly labelled clips) is not used by any driver')
        dev = outputs['pred_logits'].device
        layers = [outputs] + list(outputs.get('aux_outputs', []))
        logits = torch.stack([o['pred_logits'][strong_mask] for o in layers])
        boxes = torch.stack([o['pred_boxes'][strong_mask] for o in layers])
        st = targets[strong_mask]
        all_idx = self.matcher.match_layers(logits.detach(), boxes.detach(), st)
        coef0 = self.matcher.coefficients(all_idx[0], st, normalize)
        num_boxes = torch.cat(coef0).sum().clamp(min=0).to(dev) if len(coef0) else torch.zeros([], device=dev)
        tgt_lengths = torch.as_tensor([len(v["labels"]) for v in targets], device=dev)
        n_strong = logits.shape[1]
        losses = {}
        for li, o in enumerate(layers):
            coef = coef0 if li == 0 else self.matcher.coefficients(all_idx[li], st, False)
            m = self._flat_match(all_idx[li], coef, st, dev)
            d = {}
            for loss in self.losses:
                if loss == 'labels':
                    d.update(self._labels(o['pred_logits'][strong_mask], m, num_boxes, log=(li == 0)))
                elif loss == 'boxes':
                    d.update(self._boxes(o['pred_boxes'], m, num_boxes))
                elif loss == 'cardinality':
                    d.update(self._cardinality(o['pred_logits'], tgt_lengths))
                elif loss == 'weak' and li == 0:
                    d.update(self._weak(o, targets, strong_mask, weak_mask))
                elif loss == 'feature':
                    d.update(self._feature(o, m, n_strong, num_boxes))
            losses.update(d if li == 0 else {k + f'_{li - 1}': v for k, v in d.items()})
        return losses, all_idx[0]


class PostProcess(nn.Module):
    """reference sedt.py:355-396: logits/boxes -> per-clip scores, labels, (onset, offset) in seconds"""

    @torch.no_grad()
    def forward(self, outputs, target_sizes, audio_tags=None, at_m=2, is_semi=False, threshold=0.5):
        out_logits, out_bbox = outputs['pred_logits'], outputs['pred_boxes']
        bs, num_q, _ = out_logits.shape
        prob = F.softmax(out_logits.float(), -1)
        if audio_tags is not None:
            cls = prob[..., :-1]
            at = audio_tags.to(prob.device).float()
            if at_m in (2, 3):
                best_q = cls.argmax(1)                                          # (B, C): query with the max prob per class
                best = cls.gather(1, best_q[:, None, :])[:, 0, :]
                lift = best < threshold
                if at_m == 3:
                    lift = lift & at.bool()
                raised = torch.where(lift, torch.full_like(best, threshold), best)
                cls = cls.scatter(1, best_q[:, None, :], raised[:, None, :])
            if at_m in (1, 2):
                cls = cls * at[:, None, :]
            prob = torch.cat([cls, prob[..., -1:]], dim=-1)
        scores, labels = prob[..., :-1].max(-1)
        if not is_semi:
            boxes = box_ops.box_cxcywh_to_se(out_bbox.float()) * target_sizes.to(out_bbox.device).float().view(-1, 1, 1)
        else:
            boxes = out_bbox
        return [{'scores': s, 'labels': l, 'boxes': b} for s, l, b in zip(scores, labels, boxes)]
