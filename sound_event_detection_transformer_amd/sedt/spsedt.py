"""SP-SEDT pre-training model on the HIP path - counterpart of reference sedt/spsedt.py."""
import torch
from torch import nn

from .. import functional as Fn
from .. import ops, runtime
from ..ops import ACT_SIGMOID
from ..utilities.utils import NestedTensor
from .sedt import SEDT, MLP, HipLinear


class SPSEDT(SEDT):
    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, dec_at=False, feature_recon=True,
                 query_shuffle=False, mask_ratio=0.1, num_patches=10, pooling=None):
        super().__init__(backbone, transformer, num_classes, num_queries, aux_loss, dec_at, pooling)
        hidden_dim = transformer.d_model
        self.patch2query = HipLinear(backbone.num_channels, hidden_dim)
        self.num_patches = num_patches
        self.mask_ratio = mask_ratio
        self.feature_recon = feature_recon
        if self.feature_recon:
            self.feature_align = MLP(hidden_dim, hidden_dim, backbone.num_channels, 2)
        self.query_shuffle = query_shuffle
        assert num_queries % num_patches == 0
        qpp = num_queries // num_patches
        am = torch.ones(self.num_queries, self.num_queries) * float('-inf')
        for i in range(num_patches):
            am[i * qpp:(i + 1) * qpp, i * qpp:(i + 1) * qpp] = 0
        self.register_buffer('attention_mask', am, persistent=False)   # reference keeps a plain attribute (spsedt.py:29-32)

    def forward(self, samples, patches: torch.Tensor, query_mask=None):
        """samples = (tensors (B,1,T,F), mask (B,T,F)); patches (B,P,1,h,w).  ``query_mask`` (Q,B,1) optionally injects
        the Bernoulli(1-mask_ratio) query-patch mask that spsedt.py:65 draws with torch.rand."""
        with self.pack_plan():
            return self._forward_sp(samples, patches, query_mask)

    def _forward_sp(self, samples, patches, query_mask=None):
        dev = self.query_embed.weight.device
        if any(p.requires_grad for p in self.backbone.parameters()):
            # gt_feature is computed without autograd (the reference's recipe freezes the backbone: train_spsedt.py:50)
            raise NotImplementedError('SP-SEDT on the HIP path needs a frozen backbone (lr_backbone = 0)')
        bnp = patches.shape[1]
        samples = NestedTensor(samples[0].to(dev), samples[1].to(dev))
        patches = patches.to(dev)
        feature, pos = self.backbone(samples)
        src, mask = feature[-1].decompose()
        bs = patches.shape[0]
        pf = self.backbone(patches.flatten(0, 1))[-1]                        # (B*P, 2048, h', w') NHWC memory
        BP, C, ph, pw = pf.shape
        gt = ops.avgpool(runtime.compute_dtype(), pf.permute(0, 2, 3, 1).reshape(BP * ph * pw, C), BP, ph * pw, C)
        pq = self.patch2query(gt)                                            # [B*P, d], compute dtype
        start = 1 if self.dec_at else 0
        qpp = self.num_queries // self.num_patches
        d = pq.shape[-1]
        dt = runtime.compute_dtype()
        if self.training:
            if bnp != self.num_patches:
                raise ValueError(f'training uses a fixed number of query patches ({self.num_patches}, spsedt.py:63), got {bnp}')
            qe = self.query_embed.weight[start:, :] if start else self.query_embed.weight      # (no slice node when there is nothing to cut)
            if self.query_shuffle:                                          # spsedt.py:60: the queries are permuted
                qe = qe[torch.randperm(self.num_queries, device=dev)]
            nq = self.num_queries
            # spsedt.py:65-67 as ONE launch (csrc/misc.hip: spsedt_dec_in): 2 * query + patch * Bernoulli(1 - mask_ratio) mask, drawn in the launch
            tok = Fn.SpDecInFn.apply(pq, qe, query_mask, bs, nq, bnp, qpp, True, float(self.mask_ratio), dt)
            am = self.attention_mask
        else:
            nq = bnp * self.num_queries // self.num_patches
            qe = self.query_embed.weight[start:nq, :]                       # spsedt.py:74 (with dec_at the reference's shapes do not add up either)
            if qe.shape[0] != nq:
                raise ValueError(f'query_embed.weight[{start}:{nq}] has {qe.shape[0]} rows for {nq} patch queries (spsedt.py:74)')
            tok = Fn.SpDecInFn.apply(pq, qe, None, bs, nq, bnp, qpp, False, 0.0, dt)
            am = self.attention_mask[:nq, :nq]
        dec_in = tok.view(bs, nq, d).permute(1, 0, 2)                       # (Q, B, d) view of the token-major rows: the transformer's own
        #                                                                     permute(1, 0, 2).reshape(B * Q, d) is then a zero-copy view again
        hs, memory = self.transformer(self.input_proj(src), mask, dec_in, pos[-1], decoder_mask=am)
        if torch.is_grad_enabled() and hs.requires_grad:
            # three (two) heads read hs: one gradient sum in the backward instead of autograd's pairwise adds (functional.FanoutFn)
            hs_c, hs_b, hs_f = Fn.FanoutFn.apply(hs, 3, dt) if self.feature_recon else (Fn.FanoutFn.apply(hs, 2, dt) + (None,))
        else:
            hs_c = hs_b = hs_f = hs
        outputs_class = self.class_embed(hs_c, out_f32=True)
        outputs_coord = self.bbox_embed(hs_b, final_act=ACT_SIGMOID, out_f32=True)
        if self.feature_recon:
            outputs_feature = self.feature_align(hs_f, out_f32=True)
            out = {'pred_logits': outputs_class[-1], 'pred_feature': outputs_feature[-1], 'gt_feature': gt,
                   'pred_boxes': outputs_coord[-1]}
            if self.aux_loss:
                out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b, 'pred_feature': c, 'gt_feature': gt}
                                      for a, b, c in zip(outputs_class[:-1], outputs_coord[:-1], outputs_feature[:-1])]
                out['_stacked'] = (outputs_class, outputs_coord)      # all decoder layers, for the fused criterion kernels
                out['_stacked_feature'] = outputs_feature
        else:
            out = {'pred_logits': outputs_class[-1], 'pred_boxes': outputs_coord[-1]}
            if self.aux_loss:
                out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b}
                                      for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
                out['_stacked'] = (outputs_class, outputs_coord)
        return out
