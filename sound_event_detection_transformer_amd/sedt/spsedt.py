"""SP-SEDT pre-training model on the HIP path - counterpart of reference sedt/spsedt.py."""
import torch
from torch import nn

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
        pq = self.patch2query(gt).float().view(bs, bnp, 1, -1).repeat(1, 1, self.num_queries // self.num_patches, 1) \
            .flatten(1, 2).permute(1, 0, 2).contiguous()                     # (Q, B, d)
        start = 1 if self.dec_at else 0
        if self.training:
            qe = self.query_embed.weight[start:, :]
            if self.query_shuffle:                                          # spsedt.py:60: the queries are permuted
                qe = qe[torch.randperm(self.num_queries, device=dev)]
            if query_mask is None:
                query_mask = (torch.rand(self.num_queries, bs, 1, device=dev) > self.mask_ratio).float()
            dec_in = (qe * 2).unsqueeze(1) + pq * query_mask.to(dev)        # spsedt.py:66-67: 2*query + patch*mask
            am = self.attention_mask
        else:
            nq = bnp * self.num_queries // self.num_patches
            dec_in = pq + self.query_embed.weight[start:nq, :].unsqueeze(1)
            am = self.attention_mask[:nq, :nq]
        hs, memory = self.transformer(self.input_proj(src), mask, dec_in, pos[-1], decoder_mask=am)
        outputs_class = self.class_embed(hs, out_f32=True)
        outputs_coord = self.bbox_embed(hs, final_act=ACT_SIGMOID, out_f32=True)
        if self.feature_recon:
            outputs_feature = self.feature_align(hs, out_f32=True)
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
