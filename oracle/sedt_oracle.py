"""Pure-PyTorch CPU restatement of the SEDT model path (TEST INFRASTRUCTURE).

Every class cites the reference file:line (under /root/reference) whose
behaviour it restates.  ``state_dict`` key names are identical to the
reference's, so one seeded weight set loads into the reference, this oracle and
the HIP product alike.

Attention is written out as explicit matmuls instead of going through
``nn.MultiheadAttention`` so that the arithmetic the HIP kernels must reproduce
is visible here (scale on q, additive -inf masks, softmax, dropout on the
probabilities, output projection).
"""
import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn, Tensor


# --------------------------------------------------------------------------
# boundary types  (reference utilities/utils.py:470-492, 526-560)
# --------------------------------------------------------------------------
class NestedTensor(object):
    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        m = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), m)

    def decompose(self):
        return self.tensors, self.mask

    def __getitem__(self, i):
        if isinstance(i, slice):
            return NestedTensor(self.tensors[i], self.mask[i])


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """utils.py:470-492 — zero-pad (C,T,F) clips to the batch max, mask True on padding."""
    if tensor_list[0].ndim != 3:
        raise ValueError('not supported')
    shapes = [list(t.shape) for t in tensor_list]
    c = max(s[0] for s in shapes)
    h = max(s[1] for s in shapes)
    w = max(s[2] for s in shapes)
    b = len(tensor_list)
    tensor = torch.zeros((b, c, h, w), dtype=tensor_list[0].dtype, device=tensor_list[0].device)
    mask = torch.ones((b, h, w), dtype=torch.bool, device=tensor_list[0].device)
    for i, img in enumerate(tensor_list):
        tensor[i, :img.shape[0], :img.shape[1], :img.shape[2]].copy_(img)
        mask[i, :img.shape[1], :img.shape[2]] = False
    return NestedTensor(tensor, mask)


# --------------------------------------------------------------------------
# backbone  (reference sedt/backbone.py; torchvision resnet.py restated)
# --------------------------------------------------------------------------
class FrozenBatchNorm2d(nn.Module):
    """backbone.py:17-53 — affine from buffers, eps 1e-5 added before rsqrt."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + 'num_batches_tracked', None)   # backbone.py:33-41
        super()._load_from_state_dict(state_dict, prefix, *args)

    def forward(self, x):
        scale = self.weight * (self.running_var + 1e-5).rsqrt()
        bias = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


class Bottleneck(nn.Module):
    """torchvision resnet.py Bottleneck, "v1.5": 1x1 -> 3x3 (carries the stride,
    padding = dilation) -> 1x1 x4, downsample = 1x1 conv(stride) + norm."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = norm_layer(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = norm_layer(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet50Body(nn.Module):
    """conv0 + torchvision resnet50(replace_stride_with_dilation=[F,F,dilation]) cut after
    layer4 — what IntermediateLayerGetter keeps (backbone.py:97-113, 66-69).  Module
    names (= state_dict keys) match the reference's ``backbone.0.body.*``."""

    def __init__(self, dilation=True, norm_layer=FrozenBatchNorm2d):
        super().__init__()
        self.conv0 = nn.Conv2d(1, 3, 1)                                   # backbone.py:102
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.inplanes, self.dilation = 64, 1
        self._norm = norm_layer
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 6, stride=2)
        self.layer4 = self._make_layer(512, 3, stride=2, dilate=dilation)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        previous_dilation = self.dilation
        if dilate:                       # stride replaced by dilation for blocks 1..
            self.dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                       self._norm(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, previous_dilation, self._norm)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation, norm_layer=self._norm))
        return nn.Sequential(*layers)

    def forward(self, x, return_stages=False):
        stages = {}
        x = self.conv0(x)
        x = self.relu(self.bn1(self.conv1(x)))
        stages['conv1'] = x
        x = self.maxpool(x)
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            x = getattr(self, name)(x)
            stages[name] = x
        return (x, stages) if return_stages else x


class BackboneBase(nn.Module):
    """backbone.py:56-86 — freeze rule and mask nearest-resize."""

    def __init__(self, train_backbone=True, dilation=True):
        super().__init__()
        self.body = ResNet50Body(dilation)
        for name, p in self.body.named_parameters():
            if not train_backbone or ('conv0' not in name and 'layer2' not in name
                                      and 'layer3' not in name and 'layer4' not in name):
                p.requires_grad_(False)                                   # backbone.py:60-62
        self.num_channels = 2048

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            x = self.body(tensor_list.tensors)
            m = tensor_list.mask
            assert m is not None
            mask = F.interpolate(m[None].float(), size=x.shape[-2:]).to(torch.bool)[0]   # backbone.py:81
            return {'0': NestedTensor(x, mask)}
        return {'0': self.body(tensor_list)}


class PositionEmbeddingSine(nn.Module):
    """position_encoding.py:11-47 — time-axis only, normalised, interleaved sin/cos."""

    def __init__(self, num_pos_feats=256, temperature=10000, normalize=True, scale=None):
        super().__init__()
        self.num_pos_feats, self.temperature, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        y_embed = (~mask).cumsum(1, dtype=torch.float32)
        if self.normalize:
            y_embed = y_embed / (y_embed[:, -1:, :] + 1e-6) * self.scale
        dim_t = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_pos_feats)
        pos_y = y_embed[:, :, :, None] / dim_t
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).flatten(3)
        return pos_y.permute(0, 3, 1, 2)


class Joiner(nn.Sequential):
    """backbone.py:116-132."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)
        self.num_channels = backbone.num_channels

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            xs = self[0](tensor_list)
            out, pos = [], []
            for _, x in xs.items():
                out.append(x)
                pos.append(self[1](x).to(x.tensors.dtype))
            return out, pos
        return list(self[0](tensor_list).values())


# --------------------------------------------------------------------------
# transformer  (reference sedt/transformer.py)
# --------------------------------------------------------------------------
class MultiheadAttention(nn.Module):
    """torch.nn.MultiheadAttention as called at transformer.py:160,220-221 (seq-first,
    packed in_proj, need_weights path): q*dh^-0.5, additive masks, softmax, dropout, out_proj."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None):
        L, B, E = query.shape
        S = key.shape[0]
        h, dh = self.num_heads, self.head_dim
        w, b = self.in_proj_weight, self.in_proj_bias
        q = F.linear(query, w[:E], b[:E])
        k = F.linear(key, w[E:2 * E], b[E:2 * E])
        v = F.linear(value, w[2 * E:], b[2 * E:])
        q = q.reshape(L, B * h, dh).transpose(0, 1) * (1.0 / math.sqrt(dh))
        k = k.reshape(S, B * h, dh).transpose(0, 1)
        v = v.reshape(S, B * h, dh).transpose(0, 1)
        scores = torch.bmm(q, k.transpose(1, 2))                         # (B*h, L, S)
        if attn_mask is not None:
            scores = scores + attn_mask.to(scores.dtype).unsqueeze(0)
        if key_padding_mask is not None:
            kpm = torch.zeros(B, 1, 1, S, dtype=scores.dtype).masked_fill(
                key_padding_mask.view(B, 1, 1, S), float('-inf'))
            scores = (scores.view(B, h, L, S) + kpm).view(B * h, L, S)
        p = F.softmax(scores, dim=-1)
        p = F.dropout(p, self.dropout, self.training)
        ctx = torch.bmm(p, v).transpose(0, 1).reshape(L, B, E)
        return self.out_proj(ctx), None


def _with_pos(t, pos):
    return t if pos is None else t + pos


def _activation(name):
    """transformer.py:423-431 (`_get_activation_fn`): "relu" / "gelu" / "glu" by name, RuntimeError otherwise.  (F.glu halves the
    hidden width, so linear2 of a layer built with it fails in the reference too; kept for the same behaviour.)"""
    if name == "relu":
        return F.relu
    if name == "gelu":
        return F.gelu
    if name == "glu":
        return F.glu
    raise RuntimeError(F"activation should be relu/gelu, not {name}.")


class TransformerEncoderLayer(nn.Module):
    """transformer.py:155-212."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, normalize_before=False, activation="relu"):
        super().__init__()
        self.activation = _activation(activation)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward(self, src, src_mask=None, src_key_padding_mask=None, pos=None):
        if self.normalize_before:                                        # forward_pre :192-204
            src2 = self.norm1(src)
            q = k = _with_pos(src2, pos)
            src2 = self.self_attn(q, k, src2, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)[0]
            src = src + self.dropout1(src2)
            src2 = self.norm2(src)
            src2 = self.linear2(self.dropout(self.activation(self.linear1(src2))))
            return src + self.dropout2(src2)
        q = k = _with_pos(src, pos)                                      # forward_post :177-190
        src2 = self.self_attn(q, k, src, attn_mask=src_mask, key_padding_mask=src_key_padding_mask)[0]
        src = self.norm1(src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout2(src2))


class TransformerDecoderLayer(nn.Module):
    """transformer.py:215-297."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, normalize_before=False, activation="relu"):
        super().__init__()
        self.activation = _activation(activation)
        self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.norm3 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.dropout3 = nn.Dropout(dropout)
        self.normalize_before = normalize_before

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, query_pos=None):
        if self.normalize_before:                                        # forward_pre :263-284
            t2 = self.norm1(tgt)
            q = k = _with_pos(t2, query_pos)
            t2 = self.self_attn(q, k, t2, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
            tgt = tgt + self.dropout1(t2)
            t2 = self.norm2(tgt)
            t2 = self.multihead_attn(_with_pos(t2, query_pos), _with_pos(memory, pos), memory,
                                     attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask)[0]
            tgt = tgt + self.dropout2(t2)
            t2 = self.norm3(tgt)
            t2 = self.linear2(self.dropout(self.activation(self.linear1(t2))))
            return tgt + self.dropout3(t2)
        q = k = _with_pos(tgt, query_pos)                                # forward_post :240-261
        t2 = self.self_attn(q, k, tgt, attn_mask=tgt_mask, key_padding_mask=tgt_key_padding_mask)[0]
        tgt = self.norm1(tgt + self.dropout1(t2))
        t2 = self.multihead_attn(_with_pos(tgt, query_pos), _with_pos(memory, pos), memory,
                                 attn_mask=memory_mask, key_padding_mask=memory_key_padding_mask)[0]
        tgt = self.norm2(tgt + self.dropout2(t2))
        t2 = self.linear2(self.dropout(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout3(t2))


class TransformerEncoder(nn.Module):
    """transformer.py:90-111."""

    def __init__(self, make_layer, num_layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.norm = norm

    def forward(self, src, mask=None, src_key_padding_mask=None, pos=None):
        out = src
        for layer in self.layers:
            out = layer(out, src_mask=mask, src_key_padding_mask=src_key_padding_mask, pos=pos)
        return self.norm(out) if self.norm is not None else out


class TransformerDecoder(nn.Module):
    """transformer.py:114-152 — the same LayerNorm on every layer's output for the stack."""

    def __init__(self, make_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = nn.ModuleList([make_layer() for _ in range(num_layers)])
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward(self, tgt, memory, tgt_mask=None, memory_key_padding_mask=None, pos=None, query_pos=None):
        out = tgt
        inter = []
        for layer in self.layers:
            out = layer(out, memory, tgt_mask=tgt_mask, memory_key_padding_mask=memory_key_padding_mask,
                        pos=pos, query_pos=query_pos)
            if self.return_intermediate:
                inter.append(self.norm(out))
        if self.return_intermediate:
            return torch.stack(inter)
        return self.norm(out).unsqueeze(0)


class Transformer(nn.Module):
    """transformer.py:18-86 (enc_at_embed branch :70-80 is unreachable from SEDT and omitted)."""

    def __init__(self, d_model=256, nhead=8, num_encoder_layers=3, num_decoder_layers=3, dim_feedforward=2048,
                 dropout=0.1, normalize_before=True, return_intermediate_dec=True, self_sup=False, activation="relu"):
        super().__init__()
        enc_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = TransformerEncoder(
            lambda: TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, normalize_before, activation),
            num_encoder_layers, enc_norm)
        self.decoder = TransformerDecoder(
            lambda: TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, normalize_before, activation),
            num_decoder_layers, nn.LayerNorm(d_model), return_intermediate=return_intermediate_dec)
        for p in self.parameters():                                      # transformer.py:42-45
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self.d_model, self.nhead, self.self_sup = d_model, nhead, self_sup

    def forward(self, src, mask, query_embed, pos_embed, enc_at_embed=None, decoder_mask=None):
        assert enc_at_embed is None
        bs, c, h, w = src.shape
        src = src.flatten(2).permute(2, 0, 1)
        pos_embed = pos_embed.flatten(2).permute(2, 0, 1)
        mask = mask.flatten(1)
        if not self.self_sup:
            query_embed = query_embed.unsqueeze(1).repeat(1, bs, 1)
        tgt = torch.zeros_like(query_embed)
        memory = self.encoder(src, src_key_padding_mask=mask, pos=pos_embed)
        hs = self.decoder(tgt, memory, tgt_mask=decoder_mask if self.self_sup else None,
                          memory_key_padding_mask=mask, pos=pos_embed, query_pos=query_embed)
        if self.self_sup:
            return hs.transpose(1, 2), memory.permute(1, 2, 0).view(bs, c, h, w)
        return hs.transpose(1, 2), memory.permute(1, 0, 2)


# --------------------------------------------------------------------------
# SEDT / SPSEDT  (reference sedt/sedt.py:17-131,398-409; sedt/spsedt.py)
# --------------------------------------------------------------------------
class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = F.relu(layer(x)) if i < self.num_layers - 1 else layer(x)
        return x


class SEDT(nn.Module):
    """sedt.py:17-131, including the --pooling variants (:47-61 construction, :96-119 forward)."""

    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, dec_at=False, pooling=None):
        super().__init__()
        self.num_queries, self.transformer = num_queries, transformer
        d = transformer.d_model
        self.class_embed = nn.Linear(d, num_classes + 1)
        self.bbox_embed = MLP(d, d, 2, 3)
        self.input_proj = nn.Conv2d(backbone.num_channels, d, kernel_size=1)
        self.backbone, self.aux_loss, self.dec_at, self.pooling = backbone, aux_loss, dec_at, pooling
        self.query_embed = nn.Embedding(num_queries + (1 if dec_at else 0), d)
        if dec_at:
            self.weak_class_embed = nn.Linear(d, num_classes)
        if pooling is not None and 'attn' in pooling:      # sedt.py:52-61
            self.attn_dense_softmax = nn.Linear(d, num_classes)

    def _pool(self, hs_last, logits_last, boxes_last):
        """sedt.py:96-106 / :112-119: clip-level probabilities from the event queries' class probabilities."""
        y = F.softmax(logits_last, -1)[:, :, :-1]                        # [B, Q, C]
        if self.dec_at and 'weighted_sum' in self.pooling:               # :98-100 (only the dec_at branch has it)
            return (y * boxes_last[:, :, 1][:, :, None]).sum(1).clip(0, 1)
        if 'attn' in self.pooling:                                       # :53-60
            sof = torch.clamp(F.softmax(self.attn_dense_softmax(hs_last), -1), min=1e-7, max=1)
            return (sof * y).sum(1) / sof.sum(1)
        if 'max' in self.pooling:                                        # AdaptiveMaxPool2d((1, None)) over the query axis
            return y.max(1)[0].squeeze()
        if 'avg' in self.pooling:                                        # AdaptiveAvgPool2d((1, None))
            return y.mean(1).squeeze()
        raise AttributeError("'SEDT' object has no attribute 'pooling_func'")     # what the reference does here

    def forward(self, samples):
        if isinstance(samples, (list, torch.Tensor)):
            samples = nested_tensor_from_tensor_list(samples)
        features, pos = self.backbone(samples)
        src, mask = features[-1].decompose()
        hs, _ = self.transformer(self.input_proj(src), mask, self.query_embed.weight, pos[-1])
        out = {}
        q0 = 1 if self.dec_at else 0
        if self.dec_at:
            outputs_class = self.class_embed(hs[:, :, 1:, :])
            outputs_coord = self.bbox_embed(hs[:, :, 1:, :]).sigmoid()
            out['at'] = self.weak_class_embed(hs[-1, :, 0, :]).squeeze().sigmoid()
        else:
            outputs_class = self.class_embed(hs)
            outputs_coord = self.bbox_embed(hs).sigmoid()
        out['pred_logits'], out['pred_boxes'] = outputs_class[-1], outputs_coord[-1]
        if self.pooling is not None:
            out['at_p'] = self._pool(hs[-1, :, q0:, :], outputs_class[-1], outputs_coord[-1])
        if self.aux_loss:
            out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b}
                                  for a, b in zip(outputs_class[:-1], outputs_coord[:-1])]
        return out


class SPSEDT(SEDT):
    """spsedt.py:14-95.  ``query_mask`` lets a test inject the Bernoulli(1-mask_ratio) mask that
    spsedt.py:65 draws with torch.rand, so train-mode parity is checkable."""

    def __init__(self, backbone, transformer, num_classes, num_queries, aux_loss=False, feature_recon=True,
                 mask_ratio=0.1, num_patches=10):
        super().__init__(backbone, transformer, num_classes, num_queries, aux_loss, dec_at=False)
        d = transformer.d_model
        self.patch2query = nn.Linear(backbone.num_channels, d)
        self.num_patches, self.mask_ratio, self.feature_recon = num_patches, mask_ratio, feature_recon
        if feature_recon:
            self.feature_align = MLP(d, d, backbone.num_channels, 2)
        assert num_queries % num_patches == 0
        qpp = num_queries // num_patches
        am = torch.ones(num_queries, num_queries) * float('-inf')
        for i in range(num_patches):
            am[i * qpp:(i + 1) * qpp, i * qpp:(i + 1) * qpp] = 0
        self.register_buffer('attention_mask', am, persistent=False)

    def forward(self, samples, patches, query_mask=None):
        bnp = patches.shape[1]
        samples = NestedTensor(samples[0], samples[1])
        feature, pos = self.backbone(samples)
        src, mask = feature[-1].decompose()
        bs = patches.shape[0]
        pf = self.backbone(patches.flatten(0, 1))
        gt = F.adaptive_avg_pool2d(pf[-1], (1, 1)).flatten(1)
        pq = self.patch2query(gt).view(bs, bnp, 1, -1).repeat(1, 1, self.num_queries // self.num_patches, 1) \
            .flatten(1, 2).permute(1, 0, 2).contiguous()
        if self.training:
            if query_mask is None:
                query_mask = (torch.rand(self.num_queries, bs, 1) > self.mask_ratio).float()
            dec_in = self.query_embed.weight.unsqueeze(1).repeat(1, bs, 1)
            dec_in = dec_in + pq * query_mask + dec_in                   # spsedt.py:66-67 (2*query + patch*mask)
            am = self.attention_mask
        else:
            nq = bnp * self.num_queries // self.num_patches
            dec_in = pq + self.query_embed.weight[:nq].unsqueeze(1).repeat(1, bs, 1)
            am = self.attention_mask[:nq, :nq]
        hs, _ = self.transformer(self.input_proj(src), mask, dec_in, pos[-1], decoder_mask=am)
        oc = self.class_embed(hs)
        ob = self.bbox_embed(hs).sigmoid()
        if self.feature_recon:
            of = self.feature_align(hs)
            out = {'pred_logits': oc[-1], 'pred_feature': of[-1], 'gt_feature': gt, 'pred_boxes': ob[-1]}
            if self.aux_loss:
                out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b, 'pred_feature': c, 'gt_feature': gt}
                                      for a, b, c in zip(oc[:-1], ob[:-1], of[:-1])]
        else:
            out = {'pred_logits': oc[-1], 'pred_boxes': ob[-1]}
            if self.aux_loss:
                out['aux_outputs'] = [{'pred_logits': a, 'pred_boxes': b} for a, b in zip(oc[:-1], ob[:-1])]
        return out


# --------------------------------------------------------------------------
# construction + canonical seeded weights
# --------------------------------------------------------------------------
def build_oracle_model(num_classes=10, num_queries=10, enc_layers=3, dec_layers=3, dec_at=True, aux_loss=True,
                       pre_norm=True, dropout=0.1, hidden_dim=256, nheads=8, dim_feedforward=2048, dilation=True,
                       self_sup=False, num_patches=10, feature_recon=True, train_backbone=True, pooling=None):
    """sedt/__init__.py:8-38 (model part)."""
    backbone = Joiner(BackboneBase(train_backbone, dilation), PositionEmbeddingSine(hidden_dim, normalize=True))
    transformer = Transformer(hidden_dim, nheads, enc_layers, dec_layers, dim_feedforward, dropout, pre_norm,
                              True, self_sup)
    if self_sup:
        return SPSEDT(backbone, transformer, 1, num_queries, aux_loss, feature_recon, num_patches=num_patches)
    return SEDT(backbone, transformer, num_classes, num_queries, aux_loss, dec_at, pooling)


def seeded_state_dict(template: dict, seed: int) -> dict:
    """Canonical deterministic weights: iterate SORTED state_dict keys with one CPU generator.

    The same function seeds the reference (in tests/golden/make_golden.py), this oracle and the
    HIP model, so no weight blobs are committed.  Scales are chosen so that activations stay
    O(1)-O(100) through 16 residual blocks with FrozenBatchNorm (no pretrained weights offline)."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(template.keys()):
        shape = tuple(template[k].shape)
        leaf = k.split('.')[-1]
        is_bn = ('.bn' in k or 'downsample.1' in k) and 'body' in k
        if is_bn:
            last = ('bn3' in k) or ('downsample.1' in k)
            if leaf == 'weight':
                lo, hi = (0.3, 0.6) if last else (0.7, 1.1)
                t = torch.rand(shape, generator=g) * (hi - lo) + lo
            elif leaf == 'running_var':
                t = torch.rand(shape, generator=g) + 0.5
            else:                                   # bias, running_mean
                t = torch.randn(shape, generator=g) * 0.1
        elif 'norm' in k:                           # LayerNorm
            t = torch.rand(shape, generator=g) * 0.4 + 0.8 if leaf == 'weight' \
                else torch.randn(shape, generator=g) * 0.05
        elif k.endswith('query_embed.weight'):
            t = torch.randn(shape, generator=g)
        elif len(shape) >= 2:                       # conv / linear / in_proj weights
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        else:                                       # biases
            t = torch.randn(shape, generator=g) * 0.05
        out[k] = t.to(template[k].dtype)
    return out
