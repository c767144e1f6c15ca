"""Backbone modules on the HIP path - counterpart of reference sedt/backbone.py.

Module/parameter names equal the reference's (``backbone.0.body.conv0.weight`` ...), so its
checkpoints load.  The conv/BN modules here only HOLD parameters and buffers; the arithmetic runs in
``functional.StemFn`` / ``functional.StageFn`` (NHWC implicit-GEMM kernels with the FrozenBatchNorm
affine folded into the GEMM epilogue)."""
import torch
from torch import nn

from .. import functional as Fn
from .. import ops, runtime
from ..utilities.utils import NestedTensor, derived_from_static_mask
from .position_encoding import build_position_encoding


class FrozenBatchNorm2d(nn.Module):
    """reference backbone.py:17-53: statistics and affine are buffers; eps 1e-5 added before rsqrt.
    Folded into scale/bias by the sedt_bn_fold kernel each forward (buffers may be reloaded at any time)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer("weight", torch.ones(n))
        self.register_buffer("bias", torch.zeros(n))
        self.register_buffer("running_mean", torch.zeros(n))
        self.register_buffer("running_var", torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *args):
        state_dict.pop(prefix + 'num_batches_tracked', None)            # backbone.py:33-41
        super()._load_from_state_dict(state_dict, prefix, *args)

    def tensors(self):
        return (self.weight, self.bias, self.running_mean, self.running_var)


class Bottleneck(nn.Module):
    """parameter holder of a torchvision v1.5 Bottleneck (1x1 -> 3x3 carrying stride/dilation -> 1x1 x4)"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = FrozenBatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = FrozenBatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = FrozenBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.cfg = Fn.BlockCfg(inplanes, planes, stride, dilation, downsample is not None)

    def tensors(self):
        t = [self.conv1.weight, *self.bn1.tensors(), self.conv2.weight, *self.bn2.tensors(), self.conv3.weight,
             *self.bn3.tensors()]
        if self.downsample is not None:
            t += [self.downsample[0].weight, *self.downsample[1].tensors()]
        return t


class ResNet50Body(nn.Module):
    """conv0 + ResNet-50 (layer4 dilated) cut after layer4: what IntermediateLayerGetter keeps in the reference
    (backbone.py:97-113, 66-69)."""

    def __init__(self, dilation=True):
        super().__init__()
        self.conv0 = nn.Conv2d(1, 3, 1)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = FrozenBatchNorm2d(64)
        self.inplanes, self.dilation = 64, 1
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 6, stride=2)
        self.layer4 = self._make_layer(512, 3, stride=2, dilate=dilation)
        # set by SEDT: its input_proj returns the feature-map gradient already masked by [feature > 0]
        self.premasked_consumer = False
        # token-level outputs of layer1..layer4 of the last forward, kept only on request (engine.GraphedTrainStep cuts the
        # backward after layer2 to overlap the gradient all-reduce of everything above with the backward of everything
        # below); holding them keeps the forward's autograd graph alive, so the default is off
        self.keep_stage_out = False
        self.stage_out = [None] * 4

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               FrozenBatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, ds, prev)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)

    def forward(self, x, premasked=False):
        """x (B,1,T,F) f32 -> feature map (B,2048,H,W) in the compute dtype, channels-last strides (NHWC memory).
        premasked: the caller promises that the gradient it returns for the feature map is already multiplied by
        [feature > 0] (SEDT.input_proj does, fused in its dgrad epilogue)"""
        if not x.is_cuda:
            raise RuntimeError('the SEDT backbone runs on the MI355X HIP path only (no CPU fallback)')
        dt = runtime.compute_dtype()
        B, _, T, Fq = x.shape
        tok = Fn.StemFn.apply(x, self.conv0.weight, self.conv0.bias, self.conv1.weight, *self.bn1.tensors(), dt)
        H, W = (T - 1) // 2 + 1, (Fq - 1) // 2 + 1            # conv1 7x7 s2 p3
        H, W = (H - 1) // 2 + 1, (W - 1) // 2 + 1              # maxpool 3x3 s2 p1
        bits = None                                           # sign bits of the running stage output (functional.StageFn)
        chain = ops.BackwardChain()                           # (a stage's closing reduce launch touches what the stage below streams first)
        for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            blocks = [b.cfg for b in layer]
            # consumers of a stage output (the next stage, or SEDT.input_proj) return gradients already masked by the
            # stage's final ReLU, so the stage backward need not mask again (premasked=False -> standalone use)
            holder = {}
            meta = dict(dt=dt, B=B, H=H, W=W, blocks=blocks, mask_input=li > 0,
                        grad_premasked=True if li < 3 else (premasked or self.premasked_consumer), x_bits=bits, holder=holder,
                        chain=chain)
            ts = [t for b in layer for t in b.tensors()]
            tok = Fn.StageFn.apply(tok, meta, *ts)
            bits = holder.get('bits')
            if self.keep_stage_out:
                self.stage_out[li] = tok
            for c in blocks:
                H, W = (H - 1) // c.stride + 1, (W - 1) // c.stride + 1
        C = tok.shape[1]
        self.out_bits = bits                                  # (for a consumer that masks its input gradient itself: SEDT.input_proj)
        return tok.view(B, H, W, C).permute(0, 3, 1, 2)


class BackboneBase(nn.Module):
    """reference backbone.py:56-86"""

    def __init__(self, body: nn.Module, train_backbone: bool, num_channels: int):
        super().__init__()
        for name, parameter in body.named_parameters():
            if not train_backbone or ('conv0' not in name and 'layer2' not in name and 'layer3' not in name
                                      and 'layer4' not in name):
                parameter.requires_grad_(False)                     # backbone.py:60-62
        self.body = body
        self.num_channels = num_channels

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            x = self.body(tensor_list.tensors)
            m = tensor_list.mask
            assert m is not None
            mask = derived_from_static_mask(m, ('resize', x.shape[-2], x.shape[-1]),
                                            lambda: ops.mask_resize(m.contiguous().view(torch.uint8), x.shape[-2], x.shape[-1]).view(torch.bool))
            return {'0': NestedTensor(x, mask)}
        return {'0': self.body(tensor_list)}


class Backbone(BackboneBase):
    """ResNet-50 backbone with frozen BatchNorm (reference backbone.py:89-113).  Pretrained ImageNet weights cannot be
    downloaded offline: initialise with load_state_dict or a seeded init."""

    def __init__(self, name: str, train_backbone: bool, return_interm_layers: bool, dilation: bool):
        if name != 'resnet50':
            raise ValueError(f'only resnet50 is on the HIP path (got {name})')
        if return_interm_layers:
            raise ValueError('return_interm_layers is not used by SEDT')
        super().__init__(ResNet50Body(dilation), train_backbone, 2048)


class Joiner(nn.Sequential):
    """reference backbone.py:116-132"""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def forward(self, tensor_list):
        if isinstance(tensor_list, NestedTensor):
            xs = self[0](tensor_list)
            out, pos = [], []
            for _, x in xs.items():
                out.append(x)
                pos.append(self[1](x))
            return out, pos
        return list(self[0](tensor_list).values())


def build_backbone(args):
    position_embedding = build_position_encoding(args)
    train_backbone = args.lr_backbone > 0
    backbone = Backbone(args.backbone, train_backbone, False, args.dilation)
    model = Joiner(backbone, position_embedding)
    model.num_channels = backbone.num_channels
    return model
