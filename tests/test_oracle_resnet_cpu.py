"""CPU: the oracle's ResNet-50 restatement against an INDEPENDENT implementation of the same published architecture.

torchvision is absent from this image and the reference does not pin its version, so the backbone arithmetic (85 % of the path's
flops; reference call site sedt/backbone.py:98-100) cannot be pinned by a fixture generated from the reference ("parity unpinned",
oracle/__init__.py).  What CAN be checked: HuggingFace ``transformers`` ships its own ResNet (``ResNetModel``, v1.5 bottleneck: the
3x3 carries the stride when ``downsample_in_bottleneck=False``), written independently of torchvision's resnet.py and of this
repository.  With the oracle's weights remapped into it the two must agree on every stage output:

* the non-dilated ResNet-50 (HF has no dilation option): stem + layer1..layer4, stage by stage;
* the dilated layer4 of the reference's configuration (``replace_stride_with_dilation=[F, F, True]``): HF's stage 4 with its three
  3x3 convolutions (and the projection) re-parameterised in place - stride 1, dilation / padding (1, 2, 2) - which is torchvision's
  rule as published (block 0 keeps the previous dilation, the later blocks take the new one).

BatchNorm2d in eval mode with eps 1e-5 is FrozenBatchNorm2d's affine (backbone.py:43-53).  Tolerance 2e-5 of the stage maximum (f32
summation order differs between the two module trees only through the BN formulation)."""
import pytest
import torch

from oracle import sedt_oracle as O

transformers = pytest.importorskip('transformers')


def _oracle_body(dilation, seed):
    body = O.ResNet50Body(dilation=dilation).eval()
    sd = O.seeded_state_dict(body.state_dict(), seed)
    g = torch.Generator().manual_seed(seed + 1)
    for k in sd:                         # non-trivial frozen statistics: the BN fold must matter
        if k.endswith('running_var'):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=g)
        elif k.endswith('running_mean'):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
        elif '.bn' in k or k.startswith('bn') or 'downsample.1' in k:
            sd[k] = (1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)) if k.endswith('weight') else 0.1 * torch.randn(sd[k].shape, generator=g)
    body.load_state_dict(sd)
    return body


def _hf_from_oracle(body):
    from transformers import ResNetConfig, ResNetModel
    hf = ResNetModel(ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                                  layer_type='bottleneck', downsample_in_bottleneck=False)).eval()
    src = body.state_dict()
    dst = {}

    def conv_bn(hf_prefix, conv_key, bn_prefix):
        dst[hf_prefix + '.convolution.weight'] = src[conv_key]
        for f in ('weight', 'bias', 'running_mean', 'running_var'):
            dst[hf_prefix + '.normalization.' + f] = src[bn_prefix + '.' + f]
        dst[hf_prefix + '.normalization.num_batches_tracked'] = torch.tensor(0)

    conv_bn('embedder.embedder', 'conv1.weight', 'bn1')
    for s, nblk in enumerate((3, 4, 6, 3)):
        for b in range(nblk):
            o, h = f'layer{s + 1}.{b}', f'encoder.stages.{s}.layers.{b}'
            for i in range(3):
                conv_bn(f'{h}.layer.{i}', f'{o}.conv{i + 1}.weight', f'{o}.bn{i + 1}')
            if b == 0:
                conv_bn(f'{h}.shortcut', f'{o}.downsample.0.weight', f'{o}.downsample.1')
    missing, unexpected = hf.load_state_dict(dst, strict=True)
    assert not missing and not unexpected
    return hf


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max()).item()


def _oracle_stages(body, x3):
    """the oracle body from conv1 on (conv0 is the reference's own 1 -> 3 channel adapter, backbone.py:102; HF's stem takes 3 channels)"""
    out = {}
    x = body.relu(body.bn1(body.conv1(x3)))
    x = body.maxpool(x)
    out['stem'] = x
    for name in ('layer1', 'layer2', 'layer3', 'layer4'):
        x = getattr(body, name)(x)
        out[name] = x
    return out


@pytest.mark.parametrize('H,W', [(96, 64), (125, 64)])
def test_oracle_resnet50_equals_huggingface_resnet_stage_by_stage(H, W):
    torch.manual_seed(0)
    body = _oracle_body(False, 5)
    hf = _hf_from_oracle(body)
    x3 = torch.randn(2, 3, H, W, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        mine = _oracle_stages(body, x3)
        theirs = hf(x3, output_hidden_states=True).hidden_states       # (stem, stage1..4)
    assert len(theirs) == 5
    for name, t in zip(('stem', 'layer1', 'layer2', 'layer3', 'layer4'), theirs):
        assert mine[name].shape == t.shape, (name, mine[name].shape, t.shape)
        assert _rel(mine[name], t) < 2e-5, (name, _rel(mine[name], t))
    # 23,454,918 parameters with conv0 and without fc / BN affine as parameters: conv weights must account for all of it
    n_conv = sum(p.numel() for p in body.parameters())
    assert n_conv == 23_454_918


def test_oracle_dilated_layer4_equals_huggingface_stage4_with_dilated_convs():
    """the reference's configuration: layer4's stride replaced by dilation 2 (block 0: stride 1, dilation 1; blocks 1, 2: dilation 2)"""
    body = _oracle_body(True, 6)
    hf = _hf_from_oracle(body)
    stage = hf.encoder.stages[3]
    for b, blk in enumerate(stage.layers):
        c = blk.layer[1].convolution
        c.stride = (1, 1)
        d = 1 if b == 0 else 2
        c.dilation, c.padding = (d, d), (d, d)
    stage.layers[0].shortcut.convolution.stride = (1, 1)
    x3 = torch.randn(2, 3, 125, 64, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        mine = _oracle_stages(body, x3)
        h = hf.embedder(x3)
        for s in hf.encoder.stages:
            h = s(h)
    assert mine['layer4'].shape == h.shape == (2, 2048, 8, 4)          # the stride-16 map the reference's transformer sees
    assert _rel(mine['layer4'], h) < 2e-5
