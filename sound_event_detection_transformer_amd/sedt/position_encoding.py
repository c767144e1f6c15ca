"""Position encoding on the HIP path - counterpart of reference sedt/position_encoding.py."""
import math

import torch
from torch import nn

from .. import ops, runtime
from ..utilities.utils import NestedTensor, derived_from_static_mask


class PositionEmbeddingSine(nn.Module):
    """reference position_encoding.py:11-47: sine embedding over the TIME axis only, normalised, interleaved sin/cos."""

    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        if temperature != 10000 or not normalize or (scale is not None and abs(scale - 2 * math.pi) > 1e-9):
            raise ValueError('the HIP position encoding implements the reference configuration only '
                             '(temperature 10000, normalize=True, scale 2*pi)')
        self.num_pos_feats = num_pos_feats

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        assert mask is not None
        B, H, W = mask.shape
        dt = runtime.compute_dtype()
        pos = derived_from_static_mask(mask, ('posenc', dt, self.num_pos_feats),
                                       lambda: ops.posenc(dt, mask.contiguous().view(torch.uint8), self.num_pos_feats))   # (B, H*W, D)
        return pos.view(B, H, W, self.num_pos_feats).permute(0, 3, 1, 2)


def build_position_encoding(args):
    if args.position_embedding in ('v2', 'sine'):
        return PositionEmbeddingSine(args.hidden_dim, normalize=True)      # position_encoding.py:80-83
    raise ValueError(f"not supported {args.position_embedding} (the reference's learned embedding is unusable: "
                     "it yields 2*hidden_dim channels)")
