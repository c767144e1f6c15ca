"""DETR-style transformer on the HIP path - counterpart of reference sedt/transformer.py.

Same module tree / parameter names (nn.MultiheadAttention instances hold in_proj_weight, in_proj_bias and
out_proj exactly as in the reference), same ``Transformer.forward`` signature and return values.  Tokens are
kept batch-first [B*S, d] internally; every layer is one autograd Function over HIP kernels."""
import copy
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as Fn
from .. import runtime


def _tokens(x):
    """(B,C,H,W) any strides -> [B*H*W, C] (zero-copy when the memory is NHWC)"""
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C)


def _u8(mask):
    return None if mask is None else mask.contiguous().view(torch.uint8)


def _activation_name(activation):
    """reference transformer.py:423-431 accepts "relu" / "gelu" / "glu".  relu: fused into linear1's epilogue (and the slab encoder);
    gelu: sedt_gelu_fwd / sedt_gelu_bwd around the per-op FFN GEMMs; glu halves the hidden width, so the reference's own linear2
    (dim_feedforward -> d_model) rejects its output - refused at construction here instead of at the first forward"""
    if activation in ("relu", "gelu"):
        return activation
    if activation == "glu":
        raise ValueError('activation "glu" halves the FFN width: linear2 of the reference layer (transformer.py:160) fails on it too')
    raise RuntimeError(F"activation should be relu/gelu, not {activation}.")


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.activation = _activation_name(activation)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.normalize_before = normalize_before
        self.nhead, self.p = nhead, dropout

    def params(self):
        a = self.self_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.linear1.weight,
                self.linear1.bias, self.linear2.weight, self.linear2.bias, self.norm1.weight, self.norm1.bias,
                self.norm2.weight, self.norm2.bias)

    def forward_tokens(self, x, pos, kpm, B, S, src_mask=None, chain=None, wg_share=None, layer_idx=0):
        """chain (ops.BackwardChain or None): links this layer's backward to the one that runs after it (weight prefetch hints);
        wg_share (ops.WgradShare or None): the stack's shared weight-gradient batch, flushed by layer 0's backward"""
        cfg = dict(dt=runtime.compute_dtype(), B=B, S=S, H=self.nhead, dropout=self.p, training=self.training,
                   pre_norm=self.normalize_before, chain=chain, act=self.activation, wg_share=wg_share, layer_idx=layer_idx)
        return Fn.EncoderLayerFn.apply(x, pos, kpm, src_mask, cfg, *self.params())

    def forward(self, src, src_mask: Optional[Tensor] = None, src_key_padding_mask: Optional[Tensor] = None,
                pos: Optional[Tensor] = None):
        """seq-first (S,B,d) interface of the reference layer"""
        S, B, d = src.shape
        x = src.transpose(0, 1).reshape(B * S, d)
        p = (pos if pos is not None else torch.zeros_like(src)).transpose(0, 1).reshape(B * S, d)
        y = self.forward_tokens(x, p, _u8(src_key_padding_mask), B, S, src_mask)
        return y.view(B, S, d).transpose(0, 1)


class TransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", normalize_before=False):
        super().__init__()
        self.activation = _activation_name(activation)
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
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
        self.nhead, self.p = nhead, dropout

    def params(self):
        a, c = self.self_attn, self.multihead_attn
        return (a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias,
                c.in_proj_weight, c.in_proj_bias, c.out_proj.weight, c.out_proj.bias,
                self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias, self.norm3.weight, self.norm3.bias)

    def forward_tokens(self, tgt, mem, mem_pos, qpos, kpm, B, S, Q, tgt_mask=None, kv_fused=False, out=None, acc=None, share=None,
                       layer_idx=0, chain=None, wg_share=None):
        """kv_fused: mem_pos is mem + a constant (the decoder's own sine position add): the key and value input gradients
        of the cross-attention are returned as ONE tensor on `mem` (one K = 2E GEMM) and nothing on `mem_pos`.
        acc: (GradAccumulator for mem, GradAccumulator for qpos) shared by the layers of one decoder pass, or None"""
        cfg = dict(kv_fused=kv_fused, dt=runtime.compute_dtype(), B=B, S=S, Q=Q, H=self.nhead, dropout=self.p, training=self.training,
                   pre_norm=self.normalize_before, out=out, acc_mem=None if acc is None else acc[0],
                   acc_qpos=None if acc is None else acc[1], share=share, layer_idx=layer_idx, chain=chain, act=self.activation,
                   wg_share=wg_share)
        return Fn.DecoderLayerFn.apply(tgt, mem, mem_pos, qpos, kpm, tgt_mask, cfg, *self.params())


def _get_clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


class TransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward_tokens(self, x, pos, kpm, B, S, chain=None):
        from .. import ops
        # one weight-gradient launch pair for the whole stack (inside a captured stepper, when every layer trains: the launch is issued
        # by layer 0's backward, which must therefore run)
        wg = ops.WgradShare.make() if (torch.is_grad_enabled() and len(self.layers) > 1
                                       and all(p.requires_grad for l in self.layers for p in l.parameters())) else None
        for li, layer in enumerate(self.layers):
            x = layer.forward_tokens(x, pos, kpm, B, S, chain=chain, wg_share=wg, layer_idx=li)
        if self.norm is not None:
            x = Fn.LayerNormFn.apply(x, self.norm.weight, self.norm.bias, runtime.compute_dtype())
        return x


class TransformerDecoder(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward_tokens(self, tgt, mem, pos, qpos, kpm, B, S, Q, tgt_mask=None, chain=None):
        """returns (L, B, Q, d): the shared LayerNorm applied to every layer's output (transformer.py:134-147)"""
        dt = runtime.compute_dtype()
        mem_pos = Fn.AddFn.apply(mem, pos, 0, dt)              # key input of every cross-attention
        qpos = Fn.CastFn.apply(qpos, dt)                       # once for all layers (their gradients add up in the compute dtype)
        out = tgt
        outs = []
        R, d, n = tgt.shape[0], tgt.shape[1], len(self.layers)
        # pre-norm layers end in a GEMM (FFN linear2 + residual): it writes straight into its row range of ONE buffer, so the
        # shared LayerNorm of every layer's output (transformer.py:134-147) is ONE call over all rows without a torch.cat
        stack = (torch.empty((n * R, d), device=tgt.device, dtype=runtime.torch_dtype())
                 if (self.return_intermediate and n > 1 and all(l.normalize_before for l in self.layers)) else None)
        # the layers all read mem and qpos: their gradient shares are folded as the backward goes (functional.GradAccumulator)
        # instead of being added pair by pair by autograd (pre-norm layers; every layer of the pass must take part)
        acc = ((Fn.GradAccumulator(n), Fn.GradAccumulator(n)) if (torch.is_grad_enabled() and all(l.normalize_before for l in self.layers)
                                                                  and not pos.requires_grad) else None)
        share = {} if (stack is not None and acc is not None) else None      # (see functional.StackViewFn)
        from .. import ops
        wg = ops.WgradShare.make() if (torch.is_grad_enabled() and n > 1 and all(p.requires_grad for l in self.layers for p in l.parameters())) else None
        for li, layer in enumerate(self.layers):
            out = layer.forward_tokens(out, mem, mem_pos, qpos, kpm, B, S, Q, tgt_mask, kv_fused=not pos.requires_grad,
                                       out=None if stack is None else stack[li * R:(li + 1) * R], acc=acc, share=share, layer_idx=li,
                                       chain=chain, wg_share=wg)
            outs.append(out)
        if self.return_intermediate:
            stacked = Fn.StackViewFn.apply(stack, share, *outs) if stack is not None else (torch.cat(outs) if n > 1 else outs[0])
            return Fn.LayerNormFn.apply(stacked, self.norm.weight, self.norm.bias, dt).view(n, B, Q, d)
        return Fn.LayerNormFn.apply(out, self.norm.weight, self.norm.bias, dt).view(1, B, Q, d)


class Transformer(nn.Module):
    def __init__(self, d_model=512, nhead=8, num_encoder_layers=6, num_decoder_layers=6, dim_feedforward=2048,
                 dropout=0.1, activation="relu", normalize_before=False, return_intermediate_dec=False, self_sup=False):
        super().__init__()
        if d_model // nhead != 32 or d_model % nhead:
            raise ValueError('the HIP attention kernels implement head dim 32 (d_model / nhead)')
        enc_layer = TransformerEncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        enc_norm = nn.LayerNorm(d_model) if normalize_before else None
        self.encoder = TransformerEncoder(enc_layer, num_encoder_layers, enc_norm)
        dec_layer = TransformerDecoderLayer(d_model, nhead, dim_feedforward, dropout, activation, normalize_before)
        self.decoder = TransformerDecoder(dec_layer, num_decoder_layers, nn.LayerNorm(d_model),
                                          return_intermediate=return_intermediate_dec)
        self._reset_parameters()
        self.d_model, self.nhead, self.self_sup = d_model, nhead, self_sup

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def _zero_tokens(self, rows, cols, device):
        """the decoder's all-zero input (transformer.py:67): one buffer per shape, kept and never written (a fill launch per
        forward otherwise); created fresh while a HIP graph is being captured (see utilities.utils.derived_from_static_mask)"""
        key = (rows, cols, runtime.torch_dtype(), str(device))
        z = self.__dict__.setdefault('_zeros', {}).get(key)
        if z is not None:
            from ..utilities.utils import pin_if_capturing
            return pin_if_capturing(z)           # (a graph keeps the raw pointer: survive the eviction below)
        if z is None:
            z = torch.zeros((rows, cols), device=device, dtype=runtime.torch_dtype())
            if not (z.is_cuda and torch.cuda.is_current_stream_capturing()):
                if len(self._zeros) > 8:
                    self._zeros.clear()
                self._zeros[key] = z
        return z

    def forward(self, src, mask, query_embed, pos_embed, enc_at_embed=None, decoder_mask=None):
        """reference transformer.py:48-86.  src/pos_embed (B,C,H,W); mask (B,H,W) bool; query_embed (Q,C), or
        (Q,B,C) when self_sup.  Returns hs (L,B,Q,C) and memory (B,S,C) [self_sup: (B,C,H,W)]."""
        if enc_at_embed is not None:
            raise NotImplementedError('enc_at_embed is unreachable from SEDT.forward (sedt.py:88) and not built')
        if not src.is_cuda:
            raise RuntimeError('the SEDT transformer runs on the MI355X HIP path only (no CPU fallback)')
        dt = runtime.compute_dtype()
        B, C, H, W = src.shape
        S = H * W
        x = _tokens(src)
        pos = _tokens(pos_embed)
        kpm = _u8(mask.flatten(1))
        tgt = None
        if self.self_sup:
            Q = query_embed.shape[0]
            qpos = query_embed.permute(1, 0, 2).reshape(B * Q, C)
        else:
            Q = query_embed.shape[0]
            tgt = self._zero_tokens(B * Q, C, src.device)
            qpos = Fn.BroadcastRowsFn.apply(query_embed, tgt, B, dt)        # (Q,C) -> [B*Q, C] in the compute dtype, one launch
        if tgt is None:
            tgt = self._zero_tokens(B * Q, C, src.device)
        from .. import ops
        chain = ops.BackwardChain()          # (a layer's closing reduce launch touches what the layer below streams first in its backward)
        memory = self.encoder.forward_tokens(x, pos, kpm, B, S, chain=chain)
        # the encoder output as a token matrix, kept on request: engine's data-parallel steppers cut the backward here (gradients of
        # decoder + heads are all-reduced while the encoder's backward runs)
        self.cut_memory = memory if getattr(self, 'keep_cut', False) else None
        tmask = decoder_mask.float().contiguous() if (self.self_sup and decoder_mask is not None) else None
        hs = self.decoder.forward_tokens(tgt, memory, pos, qpos, kpm, B, S, Q, tmask, chain=chain)
        if self.self_sup:
            return hs, memory.view(B, H, W, C).permute(0, 3, 1, 2)
        return hs, memory.view(B, S, C)


def build_transformer(args):
    return Transformer(d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
                       dim_feedforward=args.dim_feedforward, num_encoder_layers=args.enc_layers,
                       num_decoder_layers=args.dec_layers, normalize_before=args.pre_norm,
                       return_intermediate_dec=True, self_sup=args.self_sup)
