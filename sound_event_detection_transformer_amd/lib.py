"""ctypes binding of libsedt_hip.so (the C ABI declared in include/sedt_hip.h).

The product path has no fallback: if the library is missing or a call fails, this raises."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libsedt_hip.so')
if os.environ.get('SEDT_LIB_AB') and os.environ.get('SEDT_DEV') == '1':   # developer A/B runs: another build of the same library (tools/README.md)
    LIB_PATH = os.environ['SEDT_LIB_AB']

F32, BF16 = 0, 1
BF16X3 = 2            # GEMM entry points only: f32 tensors, split-bf16 products (include/sedt_hip.h)
GEMM_X3 = False       # runtime.set_compute_dtype('bf16x3'): the f32 mode's contractions go through the BF16X3 code


def gemm_dtype(dtype):
    """the dtype code a GEMM entry point gets for tensors of compute dtype `dtype`"""
    return BF16X3 if (dtype == F32 and GEMM_X3) else dtype
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}


class SedtIgemm(C.Structure):
    _fields_ = [
        ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
        ('A', C.c_void_p), ('B', C.c_void_p),
        ('lda', C.c_int64), ('ldb', C.c_int64),
        ('trans', C.c_int32), ('conv', C.c_int32), ('transposed', C.c_int32),
        ('Hi', C.c_int32), ('Wi', C.c_int32), ('Ci', C.c_int32), ('Ho', C.c_int32), ('Wo', C.c_int32),
        ('KH', C.c_int32), ('KW', C.c_int32), ('sh', C.c_int32), ('sw', C.c_int32), ('ph', C.c_int32),
        ('pw', C.c_int32), ('dh', C.c_int32), ('dw', C.c_int32),
        ('C', C.c_void_p), ('ldc', C.c_int64), ('out_f32', C.c_int32),
        ('scale', C.c_void_p), ('bias', C.c_void_p),
        ('res', C.c_void_p), ('ldr', C.c_int64), ('res_mod', C.c_int32),
        ('mask', C.c_void_p), ('ldm', C.c_int64),
        ('act', C.c_int32), ('act_post_res', C.c_int32),
        ('alpha', C.c_float), ('drop_p', C.c_float), ('seed', C.c_uint32), ('seed_ptr', C.c_void_p),
        ('splitk', C.c_int32), ('slab', C.c_void_p),
        ('tile_m', C.c_int32), ('tile_n', C.c_int32),
        ('colsum_out', C.c_void_p),
        ('bits_out', C.c_void_p), ('ldbits', C.c_int64), ('mask_bits', C.c_int32), ('f32ep', C.c_int32),
        ('omap', C.c_int32), ('o_Hi', C.c_int32), ('o_Wi', C.c_int32), ('o_sh', C.c_int32), ('o_sw', C.c_int32), ('o_h0', C.c_int32),
        ('o_w0', C.c_int32), ('btap_on', C.c_int32), ('btap', C.c_int32 * 8), ('split_out', C.c_void_p), ('awrap', C.c_int32), ('pad2_', C.c_int32), ('bfrag', C.c_void_p),
    ]


class SedtReduceJob(C.Structure):
    _fields_ = [('slab', C.c_void_p), ('rowscale', C.c_void_p), ('out', C.c_void_p), ('colsum_slab', C.c_void_p),
                ('bias_out', C.c_void_p), ('splitk', C.c_int32), ('R', C.c_int32), ('taps', C.c_int32), ('Ci', C.c_int32),
                ('blk0', C.c_int32), ('cs_splitk', C.c_int32)]


CRIT_MAXL = 8


class SedtCriterion(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('logits', 'boxes', 'at', 'tc', 'coef', 'wbox', 'tbox', 'gt_weak', 'tgt_len',
                                           'num_boxes', 'empty_weight', 'dlogits', 'dboxes', 'dboxes2', 'dat', 'out')] + \
               [(n, C.c_int32) for n in ('L', 'B', 'ns', 'Q', 'C', 'n_lab', 'Bat')] + \
               [('layer_of', C.c_int32 * CRIT_MAXL), ('w_ce', C.c_float * CRIT_MAXL), ('w_bbox', C.c_float * CRIT_MAXL),
                ('w_giou', C.c_float * CRIT_MAXL), ('w_weak', C.c_float), ('fl', C.c_int32), ('alpha_fl', C.c_float),
                ('gamma_fl', C.c_float), ('nonfinite', C.c_void_p), ('split', C.c_void_p), ('Qs', C.c_int32), ('q0', C.c_int32),
                ('total', C.c_void_p), ('at_p', C.c_void_p), ('dat_p', C.c_void_p), ('w_weak_p', C.c_float), ('wp_all', C.c_int32),
                ('Bp', C.c_int32)]


class SedtPoolAt(C.Structure):
    _fields_ = [('logits', C.c_void_p), ('boxes', C.c_void_p), ('attn', C.c_void_p)] + \
               [(n, C.c_int32) for n in ('B', 'Qs', 'q0', 'Q', 'C', 'mode')]


POOL_MODES = {'max': 0, 'avg': 1, 'attn': 2, 'weighted_sum': 3}


class SedtMatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('logits', 'boxes', 'lab_cat', 'lab_off', 'box_cat', 'box_off', 'ratio_cat', 'tc',
                                           'coef', 'wbox', 'tbox', 'tidx', 'tgt_len', 'gt_weak', 'assign')] + \
               [(n, C.c_int32) for n in ('L', 'B', 'ns', 'Q', 'C', 'n_lab', 'max_targets')] + \
               [('layer_of', C.c_int32 * CRIT_MAXL), ('w_class', C.c_float), ('w_bbox', C.c_float), ('w_giou', C.c_float),
                ('fl', C.c_int32), ('fine_tune', C.c_int32), ('normalize', C.c_int32), ('alpha_fl', C.c_float),
                ('gamma_fl', C.c_float), ('epsilon', C.c_float), ('alpha', C.c_float), ('ft_rand', C.c_void_p),
                ('ft_seed', C.c_uint32), ('seed_ptr', C.c_void_p), ('split', C.c_void_p), ('Qs', C.c_int32), ('q0', C.c_int32)]


class SedtSplitJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('ld', C.c_int64), ('dst', C.c_void_p), ('rows', C.c_int32), ('cols', C.c_int32),
                ('pattern', C.c_int32), ('blk0', C.c_int32)]


class SedtCopyJob(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('src_stride', C.c_int64), ('dst_stride', C.c_int64), ('outer', C.c_int32),
                ('inner', C.c_int32), ('blk0', C.c_int32), ('pad_', C.c_int32)]


class SedtPrefetch(C.Structure):
    _fields_ = [('ptr', C.c_void_p * 3), ('bytes', C.c_size_t * 3)]


def prefetch_arg(tensors):
    """a SedtPrefetch* for up to three tensors (None entries allowed) the launch AFTER the one given this hint will stream, or None"""
    ts = [t for t in (tensors or ()) if t is not None][:3]
    if not ts:
        return None
    a = SedtPrefetch()
    for i, t in enumerate(ts):
        a.ptr[i], a.bytes[i] = t.data_ptr(), t.numel() * t.element_size()
    return C.byref(a)


MAX_REDUCE_JOBS = 40
_vp, _i, _i64, _f, _u32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_size_t

# name -> (restype, argtypes); mirrors include/sedt_hip.h one to one
SIGNATURES = {
    'sedt_last_error': (C.c_char_p, []),
    'sedt_version': (_i, []),
    'sedt_sizeof': (_i, [_i]),
    'sedt_igemm': (_i, [C.POINTER(SedtIgemm), _i, _vp]),
    'sedt_igemm_group': (_i, [C.POINTER(SedtIgemm), _i, _i, _vp]),
    'sedt_wgrad_group': (_i, [C.POINTER(SedtIgemm), _i, _i, _vp]),
    'sedt_igemm_co': (_i, [C.POINTER(SedtIgemm), C.POINTER(SedtIgemm), _i, _i, _vp, C.POINTER(C.c_int)]),
    'sedt_igemm_splitk': (_i, [_i, _i, _i, _i]),
    'sedt_igemm_describe': (_i, [C.POINTER(SedtIgemm), _i, _i, C.c_char_p, _i]),
    'sedt_igemm_group_describe': (_i, [C.POINTER(SedtIgemm), _i, _i, C.c_char_p, _i]),
    'sedt_wgrad_reduce': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'sedt_wgrad_reduce_bias': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'sedt_split3': (_i, [C.POINTER(SedtSplitJob), _i, _vp]),
    'sedt_copy2d': (_i, [C.POINTER(SedtCopyJob), _i, _vp]),
    'sedt_multi_wgrad_reduce': (_i, [C.POINTER(SedtReduceJob), _i, C.POINTER(SedtPrefetch), _vp]),
    'sedt_skinny_linear_fwd': (_i, [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    'sedt_skinny_linear_bwd_scratch': (_sz, [_i]),
    'sedt_skinny_linear_bwd': (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'sedt_colsum': (_i, [_vp, _i64, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    'sedt_colsum_scratch': (_sz, [_i, _i]),
    'sedt_dropout_grad': (_i, [_vp, _i64, _vp, _i64, _i, _i, _f, _u32, _vp, _i, _vp]),
    'sedt_add': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'sedt_add_n': (_i, [C.POINTER(C.c_void_p), _i, _vp, _i64, _i, _vp]),
    'sedt_cast': (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    'sedt_relu_mask': (_i, [_vp, _vp, _vp, _i64, _i, _vp]),
    'sedt_spsedt_dec_in': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u32, _vp, _i, _vp]),
    'sedt_spsedt_dec_in_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'sedt_gelu_fwd': (_i, [_vp, _vp, _i64, _f, _u32, _vp, _i, _vp]),
    'sedt_gelu_bwd': (_i, [_vp, _vp, _vp, _i64, _f, _u32, _vp, _i, _vp]),
    'sedt_sigmoid_grad': (_i, [_vp, _vp, _vp, _i64, _vp]),
    'sedt_layernorm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'sedt_layernorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _vp]),
    'sedt_layernorm_bwd_scratch': (_sz, [_i, _i]),
    'sedt_layernorm_bwd_drop': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp, _f, _u32, _vp, _i, _vp]),
    'sedt_layernorm_bwd_final': (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    'sedt_attention_fwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _f, _u32,
                                _vp, _i, _vp]),
    'sedt_attention_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp,
                                _i64, _vp, _i64, _i, _i, _i, _i, _f, _u32, _vp, _i, _vp]),
    'sedt_pack_frag': (_i, [_vp, _i, _i, _vp]),
    'sedt_encoder_slab_ok': (_i, [_i, _i, _i, _i, _i]),
    'sedt_encoder_qkv_fwd': (_i, [_vp] * 12 + [_i, _i, C.POINTER(SedtPrefetch), _vp]),
    'sedt_encoder_attn_ffn_fwd': (_i, [_vp] * 20 + [_i, _i, _i, _f, _u32, _u32, _u32, _u32, _vp, _vp]),
    'sedt_bneck_ok': (_i, [_i] * 7),
    'sedt_bneck3_ok': (_i, [_i] * 9),
    'sedt_bneck3_fwd': (_i, [_vp] * 16 + [_i, _i, C.POINTER(SedtPrefetch), _vp]),
    'sedt_bneck3_bwd': (_i, [_vp] * 10 + [_i, _i, C.POINTER(SedtPrefetch), _vp]),
    'sedt_bneck0_ok': (_i, [_i] * 7),
    'sedt_bneck2_ok': (_i, [_i] * 7),
    'sedt_bneck2_fwd': (_i, [_vp] * 17 + [_i, _i, _vp]),
    'sedt_bneck0_fwd': (_i, [_vp] * 19 + [_i, _i, _vp]),
    'sedt_bneck_fwd': (_i, [_vp] * 16 + [_i] * 5 + [_vp]),
    'sedt_bneck_bwd': (_i, [_vp] * 10 + [_i] * 5 + [_vp]),
    'sedt_heads_slab_ok': (_i, [_i, _i, _i, _i]),
    'sedt_heads_fwd': (_i, [_vp] * 16 + [_i] * 5 + [_vp]),
    'sedt_heads_bwd_part_floats': (_sz, [_i] * 5),
    'sedt_heads_bwd': (_i, [_vp] * 17 + [_i] * 5 + [_vp]),
    'sedt_encoder_ffn_bwd': (_i, [_vp] * 15 + [_i, _i, _i, _f, _u32, _u32, _vp, _vp]),
    'sedt_encoder_qkv_bwd': (_i, [_vp] * 10 + [_i, _i, _vp]),
    'sedt_posenc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sedt_mask_resize': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sedt_bn_fold': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'sedt_pack_conv': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    'sedt_stem_prep': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'sedt_stem_im2col': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'sedt_stem_conv0_grad': (_i, [_vp, _vp, _vp, _vp, _vp]),
    'sedt_conv3x3_c64': (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'sedt_stem_pool_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'sedt_stem_pool_wgrad_slabs': (_i, [_i, _i]),
    'sedt_stem_pool_wgrad': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'sedt_maxpool_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sedt_maxpool_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sedt_maxpool_bwd_y': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'sedt_avgpool': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'sedt_sumsq': (_i, [_vp, _i64, _vp, _vp, _sz, _i, _vp]),
    'sedt_sumsq_scratch': (_sz, [_i64]),
    'sedt_multi_bn_fold': (_i, [_vp, _i, _vp]),
    'sedt_multi_pack': (_i, [_vp, _i, _i, _i, _vp]),
    'sedt_multi_gather': (_i, [_vp, _i, _i, _vp]),
    'sedt_multi_ema': (_i, [_vp, _i, _f, _vp, _vp]),
    'sedt_multi_sumsq': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'sedt_multi_adamw': (_i, [_vp, _i, _vp, _f, _f, _f, _f, _vp, _vp, _vp]),
    'sedt_set_criterion_scratch': (_sz, [_i, _i, _i]),
    'sedt_set_criterion': (_i, [C.POINTER(SedtCriterion), _vp, _vp]),
    'sedt_set_criterion_bwd': (_i, [C.POINTER(SedtCriterion), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'sedt_match_targets': (_i, [C.POINTER(SedtMatch), _vp]),
    'sedt_feature_loss': (_i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'sedt_scale_layers': (_i, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int32), _i, _i64, _vp]),
    'sedt_sum_f32': (_i, [_vp, _i, _vp, _vp]),
    'sedt_pool_at': (_i, [C.POINTER(SedtPoolAt), _vp, _vp]),
    'sedt_pool_at_bwd': (_i, [C.POINTER(SedtPoolAt), _vp, _vp, _vp, _vp, _vp]),
    'sedt_box_transform': (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    'sedt_mixup': (_i, [_vp, _vp, _vp, _i, _i64, _vp, _vp]),
    'sedt_mixup_targets': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'sedt_query_patches': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    'sedt_postprocess': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    'sedt_pseudo_labels': (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'sedt_hungarian_batch': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'sedt_adamw_clip': (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _i, _vp]),
}

_lib = None


def load():
    """Load the HIP library; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no CPU/PyTorch fallback for the SEDT hot path)')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


LAUNCH_LOG = None     # a collections.Counter while a launch_log() scope is open: entry-point name -> launches that returned 0


class launch_log(object):
    """count the C-ABI launches issued inside the scope, by entry point (the name every ops.* wrapper passes to check()): the parity
    tests assert with it WHICH kernel family a model-level run dispatched (tests/test_headline_parity_gpu.py), so a fill rule that
    silently sends a test batch down another path cannot make a parity claim about kernels that never ran.
    ``with lib.launch_log() as log: ...; log['encoder_qkv_fwd']``"""

    def __enter__(self):
        import collections
        global LAUNCH_LOG
        self.prev, LAUNCH_LOG = LAUNCH_LOG, collections.Counter()
        return LAUNCH_LOG

    def __exit__(self, *exc):
        global LAUNCH_LOG
        LAUNCH_LOG = self.prev
        return False


def check(status, what=''):
    if status != 0:
        raise RuntimeError(f'{what}: {load().sedt_last_error().decode()}')
    if LAUNCH_LOG is not None:
        LAUNCH_LOG[what] += 1


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    """device pointer of a tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())
