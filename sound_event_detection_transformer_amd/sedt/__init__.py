"""Drop-in counterpart of the reference ``sedt`` package (sedt/__init__.py:1-63) on the MI355X HIP path."""
from .spsedt import SPSEDT
from .sedt import SEDT, SetCriterion, PostProcess, MLP, TargetTables
from .backbone import build_backbone
from .transformer import build_transformer, Transformer, TransformerDecoder, TransformerDecoderLayer
from .matcher import build_matcher, HungarianMatcher


def build_model(args):
    """same contract as the reference: (model, criterion, postprocessors) from the train_sedt.py argparse namespace"""
    num_classes = 1 if args.self_sup else args.num_classes
    backbone = build_backbone(args)
    transformer = build_transformer(args)
    if args.self_sup:
        model = SPSEDT(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries, aux_loss=args.aux_loss,
                       feature_recon=args.feature_recon, query_shuffle=args.query_shuffle, num_patches=args.num_patches)
    else:
        model = SEDT(backbone, transformer, num_classes=num_classes, num_queries=args.num_queries, aux_loss=args.aux_loss,
                     dec_at=args.dec_at, pooling=args.pooling)
    matcher = build_matcher(args)
    weight_dict = {'loss_ce': args.ce_loss_coef, 'loss_bbox': args.bbox_loss_coef, 'loss_giou': args.giou_loss_coef}
    losses = ['labels', 'boxes', 'cardinality']
    if not args.self_sup:
        if args.dec_at:
            weight_dict['loss_weak'] = args.weak_loss_coef
            losses += ['weak']
        if args.pooling:
            weight_dict['loss_weak_p'] = args.weak_loss_p_coef
    elif args.feature_recon:
        losses += ['feature']
        weight_dict['loss_feature'] = 1
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f'_{i}': v for k, v in weight_dict.items()})
        weight_dict.update(aux)
    criterion = SetCriterion(num_classes, matcher=matcher, weight_dict=weight_dict, eos_coef=args.eos_coef, losses=losses)
    postprocessors = {'bbox': PostProcess()}
    return model, criterion, postprocessors


def default_args(**over):
    """the model-relevant defaults of reference train_sedt.py:28-129 as a namespace (URBAN-SED recipe)"""
    import argparse
    a = argparse.Namespace(
        num_classes=10, lr_backbone=1e-4, backbone='resnet50', dilation=True, position_embedding='sine', enc_layers=3,
        dec_layers=3, dim_feedforward=2048, hidden_dim=256, dropout=0.1, nheads=8, num_queries=10, pre_norm=True,
        aux_loss=True, dec_at=True, pooling=None, self_sup=False, set_cost_class=1, set_cost_bbox=5, set_cost_giou=2,
        epsilon=1, alpha=1, ce_loss_coef=1, bbox_loss_coef=5, giou_loss_coef=2, eos_coef=0.1, weak_loss_coef=1,
        weak_loss_p_coef=1, feature_recon=True, query_shuffle=False, num_patches=10)
    for k, v in over.items():
        setattr(a, k, v)
    return a
