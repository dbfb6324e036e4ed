"""Open-vocabulary query: per-point feature x CLIP-text similarity + argmax.

Host-side mirror of ``run/evaluate.py:283-324`` (same three modes) and of the
validation query ``run/distill.py:423-425``; each call is ONE fused HIP launch
(gather + fp16 cast + MFMA GEMM + fp16 rounding + argmax) instead of the
reference's gather -> half -> matmul -> max chain.  ``text_features`` is the
L2-normalised fp16 CLIP matrix [C, D] of ``util/util.py:41-44``.
"""
from . import ops


def query_distill(predictions, text_features, inds_reverse=None, return_scores=False):
    """run/evaluate.py:288-292:  pred = predictions[inds_reverse].half() @ text.t();  max(pred, 1)[1].
    predictions: float32 [N_vox, D] (the network output); returns int64 labels [N_pts]
    (and the fp16 score matrix if return_scores)."""
    scores, labels = ops.cosine_query(predictions, text_features, inds_reverse, want_scores=return_scores)
    return (labels, scores) if return_scores else labels


def query_fusion(feat_3d, text_features, inds_reverse=None, mask=None, unknown_label=None, return_scores=False):
    """run/evaluate.py:293-300 (fused 2-D features instead of the network output); points without a
    fused feature (``mask`` False) get ``unknown_label`` when mark_no_feature_to_unknown is on."""
    scores, labels = ops.cosine_query(feat_3d, text_features, inds_reverse, want_scores=return_scores)
    if mask is not None and unknown_label is not None:
        m = mask if inds_reverse is None else mask[inds_reverse]
        labels[~m.to(labels.device)] = unknown_label
    return (labels, scores) if return_scores else labels


def query_ensemble(predictions, feat_3d, text_features, inds_reverse=None, return_scores=False):
    """run/evaluate.py:302-324: pick per point the source (distilled 3-D vs fused 2-D) whose best
    cosine score is larger, then score the selected un-normalised fp16 feature.
    Returns (labels, used_fusion bool [N_pts][, scores])."""
    scores, labels, sel = ops.query_ensemble(predictions, feat_3d, text_features, inds_reverse, inds_reverse,
                                             want_scores=return_scores)
    return (labels, sel, scores) if return_scores else (labels, sel)
