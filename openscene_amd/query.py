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


def head_times_text(final_kernel, text_features):
    """M = W_final @ text^T  (float32 [C_in, n_labels]): the 1x1 head of the network (models/mink_unet.py:108-113)
    folded into the CLIP text matrix.  96 x 768 x n_labels multiply-adds -- done once per label set."""
    return final_kernel.detach().float() @ text_features.float().t()


def query_distill_fused(features, final_kernel, text_features, inds_reverse=None, return_scores=False):
    """the distill-mode query with the head folded in,
        labels = argmax_c  features[inds_reverse] @ (W_final @ text^T)
    instead of  (features @ W_final)[inds_reverse].half() @ text^T  (models/mink_unet.py:174 + run/evaluate.py:288-292):
    the [N, 768] feature matrix (310 MB per 100 k voxels) is neither written nor re-read.  `features` = the float32
    [N_vox, 96] input of the final conv (``MinkUNetBase.forward_features``).  The contraction runs on the split-bf16
    convolution kernel (K = 1), i.e. at fp32 accuracy: the scores differ from the reference's by the reference's OWN
    fp16 rounding of the 768-d vector (<= 2e-3 absolute on O(1) scores, the same bound as the unfused path), and labels
    agree wherever the reference's top-2 margin exceeds twice that.  Returns int64 labels [N_pts] (and float32 scores
    [N_vox, n_labels] per VOXEL if return_scores)."""
    scores = _rows_times(features, head_times_text(final_kernel, text_features))
    labels = ops.rows_argmax(scores, inds_reverse)          # argmax + point -> voxel gather, one launch
    return (labels, scores) if return_scores else labels


def _rows_times(features, M):
    """features [N, c_in] (float32, device) @ M [c_in, c] at fp32 accuracy on the split-bf16 1x1-convolution kernel (csrc/dense.hip)."""
    import torch
    c = M.shape[1]
    cp = (c + 3) // 4 * 4
    if cp != c:
        M = torch.cat([M, M.new_full((M.shape[0], cp - c), 0.0)], 1)
    n = features.shape[0]
    if ops.dense_eligible(M.shape[0], cp):
        wf, _ = ops.weight_prep_tl(M.contiguous(), want_dgrad=False)
        out = ops.dense_fwd(features, wf, cp)
    else:
        out = ops.spconv_fwd(features, M.contiguous(), None, n)
    return out[:, :c]


def query_ensemble_fused(features, final_kernel, feat_3d, text_features, inds_reverse=None, return_scores=False):
    """SURVEY.md 8(f) row 2, ensemble variant (run/evaluate.py:302-324 with the head folded in): the distilled source is
    never expanded to 768-d.  With W = the final 1x1 kernel [96, D], M = W text^T [96, C] and the Gram matrix
    G = W W^T [96, 96]:
        scores_d = F M,   |F W| = sqrt(rowsum((F G) * F))      (the 768-d norm from 96-d quantities)
        best_d   = max_c scores_d / (|F W| + 1e-5),   best_f = max_c (x_f.half() text^T) / (|x_f| + 1e-5)
    a point takes the fused 2-D feature iff best_d < best_f (the reference's rule), then is scored with the selected
    UN-normalised feature.  `features`: float32 [N_vox, 96] (``forward_features``), `feat_3d`: the fused features per
    POINT [N_pts, D], `inds_reverse`: point -> voxel.  Differences to the reference expression: it rounds the normalised
    768-d vectors and their scores to fp16 before comparing, here the distilled side stays fp32 -- the selection agrees
    wherever |best_d - best_f| exceeds that rounding (4e-3), scores of distill-selected points differ by the fp16 rounding
    of the 768-d vector (as in query_distill_fused).  Returns (labels [N_pts], used_fusion bool [N_pts][, float32 scores])."""
    import torch
    W = final_kernel.detach().float()
    M = head_times_text(final_kernel, text_features)
    sd = _rows_times(features, M)                                            # [N_vox, C]
    nd = (_rows_times(features, (W @ W.t()).contiguous()) * features).sum(1).clamp_min(0).sqrt()
    best_d = sd.max(1)[0] / (nd + 1e-5)
    sf, _ = ops.cosine_query(feat_3d, text_features, None, want_scores=True)  # [N_pts, C] fp16, un-normalised fused features
    sf = sf.float()
    best_f = sf.max(1)[0] / (feat_3d.float().norm(dim=1) + 1e-5)
    if inds_reverse is not None:
        sd, best_d = sd[inds_reverse], best_d[inds_reverse]
    use_f = best_d < best_f
    scores = torch.where(use_f[:, None], sf, sd)
    labels = scores.argmax(1)
    return (labels, use_f, scores) if return_scores else (labels, use_f)
