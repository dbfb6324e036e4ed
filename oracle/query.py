"""torch-CPU restatement of the open-vocabulary query (test oracle).

Follows ``run/evaluate.py:288-292`` (distill / fusion modes):
    pred = feats[inds_reverse].half() @ text_features.t();  argmax over labels
and ``run/evaluate.py:302-324`` (ensemble mode).  ``text`` is the L2-normalised
fp16 CLIP text matrix [C, D] (``util/util.py:41-44``).  The half GEMM is
evaluated the way cuBLAS/hipBLASLt do it: fp16 inputs, fp32 accumulation, one
rounding to fp16 at the end.
"""
import torch


def half_matmul(x_half, text_half):
    return (x_half.float() @ text_half.float().t()).half()


def query(feats, text_half, gather=None):
    """(scores fp16 [N_pts, C], argmax int64 [N_pts])."""
    x = feats if gather is None else feats[gather]
    scores = half_matmul(x.half(), text_half)
    return scores, torch.max(scores, 1)[1]


def query_ensemble(feat_distill, feat_fusion, text_half, gather=None):
    """run/evaluate.py:302-324; returns (scores, argmax, selected fp16 features)."""
    fd = feat_distill if gather is None else feat_distill[gather]
    ff = feat_fusion if gather is None else feat_fusion[gather]
    pf = half_matmul((ff / (ff.norm(dim=-1, keepdim=True) + 1e-5)).half(), text_half)
    pd = half_matmul((fd / (fd.norm(dim=-1, keepdim=True) + 1e-5)).half(), text_half)
    ens = fd.clone().half()
    m = pd.max(dim=-1)[0] < pf.max(dim=-1)[0]
    ens[m] = ff[m].half()
    scores = half_matmul(ens, text_half)
    return scores, torch.max(scores, 1)[1], ens
