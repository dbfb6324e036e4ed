"""ORACLE (test infrastructure, never imported by the product): CPU restatement of the distillation loss of
run/distill.py:322-328 in float64 numpy, with its closed-form gradient with respect to the FULL network output.

    output_3d = output_3d[mask]
    loss = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()      # 'cosine'
    loss = torch.nn.L1Loss()(output_3d, feat_3d)                             # 'l1'

torch.nn.CosineSimilarity(dim=1, eps=1e-8) divides each operand by its norm clamped from below by eps and sums the
products.  Pinned in tests/test_host_logic.py against torch's own forward AND autograd gradient of exactly those lines."""
import numpy as np

EPS = 1e-8


def distill_loss(out, sel, target, loss_type="cosine"):
    """-> (loss, d loss / d out [N, D]) in float64; sel int rows of `out`, target [len(sel), D]."""
    out = np.asarray(out, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    sel = np.asarray(sel, dtype=np.int64)
    a = out[sel]
    n_sel, d = a.shape
    g = np.zeros_like(out)
    if loss_type == "cosine":
        na = np.maximum(np.linalg.norm(a, axis=1), EPS)
        nb = np.maximum(np.linalg.norm(target, axis=1), EPS)
        dot = (a * target).sum(1)
        cos = dot / (na * nb)
        loss = (1.0 - cos).mean()
        live = (np.linalg.norm(a, axis=1) > EPS)[:, None]
        ga = target / (na * nb)[:, None] - np.where(live, a * (dot / (na ** 3 * nb))[:, None], 0.0)
        g[sel] = -ga / n_sel
    elif loss_type == "l1":
        loss = np.abs(a - target).mean()
        g[sel] = np.sign(a - target) / (n_sel * d)
    else:
        raise ValueError(loss_type)
    return loss, g
