"""The reference's query lines, unchanged (run/evaluate.py:289-330, run/distill.py:421-425), on the lazy row gather of
openscene_amd/lazy_rows.py: every expression the reference writes gives what plain torch gives (CPU: the fall-back paths, bit for bit;
GPU: the fused kernel's scores = osn_cosine_query's, labels = torch's wherever the top-2 margin is clear)."""
import pytest
import torch

from openscene_amd import lazy_rows
from openscene_amd.lazy_rows import GatheredRows, NetworkOutput, wrap_output


def _case(dev, n_vox=300, n_pts=1000, d=64, c=20, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = torch.randn(n_vox, d, generator=g).to(dev)
    inds = torch.randint(0, n_vox, (n_pts,), generator=g).to(dev)
    text = torch.nn.functional.normalize(torch.randn(c, d, generator=g), dim=1).half().to(dev)
    return out, inds, text


def test_reference_expressions_on_cpu_equal_plain_torch():
    out, inds, text = _case("cpu")
    with torch.no_grad():
        predictions = wrap_output(out.clone())
        assert type(predictions) is NetworkOutput
        predictions = predictions[inds, :]                                   # run/evaluate.py:290
        assert type(predictions) is GatheredRows and predictions.shape == (1000, 64) and predictions.dtype == torch.float32
        assert len(predictions) == 1000 and predictions.size(1) == 64 and predictions.device.type == "cpu"
        h = predictions.half()
        assert type(h) is GatheredRows and h.dtype == torch.float16 and h._real is None
        pred = h @ text.t()                                                  # :291 (no fused kernel on the CPU: torch's own matmul)
        ref = out[inds, :].half() @ text.t()
        assert type(pred) is torch.Tensor and torch.equal(pred, ref)
        assert torch.equal(torch.max(pred, 1)[1], torch.max(ref, 1)[1])      # :292
        # the ensemble branch (:307-330): norm, division, clone, masked assignment, .cpu().numpy()
        pd = (predictions / (predictions.norm(dim=-1, keepdim=True) + 1e-5)).half() @ text.t()
        rows = out[inds, :]
        assert torch.equal(pd, (rows / (rows.norm(dim=-1, keepdim=True) + 1e-5)).half() @ text.t())
        fe = predictions.clone().half()
        m = torch.arange(1000) % 3 == 0
        fe[m] = fe.flip(0)[m]
        rf = rows.clone().half()
        rf[m] = rf.flip(0)[m]
        assert torch.equal(fe, rf)
        assert (predictions.cpu().numpy() == rows.numpy()).all()
        assert predictions._real is not None                                 # gathered once, cached
        assert "tensor(" in repr(predictions)
        # copies and serialisation see plain tensors (run/evaluate.py:330 saves features with np.save; a user may torch.save them)
        import copy, io, pickle
        o = wrap_output(out.clone())
        lazy = o[inds]
        for obj, ref_t in ((o, out), (lazy, out[inds])):
            assert type(copy.deepcopy(obj)) is torch.Tensor and torch.equal(copy.deepcopy(obj), ref_t)
            assert type(pickle.loads(pickle.dumps(obj))) is torch.Tensor
            buf = io.BytesIO()
            torch.save(obj, buf)
            buf.seek(0)
            back = torch.load(buf)
            assert type(back) is torch.Tensor and torch.equal(back, ref_t)
        assert torch.equal(lazy.to(torch.float16), out[inds].half()) and lazy.T.shape == (64, 1000) and lazy[0].shape == (64,)


def test_only_plain_row_gathers_are_lazy_and_nothing_under_autograd():
    out, inds, text = _case("cpu", seed=1)
    with torch.no_grad():
        o = wrap_output(out.clone())
        assert type(o[inds]) is GatheredRows                                 # run/distill.py:422 writes output[inds_reverse, :]; [inds] is the same gather
        assert type(o[3]) is torch.Tensor and type(o[:, :8]) is torch.Tensor and type(o[inds, 2]) is torch.Tensor
        assert type(o[inds.int()]) is torch.Tensor                           # not an int64 vector: torch's path
        mask = torch.arange(300) % 2 == 0
        assert type(o[mask]) is torch.Tensor and torch.equal(o[mask], out[mask])      # run/distill.py:322 (boolean mask)
        assert type(o * 2) is torch.Tensor and type(o.sum()) is torch.Tensor
        assert wrap_output(out.half()) .__class__ is torch.Tensor and wrap_output(out[0]).__class__ is torch.Tensor
    w = out.clone().requires_grad_(True)
    assert type(wrap_output(w * 1.0)) is torch.Tensor                        # autograd on: left alone
    with torch.no_grad():
        assert type(wrap_output(w)) is torch.Tensor                          # a tensor that requires grad: left alone
    saved = lazy_rows.ENABLED
    try:
        lazy_rows.ENABLED = False
        with torch.no_grad():
            assert type(wrap_output(out.clone())) is torch.Tensor
    finally:
        lazy_rows.ENABLED = saved


@pytest.mark.gpu
def test_unchanged_call_site_runs_the_fused_query_kernel():
    from openscene_amd.query import query_distill
    dev = torch.device("cuda", 0)
    for (n_vox, n_pts, d, c) in ((5000, 20000, 768, 20), (4000, 9001, 512, 160), (300, 1000, 64, 20)):
        out, inds, text = _case(dev, n_vox, n_pts, d, c, seed=n_pts)
        with torch.no_grad():
            predictions = wrap_output(out.clone())
            predictions = predictions[inds, :]
            pred = predictions.half() @ text.t()
            logits = torch.max(pred, 1)[1]
            assert type(pred) is torch.Tensor and pred.dtype == torch.float16 and pred.shape == (n_pts, c)
            assert predictions._real is None                                 # the [n_pts, d] matrix was never gathered
            labels, scores = query_distill(out, text, inds, return_scores=True)
            assert torch.equal(pred, scores) and torch.equal(logits, labels)          # torch.max and the kernel's argmax: first maximum
            ref = out[inds, :].half() @ text.t()                             # torch's chain (rocBLAS): fp16 rounding of an fp32-accumulated sum
            assert (pred.float() - ref.float()).abs().max().item() <= 4e-3 * max(1.0, ref.float().abs().max().item())
            top2 = ref.float().topk(2, dim=1)[0]
            clear = (top2[:, 0] - top2[:, 1]) > 8e-3
            assert torch.equal(logits[clear], torch.max(ref, 1)[1][clear])
            # and the fall-back on the device: the same rows as torch's gather
            assert torch.equal(predictions.clone(), out[inds, :])
