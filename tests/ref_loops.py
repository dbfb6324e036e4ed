"""TEST INFRASTRUCTURE: the reference's own training and evaluation LOOPS, executed unmodified.

`run/distill.py` and `run/evaluate.py` of /root/reference are imported as they are and driven through their own `main()`
(argument parser -> yaml config -> `main_worker` -> data loaders -> `distill()` / `validate()` / `evaluate()` ->
`save_checkpoint`), on two small synthetic scenes written in the reference's on-disk formats by `openscene_amd.io`,
with `MinkowskiEngine` resolved to `openscene_amd.minkowski` through the package's alias.  Nothing in those files is
edited; what the harness supplies around them:

  * stand-ins for packages this image does not have and the path does not need (`tensorboardX`, `clip`, `SharedArray`,
    `open3d`, `cv2`): the scalar writer and the point-cloud exporter RECORD what the loops hand them, which is how the
    losses, learning rates, validation metrics and predicted labels are read out;
  * the CLIP text embeddings as a pre-saved file (run/distill.py:265-268 loads it first) / as the return value of
    `extract_text_feature` (run/evaluate.py:101): seeded unit vectors -- CLIP itself is outside SURVEY.md section 8;
  * deterministic weights (tests/golden/unet_recipe.parameter) loaded into the model `get_model` built, so that a
    replay without the reference tree (the GPU box) starts from the same network;
  * recorders around `SparseTensor` and `metric.evaluate` (module attributes, not file edits) that keep the batches the
    loaders produced and the predictions the loops scored;
  * without a GPU: `Tensor.cuda` / `Module.cuda` / storage `.cuda` are the identity and `tests/cpu_backend.py` stands in
    for the HIP operators (the product package has no CPU path).

`run_reference()` returns everything a replay needs (`tests/golden/make_golden_loops.py` stores it as
tests/golden/ref_loops.npz); `replay()` is that replay through openscene_amd's own classes -- on the CPU backend here,
through the HIP library on the GPU box -- and `compare()` states the tolerances.
"""
import collections
import collections.abc
import contextlib
import os
import random
import sys
import types

import numpy as np
import torch

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import unet_recipe as R  # noqa: E402

D_TEXT = 768                      # feature_2d_extractor: openseg (models/disnet.py:30-31)
N_LABELS = 20                     # SCANNET_LABELS_20
VOXEL = 0.05
ARCH = "MinkUNet18A"
BASE_LR = 1e-4
SEED = 1463                       # config/scannet/ours_openseg.yaml manual_seed


def _rng(*key):
    return R._rng("ref_loops", *key)


def text_features():
    """[20, 768] unit rows, fp16 (util/util.py:43-44 normalises; run/distill.py:268 loads a saved tensor)."""
    t = torch.from_numpy(_rng("text").standard_normal((N_LABELS, D_TEXT))).float()
    return torch.nn.functional.normalize(t, dim=1).half()


def codebook():
    """The 2-D features a scene's points carry: one of 20 prototype vectors per point (pixel features of one class look
    alike), column 0 = code + 1 exactly (so a recorded feature row can be stored as one byte: 0 = the zero row)."""
    cb = _rng("codebook").standard_normal((N_LABELS, D_TEXT)).astype(np.float32)
    cb[:, 0] = np.arange(1, N_LABELS + 1)
    return torch.from_numpy(cb).half()


def scenes():
    """Two room-like clouds: (name, xyz float64 [N, 3], colours in [-1, 1], labels float64 with -100 = ignore, mask_full, code)."""
    out = []
    for si, n in enumerate((4200, 3300)):
        r = _rng("scene", si)
        floor = np.stack([r.random(n // 2) * 2.4, r.random(n // 2) * 1.8, r.normal(0, 0.004, n // 2)], 1)
        wall = np.stack([r.random(n // 4) * 2.4, np.full(n // 4, 1.8) + r.normal(0, 0.004, n // 4), r.random(n // 4) * 1.2], 1)
        m = n - n // 2 - n // 4
        box = np.stack([0.8 + r.random(m) * 0.6, 0.5 + r.random(m) * 0.5, 0.4 + r.normal(0, 0.004, m)], 1)
        xyz = np.concatenate([floor, wall, box])[r.permutation(n)]
        colors = r.random((n, 3)) * 2 - 1
        # labels follow position (so that IoU is not chance): 3 coarse regions x a few classes
        labels = (np.floor(xyz[:, 0] / 0.4) + 6 * (xyz[:, 2] > 0.2) + 3 * (xyz[:, 1] > 1.7)).astype(np.int64) % N_LABELS
        labels = labels.astype(np.float64)
        labels[r.random(n) < 0.04] = -100
        mask_full = r.random(n) < 0.6
        code = np.where(labels >= 0, labels, 0).astype(np.int64)
        flip = r.random(n) < 0.15                                  # 2-D predictions are not perfect
        code[flip] = r.integers(0, N_LABELS, int(flip.sum()))
        out.append(("scene%04d_00" % si, xyz, colors, labels, mask_full, code))
    return out


def write_dataset(root):
    """The reference's directory layout (dataset/point_loader.py:82, dataset/feature_loader.py:42-47) through openscene_amd.io."""
    from openscene_amd import io as oio
    data = os.path.join(root, "scannet_3d")
    feat = os.path.join(root, "scannet_multiview_openseg")
    for split in ("train", "val"):
        os.makedirs(os.path.join(data, split))
    os.makedirs(feat)
    cb = codebook()
    for name, xyz, colors, labels, mask_full, code in scenes():
        for split in ("train", "val"):
            oio.save_scene(os.path.join(data, split, name + "_vh_clean_2.pth"), xyz, colors, labels.copy())
        oio.save_fused_features(os.path.join(feat, name + "_0.pt"), cb[torch.from_numpy(code[mask_full])],
                                torch.from_numpy(mask_full))
    os.makedirs(os.path.join(root, "saved_text_embeddings"))
    torch.save(text_features(), os.path.join(root, "saved_text_embeddings", "clip_scannet_labels_768.pt"))
    return data, feat


CONFIG = """DATA:
  data_root: {data}
  data_root_2d_fused_feature: {feat}
  feature_2d_extractor: openseg
  classes: 20
  aug: True
  voxel_size: {voxel}
  input_color: False
  use_shm: False

DISTILL:
  arch_3d: {arch}
  ignore_label: 255
  train_gpu: [0]
  workers: 0
  batch_size: 2
  batch_size_val: 1
  base_lr: {lr}
  loss_type: cosine
  loop: 2
  epochs: 1
  start_epoch: 0
  power: 0.9
  momentum: 0.9
  manual_seed: {seed}
  print_freq: 1
  save_freq: 1
  save_path: {save}
  resume:
  evaluate: True
  eval_freq: 1

TEST:
  split: val
  prompt_eng: True
  mark_no_feature_to_unknown: True
  feature_type: '{feature_type}'
  save_feature_as_numpy: False
  vis_input: False
  vis_pred: False
  vis_gt: False
  test_workers: 0
  test_gpu: [0]
  test_batch_size: 1
  test_repeats: 2
  model_path: {model_path}
  save_folder: {save}/eval_{feature_type}

Distributed:
  dist_url: tcp://127.0.0.1:6787
  dist_backend: 'nccl'
  multiprocessing_distributed: True
  world_size: 1
  rank: 0
"""


class Recorder:
    def __init__(self):
        self.scalars = []            # (tag, value, step) from SummaryWriter.add_scalar
        self.clouds = []             # (file name, points, colors) from export_pointcloud -> open3d
        self.tensors = []            # (where, feats, coords) from SparseTensor(...)
        self.evals = []              # (pred, gt, result) from metric.evaluate
        self.batches = []            # (where, tuple of tensors) from the loaders' collate functions


def _stub_modules(rec):
    """Modules the reference imports at file scope and this image lacks."""
    mods = {}
    tb = types.ModuleType("tensorboardX")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, tag, value, step=None):
            rec.scalars.append((tag, float(value), step))

        def close(self):
            pass
    tb.SummaryWriter = SummaryWriter
    mods["tensorboardX"] = tb
    mods["clip"] = types.ModuleType("clip")                    # util/util.py:11 (only extract_clip_feature uses it)
    mods["SharedArray"] = types.ModuleType("SharedArray")      # shared-memory cache: off (use_shm False)
    mods["cv2"] = types.ModuleType("cv2")
    o3d = types.ModuleType("open3d")

    class _PC:
        pass
    o3d.geometry = types.SimpleNamespace(PointCloud=_PC, TriangleMesh=_PC)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a), Vector3iVector=lambda a: np.asarray(a))
    o3d.io = types.SimpleNamespace(write_point_cloud=lambda name, pcd: rec.clouds.append(
        (os.path.basename(name), np.asarray(pcd.points), np.asarray(getattr(pcd, "colors", None)))),
        write_triangle_mesh=lambda *a: None)
    mods["open3d"] = o3d
    return mods


@contextlib.contextmanager
def reference_environment(root, rec, monkeypatch, have_gpu):
    """Everything the unmodified files need around them; undone on exit (monkeypatch)."""
    import openscene_amd
    collections.Sequence = collections.abc.Sequence            # dataset/voxelization_utils.py:6, voxelizer.py:55 (Python < 3.10)
    collections.Iterable = collections.abc.Iterable
    for name in [m for m in sys.modules if m.split(".")[0] in ("MinkowskiEngine", "models", "run", "util", "dataset")]:
        monkeypatch.delitem(sys.modules, name)
    for name, mod in _stub_modules(rec).items():
        if name not in sys.modules:
            monkeypatch.setitem(sys.modules, name, mod)
    openscene_amd.install_minkowski_alias()
    monkeypatch.syspath_prepend(REFERENCE)
    real_load = torch.load
    monkeypatch.setattr(torch, "load", lambda *a, **k: real_load(*a, **dict({"weights_only": False}, **k)))   # torch 1.x default
    if not have_gpu:
        import cpu_backend
        cpu_backend.install(monkeypatch)
        monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
        monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
        monkeypatch.setattr(torch.UntypedStorage, "cuda", lambda self, *a, **k: self)
        monkeypatch.setattr(torch.storage.TypedStorage, "cuda", lambda self, *a, **k: self)
    monkeypatch.chdir(root)
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", ""))   # (run/distill.py:89 overwrites it)
    try:
        yield
    finally:
        for name in [m for m in sys.modules if m.split(".")[0] in ("MinkowskiEngine", "models", "run", "util", "dataset")]:
            sys.modules.pop(name, None)


def load_recipe_weights(model):
    sd = model.state_dict()
    with torch.no_grad():
        for name, t in sd.items():
            v = R.parameter(name.replace("net3d.", "").replace("module.", ""), tuple(t.shape))
            if v is not None:
                t.copy_(torch.from_numpy(v).to(t.dtype))
    return model


def _instrument(mod, rec, where):
    """Recorders as module attributes of an imported reference file (its source is untouched)."""
    real_st = mod.SparseTensor

    def recording_sparse_tensor(feats, coords, *a, **k):
        rec.tensors.append((where, feats.detach().cpu().clone(), coords.detach().cpu().clone()))
        return real_st(feats, coords, *a, **k)
    mod.SparseTensor = recording_sparse_tensor
    real_get = mod.get_model
    mod.get_model = lambda cfg: load_recipe_weights(real_get(cfg))


def _record_collate(owner, name, rec, where):
    """Wrap a collate function (a module attribute the reference resolves when it builds its DataLoader): keep a copy of every
    batch it returns -- the loops shift `coords` in place afterwards (run/distill.py:316)."""
    real = getattr(owner, name)
    if getattr(real, "_osn_recorder", False):
        real = real._osn_real

    def collate(batch):
        res = real(batch)
        rec.batches.append((where, tuple(t.clone() for t in res)))
        return res
    collate._osn_recorder, collate._osn_real = True, real
    setattr(owner, name, collate)


def run_reference(root, monkeypatch, have_gpu=False):
    """main() of run/distill.py (1 epoch: 2 iterations, validation, checkpoint), then main() of run/evaluate.py in the
    'distill' and 'ensemble' modes on that checkpoint.  -> dict of numpy arrays (the golden)."""
    rec = Recorder()
    data, feat = write_dataset(root)
    save = os.path.join(root, "exp")
    out = {}
    with reference_environment(root, rec, monkeypatch, have_gpu):
        def cfg(feature_type, model_path=""):
            p = os.path.join(root, "cfg_%s.yaml" % feature_type)
            with open(p, "w") as f:
                f.write(CONFIG.format(data=data, feat=feat, voxel=VOXEL, arch=ARCH, lr=BASE_LR, seed=SEED, save=save,
                                      feature_type=feature_type, model_path=model_path))
            return p
        import run.distill as distill_mod                      # noqa: E402  (the reference's file, unmodified)
        _instrument(distill_mod, rec, "distill")
        _record_collate(distill_mod, "collation_fn", rec, "train")
        _record_collate(distill_mod, "collation_fn_eval_all", rec, "val")
        monkeypatch.setattr(sys, "argv", ["distill.py", "--config", cfg("distill")])
        distill_mod.main()
        ckpt = os.path.join(save, "model", "model_last.pth.tar")
        assert os.path.isfile(ckpt), "run/distill.py did not write its checkpoint"
        # ---- what the training loop saw and produced
        train = [t for t in rec.tensors if t[0] == "distill"]
        sc = collections.defaultdict(list)
        for tag, v, step in rec.scalars:
            sc[tag].append(v)
        out["train_loss_batch"] = np.asarray(sc["loss_train_batch"])
        out["train_lr"] = np.asarray(sc["learning_rate"])
        out["train_loss_epoch"] = np.asarray(sc["loss_train"])
        out["val"] = np.asarray([sc["loss_val"][0], sc["mIoU_val"][0], sc["mAcc_val"][0], sc["allAcc_val"][0]])
        tb = [b for w, b in rec.batches if w == "train"]
        vb = [b for w, b in rec.batches if w == "val"]
        n_train = len(tb)
        assert n_train == len(out["train_loss_batch"]) == 2 and len(train) == n_train + len(vb)
        cb0 = codebook()[:, 0].float()
        for i, ((_, feats, coords), (c0, f0, lab, feat_3d, mask)) in enumerate(zip(train[:n_train], tb)):
            assert torch.all(feats == 1), "input_color False: all-ones features (feature_loader.py:184)"
            shift = (coords[:, 1:4] - c0[:, 1:4]).unique(dim=0)
            assert shift.shape[0] == 1 and torch.equal(coords[:, 0], c0[:, 0]), "run/distill.py:316 shifts a batch by ONE lattice vector"
            out["train%d_coords" % i] = coords.numpy().astype(np.int32)          # as the network saw them (shifted)
            out["train%d_mask" % i] = mask.numpy()
            out["train%d_labels" % i] = lab.numpy().astype(np.uint8)
            code = feat_3d[:, 0].float().round().to(torch.int64)                 # column 0 of a codebook row = code + 1
            assert torch.equal(codebook()[code - 1], feat_3d), "feat_3d rows are codebook rows"
            out["train%d_code" % i] = code.numpy().astype(np.uint8)
        for i, ((_, feats, coords), (c0, f0, lab, inds)) in enumerate(zip(train[n_train:], vb)):
            assert torch.equal(coords, c0)
            out["val%d_coords" % i] = coords.numpy().astype(np.int32)
            out["val%d_labels" % i] = lab.numpy().astype(np.uint8)
            out["val%d_inds" % i] = inds.numpy().astype(np.int32)
        out["n_val"] = np.asarray(len(vb))
        del cb0
        out["pred_cloud_colors"] = next(c for n, p, c in rec.clouds if n.startswith("pred_"))
        sd = torch.load(ckpt, map_location="cpu")
        out["ckpt_epoch"] = np.asarray(sd["epoch"])
        for key in ("net3d.conv0p1s1.kernel", "net3d.final.kernel", "net3d.bn0.bn.weight", "net3d.block8.1.norm2.bn.running_var"):
            out["ckpt:" + key] = sd["state_dict"][key].cpu().numpy()
        out["ckpt_keys"] = np.asarray(list(sd["state_dict"].keys()))
        out["optimizer_steps"] = np.asarray(float(next(iter(sd["optimizer"]["state"].values()))["step"]))
        # ---- evaluation, both feature types the path covers (fusion mode runs no network)
        import run.evaluate as evaluate_mod                    # noqa: E402  (the reference's file, unmodified)
        from util import metric
        evaluate_mod.extract_text_feature = lambda labelset, args: text_features().cuda()
        real_metric = metric.evaluate

        def recording_metric(pred, gt, **k):
            p0, g0 = np.asarray(pred).copy(), np.asarray(gt).copy()          # (confusion_matrix rewrites 256 -> n_classes in place)
            res = real_metric(pred, gt, **k)
            rec.evals.append((p0, g0, res))
            return res
        metric.evaluate = recording_metric
        import dataset.feature_loader as fl
        _record_collate(fl, "collation_fn_eval_all", rec, "eval")
        _instrument(evaluate_mod, rec, "evaluate")
        for mode in ("distill", "ensemble"):
            n0, e0, b0 = len(rec.tensors), len(rec.evals), len(rec.batches)
            monkeypatch.setattr(sys, "argv", ["evaluate.py", "--config", cfg(mode, ckpt)])
            random.seed(7); np.random.seed(7); torch.manual_seed(7)
            evaluate_mod.main()
            seen, eb = rec.tensors[n0:], [b for w, b in rec.batches[b0:] if w == "eval"]
            assert len(seen) == len(eb)
            for i, ((_, feats, coords), (c0, f0, lab, feat_3d, mask, inds)) in enumerate(zip(seen, eb)):
                assert torch.equal(coords, c0)
                out["eval_%s_%d_coords" % (mode, i)] = coords.numpy().astype(np.int32)
                out["eval_%s_%d_labels" % (mode, i)] = lab.numpy().astype(np.uint8)
                out["eval_%s_%d_inds" % (mode, i)] = inds.numpy().astype(np.int32)
                out["eval_%s_%d_mask" % (mode, i)] = mask.numpy()
                code = feat_3d[:, 0].float().round().to(torch.int64)             # 0 = the zero row of a voxel without a 2-D feature
                full = torch.zeros_like(feat_3d)
                full[code > 0] = codebook()[code[code > 0] - 1]
                assert torch.equal(full, feat_3d)
                out["eval_%s_%d_code" % (mode, i)] = code.numpy().astype(np.uint8)
            out["eval_%s_batches" % mode] = np.asarray(len(seen))
            for j, (pred, gt, res) in enumerate(rec.evals[e0:]):
                out["eval_%s_rep%d_pred" % (mode, j)] = pred.astype(np.int16)
                out["eval_%s_rep%d_gt" % (mode, j)] = gt.astype(np.int16)
                out["eval_%s_rep%d_miou" % (mode, j)] = np.asarray(float(res))
            out["eval_%s_reps" % mode] = np.asarray(len(rec.evals) - e0)
        metric.evaluate = real_metric
    return out




# ------------------------------------------------------------------------------------------------- replay
def _iou_counts(output, target, K, ignore_index=255):
    """util/util.py:133-146 (intersectionAndUnionGPU) on whatever device the tensors are on."""
    output, target = output.reshape(-1).clone(), target.reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    hist = lambda t: torch.histc(t.float().cpu(), bins=K, min=0, max=K - 1)
    a_i, a_o, a_t = hist(inter), hist(output), hist(target)
    return a_i.numpy(), (a_o + a_t - a_i).numpy(), a_t.numpy()


def mean_iou(pred, gt, n_classes):
    """util/metric.py:46-75: confusion over the points whose ground truth is not 255; IoU of every class that occurs in the
    ground truth, summed and divided by the NUMBER OF CLASSES (absent classes count as 0 -- the reference's convention)."""
    keep = gt != 255
    conf = np.bincount(pred[keep] * n_classes + gt[keep], minlength=n_classes ** 2).reshape(n_classes, n_classes).astype(np.int64)
    total = 0.0
    for c in range(n_classes):
        if (gt == c).sum() == 0:
            continue
        tp = conf[c, c]
        denom = conf[c, :].sum() + conf[:, c].sum() - tp
        total += float(tp) / denom
    return total / n_classes


def _top2_margin(scores):
    """(top-1 score - top-2 score) / max |score| per point, fp32: how far a label is from flipping."""
    s = scores.float()
    t = s.topk(2, dim=1)[0]
    return ((t[:, 0] - t[:, 1]) / s.abs().max().clamp_min(1e-20)).cpu().numpy().astype(np.float32)


def replay(g, device):
    """The operations of run/distill.py:295-447 and run/evaluate.py:224-425, in their order, on the batches the reference's
    loaders produced (fixture `g`), through openscene_amd's OWN classes: DisNet / SparseTensor (HIP on the GPU box, the test
    backend on the CPU), torch.optim.Adam and torch operators exactly where the loops use torch operators."""
    from openscene_amd.disnet import DisNet
    from openscene_amd.sparse import SparseTensor
    cfg = types.SimpleNamespace(arch_3d=ARCH, feature_2d_extractor="openseg")
    model = load_recipe_weights(DisNet(cfg)).to(device)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}
    optimizer = torch.optim.Adam(model.parameters(), lr=BASE_LR)
    text, cb = text_features().to(device), codebook().to(device)
    T = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    out = {}
    # ---- distill(): run/distill.py:295-359
    model.train()
    n_it = 2
    losses, lrs = [], []
    for i in range(n_it):
        coords, mask = T(g["train%d_coords" % i]), T(g["train%d_mask" % i])
        feat_3d = cb[T(g["train%d_code" % i], torch.int64) - 1]
        sinput = SparseTensor(torch.ones(coords.shape[0], 3, device=device), coords)
        output_3d = model(sinput)[mask]
        loss = (1 - torch.nn.CosineSimilarity()(output_3d, feat_3d)).mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        losses.append(loss.item())
        lr = BASE_LR * (1 - float(i + 1) / n_it) ** 0.9             # util/util.py poly_learning_rate
        for pg in optimizer.param_groups:
            pg["lr"] = lr * 10                                      # run/distill.py:343-346 (index_split = 0)
        lrs.append(lr)
    out["train_loss_batch"], out["train_lr"] = np.asarray(losses), np.asarray(lrs)
    # (run/distill.py:375-378: the label picture of the last batch's first scene)
    first = (coords[mask][:, 0] == 0)
    out["train_pred_labels"] = torch.max(output_3d[first].half() @ text.t(), 1)[1].cpu().numpy()
    # ---- validate(): run/distill.py:403-447 -- NOTE no model.eval(): batch statistics, running statistics move
    crit = torch.nn.CrossEntropyLoss(ignore_index=255)
    inter = union = target = 0.0
    vloss = []
    with torch.no_grad():
        for i in range(int(g["n_val"])):
            coords, label, inds = T(g["val%d_coords" % i]), T(g["val%d_labels" % i], torch.int64), T(g["val%d_inds" % i], torch.int64)
            output = model(SparseTensor(torch.ones(coords.shape[0], 3, device=device), coords))
            output = output[inds, :].half() @ text.t()
            vloss.append(crit(output, label).item())
            out["val%d_margin" % i] = _top2_margin(output)
            pred = torch.max(output, 1)[1]
            out["val%d_pred" % i] = pred.cpu().numpy().astype(np.int16)
            a, b, c = _iou_counts(pred, label, N_LABELS)
            inter, union, target = inter + a, union + b, target + c
    out["val"] = np.asarray([float(np.mean(vloss)), float(np.mean(inter / (union + 1e-10))),
                             float(np.mean(inter / (target + 1e-10))), float(inter.sum() / (target.sum() + 1e-10))])
    sd = model.state_dict()
    for key in ("net3d.conv0p1s1.kernel", "net3d.final.kernel", "net3d.bn0.bn.weight", "net3d.block8.1.norm2.bn.running_var"):
        out["ckpt:" + key] = sd[key].detach().cpu().numpy()
        out["init:" + key] = init[key].cpu().numpy()
    # ---- evaluate(): run/evaluate.py:224-425, feature_type 'distill' and 'ensemble', test_repeats 2
    model.eval()
    with torch.no_grad():
        for mode in ("distill", "ensemble"):
            nb, reps = int(g["eval_%s_batches" % mode]), int(g["eval_%s_reps" % mode])
            per_rep = nb // reps
            store = 0.0
            for rep in range(reps):
                preds = []
                for i in range(rep * per_rep, (rep + 1) * per_rep):
                    coords, inds = T(g["eval_%s_%d_coords" % (mode, i)]), T(g["eval_%s_%d_inds" % (mode, i)], torch.int64)
                    predictions = model(SparseTensor(torch.ones(coords.shape[0], 3, device=device), coords))[inds, :]
                    if mode == "distill":
                        pred = predictions.half() @ text.t()
                    else:
                        code = T(g["eval_%s_%d_code" % (mode, i)], torch.int64)
                        feat_3d = torch.zeros(code.shape[0], D_TEXT, dtype=torch.float16, device=device)
                        feat_3d[code > 0] = cb[code[code > 0] - 1]
                        feat_fuse = feat_3d[inds, :]
                        pred_fusion = (feat_fuse / (feat_fuse.norm(dim=-1, keepdim=True) + 1e-5)).half() @ text.t()
                        pred_distill = (predictions / (predictions.norm(dim=-1, keepdim=True) + 1e-5)).half() @ text.t()
                        feat_ensemble = predictions.clone().half()
                        mask_ = pred_distill.max(dim=-1)[0] < pred_fusion.max(dim=-1)[0]
                        feat_ensemble[mask_] = feat_fuse[mask_]
                        pred = feat_ensemble @ text.t()
                        out["eval_ensemble_%d_took_fusion" % i] = mask_.cpu().numpy()
                    preds.append(pred.cpu())
                store = torch.cat(preds) + store
                out["eval_%s_rep%d_pred" % (mode, rep)] = store.float().max(1)[1].numpy().astype(np.int16)
                out["eval_%s_rep%d_margin" % (mode, rep)] = _top2_margin(store)
                out["eval_%s_rep%d_miou" % (mode, rep)] = np.asarray(
                    mean_iou(out["eval_%s_rep%d_pred" % (mode, rep)].astype(np.int64), g["eval_%s_rep%d_gt" % (mode, rep)].astype(np.int64),
                             N_LABELS))
    return out


MARGIN = 2e-2       # a point whose top-2 scores are closer than this (relative to the largest score) may flip between engines


def compare(gold, got, margins, loss_tol=2e-4, what="", agree_tol=0.999, update_tol=5e-2):
    """Tolerances of a replay (or of the reference run itself) against the fixture; `margins`: where the top-2 margins come
    from (the fixture's CPU replay).  Returns a dict of the measured deviations.
    Why a GPU replay cannot be held tighter than this (measured, tests/test_reference_loops.py prints it for two arithmetic modes):
    the loops train with torch.optim.Adam for TWO steps from fresh moments, where the update of a weight is -lr * sign-like
    (g / (|g| + eps)): an element whose gradient is within fp32 round-off of zero gets +lr or -lr depending on the LAST BIT of a sum of
    ~10^4 products, i.e. on the summation order of the engine.  The exact-fp32 HIP kernels (OSN_CONV_MODE=fp32, bit-for-bit fmaf
    chains) deviate from the CPU fixture by as much as the split-bf16 ones: the tolerance is the arithmetic's, not a kernel's."""
    dev = {}
    dev["train_loss"] = float(np.abs(got["train_loss_batch"] - gold["train_loss_batch"]).max())
    assert dev["train_loss"] <= loss_tol, "%s training losses %s vs %s" % (what, got["train_loss_batch"], gold["train_loss_batch"])
    assert np.allclose(got["train_lr"], gold["train_lr"], rtol=1e-12, atol=0), what
    dev["val_loss_rel"] = float(abs(got["val"][0] - gold["val"][0]) / abs(gold["val"][0]))
    assert dev["val_loss_rel"] <= 5e-3, "%s validation loss %s vs %s" % (what, got["val"][0], gold["val"][0])
    dev["val_metrics"] = float(np.abs(got["val"][1:] - gold["val"][1:]).max())
    assert dev["val_metrics"] <= 5e-3, "%s validation mIoU / mAcc / allAcc %s vs %s" % (what, got["val"][1:], gold["val"][1:])
    for key in [k for k in gold.keys() if k.startswith("ckpt:")]:
        if key in got:
            a, b = got[key].astype(np.float64), gold[key].astype(np.float64)
            if "init:" + key[5:] in got and "running" not in key:
                w0 = got["init:" + key[5:]].astype(np.float64)        # two Adam steps: compare the UPDATE, not the weights
                e = np.linalg.norm((a - w0) - (b - w0)) / max(np.linalg.norm(b - w0), 1e-30)
                assert e <= update_tol, "%s %s: update off by %.3e" % (what, key, e)
            else:
                e = np.linalg.norm(a - b) / np.linalg.norm(b)
                assert e <= 1e-3, "%s %s off by %.3e" % (what, key, e)
            dev[key] = float(e)
    for mode in ("distill", "ensemble"):
        for rep in range(int(gold["eval_%s_reps" % mode])):
            k = "eval_%s_rep%d_" % (mode, rep)
            clear = margins[k + "margin"] > MARGIN
            agree = float((got[k + "pred"][clear] == gold[k + "pred"][clear]).mean())
            dev[k + "agree"] = agree
            dev[k + "clear_frac"] = float(clear.mean())
            assert clear.mean() > 0.5 and agree >= agree_tol, "%s %s labels agree on %.4f of the %.2f clear points" % (what, k, agree, clear.mean())
            dev[k + "miou"] = float(abs(float(got[k + "miou"]) - float(gold[k + "miou"])))
            assert dev[k + "miou"] <= 5e-3, "%s %s mIoU %s vs %s" % (what, k, got[k + "miou"], gold[k + "miou"])
    return dev
