"""Mint tests/golden/unet_dense_ref.npz: the REFERENCE's own models/mink_unet.py, unmodified, executed on
tests/golden/dense_me.py -- a stand-in MinkowskiEngine whose convolutions are torch's dense conv3d /
conv_transpose3d (MinkowskiEngine itself is an un-vendored dependency, absent from /root/reference).

Run in the authoring container only (needs /root/reference; ~2 minutes of float64 dense convolutions per fixture):
    python tests/golden/make_golden_unet.py [a] [b]      (a: MinkUNet18A -> unet_dense_ref.npz; b: MinkUNet34C -> unet_dense_ref_34c.npz,
                                                          same cloud and parameter recipe, every 4th output row stored)
Inputs and parameters come from unet_recipe.py (numpy generators: reproducible anywhere).  Stored, all float64:
  coords, feats                           the input (two scenes, 48^3 grid)
  out_eval                                model.eval() forward with the recipe's running statistics
  out_train                               model.train() forward (batch statistics)
  gfeats                                  d loss / d feats, loss = sum(out_train * unet_recipe.output_weights)
  names, gproj, gnorm                     per parameter: <grad, probe(name)> and |grad|
  rnames, rproj                           per running buffer AFTER the training forward: <buffer, probe(name)>
The GPU box has no /root/reference: tests read the fixture only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dense_me  # noqa: E402
import unet_recipe as R  # noqa: E402


def mint(arch, fname, row_step):
    from models.mink_unet import mink_unet                     # the reference file, unmodified
    import MinkowskiEngine as ME
    assert ME.__dense_emulation__
    model = mink_unet(R.IN_CH, R.OUT_CH, 3, arch).double()
    sd = model.state_dict()
    for name, t in sd.items():
        v = R.parameter(name, tuple(t.shape))
        if v is not None:
            t.copy_(torch.from_numpy(v))
    model.load_state_dict(sd)
    coords, feats = R.cloud()
    x = torch.from_numpy(feats).requires_grad_(True)

    model.eval()
    with torch.no_grad():
        out_eval = model(ME.SparseTensor(x.detach(), torch.from_numpy(coords)))
    model.train()
    out_train = model(ME.SparseTensor(x, torch.from_numpy(coords)))
    loss = (out_train * torch.from_numpy(R.output_weights(coords.shape[0]))).sum()
    loss.backward()

    names, gproj, gnorm = [], [], []
    for name, p in model.named_parameters():
        names.append(name)
        gproj.append(float((p.grad * torch.from_numpy(R.probe(name, tuple(p.shape)))).sum()))
        gnorm.append(float(p.grad.norm()))
    rnames, rproj = [], []
    for name, b in model.named_buffers():
        if "running" in name:
            rnames.append(name)
            rproj.append(float((b * torch.from_numpy(R.probe(name, tuple(b.shape)))).sum()))
    path = os.path.join(HERE, fname)
    extra = {} if row_step == 1 else {"row_step": row_step}
    coords_feats = {"coords": coords, "feats": feats} if row_step == 1 else {}      # (the second fixture shares the first one's cloud)
    np.savez_compressed(path, out_eval=out_eval.numpy()[::row_step], out_train=out_train.detach().numpy()[::row_step],
                        gfeats=x.grad.numpy()[::row_step], names=np.array(names), gproj=np.array(gproj), gnorm=np.array(gnorm),
                        rnames=np.array(rnames), rproj=np.array(rproj), loss=float(loss.detach()), **coords_feats, **extra)
    print("wrote %s (%s): %d voxels, |out_train| %.6g, |out_eval| %.6g, loss %.9g, %d parameters, %.1f KB"
          % (path, arch, coords.shape[0], float(out_train.norm()), float(out_eval.norm()), float(loss), len(names),
             os.path.getsize(path) / 1024))


def main():
    dense_me.install()
    sys.path.insert(0, "/root/reference")
    torch.set_num_threads(os.cpu_count() or 8)
    which = sys.argv[1:] or ["a", "b"]
    if "a" in which:
        mint(R.ARCH, "unet_dense_ref.npz", 1)
    if "b" in which:
        mint(R.ARCH_B, "unet_dense_ref_34c.npz", R.ROW_STEP_B)


if __name__ == "__main__":
    main()
