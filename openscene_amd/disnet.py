"""DisNet -- the distillation network wrapper, mirror of ``models/disnet.py:21-40``:
picks the output width from ``cfg.feature_2d_extractor`` (LSeg 512 / OpenSeg 768) and
wraps ``mink_unet(in_channels=3, out_channels=width, D=3, arch=cfg.arch_3d)`` as
``self.net3d`` (=> checkpoint keys ``net3d.*`` as released by the reference)."""
from collections import OrderedDict

from torch import nn

from .mink_unet import mink_unet


def state_dict_remove_moudle(state_dict):
    """Strip DDP's ``module.`` prefix (name kept as in models/disnet.py:8-13)."""
    return OrderedDict((k.replace("module.", ""), v) for k, v in state_dict.items())


class DisNet(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        if not hasattr(cfg, "feature_2d_extractor"):
            cfg.feature_2d_extractor = "openseg"
        if "lseg" in cfg.feature_2d_extractor:
            last_dim = 512
        elif "openseg" in cfg.feature_2d_extractor:
            last_dim = 768
        else:
            raise NotImplementedError
        self.net3d = mink_unet(in_channels=3, out_channels=last_dim, D=3, arch=cfg.arch_3d)

    def forward(self, sparse_3d, rows=None):
        """rows: see MinkUNetBase.forward (only the supervised rows of the output; models/disnet.py:38-40 has no such argument --
        an opt-in of the edited call site)."""
        return self.net3d(sparse_3d) if rows is None else self.net3d(sparse_3d, rows=rows)

    def forward_features(self, sparse_3d):
        """Penultimate features + the head weight, for the fused-head query (openscene_amd.query.query_distill_fused)."""
        return self.net3d.forward_features(sparse_3d), self.net3d.final.kernel
