"""Mint tests/golden/ref_loops.npz: the reference's run/distill.py and run/evaluate.py executed UNMODIFIED (tests/ref_loops.py)
on the CPU test backend, plus the top-2 margins of a CPU replay (which labels are stable enough to compare across engines).

Authoring container only (needs /root/reference):   python tests/golden/make_golden_loops.py
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ref_loops as RL  # noqa: E402


def main():
    root = tempfile.mkdtemp(prefix="osn_loops_")
    mp = pytest.MonkeyPatch()
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            gold = RL.run_reference(root, mp)
    finally:
        mp.undo()
        shutil.rmtree(root, ignore_errors=True)
    mp = pytest.MonkeyPatch()
    try:
        import cpu_backend
        cpu_backend.install(mp)
        import torch
        rep = RL.replay(gold, torch.device("cpu"))
    finally:
        mp.undo()
    for k, v in rep.items():
        if k.endswith("margin") or k.endswith("took_fusion"):
            gold["replay:" + k] = v
    margins = {k[len("replay:"):]: v for k, v in gold.items() if k.startswith("replay:")}
    dev = RL.compare(gold, rep, margins, what="CPU replay vs the reference run")
    print({k: (round(v, 6) if isinstance(v, float) else v) for k, v in dev.items()})
    gold.pop("pred_cloud_colors", None)
    np.savez_compressed(os.path.join(HERE, "ref_loops.npz"), **gold)
    print("wrote ref_loops.npz: %d KB" % (os.path.getsize(os.path.join(HERE, "ref_loops.npz")) // 1024))


if __name__ == "__main__":
    main()
