"""The N > 1 path ON the GPU: two ranks sharing device 0 over gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run
is the driver's), the real network executor, the real kernels.  What the CPU tests (test_ddp_gloo.py) cannot see: that the
slices the executor hands to the exchange DURING the backward pass are final when the collective reads them, and that the
collective's result is not overwritten by a later kernel -- run/distill.py:149-150 (DistributedDataParallel around the
model) is the reference behaviour: every rank ends a step with the mean gradient and identical parameters."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(HERE))
    from openscene_amd import executor as E
    from openscene_amd import synthetic as syn
    from openscene_amd.distributed import FlatGradAllReduce
    from openscene_amd.mink_unet import mink_unet
    from openscene_amd.sparse import SparseTensor
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(100 + rank)                          # different init per rank: the exchange must broadcast rank 0's
    model = mink_unet(3, 64, 3, "MinkUNet18A").to(dev).train()
    vox = syn.shuffled(syn.grid_voxels(syn.room_points(10 + rank, n_pts=20000), 0.04), 10 + rank)
    coords = torch.from_numpy(np.concatenate([np.zeros((vox.shape[0], 1), np.int32), vox.astype(np.int32)], 1)).to(dev)
    feats = torch.rand(coords.shape[0], 3, generator=torch.Generator().manual_seed(rank)).to(dev)
    ex = E.for_model(model)
    exchange = FlatGradAllReduce(model)
    seen = []
    if mode != "one":
        exchange.attach(ex, segments=int(mode))
        on_slice = ex.grad_ready_hook

        def spy(flat, lo, hi, last):
            seen.append((lo, hi))
            return on_slice(flat, lo, hi, last)
        ex.grad_ready_hook = spy
    opt = torch.optim.SGD(model.parameters(), lr=0.05)     # (not Adam: its update hides a wrongly scaled gradient)
    record = {"n_vox": int(coords.shape[0])}
    for step in range(STEPS):
        opt.zero_grad(set_to_none=True)
        out = model(SparseTensor(feats, coords))
        out.square().mean().backward()
        if mode == "one" and step == 0:                    # the definition: mean over ranks of the local gradients
            local = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
            both = [torch.empty_like(local) for _ in range(world)]
            dist.all_gather(both, local)
            record["expected0"] = (both[0] / world + both[1] / world).cpu()
        exchange.reduce_gradients()
        if step == 0:
            record["grads0"] = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
        opt.step()
    assert len(seen) == (0 if mode == "one" else int(mode) * STEPS), "the executor's backward pass must have fed the exchange"
    record["params"] = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    record["running"] = torch.cat([b.detach().reshape(-1).float() for b in model.buffers()]).cpu()
    torch.save(record, os.path.join(out_dir, "%s_%d.pt" % (mode, rank)))
    dist.destroy_process_group()


def _run(mode, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), mode, str(tmp_path)), nprocs=2, join=True)
    return [torch.load(os.path.join(str(tmp_path), "%s_%d.pt" % (mode, r))) for r in range(2)]


def test_two_ranks_on_one_device_sliced_exchange_equals_one_collective(tmp_path):
    one = _run("one", tmp_path)
    assert one[0]["n_vox"] != one[1]["n_vox"]                                  # two different scenes
    assert torch.equal(one[0]["grads0"], one[0]["expected0"])                  # the exchanged gradient IS the mean of the local ones
    assert torch.equal(one[0]["grads0"], one[1]["grads0"])
    assert torch.equal(one[0]["params"], one[1]["params"])                     # ranks stay in step
    assert not torch.equal(one[0]["running"], one[1]["running"])               # batch statistics are local (no SyncBN in the reference)
    for mode in ("4", "2"):
        sl = _run(mode, tmp_path)
        assert torch.equal(sl[0]["grads0"], one[0]["grads0"]), mode            # slices during backward: same bits
        assert torch.equal(sl[0]["grads0"], sl[1]["grads0"]), mode
        assert torch.equal(sl[0]["params"], one[0]["params"]), mode            # ... also after STEPS optimizer steps
        assert torch.equal(sl[0]["params"], sl[1]["params"]), mode
