"""The N>1 path on CPU: two gloo ranks, one scene each, torch DDP around the model exactly as
run/distill.py:149-150 wraps it; the all-reduced gradients must equal the mean of the two
single-rank gradients and parameters/BN buffers must be broadcast from rank 0.  The HIP ops
are replaced by tests/cpu_backend.py inside the worker processes (host-logic test)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _install_cpu_backend():
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import cpu_backend
    import openscene_amd.ops as ops
    for n in cpu_backend._NAMES:
        setattr(ops, n, getattr(cpu_backend, n))


def _scene(seed, n_draw=220):
    rng = np.random.default_rng(seed)
    g = np.unique(rng.integers(0, 10, (n_draw, 3)), axis=0)
    g = g[rng.permutation(g.shape[0])]
    return torch.from_numpy(np.concatenate([np.zeros((g.shape[0], 1)), g], 1).astype(np.int32))


def _loss(model, seed, n_draw=220):
    from openscene_amd.sparse import SparseTensor
    c = _scene(seed, n_draw)
    out = model(SparseTensor(torch.ones(c.shape[0], 3, dtype=torch.float64), c))
    tgt = torch.randn(out.shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)
    return (1 - torch.nn.functional.cosine_similarity(out, tgt)).mean()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    _install_cpu_backend()
    from openscene_amd.mink_unet import mink_unet
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                          # different init per rank: DDP must broadcast rank 0's
    model = mink_unet(3, 8, 3, "MinkUNet14A").double()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    loss = _loss(ddp, seed=10 + rank)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in model.named_parameters()}
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    torch.save({"grads": grads, "params": params, "loss": float(loss)}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for n in r0["params"]:
        assert torch.equal(r0["params"][n], r1["params"][n]), "parameter %s not broadcast" % n
        assert torch.equal(r0["grads"][n], r1["grads"][n]), "gradient %s not all-reduced" % n
    # single-process reference: same weights (rank 0's), mean of the two scenes' gradients
    _install_cpu_backend()
    try:
        from openscene_amd.mink_unet import mink_unet
        torch.manual_seed(100)
        model = mink_unet(3, 8, 3, "MinkUNet14A").double()
        for n, p in model.named_parameters():
            assert torch.equal(p.detach(), r0["params"][n])
        acc = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
        for seed in (10, 11):
            model.zero_grad()
            _loss(model, seed).backward()
            for n, p in model.named_parameters():
                acc[n] += p.grad / world
        for n in acc:
            scale = acc[n].abs().max().item() + 1e-12
            assert (acc[n] - r0["grads"][n]).abs().max().item() <= 1e-9 * scale + 1e-12, n
    finally:
        import importlib
        import openscene_amd.ops as ops
        importlib.reload(ops)


# ---- four ranks, scenes of UNEQUAL size handed out by DistributedSampler (run/distill.py:183-184), two steps with an
# optimizer in between: after every step all ranks hold the same parameters, and the first step's gradients are the
# mean over the four ranks' scenes.
_SIZES = [60, 140, 220, 400, 90, 300, 180, 260]            # scene i draws _SIZES[i] lattice points


def _worker4(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    _install_cpu_backend()
    from openscene_amd.mink_unet import mink_unet
    from torch.utils.data.distributed import DistributedSampler
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(7 + rank)
    model = mink_unet(3, 8, 3, "MinkUNet14A").double()
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    optim = torch.optim.SGD(ddp.parameters(), lr=1e-2)
    sampler = DistributedSampler(list(range(len(_SIZES))), num_replicas=world, rank=rank, shuffle=False)
    mine = list(iter(sampler))                              # two scenes per rank
    first_grads = None
    for step, idx in enumerate(mine):
        optim.zero_grad(set_to_none=True)
        _loss(ddp, seed=20 + idx, n_draw=_SIZES[idx]).backward()
        if step == 0:
            first_grads = {n: p.grad.clone() for n, p in model.named_parameters()}
        optim.step()
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in model.named_buffers()}
    torch.save({"grads": first_grads, "params": params, "bufs": bufs, "mine": mine}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_four_ranks_unequal_scenes_distributed_sampler(tmp_path):
    world = 4
    mp.spawn(_worker4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in range(world)]
    assert sorted(i for r in res for i in r["mine"]) == list(range(len(_SIZES)))      # every scene exactly once
    for r in res[1:]:
        for n in res[0]["params"]:
            assert torch.equal(res[0]["params"][n], r["params"][n]), "rank parameters diverged at %s" % n
            assert torch.equal(res[0]["grads"][n], r["grads"][n]), "gradient %s not all-reduced" % n
    # BN statistics are LOCAL (no SyncBN in the reference): running buffers differ between ranks with different scenes
    assert any(not torch.equal(res[0]["bufs"][n], res[1]["bufs"][n]) for n in res[0]["bufs"] if "running_mean" in n)
    _install_cpu_backend()
    try:
        from openscene_amd.mink_unet import mink_unet
        torch.manual_seed(7)
        model = mink_unet(3, 8, 3, "MinkUNet14A").double()
        acc = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
        for r in res:
            idx = r["mine"][0]
            model.zero_grad()
            for m in model.modules():                        # fresh BN buffers per rank, as each rank started
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.reset_running_stats()
            _loss(model, 20 + idx, _SIZES[idx]).backward()
            for n, p in model.named_parameters():
                acc[n] += p.grad / world
        for n in acc:
            scale = acc[n].abs().max().item() + 1e-12
            assert (acc[n] - res[0]["grads"][n]).abs().max().item() <= 1e-9 * scale + 1e-12, n
    finally:
        import importlib
        import openscene_amd.ops as ops
        importlib.reload(ops)


# ---- the flat one-collective exchange (openscene_amd/distributed.py) gives what DDP gives: same broadcast parameters,
# gradients = mean over the ranks (same scenes as above), buffers of rank 0 on every rank after sync_buffers.
def _worker_flat(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    _install_cpu_backend()
    from openscene_amd.distributed import FlatGradAllReduce
    from openscene_amd.mink_unet import mink_unet
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    model = mink_unet(3, 8, 3, "MinkUNet14A").double()
    ex = FlatGradAllReduce(model)
    optim = torch.optim.SGD(model.parameters(), lr=1e-2)
    first = None
    for step in range(2):
        ex.sync_buffers()
        optim.zero_grad(set_to_none=True)
        _loss(model, seed=10 + rank + 2 * step).backward()
        ex.reduce_gradients()
        if step == 0:
            first = {n: p.grad.clone() for n, p in model.named_parameters()}
        optim.step()
    torch.save({"grads": first, "params": {n: p.detach().clone() for n, p in model.named_parameters()},
                "bufs": {n: b.detach().clone() for n, b in model.named_buffers()}}, os.path.join(out_dir, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_equals_ddp(tmp_path):
    world = 2
    mp.spawn(_worker_flat, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)          # DDP, first step only
    f0 = torch.load(os.path.join(tmp_path, "f0.pt"))
    f1 = torch.load(os.path.join(tmp_path, "f1.pt"))
    d0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    for n in f0["params"]:
        assert torch.equal(f0["params"][n], f1["params"][n]), "parameters diverged at %s" % n
        assert torch.equal(f0["grads"][n], f1["grads"][n]), "gradient %s differs between ranks" % n
        scale = d0["grads"][n].abs().max().item() + 1e-12
        assert (f0["grads"][n] - d0["grads"][n]).abs().max().item() <= 1e-12 * scale + 1e-15, n     # == DDP's mean


def _worker_flat_edge(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    from openscene_amd.distributed import FlatGradAllReduce
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(rank)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3), torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.full((5,), float(rank)))
    net.register_parameter("unused", unused)
    ex = FlatGradAllReduce(net)
    x = torch.randn(8, 4, generator=torch.Generator().manual_seed(50 + rank))
    net(x).sum().backward()                                 # `unused` gets no gradient; BN buffers move locally
    ex.reduce_gradients()
    before = {n: b.clone() for n, b in net.named_buffers()}
    ex.sync_buffers()
    torch.save({"g": {n: p.grad.clone() for n, p in net.named_parameters()}, "p": {n: p.detach().clone() for n, p in net.named_parameters()},
                "buf_before": before, "buf": {n: b.clone() for n, b in net.named_buffers()}}, os.path.join(out_dir, "e%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_exchange_unused_parameter_and_buffer_sync(tmp_path):
    mp.spawn(_worker_flat_edge, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "e0.pt"))
    b = torch.load(os.path.join(tmp_path, "e1.pt"))
    for n in a["p"]:
        assert torch.equal(a["p"][n], b["p"][n])            # rank 0's parameters everywhere (also the unused one: all 0.0)
        assert torch.equal(a["g"][n], b["g"][n])
    assert torch.equal(a["g"]["unused"], torch.zeros(5))    # no gradient anywhere -> zero, not None
    assert any(not torch.equal(a["buf_before"][n], b["buf_before"][n]) for n in a["buf"] if "running" in n)   # local statistics ...
    for n in a["buf"]:
        assert torch.equal(a["buf"][n], b["buf"][n]) and torch.equal(a["buf"][n], a["buf_before"][n])         # ... until sync_buffers


def _worker_flat_views(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    from openscene_amd.distributed import FlatGradAllReduce
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    ex = FlatGradAllReduce(net)
    # gradients handed out as views of ONE flat buffer with 4-float-aligned slices (what the network executor does)
    params = list(net.parameters())
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    flat = torch.full((off,), float("nan"))                 # padding holds garbage: it is exchanged but never read
    for p, o in zip(params, offs):
        v = flat[o:o + p.numel()].view_as(p)
        v.copy_(torch.full_like(p, float(rank + 1)) * (o + 1))
        p.grad = v
    ptr = flat.data_ptr()
    ex.reduce_gradients()
    assert all(p.grad._base is flat and p.grad.data_ptr() == ptr + 4 * o for p, o in zip(params, offs)), "gradients were copied"
    torch.save({"g": [p.grad.clone() for p in params], "offs": offs}, os.path.join(out_dir, "v%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_exchange_reduces_the_executors_gradient_buffer_in_place(tmp_path):
    """Gradients that already are views of one flat buffer (the network executor's layout) are all-reduced in place:
    no gather copy, p.grad keeps pointing into that buffer, value = mean over ranks."""
    mp.spawn(_worker_flat_views, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "v0.pt"))
    b = torch.load(os.path.join(tmp_path, "v1.pt"))
    for ga, gb, o in zip(a["g"], b["g"], a["offs"]):
        assert torch.equal(ga, gb) and torch.equal(ga, torch.full_like(ga, 1.5 * (o + 1)))


def _worker_sliced(rank, world, port, out_dir):
    """The overlapped exchange on a stand-in for the executor: a backward pass played in segments calls the hook with the
    slice of kernel gradients each segment finished (highest ops first); reduce_gradients() sends what is left."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    from openscene_amd.distributed import FlatGradAllReduce
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(*[torch.nn.Linear(7, 7) for _ in range(6)])
    ex = FlatGradAllReduce(net)

    class FakeExecutor:
        grad_ready_hook = None
        grad_segments = 1
        _cuts = None
    fake = FakeExecutor()
    ex.attach(fake, segments=3)
    assert fake.grad_ready_hook is not None and fake.grad_segments == 3
    # the executor's layout: every weight first ("kernels"), then the biases ("batch-norm region"), 4-float aligned slices
    params = [m.weight for m in net] + [m.bias for m in net]
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) // 4 * 4
    kernels_end = offs[6]
    for step in range(3):                                   # three steps: the state of the exchange resets between them
        flat = torch.full((off,), float("nan"))
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view_as(p)
            v.copy_(torch.full_like(p, float(rank + 1 + step)) * (o + 1))
            # what autograd leaves in p.grad: the slice's memory, detached (no `_base`) -- steps 0, 1; or a CLONE taken when
            # the node returned, i.e. right after the last hook call (hooks on a parameter make AccumulateGrad copy) -- step 2
            p.grad = v.detach() if step < 2 else None
        for lo_i, hi_i in ((4, 6), (2, 4), (0, 2)):          # segments, highest "ops" first
            fake.grad_ready_hook(flat, offs[lo_i], offs[hi_i] if hi_i < 6 else kernels_end, lo_i == 0)
        assert ex._pending == []                            # the last segment's hook waited for every slice in flight
        if step == 2:
            for p, o in zip(params, offs):
                p.grad = flat[o:o + p.numel()].view_as(p).clone()
        ex.reduce_gradients()
        if step < 2:
            assert all(p.grad._base is None and p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
                       for p in params), "gradients were copied"
        assert ex._pending == [] and ex._sliced is None
    torch.save({"g": [p.grad.clone() for p in params], "offs": offs}, os.path.join(out_dir, "s%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sliced_exchange_during_a_segmented_backward_pass(tmp_path):
    """the all-reduce goes out in pieces as the backward pass finishes them; every element ends up as the
    mean over ranks exactly as with the one-collective exchange."""
    mp.spawn(_worker_sliced, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a = torch.load(os.path.join(tmp_path, "s0.pt"))
    b = torch.load(os.path.join(tmp_path, "s1.pt"))
    for ga, gb, o in zip(a["g"], b["g"], a["offs"]):
        assert torch.equal(ga, gb) and torch.equal(ga, torch.full_like(ga, 3.5 * (o + 1)))     # step 2: mean of 3 and 4
