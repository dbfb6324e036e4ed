"""Stand-alone cost of the batch-norm calls on the step's shapes: the three-launch path against the one-launch kernels
(osn_bn_forward_train3 / osn_bn_backward_multi3, csrc/bn.hip).  HIP events over back-to-back calls, launch gaps included.
usage: python tools/micro_bn.py [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from openscene_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def timed_graph(f):
    """The same calls replayed from a HIP graph: the device's own pace (kernel time + dependent-launch boundaries), no host in the loop."""
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            f()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (5 * iters)


MODE = sys.argv[2] if len(sys.argv) > 2 else "eager"
if MODE == "graph":
    timed = timed_graph
print("# timing mode: %s" % MODE)
print("# us per call (fwd = statistics + apply; bwd-x = ReLU mask from x; bwd-res = residual block tail): three launches | one launch")
for n, c in ((730, 256), (730, 512), (3326, 128), (3326, 384), (13393, 64), (13393, 128), (13393, 192), (52125, 32), (52125, 96),
             (52125, 128), (100999, 32), (100999, 96)):
    x = torch.randn(n, c, device=dev)
    gy = torch.randn(n, c, device=dev)
    res = torch.randn(n, c, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    row = []
    for on in (0, 1):
        ops.bn_xb_config(on, 1, 200000)
        y, mean, var = ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1)
        yr, _, _ = ops.bn_forward_train(x, gamma, beta, 1e-5, res, True, rm, rv, 0.1)
        t_f = timed(lambda: ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1))
        t_b = timed(lambda: ops.bn_backward(x, None, gy, mean, var, gamma, 1e-5, True, True, False, beta=beta))
        t_br = timed(lambda: ops.bn_backward(x, yr, gy, mean, var, gamma, 1e-5, True, True, True))
        row.append((t_f, t_b, t_br))
    ops.bn_xb_config(0, 0, 16384)
    ops.bn_sync_check(dev)
    print("n %6d c %3d (%5.1f MB): fwd %5.1f | %5.1f   bwd-x %5.1f | %5.1f   bwd-res %5.1f | %5.1f" % (
        n, c, n * c * 4 / 1e6, row[0][0], row[1][0], row[0][1], row[1][1], row[0][2], row[1][2]), flush=True)
