"""Stand-alone cost of the batch-norm calls on the step's shapes.  HIP events over back-to-back calls.
usage: python tools/micro_bn.py [iters] [eager|graph]
eager: the calls as Python issues them -- on the small maps this measures the HOST (three ctypes calls + torch.empty per call:
~20 us whatever the size); graph: the same calls replayed from a HIP graph -- the device's own pace (kernel time + dependent-launch
boundaries): 8.5 - 15 us on the <= 13 k-row maps (profiles/r04_s13_bn_one_launch_exchange_rejected.txt)."""
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
print("# us per call (fwd = statistics + apply; bwd-x = ReLU mask from x; bwd-res = residual block tail)")
for n, c in ((730, 256), (730, 512), (3326, 128), (3326, 384), (13393, 64), (13393, 128), (13393, 192), (52125, 32), (52125, 96),
             (52125, 128), (100999, 32), (100999, 96)):
    x = torch.randn(n, c, device=dev)
    gy = torch.randn(n, c, device=dev)
    res = torch.randn(n, c, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    y, mean, var = ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1)
    yr, _, _ = ops.bn_forward_train(x, gamma, beta, 1e-5, res, True, rm, rv, 0.1)
    t_f = timed(lambda: ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1))
    t_b = timed(lambda: ops.bn_backward(x, None, gy, mean, var, gamma, 1e-5, True, True, False, beta=beta))
    t_br = timed(lambda: ops.bn_backward(x, yr, gy, mean, var, gamma, 1e-5, True, True, True))
    mb = n * c * 4 / 1e6
    print("n %6d c %3d (%5.1f MB): fwd %5.1f us (3 passes: %.2f TB/s)   bwd-x %5.1f us (5 passes: %.2f TB/s)   bwd-res %5.1f us (7 passes: %.2f TB/s)" % (
        n, c, mb, t_f, 3 * mb / t_f, t_b, 5 * mb / t_b, t_br, 7 * mb / t_br), flush=True)
