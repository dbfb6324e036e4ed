"""Stand-alone cost of the batch-norm kernels on the step's large shapes (run under rocprofv3 --kernel-trace for per-kernel
durations; the event times printed here include the launch gaps of the 3 launches of a call).
usage: python tools/micro_bn.py [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from openscene_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for n, c in ((100999, 96), (100999, 32), (47618, 96), (47618, 64), (12912, 128), (3400, 256)):
    x = torch.randn(n, c, device=dev)
    gy = torch.randn(n, c, device=dev)
    res = torch.randn(n, c, device=dev)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.1
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    y, mean, var = ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1)
    yr, _, _ = ops.bn_forward_train(x, gamma, beta, 1e-5, res, True, rm, rv, 0.1)

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

    mb = n * c * 4 / 1e6
    t_f = timed(lambda: ops.bn_forward_train(x, gamma, beta, 1e-5, None, True, rm, rv, 0.1))
    t_b = timed(lambda: ops.bn_backward(x, None, gy, mean, var, gamma, 1e-5, True, True, False, beta=beta))
    t_br = timed(lambda: ops.bn_backward(x, yr, gy, mean, var, gamma, 1e-5, True, True, True))
    print("n %6d c %3d (%.1f MB): fwd %.1f us (3 passes: %.2f TB/s)  bwd mask-from-x %.1f us (5 passes: %.2f TB/s)  bwd residual %.1f us (7 passes: %.2f TB/s)" % (
        n, c, mb, t_f, 3 * mb / t_f, t_b, 5 * mb / t_b, t_br, 7 * mb / t_br), flush=True)
