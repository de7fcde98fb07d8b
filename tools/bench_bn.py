"""Developer micro-benchmark: in-graph time of the BatchNorm forward / backward op pair at the SRGAN shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd.layers import BatchNorm2d, Conv2d  # noqa: E402
from torchsr_amd._lib import ACT_PRELU  # noqa: E402

dev = torch.device('cuda:0')


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, n, h, w, c in (('G 64ch @24', 16, 24, 24, 64), ('D 128ch @48', 16, 48, 48, 128), ('D 512ch @6', 16, 6, 6, 512)):
    conv = Conv2d(c, c, 3, 1, 1, bias=False).to(dev)
    bn = BatchNorm2d(c).to(dev).train()
    prelu = torch.nn.Parameter(torch.full((1,), 0.25, device=dev))
    x = torch.rand(n, h, w, c, device=dev)
    reps = 20

    def fwd_only():
        with torch.no_grad():
            y, part = conv(x, want_stats=True)
            return bn(y, part, act=ACT_PRELU, prelu=prelu)

    def conv_only():
        with torch.no_grad():
            return conv(x, want_stats=True)

    res = {}
    for tag, fn in (('conv', conv_only), ('conv+bn', fwd_only)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        res[tag] = timeit(g.replay) / reps
    xg = x.clone().requires_grad_(True)

    def fwd_bwd():
        y, part = conv(xg, want_stats=True)
        out = bn(y, part, act=ACT_PRELU, prelu=prelu)
        torch.autograd.grad(out, y, gout)

    def fwd_grad():
        y, part = conv(xg, want_stats=True)
        return bn(y, part, act=ACT_PRELU, prelu=prelu)

    gout = torch.rand(n, h, w, c, device=dev)
    for tag, fn in (('fwd(grad)', fwd_grad), ('fwd+bnbwd', fwd_bwd)):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        res[tag] = timeit(g.replay) / reps
    print(f'{name:14s} conv {res["conv"]:6.2f} us   BN forward {res["conv+bn"] - res["conv"]:6.2f} us   '
          f'BN backward {res["fwd+bnbwd"] - res["fwd(grad)"]:6.2f} us', flush=True)
