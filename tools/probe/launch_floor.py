"""Probe: cost per trivial kernel inside a replayed hipGraph vs eager back-to-back launches."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchsr_amd import _lib
dev = torch.device('cuda:0')
x = torch.rand(16, device=dev); y = torch.empty_like(x)
big = torch.rand(16 * 24 * 24 * 64, device=dev); bigy = torch.empty_like(big)
s = torch.cuda.current_stream().cuda_stream
def tiny(): _lib.call('srx_sigmoid_fwd', x.data_ptr(), y.data_ptr(), 16, s)
def med(): _lib.call('srx_lrelu_fwd', big.data_ptr(), bigy.data_ptr(), big.numel(), 0.2, s)
for name, fn in (('1-block kernel', tiny), ('2.4 MB lrelu pass', med)):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(500): fn()
    s = torch.cuda.current_stream().cuda_stream
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(f'{name:20s}: {(time.perf_counter() - t0) / 10 / 500 * 1e6:6.2f} us per kernel in a hipGraph of 500')
