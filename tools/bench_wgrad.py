"""In-graph time of the fp32 weight-gradient launch of one layer shape over row-split counts (developer tool).

    python tools/bench_wgrad.py N,H,W,Cin,Cout[,nprob[,stride]] [ns ns ...]      (ns 0 = the planner's choice)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
spec = [int(v) for v in sys.argv[1].split(',')]
n, h, w, cin, cout = spec[:5]
nprob = spec[5] if len(spec) > 5 else 1
stride = spec[6] if len(spec) > 6 else 1
splits = [int(v) for v in sys.argv[2:]] or [0]
ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, stride, 1, 0, 0, 0.0, 0, 0)
xs = [torch.rand(n, h, w, cin, device=dev) for _ in range(nprob)]
dys = [torch.rand(n, ho, wo, cout, device=dev) for _ in range(nprob)]
dws = [torch.zeros(cout, cin, 3, 3, device=dev) for _ in range(nprob)]
arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
gf = 2.0 * n * ho * wo * cout * cin * 9 * nprob / 1e9
reps = 20
for ns in splits:
    if ns:
        os.environ['SRX_WGRAD_NSPLIT'] = str(ns)
    else:
        os.environ.pop('SRX_WGRAD_NSPLIT', None)
    nws = _lib.lib().srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), nprob)
    ws = torch.empty(nws, device=dev)

    def run():
        _lib.call('srx_conv2d_bwd_weight_multi', C.byref(d), nprob, 1, arr(xs), arr(dys), arr(dws), 1, None, ws.data_ptr(), nws,
                  torch.cuda.current_stream().cuda_stream)
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            run()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * reps)
    print(f'{sys.argv[1]} ns {ns:2d}: {us:8.2f} us per launch (+ reduce) in-graph, {gf / us * 1e3:6.1f} TF/s', flush=True)
