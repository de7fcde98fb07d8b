"""Developer tool: tile plans for ESRGAN's dense-block convs (bf16 products, 16x32x32 pixels), timed as 20 back-to-back
launches inside a replayed hipGraph.  SRX_FORCE_PLAN="BM,BN,split,ks" overrides the planner per run.

    python tools/experiments/tune_dense.py
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchsr_amd import _lib  # noqa: E402
from torchsr_amd.layers import Conv2d, set_conv_precision  # noqa: E402

dev = torch.device('cuda:0')
n, h, w = 16, 32, 32
REPS = 20


def graph_time(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * REPS) * 1e3


def run(cin, cout, which, plans):
    conv = Conv2d(cin, cout, 3, 1, 1, act=2, slope=0.2).to(dev)
    set_conv_precision(conv, 'bf16')
    x = torch.rand(n, h, w, 192, device=dev)
    conv(x[..., :cin].contiguous())  # packs
    st = conv._st
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    d = _lib.Conv2dDesc(n, h, w, cin, 192, cout, 192, 3, 3, 1, 1, 0, 2, 0.2, 0, 1)
    out = torch.empty(n, h, w, 192, device=dev)
    ws = torch.empty(1 << 24, device=dev)
    res = []
    for plan in plans:
        if plan:
            os.environ['SRX_FORCE_PLAN'] = plan
        else:
            os.environ.pop('SRX_FORCE_PLAN', None)
        try:
            if which == 'fwd':
                fn = lambda: _lib.call('srx_conv2d_fwd', C.byref(d), x.data_ptr(), st.wpk_fwd.data_ptr(), conv.bias.data_ptr(),  # noqa: E731
                                       out.data_ptr(), None, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            else:
                fn = lambda: _lib.call('srx_conv2d_bwd_data_act', C.byref(d), out.data_ptr(), st.wpk_bwd.data_ptr(), x.data_ptr(),  # noqa: E731
                                       0.2, max(cin - 32, 0), cin, 1, x.data_ptr(), ws.data_ptr(), ws.numel(),
                                       torch.cuda.current_stream().cuda_stream)
            t = graph_time(fn)
            res.append(f'{plan or "planner":>12s} {t:6.2f} us')
        except RuntimeError as e:
            res.append(f'{plan:>12s} failed ({str(e)[-40:]})')
    os.environ.pop('SRX_FORCE_PLAN', None)
    print(f'{which:5s} {cin:3d}->{cout:3d}:  ' + ' | '.join(res), flush=True)
    del s


FWD32 = ['', '64,32,1,4', '128,32,1,1', '128,32,2,1']
for cin in (64, 96, 128, 160):
    run(cin, 32, 'fwd', FWD32)
run(192, 64, 'fwd', ['', '64,64,1,2', '64,64,1,1', '128,64,1,1', '128,64,2,1', '64,64,2,1'])
for cin in (64, 96, 128, 160, 192):
    cout = 64 if cin == 192 else 32
    plans = ['', '128,64,1,1', '64,64,1,1', '64,64,1,2', '128,64,2,1', '64,64,2,1']
    if cin in (128, 192) or cin == 96:
        plans.append('128,128,1,1')
    run(cin, cout, 'dgrad', plans)
