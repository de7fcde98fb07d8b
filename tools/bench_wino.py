"""Winograd F(2x2, 3x3) (csrc/wino.hip) against the direct gather-GEMM (gconv.hip) on the VGG19 layer shapes of the SRGAN step
(forward at batch 32, data gradient at batch 16), each as 10 back-to-back launches inside a replayed hipGraph (developer tool).
    python3 tools/bench_wino.py [fwd|bwd]          SRX_WINO_ZSPLIT / SRX_WINO_BN override the planner"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
LAYERS = [(96, 64, 64), (48, 64, 128), (48, 128, 128), (24, 128, 256), (24, 256, 256), (12, 256, 512), (12, 512, 512), (6, 512, 512)]
which = sys.argv[1] if len(sys.argv) > 1 else 'both'


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


for mode, n in (('fwd', 32), ('bwd', 16)):
    if which not in ('both', mode):
        continue
    print(f'== {mode}, batch {n}')
    tot_w = tot_d = 0.0
    for hw, cin, cout in LAYERS:
        d = _lib.Conv2dDesc(n, hw, hw, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 0)
        dref = C.byref(d)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        bias = torch.zeros(cout, device=dev)
        x = torch.rand(n, hw, hw, cin, device=dev)
        dy = torch.randn(n, hw, hw, cout, device=dev)
        y = torch.empty(n, hw, hw, cout, device=dev)
        dx = torch.empty(n, hw, hw, cin, device=dev)
        nf = L.srx_wino_packed_floats(dref)
        uf, ub = torch.empty(nf, device=dev), torch.empty(nf, device=dev)
        s0 = torch.cuda.current_stream().cuda_stream
        _lib.call('srx_wino_pack', dref, w.data_ptr(), uf.data_ptr(), 0, s0)
        _lib.call('srx_wino_pack', dref, w.data_ptr(), ub.data_ptr(), 1, s0)
        wf = torch.empty(L.srx_conv2d_packed_fwd_floats(dref), device=dev)
        wb = torch.empty(max(L.srx_conv2d_packed_bwd_floats(dref), 4), device=dev)
        _lib.call('srx_conv2d_pack', dref, w.data_ptr(), wf.data_ptr(), wb.data_ptr(), s0)
        k = 0 if mode == 'fwd' else 1
        nws = L.srx_wino_ws_floats(dref, k)
        ws = torch.empty(max(nws, 4), device=dev)
        ndw = L.srx_conv2d_fwd_ws_floats(dref) if mode == 'fwd' else L.srx_conv2d_bwd_data_ws_floats(dref)
        dws = torch.empty(max(ndw, 4), device=dev)
        plan = (C.c_int * 6)()
        _lib.call('srx_wino_plan', dref, k, plan)

        def wino():
            s = torch.cuda.current_stream().cuda_stream
            if mode == 'fwd':
                _lib.call('srx_wino_fwd', dref, x.data_ptr(), uf.data_ptr(), bias.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, s)
            else:
                _lib.call('srx_wino_bwd_data', dref, dy.data_ptr(), ub.data_ptr(), x.data_ptr(), dx.data_ptr(), ws.data_ptr(), nws, s)

        def direct():
            s = torch.cuda.current_stream().cuda_stream
            if mode == 'fwd':
                _lib.call('srx_conv2d_fwd', dref, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), None, dws.data_ptr(), ndw, s)
            else:
                _lib.call('srx_conv2d_bwd_data_act', dref, dy.data_ptr(), wb.data_ptr(), x.data_ptr(), 0.0, 0, cin, 0, dx.data_ptr(),
                          dws.data_ptr(), ndw, s)

        tw, td = timed(wino), timed(direct)
        gf = 2.0 * n * hw * hw * cout * 9 * cin / 1e9
        mult = {96: 1, 48: 1, 24: 1, 12: 1, 6: 1}[hw]
        reps = {(96, 64, 64): 1, (48, 64, 128): 1, (48, 128, 128): 1, (24, 128, 256): 1, (24, 256, 256): 3, (12, 256, 512): 1,
                (12, 512, 512): 3, (6, 512, 512): 4}[(hw, cin, cout)]
        tot_w += reps * min(tw, td) if False else reps * tw
        tot_d += reps * td
        print(f'{hw:3d}x{hw:<3d} {cin:3d}->{cout:<3d} {gf:6.2f} GF | wino {tw:7.1f} us ({gf / tw * 1e3:6.1f} TF)  plan BN {plan[0]} split {plan[1]} '
              f'wgs {plan[2]} chunks {plan[4]} | direct {td:7.1f} us ({gf / td * 1e3:6.1f} TF) | x{td / tw:.2f}')
    print(f'VGG19[:36] {mode}: wino {tot_w:.0f} us, direct {tot_d:.0f} us (layers weighted by their count in cfg E)')
