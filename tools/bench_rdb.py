"""srx_rdb_fwd alone: 69 dependent launches (one generator forward's dense blocks) at batch 16, 32x32 (developer tool)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd import _lib  # noqa: E402

dev = torch.device('cuda:0')
L = _lib.lib()
n, h, w, nb = 16, 32, 32, 69
ws = [torch.randn(32 if k < 4 else 64, 64 + 32 * k, 3, 3, device=dev) * 0.02 for k in range(5)]
bs = [torch.zeros(32 if k < 4 else 64, device=dev) for k in range(5)]
table = torch.tensor([t.data_ptr() for t in ws] * nb, dtype=torch.int64).to(dev)
per = L.srx_rdb_packed_bytes()
pk = torch.empty(per * nb, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream().cuda_stream
_lib.call('srx_rdb_pack', table.data_ptr(), nb, pk.data_ptr(), s)
bufs = [torch.randn(n, h, w, 192, device=dev) for _ in range(4)]
biases = (C.c_void_p * 5)(*[t.data_ptr() for t in bs])


pkb = torch.empty(per * nb, dtype=torch.uint8, device=dev)
_lib.call('srx_rdb_pack_bwd', table.data_ptr(), nb, pkb.data_ptr(), s)
gbufs = [torch.empty(n, h, w, 192, device=dev) for _ in range(2)]
dys = [torch.randn(n, h, w, 64, device=dev) for _ in range(2)]


def chain_bwd():
    s = torch.cuda.current_stream().cuda_stream
    for i in range(nb):
        _lib.call('srx_rdb_bwd', n, h, w, dys[i % 2].data_ptr(), 64, 0.2, bufs[i % 4].data_ptr(), 192, pkb.data_ptr() + i * per, 0.2,
                  gbufs[i % 2].data_ptr(), 192, dys[i % 2].data_ptr(), 64, 1.0, None, 0, dys[(i + 1) % 2].data_ptr(), 64, s)


def chain():
    s = torch.cuda.current_stream().cuda_stream  # (the capture stream inside torch.cuda.graph)
    for i in range(nb):
        _lib.call('srx_rdb_fwd', n, h, w, bufs[i % 4].data_ptr(), 192, pk.data_ptr() + i * per, biases, 0.2, 0.2, 1.0, None, 0,
                  bufs[(i + 1) % 4].data_ptr(), 192, s)


if len(sys.argv) > 1 and sys.argv[1] == 'stamps':
    # in-kernel clock stamps (developer entry point, bound here only): where a block's 25 us go, per weight unit
    import numpy as np
    fn = L.srx_rdb_fwd_dbg
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 3 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    dbg = torch.zeros(256 * 8 * 48, dtype=torch.int32, device=dev)
    for i in range(6):
        rc = fn(n, h, w, bufs[i % 4].data_ptr(), 192, pk.data_ptr() + i * per, biases, 0.2, 0.2, bufs[(i + 1) % 4].data_ptr(), 192, s, dbg.data_ptr())
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    d = (dbg.view(256, 8, 48).cpu().numpy().astype('int64') & 0xffffffff)
    t0 = d[:, :, 0].min(axis=1, keepdims=True)[:, :, None]
    d = (d - t0) & 0xffffffff
    comp, load = d[:, :4, :], d[:, 4:, :]
    print('stamps in s_memtime ticks (100 MHz -> 10 ns each), mean over the 256 workgroups')
    print('patch staged at %.0f (compute waves), %.0f (loader waves); block ends at %.0f' % (comp[:, :, 1].mean(), load[:, :, 1].mean(), comp[:, :, 41].max(axis=1).mean()))
    print('unit conv  open  | compute waves: work mean / max wave, wait at next barrier | loader: issue+wait')
    for u in range(20):
        conv = 1 if u < 2 else 2 if u < 5 else 3 if u < 9 else 4 if u < 14 else 5
        opened = comp[:, :, 2 + 2 * u].mean()
        work = comp[:, :, 3 + 2 * u] - comp[:, :, 2 + 2 * u]
        nxt = (comp[:, :, 4 + 2 * u] - comp[:, :, 3 + 2 * u]) if u < 19 else np.zeros_like(work)
        lw = load[:, :, 3 + 2 * u] - load[:, :, 2 + 2 * u]
        print('%4d %4d %6.0f | %6.1f / %6.1f   %6.1f | %6.1f' % (u, conv, opened, work.mean(), work.max(axis=1).mean(), nxt.mean(), lw.mean()))
    sys.exit(0)

gf = 2.0 * n * h * w * 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) / 1e9
for name, fn in (('rdb_fwd', chain), ('rdb_bwd', chain_bwd)):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / nb)
    ts.sort()
    print(f'{name}: {ts[len(ts) // 2]:.2f} us per block in-graph (min {ts[0]:.2f}); {gf:.3f} GFLOP -> {gf / ts[len(ts) // 2] * 1e3:.1f} TFLOP/s')

