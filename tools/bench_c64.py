"""Times srx_conv3x3_c64_bf16_fwd (the bf16-native 64-channel conv) on BASELINE config 5's layer shapes, back to back on one
stream with HIP events around the batch: trunk conv 64 -> 64 at 1080 x 1920 (with and without the skip addend), the two
sub-pixel layers (64 -> 256 + PixelShuffle) at 1080p and at 2160 x 3840 (whole frame: 8K feature map, 4.2 GB of bf16)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchsr_amd import _lib

dev = torch.device('cuda:0')
L = _lib.lib()
s = torch.cuda.current_stream().cuda_stream


def run(n, h, w, cout, shuffle, res, reps=20):
    x = (torch.rand(n, h, w, 64, device=dev) - 0.5).bfloat16()
    wt = torch.randn(cout, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
    b = torch.randn(cout, device=dev) * 0.1
    pk = torch.empty(L.srx_conv3x3_c64_bf16_packed_bytes(cout), dtype=torch.uint8, device=dev)
    _lib.call('srx_conv3x3_c64_bf16_pack', wt.data_ptr(), b.data_ptr(), None, cout, shuffle, pk.data_ptr(), s)
    oh, ow, oc = (2 * h, 2 * w, 64) if shuffle else (h, w, cout)
    y = torch.empty((n, oh, ow, oc), dtype=torch.bfloat16, device=dev)
    r = (torch.rand(n, oh, ow, oc, device=dev) - 0.5).bfloat16() if res else None

    def call():
        _lib.call('srx_conv3x3_c64_bf16_fwd', n, h, w, cout, shuffle, x.data_ptr(), pk.data_ptr(), 0.25,
                  None if r is None else r.data_ptr(), y.data_ptr(), oc, s)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gf = 2.0 * n * h * w * cout * 576 / 1e9
    byt = n * h * w * 128 + n * oh * ow * oc * 2 * (2 if res else 1)
    print(f'{n}x{h}x{w} 64->{cout} shuffle={shuffle} res={int(res)}: {us:9.1f} us  {gf / us * 1e3:8.1f} TFLOP/s  '
          f'{byt / us / 1e3:7.1f} GB/s algorithmic', flush=True)


def stamps(n, h, w, cout, shuffle, res):
    """In-kernel cycle stamps per phase (developer entry point, bound here only)."""
    import ctypes as C
    fn = L.srx_conv3x3_c64_bf16_fwd_dbg
    fn.restype = C.c_int
    fn.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    x = (torch.rand(n, h, w, 64, device=dev) - 0.5).bfloat16()
    wt = torch.randn(cout, 64, 3, 3, device=dev) * (2.0 / 576) ** 0.5
    pk = torch.empty(L.srx_conv3x3_c64_bf16_packed_bytes(cout), dtype=torch.uint8, device=dev)
    _lib.call('srx_conv3x3_c64_bf16_pack', wt.data_ptr(), None, None, cout, shuffle, pk.data_ptr(), s)
    oh, ow, oc = (2 * h, 2 * w, 64) if shuffle else (h, w, cout)
    y = torch.empty((n, oh, ow, oc), dtype=torch.bfloat16, device=dev)
    r = (torch.rand(n, oh, ow, oc, device=dev) - 0.5).bfloat16() if res else None
    dbg = torch.zeros(1024 * 8 * 8, dtype=torch.int32, device=dev)
    for _ in range(5):
        rc = fn(n, h, w, cout, shuffle, x.data_ptr(), pk.data_ptr(), 0.25, None if r is None else r.data_ptr(), y.data_ptr(), oc, s,
                dbg.data_ptr())
        assert rc == 0
    torch.cuda.synchronize()
    d = dbg.view(-1, 8, 8).cpu().numpy().astype('int64') & 0xffffffff
    live = d[:, 0, 3] > 0
    m, hlp = d[live][:, :4, :], d[live][:, 4:, :]
    steps = m[:, :, 3].mean()
    print(f'{int(live.sum())} workgroups, {steps:.1f} steps each (cycles per step, mean over waves)')
    print('  matrix waves: barrier %.0f  mfma loop %.0f  dump %.0f  total %.0f' % tuple(m[:, :, i].mean() / steps for i in (0, 1, 2, 4)))
    print('  helper waves: barrier %.0f  requests %.0f  addend wait %.0f  finishing %.0f  final wait %.0f  total %.0f'
          % tuple(hlp[:, :, i].mean() / steps for i in (0, 1, 2, 3, 4, 5)))


if len(sys.argv) > 1 and sys.argv[1] == 'stamps':
    stamps(1, 1080, 1920, 64, 0, False)
    stamps(1, 1080, 1920, 64, 0, True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == 'trunk':  # one shape (profiler passes)
    run(1, 1080, 1920, 64, 0, False, reps=10)
    sys.exit(0)
run(1, 1080, 1920, 64, 0, False)
run(1, 1080, 1920, 64, 0, True)
run(1, 1080, 1920, 256, 2, False)
run(1, 2160, 3840, 256, 2, False, reps=5)
run(16, 24, 24, 64, 0, True, reps=50)
run(16, 32, 32, 64, 0, False, reps=50)
run(16, 96, 96, 64, 0, False, reps=50)
