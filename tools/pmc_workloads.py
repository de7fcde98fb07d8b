"""Single-kernel workloads for rocprofv3 --pmc passes (developer tool; the summaries go to profiles/).

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_workloads.py <name>

  rt36     : the north-star kernel, 3x3 64->64 forward at 16x24x24 (66 launches per GAN step)
  wgrad33  : the grouped weight gradient of the generator's 33 residual convs (one launch per step)
  wgrad256 : the weight gradient of the discriminator's 128->256 conv at 32x24x24 (M 18432 N 256 K 1152)
  vgg256   : 3x3 256->256 at 32x24x24 -- the VGG19 block-3 layers with source and target as one batch
  vgg256h  : the same at 16x24x24 (its data gradient: source half only)
  vgg512   : 3x3 512->512 at 32x12x12 (VGG19 block 4, M 4608 N 512 K 4608);  vgg512h: at 16x12x12 (M 2304)
  vgg128   : 3x3 64->128 at 32x48x48 (VGG19 block 2's first layer, M 73728 N 128 K 576)
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd import _lib  # noqa: E402
from torchsr_amd.layers import Conv2d  # noqa: E402

dev = torch.device('cuda:0')
name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
torch.manual_seed(0)
SHAPES = {'rt36': (16, 24, 24, 64, 64), 'vgg256': (32, 24, 24, 256, 256), 'vgg256h': (16, 24, 24, 256, 256),
          'vgg512': (32, 12, 12, 512, 512), 'vgg512h': (16, 12, 12, 512, 512), 'vgg128': (32, 48, 48, 64, 128),
          # round 4: the other GEMM shapes the headline step runs on the dominant kernel (stride-1 stand-ins of the same M x N x K)
          's18432x256x1152': (32, 24, 24, 128, 256), 's73728x128x1152': (32, 48, 48, 128, 128),
          's36864x256x576': (16, 48, 48, 64, 256, 2), 's9216x128x2304': (16, 24, 24, 256, 128),
          's36864x128x1152': (16, 48, 48, 128, 128), 's4608x256x4608': (32, 12, 12, 512, 256),
          's36864x128x576': (16, 48, 48, 64, 128), 's4608x256x2304': (32, 12, 12, 256, 256),
          's2304x512x2304': (16, 12, 12, 256, 512)}
if name in SHAPES:
    n, h, w, cin, cout = SHAPES[name][:5]
    shuffle = SHAPES[name][5] if len(SHAPES[name]) > 5 else 0
    # (round 5: bias + ReLU as the VGG19 layers have them -- the wide stride-1 layers then take the step's Winograd path)
    conv = Conv2d(cin, cout, 3, 1, 1, bias=not shuffle, act=0 if shuffle else 1, shuffle=shuffle).to(dev)
    x = torch.rand(n, h, w, cin, device=dev)
    with torch.no_grad():
        for _ in range(reps):
            conv(x)
elif name in ('wgrad33', 'wgrad256'):
    nprob, (n, h, w, cin, cout) = (33, (16, 24, 24, 64, 64)) if name == 'wgrad33' else (1, (32, 24, 24, 128, 256))
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    xs = [torch.rand(n, h, w, cin, device=dev) for _ in range(nprob)]
    dys = [torch.rand(n, h, w, cout, device=dev) for _ in range(nprob)]
    dws = [torch.zeros(cout, cin, 3, 3, device=dev) for _ in range(nprob)]
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    nws = _lib.lib().srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), nprob)
    ws = torch.empty(nws, device=dev)
    for _ in range(reps):
        _lib.call('srx_conv2d_bwd_weight_multi', C.byref(d), nprob, 1, arr(xs), arr(dys), arr(dws), 1, None, ws.data_ptr(), nws,
                  torch.cuda.current_stream().cuda_stream)
else:
    sys.exit(f'unknown workload {name}')
torch.cuda.synchronize()
print('done', name)
