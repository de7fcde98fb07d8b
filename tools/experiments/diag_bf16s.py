"""Where do the bf16-storage and the fp32-storage forms of the frozen VGG19 stack part?  Truncated stacks, output by output."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from torchsr_amd import _dev, functional as F  # noqa: E402
from torchsr_amd.layers import set_conv_precision  # noqa: E402
from torchsr_amd.srgan.loss import VGGLoss  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
vgg = VGGLoss(weights='random').to(dev)
set_conv_precision(vgg, 'bf16')
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
src = F.to_nhwc(torch.rand(2, 3, size, size, device=dev), 4)
tgt = F.to_nhwc(torch.rand(2, 3, size, size, device=dev), 4)
layers = vgg._stack()
for k in range(2, len(layers) + 1):
    if layers[k - 1][0] != 'conv':
        continue
    outs = []
    for off in (False, True):
        _dev.NO_BF16S = off
        with torch.no_grad():
            fs, ft = F.frozen_conv_stack(src, tgt, layers[:k])
        outs.append(torch.cat([fs, ft]).double())
    a, b = outs
    print(k, tuple(a.shape), 'max rel %.3e  rms rel %.3e' % (((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()))
