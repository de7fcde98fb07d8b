"""Developer check: losses and parameter digests after a few GAN steps (compare a run with SRX_NO_OVERLAP=1 against one without:
the two-branch graph runs the same kernels on the same operands, so every printed value must match to the last bit).
    python tools/overlap_check.py [srgan|esrgan] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else 'srgan'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
torch.manual_seed(0)
if kind == 'srgan':
    from torchsr_amd.srgan.trainer import SRGANTrainer as T
    t = T(dev, bench._targs(16, False), [], [], 16, 16)
    lr, hr = bench._crops(dev, 16, 96, 78)
else:
    from torchsr_amd.esrgan.trainer import ESRGANTrainer as T
    t = T(dev, bench._targs(16, True), [], [], 16, 16)
    lr, hr = bench._crops(dev, 16, 128, 78)
for i in range(steps):
    losses = t.gan_step(lr, hr)
    torch.cuda.synchronize()
    print(i, {k: float(v).hex() for k, v in sorted(losses.items())})
for name, m in (('G', t.generator), ('D', t.discriminator)):
    tot = sum(float(p.double().sum()) for p in m.parameters())
    print(name, float(tot).hex())
