"""SRGAN GAN step (the headline: 96x96 crops, batch 16, fp32) replayed N times, for rocprofv3 (developer tool):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/srgan_step.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchsr_amd.srgan.trainer import SRGANTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
torch.manual_seed(0)
t = SRGANTrainer(dev, bench._targs(16, False), [], [], 16, 16)
lr, hr = bench._crops(dev, 16, 96, 78)
for _ in range(5):
    t.gan_step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    t.gan_step(lr, hr)
torch.cuda.synchronize()
print(f'srgan fp32 b16: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step over {steps} steps (+5 set-up/warm-up)')
