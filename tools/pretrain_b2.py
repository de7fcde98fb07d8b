"""SRGAN pre-training step at batch 2 (BASELINE configs[0]) replayed N times, for rocprofv3 --kernel-trace --stats (developer tool)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchsr_amd.srgan.trainer import SRGANTrainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
torch.manual_seed(0)
t = SRGANTrainer(dev, bench._targs(2, False), [], [], 2, 2)
lr, hr = bench._crops(dev, 2, 96, 77)
for _ in range(6):
    t.pretrain_step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    t.pretrain_step(lr, hr)
torch.cuda.synchronize()
print(f'srgan pretrain b2: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step over {steps} steps')
