"""Developer aid: the ESRGAN GAN step's conv launches by (kernel, layer shape) from one instrumented eager pass (srx_prof_*), on ONE stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from torchsr_amd.esrgan.trainer import ESRGANTrainer  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
t = ESRGANTrainer(dev, bench._targs(16, True), [], [], 16, 16)
lr, hr = bench._crops(dev, 16, 128, 78)
t.use_graphs = t.overlap_branches = False
pairs, _ = bench.prof_tables(lambda: t.gan_step(lr, hr), reps=1, slots=8192)
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
    print(f'{v[0] * 1e3:9.1f} us/step {v[2]:3d} launches {v[0] / v[2] * 1e3:8.1f} us each {v[1] / (v[0] * 1e-3) / 1e12:7.1f} TF/s  {k}')
