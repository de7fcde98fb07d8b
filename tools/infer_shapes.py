"""Per-(kernel, shape) device time of one 1080p -> 8K inference (developer tool): python tools/infer_shapes.py [fp32|bf16]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from torchsr_amd.srgan.generator import Generator  # noqa: E402
from torchsr_amd.test import upscale  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
device = torch.device('cuda:0')
torch.manual_seed(0)
gen = Generator().to(device).eval()
lr = torch.rand(1, 3, 1080, 1920, device=device)
kw = {} if precision == 'fp32' else {'precision': precision}
upscale(gen, lr, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    upscale(gen, lr, **kw)
torch.cuda.synchronize()
print(f'{precision}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per image')
pairs, _ = bench.prof_tables(lambda: upscale(gen, lr, **kw), reps=1, slots=8192)
tot = sum(v[0] for v in pairs.values())
for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
    print(f'{v[0]:8.3f} ms {v[2]:4d} launches {v[0] / v[2] * 1e3:9.1f} us each {v[1] / (v[0] * 1e-3) / 1e12:7.1f} TF/s  {k}')
print(f'profiled conv kernels: {tot:.2f} ms')
