"""BASELINE config 5 (1080p -> 8K generator forward) in fp32 / bf16, with a per-kernel table from the srx_prof_* pass."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device('cuda:0')
from torchsr_amd.srgan.generator import Generator
from torchsr_amd.test import upscale

torch.manual_seed(0)
gen = Generator().to(dev).eval()
lr = torch.rand(1, 3, 1080, 1920, device=dev)
for prec in sys.argv[1:] or ['bf16']:
    kw = {} if prec == 'fp32' else {'precision': prec}
    dt = bench._timed(lambda: upscale(gen, lr, **kw), 5, 2)
    print(f'{prec}: {dt * 1e3:.2f} ms per image', flush=True)
    pairs, kernels = bench.prof_tables(lambda: upscale(gen, lr, **kw), reps=1, slots=8192)
    for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
        print(f'  {v[0]:8.3f} ms {v[2]:3d} launches {v[0] / v[2] * 1e3:9.1f} us each {v[1] / (v[0] * 1e-3) / 1e12:7.1f} TF/s  {k}')
