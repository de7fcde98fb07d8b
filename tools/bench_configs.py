"""Measure the BASELINE.json configurations other than the headline one (developer tool; results are
quoted in DESIGN.md).

    python tools/bench_configs.py [pretrain] [esrgan] [infer]

  pretrain : SRGAN SRResNet pre-training step, 96x96 crops, batch 16 and batch 2 (config 1 on the GPU)
  srgan-amp: SRGAN full GAN step, batch 16, with bf16 conv products (the reference's default autocast mode)
  esrgan   : ESRGAN full GAN step, 128x128 crops, batch 16 (config 4), fp32 and with bf16 conv products (amp)
  infer    : SRGAN generator 1080p -> 8K, batch 1, eval mode, tiled (config 5)
"""
import os
import sys
import time
import warnings
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
dev = torch.device('cuda:0')


def targs(batch, amp=False):
    return Namespace(disable_amp=not amp, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=True, vgg_weights='random')


def timed(fn, steps, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def batch(n, crop):
    hr = torch.rand(n, 3, crop, crop)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', antialias=True).clamp(0, 1)
    return lr.to(dev), hr.to(dev)


which = set(sys.argv[1:]) or {'pretrain', 'esrgan', 'infer'}
if 'pretrain' in which:
    from torchsr_amd.srgan.trainer import SRGANTrainer
    for b in (16, 2):
        torch.manual_seed(0)
        t = SRGANTrainer(dev, targs(b), [], [], b, b)
        lr, hr = batch(b, 96)
        dt = timed(lambda: t.pretrain_step(lr, hr), 30)
        gf = 7.648 * b
        print(f'SRGAN pretrain step  batch {b:2d}: {dt * 1e3:7.3f} ms/step  {b / dt:8.1f} crops/s  {gf / dt / 1e3:6.1f} TFLOP/s', flush=True)
        del t
if 'srgan-amp' in which:
    from torchsr_amd.srgan.trainer import SRGANTrainer
    torch.manual_seed(0)
    t = SRGANTrainer(dev, targs(16, True), [], [], 16, 16)
    lr, hr = batch(16, 96)
    dt = timed(lambda: t.gan_step(lr, hr), 30)
    print(f'SRGAN GAN step       batch 16: {dt * 1e3:7.3f} ms/step  {16 / dt:8.1f} crops/s  {691.67 / dt / 1e3:6.1f} TFLOP/s '
          f'(bf16 products (amp))', flush=True)
    del t
if 'esrgan' in which:
    from torchsr_amd.esrgan.trainer import ESRGANTrainer
    torch.manual_seed(0)
    for amp in ((True,) if 'amp-only' in which else (False, True)):
        torch.manual_seed(0)
        t = ESRGANTrainer(dev, targs(16, amp), [], [], 16, 16)
        lr, hr = batch(16, 128)
        dt = timed(lambda: t.gan_step(lr, hr), 10, warm=4)
        what = 'bf16 products (amp)' if amp else 'fp32'
        # executed work: the reference's 3622 GFLOP minus its second generator forward (587.4), which is not run
        print(f'ESRGAN GAN step      batch 16: {dt * 1e3:7.3f} ms/step  {16 / dt:8.1f} crops/s  {3034.6 / dt / 1e3:6.1f} TFLOP/s '
              f'({what}, 3035 GFLOP executed per step)', flush=True)
        del t
if 'infer' in which:
    from torchsr_amd.srgan.generator import Generator
    from torchsr_amd.test import upscale
    torch.manual_seed(0)
    gen = Generator().to(dev).eval()
    lr = torch.rand(1, 3, 1080, 1920, device=dev)
    dt = timed(lambda: upscale(gen, lr), 3, warm=1)
    print(f'SRGAN 1080p -> 8K    batch  1: {dt * 1e3:7.1f} ms/image  {9199.0 / dt / 1e3:6.1f} TFLOP/s (tiled, eager launches)',
          flush=True)
