"""Shared by the developer tools: trainer arguments and a synthetic batch."""
from argparse import Namespace

import torch


def targs(batch, amp=False):
    return Namespace(disable_amp=not amp, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=True, vgg_weights='random')


def batch(n, crop, dev='cuda:0'):
    hr = torch.rand(n, 3, crop, crop)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', antialias=True).clamp(0, 1)
    return lr.to(dev), hr.to(dev)
