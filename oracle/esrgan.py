"""ORACLE (test infrastructure, never the product path).

CPU restatement of the ESRGAN hot path of roclark/torchsr in plain ``torch.nn.functional`` calls
on NCHW fp32 tensors; state dicts keyed like the reference modules.  Pinned by
``oracle/gen_golden.py`` against the imported reference (``tests/golden/esrgan_*.npz``).
"""
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from .srgan import Adam, State, _bn, _leaves, conv2d, vgg_loss


def residual_dense_block(sd: State, p: str, x: Tensor, scale_ratio: float = 0.2) -> Tensor:
    """ResidualDenseBlock.forward, torchsr/esrgan/residual.py:65-86."""
    def conv(i, inp, act=True):
        k = f'{p}conv{i}.0.' if i < 5 else f'{p}conv5.'
        out = conv2d(inp, sd[k + 'weight'], sd[k + 'bias'], 1, 1)
        return F.leaky_relu(out, 0.2) if act else out
    conv1 = conv(1, x)                                              # :81
    conv2 = conv(2, torch.cat((x, conv1), dim=1))                   # :82
    conv3 = conv(3, torch.cat((x, conv1, conv2), dim=1))            # :83
    conv4 = conv(4, torch.cat((x, conv1, conv2, conv3), dim=1))     # :84
    conv5 = conv(5, torch.cat((x, conv1, conv2, conv3, conv4), dim=1), act=False)  # :85
    return conv5 * scale_ratio + x                                  # :86


def rrdb(sd: State, p: str, x: Tensor) -> Tensor:
    """ResidualInResidualDenseBlock.forward, torchsr/esrgan/residual.py:110-129."""
    out = residual_dense_block(sd, p + 'RDB1.', x)
    out = residual_dense_block(sd, p + 'RDB2.', out)
    out = residual_dense_block(sd, p + 'RDB3.', out)
    return out * 0.2 + x


def generator_forward(sd: State, x: Tensor, prefix: str = '') -> Tensor:
    """Generator.forward, torchsr/esrgan/generator.py:54-81."""
    p = prefix
    conv1 = conv2d(x, sd[p + 'conv1.weight'], sd[p + 'conv1.bias'], 1, 1)        # :69
    n_blocks = len({k.split('.')[1] for k in sd if k.startswith(p + 'blocks.')})
    block = conv1
    for i in range(n_blocks):                                                        # :70
        block = rrdb(sd, f'{p}blocks.{i}.', block)
    conv2 = conv2d(block, sd[p + 'conv2.weight'], sd[p + 'conv2.bias'], 1, 1)    # :71
    out = torch.add(conv1, conv2)                                                    # :72
    for name in ('upsample1', 'upsample2'):                                          # :73-78
        out = F.interpolate(out, scale_factor=2, mode='nearest')
        out = F.leaky_relu(conv2d(out, sd[f'{p}{name}.weight'], sd[f'{p}{name}.bias'], 1, 1), 0.2)
    out = F.leaky_relu(conv2d(out, sd[p + 'conv3.0.weight'], sd[p + 'conv3.0.bias'], 1, 1), 0.2)  # :79
    return conv2d(out, sd[p + 'conv4.weight'], sd[p + 'conv4.bias'], 1, 1)        # :80


D_CONVS = [(2, 3, 2), (5, 6, 1), (8, 9, 2), (11, 12, 1), (14, 15, 2), (17, 18, 1), (20, 21, 2), (23, 24, 1),
           (26, 27, 2)]  # (conv index, bn index, stride), torchsr/esrgan/discriminator.py:34-71


def discriminator_forward(sd: State, x: Tensor, training: bool = True, prefix: str = '') -> Tensor:
    """Discriminator.forward, torchsr/esrgan/discriminator.py:78-95 (returns logits)."""
    p = prefix
    out = F.leaky_relu(conv2d(x, sd[p + 'features.0.weight'], sd[p + 'features.0.bias'], 1, 1), 0.2)
    for ci, bi, stride in D_CONVS:
        out = conv2d(out, sd[f'{p}features.{ci}.weight'], None, stride, 1)
        out = F.leaky_relu(_bn(sd, f'{p}features.{bi}.', out, training), 0.2)
    out = torch.flatten(out, 1)
    out = F.leaky_relu(F.linear(out, sd[p + 'classifier.0.weight'], sd[p + 'classifier.0.bias']), 0.2)
    return F.linear(out, sd[p + 'classifier.2.weight'], sd[p + 'classifier.2.bias'])


class ESRGANStepOracle:
    """Inner-loop bodies of ESRGANTrainer on CPU (AMP is inactive on CPU / disabled).

    ``pretrain_step``: torchsr/esrgan/trainer.py:378-390; ``gan_step``: ``_gan_loop`` :418-484.
    """

    def __init__(self, g_sd: State, d_sd: State, vgg_sd: State):
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.vgg = {k: v.clone() for k, v in vgg_sd.items()}
        g_params, d_params = _leaves(self.g), _leaves(self.d)
        self.psnr_optimizer = Adam(g_params)
        self.disc_optimizer = Adam(d_params)
        self.gen_optimizer = Adam(g_params)

    def pretrain_step(self, low_res: Tensor, high_res: Tensor) -> float:
        self.psnr_optimizer.zero_grad()                               # :382
        loss = F.l1_loss(generator_forward(self.g, low_res), high_res)  # :385-386
        loss.backward()
        self.psnr_optimizer.step()
        return float(loss.detach())

    def gan_step(self, low_res: Tensor, high_res: Tensor) -> Tuple[float, float, float, float, float]:
        bce = F.binary_cross_entropy_with_logits
        batch = low_res.size(0)
        real_label = torch.full((batch, 1), 1, dtype=low_res.dtype)   # :441
        fake_label = torch.full((batch, 1), 0, dtype=low_res.dtype)   # :442
        self.disc_optimizer.zero_grad()                               # :444
        super_res = generator_forward(self.g, low_res)                # :447
        real_output = discriminator_forward(self.d, high_res, True)   # :448
        fake_output = discriminator_forward(self.d, super_res.detach(), True)  # :449
        d_real = bce(real_output - torch.mean(fake_output), real_label)       # :451
        d_fake = bce(fake_output - torch.mean(real_output), fake_label)       # :452
        disc_loss = (d_real + d_fake) / 2                             # :453
        disc_loss.backward()                                          # :455
        self.disc_optimizer.step()                                    # :456
        self.gen_optimizer.zero_grad()                                # :459
        super_res = generator_forward(self.g, low_res)                # :462
        real_output = discriminator_forward(self.d, high_res.detach(), True)  # :463
        fake_output = discriminator_forward(self.d, super_res, True)  # :464
        pixel = F.l1_loss(super_res, high_res.detach())               # :466
        content = vgg_loss(self.vgg, super_res, high_res.detach())    # :467
        adversarial = bce(fake_output - torch.mean(real_output), real_label)  # :468
        gen_loss = 0.01 * pixel + 1 * content + 0.005 * adversarial   # :469
        gen_loss.backward()                                           # :480
        self.gen_optimizer.step()                                     # :481
        return tuple(float(v.detach()) for v in (disc_loss, pixel, content, adversarial, gen_loss))


class ESRGANDataParallelOracle:
    """``world`` data-parallel replicas of ``ESRGANTrainer``'s loop bodies in ONE process: what
    ``DistributedDataParallel`` does to them (torchsr/esrgan/trainer.py:142-157, same wrapping as SRGAN's): every rank runs
    the body on its own shard with identical weights, gradients are averaged over the ranks before each optimiser step,
    the discriminator's BatchNorm is NOT synchronised (per-rank batch statistics and running buffers; the generator has
    none).  One set of weight leaves is shared, so the mean of the per-rank losses back-propagates the averaged gradient.
    The relativistic means (:451-452,468) are per rank, as each rank computes them from its own logits."""

    def __init__(self, g_sd: State, d_sd: State, vgg_sd: State, world: int):
        self.world = world
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.vgg = {k: v.clone() for k, v in vgg_sd.items()}
        g_params, d_params = _leaves(self.g), _leaves(self.d)
        self.psnr_optimizer, self.disc_optimizer, self.gen_optimizer = Adam(g_params), Adam(d_params), Adam(g_params)
        self.d_ranks = [self.d] + [{k: (v if v.requires_grad else v.clone()) for k, v in self.d.items()}
                                   for _ in range(1, world)]

    def pretrain_step(self, low_res, high_res):
        self.psnr_optimizer.zero_grad()
        losses = [F.l1_loss(generator_forward(self.g, low_res[r]), high_res[r]) for r in range(self.world)]
        (sum(losses) / self.world).backward()
        self.psnr_optimizer.step()
        return [float(v.detach()) for v in losses]

    def gan_step(self, low_res, high_res):
        bce, W = F.binary_cross_entropy_with_logits, self.world
        ones = [torch.full((low_res[r].size(0), 1), 1.0) for r in range(W)]
        zeros = [torch.full((low_res[r].size(0), 1), 0.0) for r in range(W)]
        self.disc_optimizer.zero_grad()
        sr = [generator_forward(self.g, low_res[r]) for r in range(W)]
        disc = []
        for r in range(W):
            real_output = discriminator_forward(self.d_ranks[r], high_res[r], True)
            fake_output = discriminator_forward(self.d_ranks[r], sr[r].detach(), True)
            d_real = bce(real_output - torch.mean(fake_output), ones[r])
            d_fake = bce(fake_output - torch.mean(real_output), zeros[r])
            disc.append((d_real + d_fake) / 2)
        (sum(disc) / W).backward()
        self.disc_optimizer.step()
        self.gen_optimizer.zero_grad()
        gen, parts = [], []
        for r in range(W):
            real_output = discriminator_forward(self.d_ranks[r], high_res[r].detach(), True)
            fake_output = discriminator_forward(self.d_ranks[r], sr[r], True)
            pixel = F.l1_loss(sr[r], high_res[r].detach())
            content = vgg_loss(self.vgg, sr[r], high_res[r].detach())
            adversarial = bce(fake_output - torch.mean(real_output), ones[r])
            gen.append(0.01 * pixel + 1 * content + 0.005 * adversarial)
            parts.append(tuple(float(v.detach()) for v in (disc[r], pixel, content, adversarial, gen[-1])))
        (sum(gen) / W).backward()
        self.gen_optimizer.step()
        return parts
