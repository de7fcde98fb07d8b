"""ESRGAN generator / discriminator / GAN step on the HIP path vs golden vectors captured from the
imported reference (tests/golden/esrgan.npz, written by oracle/gen_golden.py)."""
import os
import warnings
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.weights import closed_form_state, tensor_digest

pytestmark = pytest.mark.gpu
TOL = 1e-3


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


def digest_rel(t, ref):
    d = tensor_digest(t.detach().cpu())
    return max(abs(d[0] - ref[0]), abs(d[1] - ref[1]), abs(d[2] - ref[2])) / max(abs(ref[1]), 1e-12)


def test_esrgan_generator_vs_golden(dev):
    from torchsr_amd.esrgan.generator import Generator
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    gen = Generator(num_rrdb_blocks=2)
    gen.load_state_dict(closed_form_state(gen.state_dict()))
    gen = gen.to(dev).train()
    x = torch.from_numpy(gold['g_x']).to(dev).requires_grad_(True)
    y = gen(x)
    assert y.shape == (2, 3, 32, 40)
    assert rel(y, gold['g_y']) < TOL
    loss = y.square().mean()
    assert abs(loss.item() - float(gold['g_loss'])) < TOL * float(gold['g_loss'])
    loss.backward()
    assert rel(x.grad, gold['g_dx']) < TOL
    grads = dict(gen.named_parameters())
    for k, dg in zip(gold['g_grad_keys'], gold['g_grad_digest']):
        assert digest_rel(grads[str(k)].grad, dg) < TOL, k


def test_esrgan_discriminator_vs_golden(dev):
    from torchsr_amd import functional as F
    from torchsr_amd.esrgan.discriminator import Discriminator
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    disc = Discriminator(image_size=64)
    disc.load_state_dict(closed_form_state(disc.state_dict()))
    disc = disc.to(dev).train()
    x = torch.from_numpy(gold['d_x']).to(dev).requires_grad_(True)
    logits = disc(x)
    assert rel(logits, gold['d_logits']) < TOL
    loss = F.bce_with_logits(logits, 1.0, shift=torch.tensor(0.3, device=dev))
    assert abs(loss.item() - float(gold['d_loss'])) < TOL * float(gold['d_loss'])
    loss.backward()
    assert digest_rel(x.grad, gold['d_dx_digest']) < TOL
    grads = dict(disc.named_parameters())
    for k, dg in zip(gold['d_grad_keys'], gold['d_grad_digest']):
        assert digest_rel(grads[str(k)].grad, dg) < TOL, k


def test_esrgan_gan_steps_vs_reference_trainer(dev):
    from torchsr_amd.esrgan.trainer import ESRGANTrainer
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    args = Namespace(disable_amp=True, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t = ESRGANTrainer(dev, args, [], [], 2, 2, distributed=False)
    t.generator.load_state_dict(closed_form_state(t.generator.state_dict()))
    t.discriminator.load_state_dict(closed_form_state(t.discriminator.state_dict()))
    t.vgg_loss.features.load_state_dict(closed_form_state(t.vgg_loss.features.state_dict(), prefix='features.'))
    t.generator.train()
    t.discriminator.train()
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    for step in range(2):
        losses = t.gan_step(lr, hr)
        got = [losses[k].item() for k in ('gan/disc-loss', 'gan/pixel-loss', 'gan/content-loss',
                                           'gan/adversarial-loss', 'gan/train-loss')]
        want = gold['gan_losses'][step]
        tol = 1e-3 if step == 0 else 2e-2
        for g, w in zip(got, want):
            assert abs(g - w) <= tol * max(abs(w), 1e-3), (step, got, list(want))
        assert abs(got[4] - gold['gan_ref_gen_losses'][step]) <= tol * gold['gan_ref_gen_losses'][step]


def test_esrgan_gan_step_with_bf16_products(dev):
    """Without --disable-amp the trainers multiply bf16-rounded conv operands (fp32 accumulation, fp32
    everything else).  The first GAN step must then land within bf16 rounding of the fp32 golden losses:
    2e-2 relative (bf16 carries 8 bits of mantissa: 4e-3 per operand, deep networks compound it)."""
    from torchsr_amd.esrgan.trainer import ESRGANTrainer
    from torchsr_amd.layers import Conv2d
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    args = Namespace(disable_amp=False, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t = ESRGANTrainer(dev, args, [], [], 2, 2, distributed=False)
    assert all(m._st.precision == 1 for m in t.generator.modules() if isinstance(m, Conv2d))
    t.generator.load_state_dict(closed_form_state(t.generator.state_dict()))
    t.discriminator.load_state_dict(closed_form_state(t.discriminator.state_dict()))
    t.vgg_loss.features.load_state_dict(closed_form_state(t.vgg_loss.features.state_dict(), prefix='features.'))
    t.generator.train()
    t.discriminator.train()
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    losses = t.gan_step(lr, hr)
    got = [losses[k].item() for k in ('gan/disc-loss', 'gan/pixel-loss', 'gan/content-loss', 'gan/adversarial-loss',
                                       'gan/train-loss')]
    want = gold['gan_losses'][0]
    for g, w in zip(got, want):
        assert np.isfinite(g) and abs(g - w) <= 2e-2 * max(abs(w), 1e-3), (got, list(want))
    assert any(abs(g - w) > 1e-7 * max(abs(w), 1e-3) for g, w in zip(got, want))  # it really is a different precision
