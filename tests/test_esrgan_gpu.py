"""ESRGAN generator / discriminator / GAN step on the HIP path vs golden vectors captured from the
imported reference (tests/golden/esrgan.npz, written by oracle/gen_golden.py)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle.weights import closed_form_state, seeded_input, step_state, tensor_digest

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


LOSS_KEYS = ('gan/disc-loss', 'gan/pixel-loss', 'gan/content-loss', 'gan/adversarial-loss', 'gan/train-loss')


def make_trainer(dev, batch=2, disable_amp=True, use_graphs=False):
    from torchsr_amd.esrgan.trainer import ESRGANTrainer
    args = Namespace(disable_amp=disable_amp, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0,
                     pretrain_epochs=1, psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1,
                     use_graphs=use_graphs, vgg_weights='random')
    t = ESRGANTrainer(dev, args, [], [], batch, batch, distributed=False)
    t.generator.load_state_dict(step_state(t.generator.state_dict(), 'esrgan.G'))
    t.discriminator.load_state_dict(step_state(t.discriminator.state_dict(), 'esrgan.D'))
    t.vgg_loss.features.load_state_dict(closed_form_state(t.vgg_loss.features.state_dict(), prefix='features.'))
    t.generator.train()
    t.discriminator.train()
    return t


def assert_digests(keys, sd, digests, what):
    for k, dg in zip(keys, digests):
        if str(k) == 'classifier.2.bias':
            # the relativistic losses only see logit DIFFERENCES (real - mean(fake), fake - mean(real)), in which
            # the last layer's bias cancels: its exact gradient is 0, what any fp32 implementation computes is
            # rounding noise, and Adam turns noise into +-lr steps.  Nothing to pin.
            continue
        d = tensor_digest(sd[str(k)].cpu())
        # digests are sums over the tensor.  Every element moves by ~lr = 1e-4 per Adam step, and the few whose
        # gradient sits at the fp32 noise floor move in a direction that differs between any two fp32
        # implementations (the elementwise test below allows 0.2 % of a tensor to do so): each costs 2e-4 of a sum
        slack = 2e-4 * max(abs(dg[1]), 1e-6) + 1e-6 + max(2, 2e-3 * sd[str(k)].numel()) * 2.1e-4
        # [0] sum, [1] abs-sum, [2] cos(index)-weighted sum: the order-sensitive entry (|weight| <= 1: same slack)
        assert all(abs(d[i] - dg[i]) <= slack for i in range(3)), (what, k, d[:3], dg[:3])


def test_esrgan_gan_steps_vs_reference_trainer(dev):
    """Three steps of the UNMODIFIED ESRGANTrainer._gan_loop (23 RRDBs, 128x128 crops, batch 2): the five losses
    at 1e-3 after every step, the post-step parameter digests of G (702 entries) and D (60 entries) after the first.

    Parameters are not compared after later steps: the reference's own arithmetic does not determine them.  Its
    step oracle evaluated in fp32 and in fp64 (tools/experiments/esrgan_noise_floor.py) disagrees on 17-22 % of the
    elements of the discriminator's conv weights by more than 2e-6 after the second step (the relativistic gradients
    cancel heavily and Adam normalises what is left), and its losses by 1.6e-4 after the third.  Another fp32
    evaluation order (this one) lands as far away again: steps 0 and 1 hold 1e-3, step 2 is given 3e-3."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    t = make_trainer(dev)
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    for step in range(3):
        losses = t.gan_step(lr, hr)
        got = [losses[k].item() for k in LOSS_KEYS]
        want = gold['gan_losses'][step]
        tol = TOL if step < 2 else 3 * TOL
        for g, w in zip(got, want):
            assert abs(g - w) <= tol * max(abs(w), 1e-3), (step, got, list(want))
        assert abs(got[4] - gold['gan_ref_gen_losses'][step]) <= tol * gold['gan_ref_gen_losses'][step]
        if step == 0:
            assert_digests(gold['gs_keys'], t.generator.state_dict(), gold['gan_g_digest'][step], f'G step {step}')
            assert_digests(gold['ds_keys'], t.discriminator.state_dict(), gold['gan_d_digest'][step], f'D step {step}')


def assert_elementwise(got_sd, ref_sd, name, skip=()):
    """Every parameter within 2e-6 absolute of ``ref_sd`` after ONE Adam step (updates are ~1e-4: pins the update
    direction of every element whose gradient is above the noise floor; 0.5 % of a tensor, at least 2 elements, may sit
    at that floor -- tools/experiments/esrgan_noise_floor.py).  Returns the worst tensor for the record."""
    worst = ('', 0.0)
    for k, v in got_sd.items():
        r = ref_sd[k].detach()
        if not v.is_floating_point():
            assert int(v) == int(r), (name, k)
            continue
        if k in skip:
            continue
        diff = (v.detach().cpu() - r.cpu()).abs()
        if 'running_' in k:
            assert (diff.max() / r.abs().max().clamp_min(1e-6)).item() < 1e-3, (name, k)
        else:
            n_bad = int((diff > 2e-6).sum())
            assert n_bad <= max(2, int(5e-3 * diff.numel())), (name, k, n_bad, diff.max().item())
            assert diff.max().item() <= 2.1e-4, (name, k, diff.max().item())
            if n_bad / diff.numel() > worst[1]:
                worst = (k, n_bad / diff.numel())
    return worst


def test_esrgan_first_step_matches_oracle_elementwise(dev):
    """One ESRGAN GAN step (23 RRDBs, 128x128 crops, batch 2): every parameter of G and D within 2e-6 absolute of
    the CPU oracle's (updates are ~1e-4: this pins the update direction of every element whose gradient is above
    the noise floor).  The oracle's own fp32 and fp64 evaluations disagree on 0.23 % of a weight tensor's elements
    and on single elements of the 32-entry biases (tools/experiments/esrgan_noise_floor.py): 0.5 %, at least 2."""
    from oracle import esrgan as OE
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    t = make_trainer(dev)
    vgg_sd = {k: v.detach().cpu().clone() for k, v in t.vgg_loss.features.state_dict().items()}
    orc = OE.ESRGANStepOracle(step_state(t.generator.state_dict(), 'esrgan.G'), step_state(t.discriminator.state_dict(), 'esrgan.D'),
                              vgg_sd)
    lr, hr = torch.from_numpy(gold['low_res']), torch.from_numpy(gold['high_res'])
    orc.gan_step(lr, hr)
    t.gan_step(lr.to(dev), hr.to(dev))
    for name, mod, ref in (('G', t.generator, orc.g), ('D', t.discriminator, orc.d)):
        for k, v in mod.state_dict().items():
            if not v.is_floating_point():
                assert int(v) == int(ref[k]), (name, k)
                continue
            if (name, k) == ('D', 'classifier.2.bias'):
                continue  # exact gradient 0 (see assert_digests): Adam steps on rounding noise
            diff = (v.cpu() - ref[k].detach()).abs()
            if 'running_' in k:
                assert (diff.max() / ref[k].detach().abs().max().clamp_min(1e-6)).item() < 1e-3, (name, k)
            else:
                n_bad = int((diff > 2e-6).sum())
                assert n_bad <= max(2, int(5e-3 * diff.numel())), (name, k, n_bad, diff.max().item())
                assert diff.max().item() <= 2.1e-4, (name, k, diff.max().item())


def test_esrgan_config4_geometry_step_vs_reference_trainer(dev):
    """BASELINE config 4's geometry -- full 23-RRDB generator, 128x128 crops -- at batch 4 in fp32 against the
    reference trainer's losses and post-step parameter digests (seeded inputs; the fixture holds results only)."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    s_lr, s_hr = (int(v) for v in gold['b4_seeds'])
    lr, hr = seeded_input((4, 3, 32, 32), s_lr).to(dev), seeded_input((4, 3, 128, 128), s_hr).to(dev)
    t = make_trainer(dev, batch=4)
    losses = t.gan_step(lr, hr)
    got = [losses[k].item() for k in LOSS_KEYS]
    for g, w in zip(got, gold['b4_gan_losses']):
        assert abs(g - w) <= TOL * max(abs(w), 1e-3), (got, list(gold['b4_gan_losses']))
    assert abs(got[4] - float(gold['b4_gan_ref_gen_loss'])) <= TOL * float(gold['b4_gan_ref_gen_loss'])
    assert_digests(gold['gs_keys'], t.generator.state_dict(), gold['b4_gan_g_digest'], 'G b4')
    assert_digests(gold['ds_keys'], t.discriminator.state_dict(), gold['b4_gan_d_digest'], 'D b4')
    # ... and every element against the CPU oracle at this size (the digests above are sums over up to 8e5 elements)
    from oracle import esrgan as OE
    vgg_sd = {k: v.detach().cpu().clone() for k, v in t.vgg_loss.features.state_dict().items()}
    orc = OE.ESRGANStepOracle(step_state(t.generator.state_dict(), 'esrgan.G'), step_state(t.discriminator.state_dict(), 'esrgan.D'),
                              vgg_sd)
    orc.gan_step(lr.cpu(), hr.cpu())
    assert_elementwise(t.generator.state_dict(), orc.g, 'G b4')
    assert_elementwise(t.discriminator.state_dict(), orc.d, 'D b4', skip=('classifier.2.bias',))


def test_esrgan_config4_full_batch_step_vs_reference_trainer(dev):
    """BASELINE config 4 at its FULL size -- 23 RRDBs, 128x128 crops, batch 16 per GPU -- tied to the
    reference trainer without a 16-crop CPU run: the batch is the batch-4 fixture's four crops repeated four times.  Every
    loss is a batch mean, every gradient the gradient of one, the discriminator's BatchNorm statistics (biased, as training
    mode normalises with) are those of the four crops: losses and post-step parameters must equal the batch-4 golden's.
    (Only the running variances differ -- PyTorch stores the unbiased estimate, n/(n-1) -- and are left out.)"""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    s_lr, s_hr = (int(v) for v in gold['b4_seeds'])
    lr = seeded_input((4, 3, 32, 32), s_lr).repeat(4, 1, 1, 1).to(dev)
    hr = seeded_input((4, 3, 128, 128), s_hr).repeat(4, 1, 1, 1).to(dev)
    t = make_trainer(dev, batch=16)
    losses = t.gan_step(lr, hr)
    got = [losses[k].item() for k in LOSS_KEYS]
    for g, w in zip(got, gold['b4_gan_losses']):
        assert abs(g - w) <= TOL * max(abs(w), 1e-3), (got, list(gold['b4_gan_losses']))
    assert_digests(gold['gs_keys'], t.generator.state_dict(), gold['b4_gan_g_digest'], 'G b16')
    keep = [i for i, k in enumerate(gold['ds_keys']) if 'running_var' not in str(k)]
    assert_digests(gold['ds_keys'][keep], t.discriminator.state_dict(), gold['b4_gan_d_digest'][keep], 'D b16')
    # element by element: the batch-16 step of the replicated crops against the batch-4 step of the crops themselves
    # (which test_esrgan_config4_geometry_step_vs_reference_trainer ties to the oracle elementwise)
    t4 = make_trainer(dev, batch=4)
    t4.gan_step(lr[:4].contiguous(), hr[:4].contiguous())
    assert_elementwise(t.generator.state_dict(), t4.generator.state_dict(), 'G b16 vs b4')
    ref_d = {k: v for k, v in t4.discriminator.state_dict().items()}
    got_d = {k: v for k, v in t.discriminator.state_dict().items() if 'running_var' not in k}
    assert_elementwise(got_d, ref_d, 'D b16 vs b4', skip=('classifier.2.bias',))
    del t4
    # the same step in the precision config 4 is quoted at (autocast in both phases): within bf16 rounding of it
    tb = make_trainer(dev, batch=16, disable_amp=False)
    lb = tb.gan_step(lr, hr)
    for k, w in zip(LOSS_KEYS, gold['b4_gan_losses']):
        assert abs(lb[k].item() - w) <= 2e-2 * max(abs(w), 1e-3), (k, lb[k].item(), w)


def _bf16_oracle_pair(t, lr, hr):
    """Two CPU evaluations of ONE bf16 recipe (oracle.srgan.bf16_products) from the trainer's state: ``fp32sum`` accumulates the
    products of the rounded operands in fp32 in torch's order (what round 3 / 4 compared against), ``exact`` accumulates them
    exactly (fp64, one rounding) -- the recipe's own value.  Returns (oracle fp32sum, its losses, oracle exact, its losses)."""
    from oracle import esrgan as OE
    from oracle import srgan as O
    vgg_sd = {k: v.detach().cpu().clone() for k, v in t.vgg_loss.features.state_dict().items()}
    out = []
    for exact in (False, True):
        orc = OE.ESRGANStepOracle(step_state(t.generator.state_dict(), 'esrgan.G'), step_state(t.discriminator.state_dict(), 'esrgan.D'),
                                  vgg_sd)
        with O.bf16_products(exact_sums=exact):
            losses = orc.gan_step(lr, hr)
        out += [orc, losses]
    return out


def _bf16_within_yardstick(got_sd, fp32sum_sd, exact_sd, name, skip=()):
    """The bf16 step's parameters after ONE Adam step, judged by a yardstick computed here instead of a budget set by hand
    (round-4 review, item 7): any fp32-accumulating evaluation of the recipe -- this package's kernels, the oracle's own conv
    calls -- sits some distance from the exact-sum value, because an intermediate one fp32 ulp apart rounds to the other bf16
    neighbour in the next layer and Adam turns a gradient at the noise floor into a +-lr step either way.  The product may be
    no further from the exact-sum result than the ORACLE'S fp32-sum evaluation is, times 1.5: per tensor the count of elements
    more than 2e-6 away (plus the sampling noise of that count), the largest deviation (one Adam step: 2.1e-4),
    and for the BatchNorm running statistics the largest deviation relative to the tensor's scale."""
    for k, v in got_sd.items():
        if not v.is_floating_point() or k in skip:
            continue
        e, a = exact_sd[k].detach().cpu(), fp32sum_sd[k].detach().cpu()
        dp, da = (v.detach().cpu() - e).abs(), (a - e).abs()
        if 'running_' in k:
            scale = e.abs().max().clamp_min(1e-6)
            assert (dp.max() / scale).item() <= 1.5 * (da.max() / scale).item() + 2e-3, (name, k, (dp.max() / scale).item(), (da.max() / scale).item())
            continue
        bad_p, bad_a = int((dp > 2e-6).sum()), int((da > 2e-6).sum())
        # (which elements sit at the noise floor is a draw: the count of flipped ones scatters like a Poisson variable, three
        # standard deviations of the yardstick's own count are allowed on top -- it matters for the 32..512-entry vectors only)
        assert bad_p <= 1.5 * bad_a + 3.0 * max(bad_a, 1) ** 0.5 + 2, (name, k, bad_p, bad_a, dp.numel())
        # ... and an ABSOLUTE ceiling on top of the relative yardstick (round-5 advice): should the oracle's own fp32-sum evaluation
        # ever drift or get noisy, the product's allowance must not grow with it -- round 4's hand-set value, 8 % of a tensor
        # (+ the same small-vector slack), stays as the cap
        assert bad_p <= 0.08 * dp.numel() + 3.0 * max(0.08 * dp.numel(), 1.0) ** 0.5 + 2, (name, k, bad_p, dp.numel())
        assert dp.max().item() <= 2.1e-4, (name, k, dp.max().item())


def test_esrgan_config4_bf16_launch_geometry(dev):
    """The configuration ``bench.py`` times for BASELINE configs[3] -- 23 RRDBs, 128x128 crops, batch 16, bf16 products: the
    fused dense-block kernels on 16 x 32 x 32 pixels (256 workgroups), the image-row weight gradient on 16 384 rows, the
    paired problems -- pinned at that geometry, not only through its losses.  Both the batch-16 step (the fixture's four crops x 4:
    every loss a batch mean, BatchNorm statistics those of the four crops) and the batch-4 step of the crops themselves are held
    against the EXACT-SUM evaluation of the bf16 recipe on the CPU (``oracle.srgan.bf16_products(exact_sums=True)``): losses at
    2e-3, parameters element by element no further from it than the oracle's own fp32-sum evaluation is (x 1.5,
    ``_bf16_within_yardstick``) -- the yardstick is computed in the test, not raised to what a tile plan happens to produce
    (rounds 3 / 4 held 5 %, then 8 % and 15 %, then 1e-2 on the running statistics)."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    s_lr, s_hr = (int(v) for v in gold['b4_seeds'])
    lr4, hr4 = seeded_input((4, 3, 32, 32), s_lr), seeded_input((4, 3, 128, 128), s_hr)
    t4 = make_trainer(dev, batch=4, disable_amp=False)
    orc_a, want_a, orc_e, want_e = _bf16_oracle_pair(t4, lr4, hr4)
    l4 = t4.gan_step(lr4.to(dev), hr4.to(dev))
    got4 = [l4[k].item() for k in LOSS_KEYS]
    t16 = make_trainer(dev, batch=16, disable_amp=False)
    l16 = t16.gan_step(lr4.repeat(4, 1, 1, 1).to(dev), hr4.repeat(4, 1, 1, 1).to(dev))
    got16 = [l16[k].item() for k in LOSS_KEYS]
    for got in (got4, got16):
        for g, w, wa in zip(got, want_e, want_a):
            # 2e-3 of the exact-sum loss, or 1.5 x the oracle's own fp32-sum distance where that is larger -- but never more than
            # 5e-3 (round 4's ceiling: the relative yardstick may not grow without bound with the oracle's own noise)
            tol = min(max(2e-3 * max(abs(w), 1e-3), 1.5 * abs(wa - w)), 5e-3 * max(abs(w), 1e-3))
            assert abs(g - w) <= tol, (got, want_e, want_a)
    for a, b in zip(got16, got4):  # the same recipe on two launch shapes
        assert abs(a - b) <= 1e-3 * max(abs(b), 1e-3), (got16, got4)
    # ... and element by element: the batch-16 step of the four crops x 4 and the batch-4 step are the same arithmetic on two
    # launch shapes, so their parameters after the step may differ by one Adam step (an element at the noise floor going the
    # other way) and nowhere by more -- a bug that depends on the launch shape cannot hide inside the oracle's noise
    for net4, net16 in ((t4.generator, t16.generator), (t4.discriminator, t16.discriminator)):
        sd4 = net4.state_dict()
        for k, v in net16.state_dict().items():
            if v.is_floating_point() and 'running_' not in k:
                dev_ = (v - sd4[k]).abs().max().item()
                assert dev_ <= 2.1e-4, (k, dev_)
    _bf16_within_yardstick(t4.generator.state_dict(), orc_a.g, orc_e.g, 'G bf16 b4')
    _bf16_within_yardstick(t4.discriminator.state_dict(), orc_a.d, orc_e.d, 'D bf16 b4', skip=('classifier.2.bias',))
    _bf16_within_yardstick(t16.generator.state_dict(), orc_a.g, orc_e.g, 'G bf16 b16')
    got_d = {k: v for k, v in t16.discriminator.state_dict().items() if 'running_var' not in k}  # (unbiased: n / (n - 1) differs)
    _bf16_within_yardstick(got_d, orc_a.d, orc_e.d, 'D bf16 b16', skip=('classifier.2.bias',))


def test_esrgan_gan_step_with_bf16_products(dev):
    """Without --disable-amp both ESRGAN phases sit in the reference's autocast regions (esrgan/trainer.py:384,446,
    461): every generic conv of G, D and VGG19 multiplies bf16-rounded operands (fp32 accumulation, fp32 everything
    else).  The first GAN step must land within bf16 rounding of the fp32 golden losses: 2e-2 relative (bf16 carries
    8 bits of mantissa: 4e-3 per operand, deep networks compound it)."""
    from torchsr_amd.layers import Conv2d
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    t = make_trainer(dev, disable_amp=False)
    assert t.amp_phases == ('psnr', 'gan')
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    losses = t.gan_step(lr, hr)
    assert all(m._st.precision == 1 for net in (t.generator, t.discriminator, t.vgg_loss) for m in net.modules()
               if isinstance(m, Conv2d))
    got = [losses[k].item() for k in LOSS_KEYS]
    want = gold['gan_losses'][0]
    for g, w in zip(got, want):
        assert np.isfinite(g) and abs(g - w) <= 2e-2 * max(abs(w), 1e-3), (got, list(want))
    assert any(abs(g - w) > 1e-7 * max(abs(w), 1e-3) for g, w in zip(got, want))  # it really is a different precision


def test_esrgan_bf16_step_vs_bf16_oracle(dev):
    """The autocast step against an oracle that rounds the SAME operands (oracle.srgan.bf16_products: both factors of
    every product in the forward pass, the data gradients and the weight gradients to bf16, everything else fp32), summed
    exactly: the five losses of the first step within 2e-3, and every parameter after the first Adam step no further from
    the exact-sum result than the oracle's own fp32-sum evaluation of the recipe is (x 1.5) -- what 'within bf16 rounding
    of the fp32 golden' (2e-2) could not show."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    t = make_trainer(dev, disable_amp=False)
    lr, hr = torch.from_numpy(gold['low_res']), torch.from_numpy(gold['high_res'])
    orc_a, want_a, orc_e, want_e = _bf16_oracle_pair(t, lr, hr)
    losses = t.gan_step(lr.to(dev), hr.to(dev))
    got = [losses[k].item() for k in LOSS_KEYS]
    for g, w, wa in zip(got, want_e, want_a):
        assert abs(g - w) <= max(2e-3 * max(abs(w), 1e-3), 1.5 * abs(wa - w)), (got, want_e, want_a)
    fp32 = gold['gan_losses'][0]
    assert max(abs(w - f) / max(abs(f), 1e-3) for w, f in zip(want_e, fp32)) > 2e-4   # the oracle did change its arithmetic
    _bf16_within_yardstick(t.generator.state_dict(), orc_a.g, orc_e.g, 'G bf16 b2')
    _bf16_within_yardstick(t.discriminator.state_dict(), orc_a.d, orc_e.d, 'D bf16 b2', skip=('classifier.2.bias',))


def test_esrgan_segmented_step_equals_fused(dev):
    """The data-parallel form of the ESRGAN step (backward paused at 'd.head' / 'g.tail', gradient buckets between hipGraph
    segments; world size 1, so the all-reduces are no-ops) against the single-graph step."""
    from torchsr_amd import functional as F
    from torchsr_amd.ddp import BackwardCuts, GradBuckets
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    ta, tb = make_trainer(dev), make_trainer(dev, use_graphs=True)
    tb.distributed = True
    tb._cuts = BackwardCuts()
    tb.gen_sync = GradBuckets(tb.gen_flat, (tb.gen_tail_bucket,), tb.generator)
    tb.disc_sync = GradBuckets(tb.disc_flat, (tb.disc_head_bucket,), tb.discriminator)
    tail = 3 * (64 * 64 * 9 + 64) + 3 * 64 * 9 + 3   # upsample1/2, conv3.0, conv4 (+ the flat buffer's alignment padding)
    assert tail <= tb.gen_sync.slices[1].numel() < tail + 4
    for step in range(4):
        la, lb = ta.gan_step(lr, hr), tb.gan_step(lr, hr)
        for k in la:
            assert la[k].item() == pytest.approx(lb[k].item(), rel=1e-5, abs=1e-7), (step, k)
        assert not tb._cuts.pairs
    assert {'gan.disc.head', 'gan.disc.body', 'gan.content', 'gan.gen.head', 'gan.gen.body', 'gan.gopt'} <= set(tb._graphs)
    assert F.cut_hook[0] is None


def test_esrgan_step_is_bitwise_reproducible(dev):
    """Two ESRGAN trainers (bf16 products, the RRDB trunk node, paired and scaled weight-gradient groups, hipGraph replays
    from the third step) from the same state: bit-identical losses and parameters after four steps."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    runs = []
    for _ in range(2):
        t = make_trainer(dev, disable_amp=False, use_graphs=True)
        losses = [[v.item() for v in t.gan_step(lr, hr).values()] for _step in range(4)]
        assert 'gan.all' in t._graphs
        runs.append((losses, {k: v.clone() for k, v in t.generator.state_dict().items()},
                     {k: v.clone() for k, v in t.discriminator.state_dict().items()}))
        del t
        torch.cuda.empty_cache()
        junk = torch.full((64 << 20,), float('nan'), device=dev)
        del junk
    assert runs[0][0] == runs[1][0]
    for a, b in ((runs[0][1], runs[1][1]), (runs[0][2], runs[1][2])):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_esrgan_two_branch_graph_equals_the_single_stream_step(dev):
    """ESRGAN's captured step runs the perceptual loss's FORWARD (esrgan/trainer.py:458-459) on a second graph branch next to
    the discriminator's update; its backward stays where autograd puts it (a three-term sum at the generator's output keeps
    its order).  Four steps with and without the second branch: bit-identical losses and parameters."""
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)
    runs = []
    for overlap in (True, False):
        t = make_trainer(dev, disable_amp=False, use_graphs=True)
        t.overlap_branches = overlap
        losses = [[v.item() for _k, v in sorted(t.gan_step(lr, hr).items())] for _step in range(4)]
        assert 'gan.all' in t._graphs
        runs.append((losses, {k: v.clone() for k, v in t.generator.state_dict().items()},
                     {k: v.clone() for k, v in t.discriminator.state_dict().items()}))
        del t
    assert runs[0][0] == runs[1][0]
    for a, b in ((runs[0][1], runs[1][1]), (runs[0][2], runs[1][2])):
        for k in a:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('bf16', [False, True])
def test_rrdb_modules_alone_equal_the_trunk_node(dev, bf16):
    """`ResidualInResidualDenseBlock` / `ResidualDenseBlock` called directly (esrgan/residual.py:65-86,110-129: one
    autograd node per dense block, `out * 0.2 + x` as an elementwise pass) against the generator's one-node chain
    (`functional.rrdb_trunk`, which the goldens pin): outputs, input gradient and every parameter gradient."""
    from torchsr_amd import functional as F
    from torchsr_amd.esrgan.residual import ResidualInResidualDenseBlock
    from torchsr_amd.layers import set_conv_precision
    torch.manual_seed(9)
    blocks = torch.nn.Sequential(*[ResidualInResidualDenseBlock() for _ in range(2)]).to(dev)
    with torch.no_grad():
        for p in blocks.parameters():   # (the reference's x0.1 init makes the residual branches tiny: scale them up)
            p.mul_(4.0)
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    if bf16:
        set_conv_precision(blocks, 'bf16')
    x = torch.randn(2, 16, 16, 64, device=dev)
    gy = torch.randn(2, 16, 16, 64, device=dev)
    def run(how):
        blocks.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = blocks(xi) if how == 'modules' else F.rrdb_trunk(xi, list(blocks))
        y.backward(gy)
        return y.detach(), xi.grad.clone(), [p.grad.clone() for p in blocks.parameters()]

    (ya, dxa, ga), (yb, dxb, gb) = run('modules'), run('trunk')
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()  # noqa: E731
    # fp32: the same arithmetic in another order.  bf16 products: the modules round conv5's output gradient AFTER the
    # block's 0.2 (bf16(0.2 dy), what autograd hands the conv), the trunk rounds dy and applies 0.2 to the fp32 result
    # (0.2 W^T bf16(dy)) -- 0.2 is not a power of two, so the two differ by bf16 rounding noise on the way back
    # -- and every later operand that differs in its last fp32 bits may round to the other bf16 neighbour (2^-9 of that
    # product).  The parameter gradients are sums over only 512 pixels here: compared in the L2 norm, loosely element-wise.
    rel2 = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()  # noqa: E731
    # (forward, bf16: the trunk runs a dense block as one launch -- srx_rdb_fwd -- which sums the same products in another
    # order than the modules' five launches; an intermediate that then differs in its last fp32 bit can round to the other
    # bf16 neighbour, 2^-9 of one product)
    assert rel(ya, yb) < (1e-4 if bf16 else 1e-5) and rel(dxa, dxb) < (5e-4 if bf16 else 2e-5), (rel(ya, yb), rel(dxa, dxb))
    if bf16:  # yardstick: the exact-fp32 gradients.  Two bf16 evaluation orders must be closer to each other than to those
        set_conv_precision(blocks, 'fp32')
        _, dxe, ge = run('modules')
        assert rel2(dxa, dxb) < rel2(dxa, dxe) and rel2(dxb, dxe) < 2e-2, (rel2(dxa, dxb), rel2(dxa, dxe), rel2(dxb, dxe))
    for i, ((name, _), a, b) in enumerate(zip(blocks.named_parameters(), ga, gb)):
        if bf16:
            far_a, far_b = rel2(a, ge[i]), rel2(b, ge[i])
            # (1e-6: a bias gradient is an fp32 sum of the output gradient -- no bf16 product in it -- so all three evaluations agree to
            # fp32 summation-order rounding and "closer to each other than to exact fp32" compares 1.8e-7 with 1.8e-7 there: the first
            # failure of this line, round 6, after the weight-gradient kernel's bias rows changed their summation order)
            assert rel2(a, b) < max(far_a, far_b, 1e-6) and far_b < max(2 * far_a, 1e-2), (name, rel2(a, b), far_a, far_b)
        else:
            assert rel(a, b) < 2e-5, (name, rel(a, b))


def test_graphs_survive_a_switch_between_phases_with_other_pack_tables(dev):
    """Round-4 advice: a pre-training graph captured with fused dense blocks (bf16 products: their convs' fp32 packs are left
    out of the repack table) and a GAN graph captured in exact fp32 (all convs in the table) replay different repack work after
    their Adam steps; switching between them must neither replay a graph that holds a freed table nor start a step on stale
    packed weights.  A trainer that autocasts ONLY the pre-training phase (as the reference's SRGAN trainer does) alternates
    the two with hipGraphs on; every loss must equal the graph-free trainer's, which repacks lazily per layer."""
    from torchsr_amd.esrgan.trainer import ESRGANTrainer

    class PsnrOnlyAmp(ESRGANTrainer):
        amp_phases = ('psnr',)

    def build(use_graphs):
        torch.manual_seed(0)
        args = Namespace(disable_amp=False, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                         psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=use_graphs,
                         vgg_weights='random')
        t = PsnrOnlyAmp(dev, args, [], [], 2, 2, distributed=False)
        t.generator.load_state_dict(step_state(t.generator.state_dict(), 'esrgan.G'))
        t.discriminator.load_state_dict(step_state(t.discriminator.state_dict(), 'esrgan.D'))
        t.generator.train()
        t.discriminator.train()
        return t

    lr, hr = seeded_input((2, 3, 32, 32), 5).to(dev), seeded_input((2, 3, 128, 128), 6).to(dev)
    plan = ['psnr'] * 4 + ['gan'] * 4 + ['psnr'] * 3 + ['gan'] * 3 + ['psnr', 'gan', 'psnr', 'gan']
    out = {}
    for use_graphs in (False, True):
        t = build(use_graphs)
        vals = []
        for ph in plan:
            if ph == 'psnr':
                vals.append(float(t.pretrain_step(lr, hr)))
            else:
                vals.append(float(t.gan_step(lr, hr)['gan/train-loss']))
        out[use_graphs] = vals
        if use_graphs:
            assert t._graphs, 'the graph trainer never captured'
    for i, (a, b) in enumerate(zip(out[False], out[True])):
        assert abs(a - b) <= 1e-6 * max(abs(a), 1e-6), (i, plan[i], a, b)  # (same kernels, same order: equal unless a pack was stale)
