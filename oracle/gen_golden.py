"""ORACLE pinning script -- run ONLY in the build container (needs /root/reference).

    python oracle/gen_golden.py

1. imports the reference's own modules (``torchsr.srgan.{generator,discriminator}``) and,
   under a minimal ``torchvision`` stub (torchvision is not installed here; the stub only
   supplies a plain-``nn`` VGG19 cfg 'E' container, ToTensor/Resize and save_image), the
   UNMODIFIED ``torchsr.srgan.trainer.SRGANTrainer``;
2. fills every parameter by key name from ``oracle/weights.py`` (closed form);
3. asserts the CPU restatement in ``oracle/srgan.py`` against the reference outputs;
4. writes small fixtures (inputs, outputs, gradient digests, step losses) to ``tests/golden``.

The fixtures are data; no reference source is copied.  The GPU box never runs this file.
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
sys.path.insert(0, REF)

from oracle import esrgan as OE  # noqa: E402
from oracle import srgan as O  # noqa: E402
from oracle.weights import closed_form_state, sample_table, seeded_input, step_state, tensor_digest  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def install_torchvision_stub():
    """torchvision is absent in this image (plain ModuleNotFoundError).  Provide just what
    torchsr/srgan/{loss,trainer}.py import: models.vgg19, utils.save_image, transforms.*."""
    tv = types.ModuleType('torchvision')
    models = types.ModuleType('torchvision.models')
    utils = types.ModuleType('torchvision.utils')
    transforms = types.ModuleType('torchvision.transforms')
    tfunc = types.ModuleType('torchvision.transforms.functional')
    ttrans = types.ModuleType('torchvision.transforms.transforms')

    class VGG(nn.Module):
        def __init__(self):
            super().__init__()
            layers, cin = [], 3
            for v in O.VGG19_CFG:
                if v == 'M':
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                    cin = v
            self.features = nn.Sequential(*layers)

    def vgg19(pretrained=False, **kw):
        m = VGG()
        m.load_state_dict(closed_form_state(m.state_dict()))  # keys 'features.N.weight'
        return m

    class InterpolationMode:
        BICUBIC = 'bicubic'

    class ToTensor:
        def __call__(self, img):
            a = np.asarray(img.convert('RGB'), dtype=np.float32) / 255.0
            return torch.from_numpy(a).permute(2, 0, 1).contiguous()

    class Resize:
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    models.vgg19 = vgg19
    utils.save_image = lambda *a, **k: None
    tfunc.InterpolationMode = InterpolationMode
    ttrans.Resize, ttrans.ToTensor = Resize, ToTensor
    transforms.functional, transforms.transforms = tfunc, ttrans
    transforms.ToTensor, transforms.Resize = ToTensor, Resize
    tv.models, tv.utils, tv.transforms = models, utils, transforms
    for name, mod in [('torchvision', tv), ('torchvision.models', models), ('torchvision.utils', utils),
                      ('torchvision.transforms', transforms), ('torchvision.transforms.functional', tfunc),
                      ('torchvision.transforms.transforms', ttrans)]:
        sys.modules[name] = mod


def grads_of(module, loss):
    module.zero_grad()
    loss.backward()
    return {k: p.grad.clone() for k, p in module.named_parameters()}


def digest_table(d):
    keys = sorted(d.keys())
    return np.array(keys), np.stack([tensor_digest(d[k]) for k in keys])


def check(name, a, b, tol=1e-6):
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    assert err <= tol * max(scale, 1.0), f'{name}: oracle differs from reference by {err}'
    print(f'  oracle == reference  {name:40s} max|diff| {err:.3e}')


def well_conditioned_seed(fn64, fn32, shape, seeds, tol=2e-5):
    """PReLU / LeakyReLU / ReLU / max-pool are not differentiable at 0: a pre-activation that lands
    within rounding of 0 makes the fp32 reference's own gradient differ from its fp64 evaluation by
    ~1e-3 (seen with seed 11).  Such an input pins nothing, so pick the first seed for which the
    reference agrees with an fp64 evaluation of itself."""
    for seed in seeds:
        x = seeded_input(shape, seed)
        a, b = fn32(x), fn64(x)
        err = ((a.double() - b).abs().max() / b.abs().max()).item()
        if err < tol:
            return seed
        print(f'  seed {seed} skipped: fp32 vs fp64 gradient differ by {err:.2e} (activation kink)')
    raise RuntimeError('no well conditioned seed found')


def _dx_of(forward, sd0, dtype):
    def fn(x):
        sd = {k: (v.to(dtype).clone() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        xo = x.to(dtype).clone().requires_grad_(True)
        forward(sd, xo).square().mean().backward()
        return xo.grad
    return fn


def kink_margin(module, x):
    """Smallest |pre-activation| / rms over every PReLU / LeakyReLU / ReLU input of ``module`` for
    input ``x``: how far the closest activation is from its non-differentiable point."""
    margins, hooks = [], []

    def pre(_m, inp):
        t = inp[0].detach().double()
        margins.append((t.abs().min() / t.square().mean().sqrt()).item())

    for m in module.modules():
        if isinstance(m, (nn.PReLU, nn.LeakyReLU, nn.ReLU)):
            hooks.append(m.register_forward_pre_hook(pre))
    with torch.no_grad():
        module(x)
    for h in hooks:
        h.remove()
    return min(margins)


def widest_margin_seed(module, sd0, shape, seeds):
    """Any two fp32 implementations disagree on the sign of a pre-activation that is within rounding
    of zero, and one flipped PReLU moves the input gradient by ~1e-3 -- more than the parity
    tolerance.  Golden inputs are therefore chosen so that the reference keeps every activation at
    least ~1e-5 rms away from its kink (forward agreement is ~1e-6)."""
    best, best_m = None, -1.0
    for seed in seeds:
        module.load_state_dict(sd0)
        module.train()
        m = kink_margin(module, seeded_input(shape, seed))
        if m > best_m:
            best, best_m = seed, m
    print(f'  input {shape}: seed {best} keeps activations >= {best_m:.2e} rms from their kinks')
    assert best_m > 5e-6, 'no input with a usable kink margin'
    return best


def gen_generator():
    from torchsr.srgan.generator import Generator
    ref = Generator()
    sd0 = closed_form_state(ref.state_dict())
    out = {}
    for tag, shape, seeds in [('a', (2, 3, 8, 8), range(100, 140)), ('b', (1, 3, 6, 10), range(200, 240))]:
        seed = widest_margin_seed(ref, sd0, shape, seeds)
        ref.load_state_dict(sd0)
        x = seeded_input(shape, seed).requires_grad_(True)
        ref.train()
        y = ref(x)
        loss = y.square().mean()
        g = grads_of(ref, loss)
        # oracle, same state
        sd = {k: v.clone() for k, v in sd0.items()}
        leaves = O._leaves(sd)
        xo = x.detach().clone().requires_grad_(True)
        yo = O.generator_forward(sd, xo, True)
        yo.square().mean().backward()
        check(f'G[{tag}] train output', yo, y)
        check(f'G[{tag}] dx', xo.grad, x.grad)
        names = [k for k, v in sd0.items() if v.is_floating_point() and 'running_' not in k]
        for k, leaf in zip(names, leaves):
            check(f'G[{tag}] grad {k}' if k in ('conv1.0.weight', 'blocks.7.conv2.weight', 'conv3.bias') else k,
                  leaf.grad, g[k]) if k in ('conv1.0.weight', 'blocks.7.conv2.weight', 'conv3.bias',
                                            'conv_layers.1.conv.weight', 'blocks.0.prelu.weight') else None
        post = ref.state_dict()
        for k in ('blocks.5.bn2.running_mean', 'conv2.1.running_var'):
            check(f'G[{tag}] {k}', sd[k], post[k])
        ref.eval()
        with torch.no_grad():
            ye = ref(x.detach())
        check(f'G[{tag}] eval output', O.generator_forward({k: v.clone() for k, v in post.items()}, x.detach(), False), ye)
        gk, gd = digest_table(g)
        rk, rd = digest_table({k: v for k, v in post.items() if 'running_' in k})
        out.update({f'{tag}_x': x.detach().numpy(), f'{tag}_y_train': y.detach().numpy(),
                    f'{tag}_y_eval': ye.numpy(), f'{tag}_dx': x.grad.numpy(), f'{tag}_grad_keys': gk,
                    f'{tag}_grad_digest': gd, f'{tag}_running_keys': rk, f'{tag}_running_digest': rd,
                    f'{tag}_loss': np.float64(loss.item())})
    np.savez_compressed(os.path.join(OUT, 'srgan_generator.npz'), **out)


def gen_discriminator():
    from torchsr.srgan.discriminator import Discriminator
    out = {}
    dfwd = lambda sd, x: O.discriminator_forward(sd, x, True)  # noqa: E731
    for tag, size, seeds in [('s32', 32, range(31, 40)), ('s96', 96, range(41, 50))]:
        ref = Discriminator(image_size=size)
        sd0 = closed_form_state(ref.state_dict())
        seed = well_conditioned_seed(_dx_of(dfwd, sd0, torch.float64), _dx_of(dfwd, sd0, torch.float32),
                                     (2, 3, size, size), seeds)
        ref.load_state_dict(sd0)
        x = seeded_input((2, 3, size, size), seed).requires_grad_(True)
        ref.train()
        p = ref(x)
        loss = torch.nn.functional.binary_cross_entropy(p, torch.full((2, 1), 1.0))
        g = grads_of(ref, loss)
        sd = {k: v.clone() for k, v in sd0.items()}
        leaves = O._leaves(sd)
        xo = x.detach().clone().requires_grad_(True)
        po = O.discriminator_forward(sd, xo, True)
        torch.nn.functional.binary_cross_entropy(po, torch.full((2, 1), 1.0)).backward()
        check(f'D[{tag}] output', po, p)
        check(f'D[{tag}] dx', xo.grad, x.grad)
        names = [k for k, v in sd0.items() if v.is_floating_point() and 'running_' not in k]
        for k, leaf in zip(names, leaves):
            if k in ('features.0.weight', 'features.8.weight', 'features.21.bias', 'classifier.0.weight'):
                check(f'D[{tag}] grad {k}', leaf.grad, g[k])
        post = ref.state_dict()
        ref.eval()
        with torch.no_grad():
            pe = ref(x.detach())
        gk, gd = digest_table(g)
        rk, rd = digest_table({k: v for k, v in post.items() if 'running_' in k})
        out.update({f'{tag}_x': x.detach().numpy(), f'{tag}_p_train': p.detach().numpy(), f'{tag}_p_eval': pe.numpy(),
                    f'{tag}_dx_digest': tensor_digest(x.grad), f'{tag}_grad_keys': gk, f'{tag}_grad_digest': gd,
                    f'{tag}_running_keys': rk, f'{tag}_running_digest': rd, f'{tag}_loss': np.float64(loss.item())})
        if size == 32:
            out[f'{tag}_dx'] = x.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'srgan_discriminator.npz'), **out)


def gen_vgg():
    from torchsr.srgan.loss import VGGLoss
    ref = VGGLoss()  # stubbed torchvision.models.vgg19 with closed-form weights
    sd = {k: v.clone() for k, v in ref.features.state_dict().items()}
    vfwd = lambda sd_, x: O.vgg_features(sd_, x)  # noqa: E731
    seed = well_conditioned_seed(_dx_of(vfwd, sd, torch.float64), _dx_of(vfwd, sd, torch.float32), (2, 3, 32, 32),
                                 range(51, 60))
    src = seeded_input((2, 3, 32, 32), seed).requires_grad_(True)
    tgt = seeded_input((2, 3, 32, 32), 99)
    loss = ref(src, tgt)
    loss.backward()
    so = src.detach().clone().requires_grad_(True)
    lo = O.vgg_loss(sd, so, tgt)
    lo.backward()
    check('VGG loss', lo, loss)
    check('VGG d(source)', so.grad, src.grad)
    with torch.no_grad():
        feat = ref.features(src.detach())
    np.savez_compressed(os.path.join(OUT, 'vgg19.npz'), src=src.detach().numpy(), tgt=tgt.numpy(),
                        features=feat.numpy(), loss=np.float64(loss.item()), dsrc=src.grad.numpy())


def _reference_srgan_trainer(batch):
    from torchsr.srgan.trainer import SRGANTrainer
    args = Namespace(disable_amp=True, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1)
    t = SRGANTrainer('cpu', args, [], [], batch, batch, distributed=False)
    t.generator.load_state_dict(step_state(t.generator.state_dict(), 'srgan.G'))
    t.discriminator.load_state_dict(step_state(t.discriminator.state_dict(), 'srgan.D'))
    t.generator.train()
    t.discriminator.train()
    return t


def _oracle_for(t, with_vgg=True):
    return O.SRGANStepOracle(step_state(t.generator.state_dict(), 'srgan.G'), step_state(t.discriminator.state_dict(), 'srgan.D'),
                             {k: v.clone() for k, v in t.vgg_loss.features.state_dict().items()} if with_vgg else {})


def _reference_pretrain_body(t, lr_img, hr_img):
    """srgan/trainer.py:380-388: the loop is inline in _pretrain, so its statements are executed here verbatim on
    the reference trainer's own objects."""
    import torch.cuda.amp as amp
    t.psnr_optimizer.zero_grad()
    with amp.autocast(enabled=t.amp):
        super_res = t.generator(lr_img)
        loss = t.mse_loss(super_res, hr_img)
    t.scaler.scale(loss).backward()
    t.scaler.step(t.psnr_optimizer)
    t.scaler.update()
    return float(loss)


def gen_steps():
    """Three consecutive SRGANTrainer._gan_loop steps and three _pretrain bodies on a fixed batch of 2, plus one
    of each at BASELINE config 2's size (batch 16, 96x96 crops).  Weights: ``step_state`` (closed form, conditioned
    so that the discriminator does not saturate: every step pins arithmetic, not log(1e-6))."""
    os.chdir(REF)  # the trainer opens 'media/waterfalls-low-res.png' relative to the CWD
    lr_img = seeded_input((2, 3, 24, 24), 41)
    hr_img = seeded_input((2, 3, 96, 96), 42)
    out = {'low_res': lr_img.numpy(), 'high_res': hr_img.numpy()}

    # ---- GAN phase (srgan/trainer.py:416-469), unmodified method
    t = _reference_srgan_trainer(2)
    logged = []
    t._log_wandb = lambda contents, step=None: logged.append(float(contents['gan/train-loss']))
    orc = _oracle_for(t)
    gan_losses, gdig, ddig, ref_gen_losses, gsam, dsam = [], [], [], [], [], []
    for step in range(3):
        t._gan_loop(lr_img, hr_img, step)
        dl, cl, al, gl = orc.gan_step(lr_img, hr_img)
        # step 0 agrees to rounding; later steps carry the difference between this file's Adam (mul_/add_, as in
        # the pinned torch 1.11) and torch 2.10's (lerp_): a few 1e-6 of a loss in the un-saturated regime
        tol = 1e-6 if step == 0 else 2e-5
        assert abs(gl - logged[-1]) <= tol * max(1, abs(gl)), (gl, logged[-1])
        ref_gen_losses.append(logged[-1])
        for k in ('conv3.weight', 'blocks.0.conv1.weight', 'blocks.15.bn2.running_var'):
            check(f'step{step} G {k}', orc.g[k].detach(), t.generator.state_dict()[k], tol=1e-6 if step == 0 else 1e-4)
        for k in ('features.0.weight', 'classifier.0.weight', 'features.21.running_mean'):
            check(f'step{step} D {k}', orc.d[k].detach(), t.discriminator.state_dict()[k], tol=1e-6 if step == 0 else 1e-4)
        gan_losses.append([dl, cl, al, gl])
        gdig.append(np.stack([tensor_digest(v) for k, v in sorted(t.generator.state_dict().items())]))
        ddig.append(np.stack([tensor_digest(v) for k, v in sorted(t.discriminator.state_dict().items())]))
        # (round 6) the large tensors element by element: a strided sample of the REFERENCE's post-step values
        gsam.append(sample_table(t.generator.state_dict()))
        dsam.append(sample_table(t.discriminator.state_dict()))
        print(f'  gan step {step}: disc {dl:.6f} content {cl:.6f} adv {al:.6f} gen {gl:.6f}')
    out.update(gan_ref_gen_losses=np.array(ref_gen_losses), gan_losses=np.array(gan_losses), gan_g_digest=np.stack(gdig), gan_d_digest=np.stack(ddig),
               gan_g_sample=np.stack(gsam), gan_d_sample=np.stack(dsam),
               g_keys=np.array(sorted(t.generator.state_dict().keys())),
               d_keys=np.array(sorted(t.discriminator.state_dict().keys())))
    with torch.no_grad():
        t.generator.eval()
        sr = t.generator(lr_img)
    out['gan_sr_after3'] = sr.numpy()
    out['gan_psnr_after3'] = np.float64(O.psnr(sr, hr_img))

    # ---- pretrain body (srgan/trainer.py:376-388)
    t = _reference_srgan_trainer(2)
    orc = _oracle_for(t, with_vgg=False)
    pre_losses, pdig, psam = [], [], []
    for step in range(3):
        loss = _reference_pretrain_body(t, lr_img, hr_img)
        lo = orc.pretrain_step(lr_img, hr_img)
        assert abs(lo - loss) <= (1e-6 if step == 0 else 2e-5) * max(1, abs(lo)), (lo, loss)
        check(f'pretrain step{step} conv3.weight', orc.g['conv3.weight'].detach(), t.generator.state_dict()['conv3.weight'], tol=1e-6 if step == 0 else 1e-4)
        pre_losses.append(loss)
        pdig.append(np.stack([tensor_digest(v) for k, v in sorted(t.generator.state_dict().items())]))
        psam.append(sample_table(t.generator.state_dict()))
        print(f'  pretrain step {step}: mse {loss:.6f}')
    out.update(pre_losses=np.array(pre_losses), pre_g_digest=np.stack(pdig), pre_g_sample=np.stack(psam))

    # ---- BASELINE config 2 size: batch 16 (inputs are seeded: only losses and digests are stored)
    lr16, hr16 = seeded_input((16, 3, 24, 24), 141), seeded_input((16, 3, 96, 96), 142)
    t = _reference_srgan_trainer(16)
    logged = []
    t._log_wandb = lambda contents, step=None: logged.append(float(contents['gan/train-loss']))
    orc = _oracle_for(t)
    t._gan_loop(lr16, hr16, 0)
    res = orc.gan_step(lr16, hr16)
    assert abs(res[3] - logged[-1]) <= 1e-6 * max(1, abs(res[3])), (res, logged)
    check('b16 G conv3.weight', orc.g['conv3.weight'].detach(), t.generator.state_dict()['conv3.weight'])
    check('b16 D classifier.0.weight', orc.d['classifier.0.weight'].detach(), t.discriminator.state_dict()['classifier.0.weight'])
    print(f'  batch-16 gan step: disc {res[0]:.6f} content {res[1]:.6f} adv {res[2]:.6f} gen {res[3]:.6f}')
    out.update(b16_seeds=np.array([141, 142]), b16_gan_losses=np.array(res), b16_gan_ref_gen_loss=np.float64(logged[-1]),
               b16_gan_g_digest=np.stack([tensor_digest(v) for k, v in sorted(t.generator.state_dict().items())]),
               b16_gan_d_digest=np.stack([tensor_digest(v) for k, v in sorted(t.discriminator.state_dict().items())]),
               b16_gan_g_sample=sample_table(t.generator.state_dict()), b16_gan_d_sample=sample_table(t.discriminator.state_dict()))
    t = _reference_srgan_trainer(16)
    loss = _reference_pretrain_body(t, lr16, hr16)
    print(f'  batch-16 pretrain step: mse {loss:.6f}')
    out.update(b16_pre_loss=np.float64(loss),
               b16_pre_g_digest=np.stack([tensor_digest(v) for k, v in sorted(t.generator.state_dict().items())]),
               b16_pre_g_sample=sample_table(t.generator.state_dict()))
    np.savez_compressed(os.path.join(OUT, 'srgan_steps.npz'), **out)
    os.chdir(ROOT)


def gen_esrgan():
    """ESRGAN generator (2 RRDBs), discriminator (64 px), three steps of the UNMODIFIED ESRGANTrainer._gan_loop
    (full 23-RRDB generator, 128x128 crops, batch 2) with post-step parameter digests, and one step at batch 4."""
    from torchsr.esrgan.discriminator import Discriminator
    from torchsr.esrgan.generator import Generator
    out = {}
    ref = Generator(num_rrdb_blocks=2)
    sd0 = closed_form_state(ref.state_dict())
    seed = well_conditioned_seed(_dx_of(lambda sd, x: OE.generator_forward(sd, x), sd0, torch.float64),
                                 _dx_of(lambda sd, x: OE.generator_forward(sd, x), sd0, torch.float32), (2, 3, 8, 10),
                                 range(300, 320))
    ref.load_state_dict(sd0)
    x = seeded_input((2, 3, 8, 10), seed).requires_grad_(True)
    y = ref(x)
    loss = y.square().mean()
    g = grads_of(ref, loss)
    sd = {k: v.clone() for k, v in sd0.items()}
    leaves = O._leaves(sd)
    xo = x.detach().clone().requires_grad_(True)
    yo = OE.generator_forward(sd, xo)
    yo.square().mean().backward()
    check('ESRGAN G output', yo, y)
    check('ESRGAN G dx', xo.grad, x.grad)
    names = [k for k, v in sd0.items() if v.is_floating_point()]
    for k, leaf in zip(names, leaves):
        if k in ('conv1.weight', 'blocks.1.RDB2.conv3.0.weight', 'blocks.0.RDB1.conv5.bias', 'conv4.weight'):
            check(f'ESRGAN G grad {k}', leaf.grad, g[k])
    gk, gd = digest_table(g)
    out.update(g_x=x.detach().numpy(), g_y=y.detach().numpy(), g_dx=x.grad.numpy(), g_grad_keys=gk, g_grad_digest=gd,
               g_loss=np.float64(loss.item()))

    refd = Discriminator(image_size=64)
    sd0 = closed_form_state(refd.state_dict())
    dfwd = lambda sd_, x_: OE.discriminator_forward(sd_, x_, True)  # noqa: E731
    seed = well_conditioned_seed(_dx_of(dfwd, sd0, torch.float64), _dx_of(dfwd, sd0, torch.float32), (2, 3, 64, 64),
                                 range(320, 340))
    refd.load_state_dict(sd0)
    x = seeded_input((2, 3, 64, 64), seed).requires_grad_(True)
    refd.train()
    logits = refd(x)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits - 0.3, torch.full((2, 1), 1.0))
    g = grads_of(refd, loss)
    sd = {k: v.clone() for k, v in sd0.items()}
    xo = x.detach().clone().requires_grad_(True)
    lo = OE.discriminator_forward(sd, xo, True)
    torch.nn.functional.binary_cross_entropy_with_logits(lo - 0.3, torch.full((2, 1), 1.0)).backward()
    check('ESRGAN D logits', lo, logits)
    check('ESRGAN D dx', xo.grad, x.grad)
    gk, gd = digest_table(g)
    out.update(d_x=x.detach().numpy(), d_logits=logits.detach().numpy(), d_dx_digest=tensor_digest(x.grad),
               d_grad_keys=gk, d_grad_digest=gd, d_loss=np.float64(loss.item()))

    os.chdir(REF)
    from torchsr.esrgan.trainer import ESRGANTrainer

    def run(batch, lr_img, hr_img, steps, tag):
        args = Namespace(disable_amp=True, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                         psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1)
        t = ESRGANTrainer('cpu', args, [], [], batch, batch, distributed=False)
        t.generator.load_state_dict(step_state(t.generator.state_dict(), 'esrgan.G'))
        t.discriminator.load_state_dict(step_state(t.discriminator.state_dict(), 'esrgan.D'))
        t.generator.train()
        t.discriminator.train()
        logged = []
        t._log_wandb = lambda contents, step=None: logged.append(float(contents['gan/train-loss']))
        orc = OE.ESRGANStepOracle(step_state(t.generator.state_dict(), 'esrgan.G'),
                                  step_state(t.discriminator.state_dict(), 'esrgan.D'),
                                  {k: v.clone() for k, v in t.vgg_loss.features.state_dict().items()})
        losses, gdig, ddig = [], [], []
        for step in range(steps):
            t._gan_loop(lr_img, hr_img, step)
            res = orc.gan_step(lr_img, hr_img)
            assert abs(res[4] - logged[-1]) <= (1e-6 if step == 0 else 5e-5) * max(1, abs(res[4])), (res, logged[-1])
            losses.append(res)
            gdig.append(np.stack([tensor_digest(v) for k, v in sorted(t.generator.state_dict().items())]))
            ddig.append(np.stack([tensor_digest(v) for k, v in sorted(t.discriminator.state_dict().items())]))
            print(f'  esrgan {tag} gan step {step}: disc {res[0]:.6f} pixel {res[1]:.6f} content {res[2]:.6f} adv {res[3]:.6f} '
                  f'gen {res[4]:.6f}')
        check(f'ESRGAN {tag} G conv4.weight', orc.g['conv4.weight'].detach(), t.generator.state_dict()['conv4.weight'], tol=1e-4)
        check(f'ESRGAN {tag} D classifier.0.weight', orc.d['classifier.0.weight'].detach(),
              t.discriminator.state_dict()['classifier.0.weight'], tol=1e-4)
        return t, np.array(losses), np.array(logged), np.stack(gdig), np.stack(ddig)

    lr_img, hr_img = seeded_input((2, 3, 32, 32), 61), seeded_input((2, 3, 128, 128), 62)
    t, losses, logged, gdig, ddig = run(2, lr_img, hr_img, 3, 'b2')
    out.update(low_res=lr_img.numpy(), high_res=hr_img.numpy(), gan_losses=losses, gan_ref_gen_losses=logged,
               gan_g_digest=gdig, gan_d_digest=ddig, gs_keys=np.array(sorted(t.generator.state_dict().keys())),
               ds_keys=np.array(sorted(t.discriminator.state_dict().keys())))
    # BASELINE config 4's geometry (23 RRDBs, 128x128 crops) at batch 4, fp32: seeded inputs, only losses + digests stored
    t, losses, logged, gdig, ddig = run(4, seeded_input((4, 3, 32, 32), 161), seeded_input((4, 3, 128, 128), 162), 1, 'b4')
    out.update(b4_seeds=np.array([161, 162]), b4_gan_losses=losses[0], b4_gan_ref_gen_loss=np.float64(logged[0]),
               b4_gan_g_digest=gdig[0], b4_gan_d_digest=ddig[0])
    np.savez_compressed(os.path.join(OUT, 'esrgan.npz'), **out)
    os.chdir(ROOT)


def gen_pil():
    """What the reference's data pipeline does to a crop (torchsr/dataset.py:88-99,121-125): ToPILImage ->
    Resize(crop/4, BICUBIC) -> ToTensor, i.e. PIL's antialiased Keys bicubic on the 8-bit image.  A handful of
    (uint8 crop -> PIL low-resolution image) pairs pin ``srx_bicubic_down`` to the reference's resampler."""
    from PIL import Image
    rng = np.random.Generator(np.random.PCG64(2024))
    crops, lows = [], []
    yy, xx = np.mgrid[0:96, 0:96]
    for i in range(4):
        if i < 2:   # noise: the worst case for 8-bit intermediate rounding
            a = rng.integers(0, 256, (96, 96, 3), dtype=np.uint8)
        else:       # smooth structure with edges, like a photograph
            base = 127 + 90 * np.sin(xx / (5.0 + 3 * i)) * np.cos(yy / (7.0 + i)) + 30 * ((xx // 16 + yy // 24) % 2)
            a = np.clip(np.stack([base, base[::-1], base.T], -1) + rng.normal(0, 6, (96, 96, 3)), 0, 255).astype(np.uint8)
        low = np.asarray(Image.fromarray(a).resize((24, 24), Image.BICUBIC))
        crops.append(a)
        lows.append(low)
    np.savez_compressed(os.path.join(OUT, 'pil_bicubic.npz'), crops=np.stack(crops), lows=np.stack(lows))
    print('  PIL', Image.__version__, 'BICUBIC x1/4:', np.stack(lows).shape)


def gen_checkpoint():
    """What the reference WRITES as a checkpoint (srgan/trainer.py:233-258, esrgan/trainer.py:233-258): the dict of
    ``_model_state(epoch, phase)`` of its own trainers, single-process and wrapped the way a distributed run wraps the
    generator (``module.`` key prefix, srgan/trainer.py:142-149; test.py:44-52 strips it).  The fixture holds the key
    lists and a digest per entry -- the loaders of this package are tested against a file rebuilt from them."""
    os.chdir(REF)
    out = {}
    t = _reference_srgan_trainer(2)
    ck = t._model_state(3, 'srgan-gan')
    out['srgan_top_keys'] = np.array(list(ck.keys()))
    out['srgan_epoch'], out['srgan_phase'] = np.int64(ck['epoch']), np.array(ck['phase'])
    out['srgan_state_keys'] = np.array(list(ck['state'].keys()))
    out['srgan_state_digest'] = np.stack([tensor_digest(v.float()) for v in ck['state'].values()])
    out['srgan_state_dtypes'] = np.array([str(v.dtype) for v in ck['state'].values()])
    wrapped = nn.DataParallel(t.generator)  # same key prefix as DistributedDataParallel; needs no process group
    t.generator = wrapped
    ckw = t._model_state(7, 'srgan-psnr')
    out['srgan_wrapped_state_keys'] = np.array(list(ckw['state'].keys()))
    assert all(k.startswith('module.') for k in ckw['state'])
    from torchsr.esrgan.trainer import ESRGANTrainer
    args = Namespace(disable_amp=True, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1)
    te = ESRGANTrainer('cpu', args, [], [], 2, 2, distributed=False)
    te.generator.load_state_dict(step_state(te.generator.state_dict(), 'esrgan.G'))
    cke = te._model_state(2, 'esrgan-gan')
    out['esrgan_top_keys'] = np.array(list(cke.keys()))
    out['esrgan_state_keys'] = np.array(list(cke['state'].keys()))
    out['esrgan_state_digest'] = np.stack([tensor_digest(v.float()) for v in cke['state'].values()])
    np.savez_compressed(os.path.join(OUT, 'checkpoint.npz'), **out)
    print('  checkpoint: srgan', len(ck['state']), 'entries, esrgan', len(cke['state']), 'entries; top-level keys', list(ck.keys()))


if __name__ == '__main__':
    import warnings
    warnings.simplefilter('ignore')
    install_torchvision_stub()
    if sys.argv[1:] == ['checkpoint']:
        gen_checkpoint()
        sys.exit(0)
    if sys.argv[1:] == ['steps']:
        gen_steps()
        sys.exit(0)
    print('generator'); gen_generator()
    print('discriminator'); gen_discriminator()
    print('vgg19'); gen_vgg()
    print('train steps'); gen_steps()
    print('esrgan'); gen_esrgan()
    print('pil'); gen_pil()
    print('checkpoint'); gen_checkpoint()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
