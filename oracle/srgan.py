"""ORACLE (test infrastructure, never the product path).

CPU restatement, in plain ``torch.nn.functional`` calls on NCHW fp32 tensors, of the
SRGAN hot path of roclark/torchsr.  Every function cites the reference lines it
follows (paths relative to /root/reference).  State is a flat ``dict`` keyed exactly
like the reference modules' ``state_dict()``; BatchNorm running statistics in it are
updated in place, as ``nn.BatchNorm2d`` does in training mode.

Pinned: ``oracle/gen_golden.py`` (run in the build container, where the reference is
importable) asserts these functions against the reference's own modules and writes
``tests/golden/*.npz``; ``tests/test_cpu.py`` re-checks the oracle against those files
everywhere.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this package.
"""
import contextlib
import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

State = Dict[str, Tensor]

# ------------------------------------------------------------------ conv arithmetic of an autocast region
# The reference's autocast regions (srgan/trainer.py:382-385; esrgan/trainer.py:384,446,461) run their convs in half
# precision.  BASELINE config 4 asks for bf16 on MI355X, which the product implements as "bf16 products": both operands
# of every multiplication are rounded to bf16 (round to nearest even), products and sums stay fp32 -- in the forward
# pass, in the stride-1 data gradient (dy and W rounded) and in the weight gradient (x and dy rounded); strided data
# gradients and the layers with a 3-channel side keep exact fp32 operands.  ``bf16_products()`` switches this oracle to
# the same recipe so that an autocast step can be pinned at ~1e-3 instead of "within bf16 rounding of the fp32 step".
_BF16 = [False]
_BF16_THIN_OUT = [False]  # inference (test.upscale(precision='bf16')): the 64 -> 3 output conv rounds its operands too
# inference, round 4: between the generator's first and last conv the product STORES its 64-channel activations as bf16
# (csrc/c64.hip).  For a conv's own operands that is the rounding it applied anyway; what changes is that the skip inputs
# (`x + ...` of a residual block, `conv1 + conv2` of the generator) are rounded too.  ``storage=True`` rounds every such
# tensor where the product writes it.
_BF16_STORAGE = [False]
# round 5 (a yardstick, not a product mode): the products of the bf16-rounded operands summed EXACTLY (fp64, one rounding to fp32
# at the end) instead of in fp32 in torch's order.  Two evaluations of one bf16 recipe differ by their fp32 summation order; the
# exact-sum result is the recipe's own value, and an implementation's distance from it is judged against the distance of this
# oracle's fp32-sum evaluation from it (tests/test_esrgan_gpu.py: the rule DESIGN.md section 4 uses for the fp32 kinks).
_BF16_EXACT_SUMS = [False]


@contextlib.contextmanager
def bf16_products(thin_out: bool = False, storage: bool = False, exact_sums: bool = False):
    old = _BF16[0], _BF16_THIN_OUT[0], _BF16_STORAGE[0], _BF16_EXACT_SUMS[0]
    _BF16[0], _BF16_THIN_OUT[0], _BF16_STORAGE[0], _BF16_EXACT_SUMS[0] = True, thin_out, storage, exact_sums
    try:
        yield
    finally:
        _BF16[0], _BF16_THIN_OUT[0], _BF16_STORAGE[0], _BF16_EXACT_SUMS[0] = old


def _r(t: Tensor) -> Tensor:
    return t.to(torch.bfloat16).to(t.dtype)


def _stored(t: Tensor) -> Tensor:
    """An activation the bf16-native inference chain writes to HBM (identity outside ``bf16_products(storage=True)``)."""
    return _r(t) if (_BF16[0] and _BF16_STORAGE[0]) else t


class _ConvBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad):
        cout, cin = w.shape[0], w.shape[1]
        ctx.cfg = (stride, pad, b is not None)
        ctx.thin_in, ctx.thin_out = (cin <= 4 and cout == 64), (cout <= 4 and cin == 64)
        ctx.save_for_backward(x, w)
        ctx.exact = _BF16_EXACT_SUMS[0]
        if ctx.thin_out and not _BF16_THIN_OUT[0]:  # the product's thin forward kernel is exact fp32 (training paths)
            return F.conv2d(x, w, b, stride, pad)
        if ctx.exact:
            return F.conv2d(_r(x).double(), _r(w).double(), None if b is None else b.double(), stride, pad).float()
        return F.conv2d(_r(x), _r(w), b, stride, pad)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_b = ctx.cfg
        dx = dw = db = None
        up = (lambda t: t.double()) if ctx.exact else (lambda t: t)  # exact sums: fp64 accumulation of the same rounded operands
        if ctx.needs_input_grad[0]:
            exact = ctx.thin_in  # (round 4: the strided data gradients multiply bf16 operands too)
            dx = torch.nn.grad.conv2d_input(x.shape, up(w if exact else _r(w)), up(dy if exact else _r(dy)), stride=stride,
                                            padding=pad).to(x.dtype)
        if ctx.needs_input_grad[1]:
            exact = ctx.thin_in or ctx.thin_out
            dw = torch.nn.grad.conv2d_weight(up(x if exact else _r(x)), w.shape, up(dy if exact else _r(dy)), stride=stride,
                                             padding=pad).to(w.dtype)
        if has_b and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None


def conv2d(x: Tensor, w: Tensor, b, stride: int, pad: int) -> Tensor:
    """``nn.Conv2d`` of the reference: exact fp32, or bf16 products inside ``bf16_products()``."""
    if _BF16[0]:
        return _ConvBF16.apply(x, w, b, stride, pad)
    return F.conv2d(x, w, b, stride, pad)


NUM_RESIDUAL = 16  # torchsr/srgan/generator.py:20
BN_EPS, BN_MOMENTUM = 1e-5, 0.1  # nn.BatchNorm2d defaults used at srgan/residual.py:65


def _bn(sd: State, p: str, x: Tensor, training: bool) -> Tensor:
    """nn.BatchNorm2d (srgan/residual.py:65,68; generator.py:49; discriminator.py:36-60)."""
    if training:
        sd[p + 'num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'], sd[p + 'weight'], sd[p + 'bias'],
                        training, BN_MOMENTUM, BN_EPS)


def residual_block(sd: State, p: str, x: Tensor, training: bool) -> Tensor:
    """ResidualBlock.forward, torchsr/srgan/residual.py:70-92."""
    out = conv2d(x, sd[p + 'conv1.weight'], None, 1, 1)          # :86
    out = _bn(sd, p + 'bn1.', out, training)                       # :87
    out = _stored(F.prelu(out, sd[p + 'prelu.weight']))            # :88
    out = conv2d(out, sd[p + 'conv2.weight'], None, 1, 1)        # :89
    out = _bn(sd, p + 'bn2.', out, training)                       # :90
    return _stored(out + x)                                        # :91


def subpixel_layer(sd: State, p: str, x: Tensor) -> Tensor:
    """SubpixelConvolutionLayer.forward, torchsr/srgan/residual.py:31-48."""
    out = conv2d(x, sd[p + 'conv.weight'], sd[p + 'conv.bias'], 1, 1)  # :45
    out = F.pixel_shuffle(out, 2)                                        # :46
    return _stored(F.prelu(out, sd[p + 'prelu.weight']))                 # :47


def generator_forward(sd: State, x: Tensor, training: bool = True, prefix: str = '') -> Tensor:
    """Generator.forward, torchsr/srgan/generator.py:60-81."""
    p = prefix
    conv1 = _stored(F.prelu(conv2d(x, sd[p + 'conv1.0.weight'], sd[p + 'conv1.0.bias'], 1, 4), sd[p + 'conv1.1.weight']))
    block = conv1
    for i in range(NUM_RESIDUAL):                                  # :76
        block = residual_block(sd, f'{p}blocks.{i}.', block, training)
    conv2 = conv2d(block, sd[p + 'conv2.0.weight'], None, 1, 1)  # :77
    conv2 = _bn(sd, p + 'conv2.1.', conv2, training)
    out = _stored(torch.add(conv1, conv2))                         # :78
    n_up = len({k.split('.')[1] for k in sd if k.startswith(p + 'conv_layers.')})
    for u in range(n_up):                                          # :79
        out = subpixel_layer(sd, f'{p}conv_layers.{u}.', out)
    return conv2d(out, sd[p + 'conv3.weight'], sd[p + 'conv3.bias'], 1, 4)  # :80


D_CONVS = [(2, 3, 2), (5, 6, 1), (8, 9, 2), (11, 12, 1), (14, 15, 2), (17, 18, 1), (20, 21, 2)]  # (conv, bn, stride)


def discriminator_forward(sd: State, x: Tensor, training: bool = True, prefix: str = '') -> Tensor:
    """Discriminator.forward, torchsr/srgan/discriminator.py:71-88 (layers :31-69)."""
    p = prefix
    out = F.leaky_relu(conv2d(x, sd[p + 'features.0.weight'], sd[p + 'features.0.bias'], 1, 1), 0.2)
    for ci, bi, stride in D_CONVS:
        out = conv2d(out, sd[f'{p}features.{ci}.weight'], None, stride, 1)
        out = F.leaky_relu(_bn(sd, f'{p}features.{bi}.', out, training), 0.2)
    out = torch.flatten(out, 1)                                    # :86
    out = F.leaky_relu(F.linear(out, sd[p + 'classifier.0.weight'], sd[p + 'classifier.0.bias']), 0.2)
    out = F.linear(out, sd[p + 'classifier.2.weight'], sd[p + 'classifier.2.bias'])
    return torch.sigmoid(out)                                      # :68


# torchvision.models.vgg cfgs['E']; features[:36] ends after relu5_4 (srgan/loss.py:30-31)
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


def vgg_feature_layout(feature_layer: int = 36):
    """[(index, kind)] of ``vgg19().features[:feature_layer]``."""
    out, idx = [], 0
    for v in VGG19_CFG:
        if v == 'M':
            out.append((idx, 'pool'))
            idx += 1
        else:
            out.append((idx, 'conv'))
            out.append((idx + 1, 'relu'))
            idx += 2
    return out[:feature_layer]


def vgg_features(sd: State, x: Tensor, feature_layer: int = 36, prefix: str = '') -> Tensor:
    """``self.features(x)`` of VGGLoss, torchsr/srgan/loss.py:31,52 (no input normalisation)."""
    out = x
    for idx, kind in vgg_feature_layout(feature_layer):
        if kind == 'conv':
            out = conv2d(out, sd[f'{prefix}{idx}.weight'], sd[f'{prefix}{idx}.bias'], 1, 1)
        elif kind == 'relu':
            out = F.relu(out)
        else:
            out = F.max_pool2d(out, 2, 2)
    return out


def vgg_loss(sd: State, source: Tensor, target: Tensor, prefix: str = '') -> Tensor:
    """VGGLoss.forward, torchsr/srgan/loss.py:36-54."""
    return F.l1_loss(vgg_features(sd, source, prefix=prefix), vgg_features(sd, target, prefix=prefix))


def psnr(super_res: Tensor, high_res: Tensor) -> float:
    """Per-batch PSNR of SRGANTrainer._test, torchsr/srgan/trainer.py:296."""
    return 10 * math.log10(1 / ((super_res - high_res) ** 2).mean().item())


# ------------------------------------------------------------------------- Adam
class Adam:
    """torch.optim.Adam(lr=1e-4, betas=(0.9,0.999)) as configured at srgan/trainer.py:171-185.

    Restated explicitly (single-tensor form of torch/optim/adam.py) so the oracle does not
    depend on optimiser implementation details.
    """

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.params = list(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-(self.lr / bc1))

    def zero_grad(self):
        for p in self.params:
            p.grad = None


def _leaves(sd: State):
    """Make the floating-point non-buffer entries trainable leaves (what nn.Module.parameters() yields)."""
    names = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]
    for k in names:
        sd[k] = sd[k].detach().clone().requires_grad_(True)
    return [sd[k] for k in names]


class SRGANStepOracle:
    """The two inner-loop bodies of SRGANTrainer on CPU.

    ``pretrain_step`` follows torchsr/srgan/trainer.py:376-388 (AMP is a no-op on CPU / disabled);
    ``gan_step`` follows ``_gan_loop``, torchsr/srgan/trainer.py:416-469.
    """

    def __init__(self, g_sd: State, d_sd: State, vgg_sd: State):
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.vgg = {k: v.clone() for k, v in vgg_sd.items()}
        g_params, d_params = _leaves(self.g), _leaves(self.d)
        self.psnr_optimizer = Adam(g_params)   # :171-175
        self.disc_optimizer = Adam(d_params)   # :176-180
        self.gen_optimizer = Adam(g_params)    # :181-185

    def pretrain_step(self, low_res: Tensor, high_res: Tensor) -> float:
        self.psnr_optimizer.zero_grad()                              # :380
        super_res = generator_forward(self.g, low_res, True)         # :383
        loss = F.mse_loss(super_res, high_res)                       # :384
        loss.backward()                                              # :386
        self.psnr_optimizer.step()                                   # :387
        return float(loss.detach())

    def gan_step(self, low_res: Tensor, high_res: Tensor) -> Tuple[float, float, float, float]:
        batch = low_res.size(0)
        real_label = torch.full((batch, 1), 1, dtype=low_res.dtype)  # :439
        fake_label = torch.full((batch, 1), 0, dtype=low_res.dtype)  # :440
        self.disc_optimizer.zero_grad()                              # :442
        super_res = generator_forward(self.g, low_res, True)         # :444
        d_real = F.binary_cross_entropy(discriminator_forward(self.d, high_res, True), real_label)            # :446
        d_fake = F.binary_cross_entropy(discriminator_forward(self.d, super_res.detach(), True), fake_label)  # :447
        disc_loss = d_real + d_fake                                  # :448
        disc_loss.backward()                                         # :450
        self.disc_optimizer.step()                                   # :451
        self.gen_optimizer.zero_grad()                               # :453
        content = vgg_loss(self.vgg, super_res, high_res.detach())   # :455
        adversarial = F.binary_cross_entropy(discriminator_forward(self.d, super_res, True), real_label)      # :456
        gen_loss = content + 0.001 * adversarial                     # :457
        gen_loss.backward()                                          # :468
        self.gen_optimizer.step()                                    # :469
        return tuple(float(v.detach()) for v in (disc_loss, content, adversarial, gen_loss))

    def state(self, which: str) -> State:
        return {k: v.detach() for k, v in getattr(self, which).items()}


class SRGANDataParallelOracle:
    """``world`` data-parallel replicas of the two loop bodies in ONE process (config 3).

    What ``DistributedDataParallel`` does to them (torchsr/srgan/trainer.py:142-157): every rank runs the loop
    body on its own shard with identical weights, the gradients are averaged over the ranks before each
    optimiser step, and BatchNorm is NOT synchronised -- every rank normalises with its own batch statistics
    and keeps its own running statistics (plain ``nn.BatchNorm2d``; ``broadcast_buffers=False`` for D at :156,
    and G's broadcast of rank 0's buffers never influences a training-mode forward).  Because the weights stay
    identical, one set of weight leaves is shared and each rank only owns its BatchNorm buffers: the mean of
    the per-rank losses back-propagates exactly the averaged gradient.
    """

    def __init__(self, g_sd: State, d_sd: State, vgg_sd: State, world: int):
        self.world = world
        self.g = {k: v.clone() for k, v in g_sd.items()}
        self.d = {k: v.clone() for k, v in d_sd.items()}
        self.vgg = {k: v.clone() for k, v in vgg_sd.items()}
        g_params, d_params = _leaves(self.g), _leaves(self.d)
        self.psnr_optimizer, self.disc_optimizer, self.gen_optimizer = Adam(g_params), Adam(d_params), Adam(g_params)

        def per_rank(sd):
            out = [sd]  # rank 0 keeps the dict itself: its buffers are the ones a checkpoint would hold
            for _ in range(1, world):
                out.append({k: (v if v.requires_grad else v.clone()) for k, v in sd.items()})
            return out
        self.g_ranks, self.d_ranks = per_rank(self.g), per_rank(self.d)

    def pretrain_step(self, low_res, high_res):
        """``low_res`` / ``high_res``: one shard per rank."""
        self.psnr_optimizer.zero_grad()
        losses = [F.mse_loss(generator_forward(self.g_ranks[r], low_res[r], True), high_res[r]) for r in range(self.world)]
        (sum(losses) / self.world).backward()
        self.psnr_optimizer.step()
        return [float(v.detach()) for v in losses]

    def gan_step(self, low_res, high_res):
        W = self.world
        ones = [torch.full((low_res[r].size(0), 1), 1.0) for r in range(W)]
        zeros = [torch.full((low_res[r].size(0), 1), 0.0) for r in range(W)]
        self.disc_optimizer.zero_grad()
        sr = [generator_forward(self.g_ranks[r], low_res[r], True) for r in range(W)]
        disc = []
        for r in range(W):
            d_real = F.binary_cross_entropy(discriminator_forward(self.d_ranks[r], high_res[r], True), ones[r])
            d_fake = F.binary_cross_entropy(discriminator_forward(self.d_ranks[r], sr[r].detach(), True), zeros[r])
            disc.append(d_real + d_fake)
        (sum(disc) / W).backward()
        self.disc_optimizer.step()
        self.gen_optimizer.zero_grad()
        gen, parts = [], []
        for r in range(W):
            content = vgg_loss(self.vgg, sr[r], high_res[r].detach())
            adversarial = F.binary_cross_entropy(discriminator_forward(self.d_ranks[r], sr[r], True), ones[r])
            gen.append(content + 0.001 * adversarial)
            parts.append((float(disc[r].detach()), float(content.detach()), float(adversarial.detach()),
                          float(gen[-1].detach())))
        (sum(gen) / W).backward()
        self.gen_optimizer.step()
        return parts
