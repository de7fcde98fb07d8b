"""The discriminator head + adversarial loss node (csrc/head.hip, functional.gan_head) against stock torch autograd on CPU in
fp64: srgan/discriminator.py:65-69 + srgan/trainer.py:446-448,456-457; esrgan/discriminator.py:73-76 + esrgan/trainer.py:451-453,
468-469.  Tolerance 2e-5 of each tensor's scale (fp32 sums of <= 18432 terms against fp64)."""
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel(a, b, floor=1e-9):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return ((a - b).abs().max() / b.abs().max().clamp_min(floor)).item()


def reference(mode, x, w1, b1, w2, b2, n, weight, shift, addend):
    """the reference's expressions, fp64"""
    h = TF.leaky_relu(TF.linear(x, w1, b1), 0.2)
    z = TF.linear(h, w2, b2)
    if mode == 0:
        p = torch.sigmoid(z)
        return TF.binary_cross_entropy(p[:n], torch.ones_like(p[:n])) + TF.binary_cross_entropy(p[n:], torch.zeros_like(p[n:]))
    if mode == 1:
        p = torch.sigmoid(z)
        return addend + weight * TF.binary_cross_entropy(p, torch.ones_like(p))
    if mode == 2:
        real, fake = z[:n], z[n:]
        a = TF.binary_cross_entropy_with_logits(real - fake.mean(), torch.ones_like(real))
        c = TF.binary_cross_entropy_with_logits(fake - real.mean(), torch.zeros_like(fake))
        return (a + c) / 2
    return addend + weight * TF.binary_cross_entropy_with_logits(z - shift, torch.ones_like(z))


@pytest.mark.parametrize('mode,b,k,j,n', [(0, 32, 18432, 1024, 16), (1, 16, 18432, 1024, 0), (0, 4, 512, 1024, 2),
                                          (2, 32, 8192, 100, 16), (3, 16, 8192, 100, 0), (2, 6, 256, 100, 3), (0, 24, 300 * 4, 40, 9)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize('direct', [False, True], ids=['returned', 'into_grad'])
def test_gan_head(dev, mode, b, k, j, n, direct):
    from torchsr_amd import functional as F
    from torchsr_amd.layers import Linear
    g = torch.Generator().manual_seed(7 + mode)
    x = torch.randn(b, k, generator=g)
    lin1, lin2 = Linear(k, j), Linear(j, 1)
    with torch.no_grad():
        lin1.weight.copy_(torch.randn(j, k, generator=g) * (2.0 / k) ** 0.5)
        lin1.bias.copy_(torch.randn(j, generator=g) * 0.1)
        lin2.weight.copy_(torch.randn(1, j, generator=g) * (4.0 / j) ** 0.5)
        lin2.bias.copy_(torch.randn(1, generator=g) * 0.1)
    weight = 0.001 if mode == 1 else 0.005
    shift = torch.tensor(0.3) if mode == 3 else None
    addend = torch.tensor(1.25) if mode in (1, 3) else None
    # fp64 reference
    xr = x.double().requires_grad_(True)
    ps = [p.detach().double().requires_grad_(True) for p in (lin1.weight, lin1.bias, lin2.weight, lin2.bias)]
    ar = None if addend is None else addend.double().requires_grad_(True)
    ref = reference(mode, xr, *ps, n, weight, None if shift is None else shift.double(), ar)
    up = 0.7
    (ref * up).backward()
    # the node
    lin1, lin2 = lin1.to(dev), lin2.to(dev)
    xg = x.to(dev).requires_grad_(True)
    ag = None if addend is None else addend.to(dev).requires_grad_(True)
    params = [lin1.weight, lin1.bias, lin2.weight, lin2.bias]
    old = F.direct_grads[0]
    F.direct_grads[0] = direct
    try:
        if direct:  # the trainers' form: gradients ACCUMULATE into existing .grad buffers
            for p in params:
                p.grad = torch.full_like(p, 0.5)
        loss, aux = F.gan_head(xg, lin1, lin2, mode, n_first=n, slope=0.2, adv_weight=weight,
                               shift=None if shift is None else shift.to(dev), addend=ag)
        loss.backward(torch.tensor(up, device=dev))
    finally:
        F.direct_grads[0] = old
    assert rel(loss, ref) < 2e-5
    assert rel(xg.grad, xr.grad) < 2e-5
    # scale of a logit gradient: the relativistic discriminator loss does not depend on the last bias at all (the mean of
    # the other half is subtracted from every logit), so that gradient is an exact 0 that fp32 meets to rounding of its terms
    zscale = up / b
    for p, r in zip(params, ps):
        got = p.grad - 0.5 if direct else p.grad
        assert rel(got, r.grad, floor=zscale) < (2e-4 if direct else 2e-5), (tuple(p.shape))
    if ag is not None:
        assert rel(ag.grad, ar.grad) < 1e-6
        assert rel(aux[1], (ref - addend.double()) / weight) < 2e-4  # the adversarial term itself


def test_gan_head_without_weight_gradients(dev):
    """the discriminator pass of the generator update (srgan/trainer.py:456 under layers.no_weight_grad): input gradient only"""
    from torchsr_amd import functional as F
    from torchsr_amd.layers import Linear, no_weight_grad
    torch.manual_seed(3)
    lin1, lin2 = Linear(256, 64).to(dev), Linear(64, 1).to(dev)
    x = torch.randn(8, 256, device=dev, requires_grad=True)
    c = torch.tensor(2.0, device=dev, requires_grad=True)
    with no_weight_grad():
        loss, aux = F.gan_head(x, lin1, lin2, F.HEAD_SRGAN_G, adv_weight=0.001, addend=c)
    loss.backward()
    assert x.grad is not None and c.grad is not None and float(c.grad) == 1.0
    assert all(p.grad is None for p in list(lin1.parameters()) + list(lin2.parameters()))
    xr = x.detach().cpu().double().requires_grad_(True)
    ref = reference(1, xr, *[p.detach().cpu().double() for p in (lin1.weight, lin1.bias, lin2.weight, lin2.bias)], 0, 0.001, None,
                    torch.tensor(2.0, dtype=torch.float64))
    ref.backward()
    assert rel(x.grad, xr.grad) < 2e-5 and rel(loss, ref) < 1e-6
