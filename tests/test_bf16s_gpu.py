"""bf16 STORAGE of activations inside a frozen conv stack under autocast (csrc/gconv.hip ``gconv_kernel<..., PR = 2>``,
``srx_conv3x3_bf16s_*``; the perceptual loss of the ESRGAN step, esrgan/trainer.py:461-467, srgan/loss.py:52-53).

The arithmetic is the bf16-product recipe of ``precision = 1``: bf16-rounded operands, fp32 accumulation.  Each entry point
is held against stock torch fp64 ON THE ROUNDED OPERANDS -- fp32 outputs at 2e-5 of the output scale (fp32 summation order),
bf16 outputs at half a bf16 ulp of each value (2^-8 relative, plus the fp32 floor) -- and the stack as a whole against the
same stack with fp32-stored activations (developer switch), which rounds the same values in its loaders."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


def bf(t):
    return t.to(torch.bfloat16)


def within_half_ulp(got16, want64):
    """|got - want| <= 2^-8 |want| + 1e-5 of the scale (rounding of the fp32 sum to bf16)"""
    g, w = got16.double().cpu(), want64.double().cpu()
    return bool(((g - w).abs() <= 2.0 ** -8 * w.abs() + 1e-5 * w.abs().max()).all())


CASES = [
    # N, H, W, Cin, Cout
    (2, 16, 16, 64, 64),
    (2, 24, 24, 256, 256),
    (4, 8, 8, 512, 512),     # few pixels, deep K: the planner splits K over the last round (fix-up pass writes the bf16 output)
    (1, 32, 32, 128, 256),
    (3, 10, 6, 64, 128),     # ragged: 180 pixels, a partly filled tile
    (16, 16, 16, 256, 512),
]


@pytest.mark.parametrize('cfg', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_bf16_storage_conv_forward_and_data_gradient(dev, cfg):
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn(n, cin, h, w, generator=g).relu())
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    dy = bf(torch.randn(n, cout, h, w, generator=g))
    w16 = bf(wt).double()
    xr = x.double().requires_grad_(True)
    pre = TF.conv2d(xr, w16, None, padding=1)
    yr = TF.relu(pre.detach() + bias.double().view(1, -1, 1, 1))
    dx_plain = torch.autograd.grad(pre, xr, dy.double())[0]
    dx_masked = dx_plain * (x.double() > 0)

    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 1)
    dref = C.byref(d)
    assert L.srx_conv3x3_bf16s_applicable(dref) == 1
    nb = L.srx_conv3x3_bf16s_packed_bytes(dref)
    assert nb == cout * cin * 9 * 2
    s = torch.cuda.current_stream().cuda_stream
    wg = wt.to(dev)
    wf = torch.empty(nb // 2, dtype=torch.bfloat16, device=dev)
    wb = torch.empty(nb // 2, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_conv3x3_bf16s_pack', dref, wg.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    assert torch.equal(wf.view(cout, 3, 3, cin).cpu(), bf(wt).permute(0, 2, 3, 1))
    assert torch.equal(wb.view(cin, 3, 3, cout).cpu(), bf(wt).flip(2, 3).permute(1, 2, 3, 0))

    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    bg = bias.to(dev)
    nws = L.srx_conv3x3_bf16s_ws_floats(dref, 0)
    ws = torch.empty(max(nws, 4), device=dev)
    y32 = torch.empty(n, h, w, cout, device=dev)
    y16 = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_conv3x3_bf16s_fwd', dref, xg.data_ptr(), wf.data_ptr(), bg.data_ptr(), 1, y32.data_ptr(), 0, ws.data_ptr(), nws, s)
    _lib.call('srx_conv3x3_bf16s_fwd', dref, xg.data_ptr(), wf.data_ptr(), bg.data_ptr(), 1, y16.data_ptr(), 1, ws.data_ptr(), nws, s)
    assert rel(y32.permute(0, 3, 1, 2), yr) < 2e-5
    assert within_half_ulp(y16.permute(0, 3, 1, 2), yr)
    # without bias / activation
    _lib.call('srx_conv3x3_bf16s_fwd', dref, xg.data_ptr(), wf.data_ptr(), None, 0, y32.data_ptr(), 0, ws.data_ptr(), nws, s)
    assert rel(y32.permute(0, 3, 1, 2), pre) < 2e-5

    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    nwb = L.srx_conv3x3_bf16s_ws_floats(dref, 1)
    wsb = torch.empty(max(nwb, 4), device=dev)
    dx32 = torch.empty(n, h, w, cin, device=dev)
    dx16 = torch.empty(n, h, w, cin, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_conv3x3_bf16s_bwd_data', dref, dyg.data_ptr(), wb.data_ptr(), None, dx32.data_ptr(), 0, wsb.data_ptr(), nwb, s)
    assert rel(dx32.permute(0, 3, 1, 2), dx_plain) < 2e-5
    _lib.call('srx_conv3x3_bf16s_bwd_data', dref, dyg.data_ptr(), wb.data_ptr(), xg.data_ptr(), dx32.data_ptr(), 0, wsb.data_ptr(), nwb, s)
    assert rel(dx32.permute(0, 3, 1, 2), dx_masked) < 2e-5
    _lib.call('srx_conv3x3_bf16s_bwd_data', dref, dyg.data_ptr(), wb.data_ptr(), xg.data_ptr(), dx16.data_ptr(), 1, wsb.data_ptr(), nwb, s)
    assert within_half_ulp(dx16.permute(0, 3, 1, 2), dx_masked)
    assert bool(((dx16.permute(0, 3, 1, 2).cpu().float() == 0) | (x > 0)).all())   # nothing passes a closed ReLU


def test_bf16_storage_refuses_other_layers(dev):
    from torchsr_amd import _lib
    L = _lib.lib()
    for args in ((2, 8, 8, 64, 64, 64, 64, 3, 3, 1, 1, 0, 1, 0.0, 0, 0),      # fp32 arithmetic (precision = 0)
                 (2, 8, 8, 32, 32, 64, 64, 3, 3, 1, 1, 0, 1, 0.0, 0, 1),      # 32 input channels
                 (2, 8, 8, 64, 64, 64, 64, 3, 3, 2, 1, 0, 1, 0.0, 0, 1)):     # stride 2
        assert L.srx_conv3x3_bf16s_applicable(C.byref(_lib.Conv2dDesc(*args))) == 0
        assert L.srx_conv3x3_bf16s_packed_bytes(C.byref(_lib.Conv2dDesc(*args))) == 0


def test_pools_and_first_layer_with_bf16_outputs(dev):
    """The pools compare fp32 values and round what they hand on; the 3 -> 64 first layer's bf16 output is the rounding of
    its fp32 output; the topmost activation backward likewise -- exact comparisons."""
    from torchsr_amd import _lib, functional as F
    from torchsr_amd.layers import Conv2d, set_conv_precision
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(9)
    n, h, w, c = 3, 12, 20, 64
    x = torch.randn(n, h, w, c, generator=g).relu().to(dev)
    y16 = torch.empty(n, h // 2, w // 2, c, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_maxpool2x2_fwd_to_bf16', x.data_ptr(), y16.data_ptr(), n, h, w, c, s)
    y32 = torch.empty(n, h // 2, w // 2, c, device=dev)
    _lib.call('srx_maxpool2x2_fwd', x.data_ptr(), y32.data_ptr(), n, h, w, c, s)
    assert torch.equal(y16, y32.to(torch.bfloat16))
    dy16 = bf(torch.randn(n, h // 2, w // 2, c, generator=g)).to(dev)
    dx16 = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_maxpool2x2_relu_bwd_bf16', dy16.data_ptr(), x.data_ptr(), dx16.data_ptr(), n, h, w, c, s)
    dx32 = torch.empty(n, h, w, c, device=dev)
    dy32 = dy16.float()
    _lib.call('srx_maxpool2x2_relu_bwd', dy32.data_ptr(), x.data_ptr(), dx32.data_ptr(), n, h, w, c, s)
    assert torch.equal(dx16.float(), dx32)
    gr = torch.randn(n, h, w, c, generator=g).to(dev)
    a16 = torch.empty(n, h, w, c, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_act_bwd_from_out_to_bf16', gr.data_ptr(), x.data_ptr(), a16.data_ptr(), gr.numel(), _lib.ACT_RELU, 0.0, s)
    assert torch.equal(a16, (gr * (x > 0)).to(torch.bfloat16))
    # first layer
    torch.manual_seed(1)
    conv = Conv2d(3, 64, 3, 1, 1, act=F.ACT_RELU).to(dev)
    set_conv_precision(conv, 'bf16')
    img = F.to_nhwc(torch.rand(2, 3, 24, 40, device=dev), 4)
    st = conv._st
    d = st.desc(2, 24, 40)
    st.pack(conv.weight, d)
    bias = conv.bias.detach()
    ref = torch.empty(2, 24, 40, 64, device=dev)
    _lib.call('srx_conv2d_fwd', C.byref(d), img.data_ptr(), st.wpk_fwd.data_ptr(), bias.data_ptr(), ref.data_ptr(), None, None, 0, s)
    o16 = torch.empty(2, 24, 40, 64, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_conv2d_fwd_first3_to_bf16', C.byref(d), img.data_ptr(), st.wpk_fwd.data_ptr(), bias.data_ptr(), o16.data_ptr(), s)
    assert torch.equal(o16, ref.to(torch.bfloat16))


def test_vgg_stack_with_bf16_storage_equals_fp32_storage(dev, monkeypatch):
    """The perceptual-loss node under autocast with bf16-stored activations against the same node with fp32-stored ones
    (developer switch): both multiply the same bf16-rounded operands; they differ in the order of their fp32 sums only, and an
    intermediate one fp32 ulp apart rounds to the other bf16 neighbour now and then."""
    from torchsr_amd import _dev, functional as F
    from torchsr_amd.layers import set_conv_precision
    from torchsr_amd.srgan.loss import VGGLoss
    torch.manual_seed(0)
    vgg = VGGLoss(weights='random').to(dev)
    set_conv_precision(vgg, 'bf16')
    src = torch.rand(2, 3, 64, 64, device=dev)
    tgt = torch.rand(2, 3, 64, 64, device=dev)
    out, names = {}, {False: [], True: []}
    real_call = F.call
    for off in (False, True):
        monkeypatch.setattr(_dev, 'NO_BF16S', off)
        monkeypatch.setattr(F, 'call', lambda name, *a, _n=names[off]: (_n.append(name), real_call(name, *a))[1])
        s = src.clone().requires_grad_(True)
        loss = vgg(s, tgt)
        loss.backward()
        out[off] = (loss.detach().clone(), s.grad.clone())
    monkeypatch.setattr(F, 'call', real_call)
    assert names[False].count('srx_conv3x3_bf16s_fwd') == 15 and names[False].count('srx_conv3x3_bf16s_bwd_data') == 15
    assert names[False].count('srx_maxpool2x2_fwd_to_bf16') == 4 and 'srx_conv3x3_bf16s_fwd' not in names[True]
    # Two evaluations of one bf16 recipe part ways chaotically: an fp32 sum one ulp apart (another tile plan, another order)
    # rounds to the other bf16 neighbour in one element in 2^15, that 2^-8 step flips more roundings in the next layer, and
    # after sixteen layers the two agree to the bf16 noise level only (measured rms 1e-7 after conv1_2, 1.5e-5 after conv2_1,
    # 1.5e-3 after conv3_4, 5.8e-3 at the end: tools/experiments/diag_bf16s.py).  So: the first layers exactly, the whole
    # stack at that level.
    layers = vgg._stack()
    src4, tgt4 = F.to_nhwc(src, 4), F.to_nhwc(tgt, 4)
    for k, tol in ((2, 1e-6), (4, 2e-3), (len(layers), None)):
        fs = {}
        for off in (False, True):
            monkeypatch.setattr(_dev, 'NO_BF16S', off)
            with torch.no_grad():
                fs[off] = torch.cat(F.frozen_conv_stack(src4, tgt4, layers[:k])).double()
        if tol is not None:
            assert rel(fs[False], fs[True]) < tol, k
        else:
            assert float((fs[False] - fs[True]).norm() / fs[True].norm()) < 2e-2
    assert rel(out[False][0], out[True][0]) < 2e-2
    # (the input gradient also passes four max-pools: where the two forward passes differ at the 5e-3 level a window's
    # maximum moves to the neighbouring pixel in a per cent or two of the windows and takes the whole gradient of that window
    # with it -- measured 0.23 of the gradient's norm; the backward chain itself is pinned in the test below)
    a, b = out[False][1].double().flatten(), out[True][1].double().flatten()
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.9
    # and the fp32 step is untouched by the switch
    set_conv_precision(vgg, 'fp32')
    monkeypatch.setattr(_dev, 'NO_BF16S', False)
    s = src.clone().requires_grad_(True)
    vgg(s, tgt).backward()
    assert torch.isfinite(s.grad).all()


class _Round(torch.autograd.Function):
    """bf16 rounding of the value (fwd) and / or of the gradient (bwd): where the bf16-storage stack stores a tensor."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(torch.bfloat16).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.bwd else g), None, None


class _FirstConv(torch.autograd.Function):
    """The 3 -> 64 layer: bf16 products forward, exact fp32 data gradient (the 3-channel kernels)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        r = lambda t: t.to(torch.bfloat16).to(t.dtype)  # noqa: E731
        return TF.conv2d(r(x), r(w), b, padding=1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        return torch.nn.grad.conv2d_input(x.shape, w, g, padding=1), None, None


def test_bf16_storage_backward_chain_against_the_recipe_in_fp64(dev):
    """conv1_1, conv1_2, pool, conv2_1, conv2_2 of the stack with bf16 storage against torch autograd in fp64 that rounds values
    and gradients exactly where the stack stores them (``_Round``): the loss, and the input gradient through two storage-mode
    data gradients with folded ReLU masks, the pool backward on fp32 activations and the fp32 first-layer gradient.  Five layers
    deep the bf16 flips between an fp32-sum and an fp64-sum evaluation are rare (one rounding in 2^15) and do not cascade."""
    from torchsr_amd import functional as F
    from torchsr_amd.layers import set_conv_precision
    from torchsr_amd.srgan.loss import VGGLoss
    torch.manual_seed(3)
    vgg = VGGLoss(weights='random').to(dev)
    set_conv_precision(vgg, 'bf16')
    layers = vgg._stack()[:5]
    assert [k for k, _ in layers] == ['conv', 'conv', 'pool', 'conv', 'conv']
    src = torch.rand(2, 3, 32, 48, device=dev)
    coef = torch.randn(2, 16, 24, 128, device=dev)
    s4 = F.to_nhwc(src, 4).requires_grad_(True)
    assert F._bf16_stack_ok(layers, tuple(s4.shape))
    fs, _ = F.frozen_conv_stack(s4, None, layers)
    loss = (fs * coef).sum()
    loss.backward()
    got = s4.grad[..., :3].permute(0, 3, 1, 2).double().cpu()

    r = lambda t: t.to(torch.bfloat16).to(t.dtype)  # noqa: E731
    convs = [m for k, m in layers if k == 'conv']
    w = [m.weight.detach().double().cpu() for m in convs]
    b = [m.bias.detach().double().cpu() for m in convs]
    x = src.double().cpu().requires_grad_(True)
    a1 = _Round.apply(TF.relu(_FirstConv.apply(x, w[0], b[0])), True, False)          # stored bf16; its gradient stays fp32
    pre2 = _Round.apply(TF.conv2d(a1, r(w[1]), b[1], padding=1), False, True)         # fp32 (feeds the pool); gradient stored bf16
    p = _Round.apply(TF.max_pool2d(TF.relu(pre2), 2), True, True)
    a3 = _Round.apply(TF.relu(TF.conv2d(p, r(w[2]), b[2], padding=1)), True, True)
    a4 = _Round.apply(TF.relu(TF.conv2d(a3, r(w[3]), b[3], padding=1)), False, True)   # fp32 features; gradient stored bf16
    want_loss = (a4 * coef.permute(0, 3, 1, 2).double().cpu()).sum()
    want_loss.backward()
    assert abs(loss.item() - want_loss.item()) <= 2e-4 * abs(want_loss.item()) + 1e-3
    d = got - x.grad
    assert float(d.norm() / x.grad.norm()) < 2e-3, float(d.norm() / x.grad.norm())
    assert rel(got, x.grad) < 2e-2


@pytest.mark.parametrize('cfg', [(32, 64, 64, 128, 128), (32, 16, 16, 512, 512), (16, 128, 128, 64, 64)], ids=lambda c: 'x'.join(map(str, c)))
def test_bf16_storage_conv_equals_the_fp32_stored_form_at_the_step_sizes(dev, cfg):
    """At the sizes of the ESRGAN step (VGG19 on 128 x 128 crops, batch 32 forward / 16 backward) the bf16-storage conv against
    ``srx_conv2d_fwd`` / ``srx_conv2d_bwd_data_act`` with ``precision = 1`` on the same bf16-representable operands stored as
    fp32: the same products, fp32 sums in a possibly different order -- 2e-5 of the output scale."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(19)
    x16 = bf(torch.randn(n, h, w, cin, generator=g).relu()).to(dev)
    dy16 = bf(torch.randn(n, h, w, cout, generator=g)).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 1)
    dref = C.byref(d)
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(dref), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(dref), device=dev)
    _lib.call('srx_conv2d_pack', dref, wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    nb = L.srx_conv3x3_bf16s_packed_bytes(dref) // 2
    pf, pb = torch.empty(nb, dtype=torch.bfloat16, device=dev), torch.empty(nb, dtype=torch.bfloat16, device=dev)
    _lib.call('srx_conv3x3_bf16s_pack', dref, wt.data_ptr(), pf.data_ptr(), pb.data_ptr(), s)

    def ws_for(nfl):
        return torch.empty(max(int(nfl), 4), device=dev)
    x32, dy32 = x16.float(), dy16.float()
    ya, yb = torch.empty(n, h, w, cout, device=dev), torch.empty(n, h, w, cout, device=dev)
    nws = L.srx_conv2d_fwd_ws_floats(dref)
    _lib.call('srx_conv2d_fwd', dref, x32.data_ptr(), wf.data_ptr(), bias.data_ptr(), ya.data_ptr(), None, ws_for(nws).data_ptr(), nws, s)
    nw2 = L.srx_conv3x3_bf16s_ws_floats(dref, 0)
    _lib.call('srx_conv3x3_bf16s_fwd', dref, x16.data_ptr(), pf.data_ptr(), bias.data_ptr(), 1, yb.data_ptr(), 0, ws_for(nw2).data_ptr(), nw2, s)
    assert rel(yb, ya) < 2e-5
    da, db = torch.empty(n, h, w, cin, device=dev), torch.empty(n, h, w, cin, device=dev)
    nws = L.srx_conv2d_bwd_data_ws_floats(dref)
    _lib.call('srx_conv2d_bwd_data_act', dref, dy32.data_ptr(), wb.data_ptr(), x32.data_ptr(), 0.0, 0, cin, 0, da.data_ptr(),
              ws_for(nws).data_ptr(), nws, s)
    nw2 = L.srx_conv3x3_bf16s_ws_floats(dref, 1)
    _lib.call('srx_conv3x3_bf16s_bwd_data', dref, dy16.data_ptr(), pb.data_ptr(), x16.data_ptr(), db.data_ptr(), 0, ws_for(nw2).data_ptr(), nw2, s)
    assert rel(db, da) < 2e-5
