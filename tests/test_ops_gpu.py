"""Parity of every HIP kernel (through the C ABI) against stock torch fp32/fp64 ops on CPU.

Tolerance: north_star asks for 1e-3 relative fp32; the kernels are exact-fp32 MFMA chains,
so these tests hold them to 2e-4 of the tensor's scale (1e-5 for pure data movement).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-6)).item()


def rnd(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def nhwc(x, cs=None):  # cpu NCHW -> cpu NHWC padded
    n, c, h, w = x.shape
    cs = cs or (c + 3) // 4 * 4
    y = torch.zeros(n, h, w, cs)
    y[..., :c] = x.permute(0, 2, 3, 1)
    return y.contiguous()


def nchw(y, c):
    return y[..., :c].permute(0, 3, 1, 2).contiguous()


def test_layout_roundtrip(dev):
    from torchsr_amd import functional as F
    x = rnd((3, 3, 10, 14), 0)
    y = F.to_nhwc(x.to(dev), 4)
    assert y.shape == (3, 10, 14, 4)
    assert torch.equal(y.cpu(), nhwc(x, 4))
    z = F.to_nchw(y, 3)
    assert torch.equal(z.cpu(), x)
    f = F.flatten_nchw(F.to_nhwc(rnd((2, 8, 3, 5), 1).to(dev)))
    assert torch.equal(f.cpu(), rnd((2, 8, 3, 5), 1).flatten(1))


CONV_CASES = [
    # N, H, W, Cin, Cout, k, s, p, bias, act, shuffle
    (2, 24, 24, 64, 64, 3, 1, 1, False, 0, 0),     # SRGAN residual conv (srgan/residual.py:64)
    (2, 24, 24, 3, 64, 9, 1, 4, True, 0, 0),       # generator conv1 (generator.py:38)
    (2, 12, 12, 64, 256, 3, 1, 1, True, 0, 2),     # sub-pixel conv + PixelShuffle (residual.py:27-28)
    (1, 40, 40, 64, 3, 9, 1, 4, True, 0, 0),       # generator conv3 (generator.py:58)
    (2, 32, 32, 3, 64, 3, 1, 1, True, 2, 0),       # discriminator conv + LeakyReLU (discriminator.py:32-33)
    (2, 32, 32, 64, 64, 3, 2, 1, False, 0, 0),     # stride 2 (discriminator.py:35)
    (2, 16, 16, 128, 256, 3, 1, 1, True, 1, 0),    # VGG conv + ReLU
    (2, 12, 12, 256, 512, 3, 1, 1, False, 0, 0),
    (2, 6, 6, 512, 512, 3, 1, 1, True, 1, 0),      # small M, deep K: split-K path
    (2, 12, 12, 512, 512, 3, 2, 1, False, 0, 0),   # last discriminator conv: split-K + BN stats
    (1, 5, 7, 8, 12, 3, 1, 1, True, 0, 0),         # ragged everything
    (3, 9, 11, 4, 8, 3, 2, 1, False, 0, 0),        # odd sizes, stride 2
    (2, 7, 9, 16, 20, 5, 1, 2, True, 0, 0),        # 5x5
    (1, 8, 8, 32, 32, 1, 1, 0, True, 1, 0),        # 1x1
    (4, 96, 96, 16, 128, 3, 1, 1, False, 0, 0),    # large M: 128x128 tile
    (2, 11, 13, 24, 40, 3, 3, 1, True, 0, 0),      # stride 3
    # 36-pixel row tiles (rowtile.hip): 3x3 64->64 with H*W % 36 == 0
    (16, 24, 24, 64, 64, 3, 1, 1, True, 2, 0),     # the reference batch: 256 workgroups, bias + LeakyReLU
    (3, 12, 12, 64, 64, 3, 1, 1, True, 1, 0),      # tiles span 4 image rows
    (1, 6, 6, 64, 64, 3, 1, 1, False, 0, 0),       # one workgroup, 6-wide image
    (2, 18, 10, 64, 64, 3, 1, 1, False, 0, 0),     # tiles start at any column
    (1, 48, 48, 64, 64, 3, 1, 1, False, 0, 0),     # widest patch that still fits
    (1, 9, 4, 64, 64, 3, 1, 1, True, 0, 0),        # 4-wide image
    (2, 24, 26, 64, 64, 3, 1, 1, False, 0, 0),     # H*W % 36 != 0: generic kernel on the same layer
]


@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_fwd_bwd(dev, case):
    from torchsr_amd import functional as F
    from torchsr_amd.layers import Conv2d
    n, h, w, cin, cout, k, s, p, bias, act, shuffle = case
    seed = hash(case) % 1000
    x = rnd((n, cin, h, w), seed)
    conv = Conv2d(cin, cout, k, s, p, bias=bias, act=act, slope=0.2, shuffle=shuffle)
    with torch.no_grad():
        conv.weight.copy_(rnd(conv.weight.shape, seed + 1) * (2.0 / (cin * k * k)) ** 0.5 * 1.7)
        if bias:
            conv.bias.copy_(rnd(conv.bias.shape, seed + 2) * 0.3)
    wc = conv.weight.detach().clone().requires_grad_(True)
    bc = conv.bias.detach().clone().requires_grad_(True) if bias else None
    xc = x.clone().requires_grad_(True)
    yc = TF.conv2d(xc, wc, bc, s, p)
    pre = yc
    if act == 1:
        yc = TF.relu(yc)
    elif act == 2:
        yc = TF.leaky_relu(yc, 0.2)
    if shuffle:
        yc = TF.pixel_shuffle(yc, 2)
    gy = rnd(yc.shape, seed + 3)
    yc.backward(gy)

    conv = conv.to(dev)
    xg = nhwc(x).to(dev).requires_grad_(True)
    want_stats = (not bias) and act == 0 and not shuffle
    out = conv(xg, want_stats=want_stats)
    yg, part = out if want_stats else (out, None)
    cout_l = cout // 4 if shuffle else cout
    assert rel_err(nchw(yg.cpu(), cout_l), yc) < 2e-4
    if cout_l % 4:
        assert float(yg[..., cout_l:].abs().max()) == 0.0
    if want_stats:
        m = pre.numel() // cout
        s1 = part[:, :, 0].sum(0).cpu().double()
        s2 = part[:, :, 1].sum(0).cpu().double()
        ref1 = pre.detach().double().sum((0, 2, 3))
        ref2 = pre.detach().double().square().sum((0, 2, 3))
        assert ((s1 - ref1).abs().max() / ref2.sqrt().max()).item() < 1e-4
        assert rel_err(s2, ref2) < 2e-4
    yg.backward(nhwc(gy, yg.shape[-1]).to(dev))
    assert rel_err(nchw(xg.grad.cpu(), cin), xc.grad) < 2e-4
    if cin % 4:
        assert float(xg.grad[..., cin:].abs().max()) == 0.0
    assert rel_err(conv.weight.grad, wc.grad) < 2e-4
    if bias:
        assert rel_err(conv.bias.grad, bc.grad) < 2e-4


@pytest.mark.parametrize('plan', ['144,64,1,1', '144,128,1,1', '144,64,4,1', '144,128,3,1'])
@pytest.mark.parametrize('case', [(2, 24, 24, 128, 128, 3, 1, 1, False, 0, 0),    # BN statistics, whole 144-row tiles
                                  (3, 13, 11, 64, 256, 3, 1, 1, True, 2, 0),     # ragged M, bias + LeakyReLU
                                  (2, 12, 12, 64, 256, 3, 1, 1, True, 0, 2),     # PixelShuffle store
                                  (2, 16, 16, 128, 128, 3, 2, 1, False, 0, 0)],  # stride 2 forward
                         ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_144_row_tiles(dev, case, plan, monkeypatch):
    """The 128 + 16 row tile (16 extra rows on 16x16x4 MFMAs) forced on several layers, whole and K-split
    (fix-up kernel); the data gradient of a stride-1 layer takes the same plan."""
    monkeypatch.setenv('SRX_FORCE_PLAN', plan)
    test_conv2d_fwd_bwd(dev, case)


@pytest.mark.parametrize('case', [(2, 24, 24, 64, 64, 3, 1, 1, True), (2, 16, 16, 128, 256, 3, 1, 1, False),
                                  (3, 13, 11, 96, 32, 3, 1, 1, True), (1, 32, 32, 192, 64, 3, 1, 1, True),
                                  (2, 12, 12, 512, 512, 3, 1, 1, False), (2, 20, 20, 64, 64, 3, 2, 1, False),
                                  (2, 32, 32, 128, 128, 3, 2, 1, True), (4, 24, 24, 256, 256, 3, 2, 1, False),
                                  # strided data gradients with bf16 products beyond the discriminator's own shapes (round-4 advice):
                                  # no padding (classes whose grid is larger than the gradient image), odd extents, stride 3
                                  (2, 17, 15, 64, 64, 3, 2, 0, False), (3, 9, 11, 64, 128, 3, 2, 1, True),
                                  (2, 13, 13, 128, 64, 3, 2, 0, True), (2, 11, 13, 64, 64, 3, 3, 1, False)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_bf16_products(dev, case):
    """precision = 1 (the autocast region of the reference): operands rounded to bf16, products exact,
    fp32 accumulation.  Oracle: the fp32 CPU conv on operands rounded the same way -- agreement to fp32
    accumulation-order noise -- the data gradient of a strided layer (its stride-parity classes in one launch) included since
    round 4."""
    from torchsr_amd.layers import Conv2d, set_conv_precision
    n, h, w, cin, cout, k, s, p, bias = case
    seed = hash(case) % 1000
    x = rnd((n, cin, h, w), seed)
    conv = Conv2d(cin, cout, k, s, p, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(rnd(conv.weight.shape, seed + 1) * (2.0 / (cin * k * k)) ** 0.5 * 1.7)
        if bias:
            conv.bias.copy_(rnd(conv.bias.shape, seed + 2) * 0.3)
    set_conv_precision(conv, 'bf16')
    r16 = lambda t: t.detach().bfloat16().float()  # noqa: E731
    bc = conv.bias.detach().clone() if bias else None
    yc = TF.conv2d(r16(x), r16(conv.weight), bc, s, p)
    gy = rnd(yc.shape, seed + 3)
    wr = r16(conv.weight).requires_grad_(False)
    xa = x.clone().requires_grad_(True)
    TF.conv2d(xa, wr, None, s, p).backward(r16(gy))          # dx = conv_transpose(bf16(dy), bf16(W)), any stride
    wa = conv.weight.detach().clone().requires_grad_(True)
    TF.conv2d(r16(x), wa, None, s, p).backward(r16(gy))      # dW = sum bf16(dy) * bf16(x)

    conv = conv.to(dev)
    xg = nhwc(x).to(dev).requires_grad_(True)
    yg = conv(xg)
    assert rel_err(nchw(yg.cpu(), cout), yc) < 2e-5
    yg.backward(nhwc(gy, yg.shape[-1]).to(dev))
    assert rel_err(nchw(xg.grad.cpu(), cin), xa.grad) < 2e-5
    assert rel_err(conv.weight.grad, wa.grad) < 2e-5


@pytest.mark.parametrize('case', [(2, 12, 12, 64, 64, 3, 1, 1, False), (2, 16, 16, 64, 32, 3, 2, 1, True),
                                  (1, 10, 14, 3, 64, 3, 1, 1, True)])
def test_grouped_weight_gradient(dev, case):
    """srx_conv2d_bwd_weight_multi: several problems of one geometry in one launch (the generator's residual
    convs: distinct outputs) and segments of one gradient (the discriminator's real + fake passes: per_out = 2),
    accumulating into existing .grad buffers, with the bias gradient riding along -- against torch's
    conv2d_weight on the CPU, problem by problem."""
    from torchsr_amd import _lib
    n, h, w, cin, cout, k, stride, pad, bias = case
    cs_in, cs_out = (cin + 3) // 4 * 4, (cout + 3) // 4 * 4
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    d = _lib.Conv2dDesc(n, h, w, cin, cs_in, cout, cs_out, k, k, stride, pad, 0, 0, 0.0, 0, 0)
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    arr = lambda ts: (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])  # noqa: E731

    def run(nprob, per_out):
        xs = [rnd((n, cin, h, w), 100 + i) for i in range(nprob)]
        dys = [rnd((n, cout, ho, wo), 200 + i) for i in range(nprob)]
        nout = nprob // per_out
        base_w = [rnd((cout, cin, k, k), 300 + o) for o in range(nout)]   # accumulate = 1: .grad already holds something
        base_b = [rnd((cout,), 400 + o) for o in range(nout)]
        want_w, want_b = [b.clone() for b in base_w], [b.clone() for b in base_b]
        for i in range(nprob):
            o = i // per_out
            want_w[o] += torch.nn.grad.conv2d_weight(xs[i].double(), (cout, cin, k, k), dys[i].double(), stride=stride,
                                                     padding=pad).float()
            want_b[o] += dys[i].sum((0, 2, 3))
        gx = [nhwc(x, cs_in).to(dev) for x in xs]
        gdy = [nhwc(t, cs_out).to(dev) for t in dys]
        gw = [b.to(dev) for b in base_w]
        gb = [b.to(dev) for b in base_b] if bias else None
        nws = L.srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), nprob)
        ws = torch.empty(max(nws, 4), device=dev)
        _lib.call('srx_conv2d_bwd_weight_multi', C.byref(d), nprob, per_out, arr(gx), arr(gdy), arr(gw), 1,
                  arr(gb) if bias else None, ws.data_ptr(), nws, s)
        for o in range(nout):
            assert rel_err(gw[o], want_w[o]) < 2e-4, (nprob, per_out, o)
            if bias:
                assert rel_err(gb[o], want_b[o]) < 2e-4, (nprob, per_out, o)

    run(1, 1)
    run(5, 1)     # five layers, five gradients
    run(6, 2)     # three gradients of two segments each
    run(33, 1)    # the generator's residual tower
    with pytest.raises(RuntimeError, match='whole number of outputs'):
        _lib.call('srx_conv2d_bwd_weight_multi', C.byref(d), 5, 2, None, None, None, 1, None, None, 0, s)


def test_deferred_weight_grads_equal_immediate(dev):
    """functional.deferred_weight_grads: the same .grad buffers as the layer-by-layer backward pass."""
    from torchsr_amd import functional as F
    from torchsr_amd.optim import FlatParams
    from torchsr_amd.srgan.generator import Generator
    torch.manual_seed(11)
    gens = [Generator().to(dev).train() for _ in range(2)]
    gens[1].load_state_dict(gens[0].state_dict())
    flats = [FlatParams(g) for g in gens]
    x = rnd((2, 3, 12, 12), 5, 0.0, 1.0).to(dev)
    old = F.direct_grads[0]
    F.direct_grads[0] = True
    try:
        gens[0](x).square().mean().backward()
        with F.deferred_weight_grads() as q:
            gens[1](x).square().mean().backward()
            assert sum(len(items) for _, items in q.groups.values()) == 37   # 33 residual + conv1, 2 sub-pixel, conv3
            assert float(gens[1].blocks[3].conv1.weight.grad.abs().sum()) == 0.0  # nothing issued yet
    finally:
        F.direct_grads[0] = old
    assert F.wgrad_queue[0] is None
    assert rel_err(flats[1].grad, flats[0].grad) < 1e-5


@pytest.mark.parametrize('case', [(2, 24, 24, 64, 64), (16, 6, 6, 512, 512), (2, 12, 12, 128, 256), (1, 10, 14, 64, 128)])
def test_dgrad_with_folded_activation_backward(dev, case):
    """srx_conv2d_bwd_data_act: dx = conv_transpose(dy, W) * (x > 0 ? 1 : slope) in one kernel, against the two-step
    form on the CPU (whole tiles, K-split tail tiles with their fix-up pass, the 144-row tiles' extra rows)."""
    from torchsr_amd import _lib
    from torchsr_amd.layers import Conv2d
    n, h, w, cin, cout = case
    torch.manual_seed(cin + h)
    conv = Conv2d(cin, cout, 3, 1, 1, bias=False).to(dev)
    x = rnd((n, cin, h, w), 1)                 # "activation output": about half of it <= 0
    dy = rnd((n, cout, h, w), 2)
    xg = nhwc(x).to(dev)
    y = conv(xg)                               # packs the weights
    st = conv._st
    d = st.desc(n, h, w)
    L = _lib.lib()
    for slope in (0.0, 0.2):
        want = torch.nn.grad.conv2d_input((n, cin, h, w), conv.weight.detach().cpu().double(), dy.double(), padding=1)
        want = want * torch.where(x > 0, 1.0, slope).double()
        dx = torch.empty_like(xg)
        nws = L.srx_conv2d_bwd_data_ws_floats(C.byref(d))
        ws = torch.empty(max(nws, 4), device=dev)
        _lib.call('srx_conv2d_bwd_data_act', C.byref(d), nhwc(dy).to(dev).data_ptr(), st.wpk_bwd.data_ptr(), xg.data_ptr(),
                  slope, 0, st.cin_s, 0, dx.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
        assert rel_err(nchw(dx.cpu(), cin), want.float()) < 2e-4, slope
    # a channel range on top of an accumulated gradient (the dense block's shared buffer): dx += dgrad, then the slice
    # [cin - 32, cin) takes the mask
    base = rnd((n, cin, h, w), 3)
    want = base.double() + torch.nn.grad.conv2d_input((n, cin, h, w), conv.weight.detach().cpu().double(), dy.double(), padding=1)
    want[:, cin - 32:] *= torch.where(x[:, cin - 32:] > 0, 1.0, 0.2).double()
    dx = nhwc(base).to(dev)
    _lib.call('srx_conv2d_bwd_data_act', C.byref(d), nhwc(dy).to(dev).data_ptr(), st.wpk_bwd.data_ptr(), xg.data_ptr(),
              0.2, cin - 32, cin, 1, dx.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    assert rel_err(nchw(dx.cpu(), cin), want.float()) < 2e-4
    del y


@pytest.mark.parametrize('case', [(2, 16, 16, 192, 64, 192, 64, 64),     # a dense block's conv5: strided out, dense addend on 64 channels
                                  (2, 16, 16, 64, 32, 64, 192, 64),      # ... conv1: dense out, addend read from the 192-wide buffer
                                  (16, 32, 32, 192, 64, 192, 64, 64),    # config 4's size (K-split tails, several column tiles)
                                  (1, 10, 14, 96, 32, 128, 96, 32),      # ragged image, padded strides
                                  (2, 24, 24, 64, 64, 64, 64, 64)])      # 144-row tiles (extra rows) + plain layout, scales only
@pytest.mark.parametrize('bf16', [False, True])
def test_dgrad_general_epilogue(dev, case, bf16):
    """srx_conv2d_bwd_data_ex: dx = out_scale * conv_transpose(dy, W) + addend_scale * addend on the leading channels,
    the addend with a row stride of its own, then the activation mask on a channel range -- against the same
    arithmetic in fp64 on the CPU."""
    from torchsr_amd import _lib
    from torchsr_amd.layers import Conv2d, set_conv_precision
    n, h, w, cin, cout, ld_dx, ld_add, add_ch = case
    torch.manual_seed(cin + h)
    conv = Conv2d(cin, cout, 3, 1, 1, bias=False).to(dev)
    if bf16:
        set_conv_precision(conv, 'bf16')
    conv(torch.zeros(1, 4, 4, (cin + 3) // 4 * 4, device=dev))   # packs the weights
    st = conv._st
    x = rnd((n, cin, h, w), 1)
    dy = rnd((n, cout, h, w), 2)
    add = rnd((n, add_ch, h, w), 3)
    wt = conv.weight.detach().cpu()
    r = (lambda t: t.bfloat16().double()) if bf16 else (lambda t: t.double())
    mm = torch.nn.grad.conv2d_input((n, cin, h, w), r(wt), r(dy), padding=1)
    d = _lib.Conv2dDesc(n, h, w, cin, ld_dx, cout, (cout + 3) // 4 * 4, 3, 3, 1, 1, 0, 0, 0.0, 0, 1 if bf16 else 0)
    L = _lib.lib()
    nws = L.srx_conv2d_bwd_data_ws_floats(C.byref(d))
    ws = torch.empty(max(nws, 4), device=dev)
    gdy, gadd, gx = nhwc(dy).to(dev), nhwc(add, ld_add).to(dev), nhwc(x, ld_dx).to(dev)
    for out_scale, add_scale, masked in ((0.04, 0.2, True), (1.0, 1.0, False), (0.2, 1.0, True)):
        want = out_scale * mm
        want[:, :add_ch] += add_scale * add.double()
        lo, hi = (cin - 32, cin) if masked else (0, 0)
        if masked:
            want[:, lo:hi] *= torch.where(x[:, lo:hi] > 0, 1.0, 0.2).double()
        e = _lib.DgradEpilogue()
        e.out_scale, e.addend, e.addend_ld, e.addend_channels, e.addend_scale = out_scale, gadd.data_ptr(), ld_add, add_ch, add_scale
        if masked:
            e.act_out, e.act_slope, e.c_lo, e.c_hi = gx.data_ptr(), 0.2, lo, hi
        dx = torch.full((n, h, w, ld_dx), 7.0, device=dev)
        _lib.call('srx_conv2d_bwd_data_ex', C.byref(d), gdy.data_ptr(), st.wpk_bwd.data_ptr(), dx.data_ptr(), C.byref(e),
                  ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
        assert rel_err(nchw(dx.cpu(), cin), want.float()) < (2e-5 if bf16 else 2e-4), (out_scale, add_scale, masked)
        if ld_dx > cin:
            assert bool((dx[..., cin:] == 7.0).all())   # channels past Cin belong to someone else
    e = _lib.DgradEpilogue()
    e.out_scale = 0.5
    with pytest.raises(RuntimeError, match='without an addend'):
        _lib.call('srx_conv2d_bwd_data_ex', C.byref(d), gdy.data_ptr(), st.wpk_bwd.data_ptr(), dx.data_ptr(), C.byref(e),
                  ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)


def test_scaled_grouped_weight_gradient_and_channel_axpby(dev):
    """srx_conv2d_bwd_weight_multi_scaled (one multiplier per output, on weight and bias gradient) and
    srx_axpby_channels (axpby between channel slices of tensors with different row strides)."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = 2, 12, 12, 64, 32
    d = _lib.Conv2dDesc(n, h, w, cin, 192, cout, 32, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    xs, dys = [rnd((n, cin, h, w), 10 + i) for i in range(4)], [rnd((n, cout, h, w), 20 + i) for i in range(4)]
    scales = [1.0, 0.2, 0.04, 3.0]
    gw = [torch.zeros(cout, cin, 3, 3, device=dev) for _ in range(4)]
    gb = [torch.zeros(cout, device=dev) for _ in range(4)]
    nws = L.srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), 4)
    ws = torch.empty(max(nws, 4), device=dev)
    gx, gdy = [nhwc(x, 192).to(dev) for x in xs], [nhwc(t).to(dev) for t in dys]
    _lib.call('srx_conv2d_bwd_weight_multi_scaled', C.byref(d), 4, 1, arr(gx), arr(gdy), arr(gw), 0, arr(gb),
              (C.c_float * 4)(*scales), ws.data_ptr(), nws, s)
    for i in range(4):
        want = scales[i] * torch.nn.grad.conv2d_weight(xs[i].double(), (cout, cin, 3, 3), dys[i].double(), padding=1)
        assert rel_err(gw[i], want.float()) < 2e-4, i
        assert rel_err(gb[i], scales[i] * dys[i].sum((0, 2, 3))) < 2e-4, i
    m = 2 * 5 * 7
    a, b = torch.randn(m, 192, device=dev), torch.randn(m, 96, device=dev)
    y = torch.full((m, 64), 5.0, device=dev)
    _lib.call('srx_axpby_channels', a.data_ptr(), 192, 64, b.data_ptr(), 96, 32, y.data_ptr(), 64, 8, 48, m, 0.2, 1.0, s)
    assert torch.allclose(y[:, 8:56], 0.2 * a[:, 64:112] + b[:, 32:80], rtol=1e-6, atol=1e-6)
    assert bool((y[:, :8] == 5.0).all()) and bool((y[:, 56:] == 5.0).all())
    _lib.call('srx_axpby_channels', a.data_ptr(), 192, 0, b.data_ptr(), 96, 0, a.data_ptr(), 192, 0, 64, m, 0.2, 1.0, s)  # in place
    with pytest.raises(RuntimeError, match='slice out of range'):
        _lib.call('srx_axpby_channels', a.data_ptr(), 192, 160, b.data_ptr(), 96, 0, y.data_ptr(), 64, 0, 64, m, 1.0, 1.0, s)


@pytest.mark.parametrize('bf16,geom', [(False, (2, 16, 16)), (True, (2, 16, 16)),
                                       (True, (16, 32, 32))])  # config 4's paired problems on the image-row kernel
def test_paired_weight_gradient(dev, bf16, geom):
    """srx_conv2d_bwd_weight_multi_pair: conv_a (cin_lo -> 32) and conv_b (Cin -> 32) of a dense block read the same
    192-channel buffer and their output gradients are adjacent 32-channel slices of one gradient buffer; issued as ONE
    64-column problem each pair, three pairs in one launch, accumulating into existing gradients, biases riding along."""
    from torchsr_amd import _lib
    (n, h, w), cin_lo, cin, g = geom, 128, 160, 32
    d = _lib.Conv2dDesc(n, h, w, cin, 192, 2 * g, 192, 3, 3, 1, 1, 0, 2, 0.2, 0, 1 if bf16 else 0)
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    arr = lambda ts: (C.c_void_p * len(ts))(*[t if isinstance(t, int) else t.data_ptr() for t in ts])  # noqa: E731
    r = (lambda t: t.bfloat16().double()) if bf16 else (lambda t: t.double())
    nprob = 3
    bufs = [rnd((n, 192, h, w), 10 + i) for i in range(nprob)]
    gbufs = [rnd((n, 192, h, w), 20 + i) for i in range(nprob)]
    base = lambda shape, seed: [rnd(shape, seed + i) for i in range(nprob)]  # noqa: E731
    wa, wb, ba, bb = base((g, cin_lo, 3, 3), 30), base((g, cin, 3, 3), 40), base((g,), 50), base((g,), 60)
    gx, gg = [nhwc(t).to(dev) for t in bufs], [nhwc(t).to(dev) for t in gbufs]
    gwa, gwb, gba, gbb = ([t.to(dev) for t in ts] for ts in (wa, wb, ba, bb))
    nws = L.srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), nprob)
    ws = torch.empty(max(nws, 4), device=dev)
    _lib.call('srx_conv2d_bwd_weight_multi_pair', C.byref(d), nprob, arr(gx), arr([t.data_ptr() + 4 * cin_lo for t in gg]),
              arr(gwa), arr(gwb), cin_lo, 1, arr(gba), arr(gbb), ws.data_ptr(), nws, s)
    tol = 2e-5 if bf16 else 2e-4
    for i in range(nprob):
        dya, dyb = gbufs[i][:, cin_lo:cin_lo + g], gbufs[i][:, cin_lo + g:cin_lo + 2 * g]
        want_a = wa[i].double() + torch.nn.grad.conv2d_weight(r(bufs[i][:, :cin_lo]), (g, cin_lo, 3, 3), r(dya), padding=1)
        want_b = wb[i].double() + torch.nn.grad.conv2d_weight(r(bufs[i][:, :cin]), (g, cin, 3, 3), r(dyb), padding=1)
        assert rel_err(gwa[i], want_a.float()) < tol and rel_err(gwb[i], want_b.float()) < tol, i
        assert rel_err(gba[i], ba[i] + dya.sum((0, 2, 3))) < 2e-4 and rel_err(gbb[i], bb[i] + dyb.sum((0, 2, 3))) < 2e-4, i
    with pytest.raises(RuntimeError, match='both convs or neither'):
        _lib.call('srx_conv2d_bwd_weight_multi_pair', C.byref(d), nprob, arr(gx), arr(gg), arr(gwa), arr(gwb), cin_lo, 1,
                  arr(gba), None, ws.data_ptr(), nws, s)


def test_maxpool_relu_backward(dev):
    from torchsr_amd import _lib
    x = torch.relu(rnd((2, 8, 8, 8), 3)).to(dev)    # NHWC, a ReLU output: many exact zeros
    dy = rnd((2, 4, 4, 8), 4).to(dev)
    dx = torch.empty_like(x)
    _lib.call('srx_maxpool2x2_relu_bwd', dy.data_ptr(), x.data_ptr(), dx.data_ptr(), 2, 8, 8, 8,
              torch.cuda.current_stream().cuda_stream)
    xc = x.cpu().permute(0, 3, 1, 2).clone().requires_grad_(True)
    TF.max_pool2d(torch.relu(xc), 2).backward(dy.cpu().permute(0, 3, 1, 2))
    # where a whole window is 0 ATen sends the gradient to its first element and ReLU's backward then drops it
    want = (xc.grad * (xc.detach() > 0)).permute(0, 2, 3, 1)
    assert torch.equal(dx.cpu(), want)


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 64, True, 2), (1, 9, 13, 64, 32, False, 0), (16, 32, 32, 64, 64, True, 0)])
@pytest.mark.parametrize('bf16', [False, True])
def test_conv_with_fused_nearest_upsample(dev, case, bf16):
    """up = 2: ``conv(F.interpolate(x, scale_factor=2, mode='nearest'))`` (esrgan/generator.py:73-78) with the upsampled
    tensor never written -- forward, input gradient (sum over each 2x2 block of the upsampled gradient), weight and bias
    gradients against torch on the CPU, incl. ESRGAN's own 64->64 layer at 16x32x32 -> 64x64 (that one without its
    LeakyReLU: among 4 M activations a few land within rounding of 0 and take the other slope on the two sides)."""
    from torchsr_amd.layers import Conv2d, set_conv_precision
    n, h, w, cin, cout, bias, act = case
    torch.manual_seed(h * w)
    conv = Conv2d(cin, cout, 3, 1, 1, bias=bias, act=act, slope=0.2, up=2)
    ref = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=bias)
    ref.load_state_dict(conv.state_dict())
    conv = conv.to(dev)
    if bf16:
        set_conv_precision(conv, 'bf16')
    x = rnd((n, cin, h, w), 1)
    xc = x.clone().requires_grad_(True)
    xin = xc.to(torch.bfloat16).float() if bf16 else xc
    wref = ref.weight.to(torch.bfloat16).float() if bf16 else ref.weight
    yc = TF.conv2d(TF.interpolate(xin, scale_factor=2, mode='nearest'), wref, ref.bias, 1, 1)
    if act:
        yc = TF.leaky_relu(yc, 0.2)
    go = rnd(yc.shape, 2)
    yc.backward(go)
    xg = nhwc(x).to(dev).requires_grad_(True)
    yg = conv(xg)
    assert yg.shape == (n, 2 * h, 2 * w, cout)
    assert rel_err(nchw(yg.cpu(), cout), yc) < 2e-4
    yg.backward(nhwc(go).to(dev))
    tol = 1e-2 if bf16 else 2e-4   # (bf16: the product also rounds dy in the gradients, torch's autograd does not)
    assert rel_err(nchw(xg.grad.cpu(), cin), xc.grad) < tol
    if not bf16:
        assert rel_err(conv.weight.grad, ref.weight.grad) < tol
    if bias:
        assert rel_err(conv.bias.grad, ref.bias.grad) < 2e-4


def test_row_tile_plan(dev):
    """The SRGAN residual conv at the reference batch runs as 256 workgroups of 36 pixels (forward and
    data gradient); a layer the row tile does not cover falls back to the generic plan."""
    from torchsr_amd import _lib
    d = _lib.Conv2dDesc(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0)
    out = (C.c_int * 6)()
    for which in (0, 1):
        _lib.call('srx_conv2d_plan', C.byref(d), which, out)
        assert list(out)[:4] == [36, 64, 1, 256]
    assert _lib.lib().srx_conv2d_stat_rows(C.byref(d)) == 256
    assert _lib.lib().srx_conv2d_fwd_ws_floats(C.byref(d)) == 0
    d2 = _lib.Conv2dDesc(16, 24, 26, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0)
    _lib.call('srx_conv2d_plan', C.byref(d2), 0, out)
    assert out[0] in (64, 128, 144)
    # the reference's CPU batch size (BASELINE configs[0]): 12-pixel tiles, 96 workgroups instead of 32 of 36 pixels
    d3 = _lib.Conv2dDesc(2, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0)
    _lib.call('srx_conv2d_plan', C.byref(d3), 0, out)
    assert list(out)[:4] == [12, 64, 1, 96] and _lib.lib().srx_conv2d_stat_rows(C.byref(d3)) == 96


@pytest.mark.parametrize('cfg', [(2, 24, 24, 64, 'prelu', True), (2, 12, 12, 128, 'lrelu', False),
                                 (3, 5, 7, 8, 'none', True), (2, 6, 6, 512, 'lrelu', False)],
                         ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('training', [True, False])
def test_bn_act(dev, cfg, training):
    from torchsr_amd import functional as F
    from torchsr_amd.layers import BatchNorm2d
    from torchsr_amd._lib import ACT_LRELU, ACT_NONE, ACT_PRELU
    n, h, w, c, act, with_res = cfg
    y = rnd((n, c, h, w), 5) * 2 + 0.3
    res = rnd((n, c, h, w), 6) if with_res else None
    bn_ref = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + 0.2 * rnd((c,), 7))
        bn_ref.bias.copy_(0.1 * rnd((c,), 8))
        bn_ref.running_mean.copy_(0.1 * rnd((c,), 9))
        bn_ref.running_var.copy_(1 + 0.3 * rnd((c,), 10))
    bn = BatchNorm2d(c)
    bn.load_state_dict(bn_ref.state_dict())
    slope = torch.tensor([0.25], requires_grad=True)
    bn_ref.train(training)
    bn.train(training)
    yc = y.clone().requires_grad_(True)
    rc = res.clone().requires_grad_(True) if with_res else None
    o = bn_ref(yc)
    if act == 'prelu':
        o = TF.prelu(o, slope)
    elif act == 'lrelu':
        o = TF.leaky_relu(o, 0.2)
    if with_res:
        o = o + rc
    go = rnd(o.shape, 11)
    o.backward(go)

    bn = bn.to(dev)
    yg = nhwc(y).to(dev).requires_grad_(True)
    rg = nhwc(res).to(dev).requires_grad_(True) if with_res else None
    sg = torch.tensor([0.25], device=dev, requires_grad=True)
    code = {'prelu': ACT_PRELU, 'lrelu': ACT_LRELU, 'none': ACT_NONE}[act]
    og = bn(yg, None, act=code, slope=0.2, prelu=sg if act == 'prelu' else None, residual=rg)
    assert rel_err(nchw(og.cpu(), c), o) < 2e-4
    og.backward(nhwc(go).to(dev))
    assert rel_err(nchw(yg.grad.cpu(), c), yc.grad) < 5e-4
    assert rel_err(bn.weight.grad, bn_ref.weight.grad) < 5e-4
    assert rel_err(bn.bias.grad, bn_ref.bias.grad) < 5e-4
    if act == 'prelu':
        assert rel_err(sg.grad, slope.grad) < 5e-4
    if with_res:
        assert rel_err(nchw(rg.grad.cpu(), c), rc.grad) < 1e-6
    assert rel_err(bn.running_mean, bn_ref.running_mean) < 1e-5
    assert rel_err(bn.running_var, bn_ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(bn_ref.num_batches_tracked)


@pytest.mark.parametrize('groups', [2, 3])
def test_bn_act_groups_equal_consecutive_calls(dev, groups):
    """groups = G: one call over G*N images == G consecutive training-mode calls of torch's BatchNorm2d (own batch
    statistics each, running statistics updated G times in order, parameter gradients summed)."""
    from torchsr_amd import functional as F
    from torchsr_amd.layers import BatchNorm2d
    from torchsr_amd._lib import ACT_LRELU
    n, h, w, c = 2, 16, 16, 64
    ys = [rnd((n, c, h, w), 20 + g) * (1 + g) + 0.2 * g for g in range(groups)]
    gos = [rnd((n, c, h, w), 40 + g) for g in range(groups)]
    bn_ref = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn_ref.weight.copy_(1 + 0.2 * rnd((c,), 7))
        bn_ref.bias.copy_(0.1 * rnd((c,), 8))
    bn = BatchNorm2d(c)
    bn.load_state_dict(bn_ref.state_dict())
    bn_ref.train()
    ycs = [y.clone().requires_grad_(True) for y in ys]
    for yc, go in zip(ycs, gos):
        TF.leaky_relu(bn_ref(yc), 0.2).backward(go)
    bn = bn.to(dev).train()
    yg = torch.cat([nhwc(y) for y in ys], 0).to(dev).requires_grad_(True)
    assert F.bn_groups_ok(groups * n * h * w, None, groups)
    og = bn(yg, None, act=ACT_LRELU, slope=0.2, groups=groups)
    og.backward(torch.cat([nhwc(g) for g in gos], 0).to(dev))
    for g in range(groups):
        assert rel_err(nchw(yg.grad[g * n:(g + 1) * n].cpu(), c), ycs[g].grad) < 5e-4, g
    assert rel_err(bn.weight.grad, bn_ref.weight.grad) < 5e-4 and rel_err(bn.bias.grad, bn_ref.bias.grad) < 5e-4
    assert rel_err(bn.running_mean, bn_ref.running_mean) < 1e-5 and rel_err(bn.running_var, bn_ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == groups
    assert not F.bn_groups_ok(3 * 100, None, 2) and not F.bn_groups_ok(2 * 144, 1, 2)
    with pytest.raises(RuntimeError, match='straddle|split'):
        bn(torch.zeros(1, 6, 8, c, device=dev), None, groups=4)   # 48 rows: 12 per group, row blocks are 32


def test_activations_and_pool(dev):
    from torchsr_amd import functional as F
    x = rnd((2, 16, 10, 12), 20)
    xg = nhwc(x).to(dev).requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    a = torch.tensor([0.3], requires_grad=True)
    ag = torch.tensor([0.3], device=dev, requires_grad=True)
    g = rnd(x.shape, 21)
    TF.prelu(xc, a).backward(g)
    y = F.prelu(xg, ag)
    y.backward(nhwc(g).to(dev))
    assert rel_err(nchw(y.cpu(), 16), TF.prelu(x, a)) < 1e-6
    assert rel_err(nchw(xg.grad.cpu(), 16), xc.grad) < 1e-6
    assert rel_err(ag.grad, a.grad) < 1e-4
    # leaky relu, sigmoid on an odd-sized tensor
    v = rnd((7, 1), 22) * 4
    vg = v.to(dev).requires_grad_(True)
    vc = v.clone().requires_grad_(True)
    torch.sigmoid(TF.leaky_relu(vc, 0.2)).sum().backward()
    F.sigmoid(F.leaky_relu(vg, 0.2)).backward(torch.ones(7, 1, device=dev))
    assert rel_err(vg.grad, vc.grad) < 1e-5
    # axpby
    z = rnd((5, 3), 23)
    assert rel_err(F.axpby(v[:5].expand(5, 3).contiguous().to(dev), z.to(dev), 0.2, 1.0),
                   0.2 * v[:5].expand(5, 3) + z) < 1e-6
    # maxpool
    xc2 = x.clone().requires_grad_(True)
    pc = TF.max_pool2d(xc2, 2, 2)
    gp = rnd(pc.shape, 24)
    pc.backward(gp)
    xg2 = nhwc(x).to(dev).requires_grad_(True)
    pg = F.maxpool2x2(xg2)
    pg.backward(nhwc(gp).to(dev))
    assert torch.equal(nchw(pg.cpu(), 16), pc.detach())
    assert torch.equal(nchw(xg2.grad.cpu(), 16), xc2.grad)


@pytest.mark.parametrize('cfg', [(16, 18432, 1024, 2), (16, 1024, 1, 0), (3, 20, 5, 1), (16, 2048, 100, 2),
                                 (33, 256, 40, 0), (32, 18432, 1024, 2), (64, 2048, 100, 2), (70, 512, 48, 0)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_linear(dev, cfg):
    from torchsr_amd import functional as F
    b, k, j, act = cfg
    x = rnd((b, k), 30)
    w = rnd((j, k), 31) * (1.0 / k) ** 0.5
    bias = rnd((j,), 32) * 0.1
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yc = TF.linear(xc, wc, bc)
    if act == 1:
        yc = TF.relu(yc)
    elif act == 2:
        yc = TF.leaky_relu(yc, 0.2)
    g = rnd(yc.shape, 33)
    yc.backward(g)
    xg, wg, bg = (t.clone().to(dev).requires_grad_(True) for t in (x, w, bias))
    yg = F.linear(xg, wg, bg, act, 0.2)
    yg.backward(g.to(dev))
    assert rel_err(yg, yc) < 2e-4
    assert rel_err(xg.grad, xc.grad) < 2e-4
    assert rel_err(wg.grad, wc.grad) < 2e-4
    assert rel_err(bg.grad, bc.grad) < 2e-4


def test_losses(dev):
    from torchsr_amd import functional as F
    a, b = rnd((2, 3, 17, 19), 40), rnd((2, 3, 17, 19), 41)
    for ours, ref in [(F.mse_loss, TF.mse_loss), (F.l1_loss, TF.l1_loss)]:
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        lc = ref(ac, bc)
        (lc * 1.7).backward()
        ag, bgp = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
        lg = ours(ag, bgp)
        F.axpby(lg, lg, 1.7, 0.0).backward()
        assert rel_err(lg, lc) < 1e-5
        assert rel_err(ag.grad, ac.grad) < 1e-5
        assert rel_err(bgp.grad, bc.grad) < 1e-5
    p = torch.sigmoid(rnd((16, 1), 42) * 6)
    p[0] = 1.0   # saturated probabilities hit the -100 clamp
    p[1] = 0.0
    for t in (0.0, 1.0):
        pc = p.clone().requires_grad_(True)
        lc = TF.binary_cross_entropy(pc, torch.full_like(p, t))
        lc.backward()
        pg = p.to(dev).requires_grad_(True)
        lg = F.bce_loss(pg, t)
        lg.backward()
        assert rel_err(lg, lc) < 1e-5
        assert rel_err(pg.grad, pc.grad) < 1e-5
    x = rnd((16, 1), 43) * 8
    sh = torch.tensor(0.37)
    for t in (0.0, 1.0):
        xc, sc = x.clone().requires_grad_(True), sh.clone().requires_grad_(True)
        lc = TF.binary_cross_entropy_with_logits(xc - sc, torch.full_like(x, t))
        lc.backward()
        xg, sg = x.to(dev).requires_grad_(True), sh.to(dev).requires_grad_(True)
        lg = F.bce_with_logits(xg, t, sg)
        lg.backward()
        assert rel_err(lg, lc) < 1e-5
        assert rel_err(xg.grad, xc.grad) < 1e-5
        assert rel_err(sg.grad, sc.grad) < 1e-5


def test_adam_matches_torch(dev):
    from torchsr_amd._lib import call
    n = 1003
    p0, g_all = rnd((n,), 50), [rnd((n,), 51 + i) * (10.0 ** -(i % 3)) for i in range(5)]
    pc = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pc], lr=1e-4, betas=(0.9, 0.999))
    pad = 1004
    p = torch.zeros(pad, device=dev)
    p[:n] = p0.to(dev)
    g = torch.zeros(pad, device=dev)
    m, v = torch.zeros(pad, device=dev), torch.zeros(pad, device=dev)
    step = torch.zeros((), dtype=torch.int64, device=dev)
    lr = torch.tensor(1e-4, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    for gi in g_all:
        pc.grad = gi.clone()
        opt.step()
        g[:n] = gi.to(dev)
        call('srx_adam_step', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr.data_ptr(), 0.9, 0.999,
             1e-8, 1.0, step.data_ptr(), s)
    assert int(step) == 5
    assert (p[:n].cpu() - pc.detach()).abs().max().item() < 2e-7
    assert float(p[n:].abs().max()) == 0.0


def test_device_augmentation_kernels(dev):
    """srx_crop_flip_u8 against numpy indexing (exact); srx_bicubic_down against torch's antialiased bicubic
    (the Keys a=-0.5 filter PIL's BICUBIC uses) on the CPU, unquantised to 2e-6 and quantised to one 8-bit step."""
    from torchsr_amd import _lib
    rng = np.random.RandomState(3)
    shapes = [(100, 130), (96, 96), (150, 97)]
    imgs = [torch.from_numpy((rng.rand(h, w, 3) * 255).astype('uint8')) for h, w in shapes]
    dimgs = [im.to(dev) for im in imgs]
    crop = 96
    meta = [[100, 130, 3, 20, 1, 0], [96, 96, 0, 0, 0, 1], [150, 97, 54, 1, 1, 1]]
    ptrs = torch.tensor([im.data_ptr() for im in dimgs], dtype=torch.int64, device=dev)
    meta_t = torch.tensor(meta, dtype=torch.int32, device=dev)
    hr = torch.empty((3, 3, crop, crop), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.call('srx_crop_flip_u8', ptrs.data_ptr(), meta_t.data_ptr(), hr.data_ptr(), 3, crop, s)
    for n, (im, (h, w, top, left, hf, vf)) in enumerate(zip(imgs, meta)):
        ref = im[top:top + crop, left:left + crop].numpy()
        if hf:
            ref = ref[:, ::-1]
        if vf:
            ref = ref[::-1]
        ref = torch.from_numpy(ref.copy()).permute(2, 0, 1).float() / 255.0
        assert torch.equal(hr[n].cpu(), ref)
    for scale, size in ((4, 96), (4, 128), (2, 50), (3, 99)):
        x = rnd((2, 3, size, size + 4 * scale), 7, 0.0, 1.0)
        want = TF.interpolate(x, scale_factor=1.0 / scale, mode='bicubic', antialias=True, align_corners=False)
        out = torch.empty(want.shape, device=dev)
        _lib.call('srx_bicubic_down', x.to(dev).data_ptr(), out.data_ptr(), 2, 3, size, size + 4 * scale, scale, 0, s)
        assert float((out.cpu() - want).abs().max()) < 2e-6
        _lib.call('srx_bicubic_down', x.to(dev).data_ptr(), out.data_ptr(), 2, 3, size, size + 4 * scale, scale, 1, s)
        q = (want.clamp(0, 1) * 255).round() / 255
        assert float((out.cpu() - q).abs().max()) <= 1.0 / 255 + 1e-6       # ties may round either way
        assert float(((out.cpu() - q).abs() > 1e-6).float().mean()) < 1e-3


def test_bicubic_down_vs_pil_fixture(dev):
    """``srx_bicubic_down`` (quantised) against what the reference's pipeline feeds the generator: PIL's
    ``Resize(crop/4, BICUBIC)`` of the 8-bit crop (torchsr/dataset.py:88-92,121-125; fixture written by
    oracle/gen_golden.py with the PIL of the build container).  PIL rounds its horizontal pass to 8 bits before the
    vertical one, the kernel keeps the intermediate in fp32, so the two may differ by ONE 8-bit step and must do
    so rarely."""
    from conftest import GOLDEN
    from torchsr_amd import _lib
    gold = np.load(os.path.join(GOLDEN, 'pil_bicubic.npz'))
    crops, lows = gold['crops'], gold['lows'].astype(np.int32)          # [4,96,96,3] u8, [4,24,24,3] u8
    hr = (torch.from_numpy(crops).permute(0, 3, 1, 2).float() / 255.0).contiguous().to(dev)  # ToTensor
    out = torch.empty((4, 3, 24, 24), device=dev)
    _lib.call('srx_bicubic_down', hr.data_ptr(), out.data_ptr(), 4, 3, 96, 96, 4, 1, torch.cuda.current_stream().cuda_stream)
    got = (out.cpu() * 255.0).round().to(torch.int32).permute(0, 2, 3, 1).numpy()
    diff = np.abs(got - lows)
    assert diff.max() <= 1, diff.max()
    # torch's CPU antialiased bicubic (fp32 intermediate, like the kernel) differs from PIL on 10.9 % of these pixels
    assert (diff != 0).mean() < 0.15, (diff != 0).mean()


FULL_SIZE_LAYERS = [
    # every distinct conv shape of the batch-16 SRGAN GAN step (BASELINE configs[1]): N, H, W, Cin, Cout, k, s, p, shuffle
    (16, 24, 24, 64, 64, 3, 1, 1, 0), (16, 24, 24, 3, 64, 9, 1, 4, 0), (16, 24, 24, 64, 256, 3, 1, 1, 2),
    (16, 48, 48, 64, 256, 3, 1, 1, 2), (16, 96, 96, 64, 3, 9, 1, 4, 0), (16, 96, 96, 3, 64, 3, 1, 1, 0),
    (16, 96, 96, 64, 64, 3, 2, 1, 0), (16, 48, 48, 64, 128, 3, 1, 1, 0), (16, 48, 48, 128, 128, 3, 2, 1, 0),
    (16, 24, 24, 128, 256, 3, 1, 1, 0), (16, 24, 24, 256, 256, 3, 2, 1, 0), (16, 12, 12, 256, 512, 3, 1, 1, 0),
    (16, 12, 12, 512, 512, 3, 2, 1, 0), (32, 96, 96, 64, 64, 3, 1, 1, 0), (32, 48, 48, 128, 128, 3, 1, 1, 0),
    (32, 24, 24, 256, 256, 3, 1, 1, 0), (32, 12, 12, 512, 512, 3, 1, 1, 0), (32, 6, 6, 512, 512, 3, 1, 1, 0),
]


@pytest.mark.parametrize('case', FULL_SIZE_LAYERS, ids=lambda c: 'x'.join(map(str, c)))
def test_full_size_adjoint_identities(dev, case):
    """At the benchmark's own sizes a CPU oracle is too slow, so the three conv kernels are tied together by
    properties that hold at any size:  <dy, conv(x; W)> = <x, dgrad(dy; W)> = <W, wgrad(x, dy)>  (the
    forward is linear in x and in W, and the two gradients are its adjoints), and conv is additive in x."""
    from torchsr_amd.layers import Conv2d
    n, h, w, cin, cout, k, s, p, shuffle = case
    torch.manual_seed(hash(case) % 997)
    conv = Conv2d(cin, cout, k, s, p, bias=False, shuffle=shuffle).to(dev)
    cs = (cin + 3) // 4 * 4
    x = torch.zeros(n, h, w, cs, device=dev)
    x[..., :cin] = torch.rand(n, h, w, cin, device=dev) - 0.5
    x.requires_grad_(True)
    y = conv(x)
    cl = cout // 4 if shuffle else cout
    dy = torch.zeros_like(y)
    dy[..., :cl] = torch.rand(y.shape[:-1] + (cl,), device=dev) - 0.5
    dx, dw = torch.autograd.grad(y, (x, conv.weight), dy)
    lhs = (dy.double() * y.detach().double()).sum().item()
    via_x = (x.detach().double() * dx.double()).sum().item()
    via_w = (conv.weight.detach().double() * dw.double()).sum().item()
    scale = (dy.double().norm() * y.detach().double().norm()).item()
    assert abs(lhs - via_x) <= 2e-5 * scale, (lhs, via_x)
    assert abs(lhs - via_w) <= 2e-5 * scale, (lhs, via_w)
    with torch.no_grad():
        x2 = torch.zeros_like(x)
        x2[..., :cin] = torch.rand(n, h, w, cin, device=dev) - 0.5
        y12 = conv(0.5 * x.detach() - 2.0 * x2)
        y2 = conv(x2)
    assert rel_err(y12, 0.5 * y.detach() - 2.0 * y2) < 2e-5


def _random_conv_cases(count, seed):
    rng = np.random.RandomState(seed)
    cases = []
    while len(cases) < count:
        k = int(rng.choice([1, 3, 3, 3, 5]))
        s = int(rng.choice([1, 1, 2, 3]))
        p = int(rng.randint(0, k // 2 + 2))
        cin = int(rng.choice([4, 8, 12, 24, 32, 40, 64, 96, 128, 160, 200]))
        cout = int(rng.choice([4, 8, 12, 20, 32, 48, 64, 100, 128, 256]))
        n, h, w = int(rng.randint(1, 4)), int(rng.randint(k + 1, 30)), int(rng.randint(k + 1, 30))
        if (h + 2 * p - k) // s + 1 < 1 or (w + 2 * p - k) // s + 1 < 1:
            continue
        cases.append((n, h, w, cin, cout, k, s, p, bool(rng.randint(0, 2)), int(rng.randint(0, 3)), 0))
    return cases


@pytest.mark.parametrize('case', _random_conv_cases(36, 20260401), ids=lambda c: 'x'.join(map(str, c)))
def test_conv2d_random_shapes(dev, case):
    """Seeded sweep over ragged shapes, strides 1-3, paddings from 0 to beyond 'same', odd channel counts:
    whatever tile / split plan the library picks, forward, data gradient, weight and bias gradients must agree
    with the CPU convolution."""
    test_conv2d_fwd_bwd(dev, case)


@pytest.mark.parametrize('n,h,w', [(2, 16, 16), (1, 13, 21), (3, 8, 8), (1, 5, 40),
                                   (16, 32, 32)])  # BASELINE config 4's own launch: 256 workgroups, one per CU
def test_fused_dense_block_forward(dev, n, h, w):
    """``srx_rdb_fwd`` (one launch: five convs, intermediates in LDS, bf16 products) against an fp64 evaluation of the same
    arithmetic -- every conv multiplies bf16-rounded inputs and weights, sums exactly, adds the fp32 bias; c1..c4 are kept
    in fp32 and rounded again where the next conv reads them (torchsr/esrgan/residual.py:81-86) -- on tiles that are whole,
    ragged (H, W not multiples of 8) and narrower than the halo, through the C ABI."""
    import ctypes as C
    import torch.nn.functional as TF
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(100 * h + w)
    r16 = lambda t: t.to(torch.bfloat16).double()  # noqa: E731
    ws = [torch.randn(32 if k < 4 else 64, 64 + 32 * k, 3, 3, generator=g) * (2.0 / (9 * (64 + 32 * k))) ** 0.5 for k in range(5)]
    bs = [torch.randn(32 if k < 4 else 64, generator=g) * 0.1 for k in range(5)]
    x = torch.randn(n, 64, h, w, generator=g)
    # device
    wd = [t.to(dev).contiguous() for t in ws]
    bd = [t.to(dev).contiguous() for t in bs]
    table = torch.tensor([t.data_ptr() for t in wd], dtype=torch.int64).to(dev)
    per = L.srx_rdb_packed_bytes()
    assert per == 479232
    pk = torch.empty(per, dtype=torch.uint8, device=dev)
    _lib.call('srx_rdb_pack', table.data_ptr(), 1, pk.data_ptr(), s)
    buf = torch.full((n, h, w, 192), float('nan'), device=dev)
    buf[..., :64] = x.permute(0, 2, 3, 1).to(dev)
    out = torch.full((n, h, w, 192), float('nan'), device=dev)
    biases = (C.c_void_p * 5)(*[t.data_ptr() for t in bd])
    extra = torch.randn(n, h, w, 64, generator=g) if h % 2 else None  # (ragged cases: with the RRDB-end addend, esrgan/residual.py:128)
    exd = None if extra is None else extra.to(dev)
    _lib.call('srx_rdb_fwd', n, h, w, buf.data_ptr(), 192, pk.data_ptr(), biases, 0.2, 0.2, 0.7, None if exd is None else exd.data_ptr(),
              64, out.data_ptr(), 192, s)
    torch.cuda.synchronize()
    got_c = buf[..., 64:].permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got_c).all()
    # reference, conv by conv ON THE DEVICE'S OWN intermediates (an intermediate one fp32 ulp away from its fp64 value may
    # round to the other bf16 neighbour in the next conv: 2^-9 of a product, not an error of this kernel)
    feats = [x]
    for k in range(4):
        z = TF.conv2d(r16(torch.cat(feats, 1)), r16(ws[k]), bs[k].double(), 1, 1)
        want = torch.where(z > 0, z, z * 0.2)
        got = got_c[:, 32 * k:32 * k + 32]
        assert ((got.double() - want).abs().max() / want.abs().max()).item() < 2e-5, (k, ((got - want).abs().max() / want.abs().max()).item())
        feats.append(got)
    y = TF.conv2d(r16(torch.cat(feats, 1)), r16(ws[4]), bs[4].double(), 1, 1) * 0.2 + x.double()
    if extra is not None:
        y = y * 0.7 + extra.permute(0, 3, 1, 2).double()
    got_y = out[..., :64].permute(0, 3, 1, 2).cpu().double()
    assert ((got_y - y).abs().max() / y.abs().max()).item() < 2e-5
    assert torch.isnan(out[..., 64:]).all() and torch.equal(buf[..., :64].cpu(), x.permute(0, 2, 3, 1))  # nothing else touched


@pytest.mark.parametrize('n,h,w', [(2, 16, 16), (1, 13, 21), (1, 5, 40), (16, 32, 32)])  # (16, 32, 32): config 4's launch
def test_fused_dense_block_backward(dev, n, h, w):
    """``srx_rdb_bwd`` (the data-gradient chain of a dense block in one launch, bf16 products) against fp64: the slice
    gradients g4..g1 and the input gradient, stage by stage on the device's own earlier stages (see
    test_fused_dense_block_forward for why), whole / ragged / narrow tiles, through the C ABI."""
    import torch.nn.functional as TF
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(77 * h + w)
    r16 = lambda t: t.to(torch.bfloat16).double()  # noqa: E731
    ws = [torch.randn(32 if k < 4 else 64, 64 + 32 * k, 3, 3, generator=g) * (2.0 / (9 * (64 + 32 * k))) ** 0.5 for k in range(5)]
    acts = torch.randn(n, 192, h, w, generator=g)          # x, c1..c4 as the forward left them (only the signs of c matter)
    dy = torch.randn(n, 64, h, w, generator=g)
    skip = torch.randn(n, 64, h, w, generator=g)
    scale, slope, skip_scale = 0.2 * 0.7, 0.2, 0.9
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)  # noqa: E731
    wd = [t.to(dev).contiguous() for t in ws]
    table = torch.tensor([t.data_ptr() for t in wd], dtype=torch.int64).to(dev)
    pk = torch.empty(L.srx_rdb_packed_bytes(), dtype=torch.uint8, device=dev)
    _lib.call('srx_rdb_pack_bwd', table.data_ptr(), 1, pk.data_ptr(), s)
    buf, dyd, skd = nhwc(acts), nhwc(dy), nhwc(skip)
    extra = torch.randn(n, 64, h, w, generator=g) if h % 2 else None  # (the RRDB's own skip gradient on its first block)
    exd = None if extra is None else nhwc(extra)
    gbuf = torch.full((n, h, w, 192), float('nan'), device=dev)
    dx = torch.full((n, h, w, 64), float('nan'), device=dev)
    _lib.call('srx_rdb_bwd', n, h, w, dyd.data_ptr(), 64, scale, buf.data_ptr(), 192, pk.data_ptr(), slope, gbuf.data_ptr(), 192,
              skd.data_ptr(), 64, skip_scale, None if exd is None else exd.data_ptr(), 64, dx.data_ptr(), 64, s)
    torch.cuda.synchronize()
    assert torch.isnan(gbuf[..., :64]).all() and torch.isfinite(gbuf[..., 64:]).all() and torch.isfinite(dx).all()
    got_g = gbuf[..., 64:].permute(0, 3, 1, 2).cpu()       # g1..g4
    gs = {5: (dy * scale)}                                  # conv k's output gradient, fp32 as stored
    rel = lambda a, b: ((a.double() - b).abs().max() / b.abs().max()).item()  # noqa: E731

    def share(k, lo, hi):  # conv_k^T(bf16(g_k))[input channels lo:hi] with bf16 weights
        full = torch.nn.grad.conv2d_input((n, 64 + 32 * (k - 1), h, w), r16(ws[k - 1]), r16(gs[k]), padding=1)
        return full[:, lo:hi]

    for j in (4, 3, 2, 1):
        lo = 64 + 32 * (j - 1)
        a = sum(share(k, lo, lo + 32) for k in range(j + 1, 6))
        c = acts[:, lo:lo + 32].double()
        want = torch.where(c > 0, a, a * slope)
        got = got_g[:, 32 * (j - 1):32 * j]
        assert rel(got, want) < 2e-5, (j, rel(got, want))
        gs[j] = got
    want_dx = sum(share(k, 0, 64) for k in range(1, 6)) + skip_scale * skip.double()
    if extra is not None:
        want_dx = want_dx + extra.double()
    assert rel(dx.permute(0, 3, 1, 2).cpu(), want_dx) < 2e-5


@pytest.mark.parametrize('prelu', [False, True])
def test_data_gradient_with_bn_backward_reduce_epilogue(dev, prelu):
    """``srx_conv2d_bwd_data_bn`` + ``srx_bn_act_bwd_finish`` (the BatchNorm backward's reduce pass in the epilogue of the
    data gradient that produces its output gradient) against the separate launches: ``srx_conv2d_bwd_data[_add]`` then
    ``srx_bn_act_bwd`` -- same dx (bit for bit: same kernel body), same sums / dy / parameter gradients to rounding."""
    import ctypes as C
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    n, h, w, c = 4, 12, 12, 64
    m = n * h * w
    d = _lib.Conv2dDesc(n, h, w, c, c, c, c, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    rows = L.srx_conv2d_bwd_data_bn_rows(C.byref(d))
    plan = (C.c_int * 6)()
    _lib.call('srx_conv2d_plan', C.byref(d), 1, plan)
    assert plan[0] in (12, 36) and rows == m // plan[0]  # (576 pixels: the 12-pixel tiles of small batches)
    g = torch.Generator().manual_seed(31)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)  # noqa: E731
    wt = rnd(c, c, 3, 3) * 0.05
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(C.byref(d)), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(C.byref(d)), device=dev)
    _lib.call('srx_conv2d_pack', C.byref(d), wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    dy, addend, y = rnd(n, h, w, c), rnd(n, h, w, c), rnd(n, h, w, c)
    mean, invstd = rnd(c) * 0.1, torch.rand(c, generator=g).to(dev) + 0.5
    gamma, beta = rnd(c), rnd(c) * 0.3
    slope = torch.tensor([0.25], device=dev) if prelu else None
    act = _lib.ACT_PRELU if prelu else _lib.ACT_NONE
    W = 2 * c + 4
    for add in (None, addend):
        # separate launches
        dx0 = torch.empty_like(dy)
        nws = L.srx_conv2d_bwd_data_ws_floats(C.byref(d))
        ws = torch.empty(max(nws, 4), device=dev)
        if add is None:
            _lib.call('srx_conv2d_bwd_data', C.byref(d), dy.data_ptr(), wb.data_ptr(), dx0.data_ptr(), 0, ws.data_ptr(), nws, s)
        else:
            _lib.call('srx_conv2d_bwd_data_add', C.byref(d), dy.data_ptr(), wb.data_ptr(), add.data_ptr(), dx0.data_ptr(), ws.data_ptr(), nws, s)
        sums0, out0 = torch.empty(W, device=dev), torch.empty_like(dy)
        gg0, gb0, gp0 = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
        nbw = L.srx_bn_bwd_ws_floats(m, c)
        bws = torch.empty(nbw, device=dev)
        _lib.call('srx_bn_act_bwd', dx0.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  sums0.data_ptr(), out0.data_ptr(), m, c, 1, act, 0.0, None if slope is None else slope.data_ptr(), 1, gg0.data_ptr(),
                  gb0.data_ptr(), gp0.data_ptr() if prelu else None, bws.data_ptr(), nbw, s)
        # fused
        dx1 = torch.empty_like(dy)
        table = torch.full((rows, W), float('nan'), device=dev)
        _lib.call('srx_conv2d_bwd_data_bn', C.byref(d), dy.data_ptr(), wb.data_ptr(), None if add is None else add.data_ptr(),
                  dx1.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                  None if slope is None else slope.data_ptr(), table.data_ptr(), s)
        sums1, out1 = torch.empty(W, device=dev), torch.empty_like(dy)
        gg1, gb1, gp1 = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
        _lib.call('srx_bn_act_bwd_finish', dx1.data_ptr(), y.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
                  beta.data_ptr(), table.data_ptr(), rows, 2, sums1.data_ptr(), out1.data_ptr(), m, c, act, 0.0,
                  None if slope is None else slope.data_ptr(), gg1.data_ptr(), gb1.data_ptr(), gp1.data_ptr() if prelu else None, s)
        torch.cuda.synchronize()
        assert torch.equal(dx0, dx1)
        assert torch.isfinite(table[:, :2 * c + 2]).all()
        assert rel_err(sums1[:2 * c], sums0[:2 * c]) < 1e-5 and rel_err(out1, out0) < 1e-5
        assert rel_err(gg1, gg0) < 1e-5 and rel_err(gb1, gb0) < 1e-5
        if prelu:
            assert abs(gp1.item() - gp0.item()) <= 1e-5 * max(abs(gp0.item()), 1.0)
    # layers the row-tile kernel does not serve are refused, and say so beforehand
    d2 = _lib.Conv2dDesc(1, 100, 100, c, c, c, c, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    assert L.srx_conv2d_bwd_data_bn_rows(C.byref(d2)) == 0
    rc = L.srx_conv2d_bwd_data_bn(C.byref(d2), dy.data_ptr(), wb.data_ptr(), None, dy.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                  invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, table.data_ptr(), s)
    assert rc != 0 and 'row tile' in _lib.last_error()


@pytest.mark.parametrize('prelu', [True, False])
@pytest.mark.parametrize('below', [True, False])
@pytest.mark.parametrize('shape', [(4, 12, 12), (2, 48, 48)])  # (48 wide: two batches of patch loads per thread)
def test_data_gradient_with_batchnorm_backward_on_its_input(dev, prelu, below, shape):
    """``srx_conv2d_bwd_data_bn_in`` (the apply pass of the BatchNorm (+ PReLU) backward ABOVE a conv formed while the conv's
    data gradient stages its input, the conv's output gradient written on the side; optionally the reduce pass of the
    BatchNorm BELOW in the epilogue) against the separate launches: same dy, same dx, same table to rounding."""
    import ctypes as C
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    (n, h, w), c = shape, 64
    m = n * h * w
    d = _lib.Conv2dDesc(n, h, w, c, c, c, c, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    assert L.srx_conv2d_bwd_data_bn_in_ok(C.byref(d)) == 1
    rows = L.srx_conv2d_bwd_data_bn_rows(C.byref(d))
    g = torch.Generator().manual_seed(41)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)  # noqa: E731
    wt = rnd(c, c, 3, 3) * 0.05
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(C.byref(d)), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(C.byref(d)), device=dev)
    _lib.call('srx_conv2d_pack', C.byref(d), wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    dout, y_above, addend, y_below = rnd(n, h, w, c), rnd(n, h, w, c), rnd(n, h, w, c), rnd(n, h, w, c)
    mean, invstd, gamma, beta = rnd(c) * 0.1, torch.rand(c, generator=g).to(dev) + 0.5, rnd(c), rnd(c) * 0.3
    mean_b, invstd_b, gamma_b, beta_b = rnd(c) * 0.1, torch.rand(c, generator=g).to(dev) + 0.5, rnd(c), rnd(c) * 0.3
    slope = torch.tensor([0.25], device=dev) if prelu else None
    sp = None if slope is None else slope.data_ptr()
    act = _lib.ACT_PRELU if prelu else _lib.ACT_NONE
    W = 2 * c + 4
    # the sums of the layer above (as its producer's table would give them)
    sums = torch.empty(W, device=dev)
    nbw = L.srx_bn_bwd_ws_floats(m, c)
    bws = torch.empty(nbw, device=dev)
    dy0 = torch.empty_like(dout)
    gg, gb, gp = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
    _lib.call('srx_bn_act_bwd', dout.data_ptr(), y_above.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              sums.data_ptr(), dy0.data_ptr(), m, c, 1, act, 0.0, sp, 1, gg.data_ptr(), gb.data_ptr(), gp.data_ptr() if prelu else None,
              bws.data_ptr(), nbw, s)
    # separate launches: dy0 from above, then the data gradient (+ addend) with or without the reduce epilogue
    dx0, table0 = torch.empty_like(dout), torch.full((rows, W), float('nan'), device=dev)
    if below:
        _lib.call('srx_conv2d_bwd_data_bn', C.byref(d), dy0.data_ptr(), wb.data_ptr(), addend.data_ptr(), dx0.data_ptr(), y_below.data_ptr(),
                  mean_b.data_ptr(), invstd_b.data_ptr(), gamma_b.data_ptr(), beta_b.data_ptr(), None, table0.data_ptr(), s)
    else:
        nws = L.srx_conv2d_bwd_data_ws_floats(C.byref(d))
        ws = torch.empty(max(nws, 4), device=dev)
        _lib.call('srx_conv2d_bwd_data_add', C.byref(d), dy0.data_ptr(), wb.data_ptr(), addend.data_ptr(), dx0.data_ptr(), ws.data_ptr(), nws, s)
    # one launch
    dy1, dx1 = torch.full_like(dout, float('nan')), torch.empty_like(dout)
    table1 = torch.full((rows, W), float('nan'), device=dev)
    _lib.call('srx_conv2d_bwd_data_bn_in', C.byref(d), dout.data_ptr(), y_above.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
              gamma.data_ptr(), beta.data_ptr(), sp, sums.data_ptr(), dy1.data_ptr(), wb.data_ptr(), addend.data_ptr(), dx1.data_ptr(),
              y_below.data_ptr() if below else None, mean_b.data_ptr() if below else None, invstd_b.data_ptr() if below else None,
              gamma_b.data_ptr() if below else None, beta_b.data_ptr() if below else None, None,
              table1.data_ptr() if below else None, s)
    torch.cuda.synchronize()
    # (to rounding, not bit for bit: the two kernels contract the apply expression's multiplies and adds differently)
    assert rel_err(dy1, dy0) < 2e-6 and rel_err(dx1, dx0) < 1e-5
    assert not torch.isnan(dy1).any()   # every pixel written, by the workgroup that owns it
    if below:
        assert rel_err(table1[:, :2 * c], table0[:, :2 * c]) < 1e-5
    # finalize alone (dy = NULL) leaves the sums the full call leaves
    tab = torch.rand(rows, W, generator=g).to(dev)
    s_a, s_b, out_b = torch.empty(W, device=dev), torch.empty(W, device=dev), torch.empty_like(dout)
    z = lambda: torch.zeros(c, device=dev)  # noqa: E731
    _lib.call('srx_bn_act_bwd_finish', None, None, None, None, None, None, tab.data_ptr(), rows, 2, s_a.data_ptr(), None, m, c, act, 0.0,
              sp, z().data_ptr(), z().data_ptr(), gp.data_ptr() if prelu else None, s)
    _lib.call('srx_bn_act_bwd_finish', dout.data_ptr(), y_above.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              tab.data_ptr(), rows, 2, s_b.data_ptr(), out_b.data_ptr(), m, c, act, 0.0, sp, z().data_ptr(), z().data_ptr(),
              gp.data_ptr() if prelu else None, s)
    torch.cuda.synchronize()
    assert torch.equal(s_a[:2 * c + 1], s_b[:2 * c + 1])


@pytest.mark.parametrize('precision', [0, 1])
@pytest.mark.parametrize('n,h,w,cin,cout,stride', [(4, 24, 24, 64, 64, 2), (2, 13, 17, 64, 128, 2), (32, 96, 96, 64, 64, 2),
                                                   (2, 20, 20, 64, 64, 1)])
def test_strided_data_gradient_with_activation_backward(dev, n, h, w, cin, cout, stride, precision):
    """``srx_conv2d_bwd_data_act`` on STRIDED layers (the four stride-parity classes of the data gradient each mask their own
    output pixels): the data gradient followed by the backward of the LeakyReLU that produced the conv's input, against the
    two separate launches -- bit for bit.  (The discriminators' first conv + LeakyReLU under their stride-2 second conv.)"""
    import ctypes as C
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    if precision and n * h * w > 100000:
        pytest.skip('the large geometry is the fp32 step\'s; the bf16 step runs it at 128 x 128 (test_esrgan_gpu.py)')
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, stride, 1, 0, 0, 0.0, 0, precision)
    g = torch.Generator().manual_seed(3 + h)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)  # noqa: E731
    wt = rnd(cout, cin, 3, 3) * 0.05
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(C.byref(d)), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(C.byref(d)), device=dev)
    _lib.call('srx_conv2d_pack', C.byref(d), wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    dy, x = rnd(n, ho, wo, cout), rnd(n, h, w, cin)  # x: the activation output the conv read in the forward pass
    nws = L.srx_conv2d_bwd_data_ws_floats(C.byref(d))
    ws = torch.empty(max(nws, 4), device=dev)
    dx0, dx1, dx2 = torch.empty_like(x), torch.empty_like(x), torch.full_like(x, float('nan'))
    _lib.call('srx_conv2d_bwd_data', C.byref(d), dy.data_ptr(), wb.data_ptr(), dx0.data_ptr(), 0, ws.data_ptr(), nws, s)
    _lib.call('srx_act_bwd_from_out', dx0.data_ptr(), x.data_ptr(), dx1.data_ptr(), x.numel(), _lib.ACT_LRELU, 0.2, s)
    _lib.call('srx_conv2d_bwd_data_act', C.byref(d), dy.data_ptr(), wb.data_ptr(), x.data_ptr(), 0.2, 0, cin, 0, dx2.data_ptr(),
              ws.data_ptr(), nws, s)
    torch.cuda.synchronize()
    assert torch.equal(dx1, dx2)
    r = (lambda t: t.bfloat16().double()) if precision else (lambda t: t.double())   # (precision = 1: bf16-rounded dy and weights)
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), r(wt.cpu()), r(dy.permute(0, 3, 1, 2).cpu()), stride=stride, padding=1)
    ref = ref * torch.where(x.permute(0, 3, 1, 2).cpu() > 0, 1.0, 0.2)
    assert rel_err(dx2.permute(0, 3, 1, 2), ref) < 1e-5


@pytest.mark.parametrize('k,n,h,w', [(9, 2, 40, 70), (3, 1, 33, 45), (3, 40, 96, 96), (9, 1, 7, 5)])
def test_output_conv_with_bf16_products(dev, k, n, h, w):
    """``srx_conv2d_t::precision = 2``: the 64 -> 3 output convs (9x9 SRGAN, 3x3 ESRGAN) with bf16-rounded operands on
    ``v_mfma_f32_4x4x4_16b_bf16`` (inference) against fp64 of the rounded operands; images of several tiles, of less than one
    tile, and enough tiles for the six-rows-per-wave variant."""
    from torchsr_amd import functional as F
    from torchsr_amd.layers import Conv2d
    torch.manual_seed(k + h)
    conv = Conv2d(64, 3, kernel_size=k, stride=1, padding=(k - 1) // 2).to(dev)
    conv._st.precision = 2
    x = torch.rand(n, 64, h, w, device=dev) - 0.5
    with torch.no_grad():
        y = F.to_nchw(conv(F.to_nhwc(x)), 3)
    r16 = lambda t: t.to(torch.bfloat16).double().cpu()  # noqa: E731
    ref = torch.nn.functional.conv2d(r16(x), r16(conv.weight.detach()), conv.bias.detach().double().cpu(), 1, (k - 1) // 2)
    assert rel_err(y, ref) < 2e-5
    conv._st.precision = 0
    with torch.no_grad():
        y0 = F.to_nchw(conv(F.to_nhwc(x)), 3)
    exact = torch.nn.functional.conv2d(x.double().cpu(), conv.weight.detach().double().cpu(), conv.bias.detach().double().cpu(), 1, (k - 1) // 2)
    assert rel_err(y0, exact) < 2e-5 and rel_err(y, exact) > 1e-4  # (and the two settings are different arithmetic)


@pytest.mark.parametrize('prelu', [True, False])
@pytest.mark.parametrize('n,h,w', [(4, 12, 12), (16, 24, 24), (1, 6, 18), (2, 48, 48)])  # (48 wide: two batches of patch loads)
def test_forward_conv_with_batchnorm_on_its_input(dev, prelu, n, h, w):
    """``srx_conv2d_fwd_bn_in`` (normalise + PReLU of the conv below -- or normalise + skip addend, the end of a residual block --
    applied while this conv stages its input, the result tensor written on the side) against the separate launches
    ``srx_bn_act_fwd`` then ``srx_conv2d_fwd``: the same tensor, conv output and BatchNorm partial sums bit for bit (same
    expressions, same kernel body)."""
    import ctypes as C
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    c = 64
    m = n * h * w
    d = _lib.Conv2dDesc(n, h, w, c, c, c, c, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    assert L.srx_conv2d_fwd_bn_in_ok(C.byref(d)) == 1
    g = torch.Generator().manual_seed(7 + n)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)  # noqa: E731
    wt = rnd(c, c, 3, 3) * 0.05
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(C.byref(d)), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(C.byref(d)), device=dev)
    _lib.call('srx_conv2d_pack', C.byref(d), wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    y_in = rnd(n, h, w, c)
    mean, invstd = rnd(c) * 0.1, torch.rand(c, generator=g).to(dev) + 0.5
    gamma, beta = rnd(c), rnd(c) * 0.3
    slope = torch.tensor([0.25], device=dev) if prelu else None
    act = _lib.ACT_PRELU if prelu else _lib.ACT_NONE
    rows = L.srx_conv2d_stat_rows(C.byref(d))
    # separate launches
    a0 = torch.empty_like(y_in)
    res = None if prelu else rnd(n, h, w, c)  # bn2 + skip (no activation, an addend) / bn1 + PReLU (no addend)
    _lib.call('srx_bn_act_fwd', y_in.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
              None if res is None else res.data_ptr(), a0.data_ptr(), m, c, act, 0.0, None if slope is None else slope.data_ptr(), s)
    y0, part0 = torch.empty_like(y_in), torch.empty(rows, c, 2, device=dev)
    nws = L.srx_conv2d_fwd_ws_floats(C.byref(d))
    ws = torch.empty(max(nws, 4), device=dev)
    _lib.call('srx_conv2d_fwd', C.byref(d), a0.data_ptr(), wf.data_ptr(), None, y0.data_ptr(), part0.data_ptr(), ws.data_ptr(), nws, s)
    # one launch
    a1 = torch.full_like(y_in, float('nan'))
    y1, part1 = torch.empty_like(y_in), torch.empty(rows, c, 2, device=dev)
    _lib.call('srx_conv2d_fwd_bn_in', C.byref(d), y_in.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(),
              beta.data_ptr(), None if slope is None else slope.data_ptr(), None if res is None else res.data_ptr(), a1.data_ptr(),
              wf.data_ptr(), None, y1.data_ptr(), part1.data_ptr(), s)
    torch.cuda.synchronize()
    assert torch.equal(a0, a1)          # every pixel written exactly once, by the workgroup that owns it
    assert torch.equal(y0, y1) and torch.equal(part0, part1)
    # and against plain torch
    z = (y_in - mean) * invstd * gamma + beta
    ref_a = torch.where(z > 0, z, z * 0.25) if prelu else z + res
    assert rel_err(a1, ref_a) < 1e-5
    ref_y = torch.nn.functional.conv2d(ref_a.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert rel_err(y1.cpu().double(), ref_y) < 1e-5
    # layers the row-tile kernel does not serve are refused, and say so beforehand
    d2 = _lib.Conv2dDesc(1, 100, 100, c, c, c, c, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    assert L.srx_conv2d_fwd_bn_in_ok(C.byref(d2)) == 0
    rc = L.srx_conv2d_fwd_bn_in(C.byref(d2), y_in.data_ptr(), mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                None, None, a1.data_ptr(), wf.data_ptr(), None, y1.data_ptr(), None, s)
    assert rc != 0 and 'row tile' in _lib.last_error()


@pytest.mark.parametrize('n,h,w,cin', [(2, 32, 32, 192), (3, 7, 32, 96), (1, 5, 16, 64),
                                       (16, 32, 32, 192)])  # BASELINE config 4: 16 384 rows per problem
def test_bf16_weight_gradient_on_image_rows(dev, n, h, w, cin):
    """The image-row bf16 weight-gradient kernel (3x3 / stride 1 / 64 output columns / rows of 16 or 32 pixels: ESRGAN's dense
    blocks) through ``srx_conv2d_bwd_weight_multi_scaled``: two problems with their own multipliers, bias gradients riding
    along, accumulation into existing gradients, against fp64 of the bf16-rounded operands -- images of several rows, of
    few rows (every row a border row) and a 192-strided input read on its first `cin` channels."""
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    d = _lib.Conv2dDesc(n, h, w, cin, 192, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 1)
    arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    r16 = lambda t: t.bfloat16().double()  # noqa: E731
    xs = [rnd((n, 192, h, w), 40 + i) for i in range(2)]
    dys = [rnd((n, 64, h, w), 50 + i) for i in range(2)]
    scales = [0.2, 1.5]
    gw = [torch.full((64, cin, 3, 3), 0.5, device=dev) for _ in range(2)]
    gb = [torch.full((64,), -1.0, device=dev) for _ in range(2)]
    nws = L.srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), 2)
    ws = torch.empty(max(nws, 4), device=dev)
    xd = [nhwc(x, 192).to(dev) for x in xs]
    dd = [nhwc(t).to(dev) for t in dys]
    _lib.call('srx_conv2d_bwd_weight_multi_scaled', C.byref(d), 2, 1, arr(xd), arr(dd), arr(gw), 1, arr(gb), (C.c_float * 2)(*scales),
              ws.data_ptr(), nws, s)
    torch.cuda.synchronize()
    for i in range(2):
        want = scales[i] * torch.nn.grad.conv2d_weight(r16(xs[i][:, :cin]), (64, cin, 3, 3), r16(dys[i]), padding=1) + 0.5
        assert rel_err(gw[i].cpu(), want.float()) < 2e-5, (i, rel_err(gw[i].cpu(), want.float()))
        want_b = scales[i] * dys[i].double().sum((0, 2, 3)) - 1.0
        assert rel_err(gb[i].cpu(), want_b.float()) < 2e-5, i


@pytest.mark.parametrize('n,h,w,cout,shuffle,res', [
    (1, 40, 150, 64, 0, True),      # wide image, ragged last strip (150 = 128 + 22): four 32-pixel segments side by side
    (2, 24, 24, 64, 0, True),       # narrow image: the waves take rows (W <= 32)
    (1, 37, 64, 64, 0, False),      # 2 x 2 segments (W <= 64), odd row count
    (3, 9, 33, 128, 0, False),      # two channel groups written into 128-channel pixels
    (1, 20, 45, 256, 2, False),     # sub-pixel layer: 256 channels, PixelShuffle(2) in the store
    (1, 300, 200, 64, 0, True),     # several row chunks per column strip (persistent workgroups walk more than one item)
])
def test_bf16_native_conv3x3_c64(dev, n, h, w, cout, shuffle, res):
    """``srx_conv3x3_c64_bf16_fwd`` (bf16 tensors in HBM, weights resident in registers, a rolling window of image rows in
    LDS) through the C ABI against fp64 of the same operands: y = bf16(act(conv(x, bf16(W)) + b) [+ skip]).  The kernel sums
    in fp32 and rounds ONCE, so every output is within half a bf16 ulp (8 significant bits: at most 2^-8 relative) plus the
    fp32 summation error of the fp64 value."""
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(1000 * h + w + cout)
    x = (torch.rand(n, 64, h, w, generator=g) - 0.5).bfloat16()
    wt = torch.randn(cout, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    slope = 0.25
    oh, ow, oc = (2 * h, 2 * w, 64) if shuffle else (h, w, cout)
    skip = (torch.rand(n, oc, oh, ow, generator=g) - 0.5).bfloat16() if res else None
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, bd = wt.to(dev), b.to(dev)
    pk = torch.empty(L.srx_conv3x3_c64_bf16_packed_bytes(cout), dtype=torch.uint8, device=dev)
    _lib.call('srx_conv3x3_c64_bf16_pack', wd.data_ptr(), bd.data_ptr(), None, cout, shuffle, pk.data_ptr(), s)
    y = torch.full((n, oh, ow, oc), float('nan'), dtype=torch.bfloat16, device=dev)
    sd = None if skip is None else skip.permute(0, 2, 3, 1).contiguous().to(dev)
    _lib.call('srx_conv3x3_c64_bf16_fwd', n, h, w, cout, shuffle, xd.data_ptr(), pk.data_ptr(), slope,
              None if sd is None else sd.data_ptr(), y.data_ptr(), oc, s)
    torch.cuda.synchronize()
    z = TF.conv2d(x.double(), wt.bfloat16().double(), b.double(), 1, 1)
    if shuffle:
        z = TF.pixel_shuffle(z, 2)
    z = torch.where(z > 0, z, z * slope)
    if skip is not None:
        z = z + skip.double()
    got = y.permute(0, 3, 1, 2).float().cpu().double()
    assert torch.isfinite(got).all()
    err = (got - z).abs()
    bound = z.abs() * (2.0 ** -8 * 1.001) + 2e-6 * z.abs().max()
    assert (err <= bound).all(), ((err - bound).max().item(), err.max().item())
    # ... and it IS a rounding of the right value: re-rounding the fp64 result gives the same bf16 on nearly every element
    same = (z.float().bfloat16().double() == got).double().mean().item()
    assert same > 0.995, same
    # refusals: in place, a shuffled layer with an addend, a channel stride that cannot hold the channels
    with pytest.raises(RuntimeError, match='in place'):
        _lib.call('srx_conv3x3_c64_bf16_fwd', n, h, w, cout, shuffle, xd.data_ptr(), pk.data_ptr(), slope, None, xd.data_ptr(), oc, s)
    with pytest.raises(RuntimeError, match='channel stride'):
        _lib.call('srx_conv3x3_c64_bf16_fwd', n, h, w, cout, shuffle, xd.data_ptr(), pk.data_ptr(), slope, None, y.data_ptr(), 60, s)


@pytest.mark.parametrize('n,h,w,cout', [(1, 40, 150, 3), (2, 9, 33, 3), (1, 300, 130, 3), (1, 5, 7, 1), (1, 64, 256, 2)])
def test_bf16_output_conv_with_taps_as_gemm_columns(dev, n, h, w, cout):
    """``srx_conv9x9_c64_thin_bf16_fwd`` (the 9x9 64 -> 3 output conv as a GEMM with N = 9 row taps x 3 channels, the vertical
    sum in an LDS ring of output rows) through the C ABI against fp64 of the bf16 operands: ragged strips, images shorter than
    the 9-row window, several row chunks per strip, fewer than three output channels."""
    from torchsr_amd import _lib
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(7 * h + w)
    x = (torch.rand(n, 64, h, w, generator=g) - 0.5).bfloat16()
    wt = torch.randn(cout, 64, 9, 9, generator=g) * (1.0 / 5184) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd, bd = wt.to(dev), b.to(dev)
    pk = torch.empty(L.srx_conv9x9_c64_thin_bf16_packed_bytes(), dtype=torch.uint8, device=dev)
    _lib.call('srx_conv9x9_c64_thin_bf16_pack', wd.data_ptr(), bd.data_ptr(), cout, pk.data_ptr(), s)
    y = torch.full((n, h, w, 4), float('nan'), device=dev)
    _lib.call('srx_conv9x9_c64_thin_bf16_fwd', n, h, w, xd.data_ptr(), pk.data_ptr(), y.data_ptr(), s)
    torch.cuda.synchronize()
    z = TF.conv2d(x.double(), wt.bfloat16().double(), b.double(), 1, 4)
    got = y.permute(0, 3, 1, 2).cpu().double()
    assert torch.isfinite(got).all()
    assert ((got[:, :cout] - z).abs().max() / z.abs().max()).item() < 2e-5
    assert float(got[:, cout:].abs().max()) == 0.0
