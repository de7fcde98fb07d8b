"""Winograd F(2x2, 3x3) convolution (csrc/wino.hip) through the C ABI against stock torch fp64 on CPU: forward with bias + ReLU
(torchvision VGG19 cfg 'E' layers, srgan/loss.py:30-31) and the data gradient with the ReLU mask of the layer below folded in.
The arithmetic is fp32 throughout; what differs from the direct convolution is the order of operations, so the tolerance is a
few fp32 roundings of the tensor's scale (2e-5), as for the exact-fp32 direct kernels."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


CASES = [
    # N, H, W, Cin, Cout
    (2, 24, 24, 64, 64),     # VGG conv1_2 geometry (2304-tile rows cut into 32-tile blocks)
    (2, 12, 12, 256, 256),   # conv3_x
    (1, 6, 6, 512, 512),     # conv5_x: 9 tiles, the planner splits the input channels
    (1, 10, 6, 128, 96),     # 15 tiles (a partial block), 96 output channels -> 32-column workgroups
    (3, 8, 20, 32, 160),     # one chunk of input channels
    (16, 24, 24, 256, 256),  # the training batch's data-gradient size: 2.25 rounds of tiles -> the last round's tiles are cut in parts
    (16, 12, 12, 512, 512),  # the same with 32-column workgroups, 16 chunks per tile
]


@pytest.mark.parametrize('cfg', CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_forward_and_data_gradient(dev, cfg):
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).relu()             # a ReLU output, like every VGG layer's input
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    dy = torch.randn(n, cout, h, w, generator=g)
    xr = x.double().requires_grad_(True)
    yr = TF.relu(TF.conv2d(xr, wt.double(), bias.double(), padding=1))
    pre = TF.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    # data gradient of the conv alone for an output gradient dy, then the mask of the ReLU that produced x
    dx_plain = torch.autograd.grad(TF.conv2d(xr, wt.double(), None, padding=1), xr, dy.double())[0]
    dx_masked = dx_plain * (x.double() > 0)

    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 0)
    dref = C.byref(d)
    # (srx_wino_applicable also asks whether the layer is large enough to be worth it; the kernel itself takes any such shape)
    s = torch.cuda.current_stream().cuda_stream
    nf = L.srx_wino_packed_floats(dref)
    assert nf == 16 * cin * cout
    wg = wt.to(dev)
    uf, ub = torch.empty(nf, device=dev), torch.empty(nf, device=dev)
    _lib.call('srx_wino_pack', dref, wg.data_ptr(), uf.data_ptr(), 0, s)
    _lib.call('srx_wino_pack', dref, wg.data_ptr(), ub.data_ptr(), 1, s)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    y = torch.empty(n, h, w, cout, device=dev)
    nws = L.srx_wino_ws_floats(dref, 0)
    ws = torch.empty(max(nws, 4), device=dev)
    _lib.call('srx_wino_fwd', dref, xg.data_ptr(), uf.data_ptr(), bias.to(dev).data_ptr(), y.data_ptr(), ws.data_ptr(), nws, s)
    assert rel(y.permute(0, 3, 1, 2), yr) < 2e-5
    # no activation, no bias: the raw convolution
    d0 = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_NONE, 0.0, 0, 0)
    _lib.call('srx_wino_fwd', C.byref(d0), xg.data_ptr(), uf.data_ptr(), None, y.data_ptr(), ws.data_ptr(), nws, s)
    assert rel(y.permute(0, 3, 1, 2), pre - bias.double().view(1, -1, 1, 1)) < 2e-5
    # data gradient, plain and with the ReLU mask of the layer below
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    dx = torch.empty(n, h, w, cin, device=dev)
    nwb = L.srx_wino_ws_floats(dref, 1)
    wsb = torch.empty(max(nwb, 4), device=dev)
    _lib.call('srx_wino_bwd_data', dref, dyg.data_ptr(), ub.data_ptr(), None, dx.data_ptr(), wsb.data_ptr(), nwb, s)
    assert rel(dx.permute(0, 3, 1, 2), dx_plain) < 2e-5
    _lib.call('srx_wino_bwd_data', dref, dyg.data_ptr(), ub.data_ptr(), xg.data_ptr(), dx.data_ptr(), wsb.data_ptr(), nwb, s)
    assert rel(dx.permute(0, 3, 1, 2), dx_masked) < 2e-5
    plan = (C.c_int * 6)()
    _lib.call('srx_wino_plan', dref, 0, plan)
    # (a workspace exactly when every tile's channels are split, or when the tiles of the last round of the chip are cut in parts)
    assert plan[0] in (32, 64) and plan[1] >= 1 and plan[5] >= 1 and (nws > 0) == (plan[1] > 1 or plan[5] > 1)
    if cfg == (16, 24, 24, 256, 256):
        assert plan[1] == 1 and plan[5] > 1   # 576 tiles on 256 CUs: two rounds of whole tiles + a short third one


def test_winograd_refuses_what_it_does_not_implement(dev):
    from torchsr_amd import _lib
    L = _lib.lib()
    for desc in (_lib.Conv2dDesc(2, 24, 24, 64, 64, 64, 64, 3, 3, 2, 1, 0, 0, 0.0, 0, 0),    # stride 2
                 _lib.Conv2dDesc(2, 24, 24, 3, 4, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0),      # 3 input channels
                 _lib.Conv2dDesc(2, 23, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0),    # odd height
                 _lib.Conv2dDesc(2, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 1),    # bf16 products
                 _lib.Conv2dDesc(2, 24, 24, 64, 192, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)):  # strided input rows (dense block)
        assert L.srx_wino_applicable(C.byref(desc)) == 0
        assert L.srx_wino_packed_floats(C.byref(desc)) == 0
    # large enough and of the right shape: yes; the residual tower's small 64 -> 64 layers keep their row-tile kernel
    assert L.srx_wino_applicable(C.byref(_lib.Conv2dDesc(32, 24, 24, 256, 256, 256, 256, 3, 3, 1, 1, 0, 1, 0.0, 0, 0))) == 1
    assert L.srx_wino_applicable(C.byref(_lib.Conv2dDesc(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0))) == 0


def test_vgg_stack_on_winograd_equals_the_direct_kernels(dev, monkeypatch):
    """The perceptual-loss node (functional.frozen_conv_stack) with its wide layers on Winograd against the same node on the
    direct gather-GEMM kernels (developer switch): loss and input gradient agree to fp32 rounding of their scale."""
    import torch
    from torchsr_amd import _dev, functional as F
    from torchsr_amd.srgan.loss import VGGLoss
    torch.manual_seed(0)
    vgg = VGGLoss(weights='random').to(dev)
    src = torch.rand(2, 3, 96, 96, device=dev)
    tgt = torch.rand(2, 3, 96, 96, device=dev)
    out = {}
    for direct in (False, True):
        monkeypatch.setattr(_dev, 'NO_WINO', direct)
        s = src.clone().requires_grad_(True)
        loss = vgg(s, tgt)
        loss.backward()
        out[direct] = (loss.detach().clone(), s.grad.clone())
    assert rel(out[False][0], out[True][0]) < 1e-5
    # the gradient passes through 16 ReLUs and 4 max-pools: a pre-activation within rounding of 0 (or a tie in a pooling window)
    # takes the other branch in any two fp32 evaluations, so single pixels differ; the tensors as a whole must not
    a, b = out[False][1].double().flatten(), out[True][1].double().flatten()
    assert float((a - b).norm() / b.norm()) < 2e-3
    assert float(torch.dot(a, b) / (a.norm() * b.norm())) > 0.99999


@pytest.mark.parametrize('cfg', [(4, 24, 24, 64, 128), (2, 12, 12, 256, 512), (1, 10, 6, 128, 96),
                                 (32, 24, 24, 128, 256), (31, 12, 12, 256, 512)],   # 2.25 rounds of tiles: the last round's are cut in parts
                         ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_forward_with_batchnorm_partials(dev, cfg):
    """srx_wino_fwd_stats: the conv's output plus, per block of 32 consecutive 2x2 tiles, the per-channel (sum, sum of squares)
    that a training-mode BatchNorm2d behind the layer reduces (srgan/discriminator.py:35-61)."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    yr = TF.conv2d(x.double(), wt.double(), None, padding=1)
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_NONE, 0.0, 0, 0)
    dref = C.byref(d)
    s = torch.cuda.current_stream().cuda_stream
    uf = torch.empty(L.srx_wino_packed_floats(dref), device=dev)
    _lib.call('srx_wino_pack', dref, wt.to(dev).data_ptr(), uf.data_ptr(), 0, s)
    rows = L.srx_wino_stat_rows(dref)
    assert rows == -(-(n * (h // 2) * (w // 2)) // 32)
    y = torch.empty(n, h, w, cout, device=dev)
    part = torch.full((rows, cout, 2), float('nan'), device=dev)
    nws = L.srx_wino_ws_floats(dref, 2)
    ws = torch.empty(max(nws, 4), device=dev)
    _lib.call('srx_wino_fwd_stats', dref, x.permute(0, 2, 3, 1).contiguous().to(dev).data_ptr(), uf.data_ptr(), None, y.data_ptr(),
              part.data_ptr(), ws.data_ptr(), nws, s)
    assert rel(y.permute(0, 3, 1, 2), yr) < 2e-5
    # reference partials: tiles in (image, tile row, tile column) order, 32 per block
    tiles = yr.reshape(n, cout, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(-1, cout, 4)   # [T][C][4 pixels]
    pad = rows * 32 - tiles.shape[0]
    tiles = torch.cat([tiles, torch.zeros(pad, cout, 4, dtype=tiles.dtype)]) if pad else tiles
    blocks = tiles.reshape(rows, 32, cout, 4)
    s1, s2 = blocks.sum(dim=(1, 3)), blocks.square().sum(dim=(1, 3))
    assert rel(part[..., 0], s1) < 2e-5 and rel(part[..., 1], s2) < 2e-5


@pytest.mark.parametrize('cfg', [(1, 24, 40, 64, 64), (2, 12, 12, 64, 128)], ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_inference_forms(dev, cfg):
    """``srx_wino_fwd_act``: LeakyReLU (the folded single-parameter PReLU of a residual block) and the skip addend in the Winograd
    epilogue (functional.FoldedConv at inference, srgan/residual.py:86-91) against fp64."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(n, cout, h, w, generator=g)
    pre = TF.conv2d(x.double(), wt.double(), bias.double(), padding=1)
    s = torch.cuda.current_stream().cuda_stream
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    rg = res.permute(0, 2, 3, 1).contiguous().to(dev)
    bg = bias.to(dev)
    for act, slope, with_res in ((_lib.ACT_LRELU, 0.25, False), (_lib.ACT_NONE, 0.0, True), (_lib.ACT_LRELU, 0.1, True), (_lib.ACT_RELU, 0.0, True)):
        d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, act, slope, 0, 0)
        dref = C.byref(d)
        uf = torch.empty(L.srx_wino_packed_floats(dref), device=dev)
        _lib.call('srx_wino_pack', dref, wt.to(dev).data_ptr(), uf.data_ptr(), 0, s)
        nws = L.srx_wino_ws_floats(dref, 0)
        ws = torch.empty(max(nws, 4), device=dev)
        y = torch.empty(n, h, w, cout, device=dev)
        _lib.call('srx_wino_fwd_act', dref, xg.data_ptr(), uf.data_ptr(), bg.data_ptr(), rg.data_ptr() if with_res else None, y.data_ptr(),
                  ws.data_ptr(), nws, s)
        want = pre if act == _lib.ACT_NONE else (pre.clamp_min(0) if act == _lib.ACT_RELU else torch.where(pre > 0, pre, pre * slope))
        if with_res:
            want = want + res.double()
        assert rel(y.permute(0, 3, 1, 2), want) < 2e-5, (act, slope, with_res)


def test_winograd_forward_with_pixel_shuffle(dev):
    """The sub-pixel conv of the generator at inference (srgan/residual.py:27-29: Conv2d(64, 256, 3, 1, 1), PixelShuffle(2), PReLU
    -- the folded PReLU commutes with the shuffle): Winograd with the shuffle in its store against fp64."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = 2, 12, 20, 64, 256
    L = _lib.lib()
    g = torch.Generator().manual_seed(33)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    pre = TF.pixel_shuffle(TF.conv2d(x.double(), wt.double(), bias.double(), padding=1), 2)
    want = torch.where(pre > 0, pre, pre * 0.25)
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout // 4, 3, 3, 1, 1, 2, _lib.ACT_LRELU, 0.25, 0, 0)
    dref = C.byref(d)
    assert L.srx_wino_applicable(dref) == 0          # (the training path keeps the direct kernel for these layers)
    s = torch.cuda.current_stream().cuda_stream
    uf = torch.empty(L.srx_wino_packed_floats(dref), device=dev)
    assert uf.numel() == 16 * cin * cout
    _lib.call('srx_wino_pack', dref, wt.to(dev).data_ptr(), uf.data_ptr(), 0, s)
    y = torch.empty(n, 2 * h, 2 * w, cout // 4, device=dev)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    _lib.call('srx_wino_fwd_act', dref, xg.data_ptr(), uf.data_ptr(), bias.to(dev).data_ptr(), None, y.data_ptr(), None, 0, s)
    assert rel(y.permute(0, 3, 1, 2), want) < 2e-5


@pytest.mark.parametrize('cfg', [(32, 24, 24, 256, 256), (32, 12, 12, 512, 512), (32, 96, 96, 64, 64), (16, 48, 48, 128, 128),
                                 (32, 6, 6, 512, 512)], ids=lambda c: 'x'.join(map(str, c)))
def test_winograd_equals_the_direct_kernel_at_the_step_sizes(dev, cfg):
    """At the sizes the BASELINE step runs (batch 32 / 16 of 96 x 96 crops through VGG19: too large for a CPU reference in the
    suite's time) the Winograd forward and data gradient against the exact-fp32 direct gather-GEMM on the same operands
    (``srx_conv2d_fwd`` / ``srx_conv2d_bwd_data_act``, themselves pinned against fp64 at small sizes): 2e-5 of the output scale,
    and linearity of the Winograd path in its input (conv(2 x + x') = 2 conv(x) + conv(x') without bias)."""
    from torchsr_amd import _lib
    n, h, w, cin, cout = cfg
    L = _lib.lib()
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, h, w, cin, generator=g).relu().to(dev)
    x2 = torch.randn(n, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    dy = torch.randn(n, h, w, cout, generator=g).to(dev)
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 0)
    d0 = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_NONE, 0.0, 0, 0)
    dref = C.byref(d)
    wf = torch.empty(L.srx_conv2d_packed_fwd_floats(dref), device=dev)
    wb = torch.empty(L.srx_conv2d_packed_bwd_floats(dref), device=dev)
    _lib.call('srx_conv2d_pack', dref, wt.data_ptr(), wf.data_ptr(), wb.data_ptr(), s)
    uf, ub = torch.empty(L.srx_wino_packed_floats(dref), device=dev), torch.empty(L.srx_wino_packed_floats(dref), device=dev)
    _lib.call('srx_wino_pack', dref, wt.data_ptr(), uf.data_ptr(), 0, s)
    _lib.call('srx_wino_pack', dref, wt.data_ptr(), ub.data_ptr(), 1, s)

    def ws_for(nfl):
        return torch.empty(max(int(nfl), 4), device=dev)
    ya, yb = torch.empty(n, h, w, cout, device=dev), torch.empty(n, h, w, cout, device=dev)
    nws = L.srx_conv2d_fwd_ws_floats(dref)
    _lib.call('srx_conv2d_fwd', dref, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), ya.data_ptr(), None, ws_for(nws).data_ptr(), nws, s)
    nww = L.srx_wino_ws_floats(dref, 0)
    _lib.call('srx_wino_fwd', dref, x.data_ptr(), uf.data_ptr(), bias.data_ptr(), yb.data_ptr(), ws_for(nww).data_ptr(), nww, s)
    assert rel(yb, ya) < 2e-5
    da, db = torch.empty_like(x), torch.empty_like(x)
    nws = L.srx_conv2d_bwd_data_ws_floats(dref)
    _lib.call('srx_conv2d_bwd_data_act', dref, dy.data_ptr(), wb.data_ptr(), x.data_ptr(), 0.0, 0, cin, 0, da.data_ptr(),
              ws_for(nws).data_ptr(), nws, s)
    nww = L.srx_wino_ws_floats(dref, 1)
    _lib.call('srx_wino_bwd_data', dref, dy.data_ptr(), ub.data_ptr(), x.data_ptr(), db.data_ptr(), ws_for(nww).data_ptr(), nww, s)
    assert rel(db, da) < 2e-5
    # linearity (no bias, no activation)
    nww = L.srx_wino_ws_floats(C.byref(d0), 0)
    y1, y2, y3 = (torch.empty(n, h, w, cout, device=dev) for _ in range(3))
    mix = 2.0 * x + x2
    for src, dst in ((x, y1), (x2, y2), (mix, y3)):
        _lib.call('srx_wino_fwd', C.byref(d0), src.data_ptr(), uf.data_ptr(), None, dst.data_ptr(), ws_for(nww).data_ptr(), nww, s)
    assert rel(y3, 2.0 * y1 + y2) < 2e-5


def test_every_forced_plan_computes_the_same_convolution(dev):
    """srx_wino_force_plan (the measurement aid behind tools/lab/wino_sweep.cpp): every plan the sweep times -- 64- and 32-column
    workgroups, channel splits of every tile, the last round's tiles cut in parts -- computes the layer the planner's own plan
    computes (forward with bias + ReLU and masked data gradient, against fp64), and (0, 0, 0) hands the choice back."""
    from torchsr_amd import _lib
    L = _lib.lib()
    n, h, w, cin, cout = 16, 24, 24, 256, 128   # 2304 tiles: 72 tile blocks x 2 / 4 channel blocks; data gradient 256 <- 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g).relu()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    dy = torch.randn(n, cout, h, w, generator=g)
    xr = x.double().requires_grad_(True)
    yr = TF.relu(TF.conv2d(xr, wt.double(), bias.double(), padding=1))
    dxr = torch.autograd.grad(TF.conv2d(xr, wt.double(), None, padding=1), xr, dy.double())[0] * (x.double() > 0)
    d = _lib.Conv2dDesc(n, h, w, cin, cin, cout, cout, 3, 3, 1, 1, 0, _lib.ACT_RELU, 0.0, 0, 0)
    dref = C.byref(d)
    s = torch.cuda.current_stream().cuda_stream
    nf = L.srx_wino_packed_floats(dref)
    wg = wt.to(dev)
    uf, ub = torch.empty(nf, device=dev), torch.empty(nf, device=dev)
    _lib.call('srx_wino_pack', dref, wg.data_ptr(), uf.data_ptr(), 0, s)
    _lib.call('srx_wino_pack', dref, wg.data_ptr(), ub.data_ptr(), 1, s)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    dyg = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    bg = bias.to(dev)
    plan = (C.c_int * 6)()
    seen = set()
    try:
        for forced in ((0, 0, 0), (64, 1, 1), (32, 1, 1), (64, 2, 1), (32, 3, 1), (64, 8, 1), (64, 1, 2), (32, 1, 4)):
            _lib.call('srx_wino_force_plan', *forced)
            for which in (0, 1):
                _lib.call('srx_wino_plan', dref, which, plan)
                seen.add((which, plan[0], plan[1], plan[5]))
                nws = L.srx_wino_ws_floats(dref, which)
                ws = torch.empty(max(nws, 4), device=dev)
                if which == 0:
                    y = torch.full((n, h, w, cout), float('nan'), device=dev)
                    _lib.call('srx_wino_fwd', dref, xg.data_ptr(), uf.data_ptr(), bg.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, s)
                    assert rel(y.permute(0, 3, 1, 2), yr) < 2e-5, (forced, tuple(plan))
                else:
                    dx = torch.full((n, h, w, cin), float('nan'), device=dev)
                    _lib.call('srx_wino_bwd_data', dref, dyg.data_ptr(), ub.data_ptr(), xg.data_ptr(), dx.data_ptr(), ws.data_ptr(), nws, s)
                    assert rel(dx.permute(0, 3, 1, 2), dxr) < 2e-5, (forced, tuple(plan))
    finally:
        _lib.call('srx_wino_force_plan', 0, 0, 0)
    # the forced plans were really taken where the layer can run them (not silently replaced by the planner's)
    assert {(0, 64, 2, 1), (0, 32, 3, 1), (0, 64, 8, 1), (1, 64, 2, 1), (0, 32, 1, 1)} <= seen, seen  # (the gradient contracts 128 channels: 4 chunks, no 8 splits)
    with pytest.raises(RuntimeError):
        _lib.call('srx_wino_force_plan', 48, 1, 1)
