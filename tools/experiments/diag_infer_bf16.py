"""Where do the eval-mode bf16-products generator (FoldedConv path) and the bf16-products oracle part ways?  Layer by layer
on one small frame (developer diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from oracle import srgan as O  # noqa: E402
from oracle.weights import closed_form_state  # noqa: E402
from test_cli_gpu import _bn_folded_state  # noqa: E402
from torchsr_amd import functional as F  # noqa: E402
from torchsr_amd.layers import set_conv_precision  # noqa: E402
from torchsr_amd.srgan.generator import Generator  # noqa: E402

dev = torch.device('cuda:0')
gen = Generator().to(dev)
sd = closed_form_state(gen.state_dict())
gen.load_state_dict(sd)
gen.eval()
folded = _bn_folded_state(sd)
x = torch.rand(1, 3, 64, 80, generator=torch.Generator().manual_seed(3))


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-9)).item()


def nchw(t, c=None):
    return F.to_nchw(t, c).cpu()


for mode in ('fp32', 'bf16'):
    set_conv_precision(gen, mode)
    with torch.no_grad():
        x4 = F.to_nhwc(x.to(dev), 4)
        gen.forward_nhwc(x4)  # builds the folded layers
        f = gen.__dict__['_folded']
        got = {}
        c1 = f[0](x4)
        got['conv1'] = nchw(c1)
        t = c1
        for i, blk in enumerate(gen.blocks):
            t = blk(t)
            if i in (0, 1, 7, 15):
                got[f'block{i}'] = nchw(t)
        out = f[1](t, residual=c1)
        got['trunk'] = nchw(out)
        u = out
        for i, layer in enumerate(gen.conv_layers):
            u = layer(u)
            got[f'up{i}'] = nchw(u)
        got['out'] = nchw(gen.conv3(u), 3)
        import contextlib
        ctx = O.bf16_products() if mode == 'bf16' else contextlib.nullcontext()
        with ctx:
            s = folded
            ref = {}
            r1 = torch.nn.functional.prelu(O.conv2d(x, s['conv1.0.weight'], s['conv1.0.bias'], 1, 4), s['conv1.1.weight'])
            ref['conv1'] = r1
            t = r1
            for i in range(16):
                t = O.residual_block(s, f'blocks.{i}.', t, False)
                if i in (0, 1, 7, 15):
                    ref[f'block{i}'] = t
            c2 = O._bn(s, 'conv2.1.', O.conv2d(t, s['conv2.0.weight'], None, 1, 1), False)
            o = r1 + c2
            ref['trunk'] = o
            for i in range(2):
                o = O.subpixel_layer(s, f'conv_layers.{i}.', o)
                ref[f'up{i}'] = o
            ref['out'] = O.conv2d(o, s['conv3.weight'], s['conv3.bias'], 1, 4)
    print(mode, ' '.join(f'{k}:{rel(got[k], ref[k]):.2e}' for k in got))

# --- block 0, conv1: the kernel against its own operands, and its operands against the oracle's
set_conv_precision(gen, 'bf16')
with torch.no_grad():
    x4 = F.to_nhwc(x.to(dev), 4)
    gen.forward_nhwc(x4)
    f = gen.__dict__['_folded']
    c1 = f[0](x4)
    blk = gen.blocks[0]
    blk(c1)
    fa = blk.__dict__['_folded'][0]
    a = nchw(fa(c1))
    r16 = lambda t: t.to(torch.bfloat16).double()  # noqa: E731
    c1n = nchw(c1)
    w_p, b_p = fa.w.cpu(), fa.b.cpu()
    slope = fa.st.slope
    z = torch.nn.functional.conv2d(r16(c1n), r16(w_p), b_p.double(), 1, 1)
    want = torch.where(z > 0, z, z * slope).float()
    print('kernel vs its own operands:', rel(a, want))
    w_o = folded['blocks.0.conv1.weight']
    print('folded weight product vs test: max abs diff', (w_p - w_o).abs().max().item(), 'max', w_o.abs().max().item(),
          'bf16 differs on', int((w_p.bfloat16() != w_o.bfloat16()).sum()), 'of', w_o.numel())
    b_o = folded['blocks.0.bn1.bias']
    print('folded bias diff', (b_p - b_o).abs().max().item(), 'slope', slope, float(sd['blocks.0.prelu.weight']))
    with O.bf16_products():
        zo = O.conv2d(c1n, w_o, None, 1, 1) + b_o.view(1, -1, 1, 1)
    print('oracle conv on the product input vs fp64 of same:', rel(zo, z.float()))
    fb = blk.__dict__['_folded'][1]
    a_dev = fa(c1)
    y = nchw(fb(a_dev, residual=c1))
    z2 = torch.nn.functional.conv2d(r16(nchw(a_dev)), r16(fb.w.cpu()), fb.b.cpu().double(), 1, 1) + c1n.double()
    print('conv2 + residual vs its own operands:', rel(y, z2.float()))
    y_nores = nchw(fb(a_dev))
    print('conv2 alone vs its own operands:', rel(y_nores, (z2 - c1n.double()).float()))
    with O.bf16_products():
        ro = O.residual_block(folded, 'blocks.0.', c1n, False)
    print('oracle block on the product input vs product block:', rel(y, ro), ' vs fp64 of the product operands:', rel(ro, z2.float()))
    w2 = folded['blocks.0.conv2.weight']
    print('conv2 folded weight: bf16 differs on', int((fb.w.cpu().bfloat16() != w2.bfloat16()).sum()), 'bias diff',
          (fb.b.cpu() - folded['blocks.0.bn2.bias']).abs().max().item())
