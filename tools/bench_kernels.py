"""Developer micro-benchmark: time the conv kernels on the layer shapes of the SRGAN step.

    python tools/bench_kernels.py [--reps 20]

Prints per-shape forward / data-gradient / weight-gradient time and TFLOP/s against the
157.3 TFLOP/s fp32 MFMA peak of MI355X.  Not part of the product or of the judged bench.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchsr_amd import functional as F  # noqa: E402
from torchsr_amd.layers import Conv2d  # noqa: E402

SHAPES = [
    # name, N, H, W, Cin, Cout, k, s, p, shuffle
    ('G.res 64->64 @24', 16, 24, 24, 64, 64, 3, 1, 1, 0),
    ('G.conv1 9x9 3->64 @24', 16, 24, 24, 3, 64, 9, 1, 4, 0),
    ('G.sub1 64->256 @24', 16, 24, 24, 64, 256, 3, 1, 1, 2),
    ('G.sub2 64->256 @48', 16, 48, 48, 64, 256, 3, 1, 1, 2),
    ('G.conv3 9x9 64->3 @96', 16, 96, 96, 64, 3, 9, 1, 4, 0),
    ('D.c1 3->64 @96', 16, 96, 96, 3, 64, 3, 1, 1, 0),
    ('D.c2 64->64 s2 @96', 16, 96, 96, 64, 64, 3, 2, 1, 0),
    ('D.c3 64->128 @48', 16, 48, 48, 64, 128, 3, 1, 1, 0),
    ('D.c4 128->128 s2 @48', 16, 48, 48, 128, 128, 3, 2, 1, 0),
    ('D.c5 128->256 @24', 16, 24, 24, 128, 256, 3, 1, 1, 0),
    ('D.c6 256->256 s2 @24', 16, 24, 24, 256, 256, 3, 2, 1, 0),
    ('D.c7 256->512 @12', 16, 12, 12, 256, 512, 3, 1, 1, 0),
    ('D.c8 512->512 s2 @12', 16, 12, 12, 512, 512, 3, 2, 1, 0),
    ('V.1_2 64->64 @96', 16, 96, 96, 64, 64, 3, 1, 1, 0),
    ('V.2_2 128->128 @48', 16, 48, 48, 128, 128, 3, 1, 1, 0),
    ('V.3_2 256->256 @24', 16, 24, 24, 256, 256, 3, 1, 1, 0),
    ('V.4_2 512->512 @12', 16, 12, 12, 512, 512, 3, 1, 1, 0),
    ('V.5_2 512->512 @6', 16, 6, 6, 512, 512, 3, 1, 1, 0),
    ('D.c2 64->64 s2 @96 N=32', 32, 96, 96, 64, 64, 3, 2, 1, 0),
    ('D.c4 128->128 s2 @48 N=32', 32, 48, 48, 128, 128, 3, 2, 1, 0),
    ('D.c6 256->256 s2 @24 N=32', 32, 24, 24, 256, 256, 3, 2, 1, 0),
    ('D.c8 512->512 s2 @12 N=32', 32, 12, 12, 512, 512, 3, 2, 1, 0),
    ('V.1_2 N=32', 32, 96, 96, 64, 64, 3, 1, 1, 0),
    ('V.5_2 N=32', 32, 6, 6, 512, 512, 3, 1, 1, 0),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', type=str, default='')
    ap.add_argument('--graph', action='store_true', help='time the forward as a replayed hipGraph of `reps` calls')
    ap.add_argument('--bf16', action='store_true', help='bf16 products (srx_conv2d_t::precision = 1)')
    ap.add_argument('--shape', action='append', default=[], help='extra 3x3/s1/p1 layer: N,H,W,Cin,Cout (repeatable)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    print(f'{"layer":26s} {"GF":>7s} | {"fwd us":>8s} {"TF/s":>6s} | {"dgrad us":>8s} {"TF/s":>6s} | {"wgrad us":>8s} {"TF/s":>6s}')
    shapes = list(SHAPES)
    for spec in args.shape:
        n, h, w, cin, cout = (int(v) for v in spec.split(','))
        shapes.append((f'custom {spec}', n, h, w, cin, cout, 3, 1, 1, 0))
    for name, n, h, w, cin, cout, k, s, p, sh in shapes:
        if args.only and args.only not in name:
            continue
        conv = Conv2d(cin, cout, k, s, p, bias=False, shuffle=sh).to(dev)
        conv._st.precision = 1 if args.bf16 else 0
        cin_s = (cin + 3) // 4 * 4
        x = torch.rand(n, h, w, cin_s, device=dev).requires_grad_(True)
        y = conv(x)
        gy = torch.rand_like(y)
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        gf = 2.0 * n * ho * wo * cout * cin * k * k / 1e9
        if args.graph:
            xd0 = x.detach()
            with torch.no_grad():
                conv(xd0)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for _ in range(args.reps):
                        conv(xd0)
            t_f = timeit(g.replay, 5) / args.reps

            def bwd_graph(wrt_weight):
                # forward + one gradient per repetition inside the capture (as the trainer does); the
                # forward time measured above is subtracted.  Leaves are created inside the capture and
                # the weight gradient goes through the direct-accumulation sink, so that no
                # AccumulateGrad node bound to the default stream takes part.
                conv.weight.requires_grad_(wrt_weight)
                F.direct_grads[0] = True
                if wrt_weight:
                    conv.weight.grad = torch.zeros_like(conv.weight)

                def once():
                    if wrt_weight:
                        conv(xd0).backward(gy)
                    else:
                        xin = xd0.detach().requires_grad_(True)
                        torch.autograd.grad(conv(xin), xin, gy)
                once()
                torch.cuda.synchronize()
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb):
                    for _ in range(args.reps):
                        once()
                return timeit(gb.replay, 5) / args.reps - t_f
            t_d = bwd_graph(False)
            t_w = bwd_graph(True)
            print(f'{name:26s} {gf:7.3f} | graph fwd {t_f:7.2f} us {gf / t_f * 1e3:6.1f} | dgrad {t_d:7.2f} us '
                  f'{gf / t_d * 1e3:6.1f} | wgrad {t_w:7.2f} us {gf / t_w * 1e3:6.1f} TF/s', flush=True)
            continue
        t_f = timeit(lambda: conv(x.detach()), args.reps)
        xd = x.detach().requires_grad_(True)
        conv.weight.requires_grad_(False)
        yd = conv(xd)
        t_d = timeit(lambda: torch.autograd.grad(yd, xd, gy, retain_graph=True), args.reps)
        conv.weight.requires_grad_(True)
        yw = conv(x.detach())
        t_w = timeit(lambda: torch.autograd.grad(yw, conv.weight, gy, retain_graph=True), args.reps)
        print(f'{name:26s} {gf:7.3f} | {t_f:8.1f} {gf / t_f * 1e3:6.1f} | {t_d:8.1f} {gf / t_d * 1e3:6.1f} | '
              f'{t_w:8.1f} {gf / t_w * 1e3:6.1f}', flush=True)


if __name__ == '__main__':
    main()
