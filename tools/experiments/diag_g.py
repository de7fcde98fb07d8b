"""Developer diagnostic: per-parameter gradient error of the SRGAN generator vs the CPU oracle."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import srgan as O  # noqa: E402
from oracle.weights import closed_form_state, seeded_input  # noqa: E402
from torchsr_amd.srgan.generator import Generator  # noqa: E402

dev = torch.device('cuda:0')
gen = Generator()
sd = closed_form_state(gen.state_dict())
gen.load_state_dict(sd)
gen = gen.to(dev).train()
x = seeded_input((2, 3, 12, 12), 7)
xg = x.to(dev).requires_grad_(True)
y = gen(xg)
y.square().mean().backward()
so = {k: v.clone() for k, v in sd.items()}
names = [k for k, v in so.items() if v.is_floating_point() and 'running_' not in k]
leaves = O._leaves(so)
xo = x.clone().requires_grad_(True)
yo = O.generator_forward(so, xo, True)
yo.square().mean().backward()
print('y', ((y.cpu() - yo).abs().max() / yo.abs().max()).item())
grads = dict(gen.named_parameters())
for k, leaf in zip(names, leaves):
    g = grads[k].grad.cpu()
    e = ((g - leaf.grad).abs().max() / leaf.grad.abs().max().clamp_min(1e-12)).item()
    print(f'{k:40s} {e:.3e}  |g| {leaf.grad.abs().max().item():.3e}')
print('dx', ((xg.grad.cpu() - xo.grad).abs().max() / xo.grad.abs().max()).item())
