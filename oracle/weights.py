"""ORACLE helper: closed-form, key-addressed parameter values.

Golden fixtures must not depend on ``torch.manual_seed`` streams or on module
construction order, and full-size weights (SRGAN D 94 MB, VGG19 80 MB) must cost
nothing to store.  Every ``state_dict`` entry is therefore filled, by key name, from

    value[i] = offset(key) + amp(key, shape) * u(crc32(key), i),   u = splitmix64 hash -> [-1, 1)

evaluated in integer / float64 arithmetic and rounded to float32.  The same function fills the imported
reference modules (``gen_golden.py``) and the modules under test.
"""
import zlib
from typing import Dict

import numpy as np
import torch


def _wave(n: int, key: str) -> np.ndarray:
    """Deterministic uniform [-1, 1) values: splitmix64 of (crc32(key), index).

    (A smooth sin() fill was tried first: it makes the conv stacks nearly rank deficient and the
    fp32 reference itself then sits 1e-2 away from an fp64 evaluation of its own gradients.)"""
    with np.errstate(over='ignore'):
        z = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(zlib.crc32(key.encode()))
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def closed_form_tensor(key: str, like: torch.Tensor) -> torch.Tensor:
    n, shape = like.numel(), tuple(like.shape)
    leaf = key.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=like.dtype)
    w = _wave(n, key)
    if leaf == 'running_var':
        v = 1.0 + 0.3 * w
    elif leaf == 'running_mean':
        v = 0.1 * w
    elif leaf == 'weight' and len(shape) == 1 and n == 1:      # nn.PReLU()
        v = np.full(n, 0.25)
    elif leaf == 'weight' and len(shape) == 1:                  # BatchNorm gamma
        v = 1.0 + 0.1 * w
    elif leaf == 'bias':
        v = 0.05 * w
    elif leaf == 'weight':                                      # conv OIHW / linear [out][in]
        fan_in = int(np.prod(shape[1:]))
        v = np.sqrt(6.0 / fan_in) * w
    else:
        v = 0.1 * w
    return torch.from_numpy(v.astype(np.float32)).reshape(shape).to(like.dtype)


def closed_form_state(template: Dict[str, torch.Tensor], prefix: str = '') -> Dict[str, torch.Tensor]:
    """A state_dict with the same keys/shapes as ``template`` and closed-form values."""
    return {k: closed_form_tensor(prefix + k, v) for k, v in template.items()}


# Multi-step train fixtures.  With the plain closed-form fill one Adam step saturates the discriminator (every one
# of its 23.6 M weights moves by lr = 1e-4 in the direction that helps: disc-loss 1.39 -> 4e-5 -> 1e-6) and later
# steps then compare two fp32 evaluations of log(1e-6); ESRGAN's un-damped dense blocks blow its output up to
# |pixel| ~ 100.  These per-key factors keep the GAN game at O(1) losses for several steps (SRGAN disc-loss
# 1.38 -> 1.21 -> 1.05, ESRGAN 0.70 -> 0.63 -> 0.56), so steps 1.. can be held to the same tolerance as step 0:
# the last classifier layer is scaled down (logit sensitivity to a 1e-4 move of the layers below), the dense-block
# convs get the x0.1 the reference's own initialiser gives them (torchsr/esrgan/residual.py:58-63) and ESRGAN's
# output conv is scaled so that the image is O(1).
STEP_SCALES = {
    'srgan.D': (('classifier.2.weight', 0.01),),
    'esrgan.D': (('classifier.2.weight', 0.01),),
    'srgan.G': (),
    'esrgan.G': (('conv4.weight', 0.01), ('RDB', 0.1)),
}


def step_state(template: Dict[str, torch.Tensor], kind: str) -> Dict[str, torch.Tensor]:
    """``closed_form_state`` with the conditioning of ``STEP_SCALES[kind]`` ('srgan.G', 'srgan.D', 'esrgan.G',
    'esrgan.D'): the weights the train-step fixtures and tests start from."""
    sd = closed_form_state(template)
    for pat, factor in STEP_SCALES[kind]:
        for k in sd:
            if k.endswith('weight') and sd[k].dim() > 1 and pat in k:
                sd[k] = sd[k] * factor
    return sd


def seeded_input(shape, seed: int) -> torch.Tensor:
    """Uniform [0,1) fp32 from a numpy PCG64 stream (stable across torch versions)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.random(shape, dtype=np.float32))


def tensor_digest(t: torch.Tensor):
    """Small order-sensitive summary used for gradients / post-step parameters."""
    f = t.detach().double().flatten()
    idx = torch.arange(1, f.numel() + 1, dtype=torch.float64)
    return np.array([f.sum().item(), f.abs().sum().item(), (f * torch.cos(idx * 0.001)).sum().item(),
                     f[0].item(), f[-1].item()], dtype=np.float64)


SAMPLE_MIN_NUMEL = 4096  # tensors at least this large are pinned by a strided SAMPLE of their elements (a digest is a sum: its
SAMPLE_COUNT = 512       # tolerance would have to grow with the element count and ends up pinning nothing on a 19 M-element tensor)


def sample_indices(numel: int) -> torch.Tensor:
    """The same ``SAMPLE_COUNT`` (or fewer) element indices for the fixture generator and the tests: an even stride over the
    flattened tensor, offset so that neither the first nor the last element is special."""
    stride = max(1, numel // SAMPLE_COUNT)
    return torch.arange(stride // 2, numel, stride)[:SAMPLE_COUNT]


def sample_keys(sd) -> list:
    """Sorted keys of the floating-point entries of a state dict that are pinned by samples."""
    return [k for k, v in sorted(sd.items()) if v.is_floating_point() and v.numel() >= SAMPLE_MIN_NUMEL]


def sample_table(sd) -> np.ndarray:
    """[len(sample_keys(sd))][SAMPLE_COUNT] float32: the sampled elements of every large tensor of a state dict."""
    rows = []
    for k in sample_keys(sd):
        f = sd[k].detach().float().cpu().flatten()
        row = f[sample_indices(f.numel())]
        rows.append(torch.nn.functional.pad(row, (0, SAMPLE_COUNT - row.numel())).numpy())
    return np.stack(rows).astype(np.float32)
