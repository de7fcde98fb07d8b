"""Perceptual (VGG19) loss -- interface of torchsr/srgan/loss.py:18-54.

The reference pulls ``torchvision.models.vgg19(pretrained=True).features[:36]``.
Neither torchvision nor the weight file ``vgg19-dcbb9e9d.pth`` (Dockerfile:10) ship
with this repository, so the graph is restated from the public cfg 'E' definition
and the weights are loaded from a user supplied state_dict (``weights=`` argument,
``$TORCHSR_VGG19_WEIGHTS`` or the torch hub cache).  Without a file the features are
seeded-random (kaiming-normal, as torchvision initialises VGG) -- good for
benchmarks and parity tests, not for training quality: library callers get a warning,
the ``train`` command line refuses to start unless ``--vgg-weights random`` asks for it.
"""
import os
import warnings
from typing import Optional

import torch
from torch import nn, Tensor

from .. import functional as F
from ..layers import ACT_RELU, Conv2d, Marker

# torchvision.models.vgg cfgs['E']
VGG19_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
VGG19_FILE = 'vgg19-dcbb9e9d.pth'


class MaxPool2x2(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return F.maxpool2x2(x)


def make_vgg19_features(feature_layer: int = 36) -> nn.Sequential:
    """``vgg19().features[:feature_layer]`` with the same child indices (conv at 0,2,5,...)."""
    layers, cin = [], 3
    for v in VGG19_CFG:
        if v == 'M':
            layers.append(MaxPool2x2())
        else:
            layers += [Conv2d(cin, v, kernel_size=3, stride=1, padding=1, act=ACT_RELU), Marker('ReLU (conv epilogue)')]
            cin = v
    layers = layers[:feature_layer]
    if layers and isinstance(layers[-1], Conv2d):
        raise RuntimeError('feature_layer must not cut between a conv and its ReLU')
    return nn.Sequential(*layers)


def _find_weights(path: Optional[str]) -> Optional[str]:
    cands = [path, os.environ.get('TORCHSR_VGG19_WEIGHTS'),
             os.path.join(torch.hub.get_dir(), 'checkpoints', VGG19_FILE)]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


class VGGLoss(nn.Module):
    """``VGGLoss(feature_layer=36)``; ``forward(source, target) -> 0-dim loss`` (L1 of features).

    As in the reference there is no ImageNet mean/std normalisation, the features are
    frozen and in eval mode.
    """

    def __init__(self, feature_layer: int = 36, weights: Optional[str] = None, seed: int = 1234) -> None:
        super().__init__()
        self.features = make_vgg19_features(feature_layer).eval()
        path = None if weights == 'random' else _find_weights(weights)
        if path is not None:
            state = torch.load(path, map_location='cpu')
            own = self.features.state_dict()
            picked = {k[len('features.'):]: v for k, v in state.items()
                      if k.startswith('features.') and k[len('features.'):] in own}
            self.features.load_state_dict(picked, strict=True)
            self.pretrained = True
        else:
            if weights != 'random':  # 'random' is the explicit opt-in of benchmarks, tests and `--vgg-weights random`
                warnings.warn(f'{VGG19_FILE} not found (pass weights=, or set TORCHSR_VGG19_WEIGHTS): VGGLoss uses '
                              'seeded random features', stacklevel=2)
            g = torch.Generator().manual_seed(seed)
            for m in self.features:
                if isinstance(m, Conv2d):
                    fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                    with torch.no_grad():
                        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_out) ** 0.5)
                        m.bias.zero_()
            self.pretrained = False
        for _, param in self.features.named_parameters():
            param.requires_grad = False

    def features_nhwc(self, x4: Tensor) -> Tensor:
        out = x4
        for m in self.features:
            if not isinstance(m, Marker):
                out = m(out)
        return out

    @torch.no_grad()
    def target_features(self, target: Tensor) -> Tensor:
        """``self.features(target)`` of the detached branch (loss.py:53); no autograd state."""
        return self.features_nhwc(F.to_nhwc(target, 4))

    def _stack(self):
        return [('pool', None) if isinstance(m, MaxPool2x2) else ('conv', m) for m in self.features if not isinstance(m, Marker)]

    def forward(self, source: Tensor, target: Tensor = None, target_features: Tensor = None) -> Tensor:
        """``l1_loss(features(source), features(target))`` (loss.py:52-54).  Source and target run through the frozen
        stack as ONE batch and the backward pass is the data-gradient chain of the source half with the ReLU
        backwards folded into it (``functional.frozen_conv_stack``).  ``target_features`` may carry the second term
        when the caller has already computed it."""
        with torch.no_grad():
            tgt4 = None if target is None else F.to_nhwc(target, 4)
        return self.forward_nhwc(F.to_nhwc(source, 4), tgt4, target_features)

    def forward_nhwc(self, src4: Tensor, tgt4: Tensor = None, target_features: Tensor = None) -> Tensor:
        """``forward`` on NHWC ``[N,H,W,4]`` images (4th channel zero), as the trainers call it."""
        if target_features is not None:
            fs, _ = F.frozen_conv_stack(src4, None, self._stack())
            return F.l1_loss(fs, target_features)
        fs, ft = F.frozen_conv_stack(src4, tgt4.detach(), self._stack())
        return F.l1_loss(fs, ft)
