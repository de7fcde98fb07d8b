"""SRGAN generator (SRResNet) -- interface of torchsr/srgan/generator.py:33-81."""
import math

import torch
from torch import nn, Tensor

from .. import _dev
from .. import functional as F
from ..layers import BatchNorm2d, Conv2d, PReLU
from .residual import ResidualBlock, SubpixelConvolutionLayer

NUM_RESIDUAL = 16  # torchsr/srgan/generator.py:20


class Generator(nn.Module):
    """``Generator(scale_factor=4)``; ``forward([N,3,h,w]) -> [N,3,s*h,s*w]`` (unclamped).

    Same submodule names and ``state_dict`` schema as the reference; internally
    NHWC fp32 on hand-written gfx950 kernels.
    """

    # low-resolution pixels a tile must see beyond its border for tiled inference to equal the untiled result
    # (receptive field radius: 4 + 16 * 2 + 1 + 1 + 1/2 + 1 = 39.5)
    halo = 48

    def __init__(self, scale_factor: int = 4) -> None:
        super().__init__()
        num_conv_layers = int(math.log(scale_factor, 2))
        self.conv1 = nn.Sequential(Conv2d(3, 64, kernel_size=9, stride=1, padding=4), PReLU())
        self.blocks = nn.Sequential(*[ResidualBlock(channels=64) for _ in range(NUM_RESIDUAL)])
        self.conv2 = nn.Sequential(Conv2d(64, 64, kernel_size=3, stride=1, padding=1, bias=False), BatchNorm2d(64))
        self.conv_layers = nn.Sequential(*[SubpixelConvolutionLayer(64) for _ in range(num_conv_layers)])
        self.conv3 = Conv2d(64, 3, kernel_size=9, stride=1, padding=4)

    # feature-map pixels (input of the last sub-pixel layer) a head tile must see beyond its border: 1 for that layer's
    # 3x3 conv + 4 output pixels of conv3's 9x9 window = 2 feature pixels
    head_halo = 4

    def infer_trunk_nhwc(self, x4: Tensor) -> Tensor:
        """Inference, first stage: NHWC ``[N,h,w,4]`` -> the 64-channel feature map that enters the LAST sub-pixel layer
        (``[N,s*h/2,s*w/2,64]``): conv1, the residual tower, conv2 + skip and all sub-pixel layers but the last.  33 of the
        36 convs; their receptive field (radius 38 pixels) is why ``test.upscale`` runs this stage on the whole image."""
        f = self.__dict__.get('_folded')
        if f is None:
            f = self.__dict__.setdefault('_folded', (F.FoldedConv(self.conv1[0], None, self.conv1[1]),
                                                     F.FoldedConv(self.conv2[0], self.conv2[1], None)))
        conv1 = f[0](x4)
        if self.bf16_native():  # from here to the output conv the activations are STORED as bf16 (csrc/c64.hip)
            conv1 = F.to_bf16(conv1)
        out = f[1](self.blocks(conv1), residual=conv1)
        for layer in list(self.conv_layers)[:-1]:
            out = layer(out)
        return out

    def infer_head_nhwc(self, feat: Tensor) -> Tensor:
        """Inference, second stage: feature map (or a tile of it with ``head_halo`` pixels around) -> NHWC image."""
        out = self.conv_layers[-1](feat)
        if out.dtype == torch.bfloat16:
            return F.conv2d_bf16in(self.conv3, out)
        return self.conv3(out)

    def bf16_native(self) -> bool:
        """Inference with bf16 products (``test.upscale(precision='bf16')``): the 64-channel layers keep their activations
        in bf16 between kernels (``srx_conv3x3_c64_bf16_fwd``) -- half the HBM traffic, no conversion on the way into LDS."""
        return self.conv2[0]._st.precision == 1 and not _dev.NO_C64 and F.inference_mode(self)

    def forward_nhwc(self, x4: Tensor) -> Tensor:
        """NHWC ``[N,h,w,4]`` -> NHWC ``[N,s*h,s*w,4]`` (4th channel zero)."""
        if F.inference_mode(self):
            return self.infer_head_nhwc(self.infer_trunk_nhwc(x4))
        conv1 = self.conv1[1](self.conv1[0](x4))
        if len(self.blocks) and all(F.residual_block_fused_ok(b) for b in self.blocks):
            block = F.residual_tower(conv1, self.blocks)  # training under a trainer: the whole chain is one autograd node
        else:
            block = self.blocks(conv1)
        bn = self.conv2[1]
        y, part = self.conv2[0](block, want_stats=True) if bn.training else (self.conv2[0](block), None)
        out = bn(y, part, residual=conv1)  # torch.add(conv1, conv2), generator.py:78
        out = F.cut_point('g.tail', out)   # data parallel: conv_layers.* / conv3.* gradients go out first
        out = self.conv_layers(out)
        return self.conv3(out)

    def forward(self, x: Tensor) -> Tensor:
        return F.to_nchw(self.forward_nhwc(F.to_nhwc(x, 4)), 3)
