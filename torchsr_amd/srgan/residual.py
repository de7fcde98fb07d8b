"""SRGAN building blocks -- same classes / parameters as torchsr/srgan/residual.py:16-92."""
from torch import nn, Tensor

from .. import functional as F
from .._lib import ACT_PRELU
from ..layers import BatchNorm2d, Conv2d, Marker, PReLU


class SubpixelConvolutionLayer(nn.Module):
    """conv3x3(C -> 4C, bias) -> PixelShuffle(2) -> PReLU (torchsr/srgan/residual.py:25-48).

    The pixel shuffle is folded into the convolution's store (the GEMM's output
    columns are ordered (i, j, c) so a wave writes 128 contiguous bytes per row of
    the shuffled image); PReLU runs as one streaming pass that keeps the
    pre-activation for the backward.
    """

    def __init__(self, channels: int = 64) -> None:
        super().__init__()
        self.conv = Conv2d(channels, channels * 4, kernel_size=3, stride=1, padding=1, shuffle=2)
        self.pixel_shuffle = Marker('PixelShuffle(2) (in the conv store)')
        self.prelu = PReLU()

    def forward(self, x: Tensor) -> Tensor:
        if F.inference_mode(self):  # conv + PixelShuffle + PReLU in one kernel
            f = self.__dict__.get('_folded') or self.__dict__.setdefault('_folded', F.FoldedConv(self.conv, None, self.prelu))
            return f(x)
        out = self.conv(x)
        return self.prelu(out)


class ResidualBlock(nn.Module):
    """x + BN2(conv2(PReLU(BN1(conv1(x))))) (torchsr/srgan/residual.py:61-92).

    conv -> per-channel (sum, sum^2) partials in the conv epilogue -> finalize ->
    one fused normalise+PReLU (or normalise+residual add) pass.
    """

    def __init__(self, channels: int = 64) -> None:
        super().__init__()
        self.conv1 = Conv2d(channels, channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn1 = BatchNorm2d(channels)
        self.prelu = PReLU()
        self.conv2 = Conv2d(channels, channels, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = BatchNorm2d(channels)

    def forward(self, x: Tensor) -> Tensor:
        if F.inference_mode(self):  # two kernels: BatchNorm folded, PReLU / skip connection in the epilogues
            f = self.__dict__.get('_folded')
            if f is None:
                f = self.__dict__.setdefault('_folded', (F.FoldedConv(self.conv1, self.bn1, self.prelu),
                                                         F.FoldedConv(self.conv2, self.bn2, None)))
            return f[1](f[0](x), residual=x)
        if F.residual_block_fused_ok(self):  # training under a trainer: one autograd node (skip gradient fused into conv1's dgrad)
            return F.residual_block(x, self)
        y, part = self.conv1(x, want_stats=True) if self.bn1.training else (self.conv1(x), None)
        out = self.bn1(y, part, act=ACT_PRELU, prelu=self.prelu.weight)
        y, part = self.conv2(out, want_stats=True) if self.bn2.training else (self.conv2(out), None)
        return self.bn2(y, part, residual=x)
