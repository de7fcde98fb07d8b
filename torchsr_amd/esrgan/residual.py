"""ESRGAN building blocks -- same classes / parameters as torchsr/esrgan/residual.py:17-129."""
from torch import nn, Tensor

from .. import functional as F
from ..layers import ACT_LRELU, Conv2d, Marker


class ResidualDenseBlock(nn.Module):
    """Five dense 3x3 convs (torchsr/esrgan/residual.py:31-86).

    ``convK`` sees ``cat(x, conv1..convK-1)``; LeakyReLU(0.2) is fused into the first four conv
    epilogues; the output is ``conv5 * scale_ratio + x``.  The concatenations are not materialised:
    every conv reads a prefix of one shared 192-channel buffer and writes its outputs behind it
    (``functional._DenseBlock``).
    Initialisation reproduces the reference: kaiming_normal_ * 0.1, zero bias (:58-63).
    """

    def __init__(self, channels: int = 64, growth_channels: int = 32, scale_ratio: float = 0.2):
        super().__init__()

        def block(i: int) -> nn.Sequential:
            return nn.Sequential(
                Conv2d(channels + i * growth_channels, growth_channels, kernel_size=3, stride=1, padding=1,
                       act=ACT_LRELU, slope=0.2),
                Marker('LeakyReLU(0.2) (conv epilogue)'))

        self.conv1, self.conv2, self.conv3, self.conv4 = block(0), block(1), block(2), block(3)
        self.conv5 = Conv2d(channels + 4 * growth_channels, channels, kernel_size=3, stride=1, padding=1)
        self.scale_ratio = scale_ratio
        for module in self.modules():
            if isinstance(module, nn.Conv2d):
                nn.init.kaiming_normal_(module.weight)
                module.weight.data *= 0.1
                if module.bias is not None:
                    module.bias.data.zero_()

    def forward(self, x: Tensor) -> Tensor:
        # one autograd node, one shared 192-channel buffer instead of the four torch.cat copies (:81-86)
        return F.dense_block(x, self.scale_ratio,
                             (self.conv1[0], self.conv2[0], self.conv3[0], self.conv4[0], self.conv5))


class ResidualInResidualDenseBlock(nn.Module):
    """Three RDBs, ``out * 0.2 + x`` (torchsr/esrgan/residual.py:100-129)."""

    def __init__(self, channels: int = 64, growth_channels: int = 32, scale_ratio: float = 0.2):
        super().__init__()
        self.RDB1 = ResidualDenseBlock(channels, growth_channels, scale_ratio)
        self.RDB2 = ResidualDenseBlock(channels, growth_channels, scale_ratio)
        self.RDB3 = ResidualDenseBlock(channels, growth_channels, scale_ratio)

    def forward(self, x: Tensor) -> Tensor:
        out = self.RDB3(self.RDB2(self.RDB1(x)))
        return F.axpby(out, x, 0.2, 1.0)
