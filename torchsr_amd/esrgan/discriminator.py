"""ESRGAN discriminator -- interface of torchsr/esrgan/discriminator.py:26-95 (logits, no sigmoid)."""
from torch import nn, Tensor

from .. import functional as F
from ..layers import ACT_LRELU, BatchNorm2d, Conv2d, Linear, Marker


class Discriminator(nn.Module):
    """``Discriminator(image_size=128)``; ``forward([N,3,S,S]) -> [N,1]`` logits."""

    def __init__(self, image_size: int = 128) -> None:
        super().__init__()
        feature_map_size = image_size // 32
        layers = [Conv2d(3, 64, kernel_size=3, stride=1, padding=1, act=ACT_LRELU, slope=0.2),
                  Marker('LeakyReLU(0.2) (conv epilogue)')]
        for cin, cout, stride in [(64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1), (256, 256, 2),
                                  (256, 512, 1), (512, 512, 2), (512, 512, 1), (512, 512, 2)]:
            layers += [Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False), BatchNorm2d(cout),
                       Marker('LeakyReLU(0.2) (BN apply pass)')]
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(
            Linear(512 * feature_map_size * feature_map_size, 100),
            Marker('LeakyReLU(0.2) (linear epilogue)'),
            Linear(100, 1),
        )

    def forward_nhwc(self, x4: Tensor) -> Tensor:
        mods = list(self.features)
        out = mods[0](x4)
        i = 2
        while i < len(mods):
            conv, bn = mods[i], mods[i + 1]
            y, part = conv(out, want_stats=True) if bn.training else (conv(out), None)
            out = bn(y, part, act=ACT_LRELU, slope=0.2)
            i += 3
        out = F.cut_point('d.head', out)  # data parallel: classifier.* gradients are a bucket of their own
        out = F.flatten_nchw(out)
        out = self.classifier[0](out, act=ACT_LRELU, slope=0.2)
        return self.classifier[2](out)

    def forward(self, x: Tensor) -> Tensor:
        return self.forward_nhwc(F.to_nhwc(x, 4))
