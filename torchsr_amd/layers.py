"""Parameter-holding layers with the reference's ``state_dict`` schema.

Each class subclasses the stock ``torch.nn`` module the reference instantiates, so
parameter names, shapes, creation order (and therefore seeded default init) are
identical to roclark/torchsr, while ``forward`` runs the MI355X HIP kernels on NHWC
activations.  These layers are the internal (NHWC) building blocks; the public
``Generator`` / ``Discriminator`` / ``VGGLoss`` modules accept and return NCHW like
the reference.
"""
import contextlib

from torch import nn, Tensor

from . import functional as F
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU


_frozen = [False]


@contextlib.contextmanager
def no_weight_grad():
    """Run forwards whose parameters take no gradient while activations still do.

    Used for the discriminator pass inside the generator update (torchsr/srgan/trainer.py:456):
    the reference lets autograd compute -- and DDP all-reduce -- discriminator weight gradients
    there that are zeroed before anyone reads them.  Skipping them
    changes no result and removes 28 GFLOP and 94 MB of all-reduce per step.
    """
    old = _frozen[0]
    _frozen[0] = True
    try:
        yield
    finally:
        _frozen[0] = old


def _w(p):
    return p.detach() if (_frozen[0] and p is not None) else p


class Conv2d(nn.Conv2d):
    """nn.Conv2d on NHWC activations with an optional fused epilogue.

    ``act``: fused ReLU / LeakyReLU after the bias; ``shuffle=2`` fuses the
    ``nn.PixelShuffle(2)`` that follows the conv in srgan/residual.py:27-28; ``up=2`` fuses the
    ``F.interpolate(scale_factor=2, mode='nearest')`` that precedes it in esrgan/generator.py:73-78.
    """

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, act=ACT_NONE,
                 slope=0.0, shuffle=0, up=0):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                         bias=bias)
        self._st = F.ConvState(in_channels, out_channels, kernel_size, stride, padding, shuffle=shuffle, act=act,
                               slope=slope, up=up)

    def forward(self, x: Tensor, want_stats: bool = False, in_act=None, act_bwd_folded=None):
        """``in_act`` / ``act_bwd_folded``: a consumer / producer pair of flags that moves the backward of the producer's
        fused activation into the consumer's data gradient (``functional._Conv2d.forward``)."""
        y, part = F.conv2d(x, _w(self.weight), _w(self.bias), self._st, want_stats, self.weight, in_act, act_bwd_folded)
        return (y, part) if want_stats else y

    def repack(self) -> None:
        """Refresh the packed weight copies after an optimiser step (needs a known input shape)."""
        st = self._st
        if st._descs:
            st.pack(self.weight, next(iter(st._descs.values())), force=True)

    def __deepcopy__(self, memo):
        new = Conv2d(self.in_channels, self.out_channels, self.kernel_size[0], self.stride[0], self.padding[0],
                     self.bias is not None, self._st.act, self._st.slope, self._st.shuffle, self._st.up)
        new.load_state_dict(self.state_dict())
        return new


def set_conv_precision(module: nn.Module, precision: str) -> None:
    """``'bf16'``: every conv of ``module`` multiplies bf16-rounded operands and accumulates in fp32 (forward
    and stride-1 data gradient) -- this package's reading of the reference's ``autocast`` region
    (srgan/trainer.py:379-383).  ``'fp32'`` restores exact fp32."""
    p = {'fp32': 0, 'bf16': 1}[precision]
    for m in module.modules():
        if isinstance(m, Conv2d):
            m._st.precision = p


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameter holder; applied through ``functional.bn_act``."""

    def forward(self, y: Tensor, part=None, act=ACT_NONE, slope=0.0, prelu=None, residual=None, groups: int = 1) -> Tensor:
        return F.bn_act(y, part, self, act=act, slope=slope, prelu=_w(prelu), residual=residual, frozen=_frozen[0],
                        groups=groups)


class PReLU(nn.PReLU):
    def forward(self, x: Tensor) -> Tensor:
        return F.prelu(x, _w(self.weight))


class LeakyReLU(nn.LeakyReLU):
    def forward(self, x: Tensor) -> Tensor:
        return F.leaky_relu(x, self.negative_slope)


class Linear(nn.Linear):
    def forward(self, x: Tensor, act=ACT_NONE, slope=0.0) -> Tensor:
        return F.linear(x, _w(self.weight), _w(self.bias), act, slope)


class Marker(nn.Module):
    """Placeholder keeping ``nn.Sequential`` indices equal to the reference's when the op it
    stands for (ReLU / LeakyReLU / Sigmoid / PixelShuffle) is fused into a neighbouring kernel."""

    def __init__(self, what: str):
        super().__init__()
        self.what = what

    def extra_repr(self) -> str:
        return f'fused {self.what}'

    def forward(self, x):
        return x


def repack_module(module: nn.Module) -> None:
    for m in module.modules():
        if isinstance(m, Conv2d):
            m.repack()


__all__ = ['no_weight_grad', 'Conv2d', 'BatchNorm2d', 'PReLU', 'LeakyReLU', 'Linear', 'Marker', 'repack_module', 'ACT_NONE',
           'ACT_RELU', 'ACT_LRELU']
