"""ESRGAN discriminator -- interface of torchsr/esrgan/discriminator.py:26-95 (logits, no sigmoid)."""

import torch
from torch import nn, Tensor

from .. import _dev
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

    def features_flat_nhwc(self, x4: Tensor, groups: int = 1) -> Tensor:
        """``torch.flatten(self.features(x), 1)``.  ``groups`` > 1: the batch holds that many forward calls of the reference
        back to back (``forward_pair``); every BatchNorm then normalises each call's rows with their own statistics."""
        mods = list(self.features)
        # the first conv's LeakyReLU backward rides in the second conv's (strided) data gradient: its output feeds nothing else
        fold = torch.is_grad_enabled() and mods[0]._st.act == ACT_LRELU and not _dev.NO_ACT_FOLD  # (developer switch)
        in_act = F.ActFold(ACT_LRELU, mods[0]._st.slope) if fold else None  # one token: producer skips, consumer masks
        out = mods[0](x4, act_bwd_folded=in_act)
        i = 2
        while i < len(mods):
            conv, bn = mods[i], mods[i + 1]
            stats = bn.training
            if groups > 1 and bn.training:
                # a conv tile must not straddle two calls; where the plan's tiles do, the BatchNorm pass gathers its
                # own statistics (row blocks of srx_bn_rows_per_block) instead of taking them from the conv epilogue
                n, h, w, _ = out.shape
                stats = F.bn_groups_ok(conv._st.out_rows(n, h, w), F.conv_stat_tile_rows(conv._st, n, h, w), groups)
            y, part = conv(out, want_stats=True, in_act=in_act) if stats else (conv(out, in_act=in_act), None)
            in_act = None
            out = bn(y, part, act=ACT_LRELU, slope=0.2, groups=groups)
            i += 3
        out = F.cut_point('d.head', out)  # data parallel: classifier.* gradients are a bucket of their own
        return F.flatten_nchw(out)

    def forward_nhwc(self, x4: Tensor, groups: int = 1) -> Tensor:
        out = self.features_flat_nhwc(x4, groups)
        out = self.classifier[0](out, act=ACT_LRELU, slope=0.2)
        return self.classifier[2](out)

    # ---- the classifier and the relativistic-average loss behind it as one autograd node (functional.gan_head,
    # ---- csrc/head.hip): what the trainer calls; forward / forward_pair keep the reference's module surface (logits out)
    def pair_loss_nhwc(self, real4: Tensor, fake4: Tensor):
        """``(BCEWithLogits(D(real) - mean(D(fake)), 1) + BCEWithLogits(D(fake) - mean(D(real)), 0)) / 2``
        (esrgan/trainer.py:448-453) on NHWC inputs; ``(loss, aux)``."""
        n = real4.shape[0]
        if fake4.shape != real4.shape or (self.training and not self._pair_fits(2 * n, real4.shape[1], real4.shape[2])):
            flat = torch.cat([self.features_flat_nhwc(real4), self.features_flat_nhwc(fake4)], dim=0)
        else:
            flat = self.features_flat_nhwc(torch.cat([real4, fake4], dim=0), groups=2 if self.training else 1)
        return F.gan_head(flat, self.classifier[0], self.classifier[2], F.HEAD_ESRGAN_D, n_first=n, slope=0.2)

    def adversarial_loss_nhwc(self, fake4: Tensor, real_mean: Tensor, addend: Tensor, weight: float):
        """``addend + weight * BCEWithLogits(D(fake) - real_mean, 1)`` (esrgan/trainer.py:464-469); ``(loss, aux)``."""
        flat = self.features_flat_nhwc(fake4)
        return F.gan_head(flat, self.classifier[0], self.classifier[2], F.HEAD_ESRGAN_G, slope=0.2, adv_weight=weight,
                          shift=real_mean, addend=addend)

    def forward(self, x: Tensor) -> Tensor:
        return self.forward_nhwc(F.to_nhwc(x, 4))

    def _pair_fits(self, n2: int, h: int, w: int) -> bool:
        """Do the conv tiles and reduction row blocks of every BatchNorm layer respect the boundary between the two
        calls at this geometry?  (Pure geometry: nothing is launched, no state is touched.)"""
        cache = self.__dict__.setdefault('_pair_cache', {})
        key = (n2, h, w)
        if key not in cache:
            ok, mods, i = True, list(self.features), 2
            n2, h, w, _ = mods[0]._st.out_shape(n2, h, w)
            while ok and i < len(mods):  # (every layer can fall back to srx_bn_partial_stats: only its row blocks matter)
                n2, h, w, _ = mods[i]._st.out_shape(n2, h, w)
                ok = F.bn_groups_ok(n2 * h * w, None, 2)
                i += 3
            cache[key] = ok
        return cache[key]

    def forward_pair(self, first: Tensor, second: Tensor):
        """``(self(first), self(second))`` -- two consecutive forward calls of the reference (the real and the fake
        batch, trainer.py:446-447) executed as ONE batch of 2N: same weights, half the launches, twice the rows per
        launch, and the 75 MB classifier weight is streamed once.  Training-mode BatchNorm keeps the two calls
        apart (own batch statistics, running statistics updated for ``first`` and then for ``second``).  Falls
        back to two calls when a layer's tiles would straddle the two halves (``_pair_fits``)."""
        return self.forward_pair_nhwc(F.to_nhwc(first, 4), F.to_nhwc(second, 4))

    def forward_pair_nhwc(self, first4: Tensor, second4: Tensor):
        """``forward_pair`` on NHWC ``[N,S,S,4]`` inputs (4th channel zero), as the trainers call it."""
        n = first4.shape[0]
        if second4.shape != first4.shape or (self.training and not self._pair_fits(2 * n, first4.shape[1], first4.shape[2])):
            return self.forward_nhwc(first4), self.forward_nhwc(second4)
        x4 = torch.cat([first4, second4], dim=0)
        out = self.forward_nhwc(x4, groups=2 if self.training else 1)
        return F.split_batch(out, n)
