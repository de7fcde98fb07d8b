"""Single-image 4x upscaling -- the ``torchsr test <image>`` path (torchsr/test.py:22-63).

The reference's function cannot run as shipped (it iterates the outer checkpoint dict and leaves
``name`` unbound for keys without the ``module.`` prefix).  This is the intended
behaviour: load ``{model}-gan-best.pth`` (``{"epoch","phase","state"}`` or a bare state_dict, with or
without DDP's ``module.`` prefix), run the generator in eval mode without autograd, write
``upres-<image>``.

Large images (BASELINE config 5: 1080p -> 8K) are processed in spatial tiles with a halo: the
generator is fully convolutional and, in eval mode, BatchNorm is a per-channel affine map, so a tile
whose halo covers the receptive field reproduces the untiled result exactly while every conv call
stays below the kernels' 2^24-pixel / 2^31-element addressing limits.  That holds for SRGAN, whose
receptive field radius is < 40 low-resolution pixels (``Generator.halo`` = 48).  RRDBNet's is ~350 pixels
(69 dense blocks of five 3x3 convs), more than a tile can carry: ESRGAN tiles use a 64-pixel halo and
are an approximation near tile borders (the dense blocks' 0.2 residual scaling makes far pixels count
little); pass ``halo=`` / ``max_tile_pixels=`` to trade time for accuracy.
"""
import os
from argparse import Namespace
from collections import OrderedDict

import torch
from torch import Tensor

HALO = 48            # LR pixels: default when the generator does not name its own (``Generator.halo``)
MAX_TILE_PIXELS = 600 * 1000  # LR pixels per tile (x16 HR pixels must stay < 2^24)
TRUNK_MAX_PIXELS = 8 * 1000 * 1000  # LR pixels the staged path runs untiled (x256 floats of the sub-pixel conv < 2^31)


def load_generator_state(path: str) -> OrderedDict:
    ckpt = torch.load(path, map_location='cpu')
    state = ckpt['state'] if isinstance(ckpt, dict) and 'state' in ckpt else ckpt
    return OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in state.items())


@torch.no_grad()
def upscale(generator: torch.nn.Module, low_res: Tensor, halo: int = None,
            max_tile_pixels: int = MAX_TILE_PIXELS, scale: int = 4, precision: str = None, staged: bool = True) -> Tensor:
    """``generator(low_res)`` in eval mode, tiled when the image is large.  ``low_res``: [N,3,h,w].

    ``precision``: ``'fp32'`` (exact; the reference's ``test`` runs no autocast, test.py:57-62) or ``'bf16'`` (bf16
    products with fp32 accumulation in every conv but the 3-channel INPUT conv); ``None`` keeps
    whatever the generator's convs are set to.  The setting is restored afterwards.  ``staged=False`` forces the halo
    tiling for a generator that offers the two-stage interface (``_upscale_staged``)."""
    from .layers import Conv2d, set_conv_precision
    generator.eval()
    if precision is not None:
        if precision not in ('fp32', 'bf16'):
            raise ValueError(f"upscale: precision must be 'fp32' or 'bf16', got {precision!r}")
        saved = [(m, m._st.precision) for m in generator.modules() if isinstance(m, Conv2d)]
        set_conv_precision(generator, precision)
        if precision == 'bf16':  # inference: the 64 -> 3 output conv multiplies bf16 operands too (srx_conv2d_t::precision = 2)
            for m, _ in saved:
                if m.out_channels <= 4 and m.in_channels == 64:
                    m._st.precision = 2
        try:
            return upscale(generator, low_res, halo, max_tile_pixels, scale, None, staged)
        finally:
            for m, p in saved:
                m._st.precision = p
    if halo is None:
        halo = int(getattr(generator, 'halo', HALO))
    n, c, h, w = low_res.shape
    if n * h * w <= max_tile_pixels:
        return generator(low_res)
    if staged and hasattr(generator, 'infer_trunk_nhwc') and n * h * w <= TRUNK_MAX_PIXELS:
        return _upscale_staged(generator, low_res, max_tile_pixels * scale * scale, scale)
    rows = max(1, -(-h * w * n // max_tile_pixels))
    th = -(-h // int(rows ** 0.5 + 0.999))
    tw = max(1, max_tile_pixels // (n * (th + 2 * halo))) - 2 * halo
    tw = max(64, min(tw, w))
    out = torch.empty((n, c, h * scale, w * scale), dtype=low_res.dtype, device=low_res.device)
    for y0 in range(0, h, th):
        for x0 in range(0, w, tw):
            y1, x1 = min(h, y0 + th), min(w, x0 + tw)
            ya, xa = max(0, y0 - halo), max(0, x0 - halo)
            yb, xb = min(h, y1 + halo), min(w, x1 + halo)
            sr = generator(low_res[:, :, ya:yb, xa:xb].contiguous())
            out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
                sr[:, :, (y0 - ya) * scale:(y1 - ya) * scale, (x0 - xa) * scale:(x1 - xa) * scale]
    return out


def _upscale_staged(generator: torch.nn.Module, low_res: Tensor, max_out_pixels: int, scale: int) -> Tensor:
    """Large image, generator with a two-stage inference interface (SRGAN): the low-resolution trunk -- 33 of the 36
    convs, receptive field radius 38 pixels -- runs ONCE on the whole image (its largest tensor, h*w*256 floats, stays
    below the kernels' 2^31-element limit up to ``TRUNK_MAX_PIXELS``), and only the last sub-pixel layer + conv3, whose
    tensors are 16x larger, run on row strips of the trunk's feature map with ``head_halo`` rows around.  Exact like the
    halo tiling below, without recomputing the trunk on every tile's halo (1080p: 20 % of the work)."""
    from . import functional as F
    n, c, h, w = low_res.shape
    feat = generator.infer_trunk_nhwc(F.to_nhwc(low_res, 4))
    fh, fw = feat.shape[1], feat.shape[2]
    up = h * scale // fh  # output pixels per feature pixel
    hh = int(generator.head_halo)
    out = torch.empty((n, c, h * scale, w * scale), dtype=low_res.dtype, device=low_res.device)
    rows = max(8, max_out_pixels // (n * fw * up * up) - 2 * hh)  # feature rows per strip
    rows = -(-fh // -(-fh // rows))  # equal strips
    for y0 in range(0, fh, rows):
        y1 = min(fh, y0 + rows)
        ya, yb = max(0, y0 - hh), min(fh, y1 + hh)
        sr = F.to_nchw(generator.infer_head_nhwc(feat[:, ya:yb].contiguous()), c)
        out[:, :, y0 * up:y1 * up] = sr[:, :, (y0 - ya) * up:(y1 - ya) * up]
    return out


def test(args: Namespace, model: object, device) -> None:
    """``test(args, GeneratorClass, device)`` as called from the CLI (torchsr/torchsr.py:253-255)."""
    import numpy as np
    from PIL import Image
    from .srgan.trainer import save_image
    generator = model().to(device)
    generator.load_state_dict(load_generator_state(f'{args.model.lower()}-gan-best.pth'))
    image = np.asarray(Image.open(args.image).convert('RGB'), dtype='float32') / 255.0
    low_res = torch.from_numpy(image).permute(2, 0, 1).unsqueeze(0).contiguous().to(device)
    super_res = upscale(generator, low_res, precision=getattr(args, 'precision', None) or 'fp32')
    head, tail = os.path.split(args.image)
    save_image(super_res, os.path.join(head, f'upres-{tail}'))
