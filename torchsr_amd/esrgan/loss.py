"""Perceptual loss of ESRGAN -- torchsr/esrgan/loss.py:18-54 is the same VGGLoss as SRGAN's."""
from ..srgan.loss import VGGLoss, make_vgg19_features  # noqa: F401
