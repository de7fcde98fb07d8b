"""ESRGAN trainer placeholder (filled in below in this round)."""
from ..srgan.trainer import SRGANTrainer


class ESRGANTrainer(SRGANTrainer):
    phase_prefix = 'esrgan'

    def __init__(self, *a, **k):
        raise NotImplementedError('ESRGAN trainer: in progress')
