"""Model registry -- interface of torchsr/models.py:10-82 (same dict names, same selectors,
same RuntimeError on unknown models)."""
from argparse import Namespace
from typing import Tuple

from torchsr_amd.esrgan.generator import Generator as ESRGANGen
from torchsr_amd.esrgan.trainer import ESRGANTrainer
from torchsr_amd.srgan.generator import Generator as SRGANGen
from torchsr_amd.srgan.trainer import SRGANTrainer

MODELS = {
    'esrgan': ESRGANTrainer,
    'srgan': SRGANTrainer
}

CROP_SIZE = {
    'esrgan': 128,
    'srgan': 96
}

GENERATORS = {
    'esrgan': ESRGANGen,
    'srgan': SRGANGen
}


def select_trainer_model(args: Namespace) -> Tuple[object, int]:
    """Trainer class and crop size for ``args.model`` (case-insensitive), models.py:26-53."""
    name = args.model.lower()
    if name in MODELS:
        return MODELS[name], CROP_SIZE[name]
    raise RuntimeError(f'{args.model} not supported. Please choose from: {MODELS.keys()}')


def select_test_model(args: Namespace) -> object:
    """Generator class for ``args.model``, models.py:56-82."""
    name = args.model.lower()
    if name in GENERATORS:
        return GENERATORS[name]
    raise RuntimeError(f'{args.model} not supported. Please choose from: {GENERATORS.keys()}')
