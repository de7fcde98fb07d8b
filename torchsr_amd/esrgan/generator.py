"""ESRGAN generator placeholder (filled in below in this round)."""
from torch import nn


class Generator(nn.Module):
    def __init__(self, num_rrdb_blocks: int = 23) -> None:
        super().__init__()
        raise NotImplementedError('ESRGAN generator: in progress')
