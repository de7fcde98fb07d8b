"""torchsr_amd -- MI355X-native SRGAN / ESRGAN training and inference hot path.

Drop-in for the hot path of roclark/torchsr (``torchsr.srgan`` / ``torchsr.esrgan``
Generator + Discriminator + VGGLoss + trainers) on AMD Instinct MI355X (gfx950):
a Python host on PyTorch-ROCm calling hand-written HIP kernels through a C ABI
(``include/srx.h``).  There is no CPU fallback; see ``oracle/`` for the CPU
restatement used only by the tests.
"""
from .__version__ import VERSION as __version__  # noqa: F401
