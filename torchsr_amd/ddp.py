"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference wraps both networks in ``DistributedDataParallel`` (torchsr/srgan/trainer.py:142-157)
and inherits its 25 MiB bucketing.  Here every model's gradients already live in one flat fp32
buffer (``optim.FlatParams``), so the exchange is ONE all-reduce (SUM) per model, launched
asynchronously on RCCL's stream as soon as that model's backward has been enqueued and awaited
right before its optimiser step; the 1/world scaling is folded into ``srx_adam_step``.

* discriminator gradients (94 MB) are exchanged while the VGG19 perceptual-loss forward runs;
* generator gradients (6 MB) are exchanged at the end of the step;
* the reference's second, unused discriminator all-reduce and the per-forward
  BatchNorm buffer broadcast (C2) are not issued: BN statistics stay rank-local exactly as in the
  reference (no SyncBatchNorm), rank 0's running stats are the ones checkpointed.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 94 MB is ~1.1 ms
un-overlapped, about a tenth of the step; overlapped with the ~3 ms VGG forward it is hidden.
Works with any ``torch.distributed`` backend (``nccl`` = RCCL on ROCm; ``gloo`` for CPU tests).
"""
from typing import Optional

import torch
import torch.distributed as dist


class GradAllReduce:
    """Asynchronous SUM all-reduce of a flat gradient buffer."""

    def __init__(self, flat, group: Optional[dist.ProcessGroup] = None):
        self.flat = flat
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self._work = None

    @property
    def scale(self) -> float:
        """Multiplier that turns the summed gradient into DDP's mean."""
        return 1.0 / self.world_size

    def launch(self) -> None:
        if self.world_size > 1:
            self._work = dist.all_reduce(self.flat.grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self) -> None:
        if self._work is not None:
            self._work.wait()  # stream-orders the compute stream after the collective (no host block on RCCL)
            self._work = None


def broadcast_module(module: torch.nn.Module, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Initial parameter + buffer broadcast (what ``DDP.__init__`` does)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
