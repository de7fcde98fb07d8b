"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference wraps both networks in ``DistributedDataParallel`` (torchsr/srgan/trainer.py:142-157)
and inherits its reducer: 25 MiB buckets in reverse parameter order, each all-reduced as soon as
autograd has produced its gradients, i.e. overlapped with the rest of the backward pass.  Here every
model's gradients already live in ONE flat fp32 buffer (``optim.FlatParams``, parameter order), so a
bucket is a contiguous slice of it -- no flatten / unflatten copies -- and the buckets are cut where
the backward pass is cut:

* discriminator: ``classifier.*`` (75.5 of the 94 MB; autograd produces it FIRST) is on the wire while
  the eight convolutions' backward still runs, the 19 MB convolution slice follows, and both ride
  under the VGG19 perceptual-loss forward before the discriminator's Adam needs them;
* generator: the sub-pixel tail (``conv_layers.*``, ``conv3.*``) goes out while the residual tower's
  backward runs, the tower's slice at the end.

The backward pass is paused at a bucket boundary with ``BackwardCuts``: the activation at the cut is
detached in the forward pass, ``backward()`` stops there, the finished slice is handed to RCCL
(``async_op=True``: the collective runs on RCCL's own stream) and ``resume()`` continues from the saved
gradient.  That keeps every collective OUTSIDE the hipGraph segments the trainer replays; nothing
depends on capturing RCCL calls into a graph.  The 1/world scaling is folded into ``srx_adam_step``.

Not issued: the reference's second, unused discriminator all-reduce and the
per-forward BatchNorm buffer broadcast (C2) -- BatchNorm statistics stay rank-local exactly as in the
reference (no SyncBatchNorm); rank 0's running statistics are the ones checkpointed.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of 94 MB is ~1.1 ms
un-overlapped, about a tenth of the step.  Works with any ``torch.distributed`` backend (``nccl`` =
RCCL on ROCm; ``gloo`` for the CPU and one-GPU rehearsal tests).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


class BackwardCuts:
    """Pause points of the backward pass (installed as ``functional.cut_hook``).

    ``functional.cut_point(name, t)`` in a module's forward returns ``t`` unchanged unless ``name`` is
    armed here; then the graph is cut: the module continues on a detached leaf, ``loss.backward()``
    stops at that leaf, and ``resume(name)`` later back-propagates the leaf's gradient through the
    part of the graph below the cut.  A name may be hit several times per step (the discriminator
    runs on the real and on the fake batch): all hits are resumed together.
    """

    def __init__(self, names: Sequence[str] = ()):
        self.names = set(names)
        self.pairs: Dict[str, List[Tuple[Tensor, Tensor]]] = {}

    def __call__(self, name: str, t: Tensor) -> Tensor:
        if name not in self.names or not torch.is_grad_enabled() or not t.requires_grad:
            return t
        leaf = t.detach().requires_grad_(True)
        self.pairs.setdefault(name, []).append((t, leaf))
        return leaf

    def pending(self, name: str) -> bool:
        return bool(self.pairs.get(name))

    def resume(self, name: str) -> None:
        """Continue the backward pass below cut ``name`` (no-op when nothing was cut there)."""
        pairs = [(t, leaf) for t, leaf in self.pairs.pop(name, []) if leaf.grad is not None]
        if pairs:
            torch.autograd.backward([t for t, _ in pairs], [leaf.grad for _, leaf in pairs])

    def clear(self) -> None:
        self.pairs.clear()


class GradBuckets:
    """Asynchronous SUM all-reduce of a flat gradient buffer in contiguous buckets.

    ``boundaries``: names of the parameters at which a new bucket starts (in ``FlatParams`` order);
    bucket ``i`` is the slice between boundary ``i-1`` and boundary ``i``.  ``launch(i)`` hands one
    bucket to the process group as soon as the caller knows its gradients are complete; ``wait()``
    stream-orders the compute stream behind everything launched.
    """

    def __init__(self, flat, boundaries: Sequence[str] = (), module: Optional[torch.nn.Module] = None,
                 group: Optional[dist.ProcessGroup] = None, force: bool = False):
        self.flat = flat
        self.group = group
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        # ``force``: hand the buckets to the process group at world size 1 too (a no-op sum, but the real RCCL call on
        # RCCL's stream): how the exchange is exercised on a one-GPU box (tests/test_rccl_gpu.py, bench.py)
        self.force = bool(force) and dist.is_initialized()
        cuts = [0]
        if boundaries:
            if module is None:
                raise RuntimeError('GradBuckets: boundaries are parameter names of `module`')
            offs = flat.offsets_by_name(module)
            for name in boundaries:
                if name not in offs:
                    raise RuntimeError(f'GradBuckets: no trainable parameter {name!r}')
                cuts.append(offs[name])
        cuts.append(flat.numel)
        if sorted(cuts) != cuts or len(set(cuts)) != len(cuts):
            raise RuntimeError(f'GradBuckets: boundaries must be distinct and in parameter order, got offsets {cuts}')
        self.slices = [flat.grad[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        self._work = []
        self.issued = 0  # collectives handed to the process group so far

    def __len__(self) -> int:
        return len(self.slices)

    @property
    def scale(self) -> float:
        """Multiplier that turns the summed gradient into DDP's mean."""
        return 1.0 / self.world_size

    def launch(self, i: Optional[int] = None) -> None:
        """All-reduce bucket ``i`` (all buckets when ``None``) without blocking the host."""
        if self.world_size <= 1 and not self.force:
            return
        for s in (self.slices if i is None else [self.slices[i]]):
            self._work.append(dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.issued += 1

    def wait(self) -> None:
        for w in self._work:
            w.wait()  # stream-orders the compute stream after the collective (no host block on RCCL)
        self._work = []


def GradAllReduce(flat, group: Optional[dist.ProcessGroup] = None) -> GradBuckets:
    """One bucket: the whole flat gradient buffer (the round-1 exchange; kept for small models)."""
    return GradBuckets(flat, (), None, group)


def broadcast_module(module: torch.nn.Module, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Initial parameter + buffer broadcast (what ``DDP.__init__`` does)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def describe_group() -> dict:
    """What the process group really is (printed by ``bench.py`` and the CLI)."""
    if not dist.is_initialized():
        return {'backend': None, 'world_size': 1, 'rank': 0}
    return {'backend': dist.get_backend(), 'world_size': dist.get_world_size(), 'rank': dist.get_rank()}
