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

    # a bucket at least this large opens a "communication window": from its launch() to the next wait() the launch plans
    # leave ``reserve_cus`` compute units to RCCL's channel workgroups (srx_set_reserved_cus) -- the generator's 1.2 / 4.9 MB
    # buckets are on the wire for tens of microseconds and do not
    window_bytes = 8 << 20

    def __init__(self, flat, boundaries: Sequence[str] = (), module: Optional[torch.nn.Module] = None,
                 group: Optional[dist.ProcessGroup] = None, force: bool = False, reserve_cus: int = 0, window_hook=None):
        self.reserve_cus = int(reserve_cus)
        self.window_hook = window_hook  # rehearsal aid: .begin() / .end() around a window (bench.py holds CUs there)
        self._window = False
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
        # measurement aid (bench.py --gpus N): with `timing` on, every bucket gets three HIP events on the compute stream -- at its
        # launch(), in front of its wait() and behind it -- so that a first real N > 1 run says by itself how long each bucket had
        # to hide (launch -> wait) and how long the compute stream stood still for it (the exposed part)
        self.timing = False
        self._marks = []
        self._launched = []

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
            if not self._window and s.numel() * 4 >= self.window_bytes and (self.reserve_cus > 0 or self.window_hook is not None):
                # what is launched (or captured) from here to wait() runs next to RCCL's channel kernels
                self._window = True
                self._base_reserved = reserved_cus()  # (a reservation made for the whole run, SRX_RESERVED_CUS, stays)
                set_reserved_cus(max(self.reserve_cus, self._base_reserved))
                if self.window_hook is not None:
                    self.window_hook.begin()
            ev = None
            if self.timing:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
            self._launched.append((int(s.numel()) * 4, ev))
            self._work.append(dist.all_reduce(s, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.issued += 1

    def wait(self) -> None:
        for w, (nbytes, ev0) in zip(self._work, self._launched):
            if ev0 is not None:
                ev1, ev2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev1.record()
                w.wait()  # stream-orders the compute stream after the collective (no host block on RCCL)
                ev2.record()
                self._marks.append((nbytes, ev0, ev1, ev2))
            else:
                w.wait()
        self._work, self._launched = [], []
        self._close_window()

    def _close_window(self) -> None:
        if self._window:
            self._window = False
            if self.window_hook is not None:
                self.window_hook.end()
            set_reserved_cus(self._base_reserved)

    def abort(self) -> None:
        """A step raised between ``launch()`` and ``wait()``: drop the pending work handles and give the reserved compute units
        (and the rehearsal hook's holder) back, so that no later plan or captured graph is cut for a chip with fewer CUs."""
        for w in self._work:
            try:
                w.wait()
            except Exception:  # noqa: BLE001  (the collective itself may be what failed)
                pass
        self._work, self._launched = [], []
        self._close_window()

    def timings(self, reset: bool = True) -> list:
        """Per bucket launch since the last call (``timing`` on): bytes, device ms from its launch() to the front of its wait()
        (the compute it had to hide under) and device ms the compute stream stood in that wait() (the exposed part)."""
        torch.cuda.synchronize()
        out = [{'bytes': nb, 'launch_to_wait_ms': round(e0.elapsed_time(e1), 4), 'exposed_wait_ms': round(e1.elapsed_time(e2), 4)}
               for nb, e0, e1, e2 in self._marks]
        if reset:
            self._marks = []
        return out


def reserved_cus() -> int:
    """Compute units the launch plans currently leave free (device CUs less ``srx_plan_cus``)."""
    from . import _lib
    return max(0, int(_lib.lib().srx_device_cus()) - int(_lib.lib().srx_plan_cus()))


def set_reserved_cus(k: int) -> None:
    """Compute units the launch plans made from now on leave to someone else (``srx_set_reserved_cus``)."""
    from . import _lib
    _lib.call('srx_set_reserved_cus', int(k))


# RCCL channels (= channel workgroups per collective kernel) of a data-parallel run, and the compute units the launch plans
# leave to them while a large gradient bucket is on the wire.  The exchange has ~3.5 ms of compute to hide under (94 MB of
# discriminator gradients go out before the VGG19 forward, the optimiser that needs them comes after it): 1.75 x 94 MB per
# GPU in 3.5 ms is ~50 GB/s, a fraction of one xGMI link pair -- a few channels carry it, and every channel is a
# workgroup that shares a CU with the convolutions.  RCCL's default (tens of channels) is sized for bandwidth benchmarks.
DEFAULT_CHANNELS = 8


def configure_comm(world_size: int, backend: str = 'nccl') -> dict:
    """Call BEFORE ``init_process_group`` of a data-parallel run.  At world size > 1 on RCCL: bounds the channel count
    (``NCCL_MIN_NCHANNELS`` / ``NCCL_MAX_NCHANNELS``; values already in the environment win, ``SRX_NCCL_CHANNELS`` sets both)
    and returns what the trainers pass to ``GradBuckets(reserve_cus=...)`` (``SRX_COMM_RESERVED_CUS`` overrides; default one
    CU per channel).  Nothing is left to chance silently: the returned dict is printed by ``bench.py`` and the CLI."""
    import os
    out = {'world_size': int(world_size), 'backend': backend, 'nccl_min_nchannels': os.environ.get('NCCL_MIN_NCHANNELS'),
           'nccl_max_nchannels': os.environ.get('NCCL_MAX_NCHANNELS'), 'reserved_cus_in_comm_window': 0,
           'window_bytes': GradBuckets.window_bytes}
    if world_size <= 1 or backend != 'nccl':
        return out
    k = int(os.environ.get('SRX_NCCL_CHANNELS', str(DEFAULT_CHANNELS)))
    os.environ.setdefault('NCCL_MIN_NCHANNELS', str(k))
    os.environ.setdefault('NCCL_MAX_NCHANNELS', str(k))
    out['nccl_min_nchannels'] = os.environ['NCCL_MIN_NCHANNELS']
    out['nccl_max_nchannels'] = os.environ['NCCL_MAX_NCHANNELS']
    out['reserved_cus_in_comm_window'] = int(os.environ.get('SRX_COMM_RESERVED_CUS', os.environ['NCCL_MAX_NCHANNELS']))
    return out


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
