"""SRGAN trainer on MI355X -- interface of torchsr/srgan/trainer.py:39-543.

Same constructor, same two phases (PSNR pre-training, then GAN training), same losses,
optimiser settings, checkpoint files and per-epoch PSNR test as the reference.  What is
different is underneath:

* all device work goes through the hand-written HIP kernels (``torchsr_amd.functional``);
* the whole train step is captured once into a hipGraph and replayed (the step is ~600 kernel
  launches of a few microseconds each; eager launches would be host bound);
* data parallelism is one process per GPU with bucketed flat-buffer all-reduces on RCCL
  (``torchsr_amd.ddp``) instead of ``DistributedDataParallel``: the backward pass is paused where a
  bucket is complete, so the discriminator's 75 MB classifier gradient is on xGMI while its
  convolutions' backward still runs, and the rest rides under the VGG19 forward;
* the discriminator pass inside the generator update does not compute the discriminator weight
  gradients the reference computes there and never uses.

AMP follows the reference region by region.  SRGAN autocasts ONLY the pre-training body
(trainer.py:382-385, fp16 + GradScaler on CUDA); ``_gan_loop`` (:416-469) has no autocast and runs in
fp32 whatever ``--disable-amp`` says.  Here an autocast region selects bf16 products with fp32
accumulation in the convolutions of the networks it encloses (``srx_conv2d_t::precision``; BASELINE
config 4 asks for bf16 on MI355X, which needs no loss scaling); tensors, BatchNorm, losses, weight
gradients and Adam stay fp32, and everything outside such a region is exact fp32.  ``amp_phases`` names
the phases the reference autocasts: ``('psnr',)`` here, ``('psnr', 'gan')`` for ESRGAN
(esrgan/trainer.py:384,446,461).
"""
import os
import time
from argparse import Namespace
from math import log10
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .. import _dev
from .. import functional as F
from ..ddp import BackwardCuts, GradBuckets, broadcast_module
from ..layers import no_weight_grad, set_conv_precision
from ..optim import FlatAdam, FlatParams, StepLR
from .discriminator import Discriminator
from .generator import Generator
from .loss import VGGLoss

try:  # optional, exactly as in the reference (trainer.py:23-26)
    import wandb
except ImportError:
    wandb = None


def save_image(tensor: Tensor, path: str) -> None:
    """``torchvision.utils.save_image`` for a single image batch (trainer.py:336)."""
    from PIL import Image
    img = tensor.detach()[0].clamp(0, 1).mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to('cpu', torch.uint8)
    Image.fromarray(img.numpy()).save(path)


class LossRing:
    """Every step's train loss for wandb without a device -> host sync per step.

    The reference hands the loss TENSOR to ``wandb.log`` on every step (trainer.py:393-399,459-466), which reads it
    back -- a sync per step, ~10 % of a 9 ms step.  Here the step itself appends the scalar to a device ring
    (``srx_ring_push``, the last kernel of the replayed hipGraph), the host only notes the step id and the host-side
    values that go with it, and ``flush`` reads the ring back in ONE copy per ``cap`` steps and emits the same
    ``wandb.log`` calls -- same keys, same step axis, every sample.
    """

    def __init__(self, device, cap: int = 50):
        self.cap = cap
        # ring and counter share ONE buffer so that a flush reads both in the same copy: the counter says which slots the
        # device wrote last, whatever the host believes
        self._buf = torch.zeros(cap * 4 + 4, dtype=torch.float32, device=device)
        self.ring = self._buf[:cap * 4]
        self.counter = self._buf[cap * 4:cap * 4 + 1].view(torch.int32)
        self.read = 0        # device records accounted for (flushed or skipped)
        self.pushed = 0      # records the HOST knows were pushed (one per train step, counted where the step is issued)
        self.pending = []    # [(key, step, other contents, record index)] in push order
        self.unnoted = 0     # pushes that never got a note (direct step calls outside the training loops)

    def push(self, loss: Tensor) -> None:
        """Device side (captured into the step's hipGraph)."""
        F.call('srx_ring_push', loss.data_ptr(), None, None, None, 1, self.ring.data_ptr(), self.counter.data_ptr(),
               self.cap, torch.cuda.current_stream().cuda_stream)

    def mark_push(self) -> None:
        """Host side, once per issued step (a replayed graph pushes without any host code of ``push`` running)."""
        self.pushed += 1

    def note(self, key: str, step: int, contents: dict) -> bool:
        """Host side, right after the step whose record this is; True when the ring must be flushed before it wraps.  The
        note remembers WHICH record it belongs to (the last one issued), so a step that pushed without being noted -- a direct
        ``gan_step`` between two loop steps -- cannot shift the samples behind it onto the wrong step."""
        self.pending.append((key, step, contents, self.pushed - 1))
        return self.pushed - self.pending[0][3] >= self.cap

    def flush(self, log) -> None:
        """Deliver the pending samples: one device -> host copy (none when nothing is pending).  The device counter, read in
        the same copy as the ring, is checked against the host's count of issued pushes; a disagreement (a step that raised
        between its push and the host's mark) is reported and the mapping falls back to "the last len(pending) records"."""
        if not self.pending:
            self.unnoted += self.pushed - self.read  # (host bookkeeping only: no copy, no sync)
            self.read = self.pushed
            return
        host = self._buf.cpu()  # the one sync
        count = int(host[self.cap * 4:self.cap * 4 + 1].view(torch.int32)[0])
        n = len(self.pending)
        if count < n:
            raise RuntimeError(f'LossRing: {n} samples noted but the device pushed only {count} records')
        index = [p[3] for p in self.pending]
        if count != self.pushed:
            import warnings
            warnings.warn(f'LossRing: the device holds {count} records, the host issued {self.pushed}; mapping the {n} pending '
                          'samples to the last records', stacklevel=2)
            index = list(range(count - n, count))
            self.pushed = count
        if count - min(index) > self.cap:
            raise RuntimeError(f'LossRing: a pending sample is {count - min(index)} records old in a ring of {self.cap} slots '
                               '(flush when note() says so)')
        self.unnoted += (count - self.read) - n
        values = host[:self.cap * 4].view(self.cap, 4)[:, 0]
        for (key, step, contents, _), idx in zip(self.pending, index):
            log({**contents, key: float(values[idx % self.cap])}, step=step)
        self.read = count
        self.pending = []


class SRGANTrainer:
    """``SRGANTrainer(device, args, train_loader, test_loader, train_len, test_len, distributed)``.

    ``args`` needs the attributes the reference reads (trainer.py:67-84): ``disable_amp, batch_size,
    epochs, gan_checkpoint, local_rank, pretrain_epochs, psnr_checkpoint, skip_image_save,
    world_size, rank``.  Optional extras: ``use_graphs`` (default True), ``vgg_weights``.
    """

    phase_prefix = 'srgan'
    # single-graph GAN step: the perceptual loss's BACKWARD on the side stream too (``_gan_all``).  Exact here -- two gradients meet
    # at the generator's output and a two-term sum has one rounding whatever its order
    deep_overlap = True
    # False: the perceptual-loss branch on the main stream (same launches one after the other, same results).  bench.py's
    # instrumented eager pass sets it so that every event pair of the launch profiler brackets a kernel that has the chip alone
    overlap_branches = True
    generator_cls = Generator
    discriminator_cls = Discriminator
    amp_phases = ('psnr',)  # phases the reference wraps in amp.autocast (trainer.py:382); the GAN loop is fp32
    # data parallel: first parameter of the gradient bucket that autograd completes FIRST (see ddp.py)
    gen_tail_bucket = 'conv_layers.0.conv.weight'
    disc_head_bucket = 'classifier.0.weight'
    wandb_flush_every = 50  # steps between read-backs of the per-step train losses (LossRing): one sync per flush

    def __init__(self, device, args: Namespace, train_loader, test_loader, train_len: int, test_len: int,
                 distributed: bool = False) -> None:
        self.amp = not args.disable_amp  # trainer.py:67; what it enables is decided per phase (_enter_phase)
        self.batch_size = args.batch_size
        self.best_psnr = -1.0
        self.device = torch.device(device)
        self.distributed = distributed
        self.epochs = args.epochs
        self.gan_checkpoint = args.gan_checkpoint
        self.local_rank = args.local_rank
        self.pre_epochs = args.pretrain_epochs
        self.psnr_checkpoint = args.psnr_checkpoint
        self.save_image = not args.skip_image_save
        self.test_loader = test_loader
        self.test_len = test_len
        self.train_loader = train_loader
        self.train_len = train_len
        self.world_size = args.world_size
        self.main_process = args.rank in [-1, 0]
        self.use_graphs = bool(getattr(args, 'use_graphs', True))
        self.vgg_weights = getattr(args, 'vgg_weights', None)
        # issue the gradient all-reduces even at world size 1 (a one-GPU rehearsal of the RCCL path)
        self.force_collectives = bool(getattr(args, 'force_collectives', False))
        # compute units the launch plans leave to RCCL's channel workgroups while a large gradient bucket is on the wire
        # (ddp.configure_comm decides; 0 at world size 1) and an optional rehearsal hook around those windows
        self.comm_reserved_cus = int(getattr(args, 'comm_reserved_cus', 0) or 0)
        self.comm_window_hook = getattr(args, 'comm_window_hook', None)
        if self.device.type != 'cuda':
            raise RuntimeError('torchsr_amd trains on an MI355X (device "cuda"); there is no CPU path')
        if self.device.index is None:
            # single-process runs carry local_rank -1 (torchsr.py:148-150)
            self.device = torch.device('cuda', max(int(self.local_rank or 0), 0))
        torch.cuda.set_device(self.device)
        if self.save_image and self.main_process and not os.path.exists('output'):
            os.makedirs('output')
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self._graph_tables: Dict[str, tuple] = {}  # pack-table generations each graph was captured with
        self._eager_steps = 0                      # steps to run without graphs (after a phase switch)
        self._graph_pool = None
        self._calls: Dict[str, int] = {}
        self._static: Dict[str, Tensor] = {}
        self._phase = None
        self._cuts = BackwardCuts(('g.tail', 'd.head')) if distributed else None
        # per-step train losses for wandb: only when a run is active on this process (decided before graph capture)
        self._ring = LossRing(self.device, self.wandb_flush_every) \
            if (wandb and self.main_process and getattr(wandb, 'run', None) is not None) else None
        F.direct_grads[0] = True  # parameter gradients accumulate straight into the flat .grad views
        self._initialize_trainer()
        self._create_test_image()

    # ------------------------------------------------------------------ set-up
    def _initialize_trainer(self) -> None:
        self._initialize_models()
        self._initialize_loss()
        self._initialize_optimizers()
        self._enter_phase('gan')

    def _enter_phase(self, phase: str) -> None:
        """Select the arithmetic of ``phase`` ('psnr', 'gan' or 'test'): bf16 products inside the regions the
        reference autocasts (``amp_phases``, unless ``--disable-amp``), exact fp32 everywhere else -- the
        validation pass (trainer.py:286-304) has no autocast.  Packed weight layouts do not depend on the
        precision, so switching costs nothing; each phase has its own captured hipGraphs."""
        if phase == self._phase:
            return
        if self._graphs:
            # A replayed graph repacks, after its Adam step, exactly the weights ITS phase's kernels read (a phase whose dense
            # blocks run fused keeps their fp32 packs out of its table); the other phase's graphs would then start on stale
            # packs.  One eager step after a switch goes through the lazy per-layer check and refreshes whatever is stale.
            self._eager_steps = 1
        self._phase = phase
        precision = 'bf16' if (self.amp and phase in self.amp_phases) else 'fp32'
        for module in (self.generator, self.discriminator, self.vgg_loss):
            set_conv_precision(module, precision)

    def _initialize_models(self) -> None:
        """trainer.py:136-157.  DDP wrapping is replaced by flat buffers + explicit all-reduce."""
        self.generator = self.generator_cls().to(self.device)
        self.discriminator = self.discriminator_cls().to(self.device)
        if self.distributed:
            broadcast_module(self.generator)
            broadcast_module(self.discriminator)
        self.gen_flat = FlatParams(self.generator)
        self.disc_flat = FlatParams(self.discriminator)
        # gradient buckets in parameter order: [0] = the body (complete LAST in the backward pass), [1] = the slice
        # autograd completes first (the generator's sub-pixel tail, the discriminator's classifier)
        force = self.force_collectives
        kw = {'force': force, 'reserve_cus': self.comm_reserved_cus, 'window_hook': self.comm_window_hook}
        self.gen_sync = GradBuckets(self.gen_flat, (self.gen_tail_bucket,), self.generator, **kw) if self.distributed else None
        self.disc_sync = GradBuckets(self.disc_flat, (self.disc_head_bucket,), self.discriminator, **kw) \
            if self.distributed else None

    def _initialize_loss(self) -> None:
        """trainer.py:159-165 (MSELoss / BCELoss are kernels in torchsr_amd.functional)."""
        self.mse_loss = F.mse_loss
        self.pixel_loss = F.mse_loss  # pre-training objective (trainer.py:384)
        self.bce_loss = F.bce_loss
        self.vgg_loss = VGGLoss(weights=self.vgg_weights).to(self.device)

    def _initialize_optimizers(self) -> None:
        """trainer.py:167-196: three Adam states, two StepLR (epochs // 8, gamma 0.6)."""
        self.psnr_optimizer = FlatAdam(self.gen_flat, lr=0.0001, betas=(0.9, 0.999))
        self.disc_optimizer = FlatAdam(self.disc_flat, lr=0.0001, betas=(0.9, 0.999))
        self.gen_optimizer = FlatAdam(self.gen_flat, lr=0.0001, betas=(0.9, 0.999))
        if self.distributed:
            for opt in (self.psnr_optimizer, self.gen_optimizer):
                opt.grad_scale = self.gen_sync.scale
            self.disc_optimizer.grad_scale = self.disc_sync.scale
        from ..layers import Conv2d
        gen_table = F.PackTable(m for m in self.generator.modules() if isinstance(m, Conv2d))
        disc_table = F.PackTable(m for m in self.discriminator.modules() if isinstance(m, Conv2d))
        self.psnr_optimizer.pack_table = self.gen_optimizer.pack_table = gen_table
        self.disc_optimizer.pack_table = disc_table
        self.disc_scheduler = StepLR(self.disc_optimizer, step_size=self.epochs // 8, gamma=0.6)
        self.gen_scheduler = StepLR(self.gen_optimizer, step_size=self.epochs // 8, gamma=0.6)

    def _create_test_image(self) -> None:
        """trainer.py:128-134; falls back to a synthetic 480x320 image when the asset is absent."""
        path = 'media/waterfalls-low-res.png'
        if os.path.exists(path):
            import numpy as np
            from PIL import Image
            a = np.asarray(Image.open(path).convert('RGB'), dtype='float32') / 255.0
            image = torch.from_numpy(a).permute(2, 0, 1).contiguous()
        else:
            yy, xx = torch.meshgrid(torch.linspace(0, 1, 320), torch.linspace(0, 1, 480), indexing='ij')
            image = torch.stack([xx, yy, 0.5 + 0.5 * torch.sin(12 * xx * yy)])
        self.test_image = image.unsqueeze(0).to(self.device)

    # ------------------------------------------------------------------ plumbing
    def _log(self, statement: str) -> None:
        if self.main_process:
            print(statement)

    def _log_wandb(self, contents: dict, step: int = None) -> None:
        """trainer.py:219-231.  ``wandb.init`` is the CLI's job (torchsr.py:242-243); a trainer built by other
        code without a run simply does not log."""
        if wandb and self.main_process and getattr(wandb, 'run', None) is not None:
            wandb.log({k: (v.item() if torch.is_tensor(v) else v) for k, v in contents.items()}, step=step)

    def _cleanup(self) -> None:
        if wandb and getattr(wandb, 'run', None) is not None:
            wandb.finish()

    def _model_state(self, epoch: int, phase: str) -> dict:
        """trainer.py:233-258 writes {"epoch", "phase", "state"} with the generator only, so a resumed run
        restarts the discriminator, all three Adam states and both schedules from scratch.  The same three
        keys are written here (the reference loads these files unchanged) plus a ``resume`` entry with
        everything else a run needs to continue where it stopped."""
        cpu = lambda sd: {k: (v.detach().clone().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}  # noqa: E731
        return {'epoch': epoch, 'phase': phase, 'state': cpu(self.generator.state_dict()),
                'resume': {'discriminator': cpu(self.discriminator.state_dict()),
                           'psnr_optimizer': cpu(self.psnr_optimizer.state_dict()),
                           'gen_optimizer': cpu(self.gen_optimizer.state_dict()),
                           'disc_optimizer': cpu(self.disc_optimizer.state_dict()),
                           'gen_scheduler': self.gen_scheduler.state_dict(),
                           'disc_scheduler': self.disc_scheduler.state_dict(),
                           'best_psnr': self.best_psnr,
                           'rng': torch.get_rng_state(), 'cuda_rng': torch.cuda.get_rng_state(self.device)}}

    def _restore_resume_state(self, checkpoint: Optional[dict]) -> bool:
        """Everything beyond the generator, when the checkpoint was written by this package."""
        extra = (checkpoint or {}).get('resume')
        if not extra:
            return False
        self.discriminator.load_state_dict(extra['discriminator'])
        for name in ('psnr_optimizer', 'gen_optimizer', 'disc_optimizer', 'gen_scheduler', 'disc_scheduler'):
            getattr(self, name).load_state_dict(extra[name])
        self.best_psnr = float(extra['best_psnr'])
        torch.set_rng_state(extra['rng'])
        torch.cuda.set_rng_state(extra['cuda_rng'], self.device)
        F.bump_pack_epoch()
        return True

    def _load_checkpoint(self, path: str) -> Optional[dict]:
        """trainer.py:104-126, plus map_location and tolerance for DDP's 'module.' key prefix."""
        if path and os.path.exists(path):
            ckpt = torch.load(path, map_location='cpu')
            state = ckpt['state'] if 'state' in ckpt else ckpt
            ckpt = dict(ckpt) if 'state' in ckpt else {'epoch': 1, 'phase': '', 'state': state}
            ckpt['state'] = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state.items()}
            return ckpt
        return None

    def _exec(self, key: str, fn: Callable[[], None]) -> None:
        """Run ``fn`` eagerly (first two calls: warm-up) and from then on as a replayed hipGraph."""
        if not self.use_graphs or self._eager_steps > 0:
            fn()
            return
        g = self._graphs.get(key)
        if g is not None and self._graph_tables.get(key) != self._table_generations():
            # a pack table was rebuilt since the capture (the set of fused-only convs changed): the graph holds the freed
            # table's address and record count -- drop it, run this call eagerly and capture again at the next one
            del self._graphs[key]
            self._calls[key] = 2
            g = None
            fn()
            return
        if g is not None:
            g.replay()
            F.bump_pack_epoch()  # the replay ran Adam: packed weights seen by eager code are stale
            return
        n = self._calls.get(key, 0)
        self._calls[key] = n + 1
        if n < 2:
            fn()
            return
        if self._graph_pool is None:
            self._graph_pool = torch.cuda.graph_pool_handle()
        g = torch.cuda.CUDAGraph()
        # with a process group alive (whether or not THIS trainer exchanges gradients), RCCL's watchdog thread queries its
        # events while we capture: under a 'global' capture HIP fails that query ("operation not permitted when stream is
        # capturing") and the watchdog aborts the process; only the capturing thread's own calls are policed then
        mode = 'thread_local' if (self.distributed or dist.is_initialized()) else 'global'
        cut_state = {k: list(v) for k, v in self._cuts.pairs.items()} if self._cuts is not None else None
        try:
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode=mode):
                fn()
        except Exception as exc:  # keep training: run this and all later steps eagerly
            self._log(f'hipGraph capture of {key} failed ({type(exc).__name__}: {exc}); continuing without graphs')
            self.use_graphs = False
            self._graphs.clear()
            torch.cuda.synchronize()
            F.bump_pack_epoch()  # packs "refreshed" during the failed capture never ran
            if cut_state is not None:
                self._cuts.pairs = cut_state  # what the failed capture cut or resumed never ran either
            fn()
            return
        self._graphs[key] = g
        self._graph_tables[key] = self._table_generations()
        g.replay()  # capture records without executing; run this step's work now

    def _table_generations(self):
        """(re)build counters of the pack tables a captured step replays (functional.PackTable.generation)"""
        return tuple(getattr(opt, 'pack_table', None).generation if getattr(opt, 'pack_table', None) is not None else -1
                     for opt in (self.gen_optimizer, self.disc_optimizer))

    def _end_step(self) -> None:
        if self._eager_steps > 0:
            self._eager_steps -= 1

    def _stage(self, name: str, value: Tensor) -> Tensor:
        """Copy a batch into a persistent device buffer (graph replays read fixed addresses)."""
        buf = self._static.get(name)
        if buf is None or buf.shape != value.shape:
            if buf is not None and self._graphs:
                raise RuntimeError(f'{name}: batch shape changed from {tuple(buf.shape)} to {tuple(value.shape)} after '
                                   'graph capture; use drop_last=True or use_graphs=False')
            buf = torch.empty(value.shape, dtype=torch.float32, device=self.device)
            self._static[name] = buf
        buf.copy_(value, non_blocking=True)
        return buf

    def _backward(self, loss: Tensor) -> None:
        """``loss.backward()`` with the conv weight gradients of the pass collected and issued together at its end
        (``functional.WeightGradQueue``): they are off the critical path until the optimiser step."""
        # (the root gradient is a persistent 1.0 on the device: autograd's own ones_like is a fill launch per backward pass)
        one = self.__dict__.get('_one')
        if one is None:
            one = self._one = torch.ones((), dtype=torch.float32, device=self.device)
        with F.deferred_weight_grads():
            loss.backward(one)

    def _resume(self, cut: str) -> None:
        with F.deferred_weight_grads():
            self._cuts.resume(cut)

    def _push_loss(self, key: str) -> None:
        """Last kernel of a step: append the step's train loss to the wandb ring (no-op without a wandb run)."""
        if self._ring is not None:
            self._ring.push(self._losses[key])

    def _note_loss(self, key: str, step: int, contents: dict) -> None:
        """trainer.py:393-399 / :459-466: one wandb sample per step, delivered at the next flush."""
        if self._ring is not None and self._ring.note(key, step, contents):
            self._ring.flush(self._log_wandb)

    def _flush_losses(self) -> None:
        if self._ring is not None:
            self._ring.flush(self._log_wandb)

    # ------------------------------------------------------------------ pre-training
    def _pretrain_body(self) -> None:
        """Loop body of ``_pretrain``, trainer.py:380-386.  The autocast region is the generator forward + MSE
        (``_enter_phase('psnr')``); bf16 needs no GradScaler, so ``scaler.scale / step / update`` reduce to
        ``backward`` + ``step``."""
        self.psnr_optimizer.zero_grad()
        super_res = self.generator(self._static['low_res'])
        loss = self.pixel_loss(super_res, self._static['high_res'])
        self._backward(loss)
        self._losses['psnr/train-loss'] = loss.detach()

    def pretrain_step(self, low_res: Tensor, high_res: Tensor) -> Tensor:
        """One SRResNet pre-training step on device tensors; returns the (device) loss."""
        self._enter_phase('psnr')
        self._losses = getattr(self, '_losses', {})
        self._stage('low_res', low_res)
        self._stage('high_res', high_res)
        if self.distributed:
            F.cut_hook[0] = self._cuts
            try:
                self._cuts.names = {'g.tail'}
                self._exec('psnr.head', self._pretrain_body)           # stops at the generator's 'g.tail' cut
                self.gen_sync.launch(1)                                # conv_layers / conv3 gradients: on the wire
                self._exec('psnr.body', lambda: self._resume('g.tail'))       # residual tower backward
                self.gen_sync.launch(0)
                self.gen_sync.wait()
                self._exec('psnr.opt', lambda: (self.psnr_optimizer.step(), self._push_loss('psnr/train-loss')))
            except BaseException:
                self.gen_sync.abort()  # a bucket may be on the wire with compute units reserved for it: give them back
                raise
            finally:
                F.cut_hook[0] = None
                self._cuts.clear()
        else:
            self._exec('psnr.all', lambda: (self._pretrain_body(), self.psnr_optimizer.step(),
                                            self._push_loss('psnr/train-loss')))
        self._end_step()
        if self._ring is not None:
            self._ring.mark_push()
        return self._losses['psnr/train-loss']

    def _pretrain(self) -> None:
        """trainer.py:345-414."""
        self._log('=' * 80)
        self._log('Starting pre-training')
        epoch = 1
        path = self.psnr_checkpoint or f'{self.phase_prefix}-psnr-latest.pth'
        checkpoint = self._load_checkpoint(path)
        if checkpoint:
            self.generator.load_state_dict(checkpoint['state'])
            F.bump_pack_epoch()
            epoch = checkpoint['epoch']
            if self._restore_resume_state(checkpoint):
                epoch += 1
        step = 0
        for epoch in range(epoch, self.pre_epochs + 1):
            self._log('-' * 80)
            self._log(f'Starting epoch {epoch} out of {self.pre_epochs}')
            self.generator.train()
            self.discriminator.train()
            start_time = time.time()
            loss = None
            for sub_step, (low_res, high_res) in enumerate(self.train_loader):
                loss = self.pretrain_step(low_res, high_res)
                step = (sub_step * self.batch_size * self.world_size) + ((epoch - 1) * self.train_len)
                self._note_loss('psnr/train-loss', step, {'psnr/epoch': epoch})   # trainer.py:393-399, every step
            torch.cuda.synchronize()
            self._flush_losses()
            time_taken = time.time() - start_time
            throughput = len(self.train_loader) * self.batch_size * self.world_size / time_taken
            self._log(f'Throughput: {round(throughput, 3)} images/sec')
            self._log_wandb({'psnr/throughput/train': throughput, 'psnr/epoch': epoch}, step=step)
            self._test(epoch, f'{self.phase_prefix}-psnr', step)

    # ------------------------------------------------------------------ GAN phase
    def _phase_disc(self) -> None:
        """trainer.py:442-450: G forward, D on real and fake, D backward (down to the 'd.head' cut when data
        parallel: the classifier's gradients are complete then, the convolutions' follow in ``_phase_disc_body``)."""
        # the three networks exchange NHWC tensors directly: the batch is converted once, the super-resolved image
        # never goes through the NCHW module boundary (8 layout passes forward, 4 backward in the reference's call form)
        self._phase_disc_gen()
        self._phase_disc_loss()

    def _phase_disc_gen(self) -> None:
        with torch.no_grad():
            low4 = F.to_nhwc(self._static['low_res'], 4)
            self._high4 = F.to_nhwc(self._static['high_res'], 4)
        self.disc_optimizer.zero_grad()                                      # :442
        self._super_res = self.generator.forward_nhwc(low4)                  # :444

    def _phase_disc_loss(self) -> None:
        # :446-448 -- D(real) and D(fake) as one batch; classifier tail, Sigmoid, both BCE terms and their sum are one launch
        # forward and one backward (functional.gan_head)
        disc_loss, _ = self.discriminator.pair_loss_nhwc(self._high4, self._super_res.detach())
        self._backward(disc_loss)                                            # :450
        self._losses['gan/disc-loss'] = disc_loss.detach()

    def _phase_disc_body(self) -> None:
        self._resume('d.head')

    def _phase_content(self) -> None:
        """trainer.py:453-455: VGG19 perceptual loss (does not need the updated discriminator)."""
        self.gen_optimizer.zero_grad()                                       # :453
        self._content = self.vgg_loss.forward_nhwc(self._super_res, self._high4)  # :455

    def _phase_gen(self) -> None:
        """trainer.py:451,456-468: D update, adversarial term through the UPDATED D, G backward (down to the
        'g.tail' cut when data parallel)."""
        self.disc_optimizer.step()                                           # :451
        with no_weight_grad():  # C5: D's weight gradients are never consumed here
            # :456-457 -- gen_loss = content + 0.001 * BCELoss(D(G(lr)), 1), formed in the discriminator head's launch
            gen_loss, aux = self.discriminator.adversarial_loss_nhwc(self._super_res, self._content, 0.001)
        self._backward(gen_loss)                                             # :468
        self._losses['gan/content-loss'] = self._content.detach()
        self._losses['gan/adversarial-loss'] = aux[1]
        self._losses['gan/train-loss'] = gen_loss.detach()
        self._super_res = self._content = self._high4 = None

    def _phase_gen_body(self) -> None:
        self._resume('g.tail')

    def _gan_all(self) -> None:
        if _dev.NO_OVERLAP or not self.overlap_branches or self.device.type != 'cuda':
            self._phase_disc()
            self._phase_content()
            self._phase_gen()
            self.gen_optimizer.step()                                        # :469
            self._push_loss('gan/train-loss')
            return
        # Two branches of ONE captured graph.  The perceptual loss (trainer.py:455) needs the generator's output and the
        # targets, nothing of the discriminator -- and its gradient needs nothing either: d gen_loss / d content is the root
        # gradient itself.  So the whole VGG19 round trip (forward, L1, backward down to the generator's output) runs on a side
        # stream, on a detached leaf of the super-resolved batch, next to the discriminator's update (forward on real + fake,
        # backward, weight gradients, Adam) AND the adversarial term's pass through the updated discriminator; the two
        # gradients arriving at the generator's output are added where autograd would have added them, and the generator's
        # backward runs once.  Every launch of either branch is cut for the whole chip; what the second branch buys is the
        # other branch's idle slots: partly filled last rounds, prologues and epilogues, the boundary between dependent
        # launches.  Same kernels on the same operands, the same two-term sum: the step's results do not change by a bit
        # (tools/overlap_check.py).
        main = torch.cuda.current_stream()
        side = self._side_stream()
        self._phase_disc_gen()
        sr = self._super_res
        deep = self.deep_overlap and not _dev.FWD_OVERLAP_ONLY
        sr_v = sr.detach().requires_grad_(True) if deep else sr
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._super_res = sr_v
            self._phase_content()
            content = self._content
            fwd_done = torch.cuda.Event()
            fwd_done.record(side)
            if deep:
                self._backward(content)           # VGG19 backward: d content / d sr_v, on the side stream
        self._super_res = sr
        self._phase_disc_loss()
        if not deep:
            main.wait_stream(side)
            content.record_stream(main)
            self._phase_gen()
        else:
            main.wait_event(fwd_done)             # the adversarial head adds the content VALUE to its term: forward only
            content.record_stream(main)
            sr_d = sr.detach().requires_grad_(True)
            self._super_res, self._content = sr_d, content.detach()
            self._phase_gen()                     # D update, adversarial term through the updated D, backward down to sr_d
            main.wait_stream(side)
            sr_v.grad.record_stream(main)
            with F.deferred_weight_grads():
                sr.backward(sr_d.grad + sr_v.grad)
        self.gen_optimizer.step()                                            # :469
        self._push_loss('gan/train-loss')

    def _side_stream(self):
        s = self.__dict__.get('_side')
        if s is None:
            s = self._side = torch.cuda.Stream(device=self.device)
        return s

    def gan_step(self, low_res: Tensor, high_res: Tensor) -> Dict[str, Tensor]:
        """One full GAN step (``_gan_loop`` without the logging); returns device loss tensors."""
        self._enter_phase('gan')
        self._losses = getattr(self, '_losses', {})
        self._stage('low_res', low_res)
        self._stage('high_res', high_res)
        if self.distributed:
            # hipGraph segments between the collectives; every all-reduce is launched the moment its bucket's
            # gradients are enqueued and awaited right before the optimiser that consumes it
            F.cut_hook[0] = self._cuts
            try:
                self._cuts.names = {'g.tail', 'd.head'}
                self._exec('gan.disc.head', self._phase_disc)
                self.disc_sync.launch(1)           # classifier.*: 75.5 MB, under D's conv backward + the VGG forward
                self._exec('gan.disc.body', self._phase_disc_body)
                self.disc_sync.launch(0)           # features.*: 19 MB, under the VGG forward
                self._cuts.names = {'g.tail'}
                self._exec('gan.content', self._phase_content)
                self.disc_sync.wait()
                self._cuts.names = set()           # the discriminator pass below is differentiated in ONE piece
                self._exec('gan.gen.head', self._phase_gen)
                self.gen_sync.launch(1)            # conv_layers.* / conv3.*: under the residual tower's backward
                self._exec('gan.gen.body', self._phase_gen_body)
                self.gen_sync.launch(0)
                self.gen_sync.wait()
                self._exec('gan.gopt', lambda: (self.gen_optimizer.step(), self._push_loss('gan/train-loss')))
            except BaseException:
                # (OOM, an error in user code inside a segment) buckets may be on the wire and compute units reserved for
                # RCCL's channels: without this every later plan and captured graph would be cut for a smaller chip
                self.disc_sync.abort()
                self.gen_sync.abort()
                raise
            finally:
                F.cut_hook[0] = None
                self._cuts.clear()  # a segment that raised between a cut and its resume must not leak into the next step
        else:
            self._exec('gan.all', self._gan_all)
        self._end_step()
        if self._ring is not None:
            self._ring.mark_push()
        return self._losses

    def _gan_loop(self, low_res: Tensor, high_res: Tensor, step: int) -> None:
        """trainer.py:416-469."""
        self.gan_step(low_res, high_res)
        self._note_loss('gan/train-loss', step, {'gan/disc-lr': self.disc_scheduler.get_last_lr()[0],   # :459-466,
                                                 'gan/gen-lr': self.gen_scheduler.get_last_lr()[0]})     # every step

    def _gan_train(self) -> None:
        """trainer.py:471-531."""
        self._log('=' * 80)
        self._log('Starting training loop')
        epoch = 1
        self.best_psnr = -1.0
        checkpoint = self._load_checkpoint(self.gan_checkpoint or f'{self.phase_prefix}-gan-latest.pth')
        if checkpoint:
            self.generator.load_state_dict(checkpoint['state'])
            epoch = checkpoint['epoch']
            if self._restore_resume_state(checkpoint):
                epoch += 1  # the checkpoint is written AFTER its epoch: continue with the next one
                            # (the reference re-runs the saved epoch, trainer.py:483-497)
        else:
            checkpoint = self._load_checkpoint(f'{self.phase_prefix}-psnr-latest.pth')
            if checkpoint:
                self.generator.load_state_dict(checkpoint['state'])
        F.bump_pack_epoch()
        step = 0
        for epoch in range(epoch, self.epochs + 1):
            self._log('-' * 80)
            self._log(f'Starting epoch {epoch} out of {self.epochs}')
            self.generator.train()
            self.discriminator.train()
            start_time = time.time()
            for sub_step, (low_res, high_res) in enumerate(self.train_loader):
                step = (sub_step * self.batch_size * self.world_size) + \
                       ((self.pre_epochs + epoch - 1) * self.train_len)
                self._gan_loop(low_res, high_res, step)
            torch.cuda.synchronize()
            self._flush_losses()
            time_taken = time.time() - start_time
            throughput = len(self.train_loader) * self.batch_size * self.world_size / time_taken
            self._log(f'Throughput: {round(throughput, 3)} images/sec')
            self._log_wandb({'gan/throughput/train': throughput, 'gan/epoch': epoch}, step=step)
            self.disc_scheduler.step()
            self.gen_scheduler.step()
            self._test(epoch, f'{self.phase_prefix}-gan', step)

    # ------------------------------------------------------------------ validation
    def _test(self, epoch: int, phase: str, step: int) -> None:
        """trainer.py:260-343: eval-mode G (fp32: no autocast there), per-batch PSNR (unclamped SR), best/latest
        checkpoints, monitor image."""
        self._enter_phase('test')
        self.generator.eval()
        self._log(f'Testing results after epoch {epoch}')
        with torch.no_grad():
            loss, psnr, batches = 0.0, 0.0, 0
            start_time = time.time()
            for low_res, _, high_res in self.test_loader:
                low_res = low_res.to(self.device)
                high_res = high_res.to(self.device)
                super_res = self.generator(low_res)
                mse = self.mse_loss(super_res, high_res).item()
                psnr += 10 * log10(1 / mse)                                   # :296
                loss += mse
                batches += 1
            if batches == 0:
                # a PSNR of 0 would be written to every later checkpoint decision; the reference divides by
                # len(test_loader) == 0 here
                raise RuntimeError('the test loader yielded no batch on this rank: the test shard is empty '
                                   f'(test_len {self.test_len}, world size {self.world_size})')
            time_taken = max(time.time() - start_time, 1e-9)
            throughput = batches * self.batch_size * self.world_size / time_taken
            psnr, loss = psnr / batches, loss / batches
            self._log(f'PSNR: {round(psnr, 3)}, Throughput: {round(throughput, 3)} images/sec')
            short_phase = ''.join(phase.split('-')[1:])
            self._log_wandb({f'{short_phase}/PSNR': psnr, f'{short_phase}/val-loss': loss,
                             f'{short_phase}/throughput/test': throughput, f'{short_phase}/epoch': epoch}, step=step)
            if psnr > self.best_psnr and self.main_process:
                self.best_psnr = psnr
                torch.save(self._model_state(epoch, phase), f'{phase}-best.pth')
            if self.main_process:
                torch.save(self._model_state(epoch, phase), f'{phase}-latest.pth')
            if self.save_image and self.main_process:
                super_res = self.generator(self.test_image)
                save_image(super_res, f'output/SR_epoch{epoch}.png')
                if wandb and getattr(wandb, 'run', None) is not None:        # :337-343, at 1/4 size
                    _, _, height, width = super_res.shape
                    small = torch.nn.functional.interpolate(super_res.clamp(0, 1).cpu(), size=(height // 4, width // 4),
                                                            mode='bicubic', align_corners=False).clamp(0, 1)
                    self._log_wandb({f'images/epoch{epoch}': wandb.Image(small[0].permute(1, 2, 0).numpy())})
        self.generator.train()

    def train(self) -> None:
        """trainer.py:533-543."""
        self._pretrain()
        torch.cuda.empty_cache()
        self._gan_train()
        self._cleanup()
