"""SRGAN trainer on MI355X -- interface of torchsr/srgan/trainer.py:39-543.

Same constructor, same two phases (PSNR pre-training, then GAN training), same losses,
optimiser settings, checkpoint files and per-epoch PSNR test as the reference.  What is
different is underneath:

* all device work goes through the hand-written HIP kernels (``torchsr_amd.functional``);
* the whole train step is captured once into a hipGraph and replayed (the step is ~700 kernel
  launches of a few microseconds each; eager launches would be host bound);
* data parallelism is one process per GPU with two flat-buffer all-reduces per step on RCCL
  (``torchsr_amd.ddp``) instead of ``DistributedDataParallel``; the discriminator's 94 MB
  gradient exchange overlaps the VGG19 forward;
* the discriminator pass inside the generator update does not compute the discriminator weight
  gradients the reference computes there and never uses.

AMP: the reference autocasts the pre-training phase to fp16 with loss scaling on CUDA
(trainer.py:382-388) and runs the GAN phase in fp32.  This implementation is fp32 in both phases
(exact-fp32 MFMA); ``--disable-amp`` is accepted and ignored.
"""
import os
import time
from argparse import Namespace
from math import log10
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .. import functional as F
from ..ddp import GradAllReduce, broadcast_module
from ..layers import no_weight_grad
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


class SRGANTrainer:
    """``SRGANTrainer(device, args, train_loader, test_loader, train_len, test_len, distributed)``.

    ``args`` needs the attributes the reference reads (trainer.py:67-84): ``disable_amp, batch_size,
    epochs, gan_checkpoint, local_rank, pretrain_epochs, psnr_checkpoint, skip_image_save,
    world_size, rank``.  Optional extras: ``use_graphs`` (default True), ``vgg_weights``.
    """

    phase_prefix = 'srgan'
    generator_cls = Generator
    discriminator_cls = Discriminator

    def __init__(self, device, args: Namespace, train_loader, test_loader, train_len: int, test_len: int,
                 distributed: bool = False) -> None:
        # The reference wraps its forward passes in torch.cuda.amp.autocast unless --disable-amp is given
        # (trainer.py:379-383,438-462).  Here that selects bf16 products with fp32 accumulation in every
        # convolution (forward + stride-1 data gradient, srx_conv2d_t::precision); tensors, BatchNorm, the
        # losses, the weight gradients and Adam stay fp32, so no GradScaler is needed.
        self.amp = not args.disable_amp
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
        # VGG19(high_res) on a second stream: measured 2 % slower (12.86 vs 12.60 ms/step) on MI355X, and a
        # fork / join pair cannot straddle the data-parallel graph segments, so off by default
        self.overlap_target_vgg = bool(getattr(args, 'overlap_target_vgg', False)) and not distributed
        self._target_feat = None
        if self.device.type != 'cuda':
            raise RuntimeError('torchsr_amd trains on an MI355X (device "cuda"); there is no CPU path')
        if self.device.index is None:
            # single-process runs carry local_rank -1 (torchsr.py:148-150)
            self.device = torch.device('cuda', max(int(self.local_rank or 0), 0))
        torch.cuda.set_device(self.device)
        if self.save_image and self.main_process and not os.path.exists('output'):
            os.makedirs('output')
        self._graphs: Dict[str, torch.cuda.CUDAGraph] = {}
        self._graph_pool = None
        self._calls: Dict[str, int] = {}
        self._static: Dict[str, Tensor] = {}
        F.direct_grads[0] = True  # parameter gradients accumulate straight into the flat .grad views
        # weight gradients on a second stream (a parallel hipGraph branch): measured 8 % SLOWER on MI355X
        # (cross-queue dependencies cost more than the overlap wins), so off unless asked for
        F.side_stream_enabled[0] = bool(getattr(args, 'side_stream', False))
        self._initialize_trainer()
        self._create_test_image()

    # ------------------------------------------------------------------ set-up
    def _initialize_trainer(self) -> None:
        self._initialize_models()
        self._initialize_loss()
        self._initialize_optimizers()
        if self.amp:
            from ..layers import set_conv_precision
            for module in (self.generator, self.discriminator, self.vgg_loss):
                set_conv_precision(module, 'bf16')

    def _initialize_models(self) -> None:
        """trainer.py:136-157.  DDP wrapping is replaced by flat buffers + explicit all-reduce."""
        self.generator = self.generator_cls().to(self.device)
        self.discriminator = self.discriminator_cls().to(self.device)
        if self.distributed:
            broadcast_module(self.generator)
            broadcast_module(self.discriminator)
        self.gen_flat = FlatParams(self.generator)
        self.disc_flat = FlatParams(self.discriminator)
        self.gen_sync = GradAllReduce(self.gen_flat) if self.distributed else None
        self.disc_sync = GradAllReduce(self.disc_flat) if self.distributed else None

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
        if wandb and self.main_process:
            wandb.log(contents, step=step)

    def _cleanup(self) -> None:
        if wandb:
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
        if not self.use_graphs:
            fn()
            return
        g = self._graphs.get(key)
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
        # with a process group alive, RCCL's watchdog thread may query events while we capture; only
        # the capturing thread's own calls should be policed then
        mode = 'thread_local' if self.distributed else 'global'
        try:
            with torch.cuda.graph(g, pool=self._graph_pool, capture_error_mode=mode):
                fn()
        except Exception as exc:  # keep training: run this and all later steps eagerly
            self._log(f'hipGraph capture of {key} failed ({type(exc).__name__}: {exc}); continuing without graphs')
            self.use_graphs = False
            self._graphs.clear()
            torch.cuda.synchronize()
            F.bump_pack_epoch()  # packs "refreshed" during the failed capture never ran
            fn()
            return
        self._graphs[key] = g
        g.replay()  # capture records without executing; run this step's work now

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

    # ------------------------------------------------------------------ pre-training
    def _pretrain_body(self) -> None:
        """Loop body of ``_pretrain``, trainer.py:380-388 (no autocast / GradScaler: fp32)."""
        self.psnr_optimizer.zero_grad()
        super_res = self.generator(self._static['low_res'])
        loss = self.pixel_loss(super_res, self._static['high_res'])
        loss.backward()
        F.join_side_stream()
        self._losses['psnr/train-loss'] = loss.detach()

    def pretrain_step(self, low_res: Tensor, high_res: Tensor) -> Tensor:
        """One SRResNet pre-training step on device tensors; returns the (device) loss."""
        self._losses = getattr(self, '_losses', {})
        self._stage('low_res', low_res)
        self._stage('high_res', high_res)
        if self.distributed:
            self._exec('psnr.fwdbwd', self._pretrain_body)
            self.gen_sync.launch()
            self.gen_sync.wait()
            self._exec('psnr.opt', self.psnr_optimizer.step)
        else:
            self._exec('psnr.all', lambda: (self._pretrain_body(), self.psnr_optimizer.step()))
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
                if wandb:
                    self._log_wandb({'psnr/train-loss': loss, 'psnr/epoch': epoch}, step=step)
            torch.cuda.synchronize()
            time_taken = time.time() - start_time
            throughput = len(self.train_loader) * self.batch_size * self.world_size / time_taken
            self._log(f'Throughput: {round(throughput, 3)} images/sec')
            self._log_wandb({'psnr/throughput/train': throughput, 'psnr/epoch': epoch}, step=step)
            self._test(epoch, f'{self.phase_prefix}-psnr', step)

    # ------------------------------------------------------------------ GAN phase
    def _fork_target_features(self, high_res: Tensor) -> None:
        """VGG19(high_res) (loss.py:53, detached) depends on nothing but the HR batch: issue its 115 GFLOP on
        a second stream at the top of the step so that the big VGG kernels fill the CUs the generator's
        small launches leave idle (one fork / one join: a single parallel branch in the hipGraph)."""
        self._target_feat = None
        if not self.overlap_target_vgg:
            return
        main = torch.cuda.current_stream()
        side = F.side_stream(self.device.index)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._target_feat = self.vgg_loss.target_features(high_res)
        self._target_feat.record_stream(main)

    def _join_target_features(self) -> Optional[Tensor]:
        if self._target_feat is None:
            return None
        torch.cuda.current_stream().wait_stream(F.side_stream(self.device.index))
        feat, self._target_feat = self._target_feat, None
        return feat

    def _phase_disc(self) -> None:
        """trainer.py:442-450: G forward, D on real and fake, D backward."""
        low_res, high_res = self._static['low_res'], self._static['high_res']
        self._fork_target_features(high_res)
        self.disc_optimizer.zero_grad()                                      # :442
        self._super_res = self.generator(low_res)                            # :444
        d_real = self.bce_loss(self.discriminator(high_res), 1.0)            # :446
        d_fake = self.bce_loss(self.discriminator(self._super_res.detach()), 0.0)  # :447
        disc_loss = F.axpby(d_real, d_fake, 1.0, 1.0)                        # :448
        disc_loss.backward()                                                 # :450
        F.join_side_stream()
        self._losses['gan/disc-loss'] = disc_loss.detach()

    def _phase_content(self) -> None:
        """trainer.py:453-455: VGG19 perceptual loss (does not need the updated discriminator)."""
        self.gen_optimizer.zero_grad()                                       # :453
        self._content = self.vgg_loss(self._super_res, self._static['high_res'],
                                      target_features=self._join_target_features())  # :455

    def _phase_gen(self) -> None:
        """trainer.py:451,456-468: D update, adversarial term through the UPDATED D, G backward."""
        self.disc_optimizer.step()                                           # :451
        with no_weight_grad():  # C5: D's weight gradients are never consumed here
            adversarial = self.bce_loss(self.discriminator(self._super_res), 1.0)  # :456
        gen_loss = F.axpby(self._content, adversarial, 1.0, 0.001)           # :457
        gen_loss.backward()                                                  # :468
        F.join_side_stream()
        self._losses['gan/content-loss'] = self._content.detach()
        self._losses['gan/adversarial-loss'] = adversarial.detach()
        self._losses['gan/train-loss'] = gen_loss.detach()
        self._super_res = self._content = None

    def _gan_all(self) -> None:
        self._phase_disc()
        self._phase_content()
        self._phase_gen()
        self.gen_optimizer.step()                                            # :469

    def gan_step(self, low_res: Tensor, high_res: Tensor) -> Dict[str, Tensor]:
        """One full GAN step (``_gan_loop`` without the logging); returns device loss tensors."""
        self._losses = getattr(self, '_losses', {})
        self._stage('low_res', low_res)
        self._stage('high_res', high_res)
        if self.distributed:
            self._exec('gan.disc', self._phase_disc)
            self.disc_sync.launch()            # 94 MB all-reduce rides under the VGG forward
            self._exec('gan.content', self._phase_content)
            self.disc_sync.wait()
            self._exec('gan.gen', self._phase_gen)
            self.gen_sync.launch()
            self.gen_sync.wait()
            self._exec('gan.gopt', self.gen_optimizer.step)
        else:
            self._exec('gan.all', self._gan_all)
        return self._losses

    def _gan_loop(self, low_res: Tensor, high_res: Tensor, step: int) -> None:
        """trainer.py:416-469."""
        losses = self.gan_step(low_res, high_res)
        if wandb:
            self._log_wandb({'gan/disc-lr': self.disc_scheduler.get_last_lr()[0],
                             'gan/gen-lr': self.gen_scheduler.get_last_lr()[0],
                             'gan/train-loss': losses['gan/train-loss']}, step=step)

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
            time_taken = time.time() - start_time
            throughput = len(self.train_loader) * self.batch_size * self.world_size / time_taken
            self._log(f'Throughput: {round(throughput, 3)} images/sec')
            self._log_wandb({'gan/throughput/train': throughput, 'gan/epoch': epoch}, step=step)
            self.disc_scheduler.step()
            self.gen_scheduler.step()
            self._test(epoch, f'{self.phase_prefix}-gan', step)

    # ------------------------------------------------------------------ validation
    def _test(self, epoch: int, phase: str, step: int) -> None:
        """trainer.py:260-343: eval-mode G, per-batch PSNR (unclamped SR), best/latest checkpoints."""
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
            time_taken = max(time.time() - start_time, 1e-9)
            batches = max(batches, 1)
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
        self.generator.train()

    def train(self) -> None:
        """trainer.py:533-543."""
        self._pretrain()
        torch.cuda.empty_cache()
        self._gan_train()
        self._cleanup()
