"""Flat parameter / gradient buffers and the fused Adam of the trainers.

The reference builds three ``torch.optim.Adam`` instances over ~150 small tensors
(torchsr/srgan/trainer.py:171-185) and lets DDP bucket the gradients.  Here each
model's parameters are re-pointed into ONE contiguous fp32 buffer (and ``.grad``
into a second one), so that

* the optimiser is a single HBM-bound kernel launch (``srx_adam_step``),
* the data-parallel exchange is one RCCL all-reduce per model on the flat gradient
  buffer, with no flatten / unflatten copies,
* ``zero_grad`` is one memset,

while ``state_dict()`` / ``load_state_dict()`` keep working on the (now view) parameters.
"""
from typing import Iterable, List

import torch
from torch import nn, Tensor

from . import functional as F
from ._lib import call


class FlatParams:
    """Re-point ``module``'s parameters and gradients into flat, 16-byte aligned buffers."""

    ALIGN = 4  # floats

    def __init__(self, module: nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise RuntimeError('FlatParams: module has no trainable parameters')
        dev, dt = params[0].device, params[0].dtype
        if dt != torch.float32:
            raise RuntimeError('FlatParams: fp32 parameters expected')
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.data = torch.zeros(total, dtype=dt, device=dev)
        self.grad = torch.zeros(total, dtype=dt, device=dev)
        self.params: List[nn.Parameter] = params
        self.offsets: List[int] = offs
        self.numel = total
        # bumped by every optimiser step on this buffer: the convs of THIS model repack, other models' do not
        self.pack_epoch = [0]
        for m in module.modules():
            st = getattr(m, '_st', None)
            if isinstance(st, F.ConvState):
                st.model_epoch = self.pack_epoch
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.data[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.data[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)

    def offsets_by_name(self, module: nn.Module) -> dict:
        """``{parameter name: offset (floats) in the flat buffers}`` for the module this buffer was built from."""
        where = {id(p): o for p, o in zip(self.params, self.offsets)}
        return {n: where[id(p)] for n, p in module.named_parameters() if id(p) in where}

    def zero_grad(self) -> None:
        """One memset; re-attaches the views in case something set ``.grad = None``."""
        self.grad.zero_()
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * 4:
                p.grad = self.grad[off:off + n].view(p.shape)
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN


class FlatAdam:
    """``torch.optim.Adam(params, lr, betas)`` semantics on a :class:`FlatParams` buffer.

    Several instances may share one ``FlatParams`` (the reference keeps ``psnr_optimizer`` and
    ``gen_optimizer`` as separate Adam states over the same generator, trainer.py:171-185).
    ``lr`` and the step count live on the device (hipGraph friendly).
    """

    def __init__(self, flat: FlatParams, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8):
        self.flat = flat
        self.betas, self.eps = betas, eps
        dev = flat.data.device
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.step_count = torch.zeros((), dtype=torch.int64, device=dev)
        self.lr_dev = torch.tensor(float(lr), dtype=torch.float32, device=dev)
        self._lr = float(lr)
        self.grad_scale = 1.0
        self.pack_table = None  # functional.PackTable of the model's convs (set by the trainer)
        self.param_groups = [{'lr': float(lr), 'initial_lr': float(lr)}]  # StepLR-style access

    @property
    def lr(self) -> float:
        return self._lr

    def set_lr(self, lr: float) -> None:
        self._lr = float(lr)
        self.param_groups[0]['lr'] = float(lr)
        self.lr_dev.fill_(float(lr))

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self) -> None:
        f = self.flat
        call('srx_adam_step', f.data.data_ptr(), f.grad.data_ptr(), self.exp_avg.data_ptr(),
             self.exp_avg_sq.data_ptr(), f.numel, self.lr_dev.data_ptr(), self.betas[0], self.betas[1], self.eps,
             self.grad_scale, self.step_count.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self.flat.pack_epoch[0] += 1
        if self.pack_table is not None:  # every conv of the model repacked by one launch
            self.pack_table.run()

    def state_dict(self):
        return {'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'step': self.step_count, 'lr': self._lr}

    def load_state_dict(self, sd) -> None:
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.step_count.copy_(sd['step'])
        self.set_lr(sd['lr'])


class StepLR:
    """torch.optim.lr_scheduler.StepLR(step_size, gamma) as used at torchsr/srgan/trainer.py:186-195.

    The reference passes ``step_size=epochs // 8``, which is 0 (ZeroDivisionError at the first
    ``step()``) for ``--epochs < 8``; here the step size is clamped to 1 instead.
    """

    def __init__(self, optimizer: FlatAdam, step_size: int, gamma: float = 0.1):
        self.optimizer, self.step_size, self.gamma = optimizer, max(1, int(step_size)), gamma
        self.base_lr = optimizer.lr
        self.last_epoch = 0

    def step(self) -> None:
        self.last_epoch += 1
        self.optimizer.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))

    def get_last_lr(self):
        return [self.optimizer.lr]

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'base_lr': self.base_lr}

    def load_state_dict(self, sd) -> None:
        self.last_epoch, self.base_lr = sd['last_epoch'], sd['base_lr']
        self.optimizer.set_lr(self.base_lr * self.gamma ** (self.last_epoch // self.step_size))
