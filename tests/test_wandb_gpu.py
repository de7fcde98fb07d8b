"""wandb logging parity: the keys, the step axis and the per-step cadence of
torchsr/srgan/trainer.py:311-319,340-343,393-412,459-466,520-526 and torchsr/torchsr.py:242-243 (the ESRGAN trainer
logs the same keys, esrgan/trainer.py:311-319,393-412,471-478,536-542), checked with a stub ``wandb`` module -- the real
package is not installed, exactly as the reference allows (its import is optional, trainer.py:23-26).

What differs from the reference by design: the per-step ``*/train-loss`` samples are delivered in batches (one device
-> host copy per ``wandb_flush_every`` steps, ``trainer.LossRing``) instead of one tensor read-back per step; every
sample is still logged, under its own step id, before any later-step entry.
"""
import math
import types
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu


class StubWandb(types.ModuleType):
    """init / log / finish / Image / run: what the reference touches."""

    def __init__(self):
        super().__init__('wandb')
        self.run = None
        self.inits, self.logs, self.finished = [], [], 0

    def init(self, **kw):
        self.inits.append(kw)
        self.run = object()
        return self.run

    def log(self, contents, step=None):
        assert self.run is not None
        self.logs.append((dict(contents), step))

    def finish(self):
        self.finished += 1
        self.run = None

    class Image:  # noqa: D106
        def __init__(self, data):
            self.data = data


def install(monkeypatch):
    import torchsr_amd.srgan.trainer as trainer_mod
    import torchsr_amd.torchsr as cli_mod
    stub = StubWandb()
    monkeypatch.setattr(trainer_mod, 'wandb', stub)
    monkeypatch.setattr(cli_mod, 'wandb', stub)
    return stub


@pytest.mark.parametrize('model,crop', [('srgan', 96), ('esrgan', 128)])
def test_cli_logs_the_reference_keys_on_the_reference_step_axis(dev, tmp_path, monkeypatch, model, crop):
    from torchsr_amd.torchsr import main
    import torchsr_amd.srgan.trainer as trainer_mod
    stub = install(monkeypatch)
    monkeypatch.setattr(trainer_mod.SRGANTrainer, 'wandb_flush_every', 3)  # 4 steps per epoch: a full ring AND a remainder
    monkeypatch.chdir(tmp_path)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'SLURM_NTASKS'):
        monkeypatch.delenv(k, raising=False)
    batch, n, pre, epochs = (4, 16, 1, 2) if model == 'srgan' else (2, 8, 1, 1)
    argv = ['train', '--model', model, '--train-dir', f'synthetic:{n}', '--batch-size', str(batch), '--epochs', str(epochs),
            '--pretrain-epochs', str(pre), '--seed', '3', '--vgg-weights', 'random']
    if model == 'srgan':
        argv.append('--disable-amp')
    main(argv)
    # torchsr.py:242-243
    assert len(stub.inits) == 1 and stub.inits[0]['name'] == 'TorchSR' and stub.inits[0]['project'] == 'torchsr'
    assert isinstance(stub.inits[0]['config'], Namespace) and stub.finished == 1
    steps_per_epoch = n // batch

    want = []
    for epoch in range(1, pre + 1):  # trainer.py:345-414
        for sub in range(steps_per_epoch):
            want.append(({'psnr/train-loss', 'psnr/epoch'}, sub * batch + (epoch - 1) * n))
        last = (steps_per_epoch - 1) * batch + (epoch - 1) * n
        want.append(({'psnr/throughput/train', 'psnr/epoch'}, last))
        want.append(({'psnr/PSNR', 'psnr/val-loss', 'psnr/throughput/test', 'psnr/epoch'}, last))
        want.append(({f'images/epoch{epoch}'}, None))
    for epoch in range(1, epochs + 1):  # trainer.py:471-531
        for sub in range(steps_per_epoch):
            want.append(({'gan/disc-lr', 'gan/gen-lr', 'gan/train-loss'}, sub * batch + (pre + epoch - 1) * n))
        last = (steps_per_epoch - 1) * batch + (pre + epoch - 1) * n
        want.append(({'gan/throughput/train', 'gan/epoch'}, last))
        want.append(({'gan/PSNR', 'gan/val-loss', 'gan/throughput/test', 'gan/epoch'}, last))
        want.append(({f'images/epoch{epoch}'}, None))
    got = [(set(c), s) for c, s in stub.logs]
    assert got == want, (got, want)
    for contents, _ in stub.logs:
        for k, v in contents.items():
            if k.startswith('images/'):
                assert isinstance(v, StubWandb.Image) and v.data.shape[2] == 3
            else:
                assert isinstance(v, (int, float)) and math.isfinite(v), (k, v)
    lrs = [c['gan/gen-lr'] for c, _ in stub.logs if 'gan/gen-lr' in c]
    assert lrs[0] == pytest.approx(1e-4)


def test_loss_ring_delivers_every_steps_loss(dev, monkeypatch):
    """The ring (pushed by the last kernel of the replayed hipGraph) against the loss read back step by step."""
    import torchsr_amd.srgan.trainer as trainer_mod
    stub = install(monkeypatch)
    stub.init(name='t')
    monkeypatch.setattr(trainer_mod.SRGANTrainer, 'wandb_flush_every', 4)
    args = Namespace(disable_amp=True, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=True, vgg_weights='random')
    torch.manual_seed(0)
    t = trainer_mod.SRGANTrainer(dev, args, [], [], 2, 2)
    assert t._ring is not None and t._ring.cap == 4
    lr, hr = torch.rand(2, 3, 24, 24, device=dev), torch.rand(2, 3, 96, 96, device=dev)
    truth = []
    for step in range(11):  # 2 eager, capture, replays; two full rings and a remainder of three
        t._gan_loop(lr, hr, 100 + step)
        truth.append(t._losses['gan/train-loss'].item())
    assert 'gan.all' in t._graphs
    assert len(stub.logs) == 8  # two flushes so far
    t._flush_losses()
    got = [(c['gan/train-loss'], s) for c, s in stub.logs]
    assert [s for _, s in got] == list(range(100, 111))
    assert [v for v, _ in got] == truth
    for step in range(3):  # the pre-training body shares the ring
        loss = t.pretrain_step(lr, hr)
        t._note_loss('psnr/train-loss', 200 + step, {'psnr/epoch': 1})
        truth.append(loss.item())
    t._flush_losses()
    tail = [(c['psnr/train-loss'], s, c['psnr/epoch']) for c, s in stub.logs[11:]]
    assert tail == [(truth[11 + i], 200 + i, 1) for i in range(3)]


def test_loss_ring_survives_steps_called_without_a_note(dev, monkeypatch):
    """``gan_step`` called directly (bench, tests, user code) pushes a record nobody notes: the next flush must not
    hand every later wandb sample another step's loss -- the ring resynchronises from the device counter."""
    import torchsr_amd.srgan.trainer as trainer_mod
    stub = install(monkeypatch)
    stub.init(name='t')
    monkeypatch.setattr(trainer_mod.SRGANTrainer, 'wandb_flush_every', 4)
    args = Namespace(disable_amp=True, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=True, vgg_weights='random')
    torch.manual_seed(0)
    t = trainer_mod.SRGANTrainer(dev, args, [], [], 2, 2)
    lr, hr = torch.rand(2, 3, 24, 24, device=dev), torch.rand(2, 3, 96, 96, device=dev)
    for _ in range(3):
        t.gan_step(lr, hr)  # three pushes, no note
    t._flush_losses()       # nothing to deliver; the three records are skipped
    assert stub.logs == [] and t._ring.unnoted == 3
    truth = []
    for step in range(6):
        if step == 2:
            t.gan_step(lr, hr)  # a stray direct call in the middle of a loop's window
        t._gan_loop(lr, hr, 300 + step)
        truth.append(t._losses['gan/train-loss'].item())
    t._flush_losses()
    got = [(c['gan/train-loss'], s) for c, s in stub.logs]
    assert [s for _, s in got] == list(range(300, 306))
    # every note remembers which record it belongs to: the stray record in the middle of the window shifts nothing
    assert [v for v, _ in got] == truth
    assert t._ring.unnoted == 4
