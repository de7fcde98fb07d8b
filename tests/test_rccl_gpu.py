"""The data-parallel product path on its real backend: RCCL (``torch.distributed`` backend ``nccl`` on ROCm).

One MI355X is all a test box has and RCCL refuses two ranks on one device, so the process group has world size 1:
the sums are no-ops, but everything RCCL-specific runs -- ``init_process_group('nccl', device_id=...)`` as in
torchsr/torchsr.py:257-258, ``all_reduce(async_op=True)`` of the four flat-buffer buckets on RCCL's own stream,
``Work.wait()`` ordering the compute stream behind them BETWEEN replayed hipGraph segments, hipGraph capture with
RCCL's watchdog thread alive (``capture_error_mode='thread_local'``).  The segmented trainer
(``distributed=True``: backward paused at 'd.head' / 'g.tail', seven segments; what torchsr/srgan/trainer.py:142-157
wraps in DistributedDataParallel) must track the single-graph trainer step for step, for SRGAN and ESRGAN.
The arithmetic of N > 1 ranks is pinned by tests/test_ddp_gpu.py (gloo, two ranks, data-parallel oracle).

The process group lives in a spawned child: the pytest process has long initialised the GPU and must not grow a
second HIP context owner mid-run; the child initialises RCCL before anything else touches the card.
"""
import os
import socket
import warnings
from argparse import Namespace

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

STEPS = 5  # graph trainers: 2 eager warm-ups, capture at step 2, two replays
SEGMENTS = {'gan.disc.head', 'gan.disc.body', 'gan.content', 'gan.gen.head', 'gan.gen.body', 'gan.gopt'}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import numpy as np
    import torch.distributed as dist
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    try:
        from oracle.weights import closed_form_state, step_state
        from torchsr_amd import functional as F
        from torchsr_amd.esrgan.trainer import ESRGANTrainer
        from torchsr_amd.srgan.trainer import SRGANTrainer
        golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
        report = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
        for tag, cls, fixture in (('srgan', SRGANTrainer, 'srgan_steps.npz'), ('esrgan', ESRGANTrainer, 'esrgan.npz')):
            gold = np.load(os.path.join(golden, fixture))
            lr, hr = torch.from_numpy(gold['low_res']).to(dev), torch.from_numpy(gold['high_res']).to(dev)

            def make(distributed):
                args = Namespace(disable_amp=True, batch_size=2, epochs=8, gan_checkpoint=None, local_rank=0,
                                 pretrain_epochs=1, psnr_checkpoint=None, skip_image_save=True, world_size=1,
                                 rank=0 if distributed else -1, use_graphs=True, vgg_weights='random',
                                 force_collectives=True)
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    t = cls(dev, args, [], [], 2, 2, distributed=distributed)
                t.generator.load_state_dict(step_state(t.generator.state_dict(), f'{tag}.G'))
                t.discriminator.load_state_dict(step_state(t.discriminator.state_dict(), f'{tag}.D'))
                t.vgg_loss.features.load_state_dict(closed_form_state(t.vgg_loss.features.state_dict(), prefix='features.'))
                t.generator.train()
                t.discriminator.train()
                return t

            fused, seg = make(False), make(True)
            rep = {'loss_gap': 0.0, 'buckets': (len(seg.gen_sync), len(seg.disc_sync))}
            for _step in range(STEPS):
                lf, ls = fused.gan_step(lr, hr), seg.gan_step(lr, hr)
                for k in lf:
                    a, b = lf[k].item(), ls[k].item()
                    rep['loss_gap'] = max(rep['loss_gap'], abs(a - b) / max(abs(a), 1e-3))
                assert not seg._cuts.pairs
            rep['issued_gan'] = seg.gen_sync.issued + seg.disc_sync.issued
            pf, ps = fused.pretrain_step(lr, hr).item(), seg.pretrain_step(lr, hr).item()
            rep['pre_gap'] = abs(pf - ps) / max(abs(pf), 1e-3)
            rep['issued_all'] = seg.gen_sync.issued + seg.disc_sync.issued
            gap = 0.0
            for mf, ms in ((fused.generator, seg.generator), (fused.discriminator, seg.discriminator)):
                for (k, a), (_, b) in zip(mf.state_dict().items(), ms.state_dict().items()):
                    if a.is_floating_point():
                        gap = max(gap, ((a - b).abs().max() / a.abs().max().clamp_min(1e-6)).item())
            rep['param_gap'] = gap
            rep['graphs'] = sorted(seg._graphs)
            rep['fused_graphs'] = sorted(fused._graphs)
            rep['use_graphs'] = (fused.use_graphs, seg.use_graphs)
            rep['cut_hook_reset'] = F.cut_hook[0] is None
            report[tag] = rep
            del fused, seg
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        out[0] = report
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_segmented_trainers_on_rccl_world_size_one(dev):
    port = _free_port()
    mgr = mp.get_context('spawn').Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(port, out), nprocs=1, join=True)
    rep = out[0]
    assert rep['backend'] == 'nccl' and rep['world'] == 1, rep
    for tag in ('srgan', 'esrgan'):
        r = rep[tag]
        assert r['use_graphs'] == (True, True), (tag, r)  # no capture fell back to eager under the RCCL watchdog
        assert SEGMENTS <= set(r['graphs']) and 'gan.all' in r['fused_graphs'], (tag, r)
        assert r['buckets'] == (2, 2), (tag, r)
        assert r['issued_gan'] == 4 * STEPS and r['issued_all'] == 4 * STEPS + 2, (tag, r)  # every bucket went through RCCL
        assert r['loss_gap'] < 1e-5 and r['pre_gap'] < 1e-5 and r['param_gap'] < 1e-4, (tag, r)
        assert r['cut_hook_reset'], (tag, r)
