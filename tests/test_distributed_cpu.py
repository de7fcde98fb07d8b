"""world_size-2 gloo tests (CPU): the bucketed flat-buffer gradient exchange that replaces DDP
(torchsr/srgan/trainer.py:142-157), the paused backward pass that feeds it, and the launcher's
environment parsing.  (The trainers themselves need the GPU: tests/test_ddp_gpu.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from torchsr_amd import functional as F
    from torchsr_amd.ddp import BackwardCuts, GradBuckets, broadcast_module, describe_group
    from torchsr_amd.optim import FlatParams
    assert describe_group() == {'backend': 'gloo', 'world_size': world, 'rank': rank}
    torch.manual_seed(100 + rank)                      # different init per rank ...

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.body = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh())
            self.head = torch.nn.Linear(5, 3)

        def forward(self, x):
            return self.head(F.cut_point('head', self.body(x)))  # the trainers' modules cut the same way

    model = Net()
    broadcast_module(model)                            # ... made identical, as DDP.__init__ does
    flat = FlatParams(model)
    sync = GradBuckets(flat, ('head.weight',), model)  # [0] = body (complete last), [1] = head (complete first)
    assert len(sync) == 2 and sync.slices[0].numel() == 32 + 8 and sync.slices[1].numel() == 16 + 4
    assert sync.world_size == world and abs(sync.scale - 1.0 / world) < 1e-12
    torch.manual_seed(7)
    x_all, y_all = torch.rand(8, 6), torch.rand(8, 3)  # global batch; this rank takes its shard
    xs, ys = x_all[rank::world], y_all[rank::world]
    flat.zero_grad()
    cuts = BackwardCuts(('head',))
    F.cut_hook[0] = cuts
    try:
        torch.nn.functional.mse_loss(model(xs), ys).backward()
    finally:
        F.cut_hook[0] = None
    # the backward pass stopped at the cut: the head's gradients exist, the body's do not yet
    assert model.head.weight.grad.abs().sum() > 0 and float(model.body[0].weight.grad.abs().sum()) == 0.0
    assert cuts.pending('head')
    sync.launch(1)                                     # head bucket is on the wire ...
    cuts.resume('head')                                # ... while the body's backward runs
    assert not cuts.pending('head') and model.body[0].weight.grad.abs().sum() > 0
    sync.launch(0)
    sync.wait()
    mean_grad = flat.grad * sync.scale                 # the optimiser folds this scale in
    # reference: one process, whole batch (per-rank mean losses average to the global mean loss)
    ref = Net()
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.mse_loss(ref(x_all), y_all).backward()
    ref_grad = torch.cat([torch.nn.functional.pad(p.grad.flatten(), (0, (-p.numel()) % 4)) for p in ref.parameters()])
    out[rank] = (float((mean_grad - ref_grad).abs().max()), float(flat.data.sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_with_paused_backward_equals_large_batch_gradient():
    world, port = 2, _free_port()
    mgr = mp.get_context('spawn').Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert len(out) == world
    errs = [out[r][0] for r in range(world)]
    sums = [out[r][1] for r in range(world)]
    assert max(errs) < 1e-6, errs
    assert abs(sums[0] - sums[1]) < 1e-7          # parameters were broadcast from rank 0


def test_launcher_env_parsing(monkeypatch):
    from argparse import Namespace
    from torchsr_amd.torchsr import distributed_params, parse_args, positive_integer
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'SLURM_NTASKS', 'SLURM_PROCID'):
        monkeypatch.delenv(k, raising=False)
    a, d = distributed_params(Namespace(seed=0))
    assert not d and (a.world_size, a.rank, a.local_rank, a.local_world_size) == (1, -1, -1, 1)
    monkeypatch.setenv('WORLD_SIZE', '8'); monkeypatch.setenv('RANK', '3')
    monkeypatch.setenv('LOCAL_RANK', '3'); monkeypatch.setenv('LOCAL_WORLD_SIZE', '8')
    a, d = distributed_params(Namespace(seed=5))
    assert d and (a.world_size, a.rank, a.local_rank) == (8, 3, 3)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE'):
        monkeypatch.delenv(k)
    monkeypatch.setenv('SLURM_NTASKS', '16'); monkeypatch.setenv('SLURM_PROCID', '9')
    monkeypatch.setenv('SLURM_LOCALID', '1'); monkeypatch.setenv('SLURM_NTASKS_PER_NODE', '8')
    a, d = distributed_params(Namespace(seed=0))
    assert d and (a.world_size, a.rank, a.local_rank, a.local_world_size) == (16, 9, 1, 8)
    # defaults mirror torchsr/constants.py and the parser at torchsr.py:173-235
    t = parse_args(['train'])
    assert (t.batch_size, t.epochs, t.pretrain_epochs, t.train_dir, t.model, t.data_workers, t.seed) == \
        (64, 1000, 1000, 'dataset', 'ESRGAN', 16, 0)
    assert parse_args(['test', 'img.png', '--model', 'srgan']).image == 'img.png'
    with pytest.raises(Exception):
        positive_integer('0')


def test_synthetic_dataset_contract():
    from torchsr_amd.dataset import initialize_datasets
    tr, te, ntr, nte = initialize_datasets('synthetic:12', batch_size=4, crop_size=96, upscale_factor=4)
    assert ntr == 12 and nte == 1 and len(tr) == 3
    lr, hr = next(iter(tr))
    assert lr.shape == (4, 3, 24, 24) and hr.shape == (4, 3, 96, 96) and 0 <= float(lr.min()) and float(hr.max()) <= 1
    # a test shard smaller than the batch size still yields its (partial) batch: the reference's test loader
    # drops nothing (torchsr/dataset.py:345-360), and a PSNR over zero batches would poison the checkpoints
    assert len(te) == 1
    lr, bic, hr = next(iter(te))
    assert lr.shape == (1, 3, 24, 24) and bic.shape == hr.shape == (1, 3, 96, 96)
