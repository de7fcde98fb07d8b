"""Headline benchmark: 96x96 HR crops/sec of the SRGAN GAN train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / [2]): full SRGAN GAN step -- generator forward, three
discriminator forwards, discriminator backward + Adam, VGG19 perceptual loss (two forwards, one
data-gradient backward), generator backward + Adam -- on a batch of 16 synthetic 96x96 HR crops
per GPU, fp32, random-init weights (VGG19 features seeded-random: the pretrained file is not
available offline).  Inputs are resident in HBM before the timed region.  One process per GPU;
at N > 1 the two flat gradient buffers are all-reduced over RCCL (weak scaling).

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  "roofline":     the dominant kernel (by device time) measured live with HIP events on the
                  launch stream in an instrumented pass of the same step: algorithmic FLOPs of
                  its launches / their event-measured duration, against the 157.3 TFLOP/s fp32
                  MFMA peak of MI355X;
  "cpu_baseline": the CPU oracle (oracle/srgan.py, stock torch ops) running the identical step
                  on the host cores -- a reported baseline, not the target.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from argparse import Namespace

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BATCH = 16          # per GPU (BASELINE.json configs[1])
CROP = 96
GF_PER_CROP = 43.23  # necessary algorithmic GFLOP per crop of the GAN step
PEAK_TFLOPS = 157.3  # fp32 MFMA, /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"


def synth_batch(device, rank):
    g = torch.Generator().manual_seed(1234 + rank)
    hr = torch.rand(BATCH, 3, CROP, CROP, generator=g)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', align_corners=False,
                                         antialias=True).clamp(0, 1)
    return lr.to(device), hr.to(device)


def roofline_pass(trainer, lr, hr, reps=2):
    """Eager pass with the library's launch profiler on: every conv kernel launch is bracketed by two
    HIP events recorded on its own stream inside libsrx_hip.so (srx_prof_*), around that one kernel."""
    from torchsr_amd import _lib
    was = trainer.use_graphs
    trainer.use_graphs = False
    trainer.gan_step(lr, hr)  # eager warm-up (repacks, allocator)
    torch.cuda.synchronize()
    _lib.call('srx_prof_start', 4096 * reps)
    try:
        for _ in range(reps):
            torch.cuda._sleep(int(2.0e8))  # let the host run ahead so that launches queue back to back
            trainer.gan_step(lr, hr)
        torch.cuda.synchronize()
    finally:
        n = _lib.lib().srx_prof_stop()
        trainer.use_graphs = was
    groups = {}
    name = C.create_string_buffer(64)
    ms, fl = C.c_float(), C.c_double()
    for i in range(n):
        _lib.call('srx_prof_get', i, name, 64, C.byref(ms), C.byref(fl))
        gsum = groups.setdefault(name.value.decode(), [0.0, 0.0, 0])
        gsum[0] += ms.value
        gsum[1] += fl.value
        gsum[2] += 1
    total_ms = sum(v[0] for v in groups.values())
    name, (ms, fl, cnt) = max(groups.items(), key=lambda kv: kv[1][0])
    achieved = fl / (ms * 1e-3) / 1e12
    table = {k: {'ms_per_step': v[0] / reps, 'gflop_per_step': v[1] / reps / 1e9, 'launches_per_step': v[2] // reps,
                 'tflops': v[1] / (v[0] * 1e-3) / 1e12} for k, v in groups.items()}
    # HBM bytes per launch of that kernel come from separate rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE,
    # tools/pmc_traffic.py); counters cannot be read from inside the process, so the committed summary is used
    traffic = None
    for f in sorted(os.listdir(os.path.join(ROOT, 'profiles'))) if os.path.isdir(os.path.join(ROOT, 'profiles')) else []:
        if f.startswith('r01_traffic') and f.endswith('.json'):
            t = json.load(open(os.path.join(ROOT, 'profiles', f)))
            if t.get('kernel') == name:
                traffic = round(t['hbm_bytes_per_launch'])
    return {
        'bound': 'mfma', 'kernel': name, 'achieved': round(achieved, 2), 'peak': PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_TFLOPS, 4), 'traffic': traffic,
        'avg_launch_us': round(ms / cnt * 1e3, 2), 'launches_per_step': cnt // reps,
        'gflop_per_launch': round(fl / cnt / 1e9, 4),
        'conv_ms_per_step': round(total_ms / reps, 3), 'by_kernel': table,
    }


def cpu_baseline(trainer, lr, hr, steps=2):
    """The oracle's GAN step on the host cores, same weights, same batch."""
    from oracle import srgan as O
    # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get('SRX_CPU_BASELINE_THREADS', '16'))))
    torch.set_num_threads(cores)
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}  # noqa: E731
    orc = O.SRGANStepOracle(cpu(trainer.generator.state_dict()), cpu(trainer.discriminator.state_dict()),
                            cpu(trainer.vgg_loss.features.state_dict()))
    lrc, hrc = lr.cpu(), hr.cpu()
    t0 = time.perf_counter()
    orc.gan_step(lrc, hrc)  # warm-up
    first = time.perf_counter() - t0
    times = []
    if first > 15.0:  # keep the default run within minutes on a slow host: the warm-up is the sample
        times, steps = [first], 0
    for _ in range(steps):
        t0 = time.perf_counter()
        orc.gan_step(lrc, hrc)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {'value': round(BATCH / med, 3), 'unit': 'crops/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{max(steps, 1)} GAN step(s) of batch {BATCH} (oracle/srgan.py, torch {torch.__version__} CPU ops), '
                      f'median {med:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-graphs', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # SRX_BENCH_ONE_GPU=1 + SRX_BENCH_BACKEND=gloo: rehearse the N>1 control flow with all ranks on one card
    dev_index = 0 if os.environ.get('SRX_BENCH_ONE_GPU') == '1' else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    distributed = world > 1
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SRX_BENCH_BACKEND', 'nccl')  # 'nccl' is RCCL on ROCm
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    import warnings
    warnings.filterwarnings('ignore', message='.*seeded random features.*')
    from torchsr_amd.srgan.trainer import SRGANTrainer

    torch.manual_seed(0)  # identical init on every rank (and an explicit broadcast in the trainer)
    targs = Namespace(disable_amp=True, batch_size=BATCH, epochs=8, gan_checkpoint=None, local_rank=local_rank,
                      pretrain_epochs=1, psnr_checkpoint=None, skip_image_save=True, world_size=world,
                      rank=rank if distributed else -1, use_graphs=not args.no_graphs)
    trainer = SRGANTrainer(device, targs, [], [], BATCH, BATCH, distributed=distributed)
    trainer.generator.train()
    trainer.discriminator.train()
    lr, hr = synth_batch(device, rank)

    for _ in range(3):  # set-up: two eager passes + hipGraph capture (not warm-up, not timed)
        trainer.gan_step(lr, hr)
    for _ in range(args.warmup):
        trainer.gan_step(lr, hr)

    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = trainer.gan_step(lr, hr)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    gen_loss = float(losses['gan/train-loss'])
    if not (gen_loss == gen_loss):
        sys.exit('bench.py: generator loss is NaN')

    if rank == 0:
        value = world * BATCH * args.steps / elapsed
        out = {
            'metric': '96x96 HR crops/sec (SRGAN GAN train step)', 'value': round(value, 2), 'unit': 'crops/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'SRGAN full GAN step (Generator + Discriminator + VGG19 perceptual loss), '
                                   '96x96 HR crops, batch 16 per GPU, fp32',
                       'per_gpu_batch': BATCH, 'global_batch': BATCH * world, 'crop': CROP,
                       'parallelism': f'dp{world}', 'hip_graph': not args.no_graphs,
                       'vgg19_weights': 'pretrained' if trainer.vgg_loss.pretrained else 'seeded-random'},
            'step_tflops': round(value * GF_PER_CROP / 1e3, 2),
            'step_frac_of_fp32_mfma_peak': round(value * GF_PER_CROP / 1e3 / (PEAK_TFLOPS * world), 4),
            'final_gen_loss': round(gen_loss, 6),
        }
        if world == 1:
            # the headline line must survive a failure of either side measurement
            if not args.no_roofline:
                try:
                    out['roofline'] = roofline_pass(trainer, lr, hr)
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: roofline pass failed: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out['roofline'] = None
            if not args.no_cpu_baseline:
                try:
                    out['cpu_baseline'] = cpu_baseline(trainer, lr, hr)
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: cpu baseline failed: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
