"""Headline benchmark: 96x96 HR crops/sec of the SRGAN GAN train step on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          (N > 1 outside torchrun: starts its own N ranks as a child process, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] / [2]): full SRGAN GAN step -- generator forward, three
discriminator forwards, discriminator backward + Adam, VGG19 perceptual loss (two forwards, one
data-gradient backward), generator backward + Adam -- on a batch of 16 synthetic 96x96 HR crops
per GPU, fp32, random-init weights (VGG19 features seeded-random: the pretrained file is not
available offline).  Inputs are resident in HBM before the timed region.  One process per GPU;
at N > 1 the gradient buckets are all-reduced over RCCL as the backward pass completes them
(weak scaling); the line then carries the process group it actually ran on.

Rank 0 prints ONE JSON line.  At N = 1 it also carries
  "parity":       the four losses of the FIRST step (seeded default-init weights) on the HIP path and on the
                  CPU oracle from the same weights and batch; ``rel`` is the generator-loss difference;
  "roofline":     the dominant kernel by device time, measured live with HIP events on the launch stream in an
                  instrumented eager pass of the same step: the MFMA FLOPs its launches EXECUTE / their
                  event-measured duration, against the 157.3 TFLOP/s fp32 MFMA peak of MI355X (``frac`` <= 1: a Winograd
                  kernel executes 16/36 of the direct convolution's multiplications; the direct-form rate it delivers is
                  ``algorithmic_tflops``, never a fraction); ``by_shape``
                  lists the layer shapes it serves, ``traffic`` = HBM bytes per launch of the heaviest one from
                  the committed rocprofv3 --pmc passes (profiles/r03_traffic.json; null if not profiled);
                  ``north_star`` = the 3x3 64->64 residual conv at 16x24x24 timed the way the step runs it,
                  as back-to-back launches inside a replayed hipGraph (HIP events on the replay stream);
  "other_configs": BASELINE.json's other configurations timed on this GPU after the headline region: the SRResNet
                  pre-training step (batch 2 = configs[0]'s shape, and batch 16), the ESRGAN GAN step (batch 16, 128x128, bf16
                  products = configs[3]) and the 1080p -> 8K generator forward in fp32 and with bf16 products (configs[4]);
                  each with its EXECUTED algorithmic GFLOP, the fraction of the matching MFMA peak and its dominant conv kernel;
  "dp_rehearsal": the data-parallel form of the headline step (seven hipGraph segments, four gradient buckets all-reduced
                  asynchronously) on a real RCCL process group at world size 1, in a child process, next to the single-graph
                  step in the same process: what the segmentation costs on one GPU.  SRX_BENCH_FORCE_DIST=1 takes that path
                  directly (also under `torchrun --nproc-per-node 1`).  Scaling to N > 1 is not measured by this file's author;
  "cpu_baseline": the CPU oracle (oracle/srgan.py, stock torch ops) running the identical step on the host
                  cores, BASELINE.md section 5 protocol (3 warm-up + 10 timed steps, median), plus the
                  config-1 leg (pre-training step, batch 2) -- a reported baseline, not the target.
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
NORTH_STAR_GF = 0.6795  # 3x3 64->64 conv at 16x24x24: 2 * 9216 * 64 * 576
WINO_EXECUTED = 16.0 / 36.0  # Winograd F(2x2, 3x3): multiplications executed per multiplication of the direct form


def executed_ratio(kernel):
    """MFMA FLOPs a kernel EXECUTES per algorithmic (direct-convolution) FLOP the library's profiler books for it: the Winograd
    kernels (csrc/wino.hip) multiply 16 times where the direct form multiplies 36 times, every other kernel runs the direct count.
    Every ``frac`` of this file is executed FLOPs / time / peak (a share of the matrix pipe, <= 1); the direct-form rate is
    carried beside it as ``algorithmic_tflops``."""
    return WINO_EXECUTED if kernel.startswith('wino_') else 1.0


def rates(kernel, flops, ms, peak):
    """{'tflops' (executed), 'frac' (executed / peak), 'algorithmic_tflops'[, 'algorithmic_speedup']} of `flops` algorithmic
    FLOPs in `ms` milliseconds on `kernel`."""
    alg = flops / (ms * 1e-3) / 1e12
    r = executed_ratio(kernel)
    out = {'tflops': round(alg * r, 2), 'frac': round(alg * r / peak, 4), 'algorithmic_tflops': round(alg, 2)}
    if r != 1.0:
        out['algorithmic_speedup'] = round(1.0 / r, 3)
    return out


def synth_batch(device, rank, batch=BATCH):
    g = torch.Generator().manual_seed(1234 + rank)
    hr = torch.rand(batch, 3, CROP, CROP, generator=g)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', align_corners=False,
                                         antialias=True).clamp(0, 1)
    return lr.to(device), hr.to(device)


def cpu_states(trainer):
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}  # noqa: E731
    return cpu(trainer.generator.state_dict()), cpu(trainer.discriminator.state_dict()), \
        cpu(trainer.vgg_loss.features.state_dict())


def host_threads():
    """Host cores this process may use: the GPU box gives one GPU's share of the machine (cgroup / affinity),
    ``os.cpu_count()`` reports the whole machine."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, when there is one
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('SRX_CPU_BASELINE_THREADS', '64'))))


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def parity_check(states, lr, hr, gpu_losses):
    """First GAN step from the same weights and batch on the CPU oracle vs what the HIP path just computed."""
    from oracle import srgan as O
    torch.set_num_threads(host_threads())
    orc = O.SRGANStepOracle(*states)
    want = orc.gan_step(lr.cpu(), hr.cpu())
    keys = ('gan/disc-loss', 'gan/content-loss', 'gan/adversarial-loss', 'gan/train-loss')
    got = [float(gpu_losses[k]) for k in keys]
    rels = [abs(g - w) / max(abs(w), 1e-3) for g, w in zip(got, want)]
    return {'gen_loss_gpu': got[3], 'gen_loss_oracle': want[3], 'rel': rels[3], 'max_rel_of_4_losses': max(rels),
            'losses_gpu': got, 'losses_oracle': list(want), 'tolerance': 1e-3, 'ok': max(rels) < 1e-3,
            'what': 'first GAN step, batch 16, seeded default init, oracle/srgan.py on the host',
            'scope': 'this check: an unscreened synthetic batch, the four losses at 1e-3.  The test suite pins more (tests/test_*_gpu.py): '
                     'reference fixtures whose INPUTS were screened for a margin from activation kinks (oracle/gen_golden.py: '
                     'widest_margin_seed) at 1e-3 / digests / sampled elements; unscreened inputs against the oracle only, at looser '
                     'bounds (test_generator_vs_oracle_fresh_inputs)'}


def north_star_in_graph(device, reps=66, replays=20):
    """The north-star kernel the way the step runs it: `reps` back-to-back launches (the 66 per step) inside one
    hipGraph, replayed; HIP events on the replay stream around each replay."""
    from torchsr_amd.layers import Conv2d
    torch.manual_seed(3)
    conv = Conv2d(64, 64, 3, 1, 1, bias=False).to(device)
    x = torch.rand(16, 24, 24, 64, device=device)
    with torch.no_grad():
        conv(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = x
            for _ in range(reps):
                y = conv(x)  # same input: every launch does identical work, launches stay dependent on the stream
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / reps)
    times.sort()
    us = times[len(times) // 2]
    tf = NORTH_STAR_GF / us * 1e3
    return {'kernel': 'rt36_conv3x3_c64_kernel<1>', 'shape': '3x3 64->64 @16x24x24 (M 9216, N 64, K 576)',
            'launches_per_replay': reps, 'us_per_launch_in_graph': round(us, 3), 'tflops': round(tf, 2),
            'frac': round(tf / PEAK_TFLOPS, 4), 'algorithmic_tflops': round(tf, 2), 'gflop_per_launch': NORTH_STAR_GF,
            'note': 'includes the ~1.5 us kernel boundary between dependent launches'}


def dominant_shape_in_graph(device, reps=20, replays=10):
    """The dominant kernel's top layer shape (VGG19 block 3, 3x3 256->256 + bias + ReLU on source and target as one batch of
    32 at 24x24: M 18432, N 256, K 2304) as `reps` back-to-back launches inside a replayed hipGraph -- the per-launch time
    without the dispatch gaps that the eager event pairs of `roofline_pass` include."""
    from torchsr_amd.layers import ACT_RELU, Conv2d
    torch.manual_seed(4)
    conv = Conv2d(256, 256, 3, 1, 1, act=ACT_RELU).to(device)
    x = torch.rand(32, 24, 24, 256, device=device)
    with torch.no_grad():
        conv(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                conv(x)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    times = []
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e3 / reps)
    times.sort()
    us = times[len(times) // 2]
    gf = 2.0 * 18432 * 256 * 2304 / 1e9
    from torchsr_amd import functional as F
    st = conv._st
    kern = 'wino_kernel' if st._descs and F.wino_layer_ok(st, next(iter(st._descs.values()))) else 'gconv_kernel'
    return {'MxNxK': '18432x256x2304', 'what': 'VGG19 block 3 (3x3 256->256 + bias + ReLU, batch 32 at 24x24) on the kernel the step runs it on '
                                            '(Winograd F(2x2,3x3) since round 5; SRX_NO_WINO=1: the direct gather-GEMM)',
            'kernel': kern, 'launches_per_replay': reps, 'us_per_launch_in_graph': round(us, 2),
            'gflop_per_launch': round(gf, 3), **rates(kern, gf * 1e9, us * 1e-3, PEAK_TFLOPS)}


def prof_tables(step_fn, reps=2, slots=4096):
    """Run ``step_fn`` eagerly ``reps`` times with the library's launch profiler on: every conv kernel launch is
    bracketed by two HIP events recorded on its own stream inside libsrx_hip.so (srx_prof_*), around that one kernel.
    Returns ({'kernel MxNxK=..': [ms, flops, launches]}, {'kernel': [ms, flops, launches]}) summed over the reps."""
    from torchsr_amd import _lib
    step_fn()  # eager warm-up (repacks, allocator)
    torch.cuda.synchronize()
    _lib.call('srx_prof_start', slots * reps)
    try:
        for _ in range(reps):
            torch.cuda._sleep(int(2.0e8))  # let the host run ahead so that launches queue back to back
            step_fn()
        torch.cuda.synchronize()
    finally:
        n = _lib.lib().srx_prof_stop()
    pairs, kernels = {}, {}
    name = C.create_string_buffer(112)
    ms, fl = C.c_float(), C.c_double()
    for i in range(n):
        _lib.call('srx_prof_get', i, name, 112, C.byref(ms), C.byref(fl))
        full = name.value.decode()
        for table, key in ((pairs, full), (kernels, full.split(' MxNxK=')[0])):
            gsum = table.setdefault(key, [0.0, 0.0, 0])
            gsum[0] += ms.value
            gsum[1] += fl.value
            gsum[2] += 1
    return pairs, kernels


def dominant_of(step_fn, peak_tflops, reps=2, slots=4096):
    """The conv kernel with the most device time per call of ``step_fn`` (other_configs legs)."""
    _pairs, kernels = prof_tables(step_fn, reps, slots)
    if not kernels:
        return None
    kname, (ms, fl, cnt) = max(kernels.items(), key=lambda kv: kv[1][0])
    return {'kernel': kname, 'ms': round(ms / reps, 3), 'launches': cnt // reps, 'avg_launch_us': round(ms / cnt * 1e3, 2),
            **rates(kname, fl, ms, peak_tflops),
            'timing': 'eager event pair per launch (dispatch gaps included: the sum can exceed the replayed wall time by ~10 %)',
            'conv_kernel_ms': round(sum(v[0] for v in kernels.values()) / reps, 3),
            'conv_launches': sum(v[2] for v in kernels.values()) // reps,
            # algorithmic GFLOP per call that the Winograd kernels do NOT multiply (their launches' direct-form count x 20/36)
            'gflop_not_executed': round(sum(v[1] * (1.0 - executed_ratio(k)) for k, v in kernels.items()) / reps / 1e9, 3)}


def roofline_pass(trainer, lr, hr, reps=2):
    """The headline step's dominant kernel (see ``prof_tables``)."""
    # the timed step runs the perceptual-loss branch next to the discriminator's on a second graph branch; an event pair around a
    # launch that shares the chip with the other branch's launches times the pair of them, so the instrumented pass runs the
    # SAME launches on one stream (overlap_branches = False) and each duration is that kernel's alone
    was = trainer.use_graphs, trainer.overlap_branches
    trainer.use_graphs, trainer.overlap_branches = False, False
    try:
        pairs, kernels = prof_tables(lambda: trainer.gan_step(lr, hr), reps)
    finally:
        trainer.use_graphs, trainer.overlap_branches = was
    total_ms = sum(v[0] for v in kernels.values())
    if os.environ.get('SRX_BENCH_SHAPES'):  # developer aid: the full (kernel, shape) table
        with open(os.environ['SRX_BENCH_SHAPES'], 'w') as f:
            for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
                f.write(f'{v[0] / reps * 1e3:9.1f} us/step {v[2] // reps:3d} launches {v[0] / v[2] * 1e3:8.1f} us each '
                        f'{v[1] / (v[0] * 1e-3) / 1e12:6.1f} TF/s  {k}\n')
    # the dominant kernel = the kernel (template instance) with the most device time per step, over all the layer
    # shapes it serves; its shapes are listed one by one, each with the HBM traffic of the committed --pmc passes
    kname, (ms, fl, cnt) = max(kernels.items(), key=lambda kv: kv[1][0])
    top_rates = rates(kname, fl, ms, PEAK_TFLOPS)
    table = {k: {'ms_per_step': round(v[0] / reps, 4), 'gflop_per_step': round(v[1] / reps / 1e9, 3),
                 'launches_per_step': v[2] // reps, **rates(k, v[1], v[0], PEAK_TFLOPS)}
             for k, v in sorted(kernels.items(), key=lambda kv: -kv[1][0])}
    saved_gf = sum(v[1] * (1.0 - executed_ratio(k)) for k, v in kernels.items()) / reps / 1e9
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE x2, WRITE_SIZE: tools/pmc_run.sh) of
    # tools/pmc_workloads.py restricted to ONE layer shape; counters cannot be read from inside the process
    measured = {}
    for tname in ('r03_traffic.json', 'r04_traffic.json', 'r05_traffic.json', 'r06_traffic.json'):  # (later rounds add shapes / supersede entries)
        tpath = os.path.join(ROOT, 'profiles', tname)
        if os.path.exists(tpath):
            measured.update(json.load(open(tpath)))
    shapes = []
    for full, v in sorted(pairs.items(), key=lambda kv: -kv[1][0]):
        if full.split(' MxNxK=')[0] != kname:
            continue
        ent = measured.get(full) or {}
        shapes.append({'MxNxK': full.split(' MxNxK=')[1], 'launches_per_step': v[2] // reps,
                       'avg_launch_us': round(v[0] / v[2] * 1e3, 2), **rates(kname, v[1], v[0], PEAK_TFLOPS),
                       'traffic': round(ent['hbm_bytes_per_launch']) if ent else None,
                       'algorithmic_bytes': ent.get('algorithmic_bytes_per_launch')})
    top = shapes[0] if shapes else {}
    wino = executed_ratio(kname) != 1.0
    out = {
        # `achieved` / `frac`: the MFMA FLOPs the dominant kernel EXECUTES per second, against the fp32 MFMA peak -- the share of the
        # matrix pipe it uses (<= 1).  A Winograd F(2x2, 3x3) kernel executes 16 multiplications where the direct convolution
        # executes 36, so the rate of the convolution it delivers (`algorithmic_tflops` = 2 x M x N x K / time, what every
        # direct kernel's `tflops` is too) is 2.25 x its executed rate and may exceed the peak; it is never called a fraction
        'bound': 'mfma', 'kernel': kname, 'achieved': top_rates['tflops'], 'peak': PEAK_TFLOPS, 'unit': 'TFLOP/s',
        'frac': top_rates['frac'],
        'algorithm': 'Winograd F(2x2,3x3), fp32' if wino else 'direct gather-GEMM, fp32',
        'algorithmic_tflops': top_rates['algorithmic_tflops'],
        'algorithmic_speedup': round(1.0 / executed_ratio(kname), 3),
        'executed_over_algorithmic_flops': round(executed_ratio(kname), 4),
        'timing': 'eager pass on ONE stream (the timed step overlaps its two branches; here every launch has the chip alone), one HIP event pair per launch on the launch stream (srx_prof_*): dispatch gaps included, a few per cent pessimistic against the replayed graph',
        'step_gflop_algorithmic': round(GF_PER_CROP * BATCH, 2),
        'step_gflop_executed': round(GF_PER_CROP * BATCH - saved_gf, 2),

        'traffic': top.get('traffic'), 'traffic_shape_MxNxK': top.get('MxNxK'),
        'algorithmic_bytes_per_launch': top.get('algorithmic_bytes'),
        'avg_launch_us': round(ms / cnt * 1e3, 2), 'launches_per_step': cnt // reps,
        'gflop_per_launch': round(fl / cnt / 1e9, 4),
        'conv_ms_per_step': round(total_ms / reps, 3), 'by_shape': shapes, 'by_kernel': table,
    }
    return out


PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
ESRGAN_EXECUTED_GF = 3034.6  # per batch-16 step: SURVEY.md 8d's 3622 GF minus the second generator forward (587.43),
#                              which recomputes the first bit for bit and is not run (torchsr_amd/esrgan/trainer.py)
INFER_GF = 9199.0  # SRGAN generator forward, 1080x1920 -> 4320x7680


def _targs(batch, amp, **extra):
    return Namespace(disable_amp=not amp, batch_size=batch, epochs=8, gan_checkpoint=None, local_rank=0, pretrain_epochs=1,
                     psnr_checkpoint=None, skip_image_save=True, world_size=1, rank=-1, use_graphs=True, vgg_weights='random',
                     **extra)


def _timed(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def _crops(device, n, crop, seed):
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(n, 3, crop, crop, generator=g)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', align_corners=False, antialias=True).clamp(0, 1)
    return lr.to(device), hr.to(device)


def leg_rates(alg_gf, dt, peak, dom):
    """Rates of one configuration's step: `alg_gf` algorithmic GFLOP (2 x MACs of conv + linear, direct form) in `dt` seconds.
    ``tflops`` / ``frac`` count the MFMA FLOPs the step EXECUTES (the algorithmic count less what its Winograd launches do not
    multiply, from the same eager profiler pass that names the dominant kernel) -- a share of the matrix pipe, <= 1."""
    saved = (dom or {}).get('gflop_not_executed', 0.0)
    ex = alg_gf - saved
    out = {'algorithmic_gflop': round(alg_gf, 2), 'executed_gflop': round(ex, 2), 'tflops': round(ex / dt / 1e3, 2),
           'peak_tflops': peak, 'frac': round(ex / dt / 1e3 / peak, 4), 'algorithmic_tflops': round(alg_gf / dt / 1e3, 2)}
    return out


def other_configs(device):
    """BASELINE.json's other configurations, each timed on this GPU after the headline region (N = 1 only): wall time
    per step over hipGraph replays with a synchronize on both sides, inputs resident in HBM, the EXECUTED algorithmic
    GFLOP (2 x MACs of conv + linear), the fraction of the MFMA peak of the arithmetic the configuration asks for, and
    the dominant conv kernel from an eager ``srx_prof_*`` pass.  A leg that fails reports its error instead."""
    from torchsr_amd.esrgan.trainer import ESRGANTrainer
    from torchsr_amd.srgan.generator import Generator
    from torchsr_amd.srgan.trainer import SRGANTrainer
    from torchsr_amd.test import upscale
    out = {}

    def leg(name, fn):
        try:
            out[name] = fn()
        except Exception as exc:  # noqa: BLE001
            print(f'bench.py: other_configs[{name}] failed: {type(exc).__name__}: {exc}', file=sys.stderr)
            out[name] = {'error': f'{type(exc).__name__}: {exc}'}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def pretrain(b):
        torch.manual_seed(0)
        t = SRGANTrainer(device, _targs(b, False), [], [], b, b)
        lr, hr = _crops(device, b, 96, 77)
        dt = _timed(lambda: t.pretrain_step(lr, hr), 50, 6)
        gf = 7.648 * b
        t.use_graphs = False
        dom = dominant_of(lambda: t.pretrain_step(lr, hr), PEAK_TFLOPS)
        return {'workload': f'SRGAN SRResNet pre-training step (BASELINE configs[0] shape), 96x96 crops, batch {b}, fp32',
                'ms': round(dt * 1e3, 3), 'crops_per_s': round(b / dt, 1), **leg_rates(gf, dt, PEAK_TFLOPS, dom),
                'dtype': 'f32', 'dominant_kernel': dom}

    def esrgan():
        torch.manual_seed(0)
        t = ESRGANTrainer(device, _targs(16, True), [], [], 16, 16)
        lr, hr = _crops(device, 16, 128, 78)
        dt = _timed(lambda: t.gan_step(lr, hr), 15, 5)
        gf = ESRGAN_EXECUTED_GF
        t.use_graphs = t.overlap_branches = False  # the instrumented pass on one stream: every launch timed alone (see roofline_pass)
        dom = dominant_of(lambda: t.gan_step(lr, hr), PEAK_BF16_TFLOPS, reps=1, slots=8192)
        return {'workload': 'ESRGAN full GAN step (23-RRDB generator + relativistic discriminator + VGG19), 128x128 crops, '
                            'batch 16, bf16 products / fp32 accumulate (BASELINE configs[3])',
                'ms': round(dt * 1e3, 3), 'crops_per_s': round(16 / dt, 1), **leg_rates(gf, dt, PEAK_BF16_TFLOPS, dom),
                'reference_executes_gflop': 3622.0, 'dtype': 'bf16 products, f32 accumulate',
                'dominant_kernel': dom}

    def infer(precision):
        torch.manual_seed(0)
        gen = Generator().to(device).eval()
        lr = torch.rand(1, 3, 1080, 1920, device=device)
        kw = {} if precision == 'fp32' else {'precision': precision}
        dt = _timed(lambda: upscale(gen, lr, **kw), 5, 2)
        peak = PEAK_TFLOPS if precision == 'fp32' else PEAK_BF16_TFLOPS
        dom = dominant_of(lambda: upscale(gen, lr, **kw), peak, reps=1, slots=8192)
        return {'workload': f'SRGAN generator 1920x1080 -> 7680x4320, batch 1, eval mode, tiled, {precision} '
                            '(BASELINE configs[4])',
                'ms_per_image': round(dt * 1e3, 2), **leg_rates(INFER_GF, dt, peak, dom),
                'dtype': 'f32' if precision == 'fp32' else 'bf16 products, f32 accumulate', 'dominant_kernel': dom}

    leg('srgan_pretrain_b2', lambda: pretrain(2))
    leg('srgan_pretrain_b16', lambda: pretrain(16))
    leg('esrgan_gan_b16_bf16', esrgan)
    leg('infer_1080p_fp32', lambda: infer('fp32'))
    leg('infer_1080p_bf16', lambda: infer('bf16'))
    return out


def _free_port():
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def self_launch(args, argv):
    """``python bench.py --gpus N`` with N > 1 and no torchrun environment: this process touches NO GPU call; it starts
    ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...``
    as a CHILD (never exec: a process must not be replaced once anything may have initialised the GPU), relays the rank-0
    JSON line and returns the child's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print('bench.py: launching ' + ' '.join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:  # the ranks' other output goes on to stderr; the one JSON line is relayed on stdout
        if ln.startswith('{') and '"metric"' in ln:
            line = ln.rstrip('\n')
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print('bench.py: the ranks exited 0 without a JSON line', file=sys.stderr)
        rc = 1
    return rc


def _dp_child(extra_env, steps=40, timeout_s=240):
    """One run of the data-parallel form of the step at world size 1 on RCCL, in a child process (its own process group and
    its own plans: ``SRX_RESERVED_CUS`` is read when the library loads).  Returns the child's ``dp`` object."""
    import subprocess
    env = dict(os.environ, SRX_BENCH_FORCE_DIST='1', **extra_env)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', '5', '--no-parity', '--no-roofline',
           '--no-cpu-baseline', '--no-other-configs']
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
    if res.returncode != 0:
        raise RuntimeError(f'child exited {res.returncode}: {res.stderr[-400:]}')
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]
    return json.loads(line)['dp']


def dp_rehearsal(contention=(16,)):
    """The data-parallel form of the step on RCCL at world size 1, in a child process (a hang in there cannot take the
    headline with it): ``SRX_BENCH_FORCE_DIST=1 python bench.py``.  Returns the child's ``dp`` object -- backend, buckets,
    segmented vs fused ms per step on the same GPU -- plus ``cu_contention``: the segmented step while a side-stream kernel
    HOLDS k compute units (``srx_occupy_cus``: k workgroups that each claim a CU's whole LDS, so nothing else becomes
    resident there -- the worst case for RCCL's channel workgroups, which in fact co-reside and mostly wait), with plans cut
    for the whole chip and with plans told about the k CUs (``SRX_RESERVED_CUS``).  The one-GPU estimate of what the
    all-reduce's channel kernels cost grids that are cut for exactly 256 CUs; N > 1 itself stays unmeasured.  The default run
    holds 16 CUs (two child processes); ``--dp-contention 8,16,32`` gives the table DESIGN.md section 6 quotes."""
    dp = _dp_child({})
    table = []
    for k in contention:
        row = {'cus_held': k}
        for name, env in (('plans_for_256', {'SRX_BENCH_OCCUPY_CUS': str(k)}),
                          ('plans_for_free_cus', {'SRX_BENCH_OCCUPY_CUS': str(k), 'SRX_RESERVED_CUS': str(k)})):
            try:
                child = _dp_child(env, steps=30)
                row[name + '_ms'] = child['segmented_ms_per_step']
                row.setdefault('plan_cus', {})[name] = child.get('plan_cus')
            except Exception as exc:  # noqa: BLE001
                row[name + '_ms'] = None
                row[name + '_error'] = f'{type(exc).__name__}: {str(exc)[-200:]}'
        # the same k CUs held only while a large gradient bucket would be on the wire (from the launch() of the 75.5 MB
        # classifier bucket to the wait() in front of the discriminator's optimiser: D's conv backward + the VGG19 forward),
        # with plans that know nothing and with plans that leave k CUs free INSIDE that window only (GradBuckets.reserve_cus)
        for name, env in (('window_plans_for_256', {'SRX_BENCH_WINDOW_OCCUPY_CUS': str(k)}),
                          ('window_plans_aware', {'SRX_BENCH_WINDOW_OCCUPY_CUS': str(k), 'SRX_BENCH_WINDOW_RESERVED_CUS': str(k)})):
            try:
                child = _dp_child(env, steps=30)
                row[name + '_ms'] = child['segmented_ms_per_step']
            except Exception as exc:  # noqa: BLE001
                row[name + '_ms'] = None
                row[name + '_error'] = f'{type(exc).__name__}: {str(exc)[-200:]}'
        table.append(row)
    dp['cu_contention'] = {'unheld_segmented_ms': dp['segmented_ms_per_step'], 'rows': table,
                           'how': 'k whole CUs held by srx_occupy_cus on a side stream: for the whole timed region (plans_*: an upper '
                                  'bound nobody reaches -- RCCL occupies channels only while a bucket is in flight) and only inside '
                                  'the communication window of the large discriminator buckets (window_*)'}
    return dp


class _WindowHolder:
    """Rehearsal stand-in for RCCL's channel kernels (``GradBuckets(window_hook=...)``): holds k compute units on a side stream
    from the launch() of a large gradient bucket to the wait() that closes the window -- not for the whole timed region."""

    def __init__(self, device, k, whole_cu):
        self.k, self.whole_cu = k, whole_cu
        # (high priority: HIP maps streams onto a few hardware queues and packets of one queue run in order -- on a queue
        # shared with RCCL's stream the holder kept the all-reduce kernel, and through its wait() the whole step, waiting for
        # the holder's 50 ms deadline: 58 ms per step, measured; streams of another priority take another queue)
        self.side = torch.cuda.Stream(device, priority=-1)
        self.flags = [torch.zeros(1, dtype=torch.int32, device=device) for _ in range(2)]
        self.i, self.windows = 0, 0
        self.ev = torch.cuda.Event()

    def begin(self):
        from torchsr_amd import _lib
        flag = self.flags[self.i]
        self.ev.record()  # the window opens where the compute stream is now
        with torch.cuda.stream(self.side):
            flag.zero_()
            self.side.wait_event(self.ev)
            _lib.call('srx_occupy_cus', self.k, self.whole_cu, flag.data_ptr(), 50, self.side.cuda_stream)  # bounded: 50 ms
        self.windows += 1

    def end(self):
        self.flags[self.i].fill_(1)  # on the compute stream: the holder exits when the compute stream gets here
        self.i ^= 1


def _window_holder(device):
    k = int(os.environ.get('SRX_BENCH_WINDOW_OCCUPY_CUS', '0'))
    return _WindowHolder(device, k, int(os.environ.get('SRX_BENCH_OCCUPY_WHOLE_CU', '1'))) if k > 0 else None


def _plan_cus():
    from torchsr_amd import _lib
    return int(_lib.lib().srx_plan_cus())


def cpu_baseline(states, lr, hr, warmup=3, steps=10, budget_s=60.0):
    """The oracle's steps on the host cores, same weights, same batch (3 warm-up + 10 timed
    steps, median).  A slow host cuts the timed steps short at ``budget_s`` seconds and says so."""
    from oracle import srgan as O
    cores = host_threads()
    torch.set_num_threads(cores)
    lrc, hrc = lr.cpu(), hr.cpu()

    def timed(fn):
        t_begin = time.perf_counter()
        for _ in range(warmup):
            fn()
            if time.perf_counter() - t_begin > budget_s / 2:
                break
        times = []
        t_begin = time.perf_counter()
        for _ in range(steps):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s:
                break
        times.sort()
        return times[len(times) // 2], len(times)

    orc = O.SRGANStepOracle(*states)
    med, n = timed(lambda: orc.gan_step(lrc, hrc))
    orc2 = O.SRGANStepOracle(*states)
    med2, n2 = timed(lambda: orc2.pretrain_step(lrc[:2], hrc[:2]))
    return {'value': round(BATCH / med, 3), 'unit': 'crops/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} timed GAN steps of batch {BATCH} after {warmup} warm-up (oracle/srgan.py, torch {torch.__version__} '
                      f'CPU ops), median {med:.3f} s/step',
            'cpu_model': cpu_model(), 'os_cpu_count': os.cpu_count(), 'threads': torch.get_num_threads(),
            'config1_pretrain_b2': {'value': round(2 / med2, 3), 'unit': 'crops/s',
                                    'sample': f'{n2} timed pre-training steps of batch 2, median {med2 * 1e3:.1f} ms/step'}}


def assert_fracs(obj, path='line'):
    """Every ``frac`` of the line is a share of a hardware peak: none may exceed 1."""
    if isinstance(obj, dict):
        for k, v in obj.items():
            if 'frac' in k.split('_') and not isinstance(v, (dict, list, str)):
                assert v is None or 0.0 <= v <= 1.0, f'{path}.{k} = {v}: not a fraction of a peak'
            else:
                assert_fracs(v, f'{path}.{k}')
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            assert_fracs(v, f'{path}[{i}]')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-graphs', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--no-dp-rehearsal', action='store_true')
    ap.add_argument('--dp-contention', default='16', help='CUs held during the data-parallel rehearsal, comma separated (default 16)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))  # before anything touches the GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # SRX_BENCH_ONE_GPU=1 + SRX_BENCH_BACKEND=gloo: rehearse the N>1 control flow with all ranks on one card
    dev_index = 0 if os.environ.get('SRX_BENCH_ONE_GPU') == '1' else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    # SRX_BENCH_FORCE_DIST=1: take the data-parallel (segmented, bucketed) path at world size 1 too, on a real RCCL
    # process group, with the all-reduces issued -- `torchrun --nproc-per-node 1 bench.py` or a plain `python bench.py`
    forced = world == 1 and os.environ.get('SRX_BENCH_FORCE_DIST') == '1'
    distributed = world > 1 or forced
    if forced:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(_free_port())
    comm = None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('SRX_BENCH_BACKEND', 'nccl')  # 'nccl' is RCCL on ROCm
        # at world > 1: bound RCCL's channel count and tell the launch plans how many CUs the channel workgroups get while a
        # large gradient bucket is on the wire -- decided here, BEFORE the process group exists, and printed in `config`
        from torchsr_amd.ddp import configure_comm
        comm = configure_comm(world, backend)
        if forced and os.environ.get('SRX_BENCH_WINDOW_RESERVED_CUS'):  # one-GPU rehearsal of the windowed reservation
            comm['reserved_cus_in_comm_window'] = int(os.environ['SRX_BENCH_WINDOW_RESERVED_CUS'])
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from torchsr_amd.ddp import describe_group
    from torchsr_amd.srgan.trainer import SRGANTrainer

    torch.manual_seed(0)  # identical init on every rank (and an explicit broadcast in the trainer)
    targs = Namespace(disable_amp=True, batch_size=BATCH, epochs=8, gan_checkpoint=None, local_rank=local_rank,
                      pretrain_epochs=1, psnr_checkpoint=None, skip_image_save=True, world_size=world,
                      rank=rank if distributed else -1, use_graphs=not args.no_graphs, vgg_weights='random',
                      force_collectives=forced, comm_reserved_cus=comm['reserved_cus_in_comm_window'] if comm else 0,
                      comm_window_hook=_window_holder(device))
    trainer = SRGANTrainer(device, targs, [], [], BATCH, BATCH, distributed=distributed)
    trainer.generator.train()
    trainer.discriminator.train()
    lr, hr = synth_batch(device, rank)

    single = world == 1 and not forced  # the side measurements of the default one-GPU run
    want_parity = single and not args.no_parity
    states0 = cpu_states(trainer) if (single and not (args.no_parity and args.no_cpu_baseline)) else None
    first = trainer.gan_step(lr, hr)  # set-up step 1 of 3 (eager); its losses feed the parity check
    first = {k: float(v) for k, v in first.items()} if want_parity else None
    for _ in range(2):  # set-up: second eager pass + hipGraph capture (not warm-up, not timed)
        trainer.gan_step(lr, hr)
    for _ in range(args.warmup):
        trainer.gan_step(lr, hr)

    # SRX_BENCH_OCCUPY_CUS=k (dp_rehearsal's contention legs): k whole CUs are held by a side-stream kernel for the timed region
    occupy = int(os.environ.get('SRX_BENCH_OCCUPY_CUS', '0'))
    side, stop_flag = None, None
    if occupy > 0:
        from torchsr_amd import _lib
        torch.cuda.synchronize()
        stop_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        side = torch.cuda.Stream()
        _lib.call('srx_occupy_cus', occupy, 1, stop_flag.data_ptr(), 20000, side.cuda_stream)  # launched on an idle chip
        time.sleep(0.05)                                                                        # ... and resident before the steps

    if distributed:
        dist.barrier()
    if side is None:
        torch.cuda.synchronize()
    else:
        torch.cuda.current_stream().synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = trainer.gan_step(lr, hr)
    torch.cuda.current_stream().synchronize()
    own_elapsed = time.perf_counter() - t0  # this rank's steps alone, before it waits for the others
    if side is not None:
        stop_flag[0] = 1  # the holder kernel sees the flag and exits (or its own deadline ends it)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = own_elapsed if side is not None else time.perf_counter() - t0
    rank_ms = None
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([own_elapsed / args.steps * 1e3], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per = [float(v.item()) for v in every]
        rank_ms = {'min': round(min(per), 3), 'max': round(max(per), 3), 'per_rank': [round(v, 3) for v in per],
                   'what': "each rank's own ms per step (host clock around its timed steps, before the closing barrier)"}
    gen_loss = float(losses['gan/train-loss'])
    if not (gen_loss == gen_loss):
        sys.exit('bench.py: generator loss is NaN')

    # spread of the step time (outside the timed region): per-step HIP events on the step's stream
    spread = None
    dp = None
    if distributed:
        # the same step as ONE hipGraph without cuts, buckets or collectives, same process, same GPU: at world size 1 what the
        # segmentation itself costs (the ceiling on multi-GPU efficiency before any wire time); at N > 1 the difference to the
        # timed data-parallel step is everything the exchange costs this rank -- segmentation, CUs lent to RCCL's channels and
        # the time the compute stream stands in a bucket's wait() -- so a first real N > 1 run diagnoses itself
        torch.manual_seed(0)
        fargs = Namespace(**{**vars(targs), 'rank': -1, 'force_collectives': False})
        fused = SRGANTrainer(device, fargs, [], [], BATCH, BATCH, distributed=False)
        fused.generator.train()
        fused.discriminator.train()
        for _ in range(3 + args.warmup):
            fused.gan_step(lr, hr)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fused.gan_step(lr, hr)
        torch.cuda.synchronize()
        fused_ms = (time.perf_counter() - t1) / args.steps * 1e3
        seg_ms = elapsed / args.steps * 1e3
        del fused
        torch.cuda.empty_cache()
        # per bucket, a few extra steps with HIP events at launch() / in front of wait() / behind it (outside the timed region)
        for sync in (trainer.gen_sync, trainer.disc_sync):
            sync.timing = True
        for _ in range(8):
            trainer.gan_step(lr, hr)
        buckets = {}
        for name, sync in (('generator', trainer.gen_sync), ('discriminator', trainer.disc_sync)):
            sync.timing = False
            marks = sync.timings()
            by_size = {}
            for m in marks:
                by_size.setdefault(m['bytes'], []).append(m)
            med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
            buckets[name] = [{'bytes': nb, 'launches_timed': len(ms), 'launch_to_wait_ms': round(med([m['launch_to_wait_ms'] for m in ms]), 4),
                              'exposed_wait_ms': round(med([m['exposed_wait_ms'] for m in ms]), 4)} for nb, ms in sorted(by_size.items())]
        exposed = sum(b['exposed_wait_ms'] for bs in buckets.values() for b in bs)
        dp = {'process_group': describe_group(), 'collectives_issued_per_step': 4,
              'exposed_comm_ms': round(seg_ms - fused_ms, 3),
              'exposed_comm_what': 'data-parallel step minus the single-graph step without collectives, same process and GPU (max over ranks '
                                   'for the former at N > 1): segmentation + CUs lent to the channels + waits',
              'bucket_timings': buckets, 'sum_of_exposed_waits_ms': round(exposed, 4),
              'collectives_issued_total': trainer.gen_sync.issued + trainer.disc_sync.issued,
              'grad_buckets': {'generator': len(trainer.gen_sync), 'discriminator': len(trainer.disc_sync)},
              'bucket_bytes': {'generator': [int(x.numel()) * 4 for x in trainer.gen_sync.slices],
                               'discriminator': [int(x.numel()) * 4 for x in trainer.disc_sync.slices]},
              'graph_segments': sorted(k for k in trainer._graphs), 'hip_graph': trainer.use_graphs,
              'cus_held': occupy, 'plan_cus': _plan_cus(), 'comm': comm,
              'cus_held_in_comm_windows': int(os.environ.get('SRX_BENCH_WINDOW_OCCUPY_CUS', '0')),
              'comm_windows_opened': getattr(targs.comm_window_hook, 'windows', None),
              'segmented_ms_per_step': round(seg_ms, 3), 'fused_ms_per_step': round(fused_ms, 3),
              'segmentation_overhead': round(seg_ms / fused_ms - 1.0, 4), 'steps': args.steps,
              'what': ('SRGAN GAN step, batch 16, world size 1 on RCCL: 7 hipGraph segments + 4 async all-reduces '
                       '(no-op sums) vs the single-graph step; unmeasured at N > 1') if world == 1 else
                      f'SRGAN GAN step, batch 16 per GPU, world size {world}: the timed data-parallel step vs the single-graph step '
                      'without collectives on this rank'}
    if rank == 0 and single:
        evs = []
        for _ in range(min(args.steps, 50)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            trainer.gan_step(lr, hr)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        spread = {'p10': round(ts[len(ts) // 10], 3), 'median': round(ts[len(ts) // 2], 3),
                  'p90': round(ts[(len(ts) * 9) // 10], 3), 'samples': len(ts), 'how': 'HIP events around single steps'}

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
                       'vgg19_weights': 'pretrained' if trainer.vgg_loss.pretrained else 'seeded-random',
                       'process_group': describe_group(),
                       'grad_buckets': ({'generator': len(trainer.gen_sync), 'discriminator': len(trainer.disc_sync)}
                                        if distributed else None),
                       'plan_cus': _plan_cus(), 'comm': comm},
            'per_rank_ms_per_step': rank_ms,
            # the convolution work the step DELIVERS per second (direct-form count); the share of the matrix
            # pipe the step uses is roofline.step_frac_of_fp32_mfma_peak (executed FLOPs: Winograd layers multiply 16/36 of it)
            'step_algorithmic_tflops': round(value * GF_PER_CROP / 1e3, 2),
            'step_ms_spread': spread,
            'final_gen_loss': round(gen_loss, 6),
        }
        if dp is not None:
            out['dp'] = dp
        if single:
            # the headline line must survive a failure of any side measurement
            if want_parity:
                try:
                    out['parity'] = parity_check(states0, lr, hr, first)
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: parity check failed to run: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out['parity'] = None
            if not args.no_roofline:
                try:
                    out['roofline'] = roofline_pass(trainer, lr, hr)
                    ex_tf = out['roofline']['step_gflop_executed'] / (elapsed / args.steps) / 1e3
                    out['roofline']['step_executed_tflops'] = round(ex_tf, 2)
                    out['roofline']['step_frac_of_fp32_mfma_peak'] = round(ex_tf / PEAK_TFLOPS, 4)
                    out['roofline']['north_star'] = north_star_in_graph(device)
                    # the forms of the north-star conv the step REALLY runs (the BatchNorm folds of the residual tower), from the
                    # same eager event pairs as by_kernel: 'BNL' normalises + activates its input while staging it, 'BNR' / 'BNB'
                    # are the two passes of a BatchNorm backward folded into a data gradient (csrc/rowtile.hip)
                    out['roofline']['north_star']['forms_in_step'] = {
                        k: {'launches_per_step': v['launches_per_step'], 'avg_launch_us_eager': round(v['ms_per_step'] / v['launches_per_step'] * 1e3, 2),
                            'tflops': v['tflops'], 'frac': v['frac']}
                        for k, v in out['roofline']['by_kernel'].items() if k.startswith('rt36_conv3x3_c64_kernel')}
                    out['roofline']['dominant_shape_in_graph'] = dominant_shape_in_graph(device)
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: roofline pass failed: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out.setdefault('roofline', None)
            if not args.no_cpu_baseline:
                try:
                    out['cpu_baseline'] = cpu_baseline(states0, lr, hr)
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: cpu baseline failed: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out['cpu_baseline'] = None
            if not args.no_other_configs:
                del trainer
                torch.cuda.empty_cache()
                out['other_configs'] = other_configs(device)
            if not args.no_dp_rehearsal:
                try:
                    out['dp_rehearsal'] = dp_rehearsal(tuple(int(v) for v in args.dp_contention.split(',') if v))
                except Exception as exc:  # noqa: BLE001
                    print(f'bench.py: dp rehearsal failed: {type(exc).__name__}: {exc}', file=sys.stderr)
                    out['dp_rehearsal'] = {'error': f'{type(exc).__name__}: {str(exc)[-300:]}'}
        assert_fracs(out)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
