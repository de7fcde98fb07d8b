"""One-off robustness run at the CLI's default batch size (64): four GAN steps of each trainer (eager, then the replayed
hipGraph), losses finite and equal between the eager and the replayed steps' magnitude."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_configs_common import batch, targs  # noqa: E402

dev = torch.device('cuda:0')
for name, crop, amp in (('srgan', 96, False), ('esrgan', 128, True)):
    if name == 'srgan':
        from torchsr_amd.srgan.trainer import SRGANTrainer as T
    else:
        from torchsr_amd.esrgan.trainer import ESRGANTrainer as T
    torch.manual_seed(0)
    t = T(dev, targs(64, amp), [], [], 64, 64)
    lr, hr = batch(64, crop)
    for step in range(4):
        out = {k: float(v) for k, v in t.gan_step(lr, hr).items()}
        assert all(v == v and abs(v) < 1e6 for v in out.values()), out
        print(name, step, {k: round(v, 5) for k, v in out.items()}, flush=True)
    pre = float(t.pretrain_step(lr, hr))
    print(name, 'pretrain', round(pre, 5), 'graphs', sorted(t._graphs), flush=True)
    del t
    torch.cuda.empty_cache()
print('ok')
