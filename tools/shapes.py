"""Per (kernel, layer shape) launch times of one eager GAN step (developer tool): the library's own launch profiler
(HIP events around every conv kernel), written to the file named on the command line.

    python tools/shapes.py {srgan|esrgan} {fp32|amp} OUT.txt
"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import torch  # noqa: E402

import bench  # noqa: E402
from tools.bench_configs_common import batch, targs  # noqa: E402

model, prec, out = sys.argv[1:4]
os.environ['SRX_BENCH_SHAPES'] = out
dev = torch.device('cuda:0')
torch.manual_seed(0)
if model == 'esrgan':
    from torchsr_amd.esrgan.trainer import ESRGANTrainer as T
    crop = 128
else:
    from torchsr_amd.srgan.trainer import SRGANTrainer as T
    crop = 96
t = T(dev, targs(16, prec == 'amp'), [], [], 16, 16)
t.use_graphs = False
lr, hr = batch(16, crop)
r = bench.roofline_pass(t, lr, hr, reps=2)
print(model, prec, 'conv ms/step', r['conv_ms_per_step'])
