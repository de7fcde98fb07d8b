"""CPU-side tests (no GPU): the oracle against the golden vectors captured from the reference,
the C-ABI library's exports, host logic (registry, flat buffers, schedulers, schema)."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from oracle import srgan as O
from oracle.weights import closed_form_state, step_state, tensor_digest


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


# ------------------------------------------------------------------ oracle pinned by the goldens
@pytest.mark.parametrize('tag', ['a', 'b'])
def test_oracle_generator_matches_reference_golden(tag):
    from torchsr_amd.srgan.generator import Generator
    gold = np.load(os.path.join(GOLDEN, 'srgan_generator.npz'))
    sd = closed_form_state(Generator().state_dict())
    leaves = O._leaves(sd)
    x = torch.from_numpy(gold[f'{tag}_x']).requires_grad_(True)
    y = O.generator_forward(sd, x, True)
    assert rel(y, gold[f'{tag}_y_train']) < 1e-5
    y.square().mean().backward()
    assert rel(x.grad, gold[f'{tag}_dx']) < 1e-4
    names = [k for k, v in sd.items() if v.is_floating_point() and 'running_' not in k]
    dig = dict(zip((str(k) for k in gold[f'{tag}_grad_keys']), gold[f'{tag}_grad_digest']))
    for k, leaf in zip(names, leaves):
        d = tensor_digest(leaf.grad)
        assert abs(d[0] - dig[k][0]) <= 1e-4 * max(dig[k][1], 1e-12), k
    ye = O.generator_forward(sd, x.detach(), False)
    assert rel(ye, gold[f'{tag}_y_eval']) < 1e-5


@pytest.mark.parametrize('tag,size', [('s32', 32), ('s96', 96)])
def test_oracle_discriminator_matches_reference_golden(tag, size):
    from torchsr_amd.srgan.discriminator import Discriminator
    gold = np.load(os.path.join(GOLDEN, 'srgan_discriminator.npz'))
    sd = closed_form_state(Discriminator(image_size=size).state_dict())
    p = O.discriminator_forward(sd, torch.from_numpy(gold[f'{tag}_x']), True)
    assert rel(p, gold[f'{tag}_p_train']) < 1e-5
    pe = O.discriminator_forward(sd, torch.from_numpy(gold[f'{tag}_x']), False)
    assert rel(pe, gold[f'{tag}_p_eval']) < 1e-5


def test_oracle_vgg_matches_reference_golden():
    from torchsr_amd.srgan.loss import make_vgg19_features
    gold = np.load(os.path.join(GOLDEN, 'vgg19.npz'))
    sd = closed_form_state(make_vgg19_features().state_dict(), prefix='features.')
    feat = O.vgg_features(sd, torch.from_numpy(gold['src']))
    assert rel(feat, gold['features']) < 1e-5
    loss = O.vgg_loss(sd, torch.from_numpy(gold['src']), torch.from_numpy(gold['tgt']))
    assert abs(loss.item() - float(gold['loss'])) < 1e-5 * float(gold['loss'])


def test_oracle_gan_steps_match_reference_trainer_golden():
    """Two consecutive steps: the fixtures start from ``step_state`` weights, which keep the discriminator out of
    saturation, so the second step pins arithmetic as well as the first."""
    from torchsr_amd.srgan.discriminator import Discriminator
    from torchsr_amd.srgan.generator import Generator
    from torchsr_amd.srgan.loss import make_vgg19_features
    gold = np.load(os.path.join(GOLDEN, 'srgan_steps.npz'))
    torch.set_num_threads(8)
    orc = O.SRGANStepOracle(step_state(Generator().state_dict(), 'srgan.G'), step_state(Discriminator().state_dict(), 'srgan.D'),
                            closed_form_state(make_vgg19_features().state_dict(), prefix='features.'))
    keys = [str(k) for k in gold['g_keys']]
    assert 0.5 < gold['gan_losses'][2][0] < 2.0  # disc-loss is still O(1) at the third step
    for step in range(2):
        out = orc.gan_step(torch.from_numpy(gold['low_res']), torch.from_numpy(gold['high_res']))
        assert np.allclose(out, gold['gan_losses'][step], rtol=1e-5)
        assert abs(out[3] - gold['gan_ref_gen_losses'][step]) < 2e-5 * out[3]
        for k, dg in zip(keys, gold['gan_g_digest'][step]):
            d = tensor_digest(orc.g[k].detach())
            assert abs(d[1] - dg[1]) <= 1e-5 * max(dg[1], 1e-9), (step, k)


def test_oracle_esrgan_matches_reference_golden():
    from oracle import esrgan as OE
    from torchsr_amd.esrgan.discriminator import Discriminator
    from torchsr_amd.esrgan.generator import Generator
    gold = np.load(os.path.join(GOLDEN, 'esrgan.npz'))
    sd = closed_form_state(Generator(num_rrdb_blocks=2).state_dict())
    assert rel(OE.generator_forward(sd, torch.from_numpy(gold['g_x'])), gold['g_y']) < 1e-5
    sd = closed_form_state(Discriminator(image_size=64).state_dict())
    assert rel(OE.discriminator_forward(sd, torch.from_numpy(gold['d_x']), True), gold['d_logits']) < 1e-5
    assert len(Generator().state_dict()) == 702 and len(Discriminator().state_dict()) == 60
    assert 0.3 < gold['gan_losses'][2][0] < 1.0 and gold['gan_losses'][0][1] < 5.0  # un-saturated D, O(1) image


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    from torchsr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    handle = _lib.load_handle(_lib.LIB_PATH)   # (on a GPU box: after torch has initialised HIP)
    header = open(os.path.join(ROOT, 'include', 'srx.h')).read()
    declared = set(re.findall(r'\b(srx_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations found'
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in include/srx.h but not exported'
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert handle.srx_version() == 100


def test_size_queries_and_error_reporting_without_gpu():
    from torchsr_amd import _lib
    L = _lib.lib()
    d = _lib.Conv2dDesc(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0)
    assert L.srx_conv2d_packed_fwd_floats(ctypes.byref(d)) == 64 * 576
    assert L.srx_conv2d_packed_bwd_floats(ctypes.byref(d)) == 64 * 576
    d2 = _lib.Conv2dDesc(16, 96, 96, 64, 64, 64, 64, 3, 3, 2, 1, 0, 0, 0.0, 0)
    # stride 2: four parity classes with 1+2+2+4 = 9 taps in total, no wasted work
    assert L.srx_conv2d_packed_bwd_floats(ctypes.byref(d2)) == 64 * 64 * 9
    bad = _lib.Conv2dDesc(16, 24, 24, 64, 62, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0)
    assert L.srx_conv2d_packed_fwd_floats(ctypes.byref(bad)) == 0
    assert 'Cin_s' in _lib.last_error()
    with pytest.raises(RuntimeError, match='bad argument'):
        _lib.call('srx_nchw_to_nhwc', None, None, 1, 3, 4, 4, 4, None)


def test_ops_refuse_cpu_tensors():
    from torchsr_amd import functional as F
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.to_nhwc(torch.rand(1, 3, 4, 4))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        F.mse_loss(torch.rand(4), torch.rand(4))


# ------------------------------------------------------------------ host logic
def test_state_dict_schema_and_seeded_init_match_reference_counts():
    from torchsr_amd.srgan.discriminator import Discriminator
    from torchsr_amd.srgan.generator import Generator
    g, d = Generator(), Discriminator()
    assert len(g.state_dict()) == 225 and len(d.state_dict()) == 48
    assert sum(p.numel() for p in g.parameters()) == 1547350
    assert sum(p.numel() for p in d.parameters()) == 23563649
    assert g.state_dict()['conv_layers.1.conv.weight'].shape == (256, 64, 3, 3)
    assert d.state_dict()['classifier.0.weight'].shape == (1024, 18432)
    assert Discriminator(image_size=32).classifier[0].in_features == 2048
    assert len(Generator(scale_factor=2).conv_layers) == 1


def test_vgg_layout_and_freeze():
    from torchsr_amd.srgan.loss import VGGLoss, make_vgg19_features
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        v = VGGLoss()
    keys = list(v.features.state_dict().keys())
    convs = [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28, 30, 32, 34]
    assert keys == [f'{i}.{p}' for i in convs for p in ('weight', 'bias')]
    assert len(v.features) == 36 and sum(p.numel() for p in v.parameters()) == 20024384
    assert not any(p.requires_grad for p in v.parameters()) and not v.features.training
    assert [i for i, k in O.vgg_feature_layout() if k == 'conv'] == convs
    with pytest.raises(RuntimeError):
        make_vgg19_features(1)


def test_flat_params_views_and_zero_grad():
    from torchsr_amd.optim import FlatParams
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))
    before = {k: v.clone() for k, v in m.state_dict().items()}
    f = FlatParams(m)
    assert f.numel == 16 + 4 + 8 + 4  # each tensor padded to 4 floats
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    m(torch.rand(4, 5)).sum().backward()
    assert f.grad.abs().sum() > 0 and m[0].weight.grad.data_ptr() == f.grad.data_ptr()
    f.zero_grad()
    assert f.grad.abs().sum() == 0
    with torch.no_grad():
        f.data.add_(1.0)
    assert torch.allclose(m[1].bias, before['1.bias'] + 1.0)
    m.load_state_dict(before)
    assert torch.equal(f.data[:15], before['0.weight'].flatten())


def test_vgg19_cfg_is_torchvision_cfg_e():
    """A literal copy of torchvision.models.vgg cfgs['E'] (public definition), written out here so that a typo
    shared by oracle.srgan.VGG19_CFG, the golden-fixture stub and the product cannot pass unnoticed."""
    from torchsr_amd.srgan.loss import VGG19_CFG
    cfg_e = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']
    assert VGG19_CFG == cfg_e and O.VGG19_CFG == cfg_e
    assert sum(1 for v in cfg_e if v != 'M') == 16 and len(cfg_e) + 16 == 37  # features[:36] drops the 5th max-pool


def test_steplr_values_match_torch():
    """optim.StepLR against torch.optim.lr_scheduler.StepLR for the reference's settings
    (step_size = epochs // 8, gamma 0.6, stepped once per epoch: torchsr/srgan/trainer.py:186-195,528-529)."""
    from torchsr_amd.optim import StepLR

    class Opt:  # the slice of FlatAdam that StepLR touches; no device needed
        def __init__(self, lr):
            self._lr = lr

        lr = property(lambda self: self._lr)

        def set_lr(self, lr):
            self._lr = float(lr)

    for epochs in (8, 20, 64, 1000):
        mine_opt = Opt(1e-4)
        mine = StepLR(mine_opt, step_size=epochs // 8, gamma=0.6)
        p = torch.nn.Parameter(torch.zeros(1))
        ref_opt = torch.optim.Adam([p], lr=1e-4)
        ref = torch.optim.lr_scheduler.StepLR(ref_opt, step_size=epochs // 8, gamma=0.6)
        for epoch in range(min(epochs, 40)):
            ref_opt.step()
            ref.step()
            mine.step()
            assert mine.get_last_lr()[0] == pytest.approx(ref.get_last_lr()[0], rel=1e-12), (epochs, epoch)
        sd = mine.state_dict()
        again = StepLR(Opt(1e-4), step_size=epochs // 8, gamma=0.6)
        again.load_state_dict(sd)
        assert again.get_last_lr() == mine.get_last_lr()
    assert StepLR(Opt(1e-4), step_size=0, gamma=0.6).step_size == 1  # --epochs < 8: the reference divides by zero


def test_cli_refuses_to_train_against_random_vgg_features(monkeypatch, tmp_path):
    """`torchsr train` without the pretrained VGG19 file must not silently optimise a meaningless perceptual
    loss (the reference always loads vgg19(pretrained=True), torchsr/srgan/loss.py:30)."""
    from torchsr_amd.torchsr import main
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv('TORCHSR_VGG19_WEIGHTS', raising=False)
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE', 'SLURM_NTASKS'):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit, match='vgg19-dcbb9e9d.pth not found'):
        main(['train', '--model', 'srgan', '--train-dir', 'synthetic:8', '--batch-size', '2'])


def test_backward_cuts_pause_and_resume():
    """ddp.BackwardCuts: a cut splits backward() into two calls whose combined effect is the plain backward."""
    from torchsr_amd import functional as F
    from torchsr_amd.ddp import BackwardCuts
    torch.manual_seed(3)
    a, b = torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    x = torch.rand(5, 4)

    def loss_of():  # the body runs twice per step, as the discriminator does (real and fake batch)
        h1, h2 = F.cut_point('mid', torch.tanh(a(x))), F.cut_point('mid', torch.tanh(a(2 * x)))
        return (b(h1) - b(h2)).square().mean()

    loss_of().backward()
    want = [p.grad.clone() for p in list(a.parameters()) + list(b.parameters())]
    for p in list(a.parameters()) + list(b.parameters()):
        p.grad = None
    cuts = BackwardCuts(('mid',))
    F.cut_hook[0] = cuts
    try:
        loss_of().backward()
        assert a.weight.grad is None and b.weight.grad is not None and len(cuts.pairs['mid']) == 2
        cuts.resume('mid')
        with torch.no_grad():  # nothing to cut where nothing is differentiated
            assert F.cut_point('mid', a(x)).requires_grad is False and not cuts.pending('mid')
        cuts.names = set()     # un-armed names pass through
        t = a(x)
        assert F.cut_point('mid', t) is t
    finally:
        F.cut_hook[0] = None
    got = [p.grad for p in list(a.parameters()) + list(b.parameters())]
    for g, w in zip(got, want):
        assert torch.allclose(g, w, rtol=1e-6, atol=1e-8)


def test_steplr_and_registry():
    from argparse import Namespace
    from torchsr_amd import models
    cls, crop = models.select_trainer_model(Namespace(model='SRGAN'))
    assert crop == 96 and cls.__name__ == 'SRGANTrainer'
    assert models.select_test_model(Namespace(model='srgan')).__name__ == 'Generator'
    with pytest.raises(RuntimeError, match='not supported'):
        models.select_trainer_model(Namespace(model='nope'))
    assert set(models.MODELS) == {'esrgan', 'srgan'} and models.CROP_SIZE == {'esrgan': 128, 'srgan': 96}


def test_descriptor_validation_without_a_gpu():
    """Host-side argument checking of the C ABI needs no device: limits and unsupported configurations are
    reported through the status code + srx_last_error, never by aborting."""
    import ctypes as C
    from torchsr_amd import _lib
    lib = _lib.lib()
    out = (C.c_int * 6)()

    def plan_error(*fields):
        d = _lib.Conv2dDesc(*fields)
        rc = lib.srx_conv2d_plan(C.byref(d), 0, out)
        buf = C.create_string_buffer(256)
        lib.srx_last_error(buf, 256)
        return rc, buf.value.decode()

    ok = (16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    assert plan_error(*ok)[0] == 0 and list(out)[:2] == [36, 64]
    # > 2^24 pixels: a whole-frame call (round 4: 64-bit tile bases, exact index division) on 128-row tiles
    assert plan_error(1, 5000, 5000, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)[0] == 0 and out[0] == 128
    rc, msg = plan_error(1, 50000, 50000, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)        # > 2^31 pixels
    assert rc != 0 and 'tile the image' in msg
    rc, msg = plan_error(16, 24, 24, 64, 62, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)              # stride < channels
    assert rc != 0 and 'Cin_s' in msg
    rc, msg = plan_error(16, 24, 24, 64, 64, 60, 60, 3, 3, 1, 1, 2, 0, 0.0, 0, 0)              # shuffle needs Cout % 16
    assert rc != 0 and 'shuffle' in msg
    rc, msg = plan_error(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 3, 0)              # upsample factor
    assert rc != 0 and 'up must be' in msg
    rc, msg = plan_error(16, 24, 24, 64, 64, 64, 64, 3, 3, 2, 1, 0, 0, 0.0, 2, 0)              # fused upsample: stride 1 only
    assert rc != 0 and 'up = 2' in msg
    assert plan_error(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 2, 0)[0] == 0 and out[3] >= 256  # 48x48 output rows
    rc, msg = plan_error(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 7)              # precision
    assert rc != 0 and 'precision' in msg
    rc, msg = plan_error(16, 2, 2, 64, 64, 64, 64, 5, 5, 1, 0, 0, 0, 0.0, 0, 0)                # empty output
    assert rc != 0 and 'empty output' in msg
    d = _lib.Conv2dDesc(*ok)
    assert lib.srx_conv2d_packed_fwd_floats(C.byref(d)) == 64 * 576
    assert lib.srx_conv2d_packed_bwd_floats(C.byref(d)) == 64 * 576
    assert lib.srx_conv2d_stat_rows(C.byref(d)) == 256


def test_abi_refuses_out_of_range_addends_slices_and_scales_without_a_gpu():
    """The three entry points added for the RRDB trunk (round 2) address tensors through caller-supplied strides, channel
    ranges and a host array; every combination that would read or write outside a row is refused with a status code
    BEFORE anything is launched ('the esr17 fault') -- so the checks run here, with fake pointers, on CPU."""
    import ctypes as C
    from torchsr_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('fake pointers: argument validation is exercised where a missed check cannot reach a device')
    lib = _lib.lib()
    fake = 0x10000  # never dereferenced: the calls below must fail in argument validation

    def err():
        buf = C.create_string_buffer(256)
        lib.srx_last_error(buf, 256)
        return buf.value.decode()

    # conv5 of a dense block: 192 input channels (row stride 192), dy dense 64 channels
    d = _lib.Conv2dDesc(2, 8, 8, 192, 192, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    e = _lib.DgradEpilogue()
    e.addend, e.addend_ld, e.addend_channels, e.addend_scale, e.out_scale = fake, 64, 0, 1.0, 0.2  # 192 channels from 64-float rows
    rc = lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None)
    assert rc != 0 and 'row stride' in err()
    e.addend_channels = 128                                                                       # still wider than the row
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0 and 'row stride' in err()
    e.addend_ld, e.addend_channels = 62, 60                                                       # not whole quads
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0 and 'quads' in err()
    e = _lib.DgradEpilogue()
    e.addend, e.addend_scale = fake + 4096, 0.5                                                   # the addend IS the output
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0 and 'of its own' in err()
    e = _lib.DgradEpilogue()
    e.accumulate, e.out_scale = 1, 0.2                                                            # scaled output + accumulate
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0
    e = _lib.DgradEpilogue()
    e.act_out, e.act_slope, e.c_lo, e.c_hi = fake, 0.2, 192, 224                                  # mask beyond the row
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0 and 'masked' in err()
    e.c_lo, e.c_hi = 160, 150                                                                     # empty / unaligned range
    assert lib.srx_conv2d_bwd_data_ex(C.byref(d), fake, fake, fake + 4096, C.byref(e), None, 0, None) != 0 and 'masked' in err()

    # channel slices of srx_axpby_channels / srx_copy_channels
    assert lib.srx_axpby_channels(fake, 192, 160, fake, 192, 0, fake, 64, 0, 64, 128, 0.2, 1.0, None) != 0 and 'out of range' in err()
    assert lib.srx_axpby_channels(fake, 192, 0, fake, 192, 0, fake, 64, 32, 64, 128, 0.2, 1.0, None) != 0 and 'out of range' in err()
    assert lib.srx_axpby_channels(fake, 192, -4, fake, 192, 0, fake, 64, 0, 64, 128, 0.2, 1.0, None) != 0 and 'out of range' in err()
    assert lib.srx_axpby_channels(fake, 192, 2, fake, 192, 0, fake, 64, 0, 64, 128, 0.2, 1.0, None) != 0 and 'multiples of 4' in err()
    assert lib.srx_copy_channels(fake, 64, 0, fake, 192, 160, 64, 128, 0, None) != 0 and 'out of range' in err()

    # grouped / scaled / paired weight gradients
    arr = lambda *v: (C.c_void_p * len(v))(*v)  # noqa: E731
    nws = lib.srx_conv2d_bwd_weight_multi_ws_floats(C.byref(d), 2)
    assert nws > 0
    bad = (C.c_float * 2)(1.0, float('nan'))
    rc = lib.srx_conv2d_bwd_weight_multi_scaled(C.byref(d), 2, 1, arr(fake, fake), arr(fake, fake), arr(fake, fake), 1, None, bad,
                                                fake, nws, None)
    assert rc != 0 and 'not finite' in err()
    ok = (C.c_float * 2)(1.0, 0.2)
    rc = lib.srx_conv2d_bwd_weight_multi_scaled(C.byref(d), 2, 1, arr(fake, fake), arr(fake, fake), arr(fake, fake), 1, None, ok,
                                                fake, 4, None)
    assert rc != 0 and 'workspace' in err()                                                      # short workspace
    rc = lib.srx_conv2d_bwd_weight_multi_scaled(C.byref(d), 3, 2, arr(fake, fake, fake), arr(fake, fake, fake), arr(fake, fake), 1,
                                                None, ok, fake, 1 << 40, None)
    assert rc != 0 and 'whole number of outputs' in err()
    rc = lib.srx_conv2d_bwd_weight_multi_scaled(C.byref(d), 2, 1, arr(fake, None), arr(fake, fake), arr(fake, fake), 1, None, ok,
                                                fake, nws, None)
    assert rc != 0 and 'null tensor' in err()
    dp = _lib.Conv2dDesc(2, 8, 8, 96, 192, 64, 192, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)                  # conv1 + conv2 of a dense block
    rc = lib.srx_conv2d_bwd_weight_multi_pair(C.byref(dp), 1, arr(fake), arr(fake), arr(fake), arr(fake), 128, 1, None, None,
                                              fake, 1 << 40, None)
    assert rc != 0 and 'cin_lo' in err()                                                         # first conv wider than the second
    rc = lib.srx_conv2d_bwd_weight_multi_pair(C.byref(dp), 1, arr(fake), arr(fake), arr(fake), arr(fake), 62, 1, None, None,
                                              fake, 1 << 40, None)
    assert rc != 0 and 'quads' in err()
    rc = lib.srx_conv2d_bwd_weight_multi_pair(C.byref(dp), 1, arr(fake), arr(fake), arr(fake), arr(fake), 64, 1, arr(fake), None,
                                              fake, 1 << 40, None)
    assert rc != 0 and 'both convs or neither' in err()

    # the fused dense block: its output never is the block buffer itself, its second addend never the output
    biases = arr(fake, fake, fake, fake, fake)
    rc = lib.srx_rdb_fwd(1, 8, 8, fake, 192, fake, biases, 0.2, 0.2, 1.0, None, 0, fake, 192, None)
    assert rc != 0 and 'alias' in err()
    rc = lib.srx_rdb_fwd(1, 8, 8, fake, 192, fake, biases, 0.2, 0.2, 0.2, fake + 4096, 192, fake + 4096, 192, None)
    assert rc != 0 and 'of its own' in err()


def test_library_carries_the_digest_of_its_sources(tmp_path):
    """A build is identified by the sha256 of its sources, compiled into the binary (``srx_build_info``): a stale library
    next to newer sources is detected by content, not by file times the builder controls."""
    import ctypes as C
    from torchsr_amd import _lib
    want = _lib.source_digest()
    assert len(want) == 64 and _lib.built_digest() == want
    buf = C.create_string_buffer(80)
    assert _lib.lib().srx_build_info(buf, 80) == 0 and buf.value.decode() == want
    # a binary built from other sources is told apart whatever its mtime
    blob = open(_lib.LIB_PATH, 'rb').read().replace(want.encode(), b'0' * 64)
    other = tmp_path / 'libsrx_other.so'
    other.write_bytes(blob)
    assert _lib.built_digest(str(other)) == '0' * 64 != want


def test_plans_follow_the_reserved_compute_units_without_a_gpu():
    """``srx_set_reserved_cus``: the residual tower's row tile cuts 16 x 24 x 24 pixels into 256 workgroups of 36 pixels on a
    free chip and into 192 of 48 once CUs are held by a collective's channel kernels; the generic planner sizes its rounds
    for the same count.  Plans are host code: no GPU needed (without one the library assumes 256 CUs)."""
    import ctypes as C
    from torchsr_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('plan arithmetic for a 256-CU part; on a GPU box the count is the device\'s')
    lib = _lib.lib()
    d = _lib.Conv2dDesc(16, 24, 24, 64, 64, 64, 64, 3, 3, 1, 1, 0, 0, 0.0, 0, 0)
    out = (C.c_int * 6)()
    try:
        assert lib.srx_plan_cus() == 256
        assert lib.srx_conv2d_plan(C.byref(d), 0, out) == 0 and (out[0], out[3]) == (36, 256)
        assert lib.srx_conv2d_stat_rows(C.byref(d)) == 256
        assert lib.srx_set_reserved_cus(16) == 0 and lib.srx_plan_cus() == 240
        assert lib.srx_conv2d_plan(C.byref(d), 0, out) == 0 and (out[0], out[3]) == (48, 192)
        assert lib.srx_conv2d_stat_rows(C.byref(d)) == 192
        # a VGG layer (M = 18432 pixels, 256 channels): whole rounds of 240 CUs, the rest K-split
        v = _lib.Conv2dDesc(32, 24, 24, 256, 256, 256, 256, 3, 3, 1, 1, 0, 1, 0.0, 0, 0)
        assert lib.srx_conv2d_plan(C.byref(v), 0, out) == 0
        full_rounds = (out[3] // 240) * 240
        assert out[3] >= full_rounds and out[3] != 256
        assert lib.srx_set_reserved_cus(500) != 0
    finally:
        lib.srx_set_reserved_cus(0)
    assert lib.srx_plan_cus() == 256


def test_vgg_loss_loads_a_torchvision_shaped_state_dict(tmp_path):
    """The pretrained branch of ``VGGLoss.__init__`` (torchsr/srgan/loss.py:30: ``vgg19(pretrained=True)``): the file
    torchvision downloads is a full VGG19 state dict -- ``features.N.{weight,bias}`` for the 16 convs plus the three
    ``classifier`` layers -- of which ``features[:36]`` is kept.  The real values are not available offline, so the file is
    written here with the closed-form fill (and token-sized classifier tensors: they are dropped unread)."""
    from torchsr_amd.srgan.loss import VGGLoss, make_vgg19_features
    own = make_vgg19_features(36).state_dict()
    full = {f'features.{k}': v for k, v in closed_form_state(own, prefix='features.').items()}
    for i in (0, 3, 6):
        full[f'classifier.{i}.weight'], full[f'classifier.{i}.bias'] = torch.ones(4, 4), torch.ones(4)
    path = tmp_path / 'vgg19-dcbb9e9d.pth'
    torch.save(full, path)
    v = VGGLoss(weights=str(path))
    assert v.pretrained and not v.features.training and not any(p.requires_grad for p in v.parameters())
    got = v.features.state_dict()
    assert list(got) == list(own)
    for k, t in got.items():
        assert torch.equal(t, full['features.' + k]), k
    # found through the environment variable too; a file without the feature tensors is an error, not a silent random init
    import os
    os.environ['TORCHSR_VGG19_WEIGHTS'] = str(path)
    try:
        assert VGGLoss().pretrained
    finally:
        del os.environ['TORCHSR_VGG19_WEIGHTS']
    torch.save({k: v for k, v in full.items() if not k.startswith('features.34')}, tmp_path / 'short.pth')
    with pytest.raises(RuntimeError, match='Missing key'):
        VGGLoss(weights=str(tmp_path / 'short.pth'))


@pytest.mark.parametrize('model', ['srgan', 'esrgan'])
def test_loaders_read_what_the_reference_trainer_writes(tmp_path, model):
    """tests/golden/checkpoint.npz: the keys and per-entry digests of ``_model_state`` of the reference's own trainers
    (srgan/trainer.py:233-258), plain and with the ``module.`` prefix a wrapped generator gives.  A file rebuilt from
    them must load through the trainer's ``_load_checkpoint`` and through ``test.load_generator_state`` into this
    package's generator, entry for entry."""
    import importlib
    from torchsr_amd.srgan.trainer import SRGANTrainer
    from torchsr_amd.test import load_generator_state
    gold = np.load(os.path.join(GOLDEN, 'checkpoint.npz'))
    assert list(gold[f'{model}_top_keys']) == ['epoch', 'phase', 'state']
    gen = importlib.import_module(f'torchsr_amd.{model}.generator').Generator()
    keys = [str(k) for k in gold[f'{model}_state_keys']]
    assert keys == list(gen.state_dict().keys())  # the checkpoint ABI: same names, same order
    values = step_state(gen.state_dict(), f'{model}.G')
    for k, dg in zip(keys, gold[f'{model}_state_digest']):
        assert np.allclose(tensor_digest(values[k].float()), dg, rtol=1e-12, atol=1e-12), k  # what the reference held
    prefixes = ['', 'module.'] if model == 'srgan' else ['']
    if model == 'srgan':
        assert [str(k) for k in gold['srgan_wrapped_state_keys']] == ['module.' + k for k in keys]
    for prefix in prefixes:
        path = str(tmp_path / f'{model}-gan-best{len(prefix)}.pth')
        torch.save({'epoch': 3, 'phase': f'{model}-gan', 'state': {prefix + k: values[k] for k in keys}}, path)
        ck = SRGANTrainer._load_checkpoint(None, path)
        assert ck['epoch'] == 3 and ck['phase'] == f'{model}-gan' and list(ck['state']) == keys
        for state in (ck['state'], load_generator_state(path)):
            fresh = importlib.import_module(f'torchsr_amd.{model}.generator').Generator()
            fresh.load_state_dict(state, strict=True)
            for k, v in fresh.state_dict().items():
                assert torch.equal(v, values[k]), (prefix, k)
    # a bare state dict (no wrapper) is accepted as well
    bare = str(tmp_path / 'bare.pth')
    torch.save({k: values[k] for k in keys}, bare)
    assert list(load_generator_state(bare)) == keys and list(SRGANTrainer._load_checkpoint(None, bare)['state']) == keys


def test_device_loader_shards_have_equal_length_on_every_rank():
    """Uneven shards would leave one rank in a collective alone: every rank must see the same number of
    train batches and a non-empty test shard, like the reference's DistributedSampler (torchsr/dataset.py:279,343-360)."""
    from torchsr_amd.dataset import DeviceLoader
    images = [torch.zeros(8, 8, 3, dtype=torch.uint8) for _ in range(127)]
    for world, batch, n in ((8, 16, 127), (8, 4, 33), (3, 2, 7), (2, 5, 11), (8, 16, 1)):
        for test in (False, True):
            loaders = [DeviceLoader(images[:n], 'cpu', batch, 8, 4, test, 0, 1, rank, world) for rank in range(world)]
            lens = [len(ld) for ld in loaders]
            assert len(set(lens)) == 1, (world, batch, n, test, lens)
            shards = [ld._shard(list(ld.order)) for ld in loaders]
            assert len({len(s) for s in shards}) == 1 and all(len(s) >= 1 for s in shards)
            assert set(i for s in shards for i in s) == set(range(n))  # every sample is somewhere
            if test:
                assert lens[0] >= 1
            # what __iter__ slices must exist on every rank
            assert all(lens[0] * batch - len(s) < batch for s in shards) if test else all(lens[0] * batch <= len(s) for s in shards)


def test_bf16_native_conv_chunks_stay_inside_32_bit_offsets():
    """srx_conv3x3_c64_bf16_fwd addresses a chunk's rows by 32-bit byte offsets from the chunk's first row: the planner must
    cut chunks so that rows per chunk (+ the rows in flight around it) x W x 128 bytes stays below 4 GiB, whatever H is
    (round-4 advice: rows_per_chunk could grow to H and the offsets wrapped silently).  Host only: nothing is launched."""
    import ctypes as C
    from torchsr_amd import _lib
    lib = _lib.lib()
    out = (C.c_int * 3)()
    for n, h, w in ((1, 1080, 1920), (1, 4320, 7680), (1, 400000, 5000), (1, 40000, 50000), (2, 9000, 200000), (1, 64, 24)):
        if n * h * w >= 2 ** 31:
            continue
        assert lib.srx_conv3x3_c64_bf16_plan(n, h, w, 64, out) == 0, (n, h, w)
        rpc, chunks, strips = out[0], out[1], out[2]
        rw = 4 if w <= 32 else (2 if w <= 64 else 1)
        assert rpc >= rw and rpc % rw == 0 and rpc * chunks >= h and strips >= 1
        assert (rpc + 2 + 3 * rw + 1) * w * 128 < 2 ** 32, (n, h, w, rpc)
    assert lib.srx_conv3x3_c64_bf16_plan(1, 16, 300000, 64, out) != 0  # one row of the window alone is past the range


def test_comm_configuration_is_decided_before_the_process_group(monkeypatch):
    """torchsr/torchsr.py:257-258 + srgan/trainer.py:142-157 (DistributedDataParallel): at world size > 1 the RCCL channel count
    is bounded and the launch plans are told how many CUs the channel kernels get while a large bucket is on the wire -- by
    default, not through a hand-set switch -- and a user's own NCCL_* values win."""
    from torchsr_amd import ddp
    for k in ('NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'SRX_NCCL_CHANNELS', 'SRX_COMM_RESERVED_CUS'):
        monkeypatch.delenv(k, raising=False)
    one = ddp.configure_comm(1, 'nccl')
    assert one['reserved_cus_in_comm_window'] == 0 and 'NCCL_MAX_NCHANNELS' not in os.environ
    gloo = ddp.configure_comm(2, 'gloo')
    assert gloo['reserved_cus_in_comm_window'] == 0 and 'NCCL_MAX_NCHANNELS' not in os.environ
    eight = ddp.configure_comm(8, 'nccl')
    assert os.environ['NCCL_MAX_NCHANNELS'] == os.environ['NCCL_MIN_NCHANNELS'] == str(ddp.DEFAULT_CHANNELS)
    assert eight['reserved_cus_in_comm_window'] == ddp.DEFAULT_CHANNELS and eight['nccl_max_nchannels'] == str(ddp.DEFAULT_CHANNELS)
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '4')
    monkeypatch.setenv('NCCL_MIN_NCHANNELS', '2')
    mine = ddp.configure_comm(8, 'nccl')
    assert mine['nccl_max_nchannels'] == '4' and mine['nccl_min_nchannels'] == '2' and mine['reserved_cus_in_comm_window'] == 4
    monkeypatch.setenv('SRX_COMM_RESERVED_CUS', '12')
    assert ddp.configure_comm(8, 'nccl')['reserved_cus_in_comm_window'] == 12


def test_bench_fractions_are_shares_of_a_peak():
    """bench.py's ``frac`` fields are executed MFMA FLOPs over a hardware peak: a Winograd F(2x2, 3x3) kernel books 16/36 of
    the direct convolution's multiplications, its direct-form rate travels as ``algorithmic_tflops``, and no line with a
    fraction above 1 is printed."""
    import bench
    r = bench.rates('wino_kernel<64>', 380.5e9, 1.881, bench.PEAK_TFLOPS)  # round 5's dominant kernel, per replay
    assert abs(r['algorithmic_tflops'] - 202.3) < 0.1 and abs(r['tflops'] - 89.9) < 0.1 and abs(r['frac'] - 0.5716) < 1e-3
    assert r['algorithmic_speedup'] == 2.25
    d = bench.rates('gconv_kernel<128, 64, 32, 32, 1, 16, 0>', 41.448e9, 0.4631, bench.PEAK_TFLOPS)
    assert d['tflops'] == d['algorithmic_tflops'] and 'algorithmic_speedup' not in d
    leg = bench.leg_rates(9199.0, 62.43e-3, bench.PEAK_TFLOPS, {'gflop_not_executed': 9199.0 * 0.9 * 20 / 36})
    assert leg['frac'] < 1.0 and leg['algorithmic_tflops'] > leg['tflops']
    bench.assert_fracs({'roofline': {'frac': 0.57, 'step_frac_of_fp32_mfma_peak': 0.41, 'by_shape': [{'frac': 0.6}]}})
    for bad in ({'roofline': {'frac': 1.1218}}, {'x': [{'dominant_kernel': {'frac': 1.03}}]}, {'step_frac_of_fp32_mfma_peak': 1.2}):
        with pytest.raises(AssertionError):
            bench.assert_fracs(bad)
