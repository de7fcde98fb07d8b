"""ctypes binding of ``libsrx_hip.so`` -- the C ABI declared in ``include/srx.h``.

The product path has no CPU fallback: if the HIP library is missing, loading
raises.  Every entry point returns an ``int`` status; non-zero is turned into a
``RuntimeError`` carrying the library's thread-local message.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.environ.get('SRX_LIB') or os.path.join(CSRC, 'libsrx_hip.so')  # SRX_LIB: developer A/B builds on one GPU box
SOURCES = ['api.cpp', 'gconv.hip', 'c64.hip', 'thin9.hip', 'rdb.hip', 'thin.hip', 'rowtile.hip', 'augment.hip', 'norm.hip', 'eltwise.hip', 'linear.hip', 'loss.hip', 'head.hip', 'wino.hip', 'optim.hip']

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_PRELU = 0, 1, 2, 3


class Conv2dDesc(C.Structure):
    """Mirror of ``srx_conv2d_t`` (include/srx.h)."""
    _fields_ = [
        ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
        ('Cin', C.c_int32), ('Cin_s', C.c_int32), ('Cout', C.c_int32), ('Cout_s', C.c_int32),
        ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
        ('shuffle', C.c_int32), ('act', C.c_int32), ('slope', C.c_float), ('up', C.c_int32),
        ('precision', C.c_int32),
    ]


class DgradEpilogue(C.Structure):
    """Mirror of ``srx_dgrad_epilogue_t`` (include/srx.h); zero fields are the defaults."""
    _fields_ = [
        ('accumulate', C.c_int32), ('out_scale', C.c_float), ('addend', C.c_void_p), ('addend_ld', C.c_int32),
        ('addend_channels', C.c_int32), ('addend_scale', C.c_float), ('act_out', C.c_void_p), ('act_slope', C.c_float),
        ('c_lo', C.c_int32), ('c_hi', C.c_int32),
    ]


class GanHead(C.Structure):
    """Mirror of ``srx_gan_head_t`` (include/srx.h)."""
    _fields_ = [('mode', C.c_int32), ('B', C.c_int32), ('J', C.c_int32), ('n_first', C.c_int32), ('slope', C.c_float),
                ('adv_weight', C.c_float)]


HEAD_SRGAN_D, HEAD_SRGAN_G, HEAD_ESRGAN_D, HEAD_ESRGAN_G = 0, 1, 2, 3


def source_digest() -> str:
    """sha256 over every source of the library (names + bytes, fixed order): the identity of a build."""
    import hashlib
    h = hashlib.sha256()
    for name in SOURCES + ['srx_common.h', os.path.join('..', '..', 'include', 'srx.h')]:
        h.update(os.path.basename(name).encode() + b'\0')
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def built_digest(path: str = None):
    """The source digest a built library carries (``srx_build_info``), or None when it cannot be read."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    try:
        with open(path, 'rb') as f:
            blob = f.read()
    except OSError:
        return None
    tag = b'srx-build-sources-sha256:'
    at = blob.find(tag)
    if at < 0:
        return None
    return blob[at + len(tag):at + len(tag) + 64].decode('ascii', 'replace')


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``csrc/libsrx_hip.so`` (in-tree).

    A build is identified by the sha256 of its sources, compiled into the library (``srx_build_info``): an existing
    library is reused only when it carries the digest of the sources lying next to it -- never by file times.
    ``force`` (or ``SRX_FORCE_BUILD=1``) recompiles every translation unit.  Objects are compiled in parallel into
    ``csrc/build/`` and cached per source digest, so an edit recompiles one file."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    force = force or os.environ.get('SRX_FORCE_BUILD') == '1'
    digest = source_digest()
    if not force and built_digest(LIB_PATH) == digest:
        return LIB_PATH
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objdir = os.path.join(CSRC, 'build')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
    with open(os.path.join(CSRC, 'srx_common.h'), 'rb') as f:
        hdr = f.read()
    with open(os.path.join(_HERE, '..', 'include', 'srx.h'), 'rb') as f:
        hdr += f.read()

    def compile_one(name):
        src = os.path.join(CSRC, name)
        extra = ['-DSRX_SOURCES_SHA256="%s"' % digest] if name == 'api.cpp' else []
        with open(src, 'rb') as f:
            key = hashlib.sha256(hdr + f.read() + ' '.join(flags + extra).encode()).hexdigest()[:24]
        obj = os.path.join(objdir, f'{os.path.splitext(name)[0]}.{key}.o')
        if force or not os.path.exists(obj):
            for stale in os.listdir(objdir):
                if stale.startswith(os.path.splitext(name)[0] + '.') and stale.endswith('.o'):
                    os.unlink(os.path.join(objdir, stale))
            cmd = [hipcc] + flags + extra + ['-c', src, '-o', obj + '.tmp']
            if verbose:
                print(' '.join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True, cwd=CSRC)
            os.replace(obj + '.tmp', obj)
        return obj

    jobs = max(1, min(len(SOURCES), int(os.environ.get('SRX_BUILD_JOBS', str(min(8, os.cpu_count() or 1))))))
    with ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB_PATH + '.tmp'] + objs
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(LIB_PATH + '.tmp', LIB_PATH)
    assert built_digest(LIB_PATH) == digest, 'the built library does not carry its source digest'
    return LIB_PATH


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_F = C.c_float
_Z = C.c_size_t
_D = C.POINTER(Conv2dDesc)

# name -> (restype, argtypes); status-returning functions have restype int and are checked
_SIGS = {
    'srx_version': (_I, []),
    'srx_last_error': (_I, [C.c_char_p, _Z]),
    'srx_device_cus': (_I, []),
    'srx_build_info': (_I, [C.c_char_p, _Z]),
    'srx_set_reserved_cus': (_I, [_I]),
    'srx_plan_cus': (_I, []),
    'srx_occupy_cus': (_I, [_I, _I, _P, _I, _P]),
    'srx_prof_start': (_I, [_I]),
    'srx_prof_stop': (_I, []),
    'srx_prof_get': (_I, [_I, C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    'srx_nchw_to_nhwc': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'srx_nhwc_to_nchw': (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    'srx_conv2d_packed_fwd_floats': (_Z, [_D]),
    'srx_conv2d_packed_bwd_floats': (_Z, [_D]),
    'srx_conv2d_fwd_ws_floats': (_Z, [_D]),
    'srx_conv2d_bwd_data_ws_floats': (_Z, [_D]),
    'srx_conv2d_bwd_weight_ws_floats': (_Z, [_D]),
    'srx_conv2d_stat_rows': (_I, [_D]),
    'srx_conv2d_plan': (_I, [_D, _I, C.POINTER(C.c_int)]),
    'srx_pack_table_bytes': (_Z, [_I]),
    'srx_pack_table_build': (_I, [_P, _I, _P, _P, _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    'srx_pack_table_add_wino': (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_longlong), _D, _P, _P, _I]),
    'srx_pack_table_run': (_I, [_P, _I, C.c_longlong, _P]),
    'srx_conv2d_pack': (_I, [_D, _P, _P, _P, _P]),
    'srx_conv2d_fwd': (_I, [_D, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_conv3x3_c64_bf16_packed_bytes': (_Z, [_I]),
    'srx_conv3x3_c64_bf16_pack': (_I, [_P, _P, _P, _I, _I, _P, _P]),
    'srx_conv3x3_c64_bf16_fwd': (_I, [_I, _I, _I, _I, _I, _P, _P, _F, _P, _P, _I, _P]),
    'srx_conv3x3_c64_bf16_plan': (_I, [_I, _I, _I, _I, C.POINTER(C.c_int)]),
    'srx_f32_to_bf16': (_I, [_P, _P, _L, _P]),
    'srx_bf16_to_f32': (_I, [_P, _P, _L, _P]),
    'srx_conv9x9_c64_thin_bf16_packed_bytes': (_Z, []),
    'srx_conv9x9_c64_thin_bf16_pack': (_I, [_P, _P, _I, _P, _P]),
    'srx_conv9x9_c64_thin_bf16_fwd': (_I, [_I, _I, _I, _P, _P, _P, _P]),
    'srx_conv2d_fwd_bf16in': (_I, [_D, _P, _P, _P, _P, _P]),
    'srx_conv2d_fwd_residual': (_I, [_D, _P, _P, _P, _P, _F, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_data': (_I, [_D, _P, _P, _P, _I, _P, _Z, _P]),
    'srx_conv2d_bwd_data_add': (_I, [_D, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_data_act': (_I, [_D, _P, _P, _P, _F, _I, _I, _I, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_data_bn_rows': (_I, [_D]),
    'srx_conv2d_bwd_data_bn': (_I, [_D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'srx_conv2d_fwd_bn_in_ok': (_I, [_D]),
    'srx_conv2d_bwd_data_bn_in_ok': (_I, [_D]),
    'srx_conv2d_bwd_data_bn_in': (_I, [_D] + [_P] * 20),
    'srx_conv2d_fwd_bn_in': (_I, [_D] + [_P] * 13),
    'srx_bn_act_bwd_finish': (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P, _P]),
    'srx_conv2d_bwd_data_ex': (_I, [_D, _P, _P, _P, C.POINTER(DgradEpilogue), _P, _Z, _P]),
    'srx_conv2d_bwd_weight': (_I, [_D, _P, _P, _P, _I, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_weight_multi_ws_floats': (_Z, [_D, _I]),
    'srx_conv2d_bwd_weight_multi': (_I, [_D, _I, _I, _P, _P, _P, _I, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_weight_multi_scaled': (_I, [_D, _I, _I, _P, _P, _P, _I, _P, _P, _P, _Z, _P]),
    'srx_conv2d_bwd_weight_multi_pair': (_I, [_D, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _Z, _P]),
    'srx_rdb_packed_bytes': (_Z, []),
    'srx_rdb_pack': (_I, [_P, _I, _P, _P]),
    'srx_rdb_pack_bwd': (_I, [_P, _I, _P, _P]),
    'srx_rdb_bwd': (_I, [_I, _I, _I, _P, _I, _F, _P, _I, _P, _F, _P, _I, _P, _I, _F, _P, _I, _P, _I, _P]),
    'srx_rdb_fwd': (_I, [_I, _I, _I, _P, _I, _P, _P, _F, _F, _F, _P, _I, _P, _I, _P]),
    'srx_colsum_ws_floats': (_Z, [_L, _I]),
    'srx_colsum': (_I, [_P, _P, _L, _I, _I, _I, _P, _Z, _P]),
    'srx_crop_flip_u8': (_I, [_P, _P, _P, _I, _I, _P]),
    'srx_bicubic_down': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    'srx_act_bwd_from_out': (_I, [_P, _P, _P, _L, _I, _F, _P]),
    'srx_act_bwd_from_out_strided': (_I, [_P, _I, _P, _I, _P, _I, _L, _I, _I, _F, _P]),
    'srx_prelu_fwd': (_I, [_P, _P, _P, _L, _P]),
    'srx_prelu_bwd': (_I, [_P, _P, _P, _P, _P, _I, _L, _P, _P]),
    'srx_lrelu_fwd': (_I, [_P, _P, _L, _F, _P]),
    'srx_axpby': (_I, [_P, _P, _P, _L, _F, _F, _P]),
    'srx_ring_push': (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _P]),
    'srx_axpby_channels': (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _L, _F, _F, _P]),
    'srx_copy_channels': (_I, [_P, _I, _I, _P, _I, _I, _I, _L, _I, _P]),
    'srx_upsample_nearest2x_fwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'srx_upsample_nearest2x_bwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'srx_mean_fwd': (_I, [_P, _P, _L, _P, _P]),
    'srx_mean_bwd': (_I, [_P, _P, _P, _L, _P]),
    'srx_sigmoid_fwd': (_I, [_P, _P, _L, _P]),
    'srx_sigmoid_bwd': (_I, [_P, _P, _P, _L, _P]),
    'srx_bn_stat_rows': (_I, [_L]),
    'srx_bn_rows_per_block': (_I, [_L]),
    'srx_bn_partial_stats': (_I, [_P, _P, _L, _I, _P]),
    'srx_bn_finalize': (_I, [_P, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    'srx_bn_eval_stats': (_I, [_P, _P, _I, _F, _P, _P, _P]),
    'srx_bn_act_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _P]),
    'srx_bn_bwd_ws_floats': (_Z, [_L, _I]),
    'srx_bn_act_bwd_reduce': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_bn_act_bwd_apply': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P, _I, _P]),
    'srx_bn_train_fwd': (_I, [_P, _P, _I, _L, _I, _I, _F, _F, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _P, _P]),
    'srx_bn_act_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _F, _P, _I, _P, _P, _P, _P, _Z, _P]),
    'srx_maxpool2x2_fwd': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'srx_maxpool2x2_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'srx_maxpool2x2_relu_bwd': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'srx_linear_ws_floats': (_Z, [_I, _I, _I]),
    'srx_linear_fwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _Z, _P]),
    'srx_linear_bwd_data': (_I, [_P, _P, _P, _I, _I, _I, _P, _Z, _P]),
    'srx_linear_bwd_weight': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'srx_mse_fwd': (_I, [_P, _P, _P, _L, _P, _P]),
    'srx_l1_fwd': (_I, [_P, _P, _P, _L, _P, _P]),
    'srx_mse_bwd': (_I, [_P, _P, _P, _P, _P, _L, _P]),
    'srx_l1_bwd': (_I, [_P, _P, _P, _P, _P, _L, _P]),
    'srx_l1_fwd_count': (_I, [_P, _P, _P, _L, _L, _P, _P]),
    'srx_l1_bwd_count': (_I, [_P, _P, _P, _P, _P, _L, _L, _P]),
    'srx_bce_fwd': (_I, [_P, _F, _P, _L, _P, _P]),
    'srx_bce_bwd': (_I, [_P, _F, _P, _P, _L, _P]),
    'srx_bce_logits_fwd': (_I, [_P, _P, _F, _P, _L, _P, _P]),
    'srx_bce_logits_bwd': (_I, [_P, _P, _F, _P, _P, _L, _P]),
    'srx_wino_applicable': (_I, [_D]),
    'srx_wino_packed_floats': (_Z, [_D]),
    'srx_wino_ws_floats': (_Z, [_D, _I]),
    'srx_wino_plan': (_I, [_D, _I, C.POINTER(C.c_int)]),
    'srx_wino_pack': (_I, [_D, _P, _P, _I, _P]),
    'srx_wino_fwd': (_I, [_D, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_wino_bwd_data': (_I, [_D, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_conv3x3_bf16s_applicable': (_I, [_D]),
    'srx_conv3x3_bf16s_packed_bytes': (_Z, [_D]),
    'srx_conv3x3_bf16s_pack': (_I, [_D, _P, _P, _P, _P]),
    'srx_conv3x3_bf16s_ws_floats': (_Z, [_D, _I]),
    'srx_conv3x3_bf16s_fwd': (_I, [_D, _P, _P, _P, _I, _P, _I, _P, _Z, _P]),
    'srx_conv3x3_bf16s_bwd_data': (_I, [_D, _P, _P, _P, _P, _I, _P, _Z, _P]),
    'srx_conv2d_fwd_first3_to_bf16': (_I, [_D, _P, _P, _P, _P, _P]),
    'srx_maxpool2x2_fwd_to_bf16': (_I, [_P, _P, _I, _I, _I, _I, _P]),
    'srx_maxpool2x2_relu_bwd_bf16': (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    'srx_act_bwd_from_out_to_bf16': (_I, [_P, _P, _P, C.c_int64, _I, C.c_float, _P]),
    'srx_wino_infer_applicable': (_I, [_D]),
    'srx_wino_fwd_act': (_I, [_D, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_wino_force_plan': (_I, [_I, _I, _I]),
    'srx_wino_stat_rows': (_I, [_D]),
    'srx_wino_fwd_stats': (_I, [_D, _P, _P, _P, _P, _P, _P, _Z, _P]),
    'srx_gan_head_fwd': (_I, [C.POINTER(GanHead), _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'srx_gan_head_bwd': (_I, [C.POINTER(GanHead), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'srx_adam_step': (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _P]),
}
# functions whose int return value is data, not a status
_UNCHECKED = {'srx_wino_applicable', 'srx_wino_infer_applicable', 'srx_conv3x3_bf16s_applicable', 'srx_wino_stat_rows', 'srx_conv2d_bwd_data_bn_rows', 'srx_conv2d_fwd_bn_in_ok', 'srx_conv2d_bwd_data_bn_in_ok', 'srx_pack_table_bytes', 'srx_version', 'srx_last_error', 'srx_device_cus', 'srx_plan_cus', 'srx_prof_stop', 'srx_conv2d_stat_rows', 'srx_bn_stat_rows', 'srx_bn_rows_per_block'}

EXPORTS = tuple(_SIGS.keys())

_lib = None


def load_handle(path: str) -> C.CDLL:
    """dlopen the library -- after PyTorch has initialised HIP when there is a GPU.  Loading it first (its fat binary
    then registers with a runtime torch initialises later) makes every launch from it fail with 'no ROCm-capable device
    is detected' on this stack (measured: tools/experiments/bisect_smoke.py).  Without a GPU it simply loads."""
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
    return C.CDLL(path)


def _ensure_fresh():
    """The library must carry the digest of the sources next to it.  A stale one is rebuilt (under a file lock: test
    runs start several processes) when hipcc is there; otherwise loading fails loudly -- never a silent old binary."""
    want = source_digest()
    if built_digest(LIB_PATH) == want:
        return
    import fcntl
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if built_digest(LIB_PATH) == want:
            return
        have = built_digest(LIB_PATH)
        print(f'torchsr_amd: {LIB_PATH} was built from other sources ({have and have[:12]} != {want[:12]}); rebuilding',
              file=sys.stderr)
        try:
            build()
        except (OSError, subprocess.CalledProcessError) as exc:
            raise RuntimeError(f'{LIB_PATH} is stale (built from sources {have and have[:12]}, the tree has {want[:12]}) and '
                               f'could not be rebuilt: {exc}.  Run `python -c "import __graft_entry__ as g; g.build()"`.') from exc


def lib() -> C.CDLL:
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the MI355X HIP extension has not been built. '
                'Run `python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). '
                'There is no CPU fallback for the product path.')
        if not os.environ.get('SRX_LIB') and os.environ.get('SRX_ALLOW_STALE') != '1':
            _ensure_fresh()
        handle = load_handle(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                if os.environ.get('SRX_LIB'):  # a developer's older A/B build: calling the missing entry point raises
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().srx_last_error(buf, 512)
    return buf.value.decode('utf-8', 'replace')


def call(name: str, *args):
    """Call a status-returning entry point; raise RuntimeError on failure."""
    rc = getattr(lib(), name)(*args)
    if name not in _UNCHECKED and _SIGS[name][0] is _I and rc != 0:
        raise RuntimeError(f'{name} failed (code {rc}): {last_error()}')
    return rc
