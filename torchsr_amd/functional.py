"""Host-side operators: ``torch.autograd.Function`` wrappers around the C-ABI HIP kernels.

PyTorch is plumbing here (device memory via the caching allocator, the current
stream, the autograd tape); every device computation on the hot path is a
hand-written gfx950 kernel reached through ``_lib.call``.  Activations are NHWC
fp32 tensors ``[N, H, W, C_s]``; parameters keep the reference's layouts.

Nothing in this module falls back to CPU or to ATen kernels for the hot ops:
non-CUDA inputs raise.
"""
import contextlib
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _dev, _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU, HEAD_ESRGAN_D, HEAD_ESRGAN_G, HEAD_SRGAN_D, HEAD_SRGAN_G,  # noqa: F401
                   Conv2dDesc, call)

import ctypes as C


# When enabled (by the trainers, whose parameters' ``.grad`` are persistent views of one flat
# buffer) parameter gradients are accumulated straight into ``param.grad`` by the kernels and the
# Functions return ``None`` for them: no per-parameter AccumulateGrad add kernels (~200 per step).
direct_grads = [False]


def _sink(param: Optional[Tensor]) -> Optional[Tensor]:
    if not direct_grads[0] or param is None or not param.requires_grad:
        return None
    g = param.grad
    if g is None or not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
        return None
    return g


# Pause points of the backward pass.  A data-parallel trainer installs a ``ddp.BackwardCuts`` here so that
# ``loss.backward()`` stops where a gradient bucket is complete and the bucket's all-reduce can be launched
# before the rest of the backward pass is issued; without a hook a cut point is the identity.
cut_hook = [None]


def cut_point(name: str, t: Tensor) -> Tensor:
    hook = cut_hook[0]
    return t if hook is None else hook(name, t)


class WeightGradQueue:
    """Weight-gradient work of one backward pass, collected and issued together.

    Autograd reaches a conv's weight gradient right after its data gradient, one layer at a time, but nothing
    reads a weight gradient before the optimiser.  For the generator's 33 identical 3x3 64->64 convs a single
    problem is 0.68 GFLOP -- launched alone it is mostly pipeline fill plus a slab reduction -- so, while a queue
    is installed (``deferred_weight_grads``), the conv Functions only record ``(descriptor, x, dy, sink)`` and the
    queue hands every group of equal-geometry problems to ``srx_conv2d_bwd_weight_multi`` when it is flushed:
    one launch with long loops instead of 33 short ones.  Problems that share a sink (the discriminator's real
    and fake passes) become segments of one gradient.  Only gradients that accumulate straight into the flat
    ``.grad`` buffers are deferred; the queued tensors stay alive until the flush.
    """
    MAXP = 72  # WG_MAXP of gconv.hip

    def __init__(self):
        self.groups = {}
        self.pairs = {}

    def add(self, d: Conv2dDesc, x_ptr: int, dy_ptr: int, sink_ptr: int, bias_ptr, keep, scale: float = 1.0) -> None:
        """``scale``: the tensor at ``dy_ptr`` stands for ``scale * dy`` (one value per sink)."""
        key = tuple(getattr(d, f) for f, _ in Conv2dDesc._fields_)
        self.groups.setdefault(key, [d, []])[1].append((x_ptr, dy_ptr, sink_ptr, bias_ptr or 0, keep, float(scale)))

    def add_pair(self, d: Conv2dDesc, cin_lo: int, x_ptr: int, dy_ptr: int, sinks, bias_sinks, keep) -> None:
        """Two convs that read the same input buffer and whose output gradients are adjacent channel slices, as ONE
        problem (``srx_conv2d_bwd_weight_multi_pair``): ``sinks`` / ``bias_sinks`` are (first conv, second conv)."""
        key = ('pair', cin_lo, bias_sinks is None) + tuple(getattr(d, f) for f, _ in Conv2dDesc._fields_)
        self.pairs.setdefault(key, [d, cin_lo, []])[2].append((x_ptr, dy_ptr, sinks, bias_sinks, keep))

    def _flush_pairs(self) -> None:
        pairs, self.pairs = self.pairs, {}
        L, s = _lib.lib(), _stream()
        arr = lambda vals: (C.c_void_p * len(vals))(*vals)  # noqa: E731
        for d, cin_lo, items in pairs.values():
            dref = C.byref(d)
            for i in range(0, len(items), self.MAXP):
                part = items[i:i + self.MAXP]
                n = len(part)
                nws = L.srx_conv2d_bwd_weight_multi_ws_floats(dref, n)
                ws = torch.empty(max(int(nws), 4), dtype=torch.float32, device=part[0][4][0].device)
                with_bias = part[0][3] is not None
                call('srx_conv2d_bwd_weight_multi_pair', dref, n, arr([p[0] for p in part]), arr([p[1] for p in part]),
                     arr([p[2][0] for p in part]), arr([p[2][1] for p in part]), cin_lo, 1,
                     arr([p[3][0] for p in part]) if with_bias else None, arr([p[3][1] for p in part]) if with_bias else None,
                     _p(ws), nws, s)

    def flush(self) -> None:
        if self.pairs:
            self._flush_pairs()
        groups, self.groups = self.groups, {}
        if not groups:
            return
        L, s = _lib.lib(), _stream()
        for d, items in groups.values():
            by_sink = {}
            for it in items:
                by_sink.setdefault((it[2], it[3]), []).append(it)
            by_count = {}
            for sink, its in by_sink.items():  # outputs with the same number of segments go into one launch
                by_count.setdefault(len(its), []).append((sink, its))
            dref = C.byref(d)
            for per_out, outs in by_count.items():
                step = max(1, self.MAXP // per_out)
                for i in range(0, len(outs), step):
                    part = outs[i:i + step]
                    probs = [it for _, its in part for it in its]
                    n = len(probs)
                    arr = lambda vals: (C.c_void_p * len(vals))(*vals)  # noqa: E731
                    xs, dys = arr([p[0] for p in probs]), arr([p[1] for p in probs])
                    dws, dbs = arr([sk[0] for sk, _ in part]), arr([sk[1] or None for sk, _ in part])
                    nws = L.srx_conv2d_bwd_weight_multi_ws_floats(dref, n)
                    ws = torch.empty(max(int(nws), 4), dtype=torch.float32, device=probs[0][4][0].device)
                    scales = [its[0][5] for _, its in part]
                    if any(its[k][5] != its[0][5] for _, its in part for k in range(len(its))):
                        raise RuntimeError('WeightGradQueue: the segments of one gradient must share their scale')
                    if all(v == 1.0 for v in scales):
                        call('srx_conv2d_bwd_weight_multi', dref, n, per_out, xs, dys, dws, 1, dbs, _p(ws), nws, s)
                    else:
                        call('srx_conv2d_bwd_weight_multi_scaled', dref, n, per_out, xs, dys, dws, 1, dbs,
                             (C.c_float * len(scales))(*scales), _p(ws), nws, s)


wgrad_queue = [None]


@contextlib.contextmanager
def deferred_weight_grads():
    """Collect the conv weight gradients of the backward passes run inside and issue them on exit."""
    q, old = WeightGradQueue(), wgrad_queue[0]
    wgrad_queue[0] = q
    try:
        yield q
    finally:
        wgrad_queue[0] = old
    q.flush()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, what: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f'{what}: expected a tensor on the MI355X (cuda) device, got {t.device}; '
                           'torchsr_amd has no CPU fallback')
    if t.dtype != torch.float32:
        raise RuntimeError(f'{what}: expected float32, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


def _chk16(t: Tensor, what: str) -> Tensor:
    """A bf16 NHWC activation of the bf16-native inference chain (csrc/c64.hip)."""
    if t.device.type != 'cuda':
        raise RuntimeError(f'{what}: the MI355X path needs a CUDA/HIP tensor, got {t.device} (no CPU fallback)')
    if t.dtype != torch.bfloat16:
        raise RuntimeError(f'{what}: expected a bfloat16 tensor, got {t.dtype}')
    return t if t.is_contiguous() else t.contiguous()


def to_bf16(x: Tensor) -> Tensor:
    """fp32 -> bf16 (round to nearest even), same shape: the entry of the bf16-native inference chain."""
    x = _chk(x, 'to_bf16.input')
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call('srx_f32_to_bf16', _p(x), _p(y), x.numel(), _stream())
    return y


def to_f32(x: Tensor) -> Tensor:
    x = _chk16(x, 'to_f32.input')
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    call('srx_bf16_to_f32', _p(x), _p(y), x.numel(), _stream())
    return y


def conv2d_bf16in(conv, x: Tensor) -> Tensor:
    """The generator's 64 -> 3 output conv on a bf16 input (``srx_conv2d_fwd_bf16in``; inference, precision 2); fp32 out."""
    st = conv._st
    x = _chk16(x, 'conv2d_bf16in.input')
    n, h, w, cs = x.shape
    if st.precision != 2 or torch.is_grad_enabled():
        return conv(to_f32(x))
    d = st.desc(n, h, w)
    y = torch.empty(st.out_shape(n, h, w), dtype=torch.float32, device=x.device)
    b = None if conv.bias is None else _chk(conv.bias.detach(), 'conv2d.bias')
    if st.k == 9 and st.pad == 4 and st.stride == 1 and st.cin == 64 and st.cout <= 3 and not _dev.NO_T9:
        # the taps as the GEMM's N (csrc/thin9.hip): 32x32x16 bf16 MFMAs instead of 4x4x4
        key = st.pack_key(conv.weight) + (None if b is None else (b.data_ptr(), conv.bias._version),)
        if conv.__dict__.get('_t9_key') != key:
            wpk = conv.__dict__.get('_t9_wpk')
            if wpk is None or wpk.device != x.device:
                wpk = conv.__dict__['_t9_wpk'] = torch.empty(_lib.lib().srx_conv9x9_c64_thin_bf16_packed_bytes(), dtype=torch.uint8,
                                                             device=x.device)
            call('srx_conv9x9_c64_thin_bf16_pack', _p(_chk(conv.weight.detach(), 'conv2d.weight')), _p(b), st.cout, _p(wpk), _stream())
            conv.__dict__['_t9_key'] = key
        call('srx_conv9x9_c64_thin_bf16_fwd', n, h, w, _p(x), _p(conv.__dict__['_t9_wpk']), _p(y), _stream())
        return y
    st.pack(conv.weight, d)
    call('srx_conv2d_fwd_bf16in', C.byref(d), _p(x), _p(st.wpk_fwd), _p(b), _p(y), _stream())
    return y


def _ws(n: int, like: Tensor) -> Tensor:
    return torch.empty(max(int(n), 4), dtype=torch.float32, device=like.device)


def round4(c: int) -> int:
    return (c + 3) // 4 * 4


# --------------------------------------------------------------------------- layout
class _ToNHWC(Function):
    @staticmethod
    def forward(ctx, x: Tensor, cs: int) -> Tensor:
        ctx.set_materialize_grads(False)
        x = _chk(x, 'to_nhwc')
        n, c, h, w = x.shape
        ctx.c = c
        y = torch.empty((n, h, w, cs), dtype=torch.float32, device=x.device)
        call('srx_nchw_to_nhwc', _p(x), _p(y), n, c, h, w, cs, _stream())
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = _chk(dy, 'to_nhwc.bwd')
        n, h, w, cs = dy.shape
        dx = torch.empty((n, ctx.c, h, w), dtype=torch.float32, device=dy.device)
        call('srx_nhwc_to_nchw', _p(dy), _p(dx), n, ctx.c, h, w, cs, _stream())
        return dx, None


class _ToNCHW(Function):
    @staticmethod
    def forward(ctx, x: Tensor, c: int) -> Tensor:
        ctx.set_materialize_grads(False)
        x = _chk(x, 'to_nchw')
        n, h, w, cs = x.shape
        ctx.cs = cs
        y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
        call('srx_nhwc_to_nchw', _p(x), _p(y), n, c, h, w, cs, _stream())
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = _chk(dy, 'to_nchw.bwd')
        n, c, h, w = dy.shape
        dx = torch.empty((n, h, w, ctx.cs), dtype=torch.float32, device=dy.device)
        call('srx_nchw_to_nhwc', _p(dy), _p(dx), n, c, h, w, ctx.cs, _stream())
        return dx, None


def to_nhwc(x: Tensor, cs: Optional[int] = None) -> Tensor:
    """NCHW ``[N,C,H,W]`` -> NHWC ``[N,H,W,cs]`` (extra channels zero)."""
    return _ToNHWC.apply(x, round4(x.shape[1]) if cs is None else cs)


def to_nchw(x: Tensor, c: Optional[int] = None) -> Tensor:
    """NHWC ``[N,H,W,cs]`` -> NCHW ``[N,c,H,W]`` (drops padding channels)."""
    return _ToNCHW.apply(x, x.shape[3] if c is None else c)


def flatten_nchw(x: Tensor) -> Tensor:
    """``torch.flatten(out, 1)`` of the reference (srgan/discriminator.py:86): (c,h,w) order."""
    return to_nchw(x).reshape(x.shape[0], -1)


# --------------------------------------------------------------------------- conv2d
class ConvState:
    """Per-layer host state: geometry descriptors and packed weight copies."""

    def __init__(self, cin, cout, k, stride, pad, shuffle=0, act=ACT_NONE, slope=0.0, up=0):
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.shuffle, self.act, self.slope = shuffle, act, float(slope)
        self.up = 2 if up == 2 else 0  # nearest x2 upsampling of the input fused into the gather (srx_conv2d_t::up)
        self.cin_s = round4(cin)
        self.cout_s = round4(cout // 4) if shuffle else round4(cout)
        self._descs = {}
        self.model_epoch = [0]  # replaced by the owning FlatParams' counter
        self.precision = 0  # 1: bf16 products in the forward / stride-1 data gradient (srx_conv2d_t::precision)
        self.wpk_fwd = None
        self.wpk_bwd = None
        self._key = None
        self.fused_only = False  # set while the layer runs inside a fused multi-conv kernel that has its own weight stream

    def desc(self, n, h, w) -> Conv2dDesc:
        d = self._descs.get((n, h, w, self.precision))
        if d is None:
            d = Conv2dDesc(n, h, w, self.cin, self.cin_s, self.cout, self.cout_s, self.k, self.k, self.stride,
                           self.pad, self.shuffle, self.act, self.slope, self.up, self.precision)
            self._descs[(n, h, w, self.precision)] = d
        return d

    def out_rows(self, n, h, w) -> int:
        """Rows of the output matrix (pixels the conv computes; before any PixelShuffle)."""
        uf = 2 if self.up == 2 else 1
        ho = (uf * h + 2 * self.pad - self.k) // self.stride + 1
        wo = (uf * w + 2 * self.pad - self.k) // self.stride + 1
        return n * ho * wo

    def out_shape(self, n, h, w):
        uf = 2 if self.up == 2 else 1
        ho = (uf * h + 2 * self.pad - self.k) // self.stride + 1
        wo = (uf * w + 2 * self.pad - self.k) // self.stride + 1
        if self.shuffle:
            return (n, 2 * ho, 2 * wo, self.cout_s)
        return (n, ho, wo, self.cout_s)

    def pack_key(self, weight: Tensor):
        """What the packed copies were made from: the tensor, its in-place version, the global epoch (bumped
        when raw-pointer code may have changed any parameter) and the model's own epoch (bumped by its
        optimiser, ``optim.FlatParams.pack_epoch``).  Frozen parameters (VGG19) never repack."""
        if not weight.requires_grad:
            return (weight.data_ptr(), weight._version, -1, -1, self.precision)
        return (weight.data_ptr(), weight._version, _pack_epoch[0], self.model_epoch[0], self.precision)

    def pack(self, weight: Tensor, d: Conv2dDesc, force: bool = False) -> None:
        """(Re)build the packed copies when the master OIHW weight changed.

        ``weight`` is the module's Parameter: its ``_version`` catches in-place torch updates,
        the optimiser epoch catches raw-pointer updates by ``srx_adam_step`` (frozen parameters
        such as VGG19's never repack)."""
        key = self.pack_key(weight)
        if not force and key == self._key and self.wpk_fwd is not None:
            return
        dref = C.byref(d)
        if self.wpk_fwd is None or self.wpk_fwd.device != weight.device:
            self.wpk_fwd = torch.empty(_lib.lib().srx_conv2d_packed_fwd_floats(dref), dtype=torch.float32,
                                       device=weight.device)
            self.wpk_bwd = torch.empty(max(_lib.lib().srx_conv2d_packed_bwd_floats(dref), 4), dtype=torch.float32,
                                       device=weight.device)
        w = _chk(weight.detach(), 'conv2d.weight')
        call('srx_conv2d_pack', dref, _p(w), _p(self.wpk_fwd), _p(self.wpk_bwd), _stream())
        self._key = key
        self.last_desc = d

    def pack_wino(self, weight: Tensor, d: Conv2dDesc, need_bwd: bool) -> None:
        """The Winograd-domain weights (``srx_wino_pack``: U = G g G^T of the layer and, for its data gradient, of the
        transposed / tap-flipped layer), rebuilt when the master weight changed -- same key as ``pack``."""
        key = self.pack_key(weight)
        fresh = key == self.__dict__.get('_wino_key')
        if fresh and (self.__dict__.get('wino_bwd') is not None or not need_bwd):
            return
        dref = C.byref(d)
        n = _lib.lib().srx_wino_packed_floats(dref)
        w = _chk(weight.detach(), 'conv2d.weight')
        if not fresh or self.__dict__.get('wino_fwd') is None:
            if self.__dict__.get('wino_fwd') is None or self.wino_fwd.device != weight.device:
                self.wino_fwd = torch.empty(n, dtype=torch.float32, device=weight.device)
                self.wino_bwd = None
            call('srx_wino_pack', dref, _p(w), _p(self.wino_fwd), 0, _stream())
            if self.wino_bwd is not None:
                call('srx_wino_pack', dref, _p(w), _p(self.wino_bwd), 1, _stream())
        if need_bwd and self.wino_bwd is None:
            self.wino_bwd = torch.empty(n, dtype=torch.float32, device=weight.device)
            call('srx_wino_pack', dref, _p(w), _p(self.wino_bwd), 1, _stream())
        self._wino_key = key
        self._wino_desc = d

    def pack_bf16s(self, weight: Tensor, d: Conv2dDesc, need_bwd: bool) -> None:
        """bf16 weight copies of a layer inside a bf16-storage stack (``srx_conv3x3_bf16s_pack``), same key as ``pack``."""
        key = self.pack_key(weight)
        have = self.__dict__.get('bf16s_fwd')
        if key == self.__dict__.get('_bf16s_key') and have is not None and (self.__dict__.get('bf16s_bwd') is not None or not need_bwd):
            return
        dref = C.byref(d)
        n = _lib.lib().srx_conv3x3_bf16s_packed_bytes(dref) // 2
        if have is None or have.device != weight.device:
            self.bf16s_fwd = torch.empty(n, dtype=torch.bfloat16, device=weight.device)
            self.bf16s_bwd = None
        if need_bwd and self.__dict__.get('bf16s_bwd') is None:
            self.bf16s_bwd = torch.empty(n, dtype=torch.bfloat16, device=weight.device)
        w = _chk(weight.detach(), 'conv2d.weight')
        call('srx_conv3x3_bf16s_pack', dref, _p(w), _p(self.bf16s_fwd), _p(self.bf16s_bwd), _stream())
        self._bf16s_key = key


def wino_forward_only_ok(st: ConvState, d: Conv2dDesc) -> bool:
    """A sub-pixel conv (3x3 + nn.PixelShuffle(2), srgan/residual.py:27-28) in exact fp32: its FORWARD runs on Winograd with the
    shuffle in the store (csrc/wino.hip); its data gradient (a gather from the shuffled gradient) keeps the direct kernel."""
    return (not _dev.NO_WINO and st.shuffle == 2 and st.act == ACT_NONE and st.precision == 0
            and _lib.lib().srx_wino_infer_applicable(C.byref(d)) == 1)


def wino_layer_ok(st: ConvState, d: Conv2dDesc) -> bool:
    """Does this layer's forward (and data gradient) run on Winograd F(2x2, 3x3) (csrc/wino.hip) at this size?  Wide 3x3 /
    stride 1 / pad 1 fp32 layers without a fused LeakyReLU; small 64 -> 64 layers keep the row-tile kernel."""
    return (not _dev.NO_WINO and st.act in (ACT_NONE, ACT_RELU) and _lib.lib().srx_wino_applicable(C.byref(d)) == 1)


def wino_forward_runs(st: ConvState, d: Conv2dDesc, want_stats: bool) -> bool:
    """THE decision whether this forward call runs on the Winograd kernel -- shared by ``_Conv2d.forward`` and
    ``conv_stat_tile_rows`` so that the row granularity of a partial-statistics table is always the one of the kernel that
    writes it: the statistics epilogue exists for linear layers only (``srx_wino_fwd_stats``)."""
    return wino_layer_ok(st, d) and (not want_stats or st.act == ACT_NONE)


class PackTable:
    """Every conv of a model repacked by ONE launch after an optimiser step (``srx_pack_table_*``).

    Built once all layers have seen an input (their packed buffers and descriptors exist); the record
    table lives in device memory.  ``run()`` also stamps each layer's pack key, so the lazy per-layer
    ``ConvState.pack`` in the next forward finds nothing to do.
    """

    def __init__(self, convs):
        self.convs = list(convs)
        self.table = None
        self.nrec, self.maxn = 0, 0
        self.fused_sig = None
        self.generation = 0  # bumped by every (re)build

    def _build(self) -> bool:
        # (convs whose packed fp32 copies nothing reads -- dense blocks running as fused bf16 launches, RDBPack -- are left out)
        every = [(c._st, c.weight) for c in self.convs if c.weight.requires_grad and not c._st.fused_only]
        # a layer is ready once it has run: it then has its direct packs (gconv / thin / row-tile kernels), its Winograd-domain
        # copies (wino.hip: such a layer needs no direct pack at all), or both (it took both paths at different sizes)
        direct = [(st, wt) for st, wt in every if st.wpk_fwd is not None and getattr(st, 'last_desc', None) is not None]
        wino = [(st, wt) for st, wt in every if st.__dict__.get('wino_fwd') is not None]
        seen = {id(st) for st, _ in direct} | {id(st) for st, _ in wino}
        if not direct or any(id(st) not in seen for st, _ in every):
            return False
        items = direct
        n = len(items)
        descs = (Conv2dDesc * n)(*[st.last_desc for st, _ in items])
        arr = lambda ptrs: (C.c_void_p * n)(*ptrs)  # noqa: E731
        w, f, b = (arr([wt.data_ptr() for _, wt in items]), arr([st.wpk_fwd.data_ptr() for st, _ in items]),
                   arr([st.wpk_bwd.data_ptr() for st, _ in items]))
        nbytes = _lib.lib().srx_pack_table_bytes(n + len(wino))
        host = torch.empty(nbytes, dtype=torch.uint8)
        nrec, maxn = C.c_int(0), C.c_longlong(0)
        call('srx_pack_table_build', descs, n, w, f, b, host.data_ptr(), C.byref(nrec), C.byref(maxn))
        # layers that run on Winograd: their transformed weights are refreshed by the same launch
        self.wino_items = wino
        for st, wt in self.wino_items:
            dref = C.byref(st._wino_desc)
            call('srx_pack_table_add_wino', host.data_ptr(), C.byref(nrec), C.byref(maxn), dref, wt.data_ptr(), st.wino_fwd.data_ptr(), 0)
            if st.wino_bwd is not None:
                call('srx_pack_table_add_wino', host.data_ptr(), C.byref(nrec), C.byref(maxn), dref, wt.data_ptr(), st.wino_bwd.data_ptr(), 1)
        self.table = host.to(items[0][1].device)
        self.items, self.nrec, self.maxn = items, nrec.value, maxn.value
        self.fused_sig = self._fused_signature()
        self.generation += 1
        return True

    def _fused_signature(self):
        """what decides the table's records: which convs are left out (fused dense blocks) and which carry Winograd copies"""
        return tuple((bool(c._st.fused_only), c._st.wpk_fwd is not None, c._st.__dict__.get('wino_fwd') is not None,
                      c._st.__dict__.get('wino_bwd') is not None) for c in self.convs)

    def run(self) -> bool:
        """False (and nothing done) until every layer is ready: the lazy per-layer path still covers that."""
        # ``fused_only`` is re-decided per forward (precision, developer switch): a table built while the dense blocks ran
        # fused leaves their convs out, and would leave their fp32 packs stale after an optimiser step once they run unfused
        if self.table is not None and self.fused_sig != self._fused_signature():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('PackTable: the set of fused-only convs changed inside a hipGraph capture; run one eager step '
                                   'after switching precision / SRX_NO_RDB_FUSED')
            self.table = None  # rebuilt below; ``generation`` tells holders of captured graphs to drop them
        if self.table is None and (torch.cuda.is_current_stream_capturing() or not self._build()):
            return False
        call('srx_pack_table_run', self.table.data_ptr(), self.nrec, self.maxn, _stream())
        for st, wt in self.items:
            st._key = st.pack_key(wt)
        for st, wt in self.wino_items:
            st._wino_key = st.pack_key(wt)
        return True


# bumped by the optimiser (optim.FlatAdam.step) because a raw-pointer update does not
# touch ``Tensor._version``
_pack_epoch = [0]


def bump_pack_epoch() -> None:
    _pack_epoch[0] += 1


class ActFold:
    """Token that pairs a PRODUCER conv whose fused ReLU / LeakyReLU backward is skipped (``act_bwd_folded=token``) with the
    ONE consumer conv whose data gradient applies it instead (``in_act=token``: ``srx_conv2d_bwd_data_act``, mask = the
    consumer's input).  The consumer's backward marks the token; the producer's backward refuses to run with an unmarked
    one -- a producer output that reached another consumer, a consumer without an input gradient, or a consumer on a kernel
    without the masked epilogue would otherwise give silently wrong gradients."""
    __slots__ = ('act', 'slope', 'masked', 'consumers')

    def __init__(self, act: int, slope: float):
        self.act, self.slope, self.masked, self.consumers = act, float(slope), False, 0


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], st: ConvState, want_stats: bool,
                master: Tensor, in_act: Optional[ActFold] = None, act_bwd_folded: Optional[ActFold] = None):
        # in_act: x is the output of a fused ReLU / LeakyReLU whose backward this conv's data gradient applies in its
        # epilogue (srx_conv2d_bwd_data_act, mask = x); the producer is called with the SAME token as act_bwd_folded
        # and skips the elementwise pass (a read of two tensors and a write of one, 226 MB for the discriminator's first
        # layer at batch 32).  Only for a producer whose output feeds this conv and nothing else (see ActFold).
        ctx.set_materialize_grads(False)
        if in_act is not None:
            if not isinstance(in_act, ActFold):
                raise TypeError('conv2d: in_act must be the ActFold token the producer was called with')
            in_act.consumers += 1
            if in_act.consumers > 1:
                raise RuntimeError('conv2d: an ActFold token has one consumer; this producer output feeds a second conv')
        if act_bwd_folded is not None and not isinstance(act_bwd_folded, ActFold):
            raise TypeError('conv2d: act_bwd_folded must be an ActFold token shared with the consumer')
        ctx.in_act, ctx.act_bwd_folded = in_act, act_bwd_folded
        x = _chk(x, 'conv2d.input')
        n, h, w, cs = x.shape
        if cs != st.cin_s:
            raise RuntimeError(f'conv2d: input has {cs} channels (stride), layer expects {st.cin_s}')
        d = st.desc(n, h, w)
        dref = C.byref(d)
        L = _lib.lib()
        y = torch.empty(st.out_shape(n, h, w), dtype=torch.float32, device=x.device)
        b = None if bias is None else _chk(bias.detach(), 'conv2d.bias')
        # wide 3x3 layers: Winograd F(2x2, 3x3), 2.25x fewer fp32 multiplications (csrc/wino.hip) -- the VGG19 features called
        # layer by layer, the discriminators' stride-1 layers (with the BatchNorm partial statistics in the epilogue)
        fwd_only = not want_stats and wino_forward_only_ok(st, d)
        ctx.wino = fwd_only or wino_forward_runs(st, d, want_stats)
        part = None
        if ctx.wino:
            # (trainable layers: PackTable refreshes the Winograd-domain weights inside the captured step)
            st.pack_wino(master, d, need_bwd=ctx.needs_input_grad[0] and not fwd_only)
            if want_stats:
                part = torch.empty((L.srx_wino_stat_rows(dref), st.cout, 2), dtype=torch.float32, device=x.device)
                nws = L.srx_wino_ws_floats(dref, 2)
                call('srx_wino_fwd_stats', dref, _p(x), _p(st.wino_fwd), _p(b), _p(y), _p(part), _p(_ws(nws, x)) if nws else None, nws,
                     _stream())
            else:
                nws = L.srx_wino_ws_floats(dref, 0)
                call('srx_wino_fwd', dref, _p(x), _p(st.wino_fwd), _p(b), _p(y), _p(_ws(nws, x)) if nws else None, nws, _stream())
        else:
            st.pack(master, d)
            if want_stats:
                rows = L.srx_conv2d_stat_rows(dref)
                part = torch.empty((rows, st.cout, 2), dtype=torch.float32, device=x.device)
            nws = L.srx_conv2d_fwd_ws_floats(dref)
            ws = _ws(nws, x) if nws else None
            call('srx_conv2d_fwd', dref, _p(x), _p(st.wpk_fwd), _p(b), _p(y), _p(part), _p(ws), nws, _stream())
        ctx.master = master
        ctx.st, ctx.d = st, d
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)
        ctx.save_for_backward(x, y if st.act != ACT_NONE else None)
        ctx.wpk_bwd = st.wpk_bwd
        if want_stats:
            ctx.mark_non_differentiable(part)
            return y, part
        return y, None

    @staticmethod
    def backward(ctx, dy: Tensor, _dpart):
        st, d = ctx.st, ctx.d
        dref = C.byref(d)
        L = _lib.lib()
        x, y = ctx.saved_tensors
        dy = _chk(dy, 'conv2d.grad')
        s = _stream()
        if ctx.act_bwd_folded is not None and not ctx.act_bwd_folded.masked:
            raise RuntimeError('conv2d: this layer\'s activation backward was handed to a consumer (ActFold) whose data '
                               'gradient never ran with the mask: the gradients would be wrong')
        if st.act != ACT_NONE and ctx.act_bwd_folded is None:
            g = torch.empty_like(dy)
            call('srx_act_bwd_from_out', _p(dy), _p(y), _p(g), dy.numel(), st.act, st.slope, s)
            dy = g
        dx = dw = db = None
        if ctx.needs_input_grad[0] and ctx.wino and (ctx.in_act is None or ctx.in_act.act == ACT_RELU) and st.wino_bwd is not None:
            dx = torch.empty_like(x)
            nws = L.srx_wino_ws_floats(dref, 1)
            call('srx_wino_bwd_data', dref, _p(dy), _p(st.wino_bwd), _p(x) if ctx.in_act is not None else None, _p(dx),
                 _p(_ws(nws, x)) if nws else None, nws, s)
            if ctx.in_act is not None:
                ctx.in_act.masked = True
        elif ctx.needs_input_grad[0]:
            if ctx.wino:  # (a Winograd forward whose data gradient takes the direct kernel: its packed copy is made now)
                st.pack(ctx.master, d)
                ctx.wpk_bwd = st.wpk_bwd
            dx = torch.empty_like(x)
            nws = L.srx_conv2d_bwd_data_ws_floats(dref)
            ws = _ws(nws, x) if nws else None
            if ctx.in_act is not None:  # ... and the backward of the activation that produced x
                slope = 0.0 if ctx.in_act.act == ACT_RELU else ctx.in_act.slope
                call('srx_conv2d_bwd_data_act', dref, _p(dy), _p(ctx.wpk_bwd), _p(x), slope, 0, st.cin_s, 0, _p(dx), _p(ws), nws, s)
                ctx.in_act.masked = True
            else:
                call('srx_conv2d_bwd_data', dref, _p(dy), _p(ctx.wpk_bwd), _p(dx), 0, _p(ws), nws, s)
        wparam, bparam = ctx.params
        bias_done = False
        if ctx.needs_input_grad[1]:
            sink = _sink(wparam)
            dw = None if sink is not None else torch.empty((st.cout, st.cin, st.k, st.k), dtype=torch.float32,
                                                             device=x.device)
            # the bias gradient rides along in the weight-gradient kernel (it stages every dy row anyway)
            # when both results go the same way: both accumulated into .grad, or both returned
            bptr = None
            if ctx.has_bias and ctx.needs_input_grad[2]:  # (PixelShuffle layers too: the reduction maps packed columns back)
                bsink = _sink(bparam)
                if (bsink is None) == (sink is None):
                    if bsink is None:
                        db = torch.empty(st.cout, dtype=torch.float32, device=x.device)
                    bptr = _p(db if bsink is None else bsink)
                    bias_done = True
            queue = wgrad_queue[0]
            if queue is not None and sink is not None:
                queue.add(d, _p(x), _p(dy), _p(sink), bptr, (x, dy))
            else:
                nws = L.srx_conv2d_bwd_weight_ws_floats(dref)
                ws = _ws(nws, x)
                call('srx_conv2d_bwd_weight', dref, _p(x), _p(dy), _p(dw if sink is None else sink),
                     0 if sink is None else 1, bptr, _p(ws), nws, s)
        if ctx.has_bias and ctx.needs_input_grad[2] and not bias_done:
            sink = None if st.shuffle else _sink(bparam)
            if sink is not None:
                m = dy.numel() // st.cout_s
                nws = L.srx_colsum_ws_floats(m, st.cout)
                call('srx_colsum', _p(dy), _p(sink), m, st.cout, st.cout_s, 1, _p(_ws(nws, dy)), nws, s)
            elif st.shuffle:
                # dy is [N, 2Ho, 2Wo, cps]; bias index co = c*4 + i*2 + j
                n, h2, w2, cps = dy.shape
                rows, cols = n * (h2 // 2), 2 * w2 * cps
                t = torch.empty(cols, dtype=torch.float32, device=dy.device)
                nws = L.srx_colsum_ws_floats(rows, cols)
                call('srx_colsum', _p(dy), _p(t), rows, cols, cols, 0, _p(_ws(nws, dy)), nws, s)
                # t is [i][wo][j][c]; fold wo (tiny tensor, plumbing) and permute to (c,i,j)
                db = t.view(2, w2 // 2, 2, cps).sum(1).permute(2, 0, 1).reshape(-1)[:st.cout].contiguous()
            else:
                m = dy.numel() // st.cout_s
                db = torch.empty(st.cout, dtype=torch.float32, device=dy.device)
                nws = L.srx_colsum_ws_floats(m, st.cout)
                call('srx_colsum', _p(dy), _p(db), m, st.cout, st.cout_s, 0, _p(_ws(nws, dy)), nws, s)
        return dx, dw, db, None, None, None, None, None


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], st: ConvState, want_stats: bool = False,
           master: Optional[Tensor] = None, in_act: Optional[ActFold] = None,
           act_bwd_folded: Optional[ActFold] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """NHWC conv.  Returns ``(y, bn_partials)``; ``bn_partials`` is ``None`` unless requested.  ``in_act`` /
    ``act_bwd_folded``: ONE shared ``ActFold`` token for a producer / consumer pair of convs (see ``_Conv2d.forward``)."""
    return _Conv2d.apply(x, weight, bias, st, want_stats, weight if master is None else master, in_act, act_bwd_folded)


# --------------------------------------------------------------------------- batch norm (+act, +residual)
class _BNAct(Function):
    @staticmethod
    def forward(ctx, y: Tensor, part: Optional[Tensor], gamma: Tensor, beta: Tensor, prelu: Optional[Tensor],
                residual: Optional[Tensor], running_mean: Tensor, running_var: Tensor, nbt: Optional[Tensor],
                training: bool, eps: float, momentum: float, act: int, slope: float, groups: int):
        ctx.set_materialize_grads(False)
        y = _chk(y, 'bn.input')
        c = y.shape[-1]
        m = y.numel() // c
        s = _stream()
        if not training:
            groups = 1  # eval: one set of running statistics for every row
        mean = torch.empty(groups * c, dtype=torch.float32, device=y.device)
        invstd = torch.empty(groups * c, dtype=torch.float32, device=y.device)
        g = _chk(gamma.detach(), 'bn.weight')
        b = _chk(beta.detach(), 'bn.bias')
        pw = None if prelu is None else _chk(prelu.detach(), 'prelu.weight')
        res = None if residual is None else _chk(residual, 'bn.residual')
        out = torch.empty_like(y)
        if training:
            if part is None:
                rows = _lib.lib().srx_bn_stat_rows(m)
                part = torch.empty((rows, c, 2), dtype=torch.float32, device=y.device)
                call('srx_bn_partial_stats', _p(y), _p(part), m, c, s)
            call('srx_bn_train_fwd', _p(y), _p(part), part.shape[0], m, c, groups, eps, momentum, _p(g), _p(b), _p(res),
                 _p(out), act, slope, _p(pw), _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(nbt), s)
        else:
            call('srx_bn_eval_stats', _p(running_mean), _p(running_var), c, eps, _p(mean), _p(invstd), s)
            call('srx_bn_act_fwd', _p(y), _p(mean), _p(invstd), _p(g), _p(b), _p(res), _p(out), m, c, act, slope, _p(pw),
                 s)
        ctx.save_for_backward(y, mean, invstd, g, b, pw)
        ctx.cfg = (m, c, act, slope, training, residual is not None, prelu is not None, groups)
        ctx.params = (gamma, beta, prelu)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        y, mean, invstd, g, b, pw = ctx.saved_tensors
        m, c, act, slope, training, has_res, has_prelu, groups = ctx.cfg
        dout = _chk(dout, 'bn.grad')
        s = _stream()
        sums = torch.empty(groups * (2 * c + 4), dtype=torch.float32, device=y.device)
        nws = _lib.lib().srx_bn_bwd_ws_floats(m, c)
        gs, bs, ps = (_sink(t) for t in ctx.params)
        dy = torch.empty_like(y) if ctx.needs_input_grad[0] else None
        call('srx_bn_act_bwd', _p(dout), _p(y), _p(mean), _p(invstd), _p(g), _p(b), _p(sums), _p(dy), m, c, groups, act,
             slope, _p(pw), 1 if training else 0, _p(gs), _p(bs), _p(ps), _p(_ws(nws, y)), nws, s)
        want = ((ctx.needs_input_grad[2] and gs is None), (ctx.needs_input_grad[3] and bs is None),
                (has_prelu and ctx.needs_input_grad[4] and ps is None))
        dgamma = dbeta = dprelu = None
        if any(want):  # gradients returned to autograd instead of accumulated by the kernel: total over the groups
            per = sums.view(groups, 2 * c + 4)
            tot = per[0] if groups == 1 else per.sum(0)
            dgamma = tot[c:2 * c] if want[0] else None
            dbeta = tot[:c] if want[1] else None
            dprelu = tot[2 * c:2 * c + 1] if want[2] else None
        dres = dout if (has_res and ctx.needs_input_grad[5]) else None
        return (dy, None, dgamma, dbeta, dprelu, dres) + (None,) * 9


def bn_groups_ok(m: int, tile_rows: Optional[int], groups: int) -> bool:
    """Can ``m`` rows be normalised as ``groups`` independent row ranges?  ``tile_rows``: rows per entry of the conv
    epilogue's partial-statistics table (``conv_stat_tile_rows``; ``None``: the statistics come from
    ``srx_bn_partial_stats``).  No conv tile and no row block of the streaming reductions may straddle two groups."""
    if groups == 1:
        return True
    if m % groups:
        return False
    per = m // groups
    if per % _lib.lib().srx_bn_rows_per_block(m):
        return False
    return tile_rows is None or per % tile_rows == 0


def conv_stat_tile_rows(st: ConvState, n: int, h: int, w: int) -> int:
    """Output rows (pixels) summarised by one row of the partial-statistics table ``conv2d(..., want_stats=True)``
    returns for this layer at this input size: the tile height of the launch plan."""
    if wino_forward_runs(st, st.desc(n, h, w), True):
        return 128  # a Winograd tile block: 32 consecutive 2x2 tiles (image-major)
    out = (C.c_int * 6)()
    call('srx_conv2d_plan', C.byref(st.desc(n, h, w)), 0, out)
    return int(out[0])


def bn_act(y, part, bn, act=ACT_NONE, slope=0.0, prelu: Optional[Tensor] = None,
           residual: Optional[Tensor] = None, frozen: bool = False, groups: int = 1) -> Tensor:
    """``act(BatchNorm2d(y)) [+ residual]`` with ``bn`` an ``nn.BatchNorm2d``-compatible module.  ``groups``: the batch
    is that many forward calls of the reference run together (independent batch statistics, see norm.hip)."""
    training = bn.training or bn.running_mean is None
    momentum = 0.1 if bn.momentum is None else bn.momentum
    gamma, beta = (bn.weight.detach(), bn.bias.detach()) if frozen else (bn.weight, bn.bias)
    return _BNAct.apply(y, part, gamma, beta, prelu, residual, bn.running_mean, bn.running_var,
                        bn.num_batches_tracked if training else None, training, bn.eps, momentum, act, float(slope),
                        int(groups))


class _ResidualBlock(Function):
    """SRGAN's ``ResidualBlock.forward`` -- ``x + BN2(conv2(PReLU(BN1(conv1(x)))))`` (srgan/residual.py:86-91) -- as ONE
    autograd node in training mode.  Same kernels as the layer-by-layer path; what the node buys is the backward
    pass: autograd would add the skip connection's gradient to conv1's input gradient in a pass of its own (17 such
    adds per generator backward); here that sum is the epilogue of conv1's data gradient
    (``srx_conv2d_bwd_data_add``).  Parameter gradients accumulate straight into the flat ``.grad`` buffers; the
    conv weight gradients go to the ``WeightGradQueue`` when one is installed.
    """

    @staticmethod
    def forward(ctx, x: Tensor, block, *params):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'residual_block.input')
        n, h, w, c = x.shape
        m = n * h * w
        L, s = _lib.lib(), _stream()
        convs, bns = (block.conv1, block.conv2), (block.bn1, block.bn2)
        descs, ys, stats = [], [], []
        inp = x
        for i in range(2):
            st, bn = convs[i]._st, bns[i]
            d = st.desc(n, h, w)
            dref = C.byref(d)
            st.pack(convs[i].weight, d)
            y = torch.empty_like(x)
            part = torch.empty((L.srx_conv2d_stat_rows(dref), c, 2), dtype=torch.float32, device=x.device)
            nws = L.srx_conv2d_fwd_ws_floats(dref)
            call('srx_conv2d_fwd', dref, _p(inp), _p(st.wpk_fwd), None, _p(y), _p(part), _p(_ws(nws, x)) if nws else None, nws, s)
            mean = torch.empty(c, dtype=torch.float32, device=x.device)
            invstd = torch.empty(c, dtype=torch.float32, device=x.device)
            out = torch.empty_like(x)
            momentum = 0.1 if bn.momentum is None else bn.momentum
            g, b = bn.weight.detach(), bn.bias.detach()
            if i == 0:   # BN1 + PReLU
                call('srx_bn_train_fwd', _p(y), _p(part), part.shape[0], m, c, 1, bn.eps, momentum, _p(g), _p(b), None, _p(out),
                     ACT_PRELU, 0.0, _p(block.prelu.weight.detach()), _p(mean), _p(invstd), _p(bn.running_mean),
                     _p(bn.running_var), _p(bn.num_batches_tracked), s)
            else:        # BN2 + skip connection
                call('srx_bn_train_fwd', _p(y), _p(part), part.shape[0], m, c, 1, bn.eps, momentum, _p(g), _p(b), _p(x), _p(out),
                     ACT_NONE, 0.0, None, _p(mean), _p(invstd), _p(bn.running_mean), _p(bn.running_var),
                     _p(bn.num_batches_tracked), s)
            descs.append(d)
            ys.append(y)
            stats += [mean, invstd]
            if i == 0:
                a1 = inp = out
        ctx.block, ctx.descs = block, descs
        ctx.packs = (convs[0]._st.wpk_bwd, convs[1]._st.wpk_bwd)
        ctx.save_for_backward(x, ys[0], a1, ys[1], *stats)
        return out

    @staticmethod
    def backward(ctx, dout: Tensor):
        x, y1, a1, y2, mean1, inv1, mean2, inv2 = ctx.saved_tensors
        block = ctx.block
        dout = _chk(dout, 'residual_block.grad')
        n, h, w, c = x.shape
        m = n * h * w
        L, s = _lib.lib(), _stream()
        queue = wgrad_queue[0]

        def bn_bwd(dz_in, y, mean, invstd, bn, act, prelu):
            sums = torch.empty(2 * c + 4, dtype=torch.float32, device=x.device)
            dy = torch.empty_like(y)
            nws = L.srx_bn_bwd_ws_floats(m, c)
            pw = None if prelu is None else prelu.detach()
            call('srx_bn_act_bwd', _p(dz_in), _p(y), _p(mean), _p(invstd), _p(bn.weight.detach()), _p(bn.bias.detach()),
                 _p(sums), _p(dy), m, c, 1, act, 0.0, _p(pw), 1, _p(bn.weight.grad), _p(bn.bias.grad),
                 None if prelu is None else _p(prelu.grad), _p(_ws(nws, y)), nws, s)
            return dy

        def wgrad(conv, d, inp, dy):
            if queue is not None:
                queue.add(d, _p(inp), _p(dy), _p(conv.weight.grad), None, (inp, dy))
                return
            dref = C.byref(d)
            nws = L.srx_conv2d_bwd_weight_ws_floats(dref)
            call('srx_conv2d_bwd_weight', dref, _p(inp), _p(dy), _p(conv.weight.grad), 1, None, _p(_ws(nws, inp)), nws, s)

        dy2 = bn_bwd(dout, y2, mean2, inv2, block.bn2, ACT_NONE, None)
        wgrad(block.conv2, ctx.descs[1], a1, dy2)
        da1 = torch.empty_like(x)
        dref = C.byref(ctx.descs[1])
        nws = L.srx_conv2d_bwd_data_ws_floats(dref)
        call('srx_conv2d_bwd_data', dref, _p(dy2), _p(ctx.packs[1]), _p(da1), 0, _p(_ws(nws, x)) if nws else None, nws, s)
        dy1 = bn_bwd(da1, y1, mean1, inv1, block.bn1, ACT_PRELU, block.prelu.weight)
        wgrad(block.conv1, ctx.descs[0], x, dy1)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            dref = C.byref(ctx.descs[0])
            nws = L.srx_conv2d_bwd_data_ws_floats(dref)
            call('srx_conv2d_bwd_data_add', dref, _p(dy1), _p(ctx.packs[0]), _p(dout), _p(dx), _p(_ws(nws, x)) if nws else None,
                 nws, s)
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


class _ResidualTower(Function):
    """The chain of SRGAN ``ResidualBlock``s (srgan/generator.py:42-45,76) as ONE autograd node: the same kernels as
    ``_ResidualBlock`` per block, plus what only a node that sees neighbouring layers can do -- the first pass of every
    BatchNorm backward (the per-channel sums of dz and dz * xhat) rides in the epilogue of the data gradient that
    PRODUCES that BatchNorm's output gradient (``srx_conv2d_bwd_data_bn``): conv2's data gradient reduces for bn1 + PReLU
    of its own block, conv1's (plus the skip gradient) for bn2 of the block BEFORE.  32 of the 33 ``bn_bwd_reduce``
    launches of a generator backward pass (and their second read of two 2.4 MB tensors each) disappear.
    """

    @staticmethod
    def forward(ctx, x: Tensor, blocks, *params):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'residual_tower.input')
        n, h, w, c = x.shape
        m = n * h * w
        L, s = _lib.lib(), _stream()
        saved, descs = [], None
        # A normalise pass (bn1 + PReLU inside a block, bn2 + skip at its end) is not launched where the NEXT conv can apply it
        # while staging its input (srx_conv2d_fwd_bn_in: `pending` = (y, mean, invstd, gamma, beta, slope, residual, result));
        # the result tensor is written by that conv on the side.  Only the last block's bn2 + skip is a launch of its own.
        pending = None
        fuse = None
        for bi, block in enumerate(blocks):
            convs, bns = (block.conv1, block.conv2), (block.bn1, block.bn2)
            inp, ys, stats = x, [], []
            for i in range(2):
                st, bn = convs[i]._st, bns[i]
                d = st.desc(n, h, w)
                dref = C.byref(d)
                if fuse is None:
                    fuse = L.srx_conv2d_fwd_bn_in_ok(dref) == 1
                st.pack(convs[i].weight, d)
                y = torch.empty_like(x)
                part = torch.empty((L.srx_conv2d_stat_rows(dref), c, 2), dtype=torch.float32, device=x.device)
                if pending is not None:
                    py, pmean, pinv, pg, pb, pslope, pres, pout = pending
                    call('srx_conv2d_fwd_bn_in', dref, _p(py), _p(pmean), _p(pinv), _p(pg), _p(pb), _p(pslope), _p(pres), _p(pout),
                         _p(st.wpk_fwd), None, _p(y), _p(part), s)
                    pending = None
                else:
                    nws = L.srx_conv2d_fwd_ws_floats(dref)
                    call('srx_conv2d_fwd', dref, _p(inp), _p(st.wpk_fwd), None, _p(y), _p(part), _p(_ws(nws, x)) if nws else None, nws, s)
                mean = torch.empty(c, dtype=torch.float32, device=x.device)
                invstd = torch.empty(c, dtype=torch.float32, device=x.device)
                out = torch.empty_like(x)
                momentum = 0.1 if bn.momentum is None else bn.momentum
                g, b = bn.weight.detach(), bn.bias.detach()
                last = i == 1 and bi == len(blocks) - 1
                if fuse and not last:  # statistics only; the normalise pass rides in the next conv's launch
                    call('srx_bn_finalize', _p(part), part.shape[0], m, c, bn.eps, momentum, _p(mean), _p(invstd),
                         _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), s)
                    pending = (y, mean, invstd, g, b, block.prelu.weight.detach(), None, out) if i == 0 else \
                        (y, mean, invstd, g, b, None, x, out)
                elif i == 0:   # BN1 + PReLU
                    call('srx_bn_train_fwd', _p(y), _p(part), part.shape[0], m, c, 1, bn.eps, momentum, _p(g), _p(b), None, _p(out),
                         ACT_PRELU, 0.0, _p(block.prelu.weight.detach()), _p(mean), _p(invstd), _p(bn.running_mean),
                         _p(bn.running_var), _p(bn.num_batches_tracked), s)
                else:        # BN2 + skip connection
                    call('srx_bn_train_fwd', _p(y), _p(part), part.shape[0], m, c, 1, bn.eps, momentum, _p(g), _p(b), _p(x), _p(out),
                         ACT_NONE, 0.0, None, _p(mean), _p(invstd), _p(bn.running_mean), _p(bn.running_var),
                         _p(bn.num_batches_tracked), s)
                if i == 0:
                    a1 = inp = out
                ys.append(y)
                stats += [mean, invstd]
                descs = d
            saved += [x, ys[0], a1, ys[1], *stats]
            x = out
        ctx.blocks, ctx.desc = blocks, descs
        ctx.packs = [(b.conv1._st.wpk_bwd, b.conv2._st.wpk_bwd) for b in blocks]
        ctx.save_for_backward(*saved)
        return x

    @staticmethod
    def backward(ctx, dout: Tensor):
        saved, blocks, d = ctx.saved_tensors, ctx.blocks, ctx.desc
        grad = _chk(dout, 'residual_tower.grad')
        n, h, w, c = grad.shape
        m = n * h * w
        L, s = _lib.lib(), _stream()
        queue = wgrad_queue[0]
        dref = C.byref(d)
        rows = L.srx_conv2d_bwd_data_bn_rows(dref)  # 0: the layers do not run on the row-tile kernel at this size
        if _dev.NO_BN_DGRAD_FUSE:  # developer switch (A/B runs): separate reduce launches
            rows = 0
        W = 2 * c + 4
        nws_d = L.srx_conv2d_bwd_data_ws_floats(dref)

        def finish(dz_in, y, mean, invstd, bn, act, prelu, table):
            """dy of act(BN(y)) given dz_in; the reduce pass comes from `table` when the producer of dz_in filled one"""
            sums = torch.empty(W, dtype=torch.float32, device=grad.device)
            dy = torch.empty_like(y)
            pw = None if prelu is None else prelu.detach()
            gw, gb = bn.weight.detach(), bn.bias.detach()
            pg = None if prelu is None else _p(prelu.grad)
            if table is not None:
                call('srx_bn_act_bwd_finish', _p(dz_in), _p(y), _p(mean), _p(invstd), _p(gw), _p(gb), _p(table), rows, 2, _p(sums),
                     _p(dy), m, c, act, 0.0, _p(pw), _p(bn.weight.grad), _p(bn.bias.grad), pg, s)
            else:
                nws = L.srx_bn_bwd_ws_floats(m, c)
                call('srx_bn_act_bwd', _p(dz_in), _p(y), _p(mean), _p(invstd), _p(gw), _p(gb), _p(sums), _p(dy), m, c, 1, act, 0.0,
                     _p(pw), 1, _p(bn.weight.grad), _p(bn.bias.grad), pg, _p(_ws(nws, y)), nws, s)
            return dy

        def dgrad(pack, dy, addend, below):
            """conv^T(dy) [+ addend]; `below` = (y, mean, invstd, bn, prelu) of the BatchNorm the result arrives at, or None"""
            dx = torch.empty_like(dy)
            if below is not None and rows:
                y, mean, invstd, bn, prelu = below
                table = torch.empty((rows, W), dtype=torch.float32, device=dy.device)
                call('srx_conv2d_bwd_data_bn', dref, _p(dy), _p(pack), _p(addend), _p(dx), _p(y), _p(mean), _p(invstd),
                     _p(bn.weight.detach()), _p(bn.bias.detach()), None if prelu is None else _p(prelu.detach()), _p(table), s)
                return dx, table
            ws = _ws(nws_d, dy) if nws_d else None
            if addend is None:
                call('srx_conv2d_bwd_data', dref, _p(dy), _p(pack), _p(dx), 0, _p(ws), nws_d, s)
            else:
                call('srx_conv2d_bwd_data_add', dref, _p(dy), _p(pack), _p(addend), _p(dx), _p(ws), nws_d, s)
            return dx, None

        def wgrad(conv, inp, dy):
            if queue is not None:
                queue.add(d, _p(inp), _p(dy), _p(conv.weight.grad), None, (inp, dy))
                return
            nws = L.srx_conv2d_bwd_weight_ws_floats(dref)
            call('srx_conv2d_bwd_weight', dref, _p(inp), _p(dy), _p(conv.weight.grad), 1, None, _p(_ws(nws, inp)), nws, s)

        # With a table from the producer, the APPLY pass of a BatchNorm backward rides in the data gradient that consumes its
        # result too (srx_conv2d_bwd_data_bn_in: the layer's input gradient is formed while the patch is staged and written on
        # the side for the weight gradient): per block two more launches disappear, finalize -> data gradient is all that is left
        fuse_in = bool(rows) and L.srx_conv2d_bwd_data_bn_in_ok(dref) == 1

        def sums_only(y, bn, act, prelu, table):
            """finalize pass alone: the reduced sums (and the parameter gradients) from the producer's table"""
            sums = torch.empty(W, dtype=torch.float32, device=grad.device)
            pw = None if prelu is None else prelu.detach()
            pg = None if prelu is None else _p(prelu.grad)
            call('srx_bn_act_bwd_finish', None, None, None, None, None, None, _p(table), rows, 2, _p(sums), None, m, c, act, 0.0,
                 _p(pw), _p(bn.weight.grad), _p(bn.bias.grad), pg, s)
            return sums

        def dgrad_in(pack, dz_in, above, sums, addend, below):
            """conv^T(dy) [+ addend] with dy = the input gradient of the BatchNorm (+ PReLU) layer `above` = (y, mean, invstd,
            bn, prelu), formed from dz_in on load and returned as well; `below` as in dgrad"""
            y, mean, invstd, bn, prelu = above
            dy, dx = torch.empty_like(y), torch.empty_like(y)
            table = None
            by = bmean = binv = bg = bb = bp = None
            if below is not None:
                table = torch.empty((rows, W), dtype=torch.float32, device=y.device)
                by, bmean, binv, bbn, bprelu = below
                bg, bb = bbn.weight.detach(), bbn.bias.detach()
                bp = None if bprelu is None else bprelu.detach()
            call('srx_conv2d_bwd_data_bn_in', dref, _p(dz_in), _p(y), _p(mean), _p(invstd), _p(bn.weight.detach()),
                 _p(bn.bias.detach()), None if prelu is None else _p(prelu.detach()), _p(sums), _p(dy), _p(pack), _p(addend), _p(dx),
                 _p(by), _p(bmean), _p(binv), _p(bg), _p(bb), _p(bp), _p(table), s)
            return dy, dx, table

        table2 = None  # the reduce pass of the top block's bn2 has no producer inside this node
        for i in range(len(blocks) - 1, -1, -1):
            block = blocks[i]
            x, y1, a1, y2, mean1, inv1, mean2, inv2 = saved[8 * i:8 * i + 8]
            below1 = (y1, mean1, inv1, block.bn1, block.prelu.weight)
            if fuse_in and table2 is not None:
                sums2 = sums_only(y2, block.bn2, ACT_NONE, None, table2)
                dy2, da1, table1 = dgrad_in(ctx.packs[i][1], grad, (y2, mean2, inv2, block.bn2, None), sums2, None, below1)
            else:
                dy2 = finish(grad, y2, mean2, inv2, block.bn2, ACT_NONE, None, table2)
                da1, table1 = dgrad(ctx.packs[i][1], dy2, None, below1)
            wgrad(block.conv2, a1, dy2)
            need_dx = i > 0 or ctx.needs_input_grad[0]
            below = None
            if i > 0:
                pb = blocks[i - 1]
                below = (saved[8 * (i - 1) + 3], saved[8 * (i - 1) + 6], saved[8 * (i - 1) + 7], pb.bn2, None)
            if fuse_in and table1 is not None and need_dx:
                sums1 = sums_only(y1, block.bn1, ACT_PRELU, block.prelu.weight, table1)
                dy1, grad, table2 = dgrad_in(ctx.packs[i][0], da1, below1, sums1, grad, below)
                wgrad(block.conv1, x, dy1)
                continue
            dy1 = finish(da1, y1, mean1, inv1, block.bn1, ACT_PRELU, block.prelu.weight, table1)
            wgrad(block.conv1, x, dy1)
            if need_dx:
                grad, table2 = dgrad(ctx.packs[i][0], dy1, grad, below)
            else:
                grad = None
        return (grad, None) + (None,) * (len(ctx.needs_input_grad) - 2)


def residual_tower(x: Tensor, blocks) -> Tensor:
    """``nn.Sequential(*blocks)(x)`` for ``ResidualBlock`` modules that all satisfy ``residual_block_fused_ok``."""
    ps = []
    for b in blocks:
        ps += [b.conv1.weight, b.bn1.weight, b.bn1.bias, b.prelu.weight, b.conv2.weight, b.bn2.weight, b.bn2.bias]
    return _ResidualTower.apply(x, list(blocks), *ps)


def residual_block_fused_ok(block) -> bool:
    """The one-node form applies in training mode with autograd on, when every parameter of the block accumulates
    its gradient straight into a flat ``.grad`` buffer (``direct_grads``: the trainers)."""
    if not (block.training and torch.is_grad_enabled() and direct_grads[0]):
        return False
    ps = (block.conv1.weight, block.bn1.weight, block.bn1.bias, block.prelu.weight, block.conv2.weight, block.bn2.weight,
          block.bn2.bias)
    return block.bn1.training and block.bn2.training and block.prelu.weight.numel() == 1 and \
        all(p.requires_grad and _sink(p) is not None for p in ps)


def residual_block(x: Tensor, block) -> Tensor:
    ps = (block.conv1.weight, block.bn1.weight, block.bn1.bias, block.prelu.weight, block.conv2.weight, block.bn2.weight,
          block.bn2.bias)
    return _ResidualBlock.apply(x, block, *ps)


# --------------------------------------------------------------------------- activations
class _PReLU(Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'prelu.input')
        wd = _chk(w.detach(), 'prelu.weight')
        y = torch.empty_like(x)
        call('srx_prelu_fwd', _p(x), _p(wd), _p(y), x.numel(), _stream())
        ctx.save_for_backward(x, wd)
        ctx.param = w
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w = ctx.saved_tensors
        dy = _chk(dy, 'prelu.grad')
        dx = torch.empty_like(x)
        sink = _sink(ctx.param) if ctx.needs_input_grad[1] else None
        dw = None if sink is not None else torch.empty(1, dtype=torch.float32, device=x.device)
        call('srx_prelu_bwd', _p(dy), _p(x), _p(w), _p(dx), _p(dw if sink is None else sink), 0 if sink is None else 1,
             x.numel(), _p(_ws(1024, x)), _stream())
        return dx, dw


def prelu(x: Tensor, w: Tensor) -> Tensor:
    if w.numel() != 1:
        raise RuntimeError('prelu: only the single-slope nn.PReLU() of the reference is implemented')
    return _PReLU.apply(x, w)


class _LReLU(Function):
    @staticmethod
    def forward(ctx, x: Tensor, slope: float):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'lrelu.input')
        y = torch.empty_like(x)
        call('srx_lrelu_fwd', _p(x), _p(y), x.numel(), slope, _stream())
        ctx.slope = slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (y,) = ctx.saved_tensors
        dy = _chk(dy, 'lrelu.grad')
        dx = torch.empty_like(dy)
        call('srx_act_bwd_from_out', _p(dy), _p(y), _p(dx), dy.numel(), ACT_LRELU, ctx.slope, _stream())
        return dx, None


def leaky_relu(x: Tensor, slope: float) -> Tensor:
    return _LReLU.apply(x, float(slope))


class _Sigmoid(Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'sigmoid.input')
        y = torch.empty_like(x)
        call('srx_sigmoid_fwd', _p(x), _p(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (y,) = ctx.saved_tensors
        dy = _chk(dy, 'sigmoid.grad')
        dx = torch.empty_like(y)
        call('srx_sigmoid_bwd', _p(dy), _p(y), _p(dx), y.numel(), _stream())
        return dx


def sigmoid(x: Tensor) -> Tensor:
    return _Sigmoid.apply(x)


class _Axpby(Function):
    @staticmethod
    def forward(ctx, x: Tensor, z: Tensor, a: float, b: float):
        ctx.set_materialize_grads(False)
        x, z = _chk(x, 'axpby.x'), _chk(z, 'axpby.z')
        y = torch.empty_like(x)
        call('srx_axpby', _p(x), _p(z), _p(y), x.numel(), a, b, _stream())
        ctx.ab = (a, b)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        a, b = ctx.ab
        dy = _chk(dy, 'axpby.grad')
        s = _stream()
        dx = dz = None
        if ctx.needs_input_grad[0]:
            dx = dy if a == 1.0 else torch.empty_like(dy)
            if a != 1.0:
                call('srx_axpby', _p(dy), _p(dy), _p(dx), dy.numel(), a, 0.0, s)
        if ctx.needs_input_grad[1]:
            dz = dy if b == 1.0 else torch.empty_like(dy)
            if b != 1.0:
                call('srx_axpby', _p(dy), _p(dy), _p(dz), dy.numel(), b, 0.0, s)
        return dx, dz, None, None


def axpby(x: Tensor, z: Tensor, a: float = 1.0, b: float = 1.0) -> Tensor:
    """``a*x + b*z`` (residual adds / scalings)."""
    return _Axpby.apply(x, z, float(a), float(b))


# --------------------------------------------------------------------------- pooling
class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'maxpool.input')
        n, h, w, c = x.shape
        y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
        call('srx_maxpool2x2_fwd', _p(x), _p(y), n, h, w, c, _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (x,) = ctx.saved_tensors
        dy = _chk(dy, 'maxpool.grad')
        n, h, w, c = x.shape
        dx = torch.empty_like(x)
        call('srx_maxpool2x2_bwd', _p(dy), _p(x), _p(dx), n, h, w, c, _stream())
        return dx


def maxpool2x2(x: Tensor) -> Tensor:
    return _MaxPool.apply(x)


class _FrozenConvStack(Function):
    """A frozen stack of ``conv3x3 + ReLU`` and ``MaxPool2d(2)`` layers as ONE autograd node: the VGG19 feature
    extractor of the perceptual loss (srgan/loss.py:30-34,52-53: pretrained, ``requires_grad = False``, eval).

    Forward: ``source`` and ``target`` (loss.py:52 and :53) go through the stack TOGETHER as one batch of 2N -- same
    weights, twice the rows per launch (the 6x6 and 12x12 layers alone fill a quarter of the chip).  Backward: only
    the source half is differentiated (the target is ``high_res.detach()``, srgan/trainer.py:455), weights take no
    gradient, and every ReLU backward rides in the epilogue of the data gradient above it
    (``srx_conv2d_bwd_data_act``) or in the max-pool backward (``srx_maxpool2x2_relu_bwd``): 16 data-gradient
    launches and 4 pool launches, no elementwise passes besides the topmost ReLU.
    """

    @staticmethod
    def forward(ctx, source: Tensor, target: Optional[Tensor], layers, *weights):
        ctx.set_materialize_grads(False)
        source = _chk(source, 'conv_stack.source')
        n_src = source.shape[0]
        x = source if target is None else torch.cat([source, _chk(target, 'conv_stack.target')], dim=0)
        L, s = _lib.lib(), _stream()
        ctx.bf16s = _bf16_stack_ok(layers, x.shape)
        if ctx.bf16s:
            return _FrozenConvStack._forward_bf16s(ctx, x, n_src, target is None, layers)
        saved, plan = [x], []
        for kind, conv in layers:
            n, h, w, c = x.shape
            if kind == 'pool':
                y = torch.empty((n, h // 2, w // 2, c), dtype=torch.float32, device=x.device)
                call('srx_maxpool2x2_fwd', _p(x), _p(y), n, h, w, c, s)
                plan.append(('pool', None, None))
            else:
                st = conv._st
                if st.act not in (ACT_RELU, ACT_LRELU) or st.stride != 1 or st.shuffle:
                    raise RuntimeError('conv_stack: stride-1 conv + ReLU / LeakyReLU layers only')
                d = st.desc(n, h, w)
                dref = C.byref(d)
                y = torch.empty(st.out_shape(n, h, w), dtype=torch.float32, device=x.device)
                b = None if conv.bias is None else _chk(conv.bias.detach(), 'conv_stack.bias')
                # wide 3x3 layers: Winograd F(2x2, 3x3), 2.25x fewer fp32 multiplications (csrc/wino.hip)
                wino = wino_layer_ok(st, d)
                if wino:
                    st.pack_wino(conv.weight, d, need_bwd=ctx.needs_input_grad[0])
                    nws = L.srx_wino_ws_floats(dref, 0)
                    ws = _ws(nws, x) if nws else None
                    call('srx_wino_fwd', dref, _p(x), _p(st.wino_fwd), _p(b), _p(y), _p(ws), nws, s)
                    plan.append(('conv', st, ('wino', st.wino_bwd)))
                else:
                    st.pack(conv.weight, d)
                    nws = L.srx_conv2d_fwd_ws_floats(dref)
                    ws = _ws(nws, x) if nws else None
                    call('srx_conv2d_fwd', dref, _p(x), _p(st.wpk_fwd), _p(b), _p(y), None, _p(ws), nws, s)
                    plan.append(('conv', st, st.wpk_bwd))
            saved.append(y)
            x = y
        ctx.plan, ctx.n_src = plan, n_src
        ctx.masters = [conv.weight if kind == 'conv' else None for kind, conv in layers]
        ctx.save_for_backward(*saved)
        if target is None:
            return x, None
        fs, ft = x[:n_src], x[n_src:]
        ctx.mark_non_differentiable(ft)
        return fs, ft

    @staticmethod
    def _forward_bf16s(ctx, x: Tensor, n_src: int, single: bool, layers):
        """Under autocast (``precision = 1`` on every layer; esrgan/trainer.py:461-467) the activations BETWEEN the convs are
        stored as bf16 (csrc/gconv.hip "bf16 STORAGE"): the same bf16 products as before -- an operand is rounded once, by
        its producer -- at half the bytes and without the loaders' fp32 -> bf16 conversion.  A conv in front of a pool and
        the last conv write fp32: the pool compares fp32 values (as the recipe does), the features go to the fp32 MSE."""
        L, s = _lib.lib(), _stream()
        saved, plan = [x], []
        need_bwd = ctx.needs_input_grad[0]
        for i, (kind, conv) in enumerate(layers):
            n, h, w, c = x.shape
            to_f32 = i + 1 == len(layers) or layers[i + 1][0] == 'pool'
            if kind == 'pool':
                y = torch.empty((n, h // 2, w // 2, c), dtype=torch.bfloat16, device=x.device)
                call('srx_maxpool2x2_fwd_to_bf16', _p(x), _p(y), n, h, w, c, s)
                plan.append(('pool', None))
            else:
                st = conv._st
                d = st.desc(n, h, w)
                dref = C.byref(d)
                y = torch.empty(st.out_shape(n, h, w), dtype=torch.float32 if to_f32 else torch.bfloat16, device=x.device)
                b = None if conv.bias is None else _chk(conv.bias.detach(), 'conv_stack.bias')
                if i == 0:   # the 3 -> 64 first layer: fp32 image in, its own kernel
                    st.pack(conv.weight, d)
                    call('srx_conv2d_fwd_first3_to_bf16', dref, _p(x), _p(st.wpk_fwd), _p(b), _p(y), s)
                else:
                    st.pack_bf16s(conv.weight, d, need_bwd)
                    nws = L.srx_conv3x3_bf16s_ws_floats(dref, 0)
                    ws = _ws(nws, x) if nws else None
                    call('srx_conv3x3_bf16s_fwd', dref, _p(x), _p(st.bf16s_fwd), _p(b), 1, _p(y), 0 if to_f32 else 1, _p(ws), nws, s)
                plan.append(('conv', st))
            saved.append(y)
            x = y
        ctx.plan, ctx.n_src = plan, n_src
        ctx.save_for_backward(*saved)
        if single:
            return x, None
        fs, ft = x[:n_src], x[n_src:]
        ctx.mark_non_differentiable(ft)
        return fs, ft

    @staticmethod
    def _backward_bf16s(ctx, dfs: Tensor):
        saved, plan, n = ctx.saved_tensors, ctx.plan, ctx.n_src
        L, s = _lib.lib(), _stream()
        g32 = _chk(dfs, 'conv_stack.grad')
        top = saved[-1][:n]
        g = torch.empty(g32.shape, dtype=torch.bfloat16, device=g32.device)
        call('srx_act_bwd_from_out_to_bf16', _p(g32), _p(top), _p(g), g32.numel(), ACT_RELU, 0.0, s)
        for i in range(len(plan) - 1, 0, -1):
            kind, st = plan[i]
            x = saved[i][:n]                      # this layer's input (source half)
            if kind == 'pool':                    # x: the fp32 output of the conv + ReLU below
                _, h, w, c = x.shape
                dx = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
                call('srx_maxpool2x2_relu_bwd_bf16', _p(g), _p(x), _p(dx), n, h, w, c, s)
                g = dx
                continue
            _, h, w, _c = x.shape
            d = st.desc(n, h, w)
            dref = C.byref(d)
            fold = plan[i - 1][0] == 'conv'       # x = relu(conv below): that ReLU's backward on the way out
            to_f32 = i == 1                       # the first layer's data gradient (3-channel kernel) reads fp32
            dx = torch.empty(x.shape, dtype=torch.float32 if to_f32 else torch.bfloat16, device=x.device)
            nws = L.srx_conv3x3_bf16s_ws_floats(dref, 1)
            ws = _ws(nws, dx) if nws else None
            call('srx_conv3x3_bf16s_bwd_data', dref, _p(g), _p(st.bf16s_bwd), _p(x) if fold else None, _p(dx), 0 if to_f32 else 1,
                 _p(ws), nws, s)
            g = dx
        st = plan[0][1]
        x = saved[0][:n]
        _, h, w, _c = x.shape
        d = st.desc(n, h, w)
        dref = C.byref(d)
        dx = torch.empty_like(x)
        nws = L.srx_conv2d_bwd_data_ws_floats(dref)
        ws = _ws(nws, x) if nws else None
        call('srx_conv2d_bwd_data', dref, _p(g), _p(st.wpk_bwd), _p(dx), 0, _p(ws), nws, s)
        return (dx, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)

    @staticmethod
    def backward(ctx, dfs: Tensor, _dft=None):
        saved, plan, n = ctx.saved_tensors, ctx.plan, ctx.n_src
        if dfs is None or not ctx.needs_input_grad[0]:
            return (None,) * len(ctx.needs_input_grad)
        if ctx.bf16s:
            return _FrozenConvStack._backward_bf16s(ctx, dfs)
        L, s = _lib.lib(), _stream()
        g = _chk(dfs, 'conv_stack.grad')
        # the topmost activation's backward is the only elementwise pass
        top = saved[-1][:n]
        kind, st, _ = plan[-1]
        if kind == 'conv':
            g2 = torch.empty_like(g)
            call('srx_act_bwd_from_out', _p(g), _p(top), _p(g2), g.numel(), st.act, st.slope, s)
            g = g2
        for i in range(len(plan) - 1, -1, -1):
            kind, st, wpk_bwd = plan[i]
            x = saved[i][:n]                      # this layer's input (source half: batch-major, so a prefix)
            below = plan[i - 1] if i > 0 else None
            if kind == 'pool':
                _, h, w, c = x.shape
                dx = torch.empty_like(x)
                # the pool's input is always a ReLU output in cfg 'E'; a pool fed by something else keeps the plain form
                fn = 'srx_maxpool2x2_relu_bwd' if (below is not None and below[0] == 'conv' and below[1].act == ACT_RELU) \
                    else 'srx_maxpool2x2_bwd'
                call(fn, _p(g), _p(x), _p(dx), n, h, w, c, s)
                if fn == 'srx_maxpool2x2_bwd' and below is not None and below[0] == 'conv':
                    g2 = torch.empty_like(dx)
                    call('srx_act_bwd_from_out', _p(dx), _p(x), _p(g2), dx.numel(), below[1].act, below[1].slope, s)
                    dx = g2
                g = dx
                continue
            _, h, w, _c = x.shape
            d = st.desc(n, h, w)
            dref = C.byref(d)
            dx = torch.empty_like(x)
            fold = below is not None and below[0] == 'conv'   # x = act(conv below): fold that activation's backward in
            if isinstance(wpk_bwd, tuple) and (not fold or below[1].act == ACT_RELU) and wpk_bwd[1] is not None:
                nws = L.srx_wino_ws_floats(dref, 1)
                ws = _ws(nws, x) if nws else None
                call('srx_wino_bwd_data', dref, _p(g), _p(wpk_bwd[1]), _p(x) if fold else None, _p(dx), _p(ws), nws, s)
                g = dx
                continue
            if isinstance(wpk_bwd, tuple):  # (a Winograd forward whose data gradient takes the direct kernel: pack it now)
                st.pack(ctx.masters[i], d)
                wpk_bwd = st.wpk_bwd
            nws = L.srx_conv2d_bwd_data_ws_floats(dref)
            ws = _ws(nws, x) if nws else None
            if fold:
                slope = 0.0 if below[1].act == ACT_RELU else below[1].slope
                call('srx_conv2d_bwd_data_act', dref, _p(g), _p(wpk_bwd), _p(x), slope, 0, st.cin_s, 0, _p(dx), _p(ws), nws, s)
            else:
                call('srx_conv2d_bwd_data', dref, _p(g), _p(wpk_bwd), _p(dx), 0, _p(ws), nws, s)
            g = dx
        return (g, None, None) + (None,) * (len(ctx.needs_input_grad) - 3)


def _bf16_stack_ok(layers, shape) -> bool:
    """May this frozen stack keep its inner activations as bf16 (``_FrozenConvStack._forward_bf16s``)?  Every conv multiplies
    bf16 products (autocast), starts with the 3 -> 64 layer, every other conv is a 3x3 / stride 1 / pad 1 + ReLU layer with
    channel counts that are multiples of 64, pools sit between convs and the stack ends with a conv."""
    if _dev.NO_BF16S or len(layers) < 2 or layers[0][0] != 'conv' or layers[-1][0] != 'conv' or layers[1][0] != 'conv':
        return False
    L = _lib.lib()
    n, h, w, _c = shape
    for i, (kind, conv) in enumerate(layers):
        if kind == 'pool':
            if layers[i - 1][0] != 'conv' or layers[i + 1][0] != 'conv' or h % 2 or w % 2:
                return False
            h, w = h // 2, w // 2
            continue
        st = conv._st
        if st.precision != 1 or st.act != ACT_RELU or st.stride != 1 or st.shuffle or st.k != 3 or st.pad != 1:
            return False
        if i == 0:
            if st.cin > 4 or st.cout != 64:
                return False
        elif L.srx_conv3x3_bf16s_applicable(C.byref(st.desc(n, h, w))) != 1:
            return False
    return True


def frozen_conv_stack(source: Tensor, target: Optional[Tensor], layers):
    """``layers``: [('conv', layers.Conv2d) | ('pool', None)].  Returns ``(features(source), features(target))``."""
    weights = [p for _, m in layers if m is not None for p in (m.weight, m.bias) if p is not None]
    if any(p.requires_grad for p in weights):
        raise RuntimeError('frozen_conv_stack: the stack must be frozen (requires_grad = False on every parameter)')
    return _FrozenConvStack.apply(source, target, layers, *weights)


# --------------------------------------------------------------------------- linear
class _Linear(Function):
    @staticmethod
    def forward(ctx, x: Tensor, w: Tensor, bias: Optional[Tensor], act: int, slope: float):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'linear.input')
        wd = _chk(w.detach(), 'linear.weight')
        bsz, k = x.shape
        j = wd.shape[0]
        y = torch.empty((bsz, j), dtype=torch.float32, device=x.device)
        nws = _lib.lib().srx_linear_ws_floats(bsz, k, j)
        b = None if bias is None else _chk(bias.detach(), 'linear.bias')
        call('srx_linear_fwd', _p(x), _p(wd), _p(b), _p(y), bsz, k, j, act, slope, _p(_ws(nws, x)), nws, _stream())
        ctx.cfg = (bsz, k, j, act, slope, bias is not None, nws)
        ctx.params = (w, bias)
        ctx.save_for_backward(x, wd, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, w, y = ctx.saved_tensors
        bsz, k, j, act, slope, has_bias, nws = ctx.cfg
        dy = _chk(dy, 'linear.grad')
        s = _stream()
        if act != ACT_NONE:
            g = torch.empty_like(dy)
            call('srx_act_bwd_from_out', _p(dy), _p(y), _p(g), dy.numel(), act, slope, s)
            dy = g
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            call('srx_linear_bwd_data', _p(dy), _p(w), _p(dx), bsz, k, j, _p(_ws(nws, x)), nws, s)
        wparam, bparam = ctx.params
        if ctx.needs_input_grad[1]:
            sink = _sink(wparam)
            dw = None if sink is not None else torch.empty_like(w)
            call('srx_linear_bwd_weight', _p(x), _p(dy), _p(dw if sink is None else sink), 0 if sink is None else 1,
                 bsz, k, j, s)
        if has_bias and ctx.needs_input_grad[2]:
            sink = _sink(bparam)
            db = None if sink is not None else torch.empty(j, dtype=torch.float32, device=x.device)
            n2 = _lib.lib().srx_colsum_ws_floats(bsz, j)
            call('srx_colsum', _p(dy), _p(db if sink is None else sink), bsz, j, j, 0 if sink is None else 1,
                 _p(_ws(n2, x)), n2, s)
        return dx, dw, db, None, None


def linear(x: Tensor, w: Tensor, bias: Optional[Tensor], act: int = ACT_NONE, slope: float = 0.0) -> Tensor:
    return _Linear.apply(x, w, bias, act, float(slope))


class _GanHead(Function):
    """Discriminator head + adversarial loss of a GAN train step as ONE autograd node (csrc/head.hip):
    ``hidden = LeakyReLU(Linear1(x))`` on the streaming linear kernels, then last Linear -> [Sigmoid] -> loss in one
    launch, and in the backward pass one launch for everything between the loss and the hidden layer's pre-activation
    (loss, sigmoid, last Linear, LeakyReLU backwards; both bias gradients and the last weight gradient) followed by the
    hidden layer's data and weight gradients.  Returns ``(loss, aux)``; ``aux`` (8 floats, no gradient) holds the terms.
    """

    @staticmethod
    def forward(ctx, x: Tensor, w1: Tensor, b1: Optional[Tensor], w2: Tensor, b2: Optional[Tensor], mode: int, n_first: int,
                slope: float, adv_weight: float, shift: Optional[Tensor], addend: Optional[Tensor]):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'gan_head.input')
        w1d, w2d = _chk(w1.detach(), 'gan_head.weight1'), _chk(w2.detach(), 'gan_head.weight2')
        bsz, k = x.shape
        j = w1d.shape[0]
        if w2d.numel() != j:
            raise RuntimeError(f'gan_head: the last layer must map {j} hidden units to ONE output, got {tuple(w2d.shape)}')
        s = _stream()
        hidden = torch.empty((bsz, j), dtype=torch.float32, device=x.device)
        nws = _lib.lib().srx_linear_ws_floats(bsz, k, j)
        b1d = None if b1 is None else _chk(b1.detach(), 'gan_head.bias1')
        b2d = None if b2 is None else _chk(b2.detach(), 'gan_head.bias2')
        call('srx_linear_fwd', _p(x), _p(w1d), _p(b1d), _p(hidden), bsz, k, j, ACT_LRELU, slope, _p(_ws(nws, x)), nws, s)
        h = _lib.GanHead(mode, bsz, j, n_first, slope, adv_weight)
        zp = torch.empty(bsz, dtype=torch.float32, device=x.device)
        out = torch.empty(8, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        sh = None if shift is None else _chk(shift.detach().reshape(1), 'gan_head.shift')
        ad = None if addend is None else _chk(addend.detach().reshape(1), 'gan_head.addend')
        call('srx_gan_head_fwd', C.byref(h), _p(hidden), _p(w2d), _p(b2d), _p(sh), _p(ad), _p(zp), _p(out), _p(loss), s)
        ctx.h, ctx.nws = h, nws
        ctx.params = (w1, b1, w2, b2)
        ctx.save_for_backward(x, w1d, w2d, hidden, zp, out, sh)
        ctx.mark_non_differentiable(out)
        return loss, out

    @staticmethod
    def backward(ctx, g: Tensor, _gaux=None):
        x, w1, w2, hidden, zp, out, sh = ctx.saved_tensors
        h = ctx.h
        bsz, k = x.shape
        j = h.J
        s = _stream()
        g = _chk(g, 'gan_head.grad').reshape(1)
        w1p, b1p, w2p, b2p = ctx.params
        need = ctx.needs_input_grad  # (x, w1, b1, w2, b2, ..., shift, addend)
        dw1 = db1 = dw2 = db2 = None
        sinks = [(_sink(p) if (p is not None and need[i + 1]) else None) for i, p in enumerate(ctx.params)]
        wanted = [p is not None and need[i + 1] for i, p in enumerate(ctx.params)]
        # the small gradients either ALL accumulate into flat .grad buffers (the trainers) or are all returned
        direct = all(sk is not None for sk, wnt in zip(sinks, wanted) if wnt) and any(wanted)
        ptrs = [None, None, None, None]
        if not direct:
            if wanted[1]:
                db1 = torch.empty(j, dtype=torch.float32, device=x.device)
            if wanted[2]:
                dw2 = torch.empty_like(w2)
            if wanted[3]:
                db2 = torch.empty(1, dtype=torch.float32, device=x.device)
            ptrs = [None, _p(db1), _p(dw2), _p(db2)]
        else:
            ptrs = [None] + [_p(sk) if wnt else None for sk, wnt in list(zip(sinks, wanted))[1:]]
        dpre = torch.empty((bsz, j), dtype=torch.float32, device=x.device)
        call('srx_gan_head_bwd', C.byref(h), _p(hidden), _p(w2), _p(zp), _p(out), _p(sh), _p(g), _p(dpre), ptrs[2], ptrs[3],
             ptrs[1], 1 if direct else 0, s)
        dx = None
        if need[0]:
            dx = torch.empty_like(x)
            call('srx_linear_bwd_data', _p(dpre), _p(w1), _p(dx), bsz, k, j, _p(_ws(ctx.nws, x)), ctx.nws, s)
        if wanted[0]:
            sink = sinks[0] if direct else None
            if sink is None:
                dw1 = torch.empty_like(w1)
            call('srx_linear_bwd_weight', _p(x), _p(dpre), _p(dw1 if sink is None else sink), 0 if sink is None else 1, bsz, k, j, s)
        dadd = g.view(()) if need[10] else None
        return dx, dw1, db1, (None if dw2 is None else dw2.view_as(w2p)), db2, None, None, None, None, None, dadd


def gan_head(x: Tensor, lin1, lin2, mode: int, n_first: int = 0, slope: float = 0.2, adv_weight: float = 1.0,
             shift: Optional[Tensor] = None, addend: Optional[Tensor] = None):
    """``(loss, aux)`` of a discriminator's classifier (``lin1`` -> LeakyReLU -> ``lin2`` [-> Sigmoid]) and the adversarial
    loss behind it on the flattened features ``x`` [B, K]; modes and ``aux`` layout: ``srx_gan_head_t`` in include/srx.h.
    ``lin1`` / ``lin2`` are ``layers.Linear`` modules (``no_weight_grad`` is honoured)."""
    from .layers import _w
    return _GanHead.apply(x, _w(lin1.weight), _w(lin1.bias), _w(lin2.weight), _w(lin2.bias), int(mode), int(n_first), float(slope),
                          float(adv_weight), shift, addend)


# --------------------------------------------------------------------------- losses
class _PairLoss(Function):
    """mean((a-b)^2) / mean(|a-b|); ``count``: the mean's divisor when not every stored element is a real one"""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor, kind: str, count: int = 0):
        ctx.set_materialize_grads(False)
        a, b = _chk(a, f'{kind}.input'), _chk(b, f'{kind}.target')
        if a.shape != b.shape:
            raise RuntimeError(f'{kind}_loss: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}')
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        if count:
            call(f'srx_{kind}_fwd_count', _p(a), _p(b), _p(loss), a.numel(), count, _p(_ws(2048, a)), _stream())
        else:
            call(f'srx_{kind}_fwd', _p(a), _p(b), _p(loss), a.numel(), _p(_ws(2048, a)), _stream())
        ctx.kind, ctx.count = kind, count
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        a, b = ctx.saved_tensors
        g = _chk(g, 'loss.grad')
        da = torch.empty_like(a)
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        if ctx.count:
            call(f'srx_{ctx.kind}_bwd_count', _p(a), _p(b), _p(g), _p(da), _p(db), a.numel(), ctx.count, _stream())
        else:
            call(f'srx_{ctx.kind}_bwd', _p(a), _p(b), _p(g), _p(da), _p(db), a.numel(), _stream())
        return (da if ctx.needs_input_grad[0] else None), db, None, None


def mse_loss(a: Tensor, b: Tensor) -> Tensor:
    """nn.MSELoss() (srgan/trainer.py:163)."""
    return _PairLoss.apply(a, b, 'mse')


def l1_loss(a: Tensor, b: Tensor, count: int = 0) -> Tensor:
    """F.l1_loss / nn.L1Loss() (srgan/loss.py:52).  ``count``: divisor of the mean when the tensors carry padding (NHWC images
    with a zero 4th channel: ``count = 3 * pixels``)."""
    return _PairLoss.apply(a, b, 'l1', int(count))


class _BCE(Function):
    @staticmethod
    def forward(ctx, p: Tensor, target: float):
        ctx.set_materialize_grads(False)
        p = _chk(p, 'bce.input')
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        call('srx_bce_fwd', _p(p), target, _p(loss), p.numel(), _p(_ws(2048, p)), _stream())
        ctx.target = target
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        (p,) = ctx.saved_tensors
        g = _chk(g, 'bce.grad')
        dp = torch.empty_like(p)
        call('srx_bce_bwd', _p(p), ctx.target, _p(g), _p(dp), p.numel(), _stream())
        return dp, None


def bce_loss(p: Tensor, target: float) -> Tensor:
    """nn.BCELoss() against a constant label tensor (srgan/trainer.py:439-447)."""
    return _BCE.apply(p, float(target))


class _BCELogits(Function):
    @staticmethod
    def forward(ctx, x: Tensor, shift: Optional[Tensor], target: float):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'bce_logits.input')
        sh = None if shift is None else _chk(shift.detach().reshape(1), 'bce_logits.shift')
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        call('srx_bce_logits_fwd', _p(x), _p(sh), target, _p(loss), x.numel(), _p(_ws(2048, x)), _stream())
        ctx.target = target
        ctx.save_for_backward(x, sh)
        return loss

    @staticmethod
    def backward(ctx, g: Tensor):
        x, sh = ctx.saved_tensors
        g = _chk(g, 'bce_logits.grad')
        dx = torch.empty_like(x)
        call('srx_bce_logits_bwd', _p(x), _p(sh), ctx.target, _p(g), _p(dx), x.numel(), _stream())
        dsh = None
        if sh is not None and ctx.needs_input_grad[1]:
            dsh = -dx.sum()  # one scalar; plumbing
        return dx, dsh, None


def bce_with_logits(x: Tensor, target: float, shift: Optional[Tensor] = None) -> Tensor:
    """nn.BCEWithLogitsLoss()(x - shift, full(target)) (esrgan/trainer.py:451-453)."""
    return _BCELogits.apply(x, shift, float(target))


class _Mean(Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'mean.input')
        out = torch.empty((), dtype=torch.float32, device=x.device)
        call('srx_mean_fwd', _p(x), _p(out), x.numel(), _p(_ws(2048, x)), _stream())
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        g = _chk(g, 'mean.grad')
        dx = torch.empty_like(x)
        call('srx_mean_bwd', _p(x), _p(g), _p(dx), x.numel(), _stream())
        return dx


class _SplitBatch(Function):
    """``(x[:n], x[n:])`` of a batch-major tensor as two independent tensors; the backward is one concatenation."""

    @staticmethod
    def forward(ctx, x: Tensor, n: int):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.n = x.shape, n
        return x[:n].clone(), x[n:].clone()

    @staticmethod
    def backward(ctx, da, db):
        n, shape = ctx.n, ctx.shape
        ref = da if da is not None else db
        if ref is None:
            return None, None
        if da is None:
            da = torch.zeros((n,) + tuple(shape[1:]), dtype=ref.dtype, device=ref.device)
        if db is None:
            db = torch.zeros((shape[0] - n,) + tuple(shape[1:]), dtype=ref.dtype, device=ref.device)
        return torch.cat([da, db], dim=0), None


def split_batch(x: Tensor, n: int):
    return _SplitBatch.apply(x, n)


def mean(x: Tensor) -> Tensor:
    """torch.mean(x) over all elements (esrgan/trainer.py:451-452)."""
    return _Mean.apply(x)


# --------------------------------------------------------------------------- ESRGAN helpers
class _ConcatChannels(Function):
    """torch.cat(tensors, dim=1) of the reference's NCHW tensors == channel concat in NHWC."""

    @staticmethod
    def forward(ctx, *xs: Tensor):
        ctx.set_materialize_grads(False)
        xs = [_chk(x, 'concat.input') for x in xs]
        lead = xs[0].shape[:-1]
        cs = [x.shape[-1] for x in xs]
        m = xs[0].numel() // cs[0]
        out = torch.empty(tuple(lead) + (sum(cs),), dtype=torch.float32, device=xs[0].device)
        off, s = 0, _stream()
        for x, c in zip(xs, cs):
            call('srx_copy_channels', _p(x), c, 0, _p(out), sum(cs), off, c, m, 0, s)
            off += c
        ctx.cs, ctx.m = cs, m
        return out

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = _chk(dy, 'concat.grad')
        total, s = sum(ctx.cs), _stream()
        grads, off = [], 0
        for i, c in enumerate(ctx.cs):
            if ctx.needs_input_grad[i]:
                g = torch.empty(dy.shape[:-1] + (c,), dtype=torch.float32, device=dy.device)
                call('srx_copy_channels', _p(dy), total, off, _p(g), c, 0, c, ctx.m, 0, s)
                grads.append(g)
            else:
                grads.append(None)
            off += c
        return tuple(grads)


class FoldedConv:
    """Inference form of ``conv [-> BatchNorm2d (running statistics)] [-> PReLU] [+ residual]``: ONE kernel.

    The eval-mode BatchNorm is an affine map per output channel, so it folds into the conv's weights and
    bias (``w' = w * g / sqrt(var + eps)``, ``b' = (b - mean) * g / sqrt(var + eps) + beta``); a
    single-parameter PReLU is the conv epilogue's LeakyReLU with the learnt slope (it commutes with the
    fused PixelShuffle); the skip connection is the epilogue's addend (``srx_conv2d_fwd_residual``).
    Replaces an elementwise pass over the activation per BatchNorm / PReLU of the generator at inference
    (SURVEY.md section 8f row 1).  Folded tensors are cached and rebuilt when any source tensor changes.
    """

    def __init__(self, conv, bn=None, prelu=None):
        self.conv, self.bn, self.prelu = conv, bn, prelu
        self._key = None
        self.st = None

    def _sources(self):
        ts = [self.conv.weight, self.conv.bias]
        if self.bn is not None:
            ts += [self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var]
        if self.prelu is not None:
            ts.append(self.prelu.weight)
        return [t for t in ts if t is not None]

    def _refresh(self) -> None:
        key = tuple((t.data_ptr(), t._version) for t in self._sources()) + (_pack_epoch[0], self.conv._st.model_epoch[0],
                                                                            self.conv._st.precision)
        if key == self._key:
            return
        conv, bn = self.conv, self.bn
        with torch.no_grad():
            w = conv.weight.detach()
            b = conv.bias.detach() if conv.bias is not None else None
            if bn is not None:
                scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
                w = w * scale.view(-1, 1, 1, 1)
                b = ((b if b is not None else 0.0) - bn.running_mean) * scale + bn.bias.detach()
            self.w = w.contiguous()
            self.b = None if b is None else b.contiguous()
            src = conv._st
            act, slope = src.act, src.slope
            if self.prelu is not None:
                if self.prelu.weight.numel() != 1 or src.act != ACT_NONE:
                    raise RuntimeError('FoldedConv: PReLU must have one parameter and follow a linear conv')
                act, slope = ACT_LRELU, float(self.prelu.weight.detach().reshape(()).item())
            self.st = ConvState(src.cin, src.cout, src.k, src.stride, src.pad, shuffle=src.shuffle, act=act, slope=slope,
                                up=src.up)
            self.st.precision = src.precision
        self._key = key

    def bf16_native_ok(self) -> bool:
        """3x3 / stride 1 / pad 1, 64 input channels, a multiple of 64 outputs: the layers ``srx_conv3x3_c64_bf16_fwd`` runs."""
        st = self.conv._st
        return (st.k == 3 and st.stride == 1 and st.pad == 1 and st.cin == 64 and st.cout % 64 == 0 and st.up == 0
                and (st.shuffle == 0 or (st.shuffle == 2 and st.cout == 256)))

    def _call_bf16(self, x: Tensor, residual: Optional[Tensor]) -> Tensor:
        """bf16 in, bf16 out: the bf16-native chain (activations stored as bf16, weights resident in registers)."""
        st = self.st
        if not self.bf16_native_ok():
            raise RuntimeError('folded_conv: a bf16 input needs a 3x3 / 64-input-channel layer')
        x = _chk16(x, 'folded_conv.input')
        n, h, w, cs = x.shape
        if cs != 64:
            raise RuntimeError(f'folded_conv: bf16 input has {cs} channels, the layer expects 64')
        if self.__dict__.get('_key16') != self._key:
            nbytes = _lib.lib().srx_conv3x3_c64_bf16_packed_bytes(st.cout)
            wpk = self.__dict__.get('_wpk16')
            if wpk is None or wpk.numel() != nbytes or wpk.device != x.device:
                wpk = self._wpk16 = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            call('srx_conv3x3_c64_bf16_pack', _p(self.w), _p(self.b), None, st.cout, st.shuffle, _p(wpk), _stream())
            self._key16 = self._key
        y = torch.empty(st.out_shape(n, h, w), dtype=torch.bfloat16, device=x.device)
        slope = 1.0 if st.act == ACT_NONE else (0.0 if st.act == ACT_RELU else st.slope)
        r = None
        if residual is not None:
            r = _chk16(residual, 'folded_conv.residual')
            if r.shape != y.shape:
                raise RuntimeError('folded_conv: residual must have the output shape')
        call('srx_conv3x3_c64_bf16_fwd', n, h, w, st.cout, st.shuffle, _p(x), _p(self._wpk16), float(slope), _p(r), _p(y),
             y.shape[3], _stream())
        return y

    def __call__(self, x: Tensor, residual: Optional[Tensor] = None) -> Tensor:
        self._refresh()
        if x.dtype == torch.bfloat16:
            return self._call_bf16(x, residual)
        st = self.st
        x = _chk(x, 'folded_conv.input')
        n, h, w, cs = x.shape
        if cs != st.cin_s:
            raise RuntimeError(f'folded_conv: input has {cs} channels (stride), layer expects {st.cin_s}')
        d = st.desc(n, h, w)
        dref = C.byref(d)
        L = _lib.lib()
        y = torch.empty(st.out_shape(n, h, w), dtype=torch.float32, device=x.device)
        r = None
        if residual is not None:
            r = _chk(residual, 'folded_conv.residual')
            if r.shape != y.shape:
                raise RuntimeError('folded_conv: residual must have the output shape')
        # exact-fp32 inference: the 3x3 layers of the trunk as Winograd F(2x2, 3x3) (csrc/wino.hip; 4/9 of the multiplications,
        # PReLU / skip in its epilogue) -- at 1080p each 64 -> 64 layer is 16 200 tile blocks
        if (st.precision == 0 and not _dev.NO_WINO and st.act in (ACT_NONE, ACT_RELU, ACT_LRELU)
                and L.srx_wino_infer_applicable(dref) == 1):
            st.pack_wino(self.w, d, need_bwd=False)
            nws = L.srx_wino_ws_floats(dref, 0)
            ws = _ws(nws, x) if nws else None
            call('srx_wino_fwd_act', dref, _p(x), _p(st.wino_fwd), _p(self.b), _p(r), _p(y), _p(ws), nws, _stream())
            return y
        st.pack(self.w, d)
        nws = L.srx_conv2d_fwd_ws_floats(dref)
        ws = _ws(nws, x) if nws else None
        if residual is None:
            call('srx_conv2d_fwd', dref, _p(x), _p(st.wpk_fwd), _p(self.b), _p(y), None, _p(ws), nws, _stream())
        else:
            call('srx_conv2d_fwd_residual', dref, _p(x), _p(st.wpk_fwd), _p(self.b), _p(r), 1.0, _p(y), _p(ws), nws, _stream())
        return y


def inference_mode(module) -> bool:
    """The folded single-kernel forms apply when nothing will be differentiated and BatchNorm is in eval mode."""
    return not module.training and not torch.is_grad_enabled()


def concat_channels(xs) -> Tensor:
    return _ConcatChannels.apply(*xs)


class _DenseBlock(Function):
    """ESRGAN's ResidualDenseBlock as ONE autograd node (esrgan/residual.py:65-86).

    ``c_k = LeakyReLU(conv_k(cat(x, c_1..c_{k-1})))`` for k = 1..4, ``y = conv_5(cat(x, c_1..c_4)) * scale + x``.
    The five ``torch.cat`` copies (64+96+128+160+192 channels per pixel, and as many again in the
    backward) are replaced by one ``[N,H,W,64+4*32]`` buffer: conv_k reads its first ``64 + 32(k-1)``
    channels (channel stride 192) and writes its 32 outputs right behind them; in the backward the
    input gradients of all five convs are summed in place in a second buffer of the same shape
    (``srx_conv2d_bwd_data(..., accumulate=1)``), which is exactly the adjoint of the concatenations.
    """

    @staticmethod
    def forward(ctx, x: Tensor, scale: float, states, masters, *wb):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'dense_block.input')
        n, h, w, c0 = x.shape
        g = states[0].cout
        total = c0 + 4 * g
        m = n * h * w
        L = _lib.lib()
        s = _stream()
        dev = x.device
        buf = torch.empty((n, h, w, total), dtype=torch.float32, device=dev)
        call('srx_copy_channels', _p(x), c0, 0, _p(buf), total, 0, c0, m, 0, s)
        descs = []
        for k in range(5):
            st = states[k]
            cin = c0 + k * g
            last = k == 4
            d = Conv2dDesc(n, h, w, cin, total, st.cout, st.cout if last else total, st.k, st.k, st.stride, st.pad, 0,
                           st.act, st.slope, 0, st.precision)
            descs.append(d)
            dref = C.byref(d)
            st.pack(masters[k], d)
            bias = wb[2 * k + 1]
            bp = None if bias is None else _p(_chk(bias.detach(), 'dense_block.bias'))
            nws = L.srx_conv2d_fwd_ws_floats(dref)
            ws = _ws(nws, x) if nws else None
            if last:  # y = conv5(...) * scale + x in the conv's epilogue (esrgan/residual.py:86)
                y = torch.empty_like(x)
                call('srx_conv2d_fwd_residual', dref, _p(buf), _p(st.wpk_fwd), bp, _p(x), float(scale), _p(y), _p(ws), nws, s)
            else:
                call('srx_conv2d_fwd', dref, _p(buf), _p(st.wpk_fwd), bp, buf.data_ptr() + 4 * cin, None, _p(ws), nws, s)
        ctx.states, ctx.descs, ctx.scale = states, descs, float(scale)
        ctx.dims = (n, h, w, c0, g, total, m)
        ctx.params = wb
        ctx.packs = [st.wpk_bwd for st in states]
        ctx.save_for_backward(buf)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        (buf,) = ctx.saved_tensors
        dy = _chk(dy, 'dense_block.grad')
        n, h, w, c0, g, total, m = ctx.dims
        L = _lib.lib()
        s = _stream()
        dev = dy.device
        gbuf = torch.empty((n, h, w, total), dtype=torch.float32, device=dev)
        g5 = torch.empty_like(dy)
        call('srx_axpby', _p(dy), _p(dy), _p(g5), dy.numel(), ctx.scale, 0.0, s)
        grads = [None] * 10
        for k in (4, 3, 2, 1, 0):
            st, d = ctx.states[k], ctx.descs[k]
            dref = C.byref(d)
            cin = c0 + k * g
            if k == 4:
                gk, ldg = g5.data_ptr(), st.cout
            else:  # this conv's output gradient is complete AND masked: the data gradient of conv k+1 finished the slice
                gk, ldg = gbuf.data_ptr() + 4 * cin, total
            wparam, bparam = ctx.params[2 * k], ctx.params[2 * k + 1]
            bias_done = False
            if ctx.needs_input_grad[4 + 2 * k]:
                sink = _sink(wparam)
                dw = None if sink is not None else torch.empty((st.cout, st.cin, st.k, st.k), dtype=torch.float32,
                                                                 device=dev)
                bptr = None
                if bparam is not None and ctx.needs_input_grad[5 + 2 * k]:  # bias gradient in the same kernel
                    bsink = _sink(bparam)
                    if (bsink is None) == (sink is None):
                        if bsink is None:
                            grads[2 * k + 1] = torch.empty(st.cout, dtype=torch.float32, device=dev)
                        bptr = _p(grads[2 * k + 1] if bsink is None else bsink)
                        bias_done = True
                queue = wgrad_queue[0]
                if queue is not None and sink is not None:
                    queue.add(d, _p(buf), gk, _p(sink), bptr, (buf, gbuf, g5))
                else:
                    nws = L.srx_conv2d_bwd_weight_ws_floats(dref)
                    call('srx_conv2d_bwd_weight', dref, _p(buf), gk, _p(dw if sink is None else sink),
                         0 if sink is None else 1, bptr, _p(_ws(nws, dy)), nws, s)
                grads[2 * k] = dw
            if bparam is not None and ctx.needs_input_grad[5 + 2 * k] and not bias_done:
                sink = _sink(bparam)
                db = None if sink is not None else torch.empty(st.cout, dtype=torch.float32, device=dev)
                nws = L.srx_colsum_ws_floats(m, st.cout)
                call('srx_colsum', gk, _p(db if sink is None else sink), m, st.cout, ldg, 0 if sink is None else 1,
                     _p(_ws(nws, dy)), nws, s)
                grads[2 * k + 1] = db
            if k > 0 or ctx.needs_input_grad[0]:
                nws = L.srx_conv2d_bwd_data_ws_floats(dref)
                ws = _ws(nws, dy) if nws else None
                # conv5 writes all `total` channels; the others add their share to the first `cin`.  The slice of
                # conv k's output, channels [cin - g, cin), is complete with this call (every later conv has added
                # its share): the LeakyReLU backward of conv k (esrgan/residual.py:81-84) is applied to it on the way out
                if k > 0:
                    prev = ctx.states[k - 1]
                    call('srx_conv2d_bwd_data_act', dref, gk, _p(ctx.packs[k]), _p(buf), prev.slope, cin - g, cin,
                         0 if k == 4 else 1, _p(gbuf), _p(ws), nws, s)
                else:
                    call('srx_conv2d_bwd_data', dref, gk, _p(ctx.packs[k]), _p(gbuf), 1, _p(ws), nws, s)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(dy)
            call('srx_copy_channels', _p(gbuf), total, 0, _p(dx), c0, 0, c0, m, 0, s)
            call('srx_axpby', _p(dx), _p(dy), _p(dx), dy.numel(), 1.0, 1.0, s)
        return (dx, None, None, None, *grads)


def dense_block(x: Tensor, scale: float, convs) -> Tensor:
    """``convs``: the block's five ``layers.Conv2d`` modules (conv1..conv5)."""
    from .layers import _w
    wb = []
    for c in convs:
        wb += [_w(c.weight), _w(c.bias)]
    return _DenseBlock.apply(x, scale, [c._st for c in convs], [c.weight for c in convs], *wb)


class RDBPack:
    """The bf16 weight streams of a chain of dense blocks for the one-launch forward (``srx_rdb_fwd``): one
    ``srx_rdb_pack`` launch for all blocks, re-run when any weight changed (the same staleness key as ``ConvState.pack``)."""

    def __init__(self):
        self.buf = self.buf_bwd = self.table = None
        self._key = None
        self._has_bwd = False
        self.per_block = 0

    def ensure(self, states, masters, need_bwd: bool = False) -> None:
        ws = [w for row in masters for w in row]
        key = (ws[0].data_ptr(), len(ws), _pack_epoch[0], states[0][0].model_epoch[0], sum(w._version for w in ws))
        if key == self._key and (self._has_bwd or not need_bwd):
            return
        dev = ws[0].device
        if self.table is None or self.table.numel() != len(ws) or self.table.device != dev:
            self.per_block = int(_lib.lib().srx_rdb_packed_bytes())
            self.table = torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64).to(dev)
            self.buf = torch.empty(len(states) * self.per_block, dtype=torch.uint8, device=dev)
            self._ptrs = [w.data_ptr() for w in ws]
        elif self._ptrs != [w.data_ptr() for w in ws]:
            self.table.copy_(torch.tensor([w.data_ptr() for w in ws], dtype=torch.int64))
            self._ptrs = [w.data_ptr() for w in ws]
        call('srx_rdb_pack', _p(self.table), len(states), _p(self.buf), _stream())
        self._has_bwd = bool(need_bwd)
        if need_bwd:  # the transposed, tap-flipped streams of the data-gradient chain (srx_rdb_bwd)
            if self.buf_bwd is None or self.buf_bwd.numel() != self.buf.numel() or self.buf_bwd.device != dev:
                self.buf_bwd = torch.empty_like(self.buf)
            call('srx_rdb_pack_bwd', _p(self.table), len(states), _p(self.buf_bwd), _stream())
        self._key = key

    def block_ptr(self, i: int) -> int:
        return self.buf.data_ptr() + i * self.per_block

    def bwd_ptr(self, i: int) -> int:
        return self.buf_bwd.data_ptr() + i * self.per_block


def rdb_fused_ok(states, wb_row, c0: int) -> bool:
    """The one-launch dense block applies to the reference's geometry (64 + 4 x 32 channels, 3x3 / stride 1 / pad 1,
    LeakyReLU on conv1..4, biases) with bf16 products; ``SRX_NO_RDB_FUSED=1`` keeps the per-conv launches (A/B runs)."""
    if _dev.NO_RDB_FUSED or c0 != 64 or len(states) != 5:
        return False
    for k, st in enumerate(states):
        if (st.precision != 1 or st.k != 3 or st.stride != 1 or st.pad != 1 or st.shuffle or st.up or st.cin != 64 + 32 * k
                or st.cout != (64 if k == 4 else 32) or st.act != (ACT_NONE if k == 4 else ACT_LRELU)):
            return False
        if k < 4 and st.slope != states[0].slope:
            return False
    return all(wb_row[2 * k + 1] is not None for k in range(5))


class _RRDBTrunk(Function):
    """ESRGAN's chain of residual-in-residual dense blocks (esrgan/generator.py:54-56,70; esrgan/residual.py:81-86,
    125-128) as ONE autograd node, so that nothing but convolutions touches the activations.

    Every dense block owns a ``[N,H,W,64+4*32]`` buffer as in ``_DenseBlock``; here the block's INPUT lives in the
    first 64 channels of its own buffer, because the conv5 of the block before wrote it there
    (``conv5 * scale + x`` in the conv epilogue, 192-channel row stride on both sides).  Per dense block the forward
    is five conv launches (before: a channel copy, five convs, and every third block an ``out * 0.2 + x`` pass that
    stays -- it has two addends).  In the backward the block's output gradient ``dy`` is never scaled or copied:
    conv5's data gradient takes ``scale`` and adds ``dy`` onto the first 64 channels (``srx_conv2d_bwd_data_ex``), its
    weight gradient takes ``scale`` in the slab reduction (``srx_conv2d_bwd_weight_multi_scaled``), and conv1's data
    gradient writes the block's input gradient as a dense tensor, adding what conv2..5 left in the shared gradient
    buffer.  Per RRDB one elementwise pass is left in each direction (before: four and ten).

    With bf16 products the scale meets the rounding in the other order than in autograd's graph: conv5's data and weight
    gradient multiply bf16(dy) and apply ``scale`` to the fp32 result, where the chain of separate nodes rounds
    ``scale * dy`` -- two evaluation orders of the same product, both within bf16 rounding of the exact gradient and
    closer to each other than either is to it (``tests/test_esrgan_gpu.py::test_rrdb_modules_alone_equal_the_trunk_node``).
    """

    @staticmethod
    def forward(ctx, x: Tensor, rdb_scales, rrdb_scale: float, states, masters, pack, *wb):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'rrdb_trunk.input')
        n, h, w, c0 = x.shape
        nb = len(states)
        g = states[0][0].cout
        total = c0 + 4 * g
        m = n * h * w
        L = _lib.lib()
        s = _stream()
        need_bwd = any(ctx.needs_input_grad)
        # without a backward pass four rotating buffers are enough (an RRDB reads its first block's input at its end)
        pool = [torch.empty((n, h, w, total), dtype=torch.float32, device=x.device) for _ in range(nb + 1 if need_bwd else 4)]
        bufs = [pool[i % len(pool)] for i in range(nb + 1)]
        call('srx_copy_channels', _p(x), c0, 0, _p(bufs[0]), total, 0, c0, m, 0, s)
        descs = []
        y = None
        # bf16 products: a whole dense block is ONE launch (srx_rdb_fwd: the five convs on 8x8 pixel tiles with the
        # intermediates resident in LDS); exact fp32 keeps one launch per conv
        fused = pack is not None and all(rdb_fused_ok(states[i], wb[10 * i:10 * i + 10], c0) for i in range(nb))
        if fused:
            pack.ensure(states, masters, need_bwd)
        for i in range(nb):
            buf, nxt = bufs[i], bufs[i + 1]
            row = []
            for k in range(5):
                st = states[i][k]
                cin = c0 + k * g
                d = Conv2dDesc(n, h, w, cin, total, st.cout, total, st.k, st.k, st.stride, st.pad, 0, st.act, st.slope, 0,
                               st.precision)
                row.append(d)
                dref = C.byref(d)
                st.fused_only = fused  # fused: forward and data gradients read RDBPack's bf16 streams, nothing reads wpk_*
                if fused:
                    continue
                st.pack(masters[i][k], d)
                bias = wb[10 * i + 2 * k + 1]
                bp = None if bias is None else _p(_chk(bias.detach(), 'rrdb_trunk.bias'))
                nws = L.srx_conv2d_fwd_ws_floats(dref)
                ws = _ws(nws, x) if nws else None
                if k == 4:  # the next block's input = conv5 * scale + x, x = the first 64 channels of this buffer (:86)
                    call('srx_conv2d_fwd_residual', dref, _p(buf), _p(st.wpk_fwd), bp, _p(buf), float(rdb_scales[i]), _p(nxt),
                         _p(ws), nws, s)
                else:
                    call('srx_conv2d_fwd', dref, _p(buf), _p(st.wpk_fwd), bp, buf.data_ptr() + 4 * cin, None, _p(ws), nws, s)
            if fused:
                biases = (C.c_void_p * 5)(*[_p(_chk(wb[10 * i + 2 * k + 1].detach(), 'rrdb_trunk.bias')) for k in range(5)])
                if i % 3 == 2:  # end of an RRDB: its `out * 0.2 + x` (:128) is the second addend of this block's last epilogue
                    out, out_ld = (nxt, total) if i < nb - 1 else (torch.empty_like(x), c0)
                    call('srx_rdb_fwd', n, h, w, _p(buf), total, pack.block_ptr(i), biases, float(rdb_scales[i]),
                         float(states[i][0].slope), float(rrdb_scale), _p(bufs[i - 2]), total, _p(out), out_ld, s)
                    if i == nb - 1:
                        y = out
                else:
                    call('srx_rdb_fwd', n, h, w, _p(buf), total, pack.block_ptr(i), biases, float(rdb_scales[i]),
                         float(states[i][0].slope), 1.0, None, 0, _p(nxt), total, s)
            descs.append(row)
            if i % 3 == 2 and not fused:  # end of an RRDB: out * 0.2 + x, x = the input of its first dense block (:128)
                first = bufs[i - 2]
                if i == nb - 1:
                    y = torch.empty_like(x)
                    call('srx_axpby_channels', _p(nxt), total, 0, _p(first), total, 0, _p(y), c0, 0, c0, m, float(rrdb_scale),
                         1.0, s)
                else:
                    call('srx_axpby_channels', _p(nxt), total, 0, _p(first), total, 0, _p(nxt), total, 0, c0, m,
                         float(rrdb_scale), 1.0, s)
        ctx.states, ctx.descs = states, descs
        ctx.pack = pack if fused else None
        ctx.scales = ([float(v) for v in rdb_scales], float(rrdb_scale))
        ctx.dims = (n, h, w, c0, g, total, m, nb)
        ctx.params = wb
        if need_bwd:
            ctx.save_for_backward(*bufs[:nb])
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        bufs = ctx.saved_tensors
        grad = _chk(dy, 'rrdb_trunk.grad')
        n, h, w, c0, g, total, m, nb = ctx.dims
        rdb_scales, rrdb_scale = ctx.scales
        L = _lib.lib()
        s = _stream()
        dev = grad.device
        grads = [None] * (10 * nb)
        queue = wgrad_queue[0]
        rrdb_grad = None
        for i in range(nb - 1, -1, -1):
            j = i % 3
            buf = bufs[i]
            gbuf = torch.empty((n, h, w, total), dtype=torch.float32, device=dev)
            # `grad` is the gradient of the value this block produced, up to the factor `eff`: the last block of an RRDB
            # receives the RRDB's output gradient, of which it sees rrdb_scale (out * 0.2 + x, :128)
            if j == 2:
                rrdb_grad, eff = grad, rrdb_scale
            else:
                eff = 1.0
            # the `+ x` of :86 hands eff * grad to the block's input; the first block's input is also the RRDB's x
            skip, skip_scale = grad, eff
            if j == 0 and ctx.pack is None:
                skip = torch.empty_like(grad)
                call('srx_axpby', _p(grad), _p(rrdb_grad), _p(skip), grad.numel(), eff, 1.0, s)
                skip_scale = 1.0
            dx = torch.empty((n, h, w, c0), dtype=torch.float32, device=dev)
            if ctx.pack is not None:
                # bf16 products: the whole data-gradient chain of the block in one launch (srx_rdb_bwd): g5 = eff * scale * dy,
                # the masked slice gradients g4..g1 (what the weight gradients below read) into gbuf, dx = the x share + skip
                # (+ the RRDB's own skip gradient where this block is the first of its RRDB)
                extra = rrdb_grad if j == 0 else None
                call('srx_rdb_bwd', n, h, w, _p(grad), c0, float(eff * rdb_scales[i]), _p(buf), total, ctx.pack.bwd_ptr(i),
                     float(ctx.states[i][0].slope), _p(gbuf), total, _p(skip), c0, float(skip_scale), _p(extra), c0, _p(dx), c0, s)
            keep = (buf, gbuf, grad)
            # conv1 + conv2 and conv3 + conv4 read the same buffer and their output gradients are adjacent slices of gbuf:
            # queued as two 64-column weight-gradient problems instead of four 32-column ones (half a tile of padding each)
            wsinks = [_sink(ctx.params[10 * i + 2 * k]) for k in range(4)]
            bsinks = [_sink(ctx.params[10 * i + 2 * k + 1]) for k in range(4)]
            paired = (queue is not None and all(v is not None for v in wsinks) and all(v is not None for v in bsinks)
                      and all(ctx.needs_input_grad[6 + 10 * i + kk] for kk in range(8)))
            if paired:
                for lo in (0, 2):
                    st = ctx.states[i][lo + 1]
                    dp = Conv2dDesc(n, h, w, c0 + (lo + 1) * g, total, 2 * g, total, st.k, st.k, st.stride, st.pad, 0, st.act,
                                    st.slope, 0, st.precision)
                    queue.add_pair(dp, c0 + lo * g, _p(buf), gbuf.data_ptr() + 4 * (c0 + lo * g),
                                   (_p(wsinks[lo]), _p(wsinks[lo + 1])), (_p(bsinks[lo]), _p(bsinks[lo + 1])), keep)
            for k in (4, 3, 2, 1, 0):
                st, d = ctx.states[i][k], ctx.descs[i][k]
                cin = c0 + k * g
                if k == 4:  # conv5's output gradient is the block's: a dense 64-channel tensor (the forward wrote 192-strided)
                    d = Conv2dDesc(n, h, w, cin, total, st.cout, st.cout, st.k, st.k, st.stride, st.pad, 0, st.act, st.slope, 0,
                                   st.precision)
                    gk, wscale = grad.data_ptr(), eff * rdb_scales[i]
                else:  # complete and masked: the data gradient of conv k+1 finished this slice of the shared buffer
                    gk, wscale = gbuf.data_ptr() + 4 * cin, 1.0
                dref = C.byref(d)
                wparam, bparam = ctx.params[10 * i + 2 * k], ctx.params[10 * i + 2 * k + 1]
                if paired and k < 4:
                    pass  # (queued above: the queue runs after the whole backward pass, when every slice of gbuf is complete)
                elif ctx.needs_input_grad[6 + 10 * i + 2 * k]:
                    sink = _sink(wparam)
                    dw = None if sink is not None else torch.empty((st.cout, st.cin, st.k, st.k), dtype=torch.float32,
                                                                     device=dev)
                    bptr = None
                    if bparam is not None and ctx.needs_input_grad[7 + 10 * i + 2 * k]:
                        bsink = _sink(bparam)
                        if (bsink is None) != (sink is None):
                            raise RuntimeError('rrdb_trunk: weight and bias of a conv must both (or neither) have a gradient sink')
                        if bsink is None:
                            grads[10 * i + 2 * k + 1] = torch.empty(st.cout, dtype=torch.float32, device=dev)
                        bptr = _p(grads[10 * i + 2 * k + 1] if bsink is None else bsink)
                    if queue is not None and sink is not None:
                        queue.add(d, _p(buf), gk, _p(sink), bptr, keep, wscale)
                    else:
                        nws = L.srx_conv2d_bwd_weight_ws_floats(dref)
                        one = lambda v: (C.c_void_p * 1)(v)  # noqa: E731
                        call('srx_conv2d_bwd_weight_multi_scaled', dref, 1, 1, one(_p(buf)), one(gk),
                             one(_p(dw if sink is None else sink)), 0 if sink is None else 1, one(bptr),
                             (C.c_float * 1)(wscale), _p(_ws(nws, grad)), nws, s)
                    grads[10 * i + 2 * k] = dw
                elif bparam is not None and ctx.needs_input_grad[7 + 10 * i + 2 * k]:
                    raise RuntimeError('rrdb_trunk: a bias gradient without its weight gradient is not implemented')
                if ctx.pack is not None:
                    continue  # (the block's five data gradients are one launch, below)
                e = _lib.DgradEpilogue()
                if k > 0:  # LeakyReLU backward of conv k on the slice this call completes (esrgan/residual.py:81-84)
                    e.act_out, e.act_slope, e.c_lo, e.c_hi = _p(buf), ctx.states[i][k - 1].slope, cin - g, cin
                if k == 4:    # overwrites all 192 channels; `+ x`: the block's output gradient lands on the first 64
                    e.out_scale = eff * rdb_scales[i]
                    e.addend, e.addend_ld, e.addend_channels, e.addend_scale = _p(skip), c0, c0, skip_scale
                    out, dd = gbuf, d
                elif k > 0:
                    e.accumulate = 1
                    out, dd = gbuf, d
                else:         # the block's input gradient, dense: this conv's share + the first 64 channels of the buffer
                    e.addend, e.addend_ld, e.addend_channels = _p(gbuf), total, c0
                    dd = Conv2dDesc(n, h, w, c0, c0, st.cout, total, st.k, st.k, st.stride, st.pad, 0, st.act, st.slope, 0,
                                    st.precision)
                    out = dx
                ddref = C.byref(dd)
                nws = L.srx_conv2d_bwd_data_ws_floats(ddref)
                ws = _ws(nws, grad) if nws else None
                call('srx_conv2d_bwd_data_ex', ddref, gk, _p(st.wpk_bwd), _p(out), C.byref(e), _p(ws), nws, s)
            grad = dx
        return (grad, None, None, None, None, None, *grads)


def rrdb_trunk(x: Tensor, rrdbs) -> Tensor:
    """``rrdbs``: the generator's ``ResidualInResidualDenseBlock`` modules, in order (``nn.Sequential(*blocks)(x)``)."""
    from .layers import _w
    if not rrdbs:
        return x
    states, masters, wb, scales = [], [], [], []
    for rr in rrdbs:
        for rdb in (rr.RDB1, rr.RDB2, rr.RDB3):
            convs = (rdb.conv1[0], rdb.conv2[0], rdb.conv3[0], rdb.conv4[0], rdb.conv5)
            states.append([c._st for c in convs])
            masters.append([c.weight for c in convs])
            scales.append(rdb.scale_ratio)
            for c in convs:
                wb += [_w(c.weight), _w(c.bias)]
    pack = rrdbs[0].__dict__.get('_rdb_pack')
    if pack is None:
        pack = rrdbs[0].__dict__.setdefault('_rdb_pack', RDBPack())
    return _RRDBTrunk.apply(x, scales, 0.2, states, masters, pack, *wb)


class _Upsample2x(Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        ctx.set_materialize_grads(False)
        x = _chk(x, 'upsample.input')
        n, h, w, c = x.shape
        y = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.float32, device=x.device)
        call('srx_upsample_nearest2x_fwd', _p(x), _p(y), n, h, w, c, _stream())
        ctx.shape = (n, h, w, c)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        dy = _chk(dy, 'upsample.grad')
        n, h, w, c = ctx.shape
        dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dy.device)
        call('srx_upsample_nearest2x_bwd', _p(dy), _p(dx), n, h, w, c, _stream())
        return dx


def upsample_nearest2x(x: Tensor) -> Tensor:
    """F.interpolate(x, scale_factor=2, mode='nearest') (esrgan/generator.py:73,76) on NHWC."""
    return _Upsample2x.apply(x)
