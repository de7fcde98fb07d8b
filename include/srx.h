/*
 * srx.h -- C ABI of the MI355X (gfx950) SRGAN/ESRGAN hot-path library `libsrx_hip.so`.
 *
 * This is the drop-in boundary (level 2).  The reference
 * (roclark/torchsr) has no FFI of its own: every FLOP of its hot path is issued
 * through `torch.nn` modules.  Each entry point below therefore cites the
 * reference construct (file:line under /root/reference) whose device work it
 * replaces.  The Python host (`torchsr_amd/`) binds these with ctypes and wraps
 * them in `torch.autograd.Function`s behind the reference's own
 * `Generator` / `Discriminator` / `VGGLoss` / trainer surface.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / HIP types in signatures
 *    (`stream` is a `hipStream_t` passed as `void*`; NULL = default stream).
 *  - all tensors are fp32, activations are NHWC ([N][H][W][C_s]) with a channel
 *    stride C_s that is a multiple of 4 (3-channel images are stored with C_s=4,
 *    4th channel zero).  Parameters stay in the reference's own layouts
 *    (Conv2d OIHW, Linear [out][in]) so `state_dict()` is byte compatible;
 *    MFMA-friendly packed copies are produced by `srx_conv2d_pack`.
 *  - every call is asynchronous on `stream`, allocates nothing, never
 *    synchronises and is hipGraph-capture safe.  Scratch memory is handed in by
 *    the caller (`*_ws_floats` queries say how much).
 *  - return value: 0 = ok, otherwise an SRX_E_* code; a thread-local message is
 *    available through `srx_last_error`.
 *  - re-entrant: no mutable global state except kernel attributes set once.
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRX_VERSION 100 /* 0.1.0, mirrors torchsr/__version__.py:13 */

enum {
  SRX_OK = 0,
  SRX_E_BADARG = 1,      /* shape / alignment / null pointer problem */
  SRX_E_UNSUPPORTED = 2, /* configuration outside what the kernels implement */
  SRX_E_HIP = 3,         /* a HIP runtime call failed */
  SRX_E_WORKSPACE = 4    /* workspace too small */
};

enum { SRX_ACT_NONE = 0, SRX_ACT_RELU = 1, SRX_ACT_LRELU = 2, SRX_ACT_PRELU = 3 };

int srx_version(void);
/* copies the calling thread's last error message (NUL terminated) into buf */
int srx_last_error(char* buf, size_t n);
/* sha256 (64 hex digits) of the sources this binary was compiled from -- the build's identity: the host refuses (or
 * rebuilds) a library whose digest differs from the sources lying next to it (torchsr_amd/_lib.py).  No reference counterpart. */
int srx_build_info(char* buf, size_t n);
/* number of compute units of the current device (used by the host for launch heuristics) */
int srx_device_cus(void);
/* Compute units the launch plans may count on = srx_device_cus() less `k` reserved ones (0..128).  Data-parallel training
 * overlaps the gradient all-reduce (torchsr/srgan/trainer.py:142-157: DistributedDataParallel) with the backward pass, and
 * RCCL's channel workgroups then hold CUs: a grid cut for exactly 256 CUs runs two rounds on 248.  Set ONCE, before any
 * model is built (row counts of BatchNorm partial tables and workspace sizes follow the plans); also SRX_RESERVED_CUS at
 * load time.  srx_plan_cus() returns the count the plans use. */
int srx_set_reserved_cus(int k);
int srx_plan_cus(void);
/* Measurement aid (bench.py's data-parallel rehearsal; no reference counterpart): occupies `k` compute units -- k workgroups that each
 * claim a whole CU's LDS, so nothing else becomes resident there (`whole_cu` != 0), or k co-resident workgroups that only
 * spin (`whole_cu` == 0) -- until *stop_flag (host-visible, 4 bytes) is non-zero or `max_ms` milliseconds have passed on
 * the device clock, whichever comes first; every wave reaches that exit. */
int srx_occupy_cus(int k, int whole_cu, const int* stop_flag, int max_ms, void* stream);
/* Per-launch timing of the convolution kernels (measurement aid for bench.py's roofline leg; no
 * reference counterpart).  Between start and stop every conv kernel is dispatched with its own start /
 * stop HIP events on its stream (hipExtLaunchKernelGGL) -- the main kernel only, not its fix-up / reduce
 * companion.  Not to be used inside a hipGraph capture.  srx_prof_stop returns the number of records; srx_prof_get
 * (after stop) synchronises on record i and returns its kernel name, duration and FLOPs. */
int srx_prof_start(int max_launches);
int srx_prof_stop(void);
int srx_prof_get(int i, char* name, size_t n, float* ms, double* flops);

/* ------------------------------------------------------------------ layout */
/* NCHW [N][C][H][W] -> NHWC [N][H][W][Cs] (channels C..Cs-1 written as 0).
 * Replaces the implicit NCHW contract of every reference module's forward
 * (srgan/generator.py:60, srgan/discriminator.py:71, srgan/loss.py:36). */
int srx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cs, void* stream);
/* NHWC [N][H][W][Cs] -> NCHW [N][C][H][W]; also `torch.flatten(out, 1)` in NCHW
 * order for the discriminator head (srgan/discriminator.py:86). */
int srx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int Cs, void* stream);

/* ------------------------------------------------------------------ conv2d */
/* One nn.Conv2d instance: srgan/residual.py:27,64,67; srgan/generator.py:38,48,58;
 * srgan/discriminator.py:32-59; torchvision VGG19 cfg 'E' (srgan/loss.py:30-31);
 * esrgan/residual.py:31-56; esrgan/generator.py:36-52. */
typedef struct srx_conv2d {
  int32_t N, H, W;    /* input batch / spatial size */
  int32_t Cin;        /* Conv2d.in_channels  */
  int32_t Cin_s;      /* channel stride of the input tensor  (>= Cin, multiple of 4) */
  int32_t Cout;       /* Conv2d.out_channels */
  int32_t Cout_s;     /* channel stride of the output tensor (>= Cout, multiple of 4);
                         with shuffle=2 this is the stride of the shuffled tensor (>= Cout/4) */
  int32_t KH, KW, stride, pad;
  int32_t shuffle;    /* 0, or 2: nn.PixelShuffle(2) fused into the store
                         (srgan/residual.py:28): output is [N][2Ho][2Wo][Cout/4] */
  int32_t act;        /* fused epilogue: SRX_ACT_NONE / RELU / LRELU (after bias) */
  float   slope;      /* LeakyReLU negative_slope */
  int32_t up;         /* 0/1, or 2: F.interpolate(scale_factor=2, mode='nearest') of the
                         input fused into the gather (esrgan/generator.py:73,76): the forward reads
                         pixel (h >> 1, w >> 1), the upsampled tensor is never written; H,W are then
                         the size of the tensor BEFORE upsampling (stride 1, no shuffle).  The data
                         gradient is taken at the upsampled size in scratch and summed per 2x2 block,
                         the weight gradient reads a scratch copy (both from `ws`) */
  int32_t precision;  /* 0: exact fp32 MFMAs.  1: bf16 products with fp32 accumulation for the forward
                         and the stride-1 data gradient -- the reference's torch.cuda.amp.autocast
                         region (srgan/trainer.py:379-383, esrgan/trainer.py:418-484); tensors stay
                         fp32 in memory, operands are rounded when staged into LDS; strided data
                         gradients and the 3-channel (thin) layers remain fp32.  2: only on the forward
                         of a 64 -> <= 4 channel layer (the generators' output conv): bf16 products there
                         as well -- inference with every conv in bf16 (test.upscale(precision='bf16')) */
} srx_conv2d_t;

/* sizes (in floats) of the packed weight copies and of scratch buffers */
size_t srx_conv2d_packed_fwd_floats(const srx_conv2d_t* d);
size_t srx_conv2d_packed_bwd_floats(const srx_conv2d_t* d);
size_t srx_conv2d_fwd_ws_floats(const srx_conv2d_t* d);
size_t srx_conv2d_bwd_data_ws_floats(const srx_conv2d_t* d);
size_t srx_conv2d_bwd_weight_ws_floats(const srx_conv2d_t* d);
/* rows of the per-channel (sum, sum of squares) partial table written by
 * srx_conv2d_fwd when `bn_partials` is non-NULL: table is [rows][Cout][2] */
int srx_conv2d_stat_rows(const srx_conv2d_t* d);

/* launch plan the library will use (for profiling / the bench's roofline bookkeeping):
 * which = 0 forward, 1 data gradient; out[6] = {tile rows BM, tile cols BN, tail split-K factor, workgroups,
 * KS (wave groups splitting K inside a workgroup), multi (1: stride-parity classes in one launch; 2: in one workgroup per tile, gconv_s2f_kernel)}.
 * BM = 144 is the 128 + 16 row tile, BM = 36 the row-tile kernel of rowtile.hip. */
int srx_conv2d_plan(const srx_conv2d_t* d, int which, int* out);

/* OIHW master weights -> packed forward ([Cout_p][K_p], K=(kh,kw,ci)) and, when
 * wpk_bwd != NULL, packed data-gradient operands (per stride-parity class,
 * taps flipped, [Cin_p][K'_p], K'=(tap,co)). */
int srx_conv2d_pack(const srx_conv2d_t* d, const float* w_oihw, float* wpk_fwd, float* wpk_bwd, void* stream);

/* The same for every conv of a model in ONE launch (after an optimiser step): build fills a host buffer of
 * srx_pack_table_bytes(n) bytes with one record per packed operand; the caller keeps a copy of it in
 * device memory (all pointers and sizes in it are fixed) and replays it with srx_pack_table_run.
 * wpk_bwd[i] may be NULL (layer whose input needs no gradient). */
size_t srx_pack_table_bytes(int n_layers);
int srx_pack_table_build(const srx_conv2d_t* descs, int n, const float* const* w_oihw, float* const* wpk_fwd,
                         float* const* wpk_bwd, void* host_table, int* n_records, long long* max_elems);
int srx_pack_table_run(const void* dev_table, int n_records, long long max_elems, void* stream);
/* appends to a host table under construction (srx_pack_table_build's output; room: srx_pack_table_bytes) the record that refreshes a
 * layer's Winograd-domain weights (srx_wino_pack(d, w, upk, transpose)) in the same launch */
int srx_pack_table_add_wino(void* host_table, int* nrec, long long* max_elems, const srx_conv2d_t* d, const float* w, float* upk,
                            int transpose);

/* y = act(conv(x, W) + bias).  bias may be NULL.  bn_partials may be NULL; when
 * given it receives per-row-block sums for the training-mode BatchNorm that
 * follows (srgan/residual.py:65,68; srgan/discriminator.py:36-60). */
int srx_conv2d_fwd(const srx_conv2d_t* d, const float* x, const float* wpk_fwd, const float* bias,
                   float* y, float* bn_partials, float* ws, size_t ws_floats, void* stream);
/* ---------------------------------------------------------------- bf16-native inference chain (round 4) */
/* `torchsr test` (torchsr/test.py:57-62) with --precision bf16: from the first conv's output to the last conv's input the
 * generator's activations are STORED as bf16 NHWC (128 bytes per 64-channel pixel).  Entry points take `void*` bf16 tensors.
 *
 * srx_conv3x3_c64_bf16_*: nn.Conv2d(64, Cout, 3, 1, 1) with Cout a multiple of 64 -- the residual blocks' convs with the
 * eval-mode BatchNorm folded in (srgan/residual.py:64-68,86-91), the generator's conv2 (srgan/generator.py:48,76-78) and
 * the sub-pixel layers (srgan/residual.py:27-29, Cout = 256 with PixelShuffle(2) in the store).
 *   pack: w OIHW fp32 [Cout][64][3][3], bias [Cout] or NULL, out_scale [Cout] or NULL (multiplies the weights of each
 *         output channel: the folded BatchNorm's gamma / sqrt(var + eps)); wpk: srx_conv3x3_c64_bf16_packed_bytes(Cout) bytes.
 *   fwd:  y = act(conv(x) + bias) [+ residual], act(v) = v > 0 ? v : v * slope (none: slope = 1, PReLU: its parameter);
 *         x: bf16 [N][H][W][64]; y: bf16 [N][H][W][y_cs] (shuffle = 2: [N][2H][2W][y_cs], channels 0..63), y_cs a multiple of
 *         8; residual: bf16 laid out like y, a tensor of its own, not with shuffle.  Any H, W; tensors above 4 GiB are fine
 *         (64-bit row bases). */
size_t srx_conv3x3_c64_bf16_packed_bytes(int Cout);
int srx_conv3x3_c64_bf16_pack(const float* w, const float* bias, const float* out_scale, int Cout, int shuffle, void* wpk,
                              void* stream);
int srx_conv3x3_c64_bf16_fwd(int N, int H, int W, int Cout, int shuffle, const void* x, const void* wpk, float slope,
                             const void* residual, void* y, int y_cs, void* stream);
/* host only: out[3] = {rows per chunk, chunks per column strip, column strips} of the launch _fwd makes at this size.  A chunk is
 * addressed with 32-bit byte offsets from its first row, so rows per chunk x W x 128 bytes stays below 4 GiB (shorter chunks) */
int srx_conv3x3_c64_bf16_plan(int N, int H, int W, int Cout, int* out);
/* fp32 <-> bf16 (round to nearest even), n elements, a multiple of 4: the two ends of the chain */
int srx_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int srx_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* The generator's output conv (nn.Conv2d(64, 3, 9, 1, 4), srgan/generator.py:58,80) reading a bf16 input: d->precision
 * must be 2 (bf16 products in the 64 -> 3 layer); y is fp32 [N][H][W][4]. */
int srx_conv2d_fwd_bf16in(const srx_conv2d_t* d, const void* x_bf16, const float* wpk_fwd, const float* bias, float* y,
                          void* stream);

/* The same conv (nn.Conv2d(64, Cout <= 3, 9, 1, 4), srgan/generator.py:58,80) as a GEMM whose N is supplied by the taps -- 9
 * row taps x 3 channels = 27 MFMA rows, K = 9 column taps x 64 channels -- on v_mfma_f32_32x32x16_bf16 (thin9.hip): x bf16
 * [N][H][W][64], y fp32 [N][H][W][4] (channels >= Cout written as 0), bf16 products, fp32 accumulation; any H, W.
 * pack: w OIHW fp32 [Cout][64][9][9], bias [Cout] or NULL -> srx_conv9x9_c64_thin_bf16_packed_bytes() bytes. */
size_t srx_conv9x9_c64_thin_bf16_packed_bytes(void);
int srx_conv9x9_c64_thin_bf16_pack(const float* w, const float* bias, int Cout, void* wpk, void* stream);
int srx_conv9x9_c64_thin_bf16_fwd(int N, int H, int W, const void* x, const void* wpk, float* y, void* stream);

/* y = act(conv(x, W) + bias) * out_scale + residual, residual laid out like y (not for shuffle layers).
 * With the eval-mode BatchNorm folded into W and bias by the host this is a whole `x + BN(conv(.))` of the
 * residual block (srgan/residual.py:86-91, srgan/generator.py:77-78) in one kernel; with out_scale = 0.2 it
 * is `conv5 * scale_ratio + x` of the dense block (esrgan/residual.py:86). */
int srx_conv2d_fwd_residual(const srx_conv2d_t* d, const float* x, const float* wpk_fwd, const float* bias,
                            const float* residual, float out_scale, float* y, float* ws, size_t ws_floats, void* stream);
/* dx = conv_transpose(dy, W)  (autograd of nn.Conv2d wrt its input).  accumulate != 0 adds into dx
 * (stride-1 layers on the generic kernel): the dense block's convs share one 192-channel input buffer
 * (channel stride Cin_s, the first Cin channels are this layer's input), so their input gradients sum in
 * place of the torch.cat adjoint (esrgan/residual.py:81-85). */
int srx_conv2d_bwd_data(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, float* dx,
                        int accumulate, float* ws, size_t ws_floats, void* stream);
/* dx = conv_transpose(dy, W) + addend (laid out like dx, not dx itself): the gradient of `x` in `x + f(conv(x))` --
 * the skip connection of the residual block (srgan/residual.py:86-91) -- without autograd's separate add pass.
 * Stride-1 layers. */
int srx_conv2d_bwd_data_add(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* addend,
                            float* dx, float* ws, size_t ws_floats, void* stream);
/* The same followed by the backward of the activation that PRODUCED this conv's input, for the input channels
 * [c_lo, c_hi): dx[.., c] = (accumulate ? dx[.., c] : 0) + conv_transpose(dy, W)[.., c], then for c in the range
 * dx[.., c] *= (x[.., c] > 0 ? 1 : slope), x = that activation's output = the conv's saved input (laid out like dx).
 * In the VGG19 feature stack (srgan/loss.py:30-31,52: conv, ReLU, conv, ReLU, ...) every ReLU backward then rides
 * in the epilogue of the data gradient above it (range = all channels); in ESRGAN's dense block
 * (esrgan/residual.py:81-85) the conv that completes the gradient of a 32-channel slice of the shared buffer also
 * applies that slice's LeakyReLU backward.  Stride-1 layers, generic kernel; range in whole channel quads. */
int srx_conv2d_bwd_data_act(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* x, float slope,
                            int c_lo, int c_hi, int accumulate, float* dx, float* ws, size_t ws_floats, void* stream);
/* The general form of the three calls above, for a chain of blocks whose tensors have different channel strides:
 *   dx[m][c] = (accumulate ? dx[m][c] : 0) + out_scale * conv_transpose(dy, W)[m][c]
 *              + (c < addend_channels ? addend_scale * addend[m * addend_ld + c] : 0),
 * followed by the activation backward on [c_lo, c_hi) when act_out is given.  ESRGAN's RRDB trunk (esrgan/generator.py:
 * 54-56,70; esrgan/residual.py:81-86,125-128) keeps every dense block's input in the first 64 channels of that block's
 * 192-channel buffer: with this call conv5's data gradient takes `scale_ratio` and the block's own output gradient
 * (the `+ x` of :86, a dense 64-channel tensor) in its epilogue, and conv1's writes the block's input gradient as a
 * dense tensor, adding what the other four convs left in the first 64 channels of the shared gradient buffer --
 * the three elementwise passes per block that autograd runs for `out * 0.2 + x` are gone.  Zero fields mean: no
 * accumulate, scales 1, addend laid out like dx, all channels.  Stride-1 layers on the generic kernel; an addend
 * excludes accumulate; strides and channel counts in whole quads. */
typedef struct srx_dgrad_epilogue {
  int accumulate;
  float out_scale;
  const float* addend;
  int addend_ld;
  int addend_channels;
  float addend_scale;
  const float* act_out;
  float act_slope;
  int c_lo, c_hi;
} srx_dgrad_epilogue_t;
int srx_conv2d_bwd_data_ex(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, float* dx,
                           const srx_dgrad_epilogue_t* e, float* ws, size_t ws_floats, void* stream);
/* dx = conv^T(dy) [+ addend] AND, in the same launch, the first pass of the backward of the BatchNorm (+ PReLU) layer whose
 * output gradient dx is -- the layer that fed this conv in the forward pass (srgan/residual.py:86-90: conv2's data gradient
 * arrives at bn1 + prelu, conv1's (plus the skip gradient) at the previous block's bn2): table[row block][2C+4] = per-channel
 * sums of dz = dx * act'(bn(y)) and dz * xhat, and the PReLU slope partial in columns 2C and 2C+1.  bn_prelu: the slope
 * (device scalar) or NULL for a BatchNorm without activation.  Finish with srx_bn_act_bwd_finish(..., rows, 2, ...).
 * Only layers for which srx_conv2d_bwd_data_bn_rows returns non-zero (3x3, 64 -> 64, stride 1, few pixels: the residual tower). */
int srx_conv2d_bwd_data_bn_rows(const srx_conv2d_t* d);
int srx_conv2d_bwd_data_bn(const srx_conv2d_t* d, const float* dy, const float* wpk_bwd, const float* addend, float* dx,
                           const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                           const float* bn_beta, const float* bn_prelu, float* table, void* stream);
/* srx_conv2d_bwd_data_bn with the conv's output gradient produced on the way in: dout is the gradient arriving at the OUTPUT of
 * the BatchNorm (+ PReLU) layer above this conv (in_*: that layer's forward input y, statistics, parameters, slope or NULL, and
 * its finalised backward sums from srx_bn_act_bwd_finish(..., dy = NULL)); the second pass of that layer's backward runs while
 * the data gradient stages its input, and dy_out receives the conv's output gradient (its weight gradient reads it).
 * table == NULL: no BatchNorm below (then the bn_* pointers are ignored).  Same layers as srx_conv2d_bwd_data_bn. */
int srx_conv2d_bwd_data_bn_in_ok(const srx_conv2d_t* d);
int srx_conv2d_bwd_data_bn_in(const srx_conv2d_t* d, const float* dout, const float* in_y, const float* in_mean,
                              const float* in_invstd, const float* in_gamma, const float* in_beta, const float* in_prelu,
                              const float* in_sums, float* dy_out, const float* wpk_bwd, const float* addend, float* dx,
                              const float* bn_y, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                              const float* bn_beta, const float* bn_prelu, float* table, void* stream);
/* y = conv(act(BatchNorm(y_in))) in ONE launch: y_in is the output of the conv below (srgan/residual.py:86-88: conv1 -> bn1 ->
 * prelu -> conv2), the training-mode statistics are already finalised (srx_bn_finalize), and the normalise + activate pass
 * runs while the conv stages its input; act_out receives the activation tensor (what srx_bn_act_fwd would have written: the
 * backward pass and the weight gradient of this conv read it).  bn_prelu: the slope (device scalar) or NULL for no
 * activation.  bn_residual: NULL, or a tensor added behind the activation -- `x + bn2(conv2(.))` at the end of a residual block
 * (srgan/residual.py:90-91) formed while the NEXT block's conv1 stages its input (a second patch load).  bn_partials as in srx_conv2d_fwd.  Only layers for which srx_conv2d_fwd_bn_in_ok returns 1 (3x3, 64 -> 64,
 * stride 1, few pixels: the residual tower at training sizes). */
int srx_conv2d_fwd_bn_in_ok(const srx_conv2d_t* d);
int srx_conv2d_fwd_bn_in(const srx_conv2d_t* d, const float* y_in, const float* bn_mean, const float* bn_invstd,
                         const float* bn_gamma, const float* bn_beta, const float* bn_prelu, const float* bn_residual,
                         float* act_out, const float* wpk, const float* bias, float* y, float* bn_partials, void* stream);
int srx_bn_act_bwd_finish(const float* dout, const float* y, const float* mean, const float* invstd, const float* gamma,
                          const float* beta, const float* table, int rows, int prelu_cols, float* sums, float* dy, int64_t M,
                          int C, int act, float slope, const float* prelu, float* dgamma_acc, float* dbeta_acc,
                          float* dprelu_acc, void* stream);
/* dw (OIHW) = autograd of nn.Conv2d wrt its weight; accumulate != 0 adds into dw (a .grad buffer)
 * instead of overwriting it.  db (may be NULL) receives the bias gradient
 * sum_m dy[m][co] under the same flag: the kernel stages every dy row anyway.  The same flag exists on srx_colsum, srx_linear_bwd_weight,
 * srx_prelu_bwd; srx_bn_act_bwd_reduce takes optional accumulation targets. */
int srx_conv2d_bwd_weight(const srx_conv2d_t* d, const float* x, const float* dy, float* dw_oihw,
                          int accumulate, float* db, float* ws, size_t ws_floats, void* stream);
/* The same for `nprob` (<= 72) problems of ONE geometry in one launch.  Autograd produces the weight gradients of
 * the generator's 33 identical 3x3 64->64 convs (srgan/residual.py:64,67; srgan/generator.py:48) one by one
 * between the data gradients, each 0.68 GFLOP -- too small to fill the chip -- and nothing reads them before
 * optimizer.step() (srgan/trainer.py:386-387,468-469), so the host may collect (x, dy) pairs during the backward
 * pass and hand them over together.  `per_out` consecutive problems are segments of one gradient and are summed
 * into one output (the discriminator's real and fake passes, srgan/trainer.py:446-450): xs / dys hold nprob
 * pointers, dws / dbs hold nprob / per_out (dbs, or single entries of it, may be NULL). */
size_t srx_conv2d_bwd_weight_multi_ws_floats(const srx_conv2d_t* d, int nprob);
int srx_conv2d_bwd_weight_multi(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs,
                                const float* const* dys, float* const* dws, int accumulate, float* const* dbs,
                                float* ws, size_t ws_floats, void* stream);
/* The same with one multiplier per output (out_scales: nprob / per_out host floats, NULL = all 1), applied to the
 * weight and the bias gradient: the dy handed over stands for out_scale * dy.  The dense block's conv5 sees the
 * block's output gradient times scale_ratio (esrgan/residual.py:86), which then is never written out. */
int srx_conv2d_bwd_weight_multi_scaled(const srx_conv2d_t* d, int nprob, int per_out, const float* const* xs,
                                       const float* const* dys, float* const* dws, int accumulate, float* const* dbs,
                                       const float* out_scales, float* ws, size_t ws_floats, void* stream);
/* Pairs: every problem is TWO convs of d->Cout / 2 output channels each that read the same input buffer and whose
 * output gradients are adjacent channel slices of one tensor -- conv1 + conv2 and conv3 + conv4 of a dense block
 * (esrgan/residual.py:81-85: conv k reads the first 64 + 32 (k - 1) channels of the block's buffer and its output gradient
 * is the next 32-channel slice of the gradient buffer).  d describes the pair as one layer: Cin = the larger input
 * width, Cout = both slices; the first conv has only cin_lo input channels, the rest of its rows is not written.
 * Alone, a 32-column problem leaves half of every 64-column MFMA tile multiplying padding; paired, 10-17 % of the tile
 * is unused.  dws_lo / dws_hi (and dbs_lo / dbs_hi, both or neither) hold nprob pointers each. */
int srx_conv2d_bwd_weight_multi_pair(const srx_conv2d_t* d, int nprob, const float* const* xs, const float* const* dys,
                                     float* const* dws_lo, float* const* dws_hi, int cin_lo, int accumulate,
                                     float* const* dbs_lo, float* const* dbs_hi, float* ws, size_t ws_floats, void* stream);

/* ------------------------------------------------- elementwise / reductions */
/* out[c] = sum_m x[m][c]  (bias gradient of Conv2d / Linear); ws >= 2*rows*C floats */
size_t srx_colsum_ws_floats(int64_t M, int C);
int srx_colsum(const float* x, float* out, int64_t M, int C, int Cs, int accumulate, float* ws, size_t ws_floats,
               void* stream);
/* dx = dy * act'(y) for sign-recoverable activations fused into a conv epilogue (ReLU, LeakyReLU) */
int srx_act_bwd_from_out(const float* dy, const float* y, float* dx, int64_t n, int act, float slope, void* stream);
/* the same on a channel slice of wider NHWC tensors (rows M, channels [0, C), row strides ldy / ly / ldx in
 * floats; dx may alias dy): the LeakyReLU of the dense block's conv1..4, whose outputs and gradients live
 * in shared 192-channel buffers instead of torch.cat copies (esrgan/residual.py:81-85) */
int srx_act_bwd_from_out_strided(const float* dy, int ldy, const float* y, int ly, float* dx, int ldx, int64_t M, int C,
                                 int act, float slope, void* stream);
/* nn.PReLU (one shared slope): srgan/generator.py:39, srgan/residual.py:29,66 */
int srx_prelu_fwd(const float* x, const float* slope, float* y, int64_t n, void* stream);
/* dx and d(slope); ws >= 1024 floats */
int srx_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* dslope,
                  int accumulate, int64_t n, float* ws, void* stream);
/* nn.LeakyReLU as a standalone op (ESRGAN, esrgan/residual.py:36-55) */
int srx_lrelu_fwd(const float* x, float* y, int64_t n, float slope, void* stream);
/* y = a*x + b*z  (residual scaling of esrgan/residual.py:86,128; torch.add of srgan/generator.py:78) */
int srx_axpby(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream);
/* ESRGAN ResidualDenseBlock.forward (esrgan/residual.py:65-86) as ONE launch, bf16 products / fp32 accumulate:
 *   c_k = LeakyReLU(conv_k(cat(x, c_1..c_{k-1})) + b_k, slope), k = 1..4;  out = (conv_5(cat(x, c_1..c_4)) + b_5) * scale + x
 * buf: NHWC [N][H][W][ld], ld >= 192: x in channels 0..63 on entry; c_1..c_4 (fp32, what the backward pass reads) are
 * written to channels 64..191.  out: [N][H][W][out_ld], channels 0..63 (the next block's buffer in an RRDB chain).
 * wpk: this block's stream of srx_rdb_packed_bytes() bytes written by srx_rdb_pack; bias5: HOST array of the five
 * device bias pointers (32, 32, 32, 32, 64 floats).  One workgroup per 8x8 pixel tile; any H, W. */
size_t srx_rdb_packed_bytes(void);
/* w_table_dev: DEVICE array of 5 * nblk pointers to the OIHW fp32 weights (conv1..conv5 of block 0, of block 1, ...);
 * dst: nblk * srx_rdb_packed_bytes() bytes.  One launch for all blocks (after an optimiser step). */
int srx_rdb_pack(const float* const* w_table_dev, int nblk, void* dst, void* stream);
/* extra (may be NULL): out = (that) * post_scale + extra -- the `out * 0.2 + x` that ends a ResidualInResidualDenseBlock
 * (esrgan/residual.py:128) in the epilogue of its third dense block; [N][H][W][extra_ld], channels 0..63.  `extra` may be an
 * OLDER block's buffer but never `out`, and `out` is never `buf` itself (neighbouring tiles read their halo from it while
 * this one writes): both are refused. */
int srx_rdb_fwd(int N, int H, int W, float* buf, int ld, const void* wpk, const float* const* bias5, float scale,
                float slope, float post_scale, const float* extra, int extra_ld, float* out, int out_ld, void* stream);
/* The block's data-gradient chain (autograd of the five convs and four LeakyReLUs of esrgan/residual.py:81-86 with
 * respect to their inputs) as ONE launch, the forward's schedule run in reverse:
 *   g5 = scale * dy;  g_j = LeakyReLU'(c_j) * sum_{k > j} conv_k^T(g_k)[c_j], j = 4..1;  dx = sum_k conv_k^T(g_k)[x] + skip_scale * skip
 * dy: [N][H][W][dy_ld] (the gradient of the block's output; `scale` = the block's scale_ratio times whatever factor the
 * caller's chain rule has collected); buf: the block's saved buffer (x, c1..c4, ld >= 192: read for the masks);
 * gbuf: [N][H][W][gld >= 192], receives g1..g4 (fp32) at channels 64..191 -- the output gradients the convs' weight
 * gradients are computed from (srx_conv2d_bwd_weight_multi_pair / _scaled; conv5's is dy); skip: [N][H][W][skip_ld]: the
 * gradient that reaches x around the convs (`+ x` of :86); dx: [N][H][W][dx_ld] channels 0..63, must not alias dy / skip.
 * wpk_bwd: this block's stream written by srx_rdb_pack_bwd (transposed, tap-flipped; same size as the forward's). */
int srx_rdb_pack_bwd(const float* const* w_table_dev, int nblk, void* dst, void* stream);
/* extra (may be NULL): a second gradient added to dx as it is (the RRDB's own skip connection reaching its first block) */
int srx_rdb_bwd(int N, int H, int W, const float* dy, int dy_ld, float scale, const float* buf, int ld, const void* wpk_bwd,
                float slope, float* gbuf, int gld, const float* skip, int skip_ld, float skip_scale, const float* extra,
                int extra_ld, float* dx, int dx_ld, void* stream);
/* Per-step scalars without a per-step device->host sync: append n <= 4 device scalars (*a, *b, *c, *d) as one
 * 4-float record to ring[(*counter % cap) * 4 ...] and increment *counter (device int32).  The launch is the same every
 * step, so it sits inside the replayed hipGraph; the host reads `cap` records back in one copy.  Replaces the
 * reference's `wandb.log({... 'train-loss': loss}, step)` tensor -> host read on every step
 * (srgan/trainer.py:393-399,459-466; esrgan/trainer.py:393-399,471-478) */
int srx_ring_push(const float* a, const float* b, const float* c, const float* d, int n, float* ring, int* counter,
                  int cap, void* stream);
/* y[m][y_off+c] = a*x[m][x_off+c] + b*z[m][z_off+c], c < C: the same on channel slices of tensors with their own
 * channel strides (`out * 0.2 + x` of an RRDB, esrgan/residual.py:128, between two dense-block buffers) */
int srx_axpby_channels(const float* x, int x_cs, int x_off, const float* z, int z_cs, int z_off, float* y, int y_cs,
                       int y_off, int C, int64_t M, float a, float b, void* stream);
/* channel concat / split of NHWC tensors: dst[m][dst_off+c] (+)= src[m][src_off+c], c < C
 * (torch.cat((x, conv1, ...), dim=1) of the dense block, esrgan/residual.py:82-85, and its adjoint) */
int srx_copy_channels(const float* src, int src_cs, int src_off, float* dst, int dst_cs, int dst_off, int C,
                      int64_t M, int accumulate, void* stream);
/* F.interpolate(scale_factor=2, mode='nearest') (esrgan/generator.py:73,76) and its adjoint; x is [N][H][W][C] */
int srx_upsample_nearest2x_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int srx_upsample_nearest2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* nn.Sigmoid of the SRGAN discriminator head (srgan/discriminator.py:68) */
int srx_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int srx_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* ------------------------------------------------ on-device data pipeline */
/* TrainData.__getitem__ of the reference (dataset.py:88-99,121-125: RandomCrop, RandomHorizontalFlip,
 * RandomVerticalFlip, ToTensor) for a batch of decoded images resident in device memory.
 * imgs: device array of N pointers to HWC uint8 RGB images; meta: device int32 [N][6] =
 * {H, W, top, left, hflip, vflip}; out: [N][3][crop][crop] float in [0, 1]. */
int srx_crop_flip_u8(const void* const* imgs, const int32_t* meta, float* out_nchw, int N, int crop, void* stream);
/* Resize(crop // scale, BICUBIC) of the reference's low-resolution branch (dataset.py:93-96): antialiased
 * Keys bicubic (a = -0.5) reduction by an integer factor on NCHW floats; quantize != 0 rounds the result
 * to 8 bits like the PIL image the reference converts back with ToTensor. */
int srx_bicubic_down(const float* in_nchw, float* out_nchw, int N, int C, int H, int W, int scale, int quantize,
                     void* stream);

/* -------------------------------------------------------------- batch norm */
/* nn.BatchNorm2d(C), eps 1e-5, momentum 0.1 (srgan/residual.py:65,68;
 * srgan/generator.py:49; srgan/discriminator.py:36-60).
 * stats from an activation tensor: partial table [rows][C][2] */
int srx_bn_stat_rows(int64_t M);
int srx_bn_rows_per_block(int64_t M);  /* rows summed per partial row (and per row block of the backward reduction) */
int srx_bn_partial_stats(const float* y, float* partials, int64_t M, int C, void* stream);
/* reduce partials -> save_mean, save_invstd (biased variance); update running stats
 * (unbiased variance, momentum) and num_batches_tracked (int64) when non-NULL */
int srx_bn_finalize(const float* partials, int rows, int64_t M, int C, float eps, float momentum,
                    float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, void* stream);
/* eval mode: save_mean = running_mean, save_invstd = rsqrt(running_var + eps) */
int srx_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                      float* save_mean, float* save_invstd, void* stream);
/* out = act(gamma*(y-mean)*invstd + beta) [+ residual];  act: NONE / LRELU(slope) / PRELU(*prelu) */
int srx_bn_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, const float* residual, float* out, int64_t M, int C, int act,
                   float slope, const float* prelu, void* stream);
/* backward, pass 1: sums[0..C) = sum dz, sums[C..2C) = sum dz*xhat, sums[2C] = d(prelu slope)
 * (dz = dout * act'(bn(y))).  ws >= srx_bn_bwd_ws_floats */
size_t srx_bn_bwd_ws_floats(int64_t M, int C);
int srx_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float* sums, int64_t M, int C, int act,
                          float slope, const float* prelu, float* dgamma_acc, float* dbeta_acc,
                          float* dprelu_acc, float* ws, size_t ws_floats, void* stream);
/* backward, pass 2: dy = gamma*invstd*(dz - sum_dz/M - xhat*sum_dzxhat/M) (training) or
 * dy = gamma*invstd*dz (eval, training=0) */
int srx_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, const float* sums, float* dy, int64_t M,
                         int C, int act, float slope, const float* prelu, int training, void* stream);

/* one-call forms used by the trainers.  `groups` > 1: the M rows are `groups` equal consecutive row ranges that are
 * normalised independently -- several forward calls of the reference run as one batch (the discriminator on the real
 * and on the fake images, srgan/trainer.py:446-447): save_mean / save_invstd are [groups][C], sums [groups][2C+4],
 * the running statistics are updated once per group in order, num_batches_tracked advances by `groups`, and the
 * parameter gradients sum over the groups.  Row blocks (conv tiles, srx_bn_stat_rows) must not straddle groups. */
int srx_bn_train_fwd(const float* y, const float* partials, int rows, int64_t M, int C, int groups, float eps,
                     float momentum, const float* gamma, const float* beta, const float* residual, float* out, int act,
                     float slope, const float* prelu, float* save_mean, float* save_invstd, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, void* stream);
int srx_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float* sums, float* dy, int64_t M, int C, int groups, int act, float slope,
                   const float* prelu, int training, float* dgamma_acc, float* dbeta_acc, float* dprelu_acc,
                   float* ws, size_t ws_floats, void* stream);

/* ------------------------------------------------- Winograd F(2x2, 3x3) (round 5) */
/* nn.Conv2d(Cin, Cout, 3, 1, 1) of wide layers -- the VGG19 feature extractor of the perceptual loss (srgan/loss.py:30-54;
 * torchvision cfg 'E') -- with 2.25x fewer multiplications than the direct form, all in fp32 (results differ from the direct
 * fp32 convolution by rounding only: ~5e-7 of the tensor's scale).  Layers: 3x3 / stride 1 / pad 1, precision 0, Cin and Cout
 * multiples of 32 with channel strides equal to them, even H and W, act NONE or RELU (srx_wino_applicable).
 *   pack:     upk = G g G^T of every channel pair, srx_wino_packed_floats(d) floats, in the order the kernel's waves load it;
 *             transpose = 1 packs the layer's DATA GRADIENT (a 3x3 / pad 1 conv of dy with the channels swapped and the taps flipped)
 *   fwd:      y = act(conv(x) + bias)
 *   bwd_data: dx = conv^T(dy), multiplied by the ReLU mask (relu_out > 0, laid out like dx; NULL: none) of the layer below -- the
 *             fold srx_conv2d_bwd_data_act does for the direct kernel
 * ws: srx_wino_ws_floats(d, which) floats (which: 0 forward, 1 data gradient; non-zero when the planner splits the input
 * channels over several workgroups, or cuts the tiles of the last, partly filled round of the chip along them).  x != y.
 * srx_wino_plan: out[6] = {BN, channel splits of every tile, workgroups, tile blocks, chunks per workgroup, parts per tile of the
 * last round (1: none)} (host only). */
int srx_wino_applicable(const srx_conv2d_t* d);
size_t srx_wino_packed_floats(const srx_conv2d_t* d);
size_t srx_wino_ws_floats(const srx_conv2d_t* d, int which);
int srx_wino_plan(const srx_conv2d_t* d, int which, int* out);
int srx_wino_pack(const srx_conv2d_t* d, const float* w_oihw, float* upk, int transpose, void* stream);
int srx_wino_fwd(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* ws, size_t ws_floats,
                 void* stream);
int srx_wino_bwd_data(const srx_conv2d_t* d, const float* dy, const float* upk_t, const float* relu_out, float* dx, float* ws,
                      size_t ws_floats, void* stream);
/* forward of a linear layer that a training-mode BatchNorm2d follows (srgan/discriminator.py:35-61): stats
 * [srx_wino_stat_rows(d)][Cout][2] = per tile block of 128 output pixels (32 consecutive 2x2 tiles, image-major) the per-channel
 * (sum, sum of squares) of y -- the table srx_bn_finalize / srx_bn_train_fwd take, as srx_conv2d_fwd's bn_partials */
int srx_wino_stat_rows(const srx_conv2d_t* d);
int srx_wino_fwd_stats(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, float* y, float* stats,
                       float* ws /* srx_wino_ws_floats(d, 2) floats */, size_t ws_floats, void* stream);
/* inference (functional.FoldedConv: conv with the eval-mode BatchNorm folded into weights and bias, srgan/residual.py:86-91):
 * y = act(conv(x) + bias) [+ residual], act = none / ReLU / LeakyReLU(d->slope) (a single-parameter PReLU is that); residual laid
 * out like y, a tensor of its own, may be NULL.  srx_wino_infer_applicable: as srx_wino_applicable, also for the 64 -> 64
 * layers the training path leaves to its row-tile kernel */
int srx_wino_infer_applicable(const srx_conv2d_t* d);
int srx_wino_fwd_act(const srx_conv2d_t* d, const float* x, const float* upk, const float* bias, const float* residual, float* y,
                     float* ws, size_t ws_floats, void* stream);
/* Measurement aid (tools/lab/wino_sweep.cpp: the sweep the planner's cost constants are fitted from; no reference counterpart): until
 * called again with (0, 0, 0), every Winograd plan of this process is (BN 32 / 64, channel splits of every tile, parts per tile of the
 * last round) instead of the planner's choice -- srx_wino_ws_floats / srx_wino_plan / the launches all follow it; combinations a
 * layer cannot run (BN not dividing Cout, more splits than 32-channel chunks, splits of a statistics / inference launch) fall back
 * to the planner. */
int srx_wino_force_plan(int bn, int zsplit, int tsplit);

/* ------------------------------------------------- bf16 storage of activations on the training path (round 5) */
/* Under autocast (esrgan/trainer.py:446,461) the reference's convs read and write half tensors.  These entry points keep the
 * activations and gradients BETWEEN the 3x3 / stride 1 / pad 1 convs of a frozen stack (VGG19[:36], srgan/loss.py:30-34,52-53)
 * as bf16 NHWC tensors.  The arithmetic is that of precision = 1 (bf16 products, fp32 accumulation): an operand is rounded
 * once, by its producer, instead of by every consumer's loader.  d->precision must be 1, Cin and Cout multiples of 64.
 * Packed weights are bf16: srx_conv3x3_bf16s_packed_bytes(d) bytes per copy (forward; data gradient = transposed, taps flipped).
 * which: 0 forward, 1 data gradient.  Tensors: `void*` = bf16. */
int srx_conv3x3_bf16s_applicable(const srx_conv2d_t* d);
size_t srx_conv3x3_bf16s_packed_bytes(const srx_conv2d_t* d);
int srx_conv3x3_bf16s_pack(const srx_conv2d_t* d, const float* w, void* wpk_fwd, void* wpk_bwd /* may be NULL */, void* stream);
size_t srx_conv3x3_bf16s_ws_floats(const srx_conv2d_t* d, int which);
/* y = [relu](conv(x) + bias): x bf16 [N][H][W][Cin]; y bf16 (y_is_bf16) or fp32 [N][H][W][Cout] */
int srx_conv3x3_bf16s_fwd(const srx_conv2d_t* d, const void* x, const void* wpk_fwd, const float* bias, int relu, void* y,
                          int y_is_bf16, float* ws, size_t ws_floats, void* stream);
/* dx = conv^T(dy) [* (relu_out > 0)]: dy bf16 [N][H][W][Cout]; relu_out (may be NULL): the bf16 output of the ReLU that produced
 * this layer's input; dx bf16 (dx_is_bf16) or fp32 [N][H][W][Cin] */
int srx_conv3x3_bf16s_bwd_data(const srx_conv2d_t* d, const void* dy, const void* wpk_bwd, const void* relu_out, void* dx,
                               int dx_is_bf16, float* ws, size_t ws_floats, void* stream);
/* the 3 -> 64 first layer in front of such a stack: srx_conv2d_fwd's kernel for it (precision = 1) with a bf16 output;
 * x fp32 [N][H][W][4], wpk_fwd: the layer's ordinary forward pack, y bf16 [N][H][W][64] */
int srx_conv2d_fwd_first3_to_bf16(const srx_conv2d_t* d, const float* x, const float* wpk_fwd, const float* bias, void* y,
                                  void* stream);
/* pools and the topmost activation backward of such a stack.  The pools read the fp32 output of the conv below them (the
 * choice of the maximum must not depend on bf16 rounding) and round what they hand on. */
int srx_maxpool2x2_fwd_to_bf16(const float* x, void* y, int N, int H, int W, int C, void* stream);
int srx_maxpool2x2_relu_bwd_bf16(const void* dy, const float* x, void* dx, int N, int H, int W, int C, void* stream);
int srx_act_bwd_from_out_to_bf16(const float* dy, const float* y, void* dx, int64_t n, int act, float slope, void* stream);

/* ----------------------------------------------------------------- pooling */
/* nn.MaxPool2d(2,2) of VGG19 (torchvision cfg 'E', srgan/loss.py:30-31); H, W even */
int srx_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int srx_maxpool2x2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, void* stream);
/* max-pool backward followed by the backward of the ReLU that produced x (conv, ReLU, MaxPool in cfg 'E') */
int srx_maxpool2x2_relu_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, void* stream);

/* ------------------------------------------------------------------ linear */
/* nn.Linear (srgan/discriminator.py:65,67; esrgan/discriminator.py:73,75).
 * x [B][K], w [J][K] (reference layout), y [B][J] = act(x w^T + bias) */
size_t srx_linear_ws_floats(int B, int K, int J);
int srx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int J,
                   int act, float slope, float* ws, size_t ws_floats, void* stream);
int srx_linear_bwd_data(const float* dy, const float* w, float* dx, int B, int K, int J, float* ws,
                        size_t ws_floats, void* stream);
int srx_linear_bwd_weight(const float* x, const float* dy, float* dw, int accumulate, int B, int K, int J,
                          void* stream);

/* ------------------------------------------------------------------ losses */
/* all losses reduce with mean; `loss` is one float on the device; ws >= 2048 floats.
 * nn.MSELoss (srgan/trainer.py:163,384), nn.L1Loss / F.l1_loss (srgan/loss.py:52; esrgan/trainer.py:163) */
int srx_mse_fwd(const float* a, const float* b, float* loss, int64_t n, float* ws, void* stream);
int srx_l1_fwd(const float* a, const float* b, float* loss, int64_t n, float* ws, void* stream);
/* da = gscale[0] * d(mean loss)/da ; db (if non-NULL) = -da */
int srx_mse_bwd(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n, void* stream);
int srx_l1_bwd(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n, void* stream);
/* F.l1_loss with an explicit divisor `count` (<= n): NHWC images carry a zero padding channel (3 of 4 stored channels are
 * real), and the reference's mean (esrgan/trainer.py:466) runs over the real elements only */
int srx_l1_fwd_count(const float* a, const float* b, float* loss, int64_t n, int64_t count, float* ws, void* stream);
int srx_l1_bwd_count(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n, int64_t count,
                     void* stream);
/* nn.BCELoss on probabilities against a constant label (srgan/trainer.py:164,446-447,456);
 * log terms clamped at -100 as torch does */
int srx_bce_fwd(const float* p, float target, float* loss, int64_t n, float* ws, void* stream);
int srx_bce_bwd(const float* p, float target, const float* gscale, float* dp, int64_t n, void* stream);
/* nn.BCEWithLogitsLoss(x - shift[0], target) (esrgan/trainer.py:164,451-453,468); shift may be NULL */
int srx_bce_logits_fwd(const float* x, const float* shift, float target, float* loss, int64_t n, float* ws, void* stream);
int srx_bce_logits_bwd(const float* x, const float* shift, float target, const float* gscale, float* dx, int64_t n, void* stream);
/* torch.mean over all elements (esrgan/trainer.py:451-452,468); ws >= 2048 floats */
int srx_mean_fwd(const float* x, float* out, int64_t n, float* ws, void* stream);
int srx_mean_bwd(const float* x, const float* gscale, float* dx, int64_t n, void* stream);
/* sum of squared error -> used for PSNR (srgan/trainer.py:296) is srx_mse_fwd */

/* ------------------------------------------------- discriminator head + loss */
/* The end of a discriminator pass inside the GAN steps -- last Linear layer -> [Sigmoid] -> adversarial loss -- as ONE
 * forward and ONE backward launch (as separate operators: ~25 one-workgroup launches per discriminator update).
 *   SRX_HEAD_SRGAN_D  srgan/discriminator.py:67-68 + srgan/trainer.py:446-448: rows [0, n_first) = D(real), the rest =
 *                     D(fake); out[0] = BCELoss(sigmoid(z_real), 1) + BCELoss(sigmoid(z_fake), 0), out[1], out[2] the terms
 *   SRX_HEAD_SRGAN_G  srgan/trainer.py:456-457: out[1] = BCELoss(sigmoid(z), 1), out[0] = addend[0] + adv_weight * out[1]
 *   SRX_HEAD_ESRGAN_D esrgan/discriminator.py:75 + esrgan/trainer.py:451-453: out[0] = (BCEWithLogits(z_real - mean(z_fake), 1)
 *                     + BCEWithLogits(z_fake - mean(z_real), 0)) / 2, both means differentiated; out[4], out[5] the means
 *   SRX_HEAD_ESRGAN_G esrgan/trainer.py:468-469: out[1] = BCEWithLogits(z - shift[0], 1) (shift = mean(D(real)), no
 *                     gradient), out[0] = addend[0] + adv_weight * out[1]
 * hidden [B][J] is the OUTPUT of the hidden layer's LeakyReLU(slope); z = hidden w2 + b2.  zp [B] receives what the backward
 * needs (probabilities for the SRGAN modes, logits for the ESRGAN ones), out at least 8 floats; loss (may be NULL) receives
 * out[0] once more (a tensor of its own for the caller's autograd tape).  B <= 256.
 * Backward, given g[0] = d(loss)/d(out[0]) on the device: dpre [B][J] = gradient of the hidden layer's PRE-activation
 * (LeakyReLU backward applied), dw2 [J], db2 [1], db1 [J] = column sums of dpre (the hidden layer's bias gradient); any of
 * the three may be NULL; accumulate != 0 adds to them.  (The gradient of `addend` is g[0] itself.) */
enum { SRX_HEAD_SRGAN_D = 0, SRX_HEAD_SRGAN_G = 1, SRX_HEAD_ESRGAN_D = 2, SRX_HEAD_ESRGAN_G = 3 };
typedef struct {
  int32_t mode, B, J, n_first;
  float slope, adv_weight;
} srx_gan_head_t;
int srx_gan_head_fwd(const srx_gan_head_t* h, const float* hidden, const float* w2, const float* b2, const float* shift,
                     const float* addend, float* zp, float* out, float* loss, void* stream);
int srx_gan_head_bwd(const srx_gan_head_t* h, const float* hidden, const float* w2, const float* zp, const float* out,
                     const float* shift, const float* g, float* dpre, float* dw2, float* db2, float* db1, int accumulate,
                     void* stream);

/* --------------------------------------------------------------- optimiser */
/* torch.optim.Adam(lr, betas, eps, weight_decay=0) over one flat buffer
 * (srgan/trainer.py:171-185).  `step` (int64 on device) is incremented first;
 * `lr` is a float on the device so StepLR (srgan/trainer.py:186-195) can change
 * it without re-capturing a graph.  grad_scale multiplies g first (1/world_size
 * after an all-reduce SUM). */
int srx_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1,
                  float beta2, float eps, float grad_scale, int64_t* step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
