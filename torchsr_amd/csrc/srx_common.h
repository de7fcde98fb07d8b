// Internal helpers shared by the HIP translation units of libsrx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/srx.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local error message (srx_last_error)
void srx_set_error(const char* fmt, ...);

#define SRX_FAIL(code, ...)      \
  do {                           \
    srx_set_error(__VA_ARGS__);  \
    return (code);               \
  } while (0)

#define SRX_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) SRX_FAIL(SRX_E_BADARG, __VA_ARGS__); \
  } while (0)

#define SRX_CHECK_LAUNCH(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) SRX_FAIL(SRX_E_HIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// api.cpp: developer switches, read from the environment once at load time (never on a launch path)
struct SrxDevSwitches {
  bool no_rt36, no_wgrad_rows, no_bn_bwd_fuse, no_bn_fwd_fuse, no_first3, no_c64, force_plan, no_wgrad_dma, no_wgrad_lin, no_wino, old_wgrad_reduce, wino_no_tail;
  int wgrad_nsplit, wgrad_rows_nsplit, first3_wgs_per_cu, thin_fwd_rows, reserved_cus, c64_ablate, rdb_ablate, wino_zsplit, wino_bn;
  int s2_mode;  // SRX_S2_MODE: strided data gradients -- bit 0: no fused-class kernel (gconv_s2f_kernel), bit 1: no two-group 64x64 tile
  int plan[4];  // SRX_FORCE_PLAN = "BM,BN,split,ks"
};
const SrxDevSwitches& srx_dev();
// compute units the launch plans size their grids for (device CUs less srx_set_reserved_cus)
extern "C" int srx_plan_cus(void);

// api.cpp: optional per-launch event timing (srx_prof_start / srx_prof_stop / srx_prof_get)
bool srx_prof_on();
bool srx_prof_take(const char* name, double flops, hipEvent_t* e0, hipEvent_t* e1);
// launch `kernel`; with the profiler on, as a dispatch that carries its own start / stop events
#define SRX_LAUNCH_PROF(name, flops, kernel, grid, block, lds, st, ...)                              \
  do {                                                                                               \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                       \
    if (srx_prof_on() && srx_prof_take(name, flops, &e0__, &e1__))                                   \
      hipExtLaunchKernelGGL(kernel, grid, block, (std::uint32_t)(lds), st, e0__, e1__, 0, __VA_ARGS__); \
    else                                                                                             \
      hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                 \
  } while (0)

// thin.hip: 3-channel-side convolutions on v_mfma_f32_4x4x1 (internal, called from gconv.hip)
bool srx_thin_wgrad_applicable(const srx_conv2d_t* d);
size_t srx_thin_wgrad_ws_floats(const srx_conv2d_t* d);
int srx_thin_wgrad(const srx_conv2d_t* d, const float* x, const float* dy, float* dw, int accumulate, float* ws,
                   size_t ws_floats, hipStream_t st);
bool srx_thin_fwd_applicable(const srx_conv2d_t* d);
// 3x3 / stride 1 / pad 1, <= 4 -> 64 channels, bias + ReLU / LeakyReLU: the first layers of the discriminators and of VGG19
bool srx_first3_fwd_applicable(const srx_conv2d_t* d);
int srx_first3_fwd(const srx_conv2d_t* d, const float* in, const float* wpk, int Kp, const float* bias, float* out, hipStream_t st, int out_bf16 = 0);
bool srx_thin_dgrad_applicable(const srx_conv2d_t* d);
int srx_thin_pack(const srx_conv2d_t* d, const float* w, float* p, int mode, hipStream_t st);
int srx_thin_fwd(const srx_conv2d_t* d, const float* in, const float* wpk, const float* bias, float* out, int n_out,
                 hipStream_t st, int in_bf16 = 0);  // in_bf16: `in` is a bf16 tensor (precision = 2 only)

// rowtile.hip: 3x3 / 64 -> 64 convolutions with few pixels (the SRGAN residual tower), 36 pixels per CU
bool srx_rt36_applicable(const srx_conv2d_t* d);
int srx_rt36_rows(const srx_conv2d_t* d);  // workgroups = rows of the BatchNorm partial table
// optional second epilogue of a data gradient: first pass of the backward of the BatchNorm (+ PReLU: prelu != null) layer
// whose output gradient this launch produces (see rowtile.hip); part: [workgroups][2 * 64 + 4]
struct srx_rt36_bn_t { const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta;
                       const float* prelu; float* part; };
// optional transform of a forward launch's INPUT while it is staged: the input tensor is the output of the conv BELOW and
// the patch pixels become act(BatchNorm(in)) (training-mode statistics already finalised; prelu == null: no activation) on
// their way into LDS; the workgroup also writes the transformed values of its own 36 pixels to z_out -- the tensor the
// separate normalise pass would have produced, which the backward pass reads (see rowtile.hip)
struct srx_rt36_bnl_t { const float* mean; const float* invstd; const float* gamma; const float* beta; const float* prelu;
                        const float* res;  // optional addend behind the activation (a block's skip input), same shape as `in`
                        float* z_out; };
// the same for a DATA GRADIENT whose input is the gradient arriving at the OUTPUT of a BatchNorm (+ PReLU) layer: the patch
// pixels become that layer's input gradient (second pass of its backward: dy = gamma * invstd * (dz - sum_dz / M - xhat *
// sum_dz_xhat / M), the sums already finalised) on their way into LDS -- this needs the layer's forward input y as a second
// patch -- and the workgroup writes its own 36 pixels of dy to dy_out (the conv's weight gradient reads it)
struct srx_rt36_bnb_t { const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta;
                        const float* prelu; const float* sums; float inv_m; float* dy_out; };
int srx_rt36_run(const srx_conv2d_t* d, const float* in, const float* wpk, const float* bias, const float* residual,
                 float* out, float* part, int act, float slope, hipStream_t st, const srx_rt36_bn_t* bn = nullptr,
                 const srx_rt36_bnl_t* bnl = nullptr, const srx_rt36_bnb_t* bnb = nullptr);

static inline hipStream_t srx_stream(void* s) { return (hipStream_t)s; }
static inline int64_t srx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t srx_roundup(int64_t a, int64_t b) { return srx_cdiv(a, b) * b; }

// m < 2^24 assumed (checked on the host).  q = m / d, r = m % d with one float multiply.
__device__ __forceinline__ void srx_divmod(int m, int d, float inv_d, int& q, int& r) {
  q = __float2int_rz(__int2float_rn(m) * inv_d);
  r = m - q * d;
  if (r < 0) { q -= 1; r += d; }
  if (r >= d) { q += 1; r -= d; }
}

// Raw buffer descriptor over [p, p + bytes): loads whose byte offset falls outside return 0, which is
// how the conv kernels read padding taps -- no branch, no select, and every load is issued
// unconditionally so the compiler can count outstanding loads exactly (s_waitcnt vmcnt(N)).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t srx_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 srx_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// wave-uniform value into an SGPR (scalar branches / scalar address math)
__device__ __forceinline__ int srx_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float srx_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// wino.hip's weight transform for ONE (row channel, contraction channel) pair `idx` (U = G g G^T, sixteen values scattered to
// where wino_kernel's waves load them); shared with gconv.hip's pack_table_kernel, which refreshes the Winograd-domain weights
// of trainable layers after an optimiser step.  transpose = 1: the layer's data gradient (channels swapped, taps flipped).
__device__ __forceinline__ void srx_wino_pack_one(const float* __restrict__ w, float* __restrict__ upk, int Cout, int Cin,
                                                  int transpose, int64_t idx) {
  // transpose: 0 the layer; 1 its data gradient; 2 the layer with a PixelShuffle(2) store: GEMM row r = (sub-pixel ij, channel cc)
  // is the conv's output channel cc * 4 + ij (the order gconv.hip's packs use), so a lane's four consecutive rows are four
  // consecutive channels of ONE output pixel
  const int R = transpose == 1 ? Cin : Cout, K = transpose == 1 ? Cout : Cin;  // rows (the GEMM's channels out) and contraction length
  const int r = (int)(idx / K), k = (int)(idx - (int64_t)r * K);
  const int rsrc = transpose == 2 ? (r % (Cout / 4)) * 4 + r / (Cout / 4) : r;
  float g[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      g[i][j] = transpose == 1 ? w[(((size_t)k * Cin + r) * 3 + (2 - i)) * 3 + (2 - j)] : w[(((size_t)rsrc * Cin + k) * 3 + i) * 3 + j];
  float t[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t[0][j] = g[0][j];
    t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
    t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
    t[3][j] = g[2][j];
  }
  const int nch = K / 32, rt = R / 32;
  const int jg = r >> 5, mrow = r & 31, kc = k / 32, kk = k % 32, quad = kk >> 2, e = kk & 3, s = quad >> 1, hh = quad & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xi = 4 * i + j;
      upk[(((((size_t)xi * rt + jg) * nch + kc) * 4 + s) * 64 + hh * 32 + mrow) * 4 + e] = u[j];
    }
  }
}
