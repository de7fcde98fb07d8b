// Loss reductions of the trainers (srgan/trainer.py:163-164,384,446-457; srgan/loss.py:52;
// esrgan/trainer.py:163-164,451-469).  Mean reductions are two-stage (<=1024 per-block partial
// sums in a caller workspace, then one block in fp64), so the loss value is reproducible.
#include "srx_common.h"

namespace {

enum { L_MSE = 0, L_L1 = 1, L_BCE = 2, L_BCE_LOGITS = 3, L_MEAN = 4 };

__device__ __forceinline__ float loss_term(int kind, float a, float b, float target) {
  if (kind == L_MSE) { const float d = a - b; return d * d; }
  if (kind == L_L1) return fabsf(a - b);
  if (kind == L_BCE) {
    // torch.nn.BCELoss clamps both log terms at -100
    const float lp = fmaxf(logf(a), -100.f), lq = fmaxf(log1pf(-a), -100.f);
    return -(target * lp + (1.f - target) * lq);
  }
  if (kind == L_MEAN) return a;
  // BCEWithLogits: (1-t)*x + log(1+exp(-|x|)) + max(-x,0)   (a = logit - shift)
  return (1.f - target) * a + fmaxf(-a, 0.f) + log1pf(expf(-fabsf(a)));
}

__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ shift, float target,
                                                           float* __restrict__ partial, int64_t n) {
  __shared__ float red[4];
  const float sh = shift ? shift[0] : 0.f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float av = a[i] - sh;
    const float bv = b ? b[i] : 0.f;
    acc += loss_term(kind, av, bv, target);
  }
  acc = srx_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, int nb, double inv_n,
                                                         float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] * inv_n);
}

__global__ void loss_bwd_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b,
                                const float* __restrict__ shift, float target, const float* __restrict__ gscale,
                                float* __restrict__ da, float* __restrict__ db, int64_t n, float inv_n) {
  const float g = gscale[0] * inv_n;
  const float sh = shift ? shift[0] : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float av = a[i] - sh;
    float d;
    if (kind == L_MSE) d = 2.f * (av - b[i]);
    else if (kind == L_L1) { const float t = av - b[i]; d = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f); }
    else if (kind == L_BCE) {
      // torch: grad * (p - t) / max((1-p)*p, 1e-12)
      d = (av - target) / fmaxf((1.f - av) * av, 1e-12f);
    } else if (kind == L_MEAN) {
      d = 1.f;
    } else {
      d = 1.f / (1.f + expf(-av)) - target;  // sigmoid(x) - t
    }
    da[i] = g * d;
    if (db) db[i] = -g * d;
  }
}

unsigned red_grid(int64_t n) {
  int64_t b = srx_cdiv(n, 1024);
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// count: the mean's divisor (0: n, every element counts)
int loss_fwd(int kind, const float* a, const float* b, const float* shift, float target, float* loss, int64_t n,
             float* ws, void* stream, const char* who, int64_t count = 0) {
  SRX_REQUIRE(a && loss && ws && n > 0 && count >= 0, "%s: bad argument", who);
  const unsigned nb = red_grid(n);
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(nb), dim3(256), 0, st, kind, a, b, shift, target, ws, n);
  SRX_CHECK_LAUNCH("loss_partial_kernel");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, st, ws, (int)nb, 1.0 / (double)(count ? count : n), loss);
  SRX_CHECK_LAUNCH("loss_final_kernel");
  return SRX_OK;
}

int loss_bwd(int kind, const float* a, const float* b, const float* shift, float target, const float* gscale, float* da,
             float* db, int64_t n, void* stream, const char* who, int64_t count = 0) {
  SRX_REQUIRE(a && gscale && da && n > 0 && count >= 0, "%s: bad argument", who);
  int64_t blocks = srx_cdiv(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, srx_stream(stream), kind, a, b, shift,
                     target, gscale, da, db, n, 1.0f / (float)(count ? count : n));
  SRX_CHECK_LAUNCH("loss_bwd_kernel");
  return SRX_OK;
}

}  // namespace

extern "C" int srx_mse_fwd(const float* a, const float* b, float* loss, int64_t n, float* ws, void* stream) {
  SRX_REQUIRE(b, "mse_fwd: bad argument");
  return loss_fwd(L_MSE, a, b, nullptr, 0.f, loss, n, ws, stream, "mse_fwd");
}
extern "C" int srx_l1_fwd(const float* a, const float* b, float* loss, int64_t n, float* ws, void* stream) {
  SRX_REQUIRE(b, "l1_fwd: bad argument");
  return loss_fwd(L_L1, a, b, nullptr, 0.f, loss, n, ws, stream, "l1_fwd");
}
// the same with an explicit divisor: images stored with a zero padding channel (NHWC, 3 of 4 channels real) take the
// mean over the real elements only (count = 3/4 n), which is what the reference's NCHW tensors give
extern "C" int srx_l1_fwd_count(const float* a, const float* b, float* loss, int64_t n, int64_t count, float* ws, void* stream) {
  SRX_REQUIRE(b && count > 0, "l1_fwd_count: bad argument");
  return loss_fwd(L_L1, a, b, nullptr, 0.f, loss, n, ws, stream, "l1_fwd_count", count);
}
extern "C" int srx_l1_bwd_count(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n,
                                int64_t count, void* stream) {
  SRX_REQUIRE(b && count > 0, "l1_bwd_count: bad argument");
  return loss_bwd(L_L1, a, b, nullptr, 0.f, gscale, da, db, n, stream, "l1_bwd_count", count);
}
extern "C" int srx_mse_bwd(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n,
                           void* stream) {
  SRX_REQUIRE(b, "mse_bwd: bad argument");
  return loss_bwd(L_MSE, a, b, nullptr, 0.f, gscale, da, db, n, stream, "mse_bwd");
}
extern "C" int srx_l1_bwd(const float* a, const float* b, const float* gscale, float* da, float* db, int64_t n,
                          void* stream) {
  SRX_REQUIRE(b, "l1_bwd: bad argument");
  return loss_bwd(L_L1, a, b, nullptr, 0.f, gscale, da, db, n, stream, "l1_bwd");
}
extern "C" int srx_bce_fwd(const float* p, float target, float* loss, int64_t n, float* ws, void* stream) {
  return loss_fwd(L_BCE, p, nullptr, nullptr, target, loss, n, ws, stream, "bce_fwd");
}
extern "C" int srx_bce_bwd(const float* p, float target, const float* gscale, float* dp, int64_t n, void* stream) {
  return loss_bwd(L_BCE, p, nullptr, nullptr, target, gscale, dp, nullptr, n, stream, "bce_bwd");
}
extern "C" int srx_bce_logits_fwd(const float* x, const float* shift, float target, float* loss, int64_t n, float* ws,
                                  void* stream) {
  return loss_fwd(L_BCE_LOGITS, x, nullptr, shift, target, loss, n, ws, stream, "bce_logits_fwd");
}
extern "C" int srx_bce_logits_bwd(const float* x, const float* shift, float target, const float* gscale, float* dx,
                                  int64_t n, void* stream) {
  return loss_bwd(L_BCE_LOGITS, x, nullptr, shift, target, gscale, dx, nullptr, n, stream, "bce_logits_bwd");
}
// torch.mean over all elements (relativistic average terms, esrgan/trainer.py:451-452,468)
extern "C" int srx_mean_fwd(const float* x, float* out, int64_t n, float* ws, void* stream) {
  return loss_fwd(L_MEAN, x, nullptr, nullptr, 0.f, out, n, ws, stream, "mean_fwd");
}
extern "C" int srx_mean_bwd(const float* x, const float* gscale, float* dx, int64_t n, void* stream) {
  return loss_bwd(L_MEAN, x, nullptr, nullptr, 0.f, gscale, dx, nullptr, n, stream, "mean_bwd");
}
