// torch.optim.Adam (lr 1e-4, betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) as the
// reference configures it three times (srgan/trainer.py:171-185), run over ONE flat parameter
// buffer per model: a single HBM-bound streaming pass (p, g, m, v read; p, m, v written) instead
// of ~150 per-tensor launches.  The step counter and the learning rate live on the device so a
// captured hipGraph of the train step stays valid across steps and StepLR updates.
#include "srx_common.h"

namespace {

__global__ void adam_tick_kernel(int64_t* step) { *step += 1; }

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const float* __restrict__ lr_ptr, float beta1, float beta2,
                                                   float eps, float gscale, const int64_t* __restrict__ step_ptr) {
  // same operation order as torch/optim/adam.py (_single_tensor_adam):
  //   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
  //   step_size = lr/(1-b1^t) ; denom = sqrt(v)/sqrt(1-b2^t) + eps ; p -= step_size * m/denom
  const double t = (double)(*step_ptr);
  const float bc1 = (float)(1.0 - pow((double)beta1, t));
  const float bc2 = (float)(1.0 - pow((double)beta2, t));
  const float step_size = lr_ptr[0] / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + i * 4);
    f32x4 gv = *reinterpret_cast<const f32x4*>(g + i * 4);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gv[e] * gscale;
      mv[e] = beta1 * mv[e] + (1.f - beta1) * gg;
      vv[e] = beta2 * vv[e] + (1.f - beta2) * gg * gg;
      const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
      pv[e] -= step_size * (mv[e] / denom);
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pv;
    *reinterpret_cast<f32x4*>(m + i * 4) = mv;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float gg = g[i] * gscale;
    const float mm = beta1 * m[i] + (1.f - beta1) * gg;
    const float vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

}  // namespace

extern "C" int srx_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const float* lr, float beta1,
                             float beta2, float eps, float grad_scale, int64_t* step, void* stream) {
  SRX_REQUIRE(p && g && m && v && lr && step && n > 0, "adam_step: bad argument");
  SRX_REQUIRE(((uintptr_t)p % 16 == 0) && ((uintptr_t)g % 16 == 0) && ((uintptr_t)m % 16 == 0) &&
                  ((uintptr_t)v % 16 == 0),
              "adam_step: buffers must be 16-byte aligned");
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, st, step);
  SRX_CHECK_LAUNCH("adam_tick_kernel");
  int64_t blocks = srx_cdiv(n / 4, 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps,
                     grad_scale, step);
  SRX_CHECK_LAUNCH("adam_kernel");
  return SRX_OK;
}
