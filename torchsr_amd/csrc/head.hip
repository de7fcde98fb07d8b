// The end of a discriminator pass as it sits in the GAN train steps: last Linear layer -> (Sigmoid) -> adversarial loss,
// forward in ONE launch and backward in ONE launch.
//
//   SRGAN  (srgan/discriminator.py:65-69, srgan/trainer.py:446-448,456-457): Linear(1024 -> 1), Sigmoid, BCELoss
//   ESRGAN (esrgan/discriminator.py:73-76, esrgan/trainer.py:451-453,468-469): Linear(100 -> 1), BCEWithLogitsLoss on
//          relativistic-average logits
//
// As separate operators this tail is ~25 launches of one workgroup each per discriminator update and ~15 per generator
// update (linear, sigmoid, two means, two losses, their sum, the backward of each, two bias column sums): 0.2 ms of a
// 8.9 ms SRGAN step spent on kernel boundaries around a few thousand FLOPs.  Here the forward kernel (one workgroup) computes
// the logits, the loss terms and the step's scalar loss; the backward kernel (one workgroup per 256 hidden units)
// recomputes the per-row logit gradients from the saved logits / probabilities and writes, in the same pass, the
// gradient of the hidden layer's PRE-activation (LeakyReLU backward folded in), of the last layer's weight and bias
// and of the hidden layer's bias.  Reductions run in a fixed order (fp64 row sums): results are bitwise reproducible.
#include "srx_common.h"

namespace {

constexpr int HEAD_MAX_B = 256;  // rows (one thread per row in the loss phases)

struct HeadArgs {
  int mode, B, J, n_first;
  float slope, adv_weight;
};

// torch.nn.BCELoss term (both logs clamped at -100) and torch.nn.BCEWithLogitsLoss term -- the expressions of loss.hip
__device__ __forceinline__ float bce_term(float p, float t) {
  const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(log1pf(-p), -100.f);
  return -(t * lp + (1.f - t) * lq);
}
__device__ __forceinline__ float bcel_term(float a, float t) {
  return (1.f - t) * a + fmaxf(-a, 0.f) + log1pf(expf(-fabsf(a)));
}

// sum of v over rows [lo, hi) in row order, in fp64, by thread 0 (every thread gets the result)
__device__ __forceinline__ double row_sum(const double* sh, int lo, int hi, double* out) {
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int b = lo; b < hi; ++b) s += sh[b];
    *out = s;
  }
  __syncthreads();
  return *out;
}

__global__ __launch_bounds__(256) void gan_head_fwd_kernel(const HeadArgs h, const float* __restrict__ hidden,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ shift, const float* __restrict__ addend,
                                                           float* __restrict__ zp, float* __restrict__ out,
                                                           float* __restrict__ loss) {
  __shared__ float sz[HEAD_MAX_B];
  __shared__ double sv[HEAD_MAX_B];
  __shared__ double stot;
  const int tid = threadIdx.x;
  const float bias = b2 ? b2[0] : 0.f;
  // logits: eight lanes per row, 32 rows per pass; a lane's hidden units are j = 8 i + part (coalesced 32-byte runs per row) in
  // groups of eight independent loads -- one lane per row walked the row as a chain of dependent round trips (33 us for
  // 32 x 1024 values; measured round 5) -- then a fixed butterfly over the eight lanes
  {
    const int part = tid & 7, rsub = tid >> 3;
    for (int b0 = 0; b0 < h.B; b0 += 32) {
      const int b = b0 + rsub;
      const float* hr = hidden + (size_t)min(b, h.B - 1) * h.J;
      float s = 0.f;
      for (int j0 = part; j0 < h.J; j0 += 64) {
        float hv[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int j = min(j0 + 8 * u, h.J - 1);
          hv[u] = hr[j];
          wv[u] = w2[j];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (j0 + 8 * u < h.J) s += hv[u] * wv[u];
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if (part == 0 && b < h.B) sz[b] = s + bias;
    }
  }
  __syncthreads();
  const int B = h.B, n = h.n_first;
  const bool row = tid < B;
  const float z = row ? sz[tid] : 0.f;
  if (h.mode == SRX_HEAD_SRGAN_D || h.mode == SRX_HEAD_SRGAN_G) {
    // Sigmoid + BCELoss: rows [0, n) against 1 and rows [n, B) against 0 (discriminator update), or all rows against 1
    const float p = 1.0f / (1.0f + expf(-z));
    if (row) zp[tid] = p;
    const float t = (h.mode == SRX_HEAD_SRGAN_G || tid < n) ? 1.f : 0.f;
    sv[tid] = row ? (double)bce_term(p, t) : 0.0;
    if (h.mode == SRX_HEAD_SRGAN_D) {
      const float a = (float)(row_sum(sv, 0, n, &stot) * (1.0 / (double)n));
      const float c = (float)(row_sum(sv, n, B, &stot) * (1.0 / (double)(B - n)));
      if (tid == 0) { out[0] = a + c; out[1] = a; out[2] = c; out[3] = 0.f; if (loss) loss[0] = out[0]; }
    } else {
      const float adv = (float)(row_sum(sv, 0, B, &stot) * (1.0 / (double)B));
      if (tid == 0) { out[0] = (addend ? addend[0] : 0.f) + h.adv_weight * adv; out[1] = adv; out[2] = 0.f; out[3] = 0.f; if (loss) loss[0] = out[0]; }
    }
    return;
  }
  if (row) zp[tid] = z;
  if (h.mode == SRX_HEAD_ESRGAN_D) {
    // relativistic average: BCEWithLogits(real - mean(fake), 1), BCEWithLogits(fake - mean(real), 0), half their sum
    sv[tid] = row ? (double)z : 0.0;
    const float mr = (float)(row_sum(sv, 0, n, &stot) * (1.0 / (double)n));
    const float mf = (float)(row_sum(sv, n, B, &stot) * (1.0 / (double)(B - n)));
    __syncthreads();
    sv[tid] = row ? (double)(tid < n ? bcel_term(z - mf, 1.f) : bcel_term(z - mr, 0.f)) : 0.0;
    const float a = (float)(row_sum(sv, 0, n, &stot) * (1.0 / (double)n));
    const float c = (float)(row_sum(sv, n, B, &stot) * (1.0 / (double)(B - n)));
    if (tid == 0) { out[0] = 0.5f * a + 0.5f * c; out[1] = a; out[2] = c; out[3] = 0.f; out[4] = mr; out[5] = mf; if (loss) loss[0] = out[0]; }
    return;
  }
  // SRX_HEAD_ESRGAN_G: BCEWithLogits(fake - shift, 1), shift = mean(D(real)) computed by the caller without a graph
  const float sh = shift[0];
  sv[tid] = row ? (double)bcel_term(z - sh, 1.f) : 0.0;
  const float adv = (float)(row_sum(sv, 0, B, &stot) * (1.0 / (double)B));
  if (tid == 0) { out[0] = (addend ? addend[0] : 0.f) + h.adv_weight * adv; out[1] = adv; out[2] = 0.f; out[3] = 0.f; if (loss) loss[0] = out[0]; }
}

// grid: ceil(J / 256) workgroups; thread = one hidden unit j.  Every workgroup first forms the B logit gradients (a few
// hundred FLOPs, cheaper than a launch that would hand them over).
__global__ __launch_bounds__(256) void gan_head_bwd_kernel(const HeadArgs h, const float* __restrict__ hidden,
                                                           const float* __restrict__ w2, const float* __restrict__ zp,
                                                           const float* __restrict__ out, const float* __restrict__ shift,
                                                           const float* __restrict__ g, float* __restrict__ dpre,
                                                           float* __restrict__ dw2, float* __restrict__ db2,
                                                           float* __restrict__ db1, int accumulate) {
  __shared__ float sdz[HEAD_MAX_B];
  __shared__ double sv[HEAD_MAX_B];
  __shared__ double stot;
  const int tid = threadIdx.x;
  const int B = h.B, n = h.n_first;
  const bool row = tid < B;
  const float gs = g[0];
  const float v = row ? zp[tid] : 0.f;
  float dz = 0.f;
  if (h.mode == SRX_HEAD_SRGAN_D || h.mode == SRX_HEAD_SRGAN_G) {
    // BCELoss backward (torch: grad * (p - t) / max((1 - p) p, 1e-12) / count), then the sigmoid's dy * p * (1 - p)
    const bool gen = h.mode == SRX_HEAD_SRGAN_G;
    const float t = (gen || tid < n) ? 1.f : 0.f;
    const float cnt = gen ? (float)B : (tid < n ? (float)n : (float)(B - n));
    const float gl = gen ? h.adv_weight * gs : gs;
    const float dp = (gl * (1.0f / cnt)) * ((v - t) / fmaxf((1.f - v) * v, 1e-12f));
    dz = dp * v * (1.0f - v);
  } else if (h.mode == SRX_HEAD_ESRGAN_D) {
    const float mr = out[4], mf = out[5];
    const float gl = 0.5f * gs;
    // direct terms: sigmoid(a) - t on the shifted logits, over the count of their half
    float d = 0.f;
    if (row) d = tid < n ? (gl * (1.0f / (float)n)) * (1.0f / (1.0f + expf(-(v - mf))) - 1.f)
                         : (gl * (1.0f / (float)(B - n))) * (1.0f / (1.0f + expf(-(v - mr))));
    // through the means: mean(fake) shifts every real row (gradient -sum of the real rows' direct terms, spread over the
    // fake rows), and the other way round
    sv[tid] = row ? (double)d : 0.0;
    const float sum_r = (float)row_sum(sv, 0, n, &stot);
    const float sum_f = (float)row_sum(sv, n, B, &stot);
    dz = tid < n ? d + (-sum_f) * (1.0f / (float)n) : d + (-sum_r) * (1.0f / (float)(B - n));
  } else {
    const float sh = shift[0];
    dz = ((h.adv_weight * gs) * (1.0f / (float)B)) * (1.0f / (1.0f + expf(-(v - sh))) - 1.f);
  }
  if (row) sdz[tid] = dz;
  __syncthreads();
  const int j = blockIdx.x * 256 + tid;
  if (j < h.J) {
    const float w = w2[j];
    float aw = 0.f, ab = 0.f;
    for (int b0 = 0; b0 < B; b0 += 8) {  // eight rows per trip, loads first; additions in row order
      float hv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) hv[u] = hidden[(size_t)min(b0 + u, B - 1) * h.J + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (b0 + u >= B) continue;
        const float d = sdz[b0 + u];
        const float dh = d * w;
        const float dp = hv[u] > 0.f ? dh : dh * h.slope;  // LeakyReLU backward from the activation's output
        dpre[(size_t)(b0 + u) * h.J + j] = dp;
        aw += d * hv[u];
        ab += dp;
      }
    }
    if (dw2) dw2[j] = accumulate ? dw2[j] + aw : aw;
    if (db1) db1[j] = accumulate ? db1[j] + ab : ab;
  }
  if (db2 && blockIdx.x == 0) {
    sv[tid] = row ? (double)dz : 0.0;
    const float s = (float)row_sum(sv, 0, B, &stot);
    if (tid == 0) db2[0] = accumulate ? db2[0] + s : s;
  }
}

int check(const srx_gan_head_t* h, const char* who) {
  SRX_REQUIRE(h, "%s: null descriptor", who);
  SRX_REQUIRE(h->mode >= SRX_HEAD_SRGAN_D && h->mode <= SRX_HEAD_ESRGAN_G, "%s: bad mode %d", who, h->mode);
  SRX_REQUIRE(h->B > 0 && h->B <= HEAD_MAX_B && h->J > 0, "%s: 1..%d rows and at least one hidden unit", who, HEAD_MAX_B);
  if (h->mode == SRX_HEAD_SRGAN_D || h->mode == SRX_HEAD_ESRGAN_D)
    SRX_REQUIRE(h->n_first > 0 && h->n_first < h->B, "%s: the discriminator update needs rows of both calls (0 < n_first < B)", who);
  return SRX_OK;
}

}  // namespace

extern "C" int srx_gan_head_fwd(const srx_gan_head_t* h, const float* hidden, const float* w2, const float* b2,
                                const float* shift, const float* addend, float* zp, float* out, float* loss, void* stream) {
  if (int rc = check(h, "gan_head_fwd")) return rc;
  SRX_REQUIRE(hidden && w2 && zp && out, "gan_head_fwd: null pointer");
  SRX_REQUIRE(h->mode != SRX_HEAD_ESRGAN_G || shift, "gan_head_fwd: the relativistic generator term needs mean(D(real)) as shift");
  const HeadArgs a{h->mode, h->B, h->J, h->n_first, h->slope, h->adv_weight};
  hipLaunchKernelGGL(gan_head_fwd_kernel, dim3(1), dim3(256), 0, srx_stream(stream), a, hidden, w2, b2, shift, addend, zp, out, loss);
  SRX_CHECK_LAUNCH("gan_head_fwd_kernel");
  return SRX_OK;
}

extern "C" int srx_gan_head_bwd(const srx_gan_head_t* h, const float* hidden, const float* w2, const float* zp,
                                const float* out, const float* shift, const float* g, float* dpre, float* dw2, float* db2,
                                float* db1, int accumulate, void* stream) {
  if (int rc = check(h, "gan_head_bwd")) return rc;
  SRX_REQUIRE(hidden && w2 && zp && out && g && dpre, "gan_head_bwd: null pointer");
  SRX_REQUIRE(h->mode != SRX_HEAD_ESRGAN_G || shift, "gan_head_bwd: the relativistic generator term needs its shift");
  const HeadArgs a{h->mode, h->B, h->J, h->n_first, h->slope, h->adv_weight};
  hipLaunchKernelGGL(gan_head_bwd_kernel, dim3((unsigned)srx_cdiv(h->J, 256)), dim3(256), 0, srx_stream(stream), a, hidden, w2,
                     zp, out, shift, g, dpre, dw2, db2, db1, accumulate);
  SRX_CHECK_LAUNCH("gan_head_bwd_kernel");
  return SRX_OK;
}
