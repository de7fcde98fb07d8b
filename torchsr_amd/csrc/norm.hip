// Training / eval BatchNorm2d on NHWC fp32 activations, fused with the activation
// (PReLU / LeakyReLU) and the residual add that follow it in the reference graphs
// (srgan/residual.py:64-68,86-91; srgan/generator.py:48-49,77-78;
//  srgan/discriminator.py:35-61).  All kernels are HBM-bound streaming passes with
// 16-byte accesses; reductions are two-stage (per row block, then a tiny finalize)
// so results are bitwise reproducible -- no float atomics.
#include "srx_common.h"
#include <cstdlib>

namespace {

// rows of the activation matrix summed by one workgroup: small tensors get small blocks so that the
// reduction still fills the chip (a 16x24x24x64 tensor is only 9216 rows)
__host__ __device__ inline int rows_per_block(int64_t M) { return M >= 131072 ? 512 : (M >= 32768 ? 128 : 32); }

__device__ __forceinline__ float act_fwd(float z, int act, float slope) {
  if (act == SRX_ACT_NONE) return z;
  if (act == SRX_ACT_RELU) return fmaxf(z, 0.f);
  return z > 0.f ? z : z * slope;  // LRELU / PRELU
}
__device__ __forceinline__ float act_grad(float z, int act, float slope) {
  if (act == SRX_ACT_NONE) return 1.f;
  if (act == SRX_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  return z > 0.f ? 1.f : slope;
}

// per row block: partial[b][c][0] = sum, partial[b][c][1] = sum of squares
__global__ __launch_bounds__(256) void bn_partial_stats_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                               int64_t M, int C) {
  __shared__ f32x4 red[2][256];
  const int cq = C / 4, nrl = 256 / cq;
  const int tid = threadIdx.x;
  const int q = tid % cq, rl = tid / cq;
  const int64_t rbeg = (int64_t)blockIdx.x * rows_per_block(M);
  const int64_t rend = min(M, rbeg + rows_per_block(M));
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl) {
    for (int64_t r = rbeg + rl; r < rend; r += nrl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * C + q * 4);
      s += v;
      s2 += v * v;
    }
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  __syncthreads();
  if (tid < cq) {
    f32x4 t = red[0][tid], t2 = red[1][tid];
    for (int k = 1; k < nrl; ++k) { t += red[0][tid + k * cq]; t2 += red[1][tid + k * cq]; }
    float* o = part + ((size_t)blockIdx.x * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = t[e]; o[2 * e + 1] = t2[e]; }
  }
}

__global__ void bn_finalize_kernel(const float* __restrict__ part, int rows, int64_t M, int C, float eps, float mom,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                   float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  // one wave per channel: lanes stride over the partial rows, fp64 butterfly reduction
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c == 0 && lane == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double s = 0.0, s2 = 0.0;
  for (int r = lane; r < rows; r += 64) {
    const float2 v = *reinterpret_cast<const float2*>(part + ((size_t)r * C + c) * 2);
    s += (double)v.x;
    s2 += (double)v.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane != 0) return;
  const double mu = s / (double)M;
  double var = s2 / (double)M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    rmean[c] = (float)((1.0 - mom) * (double)rmean[c] + mom * mu);
    rvar[c] = (float)((1.0 - mom) * (double)rvar[c] + mom * unbiased);
  }
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, int C, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = rmean[c];
  invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ res,
                                                         float* __restrict__ out, int64_t n4, int cq, int act,
                                                         float slope, const float* __restrict__ prelu) {
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_fwd((v[e] - mu[e]) * (is[e] * g[e]) + b[e], act, slope);
    if (res) o += *reinterpret_cast<const f32x4*>(res + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

// backward pass 1: per row block partial sums of dz, dz*xhat (per channel) and of the PReLU slope gradient
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ ws,
                                                            int64_t M, int C, int act, float slope,
                                                            const float* __restrict__ prelu, int rpb) {
  __shared__ f32x4 red[2][256];
  __shared__ float redp[256];
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  const int cq = C / 4, nrl = 256 / cq;
  const int tid = threadIdx.x;
  const int q = tid % cq, rl = tid / cq;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  const int64_t rend = min(M, rbeg + rpb);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  float sp = 0.f;
  if (rl < nrl) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + q * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + q * 4);
    for (int64_t r = rbeg + rl; r < rend; r += nrl) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * C + q * 4);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dout + r * C + q * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (v[e] - mu[e]) * is[e];
        const float z = xh * g[e] + b[e];
        const float dz = d[e] * act_grad(z, act, slope);
        s[e] += dz;
        s2[e] += dz * xh;
        if (act == SRX_ACT_PRELU && !(z > 0.f)) sp += d[e] * z;
      }
    }
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  redp[tid] = sp;
  __syncthreads();
  float* o = ws + (size_t)blockIdx.x * (2 * C + 4);
  if (tid < cq) {
    f32x4 t = red[0][tid], t2 = red[1][tid];
    for (int k = 1; k < nrl; ++k) { t += red[0][tid + k * cq]; t2 += red[1][tid + k * cq]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[tid * 4 + e] = t[e]; o[C + tid * 4 + e] = t2[e]; }
  }
  if (tid == 0) {
    float t = 0.f;
    for (int k = 0; k < 256; ++k) t += redp[k];
    o[2 * C] = t;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int rows, int C,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dprelu) {
  // one wave per column: lanes stride over the partial rows (a few hundred at most), fp64 butterfly
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c > 2 * C) return;
  double s = 0.0;
  for (int r = lane; r < rows; r += 64) s += (double)ws[(size_t)r * (2 * C + 4) + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    sums[c] = (float)s;
    // optional direct accumulation into the parameters' .grad buffers (saves three tiny adds per layer)
    if (c < C) { if (dbeta) dbeta[c] += (float)s; }
    else if (c < 2 * C) { if (dgamma) dgamma[c - C] += (float)s; }
    else if (dprelu) dprelu[0] += (float)s;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ sums, float* __restrict__ dy,
                                                           int64_t n4, int C, float invM, int act, float slope,
                                                           const float* __restrict__ prelu, int training) {
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  const int cq = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 sd = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
    if (training) {
      sd = *reinterpret_cast<const f32x4*>(sums + c);
      sx = *reinterpret_cast<const f32x4*>(sums + C + c);
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu[e]) * is[e];
      const float z = xh * g[e] + b[e];
      const float dz = d[e] * act_grad(z, act, slope);
      o[e] = g[e] * is[e] * (dz - sd[e] * invM - xh * sx[e] * invM);
    }
    *reinterpret_cast<f32x4*>(dy + i * 4) = o;
  }
}

// ---- small tensors: finalize fused into the apply pass ---------------------------------------
// Each workgroup re-reduces the (small) partial table into LDS, then streams its share of the
// tensor; workgroup 0 also publishes the statistics.  Saves one kernel boundary per BatchNorm in
// each direction, which is what the 16x24x24x64 generator layers are bound by.
__global__ __launch_bounds__(256) void bn_train_fwd_fused_kernel(
    const float* __restrict__ y, const float* __restrict__ part, int rows, int64_t M, int C, float eps, float mom,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ res,
    float* __restrict__ out, int act, float slope, const float* __restrict__ prelu, float* __restrict__ save_mean,
    float* __restrict__ save_invstd, float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  extern __shared__ float sm[];  // [C] scale, [C] shift
  float* s_scale = sm;
  float* s_shift = sm + C;
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  // Table [rows][C][2] read as float4 = (sum, sumsq) of two channels: thread t owns channel pair t % (C/2)
  // and the rows t / (C/2) + k * RL (C/2 divides 256): independent, fully coalesced loads, fp64 sums.
  __shared__ double red[256][4];
  const int cp2 = C / 2, RL = 256 / cp2;
  const int cp = threadIdx.x % cp2, rl = threadIdx.x / cp2;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const f32x4* tab = reinterpret_cast<const f32x4*>(part);
#pragma unroll 8
  for (int r = rl; r < rows; r += RL) {
    const f32x4 v = tab[(size_t)r * cp2 + cp];
    a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
  }
  red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1; red[threadIdx.x][2] = a2; red[threadIdx.x][3] = a3;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k) {
      a0 += red[threadIdx.x + k * cp2][0]; a1 += red[threadIdx.x + k * cp2][1];
      a2 += red[threadIdx.x + k * cp2][2]; a3 += red[threadIdx.x + k * cp2][3];
    }
    const double ss[2] = {a0, a2}, s2s[2] = {a1, a3};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int c = 2 * cp + e;
      const double mu = ss[e] / (double)M;
      double var = s2s[e] / (double)M - mu * mu;
      if (var < 0.0) var = 0.0;
      const float is = (float)(1.0 / sqrt(var + (double)eps));
      const float sc = is * gamma[c];
      s_scale[c] = sc;
      s_shift[c] = beta[c] - (float)mu * sc;
      if (blockIdx.x == 0) {
        save_mean[c] = (float)mu;
        save_invstd[c] = is;
        if (rmean) {
          const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
          rmean[c] = (float)((1.0 - mom) * (double)rmean[c] + mom * mu);
          rvar[c] = (float)((1.0 - mom) * (double)rvar[c] + mom * unbiased);
        }
      }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  const int cq = C / 4;
  const int64_t n4 = M * cq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = act_fwd(v[e] * s_scale[c + e] + s_shift[c + e], act, slope);
    if (res) o += *reinterpret_cast<const f32x4*>(res + i * 4);
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_fused_kernel(
    const float* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ ws, int rows, float* __restrict__ sums, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dprelu, float* __restrict__ dy, int64_t M, int C, int act,
    float slope, const float* __restrict__ prelu, int want_dy) {
  extern __shared__ float sm[];  // [2C+4] reduced sums
  __shared__ double red[256][4];
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  // Table [rows][2C+4] read as float4 quads: thread t owns quad t % QN and the rows t / QN + k * RL.
  const int QN = (2 * C + 4) / 4, RL = 256 / QN;
  const int q = threadIdx.x % QN, rl = threadIdx.x / QN;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const f32x4* tab = reinterpret_cast<const f32x4*>(ws);
  if (rl < RL) {
#pragma unroll 8
    for (int r = rl; r < rows; r += RL) {
      const f32x4 v = tab[(size_t)r * QN + q];
      a0 += (double)v[0]; a1 += (double)v[1]; a2 += (double)v[2]; a3 += (double)v[3];
    }
  }
  red[threadIdx.x][0] = a0; red[threadIdx.x][1] = a1; red[threadIdx.x][2] = a2; red[threadIdx.x][3] = a3;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k) {
      a0 += red[threadIdx.x + k * QN][0]; a1 += red[threadIdx.x + k * QN][1];
      a2 += red[threadIdx.x + k * QN][2]; a3 += red[threadIdx.x + k * QN][3];
    }
    const double t4[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * q + e;
      const float t = (float)t4[e];
      sm[c] = t;
      if (blockIdx.x == 0 && c <= 2 * C) {
        sums[c] = t;
        if (c < C) { if (dbeta) dbeta[c] += t; }
        else if (c < 2 * C) { if (dgamma) dgamma[c - C] += t; }
        else if (dprelu) dprelu[0] += t;
      }
    }
  }
  __syncthreads();
  if (!want_dy) return;
  const int cq = C / 4;
  const int64_t n4 = M * cq;
  const float invM = 1.0f / (float)M;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (v[e] - mu[e]) * is[e];
      const float z = xh * g[e] + b[e];
      const float dz = d[e] * act_grad(z, act, slope);
      o[e] = g[e] * is[e] * (dz - sm[c + e] * invM - xh * sm[C + c + e] * invM);
    }
    *reinterpret_cast<f32x4*>(dy + i * 4) = o;
  }
}

// Measured on MI355X, in a hipGraph (tools/bench_bn.py), fused vs finalize + apply as two kernels:
//   16x24x24x64 (generator):     forward 8.3 vs 7.2 us, backward 14.7 vs 12.1 us  -> slower
//   16x48x48x128 (discriminator): forward 4.1 vs 7.0 us, backward 57 vs 32 us     -> mixed
// Every workgroup of the apply pass re-reducing the partial table (wide independent loads, fp64 sums) costs
// more than the kernel boundary + one-workgroup finalize kernel it saves, and fewer / fatter reduction
// workgroups starve the backward reduce.  The fused forms stay off; SRX_BN_FUSE_MAX=<floats> enables them
// for tables up to that size (experiments).
static int64_t fuse_max_table() {
  static const char* e = getenv("SRX_BN_FUSE_MAX");
  return e ? atoll(e) : 0;
}
#define FUSE_MAX_TABLE fuse_max_table()

// rows per workgroup of the backward reduction when its table is re-reduced by the fused apply kernel:
// about 64 partial rows whatever the tensor size (64 workgroups stream a 16x24x24x64 pair in ~1 us)
__host__ inline int fused_bwd_rows_per_block(int64_t M) {
  int64_t r = srx_roundup(srx_cdiv(M, 64), 4);
  return (int)(r < 32 ? 32 : r);
}

int check_c(int C, const char* who) {
  SRX_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, "%s: C must be a multiple of 4 in [4,1024]", who);
  return SRX_OK;
}

unsigned stream_grid(int64_t n4) {
  int64_t b = srx_cdiv(n4, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int srx_bn_stat_rows(int64_t M) { return (int)srx_cdiv(M, rows_per_block(M)); }

extern "C" int srx_bn_partial_stats(const float* y, float* partials, int64_t M, int C, void* stream) {
  if (int rc = check_c(C, "bn_partial_stats")) return rc;
  SRX_REQUIRE(y && partials && M > 0, "bn_partial_stats: bad argument");
  hipLaunchKernelGGL(bn_partial_stats_kernel, dim3((unsigned)srx_bn_stat_rows(M)), dim3(256), 0, srx_stream(stream), y,
                     partials, M, C);
  SRX_CHECK_LAUNCH("bn_partial_stats_kernel");
  return SRX_OK;
}

extern "C" int srx_bn_finalize(const float* partials, int rows, int64_t M, int C, float eps, float momentum,
                               float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                               int64_t* nbt, void* stream) {
  SRX_REQUIRE(partials && save_mean && save_invstd && rows > 0 && M > 0 && C > 0, "bn_finalize: bad argument");
  SRX_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)srx_cdiv(C, 4)), dim3(256), 0, srx_stream(stream), partials,
                     rows, M, C, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt);
  SRX_CHECK_LAUNCH("bn_finalize_kernel");
  return SRX_OK;
}

extern "C" int srx_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps,
                                 float* save_mean, float* save_invstd, void* stream) {
  SRX_REQUIRE(running_mean && running_var && save_mean && save_invstd && C > 0, "bn_eval_stats: bad argument");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((unsigned)srx_cdiv(C, 64)), dim3(64), 0, srx_stream(stream),
                     running_mean, running_var, C, eps, save_mean, save_invstd);
  SRX_CHECK_LAUNCH("bn_eval_stats_kernel");
  return SRX_OK;
}

extern "C" int srx_bn_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* residual, float* out, int64_t M, int C, int act,
                              float slope, const float* prelu, void* stream) {
  if (int rc = check_c(C, "bn_act_fwd")) return rc;
  SRX_REQUIRE(y && mean && invstd && gamma && beta && out && M > 0, "bn_act_fwd: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_fwd: PReLU needs its slope pointer");
  const int64_t n4 = M * C / 4;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_grid(n4)), dim3(256), 0, srx_stream(stream), y, mean, invstd, gamma,
                     beta, residual, out, n4, C / 4, act, slope, prelu);
  SRX_CHECK_LAUNCH("bn_act_fwd_kernel");
  return SRX_OK;
}

extern "C" size_t srx_bn_bwd_ws_floats(int64_t M, int C) { return (size_t)srx_bn_stat_rows(M) * (2 * C + 4); }

extern "C" int srx_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* sums, int64_t M, int C, int act,
                                     float slope, const float* prelu, float* dgamma_acc, float* dbeta_acc,
                                     float* dprelu_acc, float* ws, size_t ws_floats, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd_reduce")) return rc;
  SRX_REQUIRE(dout && y && mean && invstd && gamma && beta && sums && ws && M > 0, "bn_act_bwd_reduce: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd_reduce: PReLU needs its slope pointer");
  if (ws_floats < srx_bn_bwd_ws_floats(M, C)) SRX_FAIL(SRX_E_WORKSPACE, "bn_act_bwd_reduce: workspace too small");
  const int rows = srx_bn_stat_rows(M);
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, dout, y, mean, invstd, gamma, beta,
                     ws, M, C, act, slope, prelu, rows_per_block(M));
  SRX_CHECK_LAUNCH("bn_bwd_reduce_kernel");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)srx_cdiv(2 * C + 1, 4)), dim3(256), 0, st, ws, rows, C,
                     sums, dgamma_acc, dbeta_acc, dprelu_acc);
  SRX_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  return SRX_OK;
}

extern "C" int srx_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const float* sums, float* dy, int64_t M,
                                    int C, int act, float slope, const float* prelu, int training, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd_apply")) return rc;
  SRX_REQUIRE(dout && y && mean && invstd && gamma && beta && dy && M > 0, "bn_act_bwd_apply: bad argument");
  SRX_REQUIRE(!training || sums, "bn_act_bwd_apply: training mode needs the reduced sums");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd_apply: PReLU needs its slope pointer");
  const int64_t n4 = M * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, srx_stream(stream), dout, y, mean,
                     invstd, gamma, beta, sums, dy, n4, C, 1.0f / (float)M, act, slope, prelu, training);
  SRX_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return SRX_OK;
}

// Training-mode BatchNorm forward in one call: statistics from the conv's partial table, running
// statistics update, normalise + activation (+ residual).  Fuses finalize into the apply pass when
// the partial table is small; otherwise two kernels as before.
extern "C" int srx_bn_train_fwd(const float* y, const float* partials, int rows, int64_t M, int C, float eps,
                                float momentum, const float* gamma, const float* beta, const float* residual,
                                float* out, int act, float slope, const float* prelu, float* save_mean,
                                float* save_invstd, float* running_mean, float* running_var, int64_t* nbt,
                                void* stream) {
  if (int rc = check_c(C, "bn_train_fwd")) return rc;
  SRX_REQUIRE(y && partials && gamma && beta && out && save_mean && save_invstd && rows > 0 && M > 0,
              "bn_train_fwd: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_train_fwd: PReLU needs its slope pointer");
  SRX_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats must come in pairs");
  if ((int64_t)rows * C * 2 > FUSE_MAX_TABLE || C < 8 || C > 512 || 256 % (C / 2) != 0) {
    if (int rc = srx_bn_finalize(partials, rows, M, C, eps, momentum, save_mean, save_invstd, running_mean,
                                 running_var, nbt, stream))
      return rc;
    return srx_bn_act_fwd(y, save_mean, save_invstd, gamma, beta, residual, out, M, C, act, slope, prelu, stream);
  }
  const int64_t n4 = M * C / 4;
  int64_t blocks = srx_cdiv(n4, 256 * 4);  // a few float4 per thread so the table re-reduction amortises
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bn_train_fwd_fused_kernel, dim3((unsigned)blocks), dim3(256), 2 * C * sizeof(float),
                     srx_stream(stream), y, partials, rows, M, C, eps, momentum, gamma, beta, residual, out, act, slope,
                     prelu, save_mean, save_invstd, running_mean, running_var, nbt);
  SRX_CHECK_LAUNCH("bn_train_fwd_fused_kernel");
  return SRX_OK;
}

// Backward of act(BN(y)) in one call (reduce -> [finalize + apply]); same outputs as
// srx_bn_act_bwd_reduce + srx_bn_act_bwd_apply.  dy may be NULL when only parameter gradients are needed.
extern "C" int srx_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float* sums, float* dy, int64_t M, int C,
                              int act, float slope, const float* prelu, int training, float* dgamma_acc,
                              float* dbeta_acc, float* dprelu_acc, float* ws, size_t ws_floats, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd")) return rc;
  SRX_REQUIRE(dout && y && mean && invstd && gamma && beta && sums && ws && M > 0, "bn_act_bwd: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd: PReLU needs its slope pointer");
  const int rpb = fused_bwd_rows_per_block(M);
  const int rows = (int)srx_cdiv(M, rpb);  // (never more than srx_bn_stat_rows(M): the workspace bound holds)
  if (!training || (int64_t)rows * (2 * C + 4) > FUSE_MAX_TABLE || (2 * C + 4) / 4 > 256 || rows > srx_bn_stat_rows(M)) {
    if (int rc = srx_bn_act_bwd_reduce(dout, y, mean, invstd, gamma, beta, sums, M, C, act, slope, prelu, dgamma_acc,
                                       dbeta_acc, dprelu_acc, ws, ws_floats, stream))
      return rc;
    if (!dy) return SRX_OK;
    return srx_bn_act_bwd_apply(dout, y, mean, invstd, gamma, beta, sums, dy, M, C, act, slope, prelu, training,
                                stream);
  }
  if (ws_floats < srx_bn_bwd_ws_floats(M, C)) SRX_FAIL(SRX_E_WORKSPACE, "bn_act_bwd: workspace too small");
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, dout, y, mean, invstd, gamma, beta,
                     ws, M, C, act, slope, prelu, rpb);
  SRX_CHECK_LAUNCH("bn_bwd_reduce_kernel");
  const int64_t n4 = M * C / 4;
  int64_t blocks = srx_cdiv(n4, 256 * 4);
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bn_bwd_apply_fused_kernel, dim3((unsigned)blocks), dim3(256), (2 * C + 4) * sizeof(float), st, dout,
                     y, mean, invstd, gamma, beta, ws, rows, sums, dgamma_acc, dbeta_acc, dprelu_acc, dy, M, C, act,
                     slope, prelu, dy ? 1 : 0);
  SRX_CHECK_LAUNCH("bn_bwd_apply_fused_kernel");
  return SRX_OK;
}
