// Training / eval BatchNorm2d on NHWC fp32 activations, fused with the activation
// (PReLU / LeakyReLU) and the residual add that follow it in the reference graphs
// (srgan/residual.py:64-68,86-91; srgan/generator.py:48-49,77-78;
//  srgan/discriminator.py:35-61).  All kernels are HBM-bound streaming passes with
// 16-byte accesses; reductions are two-stage (per row block, then a tiny finalize)
// so results are bitwise reproducible -- no float atomics.
//
// Groups: the rows of the activation matrix may be `groups` equal, consecutive row ranges that are
// normalised INDEPENDENTLY -- several forward calls of the reference executed as one batch (the discriminator
// on the real and on the fake images, srgan/trainer.py:446-447: two calls of a training-mode BatchNorm2d,
// each with its own batch statistics and its own running-statistics update, in call order).  Statistics
// tensors are then [groups][C], the backward sums [groups][2C+4]; parameter gradients sum over the groups.
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
    for (int64_t r = rbeg + rl; r < rend; r += 4 * nrl) {  // four rows per trip, loads first (see bn_finalize_kernel)
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(y + min(r + u * nrl, rend - 1) * C + q * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * nrl < rend) { s += v[u]; s2 += v[u] * v[u]; }
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

__global__ void bn_finalize_kernel(const float* __restrict__ part, int rows, int64_t M, int C, int groups, float eps,
                                   float mom, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt) {
  // one wave per channel: lanes stride over the partial rows, fp64 butterfly reduction; the groups are taken
  // in order so that the running statistics see the same sequence of updates as separate forward calls
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c == 0 && lane == 0 && nbt) *nbt += groups;
  if (c >= C) return;
  const int rpg = rows / groups;
  const double Mg = (double)(M / groups);
  double rm = 0.0, rv = 0.0;
  if (rmean && lane == 0) { rm = (double)rmean[c]; rv = (double)rvar[c]; }
  for (int g = 0; g < groups; ++g) {
    double s = 0.0, s2 = 0.0;
    // four rows per lane and trip, all four loads issued before the first is added (same order of additions as one row
    // per trip): with a few hundred rows the kernel is one L2 round trip long instead of four (4.8 -> ~2.7 us, 47 launches a step)
    const int rend = (g + 1) * rpg;
    for (int r = g * rpg + lane; r < rend; r += 256) {
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(part + ((size_t)min(r + 64 * u, rend - 1) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 64 * u < rend) { s += (double)v[u].x; s2 += (double)v[u].y; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    if (lane != 0) continue;
    const double mu = s / Mg;
    double var = s2 / Mg - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[(size_t)g * C + c] = (float)mu;
    invstd[(size_t)g * C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {  // (rounded to fp32 after every update, as consecutive calls would)
      const double unbiased = Mg > 1.0 ? var * Mg / (Mg - 1.0) : var;
      rm = (double)(float)((1.0 - mom) * rm + mom * mu);
      rv = (double)(float)((1.0 - mom) * rv + mom * unbiased);
    }
  }
  if (rmean && lane == 0) { rmean[c] = (float)rm; rvar[c] = (float)rv; }
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
                                                         float* __restrict__ out, int64_t n4, int cq, int64_t n4_per_group,
                                                         int act, float slope, const float* __restrict__ prelu) {
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  const int C = cq * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const int gc = (int)(i / n4_per_group) * C + c;  // this row's group
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + gc);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + gc);
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
                                                            const float* __restrict__ prelu, int rpb,
                                                            int64_t rows_per_group) {
  __shared__ f32x4 red[2][256];
  __shared__ float redp[4];
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  const int cq = C / 4, nrl = 256 / cq;
  const int tid = threadIdx.x;
  const int q = tid % cq, rl = tid / cq;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  const int64_t rend = min(M, rbeg + rpb);
  const int goff = (int)(rbeg / rows_per_group) * C;  // a row block never straddles two groups (host-checked)
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  float sp = 0.f;
  if (rl < nrl) {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + goff + q * 4);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + goff + q * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + q * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + q * 4);
    for (int64_t r = rbeg + rl; r < rend; r += 4 * nrl) {  // four rows per trip, all eight loads before the first use
      f32x4 vv[4], dd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t rr = min(r + u * nrl, rend - 1);
        vv[u] = *reinterpret_cast<const f32x4*>(y + rr * C + q * 4);
        dd[u] = *reinterpret_cast<const f32x4*>(dout + rr * C + q * 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (r + u * nrl >= rend) continue;
        const f32x4 v = vv[u], d = dd[u];
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
  }
  red[0][tid] = s;
  red[1][tid] = s2;
  // PReLU slope gradient: a fixed butterfly inside each wave, then the four waves in order (one thread walking
  // 256 LDS words took as long as the streaming part of this kernel on the generator's 2.4 MB tensors)
  sp = srx_wave_sum(sp);
  if ((tid & 63) == 0) redp[tid >> 6] = sp;
  __syncthreads();
  float* o = ws + (size_t)blockIdx.x * (2 * C + 4);
  if (tid < cq) {
    f32x4 t = red[0][tid], t2 = red[1][tid];
    for (int k = 1; k < nrl; ++k) { t += red[0][tid + k * cq]; t2 += red[1][tid + k * cq]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[tid * 4 + e] = t[e]; o[C + tid * 4 + e] = t2[e]; }
  }
  if (tid == 0) o[2 * C] = (redp[0] + redp[1]) + (redp[2] + redp[3]);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ ws, int rows, int C, int groups,
                                                              float* __restrict__ sums, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ dprelu,
                                                              int prelu_cols) {
  // one wave per column: lanes stride over the partial rows (a few hundred at most), fp64 butterfly; per group
  // (each group's own sums feed its rows' input gradient), the parameter gradients take the total
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c > 2 * C) return;
  const int rpg = rows / groups;
  double total = 0.0;
  for (int g = 0; g < groups; ++g) {
    double s = 0.0;
    const int rend = (g + 1) * rpg;
    const bool two = c == 2 * C && prelu_cols == 2;  // (wave-uniform)
    for (int r = g * rpg + lane; r < rend; r += 256) {  // four independent loads per trip (see bn_finalize_kernel)
      float v[4], v2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t o = (size_t)min(r + 64 * u, rend - 1) * (2 * C + 4) + c;
        v[u] = ws[o];
        // (a table written by a conv epilogue holds the PReLU partial of each of its two column waves: columns 2C, 2C + 1)
        v2[u] = two ? ws[o + 1] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + 64 * u < rend) {
          s += (double)v[u];
          if (two) s += (double)v2[u];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) sums[(size_t)g * (2 * C + 4) + c] = (float)s;
    total += (double)(float)s;
  }
  if (lane == 0) {
    // optional direct accumulation into the parameters' .grad buffers (saves three tiny adds per layer)
    if (c < C) { if (dbeta) dbeta[c] += (float)total; }
    else if (c < 2 * C) { if (dgamma) dgamma[c - C] += (float)total; }
    else if (dprelu) dprelu[0] += (float)total;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ sums, float* __restrict__ dy,
                                                           int64_t n4, int C, int64_t n4_per_group, float invM, int act,
                                                           float slope, const float* __restrict__ prelu, int training) {
  if (act == SRX_ACT_PRELU) slope = prelu[0];
  const int cq = C / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cq) * 4;
    const int grp = (int)(i / n4_per_group);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + i * 4);
    const f32x4 d = *reinterpret_cast<const f32x4*>(dout + i * 4);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + grp * C + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + grp * C + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c);
    f32x4 sd = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
    if (training) {
      sd = *reinterpret_cast<const f32x4*>(sums + grp * (2 * C + 4) + c);
      sx = *reinterpret_cast<const f32x4*>(sums + grp * (2 * C + 4) + C + c);
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


int check_groups(int64_t M, int rows, int groups, int64_t rows_per_block_, const char* who) {
  SRX_REQUIRE(groups >= 1 && groups <= 64, "%s: groups must be in [1,64]", who);
  SRX_REQUIRE(M % groups == 0, "%s: %lld rows do not split into %d groups", who, (long long)M, groups);
  if (rows > 0) SRX_REQUIRE(rows % groups == 0, "%s: %d partial rows do not split into %d groups", who, rows, groups);
  if (groups > 1 && rows_per_block_ > 0)  // (one group: the last row block is simply partial)
    SRX_REQUIRE((M / groups) % rows_per_block_ == 0, "%s: a row block of %lld rows would straddle two groups of %lld", who,
                (long long)rows_per_block_, (long long)(M / groups));
  return SRX_OK;
}

int finalize_impl(const float* partials, int rows, int64_t M, int C, int groups, float eps, float momentum, float* save_mean,
                  float* save_invstd, float* running_mean, float* running_var, int64_t* nbt, void* stream) {
  SRX_REQUIRE(partials && save_mean && save_invstd && rows > 0 && M > 0 && C > 0, "bn_finalize: bad argument");
  SRX_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must come in pairs");
  if (int rc = check_groups(M, rows, groups, 0, "bn_finalize")) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)srx_cdiv(C, 4)), dim3(256), 0, srx_stream(stream), partials,
                     rows, M, C, groups, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt);
  SRX_CHECK_LAUNCH("bn_finalize_kernel");
  return SRX_OK;
}

int act_fwd_impl(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                 const float* residual, float* out, int64_t M, int C, int groups, int act, float slope, const float* prelu,
                 void* stream) {
  if (int rc = check_c(C, "bn_act_fwd")) return rc;
  SRX_REQUIRE(y && mean && invstd && gamma && beta && out && M > 0, "bn_act_fwd: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_fwd: PReLU needs its slope pointer");
  if (int rc = check_groups(M, 0, groups, 0, "bn_act_fwd")) return rc;
  const int64_t n4 = M * C / 4;
  hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(stream_grid(n4)), dim3(256), 0, srx_stream(stream), y, mean, invstd, gamma,
                     beta, residual, out, n4, C / 4, n4 / groups, act, slope, prelu);
  SRX_CHECK_LAUNCH("bn_act_fwd_kernel");
  return SRX_OK;
}

int bwd_reduce_impl(const float* dout, const float* y, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, float* sums, int64_t M, int C, int groups, int act, float slope, const float* prelu,
                    float* dgamma_acc, float* dbeta_acc, float* dprelu_acc, float* ws, size_t ws_floats, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd_reduce")) return rc;
  SRX_REQUIRE(dout && y && mean && invstd && gamma && beta && sums && ws && M > 0, "bn_act_bwd_reduce: bad argument");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd_reduce: PReLU needs its slope pointer");
  if (ws_floats < srx_bn_bwd_ws_floats(M, C)) SRX_FAIL(SRX_E_WORKSPACE, "bn_act_bwd_reduce: workspace too small");
  if (int rc = check_groups(M, 0, groups, rows_per_block(M), "bn_act_bwd_reduce")) return rc;
  const int rows = srx_bn_stat_rows(M);
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)rows), dim3(256), 0, st, dout, y, mean, invstd, gamma, beta,
                     ws, M, C, act, slope, prelu, rows_per_block(M), M / groups);
  SRX_CHECK_LAUNCH("bn_bwd_reduce_kernel");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)srx_cdiv(2 * C + 1, 4)), dim3(256), 0, st, ws, rows, C, groups,
                     sums, dgamma_acc, dbeta_acc, dprelu_acc, 1);
  SRX_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  return SRX_OK;
}

int bwd_apply_impl(const float* dout, const float* y, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, const float* sums, float* dy, int64_t M, int C, int groups, int act, float slope,
                   const float* prelu, int training, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd_apply")) return rc;
  SRX_REQUIRE(dout && y && mean && invstd && gamma && beta && dy && M > 0, "bn_act_bwd_apply: bad argument");
  SRX_REQUIRE(!training || sums, "bn_act_bwd_apply: training mode needs the reduced sums");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd_apply: PReLU needs its slope pointer");
  if (int rc = check_groups(M, 0, groups, 0, "bn_act_bwd_apply")) return rc;
  const int64_t n4 = M * C / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4)), dim3(256), 0, srx_stream(stream), dout, y, mean,
                     invstd, gamma, beta, sums, dy, n4, C, n4 / groups, 1.0f / (float)(M / groups), act, slope, prelu, training);
  SRX_CHECK_LAUNCH("bn_bwd_apply_kernel");
  return SRX_OK;
}

}  // namespace

extern "C" int srx_bn_stat_rows(int64_t M) { return (int)srx_cdiv(M, rows_per_block(M)); }
extern "C" int srx_bn_rows_per_block(int64_t M) { return rows_per_block(M); }

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
  return finalize_impl(partials, rows, M, C, 1, eps, momentum, save_mean, save_invstd, running_mean, running_var, nbt, stream);
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
  return act_fwd_impl(y, mean, invstd, gamma, beta, residual, out, M, C, 1, act, slope, prelu, stream);
}

extern "C" size_t srx_bn_bwd_ws_floats(int64_t M, int C) { return (size_t)srx_bn_stat_rows(M) * (2 * C + 4); }

extern "C" int srx_bn_act_bwd_reduce(const float* dout, const float* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, float* sums, int64_t M, int C, int act,
                                     float slope, const float* prelu, float* dgamma_acc, float* dbeta_acc,
                                     float* dprelu_acc, float* ws, size_t ws_floats, void* stream) {
  return bwd_reduce_impl(dout, y, mean, invstd, gamma, beta, sums, M, C, 1, act, slope, prelu, dgamma_acc, dbeta_acc,
                         dprelu_acc, ws, ws_floats, stream);
}

extern "C" int srx_bn_act_bwd_apply(const float* dout, const float* y, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const float* sums, float* dy, int64_t M,
                                    int C, int act, float slope, const float* prelu, int training, void* stream) {
  return bwd_apply_impl(dout, y, mean, invstd, gamma, beta, sums, dy, M, C, 1, act, slope, prelu, training, stream);
}

// Training-mode BatchNorm forward in one call: statistics from the conv's partial table, running statistics
// update, normalise + activation (+ residual); `groups` independent row ranges (see the top of this file).
extern "C" int srx_bn_train_fwd(const float* y, const float* partials, int rows, int64_t M, int C, int groups, float eps,
                                float momentum, const float* gamma, const float* beta, const float* residual,
                                float* out, int act, float slope, const float* prelu, float* save_mean,
                                float* save_invstd, float* running_mean, float* running_var, int64_t* nbt,
                                void* stream) {
  if (int rc = check_c(C, "bn_train_fwd")) return rc;
  SRX_REQUIRE(y && partials && gamma && beta && out && save_mean && save_invstd && rows > 0 && M > 0,
              "bn_train_fwd: bad argument");
  if (int rc = finalize_impl(partials, rows, M, C, groups, eps, momentum, save_mean, save_invstd, running_mean,
                             running_var, nbt, stream))
    return rc;
  return act_fwd_impl(y, save_mean, save_invstd, gamma, beta, residual, out, M, C, groups, act, slope, prelu, stream);
}

// Backward of act(BN(y)) in one call (reduce -> finalize -> apply); dy may be NULL when only parameter gradients
// are needed.  sums: [groups][2C+4].
extern "C" int srx_bn_act_bwd(const float* dout, const float* y, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float* sums, float* dy, int64_t M, int C, int groups,
                              int act, float slope, const float* prelu, int training, float* dgamma_acc,
                              float* dbeta_acc, float* dprelu_acc, float* ws, size_t ws_floats, void* stream) {
  if (int rc = bwd_reduce_impl(dout, y, mean, invstd, gamma, beta, sums, M, C, groups, act, slope, prelu, dgamma_acc,
                               dbeta_acc, dprelu_acc, ws, ws_floats, stream))
    return rc;
  if (!dy) return SRX_OK;
  return bwd_apply_impl(dout, y, mean, invstd, gamma, beta, sums, dy, M, C, groups, act, slope, prelu, training, stream);
}

// The last two passes of srx_bn_act_bwd on a partial table somebody else produced: `table` [rows][2C+4] holds, per row
// block, sum dz ([0,C)), sum dz * xhat ([C,2C)) and the PReLU slope partial in `prelu_cols` (1 or 2) columns from 2C --
// written by srx_conv2d_bwd_data_bn, the data gradient that produced `dout`.  One group.
extern "C" int srx_bn_act_bwd_finish(const float* dout, const float* y, const float* mean, const float* invstd,
                                     const float* gamma, const float* beta, const float* table, int rows, int prelu_cols,
                                     float* sums, float* dy, int64_t M, int C, int act, float slope, const float* prelu,
                                     float* dgamma_acc, float* dbeta_acc, float* dprelu_acc, void* stream) {
  if (int rc = check_c(C, "bn_act_bwd_finish")) return rc;
  SRX_REQUIRE(table && sums && M > 0 && rows > 0 && (!dy || (dout && y && mean && invstd && gamma && beta)),
              "bn_act_bwd_finish: bad argument");
  SRX_REQUIRE(prelu_cols == 1 || prelu_cols == 2, "bn_act_bwd_finish: the PReLU partial sits in 1 or 2 table columns");
  SRX_REQUIRE(act != SRX_ACT_PRELU || prelu, "bn_act_bwd_finish: PReLU needs its slope pointer");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)srx_cdiv(2 * C + 1, 4)), dim3(256), 0, srx_stream(stream), table, rows,
                     C, 1, sums, dgamma_acc, dbeta_acc, dprelu_acc, prelu_cols);
  SRX_CHECK_LAUNCH("bn_bwd_finalize_kernel");
  if (!dy) return SRX_OK;  // sums and parameter gradients only: the apply pass rides in the consumer (srx_conv2d_bwd_data_bn_in)
  return bwd_apply_impl(dout, y, mean, invstd, gamma, beta, sums, dy, M, C, 1, act, slope, prelu, 1, stream);
}
