// HBM-bound streaming kernels of the SRGAN/ESRGAN path: layout changes at the
// module boundary, standalone activations, residual scaling, column sums (bias
// gradients) and the 2x2 max pooling of VGG19.  16-byte accesses, grid capped at
// 4096 workgroups with a grid-stride loop (256 CUs x 8 blocks and change).
#include "srx_common.h"

namespace {

unsigned stream_grid(int64_t n) {
  int64_t b = srx_cdiv(n, 256);
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

#define GRID_STRIDE(i, n) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int Cs) {
  const int64_t total = (int64_t)N * H * W * Cs;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Cs);
    const int64_t pix = i / Cs;
    const int w = (int)(pix % W);
    const int64_t t = pix / W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    dst[i] = c < C ? src[(((int64_t)n * C + c) * H + h) * W + w] : 0.f;
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W,
                                    int Cs) {
  const int64_t total = (int64_t)N * C * H * W;
  GRID_STRIDE(i, total) {
    const int w = (int)(i % W);
    int64_t t = i / W;
    const int h = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    dst[i] = src[(((int64_t)n * H + h) * W + w) * Cs + c];
  }
}

// ------------------------------------------------------------- activations
__global__ void act_bwd_from_out_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                        float* __restrict__ dx, int64_t n, int act, float slope) {
  const int64_t n4 = n / 4;
  GRID_STRIDE(i, n4) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 o = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = o[e] > 0.f ? g[e] : (act == SRX_ACT_RELU ? 0.f : g[e] * slope);
    *reinterpret_cast<f32x4*>(dx + i * 4) = r;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    dx[i] = y[i] > 0.f ? dy[i] : (act == SRX_ACT_RELU ? 0.f : dy[i] * slope);
  }
}

// the same on a channel slice of wider NHWC tensors: row m, channels [0, C) at strides ldy / ly / ldx
__global__ void act_bwd_from_out_strided_kernel(const float* __restrict__ dy, int ldy, const float* __restrict__ y, int ly,
                                                float* __restrict__ dx, int ldx, int64_t M, int C, int act,
                                                float slope) {
  const int cq = C / 4;
  const int64_t total = M * cq;
  GRID_STRIDE(i, total) {
    const int64_t m = i / cq;
    const int q = (int)(i - m * cq);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + m * ldy + q * 4);
    const f32x4 o = *reinterpret_cast<const f32x4*>(y + m * ly + q * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = o[e] > 0.f ? g[e] : (act == SRX_ACT_RELU ? 0.f : g[e] * slope);
    *reinterpret_cast<f32x4*>(dx + m * ldx + q * 4) = r;
  }
}

__global__ void leaky_fwd_kernel(const float* __restrict__ x, const float* __restrict__ slope_ptr, float slope,
                                 float* __restrict__ y, int64_t n) {
  if (slope_ptr) slope = slope_ptr[0];
  const int64_t n4 = n / 4;
  GRID_STRIDE(i, n4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
  }
}

// dx = x>0 ? dy : a*dy ; per-block partial of d(a) = sum_{x<=0} dy*x
__global__ __launch_bounds__(256) void prelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ slope_ptr, float* __restrict__ dx,
                                                        float* __restrict__ partial, int64_t n) {
  __shared__ float red[4];
  const float a = slope_ptr[0];
  float acc = 0.f;
  const int64_t n4 = n / 4;
  GRID_STRIDE(i, n4) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool pos = v[e] > 0.f;
      r[e] = pos ? g[e] : a * g[e];
      acc += pos ? 0.f : g[e] * v[e];
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = r;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const bool pos = x[i] > 0.f;
    dx[i] = pos ? dy[i] : a * dy[i];
    acc += pos ? 0.f : dy[i] * x[i];
  }
  acc = srx_wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nb, float scale,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(red[0] * (double)scale);
}

__global__ void axpby_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ y,
                             int64_t n, float a, float b) {
  const int64_t n4 = n / 4;
  GRID_STRIDE(i, n4) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(x + i * 4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = a * u + b * v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    y[i] = a * x[i] + b * z[i];
  }
}

__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = 1.0f / (1.0f + expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                   int64_t n) {
  GRID_STRIDE(i, n) dx[i] = dy[i] * y[i] * (1.0f - y[i]);
}

// ------------------------------------------------------------------ colsum
// stage 1: grid (row blocks, column chunks of 64 float4 quads); a block is qpb quads x (256/qpb) row lanes,
// every lane streams float4s down its rows, LDS folds the row lanes.  C4 = C rounded up to 4 (<= Cs).
constexpr int CS_MAX_PART_ROWS = 256;
__host__ __device__ inline int64_t colsum_rows_per_block(int64_t M) {
  int64_t rpb = (M + CS_MAX_PART_ROWS - 1) / CS_MAX_PART_ROWS;
  return rpb < 16 ? 16 : rpb;
}
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                             int64_t M, int C4, int Cs, int qpb, int64_t rpb) {
  __shared__ f32x4 red[256];
  const int cq = C4 / 4;
  const int nrl = 256 / qpb;
  const int tid = threadIdx.x;
  const int ql = tid % qpb, rl = tid / qpb;
  const int q = blockIdx.y * qpb + ql;
  const int64_t rbeg = (int64_t)blockIdx.x * rpb;
  const int64_t rend = min(M, rbeg + rpb);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl && q < cq)
    for (int64_t r = rbeg + rl; r < rend; r += 4 * nrl) {  // four rows per trip, loads first (one L2 round trip per four rows)
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + min(r + u * nrl, rend - 1) * Cs + q * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * nrl < rend) s += v[u];
    }
  red[tid] = s;
  __syncthreads();
  if (tid < qpb && q < cq) {
    f32x4 t = red[tid];
    for (int k = 1; k < nrl; ++k) t += red[tid + k * qpb];
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C4 + q * 4) = t;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ part, int rows, int C, int C4, float* __restrict__ out,
                                    int accumulate) {
  // one wave per column, lanes over the (<= 256) partial rows
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  double s = 0.0;
  for (int r = lane; r < rows; r += 256) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = part[(size_t)min(r + 64 * u, rows - 1) * C4 + c];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + 64 * u < rows) s += (double)v[u];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[c] = accumulate ? out[c] + (float)s : (float)s;
}
// fallback for a channel stride that is not a multiple of 4 (e.g. the [B][1] output of the last Linear)
__global__ void colsum_scalar_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M, int C, int Cs,
                                     int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int64_t r = 0; r < M; ++r) s += (double)x[r * Cs + c];
  out[c] = accumulate ? out[c] + (float)s : (float)s;
}

// ----------------------------------------------------------------- pooling
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float* p = x + (((int64_t)n * H + 2 * oh) * W + 2 * ow) * C + q * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + C);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C + C);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    *reinterpret_cast<f32x4*>(y + i * 4) = m;
  }
}

// gradient goes to the first maximum in scan order, as ATen's max_pool2d does.  RELU: x is the output of a ReLU,
// whose backward (grad * (x > 0)) is applied on the way: a window whose maximum is 0 passes nothing.
template <bool RELU>
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                   int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int64_t base = (((int64_t)n * H + 2 * oh) * W + 2 * ow) * C + q * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + base);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + base + C);
    const f32x4 c = *reinterpret_cast<const f32x4*>(x + base + (int64_t)W * C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(x + base + (int64_t)W * C + C);
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    f32x4 ga, gb, gc, gd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int idx = 0;
      float m = a[e];
      if (b[e] > m) { m = b[e]; idx = 1; }
      if (c[e] > m) { m = c[e]; idx = 2; }
      if (d[e] > m) { m = d[e]; idx = 3; }
      if (RELU && !(m > 0.f)) idx = -1;
      ga[e] = idx == 0 ? g[e] : 0.f;
      gb[e] = idx == 1 ? g[e] : 0.f;
      gc[e] = idx == 2 ? g[e] : 0.f;
      gd[e] = idx == 3 ? g[e] : 0.f;
    }
    *reinterpret_cast<f32x4*>(dx + base) = ga;
    *reinterpret_cast<f32x4*>(dx + base + C) = gb;
    *reinterpret_cast<f32x4*>(dx + base + (int64_t)W * C) = gc;
    *reinterpret_cast<f32x4*>(dx + base + (int64_t)W * C + C) = gd;
  }
}

// ---- bf16-storage forms (round 5: the frozen VGG19 stack under autocast keeps its activations and gradients as bf16 between
// convs, gconv.hip "bf16 STORAGE").  The pools keep reading the fp32 output of the conv below them: which of two window
// entries is the maximum must not depend on bf16 rounding (ties would route the gradient to another pixel than the
// recipe's fp32 comparison does); their OUTPUTS feed convs only and are rounded here instead of in the consumer's loader.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 pack_bf16x4(const f32x4& v) {
  const bf16x2_t lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};
  return make_uint2(__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi));
}
__device__ __forceinline__ f32x4 unpack_bf16x4(const uint2 u) {
  return (f32x4){__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                 __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
}

__global__ void maxpool_fwd_to_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float* p = x + (((int64_t)n * H + 2 * oh) * W + 2 * ow) * C + q * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + C);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(p + (int64_t)W * C + C);
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
    *reinterpret_cast<uint2*>(y + i * 4) = pack_bf16x4(m);
  }
}

// maxpool_bwd_kernel<true> with a bf16 gradient in and out (the values are routed, never added: rounding commutes)
__global__ void maxpool_relu_bwd_bf16_kernel(const unsigned short* __restrict__ dy, const float* __restrict__ x,
                                             unsigned short* __restrict__ dx, int N, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, cq = C / 4;
  const int64_t total = (int64_t)N * Ho * Wo * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int64_t base = (((int64_t)n * H + 2 * oh) * W + 2 * ow) * C + q * 4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + base);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + base + C);
    const f32x4 c = *reinterpret_cast<const f32x4*>(x + base + (int64_t)W * C);
    const f32x4 d = *reinterpret_cast<const f32x4*>(x + base + (int64_t)W * C + C);
    const f32x4 g = unpack_bf16x4(*reinterpret_cast<const uint2*>(dy + i * 4));
    f32x4 ga, gb, gc, gd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int idx = 0;
      float m = a[e];
      if (b[e] > m) { m = b[e]; idx = 1; }
      if (c[e] > m) { m = c[e]; idx = 2; }
      if (d[e] > m) { m = d[e]; idx = 3; }
      if (!(m > 0.f)) idx = -1;
      ga[e] = idx == 0 ? g[e] : 0.f;
      gb[e] = idx == 1 ? g[e] : 0.f;
      gc[e] = idx == 2 ? g[e] : 0.f;
      gd[e] = idx == 3 ? g[e] : 0.f;
    }
    *reinterpret_cast<uint2*>(dx + base) = pack_bf16x4(ga);
    *reinterpret_cast<uint2*>(dx + base + C) = pack_bf16x4(gb);
    *reinterpret_cast<uint2*>(dx + base + (int64_t)W * C) = pack_bf16x4(gc);
    *reinterpret_cast<uint2*>(dx + base + (int64_t)W * C + C) = pack_bf16x4(gd);
  }
}

// act_bwd_from_out_kernel with a bf16 result (n a multiple of 4)
__global__ void act_bwd_from_out_to_bf16_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                unsigned short* __restrict__ dx, int64_t n4, int act, float slope) {
  GRID_STRIDE(i, n4) {
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
    const f32x4 o = *reinterpret_cast<const f32x4*>(y + i * 4);
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = o[e] > 0.f ? g[e] : (act == SRX_ACT_RELU ? 0.f : g[e] * slope);
    *reinterpret_cast<uint2*>(dx + i * 4) = pack_bf16x4(r);
  }
}

// dst[m][dst_off + c] (+)= src[m][src_off + c], c < C: channel concat / split of NHWC tensors
__global__ void copy_channels_kernel(const float* __restrict__ src, int scs, int soff, float* __restrict__ dst, int dcs,
                                     int doff, int C, int64_t M, int accumulate) {
  const int cq = C / 4;
  const int64_t total = M * cq;
  GRID_STRIDE(i, total) {
    const int64_t m = i / cq;
    const int q = (int)(i - m * cq);
    f32x4 v = *reinterpret_cast<const f32x4*>(src + m * scs + soff + q * 4);
    f32x4* d = reinterpret_cast<f32x4*>(dst + m * dcs + doff + q * 4);
    if (accumulate) v += *d;
    *d = v;
  }
}

// y[m][yoff + c] = a * x[m][xoff + c] + b * z[m][zoff + c]: axpby on channel slices of tensors with different row strides
// (ESRGAN's trunk: `out * 0.2 + x` of an RRDB between the first 64 channels of two 192-channel dense-block buffers)
__global__ void axpby_channels_kernel(const float* __restrict__ x, int xcs, int xoff, const float* __restrict__ z, int zcs,
                                      int zoff, float* __restrict__ y, int ycs, int yoff, int C, int64_t M, float a, float b) {
  const int cq = C / 4;
  const int64_t total = M * cq;
  GRID_STRIDE(i, total) {
    const int64_t m = i / cq;
    const int q = (int)(i - m * cq);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + m * xcs + xoff + q * 4);
    const f32x4 zv = *reinterpret_cast<const f32x4*>(z + m * zcs + zoff + q * 4);
    *reinterpret_cast<f32x4*>(y + m * ycs + yoff + q * 4) = xv * a + zv * b;
  }
}

// F.interpolate(scale_factor=2, mode='nearest') on NHWC and its adjoint
__global__ void upsample2x_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int cq = C / 4;
  const int64_t total = (int64_t)N * 2 * H * 2 * W * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int ow = (int)(t % (2 * W));
    t /= 2 * W;
    const int oh = (int)(t % (2 * H));
    const int n = (int)(t / (2 * H));
    *reinterpret_cast<f32x4*>(y + i * 4) =
        *reinterpret_cast<const f32x4*>(x + (((int64_t)n * H + (oh >> 1)) * W + (ow >> 1)) * C + q * 4);
  }
}
__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
  const int cq = C / 4;
  const int64_t total = (int64_t)N * H * W * cq;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % cq);
    int64_t t = i / cq;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float* p = dy + (((int64_t)n * 2 * H + 2 * h) * 2 * W + 2 * w) * C + q * 4;
    const f32x4 s = *reinterpret_cast<const f32x4*>(p) + *reinterpret_cast<const f32x4*>(p + C) +
                    *reinterpret_cast<const f32x4*>(p + (int64_t)2 * W * C) +
                    *reinterpret_cast<const f32x4*>(p + (int64_t)2 * W * C + C);
    *reinterpret_cast<f32x4*>(dx + i * 4) = s;
  }
}

}  // namespace

extern "C" int srx_copy_channels(const float* src, int src_cs, int src_off, float* dst, int dst_cs, int dst_off, int C,
                                 int64_t M, int accumulate, void* stream) {
  SRX_REQUIRE(src && dst && C > 0 && M > 0, "copy_channels: bad argument");
  SRX_REQUIRE(C % 4 == 0 && src_cs % 4 == 0 && dst_cs % 4 == 0 && src_off % 4 == 0 && dst_off % 4 == 0,
              "copy_channels: channel counts and offsets must be multiples of 4");
  SRX_REQUIRE(src_off + C <= src_cs && dst_off + C <= dst_cs, "copy_channels: slice out of range");
  hipLaunchKernelGGL(copy_channels_kernel, dim3(stream_grid(M * (C / 4))), dim3(256), 0, srx_stream(stream), src, src_cs,
                     src_off, dst, dst_cs, dst_off, C, M, accumulate);
  SRX_CHECK_LAUNCH("copy_channels_kernel");
  return SRX_OK;
}

extern "C" int srx_axpby_channels(const float* x, int x_cs, int x_off, const float* z, int z_cs, int z_off, float* y,
                                  int y_cs, int y_off, int C, int64_t M, float a, float b, void* stream) {
  SRX_REQUIRE(x && z && y && C > 0 && M > 0, "axpby_channels: bad argument");
  SRX_REQUIRE(C % 4 == 0 && x_cs % 4 == 0 && z_cs % 4 == 0 && y_cs % 4 == 0 && x_off % 4 == 0 && z_off % 4 == 0 && y_off % 4 == 0,
              "axpby_channels: channel counts and offsets must be multiples of 4");
  SRX_REQUIRE(x_off >= 0 && z_off >= 0 && y_off >= 0 && x_off + C <= x_cs && z_off + C <= z_cs && y_off + C <= y_cs,
              "axpby_channels: slice out of range");
  hipLaunchKernelGGL(axpby_channels_kernel, dim3(stream_grid(M * (C / 4))), dim3(256), 0, srx_stream(stream), x, x_cs, x_off,
                     z, z_cs, z_off, y, y_cs, y_off, C, M, a, b);
  SRX_CHECK_LAUNCH("axpby_channels_kernel");
  return SRX_OK;
}

extern "C" int srx_upsample_nearest2x_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRX_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample_nearest2x_fwd: bad argument");
  hipLaunchKernelGGL(upsample2x_fwd_kernel, dim3(stream_grid((int64_t)N * 4 * H * W * (C / 4))), dim3(256), 0,
                     srx_stream(stream), x, y, N, H, W, C);
  SRX_CHECK_LAUNCH("upsample2x_fwd_kernel");
  return SRX_OK;
}

extern "C" int srx_upsample_nearest2x_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  SRX_REQUIRE(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample_nearest2x_bwd: bad argument");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(stream_grid((int64_t)N * H * W * (C / 4))), dim3(256), 0,
                     srx_stream(stream), dy, dx, N, H, W, C);
  SRX_CHECK_LAUNCH("upsample2x_bwd_kernel");
  return SRX_OK;
}

extern "C" int srx_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cs, void* stream) {
  SRX_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cs >= C, "nchw_to_nhwc: bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(stream_grid((int64_t)N * H * W * Cs)), dim3(256), 0, srx_stream(stream),
                     src, dst, N, C, H, W, Cs);
  SRX_CHECK_LAUNCH("nchw_to_nhwc_kernel");
  return SRX_OK;
}

extern "C" int srx_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, int Cs, void* stream) {
  SRX_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Cs >= C, "nhwc_to_nchw: bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(stream_grid((int64_t)N * C * H * W)), dim3(256), 0, srx_stream(stream),
                     src, dst, N, C, H, W, Cs);
  SRX_CHECK_LAUNCH("nhwc_to_nchw_kernel");
  return SRX_OK;
}

extern "C" size_t srx_colsum_ws_floats(int64_t M, int C) {
  const int64_t rpb = colsum_rows_per_block(M);
  return (size_t)srx_cdiv(M, rpb) * (size_t)srx_roundup(C, 4);
}

extern "C" int srx_colsum(const float* x, float* out, int64_t M, int C, int Cs, int accumulate, float* ws,
                          size_t ws_floats, void* stream) {
  SRX_REQUIRE(x && out && ws && M > 0 && C > 0 && Cs >= C, "colsum: bad argument");
  hipStream_t st = srx_stream(stream);
  if (Cs % 4 != 0 || ((uintptr_t)x % 16) != 0) {
    SRX_REQUIRE(M <= 65536, "colsum: unaligned input only supported for small M");
    hipLaunchKernelGGL(colsum_scalar_kernel, dim3((unsigned)srx_cdiv(C, 64)), dim3(64), 0, st, x, out, M, C, Cs,
                       accumulate);
    SRX_CHECK_LAUNCH("colsum_scalar_kernel");
    return SRX_OK;
  }
  if (ws_floats < srx_colsum_ws_floats(M, C)) SRX_FAIL(SRX_E_WORKSPACE, "colsum: workspace too small");
  const int C4 = (int)srx_roundup(C, 4), cq = C4 / 4;
  int qpb = 1;
  while (qpb < cq && qpb < 64) qpb *= 2;  // power of two <= 64 so it divides 256
  const int64_t rpb = colsum_rows_per_block(M);
  const int rows = (int)srx_cdiv(M, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)rows, (unsigned)srx_cdiv(cq, qpb)), dim3(256), 0, st, x, ws,
                     M, C4, Cs, qpb, rpb);
  SRX_CHECK_LAUNCH("colsum_partial_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)srx_cdiv(C, 4)), dim3(256), 0, st, ws, rows, C, C4, out,
                     accumulate);
  SRX_CHECK_LAUNCH("colsum_final_kernel");
  return SRX_OK;
}

extern "C" int srx_act_bwd_from_out(const float* dy, const float* y, float* dx, int64_t n, int act, float slope,
                                    void* stream) {
  SRX_REQUIRE(dy && y && dx && n > 0, "act_bwd_from_out: bad argument");
  SRX_REQUIRE(act == SRX_ACT_RELU || act == SRX_ACT_LRELU, "act_bwd_from_out: act must be RELU or LRELU");
  hipLaunchKernelGGL(act_bwd_from_out_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, srx_stream(stream), dy, y, dx, n,
                     act, slope);
  SRX_CHECK_LAUNCH("act_bwd_from_out_kernel");
  return SRX_OK;
}

extern "C" int srx_act_bwd_from_out_strided(const float* dy, int ldy, const float* y, int ly, float* dx, int ldx,
                                            int64_t M, int C, int act, float slope, void* stream) {
  SRX_REQUIRE(dy && y && dx && M > 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && ly % 4 == 0 && ldx % 4 == 0 &&
                  ldy >= C && ly >= C && ldx >= C,
              "act_bwd_from_out_strided: bad argument");
  SRX_REQUIRE(act == SRX_ACT_RELU || act == SRX_ACT_LRELU, "act_bwd_from_out_strided: act must be RELU or LRELU");
  hipLaunchKernelGGL(act_bwd_from_out_strided_kernel, dim3(stream_grid(M * (C / 4))), dim3(256), 0, srx_stream(stream),
                     dy, ldy, y, ly, dx, ldx, M, C, act, slope);
  SRX_CHECK_LAUNCH("act_bwd_from_out_strided_kernel");
  return SRX_OK;
}

extern "C" int srx_prelu_fwd(const float* x, const float* slope, float* y, int64_t n, void* stream) {
  SRX_REQUIRE(x && slope && y && n > 0, "prelu_fwd: bad argument");
  hipLaunchKernelGGL(leaky_fwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, srx_stream(stream), x, slope, 0.f, y, n);
  SRX_CHECK_LAUNCH("leaky_fwd_kernel");
  return SRX_OK;
}

extern "C" int srx_lrelu_fwd(const float* x, float* y, int64_t n, float slope, void* stream) {
  SRX_REQUIRE(x && y && n > 0, "lrelu_fwd: bad argument");
  hipLaunchKernelGGL(leaky_fwd_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, srx_stream(stream), x,
                     (const float*)nullptr, slope, y, n);
  SRX_CHECK_LAUNCH("leaky_fwd_kernel");
  return SRX_OK;
}

extern "C" int srx_prelu_bwd(const float* dy, const float* x, const float* slope, float* dx, float* dslope,
                             int accumulate, int64_t n, float* ws, void* stream) {
  SRX_REQUIRE(dy && x && slope && dx && dslope && ws && n > 0, "prelu_bwd: bad argument");
  unsigned nb = stream_grid(n / 4);
  if (nb > 1024) nb = 1024;
  hipStream_t st = srx_stream(stream);
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(nb), dim3(256), 0, st, dy, x, slope, dx, ws, n);
  SRX_CHECK_LAUNCH("prelu_bwd_kernel");
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, ws, (int)nb, 1.0f, dslope, accumulate);
  SRX_CHECK_LAUNCH("sum_partials_kernel");
  return SRX_OK;
}

extern "C" int srx_axpby(const float* x, const float* z, float* y, int64_t n, float a, float b, void* stream) {
  SRX_REQUIRE(x && z && y && n > 0, "axpby: bad argument");
  hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, srx_stream(stream), x, z, y, n, a, b);
  SRX_CHECK_LAUNCH("axpby_kernel");
  return SRX_OK;
}

extern "C" int srx_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream) {
  SRX_REQUIRE(x && y && n > 0, "sigmoid_fwd: bad argument");
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(stream_grid(n)), dim3(256), 0, srx_stream(stream), x, y, n);
  SRX_CHECK_LAUNCH("sigmoid_fwd_kernel");
  return SRX_OK;
}

extern "C" int srx_sigmoid_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  SRX_REQUIRE(dy && y && dx && n > 0, "sigmoid_bwd: bad argument");
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(stream_grid(n)), dim3(256), 0, srx_stream(stream), dy, y, dx, n);
  SRX_CHECK_LAUNCH("sigmoid_bwd_kernel");
  return SRX_OK;
}

extern "C" int srx_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SRX_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_fwd: bad argument");
  SRX_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2x2_fwd: H, W must be even and C a multiple of 4");
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     srx_stream(stream), x, y, N, H, W, C);
  SRX_CHECK_LAUNCH("maxpool_fwd_kernel");
  return SRX_OK;
}

// y = bf16(maxpool2x2(x)): x fp32 NHWC, y bf16 NHWC (bf16-storage conv stacks)
extern "C" int srx_maxpool2x2_fwd_to_bf16(const float* x, void* y, int N, int H, int W, int C, void* stream) {
  SRX_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_fwd_to_bf16: bad argument");
  SRX_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2x2_fwd_to_bf16: H, W must be even and C a multiple of 4");
  hipLaunchKernelGGL(maxpool_fwd_to_bf16_kernel, dim3(stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     srx_stream(stream), x, static_cast<unsigned short*>(y), N, H, W, C);
  SRX_CHECK_LAUNCH("maxpool_fwd_to_bf16_kernel");
  return SRX_OK;
}

// srx_maxpool2x2_relu_bwd with bf16 gradients: dy bf16 [N][H/2][W/2][C], x fp32 [N][H][W][C] (a ReLU output), dx bf16 like x
extern "C" int srx_maxpool2x2_relu_bwd_bf16(const void* dy, const float* x, void* dx, int N, int H, int W, int C, void* stream) {
  SRX_REQUIRE(dy && x && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_relu_bwd_bf16: bad argument");
  SRX_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2x2_relu_bwd_bf16: H, W must be even and C a multiple of 4");
  hipLaunchKernelGGL(maxpool_relu_bwd_bf16_kernel, dim3(stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     srx_stream(stream), static_cast<const unsigned short*>(dy), x, static_cast<unsigned short*>(dx), N, H, W, C);
  SRX_CHECK_LAUNCH("maxpool_relu_bwd_bf16_kernel");
  return SRX_OK;
}

// srx_act_bwd_from_out with a bf16 result: dx = bf16(dy * act'(y)), n a multiple of 4
extern "C" int srx_act_bwd_from_out_to_bf16(const float* dy, const float* y, void* dx, int64_t n, int act, float slope, void* stream) {
  SRX_REQUIRE(dy && y && dx && n > 0 && n % 4 == 0, "act_bwd_from_out_to_bf16: bad argument");
  SRX_REQUIRE(act == SRX_ACT_RELU || act == SRX_ACT_LRELU, "act_bwd_from_out_to_bf16: act must be RELU or LRELU");
  hipLaunchKernelGGL(act_bwd_from_out_to_bf16_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, srx_stream(stream), dy, y,
                     static_cast<unsigned short*>(dx), n / 4, act, slope);
  SRX_CHECK_LAUNCH("act_bwd_from_out_to_bf16_kernel");
  return SRX_OK;
}

extern "C" int srx_maxpool2x2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C,
                                  void* stream) {
  SRX_REQUIRE(dy && x && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_bwd: bad argument");
  SRX_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2x2_bwd: H, W must be even and C a multiple of 4");
  hipLaunchKernelGGL(maxpool_bwd_kernel<false>, dim3(stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     srx_stream(stream), dy, x, dx, N, H, W, C);
  SRX_CHECK_LAUNCH("maxpool_bwd_kernel");
  return SRX_OK;
}

extern "C" int srx_maxpool2x2_relu_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C,
                                       void* stream) {
  SRX_REQUIRE(dy && x && dx && N > 0 && H > 0 && W > 0 && C > 0, "maxpool2x2_relu_bwd: bad argument");
  SRX_REQUIRE(H % 2 == 0 && W % 2 == 0 && C % 4 == 0, "maxpool2x2_relu_bwd: H, W must be even and C a multiple of 4");
  hipLaunchKernelGGL(maxpool_bwd_kernel<true>, dim3(stream_grid((int64_t)N * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                     srx_stream(stream), dy, x, dx, N, H, W, C);
  SRX_CHECK_LAUNCH("maxpool_bwd_kernel");
  return SRX_OK;
}

// ------------------------------------------------------------------ loss ring
// One record per train step: up to four device scalars appended to a ring, the slot taken from a device-side
// counter, so that the launch is identical every step and can sit inside a replayed hipGraph.
namespace {
__global__ void ring_push_kernel(const float* a, const float* b, const float* c, const float* d, int n, float* ring,
                                 int* counter, int cap) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int i = *counter;
  float* slot = ring + (size_t)(i % cap) * 4;
  slot[0] = *a;
  if (n > 1) slot[1] = *b;
  if (n > 2) slot[2] = *c;
  if (n > 3) slot[3] = *d;
  *counter = i + 1;
}
}  // namespace

extern "C" int srx_ring_push(const float* a, const float* b, const float* c, const float* d, int n, float* ring,
                             int* counter, int cap, void* stream) {
  SRX_REQUIRE(n >= 1 && n <= 4 && a && (n < 2 || b) && (n < 3 || c) && (n < 4 || d) && ring && counter && cap > 0,
              "ring_push: bad argument");
  hipLaunchKernelGGL(ring_push_kernel, dim3(1), dim3(64), 0, srx_stream(stream), a, b, c, d, n, ring, counter, cap);
  SRX_CHECK_LAUNCH("ring_push_kernel");
  return SRX_OK;
}
