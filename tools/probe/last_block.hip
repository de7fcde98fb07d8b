// Probe: what does it cost to finish a two-stage reduction in the PRODUCER's launch ("the last workgroup to arrive reduces the table")
// instead of in a tiny launch of its own?  The shape is the residual tower's: 256 workgroups, each writes 36 rows x 64 channels of
// output and one row [64][2] of partial sums; a finalize turns the 256 x 64 x 2 table into mean / invstd (fp64 sums).
//   A  producer alone                           (floor: the dependent chain of producers)
//   B  producer + finalize launch               (what the step runs today)
//   C  producer with tail: __threadfence() + ticket, last workgroup reduces with plain loads behind a second fence
//   D  producer with tail: partial rows stored and read with agent-scope (sc1) accesses, no L2 write-back / invalidate fence
// Each as a chain of 66 dependent launches in a replayed hipGraph; prints us per layer.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int WGS = 256, C = 64, RT = 36;

__device__ __forceinline__ void finalize_body(const float* part, int rows, float* mean, float* invstd, int mode, double* red) {
  // thread = (channel tid & 63, row group tid >> 6): its rows in order, then the four groups in order
  const int tid = threadIdx.x, c = tid & 63, rg = tid >> 6;
  const int per = rows / 4;
  double s = 0.0, s2 = 0.0;
  for (int r0 = rg * per; r0 < (rg + 1) * per; r0 += 16) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float* p = part + ((size_t)(r0 + u) * C + c) * 2;
      if (mode == 2) {
        v[u].x = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[u].y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        v[u] = *reinterpret_cast<const float2*>(p);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) { s += (double)v[u].x; s2 += (double)v[u].y; }
  }
  red[tid * 2] = s; red[tid * 2 + 1] = s2;
  __syncthreads();
  if (tid < 64) {
    double t = 0.0, t2 = 0.0;
    for (int g = 0; g < 4; ++g) { t += red[(g * 64 + tid) * 2]; t2 += red[(g * 64 + tid) * 2 + 1]; }
    const double M = (double)rows * RT, mu = t / M;
    double var = t2 / M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[tid] = (float)mu;
    invstd[tid] = (float)(1.0 / sqrt(var + 1e-5));
  }
}

// mode 0: no tail, 1: fence + ticket, 2: sc1 accesses + ticket
template <int MODE>
__global__ __launch_bounds__(256) void producer(const float* __restrict__ in, const float* __restrict__ mean_in, float* __restrict__ out,
                                                float* __restrict__ part, unsigned* counter, float* mean, float* invstd) {
  __shared__ double red[512];
  __shared__ unsigned ticket;
  const int tid = threadIdx.x, col = tid & 63, rq = tid >> 6;
  const float mu = mean_in[col];  // (dependence on the finalize of the layer before)
  float s1 = 0.f, s2 = 0.f;
  for (int r = rq; r < RT; r += 4) {
    const size_t o = ((size_t)blockIdx.x * RT + r) * C + col;
    const float v = in[o] * 0.5f + mu * 1e-3f;
    out[o] = v;
    s1 += v; s2 += v * v;
  }
  __shared__ float ps[2][256];
  ps[0][tid] = s1; ps[1][tid] = s2;
  __syncthreads();
  if (tid < 64) {
    const float a = (ps[0][tid] + ps[0][tid + 64]) + (ps[0][tid + 128] + ps[0][tid + 192]);
    const float b = (ps[1][tid] + ps[1][tid + 64]) + (ps[1][tid + 128] + ps[1][tid + 192]);
    float* p = part + ((size_t)blockIdx.x * C + tid) * 2;
    if (MODE >= 2) {
      __hip_atomic_store(p, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      p[0] = a; p[1] = b;
    }
  }
  if (MODE == 0) return;
  if (MODE == 1) __threadfence();
  if (MODE >= 2) __builtin_amdgcn_s_waitcnt(0);  // the write-through stores of this wave have been acknowledged
  __syncthreads();
  if (MODE == 4) {  // two levels: eight counters (workgroups are dealt to the XCDs round robin), then one
    if (tid == 0) {
      unsigned t = __hip_atomic_fetch_add(counter + 16 * (1 + (blockIdx.x & 7)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned last = 0;
      if (t == gridDim.x / 8 - 1) {
        __hip_atomic_store(counter + 16 * (1 + (blockIdx.x & 7)), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 7 ? 1u : 0u;
      }
      ticket = last ? gridDim.x - 1 : 0;
    }
  } else if (tid == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket != gridDim.x - 1) return;
  if (MODE == 1) __threadfence();
  if (MODE != 3) finalize_body(part, gridDim.x, mean, invstd, MODE == 4 ? 2 : MODE, red);
  if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
}

// G: no finalize anywhere -- EVERY workgroup of the consumer reduces the whole table of the layer before in its prologue
template <int F64>
__global__ __launch_bounds__(256) void consumer_reduces(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ part_in,
                                                        float* __restrict__ part, int rows) {
  __shared__ double red[512];
  __shared__ float smean[64];
  __shared__ float ps[2][256];
  const int tid = threadIdx.x, col = tid & 63, rq = tid >> 6;
  {
    const int c = tid & 63, rg = tid >> 6, per = rows / 4;
    double s = 0.0, s2 = 0.0;
    float fs = 0.f, fs2 = 0.f;
    for (int r0 = rg * per; r0 < (rg + 1) * per; r0 += 16) {
      float2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float2*>(part_in + ((size_t)(r0 + u) * C + c) * 2);
#pragma unroll
      for (int u = 0; u < 16; ++u) { if (F64) { s += (double)v[u].x; s2 += (double)v[u].y; } else { fs += v[u].x; fs2 += v[u].y; } }
    }
    if (!F64) { s = fs; s2 = fs2; }
    red[tid * 2] = s; red[tid * 2 + 1] = s2;
    __syncthreads();
    if (tid < 64) {
      double t = 0.0, t2 = 0.0;
      for (int g = 0; g < 4; ++g) { t += red[(g * 64 + tid) * 2]; t2 += red[(g * 64 + tid) * 2 + 1]; }
      smean[tid] = (float)(t / ((double)rows * RT));
    }
    __syncthreads();
  }
  const float mu = smean[col];
  float s1 = 0.f, s2 = 0.f;
  for (int r = rq; r < RT; r += 4) {
    const size_t o = ((size_t)blockIdx.x * RT + r) * C + col;
    const float v = in[o] * 0.5f + mu * 1e-3f;
    out[o] = v;
    s1 += v; s2 += v * v;
  }
  ps[0][tid] = s1; ps[1][tid] = s2;
  __syncthreads();
  if (tid < 64) {
    float* p = part + ((size_t)blockIdx.x * C + tid) * 2;
    p[0] = (ps[0][tid] + ps[0][tid + 64]) + (ps[0][tid + 128] + ps[0][tid + 192]);
    p[1] = (ps[1][tid] + ps[1][tid + 64]) + (ps[1][tid + 128] + ps[1][tid + 192]);
  }
}

__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ part, int rows, float* mean, float* invstd) {
  // the library's form: one wave per channel, lanes stride over the rows, fp64 butterfly
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  double s = 0.0, s2 = 0.0;
  for (int r = lane; r < rows; r += 256) {
    float2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(part + ((size_t)min(r + 64 * u, rows - 1) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (r + 64 * u < rows) { s += (double)v[u].x; s2 += (double)v[u].y; }
  }
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); s2 += __shfl_xor(s2, o, 64); }
  if (lane) return;
  const double M = (double)rows * RT, mu = s / M;
  double var = s2 / M - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + 1e-5));
}

int main() {
  const size_t n = (size_t)WGS * RT * C;
  float *buf[2], *part, *part2, *mean, *invstd; unsigned* counter;
  CK(hipMalloc(&buf[0], n * 4)); CK(hipMalloc(&buf[1], n * 4));
  CK(hipMalloc(&part, WGS * C * 2 * 4)); CK(hipMalloc(&part2, WGS * C * 2 * 4)); CK(hipMemset(part, 0, WGS * C * 2 * 4)); CK(hipMemset(part2, 0, WGS * C * 2 * 4)); CK(hipMalloc(&mean, C * 4)); CK(hipMalloc(&invstd, C * 4)); CK(hipMalloc(&counter, 4096));
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.3f;
  CK(hipMemcpy(buf[0], h.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemset(mean, 0, C * 4)); CK(hipMemset(counter, 0, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int LAYERS = 66;
  float ref_mean[C];
  for (int variant = 0; variant < 8; ++variant) {
    CK(hipMemcpy(buf[0], h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(mean, 0, C * 4));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int l = 0; l < LAYERS; ++l) {
      const float* in = buf[l & 1]; float* out = buf[(l + 1) & 1];
      if (variant == 0) hipLaunchKernelGGL(producer<0>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
      if (variant == 1) {
        hipLaunchKernelGGL(producer<0>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
        hipLaunchKernelGGL(finalize_kernel, dim3(16), dim3(256), 0, st, part, WGS, mean, invstd);
      }
      if (variant == 2) hipLaunchKernelGGL(producer<1>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
      if (variant == 3) hipLaunchKernelGGL(producer<2>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
      if (variant == 6) hipLaunchKernelGGL(consumer_reduces<1>, dim3(WGS), dim3(256), 0, st, in, out, (l & 1) ? part2 : part, (l & 1) ? part : part2, WGS);
      if (variant == 7) hipLaunchKernelGGL(consumer_reduces<0>, dim3(WGS), dim3(256), 0, st, in, out, (l & 1) ? part2 : part, (l & 1) ? part : part2, WGS);
      if (variant == 4) hipLaunchKernelGGL(producer<3>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
      if (variant == 5) hipLaunchKernelGGL(producer<4>, dim3(WGS), dim3(256), 0, st, in, mean, out, part, counter, mean, invstd);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float hm[C];
    CK(hipMemcpy(hm, mean, C * 4, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    const int REP = 50;
    for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int bad = 0;
    if (variant == 1) for (int c = 0; c < C; ++c) ref_mean[c] = hm[c];
    if (variant >= 2 && variant != 4 && variant < 6) for (int c = 0; c < C; ++c) bad += hm[c] != ref_mean[c];
    const char* names[] = {"A producer alone", "B producer + finalize launch", "C tail: fences + ticket", "D tail: sc1 accesses + ticket", "E sc1 stores + ticket, NO reduce", "F sc1 + two-level tickets", "G every consumer workgroup reduces (fp64)", "H ... (fp32 row sums)"};
    printf("%-32s %7.2f us per layer   mean[0] %.9g  mismatches vs B %d\n", names[variant], ms * 1e3 / REP / LAYERS, hm[0], bad);
  }
  return 0;
}
