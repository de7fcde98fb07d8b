// Probe: does v_mfma_f32_32x32x2_f32 (f32 in, the fp32 VECTOR rate) overlap with plain VALU work on the same SIMD?
//  (a) one wave per SIMD: M MFMAs per iteration with K independent v_fma_f32 placed between them;
//  (b) two waves per SIMD: wave 0..3 MFMA only, waves 4..7 VALU only (same SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int K>
__global__ __launch_bounds__(256) void mixed(float* out, int iters, float seed) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  float x = seed + threadIdx.x * 0.37f, y = seed * 1.3f - threadIdx.x * 0.11f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// 512 threads: waves 0..3 MFMA (mode bit 0), waves 4..7 VALU (mode bit 1): K fma per "slot" of 64 clocks
template <int K>
__global__ __launch_bounds__(512) void split(float* out, int iters, float seed, int mode) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float x = seed + threadIdx.x * 0.37f, y = seed * 1.3f - threadIdx.x * 0.11f, s = 0.f;
  if (wave < 4) {
    if (!(mode & 1)) { out[blockIdx.x * 512 + threadIdx.x] = 0.f; return; }
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  } else {
    if (!(mode & 2)) { out[blockIdx.x * 512 + threadIdx.x] = 0.f; return; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = x + i;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 4 * K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
    }
    for (int i = 0; i < 8; ++i) s += v[i];
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 512 * 4);
  const int iters = 20000;
  { float w = timeit([&] { hipLaunchKernelGGL(mixed<0>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); }); (void)w; }
#define RUN(K) { float ms = timeit([&] { hipLaunchKernelGGL(mixed<K>, dim3(256), dim3(256), 0, 0, d, iters, 0.5f); }); \
    printf("one wave / SIMD, %2d v_fma_f32 after each MFMA: %.3f ms = %.1f clocks (at 2.4 GHz) per MFMA slot\n", K, ms, ms * 1e-3 * 2.4e9 / (4.0 * iters)); }
  RUN(0) RUN(2) RUN(4) RUN(8) RUN(12) RUN(16) RUN(24)
#define RUNS(K, mode) { float ms = timeit([&] { hipLaunchKernelGGL(split<K>, dim3(256), dim3(512), 0, 0, d, iters, 0.5f, mode); }); \
    printf("two waves / SIMD, mode %d (1 = MFMA waves only, 2 = VALU waves only, 3 = both), %2d v_fma_f32 per MFMA slot: %.3f ms = %.1f clocks per slot\n", mode, K, ms, ms * 1e-3 * 2.4e9 / (4.0 * iters)); }
  RUNS(8, 1) RUNS(8, 2) RUNS(8, 3) RUNS(16, 2) RUNS(16, 3) RUNS(4, 2) RUNS(4, 3)
  return 0;
}
