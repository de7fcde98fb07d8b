// Probe: issue rate of the fp32 MFMA shapes on gfx950 (cycles per instruction per SIMD, one wave per SIMD),
// independent accumulators, and of a 32x32x2 / 4x4x1 alternation (rowtile.hip's inner loop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
  f32x16 a0, a1;
  f32x4 b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0, c0 = b0, c1 = b0, c2 = b0, c3 = b0;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  float x = seed + threadIdx.x * 0.37f, y = seed * 1.3f - threadIdx.x * 0.11f;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {  // 4 x 32x32x2
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a1, 0, 0, 0);
    } else if (MODE == 1) {  // 8 x 4x4x1, independent
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, b0, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, b1, 0, 0, 0);
      b2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, b2, 0, 0, 0);
      b3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, b3, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, c3, 0, 0, 0);
    } else if (MODE == 2) {  // 4 x (32x32x2 + 4x4x1)
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, b0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, b1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a0, 0, 0, 0);
      b2 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, b2, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a1, 0, 0, 0);
      b3 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, b3, 0, 0, 0);
    } else if (MODE == 3) {  // 4 x 16x16x4
      b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, b0, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, b1, 0, 0, 0);
      b2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, b2, 0, 0, 0);
      b3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, b3, 0, 0, 0);
    } else {  // 8 x 4x4x1, one dependent chain
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, x, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x, x, b0, 0, 0, 0);
      b0 = __builtin_amdgcn_mfma_f32_4x4x1f32(y, y, b0, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  for (int r = 0; r < 4; ++r) s += b0[r] + b1[r] + b2[r] + b3[r] + c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* what, int per_iter, float* d, long long* c) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, c, 100, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, c, iters, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
  printf("%-34s %.3f ms, %.1f ns per instruction, s_memtime ticks/instr %.2f\n", what, ms,
         ms * 1e6 / ((double)iters * per_iter), (double)cyc / ((double)iters * per_iter));
}
int main() {
  float* d; long long* c;
  hipMalloc(&d, 256 * 256 * 4); hipMalloc(&c, 8);
  run<0>("32x32x2 (2 chains)", 4, d, c);
  run<1>("4x4x1 (8 independent)", 8, d, c);
  run<4>("4x4x1 (1 dependent chain)", 8, d, c);
  run<3>("16x16x4 (4 independent)", 4, d, c);
  run<2>("32x32x2 + 4x4x1 alternating (pair)", 4, d, c);
  return 0;
}
