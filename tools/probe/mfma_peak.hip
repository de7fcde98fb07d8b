// Probe: sustained fp32 MFMA rate (v_mfma_f32_32x32x2_f32) with operands in registers, random-ish data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  f32x16 a0, a1, a2, a3;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
  float x = seed + threadIdx.x * 0.37f, y = seed * 1.3f - threadIdx.x * 0.11f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    x = x * 0.999f + 0.001f; y = y * 1.001f - 0.002f;
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int blocks = 256 * wpc, iters = 20000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1000, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 5.0 * blocks * 4 /*waves*/ * (double)iters * 4 * 4096.0;
    printf("blocks/CU %d: %.1f TFLOP/s fp32 MFMA (%.2f ms)\n", wpc, flops / (ms * 1e-3) / 1e12, ms);
  }
  return 0;
}
