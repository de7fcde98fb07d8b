// Probe: operand layout and issue rate of v_mfma_f32_32x32x16_bf16 on gfx950.
//   expectation: A[i = l%32][k = 8(l/32) .. +7], B[k = 8(l/32) .. +7][j = l%32],
//                D: col = l%32, row = (r&3) + 8(r>>2) + 4(l/32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ void layout(const float* A /*[32][16]*/, const float* B /*[16][32]*/, float* D /*[32][32]*/) {
  const int l = threadIdx.x, i = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i * 16 + 8 * h + e]; b[e] = (__bf16)B[(8 * h + e) * 32 + i]; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}
__global__ __launch_bounds__(256) void rate(float* out, long long* cyc, int iters) {
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  bf16x8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.5f + threadIdx.x * 0.01f + e); y[e] = (__bf16)(1.5f - e * 0.1f); }
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, y, a1, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  (void)hipMalloc(&dA, 2048); (void)hipMalloc(&dB, 2048); (void)hipMalloc(&dD, 4096);
  (void)hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int k = 0; k < 16; ++k) s += hA[i * 16 + k] * hB[k * 32 + j];
    err = fmax(err, fabs(s - hD[i * 32 + j]));
  }
  printf("layout check: max |D - A*B| = %g (0 means the assumed operand layout is right)\n", err);
  float* o; long long* c; (void)hipMalloc(&o, 256 * 256 * 4); (void)hipMalloc(&c, 8);
  hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, o, c, 100);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(rate, dim3(256), dim3(256), 0, 0, o, c, 20000);
  hipDeviceSynchronize();
  long long cyc; (void)hipMemcpy(&cyc, c, 8, hipMemcpyDeviceToHost);
  printf("32x32x16 bf16: %.2f s_memtime ticks per instruction\n", (double)cyc / (20000.0 * 4));
  return 0;
}
