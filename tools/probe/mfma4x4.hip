// Probe the lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950: D_b[i][j] += A_b[i] * B_b[j] for 16 blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, c, 0, 0, 0);   // reveals the A source lane
  d = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, d, 0, 0, 0);   // reveals the B source lane
  for (int r = 0; r < 4; ++r) { out[l * 8 + r] = c[r]; out[l * 8 + 4 + r] = d[r]; }
}
int main() {
  float* d; hipMalloc(&d, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
  for (int l = 0; l < 12; ++l) {
    printf("lane %2d: A lanes [%g %g %g %g]  B lanes [%g %g %g %g]\n", l, h[l*8], h[l*8+1], h[l*8+2], h[l*8+3], h[l*8+4], h[l*8+5], h[l*8+6], h[l*8+7]);
  }
  return 0;
}
