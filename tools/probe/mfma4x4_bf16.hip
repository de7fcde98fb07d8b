// Probe the lane / k layout of v_mfma_f32_4x4x4_16b_bf16 on gfx950 (16 blocks of 4x4x4):
//   expected D[reg i] on lane l += sum_k A(lane 4*(l/4)+i)[k] * B(lane l)[k]
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ short bf(float v) { return (short)(__builtin_bit_cast(unsigned, v) >> 16); }  // exact for small integers
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, d = {0.f, 0.f, 0.f, 0.f}, e = {0.f, 0.f, 0.f, 0.f};
  const s16x4 ones = {bf(1.f), bf(1.f), bf(1.f), bf(1.f)};
  const s16x4 lanev = {bf((float)l), 0, 0, 0};
  const s16x4 kv = {bf(1.f), bf(10.f), bf(100.f), bf(1000.f)};          // k-position weights
  const s16x4 sel = {(short)(l % 4 == 0 ? bf(1.f) : 0), (short)(l % 4 == 1 ? bf(1.f) : 0), (short)(l % 4 == 2 ? bf(1.f) : 0), (short)(l % 4 == 3 ? bf(1.f) : 0)};
  c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(lanev, ones, c, 0, 0, 0);   // A source lane of register i
  d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, lanev, d, 0, 0, 0);   // B source lane
  e = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(sel, kv, e, 0, 0, 0);       // A lane (4b+i) has a one at k = i: register i = kv[i] if k pairs with k
  for (int r = 0; r < 4; ++r) { out[l * 12 + r] = c[r]; out[l * 12 + 4 + r] = d[r]; out[l * 12 + 8 + r] = e[r]; }
}
int main() {
  float* d; hipMalloc(&d, 64 * 12 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[768]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 10; ++l)
    printf("lane %2d: A lanes [%g %g %g %g]  B lanes [%g %g %g %g]  k-pairing [%g %g %g %g]\n", l, h[l*12], h[l*12+1], h[l*12+2], h[l*12+3],
           h[l*12+4], h[l*12+5], h[l*12+6], h[l*12+7], h[l*12+8], h[l*12+9], h[l*12+10], h[l*12+11]);
  return 0;
}
