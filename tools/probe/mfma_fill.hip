// Probe: what one filler instruction costs a wave that otherwise issues v_mfma_f32_32x32x2_f32 back to back (one wave per SIMD).
// FILL: 0 v_fma_f32, 1 v_add_f32, 2 v_pk_add_f32, 3 v_pk_fma_f32, 4 v_mov_b32, 5 v_cndmask, 6 v_mul_lo_u32, 7 ds_write_b64, 8 ds_read_b128,
//       9 buffer_load_dwordx2 (L1/L2 hit), 10 s_add (SALU), 11 v_max_f32, 12 ds_write_b128, 13 v_cvt_pk_bf16_f32, 14 bf16 MFMA host (32x32x16) with v_fma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int FILL, int K>
__global__ __launch_bounds__(256) void mixed(float* out, const float* in, int iters, float seed) {
  __shared__ f32x4 lds[1024];
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  float x = seed + threadIdx.x * 0.37f, y = seed * 1.3f - threadIdx.x * 0.11f;
  float v[8]; f32x2 p[4]; f32x4 q[4]; int si = 0;
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  for (int i = 0; i < 4; ++i) { p[i] = f32x2{x, y}; q[i] = f32x4{x, y, x, y}; }
  lds[threadIdx.x] = q[0];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), (short)0, 1 << 20, 0x00020000);
  bf16x8 bx, by; for (int i = 0; i < 8; ++i) { bx[i] = (short)(threadIdx.x + i); by[i] = (short)(3 * threadIdx.x + i); }
  __syncthreads();
  auto fill = [&](int k) {
    if constexpr (FILL == 0 || FILL == 14) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[k % 8]) : "v"(x), "v"(y));
    if constexpr (FILL == 1) asm volatile("v_add_f32 %0, %1, %0" : "+v"(v[k % 8]) : "v"(x));
    if constexpr (FILL == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[k % 4]) : "v"(p[(k + 1) % 4]));
    if constexpr (FILL == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[k % 4]) : "v"(p[(k + 1) % 4]));
    if constexpr (FILL == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k % 8]) : "v"(x));
    if constexpr (FILL == 5) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[k % 8]) : "v"(x), "v"(y));
    if constexpr (FILL == 6) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(v[k % 8]) : "v"(x), "v"(y));
    if constexpr (FILL == 7) asm volatile("ds_write_b64 %0, %1" ::"v"((threadIdx.x & 63) * 8 + (k % 4) * 1024), "v"(p[k % 4]));
    if constexpr (FILL == 12) asm volatile("ds_write_b128 %0, %1" ::"v"((threadIdx.x & 63) * 16 + (k % 4) * 2048), "v"(q[k % 4]));
    if constexpr (FILL == 8) { asm volatile("ds_read_b128 %0, %1" : "=v"(q[k % 4]) : "v"((threadIdx.x & 63) * 16 + (k % 4) * 2048)); }
    if constexpr (FILL == 9) { asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(p[k % 4]) : "v"((threadIdx.x & 63) * 8 + (k % 4) * 1024), "s"(rs)); }
    if constexpr (FILL == 10) asm volatile("s_add_u32 %0, %0, 1" : "+s"(si) : : "scc");  // (SCC is the loop branch's condition: without the clobber the probe never ends)
    if constexpr (FILL == 11) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[k % 8]) : "v"(x));
    if constexpr (FILL == 13) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(v[k % 8]) : "v"(x), "v"(y));
  };
  for (int i = 0; i < iters; ++i) {
    if constexpr (FILL == 14) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bx, by, a0, 0, 0, 0);
    else a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) fill(k);
    if constexpr (FILL == 14) a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(by, bx, a1, 0, 0, 0);
    else a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; ++k) fill(K + k);
    if constexpr (FILL == 8 || FILL == 9) { if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  float s = si;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + q[i][0] + q[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[(threadIdx.x + 1) & 1023][0];
}
template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
  float *d, *in; hipMalloc(&d, 4096 * 256 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 0, 1 << 20);
  const int iters = 20000;
  const char* names[] = {"v_fma_f32", "v_add_f32", "v_pk_add_f32", "v_pk_fma_f32", "v_mov_b32", "v_cndmask_b32", "v_mul_lo_u32", "ds_write_b64", "ds_read_b128",
                         "buffer_load_dwordx2", "s_add_u32", "v_max_f32", "ds_write_b128", "v_cvt_pk_bf16_f32", "v_fma_f32 beside bf16 32x32x16 MFMA"};
  { float w = timeit([&] { hipLaunchKernelGGL((mixed<0, 0>), dim3(256), dim3(256), 0, 0, d, in, iters, 0.5f); }); (void)w; }
#define RUN(F, K) { float ms = timeit([&] { hipLaunchKernelGGL((mixed<F, K>), dim3(256), dim3(256), 0, 0, d, in, iters, 0.5f); }); \
    const double c = ms * 1e-3 * 2.4e9 / (2.0 * iters); base[K == 0 ? 0 : 1] = K == 0 ? c : base[1]; \
    if (K == 0) b0 = c; else printf("%-36s %2d per MFMA: %6.1f clocks per MFMA slot (bare %.1f) -> %.2f clocks each\n", names[F], K, c, b0, (c - b0) / K); fflush(stdout); }
  double base[2] = {0, 0}, b0 = 0;
  RUN(0, 0) RUN(0, 8) RUN(1, 8) RUN(2, 8) RUN(3, 8) RUN(4, 8) RUN(11, 8) RUN(13, 8) RUN(10, 8)
  RUN(14, 0) RUN(14, 4) RUN(14, 8)
  return 0;
}
