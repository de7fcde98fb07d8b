// Probe: does the raw-buffer range check on gfx950 include the scalar offset?  It does (MI355X, round 6): with a 256-byte descriptor
// and soffset 128 lanes 0..31 read floats 32..63 and lanes 32..63 read 0; with soffset 256 every lane reads 0.  Kernels may therefore
// put wave-uniform row / tap / split offsets into the instruction's scalar offset and still rely on "past the tensor reads 0".
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, float* out, unsigned bytes, unsigned soff) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, (int)bytes, 0x00020000);
  unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, (int)(threadIdx.x * 4), (int)soff, 0);
  out[threadIdx.x] = __builtin_bit_cast(float, v);
}
int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  (void)hipMalloc(&d, 1024); (void)hipMalloc(&o, 256);
  (void)hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  // descriptor covers the first 64 floats (256 bytes); lanes read float (soff/4 + lane)
  for (unsigned soff : {0u, 128u, 256u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 256u, soff);
    float r[64];
    (void)hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
    printf("soffset %3u: lane0 %.0f lane31 %.0f lane32 %.0f lane63 %.0f\n", soff, r[0], r[31], r[32], r[63]);
  }
  return 0;
}
