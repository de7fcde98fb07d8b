// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) with some lanes out of the descriptor's range -- does the hardware write
// zeros to LDS for those lanes (like a register load returns 0) or leave the LDS bytes alone?  And: is the destination
// M0 + lane * 16 for the dwordx4 form?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/lds_dma_oob.hip -o tools/probe/lds_dma_oob && tools/probe/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const unsigned long long p = (unsigned long long)src;
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)p);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32) & 0xffffu);
  rs[2] = __builtin_amdgcn_readfirstlane(nbytes);
  rs[3] = 0x00020000u;
  // lanes 0..47 in range (16 bytes each), 48..55 past the end, 56..63 offset 0xffffffff
  unsigned voff = lane < 56 ? lane * 16u : 0xffffffffu;
  const unsigned ldsbase = (unsigned)(size_t)(&lds[16]);  // base + 64 bytes
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(ldsbase), "s"(rs) : "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<unsigned> h(64 * 4);
  for (int i = 0; i < 64 * 4; ++i) h[i] = 0x1000 + i;
  unsigned *src, *out;
  hipMalloc(&src, 64 * 16);
  hipMalloc(&out, (64 * 4 + 64) * 4);
  hipMemcpy(src, h.data(), 64 * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, 48 * 16, out);
  std::vector<unsigned> r(64 * 4 + 64);
  hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
  printf("pad before: %08x %08x\n", r[0], r[15]);
  for (int l : {0, 1, 47, 48, 55, 56, 63}) printf("lane %2d: %08x %08x %08x %08x\n", l, r[16 + 4 * l], r[17 + 4 * l], r[18 + 4 * l], r[19 + 4 * l]);
  printf("pad after: %08x %08x\n", r[16 + 256], r[16 + 256 + 47]);
  return 0;
}
