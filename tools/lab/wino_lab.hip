// Developer lab (not part of the product, not built by build()): csrc/wino.hip's kernel with ablation switches and in-kernel
// stamps, timed on the VGG19 layer shapes of the SRGAN step.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o gpurun_out/wino_lab tools/lab/wino_lab.hip && gpurun_out/wino_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t srx_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 srx_bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ int srx_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
static inline int64_t srx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#include "wino_base.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { int n, hw, cin, cout; };

static WinoArgs make_args(const Shape& s, const float* x, const float* u, const float* bias, float* y, int bn) {
  WinoArgs a{};
  a.in = x; a.upk = u; a.bias = bias; a.out = y;
  a.N = s.n; a.H = s.hw; a.W = s.hw; a.Cin = s.cin; a.Cout = s.cout;
  a.TH = s.hw / 2; a.TW = s.hw / 2; a.T = s.n * a.TH * a.TW;
  a.tblocks = (int)srx_cdiv(a.T, WT);
  a.nch = s.cin / WKC;
  a.relu = 1; a.zsplit = 1; a.ncb = s.cout / bn;
  a.in_bytes = (unsigned)((size_t)s.n * s.hw * s.hw * s.cin * 4);
  a.upk_bytes = (unsigned)((size_t)16 * s.cin * s.cout * 4);
  a.out_elems = (size_t)s.n * s.hw * s.hw * s.cout;
  a.full = a.tblocks * a.ncb; a.tsplit = 1;
  return a;
}

template <typename K>
static float time_kernel(K launch, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 40; ++i) launch();  // (a few ms: the clock settles)
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f / reps);
  }
  std::sort(ts.begin(), ts.end());
  CK(hipGetLastError());
  return ts[ts.size() / 2];
}

template <int BN, int ABL>
static float run_base(const WinoArgs& a) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<BN, ABL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); once = true; }
  return time_kernel([&] { hipLaunchKernelGGL((wino_kernel<BN, ABL, false>), dim3(a.full), dim3(512), WINO_LDS, 0, a); });
}

#ifdef WITH_V3
#include "wino_v3.inc"
#endif

static void stamp_report(const WinoArgs& a0, unsigned long long* dst, int bn) {
  WinoArgs a = a0; a.stamps = dst;
  const int g = a.full;
  CK(hipMemset(dst, 0, (size_t)g * 64));
  if (bn == 64) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<64, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wino_kernel<64, 0, true>), dim3(g), dim3(512), WINO_LDS, 0, a);
  } else {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_kernel<32, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((wino_kernel<32, 0, true>), dim3(g), dim3(512), WINO_LDS, 0, a);
  }
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)g * 8);
  CK(hipMemcpy(h.data(), dst, (size_t)g * 64, hipMemcpyDeviceToHost));
  // per phase: median over workgroups of (stamp[i+1] - stamp[i]) in cycles; clock = d(memtime) / d(memrealtime) x 100 MHz
  std::vector<double> ph[4], clk;
  unsigned long long t_first = ~0ull, t_last = 0;
  for (int b = 0; b < g; ++b) {
    const unsigned long long* s = &h[(size_t)b * 8];
    for (int i = 0; i < 4; ++i) ph[i].push_back((double)(s[i + 1] - s[i]));
    if (s[7] > s[6]) clk.push_back((double)(s[4] - s[0]) / (double)(s[7] - s[6]) * 100.0);
    t_first = std::min(t_first, s[6]); t_last = std::max(t_last, s[7]);
  }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
  const double mhz = med(clk);
  printf("    stamps (median over %d workgroups, clock %.0f MHz): prologue %.2f us, chunk loop %.2f us (%.2f us/chunk), M exchange %.2f us, "
         "output transform + stores %.2f us; first start -> last end %.1f us\n", g, mhz, med(ph[0]) / mhz, med(ph[1]) / mhz,
         med(ph[1]) / mhz / a.nch, med(ph[2]) / mhz, med(ph[3]) / mhz, (double)(t_last - t_first) / 100.0);
}

int main(int argc, char** argv) {
  const Shape shapes[] = {{32, 96, 64, 64}, {32, 48, 64, 128}, {32, 48, 128, 128}, {32, 24, 128, 256}, {32, 24, 256, 256},
                          {32, 12, 256, 512}, {32, 12, 512, 512}, {32, 6, 512, 512}};
  const char* only = argc > 1 ? argv[1] : "";
  for (const Shape& s : shapes) {
    char tag[64]; snprintf(tag, sizeof(tag), "%dx%d_%d_%d", s.hw, s.hw, s.cin, s.cout);
    if (only[0] && !strstr(tag, only)) continue;
    const size_t nin = (size_t)s.n * s.hw * s.hw * s.cin, nout = (size_t)s.n * s.hw * s.hw * s.cout, nu = (size_t)16 * s.cin * s.cout;
    std::vector<float> hx(nin), hu(nu), hb(s.cout);
    uint32_t r = 12345;
    auto rnd = [&] { r = r * 1664525u + 1013904223u; return (float)((r >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd() + 0.5f;
    for (auto& v : hu) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd() * 0.1f;
    float *x, *u, *b, *y, *y2; unsigned long long* st;
    CK(hipMalloc(&x, nin * 4)); CK(hipMalloc(&u, nu * 4)); CK(hipMalloc(&b, s.cout * 4)); CK(hipMalloc(&y, nout * 4)); CK(hipMalloc(&y2, nout * 4));
    CK(hipMalloc(&st, (size_t)1 << 22));
    CK(hipMemcpy(x, hx.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(u, hu.data(), nu * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), s.cout * 4, hipMemcpyHostToDevice));
    const WinoArgs a = make_args(s, x, u, b, y, 64);
    const double gf = 2.0 * s.n * s.hw * s.hw * (double)s.cout * 9.0 * s.cin / 1e9, exgf = gf * 16.0 / 36.0;
    const double rounds = (double)a.full / 256.0;
    printf("== %s: %d tile blocks x %d channel blocks = %d workgroups (%.2f rounds of 256), %d chunks, %.2f GF direct / %.2f GF executed\n", tag,
           a.tblocks, a.ncb, a.full, rounds, a.nch, gf, exgf);
    struct { const char* name; float us; } res[] = {
        {"baseline BN=64", run_base<64, 0>(a)},
            };
    for (auto& e : res) printf("  %-66s %8.1f us  %6.1f TF/s executed (%.3f of 157.3)\n", e.name, e.us, exgf / e.us * 1e3, exgf / e.us * 1e3 / 157.3);
    stamp_report(a, st, 64);
#ifdef WITH_V3
    v3_report(s, x, u, b, y, y2, st, exgf);
#endif
    hipFree(x); hipFree(u); hipFree(b); hipFree(y); hipFree(y2); hipFree(st);
  }
  return 0;
}
