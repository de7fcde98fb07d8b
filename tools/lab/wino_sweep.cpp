// Developer tool: every Winograd plan (BN, channel splits, tail parts) of every Winograd layer shape of the SRGAN step, timed through
// the C ABI (srx_wino_force_plan) -- the table wino_plan's cost constants are fitted from (tools/lab/fit_wino_plan.py).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/lab/wino_sweep tools/lab/wino_sweep.cpp -Ltorchsr_amd/csrc -lsrx_hip -Wl,-rpath,'$ORIGIN/../../torchsr_amd/csrc'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/srx.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
template <typename K> static float time_us(K launch, int reps = 10) {
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0) { CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); }
  for (int i = 0; i < 12; ++i) launch();
  CK(hipDeviceSynchronize());
  float best[3];
  for (int r = 0; r < 3; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best[r] = ms * 1e3f / reps;
  }
  std::sort(best, best + 3);
  return best[1];
}
struct Case { const char* what; int mode; int n, hw, cin, cout; int count; };  // mode 0 fwd (bias + ReLU), 1 data gradient (ReLU mask), 2 fwd + BatchNorm statistics
int main() {
  const Case cases[] = {
      {"vgg fwd", 0, 32, 96, 64, 64, 1}, {"vgg fwd", 0, 32, 48, 64, 128, 1}, {"vgg fwd", 0, 32, 48, 128, 128, 1}, {"vgg fwd", 0, 32, 24, 128, 256, 1},
      {"vgg fwd", 0, 32, 24, 256, 256, 3}, {"vgg fwd", 0, 32, 12, 256, 512, 1}, {"vgg fwd", 0, 32, 12, 512, 512, 3}, {"vgg fwd", 0, 32, 6, 512, 512, 4},
      {"vgg bwd", 1, 16, 96, 64, 64, 1}, {"vgg bwd", 1, 16, 48, 64, 128, 1}, {"vgg bwd", 1, 16, 48, 128, 128, 1}, {"vgg bwd", 1, 16, 24, 128, 256, 1},
      {"vgg bwd", 1, 16, 24, 256, 256, 3}, {"vgg bwd", 1, 16, 12, 256, 512, 1}, {"vgg bwd", 1, 16, 12, 512, 512, 3}, {"vgg bwd", 1, 16, 6, 512, 512, 4},
      {"D fwd+stats", 2, 32, 48, 64, 128, 1}, {"D fwd+stats", 2, 32, 24, 128, 256, 1}, {"D fwd+stats", 2, 32, 12, 256, 512, 1},
      {"D fwd+stats", 2, 16, 48, 64, 128, 1}, {"D fwd+stats", 2, 16, 24, 128, 256, 1}, {"D fwd+stats", 2, 16, 12, 256, 512, 1},
      {"D bwd", 1, 32, 48, 64, 128, 1}, {"D bwd", 1, 32, 24, 128, 256, 1}, {"D bwd", 1, 32, 12, 256, 512, 1},
      {"D bwd", 1, 16, 48, 64, 128, 1}, {"D bwd", 1, 16, 24, 128, 256, 1}, {"D bwd", 1, 16, 12, 256, 512, 1},
  };
  // a few ms of launches first: the clock settles
  for (const Case& c : cases) {
    srx_conv2d_t d{}; d.N = c.n; d.H = c.hw; d.W = c.hw; d.Cin = c.cin; d.Cin_s = c.cin; d.Cout = c.cout; d.Cout_s = c.cout; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1;
    d.act = c.mode == 0 ? SRX_ACT_RELU : SRX_ACT_NONE;
    const int kin = c.mode == 1 ? c.cout : c.cin, kout = c.mode == 1 ? c.cin : c.cout;  // the GEMM's contraction / output channels
    const size_t nx = (size_t)c.n * c.hw * c.hw * kin, ny = (size_t)c.n * c.hw * c.hw * kout, nu = (size_t)16 * c.cin * c.cout;
    std::vector<float> hx(nx), hu(nu), hb(kout), hm(ny);
    unsigned r = 777;
    auto rnd = [&] { r = r * 1664525u + 1013904223u; return (float)((r >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hx) v = rnd() + (c.mode == 1 ? 0.f : 0.5f);
    for (auto& v : hu) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd() * 0.1f;
    for (auto& v : hm) v = rnd();
    float *x, *u, *b, *y, *m, *stats, *ws;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&u, nu * 4)); CK(hipMalloc(&b, kout * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&m, ny * 4));
    CK(hipMalloc(&stats, (size_t)(srx_wino_stat_rows(&d) + 1) * c.cout * 2 * 4));
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(u, hu.data(), nu * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), kout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(m, hm.data(), ny * 4, hipMemcpyHostToDevice));
    const size_t ws_cap = (size_t)17 * ny;
    CK(hipMalloc(&ws, ws_cap * 4));
    const int T = c.n * (c.hw / 2) * (c.hw / 2), nch = kin / 32;
    auto run = [&] {
      const int which = c.mode == 1 ? 1 : (c.mode == 2 ? 2 : 0);
      const size_t nws = srx_wino_ws_floats(&d, which);
      if (nws > ws_cap) { printf("workspace too small\n"); exit(1); }
      int rc;
      if (c.mode == 0) rc = srx_wino_fwd(&d, x, u, b, y, ws, nws, nullptr);
      else if (c.mode == 1) rc = srx_wino_bwd_data(&d, x, u, m, y, ws, nws, nullptr);
      else rc = srx_wino_fwd_stats(&d, x, u, nullptr, y, stats, ws, nws, nullptr);
      if (rc != 0) { char msg[256]; srx_last_error(msg, sizeof(msg)); printf("launch failed: %s\n", msg); exit(1); }
    };
    int plan[6];
    srx_wino_force_plan(0, 0, 0);
    srx_wino_plan(&d, c.mode == 1 ? 1 : 0, plan);
    const float t_auto = time_us(run);
    printf("case %-12s N %2d %3dx%-3d K %4d -> %4d  T %6d nch %2d count %d | planner: BN %d zs %d ts %d wgs %d: %7.1f us\n", c.what, c.n, c.hw, c.hw, kin, kout, T, nch,
           c.count, plan[0], plan[1], plan[5], plan[2], t_auto);
    for (int bn = 64; bn >= 32; bn -= 32) {
      if (kout % bn) continue;
      const long wgs = (long)((T + 31) / 32) * (kout / bn);
      for (int zs = 1; zs <= nch && zs <= 8; ++zs) {
        if (c.mode == 2 && zs > 1) break;
        for (int ts = 1; ts <= 8 && ts <= nch; ++ts) {
          if (ts > 1 && (zs > 1 || wgs <= 256 || wgs % 256 == 0 || (wgs % 256) * ts > 512)) continue;
          if (srx_wino_force_plan(bn, zs, ts) != 0) continue;
          srx_wino_plan(&d, c.mode == 1 ? 1 : 0, plan);
          // (statistics launches report the plain forward's plan here; the launch itself refuses a split and falls back)
          const float t = time_us(run);
          printf("  plan BN %2d zs %d ts %d wgs %5ld (+tail %4ld x %d): %7.1f us\n", bn, zs, ts, wgs * zs, ts > 1 ? wgs % 256 : 0, ts, t);
          fflush(stdout);
        }
      }
    }
    srx_wino_force_plan(0, 0, 0);
    hipFree(x); hipFree(u); hipFree(b); hipFree(y); hipFree(m); hipFree(stats); hipFree(ws);
  }
  return 0;
}
