// libsrx_hip.so: version / error plumbing of the C ABI (include/srx.h).
#include "srx_common.h"
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void srx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int srx_version(void) { return SRX_VERSION; }

// The sha256 of the sources this library was compiled from (torchsr_amd/_lib.py: source_digest), so that a stale binary
// lying next to newer sources can be told from a fresh one without trusting file times.
#ifndef SRX_SOURCES_SHA256
#define SRX_SOURCES_SHA256 "unknown"
#endif
static const char g_build_info[] = "srx-build-sources-sha256:" SRX_SOURCES_SHA256;
extern "C" int srx_build_info(char* buf, size_t n) {
  if (!buf || n == 0) return SRX_E_BADARG;
  strncpy(buf, g_build_info + sizeof("srx-build-sources-sha256:") - 1, n - 1);
  buf[n - 1] = 0;
  return SRX_OK;
}

// Developer switches (A/B runs, tile experiments): read ONCE, when the library is loaded -- never on a launch path.
static SrxDevSwitches read_switches() {
  SrxDevSwitches s{};
  auto flag = [](const char* n) { const char* e = getenv(n); return e != nullptr && e[0] != 0 && !(e[0] == '0' && e[1] == 0); };
  auto num = [](const char* n) { const char* e = getenv(n); return e ? atoi(e) : 0; };
  s.no_rt36 = flag("SRX_NO_RT36");
  s.no_wgrad_rows = flag("SRX_NO_WGRAD_ROWS");
  s.no_bn_bwd_fuse = flag("SRX_NO_BN_BWD_FUSE");
  s.no_bn_fwd_fuse = flag("SRX_NO_BN_FWD_FUSE");
  s.no_first3 = flag("SRX_NO_FIRST3");
  s.no_c64 = flag("SRX_NO_C64");
  s.no_wgrad_dma = flag("SRX_NO_WGRAD_DMA");
  s.no_wgrad_lin = flag("SRX_NO_WGRAD_LIN");
  s.wgrad_nsplit = num("SRX_WGRAD_NSPLIT");
  s.wgrad_rows_nsplit = num("SRX_WGRAD_ROWS_NSPLIT");
  s.first3_wgs_per_cu = num("SRX_FIRST3_WGS_PER_CU");
  s.thin_fwd_rows = num("SRX_THIN_FWD_ROWS");
  s.reserved_cus = num("SRX_RESERVED_CUS");
  s.c64_ablate = num("SRX_C64_ABLATE");
  s.rdb_ablate = num("SRX_RDB_ABLATE");
  s.no_wino = flag("SRX_NO_WINO");
  s.old_wgrad_reduce = flag("SRX_OLD_WGRAD_REDUCE");
  s.wino_no_tail = flag("SRX_WINO_NO_TAIL");
  s.wino_zsplit = num("SRX_WINO_ZSPLIT");
  s.wino_bn = num("SRX_WINO_BN");
  s.s2_mode = num("SRX_S2_MODE");
  // Result-CHANGING switches (the ablation instantiations compute garbage by design: they exist to time a kernel without one of
  // its parts) need a second, explicit opt-in and say so once on stderr -- a stray variable must not silently corrupt a run
  if (s.rdb_ablate != 0 || s.c64_ablate != 0) {
    const char* ok = getenv("SRX_ALLOW_GARBAGE_RESULTS");
    if (ok && ok[0] == '1') {
      fprintf(stderr, "libsrx_hip: SRX_RDB_ABLATE=%d SRX_C64_ABLATE=%d with SRX_ALLOW_GARBAGE_RESULTS=1: the fused dense-block / bf16 trunk "
                      "kernels of this process compute GARBAGE (timing ablations)\n", s.rdb_ablate, s.c64_ablate);
    } else {
      fprintf(stderr, "libsrx_hip: SRX_RDB_ABLATE / SRX_C64_ABLATE ignored: ablation kernels compute garbage and need "
                      "SRX_ALLOW_GARBAGE_RESULTS=1 as well\n");
      s.rdb_ablate = 0; s.c64_ablate = 0;
    }
  }
  if (const char* f = getenv("SRX_FORCE_PLAN")) {
    int v[4] = {0, 0, 1, 1};
    if (sscanf(f, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) { s.force_plan = true; for (int i = 0; i < 4; ++i) s.plan[i] = v[i]; }
  }
  return s;
}
static SrxDevSwitches g_switches = read_switches();
const SrxDevSwitches& srx_dev() { return g_switches; }

extern "C" int srx_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return SRX_E_BADARG;
  strncpy(buf, g_err, n - 1);
  buf[n - 1] = 0;
  return SRX_OK;
}

extern "C" int srx_device_cus(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
  return cus;
}

// Compute units the launch plans may count on: the device's, less the ones the caller says something else holds while these
// kernels run (RCCL's channel workgroups during an overlapped gradient all-reduce).  srx_set_reserved_cus / SRX_RESERVED_CUS.
static std::atomic<int> g_reserved{-1};
extern "C" int srx_set_reserved_cus(int k) {
  SRX_REQUIRE(k >= 0 && k <= 128, "set_reserved_cus: 0..128");
  g_reserved.store(k);
  return SRX_OK;
}
extern "C" int srx_plan_cus(void) {
  static int cus = 0;
  if (cus <= 0) { cus = srx_device_cus(); if (cus <= 0) cus = 256; }
  int k = g_reserved.load();
  if (k < 0) k = srx_dev().reserved_cus > 0 ? srx_dev().reserved_cus : 0;
  return cus - k > 16 ? cus - k : 16;
}

// ---------------------------------------------------------------------------
// Per-launch timing of the convolution kernels: the ONE kernel named in the record (not its fix-up /
// reduce companions) is dispatched with hipExtLaunchKernelGGL and its own start / stop HIP events, i.e.
// the timestamps of that dispatch on its stream -- events recorded as separate packets around a launch
// add ~7 us of command-processor time to a 50 us kernel.  bench.py's roofline leg reads these; off by
// default and never on inside a hipGraph capture.
// ---------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t e0, e1; char name[112]; double flops; };
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
int g_prof_n = 0;
bool g_prof_on = false;
}  // namespace

bool srx_prof_on() { return g_prof_on; }

bool srx_prof_take(const char* name, double flops, hipEvent_t* e0, hipEvent_t* e1) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on || g_prof_n >= (int)g_prof.size()) return false;
  ProfRec& r = g_prof[g_prof_n++];
  strncpy(r.name, name, sizeof(r.name) - 1);
  r.name[sizeof(r.name) - 1] = 0;
  r.flops = flops;
  *e0 = r.e0;
  *e1 = r.e1;
  return true;
}

extern "C" int srx_prof_start(int max_launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  SRX_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), "prof_start: bad capacity");
  while ((int)g_prof.size() < max_launches) {
    ProfRec r{};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess)
      SRX_FAIL(SRX_E_HIP, "prof_start: hipEventCreate failed");
    g_prof.push_back(r);
  }
  g_prof_n = 0;
  g_prof_on = true;
  return SRX_OK;
}

extern "C" int srx_prof_stop(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = false;
  return g_prof_n;
}

extern "C" int srx_prof_get(int i, char* name, size_t n, float* ms, double* flops) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  SRX_REQUIRE(!g_prof_on && i >= 0 && i < g_prof_n && name && n > 0 && ms && flops, "prof_get: bad argument");
  ProfRec& r = g_prof[i];
  if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess)
    SRX_FAIL(SRX_E_HIP, "prof_get: event query failed");
  strncpy(name, r.name, n - 1);
  name[n - 1] = 0;
  *flops = r.flops;
  return SRX_OK;
}

// ---------------------------------------------------------------------------
// srx_occupy_cus: stand-in for the CUs a collective's channel kernels hold while the backward pass runs.
// ---------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(const int* stop_flag, long long max_ticks) {
  extern __shared__ char hold[];  // (claimed, never touched: keeps other workgroups off this CU when the launch asks for it)
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();  // 100 MHz
  // bounded spin: the flag or the deadline, whichever comes first; the load bypasses the caches (host-visible memory)
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < max_ticks) {
    if (__hip_atomic_load(stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
    __builtin_amdgcn_s_sleep(64);
  }
  (void)hold;
}
}  // namespace

extern "C" int srx_occupy_cus(int k, int whole_cu, const int* stop_flag, int max_ms, void* stream) {
  SRX_REQUIRE(k > 0 && k <= 128 && stop_flag && max_ms > 0 && max_ms <= 60000, "occupy_cus: 1..128 CUs, a flag, 1..60000 ms");
  const size_t lds = whole_cu ? 160 * 1024 : 0;
  static std::once_flag once;
  std::call_once(once, [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  hipLaunchKernelGGL(occupy_kernel, dim3((unsigned)k), dim3(256), lds, srx_stream(stream), stop_flag, (long long)max_ms * 100000LL);
  SRX_CHECK_LAUNCH("occupy_kernel");
  return SRX_OK;
}
