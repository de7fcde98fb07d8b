// libsrx_hip.so: version / error plumbing of the C ABI (include/srx.h).
#include "srx_common.h"
#include <cstdarg>
#include <cstdio>
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
