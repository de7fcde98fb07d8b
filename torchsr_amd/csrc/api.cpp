// libsrx_hip.so: version / error plumbing of the C ABI (include/srx.h).
#include "srx_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

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
