// Error plumbing and build identification of libmadrigal_hip.so.
#include "mdg_common.h"

namespace {
thread_local char g_err[512] = "";
}

void mdg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* mdg_last_error(void) { return g_err; }
extern "C" const char* mdg_build_arch(void) { return "gfx950"; }
extern "C" int mdg_abi_version(void) { return 3; }
