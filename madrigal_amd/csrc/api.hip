// Error plumbing and build identification of libmadrigal_hip.so.
#include "mdg_common.h"
#include <stdlib.h>

namespace {
thread_local char g_err[512] = "";
}

void mdg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Tuning switches (MDG_* environment variables: experiment knobs that change speed only, never results) are read ONCE per
// process and per switch, not per launch; mdg_tuning_reload() makes every switch re-read the environment at its next use
// (tests and the variant-timing scripts flip a switch between two launches).  Diagnostics only; not thread-safe against launches.
std::atomic<int> g_mdg_env_generation{0};
int MdgEnvInt::get() {
  const int g = g_mdg_env_generation.load(std::memory_order_relaxed);
  if (gen != g) {
    const char* e = getenv(name);
    val = e ? atoi(e) : dflt;
    gen = g;
  }
  return val;
}
extern "C" void mdg_tuning_reload(void) { g_mdg_env_generation.fetch_add(1, std::memory_order_relaxed); }

extern "C" const char* mdg_last_error(void) { return g_err; }
extern "C" const char* mdg_build_arch(void) { return "gfx950"; }
extern "C" int mdg_abi_version(void) { return 9; }
