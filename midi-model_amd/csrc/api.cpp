// Host-side plumbing shared by every entry point of libmidihip.so: thread-local error text + version.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/midihip.h"

static thread_local char g_err[512] = "";

void mh_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- runtime options -------------------------------------------------------------------------------------
// "gemm": 0 = first structure (128x128, 2 LDS stages), 1 = pipelined structure (256x128, 3 stages, counted
// vmcnt; bf16 only).  Initial value from the environment variable MH_GEMM (default 0: the pipelined kernel is correct but measured slower, profiles/r01_run3).
#include <stdlib.h>
#include <string.h>
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
int g_mh_gemm_variant = env_int("MH_GEMM", 3);
int g_mh_gemm_ablate = 0;  // micro-benchmark only: bit0 = no tile loads after the first, bit1 = no LDS reads / MFMA

extern "C" int mh_set_option(const char* name, int value) {
  if (strcmp(name, "gemm") == 0) {
    g_mh_gemm_variant = value;
    return 0;
  }
  if (strcmp(name, "gemm_ablate") == 0) {
    g_mh_gemm_ablate = value;
    return 0;
  }
  mh_set_error("unknown option %s", name);
  return MH_ERR_ARG;
}
extern "C" int mh_get_option(const char* name) {
  if (strcmp(name, "gemm") == 0) return g_mh_gemm_variant;
  return -1;
}

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 1; }
