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

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 1; }
