/* A plain C99 host of the C-ABI (include/midihip.h): what a non-Python maintainer's binding starts from.  Built and run by
 * tests/test_abi_and_ddp.py::test_header_is_plain_c_and_a_c_host_can_call_the_library (no GPU needed: only entry points that
 * enqueue nothing are called; every other declared symbol is looked up).  usage: host <libmidihip.so> <symbol>... */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "midihip.h"

typedef int (*fn_i_v)(void);
typedef const char* (*fn_s_v)(void);
typedef int (*fn_setopt)(const char*, int);
typedef int (*fn_getopt)(const char*);

/* dlsym returns an object pointer; ISO C has no cast from it to a function pointer, POSIX guarantees the round trip */
static void* sym(void* h, const char* name) { return dlsym(h, name); }

int main(int argc, char** argv) {
  void* h;
  fn_i_v version, ab;
  fn_s_v last_error;
  fn_setopt set_option;
  fn_getopt get_option;
  int i, rc;
  if (argc < 2) return 1;
  h = dlopen(argv[1], RTLD_NOW);
  if (!h) {
    printf("dlopen failed: %s\n", dlerror());
    return 2;
  }
  for (i = 2; i < argc; ++i)
    if (!sym(h, argv[i])) {
      printf("missing symbol %s\n", argv[i]);
      return 3;
    }
  *(void**)(&version) = sym(h, "mh_version");
  *(void**)(&ab) = sym(h, "mh_ab_builds");
  *(void**)(&last_error) = sym(h, "mh_last_error");
  *(void**)(&set_option) = sym(h, "mh_set_option");
  *(void**)(&get_option) = sym(h, "mh_get_option");
  if (version() < 1) return 4;
  rc = set_option("no_such_option", 1);
  if (rc != MH_ERR_ARG || strstr(last_error(), "no_such_option") == NULL) return 5;
  if (set_option("gemm_k64", 0) != MH_OK || get_option("gemm_k64") != 0) return 6;
  printf("C host ok: version %d, A/B library %d, %d symbols, error text '%s'\n", version(), ab(), argc - 2, last_error());
  return 0;
}
