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
// "gemm": 1 = the production bf16 kernel (gemm_pp256.hip: 256x256 tile, 4-stage LDS-DMA ring, ping-pong wave groups),
// 0 = the first structure (gemm.hip: 128x128, 2 stages), which also serves fp32; kept as an independent check of the
// production kernel in the GPU tests.  Initial value from the environment variable MH_GEMM (default 1).
// "gemm_ablate": selects the micro-benchmark builds of the production kernel (wrong results; tools/bench_gemm.py).
#include <stdlib.h>
#include <string.h>
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
int g_mh_gemm_variant = env_int("MH_GEMM", 1);
int g_mh_gemm_ablate = 0;
// "gemm_k64": 1 (default) = products of two row-major operands (every forward projection) run the K-step-64 main loop of
// gemm_pp256_kernel (whole-line LDS-DMA, r03) and so does the dgrad form (A row-major, B contraction-major), 2 = only the former,
// 0 = the K-step-32 loop everywhere (bit-identical results; A/B runs, MH_GEMM_K64)
int g_mh_gemm_k64 = env_int("MH_GEMM_K64", 1);
extern int g_skinny_mb, g_skinny_nbt;  // gemm_skinny.hip
extern int g_attn_v3, g_attn_v3_wps;      // attention_mfma3.hip

extern "C" int mh_set_option(const char* name, int value) {
  if (strcmp(name, "gemm") == 0) {
    g_mh_gemm_variant = value;
    return 0;
  }
  if (strcmp(name, "gemm_ablate") == 0) {
    g_mh_gemm_ablate = value;
    return 0;
  }
  if (strcmp(name, "gemm_k64") == 0) {
    g_mh_gemm_k64 = value;
    return 0;
  }
  if (strcmp(name, "skinny_mb") == 0) {  // 16-row blocks of the activation per workgroup of mh_gemm_skinny (0 = default)
    g_skinny_mb = value;
    return 0;
  }
  if (strcmp(name, "attn_v3") == 0) {  // third form of the event-level attention kernels: bit 0 forward, bit 1 dQ, bit 2 dK/dV,
                                       // bit 3 transpose reads in the backward pair (no transposed copies; needs bits 1 and 2),
                                       // bit 4 in the forward, bit 5: callers use mh_attn_bwd_o (delta inside the dQ kernel), bit 6: three K/V stages in the forward
    g_attn_v3 = value;
    return 0;
  }
  if (strcmp(name, "attn_v3_wps") == 0) {  // its register budget in waves per SIMD (0 = default; A/B runs)
    g_attn_v3_wps = value;
    return 0;
  }
  if (strcmp(name, "skinny_nbt") == 0) {  // 16-column blocks per workgroup of its plain form (0 = default)
    g_skinny_nbt = value;
    return 0;
  }
  mh_set_error("unknown option %s", name);
  return MH_ERR_ARG;
}
extern "C" int mh_get_option(const char* name) {
  if (strcmp(name, "gemm") == 0) return g_mh_gemm_variant;
  if (strcmp(name, "skinny_mb") == 0) return g_skinny_mb;
  if (strcmp(name, "attn_v3") == 0) return g_attn_v3;
  if (strcmp(name, "attn_v3_wps") == 0) return g_attn_v3_wps;
  if (strcmp(name, "skinny_nbt") == 0) return g_skinny_nbt;
  if (strcmp(name, "gemm_k64") == 0) return g_mh_gemm_k64;
  if (strcmp(name, "gemm_ablate") == 0) return g_mh_gemm_ablate;
  return -1;
}

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 1; }
