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
// Every option is THREAD-LOCAL: mh_set_option changes what the CALLING host thread's later launches select and nothing
// else, so a test or probe that flips a kernel form cannot change what a generator running on another thread of the
// process launches (app.py:496 runs up to 10 on one model).  A new thread starts from the defaults below (environment
// variables are read once per thread).  The forms that exist only for A/B measurements (first-form attention kernels, the
// 128x128 bf16 GEMM, the wrong-by-design ablation builds of the production GEMM) are compiled only with -DMH_AB_BUILDS
// into libmidihip_ab.so (build.py); the production library refuses the option values that would select them.
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
thread_local int g_mh_gemm_variant = env_int("MH_GEMM", 1);
thread_local int g_mh_gemm_ablate = 0;
// "gemm_k64": 1 (default) = products of two row-major operands (every forward projection) run the K-step-64 main loop of
// gemm_pp256_kernel (whole-line LDS-DMA, r03) and so does the dgrad form (A row-major, B contraction-major), 2 = only the former,
// 0 = the K-step-32 loop everywhere (bit-identical results; A/B runs, MH_GEMM_K64)
thread_local int g_mh_gemm_k64 = env_int("MH_GEMM_K64", 1);
// "gemm_lean_epi": 1 (default) = interior tiles of the production kernel take the r06 forms of the plain and SwiGLU-backward
// epilogues (descriptor addressing, packed arithmetic), 0 = the general forms everywhere (bit-identical results; A/B runs)
thread_local int g_mh_gemm_lean_epi = env_int("MH_GEMM_LEAN_EPI", 1);
extern thread_local int g_skinny_mb, g_skinny_nbt;  // gemm_skinny.hip
extern thread_local int g_tokattn_bwd_batched;      // attention_small.hip
extern thread_local int g_attn_v3, g_attn_v3_wps, g_attn_passes;      // attention_mfma3.hip

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
  if (strcmp(name, "gemm_lean_epi") == 0) {
    g_mh_gemm_lean_epi = value;
    return 0;
  }
  if (strcmp(name, "tokattn_bwd_batched") == 0) {
    g_tokattn_bwd_batched = value;
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
  if (strcmp(name, "attn_passes") == 0) {  // work order of the event-level attention kernels: tile ranks in `value` chunks, light chunks last
    g_attn_passes = value;
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
  if (strcmp(name, "attn_passes") == 0) return g_attn_passes;
  if (strcmp(name, "skinny_nbt") == 0) return g_skinny_nbt;
  if (strcmp(name, "gemm_k64") == 0) return g_mh_gemm_k64;
  if (strcmp(name, "gemm_ablate") == 0) return g_mh_gemm_ablate;
  if (strcmp(name, "gemm_lean_epi") == 0) return g_mh_gemm_lean_epi;
  if (strcmp(name, "tokattn_bwd_batched") == 0) return g_tokattn_bwd_batched;
  return -1;
}

extern "C" const char* mh_last_error(void) { return g_err; }
extern "C" int mh_version(void) { return 1; }
// 1 when this library holds the A/B-only kernel forms (libmidihip_ab.so), 0 for the production library
extern "C" int mh_ab_builds(void) {
#ifdef MH_AB_BUILDS
  return 1;
#else
  return 0;
#endif
}
