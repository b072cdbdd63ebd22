// VERDICT r03 item 2, measured directly: a chain of dependent decode-step projections -- X_{p+1}[64 x 1024] = X_p . W_p^T, bf16, the
// N = K = 1024 launch class of the token-level stack (60 launches x 4.9 us per generated event) -- run
//   (a) as one launch per phase, hipGraph-captured (what decode.py does today), and
//   (b) as ONE persistent launch: 256 workgroups (one 16-row x 16-column tile each and all of K, the tiling of mh_gemm_skinny for
//       this shape), phase p + 1 of a row block starting as soon as the 64 column tiles of ITS rows in phase p have published: the
//       tile is written with write-through (sc1) stores, drained (s_waitcnt vmcnt(0)), then an agent-scope arrive on the
//       (phase, row block) counter; consumers poll that counter from one lane (relaxed sc1 load + s_sleep) and read the rows with
//       sc1 loads (MI355X_MICROARCH.md, "valid forms": sc1 stores + drained flag / sc1 loads after the poll).  Every spin is
//       bounded: a protocol error shows up as a timeout count, never as a hang.
// Both forms run the SAME tile function on the same inputs, so the final activations must agree bit for bit; the probe reports the
// time per phase of each.  The weights cycle through NW_MATS matrices (10 MB: L2 / Infinity-Cache resident, as the token-level
// stack's 51 MB are across the eight token steps of an event).
// Build: hipcc --offload-arch=gfx950 -O3 tools/persist_probe.hip -o tools/bin/persist_probe -Lmidi-model_amd -lmidihip
//        -Wl,-rpath,'$ORIGIN/../../midi-model_amd' ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int M = 64, D = 1024, NWAVE = 8, NW_MATS = 5;

// one 16 x 16 output tile: rows rb*16.., columns ct*16.., all of K.  COH: the activations are another workgroup's output of
// THIS launch (sc1 loads / stores); otherwise plain accesses (launch boundaries make them visible).
template <bool COH>
__device__ inline void tile(const bf16* __restrict__ X, const bf16* __restrict__ W, bf16* __restrict__ Y, int rb, int ct,
                            float (*red)[16][17]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const bf16* xrow = X + (int64_t)(rb * 16 + fi) * D + fg * 8;
  const bf16* wrow = W + (int64_t)(ct * 16 + fi) * D + fg * 8;
  bf16x8 wf[4], xf[4];
  u32x4 xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (wave + NWAVE * i) * 32;
    wf[i] = *reinterpret_cast<const bf16x8*>(wrow + k);
    if (COH) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(xv[i]) : "v"(xrow + k) : "memory");
    else xf[i] = *reinterpret_cast<const bf16x8*>(xrow + k);
  }
  if (COH) {  // (the wait names the four destinations as its outputs: nothing may read them before it)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xv[0]), "+v"(xv[1]), "+v"(xv[2]), "+v"(xv[3])::"memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) xf[i] = *reinterpret_cast<bf16x8*>(&xv[i]);
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[i], acc, 0, 0, 0);
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][fi][4 * fg + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {  // 64 writers: row threadIdx.x >> 2, four columns each, one 8-byte store
    const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 4;
    union {
      bf16 h[4];
      uint64_t u;
    } o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NWAVE; ++w) t += red[w][r][c0 + j];
      o.h[j] = (bf16)t;
    }
    uint64_t* dst = reinterpret_cast<uint64_t*>(Y + (int64_t)(rb * 16 + r) * D + ct * 16 + c0);
    if (COH) __hip_atomic_store(dst, o.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dwordx2 ... sc1
    else *dst = o.u;
  }
  __syncthreads();
}

__global__ __launch_bounds__(512) void phase_kernel(const bf16* X, const bf16* W, bf16* Y) {
  __shared__ float red[NWAVE][16][17];
  tile<false>(X, W, Y, blockIdx.x >> 6, blockIdx.x & 63, red);
}

// X: [nphase + 1][M][D] activations (phase p reads slab p, writes slab p + 1); cnt: [nphase + 1][4] arrivals per (slab, row block)
// PROTO 0: sc1 payload stores, drained, relaxed arrive; relaxed poll, sc1 payload loads.  PROTO 1: plain payload stores, agent-scope
// RELEASE fence, relaxed arrive; relaxed poll, agent-scope ACQUIRE fence, plain loads.  SLEEP: s_sleep between polls.
template <int PROTO, bool SLEEP>
__global__ __launch_bounds__(512) void persistent_kernel(bf16* X, const bf16* W, int nphase, unsigned* cnt, unsigned* err,
                                                      unsigned long long* t_wait) {
  __shared__ float red[NWAVE][16][17];
  __shared__ int bail;
  const int rb = blockIdx.x >> 6, ct = blockIdx.x & 63;
  if (threadIdx.x == 0) bail = 0;
  __syncthreads();
  unsigned long long waited = 0;
  for (int p = 0; p < nphase; ++p) {
    if (p > 0) {  // the 64 tiles of this row block in slab p
      if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        unsigned spins = 0;
        while (__hip_atomic_load(&cnt[p * 4 + rb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 64u) {
          if (++spins > (1u << 22)) {
            atomicAdd(&err[0], 1u);
            bail = 1;
            break;
          }
          if (SLEEP) __builtin_amdgcn_s_sleep(1);
        }
        waited += __builtin_readcyclecounter() - t0;
        if (PROTO == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      if (bail) break;
    }
    tile<PROTO == 0>(X + (int64_t)p * M * D, W + (int64_t)(p % NW_MATS) * D * D, X + (int64_t)(p + 1) * M * D, rb, ct, red);
    if (PROTO == 0) {
      if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the writers' stores have left this CU
      __syncthreads();
    } else if (threadIdx.x == 0) {  // (tile() ends with a barrier: every writer's stores are issued)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&cnt[(p + 1) * 4 + rb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *t_wait = waited;
}

// the production projection through the C-ABI (link with -lmidihip): the same chain, for comparison with the bare tile above
extern "C" int mh_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* R,
                              int64_t ldr, int mode, float norm_eps, const int64_t* row_ids, const int64_t* res_ids, int64_t M,
                              int64_t N, int64_t K, int dtype, void* stream);

static float frand(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
  const int nphase = argc > 1 ? atoi(argv[1]) : 120;  // 8 token steps x 3 layers x 5 phases
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  std::vector<bf16> hx((size_t)M * D), hw((size_t)NW_MATS * D * D);
  uint32_t seed = 12345;
  for (auto& v : hx) v = (bf16)frand(seed);
  for (auto& v : hw) v = (bf16)(frand(seed) * 0.108f);  // keeps the activations' scale across phases (var 1/12 * 0.108^2 * 12 ~ 1/1024)
  bf16 *X, *X2, *W;
  unsigned *cnt, *err;
  unsigned long long* t_wait;
  const size_t slab = (size_t)M * D * sizeof(bf16);
  CK(hipMalloc(&X, slab * (nphase + 1)));
  CK(hipMalloc(&X2, slab * (nphase + 1)));
  CK(hipMalloc(&W, hw.size() * sizeof(bf16)));
  CK(hipMalloc(&cnt, (nphase + 1) * 4 * sizeof(unsigned)));
  CK(hipMalloc(&err, 16));
  CK(hipMalloc(&t_wait, 8));
  CK(hipMemcpy(W, hw.data(), hw.size() * sizeof(bf16), hipMemcpyHostToDevice));
  CK(hipMemset(X, 0, slab * (nphase + 1)));
  CK(hipMemset(X2, 0, slab * (nphase + 1)));
  CK(hipMemcpy(X, hx.data(), slab, hipMemcpyHostToDevice));
  CK(hipMemcpy(X2, hx.data(), slab, hipMemcpyHostToDevice));
  CK(hipMemset(err, 0, 16));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  // (a) one launch per phase, captured in a graph
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int p = 0; p < nphase; ++p)
    phase_kernel<<<256, 512, 0, st>>>(X2 + (size_t)p * M * D, W + (size_t)(p % NW_MATS) * D * D, X2 + (size_t)(p + 1) * M * D);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(exec, st));
  CK(hipStreamSynchronize(st));
  float ms_graph = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < ms_graph) ms_graph = ms;
  }

  // (a') the production kernel in the same harness: plain, with a residual, with the RMSNorm row scale
  float ms_prod[3] = {0, 0, 0};
  for (int v = 0; v < 3; ++v) {
    hipGraph_t g2;
    hipGraphExec_t ex2;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int p = 0; p < nphase; ++p) {
      const bf16* xin = X2 + (size_t)p * M * D;
      int rc = mh_gemm_skinny(xin, D, W + (size_t)(p % NW_MATS) * D * D, D, X2 + (size_t)(p + 1) * M * D, D, v == 1 ? xin : nullptr, D, 0,
                              v == 2 ? 1e-6f : 0.f, nullptr, nullptr, M, D, D, 1, st);
      if (rc != 0) {
        printf("mh_gemm_skinny failed\n");
        return 1;
      }
    }
    CK(hipStreamEndCapture(st, &g2));
    CK(hipGraphInstantiate(&ex2, g2, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ex2, st));
    CK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ex2, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    ms_prod[v] = best;
  }
  // restore the reference chain's activations (the production variants with residual / scale wrote other values)
  CK(hipGraphLaunch(exec, st));
  CK(hipStreamSynchronize(st));

  // (b) one persistent launch, three hand-off protocols
  struct Res { const char* what; float best, med; unsigned timeouts; size_t diff; unsigned long long wait; } res[3];
  for (int v = 0; v < 3; ++v) {
    std::vector<float> all;
    CK(hipMemset(err, 0, 16));
    for (int i = 0; i < reps + 3; ++i) {
      CK(hipMemsetAsync(cnt, 0, (nphase + 1) * 4 * sizeof(unsigned), st));
      CK(hipMemsetAsync(X + (size_t)M * D, 0, slab * nphase, st));
      CK(hipEventRecord(e0, st));
      if (v == 0) persistent_kernel<0, true><<<256, 512, 0, st>>>(X, W, nphase, cnt, err, t_wait);
      else if (v == 1) persistent_kernel<0, false><<<256, 512, 0, st>>>(X, W, nphase, cnt, err, t_wait);
      else persistent_kernel<1, true><<<256, 512, 0, st>>>(X, W, nphase, cnt, err, t_wait);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 3) all.push_back(ms);
    }
    std::sort(all.begin(), all.end());
    unsigned herr[4];
    CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
    unsigned long long hwait;
    CK(hipMemcpy(&hwait, t_wait, 8, hipMemcpyDeviceToHost));
    std::vector<bf16> a((size_t)M * D), b((size_t)M * D);
    CK(hipMemcpy(a.data(), X + (size_t)nphase * M * D, slab, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), X2 + (size_t)nphase * M * D, slab, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < a.size(); ++i) diff += memcmp(&a[i], &b[i], 2) != 0;
    res[v] = {v == 0 ? "sc1 stores + drain + arrive | poll (s_sleep) + sc1 loads" : v == 1 ? "the same, polling without s_sleep"
                     : "plain stores + release fence + arrive | poll + acquire fence + plain loads",
              all[0], all[all.size() / 2], herr[0], diff, hwait};
  }
  std::vector<bf16> b((size_t)M * D);
  CK(hipMemcpy(b.data(), X2 + (size_t)nphase * M * D, slab, hipMemcpyDeviceToHost));
  double ss = 0;
  for (size_t i = 0; i < b.size(); ++i) ss += (double)(float)b[i] * (double)(float)b[i];
  printf("chain of %d dependent projections, 64 rows, N = K = 1024, bf16 (rms of the final activations %.3g)\n", nphase, sqrt(ss / b.size()));
  printf("  (a) one launch per phase, hipGraph:            %8.1f us = %5.2f us per phase (best of %d)\n", 1e3f * ms_graph, 1e3f * ms_graph / nphase, reps);
  printf("  (a') mh_gemm_skinny in the same graph harness: plain %5.2f, + residual %5.2f, + RMSNorm row scale %5.2f us per phase\n",
         1e3f * ms_prod[0] / nphase, 1e3f * ms_prod[1] / nphase, 1e3f * ms_prod[2] / nphase);
  for (int v = 0; v < 3; ++v)
    printf("  (b%d) one persistent launch, %-78s %8.1f us = %5.2f us per phase (best; median %5.2f); timeouts %u; final activations differ in %zu of %d; workgroup 0 polled %.0f ticks per phase\n",
           v, res[v].what, 1e3f * res[v].best, 1e3f * res[v].best / nphase, 1e3f * res[v].med / nphase, res[v].timeouts, res[v].diff, M * D, (double)res[v].wait / nphase);
  return 0;
}
