// Prices of the two primitives a per-XCD persistent decode kernel would live on, measured on the device:
//   * an XCD-LOCAL barrier: the 32 workgroups that the dispatcher puts on one XCD (block b -> XCD b % 8, checked here against
//     HW_REG_XCC_ID) meet on one counter in that XCD's L2 (relaxed agent-scope atomic add, relaxed polling load, s_sleep),
//     data handed over by plain stores + s_waitcnt vmcnt(0) before the arrive and L1-bypassing (sc1) loads after it;
//   * weight streaming: every CU reads its 1/32 slice of a weight matrix per phase while all eight XCDs stream the SAME matrix
//     (each XCD holds 8 of the 64 sequences, so each needs every weight): bytes per CU per phase = rows(W) * K * 2 / 32.
// Modes: barrier only, stream only, both.  Every phase also re-reads the 16 KiB "activation" block its XCD wrote in the previous
// phase and counts stale words (a wrong placement assumption or a missing fence shows up as stale > 0, not as a hang: all
// spins are bounded).
// Build: hipcc --offload-arch=gfx950 -O3 tools/xcd_probe.hip -o tools/bin/xcd_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ inline u32x4 load_sc1(const void* p) {  // 16-byte load that bypasses the CU's L1 (served by the XCD's L2)
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// counters: cnt[xcd * 32] (one 128-byte line per XCD); err[0] = timeouts, err[1] = stale words, err[2] = placement mismatches
template <bool BARRIER, bool STREAM>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ W, int64_t op_chunks /* 16-B chunks of one matrix */,
                                             int nops, int64_t slice_chunks, int nphase, unsigned* cnt, unsigned* act /* [8][4096] */,
                                             unsigned* err, unsigned* sink, uint64_t* stamps) {
  extern __shared__ char lds_pad[];  // (occupies the CU: one workgroup per CU)
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 7) != xcd) atomicAdd(&err[2], 1u);
  }
  unsigned acc = 0;
  unsigned* myact = act + xcd * 4096;
  for (int p = 0; p < nphase; ++p) {
    // 1. the block the XCD's workgroups wrote in phase p-1 (4096 words = 16 KiB), L1 bypassed
    if (p > 0) {
      for (int i = tid; i < 1024; i += 256) {
        const u32x4 v = load_sc1(myact + 4 * i);
        const unsigned want = (unsigned)(p - 1) * 131u + (unsigned)((4 * i) >> 7);  // word w written by workgroup w >> 7
        if (v[0] != want || v[1] != want || v[2] != want || v[3] != want) atomicAdd(&err[1], 1u);
        acc ^= v[0];
      }
    }
    // 2. this CU's slice of the phase's weight matrix
    if (STREAM) {
      const u32x4* src = W + (int64_t)(p % nops) * op_chunks + (int64_t)local * slice_chunks;
      u32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      for (int64_t i = tid; i < slice_chunks; i += 256 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (i + 256 * j < slice_chunks) ? __builtin_nontemporal_load(src + i + 256 * j) : a0;
        a0 ^= v[0] ^ v[4];
        a1 ^= v[1] ^ v[5];
        a2 ^= v[2] ^ v[6];
        a3 ^= v[3] ^ v[7];
      }
      acc ^= a0[0] ^ a1[1] ^ a2[2] ^ a3[3];
    }
    // 3. this workgroup's 128 words of the next block
    if (tid < 128) myact[local * 128 + tid] = (unsigned)p * 131u + (unsigned)local;
    // 4. XCD-local barrier
    if (BARRIER) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_add(&cnt[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = 32u * (unsigned)(p + 1);
        int spins = 0;
        while (__hip_atomic_load(&cnt[xcd * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) {
            atomicAdd(&err[0], 1u);
            break;
          }
        }
        if (blockIdx.x == 0 && p < 512) stamps[p] = __builtin_amdgcn_s_memrealtime() - t0;
      }
      __syncthreads();
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <bool BARRIER, bool STREAM>
static void run(const char* name, const u32x4* W, int64_t op_bytes, int nops, int nphase, unsigned* cnt, unsigned* act, unsigned* err,
                unsigned* sink, uint64_t* stamps) {
  const int64_t op_chunks = op_bytes / 16, slice_chunks = op_chunks / 32;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<BARRIER, STREAM>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(cnt, 0, 8 * 32 * 4));
    CK(hipMemset(err, 0, 16));
    CK(hipMemset(act, 0, 8 * 4096 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    probe<BARRIER, STREAM><<<256, 256, 96 * 1024>>>(W, op_chunks, nops, slice_chunks, nphase, cnt, act, err, sink, stamps);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  unsigned h[4];
  CK(hipMemcpy(h, err, 16, hipMemcpyDeviceToHost));
  std::vector<uint64_t> st(512);
  CK(hipMemcpy(st.data(), stamps, 512 * 8, hipMemcpyDeviceToHost));
  double wsum = 0;
  int n = nphase < 512 ? nphase : 512;
  for (int i = 8; i < n; ++i) wsum += (double)st[i];
  printf("%-44s op %6.2f MB (%6.1f KB per CU): %7.2f us per phase", name, op_bytes / 1e6, op_bytes / 32.0 / 1024, best * 1e3 / nphase);
  if (STREAM) printf("  = %5.2f TB/s chip-wide (8 XCDs x the matrix)", 8.0 * op_bytes / (best * 1e-3 / nphase) / 1e12);
  if (BARRIER) printf("  [wg 0 waits %.2f us in the barrier]", n > 8 ? wsum / (n - 8) / 100.0 : 0.0);
  printf("  timeouts %u stale %u misplaced %u\n", h[0], h[1], h[2]);
}

int main() {
  const int64_t total = 64ll << 20;  // 64 MiB of "weights" (the token-level stack + lm_head are 51 MB)
  u32x4* W;
  unsigned *cnt, *act, *err, *sink;
  uint64_t* stamps;
  CK(hipMalloc(&W, total));
  CK(hipMemset(W, 1, total));
  CK(hipMalloc(&cnt, 8 * 32 * 4));
  CK(hipMalloc(&act, 8 * 4096 * 4));
  CK(hipMalloc(&err, 16));
  CK(hipMalloc(&sink, 16));
  CK(hipMalloc(&stamps, 512 * 8));
  run<true, false>("barrier only", W, 2 << 20, 1, 400, cnt, act, err, sink, stamps);
  // op sizes of the token-level stack (tv2o-medium): o / down 2 MB, gate|up 4 MB, q|k|v 6 MB, lm_head 7 MB
  for (int64_t mb : {2, 4, 6, 7}) {
    const int nops = (int)(total / (mb << 20));
    run<false, true>("stream only (fresh matrix every phase)", W, mb << 20, nops, 400, cnt, act, err, sink, stamps);
    run<true, true>("stream + barrier (fresh matrix every phase)", W, mb << 20, nops, 400, cnt, act, err, sink, stamps);
  }
  run<true, true>("stream + barrier (same 2 MB matrix: L2 hits)", W, 2 << 20, 1, 400, cnt, act, err, sink, stamps);
  return 0;
}
