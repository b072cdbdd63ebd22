// VERDICT r04 item 4, priced by measurement: could a dK/dV-stationary attention backward also produce dQ, accumulating it with fp32
// atomics, and so replace the separate dQ kernel (264 us per layer at 16 x 2048, 16 heads, head_dim 64)?
// This probe issues exactly the atomic traffic such a kernel would add and NOTHING else: one workgroup of 4 waves per (batch x head,
// 64-key tile); for every 64-query tile at or below the diagonal it adds a 64 x 64 fp32 tile into dQ[b, h, q0 .. q0 + 64, 0 .. 64)
// (each wave 16 rows; a lane 16 consecutive floats of a row: unsafe-fp-atomics global_atomic_add_f32, no return).  Work items are
// placed like the production kernels' ((batch x head) -> XCD), so every address is only ever touched from one XCD.
// The time of this kernel is a LOWER bound on what the one-pass backward would add to the dK/dV kernel (356 us per layer).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_dq_probe.hip -o tools/bin/atomic_dq_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

// MODE 0: atomics; 1: plain stores of the same tiles (what the traffic costs without the read-modify-write); 2: atomics of a
// quarter of the tiles (every fourth query tile) -- a scaling check
template <int MODE>
__global__ __launch_bounds__(256) void dq_traffic(float* dq, int S, int nbh) {
  const int ntile = S / 64;
  // (batch x head) -> XCD: blocks b % 8 run on XCD b % 8; give XCD x the heads x, x + 8, ...
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int per_xcd = nbh / 8;
  const int bh = xcd + 8 * (idx % per_xcd), kt = idx / per_xcd;
  if (kt >= ntile) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* base = dq + (size_t)bh * S * 64;
  const float v = 1.0f + (float)kt * 1e-3f;
  for (int qt = kt; qt < ntile; qt += (MODE == 2 ? 4 : 1)) {  // causal: query tiles at or below the key tile's diagonal
    float* t = base + (size_t)(qt * 64 + wave * 16) * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = t + r * 64 + lane;  // 64 lanes = one 256-byte row: the most coalesced form there is
      if (MODE == 1) *p = v;
      else atomicAdd(p, v);
    }
  }
}

int main() {
  const int B = 16, H = 16, S = 2048, nbh = B * H;
  float* dq;
  CK(hipMalloc(&dq, (size_t)nbh * S * 64 * 4));
  CK(hipMemset(dq, 0, (size_t)nbh * S * 64 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = nbh * (S / 64);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
      CK(hipEventRecord(e0));
      if (mode == 0) dq_traffic<0><<<grid, 256>>>(dq, S, nbh);
      else if (mode == 1) dq_traffic<1><<<grid, 256>>>(dq, S, nbh);
      else dq_traffic<2><<<grid, 256>>>(dq, S, nbh);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double tiles = (double)nbh * (S / 64) * (S / 64 + 1) / 2 / (mode == 2 ? 4 : 1);
    printf("%s: %.1f us per layer-sized launch (B = %d, H = %d, S = %d: %.0f tile updates of 64 x 64 fp32 = %.2f GB %s)\n",
           mode == 0 ? "fp32 atomic adds" : mode == 1 ? "plain stores     " : "atomics, 1/4 tiles", best * 1e3, B, H, S, tiles,
           tiles * 16384 / 1e9, mode == 1 ? "written" : "read-modify-written");
  }
  // check: every element of a query tile received (number of key tiles at or above it) additions
  float* h = (float*)malloc(64 * 4);
  CK(hipMemcpy(h, dq + (size_t)5 * S * 64 + (size_t)(S - 1) * 64, 64 * 4, hipMemcpyDeviceToHost));
  printf("sample dq value %.4f (last query row of head 5)\n", h[7]);
  return 0;
}
