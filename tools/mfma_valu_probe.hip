// Issue model of one gfx950 SIMD for attention-like streams: how v_mfma_f32_32x32x16_bf16 and VALU / v_exp_f32 work from
// the same wave (fine interleave: each MFMA followed by its share of VALU) or from phases (all MFMAs, then all VALU)
// overlap at 1, 2 and 3 waves per SIMD.  Every workgroup is 4 waves (one per SIMD); occupancy is set by the LDS request.
// Prints shader cycles (s_memtime) per loop iteration of one wave and per-SIMD cycles per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_valu_probe tools/mfma_valu_probe.hip && tools/bin/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

// FINE: per MFMA, N plain VALU + X transcendental right behind it.  !FINE: M MFMAs back to back, then M*N + M*X VALU.
template <int M, int N, int X, bool FINE>
__global__ __launch_bounds__(256) void probe(float* out, uint64_t* cyc, int R) {
  extern __shared__ char smem[];
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(float)(threadIdx.x & 3);
    b[i] = (__bf16)(float)((threadIdx.x >> 2) & 3);
  }
  float x[8], y[4];
  for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x * 1e-3f + i;
  for (int i = 0; i < 4; ++i) y[i] = (float)threadIdx.x * 1e-4f;
  const float c1 = 0.999f, c2 = 1e-3f;
  if (smem[threadIdx.x] == 77) x[0] += 1.f;  // (keeps the LDS allocation)
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < R; ++r) {
    if (FINE) {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if (i & 1)
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
#pragma unroll
        for (int j = 0; j < X; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(y[j & 3]));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if (i & 1)
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < M * N; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
#pragma unroll
      for (int j = 0; j < M * X; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(y[j & 3]));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float* d_out;
static uint64_t* d_cyc;

template <int M, int N, int X, bool FINE>
static void run(int W) {
  const int R = 2000;
  const int lds = W == 1 ? 100 * 1024 : W == 2 ? 70 * 1024 : W == 3 ? 50 * 1024 : 36 * 1024;
  auto k = probe<M, N, X, FINE>;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = 256 * W;
  k<<<grid, 256, lds>>>(d_out, d_cyc, 10);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  k<<<grid, 256, lds>>>(d_out, d_cyc, R);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  static uint64_t h[1024];
  CHECK(hipMemcpy(h, d_cyc, grid * sizeof(uint64_t), hipMemcpyDeviceToHost));
  double avg = 0;
  for (int i = 0; i < grid; ++i) avg += (double)h[i];
  avg /= grid;
  const double per_iter = avg / R;  // cycles of one wave per iteration (s_memtime: 100 MHz ticks? printed raw)
  printf("%s M=%d N=%d X=%d W=%d : %8.1f ticks/iter/wave  wall %7.3f ms -> %7.1f ns per (iteration x waves/SIMD)  = %6.2f ns per MFMA per SIMD\n",
         FINE ? "fine  " : "coarse", M, N, X, W, per_iter, ms, ms * 1e6 / R / W, ms * 1e6 / R / W / M);
}

template <int N, int X>
static void sweep() {
  for (int W = 1; W <= 3; ++W) run<8, N, X, true>(W);
  for (int W = 1; W <= 3; ++W) run<8, N, X, false>(W);
}

int main() {
  CHECK(hipMalloc(&d_out, 1024 * 256 * sizeof(float)));
  CHECK(hipMalloc(&d_cyc, 1024 * sizeof(uint64_t)));
  sweep<0, 0>();
  sweep<2, 0>();
  sweep<4, 0>();
  sweep<6, 0>();
  sweep<8, 0>();
  sweep<12, 0>();
  sweep<4, 2>();
  sweep<6, 2>();
  sweep<8, 2>();
  printf("(8 MFMA 32x32x16 bf16 per iteration; ideal MFMA-bound = 32 cycles per MFMA per SIMD = %.2f ns at 2.4 GHz)\n", 32 / 2.4);
  return 0;
}
