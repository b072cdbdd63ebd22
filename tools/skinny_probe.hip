// In-kernel timeline of the decode projection (mh_gemm_skinny's structure) inside a replayed hipGraph, from
// s_memrealtime stamps (100 MHz constant clock, comparable across CUs and launches):
//   t0 kernel entry | t1 all operand loads issued | t2 loads landed | t3 MFMAs + LDS partials + barrier |
//   t4 reduction + residual + stores issued | t5 stores complete
// and the gap between the last t5 of launch i and the first t0 of launch i+1.  Variants:
//   V=0  the shipped structure (NW waves interleave over 32-deep K chunks, residual read after the reduction)
//   V=1  residual requested up front (one memory round trip less on the critical path)
// Build: hipcc --offload-arch=gfx950 -O3 tools/skinny_probe.hip -o tools/bin/skinny_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ inline uint64_t now() { return __builtin_amdgcn_s_memrealtime(); }

template <int NW, int MB, int V, bool STAMP>
__global__ __launch_bounds__(NW * 64) void skinny(const bf16* __restrict__ A, const bf16* __restrict__ W, bf16* __restrict__ C,
                                                 const bf16* __restrict__ R, int M, int N, int K, uint64_t* __restrict__ stamps) {
  constexpr int MR = MB * 16;
  __shared__ float red[NW][MR][17];
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
  if (STAMP) t0 = now();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int nc_w = K / (32 * NW);
  const int m0 = blockIdx.y * MR;
  const bf16* arow[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) arow[mb] = A + (int64_t)(m0 + mb * 16 + fi) * K + fg * 8;
  const bf16* wrow = W + (int64_t)(blockIdx.x * 16 + fi) * K + fg * 8;
  // residual prefetch (V=1): the four threads of a row each need 4 consecutive bf16 of R
  const int ml = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * 4;
  uint64_t rpre = 0;
  if (V == 1 && threadIdx.x < MR * 4)
    rpre = *reinterpret_cast<const uint64_t*>(R + (int64_t)(m0 + ml) * N + blockIdx.x * 16 + c0);
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NCMAX = 32 / NW;  // K <= 1024 in this probe
  bf16x8 wf[NCMAX], xf[NCMAX][MB];
#pragma unroll
  for (int ci = 0; ci < NCMAX; ++ci) {
    if (ci < nc_w) {
      const int k = (wave + NW * ci) * 32;
      wf[ci] = *reinterpret_cast<const bf16x8*>(wrow + k);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xf[ci][mb] = *reinterpret_cast<const bf16x8*>(arow[mb] + k);
    }
  }
  if (STAMP) {
    t1 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t2 = now();
  }
#pragma unroll
  for (int ci = 0; ci < NCMAX; ++ci)
    if (ci < nc_w)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ci], xf[ci][mb], acc[mb], 0, 0, 0);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][mb * 16 + fi][4 * fg + e] = acc[mb][e];
  __syncthreads();
  if (STAMP) t3 = now();
  if (threadIdx.x < MR * 4) {
    const int m = m0 + ml;
    bf16 out[4];
    const bf16* rp = (V == 1) ? reinterpret_cast<const bf16*>(&rpre) : R + (int64_t)m * N + blockIdx.x * 16 + c0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = red[0][ml][c0 + j];
#pragma unroll
      for (int w = 1; w < NW; ++w) t += red[w][ml][c0 + j];
      out[j] = (bf16)(t + (float)rp[j]);
    }
    *reinterpret_cast<uint64_t*>(C + (int64_t)m * N + blockIdx.x * 16 + c0) = *reinterpret_cast<uint64_t*>(out);
  }
  if (STAMP) {
    t4 = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t5 = now();
    if (threadIdx.x == 0) {
      uint64_t* s = stamps + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 6;
      s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5;
    }
  }
}

template <int NW, int MB, int V, bool STAMP>
double run(const char* name, int M, int N, int K, int L, bf16* x0, bf16* x1, std::vector<bf16*>& Ws, uint64_t* stamps, hipStream_t st,
           bool print_stamps) {
  dim3 grid(N / 16, (M / 16) / MB);
  const int nwg = grid.x * grid.y;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < L; ++l) {
    bf16* in = (l & 1) ? x1 : x0;
    bf16* out = (l & 1) ? x0 : x1;
    skinny<NW, MB, V, STAMP><<<grid, NW * 64, 0, st>>>(in, Ws[l % Ws.size()], out, in, M, N, K, stamps + (int64_t)l * nwg * 6);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int reps = 20;
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = 1e3 * ms / reps / L;
  printf("%-34s M=%d N=%d K=%d grid %dx%d x %d thr: %6.2f us per launch", name, M, N, K, grid.x, grid.y, NW * 64, us);
  if (STAMP && print_stamps) {
    std::vector<uint64_t> h((size_t)L * nwg * 6);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    // per launch: first entry, median of each phase over workgroups, last exit; then gaps between launches
    double ph[5] = {0, 0, 0, 0, 0}, span = 0, gap = 0, first_to_last_entry = 0;
    int ng = 0;
    for (int l = 1; l < L; ++l) {  // skip the first launch of the graph
      uint64_t first = ~0ull, last_entry = 0, last = 0;
      std::vector<double> d[5];
      for (int w = 0; w < nwg; ++w) {
        const uint64_t* s = &h[((size_t)l * nwg + w) * 6];
        first = std::min(first, s[0]);
        last_entry = std::max(last_entry, s[0]);
        last = std::max(last, s[5]);
        for (int p = 0; p < 5; ++p) d[p].push_back((double)(s[p + 1] - s[p]) * 0.01);
      }
      for (int p = 0; p < 5; ++p) {
        std::sort(d[p].begin(), d[p].end());
        ph[p] += d[p][d[p].size() / 2];
      }
      span += (double)(last - first) * 0.01;
      first_to_last_entry += (double)(last_entry - first) * 0.01;
      uint64_t prev_last = 0;
      for (int w = 0; w < nwg; ++w) prev_last = std::max(prev_last, h[((size_t)(l - 1) * nwg + w) * 6 + 5]);
      gap += (double)((int64_t)first - (int64_t)prev_last) * 0.01;
      ++ng;
    }
    printf("\n    median per workgroup (us): issue %.2f | loads land %.2f | mfma+lds+barrier %.2f | reduce+store issue %.2f | store done %.2f"
           "   kernel span first-entry..last-exit %.2f (entries spread over %.2f), gap to next launch %.2f",
           ph[0] / ng, ph[1] / ng, ph[2] / ng, ph[3] / ng, ph[4] / ng, span / ng, first_to_last_entry / ng, gap / ng);
  }
  printf("\n");
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return us;
}

__global__ void empty_kernel(bf16* p) {
  if (p == nullptr) p[0] = (bf16)0.f;
}

void run_empty(int gx, int gy, int threads, int L, hipStream_t st) {
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int l = 0; l < L; ++l) empty_kernel<<<dim3(gx, gy), threads, 0, st>>>((bf16*)(uintptr_t)16);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty kernel, grid %dx%d x %d threads: %.2f us per graph node\n", gx, gy, threads, 1e3 * ms / 20 / L);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
}

__global__ void fill(bf16* p, int64_t n, float scale, uint32_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (bf16)(scale * ((float)(h & 0xffff) / 32768.f - 1.f));
  }
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int M = 64, L = 48;
  uint64_t* stamps;
  CK(hipMalloc(&stamps, (size_t)L * 1024 * 6 * 8));
  run_empty(1, 1, 64, L, st);
  run_empty(64, 1, 512, L, st);
  run_empty(64, 4, 512, L, st);
  run_empty(64, 4, 256, L, st);
  run_empty(256, 4, 256, L, st);
  for (int K : {1024}) {
    for (int N : {1024}) {
      bf16 *x0, *x1;
      CK(hipMalloc(&x0, (size_t)M * 4096 * 2));
      CK(hipMalloc(&x1, (size_t)M * 4096 * 2));
      fill<<<64, 256, 0, st>>>(x0, (int64_t)M * 4096, 1.f, 1);
      fill<<<64, 256, 0, st>>>(x1, (int64_t)M * 4096, 1.f, 2);
      for (int nW : {1, 12}) {  // 1: the same 2 MB matrix every launch (cache resident); 12: a 24 MB working set
        std::vector<bf16*> Ws(nW);
        for (auto& w : Ws) {
          CK(hipMalloc(&w, (size_t)N * K * 2));
          fill<<<256, 256, 0, st>>>(w, (int64_t)N * K, 0.02f, 7);
        }
        CK(hipStreamSynchronize(st));
        printf("--- %d distinct weight matrices of %d x %d (%.1f MB each) ---\n", nW, N, K, N * K * 2 / 1e6);
        run<8, 4, 0, false>("V0 NW8 MB4 (r01 form)", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<8, 4, 0, true>("V0 NW8 MB4 stamped", M, N, K, L, x0, x1, Ws, stamps, st, true);
        run<8, 1, 0, false>("V0 NW8 MB1", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<8, 1, 0, true>("V0 NW8 MB1 stamped", M, N, K, L, x0, x1, Ws, stamps, st, true);
        run<8, 1, 1, false>("V1 NW8 MB1 residual up front", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<8, 1, 1, true>("V1 NW8 MB1 stamped", M, N, K, L, x0, x1, Ws, stamps, st, true);
        run<4, 1, 1, false>("V1 NW4 MB1", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<4, 1, 1, true>("V1 NW4 MB1 stamped", M, N, K, L, x0, x1, Ws, stamps, st, true);
        run<4, 2, 1, false>("V1 NW4 MB2", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<2, 1, 1, false>("V1 NW2 MB1", M, N, K, L, x0, x1, Ws, stamps, st, false);
        run<4, 4, 1, false>("V1 NW4 MB4", M, N, K, L, x0, x1, Ws, stamps, st, false);
        for (auto& w : Ws) CK(hipFree(w));
      }
      CK(hipFree(x0));
      CK(hipFree(x1));
    }
  }
  return 0;
}
