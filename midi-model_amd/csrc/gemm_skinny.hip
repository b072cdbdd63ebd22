// Skinny projection for the decode step: C[M,N] = op(A)[M,K] * W[N,K]^T (+ R), M <= 64 rows (one per sequence), bf16.
//
// A decode step multiplies a handful of activation rows with every weight matrix once, so the work is streaming the
// weights from HBM: 6.3 MB for q|k|v, 16.8 MB for gate|up of tv2o-medium.  The training GEMM (256x256 tile) would put
// 4-32 workgroups on 256 CUs and needs a second launch to reduce its split-K partials; here a workgroup owns 16 or 32
// output columns, its 4 waves interleave over the 32-deep K chunks (so the workgroup reads 256 contiguous bytes of each
// weight row at a time), every lane keeps several 16-byte weight loads in flight, and the four partial tiles are summed
// through LDS.  The activation rows come from L2 (every workgroup re-reads them: 128 KiB for K = 1024).
//
// Prologues on A, so that a decoder layer is 6 launches instead of 13 (decode is launch-bound, profiles/r01_run6):
//   MH_SKINNY_NORM    A <- w_norm * round(A * rsqrt(mean(A^2) + eps))        (LlamaRMSNorm, modeling_llama.py:62-67)
//   MH_SKINNY_SWIGLU  A <- round(silu(gate)) * up, gate|up = A[M, 2K]        (LlamaMLP, modeling_llama.py:174-176)
// with the roundings of rmsnorm_fwd_kernel / swiglu_fwd_kernel (elementwise.hip).
// Roofline: HBM; algorithmic bytes = N*K*2 (weights once).
#include "common.h"

namespace {

constexpr int SK_MAXC = 8;  // K-chunks per wave the NORM prologue keeps in registers (K <= 1024)

template <int MODE, int NBT>  // NBT: 16-column blocks per workgroup
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const bf16* __restrict__ A, int64_t lda,
                                                          const bf16* __restrict__ W, int64_t ldw, bf16* __restrict__ C,
                                                          int64_t ldc, const bf16* __restrict__ R, int64_t ldr,
                                                          const bf16* __restrict__ nw, float eps, int M, int N, int K) {
  constexpr int NB = NBT * 16;
  __shared__ float red[4][64][NB + 1];
  __shared__ float ssq[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int n0 = blockIdx.x * NB;
  const int nc_w = K / 128;  // chunks of 32 per wave (wave w takes chunks w, w+4, ...)

  const bf16* arow[4];
  const bf16* wrow[NBT];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int m = mb * 16 + fi;
    arow[mb] = A + (int64_t)(m < M ? m : M - 1) * lda + fg * 8;
  }
#pragma unroll
  for (int nb = 0; nb < NBT; ++nb) {
    const int n = n0 + nb * 16 + fi;
    wrow[nb] = W + (int64_t)(n < N ? n : N - 1) * ldw + fg * 8;
  }

  f32x4 acc[4][NBT];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  bf16x8 af[MODE == 1 ? SK_MAXC : 1][4];
  if constexpr (MODE == 1) {
    // pass 1: this wave's share of the rows in registers, sum of squares per row
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ci = 0; ci < SK_MAXC; ++ci)
      if (ci < nc_w) {
        const int k = (wave + 4 * ci) * 32;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
          af[ci][mb] = *reinterpret_cast<const bf16x8*>(arow[mb] + k);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss[mb] += (float)af[ci][mb][e] * (float)af[ci][mb][e];
        }
      }
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      ss[mb] += __shfl_xor(ss[mb], 16, 64);
      ss[mb] += __shfl_xor(ss[mb], 32, 64);
      if (fg == 0) ssq[wave][mb * 16 + fi] = ss[mb];
    }
    __syncthreads();
    float rstd[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int r = mb * 16 + fi;
      rstd[mb] = rsqrtf((ssq[0][r] + ssq[1][r] + ssq[2][r] + ssq[3][r]) / (float)K + eps);
    }
#pragma unroll
    for (int ci = 0; ci < SK_MAXC; ++ci)
      if (ci < nc_w) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(nw + (wave + 4 * ci) * 32 + fg * 8);
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
          for (int e = 0; e < 8; ++e) af[ci][mb][e] = (bf16)((float)wv[e] * (float)(bf16)((float)af[ci][mb][e] * rstd[mb]));
      }
  }

  auto chunk = [&](int ci) __attribute__((always_inline)) {
    const int k = (wave + 4 * ci) * 32;
    bf16x8 wf[NBT], xf[4];
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) wf[nb] = *reinterpret_cast<const bf16x8*>(wrow[nb] + k);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      if constexpr (MODE == 0) {
        xf[mb] = *reinterpret_cast<const bf16x8*>(arow[mb] + k);
      } else if constexpr (MODE == 2) {
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(arow[mb] + k);
        const bf16x8 u = *reinterpret_cast<const bf16x8*>(arow[mb] + K + k);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gv = (float)g[e];
          const float s = (float)(bf16)(gv / (1.f + __expf(-gv)));
          xf[mb][e] = (bf16)(s * (float)u[e]);
        }
      }
    }
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
        acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], MODE == 1 ? af[MODE == 1 ? ci : 0][mb] : xf[mb],
                                                              acc[mb][nb], 0, 0, 0);
  };
  if constexpr (MODE == 1) {
#pragma unroll
    for (int ci = 0; ci < SK_MAXC; ++ci)
      if (ci < nc_w) chunk(ci);
  } else {
#pragma unroll 4
    for (int ci = 0; ci < nc_w; ++ci) chunk(ci);
  }

  // sum the four waves' partial tiles: lane (fi, fg) of acc[mb][nb] holds row mb*16+fi, columns nb*16 + 4 fg + e
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][mb * 16 + fi][nb * 16 + 4 * fg + e] = acc[mb][nb][e];
  __syncthreads();
  constexpr int CPT = NB / 4;  // columns per thread
  const int m = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * CPT;
  if (m < M) {
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      const int n = n0 + c0 + j;
      if (n < N) {
        float v = red[0][m][c0 + j] + red[1][m][c0 + j] + red[2][m][c0 + j] + red[3][m][c0 + j];
        if (R != nullptr) v += (float)R[(int64_t)m * ldr + n];
        C[(int64_t)m * ldc + n] = (bf16)v;
      }
    }
  }
}

template <int MODE>
int launch_mode(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* R, int64_t ldr,
                const void* nw, float eps, int M, int N, int K, hipStream_t st) {
  if (N <= 2048) {
    gemm_skinny_kernel<MODE, 1><<<(N + 15) / 16, 256, 0, st>>>((const bf16*)A, lda, (const bf16*)W, ldw, (bf16*)C, ldc,
                                                              (const bf16*)R, ldr, (const bf16*)nw, eps, M, N, K);
  } else {
    gemm_skinny_kernel<MODE, 2><<<(N + 31) / 32, 256, 0, st>>>((const bf16*)A, lda, (const bf16*)W, ldw, (bf16*)C, ldc,
                                                              (const bf16*)R, ldr, (const bf16*)nw, eps, M, N, K);
  }
  MH_LAUNCH_CHECK();
  return MH_OK;
}

}  // namespace

extern "C" int mh_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* R,
                              int64_t ldr, const void* norm_w, float eps, int mode, int64_t M, int64_t N, int64_t K,
                              int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16, "gemm_skinny: bf16 only (the fp32 verification mode uses mh_gemm)");
  MH_REQUIRE(M > 0 && M <= 64 && N > 0 && N < (1 << 30), "gemm_skinny: needs 1 <= M <= 64 rows (M=%ld N=%ld)", (long)M, (long)N);
  MH_REQUIRE(K > 0 && K % 128 == 0 && K < (1 << 24), "gemm_skinny: K=%ld must be a multiple of 128", (long)K);
  MH_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_skinny: A/W rows must be 16-byte aligned");
  MH_REQUIRE(mode >= 0 && mode <= 2, "gemm_skinny: mode %d", mode);
  MH_REQUIRE(mode != MH_SKINNY_NORM || (norm_w != nullptr && K <= SK_MAXC * 128 && ((uintptr_t)norm_w & 15) == 0),
             "gemm_skinny: the RMSNorm prologue needs norm_w and K <= %d", SK_MAXC * 128);
  MH_REQUIRE(mode != MH_SKINNY_SWIGLU || lda >= 2 * K, "gemm_skinny: the SwiGLU prologue reads gate|up rows of 2K elements");
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case MH_SKINNY_NORM: return launch_mode<1>(A, lda, W, ldw, C, ldc, R, ldr, norm_w, eps, (int)M, (int)N, (int)K, st);
    case MH_SKINNY_SWIGLU: return launch_mode<2>(A, lda, W, ldw, C, ldc, R, ldr, norm_w, eps, (int)M, (int)N, (int)K, st);
    default: return launch_mode<0>(A, lda, W, ldw, C, ldc, R, ldr, norm_w, eps, (int)M, (int)N, (int)K, st);
  }
}
