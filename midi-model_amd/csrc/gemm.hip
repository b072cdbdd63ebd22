// Dense projections on the matrix cores: C = alpha * A[M,K] * B[N,K]^T + beta * R.
//
// gfx950 design (first structure; see DESIGN.md for the roofline and the planned 256^2 8-phase step):
//   * 128x128 block tile, 4 waves (2x2), each wave 64x64 = 4x4 fragments of v_mfma_f32_16x16x32_bf16
//     (fp32 verification mode: v_mfma_f32_16x16x4_f32, exact fp32).
//   * K-step = 128 bytes per row (64 bf16 / 32 fp32).  Tiles go HBM -> LDS with global_load_lds_dwordx4
//     (no VGPR round trip, no ds_write pass), double-buffered, one barrier per K-step.
//   * LDS rows are 128 B with the 16-byte chunks XOR-swizzled (common.h) so every ds_read_b128 lane
//     group touches 16 distinct slots; the swizzle is applied on the per-lane SOURCE address because the
//     LDS-DMA destination is lane-linear.
//   * Out-of-range rows / contraction tails read a 16-byte zero block instead of being predicated.
//   * Block ids are remapped so each XCD (private L2) owns a contiguous run of tiles that share the A panel.
//   * The MFMA is issued "swapped" (B rows as the MFMA row index) so each lane ends with 4 consecutive
//     output columns -> 8/16-byte stores.
//   * split-K (grid.z) for the weight-gradient shapes (small M,N, huge K): fp32 partials + reduce kernel.
#include <type_traits>

#include "common.h"

__device__ __attribute__((aligned(16))) char g_zero16[16];

template <typename T> struct Mma;
template <> struct Mma<bf16> {
  __device__ static inline f32x4 run(const Pack<bf16>& a, const Pack<bf16>& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // lane group g holds k = 4g..4g+3 of a 16-wide k block: four exact-fp32 16x16x4 MFMAs, element e
  // pairing A's and B's e-th value (any consistent k assignment gives the same dot product).
  __device__ static inline f32x4 run(const Pack<float>& a, const Pack<float>& b, f32x4 c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[e], b.v[e], c, 0, 0, 0);
    return c;
  }
};

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = 128 * 128;  // 128 rows x 128 B

template <typename T>
__device__ inline void stage_tile(const T* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                  int64_t kend, char* lds_tile, int wave, int lane) {
  constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int g8 = wave + 4 * it;  // 8-row group handled by this wave
    const int r = g8 * 8 + rsub;
    const int c = pc ^ lds_swz(r);
    const int64_t grow = row0 + r;
    const int64_t k = k0 + (int64_t)c * EPC;
    const void* src = (grow < nrows && k < kend) ? (const void*)(base + grow * ld + k) : (const void*)g_zero16;
    glds16(src, lds_tile + g8 * 1024);
  }
}

// ---- "T-mode" operands: stored contraction-major, X[k][r] (r contiguous) --------------------------------
// This is what dgrad (W[N][K] read as K rows x N contraction) and wgrad (dY[M][N], X[M][K] contracted over M)
// present; instead of re-laying them out in HBM the tile is staged as it lies — 64 k-rows x 256 B — and the MFMA
// fragment (8 consecutive k for one row r) is assembled by two ds_read_b64_tr_b16 transpose reads (semantics
// pinned by tools/probe.hip: within a 16-lane group, lane p supplies 4 contiguous bf16 of row p>>2; output lane
// i receives element i of each of the 4 rows).  The 32-byte granules (= one fragment's 16 rows) of k-row k are
// XOR-swizzled with (k&3)|((k>>1)&4) so the 8 k-rows a 32-lane half touches land on 8 different granules.
__device__ inline int tswz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }

__device__ inline void stage_tile_t(const bf16* __restrict__ base, int64_t ld, int64_t r0, int64_t nrows, int64_t k0,
                                    int64_t kend, char* lds_tile, int wave, int lane) {
  const int ksub = lane >> 4, pc = lane & 15;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int q = wave + 4 * it;          // group of 4 k-rows (1 KiB)
    const int krow = q * 4 + ksub;
    const int lg = (pc >> 1) ^ tswz(krow);  // logical granule held by this physical slot
    const int64_t r = r0 + lg * 16 + (pc & 1) * 8;
    const int64_t k = k0 + krow;
    const void* src = (k < kend && r < nrows) ? (const void*)(base + k * ld + r) : (const void*)g_zero16;
    glds16(src, lds_tile + q * 1024);
  }
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ inline Pack<bf16> frag_t(const char* tile, int rl /* fragment's first row in the tile, multiple of 16 */, int kk,
                                    int fi, int fg) {
  union {
    Pack<bf16> p;
    s16x4_t h[2];
  } u;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int krow = kk * 32 + fg * 8 + t * 4 + (fi >> 2);
    const int off = krow * 256 + (((rl >> 4) ^ tswz(krow)) << 5) + ((fi & 3) << 3);
    u.h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + off));
  }
  return u.p;
}

template <typename T, bool TR>
__device__ inline void stage_any(const T* __restrict__ base, int64_t ld, int64_t r0, int64_t nrows, int64_t k0, int64_t kend,
                                 char* tile, int wave, int lane) {
  if constexpr (TR) {
    static_assert(sizeof(T) == 2, "T-mode operands are bf16 only");
    stage_tile_t((const bf16*)base, ld, r0, nrows, k0, kend, tile, wave, lane);
  } else {
    stage_tile<T>(base, ld, r0, nrows, k0, kend, tile, wave, lane);
  }
}
template <typename T, bool TR>
__device__ inline Pack<T> frag_any(const char* tile, int row0, int kk, int fi, int fg) {
  if constexpr (TR) {
    return frag_t(tile, row0, kk, fi, fg);
  } else {
    return *reinterpret_cast<const Pack<T>*>(tile + lds_tile_off(row0 + fi, kk * 4 + fg));
  }
}

template <typename T, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B,
                                                      int64_t ldb, T* C, int64_t ldc, const T* R, int64_t ldr, int64_t M, int64_t N,
                                                      int64_t K, float alpha, float beta, int tiles_n, int nwg,
                                                      int64_t k_per_split, float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [stage][A|B]
  constexpr int BK = 128 / sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap: dispatcher places block b on XCD b%8; give each XCD a contiguous tile run.
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  int tm, tn;
  gemm_tile_of(swz, nwg / tiles_n, tiles_n, 8, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int nt = (int)((kend - kbeg + BK - 1) / BK);

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fg = lane >> 4;

  if (nt > 0) {
    stage_any<T, TA>(A, lda, m0, M, kbeg, kend, smem, wave, lane);
    stage_any<T, TB>(B, ldb, n0, N, kbeg, kend, smem + TILE_BYTES, wave, lane);
  }
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
    if (t + 1 < nt) {
      const int64_t k0 = kbeg + (int64_t)(t + 1) * BK;
      stage_any<T, TA>(A, lda, m0, M, k0, kend, nxt, wave, lane);
      stage_any<T, TB>(B, ldb, n0, N, k0, kend, nxt + TILE_BYTES, wave, lane);
    }
    const char* tA = cur;
    const char* tB = cur + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      Pack<T> fx[4], fw[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fw[f] = frag_any<T, TB>(tB, wn * 64 + f * 16, kk, fi, fg);
        fx[f] = frag_any<T, TA>(tA, wm * 64 + f * 16, kk, fi, fg);
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm) acc[fn][fm] = Mma<T>::run(fw[fn], fx[fm], acc[fn][fm]);
    }
    __syncthreads();
  }

  // epilogue: lane (fi, fg) of fragment (fn, fm) holds C[m][n..n+3], m = ..+fi, n = ..+4*fg
  const bool partial = (gridDim.z > 1);
  float* wsz = partial ? ws + (int64_t)blockIdx.z * M * N : nullptr;
  const bool vec_ok = partial ? ((N & 3) == 0)
                              : ((ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                                 (R == nullptr || ((ldr & 3) == 0 && ((uintptr_t)R & 15) == 0)));
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int64_t m = m0 + wm * 64 + fm * 16 + fi;
    if (m >= M) continue;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int64_t n = n0 + wn * 64 + fn * 16 + fg * 4;
      if (n >= N) continue;
      f32x4 v = acc[fn][fm];
      if (partial) {
        float* dst = wsz + m * N + n;
        if (vec_ok && n + 3 < N) {
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
          for (int e = 0; e < 4 && n + e < N; ++e) dst[e] = v[e];
        }
        continue;
      }
      if (vec_ok && n + 3 < N) {
        if (R != nullptr && beta != 0.f) {
          if constexpr (sizeof(T) == 2) {
            bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + m * ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = alpha * v[e] + beta * (float)rv[e];
          } else {
            f32x4 rv = *reinterpret_cast<const f32x4*>(R + m * ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = alpha * v[e] + beta * rv[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = alpha * v[e];
        }
        if constexpr (sizeof(T) == 2) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
          *reinterpret_cast<bf16x4*>(C + m * ldc + n) = o;
        } else {
          *reinterpret_cast<f32x4*>(C + m * ldc + n) = v;
        }
      } else {
        for (int e = 0; e < 4 && n + e < N; ++e) {
          float x = alpha * v[e];
          if (R != nullptr && beta != 0.f) x += beta * to_f(R[m * ldr + n + e]);
          C[m * ldc + n + e] = from_f<T>(x);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, T* __restrict__ C,
                                                            int64_t ldc, const T* __restrict__ R, int64_t ldr,
                                                            int64_t M, int64_t N, int splitk, float alpha,
                                                            float beta) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splitk; ++z) s += ws[(int64_t)z * total + i];
    const int64_t m = i / N, n = i - m * N;
    float x = alpha * s;
    if (R != nullptr && beta != 0.f) x += beta * to_f(R[m * ldr + n]);
    C[m * ldc + n] = from_f<T>(x);
  }
}

extern thread_local int g_mh_gemm_variant;  // api.cpp
int mh_gemm_pp256_bf16(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                       const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                       void* workspace, hipStream_t st);  // gemm_pp256.hip

template <typename T>
static int gemm_launch(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                       const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                       void* workspace, hipStream_t st) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int BK = 128 / sizeof(T);
  MH_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  MH_REQUIRE(lda % EPC == 0 && ldb % EPC == 0, "gemm: lda/ldb must be multiples of %d elements (lda=%ld ldb=%ld)", EPC,
             (long)lda, (long)ldb);
  MH_REQUIRE(sizeof(T) == 2 || (!ta && !tb), "gemm: contraction-major (transposed) operands are bf16 only");
  MH_REQUIRE((ta ? lda >= M : lda >= K) && (tb ? ldb >= N : ldb >= K), "gemm: leading dimension smaller than the row length");
  MH_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "gemm: A/B must be 16-byte aligned");
  MH_REQUIRE(beta == 0.f || R != nullptr, "gemm: beta != 0 needs R");
  const int64_t tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  MH_REQUIRE(tiles_m * tiles_n < (1ll << 30), "gemm: too many tiles");
  if (splitk < 1) splitk = 1;
  // every z-slice gets a BK-aligned k range; slices past K (possible after the rounding) store zero partials
  const int64_t kps = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
  MH_REQUIRE(splitk == 1 || workspace != nullptr, "gemm: split-K needs a workspace");
  if constexpr (sizeof(T) == 2) {
#ifdef MH_AB_BUILDS
    if (g_mh_gemm_variant != 0)  // production kernel; 0 = the first structure below (kept as an independent check)
#else  // the bf16 instantiations of the first structure are only in the A/B test library (libmidihip_ab.so)
    MH_REQUIRE(g_mh_gemm_variant != 0, "gemm(bf16): option gemm = 0 selects the 128x128 kernel, which is only in the A/B test library");
#endif
      return mh_gemm_pp256_bf16(A, lda, ta, B, ldb, tb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st);
  }
  const int nwg = (int)(tiles_m * tiles_n);
  dim3 grid(nwg, 1, splitk);
#define MH_GEMM_LAUNCH(TA_, TB_)                                                                                      \
  gemm_nt_kernel<T, TA_, TB_><<<grid, 256, 0, st>>>((const T*)A, lda, (const T*)B, ldb, (T*)C, ldc, (const T*)R, ldr, M, N, \
                                                     K, alpha, beta, (int)tiles_n, nwg, kps, (float*)workspace)
  if constexpr (sizeof(T) == 2) {
#ifdef MH_AB_BUILDS
    if (ta && tb) MH_GEMM_LAUNCH(true, true);
    else if (ta) MH_GEMM_LAUNCH(true, false);
    else if (tb) MH_GEMM_LAUNCH(false, true);
    else MH_GEMM_LAUNCH(false, false);
#endif
  } else {
    MH_GEMM_LAUNCH(false, false);
  }
#undef MH_GEMM_LAUNCH
  MH_LAUNCH_CHECK();
  return MH_OK;
}

template <bool FOLD>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ ws, bf16* __restrict__ C, int64_t ldc,
                                                             const bf16* R, int64_t ldr, int64_t M, int64_t N, int splitk,
                                                             float alpha, float beta, const bf16* __restrict__ wnorm,
                                                             const bf16* __restrict__ W, int64_t ldw, float* __restrict__ colpart);

template <typename T>
static int splitk_reduce_launch(const void* workspace, void* C, int64_t ldc, const void* R, int64_t ldr, int64_t M,
                                int64_t N, int splitk, float alpha, float beta, hipStream_t st) {
  MH_REQUIRE(M > 0 && N > 0 && splitk >= 1 && workspace != nullptr, "splitk_reduce: bad args");
  if constexpr (std::is_same<T, bf16>::value) {
    // r06: four columns per thread, 16-byte loads of the partials (the scalar form below moved 4 bytes per load: the reductions of
    // a training step, 4.2 GB of partials, ran at 4.1 TB/s); same sums in the same order
    if (N % 4 == 0 && ldc % 4 == 0 && (R == nullptr || ldr % 4 == 0) && (((uintptr_t)C | (uintptr_t)R) & 7) == 0 &&
        ((uintptr_t)workspace & 15) == 0) {
      const int64_t rows_y = M < 2048 ? M : 2048;
      dim3 grid((unsigned)((N / 4 + 255) / 256), (unsigned)rows_y);
      splitk_reduce4_kernel<false><<<grid, 256, 0, st>>>((const float*)workspace, (bf16*)C, ldc, (const bf16*)R, ldr, M, N, splitk, alpha,
                                                         beta, nullptr, nullptr, 0, nullptr);
      MH_LAUNCH_CHECK();
      return MH_OK;
    }
  }
  const int64_t total = M * N;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  splitk_reduce_kernel<T><<<blocks, 256, 0, st>>>((const float*)workspace, (T*)C, ldc, (const T*)R, ldr, M, N, splitk,
                                                  alpha, beta);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_gemm_splitk_reduce(const void* workspace, void* C, int64_t ldc, const void* R, int64_t ldr, int64_t M,
                                     int64_t N, int splitk, float alpha, float beta, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MH_BF16) return splitk_reduce_launch<bf16>(workspace, C, ldc, R, ldr, M, N, splitk, alpha, beta, st);
  if (dtype == MH_F32) return splitk_reduce_launch<float>(workspace, C, ldc, R, ldr, M, N, splitk, alpha, beta, st);
  mh_set_error("splitk_reduce: bad dtype %d", dtype);
  return MH_ERR_ARG;
}

extern "C" int mh_gemm(const void* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB, void* C,
                       int64_t ldc, const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                       int dtype, int splitk, void* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MH_BF16)
    return gemm_launch<bf16>(A, lda, transA, B, ldb, transB, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st);
  if (dtype == MH_F32)
    return gemm_launch<float>(A, lda, transA, B, ldb, transB, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st);
  mh_set_error("gemm: bad dtype %d", dtype);
  return MH_ERR_ARG;
}

int mh_gemm_pp256_dswiglu_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu,
                               void* DGU, int64_t lddgu, int64_t M, int64_t I, int64_t K, hipStream_t st, const float* rowscale);  // gemm_pp256.hip
int mh_gemm_pp256_scaled_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, const float* rowscale, hipStream_t st);  // gemm_pp256.hip

int mh_gemm_pp256_swiglu_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                              int64_t ldact, int64_t M, int64_t I, int64_t K, hipStream_t st, const float* rowscale);  // gemm_pp256.hip

static int gemm_swiglu_any(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                           int64_t ldact, const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype, void* stream);

extern "C" int mh_gemm_swiglu(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                              int64_t ldact, int64_t M, int64_t I, int64_t K, int dtype, void* stream) {
  return gemm_swiglu_any(A, lda, W, ldw, GU, ldgu, ACT, ldact, nullptr, M, I, K, dtype, stream);
}
extern "C" int mh_gemm_swiglu_scaled(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                                     int64_t ldact, const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(rowscale != nullptr && ((uintptr_t)rowscale & 15) == 0, "gemm_swiglu_scaled: rowscale must be a 16-byte aligned fp32 [M]");
  return gemm_swiglu_any(A, lda, W, ldw, GU, ldgu, ACT, ldact, rowscale, M, I, K, dtype, stream);
}

static int gemm_swiglu_any(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                           int64_t ldact, const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0,
             "gemm_swiglu: served by the production bf16 kernel only (use mh_gemm + mh_swiglu_fwd otherwise)");
  MH_REQUIRE(M > 0 && I > 0 && K > 0 && I % 128 == 0, "gemm_swiglu: bad shape M=%ld I=%ld K=%ld (I must be a multiple of 128)",
             (long)M, (long)I, (long)K);
  // (GU == NULL: the forward-only form -- a prompt prefill has no backward to keep gate|up for: only ACT is written)
  MH_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldact % 8 == 0 && lda >= K && ldw >= K && ldact >= I &&
                 (GU == nullptr || (ldgu % 8 == 0 && ldgu >= 2 * I)),
             "gemm_swiglu: leading dimensions must be multiples of 8 elements and cover the rows");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)GU | (uintptr_t)ACT) & 15) == 0, "gemm_swiglu: 16-byte alignment");
  return mh_gemm_pp256_swiglu_bf16(A, lda, W, ldw, GU, ldgu, ACT, ldact, M, I, K, (hipStream_t)stream, rowscale);
}

int mh_gemm_pp256_rope_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                            int64_t S, int64_t pos0, int64_t M, int64_t N, int64_t K, hipStream_t st, const float* rowscale);

static int gemm_rope_any(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                         int64_t npos, int64_t S, int64_t pos0, int head_dim, const float* rowscale, int64_t M, int64_t N, int64_t K,
                         int dtype, void* stream);

extern "C" int mh_gemm_rope(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                            int64_t npos, int64_t S, int64_t pos0, int head_dim, int64_t M, int64_t N, int64_t K, int dtype,
                            void* stream) {
  return gemm_rope_any(A, lda, W, ldw, C, ldc, table, npos, S, pos0, head_dim, nullptr, M, N, K, dtype, stream);
}
extern "C" int mh_gemm_rope_scaled(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                                   int64_t npos, int64_t S, int64_t pos0, int head_dim, const float* rowscale, int64_t M, int64_t N,
                                   int64_t K, int dtype, void* stream) {
  MH_REQUIRE(rowscale != nullptr && ((uintptr_t)rowscale & 15) == 0, "gemm_rope_scaled: rowscale must be a 16-byte aligned fp32 [M]");
  return gemm_rope_any(A, lda, W, ldw, C, ldc, table, npos, S, pos0, head_dim, rowscale, M, N, K, dtype, stream);
}

static int gemm_rope_any(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                         int64_t npos, int64_t S, int64_t pos0, int head_dim, const float* rowscale, int64_t M, int64_t N, int64_t K,
                         int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0,
             "gemm_rope: served by the production bf16 kernel only (use mh_gemm + mh_rope otherwise)");
  MH_REQUIRE(M > 0 && K > 0 && N > 0 && N % 192 == 0 && head_dim == 64,
             "gemm_rope: bad shape M=%ld N=%ld K=%ld head_dim=%d (N = 3 * heads * 64)", (long)M, (long)N, (long)K, head_dim);
  MH_REQUIRE(S > 0 && pos0 >= 0 && pos0 + (S < M ? S : M) <= npos && npos < (int64_t(1) << 31), "gemm_rope: table too short");
  MH_REQUIRE(S < (int64_t(1) << 31), "gemm_rope: S = %ld does not fit the 32 bits it travels in", (long)S);
  MH_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N,
             "gemm_rope: leading dimensions must be multiples of 8 elements and cover the rows");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)table) & 15) == 0, "gemm_rope: 16-byte alignment");
  return mh_gemm_pp256_rope_bf16(A, lda, W, ldw, C, ldc, table, S, pos0, M, N, K, (hipStream_t)stream, rowscale);
}

int mh_gemm_pp256_rowss_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R,
                             int64_t ldr, int64_t M, int64_t N, int64_t K, float* rowss, hipStream_t st);  // gemm_pp256.hip

extern "C" int mh_gemm_rowss(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R,
                             int64_t ldr, float* rowss, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0, "gemm_rowss: served by the production bf16 kernel only");
  MH_REQUIRE(M > 0 && N > 0 && K > 0 && N % 64 == 0 && rowss != nullptr, "gemm_rowss: bad shape M=%ld N=%ld K=%ld (N must be a multiple of 64)",
             (long)M, (long)N, (long)K);
  MH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N && (R == nullptr || (ldr % 8 == 0 && ldr >= N)),
             "gemm_rowss: leading dimensions must be multiples of 8 elements and cover the rows");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)R) & 15) == 0, "gemm_rowss: 16-byte alignment");
  MH_REQUIRE(((M + 255) / 256) * ((N + 255) / 256) < (1ll << 30), "gemm_rowss: too many tiles");
  return mh_gemm_pp256_rowss_bf16(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, rowss, (hipStream_t)stream);
}

extern "C" int mh_gemm_dswiglu(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu,
                               void* DGU, int64_t lddgu, int64_t M, int64_t I, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0,
             "gemm_dswiglu: served by the production bf16 kernel only (use mh_gemm + mh_swiglu_bwd otherwise)");
  MH_REQUIRE(M > 0 && I > 0 && K > 0 && I % 8 == 0, "gemm_dswiglu: bad shape M=%ld I=%ld K=%ld", (long)M, (long)I, (long)K);
  MH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0 && lda >= K && ldb >= I && ldgu >= 2 * I &&
                 lddgu >= 2 * I,
             "gemm_dswiglu: leading dimensions must be multiples of 8 elements and cover the rows");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)GU | (uintptr_t)DGU) & 15) == 0, "gemm_dswiglu: 16-byte alignment");
  return mh_gemm_pp256_dswiglu_bf16(A, lda, B, ldb, GU, ldgu, DGU, lddgu, M, I, K, (hipStream_t)stream, nullptr);
}

extern "C" int mh_gemm_dswiglu_scaled(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu,
                                      void* DGU, int64_t lddgu, const float* rowscale, int64_t M, int64_t I, int64_t K, int dtype,
                                      void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0, "gemm_dswiglu_scaled: served by the production bf16 kernel only");
  MH_REQUIRE(rowscale != nullptr && ((uintptr_t)rowscale & 15) == 0, "gemm_dswiglu_scaled: rowscale must be a 16-byte aligned fp32 [M]");
  MH_REQUIRE(M > 0 && I > 0 && K > 0 && I % 8 == 0, "gemm_dswiglu_scaled: bad shape M=%ld I=%ld K=%ld", (long)M, (long)I, (long)K);
  MH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldgu % 8 == 0 && lddgu % 8 == 0 && lda >= K && ldb >= I && ldgu >= 2 * I &&
                 lddgu >= 2 * I,
             "gemm_dswiglu_scaled: leading dimensions must be multiples of 8 elements and cover the rows");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)GU | (uintptr_t)DGU) & 15) == 0, "gemm_dswiglu_scaled: 16-byte alignment");
  return mh_gemm_pp256_dswiglu_bf16(A, lda, B, ldb, GU, ldgu, DGU, lddgu, M, I, K, (hipStream_t)stream, rowscale);
}

extern "C" int mh_gemm_nt_scaled(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                                 const float* rowscale, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16 && g_mh_gemm_variant != 0, "gemm_nt_scaled: served by the production bf16 kernel only");
  MH_REQUIRE(rowscale != nullptr && ((uintptr_t)rowscale & 15) == 0, "gemm_nt_scaled: rowscale must be a 16-byte aligned fp32 [M]");
  MH_REQUIRE(M > 0 && N > 0 && K > 0, "gemm_nt_scaled: empty problem");
  MH_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N, "gemm_nt_scaled: leading dimensions");
  MH_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "gemm_nt_scaled: 16-byte alignment");
  MH_REQUIRE(((M + 255) / 256) * ((N + 255) / 256) < (1ll << 30), "gemm_nt_scaled: too many tiles");
  return mh_gemm_pp256_scaled_bf16(A, lda, B, ldb, C, ldc, M, N, K, rowscale, (hipStream_t)stream);
}

// ---- the weight gradient of a projection behind a FOLDED RMSNorm (r06) ------------------------------------------------------------
// The folded forward multiplies x by W' = W (.) w (w = the norm weight, along the contraction), so the split-K weight-gradient
// GEMM delivers G' = d z^T x, the gradient with respect to W'.  This reduction of its fp32 partials applies the chain rule in
// the same pass:  dW[n,k] = alpha G'[n,k] w[k] (+ beta R[n,k]),  and block partials of  dw[k] = sum_n alpha G'[n,k] W[n,k]  (fp32,
// [mh_splitk_fold_blocks(M)][N]; mh_colsum folds them, deterministic).  Rows of the output are the projection's output features.
// Also the vectorised form of the plain reduction: four columns per thread, 16-byte partial loads.
constexpr int SKF_MAX_BLOCKS = 1024;  // (256 row blocks -- one workgroup per CU -- left the reduction latency-bound: +10 % on the weight-gradient launches)
extern "C" int mh_splitk_fold_blocks(int64_t M) { return (int)(M < SKF_MAX_BLOCKS ? (M < 1 ? 1 : M) : SKF_MAX_BLOCKS); }

template <bool FOLD>
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const float* __restrict__ ws, bf16* __restrict__ C, int64_t ldc,
                                                             const bf16* R, int64_t ldr, int64_t M, int64_t N, int splitk,
                                                             float alpha, float beta, const bf16* __restrict__ wnorm,
                                                             const bf16* __restrict__ W, int64_t ldw, float* __restrict__ colpart) {
  const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (c >= N) return;  // (N % 4 == 0: whole groups)
  const int64_t total = M * N;
  float wn[4] = {1.f, 1.f, 1.f, 1.f}, cs[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FOLD) {
    const bf16x4 w4 = *reinterpret_cast<const bf16x4*>(wnorm + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) wn[e] = (float)w4[e];
  }
  const bool use_r = (R != nullptr && beta != 0.f);
  for (int64_t m = blockIdx.y; m < M; m += gridDim.y) {
    f32x4 s = *reinterpret_cast<const f32x4*>(ws + m * N + c);
    if (splitk <= 8) {  // (all slices requested before the first addition; same order of additions)
      f32x4 p[7];
#pragma unroll
      for (int z = 1; z < 8; ++z)
        if (z < splitk) p[z - 1] = *reinterpret_cast<const f32x4*>(ws + (int64_t)z * total + m * N + c);
#pragma unroll
      for (int z = 1; z < 8; ++z)
        if (z < splitk) {
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += p[z - 1][e];
        }
    } else {
      for (int z = 1; z < splitk; ++z) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(ws + (int64_t)z * total + m * N + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += p[e];
      }
    }
    float x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = alpha * s[e];
    if constexpr (FOLD) {
      const bf16x4 W4 = *reinterpret_cast<const bf16x4*>(W + m * ldw + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        cs[e] += x[e] * (float)W4[e];
        x[e] *= wn[e];
      }
    }
    if (use_r) {
      const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(R + m * ldr + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] += beta * (float)r4[e];
    }
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (bf16)x[e];
    *reinterpret_cast<bf16x4*>(C + m * ldc + c) = o;
  }
  if constexpr (FOLD) *reinterpret_cast<f32x4*>(colpart + (int64_t)blockIdx.y * N + c) = f32x4{cs[0], cs[1], cs[2], cs[3]};
}

extern "C" int mh_gemm_splitk_reduce_fold(const void* workspace, void* C, int64_t ldc, const void* R, int64_t ldr, int64_t M,
                                          int64_t N, int splitk, float alpha, float beta, const void* wnorm, const void* W,
                                          int64_t ldw, float* colpart, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16, "splitk_reduce_fold: bf16 only");
  MH_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && splitk >= 1 && workspace != nullptr && wnorm != nullptr && W != nullptr && colpart != nullptr,
             "splitk_reduce_fold: bad args (N must be a multiple of 4)");
  MH_REQUIRE(ldc % 4 == 0 && ldw % 4 == 0 && ldw >= N && (R == nullptr || ldr % 4 == 0) &&
                 (((uintptr_t)C | (uintptr_t)W | (uintptr_t)wnorm | (uintptr_t)R) & 7) == 0 && ((uintptr_t)workspace & 15) == 0 &&
                 ((uintptr_t)colpart & 15) == 0,
             "splitk_reduce_fold: alignment");
  dim3 grid((unsigned)((N / 4 + 255) / 256), (unsigned)mh_splitk_fold_blocks(M));
  splitk_reduce4_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>((const float*)workspace, (bf16*)C, ldc, (const bf16*)R, ldr, M, N,
                                                                      splitk, alpha, beta, (const bf16*)wnorm, (const bf16*)W, ldw, colpart);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                          const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta,
                          int dtype, int splitk, void* workspace, void* stream) {
  return mh_gemm(A, lda, 0, B, ldb, 0, C, ldc, R, ldr, M, N, K, alpha, beta, dtype, splitk, workspace, stream);
}

// ---- transpose: out[c][r] = in[r][c]; 64x64 tiles through LDS, 16-byte global accesses both ways ----
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int64_t ldi, T* __restrict__ out,
                                                        int64_t ldo, int64_t rows, int64_t cols) {
  constexpr int EPC = 16 / sizeof(T);   // elements per 16-byte chunk
  constexpr int CPR = 64 / EPC;         // chunks per 64-element tile row
  constexpr int RPP = 256 / CPR;        // tile rows covered per pass
  constexpr int LD = 64 + 4 / sizeof(T);  // +1 dword of padding per row
  __shared__ T tile[64][LD];
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int t = threadIdx.x;
  const int ch = t % CPR, rr = t / CPR;
  const bool in_vec = (ldi % EPC == 0) && (((uintptr_t)in & 15) == 0);
  const bool out_vec = (ldo % EPC == 0) && (((uintptr_t)out & 15) == 0);
  for (int p = 0; p < 64 / RPP; ++p) {
    const int i = rr + p * RPP;
    const int64_t r = r0 + i, c = c0 + ch * EPC;
    if (in_vec && r < rows && c + EPC <= cols) {
      Pack<T> v = ld16(in + r * ldi + c);
#pragma unroll
      for (int e = 0; e < EPC; ++e) tile[i][ch * EPC + e] = v.v[e];
    } else {
      for (int e = 0; e < EPC; ++e)
        tile[i][ch * EPC + e] = (r < rows && c + e < cols) ? in[r * ldi + c + e] : from_f<T>(0.f);
    }
  }
  __syncthreads();
  for (int p = 0; p < 64 / RPP; ++p) {
    const int i = rr + p * RPP;  // output row within tile = input column
    const int64_t c = c0 + i, r = r0 + ch * EPC;
    if (c >= cols) continue;
    if (out_vec && r + EPC <= rows) {
      Pack<T> v;
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.v[e] = tile[ch * EPC + e][i];
      st16(out + c * ldo + r, v);
    } else {
      for (int e = 0; e < EPC && r + e < rows; ++e) out[c * ldo + r + e] = tile[ch * EPC + e][i];
    }
  }
}

extern "C" int mh_transpose(const void* in, int64_t ldi, void* out, int64_t ldo, int64_t rows, int64_t cols,
                            int dtype, void* stream) {
  MH_REQUIRE(rows > 0 && cols > 0, "transpose: empty");
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  MH_REQUIRE(grid.y < 65536, "transpose: too many row tiles (%u)", grid.y);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MH_BF16)
    transpose_kernel<bf16><<<grid, 256, 0, st>>>((const bf16*)in, ldi, (bf16*)out, ldo, rows, cols);
  else
    transpose_kernel<float><<<grid, 256, 0, st>>>((const float*)in, ldi, (float*)out, ldo, rows, cols);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
