// Pipelined bf16 projection GEMM (second structure): C = alpha * opA(A) * opB(B)^T + beta * R.
//
// r01 run 2 measured the first structure (gemm.hip: 128x128 tile, 2 LDS stages, vmcnt(0)+barrier per K-step) at
// 450-900 TFLOP/s over the model's shapes: every K-step exposes the full HBM/L2 latency of the tile it just
// requested, hidden only by the second resident block.  This kernel keeps TWO K-tiles of LDS-DMA in flight:
//   * 256x128 block tile, 8 waves (4 along M x 2 along N, each 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 fragments),
//     K-step 64, three LDS stages of 48 KiB (144 KiB, one block per CU, two waves per SIMD);
//   * per K-step: s_waitcnt vmcnt(6) (all of this wave's loads except the newest tile have landed) -> raw
//     s_barrier (everyone's tile t landed, everyone finished tile t-1) -> issue tile t+2 into the stage tile t-1
//     vacated -> ds_read + MFMA on tile t.  No vmcnt(0) in the loop, loads span barriers
//     (cdna_hip_programming.md "Pipelining across barriers": LDS-DMA data is ordered for a ds_read by the issuing
//     waves' counted vmcnt followed by a barrier the reader has passed);
//   * operands in either layout (row-major K-contiguous, or contraction-major via ds_read_b64_tr_b16), same LDS
//     swizzles and zero-block tail handling as gemm.hip; same epilogue (swapped MFMA -> 8-byte stores, split-K).
// Roofline: MFMA, 2.5 PFLOP/s dense bf16.
#include <limits.h>

#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) char g_zero16[16];  // source of out-of-range chunks

constexpr int PBM = 256, PBN = 128, PBK = 64;
constexpr int A_BYTES = PBM * 128, B_BYTES = PBN * 128, STAGE_BYTES = A_BYTES + B_BYTES;  // 32 KiB + 16 KiB
constexpr int NSTAGE = 3;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 147456
constexpr int NWAVE = 8;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ inline int tswz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }

// Per-lane staging context, computed once per block: for each of this wave's LDS-DMA instructions of a tile the
// lane's source pointer at contraction offset 0 and the contraction limit below which its 16-byte chunk is in
// range (INT_MIN for an out-of-range row).  Per K-tile a chunk then costs one compare, one 64-bit add and one
// select instead of re-deriving rows, swizzles and bounds (r01 PMC: 3.7 VALU instructions per MFMA before).
template <int ROWS>
struct StageCtx {
  static constexpr int NI = ROWS / 8 / NWAVE;
  const bf16* p[NI];
  int klim[NI];
};

// row-major operand X[r][k]: tile ROWS x 128 B, 16-byte chunks swizzled by ((r>>1)&7)
template <int ROWS>
__device__ inline void stage_init_n(StageCtx<ROWS>& c, const bf16* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows,
                                    int64_t kend, int wave, int lane) {
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int it = 0; it < StageCtx<ROWS>::NI; ++it) {
    const int r = (wave + NWAVE * it) * 8 + rsub;
    const int ch = pc ^ ((r >> 1) & 7);
    const int64_t grow = row0 + r;
    c.p[it] = base + (grow < nrows ? grow : 0) * ld + ch * 8;
    c.klim[it] = (grow < nrows) ? (int)kend - ch * 8 : INT_MIN;
  }
}
template <int ROWS>
__device__ inline void stage_n(const StageCtx<ROWS>& c, int k0, char* tile, int wave) {
#pragma unroll
  for (int it = 0; it < StageCtx<ROWS>::NI; ++it) {
    const void* src = (k0 < c.klim[it]) ? (const void*)(c.p[it] + k0) : (const void*)g_zero16;
    glds16(src, tile + (wave + NWAVE * it) * 1024);
  }
}

// contraction-major operand X[k][r]: tile 64 k-rows x (2*ROWS) bytes, 32-byte granules swizzled by tswz(k)
template <int ROWS>
__device__ inline void stage_init_t(StageCtx<ROWS>& c, const bf16* __restrict__ base, int64_t ld, int64_t r0, int64_t nrows,
                                    int64_t kend, int wave, int lane) {
  constexpr int CPR = ROWS / 8;        // 16-byte chunks per k-row
  constexpr int KPI = 64 / CPR;        // k-rows per 1 KiB instruction
  const int ksub = lane / CPR, pc = lane % CPR;
#pragma unroll
  for (int it = 0; it < StageCtx<ROWS>::NI; ++it) {
    const int krow = (wave + NWAVE * it) * KPI + ksub;
    const int lg = (pc >> 1) ^ tswz(krow);
    const int64_t r = r0 + lg * 16 + (pc & 1) * 8;
    c.p[it] = base + (int64_t)krow * ld + (r < nrows ? r : 0);
    c.klim[it] = (r < nrows) ? (int)kend - krow : INT_MIN;
  }
}
template <int ROWS>
__device__ inline void stage_t(const StageCtx<ROWS>& c, int k0, int64_t ld, char* tile, int wave) {
  const int64_t koff = (int64_t)k0 * ld;
#pragma unroll
  for (int it = 0; it < StageCtx<ROWS>::NI; ++it) {
    const void* src = (k0 < c.klim[it]) ? (const void*)(c.p[it] + koff) : (const void*)g_zero16;
    glds16(src, tile + (wave + NWAVE * it) * 1024);
  }
}

template <bool TR, int ROWS>
__device__ inline bf16x8 frag(const char* tile, int row0, int kk, int fi, int fg) {
  if constexpr (!TR) {
    return *reinterpret_cast<const bf16x8*>(tile + lds_tile_off(row0 + fi, kk * 4 + fg));
  } else {
    union {
      bf16x8 v;
      s16x4_t h[2];
    } u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = kk * 32 + fg * 8 + t * 4 + (fi >> 2);
      const int off = krow * (2 * ROWS) + (((row0 >> 4) ^ tswz(krow)) << 5) + ((fi & 3) << 3);
      u.h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + off));
    }
    return u.v;
  }
}

template <bool TA, bool TB, bool PP>
__global__ __launch_bounds__(512) void gemm_pipe_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                                        int64_t ldb, bf16* C, int64_t ldc, const bf16* R, int64_t ldr,
                                                        int64_t M, int64_t N, int64_t K, float alpha, float beta, int tiles_n,
                                                        int nwg, int64_t k_per_split, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  int tm, tn;
  gemm_tile_of(swz, nwg / tiles_n, tiles_n, 4, tm, tn);
  const int64_t m0 = (int64_t)tm * PBM, n0 = (int64_t)tn * PBN;

  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int nt = (kend > kbeg) ? (int)((kend - kbeg + PBK - 1) / PBK) : 0;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fi = lane & 15, fg = lane >> 4;

  StageCtx<PBM> ca;
  StageCtx<PBN> cb;
  if constexpr (TA) stage_init_t<PBM>(ca, A, lda, m0, M, kend, wave, lane);
  else stage_init_n<PBM>(ca, A, lda, m0, M, kend, wave, lane);
  if constexpr (TB) stage_init_t<PBN>(cb, B, ldb, n0, N, kend, wave, lane);
  else stage_init_n<PBN>(cb, B, ldb, n0, N, kend, wave, lane);
  auto issue = [&](int t, int stage) {
    char* buf = smem + stage * STAGE_BYTES;
    const int k0 = (int)kbeg + t * PBK;
    if constexpr (TA) stage_t<PBM>(ca, k0, lda, buf, wave);
    else stage_n<PBM>(ca, k0, buf, wave);
    if constexpr (TB) stage_t<PBN>(cb, k0, ldb, buf + A_BYTES, wave);
    else stage_n<PBN>(cb, k0, buf + A_BYTES, wave);
  };
  // every wave issues exactly 6 LDS-DMA instructions per K-tile (4 for A, 2 for B) in both layouts
  if (nt > 0) issue(0, 0);
  if (nt > 1) issue(1, 1);
  int cur = 0;       // stage holding tile t
  int fill = 2;      // stage tile t+2 goes to (= the one tile t-1 vacated)
  if constexpr (PP) {
    // Ping-pong schedule.  Waves w and w+4 share a SIMD; group 0 (waves 0-3, rows 0-127) and group 1 (waves
    // 4-7, rows 128-255) run the same two segments per K-tile — LOAD (fragments of tile t: LDS -> registers,
    // then issue the LDS-DMA of tile t+2) and MFMA (32 matrix instructions on those registers) — but one segment
    // apart, with one workgroup barrier between segments.  So on every SIMD one wave is in its MFMA segment
    // while the other reads LDS / issues loads: the matrix pipe never waits for the LDS pipe of its own wave.
    //   group 0:  [LOAD 0][MFMA 0][LOAD 1][MFMA 1] ...                 barrier b separates segments b-1 | b
    //   group 1:  [ idle ][LOAD 0][MFMA 0][LOAD 1][MFMA 1] ...
    // Ordering: tile t is read by LOAD(t) in segments 2t (group 0) and 2t+1 (group 1); every wave waits for its
    // own LDS-DMA of tile t (vmcnt) before barrier 2t, and drains its ds_reads (lgkmcnt) at the end of a LOAD
    // segment, so the stage of tile t-1 is free when tile t+2 is issued into it after barrier 2t.
    const int grp = wave >> 2;
    bf16x8 fx[2][4], fw[2][4];
    auto load_frags = [&](int stage) {
      const char* tA = smem + stage * STAGE_BYTES;
      const char* tB = tA + A_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          fw[kk][f] = frag<TB, PBN>(tB, wn * 64 + f * 16, kk, fi, fg);
          fx[kk][f] = frag<TA, PBM>(tA, wm * 64 + f * 16, kk, fi, fg);
        }
    };
    auto mfma_all = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
          for (int fm = 0; fm < 4; ++fm)
            acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[kk][fn], fx[kk][fm], acc[fn][fm], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_tile = [&](int t) {
      if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto load_segment = [&](int t) {
      load_frags(cur);
      if (t + 2 < nt) issue(t + 2, fill);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
      fill = (fill == NSTAGE - 1) ? 0 : fill + 1;
    };
    if (grp == 0) {
      for (int t = 0; t < nt; ++t) {
        wait_tile(t);
        bar();  // barrier 2t
        load_segment(t);
        bar();  // barrier 2t+1
        mfma_all();
      }
      bar();    // barrier 2nt
    } else {
      for (int t = 0; t < nt; ++t) {
        wait_tile(t);
        bar();  // barrier 2t
        if (t > 0) mfma_all();
        bar();  // barrier 2t+1
        load_segment(t);
      }
      bar();    // barrier 2nt
      if (nt > 0) mfma_all();
    }
  } else
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < nt) issue(t + 2, fill);
    const char* tA = smem + cur * STAGE_BYTES;
    const char* tB = tA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fx[4], fw[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fw[f] = frag<TB, PBN>(tB, wn * 64 + f * 16, kk, fi, fg);
        fx[f] = frag<TA, PBM>(tA, wm * 64 + f * 16, kk, fi, fg);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
          acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[fn], fx[fm], acc[fn][fm], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    }
    cur = (cur == NSTAGE - 1) ? 0 : cur + 1;
    fill = (fill == NSTAGE - 1) ? 0 : fill + 1;
  }

  // epilogue (identical to gemm.hip): lane (fi, fg) of fragment (fn, fm) holds C[m][n..n+3]
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
          bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + m * ldr + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = alpha * v[e] + beta * (float)rv[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = alpha * v[e];
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
        *reinterpret_cast<bf16x4*>(C + m * ldc + n) = o;
      } else {
        for (int e = 0; e < 4 && n + e < N; ++e) {
          float x = alpha * v[e];
          if (R != nullptr && beta != 0.f) x += beta * (float)R[m * ldr + n + e];
          C[m * ldc + n + e] = (bf16)x;
        }
      }
    }
  }
}

template <bool TA, bool TB, bool PP>
int launch_one(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R, int64_t ldr,
               int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk, void* workspace, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<TA, TB, PP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      mh_set_error("gemm_pipe: cannot raise dynamic LDS to %d bytes: %s", LDS_BYTES, hipGetErrorString(e));
      return MH_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int64_t tiles_m = (M + PBM - 1) / PBM, tiles_n = (N + PBN - 1) / PBN;
  const int nwg = (int)(tiles_m * tiles_n);
  const int64_t kps = ((K + splitk - 1) / splitk + PBK - 1) / PBK * PBK;
  dim3 grid(nwg, 1, splitk);
  gemm_pipe_kernel<TA, TB, PP><<<grid, 512, LDS_BYTES, st>>>((const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)C, ldc, (const bf16*)R,
                                                         ldr, M, N, K, alpha, beta, (int)tiles_n, nwg, kps, (float*)workspace);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

}  // namespace

// called by gemm.hip after argument validation (bf16 only); pingpong = 0: lockstep waves, 1: ping-pong wave groups
int mh_gemm_pipe_bf16(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                      const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                      void* workspace, int pingpong, hipStream_t st) {
#define MH_PIPE(TA_, TB_)                                                                                                    \
  return pingpong ? launch_one<TA_, TB_, true>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st)  \
                  : launch_one<TA_, TB_, false>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st)
  if (ta && tb) MH_PIPE(true, true);
  if (ta) MH_PIPE(true, false);
  if (tb) MH_PIPE(false, true);
  MH_PIPE(false, false);
#undef MH_PIPE
}
