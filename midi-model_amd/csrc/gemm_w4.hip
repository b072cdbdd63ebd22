// One-wave-per-SIMD bf16 projection GEMM (256x256 tile, 4 waves): C = alpha * opA(A) * opB(B)^T + beta * R.
//
// Why another structure.  The 8-wave ping-pong kernel (gemm_pp256.hip) gives every wave a 128x64 tile: 32 MFMAs
// (512 matrix-pipe cycles) per K-step against 12 KiB of LDS fragment reads and 4 LDS-DMA issues per wave.
// Ablations on MI355X (tools/bench_gemm.py, variant 6 ablate bits) put that LOAD segment at ~890 cycles against
// the 512-cycle MFMA segment, so the matrix pipe idles half the time whatever the barriers do.  Here a workgroup is
// FOUR waves, one per SIMD, each with a 128x128 tile held in 256 accumulator registers (a wave alone on its SIMD
// may use 512): per K-step a wave issues 32 v_mfma_f32_32x32x16_bf16 (1024 pipe cycles) against 16 KiB of fragment
// reads and 8 LDS-DMA issues, i.e. half the LDS traffic and half the issue work per flop, and it software-pipelines
// them itself: the fragments of step c+1 and the LDS-DMA of step c+4 are issued BETWEEN the MFMAs of step c.
// One s_barrier per K-step hands the LDS stage over.
//
// LDS: 4 stages x (A tile 256 x 32 + B tile 256 x 32) bf16 = 128 KiB, same tile formats as gemm_pp256.hip
// (64-byte rows with XOR-swizzled 16-byte chunks; contraction-major operands as 512-byte k-rows read with
// ds_read_b64_tr_b16).  Roofline: MFMA, 2.5 PFLOP/s dense bf16.
#include <limits.h>

#include <type_traits>

#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) char g_zero16[16];  // source of out-of-range chunks

constexpr int WBM = 256, WBN = 256, WBK = 32;
constexpr int OP_BYTES = 256 * 64;
constexpr int STAGE_BYTES = 2 * OP_BYTES;
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 131072
constexpr int NWAVE = 4;
constexpr int NI = 4;  // LDS-DMA pieces (1 KiB) per wave, operand and K-step

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ inline int tswz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }
__device__ inline int xswz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

struct StageCtx {
  const bf16* p[NI];
  int klim[NI];
};

// row-major operand: piece j = tile rows 16j..16j+15, lane -> (row 16j + lane/4, 16-byte chunk lane%4 ^ swizzle)
__device__ inline void stage_init_n(StageCtx& c, const bf16* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows,
                                    int kend, int wave, int lane) {
  const int rsub = lane >> 2, pc = lane & 3;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int r = (wave + NWAVE * it) * 16 + rsub;
    const int ch = pc ^ xswz(r);
    const int64_t grow = row0 + r;
    c.p[it] = base + (grow < nrows ? grow : 0) * ld + ch * 8;
    c.klim[it] = (grow < nrows) ? kend - ch * 8 : INT_MIN;
  }
}
// contraction-major operand: piece j = k-rows 2j, 2j+1 of the step, lane -> (k-row 2j + lane/32, 8 tile rows)
__device__ inline void stage_init_t(StageCtx& c, const bf16* __restrict__ base, int64_t ld, int64_t r0, int64_t nrows,
                                    int kend, int wave, int lane) {
  const int ksub = lane >> 5, pc = lane & 31;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int krow = (wave + NWAVE * it) * 2 + ksub;
    const int lg = (pc >> 1) ^ tswz(krow);
    const int64_t r = r0 + lg * 16 + (pc & 1) * 8;
    c.p[it] = base + (int64_t)krow * ld + (r < nrows ? r : 0);
    c.klim[it] = (r < nrows) ? kend - krow : INT_MIN;
  }
}
template <bool TR>
__device__ inline void stage_piece(const StageCtx& c, int it, int k0, int64_t ld, char* tile, int wave) {
  const int64_t koff = TR ? (int64_t)k0 * ld : (int64_t)k0;
  const void* src = (k0 < c.klim[it]) ? (const void*)(c.p[it] + koff) : (const void*)g_zero16;
  glds16(src, tile + (wave + NWAVE * it) * 1024);
}

// 32x32x16 MFMA operand fragment: lane holds 8 consecutive k (k = 16 kh + 8 (lane / 32) + e) of tile row
// row32 + lane % 32, row32 = half * 128 + b * 32.  Per lane only a few byte offsets are needed; block b and k half kh
// add compile-time constants (ds_read immediate offsets):
//   row-major tile:      off[kh] + b * 2048                    (one ds_read_b128)
//   contraction-major:   off[b] + kh * 8192 + t * 2048, t=0,1  (two ds_read_b64_tr_b16: 4 k-rows x 16 tile rows per
//                        16-lane group; the 32-byte granule swizzle tswz(krow) is the same for every b, kh, t of a lane)
template <bool TR>
__device__ inline void frag_offsets(int (&off)[4], int half, int lane) {
  if constexpr (!TR) {
    const int row = half * 128 + (lane & 31);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) off[kh] = row * 64 + (((kh * 2 + (lane >> 5)) ^ xswz(row)) << 4);
    off[2] = off[3] = 0;
  } else {
    const int fi = lane & 15, g = lane >> 4;
    const int krow = 8 * (g >> 1) + (fi >> 2);  // + 16 kh + 4 t: leaves tswz unchanged
#pragma unroll
    for (int b = 0; b < 4; ++b)
      off[b] = krow * 512 + (((half * 8 + b * 2 + (g & 1)) ^ tswz(krow)) << 5) + ((fi & 3) << 3);
  }
}
template <bool TR>
__device__ inline bf16x8 frag(const char* tile, const int (&off)[4], int b, int kh) {
  if constexpr (!TR) {
    return *reinterpret_cast<const bf16x8*>(tile + off[kh] + b * 2048);
  } else {
    union {
      bf16x8 v;
      s16x4_t h[2];
    } u;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      u.h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4_t*)(tile + off[b] + kh * 8192 + t * 2048));
    return u.v;
  }
}

template <bool TA, bool TB, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_w4_kernel(
    const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb, bf16* C, int64_t ldc, const bf16* R,
    int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int tiles_n, int nwg, int64_t k_per_split,
    float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;  // 128-row / 128-column half of the tile

  // XCD-contiguous, grouped tile order (see gemm_tile_of)
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int swz = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (bid >> 3);
  int tm, tn;
  gemm_tile_of(swz, nwg / tiles_n, tiles_n, 4, tm, tn);
  const int64_t m0 = (int64_t)tm * WBM, n0 = (int64_t)tn * WBN;

  const int64_t kbeg64 = (int64_t)blockIdx.z * k_per_split;
  const int kbeg = (int)kbeg64;
  const int kend = (int)((kbeg64 + k_per_split < K) ? kbeg64 + k_per_split : K);
  const int nt = (kend > kbeg) ? (kend - kbeg + WBK - 1) / WBK : 0;

  f32x16 acc[4][4];  // [nb][mb]: columns wn*128 + nb*32.., rows wm*128 + mb*32..
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

  StageCtx ca, cb;
  if constexpr (TA) stage_init_t(ca, A, lda, m0, M, kend, wave, lane);
  else stage_init_n(ca, A, lda, m0, M, kend, wave, lane);
  if constexpr (TB) stage_init_t(cb, B, ldb, n0, N, kend, wave, lane);
  else stage_init_n(cb, B, ldb, n0, N, kend, wave, lane);
  // LDS-DMA piece j (0..3: A, 4..7: B) of K-step t; step t lives in stage t & 3
  auto issue_piece = [&](int t, int j) __attribute__((always_inline)) {
    char* buf = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const int k0 = kbeg + t * WBK;
    if (j < 4) stage_piece<TA>(ca, j, k0, lda, buf, wave);
    else stage_piece<TB>(cb, j - 4, k0, ldb, buf + OP_BYTES, wave);
  };

  int offa[4], offb[4];
  frag_offsets<TA>(offa, wm, lane);
  frag_offsets<TB>(offb, wn, lane);
  bf16x8 fx[2][4][2], fw[2][4][2];  // [set][32-row block][k half]; sets alternate between K-steps
  auto read_frag = [&](auto SET, int t, int i) __attribute__((always_inline)) {  // fragment i (0..15) of step t
    constexpr int s = decltype(SET)::value;
    const char* tA = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const int kh = i >> 3, b = i & 3;
    if ((i & 4) == 0) fw[s][b][kh] = frag<TB>(tA + OP_BYTES, offb, b, kh);
    else fx[s][b][kh] = frag<TA>(tA, offa, b, kh);
  };

  // The K loop runs an even number of steps; steps past the slice end stage zeros (their k0 fails every klim test)
  // and every step issues its 8 pieces, so the s_waitcnt counts below are constants.
  const int nt2 = (nt + 1) & ~1;

  // ---- pipeline fill: K-steps 0..3 in flight, fragments of step 0 in set 0 ------------------------------------
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(t, j);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) read_frag(std::integral_constant<int, 0>{}, 0, i);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  // ---- K-step c: 32 MFMAs on set S; between them the 16 fragment reads of step c+1 into the other set and the
  //      8 LDS-DMA pieces of step c+4 (into the stage step c just vacated) ------------------------------------
  auto step = [&](auto SET, int c) __attribute__((always_inline)) {
    constexpr int s = decltype(SET)::value;
    // own LDS-DMA of step c+1 landed (steps c+2, c+3 still in flight); after the barrier every wave's has, and every
    // wave has finished reading stage c (its fragments sit in registers)
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int kh = i >> 4, nb = (i >> 2) & 3, mb = i & 3;
      acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s][nb][kh], fx[s][mb][kh], acc[nb][mb], 0, 0, 0);
      if (i < 16 && !(ABL & 2)) read_frag(std::integral_constant<int, 1 - s>{}, c + 1, i);
      if ((i & 3) == 1 && !(ABL & 1)) issue_piece(c + 4, i >> 2);  // one piece per four MFMAs: the TA sees a steady stream
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  for (int c = 0; c < nt2; c += 2) {
    step(std::integral_constant<int, 0>{}, c);
    step(std::integral_constant<int, 1>{}, c + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup

  // ---- epilogue: register r of acc[nb][mb] is column n = (r&3) + 8 (r>>2) + 4 (lane/32) of row m = lane % 32 ----
  const bool partial = (gridDim.z > 1);
  const bool has_res = (R != nullptr && beta != 0.f);
  float* wsz = partial ? ws + (int64_t)blockIdx.z * M * N : nullptr;
  const bool vec_ok = partial ? ((N & 3) == 0 && ((uintptr_t)ws & 15) == 0)
                              : ((ldc & 3) == 0 && ((uintptr_t)C & 7) == 0 &&
                                 (!has_res || ((ldr & 3) == 0 && ((uintptr_t)R & 7) == 0)));
  const bool interior = (m0 + WBM <= M) && (n0 + WBN <= N);
  const int64_t mb0 = m0 + wm * 128 + (lane & 31), nb0 = n0 + wn * 128 + 4 * (lane >> 5);
  if (interior && vec_ok) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int64_t m = mb0 + mb * 32, n = nb0 + nb * 32 + rq * 8;
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[nb][mb][rq * 4 + e];
          if (partial) {
            *reinterpret_cast<f32x4*>(wsz + m * N + n) = v;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= alpha;
            if (has_res) {
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + m * ldr + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += beta * (float)rv[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
            *reinterpret_cast<bf16x4*>(C + m * ldc + n) = o;
          }
        }
    return;
  }
#pragma unroll
  for (int mb = 0; mb < 4; ++mb)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = mb0 + mb * 32, n = nb0 + nb * 32 + (r >> 2) * 8 + (r & 3);
        if (m >= M || n >= N) continue;
        const float v = acc[nb][mb][r];
        if (partial) {
          wsz[m * N + n] = v;
        } else {
          float x = alpha * v;
          if (has_res) x += beta * (float)R[m * ldr + n];
          C[m * ldc + n] = (bf16)x;
        }
      }
}

template <bool TA, bool TB, int ABL>
int launch_one(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R, int64_t ldr,
               int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk, void* workspace, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4_kernel<TA, TB, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      mh_set_error("gemm_w4: cannot raise dynamic LDS to %d bytes: %s", LDS_BYTES, hipGetErrorString(e));
      return MH_ERR_LAUNCH;
    }
    attr_set = true;
  }
  const int64_t tiles_m = (M + WBM - 1) / WBM, tiles_n = (N + WBN - 1) / WBN;
  const int nwg = (int)(tiles_m * tiles_n);
  const int64_t kps = ((K + splitk - 1) / splitk + WBK - 1) / WBK * WBK;
  dim3 grid(nwg, 1, splitk);
  gemm_w4_kernel<TA, TB, ABL><<<grid, 256, LDS_BYTES, st>>>((const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)C, ldc,
                                                       (const bf16*)R, ldr, M, N, K, alpha, beta, (int)tiles_n, nwg, kps,
                                                       (float*)workspace);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

}  // namespace

extern int g_mh_gemm_ablate;  // api.cpp (micro-benchmark only: 1 = no LDS-DMA after the fill, 2 = no fragment reads)

// called by gemm.hip after argument validation (bf16 only)
int mh_gemm_w4_bf16(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                    const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                    void* workspace, hipStream_t st) {
#define MH_W4(TA_, TB_, ABL_) \
  return launch_one<TA_, TB_, ABL_>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st)
  if (g_mh_gemm_ablate == 1 && !ta) { if (tb) MH_W4(false, true, 1); else MH_W4(false, false, 1); }
  if (g_mh_gemm_ablate == 2 && !ta) { if (tb) MH_W4(false, true, 2); else MH_W4(false, false, 2); }
  if (g_mh_gemm_ablate == 3 && !ta) { if (tb) MH_W4(false, true, 3); else MH_W4(false, false, 3); }
  if (ta && tb) MH_W4(true, true, 0);
  if (ta) MH_W4(true, false, 0);
  if (tb) MH_W4(false, true, 0);
  MH_W4(false, false, 0);
#undef MH_W4
}
