// Persistent ping-pong bf16 projection GEMM (256x256 tile): C = alpha * opA(A) * opB(B)^T + beta * R.
//
// Same tile, LDS layouts, staging and ping-pong schedule as gemm_pp256.hip; what changes is the outer structure.
// gemm_pp256 runs one workgroup per output tile, one workgroup per CU: a tile's pipeline fill (first loads from
// HBM) and its epilogue (128 KiB of C written by every CU at the same moment) are never overlapped with MFMA work.
// Fitting time = a + b * K-steps over the model's shapes gave a = 12 us per tile against b = 0.86 us per 32-deep
// step (profiles/r01_run4_*): at K = 1024 a third of the kernel.  Here 256 workgroups (one per CU) each walk a
// list of work items (tile, K-slice) and keep ONE continuous stream of K-steps across item boundaries: the
// LDS-DMA of the next item's first steps is issued while the current item's last steps are multiplied, and a wave
// group writes its finished accumulators while the other group of the SIMD pair is inside its MFMA segment.
// Work items are dealt so that the workgroups of one XCD run consecutive items of the rasterised tile order at
// any one time (L2 reuse as in the non-persistent kernel).
// Roofline: MFMA, 2.5 PFLOP/s dense bf16.
#include <limits.h>

#include "common.h"

namespace {

__device__ __attribute__((aligned(16))) char g_zero16[16];  // source of out-of-range chunks

constexpr int QBM = 256, QBN = 256, QBK = 32;
constexpr int OP_BYTES = 256 * 64;
constexpr int STAGE_BYTES = 2 * OP_BYTES;
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 131072
constexpr int NWAVE = 8;
constexpr int NI = 2;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ inline int tswz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }
__device__ inline int xswz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }

struct StageCtx {
  const bf16* p[NI];
  int klim[NI];
};

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
__device__ inline void stage_op(const StageCtx& c, int k0, int64_t ld, char* tile, int wave) {
  const int64_t koff = TR ? (int64_t)k0 * ld : (int64_t)k0;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const void* src = (k0 < c.klim[it]) ? (const void*)(c.p[it] + koff) : (const void*)g_zero16;
    glds16(src, tile + (wave + NWAVE * it) * 1024);
  }
}

template <bool TR>
__device__ inline bf16x8 frag(const char* tile, int row0, int fi, int fg) {
  if constexpr (!TR) {
    const int row = row0 + fi;
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((fg ^ xswz(row)) << 4));
  } else {
    union {
      bf16x8 v;
      s16x4_t h[2];
    } u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = fg * 8 + t * 4 + (fi >> 2);
      const int off = krow * 512 + (((row0 >> 4) ^ tswz(krow)) << 5) + ((fi & 3) << 3);
      u.h[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(tile + off));
    }
    return u.v;
  }
}

struct Problem {
  const bf16* A;
  const bf16* B;
  bf16* C;
  const bf16* R;
  float* ws;
  int64_t lda, ldb, ldc, ldr, M, N;
  int K, kps, tiles_m, tiles_n, splitk, partial, ablate;  // splitk = non-empty K-slices; partial: write fp32 slices to ws
  float alpha, beta;
};

// A work item = (output tile, K-slice); items are numbered slice-major, tiles in rasterised order.
struct Item {
  int tm, tn, z, kbeg, kend;
};
__device__ inline Item decode_item(const Problem& P, int item) {
  const int ntiles = P.tiles_m * P.tiles_n;
  Item it;
  it.z = item / ntiles;
  gemm_tile_of(item - it.z * ntiles, P.tiles_m, P.tiles_n, 4, it.tm, it.tn);
  it.kbeg = it.z * P.kps;
  it.kend = (it.kbeg + P.kps < P.K) ? it.kbeg + P.kps : P.K;
  return it;
}

template <bool TA, bool TB>
__global__ __launch_bounds__(512, 2) void gemm_pp256p_kernel(const Problem P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;  // M half and ping-pong group (waves w, w+4 share a SIMD)
  const int wn = wave & 3;    // N quarter
  const int fi = lane & 15, fg = lane >> 4;

  // this workgroup's item list: XCD x owns a contiguous range of items; its `nslot` workgroups take them round robin
  const int ntiles = P.tiles_m * P.tiles_n;
  const int nitems = ntiles * P.splitk;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int q = nitems >> 3, r8 = nitems & 7;
  const int xbeg = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + slot;  // my first item
  const int xcnt = q + (xcd < r8 ? 1 : 0);
  const int my_items = (xcnt > slot) ? (xcnt - slot + nslot - 1) / nslot : 0;

  // total K-steps of this workgroup: every slice has kps/32 steps except the last one (K tail)
  const int nt_full = P.kps / QBK, nt_last = (P.K - (P.splitk - 1) * P.kps + QBK - 1) / QBK;
  int G = 0;
  for (int i = 0, id = xbeg; i < my_items; ++i, id += nslot) G += (id >= (P.splitk - 1) * ntiles) ? nt_last : nt_full;

  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- issue cursor: the (item, step) whose LDS-DMA is issued next (three steps ahead of the multiply) --------
  StageCtx ca, cb;
  int is_id = xbeg, is_left = 0, is_k0 = 0, is_g = 0;
  auto issue_open = [&]() __attribute__((always_inline)) {  // staging context of item is_id
    const Item it = decode_item(P, is_id);
    is_left = (it.kend - it.kbeg + QBK - 1) / QBK;
    is_k0 = it.kbeg;
    if constexpr (TA) stage_init_t(ca, P.A, P.lda, (int64_t)it.tm * QBM, P.M, it.kend, wave, lane);
    else stage_init_n(ca, P.A, P.lda, (int64_t)it.tm * QBM, P.M, it.kend, wave, lane);
    if constexpr (TB) stage_init_t(cb, P.B, P.ldb, (int64_t)it.tn * QBN, P.N, it.kend, wave, lane);
    else stage_init_n(cb, P.B, P.ldb, (int64_t)it.tn * QBN, P.N, it.kend, wave, lane);
  };
  auto issue_step = [&]() __attribute__((always_inline)) {  // the 4 LDS-DMA instructions of global step is_g
    if (is_g >= G) return;
    if (is_left == 0) issue_open();
    if (!((P.ablate & 1) && is_g >= 3)) {  // (ablate: micro-benchmark only, no tile loads after the pipeline fill)
      char* buf = smem + (is_g & (NSTAGE - 1)) * STAGE_BYTES;
      stage_op<TA>(ca, is_k0, P.lda, buf, wave);
      stage_op<TB>(cb, is_k0, P.ldb, buf + OP_BYTES, wave);
    }
    ++is_g;
    is_k0 += QBK;
    if (--is_left == 0) is_id += nslot;
  };

  // ---- compute cursor -------------------------------------------------------------------------------------
  int cp_id = xbeg, cp_left = 0;
  int cp_tm = 0, cp_tn = 0, cp_z = 0;     // item being multiplied
  int fin_tm = 0, fin_tn = 0, fin_z = 0;  // finished item whose accumulators still wait to be written
  bool fin_pending = false;
  int sp = 0;  // wait_step calls that still have the 32 epilogue stores behind the step they wait for

  // Epilogue of the finished item.  Runs in the LOAD segment that follows the item's last MFMA segment, i.e. while
  // the other wave group of the SIMD pair multiplies.  Interior tiles take a path with exactly 32 store instructions
  // per wave, which lets wait_step() leave them in flight (s_waitcnt vmcnt counts loads and stores in issue order).
  auto store_item = [&]() __attribute__((always_inline)) {
    fin_pending = false;
    const bool partial = (P.partial != 0);
    const bool has_res = (P.R != nullptr && P.beta != 0.f);
    const bool vec_ok = partial ? ((P.N & 3) == 0 && ((uintptr_t)P.ws & 15) == 0)
                                : ((P.ldc & 3) == 0 && ((uintptr_t)P.C & 7) == 0 &&
                                   (!has_res || ((P.ldr & 3) == 0 && ((uintptr_t)P.R & 7) == 0)));
    const int64_t m0 = (int64_t)fin_tm * QBM, n0 = (int64_t)fin_tn * QBN;
    const bool interior = (m0 + QBM <= P.M) && (n0 + QBN <= P.N);
    const int64_t mb = m0 + grp * 128 + fi, nb = n0 + wn * 64 + fg * 4;
    if (interior && vec_ok) {
      if (partial) {
        float* dst = P.ws + (int64_t)fin_z * P.M * P.N + mb * P.N + nb;
#pragma unroll
        for (int fm = 0; fm < 8; ++fm)
#pragma unroll
          for (int fn = 0; fn < 4; ++fn) {
            *reinterpret_cast<f32x4*>(dst + (int64_t)fm * 16 * P.N + fn * 16) = acc[fn][fm];
            acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      } else {
        bf16* dst = P.C + mb * P.ldc + nb;
        const bf16* rsd = P.R + mb * P.ldr + nb;
#pragma unroll
        for (int fm = 0; fm < 8; ++fm)
#pragma unroll
          for (int fn = 0; fn < 4; ++fn) {
            f32x4 v = acc[fn][fm];
            acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = P.alpha * v[e];
            if (has_res) {
              bf16x4 rv = *reinterpret_cast<const bf16x4*>(rsd + (int64_t)fm * 16 * P.ldr + fn * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += P.beta * (float)rv[e];
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
            *reinterpret_cast<bf16x4*>(dst + (int64_t)fm * 16 * P.ldc + fn * 16) = o;
          }
      }
      sp = 3;
      return;
    }
    sp = 0;  // edge tile / unaligned views: the store count varies, wait_step() stays conservative
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) {
      const int64_t m = mb + fm * 16;
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        const int64_t n = nb + fn * 16;
        const f32x4 v = acc[fn][fm];
        acc[fn][fm] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (m >= P.M) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n + e >= P.N) continue;
          if (partial) {
            P.ws[(int64_t)fin_z * P.M * P.N + m * P.N + n + e] = v[e];
          } else {
            float x = P.alpha * v[e];
            if (has_res) x += P.beta * (float)P.R[m * P.ldr + n + e];
            P.C[m * P.ldc + n + e] = (bf16)x;
          }
        }
      }
    }
  };

  bf16x8 fx[8], fw[4];
  auto load_frags = [&](int g) __attribute__((always_inline)) {
    const char* tA = smem + (g & (NSTAGE - 1)) * STAGE_BYTES;
    const char* tB = tA + OP_BYTES;
    if ((P.ablate & 2) && g >= 1) return;  // micro-benchmark only: no LDS reads after the first step
#pragma unroll
    for (int f = 0; f < 4; ++f) fw[f] = frag<TB>(tB, wn * 64 + f * 16, fi, fg);
#pragma unroll
    for (int f = 0; f < 8; ++f) fx[f] = frag<TA>(tA, grp * 128 + f * 16, fi, fg);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // This wave's LDS-DMA of global step g has landed.  Issue order behind it: steps g+1, g+2 (4 instructions each)
  // and, for the three waits after a fast epilogue, its 32 stores.
  auto wait_step = [&](int g) __attribute__((always_inline)) {
    const int newer = (g + 2 < G) ? 8 : (g + 1 < G ? 4 : 0);
    if (sp > 0) {
      --sp;
      if (newer == 8) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (newer == 4) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    } else {
      if (newer == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (newer == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // Barrier 2g opens the interval in which group 0 reads stage g (LOAD) while group 1 multiplies step g-1; barrier
  // 2g+1 swaps the roles.  Both groups run the same loop body, group 1 one barrier later.  Every wave has waited for
  // its own DMA of step g before barrier 2g, so the stage is complete for all readers.
  issue_step();
  issue_step();
  issue_step();
  if (grp == 1) {
    wait_step(0);
    bar();  // barrier 0
  }
  for (int g = 0; g < G; ++g) {
    if (grp == 0) wait_step(g);
    bar();  // group 0: barrier 2g, group 1: barrier 2g+1
    // ---- LOAD segment: fragments of step g, LDS-DMA of step g+3 (its stage was vacated before barrier 2g),
    //      then the epilogue of an item finished by the previous MFMA segment
    load_frags(g);
    issue_step();
    if (fin_pending) store_item();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (grp == 1) wait_step(g + 1);
    bar();  // group 0: barrier 2g+1, group 1: barrier 2g+2
    // ---- MFMA segment
    __builtin_amdgcn_s_setprio(1);
    if (!(P.ablate & 4))  // (micro-benchmark only: no multiply)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 8; ++fm)
          acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[fn], fx[fm], acc[fn][fm], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    if (cp_left == 0) {  // first step of an item
      const Item it = decode_item(P, cp_id);
      cp_tm = it.tm; cp_tn = it.tn; cp_z = it.z;
      cp_left = (it.kend - it.kbeg + QBK - 1) / QBK;
      cp_id += nslot;
    }
    if (--cp_left == 0) {  // last step: hand the accumulators to the epilogue
      fin_tm = cp_tm; fin_tn = cp_tn; fin_z = cp_z;
      fin_pending = true;
    }
  }
  if (grp == 0) bar();  // barrier 2G
  if (fin_pending) store_item();
}

template <bool TA, bool TB>
int launch_one(const Problem& P, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp256p_kernel<TA, TB>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      mh_set_error("gemm_pp256p: cannot raise dynamic LDS to %d bytes: %s", LDS_BYTES, hipGetErrorString(e));
      return MH_ERR_LAUNCH;
    }
    attr_set = true;
  }
  gemm_pp256p_kernel<TA, TB><<<256, 512, LDS_BYTES, st>>>(P);  // one workgroup per CU
  MH_LAUNCH_CHECK();
  return MH_OK;
}

}  // namespace

extern int g_mh_gemm_ablate;  // api.cpp

// called by gemm.hip after argument validation (bf16 only)
int mh_gemm_pp256p_bf16(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                        const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                        void* workspace, hipStream_t st) {
  Problem P;
  P.A = (const bf16*)A; P.B = (const bf16*)B; P.C = (bf16*)C; P.R = (const bf16*)R; P.ws = (float*)workspace;
  P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldr = ldr; P.M = M; P.N = N;
  MH_REQUIRE(K < (1ll << 30), "gemm: K too large");
  P.K = (int)K;
  P.kps = (int)(((K + splitk - 1) / splitk + QBK - 1) / QBK * QBK);
  const int used = (int)((K + P.kps - 1) / P.kps);  // slices that hold at least one contraction element
  if (used < splitk) {                               // the reduce kernel sums all `splitk` slices: clear the rest
    hipError_t e = hipMemsetAsync((float*)workspace + (int64_t)used * M * N, 0, (size_t)(splitk - used) * M * N * 4, st);
    MH_REQUIRE(e == hipSuccess, "gemm: clearing empty split-K slices failed: %s", hipGetErrorString(e));
  }
  P.partial = splitk > 1;
  P.tiles_m = (int)((M + QBM - 1) / QBM);
  P.tiles_n = (int)((N + QBN - 1) / QBN);
  P.splitk = used > 0 ? used : 1;
  P.alpha = alpha; P.beta = beta;
  P.ablate = g_mh_gemm_ablate;
  if (ta && tb) return launch_one<true, true>(P, st);
  if (ta) return launch_one<true, false>(P, st);
  if (tb) return launch_one<false, true>(P, st);
  return launch_one<false, false>(P, st);
}
