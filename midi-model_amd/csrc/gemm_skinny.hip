// Skinny projection for the decode step: C[M,N] = A[M,K] * W[N,K]^T (+ R), M <= 64 rows (one per sequence), bf16.
//
// A decode step multiplies a handful of activation rows with every weight matrix once, so the work is streaming the
// weights from HBM: 6.3 MB for q|k|v, 16.8 MB for gate|up of tv2o-medium.  The training GEMM (256x256 tile) would put
// 4-32 workgroups on 256 CUs and needs a second launch to reduce its split-K partials.  Here a workgroup owns 16 or 32
// output columns and all of K: its 4 waves interleave over the 32-deep K chunks (the workgroup reads 256 contiguous
// bytes of each weight row at a time, every lane keeps several 16-byte loads in flight) and their four partial tiles
// are summed through LDS.  The activation rows come from L2 (128 KiB for K = 1024, shared by every workgroup).
// Splitting K across workgroups instead was measured 2x slower end to end: the partial tiles have to meet through
// agent-scope fences, which on this part write back / invalidate the whole L2 of an XCD.
//
// Optional row scaling (norm_eps > 0): every output row m is multiplied by rsqrt(mean_k A[m,k]^2 + norm_eps), the sum of
// squares being taken from the A fragments the waves load anyway.  With the RMSNorm weight folded into W beforehand
// (W' = W * w_norm, decode.py) this is LlamaRMSNorm + projection in one launch: rstd * (x . (w_norm * W)) instead of
// (w_norm * round(x * rstd)) . W -- the same product up to where the bf16 roundings fall.
//
// Optional row indirection: row m of A is row row_ids[m] of a table (the token embedding looked up by the ids just
// sampled), likewise the residual with res_ids -- the embedding lookup of a token step costs no launch of its own.
//
// Epilogues:
//   MH_SKINNY_PLAIN   C = acc (+ R)
//   MH_SKINNY_GATEUP  W = [gate rows; up rows] (2N x K); a workgroup takes 16 gate and the 16 matching up columns and
//                     writes C[M, N] = round(silu(round(g))) * round(u): gate|up never goes to memory
//                     (LlamaMLP, modeling_llama.py:174-176; roundings as gemm + swiglu_fwd_kernel)
// Roofline: HBM; algorithmic bytes = rows(W) * K * 2 (weights once).
#include "common.h"

thread_local int g_skinny_mb = 0;   // mh_set_option("skinny_mb", ...): 16-row blocks per workgroup (0 = default, see the launcher)
thread_local int g_skinny_nbt = 0;  // mh_set_option("skinny_nbt", ...): 16-column blocks per workgroup of the plain form (0 = default)

namespace {

// NBT: 16-column blocks of W per workgroup (GATEUP: 1 gate + 1 up block); NW: waves per workgroup (K is dealt to them in
// 32-deep chunks: 8 chunks per wave are in flight at a time, so long contractions take 8 waves); MB: 16-row blocks of A per
// workgroup -- blockIdx.y picks the group of MB row blocks, so MB = 1 puts a 64-row step on 4x the workgroups, each reading a
// quarter of the activation rows (the four workgroups of a column block get consecutive blockIdx.x-major ids b, b + gridDim.x,
// ... : on the same XCD when gridDim.x is a multiple of 8, so the weight slab they share is fetched from HBM once).
// BCH: chunks per wave that are requested in ONE batch (the host picks the largest of 8 / 4 / 1 dividing the wave's
// chunk count).  r02 ISA of the first form: the chunk loop was rolled -- load W, load X, s_waitcnt vmcnt(0), MFMA, branch --
// i.e. one memory round trip per 32-deep chunk and wave (4 in a row at K = 1024); the kernel arguments were fetched by six
// separate s_load + wait pairs sunk into the blocks that use them; and the residual "prefetch" was converted to fp32 at once,
// which put its wait at the top of the kernel.  Now: every argument is touched at entry (one batch of scalar loads), the
// residual stays raw until the epilogue, and a batch's BCH x (NBT + MB) fragment loads are all requested before a
// sched_barrier, behind which the MFMAs run.
template <int MODE, int NBT, int NW, bool RSTD, int MB, int BCH>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const bf16* __restrict__ A, int64_t lda,
                                                          const bf16* __restrict__ W, int64_t ldw, bf16* __restrict__ C,
                                                          int64_t ldc, const bf16* __restrict__ R, int64_t ldr, int M, int N,
                                                          int K, float norm_eps, const int64_t* __restrict__ row_ids,
                                                          const int64_t* __restrict__ res_ids, int vec_store) {
  constexpr int NB = NBT * 16;
  constexpr int MR = MB * 16;  // rows of this workgroup
  // rows padded to NB + 4 floats: 16-byte aligned, so a lane's four accumulator values go in with one ds_write_b128 and a writer's
  // group comes out with one ds_read_b128 per wave (r04: 4 + 32 scalar LDS instructions per thread before)
  constexpr int NBP = NB + 4;
  __shared__ __attribute__((aligned(16))) float red[NW][MR][NBP];
  __shared__ float ssq[RSTD ? NW : 1][MR];
  // (8 waves x 64 rows x 32 columns, the largest form instantiated, is 69.6 KiB: more than the 64 KiB every other CDNA part
  // gives a workgroup, two workgroups per CU on gfx950 -- the only target; anything larger is a mistake in the launcher's table)
  static_assert(sizeof(float) * NW * MR * (NB + 4) + sizeof(float) * (RSTD ? NW : 1) * MR <= 80 * 1024,
                "gemm_skinny: static LDS of this form exceeds what the launcher's tilings were sized for");
  // all kernel arguments in one batch of scalar loads at entry (hipcc otherwise sinks each s_load into the block that first
  // uses it: a scalar-cache miss and a wait per block)
  asm volatile("" ::"s"(A), "s"(lda), "s"(W), "s"(ldw), "s"(C), "s"(ldc), "s"(R), "s"(ldr));
  asm volatile("" ::"s"(M), "s"(N), "s"(K), "s"(norm_eps), "s"(row_ids), "s"(res_ids), "s"(vec_store));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int nc_w = K / (32 * NW);  // chunks of 32 per wave (wave w takes chunks w, w+NW, ...); a multiple of BCH
  const int m0 = blockIdx.y * MR;

  // indirections first (their results are addresses): A rows / residual row through the id tables.
  // Every address is a wave-uniform base + a 32-bit element offset (the host checks that A, W, R and the tables span fewer than
  // 2^31 elements): the loads take the SGPR-base form and the prologue loses its 64-bit multiply-adds (r04, ISA of the r02 form:
  // ~60 VALU instructions of 64-bit address arithmetic sat between the argument fetch and the first operand request).
  unsigned arow_i[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + mb * 16 + fi;
    arow_i[mb] = (unsigned)((m < M) ? m : M - 1);
  }
  constexpr int CPT = (MODE == 1) ? 4 : NB / 4;  // output columns per writing thread (four threads per row)
  const int ml = threadIdx.x >> 2;               // row within the workgroup (threads < MR * 4 write)
  const int cw0 = (threadIdx.x & 3) * CPT;
  const bool writer = threadIdx.x < MR * 4 && m0 + ml < M;
  unsigned rrow_i = writer ? (unsigned)(m0 + ml) : 0u;
  if (row_ids != nullptr) {  // (uniform)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) arow_i[mb] = (unsigned)row_ids[arow_i[mb]];
  }
  if (MODE == 0 && res_ids != nullptr) rrow_i = (unsigned)res_ids[rrow_i];

  const unsigned ulda = (unsigned)lda, uldw = (unsigned)ldw, uldr = (unsigned)ldr;
  unsigned aoff[MB], woff[NBT];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) aoff[mb] = arow_i[mb] * ulda + (unsigned)(fg * 8);  // row_ids: A rows gathered from a table
#pragma unroll
  for (int nb = 0; nb < NBT; ++nb) {
    int n, nmax;
    if constexpr (MODE == 1) {  // block 0: gate row n, block 1: up row N + n
      n = nb * N + blockIdx.x * 16 + fi;
      nmax = (nb + 1) * N - 1;
    } else {
      n = blockIdx.x * NB + nb * 16 + fi;
      nmax = N - 1;
    }
    woff[nb] = (unsigned)(n < nmax ? n : nmax) * uldw + (unsigned)(fg * 8);
  }

  // The residual values are REQUESTED here and left raw (no conversion = no wait) until the epilogue: read after the
  // reduction they would put a second memory round trip on the critical path (tools/skinny_probe.hip).
  // UNCONDITIONAL (every thread, clamped address, the weight matrix standing in when there is no residual): a load inside a
  // branch is waited for where the branch ends.
  bf16 rraw[CPT];
  const bool res_full = blockIdx.x * NB + cw0 + CPT <= N;  // the whole group is inside the row (always, but for a ragged last block)
  {
    const bool use_r = MODE == 0 && R != nullptr;
    const unsigned roff = use_r ? rrow_i * uldr + (unsigned)(res_full ? blockIdx.x * NB + cw0 : 0) : 0u;
    const bf16* rsrc = (use_r ? R : W) + roff;  // (pointer + j: the CPT element loads merge into one request)
#pragma unroll
    for (int j = 0; j < CPT; ++j) rraw[j] = rsrc[j];
  }

  f32x4 acc[MB][NBT];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  float ss[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) ss[mb] = 0.f;
  for (int c0 = 0; c0 < nc_w; c0 += BCH) {
    bf16x8 wf[BCH][NBT], xf[BCH][MB];
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
      const unsigned k = (unsigned)((wave + NW * (c0 + i)) * 32);
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb) wf[i][nb] = *reinterpret_cast<const bf16x8*>(W + (woff[nb] + k));
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) xf[i][mb] = *reinterpret_cast<const bf16x8*>(A + (aoff[mb] + k));
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the batch is in flight before the first use
#pragma unroll
    for (int i = 0; i < BCH; ++i) {
#pragma unroll
      for (int nb = 0; nb < NBT; ++nb)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][nb], xf[i][mb], acc[mb][nb], 0, 0, 0);
      if constexpr (RSTD) {  // (behind the MFMAs in program order: the squares run while the matrix pipe works)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int e = 0; e < 8; ++e) ss[mb] += (float)xf[i][mb][e] * (float)xf[i][mb][e];
      }
    }
  }

  // sum the waves' partial tiles: lane (fi, fg) of acc[mb][nb] holds row mb*16+fi, columns nb*16 + 4 fg + e
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBT; ++nb)
      *reinterpret_cast<f32x4*>(&red[wave][mb * 16 + fi][nb * 16 + 4 * fg]) = acc[mb][nb];
  if constexpr (RSTD) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {  // the four lane groups hold different k of the same row
      float t = ss[mb];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (fg == 0) ssq[wave][mb * 16 + fi] = t;
    }
  }
  __syncthreads();
  if (!writer) return;  // four threads per row write the tile
  const int m = m0 + ml;
  float rs = 1.f;
  if constexpr (RSTD) {
    float t = ssq[0][ml];
#pragma unroll
    for (int w = 1; w < NW; ++w) t += ssq[w][ml];
    rs = rsqrtf(t / (float)K + norm_eps);
  }
  // The writer's CPT (gate|up: 2 x 4) sums of the waves' partials are read in ONE batch, and the group goes out as one 8- / 16-byte
  // store when it lies inside the row and the output is aligned for it (r04, ISA of the r02 form: the per-element `n < N` test
  // made every element its own sequence of eight ds_read + wait + a 2-byte store -- four LDS round trips and four store
  // instructions in a row on the launch's critical path; tools/persist_probe.hip: 4.52 us per dependent launch against 3.70 for
  // a bare tile).
  constexpr int NT = (MODE == 1) ? 8 : CPT;
  float tot[NT];
#pragma unroll
  for (int j4 = 0; j4 < NT; j4 += 4) {  // four consecutive columns per 16-byte LDS read, the waves' partials added in wave order
    const int c = (MODE == 1) ? ((j4 >> 2) * 16 + (threadIdx.x & 3) * 4) : cw0 + j4;
    f32x4 t = *reinterpret_cast<const f32x4*>(&red[0][ml][c]);
#pragma unroll
    for (int w = 1; w < NW; ++w) t += *reinterpret_cast<const f32x4*>(&red[w][ml][c]);
#pragma unroll
    for (int e = 0; e < 4; ++e) tot[j4 + e] = t[e] * rs;
  }
  constexpr int NO = (MODE == 1) ? 4 : CPT;  // outputs of this thread
  const int nbase = (MODE == 1) ? blockIdx.x * 16 + (threadIdx.x & 3) * 4 : blockIdx.x * NB + cw0;
  bf16 outv[NO];
  if constexpr (MODE == 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = (float)(bf16)tot[j], u = (float)(bf16)tot[4 + j];
      const float sg = (float)(bf16)mh_silu(g);
      outv[j] = (bf16)(sg * u);
    }
  } else {
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float r = (R != nullptr) ? (float)rraw[j] : 0.f;
      if (R != nullptr && !res_full && nbase + j < N) r = (float)R[rrow_i * uldr + (unsigned)(nbase + j)];  // (ragged last block only)
      outv[j] = (bf16)(tot[j] + r);
    }
  }
  bf16* crow = C + ((unsigned)m * (unsigned)ldc + (unsigned)nbase);
  if (vec_store && nbase + NO <= N) {
    if constexpr (NO == 4) *reinterpret_cast<uint64_t*>(crow) = *reinterpret_cast<const uint64_t*>(outv);
    else *reinterpret_cast<bf16x8*>(crow) = *reinterpret_cast<const bf16x8*>(outv);
  } else {
#pragma unroll
    for (int j = 0; j < NO; ++j)
      if (nbase + j < N) crow[j] = outv[j];
  }
}

}  // namespace

extern "C" int mh_gemm_skinny(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* R,
                              int64_t ldr, int mode, float norm_eps, const int64_t* row_ids, const int64_t* res_ids, int64_t M,
                              int64_t N, int64_t K, int dtype, void* stream) {
  MH_REQUIRE(dtype == MH_BF16, "gemm_skinny: bf16 only (the fp32 verification mode uses mh_gemm)");
  MH_REQUIRE(M > 0 && M <= 64 && N > 0 && N < (1 << 24), "gemm_skinny: needs 1 <= M <= 64 rows (M=%ld N=%ld)", (long)M, (long)N);
  MH_REQUIRE(mode == MH_SKINNY_PLAIN || mode == MH_SKINNY_GATEUP, "gemm_skinny: mode %d", mode);
  MH_REQUIRE(K > 0 && K % 256 == 0 && K < (1 << 24), "gemm_skinny: K=%ld must be a multiple of 256", (long)K);
  MH_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
             "gemm_skinny: A/W rows must be 16-byte aligned");
  MH_REQUIRE(mode != MH_SKINNY_GATEUP || R == nullptr, "gemm_skinny: the gate|up epilogue takes no residual");
  // (32-bit element offsets inside the kernel; with row_ids / res_ids the TABLES must stay below 2^31 elements too -- the caller's
  //  contract: the embedding tables of this model are 3.5 M elements)
  MH_REQUIRE((mode == MH_SKINNY_GATEUP ? 2 * N : N) * ldw < (int64_t(1) << 31) && 64 * lda < (int64_t(1) << 31) && 64 * ldc < (int64_t(1) << 31) &&
                 (R == nullptr || 64 * ldr < (int64_t(1) << 31)) && lda < (int64_t(1) << 24) && ldr < (int64_t(1) << 24),
             "gemm_skinny: operands too large for 32-bit element offsets");
  hipStream_t st = (hipStream_t)stream;
  // Tiling (r02, tools/decode_probe.py section 1b, profiles/r02_*decode_probe*): a launch spends its time in the TA/L1 path
  // of the CUs (64 B/clk each) and in fixed latencies, so the best tile is the LARGEST one that still puts about one
  // workgroup on each of the 256 CUs: two column blocks for the wide projections (q|k|v, lm_head), and as many 16-row
  // blocks per workgroup as leave >= ~256 workgroups -- q|k|v 32 rows x 32 columns (6.9 us against 8.3 for 16 x 16), gate|up
  // of the event-level stack all 64 rows (9.4 us against 13.7), the N = 1024 projections 16 x 16 (4.3 us against 7.2 for 64 rows).
  // Options "skinny_mb" / "skinny_nbt" force a form (tests, probes).
  const int row_blocks = (int)((M + 15) / 16);
  const int nbt = (mode == MH_SKINNY_GATEUP) ? 2 : (g_skinny_nbt == 1 || g_skinny_nbt == 2 ? g_skinny_nbt : (N >= 2048 ? 2 : 1));
  const int col_tiles = (mode == MH_SKINNY_GATEUP) ? (int)((N + 15) / 16) : (int)((N + 16 * nbt - 1) / (16 * nbt));
  int mb = g_skinny_mb;
  if (mb != 1 && mb != 2 && mb != 4) {
    const int groups = 256 / col_tiles;  // row groups that keep the launch at <= ~256 workgroups
    mb = groups >= 4 ? 1 : (groups >= 2 ? 2 : 4);
  }
  if (mb > row_blocks) mb = row_blocks >= 4 ? 4 : (row_blocks >= 2 ? 2 : 1);
  const int gy = (row_blocks + mb - 1) / mb;
  // a writer thread's outputs (4, or 8 with two column blocks) leave as one store when every row of C keeps them aligned
  const int group_bytes = ((mode == MH_SKINNY_GATEUP || nbt == 1) ? 4 : 8) * 2;
  const int vec = (((uintptr_t)C % group_bytes) == 0 && (ldc * 2) % group_bytes == 0) ? 1 : 0;
#define MH_SK4(MODE_, NBT_, NW_, GRID_, RSTD_, MB_, BCH_)                                                                  \
  gemm_skinny_kernel<MODE_, NBT_, NW_, RSTD_, MB_, BCH_><<<dim3((unsigned)(GRID_), (unsigned)gy), NW_ * 64, 0, st>>>(      \
      (const bf16*)A, lda, (const bf16*)W, ldw, (bf16*)C, ldc, (const bf16*)R, ldr, (int)M, (int)N, (int)K, norm_eps,      \
      row_ids, res_ids, vec)
#define MH_SK3(MODE_, NBT_, NW_, GRID_, RSTD_, MB_)                                                                        \
  do {                                                                                                                    \
    const int ncw_ = (int)(K / (32 * NW_)); /* chunks per wave; batches of the largest of 8 / 4 / 1 that divides it */    \
    /* (r04, tools/skinny_chain.py: ONE batch where the registers allow it -- the 64-row gate|up form 10.15 -> 9.38 us per     \
       dependent launch on cold weights, the K = 4096 projection 10.9 -> 10.7) */                                         \
    if (ncw_ == 16 && (NBT_ + MB_) <= 2) MH_SK4(MODE_, NBT_, NW_, GRID_, RSTD_, MB_, 16);                                  \
    else if (ncw_ % 8 == 0 && (NBT_ + MB_) <= 6) MH_SK4(MODE_, NBT_, NW_, GRID_, RSTD_, MB_, 8);                           \
    else if (ncw_ % 4 == 0) MH_SK4(MODE_, NBT_, NW_, GRID_, RSTD_, MB_, 4);                                                \
    else MH_SK4(MODE_, NBT_, NW_, GRID_, RSTD_, MB_, 1);                                                                   \
  } while (0)
#define MH_SK2(MODE_, NBT_, NW_, GRID_, RSTD_)                                                                             \
  do {                                                                                                                    \
    if (mb == 1) MH_SK3(MODE_, NBT_, NW_, GRID_, RSTD_, 1);                                                                \
    else if (mb == 2) MH_SK3(MODE_, NBT_, NW_, GRID_, RSTD_, 2);                                                           \
    else MH_SK3(MODE_, NBT_, NW_, GRID_, RSTD_, 4);                                                                        \
  } while (0)
#define MH_SK(MODE_, NBT_, NW_, GRID_)                                                                                    \
  do {                                                                                                                    \
    if (norm_eps > 0.f) MH_SK2(MODE_, NBT_, NW_, GRID_, true);                                                             \
    else MH_SK2(MODE_, NBT_, NW_, GRID_, false);                                                                           \
  } while (0)
  if (mode == MH_SKINNY_GATEUP) MH_SK(1, 2, 4, (N + 15) / 16);
  else if (nbt == 2 && N > 4096) MH_SK(0, 2, 4, (N + 31) / 32);
  else if (nbt == 2) MH_SK(0, 2, 8, (N + 31) / 32);
  else MH_SK(0, 1, 8, (N + 15) / 16);  // 16 columns x 8 waves: <= 4 chunks per wave at K = 1024, all in flight
#undef MH_SK
#undef MH_SK2
#undef MH_SK3
#undef MH_SK4
  MH_LAUNCH_CHECK();
  return MH_OK;
}
