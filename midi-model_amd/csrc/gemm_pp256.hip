// Ping-pong bf16 projection GEMM, 256x256 block tile (third structure): C = alpha * opA(A) * opB(B)^T + beta * R.
//
// What the measurements of the first two structures said (profiles/r01_run3_*):
//   * 128x128 tiles move 64 FLOP per byte staged L2 -> LDS: at 780 TFLOP/s the kernel already pulls 12 TB/s out of
//     L2, and ablating either the loads or the MFMAs only removes 30 % of the time -> both sides matter;
//   * every global_load_lds instruction costs its wave ~100+ issue cycles, so the loads per MFMA must drop;
//   * a workgroup whose two waves per SIMD run in lockstep never overlaps LDS reads with MFMAs (256x128 pipelined
//     kernel: slower than two independent 128x128 blocks), while a ping-pong schedule does.
// This kernel therefore uses
//   * a 256x256 tile (128 FLOP per staged byte), 8 waves = 2 groups (M halves) x 4 (N quarters), wave tile 128x64:
//     per 32-deep K-step a wave issues 4 LDS-DMA + 12 ds_read_b128 for 32 MFMAs (16x16x32);
//   * K-step 32 (64-byte LDS rows) so a step's fragments are 48 VGPRs and four 32 KiB stages fit (128 KiB):
//     three K-steps of LDS-DMA stay in flight, waited with counted vmcnt only;
//   * the ping-pong schedule: waves w and w+4 share a SIMD and alternate LOAD(t) / MFMA(t) segments one segment
//     apart, one workgroup barrier per segment (ordering argument at the barriers below);
//   * 64-byte-row LDS layout: chunk c of row r sits at r*64 + ((c ^ X[(r>>2)&3]) << 4), X = {0,3,2,1}: the four
//     16-lane groups of a ds_read_b128 fragment read (rows i, chunk g) each hit 16 distinct 16-byte slots of the
//     256-byte bank row; contraction-major operands keep the 32-byte-granule swizzle and ds_read_b64_tr_b16.
// r03: products whose A operand is row-major (every forward projection, every dgrad) run the SECOND main loop further down
// (K-step 64, whole 128-byte lines per LDS-DMA piece, partial buffer re-fill); the loop described above still serves the weight
// gradients (both operands contraction-major: their 512-byte k-rows were whole lines all along) and option "gemm_k64" = 0.
// Roofline: MFMA, 2.5 PFLOP/s dense bf16.
#include <limits.h>

#include <type_traits>

#include "common.h"

namespace {

// source of out-of-range chunks; reaches the kernel as an argument (taking the symbol's address inside the K loop cost a
// scalar load and an lgkmcnt(0) wait -- for every outstanding fragment read as well -- in front of each step's LDS-DMA)
__device__ __attribute__((aligned(16))) char g_zero16[16];

constexpr int QBM = 256, QBN = 256, QBK = 32;
constexpr int OP_BYTES = 256 * 64;             // one operand's K-step: 256 rows x 64 B (or 32 k-rows x 512 B)
constexpr int STAGE_BYTES = 2 * OP_BYTES;      // 32 KiB
constexpr int NSTAGE = 4;
constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;  // 131072
constexpr int NWAVE = 8;
constexpr int NI = 2;                          // LDS-DMA instructions per wave per operand per K-step

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ inline int tswz(int krow) { return (krow & 3) | ((krow >> 1) & 4); }
__device__ inline int xswz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }  // {0,3,2,1}[(row>>2)&3]

struct StageCtx {
  const bf16* p[NI];
  int klim[NI];
};

// row-major operand X[r][k]: K-step tile 256 rows x 64 B; one LDS-DMA instruction = 16 rows
__device__ inline void stage_init_n(StageCtx& c, const bf16* __restrict__ base, int64_t ld, int64_t row0, int64_t nrows,
                                    int64_t kend, int wave, int lane, bool fullline = false, int64_t jump128 = 0) {
  const int rsub = fullline ? (lane >> 3) : (lane >> 2), pc = fullline ? (lane & 7) : (lane & 3);
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int r = (wave + NWAVE * it) * 16 + rsub;
    const int ch = fullline ? pc : (pc ^ xswz(r));
    const int64_t grow = row0 + r + (r >= 128 ? jump128 : 0);  // (jump128: the tile's upper half comes from another row block)
    c.p[it] = base + (grow < nrows ? grow : 0) * ld + ch * 8;
    c.klim[it] = (grow < nrows) ? (int)kend - ch * 8 : INT_MIN;
  }
}
// contraction-major operand X[k][r]: K-step tile 32 k-rows x 512 B; one LDS-DMA instruction = 2 k-rows
__device__ inline void stage_init_t(StageCtx& c, const bf16* __restrict__ base, int64_t ld, int64_t r0, int64_t nrows,
                                    int64_t kend, int wave, int lane) {
  const int ksub = lane >> 5, pc = lane & 31;
#pragma unroll
  for (int it = 0; it < NI; ++it) {
    const int krow = (wave + NWAVE * it) * 2 + ksub;
    const int lg = (pc >> 1) ^ tswz(krow);
    const int64_t r = r0 + lg * 16 + (pc & 1) * 8;
    c.p[it] = base + (int64_t)krow * ld + (r < nrows ? r : 0);
    c.klim[it] = (r < nrows) ? (int)kend - krow : INT_MIN;
  }
}
template <bool TR>
__device__ inline bf16x8 frag(const char* tile, int row0, int fi, int fg) {
  if constexpr (!TR) {
    const int row = row0 + fi;
    return *reinterpret_cast<const bf16x8*>(tile + row * 64 + ((fg ^ xswz(row)) << 4));
  } else {
    // Issued from inline asm (r02): through __builtin_amdgcn_ds_read_tr16_b64 hipcc's LDS-DMA alias tracking put an
    // s_waitcnt vmcnt(0) in front of the first transpose read of every K-step -- it cannot tell the ring's stages apart --
    // which drained the two steps meant to stay in flight (dgrad and both weight-gradient forms).  The asm reads are
    // invisible to its counters: load_segment's own s_waitcnt lgkmcnt(0) + sched_barrier covers them.
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    union {
      bf16x8 v;
      u32x2_t h[2];
    } u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int krow = fg * 8 + t * 4 + (fi >> 2);
      const int off = krow * 512 + (((row0 >> 4) ^ tswz(krow)) << 5) + ((fi & 3) << 3);
      const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + off);
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[t]) : "v"(addr));
    }
    return u.v;
  }
}

// ---- epilogue helpers (r06) ---------------------------------------------------------------------------------------------
// Output rows are addressed as  wave-uniform tile base (a buffer descriptor) + a per-lane 32-bit byte offset computed ONCE +
// a scalar byte offset per store: no 64-bit VALU address arithmetic, no per-row compare -- rows past M fall outside the
// descriptor's range and the hardware drops their stores (the epilogues used to spend 2-4 VALU per output element on
// `m * ld + n`, its 64-bit compare against M and, in the RoPE epilogue, a 64-bit `m % S` per row).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
// (base and size are wave-uniform by construction -- kernel arguments, blockIdx and the wave id; the readfirstlane makes that
//  provable, or hipcc wraps every buffer access in a waterfall loop: cdna_hip_programming.md T20)
__device__ inline __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, int64_t bytes) {
  const unsigned n = bytes <= 0 ? 0u : (bytes > 0xfffffff0ll ? 0xfffffff0u : (unsigned)bytes);
  const uint64_t b = (uint64_t)(uintptr_t)base;
  const uint64_t bu = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b) |
                      ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32)) << 32);
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uintptr_t)bu), 0, (unsigned)__builtin_amdgcn_readfirstlane((int)n), 0x00020000);
}
// A 16-byte store through a tile descriptor.  The row step goes into the VGPR offset, NOT into the instruction's SGPR offset
// field: hipcc assumes that a buffer store with an SGPR soffset has read its data registers when the next instruction issues
// (GCNHazardRecognizer::createsVALUHazard exempts that form) and schedules a VALU write of the data registers right behind it --
// on gfx950 that write sporadically reached the store: the first dword of the second line of a wave's rows came out as the
// NEXT line's fp32 bits (r06, same-box runs of tools/gpu_r06_dbg1.py: 30-500 wrong elements per launch, none with this form,
// where the compiler sees the hazard and keeps its wait state).
template <int AUX>
__device__ inline void tile_store_(const u32x4& v, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, AUX);
}
#define tile_store(v, r, voff, aux) tile_store_<aux>(v, r, voff)
__device__ inline u32x4 as_u32x4(const bf16x8& v) {
  union {
    bf16x8 b;
    u32x4 u;
  } x;
  x.b = v;
  return x.u;
}
__device__ inline bf16x8 as_bf16x8(const u32x4& v) {
  union {
    bf16x8 b;
    u32x4 u;
  } x;
  x.u = v;
  return x.b;
}

// ---- second main loop (r03): K-step 64, whole-line LDS-DMA, two phases per K-tile -------------------------------------
// What r01/r02 measured on the loop below (profiles/r01_run5_gemm_pp256_ablation.txt): the complete kernel runs at the rate of
// its LDS-DMA alone (0.90 us per 32-deep step against 0.65 us for the MFMAs alone), and that rate is set by the number of
// cache-line segments a request touches -- a K-step-32 piece is 16 rows x 64 B = 16 half lines per instruction, the same KiB
// as 8 rows x 128 B takes 0.61 us.  Row-major operands therefore want 128-byte LDS rows, i.e. a 64-deep K-tile, and with
// 64 KiB per K-tile only two tiles fit beside each other.  The lookahead the feed needs (1.5-2.5 us under load) then has to
// come from re-filling a PART of a buffer as soon as its last reader is done:
//   * a K-tile lives in one of two 64 KiB buffers as four 16 KiB regions [A-h0 | A-h1 | B-h0 | B-h1]; region A-h{mh} holds
//     the 64 rows wr*128 + mh*64 + [0,64) of BOTH wave groups (wr = 0, 1), region B-h{nh} the 32 columns wn*64 + nh*32 +
//     [0,32) of all four column quarters (EPI 2: nh = gate / up), so a region is read by all eight waves in the same phase;
//   * two phases per K-tile, 32 MFMAs (16x16x32) each: P1 reads A-h0, B-h0, B-h1 (16 ds_read_b128) -> quadrants (0,0) (0,1);
//     P2 reads A-h1 (8) -> (1,0) (1,1) with the B fragments still in registers;
//   * regions are issued in the order A-h0, B-h0, B-h1, A-h1 of tile t, t+1, ... (2 LDS-DMA instructions per wave and
//     region, 8 rows x 128 B each): seven regions before phase 0, then P1(t) issues A-h1(t+1) and P2(t) issues A-h0, B-h0,
//     B-h1 of tile t+2 -- each one phase after the region's last read, three phases (1.5 K-tiles) ahead of its first read;
//   * the fragment reads are waited for at the END of the load slot (the MFMA slot opens with MFMAs, and a region's
//     reads are complete before the barrier that closes the slot they were issued in: the re-fill may follow at once);
//   * counted waits only: before the barrier that opens phase q+1 a wave waits until its own pieces of the regions phase
//     q+1 reads have landed (vmcnt(8): four regions stay in flight);
//   * the two wave groups run one barrier apart (waves w and w+4 share a SIMD): one multiplies while the other reads and
//     issues, two workgroup barriers per phase = four per 64-deep K-tile, as the K-step-32 loop has.
// LDS row r of a region keeps its 16-byte chunk c at slot c ^ ((r >> 1) & 7): the four 16-lane groups of a ds_read_b128
// fragment read (rows i, chunks 4 kb + g) hit 16 distinct slots of the 256-byte bank row (checked by enumeration, and
// SQ_LDS_BANK_CONFLICT = 0 on the device); the swizzle is applied to the SOURCE chunk of the lane-linear LDS-DMA image.
// Results are bit-identical to the K-step-32 loop's: the same 32-deep MFMA products are added in the same order.
//
// Schedules measured and dropped (profiles/r03_gemm_k64_schedules.txt; [32768 x 1024 x 16384], K-step-32 loop 1080-1130 TF):
// four phases of 16 MFMAs per K-tile with the read wait at the head of the MFMA slot 1280-1350 (the guide's 8-phase shape: the
// LDS latency sits in front of every MFMA slot and twice the barriers), the same with the wait in the load slot and one
// region more lookahead 1330-1343, THIS schedule 1380-1436, its LDS-DMA issued at the tail of the MFMA slot instead 1180, one
// region per wave in every slot (load and MFMA slots alike; MFMA-slot group issuing last 1230-1250, first 1260-1340): whatever lengthens the MFMA slot is paid in full; ONE
// barrier per phase (LDS-safe: a phase's regions have landed before its interval opens and both groups' reads are complete when
// it closes; the groups then alternate by program order only) 1320-1330: they drift into reading and issuing at the same time;
// LDS-DMA ahead of the reads in the load slot, or interleaved with them: +-0 (in the build without stamps the two overlap anyway);
// the next phase's fragment reads at the tail of the MFMA slot: +-0 alone, 8-10 % slower inside the block benchmark.  In-kernel s_memtime timeline of this schedule (all eight waves, same file):
// a load slot is 16 reads complete after ~400 cycles + the issue of 2 / 6 LDS-DMA instructions 300 / 600 (the L1 path takes a
// 1 KiB request every ~40 cycles and blocks the issuing wave meanwhile) + ~130 of waits, the MFMA slot 600-650, a barrier ~100:
// the load slot of one group is what the MFMAs of the other wait for.
// What a K = 1024 tile costs beyond its main loop (same file, passes H-J): one tile per CU takes 10 us at K = 64 and 1.45 us per
// further K-tile (1480 TFLOP/s); a build without the output stores saves 5.5 us per tile -- 256 CUs x 128 KiB written at once is
// ~6 TB/s, the memory side's write rate.  Starting the XCDs out of phase changed nothing (the drain is bound per XCD), and
// PERSISTENT workgroups (one per CU walking its XCD's tiles, the next tile's LDS-DMA issued while the stores drain) measured
// +-0 on every shape: the dispatcher already overlaps teardown and fill; the drain itself is what a K = 1024 tile waits for.
// r06, two more closed: (1) the first workgroup of every CU started (index within its XCD) x 0.25 .. 2 us late, so that an XCD's 32
// CUs reach their epilogues one after the other instead of together: +-1 % on every K = 1024 shape of the step (the drain is not a
// synchronised burst either).  (2) This loop for the WEIGHT GRADIENTS (both operands contraction-major; A regions k-major, 64
// k-rows x 256 B, fragments by ds_read_b64_tr_b16 as B's): bit-identical to the K-step-32 loop and 5-10 % SLOWER on every
// weight-gradient shape of the step (1054 against 1165 TFLOP/s at [8192 x 1024 x 32768]) -- 32 transposing reads in P1's load slot
// against 32 MFMAs; the K-step-32 loop lets its two groups drift (one barrier per step), which is what those shapes want.  Its
// ablation builds on [8192 x 1024 x 32768]: complete 497 us, MFMAs + barriers alone 337, fragment reads alone 212 (113 B/clk of
// the LDS's 128), LDS-DMA + reads 324: LDS bandwidth (96 KiB of reads + 32 KiB of DMA writes per 32-deep step) and the matrix pipe
// (1024 cycles per step and SIMD) are co-limits of the 8-wave 256x256 tile.
constexpr int P8_REGION = 128 * 128;  // 16 KiB
constexpr int P8_BUF = 4 * P8_REGION;
constexpr int P8_DBG_BYTES = 8 * 2048;  // timeline builds (ABL bit 7): 256 stamps per wave behind the two K-tile buffers

// TB: B is contraction-major ([K][N], the dgrad form).  Its two regions are then the two 32-deep K-halves of the tile, 32 k-rows
// x 512 B each in the layout of the K-step-32 loop (whole lines as they lie; both are read in P1 and free after it, like the
// column halves of the row-major form), and the fragments come out of them with ds_read_b64_tr_b16 (frag<true>).
template <int EPI, int ABL, bool TB>
__device__ __forceinline__ void p8_main_loop(char* smem, const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                             int64_t ldb, int64_t M, int64_t N, int64_t m0, int64_t n0, int64_t kbeg, int64_t kend,
                                             int wave, int lane, const void* __restrict__ zero16, f32x4 (&acc)[4][8]) {
  constexpr bool TL = (ABL & 128) != 0;   // timeline build
  constexpr int LA = 7;                   // regions issued before phase 0
  const int grp = wave >> 2, wn = wave & 3;
  const int fi = lane & 15, fg = lane >> 4;
  const int nt = (kend > kbeg) ? (int)((kend - kbeg + 63) / 64) : 0;
  const int nreg = 4 * nt;  // regions of this K range, in issue order s = 4 t + {A-h0: 0, B-h0: 1, B-h1: 2, A-h1: 3}

  // ---- timeline (ABL bit 7): lane 0 of every wave stamps s_memtime into LDS, 7 stamps per phase; workgroup 0 copies them out ----
  int tl_n = 0;
  auto stamp = [&]() {
    if constexpr (TL) {
      const uint64_t c = __builtin_readcyclecounter();
      if (lane == 0 && tl_n < 252) *reinterpret_cast<uint64_t*>(smem + 2 * P8_BUF + wave * 2048 + tl_n * 8) = c;
      ++tl_n;
    }
  };
  auto mark = [&](int slot) {  // slots 252 (loop entry), 253 (first regions landed), 254 (loop done) of the wave's stamp array
    if constexpr (TL) {
      const uint64_t c = __builtin_readcyclecounter();
      if (lane == 0) *reinterpret_cast<uint64_t*>(smem + 2 * P8_BUF + wave * 2048 + slot * 8) = c;
    }
  };
  mark(252);

  // ---- LDS-DMA sources: instruction it of a region covers its rows (wave*2 + it)*8 + (lane >> 3), chunk lane & 7 ----
  const bf16* pa[2];
  const bf16* pb[2];
  int kla[2][2], klb[2][2];  // [h][it]: kend - chunk*8 when the row exists, INT_MIN otherwise
  const int64_t hoff_a = 64 * lda, hoff_b = TB ? 32 * ldb : (EPI == 2) ? (N / 2) * ldb : 32 * ldb;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int lr = (wave * 2 + it) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((lr >> 1) & 7);
    const int64_t ra = m0 + (lr >> 6) * 128 + (lr & 63);                                       // + 64 for h = 1
    pa[it] = A + (ra < M ? ra : 0) * lda + ch * 8;
    kla[0][it] = (ra < M) ? (int)kend - ch * 8 : INT_MIN;
    kla[1][it] = (ra + 64 < M) ? (int)kend - ch * 8 : INT_MIN;
    if constexpr (TB) {  // instruction it of a K-half: k-rows (wave + 8 it) * 2 + (lane >> 5), 16-byte piece lane & 31 (stage_init_t)
      const int krow = (wave + NWAVE * it) * 2 + (lane >> 5), pc = lane & 31;
      const int lg = (pc >> 1) ^ tswz(krow);
      const int64_t r = n0 + lg * 16 + (pc & 1) * 8;
      pb[it] = B + (int64_t)krow * ldb + (r < N ? r : 0);
      klb[0][it] = (r < N) ? (int)kend - krow : INT_MIN;        // (k0 + krow < kend)
      klb[1][it] = (r < N) ? (int)kend - krow - 32 : INT_MIN;   // second K-half: k0 + 32 + krow < kend
    } else {
      // EPI 2 / 3 (r06): the 32 columns of a wave's region half are staged PERMUTED -- LDS row n2*16 + j takes column
      // (j >> 2) * 8 + n2 * 4 + (j & 3) -- so that the two fragments n2 = 0, 1 of a lane (MFMA rows j = 4 fg + e) are the EIGHT
      // CONSECUTIVE columns 8 fg .. 8 fg + 7: the SwiGLU / RoPE epilogues then work on whole 16-byte pieces in registers (gate and up,
      // a column and its rotation partner are the same lane's) and turn bf16 results through LDS.  Free: the permutation only
      // changes which 128-byte line of W a group of eight lanes requests; the fragment reads are the same conflict-free pattern.
      constexpr bool PERM = (EPI == 2 || EPI == 3);
      const int lrp = PERM ? ((lr & ~31) | (((lr & 15) >> 2) << 3) | (((lr >> 4) & 1) << 2) | (lr & 3)) : lr;
      const int64_t rb = (EPI == 2) ? n0 + lrp : n0 + (lrp >> 5) * 64 + (lrp & 31);              // + N/2 or + 32 for h = 1
      const int64_t rb1 = (EPI == 2) ? rb + N / 2 : rb + 32;
      pb[it] = B + (rb < N ? rb : 0) * ldb + ch * 8;
      klb[0][it] = (rb < N) ? (int)kend - ch * 8 : INT_MIN;
      klb[1][it] = (rb1 < N) ? (int)kend - ch * 8 : INT_MIN;
    }
  }
  // region s of the issue order
  auto issue = [&](int s) {
    const int t = s >> 2, r = s & 3;            // r: 0 A-h0, 1 B-h0, 2 B-h1, 3 A-h1
    const bool isb = (r == 1 || r == 2);
    const int h = (r >= 2) ? 1 : 0;
    const int slot = isb ? 2 + h : h;           // LDS order [A-h0 | A-h1 | B-h0 | B-h1]
    const int k0 = (int)kbeg + t * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int piece = (TB && isb) ? wave + NWAVE * it : wave * 2 + it;  // (k-rows 2 piece, 2 piece + 1 / rows 8 piece ..)
      char* dst = smem + (t & 1) * P8_BUF + slot * P8_REGION + piece * 1024;
      const bf16* p = isb ? pb[it] + (h ? hoff_b : 0) : pa[it] + (h ? hoff_a : 0);
      const int kl = isb ? klb[h][it] : kla[h][it];
      const void* src = (k0 < kl) ? (const void*)((TB && isb) ? p + (int64_t)k0 * ldb : p + k0) : zero16;
      glds16(src, dst);
    }
  };

  // ---- fragment reads ----
  const int sw = (fi >> 1) & 7;
  const int la0 = (grp * 64 + fi) * 128 + ((fg ^ sw) << 4);                     // K-block 0; K-block 1 = ^ 64
  const int lb0 = 2 * P8_REGION + (wn * 32 + fi) * 128 + ((fg ^ sw) << 4);
  bf16x8 fx[4][2], fw[2][2][2];  // fx[m4][kb]: the current A half; fw[nh][n2][kb]
  auto read_a = [&](const char* buf, int mh) {
    if ((ABL & 2) && buf != smem) return;
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
        fx[m4][kb] = *reinterpret_cast<const bf16x8*>(buf + mh * P8_REGION + m4 * 2048 + (la0 ^ (kb << 6)));
  };
  auto read_b = [&](const char* buf, int nh) {
    if ((ABL & 2) && buf != smem) return;
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if constexpr (TB) fw[nh][n2][kb] = frag<true>(buf + (2 + kb) * P8_REGION, wn * 64 + nh * 32 + n2 * 16, fi, fg);
        else fw[nh][n2][kb] = *reinterpret_cast<const bf16x8*>(buf + nh * P8_REGION + n2 * 2048 + (lb0 ^ (kb << 6)));
      }
  };
  auto mfma_q = [&](int mh, int nh) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4)
          if (!(ABL & 4))
            acc[nh * 2 + n2][mh * 4 + m4] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nh][n2][kb], fx[m4][kb], acc[nh * 2 + n2][mh * 4 + m4], 0, 0, 0);
  };
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_lds = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto wait_out = [&](int out) {  // at most `out` regions (2 LDS-DMA instructions each) of this wave stay in flight
    if (out >= 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (out == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (out == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (out == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // regions 0 .. need(q)-1 must have landed before phase q = 2 t + p reads
  auto need = [&](int q) { return (q >> 1) * 4 + ((q & 1) ? 4 : 3); };
  // regions issued once phase q has issued its share
  auto issued_after = [&](int q) {
    const int n = (q >> 1) * 4 + ((q & 1) ? 11 : 8);
    return n < nreg ? n : nreg;
  };
  // the LDS-DMA share of phase q.  STEADY: every region it names exists (no branches in the loop)
  auto issue_phase = [&](int q, auto steady) {
    constexpr bool ST = decltype(steady)::value;
    if (!(ABL & 1)) {
      const int t4 = (q >> 1) * 4;
      if ((q & 1) == 0) {
        if (ST || t4 + 7 < nreg) issue(t4 + 7);
      } else {
#pragma unroll
        for (int j = 8; j < 11; ++j)
          if (ST || t4 + j < nreg) issue(t4 + j);
      }
    }
    stamp();
  };
  auto read_phase = [&](int t, int p) {
    const char* buf = smem + (t & 1) * P8_BUF;
    if (p == 0) {
      read_b(buf, 0);
      read_b(buf, 1);
      read_a(buf, 0);
    } else {
      read_a(buf, 1);
    }
    stamp();
  };
  auto load_slot = [&](int t, int p, auto steady) {
    read_phase(t, p);
    issue_phase(2 * t + p, steady);
    wait_lds();
    stamp();
  };
  auto mfma_phase = [&](int p) {
    mfma_q(p, 0);  // (no s_setprio around them: with or without measured the same here, 1400-1450 TF at K = 16384)
    mfma_q(p, 1);
    stamp();
  };
  auto wait_next = [&](int q, auto steady) {  // before the barrier that opens phase q + 1
    if constexpr (decltype(steady)::value) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (q + 1 < 2 * nt) wait_out(issued_after(q) - need(q + 1));
    stamp();
  };
  // stamps per phase -- group 0: start, reads, issue, lds wait | barrier | after barrier, MFMAs issued, vmcnt | barrier
  //                     group 1: start, MFMAs issued | barrier | after barrier, reads, issue, lds wait, vmcnt | barrier
  auto tile_g0 = [&](int t, auto steady) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      stamp();
      load_slot(t, p, steady);
      bar();
      stamp();
      mfma_phase(p);
      wait_next(2 * t + p, steady);
      bar();
    }
  };
  // (slot 2q, between barriers 2q and 2q+1: group 0 reads phase q, group 1 multiplies phase q-1; slot 2q+1: the reverse)
  auto tile_g1 = [&](int t, auto steady) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      stamp();
      if (2 * t + p > 0) mfma_phase(p ^ 1);
      else stamp();
      bar();
      stamp();
      load_slot(t, p, steady);
      wait_next(2 * t + p, steady);
      bar();
    }
  };

#pragma unroll
  for (int s = 0; s < LA; ++s)
    if (s < nreg) issue(s);
  if (nt > 0) wait_out((nreg < LA ? nreg : LA) - need(0));
  bar();  // barrier 0: the regions of phase 0 are in LDS
  mark(253);
  const int nsteady = nt > 2 ? nt - 2 : 0;  // everything P2(t) issues exists: 4 t + 10 < 4 nt
  if (grp == 0) {
    for (int t = 0; t < nsteady; ++t) tile_g0(t, std::true_type{});
    for (int t = nsteady; t < nt; ++t) tile_g0(t, std::false_type{});
  } else {
    for (int t = 0; t < nsteady; ++t) tile_g1(t, std::true_type{});
    for (int t = nsteady; t < nt; ++t) tile_g1(t, std::false_type{});
    if (nt > 0) mfma_phase(1);
  }
  mark(254);
  if constexpr (TL) {  // (the caller copies the stamps out before its epilogue)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// NRM (r05; the RMSNorm of a forward-only block folded around its projections, LlamaRMSNorm + nn.Linear of
// TF:models/llama/modeling_llama.py:62-67, 254-256, 174-176 -- the form mh_gemm_skinny runs for decode, at prefill sizes):
//   1 = PRODUCER (o / down projection + residual, plain epilogue): besides C = A B^T + R the kernel writes, for every row and
//       every 64-column chunk, the sum of squares of the STORED bf16 values: aux[(n / 64) * M + m] (fp32).  Four dot2 + three DPP
//       adds per 128-byte line, in the registers the line is stored from.  mh_row_rstd turns the N / 64 partials of a row into
//       rstd[m] (one tiny launch) -- no pass over the residual stream at all;
//   2 = CONSUMER (EPI 2 / EPI 3: gate|up + SwiGLU, q|k|v + RoPE; W carries the norm weight folded in by the caller): every row of
//       the product is multiplied by aux[m] = rstd[m] before the epilogue's own arithmetic.  Wave 0 brings the tile's 256 values
//       into LDS with ONE LDS-DMA request ahead of the main loop's (the oldest request: the first counted wait covers it), so
//       the epilogue reads them with eight ds_read_b32 and no global latency.
template <bool TA, bool TB, int ABL, int EPI, int ML, int NRM>
__global__ __launch_bounds__(512, 2) void gemm_pp256_kernel(const bf16* __restrict__ A, int64_t lda,
                                                            const bf16* __restrict__ B, int64_t ldb, bf16* C, int64_t ldc,
                                                            const bf16* R, int64_t ldr, int64_t M, int64_t N, int64_t K,
                                                            float alpha, float beta, int tiles_n, int nwg,
                                                            int64_t k_per_split, float* __restrict__ ws,
                                                            const void* __restrict__ zero16, float* aux, int lean_epi) {
  static_assert(NRM == 0 || (ML == 1 && ABL == 0 && !TA && (!TB || (EPI == 1 && NRM == 2))), "the norm forms ride on the K-step-64 loops only");
  static_assert(NRM != 1 || EPI == 0, "sum-of-squares partials come out of the plain epilogue");
  // (r06: the row scale also on the plain epilogue -- the token-level q|k|v projection of the folded training forward -- and on the
  //  SwiGLU-backward epilogue, which then delivers d z = rstd (.) d gate|up, the gradient of the UNSCALED product x W'^T: the
  //  operand both the folded dgrad and the folded weight gradient take; engine.layer_backward_folded)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ABL != 0 builds are micro-benchmarks with wrong results (tools/bench_gemm.py): 1 = no LDS-DMA after the pipeline
  // fill, 2 = no fragment reads, 4 = no MFMA, 8 = row-major pieces fetch whole 128-byte lines (8 rows x 128 B),
  // 32 = no output stores
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;      // M half (and ping-pong group: waves w, w+4 share a SIMD)
  const int wn = wave & 3;        // N quarter

  // Work item order.  The hardware deals workgroups to the 8 XCDs round robin by linear id (x fastest, then z); an
  // XCD has its own L2.  Give XCD x a CONTIGUOUS range of items, items numbered slice-major with the tiles of one
  // K-slice in grouped raster order (gemm_tile_of): the ~32 workgroups resident on an XCD then share A/B panels.
  // Without the slice-major part the 16 slices of a [1024 x 1024 x 262144] weight gradient put every tile of a slice
  // on a different XCD and the kernel ran at the HBM/fabric rate (r01: LDS-DMA alone 529 us of 608 us).
  const int nitems = nwg * (int)gridDim.z;
  const int lin = blockIdx.x + nwg * (int)blockIdx.z;
  const int q = nitems >> 3, r8 = nitems & 7, xcd = lin & 7;
  const int item = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (lin >> 3);
  const int zslice = item / nwg;
  int tm, tn;
  gemm_tile_of(item - zslice * nwg, nwg / tiles_n, tiles_n, 4, tm, tn);
  // EPI == 2 (gate|up projection with the SwiGLU forward as epilogue, mh_gemm_swiglu): B = [gate rows; up rows] (N = 2 I);
  // a tile takes 128 gate rows and the 128 matching up rows, so that every lane ends up holding gate and up of the same
  // output element (wave wn: fragments 0,1 = gate columns wn*32.., fragments 2,3 = the same up columns)
  const int64_t m0 = (int64_t)tm * QBM, n0 = (int64_t)tn * (EPI == 2 ? 128 : QBN);

  const int64_t kbeg = (int64_t)zslice * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;

  f32x4 acc[4][8];  // [fn][fm]: 64 columns x 128 rows of this wave
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fi = lane & 15, fg = lane >> 4;
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (ML >= 1) {
    if constexpr (NRM == 2) {  // rstd[m0 .. m0 + 255] -> smem + LDS_BYTES (1 KiB behind the two K-tile buffers); M % 4 == 0 (launcher)
      if (wave == 0) {
        const int64_t r = m0 + lane * 4;
        glds16((r + 4 <= M) ? (const void*)(aux + r) : zero16, smem + LDS_BYTES);
      }
    }
    p8_main_loop<EPI, ABL, TB>(smem, A, lda, B, ldb, M, N, m0, n0, kbeg, kend, wave, lane, zero16, acc);
    if constexpr ((ABL & 128) != 0) {  // timeline build: workgroup 0 hands its stamps to the caller (ws), then the normal epilogue
      __syncthreads();
      if (blockIdx.x == 0 && blockIdx.z == 0)
        for (int i = tid; i < P8_DBG_BYTES / 8; i += 512)
          reinterpret_cast<uint64_t*>(ws)[i] = reinterpret_cast<const uint64_t*>(smem + 2 * P8_BUF)[i];
      __syncthreads();
    }
  } else {
  const int nt = (kend > kbeg) ? (int)((kend - kbeg + QBK - 1) / QBK) : 0;
  StageCtx ca, cb;
  if constexpr (TA) stage_init_t(ca, A, lda, m0, M, kend, wave, lane);
  else stage_init_n(ca, A, lda, m0, M, kend, wave, lane, (ABL & 8) != 0);
  if constexpr (TB) stage_init_t(cb, B, ldb, n0, N, kend, wave, lane);
  else stage_init_n(cb, B, ldb, n0, N, kend, wave, lane, (ABL & 8) != 0, EPI == 2 ? N / 2 - 128 : 0);
  // LDS-DMA instruction j (0,1: A; 2,3: B) of K-step t; step t lives in stage t & 3
  auto issue_one = [&](int t, int j) {
    char* buf = smem + (t & (NSTAGE - 1)) * STAGE_BYTES + (j >> 1) * OP_BYTES + (wave + NWAVE * (j & 1)) * 1024;
    const int k0 = (int)kbeg + t * QBK;
    const StageCtx& c = (j >> 1) ? cb : ca;
    const bool tr = (j >> 1) ? TB : TA;
    const int64_t ld = (j >> 1) ? ldb : lda;
    const bf16* p = c.p[j & 1] + (tr ? (int64_t)k0 * ld : (int64_t)k0);
    const void* src = (k0 < c.klim[j & 1]) ? (const void*)p : zero16;
    glds16(src, buf);
  };
  // every wave issues exactly 4 LDS-DMA instructions per K-step; three steps stay in flight
#pragma unroll
  for (int s0 = 0; s0 < 3; ++s0)
    if (s0 < nt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_one(s0, j);
    }

  bf16x8 fx[8], fw[4];
  auto load_frags = [&](int t) {
    const char* tA = smem + (t & (NSTAGE - 1)) * STAGE_BYTES;
    const char* tB = tA + OP_BYTES;
    if ((ABL & 2) && t > 0) return;
#pragma unroll
    for (int f = 0; f < 4; ++f)
      fw[f] = frag<TB>(tB, EPI == 2 ? (f >> 1) * 128 + wn * 32 + (f & 1) * 16 : wn * 64 + f * 16, fi, fg);
#pragma unroll
    for (int f = 0; f < 8; ++f) fx[f] = frag<TA>(tA, grp * 128 + f * 16, fi, fg);
  };
  // MFMA segment of step t: 32 matrix instructions and nothing else (issuing part of the LDS-DMA here, between the
  // MFMAs, measured 10% slower: an LDS-DMA issue blocks the wave for 60-180 cycles and the matrix pipe starves)
  auto mfma_all = [&](int t) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
#pragma unroll
      for (int fm = 0; fm < 8; ++fm)
        if (!(ABL & 4)) acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[fn], fx[fm], acc[fn][fm], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // Before barrier 2t every wave makes sure its own LDS-DMA of step t has landed; the 4 instructions each of steps
  // t+1 and t+2 may stay in flight.
  auto wait_step = [&](int t) {
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  // (order inside the segment: issuing the LDS-DMA before the fragment reads is neutral on row-major operands and 40 %
  // slower on contraction-major ones; waiting for the fragments before issuing it costs 1-2 %; profiles/r01_run18)
  auto load_segment = [&](int t) {
    load_frags(t);
    if (t + 3 < nt && !(ABL & 1)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_one(t + 3, j);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // (no MFMA may be scheduled above the wait: the asm reads above are opaque to hipcc)
  };
  // Barrier 2t+1 keeps the two groups strictly out of phase (group 0 reads while group 1 multiplies and vice versa); LDS
  // safety only needs barrier 2t (every wave has seen its own pieces of step t land, and has finished reading step t-1
  // whose stage the LDS-DMA of step t+3 overwrites).  With a contraction-major operand the load segment carries the
  // transposing reads (24 per wave when both operands are) and outlasts the 32 MFMAs; letting the groups drift measured
  // 9-14 % faster for weight gradients and 0-8 % for the dgrad shapes (profiles/r01_run19_gemm_single_barrier_*.txt).
  // With both operands row-major it is neutral to 3 % slower, so those keep the second barrier.  (ABL bit 64 flips the
  // choice; results are identical either way.)
  constexpr bool MIDBAR = ((ABL & 64) != 0) == (TA || TB);
  if (grp == 0) {
    for (int t = 0; t < nt; ++t) {
      wait_step(t);
      bar();  // barrier 2t
      load_segment(t);
      if (MIDBAR) bar();  // barrier 2t+1
      mfma_all(t);
    }
    bar();    // barrier 2nt
  } else {
    for (int t = 0; t < nt; ++t) {
      wait_step(t);
      bar();  // barrier 2t
      if (t > 0) mfma_all(t - 1);
      if (MIDBAR) bar();  // barrier 2t+1
      load_segment(t);
    }
    bar();    // barrier 2nt
    if (nt > 0) mfma_all(nt - 1);
  }
  }  // ML == 0

  // epilogue: lane (fi, fg) of fragment (fn, fm) holds C[m][n..n+3], m = m0 + grp*128 + fm*16 + fi
  float rs[8];  // NRM 2: rstd of the lane's eight rows
  if constexpr (NRM == 2) {
    const float* rl = reinterpret_cast<const float*>(smem + LDS_BYTES);
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) rs[fm] = rl[grp * 128 + fm * 16 + fi];
  }
  const bool partial = (gridDim.z > 1);
  float* wsz = partial ? ws + (int64_t)zslice * M * N : nullptr;
  if constexpr ((ABL & 32) != 0) {  // micro-benchmark: no output
    if (acc[0][0][0] != 1.2345e38f) return;
  }
  // Whole-line stores: the wave turns its 128 x 64 fp32 tile through its own 16 KiB of the (now idle) stage ring, 64 rows
  // at a time, so that 8 lanes write one 128-byte line of C (and read one of R) instead of 16 rows x 32 bytes per
  // instruction: the [32768 x 3072 x 1024] projection went from 222 to 194 us (profiles/r01_run15, r01_run16; with no
  // stores at all it takes 168).  Row r of the half lives at r * 256 B, 16-byte chunk c at slot c ^ (r & 15): fragment
  // writes (16 rows x 4 chunks) and row reads (8 rows x 16 chunks) are bank-conflict free.  LDS operations of one wave
  // execute in order, so no barrier is needed.  C is not read again by this kernel: its stores are marked non-temporal
  // (1-2 % on the K = 1024 shapes).  Split-K partials keep the direct fragment stores below (16 rows x 64 B per instruction
  // is already half lines; turning them through LDS measured 5 % slower inside the training step).
  // Rounding the tile to bf16 before the turn (half the LDS traffic when there is no residual) measured no faster.
  if constexpr (EPI == 1) {
    if (lean_epi && m0 + QBM <= M && n0 + QBN <= N && ldc < (1 << 22) && ldr < (1 << 22)) {
      // r06: interior tile -- the form below with descriptor addressing (tile_rsrc: no 64-bit address arithmetic or bounds
      // tests per line) and the SwiGLU derivative on two-element vectors (mh_dswiglu8: the epilogue is VALU-bound, ~1900
      // instructions per wave before).  Same operations, same roundings.
      char* wreg = smem + wave * 16384;
      const int lrow = lane >> 3, c = lane & 7;
      const __amdgpu_buffer_rsrc_t rg = tile_rsrc(R + m0 * ldr + n0 + wn * 64, (255 * ldr + N + 64) * 2);
      const __amdgpu_buffer_rsrc_t rc = tile_rsrc(C + m0 * ldc + n0 + wn * 64, (255 * ldc + N + 64) * 2);
      const unsigned vo_r = (unsigned)(((grp * 128 + lrow) * (int)ldr + c * 8) * 2), so_r = (unsigned)ldr * 16u;  // + 8 rows per step
      const unsigned vo_c = (unsigned)(((grp * 128 + lrow) * (int)ldc + c * 8) * 2), so_c = (unsigned)ldc * 16u;
      const unsigned up_off = (unsigned)N * 2u;  // the up / d up half of a row
      // gate / up lines: the first half's are requested before its turn, the second half's in two groups of four as the first
      // half's lines leave their registers (all sixteen at once, as the general form does, spills beside the fast path's code)
      bf16x8 gq[2][8], uq[2][8];
      auto request = [&](int half, int i0) {
#pragma unroll
        for (int i = i0; i < i0 + 4; ++i) {
          gq[half][i] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rg, vo_r, (unsigned)(half * 8 + i) * so_r, 0));
          uq[half][i] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rg, vo_r, (unsigned)(half * 8 + i) * so_r + up_off, 0));
        }
      };
      request(0, 0);
      request(0, 4);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
          for (int fn = 0; fn < 4; ++fn) {
            const int row = fmh * 16 + fi, ch = fn * 4 + fg;
            f32x4 t4 = acc[fn][half * 4 + fmh];
            if constexpr (NRM == 2) t4 *= rs[half * 4 + fmh];  // (d a scaled by the row's rstd: SwiGLU' is linear in it)
            *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = t4;
          }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (half == 0 && (i == 2 || i == 6)) request(1, i == 2 ? 0 : 4);
          const int row = i * 8 + lrow;
          const f32x4 lo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
          const f32x4 hi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
          const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          float dv[8], g[8], u[8], dg[8], du[8];
          expand8_bf16(cvt8_bf16(v), dv);  // d a rounded to bf16 first, as the unfused pair of launches stores it
          expand8_bf16(gq[half][i], g);
          expand8_bf16(uq[half][i], u);
          mh_dswiglu8(dv, g, u, dg, du);
          const unsigned so = (unsigned)(half * 8 + i) * so_c;  // (added to the VGPR offset: see tile_store)
          tile_store(as_u32x4(cvt8_bf16(dg)), rc, vo_c + so, 2 /* nt */);
          tile_store(as_u32x4(cvt8_bf16(du)), rc, vo_c + so + up_off, 2);
        }
      }
      return;
    }
    // SwiGLU backward as the epilogue of down_proj's dgrad (mh_gemm_dswiglu): the tile is d a = dx * Wd (M x I); R holds
    // gate|up of the forward ([M, 2 I]) and C receives d gate | d up.  Same whole-line turn through LDS; the gate and up
    // lines of a 64-row half are requested before its turn.  d a is rounded to bf16 first, as the unfused pair of
    // launches (mh_gemm + mh_swiglu_bwd) stores it.
    char* wreg = smem + wave * 16384;
    const int lrow = lane >> 3, c = lane & 7;
    const int64_t n = n0 + wn * 64 + c * 8;
    const bool n_ok = (n + 8 <= N);
    // gate / up lines: the first half's are requested before its turn, the second half's right after the first half's
    // fragments have left the accumulator registers (r02: requested only when its own turn came, their latency was exposed
    // a second time per tile)
    bf16x8 gq[2][8], uq[2][8];
    auto request = [&](int half) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t m = m0 + grp * 128 + half * 64 + i * 8 + lrow;
        if (m < M && n_ok) {
          gq[half][i] = *reinterpret_cast<const bf16x8*>(R + m * ldr + n);
          uq[half][i] = *reinterpret_cast<const bf16x8*>(R + m * ldr + N + n);
        }
      }
    };
    request(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const bf16x8 (&gv)[8] = gq[half];
      const bf16x8 (&uv)[8] = uq[half];
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          const int row = fmh * 16 + fi, ch = fn * 4 + fg;
          f32x4 t4 = acc[fn][half * 4 + fmh];
          if constexpr (NRM == 2) t4 *= rs[half * 4 + fmh];
          *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = t4;
        }
      if (half == 0) request(1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = i * 8 + lrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
        const int64_t m = m0 + grp * 128 + half * 64 + row;
        if (m >= M || !n_ok) continue;
        // (alpha is 1 for this epilogue -- mh_gemm_dswiglu has no scale argument; conversions as two-element packs, the
        // sigmoid through v_rcp_f32: common.h, mh_sigmoid)
        const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        float dv[8], g[8], u[8], dg[8], du[8];
        expand8_bf16(cvt8_bf16(v), dv);  // d a rounded to bf16 first, as the unfused pair of launches stores it
        expand8_bf16(gv[i], g);
        expand8_bf16(uv[i], u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sig = mh_sigmoid(g[e]);
          const float silu = g[e] * sig;
          dg[e] = dv[e] * u[e] * (sig * (1.f + g[e] * (1.f - sig)));
          du[e] = dv[e] * silu;
        }
        const bf16x8 og = cvt8_bf16(dg), ou = cvt8_bf16(du);
        __builtin_nontemporal_store(og, reinterpret_cast<bf16x8*>(C + m * ldc + n));
        __builtin_nontemporal_store(ou, reinterpret_cast<bf16x8*>(C + m * ldc + N + n));
      }
    }
    return;
  }
  if constexpr (EPI == 2 && ML == 1) {
    // r06 form (the K-step-64 loop stages W's rows permuted: see p8_main_loop).  Lane (fi, fg) holds, for each of its eight rows
    // fm*16 + fi, gate columns 8 fg .. 8 fg + 7 of the wave's 32 in acc[0][fm] | acc[1][fm] and the matching up columns in
    // acc[2][fm] | acc[3][fm]: the whole SwiGLU chain (LlamaMLP.forward, modeling_llama.py:174-176; roundings as mh_swiglu_fwd)
    // runs in registers and only bf16 results are turned through the wave's LDS region into whole 64-byte row pieces -- a
    // quarter of the LDS traffic of the fp32 turn, one 16-byte write per 8 outputs, no address arithmetic per store.
    char* wreg = smem + wave * 16384;
    const int64_t I = N >> 1, colw = n0 + wn * 32;
    bf16* act = const_cast<bf16*>(R);
    const int64_t rows_left = (M - m0 < QBM) ? M - m0 : QBM;
    const __amdgpu_buffer_rsrc_t ra = tile_rsrc(act + m0 * ldr + colw, ((rows_left - 1) * ldr + 32) * 2);
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(C != nullptr ? C + m0 * ldc + colw : nullptr, C != nullptr ? ((rows_left - 1) * ldc + I + 32) * 2 : 0);
    const int r4 = lane >> 2, c4 = lane & 3;
    const unsigned vo_a = (unsigned)(((grp * 128 + r4) * (int)ldr + c4 * 8) * 2), so_a = (unsigned)ldr * 32u;   // + 16 rows per step
    const unsigned vo_c = (unsigned)(((grp * 128 + r4) * (int)ldc + c4 * 8) * 2), so_c = (unsigned)ldc * 32u;
    const bool keep = (C != nullptr);  // (gate|up is read again by the backward only)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh) {
        const int fm = half * 4 + fmh;
        float gf[8], uf[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gf[e] = acc[0][fm][e];
          gf[4 + e] = acc[1][fm][e];
          uf[e] = acc[2][fm][e];
          uf[4 + e] = acc[3][fm][e];
        }
        if constexpr (NRM == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            gf[e] *= rs[fm];
            uf[e] *= rs[fm];
          }
        }
        const bf16x8 og = cvt8_bf16(gf), ou = cvt8_bf16(uf);
        float gr[8], ur[8], sv[8], sr[8], av[8];
        expand8_bf16(og, gr);
        expand8_bf16(ou, ur);
        mh_silu8(gr, sv);
        expand8_bf16(cvt8_bf16(sv), sr);  // round(silu(gate)), modeling_llama.py:174-176 in bf16
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = sr[e] * ur[e];
        const bf16x8 oa = cvt8_bf16(av);
        // row r of the 64-row half: 64 bytes; 16-byte chunk c at slot c ^ ((r >> 1) & 3) (8 lanes of a write = 8 rows, one
        // chunk: distinct 16-byte slots of a 128-byte bank row)
        const int row = fmh * 16 + fi, off = row * 64 + ((fg ^ ((row >> 1) & 3)) << 4);
        // (written and read back as the SAME type: accesses of different vector types may be reordered by type-based alias analysis)
        *reinterpret_cast<u32x4*>(wreg + off) = as_u32x4(oa);
        if (keep) {
          *reinterpret_cast<u32x4*>(wreg + 4096 + off) = as_u32x4(og);
          *reinterpret_cast<u32x4*>(wreg + 8192 + off) = as_u32x4(ou);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // 16 rows x 64 B per instruction
        const int row = i * 16 + r4, off = row * 64 + ((c4 ^ ((row >> 1) & 3)) << 4);
        const unsigned so = (unsigned)(half * 4 + i);
        const u32x4 va = *reinterpret_cast<const u32x4*>(wreg + off);
        tile_store(va, ra, vo_a + so * so_a, 0);
        if (keep) {
          const u32x4 vg = *reinterpret_cast<const u32x4*>(wreg + 4096 + off);
          const u32x4 vu = *reinterpret_cast<const u32x4*>(wreg + 8192 + off);
          tile_store(vg, rc, vo_c + so * so_c, 0);
          tile_store(vu, rc, vo_c + so * so_c + (unsigned)I * 2u, 0);
        }
      }
    }
    return;
  }
  if constexpr (EPI == 2) {
    // C = gate|up [M, 2 I] (kept for the backward), R = the activation a = round(silu(gate)) * up [M, I] (written here:
    // LlamaMLP.forward, modeling_llama.py:174-176, roundings as mh_swiglu_fwd).  The wave's 64 fp32 columns are 32 gate +
    // the 32 matching up columns: 4 lanes write 64 contiguous bytes of a row to each of the three outputs.
    char* wreg = smem + wave * 16384;
    const int lrow = lane >> 2, c = lane & 3;
    const int64_t I = N >> 1, col = n0 + wn * 32 + c * 8;
    bf16* act = const_cast<bf16*>(R);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          const int row = fmh * 16 + fi, ch = fn * 4 + fg;
          f32x4 t4 = acc[fn][half * 4 + fmh];
          if constexpr (NRM == 2) t4 *= rs[half * 4 + fmh];
          *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = t4;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // 16 rows x (64 B gate, 64 B up) per pass
        const int row = i * 16 + lrow, sw = row & 15;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ sw) << 4));
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ sw) << 4));
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((8 + 2 * c) ^ sw) << 4));
        const f32x4 u1 = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((9 + 2 * c) ^ sw) << 4));
        const int64_t m = m0 + grp * 128 + half * 64 + row;
        if (m >= M || col >= I) continue;
        // (alpha is 1 for this epilogue -- mh_gemm_swiglu has no scale argument.  Eleven VALU per element instead of the 28
        // the straightforward spelling compiled to: two-element conversions, sigmoid through v_rcp_f32 -- common.h)
        const float gf[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float uf[8] = {u0[0], u0[1], u0[2], u0[3], u1[0], u1[1], u1[2], u1[3]};
        const bf16x8 og = cvt8_bf16(gf), ou = cvt8_bf16(uf);
        float gr[8], ur[8], sv[8], sr[8], av[8];
        expand8_bf16(og, gr);
        expand8_bf16(ou, ur);
#pragma unroll
        for (int e = 0; e < 8; ++e) sv[e] = mh_silu(gr[e]);
        expand8_bf16(cvt8_bf16(sv), sr);  // round(silu(gate)), modeling_llama.py:174-176 in bf16
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = sr[e] * ur[e];
        const bf16x8 oa = cvt8_bf16(av);
        if (C != nullptr) {  // (read again by the backward only; the forward-only caller passes no buffer: 1 GB less to
          *reinterpret_cast<bf16x8*>(C + m * ldc + col) = og;      //  write per net block at batch 16 x 4096 events)
          *reinterpret_cast<bf16x8*>(C + m * ldc + I + col) = ou;
        }
        *reinterpret_cast<bf16x8*>(act + m * ldr + col) = oa;
      }
    }
    return;
  }
  if constexpr (EPI == 3 && ML == 1) {
    // r06 form (W's rows staged permuted: p8_main_loop).  A wave's 64 columns are ONE head; lane (fi, fg) holds, for each of its
    // eight rows, columns 8 fg .. 8 fg + 7 of the head's first half in acc[0][fm] | acc[1][fm] and their rotation partners
    // (+ 32) in acc[2][fm] | acc[3][fm]: the projection is rounded to bf16 (as the unfused pair mh_gemm, mh_rope stores it in
    // between) and rotated in registers with the SAME products and the same fma as before (apply_rotary_pos_emb,
    // modeling_llama.py:151-169; table [pos][cos | -sin | +sin] in bf16, :126), then the bf16 rows are turned through LDS into
    // whole 128-byte lines.  Position of row m: pos0 + m % S, kept as a 32-bit running value (16 rows per step).
    const int64_t nw = n0 + wn * 64;
    if (nw >= N) return;  // (N % 64 == 0: a wave's head exists or not; no barrier follows)
    char* wreg = smem + wave * 16384;
    const bool rot = nw < 2 * (N / 3);  // (wave-uniform; the v heads take the plain path)
    const bf16* tab = R;
    const unsigned S_ = (unsigned)(ldr & 0xffffffff), pos0 = (unsigned)(ldr >> 32);
    const unsigned mrow = (unsigned)m0 + (unsigned)(grp * 128 + fi), Mu = (unsigned)M;  // (M < 2^31 - 256: launcher)
    const unsigned p0 = mrow % S_;
    bf16x8 cq[8], s1q[8], s2q[8];
    if (rot) {
#pragma unroll
      for (int fm = 0; fm < 8; ++fm) {
        unsigned pf = p0 + 16u * fm;
        if (S_ >= 128u) pf = (pf >= S_) ? pf - S_ : pf;
        else pf %= S_;
        const unsigned pidx = (mrow + 16u * fm < Mu) ? pos0 + pf : pos0;  // (rows past M are never stored: any valid table row)
        const bf16* t = tab + (size_t)pidx * 96 + fg * 8;
        cq[fm] = *reinterpret_cast<const bf16x8*>(t);
        s1q[fm] = *reinterpret_cast<const bf16x8*>(t + 32);
        s2q[fm] = *reinterpret_cast<const bf16x8*>(t + 64);
      }
    }
#pragma unroll
    for (int fm = 0; fm < 8; ++fm) {
      float x1[8], x2[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x1[e] = acc[0][fm][e];
        x1[4 + e] = acc[1][fm][e];
        x2[e] = acc[2][fm][e];
        x2[4 + e] = acc[3][fm][e];
      }
      if constexpr (NRM == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          x1[e] *= rs[fm];
          x2[e] *= rs[fm];
        }
      }
      bf16x8 b1 = cvt8_bf16(x1), b2 = cvt8_bf16(x2);  // the projection as the unfused path stores it
      if (rot) {
        float v1[8], v2[8], cf[8], s1[8], s2[8], o1[8], o2[8];
        expand8_bf16(b1, v1);
        expand8_bf16(b2, v2);
        expand8_bf16(cq[fm], cf);
        expand8_bf16(s1q[fm], s1);
        expand8_bf16(s2q[fm], s2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o1[e] = __builtin_fmaf(v1[e], cf[e], v2[e] * s1[e]);
          o2[e] = __builtin_fmaf(v2[e], cf[e], v1[e] * s2[e]);
        }
        b1 = cvt8_bf16(o1);
        b2 = cvt8_bf16(o2);
      }
      // row r of the wave's 128: 128 bytes, 16-byte chunk c at slot c ^ (r & 7) (8 lanes of a write = 8 rows, one chunk)
      const int row = fm * 16 + fi;
      // (written and read back as the SAME type: see the SwiGLU form)
      *reinterpret_cast<u32x4*>(wreg + row * 128 + ((fg ^ (row & 7)) << 4)) = as_u32x4(b1);
      *reinterpret_cast<u32x4*>(wreg + row * 128 + (((4 + fg) ^ (row & 7)) << 4)) = as_u32x4(b2);
    }
    const int64_t rows_left = (M - m0 < QBM) ? M - m0 : QBM;
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(C + m0 * ldc + nw, ((rows_left - 1) * ldc + 64) * 2);
    const int r8 = lane >> 3, c8 = lane & 7;
    const unsigned vo = (unsigned)(((grp * 128 + r8) * (int)ldc + c8 * 8) * 2), so8 = (unsigned)ldc * 16u;  // + 8 rows per step
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // 8 rows x 128 B per instruction
      const int row = i * 8 + r8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(wreg + row * 128 + ((c8 ^ (row & 7)) << 4));
      tile_store(v, rc, vo + (unsigned)i * so8, 2 /* nt: C is not read again by this kernel */);
    }
    return;
  }
  if constexpr (EPI == 3) {
    // q|k|v projection with the rotary embedding as its epilogue (mh_gemm_rope; LlamaAttention.forward + apply_rotary_pos_emb,
    // modeling_llama.py:151-169, 243-260): C = [q | k | v] rows of N = 3 D columns, heads of 64.  A wave's 64 columns are ONE
    // head, so the rotation partner of column d (d +- 32) sits in the same LDS row of the wave's whole-line turn.  R is the
    // table [pos][cos(32) | -sin(32) | +sin(32)] in bf16 (the reference casts cos / sin to the activation dtype,
    // modeling_llama.py:126), ldr = S | pos0 << 32, position of row m = pos0 + m % S.  The projection is rounded to bf16
    // before the rotation, as the unfused pair of launches (mh_gemm, mh_rope) stores it in between; products as mh_rope's.
    // alpha is 1.  The v heads take the plain path.
    char* wreg = smem + wave * 16384;
    const int lrow = lane >> 3, c = lane & 7;
    const int64_t n = n0 + wn * 64 + c * 8;
    const bool rot = (n0 + wn * 64) < 2 * (N / 3);  // (wave-uniform)
    const bf16* tab = R;
    const int S_ = (int)(ldr & 0xffffffff), pos0 = (int)(ldr >> 32);
    const int fo = (c & 3) * 8, so = (c < 4 ? 32 : 64) + fo;
    bf16x8 cq[2][8], sq[2][8];  // the table lines of a 64-row half: the first half's requested before its turn, the second
    auto request = [&](int half) {  // half's as soon as the first half's fragments have left the accumulator registers
      if (rot) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t m = m0 + grp * 128 + half * 64 + i * 8 + lrow;
          const int64_t pos = pos0 + (m < M ? m : M - 1) % S_;
          cq[half][i] = *reinterpret_cast<const bf16x8*>(tab + pos * 96 + fo);
          sq[half][i] = *reinterpret_cast<const bf16x8*>(tab + pos * 96 + so);
        }
      }
    };
    request(0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const bf16x8 (&cv)[8] = cq[half];
      const bf16x8 (&sv)[8] = sq[half];
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          const int row = fmh * 16 + fi, ch = fn * 4 + fg;
          f32x4 a4 = acc[fn][half * 4 + fmh];
          if constexpr (NRM == 2) a4 *= rs[half * 4 + fmh];
          // rounded to bf16 here, once per element (both the owner and the partner lane read it back)
          const bf16x2 lo = __builtin_convertvector(f32x2{a4[0], a4[1]}, bf16x2), hi2 = __builtin_convertvector(f32x2{a4[2], a4[3]}, bf16x2);
          f32x4 r4;
          r4[0] = (float)lo[0];
          r4[1] = (float)lo[1];
          r4[2] = (float)hi2[0];
          r4[3] = (float)hi2[1];
          *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = r4;
        }
      if (half == 0) request(1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // 8 rows x 128 B per instruction
        const int row = i * 8 + lrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
        const int64_t m = m0 + grp * 128 + half * 64 + row;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (rot) {
          const int pc = c ^ 4;
          const f32x4 plo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * pc) ^ (row & 15)) << 4));
          const f32x4 phi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * pc + 1) ^ (row & 15)) << 4));
          const float pv[8] = {plo[0], plo[1], plo[2], plo[3], phi[0], phi[1], phi[2], phi[3]};
          float cf[8], sf[8];
          expand8_bf16(cv[i], cf);
          expand8_bf16(sv[i], sf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], cf[e], pv[e] * sf[e]);
        }
        if (m >= M || n + 8 > N) continue;
        __builtin_nontemporal_store(cvt8_bf16(v), reinterpret_cast<bf16x8*>(C + m * ldc + n));
      }
    }
    return;
  }
  const bool use_r = (R != nullptr && beta != 0.f);
  const bool line_ok = !partial && (ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 &&
                       (!use_r || ((ldr & 7) == 0 && ((uintptr_t)R & 15) == 0));
  if (lean_epi && line_ok && m0 + QBM <= M && n0 + QBN <= N && ldc < (1 << 22) && ldr < (1 << 22)) {
    // r06: an INTERIOR tile (every tile of the benchmarked shapes but the ragged logits columns): the whole-line turn below with
    // its addresses as descriptor + one per-lane offset + scalar offsets (tile_rsrc), no bounds tests.  Same arithmetic.
    char* wreg = smem + wave * 16384;
    const int lrow = lane >> 3, c = lane & 7;
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(C + m0 * ldc + n0 + wn * 64, (255 * ldc + 64) * 2);
    const __amdgpu_buffer_rsrc_t rr = tile_rsrc(use_r ? R + m0 * ldr + n0 + wn * 64 : nullptr, use_r ? (255 * ldr + 64) * 2 : 0);
    const unsigned vo_c = (unsigned)(((grp * 128 + lrow) * (int)ldc + c * 8) * 2), so_c = (unsigned)ldc * 16u;  // + 8 rows per step
    const unsigned vo_r = (unsigned)(((grp * 128 + lrow) * (int)ldr + c * 8) * 2), so_r = (unsigned)ldr * 16u;
    float* auxp = nullptr;
    if constexpr (NRM == 1) auxp = aux + ((n0 + wn * 64) >> 6) * M + m0 + grp * 128 + lrow;
    // (one loop per residual choice: `use_r` comes out of a float compare, which hipcc evaluates on the VALU and then treats as
    //  divergent -- an exec-mask branch around every store)
    auto turn = [&](auto with_r) {
    constexpr bool use_r = decltype(with_r)::value;
    bf16x8 rv[2][8];
    if (use_r) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          rv[half][i] = as_bf16x8(__builtin_amdgcn_raw_buffer_load_b128(rr, vo_r, (unsigned)(half * 8 + i) * so_r, 0));
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          const int row = fmh * 16 + fi, ch = fn * 4 + fg;
          f32x4 t4 = acc[fn][half * 4 + fmh];
          if constexpr (NRM == 2) t4 *= rs[half * 4 + fmh];  // (plain epilogue with a row scale: mh_gemm_nt_scaled)
          *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = t4;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // 8 rows x 128 B per instruction
        const int row = i * 8 + lrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (use_r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = alpha * v[e] + beta * (float)rv[half][i][e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = alpha * v[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        tile_store(as_u32x4(o), rc, vo_c + (unsigned)(half * 8 + i) * so_c, 2 /* nt */);
        if constexpr (NRM == 1) {  // (see the general form below)
          union {
            bf16x8 v8;
            bf16x2 h[4];
          } u;
          u.v8 = o;
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) ss = __builtin_amdgcn_fdot2_f32_bf16(u.h[e], u.h[e], ss, false);
          ss += dpp_move<0xB1>(ss);
          ss += dpp_move<0x4E>(ss);
          ss += dpp_move<0x141>(ss);
          if (c == 0) auxp[half * 64 + i * 8] = ss;
        }
      }
    }
    };
    if (__builtin_amdgcn_readfirstlane((int)use_r)) turn(std::true_type{});
    else turn(std::false_type{});
    return;
  }
  if (line_ok) {
    char* wreg = smem + wave * 16384;
    const int lrow = lane >> 3, c = lane & 7;
    const int64_t n = n0 + wn * 64 + c * 8;
    const bool n_full = (n + 8 <= N);
    // the residual lines of the whole tile are requested before the first turn
    bf16x8 rv[2][8];
    if (use_r) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int64_t m = m0 + grp * 128 + half * 64 + i * 8 + lrow;
          if (m < M && n_full) rv[half][i] = *reinterpret_cast<const bf16x8*>(R + m * ldr + n);
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int fmh = 0; fmh < 4; ++fmh)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          const int row = fmh * 16 + fi, ch = fn * 4 + fg;
          f32x4 t4 = acc[fn][half * 4 + fmh];
          if constexpr (NRM == 2) t4 *= rs[half * 4 + fmh];  // (plain epilogue with a row scale: mh_gemm_nt_scaled)
          *reinterpret_cast<f32x4*>(wreg + row * 256 + ((ch ^ (row & 15)) << 4)) = t4;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i) {  // 8 rows x 128 B per instruction
        const int row = i * 8 + lrow;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c) ^ (row & 15)) << 4));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(wreg + row * 256 + (((2 * c + 1) ^ (row & 15)) << 4));
        const int64_t m = m0 + grp * 128 + half * 64 + row;
        if (m >= M || n >= N) continue;
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (!n_full) {  // ragged last chunk of a row (N = 3406 logits)
          for (int e = 0; n + e < N; ++e) {
            float x = alpha * v[e];
            if (use_r) x += beta * (float)R[m * ldr + n + e];
            C[m * ldc + n + e] = (bf16)x;
          }
          continue;
        }
        if (use_r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = alpha * v[e] + beta * (float)rv[half][i][e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = alpha * v[e];
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
        __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(C + m * ldc + n));
        if constexpr (NRM == 1) {  // sum of squares of the line as it is stored: 4 x v_dot2c_f32_bf16, then the 8 lanes of the row
          union {
            bf16x8 v8;
            bf16x2 h[4];
          } u;
          u.v8 = o;
          float ss = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) ss = __builtin_amdgcn_fdot2_f32_bf16(u.h[e], u.h[e], ss, false);
          ss += dpp_move<0xB1>(ss);   // quad_perm(1,0,3,2)
          ss += dpp_move<0x4E>(ss);   // quad_perm(2,3,0,1)
          ss += dpp_move<0x141>(ss);  // row_half_mirror: the other quad of the 8-lane line
          if (c == 0) aux[((n0 + wn * 64) >> 6) * M + m] = ss;
        }
      }
    }
    return;
  }
  // split-K partials and unaligned outputs: straight from the fragments
  const bool vec_ok = partial ? ((N & 3) == 0)
                              : ((ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 &&
                                 (R == nullptr || ((ldr & 3) == 0 && ((uintptr_t)R & 15) == 0)));
#pragma unroll
  for (int fm = 0; fm < 8; ++fm) {
    const int64_t m = m0 + grp * 128 + fm * 16 + fi;
    if (m >= M) continue;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      const int64_t n = n0 + wn * 64 + fn * 16 + fg * 4;
      if (n >= N) continue;
      f32x4 v = acc[fn][fm];
      if constexpr (NRM == 2) v *= rs[fm];
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

}  // namespace
extern thread_local int g_mh_gemm_lean_epi;  // api.cpp: option "gemm_lean_epi" (default 1): interior tiles take the r06 forms of the plain / SwiGLU-backward epilogues
namespace {
template <bool TA, bool TB, int ABL, int EPI = 0, int ML = 0, int NRM = 0>
int launch_one(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R, int64_t ldr,
               int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk, void* workspace, hipStream_t st,
               float* aux = nullptr) {
  constexpr int LDSB = LDS_BYTES + ((ABL & 128) ? P8_DBG_BYTES : 0) + (NRM == 2 ? 1024 : 0);
  // Both caches are PER DEVICE (a symbol's address and a function attribute belong to the device that is current when they
  // are asked for): a host that drives several GPUs from one process gets each device's own, keyed by the calling thread's
  // current device -- which is the device of `st`, as for every entry point (include/midihip.h).  Racing first calls write
  // the same values.
  constexpr int MAX_DEV = 64;
  static void* zero16_of[MAX_DEV] = {};  // 16 zero bytes in device memory: the source of out-of-range LDS-DMA chunks
  static bool attr_set_of[MAX_DEV] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) {
    mh_set_error("gemm_pp256: no current device (or ordinal %d beyond %d)", dev, MAX_DEV - 1);
    return MH_ERR_LAUNCH;
  }
  void* zero16 = __atomic_load_n(&zero16_of[dev], __ATOMIC_ACQUIRE);
  if (zero16 == nullptr) {
    if (hipGetSymbolAddress(&zero16, HIP_SYMBOL(g_zero16)) != hipSuccess) {
      mh_set_error("gemm_pp256: hipGetSymbolAddress(g_zero16) failed");
      return MH_ERR_LAUNCH;
    }
    __atomic_store_n(&zero16_of[dev], zero16, __ATOMIC_RELEASE);
  }
  if (!__atomic_load_n(&attr_set_of[dev], __ATOMIC_ACQUIRE)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp256_kernel<TA, TB, ABL, EPI, ML, NRM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    if (e != hipSuccess) {
      mh_set_error("gemm_pp256: cannot raise dynamic LDS to %d bytes: %s", LDSB, hipGetErrorString(e));
      return MH_ERR_LAUNCH;
    }
    __atomic_store_n(&attr_set_of[dev], true, __ATOMIC_RELEASE);
  }
  const int64_t tiles_m = (M + QBM - 1) / QBM, tiles_n = (EPI == 2) ? N / 256 : (N + QBN - 1) / QBN;  // (EPI 2: I / 128)
  const int nwg = (int)(tiles_m * tiles_n);
  constexpr int KSTEP = (ML == 1) ? 64 : QBK;
  const int64_t kps = ((K + splitk - 1) / splitk + KSTEP - 1) / KSTEP * KSTEP;
  dim3 grid(nwg, 1, splitk);
  gemm_pp256_kernel<TA, TB, ABL, EPI, ML, NRM><<<grid, 512, LDSB, st>>>((const bf16*)A, lda, (const bf16*)B, ldb, (bf16*)C, ldc,
                                                          (const bf16*)R, ldr, M, N, K, alpha, beta, (int)tiles_n, nwg, kps,
                                                          (float*)workspace, zero16, aux, g_mh_gemm_lean_epi);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

}  // namespace

extern thread_local int g_mh_gemm_ablate;   // api.cpp
extern thread_local int g_mh_gemm_k64;      // api.cpp: row-major x row-major products take the K-step-64 main loop (default 1)

// called by gemm.hip after argument validation (bf16 only)
int mh_gemm_pp256_bf16(const void* A, int64_t lda, int ta, const void* B, int64_t ldb, int tb, void* C, int64_t ldc,
                       const void* R, int64_t ldr, int64_t M, int64_t N, int64_t K, float alpha, float beta, int splitk,
                       void* workspace, hipStream_t st) {
#define MH_PP(TA_, TB_, ABL_) \
  return launch_one<TA_, TB_, ABL_>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st)
#ifndef MH_AB_BUILDS
  MH_REQUIRE(g_mh_gemm_ablate == 0, "gemm: option gemm_ablate = %d selects a micro-benchmark build, which is only in the A/B test "
             "library (libmidihip_ab.so)", g_mh_gemm_ablate);
#else
  // timeline build: 16 KiB of stamps go to `workspace`, which split-K partials would share
  MH_REQUIRE(g_mh_gemm_ablate != 128 || (workspace != nullptr && splitk == 1), "gemm_ablate 128 (timeline) needs a workspace and splitk 1");
  // K-step-64 main loop: only these ablations exist; anything else would silently time the production kernel
  MH_REQUIRE(!(g_mh_gemm_k64 && !ta && !tb) || g_mh_gemm_ablate == 0 || g_mh_gemm_ablate == 1 || g_mh_gemm_ablate == 4 ||
                 g_mh_gemm_ablate == 128 || g_mh_gemm_ablate == 64,
             "gemm_ablate %d is not built for the K-step-64 main loop (set gemm_k64 = 0 for the K-step-32 ablations)", g_mh_gemm_ablate);
  if (ta == tb && g_mh_gemm_ablate && !(g_mh_gemm_k64 && !ta)) {  // micro-benchmark builds (wrong results), both-row-major / both-contraction-major
#define MH_AB(X_) \
  case X_:        \
    if (ta) MH_PP(true, true, X_); \
    MH_PP(false, false, X_);
    switch (g_mh_gemm_ablate) { MH_AB(1) MH_AB(3) MH_AB(4) MH_AB(5) MH_AB(8) MH_AB(12) MH_AB(32) default: break; }
#undef MH_AB
  }
  if (g_mh_gemm_ablate == 64) {  // the other mid-barrier choice (correct results), any operand layout
    if (ta && tb) MH_PP(true, true, 64);
    if (ta) MH_PP(true, false, 64);
    if (tb) MH_PP(false, true, 64);
    MH_PP(false, false, 64);
  }
#endif  // MH_AB_BUILDS
  if (ta && tb) MH_PP(true, true, 0);
  if (ta) MH_PP(true, false, 0);
  if (tb && g_mh_gemm_k64 == 1)  // dgrad form (A row-major, B contraction-major): the K-step-64 main loop
    return launch_one<false, true, 0, 0, 1>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st);
  if (tb) MH_PP(false, true, 0);
  if (g_mh_gemm_k64) {  // both operands row-major: the K-step-64 main loop
#define MH_P8(ABL_) \
  return launch_one<false, false, ABL_, 0, 1>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, alpha, beta, splitk, workspace, st)
#ifdef MH_AB_BUILDS
    switch (g_mh_gemm_ablate) {  // (micro-benchmark builds: tools/bench_gemm.py, tools/gemm_timeline.py)
      case 1: MH_P8(1);
      case 4: MH_P8(4);
      case 128: MH_P8(128);
      default: break;
    }
#endif
    MH_P8(0);
#undef MH_P8
  }
  MH_PP(false, false, 0);
#undef MH_PP
}

// gate|up = A * [Wgate; Wup]^T and a = round(silu(gate)) * up in one launch (both operands row-major); gemm.hip validates
// (rowscale != NULL: every row of the product times rowscale[m] first -- the folded RMSNorm, NRM 2 above)
int mh_gemm_pp256_swiglu_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* GU, int64_t ldgu, void* ACT,
                              int64_t ldact, int64_t M, int64_t I, int64_t K, hipStream_t st, const float* rowscale) {
  MH_REQUIRE(ldgu < (1 << 22) && ldact < (1 << 22), "gemm_swiglu: row strides beyond 2^22 elements are not supported (32-bit tile offsets)");
  if (rowscale != nullptr) {
    MH_REQUIRE(g_mh_gemm_k64 != 0 && M % 4 == 0, "gemm_swiglu: the row-scaled form needs the K-step-64 main loop and M %% 4 == 0");
    return launch_one<false, false, 0, 2, 1, 2>(A, lda, W, ldw, GU, ldgu, ACT, ldact, M, 2 * I, K, 1.f, 0.f, 1, nullptr, st,
                                                const_cast<float*>(rowscale));
  }
  if (g_mh_gemm_k64) return launch_one<false, false, 0, 2, 1>(A, lda, W, ldw, GU, ldgu, ACT, ldact, M, 2 * I, K, 1.f, 0.f, 1, nullptr, st);
  return launch_one<false, false, 0, 2>(A, lda, W, ldw, GU, ldgu, ACT, ldact, M, 2 * I, K, 1.f, 0.f, 1, nullptr, st);
}

// [q | k | v] = A * W^T with the rotary embedding applied to the q and k heads in the epilogue; gemm.hip validates
int mh_gemm_pp256_rope_bf16(const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* table,
                            int64_t S, int64_t pos0, int64_t M, int64_t N, int64_t K, hipStream_t st, const float* rowscale) {
  MH_REQUIRE(ldc < (1 << 22) && M < (int64_t(1) << 31) - 256, "gemm_rope: row stride beyond 2^22 elements / more than 2^31 rows are not supported (32-bit tile offsets and positions)");
  if (rowscale != nullptr) {
    MH_REQUIRE(g_mh_gemm_k64 != 0 && M % 4 == 0, "gemm_rope: the row-scaled form needs the K-step-64 main loop and M %% 4 == 0");
    return launch_one<false, false, 0, 3, 1, 2>(A, lda, W, ldw, C, ldc, table, S | (pos0 << 32), M, N, K, 1.f, 0.f, 1, nullptr, st,
                                                const_cast<float*>(rowscale));
  }
  if (g_mh_gemm_k64) return launch_one<false, false, 0, 3, 1>(A, lda, W, ldw, C, ldc, table, S | (pos0 << 32), M, N, K, 1.f, 0.f, 1, nullptr, st);
  return launch_one<false, false, 0, 3>(A, lda, W, ldw, C, ldc, table, S | (pos0 << 32), M, N, K, 1.f, 0.f, 1, nullptr, st);
}

// C = rowscale (.) (A * B^T), both operands row-major (K-step-64 loop): the plain epilogue with the row scale of the folded
// RMSNorm (NRM 2) -- the token-level q|k|v projection of the folded training forward, whose RoPE rides in the attention kernel
int mh_gemm_pp256_scaled_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                              int64_t K, const float* rowscale, hipStream_t st) {
  MH_REQUIRE(g_mh_gemm_k64 != 0 && M % 4 == 0, "gemm_nt_scaled: needs the K-step-64 main loop and M %% 4 == 0");
  return launch_one<false, false, 0, 0, 1, 2>(A, lda, B, ldb, C, ldc, nullptr, 0, M, N, K, 1.f, 0.f, 1, nullptr, st,
                                              const_cast<float*>(rowscale));
}

// d gate | d up = SwiGLU'(gate|up) applied to A * B^T (A row-major [M,K], B contraction-major [K,I]); gemm.hip validates
// (rowscale != NULL: d a times rowscale[m] first, i.e. the result is rowscale (.) d gate|up -- the folded norm's d z)
int mh_gemm_pp256_dswiglu_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, const void* GU, int64_t ldgu,
                               void* DGU, int64_t lddgu, int64_t M, int64_t I, int64_t K, hipStream_t st, const float* rowscale) {
  if (rowscale != nullptr) {
    MH_REQUIRE(g_mh_gemm_k64 == 1 && M % 4 == 0, "gemm_dswiglu_scaled: needs the K-step-64 dgrad loop and M %% 4 == 0");
    return launch_one<false, true, 0, 1, 1, 2>(A, lda, B, ldb, DGU, lddgu, GU, ldgu, M, I, K, 1.f, 0.f, 1, nullptr, st,
                                               const_cast<float*>(rowscale));
  }
  if (g_mh_gemm_k64 == 1) return launch_one<false, true, 0, 1, 1>(A, lda, B, ldb, DGU, lddgu, GU, ldgu, M, I, K, 1.f, 0.f, 1, nullptr, st);
  return launch_one<false, true, 0, 1>(A, lda, B, ldb, DGU, lddgu, GU, ldgu, M, I, K, 1.f, 0.f, 1, nullptr, st);
}

// C = A * B^T + R (both operands row-major, whole-line epilogue) and rowss[(n / 64) * M + m] = sum of squares of the stored row
// chunk C[m, 64 (n / 64) .. + 64) -- the statistics of the RMSNorm that follows (NRM 1 above); gemm.hip validates
int mh_gemm_pp256_rowss_bf16(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* R,
                             int64_t ldr, int64_t M, int64_t N, int64_t K, float* rowss, hipStream_t st) {
  MH_REQUIRE(g_mh_gemm_k64 != 0, "gemm_rowss: needs the K-step-64 main loop (option gemm_k64)");
  return launch_one<false, false, 0, 0, 1, 1>(A, lda, B, ldb, C, ldc, R, ldr, M, N, K, 1.f, R ? 1.f : 0.f, 1, nullptr, st, rowss);
}
