// the structural GEMM step DESIGN section 10 specifies, built as a probe next to the production kernel.
//   C[M,N] = A[M,K] * B[N,K]^T, bf16 in / bf16 out, both operands row-major (every forward projection).
//   * 256 x 256 tile, FOUR waves (2 x 2), wave tile 128 x 128 = 16 accumulator blocks of v_mfma_f32_32x32x16_bf16 (256 accumulator
//     registers), one wave per SIMD with the whole 512-entry register file (launch_bounds(256, 1));
//   * operands staged through REGISTERS: 16 x buffer_load_dwordx4 per thread and 64-deep K-tile (whole 128-byte lines: 8 lanes per
//     row), written to LDS with ds_write_b128 one tile later -- no LDS-DMA (whose issue blocks the wave 60-180 cycles; with one wave
//     per SIMD nothing would hide that: the r01 attempt at four waves lost 10 % exactly there);
//   * two 64 KiB LDS stages, 128-byte rows with the bit-reversed XOR swizzle of common.h (fragment reads and staging writes are
//     bank-conflict free by enumeration); ONE workgroup barrier per K-tile;
//   * per 16-deep step: 16 MFMAs, the 8 fragment reads of the NEXT step, and a share of the tile's 16 ds_write_b128 / 16 loads,
//     interleaved with sched_group_barrier (1.3 other instructions per MFMA gap; the guide: <= 5 are hidden per 32-cycle gap);
//   * epilogue: the wave's tile rounded to bf16, turned through its own 32 KiB of the idle stages, whole 256-byte row segments out.
// The probe times it against mh_gemm_nt (the production 8-wave LDS-DMA ping-pong kernel) on the same buffers, interleaved, and
// checks the result against it (fp32 summation order differs: compared to bf16 rounding) and against an fp64 host product on a
// sample of elements.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm4w_probe.hip -o tools/bin/gemm4w_probe -Lmidi-model_amd -lmidihip
//        -Wl,-rpath,'$ORIGIN/../../midi-model_amd'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>
#include <vector>

#include "../include/midihip.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OPB = 256 * 128;          // one operand's K-tile: 256 rows x 128 B
constexpr int STAGE = 2 * OPB;          // 64 KiB
constexpr int LDSB = 2 * STAGE;         // 128 KiB

__device__ inline int lds_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }

__device__ inline void gemm_tile_of(int idx, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
  const int width = GM * tiles_n;
  const int gid = idx / width;
  const int first = gid * GM;
  const int gsize = (tiles_m - first < GM) ? tiles_m - first : GM;
  const int rem = idx - gid * width;
  tn = rem / gsize;
  tm = first + (rem - tn * gsize);
}

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ 0, (int)bytes, 0x00020000);
}

// SCHED: 0 = leave the order to hipcc, 1 / 2 = two sched_group_barrier interleaves (see tile_body)
template <int SCHED, bool STORE>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                                        int64_t ldb, bf16* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                        int tiles_n, int nwg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int lin = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = lin & 7;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (lin >> 3);
  int tm, tn;
  gemm_tile_of(item, nwg / tiles_n, tiles_n, 4, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: thread t moves chunk (t & 7) of rows (t >> 3) + 32 j, j = 0..7, of A and of B ----
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(A, (uint32_t)((int64_t)M * lda * 2));
  const __amdgpu_buffer_rsrc_t rb = make_rsrc(B, (uint32_t)((int64_t)N * ldb * 2));
  const int srow = tid >> 3, sch = tid & 7;
  const uint32_t voa = (uint32_t)(((int64_t)srow * lda + sch * 8) * 2), vob = (uint32_t)(((int64_t)srow * ldb + sch * 8) * 2);
  const uint32_t soa0 = (uint32_t)((int64_t)m0 * lda * 2), sob0 = (uint32_t)((int64_t)n0 * ldb * 2);
  const uint32_t stepa = (uint32_t)(32 * lda * 2), stepb = (uint32_t)(32 * ldb * 2);
  const int wofs = srow * 128 + ((sch ^ lds_swz(srow)) << 4);  // + j * 4096 (+ OPB for B, + STAGE for stage 1)
  bf16x8 sa[2][8], sb[2][8];  // two register sets: tile T lives in set T & 1 from its loads (issued during tile T - 3) to its LDS
                              // writes (during tile T - 1): 1.5 K-tiles (~1.5 us) for the loads to land, which is what the L2 /
                              // fabric needs under load (the first form of this probe gave them a third of a tile and waited)
  auto load_tile = [&](int t, int set) {
    const uint32_t kb = (uint32_t)t * (BK * 2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const i32x4 va = __builtin_amdgcn_raw_buffer_load_b128(ra, voa, soa0 + kb + j * stepa, 0);
      sa[set][j] = __builtin_bit_cast(bf16x8, va);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const i32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rb, vob, sob0 + kb + j * stepb, 0);
      sb[set][j] = __builtin_bit_cast(bf16x8, vb);
    }
  };
  auto write_a = [&](int stage, int set, int j) { *reinterpret_cast<bf16x8*>(smem + stage * STAGE + wofs + j * 4096) = sa[set][j]; };
  auto write_b = [&](int stage, int set, int j) { *reinterpret_cast<bf16x8*>(smem + stage * STAGE + OPB + wofs + j * 4096) = sb[set][j]; };

  // ---- fragments: lane (i = lane & 31, kg = lane >> 5) reads row (wave rows + blk * 32 + i), chunk 2 ks + kg ----
  const int fi = lane & 31, kg = lane >> 5;
  const int c0 = kg ^ lds_swz(fi);
  const int fa = (wr * 128 + fi) * 128 + (c0 << 4);        // ^ (ks << 5), + blk * 4096
  const int fb = OPB + (wc * 128 + fi) * 128 + (c0 << 4);
  bf16x8 fx[2][4], fw[2][4];  // [buffer][block]: A (activation rows) and B (weight rows) fragments of one 16-deep step
  auto read_frags = [&](int stage, int ks, int buf) {
    const char* base = smem + stage * STAGE;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      fx[buf][b] = *reinterpret_cast<const bf16x8*>(base + (fa ^ (ks << 5)) + b * 4096);
      fw[buf][b] = *reinterpret_cast<const bf16x8*>(base + (fb ^ (ks << 5)) + b * 4096);
    }
  };

  f32x16 acc[4][4];  // [cb][rb]: block = 32 weight rows (n) x 32 activation rows (m); lane holds m = lane & 31, n = 8 q + 4 kg + e
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  auto mfma_step = [&](int buf) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int rbk = 0; rbk < 4; ++rbk)
        acc[cb][rbk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[buf][cb], fx[buf][rbk], acc[cb][rbk], 0, 0, 0);
  };
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nt = K / BK;  // (even: the loop below is unrolled by two tiles so that stages and register sets have static names)
  // prologue: tile 0 into stage 0; tiles 1 and 2 into the register sets 1 and 0
  load_tile(0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    write_a(0, 0, j);
    write_b(0, 0, j);
  }
  load_tile(1, 1);
  load_tile(2, 0);
  bar();
  read_frags(0, 0, 0);

  // one K-tile `t` living in stage CUR (a compile-time constant): four 16-deep steps of 16 MFMAs each.
  // (no conditions inside: a branch would end the basic block the interleave lives in.  Tiles past the last are loaded, written
  //  and their first fragments read like any other -- buffer loads past the end of the operand return zeros, loads past the end of
  //  a ROW fetch the next row's bytes, and nothing multiplies them)
  // fragment reads of one step in the order the MFMAs want them (weight block 0 and activation block 0 first)
  auto read_frags_ordered = [&](int stage, int ks, int buf) {
    const char* base = smem + stage * STAGE;
    fx[buf][0] = *reinterpret_cast<const bf16x8*>(base + (fa ^ (ks << 5)));
    fw[buf][0] = *reinterpret_cast<const bf16x8*>(base + (fb ^ (ks << 5)));
#pragma unroll
    for (int b2 = 1; b2 < 4; ++b2) fx[buf][b2] = *reinterpret_cast<const bf16x8*>(base + (fa ^ (ks << 5)) + b2 * 4096);
#pragma unroll
    for (int b2 = 1; b2 < 4; ++b2) fw[buf][b2] = *reinterpret_cast<const bf16x8*>(base + (fb ^ (ks << 5)) + b2 * 4096);
  };
  auto tile_body = [&](int t, auto cur_c) {
    constexpr int CUR = decltype(cur_c)::value;
    constexpr int NXT = CUR ^ 1;     // stage AND register set of tile t + 1 (t + 1 has the parity of NXT)
    // Steps 0 and 1, SCHED 1 (the better one, 1194 / 1263 TF with / without stores at K = 16384): fragment reads and staging writes
    // ALTERNATE, one LDS operation per MFMA gap.  LDS operations keep their program order -- hipcc does not move a fragment read
    // below a staging write it cannot prove disjoint -- so the source order below is the issue order ("8 reads, then 8 writes"
    // with an alternating sched_group_barrier request came out as 7 reads back to back: 1146 / 1229).
    // SCHED 2: the 8 reads in the first eight gaps (so that the last has 256 cycles to land before the step boundary), the 8 writes
    // in the last eight: 1064 / 1153 -- four waves writing 1 KiB each in every gap is 128 B/clk against the LDS's ~79 B/clk for
    // 16-byte stores: the writes want to be spread, not the reads early.
    if constexpr (SCHED == 2) {
      read_frags_ordered(CUR, 1, 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) write_a(NXT, NXT, j);
    } else {
      const char* base = smem + CUR * STAGE;
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        fx[1][b2] = *reinterpret_cast<const bf16x8*>(base + (fa ^ (1 << 5)) + b2 * 4096);
        write_a(NXT, NXT, 2 * b2);
        fw[1][b2] = *reinterpret_cast<const bf16x8*>(base + (fb ^ (1 << 5)) + b2 * 4096);
        write_a(NXT, NXT, 2 * b2 + 1);
      }
    }
    mfma_step(0);
    if constexpr (SCHED == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    } else if constexpr (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // step 1: B chunks of tile t + 1
    if constexpr (SCHED == 2) {
      read_frags_ordered(CUR, 2, 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) write_b(NXT, NXT, j);
    } else {
      const char* base = smem + CUR * STAGE;
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) {
        fx[0][b2] = *reinterpret_cast<const bf16x8*>(base + (fa ^ (2 << 5)) + b2 * 4096);
        write_b(NXT, NXT, 2 * b2);
        fw[0][b2] = *reinterpret_cast<const bf16x8*>(base + (fb ^ (2 << 5)) + b2 * 4096);
        write_b(NXT, NXT, 2 * b2 + 1);
      }
    }
    mfma_step(1);
    if constexpr (SCHED == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    } else if constexpr (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // step 2: the 16 loads of tile t + 3 into the register set tile t + 1 has just left
    read_frags_ordered(CUR, 3, 1);
    load_tile(t + 3, NXT);
    mfma_step(0);
    if constexpr (SCHED == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
    } else if constexpr (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // step 3: every wave has written its share of tile t + 1 and read all of tile t: barrier, then step 0 of tile t + 1
    bar();
    read_frags_ordered(NXT, 0, 0);
    mfma_step(1);
    if constexpr (SCHED == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    } else if constexpr (SCHED == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nt; t += 2) {
    tile_body(t, std::integral_constant<int, 0>{});
    tile_body(t + 1, std::integral_constant<int, 1>{});
  }

  if constexpr (!STORE) {  // main loop only -- every accumulator kept live (an unused block's MFMAs would be deleted: the first
                           // form of this variant "ran" at 1900 TF with 2 of its 16 blocks left)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" ::"v"(acc[a][b]));
    return;
  }
  // ---- epilogue: bf16 tile through the wave's own 32 KiB (all stages are idle: the last barrier is behind every read) ----
  __syncthreads();
  char* wreg = smem + wave * 32768;  // 128 rows (m) x 256 B (128 n), 16-byte chunk c of row r at slot c ^ (r & 15)
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int rbk = 0; rbk < 4; ++rbk) {
      const int row = rbk * 32 + fi;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x16& v = acc[cb][rbk];
        bf16x4 o;
        const bf16x2 lo = __builtin_convertvector(f32x2{v[qd * 4 + 0], v[qd * 4 + 1]}, bf16x2);
        const bf16x2 hi = __builtin_convertvector(f32x2{v[qd * 4 + 2], v[qd * 4 + 3]}, bf16x2);
        o[0] = lo[0];
        o[1] = lo[1];
        o[2] = hi[0];
        o[3] = hi[1];
        const int col = cb * 32 + qd * 8 + kg * 4;  // bf16 column inside the wave tile
        const int ch = (col >> 3) ^ (row & 15);
        *reinterpret_cast<bf16x4*>(wreg + row * 256 + (ch << 4) + ((col & 7) << 1)) = o;
      }
    }
  // (one wave's LDS operations execute in order: no barrier between its writes and its reads)
  const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int p = 0; p < 32; ++p) {
    const int row = p * 4 + lr;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(wreg + row * 256 + ((lc ^ (row & 15)) << 4));
    const int64_t m = m0 + wr * 128 + row, n = n0 + wc * 128 + lc * 8;
    __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(C + m * ldc + n));
  }
}

static float bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}

template <int SCHED, bool STORE>
static void launch(const bf16* A, const bf16* B, bf16* C, int M, int N, int K, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<SCHED, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
    attr = true;
  }
  const int tiles_m = M / BM, tiles_n = N / BN, nwg = tiles_m * tiles_n;
  gemm4w_kernel<SCHED, STORE><<<nwg, 256, LDSB, st>>>(A, K, B, K, C, N, M, N, K, tiles_n, nwg);
}

int main(int argc, char** argv) {
  struct Shape {
    int M, N, K;
  };
  std::vector<Shape> shapes = {{32768, 1024, 1024}, {32768, 1024, 4096}, {32768, 1024, 16384}, {32768, 3072, 1024},
                               {32768, 8192, 1024}, {65536, 1024, 1024}, {262144, 1024, 1024}};
  if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
  const int only = (argc >= 5) ? atoi(argv[4]) : -1;  // counters: launch ONE variant a few times, no checks (0 production, 1 sched, 2 plain, 3 no stores)
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K;
    const size_t na = (size_t)M * K, nb = (size_t)N * K, nc = (size_t)M * N;
    std::vector<uint16_t> ha(na), hb(nb);
    uint64_t seed = 0x9E3779B97F4A7C15ull ^ (uint64_t)(M * 31 + N * 7 + K);
    auto rnd = [&]() {  // uniform [-1, 1): full-range random data (the guide: zero / sign-constant fills flatter the clock)
      seed ^= seed << 13;
      seed ^= seed >> 7;
      seed ^= seed << 17;
      return (float)((double)(seed >> 11) / 9007199254740992.0 * 2.0 - 1.0);
    };
    for (auto& x : ha) x = f2bf(rnd());
    for (auto& x : hb) x = f2bf(rnd() * 0.05f);
    bf16 *dA, *dB, *dC, *dR;
    CK(hipMalloc(&dA, na * 2));
    CK(hipMalloc(&dB, nb * 2));
    CK(hipMalloc(&dC, nc * 2));
    CK(hipMalloc(&dR, nc * 2));
    CK(hipMemcpy(dA, ha.data(), na * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, nc * 2));
    if (only >= 0) {
      for (int i = 0; i < 5; ++i) {
        if (only == 0) mh_gemm_nt(dA, K, dB, K, dR, N, nullptr, 0, M, N, K, 1.f, 0.f, MH_BF16, 1, nullptr, st);
        else if (only == 1) launch<1, true>(dA, dB, dC, M, N, K, st);
        else if (only == 2) launch<2, true>(dA, dB, dC, M, N, K, st);
        else launch<1, false>(dA, dB, dC, M, N, K, st);
      }
      CK(hipStreamSynchronize(st));
      printf("ran variant %d five times\n", only);
      continue;
    }
    // correctness
    launch<1, true>(dA, dB, dC, M, N, K, st);
    if (mh_gemm_nt(dA, K, dB, K, dR, N, nullptr, 0, M, N, K, 1.f, 0.f, MH_BF16, 1, nullptr, st) != 0) {
      printf("mh_gemm_nt failed: %s\n", mh_last_error());
      return 1;
    }
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> hc(nc), hr(nc);
    CK(hipMemcpy(hc.data(), dC, nc * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), dR, nc * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, neq = 0;
    double worst = 0;
    for (size_t i = 0; i < nc; ++i) {
      const float a = bf2f(hc[i]), b = bf2f(hr[i]);
      if (hc[i] != hr[i]) ++neq;
      const double d = fabs((double)a - b), tol = 1.0 / 128 * std::max(fabs((double)b), 0.05);
      if (!(d <= tol)) ++bad;
      worst = std::max(worst, d);
    }
    // fp64 host product on a sample of elements
    double worst64 = 0;
    for (int sidx = 0; sidx < 512; ++sidx) {
      const size_t m = (size_t)(((uint64_t)sidx * 2654435761ull) % (uint64_t)M), n = (size_t)(((uint64_t)sidx * 40503ull + 17) % (uint64_t)N);
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)bf2f(ha[m * K + k]) * bf2f(hb[n * K + k]);
      worst64 = std::max(worst64, fabs(acc - bf2f(hc[m * N + n])) / std::max(fabs(acc), 0.05));
    }
    // timing: interleaved rounds
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = std::max(3, (int)(2e12 / (2.0 * M * N * K)));
    auto time_it = [&](int which) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) {
        if (which == 0) mh_gemm_nt(dA, K, dB, K, dR, N, nullptr, 0, M, N, K, 1.f, 0.f, MH_BF16, 1, nullptr, st);
        else if (which == 1) launch<1, true>(dA, dB, dC, M, N, K, st);
        else if (which == 2) launch<2, true>(dA, dB, dC, M, N, K, st);
        else launch<1, false>(dA, dB, dC, M, N, K, st);
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return (double)ms / iters;
    };
    double best[4] = {1e30, 1e30, 1e30, 1e30};
    for (int w = 0; w < 4; ++w) time_it(w);  // warm-up
    for (int r = 0; r < 4; ++r)
      for (int w = 0; w < 4; ++w) best[w] = std::min(best[w], time_it(w));
    const double fl = 2.0 * M * N * K;
    printf("[%6d x %5d x %5d] production %7.1f us %7.1f TF | 4-wave sched %7.1f us %7.1f TF | 4-wave reads-early %7.1f us %7.1f TF | "
           "4-wave no stores %7.1f us %7.1f TF || vs production: %zu of %zu differ in the last bf16 bit(s), %zu beyond 2 ulp, max |d| %.4f; "
           "vs fp64 sample: max rel %.5f\n",
           M, N, K, best[0] * 1e3, fl / best[0] / 1e9, best[1] * 1e3, fl / best[1] / 1e9, best[2] * 1e3, fl / best[2] / 1e9, best[3] * 1e3,
           fl / best[3] / 1e9, neq, nc, bad, worst, worst64);
    fflush(stdout);
    CK(hipFree(dA));
    CK(hipFree(dB));
    CK(hipFree(dC));
    CK(hipFree(dR));
  }
  return 0;
}
