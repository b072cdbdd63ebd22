// Helpers shared by the event-level MFMA attention kernels (attention_mfma.hip, attention_mfma3.hip): the 32x32x16
// bf16 MFMA wrapper, the LDS-DMA tile staging, the gradient-row store with the RoPE transpose, the XCD-aware work order.
#pragma once
#include "common.h"

constexpr int HD = 64;
constexpr int TILE64 = 64 * 128;  // bytes
constexpr int DKV_STAGE = 4 * TILE64 + 2048;  // bytes of one stage of the dK/dV kernel (4 tiles + 1 KiB lse + 1 KiB delta)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 4.0f;  // a row's reference max may lag its true max by a factor <= 2^4

__device__ inline f32x16 mfma32(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ inline f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
__device__ inline bf16x8 lds_frag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + lds_tile_off(row, chunk));
}
__device__ inline bf16x8 pack8(const f32x16& v, int base) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)v[base + e];
  return o;
}
__device__ inline float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32
// register r of lane-half hi <-> reduction index (within a 32-block) 16*(r>>3) + 8*hi + (r&7)
__device__ inline int reg_index(int r, int hi) { return 16 * (r >> 3) + 8 * hi + (r & 7); }

// stage a 64-row x 64-col bf16 tile: rows row0.. (clamped to row_clamp), columns col0..col0+63
__device__ inline void stage64(const bf16* __restrict__ base, int64_t ld, int64_t row0, int64_t row_clamp, int64_t col0,
                               char* lds_tile, int wave, int lane) {
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int g8 = wave + 4 * it;
    const int r = g8 * 8 + rsub;
    const int c = pc ^ lds_swz(r);
    int64_t grow = row0 + r;
    if (grow > row_clamp) grow = row_clamp;
    glds16(base + grow * ld + col0 + c * 8, lds_tile + g8 * 1024);
  }
}

// The same with a wave-uniform 64-bit base and a 32-bit per-lane BYTE offset (all operands of one (batch, head) panel lie
// within 2^31 bytes of its base), issued from inline asm in the SGPR-base form `global_load_lds_dwordx4 voff, s[base]`: one
// address VGPR per request instead of a 64-bit pair, no 64-bit VALU adds in the tile loops (hipcc does not select this form
// for the builtin; the first form kept a dozen address registers alive across the loops and re-derived each address with
// v_lshl_add_u64).  M0 (the LDS destination) is saved and restored inside the statement (cdna_hip_programming.md 5.7).
// hipcc does NOT count these loads: every wait on them is an explicit s_waitcnt vmcnt (stage_wait_all) before the barrier.
__device__ inline void glds16_s(const void* __restrict__ sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
__device__ inline void stage_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ inline unsigned lds_u32(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ inline void stage64u(const bf16* __restrict__ base, int ld, int row0, int row_clamp, int col0, char* lds_tile,
                                int wave, int lane) {
  const int rsub = lane >> 3, pc = lane & 7;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_u32(lds_tile));
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int g8 = wave + 4 * it;
    const int r = g8 * 8 + rsub;
    const int c = pc ^ lds_swz(r);
    int grow = row0 + r;
    grow = grow > row_clamp ? row_clamp : grow;
    const unsigned off = (unsigned)(grow * ld + col0 + c * 8) * 2u;
    glds16_s(base, off, lds0 + g8 * 1024);
  }
}

// The transpose reads are issued from inline asm: through the builtin hipcc puts an s_waitcnt vmcnt(0) in front of every
// ds_read_b64_tr_b16 while an LDS-DMA is outstanding (it cannot tell the stages apart), which would drain the stage requested
// at the top of the tile.  The asm reads are invisible to its counters, so they are waited for by hand (lgkmcnt(0) +
// sched_barrier before the first consumer; cdna_hip_programming.md 5.7 form iii).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <int OFF>
__device__ inline u32x2 ds_tr16(unsigned lds_addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "i"(OFF));
  return r;
}
__device__ inline unsigned lds_addr32(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ inline bf16x8 join8(const u32x2& a, const u32x2& b) {
  union {
    bf16x8 v;
    u32x2 h[2];
  } u;
  u.h[0] = a;
  u.h[1] = b;
  return u.v;
}

// Per-lane byte offsets for reading a TRANSPOSED 32x32x16 operand fragment out of a row-major 64-row tile (rows = the
// contraction index, 128-byte swizzled rows) with ds_read_b64_tr_b16: a 16-lane group reads a [4 rows][16 columns] block and
// lane i of the group receives column i (4 rows); supplier p of group g points at row 8 hi + 4 half + (p >> 2), columns
// xb*32 + 16 (g&1) + 8 (p&1) + 4 ((p>>1)&1) .. +3, which hands output lane i column xb*32 + pi32(16 (g&1) + i) -- the row
// permutation the accumulator layout wants.  Two reads (half = 0, 1) make one 8-deep fragment; the 16-row step t of the
// contraction is the immediate t * 2048 (it leaves the swizzle term lds_swz(row) unchanged).
__device__ inline void tr_frag_offsets(int lane, int (&voff)[2][2]) {
  const int p = lane & 15, gb = (lane >> 4) & 1, hi = lane >> 5;
#pragma unroll
  for (int xb = 0; xb < 2; ++xb)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 8 * hi + 4 * half + (p >> 2);
      const int col = xb * 32 + 16 * gb + 8 * (p & 1) + 4 * ((p >> 1) & 1);
      voff[xb][half] = lds_tile_off(row, col >> 3) + (col & 7) * 2;
    }
}

// Store one lane's share of a 64-wide gradient row (acc[hb][16]: elements hb*32 + 16*r8 + 8*hi + e), optionally through
// the transpose of the RoPE rotation at position `pos` (the gradient with respect to the unrotated projection): the
// partners d and d + 32 are acc[0][.] and acc[1][.] of the same lane.  Roundings as the separate pass it replaces
// (mh_rope with dir = -1 on the stored bf16 gradient; cos/sin rounded to bf16, modeling_llama.py:126).
__device__ inline void store_grad_row(bf16* orow, const f32x16 (&acc)[2], float scale, int hi, const float* cos_t,
                                      const float* sin_t, int pos) {
#pragma unroll
  for (int r8 = 0; r8 < 2; ++r8) {
    bf16x8 v0, v1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v0[e] = (bf16)(acc[0][8 * r8 + e] * scale);
      v1[e] = (bf16)(acc[1][8 * r8 + e] * scale);
    }
    if (cos_t != nullptr) {
      const int i0 = 16 * r8 + 8 * hi;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cos_t + pos * 32 + i0), c1 = *reinterpret_cast<const f32x4*>(cos_t + pos * 32 + i0 + 4);
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sin_t + pos * 32 + i0), s1 = *reinterpret_cast<const f32x4*>(sin_t + pos * 32 + i0 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = (float)(bf16)(e < 4 ? c0[e] : c1[e - 4]), sn = -(float)(bf16)(e < 4 ? s0[e] : s1[e - 4]);
        const float x1 = (float)v0[e], x2 = (float)v1[e];
        v0[e] = (bf16)(x1 * c - x2 * sn);
        v1[e] = (bf16)(x2 * c + x1 * sn);
      }
    }
    *reinterpret_cast<bf16x8*>(orow + 16 * r8 + 8 * hi) = v0;
    *reinterpret_cast<bf16x8*>(orow + 32 + 16 * r8 + 8 * hi) = v1;
  }
}

// Work assignment.  The dispatcher places workgroup b on XCD b % 8 (each XCD has a private 4 MiB L2), so a 1-D
// grid is decoded as  xcd = b & 7, i = b >> 3, head = (i / ntile) * 8 + xcd, tile = i % ntile : consecutive
// workgroups of one XCD walk the tiles of ONE (batch, head) pair, whose K/V (or Q/dO) panels then stay in that
// XCD's L2 instead of being re-fetched (r01 PMC with the (tile, head) 2-D grid: L2 hit rate 34 % fwd, 14 % dK/dV).
// r06: PASSES.  A workgroup's work shrinks with its tile rank (rank 0: the whole sequence, the last rank: two key tiles), and with
// one pass the order on an XCD is a sawtooth -- pair after pair from heaviest to lightest -- so the kernel ends with the heavy
// workgroups of the LAST pairs still running beside empty slots: the residency census of the instrumented forward
// (tools/attn_timeline.py, mh_attn_fwd_timeline) counts 2.3-2.5 workgroups per CU on average where 3 fit.  With P passes
// the ranks are cut into P contiguous chunks and pass p walks chunk p of EVERY pair: the light chunks come last, the tail is as
// short as the lightest workgroups, and a pair's K/V panels still serve a whole chunk of its tiles while they sit in L2.
// BH arrives packed with the pass count (BH | P << 24; 0 = 1 pass) and is unpacked here.
__device__ inline bool attn_work(int& BH, int ntile, int& bh, int& tile) {
  int P = BH >> 24;
  BH &= 0xffffff;
  P = P < 1 ? 1 : (P > ntile ? ntile : P);
  const int lin = blockIdx.x, xcd = lin & 7;
  int i = lin >> 3;
  const int nG = (BH + 7) >> 3;  // (batch, head) pairs per XCD
  int lo = 0, n = ntile;
  for (int p = 0; p < P; ++p) {
    lo = p * ntile / P;
    n = (p + 1) * ntile / P - lo;
    if (i < nG * n) break;
    i -= nG * n;
  }
  const int g = i / n;
  bh = g * 8 + xcd;
  tile = lo + (i - g * n);
  return bh < BH;
}
