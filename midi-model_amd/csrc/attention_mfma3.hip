// Event-level causal attention (head_dim 64, bf16), third form of the three MFMA kernels of attention_mfma.hip -- the
// default (same orientation, same LDS tile format, same results to rounding; selected per kernel by
// mh_set_option("attn_v3", bits): bit 0 forward, bit 1 dQ, bit 2 dK/dV; bits 3 / 4: the backward pair / the forward take their
// transposed operands out of the row-major tiles with ds_read_b64_tr_b16 instead of from prepared [B,H,64,Sp] copies; bits 5 / 6
// below, bit 7 = the forward's lazy reference maximum (r06); default 255 = all of them).
//
// What the ISA of the first form showed (r02, `hipcc -S` of attention_mfma.hip) and what changes here:
//  * every MFMA pair sat behind its own `ds_read_b128 ; s_waitcnt lgkmcnt(0)`: 16-32 exposed LDS round trips per tile and
//    wave (PMC: waves 32 % of their cycles in s_waitcnt).  Here the fragments of a phase are requested in ONE batch ahead of
//    the MFMAs that use them, and the batch of the next phase goes out before the arithmetic of the current one
//    (sched_barrier keeps hipcc from sinking the reads back to their uses); the compiler's own counted lgkmcnt waits remain.
//  * the tile loop chose between the masked and the unmasked instantiation INSIDE the loop; the two inlined bodies got
//    different register assignments for the loop-carried accumulators and hipcc reconciled them with 16-64 v_mov_b64 per
//    tile (the dK/dV kernel copied all four accumulators in and out).  A wave's tiles come in a fixed order -- dK/dV: tiles
//    before its keys (nothing to do), the one or two diagonal tiles (masked), the full tiles, a ragged last tile (masked);
//    forward and dQ: full tiles, one diagonal tile, then tiles only its neighbours need -- so the loop is split into one
//    loop per class, each with a single body.
//  * the forward's uniform `if (__any(rescale))` produced a copy of the O accumulator on the common path; the test is now
//    per lane (a row rescales when ITS maximum grew): the update is executed in place under EXEC and skipped when no lane
//    needs it.
//  * the row maximum crosses the two wave halves through v_permlane32_swap (VALU) instead of ds_bpermute, whose wait would
//    also wait for the fragment batch in flight.
//  * half of the forward's time was LDS-DMA issue and the barrier (ablation below): the requests go out in the SGPR-base form
//    (one address VGPR each, no 64-bit VALU adds), and the kernels stage only row-major tiles -- K and V; K and V; Q and dO --
//    reading V^T, K^T, Q^T, dO^T out of them with transpose reads: 4 / 4 / 5 requests per wave and tile instead of 4 / 6 / 9,
//    and no transposed copies in HBM.
//  * delta rides in the matrix pipe: the dP chains start from C = delta and multiply negated dO (dQ) / V (dK/dV) fragments.
//  * mh_attn_bwd_o (bit 5, the host side's default): the dQ kernel computes delta = rowsum(dO * O) from the rows its lanes
//    hold and leaves delta and -lse * log2(e) in the scratch buffer for the dK/dV kernel launched behind it: no delta pass
//    over O and dO, one multiply less per score in dK/dV (32 of its ~210 VALU issue slots per 64-query tile).
//  * the forward keeps three K/V stages in LDS (bit 6; counted vmcnt wait): a tile's requests get two tile times to land.
// Bound: VALU issue slots and per-wave serialisation at head_dim 64, not the matrix pipe (39 % busy) -- DESIGN.md section 4,
// "the SIMD issue model".
#include <stdlib.h>

#include "attn_mfma_common.h"

// (both start from the environment in EVERY host thread -- MH_ATTN_V3, MH_ATTN_V3_WPS, as MH_GEMM does in api.cpp -- so an A/B run
// selected by the environment also reaches launches made from other threads, e.g. autograd's backward thread)
static int attn_env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
thread_local int g_attn_v3 = attn_env_int("MH_ATTN_V3", 255);     // mh_set_option("attn_v3", bits): 1 forward, 2 dQ, 4 dK/dV, 8 transpose reads in dQ + dK/dV (needs 2 | 4),
                        // 16 transpose reads in the forward (needs 1; the caller then passes no V^T copy), 32 the host side
                        // calls mh_attn_bwd_o (delta computed inside the dQ kernel; needs 2 | 4 | 8), 64 three K/V stages
                        // in the forward (needs 1 | 16), 128 lazy reference maximum in the forward (needs 1 | 16; fwd3_tile)
// mh_set_option("attn_passes", P): attn_work's pass count (attn_mfma_common.h).  Same-box A/B at B = 16, H = 16 (tools/attn_fwd_ab.py,
// profiles/r06_attn_passes_ab.txt): forward 196 -> 170 us at S = 2048 and 658 -> 612 us at S = 4096 with P = 5, backward pair 594 -> 534
// and 1939 -> 1879 us; P = 2 / 3 / 8 / 16 lie between (more passes re-fetch K/V panels more often: at S = 4096 they no longer
// fit the Infinity Cache together).  Identical results.
thread_local int g_attn_passes = attn_env_int("MH_ATTN_PASSES", 5);
thread_local int g_attn_v3_wps = attn_env_int("MH_ATTN_V3_WPS", 0);
// row scale of the NEXT backward launched by this thread (mh_attn_bwd_o_scaled sets it around its call; NULL otherwise)
thread_local const float* g_attn_bwd_rowscale = nullptr;  // mh_set_option("attn_v3_wps", n): register budget (waves per SIMD) override for A/B runs, 0 = default

__device__ inline bf16x8 ldsv(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
// eight consecutive accumulator registers -> one bf16 operand fragment, as four explicit two-element conversions (each one
// v_cvt_pk_bf16_f32): from eight single casts hipcc assembled the dS fragments (products of v_pk_mul_f32) out of single
// conversions, 6 v_mov + 3 v_alignbit per fragment.  (Spelling the instruction in inline asm instead is wrong: hipcc does not
// pad the VALU-write -> MFMA-operand hazard behind an asm statement, cdna_hip_programming.md 5.7 -- NaNs on the device.)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ inline bf16x8 pack8_pk(const f32x16& v, int base) {
  union {
    bf16x8 v;
    bf16x2 h[4];
  } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.h[i] = __builtin_convertvector(f32x2{v[base + 2 * i], v[base + 2 * i + 1]}, bf16x2);
  return r.v;
}
__device__ inline float xhalf_max(float v) {  // max with the other wave half's value (lane ^ 32), VALU only
  const int iv = __float_as_int(v);
  const auto pr = __builtin_amdgcn_permlane32_swap(iv, iv, false, false);
  return fmaxf(__int_as_float(pr[0]), __int_as_float(pr[1]));
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// one 64-key tile for one wave (32 query rows).  foff[s]: this lane's byte offset of fragment s inside a 32-row tile block
// (row pi32(lane & 31), chunk 2s + hi); the second block of a tile is 4096 bytes further.  qrel = query row - first key.
// (Where the time goes, measured by leaving parts out of this kernel -- profiles/r02_run16_attn_forms_ab_and_fwd_ablation.txt,
// B=16, S=4096, 684 us complete: no v_exp -40, no row maximum -8, no P V MFMAs -100, no Q K^T MFMAs -68, no MFMAs -151, no
// fragment reads -55, neither -278; staging + barriers + the sums alone 333 us.)
// TR: tV is the row-major V tile [key][d] and the V^T fragments are transpose reads (no prepared [B,H,64,Sp] copy of V).
// LZ (r06, bit 7 of attn_v3): the reference maximum is LAZY.  The probabilities are taken against the row's reference maximum as
// it stands -- no row maximum, no half-wave exchange, no compare: 16 v_max3 + ~6 of the ~145 VALU of a tile -- and the tile's
// partial row sum, which is computed anyway, is the overflow detector: max p <= sum p, so a sum below LZ_BOUND = 2^40 proves that
// every probability of the lane is (the reference may lag the true maximum by 40 binades instead of RESCALE_THR = 4: floating
// point keeps the relative precision of P, O and l; only overflow to inf has to be excluded).  A sum that is larger, infinite
// or NaN (first tile: m = -inf) in ANY lane sends the wave through the classic path -- maximum, rescale under EXEC, the
// probabilities again from the scores, which the fast path leaves intact (P goes straight into its packed bf16 fragments).
constexpr float LZ_BOUND = 1099511627776.f;  // 2^40
// TL (A/B library only): s_memtime at the seams of the tile's segments, ts[2..6] (mh_attn_fwd_timeline; the values are read at the
// end of the tile, behind one s_waitcnt lgkmcnt(0) at a point where no LDS operation is outstanding -- an SMEM result in
// flight only makes hipcc's counted LDS waits conservative)
__device__ inline uint64_t tl_now() {
  uint64_t t;
  asm volatile("s_memtime %0" : "=s"(t));
  return t;
}
template <bool MASK, bool TR, bool LZ = false, bool TL = false>
__device__ inline void fwd3_tile(const char* tK, const char* tV, const int (&trof)[2][2], const int (&foff)[4],
                                 const bf16x8 (&qf)[4], f32x16 (&oacc)[2], float& m, float& l, int hi, int qrel, float sc,
                                 uint64_t* ts = nullptr) {
  bf16x8 kf[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) kf[kb][s] = ldsv(tK + foff[s] + kb * 4096);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TL) ts[2] = tl_now();  // K fragment reads issued
  f32x16 sacc[2] = {zero16(), zero16()};
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) sacc[kb] = mfma32(kf[kb][s], qf[s], sacc[kb]);  // two independent accumulator chains
  bf16x8 vf[4][2];    // the V^T fragments: in flight under the softmax arithmetic
  u32x2 vr[4][2][2];  // the same as transpose reads: [t][db][half]
  if (TR) {
    unsigned tv[2][2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int half = 0; half < 2; ++half) tv[db][half] = lds_addr32(tV) + (unsigned)trof[db][half];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        vr[0][db][half] = ds_tr16<0>(tv[db][half]);
        vr[1][db][half] = ds_tr16<2048>(tv[db][half]);
        vr[2][db][half] = ds_tr16<4096>(tv[db][half]);
        vr[3][db][half] = ds_tr16<6144>(tv[db][half]);
      }
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int db = 0; db < 2; ++db) vf[t][db] = ldsv(tV + foff[t] + db * 4096);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (TL) {
    ts[3] = tl_now();  // S MFMAs and V^T reads issued
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (LZ) {
    if (MASK) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kb * 32 + 16 * (r >> 3) + (r & 7) > qrel - 8 * hi) sacc[kb][r] = -INFINITY;  // (see below)
    }
    bf16x8 pf[4];
    float ps;
    auto probabilities = [&]() {  // P against the current m -> packed fragments + this lane's partial row sum; sacc untouched
      float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        union {
          bf16x8 v;
          bf16x2 h[4];
        } u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p0 = fast_exp2(__builtin_fmaf(sacc[t >> 1][8 * (t & 1) + 2 * i], sc, -m));
          const float p1 = fast_exp2(__builtin_fmaf(sacc[t >> 1][8 * (t & 1) + 2 * i + 1], sc, -m));
          ps0 += p0;
          ps1 += p1;
          u.h[i] = __builtin_convertvector(f32x2{p0, p1}, bf16x2);
        }
        pf[t] = u.v;
      }
      ps = ps0 + ps1;
    };
    probabilities();
    if (__builtin_expect(__any(!(ps <= LZ_BOUND)), 0)) {  // (wave-uniform; also the first tile of every row: m = -inf)
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
      mx = xhalf_max(mx) * sc;
      if (mx > m + RESCALE_THR) {
        const float mn = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mn);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
        l *= alpha;
        m = mn;
      }
      probabilities();
    }
    l += ps;
    if constexpr (TL) {
      __builtin_amdgcn_sched_barrier(0);
      ts[4] = tl_now();  // softmax arithmetic issued
    }
    if (TR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transpose reads (asm: invisible to hipcc's counters)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (TL) {
      ts[5] = tl_now();  // V^T fragments landed
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int db = 0; db < 2; ++db) oacc[db] = mfma32(TR ? join8(vr[t][db][0], vr[t][db][1]) : vf[t][db], pf[t], oacc[db]);
    if constexpr (TL) {
      __builtin_amdgcn_sched_barrier(0);
      ts[6] = tl_now();  // P V MFMAs issued
    }
    return;
  }
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MASK) {
        // (register r <-> key kb*32 + 16 (r>>3) + (r&7) + 8 hi: a compile-time constant against ONE lane value, qrel - 8 hi;
        //  spelled with reg_index(r, hi) hipcc kept 32 per-lane index registers alive for the masked tile)
        if (kb * 32 + 16 * (r >> 3) + (r & 7) > qrel - 8 * hi) sacc[kb][r] = -INFINITY;
      }
      mx = fmaxf(mx, sacc[kb][r]);
    }
  mx = xhalf_max(mx) * sc;  // running max kept in scaled (log2) units
  // A row moves its reference maximum only when its true maximum grew by more than RESCALE_THR (log2 units): until then
  // its probabilities may reach 2^THR instead of 1, and l / O / lse stay mutually consistent (exact maths; only the bf16
  // rounding of P sees the larger magnitudes).  Per lane, in place under EXEC (both halves of a row decide alike).
  if (mx > m + RESCALE_THR) {
    const float mn = fmaxf(m, mx);
    const float alpha = fast_exp2(m - mn);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    l *= alpha;
    m = mn;
  }
  float ps0 = 0.f, ps1 = 0.f;  // (two partial sums: half the length of the dependent add chain)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float p0 = fast_exp2(__builtin_fmaf(sacc[0][r], sc, -m)), p1 = fast_exp2(__builtin_fmaf(sacc[1][r], sc, -m));
    sacc[0][r] = p0;
    sacc[1][r] = p1;
    ps0 += p0;
    ps1 += p1;
  }
  l += ps0 + ps1;
  bf16x8 pf[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) pf[t] = pack8_pk(sacc[t >> 1], 8 * (t & 1));
  if constexpr (TL) {
    __builtin_amdgcn_sched_barrier(0);
    ts[4] = tl_now();  // softmax arithmetic issued
  }
  if (TR) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transpose reads (asm: invisible to hipcc's counters)
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (TL) {
    ts[5] = tl_now();  // V^T fragments landed
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int db = 0; db < 2; ++db) oacc[db] = mfma32(TR ? join8(vr[t][db][0], vr[t][db][1]) : vf[t][db], pf[t], oacc[db]);
  if constexpr (TL) {
    __builtin_amdgcn_sched_barrier(0);
    ts[6] = tl_now();  // P V MFMAs issued
  }
}

constexpr int TL_TILES = 32, TL_FIRST = 8, TL_STAMPS = 9;  // timeline build: tiles TL_FIRST .. +31 of a workgroup's loop, 9 stamps each
constexpr int TL_BYTES = 4 * TL_TILES * TL_STAMPS * 4;       // 4 waves x 32-bit stamps = 4608 bytes behind the stages (still 3 workgroups per CU)
template <int WPS, bool TR, int NS = 2 /* K/V stages in LDS: tile kt + NS - 1 is requested while tile kt is computed */, bool LZ = false,
          bool TL = false /* timeline build: `vt` is the uint32 output [16 workgroups][4 waves][TL_TILES][TL_STAMPS] */>
__global__ __launch_bounds__(256, WPS) void attn_fwd3_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ vt,
                                                          bf16* __restrict__ o, float* __restrict__ lse, int S, int Sp, int H,
                                                          float sc /* scale*log2(e) */, int BH, int nqt,
                                                          int nqt_all /* 128-row query tiles of the sequence; the launch covers the
                                                                         LAST nqt of them (mh_attn_fwd_tail: a chunk behind cached rows) */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 tiles = [stage][K | V^T] (dynamic: see attn_fwd_kernel)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh_, tile_;
  if (!attn_work(BH, nqt, bh_, tile_)) return;
  const int64_t bh = bh_;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int q0 = (nqt_all - 1 - tile_) * 128;  // heavy (late) query tiles first
  const int qw0 = q0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int qrow = qw0 + li;
  const int qld = (qrow < S) ? qrow : S - 1;

  const bf16* kbase = qkv + b * S * D3 + D + (int64_t)h * HD;
  const bf16* vtbase = (TR || TL) ? kbase + D : vt + bh * HD * Sp;  // (TR: V row-major, straight out of the fused qkv rows)
  uint64_t tl_t0 = 0;
  int tl_first = TL_FIRST;  // first recorded tile: the word behind the census, written by the host (default 8)
  if constexpr (TL) {
    tl_t0 = tl_now();  // (census: when this workgroup started, see the end of the kernel)
    tl_first = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(vt)[16 * (TL_BYTES / 4) + (size_t)gridDim.x * 8]);
  }
  uint64_t ts[TL_STAMPS];
  uint32_t* tl_lds = reinterpret_cast<uint32_t*>(smem + NS * 2 * TILE64) + wave * TL_TILES * TL_STAMPS;
  auto tl_flush = [&](int kt) {  // (TL) the tile's stamps -> LDS, low words
    if constexpr (TL) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts[0]), "+s"(ts[1]), "+s"(ts[2]), "+s"(ts[3]), "+s"(ts[4]), "+s"(ts[5]), "+s"(ts[6]), "+s"(ts[7]), "+s"(ts[8])::"memory");
      if (kt >= tl_first && kt < tl_first + TL_TILES && lane == 0) {
#pragma unroll
        for (int i = 0; i < TL_STAMPS; ++i) tl_lds[(kt - tl_first) * TL_STAMPS + i] = (uint32_t)ts[i];
      }
    }
  };

  bf16x8 qf[4];
  {
    const bf16* qp = qkv + (b * S + qld) * D3 + (int64_t)h * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  f32x16 oacc[2] = {zero16(), zero16()};
  float m = -INFINITY, l = 0.f;

  int last_q = q0 + 127;
  if (last_q > S - 1) last_q = S - 1;
  const int kt_last = last_q / 64;
  // this wave's tiles, in order: n_full tiles entirely below its first row, ONE tile crossing its diagonal (64 n_full <=
  // qw0 < 64 n_full + 64), then the tiles only the later waves of the workgroup still need
  int n_full = qw0 >> 6;
  if (n_full > kt_last + 1) n_full = kt_last + 1;  // (a wave whose rows all lie past S)
  int foff[4];
  {
    const int pli = pi32(li);
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = lds_tile_off(pli, 2 * s + hi);
  }

  const int iD3 = (int)D3;
  int trof[2][2];
  tr_frag_offsets(lane, trof);
  auto stage_tile = [&](int t, int buf) {  // 4 LDS-DMA requests per wave
    char* dst = smem + buf * 2 * TILE64;
    stage64u(kbase, iD3, t * 64, S - 1, 0, dst, wave, lane);
    if (TR) stage64u(vtbase, iD3, t * 64, S - 1, 0, dst + TILE64, wave, lane);
    else stage64u(vtbase, Sp, 0, HD - 1, t * 64, dst + TILE64, wave, lane);
  };
  // Three stages: a tile's requests get two tile times to land instead of one (with two stages an iteration cannot be
  // shorter than one memory round trip, which only the other resident workgroups cover).  wait_next(kt): tile kt + 1 has
  // landed = everything but the youngest stage's four requests, when one was issued behind it.
  auto wait_next = [&](int kt) {
    if (NS == 3 && kt + 2 <= kt_last) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else stage_wait_all();
  };
  stage_tile(0, 0);
  if (NS == 3 && 1 <= kt_last) stage_tile(1, 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));  // (see attn_bwd_dkv_kernel)
  wait_next(-1);
  __syncthreads();
  int cb = 0, nb = NS - 1;  // buffer of tile kt, buffer tile kt + NS - 1 goes to
  auto stage_next = [&](int kt) {
    if (kt + NS - 1 <= kt_last) stage_tile(kt + NS - 1, nb);
    nb = (nb + 1 == NS) ? 0 : nb + 1;
  };
  auto advance = [&](int kt) {
    wait_next(kt);
    __syncthreads();
    cb = (cb + 1 == NS) ? 0 : cb + 1;
  };
  int kt = 0;
  if constexpr (TL) {
#pragma unroll
    for (int i = 0; i < TL_STAMPS; ++i) ts[i] = 0;
  }
  for (; kt < n_full; ++kt) {
    if constexpr (TL) ts[0] = tl_now();  // top of the tile
    stage_next(kt);
    if constexpr (TL) ts[1] = tl_now();  // LDS-DMA requests issued
    const char* cur = smem + cb * 2 * TILE64;
    fwd3_tile<false, TR, LZ, TL>(cur, cur + TILE64, trof, foff, qf, oacc, m, l, hi, 0, sc, ts);
    if constexpr (TL) {
      wait_next(kt);
      ts[7] = tl_now();  // the next tile's requests have landed (this wave's)
      __syncthreads();
      ts[8] = tl_now();  // barrier passed
      cb = (cb + 1 == NS) ? 0 : cb + 1;
      tl_flush(kt);
    } else {
      advance(kt);
    }
  }
  if (kt <= kt_last) {
    stage_next(kt);
    const char* cur = smem + cb * 2 * TILE64;
    fwd3_tile<true, TR, LZ>(cur, cur + TILE64, trof, foff, qf, oacc, m, l, hi, qrow - kt * 64, sc);
    advance(kt);
    ++kt;
  }
  for (; kt <= kt_last; ++kt) {
    stage_next(kt);
    advance(kt);
  }
  if constexpr (TL) {  // workgroups 0, 8, .. 120 (the first sixteen of XCD 0: the heaviest query tiles of head 0) hand their stamps out
    __syncthreads();
    if ((blockIdx.x & 7) == 0 && (blockIdx.x >> 3) < 16) {
      uint32_t* out = reinterpret_cast<uint32_t*>(const_cast<bf16*>(vt)) + (blockIdx.x >> 3) * (TL_BYTES / 4);
      const uint32_t* src = reinterpret_cast<const uint32_t*>(smem + NS * 2 * TILE64);
      for (int i = tid; i < TL_BYTES / 4; i += 256) out[i] = src[i];
    }
    // census of EVERY workgroup behind the sixteen stamp arrays: [start lo, hi, end lo, hi, HW_ID, XCC_ID, key tiles, 0] -- which CU
    // it ran on and when, i.e. how many workgroups a CU really holds at a time (tools/attn_timeline.py)
    if (tid == 0) {
      const uint64_t t1 = tl_now();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tl_t0) : "s"(t1) : "memory");
      uint32_t* c = reinterpret_cast<uint32_t*>(const_cast<bf16*>(vt)) + 16 * (TL_BYTES / 4) + (size_t)blockIdx.x * 8;
      c[0] = (uint32_t)tl_t0;
      c[1] = (uint32_t)(tl_t0 >> 32);
      c[2] = (uint32_t)t1;
      c[3] = (uint32_t)(t1 >> 32);
      c[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
      c[5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
      c[6] = (uint32_t)(kt_last + 1);
      c[7] = 0;
    }
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  if (qrow < S) {
    const float inv = 1.f / lt;
    bf16* orow = o + (b * S + qrow) * D + (int64_t)h * HD;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r8 = 0; r8 < 2; ++r8) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)(oacc[db][8 * r8 + e] * inv);
        *reinterpret_cast<bf16x8*>(orow + db * 32 + 16 * r8 + 8 * hi) = v;
      }
    if (hi == 0) lse[bh * Sp + qrow] = (m + log2f(lt)) * 0.6931471805599453f;
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, dQ: block = 128 query rows, loop over key tiles
// ---------------------------------------------------------------------------------------------------
// one 32-key block (KB) of a 64-key tile for one wave (32 query rows).  TR: the K^T fragments of dQ^T += K^T dS^T are read
// out of the row-major K tile with ds_read_b64_tr_b16 (tkt = the LDS addresses tK + tr_frag_offsets) -- no transposed
// [B,H,64,Sp] copy of K in HBM, a third fewer LDS-DMA requests per tile (each costs its wave 60-180 issue cycles:
// profiles/r02_run16_attn_forms_ab_and_fwd_ablation.txt, staging + barriers alone are half of the forward).  Otherwise from
// the prepared K^T tile (tKT).
template <bool MASK, bool TR, int KB>
__device__ inline void dq3_half(const char* tK, const char* tV, const char* tKT, const unsigned (&tkt)[2][2],
                                const int (&foff)[4], const bf16x8 (&qf)[4], const bf16x8 (&dof)[4] /* -dO */,
                                f32x16 (&dqacc)[2], int hi, int qrel, float sc, float lse2,
                                const f32x16& dlt /* the row's delta in all 16 registers */) {
  // delta goes into the matrix pipe: the dP chain starts from C = delta and multiplies the NEGATED dO fragments, so its
  // result is delta - dP and dS comes out negated with one multiply per element (the subtraction was 32 VALU issue slots
  // per tile: tools/mfma_valu_probe.hip -- a SIMD issues one VALU per ~4 cycles whatever the number of waves); the kernel
  // stores -(-dQ).
  bf16x8 kf[4], vf[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf[s] = ldsv(tK + foff[s] + KB * 4096);
    vf[s] = ldsv(tV + foff[s] + KB * 4096);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 sacc = zero16(), pacc;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    sacc = mfma32(kf[s], qf[s], sacc);
    pacc = mfma32(vf[s], dof[s], s == 0 ? dlt : pacc);
  }
  bf16x8 ktf[2][2];   // [t][hb]: keys KB*32 + 16 t + 8 hi .. +7 of d-row block hb
  u32x2 kr[2][2][2];  // the same as transpose reads: [t][hb][half]
  if (TR) {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        kr[0][hb][half] = ds_tr16<(2 * KB) * 2048>(tkt[hb][half]);
        kr[1][hb][half] = ds_tr16<(2 * KB + 1) * 2048>(tkt[hb][half]);
      }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) ktf[t][hb] = ldsv(tKT + foff[2 * KB + t] + hb * 4096);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {  // S -> -dS (unscaled), in place
    float p = fast_exp2(__builtin_fmaf(sacc[r], sc, -lse2));
    if (MASK) {
      if (KB * 32 + 16 * (r >> 3) + (r & 7) > qrel - 8 * hi) p = 0.f;  // (see fwd3_tile)
    }
    sacc[r] = p * pacc[r];
  }
  const bf16x8 dsf0 = pack8_pk(sacc, 0), dsf1 = pack8_pk(sacc, 8);
  if (TR) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the transpose reads (asm: invisible to hipcc's counters)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) dqacc[hb] = mfma32(join8(kr[0][hb][0], kr[0][hb][1]), dsf0, dqacc[hb]);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) dqacc[hb] = mfma32(join8(kr[1][hb][0], kr[1][hb][1]), dsf1, dqacc[hb]);
  } else {
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) dqacc[hb] = mfma32(ktf[0][hb], dsf0, dqacc[hb]);
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) dqacc[hb] = mfma32(ktf[1][hb], dsf1, dqacc[hb]);
  }
}

template <bool MASK, bool TR>
__device__ inline void dq3_tile(const char* tK, const char* tV, const char* tKT, const int (&trof)[2][2], const int (&foff)[4],
                                const bf16x8 (&qf)[4], const bf16x8 (&dof)[4], f32x16 (&dqacc)[2], int hi, int qrel, float sc,
                                float lse2, const f32x16& dlt) {
  unsigned tkt[2][2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int half = 0; half < 2; ++half) tkt[hb][half] = lds_addr32(tK) + (unsigned)trof[hb][half];
  dq3_half<MASK, TR, 0>(tK, tV, tKT, tkt, foff, qf, dof, dqacc, hi, qrel, sc, lse2, dlt);
  dq3_half<MASK, TR, 1>(tK, tV, tKT, tkt, foff, qf, dof, dqacc, hi, qrel, sc, lse2, dlt);
}

template <int WPS, bool TR>
__global__ __launch_bounds__(256, WPS) void attn_bwd_dq3_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              const bf16* __restrict__ kt_, bf16* __restrict__ dqkv, int S,
                                                              int Sp, int H, float scale, int BH, int nqt,
                                                              const float* __restrict__ cos_t,
                                                              const float* __restrict__ sin_t,
                                                              const bf16* __restrict__ o_ /* != NULL: delta is computed here */,
                                                              float* __restrict__ delta_w,
                                                              const float* __restrict__ rowscale /* != NULL: row m of dqkv times rowscale[m] */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [stage][K | V | K^T], or [stage][K | V] with transpose reads
  constexpr int NT = TR ? 2 : 3;  // tiles per stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh_, tile_;
  if (!attn_work(BH, nqt, bh_, tile_)) return;
  const int64_t bh = bh_;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int q0 = (nqt - 1 - tile_) * 128;
  const int qw0 = q0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int qrow = qw0 + li;
  const int qld = (qrow < S) ? qrow : S - 1;
  const float sc = scale * LOG2E;

  const bf16* kbase = qkv + b * S * D3 + D + (int64_t)h * HD;
  const bf16* vbase = kbase + D;
  const bf16* ktbase = TR ? nullptr : kt_ + bh * HD * Sp;

  bf16x8 qf[4], dof[4];
  float dl = 0.f;
  {
    const bf16* qp = qkv + (b * S + qld) * D3 + (int64_t)h * HD + 8 * hi;
    const bf16* dp = dout + (b * S + qld) * D + (int64_t)h * HD + 8 * hi;
    bf16x8 of[4];
    if (o_ != nullptr) {
      const bf16* op = o_ + (b * S + qld) * D + (int64_t)h * HD + 8 * hi;
#pragma unroll
      for (int s = 0; s < 4; ++s) of[s] = *reinterpret_cast<const bf16x8*>(op + 16 * s);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
#pragma unroll
      for (int e = 0; e < 8; ++e) dof[s][e] = -d[e];  // (see dq3_tile)
      if (o_ != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) dl = __builtin_fmaf((float)d[e], (float)of[s][e], dl);
      }
    }
  }
  float lse2 = lse[bh * Sp + qld] * LOG2E;
  if (o_ != nullptr) {
    // delta = rowsum(dO * O) of this lane's query row: the two half-waves hold 32 of the 64 columns each; the dK/dV kernel
    // (launched behind this one) reads it from delta_w -- no pass of its own over O and dO
    dl += __shfl_xor(dl, 32, 64);
    if (hi == 0 && qrow < S) {
      delta_w[bh * Sp + qrow] = dl;
      delta_w[(int64_t)BH * Sp + bh * Sp + qrow] = -lse2;  // second half of the scratch: -lse * log2(e) for the dK/dV kernel
    }
  } else {
    dl = delta[bh * Sp + qld];
  }
  f32x16 dlt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dlt[r] = dl;
  f32x16 dqacc[2] = {zero16(), zero16()};

  int last_q = q0 + 127;
  if (last_q > S - 1) last_q = S - 1;
  const int kt_last = last_q / 64;
  int n_full = qw0 >> 6;  // (tile classes of a wave: see attn_fwd3_kernel)
  if (n_full > kt_last + 1) n_full = kt_last + 1;
  int foff[4];
  {
    const int pli = pi32(li);
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = lds_tile_off(pli, 2 * s + hi);
  }

  const int iD3 = (int)D3;
  int trof[2][2];
  tr_frag_offsets(lane, trof);
  stage64u(kbase, iD3, 0, S - 1, 0, smem, wave, lane);
  stage64u(vbase, iD3, 0, S - 1, 0, smem + TILE64, wave, lane);
  if (!TR) stage64u(ktbase, Sp, 0, HD - 1, 0, smem + 2 * TILE64, wave, lane);
#pragma unroll
  for (int s = 0; s < 4; ++s) {  // (see attn_bwd_dkv_kernel)
    asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(dof[s]));
  }
  asm volatile("" : "+v"(lse2));
  asm volatile("" : "+v"(dlt));
  stage_wait_all();
    __syncthreads();
  auto stage_next = [&](int kt) {
    if (kt + 1 <= kt_last) {
      char* nxt = smem + ((kt + 1) & 1) * NT * TILE64;
      stage64u(kbase, iD3, (kt + 1) * 64, S - 1, 0, nxt, wave, lane);
      stage64u(vbase, iD3, (kt + 1) * 64, S - 1, 0, nxt + TILE64, wave, lane);
      if (!TR) stage64u(ktbase, Sp, 0, HD - 1, (kt + 1) * 64, nxt + 2 * TILE64, wave, lane);
    }
  };
  int kt = 0;
  for (; kt < n_full; ++kt) {
    stage_next(kt);
    const char* cur = smem + (kt & 1) * NT * TILE64;
    dq3_tile<false, TR>(cur, cur + TILE64, cur + 2 * TILE64, trof, foff, qf, dof, dqacc, hi, 0, sc, lse2, dlt);
    stage_wait_all();
    __syncthreads();
  }
  if (kt <= kt_last) {
    stage_next(kt);
    const char* cur = smem + (kt & 1) * NT * TILE64;
    dq3_tile<true, TR>(cur, cur + TILE64, cur + 2 * TILE64, trof, foff, qf, dof, dqacc, hi, qrow - kt * 64, sc, lse2, dlt);
    stage_wait_all();
    __syncthreads();
    ++kt;
  }
  for (; kt <= kt_last; ++kt) {
    stage_next(kt);
    stage_wait_all();
    __syncthreads();
  }
  if (qrow < S) {
    bf16* orow = dqkv + (b * S + qrow) * D3 + (int64_t)h * HD;
    // (rowscale, r06: the q|k|v projection sits behind a FOLDED RMSNorm -- the stored gradient is the one of the unscaled
    //  product x W'^T, rstd (.) d qkv, which both the folded dgrad and the folded weight gradient take)
    const float rsc = rowscale != nullptr ? rowscale[b * S + qrow] : 1.f;
    store_grad_row(orow, dqacc, -scale * rsc, hi, cos_t, sin_t, qrow);  // (the accumulator holds -dQ)
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, dK/dV: block = 128 key rows, loop over the query tiles that see them
// ---------------------------------------------------------------------------------------------------
// one 32-query block (QB) of a 64-query tile for one wave (32 key rows).  krel = key row - first query of the tile; qlim =
// number of valid queries in the tile (S - q0, may exceed 64).  TR: the dO^T / Q^T fragments of dV^T += dO^T P and
// dK^T += Q^T dS come out of the row-major dO / Q tiles through ds_read_b64_tr_b16 (addresses tdo / tq = tile +
// tr_frag_offsets): no transposed copies of Q and dO in HBM, 5 instead of 9 LDS-DMA requests per wave and tile.
template <bool MASK, bool TR, int QB, bool PS>
__device__ inline void dkv3_half(const char* tQ, const char* tDO, const char* tQT, const char* tDOT, const char* tLD,
                                 const unsigned (&tq)[2][2], const unsigned (&tdo)[2][2], const int (&foff)[4],
                                 const bf16x8 (&kf)[4], const bf16x8 (&vf)[4] /* -V */, f32x16 (&dkacc)[2], f32x16 (&dvacc)[2],
                                 int hi, int krel, int qlim, float sc) {
  bf16x8 qf[4], dof[4];  // Q / dO fragments of the query block
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    qf[s] = ldsv(tQ + foff[s] + QB * 4096);
    dof[s] = ldsv(tDO + foff[s] + QB * 4096);
  }
  __builtin_amdgcn_sched_barrier(0);
  // delta goes into the matrix pipe (see dq3_half): the dP chain starts from C = the block's 16 delta values as they lie
  // in the stage (register r <-> query QB*32 + 16 (r>>3) + 8 hi + (r&7): four 16-byte reads) and multiplies the NEGATED V
  // fragments, so it delivers delta - dP; dS and dK come out negated (the kernel stores -(-dK)), dV is untouched.
  f32x16 pacc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(tLD + 1024 + (QB * 32 + 16 * (i >> 1) + 8 * hi + 4 * (i & 1)) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) pacc[4 * i + e] = d4[e];
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x16 sacc = zero16();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    sacc = mfma32(qf[s], kf[s], sacc);
    pacc = mfma32(dof[s], vf[s], pacc);
  }
  // The block's second half runs in two steps of 16 queries (t): the dO^T / Q^T fragments (queries QB*32 + 16 t + 8 hi .. +7
  // of row block xb) and the 8 lse values of step t are requested one step ahead -- step 0's under the MFMAs above, step
  // 1's under the four MFMAs of step 0 -- so that 16 + 8 registers hold them instead of 64.
  bf16x8 dotf[2][2], qtf[2][2];            // [t][xb]
  u32x2 dor[2][2][2], qr[2][2][2];         // the same as transpose reads: [t][xb][half]
  f32x4 la[2][2];                          // [t][v4]: queries QB*32 + 16 t + 8 hi + 4 v4 .. +3
#define MH_DKV_REQUEST(T)                                                                      \
  {                                                                                            \
    if (TR) {                                                                                  \
      _Pragma("unroll") for (int xb = 0; xb < 2; ++xb) _Pragma("unroll") for (int half = 0; half < 2; ++half) { \
        dor[T][xb][half] = ds_tr16<(2 * QB + (T)) * 2048>(tdo[xb][half]);                      \
        qr[T][xb][half] = ds_tr16<(2 * QB + (T)) * 2048>(tq[xb][half]);                        \
      }                                                                                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int xb = 0; xb < 2; ++xb) {                                       \
        dotf[T][xb] = ldsv(tDOT + foff[2 * QB + (T)] + xb * 4096);                             \
        qtf[T][xb] = ldsv(tQT + foff[2 * QB + (T)] + xb * 4096);                               \
      }                                                                                        \
    }                                                                                          \
    _Pragma("unroll") for (int v4 = 0; v4 < 2; ++v4)                                           \
        la[T][v4] = *reinterpret_cast<const f32x4*>(tLD + (QB * 32 + 16 * (T) + 8 * hi + 4 * v4) * 4); \
  }
  MH_DKV_REQUEST(0)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int v4 = 0; v4 < 2; ++v4)
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // S -> P, delta - dP -> -dS (unscaled), in place
        const int r = 8 * t + 4 * v4 + e;
        // (PS: the staged statistic already is -lse * log2(e), written by the dQ kernel: one multiply less per element)
        float p = fast_exp2(__builtin_fmaf(sacc[r], sc, PS ? la[t][v4][e] : -la[t][v4][e] * LOG2E));
        float ds = p * pacc[r];
        if (MASK) {
          const int q = QB * 32 + 16 * t + 4 * v4 + e;  // (+ 8 hi, moved to the other side: see fwd3_tile)
          const bool ok = (q >= krel - 8 * hi) && (q < qlim - 8 * hi);
          p = ok ? p : 0.f;
          ds = ok ? ds : 0.f;
        }
        sacc[r] = p;
        pacc[r] = ds;
      }
    const bf16x8 pf = pack8_pk(sacc, 8 * t), dsf = pack8_pk(pacc, 8 * t);
    if (t == 0) {
      MH_DKV_REQUEST(1)
      // (step 0's transpose reads were issued ahead of its lse reads, which the arithmetic above has waited for -- LDS
      // returns in order; the counted wait only states it: step 1's 8 + 2 requests may stay in flight)
      if (TR) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
    } else if (TR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int xb = 0; xb < 2; ++xb) {
      dvacc[xb] = mfma32(TR ? join8(dor[t][xb][0], dor[t][xb][1]) : dotf[t][xb], pf, dvacc[xb]);
      dkacc[xb] = mfma32(TR ? join8(qr[t][xb][0], qr[t][xb][1]) : qtf[t][xb], dsf, dkacc[xb]);
    }
  }
#undef MH_DKV_REQUEST
}

template <bool MASK, bool TR, bool PS>
__device__ inline void dkv3_tile(const char* tQ, const char* tDO, const char* tQT, const char* tDOT, const char* tLD,
                                 const int (&trof)[2][2], const int (&foff)[4], const bf16x8 (&kf)[4], const bf16x8 (&vf)[4],
                                 f32x16 (&dkacc)[2], f32x16 (&dvacc)[2], int hi, int krel, int qlim, float sc) {
  unsigned tq[2][2], tdo[2][2];
#pragma unroll
  for (int xb = 0; xb < 2; ++xb)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      tq[xb][half] = lds_addr32(tQ) + (unsigned)trof[xb][half];
      tdo[xb][half] = lds_addr32(tDO) + (unsigned)trof[xb][half];
    }
  dkv3_half<MASK, TR, 0, PS>(tQ, tDO, tQT, tDOT, tLD, tq, tdo, foff, kf, vf, dkacc, dvacc, hi, krel, qlim, sc);
  dkv3_half<MASK, TR, 1, PS>(tQ, tDO, tQT, tDOT, tLD, tq, tdo, foff, kf, vf, dkacc, dvacc, hi, krel, qlim, sc);
}

template <bool TR, bool PS = false /* lse holds -lse * log2(e) (mh_attn_bwd_o: written by the dQ kernel) */>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv3_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               const bf16* __restrict__ qt_, const bf16* __restrict__ dot_,
                                                               bf16* __restrict__ dqkv, int S, int Sp, int H, float scale,
                                                               int BH, int nkt, const float* __restrict__ cos_t,
                                                               const float* __restrict__ sin_t,
                                                               const float* __restrict__ rowscale /* see attn_bwd_dq3_kernel */) {
  // one stage: [Q | dO | Q^T | dO^T | lse (256 B of a KiB) | delta (256 B of a KiB)], with transpose reads [Q | dO | lse | delta]
  constexpr int NT = TR ? 2 : 4, STG = NT * TILE64 + 2048;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh_, tile_;
  if (!attn_work(BH, nkt, bh_, tile_)) return;
  const int64_t bh = bh_;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int k0 = tile_ * 128;
  const int kw0 = k0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int krow = kw0 + li;
  const int kld = (krow < S) ? krow : S - 1;
  const float sc = scale * LOG2E;

  const bf16* qbase = qkv + b * S * D3 + (int64_t)h * HD;
  const bf16* dobase = dout + b * S * D + (int64_t)h * HD;
  const bf16* qtbase = TR ? nullptr : qt_ + bh * HD * Sp;
  const bf16* dotbase = TR ? nullptr : dot_ + bh * HD * Sp;
  const float* lse_b = lse + bh * Sp;
  const float* delta_b = delta + bh * Sp;

  bf16x8 kf[4], vf[4];
  {
    const bf16* kp = qkv + (b * S + kld) * D3 + D + (int64_t)h * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(kp + D + 16 * s);
#pragma unroll
      for (int e = 0; e < 8; ++e) vf[s][e] = -v[e];  // (see dkv3_tile)
    }
  }
  f32x16 dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
  int foff[4];
  {
    const int pli = pi32(li);
#pragma unroll
    for (int s = 0; s < 4; ++s) foff[s] = lds_tile_off(pli, 2 * s + hi);
  }
  const int qt_first = k0 / 64, qt_last = (S - 1) / 64;

  auto stage_all = [&](int qt, char* dst) {  // (see attn_bwd_dkv_kernel)
    stage64u(qbase, (int)D3, qt * 64, S - 1, 0, dst, wave, lane);
    stage64u(dobase, (int)D, qt * 64, S - 1, 0, dst + TILE64, wave, lane);
    if (!TR) {
      stage64u(qtbase, Sp, 0, HD - 1, qt * 64, dst + 2 * TILE64, wave, lane);
      stage64u(dotbase, Sp, 0, HD - 1, qt * 64, dst + 3 * TILE64, wave, lane);
    }
    if (wave < 2) {
      const float* base = (wave == 0 ? lse_b : delta_b) + (int64_t)qt * 64;
      glds16_s(base, (unsigned)(lane & 15) * 16u, __builtin_amdgcn_readfirstlane(lds_u32(dst + NT * TILE64 + wave * 1024)));
    }
  };
  int trof[2][2];
  tr_frag_offsets(lane, trof);
  if (qt_first <= qt_last) stage_all(qt_first, smem);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    asm volatile("" : "+v"(kf[s]));
    asm volatile("" : "+v"(vf[s]));
  }
  stage_wait_all();
    __syncthreads();
  // This wave's query tiles, in order: n_skip tiles entirely before its keys (nothing to compute), ONE tile crossing its
  // diagonal (tile a = kw0 / 64; masked), the full tiles up to S / 64, and a ragged last tile (masked; S % 64 != 0).
  // Counted loops with one body each: with the class tests inside the loop conditions hipcc rotated the loops and copied
  // all four accumulators in and out on every tile.
  const int nT = qt_last - qt_first + 1;
  const int a = kw0 >> 6;
  int n_skip = a - qt_first;
  if (n_skip > nT) n_skip = nT;
  const int n_mask = (a <= qt_last) ? 1 : 0;
  int n_full = ((S >> 6) < qt_last + 1 ? (S >> 6) : qt_last + 1) - (a + 1);
  if (n_full < 0) n_full = 0;
  const int n_tail = nT - n_skip - n_mask - n_full;
  int qt = qt_first;
  auto head = [&](int q) {
    if (q + 1 <= qt_last) stage_all(q + 1, smem + (((q - qt_first) & 1) ^ 1) * STG);
    return smem + ((q - qt_first) & 1) * STG;
  };
  for (int i = 0; i < n_skip; ++i, ++qt) {
    head(qt);
    stage_wait_all();
    __syncthreads();
  }
  if (n_mask) {
    const char* cur = head(qt);
    dkv3_tile<true, TR, PS>(cur, cur + TILE64, cur + 2 * TILE64, cur + 3 * TILE64, cur + NT * TILE64, trof, foff, kf, vf, dkacc, dvacc, hi,
                    krow - qt * 64, S - qt * 64, sc);
    stage_wait_all();
    __syncthreads();
    ++qt;
  }
  for (int i = 0; i < n_full; ++i, ++qt) {
    const char* cur = head(qt);
    dkv3_tile<false, TR, PS>(cur, cur + TILE64, cur + 2 * TILE64, cur + 3 * TILE64, cur + NT * TILE64, trof, foff, kf, vf, dkacc, dvacc, hi, 0,
                     64, sc);
    stage_wait_all();
    __syncthreads();
  }
  if (n_tail > 0) {
    const char* cur = head(qt);
    dkv3_tile<true, TR, PS>(cur, cur + TILE64, cur + 2 * TILE64, cur + 3 * TILE64, cur + NT * TILE64, trof, foff, kf, vf, dkacc, dvacc, hi,
                    krow - qt * 64, S - qt * 64, sc);
    stage_wait_all();
    __syncthreads();
  }
  if (krow < S) {
    bf16* krow_out = dqkv + (b * S + krow) * D3 + D + (int64_t)h * HD;
    const float rsc = rowscale != nullptr ? rowscale[b * S + krow] : 1.f;
    store_grad_row(krow_out, dkacc, -scale * rsc, hi, cos_t, sin_t, krow);  // (the accumulator holds -dK)
    store_grad_row(krow_out + D, dvacc, rsc, hi, nullptr, nullptr, 0);
  }
}

int mh_attn_fwd_mfma3(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                      hipStream_t st, int64_t q_start) {
  MH_REQUIRE(S * 3 * H * HD < (int64_t(1) << 31), "attn_fwd: sequence too long (32-bit panel offsets)");
  MH_REQUIRE(q_start >= 0 && q_start < S, "attn_fwd: q_start %ld outside the sequence", (long)q_start);
  const int64_t Sp = (S + 63) / 64 * 64;
  const int nt_all = (int)((S + 127) / 128), BH0 = (int)(B * H);
  MH_REQUIRE(BH0 < (1 << 24), "attn_fwd: too many (batch, head) pairs");
  const int BH = BH0 | ((g_attn_passes & 127) << 24);  // (attn_work unpacks the pass count)
  const int nt = nt_all - (int)(q_start / 128);  // query tiles holding rows >= q_start (the first of them may start below it)
  const unsigned grid = (unsigned)(nt * 8 * ((BH0 + 7) / 8));
#define MH_FWD(WPS, TR_, LZ_)                                                                                                    \
  attn_fwd3_kernel<WPS, TR_, 2, LZ_><<<grid, 256, 4 * TILE64, st>>>((const bf16*)qkv, (const bf16*)vt, (bf16*)o, lse, (int)S, (int)Sp, H, \
                                                                 scale * LOG2E, BH, nt, nt_all)
#define MH_FWD3S(WPS, TR_, LZ_)                                                                                                  \
  attn_fwd3_kernel<WPS, TR_, 3, LZ_><<<grid, 256, 6 * TILE64, st>>>((const bf16*)qkv, (const bf16*)vt, (bf16*)o, lse, (int)S, (int)Sp, \
                                                                  H, scale * LOG2E, BH, nt, nt_all)
  const bool tr = (vt == nullptr);  // no prepared V^T copy: transpose reads
  const bool lz = (g_attn_v3 & 128) != 0;  // lazy reference maximum (fwd3_tile)
  // (register budget: with transpose reads the 256-register build measured 1-2 % ahead, with the prepared copy the
  //  168-register one; both fit three waves per SIMD -- profiles/r02_run18_attn_forms_ab.txt)
  if (tr && g_attn_v3_wps == 4) {  // A/B: 128 registers, two stages (32 KiB): four workgroups per CU
    if (lz) MH_FWD(4, true, true); else MH_FWD(4, true, false);
  } else if (tr && (g_attn_v3 & 64)) {  // three stages: -3 % at S = 2048, -1...-3 % at 4096 (profiles/r02_run22_*); the same in the
                                 // backward pair measured +1.5 % (their tiles are twice as long) and is not built
    if (lz) {
      if (g_attn_v3_wps == 3) MH_FWD3S(3, true, true); else MH_FWD3S(2, true, true);
    } else {
      if (g_attn_v3_wps == 3) MH_FWD3S(3, true, false); else MH_FWD3S(2, true, false);
    }
  } else if (tr) {
    if (g_attn_v3_wps == 3) MH_FWD(3, true, false); else MH_FWD(2, true, false);
  } else {
    if (g_attn_v3_wps == 2) MH_FWD(2, false, false); else MH_FWD(3, false, false);
  }
#undef MH_FWD
#undef MH_FWD3S
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// A/B library only: the production forward (three stages, transpose reads; lazy = the r06 lazy reference maximum) with s_memtime
// stamps at the seams of every tile's segments; `stamps` receives uint32 [16 workgroups][4 waves][32 tiles][9]: low words of the
// shader clock at: top of the tile, LDS-DMA issued, K reads issued, S MFMAs + V^T reads issued, softmax issued, V^T landed,
// P V MFMAs issued, next tile landed (vmcnt), barrier passed -- for tiles 8 .. 39 of the workgroups' loops (tools/attn_timeline.py);
// behind them uint32 [grid][8]: every workgroup's start / end clock, HW_ID, XCC_ID and tile count (the residency census), and
// behind those ONE int32 the host writes: the first recorded tile.
extern "C" int mh_attn_fwd_timeline(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, int lazy,
                                    uint32_t* stamps, void* stream) {
#ifdef MH_AB_BUILDS
  MH_REQUIRE(stamps != nullptr && S >= 128 && S * 3 * H * HD < (int64_t(1) << 31), "attn_fwd_timeline: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t Sp = (S + 63) / 64 * 64;
  const int nt_all = (int)((S + 127) / 128), BH0 = (int)(B * H), nt = nt_all;
  const int BH = BH0 | ((g_attn_passes & 127) << 24);
  const unsigned grid = (unsigned)(nt * 8 * ((BH0 + 7) / 8));
  if (lazy)
    attn_fwd3_kernel<2, true, 3, true, true><<<grid, 256, 6 * TILE64 + TL_BYTES, st>>>((const bf16*)qkv, (const bf16*)stamps, (bf16*)o, lse, (int)S,
                                                                                   (int)Sp, H, scale * LOG2E, BH, nt, nt_all);
  else
    attn_fwd3_kernel<2, true, 3, false, true><<<grid, 256, 6 * TILE64 + TL_BYTES, st>>>((const bf16*)qkv, (const bf16*)stamps, (bf16*)o, lse, (int)S,
                                                                                    (int)Sp, H, scale * LOG2E, BH, nt, nt_all);
  MH_LAUNCH_CHECK();
  return MH_OK;
#else
  mh_set_error("attn_fwd_timeline: only in the A/B library (libmidihip_ab.so)");
  return MH_ERR_UNSUPPORTED;
#endif
}

int mh_attn_bwd_mfma3(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt, const void* kt,
                      const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale, const float* cos_t,
                      const float* sin_t, int which /* bit 1: dQ, bit 2: dK/dV, bit 3: transpose reads (no qt / kt / dot) */,
                      hipStream_t st, const void* o /* != NULL: the dQ kernel computes delta and WRITES it (needs bit 1) */) {
  MH_REQUIRE(o == nullptr || (which & 14) == 14, "attn_bwd: delta inside the dQ kernel needs the third form of both kernels");
  MH_REQUIRE(S * 3 * H * HD < (int64_t(1) << 31), "attn_bwd: sequence too long (32-bit panel offsets)");
  const bool tr = (which & 8) != 0;
  MH_REQUIRE(tr || (qt != nullptr && kt != nullptr && dot != nullptr), "attn_bwd(bf16): needs the transposed copies (mh_attn_prep_bwd)");
  const int64_t Sp = (S + 63) / 64 * 64;
  const int nt = (int)((S + 127) / 128), BH0 = (int)(B * H);
  MH_REQUIRE(BH0 < (1 << 24), "attn_bwd: too many (batch, head) pairs");
  const int BH = BH0 | ((g_attn_passes & 127) << 24);  // (attn_work unpacks the pass count)
  const unsigned grid = (unsigned)(nt * 8 * ((BH0 + 7) / 8));
#define MH_DQ(WPS, TR_)                                                                                                     \
  attn_bwd_dq3_kernel<WPS, TR_><<<grid, 256, (TR_ ? 4 : 6) * TILE64, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta,     \
                                                                        (const bf16*)kt, (bf16*)dqkv, (int)S, (int)Sp, H, scale, \
                                                                        BH, nt, cos_t, sin_t, (const bf16*)o,         \
                                                                        const_cast<float*>(delta), g_attn_bwd_rowscale)
  if (which & 2) {
    if (g_attn_v3_wps == 2) {
      if (tr) MH_DQ(2, true); else MH_DQ(2, false);
    } else {
      if (tr) MH_DQ(3, true); else MH_DQ(3, false);
    }
    MH_LAUNCH_CHECK();
  }
#undef MH_DQ
  if (which & 4) {
    if (tr && o != nullptr)  // (statistics pre-scaled by the dQ kernel above)
      attn_bwd_dkv3_kernel<true, true><<<grid, 256, 2 * (2 * TILE64 + 2048), st>>>((const bf16*)qkv, (const bf16*)dout,
                                                                                 delta + (int64_t)BH0 * Sp, delta, nullptr, nullptr,
                                                                                 (bf16*)dqkv, (int)S, (int)Sp, H, scale, BH, nt,
                                                                                 cos_t, sin_t, g_attn_bwd_rowscale);
    else if (tr)
      attn_bwd_dkv3_kernel<true><<<grid, 256, 2 * (2 * TILE64 + 2048), st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, nullptr,
                                                                           nullptr, (bf16*)dqkv, (int)S, (int)Sp, H, scale, BH, nt,
                                                                           cos_t, sin_t, g_attn_bwd_rowscale);
    else
      attn_bwd_dkv3_kernel<false><<<grid, 256, 2 * DKV_STAGE, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, (const bf16*)qt,
                                                                  (const bf16*)dot, (bf16*)dqkv, (int)S, (int)Sp, H, scale, BH, nt,
                                                                  cos_t, sin_t, g_attn_bwd_rowscale);
    MH_LAUNCH_CHECK();
  }
  return MH_OK;
}
