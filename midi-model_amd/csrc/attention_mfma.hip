// Event-level causal attention (head_dim 64, bf16) on the matrix cores: flash forward + two-kernel backward -- the FIRST
// form of the three kernels (r01), kept as the A/B baseline of the third form (attention_mfma3.hip, the default;
// mh_set_option("attn_v3", 0) selects this one) and as an independent second implementation for the tests.  This file also
// holds the dispatch of mh_attn_fwd / mh_attn_bwd between the forms.
//
// Orientation (all three kernels): the MFMA output tile always has the softmax ROW index on the lane axis
// (lane&31) and the reduction index in registers, so running max / sum / lse / delta are lane-local and
// the probability tile is converted to the next MFMA's operand in registers (no LDS round trip, no
// cross-lane transpose):
//   forward   S^T = K Q^T   ->  O^T += V^T P^T         (lane = query row)
//   dQ        S^T, dP^T = V dO^T  ->  dQ^T += K^T dS^T  (lane = query row)
//   dK,dV     S = Q K^T, dP = dO V^T -> dV^T += dO^T P, dK^T += Q^T dS   (lane = key row)
// v_mfma_f32_32x32x16_bf16 leaves rows (r&3)+8(r>>2)+4hi in register r; feeding operand rows through the
// bit-2/3 swap pi32() makes registers 8t..8t+7 a contiguous run of 8 reduction indices = one operand
// fragment of the following MFMA (common.h).
// Operands whose reduction index is NOT the contiguous one in qkv (V^T, Q^T, K^T, dO^T) are read from
// transposed copies [B,H,64,Sp] written by the prep kernels (the third form reads them out of the row-major
// tiles with ds_read_b64_tr_b16 instead).  All tiles are 64 rows x 128 B in the swizzled LDS format of common.h and
// arrive by global_load_lds (double buffered, one barrier per tile).
//
// VALU budget.  At head_dim 64 a 64-key tile is only 16 MFMAs (512 cycles/wave) against 2048 softmax
// elements per wave, so the per-element VALU cost decides the kernel.  r01 run 1 measured 237 TF/s with ~13
// VALU instructions per element (always-on 64-bit causal-mask compares, exp2f's denormal path); this
// version keeps 3-4: the causal mask is compiled only into the diagonal-tile instantiation (MASK), index
// math is 32-bit and tile-relative, the scale is folded into one fma feeding a raw v_exp_f32, and row maxima
// are taken on the unscaled scores.
// Roofline: MFMA (2.5 PFLOP/s bf16 dense), VALU co-limited.
#include "attn_mfma_common.h"

// The first-form kernels are compiled only into the A/B test library (libmidihip_ab.so, -DMH_AB_BUILDS, build.py): the
// production library holds the third form and the dispatch below, and refuses the option values that select this form.
#ifdef MH_AB_BUILDS
// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
// one 64-key tile for one wave (32 query rows).  qrel = query row - first key of the tile.
template <bool MASK>
__device__ inline void fwd_tile(const char* tK, const char* tV, const bf16x8 (&qf)[4], f32x16 (&oacc)[2], float& m,
                                float& l, int pli, int hi, int qrel, float sc) {
  f32x16 sacc[2] = {zero16(), zero16()};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s = 0; s < 4; ++s) sacc[kb] = mfma32(lds_frag(tK, kb * 32 + pli, 2 * s + hi), qf[s], sacc[kb]);
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MASK) {
        if (kb * 32 + reg_index(r, hi) > qrel) sacc[kb][r] = -INFINITY;
      }
      mx = fmaxf(mx, sacc[kb][r]);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * sc;  // running max kept in scaled (log2) units
  // Rescale O only when some row's maximum grew by more than RESCALE_THR (log2 units): rows keep their older,
  // slightly smaller reference max, their probabilities may reach 2^THR instead of 1, and l / O / lse stay mutually
  // consistent (the maths is exact; only the bf16 rounding of P sees the larger magnitudes).  With THR = 0 a wave of
  // 32 rows still rescaled on most tiles (r01 PMC: 21 VALU per MFMA); the rescale costs an AGPR<->VGPR round trip
  // of the 32 accumulator registers.
  float mn = m, alpha = 1.f;
  if (__any(mx > m + RESCALE_THR)) {
    mn = fmaxf(m, mx);
    alpha = fast_exp2(m - mn);
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
  }
  float psum = 0.f;
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = fast_exp2(__builtin_fmaf(sacc[kb][r], sc, -mn));
      sacc[kb][r] = p;
      psum += p;
    }
  l = l * alpha + psum;
  m = mn;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const bf16x8 pf = pack8(sacc[t >> 1], 8 * (t & 1));
#pragma unroll
    for (int db = 0; db < 2; ++db) oacc[db] = mfma32(lds_frag(tV, db * 32 + pli, 2 * t + hi), pf, oacc[db]);
  }
}

// (three workgroups per CU: capped at 168 VGPRs the compiler still spills nothing and the forward went from 334 to 274 us
// per layer; at four -- 128 VGPRs -- it spills 31 registers and takes 428 us.  dQ likewise 3, dK/dV is held to 2 by its 64 KiB.)
__global__ __launch_bounds__(256, 3) void attn_fwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ vt,
                                                       bf16* __restrict__ o, float* __restrict__ lse, int S, int Sp, int H,
                                                       float sc /* scale*log2(e) */, int BH, int nqt) {
  // dynamic LDS, 4 tiles = [stage][K | V^T] (r02: with a static array hipcc's LDS-DMA alias tracking waits vmcnt(0) before
  // the first fragment read of a tile, i.e. for the NEXT tile's stage requested just above it -- the double buffer then
  // hides nothing; the barrier at the end of the iteration is where that stage has to have landed)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bh_, tile_;
  if (!attn_work(BH, nqt, bh_, tile_)) return;
  const int64_t bh = bh_;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int q0 = (nqt - 1 - tile_) * 128;  // heavy (late) query tiles first
  const int qw0 = q0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int qrow = qw0 + li;
  const int qld = (qrow < S) ? qrow : S - 1;

  const bf16* kbase = qkv + b * S * D3 + D + (int64_t)h * HD;
  const bf16* vtbase = vt + bh * HD * Sp;

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
  const int pli = pi32(li);

  stage64(kbase, D3, 0, S - 1, 0, smem, wave, lane);
  stage64(vtbase, Sp, 0, HD - 1, 0, smem + TILE64, wave, lane);
#pragma unroll
  for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[s]));  // (see attn_bwd_dkv_kernel)
  __syncthreads();
  for (int kt = 0; kt <= kt_last; ++kt) {
    const char* cur = smem + (kt & 1) * 2 * TILE64;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE64;
    if (kt + 1 <= kt_last) {
      stage64(kbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt, wave, lane);
      stage64(vtbase, Sp, 0, HD - 1, (int64_t)(kt + 1) * 64, nxt + TILE64, wave, lane);
    }
    const int k0 = kt * 64;
    if (k0 <= qw0 + 31) {  // wave-uniform: this wave still has unmasked keys in the tile
      if (k0 + 63 > qw0)   // wave-uniform: the tile crosses this wave's diagonal
        fwd_tile<true>(cur, cur + TILE64, qf, oacc, m, l, pli, hi, qrow - k0, sc);
      else
        fwd_tile<false>(cur, cur + TILE64, qf, oacc, m, l, pli, hi, 0, sc);
    }
    __syncthreads();
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
template <bool MASK>
__device__ inline void dq_tile(const char* tK, const char* tV, const char* tKT, const bf16x8 (&qf)[4],
                               const bf16x8 (&dof)[4], f32x16 (&dqacc)[2], int pli, int hi, int qrel, float sc,
                               float lse2, float dl) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = mfma32(lds_frag(tK, kb * 32 + pli, 2 * s + hi), qf[s], sacc);
      pacc = mfma32(lds_frag(tV, kb * 32 + pli, 2 * s + hi), dof[s], pacc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float p = fast_exp2(__builtin_fmaf(sacc[r], sc, -lse2));
      if (MASK) {
        if (kb * 32 + reg_index(r, hi) > qrel) p = 0.f;
      }
      sacc[r] = p * (pacc[r] - dl);  // dS (unscaled)
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 dsf = pack8(sacc, 8 * t);
#pragma unroll
      for (int hb = 0; hb < 2; ++hb)
        dqacc[hb] = mfma32(lds_frag(tKT, hb * 32 + pli, 4 * kb + 2 * t + hi), dsf, dqacc[hb]);
    }
  }
}

__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             const bf16* __restrict__ kt_, bf16* __restrict__ dqkv, int S,
                                                             int Sp, int H, float scale, int BH, int nqt,
                                                             const float* __restrict__ cos_t,
                                                             const float* __restrict__ sin_t) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 6 tiles: [stage][K | V | K^T] (dynamic: see attn_fwd_kernel)
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
  const bf16* ktbase = kt_ + bh * HD * Sp;

  bf16x8 qf[4], dof[4];
  {
    const bf16* qp = qkv + (b * S + qld) * D3 + (int64_t)h * HD + 8 * hi;
    const bf16* dp = dout + (b * S + qld) * D + (int64_t)h * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
    }
  }
  const float lse2 = lse[bh * Sp + qld] * LOG2E;
  const float dl = delta[bh * Sp + qld];
  f32x16 dqacc[2] = {zero16(), zero16()};

  int last_q = q0 + 127;
  if (last_q > S - 1) last_q = S - 1;
  const int kt_last = last_q / 64;
  const int pli = pi32(li);

  stage64(kbase, D3, 0, S - 1, 0, smem, wave, lane);
  stage64(vbase, D3, 0, S - 1, 0, smem + TILE64, wave, lane);
  stage64(ktbase, Sp, 0, HD - 1, 0, smem + 2 * TILE64, wave, lane);
#pragma unroll
  for (int s = 0; s < 4; ++s) {  // (see attn_bwd_dkv_kernel)
    asm volatile("" : "+v"(qf[s]));
    asm volatile("" : "+v"(dof[s]));
  }
  __syncthreads();
  for (int kt = 0; kt <= kt_last; ++kt) {
    const char* cur = smem + (kt & 1) * 3 * TILE64;
    char* nxt = smem + ((kt + 1) & 1) * 3 * TILE64;
    if (kt + 1 <= kt_last) {
      stage64(kbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt, wave, lane);
      stage64(vbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt + TILE64, wave, lane);
      stage64(ktbase, Sp, 0, HD - 1, (int64_t)(kt + 1) * 64, nxt + 2 * TILE64, wave, lane);
    }
    const int k0 = kt * 64;
    if (k0 <= qw0 + 31) {
      if (k0 + 63 > qw0)
        dq_tile<true>(cur, cur + TILE64, cur + 2 * TILE64, qf, dof, dqacc, pli, hi, qrow - k0, sc, lse2, dl);
      else
        dq_tile<false>(cur, cur + TILE64, cur + 2 * TILE64, qf, dof, dqacc, pli, hi, 0, sc, lse2, dl);
    }
    __syncthreads();
  }
  if (qrow < S) {
    bf16* orow = dqkv + (b * S + qrow) * D3 + (int64_t)h * HD;
    store_grad_row(orow, dqacc, scale, hi, cos_t, sin_t, qrow);
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, dK/dV: block = 128 key rows, loop over the query tiles that see them
// ---------------------------------------------------------------------------------------------------
// one 64-query tile for one wave (32 key rows).  krel = key row - first query of the tile; qlim = number of
// valid queries in the tile (S - q0, may exceed 64).
template <bool MASK>
__device__ inline void dkv_tile(const char* tQ, const char* tDO, const char* tQT, const char* tDOT, const bf16x8 (&kf)[4],
                                const bf16x8 (&vf)[4], f32x16 (&dkacc)[2], f32x16 (&dvacc)[2], const char* tLD, int pli, int hi,
                                int krel, int qlim, float sc) {
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      sacc = mfma32(lds_frag(tQ, qb * 32 + pli, 2 * s + hi), kf[s], sacc);
      pacc = mfma32(lds_frag(tDO, qb * 32 + pli, 2 * s + hi), vf[s], pacc);
    }
    // lse / delta of the tile's 64 queries come from the stage (256 B each, staged with the tile by one LDS-DMA): read
    // straight from global memory here (r01) they were 16 ordinary loads per tile whose L2 round trips sat between the
    // S / dP MFMAs and the dS arithmetic of every query block, and -- ordinary loads beside LDS-DMA -- made hipcc drain the
    // stage in flight.  Entries past S are never-written padding, discarded by the MASK select (the last tile is MASK).
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qq = qb * 32 + 16 * t + 8 * hi;
#pragma unroll
      for (int v4 = 0; v4 < 2; ++v4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tLD + (qq + 4 * v4) * 4);
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(tLD + 1024 + (qq + 4 * v4) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 8 * t + 4 * v4 + e;
          float p = fast_exp2(__builtin_fmaf(sacc[r], sc, -a[e] * LOG2E));
          float ds = p * (pacc[r] - d4[e]);
          if (MASK) {
            const int q = qq + 4 * v4 + e;
            const bool ok = (q >= krel) && (q < qlim);
            p = ok ? p : 0.f;
            ds = ok ? ds : 0.f;
          }
          sacc[r] = p;
          pacc[r] = ds;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 pf = pack8(sacc, 8 * t), dsf = pack8(pacc, 8 * t);
#pragma unroll
      for (int xb = 0; xb < 2; ++xb) {
        dvacc[xb] = mfma32(lds_frag(tDOT, xb * 32 + pli, 4 * qb + 2 * t + hi), pf, dvacc[xb]);
        dkacc[xb] = mfma32(lds_frag(tQT, xb * 32 + pli, 4 * qb + 2 * t + hi), dsf, dkacc[xb]);
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              const bf16* __restrict__ qt_, const bf16* __restrict__ dot_,
                                                              bf16* __restrict__ dqkv, int S, int Sp, int H, float scale,
                                                              int BH, int nkt, const float* __restrict__ cos_t,
                                                              const float* __restrict__ sin_t) {
  constexpr int STG = DKV_STAGE;  // [Q | dO | Q^T | dO^T | lse (256 B of a KiB) | delta (256 B of a KiB)]
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages (dynamic: see attn_fwd_kernel)
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
  const bf16* qtbase = qt_ + bh * HD * Sp;
  const bf16* dotbase = dot_ + bh * HD * Sp;
  const float* lse_b = lse + bh * Sp;
  const float* delta_b = delta + bh * Sp;

  bf16x8 kf[4], vf[4];
  {
    const bf16* kp = qkv + (b * S + kld) * D3 + D + (int64_t)h * HD + 8 * hi;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
      vf[s] = *reinterpret_cast<const bf16x8*>(kp + D + 16 * s);
    }
  }
  f32x16 dkacc[2] = {zero16(), zero16()}, dvacc[2] = {zero16(), zero16()};
  const int pli = pi32(li);
  const int qt_first = k0 / 64, qt_last = (S - 1) / 64;

  auto stage_all = [&](int qt, char* dst) {
    stage64(qbase, D3, (int64_t)qt * 64, S - 1, 0, dst, wave, lane);
    stage64(dobase, D, (int64_t)qt * 64, S - 1, 0, dst + TILE64, wave, lane);
    stage64(qtbase, Sp, 0, HD - 1, (int64_t)qt * 64, dst + 2 * TILE64, wave, lane);
    stage64(dotbase, Sp, 0, HD - 1, (int64_t)qt * 64, dst + 3 * TILE64, wave, lane);
    if (wave < 2) {  // wave 0: lse[qt*64 ..], wave 1: delta[..]: lanes 0-15 carry the 256 B, the others repeat them (the
      // LDS-DMA writes a full KiB per wave); a wave-uniform base + a lane offset recomputed here keeps no address register
      // alive across the tile loop (a per-lane pointer select was spilled and reloaded behind a vmcnt(0))
      const float* base = (wave == 0 ? lse_b : delta_b) + (int64_t)qt * 64;
      int l15 = lane & 15;
      asm volatile("" : "+v"(l15));  // (not loop-invariant for hipcc: the address is rebuilt here, nothing stays live)
      glds16(base + 4 * l15, dst + 4 * TILE64 + wave * 1024);
    }
  };
  if (qt_first <= qt_last) stage_all(qt_first, smem);
  // (register-resident operands through an empty asm: left pending, hipcc would wait vmcnt(0) at their first use inside the
  // loop on every tile, draining the stage in flight -- see attn_fwd2_kernel)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    asm volatile("" : "+v"(kf[s]));
    asm volatile("" : "+v"(vf[s]));
  }
  __syncthreads();
  for (int qt = qt_first; qt <= qt_last; ++qt) {
    const int st = (qt - qt_first) & 1;
    const char* cur = smem + st * STG;
    char* nxt = smem + (st ^ 1) * STG;
    if (qt + 1 <= qt_last) stage_all(qt + 1, nxt);
    const int qs = qt * 64;
    if (qs + 63 >= kw0) {  // wave-uniform: some query of this tile sees this wave's keys
      // the causal mask matters when the tile's first query is below the wave's last key; the ragged tail
      // (queries past S) only exists in the last tile
      if (qs < kw0 + 31 || qs + 64 > S)
        dkv_tile<true>(cur, cur + TILE64, cur + 2 * TILE64, cur + 3 * TILE64, kf, vf, dkacc, dvacc, cur + 4 * TILE64, pli, hi,
                       krow - qs, S - qs, sc);
      else
        dkv_tile<false>(cur, cur + TILE64, cur + 2 * TILE64, cur + 3 * TILE64, kf, vf, dkacc, dvacc, cur + 4 * TILE64, pli, hi,
                        0, 64, sc);
    }
    __syncthreads();
  }
  if (krow < S) {
    bf16* krow_out = dqkv + (b * S + krow) * D3 + D + (int64_t)h * HD;
    store_grad_row(krow_out, dkacc, scale, hi, cos_t, sin_t, krow);
    store_grad_row(krow_out + D, dvacc, 1.f, hi, nullptr, nullptr, 0);
  }
}

#endif  // MH_AB_BUILDS

// ---------------------------------------------------------------------------------------------------
extern thread_local int g_attn_v3;  // attention_mfma3.hip: the third form of the three kernels (bit 0 forward, bit 1 dQ, bit 2 dK/dV, bit 3
                       // transpose reads in the backward pair instead of the prepared copies)
int mh_attn_fwd_mfma3(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                      hipStream_t st, int64_t q_start = 0);
int mh_attn_bwd_mfma3(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt, const void* kt,
                      const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale, const float* cos_t,
                      const float* sin_t, int which, hipStream_t st, const void* o = nullptr);

int mh_attn_fwd_mfma(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                     hipStream_t st, int64_t q_start) {
  MH_REQUIRE(S < (1 << 24), "attn_fwd: sequence too long");
  if ((g_attn_v3 & 1) && (vt != nullptr || (g_attn_v3 & 16)))  // (vt == NULL + bit 4: V through transpose reads)
    return mh_attn_fwd_mfma3(qkv, vt, o, lse, B, S, H, scale, st, q_start);
  MH_REQUIRE(q_start == 0, "attn_fwd_tail: served by the third form of the forward kernel only");
#ifndef MH_AB_BUILDS
  MH_REQUIRE(false, "attn_fwd(bf16): option attn_v3 = %d selects the first form of the kernel, which is only in the A/B test "
             "library (libmidihip_ab.so)", g_attn_v3);
#else
  MH_REQUIRE(vt != nullptr, "attn_fwd(bf16): the first form needs the transposed V copy (mh_attn_prep_fwd)");
  const int64_t Sp = (S + 63) / 64 * 64;
  const int nt = (int)((S + 127) / 128), BH = (int)(B * H);
  const unsigned grid = (unsigned)(nt * 8 * ((BH + 7) / 8));
  attn_fwd_kernel<<<grid, 256, 4 * TILE64, st>>>((const bf16*)qkv, (const bf16*)vt, (bf16*)o, lse, (int)S, (int)Sp, H,
                                        scale * LOG2E, BH, nt);
  MH_LAUNCH_CHECK();
  return MH_OK;
#endif
}

// backward in one call, delta = rowsum(dO * O) computed by the dQ kernel (third form with transpose reads only)
int mh_attn_bwd_o_mfma(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int64_t B,
                       int64_t S, int H, float scale, const float* cos_t, const float* sin_t, hipStream_t st) {
  MH_REQUIRE(S < (1 << 24), "attn_bwd: sequence too long");
  MH_REQUIRE((g_attn_v3 & 14) == 14, "attn_bwd_o: served by the third form with transpose reads (mh_set_option(\"attn_v3\") bits 2|4|8); "
             "use mh_attn_prep_bwd + mh_attn_bwd otherwise");
  return mh_attn_bwd_mfma3(qkv, dout, lse, delta, nullptr, nullptr, nullptr, dqkv, B, S, H, scale, cos_t, sin_t, 14, st, o);
}

int mh_attn_bwd_mfma(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt,
                     const void* kt, const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale,
                     const float* cos_t, const float* sin_t, hipStream_t st) {
  MH_REQUIRE(S < (1 << 24), "attn_bwd: sequence too long");
  if ((g_attn_v3 & 14) == 14)  // third form of both kernels with transpose reads: no transposed copies needed
    return mh_attn_bwd_mfma3(qkv, dout, lse, delta, nullptr, nullptr, nullptr, dqkv, B, S, H, scale, cos_t, sin_t, 14, st);
  MH_REQUIRE(qt != nullptr && kt != nullptr && dot != nullptr, "attn_bwd(bf16): needs the transposed copies (mh_attn_prep_bwd)");
  const int64_t Sp = (S + 63) / 64 * 64;
  const int nt = (int)((S + 127) / 128), BH = (int)(B * H);
  const unsigned grid = (unsigned)(nt * 8 * ((BH + 7) / 8));
  if (g_attn_v3 & 6) {
    const int rc = mh_attn_bwd_mfma3(qkv, dout, lse, delta, qt, kt, dot, dqkv, B, S, H, scale, cos_t, sin_t, g_attn_v3 & 14, st);
    if (rc != MH_OK) return rc;
  }
#ifndef MH_AB_BUILDS
  MH_REQUIRE((g_attn_v3 & 6) == 6, "attn_bwd(bf16): option attn_v3 = %d selects first-form kernels, which are only in the A/B "
             "test library (libmidihip_ab.so)", g_attn_v3);
#else
  if (!(g_attn_v3 & 2)) {
    attn_bwd_dq_kernel<<<grid, 256, 6 * TILE64, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, (const bf16*)kt,
                                             (bf16*)dqkv, (int)S, (int)Sp, H, scale, BH, nt, cos_t, sin_t);
    MH_LAUNCH_CHECK();
  }
  if (!(g_attn_v3 & 4)) {
    attn_bwd_dkv_kernel<<<grid, 256, 2 * DKV_STAGE, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, (const bf16*)qt,
                                              (const bf16*)dot, (bf16*)dqkv, (int)S, (int)Sp, H, scale, BH, nt, cos_t, sin_t);
    MH_LAUNCH_CHECK();
  }
#endif
  (void)grid;
  return MH_OK;
}
