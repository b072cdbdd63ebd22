// Event-level causal attention (head_dim 64, bf16) on the matrix cores: flash forward + two-kernel backward.
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
// transposed copies [B,H,64,Sp] written by the prep kernels (first structure; to be replaced by
// ds_read_b64_tr_b16 staging).  All tiles are 64 rows x 128 B in the swizzled LDS format of common.h and
// arrive by global_load_lds (double buffered, one barrier per tile).
// Roofline: MFMA (2.5 PFLOP/s bf16 dense); at head_dim 64 the exp/VALU work per MFMA is twice that of
// head_dim 128, so the VALU pipe is the co-limiter.
#include "common.h"

constexpr int HD = 64;
constexpr int TILE64 = 64 * 128;  // bytes

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

// stage a 64-row x 64-col bf16 tile: rows row0.. (clamped to row_clamp), columns col0..col0+63
__device__ inline void stage64(const bf16* __restrict__ base, int64_t ld, int64_t row0, int64_t row_clamp, int64_t col0,
                               char* lds_tile, int wave, int lane) {
  const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int g8 = wave + 4 * it;
    const int r = g8 * 8 + rsub;
    const int c = pc ^ ((r >> 1) & 7);
    int64_t grow = row0 + r;
    if (grow > row_clamp) grow = row_clamp;
    glds16(base + grow * ld + col0 + c * 8, lds_tile + g8 * 1024);
  }
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ vt,
                                                       bf16* __restrict__ o, float* __restrict__ lse, int64_t S,
                                                       int64_t Sp, int H, float sc /* scale*log2(e) */) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE64];  // [stage][K | V^T]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int nqt = gridDim.x;
  const int64_t q0 = (int64_t)(nqt - 1 - blockIdx.x) * 128;  // heavy (late) query tiles first
  const int64_t qw0 = q0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int64_t qrow = qw0 + li;
  const int64_t qld = (qrow < S) ? qrow : S - 1;

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

  int64_t last_q = q0 + 127;
  if (last_q > S - 1) last_q = S - 1;
  const int kt_last = (int)(last_q / 64);
  const int pli = pi32(li);

  stage64(kbase, D3, 0, S - 1, 0, smem, wave, lane);
  stage64(vtbase, Sp, 0, HD - 1, 0, smem + TILE64, wave, lane);
  __syncthreads();
  for (int kt = 0; kt <= kt_last; ++kt) {
    const char* cur = smem + (kt & 1) * 2 * TILE64;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE64;
    if (kt + 1 <= kt_last) {
      stage64(kbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt, wave, lane);
      stage64(vtbase, Sp, 0, HD - 1, (int64_t)(kt + 1) * 64, nxt + TILE64, wave, lane);
    }
    if ((int64_t)kt * 64 <= qw0 + 31) {  // wave-uniform: this wave still has unmasked keys in the tile
      const char* tK = cur;
      const char* tV = cur + TILE64;
      f32x16 sacc[2] = {zero16(), zero16()};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[kb] = mfma32(lds_frag(tK, kb * 32 + pli, 2 * s + hi), qf[s], sacc[kb]);
      const bool need_mask = ((int64_t)kt * 64 + 63 > qw0);
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float x = sacc[kb][r] * sc;
          if (need_mask) {
            const int64_t key = (int64_t)kt * 64 + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
            if (key > qrow) x = -INFINITY;
          }
          sacc[kb][r] = x;
          mx = fmaxf(mx, x);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);
      const float alpha = exp2f(m - mn);
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = exp2f(sacc[kb][r] - mn);
          sacc[kb][r] = p;
          psum += p;
        }
      l = l * alpha + psum;
      m = mn;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bf16x8 pf = pack8(sacc[t >> 1], 8 * (t & 1));
#pragma unroll
        for (int db = 0; db < 2; ++db) oacc[db] = mfma32(lds_frag(tV, db * 32 + pli, 2 * t + hi), pf, oacc[db]);
      }
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
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          const bf16* __restrict__ kt_, bf16* __restrict__ dqkv, int64_t S,
                                                          int64_t Sp, int H, float scale) {
  __shared__ __attribute__((aligned(16))) char smem[6 * TILE64];  // [stage][K | V | K^T]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int nqt = gridDim.x;
  const int64_t q0 = (int64_t)(nqt - 1 - blockIdx.x) * 128;
  const int64_t qw0 = q0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int64_t qrow = qw0 + li;
  const int64_t qld = (qrow < S) ? qrow : S - 1;
  const float sc = scale * 1.4426950408889634f;

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
  const float lse2 = lse[bh * Sp + qld] * 1.4426950408889634f;
  const float dl = delta[bh * Sp + qld];
  f32x16 dqacc[2] = {zero16(), zero16()};

  int64_t last_q = q0 + 127;
  if (last_q > S - 1) last_q = S - 1;
  const int kt_last = (int)(last_q / 64);
  const int pli = pi32(li);

  stage64(kbase, D3, 0, S - 1, 0, smem, wave, lane);
  stage64(vbase, D3, 0, S - 1, 0, smem + TILE64, wave, lane);
  stage64(ktbase, Sp, 0, HD - 1, 0, smem + 2 * TILE64, wave, lane);
  __syncthreads();
  for (int kt = 0; kt <= kt_last; ++kt) {
    const char* cur = smem + (kt & 1) * 3 * TILE64;
    char* nxt = smem + ((kt + 1) & 1) * 3 * TILE64;
    if (kt + 1 <= kt_last) {
      stage64(kbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt, wave, lane);
      stage64(vbase, D3, (int64_t)(kt + 1) * 64, S - 1, 0, nxt + TILE64, wave, lane);
      stage64(ktbase, Sp, 0, HD - 1, (int64_t)(kt + 1) * 64, nxt + 2 * TILE64, wave, lane);
    }
    if ((int64_t)kt * 64 <= qw0 + 31) {
      const char* tK = cur;
      const char* tV = cur + TILE64;
      const char* tKT = cur + 2 * TILE64;
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
          const int64_t key = (int64_t)kt * 64 + kb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
          const float p = (key <= qrow) ? exp2f(sacc[r] * sc - lse2) : 0.f;
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
    __syncthreads();
  }
  if (qrow < S) {
    bf16* orow = dqkv + (b * S + qrow) * D3 + (int64_t)h * HD;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
      for (int r8 = 0; r8 < 2; ++r8) {
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (bf16)(dqacc[hb][8 * r8 + e] * scale);
        *reinterpret_cast<bf16x8*>(orow + hb * 32 + 16 * r8 + 8 * hi) = v;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
// backward, dK/dV: block = 128 key rows, loop over the query tiles that see them
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           const bf16* __restrict__ qt_, const bf16* __restrict__ dot_,
                                                           bf16* __restrict__ dqkv, int64_t S, int64_t Sp, int H,
                                                           float scale) {
  __shared__ __attribute__((aligned(16))) char smem[8 * TILE64];  // [stage][Q | dO | Q^T | dO^T]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD, D3 = 3 * D;
  const int64_t k0 = (int64_t)blockIdx.x * 128;
  const int64_t kw0 = k0 + wave * 32;
  const int li = lane & 31, hi = lane >> 5;
  const int64_t krow = kw0 + li;
  const int64_t kld = (krow < S) ? krow : S - 1;
  const float sc = scale * 1.4426950408889634f;

  const bf16* qbase = qkv + b * S * D3 + (int64_t)h * HD;
  const bf16* dobase = dout + b * S * D + (int64_t)h * HD;
  const bf16* qtbase = qt_ + bh * HD * Sp;
  const bf16* dotbase = dot_ + bh * HD * Sp;

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
  const int qt_first = (int)(k0 / 64), qt_last = (int)((S - 1) / 64);

  auto stage_all = [&](int qt, char* dst) {
    stage64(qbase, D3, (int64_t)qt * 64, S - 1, 0, dst, wave, lane);
    stage64(dobase, D, (int64_t)qt * 64, S - 1, 0, dst + TILE64, wave, lane);
    stage64(qtbase, Sp, 0, HD - 1, (int64_t)qt * 64, dst + 2 * TILE64, wave, lane);
    stage64(dotbase, Sp, 0, HD - 1, (int64_t)qt * 64, dst + 3 * TILE64, wave, lane);
  };
  if (qt_first <= qt_last) stage_all(qt_first, smem);
  __syncthreads();
  for (int qt = qt_first; qt <= qt_last; ++qt) {
    const int st = (qt - qt_first) & 1;
    const char* cur = smem + st * 4 * TILE64;
    char* nxt = smem + (st ^ 1) * 4 * TILE64;
    if (qt + 1 <= qt_last) stage_all(qt + 1, nxt);
    if ((int64_t)qt * 64 + 63 >= kw0) {  // wave-uniform: some query of this tile sees this wave's keys
      const char* tQ = cur;
      const char* tDO = cur + TILE64;
      const char* tQT = cur + 2 * TILE64;
      const char* tDOT = cur + 3 * TILE64;
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 sacc = zero16(), pacc = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          sacc = mfma32(lds_frag(tQ, qb * 32 + pli, 2 * s + hi), kf[s], sacc);
          pacc = mfma32(lds_frag(tDO, qb * 32 + pli, 2 * s + hi), vf[s], pacc);
        }
        // register r <-> query qbase_r + (r&7), in two runs of 8
        // lse/delta are [B,H,Sp] (Sp = S rounded up to 64): the two runs of 8 are aligned 16-byte loads;
        // entries past S are never-written padding and are discarded by the select below.
        float lsev[16], dlv[16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int64_t qq = bh * Sp + (int64_t)qt * 64 + qb * 32 + 16 * t + 8 * hi;
#pragma unroll
          for (int v4 = 0; v4 < 2; ++v4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(lse + qq + 4 * v4);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta + qq + 4 * v4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              lsev[8 * t + 4 * v4 + e] = a[e] * 1.4426950408889634f;
              dlv[8 * t + 4 * v4 + e] = d4[e];
            }
          }
        }
        f32x16 pv, dsv;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t q = (int64_t)qt * 64 + qb * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
          const bool ok = (q >= krow && q < S);
          const float p = ok ? exp2f(sacc[r] * sc - lsev[r]) : 0.f;
          pv[r] = p;
          dsv[r] = ok ? p * (pacc[r] - dlv[r]) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bf16x8 pf = pack8(pv, 8 * t), dsf = pack8(dsv, 8 * t);
#pragma unroll
          for (int xb = 0; xb < 2; ++xb) {
            dvacc[xb] = mfma32(lds_frag(tDOT, xb * 32 + pli, 4 * qb + 2 * t + hi), pf, dvacc[xb]);
            dkacc[xb] = mfma32(lds_frag(tQT, xb * 32 + pli, 4 * qb + 2 * t + hi), dsf, dkacc[xb]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (krow < S) {
    bf16* krow_out = dqkv + (b * S + krow) * D3 + D + (int64_t)h * HD;
#pragma unroll
    for (int xb = 0; xb < 2; ++xb)
#pragma unroll
      for (int r8 = 0; r8 < 2; ++r8) {
        bf16x8 vk, vv;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          vk[e] = (bf16)(dkacc[xb][8 * r8 + e] * scale);
          vv[e] = (bf16)(dvacc[xb][8 * r8 + e]);
        }
        *reinterpret_cast<bf16x8*>(krow_out + xb * 32 + 16 * r8 + 8 * hi) = vk;
        *reinterpret_cast<bf16x8*>(krow_out + D + xb * 32 + 16 * r8 + 8 * hi) = vv;
      }
  }
}

// ---------------------------------------------------------------------------------------------------
int mh_attn_fwd_mfma(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                     hipStream_t st) {
  MH_REQUIRE(vt != nullptr, "attn_fwd(bf16): needs the transposed V copy (mh_attn_prep_fwd)");
  const int64_t Sp = (S + 63) / 64 * 64;
  dim3 grid((unsigned)((S + 127) / 128), (unsigned)(B * H));
  attn_fwd_kernel<<<grid, 256, 0, st>>>((const bf16*)qkv, (const bf16*)vt, (bf16*)o, lse, S, Sp, H,
                                        scale * 1.4426950408889634f);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

int mh_attn_bwd_mfma(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt,
                     const void* kt, const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale,
                     hipStream_t st) {
  MH_REQUIRE(qt != nullptr && kt != nullptr && dot != nullptr, "attn_bwd(bf16): needs the transposed copies (mh_attn_prep_bwd)");
  const int64_t Sp = (S + 63) / 64 * 64;
  dim3 grid((unsigned)((S + 127) / 128), (unsigned)(B * H));
  attn_bwd_dq_kernel<<<grid, 256, 0, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, (const bf16*)kt,
                                           (bf16*)dqkv, S, Sp, H, scale);
  MH_LAUNCH_CHECK();
  attn_bwd_dkv_kernel<<<grid, 256, 0, st>>>((const bf16*)qkv, (const bf16*)dout, lse, delta, (const bf16*)qt,
                                            (const bf16*)dot, (bf16*)dqkv, S, Sp, H, scale);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
