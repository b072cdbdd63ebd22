// Attention kernels that are NOT matrix-core shaped:
//   * token-level attention of net_token: sequences of T<=8 tokens, head_dim 256 (one wave per head-sequence,
//     everything in registers; HBM-bound: 12 KiB in / 4 KiB out per head-sequence in bf16)
//   * single-query decode attention over a preallocated KV cache (HBM-bound KV stream)
//   * KV-cache append / prefill store
//   * the fp32 verification kernels of event-level attention (thread-per-row, exact fp32 VALU) and the
//     prep kernels (delta = rowsum(dO*O), transposed operand copies for the MFMA flash kernels)
#include "common.h"

#define DISPATCH_T(dtype, CALL)                                    \
  do {                                                             \
    if ((dtype) == MH_BF16) { using T = bf16; CALL; }              \
    else if ((dtype) == MH_F32) { using T = float; CALL; }         \
    else { mh_set_error("bad dtype %d", (int)(dtype)); return MH_ERR_ARG; } \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// token-level attention.  lane l owns dims 4l..4l+3 of the 256-wide head for all T positions.
// ---------------------------------------------------------------------------------------------------
template <typename T> __device__ inline void ld4(const T* p, float (&o)[4]);
template <> __device__ inline void ld4<float>(const float* p, float (&o)[4]) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
template <> __device__ inline void ld4<bf16>(const bf16* p, float (&o)[4]) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}
template <typename T> __device__ inline void st4(T* p, const float (&o)[4]);
template <> __device__ inline void st4<float>(float* p, const float (&o)[4]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{o[0], o[1], o[2], o[3]};
}
template <> __device__ inline void st4<bf16>(bf16* p, const float (&o)[4]) {
  bf16x4 v;
  v[0] = (bf16)o[0]; v[1] = (bf16)o[1]; v[2] = (bf16)o[2]; v[3] = (bf16)o[3];
  *reinterpret_cast<bf16x4*>(p) = v;
}

constexpr int TK = 8;  // max tokens per sequence (the octet)

// Lane layout of a 256-wide head row in the token-level kernels: elements {2l, 2l+1, 128+2l, 128+2l+1} for lane l, i.e.
// the two RoPE partners (i, i + 128) of a pair sit in the same lane, so the rotation needs no cross-lane traffic.
// Every instruction still covers two contiguous 128-byte (bf16) runs of the row.
template <typename T> __device__ inline void ldp(const T* row, int lane, float (&o)[4]);
template <> __device__ inline void ldp<float>(const float* row, int lane, float (&o)[4]) {
  const float2 a = *reinterpret_cast<const float2*>(row + 2 * lane), b = *reinterpret_cast<const float2*>(row + 128 + 2 * lane);
  o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
template <> __device__ inline void ldp<bf16>(const bf16* row, int lane, float (&o)[4]) {
  const bf16x2_t a = *reinterpret_cast<const bf16x2_t*>(row + 2 * lane), b = *reinterpret_cast<const bf16x2_t*>(row + 128 + 2 * lane);
  o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)b[0]; o[3] = (float)b[1];
}
// the same lane share as ldp, kept as loaded (2 dwords for bf16) so that a whole octet's rows can be requested up front
template <typename T> struct RawRow;
template <> struct RawRow<float> { float2 a, b; };
template <> struct RawRow<bf16> { bf16x2_t a, b; };
template <typename T> __device__ inline RawRow<T> ldraw(const T* row, int lane);
template <> __device__ inline RawRow<float> ldraw<float>(const float* row, int lane) {
  return {*reinterpret_cast<const float2*>(row + 2 * lane), *reinterpret_cast<const float2*>(row + 128 + 2 * lane)};
}
template <> __device__ inline RawRow<bf16> ldraw<bf16>(const bf16* row, int lane) {
  return {*reinterpret_cast<const bf16x2_t*>(row + 2 * lane), *reinterpret_cast<const bf16x2_t*>(row + 128 + 2 * lane)};
}
__device__ inline void unpack(const RawRow<float>& r, float (&o)[4]) { o[0] = r.a.x; o[1] = r.a.y; o[2] = r.b.x; o[3] = r.b.y; }
__device__ inline void unpack(const RawRow<bf16>& r, float (&o)[4]) {
  o[0] = (float)r.a[0]; o[1] = (float)r.a[1]; o[2] = (float)r.b[0]; o[3] = (float)r.b[1];
}
template <typename T> __device__ inline void stp(T* row, int lane, const float (&o)[4]);
template <> __device__ inline void stp<float>(float* row, int lane, const float (&o)[4]) {
  *reinterpret_cast<float2*>(row + 2 * lane) = float2{o[0], o[1]};
  *reinterpret_cast<float2*>(row + 128 + 2 * lane) = float2{o[2], o[3]};
}
template <> __device__ inline void stp<bf16>(bf16* row, int lane, const float (&o)[4]) {
  bf16x2_t a, b;
  a[0] = (bf16)o[0]; a[1] = (bf16)o[1]; b[0] = (bf16)o[2]; b[1] = (bf16)o[3];
  *reinterpret_cast<bf16x2_t*>(row + 2 * lane) = a;
  *reinterpret_cast<bf16x2_t*>(row + 128 + 2 * lane) = b;
}
// rotate one lane's share of a row: x' = x c - partner s (first half), x' = x c + partner s (second half); dir = -1
// undoes it (the gradient's way back).  c, s are rounded to T and so is the result when `round` (apply_rotary_pos_emb
// computes in the activation dtype, modeling_llama.py:130-160).
template <typename T>
__device__ inline void rope4(float (&x)[4], const float (&c)[2], const float (&s)[2], float dir, bool round) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const float x1 = x[e], x2 = x[e + 2], sn = dir * s[e];
    const float r1 = x1 * c[e] - x2 * sn, r2 = x2 * c[e] + x1 * sn;
    x[e] = round ? rnd<T>(r1) : r1;
    x[e + 2] = round ? rnd<T>(r2) : r2;
  }
}

template <typename T>
__device__ inline void rope_row(float (&x)[4], const float* rope_c, const float* rope_s, int t, int lane, float dir,
                                bool round) {
  const float rc[2] = {rope_c[t * 128 + 2 * lane], rope_c[t * 128 + 2 * lane + 1]};
  const float rs[2] = {rope_s[t * 128 + 2 * lane], rope_s[t * 128 + 2 * lane + 1]};
  rope4<T>(x, rc, rs, dir, round);
}

// One wave per (sequence, head).  K and V of the <= 8 tokens stay in registers (lane l: 4 elements per row, see ldp);
// the query rows go through one at a time, which keeps the forward at ~125 and the backward at ~160 VGPRs
// (3-4 waves per SIMD in flight instead of 1: these kernels only move bytes).
// TN = the compile-time sequence length (8, the octet of the training path) or 0 for a run-time Tn <= 8.  With TN known
// every row of the (sequence, head) -- K, V, Q (and dO) -- is requested before the first one is used; the run-time form
// branches per row and was a chain of 16 dependent HBM round trips per wave (r01_run20: 604 us forward, 1.3 ms backward
// for 2.1 / 3.8 GB).
template <typename T, int TN>
__global__ __launch_bounds__(256) void tokattn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, int64_t NH, int Tn,
                                                          int H, float scale, const float* __restrict__ cos_t,
                                                          const float* __restrict__ sin_t) {
  const int lane = threadIdx.x & 63;
  const int64_t D = (int64_t)H * 256, D3 = 3 * D;
  // RoPE on load (cos_t != nullptr): q, k come unrotated straight from the projection; position = token index
  __shared__ float rope_c[TK * 128], rope_s[TK * 128];  // rounded to T once
  const bool rot = cos_t != nullptr;
  if (rot) {
    for (int i = threadIdx.x; i < Tn * 128; i += 256) {
      rope_c[i] = rnd<T>(cos_t[i]);
      rope_s[i] = rnd<T>(sin_t[i]);
    }
    __syncthreads();
  }
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < NH; w += (int64_t)gridDim.x * 4) {
    const int64_t n = w / H;
    const int h = (int)(w - n * H);
    const T* base = qkv + n * Tn * D3 + (int64_t)h * 256;
    float k[TK][4], v[TK][4];
    RawRow<T> qraw[TN ? TN : 1];
    if constexpr (TN != 0) {
      RawRow<T> kraw[TN ? TN : 1], vraw[TN ? TN : 1];
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        kraw[t] = ldraw<T>(base + t * D3 + D, lane);
        vraw[t] = ldraw<T>(base + t * D3 + 2 * D, lane);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) qraw[t] = ldraw<T>(base + t * D3, lane);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        unpack(kraw[t], k[t]);
        unpack(vraw[t], v[t]);
        if (rot) rope_row<T>(k[t], rope_c, rope_s, t, lane, 1.f, true);
      }
    } else {
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        if (t < Tn) {
          ldp<T>(base + t * D3 + D, lane, k[t]);
          ldp<T>(base + t * D3 + 2 * D, lane, v[t]);
          if (rot) rope_row<T>(k[t], rope_c, rope_s, t, lane, 1.f, true);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TK; ++i) {
      if (TN ? i >= TN : i >= Tn) break;
      float q[4];
      if constexpr (TN != 0) unpack(qraw[i], q);
      else ldp<T>(base + i * D3, lane, q);
      if (rot) rope_row<T>(q, rope_c, rope_s, i, lane, 1.f, true);
      float s[TK];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        float p = q[0] * k[j][0] + q[1] * k[j][1] + q[2] * k[j][2] + q[3] * k[j][3];
        s[j] = wave_sum_fast(p) * scale;
        mx = fmaxf(mx, s[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        s[j] = __expf(s[j] - mx);
        den += s[j];
      }
      const float inv = 1.f / den;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const float p = s[j] * inv;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += p * v[j][e];
      }
      stp<T>(o + (n * Tn + i) * D + (int64_t)h * 256, lane, acc);
    }
  }
}

// BR (r06): the 2 (i + 1) score / d-probability dot products of query row i are reduced four at a time (wave_sum4) instead of
// one by one -- 20 batched reductions per (sequence, head) instead of 72 single ones.  Same values up to fp32 summation order.
template <typename T, int TN, bool BR>
__global__ __launch_bounds__(256) void tokattn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                          T* __restrict__ dqkv, int64_t NH, int Tn, int H, float scale,
                                                          const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                          const float* __restrict__ rowscale /* != NULL: row m of dqkv times rowscale[m] */) {
  const int lane = threadIdx.x & 63;
  const int64_t D = (int64_t)H * 256, D3 = 3 * D;
  // RoPE on load and its transpose on the way out (cos_t != nullptr): qkv is the unrotated projection output and dqkv
  // the gradient with respect to it
  __shared__ float rope_c[TK * 128], rope_s[TK * 128];
  const bool rot = cos_t != nullptr;
  if (rot) {
    for (int i = threadIdx.x; i < Tn * 128; i += 256) {
      rope_c[i] = rnd<T>(cos_t[i]);
      rope_s[i] = rnd<T>(sin_t[i]);
    }
    __syncthreads();
  }
  for (int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < NH; w += (int64_t)gridDim.x * 4) {
    const int64_t n = w / H;
    const int h = (int)(w - n * H);
    const int64_t off = (int64_t)h * 256;
    const T* base = qkv + n * Tn * D3 + off;
    T* ob = dqkv + n * Tn * D3 + off;
    float k[TK][4], v[TK][4], dk[TK][4], dv[TK][4];
    RawRow<T> qraw[TN ? TN : 1], doraw[TN ? TN : 1];
    // (the sequence's <= 8 row scales in ONE load beside the operand rows -- lane t holds row t's; broadcast below by lane index:
    //  a scalar load per stored row sat on the critical path of this bandwidth-bound kernel, +12 % on the launch)
    const float rs_lane = (rowscale != nullptr && lane < Tn) ? rowscale[n * Tn + lane] : 1.f;
#pragma unroll
    for (int t = 0; t < TK; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) dk[t][e] = dv[t][e] = 0.f;
    if constexpr (TN != 0) {
      RawRow<T> kraw[TN ? TN : 1], vraw[TN ? TN : 1];
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        kraw[t] = ldraw<T>(base + t * D3 + D, lane);
        vraw[t] = ldraw<T>(base + t * D3 + 2 * D, lane);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        qraw[t] = ldraw<T>(base + t * D3, lane);
        doraw[t] = ldraw<T>(dout + (n * Tn + t) * D + off, lane);
      }
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        unpack(kraw[t], k[t]);
        unpack(vraw[t], v[t]);
        if (rot) rope_row<T>(k[t], rope_c, rope_s, t, lane, 1.f, true);
      }
    } else {
#pragma unroll
      for (int t = 0; t < TK; ++t) {
        if (t < Tn) {
          ldp<T>(base + t * D3 + D, lane, k[t]);
          ldp<T>(base + t * D3 + 2 * D, lane, v[t]);
          if (rot) rope_row<T>(k[t], rope_c, rope_s, t, lane, 1.f, true);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TK; ++i) {
      if (TN ? i >= TN : i >= Tn) break;
      float q[4], dO[4], dq[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (TN != 0) {
        unpack(qraw[i], q);
        unpack(doraw[i], dO);
      } else {
        ldp<T>(base + i * D3, lane, q);
        ldp<T>(dout + (n * Tn + i) * D + off, lane, dO);
      }
      if (rot) rope_row<T>(q, rope_c, rope_s, i, lane, 1.f, true);
      float p[TK], dp[TK];
      float mx = -INFINITY;
      if constexpr (BR) {
        // partial dot products of the lane, pairs (j, j + 1): one batched reduction gives p[j], dp[j], p[j + 1], dp[j + 1]
#pragma unroll
        for (int j = 0; j <= i; j += 2) {
          const bool two = j + 1 <= i;
          const int j1 = two ? j + 1 : j;
          const float a0 = q[0] * k[j][0] + q[1] * k[j][1] + q[2] * k[j][2] + q[3] * k[j][3];
          const float b0 = dO[0] * v[j][0] + dO[1] * v[j][1] + dO[2] * v[j][2] + dO[3] * v[j][3];
          const float a1 = two ? q[0] * k[j1][0] + q[1] * k[j1][1] + q[2] * k[j1][2] + q[3] * k[j1][3] : 0.f;
          const float b1 = two ? dO[0] * v[j1][0] + dO[1] * v[j1][1] + dO[2] * v[j1][2] + dO[3] * v[j1][3] : 0.f;
          const float r4 = wave_sum4(a0, b0, a1, b1);
          p[j] = wave_sum4_get<0>(r4) * scale;
          dp[j] = wave_sum4_get<1>(r4);
          mx = fmaxf(mx, p[j]);
          if (two) {
            p[j1] = wave_sum4_get<2>(r4) * scale;
            dp[j1] = wave_sum4_get<3>(r4);
            mx = fmaxf(mx, p[j1]);
          }
        }
      } else {
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        float a = q[0] * k[j][0] + q[1] * k[j][1] + q[2] * k[j][2] + q[3] * k[j][3];
        float b = dO[0] * v[j][0] + dO[1] * v[j][1] + dO[2] * v[j][2] + dO[3] * v[j][3];
        p[j] = wave_sum_fast(a) * scale;
        dp[j] = wave_sum_fast(b);
        mx = fmaxf(mx, p[j]);
      }
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        p[j] = __expf(p[j] - mx);
        den += p[j];
      }
      const float inv = 1.f / den;
      float dsum = 0.f;
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        p[j] *= inv;
        dsum += p[j] * dp[j];
      }
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        const float ds = p[j] * (dp[j] - dsum) * scale;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dq[e] += ds * k[j][e];
          dk[j][e] += ds * q[e];
          dv[j][e] += p[j] * dO[e];
        }
      }
      if (rot) rope_row<T>(dq, rope_c, rope_s, i, lane, -1.f, false);
      if (rowscale != nullptr) {  // (the projection behind a folded RMSNorm takes rstd (.) d qkv: mh_tokattn_bwd_scaled)
        const float rsc = __shfl(rs_lane, i, 64);
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[e] *= rsc;
      }
      stp<T>(ob + i * D3, lane, dq);
    }
#pragma unroll
    for (int t = 0; t < TK; ++t) {
      if (TN ? t < TN : t < Tn) {
        if (rot) rope_row<T>(dk[t], rope_c, rope_s, t, lane, -1.f, false);
        if (rowscale != nullptr) {
          const float rsc = __shfl(rs_lane, t, 64);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dk[t][e] *= rsc;
            dv[t][e] *= rsc;
          }
        }
        stp<T>(ob + t * D3 + D, lane, dk[t]);
        stp<T>(ob + t * D3 + 2 * D, lane, dv[t]);
      }
    }
  }
}

extern "C" int mh_tokattn_fwd(const void* qkv, void* o, int64_t N, int Tn, int H, float scale, const float* cos_t,
                              const float* sin_t, int dtype, void* stream) {
  MH_REQUIRE(N > 0 && Tn >= 1 && Tn <= TK && H >= 1, "tokattn_fwd: bad shape N=%ld T=%d H=%d", (long)N, Tn, H);
  const int64_t NH = N * H;
  int64_t g = (NH + 3) / 4;
  if (g > 32768) g = 32768;
  if (Tn == TK)
    DISPATCH_T(dtype, (tokattn_fwd_kernel<T, TK><<<(int)g, 256, 0, (hipStream_t)stream>>>((const T*)qkv, (T*)o, NH, Tn, H, scale,
                                                                                          cos_t, sin_t)));
  else
    DISPATCH_T(dtype, (tokattn_fwd_kernel<T, 0><<<(int)g, 256, 0, (hipStream_t)stream>>>((const T*)qkv, (T*)o, NH, Tn, H, scale,
                                                                                         cos_t, sin_t)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// mh_set_option("tokattn_bwd_batched", 0 / 1): the octet form of the backward reduces its dot products four at a time (default 1)
thread_local int g_tokattn_bwd_batched = 1;
static int tokattn_bwd_any(const void* qkv, const void* dout, void* dqkv, const float* rowscale, int64_t N, int Tn, int H, float scale,
                           const float* cos_t, const float* sin_t, int dtype, void* stream) {
  MH_REQUIRE(N > 0 && Tn >= 1 && Tn <= TK && H >= 1, "tokattn_bwd: bad shape");
  const int64_t NH = N * H;
  int64_t g = (NH + 3) / 4;
  if (g > 32768) g = 32768;
  if (Tn == TK && g_tokattn_bwd_batched)
    DISPATCH_T(dtype, (tokattn_bwd_kernel<T, TK, true><<<(int)g, 256, 0, (hipStream_t)stream>>>(
                          (const T*)qkv, (const T*)dout, (T*)dqkv, NH, Tn, H, scale, cos_t, sin_t, rowscale)));
  else if (Tn == TK)
    DISPATCH_T(dtype, (tokattn_bwd_kernel<T, TK, false><<<(int)g, 256, 0, (hipStream_t)stream>>>(
                          (const T*)qkv, (const T*)dout, (T*)dqkv, NH, Tn, H, scale, cos_t, sin_t, rowscale)));
  else
    DISPATCH_T(dtype, (tokattn_bwd_kernel<T, 0, false><<<(int)g, 256, 0, (hipStream_t)stream>>>(
                          (const T*)qkv, (const T*)dout, (T*)dqkv, NH, Tn, H, scale, cos_t, sin_t, rowscale)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}
extern "C" int mh_tokattn_bwd(const void* qkv, const void* dout, void* dqkv, int64_t N, int Tn, int H, float scale,
                              const float* cos_t, const float* sin_t, int dtype, void* stream) {
  return tokattn_bwd_any(qkv, dout, dqkv, nullptr, N, Tn, H, scale, cos_t, sin_t, dtype, stream);
}
// mh_tokattn_bwd with row m = n * T + t of dqkv multiplied by rowscale[m] in the stores (the folded RMSNorm's d z)
extern "C" int mh_tokattn_bwd_scaled(const void* qkv, const void* dout, void* dqkv, const float* rowscale, int64_t N, int Tn, int H,
                                     float scale, const float* cos_t, const float* sin_t, int dtype, void* stream) {
  MH_REQUIRE(rowscale != nullptr, "tokattn_bwd_scaled: rowscale is NULL");
  return tokattn_bwd_any(qkv, dout, dqkv, rowscale, N, Tn, H, scale, cos_t, sin_t, dtype, stream);
}

// ---------------------------------------------------------------------------------------------------
// decode: KV cache [B,H,Lmax,HD]; append (with RoPE of q,k) and single-query attention
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(T* __restrict__ qkv, const float* __restrict__ cos_t,
                                                        const float* __restrict__ sin_t, T* __restrict__ kc,
                                                        T* __restrict__ vc, int64_t B, int H, int hd, int64_t Lmax,
                                                        int64_t pos, const int32_t* __restrict__ pos_dev) {
  // one thread per (b, h, pair index i < hd/2): rotate q and k, store k and v rows
  if (pos_dev != nullptr) pos = *pos_dev;  // graph replay: the position lives in device memory
  if (pos >= Lmax) return;
  const int half = hd / 2;
  const int64_t total = B * H * half;
  const int64_t D = (int64_t)H * hd;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int i = (int)(it % half);
    const int64_t bh = it / half;
    const int h = (int)(bh % H);
    const int64_t b = bh / H;
    const float c = rnd<T>(cos_t[pos * half + i]), s = rnd<T>(sin_t[pos * half + i]);
    T* row = qkv + b * 3 * D + (int64_t)h * hd;
    const float q1 = to_f(row[i]), q2 = to_f(row[i + half]);
    row[i] = from_f<T>(q1 * c - q2 * s);
    row[i + half] = from_f<T>(q2 * c + q1 * s);
    const float k1 = to_f(row[D + i]), k2 = to_f(row[D + i + half]);
    const T kr1 = from_f<T>(k1 * c - k2 * s), kr2 = from_f<T>(k2 * c + k1 * s);
    row[D + i] = kr1;
    row[D + i + half] = kr2;
    T* kd = kc + ((b * H + h) * Lmax + pos) * hd;
    T* vd = vc + ((b * H + h) * Lmax + pos) * hd;
    kd[i] = kr1;
    kd[i + half] = kr2;
    vd[i] = row[2 * D + i];
    vd[i + half] = row[2 * D + i + half];
  }
}

extern "C" int mh_kv_append(void* qkv, const float* cos_t, const float* sin_t, void* kcache, void* vcache, int64_t B,
                            int H, int hd, int64_t Lmax, int64_t pos, const int32_t* pos_dev, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && H > 0 && hd % 2 == 0 && ((pos >= 0 && pos < Lmax) || pos_dev != nullptr),
             "kv_append: bad args (pos=%ld Lmax=%ld)", (long)pos, (long)Lmax);
  const int64_t total = B * H * (hd / 2);
  DISPATCH_T(dtype, (kv_append_kernel<T><<<(int)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
                        (T*)qkv, cos_t, sin_t, (T*)kcache, (T*)vcache, B, H, hd, Lmax, pos, pos_dev)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void kv_store_prefill_kernel(const T* __restrict__ qkv, T* __restrict__ kc,
                                                               T* __restrict__ vc, int64_t B, int64_t S, int H, int hd,
                                                               int64_t Lmax, int64_t pos0) {
  constexpr int N = Pack<T>::N;
  const int cph = hd / N;
  const int64_t total = B * S * H * cph;
  const int64_t D = (int64_t)H * hd;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int c = (int)(it % cph);
    int64_t r = it / cph;
    const int h = (int)(r % H);
    r /= H;
    const int64_t s = r % S, b = r / S;
    const T* row = qkv + (b * S + s) * 3 * D + (int64_t)h * hd + c * N;
    const int64_t dst = ((b * H + h) * Lmax + pos0 + s) * hd + c * N;
    st16(kc + dst, ld16(row + D));
    st16(vc + dst, ld16(row + 2 * D));
  }
}

extern "C" int mh_kv_store_rows(const void* qkv, void* kcache, void* vcache, int64_t B, int64_t S, int H, int hd, int64_t Lmax,
                                int64_t pos0, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && pos0 >= 0 && pos0 + S <= Lmax && hd % 8 == 0, "kv_store_rows: bad args");
  const int64_t total = B * S * H * (hd / (dtype == MH_BF16 ? 8 : 4));
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  DISPATCH_T(dtype, (kv_store_prefill_kernel<T><<<(int)g, 256, 0, (hipStream_t)stream>>>((const T*)qkv, (T*)kcache,
                                                                                         (T*)vcache, B, S, H, hd, Lmax, pos0)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_kv_store_prefill(const void* qkv, void* kcache, void* vcache, int64_t B, int64_t S, int H, int hd,
                                   int64_t Lmax, int dtype, void* stream) {
  return mh_kv_store_rows(qkv, kcache, vcache, B, S, H, hd, Lmax, 0, dtype, stream);
}

// cache rows [0, n) of every (sequence, head) back into the K and V columns of rows [b * Stot, b * Stot + n) of a fused
// qkv buffer [B * Stot, 3 H hd] (their q columns are zeroed: those query rows are never used); the inverse of kv_store_rows
template <typename T>
__global__ __launch_bounds__(256) void kv_gather_rows_kernel(const T* __restrict__ kc, const T* __restrict__ vc,
                                                             T* __restrict__ qkv, int64_t B, int64_t n, int64_t Stot, int H, int hd,
                                                             int64_t Lmax) {
  constexpr int N = Pack<T>::N;
  const int cph = hd / N;
  const int64_t total = B * n * H * cph;
  const int64_t D = (int64_t)H * hd;
  Pack<T> z;
#pragma unroll
  for (int i = 0; i < N; ++i) z.set(i, 0.f);
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int c = (int)(it % cph);
    int64_t r = it / cph;
    const int h = (int)(r % H);
    r /= H;
    const int64_t s = r % n, b = r / n;
    T* row = qkv + (b * Stot + s) * 3 * D + (int64_t)h * hd + c * N;
    const int64_t src = ((b * H + h) * Lmax + s) * hd + c * N;
    st16(row, z);
    st16(row + D, ld16(kc + src));
    st16(row + 2 * D, ld16(vc + src));
  }
}

extern "C" int mh_kv_gather_rows(const void* kcache, const void* vcache, void* qkv, int64_t B, int64_t n, int64_t Stot, int H,
                                 int hd, int64_t Lmax, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && n > 0 && n <= Stot && n <= Lmax && hd % 8 == 0, "kv_gather_rows: bad args");
  const int64_t total = B * n * H * (hd / (dtype == MH_BF16 ? 8 : 4));
  int64_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  DISPATCH_T(dtype, (kv_gather_rows_kernel<T><<<(int)g, 256, 0, (hipStream_t)stream>>>((const T*)kcache, (const T*)vcache,
                                                                                       (T*)qkv, B, n, Stot, H, hd, Lmax)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// sum over the aligned group of LPK (8, 16, 32 or 64) consecutive lanes, every lane of the group gets it: DPP moves inside a
// 16-lane row, permlane swaps across rows (common.h, wave_sum_fast) -- the additions a shuffle-xor butterfly performs, without LDS
template <int LPK>
__device__ inline float group_sum(float v) {
  static_assert(LPK == 8 || LPK == 16 || LPK == 32 || LPK == 64, "group_sum: 8, 16, 32 or 64 lanes");
  v += dpp_move<0xB1>(v);   // quad_perm(1,0,3,2)
  v += dpp_move<0x4E>(v);   // quad_perm(2,3,0,1)
  v += dpp_move<0x141>(v);  // row_half_mirror
  if (LPK >= 16) v += dpp_move<0x140>(v);  // row_mirror
  if (LPK >= 32) {
    const int iv = __float_as_int(v);
    auto a = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
    v = __int_as_float(a[0]) + __int_as_float(a[1]);
  }
  if (LPK >= 64) {
    const int iw = __float_as_int(v);
    auto b = __builtin_amdgcn_permlane32_swap(iw, iw, false, false);
    v = __int_as_float(b[0]) + __int_as_float(b[1]);
  }
  return v;
}

// One block (4 waves) per (b,h).  LPK lanes cover one cached key row with 16-byte loads; every lane group
// keeps an online-softmax state over its own subset of keys, merged at the end (groups, then waves).
// APPEND: qkv holds the UNROTATED q,k,v of the new position; the block first rotates k and stores the k,v rows at index
// len-1 of its (b,h) cache (what kv_append_kernel does), rotates q in registers, then attends over rows [0, len).
template <typename T, int HD, bool APPEND>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ qkv, T* kc, T* vc, T* __restrict__ o, int H,
                                                          int64_t Lmax, int64_t len, float scale,
                                                          const int32_t* __restrict__ pos_dev,
                                                          const float* __restrict__ cos_t, const float* __restrict__ sin_t) {
  // every kernel argument fetched at entry in one batch (hipcc sinks each s_load into the block that first uses it otherwise:
  // a scalar-cache miss + wait per block on a kernel whose whole run is a few microseconds)
  asm volatile("" ::"s"(qkv), "s"(kc), "s"(vc), "s"(o), "s"(H), "s"(Lmax), "s"(len), "s"(scale));
  asm volatile("" ::"s"(pos_dev), "s"(cos_t), "s"(sin_t));
  if (pos_dev != nullptr) len = (int64_t)*pos_dev + 1;  // graph replay: attend to rows [0, pos]
  if (len > Lmax) len = Lmax;
  constexpr int N = Pack<T>::N;
  constexpr int LPK = HD / N;       // lanes per key row
  constexpr int KPW = 64 / LPK;     // keys per wave per iteration
  __shared__ float sh_m[4], sh_l[4];
  __shared__ float sh_o[4][HD];
  // APPEND: the new position's rotated k row and v row, as the cache holds them -- the block attends to them from here instead of
  // reading its own stores back through L2 (r04: a write -> read round trip on the critical path of every decode attention launch)
  __shared__ __attribute__((aligned(16))) T sh_kn[APPEND ? HD : 8], sh_vn[APPEND ? HD : 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t bh = blockIdx.x;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * HD;
  const int ch = lane % LPK, grp = lane / LPK;
  const T* qrow = qkv + b * 3 * D + (int64_t)h * HD;
  Pack<T> qv = ld16(qrow + ch * N);
  float q[N];
  if constexpr (APPEND) {
    constexpr int half = HD / 2;
    const int64_t pos = len - 1;
    if (threadIdx.x < half) {  // k row (rotated) and v row of the new position -> cache
      const int i = threadIdx.x;
      const float c = rnd<T>(cos_t[pos * half + i]), sn = rnd<T>(sin_t[pos * half + i]);
      const float k1 = to_f(qrow[D + i]), k2 = to_f(qrow[D + i + half]);
      T* kd = kc + (bh * Lmax + pos) * HD;
      T* vd = vc + (bh * Lmax + pos) * HD;
      const T ka = from_f<T>(k1 * c - k2 * sn), kb2 = from_f<T>(k2 * c + k1 * sn);
      const T va = qrow[2 * D + i], vb2 = qrow[2 * D + i + half];
      kd[i] = ka;
      kd[i + half] = kb2;
      vd[i] = va;
      vd[i + half] = vb2;
      sh_kn[i] = ka;
      sh_kn[i + half] = kb2;
      sh_vn[i] = va;
      sh_vn[i + half] = vb2;
    }
    const int base = ch * N;  // this lane's q elements and their rotation partners (i, i + half)
    Pack<T> qp = ld16(qrow + (base + half) % HD);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const int i = (base + e) % half;
      const float c = rnd<T>(cos_t[pos * half + i]), sn = rnd<T>(sin_t[pos * half + i]);
      const float r = (base < half) ? qv.get(e) * c - qp.get(e) * sn : qv.get(e) * c + qp.get(e) * sn;
      q[e] = rnd<T>(r) * scale;
    }
    __syncthreads();  // the new k,v rows are visible to the whole block
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) q[e] = qv.get(e) * scale;
  }
  float m = -INFINITY, l = 0.f, acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0.f;
  const T* kb = kc + bh * Lmax * HD;
  const T* vb = vc + bh * Lmax * HD;
  // four steps of the key loop at a time with all K / V rows requested first: one step per memory round trip was one
  // round trip per 32 keys of head_dim 64 (13 us per layer at 128 cached events, r02 trace).  r03: the NEXT four steps are
  // requested before the current four are reduced (two register sets, 16 row loads per lane in flight): at batch 64 x 1024
  // events the kernel moved 1.6 GB per generated event at 4.6 TB/s with the loads of an iteration waiting behind its
  // arithmetic (profiles/r03_run1_*); and the per-key sum over the LPK lanes of a row runs on DPP / permlane moves (group_sum)
  // instead of three to five dependent ds_bpermute round trips (same additions in the same order: identical bits).
  constexpr int UNR = 4;
  constexpr int STEP = 4 * KPW * UNR;
  const int ilen = APPEND ? (int)len - 1 : (int)len;  // (<= Lmax < 2^24: 32-bit key indices and element offsets from the uniform bases;
                                                      //  APPEND: the rows read from the cache; the new one comes from LDS below)
  const int lane_off = ch * N;
  Pack<T> kvA[UNR], vvA[UNR], kvB[UNR], vvB[UNR];
  bool okA[UNR], okB[UNR];
  auto request = [&](Pack<T> (&kv)[UNR], Pack<T> (&vv)[UNR], bool (&ok)[UNR], int j0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int j = j0 + u * 4 * KPW + grp;
      ok[u] = j < ilen;
      const unsigned off = (unsigned)((ok[u] ? j : 0) * HD + lane_off);
      kv[u] = ld16(kb + off);
      vv[u] = ld16(vb + off);
    }
  };
  auto reduce = [&](const Pack<T> (&kv)[UNR], const Pack<T> (&vv)[UNR], const bool (&ok)[UNR]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < N; ++e) s += q[e] * kv[u].get(e);
      s = group_sum<LPK>(s);
      if (ok[u]) {
        const float nm = fmaxf(m, s);
        const float a = __expf(m - nm), p = __expf(s - nm);
        l = l * a + p;
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = acc[e] * a + p * vv[u].get(e);
        m = nm;
      }
    }
  };
  int j0 = wv * KPW;
  if (j0 < ilen) {
    request(kvA, vvA, okA, j0);
    while (true) {
      const int j1 = j0 + STEP;
      if (j1 < ilen) request(kvB, vvB, okB, j1);
      // (the requests above are issued before the first dot product below: without the fence hipcc starts the arithmetic of
      //  the first row right behind its load and waits vmcnt(0) for it BEFORE issuing the others -- r02 ISA)
      __builtin_amdgcn_sched_barrier(0);
      reduce(kvA, vvA, okA);
      if (j1 >= ilen) break;
      j0 = j1 + STEP;
      if (j0 < ilen) request(kvA, vvA, okA, j0);
      __builtin_amdgcn_sched_barrier(0);
      reduce(kvB, vvB, okB);
      if (j0 >= ilen) break;
    }
  }
  if constexpr (APPEND) {  // the new position's key / value: lane group 0 of wave 0
    if (wv == 0) {
      const Pack<T> kn = ld16(sh_kn + lane_off), vn = ld16(sh_vn + lane_off);
      float sdot = 0.f;
#pragma unroll
      for (int e = 0; e < N; ++e) sdot += q[e] * kn.get(e);
      sdot = group_sum<LPK>(sdot);
      if (grp == 0) {
        const float nm = fmaxf(m, sdot);
        const float a = __expf(m - nm), p = __expf(sdot - nm);
        l = l * a + p;
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = acc[e] * a + p * vn.get(e);
        m = nm;
      }
    }
  }
  // merge the KPW lane groups of the wave (lanes with equal `ch`)
#pragma unroll
  for (int x = LPK; x < 64; x <<= 1) {
    const float om = __shfl_xor(m, x, 64), ol = __shfl_xor(l, x, 64);
    const float nm = fmaxf(m, om);
    const float a = (m == -INFINITY) ? 0.f : __expf(m - nm);
    const float bsc = (om == -INFINITY) ? 0.f : __expf(om - nm);
    l = l * a + ol * bsc;
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = acc[e] * a + __shfl_xor(acc[e], x, 64) * bsc;
    m = nm;
  }
  if (lane < LPK) {
#pragma unroll
    for (int e = 0; e < N; ++e) sh_o[wv][ch * N + e] = acc[e];
    if (lane == 0) {
      sh_m[wv] = m;
      sh_l[wv] = l;
    }
  }
  __syncthreads();
  if (threadIdx.x < HD) {
    const float gm = fmaxf(fmaxf(sh_m[0], sh_m[1]), fmaxf(sh_m[2], sh_m[3]));
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float f = (sh_m[w] == -INFINITY) ? 0.f : __expf(sh_m[w] - gm);
      num += sh_o[w][threadIdx.x] * f;
      den += sh_l[w] * f;
    }
    o[b * D + (int64_t)h * HD + threadIdx.x] = from_f<T>(num / den);
  }
}

static int attn_decode_launch(const void* qkv, const float* cos_t, const float* sin_t, void* kcache, void* vcache, void* o,
                              int64_t B, int H, int hd, int64_t Lmax, int64_t len, float scale, const int32_t* pos_dev,
                              int dtype, void* stream) {
  MH_REQUIRE(B > 0 && H > 0 && ((len > 0 && len <= Lmax) || pos_dev != nullptr), "attn_decode: bad args len=%ld Lmax=%ld",
             (long)len, (long)Lmax);
  MH_REQUIRE(hd == 64 || hd == 256, "attn_decode: head_dim %d unsupported (64 or 256)", hd);
  hipStream_t st = (hipStream_t)stream;
  const int grid = (int)(B * H);
  const bool app = cos_t != nullptr;
#define LAUNCH_DEC(TT, HDV)                                                                                               \
  do {                                                                                                                    \
    if (app)                                                                                                              \
      attn_decode_kernel<TT, HDV, true><<<grid, 256, 0, st>>>((const TT*)qkv, (TT*)kcache, (TT*)vcache, (TT*)o, H, Lmax,    \
                                                              len, scale, pos_dev, cos_t, sin_t);                         \
    else                                                                                                                  \
      attn_decode_kernel<TT, HDV, false><<<grid, 256, 0, st>>>((const TT*)qkv, (TT*)kcache, (TT*)vcache, (TT*)o, H, Lmax,   \
                                                               len, scale, pos_dev, cos_t, sin_t);                        \
  } while (0)
  if (dtype == MH_BF16) {
    if (hd == 64) LAUNCH_DEC(bf16, 64); else LAUNCH_DEC(bf16, 256);
  } else if (dtype == MH_F32) {
    if (hd == 64) LAUNCH_DEC(float, 64); else LAUNCH_DEC(float, 256);
  } else {
    mh_set_error("attn_decode: bad dtype");
    return MH_ERR_ARG;
  }
#undef LAUNCH_DEC
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_attn_decode(const void* qkv, const void* kcache, const void* vcache, void* o, int64_t B, int H, int hd,
                              int64_t Lmax, int64_t len, float scale, const int32_t* pos_dev, int dtype, void* stream) {
  return attn_decode_launch(qkv, nullptr, nullptr, (void*)kcache, (void*)vcache, o, B, H, hd, Lmax, len, scale, pos_dev, dtype,
                            stream);
}

extern "C" int mh_attn_decode_append(const void* qkv, const float* cos_t, const float* sin_t, void* kcache, void* vcache,
                                     void* o, int64_t B, int H, int hd, int64_t Lmax, int64_t pos, float scale,
                                     const int32_t* pos_dev, int dtype, void* stream) {
  MH_REQUIRE(cos_t != nullptr && sin_t != nullptr, "attn_decode_append: needs the rope tables");
  return attn_decode_launch(qkv, cos_t, sin_t, kcache, vcache, o, B, H, hd, Lmax, pos + 1, scale, pos_dev, dtype, stream);
}

// ---------------------------------------------------------------------------------------------------
// event-level attention, head_dim 64: plain verification kernels (thread per row, fp32 math).
// Used for dtype fp32 (parity mode) — the bf16 production path is attention_mfma.hip.
// ---------------------------------------------------------------------------------------------------
constexpr int AHD = 64;

template <typename T>
__global__ __launch_bounds__(64) void attn_plain_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o,
                                                            float* __restrict__ lse, int64_t S, int H, float scale,
                                                            int first_block /* query blocks below it are not computed */) {
  __shared__ float ks[32][AHD], vs[32][AHD];
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * AHD, D3 = 3 * D;
  const int64_t q0 = (int64_t)(blockIdx.x + first_block) * 64;
  const int64_t qi = q0 + threadIdx.x;
  const bool valid = qi < S;
  float q[AHD], acc[AHD];
  const T* qrow = qkv + (b * S + (valid ? qi : 0)) * D3 + (int64_t)h * AHD;
#pragma unroll
  for (int d = 0; d < AHD; ++d) {
    q[d] = to_f(qrow[d]) * scale;
    acc[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const int64_t kmax = (q0 + 63 < S - 1) ? q0 + 63 : S - 1;  // last key any row of this block can see
  for (int64_t k0 = 0; k0 <= kmax; k0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * AHD; i += 64) {
      const int r = i / AHD, d = i % AHD;
      const int64_t kj = k0 + r;
      const T* krow = qkv + (b * S + (kj < S ? kj : S - 1)) * D3 + D + (int64_t)h * AHD;
      ks[r][d] = to_f(krow[d]);
      vs[r][d] = to_f(krow[D + d]);
    }
    __syncthreads();
    if (!valid) continue;
    for (int r = 0; r < 32; ++r) {
      const int64_t kj = k0 + r;
      if (kj > qi) break;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < AHD; ++d) s += q[d] * ks[r][d];
      const float nm = fmaxf(m, s);
      const float a = __expf(m - nm), p = __expf(s - nm);
      l = l * a + p;
#pragma unroll
      for (int d = 0; d < AHD; ++d) acc[d] = acc[d] * a + p * vs[r][d];
      m = nm;
    }
  }
  if (valid) {
    T* orow = o + (b * S + qi) * D + (int64_t)h * AHD;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < AHD; ++d) orow[d] = from_f<T>(acc[d] * inv);
    lse[bh * ((S + 63) / 64 * 64) + qi] = m + __logf(l);
  }
}

// dQ: thread per query row
template <typename T>
__global__ __launch_bounds__(64) void attn_plain_bwd_dq_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               T* __restrict__ dqkv, int64_t S, int H, float scale) {
  __shared__ float ks[32][AHD], vs[32][AHD];
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * AHD, D3 = 3 * D;
  const int64_t q0 = (int64_t)blockIdx.x * 64;
  const int64_t qi = q0 + threadIdx.x;
  const bool valid = qi < S;
  float q[AHD], dO[AHD], dq[AHD];
  const int64_t qc = valid ? qi : 0;
  const T* qrow = qkv + (b * S + qc) * D3 + (int64_t)h * AHD;
  const T* drow = dout + (b * S + qc) * D + (int64_t)h * AHD;
#pragma unroll
  for (int d = 0; d < AHD; ++d) {
    q[d] = to_f(qrow[d]);
    dO[d] = to_f(drow[d]);
    dq[d] = 0.f;
  }
  const int64_t Sp = (S + 63) / 64 * 64;
  const float L = lse[bh * Sp + qc], dl = delta[bh * Sp + qc];
  const int64_t kmax = (q0 + 63 < S - 1) ? q0 + 63 : S - 1;
  for (int64_t k0 = 0; k0 <= kmax; k0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * AHD; i += 64) {
      const int r = i / AHD, d = i % AHD;
      const int64_t kj = k0 + r;
      const T* krow = qkv + (b * S + (kj < S ? kj : S - 1)) * D3 + D + (int64_t)h * AHD;
      ks[r][d] = to_f(krow[d]);
      vs[r][d] = to_f(krow[D + d]);
    }
    __syncthreads();
    if (!valid) continue;
    for (int r = 0; r < 32; ++r) {
      if (k0 + r > qi) break;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < AHD; ++d) {
        s += q[d] * ks[r][d];
        dp += dO[d] * vs[r][d];
      }
      const float p = __expf(s * scale - L);
      const float ds = p * (dp - dl) * scale;
#pragma unroll
      for (int d = 0; d < AHD; ++d) dq[d] += ds * ks[r][d];
    }
  }
  if (valid) {
    T* orow = dqkv + (b * S + qi) * D3 + (int64_t)h * AHD;
#pragma unroll
    for (int d = 0; d < AHD; ++d) orow[d] = from_f<T>(dq[d]);
  }
}

// dK, dV: thread per key row, loops over the queries that see it
template <typename T>
__global__ __launch_bounds__(64) void attn_plain_bwd_dkv_kernel(const T* __restrict__ qkv, const T* __restrict__ dout,
                                                                const float* __restrict__ lse, const float* __restrict__ delta,
                                                                T* __restrict__ dqkv, int64_t S, int H, float scale) {
  __shared__ float qs[32][AHD], ds_[32][AHD];
  __shared__ float ls[32], dls[32];
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t D = (int64_t)H * AHD, D3 = 3 * D;
  const int64_t k0 = (int64_t)blockIdx.x * 64;
  const int64_t kj = k0 + threadIdx.x;
  const bool valid = kj < S;
  float k[AHD], v[AHD], dk[AHD], dv[AHD];
  const T* krow = qkv + (b * S + (valid ? kj : 0)) * D3 + D + (int64_t)h * AHD;
#pragma unroll
  for (int d = 0; d < AHD; ++d) {
    k[d] = to_f(krow[d]);
    v[d] = to_f(krow[D + d]);
    dk[d] = dv[d] = 0.f;
  }
  for (int64_t q0 = k0 / 32 * 32; q0 < S; q0 += 32) {
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * AHD; i += 64) {
      const int r = i / AHD, d = i % AHD;
      const int64_t qi = (q0 + r < S) ? q0 + r : S - 1;
      qs[r][d] = to_f(qkv[(b * S + qi) * D3 + (int64_t)h * AHD + d]);
      ds_[r][d] = to_f(dout[(b * S + qi) * D + (int64_t)h * AHD + d]);
    }
    if (threadIdx.x < 32) {
      const int64_t qi = (q0 + threadIdx.x < S) ? q0 + threadIdx.x : S - 1;
      ls[threadIdx.x] = lse[bh * ((S + 63) / 64 * 64) + qi];
      dls[threadIdx.x] = delta[bh * ((S + 63) / 64 * 64) + qi];
    }
    __syncthreads();
    if (!valid) continue;
    for (int r = 0; r < 32; ++r) {
      const int64_t qi = q0 + r;
      if (qi >= S) break;
      if (qi < kj) continue;
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < AHD; ++d) {
        s += qs[r][d] * k[d];
        dp += ds_[r][d] * v[d];
      }
      const float p = __expf(s * scale - ls[r]);
      const float dsv = p * (dp - dls[r]) * scale;
#pragma unroll
      for (int d = 0; d < AHD; ++d) {
        dv[d] += p * ds_[r][d];
        dk[d] += dsv * qs[r][d];
      }
    }
  }
  if (valid) {
    T* orow = dqkv + (b * S + kj) * D3 + D + (int64_t)h * AHD;
#pragma unroll
    for (int d = 0; d < AHD; ++d) {
      orow[d] = from_f<T>(dk[d]);
      orow[D + d] = from_f<T>(dv[d]);
    }
  }
}

// delta[b,h,s] = sum_d dO*O ; optionally the transposed copies X^T[b,h,d,s] (s contiguous, ld = Sp)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, int64_t B, int64_t S, int H) {
  // 8 lanes per (row, head): 16-byte chunks of bf16 (or two of fp32)
  const int64_t D = (int64_t)H * AHD;
  const int64_t total = B * S * H;
  const int sub = threadIdx.x & 7;
  for (int64_t it = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 3; it < total; it += ((int64_t)gridDim.x * 256) >> 3) {
    const int h = (int)(it % H);
    const int64_t row = it / H;  // b*S + s
    const T* po = o + row * D + (int64_t)h * AHD + sub * 8;
    const T* pd = dout + row * D + (int64_t)h * AHD + sub * 8;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += to_f(po[e]) * to_f(pd[e]);
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) {
      const int64_t b = row / S, sidx = row - b * S;
      delta[(b * H + h) * ((S + 63) / 64 * 64) + sidx] = s;
    }
  }
}

// out[b,h,d,s] = in[(b*S+s)*ld + col0 + h*64 + d]   (64x64 tiles through LDS)
template <typename T>
__global__ __launch_bounds__(256) void attn_transpose_kernel(const T* __restrict__ in, int64_t ld, int64_t col0,
                                                             T* __restrict__ out, int64_t S, int64_t Sp, int H) {
  __shared__ T tile[64][64 + 4 / sizeof(T)];
  const int64_t bh = blockIdx.y;
  const int64_t b = bh / H;
  const int h = (int)(bh - b * H);
  const int64_t s0 = (int64_t)blockIdx.x * 64;
  constexpr int N = Pack<T>::N, CPR = 64 / N, RPP = 256 / CPR;
  const int ch = threadIdx.x % CPR, rr = threadIdx.x / CPR;
  for (int p = 0; p < 64 / RPP; ++p) {
    const int i = rr + p * RPP;
    const int64_t s = s0 + i;
    Pack<T> v;
    if (s < S) {
      v = ld16(in + (b * S + s) * ld + col0 + (int64_t)h * 64 + ch * N);
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) v.set(e, 0.f);
    }
#pragma unroll
    for (int e = 0; e < N; ++e) tile[i][ch * N + e] = v.v[e];
  }
  __syncthreads();
  for (int p = 0; p < 64 / RPP; ++p) {
    const int d = rr + p * RPP;
    Pack<T> v;
#pragma unroll
    for (int e = 0; e < N; ++e) v.v[e] = tile[ch * N + e][d];
    st16(out + (bh * 64 + d) * Sp + s0 + ch * N, v);
  }
}

template <typename T>
static int transpose_heads(const T* in, int64_t ld, int64_t col0, T* out, int64_t B, int64_t S, int H, hipStream_t st) {
  const int64_t Sp = (S + 63) / 64 * 64;
  dim3 grid((unsigned)(Sp / 64), (unsigned)(B * H));
  attn_transpose_kernel<T><<<grid, 256, 0, st>>>(in, ld, col0, out, S, Sp, H);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_attn_prep_fwd(const void* qkv, void* vt, int64_t B, int64_t S, int H, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_prep_fwd: bad shape");
  if (dtype != MH_BF16) return MH_OK;  // the fp32 verification kernels read qkv directly
  const int64_t D = (int64_t)H * 64;
  return transpose_heads<bf16>((const bf16*)qkv, 3 * D, 2 * D, (bf16*)vt, B, S, H, (hipStream_t)stream);
}

extern "C" int mh_attn_prep_bwd(const void* qkv, const void* o, const void* dout, float* delta, void* qt, void* kt,
                                void* dot, int64_t B, int64_t S, int H, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_prep_bwd: bad shape");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = B * S * H;
  int64_t g = (total * 8 + 255) / 256;
  if (g > 16384) g = 16384;
  DISPATCH_T(dtype, (attn_delta_kernel<T><<<(int)g, 256, 0, st>>>((const T*)o, (const T*)dout, delta, B, S, H)));
  MH_LAUNCH_CHECK();
  if (dtype != MH_BF16 || qt == nullptr) return MH_OK;  // (no buffers: the backward reads its transposed operands itself)
  const int64_t D = (int64_t)H * 64;
  int rc;
  if ((rc = transpose_heads<bf16>((const bf16*)qkv, 3 * D, 0, (bf16*)qt, B, S, H, st)) != MH_OK) return rc;
  if ((rc = transpose_heads<bf16>((const bf16*)qkv, 3 * D, D, (bf16*)kt, B, S, H, st)) != MH_OK) return rc;
  return transpose_heads<bf16>((const bf16*)dout, D, 0, (bf16*)dot, B, S, H, st);
}

// entry points shared with attention_mfma.hip (bf16 goes there)
int mh_attn_fwd_mfma(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                     hipStream_t st, int64_t q_start);
int mh_attn_bwd_mfma(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt,
                     const void* kt, const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale,
                     const float* cos_t, const float* sin_t, hipStream_t st);

int mh_attn_bwd_o_mfma(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int64_t B,
                       int64_t S, int H, float scale, const float* cos_t, const float* sin_t, hipStream_t st);

template <typename T>
static int attn_plain_fwd(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, hipStream_t st,
                          int64_t q_start = 0) {
  const int first = (int)(q_start / 64);
  dim3 grid((unsigned)((S + 63) / 64 - first), (unsigned)(B * H));
  attn_plain_fwd_kernel<T><<<grid, 64, 0, st>>>((const T*)qkv, (T*)o, lse, S, H, scale, first);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

template <typename T>
static int attn_plain_bwd(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv, int64_t B,
                          int64_t S, int H, float scale, hipStream_t st) {
  dim3 grid((unsigned)((S + 63) / 64), (unsigned)(B * H));
  attn_plain_bwd_dq_kernel<T><<<grid, 64, 0, st>>>((const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, S, H, scale);
  MH_LAUNCH_CHECK();
  attn_plain_bwd_dkv_kernel<T><<<grid, 64, 0, st>>>((const T*)qkv, (const T*)dout, lse, delta, (T*)dqkv, S, H, scale);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// `mh_attn_*_plain` force the thread-per-row kernels for either dtype (used by tests to cross-check the
// MFMA kernels on the device itself).
extern "C" int mh_attn_fwd_plain(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                                 int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_fwd: bad shape");
  DISPATCH_T(dtype, return attn_plain_fwd<T>(qkv, o, lse, B, S, H, scale, (hipStream_t)stream));
}

extern "C" int mh_attn_bwd_plain(const void* qkv, const void* dout, const float* lse, const float* delta, void* dqkv,
                                 int64_t B, int64_t S, int H, float scale, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_bwd: bad shape");
  DISPATCH_T(dtype, return attn_plain_bwd<T>(qkv, dout, lse, delta, dqkv, B, S, H, scale, (hipStream_t)stream));
}

extern "C" int mh_attn_fwd(const void* qkv, const void* vt, void* o, float* lse, int64_t B, int64_t S, int H, float scale,
                           int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_fwd: bad shape");
  if (dtype == MH_BF16) return mh_attn_fwd_mfma(qkv, vt, o, lse, B, S, H, scale, (hipStream_t)stream, 0);
  if (dtype == MH_F32) return attn_plain_fwd<float>(qkv, o, lse, B, S, H, scale, (hipStream_t)stream);
  mh_set_error("attn_fwd: bad dtype");
  return MH_ERR_ARG;
}

// A chunk of new positions behind cached ones (a cache-carrying forward with q_len > 1, TF:integrations/sdpa_attention.py:79-166
// with a non-empty DynamicCache): the rows of qkv hold the whole sequence so far -- K, V of the cached positions gathered back
// from the cache (mh_kv_gather_rows), q | k | v of the new ones -- and only the query rows >= q_start are computed (whole
// 128-row / 64-row query tiles: rows of the first tile below q_start are computed too and ignored by the caller).
extern "C" int mh_attn_fwd_tail(const void* qkv, void* o, float* lse, int64_t B, int64_t S, int H, float scale, int64_t q_start,
                                int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536 && q_start >= 0 && q_start < S, "attn_fwd_tail: bad shape");
  if (dtype == MH_BF16) return mh_attn_fwd_mfma(qkv, nullptr, o, lse, B, S, H, scale, (hipStream_t)stream, q_start);
  if (dtype == MH_F32) return attn_plain_fwd<float>(qkv, o, lse, B, S, H, scale, (hipStream_t)stream, q_start);
  mh_set_error("attn_fwd_tail: bad dtype");
  return MH_ERR_ARG;
}

extern "C" int mh_rope(void* qkv, const float* cos_t, const float* sin_t, int64_t M, int64_t S, int64_t pos0, int H, int hd,
                       int dir, int dtype, void* stream);  // elementwise.hip

extern "C" int mh_attn_bwd(const void* qkv, const void* dout, const float* lse, const float* delta, const void* qt,
                           const void* kt, const void* dot, void* dqkv, int64_t B, int64_t S, int H, float scale,
                           const float* cos_t, const float* sin_t, int dtype, void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_bwd: bad shape");
  MH_REQUIRE((cos_t == nullptr) == (sin_t == nullptr), "attn_bwd: cos and sin tables come together");
  if (dtype == MH_BF16)  // (the rotation back rides on the dq / dk stores)
    return mh_attn_bwd_mfma(qkv, dout, lse, delta, qt, kt, dot, dqkv, B, S, H, scale, cos_t, sin_t, (hipStream_t)stream);
  if (dtype == MH_F32) {
    int rc = attn_plain_bwd<float>(qkv, dout, lse, delta, dqkv, B, S, H, scale, (hipStream_t)stream);
    if (rc == MH_OK && cos_t != nullptr) rc = mh_rope(dqkv, cos_t, sin_t, B * S, S, 0, H, 64, -1, dtype, stream);
    return rc;
  }
  mh_set_error("attn_bwd: bad dtype");
  return MH_ERR_ARG;
}

extern thread_local const float* g_attn_bwd_rowscale;  // attention_mfma3.hip
extern thread_local int g_attn_v3;
// mh_attn_bwd_o with every row m of dqkv multiplied by rowscale[m] (m = b * S + position) in the kernels' stores: the gradient
// handed to a projection that sits behind a folded RMSNorm (engine.layer_backward_folded).  Third-form kernels only.
extern "C" int mh_attn_bwd_o_scaled(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                                    const float* rowscale, int64_t B, int64_t S, int H, float scale, const float* cos_t,
                                    const float* sin_t, int dtype, void* stream) {
  MH_REQUIRE(rowscale != nullptr, "attn_bwd_o_scaled: rowscale is NULL");
  MH_REQUIRE((g_attn_v3 & 46) == 46, "attn_bwd_o_scaled: needs the third form of the backward kernels (attn_v3 bits 1, 2, 3, 5)");
  g_attn_bwd_rowscale = rowscale;
  const int rc = mh_attn_bwd_o(qkv, o, dout, lse, delta, dqkv, B, S, H, scale, cos_t, sin_t, dtype, stream);
  g_attn_bwd_rowscale = nullptr;
  return rc;
}

extern "C" int mh_attn_bwd_o(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv,
                             int64_t B, int64_t S, int H, float scale, const float* cos_t, const float* sin_t, int dtype,
                             void* stream) {
  MH_REQUIRE(B > 0 && S > 0 && H > 0 && B * H < 65536, "attn_bwd_o: bad shape");
  MH_REQUIRE((cos_t == nullptr) == (sin_t == nullptr), "attn_bwd_o: cos and sin tables come together");
  MH_REQUIRE(dtype == MH_BF16, "attn_bwd_o: bf16 only (fp32: mh_attn_prep_bwd + mh_attn_bwd)");
  return mh_attn_bwd_o_mfma(qkv, o, dout, lse, delta, dqkv, B, S, H, scale, cos_t, sin_t, (hipStream_t)stream);
}
