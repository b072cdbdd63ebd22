// Cross-entropy over the vocabulary, gradient-norm clip + AdamW, and the grammar-masked softmax of the
// sampler.  All HBM-bound row / flat-buffer streamers; reductions are deterministic (no float atomics).
#include "common.h"

#define DISPATCH_T(dtype, CALL)                                    \
  do {                                                             \
    if ((dtype) == MH_BF16) { using T = bf16; CALL; }              \
    else if ((dtype) == MH_F32) { using T = float; CALL; }         \
    else { mh_set_error("bad dtype %d", (int)(dtype)); return MH_ERR_ARG; } \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// cross entropy: one wave per row.  pass 1: online (max, sum exp, argmax, target logit); pass 2: gradient.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* logits, int64_t ldl, const int64_t* __restrict__ target,
                                                            float* __restrict__ row_loss, T* dlogits,
                                                            const float* __restrict__ scale_dev,
                                                            int64_t* __restrict__ argmax_out, int64_t R, int V,
                                                            int64_t ignore) {
  const int lane = threadIdx.x & 63;
  const float scale = (dlogits != nullptr && scale_dev != nullptr) ? scale_dev[0] : 1.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += (int64_t)gridDim.x * 4) {
    const T* row = logits + r * ldl;
    float mx = -INFINITY, sum = 0.f;
    int amax = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
      const float v = to_f(row[c]);
      if (v > mx) {
        sum = sum * __expf(mx - v) + 1.f;
        mx = v;
        amax = c;
      } else {
        sum += __expf(v - mx);
      }
    }
    // combine the 64 (max,sum,argmax) triples
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float omx = __shfl_xor(mx, o, 64), osum = __shfl_xor(sum, o, 64);
      const int oam = __shfl_xor(amax, o, 64);
      const float nm = fmaxf(mx, omx);
      const float s1 = (mx == -INFINITY) ? 0.f : sum * __expf(mx - nm);
      const float s2 = (omx == -INFINITY) ? 0.f : osum * __expf(omx - nm);
      if (omx > mx || (omx == mx && oam < amax)) amax = oam;
      mx = nm;
      sum = s1 + s2;
    }
    const float lse = mx + __logf(sum);
    const int64_t tgt = target[r];
    const bool keep = (tgt != ignore);
    if (lane == 0) {
      row_loss[r] = keep ? (lse - to_f(row[tgt])) : 0.f;
      if (argmax_out != nullptr) argmax_out[r] = amax;
    }
    if (dlogits != nullptr) {
      T* drow = dlogits + r * ldl;
      for (int c = lane; c < (int)ldl; c += 64) {
        float g = 0.f;
        if (keep && c < V) {
          g = __expf(to_f(row[c]) - lse);
          if (c == (int)tgt) g -= 1.f;
          g *= scale;
        }
        drow[c] = from_f<T>(g);
      }
    }
  }
}

// bf16 rows of 16-byte-aligned stride up to NCH * 512 elements: the row is loaded once with 16-byte loads and stays in
// registers for both passes (statistics, gradient); same arithmetic as cross_entropy_kernel up to summation order.
// r06: the kernel was bound by its VALU, not by HBM (3.3 TB/s; ~40 lane-cycles per element: three bf16 -> fp32 conversions,
// a `c < V` compare + select per element and pass, the target compare + select per gradient element, the arg-max bookkeeping
// nobody reads in a training step).  Now the row is converted ONCE into fp32 registers with its padding columns set to -inf
// (exp2 of -inf is +0: every later pass is unconditional -- the sum gains exact zeros, the gradient of the padding is +0 as
// before), the one chunk that can hold padding and the one chunk that holds the target are found with wave-uniform tests, an
// ignored row stores zeros without computing anything, and the arg-max bookkeeping is its own instantiation (ARG, validation).
// Per element: 1 convert, 1 max, sub-mul-exp-add, sub-mul-exp-mul-convert.  Results are bit-identical to the previous form
// (same operations on the same values in the same order; profiles/r06_ce_ab.txt).
template <int NCH, bool ARG>
__global__ __launch_bounds__(256) void cross_entropy_vec_kernel(const bf16* logits, int64_t ldl, const int64_t* __restrict__ target,
                                                                float* __restrict__ row_loss, bf16* dlogits,
                                                                const float* __restrict__ scale_dev,
                                                                int64_t* __restrict__ argmax_out, int64_t R, int V,
                                                                int64_t ignore) {
  constexpr float LOG2E_F = 1.44269502162933349609375f;  // 0x3fb8aa3b: the constant __expf multiplies by
  const int lane = threadIdx.x & 63;
  const float scale = (dlogits != nullptr && scale_dev != nullptr) ? scale_dev[0] : 1.f;
  const int nchunk = (int)(ldl / 8);
  // the next row of the wave is requested before the current one is worked on (two rows of loads in flight per wave)
  auto load_row = [&](int64_t r, bf16x8 (&dst)[NCH]) {
    const bf16* row = logits + r * ldl;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {  // (chunks past the row re-read chunk 0: unconditional loads, their columns are >= V)
      const int ch = k * 64 + lane;
      dst[k] = *reinterpret_cast<const bf16x8*>(row + (ch < nchunk ? ch : 0) * 8);
    }
  };
  const int64_t rstep = (int64_t)gridDim.x * 4;
  int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  bf16x8 v[NCH], vn[NCH];
  if (r < R) load_row(r, v);
  for (; r < R; r += rstep) {
    const bf16* row = logits + r * ldl;
    if (r + rstep < R) load_row(r + rstep, vn);
    // (wave-uniform by construction -- one row per wave; the readfirstlane makes the tests below scalar branches)
    const int64_t tgt64 = target[r];
    const int tgt = __builtin_amdgcn_readfirstlane((int)tgt64);
    const bool keep = __builtin_amdgcn_readfirstlane((int)(tgt64 != ignore)) != 0;
    float f[NCH][8];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int rem = V - (k * 64 + lane) * 8;  // columns of this chunk inside the vocabulary (<= 0: none)
#pragma unroll
      for (int e = 0; e < 8; ++e) f[k][e] = (float)v[k][e];
      if (__builtin_amdgcn_ballot_w64(rem < 8) != 0) {  // some lane of this chunk holds padding: at most two chunks of a row
#pragma unroll
        for (int e = 0; e < 8; ++e) f[k][e] = (e < rem) ? f[k][e] : -INFINITY;
      }
    }
    float mx = -INFINITY;
    int amax = 0x7fffffff;
    if constexpr (ARG) {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = (k * 64 + lane) * 8 + e;
          if (f[k][e] > mx) {  // ascending c within the lane: the first maximum stays (padding is -inf: never greater)
            mx = f[k][e];
            amax = c;
          }
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float omx = __shfl_xor(mx, o, 64);
        const int oam = __shfl_xor(amax, o, 64);
        if (omx > mx || (omx == mx && oam < amax)) {
          mx = omx;
          amax = oam;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[k][e]);
      mx = wave_max(mx);
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += __builtin_amdgcn_exp2f((f[k][e] - mx) * LOG2E_F);
    sum = wave_sum(sum);
    const float lse = mx + __logf(sum);
    if (lane == 0) {
      row_loss[r] = keep ? (lse - (float)row[tgt]) : 0.f;
      if constexpr (ARG) argmax_out[r] = amax;
    }
    if (dlogits != nullptr) {
      bf16* drow = dlogits + r * ldl;
      if (!keep) {  // (scalar branch) an ignored row: zeros, nothing to compute
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16)0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int ch = k * 64 + lane;
          if (ch < nchunk) *reinterpret_cast<bf16x8*>(drow + ch * 8) = z;
        }
      } else {
        const int tk = tgt >> 9, tl = (tgt >> 3) & 63, te = tgt & 7;  // the target's chunk index k, lane and element
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          const int ch = k * 64 + lane;
          float g[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) g[e] = __builtin_amdgcn_exp2f((f[k][e] - lse) * LOG2E_F);
          if (k == tk) {  // (scalar branch: one chunk of the row)
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = (lane == tl && e == te) ? g[e] - 1.f : g[e];
          }
          bf16x8 g8;
#pragma unroll
          for (int e = 0; e < 8; ++e) g8[e] = (bf16)(g[e] * scale);
          if (ch < nchunk) *reinterpret_cast<bf16x8*>(drow + ch * 8) = g8;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) v[k] = vn[k];
  }
}

extern "C" int mh_cross_entropy(const void* logits, int64_t ldl, const int64_t* target, float* row_loss, void* dlogits,
                                const float* scale_dev, int64_t* argmax_out, int64_t R, int V, int64_t ignore,
                                int dtype, void* stream) {
  MH_REQUIRE(R > 0 && V > 0 && ldl >= V, "cross_entropy: bad shape R=%ld V=%d ldl=%ld", (long)R, V, (long)ldl);
  int64_t g = (R + 3) / 4;
  if (g > 65536) g = 65536;
  const bool vec = dtype == MH_BF16 && ldl % 8 == 0 && ldl <= 4096 && ((uintptr_t)logits & 15) == 0 &&
                   (dlogits == nullptr || ((uintptr_t)dlogits & 15) == 0);
  // two resident blocks per CU, every wave walks ~R / 2048 rows with the next one prefetched (r06, the 132-VGPR form: 768 blocks
  // 90.7 us against 93.0 on the step's chunk, the arg-max form 119.8 against 104.7: left at 512)
  if (vec && g > 512) g = 512;
#define MH_CE_VEC(NCH_)                                                                                                   \
  do {                                                                                                                    \
    if (argmax_out != nullptr)                                                                                            \
      cross_entropy_vec_kernel<NCH_, true><<<(int)g, 256, 0, (hipStream_t)stream>>>(                                       \
          (const bf16*)logits, ldl, target, row_loss, (bf16*)dlogits, scale_dev, argmax_out, R, V, ignore);               \
    else                                                                                                                  \
      cross_entropy_vec_kernel<NCH_, false><<<(int)g, 256, 0, (hipStream_t)stream>>>(                                      \
          (const bf16*)logits, ldl, target, row_loss, (bf16*)dlogits, scale_dev, argmax_out, R, V, ignore);               \
  } while (0)
  if (vec && ldl > 3584) MH_CE_VEC(8);
  else if (vec && ldl > 2048) MH_CE_VEC(7);
  else if (vec) MH_CE_VEC(4);
  else
  DISPATCH_T(dtype, (cross_entropy_kernel<T><<<(int)g, 256, 0, (hipStream_t)stream>>>(
                        (const T*)logits, ldl, target, row_loss, (T*)dlogits, scale_dev, argmax_out, R, V, ignore)));
#undef MH_CE_VEC
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// deterministic single-block reductions (inputs are at most a few MB)
// ---------------------------------------------------------------------------------------------------
__device__ inline float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  float t = (threadIdx.x < 16) ? sh[threadIdx.x] : 0.f;
  if (wv == 0) t = wave_sum(t);
  return t;  // valid in wave 0
}

__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float sh[16];
  float s = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {  // eight loads in flight, added in index order (one block: a latency chain)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = x[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; i += 1024) s += x[i];
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) out[0] = s;
}

extern "C" int mh_sum_f32(const float* x, int64_t n, float* out, void* stream) {
  MH_REQUIRE(n > 0, "sum_f32: empty");
  sum_f32_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(x, n, out);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

__global__ __launch_bounds__(1024) void count_valid_kernel(const int64_t* __restrict__ t, int64_t n, int64_t ignore,
                                                           float* __restrict__ count, float* __restrict__ inv) {
  __shared__ float sh[16];
  float s = 0.f;
  int64_t i = threadIdx.x;
  for (; i + 7 * 1024 < n; i += 8 * 1024) {
    int64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = t[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (v[u] != ignore) ? 1.f : 0.f;
  }
  for (; i < n; i += 1024) s += (t[i] != ignore) ? 1.f : 0.f;
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) {
    count[0] = s;
    inv[0] = 1.f / fmaxf(s, 1.f);
  }
}

extern "C" int mh_count_valid(const int64_t* target, int64_t n, int64_t ignore, float* count, float* inv, void* stream) {
  MH_REQUIRE(n > 0 && n < (1 << 24), "count_valid: n out of range (fp32 counter)");
  count_valid_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(target, n, ignore, count, inv);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// sum of squares of a flat gradient buffer: 1024 block partials, then one block folds them
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ g, int64_t n, float* __restrict__ partial) {
  constexpr int N = Pack<T>::N;
  __shared__ float sh[4];
  float s = 0.f;
  const int64_t nvec = n / N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    Pack<T> v = ld16(g + i * N);
#pragma unroll
    for (int e = 0; e < N; ++e) s += v.get(e) * v.get(e);
  }
  if (blockIdx.x == 0)
    for (int64_t i = nvec * N + threadIdx.x; i < n; i += 256) s += to_f(g[i]) * to_f(g[i]);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(1024) void fold_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out,
                                                             int accumulate) {
  __shared__ float sh[16];
  float s = (threadIdx.x < n) ? partial[threadIdx.x] : 0.f;
  s = block_sum_1024(s, sh);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

extern "C" int mh_sumsq(const void* g, int64_t n, float* partial1024, float* out, int accumulate, int dtype, void* stream) {
  MH_REQUIRE(n > 0 && ((uintptr_t)g & 15) == 0, "sumsq: empty or unaligned");
  DISPATCH_T(dtype, (sumsq_partial_kernel<T><<<1024, 256, 0, (hipStream_t)stream>>>((const T*)g, n, partial1024)));
  MH_LAUNCH_CHECK();
  fold_partials_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(partial1024, 1024, out, accumulate);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ coef,
                                 float* __restrict__ norm) {
  const float nrm = sqrtf(sumsq[0]);
  norm[0] = nrm;
  coef[0] = fminf(1.f, max_norm / (nrm + 1e-6f));
}

extern "C" int mh_clip_coef(const float* sumsq, float max_norm, float* coef, float* norm, void* stream) {
  clip_coef_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sumsq, max_norm, coef, norm);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW single-tensor semantics; gradient pre-scaled by the clip coefficient)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void adamw_kernel(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m,
                                                    T* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, const float* __restrict__ coef_dev) {
  constexpr int N = Pack<T>::N;
  const float coef = coef_dev ? coef_dev[0] : 1.f;
  const float step = lr / bc1, sq2 = sqrtf(bc2), decay = 1.f - lr * wd;
  const int64_t nvec = n / N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    Pack<T> pv = ld16(p + i * N), gv = ld16(g + i * N), mv = ld16(m + i * N), vv = ld16(v + i * N);
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float gr = rnd<T>(gv.get(e) * coef);
      float pe = rnd<T>(pv.get(e) * decay);
      const float me = rnd<T>(mv.get(e) + (gr - mv.get(e)) * (1.f - b1));
      const float ve = rnd<T>(rnd<T>(vv.get(e) * b2) + (1.f - b2) * gr * gr);
      const float den = rnd<T>(rnd<T>(sqrtf(ve)) / sq2) + eps;
      pe = pe - step * (me / rnd<T>(den));
      pv.set(e, pe);
      mv.set(e, me);
      vv.set(e, ve);
    }
    st16(p + i * N, pv);
    st16(m + i * N, mv);
    st16(v + i * N, vv);
  }
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * N + threadIdx.x; i < n; i += 256) {
      const float gr = rnd<T>(to_f(g[i]) * coef);
      float pe = rnd<T>(to_f(p[i]) * decay);
      const float me = rnd<T>(to_f(m[i]) + (gr - to_f(m[i])) * (1.f - b1));
      const float ve = rnd<T>(rnd<T>(to_f(v[i]) * b2) + (1.f - b2) * gr * gr);
      const float den = rnd<T>(rnd<T>(sqrtf(ve)) / sq2) + eps;
      pe = pe - step * (me / rnd<T>(den));
      p[i] = from_f<T>(pe);
      m[i] = from_f<T>(me);
      v[i] = from_f<T>(ve);
    }
  }
}

extern "C" int mh_adamw(void* p, const void* g, void* m, void* v, int64_t n, float lr, float beta1, float beta2, float eps,
                        float weight_decay, float bias_corr1, float bias_corr2, const float* coef_dev, int dtype,
                        void* stream) {
  MH_REQUIRE(n > 0, "adamw: empty");
  MH_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw: buffers must be 16-byte aligned");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  DISPATCH_T(dtype, (adamw_kernel<T><<<(int)blocks, 256, 0, (hipStream_t)stream>>>(
                        (T*)p, (const T*)g, (T*)m, (T*)v, n, lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2,
                        coef_dev)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// sampler front end: probs = softmax(logits / temp) * grammar mask (one wave per row)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void masked_softmax_kernel(const T* __restrict__ logits, int64_t ldl,
                                                             const int32_t* __restrict__ lo, const int32_t* __restrict__ hi,
                                                             const uint8_t* __restrict__ first_mask,
                                                             float* __restrict__ probs, int64_t B, int V, float temp) {
  const int lane = threadIdx.x & 63;
  for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < B; b += (int64_t)gridDim.x * 4) {
    const T* row = logits + b * ldl;
    float mx = -INFINITY;
    for (int c = lane; c < V; c += 64) mx = fmaxf(mx, rnd<T>(to_f(row[c]) / temp));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < V; c += 64) sum += __expf(rnd<T>(to_f(row[c]) / temp) - mx);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const int l = lo[b], h = hi[b];
    for (int c = lane; c < V; c += 64) {
      const bool ok = (l < 0) ? (first_mask[c] != 0) : (c >= l && c < h);
      probs[b * V + c] = ok ? __expf(rnd<T>(to_f(row[c]) / temp) - mx) * inv : 0.f;
    }
  }
}

extern "C" int mh_masked_softmax(const void* logits, int64_t ldl, const int32_t* lo, const int32_t* hi,
                                 const uint8_t* first_mask, float* probs, int64_t B, int V, float temp, int dtype,
                                 void* stream) {
  MH_REQUIRE(B > 0 && V > 0 && temp > 0.f, "masked_softmax: bad args");
  DISPATCH_T(dtype, (masked_softmax_kernel<T><<<(int)((B + 3) / 4), 256, 0, (hipStream_t)stream>>>(
                        (const T*)logits, ldl, lo, hi, first_mask, probs, B, V, temp)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// Fused sampler of generate(): grammar-masked softmax -> top-p / top-k filter -> draw, one block per row.
// Follows midi_model.py:202-228 + sample_top_p_k (:152-165) + torch.multinomial's single-draw path
// (argmax(p / q), q ~ Exp(1), ATen/native/Distributions.cpp) with q SUPPLIED by the caller, who draws it with
// torch.Tensor.exponential_ from the caller's generator exactly as multinomial would: q[b, j] belongs to the j-th
// largest probability of row b.  Only the top_k largest entries can have non-zero filtered probability, so the full
// sort is replaced by top_k rounds of a block-wide arg-max (value descending, index ascending = the order of torch's
// stable radix sort).  Differences from the op-by-op path are confined to fp32 summation order (softmax
// denominator, cumulative sum, renormalisation): ulp-level, affecting the draw only at exact ties.
// ---------------------------------------------------------------------------------------------------
constexpr int SAMPLE_MAX_K = 64;

// One block per row.  Only ids inside the grammar mask can be drawn, and the mask of a position is a
// contiguous id range of at most SAMPLE_MAX_RANGE ids (the largest parameter range, `duration`, has 2048): a lane keeps
// the probabilities of its <= 32 candidates (c = lo + lane + 64 t) in registers with a bit mask of the ones already taken,
// and a round is a register arg-max plus one DPP ladder over the wave -- no LDS traffic, no block barrier; this part
// runs on wave 0.  The softmax denominator runs over the whole vocabulary: one online max/sum pass shared by the block's
// four waves (the exact fp32 division by the temperature makes it the expensive part).
constexpr int SAMPLE_MAX_RANGE = 2048;

// Latency notes (r02, tools/decode_probe.py: the first form took 19-38 us per call, 8 calls per generated event).  Every
// global load whose address does not depend on the softmax statistics is requested at kernel entry -- the Exp(1) variates of
// the top_k ranks (the first form read them one by one on lane 0 inside the final loop: top_k dependent L2 round trips), the
// event id -> range-table -> candidate-logit chain -- and the vocabulary pass keeps its <= 14 values per thread in registers
// (one batch of loads, independent exponentials) instead of an online max/sum recurrence.  Ranges of at most 128 ids (six of
// the eight token positions) are ranked by counting (each lane compares its two candidates with all 128 through v_readlane)
// instead of top_k rounds of a wave-wide arg-max; the cumulative top-p filter and the final argmax(p / q) run with one rank
// per lane, the sums in the reference's sequential order.
template <typename T, int TMAX>  // TMAX: candidates per lane (64 * TMAX >= the longest mask range of this position)
__global__ __launch_bounds__(256) void sample_top_p_k_kernel(const T* __restrict__ logits, int64_t ldl,
                                                             const uint8_t* __restrict__ first_mask,
                                                             const uint8_t* __restrict__ ban_mask,
                                                             const int32_t* __restrict__ lo_tab,
                                                             const int32_t* __restrict__ hi_tab, int tab_stride,
                                                             const int64_t* __restrict__ ev, int pos, int first_lo,
                                                             int first_hi, const float* __restrict__ q,
                                                             int64_t* __restrict__ out, int64_t out_stride,
                                                             int64_t* __restrict__ out_b, int64_t* __restrict__ out_c,
                                                             int64_t B, int V, float temp, float top_p, int top_k,
                                                             int fill_rest, int64_t fill_id) {
  __shared__ float sel_v[SAMPLE_MAX_K];
  __shared__ int sel_i[SAMPLE_MAX_K];
  __shared__ float part_m[4], part_s[4];
  // Long mask ranges (TMAX > 2: `bpm`, `duration`) are spread over the block's four waves (r04): wave w keeps candidate slots
  // [w TW, (w + 1) TW) of every lane and finds ITS top_k in top_k rounds of a register arg-max over TW values (8 instead of
  // 32 for the 2048-id range) + one DPP ladder; wave 0 then ranks the 4 x top_k survivors by counting, as it ranks a short range.
  constexpr bool SPLIT = TMAX > 2;
  constexpr int TW = SPLIT ? TMAX / 4 : TMAX;  // candidate slots per lane of one wave
  __shared__ unsigned long long wl[SPLIT ? 4 : 1][SAMPLE_MAX_K];
  // (every kernel argument fetched at entry in one batch: see attn_decode_kernel)
  asm volatile("" ::"s"(logits), "s"(ldl), "s"(first_mask), "s"(ban_mask), "s"(lo_tab), "s"(hi_tab), "s"(tab_stride), "s"(ev));
  asm volatile("" ::"s"(pos), "s"(first_lo), "s"(first_hi), "s"(q), "s"(out), "s"(out_stride), "s"(out_b), "s"(out_c));
  asm volatile("" ::"s"(V), "s"(temp), "s"(top_p), "s"(top_k), "s"(fill_rest), "s"(fill_id));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = blockIdx.x;
  const T* row = logits + b * ldl;
  // ---- wave 0: everything that does not need the softmax statistics, requested first
  float qv = 1.f, zc[TW];
  int l = first_lo, h = first_hi;
  const int t0 = SPLIT ? wave * TW : 0;  // this wave's first candidate slot
  if (SPLIT || wave == 0) {
    if (wave == 0 && lane < top_k) qv = q[b * (int64_t)V + lane];
    if (pos > 0) {
      const int64_t e = ev[b];
      l = lo_tab[e * tab_stride + pos];
      h = hi_tab[e * tab_stride + pos];
    }
    if (h > V) h = V;
    // every load unconditional on a clamped index, all requested before the first use: a per-element "load or constant"
    // select makes hipcc branch around each load and wait for it -- one memory round trip per element (r02: the first form
    // of this kernel took 19-38 us that way; cdna_hip_programming.md 5 trap (c))
    T zraw[TW];
    uint8_t fm[TW], bm[TW];
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      int c = l + lane + 64 * (t0 + t);
      c = c < V ? c : V - 1;
      zraw[t] = row[c];
      fm[t] = first_mask[c];
      bm[t] = ban_mask[c];  // (always a real mask: an optional pointer costs a branch + wait per element here)
    }
#pragma unroll
    for (int t = 0; t < TW; ++t) {
      const int c = l + lane + 64 * (t0 + t);
      const bool ok = c < h && (pos > 0 || fm[t] != 0) && bm[t] == 0;
      zc[t] = ok ? rnd<T>(to_f(zraw[t]) / temp) : -INFINITY;  // -inf: not a candidate
    }
  }
  // ---- softmax statistics over the whole vocabulary (logits / temp in the activation dtype, as the reference)
  float m = -INFINITY, ssum = 0.f;
  constexpr int NZ = 14;  // 14 x 256 = 3584 >= vocab 3406: one batch
  for (int base = 0; base < V; base += NZ * 256) {
    T zr[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {  // (clamped index: unconditional loads, see above)
      const int c = base + threadIdx.x + 256 * i;
      zr[i] = row[c < V ? c : V - 1];
    }
    float zl[NZ];
#pragma unroll
    for (int i = 0; i < NZ; ++i) {
      const int c = base + threadIdx.x + 256 * i;
      zl[i] = c < V ? rnd<T>(to_f(zr[i]) / temp) : -INFINITY;
    }
    float cm = zl[0];
#pragma unroll
    for (int i = 1; i < NZ; ++i) cm = fmaxf(cm, zl[i]);
    if (cm > m) {
      ssum *= __expf(m - cm);  // (m = -inf: 0)
      m = cm;
    }
    if (m > -INFINITY) {
#pragma unroll
      for (int i = 0; i < NZ; ++i) ssum += __expf(zl[i] - m);
    }
  }
  {
    const float wm = wave_max(m);
    const float ws = wave_sum(m > -INFINITY ? ssum * __expf(m - wm) : 0.f);
    if (lane == 0) {
      part_m[wave] = wm;
      part_s[wave] = ws;
    }
  }
  if (wave == 0 && lane < SAMPLE_MAX_K) {  // fewer than top_k candidates: the tail has probability 0
    sel_v[lane] = 0.f;
    sel_i[lane] = 0x7fffffff;
  }
  __syncthreads();
  if (!SPLIT && wave != 0) return;
  const float mx = fmaxf(fmaxf(part_m[0], part_m[1]), fmaxf(part_m[2], part_m[3]));
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) tot += part_s[w] * __expf(part_m[w] - mx);
  const float inv = 1.f / tot;
  float pv[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) pv[t] = (zc[t] > -INFINITY) ? __expf(zc[t] - mx) * inv : -1.f;  // -1: not a candidate

  // ---- the top_k largest candidates in the order of torch's stable descending sort (value descending, id ascending):
  // rank j -> sel_v[j], sel_i[j]
  if constexpr (TMAX <= 2) {
    // rank = number of candidates that sort before this one.  One 64-bit key per candidate -- (probability bits, ~position):
    // larger probability first, lower id among equal probabilities; 0 = not a candidate -- so a pair costs one unsigned
    // compare + add-with-carry and the loop has no branches (written with || / && it compiled to four exec-mask branches per
    // step: 64 x ~100 instructions, 7 us of the kernel's 15, r02 trace).
    uint64_t key[TMAX];
    int rank[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      key[t] = pv[t] >= 0.f ? ((uint64_t)__float_as_uint(pv[t]) << 32) | (uint32_t)(0xffffffffu - (uint32_t)(lane + 64 * t)) : 0ull;
      rank[t] = 0;
    }
    for (int sl = 0; sl < 64; ++sl) {
#pragma unroll
      for (int u = 0; u < TMAX; ++u) {
        const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key[u] >> 32), sl);
        const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key[u], sl);
        const uint64_t ok = ((uint64_t)ohi << 32) | olo;
#pragma unroll
        for (int t = 0; t < TMAX; ++t) rank[t] += (int)(ok > key[t]);
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (key[t] != 0ull && rank[t] < top_k) {
        sel_v[rank[t]] = pv[t];
        sel_i[rank[t]] = l + lane + 64 * t;
      }
  } else {
    // this wave's top_k, in order, into wl[wave][.] (0 = the wave has fewer candidates)
    if constexpr (TW <= 2) {
      // at most 128 candidates per wave: ranked by counting (the short-range method above), no rounds
      wl[wave][lane] = 0ull;
      uint64_t key[TW];
      int rank[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) {
        key[t] = pv[t] >= 0.f ? ((uint64_t)__float_as_uint(pv[t]) << 32) | (uint32_t)(0x7fffffff - (l + lane + 64 * (t0 + t))) : 0ull;
        rank[t] = 0;
      }
      for (int sl = 0; sl < 64; ++sl) {
#pragma unroll
        for (int u = 0; u < TW; ++u) {
          const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key[u] >> 32), sl);
          const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key[u], sl);
          const uint64_t ok = ((uint64_t)ohi << 32) | olo;
#pragma unroll
          for (int t = 0; t < TW; ++t) rank[t] += (int)(ok > key[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TW; ++t)
        if (key[t] != 0ull && rank[t] < top_k) wl[wave][rank[t]] = key[t];
    } else {
    uint32_t taken = 0;
    for (int j = 0; j < top_k; ++j) {
      float bv = -1.f;
      int bt = TW;
#pragma unroll
      for (int t = 0; t < TW; ++t)
        if (!((taken >> t) & 1u) && pv[t] > bv) {  // ascending t = ascending id: the lowest id of equal values stays
          bv = pv[t];
          bt = t;
        }
      // key = (probability bits, 0x7fffffff - id): larger probability first, lower id among equal probabilities; a lane
      // without candidates left offers key 0 (below every candidate, whose low word is positive)
      const uint64_t mykey = (bt < TW) ? ((uint64_t)__float_as_uint(bv) << 32) | (uint32_t)(0x7fffffff - (l + lane + 64 * (t0 + bt))) : 0ull;
      const uint64_t wkey = wave_max_u64_fast(mykey);
      const int wi = (wkey != 0ull) ? 0x7fffffff - (int)(uint32_t)wkey : 0x7fffffff;
      if (lane == 0) wl[wave][j] = wkey;
      if (wi != 0x7fffffff && ((wi - l) & 63) == lane) taken |= 1u << (((wi - l) >> 6) - t0);
    }
    }
    __syncthreads();  // (every wave of a SPLIT block gets here: none returned above)
    if (wave != 0) return;
    // rank the 4 x top_k survivors by counting: lane j < top_k holds entry j of each wave's list; keys are distinct (distinct
    // ids) or 0, so the ranks of the non-zero keys are a permutation and the first top_k of them are the block's top_k
    uint64_t key[4];
    int rank[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      key[u] = lane < top_k ? wl[u][lane] : 0ull;
      rank[u] = 0;
    }
    for (int sl = 0; sl < top_k; ++sl) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const uint32_t ohi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key[v] >> 32), sl);
        const uint32_t olo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key[v], sl);
        const uint64_t ok = ((uint64_t)ohi << 32) | olo;
#pragma unroll
        for (int u = 0; u < 4; ++u) rank[u] += (int)(ok > key[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (key[u] != 0ull && rank[u] < top_k) {
        sel_v[rank[u]] = __uint_as_float((uint32_t)(key[u] >> 32));
        sel_i[rank[u]] = 0x7fffffff - (int)(uint32_t)key[u];
      }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS writes of this wave are done (one wave, in-order LDS queue)
  __builtin_amdgcn_wave_barrier();
  // ---- probs_sort[cumsum - probs_sort > p] = 0; keep the first k; renormalise; argmax(p / q): rank j on lane j
  const float v = lane < top_k ? sel_v[lane] : 0.f;
  const int vid = lane < top_k ? sel_i[lane] : 0x7fffffff;
  float cum = 0.f, mycum = 0.f;
  for (int j = 0; j < top_k; ++j) {  // the reference's sequential cumsum, recomputed identically by every lane
    cum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
    if (lane == j) mycum = cum;
  }
  const float kept = (lane < top_k && !(mycum - v > top_p)) ? v : 0.f;
  float s = 0.f;
  for (int j = 0; j < top_k; ++j) s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(kept), j));
  float r = (kept / s) / qv;
  if (!(r == r)) r = 0.f;  // (no candidate at all: 0 / 0)
  // first maximum = lowest rank among equal ratios (`if (r > best)` of the serial form); ratios are >= 0, so their bits order them
  const uint64_t key = lane < top_k ? ((uint64_t)__float_as_uint(r) << 32) | (uint32_t)(63 - lane) : 0ull;
  const uint64_t wkey = wave_max_u64_fast(key);
  const int bj = 63 - (int)(uint32_t)wkey;
  const int64_t id = (int64_t)__builtin_amdgcn_readlane(vid, bj);
  if (lane == 0) {
    out[b * out_stride] = id;
    for (int j = 1; j <= fill_rest; ++j) out[b * out_stride + j] = fill_id;  // position 0 opens a fresh event row
    if (out_b != nullptr) out_b[b] = id;
    if (out_c != nullptr) out_c[b] = id;
  }
}

extern "C" int mh_sample_top_p_k(const void* logits, int64_t ldl, const uint8_t* first_mask, const uint8_t* ban_mask,
                                 int first_lo, int first_hi,
                                 const int32_t* lo_tab, const int32_t* hi_tab, int tab_stride, int max_range,
                                 const int64_t* ev, int pos, const float* q,
                                 int64_t* out, int64_t out_stride, int64_t* out_b, int64_t* out_c, int64_t B, int V,
                                 float temp, float top_p, int top_k, int fill_rest, int64_t fill_id, int dtype,
                                 void* stream) {
  MH_REQUIRE(B > 0 && V > 0 && temp > 0.f && pos >= 0 && pos < tab_stride && fill_rest >= 0 && fill_rest < out_stride,
             "sample_top_p_k: bad args");
  MH_REQUIRE(first_mask != nullptr && ban_mask != nullptr && q != nullptr,
             "sample_top_p_k: first_mask, ban_mask (V bytes each; all zero = nothing banned) and q are required");
  MH_REQUIRE(pos == 0 || (ev != nullptr && lo_tab != nullptr && hi_tab != nullptr), "sample_top_p_k: position %d needs the event ids and range tables", pos);
  MH_REQUIRE(top_k >= 1 && top_k <= SAMPLE_MAX_K && top_k <= V, "sample_top_p_k: top_k=%d outside [1, %d]", top_k,
             SAMPLE_MAX_K);
  MH_REQUIRE(first_lo >= 0 && first_hi <= V && first_hi - first_lo <= SAMPLE_MAX_RANGE && max_range <= SAMPLE_MAX_RANGE,
             "sample_top_p_k: a grammar mask spans more than %d ids (first %d..%d, parameters %d)", SAMPLE_MAX_RANGE, first_lo,
             first_hi, max_range);
  const int span = pos == 0 ? first_hi - first_lo : max_range;
#define MH_SAMPLE(TMAX_)                                                                                                  \
  DISPATCH_T(dtype, (sample_top_p_k_kernel<T, TMAX_><<<(int)B, 256, 0, (hipStream_t)stream>>>(               \
                        (const T*)logits, ldl, first_mask, ban_mask, lo_tab, hi_tab, tab_stride, ev, pos, first_lo, first_hi, q, out, \
                        out_stride, out_b, out_c, B, V, temp, top_p, top_k, fill_rest, fill_id)))
  if (span <= 128) MH_SAMPLE(2);
  else if (span <= 512) MH_SAMPLE(8);
  else MH_SAMPLE(32);
#undef MH_SAMPLE
  MH_LAUNCH_CHECK();
  return MH_OK;
}
