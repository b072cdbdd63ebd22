// HBM-bound kernels of the path: embeddings, RMSNorm, RoPE, SwiGLU and their backward passes.
// All of them stream rows with 16-byte vector accesses (8 bf16 / 4 fp32 per lane), keep statistics in
// fp32, and reduce across the 64-lane wave with shuffles.  Roofline: HBM (8 TB/s spec).
#include "common.h"

#define DISPATCH_T(dtype, CALL)                                    \
  do {                                                             \
    if ((dtype) == MH_BF16) { using T = bf16; CALL; }              \
    else if ((dtype) == MH_F32) { using T = float; CALL; }         \
    else { mh_set_error("bad dtype %d", (int)(dtype)); return MH_ERR_ARG; } \
  } while (0)

static inline int grid_for(int64_t items, int per_block, int cap = 1 << 20) {
  int64_t g = (items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------------------------------------------
// embeddings
// ---------------------------------------------------------------------------------------------------
// one wave per event row: out[m,:] = sum_j table[tok[m,j],:]
template <typename T>
__global__ __launch_bounds__(256) void embed_sum_fwd_kernel(const int64_t* __restrict__ tok, const T* __restrict__ table,
                                                            T* __restrict__ out, int64_t M, int TT, int D, int64_t V) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t m = wave0; m < M; m += (int64_t)gridDim.x * 4) {
    // the octet's ids first (one batch of loads), then all table rows: id -> row inside one loop is a chain of 2 TT dependent
    // memory round trips (14 us per decode step for TT = 8, r02 trace)
    constexpr int TMAXI = 8;
    int64_t ids[TMAXI];
    if (TT <= TMAXI) {
#pragma unroll
      for (int j = 0; j < TMAXI; ++j) ids[j] = tok[m * TT + (j < TT ? j : 0)];
#pragma unroll
      for (int j = 0; j < TMAXI; ++j)
        if ((uint64_t)ids[j] >= (uint64_t)V) __builtin_trap();  // torch's embedding raises a device assert here; never read out of bounds
    }
    for (int c = lane * N; c < D; c += 64 * N) {
      float acc[N];
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = 0.f;
      if (TT <= TMAXI) {
        Pack<T> v[TMAXI];
#pragma unroll
        for (int j = 0; j < TMAXI; ++j) v[j] = ld16(table + ids[j] * D + c);
#pragma unroll
        for (int j = 0; j < TMAXI; ++j)
          if (j < TT) {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] += v[j].get(e);
          }
      } else {
        for (int j = 0; j < TT; ++j) {
          const int64_t id = tok[m * TT + j];
          if ((uint64_t)id >= (uint64_t)V) __builtin_trap();
          Pack<T> v = ld16(table + id * D + c);
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] += v.get(e);
        }
      }
      Pack<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e) o.set(e, acc[e]);
      st16(out + m * D + c, o);
    }
  }
}

extern "C" int mh_embed_sum_fwd(const int64_t* tok, const void* table, void* out, int64_t M, int T_, int64_t V, int D,
                                int dtype, void* stream) {
  MH_REQUIRE(M > 0 && T_ > 0 && D % 8 == 0, "embed_sum_fwd: bad shape M=%ld T=%d D=%d", (long)M, T_, D);
  DISPATCH_T(dtype, (embed_sum_fwd_kernel<T><<<grid_for(M, 4, 65536), 256, 0, (hipStream_t)stream>>>(
                        tok, (const T*)table, (T*)out, M, T_, D, V)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// out[m,0,:] = hidden[m,:]; out[m,j,:] = table[tok[m*ldtok + j-1],:]
template <typename T>
__global__ __launch_bounds__(256) void concat_tok_fwd_kernel(const T* __restrict__ hidden, const int64_t* __restrict__ tok,
                                                             int64_t ldtok, const T* __restrict__ table,
                                                             T* __restrict__ out, int64_t M, int TT, int D, int64_t V) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t rows = M * TT;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const int64_t m = r / TT;
    const int j = (int)(r - m * TT);
    const int64_t id = (j == 0) ? 0 : tok[m * ldtok + j - 1];
    if ((uint64_t)id >= (uint64_t)V) __builtin_trap();  // (as embed_sum_fwd)
    const T* src = (j == 0) ? hidden + m * D : table + id * D;
    for (int c = lane * N; c < D; c += 64 * N) st16(out + r * D + c, ld16(src + c));
  }
}

extern "C" int mh_concat_tok_fwd(const void* hidden, const int64_t* tok, int64_t ldtok, const void* table, void* out,
                                 int64_t M, int T_, int64_t V, int D, int dtype, void* stream) {
  MH_REQUIRE(M > 0 && T_ > 0 && D % 8 == 0, "concat_tok_fwd: bad shape");
  DISPATCH_T(dtype, (concat_tok_fwd_kernel<T><<<grid_for(M * T_, 4, 65536), 256, 0, (hipStream_t)stream>>>(
                        (const T*)hidden, tok, ldtok, (const T*)table, (T*)out, M, T_, D, V)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// scatter-add of embedding-row gradients into an fp32 [V,D] accumulator (hardware fp32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void embed_scatter_bwd_kernel(const int64_t* __restrict__ tok, int64_t ldtok, int TT,
                                                                const T* __restrict__ dout, int rows_per_m, int jstride,
                                                                int j0, float* __restrict__ dtab, int64_t M, int D,
                                                                int64_t pad_id) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  const int64_t items = M * TT;
  for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {
    const int64_t m = it / TT;
    const int j = (int)(it - m * TT);
    const int64_t id = tok[m * ldtok + j];
    if (id == pad_id) continue;
    const T* src = dout + (m * rows_per_m + (int64_t)j * jstride + j0) * (int64_t)D;
    float* dst = dtab + id * D;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> v = ld16(src + c);
#pragma unroll
      for (int e = 0; e < N; ++e) atomicAdd(dst + c + e, v.get(e));
    }
  }
}

extern "C" int mh_embed_scatter_bwd(const int64_t* tok, int64_t ldtok, int T_, const void* dout, int rows_per_m,
                                    int jstride, int j0, float* dtable_f32, int64_t M, int64_t V, int D, int64_t pad_id, int dtype,
                                    void* stream) {
  MH_REQUIRE(M > 0 && T_ > 0 && D % 8 == 0, "embed_scatter_bwd: bad shape");
  DISPATCH_T(dtype, (embed_scatter_bwd_kernel<T><<<grid_for(M * T_, 4, 16384), 256, 0, (hipStream_t)stream>>>(
                        tok, ldtok, T_, (const T*)dout, rows_per_m, jstride, j0, dtable_f32, M, D, pad_id)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// Segment form of the same gradient (the production path): the token occurrences are pre-sorted by token id
// (src[i] = row of `dout` that occurrence i reads, seg[v]..seg[v+1] = the occurrences of id v).  The sorted list is cut
// into equal pieces of 128 occurrences per wave, whatever ids they belong to: a wave looks up 64 (row, id) pairs at a time
// with one coalesced load and a binary search in `seg`, then streams the rows four at a time with 16-byte loads, summing in
// registers while the id stays the same and adding the run to the fp32 table when it changes (one atomic per element per
// run; the scatter form needs one per occurrence and serialises ~29k of them on the row of the ubiquitous "note" id).
// Equal pieces matter: with one block per id the 29.5k rows of "note" were a serial chain of dependent index -> row
// loads and the launch took 1.05 ms for 0.5 GB (profiles/r01_run9), ten times the HBM time.
// Roofline: HBM; algorithmic bytes = n_occ * D * sizeof(T) (every occurrence's row once).
// r06: ~2600 ids x two runs each x 1024 fp32 atomics, executed at the memory side, sat behind every run.  The cuts between
// the waves' pieces are now SNAPPED to segment starts: the piece of wave w begins at snap(128 w), where snap(p) = the start of
// the segment holding p when that segment has at most SEG_OWN occurrences, else p itself.  Every segment of <= SEG_OWN occurrences
// then lies inside exactly one wave's piece (the last wave whose nominal cut falls inside it, or the one whose piece it falls
// in whole) and is added to the table with plain loads + stores; only the few longer ones (the event ids, "note" first) keep
// one atomic per element per 128-occurrence run.  seg_start is searched in LDS (V + 1 32-bit counts; < 2^31 occurrences).
// The long segments' chains of atomics on one address (the 29.5 k occurrences of "note" were 230 runs adding to the same 1024
// floats) are cut as well: the waves of a workgroup (NW = 8: 1024 consecutive occurrences) meet in LDS before they touch the
// table -- every wave leaves the run it ends with in its LDS slot, and the first wave of each group of consecutive waves ending
// in the same id adds the group's slots and writes once: 29 links for "note" instead of 230 (ids are sorted, so a wave that ends
// in the id its predecessor ends in holds nothing else).  Measured (tools/embed_bwd_once.py, the step's two calls): 246 -> 238 us
// (token-level: 229 k rows of d seq gathered in id order, 2 TB/s -- what 2 KiB random gathers get) and 273 -> 191 us (event-level:
// every event's row read by its 8 tokens).  With ALL table traffic compiled out the launches take 238 / 190 us: the writes are
// no longer what they wait for, the gather is.
constexpr int SEG_OWN = 1024;

template <typename T, int NCH, bool SL, int NW>
__global__ __launch_bounds__(NW * 64) void embed_segment_bwd_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ seg,
                                                                const T* __restrict__ dout, int64_t ld,
                                                                float* __restrict__ dtab, int V, int D, int64_t n_occ,
                                                                int pad_id) {
  constexpr int N = Pack<T>::N;
  constexpr int RW = 64, G = 4;  // occurrences per pass (one per lane), rows in flight; a nominal piece is 2 RW occurrences
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  extern __shared__ __attribute__((aligned(16))) int seg_l[];  // SL: seg_start[0 .. V] for the binary searches; behind it the waves' slots
  constexpr int SLOT = NCH * 64 * N;                            // floats of one wave's run
  const int seg_words = SL ? ((V + 1 + 3) & ~3) : 0;
  float* part = reinterpret_cast<float*>(seg_l + seg_words);   // [NW][SLOT]
  int* fin_id = reinterpret_cast<int*>(part + NW * SLOT);      // [NW]: the id of the run a wave ended with (-1: none)
  if constexpr (SL) {
    for (int v = threadIdx.x; v <= V; v += NW * 64) seg_l[v] = (int)seg[v];
    __syncthreads();
  }
  auto seg_at = [&](int v) -> int64_t { return SL ? (int64_t)seg_l[v] : seg[v]; };
  auto owner = [&](int64_t p) {  // the largest v in [0, V) with seg[v] <= p
    int lo = 0, hi = V;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (seg_at(mid) <= p) lo = mid; else hi = mid;
    }
    return lo;
  };
  const int64_t n_in = (seg_at(V) < n_occ) ? seg_at(V) : n_occ;  // (occurrences of ids outside [0, V) sit behind seg[V]: skipped)
  auto snap = [&](int64_t p) -> int64_t {
    if (p >= n_in) return n_in;
    const int v = owner(p);
    return (seg_at(v + 1) - seg_at(v) <= SEG_OWN) ? seg_at(v) : p;
  };
  const int64_t w = (int64_t)blockIdx.x * NW + wv;
  const int64_t p_beg = snap(w * (2 * RW)), p_end = snap((w + 1) * (2 * RW));  // (wave-uniform; an empty piece still meets the others below)
  float acc[NCH][N];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int e = 0; e < N; ++e) acc[ch][e] = 0.f;
  int cur = -1;
  bool own = false;  // the run's id belongs to this wave alone
  auto flush = [&]() {
    float* dst = dtab + (int64_t)cur * D;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c = (ch * 64 + lane) * N;
      if (c < D) {
        if (own) {
#pragma unroll
          for (int e = 0; e < N; e += 4) {
            float4 t = *reinterpret_cast<float4*>(dst + c + e);
            t.x += acc[ch][e]; t.y += acc[ch][e + 1]; t.z += acc[ch][e + 2]; t.w += acc[ch][e + 3];
            *reinterpret_cast<float4*>(dst + c + e) = t;
          }
        } else {
#pragma unroll
          for (int e = 0; e < N; ++e) atomicAdd(dst + c + e, acc[ch][e]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) acc[ch][e] = 0.f;
      }
    }
  };
#pragma unroll 1
  for (int64_t p0 = p_beg; p0 < p_end; p0 += RW) {
  const int cnt = (p_end - p0 < RW) ? (int)(p_end - p0) : RW;
  int row_lo = 0, row_hi = 0, my_id = -1;
  if (lane < cnt) {
    const int64_t p = p0 + lane;
    const int64_t r = src[p];
    row_lo = (int)(uint32_t)r;
    row_hi = (int)(r >> 32);
    my_id = owner(p);
  }
#pragma unroll
  for (int j0 = 0; j0 < RW; j0 += G) {
    if (j0 < cnt) {  // (wave-uniform)
      Pack<T> buf[G][NCH];
      int idj[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        idj[g] = __builtin_amdgcn_readlane(my_id, j0 + g);  // -1 past the end of the piece
        const int64_t row = ((int64_t)__builtin_amdgcn_readlane(row_hi, j0 + g) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane(row_lo, j0 + g);
        if (idj[g] >= 0 && idj[g] != pad_id) {
#pragma unroll
          for (int ch = 0; ch < NCH; ++ch) {
            const int c = (ch * 64 + lane) * N;
            if (c < D) buf[g][ch] = ld16(dout + row * ld + c);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (idj[g] < 0 || idj[g] == pad_id) continue;
        if (idj[g] != cur) {
          if (cur >= 0) flush();
          cur = idj[g];
          own = seg_at(cur + 1) - seg_at(cur) <= SEG_OWN;
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          const int c = (ch * 64 + lane) * N;
          if (c < D) {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[ch][e] += buf[g][ch].get(e);
          }
        }
      }
    }
  }
  }
  // ---- the run this wave ended with: through LDS, one write per group of waves that ended in the same id
  if (cur >= 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int e = 0; e < N; e += 4)
        *reinterpret_cast<float4*>(part + wv * SLOT + (ch * 64 + lane) * N + e) = float4{acc[ch][e], acc[ch][e + 1], acc[ch][e + 2], acc[ch][e + 3]};
  }
  if (lane == 0) fin_id[wv] = cur;
  __syncthreads();
  if (cur < 0 || (wv > 0 && fin_id[wv - 1] == cur)) return;  // nothing to write / a member of an earlier wave's group
  for (int k = wv + 1; k < NW && fin_id[k] == cur; ++k) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int e = 0; e < N; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(part + k * SLOT + (ch * 64 + lane) * N + e);
        acc[ch][e] += t.x; acc[ch][e + 1] += t.y; acc[ch][e + 2] += t.z; acc[ch][e + 3] += t.w;
      }
  }
  flush();
}

extern "C" int mh_embed_segment_bwd(const int64_t* src_rows, const int64_t* seg_start, const void* dout, int64_t ld,
                                    float* dtable_f32, int64_t V, int D, int64_t n_occ, int64_t pad_id, int dtype,
                                    void* stream) {
  MH_REQUIRE(V > 0 && V < (1 << 30) && D % 8 == 0 && D <= 4096 && n_occ >= 0, "embed_segment_bwd: bad args");
  MH_REQUIRE(dtype != MH_F32 || D <= 2048, "embed_segment_bwd: fp32 supports D <= 2048");
  if (n_occ == 0) return MH_OK;
  MH_REQUIRE(n_occ < (1ll << 31), "embed_segment_bwd: too many occurrences");
  const bool sl = (V + 1) * 4 <= 16 * 1024;  // seg_start fits the LDS beside the waves' slots (every vocabulary of the reference: 3406 / 3408 ids)
  const int seg_words = sl ? (int)((V + 1 + 3) & ~3) : 0;
  const int per = 64 * (dtype == MH_F32 ? 4 : 8);
  const int nch = (D + per - 1) / per;
  // 8 waves (1024 occurrences) per workgroup while the waves' slots (NCH x 64 lanes x 16 or 32 bytes each) fit beside seg_start
  // in 64 KiB of LDS, else 4, else 1 (D = 4096: no meeting, the r01 form)
#define MH_SEG3(NCH_, SL_, NW_)                                                                                           \
  DISPATCH_T(dtype, (embed_segment_bwd_kernel<T, NCH_, SL_, NW_>                                                          \
                     <<<(unsigned)((n_occ + NW_ * 128 - 1) / (NW_ * 128)), NW_ * 64,                                       \
                        (size_t)seg_words * 4 + (size_t)NW_ * NCH_ * 64 * Pack<T>::N * 4 + NW_ * 4, (hipStream_t)stream>>>( \
                         src_rows, seg_start, (const T*)dout, ld, dtable_f32, (int)V, D, n_occ, (int)pad_id)))
#define MH_SEG(NCH_)                                                                                                      \
  do {                                                                                                                    \
    if (NCH_ <= 2) { if (sl) MH_SEG3(NCH_, true, 8); else MH_SEG3(NCH_, false, 8); }                                       \
    else if (NCH_ <= 4) { if (sl) MH_SEG3(NCH_, true, 4); else MH_SEG3(NCH_, false, 4); }                                  \
    else { if (sl) MH_SEG3(NCH_, true, 1); else MH_SEG3(NCH_, false, 1); }                                                 \
  } while (0)
  if (nch <= 1) MH_SEG(1);
  else if (nch <= 2) MH_SEG(2);
  else if (nch <= 4) MH_SEG(4);
  else MH_SEG(8);
#undef MH_SEG
#undef MH_SEG3
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---- token segments: the index preparation of embed_segment_bwd as a counting sort (r04; it was torch.sort + searchsorted +
// five index-arithmetic launches, ~25 library launches per call in the timed region of the training step).
// Occurrence i of the [n_rows, n_cols] id matrix (row r = i / n_cols, column j = i % n_cols) is grouped by token id:
// seg_start[v] = number of occurrences with id < v (v = 0 .. V; ids outside [0, V) -- a caller error, nn.Embedding raises -- are
// grouped AFTER seg_start[V] so that nothing is written out of bounds), src_rows[p] = r * row_mul + j * col_mul + add for the
// occurrence placed at p (the row of the gradient matrix that occurrence reads).  Three launches: per-chunk histograms in LDS,
// a running sum over the chunks per id, placement (LDS cursor per id; the order inside a segment is whatever order the LDS
// atomics return -- embed_segment_bwd adds a segment's rows with fp32 atomics across waves, so no order was ever promised).
// Integer work, HBM-bound and tiny: 8 bytes read + 8 written per occurrence.
constexpr int SEG_CH = 2048;  // occurrences per workgroup

__device__ inline int seg_bucket(int64_t id, int V) { return (id >= 0 && id < V) ? (int)id : V; }

__global__ __launch_bounds__(256) void token_hist_kernel(const int64_t* __restrict__ tok, int64_t ld, int64_t n, int ncol, int V,
                                                         int32_t* __restrict__ cnt) {
  extern __shared__ int seg_lds[];  // V + 1 counters
  for (int v = threadIdx.x; v <= V; v += 256) seg_lds[v] = 0;
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * SEG_CH;
#pragma unroll
  for (int s = 0; s < SEG_CH / 256; ++s) {
    const int64_t i = i0 + s * 256 + threadIdx.x;
    if (i < n) {
      const int64_t r = i / ncol;
      atomicAdd(&seg_lds[seg_bucket(tok[r * ld + (i - r * ncol)], V)], 1);
    }
  }
  __syncthreads();
  int32_t* out = cnt + (int64_t)blockIdx.x * (V + 1);
  for (int v = threadIdx.x; v <= V; v += 256) out[v] = seg_lds[v];
}

// cnt[b][v] -> number of occurrences of id v in the chunks before b; tot[v] = all of them
__global__ __launch_bounds__(256) void token_colscan_kernel(int32_t* __restrict__ cnt, int nb, int V, int32_t* __restrict__ tot) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v > V) return;
  int run = 0;
  for (int b = 0; b < nb; ++b) {
    int32_t* p = cnt + (int64_t)b * (V + 1) + v;
    const int c = *p;
    *p = run;
    run += c;
  }
  tot[v] = run;
}

__global__ __launch_bounds__(256) void token_place_kernel(const int64_t* __restrict__ tok, int64_t ld, int64_t n, int ncol, int V,
                                                          const int32_t* __restrict__ cnt, const int32_t* __restrict__ tot,
                                                          int64_t row_mul, int64_t col_mul, int64_t add,
                                                          int64_t* __restrict__ src_rows, int64_t* __restrict__ seg_start) {
  extern __shared__ int seg_lds[];  // V + 1 cursors: next position of id v for this chunk
  __shared__ int wsum[4];
  // exclusive scan of tot over the ids (every workgroup repeats it: V + 1 values, cheaper than a launch of its own)
  const int per = (V + 1 + 255) / 256, v0 = threadIdx.x * per;
  int mine = 0;
  for (int k = 0; k < per; ++k)
    if (v0 + k <= V) mine += tot[v0 + k];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int start = inc - mine;
  for (int w = 0; w < wave; ++w) start += wsum[w];
  const int32_t* mycnt = cnt + (int64_t)blockIdx.x * (V + 1);
  for (int k = 0; k < per; ++k) {
    const int v = v0 + k;
    if (v <= V) {
      if (blockIdx.x == 0) seg_start[v] = start;
      seg_lds[v] = start + mycnt[v];
      start += tot[v];
    }
  }
  __syncthreads();
  const int64_t i0 = (int64_t)blockIdx.x * SEG_CH;
#pragma unroll
  for (int s = 0; s < SEG_CH / 256; ++s) {
    const int64_t i = i0 + s * 256 + threadIdx.x;
    if (i < n) {
      const int64_t r = i / ncol;
      const int64_t j = i - r * ncol;
      const int p = atomicAdd(&seg_lds[seg_bucket(tok[r * ld + j], V)], 1);
      src_rows[p] = r * row_mul + j * col_mul + add;
    }
  }
}

extern "C" int mh_token_segments_chunk(void) { return SEG_CH; }

extern "C" int mh_token_segments(const int64_t* tok, int64_t ldtok, int64_t n_rows, int n_cols, int64_t V, int64_t row_mul,
                                 int64_t col_mul, int64_t add, int64_t* src_rows, int64_t* seg_start, int32_t* work,
                                 void* stream) {
  MH_REQUIRE(tok && src_rows && seg_start && work, "token_segments: null argument");
  MH_REQUIRE(n_rows > 0 && n_cols > 0 && ldtok >= n_cols && V > 0 && V < 16000, "token_segments: bad shape rows=%ld cols=%d V=%ld",
             (long)n_rows, n_cols, (long)V);
  const int64_t n = n_rows * n_cols;
  MH_REQUIRE(n < (1ll << 31), "token_segments: %ld occurrences (positions are 32-bit)", (long)n);
  const int nb = (int)((n + SEG_CH - 1) / SEG_CH);
  int32_t* tot = work + (int64_t)nb * (V + 1);
  const size_t lds = (size_t)(V + 1) * sizeof(int);
  hipStream_t st = (hipStream_t)stream;
  token_hist_kernel<<<nb, 256, lds, st>>>(tok, ldtok, n, n_cols, (int)V, work);
  token_colscan_kernel<<<(int)((V + 1 + 255) / 256), 256, 0, st>>>(work, nb, (int)V, tot);
  token_place_kernel<<<nb, 256, lds, st>>>(tok, ldtok, n, n_cols, (int)V, work, tot, row_mul, col_mul, add, src_rows, seg_start);
  MH_LAUNCH_CHECK();
  return MH_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n,
                                                            int accumulate) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = src[i];
    if (accumulate) v += to_f(dst[i]);
    dst[i] = from_f<T>(v);
  }
}

extern "C" int mh_cast_from_f32(const float* src, void* dst, int64_t n, int accumulate, int dtype, void* stream) {
  MH_REQUIRE(n > 0, "cast_from_f32: empty");
  DISPATCH_T(dtype, (cast_from_f32_kernel<T><<<grid_for(n, 256, 4096), 256, 0, (hipStream_t)stream>>>(src, (T*)dst, n,
                                                                                                     accumulate)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, int64_t src_ld, T* __restrict__ dst,
                                                        int64_t dst_ld, int64_t M, int D, int accumulate) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> v = ld16(src + m * src_ld + c);
      if (accumulate) {
        Pack<T> o = ld16(dst + m * dst_ld + c);
#pragma unroll
        for (int e = 0; e < N; ++e) v.set(e, v.get(e) + o.get(e));
      }
      st16(dst + m * dst_ld + c, v);
    }
  }
}

extern "C" int mh_copy_rows(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int64_t M, int D, int accumulate,
                            int dtype, void* stream) {
  MH_REQUIRE(M > 0 && D % 8 == 0 && src_ld % 8 == 0 && dst_ld % 8 == 0, "copy_rows: bad shape");
  DISPATCH_T(dtype, (copy_rows_kernel<T><<<grid_for(M, 4, 65536), 256, 0, (hipStream_t)stream>>>(
                        (const T*)src, src_ld, (T*)dst, dst_ld, M, D, accumulate)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// RMSNorm: one wave per row, two passes over the row (second one is an L1/L2 hit)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                          T* __restrict__ y, float* __restrict__ rstd, int64_t M, int D,
                                                          float eps) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const T* xr = x + m * D;
    float ss = 0.f;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> v = ld16(xr + c);
#pragma unroll
      for (int e = 0; e < N; ++e) ss += v.get(e) * v.get(e);
    }
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)D + eps);
    if (lane == 0 && rstd != nullptr) rstd[m] = r;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> v = ld16(xr + c), ww = ld16(w + c), o;
#pragma unroll
      for (int e = 0; e < N; ++e) o.set(e, ww.get(e) * rnd<T>(v.get(e) * r));
      st16(y + m * D + c, o);
    }
  }
}

// Row length D = NCH * 64 * (16 bytes of T): the row lives in registers (one 16-byte load per chunk and lane, all in
// flight together), so x is read exactly once.  Same arithmetic as rmsnorm_fwd_kernel.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_fwd_reg_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                              T* __restrict__ y, float* __restrict__ rstd, int64_t M, int D,
                                                              float eps) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  Pack<T> ww[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) ww[k] = ld16(w + (k * 64 + lane) * N);
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const T* xr = x + m * D;
    Pack<T> v[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) v[k] = ld16(xr + (k * 64 + lane) * N);
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) ss += v[k].get(e) * v[k].get(e);
    ss = wave_sum(ss);
    const float r = rsqrtf(ss / (float)D + eps);
    if (lane == 0 && rstd != nullptr) rstd[m] = r;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      Pack<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e) o.set(e, ww[k].get(e) * rnd<T>(v[k].get(e) * r));
      st16(y + m * D + (k * 64 + lane) * N, o);
    }
  }
}

extern "C" int mh_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int64_t M, int D, float eps,
                              int dtype, void* stream) {
  MH_REQUIRE(M > 0 && D % 8 == 0, "rmsnorm_fwd: bad shape M=%ld D=%d", (long)M, D);
  const int nch = D / (dtype == MH_BF16 ? 512 : 256);  // 16-byte chunks per lane
  const bool reg = nch * (dtype == MH_BF16 ? 512 : 256) == D && (nch == 1 || nch == 2 || nch == 4);
#define MH_RMSF(NCH_)                                                                                     \
  DISPATCH_T(dtype, (rmsnorm_fwd_reg_kernel<T, NCH_><<<grid_for(M, 4, 65536), 256, 0, (hipStream_t)stream>>>( \
                        (const T*)x, (const T*)w, (T*)y, rstd, M, D, eps)))
  if (reg && nch == 1) MH_RMSF(1);
  else if (reg && nch == 2) MH_RMSF(2);
  else if (reg && nch == 4) MH_RMSF(4);
  else
    DISPATCH_T(dtype, (rmsnorm_fwd_kernel<T><<<grid_for(M, 4, 65536), 256, 0, (hipStream_t)stream>>>(
                          (const T*)x, (const T*)w, (T*)y, rstd, M, D, eps)));
#undef MH_RMSF
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// rstd alone (r05: the statistics of an RMSNorm whose scaling rides on the following projection, mh_gemm_*_scaled):
//   parts != NULL: rstd[m] = rsqrt(sum_p parts[p * M + m] / D + eps) -- the per-64-column sums of squares mh_gemm_rowss left
//                  behind (a thread per row, nparts coalesced loads);
//   x != NULL:     the same from the rows themselves (a wave per row; the first layer of a stack, whose input no GEMM of
//                  ours produced).  LlamaRMSNorm.forward, TF:models/llama/modeling_llama.py:62-67: variance in fp32.
// 64 rows per workgroup, the parts of a row spread over the four waves (wave q takes parts q, q + 4, ...: at most a handful of
// independent loads per thread, 1024 workgroups for 65536 rows) and folded through LDS: the first form -- a thread per row walking
// all 16 parts, 256 workgroups -- took 10 us per launch at 65536 rows, a third of what the fold saves (profiles/r05b_block_ab.txt).
__global__ __launch_bounds__(256) void rstd_from_parts_kernel(const float* __restrict__ parts, int nparts, int64_t M, float inv_d,
                                                              float eps, float* __restrict__ rstd) {
  __shared__ float acc[4][64];
  const int r = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t m = (int64_t)blockIdx.x * 64 + r;
  float ss = 0.f;
  if (m < M) {
#pragma unroll 4
    for (int p = q; p < nparts; p += 4) ss += parts[(int64_t)p * M + m];
  }
  acc[q][r] = ss;
  __syncthreads();
  if (q == 0 && m < M) rstd[m] = rsqrtf((acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r]) * inv_d + eps);
}
template <typename T>
__global__ __launch_bounds__(256) void row_rstd_kernel(const T* __restrict__ x, int64_t ldx, int64_t M, int D, float eps,
                                                       float* __restrict__ rstd) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63;
  for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (int64_t)gridDim.x * 4) {
    const T* xr = x + m * ldx;
    float ss = 0.f;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> v = ld16(xr + c);
#pragma unroll
      for (int e = 0; e < N; ++e) ss += v.get(e) * v.get(e);
    }
    ss = wave_sum(ss);
    if (lane == 0) rstd[m] = rsqrtf(ss / (float)D + eps);
  }
}
extern "C" int mh_row_rstd(const void* x, int64_t ldx, const float* parts, int nparts, int64_t M, int D, float eps, float* rstd,
                           int dtype, void* stream) {
  MH_REQUIRE(M > 0 && D > 0 && rstd != nullptr && ((x != nullptr) != (parts != nullptr)), "row_rstd: give x OR parts (M=%ld D=%d)", (long)M, D);
  if (parts != nullptr) {
    MH_REQUIRE(nparts > 0 && (int64_t)nparts * 64 == D, "row_rstd: nparts = %d must be D / 64 (D = %d): one sum of squares per 64 columns", nparts, D);
    rstd_from_parts_kernel<<<(unsigned)((M + 63) / 64), 256, 0, (hipStream_t)stream>>>(parts, nparts, M, 1.0f / (float)D, eps, rstd);
  } else {
    MH_REQUIRE(D % 8 == 0 && ldx % 8 == 0 && ldx >= D && ((uintptr_t)x & 15) == 0, "row_rstd: rows must be 16-byte aligned");
    DISPATCH_T(dtype, (row_rstd_kernel<T><<<grid_for(M, 4, 65536), 256, 0, (hipStream_t)stream>>>((const T*)x, ldx, M, D, eps, rstd)));
  }
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// backward.  g = dy*w, xhat = x*rstd, dx = rstd*(g - xhat*mean(g*xhat)) (+dres); dw += dy*xhat per column.
// Each wave walks rows with a fixed stride, so a lane always owns the same columns and keeps its dw
// partial sums in LDS-free registers via a [D] LDS accumulator per wave (D <= 8192).
constexpr int RMS_BWD_MAX_BLOCKS = 1024;
extern "C" int mh_rmsnorm_bwd_blocks(int64_t M) {
  int64_t g = (M + 3) / 4;
  return (int)(g < RMS_BWD_MAX_BLOCKS ? (g < 1 ? 1 : g) : RMS_BWD_MAX_BLOCKS);
}

template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                          const float* __restrict__ rstd, const T* __restrict__ dy,
                                                          const T* dres, T* dx, float* __restrict__ dw_partial,
                                                          int64_t M, int D) {
  constexpr int N = Pack<T>::N;
  extern __shared__ __attribute__((aligned(16))) float dw_acc[];  // [4][D]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* mine = dw_acc + wv * D;
  for (int c = lane; c < D; c += 64) mine[c] = 0.f;
  __syncthreads();
  // (each lane only ever touches its own columns of `mine`, no barrier needed until the final reduce)
  for (int64_t m = (int64_t)blockIdx.x * 4 + wv; m < M; m += (int64_t)gridDim.x * 4) {
    const T* xr = x + m * D;
    const T* gr = dy + m * D;
    const float r = rstd[m];
    float dot = 0.f;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> xv = ld16(xr + c), gv = ld16(gr + c), wv_ = ld16(w + c);
#pragma unroll
      for (int e = 0; e < N; ++e) dot += gv.get(e) * wv_.get(e) * (xv.get(e) * r);
    }
    dot = wave_sum(dot) / (float)D;
    for (int c = lane * N; c < D; c += 64 * N) {
      Pack<T> xv = ld16(xr + c), gv = ld16(gr + c), wv_ = ld16(w + c), o;
      Pack<T> rv;
      if (dres != nullptr) rv = ld16(dres + m * D + c);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float xh = xv.get(e) * r;
        float d = r * (gv.get(e) * wv_.get(e) - xh * dot);
        if (dres != nullptr) d += rv.get(e);
        o.set(e, d);
        mine[c + e] += gv.get(e) * rnd<T>(xh);
      }
      st16(dx + m * D + c, o);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256)
    dw_partial[(int64_t)blockIdx.x * D + c] = dw_acc[c] + dw_acc[D + c] + dw_acc[2 * D + c] + dw_acc[3 * D + c];
}

// Register-resident variant (D = NCH * 64 * 16 bytes): every operand is read once, and the weight gradient of the columns a
// lane owns accumulates in registers over all the rows its wave walks; LDS only for the final 4-wave reduction.
// NT (r05): non-temporal loads and stores when an operand is larger than the Infinity Cache can keep between kernels (the token-level
// stack: 262144 rows x 1024 = 537 MB per operand: 430-450 -> 404-419 us per launch); at the event-level size (67 MB per operand, which the
// preceding kernel has just left in the 256 MiB cache) the same hints are 6 % SLOWER, so the launcher picks by size.  Same bits.
template <typename T, int NCH, bool NT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_reg_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                              const float* __restrict__ rstd, const T* __restrict__ dy,
                                                              const T* dres, T* dx, float* __restrict__ dw_partial,
                                                              int64_t M, int D) {
  constexpr int N = Pack<T>::N;
  extern __shared__ __attribute__((aligned(16))) float dw_acc[];  // [4][D]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  Pack<T> ww[NCH];
  float dwr[NCH][N];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    ww[k] = ld16(w + (k * 64 + lane) * N);
#pragma unroll
    for (int e = 0; e < N; ++e) dwr[k][e] = 0.f;
  }
  for (int64_t m = (int64_t)blockIdx.x * 4 + wv; m < M; m += (int64_t)gridDim.x * 4) {
    const float r = rstd[m];
    Pack<T> xv[NCH], gv[NCH], rv[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const int64_t c = m * D + (k * 64 + lane) * N;
      if constexpr (NT) {
        xv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(xv[k].v)*>(x + c));
        gv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(gv[k].v)*>(dy + c));
        if (dres != nullptr) rv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(rv[k].v)*>(dres + c));
      } else {
        xv[k] = ld16(x + c);
        gv[k] = ld16(dy + c);
        if (dres != nullptr) rv[k] = ld16(dres + c);
      }
    }
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int e = 0; e < N; ++e) dot += gv[k].get(e) * ww[k].get(e) * (xv[k].get(e) * r);
    dot = wave_sum(dot) / (float)D;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      Pack<T> o;
#pragma unroll
      for (int e = 0; e < N; ++e) {
        const float xh = xv[k].get(e) * r;
        float d = r * (gv[k].get(e) * ww[k].get(e) - xh * dot);
        if (dres != nullptr) d += rv[k].get(e);
        o.set(e, d);
        dwr[k][e] += gv[k].get(e) * rnd<T>(xh);
      }
      if constexpr (NT) __builtin_nontemporal_store(o.v, reinterpret_cast<decltype(o.v)*>(dx + m * D + (k * 64 + lane) * N));
      else st16(dx + m * D + (k * 64 + lane) * N, o);
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int e = 0; e < N; ++e) dw_acc[wv * D + (k * 64 + lane) * N + e] = dwr[k][e];
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256)
    dw_partial[(int64_t)blockIdx.x * D + c] = dw_acc[c] + dw_acc[D + c] + dw_acc[2 * D + c] + dw_acc[3 * D + c];
}

extern "C" int mh_rmsnorm_bwd(const void* x, const void* w, const float* rstd, const void* dy, const void* dres,
                              void* dx, float* dw_partial, int64_t M, int D, int dtype, void* stream) {
  MH_REQUIRE(M > 0 && D % 8 == 0 && D <= 8192, "rmsnorm_bwd: bad shape M=%ld D=%d", (long)M, D);
  const int blocks = mh_rmsnorm_bwd_blocks(M);
  const size_t shm = (size_t)4 * D * sizeof(float);
  const int nch = D / (dtype == MH_BF16 ? 512 : 256);  // 16-byte chunks per lane
  const bool reg = nch * (dtype == MH_BF16 ? 512 : 256) == D && (nch == 1 || nch == 2 || nch == 4);
  const bool nt = M * (int64_t)D * (dtype == MH_BF16 ? 2 : 4) > (int64_t(192) << 20);  // one operand against the 256 MiB Infinity Cache
#define MH_RMSB(NCH_)                                                                                        \
  do {                                                                                                      \
    if (nt)                                                                                                 \
      DISPATCH_T(dtype, (rmsnorm_bwd_reg_kernel<T, NCH_, true><<<blocks, 256, shm, (hipStream_t)stream>>>(    \
                            (const T*)x, (const T*)w, rstd, (const T*)dy, (const T*)dres, (T*)dx, dw_partial, M, D))); \
    else                                                                                                    \
      DISPATCH_T(dtype, (rmsnorm_bwd_reg_kernel<T, NCH_, false><<<blocks, 256, shm, (hipStream_t)stream>>>(   \
                            (const T*)x, (const T*)w, rstd, (const T*)dy, (const T*)dres, (T*)dx, dw_partial, M, D))); \
  } while (0)
  if (reg && nch == 1) MH_RMSB(1);
  else if (reg && nch == 2) MH_RMSB(2);
  else if (reg && nch == 4) MH_RMSB(4);
  else
    DISPATCH_T(dtype, (rmsnorm_bwd_kernel<T><<<blocks, 256, shm, (hipStream_t)stream>>>(
                          (const T*)x, (const T*)w, rstd, (const T*)dy, (const T*)dres, (T*)dx, dw_partial, M, D)));
#undef MH_RMSB
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---- the backward of a FOLDED RMSNorm (r06; engine.layer_backward_folded) -----------------------------------------------------------
// Forward: z = x W'^T with W' = W (.) w, y = rstd (.) z (LlamaRMSNorm + nn.Linear, TF:models/llama/modeling_llama.py:62-67).  The
// producers of d y store d z = rstd (.) d y, the dgrad on the folded weights gives t = d z W' (= rstd (.) (d h (.) w) of the
// unfolded graph), and what is left of the norm's backward is   dx = t - x (rstd^2 / D) rowdot(t, x) + dres   -- no weight vector,
// no weight-gradient column sums (those come out of the weight-gradient reduction: mh_gemm_splitk_reduce_fold).  One wave per
// row, the row in registers (D = NCH * 64 * 16 bytes) or two passes; non-temporal above the Infinity Cache size as mh_rmsnorm_bwd.
template <typename T, int NCH, bool NT>
__global__ __launch_bounds__(256) void rmsnorm_bwd_folded_kernel(const T* __restrict__ x, const float* __restrict__ rstd,
                                                                 const T* __restrict__ t, const T* dres, T* dx, int64_t M, int D) {
  constexpr int N = Pack<T>::N;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int64_t m = (int64_t)blockIdx.x * 4 + wv; m < M; m += (int64_t)gridDim.x * 4) {
    const float r = rstd[m];
    if constexpr (NCH > 0) {
      Pack<T> xv[NCH], tv[NCH], rv[NCH];
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        const int64_t c = m * D + (k * 64 + lane) * N;
        if constexpr (NT) {
          xv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(xv[k].v)*>(x + c));
          tv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(tv[k].v)*>(t + c));
          if (dres != nullptr) rv[k].v = __builtin_nontemporal_load(reinterpret_cast<const decltype(rv[k].v)*>(dres + c));
        } else {
          xv[k] = ld16(x + c);
          tv[k] = ld16(t + c);
          if (dres != nullptr) rv[k] = ld16(dres + c);
        }
      }
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int e = 0; e < N; ++e) dot += tv[k].get(e) * xv[k].get(e);
      const float cf = r * r * wave_sum(dot) / (float)D;
#pragma unroll
      for (int k = 0; k < NCH; ++k) {
        Pack<T> o;
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float d = tv[k].get(e) - xv[k].get(e) * cf;
          if (dres != nullptr) d += rv[k].get(e);
          o.set(e, d);
        }
        if constexpr (NT) __builtin_nontemporal_store(o.v, reinterpret_cast<decltype(o.v)*>(dx + m * D + (k * 64 + lane) * N));
        else st16(dx + m * D + (k * 64 + lane) * N, o);
      }
    } else {
      float dot = 0.f;
      for (int c = lane * N; c < D; c += 64 * N) {
        Pack<T> xv = ld16(x + m * D + c), tv = ld16(t + m * D + c);
#pragma unroll
        for (int e = 0; e < N; ++e) dot += tv.get(e) * xv.get(e);
      }
      const float cf = r * r * wave_sum(dot) / (float)D;
      for (int c = lane * N; c < D; c += 64 * N) {
        Pack<T> xv = ld16(x + m * D + c), tv = ld16(t + m * D + c), o, rv;
        if (dres != nullptr) rv = ld16(dres + m * D + c);
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float d = tv.get(e) - xv.get(e) * cf;
          if (dres != nullptr) d += rv.get(e);
          o.set(e, d);
        }
        st16(dx + m * D + c, o);
      }
    }
  }
}

extern "C" int mh_rmsnorm_bwd_folded(const void* x, const float* rstd, const void* t, const void* dres, void* dx, int64_t M, int D,
                                     int dtype, void* stream) {
  MH_REQUIRE(M > 0 && D % 8 == 0 && x != nullptr && rstd != nullptr && t != nullptr && dx != nullptr, "rmsnorm_bwd_folded: bad arguments M=%ld D=%d", (long)M, D);
  const int blocks = (int)grid_for(M, 4, 65536);
  const int nch = D / (dtype == MH_BF16 ? 512 : 256);
  const bool reg = nch * (dtype == MH_BF16 ? 512 : 256) == D && (nch == 1 || nch == 2 || nch == 4);
  const bool nt = M * (int64_t)D * (dtype == MH_BF16 ? 2 : 4) > (int64_t(192) << 20);
#define MH_RMSBF(NCH_)                                                                                                    \
  do {                                                                                                                   \
    if (nt)                                                                                                              \
      DISPATCH_T(dtype, (rmsnorm_bwd_folded_kernel<T, NCH_, true><<<blocks, 256, 0, (hipStream_t)stream>>>(                \
                            (const T*)x, rstd, (const T*)t, (const T*)dres, (T*)dx, M, D)));                              \
    else                                                                                                                 \
      DISPATCH_T(dtype, (rmsnorm_bwd_folded_kernel<T, NCH_, false><<<blocks, 256, 0, (hipStream_t)stream>>>(               \
                            (const T*)x, rstd, (const T*)t, (const T*)dres, (T*)dx, M, D)));                              \
  } while (0)
  if (reg && nch == 1) MH_RMSBF(1);
  else if (reg && nch == 2) MH_RMSBF(2);
  else if (reg && nch == 4) MH_RMSBF(4);
  else MH_RMSBF(0);
#undef MH_RMSBF
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// out[n, k] = W[n, k] * w[k]: the norm weight folded into the projection that follows it (engine.fold_norm_weights; rounded once,
// from the fp32 product, as the torch spelling `(W.float() * w.float()).to(bf16)` it replaces)
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_kernel(const T* __restrict__ W, int64_t ldw, const T* __restrict__ w, T* __restrict__ out,
                                                         int64_t ldo, int64_t Nr, int K) {
  constexpr int N = Pack<T>::N;
  const int cpr = K / N;
  const int64_t total = Nr * cpr;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int64_t n = it / cpr;
    const int c = (int)(it - n * cpr) * N;
    Pack<T> a = ld16(W + n * ldw + c), b = ld16(w + c), o;
#pragma unroll
    for (int e = 0; e < N; ++e) o.set(e, a.get(e) * b.get(e));
    st16(out + n * ldo + c, o);
  }
}
// the same for a LIST of matrices in one launch: jobs[j] = {W, w, out, rows} (device pointers as 64-bit integers, contiguous
// [rows, K] matrices); blockIdx.y = job.  The training step re-derives thirty folded matrices after every optimizer step.
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_batched_kernel(const int64_t* __restrict__ jobs, int K) {
  constexpr int N = Pack<T>::N;
  const int64_t* jb = jobs + 4 * (int64_t)blockIdx.y;
  const T* W = reinterpret_cast<const T*>(jb[0]);
  const T* w = reinterpret_cast<const T*>(jb[1]);
  T* out = reinterpret_cast<T*>(jb[2]);
  const int cpr = K / N;
  const int64_t total = jb[3] * cpr;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int c = (int)(it % cpr) * N;
    Pack<T> a = ld16(W + it * N), b = ld16(w + c), o;
#pragma unroll
    for (int e = 0; e < N; ++e) o.set(e, a.get(e) * b.get(e));
    st16(out + it * N, o);
  }
}
extern "C" int mh_scale_cols_batched(const int64_t* jobs, int njobs, int K, int dtype, void* stream) {
  MH_REQUIRE(jobs != nullptr && njobs > 0 && njobs < 65536 && K > 0 && K % 8 == 0, "scale_cols_batched: bad arguments");
  DISPATCH_T(dtype, (scale_cols_batched_kernel<T><<<dim3(1024, (unsigned)njobs), 256, 0, (hipStream_t)stream>>>(jobs, K)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}
extern "C" int mh_scale_cols(const void* W, int64_t ldw, const void* w, void* out, int64_t ldo, int64_t Nr, int K, int dtype, void* stream) {
  MH_REQUIRE(Nr > 0 && K > 0 && K % 8 == 0 && ldw % 8 == 0 && ldo % 8 == 0 && ldw >= K && ldo >= K, "scale_cols: bad shape");
  MH_REQUIRE((((uintptr_t)W | (uintptr_t)w | (uintptr_t)out) & 15) == 0, "scale_cols: 16-byte alignment");
  DISPATCH_T(dtype, (scale_cols_kernel<T><<<grid_for(Nr * (K / 4), 256, 16384), 256, 0, (hipStream_t)stream>>>(
                        (const T*)W, ldw, (const T*)w, (T*)out, ldo, Nr, K)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// Column sums of the [nblk, D] fp32 partials, deterministic, in two stages so that more than D/256 blocks
// work: stage A folds each of NSPLIT row ranges into the range's first row (in place: a block only touches
// its own rows x 64 columns), stage B adds the NSPLIT surviving rows.
constexpr int COLSUM_NSPLIT = 32;

__global__ __launch_bounds__(256) void colsum_stage_a_kernel(float* __restrict__ partial, int64_t nblk, int D, int64_t rows_per) {
  __shared__ float sh[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  int64_t r1 = r0 + rows_per;
  if (r1 > nblk) r1 = nblk;
  float s = 0.f;
  if (c < D)
    for (int64_t r = r0 + ty; r < r1; r += 4) s += partial[r * D + c];
  sh[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < D && r0 < nblk) partial[r0 * D + c] = sh[0][tx] + sh[1][tx] + sh[2][tx] + sh[3][tx];
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_stage_b_kernel(const float* __restrict__ partial, int64_t nblk, int64_t rows_per,
                                                             T* __restrict__ out, int D, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  float s = 0.f;
  for (int64_t r = 0; r < nblk; r += rows_per) s += partial[r * D + c];
  if (accumulate) s += to_f(out[c]);
  out[c] = from_f<T>(s);
}

extern "C" int mh_colsum(float* partial, int64_t nblk, void* out, int D, int accumulate, int dtype, void* stream) {
  MH_REQUIRE(nblk > 0 && D > 0, "colsum: bad shape");
  hipStream_t st = (hipStream_t)stream;
  int64_t rows_per = 1;
  if (nblk > COLSUM_NSPLIT) {
    rows_per = (nblk + COLSUM_NSPLIT - 1) / COLSUM_NSPLIT;
    dim3 grid((D + 63) / 64, (unsigned)((nblk + rows_per - 1) / rows_per));
    colsum_stage_a_kernel<<<grid, 256, 0, st>>>(partial, nblk, D, rows_per);
    MH_LAUNCH_CHECK();
  }
  DISPATCH_T(dtype, (colsum_stage_b_kernel<T><<<(D + 255) / 256, 256, 0, st>>>(partial, nblk, rows_per, (T*)out, D, accumulate)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// RoPE, in place on the q and k thirds of qkv[M, 3*H*hd].  Pair (i, i+hd/2); cos/sin tables are fp32
// and are rounded to the activation dtype before use (the reference casts them, modeling_llama.py:126).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(T* __restrict__ qkv, const float* __restrict__ cos_t,
                                                   const float* __restrict__ sin_t, int64_t M, int64_t S, int64_t pos0,
                                                   int H, int hd, float dir) {
  constexpr int N = Pack<T>::N;
  const int half = hd / 2;
  const int cph = half / N;                 // 16-byte chunks per half head
  const int64_t per_row = (int64_t)2 * H * cph;  // q and k
  const int64_t total = M * per_row;
  const int64_t D3 = (int64_t)3 * H * hd;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int64_t m = it / per_row;
    int rem = (int)(it - m * per_row);
    const int ch = rem % cph;
    rem /= cph;
    const int h = rem % H;
    const int part = rem / H;  // 0 = q, 1 = k
    const int64_t pos = pos0 + (m % S);
    T* p1 = qkv + m * D3 + (int64_t)part * H * hd + (int64_t)h * hd + ch * N;
    T* p2 = p1 + half;
    Pack<T> a = ld16(p1), b = ld16(p2), oa, ob;
    const float* cp = cos_t + pos * half + ch * N;
    const float* sp = sin_t + pos * half + ch * N;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float c = rnd<T>(cp[e]), s = dir * rnd<T>(sp[e]);
      const float x1 = a.get(e), x2 = b.get(e);
      oa.set(e, __builtin_fmaf(x1, c, -(x2 * s)));  // (spelled out: mh_gemm_rope's epilogue forms the same products)
      ob.set(e, __builtin_fmaf(x2, c, x1 * s));
    }
    st16(p1, oa);
    st16(p2, ob);
  }
}

extern "C" int mh_rope(void* qkv, const float* cos_t, const float* sin_t, int64_t M, int64_t S, int64_t pos0, int H,
                       int hd, int dir, int dtype, void* stream) {
  MH_REQUIRE(M > 0 && S > 0 && hd % 16 == 0, "rope: bad shape");
  const int64_t total = M * 2 * H * (hd / 2 / (dtype == MH_BF16 ? 8 : 4));
  DISPATCH_T(dtype, (rope_kernel<T><<<grid_for(total, 256, 16384), 256, 0, (hipStream_t)stream>>>(
                        (T*)qkv, cos_t, sin_t, M, S, pos0, H, hd, dir >= 0 ? 1.f : -1.f)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// SwiGLU
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const T* __restrict__ gu, T* __restrict__ a, int64_t M, int I) {
  constexpr int N = Pack<T>::N;
  const int cpr = I / N;
  const int64_t total = M * cpr;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int64_t m = it / cpr;
    const int c = (int)(it - m * cpr) * N;
    Pack<T> g = ld16(gu + m * 2 * I + c), u = ld16(gu + m * 2 * I + I + c), o;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float gv = g.get(e);
      const float s = rnd<T>(mh_silu(gv));
      o.set(e, s * u.get(e));
    }
    st16(a + m * I + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const T* __restrict__ gu, const T* __restrict__ da,
                                                         T* __restrict__ dgu, int64_t M, int I) {
  constexpr int N = Pack<T>::N;
  const int cpr = I / N;
  const int64_t total = M * cpr;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int64_t m = it / cpr;
    const int c = (int)(it - m * cpr) * N;
    Pack<T> g = ld16(gu + m * 2 * I + c), u = ld16(gu + m * 2 * I + I + c), d = ld16(da + m * I + c), og, ou;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      const float gv = g.get(e), dv = d.get(e);
      const float sig = mh_sigmoid(gv);
      const float silu = gv * sig;
      og.set(e, dv * u.get(e) * (sig * (1.f + gv * (1.f - sig))));
      ou.set(e, dv * silu);
    }
    st16(dgu + m * 2 * I + c, og);
    st16(dgu + m * 2 * I + I + c, ou);
  }
}

extern "C" int mh_swiglu_fwd(const void* gu, void* a, int64_t M, int I, int dtype, void* stream) {
  MH_REQUIRE(M > 0 && I % 8 == 0, "swiglu_fwd: bad shape");
  DISPATCH_T(dtype, (swiglu_fwd_kernel<T><<<grid_for(M * (I / 4), 256, 16384), 256, 0, (hipStream_t)stream>>>(
                        (const T*)gu, (T*)a, M, I)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

extern "C" int mh_swiglu_bwd(const void* gu, const void* da, void* dgu, int64_t M, int I, int dtype, void* stream) {
  MH_REQUIRE(M > 0 && I % 8 == 0, "swiglu_bwd: bad shape");
  DISPATCH_T(dtype, (swiglu_bwd_kernel<T><<<grid_for(M * (I / 4), 256, 16384), 256, 0, (hipStream_t)stream>>>(
                        (const T*)gu, (const T*)da, (T*)dgu, M, I)));
  MH_LAUNCH_CHECK();
  return MH_OK;
}

// ---------------------------------------------------------------------------------------------------
// Batch assembly on the device (MidiDataset.__getitem__ slicing + collate_fn, train.py:69-90): the pre-tokenised
// corpus lives in HBM as int16 octets [N, T]; a batch is B windows (first event, length) of it, widened to int64 and
// padded to the longest window with pad_id.  One launch, one thread per output token.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void collate_windows_kernel(const int16_t* __restrict__ tokens,
                                                              const int64_t* __restrict__ win_start,
                                                              const int64_t* __restrict__ win_len, int64_t* __restrict__ out,
                                                              int64_t B, int64_t L, int T, int64_t pad_id) {
  const int64_t total = B * L * T;
  for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < total; it += (int64_t)gridDim.x * 256) {
    const int j = (int)(it % T);
    const int64_t r = it / T;
    const int64_t t = r % L, b = r / L;
    out[it] = (t < win_len[b]) ? (int64_t)tokens[(win_start[b] + t) * T + j] : pad_id;
  }
}

extern "C" int mh_collate_windows(const int16_t* tokens, int64_t n_events, const int64_t* win_start, const int64_t* win_len,
                                  int64_t* out, int64_t B, int64_t L, int T, int64_t pad_id, void* stream) {
  MH_REQUIRE(B > 0 && L > 0 && T > 0 && n_events > 0, "collate_windows: bad shape B=%ld L=%ld T=%d", (long)B, (long)L, T);
  collate_windows_kernel<<<grid_for(B * L * T, 256, 16384), 256, 0, (hipStream_t)stream>>>(tokens, win_start, win_len, out, B, L, T,
                                                                                       pad_id);
  MH_LAUNCH_CHECK();
  return MH_OK;
}
