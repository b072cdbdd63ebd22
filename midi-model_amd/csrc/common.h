// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the midi-model hot path.
// wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/midihip.h"

#define MH_WAVE 64

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

// ---- error plumbing (host) ----------------------------------------------------------------
void mh_set_error(const char* fmt, ...);
#define MH_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      mh_set_error(__VA_ARGS__);         \
      return MH_ERR_ARG;                 \
    }                                    \
  } while (0)
#define MH_LAUNCH_CHECK()                                            \
  do {                                                               \
    hipError_t e_ = hipGetLastError();                               \
    if (e_ != hipSuccess) {                                          \
      mh_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return MH_ERR_LAUNCH;                                          \
    }                                                                \
  } while (0)

// ---- 16-byte vectors of T ---------------------------------------------------------------------
template <typename T> struct Pack;  // 16 bytes of T, convertible to/from fp32 lanes
template <> struct Pack<float> {
  static constexpr int N = 4;
  f32x4 v;
  __device__ float get(int i) const { return v[i]; }
  __device__ void set(int i, float f) { v[i] = f; }
};
template <> struct Pack<bf16> {
  static constexpr int N = 8;
  bf16x8 v;
  __device__ float get(int i) const { return (float)v[i]; }
  __device__ void set(int i, float f) { v[i] = (bf16)f; }
};
template <typename T> __device__ inline Pack<T> ld16(const T* p) { return *reinterpret_cast<const Pack<T>*>(p); }
template <typename T> __device__ inline void st16(T* p, const Pack<T>& v) { *reinterpret_cast<Pack<T>*>(p) = v; }

template <typename T> __device__ inline float to_f(T x) { return (float)x; }
template <typename T> __device__ inline T from_f(float x) { return (T)x; }
// round an fp32 value through T (used where the reference rounds an intermediate to the activation dtype)
template <typename T> __device__ inline float rnd(float x) { return (float)(T)x; }

// ---- SiLU / sigmoid, one spelling for every kernel that needs them --------------------------------
// sigmoid(g) = rcp(1 + exp2(-g log2 e)): v_mul, v_exp, v_add, v_rcp.  The IEEE division the plain C++ expression
// `g / (1.f + __expf(-g))` compiles to is v_div_scale x2, v_rcp, 5 fma/mul, v_div_fmas, v_div_fixup -- eleven VALU issue
// slots per element, which made the SwiGLU epilogues of the projection GEMMs (64 elements per lane and tile) a fifth of the
// tile's time (r02, ISA of gemm_pp256_kernel<.., 2>).  v_rcp_f32 is accurate to 1 ulp; results are rounded to the activation
// dtype right after.  Every kernel uses THESE functions, so fused and unfused paths agree bit for bit.
// g -> -inf gives 0 * -inf = NaN exactly as the division did; large |g| saturate (exp2 overflow -> rcp(inf) = 0).
__device__ inline float mh_sigmoid(float g) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(g * -1.4426950408889634f));
}
__device__ inline float mh_silu(float g) { return g * mh_sigmoid(g); }

// eight at a time, in two-element vectors: the multiplies and the addition become v_pk_mul_f32 / v_pk_add_f32 (one issue slot
// for two elements; hipcc does not pack the scalar spelling across the transcendentals).  IEEE operations either way: the
// same bits as mh_silu.  For the VALU-bound epilogues of the projection GEMMs (no MFMA runs beside them).
typedef __attribute__((ext_vector_type(2))) float mh_f32x2;
__device__ inline void mh_silu8(const float (&g)[8], float (&s)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const mh_f32x2 gv = {g[2 * i], g[2 * i + 1]};
    const mh_f32x2 x = gv * -1.4426950408889634f;
    const mh_f32x2 d = mh_f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])} + 1.f;
    const mh_f32x2 o = gv * mh_f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    s[2 * i] = o[0];
    s[2 * i + 1] = o[1];
  }
}

// SwiGLU backward of eight elements, the expressions of swiglu_bwd_kernel (elementwise.hip) on two-element vectors:
// dg = dv u (sig (1 + g (1 - sig))), du = dv (g sig).  The same IEEE operations in the same order (the contraction
// 1 + g (1 - sig) -> fma is taken by both spellings), 14 issue slots per pair of elements instead of 23.
__device__ inline void mh_dswiglu8(const float (&dv)[8], const float (&g)[8], const float (&u)[8], float (&dg)[8], float (&du)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const mh_f32x2 gv = {g[2 * i], g[2 * i + 1]}, uv = {u[2 * i], u[2 * i + 1]}, dvv = {dv[2 * i], dv[2 * i + 1]};
    const mh_f32x2 x = gv * -1.4426950408889634f;
    const mh_f32x2 d = 1.f + mh_f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
    const mh_f32x2 sig = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const mh_f32x2 silu = gv * sig;
    const mh_f32x2 og = dvv * uv * (sig * (1.f + gv * (1.f - sig)));
    const mh_f32x2 ou = dvv * silu;
    dg[2 * i] = og[0];
    dg[2 * i + 1] = og[1];
    du[2 * i] = ou[0];
    du[2 * i + 1] = ou[1];
  }
}

// eight fp32 -> one bf16x8 as four two-element conversions (v_cvt_pk_bf16_f32 each), and back (shift / mask)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ inline bf16x8 cvt8_bf16(const float (&v)[8]) {
  union {
    bf16x8 v;
    bf16x2 h[4];
  } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.h[i] = __builtin_convertvector(f32x2{v[2 * i], v[2 * i + 1]}, bf16x2);
  return r.v;
}
__device__ inline void expand8_bf16(const bf16x8& b, float (&v)[8]) {
  union {
    bf16x8 v;
    unsigned u[4];
  } r;
  r.v = b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(r.u[i] << 16);
    v[2 * i + 1] = __uint_as_float(r.u[i] & 0xffff0000u);
  }
}

// ---- wave reductions --------------------------------------------------------------------------
// All-lanes reductions.  The butterfly alone leaves every lane with the same value only if no lane's first addition is
// contracted with the multiply that produced its operand (fma(a, b, partner) != fma(a', b', own) in the last bit, seen
// as a 1-ulp difference between wave halves in the sampler), so the result is re-broadcast from one lane.
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// All-lanes sum without touching the LDS pipe: 4 DPP steps inside each 16-lane row (quad swaps, half-row and
// row mirrors), then v_permlane16_swap / v_permlane32_swap fold the rows (gfx950; semantics pinned by
// tools/probe.hip: swap(x,x) returns {even rows|low half, odd rows|high half} replicated, so r[0]+r[1] is the
// xor-16 / xor-32 sum in every lane).  ~10 VALU ops instead of 6 dependent ds_bpermute round trips.
template <int CTRL> __device__ inline float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ inline float wave_sum_fast(float v) {
  v += dpp_move<0xB1>(v);   // quad_perm(1,0,3,2)
  v += dpp_move<0x4E>(v);   // quad_perm(2,3,0,1)
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  const int iv = __float_as_int(v);
  auto a = __builtin_amdgcn_permlane16_swap(iv, iv, false, false);
  v = __int_as_float(a[0]) + __int_as_float(a[1]);
  const int iw = __float_as_int(v);
  auto b = __builtin_amdgcn_permlane32_swap(iw, iw, false, false);
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}

// FOUR all-lanes sums at the price of about one and a half (r06): the two swap levels come first and each folds TWO values --
// permlane32_swap(x0, x1) leaves [x0.lo | x1.lo], [x0.hi | x1.hi], whose sum holds x0's lane-pair sums in lanes 0-31 and x1's in
// lanes 32-63; permlane16_swap does the same to the rows of two such registers -- so four values end up in the four 16-lane rows
// of ONE register, and the four DPP steps inside a row finish all of them at once.  Row r of the result holds (in every lane)
// the total of x0, x2, x1, x3 for r = 0, 1, 2, 3: wave_sum4_get<j> reads value j back as a wave-uniform scalar (v_readlane).
// 3 swaps + 7 adds + 4 readlanes for four values against 4 x (2 swaps + 6 adds) -- the token-level attention backward spent
// 28 % of its VALU instructions in 72 single reductions per (sequence, head).  Summation order differs from wave_sum_fast's.
__device__ inline float wave_sum4(float x0, float x1, float x2, float x3) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_int(x0), __float_as_int(x1), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(x2), __float_as_int(x3), false, false);
  const float y0 = __int_as_float(a[0]) + __int_as_float(a[1]);  // rows 0,1: x0, rows 2,3: x1
  const float y1 = __int_as_float(b[0]) + __int_as_float(b[1]);  // rows 0,1: x2, rows 2,3: x3
  auto c = __builtin_amdgcn_permlane16_swap(__float_as_int(y0), __float_as_int(y1), false, false);
  float v = __int_as_float(c[0]) + __int_as_float(c[1]);         // row 0: x0, row 1: x2, row 2: x1, row 3: x3
  v += dpp_move<0xB1>(v);   // quad_perm(1,0,3,2)
  v += dpp_move<0x4E>(v);   // quad_perm(2,3,0,1)
  v += dpp_move<0x141>(v);  // row_half_mirror
  v += dpp_move<0x140>(v);  // row_mirror
  return v;
}
template <int J> __device__ inline float wave_sum4_get(float v) {
  constexpr int ROW = (J == 0) ? 0 : (J == 1) ? 2 : (J == 2) ? 1 : 3;
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16 * ROW));
}

// max over the wave of 64-bit keys (every lane gets it).  Same DPP / permlane ladder as wave_sum_fast, no LDS round
// trips.  An arg-max with a tie rule is a max over keys (value bits << 32 | rank of the index), see the sampler.
template <int CTRL> __device__ inline uint32_t dpp_move_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ inline uint64_t wave_max_u64_fast(uint64_t k) {
  uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
  auto take = [&](uint32_t ohi, uint32_t olo) {
    const bool better = ohi > hi || (ohi == hi && olo > lo);
    hi = better ? ohi : hi;
    lo = better ? olo : lo;
  };
  take(dpp_move_u<0xB1>(hi), dpp_move_u<0xB1>(lo));
  take(dpp_move_u<0x4E>(hi), dpp_move_u<0x4E>(lo));
  take(dpp_move_u<0x141>(hi), dpp_move_u<0x141>(lo));
  take(dpp_move_u<0x140>(hi), dpp_move_u<0x140>(lo));
  {
    auto a = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    hi = a[0];
    lo = b[0];
    take(a[1], b[1]);
  }
  {
    auto a = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    hi = a[0];
    lo = b[0];
    take(a[1], b[1]);
  }
  return ((uint64_t)hi << 32) | lo;
}

// ---- LDS tile format shared by the MFMA kernels -----------------------------------------------
// A tile is ROWS x 128 bytes (64 bf16 or 32 fp32 along the contraction).  The eight 16-byte chunks of
// a row are XOR-swizzled with ((row>>1)&7) so that the ds_read_b128 fragment reads of both MFMA
// shapes (16x16: lane (i=l&15,g=l>>4) -> row i, chunk 4kk+g; 32x32: lane (i=l&31,hi=l>>5) -> row
// pi(i), chunk 2s+hi) hit 16 distinct 16-byte slots per lane group (MI355X_MICROARCH §LDS).
// The XOR term is the BIT-REVERSED (row >> 1) & 7 (r02): any bijection of it serves the 16-byte fragment reads, and this one
// also puts rows r and r + 2 into different 64-byte halves of the 128-byte bank row, which is what a ds_read_b64_tr_b16 of a
// [4 rows][32 columns] block needs -- with the plain term rows r and r + 2 of such a read shared their 16 banks (2-way
// conflicts on every transpose read of the attention kernels: a third of their LDS-active cycles, profiles/r02_run27_*).
__device__ inline int lds_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 2) & 1) << 1) | ((row >> 3) & 1); }
__device__ inline int lds_tile_off(int row, int chunk) { return row * 128 + ((chunk ^ lds_swz(row)) << 4); }

// Direct global->LDS staging (global_load_lds_dwordx4): one call moves 1 KiB = 8 tile rows per
// wave.  The LDS image is lane-linear (base + lane*16), so the swizzle goes on the SOURCE chunk.
// `lds_rows8` is the wave-uniform LDS address of the 8-row group; `src` the lane's global address.
__device__ inline void glds16(const void* src, void* lds_rows8) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)lds_rows8, 16, 0, 0);
}

// Tile rasterisation for the GEMMs.  An XCD (private 4 MiB L2) runs a contiguous range of tile indices, and the
// 32-64 tiles resident on it at one time are consecutive indices; walking GM tile-rows column by column makes
// those tiles share GM A-panels and a few B-panels instead of one A-panel and EVERY B-panel (r01 PMC: the
// row-major order re-streamed the whole 16 MiB weight for every 128-row panel of a [32768x8192x1024] GEMM,
// 4.2 GB of L2 misses, L2 hit rate 49 %).
__device__ inline void gemm_tile_of(int idx, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
  const int width = GM * tiles_n;
  const int gid = idx / width;
  const int first = gid * GM;
  const int gsize = (tiles_m - first < GM) ? tiles_m - first : GM;
  const int rem = idx - gid * width;
  tn = rem / gsize;
  tm = first + (rem - tn * gsize);
}

// 32x32 MFMA row permutation: swap bits 2 and 3 of the row index.  Feeding operand row i from
// tile row pi(i) makes accumulator registers 8t..8t+7 of lane half `hi` correspond to tile rows
// 16t+8hi..16t+8hi+7, i.e. a contiguous 8-element run that is directly the next MFMA's operand.
__device__ inline int pi32(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }
