// Hardware-semantics probe for gfx950 (run on the GPU box, output kept under profiles/ as evidence):
//   1. fragment layouts of the MFMA shapes the kernels use (A/B lane->(row,k), C/D lane,reg->(row,col))
//   2. ds_read_b64_tr_b16 (LDS transpose read): which (lane, element) each output comes from
//   3. v_permlane32_swap / v_permlane16_swap half/row exchanges (wave reductions without LDS)
//   4. global_load_lds_dwordx4 destination = wave-uniform base + lane*16
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// D = A*B with A[i][k] = (k==ka ? i+1 : 0), B[k][j] = (k==kb ? j+1 : 0) for the lane-local guess of k.
// We instead drive operands per lane: lane l sets a[e] = (e==0 ? alane(l) : 0), b[e] likewise, and the host
// decodes which (lane_a, lane_b) pairs met in each output register.
__global__ void probe_mfma16(float* out /*[64][4]*/, int mode) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
  // mode 0: only k-group 0 lanes (l<16) carry data in element 0: a = 1+l, b = 1+l  -> D[i][j] = (1+i)(1+j)
  // mode 1: only lanes of group g=l>>4 == 1, element 3
  if (mode == 0 && l < 16) { a[0] = (__bf16)(float)(1 + l); b[0] = (__bf16)(float)(1 + l); }
  if (mode == 1 && (l >> 4) == 1) { a[3] = (__bf16)(float)(1 + (l & 15)); b[3] = (__bf16)(float)(1 + (l & 15)); }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

__global__ void probe_mfma32(float* out /*[64][16]*/) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
  if (l < 32) { a[0] = (__bf16)(float)(1 + l); b[0] = (__bf16)(float)(1 + l); }  // values <= 32 exact in bf16
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}

__global__ void probe_mfma16_f32(float* out) {
  const int l = threadIdx.x;
  float a = (l < 16) ? (float)(1 + l) : 0.f, b = (l < 16) ? (float)(1 + l) : 0.f;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

// cross-k check: do A's (group g, elem e) and B's (group g, elem e) pair up? put a single 1 at (lane la, elem ea)
// in A (row la&15) and a single 1 at (lane lb, elem eb) in B; D nonzero iff same k slot.
__global__ void probe_kslot(float* out /*[1]*/, int la, int ea, int lb, int eb) {
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)0.f; b[e] = (__bf16)0.f; }
  if (l == la) a[ea] = (__bf16)1.f;
  if (l == lb) b[eb] = (__bf16)1.f;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  float s = c[0] + c[1] + c[2] + c[3];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (l == 0) out[0] = s;
}

__global__ void probe_tr(short* out /*[64][4]*/) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  const int l = threadIdx.x;
  for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = t[e];
}

__global__ void probe_swap(int* out /*[4][64]*/) {
  const int l = threadIdx.x;
  auto r32 = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
  out[l] = r32[0];
  out[64 + l] = r32[1];
  out[128 + l] = r16[0];
  out[192 + l] = r16[1];
}

__global__ void probe_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[512];
  const int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = -1;
  __syncthreads();
  // lane l fetches 16 bytes from src + (63-l)*4 ints  -> LDS position l
  __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(src + (63 - l) * 4),
                                   (__attribute__((address_space(3))) void*)(lds + 16), 16, 0, 0);
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  float* d; CK(hipMalloc(&d, 64 * 16 * 4));
  float h[64 * 16];
  for (int mode = 0; mode < 2; ++mode) {
    probe_mfma16<<<1, 64>>>(d, mode); CK(hipMemcpy(h, d, 64 * 4 * 4, hipMemcpyDeviceToHost));
    printf("== mfma_f32_16x16x32_bf16 mode %d: lane reg -> value (decode (row+1)*(col+1); expect col=lane&15,row=4*(lane>>4)+reg)\n", mode);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      float want = (float)((4 * (l >> 4) + r + 1) * ((l & 15) + 1));
      if (h[l * 4 + r] != want) { if (ok) printf("  MISMATCH lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); ok = 0; }
    }
    printf("  C/D layout (col=lane&15,row=4*(lane>>4)+reg) and A/B (row=lane&15): %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    if (!ok) { for (int l = 0; l < 64; ++l) { printf("  l%02d:", l); for (int r = 0; r < 4; ++r) printf(" %g", h[l * 4 + r]); printf("\n"); } }
  }
  {
    probe_mfma32<<<1, 64>>>(d); CK(hipMemcpy(h, d, 64 * 16 * 4, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
      int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
      float want = (float)((row + 1) * (col + 1));
      if (h[l * 16 + r] != want) { if (ok) printf("  MISMATCH lane %d reg %d: got %g want %g\n", l, r, h[l * 16 + r], want); ok = 0; }
    }
    printf("== mfma_f32_32x32x16_bf16 C/D (col=lane&31,row=(r&3)+8*(r>>2)+4*(lane>>5)), A/B row=lane&31: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    if (!ok) { for (int l = 0; l < 64; ++l) { printf("  l%02d:", l); for (int r = 0; r < 16; ++r) printf(" %g", h[l * 16 + r]); printf("\n"); } }
  }
  {
    probe_mfma16_f32<<<1, 64>>>(d); CK(hipMemcpy(h, d, 64 * 4 * 4, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      float want = (float)((4 * (l >> 4) + r + 1) * ((l & 15) + 1));
      if (h[l * 4 + r] != want) ok = 0;
    }
    printf("== mfma_f32_16x16x4f32 layout: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  }
  {
    printf("== k-slot pairing of 16x16x32 (A lane/elem vs B lane/elem -> meets?)\n");
    int cases[][4] = {{0, 0, 0, 0}, {0, 1, 0, 1}, {0, 0, 0, 1}, {16, 0, 16, 0}, {16, 0, 0, 0}, {48, 7, 48, 7}, {17, 3, 18, 3}, {17, 3, 33, 3}};
    for (auto& c : cases) {
      probe_kslot<<<1, 64>>>(d, c[0], c[1], c[2], c[3]); CK(hipMemcpy(h, d, 4, hipMemcpyDeviceToHost));
      printf("  A(l=%d,e=%d) x B(l=%d,e=%d): %g\n", c[0], c[1], c[2], c[3], h[0]);
    }
  }
  {
    short* ds; CK(hipMalloc(&ds, 64 * 4 * 2)); short hs[256];
    probe_tr<<<1, 64>>>(ds); CK(hipMemcpy(hs, ds, 512, hipMemcpyDeviceToHost));
    printf("== ds_read_b64_tr_b16, lane-linear addresses (lane p supplies shorts 4p..4p+3): out[lane][e] = source short index\n");
    for (int l = 0; l < 64; ++l) { printf("  l%02d: %d %d %d %d\n", l, hs[l * 4], hs[l * 4 + 1], hs[l * 4 + 2], hs[l * 4 + 3]); }
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
      int grp = l >> 4, i = l & 15;
      int want = grp * 64 + (4 * e + i / 4) * 4 + (i % 4);  // result[i][e] = M[4e + i/4][i%4] within the 16-lane group
      if (hs[l * 4 + e] != want) ok = 0;
    }
    printf("  hypothesis out[i][e] = in[lane 4e+i/4][elem i%%4] per 16-lane group: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  }
  {
    int* di; CK(hipMalloc(&di, 256 * 4)); int hi[256];
    probe_swap<<<1, 64>>>(di); CK(hipMemcpy(hi, di, 1024, hipMemcpyDeviceToHost));
    const char* nm[4] = {"permlane32_swap r[0] (a=lane,b=100+lane)", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]"};
    for (int k = 0; k < 4; ++k) { printf("== %s:", nm[k]); for (int l = 0; l < 64; ++l) printf(" %d", hi[k * 64 + l]); printf("\n"); }
  }
  {
    int *src, *out; CK(hipMalloc(&src, 256 * 4)); CK(hipMalloc(&out, 512 * 4));
    int hsrc[256], hout[512];
    for (int i = 0; i < 256; ++i) hsrc[i] = i;
    CK(hipMemcpy(src, hsrc, 1024, hipMemcpyHostToDevice));
    probe_glds<<<1, 64>>>(src, out); CK(hipMemcpy(hout, out, 2048, hipMemcpyDeviceToHost));
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (hout[16 + l * 4 + e] != (63 - l) * 4 + e) ok = 0;
    for (int i = 0; i < 16; ++i) if (hout[i] != -1) ok = 0;
    printf("== global_load_lds_dwordx4: LDS dest = base + lane*16, source per lane: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
  }
  return 0;
}
