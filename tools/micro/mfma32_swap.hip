// Chain check for the head_dim-40 attention backward on gfx950 (round 6): S^T = K Q^T on v_mfma_f32_32x32x16_bf16 (reduction padded to 48
// instead of 64), the 32x32 result (lane l: column q = l & 31, rows 8 (a / 4) + 4 (l >> 5) + a % 4) packed to bf16 and turned into TWO
// 16x16x32 B operands (q tiles 0 / 1) by v_permlane16_swap, then dQ^T = K^T S^T on v_mfma_f32_16x16x32_bf16 with the K column fragments read
// by ds_read_b64_tr_b16 in the k-slot order the swap produces.  Prints the mismatch count of dQ = S K against the host (exact small integers).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma32_swap.hip -o /tmp/mfma32_swap && /tmp/mfma32_swap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) << 1; }
__device__ __forceinline__ int tile_off(int row, int col) { return row * 128 + ((((col >> 3) ^ swz(row))) << 4) + ((col & 7) << 1); }
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { bf2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }

__global__ void k(const unsigned short* Kg, const unsigned short* Qg, float* S_out, float* dQ_out) {
  __shared__ __attribute__((aligned(16))) char Ks[32 * 128];
  const int l = threadIdx.x;
  for (int idx = l; idx < 32 * 64; idx += 64) {
    const int row = idx >> 6, col = idx & 63;
    *reinterpret_cast<unsigned short*>(Ks + tile_off(row, col)) = col < 48 ? Kg[row * 48 + col] : (unsigned short)0;
  }
  __syncthreads();
  f32x16 sc;
  for (int a = 0; a < 16; ++a) sc[a] = 0.f;
  for (int ks = 0; ks < 3; ++ks) {
    const int row = l & 31, slot = ks * 2 + (l >> 5);
    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + row * 128 + ((slot ^ swz(row)) << 4));
    bf16x8 qf;
    for (int e = 0; e < 8; ++e) qf[e] = Qg[(l & 31) * 48 + ks * 16 + (l >> 5) * 8 + e];
    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, sc, 0, 0, 0);
  }
  for (int a = 0; a < 16; ++a) S_out[(l & 31) * 32 + 8 * (a / 4) + 4 * (l >> 5) + (a % 4)] = sc[a];  // S[q][key]
  // pack: X = keys j in {0,1}, Y = j in {2,3}
  unsigned x[4], y[4];
  for (int w = 0; w < 4; ++w) {
    x[w] = pk(sc[2 * w], sc[2 * w + 1]);
    y[w] = pk(sc[8 + 2 * w], sc[8 + 2 * w + 1]);
  }
  for (int w = 0; w < 4; ++w) {
    auto r = __builtin_amdgcn_permlane16_swap(x[w], y[w], false, false);
    x[w] = r[0];
    y[w] = r[1];
  }
  union { bf16x8 v; unsigned w[4]; } u0, u1;
  for (int w = 0; w < 4; ++w) { u0.w[w] = x[w]; u1.w[w] = y[w]; }
  const int g = l >> 4, s = l & 15;
  const int base = (g & 1) * 16 + (g >> 1) * 4;  // k-slot group g: keys base + {0..3}, base + 8 + {0..3}
  for (int dt = 0; dt < 3; ++dt) {
    const int r_lo = base + (s >> 2), col = dt * 16 + 4 * (s & 3);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(Ks + tile_off(r_lo, col)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(Ks + tile_off(r_lo + 8, col)));
    bf16x8 kc;
    kc[0] = lo[0]; kc[1] = lo[1]; kc[2] = lo[2]; kc[3] = lo[3]; kc[4] = hi[0]; kc[5] = hi[1]; kc[6] = hi[2]; kc[7] = hi[3];
    f32x4 d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, u0.v, d0, 0, 0, 0);
    d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, u1.v, d1, 0, 0, 0);
    for (int r = 0; r < 4; ++r) {  // lane (i = q within the tile, g): columns dt * 16 + 4 g + r
      dQ_out[(l & 15) * 48 + dt * 16 + 4 * g + r] = d0[r];
      dQ_out[(16 + (l & 15)) * 48 + dt * 16 + 4 * g + r] = d1[r];
    }
  }
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float fb(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  unsigned short hK[32 * 48], hQ[32 * 48];
  float hS[32 * 32], hD[32 * 48];
  for (int i = 0; i < 32 * 48; ++i) { hK[i] = bf((float)((i * 7) % 5 - 2)); hQ[i] = bf((float)((i * 3) % 3 - 1)); }
  for (int r = 0; r < 32; ++r) for (int c = 40; c < 48; ++c) { hK[r * 48 + c] = 0; hQ[r * 48 + c] = 0; }
  unsigned short *dK, *dQ; float *dS, *dD;
  hipMalloc(&dK, sizeof hK); hipMalloc(&dQ, sizeof hQ); hipMalloc(&dS, sizeof hS); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dK, hK, sizeof hK, hipMemcpyHostToDevice); hipMemcpy(dQ, hQ, sizeof hQ, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dK, dQ, dS, dD);
  hipMemcpy(hS, dS, sizeof hS, hipMemcpyDeviceToHost); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  int badS = 0, badD = 0;
  float S[32][32];
  for (int q = 0; q < 32; ++q) for (int key = 0; key < 32; ++key) {
    float r = 0; for (int c = 0; c < 48; ++c) r += fb(hQ[q * 48 + c]) * fb(hK[key * 48 + c]);
    S[q][key] = r;
    if (hS[q * 32 + key] != r) ++badS;
  }
  for (int q = 0; q < 32; ++q) for (int c = 0; c < 48; ++c) {
    float r = 0; for (int key = 0; key < 32; ++key) r += fb(bf(S[q][key])) * fb(hK[key * 48 + c]);
    if (hD[q * 48 + c] != r) ++badD;
  }
  printf("S = Q K^T on 32x32x16: %d mismatches of 1024; dQ = S K through permlane16_swap + 16x16x32: %d mismatches of 1536\n", badS, badD);
  return badS || badD;
}
