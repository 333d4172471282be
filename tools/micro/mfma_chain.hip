// v_mfma_f32_16x16x32_bf16 followed by v_mfma_f32_16x16x16_bf16 accumulating into its result (48-deep product), several
// independent chains back to back as in attn_fwd2_kernel: checks the compiler's hazard handling of the mixed chain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const unsigned short* A, const unsigned short* B, float* D) {  // A, B: [4][16][48]
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  f32x4 acc[4];
  bf16x8 a32[4], b32[4];
  s16x4 a16[4], b16[4];
  for (int c = 0; c < 4; ++c) {
    for (int e = 0; e < 8; ++e) { a32[c][e] = A[(c * 16 + i) * 48 + 8 * g + e]; b32[c][e] = B[(c * 16 + i) * 48 + 8 * g + e]; }
    for (int e = 0; e < 4; ++e) { a16[c][e] = A[(c * 16 + i) * 48 + 32 + 4 * g + e]; b16[c][e] = B[(c * 16 + i) * 48 + 32 + 4 * g + e]; }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x4 z = {0, 0, 0, 0};
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32[c], b32[c], z, 0, 0, 0);
    acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16[c], b16[c], z, 0, 0, 0);
  }
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) D[(c * 16 + 4 * g + r) * 16 + i] = acc[c][r];
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float fb(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const int N = 4 * 16 * 48;
  unsigned short hA[N], hB[N]; float hD[4 * 256];
  for (int i = 0; i < N; ++i) { hA[i] = bf((float)((i * 7) % 13 - 6)); hB[i] = bf((float)((i * 5) % 11 - 5)); }
  unsigned short *dA, *dB; float* dD;
  hipMalloc(&dA, N * 2); hipMalloc(&dB, N * 2); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, N * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB, N * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int c = 0; c < 4; ++c) for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
    float ref = 0; for (int kk = 0; kk < 48; ++kk) ref += fb(hA[(c * 16 + n) * 48 + kk]) * fb(hB[(c * 16 + m) * 48 + kk]);
    if (hD[(c * 16 + n) * 16 + m] != ref) ++bad;
  }
  printf("x32 -> x16 chains: %d mismatches of 1024\n", bad);
  return 0;
}
