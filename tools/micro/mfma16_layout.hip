// Operand layout check of v_mfma_f32_16x16x16_bf16 (the "_1k" builtin) on gfx950: D[n][m] = sum_k A[n][k] B[m][k] with lane
// (i = l & 15, g = l >> 4) holding A[i][4g .. 4g+3] / B[i][4g .. 4g+3]; output lane (i, g) = D[4g + r][i].
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const unsigned short* A, const unsigned short* B, float* D) {
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  s16x4 a, b;
  for (int e = 0; e < 4; ++e) { a[e] = A[i * 16 + 4 * g + e]; b[e] = B[i * 16 + 4 * g + e]; }
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = acc[r];
}
static unsigned short bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float fb(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  unsigned short hA[256], hB[256]; float hD[256];
  for (int i = 0; i < 256; ++i) { hA[i] = bf((float)((i * 7) % 13 - 6)); hB[i] = bf((float)((i * 5) % 11 - 5)); }
  unsigned short *dA, *dB; float* dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  int bad_nm = 0, bad_mn = 0;
  for (int n = 0; n < 16; ++n) for (int m = 0; m < 16; ++m) {
    float ref = 0; for (int kk = 0; kk < 16; ++kk) ref += fb(hA[n * 16 + kk]) * fb(hB[m * 16 + kk]);
    if (hD[n * 16 + m] != ref) ++bad_nm;
    if (hD[m * 16 + n] != ref) ++bad_mn;
  }
  printf("D[n][m] = A[n].B[m]: %d mismatches; transposed reading: %d mismatches\n", bad_nm, bad_mn);
  return 0;
}
