// Micro-benchmark: does a tiled (128-B-segment-per-row) store pattern cost more HBM traffic / time than
// a linear one?  Build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// linear: lane-contiguous 16 B, wave = 1 KiB contiguous
__global__ void st_linear(u32x4* out, long n16) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
    out[i] = u32x4{1u, 2u, 3u, 4u};
}
// tiled: workgroup = 128 rows x SEG bytes of a [M][N] bf16 matrix; wave instruction = (1024/SEG) rows x SEG bytes
template <int SEG>
__global__ void st_tiled(char* out, int M, int N) {
  const int tiles_n = (N * 2) / SEG;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPR = SEG / 16, RPI = 64 / LPR;  // lanes per row, rows per instruction
  for (int it = 0; it < 128 / (RPI * 4); ++it) {
    const int row = tile_m * 128 + (it * 4 + wave) * RPI + lane / LPR;
    if (row < M)
      *reinterpret_cast<u32x4*>(out + (long)row * N * 2 + tile_n * SEG + (lane % LPR) * 16) = u32x4{1u, 2u, 3u, 4u};
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int M = 12608, N = 3072;
  const long bytes = (long)M * N * 2;
  char* buf; CK(hipMalloc(&buf, bytes));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.1f us  %6.2f TB/s\n", name, ms * 50, bytes / (ms / 20 * 1e-3) / 1e12);
  };
  timeit("linear 1 KiB / wave", [&] { hipLaunchKernelGGL(st_linear, dim3(2048), dim3(256), 0, 0, (u32x4*)buf, bytes / 16); });
  const int tm = (M + 127) / 128;
  timeit("tiled 128 B segments", [&] { hipLaunchKernelGGL((st_tiled<128>), dim3(tm * (N * 2 / 128)), dim3(256), 0, 0, buf, M, N); });
  timeit("tiled 256 B segments", [&] { hipLaunchKernelGGL((st_tiled<256>), dim3(tm * (N * 2 / 256)), dim3(256), 0, 0, buf, M, N); });
  timeit("tiled 512 B segments", [&] { hipLaunchKernelGGL((st_tiled<512>), dim3(tm * (N * 2 / 512)), dim3(256), 0, 0, buf, M, N); });
  timeit("tiled 1024 B segments", [&] { hipLaunchKernelGGL((st_tiled<1024>), dim3(tm * (N * 2 / 1024)), dim3(256), 0, 0, buf, M, N); });
  CK(hipDeviceSynchronize());
  return 0;
}
