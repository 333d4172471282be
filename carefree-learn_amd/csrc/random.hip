// A12: dropout and stochastic depth (DropPath) — the reference's nn.Dropout sites (channel_mixers.py:30-41,
// mixed_stacks/api.py:130-158, ml_encoder.py:60-70, mappings.py) and DropPath (modules/core/customs.py:429-446:
// `net.div(keep) * floor(keep + U[0,1))` per sample, training only).
//
// Random bits come from Philox4x32-10 (counter-based: the stream is a pure function of (seed, offset, element index)),
// so the backward pass REGENERATES the mask from the same (seed, offset) instead of storing it: no mask tensor in HBM
// at all.  HBM-bound streaming kernels: 16-byte accesses, one Philox call per 4 elements.  For parity tests the mask
// can be injected (`mask_in`, one byte per element): given the mask the result is bit-equal to torch's
// `x * (mask / (1 - p))` with the noise formed in x's dtype (ATen Dropout.cpp).
#include "common.h"
#include <string.h>

namespace {

// uniform in [0, 1) with 24 random bits (every value exactly representable)
__device__ __forceinline__ float u01(unsigned r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// y = x * keep * scale;  keep = mask_in[i] != 0, or u01(philox(offset + i / 4)[i % 4]) >= p.  n4 groups of 4.
template <bool F32>
__global__ void dropout_kernel(const void* __restrict__ x, void* __restrict__ y, long n, float p, float scale,
                               unsigned long long seed, unsigned long long offset,
                               const unsigned char* __restrict__ mask_in, unsigned char* __restrict__ mask_out) {
  const long n4 = (n + 3) >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i4 = blockIdx.x * (long)blockDim.x + threadIdx.x; i4 < n4; i4 += stride) {
    const long base = i4 << 2;
    bool keep[4];
    if (mask_in != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) keep[e] = base + e < n && mask_in[base + e] != 0;
    } else {
      const Philox r = philox4x32_10(offset + (unsigned long long)i4, seed);
#pragma unroll
      for (int e = 0; e < 4; ++e) keep[e] = u01(r.c[e]) >= p;
    }
    if (base + 3 < n) {
      if (F32) {
        f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + base);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * scale : 0.f;
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + base) = v;
      } else {
        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(x) + base);
        const float v0 = keep[0] ? bf16lo(w[0]) * scale : 0.f, v1 = keep[1] ? bf16hi(w[0]) * scale : 0.f;
        const float v2 = keep[2] ? bf16lo(w[1]) * scale : 0.f, v3 = keep[3] ? bf16hi(w[1]) * scale : 0.f;
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + base) = u32x2{pack_bf16x2(v0, v1), pack_bf16x2(v2, v3)};
      }
      if (mask_out != nullptr) {
        *reinterpret_cast<unsigned*>(mask_out + base) =
            (unsigned)keep[0] | ((unsigned)keep[1] << 8) | ((unsigned)keep[2] << 16) | ((unsigned)keep[3] << 24);
      }
    } else {
      for (int e = 0; e < 4 && base + e < n; ++e) {
        if (F32) {
          const float v = reinterpret_cast<const float*>(x)[base + e];
          reinterpret_cast<float*>(y)[base + e] = keep[e] ? v * scale : 0.f;
        } else {
          const float v = bf16_to_f32(reinterpret_cast<const bf16_t*>(x)[base + e]);
          reinterpret_cast<bf16_t*>(y)[base + e] = f32_to_bf16(keep[e] ? v * scale : 0.f);
        }
        if (mask_out != nullptr) mask_out[base + e] = keep[e];
      }
    }
  }
}

// DropPath sample mask: mask[b] = floor(keep_prob + u_b)  (0 or 1; customs.py:439-441)
__global__ void drop_path_mask_kernel(float* __restrict__ mask, long B, float keep_prob, unsigned long long seed,
                                      unsigned long long offset) {
  const long b = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (b >= B) return;
  const Philox r = philox4x32_10(offset + (unsigned long long)(b >> 2), seed);
  mask[b] = floorf(keep_prob + u01(r.c[b & 3]));
}

// y[b][:] = (x[b][:] / keep_prob) * mask[b] — the reference's two roundings (customs.py:442: `net.div(keep) * rand`),
// true IEEE division; inner % 4 == 0: 8 / 16-byte accesses
template <bool F32>
__global__ void drop_path_kernel(const void* __restrict__ x, void* __restrict__ y, const float* __restrict__ mask,
                                 float keep_prob, long B, long inner) {
  const long i4n = inner >> 2;
  const long total = B * i4n;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / i4n;
    const float m = mask[b];
    const long base = i << 2;
    if (F32) {
      f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + base);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __fdiv_rn(v[e], keep_prob) * m;
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + base) = v;
    } else {
      // bf16 tensors: torch divides in f32 and rounds to bf16, then multiplies by the 0 / 1 mask
      const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(x) + base);
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + base) =
          u32x2{pack_bf16x2(__fdiv_rn(bf16lo(w[0]), keep_prob) * m, __fdiv_rn(bf16hi(w[0]), keep_prob) * m),
                pack_bf16x2(__fdiv_rn(bf16lo(w[1]), keep_prob) * m, __fdiv_rn(bf16hi(w[1]), keep_prob) * m)};
    }
  }
}

inline int stream_grid(long work, int threads) {
  long blocks = (work + threads - 1) / threads;
  if (blocks > 4096) blocks = 4096;
  return blocks < 1 ? 1 : (int)blocks;
}

}  // namespace

extern "C" int cfhip_dropout(const void* x, void* y, int is_f32, int64_t n, float p, uint64_t seed, uint64_t offset,
                             const uint8_t* mask_in, uint8_t* mask_out, void* stream) {
  CFHIP_REQUIRE(x && y, "dropout: null pointer");
  CFHIP_REQUIRE(n > 0, "dropout: empty tensor");
  CFHIP_REQUIRE(p >= 0.f && p < 1.f, "dropout: p = %f must be in [0, 1)", (double)p);
  CFHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)mask_out & 3) == 0,
                "dropout: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // torch forms the noise `mask / (1 - p)` in the INPUT's dtype: for bf16 tensors the scale is rounded to bf16 first
  // (the product of two bf16 values is exact in f32, so one rounding to bf16 follows, as in torch's bf16 multiply)
  float scale = 1.0f / (1.0f - p);
  if (!is_f32) {
    unsigned u;
    memcpy(&u, &scale, 4);
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;  // round-to-nearest-even to bf16 (finite, positive)
    memcpy(&scale, &u, 4);
  }
  const int grid = stream_grid((n + 3) / 4, 256);
  if (is_f32)
    hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid), dim3(256), 0, s, x, y, (long)n, p, scale,
                       (unsigned long long)seed, (unsigned long long)offset, mask_in, mask_out);
  else
    hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid), dim3(256), 0, s, x, y, (long)n, p, scale,
                       (unsigned long long)seed, (unsigned long long)offset, mask_in, mask_out);
  CFHIP_CHECK_LAUNCH("dropout");
  return CFHIP_OK;
}

extern "C" int cfhip_drop_path_mask(float* mask, int64_t B, float keep_prob, uint64_t seed, uint64_t offset,
                                    void* stream) {
  CFHIP_REQUIRE(mask && B > 0, "drop_path_mask: bad arguments");
  CFHIP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "drop_path_mask: keep_prob = %f must be in (0, 1]", (double)keep_prob);
  hipLaunchKernelGGL(drop_path_mask_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), mask, (long)B, keep_prob, (unsigned long long)seed,
                     (unsigned long long)offset);
  CFHIP_CHECK_LAUNCH("drop_path_mask");
  return CFHIP_OK;
}

extern "C" int cfhip_drop_path(const void* x, void* y, int is_f32, const float* mask, float keep_prob, int64_t B,
                               int64_t inner, void* stream) {
  CFHIP_REQUIRE(x && y && mask, "drop_path: null pointer");
  CFHIP_REQUIRE(B > 0 && inner > 0 && inner % 4 == 0, "drop_path: inner = %ld must be a positive multiple of 4", (long)inner);
  CFHIP_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "drop_path: keep_prob = %f must be in (0, 1]", (double)keep_prob);
  CFHIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "drop_path: tensors must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int grid = stream_grid(B * (inner / 4), 256);
  if (is_f32) hipLaunchKernelGGL(drop_path_kernel<true>, dim3(grid), dim3(256), 0, s, x, y, mask, keep_prob, (long)B, (long)inner);
  else hipLaunchKernelGGL(drop_path_kernel<false>, dim3(grid), dim3(256), 0, s, x, y, mask, keep_prob, (long)B, (long)inner);
  CFHIP_CHECK_LAUNCH("drop_path");
  return CFHIP_OK;
}
