// Row gather / scatter-add and row L2 normalisation: the index work of the CLIP text tower
// (multimodal/clip.py:209-256: nn.Embedding token lookup, learned positional add, EOT-token pooling
// `net[arange(B), indices.argmax(-1)]`, `l2_normalize`).  Integer index arithmetic: the gather is bit-exact.
#include "common.h"

namespace {

// out[n][:] = table[idx[n]][:] (+ pos[n % T][:]);  one wave per row, 4 floats per lane per pass
template <bool OUT_F32>
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int64_t* __restrict__ idx,
                                     const float* __restrict__ pos, void* __restrict__ out, long N, int D, int T,
                                     long V) {
  const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  long row = idx[n];
  if (row < 0 || row >= V) row = 0;  // (torch raises; out-of-range ids never reach here from the host wrapper)
  const float* src = table + row * D;
  const float* pp = pos != nullptr ? pos + (long)(n % T) * D : nullptr;
  for (int d = lane * 4; d < D; d += 256) {
    f32x4 v = *reinterpret_cast<const f32x4*>(src + d);
    if (pp != nullptr) v += *reinterpret_cast<const f32x4*>(pp + d);
    if (OUT_F32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + n * D + d) = v;
    } else {
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(out) + n * D + d) =
          u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

// dtable[idx[n]][:] += dy[n][:]  (f32 hardware atomics: several n may share a row); rows == padding_idx skipped
template <bool DY_F32>
__global__ void embedding_bwd_kernel(const void* __restrict__ dy, const int64_t* __restrict__ idx,
                                     float* __restrict__ dtable, long N, int D, long V, long padding_idx) {
  const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  const long row = idx[n];
  if (row < 0 || row >= V || row == padding_idx) return;
  float* dst = dtable + row * D;
  for (int d = lane; d < D; d += 64) {
    const float g = DY_F32 ? reinterpret_cast<const float*>(dy)[n * D + d]
                           : bf16_to_f32(reinterpret_cast<const bf16_t*>(dy)[n * D + d]);
    atomicAdd(dst + d, g);
  }
}

// y = x / ||x||_2 per row (cftool.array.l2_normalize: no epsilon); f32 rows, one wave per row
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm,
                                  long N, int D) {
  const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float v = x[n * D + d];
    s += v * v;
  }
  s = wave_sum(s);
  const float inv = 1.0f / sqrtf(s);
  for (int d = lane; d < D; d += 64) y[n * D + d] = x[n * D + d] * inv;
  if (lane == 0) inv_norm[n] = inv;
}
// dx = (dy - y <y, dy>) / ||x||
__global__ void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                  const float* __restrict__ inv_norm, float* __restrict__ dx, long N, int D) {
  const long n = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= N) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += y[n * D + d] * dy[n * D + d];
  s = wave_sum(s);
  const float inv = inv_norm[n];
  for (int d = lane; d < D; d += 64) dx[n * D + d] = (dy[n * D + d] - y[n * D + d] * s) * inv;
}

}  // namespace

extern "C" int cfhip_embedding_fwd(const float* table, const int64_t* indices, const float* pos, void* out,
                                   int out_is_f32, int64_t N, int D, int T, int64_t V, void* stream) {
  CFHIP_REQUIRE(table && indices && out && N > 0 && D > 0 && V > 0, "embedding_fwd: bad arguments");
  CFHIP_REQUIRE(D % 4 == 0 && ((uintptr_t)table & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                    (pos == nullptr || ((uintptr_t)pos & 15) == 0),
                "embedding_fwd: D must be a multiple of 4 and the buffers 16-byte aligned");
  CFHIP_REQUIRE(pos == nullptr || T > 0, "embedding_fwd: positional table needs T > 0");
  const dim3 grid((unsigned)((N + 3) / 4));
  if (out_is_f32)
    hipLaunchKernelGGL((embedding_fwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, table, indices, pos, out,
                       (long)N, D, T > 0 ? T : 1, (long)V);
  else
    hipLaunchKernelGGL((embedding_fwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, table, indices, pos, out,
                       (long)N, D, T > 0 ? T : 1, (long)V);
  CFHIP_CHECK_LAUNCH("embedding_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_embedding_bwd(const void* dy, int dy_is_f32, const int64_t* indices, float* dtable, int64_t N,
                                   int D, int64_t V, int64_t padding_idx, void* stream) {
  CFHIP_REQUIRE(dy && indices && dtable && N > 0 && D > 0 && V > 0, "embedding_bwd: bad arguments");
  const dim3 grid((unsigned)((N + 3) / 4));
  if (dy_is_f32)
    hipLaunchKernelGGL((embedding_bwd_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, dy, indices, dtable,
                       (long)N, D, (long)V, (long)padding_idx);
  else
    hipLaunchKernelGGL((embedding_bwd_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, dy, indices, dtable,
                       (long)N, D, (long)V, (long)padding_idx);
  CFHIP_CHECK_LAUNCH("embedding_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_l2norm_fwd(const float* x, float* y, float* inv_norm, int64_t N, int D, void* stream) {
  CFHIP_REQUIRE(x && y && inv_norm && N > 0 && D > 0, "l2norm_fwd: bad arguments");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y,
                     inv_norm, (long)N, D);
  CFHIP_CHECK_LAUNCH("l2norm_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int64_t N, int D,
                                void* stream) {
  CFHIP_REQUIRE(dy && y && inv_norm && dx && N > 0 && D > 0, "l2norm_bwd: bad arguments");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy, y,
                     inv_norm, dx, (long)N, D);
  CFHIP_CHECK_LAUNCH("l2norm_bwd");
  return CFHIP_OK;
}
