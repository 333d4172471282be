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

// ---- similarity logits of the two CLIP towers in fp32 (multimodal/schema.py:25-30: logit_scale * I @ T^T) --------------
// C[m][n] = alpha * sum_k A(m,k) * B(n,k); A(m,k) = ta ? A[k*lda + m] : A[m*lda + k], likewise B.  fp32 FMA chain in k
// order: the features are L2-normalised fp32 rows and the logits feed a softmax at scale ~14-100, so the operands are
// NOT rounded to bf16.  [B x W*B] x 512: < 1 GFLOP per launch, a workgroup computes a 16x16 output block from LDS tiles.
__global__ __launch_bounds__(256) void sgemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                        float* __restrict__ C, int M, int N, int K, long lda, long ldb,
                                                        int ta, int tb, const float* __restrict__ alpha_ptr, float alpha) {
  __shared__ float sa[16][17], sb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    // sa[r][c] = A(m0 + r, k0 + c), sb[r][c] = B(n0 + r, k0 + c); the lane -> element map follows the contiguous axis
    {
      const int r = ta ? tx : ty, c = ta ? ty : tx;
      const int mm = blockIdx.y * 16 + r, kk = k0 + c;
      sa[r][c] = (mm < M && kk < K) ? (ta ? A[(long)kk * lda + mm] : A[(long)mm * lda + kk]) : 0.f;
    }
    {
      const int r = tb ? tx : ty, c = tb ? ty : tx;
      const int nn = blockIdx.x * 16 + r, kk = k0 + c;
      sb[r][c] = (nn < N && kk < K) ? (tb ? B[(long)kk * ldb + nn] : B[(long)nn * ldb + kk]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) acc = fmaf(sa[ty][c], sb[tx][c], acc);
    __syncthreads();
  }
  if (m < M && n < N) C[(long)m * N + n] = acc * (alpha_ptr != nullptr ? alpha_ptr[0] * alpha : alpha);
}

// out[0] += sum_i a[i] * b[i]  (caller zeroes; d logit_scale = sum(dlogits * logits))
__global__ __launch_bounds__(256) void dot_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s = fmaf(a[i], b[i], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));
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

extern "C" int cfhip_sgemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int64_t lda, int64_t ldb,
                               int a_trans, int b_trans, const float* alpha_dev, float alpha, void* stream) {
  CFHIP_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "sgemm_f32: bad arguments");
  hipLaunchKernelGGL(sgemm_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N,
                     K, (long)lda, (long)ldb, a_trans, b_trans, alpha_dev, alpha);
  CFHIP_CHECK_LAUNCH("sgemm_f32");
  return CFHIP_OK;
}

extern "C" int cfhip_dot_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
  CFHIP_REQUIRE(a && b && out && n > 0, "dot_f32: bad arguments");
  long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(dot_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n);
  CFHIP_CHECK_LAUNCH("dot_f32");
  return CFHIP_OK;
}
