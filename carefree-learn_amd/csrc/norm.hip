// K5: LayerNorm forward / backward for gfx950 (HBM-bound; one 64-lane wave per row).
//
// Replaces nn.LayerNorm as built by NormFactory("layer") (reference modules/core/norms.py:88-89,
// 118-119; used at mixed_stacks/api.py:141,155 and in the head PreNorm, api.py:397-402).
// Semantics: biased variance, eps inside the sqrt, statistics in fp32.
//
// A row lives entirely in registers: lane l owns elements [c*256 + 4*l, +4) of every 256-wide chunk
// (8-byte bf16x4 loads, fully coalesced: one wave instruction = 512 contiguous bytes), reductions are
// wave-wide xor-shuffles, gamma / beta are hoisted into registers once per wave.  The backward
// keeps per-lane partial dgamma / dbeta in registers across all rows a wave visits, folds the
// waves of a workgroup through LDS float atomics, and a second tiny kernel reduces the per-
// workgroup partials — no global atomics, deterministic for a fixed launch geometry.
#include "common.h"

int cfhip_internal_colreduce_f32(const float* partials, int R, int D, float* out, int accumulate,
                                 hipStream_t s);

namespace {

constexpr int LN_WAVES = 8;  // waves per workgroup
constexpr int LN_THREADS = LN_WAVES * 64;

__device__ __forceinline__ void load4(const bf16_t* p, float (&v)[4]) {
  const u32x2 w = *reinterpret_cast<const u32x2*>(p);
  v[0] = bf16lo(w[0]); v[1] = bf16hi(w[0]); v[2] = bf16lo(w[1]); v[3] = bf16hi(w[1]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int D,
    long xs, long ys, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * LN_WAVES + wave;
  const int nw = gridDim.x * LN_WAVES;
  float gm[NCH][4], bt[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gm[c][e] = (col + e < D) ? gamma[col + e] : 0.f;
      bt[c][e] = (col + e < D) ? beta[col + e] : 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  for (int row = gw; row < M; row += nw) {
    const bf16_t* xr = x + (long)row * xs;
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) load4(xr + col, v[c]);
      else { v[c][0] = v[c][1] = v[c][2] = v[c][3] = 0.f; }
      s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    const float mean = wave_sum(s) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; sq += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(sq) * inv_d + eps);
    bf16_t* yr = y + (long)row * ys;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[c][e] - mean) * rstd * gm[c][e] + bt[c][e];
        store4(yr + col, o);
      }
    }
    if (lane == 0) {
      if (mean_out) mean_out[row] = mean;
      if (rstd_out) rstd_out[row] = rstd;
    }
  }
}

template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ dx, float* __restrict__ partials, int M,
    int D, long dys, long xs, long dxs) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][D]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * LN_WAVES + wave;
  const int nw = gridDim.x * LN_WAVES;
  for (int i = threadIdx.x; i < 2 * D; i += LN_THREADS) red[i] = 0.f;
  __syncthreads();

  float gm[NCH][4], dg[NCH][4], db[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gm[c][e] = (col + e < D) ? gamma[col + e] : 0.f;
      dg[c][e] = 0.f;
      db[c][e] = 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  for (int row = gw; row < M; row += nw) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    const bf16_t* xr = x + (long)row * xs;
    const bf16_t* dyr = dy + (long)row * dys;
    float xh[NCH][4], g[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float xv[4], dv[4];
        load4(xr + col, xv);
        load4(dyr + col, dv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          xh[c][e] = (xv[e] - mean) * rstd;
          g[c][e] = dv[e] * gm[c][e];
          s1 += g[c][e];
          s2 += g[c][e] * xh[c][e];
          dg[c][e] += dv[e] * xh[c][e];
          db[c][e] += dv[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { xh[c][e] = 0.f; g[c][e] = 0.f; }
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
    bf16_t* dxr = dx + (long)row * dxs;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = rstd * (g[c][e] - s1 - xh[c][e] * s2);
        if (dx_add != nullptr) {
          float a[4];
          load4(dx_add + (long)row * dxs + col, a);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += a[e];
        }
        store4(dxr + col, o);
      }
    }
  }
  // fold the workgroup's waves through LDS, then one partial row per workgroup
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e < D) {
        atomicAdd(&red[col + e], dg[c][e]);
        atomicAdd(&red[D + col + e], db[c][e]);
      }
    }
  }
  __syncthreads();
  float* out = partials + (long)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += LN_THREADS) out[i] = red[i];
}

inline int ln_grid(int M) {
  int blocks = (M + LN_WAVES - 1) / LN_WAVES;
  if (blocks > 512) blocks = 512;  // 2 workgroups of 8 waves per CU
  if (blocks < 1) blocks = 1;
  return blocks;
}

}  // namespace

#define LN_DISPATCH(KERNEL, nch, ...)                                      \
  switch (nch) {                                                           \
    case 1: KERNEL(1, __VA_ARGS__); break;                                 \
    case 2: KERNEL(2, __VA_ARGS__); break;                                 \
    case 3: KERNEL(3, __VA_ARGS__); break;                                 \
    case 4: KERNEL(4, __VA_ARGS__); break;                                 \
    case 5: case 6: KERNEL(6, __VA_ARGS__); break;                         \
    default: KERNEL(8, __VA_ARGS__); break;                                \
  }

extern "C" int cfhip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, int M, int D, int64_t x_row_stride,
                                   int64_t y_row_stride, float eps, void* stream) {
  CFHIP_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  CFHIP_REQUIRE(M > 0 && D > 0, "layernorm_fwd: empty problem");
  CFHIP_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm_fwd: D=%d must be a multiple of 4 and <= 2048", D);
  CFHIP_REQUIRE(x_row_stride % 4 == 0 && y_row_stride % 4 == 0, "layernorm_fwd: row strides must be multiples of 4");
  CFHIP_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0, "layernorm_fwd: x / y must be 8-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (D + 255) / 256;
  const int blocks = ln_grid(M);
#define LN_FWD(N_, ...)                                                                          \
  hipLaunchKernelGGL((layernorm_fwd_kernel<N_>), dim3(blocks), dim3(LN_THREADS), 0, s,            \
                     (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D, (long)x_row_stride, \
                     (long)y_row_stride, eps)
  LN_DISPATCH(LN_FWD, nch, 0)
#undef LN_FWD
  CFHIP_CHECK_LAUNCH("layernorm_fwd");
  return CFHIP_OK;
}

extern "C" size_t cfhip_layernorm_bwd_workspace(int M, int D) {
  return (size_t)ln_grid(M) * 2 * (size_t)D * sizeof(float);
}

extern "C" int cfhip_layernorm_bwd(const void* dy, const void* x, const float* gamma,
                                   const float* mean, const float* rstd, const void* dx_add, void* dx,
                                   float* dgamma, float* dbeta, int M, int D, int64_t dy_row_stride,
                                   int64_t x_row_stride, int64_t dx_row_stride,
                                   int accumulate_param_grads, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(dy && x && gamma && mean && rstd && dx, "layernorm_bwd: null pointer");
  CFHIP_REQUIRE(M > 0 && D > 0, "layernorm_bwd: empty problem");
  CFHIP_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm_bwd: D=%d must be a multiple of 4 and <= 2048", D);
  CFHIP_REQUIRE(dy_row_stride % 4 == 0 && x_row_stride % 4 == 0 && dx_row_stride % 4 == 0,
                "layernorm_bwd: row strides must be multiples of 4");
  CFHIP_REQUIRE(((uintptr_t)dy & 7) == 0 && ((uintptr_t)x & 7) == 0 && ((uintptr_t)dx & 7) == 0 &&
                    ((uintptr_t)dx_add & 7) == 0,
                "layernorm_bwd: tensors must be 8-byte aligned");
  const size_t need = cfhip_layernorm_bwd_workspace(M, D);
  if (workspace == nullptr || workspace_bytes < need) {
    cfhip_set_error("layernorm_bwd: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return CFHIP_ERR_WORKSPACE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (D + 255) / 256;
  const int blocks = ln_grid(M);
  float* partials = reinterpret_cast<float*>(workspace);
  const size_t lds = (size_t)2 * D * sizeof(float);
#define LN_BWD(N_, ...)                                                                            \
  hipLaunchKernelGGL((layernorm_bwd_kernel<N_>), dim3(blocks), dim3(LN_THREADS), lds, s,            \
                     (const bf16_t*)dy, (const bf16_t*)x, gamma, mean, rstd, (const bf16_t*)dx_add, \
                     (bf16_t*)dx, partials, M, D, (long)dy_row_stride, (long)x_row_stride,          \
                     (long)dx_row_stride)
  LN_DISPATCH(LN_BWD, nch, 0)
#undef LN_BWD
  CFHIP_CHECK_LAUNCH("layernorm_bwd");
  if (dgamma != nullptr || dbeta != nullptr) {
    // partial rows are [2*D] wide: dgamma in the first half, dbeta in the second
    if (dgamma != nullptr && dbeta == dgamma + D) {
      return cfhip_internal_colreduce_f32(partials, blocks, 2 * D, dgamma, accumulate_param_grads, s);
    }
    // separate destinations: reduce each half with a row pitch of 2*D
    // (colreduce takes a dense [R][D] matrix, so reduce into the workspace tail first)
    float* tmp = partials;  // reuse row 0 region after the reduce of both halves
    int rc = cfhip_internal_colreduce_f32(partials, blocks, 2 * D, tmp, 0, s);
    if (rc != CFHIP_OK) return rc;
    // tmp[0:D] = dgamma, tmp[D:2D] = dbeta (row 0 of the partials was consumed in place)
    if (dgamma != nullptr) {
      rc = cfhip_internal_colreduce_f32(tmp, 1, D, dgamma, accumulate_param_grads, s);
      if (rc != CFHIP_OK) return rc;
    }
    if (dbeta != nullptr) {
      rc = cfhip_internal_colreduce_f32(tmp + D, 1, D, dbeta, accumulate_param_grads, s);
      if (rc != CFHIP_OK) return rc;
    }
  }
  return CFHIP_OK;
}
