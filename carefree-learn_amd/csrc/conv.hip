// K8 (general form) / K9 / K10 and friends: the pieces around the MFMA GEMM that make the reference's
// conv -> BatchNorm -> LeakyReLU stacks (convs/basic.py:41-184,529-571; cv/encoder/vanilla.py:18-158)
// and its FCNN head (ml/fcnn.py:12-57) run on gfx950.
//
//   Conv2d (groups 1) = im2row + K1 GEMM:  rows[(b,oy,ox)][(c,ky,kx)] (zero padded, K rounded up to a
//   multiple of 8 so the GEMM's 16-byte operand rule holds) x W[Cout][(c,ky,kx)]^T.  Backward: dW is the
//   (tn) GEMM of the same rows, dX = row2im(dY W) where row2im GATHERS the <= kh*kw contributions of
//   every input pixel (no atomics, deterministic).  NCHW <-> token-major conversions are batched LDS
//   transposes.  All of it is HBM-bound glue: coalesced along the fastest output dimension.
//
//   BatchNorm (training statistics over (B, inner) per channel, biased variance for the normalisation,
//   unbiased for running_var — torch semantics, norms.py:20-27,90-93), LeakyReLU / ReLU, global average
//   pooling, focal loss (losses/basic.py:170-206).
#include "common.h"
#include <math.h>

namespace {

inline int grid_for(long work_items, int per_block, int cap = 4096) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

__device__ __forceinline__ float ld_act(const void* p, long i, bool f32) {
  return f32 ? reinterpret_cast<const float*>(p)[i] : bf16_to_f32(reinterpret_cast<const bf16_t*>(p)[i]);
}

struct ConvGeom {
  int B, C, H, W, kh, kw, stride, pad, dil, Ho, Wo, K, Kp;
};

// rows[m][k], m = (b, oy, ox), k = (c, ky, kx); thread = one (m, k): writes coalesced along k
template <bool IN_F32>
__global__ void conv_im2row_kernel(const void* __restrict__ x, bf16_t* __restrict__ rows, ConvGeom g) {
  const long total = (long)g.B * g.Ho * g.Wo * g.Kp;
  const long step = (long)gridDim.x * blockDim.x;
  const int kk = g.kh * g.kw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const long m = i / g.Kp;
    const int k = (int)(i - m * g.Kp);
    float v = 0.f;
    if (k < g.K) {
      const int c = k / kk, r = k - c * kk;
      const int ky = r / g.kw, kx = r - ky * g.kw;
      const int ox = (int)(m % g.Wo);
      const long t = m / g.Wo;
      const int oy = (int)(t % g.Ho);
      const int b = (int)(t / g.Ho);
      const int iy = oy * g.stride - g.pad + ky * g.dil, ix = ox * g.stride - g.pad + kx * g.dil;
      if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
        v = ld_act(x, (((long)b * g.C + c) * g.H + iy) * g.W + ix, IN_F32);
    }
    rows[i] = f32_to_bf16(v);
  }
}

// dx[b][c][y][x] = sum over (ky, kx) of drows[(b, oy, ox)][(c, ky, kx)] with oy*stride - pad + ky*dil == y
__global__ void conv_row2im_kernel(const bf16_t* __restrict__ drows, bf16_t* __restrict__ dx, ConvGeom g) {
  const long total = (long)g.B * g.C * g.H * g.W;
  const long step = (long)gridDim.x * blockDim.x;
  const int kk = g.kh * g.kw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const int xx = (int)(i % g.W);
    long t = i / g.W;
    const int yy = (int)(t % g.H);
    t /= g.H;
    const int c = (int)(t % g.C);
    const int b = (int)(t / g.C);
    float acc = 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int ny = yy + g.pad - ky * g.dil;
      if (ny < 0 || ny % g.stride != 0) continue;
      const int oy = ny / g.stride;
      if (oy >= g.Ho) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int nx = xx + g.pad - kx * g.dil;
        if (nx < 0 || nx % g.stride != 0) continue;
        const int ox = nx / g.stride;
        if (ox >= g.Wo) continue;
        acc += bf16_to_f32(drows[(((long)b * g.Ho + oy) * g.Wo + ox) * g.Kp + c * kk + ky * g.kw + kx]);
      }
    }
    dx[i] = f32_to_bf16(acc);
  }
}

// batched [R][C] -> [C][R] through a padded 64x64 LDS tile; source f32 or bf16, destination bf16
template <bool IN_F32>
__global__ void transpose_batched_kernel(const void* __restrict__ src, bf16_t* __restrict__ dst, int R, int C,
                                         long bs_src, long bs_dst) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const long so = (long)blockIdx.z * bs_src, dofs = (long)blockIdx.z * bs_dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < R && cc < C) ? f32_to_bf16(ld_act(src, so + (long)rr * C + cc, IN_F32)) : (bf16_t)0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, rr = r0 + tx;
    if (cc < C && rr < R) dst[dofs + (long)cc * R + rr] = tile[tx][c];
  }
}

// The same for R % 8 == 0, C % 8 == 0 and 16-byte aligned bases (every NCHW <-> NHWC hop of the UNet: 290 + 30 launches per
// 64^2 x 8 step, 3.7 ms with the 2-byte form above): 16 bytes per lane on both sides of the memory traffic.  The 64 x 64 tile
// sits in LDS with its eight 16-byte chunks per row XOR-swizzled by (row / 8), so the 2-byte gathers of the transposed read —
// lanes (j, c) fetch element (8 j + i, c) — fall into 32 different banks; 8 consecutive lanes store one 128-byte run.
template <bool IN_F32>
__global__ __launch_bounds__(256) void transpose_batched_vec_kernel(const void* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                                                    int C, long bs_src, long bs_dst) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 64];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const long so = (long)blockIdx.z * bs_src, dofs = (long)blockIdx.z * bs_dst;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = threadIdx.x + 256 * k;
    const int r = q >> 3, ch = q & 7;
    const int rr = r0 + r, cc = c0 + ch * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (rr < R && cc < C) {
      const long o = so + (long)rr * C + cc;
      if (IN_F32) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + o);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(src) + o + 4);
        v = u32x4{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
      } else {
        v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(src) + o);
      }
    }
    *reinterpret_cast<u32x4*>(tile + r * 64 + ((ch ^ ((r >> 3) & 7)) << 3)) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int q = threadIdx.x + 256 * k;
    const int c = q >> 3, j = q & 7;  // output row c0 + c, source rows r0 + 8 j .. + 7
    const int cc = c0 + c, rr = r0 + 8 * j;
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[(8 * j + i) * 64 + (((c >> 3) ^ j) << 3) + (c & 7)];
    if (cc < C && rr < R)
      *reinterpret_cast<u32x4*>(dst + dofs + (long)cc * R + rr) =
          u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16),
                (unsigned)e[4] | ((unsigned)e[5] << 16), (unsigned)e[6] | ((unsigned)e[7] << 16)};
  }
}

// ---- BatchNorm: one workgroup per channel; x [B][C][inner] --------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void bn_fwd_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     float* running_mean, float* running_var, int B, int C,
                                                     int inner, float eps, float momentum, int training) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const long n = (long)B * inner;
  float mean, rstd;
  if (training) {
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const long b = i / inner, j = i - b * inner;
      s += ld_act(x, (b * C + c) * inner + j, IN_F32);
    }
    mean = block_sum(s, red) / (float)n;
    float q = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) {
      const long b = i / inner, j = i - b * inner;
      const float d = ld_act(x, (b * C + c) * inner + j, IN_F32) - mean;
      q += d * d;
    }
    const float var = block_sum(q, red) / (float)n;
    rstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
      mean_out[c] = mean;
      rstd_out[c] = rstd;
      if (running_mean != nullptr) {
        const float unbiased = n > 1 ? var * (float)n / (float)(n - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
      }
    }
  } else {
    mean = running_mean[c];
    rstd = rsqrtf(running_var[c] + eps);
    if (threadIdx.x == 0) {
      mean_out[c] = mean;
      rstd_out[c] = rstd;
    }
  }
  const float ga = gamma != nullptr ? gamma[c] : 1.f, be = beta != nullptr ? beta[c] : 0.f;
  const float a = rstd * ga, sh = be - mean * a;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const long b = i / inner, j = i - b * inner;
    const long o = (b * C + c) * inner + j;
    y[o] = f32_to_bf16(fmaf(ld_act(x, o, IN_F32), a, sh));
  }
}

// training-mode backward; eval-mode (`training == 0`): dx = dy * gamma * rstd, the statistics are constants
template <bool IN_F32>
__global__ __launch_bounds__(256) void bn_bwd_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int B,
                                                     int C, int inner, int accumulate, int training) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const long n = (long)B * inner;
  const float mu = mean[c], rs = rstd[c], ga = gamma != nullptr ? gamma[c] : 1.f;
  float sdy = 0.f, sdyx = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const long b = i / inner, j = i - b * inner;
    const long o = (b * C + c) * inner + j;
    const float g = bf16_to_f32(dy[o]);
    sdy += g;
    sdyx += g * (ld_act(x, o, IN_F32) - mu) * rs;
  }
  sdy = block_sum(sdy, red);
  sdyx = block_sum(sdyx, red);
  if (threadIdx.x == 0) {
    if (dgamma != nullptr) dgamma[c] = accumulate ? dgamma[c] + sdyx : sdyx;
    if (dbeta != nullptr) dbeta[c] = accumulate ? dbeta[c] + sdy : sdy;
  }
  if (dx == nullptr) return;
  const float inv_n = 1.f / (float)n;
  const float k = ga * rs;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const long b = i / inner, j = i - b * inner;
    const long o = (b * C + c) * inner + j;
    const float g = bf16_to_f32(dy[o]);
    float r = g;
    if (training) {
      const float xh = (ld_act(x, o, IN_F32) - mu) * rs;
      r = g - sdy * inv_n - xh * sdyx * inv_n;
    }
    dx[o] = f32_to_bf16(k * r);
  }
}

// ---- LeakyReLU (slope 0 = ReLU), 4 bf16 per thread -------------------------------------------------------
template <bool BWD>
__global__ void leaky_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ x, bf16_t* __restrict__ out,
                             long n, float slope) {
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4; i < n; i += step * 4) {
    if (i + 4 <= n) {
      const u32x2 va = *reinterpret_cast<const u32x2*>(a + i);
      u32x2 vx = va;
      if (BWD) vx = *reinterpret_cast<const u32x2*>(x + i);
      float r[4];
      const float av[4] = {bf16lo(va[0]), bf16hi(va[0]), bf16lo(va[1]), bf16hi(va[1])};
      const float xv[4] = {bf16lo(vx[0]), bf16hi(vx[0]), bf16lo(vx[1]), bf16hi(vx[1])};
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = BWD ? (xv[e] > 0.f ? av[e] : av[e] * slope) : (av[e] > 0.f ? av[e] : av[e] * slope);
      *reinterpret_cast<u32x2*>(out + i) = u32x2{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3])};
    } else {
      for (long j = i; j < n; ++j) {
        const float av = bf16_to_f32(a[j]);
        const float xv = BWD ? bf16_to_f32(x[j]) : av;
        out[j] = f32_to_bf16(xv > 0.f ? av : av * slope);
      }
    }
  }
}

// ---- global average pool: x [BC][inner] -> y [BC]; one wave per (b, c) -------------------------------------
__global__ void avgpool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long BC, int inner) {
  const long row = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= BC) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int j = lane; j < inner; j += 64) s += bf16_to_f32(x[row * inner + j]);
  s = wave_sum(s);
  if (lane == 0) y[row] = f32_to_bf16(s / (float)inner);
}
__global__ void avgpool_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long total, int inner) {
  const long step = (long)gridDim.x * blockDim.x;
  const float inv = 1.f / (float)inner;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step)
    dx[i] = f32_to_bf16(bf16_to_f32(dy[i / inner]) * inv);
}

// ---- focal loss (losses/basic.py:170-206): p = softmax(z) + eps; L = -log(p_y) (1 - p_y)^gamma -------------
// wave per sample; loss_sum += sum_b L_b; dlogits = grad_scale * dL/dz (the +eps is a constant shift)
__global__ void softmax_focal_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                     float* __restrict__ loss_sum, float* __restrict__ dlogits, int B, int C,
                                     float gamma, float eps, float grad_scale) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const float* z = logits + (long)b * C;
  float mx = -INFINITY;
  for (int j = lane; j < C; j += 64) mx = fmaxf(mx, z[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < C; j += 64) s += expf(z[j] - mx);
  s = wave_sum(s);
  const int yb = (int)labels[b];
  const float sy = expf(z[yb] - mx) / s;
  const float p = sy + eps;
  const float om = 1.f - p;
  const float lp = logf(p);
  const float w = powf(om, gamma);
  if (lane == 0) atomicAdd(loss_sum, -lp * w);
  if (dlogits != nullptr) {
    // dL/dp = -(1-p)^g / p + g (1-p)^(g-1) log p ;  dp/dz_j = s_y (delta_yj - s_j)
    const float dldp = -w / p + gamma * powf(om, gamma - 1.f) * lp;
    for (int j = lane; j < C; j += 64) {
      const float sj = expf(z[j] - mx) / s;
      dlogits[(long)b * C + j] = grad_scale * dldp * sy * ((j == yb ? 1.f : 0.f) - sj);
    }
  }
}

// ---- GroupNorm (+ per-(b,c) additive term in front, + SiLU behind) -----------------------------------------
// UNet residual blocks (convs/residual.py:154-253): net = conv1(..) + Linear(SiLU(t))[:, :, None, None];
// net = SiLU(GroupNorm32(net)).  One workgroup per (b, group): the group's Cg * inner elements are reduced in two
// passes (mean, then centred variance), statistics in fp32.  x' = x + add[b][c] is what gets normalised.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v)); }
__device__ __forceinline__ float silu_grad_f(float v) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  return s * fmaf(v, 1.0f - s, 1.0f);
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_fwd_kernel(const void* __restrict__ x, const float* __restrict__ add,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int C, int G, int inner, float eps,
                                                     int silu, int affine_bs) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  gamma += (long)b * affine_bs;  // per-sample affine (scale-shift norm): gamma / beta are [B][C] when affine_bs == C
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const long base = ((long)b * C + (long)g * cg) * inner;
  const long n = (long)cg * inner;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = (int)(i / inner);
    s += ld_act(x, base + i, IN_F32) + (add != nullptr ? add[(long)b * C + g * cg + c] : 0.f);
  }
  const float mean = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = (int)(i / inner);
    const float d = ld_act(x, base + i, IN_F32) + (add != nullptr ? add[(long)b * C + g * cg + c] : 0.f) - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum(q, red) / (float)n + eps);
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = g * cg + (int)(i / inner);
    const float v = ld_act(x, base + i, IN_F32) + (add != nullptr ? add[(long)b * C + c] : 0.f);
    float o = (v - mean) * rstd * gamma[c] + beta[c];
    if (silu) o = silu_f(o);
    y[base + i] = f32_to_bf16(o);
  }
}

// dx (bf16), partial dgamma / dbeta per (b, group) block [B][C] (reduced over b by the caller's column reduce),
// dadd[b][c] = sum_inner dx (when the forward had an additive term)
template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                     const float* __restrict__ add, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                     float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                     float* __restrict__ dadd, int C, int G, int inner, int silu, int affine_bs) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const long base = ((long)b * C + (long)g * cg) * inner;
  const long n = (long)cg * inner;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  // pass 1: per-channel dgamma / dbeta partials and the two group sums
  float s1 = 0.f, s2 = 0.f;  // sum(dn * gamma), sum(dn * gamma * xhat)
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int j = threadIdx.x; j < inner; j += blockDim.x) {
      const long o = base + (long)cl * inner + j;
      const float xh = (ld_act(x, o, IN_F32) + ad - mu) * rs;
      float dn = bf16_to_f32(dy[o]);
      if (silu) dn *= silu_grad_f(xh * ga + be);
      sg += dn * xh;
      sb += dn;
    }
    sg = block_sum(sg, red);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) {
      dgamma_part[(long)b * C + c] = sg;
      dbeta_part[(long)b * C + c] = sb;
    }
    s1 += sb * ga;
    s2 += sg * ga;
  }
  const float inv_n = 1.f / (float)n;
  // pass 2: dx = rstd * (dn*gamma - s1/n - xhat * s2/n)
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sd = 0.f;
    for (int j = threadIdx.x; j < inner; j += blockDim.x) {
      const long o = base + (long)cl * inner + j;
      const float xh = (ld_act(x, o, IN_F32) + ad - mu) * rs;
      float dn = bf16_to_f32(dy[o]);
      if (silu) dn *= silu_grad_f(xh * ga + be);
      const float d = rs * (dn * ga - s1 * inv_n - xh * s2 * inv_n);
      dx[o] = f32_to_bf16(d);
      sd += d;
    }
    if (dadd != nullptr) {
      sd = block_sum(sd, red);
      if (threadIdx.x == 0) dadd[(long)b * C + c] = sd;
    }
  }
}

// ---- GroupNorm, vector form: inner % 8 == 0, 16-byte aligned rows.  Same arithmetic as the scalar kernels above (two-pass
// statistics in fp32), 8 elements per lane per load (one 16-byte load of bf16, two of f32) and no integer division in
// the loops; the second and third pass over a group (80 KB - 1.3 MB) come out of L2.  UNet 64^2 x 8: the scalar forms
// took 74 / 116 us per call, 11.5 ms of a 101 ms step (profiles/r01/prof_unet64_b8_implicit_summary.txt).
template <bool IN_F32>
__device__ __forceinline__ void gn_load8(const void* p, long off, float (&v)[8]) {
  if (IN_F32) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + off);
    const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + off + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
    const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(p) + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = bf16lo(w[e]);
      v[2 * e + 1] = bf16hi(w[e]);
    }
  }
}
__device__ __forceinline__ void gn_store8(bf16_t* p, long off, const float (&v)[8]) {
  *reinterpret_cast<u32x4*>(p + off) = u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                             pack_bf16x2(v[6], v[7])};
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_fwd_vec_kernel(const void* __restrict__ x, const float* __restrict__ add,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out, int C, int G, int inner, float eps,
                                                         int silu, int affine_bs) {
  __shared__ float red[4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  const long base = ((long)b * C + (long)g * cg) * inner;
  const float inv_n = 1.f / ((float)cg * (float)inner);
  float s = 0.f;
  for (int cl = 0; cl < cg; ++cl) {
    const float ad = add != nullptr ? add[(long)b * C + g * cg + cl] : 0.f;
    for (int j = threadIdx.x; j < inner8; j += 256) {
      float v[8];
      gn_load8<IN_F32>(x, base + (long)cl * inner + j * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e] + ad;
    }
  }
  const float mean = block_sum(s, red) * inv_n;
  float q = 0.f;
  for (int cl = 0; cl < cg; ++cl) {
    const float ad = (add != nullptr ? add[(long)b * C + g * cg + cl] : 0.f) - mean;
    for (int j = threadIdx.x; j < inner8; j += 256) {
      float v[8];
      gn_load8<IN_F32>(x, base + (long)cl * inner + j * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[e] + ad;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(block_sum(q, red) * inv_n + eps);
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    for (int j = threadIdx.x; j < inner8; j += 256) {
      float v[8];
      const long o = base + (long)cl * inner + j * 8;
      gn_load8<IN_F32>(x, o, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float r = ((v[e] + ad) - mean) * rstd * ga + be;
        if (silu) r = silu_f(r);
        v[e] = r;
      }
      gn_store8(y, o, v);
    }
  }
}

// Backward: the per-channel sums (dgamma / dbeta partials, dadd) are reduced per WAVE inside the channel loop (shuffles
// only) and parked in LDS; the workgroup synchronises once per pass instead of 4-6 times per channel, so the loads of
// the next channel overlap the arithmetic of this one.  Channels per group <= GN_MAX_CG (the UNet's widest: 2560 / 32).
constexpr int GN_MAX_CG = 128;
template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_bwd_vec_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                         const float* __restrict__ add, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                         float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                         float* __restrict__ dadd, int C, int G, int inner, int silu, int affine_bs) {
  __shared__ float red[4];
  __shared__ float part[2][GN_MAX_CG][4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long base = ((long)b * C + (long)g * cg) * inner;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int j = threadIdx.x; j < inner8; j += 256) {
      const long o = base + (long)cl * inner + j * 8;
      float v[8], d[8];
      gn_load8<IN_F32>(x, o, v);
      gn_load8<false>(dy, o, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] + ad - mu) * rs;
        float dn = d[e];
        if (silu) dn *= silu_grad_f(xh * ga + be);
        sg += dn * xh;
        sb += dn;
      }
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (lane == 0) {
      part[0][cl][wave] = sg;
      part[1][cl][wave] = sb;
    }
  }
  __syncthreads();
  float t1 = 0.f, t2 = 0.f;  // this thread's channel: gamma * sum(dn), gamma * sum(dn * xhat)
  if ((int)threadIdx.x < cg) {
    const int c = g * cg + threadIdx.x;
    const float sg = (part[0][threadIdx.x][0] + part[0][threadIdx.x][1]) + (part[0][threadIdx.x][2] + part[0][threadIdx.x][3]);
    const float sb = (part[1][threadIdx.x][0] + part[1][threadIdx.x][1]) + (part[1][threadIdx.x][2] + part[1][threadIdx.x][3]);
    dgamma_part[(long)b * C + c] = sg;
    dbeta_part[(long)b * C + c] = sb;
    t1 = sb * gamma[c];
    t2 = sg * gamma[c];
  }
  const float s1 = block_sum(t1, red);
  const float s2 = block_sum(t2, red);
  const float inv_n = 1.f / ((float)cg * (float)inner);
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sd = 0.f;
    for (int j = threadIdx.x; j < inner8; j += 256) {
      const long o = base + (long)cl * inner + j * 8;
      float v[8], d[8];
      gn_load8<IN_F32>(x, o, v);
      gn_load8<false>(dy, o, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] + ad - mu) * rs;
        float dn = d[e];
        if (silu) dn *= silu_grad_f(xh * ga + be);
        const float r = rs * (dn * ga - s1 * inv_n - xh * s2 * inv_n);
        d[e] = r;
        sd += r;  // the fp32 value, as in the scalar kernel (the bf16 rounding happens at the store only)
      }
      gn_store8(dx, o, d);
    }
    if (dadd != nullptr) {
      sd = wave_sum(sd);
      if (lane == 0) part[0][cl][wave] = sd;  // pass-1 contents were consumed before the block_sum barriers above
    }
  }
  if (dadd != nullptr) {
    __syncthreads();
    if ((int)threadIdx.x < cg)
      dadd[(long)b * C + g * cg + threadIdx.x] =
          (part[0][threadIdx.x][0] + part[0][threadIdx.x][1]) + (part[0][threadIdx.x][2] + part[0][threadIdx.x][3]);
  }
}

// ---- GroupNorm, split form: a (sample, group) pair is cut into S slices along `inner`, one workgroup each ------------
// The vector kernels above run ONE workgroup per (sample, group): B * G = 32 workgroups for the batch-1 256^2 UNet step (a
// tenth of the chip: 61 + 61 launches took 11.7 + 17.8 ms of a 533 ms step, profiles/r03/prof_unet256_b1_summary.txt) and
// one 4-wave workgroup per CU at 64^2 x 8.  Split form: the statistics kernel writes one (mean, M2) pair per slice — a local
// two-pass over the slice (second pass out of L2) — and every workgroup of the apply kernel merges the S pairs of its group
// in slice order (Chan's parallel-variance update: deterministic, no cancellation) before it normalises its own slice.
// Backward: per-slice per-channel sums (dn * xhat, dn) -> merged per channel by the apply kernel, which also leaves per-slice
// sums of dx for the time-embedding gradient (reduced by gn_split_dadd_kernel).
template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_split_stats_kernel(const void* __restrict__ x, const float* __restrict__ add,
                                                             float* __restrict__ part, int C, int G, int inner, int S) {
  __shared__ float red[4];
  const int bg = blockIdx.x, sl = blockIdx.y;
  const int b = bg / G, g = bg - b * G;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  const int j0 = (int)((long)sl * inner8 / S), j1 = (int)((long)(sl + 1) * inner8 / S);
  const long base = ((long)b * C + (long)g * cg) * inner;
  const float inv_n = 1.f / ((float)cg * (float)(j1 - j0) * 8.f);
  float s = 0.f;
  for (int cl = 0; cl < cg; ++cl) {
    const float ad = add != nullptr ? add[(long)b * C + g * cg + cl] : 0.f;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
      float v[8];
      gn_load8<IN_F32>(x, base + (long)cl * inner + (long)j * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[e] + ad;
    }
  }
  const float mean = block_sum(s, red) * inv_n;
  float q = 0.f;
  for (int cl = 0; cl < cg; ++cl) {
    const float ad = (add != nullptr ? add[(long)b * C + g * cg + cl] : 0.f) - mean;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
      float v[8];
      gn_load8<IN_F32>(x, base + (long)cl * inner + (long)j * 8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[e] + ad;
        q += d * d;
      }
    }
  }
  q = block_sum(q, red);
  if (threadIdx.x == 0) {
    part[((long)bg * S + sl) * 2] = mean;
    part[((long)bg * S + sl) * 2 + 1] = q;
  }
}

// (mean, rstd) of group `bg` from its S slice pairs; every thread computes the same values (S <= 64 pairs: 512 bytes)
__device__ __forceinline__ void gn_merge_slices(const float* __restrict__ part, int bg, int S, int cg, int inner8, float eps,
                                                float& mean, float& rstd) {
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int sl = 0; sl < S; ++sl) {
    const int j0 = (int)((long)sl * inner8 / S), j1 = (int)((long)(sl + 1) * inner8 / S);
    const float nb = (float)cg * (float)(j1 - j0) * 8.f;
    const float mb = part[((long)bg * S + sl) * 2], qb = part[((long)bg * S + sl) * 2 + 1];
    const float nt = n + nb, d = mb - mu;
    mu += d * (nb / nt);
    m2 += qb + d * d * (n * nb / nt);
    n = nt;
  }
  mean = mu;
  rstd = rsqrtf(m2 / n + eps);
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_split_apply_kernel(const void* __restrict__ x, const float* __restrict__ add,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ part, bf16_t* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out, int C,
                                                             int G, int inner, float eps, int silu, int affine_bs, int S) {
  const int bg = blockIdx.x, sl = blockIdx.y;
  const int b = bg / G, g = bg - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  float mean, rstd;
  gn_merge_slices(part, bg, S, cg, inner8, eps, mean, rstd);
  if (sl == 0 && threadIdx.x == 0) {
    mean_out[bg] = mean;
    rstd_out[bg] = rstd;
  }
  const int j0 = (int)((long)sl * inner8 / S), j1 = (int)((long)(sl + 1) * inner8 / S);
  const long base = ((long)b * C + (long)g * cg) * inner;
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
      float v[8];
      const long o = base + (long)cl * inner + (long)j * 8;
      gn_load8<IN_F32>(x, o, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float r = ((v[e] + ad) - mean) * rstd * ga + be;
        if (silu) r = silu_f(r);
        v[e] = r;
      }
      gn_store8(y, o, v);
    }
  }
}

// part layout (backward): [bg][slice][3][cg]: sum(dn * xhat), sum(dn), sum(dx) per channel of the slice
template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_split_bwd_stats_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                                 const float* __restrict__ add, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ part, int C,
                                                                 int G, int inner, int silu, int affine_bs, int S) {
  __shared__ float wpart[2][GN_MAX_CG][4];
  const int bg = blockIdx.x, sl = blockIdx.y;
  const int b = bg / G, g = bg - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = (int)((long)sl * inner8 / S), j1 = (int)((long)(sl + 1) * inner8 / S);
  const long base = ((long)b * C + (long)g * cg) * inner;
  const float mu = mean[bg], rs = rstd[bg];
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
      const long o = base + (long)cl * inner + (long)j * 8;
      float v[8], d[8];
      gn_load8<IN_F32>(x, o, v);
      gn_load8<false>(dy, o, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] + ad - mu) * rs;
        float dn = d[e];
        if (silu) dn *= silu_grad_f(xh * ga + be);
        sg += dn * xh;
        sb += dn;
      }
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (lane == 0) {
      wpart[0][cl][wave] = sg;
      wpart[1][cl][wave] = sb;
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < cg) {
    float* dst = part + ((long)bg * S + sl) * 3 * cg;
    const int t = threadIdx.x;
    dst[t] = (wpart[0][t][0] + wpart[0][t][1]) + (wpart[0][t][2] + wpart[0][t][3]);
    dst[cg + t] = (wpart[1][t][0] + wpart[1][t][1]) + (wpart[1][t][2] + wpart[1][t][3]);
  }
}

template <bool IN_F32>
__global__ __launch_bounds__(256) void gn_split_bwd_apply_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ x,
                                                                 const float* __restrict__ add, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ part,
                                                                 bf16_t* __restrict__ dx, float* __restrict__ dgamma_part,
                                                                 float* __restrict__ dbeta_part, int want_dadd, int C, int G,
                                                                 int inner, int silu, int affine_bs, int S) {
  __shared__ float red[4];
  __shared__ float wpart[GN_MAX_CG][4];
  const int bg = blockIdx.x, sl = blockIdx.y;
  const int b = bg / G, g = bg - b * G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int cg = C / G;
  const int inner8 = inner >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float mu = mean[bg], rs = rstd[bg];
  float t1 = 0.f, t2 = 0.f;
  if ((int)threadIdx.x < cg) {  // this thread's channel: the S slice sums in slice order
    const int c = g * cg + threadIdx.x;
    float sg = 0.f, sb = 0.f;
    for (int k = 0; k < S; ++k) {
      const float* src = part + ((long)bg * S + k) * 3 * cg;
      sg += src[threadIdx.x];
      sb += src[cg + threadIdx.x];
    }
    if (sl == 0) {
      dgamma_part[(long)b * C + c] = sg;
      dbeta_part[(long)b * C + c] = sb;
    }
    t1 = sb * gamma[c];
    t2 = sg * gamma[c];
  }
  const float s1 = block_sum(t1, red);
  const float s2 = block_sum(t2, red);
  const float inv_n = 1.f / ((float)cg * (float)inner);
  const int j0 = (int)((long)sl * inner8 / S), j1 = (int)((long)(sl + 1) * inner8 / S);
  const long base = ((long)b * C + (long)g * cg) * inner;
  for (int cl = 0; cl < cg; ++cl) {
    const int c = g * cg + cl;
    const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
    const float ga = gamma[c], be = beta[c];
    float sd = 0.f;
    for (int j = j0 + threadIdx.x; j < j1; j += 256) {
      const long o = base + (long)cl * inner + (long)j * 8;
      float v[8], d[8];
      gn_load8<IN_F32>(x, o, v);
      gn_load8<false>(dy, o, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] + ad - mu) * rs;
        float dn = d[e];
        if (silu) dn *= silu_grad_f(xh * ga + be);
        const float r = rs * (dn * ga - s1 * inv_n - xh * s2 * inv_n);
        d[e] = r;
        sd += r;
      }
      gn_store8(dx, o, d);
    }
    if (want_dadd) {
      sd = wave_sum(sd);
      if (lane == 0) wpart[cl][wave] = sd;
    }
  }
  if (want_dadd) {
    __syncthreads();
    if ((int)threadIdx.x < cg)
      part[((long)bg * S + sl) * 3 * cg + 2 * cg + threadIdx.x] =
          (wpart[threadIdx.x][0] + wpart[threadIdx.x][1]) + (wpart[threadIdx.x][2] + wpart[threadIdx.x][3]);
  }
}

__global__ __launch_bounds__(128) void gn_split_dadd_kernel(const float* __restrict__ part, float* __restrict__ dadd, int C, int G,
                                                            int S) {
  const int bg = blockIdx.x;
  const int b = bg / G, g = bg - b * G;
  const int cg = C / G;
  if ((int)threadIdx.x < cg) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[((long)bg * S + k) * 3 * cg + 2 * cg + threadIdx.x];
    dadd[(long)b * C + g * cg + threadIdx.x] = s;
  }
}

// ---- GroupNorm on NHWC rows (round 5) ------------------------------------------------------------------------------------------
// x, y, dy, dx: bf16 [B][inner = H * W][C] — the layout the implicit-GEMM convolutions read and write, so a residual block that keeps
// its activations in it needs no NCHW <-> NHWC hop around every convolution (318 transposes per 64^2 x 8 UNet step).  Same
// arithmetic as the split NCHW kernels above (two-pass statistics per slice in fp32, Chan's merge in slice order, the additive
// per-(b, c) term in front, SiLU behind, per-sample partial sums of dgamma / dbeta, deterministic: no atomics), other indexing:
// a workgroup owns a slice of ROWS of one sample and all C channels; a thread owns KC slots of 8 consecutive channels (one
// 16-byte access per row and slot: rows are read and written as whole contiguous lines) and walks the rows rl, rl + RPP, ...;
// per-channel partial sums of the RPP row lanes meet in LDS in a fixed order, 32 threads fold channels into groups.
struct GnNhwcMap {
  int cg8, rpp, rl, nslot;  // 8-channel slots per row, row lanes, this thread's row lane, its number of slots (0: idle)
  int slot[2];              // its slots (channel = slot * 8)
};
__device__ __forceinline__ GnNhwcMap gn_nhwc_map(int C) {
  GnNhwcMap m;
  m.cg8 = C >> 3;
  const int t = threadIdx.x;
  if (m.cg8 >= 256) {
    m.rpp = 1; m.rl = 0;
    m.slot[0] = t; m.slot[1] = t + 256;
    m.nslot = (t < m.cg8 ? 1 : 0) + (t + 256 < m.cg8 ? 1 : 0);
  } else {
    m.rpp = 256 / m.cg8;
    m.rl = t / m.cg8;
    m.slot[0] = t - m.rl * m.cg8; m.slot[1] = 0;
    m.nslot = m.rl < m.rpp ? 1 : 0;
  }
  return m;
}
// rows [r0, r1) of slice sl of S over `inner` rows
__device__ __forceinline__ void gn_nhwc_rows(int inner, int S, int sl, int& r0, int& r1) {
  r0 = (int)((long)sl * inner / S);
  r1 = (int)((long)(sl + 1) * inner / S);
}
// The row walk of a thread: rows r0 + rl, r0 + rl + rpp, ... of the slice.  An idle thread (no slot) gets an EMPTY range instead of a
// branch around the body, and the loop is unrolled by 4: the four 16-byte loads of an unrolled body are independent and issue back
// to back (one load in flight per thread left these kernels at ~1.3 TB/s).
#define GN_NHWC_ROW_LOOP(r) _Pragma("unroll 4") for (int r = (m.nslot > 0 ? r0 + m.rl : r1); r < r1; r += m.rpp)
// per-channel values of all row lanes (acc[k][e] of every thread) -> chan[c] in LDS, summed over the row lanes in lane order
template <int KC>
__device__ __forceinline__ void gn_nhwc_fold(const GnNhwcMap& m, const float (&acc)[KC][8], float* lanes, float* chan, int C) {
  __syncthreads();  // (the buffers may still be read from a previous use)
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (k < m.nslot) {
#pragma unroll
      for (int e = 0; e < 8; ++e) lanes[m.rl * C + m.slot[k] * 8 + e] = acc[k][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
    for (int l = 0; l < m.rpp; ++l) s += lanes[l * C + c];
    chan[c] = s;
  }
  __syncthreads();
}

// ---- group form: one workgroup per (sample, group), ONE launch each way --------------------------------------------------------------
// With enough samples (B G >= ~128 workgroups: the 64^2 x 8 step has 256) a workgroup takes all rows of one group: its cpg
// channels are a 2 cpg-byte chunk of every row (20 .. 160 bytes), read as dwords — thread (rl, dw) owns the channel pair dw of the
// rows rl, rl + RPP, ... — so statistics, normalisation and the parameter-gradient sums all stay inside the workgroup: no slices,
// no workspace, no merge.  The 32 workgroups of a sample share every row's cache lines in L2.  cpg even, cpg <= 256.
// VW (round 5): a thread owns VW consecutive dwords (2 VW channels) of the chunk and reads them with ONE 4 VW-byte access — 4 when the
// chunk is a multiple of 16 bytes (cpg = 40, 80: C = 1 280, 2 560), 2 for multiples of 8 (cpg = 20, 60), else 1: a quarter / half of
// the memory instructions for the same bytes (the chunked accesses are what these kernels spend their time on, twice as much
// inside the step as alone).
template <int VW> struct GnVecT;
template <> struct GnVecT<1> { typedef unsigned T; };
template <> struct GnVecT<2> { typedef u32x2 T; };
template <> struct GnVecT<4> { typedef u32x4 T; };
template <int VW> struct GnWords {
  unsigned w[VW];
};
template <int VW> __device__ __forceinline__ GnWords<VW> gn_ldw(const bf16_t* p) {
  union { typename GnVecT<VW>::T v; GnWords<VW> w; } u;
  u.v = *reinterpret_cast<const typename GnVecT<VW>::T*>(p);
  return u.w;
}
template <int VW> __device__ __forceinline__ void gn_stw(bf16_t* p, const GnWords<VW>& w) {
  union { typename GnVecT<VW>::T v; GnWords<VW> w; } u;
  u.w = w;
  *reinterpret_cast<typename GnVecT<VW>::T*>(p) = u.v;
}
struct GnGroupMap {
  int vpr, rpp, rl, vs;  // vector slots per row chunk, row lanes, this thread's row lane / slot
  bool active;
};
template <int VW> __device__ __forceinline__ GnGroupMap gn_group_map(int cpg) {
  GnGroupMap m;
  m.vpr = (cpg >> 1) / VW;
  m.rpp = 256 / m.vpr;
  m.rl = threadIdx.x / m.vpr;
  m.vs = threadIdx.x - m.rl * m.vpr;
  m.active = m.rl < m.rpp;
  return m;
}

// RPT > 0: the thread's rows (at most RPT / VW of them, chosen by the launcher) stay in REGISTERS between the passes — one global read
// and one write per element instead of three reads and a write; RPT = 0: every pass re-reads (out of L2).
template <int RPT, int VW>
__global__ __launch_bounds__(256) void gn_nhwc_group_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ add,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                bf16_t* __restrict__ y, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, int C, int G, int inner, float eps, int silu,
                                                                int affine_bs) {
  __shared__ float red[4];
  constexpr int NC = RPT > 0 ? RPT / VW : 1;
  constexpr int CH = 2 * VW;  // channels per thread
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const GnGroupMap m = gn_group_map<VW>(cpg);
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int c0 = g * cpg + CH * m.vs;
  const long base = (long)b * inner * C + c0;
  float a[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) a[e] = (m.active && add != nullptr) ? add[(long)b * C + c0 + e] : 0.f;
  const float n = (float)cpg * (float)inner;
  auto row_sum = [&](const GnWords<VW>& w) {
    float t = 0.f;
#pragma unroll
    for (int v = 0; v < VW; ++v) t += (bf16lo(w.w[v]) + a[2 * v]) + (bf16hi(w.w[v]) + a[2 * v + 1]);
    return t;
  };
  GnWords<VW> cache[NC];
  float s = 0.f;
  if (RPT > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int r = m.rl + i * m.rpp;
#pragma unroll
      for (int v = 0; v < VW; ++v) cache[i].w[v] = 0u;
      if (m.active && r < inner) {
        cache[i] = gn_ldw<VW>(x + base + (long)r * C);
        s += row_sum(cache[i]);
      }
    }
  } else if (m.active) {
    for (int r = m.rl; r < inner; r += m.rpp) s += row_sum(gn_ldw<VW>(x + base + (long)r * C));
  }
  const float mean = block_sum(s, red) / n;
  auto row_sq = [&](const GnWords<VW>& w) {
    float t = 0.f;
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      const float d0 = bf16lo(w.w[v]) + a[2 * v] - mean, d1 = bf16hi(w.w[v]) + a[2 * v + 1] - mean;
      t += d0 * d0 + d1 * d1;
    }
    return t;
  };
  float q = 0.f;
  if (RPT > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int r = m.rl + i * m.rpp;
      if (m.active && r < inner) q += row_sq(cache[i]);
    }
  } else if (m.active) {
    for (int r = m.rl; r < inner; r += m.rpp) q += row_sq(gn_ldw<VW>(x + base + (long)r * C));
  }
  const float rstd = rsqrtf(block_sum(q, red) / n + eps);
  if (threadIdx.x == 0) {
    mean_out[blockIdx.x] = mean;
    rstd_out[blockIdx.x] = rstd;
  }
  if (!m.active) return;
  float k[CH], o[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    k[e] = rstd * gamma[c0 + e];
    o[e] = beta[c0 + e] + (a[e] - mean) * k[e];
  }
  auto emit = [&](int r, const GnWords<VW>& w) {
    GnWords<VW> out;
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      float v0 = fmaf(bf16lo(w.w[v]), k[2 * v], o[2 * v]), v1 = fmaf(bf16hi(w.w[v]), k[2 * v + 1], o[2 * v + 1]);
      if (silu) {
        v0 = silu_f(v0);
        v1 = silu_f(v1);
      }
      out.w[v] = pack_bf16x2(v0, v1);
    }
    gn_stw<VW>(y + base + (long)r * C, out);
  };
  if (RPT > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int r = m.rl + i * m.rpp;
      if (r < inner) emit(r, cache[i]);
    }
  } else {
    for (int r = m.rl; r < inner; r += m.rpp) emit(r, gn_ldw<VW>(x + base + (long)r * C));
  }
}

// LDS: lanes [rpp][cpg] (<= 512 VW floats) + ch [2][cpg]
template <int RPT, int VW>
__global__ __launch_bounds__(256) void gn_nhwc_group_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                const float* __restrict__ add, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, bf16_t* __restrict__ dx,
                                                                float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                                                float* __restrict__ dadd, int C, int G, int inner, int silu, int affine_bs) {
  __shared__ float lanes[512 * VW];
  __shared__ float chA[256], chB[256], sums[2];
  constexpr int NC = RPT > 0 ? RPT / VW : 1;
  constexpr int CH = 2 * VW;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int cpg = C / G;
  const GnGroupMap m = gn_group_map<VW>(cpg);
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  const int c0 = g * cpg + CH * m.vs;
  const long base = (long)b * inner * C + c0;
  const float mu = mean[blockIdx.x], rs = rstd[blockIdx.x];
  float sh[CH], ga[CH], be[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) {
    sh[e] = -mu;
    ga[e] = be[e] = 0.f;
    if (m.active) {
      if (add != nullptr) sh[e] += add[(long)b * C + c0 + e];
      ga[e] = gamma[c0 + e];
      be[e] = beta[c0 + e];
    }
  }
  // a per-channel quantity of every row lane -> ch[c], summed over the row lanes in lane order
  auto fold = [&](const float (&v)[CH], float* ch) {
    __syncthreads();
    if (m.active) {
#pragma unroll
      for (int e = 0; e < CH; ++e) lanes[m.rl * cpg + CH * m.vs + e] = v[e];
    }
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
      float t = 0.f;
      for (int l = 0; l < m.rpp; ++l) t += lanes[l * cpg + threadIdx.x];
      ch[threadIdx.x] = t;
    }
    __syncthreads();
  };
  // (xhat, dn) of the thread's 2 VW elements of a row from the raw words
  auto terms = [&](const GnWords<VW>& wx, const GnWords<VW>& wd, float (&xh)[CH], float (&d)[CH]) {
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      xh[2 * v] = (bf16lo(wx.w[v]) + sh[2 * v]) * rs;
      xh[2 * v + 1] = (bf16hi(wx.w[v]) + sh[2 * v + 1]) * rs;
      d[2 * v] = bf16lo(wd.w[v]);
      d[2 * v + 1] = bf16hi(wd.w[v]);
    }
    if (silu) {
#pragma unroll
      for (int e = 0; e < CH; ++e) d[e] *= silu_grad_f(fmaf(xh[e], ga[e], be[e]));
    }
  };
  GnWords<VW> cx[NC], cd[NC];
  float A[CH], Bs[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) A[e] = Bs[e] = 0.f;
  auto accumulate = [&](const GnWords<VW>& wx, const GnWords<VW>& wd) {
    float xh[CH], d[CH];
    terms(wx, wd, xh, d);
#pragma unroll
    for (int e = 0; e < CH; ++e) {
      A[e] += d[e] * xh[e];
      Bs[e] += d[e];
    }
  };
  if (RPT > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int r = m.rl + i * m.rpp;
#pragma unroll
      for (int v = 0; v < VW; ++v) cx[i].w[v] = cd[i].w[v] = 0u;
      if (m.active && r < inner) {
        cx[i] = gn_ldw<VW>(x + base + (long)r * C);
        cd[i] = gn_ldw<VW>(dy + base + (long)r * C);
        accumulate(cx[i], cd[i]);
      }
    }
  } else if (m.active) {
    for (int r = m.rl; r < inner; r += m.rpp) accumulate(gn_ldw<VW>(x + base + (long)r * C), gn_ldw<VW>(dy + base + (long)r * C));
  }
  fold(A, chA);
  fold(Bs, chB);
  if ((int)threadIdx.x < cpg) {
    dgamma_part[(long)b * C + g * cpg + threadIdx.x] = chA[threadIdx.x];
    dbeta_part[(long)b * C + g * cpg + threadIdx.x] = chB[threadIdx.x];
  }
  if (threadIdx.x == 0) {
    float s1 = 0.f, s2 = 0.f;  // sum dn gamma, sum dn gamma xhat
    for (int c = 0; c < cpg; ++c) {
      s1 += chB[c] * gamma[g * cpg + c];
      s2 += chA[c] * gamma[g * cpg + c];
    }
    const float inv_n = 1.f / ((float)cpg * (float)inner);
    sums[0] = s1 * inv_n;
    sums[1] = s2 * inv_n;
  }
  __syncthreads();
  const float m1 = sums[0], m2 = sums[1];
  float D[CH];
#pragma unroll
  for (int e = 0; e < CH; ++e) D[e] = 0.f;
  auto emit = [&](int r, const GnWords<VW>& wx, const GnWords<VW>& wd) {
    float xh[CH], d[CH];
    terms(wx, wd, xh, d);
    GnWords<VW> out;
#pragma unroll
    for (int v = 0; v < VW; ++v) {
      const float o0 = rs * (d[2 * v] * ga[2 * v] - m1 - xh[2 * v] * m2);
      const float o1 = rs * (d[2 * v + 1] * ga[2 * v + 1] - m1 - xh[2 * v + 1] * m2);
      out.w[v] = pack_bf16x2(o0, o1);
      D[2 * v] += o0;
      D[2 * v + 1] += o1;
    }
    gn_stw<VW>(dx + base + (long)r * C, out);
  };
  if (RPT > 0) {
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int r = m.rl + i * m.rpp;
      if (m.active && r < inner) emit(r, cx[i], cd[i]);
    }
  } else if (m.active) {
    for (int r = m.rl; r < inner; r += m.rpp) emit(r, gn_ldw<VW>(x + base + (long)r * C), gn_ldw<VW>(dy + base + (long)r * C));
  }
  if (dadd != nullptr) {
    fold(D, chA);
    if ((int)threadIdx.x < cpg) dadd[(long)b * C + g * cpg + threadIdx.x] = chA[threadIdx.x];
  }
}

// ---- slice form (few samples: B G too small to fill the chip, e.g. 256^2 x 1) --------------------------------------------------------
// LDS: lanes [rpp][C] + chan [C] + grp [2 G] floats (rpp * C <= 2048 + C: at most 4 C floats)
template <int KC>
__global__ __launch_bounds__(256) void gn_nhwc_stats_kernel(const bf16_t* __restrict__ x, const float* __restrict__ add,
                                                            float* __restrict__ part, int C, int G, int inner, int S) {
  extern __shared__ float gsm[];
  const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
  const GnNhwcMap m = gn_nhwc_map(C);
  float* lanes = gsm;
  float* chan = gsm + m.rpp * C;
  float* grp = chan + C;
  const int cpg = C / G;
  int r0, r1;
  gn_nhwc_rows(inner, S, sl, r0, r1);
  const bf16_t* xb = x + (long)b * inner * C;
  float ad[KC][8], acc[KC][8];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ad[k][e] = (add != nullptr && k < m.nslot) ? add[(long)b * C + m.slot[k] * 8 + e] : 0.f;
      acc[k][e] = 0.f;
    }
  GN_NHWC_ROW_LOOP(r) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k == 0 || k < m.nslot) {
        float v[8];
        gn_load8<false>(xb, (long)r * C + m.slot[k] * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[k][e] += v[e] + ad[k][e];
      }
  }
  gn_nhwc_fold<KC>(m, acc, lanes, chan, C);
  const float cnt = (float)cpg * (float)(r1 - r0);
  if ((int)threadIdx.x < G) {
    float s = 0.f;
    for (int c = threadIdx.x * cpg; c < (int)(threadIdx.x + 1) * cpg; ++c) s += chan[c];
    grp[threadIdx.x] = cnt > 0.f ? s / cnt : 0.f;
  }
  __syncthreads();
  float mu[KC][8];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[k][e] = k < m.nslot ? ad[k][e] - grp[(m.slot[k] * 8 + e) / cpg] : 0.f;  // v + add - mean
      acc[k][e] = 0.f;
    }
  GN_NHWC_ROW_LOOP(r) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k == 0 || k < m.nslot) {
        float v[8];
        gn_load8<false>(xb, (long)r * C + m.slot[k] * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[e] + mu[k][e];
          acc[k][e] += d * d;
        }
      }
  }
  gn_nhwc_fold<KC>(m, acc, lanes, chan, C);
  if ((int)threadIdx.x < G) {
    float q = 0.f;
    for (int c = threadIdx.x * cpg; c < (int)(threadIdx.x + 1) * cpg; ++c) q += chan[c];
    float* o = part + (((long)b * S + sl) * G + threadIdx.x) * 2;
    o[0] = grp[threadIdx.x];
    o[1] = q;
  }
}

// (mean, rstd) of every group of sample b from the S slice pairs, by threads 0 .. G - 1, into grp[0 .. G) / grp[G .. 2 G)
__device__ __forceinline__ void gn_nhwc_merge(const float* __restrict__ part, int b, int S, int G, int cpg, int inner, float eps,
                                              float* grp) {
  if ((int)threadIdx.x < G) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    // (no branch in the body and unrolled: the S pair loads of a thread are independent and issue ahead of the serial merge)
#pragma unroll 8
    for (int sl = 0; sl < S; ++sl) {
      int r0, r1;
      gn_nhwc_rows(inner, S, sl, r0, r1);
      const float nb = (float)cpg * (float)(r1 - r0);
      const float* o = part + (((long)b * S + sl) * G + threadIdx.x) * 2;
      const float o0 = o[0], o1 = o[1];
      const bool some = nb > 0.f;  // (an empty slice wrote (0, 0))
      const float nt = n + nb, d = o0 - mean;
      const float w = some ? nb / nt : 0.f;
      mean += d * w;
      m2 += some ? o1 + d * d * (n * w) : 0.f;
      n = nt;
    }
    grp[threadIdx.x] = mean;
    grp[G + threadIdx.x] = rsqrtf(m2 / n + eps);
  }
  __syncthreads();
}

// one workgroup per sample merges the slice pairs ONCE (the apply kernels read the result: merging inside every apply workgroup was
// S^2 G loads per sample — 268 MB of L2 reads per call at 256^2 x 1)
__global__ void gn_nhwc_merge_kernel(const float* __restrict__ part, float* __restrict__ mean_out, float* __restrict__ rstd_out, int G,
                                     int cpg, int inner, float eps, int S) {
  __shared__ float grp[2 * 64];
  const int b = blockIdx.x;
  gn_nhwc_merge(part, b, S, G, cpg, inner, eps, grp);
  if ((int)threadIdx.x < G) {
    mean_out[b * G + threadIdx.x] = grp[threadIdx.x];
    rstd_out[b * G + threadIdx.x] = grp[G + threadIdx.x];
  }
}

template <int KC>
__global__ __launch_bounds__(256) void gn_nhwc_apply_kernel(const bf16_t* __restrict__ x, const float* __restrict__ add,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ y, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, int C, int G, int inner, int silu,
                                                            int affine_bs, int S) {
  __shared__ float grp[2 * 64];
  const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
  const GnNhwcMap m = gn_nhwc_map(C);
  const int cpg = C / G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  if ((int)threadIdx.x < G) {
    grp[threadIdx.x] = mean_in[b * G + threadIdx.x];
    grp[G + threadIdx.x] = rstd_in[b * G + threadIdx.x];
  }
  __syncthreads();
  float a[KC][8], bb[KC][8];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[k][e] = bb[k][e] = 0.f;
      if (k < m.nslot) {
        const int c = m.slot[k] * 8 + e, g = c / cpg;
        const float ad = add != nullptr ? add[(long)b * C + c] : 0.f;
        a[k][e] = grp[G + g] * gamma[c];
        bb[k][e] = beta[c] + (ad - grp[g]) * a[k][e];
      }
    }
  int r0, r1;
  gn_nhwc_rows(inner, S, sl, r0, r1);
  const bf16_t* xb = x + (long)b * inner * C;
  bf16_t* yb = y + (long)b * inner * C;
  GN_NHWC_ROW_LOOP(r) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k == 0 || k < m.nslot) {
        float v[8];
        const long off = (long)r * C + m.slot[k] * 8;
        gn_load8<false>(xb, off, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float o = fmaf(v[e], a[k][e], bb[k][e]);
          v[e] = silu ? silu_f(o) : o;
        }
        gn_store8(yb, off, v);
      }
  }
}

// backward, statistics: per slice and channel  A_c = sum dn * xhat,  B_c = sum dn  (dn = dy [* SiLU'(xhat gamma + beta)])
// part: [B][S][2][C]
template <int KC>
__global__ __launch_bounds__(256) void gn_nhwc_bwd_stats_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                const float* __restrict__ add, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, float* __restrict__ part, int C, int G,
                                                                int inner, int silu, int affine_bs, int S) {
  extern __shared__ float gsm[];
  const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
  const GnNhwcMap m = gn_nhwc_map(C);
  float* lanes = gsm;
  float* chan = gsm + m.rpp * C;
  const int cpg = C / G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  float sh[KC][8], rs[KC][8], ga[KC][8], be[KC][8], accA[KC][8], accB[KC][8];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sh[k][e] = rs[k][e] = ga[k][e] = be[k][e] = accA[k][e] = accB[k][e] = 0.f;
      if (k < m.nslot) {
        const int c = m.slot[k] * 8 + e, g = c / cpg;
        rs[k][e] = rstd[b * G + g];
        sh[k][e] = (add != nullptr ? add[(long)b * C + c] : 0.f) - mean[b * G + g];  // xhat = (x + sh) * rs
        ga[k][e] = gamma[c];
        be[k][e] = beta[c];
      }
    }
  int r0, r1;
  gn_nhwc_rows(inner, S, sl, r0, r1);
  const bf16_t* xb = x + (long)b * inner * C;
  const bf16_t* dyb = dy + (long)b * inner * C;
  GN_NHWC_ROW_LOOP(r) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k == 0 || k < m.nslot) {
        float v[8], d[8];
        const long off = (long)r * C + m.slot[k] * 8;
        gn_load8<false>(xb, off, v);
        gn_load8<false>(dyb, off, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] + sh[k][e]) * rs[k][e];
          float dn = d[e];
          if (silu) dn *= silu_grad_f(fmaf(xh, ga[k][e], be[k][e]));
          accA[k][e] += dn * xh;
          accB[k][e] += dn;
        }
      }
  }
  float* o = part + ((long)b * S + sl) * 2 * C;
  gn_nhwc_fold<KC>(m, accA, lanes, chan, C);
  for (int c = threadIdx.x; c < C; c += 256) o[c] = chan[c];
  gn_nhwc_fold<KC>(m, accB, lanes, chan, C);
  for (int c = threadIdx.x; c < C; c += 256) o[C + c] = chan[c];
}

// backward, apply: merges the slices per channel (slice order), writes the per-sample dgamma / dbeta partial rows (slice 0's
// workgroup), dx = rstd (dn gamma - s1 / n - xhat s2 / n), and per slice and channel sum(dx) for the time-embedding gradient
// LDS: lanes [rpp][C] + chA [C] + chB [C] + grp [2 G]
template <int KC>
__global__ __launch_bounds__(256) void gn_nhwc_bwd_apply_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                                const float* __restrict__ add, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ part,
                                                                bf16_t* __restrict__ dx, const float* __restrict__ dgamma_part,
                                                                const float* __restrict__ dbeta_part, float* __restrict__ dadd_part, int C,
                                                                int G, int inner, int silu, int affine_bs, int S) {
  extern __shared__ float gsm[];
  const int b = blockIdx.x / S, sl = blockIdx.x - b * S;
  const GnNhwcMap m = gn_nhwc_map(C);
  float* lanes = gsm;
  float* chA = gsm + m.rpp * C;
  float* chB = chA + C;
  float* grp = chB + C;
  const int cpg = C / G;
  gamma += (long)b * affine_bs;
  beta += (long)b * affine_bs;
  (void)part;
  for (int c = threadIdx.x; c < C; c += 256) {  // the per-sample sums gn_nhwc_bwd_merge_kernel left in dgamma_part / dbeta_part
    chA[c] = dgamma_part[(long)b * C + c];
    chB[c] = dbeta_part[(long)b * C + c];
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    float s1 = 0.f, s2 = 0.f;  // sum dn gamma, sum dn gamma xhat
    for (int c = threadIdx.x * cpg; c < (int)(threadIdx.x + 1) * cpg; ++c) {
      s1 += chB[c] * gamma[c];
      s2 += chA[c] * gamma[c];
    }
    const float inv_n = 1.f / ((float)cpg * (float)inner);
    grp[threadIdx.x] = s1 * inv_n;
    grp[G + threadIdx.x] = s2 * inv_n;
  }
  __syncthreads();
  float sh[KC][8], rs[KC][8], ga[KC][8], be[KC][8], m1[KC][8], m2[KC][8], accD[KC][8];
#pragma unroll
  for (int k = 0; k < KC; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sh[k][e] = rs[k][e] = ga[k][e] = be[k][e] = m1[k][e] = m2[k][e] = accD[k][e] = 0.f;
      if (k < m.nslot) {
        const int c = m.slot[k] * 8 + e, g = c / cpg;
        rs[k][e] = rstd[b * G + g];
        sh[k][e] = (add != nullptr ? add[(long)b * C + c] : 0.f) - mean[b * G + g];
        ga[k][e] = gamma[c];
        be[k][e] = beta[c];
        m1[k][e] = grp[g];
        m2[k][e] = grp[G + g];
      }
    }
  int r0, r1;
  gn_nhwc_rows(inner, S, sl, r0, r1);
  const bf16_t* xb = x + (long)b * inner * C;
  const bf16_t* dyb = dy + (long)b * inner * C;
  bf16_t* dxb = dx + (long)b * inner * C;
  GN_NHWC_ROW_LOOP(r) {
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k == 0 || k < m.nslot) {
        float v[8], d[8];
        const long off = (long)r * C + m.slot[k] * 8;
        gn_load8<false>(xb, off, v);
        gn_load8<false>(dyb, off, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (v[e] + sh[k][e]) * rs[k][e];
          float dn = d[e];
          if (silu) dn *= silu_grad_f(fmaf(xh, ga[k][e], be[k][e]));
          const float o = rs[k][e] * (dn * ga[k][e] - m1[k][e] - xh * m2[k][e]);
          v[e] = o;
          accD[k][e] += o;
        }
        gn_store8(dxb, off, v);
      }
  }
  if (dadd_part != nullptr) {
    gn_nhwc_fold<KC>(m, accD, lanes, chA, C);  // (chA / chB are no longer needed: every thread holds its coefficients in registers)
    float* o = dadd_part + ((long)b * S + sl) * C;
    for (int c = threadIdx.x; c < C; c += 256) o[c] = chA[c];
  }
}

#undef GN_NHWC_ROW_LOOP

// dgamma_part[b][c] / dbeta_part[b][c] = the slice sums of A_c / B_c added in slice order, once per sample (S C loads per sample)
__global__ void gn_nhwc_bwd_merge_kernel(const float* __restrict__ part, float* __restrict__ dgamma_part, float* __restrict__ dbeta_part,
                                         int C, int S) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sa = 0.f, sb = 0.f;
#pragma unroll 8
  for (int sl = 0; sl < S; ++sl) {  // (same order of additions; unrolled so that the loads run ahead of them)
    const float* o = part + ((long)b * S + sl) * 2 * C;
    sa += o[c];
    sb += o[C + c];
  }
  dgamma_part[(long)b * C + c] = sa;
  dbeta_part[(long)b * C + c] = sb;
}

// dadd[b][c] = sum over the slices of sum(dx), slice order
__global__ void gn_nhwc_dadd_kernel(const float* __restrict__ part, float* __restrict__ dadd, int C, int S) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
#pragma unroll 8
  for (int sl = 0; sl < S; ++sl) s += part[((long)b * S + sl) * C + c];
  dadd[(long)b * C + c] = s;
}

// ---- 3x3 filter repacking for the implicit-GEMM convolution -----------------------------------------------------------
// w bf16 [Cout][Cin][3][3] (the reference's layout, arena-backed shadow of the fp32 master) ->
//   FWD: wk[co][tap][c]        = w[co][c][tap]         (tap = ky*3 + kx; a K-step of the GEMM = 32 channels of one tap)
//   BWD: wr[c][tap][co]        = w[co][c][8 - tap]     (rotated by 180 degrees, channels swapped: the dX convolution)
// One lane owns 8 channels of one filter: 72 consecutive bf16 (144 B, nine 16-byte loads), transposed in registers.
template <bool BWD>
__device__ __forceinline__ void conv3x3_pack_item(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int Cout, int Cin, long idx) {
  const int c8n = Cin >> 3;
  // FWD: adjacent lanes take adjacent channel blocks (16-byte stores, contiguous over c);
  // BWD: adjacent lanes take adjacent filters (2-byte stores, contiguous over co)
  const int co = BWD ? (int)(idx % Cout) : (int)(idx / c8n);
  const int c0 = (BWD ? (int)(idx / Cout) : (int)(idx % c8n)) << 3;
  union { u32x4 v[9]; bf16_t e[72]; } u;
  const u32x4* src = reinterpret_cast<const u32x4*>(w + ((long)co * Cin + c0) * 9);
#pragma unroll
  for (int i = 0; i < 9; ++i) u.v[i] = src[i];
  if (!BWD) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      union { u32x4 v; bf16_t e[8]; } o;
#pragma unroll
      for (int cl = 0; cl < 8; ++cl) o.e[cl] = u.e[cl * 9 + tap];
      *reinterpret_cast<u32x4*>(out + ((long)co * 9 + tap) * Cin + c0) = o.v;
    }
  } else {
#pragma unroll
    for (int cl = 0; cl < 8; ++cl)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
        out[((long)(c0 + cl) * 9 + tap) * Cout + co] = u.e[cl * 9 + (8 - tap)];
  }
}

template <bool BWD>
__global__ __launch_bounds__(256) void conv3x3_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
                                                           int Cout, int Cin) {
  const long total = (long)Cout * (Cin >> 3);
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x)
    conv3x3_pack_item<BWD>(w, out, Cout, Cin, idx);
}

// The filter matrices of MANY convolutions in one launch (round 6, late: the UNet packed both forms of its 48 3x3 convolutions with 96
// launches of 12-15 us on the critical queue of every step): the problem table travels in the kernel arguments.
constexpr int PK_MAX = 64;
struct PackArgs {
  const bf16_t* w[PK_MAX];
  bf16_t* out[PK_MAX];
  int cout[PK_MAX], cin[PK_MAX], rot[PK_MAX];
  int blk0[PK_MAX + 1];  // first 256-item block of problem i; blk0[count] = number of blocks
  int count;
};
__global__ __launch_bounds__(256) void conv3x3_pack_grouped_kernel(PackArgs a) {
  int p = 0;
  while (p + 1 < a.count && (int)blockIdx.x >= a.blk0[p + 1]) ++p;
  const long idx = (long)((int)blockIdx.x - a.blk0[p]) * 256 + threadIdx.x;
  const int Cout = a.cout[p], Cin = a.cin[p];
  if (idx >= (long)Cout * (Cin >> 3)) return;
  if (a.rot[p]) conv3x3_pack_item<true>(a.w[p], a.out[p], Cout, Cin, idx);
  else conv3x3_pack_item<false>(a.w[p], a.out[p], Cout, Cin, idx);
}

// ---- SiLU on small f32 vectors (the time embedding), nearest x2 up-sampling, 2x2 average pooling ------------------
template <bool BWD>
__global__ void silu_f32_kernel(const float* __restrict__ a, const float* __restrict__ x, float* __restrict__ out, long n) {
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += step)
    out[i] = BWD ? a[i] * silu_grad_f(x[i]) : silu_f(a[i]);
}


// ---- the UNet's per-block time-embedding projections, ALL blocks in one launch each way (round 6, late) ------------------------------
// Every ResidualBlock computes Linear_i(SiLU(time_net)) from the SAME [B, K] time embedding (residual.py:226-239): per block and per step
// that was a SiLU launch, a cast, an M = B GEMM (a 128-row tile for 8 rows), and in backward a cast, a GEMM, a copy and a SiLU' launch
// plus autograd's add into the shared gradient — ~160 launches on the critical queue of the 64^2 x 8 step for 0.05 ms of arithmetic.
// Here: ONE forward launch (problem table in the kernel arguments: up to 32 weight matrices), TWO backward launches (per-column-block
// partial sums of dY_i W_i in a fixed order, then their sum times SiLU': no atomics).  Arithmetic as the per-block path has it: bf16
// SiLU(emb) and bf16 weights, f32 accumulation, f32 output; dY rounded to bf16 before it multiplies (what the GEMMs did).
// Layout: a workgroup = 64 output columns (one per lane: no cross-lane reduction) x the reduction split over its 4 waves.
constexpr int TP_MAX = 32;
struct TimeProjArgs {
  const bf16_t* w[TP_MAX];   // [N_i, K] bf16
  const float* vec[TP_MAX];  // forward: bias [N_i] or null; backward: dY_i [B, N_i] f32 or null (no gradient for this output)
  float* out[TP_MAX];        // forward: out_i [B, N_i] f32
  bf16_t* aux[TP_MAX];       // backward: dY_i as bf16 [B, N_i] (for the weight-gradient GEMMs) or null
  int n[TP_MAX];
  int blk0[TP_MAX + 1];      // first 64-column block of problem i; blk0[count] = number of blocks
  int count;
};

__device__ __forceinline__ int tp_problem(const TimeProjArgs& a, int blk) {
  int p = 0;
  while (p + 1 < a.count && blk >= a.blk0[p + 1]) ++p;
  return p;
}

__global__ __launch_bounds__(256) void time_proj_fwd_kernel(const float* __restrict__ emb, int B, int K, TimeProjArgs a,
                                                            bf16_t* __restrict__ t_out) {
  extern __shared__ __attribute__((aligned(16))) char tp_smem[];
  bf16_t* ts = reinterpret_cast<bf16_t*>(tp_smem);                        // [8][K] bf16: SiLU(emb) of this batch chunk
  float* red = reinterpret_cast<float*>(tp_smem + (size_t)8 * K * 2);     // [4][8][64] partial sums of the four K quarters
  const int b0 = blockIdx.y * 8;
  for (int idx = threadIdx.x; idx < 8 * K; idx += 256) {
    const int r = idx / K, k = idx - r * K;
    const bool ok = b0 + r < B;
    const bf16_t h = f32_to_bf16(ok ? silu_f(emb[(long)(b0 + r) * K + k]) : 0.f);
    ts[idx] = h;
    if (blockIdx.x == 0 && ok) t_out[(long)(b0 + r) * K + k] = h;
  }
  __syncthreads();
  const int p = tp_problem(a, blockIdx.x);
  const int N = a.n[p];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x - a.blk0[p]) * 64 + lane;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  if (col < N) {
    const bf16_t* wrow = a.w[p] + (long)col * K;
    const int kq = K / 4;
#pragma unroll 2
    for (int k = wave * kq; k < (wave + 1) * kq; k += 8) {
      const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wrow + k);
      float wf[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) wf[e] = bf16_to_f32((bf16_t)wv[e]);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const bf16x8 tv = *reinterpret_cast<const bf16x8*>(ts + r * K + k);  // the same address in every lane: a broadcast
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[r] = fmaf(wf[e], bf16_to_f32((bf16_t)tv[e]), acc[r]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) red[(wave * 8 + r) * 64 + lane] = acc[r];
  __syncthreads();
  if (wave == 0 && col < N) {
    const float bias = a.vec[p] != nullptr ? a.vec[p][col] : 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (b0 + r >= B) break;
      const float s = ((red[(0 * 8 + r) * 64 + lane] + red[(1 * 8 + r) * 64 + lane]) + red[(2 * 8 + r) * 64 + lane]) + red[(3 * 8 + r) * 64 + lane];
      a.out[p][(long)(b0 + r) * N + col] = s + bias;
    }
  }
}

// backward, stage 1: partial[blk][b][k] = sum over the 64 columns n of block blk of bf16(dY[b][n]) * W[n][k]   (thread = one k)
__global__ __launch_bounds__(256) void time_proj_bwd_part_kernel(int B, int K, TimeProjArgs a, float* __restrict__ partial, int Bpad) {
  __shared__ float dys[8][64];
  const int blk = blockIdx.x;
  const int p = tp_problem(a, blk);
  const int N = a.n[p];
  const int c0 = (blk - a.blk0[p]) * 64;
  const int k = blockIdx.y * 256 + threadIdx.x;
  const float* dy = a.vec[p];
  for (int b0 = 0; b0 < B; b0 += 8) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < 8 * 64; idx += 256) {
      const int r = idx >> 6, c = idx & 63;
      float v = 0.f;
      if (dy != nullptr && b0 + r < B && c0 + c < N) {
        const bf16_t h = f32_to_bf16(dy[(long)(b0 + r) * N + c0 + c]);
        v = bf16_to_f32(h);
        if (blockIdx.y == 0 && a.aux[p] != nullptr) a.aux[p][(long)(b0 + r) * N + c0 + c] = h;
      }
      dys[r][c] = v;
    }
    __syncthreads();
    if (k < K) {
      float acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = 0.f;
      if (dy != nullptr) {
        const int cn = min(64, N - c0);
        const bf16_t* wp = a.w[p] + (long)c0 * K + k;
        for (int c = 0; c < cn; ++c) {
          const float wv = bf16_to_f32(wp[(long)c * K]);
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[r] = fmaf(dys[r][c], wv, acc[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) partial[((long)blk * Bpad + b0 + r) * K + k] = acc[r];
    }
  }
}

// backward, stage 2: d_emb[b][k] = SiLU'(emb[b][k]) * sum over blocks (in block order) of partial[blk][b][k]
__global__ void time_proj_bwd_sum_kernel(const float* __restrict__ emb, const float* __restrict__ partial, float* __restrict__ d_emb,
                                         int B, int K, int Bpad, int nblk) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * K) return;
  const int b = (int)(i / K), k = (int)(i - (long)b * K);
  float s = 0.f;
  for (int blk = 0; blk < nblk; ++blk) s += partial[((long)blk * Bpad + b) * K + k];
  d_emb[i] = s * silu_grad_f(emb[i]);
}

// fwd: y[b,c,2h+dy,2w+dx] = x[b,c,h,w]; bwd: dx[b,c,h,w] = sum of the 4 dy
template <bool BWD>
__global__ void upsample2_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long BC, int H, int W) {
  const long total = BWD ? BC * H * W : BC * 4L * H * W;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    if (!BWD) {
      const int ox = (int)(i % (2 * W));
      const long t = i / (2 * W);
      const int oy = (int)(t % (2 * H));
      const long bc = t / (2 * H);
      dst[i] = src[(bc * H + oy / 2) * W + ox / 2];
    } else {
      const int xw = (int)(i % W);
      const long t = i / W;
      const int yh = (int)(t % H);
      const long bc = t / H;
      const bf16_t* p = src + (bc * 2 * H + 2 * yh) * 2 * W + 2 * xw;
      dst[i] = f32_to_bf16((bf16_to_f32(p[0]) + bf16_to_f32(p[1])) + (bf16_to_f32(p[2 * W]) + bf16_to_f32(p[2 * W + 1])));
    }
  }
}
// fwd: y[b,c,h,w] = mean of the 2x2 window (H, W = OUTPUT size); bwd: dx[.., 2h+dy, 2w+dx] = dy[.., h, w] / 4
template <bool BWD>
__global__ void avgpool2_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long BC, int H, int W) {
  const long total = BWD ? BC * 4L * H * W : BC * H * W;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    if (!BWD) {
      const int xw = (int)(i % W);
      const long t = i / W;
      const int yh = (int)(t % H);
      const long bc = t / H;
      const bf16_t* p = src + (bc * 2 * H + 2 * yh) * 2 * W + 2 * xw;
      dst[i] = f32_to_bf16(0.25f * ((bf16_to_f32(p[0]) + bf16_to_f32(p[1])) + (bf16_to_f32(p[2 * W]) + bf16_to_f32(p[2 * W + 1]))));
    } else {
      const int ox = (int)(i % (2 * W));
      const long t = i / (2 * W);
      const int oy = (int)(t % (2 * H));
      const long bc = t / (2 * H);
      dst[i] = f32_to_bf16(0.25f * bf16_to_f32(src[(bc * H + oy / 2) * W + ox / 2]));
    }
  }
}

// ---- nn.ReflectionPad2d (convs/basic.py:61-75: `padding="reflection[N]"` of the reference's Conv2d) ---------------------------
// fwd: y[bc][oy][ox] = x[bc][refl(oy - pt, H)][refl(ox - pl, W)], refl(i, n) = i < 0 ? -i : i >= n ? 2 (n - 1) - i : i.
// bwd: a GATHER (deterministic, no atomics): input pixel (iy, ix) receives the output positions that mirror onto it — along each
// axis the direct one plus at most one reflection at either border (pads are < the extent, checked on the host), <= 3 x 3 terms.
__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

template <bool IN_F32>
__global__ void reflect_pad2d_fwd_kernel(const void* __restrict__ x, bf16_t* __restrict__ y, long BC, int H, int W, int pl, int pr,
                                         int pt, int pb) {
  const int Ho = H + pt + pb, Wo = W + pl + pr;
  const long total = BC * (long)Ho * Wo;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const int ox = (int)(i % Wo);
    const long t = i / Wo;
    const int oy = (int)(t % Ho);
    const long bc = t / Ho;
    const long src = (bc * H + refl(oy - pt, H)) * W + refl(ox - pl, W);
    y[i] = IN_F32 ? f32_to_bf16(reinterpret_cast<const float*>(x)[src]) : reinterpret_cast<const bf16_t*>(x)[src];
  }
}

__global__ void reflect_pad2d_bwd_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long BC, int H, int W, int pl, int pr,
                                         int pt, int pb) {
  const int Ho = H + pt + pb, Wo = W + pl + pr;
  const long total = BC * (long)H * W;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const int ix = (int)(i % W);
    const long t = i / W;
    const int iy = (int)(t % H);
    const long bc = t / H;
    // output rows / columns that read this input row / column: the direct one, the mirror at the low border (output index
    // pt - iy for 1 <= iy <= pt), the mirror at the high border (pt + 2 (H - 1) - iy for 1 <= H - 1 - iy <= pb)
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = iy + pt;
    if (iy >= 1 && iy <= pt) ys[ny++] = pt - iy;
    if (H - 1 - iy >= 1 && H - 1 - iy <= pb) ys[ny++] = pt + 2 * (H - 1) - iy;
    xs[nx++] = ix + pl;
    if (ix >= 1 && ix <= pl) xs[nx++] = pl - ix;
    if (W - 1 - ix >= 1 && W - 1 - ix <= pr) xs[nx++] = pl + 2 * (W - 1) - ix;
    float acc = 0.f;
    const bf16_t* base = dy + bc * (long)Ho * Wo;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) acc += bf16_to_f32(base[(long)ys[a] * Wo + xs[b]]);
    dx[i] = f32_to_bf16(acc);
  }
}

// nearest x2 up-sampling on NHWC rows, 8 channels (16 bytes) per thread.  fwd: y[b][2h + i][2w + j][c] = x[b][h][w][c];
// bwd: dx[b][h][w][c] = the sum of the four dy (H, W = the SMALL size, C % 8 == 0)
template <bool BWD>
__global__ void upsample2_nhwc_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long B, int H, int W, int C) {
  const int c8 = C >> 3;
  const long total = (BWD ? B * H * W : B * 4L * H * W) * c8;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const int cc = (int)(i % c8);
    long t = i / c8;
    if (!BWD) {
      const int ox = (int)(t % (2 * W));
      t /= 2 * W;
      const int oy = (int)(t % (2 * H));
      const long b = t / (2 * H);
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + (((b * H + oy / 2) * W + ox / 2) * (long)C) + cc * 8);
      *reinterpret_cast<u32x4*>(dst + (((b * 2 * H + oy) * 2 * W + ox) * (long)C) + cc * 8) = v;
    } else {
      const int xw = (int)(t % W);
      t /= W;
      const int yh = (int)(t % H);
      const long b = t / H;
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[8];
        gn_load8<false>(src, (((b * 2 * H + 2 * yh + (q >> 1)) * 2 * W + 2 * xw + (q & 1)) * (long)C) + cc * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
      gn_store8(dst, (((b * H + yh) * W + xw) * (long)C) + cc * 8, acc);
    }
  }
}

// timestep embedding (multimodal/diffusion/unet.py:52-74): freq_i = exp(-ln(max_period) * i / half);
// out[b] = [cos(t_b * freq) | sin(t_b * freq) | 0 if dim is odd], computed in fp32
__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B, int dim,
                                          float max_period) {
  const int half = dim / 2;
  const long total = (long)B * dim;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += step) {
    const int b = (int)(i / dim), j = (int)(i - (long)b * dim);
    float v = 0.f;
    if (j < 2 * half) {
      const int k = j < half ? j : j - half;
      const float freq = expf(-logf(max_period) * (float)k / (float)half);
      const float arg = (float)t[b] * freq;
      v = j < half ? cosf(arg) : sinf(arg);
    }
    out[i] = v;
  }
}

int check_geom(const char* who, const ConvGeom& g) {
  CFHIP_REQUIRE(g.B > 0 && g.C > 0 && g.H > 0 && g.W > 0 && g.kh > 0 && g.kw > 0 && g.stride > 0 && g.dil > 0 &&
                    g.pad >= 0,
                "%s: bad geometry", who);
  CFHIP_REQUIRE(g.Ho > 0 && g.Wo > 0, "%s: empty output (%d x %d)", who, g.Ho, g.Wo);
  return CFHIP_OK;
}

ConvGeom make_geom(int B, int C, int H, int W, int kh, int kw, int stride, int pad, int dil, int Kp) {
  ConvGeom g;
  g.B = B; g.C = C; g.H = H; g.W = W; g.kh = kh; g.kw = kw; g.stride = stride; g.pad = pad; g.dil = dil;
  g.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
  g.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  g.K = C * kh * kw;
  g.Kp = Kp;
  return g;
}

}  // namespace

extern "C" int cfhip_conv_im2row(const void* x, int x_is_f32, void* rows, int B, int C, int H, int W, int kh,
                                 int kw, int stride, int pad, int dil, int Kp, void* stream) {
  CFHIP_REQUIRE(x && rows, "conv_im2row: null pointer");
  const ConvGeom g = make_geom(B, C, H, W, kh, kw, stride, pad, dil, Kp);
  int rc = check_geom("conv_im2row", g);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(Kp >= g.K, "conv_im2row: padded K %d < C*kh*kw = %d", Kp, g.K);
  const long total = (long)B * g.Ho * g.Wo * Kp;
  if (x_is_f32)
    hipLaunchKernelGGL((conv_im2row_kernel<true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)rows, g);
  else
    hipLaunchKernelGGL((conv_im2row_kernel<false>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)rows, g);
  CFHIP_CHECK_LAUNCH("conv_im2row");
  return CFHIP_OK;
}

extern "C" int cfhip_conv_row2im(const void* drows, void* dx, int B, int C, int H, int W, int kh, int kw,
                                 int stride, int pad, int dil, int Kp, void* stream) {
  CFHIP_REQUIRE(drows && dx, "conv_row2im: null pointer");
  const ConvGeom g = make_geom(B, C, H, W, kh, kw, stride, pad, dil, Kp);
  int rc = check_geom("conv_row2im", g);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(Kp >= g.K, "conv_row2im: padded K %d < C*kh*kw = %d", Kp, g.K);
  const long total = (long)B * C * H * W;
  hipLaunchKernelGGL(conv_row2im_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)drows, (bf16_t*)dx, g);
  CFHIP_CHECK_LAUNCH("conv_row2im");
  return CFHIP_OK;
}

extern "C" int cfhip_transpose_batched(const void* src, int src_is_f32, void* dst, int batch, int R, int C,
                                       void* stream) {
  CFHIP_REQUIRE(src && dst && batch > 0 && R > 0 && C > 0, "transpose_batched: bad arguments");
  CFHIP_REQUIRE(batch <= 65535, "transpose_batched: batch %d exceeds the grid limit", batch);
  const dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
  const long bs = (long)R * C;
  if (R % 8 == 0 && C % 8 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    if (src_is_f32)
      hipLaunchKernelGGL((transpose_batched_vec_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, bs, bs);
    else
      hipLaunchKernelGGL((transpose_batched_vec_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R, C, bs, bs);
    CFHIP_CHECK_LAUNCH("transpose_batched");
    return CFHIP_OK;
  }
  if (src_is_f32)
    hipLaunchKernelGGL((transpose_batched_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R,
                       C, bs, bs);
  else
    hipLaunchKernelGGL((transpose_batched_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, R,
                       C, bs, bs);
  CFHIP_CHECK_LAUNCH("transpose_batched");
  return CFHIP_OK;
}

extern "C" int cfhip_batchnorm_fwd(const void* x, int x_is_f32, const float* gamma, const float* beta, void* y,
                                   float* mean, float* rstd, float* running_mean, float* running_var, int B, int C,
                                   int inner, float eps, float momentum, int training, void* stream) {
  CFHIP_REQUIRE(x && y && mean && rstd && B > 0 && C > 0 && inner > 0, "batchnorm_fwd: bad arguments");
  CFHIP_REQUIRE(training || (running_mean && running_var), "batchnorm_fwd: eval mode needs the running statistics");
  if (x_is_f32)
    hipLaunchKernelGGL((bn_fwd_kernel<true>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y,
                       mean, rstd, running_mean, running_var, B, C, inner, eps, momentum, training);
  else
    hipLaunchKernelGGL((bn_fwd_kernel<false>), dim3(C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y,
                       mean, rstd, running_mean, running_var, B, C, inner, eps, momentum, training);
  CFHIP_CHECK_LAUNCH("batchnorm_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_batchnorm_bwd(const void* dy, const void* x, int x_is_f32, const float* gamma,
                                   const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                                   int B, int C, int inner, int accumulate, int training, void* stream) {
  CFHIP_REQUIRE(dy && x && mean && rstd && B > 0 && C > 0 && inner > 0, "batchnorm_bwd: bad arguments");
  if (x_is_f32)
    hipLaunchKernelGGL((bn_bwd_kernel<true>), dim3(C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, gamma,
                       mean, rstd, (bf16_t*)dx, dgamma, dbeta, B, C, inner, accumulate, training);
  else
    hipLaunchKernelGGL((bn_bwd_kernel<false>), dim3(C), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, gamma,
                       mean, rstd, (bf16_t*)dx, dgamma, dbeta, B, C, inner, accumulate, training);
  CFHIP_CHECK_LAUNCH("batchnorm_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_leaky_relu_fwd(const void* x, void* y, int64_t n, float slope, void* stream) {
  CFHIP_REQUIRE(x && y && n >= 0, "leaky_relu_fwd: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)y & 7) == 0, "leaky_relu_fwd: misaligned");
  hipLaunchKernelGGL((leaky_kernel<false>), dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)x, (bf16_t*)y, (long)n, slope);
  CFHIP_CHECK_LAUNCH("leaky_relu_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_leaky_relu_bwd(const void* dy, const void* x, void* dx, int64_t n, float slope, void* stream) {
  CFHIP_REQUIRE(dy && x && dx && n >= 0, "leaky_relu_bwd: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)x & 7) == 0 && ((uintptr_t)dy & 7) == 0 && ((uintptr_t)dx & 7) == 0,
                "leaky_relu_bwd: misaligned");
  hipLaunchKernelGGL((leaky_kernel<true>), dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, (long)n, slope);
  CFHIP_CHECK_LAUNCH("leaky_relu_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_avgpool_fwd(const void* x, void* y, int64_t BC, int inner, void* stream) {
  CFHIP_REQUIRE(x && y && BC > 0 && inner > 0, "avgpool_fwd: bad arguments");
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3((unsigned)((BC + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (long)BC, inner);
  CFHIP_CHECK_LAUNCH("avgpool_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_avgpool_bwd(const void* dy, void* dx, int64_t BC, int inner, void* stream) {
  CFHIP_REQUIRE(dy && dx && BC > 0 && inner > 0, "avgpool_bwd: bad arguments");
  const long total = (long)BC * inner;
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, total, inner);
  CFHIP_CHECK_LAUNCH("avgpool_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_softmax_focal(const float* logits, const int64_t* labels, float* loss_sum, float* dlogits,
                                   int B, int C, float gamma, float eps, float grad_scale, void* stream) {
  CFHIP_REQUIRE(logits && labels && loss_sum && B > 0 && C > 0, "softmax_focal: bad arguments");
  hipLaunchKernelGGL(softmax_focal_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     loss_sum, dlogits, B, C, gamma, eps, grad_scale);
  CFHIP_CHECK_LAUNCH("softmax_focal");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_affine_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                                          void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps,
                                          int silu, int affine_batch_stride, void* stream) {
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_fwd: affine batch stride must be 0 or C");
  const int affine_bs = affine_batch_stride;
  CFHIP_REQUIRE(x && gamma && beta && y && mean && rstd && B > 0 && C > 0 && G > 0 && inner > 0, "groupnorm_fwd: bad arguments");
  CFHIP_REQUIRE(C % G == 0, "groupnorm_fwd: %d channels do not split into %d groups", C, G);
  const bool vec = inner % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0;
  if (vec && x_is_f32)
    hipLaunchKernelGGL((gn_fwd_vec_kernel<true>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, add, gamma, beta,
                       (bf16_t*)y, mean, rstd, C, G, inner, eps, silu, affine_bs);
  else if (vec)
    hipLaunchKernelGGL((gn_fwd_vec_kernel<false>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, add, gamma, beta,
                       (bf16_t*)y, mean, rstd, C, G, inner, eps, silu, affine_bs);
  else if (x_is_f32)
    hipLaunchKernelGGL((gn_fwd_kernel<true>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, add, gamma, beta,
                       (bf16_t*)y, mean, rstd, C, G, inner, eps, silu, affine_bs);
  else
    hipLaunchKernelGGL((gn_fwd_kernel<false>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, x, add, gamma, beta,
                       (bf16_t*)y, mean, rstd, C, G, inner, eps, silu, affine_bs);
  CFHIP_CHECK_LAUNCH("groupnorm_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_affine_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                                          const float* beta, const float* mean, const float* rstd, void* dx,
                                          float* dgamma_part, float* dbeta_part, float* dadd, int B, int C, int G, int inner,
                                          int silu, int affine_batch_stride, void* stream) {
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_bwd: affine batch stride must be 0 or C");
  const int affine_bs = affine_batch_stride;
  CFHIP_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma_part && dbeta_part, "groupnorm_bwd: null pointer");
  CFHIP_REQUIRE(B > 0 && C > 0 && G > 0 && inner > 0 && C % G == 0, "groupnorm_bwd: bad geometry");
  CFHIP_REQUIRE((add == nullptr) == (dadd == nullptr) || dadd == nullptr, "groupnorm_bwd: dadd without add");
  const bool vec = inner % 8 == 0 && C / G <= GN_MAX_CG && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 &&
                   ((uintptr_t)dx & 15) == 0;
  if (vec && x_is_f32)
    hipLaunchKernelGGL((gn_bwd_vec_kernel<true>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, add,
                       gamma, beta, mean, rstd, (bf16_t*)dx, dgamma_part, dbeta_part, dadd, C, G, inner, silu, affine_bs);
  else if (vec)
    hipLaunchKernelGGL((gn_bwd_vec_kernel<false>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, add,
                       gamma, beta, mean, rstd, (bf16_t*)dx, dgamma_part, dbeta_part, dadd, C, G, inner, silu, affine_bs);
  else if (x_is_f32)
    hipLaunchKernelGGL((gn_bwd_kernel<true>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, add,
                       gamma, beta, mean, rstd, (bf16_t*)dx, dgamma_part, dbeta_part, dadd, C, G, inner, silu, affine_bs);
  else
    hipLaunchKernelGGL((gn_bwd_kernel<false>), dim3(B * G), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x,
                       add, gamma, beta, mean, rstd, (bf16_t*)dx, dgamma_part, dbeta_part, dadd, C, G, inner, silu, affine_bs);
  CFHIP_CHECK_LAUNCH("groupnorm_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_split_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                                         void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps, int silu,
                                         int affine_batch_stride, int splits, float* workspace, void* stream) {
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_split_fwd: affine batch stride must be 0 or C");
  CFHIP_REQUIRE(x && gamma && beta && y && mean && rstd && workspace && B > 0 && C > 0 && G > 0 && inner > 0,
                "groupnorm_split_fwd: bad arguments");
  CFHIP_REQUIRE(C % G == 0, "groupnorm_split_fwd: %d channels do not split into %d groups", C, G);
  CFHIP_REQUIRE(inner % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                "groupnorm_split_fwd: inner must be a multiple of 8 and x / y 16-byte aligned (use cfhip_groupnorm_affine_fwd)");
  CFHIP_REQUIRE(splits >= 1 && splits <= 64 && splits <= inner / 8, "groupnorm_split_fwd: %d slices for inner = %d", splits, inner);
  const dim3 grid(B * G, splits);
  hipStream_t s = (hipStream_t)stream;
  if (x_is_f32) {
    hipLaunchKernelGGL((gn_split_stats_kernel<true>), grid, dim3(256), 0, s, x, add, workspace, C, G, inner, splits);
    hipLaunchKernelGGL((gn_split_apply_kernel<true>), grid, dim3(256), 0, s, x, add, gamma, beta, workspace, (bf16_t*)y, mean, rstd,
                       C, G, inner, eps, silu, affine_batch_stride, splits);
  } else {
    hipLaunchKernelGGL((gn_split_stats_kernel<false>), grid, dim3(256), 0, s, x, add, workspace, C, G, inner, splits);
    hipLaunchKernelGGL((gn_split_apply_kernel<false>), grid, dim3(256), 0, s, x, add, gamma, beta, workspace, (bf16_t*)y, mean, rstd,
                       C, G, inner, eps, silu, affine_batch_stride, splits);
  }
  CFHIP_CHECK_LAUNCH("groupnorm_split_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_split_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                                         const float* beta, const float* mean, const float* rstd, void* dx, float* dgamma_part,
                                         float* dbeta_part, float* dadd, int B, int C, int G, int inner, int silu,
                                         int affine_batch_stride, int splits, float* workspace, void* stream) {
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_split_bwd: affine batch stride must be 0 or C");
  CFHIP_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma_part && dbeta_part && workspace, "groupnorm_split_bwd: null pointer");
  CFHIP_REQUIRE(B > 0 && C > 0 && G > 0 && inner > 0 && C % G == 0, "groupnorm_split_bwd: bad geometry");
  CFHIP_REQUIRE(dadd == nullptr || add != nullptr, "groupnorm_split_bwd: dadd without add");
  CFHIP_REQUIRE(inner % 8 == 0 && C / G <= GN_MAX_CG && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dx & 15) == 0,
                "groupnorm_split_bwd: inner %% 8, <= %d channels per group and 16-byte alignment required (use cfhip_groupnorm_affine_bwd)", GN_MAX_CG);
  CFHIP_REQUIRE(splits >= 1 && splits <= 64 && splits <= inner / 8, "groupnorm_split_bwd: %d slices for inner = %d", splits, inner);
  const dim3 grid(B * G, splits);
  hipStream_t s = (hipStream_t)stream;
  const int want = dadd != nullptr;
  if (x_is_f32) {
    hipLaunchKernelGGL((gn_split_bwd_stats_kernel<true>), grid, dim3(256), 0, s, (const bf16_t*)dy, x, add, gamma, beta, mean, rstd,
                       workspace, C, G, inner, silu, affine_batch_stride, splits);
    hipLaunchKernelGGL((gn_split_bwd_apply_kernel<true>), grid, dim3(256), 0, s, (const bf16_t*)dy, x, add, gamma, beta, mean, rstd,
                       workspace, (bf16_t*)dx, dgamma_part, dbeta_part, want, C, G, inner, silu, affine_batch_stride, splits);
  } else {
    hipLaunchKernelGGL((gn_split_bwd_stats_kernel<false>), grid, dim3(256), 0, s, (const bf16_t*)dy, x, add, gamma, beta, mean, rstd,
                       workspace, C, G, inner, silu, affine_batch_stride, splits);
    hipLaunchKernelGGL((gn_split_bwd_apply_kernel<false>), grid, dim3(256), 0, s, (const bf16_t*)dy, x, add, gamma, beta, mean, rstd,
                       workspace, (bf16_t*)dx, dgamma_part, dbeta_part, want, C, G, inner, silu, affine_batch_stride, splits);
  }
  if (want) hipLaunchKernelGGL(gn_split_dadd_kernel, dim3(B * G), dim3(128), 0, s, workspace, dadd, C, G, splits);
  CFHIP_CHECK_LAUNCH("groupnorm_split_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_fwd(const void* x, int x_is_f32, const float* add, const float* gamma, const float* beta,
                                   void* y, float* mean, float* rstd, int B, int C, int G, int inner, float eps,
                                   int silu, void* stream) {
  return cfhip_groupnorm_affine_fwd(x, x_is_f32, add, gamma, beta, y, mean, rstd, B, C, G, inner, eps, silu, 0, stream);
}

extern "C" int cfhip_groupnorm_bwd(const void* dy, const void* x, int x_is_f32, const float* add, const float* gamma,
                                   const float* beta, const float* mean, const float* rstd, void* dx,
                                   float* dgamma_part, float* dbeta_part, float* dadd, int B, int C, int G, int inner,
                                   int silu, void* stream) {
  return cfhip_groupnorm_affine_bwd(dy, x, x_is_f32, add, gamma, beta, mean, rstd, dx, dgamma_part, dbeta_part, dadd, B, C,
                                    G, inner, silu, 0, stream);
}

// ---- GroupNorm on NHWC rows: entry points ---------------------------------------------------------------------------------------
static int gn_nhwc_check(const char* who, int B, int C, int G, int inner, int splits) {
  CFHIP_REQUIRE(B > 0 && C > 0 && G > 0 && inner > 0, "%s: empty problem (B=%d C=%d G=%d inner=%d)", who, B, C, G, inner);
  CFHIP_REQUIRE(C % 8 == 0 && C % G == 0 && G <= 64 && C <= 4096, "%s: C = %d must be a multiple of 8 and of G = %d (<= 64), at most 4096", who, C, G);
  if (splits == 0) {
    const int cpg = C / G;
    CFHIP_REQUIRE(cpg % 2 == 0 && cpg <= 256, "%s: the group form (splits = 0) needs an even number of channels per group, at most 256 (got %d)", who, cpg);
    return CFHIP_OK;
  }
  CFHIP_REQUIRE(splits >= 1 && splits <= inner && splits <= 4096, "%s: splits = %d outside 1 .. min(inner, 4096)", who, splits);
  return CFHIP_OK;
}
static inline int gn_nhwc_rpp(int C) { return (C >> 3) >= 256 ? 1 : 256 / (C >> 3); }
// dwords a thread of the group form reads with one access: the largest of 4 / 2 / 1 that divides the chunk's cpg / 2 dwords
// (the chunk then starts on a 4 VW-byte boundary in every row: C * 2 is a multiple of 16)
static inline int gn_group_vw(int cpg) { const int d = cpg >> 1; return d % 4 == 0 ? 4 : d % 2 == 0 ? 2 : 1; }

extern "C" size_t cfhip_groupnorm_nhwc_workspace(int B, int C, int G, int splits, int backward, int with_add) {
  if (B <= 0 || C <= 0 || G <= 0 || splits <= 0) return 0;
  if (!backward) return (size_t)B * splits * G * 2 * sizeof(float);
  return ((size_t)B * splits * 2 * C + (with_add ? (size_t)B * splits * C : 0)) * sizeof(float);
}

extern "C" int cfhip_groupnorm_nhwc_fwd(const void* x, const float* add, const float* gamma, const float* beta, void* y, float* mean,
                                        float* rstd, int B, int C, int G, int inner, float eps, int silu, int affine_batch_stride,
                                        int splits, float* workspace, void* stream) {
  CFHIP_REQUIRE(x && gamma && beta && y && mean && rstd && (workspace || splits == 0), "groupnorm_nhwc_fwd: null argument");
  const int rc = gn_nhwc_check("groupnorm_nhwc_fwd", B, C, G, inner, splits);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_nhwc_fwd: affine_batch_stride must be 0 or C");
  hipStream_t s = (hipStream_t)stream;
  const bf16_t* xp = (const bf16_t*)x;
  if (splits == 0) {  // group form: one workgroup per (sample, group), one launch
    // dwords per thread (rows x vector width): kept in registers when they fit
    const int vw = (((uintptr_t)x | (uintptr_t)y) & 15) == 0 ? gn_group_vw(C / G) : 1;  // (rows of an odd view: dword accesses)
    const int rpp = 256 / ((C / G / 2) / vw);
    const int rpt = ((inner + rpp - 1) / rpp) * vw;
#define CFHIP_GN_GROUP_FWD_V(RPT, VW)                                                                                                  \
  hipLaunchKernelGGL((gn_nhwc_group_fwd_kernel<RPT, VW>), dim3(B * G), dim3(256), 0, s, xp, add, gamma, beta, (bf16_t*)y, mean, rstd, C, G, \
                     inner, eps, silu, affine_batch_stride)
#define CFHIP_GN_GROUP_FWD(RPT)                  \
  do {                                           \
    if (vw == 4) CFHIP_GN_GROUP_FWD_V(RPT, 4);   \
    else if (vw == 2) CFHIP_GN_GROUP_FWD_V(RPT, 2); \
    else CFHIP_GN_GROUP_FWD_V(RPT, 1);           \
  } while (0)
    if (rpt <= 48) CFHIP_GN_GROUP_FWD(48);  // (a 24-row instantiation came out of hipcc with 254 VGPRs and 132 B of scratch: not built)
    else if (rpt <= 96) CFHIP_GN_GROUP_FWD(96);
    else if (rpt <= 176) CFHIP_GN_GROUP_FWD(176);
    else CFHIP_GN_GROUP_FWD(0);
#undef CFHIP_GN_GROUP_FWD
#undef CFHIP_GN_GROUP_FWD_V
    CFHIP_CHECK_LAUNCH("groupnorm_nhwc_fwd");
    return CFHIP_OK;
  }
  const size_t lds = ((size_t)gn_nhwc_rpp(C) * C + C + 2 * 64) * sizeof(float);
  const int cpg = C / G;
  if ((C >> 3) > 256) {
    hipLaunchKernelGGL((gn_nhwc_stats_kernel<2>), dim3(B * splits), dim3(256), lds, s, xp, add, workspace, C, G, inner, splits);
    hipLaunchKernelGGL(gn_nhwc_merge_kernel, dim3(B), dim3(64), 0, s, workspace, mean, rstd, G, cpg, inner, eps, splits);
    hipLaunchKernelGGL((gn_nhwc_apply_kernel<2>), dim3(B * splits), dim3(256), 0, s, xp, add, gamma, beta, (bf16_t*)y, mean, rstd, C, G, inner,
                       silu, affine_batch_stride, splits);
  } else {
    hipLaunchKernelGGL((gn_nhwc_stats_kernel<1>), dim3(B * splits), dim3(256), lds, s, xp, add, workspace, C, G, inner, splits);
    hipLaunchKernelGGL(gn_nhwc_merge_kernel, dim3(B), dim3(64), 0, s, workspace, mean, rstd, G, cpg, inner, eps, splits);
    hipLaunchKernelGGL((gn_nhwc_apply_kernel<1>), dim3(B * splits), dim3(256), 0, s, xp, add, gamma, beta, (bf16_t*)y, mean, rstd, C, G, inner,
                       silu, affine_batch_stride, splits);
  }
  CFHIP_CHECK_LAUNCH("groupnorm_nhwc_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_groupnorm_nhwc_bwd(const void* dy, const void* x, const float* add, const float* gamma, const float* beta,
                                        const float* mean, const float* rstd, void* dx, float* dgamma_part, float* dbeta_part, float* dadd,
                                        int B, int C, int G, int inner, int silu, int affine_batch_stride, int splits, float* workspace,
                                        void* stream) {
  CFHIP_REQUIRE(dy && x && gamma && beta && mean && rstd && dx && dgamma_part && dbeta_part && (workspace || splits == 0), "groupnorm_nhwc_bwd: null argument");
  const int rc = gn_nhwc_check("groupnorm_nhwc_bwd", B, C, G, inner, splits);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(affine_batch_stride == 0 || affine_batch_stride == C, "groupnorm_nhwc_bwd: affine_batch_stride must be 0 or C");
  CFHIP_REQUIRE((dadd != nullptr) == (add != nullptr), "groupnorm_nhwc_bwd: dadd goes with add");
  hipStream_t s = (hipStream_t)stream;
  if (splits == 0) {  // group form
    const int vw = (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0 ? gn_group_vw(C / G) : 1;
    const int rpp = 256 / ((C / G / 2) / vw);
    const int rpt = ((inner + rpp - 1) / rpp) * vw;
#define CFHIP_GN_GROUP_BWD_V(RPT, VW)                                                                                                   \
  hipLaunchKernelGGL((gn_nhwc_group_bwd_kernel<RPT, VW>), dim3(B * G), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, add, gamma, beta, \
                     mean, rstd, (bf16_t*)dx, dgamma_part, dbeta_part, dadd, C, G, inner, silu, affine_batch_stride)
#define CFHIP_GN_GROUP_BWD(RPT)                  \
  do {                                           \
    if (vw == 4) CFHIP_GN_GROUP_BWD_V(RPT, 4);   \
    else if (vw == 2) CFHIP_GN_GROUP_BWD_V(RPT, 2); \
    else CFHIP_GN_GROUP_BWD_V(RPT, 1);           \
  } while (0)
    if (rpt <= 48) CFHIP_GN_GROUP_BWD(48);
    else if (rpt <= 88) CFHIP_GN_GROUP_BWD(88);
    else CFHIP_GN_GROUP_BWD(0);
#undef CFHIP_GN_GROUP_BWD
#undef CFHIP_GN_GROUP_BWD_V
    CFHIP_CHECK_LAUNCH("groupnorm_nhwc_bwd");
    return CFHIP_OK;
  }
  const size_t lds_a = ((size_t)gn_nhwc_rpp(C) * C + C) * sizeof(float);
  const size_t lds_b = ((size_t)gn_nhwc_rpp(C) * C + 2 * C + 2 * 64) * sizeof(float);
  float* part = workspace;
  float* dpart = dadd != nullptr ? workspace + (size_t)B * splits * 2 * C : nullptr;
  const bf16_t* dyp = (const bf16_t*)dy;
  const bf16_t* xp = (const bf16_t*)x;
  if ((C >> 3) > 256) {
    hipLaunchKernelGGL((gn_nhwc_bwd_stats_kernel<2>), dim3(B * splits), dim3(256), lds_a, s, dyp, xp, add, gamma, beta, mean, rstd, part, C, G,
                       inner, silu, affine_batch_stride, splits);
    hipLaunchKernelGGL(gn_nhwc_bwd_merge_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, part, dgamma_part, dbeta_part, C, splits);
    hipLaunchKernelGGL((gn_nhwc_bwd_apply_kernel<2>), dim3(B * splits), dim3(256), lds_b, s, dyp, xp, add, gamma, beta, mean, rstd, part,
                       (bf16_t*)dx, dgamma_part, dbeta_part, dpart, C, G, inner, silu, affine_batch_stride, splits);
  } else {
    hipLaunchKernelGGL((gn_nhwc_bwd_stats_kernel<1>), dim3(B * splits), dim3(256), lds_a, s, dyp, xp, add, gamma, beta, mean, rstd, part, C, G,
                       inner, silu, affine_batch_stride, splits);
    hipLaunchKernelGGL(gn_nhwc_bwd_merge_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, part, dgamma_part, dbeta_part, C, splits);
    hipLaunchKernelGGL((gn_nhwc_bwd_apply_kernel<1>), dim3(B * splits), dim3(256), lds_b, s, dyp, xp, add, gamma, beta, mean, rstd, part,
                       (bf16_t*)dx, dgamma_part, dbeta_part, dpart, C, G, inner, silu, affine_batch_stride, splits);
  }
  if (dadd != nullptr)
    hipLaunchKernelGGL(gn_nhwc_dadd_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, dpart, dadd, C, splits);
  CFHIP_CHECK_LAUNCH("groupnorm_nhwc_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_silu_f32_fwd(const float* x, float* y, int64_t n, void* stream) {
  CFHIP_REQUIRE(x && y && n > 0, "silu_f32_fwd: bad arguments");
  hipLaunchKernelGGL((silu_f32_kernel<false>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x, y, (long)n);
  CFHIP_CHECK_LAUNCH("silu_f32_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_silu_f32_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  CFHIP_REQUIRE(dy && x && dx && n > 0, "silu_f32_bwd: bad arguments");
  hipLaunchKernelGGL((silu_f32_kernel<true>), dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, (long)n);
  CFHIP_CHECK_LAUNCH("silu_f32_bwd");
  return CFHIP_OK;
}

// table: 4 int64 per problem.  Forward: {weight bf16 [N, K], bias f32 [N] or 0, out f32 [B, N], N}; backward: {weight, dY f32 [B, N] or 0
// (no gradient for that output), dY as bf16 [B, N] out or 0, N}.  A HOST array (the pointers travel in the kernel arguments).
static int time_proj_args(const int64_t* table, int count, bool fwd, TimeProjArgs* a, const char* who) {
  CFHIP_REQUIRE(table && count > 0 && count <= TP_MAX, "%s: 1 .. %d problems per call (got %d)", who, TP_MAX, count);
  int blocks = 0;
  for (int i = 0; i < count; ++i) {
    const int64_t* e = table + 4 * i;
    CFHIP_REQUIRE(e[0] != 0 && e[3] > 0 && (e[0] & 15) == 0, "%s: problem %d: weight pointer (16-byte aligned) and N > 0", who, i);
    a->w[i] = reinterpret_cast<const bf16_t*>(e[0]);
    a->vec[i] = reinterpret_cast<const float*>(e[1]);
    if (fwd) {
      CFHIP_REQUIRE(e[2] != 0, "%s: problem %d: output pointer", who, i);
      a->out[i] = reinterpret_cast<float*>(e[2]);
      a->aux[i] = nullptr;
    } else {
      a->out[i] = nullptr;
      a->aux[i] = reinterpret_cast<bf16_t*>(e[2]);
    }
    a->n[i] = (int)e[3];
    a->blk0[i] = blocks;
    blocks += ((int)e[3] + 63) / 64;
  }
  a->blk0[count] = blocks;
  a->count = count;
  return CFHIP_OK;
}

extern "C" int cfhip_time_proj_fwd(const float* emb, int B, int K, const int64_t* table, int count, void* t_bf16, void* stream) {
  CFHIP_REQUIRE(emb && t_bf16 && B > 0 && K > 0 && K % 32 == 0 && K <= 2048, "time_proj_fwd: emb [B, K] f32 with K %% 32 == 0, K <= 2048 — 8 bf16 rows of it sit in 32 KB of LDS (got B=%d K=%d)", B, K);
  TimeProjArgs a;
  const int rc = time_proj_args(table, count, true, &a, "time_proj_fwd");
  if (rc != CFHIP_OK) return rc;
  const size_t lds = (size_t)8 * K * 2 + (size_t)4 * 8 * 64 * 4;
  hipLaunchKernelGGL(time_proj_fwd_kernel, dim3(a.blk0[count], (B + 7) / 8), dim3(256), lds, (hipStream_t)stream, emb, B, K, a,
                     (bf16_t*)t_bf16);
  CFHIP_CHECK_LAUNCH("time_proj_fwd");
  return CFHIP_OK;
}

// partial: f32 workspace of (sum over problems of ceil(N_i / 64)) * ceil8(B) * K elements, owned by the caller
extern "C" int cfhip_time_proj_bwd(const float* emb, int B, int K, const int64_t* table, int count, float* partial, float* d_emb,
                                   void* stream) {
  CFHIP_REQUIRE(emb && partial && d_emb && B > 0 && K > 0 && K % 32 == 0 && K <= 2048, "time_proj_bwd: bad arguments (B=%d K=%d)", B, K);
  TimeProjArgs a;
  const int rc = time_proj_args(table, count, false, &a, "time_proj_bwd");
  if (rc != CFHIP_OK) return rc;
  const int Bpad = (B + 7) / 8 * 8, nblk = a.blk0[count];
  hipLaunchKernelGGL(time_proj_bwd_part_kernel, dim3(nblk, (K + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, K, a, partial, Bpad);
  CFHIP_CHECK_LAUNCH("time_proj_bwd(partial sums)");
  hipLaunchKernelGGL(time_proj_bwd_sum_kernel, dim3((unsigned)(((long)B * K + 255) / 256)), dim3(256), 0, (hipStream_t)stream, emb,
                     (const float*)partial, d_emb, B, K, Bpad, nblk);
  CFHIP_CHECK_LAUNCH("time_proj_bwd(sum)");
  return CFHIP_OK;
}

extern "C" int cfhip_upsample2_fwd(const void* x, void* y, int64_t BC, int H, int W, void* stream) {
  CFHIP_REQUIRE(x && y && BC > 0 && H > 0 && W > 0, "upsample2_fwd: bad arguments");
  hipLaunchKernelGGL((upsample2_kernel<false>), dim3(grid_for(BC * 4L * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (long)BC, H, W);
  CFHIP_CHECK_LAUNCH("upsample2_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_upsample2_bwd(const void* dy, void* dx, int64_t BC, int H, int W, void* stream) {
  CFHIP_REQUIRE(dy && dx && BC > 0 && H > 0 && W > 0, "upsample2_bwd: bad arguments");
  hipLaunchKernelGGL((upsample2_kernel<true>), dim3(grid_for(BC * (long)H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, (long)BC, H, W);
  CFHIP_CHECK_LAUNCH("upsample2_bwd");
  return CFHIP_OK;
}
extern "C" int cfhip_upsample2_nhwc_fwd(const void* x, void* y, int64_t B, int H, int W, int C, void* stream) {
  CFHIP_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "upsample2_nhwc_fwd: bad arguments (C must be a multiple of 8)");
  hipLaunchKernelGGL((upsample2_nhwc_kernel<false>), dim3(grid_for(B * 4L * H * W * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (long)B, H, W, C);
  CFHIP_CHECK_LAUNCH("upsample2_nhwc_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_upsample2_nhwc_bwd(const void* dy, void* dx, int64_t B, int H, int W, int C, void* stream) {
  CFHIP_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "upsample2_nhwc_bwd: bad arguments (C must be a multiple of 8)");
  hipLaunchKernelGGL((upsample2_nhwc_kernel<true>), dim3(grid_for(B * (long)H * W * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, (long)B, H, W, C);
  CFHIP_CHECK_LAUNCH("upsample2_nhwc_bwd");
  return CFHIP_OK;
}
extern "C" int cfhip_avgpool2_fwd(const void* x, void* y, int64_t BC, int Ho, int Wo, void* stream) {
  CFHIP_REQUIRE(x && y && BC > 0 && Ho > 0 && Wo > 0, "avgpool2_fwd: bad arguments");
  hipLaunchKernelGGL((avgpool2_kernel<false>), dim3(grid_for(BC * (long)Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, (long)BC, Ho, Wo);
  CFHIP_CHECK_LAUNCH("avgpool2_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_avgpool2_bwd(const void* dy, void* dx, int64_t BC, int Ho, int Wo, void* stream) {
  CFHIP_REQUIRE(dy && dx && BC > 0 && Ho > 0 && Wo > 0, "avgpool2_bwd: bad arguments");
  hipLaunchKernelGGL((avgpool2_kernel<true>), dim3(grid_for(BC * 4L * Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, (long)BC, Ho, Wo);
  CFHIP_CHECK_LAUNCH("avgpool2_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_reflect_pad2d_fwd(const void* x, int x_is_f32, void* y, int64_t BC, int H, int W, int pl, int pr, int pt, int pb,
                                       void* stream) {
  CFHIP_REQUIRE(x && y && BC > 0 && H > 0 && W > 0, "reflect_pad2d_fwd: bad arguments");
  CFHIP_REQUIRE(pl >= 0 && pr >= 0 && pt >= 0 && pb >= 0 && pl < W && pr < W && pt < H && pb < H,
                "reflect_pad2d_fwd: every pad must be >= 0 and smaller than the extent it mirrors (H=%d W=%d pads %d %d %d %d)", H, W, pl, pr, pt, pb);
  const long total = (long)BC * (H + pt + pb) * (W + pl + pr);
  if (x_is_f32)
    hipLaunchKernelGGL((reflect_pad2d_fwd_kernel<true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y,
                       (long)BC, H, W, pl, pr, pt, pb);
  else
    hipLaunchKernelGGL((reflect_pad2d_fwd_kernel<false>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y,
                       (long)BC, H, W, pl, pr, pt, pb);
  CFHIP_CHECK_LAUNCH("reflect_pad2d_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_reflect_pad2d_bwd(const void* dy, void* dx, int64_t BC, int H, int W, int pl, int pr, int pt, int pb, void* stream) {
  CFHIP_REQUIRE(dy && dx && BC > 0 && H > 0 && W > 0, "reflect_pad2d_bwd: bad arguments");
  CFHIP_REQUIRE(pl >= 0 && pr >= 0 && pt >= 0 && pb >= 0 && pl < W && pr < W && pt < H && pb < H,
                "reflect_pad2d_bwd: every pad must be >= 0 and smaller than the extent it mirrors (H=%d W=%d pads %d %d %d %d)", H, W, pl, pr, pt, pb);
  hipLaunchKernelGGL(reflect_pad2d_bwd_kernel, dim3(grid_for((long)BC * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dy, (bf16_t*)dx, (long)BC, H, W, pl, pr, pt, pb);
  CFHIP_CHECK_LAUNCH("reflect_pad2d_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_timestep_embedding(const int64_t* t, float* out, int B, int dim, float max_period, void* stream) {
  CFHIP_REQUIRE(t && out && B > 0 && dim > 1 && max_period > 1.f, "timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long)B * dim, 256)), dim3(256), 0, (hipStream_t)stream, t,
                     out, B, dim, max_period);
  CFHIP_CHECK_LAUNCH("timestep_embedding");
  return CFHIP_OK;
}

extern "C" int cfhip_conv3x3_pack_filters(const void* w, void* out, int Cout, int Cin, int rotate, void* stream) {
  CFHIP_REQUIRE(w && out && Cout > 0 && Cin > 0 && Cin % 8 == 0, "conv3x3_pack_filters: bad arguments (Cin %% 8 == 0)");
  CFHIP_REQUIRE(((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0, "conv3x3_pack_filters: 16-byte alignment");
  const long total = (long)Cout * (Cin / 8);
  if (rotate)
    hipLaunchKernelGGL((conv3x3_pack_kernel<true>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w, (bf16_t*)out, Cout, Cin);
  else
    hipLaunchKernelGGL((conv3x3_pack_kernel<false>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)w, (bf16_t*)out, Cout, Cin);
  CFHIP_CHECK_LAUNCH("conv3x3_pack_filters");
  return CFHIP_OK;
}

// table: a HOST array of 5 int64 per problem {w bf16 [Cout, Cin, 3, 3], out, Cout, Cin, rotate}; 1 .. 64 problems per call
extern "C" int cfhip_conv3x3_pack_filters_grouped(const int64_t* table, int count, void* stream) {
  CFHIP_REQUIRE(table && count > 0 && count <= PK_MAX, "conv3x3_pack_filters_grouped: 1 .. %d problems per call (got %d)", PK_MAX, count);
  PackArgs a;
  long blocks = 0;
  for (int i = 0; i < count; ++i) {
    const int64_t* e = table + 5 * i;
    CFHIP_REQUIRE(e[0] != 0 && e[1] != 0 && e[2] > 0 && e[3] > 0 && e[3] % 8 == 0 && (e[0] & 15) == 0 && (e[1] & 15) == 0,
                  "conv3x3_pack_filters_grouped: problem %d: 16-byte aligned pointers, Cout > 0, Cin %% 8 == 0", i);
    a.w[i] = reinterpret_cast<const bf16_t*>(e[0]);
    a.out[i] = reinterpret_cast<bf16_t*>(e[1]);
    a.cout[i] = (int)e[2];
    a.cin[i] = (int)e[3];
    a.rot[i] = e[4] != 0;
    a.blk0[i] = (int)blocks;
    blocks += (e[2] * (e[3] / 8) + 255) / 256;
  }
  CFHIP_REQUIRE(blocks < (1L << 30), "conv3x3_pack_filters_grouped: too many items");
  a.blk0[count] = (int)blocks;
  a.count = count;
  hipLaunchKernelGGL(conv3x3_pack_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  CFHIP_CHECK_LAUNCH("conv3x3_pack_filters_grouped");
  return CFHIP_OK;
}
