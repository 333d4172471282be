// HBM-bound helpers of the ViT training path for gfx950: casts, stand-alone GELU, residual add,
// transpose, column sums (bias gradients), ViT patch im2row, token assembly (head token +
// positional encoding), fused Adam(W) over the flat parameter arena, gradient sum-of-squares and
// softmax cross-entropy.  All kernels use 16-byte (or 8-byte bf16x4) coalesced accesses and
// grid-stride loops capped at ~8 workgroups per CU.
#include "common.h"

int cfhip_internal_colreduce_f32(const float* partials, int R, int D, float* out, int accumulate,
                                 hipStream_t s);

namespace {

inline int grid_for(long work_items, int per_block, int cap = 2048) {
  long b = (work_items + per_block - 1) / per_block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- column reduce of a dense f32 [R][D] matrix: out[d] (+)= sum_r p[r][d] ---------------------
// workgroup = 32 columns x 32 row-lanes (1024 threads): every thread owns R/32 rows of one column and
// keeps its loads independent, so the kernel is one short burst of parallel loads instead of a long
// dependent chain (the 8-row-lane version spent 13 us on a 3 MB input).
// (blockIdx.y = 1: the second matrix / output of a pair — GroupNorm's dgamma and dbeta partial sums in one launch)
__global__ __launch_bounds__(1024) void colreduce_f32_kernel(const float* __restrict__ p, int R, int D,
                                                             float* __restrict__ out, int accumulate,
                                                             const float* __restrict__ p2 = nullptr, float* __restrict__ out2 = nullptr) {
  __shared__ float red[32][33];
  if (blockIdx.y == 1) { p = p2; out = out2; }
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cx;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < D) {
    int r = ry;
    for (; r + 96 < R; r += 128) {
      a0 += p[(long)r * D + col];
      a1 += p[(long)(r + 32) * D + col];
      a2 += p[(long)(r + 64) * D + col];
      a3 += p[(long)(r + 96) * D + col];
    }
    for (; r < R; r += 32) a0 += p[(long)r * D + col];
  }
  red[ry][cx] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (ry == 0 && col < D) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) s += red[k][cx];
    out[col] = accumulate ? out[col] + s : s;
  }
}

// ---- column sums of a bf16 [M][N] matrix, stage 1: partial[y][n] = sum over the rows of slice y
// wave = one 512-column strip (64 lanes x 8 columns), waves of a workgroup take different rows.
constexpr int CS_WAVES = 4;
__global__ __launch_bounds__(CS_WAVES * 64) void colsum_partial_kernel(const bf16_t* __restrict__ x,
                                                                        float* __restrict__ partial,
                                                                        int M, int N, long ldx,
                                                                        int rows_per_slice) {
  __shared__ float red[CS_WAVES][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_slice;
  const int r1 = min(M, r0 + rows_per_slice);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < N) {  // N % 8 == 0
    for (int r = r0 + wave; r < r1; r += CS_WAVES) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(x + (long)r * ldx + col);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[2 * e] += bf16lo(w[e]);
        acc[2 * e + 1] += bf16hi(w[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[wave][lane * 8 + e] = acc[e];
  __syncthreads();
  if (wave == 0 && col < N) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < CS_WAVES; ++k) s += red[k][lane * 8 + e];
      partial[(long)blockIdx.y * N + col + e] = s;
    }
  }
}
// generic (any N, any alignment) variant: thread per column
__global__ void colsum_generic_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int M, int N,
                                      long ldx, int accumulate) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= N) return;
  float s = 0.f;
  for (int r = 0; r < M; ++r) s += bf16_to_f32(x[(long)r * ldx + col]);
  out[col] = accumulate ? out[col] + s : s;
}

inline int colsum_slices(int M) {
  int slices = (M + 31) / 32;  // >= 32 rows per slice
  if (slices > 256) slices = 256;
  if (slices < 1) slices = 1;
  return slices;
}

// ---- casts / element-wise ------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    reinterpret_cast<u32x2*>(dst)[i] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f32_to_bf16(src[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const u32x2 w = reinterpret_cast<const u32x2*>(src)[i];
    reinterpret_cast<f32x4*>(dst)[i] = f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = bf16_to_f32(src[i]);
}
// f32 <-> two bf16 words (hi = bf16(v), lo = bf16(v - hi)): the ends of the two-word residual-gradient stream (norm.hip, LO)
__global__ void split_f32_bf16x2_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    const u32x2 h = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    reinterpret_cast<u32x2*>(hi)[i] = h;
    reinterpret_cast<u32x2*>(lo)[i] = u32x2{pack_bf16x2(v[0] - bf16lo(h[0]), v[1] - bf16hi(h[0])),
                                            pack_bf16x2(v[2] - bf16lo(h[1]), v[3] - bf16hi(h[1]))};
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const bf16_t h = f32_to_bf16(src[i]);
    hi[i] = h;
    lo[i] = f32_to_bf16(src[i] - bf16_to_f32(h));
  }
}
__global__ void join_bf16x2_f32_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, float* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const u32x2 h = reinterpret_cast<const u32x2*>(hi)[i], l = reinterpret_cast<const u32x2*>(lo)[i];
    reinterpret_cast<f32x4*>(dst)[i] = f32x4{bf16lo(h[0]) + bf16lo(l[0]), bf16hi(h[0]) + bf16hi(l[0]),
                                             bf16lo(h[1]) + bf16lo(l[1]), bf16hi(h[1]) + bf16hi(l[1])};
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = bf16_to_f32(hi[i]) + bf16_to_f32(lo[i]);
}
// MODE 0: y = gelu(x); 1: dx = dy * gelu'(x); 2: out = a + b; 3 / 4: MODE 0 / 1 with quick GELU
template <int MODE>
__global__ void ew_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                               bf16_t* __restrict__ out, long n) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const u32x2 wa = reinterpret_cast<const u32x2*>(a)[i];
    float va[4] = {bf16lo(wa[0]), bf16hi(wa[0]), bf16lo(wa[1]), bf16hi(wa[1])};
    float vb[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE != 0 && MODE != 3) {
      const u32x2 wb = reinterpret_cast<const u32x2*>(b)[i];
      vb[0] = bf16lo(wb[0]); vb[1] = bf16hi(wb[0]); vb[2] = bf16lo(wb[1]); vb[3] = bf16hi(wb[1]);
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      o[e] = MODE == 0 ? gelu_erf_f(va[e]) : MODE == 1 ? va[e] * gelu_erf_grad_f(vb[e]) : MODE == 2 ? va[e] + vb[e]
             : MODE == 3 ? quick_gelu_f(va[e]) : va[e] * quick_gelu_grad_f(vb[e]);
    reinterpret_cast<u32x2*>(out)[i] = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = bf16_to_f32(a[i]);
    const float y = (MODE != 0 && MODE != 3) ? bf16_to_f32(b[i]) : 0.f;
    out[i] = f32_to_bf16(MODE == 0 ? gelu_erf_f(x) : MODE == 1 ? x * gelu_erf_grad_f(y) : MODE == 2 ? x + y
                         : MODE == 3 ? quick_gelu_f(x) : x * quick_gelu_grad_f(y));
  }
}

// ---- bf16 transpose through a padded 64x64 LDS tile -----------------------------------------------
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                      int C, long lds_, long ldd) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 256 threads: 4 rows per pass
  for (int r = ty; r < 64; r += 4) {
    const int rr = r0 + r, cc = c0 + tx;
    tile[r][tx] = (rr < R && cc < C) ? src[(long)rr * lds_ + cc] : (bf16_t)0;
  }
  __syncthreads();
  for (int c = ty; c < 64; c += 4) {
    const int cc = c0 + c, rr = r0 + tx;
    if (cc < C && rr < R) dst[(long)cc * ldd + rr] = tile[tx][c];
  }
}

// ---- ViT patch im2row: img [B,C,H,W] -> rows [B*gh*gw][C*P*P] (c, ph, pw) -------------------------
// one thread = 4 consecutive pw (P % 4 == 0): 16-byte f32 read (or 8-byte bf16), 8-byte bf16 write.
template <bool IN_BF16>
__global__ void im2row_kernel(const void* __restrict__ img, bf16_t* __restrict__ rows, int B, int C,
                              int Hh, int Ww, int P) {
  const int gh = Hh / P, gw = Ww / P;
  const int kdim = C * P * P;
  const long total4 = (long)B * gh * gw * kdim / 4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += stride) {
    const long e = i * 4;
    const long rowi = e / kdim;
    const int kk = (int)(e - rowi * kdim);
    const int c = kk / (P * P);
    const int ph = (kk - c * P * P) / P;
    const int pw = kk - c * P * P - ph * P;
    const int b = (int)(rowi / (gh * gw));
    const int pi = (int)(rowi - (long)b * gh * gw);
    const int py = pi / gw, px = pi - py * gw;
    const long src = (((long)b * C + c) * Hh + (py * P + ph)) * Ww + (px * P + pw);
    u32x2 o;
    if (IN_BF16) {
      o = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(img) + src);
    } else {
      const f32x4 v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(img) + src);
      o = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    *reinterpret_cast<u32x2*>(rows + e) = o;
  }
}

// ---- token assembly --------------------------------------------------------------------------------
template <bool OUT_F32>
__global__ void assemble_fwd_kernel(const bf16_t* __restrict__ patches, const float* __restrict__ head,
                                    const float* __restrict__ pos, void* __restrict__ x0, int B, int Np,
                                    int D) {
  const int T = Np + 1;
  const int d4 = D >> 2;
  const long total = (long)B * T * d4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int dd = (int)(i % d4) * 4;
    const long bt = i / d4;
    const int t = (int)(bt % T);
    const int b = (int)(bt / T);
    const f32x4 pe = *reinterpret_cast<const f32x4*>(pos + (long)t * D + dd);
    f32x4 v;
    if (t == 0) {
      v = *reinterpret_cast<const f32x4*>(head + dd);
    } else {
      const u32x2 w = *reinterpret_cast<const u32x2*>(patches + ((long)b * Np + (t - 1)) * D + dd);
      v = f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
    }
    v += pe;
    if (OUT_F32)
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(x0) + bt * D + dd) = v;
    else
      *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(x0) + bt * D + dd) =
          u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
  }
}
// backward: dpatches = dx0[:, 1:], dpos[t] = sum_b dx0[b, t], dhead = dpos-row-0 sum.
// Round 6: one workgroup per (token, 256-column block), eight batch groups x 32 lanes x 8 columns (16-byte loads / stores), the eight
// partial sums folded through LDS in a fixed order.  (Rounds 1-5: one thread per (token, column) walking the whole batch with 2-byte
// loads — 124 us at the ViT-B/16 step's 128 x 197 x 768, 242 us at CLIP's 256 x 50 x 768, on the tail of backward where nothing overlaps it.)
// D % 8 != 0: the scalar form below.
__global__ __launch_bounds__(256) void assemble_bwd_vec_kernel(const bf16_t* __restrict__ dx0, bf16_t* __restrict__ dpatches,
                                                               float* __restrict__ dhead, float* __restrict__ dpos, int B, int Np,
                                                               int D, int accumulate) {
  __shared__ float red[8][256];
  const int T = Np + 1;
  const int nblk = (D + 255) / 256;
  const int t = blockIdx.x / nblk, cb = blockIdx.x - t * nblk;
  const int grp = threadIdx.x >> 5, ln = threadIdx.x & 31;
  const int d0 = cb * 256 + ln * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (d0 < D) {
    for (int b = grp; b < B; b += 8) {
      const u32x4 w = *reinterpret_cast<const u32x4*>(dx0 + ((long)b * T + t) * D + d0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[2 * e] += bf16lo(w[e]); s[2 * e + 1] += bf16hi(w[e]); }
      if (t > 0 && dpatches != nullptr) *reinterpret_cast<u32x4*>(dpatches + ((long)b * Np + (t - 1)) * D + d0) = w;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[grp][ln * 8 + e] = s[e];
  __syncthreads();
  const int d = cb * 256 + (int)threadIdx.x;
  if (d < D) {
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) tot += red[g][threadIdx.x];
    const long i = (long)t * D + d;
    if (dpos != nullptr) dpos[i] = accumulate ? dpos[i] + tot : tot;
    if (t == 0 && dhead != nullptr) dhead[d] = accumulate ? dhead[d] + tot : tot;
  }
}

// one thread per (t, d): loops over the batch (coalesced across d).
__global__ void assemble_bwd_kernel(const bf16_t* __restrict__ dx0, bf16_t* __restrict__ dpatches,
                                    float* __restrict__ dhead, float* __restrict__ dpos, int B, int Np,
                                    int D, int accumulate) {
  const int T = Np + 1;
  const long total = (long)T * D;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const int t = (int)(i / D), d = (int)(i - (long)t * D);
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const bf16_t w = dx0[((long)b * T + t) * D + d];
      s += bf16_to_f32(w);
      if (t > 0 && dpatches != nullptr) dpatches[((long)b * Np + (t - 1)) * D + d] = w;
    }
    if (dpos != nullptr) dpos[i] = accumulate ? dpos[i] + s : s;
    if (t == 0 && dhead != nullptr) dhead[d] = accumulate ? dhead[d] + s : s;
  }
}

// ---- fused Adam / AdamW over the flat arena --------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, bf16_t* __restrict__ pb, long n, float lr, float b1,
                            float b2, float eps, float wd, int decoupled, float bc1, float bc2_rsqrt,
                            float gscale) {
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float step_size = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);  // (non-temporal: see adam_dev_kernel)
    f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = gv[e] * gscale;
      float pp = pv[e];
      if (wd != 0.f) {
        if (decoupled) pp *= (1.f - lr * wd);
        else gg += wd * pp;
      }
      const float mm = b1 * mv[e] + (1.f - b1) * gg;
      const float v2 = b2 * vv[e] + (1.f - b2) * gg * gg;
      const float denom = sqrtf(v2) * bc2_rsqrt + eps;
      pp -= step_size * (mm / denom);
      pv[e] = pp; mv[e] = mm; vv[e] = v2;
    }
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p) + i);  // (see adam_dev_kernel)
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
    if (pb != nullptr)
      __builtin_nontemporal_store(u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])}, reinterpret_cast<u32x2*>(pb) + i);
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    float gg = g[i] * gscale, pp = p[i];
    if (wd != 0.f) {
      if (decoupled) pp *= (1.f - lr * wd);
      else gg += wd * pp;
    }
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float v2 = b2 * v[i] + (1.f - b2) * gg * gg;
    pp -= step_size * (mm / (sqrtf(v2) * bc2_rsqrt + eps));
    p[i] = pp; m[i] = mm; v[i] = v2;
    if (pb != nullptr) pb[i] = f32_to_bf16(pp);
  }
}

// Same update with the step-dependent scalars read from device memory (hyper[0..7] = lr, beta1,
// beta2, eps, weight_decay, bias_corr1, 1/sqrt(bias_corr2), grad_scale): the launch is then
// replayable from a captured hipGraph while the host refreshes the 32-byte record each step.
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, bf16_t* __restrict__ pb, long n,
                                const float* __restrict__ hyper, int decoupled) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
  const float bc1 = hyper[5], bc2_rsqrt = hyper[6], gscale = hyper[7];
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  const float step_size = lr / bc1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    // Non-temporal accesses (round 6): every stream of the update is touched ONCE per step and is far larger than the caches (the zoo
    // UNet: 26 GB per step), and the update runs on a side lane beside the backward pass — without the hint its lines push the main
    // queue's operands out of L2 / MALL.  UNet 64^2 x 8 step 54.05 -> 53.39 ms, ViT-B/16 18.07 -> 18.03 (three alternating pairs on one
    // box, profiles/r06/adam_nontemporal_ab.txt).
    f32x4 pv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(p) + i);
    f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
    f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
    f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = gv[e] * gscale;
      float pp = pv[e];
      if (wd != 0.f) {
        if (decoupled) pp *= (1.f - lr * wd);
        else gg += wd * pp;
      }
      const float mm = b1 * mv[e] + (1.f - b1) * gg;
      const float v2 = b2 * vv[e] + (1.f - b2) * gg * gg;
      pp -= step_size * (mm / (sqrtf(v2) * bc2_rsqrt + eps));
      pv[e] = pp; mv[e] = mm; vv[e] = v2;
    }
    __builtin_nontemporal_store(pv, reinterpret_cast<f32x4*>(p) + i);
    __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i);
    __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
    if (pb != nullptr)
      __builtin_nontemporal_store(u32x2{pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])}, reinterpret_cast<u32x2*>(pb) + i);
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    float gg = g[i] * gscale, pp = p[i];
    if (wd != 0.f) {
      if (decoupled) pp *= (1.f - lr * wd);
      else gg += wd * pp;
    }
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float v2 = b2 * v[i] + (1.f - b2) * gg * gg;
    pp -= step_size * (mm / (sqrtf(v2) * bc2_rsqrt + eps));
    p[i] = pp; m[i] = mm; v[i] = v2;
    if (pb != nullptr) pb[i] = f32_to_bf16(pp);
  }
}

__global__ void sumsq_kernel(const float* __restrict__ g, float* __restrict__ out, long n) {
  __shared__ float red[4];
  const long stride = (long)gridDim.x * blockDim.x;
  float s = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// ---- softmax cross-entropy: one wave per sample ----------------------------------------------------
__global__ void softmax_xent_kernel(const float* __restrict__ logits, const long long* __restrict__ labels,
                                    float* __restrict__ loss_sum, float* __restrict__ dlogits, int B, int C,
                                    float gscale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* lr = logits + (long)row * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += __expf(lr[c] - mx);
  se = wave_sum(se);
  const int y = (int)labels[row];
  const float lse = mx + __logf(se);
  if (lane == 0) atomicAdd(loss_sum, lse - lr[y]);
  if (dlogits != nullptr) {
    float* dr = dlogits + (long)row * C;
    const float inv = 1.f / se;
    for (int c = lane; c < C; c += 64) {
      const float pr = __expf(lr[c] - mx) * inv;
      dr[c] = (pr - (c == y ? 1.f : 0.f)) * gscale;
    }
  }
}

}  // namespace

int cfhip_internal_colreduce_f32(const float* partials, int R, int D, float* out, int accumulate,
                                 hipStream_t s) {
  hipLaunchKernelGGL(colreduce_f32_kernel, dim3((D + 31) / 32), dim3(1024), 0, s, partials, R, D, out,
                     accumulate, (const float*)nullptr, (float*)nullptr);
  CFHIP_CHECK_LAUNCH("colreduce_f32");
  return CFHIP_OK;
}

// ---- EMA of the parameters (reference modules/common.py:126-137): ema = (1 - decay) * p + decay * ema ----------
// two rounded products and one rounded sum, exactly the reference's expression (no FMA contraction): bit-exact
__global__ void ema_update_kernel(float* __restrict__ ema, const float* __restrict__ p, long n, float one_minus_decay,
                                  float decay) {
#pragma clang fp contract(off)
  const long n4 = n >> 2;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += stride) {
    const f32x4 pv = reinterpret_cast<const f32x4*>(p)[i];
    f32x4 ev = reinterpret_cast<const f32x4*>(ema)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = one_minus_decay * pv[e];
      const float b = decay * ev[e];
      ev[e] = a + b;
    }
    reinterpret_cast<f32x4*>(ema)[i] = ev;
  }
  for (long i = (n4 << 2) + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += stride) {
    const float a = one_minus_decay * p[i];
    const float b = decay * ema[i];
    ema[i] = a + b;
  }
}

extern "C" int cfhip_ema_update(float* ema, const float* p, int64_t n, float one_minus_decay, float decay, void* stream) {
  CFHIP_REQUIRE(ema && p && n > 0, "ema_update: bad arguments");
  CFHIP_REQUIRE(((uintptr_t)ema & 15) == 0 && ((uintptr_t)p & 15) == 0, "ema_update: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(ema_update_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, ema, p,
                     (long)n, one_minus_decay, decay);
  CFHIP_CHECK_LAUNCH("ema_update");
  return CFHIP_OK;
}

// One wavefront that idles for `us` microseconds of the 100 MHz constant-rate counter.  Host use only: the stream
// self-check of functional.distinct_stream launches it on two streams at once; if both finish in the time of one,
// the streams sit on different hardware queues (ROCclr multiplexes streams onto GPU_MAX_HW_QUEUES queues, and two
// streams that share a queue run their kernels back to back).
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

extern "C" int cfhip_spin(int microseconds, void* stream) {
  CFHIP_REQUIRE(microseconds > 0 && microseconds <= 100000, "spin: 1..100000 us");
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)microseconds * 100);
  CFHIP_CHECK_LAUNCH("spin");
  return CFHIP_OK;
}

extern "C" int cfhip_colreduce_f32(const float* x, float* out, int R, int D, int accumulate, void* stream) {
  CFHIP_REQUIRE(x && out && R > 0 && D > 0, "colreduce_f32: bad arguments");
  return cfhip_internal_colreduce_f32(x, R, D, out, accumulate, (hipStream_t)stream);
}

extern "C" int cfhip_colreduce2_f32(const float* x_a, float* out_a, const float* x_b, float* out_b, int R, int D, int accumulate, void* stream) {
  CFHIP_REQUIRE(x_a && out_a && x_b && out_b && R > 0 && D > 0, "colreduce2_f32: bad arguments");
  hipLaunchKernelGGL(colreduce_f32_kernel, dim3((D + 31) / 32, 2), dim3(1024), 0, (hipStream_t)stream, x_a, R, D, out_a, accumulate, x_b, out_b);
  CFHIP_CHECK_LAUNCH("colreduce2_f32");
  return CFHIP_OK;
}

extern "C" size_t cfhip_colsum_workspace(int M, int N) {
  return (size_t)colsum_slices(M) * (size_t)N * sizeof(float);
}

extern "C" int cfhip_colsum_bf16(const void* X, float* out, int M, int N, int64_t ldx, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(X && out && M > 0 && N > 0, "colsum: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool fast = (N % 8 == 0) && (ldx % 8 == 0) && (((uintptr_t)X & 15) == 0);
  if (!fast) {
    hipLaunchKernelGGL(colsum_generic_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const bf16_t*)X,
                       out, M, N, (long)ldx, accumulate);
    CFHIP_CHECK_LAUNCH("colsum_generic");
    return CFHIP_OK;
  }
  const int slices = colsum_slices(M);
  const size_t need = (size_t)slices * N * sizeof(float);
  if (workspace == nullptr || workspace_bytes < need) {
    cfhip_set_error("colsum: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return CFHIP_ERR_WORKSPACE;
  }
  const int rows_per_slice = (M + slices - 1) / slices;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 511) / 512, slices), dim3(CS_WAVES * 64), 0, s,
                     (const bf16_t*)X, (float*)workspace, M, N, (long)ldx, rows_per_slice);
  CFHIP_CHECK_LAUNCH("colsum_partial");
  return cfhip_internal_colreduce_f32((const float*)workspace, slices, N, out, accumulate, s);
}

extern "C" int cfhip_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  CFHIP_REQUIRE(src && dst && n >= 0, "cast_f32_to_bf16: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "cast_f32_to_bf16: misaligned");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, src, (bf16_t*)dst, (long)n);
  CFHIP_CHECK_LAUNCH("cast_f32_to_bf16");
  return CFHIP_OK;
}
extern "C" int cfhip_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  CFHIP_REQUIRE(src && dst && n >= 0, "cast_bf16_to_f32: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 15) == 0, "cast_bf16_to_f32: misaligned");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, dst, (long)n);
  CFHIP_CHECK_LAUNCH("cast_bf16_to_f32");
  return CFHIP_OK;
}

extern "C" int cfhip_split_f32_bf16x2(const float* src, void* hi, void* lo, int64_t n, void* stream) {
  CFHIP_REQUIRE(src && hi && lo && n >= 0, "split_f32_bf16x2: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)hi & 7) == 0 && ((uintptr_t)lo & 7) == 0, "split_f32_bf16x2: misaligned");
  hipLaunchKernelGGL(split_f32_bf16x2_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)hi,
                     (bf16_t*)lo, (long)n);
  CFHIP_CHECK_LAUNCH("split_f32_bf16x2");
  return CFHIP_OK;
}
extern "C" int cfhip_join_bf16x2_f32(const void* hi, const void* lo, float* dst, int64_t n, void* stream) {
  CFHIP_REQUIRE(hi && lo && dst && n >= 0, "join_bf16x2_f32: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)dst & 15) == 0 && ((uintptr_t)hi & 7) == 0 && ((uintptr_t)lo & 7) == 0, "join_bf16x2_f32: misaligned");
  hipLaunchKernelGGL(join_bf16x2_f32_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)hi,
                     (const bf16_t*)lo, dst, (long)n);
  CFHIP_CHECK_LAUNCH("join_bf16x2_f32");
  return CFHIP_OK;
}

#define CFHIP_EW(NAME, MODE, A_, B_, OUT_)                                                         \
  CFHIP_REQUIRE(A_ && OUT_ && n >= 0, NAME ": bad arguments");                                      \
  if (n == 0) return CFHIP_OK;                                                                      \
  CFHIP_REQUIRE(((uintptr_t)(A_)&7) == 0 && ((uintptr_t)(B_)&7) == 0 && ((uintptr_t)(OUT_)&7) == 0, \
                NAME ": misaligned");                                                               \
  hipLaunchKernelGGL((ew_bf16_kernel<MODE>), dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0,          \
                     (hipStream_t)stream, (const bf16_t*)(A_), (const bf16_t*)(B_), (bf16_t*)(OUT_), \
                     (long)n);                                                                      \
  CFHIP_CHECK_LAUNCH(NAME);                                                                         \
  return CFHIP_OK;

extern "C" int cfhip_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
  CFHIP_EW("gelu_fwd", 0, x, (const void*)nullptr, y)
}
extern "C" int cfhip_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
  CFHIP_REQUIRE(x, "gelu_bwd: null x");
  CFHIP_EW("gelu_bwd", 1, dy, x, dx)
}
// ---- DDPM forward process and objective (samplers/schema.py:90-112; models/cv/diffusion.py:44-94) ----------------
// x_t[b] = sqrt_ac[t_b] * x[b] + sqrt_1mac[t_b] * noise[b]: two rounded products and a rounded sum like the
// reference's `w_net * net + w_noise * noise` (no FMA contraction) -> the f32 result is bit-exact
template <bool OUT_F32>
__global__ void q_sample_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                const int64_t* __restrict__ t, const float* __restrict__ sqrt_ac,
                                const float* __restrict__ sqrt_1mac, void* __restrict__ out, long B, long inner) {
#pragma clang fp contract(off)
  const long total = B * inner;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / inner;
    const long tb = t[b];
    const float a = sqrt_ac[tb] * x[i];
    const float c = sqrt_1mac[tb] * noise[i];
    const float v = a + c;
    if (OUT_F32) reinterpret_cast<float*>(out)[i] = v;
    else reinterpret_cast<bf16_t*>(out)[i] = f32_to_bf16(v);
  }
}

// loss_sum += sum_b mean_inner (pred - target)^2 ; dpred = grad_scale * 2 (pred - target) / inner   (bf16 pred)
__global__ void mse_loss_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                float* __restrict__ loss_sum, bf16_t* __restrict__ dpred, long total, long inner,
                                float grad_scale) {
  __shared__ float red[4];
  const long stride = (long)gridDim.x * blockDim.x;
  const float k = 2.0f * grad_scale / (float)inner;
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const float d = bf16_to_f32(pred[i]) - target[i];
    acc += d * d;
    if (dpred != nullptr) dpred[i] = f32_to_bf16(k * d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, (red[0] + red[1] + red[2] + red[3]) / (float)inner);
}

// per-sample objective of DDPMStep.loss_fn: grid.y = sample, grid.x walks its `inner` elements
template <bool L1>
__global__ void diffusion_loss_kernel(const bf16_t* __restrict__ pred, const float* __restrict__ target,
                                      const float* __restrict__ weight, float* __restrict__ per_sample,
                                      bf16_t* __restrict__ dpred, long inner) {
  __shared__ float red[4];
  const long b = blockIdx.y;
  const long base = b * inner;
  const float k = weight[b] / (float)inner;
  float acc = 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < inner; i += (long)gridDim.x * blockDim.x) {
    const float d = bf16_to_f32(pred[base + i]) - target[base + i];
    acc += L1 ? fabsf(d) : d * d;
    if (dpred != nullptr) dpred[base + i] = f32_to_bf16(L1 ? (d > 0.f ? k : (d < 0.f ? -k : 0.f)) : 2.0f * k * d);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(per_sample + b, (red[0] + red[1] + red[2] + red[3]) / (float)inner);
}

extern "C" int cfhip_diffusion_loss(const void* pred, const float* target, const float* weight, float* per_sample,
                                    void* dpred, int64_t B, int64_t inner, int loss_type, void* stream) {
  CFHIP_REQUIRE(pred && target && weight && per_sample && B > 0 && B <= 65535 && inner > 0, "diffusion_loss: bad arguments");
  CFHIP_REQUIRE(loss_type == 0 || loss_type == 1, "diffusion_loss: loss_type %d (0 = l2, 1 = l1)", loss_type);
  const dim3 grid(grid_for(inner, 256, 256), (unsigned)B);
  if (loss_type == 1)
    hipLaunchKernelGGL((diffusion_loss_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, target,
                       weight, per_sample, (bf16_t*)dpred, (long)inner);
  else
    hipLaunchKernelGGL((diffusion_loss_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pred, target,
                       weight, per_sample, (bf16_t*)dpred, (long)inner);
  CFHIP_CHECK_LAUNCH("diffusion_loss");
  return CFHIP_OK;
}

extern "C" int cfhip_q_sample(const float* x, const float* noise, const int64_t* t, const float* sqrt_ac,
                              const float* sqrt_1mac, void* out, int out_is_f32, int64_t B, int64_t inner, void* stream) {
  CFHIP_REQUIRE(x && noise && t && sqrt_ac && sqrt_1mac && out && B > 0 && inner > 0, "q_sample: bad arguments");
  const dim3 grid(grid_for(B * inner, 256));
  if (out_is_f32)
    hipLaunchKernelGGL((q_sample_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac,
                       out, (long)B, (long)inner);
  else
    hipLaunchKernelGGL((q_sample_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac,
                       out, (long)B, (long)inner);
  CFHIP_CHECK_LAUNCH("q_sample");
  return CFHIP_OK;
}

extern "C" int cfhip_mse_loss(const void* pred, const float* target, float* loss_sum, void* dpred, int64_t B,
                              int64_t inner, float grad_scale, void* stream) {
  CFHIP_REQUIRE(pred && target && loss_sum && B > 0 && inner > 0, "mse_loss: bad arguments");
  hipLaunchKernelGGL(mse_loss_kernel, dim3(grid_for(B * inner, 256, 1024)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)pred, target, loss_sum, (bf16_t*)dpred, (long)(B * inner), (long)inner, grad_scale);
  CFHIP_CHECK_LAUNCH("mse_loss");
  return CFHIP_OK;
}

// ---- batched strided copy of bf16 rows: dst[b * dst_bs + i] = src[b * src_bs + i], i < n (n % 4 == 0) -------------
// torch.cat([a, b], dim=1) of NCHW tensors (the UNet's skip connections, unet.py:311-316) = two of these; its
// backward (the split) two more.
__global__ void copy_strided_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long batch, long n,
                                         long src_bs, long dst_bs) {
  const long n4 = n >> 2;
  const long total = batch * n4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / n4, j = (i - b * n4) * 4;
    *reinterpret_cast<u32x2*>(dst + b * dst_bs + j) = *reinterpret_cast<const u32x2*>(src + b * src_bs + j);
  }
}

extern "C" int cfhip_copy_strided_bf16(const void* src, void* dst, int64_t batch, int64_t n, int64_t src_batch_stride,
                                       int64_t dst_batch_stride, void* stream) {
  CFHIP_REQUIRE(src && dst && batch > 0 && n > 0, "copy_strided_bf16: bad arguments");
  CFHIP_REQUIRE(n % 4 == 0 && src_batch_stride % 4 == 0 && dst_batch_stride % 4 == 0 && ((uintptr_t)src & 7) == 0 &&
                    ((uintptr_t)dst & 7) == 0,
                "copy_strided_bf16: lengths / strides must be multiples of 4 elements, pointers 8-byte aligned");
  hipLaunchKernelGGL(copy_strided_bf16_kernel, dim3(grid_for(batch * (n / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, (long)batch, (long)n, (long)src_batch_stride,
                     (long)dst_batch_stride);
  CFHIP_CHECK_LAUNCH("copy_strided_bf16");
  return CFHIP_OK;
}

// Two such copies with the same batch count in ONE launch: both halves of a channel concatenation, or both halves of its backward split
// (the UNet's twelve skip connections: 48 launches of 6-7 us on the critical queue of a step became 24).
__global__ void copy_strided2_bf16_kernel(const bf16_t* __restrict__ sa, bf16_t* __restrict__ da, long na, long sa_bs, long da_bs,
                                          const bf16_t* __restrict__ sb, bf16_t* __restrict__ db, long nb, long sb_bs, long db_bs,
                                          long batch) {
  const long na4 = na >> 2, n4 = na4 + (nb >> 2);
  const long total = batch * n4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long b = i / n4, r = i - b * n4;
    if (r < na4) *reinterpret_cast<u32x2*>(da + b * da_bs + r * 4) = *reinterpret_cast<const u32x2*>(sa + b * sa_bs + r * 4);
    else *reinterpret_cast<u32x2*>(db + b * db_bs + (r - na4) * 4) = *reinterpret_cast<const u32x2*>(sb + b * sb_bs + (r - na4) * 4);
  }
}

extern "C" int cfhip_copy_strided2_bf16(const void* src_a, void* dst_a, int64_t n_a, int64_t src_a_batch_stride, int64_t dst_a_batch_stride,
                                        const void* src_b, void* dst_b, int64_t n_b, int64_t src_b_batch_stride, int64_t dst_b_batch_stride,
                                        int64_t batch, void* stream) {
  CFHIP_REQUIRE(src_a && dst_a && src_b && dst_b && batch > 0 && n_a > 0 && n_b > 0, "copy_strided2_bf16: bad arguments");
  CFHIP_REQUIRE(n_a % 4 == 0 && n_b % 4 == 0 && src_a_batch_stride % 4 == 0 && dst_a_batch_stride % 4 == 0 && src_b_batch_stride % 4 == 0 &&
                    dst_b_batch_stride % 4 == 0 && (((uintptr_t)src_a | (uintptr_t)dst_a | (uintptr_t)src_b | (uintptr_t)dst_b) & 7) == 0,
                "copy_strided2_bf16: lengths / strides must be multiples of 4 elements, pointers 8-byte aligned");
  hipLaunchKernelGGL(copy_strided2_bf16_kernel, dim3(grid_for(batch * ((n_a + n_b) / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src_a, (bf16_t*)dst_a, (long)n_a, (long)src_a_batch_stride, (long)dst_a_batch_stride,
                     (const bf16_t*)src_b, (bf16_t*)dst_b, (long)n_b, (long)src_b_batch_stride, (long)dst_b_batch_stride, (long)batch);
  CFHIP_CHECK_LAUNCH("copy_strided2_bf16");
  return CFHIP_OK;
}

// ---- GEGLU (activations.py:150-158): out[m][c] = vg[m][c] * gelu(vg[m][L + c]); 4 bf16 per thread ------------------
template <bool BWD>
__global__ void geglu_kernel(const bf16_t* __restrict__ vg, const bf16_t* __restrict__ dy, bf16_t* __restrict__ out,
                             long M, int L) {
  const int l4 = L >> 2;
  const long total = M * l4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += stride) {
    const long m = i / l4;
    const int c = (int)(i - m * l4) * 4;
    const u32x2 wv = *reinterpret_cast<const u32x2*>(vg + m * 2 * L + c);
    const u32x2 wg = *reinterpret_cast<const u32x2*>(vg + m * 2 * L + L + c);
    const float v[4] = {bf16lo(wv[0]), bf16hi(wv[0]), bf16lo(wv[1]), bf16hi(wv[1])};
    const float g[4] = {bf16lo(wg[0]), bf16hi(wg[0]), bf16lo(wg[1]), bf16hi(wg[1])};
    if (!BWD) {
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[e] * gelu_erf_f(g[e]);
      *reinterpret_cast<u32x2*>(out + m * L + c) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    } else {
      const u32x2 wd = *reinterpret_cast<const u32x2*>(dy + m * L + c);
      const float d[4] = {bf16lo(wd[0]), bf16hi(wd[0]), bf16lo(wd[1]), bf16hi(wd[1])};
      float dv[4], dg[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dv[e] = d[e] * gelu_erf_f(g[e]);
        dg[e] = d[e] * v[e] * gelu_erf_grad_f(g[e]);
      }
      *reinterpret_cast<u32x2*>(out + m * 2 * L + c) = u32x2{pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3])};
      *reinterpret_cast<u32x2*>(out + m * 2 * L + L + c) = u32x2{pack_bf16x2(dg[0], dg[1]), pack_bf16x2(dg[2], dg[3])};
    }
  }
}

extern "C" int cfhip_geglu_fwd(const void* vg, void* out, int64_t M, int L, void* stream) {
  CFHIP_REQUIRE(vg && out && M > 0 && L > 0 && L % 4 == 0, "geglu_fwd: bad arguments (L must be a multiple of 4)");
  hipLaunchKernelGGL((geglu_kernel<false>), dim3(grid_for(M * (L / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)vg, (const bf16_t*)nullptr, (bf16_t*)out, (long)M, L);
  CFHIP_CHECK_LAUNCH("geglu_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_geglu_bwd(const void* dy, const void* vg, void* dvg, int64_t M, int L, void* stream) {
  CFHIP_REQUIRE(dy && vg && dvg && M > 0 && L > 0 && L % 4 == 0, "geglu_bwd: bad arguments (L must be a multiple of 4)");
  hipLaunchKernelGGL((geglu_kernel<true>), dim3(grid_for(M * (L / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)vg, (const bf16_t*)dy, (bf16_t*)dvg, (long)M, L);
  CFHIP_CHECK_LAUNCH("geglu_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_quick_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
  CFHIP_EW("quick_gelu_fwd", 3, x, (const void*)nullptr, y)
}
extern "C" int cfhip_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
  CFHIP_REQUIRE(x, "quick_gelu_bwd: null x");
  CFHIP_EW("quick_gelu_bwd", 4, dy, x, dx)
}
extern "C" int cfhip_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  CFHIP_REQUIRE(b, "add_bf16: null b");
  CFHIP_EW("add_bf16", 2, a, b, out)
}

extern "C" int cfhip_transpose_bf16(const void* src, void* dst, int R, int C, int64_t ld_src,
                                    int64_t ld_dst, void* stream) {
  CFHIP_REQUIRE(src && dst && R > 0 && C > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)src, (bf16_t*)dst, R, C, (long)ld_src,
                     (long)ld_dst);
  CFHIP_CHECK_LAUNCH("transpose_bf16");
  return CFHIP_OK;
}

extern "C" int cfhip_im2row(const void* img, int img_is_bf16, void* rows, int B, int C, int Hh, int Ww,
                            int P, void* stream) {
  CFHIP_REQUIRE(img && rows && B > 0 && C > 0 && P > 0, "im2row: bad arguments");
  CFHIP_REQUIRE(Hh % P == 0 && Ww % P == 0, "im2row: image %dx%d not divisible by patch %d", Hh, Ww, P);
  CFHIP_REQUIRE(P % 4 == 0 && Ww % 4 == 0, "im2row: patch size and width must be multiples of 4");
  CFHIP_REQUIRE(((uintptr_t)img & 15) == 0 && ((uintptr_t)rows & 7) == 0, "im2row: misaligned");
  const long total4 = (long)B * C * Hh * Ww / 4;
  if (img_is_bf16)
    hipLaunchKernelGGL((im2row_kernel<true>), dim3(grid_for(total4, 256)), dim3(256), 0,
                       (hipStream_t)stream, img, (bf16_t*)rows, B, C, Hh, Ww, P);
  else
    hipLaunchKernelGGL((im2row_kernel<false>), dim3(grid_for(total4, 256)), dim3(256), 0,
                       (hipStream_t)stream, img, (bf16_t*)rows, B, C, Hh, Ww, P);
  CFHIP_CHECK_LAUNCH("im2row");
  return CFHIP_OK;
}

extern "C" int cfhip_assemble_tokens_fwd(const void* patches, const float* head_token, const float* pos,
                                         void* x0, int x0_is_f32, int B, int Np, int D, void* stream) {
  CFHIP_REQUIRE(patches && head_token && pos && x0, "assemble_tokens_fwd: null pointer");
  CFHIP_REQUIRE(B > 0 && Np > 0 && D > 0 && D % 4 == 0, "assemble_tokens_fwd: D must be a multiple of 4");
  CFHIP_REQUIRE(((uintptr_t)patches & 7) == 0 && ((uintptr_t)x0 & (x0_is_f32 ? 15 : 7)) == 0 &&
                    ((uintptr_t)head_token & 15) == 0 && ((uintptr_t)pos & 15) == 0,
                "assemble_tokens_fwd: misaligned");
  const long total = (long)B * (Np + 1) * (D / 4);
  if (x0_is_f32)
    hipLaunchKernelGGL((assemble_fwd_kernel<true>), dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)patches, head_token, pos, x0, B, Np, D);
  else
    hipLaunchKernelGGL((assemble_fwd_kernel<false>), dim3(grid_for(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)patches, head_token, pos, x0, B, Np, D);
  CFHIP_CHECK_LAUNCH("assemble_tokens_fwd");
  return CFHIP_OK;
}
extern "C" int cfhip_assemble_tokens_bwd(const void* dx0, void* dpatches, float* dhead_token,
                                         float* dpos, int B, int Np, int D, int accumulate,
                                         void* stream) {
  CFHIP_REQUIRE(dx0 && B > 0 && Np > 0 && D > 0, "assemble_tokens_bwd: bad arguments");
  const long total = (long)(Np + 1) * D;
  if (D % 8 == 0 && ((uintptr_t)dx0 & 15) == 0 && (dpatches == nullptr || ((uintptr_t)dpatches & 15) == 0))
    hipLaunchKernelGGL(assemble_bwd_vec_kernel, dim3((unsigned)((Np + 1) * ((D + 255) / 256))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dx0, (bf16_t*)dpatches, dhead_token, dpos, B, Np, D, accumulate);
  else
    hipLaunchKernelGGL(assemble_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dx0, (bf16_t*)dpatches, dhead_token, dpos, B, Np, D, accumulate);
  CFHIP_CHECK_LAUNCH("assemble_tokens_bwd");
  return CFHIP_OK;
}

extern "C" int cfhip_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               int decoupled, int step, float grad_scale, void* stream) {
  CFHIP_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "adam_step: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)m & 15) == 0 &&
                    ((uintptr_t)v & 15) == 0 && ((uintptr_t)p_bf16 & 7) == 0,
                "adam_step: arena pointers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, p, g,
                     m, v, (bf16_t*)p_bf16, (long)n, lr, beta1, beta2, eps, weight_decay, decoupled,
                     (float)bc1, (float)(1.0 / sqrt(bc2)), grad_scale);
  CFHIP_CHECK_LAUNCH("adam_step");
  return CFHIP_OK;
}

extern "C" int cfhip_adam_step_dev(float* p, const float* g, float* m, float* v, void* p_bf16,
                                   int64_t n, const float* hyper, int decoupled, void* stream) {
  CFHIP_REQUIRE(p && g && m && v && hyper && n >= 0, "adam_step_dev: bad arguments");
  if (n == 0) return CFHIP_OK;
  CFHIP_REQUIRE(((uintptr_t)p & 15) == 0 && ((uintptr_t)g & 15) == 0 && ((uintptr_t)m & 15) == 0 &&
                    ((uintptr_t)v & 15) == 0 && ((uintptr_t)p_bf16 & 7) == 0,
                "adam_step_dev: arena pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     p, g, m, v, (bf16_t*)p_bf16, (long)n, hyper, decoupled);
  CFHIP_CHECK_LAUNCH("adam_step_dev");
  return CFHIP_OK;
}

extern "C" int cfhip_sumsq_f32(const float* g, float* out, int64_t n, void* stream) {
  CFHIP_REQUIRE(g && out && n >= 0, "sumsq: bad arguments");
  if (n == 0) return CFHIP_OK;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 256 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, g,
                     out, (long)n);
  CFHIP_CHECK_LAUNCH("sumsq");
  return CFHIP_OK;
}

extern "C" int cfhip_softmax_xent(const float* logits, const int64_t* labels, float* loss_sum,
                                  float* dlogits, int B, int C, float grad_scale, void* stream) {
  CFHIP_REQUIRE(logits && labels && loss_sum && B > 0 && C > 0, "softmax_xent: bad arguments");
  hipLaunchKernelGGL(softmax_xent_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits,
                     (const long long*)labels, loss_sum, dlogits, B, C, grad_scale);
  CFHIP_CHECK_LAUNCH("softmax_xent");
  return CFHIP_OK;
}
