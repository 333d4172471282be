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
// workgroup partials — no global atomics.  The order in which a workgroup's waves reach the LDS
// atomics is not fixed, so dgamma / dbeta are reproducible to fp32 rounding (~1e-7), not bitwise.
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
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) {
  *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<u32x2*>(p) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
}

// 64-lane sum without the LDS pipe: 4 DPP steps inside each 16-lane row (quad swaps, half-row mirror,
// row mirror), then the 4 row totals are read through v_readlane.  (A __shfl_xor chain lowers to 6
// dependent ds_bpermute round trips, which made the reductions the latency of every row.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(o);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror -> every lane holds its 16-lane row total
  const int iv = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48)));
}

// EXACT: D == NCH * 256 (no column masks).  The next row's loads are issued before the current row
// is reduced (register double buffer), so two rows per wave are in flight.
// 4 consecutive elements of a row as f32, from a bf16 (8-byte load) or an f32 (16-byte load) tensor
template <typename XT> struct Row4;
template <> struct Row4<bf16_t> {
  u32x2 w;
  __device__ __forceinline__ void load(const bf16_t* p) { w = *reinterpret_cast<const u32x2*>(p); }
  __device__ __forceinline__ void zero() { w = u32x2{0u, 0u}; }
  __device__ __forceinline__ void get(float (&v)[4]) const {
    v[0] = bf16lo(w[0]); v[1] = bf16hi(w[0]); v[2] = bf16lo(w[1]); v[3] = bf16hi(w[1]);
  }
};
template <> struct Row4<float> {
  f32x4 w;
  __device__ __forceinline__ void load(const float* p) { w = *reinterpret_cast<const f32x4*>(p); }
  __device__ __forceinline__ void zero() { w = f32x4{0.f, 0.f, 0.f, 0.f}; }
  __device__ __forceinline__ void get(float (&v)[4]) const { v[0] = w[0]; v[1] = w[1]; v[2] = w[2]; v[3] = w[3]; }
};

template <int NCH, bool EXACT, typename XT, typename YT = bf16_t>
__global__ __launch_bounds__(LN_THREADS) void layernorm_fwd_kernel(
    const XT* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    YT* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int D,
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
      gm[c][e] = (EXACT || col + e < D) ? gamma[col + e] : 0.f;
      bt[c][e] = (EXACT || col + e < D) ? beta[col + e] : 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  Row4<XT> nxt[NCH];
  if (gw < M) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (EXACT || col < D) nxt[c].load(x + (long)gw * xs + col); else nxt[c].zero();
    }
  }
  for (int row = gw; row < M; row += nw) {
    float v[NCH][4];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      nxt[c].get(v[c]);
      s += (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    if (row + nw < M) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (EXACT || col < D) nxt[c].load(x + (long)(row + nw) * xs + col); else nxt[c].zero();
      }
    }
    const float mean = wave_sum_dpp(s) * inv_d;
    float sq = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (EXACT || col < D) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[c][e] - mean; sq += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum_dpp(sq) * inv_d + eps);
    YT* yr = y + (long)row * ys;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      if (EXACT || col < D) {
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

// DO_DX / DO_PG select what the launch produces: the input gradient (critical path of backward:
// no per-column accumulators, high occupancy), the dgamma / dbeta partials (a pure streaming column
// reduction that can run on the side stream), or both in one pass.
template <int NCH, bool EXACT, typename XT, bool DO_DX, bool DO_PG>
__global__ __launch_bounds__(LN_THREADS) void layernorm_bwd_kernel(
    const bf16_t* __restrict__ dy, const XT* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ dx, float* __restrict__ partials, int M,
    int D, long dys, long xs, long dxs, const bf16_t* __restrict__ dx_add_lo, bf16_t* __restrict__ dx_lo) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][D]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * LN_WAVES + wave;
  const int nw = gridDim.x * LN_WAVES;
  if (DO_PG) {
    for (int i = threadIdx.x; i < 2 * D; i += LN_THREADS) red[i] = 0.f;
    __syncthreads();
  }

  float gm[NCH][4], dg[NCH][4], db[NCH][4];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      gm[c][e] = (EXACT || col + e < D) ? gamma[col + e] : 0.f;
      dg[c][e] = 0.f;
      db[c][e] = 0.f;
    }
  }
  const float inv_d = 1.0f / (float)D;
  Row4<XT> nx[NCH];
  u32x2 ny[NCH];
  float nmean = 0.f, nrstd = 0.f;
  if (gw < M) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      const bool ok = EXACT || col < D;
      if (ok) nx[c].load(x + (long)gw * xs + col); else nx[c].zero();
      ny[c] = ok ? *reinterpret_cast<const u32x2*>(dy + (long)gw * dys + col) : u32x2{0u, 0u};
    }
    nmean = mean_in[gw];
    nrstd = rstd_in[gw];
  }
  for (int row = gw; row < M; row += nw) {
    const float mean = nmean, rstd = nrstd;
    float xh[NCH][4], g[NCH][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float xv[4];
      nx[c].get(xv);
      const float dv[4] = {bf16lo(ny[c][0]), bf16hi(ny[c][0]), bf16lo(ny[c][1]), bf16hi(ny[c][1])};
      const int col = c * 256 + lane * 4;
      const bool ok = EXACT || col < D;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[c][e] = ok ? (xv[e] - mean) * rstd : 0.f;
        g[c][e] = dv[e] * gm[c][e];
        s1 += g[c][e];
        s2 += g[c][e] * xh[c][e];
        if (DO_PG) {
          dg[c][e] += dv[e] * xh[c][e];
          db[c][e] += dv[e];
        }
      }
    }
    // residual-gradient rows of THIS row and the operands of the NEXT row: all in flight together
    u32x2 addv[NCH], addl[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = c * 256 + lane * 4;
      addv[c] = (DO_DX && dx_add != nullptr && (EXACT || col < D))
                    ? *reinterpret_cast<const u32x2*>(dx_add + (long)row * dxs + col) : u32x2{0u, 0u};
      addl[c] = (DO_DX && dx_add_lo != nullptr && (EXACT || col < D))
                    ? *reinterpret_cast<const u32x2*>(dx_add_lo + (long)row * dxs + col) : u32x2{0u, 0u};
    }
    if (row + nw < M) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        const bool ok = EXACT || col < D;
        if (ok) nx[c].load(x + (long)(row + nw) * xs + col); else nx[c].zero();
        ny[c] = ok ? *reinterpret_cast<const u32x2*>(dy + (long)(row + nw) * dys + col) : u32x2{0u, 0u};
      }
      nmean = mean_in[row + nw];
      nrstd = rstd_in[row + nw];
    }
    if (DO_DX) {
      s1 = wave_sum_dpp(s1) * inv_d;
      s2 = wave_sum_dpp(s2) * inv_d;
      bf16_t* dxr = dx + (long)row * dxs;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int col = c * 256 + lane * 4;
        if (EXACT || col < D) {
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = rstd * (g[c][e] - s1 - xh[c][e] * s2);
          o[0] += bf16lo(addv[c][0]) + bf16lo(addl[c][0]); o[1] += bf16hi(addv[c][0]) + bf16hi(addl[c][0]);
          o[2] += bf16lo(addv[c][1]) + bf16lo(addl[c][1]); o[3] += bf16hi(addv[c][1]) + bf16hi(addl[c][1]);
          const u32x2 hw = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
          *reinterpret_cast<u32x2*>(dxr + col) = hw;
          if (dx_lo != nullptr) {  // second word of the gradient stream: what the bf16 rounding of `o` dropped
            const float r[4] = {o[0] - bf16lo(hw[0]), o[1] - bf16hi(hw[0]), o[2] - bf16lo(hw[1]), o[3] - bf16hi(hw[1])};
            store4(dx_lo + (long)row * dxs + col, r);
          }
        }
      }
    }
  }
  if (!DO_PG) return;
  // fold the workgroup's waves through LDS float atomics (ds_add_f32), one partial row per workgroup
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 256 + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (EXACT || col + e < D) {
        __hip_atomic_fetch_add(&red[col + e], dg[c][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&red[D + col + e], db[c][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  float* out = partials + (long)blockIdx.x * 2 * D;
  for (int i = threadIdx.x; i < 2 * D; i += LN_THREADS) out[i] = red[i];
}

// ---- fused backward, one launch: dx (+ residual-gradient add) AND the dgamma / dbeta partials -------------------
// Reads dy and x exactly once.  A HALF-wave (32 lanes) owns a row, so a wave works on two rows at a time and every
// lane owns G groups of 8 consecutive columns (col = g * 256 + 8 * (lane & 31)): every bf16 access is a 16-byte
// load / store (one instruction = 512 contiguous bytes per row), every f32 access two of them — the 8-byte accesses
// of the one-wave-per-row kernel above run at 0.55-0.7x the 16-byte rate (MI355X_MICROARCH.md).  dgamma / dbeta
// stay in registers across all rows a lane visits (lanes l and l + 32 own the same columns and are folded with one
// shuffle at the end), the workgroup's waves are folded through LDS in a FIXED order and the per-workgroup partial
// rows go to the same column reduce as before: no atomics anywhere, bitwise reproducible.
constexpr int LN2_WAVES = 4;
constexpr int LN2_THREADS = LN2_WAVES * 64;
constexpr int LN2_MAX_BLOCKS = 768;  // 3 workgroups of 4 waves per CU (160 VGPRs at D = 768)

// sum over the 32 lanes of each half-wave; every lane gets its own half's total
__device__ __forceinline__ float half_sum_dpp(float v, int half) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);  // every lane holds its 16-lane row total
  const int iv = __float_as_int(v);
  const float t0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16));
  const float t1 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
  return half ? t1 : t0;
}

template <typename XT> struct Row8;
template <> struct Row8<bf16_t> {
  u32x4 w;
  __device__ __forceinline__ void load(const bf16_t* p) { w = *reinterpret_cast<const u32x4*>(p); }
  __device__ __forceinline__ void zero() { w = u32x4{0u, 0u, 0u, 0u}; }
  __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = bf16lo(w[e]); v[2 * e + 1] = bf16hi(w[e]); }
  }
};
template <> struct Row8<float> {
  f32x4 a, b;
  __device__ __forceinline__ void load(const float* p) {
    a = *reinterpret_cast<const f32x4*>(p);
    b = *reinterpret_cast<const f32x4*>(p + 4);
  }
  __device__ __forceinline__ void zero() { a = f32x4{0.f, 0.f, 0.f, 0.f}; b = a; }
  __device__ __forceinline__ void get(float (&v)[8]) const {
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
  }
};

// EXACT: D = G * 256.  Otherwise (round 5: the UNet's 320- and 640-wide token rows, which the one-wave-per-row kernel above served at
// ~1 TB/s inside the step) the row is D_real < G * 256 columns wide, a multiple of 8: the lanes of the last group whose 8 columns lie
// beyond it load zeros, store nothing and keep zero sums — everything else, LDS layout included, works on the padded width.
// LO (round 6): the residual-gradient stream as TWO bf16 words per element (hi = bf16(g), lo = bf16(g - hi): 16 mantissa bits).  The
// reference's residual stream is f32 under autocast, so its gradient is f32 and only the matrix products see bf16(g); a one-word
// stream rounds the running sum twice per block, and where per-sample gradients cancel in a shared parameter (CLIP's contrastive
// loss: head token, positional encoding) that drift was 1.5 x the reference's own bf16 distance from fp32 (test_clip_b32_step_vs_oracle).
// `dx_add_lo` / `dx_lo`: the second words of dx_add / dx, same strides; the GEMMs keep reading the first word.
template <int G, typename XT, bool EXACT, bool LO = false>
__global__ __launch_bounds__(LN2_THREADS) void layernorm_bwd_fused_kernel(
    const bf16_t* __restrict__ dy, const XT* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const bf16_t* __restrict__ dx_add, bf16_t* __restrict__ dx, float* __restrict__ partials, int M,
    long dys, long xs, long dxs, int D_real, const bf16_t* __restrict__ dx_add_lo = nullptr, bf16_t* __restrict__ dx_lo = nullptr) {
  constexpr int D = G * 256;
  const int Dr = EXACT ? D : D_real;
  extern __shared__ __attribute__((aligned(16))) float lds[];  // gamma [D] | red [LN2_WAVES][2][D]
  float* gs = lds;
  float* red = lds + D;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int hl = lane & 31, half = lane >> 5;
  for (int i = threadIdx.x; i < D; i += LN2_THREADS) gs[i] = (EXACT || i < Dr) ? gamma[i] : 0.f;
  __syncthreads();

  float dg[G][8], db[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[g][e] = 0.f; db[g][e] = 0.f; }

  const float inv_d = 1.0f / (float)Dr;
  const int npairs = (M + 1) >> 1;
  const int nw = gridDim.x * LN2_WAVES;
  const bool tail_in = EXACT || (G - 1) * 256 + hl * 8 < Dr;  // this lane's columns of the last group exist
  for (int pair = blockIdx.x * LN2_WAVES + wave; pair < npairs; pair += nw) {
    const int row = 2 * pair + half;
    const bool ok = row < M;
    Row8<XT> rx[G];
    u32x4 ry[G], ra[G], rl[LO ? G : 1];
    float mean = 0.f, rstd = 0.f;
    if (ok) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int col = g * 256 + hl * 8;
        if (EXACT || g < G - 1 || tail_in) {
          rx[g].load(x + (long)row * xs + col);
          ry[g] = *reinterpret_cast<const u32x4*>(dy + (long)row * dys + col);
          ra[g] = dx_add != nullptr ? *reinterpret_cast<const u32x4*>(dx_add + (long)row * dxs + col) : u32x4{0u, 0u, 0u, 0u};
          if (LO) rl[g] = dx_add_lo != nullptr ? *reinterpret_cast<const u32x4*>(dx_add_lo + (long)row * dxs + col) : u32x4{0u, 0u, 0u, 0u};
        } else {
          rx[g].zero();
          ry[g] = u32x4{0u, 0u, 0u, 0u};
          ra[g] = ry[g];
          if (LO) rl[g] = ry[g];
        }
      }
      mean = mean_in[row];
      rstd = rstd_in[row];
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) { rx[g].zero(); ry[g] = u32x4{0u, 0u, 0u, 0u}; ra[g] = ry[g]; if (LO) rl[g] = ry[g]; }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float xv[8];
      rx[g].get(xv);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(gs + g * 256 + hl * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(gs + g * 256 + hl * 8 + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dv = (e & 1) ? bf16hi(ry[g][e >> 1]) : bf16lo(ry[g][e >> 1]);
        const float xh = (xv[e] - mean) * rstd;  // 0 for an absent row (mean = rstd = x = 0)
        const float gg = dv * (e < 4 ? g0[e] : g1[e - 4]);
        s1 += gg;
        s2 += gg * xh;
        dg[g][e] += dv * xh;
        db[g][e] += dv;
      }
    }
    s1 = half_sum_dpp(s1, half) * inv_d;
    s2 = half_sum_dpp(s2, half) * inv_d;
    if (ok) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (!EXACT && g == G - 1 && !tail_in) continue;
        float xv[8], o[8];
        rx[g].get(xv);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gs + g * 256 + hl * 8);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gs + g * 256 + hl * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float dv = (e & 1) ? bf16hi(ry[g][e >> 1]) : bf16lo(ry[g][e >> 1]);
          const float av = (e & 1) ? bf16hi(ra[g][e >> 1]) : bf16lo(ra[g][e >> 1]);
          const float xh = (xv[e] - mean) * rstd;
          o[e] = rstd * (dv * (e < 4 ? g0[e] : g1[e - 4]) - s1 - xh * s2) + av;
          if (LO) o[e] += (e & 1) ? bf16hi(rl[g][e >> 1]) : bf16lo(rl[g][e >> 1]);
        }
        const u32x4 hw = u32x4{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
        *reinterpret_cast<u32x4*>(dx + (long)row * dxs + g * 256 + hl * 8) = hw;
        if (LO) {
          float r[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = o[e] - ((e & 1) ? bf16hi(hw[e >> 1]) : bf16lo(hw[e >> 1]));
          *reinterpret_cast<u32x4*>(dx_lo + (long)row * dxs + g * 256 + hl * 8) =
              u32x4{pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7])};
        }
      }
    }
  }
  // lanes l and l + 32 own the same columns: fold them, then the waves of the workgroup in a fixed order
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dg[g][e] += __shfl_xor(dg[g][e], 32, 64);
      db[g][e] += __shfl_xor(db[g][e], 32, 64);
    }
  if (half == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float* r0 = red + (wave * 2 + 0) * D + g * 256 + hl * 8;
      float* r1 = red + (wave * 2 + 1) * D + g * 256 + hl * 8;
      *reinterpret_cast<f32x4*>(r0) = f32x4{dg[g][0], dg[g][1], dg[g][2], dg[g][3]};
      *reinterpret_cast<f32x4*>(r0 + 4) = f32x4{dg[g][4], dg[g][5], dg[g][6], dg[g][7]};
      *reinterpret_cast<f32x4*>(r1) = f32x4{db[g][0], db[g][1], db[g][2], db[g][3]};
      *reinterpret_cast<f32x4*>(r1 + 4) = f32x4{db[g][4], db[g][5], db[g][6], db[g][7]};
    }
  }
  __syncthreads();
  float* out = partials + (long)blockIdx.x * 2 * Dr;
  for (int i = threadIdx.x; i < 2 * D; i += LN2_THREADS) {
    float v = red[i];  // wave 0: [dgamma | dbeta]
#pragma unroll
    for (int w = 1; w < LN2_WAVES; ++w) v += red[w * 2 * D + i];
    if (EXACT) {
      out[i] = v;
    } else {
      const int which = i >= D ? 1 : 0, c = i - which * D;
      if (c < Dr) out[which * Dr + c] = v;  // the partial rows are [dgamma | dbeta] of the REAL width
    }
  }
}

inline int ln2_grid(int M) {
  const int pairs = (M + 1) / 2;
  const int iters = (pairs + LN2_MAX_BLOCKS * LN2_WAVES - 1) / (LN2_MAX_BLOCKS * LN2_WAVES);
  int blocks = (pairs + LN2_WAVES * iters - 1) / (LN2_WAVES * iters);
  return blocks < 1 ? 1 : blocks;
}

inline int ln_grid(int M, int cap = 512) {
  int blocks = (M + LN_WAVES - 1) / LN_WAVES;
  if (blocks > cap) blocks = cap;  // 512 = 2 workgroups of 8 waves per CU
  if (blocks < 1) blocks = 1;
  return blocks;
}

}  // namespace

#define LN_DISPATCH(KERNEL, nch, ...) /* KERNEL(NCH, EXACT) */                                      \
  switch (nch) {                                                           \
    case 1: if (exact) { KERNEL(1, true); } else { KERNEL(1, false); } break;      \
    case 2: if (exact) { KERNEL(2, true); } else { KERNEL(2, false); } break;      \
    case 3: if (exact) { KERNEL(3, true); } else { KERNEL(3, false); } break;      \
    case 4: if (exact) { KERNEL(4, true); } else { KERNEL(4, false); } break;      \
    case 5: case 6: if (exact && nch == 6) { KERNEL(6, true); } else { KERNEL(6, false); } break; \
    default: if (exact && nch == 8) { KERNEL(8, true); } else { KERNEL(8, false); } break;        \
  }

extern "C" int cfhip_layernorm_fwd(const void* x, int dtype_flags, const float* gamma, const float* beta,
                                   void* y, float* mean, float* rstd, int M, int D,
                                   int64_t x_row_stride, int64_t y_row_stride, float eps, void* stream) {
  const int x_is_f32 = dtype_flags & 1, y_is_f32 = (dtype_flags >> 1) & 1;
  CFHIP_REQUIRE(x && gamma && beta && y, "layernorm_fwd: null pointer");
  CFHIP_REQUIRE((dtype_flags & ~3) == 0, "layernorm_fwd: dtype_flags = %d (bit 0: x is f32, bit 1: y is f32)", dtype_flags);
  CFHIP_REQUIRE(!y_is_f32 || x_is_f32, "layernorm_fwd: an f32 output is built for f32 rows only");
  CFHIP_REQUIRE(M > 0 && D > 0, "layernorm_fwd: empty problem");
  CFHIP_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm_fwd: D=%d must be a multiple of 4 and <= 2048", D);
  CFHIP_REQUIRE(x_row_stride % 4 == 0 && y_row_stride % 4 == 0, "layernorm_fwd: row strides must be multiples of 4");
  CFHIP_REQUIRE(((uintptr_t)x & (x_is_f32 ? 15 : 7)) == 0 && ((uintptr_t)y & (y_is_f32 ? 15 : 7)) == 0,
                "layernorm_fwd: x / y must be 8-byte (bf16) / 16-byte (f32) aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (D + 255) / 256;
  const bool exact = (D % 256) == 0;
  const int blocks = ln_grid(M, 1024);
#define LN_FWD(N_, EX_)                                                                           \
  do {                                                                                             \
    if (y_is_f32)                                                                                  \
      hipLaunchKernelGGL((layernorm_fwd_kernel<N_, EX_, float, float>), dim3(blocks), dim3(LN_THREADS), 0, s, \
                         (const float*)x, gamma, beta, (float*)y, mean, rstd, M, D,                 \
                         (long)x_row_stride, (long)y_row_stride, eps);                              \
    else if (x_is_f32)                                                                             \
      hipLaunchKernelGGL((layernorm_fwd_kernel<N_, EX_, float>), dim3(blocks), dim3(LN_THREADS), 0, s, \
                         (const float*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D,                \
                         (long)x_row_stride, (long)y_row_stride, eps);                              \
    else                                                                                           \
      hipLaunchKernelGGL((layernorm_fwd_kernel<N_, EX_, bf16_t>), dim3(blocks), dim3(LN_THREADS), 0, s, \
                         (const bf16_t*)x, gamma, beta, (bf16_t*)y, mean, rstd, M, D,               \
                         (long)x_row_stride, (long)y_row_stride, eps);                              \
  } while (0)
  LN_DISPATCH(LN_FWD, nch, 0)
#undef LN_FWD
  CFHIP_CHECK_LAUNCH("layernorm_fwd");
  return CFHIP_OK;
}

extern "C" size_t cfhip_layernorm_bwd_workspace(int M, int D) {
  const int rows = ln_grid(M) > ln2_grid(M) ? ln_grid(M) : ln2_grid(M);  // either kernel may be picked
  return (size_t)rows * 2 * (size_t)D * sizeof(float);
}

static int g_ln_fused = 1;  // "ln_bwd_fused" option: 1 = the one-launch kernel when both outputs are asked for (default); 2 = only for D % 256 == 0 (rounds 3-4); 0 = never
int cfhip_internal_set_ln_fused(int v) {
  g_ln_fused = v;
  return CFHIP_OK;
}

// the column sums of the per-workgroup partial rows [rows][2 D] (dgamma | dbeta) -> dgamma, dbeta
static int ln_bwd_reduce(float* partials, int rows, int D, float* dgamma, float* dbeta, int accumulate_param_grads, hipStream_t s) {
  if (dgamma != nullptr && dbeta == dgamma + D)
    return cfhip_internal_colreduce_f32(partials, rows, 2 * D, dgamma, accumulate_param_grads, s);
  // separate destinations: both halves into row 0 of the partials first (colreduce takes a dense [R][D] matrix)
  int rc = cfhip_internal_colreduce_f32(partials, rows, 2 * D, partials, 0, s);
  if (rc != CFHIP_OK) return rc;
  if (dgamma != nullptr) {
    rc = cfhip_internal_colreduce_f32(partials, 1, D, dgamma, accumulate_param_grads, s);
    if (rc != CFHIP_OK) return rc;
  }
  if (dbeta != nullptr) rc = cfhip_internal_colreduce_f32(partials + D, 1, D, dbeta, accumulate_param_grads, s);
  return rc;
}

// `rows_out` != nullptr: only the row kernel runs; the partial rows stay in the workspace and *rows_out says how many there are
// (cfhip_layernorm_bwd_partials / _reduce: the reduction can then run on another stream, off the input-gradient chain)
static int ln_bwd_impl(const void* dy, const void* x, int x_is_f32, const float* gamma,
                       const float* mean, const float* rstd, const void* dx_add, void* dx,
                       float* dgamma, float* dbeta, int M, int D, int64_t dy_row_stride,
                       int64_t x_row_stride, int64_t dx_row_stride,
                       int accumulate_param_grads, void* workspace,
                       size_t workspace_bytes, void* stream, int* rows_out,
                       const void* dx_add_lo = nullptr, void* dx_lo = nullptr) {
  CFHIP_REQUIRE((dx_add_lo == nullptr || dx_add != nullptr) && (dx_lo == nullptr || dx != nullptr),
                "layernorm_bwd: a second gradient word needs its first (dx_add_lo without dx_add / dx_lo without dx)");
  CFHIP_REQUIRE(((uintptr_t)dx_add_lo & 7) == 0 && ((uintptr_t)dx_lo & 7) == 0, "layernorm_bwd: tensors must be 8-byte aligned");
  const bool lo = dx_lo != nullptr || dx_add_lo != nullptr;
  CFHIP_REQUIRE(dy && x && gamma && mean && rstd, "layernorm_bwd: null pointer");
  const bool do_dx = dx != nullptr, do_pg = dgamma != nullptr || dbeta != nullptr;
  CFHIP_REQUIRE(do_dx || do_pg, "layernorm_bwd: nothing to compute (dx, dgamma and dbeta are all NULL)");
  CFHIP_REQUIRE(M > 0 && D > 0, "layernorm_bwd: empty problem");
  CFHIP_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm_bwd: D=%d must be a multiple of 4 and <= 2048", D);
  CFHIP_REQUIRE(dy_row_stride % 4 == 0 && x_row_stride % 4 == 0 && dx_row_stride % 4 == 0,
                "layernorm_bwd: row strides must be multiples of 4");
  CFHIP_REQUIRE(((uintptr_t)dy & 7) == 0 && ((uintptr_t)x & (x_is_f32 ? 15 : 7)) == 0 && ((uintptr_t)dx & 7) == 0 &&
                    ((uintptr_t)dx_add & 7) == 0,
                "layernorm_bwd: tensors must be 8-byte aligned");
  const size_t need = cfhip_layernorm_bwd_workspace(M, D);
  if (do_pg && (workspace == nullptr || workspace_bytes < need)) {
    cfhip_set_error("layernorm_bwd: needs %zu workspace bytes, got %zu", need, workspace_bytes);
    return CFHIP_ERR_WORKSPACE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nch = (D + 255) / 256;
  const bool exact = (D % 256) == 0;
  float* partials = reinterpret_cast<float*>(workspace);
  // one launch for both outputs: D a multiple of 8 up to 1280 (round 5: not only multiples of 256), 16-byte aligned rows
  const bool fused_ok = g_ln_fused && (exact || g_ln_fused == 1) && do_dx && do_pg && D % 8 == 0 && nch <= 5 && dy_row_stride % 8 == 0 && dx_row_stride % 8 == 0 &&
                        x_row_stride % (x_is_f32 ? 4 : 8) == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0 &&
                        ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dx_add & 15) == 0 && ((uintptr_t)dx_lo & 15) == 0 &&
                        ((uintptr_t)dx_add_lo & 15) == 0 && (!lo || dx_lo != nullptr);
  if (fused_ok) {
    const int blocks2 = ln2_grid(M);
    const size_t lds2 = (size_t)(1 + 2 * LN2_WAVES) * nch * 256 * sizeof(float);
#define LN_BWD2_T(G_, XT_, EX_)                                                                                      \
  do {                                                                                                               \
    if (lo)                                                                                                          \
      hipLaunchKernelGGL((layernorm_bwd_fused_kernel<G_, XT_, EX_, true>), dim3(blocks2), dim3(LN2_THREADS), lds2, s, \
                         (const bf16_t*)dy, (const XT_*)x, gamma, mean, rstd, (const bf16_t*)dx_add, (bf16_t*)dx,    \
                         partials, M, (long)dy_row_stride, (long)x_row_stride, (long)dx_row_stride, D,               \
                         (const bf16_t*)dx_add_lo, (bf16_t*)dx_lo);                                                  \
    else                                                                                                             \
      hipLaunchKernelGGL((layernorm_bwd_fused_kernel<G_, XT_, EX_>), dim3(blocks2), dim3(LN2_THREADS), lds2, s,      \
                         (const bf16_t*)dy, (const XT_*)x, gamma, mean, rstd, (const bf16_t*)dx_add, (bf16_t*)dx,    \
                         partials, M, (long)dy_row_stride, (long)x_row_stride, (long)dx_row_stride, D,               \
                         (const bf16_t*)nullptr, (bf16_t*)nullptr);                                                  \
  } while (0)
#define LN_BWD2(G_)                                                                                                  \
  do {                                                                                                               \
    if (x_is_f32) {                                                                                                  \
      if (exact) LN_BWD2_T(G_, float, true); else LN_BWD2_T(G_, float, false);                                       \
    } else {                                                                                                         \
      if (exact) LN_BWD2_T(G_, bf16_t, true); else LN_BWD2_T(G_, bf16_t, false);                                     \
    }                                                                                                                \
  } while (0)
    switch (nch) {
      case 1: LN_BWD2(1); break;
      case 2: LN_BWD2(2); break;
      case 3: LN_BWD2(3); break;
      case 4: LN_BWD2(4); break;
      default: LN_BWD2(5); break;
    }
#undef LN_BWD2
#undef LN_BWD2_T
    CFHIP_CHECK_LAUNCH("layernorm_bwd_fused");
    if (rows_out != nullptr) {
      *rows_out = blocks2;
      return CFHIP_OK;
    }
    return ln_bwd_reduce(partials, blocks2, D, dgamma, dbeta, accumulate_param_grads, s);
  }
  // rows wider than 1280: the kernel that produces dx AND the parameter partials keeps the whole row twice in registers and
  // spilled 300-376 bytes per lane at D = 2048 — two launches (dx, then the partials) fit
  if (do_dx && do_pg && nch >= 5) {
    int rc = ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, dx_add, dx, nullptr, nullptr, M, D, dy_row_stride, x_row_stride,
                         dx_row_stride, 0, nullptr, 0, stream, nullptr, dx_add_lo, dx_lo);
    if (rc != CFHIP_OK) return rc;
    return ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, nullptr, nullptr, dgamma, dbeta, M, D, dy_row_stride, x_row_stride,
                       dx_row_stride, accumulate_param_grads, workspace, workspace_bytes, stream, rows_out);
  }
  const int blocks = ln_grid(M, do_pg ? 512 : 1024);  // dx-only: fewer registers, twice the waves
  const size_t lds = do_pg ? (size_t)2 * D * sizeof(float) : 0;
#define LN_BWD_ONE(N_, EX_, XT_, DX_, PG_)                                                               \
  hipLaunchKernelGGL((layernorm_bwd_kernel<N_, EX_, XT_, DX_, PG_>), dim3(blocks), dim3(LN_THREADS), lds, \
                     s, (const bf16_t*)dy, (const XT_*)x, gamma, mean, rstd, (const bf16_t*)dx_add,       \
                     (bf16_t*)dx, partials, M, D, (long)dy_row_stride, (long)x_row_stride,                \
                     (long)dx_row_stride, (const bf16_t*)dx_add_lo, (bf16_t*)dx_lo)
#define LN_BWD_MODE(N_, EX_, XT_)                                     \
  do {                                                                \
    if (do_dx && do_pg) {                                              \
      if constexpr (N_ < 6) LN_BWD_ONE(N_, EX_, XT_, true, true);      \
    } else if (do_dx) LN_BWD_ONE(N_, EX_, XT_, true, false);           \
    else LN_BWD_ONE(N_, EX_, XT_, false, true);                        \
  } while (0)
#define LN_BWD(N_, EX_)                        \
  do {                                         \
    if (x_is_f32) LN_BWD_MODE(N_, EX_, float); \
    else LN_BWD_MODE(N_, EX_, bf16_t);         \
  } while (0)
  LN_DISPATCH(LN_BWD, nch, 0)
#undef LN_BWD
#undef LN_BWD_MODE
#undef LN_BWD_ONE
  CFHIP_CHECK_LAUNCH("layernorm_bwd");
  if (rows_out != nullptr) {
    *rows_out = do_pg ? blocks : 0;
    return CFHIP_OK;
  }
  if (do_pg) return ln_bwd_reduce(partials, blocks, D, dgamma, dbeta, accumulate_param_grads, s);  // partial rows: dgamma | dbeta
  return CFHIP_OK;
}

extern "C" int cfhip_layernorm_bwd(const void* dy, const void* x, int x_is_f32, const float* gamma,
                                   const float* mean, const float* rstd, const void* dx_add, void* dx,
                                   float* dgamma, float* dbeta, int M, int D, int64_t dy_row_stride,
                                   int64_t x_row_stride, int64_t dx_row_stride,
                                   int accumulate_param_grads, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  return ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, dx_add, dx, dgamma, dbeta, M, D, dy_row_stride, x_row_stride, dx_row_stride,
                     accumulate_param_grads, workspace, workspace_bytes, stream, nullptr);
}

// The same with the residual-gradient stream in TWO bf16 words (see layernorm_bwd_fused_kernel): dx_add_lo / dx_lo are the second
// words of dx_add / dx (either may be NULL: a one-word incoming gradient, a one-word result), same row strides as the first.
extern "C" int cfhip_layernorm_bwd2(const void* dy, const void* x, int x_is_f32, const float* gamma,
                                    const float* mean, const float* rstd, const void* dx_add, const void* dx_add_lo, void* dx,
                                    void* dx_lo, float* dgamma, float* dbeta, int M, int D, int64_t dy_row_stride,
                                    int64_t x_row_stride, int64_t dx_row_stride, int accumulate_param_grads, void* workspace,
                                    size_t workspace_bytes, int* rows_out, void* stream) {
  if (rows_out != nullptr) {  // the `_partials` form: row kernel only, the partial rows stay in the workspace
    CFHIP_REQUIRE(workspace != nullptr, "layernorm_bwd2: null workspace");
    float* marker = reinterpret_cast<float*>(workspace);
    return ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, dx_add, dx, marker, marker + D, M, D, dy_row_stride, x_row_stride,
                       dx_row_stride, 0, workspace, workspace_bytes, stream, rows_out, dx_add_lo, dx_lo);
  }
  return ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, dx_add, dx, dgamma, dbeta, M, D, dy_row_stride, x_row_stride, dx_row_stride,
                     accumulate_param_grads, workspace, workspace_bytes, stream, nullptr, dx_add_lo, dx_lo);
}

// The same in two calls: `_partials` runs the row kernel (dx, and the per-workgroup partial sums of dgamma / dbeta into
// `workspace`: cfhip_layernorm_bwd_workspace bytes) and reports the number of partial rows; `_reduce` adds them into dgamma /
// dbeta — on whatever stream the caller likes (ordered after `_partials` by the caller): the parameter gradients are needed by
// the optimizer only, the input gradient by the next kernel of the backward chain.
extern "C" int cfhip_layernorm_bwd_partials(const void* dy, const void* x, int x_is_f32, const float* gamma,
                                            const float* mean, const float* rstd, const void* dx_add, void* dx, int M, int D,
                                            int64_t dy_row_stride, int64_t x_row_stride, int64_t dx_row_stride, void* workspace,
                                            size_t workspace_bytes, int* rows_out, void* stream) {
  CFHIP_REQUIRE(rows_out != nullptr && workspace != nullptr, "layernorm_bwd_partials: null pointer");
  float* marker = reinterpret_cast<float*>(workspace);  // (any non-null destination: selects the parameter-gradient kernels)
  return ln_bwd_impl(dy, x, x_is_f32, gamma, mean, rstd, dx_add, dx, marker, marker + D, M, D, dy_row_stride, x_row_stride,
                     dx_row_stride, 0, workspace, workspace_bytes, stream, rows_out);
}

extern "C" int cfhip_layernorm_bwd_reduce(void* workspace, int rows, int D, float* dgamma, float* dbeta, int accumulate, void* stream) {
  CFHIP_REQUIRE(workspace != nullptr && rows > 0 && D > 0 && (dgamma != nullptr || dbeta != nullptr), "layernorm_bwd_reduce: bad arguments");
  return ln_bwd_reduce(reinterpret_cast<float*>(workspace), rows, D, dgamma, dbeta, accumulate, reinterpret_cast<hipStream_t>(stream));
}

// ---- the reference's 4-D `LN` (norms.py:30-46; NormFactory("layer_norm") inside conv blocks) ----------------------------------------
// y = (x - mean_b) / (std_b + eps) * w[c] + b[c] on [B, C, H, W]: ONE mean and one UNBIASED standard deviation per sample over all
// C*H*W elements, eps added to the standard deviation (not inside a square root), per-channel affine.  Not a GroupNorm with one group
// (biased variance, eps under the root) — its own kernels.  Off every benchmarked path (no named configuration uses it): written for
// clarity and a fixed summation order (slice partials in double, block tree reduces), not for the last GB/s.
//   forward:  ln4d_partial (sum, sum of squares per slice) -> ln4d_finish (mean, std per sample) -> ln4d_apply
//   backward: g = dy * w[c];  dx = inv g - inv S1 / n - (x - mean) inv^2 S2 / ((n - 1) std),  S1 = sum g, S2 = sum g (x - mean),
//             inv = 1 / (std + eps);  dw[c] = sum_{b, hw} dy (x - mean_b) inv_b,  db[c] = sum dy   (one workgroup per channel)
namespace {

constexpr int LN4_THREADS = 256;
constexpr int LN4_MAX_SLICES = 64;

__device__ __forceinline__ double block_sum_f64(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
  for (int off = LN4_THREADS / 2; off > 0; off >>= 1) {
    if (t < off) sh[t] += sh[t + off];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// MODE 0: (sum x, sum x^2);  MODE 1: (sum g, sum g (x - mean)) with g = dy * w[c]
template <int MODE>
__global__ __launch_bounds__(LN4_THREADS) void ln4d_partial_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                     const float* __restrict__ w, const float* __restrict__ mean,
                                                                     double* __restrict__ partial, long n, int HW, int slices) {
  __shared__ double sh[LN4_THREADS];
  const int b = blockIdx.y, s = blockIdx.x;
  const long per = (n + slices - 1) / slices;
  const long lo = (long)s * per, hi = min(n, lo + per);
  const bf16_t* xb = x + (long)b * n;
  const bf16_t* gb = MODE == 1 ? dy + (long)b * n : nullptr;
  const float m = MODE == 1 ? mean[b] : 0.f;
  double a0 = 0.0, a1 = 0.0;
  for (long base = lo; base < hi; base += (long)LN4_THREADS * 64) {  // f32 inside a 64-element run, double across runs
    float f0 = 0.f, f1 = 0.f;
    for (int r = 0; r < 64; ++r) {
      const long i = base + (long)r * LN4_THREADS + threadIdx.x;
      if (i >= hi) break;
      const float xv = bf16_to_f32(xb[i]);
      if (MODE == 0) {
        f0 += xv;
        f1 = fmaf(xv, xv, f1);
      } else {
        float g = bf16_to_f32(gb[i]);
        if (w != nullptr) g *= w[(int)(i / HW)];
        f0 += g;
        f1 = fmaf(g, xv - m, f1);
      }
    }
    a0 += (double)f0;
    a1 += (double)f1;
  }
  const double s0 = block_sum_f64(a0, sh), s1 = block_sum_f64(a1, sh);
  if (threadIdx.x == 0) {
    partial[((long)b * slices + s) * 2] = s0;
    partial[((long)b * slices + s) * 2 + 1] = s1;
  }
}

// MODE 0: partials -> mean[b], std[b] (unbiased);  MODE 1: partials -> coef[b] = (inv S1 / n, inv^2 S2 / ((n - 1) std))
template <int MODE>
__global__ void ln4d_finish_kernel(const double* __restrict__ partial, float* __restrict__ out0, float* __restrict__ out1,
                                   const float* __restrict__ stdv, long n, int slices, int B, float eps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s0 = 0.0, s1 = 0.0;
  for (int s = 0; s < slices; ++s) {
    s0 += partial[((long)b * slices + s) * 2];
    s1 += partial[((long)b * slices + s) * 2 + 1];
  }
  if (MODE == 0) {
    const double mean = s0 / (double)n;
    const double var = (s1 - (double)n * mean * mean) / (double)(n - 1);  // torch.std: Bessel's correction (n = 1: NaN, as there)
    out0[b] = (float)mean;
    out1[b] = (float)sqrt(var > 0.0 ? var : (n > 1 ? 0.0 : var));
  } else {
    const double sd = (double)stdv[b], inv = 1.0 / (sd + (double)eps);
    out0[b] = (float)(inv * s0 / (double)n);
    out1[b] = sd > 0.0 ? (float)(inv * inv * s1 / ((double)(n - 1) * sd)) : 0.f;
  }
}

// forward: y = (x - mean) inv w[c] + b[c];  backward (BWD): dx = inv g - c1 - (x - mean) c2
template <bool BWD>
__global__ __launch_bounds__(LN4_THREADS) void ln4d_apply_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                                   const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                                                   bf16_t* __restrict__ out, long n, int HW, long total, float eps) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / n);
    const int c = (int)((i - (long)b * n) / HW);
    const float m = mean[b], inv = 1.0f / (stdv[b] + eps);
    const float xv = bf16_to_f32(x[i]);
    float r;
    if (!BWD) {
      r = (xv - m) * inv;
      if (w != nullptr) r = fmaf(r, w[c], bias[c]);
    } else {
      float g = bf16_to_f32(dy[i]);
      if (w != nullptr) g *= w[c];
      r = fmaf(inv, g, -c1[b]) - (xv - m) * c2[b];
    }
    out[i] = f32_to_bf16(r);
  }
}

// one workgroup per channel: dw[c] (+)= sum_{b, hw} dy xhat, db[c] (+)= sum dy
__global__ __launch_bounds__(LN4_THREADS) void ln4d_param_grad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                        const float* __restrict__ mean, const float* __restrict__ stdv,
                                                                        float* __restrict__ dw, float* __restrict__ db, int B, int C,
                                                                        int HW, float eps, int accumulate) {
  __shared__ double sh[LN4_THREADS];
  const int c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int b = 0; b < B; ++b) {
    const float m = mean[b], inv = 1.0f / (stdv[b] + eps);
    const long base = ((long)b * C + c) * HW;
    float f0 = 0.f, f1 = 0.f;
    for (int i = threadIdx.x; i < HW; i += LN4_THREADS) {
      const float g = bf16_to_f32(dy[base + i]);
      f0 = fmaf(g, (bf16_to_f32(x[base + i]) - m) * inv, f0);
      f1 += g;
    }
    a0 += (double)f0;
    a1 += (double)f1;
  }
  const double s0 = block_sum_f64(a0, sh), s1 = block_sum_f64(a1, sh);
  if (threadIdx.x == 0) {
    dw[c] = accumulate ? dw[c] + (float)s0 : (float)s0;
    db[c] = accumulate ? db[c] + (float)s1 : (float)s1;
  }
}

int ln4d_slices(long n) {
  long s = (n + 16383) / 16384;
  return (int)(s < 1 ? 1 : (s > LN4_MAX_SLICES ? LN4_MAX_SLICES : s));
}

}  // namespace

extern "C" size_t cfhip_layernorm4d_workspace(int B, int C, int HW) {
  const long n = (long)C * HW;
  return (size_t)B * ln4d_slices(n) * 2 * sizeof(double) + (size_t)B * 2 * sizeof(float);
}

extern "C" int cfhip_layernorm4d_fwd(const void* x, const float* weight, const float* bias, void* y, float* mean, float* stdv,
                                     int B, int C, int HW, float eps, void* workspace, size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(x && y && mean && stdv, "layernorm4d_fwd: null operand");
  CFHIP_REQUIRE(B > 0 && C > 0 && HW > 0, "layernorm4d_fwd: bad shape %d x %d x %d", B, C, HW);
  CFHIP_REQUIRE((weight == nullptr) == (bias == nullptr), "layernorm4d_fwd: weight and bias come together");
  CFHIP_REQUIRE(workspace != nullptr && workspace_bytes >= cfhip_layernorm4d_workspace(B, C, HW),
                "layernorm4d_fwd: workspace of %zu bytes needed (cfhip_layernorm4d_workspace)", cfhip_layernorm4d_workspace(B, C, HW));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)C * HW;
  const int slices = ln4d_slices(n);
  double* partial = reinterpret_cast<double*>(workspace);
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  hipLaunchKernelGGL(ln4d_partial_kernel<0>, dim3(slices, B), dim3(LN4_THREADS), 0, s, xb, (const bf16_t*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, partial, n, HW, slices);
  hipLaunchKernelGGL(ln4d_finish_kernel<0>, dim3((B + 63) / 64), dim3(64), 0, s, partial, mean, stdv, (const float*)nullptr, n, slices, B, eps);
  const long total = (long)B * n;
  int blocks = (int)((total + LN4_THREADS * 8L - 1) / (LN4_THREADS * 8L));
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  hipLaunchKernelGGL(ln4d_apply_kernel<false>, dim3(blocks), dim3(LN4_THREADS), 0, s, xb, (const bf16_t*)nullptr, weight, bias, mean, stdv,
                     (const float*)nullptr, (const float*)nullptr, reinterpret_cast<bf16_t*>(y), n, HW, total, eps);
  CFHIP_CHECK_LAUNCH("layernorm4d_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_layernorm4d_bwd(const void* dy, const void* x, const float* weight, const float* mean, const float* stdv, void* dx,
                                     float* dweight, float* dbias, int accumulate_param_grads, int B, int C, int HW, float eps,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(dy && x && mean && stdv, "layernorm4d_bwd: null operand");
  CFHIP_REQUIRE(B > 0 && C > 0 && HW > 0, "layernorm4d_bwd: bad shape %d x %d x %d", B, C, HW);
  CFHIP_REQUIRE((dweight == nullptr) == (dbias == nullptr), "layernorm4d_bwd: dweight and dbias come together");
  CFHIP_REQUIRE(dx != nullptr || dweight != nullptr, "layernorm4d_bwd: nothing asked for");
  CFHIP_REQUIRE(workspace != nullptr && workspace_bytes >= cfhip_layernorm4d_workspace(B, C, HW),
                "layernorm4d_bwd: workspace of %zu bytes needed (cfhip_layernorm4d_workspace)", cfhip_layernorm4d_workspace(B, C, HW));
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long n = (long)C * HW;
  const int slices = ln4d_slices(n);
  double* partial = reinterpret_cast<double*>(workspace);
  float* coef = reinterpret_cast<float*>(partial + (size_t)B * slices * 2);
  const bf16_t* xb = reinterpret_cast<const bf16_t*>(x);
  const bf16_t* gb = reinterpret_cast<const bf16_t*>(dy);
  if (dx != nullptr) {
    hipLaunchKernelGGL(ln4d_partial_kernel<1>, dim3(slices, B), dim3(LN4_THREADS), 0, s, xb, gb, weight, mean, partial, n, HW, slices);
    hipLaunchKernelGGL(ln4d_finish_kernel<1>, dim3((B + 63) / 64), dim3(64), 0, s, partial, coef, coef + B, stdv, n, slices, B, eps);
    const long total = (long)B * n;
    int blocks = (int)((total + LN4_THREADS * 8L - 1) / (LN4_THREADS * 8L));
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(ln4d_apply_kernel<true>, dim3(blocks), dim3(LN4_THREADS), 0, s, xb, gb, weight, (const float*)nullptr, mean, stdv,
                       coef, coef + B, reinterpret_cast<bf16_t*>(dx), n, HW, total, eps);
  }
  if (dweight != nullptr)
    hipLaunchKernelGGL(ln4d_param_grad_kernel, dim3(C), dim3(LN4_THREADS), 0, s, xb, gb, mean, stdv, dweight, dbias, B, C, HW, eps,
                       accumulate_param_grads);
  CFHIP_CHECK_LAUNCH("layernorm4d_bwd");
  return CFHIP_OK;
}
