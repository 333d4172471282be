// Grouped convolution (groups > 1, depthwise included): F.conv2d(..., groups=g) reached from Conv2d.forward (reference
// modules/core/convs/basic.py:160-177) and its two backward halves.  Outside the named benchmark configurations (SURVEY §8a
// lists it under the options): DIRECT kernels — one thread per output / input element, one workgroup per filter plane for the
// weight gradient — fp32 accumulation over bf16 operands, no atomics (deterministic).  A group has Cin / g input channels
// per filter: with the depthwise case (1) or the small groups these layers use there is no GEMM worth an MFMA tile, the
// work is HBM / L2 bound.  NCHW bf16 activations, weights bf16 [Cout][Cin / g][kh][kw], bias and gradients of parameters f32.
#include "common.h"

namespace {

struct GConv {
  const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y;
  const bf16_t* dy; bf16_t* dx; float* dw; float* db;
  int B, Cin, H, W, Cout, kh, kw, stride, pad, dil, groups, Ho, Wo;
  int cgi, cgo;  // channels per group: input, output
  int acc_w, acc_b;
};

__global__ __launch_bounds__(256) void gconv_fwd_kernel(GConv p) {
  const long total = (long)p.B * p.Cout * p.Ho * p.Wo;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int xo = (int)(idx % p.Wo);
    long t = idx / p.Wo;
    const int yo = (int)(t % p.Ho);
    t /= p.Ho;
    const int co = (int)(t % p.Cout), b = (int)(t / p.Cout);
    const int g = co / p.cgo;
    float acc = p.bias != nullptr ? p.bias[co] : 0.f;
    const bf16_t* wr = p.w + (long)co * p.cgi * p.kh * p.kw;
    for (int ci = 0; ci < p.cgi; ++ci) {
      const bf16_t* xp = p.x + ((long)b * p.Cin + g * p.cgi + ci) * p.H * p.W;
      for (int ky = 0; ky < p.kh; ++ky) {
        const int yi = yo * p.stride - p.pad + ky * p.dil;
        if (yi < 0 || yi >= p.H) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
          const int xi = xo * p.stride - p.pad + kx * p.dil;
          if (xi < 0 || xi >= p.W) continue;
          acc = fmaf(bf16_to_f32(xp[(long)yi * p.W + xi]), bf16_to_f32(wr[(ci * p.kh + ky) * p.kw + kx]), acc);
        }
      }
    }
    p.y[idx] = f32_to_bf16(acc);
  }
}

// dx[b][c][y][x] = sum over the filters of c's group and the taps that reach (y, x)
__global__ __launch_bounds__(256) void gconv_bwd_input_kernel(GConv p) {
  const long total = (long)p.B * p.Cin * p.H * p.W;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += (long)gridDim.x * 256L) {
    const int xi = (int)(idx % p.W);
    long t = idx / p.W;
    const int yi = (int)(t % p.H);
    t /= p.H;
    const int c = (int)(t % p.Cin), b = (int)(t / p.Cin);
    const int g = c / p.cgi, ci = c - g * p.cgi;
    float acc = 0.f;
    for (int col = 0; col < p.cgo; ++col) {
      const int co = g * p.cgo + col;
      const bf16_t* dyp = p.dy + ((long)b * p.Cout + co) * p.Ho * p.Wo;
      const bf16_t* wr = p.w + ((long)co * p.cgi + ci) * p.kh * p.kw;
      for (int ky = 0; ky < p.kh; ++ky) {
        const int ny = yi + p.pad - ky * p.dil;
        if (ny < 0 || ny % p.stride != 0) continue;
        const int yo = ny / p.stride;
        if (yo >= p.Ho) continue;
        for (int kx = 0; kx < p.kw; ++kx) {
          const int nx = xi + p.pad - kx * p.dil;
          if (nx < 0 || nx % p.stride != 0) continue;
          const int xo = nx / p.stride;
          if (xo >= p.Wo) continue;
          acc = fmaf(bf16_to_f32(dyp[(long)yo * p.Wo + xo]), bf16_to_f32(wr[ky * p.kw + kx]), acc);
        }
      }
    }
    p.dx[idx] = f32_to_bf16(acc);
  }
}

// one workgroup per filter plane (co, ci): its kh * kw taps (<= 49) and, for ci == 0, the bias gradient of co;
// threads stride over (b, yo, xo); fold by xor shuffles, then the four waves through LDS in wave order
constexpr int TAPS_MAX = 49;
template <int TAPS>
__global__ __launch_bounds__(256) void gconv_bwd_weight_kernel(GConv p) {
  __shared__ float red[4][TAPS + 1];
  const int ci = blockIdx.x % p.cgi, co = blockIdx.x / p.cgi;
  const int g = co / p.cgo;
  const int taps = p.kh * p.kw;
  float acc[TAPS + 1];
#pragma unroll
  for (int e = 0; e <= TAPS; ++e) acc[e] = 0.f;
  const long n = (long)p.B * p.Ho * p.Wo;
  for (long idx = threadIdx.x; idx < n; idx += 256) {
    const int xo = (int)(idx % p.Wo);
    long t = idx / p.Wo;
    const int yo = (int)(t % p.Ho), b = (int)(t / p.Ho);
    const float d = bf16_to_f32(p.dy[(((long)b * p.Cout + co) * p.Ho + yo) * p.Wo + xo]);
    acc[TAPS] += d;
    const bf16_t* xp = p.x + ((long)b * p.Cin + g * p.cgi + ci) * p.H * p.W;
#pragma unroll
    for (int e = 0; e < TAPS; ++e) {
      if (e < taps) {
        const int ky = e / p.kw, kx = e - ky * p.kw;
        const int yi = yo * p.stride - p.pad + ky * p.dil, xi = xo * p.stride - p.pad + kx * p.dil;
        if (yi >= 0 && yi < p.H && xi >= 0 && xi < p.W) acc[e] = fmaf(d, bf16_to_f32(xp[(long)yi * p.W + xi]), acc[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e <= TAPS; ++e) {
    float v = acc[e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = v;
  }
  __syncthreads();
  const int e = threadIdx.x;
  if (e < taps) {
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    float* out = p.dw + ((long)co * p.cgi + ci) * taps + e;
    *out = p.acc_w ? *out + v : v;
  }
  if (e == TAPS && ci == 0 && p.db != nullptr) {
    const float v = (red[0][TAPS] + red[1][TAPS]) + (red[2][TAPS] + red[3][TAPS]);
    p.db[co] = p.acc_b ? p.db[co] + v : v;
  }
}

int fill(GConv& p, int B, int Cin, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil, int groups) {
  CFHIP_REQUIRE(B > 0 && Cin > 0 && H > 0 && W > 0 && Cout > 0 && kh > 0 && kw > 0, "conv2d_grouped: empty problem");
  CFHIP_REQUIRE(groups >= 1 && Cin % groups == 0 && Cout % groups == 0, "conv2d_grouped: groups %d must divide Cin %d and Cout %d", groups, Cin, Cout);
  CFHIP_REQUIRE(stride >= 1 && dil >= 1 && pad >= 0, "conv2d_grouped: stride %d dilation %d padding %d", stride, dil, pad);
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.kh = kh; p.kw = kw;
  p.stride = stride; p.pad = pad; p.dil = dil; p.groups = groups;
  p.Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1;
  p.Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  CFHIP_REQUIRE(p.Ho > 0 && p.Wo > 0, "conv2d_grouped: the kernel does not fit the padded input");
  p.cgi = Cin / groups; p.cgo = Cout / groups;
  return CFHIP_OK;
}

unsigned blocks_for(long total) { return (unsigned)((total + 255) / 256 > 65535L * 16 ? 65535L * 16 : (total + 255) / 256); }

}  // namespace

extern "C" int cfhip_conv2d_grouped_fwd(const void* x, const void* w, const float* bias, void* y, int B, int Cin, int H, int W,
                                        int Cout, int kh, int kw, int stride, int pad, int dil, int groups, void* stream) {
  GConv p = {};
  const int rc = fill(p, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, groups);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(x && w && y, "conv2d_grouped_fwd: null operand");
  p.x = reinterpret_cast<const bf16_t*>(x); p.w = reinterpret_cast<const bf16_t*>(w); p.bias = bias; p.y = reinterpret_cast<bf16_t*>(y);
  hipLaunchKernelGGL(gconv_fwd_kernel, dim3(blocks_for((long)B * Cout * p.Ho * p.Wo)), dim3(256), 0, (hipStream_t)stream, p);
  CFHIP_CHECK_LAUNCH("conv2d_grouped_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_conv2d_grouped_bwd_input(const void* dy, const void* w, void* dx, int B, int Cin, int H, int W, int Cout, int kh,
                                              int kw, int stride, int pad, int dil, int groups, void* stream) {
  GConv p = {};
  const int rc = fill(p, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, groups);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(dy && w && dx, "conv2d_grouped_bwd_input: null operand");
  p.dy = reinterpret_cast<const bf16_t*>(dy); p.w = reinterpret_cast<const bf16_t*>(w); p.dx = reinterpret_cast<bf16_t*>(dx);
  hipLaunchKernelGGL(gconv_bwd_input_kernel, dim3(blocks_for((long)B * Cin * H * W)), dim3(256), 0, (hipStream_t)stream, p);
  CFHIP_CHECK_LAUNCH("conv2d_grouped_bwd_input");
  return CFHIP_OK;
}

extern "C" int cfhip_conv2d_grouped_bwd_weight(const void* dy, const void* x, float* dw, int accumulate, float* bias_grad,
                                               int bias_grad_accumulate, int B, int Cin, int H, int W, int Cout, int kh, int kw,
                                               int stride, int pad, int dil, int groups, void* stream) {
  GConv p = {};
  const int rc = fill(p, B, Cin, H, W, Cout, kh, kw, stride, pad, dil, groups);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(dy && x && dw, "conv2d_grouped_bwd_weight: null operand");
  CFHIP_REQUIRE(kh * kw <= TAPS_MAX, "conv2d_grouped_bwd_weight: %d x %d taps (at most %d)", kh, kw, TAPS_MAX);
  p.dy = reinterpret_cast<const bf16_t*>(dy); p.x = reinterpret_cast<const bf16_t*>(x);
  p.dw = dw; p.db = bias_grad; p.acc_w = accumulate; p.acc_b = bias_grad_accumulate;
  const dim3 grid((unsigned)(Cout * p.cgi));
  hipStream_t s = (hipStream_t)stream;
  if (kh * kw <= 9) hipLaunchKernelGGL(gconv_bwd_weight_kernel<9>, grid, dim3(256), 0, s, p);
  else if (kh * kw <= 25) hipLaunchKernelGGL(gconv_bwd_weight_kernel<25>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gconv_bwd_weight_kernel<TAPS_MAX>, grid, dim3(256), 0, s, p);
  CFHIP_CHECK_LAUNCH("conv2d_grouped_bwd_weight");
  return CFHIP_OK;
}
