// Attention with RETURNED weights: the slow path of the reference's `Attention.forward(require_weights=True)` /
// `customize_sdp` (modules/core/attentions.py:256-268: raw = q k^T / scaling, masked_fill(-inf), softmax, weights @ v).
//
// The output still comes from the fused kernels of attn.hip (same numbers as the fast path); these kernels add what the fused
// path never materialises — the probability matrix — from the log-sum-exp the fused forward saved:
//     P[b, h, i, j] = keep(b, h, i, j) ? exp(scale * q_i . k_j - lse[b, h, i]) : 0           (cfhip_attn_probs)
// and, when a gradient flows into the returned weights (dP), its contribution to dq / dk through the softmax:
//     t_i = sum_j dP_ij P_ij,   dS_ij = scale * P_ij (dP_ij - t_i),   dq_i = sum_j dS_ij k_j,   dk_j = sum_i dS_ij q_i
// (cfhip_attn_probs_bwd; dv gets nothing: the weights do not depend on v).  Plain VALU arithmetic in fp32 on the bf16 q / k —
// B H Tq Tk head_dim multiply-adds, no MFMA: this path exists for inspection / hooks / auxiliary losses on the weights, not
// for throughput.  Deterministic (no atomics).  Addressing as the fused kernels: ptr[b * stride_b + t * stride_t + h * head_dim + d].
#include "common.h"

namespace {

constexpr int ROWS = 16;    // query (or key) rows per workgroup
constexpr int NT = 256;     // threads per workgroup
constexpr int DH_MAX = 192;

struct ProbParams {
  const bf16_t* q; const bf16_t* k;
  const float* lse;
  const uint8_t* mask;
  float* P;          // [B][H][Tq][Tk]
  const float* dP;   // [B][H][Tq][Tk]
  float* dS;         // [B][H][Tq][Tk] workspace
  bf16_t* dq; bf16_t* dk;  // [B][T][H*dh] contiguous
  int B, H, Tq, Tk, dh;
  long q_sb, q_st, k_sb, k_st;
  long ms_b, ms_h, ms_q;
  float scale;
  int causal;
};

__device__ __forceinline__ bool keep_at(const ProbParams& p, int b, int h, int i, int j) {
  if (p.causal && j > i) return false;
  if (p.mask != nullptr && p.mask[(long)b * p.ms_b + (long)h * p.ms_h + (long)i * p.ms_q + j] == 0) return false;
  return true;
}

// rows i0 .. i0+15 of one (b, h): each thread owns keys j = tid, tid + 256, ...; the q rows sit in LDS as f32
// MODE 0: write P.  MODE 1: t_i = sum_j dP_ij P_ij first (second sweep over the keys), then write dS.
template <int MODE>
__global__ __launch_bounds__(NT) void attn_probs_kernel(ProbParams p) {
  __shared__ float qs[ROWS][DH_MAX];
  __shared__ float red[ROWS][NT / 64];
  __shared__ float tsum[ROWS];
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int i0 = blockIdx.x * ROWS;
  const int tid = threadIdx.x;
  for (int e = tid; e < ROWS * p.dh; e += NT) {
    const int r = e / p.dh, d = e - r * p.dh;
    const int i = i0 + r;
    qs[r][d] = i < p.Tq ? bf16_to_f32(p.q[(long)b * p.q_sb + (long)i * p.q_st + (long)h * p.dh + d]) : 0.f;
  }
  __syncthreads();
  float lse[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) lse[r] = i0 + r < p.Tq ? p.lse[((long)b * p.H + h) * p.Tq + i0 + r] : 0.f;
  const long row0 = (((long)b * p.H + h) * p.Tq + i0) * p.Tk;

  auto sweep = [&](auto&& use) {
    for (int j = tid; j < p.Tk; j += NT) {
      float s[ROWS];
#pragma unroll
      for (int r = 0; r < ROWS; ++r) s[r] = 0.f;
      const bf16_t* kr = p.k + (long)b * p.k_sb + (long)j * p.k_st + (long)h * p.dh;
      for (int d0 = 0; d0 < p.dh; d0 += 8) {
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(kr + d0);
        float kf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) kf[e] = bf16_to_f32((bf16_t)kv[e]);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          float a = s[r];
#pragma unroll
          for (int e = 0; e < 8; ++e) a = fmaf(qs[r][d0 + e], kf[e], a);
          s[r] = a;
        }
      }
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        const int i = i0 + r;
        if (i >= p.Tq) continue;
        const float pr = keep_at(p, b, h, i, j) ? __expf(s[r] * p.scale - lse[r]) : 0.f;
        use(r, j, pr);
      }
    }
  };

  if constexpr (MODE == 0) {
    sweep([&](int r, int j, float pr) { p.P[row0 + (long)r * p.Tk + j] = pr; });
  } else {
    float t[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) t[r] = 0.f;
    sweep([&](int r, int j, float pr) { t[r] = fmaf(p.dP[row0 + (long)r * p.Tk + j], pr, t[r]); });
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {  // fixed-order fold: lanes by xor shuffles, the four waves through LDS in wave order
      float v = t[r];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
      if ((tid & 63) == 0) red[r][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < ROWS) tsum[tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
    __syncthreads();
    sweep([&](int r, int j, float pr) {
      const long at = row0 + (long)r * p.Tk + j;
      p.dS[at] = p.scale * pr * (p.dP[at] - tsum[r]);
    });
  }
}

// out[x0 .. x0+15][:] = sum_y dS(x, y) * other[y][:]   —  FOR_Q: x = query i, y = key j (dq = dS k); else x = key, y = query
// (dk = dS^T q).  Thread (r, c) of the 16 x 16 arrangement owns row r and the head_dim columns c, c + 16, ...
template <bool FOR_Q>
__global__ __launch_bounds__(NT) void attn_probs_grad_kernel(ProbParams p) {
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int x0 = blockIdx.x * ROWS;
  const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
  const int x = x0 + r;
  const int nx = FOR_Q ? p.Tq : p.Tk, ny = FOR_Q ? p.Tk : p.Tq;
  const bf16_t* other = FOR_Q ? p.k : p.q;
  const long o_sb = FOR_Q ? p.k_sb : p.q_sb, o_st = FOR_Q ? p.k_st : p.q_st;
  constexpr int NC = DH_MAX / 16;
  float acc[NC];
#pragma unroll
  for (int e = 0; e < NC; ++e) acc[e] = 0.f;
  const long base = ((long)b * p.H + h) * p.Tq * (long)p.Tk;
  if (x < nx) {
    for (int y = 0; y < ny; ++y) {
      const float ds = FOR_Q ? p.dS[base + (long)x * p.Tk + y] : p.dS[base + (long)y * p.Tk + x];
      const bf16_t* orow = other + (long)b * o_sb + (long)y * o_st + (long)h * p.dh;
#pragma unroll
      for (int e = 0; e < NC; ++e) {
        const int d = c + 16 * e;
        if (d < p.dh) acc[e] = fmaf(ds, bf16_to_f32(orow[d]), acc[e]);
      }
    }
    bf16_t* out = (FOR_Q ? p.dq : p.dk) + ((long)b * nx + x) * ((long)p.H * p.dh) + (long)h * p.dh;
#pragma unroll
    for (int e = 0; e < NC; ++e) {
      const int d = c + 16 * e;
      if (d < p.dh) out[d] = f32_to_bf16(acc[e]);
    }
  }
}

int fill(ProbParams& p, const void* q, const void* k, const float* lse, const uint8_t* mask, int B, int H, int Tq, int Tk, int head_dim,
         int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal) {
  CFHIP_REQUIRE(q && k && lse, "attn_probs: null operand");
  CFHIP_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_probs: empty problem B=%d H=%d Tq=%d Tk=%d", B, H, Tq, Tk);
  CFHIP_REQUIRE(head_dim % 8 == 0 && head_dim >= 8 && head_dim <= DH_MAX, "attn_probs: head_dim %d (a multiple of 8 up to %d)", head_dim, DH_MAX);
  CFHIP_REQUIRE(((uintptr_t)k & 15u) == 0 && k_sb % 8 == 0 && k_st % 8 == 0, "attn_probs: k rows must be 16-byte aligned");
  CFHIP_REQUIRE((long)B * H < 65536, "attn_probs: B * H = %ld exceeds the grid", (long)B * H);
  p.q = reinterpret_cast<const bf16_t*>(q);
  p.k = reinterpret_cast<const bf16_t*>(k);
  p.lse = lse; p.mask = mask;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk; p.dh = head_dim;
  p.q_sb = q_sb; p.q_st = q_st; p.k_sb = k_sb; p.k_st = k_st;
  p.ms_b = ms_b; p.ms_h = ms_h; p.ms_q = ms_q;
  p.scale = scale; p.causal = causal;
  return CFHIP_OK;
}

}  // namespace

extern "C" int cfhip_attn_probs(const void* q, const void* k, const float* lse, const uint8_t* mask, float* probs, int B, int H,
                                int Tq, int Tk, int head_dim, int64_t q_stride_b, int64_t q_stride_t, int64_t k_stride_b,
                                int64_t k_stride_t, int64_t ms_b, int64_t ms_h, int64_t ms_q, float scale, int causal, void* stream) {
  ProbParams p = {};
  const int rc = fill(p, q, k, lse, mask, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t, k_stride_b, k_stride_t, ms_b, ms_h, ms_q, scale, causal);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(probs != nullptr, "attn_probs: null output");
  p.P = probs;
  hipLaunchKernelGGL(attn_probs_kernel<0>, dim3((Tq + ROWS - 1) / ROWS, B * H), dim3(NT), 0, (hipStream_t)stream, p);
  CFHIP_CHECK_LAUNCH("attn_probs");
  return CFHIP_OK;
}

extern "C" int cfhip_attn_probs_bwd(const void* q, const void* k, const float* lse, const uint8_t* mask, const float* d_probs,
                                    float* ds_workspace, void* dq, void* dk, int B, int H, int Tq, int Tk, int head_dim,
                                    int64_t q_stride_b, int64_t q_stride_t, int64_t k_stride_b, int64_t k_stride_t, int64_t ms_b,
                                    int64_t ms_h, int64_t ms_q, float scale, int causal, void* stream) {
  ProbParams p = {};
  const int rc = fill(p, q, k, lse, mask, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t, k_stride_b, k_stride_t, ms_b, ms_h, ms_q, scale, causal);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(d_probs && ds_workspace && dq && dk, "attn_probs_bwd: null operand");
  p.dP = d_probs; p.dS = ds_workspace;
  p.dq = reinterpret_cast<bf16_t*>(dq); p.dk = reinterpret_cast<bf16_t*>(dk);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_probs_kernel<1>, dim3((Tq + ROWS - 1) / ROWS, B * H), dim3(NT), 0, s, p);
  hipLaunchKernelGGL(attn_probs_grad_kernel<true>, dim3((Tq + ROWS - 1) / ROWS, B * H), dim3(NT), 0, s, p);
  hipLaunchKernelGGL(attn_probs_grad_kernel<false>, dim3((Tk + ROWS - 1) / ROWS, B * H), dim3(NT), 0, s, p);
  CFHIP_CHECK_LAUNCH("attn_probs_bwd");
  return CFHIP_OK;
}
