// K1/K2: bf16 MFMA GEMM for gfx950 with fused epilogues.
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(n,k) ),  fp32 accumulation on v_mfma_f32_16x16x32_bf16.
//
// Replaces F.linear (reference modules/core/customs.py:89, attentions.py:214) and its autograd
// backward (dX = dY W, dW = dY^T X).  All three operand layouts run on the same kernel:
//   * a "k-major" operand (k contiguous, [rows][K])  is staged as a [128][64] LDS tile and read with
//     ds_read_b128 (XOR-swizzled 16-B slots, conflict-free for the 16-lane read groups);
//   * an "m-major" operand (rows contiguous, [K][rows]) is staged as a [64][128] LDS tile and read
//     with ds_read_b64_tr_b16 (hardware transpose), so no transposed copy of activations or
//     weights is ever materialised in HBM.
// Staging is LDS-DMA (buffer_load_dwordx4 ... lds): the LDS image is lane-linear, so the swizzle
// is applied to the per-lane SOURCE address and again on the read (both-sides-or-neither rule).
// Out-of-range rows / K tails are zero-filled by the buffer descriptor's range check.
//
// Tile 128x128x64, 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles, double-buffered LDS (64 KiB,
// 2 workgroups / CU).  The MFMA is issued with swapped operands (B-fragment first) so each lane
// ends up with 4 CONSECUTIVE output columns of one row: 8-byte bf16 / 16-byte f32 stores.
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous range of tiles.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BN = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile, either orientation
constexpr unsigned OOB = 0x80000000u;     // any offset >= num_records reads as zero

struct GemmParams {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;
  const bf16_t* aux_in;
  bf16_t* aux_out;
  int M, N, K;
  long lda, ldb, ldc;
  int epilogue, out_f32, accumulate;
  int k_chunk;   // K range per z-slice (multiple of BK), == K rounded up when no split
  float* slabs;  // split-K partials [z][M][N] or nullptr
  int tiles_m, tiles_n;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
  if (bytes < 0) bytes = 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Per-lane staging plan for one operand tile: 4 LDS-DMA instructions per wave per K-step.
struct StagePlan {
  unsigned voff[4];  // byte offset from the tile base at k-step 0 (OOB when statically invalid)
  unsigned kpos[4];  // k-major only: k index (elements) of this lane's 16-B chunk inside a K-step
};

template <bool TRANS>
__device__ __forceinline__ StagePlan make_plan(int wave, int lane, long ld, int extent_valid) {
  StagePlan p;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (!TRANS) {
      // tile [128 rows][64 k]: one instruction = 8 rows x 128 B
      const int row = (wave * 4 + j) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      p.kpos[j] = chunk * 8;
      p.voff[j] = (row < extent_valid) ? (unsigned)(row * ld * 2 + chunk * 16) : OOB;
    } else {
      // tile [64 k][128 cols]: one instruction = 4 k-rows x 256 B
      const int krow = (wave * 4 + j) * 4 + (lane >> 4);
      const int s = lane & 15;
      const int key = (krow & 3) | (((krow >> 3) & 1) << 2);
      const int col = (((s >> 1) ^ key) << 4) + ((s & 1) << 3);
      p.kpos[j] = krow;
      p.voff[j] = (col < extent_valid) ? (unsigned)(krow * ld * 2 + col * 2) : OOB;
    }
  }
  return p;
}

template <bool TRANS>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave,
                                           const StagePlan& p, long ld, int k0, int klen) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned off;
    if (!TRANS) {
      off = p.voff[j] + (unsigned)(k0 * 2);
      off = ((int)(k0 + p.kpos[j]) < klen && p.voff[j] != OOB) ? off : OOB;
    } else {
      off = p.voff[j] + (unsigned)((long)k0 * ld * 2);
      off = ((int)(k0 + p.kpos[j]) < klen && p.voff[j] != OOB) ? off : OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (wave * 4 + j) * 1024), 16, off,
                                             0, 0, 0);
  }
}

// Fragment of a k-major tile: rows r0..r0+15, k-substep ks (32 deep).  lane (i = l&15, g = l>>4)
// gets the 8 bf16 at [r0 + i][ks*32 + g*8 ..].
__device__ __forceinline__ bf16x8 frag_kmajor(const char* tile, int r0, int ks, int i, int g) {
  const int row = r0 + i;
  const int slot = (ks * 4 + g) ^ ((row >> 1) & 7);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + slot * 16);
}

// Fragment of an m-major tile ([64 k][128 cols]): columns c0..c0+15, k-substep ks.  Two hardware
// transpose reads; within a 16-lane group, lane s supplies the address of k-row (s>>2), columns
// 4*(s&3).. and receives column (s) of the 4 rows.
__device__ __forceinline__ bf16x8 frag_mmajor(const char* tile, int c0, int ks, int lane) {
  const int g = lane >> 4;
  const int j = (lane & 15) >> 2;
  const int q = lane & 3;
  const int krow = ks * 32 + g * 8 + j;
  const int key = (krow & 3) | (((krow >> 3) & 1) << 2);
  const char* p = tile + krow * 256 + ((((c0 >> 4) ^ key)) << 5) + q * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 4 * 256));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <bool AT, bool BT>
__device__ __forceinline__ void compute_tile(const char* a_tile, const char* b_tile, int wm, int wn,
                                             int lane, f32x4 (&acc)[4][4]) {
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      af[t] = AT ? frag_mmajor(a_tile, wm * 64 + t * 16, ks, lane)
                 : frag_kmajor(a_tile, wm * 64 + t * 16, ks, i, g);
      bfr[t] = BT ? frag_mmajor(b_tile, wn * 64 + t * 16, ks, lane)
                  : frag_kmajor(b_tile, wn * 64 + t * 16, ks, i, g);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        // swapped operands: D[n][m] -> lane holds row m = l&15, cols n = 4*(l>>4) + 0..3
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ni], af[mi], acc[mi][ni], 0, 0, 0);
  }
}

// EPI is a template parameter so that each instantiation carries exactly one epilogue (the erf
// code is large; a runtime switch multiplied it by the 16 unrolled output tiles).
template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];

  // XCD-aware, bijective workgroup remap: XCD x (= bid % 8) owns a contiguous tile range.
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int wg = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int tile_m = wg / p.tiles_n;
  const int tile_n = wg - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = blockIdx.y;
  const int kb = z * p.k_chunk;
  const int klen = min(p.K - kb, p.k_chunk);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // buffer descriptors relative to this tile's origin (small 32-bit offsets, range-checked)
  const int rows_a = p.M - m0, rows_b = p.N - n0;
  const bf16_t* a_base = AT ? p.A + (long)kb * p.lda + m0 : p.A + (long)m0 * p.lda + kb;
  const bf16_t* b_base = BT ? p.B + (long)kb * p.ldb + n0 : p.B + (long)n0 * p.ldb + kb;
  const long a_bytes = AT ? ((long)(klen - 1) * p.lda + rows_a) * 2 : ((long)(rows_a - 1) * p.lda + klen) * 2;
  const long b_bytes = BT ? ((long)(klen - 1) * p.ldb + rows_b) * 2 : ((long)(rows_b - 1) * p.ldb + klen) * 2;
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(a_base, a_bytes);
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(b_base, b_bytes);
  const StagePlan pa = make_plan<AT>(wave, lane, p.lda, rows_a);
  const StagePlan pb = make_plan<BT>(wave, lane, p.ldb, rows_b);

  char* a0 = smem;
  char* b0 = smem + TILE_BYTES;
  char* a1 = smem + 2 * TILE_BYTES;
  char* b1 = smem + 3 * TILE_BYTES;

  f32x4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (klen + BK - 1) / BK;
  if (nk > 0) {
    stage_tile<AT>(a_rsrc, a0, wave, pa, p.lda, 0, klen);
    stage_tile<BT>(b_rsrc, b0, wave, pb, p.ldb, 0, klen);
  }
  __syncthreads();
  for (int t = 0; t < nk; t += 2) {
    if (t + 1 < nk) {
      stage_tile<AT>(a_rsrc, a1, wave, pa, p.lda, (t + 1) * BK, klen);
      stage_tile<BT>(b_rsrc, b1, wave, pb, p.ldb, (t + 1) * BK, klen);
    }
    compute_tile<AT, BT>(a0, b0, wm, wn, lane, acc);
    __syncthreads();
    if (t + 1 < nk) {
      if (t + 2 < nk) {
        stage_tile<AT>(a_rsrc, a0, wave, pa, p.lda, (t + 2) * BK, klen);
        stage_tile<BT>(b_rsrc, b0, wave, pb, p.ldb, (t + 2) * BK, klen);
      }
      compute_tile<AT, BT>(a1, b1, wm, wn, lane, acc);
      __syncthreads();
    }
  }

  // ---- epilogue: lane owns row (l&15), 4 consecutive columns 4*(l>>4).. of each 16x16 tile ----
  const int i = lane & 15, g = lane >> 4;
  const bool to_slab = p.slabs != nullptr;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int row = m0 + wm * 64 + mi * 16 + i;
    if (row >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = n0 + wn * 64 + ni * 16 + g * 4;
      if (col >= p.N) continue;  // N % 4 == 0 is required by the fast path
      f32x4 v = acc[mi][ni];
      if (to_slab) {
        float* dst = p.slabs + ((long)z * p.M + row) * p.N + col;
        *reinterpret_cast<f32x4*>(dst) = v;
        continue;
      }
      if (p.bias != nullptr) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col);
        v += bv;
      }
      const long off = (long)row * p.ldc + col;
      if (EPI == CFHIP_EPI_GELU) {
        const unsigned w0 = pack_bf16x2(v[0], v[1]), w1 = pack_bf16x2(v[2], v[3]);
        if (p.aux_out != nullptr) *reinterpret_cast<u32x2*>(p.aux_out + off) = u32x2{w0, w1};
        // GELU of the bf16-rounded pre-activation (what the saved tensor holds for backward)
        v[0] = gelu_erf_f(bf16lo(w0)); v[1] = gelu_erf_f(bf16hi(w0));
        v[2] = gelu_erf_f(bf16lo(w1)); v[3] = gelu_erf_f(bf16hi(w1));
      } else if (EPI == CFHIP_EPI_RESIDUAL) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p.aux_in + off);
        v[0] += bf16lo(r[0]); v[1] += bf16hi(r[0]); v[2] += bf16lo(r[1]); v[3] += bf16hi(r[1]);
      } else if (EPI == CFHIP_EPI_DGELU) {
        const u32x2 r = *reinterpret_cast<const u32x2*>(p.aux_in + off);
        v[0] *= gelu_erf_grad_f(bf16lo(r[0])); v[1] *= gelu_erf_grad_f(bf16hi(r[0]));
        v[2] *= gelu_erf_grad_f(bf16lo(r[1])); v[3] *= gelu_erf_grad_f(bf16hi(r[1]));
      }
      if (p.out_f32) {
        float* dst = reinterpret_cast<float*>(p.C) + off;
        if (p.accumulate) v += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        bf16_t* dst = reinterpret_cast<bf16_t*>(p.C) + off;
        *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
}

// split-K second pass: C = sum_z slab[z] (+ bias) (+ C)
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, void* C, const float* bias,
                                     int M, int N, long ldc, int splits, int out_f32, int accumulate) {
  const long n4 = N >> 2;
  const long total = (long)M * n4;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int row = (int)(idx / n4);
    const int col = (int)(idx - (long)row * n4) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z)
      v += *reinterpret_cast<const f32x4*>(slabs + ((long)z * M + row) * N + col);
    if (bias != nullptr) v += *reinterpret_cast<const f32x4*>(bias + col);
    const long off = (long)row * ldc + col;
    if (out_f32) {
      float* dst = reinterpret_cast<float*>(C) + off;
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dst);
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      bf16_t* dst = reinterpret_cast<bf16_t*>(C) + off;
      *reinterpret_cast<u32x2*>(dst) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

// Shape-agnostic kernel for operands the MFMA path cannot take (K or leading dims not multiples
// of 8, N not a multiple of 4, unaligned bases): one output element per thread, fp32 FMA chain.
// Used by tiny tabular layers (FCNN 10 -> 3 etc.); never on the ViT path.
__global__ void gemm_bf16_generic_kernel(GemmParams p, int a_trans, int b_trans) {
  const long total = (long)p.M * p.N;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
    float acc = 0.f;
    for (int k = 0; k < p.K; ++k) {
      const float a = bf16_to_f32(a_trans ? p.A[(long)k * p.lda + m] : p.A[(long)m * p.lda + k]);
      const float b = bf16_to_f32(b_trans ? p.B[(long)k * p.ldb + n] : p.B[(long)n * p.ldb + k]);
      acc = fmaf(a, b, acc);
    }
    if (p.bias != nullptr) acc += p.bias[n];
    const long off = (long)m * p.ldc + n;
    if (p.epilogue == CFHIP_EPI_GELU) {
      const float pre = bf16_to_f32(f32_to_bf16(acc));
      if (p.aux_out != nullptr) p.aux_out[off] = f32_to_bf16(acc);
      acc = gelu_erf_f(pre);
    } else if (p.epilogue == CFHIP_EPI_RESIDUAL) {
      acc += bf16_to_f32(p.aux_in[off]);
    } else if (p.epilogue == CFHIP_EPI_DGELU) {
      acc *= gelu_erf_grad_f(bf16_to_f32(p.aux_in[off]));
    }
    if (p.out_f32) {
      float* dst = reinterpret_cast<float*>(p.C) + off;
      *dst = p.accumulate ? *dst + acc : acc;
    } else {
      reinterpret_cast<bf16_t*>(p.C)[off] = f32_to_bf16(acc);
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" int cfhip_gemm_bf16(const void* A, const void* B, void* C, const float* bias,
                               const void* aux_in, void* aux_out, int M, int N, int K, int64_t lda,
                               int64_t ldb, int64_t ldc, int a_trans, int b_trans, int epilogue,
                               int out_dtype, int accumulate, int split_k, void* workspace,
                               size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(A && B && C, "gemm: null operand");
  CFHIP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  CFHIP_REQUIRE(epilogue >= 0 && epilogue <= 3, "gemm: bad epilogue %d", epilogue);
  CFHIP_REQUIRE(!(epilogue == CFHIP_EPI_RESIDUAL || epilogue == CFHIP_EPI_DGELU) || aux_in,
                "gemm: epilogue %d needs aux_in", epilogue);
  CFHIP_REQUIRE(!accumulate || out_dtype == 1, "gemm: accumulate needs f32 output");
  CFHIP_REQUIRE(!(a_trans && !b_trans), "gemm: layout (a_trans=1, b_trans=0) is not provided");
  if (split_k < 1) split_k = 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  GemmParams p;
  p.A = reinterpret_cast<const bf16_t*>(A);
  p.B = reinterpret_cast<const bf16_t*>(B);
  p.C = C;
  p.bias = bias;
  p.aux_in = reinterpret_cast<const bf16_t*>(aux_in);
  p.aux_out = reinterpret_cast<bf16_t*>(aux_out);
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.epilogue = epilogue; p.out_f32 = out_dtype; p.accumulate = accumulate;
  p.slabs = nullptr;
  p.tiles_m = (M + BM - 1) / BM;
  p.tiles_n = (N + BN - 1) / BN;
  p.k_chunk = ((K + BK - 1) / BK) * BK;

  // operand extents must keep 32-bit tile-relative offsets below 2 GiB
  const long a_span = a_trans ? (long)K * lda * 2 : (long)BM * lda * 2;
  const long b_span = b_trans ? (long)K * ldb * 2 : (long)BN * ldb * 2;
  const bool fast = (K % 8 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (N % 4 == 0) &&
                    (ldc % 4 == 0) && (!a_trans || M % 8 == 0) && (!b_trans || N % 8 == 0) &&
                    aligned16(A) && aligned16(B) && aligned16(C) &&
                    (bias == nullptr || aligned16(bias)) && (aux_in == nullptr || aligned16(aux_in)) &&
                    (aux_out == nullptr || aligned16(aux_out));
  if (!fast) {
    CFHIP_REQUIRE(split_k == 1, "gemm: split_k needs the aligned fast path");
    const long total = (long)M * N;
    const int blocks = (int)((total + 255) / 256 > 65535 ? 65535 : (total + 255) / 256);
    hipLaunchKernelGGL(gemm_bf16_generic_kernel, dim3(blocks), dim3(256), 0, s, p, a_trans, b_trans);
    CFHIP_CHECK_LAUNCH("gemm_generic");
    return CFHIP_OK;
  }

  if (split_k > 1) {
    CFHIP_REQUIRE(epilogue == CFHIP_EPI_NONE, "gemm: split_k supports epilogue NONE only");
    const int steps = (K + BK - 1) / BK;
    if (split_k > steps) split_k = steps;
    const int per = (steps + split_k - 1) / split_k;
    split_k = (steps + per - 1) / per;
    p.k_chunk = per * BK;
  }
  if (split_k > 1) {
    const size_t need = (size_t)split_k * M * N * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
      cfhip_set_error("gemm: split_k=%d needs %zu workspace bytes, got %zu", split_k, need, workspace_bytes);
      return CFHIP_ERR_WORKSPACE;
    }
    p.slabs = reinterpret_cast<float*>(workspace);
  }
  const long max_span = p.slabs ? (long)p.k_chunk : (long)K;
  CFHIP_REQUIRE((a_trans ? max_span * lda * 2 : a_span) < 0x7fffffffL &&
                    (b_trans ? max_span * ldb * 2 : b_span) < 0x7fffffffL,
                "gemm: operand tile span exceeds 2 GiB (lda=%ld ldb=%ld K=%d)", (long)lda, (long)ldb, K);

  dim3 grid(p.tiles_m * p.tiles_n, split_k);
#define CFHIP_LAUNCH_GEMM(AT_, BT_, EPI_) \
  hipLaunchKernelGGL((gemm_bf16_kernel<AT_, BT_, EPI_>), grid, dim3(NTHREADS), 0, s, p)
  if (!a_trans && !b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE: CFHIP_LAUNCH_GEMM(false, false, CFHIP_EPI_NONE); break;
      case CFHIP_EPI_GELU: CFHIP_LAUNCH_GEMM(false, false, CFHIP_EPI_GELU); break;
      case CFHIP_EPI_RESIDUAL: CFHIP_LAUNCH_GEMM(false, false, CFHIP_EPI_RESIDUAL); break;
      default:
        cfhip_set_error("gemm: epilogue %d is not provided for layout (0,0)", epilogue);
        return CFHIP_ERR_INVALID;
    }
  } else if (!a_trans && b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE: CFHIP_LAUNCH_GEMM(false, true, CFHIP_EPI_NONE); break;
      case CFHIP_EPI_DGELU: CFHIP_LAUNCH_GEMM(false, true, CFHIP_EPI_DGELU); break;
      default:
        cfhip_set_error("gemm: epilogue %d is not provided for layout (0,1)", epilogue);
        return CFHIP_ERR_INVALID;
    }
  } else {
    CFHIP_REQUIRE(epilogue == CFHIP_EPI_NONE, "gemm: epilogue %d is not provided for layout (1,1)", epilogue);
    CFHIP_LAUNCH_GEMM(true, true, CFHIP_EPI_NONE);
  }
#undef CFHIP_LAUNCH_GEMM
  CFHIP_CHECK_LAUNCH("gemm_bf16");

  if (split_k > 1) {
    const long total = (long)M * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, C, bias, M, N,
                       (long)ldc, split_k, out_dtype, accumulate);
    CFHIP_CHECK_LAUNCH("splitk_reduce");
  }
  return CFHIP_OK;
}
