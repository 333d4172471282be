// K3/K4: fused scaled-dot-product attention (forward + backward) for gfx950, head_dim 64,
// sequence lengths up to 256 (the whole K / V of one head is resident in LDS: ViT-B/16 has T = 197).
//
// Replaces F.scaled_dot_product_attention reached through sdp_attn (reference toolkit.py:953-963)
// from Attention.forward (attentions.py:254), including the head split / merge permute copies
// (attentions.py:180-185, 270-275): q / k / v are read in place from the packed [B,T,3,H,64]
// projection output and o is written directly as [B,T,H*64].
//
// Structure (all on v_mfma_f32_16x16x32_bf16, one 16-row tile per wave):
//   * "swapped" products: S^T = K Q^T, so that a lane holds 4 consecutive kv positions of ONE query
//     row (row = lane & 15) -> the row softmax is lane-local plus two xor-shuffles (16, 32), and the
//     bf16-packed probabilities are directly the register operand of the P·V MFMA (the k-slot
//     permutation this implies is matched on the V side, a dot product does not care about order).
//   * K / V / Q / dO tiles sit in LDS as [rows][64] bf16 with an XOR swizzle of the 16-byte slots;
//     row-operand fragments are ds_read_b128, column-operand fragments (V in P·V, K in dS·K,
//     dO and Q in the dK/dV pass) are ds_read_b64_tr_b16 hardware-transpose reads.
//   * backward = two passes without atomics: pass A (wave = query tile) recomputes S and dP and
//     accumulates dQ; pass B (wave = kv tile) recomputes them transposed and accumulates dK, dV.
//     Deterministic; costs 7 instead of 5 GEMM units, attention is 3.4 % of the ViT FLOPs.
#include "common.h"
#include <type_traits>
#include <math.h>

namespace {

constexpr int DH = CFHIP_ATTN_HEAD_DIM;  // 64
constexpr float LOG2E = 1.4426950408889634f;

struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  bf16_t* o; const bf16_t* o_in; const bf16_t* d_o;
  float* lse; float* delta;
  const uint8_t* mask;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int B, H, Tq, Tk;
  long q_sb, q_st, kv_sb, kv_st, o_sb, o_st;
  long ms_b, ms_h, ms_q;
  float scale;
  int causal;
  int dh;           // head_dim of the general kernels (the resident fast kernels are head_dim 64 only)
  int delta_ready;  // dK/dV pass: p.delta was already written by the dQ pass of the same call
  // dropout on the attention probabilities (general kernels, DROP instantiations): keep(b, h, i, j) = byte (j & 3) of word
  // (i & 3) of philox(drop_offset + ((b H + h) drop_bq + (i >> 2)) drop_bk + (j >> 2), drop_seed) >= drop_thresh, i.e. one
  // Philox call per 4 x 4 block of the score matrix, the same bits whichever way a kernel walks it; kept values are
  // scaled by drop_scale = 1 / (1 - drop_thresh / 256)
  unsigned long long drop_seed, drop_offset;
  int drop_bq, drop_bk;
  unsigned drop_thresh;
  float drop_scale;
#ifdef CFHIP_ABLATE
  int ablate;       // benchmarking only (forward): bit0 skip the K/V DMA, bit1 skip the tile loop, bit2 skip stores; attn_bwd_one2_kernel: 8 no phase 1, 16 no phase 2, 32 no statistics, 64 no dK / dV stores, 128 no K / V fragments, 256 no staging of the next head
#endif
};

// Timing ablations exist only in -DCFHIP_ABLATE builds (tools/build_variant.sh); the product library has none.
#ifdef CFHIP_ABLATE
int g_attn_ablate = 0;
#define ATTN_ABL(bit) (p.ablate & (bit))
#else
#define ATTN_ABL(bit) false
#endif

// XOR key of the 16-byte slots of a tile row (128 B per row: two rows per 256-byte LDS bank row).  The key moves whole
// 32-byte chunks (bit 0 clear) and takes four values over the row pairs of an 8-row block: the 16 lanes of a
// ds_read_b128 group (8 rows on each half of the bank row, two adjacent slots) land on 8 distinct slots per half, and the
// 8 rows x 32 bytes a half-wave of ds_read_b64_tr_b16 touches land on 4 distinct chunks per half — both conflict-free.
// (The first version, key (row >> 1) & 7, differed only in bit 0 between rows r and r + 2: the two slots of a 32-byte
// chunk swapped places and every transposing read was a 2-way conflict — SQ_LDS_BANK_CONFLICT 26 % of the LDS cycles.)
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) << 1; }

// byte offset of element (row, col) in a swizzled [rows][64] bf16 tile (128 B per row)
__device__ __forceinline__ int tile_off(int row, int col) {
  return row * 128 + ((((col >> 3) ^ swz(row))) << 4) + ((col & 7) << 1);
}

// Cooperative asynchronous load of `rows_pad` rows (zero beyond rows_valid) of one head into a swizzled
// tile: LDS-DMA (buffer_load_dwordx4 ... lds), one instruction = 8 rows x 128 B, no VGPR staging.  The
// LDS image of the DMA is lane-linear, so the 16-byte-slot swizzle is applied to the SOURCE address;
// rows past the end are range-checked to zero by the buffer descriptor.  Completion = the issuing
// wave's vmcnt, then a workgroup barrier.
__device__ __forceinline__ void dma_tile(char* tile, const bf16_t* base, long stride_t, int rows_valid,
                                         int rows_pad, int wave, int nwaves, int lane) {
  long bytes = ((long)(rows_valid - 1) * stride_t + DH) * 2;
  if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, (int)bytes, 0x00020000);
  const int r8 = lane >> 3, slot = lane & 7;
  for (int inst = wave; inst < rows_pad / 8; inst += nwaves) {
    const int row = inst * 8 + r8;
    const int chunk = slot ^ swz(row);
    const unsigned off = row < rows_valid ? (unsigned)(row * stride_t * 2 + chunk * 16) : 0x80000000u;
    lds_dma16(rsrc, tile + inst * 1024, off);
  }
}

// row-operand fragment: lane (i = l&15, g = l>>4) <- tile[row0 + i][ks*32 + 8g .. +8]
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 15);
  const int slot = ks * 4 + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((slot ^ swz(row)) << 4));
}

// column-operand fragment over the 32-row block starting at `r32`, columns c0..c0+15:
// lane (n = l&15, g = l>>4) <- { tile[r32 + 4g + e][c0 + n], e = 0..3 ; tile[r32 + 16 + 4g + e][c0 + n] }
// (the k-slot order that the packed S^T / P registers have).
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int r32, int c0, int lane) {
  const int g = lane >> 4, s = lane & 15;
  const int r_lo = r32 + 4 * g + (s >> 2);
  const int col = c0 + 4 * (s & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + tile_off(r_lo, col)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + tile_off(r_lo + 16, col)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// row-operand fragment straight from global memory (one 16-row tile owned by this wave)
__device__ __forceinline__ bf16x8 frag_global(const bf16_t* base, long stride_t, int row0, int rows_valid,
                                              int ks, int lane) {
  const int row = row0 + (lane & 15);
  bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < rows_valid)
    r = *reinterpret_cast<const bf16x8*>(base + (long)row * stride_t + ks * 32 + (lane >> 4) * 8);
  return r;
}

// Packed f32 VALU, measured twice in round 6 (profiles/r06/attn_packed_f32_rejected.txt): (1) the softmax math of the long-sequence backward
// passes written on vectors (16 fma + 16 mul -> 8 v_pk_fma_f32 + 8 v_pk_mul_f32 per block) made dQ 2.5 % and dK / dV 4-5 % SLOWER — the
// packed instructions are no bargain beside MFMAs (MI355X_MICROARCH.md prices one v_pk_fma_f32 at +22 cycles against two v_fma_f32);
// (2) the other direction — the v_pk_mul_f32 hipcc's SLP pass makes of the one-pass backward's `pr * (dP - delta)` pinned to single
// v_mul_f32 through inline asm — was 3.4 % slower too (12 more conversions).  And inline asm must never READ an MFMA result: hipcc's
// hazard recogniser does not see the asm as a VALU consumer and omits the wait states (the persistent forward with asm fma / max3
// on the scores returned NaN).  So: scalar source, the compiler's own choice of packing, asm only on VALU results (max_nn).

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  union { bf16x8 v; unsigned w[4]; } u;
  u.w[0] = pack_bf16x2(a[0], a[1]); u.w[1] = pack_bf16x2(a[2], a[3]);
  u.w[2] = pack_bf16x2(b[0], b[1]); u.w[3] = pack_bf16x2(b[2], b[3]);
  return u.v;
}

// One 64-column output row per lane group: lane (n = l & 15, g = l >> 4) holds columns 16 dt + 4 g .. + 3 of row n for the
// four 16-column tiles dt.  Written as they are, that is four 8-byte stores per lane, 32-byte runs per row and
// instruction — the store tail of these kernels is issue-bound (dK/dV: 34 of 105 us).  Lanes g and g ^ 1 swap half of
// their values instead (two ds_bpermute per tile pair), so that every lane owns 8 consecutive columns of ONE tile and
// the row goes out as two 16-byte stores per lane, 64-byte runs per row and instruction.  Call with all lanes active.
__device__ __forceinline__ void store_row64(bf16_t* row, const f32x4 (&t)[4], float mul, int g, bool valid) {
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    const f32x4 a = t[2 * pr] * mul, b = t[2 * pr + 1] * mul;
    const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
    const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
    const bool odd = (g & 1) != 0;
    const unsigned r0 = (unsigned)__shfl_xor((int)(odd ? a0 : b0), 16, 64);  // even g sends tile 2pr+1, odd g tile 2pr
    const unsigned r1 = (unsigned)__shfl_xor((int)(odd ? a1 : b1), 16, 64);
    // even g: own tile-2pr columns 4g..4g+3, then the partner's (g + 1); odd g: the partner's (g - 1) tile-(2pr+1)
    // columns, then its own
    const u32x4 w = odd ? u32x4{r0, r1, b0, b1} : u32x4{a0, a1, r0, r1};
    const int col = (2 * pr + (odd ? 1 : 0)) * 16 + 4 * (g & 2);
    if (valid) *reinterpret_cast<u32x4*>(row + col) = w;
  }
}

// the same for ONE pair of 16-column tiles (2 pr, 2 pr + 1) of the row: one 16-byte store per lane
__device__ __forceinline__ void store_row32(bf16_t* row, const f32x4& ta, const f32x4& tb, float mul, int g, bool valid, int pr) {
  const f32x4 a = ta * mul, b = tb * mul;
  const unsigned a0 = pack_bf16x2(a[0], a[1]), a1 = pack_bf16x2(a[2], a[3]);
  const unsigned b0 = pack_bf16x2(b[0], b[1]), b1 = pack_bf16x2(b[2], b[3]);
  const bool odd = (g & 1) != 0;
  const unsigned r0 = (unsigned)__shfl_xor((int)(odd ? a0 : b0), 16, 64);
  const unsigned r1 = (unsigned)__shfl_xor((int)(odd ? a1 : b1), 16, 64);
  const u32x4 w = odd ? u32x4{r0, r1, b0, b1} : u32x4{a0, a1, r0, r1};
  const int col = (2 * pr + (odd ? 1 : 0)) * 16 + 4 * (g & 2);
  if (valid) *reinterpret_cast<u32x4*>(row + col) = w;
}

__device__ __forceinline__ bool keep_at(const AttnParams& p, int b, int h, int i, int j) {
  if (j >= p.Tk) return false;
  if (p.causal && j > i) return false;
  if (p.mask != nullptr) return p.mask[(long)b * p.ms_b + (long)h * p.ms_h + (long)i * p.ms_q + j] != 0;
  return true;
}

__device__ __forceinline__ Philox drop_block(const AttnParams& p, int b, int h, int block_row, int block_col) {
  const unsigned long long ctr =
      p.drop_offset + ((unsigned long long)(b * p.H + h) * p.drop_bq + block_row) * p.drop_bk + block_col;
  return philox4x32_10(ctr, p.drop_seed);
}
__device__ __forceinline__ unsigned pick_word(const Philox& r, int w) {  // r.c[w & 3] without a private-memory array
  const unsigned lo = (w & 1) ? r.c[1] : r.c[0], hi = (w & 1) ? r.c[3] : r.c[2];
  return (w & 2) ? hi : lo;
}

__device__ __forceinline__ float group_max(float v) {  // across the 4 lanes sharing l & 15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
// The same through gfx950's lane swaps (round 6): `__shfl_xor` is a ds_bpermute — an LDS round trip behind ~6 VALU of address arithmetic,
// twice in a row on the critical path of every 64-key block of the long-sequence forward.  v_permlane16_swap(v, v) leaves the two halves of
// every 32-lane pair of rows side by side (result 0: the even rows' values in both rows, result 1: the odd rows'), v_permlane32_swap(v, v)
// the two 32-lane halves: one max each.
// max of two values that are never NaN (scores, running maxima: finite or -inf): fmaxf() on values that come out of a lane swap
// (integers to the compiler) is preceded by a quieting v_max_f32 v, v, v per operand — three instructions for one; the med3(x, y, +inf)
// builtin is folded back into the same maxnum.  The instruction itself, then.
__device__ __forceinline__ float max_nn(float x, float y) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float group_max_swap(float v) {
  const unsigned u = __float_as_uint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float m1 = max_nn(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned w = __float_as_uint(m1);
  const auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
  return max_nn(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------------------
// forward: workgroup = (query chunk of nwaves*16 rows, head, batch); NB = ceil(Tk / 32)
// ------------------------------------------------------------------------------------------------
// PLAIN = no mask, not causal (the ViT path): the only masking is the kv tail (j >= Tk), decided per
// 16-wide tile with wave-uniform branches — no per-element predicate registers.
template <int NB, bool PLAIN>
__global__ __launch_bounds__(512, 4) void attn_fwd_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + NB * 32 * 128;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * DH;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * DH;
  if (!ATTN_ABL(1)) {
    dma_tile(Ks, kb, p.kv_st, p.Tk, NB * 32, wave, nwaves, lane);
    dma_tile(Vs, vb, p.kv_st, p.Tk, NB * 32, wave, nwaves, lane);
  }
  lds_dma_wait_all();
  __syncthreads();  // (drains the DMA: vmcnt(0) + barrier)
  if (ATTN_ABL(2)) return;

  const int i = lane & 15, g = lane >> 4;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * DH;
  // K / V are loaded once per (batch, head); the waves walk the query tiles
  for (int row0 = (blockIdx.x * nwaves + wave) * 16; row0 < p.Tq; row0 += gridDim.x * nwaves * 16) {
  const int qi = row0 + i;
  const bf16x8 qf0 = frag_global(qb, p.q_st, row0, p.Tq, 0, lane);
  const bf16x8 qf1 = frag_global(qb, p.q_st, row0, p.Tq, 1, lane);

  f32x4 st[2 * NB];
#pragma unroll
  for (int jt = 0; jt < 2 * NB; ++jt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 0, lane), qf0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 1, lane), qf1, acc, 0, 0, 0);
    st[jt] = acc;
    if (jt & 1) __builtin_amdgcn_sched_barrier(0);  // at most 4 K fragments in flight: 128 VGPRs, 4 waves / SIMD
  }
  // row max / sum of the scaled scores in the log2 domain, masked
  const float sl2 = p.scale * LOG2E;
  float mx = -INFINITY;
  float l = 0.f;
  if (PLAIN) {
    // (sl2 > 0: the maximum commutes with the scaling, so the pass over the 8 NB scores per lane is max only and the
    // scaling rides in the FMA of the exponent: e = exp2(s * sl2 - max * sl2), two scores per v_pk_fma_f32 / v_pk_add_f32)
#pragma unroll
    for (int jt = 2 * NB - 2; jt < 2 * NB; ++jt) {  // nb = ceil(Tk / 32): only the last two tiles can touch the kv tail
      if (jt * 16 >= p.Tk) {                         // tile entirely past the end (wave-uniform)
        st[jt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      } else if (jt * 16 + 15 >= p.Tk) {             // the tile straddling Tk
#pragma unroll
        for (int r = 0; r < 4; ++r) st[jt][r] = (jt * 16 + 4 * g + r < p.Tk) ? st[jt][r] : -INFINITY;
      }
    }
#pragma unroll
    for (int jt = 0; jt < 2 * NB; ++jt) mx = fmaxf(mx, fmaxf(fmaxf(st[jt][0], st[jt][1]), fmaxf(st[jt][2], st[jt][3])));
    mx = group_max(mx) * sl2;
    const f32x4 sc4 = {sl2, sl2, sl2, sl2}, nm4 = {-mx, -mx, -mx, -mx};
    f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < 2 * NB; ++jt) {
      const f32x4 x = __builtin_elementwise_fma(st[jt], sc4, nm4);
      f32x4 e;
#pragma unroll
      for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(x[r]);  // raw v_exp_f32: arguments are <= 0
      st[jt] = e;
      l4 += e;
    }
    l = group_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
  } else {
#pragma unroll
    for (int jt = 0; jt < 2 * NB; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + 4 * g + r;
        const float x = keep_at(p, b, h, qi < p.Tq ? qi : p.Tq - 1, j) ? st[jt][r] * sl2 : -INFINITY;
        st[jt][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = group_max(mx);
#pragma unroll
    for (int jt = 0; jt < 2 * NB; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[jt][r] - mx);  // raw v_exp_f32: arguments are <= 0
        st[jt][r] = e;
        l += e;
      }
    }
    l = group_sum(l);
  }

  // O^T[d][i] = sum_j V^T[d][j] P^T[j][i]   (operands swapped: lane ends with 4 consecutive d of row i)
  f32x4 ot[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < NB; ++a) {
    const bf16x8 pa = pack8(st[2 * a], st[2 * a + 1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Vs, a * 32, dt * 16, lane), pa, ot[dt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  {
    const bool valid = qi < p.Tq && !ATTN_ABL(4);
    const float inv = 1.0f / l;
    store_row64(p.o + (long)b * p.o_sb + (long)qi * p.o_st + h * DH, ot, inv, g, valid);
    if (valid && g == 0 && p.lse != nullptr)
      p.lse[((long)b * p.H + h) * p.Tq + qi] = (mx + log2f(l)) * (1.0f / LOG2E);
  }
  }  // query-tile loop
}

// ------------------------------------------------------------------------------------------------
// backward pass A: dQ (and delta = rowsum(dO * O)).  workgroup as in forward; K, V in LDS.
// ------------------------------------------------------------------------------------------------
// PLAIN (no mask, not causal): no predicate at all — kv rows past Tk are ZERO in LDS (DMA range check),
// so their dS (finite: s = 0, dP = 0) meets K = 0 in the dQ product; padded query rows carry lse = +inf.
template <bool PLAIN>
__global__ __launch_bounds__(512, 4) void attn_bwd_dq_kernel(AttnParams p, int nb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + nb * 32 * 128;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  dma_tile(Ks, p.k + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, nb * 32, wave, nwaves, lane);
  dma_tile(Vs, p.v + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, nb * 32, wave, nwaves, lane);
  lds_dma_wait_all();
  __syncthreads();

  const int i = lane & 15, g = lane >> 4;
  for (int row0 = (blockIdx.x * nwaves + wave) * 16; row0 < p.Tq; row0 += gridDim.x * nwaves * 16) {
  const int qi = row0 + i;
  const bool qvalid = qi < p.Tq;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * DH;
  const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * DH;
  const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * DH;
  const bf16x8 qf0 = frag_global(qb, p.q_st, row0, p.Tq, 0, lane);
  const bf16x8 qf1 = frag_global(qb, p.q_st, row0, p.Tq, 1, lane);
  const bf16x8 dof0 = frag_global(dob, p.o_st, row0, p.Tq, 0, lane);
  const bf16x8 dof1 = frag_global(dob, p.o_st, row0, p.Tq, 1, lane);
  // delta_i = sum_d dO[i][d] * O[i][d]
  float delta;
  {
    const bf16x8 of0 = frag_global(ob, p.o_st, row0, p.Tq, 0, lane);
    const bf16x8 of1 = frag_global(ob, p.o_st, row0, p.Tq, 1, lane);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s += bf16_to_f32((bf16_t)dof0[e]) * bf16_to_f32((bf16_t)of0[e]);
      s += bf16_to_f32((bf16_t)dof1[e]) * bf16_to_f32((bf16_t)of1[e]);
    }
    delta = group_sum(s);
  }
  const long stat = ((long)b * p.H + h) * p.Tq + qi;
  if (qvalid && g == 0) p.delta[stat] = delta;
  const float lse2 = qvalid ? p.lse[stat] * LOG2E : INFINITY;  // +inf -> p = 0 for padded rows
  const float sl2 = p.scale * LOG2E;

  f32x4 dqt[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int a = 0; a < nb; ++a) {
    f32x4 ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int jt = 2 * a + t;
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 0, lane), qf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 1, lane), qf1, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs, jt * 16, 0, lane), dof0, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs, jt * 16, 1, lane), dof1, dp, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = __builtin_amdgcn_exp2f(s[r] * sl2 - lse2);
        if (!PLAIN) pr = keep_at(p, b, h, qvalid ? qi : p.Tq - 1, jt * 16 + 4 * g + r) ? pr : 0.f;
        ds[t][r] = pr * (dp[r] - delta);
      }
    }
    const bf16x8 dsp = pack8(ds[0], ds[1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Ks, a * 32, dt * 16, lane), dsp, dqt[dt], 0, 0, 0);
  }
  store_row64(p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * DH, dqt, p.scale, g, qvalid);
  }  // query-tile loop
}

// ------------------------------------------------------------------------------------------------
// backward pass B: dK, dV.  workgroup = (kv chunk of nwaves*16 rows, head, batch); Q, dO, lse,
// delta of the whole head in LDS; nbq = ceil(Tq / 32).
// ------------------------------------------------------------------------------------------------
// PLAIN: no predicate — a kv row past Tk only pollutes its own (never stored) dK / dV row, padded query
// rows carry lse = +inf (p = 0).
template <bool PLAIN>
__global__ __launch_bounds__(512, PLAIN ? 4 : 2) void attn_bwd_dkv_kernel(AttnParams p, int nbq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + nbq * 32 * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * nbq * 32 * 128);
  float* delta_s = lse_s + nbq * 32;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  if (!ATTN_ABL(1)) {
  dma_tile(Qs, p.q + (long)b * p.q_sb + h * DH, p.q_st, p.Tq, nbq * 32, wave, nwaves, lane);
  dma_tile(dOs, p.d_o + (long)b * p.o_sb + h * DH, p.o_st, p.Tq, nbq * 32, wave, nwaves, lane);
  // lse and delta_i = sum_d dO[i][d] * O[i][d] of every query row (delta is recomputed here from the
  // L2-resident dO / O rows so that this pass does not depend on the dQ pass: the two kernels run
  // concurrently on two streams).  8 consecutive lanes own the 8 16-byte slots of one row.
  for (int idx = threadIdx.x; idx < nbq * 32 * 8; idx += blockDim.x) {
    const int t = idx >> 3, slot = idx & 7;
    float s = 0.f;
    if (p.delta_ready) {
      if (slot == 0 && t < p.Tq) s = p.delta[((long)b * p.H + h) * p.Tq + t];
    } else if (t < p.Tq) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(p.d_o + (long)b * p.o_sb + (long)t * p.o_st + h * DH + slot * 8);
      const u32x4 c = *reinterpret_cast<const u32x4*>(p.o_in + (long)b * p.o_sb + (long)t * p.o_st + h * DH + slot * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += bf16lo(a[e]) * bf16lo(c[e]) + bf16hi(a[e]) * bf16hi(c[e]);
    }
    if (!p.delta_ready) {
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
    }
    if (slot == 0) {
      delta_s[t] = s;
      lse_s[t] = t < p.Tq ? p.lse[((long)b * p.H + h) * p.Tq + t] * LOG2E : INFINITY;
    }
  }
  }
  lds_dma_wait_all();
  __syncthreads();
  if (ATTN_ABL(2)) return;

  const int n = lane & 15, g = lane >> 4;
  for (int row0 = (blockIdx.x * nwaves + wave) * 16; row0 < p.Tk; row0 += gridDim.x * nwaves * 16) {
  const int kj = row0 + n;  // kv rows of this wave
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * DH;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * DH;
  const bf16x8 kf0 = frag_global(kb, p.kv_st, row0, p.Tk, 0, lane);
  const bf16x8 kf1 = frag_global(kb, p.kv_st, row0, p.Tk, 1, lane);
  const bf16x8 vf0 = frag_global(vb, p.kv_st, row0, p.Tk, 0, lane);
  const bf16x8 vf1 = frag_global(vb, p.kv_st, row0, p.Tk, 1, lane);
  const float sl2 = p.scale * LOG2E;

  f32x4 dkt[4], dvt[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int a = 0; a < nbq; ++a) {
    f32x4 pp[2], ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int it = 2 * a + t;
      // S[i][j] with lane <- rows i = it*16 + 4g + r, column j = kj
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs, it * 16, 0, lane), kf0, s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs, it * 16, 1, lane), kf1, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOs, it * 16, 0, lane), vf0, dp, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOs, it * 16, 1, lane), vf1, dp, 0, 0, 0);
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + it * 16 + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + it * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr = __builtin_amdgcn_exp2f(s[r] * sl2 - l4[r]);
        if (!PLAIN) {
          const int qi = it * 16 + 4 * g + r;
          pr = (qi < p.Tq && keep_at(p, b, h, qi, kj)) ? pr : 0.f;
        }
        pp[t][r] = pr;
        ds[t][r] = pr * (dp[r] - d4[r]);
      }
    }
    const bf16x8 ppk = pack8(pp[0], pp[1]);
    const bf16x8 dsk = pack8(ds[0], ds[1]);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dOs, a * 32, dt * 16, lane), ppk, dvt[dt], 0, 0, 0);
      dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qs, a * 32, dt * 16, lane), dsk, dkt[dt], 0, 0, 0);
    }
  }
  {
    const bool valid = kj < p.Tk && !ATTN_ABL(4);
    store_row64(p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dkt, p.scale, g, valid);
    store_row64(p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dvt, 1.0f, g, valid);
  }
  }  // kv-tile loop
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the dK/dV pass (PLAIN only, sequence lengths 129 .. 256 = the ViT step).
// One 16-wave workgroup per CU walks (batch, head) pairs; the operands of the NEXT head stream into a second LDS buffer
// while the waves work on the current one.  In the one-shot kernels above the prologue DMA (57 KB per head) is not
// covered by anything: every workgroup of a launch sits in it at the same time.  Worth 8 % on this pass alone and 40 us of
// its in-step duration, nothing on the wall clock of the step (profiles/r02/attn_vit_ablation.log); the same form of the
// forward and dQ passes was built, measured (no gain alone or in the step) and removed.  The DMA / statistics loads of the next head are issued UNCONDITIONALLY (out of
// range offsets when there is no next head), right after this head's own fragment loads, so that the compiler's vmcnt
// bookkeeping can wait for the fragments while the younger DMA stays in flight.
// ------------------------------------------------------------------------------------------------
constexpr int PERS_WAVES = 16;

// 2 instructions per wave cover up to 256 rows (32 instructions of 8 rows)
__device__ __forceinline__ void dma_tile_pers(char* tile, const bf16_t* base, long stride_t, int rows_valid, bool live,
                                              int wave, int lane) {
  long bytes = ((long)(rows_valid - 1) * stride_t + DH) * 2;
  if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, live ? (int)bytes : 0, 0x00020000);
  const int r8 = lane >> 3, slot = lane & 7;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int inst = wave + j * PERS_WAVES;
    const int row = inst * 8 + r8;
    const int chunk = slot ^ swz(row);
    const unsigned off = (live && row < rows_valid) ? (unsigned)(row * stride_t * 2 + chunk * 16) : 0x80000000u;
    lds_dma16(rsrc, tile + inst * 1024, off);
  }
}

template <int NBQ>
__global__ __launch_bounds__(PERS_WAVES * 64, 4) void attn_bwd_dkv_pers_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // every wave issues 2 DMA instructions per operand whatever NBQ is (a fixed count the compiler can wait on): the tile
  // slot is 256 rows, the instructions beyond ROWS write zeros into its padding
  constexpr int ROWS = NBQ * 32, TILE = 256 * 128, BUF = 2 * TILE + 2 * ROWS * 4;
  static_assert(ROWS <= 256 && ROWS * 8 <= 2 * PERS_WAVES * 64, "two DMA instructions / two statistics slots per thread");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int heads = p.B * p.H;
  const float sl2 = p.scale * LOG2E;

  auto issue = [&](int hh, char* buf, bool live) {
    const int b = hh / p.H, h = hh - b * p.H;
    dma_tile_pers(buf, p.q + (long)b * p.q_sb + h * DH, p.q_st, p.Tq, live, wave, lane);
    dma_tile_pers(buf + TILE, p.d_o + (long)b * p.o_sb + h * DH, p.o_st, p.Tq, live, wave, lane);
  };
  // lse (log2 domain) and delta of row t = idx (thread-per-row, 2 slots per thread); needs p.delta from the dQ pass
  auto stats_load = [&](int hh, bool live, float (&lv)[2], float (&dv)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = (int)threadIdx.x + j * PERS_WAVES * 64;
      const bool ok = live && t < p.Tq;
      const long at = (long)hh * p.Tq + (ok ? t : 0);
      const float l = p.lse[ok ? at : 0], d = p.delta[ok ? at : 0];
      lv[j] = ok ? l * LOG2E : INFINITY;
      dv[j] = ok ? d : 0.f;
    }
  };
  auto stats_store = [&](char* buf, const float (&lv)[2], const float (&dv)[2]) {
    float* lse_s = reinterpret_cast<float*>(buf + 2 * TILE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = (int)threadIdx.x + j * PERS_WAVES * 64;
      if (t < ROWS) {
        lse_s[t] = lv[j];
        lse_s[ROWS + t] = dv[j];
      }
    }
  };

  int hh = blockIdx.x;
  int cur = 0;
  {
    float lv[2], dv[2];
    issue(hh, smem, hh < heads);
    stats_load(hh, hh < heads, lv, dv);
    stats_store(smem, lv, dv);
  }
  lds_dma_wait_all();
  __syncthreads();
  for (; hh < heads; hh += gridDim.x) {
    const int b = hh / p.H, h = hh - b * p.H;
    char* Qs = smem + cur * BUF;
    char* dOs = Qs + TILE;
    const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * TILE);
    const float* delta_s = lse_s + ROWS;
    const int row0 = wave * 16;
    const int kj = row0 + n;
    const bf16_t* kb = p.k + (long)b * p.kv_sb + h * DH;
    const bf16_t* vb = p.v + (long)b * p.kv_sb + h * DH;
    // this head's own fragments first: older than the DMA below in the vmcnt order
    const bf16x8 kf0 = frag_global(kb, p.kv_st, row0, p.Tk, 0, lane);
    const bf16x8 kf1 = frag_global(kb, p.kv_st, row0, p.Tk, 1, lane);
    const bf16x8 vf0 = frag_global(vb, p.kv_st, row0, p.Tk, 0, lane);
    const bf16x8 vf1 = frag_global(vb, p.kv_st, row0, p.Tk, 1, lane);
    const int nxt = hh + gridDim.x;
    float lv[2], dv[2];
    issue(nxt < heads ? nxt : hh, smem + (cur ^ 1) * BUF, nxt < heads);
    stats_load(nxt < heads ? nxt : hh, nxt < heads, lv, dv);

    f32x4 dkt[4], dvt[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (row0 < p.Tk) {
      for (int a = 0; a < NBQ; ++a) {
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int it = 2 * a + t;
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs, it * 16, 0, lane), kf0, sc, 0, 0, 0);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs, it * 16, 1, lane), kf1, sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOs, it * 16, 0, lane), vf0, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOs, it * 16, 1, lane), vf1, dp, 0, 0, 0);
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + it * 16 + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_s + it * 16 + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(sc[r] * sl2 - l4[r]);
            pp[t][r] = pr;
            ds[t][r] = pr * (dp[r] - d4[r]);
          }
        }
        const bf16x8 ppk = pack8(pp[0], pp[1]);
        const bf16x8 dsk = pack8(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(dOs, a * 32, dt * 16, lane), ppk, dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Qs, a * 32, dt * 16, lane), dsk, dkt[dt], 0, 0, 0);
        }
      }
    }
    {
      const bool valid = kj < p.Tk;
      store_row64(p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dkt, p.scale, g, valid);
      store_row64(p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dvt, 1.0f, g, valid);
    }
    stats_store(smem + (cur ^ 1) * BUF, lv, dv);
    lds_dma_wait_all();
    __syncthreads();  // the next head's operands have landed (vmcnt 0) and nobody reads this head's buffer any more
    cur ^= 1;
  }
}

// The forward in the persistent form (round 4; PLAIN only — the ViT path): wave w owns query rows 16 w .. 16 w + 15 of every
// head its workgroup visits, the next head's K / V stream into the second buffer.  The per-tile arithmetic is attn_fwd_kernel's
// (same instruction order: outputs and log-sum-exp are bit-identical).  Option "attn_persistent" bit 2.
template <int NB>
__global__ __launch_bounds__(PERS_WAVES * 64, 4) void attn_fwd_pers_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 256 * 128, BUF = 2 * TILE;
  static_assert(NB * 32 <= 256, "K / V of one head fit the 256-row tile slot");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int heads = p.B * p.H;
  const float sl2 = p.scale * LOG2E;
  auto issue = [&](int hh, char* buf, bool live) {
    const int b = hh / p.H, h = hh - b * p.H;
    dma_tile_pers(buf, p.k + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, live, wave, lane);
    dma_tile_pers(buf + TILE, p.v + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, live, wave, lane);
  };
  int hh = blockIdx.x;
  int cur = 0;
  issue(hh, smem, hh < heads);
  lds_dma_wait_all();
  __syncthreads();
  const int row0 = wave * 16;
  const int qi = row0 + i;
  // this wave's Q fragments: of the first head here, of every later head right behind the S products of the head before (their
  // global-memory latency used to sit in front of the first MFMA of every head)
  bf16x8 qf0 = {0, 0, 0, 0, 0, 0, 0, 0}, qf1 = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hh < heads) {
    const int b = hh / p.H, h = hh - b * p.H;
    const bf16_t* qb = p.q + (long)b * p.q_sb + h * DH;
    qf0 = frag_global(qb, p.q_st, row0, p.Tq, 0, lane);
    qf1 = frag_global(qb, p.q_st, row0, p.Tq, 1, lane);
  }
  for (; hh < heads; hh += gridDim.x) {
    const int b = hh / p.H, h = hh - b * p.H;
    const char* Ks = smem + cur * BUF;
    const char* Vs = Ks + TILE;
    const int nxt = hh + gridDim.x;
    issue(nxt < heads ? nxt : hh, smem + (cur ^ 1) * BUF, nxt < heads);
    if (row0 < p.Tq) {
      f32x4 st[2 * NB];
#pragma unroll
      for (int jt = 0; jt < 2 * NB; ++jt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 0, lane), qf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 1, lane), qf1, acc, 0, 0, 0);
        st[jt] = acc;
        if (jt & 1) __builtin_amdgcn_sched_barrier(0);
      }
      if (nxt < heads) {
        const int bn = nxt / p.H, hn = nxt - bn * p.H;
        const bf16_t* qn = p.q + (long)bn * p.q_sb + hn * DH;
        qf0 = frag_global(qn, p.q_st, row0, p.Tq, 0, lane);
        qf1 = frag_global(qn, p.q_st, row0, p.Tq, 1, lane);
      }
#pragma unroll
      for (int jt = 2 * NB - 2; jt < 2 * NB; ++jt) {
        if (jt * 16 >= p.Tk) {
          st[jt] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        } else if (jt * 16 + 15 >= p.Tk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) st[jt][r] = (jt * 16 + 4 * g + r < p.Tk) ? st[jt][r] : -INFINITY;
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int jt = 0; jt < 2 * NB; ++jt) mx = fmaxf(mx, fmaxf(fmaxf(st[jt][0], st[jt][1]), fmaxf(st[jt][2], st[jt][3])));
      mx = group_max(mx) * sl2;
      const f32x4 sc4 = {sl2, sl2, sl2, sl2}, nm4 = {-mx, -mx, -mx, -mx};
      f32x4 l4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jt = 0; jt < 2 * NB; ++jt) {
        const f32x4 x = __builtin_elementwise_fma(st[jt], sc4, nm4);
        f32x4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(x[r]);
        st[jt] = e;
        l4 += e;
      }
      const float l = group_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
      f32x4 ot[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        const bf16x8 pa = pack8(st[2 * a], st[2 * a + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Vs, a * 32, dt * 16, lane), pa, ot[dt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bool valid = qi < p.Tq;
      const float inv = 1.0f / l;
      store_row64(p.o + (long)b * p.o_sb + (long)qi * p.o_st + h * DH, ot, inv, g, valid);
      if (valid && g == 0 && p.lse != nullptr) p.lse[(long)hh * p.Tq + qi] = (mx + log2f(l)) * (1.0f / LOG2E);
    }
    lds_dma_wait_all();
    __syncthreads();
    cur ^= 1;
  }
}

// The dQ pass in the same persistent form (round 4): one 16-wave workgroup per CU walks (batch, head) pairs, wave w owns query
// rows 16 w .. 16 w + 15 of every head, K / V of the NEXT head stream into the second LDS buffer behind the current head's
// work.  In the round-4 stream plan the persistent dK / dV pass is worth 0.12 ms of the ViT step against the one-shot form
// (it was neutral in round 2); this is the dQ pass given the same treatment.  Option "attn_persistent" bit 1.
template <int NB>
__global__ __launch_bounds__(PERS_WAVES * 64, 4) void attn_bwd_dq_pers_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 256 * 128, BUF = 2 * TILE;  // K tile + V tile, 256-row slots (rows beyond Tk are zero)
  static_assert(NB * 32 <= 256, "K / V of one head fit the 256-row tile slot");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int heads = p.B * p.H;
  const float sl2 = p.scale * LOG2E;

  auto issue = [&](int hh, char* buf, bool live) {
    const int b = hh / p.H, h = hh - b * p.H;
    dma_tile_pers(buf, p.k + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, live, wave, lane);
    dma_tile_pers(buf + TILE, p.v + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, live, wave, lane);
  };

  int hh = blockIdx.x;
  int cur = 0;
  issue(hh, smem, hh < heads);
  lds_dma_wait_all();
  __syncthreads();
  const int row0 = wave * 16;
  const int qi = row0 + i;
  const bool qvalid = qi < p.Tq;
  for (; hh < heads; hh += gridDim.x) {
    const int b = hh / p.H, h = hh - b * p.H;
    const char* Ks = smem + cur * BUF;
    const char* Vs = Ks + TILE;
    const bf16_t* qb = p.q + (long)b * p.q_sb + h * DH;
    const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * DH;
    const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * DH;
    // this head's own fragments first: older than the DMA below in the vmcnt order
    const bf16x8 qf0 = frag_global(qb, p.q_st, row0, p.Tq, 0, lane);
    const bf16x8 qf1 = frag_global(qb, p.q_st, row0, p.Tq, 1, lane);
    const bf16x8 dof0 = frag_global(dob, p.o_st, row0, p.Tq, 0, lane);
    const bf16x8 dof1 = frag_global(dob, p.o_st, row0, p.Tq, 1, lane);
    const bf16x8 of0 = frag_global(ob, p.o_st, row0, p.Tq, 0, lane);
    const bf16x8 of1 = frag_global(ob, p.o_st, row0, p.Tq, 1, lane);
    const long stat = (long)hh * p.Tq + (qvalid ? qi : 0);
    const float lse_raw = p.lse[stat];
    const int nxt = hh + gridDim.x;
    issue(nxt < heads ? nxt : hh, smem + (cur ^ 1) * BUF, nxt < heads);

    float delta;
    {
      float sacc = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sacc += bf16_to_f32((bf16_t)dof0[e]) * bf16_to_f32((bf16_t)of0[e]);
        sacc += bf16_to_f32((bf16_t)dof1[e]) * bf16_to_f32((bf16_t)of1[e]);
      }
      delta = group_sum(sacc);
    }
    if (qvalid && g == 0) p.delta[stat] = delta;
    const float lse2 = qvalid ? lse_raw * LOG2E : INFINITY;  // +inf -> p = 0 for padded rows

    f32x4 dqt[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (row0 < p.Tq) {
      for (int a = 0; a < NB; ++a) {
        f32x4 ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int jt = 2 * a + t;
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 0, lane), qf0, sc, 0, 0, 0);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks, jt * 16, 1, lane), qf1, sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs, jt * 16, 0, lane), dof0, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs, jt * 16, 1, lane), dof1, dp, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) ds[t][r] = __builtin_amdgcn_exp2f(sc[r] * sl2 - lse2) * (dp[r] - delta);
        }
        const bf16x8 dsp = pack8(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_cols(Ks, a * 32, dt * 16, lane), dsp, dqt[dt], 0, 0, 0);
      }
    }
    store_row64(p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * DH, dqt, p.scale, g, qvalid);
    lds_dma_wait_all();
    __syncthreads();  // the next head's K / V have landed (vmcnt 0) and nobody reads this head's buffer any more
    cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------------
// ONE-PASS backward for the ViT shape (round 6; PLAIN, self-attention, 128 < T <= 224, head_dim 64): dQ, dK and dV of a head from
// ONE evaluation of S and dP.  The two-pass form above recomputes S = Q K^T and dP = dO V^T in both passes (2 548 MFMAs per head at
// T = 197); the energy table of round 6 ranks attention first in joules per FLOP (2.4-2.9 J/TFLOP against 1.04-1.16 for the GEMMs,
// profiles/r06/energy_table.txt), which is what decided that this experiment gets built.
//   * a 16-wave workgroup owns a CU and walks (batch, head) pairs; Q, dO and K of the head sit in LDS (K / V fragments of a wave's own
//     key tile come from global memory into registers, as in the persistent dK / dV kernel);
//   * phase 1 (waves own KEY tiles): for every query tile of the current half — S, P = exp2(S - lse), dP, dS = P (dP - delta);
//     dV += P^T dO and dK += dS^T Q accumulate in registers over both halves; dS (bf16, the rounding the MFMAs of both passes of the
//     two-pass form use) goes to an LDS buffer [query][key];
//   * barrier; phase 2 (waves own (QUERY tile, half of the 64 columns)): dQ = dS K with the whole key range as the reduction, the
//     dS operand read back from LDS in exactly the k-slot order a packed register operand has; barrier;
//   * queries in two halves so that Q + dO + K + dS + statistics fit 160 KB (T <= 224: 147 KB); delta = rowsum(dO o O) is computed in
//     the prologue (four lanes per row) and also written to p.delta for callers that read it.
// 1 716 MFMAs per head (-33 %), one launch instead of two; what it gives up: the next head's operands no longer stream in behind
// the current head (the LDS is full), and two more barriers per head.
// ------------------------------------------------------------------------------------------------
// CAUSAL: key j attends only to queries i >= j (the CLIP text tower's triu(1) mask as a flag): P = 0 above the diagonal (a predicate on
// the probability; skipping the tiles above the diagonal costs more registers than it saves at T = 77), their zero dS is not read in
// phase 2.
template <int NB, int PW, bool CAUSAL = false>  // NB pairs of 16-row tiles (T <= 32 NB); PW waves per workgroup (>= 2 NB key tiles, >= 4 ceil(NB / 2) phase-2 items)
__global__ __launch_bounds__(PW * 64, 4) void attn_bwd_one_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(128))) char smem[];  // 128: the XOR-derived addresses below need whole tile rows
  constexpr int ROWS = NB * 32, TILE = ROWS * 128;
  constexpr int HA = (NB + 1) / 2;          // query pair-tiles per half
  constexpr int DSROW = ROWS * 2 + 16;      // bytes per dS row (ROWS keys, bf16) + 16: rows 4 apart start 16 banks apart
  static_assert(3 * TILE + 2 * ROWS * 4 + HA * 32 * DSROW <= 160 * 1024, "Q + dO + K + statistics + dS must fit the LDS");
  static_assert(2 * NB <= PW && 4 * HA <= PW && ROWS * 4 <= PW * 64, "a wave per key tile; a wave per (query tile, column half); four lanes per row");
  char* Qs = smem;
  char* dOs = Qs + TILE;
  char* Ks = dOs + TILE;
  float* lse_s = reinterpret_cast<float*>(Ks + TILE);
  float* delta_s = lse_s + ROWS;
  char* dSs = reinterpret_cast<char*>(delta_s + ROWS);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int heads = p.B * p.H;
  const float sl2 = p.scale * LOG2E;
  const int row0 = wave * 16;  // phase 1: this wave's key tile
  const int qt = wave >> 1, ch = wave & 1;  // phase 2: query tile of the half, column half
  const int kq = row0 + n - 4 * g;  // CAUSAL: key (row0 + n) is masked for query (16 it + 4 g + r) when kq > 16 it + r

  // Q and dO of head `hx` into the LDS (DMA: no registers) ...
  auto stage_q = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    dma_tile(Qs, p.q + (long)b * p.q_sb + h * DH, p.q_st, p.Tq, ROWS, wave, PW, lane);
    dma_tile(dOs, p.d_o + (long)b * p.o_sb + h * DH, p.o_st, p.Tq, ROWS, wave, PW, lane);
  };
  // ... and its statistics (log-sum-exp in the log2 domain, delta = rowsum(dO o O)): four lanes per row, 16 columns each
  auto stage_stats = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * DH;
    const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * DH;
    const int t = (int)threadIdx.x >> 2, part = (int)threadIdx.x & 3;
    float sacc = 0.f;
    if (t < p.Tq) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dob + (long)t * p.o_st + part * 16 + j * 8);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(ob + (long)t * p.o_st + part * 16 + j * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc += bf16_to_f32((bf16_t)a[e]) * bf16_to_f32((bf16_t)c[e]);
      }
    }
    sacc += __shfl_xor(sacc, 1, 64);
    sacc += __shfl_xor(sacc, 2, 64);
    if (part == 0 && t < ROWS) {
      const bool ok = t < p.Tq;
      lse_s[t] = ok ? p.lse[(long)hx * p.Tq + t] * LOG2E : INFINITY;  // +inf -> p = 0 for padded query rows
      delta_s[t] = ok ? sacc : 0.f;
      if (ok && p.delta != nullptr) p.delta[(long)hx * p.Tq + t] = sacc;
    }
  };
  auto stage_k = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    dma_tile(Ks, p.k + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, ROWS, wave, PW, lane);
  };

  // Lane-constant parts of the LDS addresses of both phases, kept as LDS-space pointers and made opaque once (see attn_bwd_dq3_kernel):
  // through generic pointers every access paid a `v_add_u32 v, 0, v` for the link-time base, and the statistics / dS regions sit beyond
  // the 64 KB an instruction's offset field reaches, so each of their twelve accesses per block paid an add of its own — 28 of the 65
  // VALU instructions of the phase-1 block.  With the region's base inside the pointer the rest are immediates: 8 adds per block.
  typedef __attribute__((address_space(3))) const char* lds_cp;
  typedef __attribute__((address_space(3))) char* lds_p;
  typedef __attribute__((address_space(3))) const bf16x8* lds_v8;
  typedef __attribute__((address_space(3))) const f32x4* lds_f4;
  typedef __attribute__((address_space(3))) s16x4* lds_s4;
  lds_cp qr0, qc0, sp;
  lds_p dsw;
  {
    // (the tiles start on 128-byte boundaries — the first one at the dynamic LDS base — so the other reduction half of a row fragment and
    // the other column tiles are one XOR of bits 5..6 away: slot ^ key with the key in bits 1..2 of the slot)
    const lds_cp q3 = (lds_cp)LDS_PTR(Qs);
    qr0 = q3 + n * 128 + ((g ^ swz(n)) << 4);
    qc0 = q3 + tile_off(4 * g + (n >> 2), 4 * (n & 3));
    sp = (lds_cp)LDS_PTR(lse_s) + 16 * g;
    dsw = (lds_p)LDS_PTR(dSs) + (row0 + n) * 2 + 4 * g * DSROW;
#define CFHIP_OPAQUE_LDS(ptr, T) { unsigned u_ = (unsigned)(uintptr_t)(ptr); asm volatile("" : "+v"(u_)); (ptr) = (T)(uintptr_t)u_; }
#define CFHIP_LDS_XOR(ptr, bits) ((lds_cp)(uintptr_t)((unsigned)(uintptr_t)(ptr) ^ (unsigned)(bits)))
    CFHIP_OPAQUE_LDS(qr0, lds_cp)
    CFHIP_OPAQUE_LDS(qc0, lds_cp)
    CFHIP_OPAQUE_LDS(sp, lds_cp)
    CFHIP_OPAQUE_LDS(dsw, lds_p)
  }
  int hh = blockIdx.x;
  if (hh < heads) {
    stage_q(hh);
    stage_k(hh);
    stage_stats(hh);
  }
  for (; hh < heads; hh += gridDim.x) {
    const int b = hh / p.H, h = hh - b * p.H;
    const bf16_t* kb = p.k + (long)b * p.kv_sb + h * DH;
    const bf16_t* vb = p.v + (long)b * p.kv_sb + h * DH;
    // this wave's key tile: K / V fragments in registers (rows beyond Tk are zero)
    const bf16x8 kf0 = frag_global(kb, p.kv_st, row0, p.Tk, 0, lane);
    const bf16x8 kf1 = frag_global(kb, p.kv_st, row0, p.Tk, 1, lane);
    const bf16x8 vf0 = frag_global(vb, p.kv_st, row0, p.Tk, 0, lane);
    const bf16x8 vf1 = frag_global(vb, p.kv_st, row0, p.Tk, 1, lane);
    lds_dma_wait_all();
    __syncthreads();
    const int nxt = hh + gridDim.x;

    f32x4 dkt[4], dvt[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
      const int a0 = hf * HA, a1 = hf == 0 ? HA : NB;
      // ---- phase 1: key tile x the query tiles of this half
      if (row0 < ROWS) {
        for (int a = a0; a < a1; ++a) {
          f32x4 pp[2], ds[2];
          const int aoff = a * (32 * 128);
          const lds_cp spa = sp + a * 128;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int it = 2 * a + t;
            f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            const lds_cp r0 = qr0 + aoff + t * 2048, r1 = CFHIP_LDS_XOR(qr0 + aoff, 64) + t * 2048;  // rows 16 it + n of Q; dO sits TILE bytes behind
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(lds_v8)r0, kf0, sc, 0, 0, 0);
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(lds_v8)r1, kf1, sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(lds_v8)(r0 + TILE), vf0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(lds_v8)(r1 + TILE), vf1, dp, 0, 0, 0);
            const f32x4 l4 = *(lds_f4)(spa + t * 64);
            const f32x4 d4 = *(lds_f4)(spa + t * 64 + ROWS * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pr = __builtin_amdgcn_exp2f(sc[r] * sl2 - l4[r]);
              if (CAUSAL && kq > it * 16 + r) pr = 0.f;
              pp[t][r] = pr;
              ds[t][r] = pr * (dp[r] - d4[r]);
            }
          }
          const bf16x8 ppk = pack8(pp[0], pp[1]);
          const bf16x8 dsk = pack8(ds[0], ds[1]);
          // dS[query 16 (2a + t) + 4g + r][key row0 + n] -> LDS [query][key]
          {
            const lds_p col = dsw + (a - a0) * (32 * DSROW);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(__attribute__((address_space(3))) short*)(col + (t * 16 + r) * DSROW) = dsk[4 * t + r];
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const lds_cp cp = CFHIP_LDS_XOR(qc0 + aoff, dt * 32);  // rows 32 a + 4 g + (n >> 2) of Q (+16: 2 048 bytes on); dO TILE bytes behind
            const s16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)cp);
            const s16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + 2048));
            const s16x4 olo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + TILE));
            const s16x4 ohi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + TILE + 2048));
            const bf16x8 dof = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
            const bf16x8 qf = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
            dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, ppk, dvt[dt], 0, 0, 0);
            dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dsk, dkt[dt], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      // Q and dO of this head are done with after the second half's phase 1: the NEXT head's stream in behind the last dQ phase
      // (which reads K and dS only); its statistics follow at the end of the head, when the dK / dV accumulators are dead
      if (hf == 1 && nxt < heads) stage_q(nxt);
      // ---- phase 2: dQ of the query tiles of this half, 32 columns per wave, reduction over every key
      if (qt < 2 * (a1 - a0)) {
        f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
        const int a_end = CAUSAL ? min(NB, ((2 * a0 + qt) * 16 + 15) / 32 + 1) : NB;  // keys beyond the tile's last query: dS = 0
        // (phase 2's three pointers are rebuilt per half: kept across phase 1 they push the CAUSAL forms into scratch)
        lds_cp kc2[2], drow = (lds_cp)LDS_PTR(dSs) + (qt * 16 + n) * DSROW + 8 * g;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          kc2[j] = (lds_cp)LDS_PTR(Ks) + tile_off(4 * g + (n >> 2), (2 * ch + j) * 16 + 4 * (n & 3));
          CFHIP_OPAQUE_LDS(kc2[j], lds_cp)
        }
        CFHIP_OPAQUE_LDS(drow, lds_cp)
#pragma unroll 1
        for (int a = 0; a < a_end; ++a) {
          typedef __attribute__((address_space(3))) const s16x4* lds_cs4;
          const s16x4 lo = *(lds_cs4)(drow + a * 64);
          const s16x4 hi = *(lds_cs4)(drow + a * 64 + 32);
          bf16x8 dsp;
          dsp[0] = lo[0]; dsp[1] = lo[1]; dsp[2] = lo[2]; dsp[3] = lo[3];
          dsp[4] = hi[0]; dsp[5] = hi[1]; dsp[6] = hi[2]; dsp[7] = hi[3];
          const lds_cp c0 = kc2[0] + a * (32 * 128), c1 = kc2[1] + a * (32 * 128);
          const s16x4 k0l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)c0), k0h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(c0 + 2048));
          const s16x4 k1l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)c1), k1h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(c1 + 2048));
          const bf16x8 kf0c = {k0l[0], k0l[1], k0l[2], k0l[3], k0h[0], k0h[1], k0h[2], k0h[3]};
          const bf16x8 kf1c = {k1l[0], k1l[1], k1l[2], k1l[3], k1h[0], k1h[1], k1h[2], k1h[3]};
          dq0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf0c, dsp, dq0, 0, 0, 0);
          dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf1c, dsp, dq1, 0, 0, 0);
        }
        const int qi = (2 * a0 + qt) * 16 + n;
        store_row32(p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * DH, dq0, dq1, p.scale, g, qi < p.Tq, ch);
      }
      __syncthreads();  // the dS buffer (and, after the second half, K) may be overwritten
    }
    if (nxt < heads) stage_k(nxt);
    {
      const int kj = row0 + n;
      const bool valid = kj < p.Tk;
      store_row64(p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dkt, p.scale, g, valid);
      store_row64(p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dvt, 1.0f, g, valid);
    }
    if (nxt < heads) stage_stats(nxt);
  }
}

// ------------------------------------------------------------------------------------------------
// The one-pass backward with TWO key tiles per wave (round 6, late; non-causal, 128 < T <= 224).  The 16-wave form above sits at 13 %
// of the matrix pipe and 26 % of the LDS: at 128 registers per wave hipcc issues the LDS reads of a block two at a time and waits for
// each pair (twelve exposed LDS latencies per 16 MFMAs).  Here a workgroup is 8 waves at 256 registers: a wave owns 32 keys, so every
// Q / dO fragment read from LDS (row form for S and dP, column form for dV and dK) feeds two MFMAs — half the LDS reads per FLOP,
// 32 MFMAs per block — and all fragments of a block are in flight before the first MFMA needs one.  Phase 2: a wave per 16-query tile
// of the half, all 64 columns (four accumulators per dS operand read instead of two).  Same LDS layout, same arithmetic per element
// (the reduction order over queries / keys inside an accumulator is unchanged): results are bit-identical to the 16-wave form.
// ------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(512, 2) void attn_bwd_one2_kernel(AttnParams p) {
  extern __shared__ __attribute__((aligned(128))) char smem[];  // 128: the XOR-derived addresses below need whole tile rows
  constexpr int PW = 8;
  constexpr int ROWS = NB * 32, TILE = ROWS * 128;
  constexpr int HA = (NB + 1) / 2;          // query pair-tiles per half
  constexpr int DSROW = ROWS * 2 + 16;      // bytes per dS row (ROWS keys, bf16) + 16: rows 4 apart start 16 banks apart
  static_assert(3 * TILE + 2 * ROWS * 4 + HA * 32 * DSROW <= 160 * 1024, "Q + dO + K + statistics + dS must fit the LDS");
  static_assert(NB <= PW && 2 * HA <= PW, "a wave per 32 keys; a wave per query tile of a half");
  char* Qs = smem;
  char* dOs = Qs + TILE;
  char* Ks = dOs + TILE;
  float* lse_s = reinterpret_cast<float*>(Ks + TILE);
  float* delta_s = lse_s + ROWS;
  char* dSs = reinterpret_cast<char*>(delta_s + ROWS);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int heads = p.B * p.H;
  const float sl2 = p.scale * LOG2E;
  const int row0 = wave * 32;  // phase 1: this wave's keys
  const int qt = wave;         // phase 2: query tile of the half

  auto stage_q = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    dma_tile(Qs, p.q + (long)b * p.q_sb + h * DH, p.q_st, p.Tq, ROWS, wave, PW, lane);
    dma_tile(dOs, p.d_o + (long)b * p.o_sb + h * DH, p.o_st, p.Tq, ROWS, wave, PW, lane);
  };
  // statistics of a head (log-sum-exp in the log2 domain, delta = rowsum(dO o O)): four lanes per row, two sweeps of 128 rows.  The
  // global loads (load_stats) are issued a whole dQ phase before their values are folded and written to the LDS (finish_stats).
  bf16x8 st_a[2][2], st_c[2][2];
  float st_l[2];
  auto load_stats = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * DH;
    const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * DH;
    const int part = (int)threadIdx.x & 3;
#pragma unroll
    for (int sw = 0; sw < 2; ++sw) {
      const int t = ((int)threadIdx.x >> 2) + sw * (PW * 16);
      const bool ok = t < p.Tq && !ATTN_ABL(32);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        st_a[sw][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        st_c[sw][j] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) {
          st_a[sw][j] = *reinterpret_cast<const bf16x8*>(dob + (long)t * p.o_st + part * 16 + j * 8);
          st_c[sw][j] = *reinterpret_cast<const bf16x8*>(ob + (long)t * p.o_st + part * 16 + j * 8);
        }
      }
      st_l[sw] = (ok && part == 0) ? p.lse[(long)hx * p.Tq + t] * LOG2E : INFINITY;  // +inf -> p = 0 for padded query rows
    }
  };
  auto finish_stats = [&](int hx) {
    const int part = (int)threadIdx.x & 3;
#pragma unroll
    for (int sw = 0; sw < 2; ++sw) {
      const int t = ((int)threadIdx.x >> 2) + sw * (PW * 16);
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) sacc += bf16_to_f32((bf16_t)st_a[sw][j][e]) * bf16_to_f32((bf16_t)st_c[sw][j][e]);
      sacc += __shfl_xor(sacc, 1, 64);
      sacc += __shfl_xor(sacc, 2, 64);
      if (part == 0 && t < ROWS) {
        lse_s[t] = st_l[sw];
        delta_s[t] = sacc;  // (rows beyond Tq: zero operands)
        if (t < p.Tq && p.delta != nullptr) p.delta[(long)hx * p.Tq + t] = sacc;
      }
    }
  };
  auto stage_k = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    dma_tile(Ks, p.k + (long)b * p.kv_sb + h * DH, p.kv_st, p.Tk, ROWS, wave, PW, lane);
  };

  typedef __attribute__((address_space(3))) const char* lds_cp;
  typedef __attribute__((address_space(3))) char* lds_p;
  typedef __attribute__((address_space(3))) const bf16x8* lds_v8;
  typedef __attribute__((address_space(3))) const f32x4* lds_f4;
  typedef __attribute__((address_space(3))) s16x4* lds_s4;
  typedef __attribute__((address_space(3))) const s16x4* lds_cs4;
  lds_cp qr0, qc0, sp, drow, kc0;
  lds_p dsw;
  {
    const lds_cp q3 = (lds_cp)LDS_PTR(Qs);
    qr0 = q3 + n * 128 + ((g ^ swz(n)) << 4);
    qc0 = q3 + tile_off(4 * g + (n >> 2), 4 * (n & 3));
    kc0 = qc0 + 2 * TILE;
    sp = (lds_cp)LDS_PTR(lse_s) + 16 * g;
    dsw = (lds_p)LDS_PTR(dSs) + (row0 + n) * 2 + 4 * g * DSROW;
    drow = (lds_cp)LDS_PTR(dSs) + (qt * 16 + n) * DSROW + 8 * g;
    CFHIP_OPAQUE_LDS(qr0, lds_cp)
    CFHIP_OPAQUE_LDS(qc0, lds_cp)
    CFHIP_OPAQUE_LDS(kc0, lds_cp)
    CFHIP_OPAQUE_LDS(sp, lds_cp)
    CFHIP_OPAQUE_LDS(dsw, lds_p)
    CFHIP_OPAQUE_LDS(drow, lds_cp)
  }
  // K / V fragments of this wave's two key tiles (rows beyond Tk are zero): loaded for the NEXT head behind the last phase 1 of a head
  bf16x8 kf[2][2], vf[2][2];
  auto load_kv = [&](int hx) {
    const int b = hx / p.H, h = hx - b * p.H;
    const bf16_t* kb = p.k + (long)b * p.kv_sb + h * DH;
    const bf16_t* vb = p.v + (long)b * p.kv_sb + h * DH;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[kt][ks] = frag_global(kb, p.kv_st, row0 + 16 * kt, ATTN_ABL(128) ? 0 : p.Tk, ks, lane);
        vf[kt][ks] = frag_global(vb, p.kv_st, row0 + 16 * kt, ATTN_ABL(128) ? 0 : p.Tk, ks, lane);
      }
  };
  int hh = blockIdx.x;
  if (hh < heads) {
    stage_q(hh);
    load_kv(hh);
    load_stats(hh);
    finish_stats(hh);
  }
  for (; hh < heads; hh += gridDim.x) {
    const int b = hh / p.H, h = hh - b * p.H;
    const int nxt = hh + gridDim.x;
    // Q, dO and the statistics of this head are in the LDS (staged behind the previous head's last phase 1); the previous head's last
    // dQ phase is over, so its K tile may go: this head's K streams in under phase 1, which takes K from registers
    lds_dma_wait_all();
    __syncthreads();
    if (!ATTN_ABL(256)) stage_k(hh);

    f32x4 dkt[2][4], dvt[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dkt[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[kt][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int a0 = hf * HA, a1 = hf == 0 ? HA : NB;
      // ---- phase 1: 32 keys x the query tiles of this half
      if (row0 < ROWS && !ATTN_ABL(8)) {
#pragma unroll 1
        for (int a = a0; a < a1; ++a) {
          const int aoff = a * (32 * 128);
          const lds_cp spa = sp + a * 128;
          bf16x8 qf[2][2], dof[2][2];
          f32x4 l4[2], d4[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const lds_cp r0 = qr0 + aoff + t * 2048, r1 = CFHIP_LDS_XOR(qr0 + aoff, 64) + t * 2048;  // rows 32 a + 16 t + n of Q; dO TILE bytes behind
            qf[t][0] = *(lds_v8)r0;
            qf[t][1] = *(lds_v8)r1;
            dof[t][0] = *(lds_v8)(r0 + TILE);
            dof[t][1] = *(lds_v8)(r1 + TILE);
            l4[t] = *(lds_f4)(spa + t * 64);
            d4[t] = *(lds_f4)(spa + t * 64 + ROWS * 4);
          }
          bf16x8 ppk[2], dsk[2];
          const lds_p col = dsw + (a - a0) * (32 * DSROW);
#pragma unroll
          for (int kt = 0; kt < 2; ++kt) {
            f32x4 pp[2], ds[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
              sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[t][0], kf[kt][0], sc, 0, 0, 0);
              sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[t][1], kf[kt][1], sc, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[t][0], vf[kt][0], dp, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[t][1], vf[kt][1], dp, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float pr = __builtin_amdgcn_exp2f(sc[r] * sl2 - l4[t][r]);
                pp[t][r] = pr;
                ds[t][r] = pr * (dp[r] - d4[t][r]);
              }
            }
            ppk[kt] = pack8(pp[0], pp[1]);
            dsk[kt] = pack8(ds[0], ds[1]);
            // dS[query 16 (2 (a - a0) + t) + 4g + r][key row0 + 16 kt + n] -> LDS [query][key]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                *(__attribute__((address_space(3))) short*)(col + kt * 32 + (t * 16 + r) * DSROW) = dsk[kt][4 * t + r];
          }
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const lds_cp cp = CFHIP_LDS_XOR(qc0 + aoff, dt * 32);  // rows 32 a + 4 g + (n >> 2) of Q (+16: 2 048 bytes on); dO TILE bytes behind
            const s16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)cp);
            const s16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + 2048));
            const s16x4 olo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + TILE));
            const s16x4 ohi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + TILE + 2048));
            const bf16x8 doc = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
            const bf16x8 qcf = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              dvt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(doc, ppk[kt], dvt[kt][dt], 0, 0, 0);
              dkt[kt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qcf, dsk[kt], dkt[kt][dt], 0, 0, 0);
            }
          }
        }
      }
      if (hf == 0) {
        lds_dma_wait_all();  // this head's K tile (issued a phase ago)
      } else {
        // dK / dV are complete: out now, behind the dQ phase, not at the end of the head
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const int kj = row0 + 16 * kt + n;
          const bool valid = kj < p.Tk && !ATTN_ABL(64);
          store_row64(p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dkt[kt], p.scale, g, valid);
          store_row64(p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * DH, dvt[kt], 1.0f, g, valid);
        }
      }
      __syncthreads();
      // Q, dO and the statistics of this head are done with after the second half's phase 1: the NEXT head's stream in behind the last
      // dQ phase — tiles by DMA, the wave's K / V fragments and the statistics' inputs into registers
      if (hf == 1 && nxt < heads) {
        if (!ATTN_ABL(256)) stage_q(nxt);
        load_kv(nxt);
        load_stats(nxt);
      }
      // ---- phase 2: dQ of the query tiles of this half, reduction over every key
      if (qt < 2 * (a1 - a0) && !ATTN_ABL(16)) {
        f32x4 dq[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int a = 0; a < NB; ++a) {
          const s16x4 lo = *(lds_cs4)(drow + a * 64);
          const s16x4 hi = *(lds_cs4)(drow + a * 64 + 32);
          const bf16x8 dsp = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const lds_cp cp = CFHIP_LDS_XOR(kc0 + a * (32 * 128), dt * 32);
            const s16x4 kl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)cp), kh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + 2048));
            const bf16x8 kcf = {kl[0], kl[1], kl[2], kl[3], kh[0], kh[1], kh[2], kh[3]};
            dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcf, dsp, dq[dt], 0, 0, 0);
          }
        }
        const int qi = (2 * a0 + qt) * 16 + n;
        store_row64(p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * DH, dq, p.scale, g, qi < p.Tq);
      }
      if (hf == 0) __syncthreads();  // the dS buffer may be overwritten (after the second half: the barrier at the top of the next head)
    }
    if (nxt < heads) finish_stats(nxt);
  }
}

#undef CFHIP_OPAQUE_LDS
#undef CFHIP_LDS_XOR

// ------------------------------------------------------------------------------------------------
// General form: any sequence length, head_dim = any multiple of 8 up to 192 (UNet cross-attention heads of 40 / 80 /
// 160 channels, convs / mixed_stacks/api.py:766-893).  The same three kernels with (1) an outer loop over CHUNKS of
// the operand that the short form keeps resident in LDS and (2) the head dimension handled as NH = ceil(dh / 64)
// zero-padded 64-column HALVES, each an ordinary swizzled [rows][64] tile (columns >= dh are range-checked to zero
// by the DMA / never loaded into fragments / never stored).  Workgroup = (128-row chunk of the walking operand,
// head, batch), one 16-row tile per wave.  Forward: online softmax across chunks (running max m, sum l, rescaled O);
// backward: p = exp2(s * scale - lse) with the saved lse, so chunks just accumulate.  The chunked operand is re-read
// once per 128-row workgroup, from L2.
// ------------------------------------------------------------------------------------------------
// Round 3 measured a TWO-slot chunk ring for head_dim <= 64 (chunk c + 1 streaming into the other slot while chunk c is
// computed, 128-row chunks, one barrier per chunk; the code path is still here: NSLOT = 2): UNet shape B1 H8 T65536 dh40 forward
// 12.47 -> 13.86 ms, dQ 12.84 -> 13.21, dK/dV after a dQ pass 17.5 -> 20.2; 256^2 step 611 -> 645 ms
// (profiles/r03/attn_long_bench_{r02kernels,two_slot_ring}.log).  The load was already covered by the second workgroup of
// the CU; what the ring costs is twice the online-softmax steps per key (128- instead of 256-key chunks) and every wave of
// the workgroup in lock-step on one barrier.  Kept: NSLOT = 1, 256-row chunks for head_dim <= 64.
template <int NH>
struct Gen {
  static constexpr int CH = NH == 1 ? 256 : 128;  // rows per staged chunk (LDS: 2 operands x NH halves x CH x 128 B per slot)
  static constexpr int CHB = CH / 32;
  static constexpr int HALF_BYTES = CH * 128;
  static constexpr int OPER_BYTES = NH * HALF_BYTES;
  static constexpr int SLOT_BYTES = 2 * OPER_BYTES;
  static constexpr int NSLOT = 1;
};

// one 64-column half of `rows_pad` rows -> swizzled [rows][64] tile; columns >= dh and rows >= rows_valid are zero.
// SKIP_PAD: the lanes whose 16-byte chunk lies beyond dh are switched off (EXEC) instead of fetching zeros, so the DMA never
// writes the pad columns of the tile: whatever the kernel put there once (zeros; attn_fwd2_kernel's column of ones) stays.
template <bool SKIP_PAD = false>
__device__ __forceinline__ void dma_half(char* tile, const bf16_t* base, long stride_t, int rows_valid, int rows_pad,
                                         int half, int dh, int wave, int nwaves, int lane) {
  long bytes = ((long)(rows_valid - 1) * stride_t + dh) * 2;
  if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base), 0, (int)bytes, 0x00020000);
  const int r8 = lane >> 3, slot = lane & 7;
  for (int inst = wave; inst < rows_pad / 8; inst += nwaves) {
    const int row = inst * 8 + r8;
    const int chunk = slot ^ swz(row);
    const bool ok = row < rows_valid && half * 64 + chunk * 8 < dh;
    const unsigned off = ok ? (unsigned)(row * stride_t * 2 + half * 128 + chunk * 16) : 0x80000000u;
    if (SKIP_PAD) {
      if (half * 64 + chunk * 8 < dh) lds_dma16(rsrc, tile + inst * 1024, off);  // (rows beyond rows_valid still fetch zeros)
    } else {
      lds_dma16(rsrc, tile + inst * 1024, off);
    }
  }
}
template <int NH, bool SKIP_PAD = false>
__device__ __forceinline__ void dma_oper(char* tiles, const bf16_t* base, long stride_t, int rows_valid, int dh,
                                         int wave, int nwaves, int lane) {
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
    dma_half<SKIP_PAD>(tiles + hf * Gen<NH>::HALF_BYTES, base, stride_t, rows_valid, Gen<NH>::CH, hf, dh, wave, nwaves, lane);
}

// row-operand fragment ks (32 columns) of the wave's own 16 rows straight from global memory, zero beyond dh
__device__ __forceinline__ bf16x8 frag_global_dh(const bf16_t* base, long stride_t, int row0, int rows_valid, int ks,
                                                 int lane, int dh) {
  const int row = row0 + (lane & 15);
  const int col = ks * 32 + (lane >> 4) * 8;
  bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < rows_valid && col < dh) r = *reinterpret_cast<const bf16x8*>(base + (long)row * stride_t + col);
  return r;
}

// (DROP: the Philox state and the keep bytes need registers beyond the 128 of four waves per SIMD — two per SIMD, no scratch)
template <int NH, bool PLAIN, bool DROP = false>
__global__ __launch_bounds__(512, (NH == 1 && !DROP) ? 4 : 2) void attn_gen_fwd_kernel(AttnParams p) {
  using G = Gen<NH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + G::OPER_BYTES;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * dh;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * dh;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * dh;
  const int row0 = (blockIdx.x * nwaves + wave) * 16;
  const bool active = row0 < p.Tq;  // inactive waves still stage and hit the barriers
  const int qi = row0 + i;
  bf16x8 qf[2 * NH];
#pragma unroll
  for (int ks = 0; ks < 2 * NH; ++ks) qf[ks] = frag_global_dh(qb, p.q_st, row0, active ? p.Tq : 0, ks, lane, dh);
  const float sl2 = p.scale * LOG2E;
  float m = -INFINITY, l = 0.f;
  f32x4 ot[4 * NH];
#pragma unroll
  for (int dt = 0; dt < 4 * NH; ++dt) ot[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int k0, char* slot) {
    const int rows_ = min(G::CH, p.Tk - k0);
    int ln = lane;  // opaque copy: the per-lane DMA offsets are recomputed per chunk (a few VALU) instead of being hoisted out
    asm volatile("" : "+v"(ln));  // of the chunk loop and spilled around it
    dma_oper<NH>(slot, kb + (long)k0 * p.kv_st, p.kv_st, rows_, dh, wave, nwaves, ln);
    dma_oper<NH>(slot + G::OPER_BYTES, vb + (long)k0 * p.kv_st, p.kv_st, rows_, dh, wave, nwaves, ln);
  };
  if (G::NSLOT == 2) issue(0, smem);
  for (int kv0 = 0, c = 0; kv0 < p.Tk; kv0 += G::CH, ++c) {
    if (G::NSLOT == 2) {
      lds_dma_wait_all();
      __syncthreads();  // chunk c has landed everywhere; every wave is done with chunk c - 1 (the other slot)
      if (kv0 + G::CH < p.Tk) issue(kv0 + G::CH, smem + ((c + 1) & 1) * G::SLOT_BYTES);
      Ks = smem + (c & 1) * G::SLOT_BYTES;
      Vs = Ks + G::OPER_BYTES;
    } else {
      __syncthreads();  // every wave is done with the previous chunk
      issue(kv0, smem);
      lds_dma_wait_all();
      __syncthreads();
    }
    if (!active) continue;
    if (p.causal && kv0 > row0 + 15) continue;  // chunk entirely in the future of this tile (wave-uniform)
    f32x4 st[2 * G::CHB];
#pragma unroll
    for (int jt = 0; jt < 2 * G::CHB; ++jt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2 * NH; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks + (ks >> 1) * G::HALF_BYTES, jt * 16, ks & 1, lane),
                                                      qf[ks], acc, 0, 0, 0);
      st[jt] = acc;
      if (jt & 1) __builtin_amdgcn_sched_barrier(0);
    }
    float mx = -INFINITY;
    const bool tail = kv0 + G::CH > p.Tk;  // wave-uniform: only the last chunk needs the j < Tk test
#pragma unroll
    for (int jt = 0; jt < 2 * G::CHB; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = kv0 + jt * 16 + 4 * g + r;
        float x = st[jt][r] * sl2;
        if (PLAIN) {
          if (tail) x = j < p.Tk ? x : -INFINITY;
        } else {
          x = keep_at(p, b, h, qi < p.Tq ? qi : p.Tq - 1, j) ? x : -INFINITY;
        }
        st[jt][r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = group_max(mx);
    const float m_new = fmaxf(m, mx);
    // rows whose every position so far is masked (m_new = -inf) must not produce NaN from (-inf) - (-inf): use 0
    // as the reference point for them (all their terms are exp2(-inf) = 0 anyway); no divergent branch around MFMAs
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m - m_use);  // m = -inf on the first chunk: 0
    float ls = 0.f;
#pragma unroll
    for (int jt = 0; jt < 2 * G::CHB; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(st[jt][r] - m_use);
        st[jt][r] = e;
        ls += e;
      }
    }
    l = l * alpha + group_sum(ls);
    m = m_new;
    if (DROP) {  // the row sum above is the softmax denominator: dropout acts on the normalised probabilities
#pragma unroll
      for (int jt = 0; jt < 2 * G::CHB; ++jt) {
        const unsigned w = pick_word(drop_block(p, b, h, qi >> 2, (kv0 >> 2) + jt * 4 + g), qi);
#pragma unroll
        for (int r = 0; r < 4; ++r) st[jt][r] = ((w >> (8 * r)) & 255u) >= p.drop_thresh ? st[jt][r] : 0.f;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4 * NH; ++dt) ot[dt] *= alpha;
#pragma unroll
    for (int a = 0; a < G::CHB; ++a) {
      const bf16x8 pa = pack8(st[2 * a], st[2 * a + 1]);
#pragma unroll
      for (int dt = 0; dt < 4 * NH; ++dt)
        ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            frag_cols(Vs + (dt >> 2) * G::HALF_BYTES, a * 32, (dt & 3) * 16, lane), pa, ot[dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (active && qi < p.Tq) {
    const float inv = (DROP ? p.drop_scale : 1.0f) / l;
    bf16_t* orow = p.o + (long)b * p.o_sb + (long)qi * p.o_st + h * dh;
#pragma unroll
    for (int dt = 0; dt < 4 * NH; ++dt) {
      const int col = dt * 16 + 4 * g;
      if (col >= dh) continue;  // dh % 8 == 0: a 4-column group is entirely inside or outside
      const f32x4 v = ot[dt] * inv;
      *reinterpret_cast<u32x2*>(orow + col) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
    if (g == 0 && p.lse != nullptr) p.lse[((long)b * p.H + h) * p.Tq + qi] = (m + log2f(l)) * (1.0f / LOG2E);
  }
}

template <int NH, bool PLAIN, bool DROP = false>
__global__ __launch_bounds__(512, NH == 1 ? 4 : 2) void attn_gen_bwd_dq_kernel(AttnParams p) {
  using G = Gen<NH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + G::OPER_BYTES;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int row0 = (blockIdx.x * nwaves + wave) * 16;
  const bool active = row0 < p.Tq;
  const int qi = row0 + i;
  const bool qvalid = active && qi < p.Tq;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * dh;
  const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * dh;
  const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * dh;
  bf16x8 qf[2 * NH], dof[2 * NH];
  float sacc = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2 * NH; ++ks) {
    qf[ks] = frag_global_dh(qb, p.q_st, row0, active ? p.Tq : 0, ks, lane, dh);
    dof[ks] = frag_global_dh(dob, p.o_st, row0, active ? p.Tq : 0, ks, lane, dh);
    const bf16x8 of = frag_global_dh(ob, p.o_st, row0, active ? p.Tq : 0, ks, lane, dh);
#pragma unroll
    for (int e = 0; e < 8; ++e) sacc += bf16_to_f32((bf16_t)dof[ks][e]) * bf16_to_f32((bf16_t)of[e]);
  }
  const float delta = group_sum(sacc);
  const long stat = ((long)b * p.H + h) * p.Tq + qi;
  if (qvalid && g == 0) p.delta[stat] = delta;
  const float lse2 = qvalid ? p.lse[stat] * LOG2E : INFINITY;
  const float sl2 = p.scale * LOG2E;
  f32x4 dqt[4 * NH];
#pragma unroll
  for (int dt = 0; dt < 4 * NH; ++dt) dqt[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int k0, char* slot) {
    const int rows_ = min(G::CH, p.Tk - k0);
    int ln = lane;  // opaque copy: the per-lane DMA offsets are recomputed per chunk (a few VALU) instead of being hoisted out
    asm volatile("" : "+v"(ln));  // of the chunk loop and spilled around it
    dma_oper<NH>(slot, p.k + (long)b * p.kv_sb + (long)k0 * p.kv_st + h * dh, p.kv_st, rows_, dh, wave, nwaves, ln);
    dma_oper<NH>(slot + G::OPER_BYTES, p.v + (long)b * p.kv_sb + (long)k0 * p.kv_st + h * dh, p.kv_st, rows_, dh, wave, nwaves, ln);
  };
  if (G::NSLOT == 2) issue(0, smem);
  for (int kv0 = 0, c = 0; kv0 < p.Tk; kv0 += G::CH, ++c) {
    const int rows = min(G::CH, p.Tk - kv0);
    if (G::NSLOT == 2) {
      lds_dma_wait_all();
      __syncthreads();  // chunk c has landed everywhere; every wave is done with chunk c - 1 (the other slot)
      if (kv0 + G::CH < p.Tk) issue(kv0 + G::CH, smem + ((c + 1) & 1) * G::SLOT_BYTES);
      Ks = smem + (c & 1) * G::SLOT_BYTES;
      Vs = Ks + G::OPER_BYTES;
    } else {
      __syncthreads();
      issue(kv0, smem);
      lds_dma_wait_all();
      __syncthreads();
    }
    if (!active) continue;
    if (p.causal && kv0 > row0 + 15) continue;
    const int nbl = (rows + 31) / 32;
    for (int a = 0; a < nbl; ++a) {
      f32x4 ds[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int jt = 2 * a + t;
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2 * NH; ++ks) {
          sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Ks + (ks >> 1) * G::HALF_BYTES, jt * 16, ks & 1, lane),
                                                       qf[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Vs + (ks >> 1) * G::HALF_BYTES, jt * 16, ks & 1, lane),
                                                       dof[ks], dp, 0, 0, 0);
        }
        unsigned w = 0u;
        if (DROP) w = pick_word(drop_block(p, b, h, qi >> 2, (kv0 >> 2) + jt * 4 + g), qi);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pr = __builtin_amdgcn_exp2f(sc[r] * sl2 - lse2);
          if (!PLAIN) pr = keep_at(p, b, h, qvalid ? qi : p.Tq - 1, kv0 + jt * 16 + 4 * g + r) ? pr : 0.f;
          float dpr = dp[r];  // d loss / d (dropped probability)
          if (DROP) dpr = ((w >> (8 * r)) & 255u) >= p.drop_thresh ? dpr * p.drop_scale : 0.f;
          ds[t][r] = pr * (dpr - delta);
        }
      }
      const bf16x8 dsp = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4 * NH; ++dt) {
        dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            frag_cols(Ks + (dt >> 2) * G::HALF_BYTES, a * 32, (dt & 3) * 16, lane), dsp, dqt[dt], 0, 0, 0);
      }
    }
  }
  if (qvalid) {
    bf16_t* dqrow = p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * dh;
#pragma unroll
    for (int dt = 0; dt < 4 * NH; ++dt) {
      const int col = dt * 16 + 4 * g;
      if (col >= dh) continue;
      const f32x4 v = dqt[dt] * p.scale;
      *reinterpret_cast<u32x2*>(dqrow + col) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
  }
}

template <int NH, bool PLAIN, bool DROP = false>
__global__ __launch_bounds__(512, NH == 1 ? 4 : 2) void attn_gen_bwd_dkv_kernel(AttnParams p) {
  using G = Gen<NH>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* dOs = smem + G::OPER_BYTES;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * G::OPER_BYTES);
  float* delta_s = lse_s + G::CH;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwaves = blockDim.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int row0 = (blockIdx.x * nwaves + wave) * 16;
  const bool active = row0 < p.Tk;
  const int kj = row0 + n;
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * dh;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * dh;
  bf16x8 kf[2 * NH], vf[2 * NH];
#pragma unroll
  for (int ks = 0; ks < 2 * NH; ++ks) {
    kf[ks] = frag_global_dh(kb, p.kv_st, row0, active ? p.Tk : 0, ks, lane, dh);
    vf[ks] = frag_global_dh(vb, p.kv_st, row0, active ? p.Tk : 0, ks, lane, dh);
  }
  const float sl2 = p.scale * LOG2E;
  f32x4 dkt[4 * NH], dvt[4 * NH];
#pragma unroll
  for (int dt = 0; dt < 4 * NH; ++dt) { dkt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int dslots = dh / 8;  // 16-byte slots of one row of dO / O

  // lse (log2 domain) and delta_i = sum_d dO[i][d] * O[i][d] of a chunk's rows: thread t of the first CH threads owns row t
  auto stats_load = [&](int q0_, float& lv, float& dv) {
    const int t = threadIdx.x;
    lv = INFINITY;
    dv = 0.f;
    if (t >= G::CH) return;
    const int tq = q0_ + t;
    if (tq >= p.Tq) return;
    if (G::NSLOT == 2 || p.delta_ready) {  // (the two-slot launches always find delta filled: attn_delta_kernel)
      dv = p.delta[((long)b * p.H + h) * p.Tq + tq];
    } else {
      const bf16_t* dor = p.d_o + (long)b * p.o_sb + (long)tq * p.o_st + h * dh;
      const bf16_t* orr = p.o_in + (long)b * p.o_sb + (long)tq * p.o_st + h * dh;
      for (int sl = 0; sl < dslots; ++sl) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(dor + sl * 8);
        const u32x4 c = *reinterpret_cast<const u32x4*>(orr + sl * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) dv += bf16lo(a[e]) * bf16lo(c[e]) + bf16hi(a[e]) * bf16hi(c[e]);
      }
    }
    lv = p.lse[((long)b * p.H + h) * p.Tq + tq] * LOG2E;
  };
  auto stats_store = [&](char* slot, float lv, float dv) {
    if ((int)threadIdx.x < G::CH) {
      float* ls = reinterpret_cast<float*>(slot + G::SLOT_BYTES);
      ls[threadIdx.x] = lv;
      ls[G::CH + threadIdx.x] = dv;
    }
  };
  auto issue = [&](int q0_, char* slot) {
    const int rows_ = min(G::CH, p.Tq - q0_);
    int ln = lane;  // opaque copy: see attn_gen_fwd_kernel
    asm volatile("" : "+v"(ln));
    dma_oper<NH>(slot, p.q + (long)b * p.q_sb + (long)q0_ * p.q_st + h * dh, p.q_st, rows_, dh, wave, nwaves, ln);
    dma_oper<NH>(slot + G::OPER_BYTES, p.d_o + (long)b * p.o_sb + (long)q0_ * p.o_st + h * dh, p.o_st, rows_, dh, wave, nwaves, ln);
  };
  constexpr int RING = G::SLOT_BYTES + 2 * G::CH * (int)sizeof(float);  // operands + statistics of one chunk
  float nlv = INFINITY, ndv = 0.f;  // statistics of the NEXT chunk: loaded before this chunk's MFMAs, stored after them
  if (G::NSLOT == 2) {
    stats_load(0, nlv, ndv);
    issue(0, smem);
    stats_store(smem, nlv, ndv);
  }
  for (int q0 = 0, c = 0; q0 < p.Tq; q0 += G::CH, ++c) {
    const int rows = min(G::CH, p.Tq - q0);
    const bool more = q0 + G::CH < p.Tq;
    if (G::NSLOT == 2) {
      lds_dma_wait_all();
      __syncthreads();  // chunk c (operands and statistics) is visible; every wave is done with chunk c - 1 (the other slot)
      if (more) {
        stats_load(q0 + G::CH, nlv, ndv);  // (issued BEFORE the DMA: the compiler's wait for these loads at the store below
        issue(q0 + G::CH, smem + ((c + 1) & 1) * RING);  //  comes after this chunk's MFMAs, when the DMA has landed anyway)
      }
    } else {
      __syncthreads();
      issue(q0, smem);
      stats_load(q0, nlv, ndv);
      stats_store(smem, nlv, ndv);
      lds_dma_wait_all();
      __syncthreads();
    }
    const bool skip = !active || (p.causal && q0 + G::CH - 1 < row0);  // inactive tile / every query of the chunk precedes it
    if (skip) {
      if (G::NSLOT == 2 && more) stats_store(smem + ((c + 1) & 1) * RING, nlv, ndv);
      continue;
    }
    // the chunk's MFMA work, once per ring slot with COMPILE-TIME LDS addresses (a run-time slot base costs an address
    // register per fragment read: the 128-register budget of two workgroups per CU does not have them)
    auto body = [&](auto slot_tag) {
      constexpr int S = decltype(slot_tag)::value;
      const char* Qs_ = smem + S * RING;
      const char* dOs_ = Qs_ + G::OPER_BYTES;
      const float* lse_ = reinterpret_cast<const float*>(Qs_ + G::SLOT_BYTES);
      const float* delta_ = lse_ + G::CH;
      const int nbl = (rows + 31) / 32;
#pragma unroll 1
      for (int a = 0; a < nbl; ++a) {
        f32x4 pp[2], ds[2];
  #pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int it = 2 * a + t;
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
          for (int ks = 0; ks < 2 * NH; ++ks) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(Qs_ + (ks >> 1) * G::HALF_BYTES, it * 16, ks & 1, lane),
                                                         kf[ks], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rows(dOs_ + (ks >> 1) * G::HALF_BYTES, it * 16, ks & 1, lane),
                                                         vf[ks], dp, 0, 0, 0);
          }
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_ + it * 16 + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(delta_ + it * 16 + 4 * g);
          Philox rb = {{0u, 0u, 0u, 0u}};
          if (DROP) rb = drop_block(p, b, h, (q0 >> 2) + it * 4 + g, kj >> 2);  // rows 4g .. 4g + 3 = words 0 .. 3
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(sc[r] * sl2 - l4[r]);
            if (!PLAIN) {
              const int qi = q0 + it * 16 + 4 * g + r;
              pr = (qi < p.Tq && keep_at(p, b, h, qi, kj)) ? pr : 0.f;
            }
            float keep_scale = 1.0f;
            if (DROP) keep_scale = ((rb.c[r] >> (8 * (kj & 3))) & 255u) >= p.drop_thresh ? p.drop_scale : 0.f;
            pp[t][r] = pr * keep_scale;
            ds[t][r] = pr * (dp[r] * keep_scale - d4[r]);
          }
        }
        const bf16x8 ppk = pack8(pp[0], pp[1]);
        const bf16x8 dsk = pack8(ds[0], ds[1]);
  #pragma unroll
        for (int dt = 0; dt < 4 * NH; ++dt) {
          dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              frag_cols(dOs_ + (dt >> 2) * G::HALF_BYTES, a * 32, (dt & 3) * 16, lane), ppk, dvt[dt], 0, 0, 0);
          dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              frag_cols(Qs_ + (dt >> 2) * G::HALF_BYTES, a * 32, (dt & 3) * 16, lane), dsk, dkt[dt], 0, 0, 0);
        }
      }
    };
    if (G::NSLOT == 2 && (c & 1)) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});
    if (G::NSLOT == 2 && more) stats_store(smem + ((c + 1) & 1) * RING, nlv, ndv);
  }
  if (active && kj < p.Tk) {
    bf16_t* dkrow = p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * dh;
    bf16_t* dvrow = p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * dh;
#pragma unroll
    for (int dt = 0; dt < 4 * NH; ++dt) {
      const int col = dt * 16 + 4 * g;
      if (col >= dh) continue;
      const f32x4 kk = dkt[dt] * p.scale;
      const f32x4 vv = dvt[dt];
      *reinterpret_cast<u32x2*>(dkrow + col) = u32x2{pack_bf16x2(kk[0], kk[1]), pack_bf16x2(kk[2], kk[3])};
      *reinterpret_cast<u32x2*>(dvrow + col) = u32x2{pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3])};
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 3, head_dim <= 64 without dropout: TWO 16-row tiles per wave.
// In the kernels above every wave reads the WHOLE staged chunk from LDS for its 16 rows: per 256-key chunk 32 ds_read_b128
// + 64 ds_read_b64_tr_b16 per wave = 256 LDS cycles, sixteen waves per CU = 4096 — exactly the matrix time of the chunk
// (64 MFMAs x 16 cycles x 4 waves per SIMD): LDS bandwidth and the matrix pipe are co-bottlenecks and neither gets above
// half.  Here a wave owns 32 query rows (forward, dQ) / 32 key rows (dK, dV): every K / V (Q / dO) fragment read from LDS
// feeds two MFMAs, a workgroup covers 256 rows per staged chunk instead of 128 (half the L2 -> LDS bytes per FLOP), and
// the two tiles are independent dependency chains inside the wave.  Scores live in registers for 64 keys at a time (32
// registers for both tiles; the 256-key form would need 128), online softmax per 64-key block; the row sum is kept per
// lane and folded across the four lanes of a row once, at the end.  Output columns beyond head_dim are not computed:
// NDT = 3 column tiles for head_dim <= 48 (the UNet's 40-channel heads), 4 otherwise.
// ------------------------------------------------------------------------------------------------
// Tried and NOT kept: columns 32 .. 47 of a head_dim <= 48 product through ONE v_mfma_f32_16x16x16_bf16 instead of the second,
// three-quarters-empty 16x16x32.  hipcc schedules the 16-deep MFMA directly behind the 32-deep one whose result it accumulates
// onto, with a DIFFERENT destination register and no wait states; the results are wrong (tools/micro/mfma_chain.hip shows the
// in-place chain is fine, tools/micro/mfma16_layout.hip that the operand layout is what one expects).  The gain would have been
// 2-5 % on the backward passes and nothing on the VALU-bound forward; a hand-padded chain is not worth a silent-corruption risk.
constexpr int A2_ROWS = 256;   // rows of the walking operand per workgroup (8 waves x 32)
constexpr int A2_CH = 256;     // rows of the staged operand per chunk (64 KB: K + V, or Q + dO)
constexpr int A2_TILE = A2_CH * 128;

// SUMCOL (head_dim = 8 mod 16, e.g. the UNet's 40: the last column tile has free columns): the softmax row sum comes out of
// the matrix pipe.  Column dh of the staged V tile is a column of ONES (written once: the chunk DMAs skip the pad columns,
// dma_half<SKIP_PAD>), so O[:, dh] = sum_j p_ij accumulates — and is rescaled by alpha — with the output it normalises,
// and the 32 adds per 64-key block (16 v_pk_add_f32 per wave in a VALU-bound loop: 163 VALU per 28 MFMAs) disappear.  The sum
// is then that of the bf16-rounded probabilities, i.e. of exactly what multiplied V.
// Both forms skip the rescale of O when no row of the wave raised its running maximum in this block (alpha == 1 for all 32
// rows — the common case after the first few blocks of a long sequence): a wave-uniform branch around 12-16 v_pk_mul_f32.
#ifndef CFHIP_ATTN_NDT4_WAVES
#define CFHIP_ATTN_NDT4_WAVES 4  // waves per SIMD of the head_dim-64 (NDT = 4) long-sequence forms: 4 keeps 12-36 B of scratch, 3 has none (A/B: profiles/r05)
#endif
// NDT = 5 / 6 (head_dim 72 .. 96: the UNet's 80-channel heads; round 6, late): the same loop over KS = 3 reduction steps and two
// 64-column halves of the staged K / V — as attn_bwd_dkv2_kernel does it: 4 waves, 128-key chunks (64 KB of LDS), two workgroups per CU
// at 256 registers.  The one-tile general kernel ran this head_dim's forward at 527 TFLOP/s against 1 020 for its dK / dV pass.
template <int NDT>
struct Fwd2 {
  static constexpr int NH = (NDT + 3) / 4;  // 64-column halves of a staged operand
  static constexpr int KS = (NDT + 1) / 2;  // 32-deep steps of the reduction over head_dim
  static constexpr int NW = NH == 1 ? 8 : 4;
  static constexpr int CH = Gen<NH>::CH;    // keys per staged chunk: 256 / 128
  static constexpr int HB = Gen<NH>::HALF_BYTES;
  static constexpr int WAVES_PER_SIMD = NH == 1 ? (NDT == 4 ? CFHIP_ATTN_NDT4_WAVES : 4) : 2;
};
template <bool PLAIN, int NDT, bool SUMCOL = false>
__global__ __launch_bounds__(Fwd2<NDT>::NW * 64, Fwd2<NDT>::WAVES_PER_SIMD) void attn_fwd2_kernel(AttnParams p) {
  using F = Fwd2<NDT>;
  constexpr int NH = F::NH, KS = F::KS, NW = F::NW, CH = F::CH, HB = F::HB;
  static_assert(!SUMCOL || NH == 1, "the ones column: single-half forms only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + NH * HB;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  if (SUMCOL) {
    // pad chunks of both tiles, once: zeros, and 1.0 in column dh of V (dh % 16 == 8: element 0 of its 16-byte chunk)
    for (int idx = threadIdx.x; idx < CH * 8; idx += NW * 64) {
      const int row = idx >> 3, chunk = idx & 7;
      if (chunk * 8 >= dh) {
        const int off = row * 128 + ((chunk ^ swz(row)) << 4);
        *reinterpret_cast<u32x4*>(Ks + off) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(Vs + off) = u32x4{chunk * 8 == dh ? 0x3F80u : 0u, 0u, 0u, 0u};
      }
    }
  }
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * dh;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * dh;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * dh;
  const int row0 = (blockIdx.x * NW + wave) * 32;
  const bool active = row0 < p.Tq;  // inactive waves still stage and hit the barriers
  bf16x8 qf[2][KS];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[t][ks] = frag_global_dh(qb, p.q_st, row0 + 16 * t, active ? p.Tq : 0, ks, lane, dh);
  const float sl2 = p.scale * LOG2E;
  float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};  // l: this LANE's share of the row sum
  f32x4 ot[2][NDT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) ot[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kv0 = 0; kv0 < p.Tk; kv0 += CH) {
    const int rows = min(CH, p.Tk - kv0);
    __syncthreads();  // every wave is done with the previous chunk
    {
      int ln = lane;  // opaque copy: the per-lane DMA offsets are recomputed per chunk instead of hoisted and spilled
      asm volatile("" : "+v"(ln));
      dma_oper<NH, SUMCOL>(Ks, kb + (long)kv0 * p.kv_st, p.kv_st, rows, dh, wave, NW, ln);
      dma_oper<NH, SUMCOL>(Vs, vb + (long)kv0 * p.kv_st, p.kv_st, rows, dh, wave, NW, ln);
    }
    lds_dma_wait_all();
    __syncthreads();
    if (!active) continue;
    if (p.causal && kv0 > row0 + 31) continue;  // chunk entirely in the future of both tiles (wave-uniform)
    const int nsb = (rows + 63) >> 6;
#pragma unroll 1
    for (int sb = 0; sb < nsb; ++sb) {
      const int k0 = sb * 64;
      if (p.causal && kv0 + k0 > row0 + 31) break;
      f32x4 st[2][4];
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        bf16x8 kf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = frag_rows(Ks + (ks >> 1) * HB, k0 + jt * 16, ks & 1, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[t][ks], acc, 0, 0, 0);
          st[t][jt] = acc;
        }
      }
      // Key positions >= Tk (zero rows of the staged chunk) exist only in the last 64-key block of a sequence whose length
      // is not a multiple of 64: a REAL wave-uniform branch around a masking pass.  Written as `if (tail) x = j < Tk ? x :
      // -inf` inside the softmax loop, hipcc if-converted the test into two v_cndmask + index arithmetic for every score of
      // every block: 104 of the loop's ~250 VALU instructions, in a kernel that is VALU-bound (10.9 VALU per MFMA:
      // profiles/r03/pmc_attn_fwd2_first.txt); an empty asm inside the branch was not enough either (the selects were
      // hoisted above it), the compared bound has to be opaque.
      if (PLAIN && kv0 + k0 + 64 > p.Tk) {
        int lim = p.Tk - (kv0 + k0);    // valid keys of this block
        asm volatile("" : "+s"(lim));  // opaque and produced INSIDE the branch: the selects cannot be hoisted above it
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              st[t][jt][r] = jt * 16 + 4 * g + r < lim ? st[t][jt][r] : -INFINITY;
      }
      float alpha[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int qi = row0 + 16 * t + i;
        float mx = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = st[t][jt][r];  // raw score: the scale is folded into the exponent below (scale > 0)
            if (!PLAIN) {
              x = keep_at(p, b, h, qi < p.Tq ? qi : p.Tq - 1, kv0 + k0 + jt * 16 + 4 * g + r) ? x : -INFINITY;
              st[t][jt][r] = x;
            }
            mx = fmaxf(mx, x);
          }
        }
        mx = group_max_swap(mx) * sl2;
        const float m_new = max_nn(m[t], mx);
        // rows whose every position so far is masked (m_new = -inf) must not produce NaN from (-inf) - (-inf): use 0
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        alpha[t] = __builtin_amdgcn_exp2f(m[t] - m_use);  // m = -inf on the first block: 0; unchanged maximum: exactly 1
        float ls = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[t][jt][r], sl2, -m_use));
            st[t][jt][r] = e;
            if (!SUMCOL) ls += e;
          }
        }
        if (!SUMCOL) l[t] = l[t] * alpha[t] + ls;
        m[t] = m_new;
      }
      if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[1] != 1.f) != 0) {  // wave-uniform
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) ot[t][dt] *= alpha[t];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const bf16x8 pa0 = pack8(st[0][2 * a], st[0][2 * a + 1]);
        const bf16x8 pa1 = pack8(st[1][2 * a], st[1][2 * a + 1]);
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
          const bf16x8 vf = frag_cols(Vs + (dt >> 2) * HB, k0 + a * 32, (dt & 3) * 16, lane);
          ot[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pa0, ot[0][dt], 0, 0, 0);
          ot[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pa1, ot[1][dt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qi = row0 + 16 * t + i;
    // SUMCOL: the row sum is column dh = (NDT - 1) * 16 + 8 of O: register 0 of the lanes with g = 2
    const float lsum = SUMCOL ? __shfl(ot[t][NDT - 1][0], 32 + i) : group_sum(l[t]);
    if (active && qi < p.Tq) {
      const float inv = 1.0f / lsum;
      bf16_t* orow = p.o + (long)b * p.o_sb + (long)qi * p.o_st + h * dh;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col >= dh) continue;  // dh % 8 == 0: a 4-column group is entirely inside or outside
        const f32x4 v = ot[t][dt] * inv;
        *reinterpret_cast<u32x2*>(orow + col) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
      if (g == 0 && p.lse != nullptr) p.lse[((long)b * p.H + h) * p.Tq + qi] = (m[t] + log2f(lsum)) * (1.0f / LOG2E);
    }
  }
}

// dQ pass, two query tiles per wave (see attn_fwd2_kernel): per 32-key block the four K / V row fragments and the NDT K column
// fragments are read once for both tiles: 22 MFMAs per 8 ds_read_b128 + 6 ds_read_b64_tr_b16 (NDT = 3) against 12 per 8 + 8.
// Key rows beyond Tk are zero rows of the staged chunk: their p is finite, their dP and their K are zero, they add nothing.
// (masked / causal instantiations: the predicate registers do not fit beside two 16-row tiles at 128 registers — two waves per
// SIMD for them, no scratch; PLAIN, the UNet's and the ViT's path, keeps four)
// NDT = 5 / 6 (head_dim 72 .. 96; round 6, late): KS = 3 reduction steps and two 64-column halves per staged operand, 4 waves, 128-key
// chunks, two workgroups per CU — the geometry of Fwd2<NDT> (the general one-tile kernel ran the UNet's 80-channel dQ at 746 TFLOP/s).
template <bool PLAIN, int NDT>
__global__ __launch_bounds__(Fwd2<NDT>::NW * 64, PLAIN ? Fwd2<NDT>::WAVES_PER_SIMD : 2) void attn_bwd_dq2_kernel(AttnParams p) {
  using F = Fwd2<NDT>;
  constexpr int NH = F::NH, KS = F::KS, NW = F::NW, CH = F::CH, HB = F::HB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + NH * HB;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int row0 = (blockIdx.x * NW + wave) * 32;
  const bool active = row0 < p.Tq;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * dh;
  const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * dh;
  const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * dh;
  bf16x8 qf[2][KS], dof[2][KS];
  float delta[2], lse2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float sacc = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[t][ks] = frag_global_dh(qb, p.q_st, row0 + 16 * t, active ? p.Tq : 0, ks, lane, dh);
      dof[t][ks] = frag_global_dh(dob, p.o_st, row0 + 16 * t, active ? p.Tq : 0, ks, lane, dh);
      const bf16x8 of = frag_global_dh(ob, p.o_st, row0 + 16 * t, active ? p.Tq : 0, ks, lane, dh);
#pragma unroll
      for (int e = 0; e < 8; ++e) sacc += bf16_to_f32((bf16_t)dof[t][ks][e]) * bf16_to_f32((bf16_t)of[e]);
    }
    delta[t] = group_sum(sacc);
    const int qi = row0 + 16 * t + i;
    const bool qvalid = active && qi < p.Tq;
    const long stat = ((long)b * p.H + h) * p.Tq + qi;
    if (qvalid && g == 0) p.delta[stat] = delta[t];
    lse2[t] = qvalid ? p.lse[stat] * LOG2E : INFINITY;
  }
  const float sl2 = p.scale * LOG2E;
  f32x4 dqt[2][NDT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dqt[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int kv0 = 0; kv0 < p.Tk; kv0 += CH) {
    const int rows = min(CH, p.Tk - kv0);
    __syncthreads();
    {
      int ln = lane;  // opaque copy: see attn_fwd2_kernel
      asm volatile("" : "+v"(ln));
      dma_oper<NH>(Ks, p.k + (long)b * p.kv_sb + (long)kv0 * p.kv_st + h * dh, p.kv_st, rows, dh, wave, NW, ln);
      dma_oper<NH>(Vs, p.v + (long)b * p.kv_sb + (long)kv0 * p.kv_st + h * dh, p.kv_st, rows, dh, wave, NW, ln);
    }
    lds_dma_wait_all();
    __syncthreads();
    if (!active) continue;
    if (p.causal && kv0 > row0 + 31) continue;
    const int nbl = (rows + 31) / 32;
#pragma unroll 1
    for (int a = 0; a < nbl; ++a) {
      f32x4 ds[2][2];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int r16 = a * 32 + jt * 16;
        bf16x8 kf[KS], vf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          kf[ks] = frag_rows(Ks + (ks >> 1) * HB, r16, ks & 1, lane);
          vf[ks] = frag_rows(Vs + (ks >> 1) * HB, r16, ks & 1, lane);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf[t][ks], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], dof[t][ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(fmaf(sc[r], sl2, -lse2[t]));
            if (!PLAIN) {
              const int qi = row0 + 16 * t + i;
              pr = keep_at(p, b, h, qi < p.Tq ? qi : p.Tq - 1, kv0 + r16 + 4 * g + r) ? pr : 0.f;
            }
            ds[t][jt][r] = pr * (dp[r] - delta[t]);
          }
        }
      }
      const bf16x8 dsp0 = pack8(ds[0][0], ds[0][1]);
      const bf16x8 dsp1 = pack8(ds[1][0], ds[1][1]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const bf16x8 kc = frag_cols(Ks + (dt >> 2) * HB, a * 32, (dt & 3) * 16, lane);
        dqt[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, dsp0, dqt[0][dt], 0, 0, 0);
        dqt[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kc, dsp1, dqt[1][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qi = row0 + 16 * t + i;
    if (active && qi < p.Tq) {
      bf16_t* dqrow = p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * dh;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col >= dh) continue;
        const f32x4 v = dqt[t][dt] * p.scale;
        *reinterpret_cast<u32x2*>(dqrow + col) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
}

// dK / dV pass, two key tiles per wave: 4-wave workgroups of 128 key rows over 128-row chunks of Q / dO (33 KB of LDS: three to
// four workgroups per CU).  Per 32-query block the eight Q / dO row fragments, the statistics and the 2 NDT column fragments are
// read once for both key tiles: 28 MFMAs per 12 ds_read_b128 + 12 ds_read_b64_tr_b16 (NDT = 3) against 16 per 12 + 16 — the
// one-tile kernel is LDS-bandwidth-bound (80 LDS cycles per wave and block x 16 waves = 1 280 per CU against 1 024 matrix cycles
// per SIMD).  ~165 registers: three waves per SIMD.  delta must be ready (the dQ pass, or attn_delta_kernel).
constexpr int A2D_CH = 128;
constexpr int A2D_TILE = A2D_CH * 128;

// NDT = 5 / 6 (head_dim 72 .. 96, the UNet's 80-channel heads; round 3b): the same loop over KS = 3 reduction steps and two
// 64-column halves of the staged operands (66 KB of LDS, ~215 registers: two workgroups per CU) — the general one-tile kernel
// ran this head_dim at 342 TFLOP/s against 800 for head_dim 40 here.
// FOLD (round 6; plain, head_dim 40): dP - delta from the matrix pipe, as in attn_bwd_dq3_kernel — ones in columns 40 / 41 of the V
// fragments, -bf16(delta) / -bf16(delta - hi) in columns 40 / 41 of the staged dO rows (written by the thread that owns the row's
// statistics; the dO DMAs skip the pad columns).
template <bool PLAIN, int NDT, bool FOLD = false>
__global__ __launch_bounds__(256, (NDT == 3 && PLAIN) ? 3 : 2) void attn_bwd_dkv2_kernel(AttnParams p) {
  static_assert(!FOLD || (PLAIN && NDT == 3), "the folded delta: plain head_dim-40 form only");
  constexpr int KS = (NDT + 1) / 2;     // 32-deep steps of the reductions over head_dim
  constexpr int NHALF = (NDT + 3) / 4;  // 64-column halves of a staged operand
  extern __shared__ __attribute__((aligned(128))) char smem[];  // 128: the XOR-derived addresses below need whole tile rows
  char* Qs = smem;
  char* dOs = smem + NHALF * A2D_TILE;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * NHALF * A2D_TILE);
  float* delta_s = lse_s + A2D_CH;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int row0 = (blockIdx.x * 4 + wave) * 32;
  const bool active = row0 < p.Tk;
  const bf16_t* kb = p.k + (long)b * p.kv_sb + h * dh;
  const bf16_t* vb = p.v + (long)b * p.kv_sb + h * dh;
  bf16x8 kf[2][KS], vf[2][KS];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[u][ks] = frag_global_dh(kb, p.kv_st, row0 + 16 * u, active ? p.Tk : 0, ks, lane, dh);
      vf[u][ks] = frag_global_dh(vb, p.kv_st, row0 + 16 * u, active ? p.Tk : 0, ks, lane, dh);
    }
  if (FOLD) {
    if (g == 1) {  // dh == 40: the (ks = 1, g = 1) fragment is columns 40 .. 47
      vf[0][KS - 1][0] = vf[0][KS - 1][1] = (short)0x3F80;
      vf[1][KS - 1][0] = vf[1][KS - 1][1] = (short)0x3F80;
    }
    for (int idx = threadIdx.x; idx < A2D_CH * 8; idx += 256) {  // the pad chunks of the dO tile, once (columns 40 / 41 are rewritten per chunk)
      const int row = idx >> 3, chunk = idx & 7;
      if (chunk * 8 >= dh) *reinterpret_cast<u32x4*>(dOs + row * 128 + ((chunk ^ swz(row)) << 4)) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  const float sl2 = p.scale * LOG2E;
  f32x4 dkt[2][NDT], dvt[2][NDT];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { dkt[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  // Lane-constant parts of the LDS addresses of a block as opaque LDS-space pointers (see attn_bwd_dq3_kernel / attn_bwd_one_kernel):
  // the other reduction half of a row fragment and the other column tiles are an XOR of bits 5..6 away, halves / dO / the second
  // transposing read / the second 16-row tile are immediates; a block adds its a * 4 096 (statistics: a * 128) bytes to three of them.
  typedef __attribute__((address_space(3))) const char* lds_cp;
  typedef __attribute__((address_space(3))) const bf16x8* lds_v8;
  typedef __attribute__((address_space(3))) const f32x4* lds_f4;
  typedef __attribute__((address_space(3))) s16x4* lds_s4;
  lds_cp qr0, qc0, sp;
  {
    const lds_cp q3 = (lds_cp)LDS_PTR(Qs);
    qr0 = q3 + n * 128 + ((g ^ swz(n)) << 4);
    qc0 = q3 + tile_off(4 * g + (n >> 2), 4 * (n & 3));
    sp = (lds_cp)LDS_PTR(lse_s) + 16 * g;
#define CFHIP_OPAQUE_LDS(ptr, T) { unsigned u_ = (unsigned)(uintptr_t)(ptr); asm volatile("" : "+v"(u_)); (ptr) = (T)(uintptr_t)u_; }
#define CFHIP_LDS_XOR(ptr, bits) ((lds_cp)(uintptr_t)((unsigned)(uintptr_t)(ptr) ^ (unsigned)(bits)))
    CFHIP_OPAQUE_LDS(qr0, lds_cp)
    CFHIP_OPAQUE_LDS(qc0, lds_cp)
    CFHIP_OPAQUE_LDS(sp, lds_cp)
  }
  constexpr int DO_OFF = NHALF * A2D_TILE;  // the dO tile(s) behind the Q tile(s)
  for (int q0 = 0; q0 < p.Tq; q0 += A2D_CH) {
    const int rows = min(A2D_CH, p.Tq - q0);
    __syncthreads();
    {
      int ln = lane;  // opaque copy: see attn_fwd2_kernel
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int hf = 0; hf < NHALF; ++hf) {
        dma_half(Qs + hf * A2D_TILE, p.q + (long)b * p.q_sb + (long)q0 * p.q_st + h * dh, p.q_st, rows, A2D_CH, hf, dh, wave, 4, ln);
        dma_half<FOLD>(dOs + hf * A2D_TILE, p.d_o + (long)b * p.o_sb + (long)q0 * p.o_st + h * dh, p.o_st, rows, A2D_CH, hf, dh, wave, 4, ln);
      }
    }
    if ((int)threadIdx.x < A2D_CH) {  // thread t owns the statistics of row t of the chunk
      const int tq = q0 + (int)threadIdx.x;
      const long at = ((long)b * p.H + h) * p.Tq + tq;
      lse_s[threadIdx.x] = tq < p.Tq ? p.lse[at] * LOG2E : INFINITY;
      const float dl = tq < p.Tq ? p.delta[at] : 0.f;
      if (FOLD) {
        const bf16_t hi = f32_to_bf16(-dl);
        const bf16_t lo = f32_to_bf16(-dl - bf16_to_f32(hi));
        *reinterpret_cast<unsigned*>(dOs + tile_off((int)threadIdx.x, 40)) = (unsigned)hi | ((unsigned)lo << 16);
      } else {
        delta_s[threadIdx.x] = dl;
      }
    }
    lds_dma_wait_all();
    __syncthreads();
    if (!active) continue;
    if (p.causal && q0 + A2D_CH - 1 < row0) continue;  // every query of the chunk precedes both kv tiles
    const int nbl = (rows + 31) / 32;
#pragma unroll 1
    for (int a = 0; a < nbl; ++a) {
      f32x4 pp[2][2], ds[2][2];  // [kv tile][query 16-row tile]
      const int aoff = a * (32 * 128);
      const lds_cp ra = qr0 + aoff, rb = CFHIP_LDS_XOR(qr0 + aoff, 64), spa = sp + a * 128;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r16 = a * 32 + it * 16;
        bf16x8 qf[KS], of[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const lds_cp rp = ((ks & 1) ? rb : ra) + (ks >> 1) * A2D_TILE + it * 2048;  // rows r16 + n, reduction half ks
          qf[ks] = *(lds_v8)rp;
          of[ks] = *(lds_v8)(rp + DO_OFF);
        }
        const f32x4 l4 = *(lds_f4)(spa + it * 64);
        f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
        if (!FOLD) d4 = *(lds_f4)(spa + it * 64 + A2D_CH * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf[u][ks], sc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(of[ks], vf[u][ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr = __builtin_amdgcn_exp2f(fmaf(sc[r], sl2, -l4[r]));
            if (!PLAIN) {
              const int qi = q0 + r16 + 4 * g + r;
              pr = (qi < p.Tq && keep_at(p, b, h, qi, row0 + 16 * u + n)) ? pr : 0.f;
            }
            pp[u][it][r] = pr;
            ds[u][it][r] = pr * (FOLD ? dp[r] : dp[r] - d4[r]);
          }
        }
      }
      const bf16x8 ppk0 = pack8(pp[0][0], pp[0][1]), ppk1 = pack8(pp[1][0], pp[1][1]);
      const bf16x8 dsk0 = pack8(ds[0][0], ds[0][1]), dsk1 = pack8(ds[1][0], ds[1][1]);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const lds_cp cp = CFHIP_LDS_XOR(qc0 + aoff, (dt & 3) * 32) + (dt >> 2) * A2D_TILE;  // rows 32 a + 4 g + (n >> 2) (+16: 2 048 bytes on)
        const s16x4 olo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + DO_OFF));
        const s16x4 ohi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + DO_OFF + 2048));
        const bf16x8 oc = {olo[0], olo[1], olo[2], olo[3], ohi[0], ohi[1], ohi[2], ohi[3]};
        dvt[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oc, ppk0, dvt[0][dt], 0, 0, 0);
        dvt[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oc, ppk1, dvt[1][dt], 0, 0, 0);
        const s16x4 qlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)cp);
        const s16x4 qhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(cp + 2048));
        const bf16x8 qc = {qlo[0], qlo[1], qlo[2], qlo[3], qhi[0], qhi[1], qhi[2], qhi[3]};
        dkt[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qc, dsk0, dkt[0][dt], 0, 0, 0);
        dkt[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qc, dsk1, dkt[1][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int kj = row0 + 16 * u + n;
    if (active && kj < p.Tk) {
      bf16_t* dkrow = p.dk + (long)b * p.kv_sb + (long)kj * p.kv_st + h * dh;
      bf16_t* dvrow = p.dv + (long)b * p.kv_sb + (long)kj * p.kv_st + h * dh;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col >= dh) continue;
        const f32x4 kk = dkt[u][dt] * p.scale;
        const f32x4 vv = dvt[u][dt];
        *reinterpret_cast<u32x2*>(dkrow + col) = u32x2{pack_bf16x2(kk[0], kk[1]), pack_bf16x2(kk[2], kk[3])};
        *reinterpret_cast<u32x2*>(dvrow + col) = u32x2{pack_bf16x2(vv[0], vv[1]), pack_bf16x2(vv[2], vv[3])};
      }
    }
  }
}
#undef CFHIP_OPAQUE_LDS
#undef CFHIP_LDS_XOR


// ------------------------------------------------------------------------------------------------
// Round 6: the backward passes of head_dim <= 48 (the UNet's 40-channel heads: 333 of the 256^2 step's 474 ms of kernel time) with the
// two reductions over head_dim — S = Q K^T and dP = dO V^T — on v_mfma_f32_32x32x16_bf16.  The kernels above run them on 16x16x32
// MFMAs: head_dim 40 is padded to 64 there (two 32-deep steps, the second three quarters empty) and the passes sit at their MFMA-issue
// bound (dK / dV: 1.6 + 1.6 + 1.2 + 1.2 = 5.6 padded units, 1 166 TFLOP/s issued).  A 32x32 tile takes the reduction in 16-deep steps:
// three of them (48) instead of 64 — 3.6 units instead of 4.4 for dQ.  (The 16-deep 16x16 MFMA costs what the 32-deep one does on gfx950
// — measured, profiles/r06/attn_k48_mfma16_rejected.txt — the 32x32x16 one is the double-rate instruction of its shape.)
// MEASURED (profiles/r06/attn_s32x32_ab.txt): dQ at T = 65 536 10.38 -> 9.79 ms (-5.6 %), at T = 4 096 x 8 410 -> 391 us: a quarter of
// what the MFMA count promised — the passes turn out to sit at ~50 % of the matrix pipe with the VALU (exp2, two multiplies, a subtract
// and the packing per score) about as busy, so removing matrix cycles returns a fraction.  The same form of the dK / dV pass (eight
// statistics loads and two swaps per block instead of four and none) was 1.8 % SLOWER (13.60 -> 13.84 ms) and is not in the tree.
// The price is the result layout: lane l holds ONE column (l & 31) and the 16 rows 8 j + 4 (l >> 5) + r, where the products that consume
// dS as an operand (dQ: output width head_dim, 48 on 16-column tiles — a 32-wide tile would pad it to 64 again) want 8
// reduction slots of one of 16 columns per lane.  v_permlane16_swap does that conversion: with X = the packed rows of j in {0, 1} and
// Y = those of j in {2, 3}, swapping X's odd 16-lane rows with Y's even ones leaves in X the complete 16x16x32 operand of columns 0..15
// and in Y that of columns 16..31, k-slot group g = lane >> 4 holding rows base(g) + {0..3, 8..11}, base = {0, 16, 4, 20}: the
// transposing LDS reads of the other operand take their four-row runs from exactly those rows (frag_cols32).  tools/micro/mfma32_swap.hip
// checks the chain bit for bit.  No mask / causal / dropout forms: the UNet's path (PLAIN); everything else stays on the kernels above.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// A operand of a 32x32x16 MFMA from a swizzled tile: lane (rho = l & 31, hk = l >> 5) <- tile[row0 + rho][ks*16 + 8 hk .. +8]
__device__ __forceinline__ bf16x8 frag_rows32(const char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 31);
  const int slot = ks * 2 + (lane >> 5);
  return *reinterpret_cast<const bf16x8*>(tile + row * 128 + ((slot ^ swz(row)) << 4));
}

// the same straight from global memory for the wave's own 32 rows, zero beyond dh / rows_valid
__device__ __forceinline__ bf16x8 frag_global32_dh(const bf16_t* base, long stride_t, int row0, int rows_valid, int ks, int lane, int dh) {
  const int row = row0 + (lane & 31);
  const int col = ks * 16 + (lane >> 5) * 8;
  bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
  if (row < rows_valid && col < dh) r = *reinterpret_cast<const bf16x8*>(base + (long)row * stride_t + col);
  return r;
}

// column-operand fragment over the 32-row block at `r32`, columns c0..c0+15, in the k-slot order of swap32_operands' results
__device__ __forceinline__ bf16x8 frag_cols32(const char* tile, int r32, int c0, int lane) {
  const int g = lane >> 4, s = lane & 15;
  const int r_lo = r32 + (g & 1) * 16 + (g >> 1) * 4 + (s >> 2);
  const int col = c0 + 4 * (s & 3);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + tile_off(r_lo, col)));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + tile_off(r_lo + 8, col)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// 32x32 MFMA result (16 f32 per lane: column l & 31, rows 8 j + 4 (l >> 5) + r) -> the bf16 B operands of two 16x16x32 MFMAs:
// `lo` for columns 0..15, `hi` for columns 16..31 (lane n = l & 15 in both), 32 reduction slots in frag_cols32's order
__device__ __forceinline__ void swap32_operands(const f32x16& t, bf16x8& lo, bf16x8& hi) {
  union { bf16x8 v; unsigned w[4]; } x, y;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    x.w[w] = pack_bf16x2(t[2 * w], t[2 * w + 1]);
    y.w[w] = pack_bf16x2(t[8 + 2 * w], t[8 + 2 * w + 1]);
  }
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const auto r = __builtin_amdgcn_permlane16_swap(x.w[w], y.w[w], false, false);
    x.w[w] = r[0];
    y.w[w] = r[1];
  }
  lo = x.v;
  hi = y.v;
}

// FOLD (head_dim 40: the reduction is padded to 48, columns 40 / 41 are free): dP - delta comes out of the matrix pipe.  The staged V tile
// carries ones in columns 40 and 41 (written once; the chunk DMAs skip the pad columns), the dO fragment -bf16(delta) and -bf16(delta - hi)
// (16 mantissa bits of delta): 16 subtractions per lane and 32 x 32 block leave a loop whose VALU is as busy as its matrix pipe.
template <int NDT, bool FOLD>
__global__ __launch_bounds__(512, 4) void attn_bwd_dq3_kernel(AttnParams p) {
  static_assert(NDT == 3, "head_dim <= 48: three 16-deep reduction steps, three 16-column output tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vs = smem + A2_TILE;
  const int b = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int dh = p.dh;
  const int row0 = (blockIdx.x * 8 + wave) * 32;
  const bool active = row0 < p.Tq;
  const bf16_t* qb = p.q + (long)b * p.q_sb + h * dh;
  const bf16_t* dob = p.d_o + (long)b * p.o_sb + h * dh;
  const bf16_t* ob = p.o_in + (long)b * p.o_sb + h * dh;
  bf16x8 qf[3], dof[3];
  float sacc = 0.f;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    qf[ks] = frag_global32_dh(qb, p.q_st, row0, active ? p.Tq : 0, ks, lane, dh);
    dof[ks] = frag_global32_dh(dob, p.o_st, row0, active ? p.Tq : 0, ks, lane, dh);
    const bf16x8 of = frag_global32_dh(ob, p.o_st, row0, active ? p.Tq : 0, ks, lane, dh);
#pragma unroll
    for (int e = 0; e < 8; ++e) sacc += bf16_to_f32((bf16_t)dof[ks][e]) * bf16_to_f32((bf16_t)of[e]);
  }
  const float delta = sacc + __shfl_xor(sacc, 32, 64);  // the two halves of a row's reduction sit in lanes l and l ^ 32
  if (FOLD) {  // dh == 40: element 0 / 1 of the (ks = 2, hk = 1) fragment are columns 40 / 41
    const bf16_t hi = f32_to_bf16(-delta);
    const bf16_t lo = f32_to_bf16(-delta - bf16_to_f32(hi));
    if (lane >= 32) { dof[2][0] = (short)hi; dof[2][1] = (short)lo; }
    for (int idx = threadIdx.x; idx < A2_CH * 8; idx += 512) {  // the pad chunks of the V tile, once: ones in columns 40 / 41
      const int row = idx >> 3, chunk = idx & 7;
      if (chunk * 8 >= dh)
        *reinterpret_cast<u32x4*>(Vs + row * 128 + ((chunk ^ swz(row)) << 4)) = u32x4{chunk * 8 == dh ? 0x3F803F80u : 0u, 0u, 0u, 0u};
    }
  }
  const int qrow = row0 + (lane & 31);
  const bool qvalid = active && qrow < p.Tq;
  const long stat = ((long)b * p.H + h) * p.Tq + qrow;
  if (qvalid && lane < 32) p.delta[stat] = delta;
  const float lse2 = qvalid ? p.lse[stat] * LOG2E : INFINITY;
  const float sl2 = p.scale * LOG2E;
  f32x4 dqt[2][NDT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dqt[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Lane-constant parts of the twelve LDS addresses of a block (the swizzle key of row a*32 + r is that of r; the V tile sits A2_TILE
  // bytes behind the K tile, the second transposing read 8 rows = 1 024 bytes behind the first): a block adds its a * 4 096 bytes to SIX
  // of them — the compiler's own form added it to all twelve, 12 of the loop's 72 VALU instructions.
  // (kept as LDS-space pointers: through generic pointers every load paid a `v_add_u32 v, 0, v` for the address-space cast)
  typedef __attribute__((address_space(3))) const char* lds_cp;
  lds_cp kr[3], kc[NDT];
  {
    const lds_cp ks3 = (lds_cp)LDS_PTR(Ks);
    const int r = lane & 31;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) kr[ks] = ks3 + r * 128 + (((ks * 2 + (lane >> 5)) ^ swz(r)) << 4);
    const int s16 = lane & 15;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) kc[dt] = ks3 + tile_off((g & 1) * 16 + (g >> 1) * 4 + (s16 >> 2), dt * 16 + 4 * (s16 & 3));
    // opaque: the tile's base is a link-time constant that hipcc otherwise adds at every USE (`v_add_u32 v, 0, v`) beside the loop's increments
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) { unsigned u = (unsigned)(uintptr_t)kr[ks]; asm volatile("" : "+v"(u)); kr[ks] = (lds_cp)(uintptr_t)u; }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { unsigned u = (unsigned)(uintptr_t)kc[dt]; asm volatile("" : "+v"(u)); kc[dt] = (lds_cp)(uintptr_t)u; }
  }
  for (int kv0 = 0; kv0 < p.Tk; kv0 += A2_CH) {
    const int rows = min(A2_CH, p.Tk - kv0);
    __syncthreads();
    {
      int ln = lane;  // opaque copy: see attn_fwd2_kernel
      asm volatile("" : "+v"(ln));
      dma_oper<1>(Ks, p.k + (long)b * p.kv_sb + (long)kv0 * p.kv_st + h * dh, p.kv_st, rows, dh, wave, 8, ln);
      dma_oper<1, FOLD>(Vs, p.v + (long)b * p.kv_sb + (long)kv0 * p.kv_st + h * dh, p.kv_st, rows, dh, wave, 8, ln);
    }
    lds_dma_wait_all();
    __syncthreads();
    if (!active) continue;
    const int nbl = (rows + 31) / 32;
#pragma unroll 1
    for (int a = 0; a < nbl; ++a) {
      f32x16 sc, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { sc[e] = 0.f; dp[e] = 0.f; }
      const int aoff = a * (32 * 128);
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const lds_cp kp = kr[ks] + aoff;
        typedef __attribute__((address_space(3))) const bf16x8* lds_v8;
        sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(lds_v8)kp, qf[ks], sc, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(lds_v8)(kp + A2_TILE), dof[ks], dp, 0, 0, 0);
      }
      // lane: query row0 + (l & 31), keys a*32 + 8 j + 4 (l >> 5) + r.  Key rows beyond Tk are zero rows of the chunk: p finite, dP = 0
#pragma unroll
      for (int e = 0; e < 16; ++e) sc[e] = __builtin_amdgcn_exp2f(fmaf(sc[e], sl2, -lse2)) * (FOLD ? dp[e] : dp[e] - delta);
      bf16x8 ds0, ds1;
      swap32_operands(sc, ds0, ds1);
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const lds_cp cp = kc[dt] + aoff;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)cp);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(cp + 8 * 128));
        const bf16x8 kcf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dqt[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcf, ds0, dqt[0][dt], 0, 0, 0);
        dqt[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kcf, ds1, dqt[1][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int qi = row0 + 16 * t + i;
    if (active && qi < p.Tq) {
      bf16_t* dqrow = p.dq + (long)b * p.q_sb + (long)qi * p.q_st + h * dh;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const int col = dt * 16 + 4 * g;
        if (col >= dh) continue;
        const f32x4 v = dqt[t][dt] * p.scale;
        *reinterpret_cast<u32x2*>(dqrow + col) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
      }
    }
  }
}

// waves per workgroup: ONE workgroup per (batch, head) (the resident K / V — or Q / dO — tiles are
// loaded once), its waves walk the 16-row tiles in rounds; pick the wave count that leaves the
// fewest idle slots in the last round (at most 8 waves).
inline int pick_waves(int T) {
  const int tiles = (T + 15) / 16;
  const int rounds = (tiles + 7) / 8;
  return (tiles + rounds - 1) / rounds;
}

int check_common(const char* who, const void* q, const void* k, const void* v, int B, int H, int Tq, int Tk,
                 int64_t q_sb, int64_t q_st, int64_t kv_sb, int64_t kv_st, int64_t o_sb, int64_t o_st) {
  CFHIP_REQUIRE(q && k && v, "%s: null q/k/v", who);
  CFHIP_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "%s: empty problem", who);
  CFHIP_REQUIRE((long)B * 1 <= 65535 && H <= 65535, "%s: B = %d / H = %d exceed the grid limits", who, B, H);
  CFHIP_REQUIRE(q_sb % 8 == 0 && q_st % 8 == 0 && kv_sb % 8 == 0 && kv_st % 8 == 0 && o_sb % 8 == 0 &&
                    o_st % 8 == 0,
                "%s: strides must be multiples of 8 elements", who);
  CFHIP_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0,
                "%s: q/k/v must be 16-byte aligned", who);
  return CFHIP_OK;
}

// delta[b][h][t] = sum_d dO[b][t][h][d] * O[b][t][h][d]: what the dQ pass leaves behind for the dK / dV pass; launched on its
// own when the dK / dV pass of the two-slot kernels runs without a dQ pass in front of it (one thread per row).
__global__ void attn_delta_kernel(AttnParams p) {
  const long total = (long)p.B * p.H * p.Tq;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % p.Tq);
    const long bh = idx / p.Tq;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const bf16_t* dor = p.d_o + (long)b * p.o_sb + (long)t * p.o_st + h * p.dh;
    const bf16_t* orr = p.o_in + (long)b * p.o_sb + (long)t * p.o_st + h * p.dh;
    float s = 0.f;
    for (int sl = 0; sl < p.dh / 8; ++sl) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(dor + sl * 8);
      const u32x4 c = *reinterpret_cast<const u32x4*>(orr + sl * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) s += bf16lo(a[e]) * bf16lo(c[e]) + bf16hi(a[e]) * bf16hi(c[e]);
    }
    p.delta[idx] = s;
  }
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char* who) {
  if (bytes > 65536) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      cfhip_set_error("%s: cannot reserve %zu bytes of LDS: %s", who, bytes, hipGetErrorString(e));
      return CFHIP_ERR_LAUNCH;
    }
  }
  return CFHIP_OK;
}

template <typename K>
int set_lds(K kernel, size_t bytes, const char* who);

// dropout variants: same geometry, DROP instantiations
template <int NH>
int launch_gen_fwd_drop(const AttnParams& p, bool plain, hipStream_t s) {
  dim3 grid((p.Tq + 127) / 128, p.H, p.B);
  const size_t lds = (size_t)Gen<NH>::NSLOT * Gen<NH>::SLOT_BYTES;
  int rc = plain ? set_lds(attn_gen_fwd_kernel<NH, true, true>, lds, "attn_fwd") : set_lds(attn_gen_fwd_kernel<NH, false, true>, lds, "attn_fwd");
  if (rc != CFHIP_OK) return rc;
  if (plain) hipLaunchKernelGGL((attn_gen_fwd_kernel<NH, true, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((attn_gen_fwd_kernel<NH, false, true>), grid, dim3(512), lds, s, p);
  CFHIP_CHECK_LAUNCH("attn_gen_fwd(dropout)");
  return CFHIP_OK;
}

template <int NH>
int launch_gen_bwd_drop(const AttnParams& p, bool plain, int parts, hipStream_t s) {
  if (parts & 1) {
    dim3 grid((p.Tq + 127) / 128, p.H, p.B);
    const size_t lds = (size_t)Gen<NH>::NSLOT * Gen<NH>::SLOT_BYTES;
    int rc = plain ? set_lds(attn_gen_bwd_dq_kernel<NH, true, true>, lds, "attn_bwd_dq")
                   : set_lds(attn_gen_bwd_dq_kernel<NH, false, true>, lds, "attn_bwd_dq");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL((attn_gen_bwd_dq_kernel<NH, true, true>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((attn_gen_bwd_dq_kernel<NH, false, true>), grid, dim3(512), lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_gen_bwd_dq(dropout)");
  }
  if (parts & 2) {
    if (Gen<NH>::NSLOT == 2 && !p.delta_ready && !(parts & 1)) {
      const long rows = (long)p.B * p.H * p.Tq;
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 255) / 256 > 4096 ? 4096 : (rows + 255) / 256)), dim3(256), 0, s, p);
      CFHIP_CHECK_LAUNCH("attn_delta");
    }
    dim3 grid((p.Tk + 127) / 128, p.H, p.B);
    const size_t lds = (size_t)Gen<NH>::NSLOT * (Gen<NH>::SLOT_BYTES + (size_t)2 * Gen<NH>::CH * sizeof(float));
    int rc = plain ? set_lds(attn_gen_bwd_dkv_kernel<NH, true, true>, lds, "attn_bwd_dkv")
                   : set_lds(attn_gen_bwd_dkv_kernel<NH, false, true>, lds, "attn_bwd_dkv");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL((attn_gen_bwd_dkv_kernel<NH, true, true>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((attn_gen_bwd_dkv_kernel<NH, false, true>), grid, dim3(512), lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_gen_bwd_dkv(dropout)");
  }
  return CFHIP_OK;
}

int g_attn_short_max = CFHIP_ATTN_MAX_T;  // "attn_short_max" option: head_dim-64 sequences up to this length take the LDS-resident kernels
int g_attn_two_tiles = 511;  // "attn_two_tiles" option (bits): 1 forward, 2 dQ pass (4: also head_dim > 48), 8 dK / dV pass of head_dim <= 64 on the two-tiles-per-wave kernels; 16: forward row sums through the ones column when head_dim = 8 mod 16; 32: dK / dV pass of head_dim 72 .. 96 on the two-tile kernel; 64 (round 6): the plain head_dim <= 48 dQ pass with S / dP on 32x32x16 MFMAs (attn_bwd_dq3_kernel); 128 (round 6): head_dim 40, plain: dP - delta out of the matrix pipe (two spare reduction columns) in that dQ pass and in the dK / dV pass; 256 (round 6, late): head_dim 72 .. 96 without a mask: forward and dQ pass on the two-tile kernels (Fwd2<5 / 6>: two 64-column halves, 4 waves, 128-key chunks)

extern int g_attn_two_tiles;
template <int NDT>
int launch_fwd2(const AttnParams& p, bool plain, hipStream_t s) {
  using F = Fwd2<NDT>;
  constexpr int ROWS = F::NW * 32;
  dim3 grid((p.Tq + ROWS - 1) / ROWS, p.H, p.B);
  const size_t lds = (size_t)2 * F::NH * F::HB;
  const dim3 block(F::NW * 64);
  if constexpr (F::NH == 2) {  // head_dim 72 .. 96: the plain form only (masked / causal sequences of that width stay on the general kernel)
    const int rc = set_lds(attn_fwd2_kernel<true, NDT>, lds, "attn_fwd");
    if (rc != CFHIP_OK) return rc;
    hipLaunchKernelGGL((attn_fwd2_kernel<true, NDT>), grid, block, lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_fwd2");
    return CFHIP_OK;
  } else {
  if (plain && p.dh == (NDT - 1) * 16 + 8 && (g_attn_two_tiles & 16)) {  // a free column in the last tile: row sums by MFMA
    const int rc = set_lds(attn_fwd2_kernel<true, NDT, true>, lds, "attn_fwd");
    if (rc != CFHIP_OK) return rc;
    hipLaunchKernelGGL((attn_fwd2_kernel<true, NDT, true>), grid, block, lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_fwd2");
    return CFHIP_OK;
  }
  int rc = plain ? set_lds(attn_fwd2_kernel<true, NDT>, lds, "attn_fwd") : set_lds(attn_fwd2_kernel<false, NDT>, lds, "attn_fwd");
  if (rc != CFHIP_OK) return rc;
  if (plain) hipLaunchKernelGGL((attn_fwd2_kernel<true, NDT>), grid, block, lds, s, p);
  else hipLaunchKernelGGL((attn_fwd2_kernel<false, NDT>), grid, block, lds, s, p);
  CFHIP_CHECK_LAUNCH("attn_fwd2");
  return CFHIP_OK;
  }
}

template <int NH>
int launch_gen_fwd(const AttnParams& p, bool plain, hipStream_t s) {
  if (NH == 1 && (g_attn_two_tiles & 1)) return p.dh <= 48 ? launch_fwd2<3>(p, plain, s) : launch_fwd2<4>(p, plain, s);
  if (NH == 2 && plain && p.dh <= 96 && (g_attn_two_tiles & 256)) return p.dh <= 80 ? launch_fwd2<5>(p, true, s) : launch_fwd2<6>(p, true, s);
  dim3 grid((p.Tq + 127) / 128, p.H, p.B);
  const size_t lds = (size_t)Gen<NH>::NSLOT * Gen<NH>::SLOT_BYTES;
  int rc = plain ? set_lds(attn_gen_fwd_kernel<NH, true>, lds, "attn_fwd") : set_lds(attn_gen_fwd_kernel<NH, false>, lds, "attn_fwd");
  if (rc != CFHIP_OK) return rc;
  if (plain) hipLaunchKernelGGL((attn_gen_fwd_kernel<NH, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((attn_gen_fwd_kernel<NH, false>), grid, dim3(512), lds, s, p);
  CFHIP_CHECK_LAUNCH("attn_gen_fwd");
  return CFHIP_OK;
}

template <int NDT>
int launch_dq2(const AttnParams& p, bool plain, hipStream_t s) {
  using F = Fwd2<NDT>;
  constexpr int ROWS = F::NW * 32;
  dim3 grid((p.Tq + ROWS - 1) / ROWS, p.H, p.B);
  const size_t lds = (size_t)2 * F::NH * F::HB;
  const dim3 block(F::NW * 64);
  if constexpr (F::NH == 2) {  // head_dim 72 .. 96: the plain form only
    const int rc = set_lds(attn_bwd_dq2_kernel<true, NDT>, lds, "attn_bwd_dq");
    if (rc != CFHIP_OK) return rc;
    hipLaunchKernelGGL((attn_bwd_dq2_kernel<true, NDT>), grid, block, lds, s, p);
  } else {
    int rc = plain ? set_lds(attn_bwd_dq2_kernel<true, NDT>, lds, "attn_bwd_dq") : set_lds(attn_bwd_dq2_kernel<false, NDT>, lds, "attn_bwd_dq");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL((attn_bwd_dq2_kernel<true, NDT>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((attn_bwd_dq2_kernel<false, NDT>), grid, block, lds, s, p);
  }
  CFHIP_CHECK_LAUNCH("attn_bwd_dq2");
  return CFHIP_OK;
}

int launch_dq3(const AttnParams& p, hipStream_t s) {
  dim3 grid((p.Tq + A2_ROWS - 1) / A2_ROWS, p.H, p.B);
  const size_t lds = (size_t)2 * A2_TILE;
  const bool fold = p.dh == 40 && (g_attn_two_tiles & 128);
  const int rc = fold ? set_lds(attn_bwd_dq3_kernel<3, true>, lds, "attn_bwd_dq") : set_lds(attn_bwd_dq3_kernel<3, false>, lds, "attn_bwd_dq");
  if (rc != CFHIP_OK) return rc;
  if (fold) hipLaunchKernelGGL((attn_bwd_dq3_kernel<3, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((attn_bwd_dq3_kernel<3, false>), grid, dim3(512), lds, s, p);
  CFHIP_CHECK_LAUNCH("attn_bwd_dq3");
  return CFHIP_OK;
}

template <int NDT>
int launch_dkv2(const AttnParams& p, bool plain, hipStream_t s) {
  dim3 grid((p.Tk + 127) / 128, p.H, p.B);
  const size_t lds = (size_t)2 * ((NDT + 3) / 4) * A2D_TILE + (size_t)2 * A2D_CH * sizeof(float);
  if (lds > 64 * 1024) {
    const int rc = plain ? set_lds(attn_bwd_dkv2_kernel<true, NDT>, lds, "attn_bwd_dkv") : set_lds(attn_bwd_dkv2_kernel<false, NDT>, lds, "attn_bwd_dkv");
    if (rc != CFHIP_OK) return rc;
  }
  if constexpr (NDT == 3) {
    if (plain && p.dh == 40 && (g_attn_two_tiles & 128)) {
      hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true, 3, true>), grid, dim3(256), lds, s, p);
      CFHIP_CHECK_LAUNCH("attn_bwd_dkv2");
      return CFHIP_OK;
    }
  }
  if (plain) hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true, NDT>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((attn_bwd_dkv2_kernel<false, NDT>), grid, dim3(256), lds, s, p);
  CFHIP_CHECK_LAUNCH("attn_bwd_dkv2");
  return CFHIP_OK;
}

template <int NH>
int launch_gen_bwd(const AttnParams& p, bool plain, int parts, hipStream_t s) {
  if ((parts & 1) && NH == 1 && (g_attn_two_tiles & 2) && (p.dh <= 48 || (g_attn_two_tiles & 4))) {  // (the 4-column-tile form spills: bit 4 to try it)
    const int rc = p.dh <= 48 ? ((plain && (g_attn_two_tiles & 64)) ? launch_dq3(p, s) : launch_dq2<3>(p, plain, s)) : launch_dq2<4>(p, plain, s);
    if (rc != CFHIP_OK) return rc;
  } else if ((parts & 1) && NH == 2 && plain && p.dh <= 96 && (g_attn_two_tiles & 256)) {  // head_dim 72 .. 96 on the two-tile form (round 6, late)
    const int rc = p.dh <= 80 ? launch_dq2<5>(p, true, s) : launch_dq2<6>(p, true, s);
    if (rc != CFHIP_OK) return rc;
  } else if (parts & 1) {
    dim3 grid((p.Tq + 127) / 128, p.H, p.B);
    const size_t lds = (size_t)Gen<NH>::NSLOT * Gen<NH>::SLOT_BYTES;
    int rc = plain ? set_lds(attn_gen_bwd_dq_kernel<NH, true>, lds, "attn_bwd_dq")
                   : set_lds(attn_gen_bwd_dq_kernel<NH, false>, lds, "attn_bwd_dq");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL((attn_gen_bwd_dq_kernel<NH, true>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((attn_gen_bwd_dq_kernel<NH, false>), grid, dim3(512), lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_gen_bwd_dq");
  }
  if ((parts & 2) && (g_attn_two_tiles & 8) && (NH == 1 || (NH == 2 && p.dh <= 96 && (g_attn_two_tiles & 32)))) {
    if (!(parts & 1) || (NH == 2 && !p.delta_ready)) {  // no (delta-writing) dQ pass in front of this one: fill delta
      const long rows = (long)p.B * p.H * p.Tq;
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 255) / 256 > 4096 ? 4096 : (rows + 255) / 256)), dim3(256), 0, s, p);
      CFHIP_CHECK_LAUNCH("attn_delta");
    }
    const int rc = p.dh <= 48 ? launch_dkv2<3>(p, plain, s) : p.dh <= 64 ? launch_dkv2<4>(p, plain, s)
                 : p.dh <= 80 ? launch_dkv2<5>(p, plain, s) : launch_dkv2<6>(p, plain, s);
    if (rc != CFHIP_OK) return rc;
  } else if (parts & 2) {
    if (Gen<NH>::NSLOT == 2 && !p.delta_ready && !(parts & 1)) {
      const long rows = (long)p.B * p.H * p.Tq;
      hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows + 255) / 256 > 4096 ? 4096 : (rows + 255) / 256)), dim3(256), 0, s, p);
      CFHIP_CHECK_LAUNCH("attn_delta");
    }
    dim3 grid((p.Tk + 127) / 128, p.H, p.B);
    const size_t lds = (size_t)Gen<NH>::NSLOT * (Gen<NH>::SLOT_BYTES + (size_t)2 * Gen<NH>::CH * sizeof(float));
    int rc = plain ? set_lds(attn_gen_bwd_dkv_kernel<NH, true>, lds, "attn_bwd_dkv")
                   : set_lds(attn_gen_bwd_dkv_kernel<NH, false>, lds, "attn_bwd_dkv");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL((attn_gen_bwd_dkv_kernel<NH, true>), grid, dim3(512), lds, s, p);
    else hipLaunchKernelGGL((attn_gen_bwd_dkv_kernel<NH, false>), grid, dim3(512), lds, s, p);
    CFHIP_CHECK_LAUNCH("attn_gen_bwd_dkv");
  }
  return CFHIP_OK;
}

// fills the dropout fields; false = no dropout (p <= 0).  The probability is quantised to 1/256 (as flash attention
// implementations do): thresh = round(256 p), kept values are scaled by 1 / (1 - thresh / 256).
bool set_dropout(AttnParams& p, float dropout_p, uint64_t seed, uint64_t offset) {
  if (!(dropout_p > 0.f)) return false;
  int t = (int)lrintf(dropout_p * 256.0f);
  t = t < 1 ? 1 : (t > 255 ? 255 : t);
  p.drop_thresh = (unsigned)t;
  p.drop_scale = 256.0f / (float)(256 - t);
  p.drop_seed = seed;
  p.drop_offset = offset;
  p.drop_bq = (p.Tq + 3) / 4;
  p.drop_bk = (p.Tk + 3) / 4;
  return true;
}

int g_attn_pers_ctas = 256;  // workgroups of a persistent launch (one per CU; fewer leave CUs to the other queues)
int g_attn_one_pass = 2;  // option "attn_one_pass": 0 = the two-pass kernels; 1 (round 6) = attn_bwd_one_kernel when both passes are asked for in one call (PLAIN or causal self-attention, 32 < T <= 224); 2 (default) = that, with 128 < T <= 224 without a mask on attn_bwd_one2_kernel (8 waves, two key tiles per wave; bit-identical results)
int g_attn_persistent = 7;  // bit 0: dK / dV pass, bit 1: dQ pass, bit 2: forward — as persistent 16-wave workgroups when 128 < T <= 256 (the ViT shape)

int check_head_dim(const char* who, int head_dim) {
  CFHIP_REQUIRE(head_dim >= 8 && head_dim <= 192 && head_dim % 8 == 0,
                "%s: head_dim %d is not a multiple of 8 in [8, 192]", who, head_dim);
  return CFHIP_OK;
}

}  // namespace

int cfhip_internal_set_attn_short_max(int v) {
  g_attn_short_max = v < 0 ? 0 : (v > CFHIP_ATTN_MAX_T ? CFHIP_ATTN_MAX_T : v);
  return CFHIP_OK;
}

int cfhip_internal_set_attn_two_tiles(int v) {
  g_attn_two_tiles = v;
  return CFHIP_OK;
}

int cfhip_internal_set_attn_pers_ctas(int v) {
  g_attn_pers_ctas = v < 1 ? 1 : v;
  return CFHIP_OK;
}

int cfhip_internal_set_attn_one_pass(int v) {
  g_attn_one_pass = v;
  return CFHIP_OK;
}
int cfhip_internal_set_attn_persistent(int v) {
  g_attn_persistent = v;
  return CFHIP_OK;
}

#ifdef CFHIP_ABLATE
int cfhip_internal_set_attn_ablate(int v) {
  g_attn_ablate = v;
  return CFHIP_OK;
}
#endif

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, float* lse,
                         const uint8_t* mask, int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b,
                         int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                         int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                         int64_t ms_q, float scale, int causal, void* stream, float dropout_p = 0.f,
                         uint64_t seed = 0, uint64_t offset = 0) {
  int rc = check_head_dim("attn_fwd", head_dim);
  if (rc != CFHIP_OK) return rc;
  rc = check_common("attn_fwd", q, k, v, B, H, Tq, Tk, q_stride_b, q_stride_t, kv_stride_b,
                        kv_stride_t, o_stride_b, o_stride_t);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(o && ((uintptr_t)o & 7) == 0, "attn_fwd: o must be non-null and 8-byte aligned");
  AttnParams p = {};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
  p.lse = lse; p.mask = mask;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.q_sb = q_stride_b; p.q_st = q_stride_t; p.kv_sb = kv_stride_b; p.kv_st = kv_stride_t;
  p.o_sb = o_stride_b; p.o_st = o_stride_t;
  p.ms_b = ms_b; p.ms_h = ms_h; p.ms_q = ms_q;
  p.scale = scale; p.causal = causal;
#ifdef CFHIP_ABLATE
  p.ablate = g_attn_ablate;
#endif
  p.dh = head_dim;
  const int nb = (Tk + 31) / 32;
  const int nw = pick_waves(Tq);
  const int tiles = (Tq + 15) / 16;
  (void)tiles;
  dim3 grid(1, H, B), block(nw * 64);
  const size_t lds = (size_t)2 * nb * 32 * 128;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool plain = mask == nullptr && !causal;
  if (set_dropout(p, dropout_p, seed, offset)) {  // dropout on the probabilities: the general kernels' DROP forms
    switch ((head_dim + 63) / 64) {
      case 1: return launch_gen_fwd_drop<1>(p, plain, s);
      case 2: return launch_gen_fwd_drop<2>(p, plain, s);
      default: return launch_gen_fwd_drop<3>(p, plain, s);
    }
  }
  if (head_dim != CFHIP_ATTN_HEAD_DIM || Tq > g_attn_short_max || Tk > g_attn_short_max) {
    // general form: chunked K / V with online softmax, head_dim as zero-padded 64-column halves
    switch ((head_dim + 63) / 64) {
      case 1: return launch_gen_fwd<1>(p, plain, s);
      case 2: return launch_gen_fwd<2>(p, plain, s);
      default: return launch_gen_fwd<3>(p, plain, s);
    }
  }
  if (plain && (g_attn_persistent & 4) && Tq > 128 && Tq <= PERS_WAVES * 16 && Tk > 128 && Tk <= 256 && lse != nullptr) {
    const size_t plds = (size_t)2 * 2 * 256 * 128;
    const int heads = B * H;
    dim3 pgrid(heads < g_attn_pers_ctas ? heads : g_attn_pers_ctas), pblock(PERS_WAVES * 64);
    int prc = CFHIP_OK;
#define CFHIP_FWD_PERS(NB_)                                                                          \
  case NB_:                                                                                          \
    prc = set_lds(attn_fwd_pers_kernel<NB_>, plds, "attn_fwd");                                      \
    if (prc != CFHIP_OK) return prc;                                                                 \
    hipLaunchKernelGGL(attn_fwd_pers_kernel<NB_>, pgrid, pblock, plds, s, p);                        \
    break;
    switch (nb) {
      CFHIP_FWD_PERS(5) CFHIP_FWD_PERS(6) CFHIP_FWD_PERS(7) CFHIP_FWD_PERS(8)
      default: cfhip_set_error("attn_fwd: bad nb %d", nb); return CFHIP_ERR_INVALID;
    }
#undef CFHIP_FWD_PERS
    CFHIP_CHECK_LAUNCH("attn_fwd(persistent)");
    return CFHIP_OK;
  }
#define CFHIP_ATTN_FWD(NB_)                                                                         \
  case NB_:                                                                                         \
    if (plain) hipLaunchKernelGGL((attn_fwd_kernel<NB_, true>), grid, block, lds, s, p);            \
    else hipLaunchKernelGGL((attn_fwd_kernel<NB_, false>), grid, block, lds, s, p);                 \
    break;
  switch (nb) {
    CFHIP_ATTN_FWD(1) CFHIP_ATTN_FWD(2) CFHIP_ATTN_FWD(3) CFHIP_ATTN_FWD(4)
    CFHIP_ATTN_FWD(5) CFHIP_ATTN_FWD(6) CFHIP_ATTN_FWD(7) CFHIP_ATTN_FWD(8)
    default: cfhip_set_error("attn_fwd: bad nb %d", nb); return CFHIP_ERR_INVALID;
  }
#undef CFHIP_ATTN_FWD
  CFHIP_CHECK_LAUNCH("attn_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                              const uint8_t* mask, int B, int H, int Tq, int Tk, int64_t q_stride_b,
                              int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                              int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                              int64_t ms_q, float scale, int causal, void* stream) {
  return attn_fwd_impl(q, k, v, o, lse, mask, B, H, Tq, Tk, CFHIP_ATTN_HEAD_DIM, q_stride_b, q_stride_t, kv_stride_b,
                       kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal, stream);
}

extern "C" int cfhip_attn_fwd_dh(const void* q, const void* k, const void* v, void* o, float* lse,
                                 const uint8_t* mask, int B, int H, int Tq, int Tk, int head_dim,
                                 int64_t q_stride_b, int64_t q_stride_t, int64_t kv_stride_b,
                                 int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b,
                                 int64_t ms_h, int64_t ms_q, float scale, int causal, void* stream) {
  return attn_fwd_impl(q, k, v, o, lse, mask, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t, kv_stride_b,
                       kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal, stream);
}

template <int NB, int PW, int PER_CU>
static int launch_one_pass(const AttnParams& p, bool causal, hipStream_t s) {
  const size_t lds = (size_t)3 * NB * 32 * 128 + (size_t)2 * NB * 32 * 4 + (size_t)((NB + 1) / 2) * 32 * (NB * 64 + 16);
  const int heads = p.B * p.H, ctas = g_attn_pers_ctas * PER_CU;
  const dim3 grid(heads < ctas ? heads : ctas), block(PW * 64);
  if constexpr (NB >= 3 && NB <= 5) {
    if (causal) {
      const int rc = set_lds(attn_bwd_one_kernel<NB, PW, true>, lds, "attn_bwd_one");
      if (rc != CFHIP_OK) return rc;
      hipLaunchKernelGGL((attn_bwd_one_kernel<NB, PW, true>), grid, block, lds, s, p);
      return CFHIP_OK;
    }
  }
  if constexpr (NB >= 5) {
    if (!causal && g_attn_one_pass >= 2) {  // two key tiles per wave: 8 waves
      const int rc = set_lds(attn_bwd_one2_kernel<NB>, lds, "attn_bwd_one");
      if (rc != CFHIP_OK) return rc;
      hipLaunchKernelGGL((attn_bwd_one2_kernel<NB>), grid, dim3(512), lds, s, p);
      return CFHIP_OK;
    }
  }
  const int rc = set_lds(attn_bwd_one_kernel<NB, PW, false>, lds, "attn_bwd_one");
  if (rc != CFHIP_OK) return rc;
  hipLaunchKernelGGL((attn_bwd_one_kernel<NB, PW, false>), grid, block, lds, s, p);
  return CFHIP_OK;
}

static int attn_bwd_impl(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                         const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk,
                         void* dv, int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b,
                         int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                         int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                         int64_t ms_q, float scale, int causal, int parts, void* stream, float dropout_p = 0.f,
                         uint64_t seed = 0, uint64_t offset = 0) {
  int rc = check_head_dim("attn_bwd", head_dim);
  if (rc != CFHIP_OK) return rc;
  rc = check_common("attn_bwd", q, k, v, B, H, Tq, Tk, q_stride_b, q_stride_t, kv_stride_b,
                        kv_stride_t, o_stride_b, o_stride_t);
  if (rc != CFHIP_OK) return rc;
  CFHIP_REQUIRE(o && d_o && lse && delta && dq && dk && dv, "attn_bwd: null pointer");
  CFHIP_REQUIRE(((uintptr_t)o & 15) == 0 && ((uintptr_t)d_o & 15) == 0 && ((uintptr_t)dq & 7) == 0 &&
                    ((uintptr_t)dk & 7) == 0 && ((uintptr_t)dv & 7) == 0,
                "attn_bwd: misaligned tensors");
  AttnParams p = {};
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v;
  p.o_in = (const bf16_t*)o; p.d_o = (const bf16_t*)d_o;
  p.lse = const_cast<float*>(lse); p.delta = delta; p.mask = mask;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.q_sb = q_stride_b; p.q_st = q_stride_t; p.kv_sb = kv_stride_b; p.kv_st = kv_stride_t;
  p.o_sb = o_stride_b; p.o_st = o_stride_t;
  p.ms_b = ms_b; p.ms_h = ms_h; p.ms_q = ms_q;
  p.scale = scale; p.causal = causal;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  p.delta_ready = (parts & 3) == 3;
#ifdef CFHIP_ABLATE
  p.ablate = g_attn_ablate;
#endif
  p.dh = head_dim;
  const bool plain = mask == nullptr && !causal;
  CFHIP_REQUIRE((parts & 3) != 0, "attn_bwd: parts must select the dQ pass (1), the dK/dV pass (2) or both (3)");
  if (set_dropout(p, dropout_p, seed, offset)) {
    switch ((head_dim + 63) / 64) {
      case 1: return launch_gen_bwd_drop<1>(p, plain, parts, s);
      case 2: return launch_gen_bwd_drop<2>(p, plain, parts, s);
      default: return launch_gen_bwd_drop<3>(p, plain, parts, s);
    }
  }
  if (head_dim != CFHIP_ATTN_HEAD_DIM || Tq > g_attn_short_max || Tk > g_attn_short_max) {  // general form
    switch ((head_dim + 63) / 64) {
      case 1: return launch_gen_bwd<1>(p, plain, parts, s);
      case 2: return launch_gen_bwd<2>(p, plain, parts, s);
      default: return launch_gen_bwd<3>(p, plain, parts, s);
    }
  }
  const int nb1 = (Tq + 31) / 32;
  if ((parts & 3) == 3 && mask == nullptr && dropout_p == 0.f && g_attn_one_pass && Tq == Tk && Tq > 32 && Tq <= 224 && o != nullptr &&
      (!causal || (nb1 >= 3 && nb1 <= 5))) {
    // dQ, dK and dV of a head from one evaluation of S and dP (attn_bwd_one_kernel): 16-wave workgroups, one per CU, for the ViT
    // lengths; 4 / 8 waves and several workgroups per CU for the short sequences (ViT-B/32 in CLIP: T = 50; its causal text tower:
    // T = 77).  Causal instantiations exist for 64 < T <= 160 (the others would spill two registers: the two-pass kernels take them).
    switch (nb1) {
      case 2: rc = launch_one_pass<2, 4, 4>(p, causal != 0, s); break;
      case 3: rc = launch_one_pass<3, 8, 2>(p, causal != 0, s); break;
      case 4: rc = launch_one_pass<4, 8, 2>(p, causal != 0, s); break;
      case 5: rc = launch_one_pass<5, 16, 1>(p, causal != 0, s); break;
      case 6: rc = launch_one_pass<6, 16, 1>(p, causal != 0, s); break;
      case 7: rc = launch_one_pass<7, 16, 1>(p, causal != 0, s); break;
      default: cfhip_set_error("attn_bwd: bad nb %d", nb1); return CFHIP_ERR_INVALID;
    }
    if (rc != CFHIP_OK) return rc;
    CFHIP_CHECK_LAUNCH("attn_bwd(one pass)");
    return CFHIP_OK;
  }
  if ((parts & 1) && plain && (g_attn_persistent & 2) && Tq > 128 && Tq <= PERS_WAVES * 16 && Tk > 128 && Tk <= 256) {
    const int nb = (Tk + 31) / 32;
    const size_t lds = (size_t)2 * 2 * 256 * 128;
    const int heads = B * H;
    dim3 grid(heads < g_attn_pers_ctas ? heads : g_attn_pers_ctas), block(PERS_WAVES * 64);
#define CFHIP_DQ_PERS(NB_)                                                                                           \
  case NB_:                                                                                                          \
    rc = set_lds(attn_bwd_dq_pers_kernel<NB_>, lds, "attn_bwd_dq");                                                 \
    if (rc != CFHIP_OK) return rc;                                                                                  \
    hipLaunchKernelGGL(attn_bwd_dq_pers_kernel<NB_>, grid, block, lds, s, p);                                       \
    break;
    switch (nb) {
      CFHIP_DQ_PERS(5) CFHIP_DQ_PERS(6) CFHIP_DQ_PERS(7) CFHIP_DQ_PERS(8)
      default: cfhip_set_error("attn_bwd: bad nb %d", nb); return CFHIP_ERR_INVALID;
    }
#undef CFHIP_DQ_PERS
    CFHIP_CHECK_LAUNCH("attn_bwd_dq(persistent)");
  } else if (parts & 1) {
    const int nb = (Tk + 31) / 32;
    const int nw = pick_waves(Tq);
    const int tiles = (Tq + 15) / 16;
    (void)tiles;
    dim3 grid(1, H, B), block(nw * 64);
    const size_t lds = (size_t)2 * nb * 32 * 128;
    if (plain) hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, block, lds, s, p, nb);
    else hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, block, lds, s, p, nb);
    CFHIP_CHECK_LAUNCH("attn_bwd_dq");
  }
  if ((parts & 2) && plain && p.delta_ready && (g_attn_persistent & 1) && Tq > 128 && Tk <= PERS_WAVES * 16) {
    // the ViT shape: persistent workgroups, next head's Q / dO streaming in behind the current one
    const int nbq = (Tq + 31) / 32;
    const size_t lds = 2 * ((size_t)2 * 256 * 128 + (size_t)2 * nbq * 32 * sizeof(float));
    const int heads = B * H;
    dim3 grid(heads < g_attn_pers_ctas ? heads : g_attn_pers_ctas), block(PERS_WAVES * 64);
#define CFHIP_DKV_PERS(NBQ_)                                                                                         \
  case NBQ_:                                                                                                         \
    rc = set_lds(attn_bwd_dkv_pers_kernel<NBQ_>, lds, "attn_bwd_dkv");                                              \
    if (rc != CFHIP_OK) return rc;                                                                                  \
    hipLaunchKernelGGL(attn_bwd_dkv_pers_kernel<NBQ_>, grid, block, lds, s, p);                                     \
    break;
    switch (nbq) {
      CFHIP_DKV_PERS(5) CFHIP_DKV_PERS(6) CFHIP_DKV_PERS(7) CFHIP_DKV_PERS(8)
      default: cfhip_set_error("attn_bwd: bad nbq %d", nbq); return CFHIP_ERR_INVALID;
    }
#undef CFHIP_DKV_PERS
    CFHIP_CHECK_LAUNCH("attn_bwd_dkv(persistent)");
  } else if (parts & 2) {
    const int nbq = (Tq + 31) / 32;
    const int nw = pick_waves(Tk);
    const int tiles = (Tk + 15) / 16;
    (void)tiles;
    dim3 grid(1, H, B), block(nw * 64);
    const size_t lds = (size_t)2 * nbq * 32 * 128 + (size_t)2 * nbq * 32 * sizeof(float);
    rc = plain ? set_lds(attn_bwd_dkv_kernel<true>, lds, "attn_bwd_dkv") : set_lds(attn_bwd_dkv_kernel<false>, lds, "attn_bwd_dkv");
    if (rc != CFHIP_OK) return rc;
    if (plain) hipLaunchKernelGGL(attn_bwd_dkv_kernel<true>, grid, block, lds, s, p, nbq);
    else hipLaunchKernelGGL(attn_bwd_dkv_kernel<false>, grid, block, lds, s, p, nbq);
    CFHIP_CHECK_LAUNCH("attn_bwd_dkv");
  }
  return CFHIP_OK;
}

extern "C" int cfhip_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                              const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk,
                              void* dv, int B, int H, int Tq, int Tk, int64_t q_stride_b,
                              int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                              int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                              int64_t ms_q, float scale, int causal, int parts, void* stream) {
  return attn_bwd_impl(q, k, v, o, d_o, lse, delta, mask, dq, dk, dv, B, H, Tq, Tk, CFHIP_ATTN_HEAD_DIM, q_stride_b,
                       q_stride_t, kv_stride_b, kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal,
                       parts, stream);
}

extern "C" int cfhip_attn_bwd_dh(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                 const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk,
                                 void* dv, int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b,
                                 int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                                 int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                                 int64_t ms_q, float scale, int causal, int parts, void* stream) {
  return attn_bwd_impl(q, k, v, o, d_o, lse, delta, mask, dq, dk, dv, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t,
                       kv_stride_b, kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal, parts, stream);
}

extern "C" int cfhip_attn_fwd_dropout(const void* q, const void* k, const void* v, void* o, float* lse,
                                      const uint8_t* mask, int B, int H, int Tq, int Tk, int head_dim,
                                      int64_t q_stride_b, int64_t q_stride_t, int64_t kv_stride_b,
                                      int64_t kv_stride_t, int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b,
                                      int64_t ms_h, int64_t ms_q, float scale, int causal, float dropout_p,
                                      uint64_t seed, uint64_t offset, void* stream) {
  CFHIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attn_fwd_dropout: p = %f is not in [0, 1)", (double)dropout_p);
  return attn_fwd_impl(q, k, v, o, lse, mask, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t, kv_stride_b,
                       kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal, stream, dropout_p, seed, offset);
}

extern "C" int cfhip_attn_bwd_dropout(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                      const float* lse, float* delta, const uint8_t* mask, void* dq, void* dk,
                                      void* dv, int B, int H, int Tq, int Tk, int head_dim, int64_t q_stride_b,
                                      int64_t q_stride_t, int64_t kv_stride_b, int64_t kv_stride_t,
                                      int64_t o_stride_b, int64_t o_stride_t, int64_t ms_b, int64_t ms_h,
                                      int64_t ms_q, float scale, int causal, int parts, float dropout_p,
                                      uint64_t seed, uint64_t offset, void* stream) {
  CFHIP_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "attn_bwd_dropout: p = %f is not in [0, 1)", (double)dropout_p);
  return attn_bwd_impl(q, k, v, o, d_o, lse, delta, mask, dq, dk, dv, B, H, Tq, Tk, head_dim, q_stride_b, q_stride_t,
                       kv_stride_b, kv_stride_t, o_stride_b, o_stride_t, ms_b, ms_h, ms_q, scale, causal, parts, stream,
                       dropout_p, seed, offset);
}

/* the keep mask the DROP kernels use, one byte per (b, h, i, j): for tests and for callers that need the mask itself */
namespace {
__global__ void attn_dropout_mask_kernel(AttnParams p, unsigned char* out) {
  const long total = (long)p.B * p.H * p.Tq * p.Tk;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int j = (int)(idx % p.Tk);
    long t = idx / p.Tk;
    const int i = (int)(t % p.Tq);
    t /= p.Tq;
    const int h = (int)(t % p.H), b = (int)(t / p.H);
    const unsigned w = pick_word(drop_block(p, b, h, i >> 2, j >> 2), i);
    out[idx] = ((w >> (8 * (j & 3))) & 255u) >= p.drop_thresh ? 1 : 0;
  }
}
}  // namespace

extern "C" int cfhip_attn_dropout_mask(void* mask_out, int B, int H, int Tq, int Tk, float dropout_p, uint64_t seed,
                                       uint64_t offset, void* stream) {
  CFHIP_REQUIRE(mask_out && B > 0 && H > 0 && Tq > 0 && Tk > 0, "attn_dropout_mask: bad arguments");
  CFHIP_REQUIRE(dropout_p > 0.f && dropout_p < 1.f, "attn_dropout_mask: p = %f is not in (0, 1)", (double)dropout_p);
  AttnParams p = {};
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  set_dropout(p, dropout_p, seed, offset);
  const long total = (long)B * H * Tq * Tk;
  hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                     (unsigned char*)mask_out);
  CFHIP_CHECK_LAUNCH("attn_dropout_mask");
  return CFHIP_OK;
}
