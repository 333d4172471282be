// K1/K2, round 4: the forward (nt) and input-gradient (nn) GEMMs on v_mfma_f32_32x32x16_bf16 with a software-pipelined wave.
//
//   C[m][n] = epilogue( sum_k A(m,k) * B(n,k) ),  A k-major [M][K]; B k-major [N][K] (forward y = x W^T) or m-major [K][N] (dX = dY W).
//
// Replaces F.linear (reference modules/core/customs.py:89, attentions.py:214) and the dX half of its autograd backward, the same
// call sites as csrc/gemm.hip; the weight gradients stay on gemm_grouped.hip.
//
// What is different from gemm.hip (whose K-step is {barrier | 10 DMA | 28 fragment reads | 48 MFMAs}, strictly in that order
// inside a wave, so that a wave's matrix pipe idles while it stages and reads and only the CU's other workgroup covers it:
// 51-54 % matrix-pipe utilisation in the K loop, profiles/r03/pmc_gemm_c14_vs_c15.txt):
//   * 32x32x16 MFMA (32 cycles of matrix pipe per instruction: 7 issue slots in its shadow instead of 3, and the higher of the
//     two bf16 peaks), wave tiles of 128x128 / 128x64 / 96x64: half the fragment bytes per FLOP of a 16x16 tiling of 96x32;
//   * fragments are double-buffered in REGISTERS per 16-deep sub-step: the reads of sub-step s + 1 are in flight while the
//     MFMAs of sub-step s issue; the fragments of K-step t + 1's first sub-step are read behind the barrier in the middle of
//     K-step t, so the matrix pipe sees no LDS latency at a K-step boundary either;
//   * ONE barrier per 64-deep K-step, placed in front of the LAST sub-step's MFMAs: in front of it every wave has retired its
//     own LDS-DMA of K-step t + 1 (counted vmcnt) and all its reads of the current ring slot (lgkmcnt(0), free by then: the
//     reads were issued a whole sub-step earlier); behind it the slot just read is refilled with K-step t + NSLOT, interleaved
//     with the MFMAs of the last sub-step;
//   * PERSIST: one workgroup per CU walks its tiles (XCD-aware order: an XCD's resident workgroups share operand panels in its
//     L2); the K-steps of ALL its tiles form one stream through the ring, so the first K-steps of tile i + 1 are in flight
//     while tile i is in its epilogue (the epilogue transposes through its own LDS strip, not through the ring), and there is
//     no barrier between an epilogue and the next K loop: waves drift apart by an epilogue's jitter and meet again at the
//     barrier of the next tile's first K-step.
// gfx9 VMEM completes in order (loads and stores share vmcnt and retire in issue order), so a counted wait across an epilogue's
// stores is exact when the stores issued since are counted in: that count is a compile-time constant (every store is issued
// unconditionally, out-of-range rows / columns are dropped by the buffer descriptor).
#include "gemm_device.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// LW > 0: LW extra "loader" waves issue every LDS-DMA of the workgroup; the WM x WN compute waves never touch VMEM inside the
// K loop.  (A DMA instruction costs its wave 60-180 cycles of issue time — the CU's texture addresser takes 64 B / clk, 16 cycles
// per 1 KiB instruction, shared by all waves — and an in-order wave cannot issue its next MFMA meanwhile: with one compute wave
// per SIMD the matrix pipe idled for most of a K-step's 16 DMA issues, profiles/r04/gemm_pp_first.log.)
template <int BM_, int BN_, int WM_, int WN_, int NSLOT_, bool PERSIST_, int LW_ = 0>
struct PCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSLOT = NSLOT_, BK = 64, LW = LW_;
  static constexpr bool PERSIST = PERSIST_;
  static constexpr int NW = WM * WN, NS = LW > 0 ? LW : NW, NT = (NW + LW) * 64;  // compute waves, staging waves, threads
  static constexpr int TM = BM / WM, TN = BN / WN;  // wave tile
  static constexpr int FM = TM / 32, FN = TN / 32;  // 32x32 blocks per wave
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int RING_BYTES = STAGE_BYTES * NSLOT;
  static constexpr int STRIP_WAVE = 32 * 64 * 4;  // [32 rows][64 columns] f32 per wave
  static constexpr int STRIP_BYTES = NW * STRIP_WAVE;
  // persistent: the ring carries the next tile's K-steps during an epilogue, the strips are their own LDS; otherwise they alias it
  static constexpr int LDS_BYTES = PERSIST ? RING_BYTES + STRIP_BYTES : (RING_BYTES > STRIP_BYTES ? RING_BYTES : STRIP_BYTES);
  static constexpr int A_INSTR = A_BYTES / 1024 / NS, B_INSTR = B_BYTES / 1024 / NS, LPS = A_INSTR + B_INSTR;
  static constexpr int WGS_PER_CU = (160 * 1024 / LDS_BYTES) > (8 / (NW + LW)) ? (8 / (NW + LW) < 1 ? 1 : 8 / (NW + LW)) : (160 * 1024 / LDS_BYTES);
  static constexpr int WAVES_PER_SIMD = WGS_PER_CU * (NW + LW) / 4 < 1 ? 1 : WGS_PER_CU * (NW + LW) / 4;
  static_assert(TM % 32 == 0 && TN % 64 == 0, "wave tiles: whole 32x32 MFMA blocks, 64-column epilogue chunks");
  static_assert(A_BYTES % (1024 * NS) == 0 && B_BYTES % (1024 * NS) == 0, "tiles must split evenly over the staging waves");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
  static_assert(NSLOT == 2 || NSLOT == 3, "ring of two or three K-steps");
  static_assert((NSLOT - 2) * LPS + 48 < 64 || true, "vmcnt is a 6-bit counter");
};

// 32 rows x 16 k of a k-major tile (row pitch 128 B, 16-byte slots XOR-swizzled by kswz<64>): the operand layout of
// v_mfma_f32_32x32x16_bf16 — lane l holds row l & 31, k = 8 (l >> 5) .. + 7.  `base` = this lane's byte offset for sub-step 0 of
// the rows r0 = 0 block: row * 128 + ((l >> 5) ^ kswz(row)) << 4; sub-step s flips bits 5..6 (the slot index is 2 s + (l >> 5)).
__device__ __forceinline__ bf16x8 frag32_k(const char* tile, int base, int r0, int s) {
  return *reinterpret_cast<const bf16x8*>(tile + ((base ^ (s << 5)) + r0 * 128));
}

// 32 columns x 16 k of an m-major tile ([64 k][R columns], 32-byte chunks XOR-swizzled by mkey32<R>): two hardware-transpose
// reads per fragment.  16-lane group G = l >> 4: columns c0 + 16 (G & 1) .., k = 16 s + 8 (G >> 1) + {0..3 | 4..7}.
template <int R>
__device__ __forceinline__ bf16x8 frag32_m(const char* tile, int c0, int s, int lane) {
  const int G = lane >> 4;
  const int j = (lane & 15) >> 2;
  const int q = lane & 3;
  const int krow = s * 16 + (G >> 1) * 8 + j;  // mkey32(krow) == mkey32(krow + 4)
  const int cb = (c0 >> 4) + (G & 1);
  const char* p = tile + krow * (R * 2) + ((cb ^ mkey32<R>(krow)) << 5) + q * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 4 * (R * 2)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

struct TileRef {
  int m0, n0;
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
};

// tile -> (tile_m, tile_n): the walk orders of gemm.hip (column groups / row groups / row-major)
__device__ __forceinline__ void tile_coords(const GemmParams& p, int tile, int& tile_m, int& tile_n) {
  if (p.group_n > 0) {
    const int per_group = p.group_n * p.tiles_m;
    const int grp = tile / per_group;
    const int first_n = grp * p.group_n;
    const int gw = min(p.group_n, p.tiles_n - first_n);
    const int r = tile - grp * per_group;
    tile_m = r / gw;
    tile_n = first_n + (r - tile_m * gw);
  } else if (p.group_n < 0) {
    const int gm = -p.group_n;
    const int per_group = gm * p.tiles_n;
    const int grp = tile / per_group;
    const int first_m = grp * gm;
    const int gh = min(gm, p.tiles_m - first_m);
    const int r = tile - grp * per_group;
    tile_n = r / gh;
    tile_m = first_m + (r - tile_n * gh);
  } else {
    tile_m = tile / p.tiles_n;
    tile_n = tile - tile_m * p.tiles_n;
  }
}

// virtual workgroup id v (v & 7 = the XCD it runs on) -> tile: XCD x owns a contiguous range of the walk; invalid ids get an
// empty descriptor (every DMA lane out of range: zero fill, no traffic — the vmcnt bookkeeping stays the same)
template <bool BT, class C>
__device__ __forceinline__ TileRef make_tile(const GemmParams& p, int v, int total) {
  TileRef t;
  if (v >= total) {
    t.m0 = t.n0 = 0;
    t.a_rsrc = make_rsrc(p.A, 0);
    t.b_rsrc = make_rsrc(p.B, 0);
    return t;
  }
  const int q8 = total >> 3, r8 = total & 7;
  const int xcd = v & 7, loc = v >> 3;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  int tile_m, tile_n;
  tile_coords(p, item, tile_m, tile_n);
  t.m0 = tile_m * C::BM;
  t.n0 = tile_n * C::BN;
  const int rows_a = p.M - t.m0, rows_b = p.N - t.n0;
  // exact descriptors: rows of a k-major operand beyond the matrix are beyond num_records (zero fill); columns of the m-major
  // B beyond N read the neighbouring elements of the same matrix — they only reach output columns >= N, which are never stored
  t.a_rsrc = make_rsrc(p.A + (long)t.m0 * p.lda, ((long)(rows_a - 1) * p.lda + p.K) * 2);
  t.b_rsrc = BT ? make_rsrc(p.B + t.n0, ((long)(p.K - 1) * p.ldb + rows_b) * 2)
                : make_rsrc(p.B + (long)t.n0 * p.ldb, ((long)(rows_b - 1) * p.ldb + p.K) * 2);
  return t;
}

// ---- epilogue ------------------------------------------------------------------------------------------------------------
// With swapped operands (D = Bfrag x Afrag) a lane of a 32x32 block holds ONE output row m = l & 31 and the columns
// n = 8 (reg >> 2) + 4 (l >> 5) + (reg & 3): four runs of 4.  A wave transposes 32 rows x 64 columns at a time through its private
// strip ([32][64] f32, 16-byte chunks XOR-ed with row & 15: conflict-free both ways), after which a lane owns 8 (bf16 output) or 4
// (f32 output) CONSECUTIVE columns of a row: residual / pre-activation traffic in 16-byte coalesced loads, every store
// instruction writes whole 128-byte (bf16) / 256-byte (f32) row segments.  All global accesses go through descriptors anchored
// at the tile origin (rows >= M and columns >= N are out-of-range offsets): straight-line code, and a compile-time number of
// VMEM instructions per tile (the K loop's counted waits rely on it).
template <int EPI, bool F32>
struct EpiCount {
  static constexpr int LOADS = (EPI == CFHIP_EPI_RESIDUAL || EPI == CFHIP_EPI_DGELU) ? (F32 ? 8 : 4) : 0;
  static constexpr int STORES = F32 ? 8 : (EPI == CFHIP_EPI_GELU ? 8 : 4);
};

template <int EPI, class C, bool F32, bool QUICK>
__device__ __forceinline__ void pp_epilogue(const GemmParams& p, f32x16 (&acc)[C::FM][C::FN],
                                            const f32x4 (&bias_r)[(C::TN / 64) * (F32 ? 1 : 2)], char* strip, int m0, int n0,
                                            int wm, int wn, int lane) {
  static_assert(!F32 || EPI == CFHIP_EPI_NONE || EPI == CFHIP_EPI_RESIDUAL, "f32 output: bias / residual only");
  constexpr bool HAS_AUX = EPI == CFHIP_EPI_RESIDUAL || EPI == CFHIP_EPI_DGELU;
  constexpr int NCH = C::FM * (C::TN / 64);  // chunks of 32 rows x 64 columns
  constexpr int NP = F32 ? 8 : 4;            // row passes per chunk: 4 / 8 rows per instruction
  const int mrow = lane & 31, h = lane >> 5;
  // read side: bf16 output -> 8 lanes x 8 columns per row, 8 rows per pass; f32 output -> 16 lanes x 4 columns, 4 rows per pass
  const int rr = F32 ? (lane >> 4) : (lane >> 3);
  const int cl = F32 ? (lane & 15) * 4 : (lane & 7) * 8;  // column inside the chunk
  constexpr int RPP = F32 ? 4 : 8;
  const int ES = F32 ? 4 : 2;
  const __amdgpu_buffer_rsrc_t c_rsrc = tile_rsrc(p.C, p.ldc, ES, m0, n0, p.M, p.N);
  __amdgpu_buffer_rsrc_t x_rsrc = c_rsrc, o_rsrc = c_rsrc;
  if constexpr (HAS_AUX) x_rsrc = tile_rsrc(p.aux_in, p.ldc, ES, m0, n0, p.M, p.N);
  if constexpr (EPI == CFHIP_EPI_GELU) o_rsrc = tile_rsrc(p.aux_out, p.ldc, 2, m0, n0, p.M, p.N);
  const bool has_pre = EPI == CFHIP_EPI_GELU && p.aux_out != nullptr;

  // element offset (from the tile origin) of pass `ps` of chunk `ch`, or OOB when this lane's columns lie beyond N (N % 8 == 0)
  auto col_of = [&](int ch) -> int { return wn * C::TN + (ch % (C::TN / 64)) * 64 + cl; };
  auto eoff = [&](int ch, int ps) -> unsigned {
    const int row = wm * C::TM + (ch / (C::TN / 64)) * 32 + ps * RPP + rr;
    return (unsigned)(row * (int)p.ldc + col_of(ch));
  };
  // operands the epilogue reads are fetched one chunk ahead (two register sets) — except for the f32 residual stream on the
  // 128x128 wave tile, where the second set (32 registers) pushed the kernel into scratch: one set, fetched per chunk
  constexpr int AD = (F32 && C::FM * C::FN >= 16) ? 1 : 2;
  u32x4 aux[AD][NP];
  auto load_aux = [&](int ch, u32x4 (&dst)[NP]) {
    const bool ok = n0 + col_of(ch) < p.N;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) dst[ps] = bload16(x_rsrc, ok ? eoff(ch, ps) * (unsigned)ES : OOB);
  };
  if constexpr (HAS_AUX) load_aux(0, aux[0]);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int mb = ch / (C::TN / 64), cc = ch % (C::TN / 64);
    if constexpr (HAS_AUX) {
      if (AD == 2 && ch + 1 < NCH) load_aux(ch + 1, aux[(ch + 1) % AD]);
      if (AD == 1 && ch > 0) load_aux(ch, aux[0]);
    }
    // accumulators -> strip
#pragma unroll
    for (int nbl = 0; nbl < 2; ++nbl)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x16& a = acc[mb][cc * 2 + nbl];
        const f32x4 v = {a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3]};
        const int chunk = nbl * 8 + rg * 2 + h;
        *reinterpret_cast<f32x4*>(strip + mrow * 256 + ((chunk ^ (mrow & 15)) << 4)) = v;
      }
    const int col = col_of(ch);
    const bool c_ok = n0 + col < p.N;
    const f32x4 b_lo = bias_r[F32 ? cc : cc * 2], b_hi = bias_r[F32 ? cc : cc * 2 + 1];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = ps * RPP + rr;
      const unsigned e = c_ok ? eoff(ch, ps) : 0u;
      if constexpr (F32) {
        f32x4 v = *reinterpret_cast<const f32x4*>(strip + r * 256 + ((((cl >> 2)) ^ (r & 15)) << 4));
        v += b_lo;
        if constexpr (HAS_AUX) v += __builtin_bit_cast(f32x4, aux[ch % AD][ps]);
        bstore16(c_rsrc, c_ok ? e * 4u : OOB, __builtin_bit_cast(u32x4, v));
      } else {
        f32x4 lo = *reinterpret_cast<const f32x4*>(strip + r * 256 + ((((cl >> 2)) ^ (r & 15)) << 4));
        f32x4 hi = *reinterpret_cast<const f32x4*>(strip + r * 256 + ((((cl >> 2) + 1) ^ (r & 15)) << 4));
        lo += b_lo;
        hi += b_hi;
        if constexpr (EPI == CFHIP_EPI_GELU) {
          // GELU of the bf16-rounded pre-activation (what the saved tensor holds for backward)
          const u32x4 w = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
          bstore16(o_rsrc, (has_pre && c_ok) ? e * 2u : OOB, w);
          if constexpr (QUICK) {
            lo = f32x4{quick_gelu_f(bf16lo(w[0])), quick_gelu_f(bf16hi(w[0])), quick_gelu_f(bf16lo(w[1])), quick_gelu_f(bf16hi(w[1]))};
            hi = f32x4{quick_gelu_f(bf16lo(w[2])), quick_gelu_f(bf16hi(w[2])), quick_gelu_f(bf16lo(w[3])), quick_gelu_f(bf16hi(w[3]))};
          } else {
            lo = f32x4{gelu_erf_f(bf16lo(w[0])), gelu_erf_f(bf16hi(w[0])), gelu_erf_f(bf16lo(w[1])), gelu_erf_f(bf16hi(w[1]))};
            hi = f32x4{gelu_erf_f(bf16lo(w[2])), gelu_erf_f(bf16hi(w[2])), gelu_erf_f(bf16lo(w[3])), gelu_erf_f(bf16hi(w[3]))};
          }
        } else if constexpr (EPI == CFHIP_EPI_RESIDUAL) {
          const u32x4 w = aux[ch % AD][ps];
          lo += f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
          hi += f32x4{bf16lo(w[2]), bf16hi(w[2]), bf16lo(w[3]), bf16hi(w[3])};
        } else if constexpr (EPI == CFHIP_EPI_DGELU) {
          const u32x4 w = aux[ch % AD][ps];
          if constexpr (QUICK) {
            lo *= f32x4{quick_gelu_grad_f(bf16lo(w[0])), quick_gelu_grad_f(bf16hi(w[0])), quick_gelu_grad_f(bf16lo(w[1])), quick_gelu_grad_f(bf16hi(w[1]))};
            hi *= f32x4{quick_gelu_grad_f(bf16lo(w[2])), quick_gelu_grad_f(bf16hi(w[2])), quick_gelu_grad_f(bf16lo(w[3])), quick_gelu_grad_f(bf16hi(w[3]))};
          } else {
            lo *= f32x4{gelu_erf_grad_f(bf16lo(w[0])), gelu_erf_grad_f(bf16hi(w[0])), gelu_erf_grad_f(bf16lo(w[1])), gelu_erf_grad_f(bf16hi(w[1]))};
            hi *= f32x4{gelu_erf_grad_f(bf16lo(w[2])), gelu_erf_grad_f(bf16hi(w[2])), gelu_erf_grad_f(bf16lo(w[3])), gelu_erf_grad_f(bf16hi(w[3]))};
          }
        }
        const u32x4 w = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
        bstore16(c_rsrc, c_ok ? e * 2u : OOB, w);
      }
    }
  }
}

// VMEM instructions one wave issues in an epilogue (loads + stores): what a counted vmcnt across a tile boundary adds
template <int EPI, class C, bool F32>
constexpr int epi_vmem_ops() {
  return C::FM * (C::TN / 64) * (EpiCount<EPI, F32>::LOADS + EpiCount<EPI, F32>::STORES);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  CFHIP_WAIT_VMCNT(N > 63 ? 63 : N);
}

// DMA instruction `j` (A: 0 .. A_INSTR - 1, then B) of K-step k0 into ring slot `st` (j is a constant after unrolling).  K is a
// multiple of 64 on this path (host-checked): every K-step is whole, the K advance rides on the instruction's scalar offset.
template <bool BT, class C>
__device__ __forceinline__ void stage_piece(const TileRef& t, const GemmParams& p, char* st, int wave, const StagePlan<C::A_INSTR>& pa,
                                            const StagePlan<C::B_INSTR>& pb, int k0, const int j) {
  if (j < C::A_INSTR) {
    lds_dma16_s(t.a_rsrc, st + (wave * C::A_INSTR + j) * 1024, pa.voff[j], (unsigned)(k0 * 2));
  } else {
    const int jb = j - C::A_INSTR;
    lds_dma16_s(t.b_rsrc, st + C::A_BYTES + (wave * C::B_INSTR + jb) * 1024, pb.voff[jb],
                BT ? (unsigned)((long)k0 * p.ldb * 2) : (unsigned)(k0 * 2));
  }
}

template <bool BT, class C>
__device__ __forceinline__ void stage_all(const TileRef& t, const GemmParams& p, char* st, int wave, const StagePlan<C::A_INSTR>& pa,
                                          const StagePlan<C::B_INSTR>& pb, int k0) {
#pragma unroll
  for (int j = 0; j < C::LPS; ++j) stage_piece<BT, C>(t, p, st, wave, pa, pb, k0, j);
}

#ifndef CFHIP_PP_SCHED
#define CFHIP_PP_SCHED 1
#endif

template <bool BT, int EPI, class C, bool F32, bool QUICK>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD)
void gemm_pp_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ring = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  char* const strip = smem + (C::PERSIST ? C::RING_BYTES : 0) + wave * C::STRIP_WAVE;

  const int total = p.tiles_m * p.tiles_n;
  const int G = gridDim.x;
  const int nk = p.K >> 6;

  // tile-independent staging plans (see make_tile: the descriptors do the row range checks)
  const int swave = C::LW > 0 ? (wave >= C::NW ? wave - C::NW : 0) : wave;  // index among the waves that stage
  const StagePlan<C::A_INSTR> pa = make_plan<false, C::BM, C::A_INSTR, 64>(swave, lane, p.lda, 1 << 30);
  const StagePlan<C::B_INSTR> pb = make_plan<BT, C::BN, C::B_INSTR, 64, true>(swave, lane, p.ldb, 1 << 30);

  if constexpr (C::LW > 0) {
    if (wave >= C::NW) {
      // ---- loader wave: the same ring protocol as below, minus everything else.  In front of barrier(t) its own pieces of
      // K-step t + 1 have landed; behind it (every compute wave has retired its reads of slot t % NSLOT) K-step t + NSLOT goes
      // into that slot.  Only DMAs on this wave's vmcnt: every count is exact.
      int v = blockIdx.x;
      TileRef cur = make_tile<BT, C>(p, v, total);
      TileRef nxt = C::PERSIST ? make_tile<BT, C>(p, v + G, total) : cur;
#pragma unroll
      for (int st = 0; st < C::NSLOT; ++st) stage_all<BT, C>(cur, p, ring + st * C::STAGE_BYTES, swave, pa, pb, st * 64);
      wait_vm<(C::NSLOT - 1) * C::LPS>();
      __builtin_amdgcn_s_barrier();
      int rd = 0;
      while (true) {
        for (int t = 0; t < nk; ++t) {
          if (C::PERSIST || t + C::NSLOT - 1 < nk) wait_vm<(C::NSLOT - 2) * C::LPS>();
          else wait_vm<0>();
          __builtin_amdgcn_s_barrier();
          const int ts = t + C::NSLOT;
          char* const st = ring + rd * C::STAGE_BYTES;
          if (ts < nk) stage_all<BT, C>(cur, p, st, swave, pa, pb, ts * 64);
          else if (C::PERSIST) stage_all<BT, C>(nxt, p, st, swave, pa, pb, (ts - nk) * 64);
          rd = rd + 1 == C::NSLOT ? 0 : rd + 1;
        }
        if (!C::PERSIST) return;
        v += G;
        if (v >= total) return;
        cur = nxt;
        nxt = make_tile<BT, C>(p, v + G, total);
      }
    }
  }

  // fragment addressing: k-major operands by one per-lane base (see frag32_k)
  const int frow = lane & 31;
  const int kbase = frow * 128 + (((lane >> 5) ^ kswz<64>(frow)) << 4);  // rows r0 + 32 b: kswz(row + 32 b) == kswz(row)
  const int a_off = kbase + wm * C::TM * 128;
  const int b_off = kbase + wn * C::TN * 128;

  f32x16 acc[C::FM][C::FN];
  bf16x8 fa0[C::FM], fb0[C::FN], fa1[C::FM], fb1[C::FN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::FN; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  };
  auto load_frags = [&](bf16x8 (&fa)[C::FM], bf16x8 (&fb)[C::FN], const char* st, const int s) {
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi) fa[mi] = frag32_k(st, a_off, mi * 32, s);
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni)
      fb[ni] = BT ? frag32_m<C::BN>(st + C::A_BYTES, wn * C::TN + ni * 32, s, lane) : frag32_k(st + C::A_BYTES, b_off, ni * 32, s);
  };
  auto mma = [&](const bf16x8 (&fa)[C::FM], const bf16x8 (&fb)[C::FN]) {
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::FN; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
  };
  // one sub-step: the fragment reads of the NEXT sub-step ride in the shadow of this one's first MFMAs (one read per MFMA: the
  // last read is then NMMA - NRD MFMAs = hundreds of cycles old when the next sub-step's first MFMA wants it)
  constexpr int NMMA = C::FM * C::FN;
  constexpr int NRD = C::FM + (BT ? 2 : 1) * C::FN;  // LDS read instructions per sub-step
  auto interleave = [&]() {
#if CFHIP_PP_SCHED
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      if (i < NMMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (NMMA > NRD) __builtin_amdgcn_sched_group_barrier(0x008, NMMA - NRD, 0);
#endif
  };

  int v = blockIdx.x;
  TileRef cur = make_tile<BT, C>(p, v, total);
  TileRef nxt = C::PERSIST ? make_tile<BT, C>(p, v + G, total) : cur;
  if (v >= total) return;  // (never: the grid is at most `total` workgroups)

  // prologue: K-steps 0 .. NSLOT - 1 of the first tile (the host guarantees nk >= NSLOT)
  if constexpr (C::LW == 0) {
#pragma unroll
    for (int st = 0; st < C::NSLOT; ++st)
      stage_all<BT, C>(cur, p, ring + st * C::STAGE_BYTES, wave, pa, pb, st * 64);
    wait_vm<(C::NSLOT - 1) * C::LPS>();
  }
  __builtin_amdgcn_s_barrier();

  // the bias of this lane's output columns (the epilogue's read side: 8 / 4 consecutive columns per 64-column chunk)
  constexpr int NBR = (C::TN / 64) * (F32 ? 1 : 2);
  f32x4 bias_r[NBR];
  const int bcol = wn * C::TN + (F32 ? (lane & 15) * 4 : (lane & 7) * 8);

  int rd = 0;              // ring slot of the K-step being computed
  bool after_epi = false;  // the epilogue's VMEM instructions are younger than the DMA of this tile's K-step 1
  load_frags(fa0, fb0, ring, 0);
  while (true) {
    zero_acc();
    for (int t = 0; t < nk; ++t) {
      char* const st = ring + rd * C::STAGE_BYTES;
      load_frags(fa1, fb1, st, 1);
      mma(fa0, fb0);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_frags(fa0, fb0, st, 2);
      mma(fa1, fb1);
      interleave();
      __builtin_amdgcn_sched_barrier(0);
      load_frags(fa1, fb1, st, 3);
      mma(fa0, fb0);
      interleave();
      // ---- the barrier of K-step t: own DMA of K-step t + 1 retired (K-steps t + 2 .. may stay in flight, and so may
      // the stores of an epilogue issued since), every read of slot `rd` retired
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (C::LW == 0) {
        if (C::PERSIST && after_epi && t == 0) {
          wait_vm<(C::NSLOT - 2) * C::LPS + epi_vmem_ops<EPI, C, F32>()>();
        } else if (C::PERSIST || t + C::NSLOT - 1 < nk) {
          wait_vm<(C::NSLOT - 2) * C::LPS>();
        } else {
          wait_vm<0>();  // not persistent: nothing was issued beyond the last K-step
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- behind it, in the shadow of the last sub-step's MFMAs: the first fragments of K-step t + 1 (of the next tile's
      // K-step 0 on a tile's last step: they survive the epilogue in registers), then the refill of slot `rd` with K-step
      // t + NSLOT (of the next tile when this one has no such step)
      const int ts = t + C::NSLOT;
      const int nrd = rd + 1 == C::NSLOT ? 0 : rd + 1;
      const bool in_cur = ts < nk;
      const bool issue = C::LW == 0 && (in_cur || C::PERSIST);
      const TileRef& src = (C::PERSIST && !in_cur) ? nxt : cur;
      const int k0 = (in_cur ? ts : ts - nk) * 64;
      const char* const nst = ring + nrd * C::STAGE_BYTES;
      const bool last = t == nk - 1;
      // The bias loads are the one compiler-visible VMEM load of the kernel, and hipcc waits for a load it can see with a
      // count that ignores the (inline-asm) DMAs: anywhere else that wait drains every DMA and every store in flight.  Here
      // nothing is in flight yet (NSLOT == 2: the wait above was vmcnt(0)); they are consumed ("touched") a few MFMAs later,
      // before this step's DMAs are issued, and cost the epilogue no wait at all.
      if (last && p.bias != nullptr) {
#pragma unroll
        for (int i = 0; i < NBR; ++i) {
          const int col = bcol + (F32 ? i * 64 : (i >> 1) * 64 + (i & 1) * 4);
          bias_r[i] = cur.n0 + col < p.N ? *reinterpret_cast<const f32x4*>(p.bias + cur.n0 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      constexpr int D0 = NMMA / 4;                                  // MFMAs in front of the first DMA
      constexpr int DPM = (C::LPS + (NMMA - D0) - 1) / (NMMA - D0);  // DMA instructions per MFMA behind it
#pragma unroll
      for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FN; ++ni) {
          const int i = mi * C::FN + ni;
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb1[ni], fa1[mi], acc[mi][ni], 0, 0, 0);
          if (i < C::FM) fa0[i] = frag32_k(nst, a_off, i * 32, 0);
          else if (i < C::FM + C::FN)
            fb0[i - C::FM] = BT ? frag32_m<C::BN>(nst + C::A_BYTES, wn * C::TN + (i - C::FM) * 32, 0, lane)
                                : frag32_k(nst + C::A_BYTES, b_off, (i - C::FM) * 32, 0);
          if (i == D0 - 1 && last && p.bias != nullptr) {
#pragma unroll
            for (int b = 0; b < NBR; ++b) asm volatile("" : "+v"(bias_r[b]));
          }
          if (i >= D0) {
#pragma unroll
            for (int d = 0; d < DPM; ++d) {
              const int j = (i - D0) * DPM + d;
              if (j < C::LPS && issue) stage_piece<BT, C>(src, p, st, wave, pa, pb, k0, j);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      rd = nrd;
    }
    // ---- epilogue of `cur` (the ring already carries the next tile's first K-steps; K-step 0 of it landed before the last
    // barrier above)
    if (p.bias == nullptr) {
#pragma unroll
      for (int b = 0; b < NBR; ++b) bias_r[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    pp_epilogue<EPI, C, F32, QUICK>(p, acc, bias_r, strip, cur.m0, cur.n0, wm, wn, lane);
    if (!C::PERSIST) break;
    v += G;
    if (v >= total) break;
    cur = nxt;
    nxt = make_tile<BT, C>(p, v + G, total);
    after_epi = true;
  }
}

// configurations: index = gemm_config - PP_BASE (17).  Measured on one MI355X (profiles/r04/gemm_pp_*.log): in long K loops
// every one of them lands where the 16x16x32 kernels of gemm.hip land (1.15-1.36 PFLOP/s on random operands: the 256x256x64
// plain kernel, gemm_config 13, is the fastest at 8192^3, PP5 at 4096^3), on the ViT step's shapes the 192x128x64 tiles of
// gemm_config 14 / 15 quantise better, and whole-step A/Bs are within 1 % either way — so pick_config() does not select them;
// they stay selectable for the A/B tools and are covered by tests/test_gpu_gemm.py::test_pipelined_gemm_configs.
using PP0 = PCfg<256, 256, 2, 2, 2, true>;      // 17: persistent, 128 KiB ring + 32 KiB strips, 4 waves of 128x128 (256 AGPRs), 1 WG / CU
using PP5 = PCfg<256, 256, 2, 4, 2, false>;     // 18: one tile per workgroup, EIGHT waves of 128x64 (two per SIMD cover each other's DMA issues)
using PP7 = PCfg<256, 128, 2, 2, 3, false, 4>;  // 19: 4 compute waves of 128x64 + 4 loader waves, 3-slot ring (DMA two K-steps ahead)
using PP2 = PCfg<192, 128, 2, 2, 2, false>;     // 20: the 192x128 tile of gemm_config 15 (2 WG / CU) with the pipelined wave

int g_pp_cus = 0;

template <bool BT, int EPI, class C, bool F32, bool QUICK>
int launch_pp(const GemmParams& p, hipStream_t s) {
  void (*kern)(GemmParams) = gemm_pp_kernel<BT, EPI, C, F32, QUICK>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      cfhip_set_error("gemm_pp: cannot reserve %d bytes of LDS: %s", C::LDS_BYTES, hipGetErrorString(e));
      return CFHIP_ERR_LAUNCH;
    }
    attr_done = true;
  }
  if (g_pp_cus == 0) {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_pp_cus = n;
  }
  const int total = p.tiles_m * p.tiles_n;
  int grid = total;
  if (C::PERSIST) {
    const int slots = g_pp_cus * C::WGS_PER_CU;
    if (grid > slots) grid = slots & ~7;  // a multiple of 8: virtual id v and v + grid run on the same XCD
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS_BYTES, s, p);
  return CFHIP_OK;
}

template <class C>
int launch_pp_cfg(GemmParams p, int b_trans, int epilogue, hipStream_t s) {
  p.tiles_m = (p.M + C::BM - 1) / C::BM;
  p.tiles_n = (p.N + C::BN - 1) / C::BN;
  if (p.group_n >= p.tiles_n || p.group_n < 0) p.group_n = 0;
  const bool f32 = p.out_f32 != 0;
  if (!b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE:
        return f32 ? launch_pp<false, CFHIP_EPI_NONE, C, true, false>(p, s) : launch_pp<false, CFHIP_EPI_NONE, C, false, false>(p, s);
      case CFHIP_EPI_GELU:
        return p.quick ? launch_pp<false, CFHIP_EPI_GELU, C, false, true>(p, s) : launch_pp<false, CFHIP_EPI_GELU, C, false, false>(p, s);
      case CFHIP_EPI_RESIDUAL:
        return f32 ? launch_pp<false, CFHIP_EPI_RESIDUAL, C, true, false>(p, s) : launch_pp<false, CFHIP_EPI_RESIDUAL, C, false, false>(p, s);
      default: break;
    }
  } else {
    switch (epilogue) {
      case CFHIP_EPI_NONE:
        return f32 ? launch_pp<true, CFHIP_EPI_NONE, C, true, false>(p, s) : launch_pp<true, CFHIP_EPI_NONE, C, false, false>(p, s);
      case CFHIP_EPI_DGELU:
        return p.quick ? launch_pp<true, CFHIP_EPI_DGELU, C, false, true>(p, s) : launch_pp<true, CFHIP_EPI_DGELU, C, false, false>(p, s);
      default: break;
    }
  }
  cfhip_set_error("gemm_pp: epilogue %d is not provided for layout (0,%d)", epilogue, b_trans);
  return CFHIP_ERR_INVALID;
}

}  // namespace

// gemm.hip hands over a fully checked problem (its own GemmParams: the same header, the same layout)
int cfhip_internal_gemm_pp_supported(int M, int N, int K, long ldc, int a_trans, int b_trans, int epilogue, int out_dtype, int accumulate,
                                     int split_k, int variant) {
  if (a_trans || accumulate || split_k > 1 || variant < 0 || variant > 3) return 0;
  if ((N & 7) != 0 || (ldc & 7) != 0) return 0;   // 16-byte bf16 row segments per lane
  if (K < 3 * 64 || (K & 63) != 0) return 0;       // whole 64-deep K-steps, at least NSLOT of them
  // exactly the (layout, epilogue, output type) combinations launch_pp_cfg instantiates: anything else is "not supported"
  // (the caller then takes the 16x16x32 kernels), never an error out of a forced configuration
  const bool f32 = out_dtype != 0;
  switch (epilogue) {
    case CFHIP_EPI_NONE: break;
    case CFHIP_EPI_GELU: if (b_trans || f32) return 0; break;
    case CFHIP_EPI_RESIDUAL: if (b_trans) return 0; break;
    case CFHIP_EPI_DGELU: if (!b_trans || f32) return 0; break;
    default: return 0;
  }
  (void)M;
  return 1;
}

int cfhip_internal_gemm_pp(const void* params, int variant, int b_trans, int epilogue, void* stream) {
  const GemmParams& p = *reinterpret_cast<const GemmParams*>(params);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (variant) {
    case 0: return launch_pp_cfg<PP0>(p, b_trans, epilogue, s);
    case 1: return launch_pp_cfg<PP5>(p, b_trans, epilogue, s);
    case 2: return launch_pp_cfg<PP7>(p, b_trans, epilogue, s);
    default: return launch_pp_cfg<PP2>(p, b_trans, epilogue, s);
  }
}
