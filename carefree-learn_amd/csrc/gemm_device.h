// Device-side building blocks shared by the GEMM translation units (gemm.hip: one work item per workgroup;
// gemm_grouped.hip: the grouped 256x256 weight-gradient kernel): parameter block, LDS-DMA staging plans, swizzles,
// MFMA fragment readers and the epilogues.  Everything lives in an anonymous namespace: each TU gets its own copy.
#pragma once
#include "common.h"
#include <string.h>

namespace {

constexpr unsigned OOB = 0x80000000u;  // any offset >= num_records reads as zero

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
  int tiles_m, tiles_n, splits;
  float* bgrad;        // (1,1) layout only: out[m] (+)= sum_k A(m,k)  — the bias gradient of a dW GEMM
  float* bgrad_slabs;  // split-K partials [z][M] of the above
  int bgrad_acc;
  int quick;   // GELU / DGELU epilogues: 0 = exact-erf GELU, 1 = quick GELU x * sigmoid(1.702 x)
  // implicit 3x3 / stride 1 / pad 1 convolution (CONV kernels only): A is the NHWC activation [M = B*H*W pixels][conv_c],
  // the GEMM's k index is (tap = ky*3+kx, channel), a K-step (32 channels of one tap) is gathered straight from A
  int conv_h, conv_w, conv_c, conv_kpt;  // image height / width, channels, K-steps per tap (= conv_c / 32)
  float conv_inv_kpt;
  // CONV == 2 (weight gradient): B is the NHWC activation, the GEMM's n index is (tap, channel), k = pixel;
  // pixel -> (y, x) per K-step by multiply-high with floor(2^32 / d) + 1 (exact while pixel * d < 2^32)
  unsigned conv_magic_w, conv_magic_h;
  int group_n;  // > 0: tiles are walked in column groups of this many tile columns (all rows of a group first); set in launch_cfg
#ifdef CFHIP_ABLATE
  int ablate;  // benchmarking only: bit0 skip in-loop DMA, bit1 skip MFMA/LDS reads, bit2 skip stores
#endif
};

// Tile configuration.  BM x BN x 64 workgroup tile, WM x WN waves (each a (BM/WM) x (BN/WN) sub-tile of
// 16x16 MFMA tiles), NSTAGE-deep LDS ring (prefetch distance NSTAGE-1 K-steps).
template <int BM_, int BN_, int WM_, int WN_, int NSTAGE_, int BK_>
struct Cfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NSTAGE = NSTAGE_, BK = BK_;
  static constexpr int NW = WM * WN, NT = NW * 64;
  static constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
  // B_BYTES: the staged B image, rounded up to whole LDS-DMA instructions per wave.  Only the 80-column-per-wave tiles of the
  // UNet's convolutions (BN = 320 on eight waves, 160 on four: N = 320 / 640 / 960 / 1280 / 1920 / 2560 are all multiples of 160
  // and of none of the power-of-two widths) round: their extra rows are statically out of range (zero-filled, never read).
  static constexpr int A_BYTES = BM * BK * 2, B_RAW = BN * BK * 2;
  static constexpr int B_BYTES = (B_RAW + 1024 * WM_ * WN_ - 1) / (1024 * WM_ * WN_) * (1024 * WM_ * WN_), STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr bool B_PADDED = B_BYTES != B_RAW;
  static constexpr int LDS_BYTES = STAGE_BYTES * NSTAGE;
  static constexpr int A_INSTR = A_BYTES / 1024 / NW, B_INSTR = B_BYTES / 1024 / NW;  // LDS-DMA instr / wave / K-step
  static constexpr int LPS = A_INSTR + B_INSTR;                        // "loads per stage" for vmcnt
  static constexpr int WGS_PER_CU = (160 * 1024 / LDS_BYTES) > (16 / NW) ? (16 / NW) : (160 * 1024 / LDS_BYTES);
  static constexpr int WAVES_PER_SIMD = WGS_PER_CU * NW / 4 < 1 ? 1 : WGS_PER_CU * NW / 4;
  static_assert(BK == 32 || BK == 64, "K-step must be 32 or 64");
  static_assert(A_BYTES % (1024 * NW) == 0 && B_BYTES % (1024 * NW) == 0, "tiles must split evenly over the waves");
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long bytes) {
  if (bytes > 0x7fffffffL) bytes = 0x7fffffffL;
  if (bytes < 0) bytes = 0;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// Per-lane staging plan for one operand tile (R rows or columns): NI LDS-DMA instructions per wave
// per K-step, each moving 1 KiB (64 lanes x 16 B) into a lane-linear LDS image.
template <int NI>
struct StagePlan {
  unsigned voff[NI];  // byte offset from the tile base at k-step 0 (OOB when statically invalid)
  unsigned kpos[NI];  // k index (elements) this lane's 16 bytes start at, inside a K-step
};

// 16-byte slot swizzle of a k-major tile row (row pitch BK*2 bytes): conflict-free ds_read_b128
template <int BK>
__device__ __forceinline__ int kswz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// 32-byte chunk swizzle key of k-row `krow` of an m-major tile with R columns: the 8 k-rows a
// half-wave touches in one ds_read_b64_tr_b16 (krow = 8 g + j: j = 0..3, two values of g) must land on distinct chunks
// modulo the 256-B bank row.  Row pitch a multiple of 256 B (R = 128, 256): 8 distinct keys; pitch = 128 mod 256
// (R = 64, 192): even / odd rows are already on different bank halves, 4 keys (which keep a chunk inside its aligned
// group of four, so the key never leaves a 12-chunk row).  mkey(krow) == mkey(krow + 4) for krow % 8 < 4.
template <int R>
__device__ __forceinline__ int mkey(int krow) {
  return R % 128 == 0 ? ((krow & 3) | (((krow >> 3) & 1) << 2)) : (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1));
}

// Swizzle key of an m-major tile read as 32-column fragments (v_mfma_f32_32x32x16_bf16 operands: a half-wave = two 16-lane
// groups on the SAME four k-rows, columns c0 .. c0+15 and c0+16 .. c0+31): the eight 32-byte pieces (k-row j, column block e)
// of one ds_read_b64_tr_b16 must land on distinct bank octets.  Pitch a multiple of 256 B: chunk (cb + e) ^ 2 j;
// pitch = 128 mod 256 (R = 64, 192): odd k-rows already sit on the other bank half, key = krow & 2.
template <int R>
__device__ __forceinline__ int mkey32(int krow) {
  return R % 128 == 0 ? ((krow & 3) << 1) : (krow & 2);
}

template <bool TRANS, int R, int NI, int BK, bool KEY32 = false>
__device__ __forceinline__ StagePlan<NI> make_plan(int wave, int lane, long ld, int extent_valid) {
  StagePlan<NI> p;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int inst = wave * NI + j;
    if constexpr (!TRANS) {
      // k-major tile [R rows][BK k]: SPR 16-B slots per row; slot s of row r holds source chunk
      // s ^ kswz(r)
      constexpr int SPR = BK / 8;
      const int row = inst * (64 / SPR) + lane / SPR;
      const int chunk = (lane % SPR) ^ kswz<BK>(row);
      p.kpos[j] = chunk * 8;
      p.voff[j] = (row < extent_valid) ? (unsigned)(row * ld * 2 + chunk * 16) : OOB;
    } else {
      // m-major tile [BK k][R cols]: one k-row = R/8 slots of 16 B; 32-B chunk c of k-row kr holds
      // source chunk c ^ mkey(kr)
      static_assert(R % 64 == 0, "m-major tiles: whole 128-byte lines per k-row (mkey / mkey32)");
      constexpr int SLOTS = R / 8;
      const int slot = inst * 64 + lane;  // 16-byte slot of the tile image (SLOTS need not divide 64: R = 192)
      const int krow = slot / SLOTS;
      const int s = slot - krow * SLOTS;
      const int col = (((s >> 1) ^ (KEY32 ? mkey32<R>(krow) : mkey<R>(krow))) << 4) + ((s & 1) << 3);
      p.kpos[j] = krow;
      p.voff[j] = (col < extent_valid) ? (unsigned)(krow * ld * 2 + col * 2) : OOB;
    }
  }
  return p;
}

template <bool TRANS, int NI, int BKS = 0>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave,
                                           const StagePlan<NI>& p, long ld, int k0, int klen) {
  // Full K-steps (every K-step but a ragged last one, and the zero-fill steps the grouped kernel issues beyond K): the K advance
  // rides on the instruction's scalar offset, the per-lane offsets (out-of-range sentinel included) are the plan's — no VALU.
  // SQ counters had 2.5 VALU instructions per MFMA in the K loop of the plain kernel, most of them this arithmetic
  // (profiles/r03/pmc_gemm_c14_vs_c15.txt).
  // BKS = the K-step depth of the caller's tile (every kpos < BKS); 0: the caller keeps the per-lane form.
  if (BKS > 0 && k0 + BKS <= klen) {  // wave-uniform
    const unsigned soff = TRANS ? (unsigned)((long)k0 * ld * 2) : (unsigned)(k0 * 2);
#pragma unroll
    for (int j = 0; j < NI; ++j) lds_dma16_s(rsrc, lds_tile + (wave * NI + j) * 1024, p.voff[j], soff);
    return;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    unsigned off = TRANS ? p.voff[j] + (unsigned)((long)k0 * ld * 2) : p.voff[j] + (unsigned)(k0 * 2);
    off = ((int)(k0 + p.kpos[j]) < klen && p.voff[j] != OOB) ? off : OOB;
    lds_dma16(rsrc, lds_tile + (wave * NI + j) * 1024, off);
  }
}

// Fragment of a k-major tile: rows r0..r0+15, k-substep ks (32 deep).  lane (i = l&15, g = l>>4)
// gets the 8 bf16 at [r0 + i][ks*32 + g*8 ..].
template <int BK>
__device__ __forceinline__ bf16x8 frag_kmajor(const char* tile, int r0, int ks, int i, int g) {
  const int row = r0 + i;
  const int slot = (ks * 4 + g) ^ kswz<BK>(row);
  return *reinterpret_cast<const bf16x8*>(tile + row * (BK * 2) + slot * 16);
}

// Fragment of an m-major tile ([64 k][R cols]): columns c0..c0+15, k-substep ks.  Two hardware
// transpose reads; within a 16-lane group, lane s supplies the address of k-row (s>>2), columns
// 4*(s&3).. and receives column (s) of the 4 rows.
template <int R>
__device__ __forceinline__ bf16x8 frag_mmajor(const char* tile, int c0, int ks, int lane) {
  const int g = lane >> 4;
  const int j = (lane & 15) >> 2;
  const int q = lane & 3;
  const int krow = ks * 32 + g * 8 + j;  // mkey<R>(krow) == mkey<R>(krow + 4)
  const char* p = tile + krow * (R * 2) + (((c0 >> 4) ^ mkey<R>(krow)) << 5) + q * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(p));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)LDS_PTR(p + 4 * (R * 2)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <bool AT, bool BT, class C>
__device__ __forceinline__ void compute_tile(const char* a_tile, const char* b_tile, int wm, int wn,
                                             int lane, f32x4 (&acc)[C::FM][C::FN]) {
  const int i = lane & 15, g = lane >> 4;
  constexpr int KS = C::BK / 32;
  // Fragment loads of the WHOLE K-step are issued before the first MFMA (register double buffer when
  // the configuration has the VGPR headroom): one exposed LDS latency per K-step instead of one per
  // half K-substep with the compiler's own just-in-time placement.
  constexpr bool PRELOAD = KS == 2 && C::WAVES_PER_SIMD <= 3;
  bf16x8 af[KS][C::FM], bfr[KS][C::FN];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int t = 0; t < C::FM; ++t)
      af[ks][t] = AT ? frag_mmajor<C::BM>(a_tile, wm * (C::FM * 16) + t * 16, ks, lane)
                     : frag_kmajor<C::BK>(a_tile, wm * (C::FM * 16) + t * 16, ks, i, g);
#pragma unroll
    for (int t = 0; t < C::FN; ++t)
      bfr[ks][t] = BT ? frag_mmajor<C::BN>(b_tile, wn * (C::FN * 16) + t * 16, ks, lane)
                      : frag_kmajor<C::BK>(b_tile, wn * (C::FN * 16) + t * 16, ks, i, g);
    if (!PRELOAD) {
#pragma unroll
      for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FN; ++ni)
          // swapped operands: D[n][m] -> lane holds row m = l&15, cols n = 4*(l>>4) + 0..3
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0);
    }
  }
  if (PRELOAD) {
    __builtin_amdgcn_sched_barrier(0);  // keep the loads above the MFMA block
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][ni], af[ks][mi], acc[mi][ni], 0, 0, 0);
  }
}

// Bias gradient on the side of a dW GEMM: the m-major A tile is dY^T, so its row sums over k are
// colsum(dY).  Run only by the waves (first tile column, wn == 0) that own the result: they re-read
// their A fragments from LDS and add them up on the VALU — kept out of compute_tile so that the
// MFMA loop of every other wave stays branch-free and within its register budget.
template <class C>
__device__ __forceinline__ void bias_rows(const char* a_tile, int wm, int lane, float (&accb)[C::FM]) {
#pragma unroll
  for (int ks = 0; ks < C::BK / 32; ++ks)
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi) {
      union { bf16x8 v; unsigned w[4]; } u;
      u.v = frag_mmajor<C::BM>(a_tile, wm * (C::FM * 16) + mi * 16, ks, lane);
      accb[mi] += ((bf16lo(u.w[0]) + bf16hi(u.w[0])) + (bf16lo(u.w[1]) + bf16hi(u.w[1]))) +
                  ((bf16lo(u.w[2]) + bf16hi(u.w[2])) + (bf16lo(u.w[3]) + bf16hi(u.w[3])));
    }
}

#define CFHIP_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// wait until at most `n` STAGES (n * LPS LDS-DMA instructions of this wave) are still in flight
template <int LPS>
__device__ __forceinline__ void wait_stages(int n) {
  switch (n) {
    case 0: CFHIP_WAIT_VMCNT(0); break;
    case 1: CFHIP_WAIT_VMCNT(1 * LPS); break;
    case 2: CFHIP_WAIT_VMCNT(2 * LPS); break;
    case 3: CFHIP_WAIT_VMCNT(3 * LPS); break;
    case 4: CFHIP_WAIT_VMCNT(4 * LPS); break;
    default: CFHIP_WAIT_VMCNT(5 * LPS); break;
  }
}

// Timing ablations (skip the in-loop DMA / the MFMAs / the stores: results are then WRONG) exist only in builds
// with -DCFHIP_ABLATE (tools/build_variant.sh ablate -DCFHIP_ABLATE -> tools/libcfhip_ablate.so); the product library has no such code.
#ifdef CFHIP_ABLATE
#define CFHIP_ABLATE_AND(cond) && (cond)
#else
#define CFHIP_ABLATE_AND(cond)
#endif


// ---- epilogue ----------------------------------------------------------------------------------------
// The MFMA result layout (lane = row l&15, 4 consecutive columns per 16x16 tile) would give 8-byte
// stores scattered over 16 rows per instruction.  Instead every wave transposes its sub-tile through
// a private LDS strip (inside `stage`, a ring slot nobody reads any more), 16 rows at a time, so that
// a lane ends up with 8 CONSECUTIVE columns of one row: residual / pre-activation traffic becomes
// 16-byte coalesced loads and every store instruction writes whole 128-byte row segments.  16-byte
// chunks are XOR-swizzled by the row (no padding: the strips of all waves exactly fill 16 KiB).
// All global traffic of the epilogue goes through buffer descriptors anchored at the tile's origin: a row beyond M
// or a column beyond N becomes an out-of-range OFFSET (loads return 0, stores are dropped by the range check), so
// the whole epilogue is straight-line code — no divergent branches, no exec-mask juggling between the passes.
// Operands the epilogue READS (residual stream, saved pre-activation) are fetched a few row passes AHEAD of their
// use into a small register ring: the stores of pass i and the loads of pass i+1 may alias as far as the compiler
// can tell, so the straightforward loop issued each pass's loads only after the previous pass's stores — one
// exposed HBM round trip (~2 us under load) per pass, 8 per tile.  With the ring the round trips overlap.
struct AuxRegs { u32x4 a, b; };  // f32 operand: 8 values (a, b); bf16 operand: 8 values in a (b is dead code)
#ifndef CFHIP_PF_F32
#define CFHIP_PF_F32 2
#endif
#ifndef CFHIP_PF_BF16
#define CFHIP_PF_BF16 4
#endif

// cache policy of the epilogue's global accesses (buffer aux bits: 1 = sc0, 2 = nt, 16 = sc1)
#ifndef CFHIP_ST_AUX
#define CFHIP_ST_AUX 0
#endif
#ifndef CFHIP_LD_AUX
#define CFHIP_LD_AUX 0
#endif
__device__ __forceinline__ u32x4 bload16(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, CFHIP_LD_AUX));
}
__device__ __forceinline__ u32x2 bload8(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, CFHIP_LD_AUX));
}
__device__ __forceinline__ void bstore16(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), r, (int)off, 0, CFHIP_ST_AUX);
}
__device__ __forceinline__ void bstore8(__amdgpu_buffer_rsrc_t r, unsigned off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), r, (int)off, 0, CFHIP_ST_AUX);
}
// The saved pre-activation of the GELU epilogues is written in the forward and read once, a whole backward later (ViT-B/16 at batch 128:
// 77 MB per block): non-temporal on both sides (aux bit 2), so that it does not push the next GEMM's operands out of L2 / MALL.
#ifdef CFHIP_EPI_AUX_TEMPORAL  // A/B builds: the plain accesses of rounds 1-5
#define CFHIP_PRE_AUX 0
#else
#define CFHIP_PRE_AUX 2
#endif
__device__ __forceinline__ u32x4 bload16_pre(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, CFHIP_PRE_AUX));
}
__device__ __forceinline__ u32x2 bload8_pre(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, CFHIP_PRE_AUX));
}
__device__ __forceinline__ void bstore16_pre(__amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, v), r, (int)off, 0, CFHIP_PRE_AUX);
}
__device__ __forceinline__ void bstore8_pre(__amdgpu_buffer_rsrc_t r, unsigned off, u32x2 v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned int, v), r, (int)off, 0, CFHIP_PRE_AUX);
}
// descriptor of an [M][ld] matrix of ES-byte elements, anchored at (m0, n0); valid bytes end with element (M-1, N-1)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, long ld, int es, int m0, int n0, int M, int N) {
  const char* origin = reinterpret_cast<const char*>(base) + ((long)m0 * ld + n0) * es;
  return make_rsrc(origin, ((long)(M - 1 - m0) * ld + (N - n0)) * es);
}

// N8: N % 8 == 0, every lane's 8 columns are all inside or all outside the matrix (one 16-byte access for bf16)
// QUICK: quick GELU x * sigmoid(1.702 x) instead of the exact-erf GELU (GELU / DGELU epilogues)
template <int EPI, class C, bool F32, bool N8, bool QUICK>
__device__ __forceinline__ void epilogue_impl(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                              int m0, int n0, int wm, int wn, int wave, int lane) {
  constexpr int WCOLS = C::FN * 16;           // columns of the wave's sub-tile
  constexpr int LPR = WCOLS / 8;              // lanes per row when every lane takes 8 columns
  constexpr int RPP = 64 / LPR;               // rows covered by one pass of the wave
  constexpr int PPM = 16 / RPP;               // passes per 16-row fragment
  constexpr int NIT = C::FM * PPM;            // passes per tile
  constexpr bool HAS_AUX = EPI == CFHIP_EPI_RESIDUAL || EPI == CFHIP_EPI_DGELU;
  constexpr bool AUX_F32 = EPI == CFHIP_EPI_RESIDUAL && F32;  // f32 residual stream: aux_in is f32 with the output's layout
  constexpr int PFW = AUX_F32 ? CFHIP_PF_F32 : CFHIP_PF_BF16;  // ring depth: 8 / 4 registers per entry
  constexpr int PF = HAS_AUX ? (PFW < NIT ? PFW : NIT) : 1;
  constexpr int ES = F32 ? 4 : 2, AES = AUX_F32 ? 4 : 2;
  float* stg = reinterpret_cast<float*>(stage) + wave * (16 * WCOLS);
  const int i = lane & 15, g = lane >> 4;
  const int rr = lane / LPR, c8 = lane % LPR;
  const int lcol = wn * WCOLS + c8 * 8;                       // column inside the tile
  const bool c_lo = n0 + lcol < p.N, c_hi = n0 + lcol + 4 < p.N;  // N % 4 == 0 on this path
  const int lrow0 = wm * (C::FM * 16) + rr;                   // row inside the tile of pass 0
  f32x4 b_lo = {0.f, 0.f, 0.f, 0.f}, b_hi = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr) {
    if (c_lo) b_lo = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcol);
    if (c_hi) b_hi = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcol + 4);
  }
  const __amdgpu_buffer_rsrc_t c_rsrc = tile_rsrc(p.C, p.ldc, ES, m0, n0, p.M, p.N);
  __amdgpu_buffer_rsrc_t x_rsrc = c_rsrc, o_rsrc = c_rsrc;
  if constexpr (HAS_AUX) x_rsrc = tile_rsrc(p.aux_in, p.ldc, AES, m0, n0, p.M, p.N);
  if constexpr (EPI == CFHIP_EPI_GELU) o_rsrc = tile_rsrc(p.aux_out, p.ldc, 2, m0, n0, p.M, p.N);
  const bool has_pre = EPI == CFHIP_EPI_GELU && p.aux_out != nullptr;
  // element offset of pass `it` (rows beyond M fall outside the descriptor by themselves)
  auto eoff = [&](int it) -> unsigned { return (unsigned)((lrow0 + (it / PPM) * 16 + (it % PPM) * RPP) * (int)p.ldc + lcol); };
  auto load_aux = [&](int it) -> AuxRegs {
    AuxRegs r;
    r.b = u32x4{0u, 0u, 0u, 0u};
    const unsigned e = eoff(it);
    if constexpr (AUX_F32) {
      r.a = bload16(x_rsrc, c_lo ? e * 4u : OOB);
      r.b = bload16(x_rsrc, c_hi ? e * 4u + 16u : OOB);
    } else if constexpr (EPI == CFHIP_EPI_DGELU) {  // the saved pre-activation: read once
      if constexpr (N8) {
        r.a = bload16_pre(x_rsrc, c_lo ? e * 2u : OOB);
      } else {
        const u32x2 h0 = bload8_pre(x_rsrc, c_lo ? e * 2u : OOB), h1 = bload8_pre(x_rsrc, c_hi ? e * 2u + 8u : OOB);
        r.a = u32x4{h0[0], h0[1], h1[0], h1[1]};
      }
    } else if constexpr (N8) {
      r.a = bload16(x_rsrc, c_lo ? e * 2u : OOB);
    } else {
      const u32x2 h0 = bload8(x_rsrc, c_lo ? e * 2u : OOB), h1 = bload8(x_rsrc, c_hi ? e * 2u + 8u : OOB);
      r.a = u32x4{h0[0], h0[1], h1[0], h1[1]};
    }
    return r;
  };
  AuxRegs ring[PF];
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int it = 0; it < PF; ++it) ring[it] = load_aux(it);
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int mi = it / PPM, ps = it % PPM;
    if (ps == 0) {
#pragma unroll
      for (int ni = 0; ni < C::FN; ++ni)
        *reinterpret_cast<f32x4*>(stg + i * WCOLS + (((ni * 4 + g) ^ (i & 7)) << 2)) = acc[mi][ni];
    }
    const int r = ps * RPP + rr;
    f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * WCOLS + (((2 * c8) ^ (r & 7)) << 2));
    f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * WCOLS + (((2 * c8 + 1) ^ (r & 7)) << 2));
    AuxRegs aux;
    if constexpr (HAS_AUX) {
      aux = ring[it % PF];
      if (it + PF < NIT) ring[it % PF] = load_aux(it + PF);
    }
    const unsigned e = eoff(it);
    lo += b_lo;
    hi += b_hi;
    if constexpr (EPI == CFHIP_EPI_GELU) {
      // GELU of the bf16-rounded pre-activation (what the saved tensor holds for backward)
      const u32x4 w = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                       pack_bf16x2(hi[2], hi[3])};
      if constexpr (N8) {
        bstore16_pre(o_rsrc, (has_pre && c_lo) ? e * 2u : OOB, w);
      } else {
        bstore8_pre(o_rsrc, (has_pre && c_lo) ? e * 2u : OOB, u32x2{w[0], w[1]});
        bstore8_pre(o_rsrc, (has_pre && c_hi) ? e * 2u + 8u : OOB, u32x2{w[2], w[3]});
      }
      if constexpr (QUICK) {
        lo = f32x4{quick_gelu_f(bf16lo(w[0])), quick_gelu_f(bf16hi(w[0])), quick_gelu_f(bf16lo(w[1])), quick_gelu_f(bf16hi(w[1]))};
        hi = f32x4{quick_gelu_f(bf16lo(w[2])), quick_gelu_f(bf16hi(w[2])), quick_gelu_f(bf16lo(w[3])), quick_gelu_f(bf16hi(w[3]))};
      } else {
        lo = f32x4{gelu_erf_f(bf16lo(w[0])), gelu_erf_f(bf16hi(w[0])), gelu_erf_f(bf16lo(w[1])), gelu_erf_f(bf16hi(w[1]))};
        hi = f32x4{gelu_erf_f(bf16lo(w[2])), gelu_erf_f(bf16hi(w[2])), gelu_erf_f(bf16lo(w[3])), gelu_erf_f(bf16hi(w[3]))};
      }
    } else if constexpr (AUX_F32) {
      lo += __builtin_bit_cast(f32x4, aux.a);
      hi += __builtin_bit_cast(f32x4, aux.b);
    } else if constexpr (HAS_AUX) {
      const u32x4 w = aux.a;
      if constexpr (EPI == CFHIP_EPI_RESIDUAL) {
        lo += f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
        hi += f32x4{bf16lo(w[2]), bf16hi(w[2]), bf16lo(w[3]), bf16hi(w[3])};
      } else if constexpr (QUICK) {
        lo *= f32x4{quick_gelu_grad_f(bf16lo(w[0])), quick_gelu_grad_f(bf16hi(w[0])), quick_gelu_grad_f(bf16lo(w[1])), quick_gelu_grad_f(bf16hi(w[1]))};
        hi *= f32x4{quick_gelu_grad_f(bf16lo(w[2])), quick_gelu_grad_f(bf16hi(w[2])), quick_gelu_grad_f(bf16lo(w[3])), quick_gelu_grad_f(bf16hi(w[3]))};
      } else {
        lo *= f32x4{gelu_erf_grad_f(bf16lo(w[0])), gelu_erf_grad_f(bf16hi(w[0])), gelu_erf_grad_f(bf16lo(w[1])), gelu_erf_grad_f(bf16hi(w[1]))};
        hi *= f32x4{gelu_erf_grad_f(bf16lo(w[2])), gelu_erf_grad_f(bf16hi(w[2])), gelu_erf_grad_f(bf16lo(w[3])), gelu_erf_grad_f(bf16hi(w[3]))};
      }
    }
    if constexpr (F32) {
      if constexpr (EPI == CFHIP_EPI_NONE) {
        if (p.accumulate) {  // wave-uniform; the dW forms only (checked on the host)
          lo += __builtin_bit_cast(f32x4, bload16(c_rsrc, c_lo ? e * 4u : OOB));
          hi += __builtin_bit_cast(f32x4, bload16(c_rsrc, c_hi ? e * 4u + 16u : OOB));
        }
      }
#ifdef CFHIP_ABLATE
      if ((p.ablate & 8) && lo[0] != 12345.678f) continue;  // timing only: epilogue math without the store
#endif
      bstore16(c_rsrc, c_lo ? e * 4u : OOB, __builtin_bit_cast(u32x4, lo));
      bstore16(c_rsrc, c_hi ? e * 4u + 16u : OOB, __builtin_bit_cast(u32x4, hi));
    } else {
      const u32x4 w = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]),
                       pack_bf16x2(hi[2], hi[3])};
#ifdef CFHIP_ABLATE
      if ((p.ablate & 8) && w[0] != 0x12345678u) continue;  // timing only: epilogue math without the store
#endif
      if constexpr (N8) {
        bstore16(c_rsrc, c_lo ? e * 2u : OOB, w);
      } else {
        bstore8(c_rsrc, c_lo ? e * 2u : OOB, u32x2{w[0], w[1]});
        bstore8(c_rsrc, c_hi ? e * 2u + 8u : OOB, u32x2{w[2], w[3]});
      }
    }
  }
}

// f32 output (the f32 residual stream, f32 logits, the un-split dW forms): a lane takes FOUR consecutive columns per
// pass (one 16-byte access), so that the lanes of a row cover one contiguous 128-byte segment per instruction.  (With
// 8 columns per lane the f32 row needed two instructions that each touched every other 16 bytes of it: the f32
// residual epilogue cost 50 us on the 25216 x 768 x 3072 GEMM where the bf16 one costs 8.)
template <int EPI, class C>
__device__ __forceinline__ void epilogue_f32(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                             int m0, int n0, int wm, int wn, int wave, int lane) {
  static_assert(EPI == CFHIP_EPI_NONE || EPI == CFHIP_EPI_RESIDUAL, "f32 output: bias / residual / accumulate only");
  constexpr int WCOLS = C::FN * 16;
  constexpr int LPR = WCOLS / 4;              // lanes per row, 4 columns each
  constexpr int RPP = 64 / LPR;               // rows per pass
  constexpr int PPM = 16 / RPP;               // passes per 16-row fragment
  constexpr int NIT = C::FM * PPM;
  constexpr bool HAS_AUX = EPI == CFHIP_EPI_RESIDUAL;
  constexpr int PF = HAS_AUX ? (CFHIP_PF_BF16 < NIT ? CFHIP_PF_BF16 : NIT) : 1;  // 4 registers per ring entry
  float* stg = reinterpret_cast<float*>(stage) + wave * (16 * WCOLS);
  const int i = lane & 15, g = lane >> 4;
  const int rr = lane / LPR, c4 = lane % LPR;
  const int lcol = wn * WCOLS + c4 * 4;
  const bool c_ok = n0 + lcol < p.N;  // N % 4 == 0 on this path
  const int lrow0 = wm * (C::FM * 16) + rr;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias != nullptr && c_ok) bias = *reinterpret_cast<const f32x4*>(p.bias + n0 + lcol);
  const __amdgpu_buffer_rsrc_t c_rsrc = tile_rsrc(p.C, p.ldc, 4, m0, n0, p.M, p.N);
  __amdgpu_buffer_rsrc_t x_rsrc = c_rsrc;
  if constexpr (HAS_AUX) x_rsrc = tile_rsrc(p.aux_in, p.ldc, 4, m0, n0, p.M, p.N);
  auto boff = [&](int it) -> unsigned {
    return c_ok ? (unsigned)((lrow0 + (it / PPM) * 16 + (it % PPM) * RPP) * (int)p.ldc + lcol) * 4u : OOB;
  };
  u32x4 ring[PF];
  if constexpr (HAS_AUX) {
#pragma unroll
    for (int it = 0; it < PF; ++it) ring[it] = bload16(x_rsrc, boff(it));
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int mi = it / PPM, ps = it % PPM;
    if (ps == 0) {
#pragma unroll
      for (int ni = 0; ni < C::FN; ++ni)
        *reinterpret_cast<f32x4*>(stg + i * WCOLS + (((ni * 4 + g) ^ (i & 7)) << 2)) = acc[mi][ni];
    }
    const int r = ps * RPP + rr;
    f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * WCOLS + ((c4 ^ (r & 7)) << 2));
    v += bias;
    if constexpr (HAS_AUX) {
      v += __builtin_bit_cast(f32x4, ring[it % PF]);
      if (it + PF < NIT) ring[it % PF] = bload16(x_rsrc, boff(it + PF));
    }
    const unsigned off = boff(it);
    if constexpr (EPI == CFHIP_EPI_NONE) {
      if (p.accumulate) v += __builtin_bit_cast(f32x4, bload16(c_rsrc, off));  // wave-uniform; the dW forms only
    }
#ifdef CFHIP_ABLATE
    if ((p.ablate & 8) && v[0] != 12345.678f) continue;  // timing only: epilogue math without the store
#endif
    bstore16(c_rsrc, off, __builtin_bit_cast(u32x4, v));
  }
}

// split-K partials: the raw accumulators of K slice z go to slab z (f32), same LDS transposition, 4 columns per lane
template <class C>
__device__ __forceinline__ void epilogue_slab(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                              int m0, int n0, int z, int wm, int wn, int wave, int lane) {
  constexpr int WCOLS = C::FN * 16, LPR = WCOLS / 4, RPP = 64 / LPR;
  float* stg = reinterpret_cast<float*>(stage) + wave * (16 * WCOLS);
  const int i = lane & 15, g = lane >> 4;
  const int rr = lane / LPR, c4 = lane % LPR;
  const int lcol = wn * WCOLS + c4 * 4;
  const bool c_ok = n0 + lcol < p.N;
  const __amdgpu_buffer_rsrc_t s_rsrc = tile_rsrc(p.slabs + (long)z * p.M * p.N, p.N, 4, m0, n0, p.M, p.N);
#pragma unroll
  for (int mi = 0; mi < C::FM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni)
      *reinterpret_cast<f32x4*>(stg + i * WCOLS + (((ni * 4 + g) ^ (i & 7)) << 2)) = acc[mi][ni];
#pragma unroll
    for (int ps = 0; ps < 16 / RPP; ++ps) {
      const int r = ps * RPP + rr;
      const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * WCOLS + ((c4 ^ (r & 7)) << 2));
      const unsigned e = (unsigned)((wm * (C::FM * 16) + mi * 16 + r) * p.N + lcol);
      bstore16(s_rsrc, c_ok ? e * 4u : OOB, __builtin_bit_cast(u32x4, v));
    }
  }
}

// Wave sub-tiles whose width does not divide a wave's 64 lanes into whole rows (80 columns per wave: the convolution tiles
// Cfg<128, 320, 2, 4, ..> / Cfg<128, 160, 2, 2, ..>).  The 16 x WCOLS strip of a row fragment is handed out as a flat list of
// 16-byte pieces (8 bf16 columns, or 4 f32 columns of a split-K slab), 64 per pass; rows are padded by 4 floats instead of the
// XOR swizzle (20 chunks per row are not a power of two).  Bias + bf16 output or split-K slabs: what the convolutions need.
template <class C>
__device__ __forceinline__ void epilogue_flat(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                              int m0, int n0, int z, int wm, int wn, int wave, int lane) {
  constexpr int WCOLS = C::FN * 16, PITCH = WCOLS + 4;
  float* stg = reinterpret_cast<float*>(stage) + wave * (16 * PITCH);
  const int i = lane & 15, g = lane >> 4;
  if (p.slabs != nullptr) {
    constexpr int CPR = WCOLS / 4, NCH = 16 * CPR, NP = (NCH + 63) / 64;
    const __amdgpu_buffer_rsrc_t s_rsrc = tile_rsrc(p.slabs + (long)z * p.M * p.N, p.N, 4, m0, n0, p.M, p.N);
    int srow[NP], scol[NP];
    bool sok[NP];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int c = ps * 64 + lane;
      srow[ps] = c / CPR;
      scol[ps] = (c - srow[ps] * CPR) * 4;
      sok[ps] = c < NCH && n0 + wn * WCOLS + scol[ps] < p.N;
    }
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi) {
#pragma unroll
      for (int ni = 0; ni < C::FN; ++ni) *reinterpret_cast<f32x4*>(stg + i * PITCH + ni * 16 + g * 4) = acc[mi][ni];
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int r = sok[ps] ? srow[ps] : 0;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + r * PITCH + (sok[ps] ? scol[ps] : 0));
        const unsigned e = (unsigned)((wm * (C::FM * 16) + mi * 16 + srow[ps]) * p.N + wn * WCOLS + scol[ps]);
        bstore16(s_rsrc, sok[ps] ? e * 4u : OOB, __builtin_bit_cast(u32x4, v));
      }
    }
    return;
  }
  constexpr int CPR = WCOLS / 8, NCH = 16 * CPR, NP = (NCH + 63) / 64;
  const __amdgpu_buffer_rsrc_t c_rsrc = tile_rsrc(p.C, p.ldc, 2, m0, n0, p.M, p.N);
  int srow[NP], scol[NP];
  bool sok[NP];
  f32x4 b_lo[NP], b_hi[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    const int c = ps * 64 + lane;
    srow[ps] = c / CPR;
    scol[ps] = (c - srow[ps] * CPR) * 8;
    const int col = n0 + wn * WCOLS + scol[ps];
    sok[ps] = c < NCH && col < p.N;  // N % 8 == 0 on this path: the 8 columns are all inside or all outside
    b_lo[ps] = b_hi[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && sok[ps]) {
      b_lo[ps] = *reinterpret_cast<const f32x4*>(p.bias + col);
      b_hi[ps] = *reinterpret_cast<const f32x4*>(p.bias + col + 4);
    }
  }
#pragma unroll
  for (int mi = 0; mi < C::FM; ++mi) {
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni) *reinterpret_cast<f32x4*>(stg + i * PITCH + ni * 16 + g * 4) = acc[mi][ni];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
      const int r = sok[ps] ? srow[ps] : 0, cc = sok[ps] ? scol[ps] : 0;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cc) + b_lo[ps];
      const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * PITCH + cc + 4) + b_hi[ps];
      const u32x4 w = {pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3])};
      const unsigned e = (unsigned)((wm * (C::FM * 16) + mi * 16 + srow[ps]) * (int)p.ldc + wn * WCOLS + scol[ps]);
      bstore16(c_rsrc, sok[ps] ? e * 2u : OOB, w);
    }
  }
}

template <class C>
constexpr bool kFlatEpilogue = 64 % (C::FN * 2) != 0;

template <int EPI, class C>
__device__ __forceinline__ void epilogue_std(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                             int m0, int n0, int z, int wm, int wn, int wave, int lane) {
  if constexpr (EPI == CFHIP_EPI_NONE) {  // split-K (host-checked: epilogue NONE only)
    if (p.slabs != nullptr) {
      epilogue_slab<C>(p, acc, stage, m0, n0, z, wm, wn, wave, lane);
      return;
    }
  }
  if constexpr (EPI == CFHIP_EPI_GELU || EPI == CFHIP_EPI_DGELU) {  // bf16 outputs (checked on the host)
    if (p.quick) {
      if ((p.N & 7) == 0) epilogue_impl<EPI, C, false, true, true>(p, acc, stage, m0, n0, wm, wn, wave, lane);
      else epilogue_impl<EPI, C, false, false, true>(p, acc, stage, m0, n0, wm, wn, wave, lane);
    } else {
      if ((p.N & 7) == 0) epilogue_impl<EPI, C, false, true, false>(p, acc, stage, m0, n0, wm, wn, wave, lane);
      else epilogue_impl<EPI, C, false, false, false>(p, acc, stage, m0, n0, wm, wn, wave, lane);
    }
  } else {
    if (p.out_f32) epilogue_f32<EPI, C>(p, acc, stage, m0, n0, wm, wn, wave, lane);
    else if ((p.N & 7) == 0) epilogue_impl<EPI, C, false, true, false>(p, acc, stage, m0, n0, wm, wn, wave, lane);
    else epilogue_impl<EPI, C, false, false, false>(p, acc, stage, m0, n0, wm, wn, wave, lane);
  }
}

template <int EPI, class C>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
                                         int m0, int n0, int z, int wm, int wn, int wave, int lane) {
  if constexpr (kFlatEpilogue<C>) {
    static_assert(!kFlatEpilogue<C> || EPI == CFHIP_EPI_NONE, "80-column wave tiles: bias / bf16 output / split-K slabs only");
    epilogue_flat<C>(p, acc, stage, m0, n0, z, wm, wn, wave, lane);
  } else {
    epilogue_std<EPI, C>(p, acc, stage, m0, n0, z, wm, wn, wave, lane);
  }
}

}  // namespace
