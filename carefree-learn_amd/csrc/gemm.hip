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
  static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
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
  static_assert(R % 64 == 0, "m-major tiles: whole 128-byte lines per k-row");
  return R % 128 == 0 ? ((krow & 3) | (((krow >> 3) & 1) << 2)) : (((krow >> 1) & 1) | (((krow >> 3) & 1) << 1));
}

template <bool TRANS, int R, int NI, int BK>
__device__ __forceinline__ StagePlan<NI> make_plan(int wave, int lane, long ld, int extent_valid) {
  StagePlan<NI> p;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int inst = wave * NI + j;
    if (!TRANS) {
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
      constexpr int SLOTS = R / 8;
      const int slot = inst * 64 + lane;  // 16-byte slot of the tile image (SLOTS need not divide 64: R = 192)
      const int krow = slot / SLOTS;
      const int s = slot - krow * SLOTS;
      const int col = (((s >> 1) ^ mkey<R>(krow)) << 4) + ((s & 1) << 3);
      p.kpos[j] = krow;
      p.voff[j] = (col < extent_valid) ? (unsigned)(krow * ld * 2 + col * 2) : OOB;
    }
  }
  return p;
}

template <bool TRANS, int NI>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave,
                                           const StagePlan<NI>& p, long ld, int k0, int klen) {
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    unsigned off = TRANS ? p.voff[j] + (unsigned)((long)k0 * ld * 2) : p.voff[j] + (unsigned)(k0 * 2);
    off = ((int)(k0 + p.kpos[j]) < klen && p.voff[j] != OOB) ? off : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (wave * NI + j) * 1024), 16, off, 0, 0, 0);
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
        bstore16(o_rsrc, (has_pre && c_lo) ? e * 2u : OOB, w);
      } else {
        bstore8(o_rsrc, (has_pre && c_lo) ? e * 2u : OOB, u32x2{w[0], w[1]});
        bstore8(o_rsrc, (has_pre && c_hi) ? e * 2u + 8u : OOB, u32x2{w[2], w[3]});
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

template <int EPI, class C>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x4 (&acc)[C::FM][C::FN], char* stage,
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

// Work item = (output tile, K slice).  Everything a wave needs to stream one item.
template <bool AT, bool BT, class C>
struct ItemCtx {
  int m0, n0, z, klen, nk, tile_n;
  __amdgpu_buffer_rsrc_t a_rsrc, b_rsrc;
  StagePlan<C::A_INSTR> pa;
  StagePlan<C::B_INSTR> pb;
  unsigned yx[C::A_INSTR];  // CONV 1: (y << 16 | x) of the pixel behind each A-staging instruction of this lane
  int bshift[C::B_INSTR];   // CONV 2: byte offset of (tap shift, channel) + this lane's k-row, at K-step 0
  int btap[C::B_INSTR];     // CONV 2: (dy + 1) | (dx + 1) << 2
  int kb;
};

template <bool AT, bool BT, class C, int CONV = 0>
__device__ __forceinline__ ItemCtx<AT, BT, C> setup_item(const GemmParams& p, int item, int wave, int lane) {
  ItemCtx<AT, BT, C> c;
  const int ntiles = p.tiles_m * p.tiles_n;
  c.z = item / ntiles;
  const int tile = item - c.z * ntiles;
  int tile_m;
  if (p.group_n > 0) {
    // column groups: the workgroups resident on an XCD at one time share `group_n` B panels (kept hot in its L2) while
    // the A panels stream through once per group
    const int per_group = p.group_n * p.tiles_m;
    const int grp = tile / per_group;
    const int first_n = grp * p.group_n;
    const int gw = min(p.group_n, p.tiles_n - first_n);
    const int r = tile - grp * per_group;
    tile_m = r / gw;
    c.tile_n = first_n + (r - tile_m * gw);
  } else if (p.group_n < 0) {
    // row groups of -group_n panels, walked column by column (rows fastest): the A panels of a group stay hot
    const int gm = -p.group_n;
    const int per_group = gm * p.tiles_n;
    const int grp = tile / per_group;
    const int first_m = grp * gm;
    const int gh = min(gm, p.tiles_m - first_m);
    const int r = tile - grp * per_group;
    c.tile_n = r / gh;
    tile_m = first_m + (r - c.tile_n * gh);
  } else {
    tile_m = tile / p.tiles_n;
    c.tile_n = tile - tile_m * p.tiles_n;
  }
  c.m0 = tile_m * C::BM;
  c.n0 = c.tile_n * C::BN;
  const int kb = c.z * p.k_chunk;
  c.kb = kb;
  c.klen = min(p.K - kb, p.k_chunk);
  c.nk = (c.klen + C::BK - 1) / C::BK;
  // buffer descriptors relative to this tile's origin (small 32-bit offsets, range-checked)
  const int rows_a = p.M - c.m0, rows_b = p.N - c.n0;
  const bf16_t* a_base = AT ? p.A + (long)kb * p.lda + c.m0 : p.A + (long)c.m0 * p.lda + kb;
  const bf16_t* b_base = BT ? p.B + (long)kb * p.ldb + c.n0 : p.B + (long)c.n0 * p.ldb + kb;
  const long a_bytes = AT ? ((long)(c.klen - 1) * p.lda + rows_a) * 2 : ((long)(rows_a - 1) * p.lda + c.klen) * 2;
  const long b_bytes = BT ? ((long)(c.klen - 1) * p.ldb + rows_b) * 2 : ((long)(rows_b - 1) * p.ldb + c.klen) * 2;
  if constexpr (CONV == 2) {
    static_assert(CONV != 2 || (AT && BT && C::BK == 32), "implicit weight gradient: layout (1,1), 32-pixel K-steps");
    const int back = min(kb, p.conv_w + 1);
    const long span = min((long)p.K - (kb - back), (long)back + c.klen + p.conv_w + 1);
    c.b_rsrc = make_rsrc(p.B + (long)(kb - back) * p.conv_c, span * p.conv_c * 2);
    constexpr int SLOTS = C::BN / 8;
#pragma unroll
    for (int j = 0; j < C::B_INSTR; ++j) {
      const int inst = wave * C::B_INSTR + j;
      const int krow = inst * (64 / SLOTS) + lane / SLOTS;
      const int sl = lane % SLOTS;
      const int col = (((sl >> 1) ^ mkey<C::BN>(krow)) << 4) + ((sl & 1) << 3);
      const int n = c.n0 + col;  // 8 consecutive n = 8 channels of one tap (conv_c % 8 == 0)
      const int tap = n / p.conv_c;
      const int ch = n - tap * p.conv_c;
      const int ky = (tap * 11) >> 5;
      const int dy = ky - 1, dx = tap - 3 * ky - 1;
      c.pb.kpos[j] = krow;
      c.pb.voff[j] = (col < rows_b) ? 0u : OOB;
      c.bshift[j] = ((back + krow + dy * p.conv_w + dx) * p.conv_c + ch) * 2;
      c.btap[j] = (dy + 1) | ((dx + 1) << 2);
    }
  } else {
    c.b_rsrc = make_rsrc(b_base, b_bytes);
    c.pb = make_plan<BT, C::BN, C::B_INSTR, C::BK>(wave, lane, p.ldb, rows_b);
  }
  if constexpr (CONV == 1) {
    static_assert(!AT && C::BK == 32, "implicit convolution: k-major A, 32-channel K-steps");
    // The descriptor starts W+1 pixels BEFORE the tile (clamped at pixel 0) so that the (-1,-1) tap is a
    // non-negative offset; it ends W+1 pixels after it.  Taps outside the image are sent out of range per lane.
    const int back = min(c.m0, p.conv_w + 1);
    const long span = min((long)p.M - (c.m0 - back), (long)back + C::BM + p.conv_w + 1);
    c.a_rsrc = make_rsrc(p.A + (long)(c.m0 - back) * p.conv_c, span * p.conv_c * 2);
#pragma unroll
    for (int j = 0; j < C::A_INSTR; ++j) {
      const int inst = wave * C::A_INSTR + j;
      const int row = inst * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ kswz<32>(row);
      c.pa.kpos[j] = chunk * 8;
      c.pa.voff[j] = (row < rows_a) ? (unsigned)(((back + row) * p.conv_c + chunk * 8) * 2) : OOB;
      const int m = c.m0 + row;
      const int q = m / p.conv_w;
      c.yx[j] = ((unsigned)(q % p.conv_h) << 16) | (unsigned)(m - q * p.conv_w);
    }
  } else {
    c.a_rsrc = make_rsrc(a_base, a_bytes);
    c.pa = make_plan<AT, C::BM, C::A_INSTR, C::BK>(wave, lane, p.lda, rows_a);
  }
  return c;
}

// CONV: K-step `kstep` of the A operand = 32 channels (c0..) of tap (ky, kx) for the tile's BM pixels
template <class C>
__device__ __forceinline__ void stage_tile_conv(const __amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave,
                                                const StagePlan<C::A_INSTR>& pl, const unsigned (&yx)[C::A_INSTR],
                                                const GemmParams& p, int kstep) {
  const int tap = (int)(((float)kstep + 0.5f) * p.conv_inv_kpt);  // kstep / conv_kpt (exact: kstep < 2^16)
  const int c0 = (kstep - tap * p.conv_kpt) * 32;
  const int ky = (tap * 11) >> 5;  // tap / 3 for tap in 0..8
  const int dy = ky - 1, dx = tap - 3 * ky - 1;
  const int shift = ((dy * p.conv_w + dx) * p.conv_c + c0) * 2;
#pragma unroll
  for (int j = 0; j < C::A_INSTR; ++j) {
    const int y = (int)(yx[j] >> 16) + dy, x = (int)(yx[j] & 0xffffu) + dx;
    const bool ok = pl.voff[j] != OOB && (unsigned)y < (unsigned)p.conv_h && (unsigned)x < (unsigned)p.conv_w;
    const unsigned off = ok ? pl.voff[j] + (unsigned)shift : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_tile + (wave * C::A_INSTR + j) * 1024), 16, off, 0, 0, 0);
  }
}

// CONV 2: K-step `kstep` of the B operand = 32 pixels x the tile's BN (tap, channel) columns of the shifted activation
template <class C, class Ctx>
__device__ __forceinline__ void stage_tile_wgrad(const Ctx& c, char* lds_tile, int wave, const GemmParams& p, int kstep) {
  const int k0 = kstep * 32;
#pragma unroll
  for (int j = 0; j < C::B_INSTR; ++j) {
    const int kl = k0 + (int)c.pb.kpos[j];
    const unsigned pix = (unsigned)(c.kb + kl);
    const unsigned q = __umulhi(pix, p.conv_magic_w);                   // pix / W
    const int x = (int)(pix - q * (unsigned)p.conv_w) + ((c.btap[j] >> 2) - 1);
    const int y = (int)(q - __umulhi(q, p.conv_magic_h) * (unsigned)p.conv_h) + ((c.btap[j] & 3) - 1);
    const bool ok = c.pb.voff[j] != OOB && kl < c.klen && (unsigned)y < (unsigned)p.conv_h && (unsigned)x < (unsigned)p.conv_w;
    const unsigned off = ok ? (unsigned)(c.bshift[j] + k0 * p.conv_c * 2) : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.b_rsrc, LDS_PTR(lds_tile + (wave * C::B_INSTR + j) * 1024), 16, off, 0, 0, 0);
  }
}

template <bool AT, bool BT, class C, int CONV = 0>
__device__ __forceinline__ void stage_step(const ItemCtx<AT, BT, C>& c, const GemmParams& p, char* smem,
                                           int slot, int wave, int kstep) {
  char* st = smem + slot * C::STAGE_BYTES;
  if constexpr (CONV == 1) stage_tile_conv<C>(c.a_rsrc, st, wave, c.pa, c.yx, p, kstep + (c.kb >> 5));
  else stage_tile<AT>(c.a_rsrc, st, wave, c.pa, p.lda, kstep * C::BK, c.klen);
  if constexpr (CONV == 2) {
    stage_tile_wgrad<C>(c, st + C::A_BYTES, wave, p, kstep);
    return;
  }
  stage_tile<BT>(c.b_rsrc, st + C::A_BYTES, wave, c.pb, p.ldb, kstep * C::BK, c.klen);
}

// One workgroup per work item (output tile x K slice), an NSTAGE-deep LDS ring fed by LDS-DMA:
//   * the DMA of K-step t + D (D = NSTAGE-1) is issued right after the barrier of step t and stays in flight
//     ACROSS barriers: waits are counted (`s_waitcnt vmcnt(N)`), the barrier is the raw s_barrier (a
//     `__syncthreads()` would drain vmcnt to 0).  One barrier per K-step — RAW: a wave passes barrier(t) only
//     after its own DMA of step t landed; WAR: the DMA issued after barrier(t) overwrites the slot read in
//     step t-1, which every wave has left;
//   * the epilogue transposes the accumulators through the ring slot of the last K-step — see epilogue<>.
// (A persistent-grid form of this kernel with cross-tile prefetch existed in round 1; it was slower inside the
// training step — it keeps every CU slot and blocks the co-scheduling with the side streams — and kept the next
// item's context live across the K loop, which put the 128-VGPR forms into scratch.  Removed.)
// BG: the launch also owes the bias gradient (bias_rows): a separate instantiation, so that the plain dW kernel
// does not carry the row accumulators and the extra fragment reads.
template <bool AT, bool BT, int EPI, class C, int CONV = 0, bool BG = false>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD)
void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int D = C::NSTAGE - 1;  // prefetch distance
  static_assert(C::NW * 16 * (C::FN * 16) * 4 <= C::STAGE_BYTES, "epilogue staging must fit in one ring slot");

  // XCD-aware, bijective remap of the grid: XCD x (= bid % 8) owns a contiguous range of work items
  // (neighbours share A / B panels in its L2).
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;

  const ItemCtx<AT, BT, C> cur = setup_item<AT, BT, C, CONV>(p, item, wave, lane);
  const int nk = cur.nk;
  int rd = 0;  // ring slot of the next K-step to compute
#pragma unroll
  for (int st = 0; st < D; ++st)
    if (st < nk) stage_step<AT, BT, C, CONV>(cur, p, smem, st, wave, st);

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  float accb[BG ? C::FM : 1];
  if constexpr (BG) {
#pragma unroll
    for (int mi = 0; mi < C::FM; ++mi) accb[mi] = 0.f;
  }
  const bool do_bg = BG && cur.tile_n == 0 && wn == 0;

  for (int t = 0; t < nk; ++t) {
    const int younger = min(D - 1, nk - 1 - t);  // younger stages allowed to stay in flight
    if (D >= 3 && younger >= 2) CFHIP_WAIT_VMCNT(2 * C::LPS);
    else if (D >= 2 && younger == 1) CFHIP_WAIT_VMCNT(1 * C::LPS);
    else CFHIP_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();
    if (t + D < nk CFHIP_ABLATE_AND(!(p.ablate & 1))) {
      int wr = rd + D;
      if (wr >= C::NSTAGE) wr -= C::NSTAGE;
      stage_step<AT, BT, C, CONV>(cur, p, smem, wr, wave, t + D);
    }
    const char* tile = smem + rd * C::STAGE_BYTES;
    if (true CFHIP_ABLATE_AND(!(p.ablate & 2))) compute_tile<AT, BT, C>(tile, tile + C::A_BYTES, wm, wn, lane, acc);
    if constexpr (BG) {
      if (do_bg) bias_rows<C>(tile, wm, lane, accb);
    }
    rd = rd + 1 == C::NSTAGE ? 0 : rd + 1;
  }
#ifdef CFHIP_ABLATE
  if (p.ablate & 4) return;
#endif
  const int m0 = cur.m0, n0 = cur.n0, z = cur.z;
  if constexpr (BG) {
    if (do_bg) {
#pragma unroll
      for (int mi = 0; mi < C::FM; ++mi) {
        float v = accb[mi];  // lane (i, g) holds the k-slots of group g: fold the 4 groups
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int m = m0 + wm * (C::FM * 16) + mi * 16 + (lane & 15);
        if ((lane >> 4) == 0 && m < p.M) {
          if (p.bgrad_slabs != nullptr) p.bgrad_slabs[(long)z * p.M + m] = v;
          else p.bgrad[m] = p.bgrad_acc ? p.bgrad[m] + v : v;
        }
      }
    }
  }
  // ---- epilogue (see epilogue<>): the strip lives in the ring slot of the last K-step, free once every wave is
  // past the raw barrier below (all DMA retired: the last wait was vmcnt(0)).
  __builtin_amdgcn_s_barrier();
  int eslot = rd - 1;
  if (eslot < 0) eslot += C::NSTAGE;
  epilogue<EPI, C>(p, acc, smem + eslot * C::STAGE_BYTES, m0, n0, z, wm, wn, wave, lane);
}

// ---- big-tile, two-group "ping-pong" variant -----------------------------------------------------------
// 128^2 tiles move 1 byte L2 -> LDS per 64 flop: at the dense MFMA rate that is ~39 TB/s, more than the
// eight L2s deliver (the DMA-only ablation of the kernel above tops out at ~19 TB/s), so the 128^2
// kernels are L2-bandwidth bound at about half the MFMA peak.  This variant works on a 256 x 256 (or
// 256 x 128) tile with 8 waves (2 x 4), one workgroup per CU, BK = 32, a 4-slot LDS ring (DMA three
// K-steps ahead, counted vmcnt, never drained inside the loop) and splits every K-step in two PHASES
// (upper / lower half of the wave's rows):
//
//     L segment: ds_read the phase's fragments        | s_barrier |
//     M segment: 16 MFMAs (s_setprio 1)                | s_barrier |
//
// The two wave groups (wm = 0 / 1; waves w and w + 4 share a SIMD) run STAGGERED by one barrier: group 1
// executes one extra barrier before the loop, group 0 one after it.  In every barrier interval one
// group is in its M segment and the other in its L segment, so each SIMD's MFMA pipe always has a
// wave with operands in registers while its partner's LDS reads are in flight.
//
// Ordering (by count, not by luck).  Intervals are numbered by barriers; group 0 runs L(t,ph) in
// interval 4t + 2ph and M(t,ph) in 4t + 2ph + 1, group 1 one interval later.
//   RAW: the DMA of K-step t+1 (issued in M(t-2, 1)) is retired by every wave's counted vmcnt BEFORE
//        the barrier that ends its L(t, 1) (intervals 4t+2 / 4t+3); the first read of step t+1 is
//        group 0's L(t+1, 0) in interval 4t+4.
//   WAR: slot (t+3)&3 held step t-1, last read in group 1's L(t-1, 1) (interval 4t-1) and retired by
//        the lgkmcnt waits in front of its MFMAs in interval 4t; the DMA of step t+3 is issued AFTER the barrier
//        that ends L(t, 1), i.e. in interval 4t+3 (group 0) / 4t+4 (group 1).
template <bool AT, bool BT, int EPI, class C, int CONV = 0>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD)
void gemm_bf16_phase_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(C::BK == 32 && C::NSTAGE >= 3 && C::NSTAGE <= 7 && C::WM == 2, "phase kernel: BK = 32, 3-7 ring slots, 2 wave rows");
  static_assert(5 * C::LPS < 64, "vmcnt is a 6-bit counter");
  constexpr int D = C::NSTAGE - 1;  // prefetch distance (K-steps)
  static_assert(C::NW * 16 * (C::FN * 16) * 4 <= C::LDS_BYTES, "epilogue staging must fit in the (free) ring");
  // phases per K-step: two (upper / lower half of the wave's rows) when a half still carries 16 MFMAs,
  // otherwise one — a barrier pair per 8 MFMAs costs more than the overlap buys
  constexpr int PH = (C::FM / 2) * C::FN >= 16 ? 2 : 1;
  constexpr int HM = C::FM / PH;  // row fragments per phase
  const int nwg = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int i = lane & 15, g = lane >> 4;

  const ItemCtx<AT, BT, C> cur = setup_item<AT, BT, C, CONV>(p, item, wave, lane);
  const int nk = cur.nk;
  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int st = 0; st < D; ++st)
    if (st < nk) stage_step<AT, BT, C, CONV>(cur, p, smem, st, wave, st);
  wait_stages<C::LPS>(min(min(nk, D) - 1, 5));  // K-step 0 has landed; the younger ones stay in flight
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the stagger

  int rd = 0, wr = D;  // ring slots of K-steps t and t + D
  for (int t = 0; t < nk; ++t) {
    const char* a_tile = smem + rd * C::STAGE_BYTES;
    const char* b_tile = a_tile + C::A_BYTES;
    bf16x8 bfr[C::FN], af[HM];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      // ---- L segment
      if (ph == 0) {
#pragma unroll
        for (int n = 0; n < C::FN; ++n)
          bfr[n] = BT ? frag_mmajor<C::BN>(b_tile, wn * (C::FN * 16) + n * 16, 0, lane)
                      : frag_kmajor<C::BK>(b_tile, wn * (C::FN * 16) + n * 16, 0, i, g);
      }
#pragma unroll
      for (int m = 0; m < HM; ++m)
        af[m] = AT ? frag_mmajor<C::BM>(a_tile, wm * (C::FM * 16) + (ph * HM + m) * 16, 0, lane)
                   : frag_kmajor<C::BK>(a_tile, wm * (C::FM * 16) + (ph * HM + m) * 16, 0, i, g);
      if (ph == PH - 1)  // own DMA of K-step t+1 retired; steps t+2 .. t+D-1 (already issued) may stay in flight
        wait_stages<C::LPS>(max(0, min(min(D - 2, nk - 2 - t), 5)));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- M segment
      // (the compiler's own lgkmcnt ladder in front of the MFMAs retires the fragment reads)
      if (ph == PH - 1 && t + D < nk) stage_step<AT, BT, C, CONV>(cur, p, smem, wr, wave, t + D);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int m = 0; m < HM; ++m)
#pragma unroll
        for (int n = 0; n < C::FN; ++n)
          acc[ph * HM + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n], af[m], acc[ph * HM + m][n], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    rd = rd + 1 == C::NSTAGE ? 0 : rd + 1;
    wr = wr + 1 == C::NSTAGE ? 0 : wr + 1;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // every wave has executed the same number of barriers
#ifdef CFHIP_ABLATE
  if (p.ablate & 4) return;
#endif
  // all DMA retired (the last wait was vmcnt(0)), all fragment reads retired: the ring is free
  epilogue<EPI, C>(p, acc, smem, cur.m0, cur.n0, cur.z, wm, wn, wave, lane);
}

// split-K second pass: C = sum_z slab[z] (+ bias) (+ C)
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, void* C, const float* bias,
                                     int M, int N, long ldc, int splits, int out_f32, int accumulate,
                                     const float* __restrict__ bgrad_slabs, float* bgrad, int bgrad_acc) {
  const long n4 = N >> 2;
  const long total = (long)M * n4;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    if (bgrad != nullptr && idx < M) {  // the dW GEMM's fused bias gradient: reduce its slabs too
      float b = 0.f;
      for (int z = 0; z < splits; ++z) b += bgrad_slabs[(long)z * M + idx];
      bgrad[idx] = bgrad_acc ? bgrad[idx] + b : b;
    }
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

// Second pass of the implicit 3x3 weight gradient: slabs [z][Cout][tap*Cin + c] (the GEMM's n order) -> the reference's
// filter layout dW[co][c][tap], (+)=.  A lane owns 4 channels of one filter: nine 16-byte loads per slab (coalesced over
// c), 36 consecutive floats out (nine 16-byte stores).  Fixed summation order: deterministic.
__global__ void splitk_reduce_filter_kernel(const float* __restrict__ slabs, float* __restrict__ dW, int Cout, int Cin,
                                            int splits, int accumulate, const float* __restrict__ bgrad_slabs,
                                            float* bgrad, int bgrad_acc) {
  const long c4n = Cin >> 2;
  const long total = (long)Cout * c4n;
  const long N = 9L * Cin;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    if (bgrad != nullptr && idx < Cout) {
      float b = 0.f;
      for (int z = 0; z < splits; ++z) b += bgrad_slabs[(long)z * Cout + idx];
      bgrad[idx] = bgrad_acc ? bgrad[idx] + b : b;
    }
    const int co = (int)(idx / c4n);
    const int c = (int)(idx - (long)co * c4n) * 4;
    float o[36];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < splits; ++z)
        v += *reinterpret_cast<const f32x4*>(slabs + ((long)z * Cout + co) * N + (long)tap * Cin + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e * 9 + tap] = v[e];
    }
    float* dst = dW + ((long)co * Cin + c) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      f32x4 v = {o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]};
      if (accumulate) v += *reinterpret_cast<const f32x4*>(dst + 4 * i);
      *reinterpret_cast<f32x4*>(dst + 4 * i) = v;
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
    float acc = 0.f, asum = 0.f;
    for (int k = 0; k < p.K; ++k) {
      const float a = bf16_to_f32(a_trans ? p.A[(long)k * p.lda + m] : p.A[(long)m * p.lda + k]);
      const float b = bf16_to_f32(b_trans ? p.B[(long)k * p.ldb + n] : p.B[(long)n * p.ldb + k]);
      acc = fmaf(a, b, acc);
      asum += a;
    }
    if (p.bgrad != nullptr && n == 0) p.bgrad[m] = p.bgrad_acc ? p.bgrad[m] + asum : asum;
    if (p.bias != nullptr) acc += p.bias[n];
    const long off = (long)m * p.ldc + n;
    if (p.epilogue == CFHIP_EPI_GELU) {
      const float pre = bf16_to_f32(f32_to_bf16(acc));
      if (p.aux_out != nullptr) p.aux_out[off] = f32_to_bf16(acc);
      acc = p.quick ? quick_gelu_f(pre) : gelu_erf_f(pre);
    } else if (p.epilogue == CFHIP_EPI_RESIDUAL) {
      acc += p.out_f32 ? reinterpret_cast<const float*>(p.aux_in)[off] : bf16_to_f32(p.aux_in[off]);
    } else if (p.epilogue == CFHIP_EPI_DGELU) {
      acc *= p.quick ? quick_gelu_grad_f(bf16_to_f32(p.aux_in[off])) : gelu_erf_grad_f(bf16_to_f32(p.aux_in[off]));
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

// tile configurations (see Cfg): index = value of the "gemm_config" option (-1 = heuristic)
using CfgA = Cfg<128, 128, 2, 2, 2, 64>;  //  64 KiB LDS, 4 waves, 2 WG / CU, prefetch 1
using CfgB = Cfg<128, 128, 2, 2, 2, 32>;  //  32 KiB LDS, 4 waves, 4 WG / CU, prefetch 1
using CfgC = Cfg<128, 128, 2, 2, 3, 32>;  //  48 KiB LDS, 4 waves, 3 WG / CU, prefetch 2
using CfgD = Cfg<128, 64, 2, 2, 2, 64>;   //  48 KiB LDS, 4 waves (64x32 each), 3 WG / CU, prefetch 1
using CfgE = Cfg<128, 128, 2, 2, 4, 32>;  //  64 KiB LDS, 4 waves, 2 WG / CU, prefetch 3
using CfgF = Cfg<128, 64, 2, 2, 2, 32>;   //  24 KiB LDS, 4 waves (64x32 each), 4 WG / CU, prefetch 1
using CfgG = Cfg<128, 64, 2, 2, 3, 32>;   //  36 KiB LDS, 4 waves (64x32 each), 4 WG / CU, prefetch 2
using CfgP = Cfg<256, 256, 2, 4, 4, 32>;  // 128 KiB LDS, 8 waves, 1 WG / CU, prefetch 3: two-group phase kernel
using CfgQ = Cfg<256, 128, 2, 4, 3, 32>;  //  72 KiB LDS, 8 waves (128x32 each), 2 WG / CU
using CfgR = Cfg<256, 128, 2, 2, 3, 32>;  //  72 KiB LDS, 4 waves (128x64 each), 2 WG / CU, prefetch 2
using CfgS = Cfg<128, 256, 2, 2, 3, 32>;  //  72 KiB LDS, 4 waves (64x128 each), 2 WG / CU, prefetch 2
using CfgT = Cfg<256, 128, 2, 4, 6, 32>;  // 144 KiB LDS, 8 waves, ONE WG / CU, prefetch 5 (probe: can one workgroup feed a CU?)
using CfgU = Cfg<256, 256, 2, 4, 5, 32>;  // 160 KiB LDS, 8 waves, ONE WG / CU, prefetch 4
using CfgW = Cfg<256, 256, 2, 4, 2, 64>;  // 128 KiB LDS, 8 waves (128x64 each), ONE WG / CU, BK = 64 (plain kernel)
using CfgY = Cfg<192, 128, 2, 4, 2, 64>;  //  80 KiB LDS, 8 waves (96x32 each), 2 WG / CU, BK = 64 (plain kernel)
constexpr int NUM_CFG = 15;  // 7 .. 12 = CfgP / CfgQ / CfgR / CfgS / CfgT / CfgU on the phase kernel; 13 / 14 = CfgW / CfgY
constexpr int BK_MAX = 64;

int g_gemm_config = -1;
#ifdef CFHIP_ABLATE
int g_gemm_ablate = 0;
#endif
int g_gemm_heuristic = 6;
int g_gemm_group_n = 8;

template <bool AT, bool BT, int EPI, class C, bool PIPE, int CONV = 0>
int launch_cfg(const GemmParams& p, dim3 grid, hipStream_t s) {
  void (*kern)(GemmParams) = nullptr;
  if (PIPE && p.bgrad != nullptr) {
    cfhip_set_error("gemm: the fused bias gradient is not provided by the big-tile kernel (gemm_config 7 / 8)");
    return CFHIP_ERR_INVALID;
  }
  if constexpr (PIPE) kern = gemm_bf16_phase_kernel<AT, BT, EPI, C, CONV>;
  else if constexpr (AT && BT && EPI == CFHIP_EPI_NONE) {
    if (p.bgrad != nullptr) kern = gemm_bf16_kernel<AT, BT, EPI, C, CONV, true>;
    else kern = gemm_bf16_kernel<AT, BT, EPI, C, CONV, false>;
  } else kern = gemm_bf16_kernel<AT, BT, EPI, C, CONV>;
  static bool attr_done_bg[2] = {false, false};  // per instantiation and per kernel picked above
  bool& attr_done = attr_done_bg[p.bgrad != nullptr ? 1 : 0];
  if (!attr_done && C::LDS_BYTES > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      cfhip_set_error("gemm: cannot reserve %d bytes of LDS: %s", C::LDS_BYTES, hipGetErrorString(e));
      return CFHIP_ERR_LAUNCH;
    }
    attr_done = true;
  }
  GemmParams q = p;
  // tile walk order (profiles/r02/gemm_group_sweep_b128.log, gemm_group_step_ab.log): outputs wider than 8 tile columns are
  // walked in groups of 8 columns (whole step -1.4 %; per-shape group widths were better in isolation, not in the step)
  const int gn = g_gemm_group_n;
  q.group_n = (gn > 0 && gn < p.tiles_n) || (gn < 0 && p.tiles_m > 1) ? gn : 0;
  hipLaunchKernelGGL(kern, grid, dim3(C::NT), C::LDS_BYTES, s, q);
  return CFHIP_OK;
}

template <class C, bool PIPE = false>
int launch_layout(GemmParams p, int a_trans, int b_trans, int epilogue, int split_k, hipStream_t s) {
  p.tiles_m = (p.M + C::BM - 1) / C::BM;
  p.tiles_n = (p.N + C::BN - 1) / C::BN;
  const long a_span = a_trans ? (long)p.k_chunk * p.lda * 2 : (long)C::BM * p.lda * 2;
  const long b_span = b_trans ? (long)p.k_chunk * p.ldb * 2 : (long)C::BN * p.ldb * 2;
  CFHIP_REQUIRE(a_span < 0x7fffffffL && b_span < 0x7fffffffL,
                "gemm: operand tile span exceeds 2 GiB (lda=%ld ldb=%ld K=%d)", p.lda, p.ldb, p.K);
  p.splits = split_k;
  dim3 grid(p.tiles_m * p.tiles_n * split_k);  // one workgroup per (output tile, K slice)
  if (!a_trans && !b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE: return launch_cfg<false, false, CFHIP_EPI_NONE, C, PIPE>(p, grid, s);
      case CFHIP_EPI_GELU: return launch_cfg<false, false, CFHIP_EPI_GELU, C, PIPE>(p, grid, s);
      case CFHIP_EPI_RESIDUAL: return launch_cfg<false, false, CFHIP_EPI_RESIDUAL, C, PIPE>(p, grid, s);
      default: break;
    }
  } else if (!a_trans && b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE: return launch_cfg<false, true, CFHIP_EPI_NONE, C, PIPE>(p, grid, s);
      case CFHIP_EPI_DGELU: return launch_cfg<false, true, CFHIP_EPI_DGELU, C, PIPE>(p, grid, s);
      default: break;
    }
  } else if (epilogue == CFHIP_EPI_NONE) {
    return launch_cfg<true, true, CFHIP_EPI_NONE, C, PIPE>(p, grid, s);
  }
  cfhip_set_error("gemm: epilogue %d is not provided for layout (%d,%d)", epilogue, a_trans, b_trans);
  return CFHIP_ERR_INVALID;
}

// Shape-aware tile choice.  What the K loops are bound by is the L2 -> LDS line traffic (round 2, profiles/README.md): a
// k-major operand staged 32 k per step uses HALF of every 128-byte line it pulls through the CU's L1 (the other half is
// gone again one K-step later), so the forward (nt) forms take BK = 64 configurations; the choice between configurations
// that are close in isolation was made by A/B runs of the whole training step (gemm_heuristic_step_ab.log), where the
// dX GEMMs share the chip with the dW GEMMs of the side stream:
//   6 (default): dW (tn) -> 128x128x32, 4 WG / CU; dX (nn) -> 256x128x32 phase kernel; forward (nt): N >= 2560 ->
//                192x128x64, otherwise 128x128x64
//   5: round-2-start table (every M >= 1024 GEMM on the 256x128x32 phase kernel); 1 .. 4: the round-1 tables
int pick_config(int M, int N, int a_trans, int b_trans) {
  if (g_gemm_config >= 0 && g_gemm_config < NUM_CFG) return g_gemm_config;
  if (g_gemm_heuristic >= 6) {
    if (a_trans) return 1;  // (192x128x64 is 10 % faster alone on the dW forms and 1.2 % slower in the step)
    if (M >= 1024) {
      if (b_trans) return 8;
      return N >= 2560 ? 14 : 0;
    }
    if (b_trans) return 1;
    return N <= 1024 ? 3 : 0;
  }
  if (g_gemm_heuristic == 1) {
    if (a_trans && (long)M * N <= 768L * 768L) return 0;  // small dW outputs: few tiles, deep split-K
    if (a_trans || b_trans) return 1;
    return N <= 1024 ? 1 : 0;
  }
  if (g_gemm_heuristic >= 2) {  // + the 256x128 two-group kernel for the wide outputs
    if (a_trans) return 1;
    if (N >= 2560 && M >= 1024) return 8;
    if (g_gemm_heuristic >= 4 && b_trans && M >= 1024) return 8;  // every dX GEMM
    if (g_gemm_heuristic >= 5 && M >= 1024) return 8;             // and every forward GEMM
    if (g_gemm_heuristic >= 3) {  // 128x128x64 once it fills the 512 resident slots at least twice
      const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
      if (tiles128 >= 1024) return 0;
    }
  }
  if (a_trans || b_trans) return 1;   // 128x128x32, 4 WG / CU
  if (N <= 1024) return 3;            // 128x64x64, 3 WG / CU
  return 0;                           // 128x128x64, 2 WG / CU
}

}  // namespace

#ifdef CFHIP_ABLATE
int cfhip_internal_set_attn_ablate(int v);  // attn.hip
#endif
int cfhip_internal_set_ln_fused(int v);  // norm.hip
int cfhip_internal_set_attn_persistent(int v);  // attn.hip

extern "C" int cfhip_set_option(const char* name, int value) {
  if (name != nullptr && strcmp(name, "gemm_config") == 0) {
    g_gemm_config = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "gemm_heuristic") == 0) {
    g_gemm_heuristic = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "gemm_group_n") == 0) {
    g_gemm_group_n = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "ln_bwd_fused") == 0) return cfhip_internal_set_ln_fused(value);
  if (name != nullptr && strcmp(name, "attn_persistent") == 0) return cfhip_internal_set_attn_persistent(value);
#ifdef CFHIP_ABLATE
  if (name != nullptr && strcmp(name, "gemm_ablate") == 0) {
    g_gemm_ablate = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "attn_ablate") == 0) return cfhip_internal_set_attn_ablate(value);
#endif
  cfhip_set_error("set_option: unknown option '%s'", name ? name : "(null)");
  return CFHIP_ERR_INVALID;
}

extern "C" int cfhip_gemm_bf16(const void* A, const void* B, void* C, const float* bias,
                               const void* aux_in, void* aux_out, int M, int N, int K, int64_t lda,
                               int64_t ldb, int64_t ldc, int a_trans, int b_trans, int epilogue,
                               int out_dtype, int accumulate, int split_k, void* workspace,
                               size_t workspace_bytes, float* bias_grad, int bias_grad_accumulate,
                               void* stream) {
  CFHIP_REQUIRE(A && B && C, "gemm: null operand");
  CFHIP_REQUIRE(bias_grad == nullptr || (a_trans && b_trans),
                "gemm: the fused bias gradient exists for layout (1,1) only");
  CFHIP_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
  CFHIP_REQUIRE(epilogue >= 0 && epilogue <= 5, "gemm: bad epilogue %d", epilogue);
  const int quick = epilogue == CFHIP_EPI_QGELU || epilogue == CFHIP_EPI_DQGELU;
  if (epilogue == CFHIP_EPI_QGELU) epilogue = CFHIP_EPI_GELU;      // same kernels, activation picked at run time
  if (epilogue == CFHIP_EPI_DQGELU) epilogue = CFHIP_EPI_DGELU;
  CFHIP_REQUIRE(!(epilogue == CFHIP_EPI_RESIDUAL || epilogue == CFHIP_EPI_DGELU) || aux_in,
                "gemm: epilogue %d needs aux_in", epilogue);
  CFHIP_REQUIRE(!accumulate || (out_dtype == 1 && epilogue == CFHIP_EPI_NONE), "gemm: accumulate needs f32 output and epilogue NONE");
  CFHIP_REQUIRE(!(epilogue == CFHIP_EPI_GELU || epilogue == CFHIP_EPI_DGELU) || out_dtype == 0,
                "gemm: the GELU / GELU' epilogues write bf16");
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
  p.tiles_m = p.tiles_n = 0;
  p.quick = quick;
#ifdef CFHIP_ABLATE
  p.ablate = g_gemm_ablate;
#endif
  p.bgrad = bias_grad;
  p.bgrad_acc = bias_grad_accumulate;
  p.bgrad_slabs = nullptr;
  p.conv_h = p.conv_w = p.conv_c = p.conv_kpt = 0;
  p.conv_inv_kpt = 0.f;
  p.conv_magic_w = p.conv_magic_h = 0u;
  p.k_chunk = ((K + BK_MAX - 1) / BK_MAX) * BK_MAX;

  // K % 8 is an alignment rule of the k-major operands only (16-byte chunks along k); in the (1,1) layout k is
  // the row index of both operands and any tail is zero-filled by the DMA's k-range check
  const bool k_ok = (a_trans && b_trans) || (K % 8 == 0);
  const bool fast = k_ok && (lda % 8 == 0) && (ldb % 8 == 0) && (N % 4 == 0) &&
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
    const int steps = (K + BK_MAX - 1) / BK_MAX;
    if (split_k > steps) split_k = steps;
    const int per = (steps + split_k - 1) / split_k;
    split_k = (steps + per - 1) / per;
    p.k_chunk = per * BK_MAX;
  }
  if (split_k > 1) {
    const size_t need = (size_t)split_k * M * N * sizeof(float) + (bias_grad ? (size_t)split_k * M * sizeof(float) : 0);
    if (workspace == nullptr || workspace_bytes < need) {
      cfhip_set_error("gemm: split_k=%d needs %zu workspace bytes, got %zu", split_k, need, workspace_bytes);
      return CFHIP_ERR_WORKSPACE;
    }
    p.slabs = reinterpret_cast<float*>(workspace);
    if (bias_grad) p.bgrad_slabs = p.slabs + (size_t)split_k * M * N;
  }

  int rc;
  switch (pick_config(M, N, a_trans, b_trans)) {
    case 1: rc = launch_layout<CfgB>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 2: rc = launch_layout<CfgC>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 3: rc = launch_layout<CfgD>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 4: rc = launch_layout<CfgE>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 5: rc = launch_layout<CfgF>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 6: rc = launch_layout<CfgG>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 7: rc = launch_layout<CfgP, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 8: rc = launch_layout<CfgQ, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 9: rc = launch_layout<CfgR, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 10: rc = launch_layout<CfgS, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 11: rc = launch_layout<CfgT, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 12: rc = launch_layout<CfgU, true>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 13: rc = launch_layout<CfgW>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 14: rc = launch_layout<CfgY>(p, a_trans, b_trans, epilogue, split_k, s); break;
    default: rc = launch_layout<CfgA>(p, a_trans, b_trans, epilogue, split_k, s); break;
  }
  if (rc != CFHIP_OK) return rc;
  CFHIP_CHECK_LAUNCH("gemm_bf16");

  if (split_k > 1) {
    const long total = (long)M * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, C, bias, M, N,
                       (long)ldc, split_k, out_dtype, accumulate, p.bgrad_slabs, p.bgrad, p.bgrad_acc);
    CFHIP_CHECK_LAUNCH("splitk_reduce");
  }
  return CFHIP_OK;
}

// ---- implicit-GEMM 3x3 convolution ------------------------------------------------------------------
template <class C, bool PIPE>
static int launch_conv(GemmParams p, int split_k, hipStream_t s) {
  p.tiles_m = (p.M + C::BM - 1) / C::BM;
  p.tiles_n = (p.N + C::BN - 1) / C::BN;
  p.splits = split_k;
  return launch_cfg<false, false, CFHIP_EPI_NONE, C, PIPE, 1>(p, dim3(p.tiles_m * p.tiles_n * split_k), s);
}

// The UNet's deeper levels have few output tiles (32^2 x 8 x 640: 160 tiles of 256x128, 8^2 x 8 x 1280: 40 of 128x128)
// under a deep reduction (K = 9 * Cin = 5 760 .. 23 040): split K until the grid covers the resident slots.
static int conv_pick_split(long pixels, int Cin, int Cout) {
  const bool phase = pixels >= 1024;
  const int bm = phase ? 256 : 128;
  const long tiles = ((pixels + bm - 1) / bm) * ((Cout + 127) / 128);
  const long slots = phase ? 512 : 1024;
  const int steps = (9 * Cin + BK_MAX - 1) / BK_MAX;
  if (tiles * 2 > slots || steps < 16) return 1;
  long split = (slots + tiles - 1) / tiles;
  if (split > steps / 8) split = steps / 8;
  if (split < 1) split = 1;
  const int per = (int)((steps + split - 1) / split);
  return (steps + per - 1) / per;
}

extern "C" size_t cfhip_conv3x3_workspace(int B, int H, int W, int Cin, int Cout) {
  const long pixels = (long)B * H * W;
  const int split = conv_pick_split(pixels, Cin, Cout);
  return split > 1 ? (size_t)split * pixels * Cout * sizeof(float) : 0;
}

extern "C" int cfhip_conv3x3_nhwc_bf16(const void* X, const void* Wk, const float* bias, void* Y, int B, int H, int W,
                                       int Cin, int Cout, void* workspace, size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(X && Wk && Y, "conv3x3: null operand");
  CFHIP_REQUIRE(B > 0 && H > 0 && W > 0 && H < 65536 && W < 65536, "conv3x3: bad image shape %d x %d x %d", B, H, W);
  CFHIP_REQUIRE(Cin > 0 && Cin % 32 == 0, "conv3x3: Cin = %d must be a multiple of 32 (one K-step = 32 channels of a tap)", Cin);
  CFHIP_REQUIRE(Cout > 0 && Cout % 8 == 0, "conv3x3: Cout = %d must be a multiple of 8", Cout);
  const long pixels = (long)B * H * W;
  CFHIP_REQUIRE(pixels < (1L << 30) && 9L * Cin < (1L << 20), "conv3x3: problem too large (%ld pixels, Cin %d)", pixels, Cin);
  CFHIP_REQUIRE(aligned16(X) && aligned16(Wk) && aligned16(Y) && (bias == nullptr || aligned16(bias)),
                "conv3x3: operands must be 16-byte aligned");
  GemmParams p;
  p.A = reinterpret_cast<const bf16_t*>(X);
  p.B = reinterpret_cast<const bf16_t*>(Wk);
  p.C = Y;
  p.bias = bias;
  p.aux_in = nullptr;
  p.aux_out = nullptr;
  p.M = (int)pixels; p.N = Cout; p.K = 9 * Cin;
  p.lda = Cin; p.ldb = 9L * Cin; p.ldc = Cout;
  p.epilogue = CFHIP_EPI_NONE; p.out_f32 = 0; p.accumulate = 0;
  p.k_chunk = ((p.K + BK_MAX - 1) / BK_MAX) * BK_MAX;
  p.slabs = nullptr;
  p.tiles_m = p.tiles_n = 0; p.splits = 1;
  p.bgrad = nullptr; p.bgrad_slabs = nullptr; p.bgrad_acc = 0;
  p.quick = 0;
#ifdef CFHIP_ABLATE
  p.ablate = 0;
#endif
  p.conv_h = H; p.conv_w = W; p.conv_c = Cin; p.conv_kpt = Cin / 32;
  p.conv_inv_kpt = 1.0f / (float)p.conv_kpt;
  p.conv_magic_w = p.conv_magic_h = 0u;
  const int split = conv_pick_split(pixels, Cin, Cout);
  if (split > 1) {
    const size_t need = (size_t)split * pixels * Cout * sizeof(float);
    if (workspace == nullptr || workspace_bytes < need) {
      cfhip_set_error("conv3x3: split_k=%d needs %zu workspace bytes (cfhip_conv3x3_workspace), got %zu", split, need, workspace_bytes);
      return CFHIP_ERR_WORKSPACE;
    }
    const int steps = (p.K + BK_MAX - 1) / BK_MAX;
    p.k_chunk = ((steps + split - 1) / split) * BK_MAX;
    p.slabs = reinterpret_cast<float*>(workspace);
  }
  hipStream_t s = (hipStream_t)stream;
  const int rc = p.M >= 1024 ? launch_conv<CfgQ, true>(p, split, s) : launch_conv<CfgB, false>(p, split, s);
  if (rc != CFHIP_OK) return rc;
  CFHIP_CHECK_LAUNCH("conv3x3_nhwc");
  if (split > 1) {
    const long total = (long)p.M * (p.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, Y, bias, p.M, p.N, (long)p.ldc, split,
                       0, 0, (const float*)nullptr, (float*)nullptr, 0);
    CFHIP_CHECK_LAUNCH("splitk_reduce");
  }
  return CFHIP_OK;
}

extern "C" size_t cfhip_conv3x3_wgrad_workspace(int Cin, int Cout, int split_k) {
  if (split_k < 1) split_k = 1;
  return (size_t)split_k * Cout * (9 * (size_t)Cin + 1) * sizeof(float);
}

extern "C" int cfhip_conv3x3_wgrad_nhwc_bf16(const void* dY, const void* X, float* dW, int accumulate, float* bias_grad,
                                             int bias_grad_accumulate, int B, int H, int W, int Cin, int Cout,
                                             int split_k, void* workspace, size_t workspace_bytes, void* stream) {
  CFHIP_REQUIRE(dY && X && dW, "conv3x3_wgrad: null operand");
  CFHIP_REQUIRE(B > 0 && H >= 2 && W >= 2 && H < 65536 && W < 65536, "conv3x3_wgrad: bad image shape %d x %d x %d", B, H, W);
  CFHIP_REQUIRE(Cin > 0 && Cin % 8 == 0 && Cout > 0 && Cout % 8 == 0, "conv3x3_wgrad: Cin = %d and Cout = %d must be multiples of 8", Cin, Cout);
  const long pixels = (long)B * H * W;
  CFHIP_REQUIRE(pixels * (W > H ? W : H) < (1L << 32) && 9L * Cin < (1L << 20),
                "conv3x3_wgrad: problem too large (%ld pixels of %d x %d, Cin %d)", pixels, H, W, Cin);
  CFHIP_REQUIRE(aligned16(dY) && aligned16(X) && aligned16(dW), "conv3x3_wgrad: operands must be 16-byte aligned");
  if (split_k < 1) split_k = 1;
  GemmParams p;
  p.A = reinterpret_cast<const bf16_t*>(dY);
  p.B = reinterpret_cast<const bf16_t*>(X);
  p.C = nullptr;  // partial sums always go through the slabs; the second pass writes the filter layout
  p.bias = nullptr; p.aux_in = nullptr; p.aux_out = nullptr;
  p.M = Cout; p.N = 9 * Cin; p.K = (int)pixels;
  p.lda = Cout; p.ldb = Cin; p.ldc = 9L * Cin;
  p.epilogue = CFHIP_EPI_NONE; p.out_f32 = 1; p.accumulate = 0;
  p.bgrad = bias_grad; p.bgrad_acc = bias_grad_accumulate; p.bgrad_slabs = nullptr;
  p.quick = 0;
#ifdef CFHIP_ABLATE
  p.ablate = 0;
#endif
  p.conv_h = H; p.conv_w = W; p.conv_c = Cin; p.conv_kpt = 0; p.conv_inv_kpt = 0.f;
  p.conv_magic_w = (unsigned)((1ULL << 32) / (unsigned)W + 1ULL);
  p.conv_magic_h = (unsigned)((1ULL << 32) / (unsigned)H + 1ULL);
  const int steps = (p.K + BK_MAX - 1) / BK_MAX;
  if (split_k > steps) split_k = steps;
  const int per = (steps + split_k - 1) / split_k;
  split_k = (steps + per - 1) / per;
  p.k_chunk = per * BK_MAX;
  CFHIP_REQUIRE(((long)p.k_chunk + 2L * W + 2) * Cin * 2 < 0x7fffffffL && (long)p.k_chunk * Cout * 2 < 0x7fffffffL,
                "conv3x3_wgrad: a K slice of %d pixels exceeds the 2 GiB descriptor range (raise split_k)", p.k_chunk);
  const size_t need = (size_t)split_k * p.M * p.N * sizeof(float) + (bias_grad ? (size_t)split_k * p.M * sizeof(float) : 0);
  if (workspace == nullptr || workspace_bytes < need) {
    cfhip_set_error("conv3x3_wgrad: split_k=%d needs %zu workspace bytes, got %zu", split_k, need, workspace_bytes);
    return CFHIP_ERR_WORKSPACE;
  }
  p.slabs = reinterpret_cast<float*>(workspace);
  if (bias_grad) p.bgrad_slabs = p.slabs + (size_t)split_k * p.M * p.N;
  p.tiles_m = (p.M + CfgC::BM - 1) / CfgC::BM;
  p.tiles_n = (p.N + CfgC::BN - 1) / CfgC::BN;
  p.splits = split_k;
  hipStream_t s = (hipStream_t)stream;
  const int rc = launch_cfg<true, true, CFHIP_EPI_NONE, CfgC, false, 2>(p, dim3(p.tiles_m * p.tiles_n * split_k), s);
  if (rc != CFHIP_OK) return rc;
  CFHIP_CHECK_LAUNCH("conv3x3_wgrad");
  const long total = (long)Cout * (Cin / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_filter_kernel, dim3(blocks), dim3(256), 0, s, p.slabs, dW, Cout, Cin, split_k, accumulate,
                     p.bgrad_slabs, p.bgrad, p.bgrad_acc);
  CFHIP_CHECK_LAUNCH("splitk_reduce_filter");
  return CFHIP_OK;
}
