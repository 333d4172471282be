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
#include "gemm_device.h"

namespace {

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
    static_assert(CONV != 2 || (AT && BT), "implicit weight gradient: layout (1,1), a K-step = BK pixels");
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
    c.pb = make_plan<BT, C::BN, C::B_INSTR, C::BK>(wave, lane, p.ldb, C::B_PADDED ? min(rows_b, C::BN) : rows_b);
  }
  if constexpr (CONV == 1) {
    static_assert(!AT, "implicit convolution: k-major A, a K-step = BK channels of one tap");
    // The descriptor starts W+1 pixels BEFORE the tile (clamped at pixel 0) so that the (-1,-1) tap is a
    // non-negative offset; it ends W+1 pixels after it.  Taps outside the image are sent out of range per lane.
    const int back = min(c.m0, p.conv_w + 1);
    const long span = min((long)p.M - (c.m0 - back), (long)back + C::BM + p.conv_w + 1);
    c.a_rsrc = make_rsrc(p.A + (long)(c.m0 - back) * p.conv_c, span * p.conv_c * 2);
#pragma unroll
    for (int j = 0; j < C::A_INSTR; ++j) {
      const int inst = wave * C::A_INSTR + j;
      constexpr int SPR = C::BK / 8;  // 16-byte slots per tile row
      const int row = inst * (64 / SPR) + lane / SPR;
      const int chunk = (lane % SPR) ^ kswz<C::BK>(row);
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

// CONV: K-step `kstep` of the A operand = BK channels (c0..) of tap (ky, kx) for the tile's BM pixels
template <class C>
__device__ __forceinline__ void stage_tile_conv(const __amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave,
                                                const StagePlan<C::A_INSTR>& pl, const unsigned (&yx)[C::A_INSTR],
                                                const GemmParams& p, int kstep) {
  const int tap = (int)(((float)kstep + 0.5f) * p.conv_inv_kpt);  // kstep / conv_kpt (exact: kstep < 2^16)
  const int c0 = (kstep - tap * p.conv_kpt) * C::BK;
  const int ky = (tap * 11) >> 5;  // tap / 3 for tap in 0..8
  const int dy = ky - 1, dx = tap - 3 * ky - 1;
  const int shift = ((dy * p.conv_w + dx) * p.conv_c + c0) * 2;
#pragma unroll
  for (int j = 0; j < C::A_INSTR; ++j) {
    const int y = (int)(yx[j] >> 16) + dy, x = (int)(yx[j] & 0xffffu) + dx;
    const bool ok = pl.voff[j] != OOB && (unsigned)y < (unsigned)p.conv_h && (unsigned)x < (unsigned)p.conv_w;
    const unsigned off = ok ? pl.voff[j] + (unsigned)shift : OOB;
    lds_dma16(rsrc, lds_tile + (wave * C::A_INSTR + j) * 1024, off);
  }
}

// CONV 2: K-step `kstep` of the B operand = BK pixels x the tile's BN (tap, channel) columns of the shifted activation
template <class C, class Ctx>
__device__ __forceinline__ void stage_tile_wgrad(const Ctx& c, char* lds_tile, int wave, const GemmParams& p, int kstep) {
  const int k0 = kstep * C::BK;
#pragma unroll
  for (int j = 0; j < C::B_INSTR; ++j) {
    const int kl = k0 + (int)c.pb.kpos[j];
    const unsigned pix = (unsigned)(c.kb + kl);
    const unsigned q = __umulhi(pix, p.conv_magic_w);                   // pix / W
    const int x = (int)(pix - q * (unsigned)p.conv_w) + ((c.btap[j] >> 2) - 1);
    const int y = (int)(q - __umulhi(q, p.conv_magic_h) * (unsigned)p.conv_h) + ((c.btap[j] & 3) - 1);
    const bool ok = c.pb.voff[j] != OOB && kl < c.klen && (unsigned)y < (unsigned)p.conv_h && (unsigned)x < (unsigned)p.conv_w;
    const unsigned off = ok ? (unsigned)(c.bshift[j] + k0 * p.conv_c * 2) : OOB;
    lds_dma16(c.b_rsrc, lds_tile + (wave * C::B_INSTR + j) * 1024, off);
  }
}

template <bool AT, bool BT, class C, int CONV = 0>
__device__ __forceinline__ void stage_step(const ItemCtx<AT, BT, C>& c, const GemmParams& p, char* smem,
                                           int slot, int wave, int kstep) {
  char* st = smem + slot * C::STAGE_BYTES;
  if constexpr (CONV == 1) stage_tile_conv<C>(c.a_rsrc, st, wave, c.pa, c.yx, p, kstep + c.kb / C::BK);
  else stage_tile<AT, C::A_INSTR, C::BK>(c.a_rsrc, st, wave, c.pa, p.lda, kstep * C::BK, c.klen);
  if constexpr (CONV == 2) {
    stage_tile_wgrad<C>(c, st + C::A_BYTES, wave, p, kstep);
    return;
  }
  stage_tile<BT, C::B_INSTR, C::BK>(c.b_rsrc, st + C::A_BYTES, wave, c.pb, p.ldb, kstep * C::BK, c.klen);
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
// XCD-aware, bijective remap of the grid: XCD x (= bid % 8) owns a contiguous range of work items (neighbours share A / B
// panels in its L2).
__device__ __forceinline__ int xcd_item() {
  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
}

// One work item of problem `p` on this workgroup (the body of gemm_bf16_kernel).
template <bool AT, bool BT, int EPI, class C, int CONV = 0, bool BG = false>
__device__ __forceinline__ void gemm_tile(const GemmParams& p, const int item, char* smem) {
  constexpr int D = C::NSTAGE - 1;  // prefetch distance
  static_assert(kFlatEpilogue<C> ? C::NW * 16 * (C::FN * 16 + 4) * 4 <= C::LDS_BYTES : C::NW * 16 * (C::FN * 16) * 4 <= C::STAGE_BYTES,
                "epilogue staging must fit in one ring slot (80-column wave tiles: in the ring)");

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
  const bool do_bg = BG && cur.tile_n == 0 && wn == 0 && (p.bgrad != nullptr || p.bgrad_slabs != nullptr);

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
  // (every wave is past its last compute_tile at that barrier: the WHOLE ring is free — the 80-column tiles' strips start at slot 0)
  epilogue<EPI, C>(p, acc, kFlatEpilogue<C> ? smem : smem + eslot * C::STAGE_BYTES, m0, n0, z, wm, wn, wave, lane);
}

template <bool AT, bool BT, int EPI, class C, int CONV = 0, bool BG = false>
__global__ __launch_bounds__(C::NT, C::WAVES_PER_SIMD)
void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_tile<AT, BT, EPI, C, CONV, BG>(p, xcd_item(), smem);
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
  static_assert(C::NW * 16 * (C::FN * 16 + 4) * 4 <= C::LDS_BYTES, "epilogue staging must fit in the (free) ring");
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

// split-K second pass: C = sum_z slab[z] (+ bias) (+ C).  (Round 6: non-temporal slab loads / gradient stores — what helps the optimizer
// kernel, elementwise.hip — cost the UNet steps 3-4 ms, and the stores alone as much: a range's gradients are read by its optimizer launch
// inside the same backward pass, out of L2 / MALL unless the hint sent them past it; profiles/r06/adam_nontemporal_ab.txt.)
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
using CfgZ = Cfg<192, 128, 2, 2, 2, 64>;  //  80 KiB LDS, 4 waves (96x64 each: 2.4 MFMAs per fragment read against 1.5), 2 WG / CU
using CfgV = Cfg<128, 256, 2, 4, 2, 64>;  //  96 KiB LDS, 8 waves (64x64 each), 1 WG / CU
// The UNet's convolutions (N = 320 .. 2560: multiples of 160, of no power-of-two tile width; M = 65 536 .. 512 pixels): four waves of
// 64 x 80 (20 MFMAs per 9 fragment reads; the 128x32 waves of CfgQ: 16 per 10).  Convolution launcher only (conv_plan), not in the
// gemm_config table.  Round 6, tools/conv_bench.py over the 26 (pixels, Cin, Cout) of the zoo UNet at 64^2 x 8, launches x isolated time:
// 9.45 ms with the round-3 table (CfgQ / CfgB), 8.08 with CfgH, 6.41 with CfgI (profiles/r06/conv_forms_bench.txt): the 64-channel
// K-step is worth more than the tile shape.  (Measured and not kept: 128x320 on the eight-wave phase kernel 8.17, CfgH on the phase
// kernel 8.17, a two-stage 128x160x32 at 4 WG / CU 9.05.)
using CfgH = Cfg<128, 160, 2, 2, 4, 32>;  //  80 KiB LDS (B image padded to 192 rows), 4 waves, 2 WG / CU, prefetch 3
using CfgI = Cfg<128, 160, 2, 2, 2, 64>;  //  72 KiB LDS, 4 waves, 2 WG / CU, prefetch 1, 64-channel K-steps (Cin % 64 == 0)
constexpr int NUM_CFG = 17;  // (17 .. 20 were the software-pipelined 32x32x16 kernels of round 4: measured, never selected, removed in round 6 — profiles/r04/gemm_pp_*.log, docs/DESIGN_HISTORY.md)  7 .. 12 = CfgP / CfgQ / CfgR / CfgS / CfgT / CfgU on the phase kernel; 13 / 14 = CfgW / CfgY; 15 / 16 = CfgZ / CfgV
constexpr int BK_MAX = 64;

int g_gemm_config = -1;
#ifdef CFHIP_ABLATE
int g_gemm_ablate = 0;
#endif
int g_gemm_heuristic = 9;  // round 5: 9 (four-wave 192x128x64 for every M >= 1024 forward / dX GEMM): ViT 17.61 -> 17.47 ms, CLIP 17.89 -> 17.60, UNet neutral (profiles/r05/heuristic9_ab.txt)
int g_gemm_group_n = 8;
int g_conv_form = -1;   // "conv_form" option: -1 = conv_plan's choice
int g_conv_split = -1;  // "conv_split" option: -1 = conv_plan's choice (tools/conv_bench.py sweeps both)

template <bool AT, bool BT, int EPI, class C, bool PIPE, int CONV = 0>
int launch_cfg(const GemmParams& p, dim3 grid, hipStream_t s) {
  void (*kern)(GemmParams) = nullptr;
  if (PIPE && p.bgrad != nullptr) {
    cfhip_set_error("gemm: the fused bias gradient is not provided by the big-tile kernel (gemm_config 7 / 8)");
    return CFHIP_ERR_INVALID;
  }
  if constexpr (PIPE) kern = gemm_bf16_phase_kernel<AT, BT, EPI, C, CONV>;
  else if constexpr (AT && BT && EPI == CFHIP_EPI_NONE) {
    if (p.bgrad != nullptr) {
      // (the bias-gradient rows of a 256-row tile do not fit beside its accumulators: 52 bytes of scratch per lane — such
      // requests are routed to configuration 1 in cfhip_gemm_bf16, the instantiation does not exist)
      if constexpr (C::BM * C::BN <= 192 * 128) kern = gemm_bf16_kernel<AT, BT, EPI, C, CONV, true>;
      else {
        cfhip_set_error("gemm: the fused bias gradient is not provided by this tile configuration");
        return CFHIP_ERR_INVALID;
      }
    } else kern = gemm_bf16_kernel<AT, BT, EPI, C, CONV, false>;
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

// instantiations that spilled and are not needed: the residual / GELU' epilogues on 128x128x32 (configuration 1: 12 / 40 bytes per
// lane — cfhip_gemm_bf16 sends those requests to 128x64x64), the weight-gradient layout on the 256x128x32 phase kernel (12 bytes)
template <class C, bool PIPE>
constexpr bool kHasEpilogues = !(C::BM == 128 && C::BN == 128 && C::BK == 32 && C::NSTAGE == 2);
template <class C, bool PIPE>
constexpr bool kHasTn = !(PIPE && C::BM == 256 && C::BN == 128 && C::WN == 4 && C::NSTAGE == 3);

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
      case CFHIP_EPI_RESIDUAL:
        if constexpr (kHasEpilogues<C, PIPE>) return launch_cfg<false, false, CFHIP_EPI_RESIDUAL, C, PIPE>(p, grid, s);
        break;
      default: break;
    }
  } else if (!a_trans && b_trans) {
    switch (epilogue) {
      case CFHIP_EPI_NONE: return launch_cfg<false, true, CFHIP_EPI_NONE, C, PIPE>(p, grid, s);
      case CFHIP_EPI_DGELU:
        if constexpr (kHasEpilogues<C, PIPE>) return launch_cfg<false, true, CFHIP_EPI_DGELU, C, PIPE>(p, grid, s);
        break;
      default: break;
    }
  } else if (epilogue == CFHIP_EPI_NONE) {
    if constexpr (kHasTn<C, PIPE>) return launch_cfg<true, true, CFHIP_EPI_NONE, C, PIPE>(p, grid, s);
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
int g_cfg_class[4] = {-1, -1, -1, -1};  // per-class override (step A/Bs): nt with N >= 2560, other nt, nn, tn; M >= 1024 only

int pick_config(int M, int N, int a_trans, int b_trans) {
  if (g_gemm_config >= 0 && g_gemm_config < NUM_CFG) return g_gemm_config;
  if (M >= 1024 || a_trans) {
    const int cls = a_trans ? 3 : b_trans ? 2 : (N >= 2560 ? 0 : 1);
    if (g_cfg_class[cls] >= 0 && g_cfg_class[cls] < NUM_CFG) return g_cfg_class[cls];
  }
  if (g_gemm_heuristic >= 7) {
    // round 3 (LDS-DMA as inline asm: the rings really prefetch now; weight gradients of the block stack in grouped launches
    // on their own stream; backward in two batch slices): 192x128x64 for every big forward AND dX GEMM — whole-step A/B
    // 19.06 -> 18.35 ms against table 6 (profiles/r03/step_variants_a.log)
    if (a_trans) return 1;
    // round 3b, the UNet's shapes (tools/gemm_unet_sweep.py, profiles/r03/gemm_unet_sweep.log): 128x64x64 when 192x128 tiles
    // would leave most CUs without a workgroup (2048 x 1280: 110 tiles, 18.8 -> 14.9 us) and for narrow outputs that 128
    // columns do not divide (N = 320: 17.7 / 18.8 -> 15.4 / 13.7 us)
    if (M >= 512) {
      const long t14 = (long)((M + 191) / 192) * ((N + 127) / 128);
      if (t14 < 160 || (N < 512 && N % 128 != 0 && N % 64 == 0)) return 3;
    }
    // round 3c: the same 192x128x64 tile on FOUR waves of 96x64 for outputs up to 1 024 columns wide (2.4 MFMAs per
    // fragment read against 1.5 with eight 96x32 waves): N = 768 shapes 5-10 % faster alone (K = 2 304 dX 898 -> 992 TFLOP/s),
    // wider outputs lose (their waves' epilogues are twice as long); profiles/r03/gemm_bench_b128_c15.log
    // round 5 (table 9): on the final round-4 kernels the four-wave form is ahead on EVERY step shape alone, the wide GELU / GELU'
    // outputs included (80.7 / 81.8 us against 99.4 / 105.4, QKV 49.3 against 56.5: profiles/r05/gemm_bench_wide_tiles.log);
    // in the step the gain is what a CU-time-bound step leaves of it (17.515 -> 17.443 ms, profiles/r05/step_variants_four_wave.log)
    if (M >= 1024) return (g_gemm_heuristic >= 9 || (g_gemm_heuristic >= 8 && N <= 1024)) ? 15 : 14;
    if (b_trans) return 1;
    return N <= 1024 ? 3 : 0;
  }
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
int cfhip_internal_set_attn_one_pass(int v);  // attn.hip
int cfhip_internal_set_attn_pers_ctas(int v);  // attn.hip
int cfhip_internal_set_grouped_variant(int v);  // gemm_grouped.hip
int cfhip_internal_set_attn_two_tiles(int v);  // attn.hip
int cfhip_internal_set_attn_short_max(int v);  // attn.hip

extern "C" int cfhip_set_option(const char* name, int value) {
  if (name != nullptr && strcmp(name, "gemm_config") == 0) {
    g_gemm_config = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "gemm_heuristic") == 0) {
    g_gemm_heuristic = value;
    return CFHIP_OK;
  }
  static const char* const cls_names[4] = {"gemm_cfg_nt_wide", "gemm_cfg_nt", "gemm_cfg_nn", "gemm_cfg_tn"};
  for (int i = 0; i < 4; ++i)
    if (name != nullptr && strcmp(name, cls_names[i]) == 0) {
      g_cfg_class[i] = value;
      return CFHIP_OK;
    }
  if (name != nullptr && strcmp(name, "gemm_group_n") == 0) {
    g_gemm_group_n = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "conv_form") == 0) {
    g_conv_form = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "conv_split") == 0) {
    g_conv_split = value;
    return CFHIP_OK;
  }
  if (name != nullptr && strcmp(name, "ln_bwd_fused") == 0) return cfhip_internal_set_ln_fused(value);
  if (name != nullptr && strcmp(name, "attn_persistent") == 0) return cfhip_internal_set_attn_persistent(value);
  if (name != nullptr && strcmp(name, "attn_one_pass") == 0) return cfhip_internal_set_attn_one_pass(value);
  if (name != nullptr && strcmp(name, "attn_pers_ctas") == 0) return cfhip_internal_set_attn_pers_ctas(value);
  if (name != nullptr && strcmp(name, "grouped_variant") == 0) return cfhip_internal_set_grouped_variant(value);
  if (name != nullptr && strcmp(name, "attn_two_tiles") == 0) return cfhip_internal_set_attn_two_tiles(value);
  if (name != nullptr && strcmp(name, "attn_short_max") == 0) return cfhip_internal_set_attn_short_max(value);
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

// Requests whose instantiation does not exist (kHasEpilogues / kHasTn / the bias-gradient note in launch_cfg) run on a clean
// sibling.  ONE rule for the launcher and for cfhip_gemm_kernel_name.  The fused bias gradient exists on tiles up to 192x128
// of the plain kernel only: every phase-kernel configuration (7 .. 12) and the 256x256 / 128x256 plain tiles (13, 16) go to 1
// (round 5, ADVICE r4: 9 / 10 / 11 used to return CFHIP_ERR_INVALID under a forced gemm_config).
static int resolve_config(int cfg, int a_trans, int epilogue, bool bias_grad) {
  if (cfg == 8 && a_trans) cfg = 1;
  if (bias_grad && ((cfg >= 7 && cfg <= 13) || cfg == 16)) cfg = 1;
  // LAST (round 6): CfgB (128x128x32 on two stages) has no RESIDUAL / DGELU epilogue; a forced big-tile configuration
  // with a fused bias gradient used to arrive here AFTER this rule had been applied and ended in CFHIP_ERR_INVALID.  CfgD
  // (128x64x64) has both the epilogues and the bias gradient (kHasEpilogues, kHasBiasGrad in launch_cfg).
  if (cfg == 1 && (epilogue == CFHIP_EPI_RESIDUAL || epilogue == CFHIP_EPI_DGELU)) cfg = 3;
  return cfg;
}

// The kernel (template instantiation, as rocprofv3 prints it) that cfhip_gemm_bf16 launches for a fast-path problem: bench.py
// groups its in-step GEMM timings by it, so that its "dominant kernel" is the row a rocprofv3 --stats summary lists first.
extern "C" int cfhip_gemm_kernel_name(int M, int N, int K, int a_trans, int b_trans, int epilogue, char* out, size_t out_bytes) {
  CFHIP_REQUIRE(out != nullptr && out_bytes >= 96, "gemm_kernel_name: buffer of at least 96 bytes");
  static const char* const cfg_names[NUM_CFG] = {
      "Cfg<128, 128, 2, 2, 2, 64>", "Cfg<128, 128, 2, 2, 2, 32>", "Cfg<128, 128, 2, 2, 3, 32>", "Cfg<128, 64, 2, 2, 2, 64>",
      "Cfg<128, 128, 2, 2, 4, 32>", "Cfg<128, 64, 2, 2, 2, 32>", "Cfg<128, 64, 2, 2, 3, 32>", "Cfg<256, 256, 2, 4, 4, 32>",
      "Cfg<256, 128, 2, 4, 3, 32>", "Cfg<256, 128, 2, 2, 3, 32>", "Cfg<128, 256, 2, 2, 3, 32>", "Cfg<256, 128, 2, 4, 6, 32>",
      "Cfg<256, 256, 2, 4, 5, 32>", "Cfg<256, 256, 2, 4, 2, 64>", "Cfg<192, 128, 2, 4, 2, 64>", "Cfg<192, 128, 2, 2, 2, 64>",
      "Cfg<128, 256, 2, 4, 2, 64>"};
  int epi = epilogue == CFHIP_EPI_QGELU ? CFHIP_EPI_GELU : epilogue == CFHIP_EPI_DQGELU ? CFHIP_EPI_DGELU : epilogue;
  int cfg = pick_config(M, N, a_trans, b_trans);
  cfg = resolve_config(cfg, a_trans, epi, false);  // the reroutes of instantiations that do not exist: the name is the kernel that RUNS
  const bool phase = cfg >= 7 && cfg <= 12;
  snprintf(out, out_bytes, "%s<%s, %s, %d, %s>", phase ? "gemm_bf16_phase_kernel" : "gemm_bf16_kernel",
           a_trans ? "true" : "false", b_trans ? "true" : "false", epi, cfg_names[cfg]);
  return CFHIP_OK;
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
  int cfg = pick_config(M, N, a_trans, b_trans);
  cfg = resolve_config(cfg, a_trans, epilogue, bias_grad != nullptr);
  switch (cfg) {
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
    case 15: rc = launch_layout<CfgZ>(p, a_trans, b_trans, epilogue, split_k, s); break;
    case 16: rc = launch_layout<CfgV>(p, a_trans, b_trans, epilogue, split_k, s); break;
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

// Tile form and K split of one convolution.  Forms: 0 = 256x128x32 phase kernel (CfgQ, 2 WG / CU), 1 = 128x128x32 plain (CfgB, 4 WG / CU)
// — the round-3 table: form 0 from 1 024 pixels on; 2 = 128x160x32 on four waves (CfgH, 2 WG / CU), 3 = 128x160x64 (CfgI, 2 WG / CU).
// Round 6: outputs that 160-column tiles cover with at most 1/8 of waste (the UNet: none) take form 3 when Cin % 64 == 0, else form 2.
// The UNet's deeper levels have few output tiles (32^2 x 8 x 640: 256 tiles of 128x160, 8^2 x 8 x 1280: 32)
// under a deep reduction (K = 9 * Cin = 5 760 .. 23 040): split K until the grid covers the resident slots.
struct ConvPlan { int form, split; };

static int conv_split_for(long tiles, long slots, int Cin) {
  const int steps = (9 * Cin + BK_MAX - 1) / BK_MAX;
  if (tiles * 2 > slots || steps < 16) return 1;
  long split = (slots + tiles - 1) / tiles;
  if (split > steps / 8) split = steps / 8;
  if (split < 1) split = 1;
  const int per = (int)((steps + split - 1) / split);
  return (steps + per - 1) / per;
}

static ConvPlan conv_plan(long pixels, int Cin, int Cout) {
  ConvPlan pl;
  pl.form = pixels >= 1024 ? 0 : 1;
  if (g_conv_form != -2 && (long)((Cout + 159) / 160) * 160 * 8 <= 9L * Cout) pl.form = 3;  // (-2: the round-3 table, A/B runs)
  if (g_conv_form >= 0 && g_conv_form <= 3) pl.form = g_conv_form;
  static const int bm[4] = {256, 128, 128, 128}, bn[4] = {128, 128, 160, 160}, slots[4] = {512, 1024, 512, 512};
  if (pl.form == 3 && Cin % 64 != 0) pl.form = 2;
  const long tiles = ((pixels + bm[pl.form] - 1) / bm[pl.form]) * ((Cout + bn[pl.form] - 1) / bn[pl.form]);
  pl.split = conv_split_for(tiles, slots[pl.form], Cin);
  if (g_conv_split >= 1) {
    const int steps = (9 * Cin + BK_MAX - 1) / BK_MAX;
    int split = g_conv_split > steps ? steps : g_conv_split;
    const int per = (steps + split - 1) / split;
    pl.split = (steps + per - 1) / per;
  }
  return pl;
}

extern "C" size_t cfhip_conv3x3_workspace(int B, int H, int W, int Cin, int Cout) {
  const long pixels = (long)B * H * W;
  const int split = conv_plan(pixels, Cin, Cout).split;
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
  const ConvPlan plan = conv_plan(pixels, Cin, Cout);
  const int split = plan.split;
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
  int rc;
  switch (plan.form) {
    case 0: rc = launch_conv<CfgQ, true>(p, split, s); break;
    case 2: rc = launch_conv<CfgH, false>(p, split, s); break;
    case 3:
      p.conv_kpt = Cin / 64;
      p.conv_inv_kpt = 1.0f / (float)p.conv_kpt;
      rc = launch_conv<CfgI, false>(p, split, s);
      break;
    default: rc = launch_conv<CfgB, false>(p, split, s); break;
  }
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
  p.splits = split_k;
  hipStream_t s = (hipStream_t)stream;
  // (round 6: the same launch on 128x128x64 / two stages — the 64-deep K-step that is worth 21 % on the forward / dX tiles — came out
  // level over the UNet's 16 filter-gradient shapes: 5.54 vs 5.57 ms per step alone, profiles/r06/conv_wgrad_bk64_not_kept.txt;
  // both operands are m-major here, every staged line is used whole at either depth)
  p.tiles_m = (p.M + CfgC::BM - 1) / CfgC::BM;
  p.tiles_n = (p.N + CfgC::BN - 1) / CfgC::BN;
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
