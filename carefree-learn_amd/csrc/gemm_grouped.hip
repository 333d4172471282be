// Grouped weight-gradient GEMM: up to 24 problems  dW_i[M_i][N_i] (+)= dY_i^T X_i  (and db_i (+)= colsum(dY_i)) in ONE launch.
//
// Replaces the parameter half of F.linear's autograd backward (reference modules/core/customs.py:89,
// attentions.py:214: grad_weight = grad_output^T @ input, grad_bias = grad_output.sum(0)) for ALL Linear layers of one or
// more transformer blocks at once.
//
// Why a group: one weight gradient of ViT-B/16 is 9 .. 36 tiles of 256 x 256 under a K = batch * tokens = 25 216 deep
// reduction — far fewer tiles than the 256 CUs.  The one-GEMM-per-launch path filled the chip by cutting K into slices
// (fp32 slabs + an ordered reduce launch per GEMM: 3 GB of slab traffic and 49 reduce launches per step).  Here the tiles of
// several GEMMs share the chip instead: the four weight gradients of TWO blocks are 216 tiles = one round on 256 CUs, every
// tile runs its whole reduction (788 K-steps of 32: prologue and store tail are noise), writes its output once, nothing is
// split, no slabs, no second pass, deterministic.
//
// Kernel = the two-group phase structure of gemm_bf16_phase_kernel (gemm.hip) on 256 x 256 x 32 tiles, 8 waves of 128 x 64,
// both operands m-major ([k][rows], rows contiguous: whole 512-byte lines per k-row whatever the K-step) read with
// ds_read_b64_tr_b16, with two changes:
//   * the LDS-DMA of K-step t + D is issued in the L (fragment-read) segments of K-step t — the A half in phase 0, the B half
//     in phase 1 — instead of in front of the MFMAs of an M segment: a wave in its L segment does not own the SIMD's matrix
//     pipe (its partner does), so the ~60-180 issue cycles of a DMA instruction no longer delay MFMAs;
//   * every K-step issues its DMA unconditionally (K-steps beyond the reduction are out-of-range offsets: zero fill, no
//     traffic), so every `s_waitcnt vmcnt` in the loop is the same compile-time count.
// The bias gradient rides on the matrix pipe: colsum(dY) = dY^T 1, i.e. one extra MFMA with a ones operand per A fragment,
// spread over the four column waves (one fragment per wave and phase).  Round 3b: the reduction of a bias gradient is SHARED by
// the tiles_n workgroups of a tile row — workgroup (tile_m, tile_n) sums the K-steps t = tile_n (mod tiles_n) — so that every
// workgroup of the launch runs the same instruction stream at the same pace.  When only the first tile column did it (every
// K-step), those workgroups ran ~8 % behind the others that share their operand panels, fell out of the L2 window (4 MiB =
// ~20 K-steps of an XCD's panels) and re-fetched both panels on their own: 2242 MB of L2 fills per launch against 1661 MB
// without bias gradients (1239 MB algorithmic; profiles/r03/grouped_dw_traffic.txt).  The tiles_n partial sums meet in a
// per-stream workspace; the LAST workgroup of a tile row to arrive (one atomic counter per tile row) adds them in tile_n
// order — deterministic, no float atomics — and writes db.
// Tile order inside a problem: the shorter tile dimension runs fastest, so that the contiguous tile range an XCD owns covers
// whole panels of the longer operand (fc2's dW is 3 x 12 tiles: row-major order made every XCD read all of X).
#include "gemm_device.h"
#include <mutex>
#include <type_traits>

namespace {

constexpr int GROUP_MAX = 24;  // problems per launch (88 bytes of kernel arguments each)

struct GroupedProblem {
  const bf16_t* A;  // dY  [K][M]  (m-major: element (m, k) at A[k * lda + m])
  const bf16_t* B;  // X   [K][N]
  float* C;         // dW  [M][N] f32
  float* bgrad;     // db  [M] f32 or nullptr
  int M, N, K;
  int lda, ldb, ldc;
  int tiles_m, tiles_n;
  int tile_end;  // tiles of problems 0 .. this one (exclusive prefix sum)
  int flags;     // bit 0: C += ; bit 1: bgrad += ; bit 2: tiles numbered with tile_m fastest
  int bg_parts;  // workgroups of a tile row that share the bias-gradient reduction (tiles_n, or 1: the first tile column alone)
  int bg_ws;     // float offset of this problem's [bg_parts][M] partial sums in GroupedParams::ws
  int bg_cnt;    // offset of its tiles_m arrival counters in GroupedParams::cnt
};

struct GroupedParams {
  GroupedProblem pr[GROUP_MAX];
  int count;
  float* ws;  // bias-gradient partial sums (per-stream workspace of the library)
  int* cnt;   // arrival counters, zero between launches
};

// instruction J (0 / 1) of an m-major operand's two-instruction K-step
template <int J>
__device__ __forceinline__ void stage_one(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, int wave, const StagePlan<2>& p, long ld,
                                          int k0, int klen) {
  unsigned off = p.voff[J] + (unsigned)((long)k0 * ld * 2);
  off = ((int)(k0 + p.kpos[J]) < klen && p.voff[J] != OOB) ? off : OOB;
  lds_dma16(rsrc, lds_tile + (wave * 2 + J) * 1024, off);
}

// Ring safety (by barrier count; intervals numbered as in gemm_bf16_phase_kernel: group 0 runs L(t, ph) in interval
// 4t + 2ph and M(t, ph) in 4t + 2ph + 1, group 1 one interval later):
//   WAR: the DMA of K-step t + D is issued in L(t, 0) / L(t, 1) (intervals >= 4t) into slot (t + D) % NSTAGE, which held
//        K-step t + D - NSTAGE <= t - 2 (D <= NSTAGE - 2): last read in group 1's L(t - 2, 1), interval 4t - 5, retired by
//        the lgkmcnt wait in front of its MFMAs in interval 4t - 4.
//   RAW: every wave waits for its own DMA of K-step t + 1 (vmcnt((D - 1) * LPS): K-steps t + 2 .. t + D stay in flight)
//        BEFORE the barrier that ends its L(t, 1) (intervals 4t + 2 / 4t + 3); the first read of K-step t + 1 is group 0's
//        L(t + 1, 0) in interval 4t + 4.
// PLACE: where the four DMA instructions of a K-step are issued.  0: A in L(t, 0), B in L(t, 1) (behind the fragment reads);
// 1: A inside M(t, 0), B inside M(t, 1), between the 8th and 9th MFMA; 2: one instruction per segment (L0, M0, L1, M1).
// The wait in L(t, 1) leaves (D - 2) * 4 + {4, 2, 3} instructions in flight: everything younger than K-step t + 1.
template <class C, int D, bool BG, int PLACE = 0>
__global__ __launch_bounds__(C::NT, 2)
void gemm_grouped_tn_kernel(GroupedParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(C::BM == 256 && C::BN == 256 && C::BK == 32 && C::WM == 2 && C::WN == 4, "grouped dW kernel: 256 x 256 x 32, 2 x 4 waves");
  static_assert(D >= 1 && D <= C::NSTAGE - 2, "DMA issued in L segments: prefetch distance <= NSTAGE - 2");
  static_assert(C::A_INSTR == 2 && C::B_INSTR == 2, "two DMA instructions per operand, wave and K-step");
  static_assert((D - 1) * C::LPS < 64, "vmcnt is a 6-bit counter");
  constexpr int HM = C::FM / 2;  // row fragments per phase (4)

  const int G = gridDim.x;
  const int bid = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  const int item = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // problem of this tile: constant indices only (a run-time index into the by-value table would go through scratch)
  GroupedProblem q = p.pr[0];
  int first = 0;
#pragma unroll
  for (int i = 1; i < GROUP_MAX; ++i) {
    if (i < p.count && item >= p.pr[i - 1].tile_end) {
      q = p.pr[i];
      first = p.pr[i - 1].tile_end;
    }
  }
  const int tile = item - first;
  int tile_m, tile_n;
  if (q.flags & 4) {
    tile_n = tile / q.tiles_m;
    tile_m = tile - tile_n * q.tiles_m;
  } else {
    tile_m = tile / q.tiles_n;
    tile_n = tile - tile_m * q.tiles_n;
  }
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
  const int rows_a = q.M - m0, rows_b = q.N - n0;
  const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(q.A + m0, ((long)(q.K - 1) * q.lda + rows_a) * 2);
  const __amdgpu_buffer_rsrc_t b_rsrc = make_rsrc(q.B + n0, ((long)(q.K - 1) * q.ldb + rows_b) * 2);
  const StagePlan<2> pa = make_plan<true, C::BM, 2, 32>(wave, lane, q.lda, rows_a);
  const StagePlan<2> pb = make_plan<true, C::BN, 2, 32>(wave, lane, q.ldb, rows_b);
  const int K = q.K;
  const int nk = (K + 31) >> 5;

  f32x4 acc[C::FM][C::FN];
#pragma unroll
  for (int mi = 0; mi < C::FM; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::FN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 accb0 = {0.f, 0.f, 0.f, 0.f}, accb1 = {0.f, 0.f, 0.f, 0.f};
  const bool do_bg = BG && q.bgrad != nullptr && tile_n < q.bg_parts;
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

#pragma unroll
  for (int st = 0; st < D; ++st) {
    stage_tile<true>(a_rsrc, smem + st * C::STAGE_BYTES, wave, pa, q.lda, st * 32, K);
    stage_tile<true>(b_rsrc, smem + st * C::STAGE_BYTES + C::A_BYTES, wave, pb, q.ldb, st * 32, K);
  }
  CFHIP_WAIT_VMCNT((D - 1) * C::LPS);  // K-step 0 has landed; the younger ones stay in flight
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();  // the stagger

  // In a bias-gradient K-step (workgroup-uniform, one in bg_parts) every wave issues ONE extra MFMA per phase: fragment `wn`
  // of the phase's four, picked with wave-uniform selects (no run-time register index), behind a scalar branch at the end of
  // the M segment.  (History: with the builtin MFMA behind that branch the compiler copied both accumulators around the join
  // on every phase, +17 %, profiles/r03/gemm_grouped_bench.log; round 3a therefore duplicated the whole K loop.)
  int rd = 0, wr = D;
  auto k_step = [&](const int t, const bool bg_step) {
    {
      const char* a_tile = smem + rd * C::STAGE_BYTES;
      const char* b_tile = a_tile + C::A_BYTES;
      char* w_tile = smem + wr * C::STAGE_BYTES;
      const int kw = (t + D) * 32;
      bf16x8 bfr[C::FN], af[HM];
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        // ---- L segment: this phase's fragments, then half of the DMA of K-step t + D
        if (ph == 0) {
#pragma unroll
          for (int n = 0; n < C::FN; ++n) bfr[n] = frag_mmajor<C::BN>(b_tile, wn * 64 + n * 16, 0, lane);
        }
#pragma unroll
        for (int m = 0; m < HM; ++m) af[m] = frag_mmajor<C::BM>(a_tile, wm * 128 + (ph * HM + m) * 16, 0, lane);
        __builtin_amdgcn_sched_barrier(0);
        if (PLACE == 0) {
          if (ph == 0) stage_tile<true>(a_rsrc, w_tile, wave, pa, q.lda, kw, K);
          else stage_tile<true>(b_rsrc, w_tile + C::A_BYTES, wave, pb, q.ldb, kw, K);
        } else if (PLACE == 2) {
          if (ph == 0) stage_one<0>(a_rsrc, w_tile, wave, pa, q.lda, kw, K);
          else stage_one<0>(b_rsrc, w_tile + C::A_BYTES, wave, pb, q.ldb, kw, K);
        }
        if (ph == 1) CFHIP_WAIT_VMCNT(PLACE == 0 ? (D - 1) * 4 : PLACE == 1 ? (D - 2) * 4 + 2 : (D - 2) * 4 + 3);  // own DMA of K-step t + 1 retired
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- M segment (the compiler's lgkmcnt ladder in front of the MFMAs retires the fragment reads)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < HM; ++m) {
#pragma unroll
          for (int n = 0; n < C::FN; ++n)
            acc[ph * HM + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[n], af[m], acc[ph * HM + m][n], 0, 0, 0);
          if (PLACE != 0 && m == HM / 2 - 1) {  // after the 8th MFMA
            __builtin_amdgcn_sched_barrier(0);
            if (PLACE == 1) {
              if (ph == 0) stage_tile<true>(a_rsrc, w_tile, wave, pa, q.lda, kw, K);
              else stage_tile<true>(b_rsrc, w_tile + C::A_BYTES, wave, pb, q.ldb, kw, K);
            } else {
              if (ph == 0) stage_one<1>(a_rsrc, w_tile, wave, pa, q.lda, kw, K);
              else stage_one<1>(b_rsrc, w_tile + C::A_BYTES, wave, pb, q.ldb, kw, K);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if constexpr (BG) {
          if (bg_step) {  // wave-uniform: a scalar branch over 12 selects + one MFMA
            const bf16x8 mine = wn == 0 ? af[0] : wn == 1 ? af[1] : wn == 2 ? af[2] : af[3];
            // inline asm: the accumulator stays in ITS registers on both sides of the branch (the builtin made the compiler
            // copy accb0 / accb1 around the join on every phase).  Nothing reads them before the loop ends.
            // `s_nop 1`: the two wait states between a VALU write (the selects, the re-materialised ones) and an MFMA reading
            // it as A / B — the compiler pads nothing inside an asm string, and without them the MFMA read stale operands.
            if (ph == 0) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb0) : "v"(ones), "v"(mine));
            else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accb1) : "v"(ones), "v"(mine));
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      rd = rd + 1 == C::NSTAGE ? 0 : rd + 1;
      wr = wr + 1 == C::NSTAGE ? 0 : wr + 1;
    }
  };
  if constexpr (BG) {
    int next_bg = do_bg ? tile_n : 0x7fffffff;  // this workgroup's share of the bias-gradient reduction: t = tile_n (mod bg_parts)
    const int bg_stride = q.bg_parts;
    for (int t = 0; t < nk; ++t) {
      const bool mine = t == next_bg;
      k_step(t, mine);
      next_bg += mine ? bg_stride : 0;
    }
  } else {
    for (int t = 0; t < nk; ++t) k_step(t, false);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();  // every wave has executed the same number of barriers
  // the trailing (zero-fill) DMAs still target ring slots: retire them everywhere before the ring becomes the epilogue's strip
  CFHIP_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();

  if constexpr (BG) {
    // accb0 / accb1: colsum over k of fragment `wn` of phase 0 / 1 (rows (ph * 4 + wn) * 16 .. + 15 of the wave's 128)
    // -> partial sums of this workgroup's K-steps in the workspace; the last workgroup of the tile row to arrive adds the
    // bg_parts partials in order and writes db.  The partials cross XCDs, whose L2s are not coherent for plain accesses:
    // relaxed agent-scope atomic stores / loads (sc1: write-through / read from memory), every wave drains its stores
    // before the barrier, then ONE relaxed ticket.  No release / acquire fences: those write back and invalidate the whole
    // L2 of the XCD, which the kernels of the main queue are working in at the same time (the fenced version cost the
    // step +0.28 ms, profiles/r03/step_variants_d.log).  The ring is free here (DMAs retired, fragment reads done).
    if (do_bg) {  // workgroup-uniform
      float* part = p.ws + q.bg_ws + (long)tile_n * q.M;
      if ((lane >> 4) == 0) {
        const int m = m0 + wm * 128 + wn * 16 + (lane & 15);
        if (m < q.M) __hip_atomic_store(part + m, accb0[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (m + 64 < q.M) __hip_atomic_store(part + m + 64, accb1[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the sc1 (write-through) stores have reached memory
      __syncthreads();
      int* last = reinterpret_cast<int*>(smem);
      if (tid == 0) {
        int* c = p.cnt + q.bg_cnt + tile_m;
        const int arrived = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last = arrived == q.bg_parts - 1;
        if (arrived == q.bg_parts - 1) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
      }
      __syncthreads();
      if (*last) {
        const int m = m0 + tid;
        if (tid < 256 && m < q.M) {
          float sum = 0.f;
          for (int j = 0; j < q.bg_parts; ++j)
            sum += __hip_atomic_load(p.ws + q.bg_ws + (long)j * q.M + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q.bgrad[m] = (q.flags & 2) ? q.bgrad[m] + sum : sum;
        }
      }
      __syncthreads();  // `last` is read before the epilogue reuses the ring
    }
  }
  GemmParams gp;
  gp.C = q.C;
  gp.bias = nullptr;
  gp.aux_in = nullptr;
  gp.aux_out = nullptr;
  gp.M = q.M;
  gp.N = q.N;
  gp.ldc = q.ldc;
  gp.accumulate = q.flags & 1;
  epilogue_f32<CFHIP_EPI_NONE, C>(gp, acc, smem, m0, n0, wm, wn, wave, lane);
}

using CfgG4 = Cfg<256, 256, 2, 4, 4, 32>;  // 128 KiB ring
using CfgG5 = Cfg<256, 256, 2, 4, 5, 32>;  // 160 KiB ring (the whole LDS)

int g_grouped_variant = 0;  // 0: 5-slot ring, DMA 3 K-steps ahead; 1: 4 slots, 2 ahead; 2: 5 slots, 2 ahead; 3 / 4: as 0 with DMA placement 1 / 2

template <class C, int D, bool BG, int PLACE = 0>
int launch_grouped(const GroupedParams& p, int tiles, hipStream_t s) {
  void (*kern)(GroupedParams) = gemm_grouped_tn_kernel<C, D, BG, PLACE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) {
      cfhip_set_error("gemm_grouped: cannot reserve %d bytes of LDS: %s", C::LDS_BYTES, hipGetErrorString(e));
      return CFHIP_ERR_LAUNCH;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(C::NT), C::LDS_BYTES, s, p);
  return CFHIP_OK;
}

template <bool BG>
int launch_variant(const GroupedParams& p, int tiles, hipStream_t s) {
  switch (g_grouped_variant) {
    case 1: return launch_grouped<CfgG4, 2, BG>(p, tiles, s);
    case 2: return launch_grouped<CfgG5, 2, BG>(p, tiles, s);
    case 3: return launch_grouped<CfgG5, 3, BG, 1>(p, tiles, s);
    case 4: return launch_grouped<CfgG5, 3, BG, 2>(p, tiles, s);
    default: return launch_grouped<CfgG5, 3, BG>(p, tiles, s);
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int g_grouped_bias_shared = 1;  // 1: a tile row shares the bias-gradient reduction; 0: its first tile alone (round-3a behaviour)
int g_grouped_short_fastest = 1;  // 1: tiles of a problem numbered with the shorter tile dimension fastest; 0: row-major

// Bias-gradient workspace: one per stream (launches on one stream run one after the other; a different stream gets its own
// partial sums and counters), grown on demand.  Allocation is illegal inside a hipGraph capture: run one eager step first.
struct BiasWorkspace {
  hipStream_t stream;
  float* ws;
  int* cnt;
  size_t floats, counters;
};
constexpr int MAX_WS = 16;
BiasWorkspace g_bias_ws[MAX_WS];
int g_bias_ws_count = 0;
std::mutex g_bias_ws_mutex;

int bias_workspace(hipStream_t s, size_t floats, size_t counters, float** ws, int** cnt) {
  std::lock_guard<std::mutex> lock(g_bias_ws_mutex);
  BiasWorkspace* w = nullptr;
  for (int i = 0; i < g_bias_ws_count; ++i)
    if (g_bias_ws[i].stream == s) w = &g_bias_ws[i];
  if (w == nullptr) {
    CFHIP_REQUIRE(g_bias_ws_count < MAX_WS, "gemm_grouped: bias gradients requested on more than %d different streams", MAX_WS);
    w = &g_bias_ws[g_bias_ws_count++];
    *w = BiasWorkspace{s, nullptr, nullptr, 0, 0};
  }
  if (w->floats < floats || w->counters < counters) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    CFHIP_REQUIRE(cap == hipStreamCaptureStatusNone, "gemm_grouped: the bias-gradient workspace must exist before a hipGraph capture (run one eager step)");
    const size_t nf = floats > (size_t)1 << 18 ? floats * 2 : (size_t)1 << 18, nc = counters > 4096 ? counters * 2 : 4096;
    if (w->ws) (void)hipFree(w->ws);  // (synchronises the device: nothing is still reading the old buffers)
    if (w->cnt) (void)hipFree(w->cnt);
    w->ws = nullptr; w->cnt = nullptr; w->floats = w->counters = 0;
    if (hipMalloc(reinterpret_cast<void**>(&w->ws), nf * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&w->cnt), nc * sizeof(int)) != hipSuccess || hipMemset(w->cnt, 0, nc * sizeof(int)) != hipSuccess) {
      cfhip_set_error("gemm_grouped: cannot allocate the bias-gradient workspace (%zu floats, %zu counters)", nf, nc);
      return CFHIP_ERR_LAUNCH;
    }
    w->floats = nf; w->counters = nc;
  }
  *ws = w->ws;
  *cnt = w->cnt;
  return CFHIP_OK;
}

}  // namespace

int cfhip_internal_set_grouped_variant(int v) {
  // 0 .. 4: ring / DMA placement; +16: bias gradients by the first tile column alone; +32: row-major tile order (A/B runs)
  // (+64 / +128 selected the 80 / 64 KB plain-tile forms of round 4: measured 1.0-1.6 ms slower per ViT step and flat on the UNet,
  // removed in round 6 — profiles/r04/dw_plain_grouped_ab.log, profiles/r06/unet_variants_ab.txt)
  g_grouped_variant = v & 15;
  g_grouped_bias_shared = (v & 16) ? 0 : 1;
  g_grouped_short_fastest = (v & 32) ? 0 : 1;
  return CFHIP_OK;
}

extern "C" int cfhip_gemm_bf16_grouped_tn(const cfhip_gemm_problem* problems, int count, void* stream) {
  CFHIP_REQUIRE(problems != nullptr && count > 0, "gemm_grouped: no problems");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < count; base += GROUP_MAX) {
    const int n = count - base < GROUP_MAX ? count - base : GROUP_MAX;
    GroupedParams p;
    memset(&p, 0, sizeof(p));
    p.count = n;
    int tiles = 0;
    bool any_bg = false;
    size_t ws_floats = 0, ws_counters = 0;
    for (int i = 0; i < n; ++i) {
      const cfhip_gemm_problem& src = problems[base + i];
      CFHIP_REQUIRE(src.A && src.B && src.C, "gemm_grouped: null operand in problem %d", base + i);
      CFHIP_REQUIRE(src.M > 0 && src.N > 0 && src.K > 0, "gemm_grouped: empty problem %d (M=%d N=%d K=%d)", base + i, src.M, src.N, src.K);
      CFHIP_REQUIRE(src.M % 8 == 0 && src.N % 8 == 0 && src.lda % 8 == 0 && src.ldb % 8 == 0 && src.ldc % 4 == 0,
                    "gemm_grouped: problem %d: M, N, lda, ldb must be multiples of 8 and ldc of 4 (M=%d N=%d lda=%ld ldb=%ld ldc=%ld)",
                    base + i, src.M, src.N, (long)src.lda, (long)src.ldb, (long)src.ldc);
      CFHIP_REQUIRE(al16(src.A) && al16(src.B) && al16(src.C), "gemm_grouped: problem %d: operands must be 16-byte aligned", base + i);
      CFHIP_REQUIRE((long)src.K * src.lda * 2 < 0x7fffffffL && (long)src.K * src.ldb * 2 < 0x7fffffffL && (long)src.M * src.ldc * 4 < 0x7fffffffL,
                    "gemm_grouped: problem %d exceeds the 2 GiB descriptor range (K=%d lda=%ld ldb=%ld)", base + i, src.K, (long)src.lda, (long)src.ldb);
      GroupedProblem& d = p.pr[i];
      d.A = reinterpret_cast<const bf16_t*>(src.A);
      d.B = reinterpret_cast<const bf16_t*>(src.B);
      d.C = reinterpret_cast<float*>(src.C);
      d.bgrad = src.bias_grad;
      d.M = src.M; d.N = src.N; d.K = src.K;
      d.lda = (int)src.lda; d.ldb = (int)src.ldb; d.ldc = (int)src.ldc;
      d.tiles_m = (src.M + 255) / 256;
      d.tiles_n = (src.N + 255) / 256;
      tiles += d.tiles_m * d.tiles_n;
      d.tile_end = tiles;
      d.flags = (src.accumulate ? 1 : 0) | (src.bias_grad_accumulate ? 2 : 0) | (g_grouped_short_fastest && d.tiles_m < d.tiles_n ? 4 : 0);
      if (src.bias_grad != nullptr) {
        any_bg = true;
        d.bg_parts = g_grouped_bias_shared ? d.tiles_n : 1;
        d.bg_ws = (int)ws_floats;
        d.bg_cnt = (int)ws_counters;
        ws_floats += (size_t)d.bg_parts * src.M;
        ws_counters += d.tiles_m;
        CFHIP_REQUIRE(ws_floats < 0x7fffffffUL, "gemm_grouped: bias-gradient workspace of problem %d out of range", base + i);
      }
    }
    if (any_bg) {
      const int rc = bias_workspace(s, ws_floats, ws_counters, &p.ws, &p.cnt);
      if (rc != CFHIP_OK) return rc;
    }
    const int rc = any_bg ? launch_variant<true>(p, tiles, s) : launch_variant<false>(p, tiles, s);
    if (rc != CFHIP_OK) return rc;
    CFHIP_CHECK_LAUNCH("gemm_grouped_tn");
  }
  return CFHIP_OK;
}
