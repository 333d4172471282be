// Shared helpers for the gfx950 kernels of libcfhip.so (wave = 64 lanes, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cfhip.h"

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// One LDS-DMA instruction (buffer_load_dwordx4 ... lds: 64 lanes x 16 bytes -> LDS[lds_dst + 16 * lane], lds_dst wave-uniform)
// as INLINE ASM.  Round 3: hipcc's s_waitcnt pass treats the builtin form (__builtin_amdgcn_raw_ptr_buffer_load_lds) as a
// pending LDS store that may alias every later ds_read and puts `s_waitcnt vmcnt(0)` in front of the next LDS read — in the
// K loops of the GEMM kernels that drained the DMA queue once per K-step, i.e. the rings never prefetched anything (the ISA
// of every round-1/2 GEMM loop shows it: tools/isa_waits.py).  The asm form is invisible to that pass: completion is counted
// by hand (CFHIP_WAIT_VMCNT, then a barrier, then the ds_read — the protocol the kernels were written for anyway).  The LDS
// address is bound to M0 through an "{m0}" input constraint: the COMPILER writes M0 (and so knows it is written: a value it
// keeps there for movrel indexing / sendmsg / the builtin DMA form is rebuilt, not silently lost; an "m0" clobber is refused
// as a reserved register); the s_nop covers the M0-write -> LDS-DMA wait state, which the hazard recogniser cannot see
// inside an asm statement.
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, const void* lds_dst, unsigned voff) {
#ifdef CFHIP_DMA_BUILTIN  // A/B builds only (tools/build_variant.sh dmabuiltin -DCFHIP_DMA_BUILTIN): the round-1/2 form
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(const_cast<void*>(lds_dst)), 16, voff, 0, 0, 0);
  return;
#endif
  const unsigned la = (unsigned)(uintptr_t)LDS_PTR(const_cast<void*>(lds_dst));
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"{m0}"(la), "v"(voff), "s"(rsrc) : "memory");
}

// Every LDS-DMA this wave has issued has landed (follow with a barrier before other waves read the tile).  Needed because
// the asm form above is invisible to the compiler: `__syncthreads()` no longer implies the vmcnt(0) it used to emit for the
// builtin form.
// The same with a SCALAR byte offset on top of the lane's (soffset operand of the buffer instruction): the K advance of a GEMM
// operand is wave-uniform, so the per-lane offsets stay what they were computed to be once per tile — no VALU per K-step.
// (gfx9 raw buffers: the range check covers voffset only, an out-of-range lane stays out of range whatever the soffset.)
__device__ __forceinline__ void lds_dma16_s(__amdgpu_buffer_rsrc_t rsrc, const void* lds_dst, unsigned voff, unsigned soff) {
  const unsigned la = (unsigned)(uintptr_t)LDS_PTR(const_cast<void*>(lds_dst));
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"{m0}"(la), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

void cfhip_set_error(const char* fmt, ...);

#define CFHIP_REQUIRE(cond, ...)          \
  do {                                    \
    if (!(cond)) {                        \
      cfhip_set_error(__VA_ARGS__);       \
      return CFHIP_ERR_INVALID;           \
    }                                     \
  } while (0)

#define CFHIP_CHECK_LAUNCH(name)                                                \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      cfhip_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return CFHIP_ERR_LAUNCH;                                                  \
    }                                                                           \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((unsigned)v) << 16);
}
// f32 -> bf16 goes through the native __bf16 type: hipcc lowers the cast to gfx950's
// v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-quieting) — one instruction per PAIR instead of
// ~6 integer ops per element for a hand-rolled rounding.
typedef __bf16 bf16x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const bf16x2_native v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// ---- exact-erf GELU and its derivative ---------------------------------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 resolution): one exp, one
// reciprocal and five FMAs instead of the ~40-instruction libm erff — the GELU epilogues run 64
// evaluations per lane per output tile, so this is what keeps them off the GEMM's critical path.
// exp(-z^2) with z = |x|/sqrt(2) is exp(-x^2/2): the same value the derivative's pdf term needs.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
#ifdef CFHIP_GELU_PROBE  // benchmarking only (tools/build_variant.sh): what the epilogue GEMMs cost WITHOUT the erf arithmetic
  cdf = 0.5f;
  e = 1.0f;
  return;
#endif
  // raw v_rcp_f32 / v_exp_f32 (1 ulp): `__frcp_rn` / a plain division expand to the IEEE sequence
  // (2 v_div_scale + v_rcp + 5 fma + v_div_fmas + v_div_fixup per element) — measured: the GELU
  // epilogue's VALU work was 55-60 us of a 210 us FF1 GEMM launch with it.
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, fabsf(x), 1.0f));
  e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));  // exp(-x^2 / 2), argument <= 0
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  const float erf_abs = fmaf(-(poly * t), e, 1.0f);  // erf(|x| / sqrt 2)
  cdf = fmaf(copysignf(0.5f, x), erf_abs, 0.5f);     // Phi(x)
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  float cdf, e;
  gelu_parts(x, cdf, e);
  return fmaf(x * 0.39894228040143267794f, e, cdf);  // Phi(x) + x phi(x)
}

// ---- quick GELU x * sigmoid(1.702 x) (reference activations.py "quick_gelu", the CLIP towers) ------------
__device__ __forceinline__ float sigmoid_1702(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * (-1.702f * 1.4426950408889634f)));
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_1702(x); }
__device__ __forceinline__ float quick_gelu_grad_f(float x) {
  const float s = sigmoid_1702(x);
  return s * fmaf(1.702f * x, 1.0f - s, 1.0f);
}

// ---- Philox4x32-10 (counter-based generator of csrc/random.hip and the attention-probability dropout) --------------
struct Philox {
  unsigned c[4];
};

__device__ __forceinline__ Philox philox4x32_10(unsigned long long ctr, unsigned long long seed) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    c0 = hi1 ^ c1 ^ k0;
    c1 = lo1;
    c2 = hi0 ^ c3 ^ k1;
    c3 = lo0;
    k0 += W0;
    k1 += W1;
  }
  return Philox{{c0, c1, c2, c3}};
}

// ---- wave-level reductions over all 64 lanes ----------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
