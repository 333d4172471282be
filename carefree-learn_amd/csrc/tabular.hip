// A16: the tabular encoder of the FCNN / ml models — `ml_encoder.Encoder.forward` + `CommonMLModel.encode`
// (reference modules/core/ml_encoder.py:131-258, models/ml/common.py:67-93) as ONE gather kernel:
//
//   merged_all[b] = [ numerical columns of x[b] (in column order)
//                   | one-hot(x[b][c]) for every one-hot column c (in column order)
//                   | W_c[x[b][c]] for every embedding column c (in column order) ]
//
// with the reference's out-of-bound imputation (value >= dim -> 0; the float -> int64 conversion truncates like
// `.to(torch.long)`).  Index arithmetic only on the categorical part: the one-hot block and the gathered embedding rows
// are bit-exact.  One thread per output element, driven by a per-output-column plan (source column, kind, payload)
// that the host builds once per encoder; embedding tables are addressed through a device array of base pointers
// (one nn.Parameter per column, as in the reference's state_dict: `embeddings.<col>.weights`).
#include "common.h"

namespace {

// plan entry per output column: src = input column; kind 0 = copy, 1 = one-hot (payload = class id),
// 2 = embedding (payload = table column; table = tables[table_id], its row pitch = pitch)
struct PlanEntry {
  int src, kind, payload, table_id, dim, pitch;
};

// Known divergences from the reference, on inputs it does not define: (1) the out-of-bound imputation
// (value >= dim -> 0, ml_encoder.py:176-183) is applied to EVERY categorical column, also on the mixed one-hot / embedding
// path where the reference indexes with the raw value and raises; (2) negative or NaN categorical values give an all-zero
// one-hot row / a zero embedding instead of an index error.  In-range inputs are bit-exact (tests/test_gpu_stochastic_tabular.py).
// The host (modules.MLEncoder._plan) checks that every categorical column index is inside the input.
__device__ __forceinline__ long categorical_index(float v, int dim) {
  if (v >= (float)dim) return 0;  // ml_encoder.py:176-183: oob -> 0
  return (long)v;                 // .to(torch.long): truncation
}

__global__ void ml_encode_fwd_kernel(const float* __restrict__ x, long B, int F, long xs, const PlanEntry* __restrict__ plan,
                                     int Fo, const float* const* __restrict__ tables, float* __restrict__ out) {
  const long total = B * (long)Fo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / Fo;
    const int j = (int)(i - b * Fo);
    const PlanEntry e = plan[j];
    const float v = x[b * xs + e.src];
    float o;
    if (e.kind == 0) {
      o = v;
    } else {
      const long idx = categorical_index(v, e.dim);
      if (e.kind == 1) o = idx == e.payload ? 1.0f : 0.0f;
      else o = (idx >= 0 && idx < e.dim) ? tables[e.table_id][idx * e.pitch + e.payload] : 0.0f;
    }
    out[i] = o;
  }
}

// indices[b][k] = the int64 index of categorical column k (the reference's EncodingResult.indices)
__global__ void ml_encode_indices_kernel(const float* __restrict__ x, long B, long xs, const int* __restrict__ cols,
                                         const int* __restrict__ dims, int K, long long* __restrict__ indices) {
  const long total = B * (long)K;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / K;
    const int k = (int)(i - b * K);
    indices[i] = categorical_index(x[b * xs + cols[k]], dims[k]);
  }
}

// backward: numerical columns -> dx (copy), embedding columns -> scatter-add into the tables' gradients
// (f32 hardware atomics: many rows share an index).  One-hot outputs carry no gradient.
__global__ void ml_encode_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, long B, int F, long xs,
                                     const PlanEntry* __restrict__ plan, int Fo, float* const* __restrict__ dtables,
                                     float* __restrict__ dx) {
  const long total = B * (long)Fo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / Fo;
    const int j = (int)(i - b * Fo);
    const PlanEntry e = plan[j];
    const float g = dout[i];
    if (e.kind == 0) {
      if (dx != nullptr) dx[b * F + e.src] = g;
    } else if (e.kind == 2) {
      float* dt = dtables[e.table_id];
      if (dt != nullptr) {
        const long idx = categorical_index(x[b * xs + e.src], e.dim);
        if (idx >= 0 && idx < e.dim) atomicAdd(dt + idx * e.pitch + e.payload, g);
      }
    }
  }
}

inline int grid_for(long total) {
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  return blocks < 1 ? 1 : (int)blocks;
}

}  // namespace

extern "C" int cfhip_ml_encode_fwd(const float* x, int64_t B, int F, int64_t x_row_stride, const int32_t* plan, int Fo,
                                   const void* const* tables, float* out, void* stream) {
  CFHIP_REQUIRE(x && plan && out, "ml_encode_fwd: null pointer");
  CFHIP_REQUIRE(B > 0 && F > 0 && Fo > 0, "ml_encode_fwd: empty problem");
  hipLaunchKernelGGL(ml_encode_fwd_kernel, dim3(grid_for(B * (long)Fo)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, (long)B, F, (long)x_row_stride, reinterpret_cast<const PlanEntry*>(plan), Fo,
                     reinterpret_cast<const float* const*>(tables), out);
  CFHIP_CHECK_LAUNCH("ml_encode_fwd");
  return CFHIP_OK;
}

extern "C" int cfhip_ml_encode_indices(const float* x, int64_t B, int64_t x_row_stride, const int32_t* cols,
                                       const int32_t* dims, int K, int64_t* indices, void* stream) {
  CFHIP_REQUIRE(x && cols && dims && indices, "ml_encode_indices: null pointer");
  CFHIP_REQUIRE(B > 0 && K > 0, "ml_encode_indices: empty problem");
  hipLaunchKernelGGL(ml_encode_indices_kernel, dim3(grid_for(B * (long)K)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, (long)B, (long)x_row_stride, cols, dims, K,
                     reinterpret_cast<long long*>(indices));
  CFHIP_CHECK_LAUNCH("ml_encode_indices");
  return CFHIP_OK;
}

extern "C" int cfhip_ml_encode_bwd(const float* dout, const float* x, int64_t B, int F, int64_t x_row_stride,
                                   const int32_t* plan, int Fo, void* const* dtables, float* dx, void* stream) {
  CFHIP_REQUIRE(dout && x && plan, "ml_encode_bwd: null pointer");
  CFHIP_REQUIRE(B > 0 && F > 0 && Fo > 0, "ml_encode_bwd: empty problem");
  hipLaunchKernelGGL(ml_encode_bwd_kernel, dim3(grid_for(B * (long)Fo)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     dout, x, (long)B, F, (long)x_row_stride, reinterpret_cast<const PlanEntry*>(plan), Fo,
                     reinterpret_cast<float* const*>(dtables), dx);
  CFHIP_CHECK_LAUNCH("ml_encode_bwd");
  return CFHIP_OK;
}
