// lstm_common.h - device helpers shared by lstm.hip (row-tile kernel) and lstm_coop.hip (column-split kernel)
#pragma once
#include "fsnp_common.h"

namespace fsnp {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == FSNP_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == FSNP_ACT_RELU6) return fminf(fmaxf(v, 0.0f), 6.0f);
    if (act == FSNP_ACT_TANH) return tanhf(v);
    return v;
}

// float index of A element (row, k) inside an A-fragment-ordered LDS matrix
__host__ __device__ __forceinline__ int a_frag_index(int row, int k) {
    return (((k >> 3) * 64) + ((k & 1) * 32) + row) * 4 + ((k >> 1) & 3);
}


}  // namespace fsnp
