// Row-softmax backward of the architecture tensors (alpha / betas / gammas), shared by the
// stand-alone launch (bnmix.hip) and the backward epilogue launch (layernorm.hip).
#pragma once
#include "common.hpp"

namespace {

struct ArchPack {
  const float* a[BMNAS_MAX_PTRS];     // fwd: logits      bwd: softmax weights
  const float* b[BMNAS_MAX_PTRS];     //                  bwd: dweights
  float* o[BMNAS_MAX_PTRS];           // fwd: weights     bwd: dlogits
  int rows[BMNAS_MAX_PTRS], cols[BMNAS_MAX_PTRS];
  int n, n_shards;
  int64_t shard_stride;               // bwd: dweights are summed over n_shards copies
};

// one wavefront per row r (of the concatenation of all tensors): lane = (shard group, column);
// shards beyond 16 are walked by the same lane
__device__ __forceinline__ void arch_softmax_bwd_row(const ArchPack& P, int r, int lane) {
  const int col = lane & 3, sg = lane >> 2;
  for (int t = 0; t < P.n; ++t) {
    if (r < P.rows[t]) {
      const int cols = P.cols[t];
      float dw = 0.f, w = 0.f;
      if (col < cols) {
        const float* dwp = P.b[t] + r * cols + col;
        for (int sh = sg; sh < P.n_shards; sh += 16) dw += dwp[(int64_t)sh * P.shard_stride];
        w = P.a[t][r * cols + col];
      }
      dw = row_stride4_sum(dw);
      dw = xor16_sum(dw);
      dw = xor32_sum(dw);                                          // every lane: total of its column
      float dot = w * dw;
      dot += lane_xor1(dot);
      dot += lane_xor2(dot);                                       // sum over the 4 columns
      if (sg == 0 && col < cols) P.o[t][r * cols + col] = w * (dw - dot);
      return;
    }
    r -= P.rows[t];
  }
}

// host side: fill a pack from the C-ABI arrays; returns the total row count or a negative error
inline int fill_arch_pack(ArchPack& P, const float* const* a, const float* const* dw, float* const* out,
                          const int* rows, const int* cols, int n, int backward, int n_shards,
                          int64_t shard_stride) {
  if (!a || !out || !rows || !cols || n < 1 || (backward && !dw) || n_shards < 1) return BMNAS_E_ARG;
  if (n > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  int total = 0;
  for (int t = 0; t < n; ++t) {
    if (!a[t] || !out[t] || rows[t] < 1 || cols[t] < 1 || cols[t] > 4 || (backward && !dw[t]))
      return BMNAS_E_ARG;
    P.a[t] = a[t];
    P.b[t] = backward ? dw[t] : nullptr;
    P.o[t] = out[t];
    P.rows[t] = rows[t];
    P.cols[t] = cols[t];
    total += rows[t];
  }
  P.n = n;
  P.n_shards = n_shards;
  P.shard_stride = shard_stride;
  return total;
}

}  // namespace
