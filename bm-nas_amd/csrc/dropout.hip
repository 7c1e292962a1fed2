// The dropout multipliers of one site as a tensor: what every kernel with a bmnas_dropout_t argument applies
// to element e of that site's (b, C, L) output, written out so that a checker can replay the same step on the
// CPU under the SAME masks (nn.Dropout's generator cannot be matched bit for bit; the Philox stream can be
// exported).  Audit / test entry point: nothing on the hypernet path launches it.
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

__global__ __launch_bounds__(256) void dropout_mask_k(DropCfg d, int64_t n4, int64_t n, float* __restrict__ out) {
  const DropRt r = drop_begin(d);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 m = drop_mult4(r, (uint64_t)(i * 4));
    if (i * 4 + 4 <= n) {
      st4(out + i * 4, m);
    } else {
      const float v[4] = {m.x, m.y, m.z, m.w};
      for (int k = 0; i * 4 + k < n; ++k) out[i * 4 + k] = v[k];
    }
  }
}

}  // namespace

extern "C" int bmnas_dropout_mask(bmnas_dropout_t drop, int64_t n_elem, float* out, void* stream) {
  if (!out || n_elem <= 0) return BMNAS_E_ARG;
  DropCfg d;
  d.thr = drop.thr; d.scale = drop.scale; d.seed = drop.seed; d.offset = drop.offset; d.step = drop.step;
  const int64_t n4 = (n_elem + 3) / 4;
  const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dropout_mask_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d, n4, n_elem, out);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
