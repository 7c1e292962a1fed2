// Multi-tensor Adam: the w- and alpha-updates of the search loop
// (torch.optim.Adam built at mmimdb_darts_searchable.py:28-33, stepped at
// train_searchable/mmimdb.py:101 and architect.py:24) as ONE launch over every tensor of
// the optimizer instead of ~10 foreach launches x param groups.
//
// Arithmetic = torch.optim.Adam (amsgrad off, maximize off, L2 weight decay folded into the
// gradient), operation by operation:
//   g   = grad + wd * p                              (only if wd != 0)
//   m   = lerp(m, g, 1 - beta1)
//   v   = v * beta2 + (1 - beta2) * g * g
//   p  += -(lr / bc1) * ( m / ( sqrt(v) / sqrt(bc2) + eps ) )
// The step-dependent scalars (-(lr/bc1), sqrt(bc2)) are computed by the host in double, as
// torch does, and handed over in a small device table (`hyp`, 8 floats per row) that the host
// refreshes before every launch: a captured hipGraph replays with new learning rates / step
// counts without re-capture.
// HBM-streaming: 4 reads + 3 writes per element = 28 B/param.
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr int kChunk = 1024;      // elements per workgroup (256 lanes x float4)

struct Hyp {
  float neg_step, bc2_sqrt, beta2, eps, wd, w1, w2;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const Hyp& h) {
  if (h.wd != 0.f) g = g + h.wd * p;
  const float d = g - m;
  m = (h.w1 < 0.5f) ? m + h.w1 * d : g - d * (1.f - h.w1);      // at::lerp's two branches
  v = v * h.beta2 + h.w2 * (g * g);
  const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
  p = p + h.neg_step * (m / denom);
}

__global__ __launch_bounds__(256) void adam_multi_k(const bmnas_adam_tensor_t* __restrict__ tensors,
                                                    const int32_t* __restrict__ chunks,
                                                    const float* __restrict__ hyp) {
  const int ti = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
  const bmnas_adam_tensor_t t = tensors[ti];
  const float* hr = hyp + 8 * t.hyp_row;
  Hyp h;
  h.neg_step = hr[0]; h.bc2_sqrt = hr[1]; h.beta2 = hr[3]; h.eps = hr[4]; h.wd = hr[5];
  h.w1 = hr[6]; h.w2 = hr[7];
  const int64_t e = (int64_t)ci * kChunk + threadIdx.x * 4;
  if (e >= t.numel) return;
  const uintptr_t al = (uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq;
  if ((al & 15) == 0 && e + 4 <= t.numel) {
    float4 p = ld4(t.param + e), m = ld4(t.exp_avg + e), v = ld4(t.exp_avg_sq + e);
    const float4 g = ld4(t.grad + e);
    adam1(p.x, g.x, m.x, v.x, h);
    adam1(p.y, g.y, m.y, v.y, h);
    adam1(p.z, g.z, m.z, v.z, h);
    adam1(p.w, g.w, m.w, v.w, h);
    st4(t.param + e, p);
    st4(t.exp_avg + e, m);
    st4(t.exp_avg_sq + e, v);
  } else {
    const int64_t end = (e + 4 < t.numel) ? e + 4 : t.numel;
    for (int64_t i = e; i < end; ++i) {
      float p = t.param[i], m = t.exp_avg[i], v = t.exp_avg_sq[i];
      adam1(p, t.grad[i], m, v, h);
      t.param[i] = p;
      t.exp_avg[i] = m;
      t.exp_avg_sq[i] = v;
    }
  }
}

}  // namespace

extern "C" int bmnas_adam_chunk_elems(void) { return kChunk; }

extern "C" int bmnas_adam_multi(const bmnas_adam_tensor_t* tensors, const int32_t* chunks, int n_chunks,
                                const float* hyp, void* stream) {
  if (!tensors || !chunks || !hyp || n_chunks < 0) return BMNAS_E_ARG;
  if (n_chunks == 0) return 0;
  hipLaunchKernelGGL(adam_multi_k, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, hyp);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
