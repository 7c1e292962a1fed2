// K1 — architecture-weighted mixed-edge sum (FusionMixedOp summed over incoming edges).
// Pure HBM-streaming kernels: one float4 per lane per input, all n_in loads issued before
// the FMAs; the backward fuses the n_in dx writes with n_in dot products
// (wave shuffle -> LDS -> one atomic per workgroup per scalar).
// Algorithmic bytes: fwd (n_in + 1) * T, bwd (2 * n_in + 1) * T, T = n_elem * 4.
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

template <int NIN>
__global__ __launch_bounds__(256) void mixsum_fwd_k(PtrsIn xs, const float* __restrict__ w,
                                                    int w_stride, float* __restrict__ out,
                                                    int64_t n4) {
  float wj[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) wj[j] = w[j * w_stride];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
    float4 acc = f4_scale(v[0], wj[0]);
#pragma unroll
    for (int j = 1; j < NIN; ++j) {
      acc.x = fmaf(wj[j], v[j].x, acc.x);
      acc.y = fmaf(wj[j], v[j].y, acc.y);
      acc.z = fmaf(wj[j], v[j].z, acc.z);
      acc.w = fmaf(wj[j], v[j].w, acc.w);
    }
    st4_wt(out + 4 * i, acc);
  }
}

template <int NIN>
__global__ __launch_bounds__(256) void mixsum_bwd_k(PtrsIn xs, PtrsOut dxs,
                                                    const float* __restrict__ w, int w_stride,
                                                    const float* __restrict__ g,
                                                    const float* __restrict__ g2, float* dw,
                                                    int dw_shards, int64_t dw_shard_stride,
                                                    uint32_t acc_mask, int64_t n4) {
  __shared__ float red[4 * NIN];
  float wj[NIN], part[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    wj[j] = w[j * w_stride];
    part[j] = 0.f;
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 g4 = reinterpret_cast<const float4*>(g)[i];
    {                                                        // (unconditional load: no wait at a join)
      const float4 g2v = reinterpret_cast<const float4*>(g2 != nullptr ? g2 : g)[i];
      g4 = f4_add(g4, g2 != nullptr ? g2v : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    if (dw != nullptr) {
      float4 v[NIN];
#pragma unroll
      for (int j = 0; j < NIN; ++j) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
#pragma unroll
      for (int j = 0; j < NIN; ++j) part[j] += f4_dot(g4, v[j]);
    }
    // dx_j: destinations may alias each other (the same state feeding two edges), so the
    // read-modify-writes stay in program order through non-restrict pointers.
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      float* d = dxs.p[j];
      if (d == nullptr) continue;
      float4 r = f4_scale(g4, wj[j]);
      if (acc_mask & (1u << j)) r = f4_add(r, reinterpret_cast<float4*>(d)[i]);
      st4_wt(d + 4 * i, r);
    }
  }
  if (dw == nullptr) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    float s = wave_sum(part[j]);
    if (lane == 0) red[wave * NIN + j] = s;
  }
  __syncthreads();
  if (threadIdx.x < NIN) {
    const int j = threadIdx.x;
    // hundreds of workgroups adding into the same n_in scalars serialise on the atomics:
    // spread them over dw_shards copies (summed by the arch-softmax backward)
    float* d = dw + (int64_t)(blockIdx.x % dw_shards) * dw_shard_stride;
    atomicAdd(d + j * w_stride, red[j] + red[NIN + j] + red[2 * NIN + j] + red[3 * NIN + j]);
  }
}


// K1 pair — the cell-level sum h = sum_j w_j x_j together with the first inner-cell sum of the
// step node it feeds.  In search mode FusionNode is called as node(h, h) (model_search.py:59), so
// NodeCell's first mixed sum (node_search.py:54) is z = (w2_0 + w2_1) * h: written from the same
// registers instead of a second launch that re-reads h twice.
template <int NIN>
__global__ __launch_bounds__(256) void mixsum_pair_fwd_k(PtrsIn xs, const float* __restrict__ w,
                                                         int w_stride, const float* __restrict__ w2,
                                                         int w2_stride, float* __restrict__ out,
                                                         float* __restrict__ out2, int64_t n4) {
  float wj[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) wj[j] = w[j * w_stride];
  const float s2 = w2[0] + w2[w2_stride];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 v[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
    float4 acc = f4_scale(v[0], wj[0]);
#pragma unroll
    for (int j = 1; j < NIN; ++j) {
      acc.x = fmaf(wj[j], v[j].x, acc.x);
      acc.y = fmaf(wj[j], v[j].y, acc.y);
      acc.z = fmaf(wj[j], v[j].z, acc.z);
      acc.w = fmaf(wj[j], v[j].w, acc.w);
    }
    st4_wt(out + 4 * i, acc);
    st4_wt(out2 + 4 * i, f4_scale(acc, s2));
  }
}

// backward of the pair: G = gh + (w2_0 + w2_1) * gz is the full gradient of h (gh: what the other
// consumers of h already accumulated, may be null); dx_j (+)= w_j G, dw_j += <G, x_j>,
// dw2_0 += <gz, h>, dw2_1 += <gz, h>.
// DOTS = false (dw == dw2 == NULL at the C ABI): nobody differentiates the edge weights — the weight step of the search
// loop — so the NIN inputs and h, read only for the dot products, are not loaded: dx_j needs G alone.
template <int NIN, bool DOTS = true>
__global__ __launch_bounds__(256) void mixsum_pair_bwd_k(PtrsIn xs, PtrsOut dxs,
                                                         const float* __restrict__ w, int w_stride,
                                                         const float* __restrict__ w2, int w2_stride,
                                                         const float* __restrict__ h,
                                                         const float* __restrict__ gh,
                                                         const float* __restrict__ gz,
                                                         const float* __restrict__ gz2, float* dw,
                                                         float* dw2, int dw_shards,
                                                         int64_t dw_shard_stride, uint32_t acc_mask,
                                                         int64_t n4) {
  __shared__ float red[4 * (NIN + 1)];
  float wj[NIN], part[NIN + 1];
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    wj[j] = w[j * w_stride];
    part[j] = 0.f;
  }
  part[NIN] = 0.f;
  const float s2 = w2[0] + w2[w2_stride];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    // every load unconditional (an absent optional operand reads gz again and is masked out): a load under `if` is
    // a branch whose join waits for vmcnt(0) — gz, gz2 and gh were three dependent round trips before the operands
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 z4 = reinterpret_cast<const float4*>(gz)[i];
    const float4 z2v = reinterpret_cast<const float4*>(gz2 != nullptr ? gz2 : gz)[i];
    float4 h4 = zero4;
    if constexpr (DOTS) h4 = reinterpret_cast<const float4*>(h)[i];
    const float4 ghv = reinterpret_cast<const float4*>(gh != nullptr ? gh : gz)[i];
    float4 v[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      if constexpr (DOTS) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
      else v[j] = zero4;
    }
    z4 = f4_add(z4, gz2 != nullptr ? z2v : zero4);
    const float4 g4 = f4_add(f4_scale(z4, s2), gh != nullptr ? ghv : zero4);
#pragma unroll
    for (int j = 0; j < NIN; ++j) part[j] += f4_dot(g4, v[j]);
    part[NIN] += f4_dot(z4, h4);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      float* d = dxs.p[j];
      if (d == nullptr) continue;
      float4 r = f4_scale(g4, wj[j]);
      if (acc_mask & (1u << j)) r = f4_add(r, reinterpret_cast<float4*>(d)[i]);
      st4_wt(d + 4 * i, r);
    }
  }
  if constexpr (!DOTS) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j <= NIN; ++j) {
    float s = wave_sum(part[j]);
    if (lane == 0) red[wave * (NIN + 1) + j] = s;
  }
  __syncthreads();
  if (threadIdx.x <= NIN + 1) {
    const int j = threadIdx.x < NIN ? threadIdx.x : NIN;
    const float v = red[j] + red[NIN + 1 + j] + red[2 * (NIN + 1) + j] + red[3 * (NIN + 1) + j];
    const int64_t sh = (int64_t)(blockIdx.x % dw_shards) * dw_shard_stride;
    if (threadIdx.x < NIN) atomicAdd(dw + sh + j * w_stride, v);
    else atomicAdd(dw2 + sh + (threadIdx.x - NIN) * w2_stride, v);
  }
}

// The FIRST cell step's pair backward when the later steps' K1 backward launches left their input gradients to it
// (write-once: every step's mixed sum reads the same N cell inputs, model_search.py:58, so dx_j is a sum over the
// steps — instead of S read-modify-write passes over the N tensors, step t > 0 stores only its G_t
// (bmnas_mixsum_pair_bwd_lazy, g_full) and this launch writes   dx_j = w_j G + sum_t wm_t[j] G_t   once).
constexpr int kMaxMoreG = 2;
struct MoreG {
  const float* g[kMaxMoreG];     // G_t (n_elem)
  const float* w[kMaxMoreG];     // softmaxed edge weights of step t's sum: w[t][j * w_stride], j < NIN
};

template <int NIN, int NX, bool DOTS = true>
__global__ __launch_bounds__(256) void mixsum_pair_bwd_x_k(PtrsIn xs, PtrsOut dxs,
                                                           const float* __restrict__ w, int w_stride,
                                                           const float* __restrict__ w2, int w2_stride,
                                                           const float* __restrict__ h,
                                                           const float* __restrict__ gh,
                                                           const float* __restrict__ gz,
                                                           const float* __restrict__ gz2, float* dw,
                                                           float* dw2, int dw_shards,
                                                           int64_t dw_shard_stride, uint32_t acc_mask,
                                                           MoreG X, int64_t n4) {
  __shared__ float red[4 * (NIN + 1)];
  float wj[NIN], wx[NX][NIN], part[NIN + 1];
#pragma unroll
  for (int j = 0; j < NIN; ++j) {
    wj[j] = w[j * w_stride];
    part[j] = 0.f;
#pragma unroll
    for (int t = 0; t < NX; ++t) wx[t][j] = X.w[t][j * w_stride];
  }
  part[NIN] = 0.f;
  const float s2 = w2[0] + w2[w2_stride];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    // every load unconditional (an absent optional operand reads gz again and is masked out): a load under `if` is
    // a branch whose join waits for vmcnt(0) — gz, gz2 and gh were three dependent round trips before the operands
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 z4 = reinterpret_cast<const float4*>(gz)[i];
    const float4 z2v = reinterpret_cast<const float4*>(gz2 != nullptr ? gz2 : gz)[i];
    float4 h4 = zero4;
    if constexpr (DOTS) h4 = reinterpret_cast<const float4*>(h)[i];
    const float4 ghv = reinterpret_cast<const float4*>(gh != nullptr ? gh : gz)[i];
    float4 gx[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t) gx[t] = reinterpret_cast<const float4*>(X.g[t])[i];
    float4 v[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      if constexpr (DOTS) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
      else v[j] = zero4;
    }
    z4 = f4_add(z4, gz2 != nullptr ? z2v : zero4);
    const float4 g4 = f4_add(f4_scale(z4, s2), gh != nullptr ? ghv : zero4);
#pragma unroll
    for (int j = 0; j < NIN; ++j) part[j] += f4_dot(g4, v[j]);
    part[NIN] += f4_dot(z4, h4);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      float* d = dxs.p[j];
      if (d == nullptr) continue;
      float4 r = f4_scale(g4, wj[j]);
#pragma unroll
      for (int t = 0; t < NX; ++t) {
        r.x = fmaf(wx[t][j], gx[t].x, r.x);
        r.y = fmaf(wx[t][j], gx[t].y, r.y);
        r.z = fmaf(wx[t][j], gx[t].z, r.z);
        r.w = fmaf(wx[t][j], gx[t].w, r.w);
      }
      if (acc_mask & (1u << j)) r = f4_add(r, reinterpret_cast<float4*>(d)[i]);
      st4_w0<10>(d + 4 * i, r);
    }
  }
  if constexpr (!DOTS) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int j = 0; j <= NIN; ++j) {
    float s = wave_sum(part[j]);
    if (lane == 0) red[wave * (NIN + 1) + j] = s;
  }
  __syncthreads();
  if (threadIdx.x <= NIN + 1) {
    const int j = threadIdx.x < NIN ? threadIdx.x : NIN;
    const float v = red[j] + red[NIN + 1 + j] + red[2 * (NIN + 1) + j] + red[3 * (NIN + 1) + j];
    const int64_t sh = (int64_t)(blockIdx.x % dw_shards) * dw_shard_stride;
    if (threadIdx.x < NIN) atomicAdd(dw + sh + j * w_stride, v);
    else atomicAdd(dw2 + sh + (threadIdx.x - NIN) * w2_stride, v);
  }
}

inline int grid_for(int64_t n4) {
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

}  // namespace

#define MIXSUM_DISPATCH(N, CALL) \
  switch (N) {                   \
    case 1: CALL(1); break;      \
    case 2: CALL(2); break;      \
    case 3: CALL(3); break;      \
    case 4: CALL(4); break;      \
    case 5: CALL(5); break;      \
    case 6: CALL(6); break;      \
    case 7: CALL(7); break;      \
    case 8: CALL(8); break;      \
    case 9: CALL(9); break;      \
    case 10: CALL(10); break;    \
    case 11: CALL(11); break;    \
    case 12: CALL(12); break;    \
    case 13: CALL(13); break;    \
    case 14: CALL(14); break;    \
    case 15: CALL(15); break;    \
    case 16: CALL(16); break;    \
    default: return BMNAS_E_LIMIT; \
  }

extern "C" int bmnas_mixsum_fwd(const float* const* xs, int n_in, const float* w, int w_stride,
                                float* out, int64_t n_elem, void* stream) {
  if (!xs || !w || !out || n_in < 1 || n_elem < 0 || w_stride < 1) return BMNAS_E_ARG;
  if (n_in > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (n_elem % 4 != 0) return BMNAS_E_SHAPE;
  if (n_elem == 0) return 0;
  PtrsIn p{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
  }
  const int64_t n4 = n_elem / 4;
  hipStream_t st = (hipStream_t)stream;
#define CALL(N) hipLaunchKernelGGL(mixsum_fwd_k<N>, dim3(grid_for(n4)), dim3(256), 0, st, p, w, w_stride, out, n4)
  MIXSUM_DISPATCH(n_in, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_mixsum_bwd(const float* const* xs, float* const* dxs, int n_in,
                                const float* w, int w_stride, const float* g, const float* g2,
                                float* dw, int dw_shards, int64_t dw_shard_stride,
                                uint32_t accumulate_mask, int64_t n_elem, void* stream) {
  if (!xs || !dxs || !w || !g || n_in < 1 || n_elem < 0 || w_stride < 1 || dw_shards < 1)
    return BMNAS_E_ARG;
  if (n_in > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (n_elem % 4 != 0) return BMNAS_E_SHAPE;
  if (n_elem == 0) return 0;
  PtrsIn p{};
  PtrsOut d{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
    d.p[j] = dxs[j];
  }
  const int64_t n4 = n_elem / 4;
  hipStream_t st = (hipStream_t)stream;
#define CALL(N) hipLaunchKernelGGL(mixsum_bwd_k<N>, dim3(grid_for(n4)), dim3(256), 0, st, p, d, w, w_stride, g, g2, dw, dw_shards, dw_shard_stride, accumulate_mask, n4)
  MIXSUM_DISPATCH(n_in, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_mixsum_pair_fwd(const float* const* xs, int n_in, const float* w, int w_stride,
                                     const float* w2, int w2_stride, float* out, float* out2,
                                     int64_t n_elem, void* stream) {
  if (!xs || !w || !w2 || !out || !out2 || n_in < 1 || n_elem < 0 || w_stride < 1 || w2_stride < 1)
    return BMNAS_E_ARG;
  if (n_in > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (n_elem % 4 != 0) return BMNAS_E_SHAPE;
  if (n_elem == 0) return 0;
  PtrsIn p{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
  }
  const int64_t n4 = n_elem / 4;
  hipStream_t st = (hipStream_t)stream;
#define CALL(N) hipLaunchKernelGGL(mixsum_pair_fwd_k<N>, dim3(grid_for(n4)), dim3(256), 0, st, p, w, w_stride, w2, w2_stride, out, out2, n4)
  MIXSUM_DISPATCH(n_in, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_mixsum_pair_bwd(const float* const* xs, float* const* dxs, int n_in,
                                     const float* w, int w_stride, const float* w2, int w2_stride,
                                     const float* h, const float* gh, const float* gz,
                                     const float* gz2, float* dw, float* dw2, int dw_shards,
                                     int64_t dw_shard_stride,
                                     uint32_t accumulate_mask, int64_t n_elem, void* stream) {
  if (!xs || !dxs || !w || !w2 || !h || !gz || n_in < 1 || n_elem < 0 || w_stride < 1 || w2_stride < 1 ||
      dw_shards < 1)
    return BMNAS_E_ARG;
  if ((dw == nullptr) != (dw2 == nullptr)) return BMNAS_E_ARG;      // both (the dot products wanted) or neither
  const bool dots = dw != nullptr;
  if (n_in > BMNAS_MAX_PTRS - 1) return BMNAS_E_LIMIT;
  if (n_elem % 4 != 0) return BMNAS_E_SHAPE;
  if (n_elem == 0) return 0;
  PtrsIn p{};
  PtrsOut d{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
    d.p[j] = dxs[j];
  }
  const int64_t n4 = n_elem / 4;
  hipStream_t st = (hipStream_t)stream;
#define CALL(N)                                                                                                    \
  do {                                                                                                             \
    if (dots)                                                                                                      \
      hipLaunchKernelGGL((mixsum_pair_bwd_k<N, true>), dim3(grid_for(n4)), dim3(256), 0, st, p, d, w, w_stride,    \
                         w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride, accumulate_mask, n4); \
    else                                                                                                           \
      hipLaunchKernelGGL((mixsum_pair_bwd_k<N, false>), dim3(grid_for(n4)), dim3(256), 0, st, p, d, w, w_stride,   \
                         w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride, accumulate_mask, n4); \
  } while (0)
  switch (n_in) {
    case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;
    case 4: CALL(4); break;   case 5: CALL(5); break;   case 6: CALL(6); break;
    case 7: CALL(7); break;   case 8: CALL(8); break;   case 9: CALL(9); break;
    case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break;
    default: return BMNAS_E_LIMIT;
  }
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_mixsum_pair_bwd_x(const float* const* xs, float* const* dxs, int n_in,
                                       const float* w, int w_stride, const float* w2, int w2_stride,
                                       const float* h, const float* gh, const float* gz,
                                       const float* gz2, float* dw, float* dw2, int dw_shards,
                                       int64_t dw_shard_stride, uint32_t accumulate_mask,
                                       const float* const* g_more, const float* const* w_more, int n_more,
                                       int64_t n_elem, void* stream) {
  if (n_more == 0)
    return bmnas_mixsum_pair_bwd(xs, dxs, n_in, w, w_stride, w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards,
                                 dw_shard_stride, accumulate_mask, n_elem, stream);
  if (!xs || !dxs || !w || !w2 || !h || !gz || !g_more || !w_more || n_in < 1 || n_elem < 0 || w_stride < 1 ||
      w2_stride < 1 || dw_shards < 1)
    return BMNAS_E_ARG;
  if ((dw == nullptr) != (dw2 == nullptr)) return BMNAS_E_ARG;      // both (the dot products wanted) or neither
  const bool dots = dw != nullptr;
  if (n_more < 0 || n_more > kMaxMoreG || n_in > BMNAS_MAX_PTRS - 1) return BMNAS_E_LIMIT;
  if (n_elem % 4 != 0) return BMNAS_E_SHAPE;
  if (n_elem == 0) return 0;
  PtrsIn p{};
  PtrsOut d{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
    d.p[j] = dxs[j];
  }
  MoreG X{};
  for (int t = 0; t < n_more; ++t) {
    if (!g_more[t] || !w_more[t]) return BMNAS_E_ARG;
    X.g[t] = g_more[t];
    X.w[t] = w_more[t];
  }
  const int64_t n4 = n_elem / 4;
  hipStream_t st = (hipStream_t)stream;
#define CALL2(N, X_)                                                                                                \
  do {                                                                                                              \
    if (dots)                                                                                                       \
      hipLaunchKernelGGL((mixsum_pair_bwd_x_k<N, X_, true>), dim3(grid_for(n4)), dim3(256), 0, st, p, d, w,         \
                         w_stride, w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride,              \
                         accumulate_mask, X, n4);                                                                   \
    else                                                                                                            \
      hipLaunchKernelGGL((mixsum_pair_bwd_x_k<N, X_, false>), dim3(grid_for(n4)), dim3(256), 0, st, p, d, w,        \
                         w_stride, w2, w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride,              \
                         accumulate_mask, X, n4);                                                                   \
  } while (0)
#define CALL(N)                   \
  do {                            \
    if (n_more == 1) CALL2(N, 1); \
    else CALL2(N, 2);             \
  } while (0)
  switch (n_in) {
    case 1: CALL(1); break;   case 2: CALL(2); break;   case 3: CALL(3); break;
    case 4: CALL(4); break;   case 5: CALL(5); break;   case 6: CALL(6); break;
    case 7: CALL(7); break;   case 8: CALL(8); break;   case 9: CALL(9); break;
    case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break;
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break;
    default: return BMNAS_E_LIMIT;
  }
#undef CALL
#undef CALL2
  BMNAS_CHECK_LAUNCH();
  return 0;
}
