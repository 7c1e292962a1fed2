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
#include <cstring>
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
    st4_wtg<5>(t.param + e, p);
    st4_wtg<5>(t.exp_avg + e, m);
    st4_wtg<5>(t.exp_avg_sq + e, v);
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

// ---- the batch into a captured step's static tensors: ONE launch (round 5) -----------------------------------------
// A replayed step reads its inputs from fixed addresses (bmnas.graph.GraphedTrainStep / GraphedForward): every call
// copies the caller's batch there first.  torch._foreach_copy_ is one launch only for lists it can batch (one dtype,
// and it fell back to a memcpy per tensor for the 8 + 1 small tensors of an NTU / Ego batch: 9 x 4.9 us in front of a
// 0.2 ms step); class-index labels (int64) were a copy of their own in any case.  Here up to kCopyMax (src, dst, bytes)
// triples of ANY dtype travel by value in the kernel arguments — no descriptor table to stage — and one grid of
// 16-byte lanes moves them all.  Replaces the reference's `.to(device)` hand-over of a batch only in so far as the
// batch is already on the device (train_searchable/mmimdb.py:60-63): host tensors keep torch's H2D copy.
namespace {
constexpr int kCopyMax = 16;
constexpr int kBlobWords = 64;                  // 256 bytes by value: eight rows of Adam scalars
struct CopyArgs {
  const void* src[kCopyMax];
  void* dst[kCopyMax];
  long long bytes[kCopyMax];
  int first[kCopyMax + 1];     // first workgroup of tensor i; first[n] = number of copy workgroups
  int n;
  // a few words that travel BY VALUE in the kernel arguments and are stored to device memory by one extra workgroup:
  // the per-step scalars of a captured optimizer step (learning rate / bias corrections, bmnas.optim.Adam) ride in
  // the launch that precedes the replay instead of an H2D copy node inside the graph (4.7 us per optimizer step)
  unsigned int* blob_dst;
  int blob_words;
  unsigned int blob[kBlobWords];
  // ... and one 64-bit device counter it advances: the dropout step counter of a captured step without a fused cell
  // prologue (found networks, evaluation forwards) — torch's `counter.add_(span)` node at the end of such a graph
  unsigned long long* add_dst;
  unsigned long long add_val;
};
constexpr int kCopyChunk = 256 * 16 * 4;        // bytes per workgroup: four 16-byte pieces per lane

__global__ __launch_bounds__(256) void copy_batch_k(CopyArgs a) {
  if ((int)blockIdx.x >= a.first[kCopyMax]) {                   // the blob's workgroup (last of the grid)
    if ((int)threadIdx.x < kBlobWords) {
      unsigned int v = a.blob[0];
#pragma unroll
      for (int i = 1; i < kBlobWords; ++i) v = ((int)threadIdx.x == i) ? a.blob[i] : v;   // selects: no indexed kernarg load
      if ((int)threadIdx.x < a.blob_words) a.blob_dst[threadIdx.x] = v;
    }
    if (threadIdx.x == 0 && a.add_dst != nullptr) a.add_dst[0] += a.add_val;
    return;
  }
  int ti = 0;
#pragma unroll
  for (int i = 1; i < kCopyMax; ++i) ti = (i < a.n && (int)blockIdx.x >= a.first[i]) ? i : ti;
  const char* __restrict__ s = (const char*)a.src[0];
  char* __restrict__ d = (char*)a.dst[0];
  long long nb = a.bytes[0];
  int f = a.first[0];
#pragma unroll
  for (int i = 1; i < kCopyMax; ++i) {        // selects over the by-value arrays: no indexed kernarg load
    const bool hit = ti == i;
    s = hit ? (const char*)a.src[i] : s;
    d = hit ? (char*)a.dst[i] : d;
    nb = hit ? a.bytes[i] : nb;
    f = hit ? a.first[i] : f;
  }
  const long long base = (long long)((int)blockIdx.x - f) * kCopyChunk;
  const long long full = nb & ~15ll;                            // bytes in whole 16-byte pieces
  if (s == nullptr) {
    // zero-fill job (workgroup-uniform): the accumulators a captured per-op step adds into with atomics (BatchNorm batch
    // sums, weight / affine gradients, the classifier's split-K output) are cleared by THIS launch, in front of the
    // replay, instead of by torch.zeros / zero_fill launches inside it (bmnas.functions.STEP_ARENA)
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    if (((uintptr_t)d & 15) == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long o = base + ((long long)r * 256 + threadIdx.x) * 16;
        if (o < full) st16_wt(d + o, z4);
      }
      if (full < nb && full >= base && full < base + kCopyChunk && (long long)threadIdx.x == ((full - base) >> 4) % 256)
        for (long long o = full; o < nb; ++o) d[o] = 0;
    } else {
      for (int r = 0; r < 4; ++r) {
        const long long o0 = base + ((long long)r * 256 + threadIdx.x) * 16;
        for (long long o = o0; o < nb && o < o0 + 16; ++o) d[o] = 0;
      }
    }
    return;
  }
  if ((((uintptr_t)s | (uintptr_t)d) & 15) == 0 && nb >= 16) {
    // four 16-byte pieces per lane, every load issued before the first store; scalars, not arrays (an indexed
    // per-lane array here was promoted to 16 KB of LDS per workgroup and the launch took 18 us for 9.4 MB)
    const long long o0 = base + (long long)threadIdx.x * 16, o1 = o0 + 4096, o2 = o0 + 8192, o3 = o0 + 12288;
    const bool f0 = o0 < full, f1 = o1 < full, f2 = o2 < full, f3 = o3 < full;
    const uint4 v0 = *reinterpret_cast<const uint4*>(s + (f0 ? o0 : 0));     // clamped: no load under a branch
    const uint4 v1 = *reinterpret_cast<const uint4*>(s + (f1 ? o1 : 0));
    const uint4 v2 = *reinterpret_cast<const uint4*>(s + (f2 ? o2 : 0));
    const uint4 v3 = *reinterpret_cast<const uint4*>(s + (f3 ? o3 : 0));
    if (f0) st16_wt(d + o0, v0);
    if (f1) st16_wt(d + o1, v1);
    if (f2) st16_wt(d + o2, v2);
    if (f3) st16_wt(d + o3, v3);
    // the last nb % 16 bytes: the lane whose piece would have held them
    if (full < nb && full >= base && full < base + kCopyChunk && (long long)threadIdx.x == ((full - base) >> 4) % 256)
      for (long long o = full; o < nb; ++o) d[o] = s[o];
  } else {
    for (int r = 0; r < 4; ++r) {
      const long long o0 = base + ((long long)r * 256 + threadIdx.x) * 16;
      for (long long o = o0; o < nb && o < o0 + 16; ++o) d[o] = s[o];
    }
  }
}
}  // namespace

extern "C" int bmnas_copy_batch_max(void) { return kCopyMax; }

extern "C" int bmnas_copy_blob_max(void) { return kBlobWords * 4; }

extern "C" int bmnas_copy_batch(const void* const* srcs, void* const* dsts, const long long* bytes, int n,
                                void* blob_dst, const void* blob, int blob_bytes, unsigned long long* add_dst,
                                unsigned long long add_val, void* stream) {
  if (n < 0 || n > kCopyMax || (n > 0 && (!srcs || !dsts || !bytes))) return BMNAS_E_ARG;
  if (blob_bytes < 0 || blob_bytes > kBlobWords * 4 || blob_bytes % 4 || (blob_bytes > 0 && (!blob_dst || !blob)))
    return BMNAS_E_ARG;
  CopyArgs a{};
  int g = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    if (bytes[i] < 0 || (bytes[i] > 0 && !dsts[i])) return BMNAS_E_ARG;      // (srcs[i] == NULL: zero-fill dsts[i])
    if (bytes[i] == 0) continue;
    a.src[m] = srcs[i]; a.dst[m] = dsts[i]; a.bytes[m] = bytes[i]; a.first[m] = g;
    g += (int)((bytes[i] + kCopyChunk - 1) / kCopyChunk);
    ++m;
  }
  a.n = m;
  for (int i = m; i <= kCopyMax; ++i) a.first[i] = g;
  if (blob_bytes > 0) {
    a.blob_dst = static_cast<unsigned int*>(blob_dst);
    a.blob_words = blob_bytes / 4;
    memcpy(a.blob, blob, (size_t)blob_bytes);
  }
  a.add_dst = add_dst;
  a.add_val = add_val;
  if (blob_bytes > 0 || add_dst != nullptr) ++g;                // one more workgroup: it stores the blob / advances the counter
  if (g == 0) return 0;
  hipLaunchKernelGGL(copy_batch_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, a);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
