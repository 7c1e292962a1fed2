// The pooling in front of the reshape convs: AdaptiveMaxPool2d of every modality's backbone
// feature map, for the whole group of modalities in ONE launch per direction, written straight in the (b, C_in, L)
// layout the grouped GEMM reads.
//   ReshapeInputLayer        (aux_models.py:62-70):   (b, C_in, T, ...) -> view (b, C_in, T, R) -> pool to (L, 1)
//                                                      (-> F.interpolate(., L): nearest, an identity at size L)
//   ReshapeInputLayer_MMIMDB (aux_models.py:101-108): (b, C_in[, H, W]) -> view (b, C_in, H, W) -> pool to (s, s), s*s = L
// Window of output row i: [floor(i H / oh), ceil((i + 1) H / oh)), columns alike (torch's adaptive rule); the first
// maximum in row-major window order wins, a NaN is a maximum (at::native adaptive_max_pool2d).  The argmax (flat h*W+w,
// int32) is kept for the backward, which is written in GATHER form: one thread per INPUT element sums the gradients
// of the <= 2 x 2 windows it belongs to and whose argmax it is — every dx element is written exactly once, so there
// is neither a zero-fill launch nor an atomic (windows overlap when H % oh != 0).
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr int kPoolGroup = BMNAS_MAX_GROUP;

struct PoolProb {
  const float* x;      // (b, C, H, W)
  float* out;          // (b, C, oh * ow)
  int* idx;            // (b, C, oh * ow), nullable in forward
  const float* g;      // backward: gradient of out
  float* dx;           // backward: (b, C, H, W)
  int C, H, W, oh, ow, G;          // G = lanes per output element (power of two <= 64)
};
struct PoolGroup {
  PoolProb p[kPoolGroup];
  long long start[kPoolGroup + 1];  // first block of problem q
  int n;
};

__device__ __forceinline__ int pool_problem(const PoolGroup& S) {
  int q = 0;
#pragma unroll
  for (int r = 1; r < kPoolGroup; ++r) q += (r < S.n && (long long)blockIdx.x >= S.start[r]) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(q);
}

// (chains over the compile-time-indexed elements: a run-time index sends the by-value argument struct through scratch)
__device__ __forceinline__ PoolProb pool_pick(const PoolGroup& S, int q, long long* start) {
  PoolProb P = S.p[0];
  *start = S.start[0];
#pragma unroll
  for (int r = 1; r < kPoolGroup; ++r)
    if (q == r) { P = S.p[r]; *start = S.start[r]; }
  return P;
}

__global__ __launch_bounds__(256) void pool_fwd_group_k(PoolGroup S, int b) {
  long long st;
  const PoolProb P = pool_pick(S, pool_problem(S), &st);
  const int G = P.G;
  const long long n_out = (long long)b * P.C * P.oh * P.ow;
  const long long o = (((long long)blockIdx.x - st) * 256 + threadIdx.x) / G;       // this lane group's output element
  const int gl = threadIdx.x & (G - 1);
  const bool on = o < n_out;
  const long long oc = on ? o : n_out - 1;
  const int j = (int)(oc % P.ow), i = (int)((oc / P.ow) % P.oh);
  const long long bc = oc / ((long long)P.oh * P.ow);
  const int hs = (i * P.H) / P.oh, he = ((i + 1) * P.H + P.oh - 1) / P.oh;
  const int ws = (j * P.W) / P.ow, we = ((j + 1) * P.W + P.ow - 1) / P.ow;
  const float* __restrict__ src = P.x + bc * P.H * P.W;
  float best = -__builtin_inff();
  int bi = hs * P.W + ws;
  bool have = false;
  for (int h = hs; h < he; ++h)
    for (int w = ws + gl; w < we; w += G) {
      const float v = src[(long long)h * P.W + w];
      const bool nan = v != v;
      if (!have || nan || v > best) {                   // first maximum in this lane's (row-major) order; NaN wins
        if (!(have && best != best)) { best = v; bi = h * P.W + w; }
        have = true;
      }
    }
  // combine the G lanes: larger value wins, a NaN beats everything, ties go to the smaller flat index (= the
  // first in row-major window order)
  for (int off = G >> 1; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    const bool oh_ = __shfl_xor((int)have, off, 64) != 0;
    const bool mine_nan = have && best != best, other_nan = oh_ && ov != ov;
    bool take = false;
    if (oh_ && !have) take = true;
    else if (oh_ && have) {
      if (mine_nan && other_nan) take = oi < bi;
      else if (other_nan) take = true;
      else if (mine_nan) take = false;
      else take = (ov > best) || (ov == best && oi < bi);
    }
    if (take) { best = ov; bi = oi; have = true; }
  }
  if (on && gl == 0) {
    P.out[o] = best;
    if (P.idx != nullptr) P.idx[o] = bi;
  }
}

__global__ __launch_bounds__(256) void pool_bwd_group_k(PoolGroup S, int b) {
  long long st;
  const PoolProb P = pool_pick(S, pool_problem(S), &st);
  const long long n_in = (long long)b * P.C * P.H * P.W;
  const long long e = ((long long)blockIdx.x - st) * 256 + threadIdx.x;
  if (e >= n_in) return;
  const int w = (int)(e % P.W), h = (int)((e / P.W) % P.H);
  const long long bc = e / ((long long)P.H * P.W);
  const int i0 = (h * P.oh) / P.H, i1 = ((h + 1) * P.oh + P.H - 1) / P.H - 1;
  const int j0 = (w * P.ow) / P.W, j1 = ((w + 1) * P.ow + P.W - 1) / P.W - 1;
  const int me = h * P.W + w;
  const long long ob = bc * P.oh * P.ow;
  float acc = 0.f;
  for (int i = i0; i <= i1; ++i)
    for (int j = j0; j <= j1; ++j) {
      const long long o = ob + (long long)i * P.ow + j;
      if (P.idx[o] == me) acc += P.g[o];
    }
  P.dx[e] = acc;
}

int fill_group(PoolGroup& S, const bmnas_pool_prob_t* probs, int n, int b, bool backward, long long* blocks) {
  if (!probs || n < 1 || b < 0) return BMNAS_E_ARG;
  if (n > kPoolGroup) return BMNAS_E_LIMIT;
  S.n = n;
  long long tot = 0;
  for (int q = 0; q < n; ++q) {
    const bmnas_pool_prob_t& p = probs[q];
    if (p.C < 1 || p.H < 1 || p.W < 1 || p.oh < 1 || p.ow < 1) return BMNAS_E_ARG;
    if ((long long)p.H * p.W >= (1ll << 31)) return BMNAS_E_LIMIT;
    if (backward ? (!p.g || !p.idx || !p.dx) : (!p.x || !p.out)) return BMNAS_E_ARG;
    PoolProb& P = S.p[q];
    P.x = p.x; P.out = p.out; P.idx = p.idx; P.g = p.g; P.dx = p.dx;
    P.C = p.C; P.H = p.H; P.W = p.W; P.oh = p.oh; P.ow = p.ow;
    const int ww = (p.W + p.ow - 1) / p.ow;            // window width (within one of the adaptive rule)
    int G = 1;
    while (G * 2 <= ww && G < 64) G *= 2;
    P.G = G;
    S.start[q] = tot;
    const long long work = backward ? (long long)b * p.C * p.H * p.W : (long long)b * p.C * p.oh * p.ow * G;
    tot += (work + 255) / 256;
  }
  S.start[n] = tot;
  if (tot >= (1ll << 31)) return BMNAS_E_LIMIT;
  *blocks = tot;
  return 0;
}

}  // namespace

extern "C" int bmnas_adaptive_maxpool_fwd_group(const bmnas_pool_prob_t* probs, int n, int b, void* stream) {
  PoolGroup S{};
  long long blocks = 0;
  if (int e = fill_group(S, probs, n, b, false, &blocks)) return e;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(pool_fwd_group_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, b);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_adaptive_maxpool_bwd_group(const bmnas_pool_prob_t* probs, int n, int b, void* stream) {
  PoolGroup S{};
  long long blocks = 0;
  if (int e = fill_group(S, probs, n, b, true, &blocks)) return e;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(pool_bwd_group_k, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, S, b);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
