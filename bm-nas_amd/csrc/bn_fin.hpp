// BatchNorm finalisation inside the CONSUMER of a conv output (no launch of its own).
//
// The forward GEMM accumulates, per output channel m, the batch sums of d = u - bias[m] and d^2 with
// fp32 atomics (conv1x1.hip: bn_tile_stats, `stat_shards` copies to spread same-address atomics).
// Every workgroup of the kernel that applies the BatchNorm (K2's mix, the out_conv / ConcatFC tail)
// turns them into the fused affine scale[m] = bn_w * rstd, shift[m] = bn_b - mean * scale in LDS at
// its start — M <= 2304 channels, a handful of loads and one rsqrt per thread — and workgroup 0 also
// writes chan = mean | rstd | scale | shift for the backward pass and updates the running
// statistics exactly like nn.BatchNorm1d (momentum 0.1, unbiased variance, num_batches_tracked).
// Eval mode: the same from the running statistics.  Replaces bmnas_bn_finalize on the search path
// (reference: nn.BatchNorm1d at node_operations.py:26,34 / :45,53, node_search.py:40,62).
#pragma once
#include "common.hpp"

struct BnFin {
  const float* stat;       // (shards, M, 2); unused in eval mode
  const float* conv_bias;  // shift of the sums (nullable: 0)
  const float* bn_w;
  const float* bn_b;
  float* running_mean;     // nullable in training mode (no update); read in eval mode
  float* running_var;
  long long* nbt;          // n_nbt consecutive int64 counters, nullable
  int shards, n_nbt, training;
  int on;                  // 0: `chan` already holds the finalised values (bmnas_bn_finalize ran)
};

// sc / sh: LDS, M floats each.  Ends with a __syncthreads().
template <int BS>
__device__ __forceinline__ void bn_fin_fill(const BnFin& f, float* __restrict__ chan, const int M, const int N,
                                            float* sc, float* sh, const bool writer) {
  constexpr float kEpsBn = 1e-5f, kMom = 0.1f;
  constexpr int kMaxShards = 8, kCh = 3;
  if (!f.on) {
    for (int m = threadIdx.x; m < M; m += BS) {
      sc[m] = chan[2 * M + m];
      sh[m] = chan[3 * M + m];
    }
    __syncthreads();
    return;
  }
  const bool upd = writer && f.training && f.running_mean != nullptr;
  for (int base = 0; base < M; base += kCh * BS) {
    // every load of the (up to kCh) channels this thread owns first — shard sums, affine, running
    // statistics; addresses are clamped, not predicated (a predicated load compiles to a branch and a
    // wait per load) — then the arithmetic: ONE memory round trip per kCh * BS channels.
    // (The sums were written by memory-side atomics: these loads miss L2.)
    float2 v[kCh][kMaxShards];
    float w[kCh], bb[kCh], cb[kCh], rm0[kCh], rv0[kCh];
#pragma unroll
    for (int i = 0; i < kCh; ++i) {
      const int m = base + (int)threadIdx.x + i * BS;
      const int mc = m < M ? m : M - 1;
      if (f.training) {
        const float2* st = reinterpret_cast<const float2*>(f.stat) + mc;
#pragma unroll
        for (int k = 0; k < kMaxShards; ++k) v[i][k] = st[(int64_t)(k < f.shards ? k : 0) * M];
      }
      w[i] = f.bn_w[mc];
      bb[i] = f.bn_b[mc];
      cb[i] = (f.training && f.conv_bias != nullptr) ? f.conv_bias[mc] : 0.f;
      rm0[i] = 0.f;
      rv0[i] = 1.f;
      if (upd || !f.training) {
        rm0[i] = f.running_mean[mc];
        rv0[i] = f.running_var[mc];
      }
    }
#pragma unroll
    for (int i = 0; i < kCh; ++i) {
      const int m = base + (int)threadIdx.x + i * BS;
      if (m >= M) continue;
      float mean, rstd;
      if (f.training) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxShards; ++k) {
          s += (k < f.shards) ? v[i][k].x : 0.f;
          q += (k < f.shards) ? v[i][k].y : 0.f;
        }
        const float inv = 1.f / (float)N;
        const float dm = s * inv;
        const float var = fmaxf(q * inv - dm * dm, 0.f);
        mean = dm + cb[i];
        rstd = 1.f / sqrtf(var + kEpsBn);
        if (upd) {
          f.running_mean[m] = (1.f - kMom) * rm0[i] + kMom * mean;
          f.running_var[m] = (1.f - kMom) * rv0[i] + kMom * (var * (float)N / (float)(N - 1));
        }
        if (writer && m < f.n_nbt && f.nbt != nullptr) f.nbt[m] += 1;
      } else {
        mean = rm0[i];
        rstd = 1.f / sqrtf(rv0[i] + kEpsBn);
      }
      const float scale = w[i] * rstd;
      const float shift = bb[i] - mean * scale;
      sc[m] = scale;
      sh[m] = shift;
      if (writer) {
        chan[m] = mean;
        chan[M + m] = rstd;
        chan[2 * M + m] = scale;
        chan[3 * M + m] = shift;
      }
    }
  }
  __syncthreads();
}
