// BatchNorm finalisation inside the CONSUMER of a conv output (no launch of its own).
//
// The forward GEMM accumulates, per output channel m, the batch sums of d = u - bias[m] and d^2 with
// fp32 atomics (conv1x1.hip: bn_tile_stats, `stat_shards` copies to spread same-address atomics).
// Every workgroup of the kernel that applies the BatchNorm (K2's mix, the out_conv / ConcatFC tail)
// turns them into the fused affine scale[m] = bn_w * rstd, shift[m] = bn_b - mean * scale in LDS at
// its start — four adjacent channels per thread, one memory round trip — and workgroup 0 also
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
// Thread t owns the four adjacent channels 4t .. 4t + 3 (M % 4 == 0, M <= 4 * BS): straight-line code,
// every load a float4 and all of them issued before the first use — ONE memory round trip (an
// earlier per-channel loop compiled into one load -> wait -> compute block per channel).
template <int BS>
__device__ __forceinline__ void bn_fin_fill(const BnFin& f, float* __restrict__ chan, const int M, const int N,
                                            float* sc, float* sh, const bool writer) {
  constexpr float kEpsBn = 1e-5f, kMom = 0.1f;
  constexpr int kMaxShards = 4;
  if (!f.on) {
    for (int m0 = 4 * (int)threadIdx.x; m0 < M; m0 += 4 * BS) {   // one trip for M <= 4 * BS
      const float4 a = ld4(chan + 2 * M + m0), b = ld4(chan + 3 * M + m0);
      st4(sc + m0, a);
      st4(sh + m0, b);
    }
    __syncthreads();
    return;
  }
  const bool wr = writer && !(f.on & 2);                      // (bit 1: timing experiments only)
  const bool upd = wr && f.training && f.running_mean != nullptr;
  for (int m0 = 4 * (int)threadIdx.x; m0 < M; m0 += 4 * BS) {   // one trip for M <= 4 * BS
    // EVERY load unconditional, from an always-valid address where its operand is absent (eval mode has no
    // sums, the conv may have no bias, training may not track running statistics): a load under `if` is a
    // branch, and hipcc copies the loaded registers at the join behind s_waitcnt — which put a full drain
    // between the sums and the affine parameters: two dependent round trips at the start of every consumer
    const bool tr = f.training != 0;
    const bool has_cb = tr && f.conv_bias != nullptr, need_run = upd || !tr;
    const int mv = m0 + 8 <= M ? m0 : (M >= 8 ? M - 8 : 0);    // (dummy reads stay inside bn_w)
    const float4* st = tr ? reinterpret_cast<const float4*>(f.stat + 2 * m0)
                          : reinterpret_cast<const float4*>(f.bn_w + mv);
    const int64_t sstr = tr ? (int64_t)(M / 2) : 0;
    float4 v[kMaxShards][2];
#pragma unroll
    for (int k = 0; k < kMaxShards; ++k) {                     // clamped shard index: no predicated loads
      const float4* p = st + (int64_t)(k < f.shards ? k : 0) * sstr;
      v[k][0] = p[0];                                          // (sum, sumsq) of channels m0, m0 + 1
      v[k][1] = p[1];                                          //              of channels m0 + 2, m0 + 3
    }
    const float4 w4 = ld4(f.bn_w + m0), b4 = ld4(f.bn_b + m0);
    float4 cb4 = ld4((has_cb ? f.conv_bias : f.bn_w) + m0);
    float4 rm4 = ld4((need_run ? f.running_mean : f.bn_w) + m0);
    float4 rv4 = ld4((need_run ? f.running_var : f.bn_w) + m0);
    long long nb = 0;
    const bool bump = wr && f.training && f.nbt != nullptr && m0 < f.n_nbt;
    if (bump) nb = f.nbt[m0];                                  // (n_nbt <= 4: the counters of thread 0)
    if (!has_cb) cb4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!need_run) {
      rm4 = make_float4(0.f, 0.f, 0.f, 0.f);
      rv4 = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    if (f.training) {
#pragma unroll
      for (int k = 0; k < kMaxShards; ++k) {
        const float on = k < f.shards ? 1.f : 0.f;
        s[0] += on * v[k][0].x; q[0] += on * v[k][0].y;
        s[1] += on * v[k][0].z; q[1] += on * v[k][0].w;
        s[2] += on * v[k][1].x; q[2] += on * v[k][1].y;
        s[3] += on * v[k][1].z; q[3] += on * v[k][1].w;
      }
    }
    const float wq[4] = {w4.x, w4.y, w4.z, w4.w}, bq[4] = {b4.x, b4.y, b4.z, b4.w};
    const float cq[4] = {cb4.x, cb4.y, cb4.z, cb4.w}, rmq[4] = {rm4.x, rm4.y, rm4.z, rm4.w};
    const float rvq[4] = {rv4.x, rv4.y, rv4.z, rv4.w};
    float mean[4], rstd[4], scale[4], shift[4], nrm[4], nrv[4];
    const float inv = 1.f / (float)N;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (f.training) {
        const float dm = s[j] * inv;
        const float var = fmaxf(q[j] * inv - dm * dm, 0.f);
        mean[j] = dm + cq[j];
        rstd[j] = 1.f / sqrtf(var + kEpsBn);
        nrm[j] = (1.f - kMom) * rmq[j] + kMom * mean[j];
        nrv[j] = (1.f - kMom) * rvq[j] + kMom * (var * (float)N / (float)(N - 1));
      } else {
        mean[j] = rmq[j];
        rstd[j] = 1.f / sqrtf(rvq[j] + kEpsBn);
        nrm[j] = nrv[j] = 0.f;
      }
      scale[j] = wq[j] * rstd[j];
      shift[j] = bq[j] - mean[j] * scale[j];
    }
    st4(sc + m0, make_float4(scale[0], scale[1], scale[2], scale[3]));
    st4(sh + m0, make_float4(shift[0], shift[1], shift[2], shift[3]));
    if (wr) {
      st4(chan + m0, make_float4(mean[0], mean[1], mean[2], mean[3]));
      st4(chan + M + m0, make_float4(rstd[0], rstd[1], rstd[2], rstd[3]));
      st4(chan + 2 * M + m0, make_float4(scale[0], scale[1], scale[2], scale[3]));
      st4(chan + 3 * M + m0, make_float4(shift[0], shift[1], shift[2], shift[3]));
      if (upd) {
        st4(f.running_mean + m0, make_float4(nrm[0], nrm[1], nrm[2], nrm[3]));
        st4(f.running_var + m0, make_float4(nrv[0], nrv[1], nrv[2], nrv[3]));
      }
      if (bump) {
        f.nbt[m0] = nb + 1;
        for (int j = 1; j < 4 && m0 + j < f.n_nbt; ++j) f.nbt[m0 + j] += 1;
      }
    }
  }
  __syncthreads();
}
