// K4 / K5 / out_conv — channel-concat + 1x1 Conv1d as fp32-MFMA GEMMs (v_mfma_f32_16x16x4_f32,
// exact fp32).  Activations are (b, C, L) with L in {4, 8, 16} contiguous, so the natural
// 16-wide MFMA dimension is one "n-group" = 16 consecutive (sample, l) columns
// (= 16/L samples).  All three GEMMs feed the matrix cores straight from global/L2 loads
// in MFMA-operand layout (no LDS round trip, no barriers in the main loops):
//   * activation as the "A" operand, contraction over channels: lane (n = lane&15,
//     slot = lane>>4) reads channel i0 + 4*slot + r for step r  -> 64-B row segments;
//   * weight rows as float4 (k-permuted: step r <-> k = k0 + 4*slot + r);
//   * weight-gradient GEMM contracts over n, where BOTH operands are float4 along l.
// The k-permutation is legal because A and B use the same one.
// FLOPs: 2*M*K*b*L each; bound: fp32 MFMA (157 TFLOP/s dense).
#include "sdpa_body.hpp"
#ifndef BMNAS_FWD_LA2
#define BMNAS_FWD_LA2 1
#endif
constexpr bool kFwdLa2 = BMNAS_FWD_LA2 != 0;
#include <algorithm>
#include <cstdlib>

namespace {

// At most 4 tensors are ever concatenated in front of a conv (x, y | node_multiplier <= 4
// states); the pointer lists are kept that short on purpose — the kernel-argument block is
// fetched by every wave, and a 2 x 16-pointer block measurably slowed the launches.
constexpr int kConvPtrs = 4;
struct ConvIn { const float* p[kConvPtrs]; };
struct ConvOut { float* p[kConvPtrs]; };

struct ConvArgs {
  ConvIn act;            // n_act sources, each (b, Ci, L): contraction rows i = q*Ci + ci
  ConvOut dst;           // n_dst destinations, each (b, Cj, L): output cols j = q*Cj + cj
  const float* W;
  const float* bias;     // per output col (fwd), nullable
  float* part;           // BN partial stats (fwd, training), nullable
  float* stat;           // alternative to part: stat_shards x J x 2 running sums (atomics), nullable
  int stat_shards;
  // data gradient with the BatchNorm input gradient folded into the operand staging: act.p[0] then
  // holds dV (gradient w.r.t. the BatchNorm OUTPUT) and the tile kernel forms
  //   dU = scale * (dV - kb - (U - mean) * rstd * kw),   kw = bn_grad[m] / N, kb = bn_grad[I + m] / N
  // (training; eval: dU = scale * dV) while it moves the chunk into LDS — no bn_bwd_apply launch
  const float* bn_U;     // conv output (b, I, L), nullable = no fold
  const float* bn_chan;  // mean | rstd | scale | shift
  const float* bn_grad;  // dBN.weight | dBN.bias
  int bn_train;
  int ldw, Ci, Cj, I, J;
  int b, L, Lb, spw, n_groups, n_part;
  uint32_t acc_mask;
  int fold;              // > 0: effective weight = W[.] + W[. + fold] along its column index
                         // (conv applied to cat[z, z], search mode: no separate folded copy)
  int probe;             // diagnostics only (BMNAS_CONV_PROBE): 1 = no MFMA, 2 = no loads
  int order;             // conv_bwd_all_pipe_k: block order of its three classes (see the kernel)
};

// BatchNorm batch statistics of the tile column group g (<= 16 valid columns) for output channel jj,
// from the epilogue registers o (4 columns per lane, the channel's 16 columns spread over h = 0..3).
//  * a.part:  one (sum, centred second moment) partial per (channel, n-group), combined later by
//             bmnas_bn_finalize with Chan's rule;
//  * a.stat:  running sums of d = u - bias and d^2 per channel, added with fp32 atomics into shard
//             g % stat_shards (same-address atomics serialise: the shards spread them); the CONSUMER
//             of the conv output turns them into mean / rstd itself (bn_fin.hpp), so no launch sits
//             between the GEMM and the BatchNorm apply.  Shifting by the bias keeps E[d^2] - E[d]^2
//             free of the cancellation a large bias would cause.
__device__ __forceinline__ void bn_tile_stats(const ConvArgs& a, const float4 o, const float bj, const bool vo,
                                              const int g, const int jj, const int h) {
  if (a.stat != nullptr) {
    float sum = 0.f, sq = 0.f;
    if (vo) {
      const float4 d = make_float4(o.x - bj, o.y - bj, o.z - bj, o.w - bj);
      sum = f4_hsum(d);
      sq = f4_dot(d, d);
    }
    sum = xor16_sum(sum);
    sum = xor32_sum(sum);
    sq = xor16_sum(sq);
    sq = xor32_sum(sq);
    if (h == 0) {
      float* pp = a.stat + ((int64_t)(g % a.stat_shards) * a.J + jj) * 2;
      atomicAdd(pp, sum);
      atomicAdd(pp + 1, sq);
    }
    return;
  }
  if (a.part == nullptr) return;
  float sum = vo ? f4_hsum(o) : 0.f;
  sum = xor16_sum(sum);
  sum = xor32_sum(sum);
  int cnt = a.b * a.L - 16 * g;
  cnt = cnt > 16 ? 16 : cnt;
  const float mean = sum / (float)cnt;
  float m2 = 0.f;
  if (vo) {
    const float4 c = make_float4(o.x - mean, o.y - mean, o.z - mean, o.w - mean);
    m2 = f4_dot(c, c);
  }
  m2 = xor16_sum(m2);
  m2 = xor32_sum(m2);
  if (h == 0) {
    float* pp = a.part + ((int64_t)jj * a.n_part + g) * 2;
    pp[0] = sum;
    pp[1] = m2;
  }
}

// ---- split-K variant: ONE memory round trip per wave ------------------------------------------
// At these sizes (0.1-0.5 GFLOP per GEMM, every operand L2-resident) the kernels are bound
// by exposed load latency, not by MFMA or bandwidth: a wave that alternates "load a slice /
// multiply a slice" pays the (loaded) L2 latency once per slice.  Here the four waves of a
// workgroup split the contraction dimension of ONE (16*TN x 16*TJ) output tile, each wave
// issues ALL of its operand loads up front (KPW blocks of 16 channels = KPW*4*(TN+TJ) VGPRs),
// waits once, runs its MFMAs back to back, and the four partial tiles are summed through
// LDS.  Splitting K four ways also quadruples the number of waves, which is what hides the
// latency at batch 128.
// ---- K2 backward as the epilogue of a data-gradient tile (small grids) -------------------------------------
// NodeCell's out_conv reads cat(states[-node_multiplier:]) (node_search.py:59-61); the last of those states is the
// output of the last inner step's NodeMixedOp and feeds nothing else, so the out_conv data gradient for its
// channels IS the complete gradient of that mixed output.  The tile that has just formed it applies the whole
// NodeMixedOp backward of bmnas_node_mix_bwd to its 16 (sample, l) x 16 channel elements (x is y: search mode)
// instead of a launch of its own reading it back: the gamma-weighted gradients of the four primitives'
// BatchNorm outputs (dV), the BatchNorm reductions, dgamma and the gradient of the op's input.
struct MixEp {
  const float* U;          // (b, 3C, L) stacked conv output of the mixed op
  const float* chan;       // its mean | rstd | scale | shift (M = 3C)
  const float* x;          // its input (b, C, L)
  const float* p1;         // attention branch output
  const float* gamma;      // 4 softmaxed weights
  float* dgamma;           // += over dgamma_shards copies
  float* dx;               // gradient of the input: (=|+=) 2 g0 g
  float* dV;               // (b, 3C, L)
  float* bn_grad;          // (6C) +=
  int64_t dg_stride;
  int dg_shards, acc_dx, q, on;
  DropCfg dglu, dfc;
};

__device__ __forceinline__ float sigmoid_ep(float v) { return 1.f / (1.f + __expf(-v)); }

// lane = (channel cj = .. + lo, four consecutive l of sample `so`); gv = gradient of the mixed output there
// (zero for padded samples: vo false).  All 64 lanes of the wave call it (shuffles).
__device__ __forceinline__ void mix_ep_tile(const MixEp& m, const float4 gv, const int so, const int cj,
                                            const int l0, const bool vo, const int C, const int L,
                                            const int shard) {
  const int M = 3 * C;
  const int sc_ = vo ? so : 0;
  const int64_t e = ((int64_t)sc_ * C + cj) * L + l0;
  const int64_t ub = ((int64_t)sc_ * M + cj) * L + l0;
  const float4 ua = ld4(m.U + ub), ug = ld4(m.U + ub + (int64_t)C * L), uf = ld4(m.U + ub + (int64_t)2 * C * L);
  const float4 xv = ld4(m.x + e), pv = ld4(m.p1 + e);
  float4 oldx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m.acc_dx) oldx = ld4(m.dx + e);
  float mu[3], rs[3], sc[3], sh[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    mu[k] = m.chan[k * C + cj];
    rs[k] = m.chan[M + k * C + cj];
    sc[k] = m.chan[2 * M + k * C + cj];
    sh[k] = m.chan[3 * M + k * C + cj];
  }
  const float g0 = m.gamma[0], g2 = m.gamma[2], g3 = m.gamma[3];
  const DropRt rglu = drop_begin(m.dglu), rfc = drop_begin(m.dfc);
  const float4 m2 = drop_mult4(rglu, (uint64_t)e), m3 = drop_mult4(rfc, (uint64_t)e);
  const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
  const float uaq[4] = {ua.x, ua.y, ua.z, ua.w}, ugq[4] = {ug.x, ug.y, ug.z, ug.w}, ufq[4] = {uf.x, uf.y, uf.z, uf.w};
  const float xq[4] = {2.f * xv.x, 2.f * xv.y, 2.f * xv.z, 2.f * xv.w};       // x + y, x is y
  const float pq[4] = {pv.x, pv.y, pv.z, pv.w};
  const float m2q[4] = {m2.x, m2.y, m2.z, m2.w}, m3q[4] = {m3.x, m3.y, m3.z, m3.w};
  float da[4], dg[4], df[4], dgam[4] = {0.f, 0.f, 0.f, 0.f}, sw[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t) {                                // the arithmetic of node_mix_bwd_k, term for term
    const float va = fmaf(uaq[t], sc[0], sh[0]), vg = fmaf(ugq[t], sc[1], sh[1]), vf = fmaf(ufq[t], sc[2], sh[2]);
    const float sg = sigmoid_ep(vg);
    dgam[0] += gq[t] * xq[t];
    dgam[1] += gq[t] * pq[t];
    dgam[2] += gq[t] * (va * sg * m2q[t]);
    dgam[3] += gq[t] * (fmaxf(vf, 0.f) * m3q[t]);
    const float gm2 = g2 * gq[t] * m2q[t];
    da[t] = gm2 * sg;
    dg[t] = gm2 * va * sg * (1.f - sg);
    df[t] = (vf > 0.f) ? g3 * gq[t] * m3q[t] : 0.f;
    sw[0] += da[t] * (uaq[t] - mu[0]) * rs[0];
    sw[1] += dg[t] * (ugq[t] - mu[1]) * rs[1];
    sw[2] += df[t] * (ufq[t] - mu[2]) * rs[2];
    sb[0] += da[t]; sb[1] += dg[t]; sb[2] += df[t];
  }
  if (vo) {
    st4(m.dV + ub, make_float4(da[0], da[1], da[2], da[3]));
    st4(m.dV + ub + (int64_t)C * L, make_float4(dg[0], dg[1], dg[2], dg[3]));
    st4(m.dV + ub + (int64_t)2 * C * L, make_float4(df[0], df[1], df[2], df[3]));
    st4(m.dx + e, f4_add(f4_scale(gv, 2.f * g0), oldx));      // dy == NULL: both halves into dx
  }
  // BatchNorm reductions over the tile's 16 columns: the four lanes h of a channel (xor 16 / 32), then one
  // atomic pair per channel and primitive; dgamma over the wave, one atomic each into this tile's shard
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    sw[k] = xor16_sum(sw[k]); sw[k] = xor32_sum(sw[k]);
    sb[k] = xor16_sum(sb[k]); sb[k] = xor32_sum(sb[k]);
  }
  const int lane = threadIdx.x & 63;
  if ((lane >> 4) == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      atomicAdd(m.bn_grad + k * C + cj, sw[k]);
      atomicAdd(m.bn_grad + M + k * C + cj, sb[k]);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) dgam[q] = wave_sum(dgam[q]);
  if (lane == 0 && m.dgamma != nullptr) {
    float* p = m.dgamma + (int64_t)(shard % m.dg_shards) * m.dg_stride;
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(p + q, dgam[q]);
  }
}

template <int TN, int TJ>
constexpr size_t conv_ksplit_lds();

// Phase probes INSIDE the split-K body (BMNAS_CONV_PROBE bits 1 = no MFMA, 2 = no loads, 4 = no stores,
// 8 = no barrier) exist only in builds with -DBMNAS_BODY_PROBES=1: left in as runtime branches they sit
// between the operand loads, and hipcc then ends each block's loads with a branch and a wait.
#ifndef BMNAS_BODY_PROBES
#define BMNAS_BODY_PROBES 0
#endif
#define KS_PROBE(a, bit) (BMNAS_BODY_PROBES && ((a).probe & (bit)))

// Wave priorities of the tile classes that share a merged launch (s_setprio: the SIMD's issue arbiter prefers the
// higher-priority wave when several are ready).  0 = leave the wave at its launch priority.
#ifndef BMNAS_PRIO_BWD_A
#define BMNAS_PRIO_BWD_A 0
#endif
#ifndef BMNAS_PRIO_BWD_W
#define BMNAS_PRIO_BWD_W 0
#endif
#ifndef BMNAS_PRIO_BWD_D
#define BMNAS_PRIO_BWD_D 0
#endif
#ifndef BMNAS_PRIO_FWD_A
#define BMNAS_PRIO_FWD_A 0
#endif
#ifndef BMNAS_PRIO_FWD_G
#define BMNAS_PRIO_FWD_G 0
#endif
#define BMNAS_SETPRIO(p)                                  \
  do {                                                    \
    if constexpr ((p) != 0) __builtin_amdgcn_s_setprio(p); \
  } while (0)

// MULTI: contractions longer than the 4 * KPW blocks a workgroup holds in registers at once (the K = 2048
// reshape layers) run as several rounds of [all loads, then all MFMAs] into the same accumulators.
template <bool TRANS, int TN, int TJ, int KPW, bool MULTI = false, bool MIXEP = false>
__device__ __forceinline__ void conv_ksplit_body(const ConvArgs& a, const int bx, const int by, char* lds,
                                                 const MixEp* mix = nullptr) {
  // caller-provided LDS (conv_ksplit_lds<TN, TJ>() bytes): merged launches pay max(), not sum()
  float4 (*part)[TN * TJ][64] = reinterpret_cast<float4 (*)[TN * TJ][64]>(lds);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int g0 = bx * TN;
  const int j0 = by * (16 * TJ);

  int64_t abase[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int g = g0 + tn;
    g = g < a.n_groups ? g : a.n_groups - 1;             // clamped, never stored
    int s = g * a.spw + (lo >> a.Lb);
    s = s < a.b ? s : a.b - 1;
    abase[tn] = ((int64_t)s * a.Ci) * a.L + (lo & (a.L - 1));
  }
  int jcl[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int jt = j0 + 16 * tj;
    jcl[tj] = (jt < a.J ? jt : a.J - 16) + lo;
  }

  const int nblk = a.I / 16;
  // BatchNorm-backward fold (data gradient only, see ConvArgs): the raw conv outputs travel with dV
  const bool fold_bn = !TRANS && !MULTI && a.bn_U != nullptr;
  f32x4 acc[TN][TJ];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) acc[tn][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < (MULTI ? nblk : 1); base += 4 * KPW) {
  float av[KPW][TN][4], bv[KPW][TJ][4], uv[KPW][TN][4];
#pragma unroll
  for (int kb = 0; kb < KPW; ++kb) {
    const int blk = base + wave * KPW + kb;
    const bool vb = blk < nblk;                          // wave-uniform
    const int i0 = (vb ? blk : nblk - 1) * 16;
    // MULTI: one source, no weight fold, no probes (host-checked) — nothing wave-uniform-but-runtime is left
    // inside the round loop, where hipcc would turn it into a branch + s_waitcnt per block
    const int q = MULTI ? 0 : i0 / a.Ci;
    const int ci = i0 - q * a.Ci + 4 * h;
    // a select chain, not a.act.p[q]: indexing the kernel-argument array with a run-time q is a MEMORY load
    // of the pointer (global_load_dwordx2 + s_waitcnt vmcnt(0)) per block — every block's operand loads
    // then wait for the previous block's, one round trip per block instead of one per wave
    const float* src = pick_ptr(a.act.p, q);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const float* pp = src + abase[tn] + (int64_t)ci * a.L;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = (!MULTI && KS_PROBE(a, 2)) ? (float)lane : pp[(int64_t)r * a.L];
        av[kb][tn][r] = vb ? t : 0.f;                    // blocks past the end contribute zero
        if (fold_bn && a.bn_train) uv[kb][tn][r] = a.bn_U[abase[tn] + (int64_t)(ci + r) * a.L];
      }
    }
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      if (!MULTI && KS_PROBE(a, 2)) {
        bv[kb][tj][0] = bv[kb][tj][1] = bv[kb][tj][2] = bv[kb][tj][3] = (float)lo;
      } else if (TRANS) {
        const float* pp = a.W + (int64_t)jcl[tj] * a.ldw + i0 + 4 * h;
        float4 w4 = ld4(pp);
        if (!MULTI && a.fold > 0) w4 = f4_add(w4, ld4(pp + a.fold));   // wave-uniform; W is L2-resident
        bv[kb][tj][0] = w4.x; bv[kb][tj][1] = w4.y; bv[kb][tj][2] = w4.z; bv[kb][tj][3] = w4.w;
      } else {
        const float* pp = a.W + (int64_t)(i0 + 4 * h) * a.ldw + jcl[tj];
        if (!MULTI && a.fold > 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[kb][tj][r] = pp[(int64_t)r * a.ldw] + pp[(int64_t)r * a.ldw + a.fold];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[kb][tj][r] = pp[(int64_t)r * a.ldw];
        }
      }
    }
  }
  if (fold_bn) {
    // per-channel (scale, kb, mean, rstd * kw) of all I channels through LDS (behind the partial tiles),
    // then dU = scale * (dV - kb - (U - mean) * rstd * kw) on the operand registers
    float4* coef = reinterpret_cast<float4*>(lds + conv_ksplit_lds<TN, TJ>());
    const float invN = 1.f / (float)(a.b * a.L);
    for (int m = threadIdx.x; m < a.I; m += 256) {
      const float sc = a.bn_chan[2 * a.I + m];
      coef[m] = a.bn_train ? make_float4(sc, a.bn_grad[a.I + m] * invN, a.bn_chan[m],
                                         a.bn_chan[a.I + m] * (a.bn_grad[m] * invN))
                           : make_float4(sc, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < KPW; ++kb) {
      const int blk = base + wave * KPW + kb;
      if (blk >= nblk) continue;                         // wave-uniform (those operands are zero)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 cf = coef[blk * 16 + 4 * h + r];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          av[kb][tn][r] = a.bn_train ? cf.x * (av[kb][tn][r] - cf.y - (uv[kb][tn][r] - cf.z) * cf.w)
                                     : cf.x * av[kb][tn][r];
      }
    }
  }
  // keep every load above this line: without the fence hipcc sinks the loads next to their
  // MFMAs to save registers, which re-serialises load -> wait -> multiply per block
  __builtin_amdgcn_sched_barrier(0);
  if (!MULTI && KS_PROBE(a, 1)) {
#pragma unroll
    for (int kb = 0; kb < KPW; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj) acc[tn][tj][r] += av[kb][tn][r] + bv[kb][tj][r];
  } else {
#pragma unroll
    for (int kb = 0; kb < KPW; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
            acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][tn][r], bv[kb][tj][r], acc[tn][tj], 0, 0, 0);
  }
  if (MULTI) __builtin_amdgcn_sched_barrier(0);
  }  // rounds

#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
      part[wave][tn * TJ + tj][lane] = make_float4(acc[tn][tj][0], acc[tn][tj][1], acc[tn][tj][2], acc[tn][tj][3]);
  if (!KS_PROBE(a, 8)) __syncthreads();

  // wave w finishes the tiles t = w, w + 4, ...:  o[r] = OUT[n = 16*g + 4h + r][j]
  const int l0 = (4 * h) & (a.L - 1);
  for (int t = wave; t < TN * TJ; t += 4) {
    const int tn = t / TJ, tj = t - tn * TJ;
    const int g = g0 + tn, jt = j0 + 16 * tj;
    if (g >= a.n_groups || jt >= a.J) continue;          // wave-uniform
    const float4 p0 = part[0][t][lane], p1 = part[1][t][lane], p2 = part[2][t][lane], p3 = part[3][t][lane];
    const int jj = jt + lo;
    const float bj = (a.bias != nullptr) ? a.bias[jj] : 0.f;
    const float4 o = make_float4((p0.x + p1.x) + (p2.x + p3.x) + bj, (p0.y + p1.y) + (p2.y + p3.y) + bj,
                                 (p0.z + p1.z) + (p2.z + p3.z) + bj, (p0.w + p1.w) + (p2.w + p3.w) + bj);
    const int q = jj / a.Cj;
    const int cj = jj - q * a.Cj;
    // (pick_ptr, not a hand-written select chain: LLVM folds select(load, load) of kernel-argument pointers back into ONE
    // indexed load — a global_load_dwordx2 from the kernarg segment + s_waitcnt vmcnt(0) in front of the tile's store, and
    // in the grouped kernels, whose descriptor is a local copy, 200 B of scratch per lane; round 5, by the ISA)
    float* d = pick_ptr(a.dst.p, q);
    const int so = g * a.spw + ((4 * h) >> a.Lb);
    const bool vo = so < a.b;
    float4 ov = o;
    if (vo && d != nullptr && (!KS_PROBE(a, 4) || o.x == 12345.f)) {
      float* pp = d + ((int64_t)so * a.Cj + cj) * a.L + l0;
      if (a.acc_mask & (1u << q)) ov = f4_add(o, ld4(pp));
      st4_wtg<4>(pp, ov);
    }
    if (MIXEP && q == mix->q)                                  // wave-uniform: a 16-channel tile lies in one source
      mix_ep_tile(*mix, vo ? ov : make_float4(0.f, 0.f, 0.f, 0.f), so, cj, l0, vo, a.Cj, a.L,
                  by * (int)gridDim.x + bx);
    bn_tile_stats(a, o, bj, vo, g, jj, h);
  }
}

template <int TN, int TJ>
constexpr size_t conv_ksplit_lds() { return (size_t)4 * TN * TJ * 64 * sizeof(float4); }

template <bool TRANS, int TN, int TJ, int KPW>
__global__ __launch_bounds__(256) void conv_ksplit_k(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[conv_ksplit_lds<TN, TJ>()];
  conv_ksplit_body<TRANS, TN, TJ, KPW>(a, blockIdx.x, blockIdx.y, lds);
}

// long contractions at small grids (C_in = 2048 reshape layers at <= 64 samples per GPU): rounds of 12
// blocks per wave instead of the whole-K LDS kernel, whose 64 workgroups walked K serially (44.7 us for
// NTU b = 64, 3 % of the MFMA peak)
template <bool TRANS>
__global__ __launch_bounds__(256) void conv_ksplit_multi_k(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[conv_ksplit_lds<1, 1>()];
  conv_ksplit_body<TRANS, 1, 1, 12, true>(a, blockIdx.x, blockIdx.y, lds);
}

// BatchNorm input gradient  dU = scale * (dV - kb - (U - mean) * kw)  (kb = sum dV / N, kw = rstd * sum dV u_hat / N)
// as  dU = alpha * dV + beta * U + gamma:  the pipelined data-gradient tiles and the weight-gradient tiles
// apply it to every operand element they stage, two FMAs instead of five operations each.
__device__ __forceinline__ float4 bn_fold_coef(float scale, float kb, float mean, float kw) {
  return make_float4(scale, -scale * kw, scale * (kw * mean - kb), 0.f);
}

// ---- pipelined LDS tile kernel (forward) -----------------------------------------------------------
// Counters (profiles/r01_pmc_gemm.txt) show the split-K kernel spending 45 % of a wave's life in one
// burst of operand fetches (83 MB through L2 for 6 MB of unique data: every 16x16 / 32x32 tile
// re-reads its operands) and 42 % queued on the MFMA pipe, the two phases in lockstep.  Here a
// workgroup owns a 64 (n) x 96 (j) tile, walks K in chunks of KC channels, and keeps two LDS
// buffers: while the four waves run the MFMAs of chunk c out of one buffer, the 16-byte global
// loads of chunk c+1 are in flight and land in the other.  Operand bytes through L2: 23 MB.
// LDS 2 x (64*KC + 96*(KC+4)) floats = 63 KB at KC = 48: two workgroups per CU, and the
// attention workgroups of the merged launch still fit next to them.
constexpr int kPipeJ = 96;

// NG = n-groups (16 columns each) per tile: 4 (64 x 96 tile) or 2 (32 x 96: twice the workgroups)
// NQ > 1 (launch of 256 NQ threads): the contraction is split over NQ groups of four waves ("quads") of ONE workgroup,
// each with its own pair of operand buffers walking its K / NQ; the partial tiles meet in LDS and quad 0 runs the
// epilogue — bias, store, BatchNorm statistics of the COMPLETE output, no atomics on it.  For long contractions on few
// tiles (the reshape layers of NTU / Ego at 64 / 48 samples: K = 2048 on 32 tiles per layer was a chain of 64 chunk
// steps per workgroup with most of the CU idle).
// JT = 16-column output tiles per wave along j: the workgroup tile is 16 NG (n) x 32 JT (j) — 96 wide by default; 64 / 32
// for the grouped reshape layers whose width is no multiple of 96, or whose contraction is long (more, smaller tiles).
template <int KC, int NG, int NQ = 1, int JT = 3>
__device__ __forceinline__ void conv_pipe_fwd_body(const ConvArgs& a, const int bx, const int by,
                                                   float* __restrict__ smem0) {
  constexpr int KP = KC + 4;
  constexpr int PJ = 32 * JT;                                   // tile width (kPipeJ = 96 for JT = 3)
  constexpr int TNC = 16 * NG, WN = NG / 2;                    // tile columns, n-groups per wave
  constexpr int A4 = TNC * KC / 4, B4 = PJ * KC / 4;       // float4 per chunk
  constexpr int NA = (A4 + 255) / 256, NB = (B4 + 255) / 256;
  // activation rows (one channel of one sample: L floats) are padded to L + 4 in LDS: the four
  // k-slots of a wave then read rows 4 apart on different banks (unpadded: 2-way conflict)
  const int ABUF = ((TNC * KC) >> a.Lb) * (a.L + 4);
  const int BUF = ABUF + PJ * KP;                   // floats per buffer
  const int quad = NQ > 1 ? (int)(threadIdx.x >> 8) : 0;
  float* __restrict__ smem = smem0 + quad * 2 * BUF;
  const int t = NQ > 1 ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
  const int wave = t >> 6, lane = t & 63, lo = lane & 15, h = lane >> 4;
  const int spt = TNC >> a.Lb;                                  // samples per tile
  const int s0 = bx * spt, j0 = by * PJ;
  const int K = a.I;
  const int cl4 = (KC << a.Lb) >> 2;                           // float4 per sample per chunk
  const float* __restrict__ act = a.act.p[0];
  const int nchunk = K / KC / NQ;                               // (the launcher picks NQ | K / KC)
  const int cbase = quad * nchunk;

  // global addresses of this thread's pieces of a chunk (chunk c adds c*KC channels)
  int64_t aoffg[NA];
  int asl[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = t + 256 * i;
    const int qq = q < A4 ? q : A4 - 1;
    const int sl = qq / cl4;                                   // sample inside the tile
    int s = s0 + sl;
    s = s < a.b ? s : a.b - 1;                                 // clamped: padded samples are never stored
    aoffg[i] = ((int64_t)s * K << a.Lb) + (int64_t)(qq - sl * cl4) * 4;
    const int row = (qq * 4) >> a.Lb, col = (qq * 4) & (a.L - 1);   // row = sample_local * KC + channel
    asl[i] = row * (a.L + 4) + col;
  }
  int64_t boffg[NB];
  int bsl[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q = t + 256 * i;
    const int qq = q < B4 ? q : B4 - 1;
    const int jr = qq / (KC / 4), c4 = qq - jr * (KC / 4);
    int j = j0 + jr;
    j = j < a.J ? j : a.J - 1;
    boffg[i] = (int64_t)j * a.ldw + 4 * c4;
    bsl[i] = ABUF + jr * KP + 4 * c4;
  }
  struct Regs {
    float4 ra[NA], rb[NB];
  };
  auto fetch = [&](Regs& R, int c) __attribute__((always_inline)) {
    c = cbase + (c < nchunk ? c : nchunk - 1);                 // past the end: a harmless repeat, no branch
#pragma unroll
    for (int i = 0; i < NA; ++i) R.ra[i] = ld4(act + aoffg[i] + ((int64_t)(c * KC) << a.Lb));
#pragma unroll
    for (int i = 0; i < NB; ++i) R.rb[i] = ld4(a.W + boffg[i] + c * KC);
  };
  // slice kb (of nkb) of a chunk's stash; nkb = 1: all of it
  auto stash = [&](float* buf, const Regs& R, int kb = 0, int nkb = 1) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if ((i * nkb) / (NA + NB) == kb && t + 256 * i < A4) st4(buf + asl[i], R.ra[i]);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (((NA + i) * nkb) / (NA + NB) == kb && t + 256 * i < B4) st4(buf + bsl[i], R.rb[i]);
  };

  const int gl0 = WN * (wave & 1), jl0 = JT * (wave >> 1);
  int aoff[WN], boff[JT];
#pragma unroll
  for (int tn = 0; tn < WN; ++tn)
    aoff[tn] = ((gl0 + tn) * a.spw + (lo >> a.Lb)) * KC * (a.L + 4) + (lo & (a.L - 1));
#pragma unroll
  for (int tj = 0; tj < JT; ++tj) boff[tj] = ABUF + ((jl0 + tj) * 16 + lo) * KP + 4 * h;
  f32x4 acc[WN][JT];
#pragma unroll
  for (int tn = 0; tn < WN; ++tn)
#pragma unroll
    for (int tj = 0; tj < JT; ++tj) acc[tn][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the epilogue's bias values go out with the first operand loads: fetched where they are used, each
  // output tile ended in load -> s_waitcnt vmcnt(0) -> store, three dependent round trips per wave
  float bjv[JT];
#pragma unroll
  for (int tj = 0; tj < JT; ++tj) {
    const int jt = j0 + 16 * (jl0 + tj);
    bjv[tj] = (a.bias != nullptr) ? a.bias[(jt < a.J ? jt : a.J - 16) + lo] : 0.f;
  }
  // A chunk step inside one wave (as in the data-gradient body): k-block kb + 1's LDS operand reads are issued in
  // front of k-block kb's MFMAs; with `nxt` the next chunk's stash into the other buffer is dealt over the k-blocks.
  auto compute = [&](const float* cur, float* nxt = nullptr, const Regs* Rn = nullptr) __attribute__((always_inline)) {
    constexpr int NKB = KC / 16;
    float av[NKB][WN][4];
    float4 bv[NKB][JT];
    auto rd = [&](int kb) __attribute__((always_inline)) {
      const int c0 = 16 * kb + 4 * h;
#pragma unroll
      for (int tn = 0; tn < WN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) av[kb][tn][r] = cur[aoff[tn] + (c0 + r) * (a.L + 4)];
#pragma unroll
      for (int tj = 0; tj < JT; ++tj) bv[kb][tj] = ld4(cur + boff[tj] + 16 * kb);
    };
    rd(0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) rd(kb + 1);
      if (nxt != nullptr) stash(nxt, *Rn, kb, NKB);
      else __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tn = 0; tn < WN; ++tn)
#pragma unroll
        for (int tj = 0; tj < JT; ++tj) {
          acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][tn][0], bv[kb][tj].x, acc[tn][tj], 0, 0, 0);
          acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][tn][1], bv[kb][tj].y, acc[tn][tj], 0, 0, 0);
          acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][tn][2], bv[kb][tj].z, acc[tn][tj], 0, 0, 0);
          acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][tn][3], bv[kb][tj].w, acc[tn][tj], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  Regs R;
  fetch(R, 0);
  stash(smem, R);
  __syncthreads();
  if (kFwdLa2 && (nchunk & 1) == 0 && nchunk >= 4) {
    // two chunks of global loads in flight (register sets by chunk parity), no conditional fetch / stash in the loop
    Regs Q;
    fetch(Q, 1);
    for (int c = 0; c + 2 < nchunk; c += 2) {
      fetch(R, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(smem, smem + BUF, &Q);
      __syncthreads();
      fetch(Q, c + 3);
      __builtin_amdgcn_sched_barrier(0);
      compute(smem + BUF, smem, &R);
      __syncthreads();
    }
    compute(smem, smem + BUF, &Q);
    __syncthreads();
    compute(smem + BUF);
  } else {
    for (int c = 0; c < nchunk; ++c) {
      if (c + 1 < nchunk) fetch(R, c + 1);                     // in flight during this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute(smem + (c & 1) * BUF);
      if (c + 1 < nchunk) {
        stash(smem + ((c + 1) & 1) * BUF, R);                  // the other buffer: nobody reads it now
        __syncthreads();
      }
    }
  }

  if (NQ > 1) {
    // the quads' partial tiles: quads 1 .. NQ - 1 park theirs in quad 0's (now idle) operand buffers — the quads run in
    // lockstep (same chunk count, same barriers), so behind this barrier nobody reads operands any more
    __syncthreads();
    float* red = smem0;                                         // [NQ - 1][WN * JT * 4][256]
    if (quad > 0) {
#pragma unroll
      for (int tn = 0; tn < WN; ++tn)
#pragma unroll
        for (int tj = 0; tj < JT; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[(((quad - 1) * WN * JT + tn * JT + tj) * 4 + r) * 256 + t] = acc[tn][tj][r];
    }
    __syncthreads();
    if (quad == 0) {
#pragma unroll
      for (int q = 1; q < NQ; ++q)
#pragma unroll
        for (int tn = 0; tn < WN; ++tn)
#pragma unroll
          for (int tj = 0; tj < JT; ++tj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tn][tj][r] += red[(((q - 1) * WN * JT + tn * JT + tj) * 4 + r) * 256 + t];
    }
    __syncthreads();                                            // (the statistics exchange below reuses smem0)
  }
  const bool lead = NQ == 1 || quad == 0;                       // the quad that owns the epilogue
  // epilogue: acc[tn][tj][r] = OUT[n = 16*g + 4h + r][j = jt + lo]
  const int l0 = (4 * h) & (a.L - 1);
  float ssum[JT], ssq[JT];                                      // stat mode: this wave's n-groups together
#pragma unroll
  for (int tj = 0; tj < JT; ++tj) ssum[tj] = ssq[tj] = 0.f;
#pragma unroll
  for (int tn = 0; tn < WN; ++tn)
#pragma unroll
    for (int tj = 0; tj < JT; ++tj) {
      const int g = bx * NG + gl0 + tn, jt = j0 + 16 * (jl0 + tj);
      if (g >= a.n_groups || jt >= a.J || !lead) continue;     // wave-uniform
      const int jj = jt + lo;
      const float bj = bjv[tj];
      const float4 o = make_float4(acc[tn][tj][0] + bj, acc[tn][tj][1] + bj, acc[tn][tj][2] + bj,
                                   acc[tn][tj][3] + bj);
      const int so = g * a.spw + ((4 * h) >> a.Lb);
      const bool vo = so < a.b;
      if (vo) st4_w0<1>(a.dst.p[0] + ((int64_t)so * a.Cj + jj) * a.L + l0, o);
      if (a.stat == nullptr) {
        bn_tile_stats(a, o, bj, vo, g, jj, h);
      } else if (vo) {                                         // d = u - bias = the bare accumulator
        ssum[tj] += (acc[tn][tj][0] + acc[tn][tj][1]) + (acc[tn][tj][2] + acc[tn][tj][3]);
        ssq[tj] += acc[tn][tj][0] * acc[tn][tj][0] + acc[tn][tj][1] * acc[tn][tj][1] +
                   acc[tn][tj][2] * acc[tn][tj][2] + acc[tn][tj][3] * acc[tn][tj][3];
      }
    }
  if (a.stat != nullptr) {
    // ONE atomic pair per channel per WORKGROUP (the tile's NG n-groups together): waves 1 and 3 hand
    // their sums to waves 0 and 2, which own the same channels, through the idle operand buffer.
    // Same-address atomics serialise at the memory side: halving them again lets half as many shard
    // copies do (the kernel that finalises the statistics reads every shard of every channel).
    float2* xch = reinterpret_cast<float2*>(smem0);              // [2][JT][16]
#pragma unroll
    for (int tj = 0; tj < JT; ++tj) {
      ssum[tj] = xor16_sum(ssum[tj]);
      ssum[tj] = xor32_sum(ssum[tj]);
      ssq[tj] = xor16_sum(ssq[tj]);
      ssq[tj] = xor32_sum(ssq[tj]);
    }
    __syncthreads();                                           // every wave is done reading operands
    if (lead && (wave & 1) && h == 0) {
#pragma unroll
      for (int tj = 0; tj < JT; ++tj) xch[((wave >> 1) * JT + tj) * 16 + lo] = make_float2(ssum[tj], ssq[tj]);
    }
    __syncthreads();
    if (lead && !(wave & 1) && h == 0) {
#pragma unroll
      for (int tj = 0; tj < JT; ++tj) {
        const int jt = j0 + 16 * (jl0 + tj);
        if (jt >= a.J) continue;
        const float2 p = xch[((wave >> 1) * JT + tj) * 16 + lo];
        float* pp = a.stat + ((int64_t)(bx % a.stat_shards) * a.J + jt + lo) * 2;
        atomicAdd(pp, ssum[tj] + p.x);
        atomicAdd(pp + 1, ssq[tj] + p.y);
      }
    }
  }
}

template <int KC, int NG>
__global__ __launch_bounds__(256) void conv_pipe_fwd_k(ConvArgs a, int gx) {
  extern __shared__ __attribute__((aligned(16))) float pipe_smem[];
  conv_pipe_fwd_body<KC, NG>(a, blockIdx.x % gx, blockIdx.x / gx, pipe_smem);
}

// Data-gradient twin: OUT[n][j] = sum_i ACT[i][n] W[i][j], long contraction (I = 3C), narrow output.
// Tile 32 (n) x 64 (j): 192 workgroups at batch 128, each wave owns 1 n-group x 2 j-tiles... (wave w:
// n-group w & 1, j-tiles 2*(w >> 1), +1).  W chunk is k-major in LDS, rows of 64 + 4 floats.
constexpr int kPipeBN = 32, kPipeBJ = 64;

// NG = n-groups per tile: 2 (32 x 64 tile, two j-tiles per wave) or 1 (16 x 64, one per wave)
template <int KC, int NG, bool FOLD, bool LA2 = false>
__device__ __forceinline__ void conv_pipe_bwd_body(const ConvArgs& a, const int bx, const int by,
                                                   float* __restrict__ smem) {
  constexpr int JP = kPipeBJ + 4;
  constexpr int TNC = 16 * NG, WJ = NG;                       // tile columns, j-tiles per wave
  constexpr int A4 = TNC * KC / 4, B4 = KC * kPipeBJ / 4;
  constexpr int NA = (A4 + 255) / 256, NB = (B4 + 255) / 256;
  const int ABUF = ((TNC * KC) >> a.Lb) * (a.L + 4);           // rows padded to L + 4 (see forward body)
  const int BUF = ABUF + KC * JP;
  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63, lo = lane & 15, h = lane >> 4;
  const int spt = TNC >> a.Lb;
  const int s0 = bx * spt, j0 = by * kPipeBJ;
  const int K = a.I;
  const int cl4 = (KC << a.Lb) >> 2;
  const float* __restrict__ act = a.act.p[0];
  const int nchunk = K / KC;
  int64_t aoffg[NA];
  int asl[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = t + 256 * i;
    const int qq = q < A4 ? q : A4 - 1;
    const int sl = qq / cl4;
    int s = s0 + sl;
    s = s < a.b ? s : a.b - 1;
    aoffg[i] = ((int64_t)s * K << a.Lb) + (int64_t)(qq - sl * cl4) * 4;
    const int row = (qq * 4) >> a.Lb, col = (qq * 4) & (a.L - 1);
    asl[i] = row * (a.L + 4) + col;
  }
  int64_t boffg[NB];
  int bsl[NB];
  const int jmax4 = a.J / 4 - 1;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int q = t + 256 * i;
    const int qq = q < B4 ? q : B4 - 1;
    const int kr = qq / (kPipeBJ / 4), c4 = qq - kr * (kPipeBJ / 4);
    int j4 = j0 / 4 + c4;
    j4 = j4 < jmax4 ? j4 : jmax4;
    boffg[i] = (int64_t)kr * a.ldw + 4 * j4;
    bsl[i] = ABUF + kr * JP + 4 * c4;
  }
  // BatchNorm-backward fold: per-channel (alpha, beta, gamma) of all K channels in LDS behind the two
  // operand buffers — dU = alpha * dV + beta * U + gamma (bn_fold_coef): two FMAs per element.  FOLD is a
  // template parameter (eval mode: alpha = scale, beta = gamma = 0, U still fetched): no run-time branches
  // inside the chunk loop, where hipcc joins them with s_waitcnt vmcnt(0).
  float4* coef = reinterpret_cast<float4*>(smem + 2 * BUF);
  int chl[NA];                                                  // channel of ra[i] inside a chunk
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = t + 256 * i;
    const int qq = q < A4 ? q : A4 - 1;
    chl[i] = ((qq - (qq / cl4) * cl4) * 4) >> a.Lb;
  }
  struct Regs {
    float4 ra[NA], ru[NA], rb[NB];
  };
  auto fetch = [&](Regs& R, int c) __attribute__((always_inline)) {
    c = c < nchunk ? c : nchunk - 1;                           // past the end: a harmless repeat, no branch
#ifdef BMNAS_PROBE_DGRAD_NOLAT
    c = 0;      // timing experiment (wrong results): every chunk re-reads chunk 0 — the loads hit the CU's L1
#endif
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      R.ra[i] = ld4(act + aoffg[i] + ((int64_t)(c * KC) << a.Lb));
      if (FOLD) R.ru[i] = ld4(a.bn_U + aoffg[i] + ((int64_t)(c * KC) << a.Lb));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) R.rb[i] = ld4(a.W + boffg[i] + (int64_t)(c * KC) * a.ldw);
  };
  auto stash = [&](float* buf, const Regs& R, int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (256 * (i + 1) <= A4 || t + 256 * i < A4) {          // whole rounds: no predicate
        float4 v = R.ra[i];
        if (FOLD) {
          const float4 cf = coef[c * KC + chl[i]];
          v.x = fmaf(cf.x, v.x, fmaf(cf.y, R.ru[i].x, cf.z));
          v.y = fmaf(cf.x, v.y, fmaf(cf.y, R.ru[i].y, cf.z));
          v.z = fmaf(cf.x, v.z, fmaf(cf.y, R.ru[i].z, cf.z));
          v.w = fmaf(cf.x, v.w, fmaf(cf.y, R.ru[i].w, cf.z));
        }
        st4(buf + asl[i], v);
      }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (256 * (i + 1) <= B4 || t + 256 * i < B4) st4(buf + bsl[i], R.rb[i]);
  };
  const int gl = (NG == 2) ? (wave & 1) : 0, jl0 = (NG == 2) ? 2 * (wave >> 1) : wave;
  const int aoff = (gl * a.spw + (lo >> a.Lb)) * KC * (a.L + 4) + (lo & (a.L - 1));
  int boff[WJ];
#pragma unroll
  for (int tj = 0; tj < WJ; ++tj) boff[tj] = ABUF + 4 * h * JP + (jl0 + tj) * 16 + lo;
  f32x4 acc[WJ];
#pragma unroll
  for (int tj = 0; tj < WJ; ++tj) acc[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // A chunk step inside one wave: k-block kb + 1's LDS operand reads are issued in front of k-block kb's MFMAs, and
  // (two-chunks-in-flight loop) the NEXT chunk's fold + stash into the other buffer is dealt over the k-blocks, so its
  // VALU / LDS-write instructions issue in the shadow of the matrix pipe.  History: rounds 2-3 read the whole chunk's
  // operands first and ran its 24 MFMAs back to back behind a fence, then stashed (19.05 us for the merged launch);
  // reads pipelined per k-block 18.5 us; with the stash slices 18.3 us; no fences at all 19.3 us.
#if BMNAS_BODY_PROBES
  unsigned long long pr_reads = 0, pr_mfma = 0, pr_rest = 0, pr_last = stamp_clock(), pr_t0 = pr_last;
#endif
  // one slice (of NKB) of a chunk's stash: the NA + NB register rounds dealt over the k-blocks of the step that runs
  // while they are written
  auto stash_slice = [&](float* buf, const Regs& R, int c, int kb, int nkb) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if ((i * nkb) / (NA + NB) == kb && (256 * (i + 1) <= A4 || t + 256 * i < A4)) {
        float4 v = R.ra[i];
        if (FOLD) {
          const float4 cf = coef[c * KC + chl[i]];
          v.x = fmaf(cf.x, v.x, fmaf(cf.y, R.ru[i].x, cf.z));
          v.y = fmaf(cf.x, v.y, fmaf(cf.y, R.ru[i].y, cf.z));
          v.z = fmaf(cf.x, v.z, fmaf(cf.y, R.ru[i].z, cf.z));
          v.w = fmaf(cf.x, v.w, fmaf(cf.y, R.ru[i].w, cf.z));
        }
        st4(buf + asl[i], v);
      }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (((NA + i) * nkb) / (NA + NB) == kb && (256 * (i + 1) <= B4 || t + 256 * i < B4)) st4(buf + bsl[i], R.rb[i]);
  };
  auto compute = [&](const float* cur, float* nxt = nullptr, const Regs* Rn = nullptr, int cn = 0)
      __attribute__((always_inline)) {
    constexpr int NKB = KC / 16;
    float av[NKB][4], bv[NKB][WJ][4];
#if BMNAS_BODY_PROBES
    { const unsigned long long t = stamp_clock(); pr_rest += t - pr_last; pr_last = t; }
#endif
    auto rd = [&](int kb) __attribute__((always_inline)) {
      const int c0 = 16 * kb + 4 * h;
#pragma unroll
      for (int r = 0; r < 4; ++r) av[kb][r] = cur[aoff + (c0 + r) * (a.L + 4)];
#pragma unroll
      for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[kb][tj][r] = cur[boff[tj] + (16 * kb + r) * JP];
    };
    rd(0);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) rd(kb + 1);
      if (nxt != nullptr) stash_slice(nxt, *Rn, cn, kb, NKB);
      else __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tj = 0; tj < WJ; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][r], bv[kb][tj][r], acc[tj], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#if BMNAS_BODY_PROBES
    { const unsigned long long t = stamp_clock(); pr_mfma += t - pr_last; pr_last = t; }
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
  Regs R;
  fetch(R, 0);
  if (FOLD) {
    // Every load of the coefficients issued at once — three rounds of 256 channels unrolled, clamped channel index,
    // nothing under `if (bn_train)` — right behind chunk 0's operand loads: written as a run-time loop with the
    // training-mode loads under a branch, each round waited for its five loads on its own (two to three dependent
    // round trips in front of every data-gradient tile's first stash).
    const float invN = 1.f / (float)(a.b * a.L);
    const float* const bg = a.bn_train ? a.bn_grad : a.bn_chan;   // (eval mode: bn_grad is null; the values are unused)
    constexpr int kRounds = 3;                                  // K <= 768 in one pass; longer contractions loop
    for (int m0 = 0; m0 < K; m0 += 256 * kRounds) {
      float scv[kRounds], g1[kRounds], mu[kRounds], rs[kRounds], g0[kRounds];
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) {
        const int m = m0 + rd * 256 + t;
        const int mc = m < K ? m : K - 1;
        scv[rd] = a.bn_chan[2 * K + mc];
        g1[rd] = bg[K + mc];
        mu[rd] = a.bn_chan[mc];
        rs[rd] = a.bn_chan[K + mc];
        g0[rd] = bg[mc];
      }
#pragma unroll
      for (int rd = 0; rd < kRounds; ++rd) {
        const int m = m0 + rd * 256 + t;
        float4 cf = make_float4(scv[rd], 0.f, 0.f, 0.f);
        if (a.bn_train) cf = bn_fold_coef(scv[rd], g1[rd] * invN, mu[rd], rs[rd] * (g0[rd] * invN));
        if (m < K) coef[m] = cf;
      }
    }
    __syncthreads();
  }
  stash(smem, R, 0);
  __syncthreads();
  if (LA2 && (nchunk & 1) == 0 && nchunk >= 4) {
    // TWO chunks of global loads in flight (register sets by chunk parity).  The steady-state loop has no
    // conditional fetch or stash: a stash that may be skipped leaves its set's loads pending on one path, and
    // hipcc then waits for them at the loop head on EVERY path (that, not the idea, is what made the first
    // two-chunk experiments slower); the last two chunks run after the loop.
    Regs Q;
    fetch(Q, 1);
    for (int c = 0; c + 2 < nchunk; c += 2) {
      fetch(R, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      compute(smem, smem + BUF, &Q, c + 1);                    // (stash into the other buffer: nobody reads it now)
      __syncthreads();
      fetch(Q, c + 3);
      __builtin_amdgcn_sched_barrier(0);
      compute(smem + BUF, smem, &R, c + 2);
      __syncthreads();
    }
    compute(smem, smem + BUF, &Q, nchunk - 1);
    __syncthreads();
    compute(smem + BUF);
  } else {
    for (int c = 0; c < nchunk; ++c) {
      fetch(R, c + 1);                                         // in flight during this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      compute(smem + (c & 1) * BUF);
      if (c + 1 < nchunk) {
        stash(smem + ((c + 1) & 1) * BUF, R, c + 1);          // the other buffer: nobody reads it now
        __syncthreads();
      }
    }
  }
#if BMNAS_BODY_PROBES
  if (threadIdx.x == 0 && g_bmnas_stamps != nullptr) {         // slot 6: [wg][reads, mfma, rest, total] summed over chunks
    const int wg = by * 4096 / 64 + bx;                        // (by < 64 tiles of 64 j, bx < 64 n-tiles at b <= 128)
    if (wg < g_bmnas_stamp_slots) {
      unsigned long long* p = g_bmnas_stamps + ((size_t)6 * g_bmnas_stamp_slots + wg) * 8;
      p[0] = pr_reads; p[1] = pr_mfma; p[2] = pr_rest; p[3] = stamp_clock() - pr_t0; p[6] = 1; p[7] = (unsigned long long)nchunk;
    }
  }
#endif
  const int l0 = (4 * h) & (a.L - 1);
  const int g = bx * NG + gl;
#pragma unroll
  for (int tj = 0; tj < WJ; ++tj) {
    const int jt = j0 + 16 * (jl0 + tj);
    if (g >= a.n_groups || jt >= a.J) continue;                // wave-uniform
    const int jj = jt + lo;
    const int q = jj / a.Cj;
    const int cj = jj - q * a.Cj;
    // (pick_ptr, not a hand-written select chain: LLVM folds select(load, load) of kernel-argument pointers back into ONE
    // indexed load — a global_load_dwordx2 from the kernarg segment + s_waitcnt vmcnt(0) in front of the tile's store, and
    // in the grouped kernels, whose descriptor is a local copy, 200 B of scratch per lane; round 5, by the ISA)
    float* d = pick_ptr(a.dst.p, q);
    const int so = g * a.spw + ((4 * h) >> a.Lb);
    if (so >= a.b || d == nullptr) continue;
    float* pp = d + ((int64_t)so * a.Cj + cj) * a.L + l0;
    const float4 o = make_float4(acc[tj][0], acc[tj][1], acc[tj][2], acc[tj][3]);
    st4_w0<7>(pp, (a.acc_mask & (1u << q)) ? f4_add(o, ld4(pp)) : o);
  }
}

template <int KC, int NG>
__global__ __launch_bounds__(256) void conv_pipe_bwd_k(ConvArgs a, int gx) {
  extern __shared__ __attribute__((aligned(16))) float pipe_smem[];
  if (a.bn_U != nullptr) conv_pipe_bwd_body<KC, NG, true>(a, blockIdx.x % gx, blockIdx.x / gx, pipe_smem);
  else conv_pipe_bwd_body<KC, NG, false>(a, blockIdx.x % gx, blockIdx.x / gx, pipe_smem);
}

template <int KC, int NG>
inline size_t conv_pipe_bwd_lds(int L) {
  return (size_t)2 * (16 * NG * KC / L * (L + 4) + KC * (kPipeBJ + 4)) * sizeof(float);
}

template <int KC, int NG, int JT = 3>
inline size_t conv_pipe_lds(int L) {
  return (size_t)2 * (16 * NG * KC / L * (L + 4) + 32 * JT * (KC + 4)) * sizeof(float);
}

// ---- GEMM + attention in one launch ---------------------------------------------------------
// In a NodeMixedOp the attention branch and the stacked LinearGLU/ConcatFC conv read the same
// input and do not depend on each other.  K3 is 16/L samples per workgroup = 128 workgroups at
// batch 128: half of the 256 CUs idle for its 7-11 us.  Here one grid carries both: the first
// `groups` workgroups run the attention body (they are the long dependent chains, so they start
// first), the rest are the GEMM's tiles.  Forward: conv + sdpa_ln_fwd.  Backward: data-gradient
// GEMM + sdpa_ln_bwd (the attention gradient goes to its own buffer; the consumer adds the two).
struct SdpaFwdArgs {
  const float *x, *y, *ln_w, *ln_b;
  float *out, *xhat, *stats;
  SdpaGeom G;
  DropCfg drop;
  int groups;
};
struct SdpaBwdArgs {
  const float *g, *gscale, *x, *y, *ln_w, *xhat, *stats;
  float *dx, *dy;
  uint32_t acc_mask;
  SdpaGeom G;
  DropCfg drop;
  int groups;
};

// KCH = 16-channel chunks per wave of the attention body = ceil(C / 64); the merged launches exist
// for the NodeMixedOp shapes only (conv input = the attention input, K = C, M = 3C), where the
// GEMM's blocks-per-wave are KCH (forward) and 3 KCH (data gradient).
template <int TN, int TJ, int KCH>
__global__ __launch_bounds__(256) void conv_fwd_sdpa_k(ConvArgs a, SdpaFwdArgs s, int gx) {
  constexpr int KPW = KCH;
  extern __shared__ __attribute__((aligned(16))) char merged_smem[];
  if ((int)blockIdx.x < s.groups) {
    sdpa_fwd_body<KCH>(blockIdx.x, s.x, s.y, s.ln_w, s.ln_b, s.out, s.xhat, s.stats, s.G, s.drop, merged_smem);
  } else {
    const int t = blockIdx.x - s.groups;
    conv_ksplit_body<true, TN, TJ, KPW>(a, t % gx, t / gx, merged_smem);
  }
}

// the same merged forward launch with the pipelined tile kernel as the GEMM half
template <int KC, int KCH, int NG>
__global__ __launch_bounds__(256) void conv_pipe_fwd_sdpa_k(ConvArgs a, SdpaFwdArgs s, int gx) {
  extern __shared__ __attribute__((aligned(16))) char merged_smem[];
  // (attention groups first: the other order measured 11.8 -> 12.1 us at MM-IMDB b128)
  // BMNAS_CONV_PROBE bits 16 / 64 (timing builds only) drop the attention groups / the GEMM tiles
  if ((a.probe & 16) && (int)blockIdx.x < s.groups) return;
  if ((a.probe & 64) && (int)blockIdx.x >= s.groups) return;
  if ((int)blockIdx.x < s.groups) {
    BMNAS_SETPRIO(BMNAS_PRIO_FWD_A);
    sdpa_fwd_body<KCH>(blockIdx.x, s.x, s.y, s.ln_w, s.ln_b, s.out, s.xhat, s.stats, s.G, s.drop, merged_smem);
  } else {
    BMNAS_SETPRIO(BMNAS_PRIO_FWD_G);
    const int t = blockIdx.x - s.groups;
    conv_pipe_fwd_body<KC, NG>(a, t % gx, t / gx, reinterpret_cast<float*>(merged_smem));
  }
}

// ---- LDS-staged variant (the one launched at production sizes) ------------------------------
// Operand fetch straight from global costs one VMEM instruction per 256 B (the activation
// rows are only L*4 = 32-64 B long), and at ~300 loads per wave the kernel is bound by VMEM
// issue, not by the matrix cores.  Here a workgroup stages a 32-deep slice of both operands
// with 1-KiB float4 loads, shares it between its four waves through LDS, and overlaps the
// next slice's global loads (held in registers) with the current slice's MFMAs:
//   sA[k][n]  activation slice, k-major, rows padded to BN+4 floats (conflict-free b32 reads:
//             the four k-slots of a wave read rows 4 apart -> banks 16 apart);
//   sB        TRANS: W[j][i] slice stored [j][k] (k contiguous, row BK+4) and read as ONE b128
//             per tile per 16 k (k-permuted: step r <-> k = kb + 4*slot + r, same permutation
//             on the A side);  !TRANS: W[i][j] slice stored [k][j] like sA.
template <bool TRANS, int BN, int BJ>
__global__ __launch_bounds__(256) void conv_lds_k(ConvArgs a) {
  constexpr int BK = 32;
  constexpr int TN = BN / 32, TJ = BJ / 32;              // 16x16 tiles per wave (waves are 2 x 2)
  constexpr int LDA = BN + 4;
  constexpr int LDB = TRANS ? (BK + 4) : (BJ + 4);
  constexpr int SB_ROWS = TRANS ? BJ : BK;
  constexpr int NA = BK * BN / 4 / 256;                  // float4 per thread per stage (1 or 2)
  constexpr int NB = BK * BJ / 4 / 256;
  __shared__ __attribute__((aligned(16))) float sA[2][BK * LDA];
  __shared__ __attribute__((aligned(16))) float sB[2][SB_ROWS * LDB];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int wn = wave & 1, wj = wave >> 1;
  const int gbase = blockIdx.x * (BN / 16);              // first n-group of the workgroup
  const int jbase = blockIdx.y * BJ;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- staging: slot v of a thread is float4 number threadIdx.x + 256*v of the slice; the
  // registers are individually named (arrays indexed in unrolled lambdas ended up in scratch)
  auto a_coord = [&](int v, int& k, int& n, int64_t& off) __attribute__((always_inline)) {
    const int idx = threadIdx.x + 256 * v;
    k = idx / (BN / 4);
    n = (idx % (BN / 4)) * 4;
    int g = gbase + n / 16;
    g = g < a.n_groups ? g : a.n_groups - 1;
    const int u = n & 15;
    int smp = g * a.spw + (u >> a.Lb);
    smp = smp < a.b ? smp : a.b - 1;                     // clamped: padded columns are never stored
    off = ((int64_t)smp * a.Ci) * a.L + (u & (a.L - 1));
  };
  auto b_coord = [&](int v, int& r, int& c) __attribute__((always_inline)) {
    const int idx = threadIdx.x + 256 * v;
    if (TRANS) {                                         // (j, 4*i4)
      r = idx / (BK / 4);
      c = (idx % (BK / 4)) * 4;
    } else {                                             // (k, 4*j4)
      r = idx / (BJ / 4);
      c = (idx % (BJ / 4)) * 4;
    }
  };
  auto gload_a = [&](int v, int i0) __attribute__((always_inline)) -> float4 {
    int k, n;
    int64_t off;
    a_coord(v, k, n, off);
    const int i = i0 + k;
    const int ic = i < a.I ? i : a.I - 1;
    const int q = ic / a.Ci;
    const float* sp = pick_ptr(a.act.p, q);
    const float4 t = ld4(sp + off + (int64_t)(ic - q * a.Ci) * a.L);
    return i < a.I ? t : z4;
  };
  auto gload_b = [&](int v, int i0) __attribute__((always_inline)) -> float4 {
    int r, c;
    b_coord(v, r, c);
    if (TRANS) {
      int j = jbase + r;
      j = j < a.J ? j : a.J - 1;
      const int i = i0 + c;                              // I % 16 == 0: a float4 is all in or all out
      const int ic = i < a.I ? i : a.I - 4;
      float4 t = ld4(a.W + (int64_t)j * a.ldw + ic);
      if (a.fold > 0) t = f4_add(t, ld4(a.W + (int64_t)j * a.ldw + ic + a.fold));
      return i < a.I ? t : z4;
    } else {
      const int i = i0 + r;
      const int ic = i < a.I ? i : a.I - 1;
      int j = jbase + c;                                 // J % 16 == 0
      j = j < a.J ? j : a.J - 4;
      float4 t = ld4(a.W + (int64_t)ic * a.ldw + j);
      if (a.fold > 0) t = f4_add(t, ld4(a.W + (int64_t)ic * a.ldw + j + a.fold));
      return i < a.I ? t : z4;
    }
  };
  auto lstore_a = [&](int v, int buf, float4 val) __attribute__((always_inline)) {
    int k, n;
    int64_t off;
    a_coord(v, k, n, off);
    *reinterpret_cast<float4*>(&sA[buf][k * LDA + n]) = val;
  };
  auto lstore_b = [&](int v, int buf, float4 val) __attribute__((always_inline)) {
    int r, c;
    b_coord(v, r, c);
    *reinterpret_cast<float4*>(&sB[buf][r * LDB + c]) = val;
  };
  float4 ra0 = z4, ra1 = z4, rb0 = z4, rb1 = z4;
  auto gload = [&](int i0) __attribute__((always_inline)) {
    ra0 = gload_a(0, i0);
    if (NA > 1) ra1 = gload_a(1, i0);
    rb0 = gload_b(0, i0);
    if (NB > 1) rb1 = gload_b(1, i0);
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    lstore_a(0, buf, ra0);
    if (NA > 1) lstore_a(1, buf, ra1);
    lstore_b(0, buf, rb0);
    if (NB > 1) lstore_b(1, buf, rb1);
  };

  f32x4 acc[TN][TJ];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) acc[tn][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const float* A = sA[buf];
    const float* B = sB[buf];
#pragma unroll
    for (int kb = 0; kb < BK; kb += 16) {
      float av[TN][4], bv[TJ][4];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) av[tn][r] = A[(kb + 4 * h + r) * LDA + (wn * TN + tn) * 16 + lo];
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) {
        const int jl = (wj * TJ + tj) * 16 + lo;
        if (TRANS) {
          const float4 t = *reinterpret_cast<const float4*>(&B[jl * LDB + kb + 4 * h]);
          bv[tj][0] = t.x; bv[tj][1] = t.y; bv[tj][2] = t.z; bv[tj][3] = t.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[tj][r] = B[(kb + 4 * h + r) * LDB + jl];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int tj = 0; tj < TJ; ++tj)
            acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tn][r], bv[tj][r], acc[tn][tj], 0, 0, 0);
    }
  };

  const int nstage = (a.I + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    if (st + 1 < nstage) gload((st + 1) * BK);           // in flight during the MFMAs below
    compute(buf);
    if (st + 1 < nstage) lstore(buf ^ 1);                // other buffer: last read one barrier ago
    __syncthreads();
  }

  // ---- epilogue: acc[tn][tj][r] = OUT[n = 16*g + 4h + r][j] ----
  const int l0 = (4 * h) & (a.L - 1);
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int jt = jbase + (wj * TJ + tj) * 16;
    if (jt >= a.J) continue;                             // wave-uniform
    const int jj = jt + lo;
    const float bj = (a.bias != nullptr) ? a.bias[jj] : 0.f;
    const int q = jj / a.Cj;
    const int cj = jj - q * a.Cj;
    // (pick_ptr, not a hand-written select chain: LLVM folds select(load, load) of kernel-argument pointers back into ONE
    // indexed load — a global_load_dwordx2 from the kernarg segment + s_waitcnt vmcnt(0) in front of the tile's store, and
    // in the grouped kernels, whose descriptor is a local copy, 200 B of scratch per lane; round 5, by the ISA)
    float* d = pick_ptr(a.dst.p, q);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int g = gbase + wn * TN + tn;
      if (g >= a.n_groups) continue;                     // wave-uniform
      const int so = g * a.spw + ((4 * h) >> a.Lb);
      const bool vo = so < a.b;
      const float4 o = make_float4(acc[tn][tj][0] + bj, acc[tn][tj][1] + bj, acc[tn][tj][2] + bj,
                                   acc[tn][tj][3] + bj);
      if (vo && d != nullptr) {
        float* pp = d + ((int64_t)so * a.Cj + cj) * a.L + l0;
        st4_wtg<4>(pp, (a.acc_mask & (1u << q)) ? f4_add(o, ld4(pp)) : o);
      }
      bn_tile_stats(a, o, bj, vo, g, jj, h);
    }
  }
}

struct ConvWArgs {
  const float* bn_U;     // as ConvArgs: dU then holds dV and the operand loads apply the BatchNorm backward
  const float* bn_chan;
  const float* bn_grad;
  int bn_train;
  const float* dU;       // (b, M, L)
  ConvIn src;            // n_src sources (b, C_src, L); K = n_src * C_src
  float* dW;
  float* dbias;          // nullable
  int ldw, C_src, M, K, dup_cols;
  int b, L, Lb, spw, n_groups, groups_per_split;
  int use_atomic;
};

// dW[m][k] += sum_n dU[m][n] * X[k][n].  Workgroup = 8 waves on ONE 32(m) x 32(k) output
// tile; wave w takes the n-groups gbeg + w, + 8, ... of the workgroup's split.  Both
// operands are float4 along l (next group's loads issued before this group's 16 MFMAs), the
// eight partial tiles meet in LDS and leave with coalesced stores (single split) or
// well-shaped fp32 atomics (128-B runs along k).
// FOLD: the BatchNorm input gradient is applied to the dU operand (ConvWArgs::bn_U != nullptr).  A template
// parameter, not a run-time test: with `if (a.bn_U)` / `if (a.bn_train)` inside the loop hipcc joined the
// branches with s_waitcnt vmcnt(0) — and with the fold written where the operands are LOADED it had to wait
// for the loads it had just issued before the previous group's MFMAs: every group paid a full memory round
// trip (the two-set pipeline existed in the source only).
template <int NW, bool FOLD>
__device__ __forceinline__ void conv_w_body(const ConvWArgs& a, const int bx, const int by, const int bz,
                                            char* lds) {
  float (*tile)[32 * 33] = reinterpret_cast<float (*)[32 * 33]>(lds);                          // [NW][1056]
  float (*brow)[32] = reinterpret_cast<float (*)[32]>(lds + (size_t)NW * 32 * 33 * sizeof(float));   // [NW][32]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int m0 = bx * 32, k0 = by * 32;
  const int gbeg = bz * a.groups_per_split;
  int gend = gbeg + a.groups_per_split;
  if (gend > a.n_groups) gend = a.n_groups;
  const bool want_bias = (a.dbias != nullptr) && (by == 0);

  // tiles past the edge are clamped (their results are never stored)
  auto a_row = [&](int t) __attribute__((always_inline)) -> int64_t {
    const int mt = m0 + 16 * t;
    return (int64_t)((mt < a.M ? mt : a.M - 16) + lo) * a.L;
  };
  auto b_row = [&](int t, const float*& sp) __attribute__((always_inline)) -> int64_t {
    const int kt = k0 + 16 * t;
    const int k = (kt < a.K ? kt : a.K - 16) + lo;
    const int q = k / a.C_src;
    sp = pick_ptr(a.src.p, q);                           // q differs per lane: selects, not an indexed load
    return (int64_t)(k - q * a.C_src) * a.L;
  };
  const float *bsrc0, *bsrc1;
  const int64_t aoff0 = a_row(0), aoff1 = a_row(1);
  const int64_t boff0 = b_row(0, bsrc0), boff1 = b_row(1, bsrc1);
  const int l0 = (4 * h) & (a.L - 1);
  const int sh = (4 * h) >> a.Lb;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x4 acc[2][2];
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) acc[tm][tk] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};

  // BatchNorm-backward fold (see ConvArgs): this lane's two dU rows are two fixed channels, their
  // (alpha, beta, gamma) of dU = alpha * dV + beta * U + gamma live in registers for the whole walk over
  // the batch (eval mode: alpha = scale, beta = gamma = 0)
  float4 cf[2] = {z4, z4};
  if (FOLD) {
    const float invN = 1.f / (float)(a.b * a.L);
    // (all ten loads unconditional and first — eval mode, where bn_grad is null, reads bn_chan in its place and
    // discards it: loads under `if (bn_train)` were waited for at the join, one channel tile after the other)
    const float* const bg = a.bn_train ? a.bn_grad : a.bn_chan;
    float scv[2], g1[2], mu[2], rs[2], g0[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      const int mt = m0 + 16 * tm;
      const int m = (mt < a.M ? mt : a.M - 16) + lo;
      scv[tm] = a.bn_chan[2 * a.M + m];
      g1[tm] = bg[a.M + m];
      mu[tm] = a.bn_chan[m];
      rs[tm] = a.bn_chan[a.M + m];
      g0[tm] = bg[m];
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      cf[tm] = make_float4(scv[tm], 0.f, 0.f, 0.f);
      if (a.bn_train) cf[tm] = bn_fold_coef(scv[tm], g1[tm] * invN, mu[tm], rs[tm] * (g0[tm] * invN));
    }
  }

  // loads are unconditional from clamped addresses; what must not contribute to the sum over
  // n (padded samples, groups past the split) is zeroed in the dU operand by a select
  struct Ops {
    float4 A[2], U[2], B[2];
    bool vs;
  };
  auto load_raw = [&](Ops& o, int g) __attribute__((always_inline)) {
    const int gc = g < gend ? g : gend - 1;
    const int s = gc * a.spw + sh;
    const int sc = s < a.b ? s : a.b - 1;
    o.vs = (g < gend) && (s < a.b);
    const int64_t uo = (int64_t)sc * a.M * a.L + l0;
    const int64_t xb = (int64_t)sc * a.C_src * a.L + l0;
    o.A[0] = ld4(a.dU + uo + aoff0);
    o.A[1] = ld4(a.dU + uo + aoff1);
    if (FOLD) {
      o.U[0] = ld4(a.bn_U + uo + aoff0);
      o.U[1] = ld4(a.bn_U + uo + aoff1);
    }
    o.B[0] = ld4(bsrc0 + xb + boff0);
    o.B[1] = ld4(bsrc1 + xb + boff1);
  };
  // the fold is applied HERE, when the set is consumed: by then the other set's loads are in flight
  auto mma_ops = [&](Ops& o) __attribute__((always_inline)) {
    float4 A[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      float4 v = o.A[tm];
      if (FOLD) {
        v.x = fmaf(cf[tm].x, v.x, fmaf(cf[tm].y, o.U[tm].x, cf[tm].z));
        v.y = fmaf(cf[tm].x, v.y, fmaf(cf[tm].y, o.U[tm].y, cf[tm].z));
        v.z = fmaf(cf[tm].x, v.z, fmaf(cf[tm].y, o.U[tm].z, cf[tm].z));
        v.w = fmaf(cf[tm].x, v.w, fmaf(cf[tm].y, o.U[tm].w, cf[tm].z));
      }
      A[tm] = o.vs ? v : z4;
    }
    if (want_bias) {
      bsum[0] += f4_hsum(A[0]);
      bsum[1] += f4_hsum(A[1]);
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tm].x, o.B[tk].x, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tm].y, o.B[tk].y, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tm].z, o.B[tk].z, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[tm].w, o.B[tk].w, acc[tm][tk], 0, 0, 0);
      }
  };
  // two-set software pipeline, straight-line steady state (groups past the split are clamped +
  // zeroed, so running one extra masked group is harmless).  NB: a variant with `break`s out of a
  // ring loop was MISCOMPILED by hipcc 7.2 (wrong accumulators on the exit paths) — keep exits out of
  // MFMA pipeline loops, predicate the tail instead.
  Ops s0, s1;
  int g = gbeg + wave;
  load_raw(s0, g);
  for (; g + NW < gend; g += 2 * NW) {
    load_raw(s1, g + NW);
    __builtin_amdgcn_sched_barrier(0);       // the loads stay IN FRONT of the other set's MFMAs (hipcc sinks them)
    mma_ops(s0);
    __builtin_amdgcn_sched_barrier(0);
    load_raw(s0, g + 2 * NW);
    __builtin_amdgcn_sched_barrier(0);
    mma_ops(s1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (g < gend) mma_ops(s0);

  // acc[tm][tk][r] = dW[m0 + 16tm + 4h + r][k0 + 16tk + lo]
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[wave][(16 * tm + 4 * h + r) * 33 + 16 * tk + lo] = acc[tm][tk][r];
  if (want_bias) {
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
      bsum[tm] = xor16_sum(bsum[tm]);
      bsum[tm] = xor32_sum(bsum[tm]);
      if (h == 0) brow[wave][16 * tm + lo] = bsum[tm];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 32; e += 64 * NW) {
    const int mm = e >> 5, kk = e & 31;
    const int m = m0 + mm, k = k0 + kk;
    if (m < a.M && k < a.K) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += tile[w][mm * 33 + kk];
      float* pp = a.dW + (int64_t)m * a.ldw + k;
#if defined(BMNAS_PROBE_W_STORE)   // timing only (variant builds): what the weight-gradient class costs WITHOUT its atomics
      if (a.use_atomic) {
        pp[((int64_t)bz * a.M * a.ldw) & 0] = v;
        continue;
      }
#endif
      if (a.use_atomic) {
        atomicAdd(pp, v);
        if (a.dup_cols > 0) atomicAdd(pp + a.dup_cols, v);
      } else {
        *pp += v;
        if (a.dup_cols > 0) pp[a.dup_cols] += v;
      }
    }
  }
  if (want_bias && threadIdx.x < 32) {
    const int m = m0 + threadIdx.x;
    if (m < a.M) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += brow[w][threadIdx.x];
      if (a.use_atomic) atomicAdd(a.dbias + m, v);
      else a.dbias[m] += v;
    }
  }
}

template <int NW>
constexpr size_t conv_w_lds() { return (size_t)NW * (32 * 33 + 32) * sizeof(float); }

__global__ __launch_bounds__(512) void conv_w_k(ConvWArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[conv_w_lds<8>()];
  if (a.bn_U != nullptr) conv_w_body<8, true>(a, blockIdx.x, blockIdx.y, blockIdx.z, lds);
  else conv_w_body<8, false>(a, blockIdx.x, blockIdx.y, blockIdx.z, lds);
}

// The whole backward of a NodeMixedOp's contractions in ONE grid (search mode): attention
// backward workgroups first (long dependent chains), then the data-gradient GEMM tiles, then the
// weight-gradient tiles (4-wave variant of conv_w_k).  All three only read dU / z / g and write
// disjoint outputs.
template <int TN, int TJ, int KCH>
__global__ __launch_bounds__(256) void conv_bwd_all_k(ConvArgs a, SdpaBwdArgs s, ConvWArgs w, int gx,
                                                      int n_w, int wx, int wy) {
  constexpr int KPW = 3 * KCH;
  extern __shared__ __attribute__((aligned(16))) char merged_smem[];
  // (block order: permutations measured at NTU b64 / Ego b48 / NTU b8 change nothing — the grid is small)
  const int blk = blockIdx.x;
  // BMNAS_CONV_PROBE bits 16 / 32 / 64: as in conv_bwd_all_pipe_k (timing diagnostics only)
  if ((a.probe & 16) && blk < s.groups) return;
  if ((a.probe & 32) && blk >= s.groups && blk < s.groups + n_w) return;
  if ((a.probe & 64) && blk >= s.groups + n_w) return;
  if (blk < s.groups) {
    sdpa_bwd_body<KCH>(blk, s.g, s.gscale, s.x, s.y, s.ln_w, s.xhat, s.stats, s.dx, s.dy, s.acc_mask, s.G,
                       s.drop, merged_smem);
  } else if (blk < s.groups + n_w) {
    // the weight-gradient tiles walk many n-groups each (long-running): dispatched before the
    // short data-gradient tiles, which then fill the gaps
    const int t = blk - s.groups;
    const int bz = t / (wx * wy), r = t - bz * wx * wy;
    if (w.bn_U != nullptr) conv_w_body<4, true>(w, r % wx, r / wx, bz, merged_smem);
    else conv_w_body<4, false>(w, r % wx, r / wx, bz, merged_smem);
  } else {
    const int t = blk - s.groups - n_w;
    conv_ksplit_body<false, TN, TJ, KPW>(a, t % gx, t / gx, merged_smem);
  }
}

// The backward of a conv + BatchNorm that has no attention branch beside it (NodeCell's out_conv,
// node_search.py:63-66) at small grids: weight-gradient tiles, then data-gradient tiles, one launch; the
// BatchNorm input gradient is applied while the operands are staged (ConvArgs::bn_U).
template <int KPW>
__global__ __launch_bounds__(256) void conv_bwd_pair_k(ConvArgs a, ConvWArgs w, int gx, int n_w, int wx,
                                                       int wy, MixEp mix) {
  extern __shared__ __attribute__((aligned(16))) char merged_smem[];
  const int blk = blockIdx.x;
  if (blk < n_w) {
    const int bz = blk / (wx * wy), r = blk - bz * wx * wy;
    if (w.bn_U != nullptr) conv_w_body<4, true>(w, r % wx, r / wx, bz, merged_smem);
    else conv_w_body<4, false>(w, r % wx, r / wx, bz, merged_smem);
  } else {
    const int t = blk - n_w;
    if (mix.on) conv_ksplit_body<false, 1, 1, KPW, false, true>(a, t % gx, t / gx, merged_smem, &mix);
    else conv_ksplit_body<false, 1, 1, KPW>(a, t % gx, t / gx, merged_smem);
  }
}

#ifndef BMNAS_PIPE_LA2
#define BMNAS_PIPE_LA2 1
#endif
constexpr bool kLa2 = BMNAS_PIPE_LA2 != 0;
#ifndef BMNAS_MERGED_OCC
#define BMNAS_MERGED_OCC 3
#endif

// (second __launch_bounds__ argument: at least 3 waves per SIMD, i.e. <= 168 VGPRs — the grid is ~750
// workgroups and all of them must be resident at once)
template <int KC, int KCH, int NG>
__global__ __launch_bounds__(256, BMNAS_MERGED_OCC) void conv_bwd_all_pipe_k(ConvArgs a, SdpaBwdArgs s, ConvWArgs w, int gx,
                                                           int n_w, int wx, int wy) {
  extern __shared__ __attribute__((aligned(16))) char merged_smem[];
  // Block order.  The grid (~750 workgroups at MM-IMDB b128: 128 attention groups, 432 weight-gradient tiles, 192
  // data-gradient tiles) is resident all at once, three workgroups per CU, so the order decides only WHERE a block
  // lands — and a data-gradient tile's 12 chunk steps are the longest chain of the launch (~18 us of its ~19).  With
  // those tiles at the front of the grid the dispatcher gives each its own CU; at the back (rounds 2-3) they filled the
  // last free slots, several to a CU: 20.95 -> 19.05 us (profiles/r04_bwd_block_order.txt; BMNAS_BWD_ORDER = 0 ... 5
  // selects A W D / D A W / D W A / W D A / W A D / A D W for that table, default 2).
  const int n_d = (int)gridDim.x - s.groups - n_w;
  int blk = (int)blockIdx.x;                 // -> logical index in [attention | weight gradient | data gradient]
  {
    const int nA = s.groups, nW = n_w, nD = n_d, p = (int)blockIdx.x;
    const int oA = 0, oW = nA, oD = nA + nW;
    switch (a.order) {
      case 1: blk = p < nD ? oD + p : (p < nD + nA ? oA + p - nD : oW + p - nD - nA); break;            // D A W
      case 2: blk = p < nD ? oD + p : (p < nD + nW ? oW + p - nD : oA + p - nD - nW); break;            // D W A
      case 3: blk = p < nW ? oW + p : (p < nW + nD ? oD + p - nW : oA + p - nW - nD); break;            // W D A
      case 4: blk = p < nW ? oW + p : (p < nW + nA ? oA + p - nW : oD + p - nW - nA); break;            // W A D
      case 5: blk = p < nA ? oA + p : (p < nA + nD ? oD + p - nA : oW + p - nA - nD); break;            // A D W
      default: break;                                                                                   // A W D
    }
  }
  // BMNAS_CONV_PROBE bits 16 / 32 / 64 drop the attention / weight-gradient / data-gradient blocks:
  // timing diagnostics only (tools/ktable.py), the results are then incomplete
  if ((a.probe & 16) && blk < s.groups) return;
  if ((a.probe & 32) && blk >= s.groups && blk < s.groups + n_w) return;
  if ((a.probe & 64) && blk >= s.groups + n_w) return;
  if (blk < s.groups) {
    BMNAS_SETPRIO(BMNAS_PRIO_BWD_A);
    sdpa_bwd_body<KCH>(blk, s.g, s.gscale, s.x, s.y, s.ln_w, s.xhat, s.stats, s.dx, s.dy, s.acc_mask, s.G,
                       s.drop, merged_smem);
  } else if (blk < s.groups + n_w) {
    BMNAS_SETPRIO(BMNAS_PRIO_BWD_W);
    const int t = blk - s.groups;
    int bx, by, bz;
    if (wx < 0) {
      // XCD-aware order (workgroups go round-robin over the 8 XCDs, each with its own L2): XCD x owns
      // batch split x / 2 and one half of the output-channel tiles, so its L2 fetches 1/8 of dU (and U)
      // and 1/4 of the sources instead of every XCD fetching all of them.  Speed only: any
      // placement gives the same tiles.
      const int hw = (-wx) >> 1, x = blockIdx.x & 7, q = t >> 3;
      bz = x >> 1;
      bx = (x & 1) * hw + q % hw;
      by = q / hw;
    } else {
      bz = t / (wx * wy);
      const int r = t - bz * wx * wy;
      bx = r % wx;
      by = r / wx;
    }
    if (w.bn_U != nullptr) conv_w_body<4, true>(w, bx, by, bz, merged_smem);
    else conv_w_body<4, false>(w, bx, by, bz, merged_smem);
  } else {
    const int t = blk - s.groups - n_w;
    int bx = t % gx, by = t / gx;
    if ((a.order == 1 || a.order == 2) && (gx & 7) == 0) {
      // XCD-aware tile order (tiles at the front of the grid: workgroup p runs on XCD p & 7): XCD x takes the n-tiles
      // [x gx / 8, (x + 1) gx / 8) — the batch quarter x / 2 whose dU / U its weight-gradient tiles read too (above) —
      // with every j-tile of an n-tile on the same XCD.  Speed only (18.44 -> 18.30 us).
      const int hw = gx >> 3, x = blockIdx.x & 7, i = t >> 3;
      bx = x * hw + i % hw;
      by = i / hw;
    }
    BMNAS_SETPRIO(BMNAS_PRIO_BWD_D);
    if (a.bn_U != nullptr) conv_pipe_bwd_body<KC, NG, true, kLa2>(a, bx, by, reinterpret_cast<float*>(merged_smem));
    else conv_pipe_bwd_body<KC, NG, false>(a, bx, by, reinterpret_cast<float*>(merged_smem));
  }
}

// ---- grouped launches: the N reshape layers in front of the fusion cell ------------------------------------
// (ReshapeInputLayer / _MMIMDB, aux_models.py:51-76, 87-115: N independent Conv1d(C_in_i -> C, k = 1) + BatchNorm
// on N modality tensors of one (b, L).)  One launch per layer left 6-8 short launches in a row, each on a fraction
// of the chip (MM-IMDB b = 128: 6 x 11.7 us forward, 10 x 9 us backward); here every layer's tiles share ONE
// grid.  Block ranges per problem; the bodies are the single-conv ones: pipelined LDS tiles where a layer has
// enough of them, the multi-round split-K body otherwise (any K, 16 x 16 tiles).
constexpr int kGroupMax = 8;
struct ConvFwdGroup {
  ConvArgs a[kGroupMax];
  int start[kGroupMax + 1];      // first block of problem p (start[n] = grid size)
  int gx[kGroupMax];
  int kind[kGroupMax];           // 0: conv_pipe_fwd_body<32, 2>; 1: conv_ksplit_body<true, 1, 1, 12, MULTI>
  int n;
};

// problem of this block: the last p with start[p] <= blockIdx.x (wave-uniform by construction; readfirstlane
// says so to the compiler)
__device__ __forceinline__ int group_problem(const int (&start)[kGroupMax + 1], int n) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < kGroupMax; ++q) p += (q < n && (int)blockIdx.x >= start[q]) ? 1 : 0;
  return __builtin_amdgcn_readfirstlane(p);
}
// arr[p] of a kernel-argument array WITHOUT indexing it at run time: a run-time index makes clang copy the whole
// by-value argument struct to scratch and read it back with vector loads (the pointers then sit in VGPRs, which
// the SGPR pointer selects of the bodies refuse); a chain over the compile-time-indexed elements keeps every
// field a scalar load
template <typename T, int N>
__device__ __forceinline__ T pick_uniform(const T (&arr)[N], int p) {
  T v = arr[0];
#pragma unroll
  for (int q = 1; q < N; ++q)
    if (p == q) v = arr[q];
  return v;
}
// ... and tell the compiler that what came out is wave-uniform (it is: p is): every field back into SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uni(T* ptr) {
  const uint64_t u = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ ConvArgs uni(ConvArgs a) {
#pragma unroll
  for (int q = 0; q < kConvPtrs; ++q) {
    a.act.p[q] = uni(a.act.p[q]);
    a.dst.p[q] = uni(a.dst.p[q]);
  }
  a.W = uni(a.W); a.bias = uni(a.bias); a.part = uni(a.part); a.stat = uni(a.stat);
  a.stat_shards = uni(a.stat_shards);
  a.bn_U = uni(a.bn_U); a.bn_chan = uni(a.bn_chan); a.bn_grad = uni(a.bn_grad); a.bn_train = uni(a.bn_train);
  a.ldw = uni(a.ldw); a.Ci = uni(a.Ci); a.Cj = uni(a.Cj); a.I = uni(a.I); a.J = uni(a.J);
  a.b = uni(a.b); a.L = uni(a.L); a.Lb = uni(a.Lb); a.spw = uni(a.spw); a.n_groups = uni(a.n_groups);
  a.n_part = uni(a.n_part); a.acc_mask = (uint32_t)uni((int)a.acc_mask); a.fold = uni(a.fold); a.probe = uni(a.probe);
  return a;
}
__device__ __forceinline__ ConvWArgs uni(ConvWArgs a) {
  a.bn_U = uni(a.bn_U); a.bn_chan = uni(a.bn_chan); a.bn_grad = uni(a.bn_grad); a.bn_train = uni(a.bn_train);
  a.dU = uni(a.dU);
#pragma unroll
  for (int q = 0; q < kConvPtrs; ++q) a.src.p[q] = uni(a.src.p[q]);
  a.dW = uni(a.dW); a.dbias = uni(a.dbias);
  a.ldw = uni(a.ldw); a.C_src = uni(a.C_src); a.M = uni(a.M); a.K = uni(a.K); a.dup_cols = uni(a.dup_cols);
  a.b = uni(a.b); a.L = uni(a.L); a.Lb = uni(a.Lb); a.spw = uni(a.spw); a.n_groups = uni(a.n_groups);
  a.groups_per_split = uni(a.groups_per_split); a.use_atomic = uni(a.use_atomic);
  return a;
}

__global__ __launch_bounds__(256) void conv_fwd_group_k(ConvFwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) char group_smem[];
  const int p = group_problem(G.start, G.n);
  const ConvArgs a = uni(pick_uniform(G.a, p));
  const int t = blockIdx.x - uni(pick_uniform(G.start, p)), gx = uni(pick_uniform(G.gx, p));
  if (uni(pick_uniform(G.kind, p)) == 0) conv_pipe_fwd_body<32, 2>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
  else conv_ksplit_body<true, 1, 1, 12, true>(a, t % gx, t / gx, group_smem);
}

// the same group when every layer takes the pipelined tiles and the longest contraction is a chain of >= 32 chunk steps
// on about one tile per CU: workgroups of NQ quads that split each tile's contraction (conv_pipe_fwd_body<.., NQ>)
template <int NQ>
__global__ __launch_bounds__(256 * NQ) void conv_fwd_group_q_k(ConvFwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) char group_smem[];
  const int p = group_problem(G.start, G.n);
  const ConvArgs a = uni(pick_uniform(G.a, p));
  const int t = blockIdx.x - uni(pick_uniform(G.start, p)), gx = uni(pick_uniform(G.gx, p));
  const int kind = uni(pick_uniform(G.kind, p));               // here: 0 / 2 / 3 = tiles 96 / 64 / 32 columns wide
  // (the 96-column body needs 141 VGPRs; a 1024-thread workgroup has 128 per lane: four quads take the narrow tiles
  // only — the launcher falls back to two quads when a layer of the group wants 96-column tiles — instead of spilling
  // 44 B / lane in every instantiation of the kernel)
  if (NQ <= 2 && kind == 0) conv_pipe_fwd_body<32, 2, (NQ <= 2 ? NQ : 2), 3>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
  else if (kind == 2) conv_pipe_fwd_body<32, 2, NQ, 2>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
  else conv_pipe_fwd_body<32, 2, NQ, 1>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
}

// backward of the group: every layer's weight-gradient tiles (first: they walk many n-groups each), then every
// layer's data-gradient tiles, the BatchNorm input gradient applied while the operands are staged (bn_U).
struct ConvBwdGroup {
  ConvArgs a[kGroupMax];
  ConvWArgs w[kGroupMax];
  int wstart[kGroupMax + 1];     // weight-gradient blocks of problem p
  int dstart[kGroupMax + 1];     // data-gradient blocks (after all weight-gradient blocks); empty: no input gradient
  int wx[kGroupMax], wy[kGroupMax];
  int gx[kGroupMax];
  int kind[kGroupMax];           // 0 / 1: conv_pipe_bwd_body<48 / 32, 2>; 3 / 6: conv_ksplit_body<false, 1, 1, KPW>
  int n, n_w;
  int probe;                     // BMNAS_CONV_PROBE bits 32 / 64: drop the weight- / data-gradient blocks (timing only)
};

__global__ __launch_bounds__(256, 3) void conv_bwd_group_k(ConvBwdGroup G) {
  extern __shared__ __attribute__((aligned(16))) char group_smem[];
  const int n_w = G.n_w;
  if ((G.probe & 32) && (int)blockIdx.x < n_w) return;
  if ((G.probe & 64) && (int)blockIdx.x >= n_w) return;
  if ((int)blockIdx.x < n_w) {
    const int p = group_problem(G.wstart, G.n);
    const ConvWArgs w = uni(pick_uniform(G.w, p));
    const int t = blockIdx.x - uni(pick_uniform(G.wstart, p)), wx = uni(pick_uniform(G.wx, p)), wy = uni(pick_uniform(G.wy, p));
    const int bz = t / (wx * wy), r = t - bz * wx * wy;
    if (w.bn_U != nullptr) conv_w_body<4, true>(w, r % wx, r / wx, bz, group_smem);
    else conv_w_body<4, false>(w, r % wx, r / wx, bz, group_smem);
  } else {
    const int p = group_problem(G.dstart, G.n);
    const ConvArgs a = uni(pick_uniform(G.a, p));
    const int t = blockIdx.x - uni(pick_uniform(G.dstart, p)), gx = uni(pick_uniform(G.gx, p)), kind = uni(pick_uniform(G.kind, p));
    if (kind == 0) {
      if (a.bn_U != nullptr) conv_pipe_bwd_body<48, 2, true, kLa2>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
      else conv_pipe_bwd_body<48, 2, false>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
    } else if (kind == 1) {
      if (a.bn_U != nullptr) conv_pipe_bwd_body<32, 2, true, kLa2>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
      else conv_pipe_bwd_body<32, 2, false>(a, t % gx, t / gx, reinterpret_cast<float*>(group_smem));
    } else if (kind == 3) {
      conv_ksplit_body<false, 1, 1, 3>(a, t % gx, t / gx, group_smem);
    } else {
      conv_ksplit_body<false, 1, 1, 6>(a, t % gx, t / gx, group_smem);
    }
  }
}

__global__ __launch_bounds__(256) void fold_weight_k(const float* __restrict__ W,
                                                     float* __restrict__ Weff, int M, int C) {
  const int c4n = C / 4;
  const int64_t total = (int64_t)M * c4n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / c4n), c4 = (int)(i - (int64_t)m * c4n);
    const float* r = W + (int64_t)m * 2 * C + 4 * c4;
    st4_wt(Weff + (int64_t)m * C + 4 * c4, f4_add(ld4(r), ld4(r + C)));
  }
}

// Which GEMM family a call was dispatched to (diagnostics for tests/test_dispatch_gpu.py: host-side
// counters, never read by a kernel).  Order = bmnas_conv_family_name().
enum ConvFamily { F_KSPLIT, F_PIPE_FWD, F_PIPE_BWD, F_LDS, F_FWD_SDPA_PIPE, F_FWD_SDPA_KSPLIT,
                  F_BWD_ALL_PIPE, F_BWD_ALL_KSPLIT, F_CONV_W, F_BWD_PAIR, F_FWD_GROUP, F_BWD_GROUP, F_FWD_QUADS_GROUP,
                  F_COUNT };
long g_family_calls[F_COUNT] = {0};
#define BMNAS_COUNT(f) (++g_family_calls[f])

inline int conv_probe() {
#if BMNAS_BODY_PROBES || defined(BMNAS_CLASS_PROBE)   // (-DBMNAS_CLASS_PROBE: the class-drop bits without the in-body stamps)
  static const int v = []() { const char* e = getenv("BMNAS_CONV_PROBE"); return e ? atoi(e) : 0; }();
  return v;
#else
  return 0;       // production builds: a stray BMNAS_CONV_PROBE cannot drop gradient tiles (timing builds only)
#endif
}

inline int check_shape(int b, int L, int* Lb, int* spw, int* n_groups) {
  if (!(L == 4 || L == 8 || L == 16)) return BMNAS_E_SHAPE;
  *Lb = ilog2_exact(L);
  *spw = 16 / L;
  *n_groups = (b + *spw - 1) / *spw;
  return 0;
}

}  // namespace

extern "C" int bmnas_conv1x1_num_partials(int b, int L) {
  int Lb, spw, ng;
  if (b < 1 || check_shape(b, L, &Lb, &spw, &ng)) return BMNAS_E_SHAPE;
  return ng;                                   // one partial per 16-column n-group
}

namespace {
// wave tile = (16*TN) x (16*TJ); pick the largest tile that still gives >= ~1500 waves
template <bool TRANS, int TN, int TJ>
bool launch_ksplit(const ConvArgs& a, hipStream_t st) {
  const int nblk = a.I / 16;
  const int kpw = (nblk + 3) / 4;
  dim3 grid((unsigned)((a.n_groups + TN - 1) / TN), (unsigned)((a.J / 16 + TJ - 1) / TJ));
  // longer contractions (K > 768: the C_in = 1024 / 2048 reshape layers) at small grids: several register
  // rounds in one launch; large grids keep the LDS tile kernels (operand reuse)
  if (TN == 1 && TJ == 1 && kpw > 12 && a.bn_U == nullptr && a.fold == 0 && a.I == a.Ci &&
      (long)grid.x * grid.y <= 512) {
    BMNAS_COUNT(F_KSPLIT);
    hipLaunchKernelGGL((conv_ksplit_multi_k<TRANS>), grid, dim3(256), 0, st, a);
    return true;
  }
  if (kpw * 4 * (TN + TJ) + 4 * TN * TJ > 232) return false;       // operand + accumulator VGPRs
#define KS_CASE(K)                                                                                   \
  if (kpw <= K) {                                                                                    \
    BMNAS_COUNT(F_KSPLIT);                                                                           \
    hipLaunchKernelGGL((conv_ksplit_k<TRANS, TN, TJ, K>), grid, dim3(256), 0, st, a);                 \
    return true;                                                                                     \
  }
  KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4) KS_CASE(6) KS_CASE(9) KS_CASE(12)
#undef KS_CASE
  return false;
}

// merged launch; false when the split-K kernel does not cover this shape (caller launches the
// two kernels separately)
template <int TN, int TJ>
bool launch_ksplit_sdpa_fwd(const ConvArgs& a, const SdpaFwdArgs& s, hipStream_t st) {
  const int kch = sdpa_kch(s.G.C);
  if (a.I != s.G.C || kch > 4) return false;                     // NodeMixedOp shape only
  if (kch * 4 * (TN + TJ) + 4 * TN * TJ > 232) return false;
  const int gx = (a.n_groups + TN - 1) / TN, gy = (a.J / 16 + TJ - 1) / TJ;
  dim3 grid((unsigned)(s.groups + gx * gy));
#define KS_CASE(K)                                                                                   \
  if (kch == K) {                                                                                    \
    BMNAS_COUNT(F_FWD_SDPA_KSPLIT);                                                                  \
    hipLaunchKernelGGL((conv_fwd_sdpa_k<TN, TJ, K>), grid, dim3(256),                                 \
                       std::max((size_t)kSdpaFwdLds, conv_ksplit_lds<TN, TJ>()), st, a, s, gx);       \
    return true;                                                                                     \
  }
  KS_CASE(1) KS_CASE(2) KS_CASE(3) KS_CASE(4)
#undef KS_CASE
  return false;
}

// fewest GEMM workgroups for which the tile kernels are used (tuned in round 2: 96; the data-gradient tiles pay
// from half of that)
inline int conv_pipe_min() { return 96; }

inline int conv_pipe_mode() {        // BMNAS_CONV_PIPE=0 falls back to the split-K kernels (A/B runs)
  static const int v = []() { const char* e = getenv("BMNAS_CONV_PIPE"); return e ? atoi(e) : 1; }();
  return v;
}

// pipelined tile kernel (forward); false when the shape is not covered
inline bool launch_pipe_fwd(const ConvArgs& a, hipStream_t st) {
  if (!conv_pipe_mode() || a.I != a.Ci || a.fold != 0 || a.acc_mask != 0 || a.ldw % 4) return false;
  const int gy = (a.J + kPipeJ - 1) / kPipeJ;
  int gx = (a.n_groups + 3) / 4;
  if (gx * gy < 96) {
    // 32-column tiles when 64-column ones leave CUs idle (reshape layers at batch 128: K = 512 through
    // the split-K kernel ran at 14 % of the MFMA peak, 18 us per layer)
    gx = (a.n_groups + 1) / 2;
    if (gx * gy < 96 || a.I % 32 != 0) return false;
    BMNAS_COUNT(F_PIPE_FWD);
    hipLaunchKernelGGL((conv_pipe_fwd_k<32, 2>), dim3((unsigned)(gx * gy)), dim3(256), (conv_pipe_lds<32, 2>(a.L)), st, a, gx);
    return true;
  }
  if (a.I % 48 == 0 && conv_pipe_lds<48, 4>(a.L) <= 65536) {     // (L = 4 pads rows to twice their size)
    BMNAS_COUNT(F_PIPE_FWD);
    hipLaunchKernelGGL((conv_pipe_fwd_k<48, 4>), dim3((unsigned)(gx * gy)), dim3(256), (conv_pipe_lds<48, 4>(a.L)), st, a, gx);
    return true;
  }
  if (a.I % 32 == 0) {
    BMNAS_COUNT(F_PIPE_FWD);
    hipLaunchKernelGGL((conv_pipe_fwd_k<32, 4>), dim3((unsigned)(gx * gy)), dim3(256), (conv_pipe_lds<32, 4>(a.L)), st, a, gx);
    return true;
  }
  return false;
}

template <bool TRANS>
void launch_gemm(const ConvArgs& a, hipStream_t st) {
  const long jt = a.J / 16, ng = a.n_groups;
  if (TRANS && launch_pipe_fwd(a, st)) return;
  if (!TRANS && conv_pipe_mode() && a.I == a.Ci && a.fold == 0 && a.I % 48 == 0 && a.J % 16 == 0 && a.ldw % 4 == 0) {
    const int gx = (a.n_groups + 1) / 2, gy = (a.J + kPipeBJ - 1) / kPipeBJ;
    if (gx * gy >= 96) {
      BMNAS_COUNT(F_PIPE_BWD);
      hipLaunchKernelGGL((conv_pipe_bwd_k<48, 2>), dim3((unsigned)(gx * gy)), dim3(256),
                         (conv_pipe_bwd_lds<48, 2>(a.L)), st, a, gx);
      return;
    }
  }
  {
    // split-K kernel.  Measured on MI355X: what costs time at batch 128 is the number of
    // ROUNDS of workgroups a CU has to run (each round pays launch + one memory round trip +
    // the store drain, ~3 us), not MFMA or bytes.  So: the largest output tile that still
    // gives every CU a workgroup (>= 256), all of them resident at once.
    // Tile choice is empirical (MI355X, batch 128, K = 192 / 576): 2x2 tiles with >= 1024
    // workgroups 14.4 us, 1x1 tiles 14.6 us; fewer, fatter workgroups were SLOWER (4x4: 22-26 us,
    // 2x2 with 384 workgroups: 18-22 us) — more waves in flight beat operand reuse here.
    auto wgs = [&](long tn, long tj) { return ((ng + tn - 1) / tn) * ((jt + tj - 1) / tj); };
    if (wgs(2, 2) >= 1024 && launch_ksplit<TRANS, 2, 2>(a, st)) return;
    if (launch_ksplit<TRANS, 1, 1>(a, st)) return;
  }
  {
    // generic fallback: whole-K LDS-staged tiles.  No shape of the reference's configurations lands here (the
    // pipelined tile kernels take the large grids, the split-K kernels the small ones); what does: channel
    // counts that are not a multiple of 32 at large grids, a fused cat of two sources with K > 768 at small
    // ones.  Biggest tile that still yields >= ~400 workgroups.
    const long wg64 = ((ng + 3) / 4) * ((jt + 3) / 4), wg3264 = ((ng + 1) / 2) * ((jt + 3) / 4);
    BMNAS_COUNT(F_LDS);
    if (wg64 >= 400) {
      hipLaunchKernelGGL((conv_lds_k<TRANS, 64, 64>), dim3((unsigned)((ng + 3) / 4), (unsigned)((jt + 3) / 4)),
                         dim3(256), 0, st, a);
    } else if (wg3264 >= 400) {
      hipLaunchKernelGGL((conv_lds_k<TRANS, 32, 64>), dim3((unsigned)((ng + 1) / 2), (unsigned)((jt + 3) / 4)),
                         dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((conv_lds_k<TRANS, 32, 32>), dim3((unsigned)((ng + 1) / 2), (unsigned)((jt + 1) / 2)),
                         dim3(256), 0, st, a);
    }
  }
}
}  // namespace

extern "C" int bmnas_conv1x1_fwd(const float* const* srcs, int n_src, int C_src, const float* W,
                                 int ldw, int fold_cols, const float* bias, float* U, float* part,
                                 int stat_shards, int b, int L, int M, void* stream) {
  if (!srcs || !W || !U || n_src < 1 || C_src < 1 || b < 0 || M < 1 || fold_cols < 0 || stat_shards < 0)
    return BMNAS_E_ARG;
  if (fold_cols % 4 || (fold_cols > 0 && ldw < n_src * C_src + fold_cols)) return BMNAS_E_SHAPE;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw % 4 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.act.p[q] = srcs[q];
  }
  a.dst.p[0] = U;
  a.W = W; a.bias = bias; a.ldw = ldw;
  a.part = stat_shards ? nullptr : part; a.stat = stat_shards ? part : nullptr; a.stat_shards = stat_shards;
  a.Ci = C_src; a.I = n_src * C_src; a.Cj = M; a.J = M;
  a.b = b; a.L = L; a.acc_mask = 0; a.n_part = a.n_groups; a.probe = conv_probe(); a.fold = fold_cols;
  launch_gemm<true>(a, (hipStream_t)stream);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_bwd_data(const float* dU, const float* W, int ldw, int fold_cols,
                                      float* const* dsrcs, int n_src, int C_src,
                                      uint32_t accumulate_mask, int b, int L, int M, void* stream) {
  if (!dU || !W || !dsrcs || n_src < 1 || C_src < 1 || b < 0 || M < 1 || fold_cols < 0) return BMNAS_E_ARG;
  if (fold_cols % 4 || (fold_cols > 0 && ldw < n_src * C_src + fold_cols)) return BMNAS_E_SHAPE;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  a.act.p[0] = dU;
  for (int q = 0; q < n_src; ++q) a.dst.p[q] = dsrcs[q];
  a.W = W; a.bias = nullptr; a.part = nullptr; a.ldw = ldw;
  a.Ci = M; a.I = M; a.Cj = C_src; a.J = n_src * C_src;
  a.b = b; a.L = L; a.acc_mask = accumulate_mask; a.probe = conv_probe(); a.fold = fold_cols;
  launch_gemm<false>(a, (hipStream_t)stream);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_fwd_sdpa(const float* const* srcs, int n_src, int C_src, const float* W,
                                      int ldw, int fold_cols, const float* bias, float* U, float* part,
                                      int stat_shards, int b, int L, int M, const float* x, const float* y,
                                      const float* ln_w, const float* ln_b, float* out, float* xhat,
                                      float* stats, int C, bmnas_dropout_t drop, void* stream) {
  if (!srcs || !W || !U || n_src < 1 || C_src < 1 || b < 0 || M < 1 || fold_cols < 0) return BMNAS_E_ARG;
  if (!x || !y || !ln_w || !ln_b || !out || !xhat || !stats) return BMNAS_E_ARG;
  if (fold_cols % 4 || (fold_cols > 0 && ldw < n_src * C_src + fold_cols)) return BMNAS_E_SHAPE;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw % 4 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  SdpaFwdArgs s{};
  if (int e = geom(b, C, L, &s.G)) return e;
  if (b == 0) return 0;
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.act.p[q] = srcs[q];
  }
  a.dst.p[0] = U;
  a.W = W; a.bias = bias; a.ldw = ldw;
  a.part = stat_shards ? nullptr : part; a.stat = stat_shards ? part : nullptr; a.stat_shards = stat_shards;
  a.Ci = C_src; a.I = n_src * C_src; a.Cj = M; a.J = M;
  a.b = b; a.L = L; a.acc_mask = 0; a.n_part = a.n_groups; a.probe = conv_probe(); a.fold = fold_cols;
  s.x = x; s.y = y; s.ln_w = ln_w; s.ln_b = ln_b; s.out = out; s.xhat = xhat; s.stats = stats;
  s.drop = to_cfg(drop);
  s.groups = (b + s.G.spw - 1) / s.G.spw;
  hipStream_t st = (hipStream_t)stream;
  const long jt = a.J / 16, ng = a.n_groups;
  bool done = false;
  {
    const int kch = sdpa_kch(C);
    // 64-column tiles only when they already give every CU two workgroups; else 32-column tiles
    // (MM-IMDB batch 128: 384 instead of 192 GEMM workgroups, 8 us per step faster)
    const int gy = (a.J + kPipeJ - 1) / kPipeJ;
    // (re-measured with the pipelined chunk step, MM-IMDB b128: 64-column tiles 12.5-13.1 us against 11.2)
    const int ngv = (a.n_groups + 3) / 4 * gy >= 512 ? 4 : 2;
    const int gx = (a.n_groups + ngv - 1) / ngv;
    if (conv_pipe_mode() && a.I == C && a.fold == 0 && a.ldw % 4 == 0 && gx * gy >= conv_pipe_min() && kch <= 4) {
      dim3 grid((unsigned)(s.groups + gx * gy));
#define PF_CASE(KCv, K)                                                                                \
  if (!done && a.I % KCv == 0 && kch == K &&                                                           \
      (ngv == 2 ? conv_pipe_lds<KCv, 2>(a.L) : conv_pipe_lds<KCv, 4>(a.L)) <= 65536) {                 \
    BMNAS_COUNT(F_FWD_SDPA_PIPE);                                                                      \
    if (ngv == 2)                                                                                      \
      hipLaunchKernelGGL((conv_pipe_fwd_sdpa_k<KCv, K, 2>), grid, dim3(256),                           \
                         std::max((size_t)kSdpaFwdLds, (conv_pipe_lds<KCv, 2>(a.L))), st, a, s, gx);        \
    else                                                                                               \
      hipLaunchKernelGGL((conv_pipe_fwd_sdpa_k<KCv, K, 4>), grid, dim3(256),                           \
                         std::max((size_t)kSdpaFwdLds, (conv_pipe_lds<KCv, 4>(a.L))), st, a, s, gx);        \
    done = true;                                                                                       \
  }
      // 32-channel chunks first: 38 KB of LDS instead of 55 KB keeps four workgroups on a CU
      // (measured 2.6 us per step faster than 48-channel chunks at K = 192; the data-gradient body
      // showed no such preference and stays at 48)
      PF_CASE(32, 1) PF_CASE(32, 2) PF_CASE(32, 3) PF_CASE(32, 4)
      PF_CASE(48, 1) PF_CASE(48, 2) PF_CASE(48, 3) PF_CASE(48, 4)
#undef PF_CASE
    }
  }
  if (!done && ((ng + 1) / 2) * ((jt + 1) / 2) >= 1024) done = launch_ksplit_sdpa_fwd<2, 2>(a, s, st);
  if (!done) done = launch_ksplit_sdpa_fwd<1, 1>(a, s, st);
  if (!done) {
    launch_gemm<true>(a, st);
    BMNAS_CHECK_LAUNCH();
    return bmnas_sdpa_ln_fwd(x, y, ln_w, ln_b, out, xhat, stats, b, C, L, drop, stream);
  }
  BMNAS_CHECK_LAUNCH();
  return 0;
}

namespace {
int g_conv_deterministic = 0;       // bmnas_conv1x1_set_deterministic
// fill the weight-gradient arguments; waves = waves per workgroup of the kernel that will run them
int fill_w_args(ConvWArgs& a, const float* dU, const float* const* srcs, int n_src, int C_src, float* dW,
                int ldw, float* dbias, int dup_cols, int b, int L, int M, int waves, dim3* grid) {
  if (!dU || !srcs || !dW || n_src < 1 || C_src < 1 || b < 0 || M < 1 || dup_cols < 0)
    return BMNAS_E_ARG;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src + dup_cols) return BMNAS_E_SHAPE;
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.src.p[q] = srcs[q];
  }
  a.dU = dU; a.dW = dW; a.dbias = dbias; a.ldw = ldw; a.C_src = C_src; a.M = M;
  a.K = n_src * C_src; a.dup_cols = dup_cols; a.b = b; a.L = L;
  const int tiles = ((M + 31) / 32) * ((a.K + 31) / 32);
  // splits of the batch: as many as keep the grid at <= ~256 eight-wave workgroups (measured at
  // 108 tiles: 2 splits 9.1 us, 3: 10.9, 4: 9.8, 1: 16.8 — every extra split is another round of
  // fp32 atomics on dW), at most 8 n-groups per wave per split, never fewer than one
  int splits = (256 * 8 / waves) / tiles;
  const int max_splits = (a.n_groups + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1 || g_conv_deterministic) splits = 1;           // deterministic mode: one walk over the batch, no atomics
  a.groups_per_split = (a.n_groups + splits - 1) / splits;
  splits = (a.n_groups + a.groups_per_split - 1) / a.groups_per_split;
  a.use_atomic = splits > 1;
  *grid = dim3((M + 31) / 32, (a.K + 31) / 32, splits);
  return 0;
}
}  // namespace

extern "C" int bmnas_conv1x1_bwd_weight(const float* dU, const float* const* srcs, int n_src,
                                        int C_src, float* dW, int ldw, float* dbias, int dup_cols,
                                        int b, int L, int M, void* stream) {
  ConvWArgs a{};
  dim3 grid;
  if (int e = fill_w_args(a, dU, srcs, n_src, C_src, dW, ldw, dbias, dup_cols, b, L, M, 8, &grid)) return e;
  if (b == 0) return 0;
  BMNAS_COUNT(F_CONV_W);
  hipLaunchKernelGGL(conv_w_k, grid, dim3(512), 0, (hipStream_t)stream, a);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_bn_bwd_apply(float* dV, const float* U, const float* chan, const float* bn_grad, int b,
                                  int M, int L, int training, void* stream);

extern "C" int bmnas_conv1x1_bwd_all_sdpa(const float* dU, const float* W, int ldw, int fold_cols,
                                          float* const* dsrcs, int n_src, int C_src,
                                          uint32_t accumulate_mask, int b, int L, int M,
                                          const float* const* wsrcs, float* dW, int ldw_grad,
                                          float* dbias, int dup_cols, const float* g,
                                          const float* gscale, const float* x, const float* y,
                                          const float* ln_w, const float* xhat, const float* stats,
                                          float* dx, float* dy, uint32_t sdpa_accumulate_mask, int C,
                                          bmnas_dropout_t drop, const float* bn_U, const float* bn_chan,
                                          const float* bn_grad, int bn_training, void* stream) {
  if (bn_U != nullptr && (!bn_chan || (bn_training && !bn_grad))) return BMNAS_E_ARG;
  if (!dU || !W || !dsrcs || n_src < 1 || C_src < 1 || b < 0 || M < 1 || fold_cols < 0) return BMNAS_E_ARG;
  if (!g || !x || !y || !ln_w || !xhat || !stats || !dx) return BMNAS_E_ARG;
  if (fold_cols % 4 || (fold_cols > 0 && ldw < n_src * C_src + fold_cols)) return BMNAS_E_SHAPE;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  for (int q = 0; q < n_src; ++q)
    if (dsrcs[q] == dx || (dy && dsrcs[q] == dy)) return BMNAS_E_ARG;
  ConvWArgs w{};
  dim3 wgrid(0, 1, 1);
  // dW == NULL: nobody wants the weight / bias gradients (the architecture step of the search loop differentiates
  // alpha / beta / gamma only): the launch then carries no weight-gradient tiles
  if (dW != nullptr) {
    if (int e = fill_w_args(w, dU, wsrcs, n_src, C_src, dW, ldw_grad, dbias, dup_cols, b, L, M, 4, &wgrid))
      return e;
  }
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  SdpaBwdArgs s{};
  if (int e = geom(b, C, L, &s.G)) return e;
  if (b == 0) return 0;
  a.act.p[0] = dU;
  for (int q = 0; q < n_src; ++q) a.dst.p[q] = dsrcs[q];
  a.W = W; a.bias = nullptr; a.part = nullptr; a.ldw = ldw;
  a.Ci = M; a.I = M; a.Cj = C_src; a.J = n_src * C_src;
  a.b = b; a.L = L; a.acc_mask = accumulate_mask; a.probe = conv_probe(); a.fold = fold_cols;
  s.g = g; s.gscale = gscale; s.x = x; s.y = y; s.ln_w = ln_w; s.xhat = xhat; s.stats = stats;
  s.dx = dx; s.dy = dy; s.acc_mask = sdpa_accumulate_mask; s.drop = to_cfg(drop);
  s.groups = (b + s.G.spw - 1) / s.G.spw;
  hipStream_t st = (hipStream_t)stream;
  const int kch = sdpa_kch(C);
  bool done = false;
  const int gy = (a.J + kPipeBJ - 1) / kPipeBJ;
  const int ngv = 2;                                  // 32 x 64 data-gradient tiles (16 x 64 measured slower: +4 us per step)
  const int gx = (a.n_groups + ngv - 1) / ngv;
  const bool pipe_ok = conv_pipe_mode() && a.I % 48 == 0 && a.fold == 0 && a.ldw % 4 == 0 && a.J % 16 == 0 &&
                       kch <= 4 && gx * gy >= conv_pipe_min() / 2;   // measured: pays from ~48 data-gradient tiles up
  // the split-K merged kernel (small grids): same conditions as its branch below
  const long jt_ = a.J / 16, ng_ = a.n_groups;
  const bool ks_big = ((ng_ + 1) / 2) * ((jt_ + 1) / 2) >= 1024 && 3 * kch * 16 + 16 <= 232;
  const int ks_tn = ks_big ? 2 : 1;
  const bool ks_ok = !pipe_ok && a.I == 3 * C && C % 64 == 0 && kch <= 4 &&
                     3 * kch * 4 * (2 * ks_tn) + 4 * ks_tn * ks_tn <= 232 &&
                     3 * kch * 4 * (3 * ks_tn) + 4 * ks_tn * ks_tn <= 232;   // (+ the raw-output registers of the fold)
  if (bn_U != nullptr) {
    if (pipe_ok || ks_ok) {                         // the GEMM kernels apply the BatchNorm backward on the fly
      a.bn_U = w.bn_U = bn_U; a.bn_chan = w.bn_chan = bn_chan; a.bn_grad = w.bn_grad = bn_grad;
      a.bn_train = w.bn_train = bn_training;
    } else {                                        // other kernel families: as its own launch, in place
      if (int e = bmnas_bn_bwd_apply(const_cast<float*>(dU), bn_U, bn_chan, bn_grad, b, M, L, bn_training, stream))
        return e;
    }
  }
  if (pipe_ok) {
    {
      const int n_w = (int)(wgrid.x * wgrid.y * wgrid.z);
      static const int order = [] { const char* e = getenv("BMNAS_BWD_ORDER"); return e ? atoi(e) : 2; }();
      a.order = order;
      dim3 grid((unsigned)(s.groups + n_w + gx * gy));
      const size_t lds = std::max(std::max(sdpa_bwd_lds(C), conv_w_lds<4>()),
                                  (conv_pipe_bwd_lds<48, 2>(a.L)) + (a.bn_U ? (size_t)a.I * sizeof(float4) : 0));
    // negative wx = XCD-aware weight-gradient tile order (see the kernel)
    const int wxa = (wgrid.z == 4 && wgrid.x % 2 == 0) ? -(int)wgrid.x : (int)wgrid.x;
#define PB_CASE(K)                                                                                     \
  if (!done && kch == K) {                                                                             \
    BMNAS_COUNT(F_BWD_ALL_PIPE);                                                                       \
    hipLaunchKernelGGL((conv_bwd_all_pipe_k<48, K, 2>), grid, dim3(256), lds, st, a, s, w, gx, n_w,    \
                       wxa, (int)wgrid.y);                                                             \
    done = true;                                                                                       \
  }
      PB_CASE(1) PB_CASE(2) PB_CASE(3) PB_CASE(4)
#undef PB_CASE
    }
  }
  if (!done && a.I == 3 * C && C % 64 == 0 && kch <= 4) {
    const long jt = a.J / 16, ng = a.n_groups;
    const bool big = ((ng + 1) / 2) * ((jt + 1) / 2) >= 1024 && 3 * kch * 16 + 16 <= 232;
    const int TNv = big ? 2 : 1;
    if (3 * kch * 4 * (2 * TNv) + 4 * TNv * TNv <= 232) {
      const int gx = (a.n_groups + TNv - 1) / TNv, gy = (a.J / 16 + TNv - 1) / TNv;
      const int n_data = gx * gy, n_w = (int)(wgrid.x * wgrid.y * wgrid.z);
      dim3 grid((unsigned)(s.groups + n_data + n_w));
      const size_t lds = std::max(std::max(sdpa_bwd_lds(C), conv_w_lds<4>()),
                                  conv_ksplit_lds<2, 2>() + (a.bn_U ? (size_t)a.I * sizeof(float4) : 0));
#define ALL_CASE(T, K)                                                                                 \
  if (!done && TNv == T && kch == K) {                                                                 \
    BMNAS_COUNT(F_BWD_ALL_KSPLIT);                                                                     \
    hipLaunchKernelGGL((conv_bwd_all_k<T, T, K>), grid, dim3(256), lds, st, a, s, w, gx, n_w,          \
                       (int)wgrid.x, (int)wgrid.y);                                                    \
    done = true;                                                                                       \
  }
      ALL_CASE(1, 1) ALL_CASE(1, 2) ALL_CASE(1, 3) ALL_CASE(1, 4)
      ALL_CASE(2, 1) ALL_CASE(2, 2) ALL_CASE(2, 3) ALL_CASE(2, 4)
#undef ALL_CASE
    }
  }
  if (!done) {                                       // shape outside the merged kernels: three launches
    if (int e = bmnas_conv1x1_bwd_data(dU, W, ldw, fold_cols, dsrcs, n_src, C_src, accumulate_mask, b, L, M,
                                       stream))
      return e;
    if (int e = bmnas_sdpa_ln_bwd(g, gscale, x, y, ln_w, xhat, stats, dx, dy, sdpa_accumulate_mask, b, C, L,
                                  drop, stream))
      return e;
    if (dW == nullptr) return 0;
    return bmnas_conv1x1_bwd_weight(dU, wsrcs, n_src, C_src, dW, ldw_grad, dbias, dup_cols, b, L, M, stream);
  }
  BMNAS_CHECK_LAUNCH();
  return 0;
}

namespace {
// the grid rule of bmnas_conv1x1_bwd_all's one-launch form (the only one that can carry the mix epilogue)
inline bool bwd_all_merged(int b, int L, int M, int J) {
  int Lb, spw, ng;
  if (check_shape(b, L, &Lb, &spw, &ng)) return false;
  const long jt = J / 16;
  const int kpw = (M / 16 + 3) / 4;
  const bool pipe = conv_pipe_mode() && M % 48 == 0 && (((long)ng + 1) / 2) * ((J + kPipeBJ - 1) / kPipeBJ) >= 96;
  return !pipe && (((long)ng + 1) / 2) * ((jt + 1) / 2) < 1024 && kpw <= 4 && kpw * 4 * 3 + 4 <= 232;
}
}  // namespace

extern "C" int bmnas_conv1x1_bwd_all_mix_ok(int b, int L, int M, int n_src, int C_src) {
  return b >= 1 && n_src >= 1 && C_src % 16 == 0 && M % 16 == 0 && bwd_all_merged(b, L, M, n_src * C_src);
}

extern "C" int bmnas_conv1x1_bwd_all_mix(const float* dU, const float* W, int ldw, int fold_cols,
                                         float* const* dsrcs, int n_src, int C_src, uint32_t accumulate_mask,
                                         int b, int L, int M, const float* const* wsrcs, float* dW,
                                         int ldw_grad, float* dbias, int dup_cols, const float* bn_U,
                                         const float* bn_chan, const float* bn_grad, int bn_training,
                                         const bmnas_mix_ep_t* mix, void* stream);

extern "C" int bmnas_conv1x1_bwd_all(const float* dU, const float* W, int ldw, int fold_cols,
                                     float* const* dsrcs, int n_src, int C_src, uint32_t accumulate_mask,
                                     int b, int L, int M, const float* const* wsrcs, float* dW,
                                     int ldw_grad, float* dbias, int dup_cols, const float* bn_U,
                                     const float* bn_chan, const float* bn_grad, int bn_training,
                                     void* stream) {
  return bmnas_conv1x1_bwd_all_mix(dU, W, ldw, fold_cols, dsrcs, n_src, C_src, accumulate_mask, b, L, M, wsrcs, dW,
                                   ldw_grad, dbias, dup_cols, bn_U, bn_chan, bn_grad, bn_training, nullptr, stream);
}

extern "C" int bmnas_conv1x1_bwd_all_mix(const float* dU, const float* W, int ldw, int fold_cols,
                                         float* const* dsrcs, int n_src, int C_src, uint32_t accumulate_mask,
                                         int b, int L, int M, const float* const* wsrcs, float* dW,
                                         int ldw_grad, float* dbias, int dup_cols, const float* bn_U,
                                         const float* bn_chan, const float* bn_grad, int bn_training,
                                         const bmnas_mix_ep_t* mix, void* stream) {
  if (bn_U != nullptr && (!bn_chan || (bn_training && !bn_grad))) return BMNAS_E_ARG;
  MixEp me{};
  if (mix != nullptr) {
    if (!mix->U || !mix->chan || !mix->x || !mix->p1 || !mix->gamma || !mix->dx || !mix->dV || !mix->bn_grad ||
        mix->q < 0 || mix->q >= n_src || mix->dgamma_shards < 1 || !dsrcs || !dsrcs[mix->q] || fold_cols != 0)
      return BMNAS_E_ARG;
    if (!bmnas_conv1x1_bwd_all_mix_ok(b > 0 ? b : 1, L, M, n_src, C_src)) return BMNAS_E_LIMIT;
    me.U = mix->U; me.chan = mix->chan; me.x = mix->x; me.p1 = mix->p1; me.gamma = mix->gamma;
    me.dgamma = mix->dgamma; me.dx = mix->dx; me.dV = mix->dV; me.bn_grad = mix->bn_grad;
    me.dg_stride = mix->dgamma_shard_stride; me.dg_shards = mix->dgamma_shards; me.acc_dx = mix->accumulate_dx;
    me.q = mix->q; me.on = 1; me.dglu = to_cfg(mix->drop_glu); me.dfc = to_cfg(mix->drop_fc);
  }
  if (!dU || !W || !dsrcs || n_src < 1 || C_src < 1 || b < 0 || M < 1 || fold_cols < 0) return BMNAS_E_ARG;
  if (fold_cols % 4 || (fold_cols > 0 && ldw < n_src * C_src + fold_cols)) return BMNAS_E_SHAPE;
  if (n_src > kConvPtrs) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvWArgs w{};
  dim3 wgrid;
  if (int e = fill_w_args(w, dU, wsrcs, n_src, C_src, dW, ldw_grad, dbias, dup_cols, b, L, M, 4, &wgrid))
    return e;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  a.act.p[0] = dU;
  for (int q = 0; q < n_src; ++q) a.dst.p[q] = dsrcs[q];
  a.W = W; a.bias = nullptr; a.part = nullptr; a.ldw = ldw;
  a.Ci = M; a.I = M; a.Cj = C_src; a.J = n_src * C_src;
  a.b = b; a.L = L; a.acc_mask = accumulate_mask; a.probe = conv_probe(); a.fold = fold_cols;
  hipStream_t st = (hipStream_t)stream;
  // merged where bmnas_conv1x1_bwd_data would take the 1x1-tile split-K kernel anyway (small grids:
  // every launch there is at the ~4.5 us floor, so two launches fewer is the whole gain)
  const long jt = a.J / 16, ng = a.n_groups;
  const int kpw = (a.I / 16 + 3) / 4;
  const bool pipe = conv_pipe_mode() && a.fold == 0 && a.I % 48 == 0 && a.ldw % 4 == 0 &&
                    ((ng + 1) / 2) * ((a.J + kPipeBJ - 1) / kPipeBJ) >= 96;
  const bool merged = !pipe && ((ng + 1) / 2) * ((jt + 1) / 2) < 1024 && kpw <= 4 &&
                      kpw * 4 * 3 + 4 <= 232;
  bool want_data = false;
  for (int q = 0; q < n_src; ++q) want_data = want_data || dsrcs[q] != nullptr;
  if (mix != nullptr && !(merged && want_data)) return BMNAS_E_LIMIT;     // (host-checked by the caller: _mix_ok)
  if (!merged || !want_data) {
    // separate launches.  The pipelined data-gradient kernel and the weight-gradient kernel both apply the
    // BatchNorm input gradient while staging their operands, so only the other data-gradient families
    // (whole-K LDS / direct kernels of the K = 2048 reshape layers) need the in-place launch first
    const bool fold_here = bn_U != nullptr && (pipe || !want_data);
    if (bn_U != nullptr && !fold_here)
      if (int e = bmnas_bn_bwd_apply(const_cast<float*>(dU), bn_U, bn_chan, bn_grad, b, M, L, bn_training, stream))
        return e;
    if (want_data) {
      if (fold_here) {
        a.bn_U = bn_U; a.bn_chan = bn_chan; a.bn_grad = bn_grad; a.bn_train = bn_training;
        const int pgx = (a.n_groups + 1) / 2, pgy = (a.J + kPipeBJ - 1) / kPipeBJ;
        BMNAS_COUNT(F_PIPE_BWD);
        hipLaunchKernelGGL((conv_pipe_bwd_k<48, 2>), dim3((unsigned)(pgx * pgy)), dim3(256),
                           (conv_pipe_bwd_lds<48, 2>(a.L)) + (size_t)a.I * sizeof(float4), st, a, pgx);
        BMNAS_CHECK_LAUNCH();
      } else if (int e = bmnas_conv1x1_bwd_data(dU, W, ldw, fold_cols, dsrcs, n_src, C_src, accumulate_mask, b,
                                                L, M, stream)) {
        return e;
      }
    }
    ConvWArgs w8{};
    dim3 grid8;
    if (int e = fill_w_args(w8, dU, wsrcs, n_src, C_src, dW, ldw_grad, dbias, dup_cols, b, L, M, 8, &grid8))
      return e;
    if (fold_here) {
      w8.bn_U = bn_U; w8.bn_chan = bn_chan; w8.bn_grad = bn_grad; w8.bn_train = bn_training;
    }
    BMNAS_COUNT(F_CONV_W);
    hipLaunchKernelGGL(conv_w_k, grid8, dim3(512), 0, st, w8);
    BMNAS_CHECK_LAUNCH();
    return 0;
  }
  if (bn_U != nullptr) {
    a.bn_U = w.bn_U = bn_U; a.bn_chan = w.bn_chan = bn_chan; a.bn_grad = w.bn_grad = bn_grad;
    a.bn_train = w.bn_train = bn_training;
  }
  const int gx = (int)ng, gy = (int)jt;
  const int n_w = (int)(wgrid.x * wgrid.y * wgrid.z);
  dim3 grid((unsigned)(n_w + gx * gy));
  const size_t lds = std::max(conv_w_lds<4>(), conv_ksplit_lds<1, 1>() + (a.bn_U ? (size_t)a.I * sizeof(float4) : 0));
  BMNAS_COUNT(F_BWD_PAIR);
#define BP_CASE(K)                                                                                     \
  case K:                                                                                              \
    hipLaunchKernelGGL((conv_bwd_pair_k<K>), grid, dim3(256), lds, st, a, w, gx, n_w, (int)wgrid.x,    \
                       (int)wgrid.y, me);                                                              \
    break;
  switch (kpw) { BP_CASE(1) BP_CASE(2) BP_CASE(3) BP_CASE(4) default: return BMNAS_E_LIMIT; }
#undef BP_CASE
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_group_ok(int n, const int* C_in, int b, int L, int M) {
  if (n < 1 || n > kGroupMax || !C_in || b < 1 || M < 16 || M % 16 || M > 384) return 0;
  int Lb, spw, ng;
  if (check_shape(b, L, &Lb, &spw, &ng)) return 0;
  for (int p = 0; p < n; ++p)
    if (C_in[p] < 16 || C_in[p] % 16) return 0;
  return 1;
}

extern "C" int bmnas_conv1x1_fwd_group(const bmnas_conv_fwd_prob_t* probs, int n, int stat_shards, int b, int L,
                                       int M, void* stream) {
  if (!probs || n < 1 || b < 0 || M < 1 || stat_shards < 0) return BMNAS_E_ARG;
  if (n > kGroupMax) return BMNAS_E_LIMIT;
  if (M % 16) return BMNAS_E_SHAPE;
  ConvFwdGroup G{};
  G.n = n;
  size_t lds = conv_ksplit_lds<1, 1>();
  int blocks = 0;
  // longest contraction first: its tiles run longest
  int order[kGroupMax];
  for (int p = 0; p < n; ++p) order[p] = p;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && probs[order[j]].C_in > probs[order[j - 1]].C_in; --j) std::swap(order[j], order[j - 1]);
  // tile family, decided for the GROUP: the pipelined 32 x 96 tiles reuse their operands 3-6x better than the
  // 16 x 16 split-K tiles and the group as a whole fills the chip with them from ~96 tiles up, even where a
  // single layer would not (NTU at 64 samples: 32 tiles per layer, 256 for the group)
  int Lb0, spw0, ng0;
  if (int e = check_shape(b > 0 ? b : 1, L, &Lb0, &spw0, &ng0)) return e;
  long pipe_tiles = 0;
  for (int q = 0; q < n; ++q)
    if (probs[q].C_in % 32 == 0) pipe_tiles += (long)((ng0 + 1) / 2) * ((M + kPipeJ - 1) / kPipeJ);
  const bool use_pipe = pipe_tiles >= 96;
  for (int q = 0; q < n; ++q) {
    const bmnas_conv_fwd_prob_t& P = probs[order[q]];
    if (!P.src || !P.W || !P.U || P.C_in < 16) return BMNAS_E_ARG;
    if (P.C_in % 16 || P.ldw % 4 || P.ldw < P.C_in) return BMNAS_E_SHAPE;
    ConvArgs& a = G.a[q];
    if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
    a.act.p[0] = P.src; a.dst.p[0] = P.U; a.W = P.W; a.bias = P.bias; a.ldw = P.ldw;
    a.part = nullptr; a.stat = stat_shards ? P.stat : nullptr; a.stat_shards = stat_shards;
    if (stat_shards && !P.stat) return BMNAS_E_ARG;
    a.Ci = P.C_in; a.I = P.C_in; a.Cj = M; a.J = M; a.b = b; a.L = L; a.n_part = a.n_groups;
    const int pgx = (a.n_groups + 1) / 2, pgy = (M + kPipeJ - 1) / kPipeJ;
    G.start[q] = blocks;
    if (P.C_in % 32 == 0 && use_pipe) {
      G.kind[q] = 0; G.gx[q] = pgx;
      blocks += pgx * pgy;
      lds = std::max(lds, conv_pipe_lds<32, 2>(L));
    } else {
      G.kind[q] = 1; G.gx[q] = a.n_groups;
      blocks += a.n_groups * (M / 16);
    }
  }
  G.start[n] = blocks;
  if (b == 0) return 0;
  BMNAS_COUNT(F_FWD_GROUP);
  // long contractions on few tiles: split every tile's contraction over the quads of a 512- / 1024-thread workgroup
  {
    bool all_pipe = use_pipe;
    int max_chunks = 0, quads = 4;
    for (int q = 0; q < n; ++q) {
      all_pipe = all_pipe && G.kind[q] == 0;
      const int nc = G.a[q].I / 32;
      max_chunks = std::max(max_chunks, nc);
      while (quads > 1 && nc % quads) quads >>= 1;
    }
    static const int lds_max = [] {
      int dev = 0, v = 65536;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
      return v;
    }();
    // (measured, NTU b64 / Ego b48, us — 96-column tiles: 1 quad 53.4 / 53.6, 2 quads 39.7 / 40.2, 4 quads 39.0 / 38.8:
    // the longest tiles are then bound by their CU's matrix pipe, 21.8 us for a 32 x 96 x 2048 tile; with the narrow
    // tiles below: 2 quads 22.3 / 31.2, 4 quads 20.1 / 27.5.  BMNAS_FWD_GROUP_QUADS = 1 / 2 for the table)
    static const int quad_mode = [] { const char* e = getenv("BMNAS_FWD_GROUP_QUADS"); return e ? atoi(e) : 4; }();
    quads = std::min(quads, quad_mode);
    size_t pipe_lds = conv_pipe_lds<32, 2>(L);
    while (quads > 1 && pipe_lds * quads > (size_t)lds_max) quads >>= 1;
    if (all_pipe && max_chunks >= 32 && blocks <= 320 && quads > 1) {
      // Tile width per layer.  A width that is no multiple of 96 (NTU / Ego: 128) wastes two thirds of its ragged
      // tile's MFMAs, and a 32 x 96 x 2048 tile keeps ONE CU's matrix pipe busy for 21.8 us while most CUs idle:
      // 64-column tiles where 96 does not divide the width, 32-column tiles for the long contractions (4x the
      // tiles, a quarter of the work each).  BMNAS_FWD_GROUP_NARROW=0: 96-column tiles only (the table in DESIGN.md).
      static const int narrow = [] { const char* e = getenv("BMNAS_FWD_GROUP_NARROW"); return e ? atoi(e) : 1; }();
      ConvFwdGroup Gq = G;                                       // (G itself stays as the plain kernel wants it)
      int qblocks = 0;
      pipe_lds = 0;
      for (int q = 0; q < n; ++q) {
        const int nc = G.a[q].I / 32, pgx = (G.a[q].n_groups + 1) / 2;
        int jt = 3;
        if (narrow && M % 96 != 0) jt = (nc >= 32 && M % 32 == 0) ? 1 : (M % 64 == 0 ? 2 : 3);
        Gq.kind[q] = jt == 3 ? 0 : (jt == 2 ? 2 : 3);
        Gq.start[q] = qblocks;
        Gq.gx[q] = pgx;
        qblocks += pgx * ((M + 32 * jt - 1) / (32 * jt));
        pipe_lds = std::max(pipe_lds, jt == 3 ? conv_pipe_lds<32, 2, 3>(L)
                                              : (jt == 2 ? conv_pipe_lds<32, 2, 2>(L) : conv_pipe_lds<32, 2, 1>(L)));
      }
      Gq.start[n] = qblocks;
      for (int q = 0; q < n; ++q)
        if (Gq.kind[q] == 0 && quads == 4) quads = 2;            // (96-column tiles: two quads, see conv_fwd_group_q_k)
      // (a stale error of an earlier call must not read as a refused launch: the plain kernel would then run ON TOP
      // of this one and add the BatchNorm sums twice)
      hipError_t err = hipGetLastError();
      if (err != hipSuccess) return (int)err;
      if (quads == 4) {
        static const hipError_t attr4 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_group_q_k<4>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)attr4;
        hipLaunchKernelGGL(conv_fwd_group_q_k<4>, dim3((unsigned)qblocks), dim3(1024), pipe_lds * 4, (hipStream_t)stream, Gq);
      } else {
        static const hipError_t attr2 = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_group_q_k<2>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        (void)attr2;
        hipLaunchKernelGGL(conv_fwd_group_q_k<2>, dim3((unsigned)qblocks), dim3(512), pipe_lds * 2, (hipStream_t)stream, Gq);
      }
      err = hipGetLastError();
      if (err == hipSuccess) {
        BMNAS_COUNT(F_FWD_QUADS_GROUP);                          // (counted on top of fwd_group: the quad form of it)
        return 0;
      }
      // (a runtime that refuses the launch — LDS limit, block size — leaves nothing behind: the plain kernel runs)
    }
  }
  hipLaunchKernelGGL(conv_fwd_group_k, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, G);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_bwd_group(const bmnas_conv_bwd_prob_t* probs, int n, int bn_training, int b, int L,
                                       int M, void* stream) {
  if (!probs || n < 1 || b < 0 || M < 1) return BMNAS_E_ARG;
  if (n > kGroupMax) return BMNAS_E_LIMIT;
  if (M % 16 || M > 384) return BMNAS_E_SHAPE;
  ConvBwdGroup G{};
  G.n = n;
  size_t lds = conv_w_lds<4>();
  int order[kGroupMax];
  for (int p = 0; p < n; ++p) order[p] = p;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && probs[order[j]].C_in > probs[order[j - 1]].C_in; --j) std::swap(order[j], order[j - 1]);
  int blocks = 0;
  for (int q = 0; q < n; ++q) {                      // weight-gradient tiles
    const bmnas_conv_bwd_prob_t& P = probs[order[q]];
    if (!P.dV || !P.W || !P.src || !P.dW || P.C_in < 16) return BMNAS_E_ARG;
    if (P.bn_U != nullptr && (!P.bn_chan || (bn_training && !P.bn_grad))) return BMNAS_E_ARG;
    if (P.C_in % 16 || P.ldw % 4 || P.ldw < P.C_in) return BMNAS_E_SHAPE;
    ConvWArgs& w = G.w[q];
    dim3 wgrid;
    const float* srcs[1] = {P.src};
    if (int e = fill_w_args(w, P.dV, srcs, 1, P.C_in, P.dW, P.ldw_grad, P.dbias, 0, b, L, M, 4, &wgrid)) return e;
    {
      // fill_w_args sizes the batch splits for a conv that has the chip to itself; here n layers share the
      // grid: as many splits as bring the GROUP to ~3 weight-gradient workgroups per CU (every split is another
      // round of fp32 atomics on dW)
      long tiles_all = 0;
      for (int r = 0; r < n; ++r) tiles_all += (long)((M + 31) / 32) * ((probs[r].C_in + 31) / 32);
      int splits = (int)std::max<long>(1, (768 + tiles_all - 1) / tiles_all);
      const int max_splits = (w.n_groups + 7) / 8;
      splits = std::min(splits, std::max(1, max_splits));
      if (g_conv_deterministic) splits = 1;
      w.groups_per_split = (w.n_groups + splits - 1) / splits;
      splits = (w.n_groups + w.groups_per_split - 1) / w.groups_per_split;
      w.use_atomic = splits > 1;
      wgrid.z = (unsigned)splits;
    }
    w.bn_U = P.bn_U; w.bn_chan = P.bn_chan; w.bn_grad = P.bn_grad; w.bn_train = bn_training;
    G.wstart[q] = blocks; G.wx[q] = (int)wgrid.x; G.wy[q] = (int)wgrid.y;
    blocks += (int)(wgrid.x * wgrid.y * wgrid.z);
  }
  G.wstart[n] = blocks;
  G.n_w = blocks;
  // data-gradient tile family, for the group as a whole (see bmnas_conv1x1_fwd_group)
  long pipe_tiles = 0;
  {
    int Lb0, spw0, ng0;
    if (int e = check_shape(b > 0 ? b : 1, L, &Lb0, &spw0, &ng0)) return e;
    for (int q = 0; q < n; ++q)
      if (probs[q].dsrc != nullptr) pipe_tiles += (long)((ng0 + 1) / 2) * ((probs[q].C_in + kPipeBJ - 1) / kPipeBJ);
  }
  for (int q = 0; q < n; ++q) {                      // data-gradient tiles
    const bmnas_conv_bwd_prob_t& P = probs[order[q]];
    ConvArgs& a = G.a[q];
    if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
    G.dstart[q] = blocks;
    if (P.dsrc == nullptr) continue;
    a.act.p[0] = P.dV; a.dst.p[0] = P.dsrc; a.W = P.W; a.ldw = P.ldw;
    a.Ci = M; a.I = M; a.Cj = P.C_in; a.J = P.C_in; a.b = b; a.L = L; a.acc_mask = P.accumulate ? 1u : 0u;
    a.bn_U = P.bn_U; a.bn_chan = P.bn_chan; a.bn_grad = P.bn_grad; a.bn_train = bn_training;
    const int pgx = (a.n_groups + 1) / 2, pgy = (a.J + kPipeBJ - 1) / kPipeBJ;
    const size_t coef = a.bn_U ? (size_t)a.I * sizeof(float4) : 0;
    if (M % 48 == 0 && pipe_tiles >= 96) {
      G.kind[q] = 0; G.gx[q] = pgx;
      blocks += pgx * pgy;
      lds = std::max(lds, conv_pipe_bwd_lds<48, 2>(L) + coef);
    } else if (M % 32 == 0 && pipe_tiles >= 96) {
      G.kind[q] = 1; G.gx[q] = pgx;
      blocks += pgx * pgy;
      lds = std::max(lds, conv_pipe_bwd_lds<32, 2>(L) + coef);
    } else {
      G.kind[q] = (M / 16 + 3) / 4 <= 3 ? 3 : 6;
      G.gx[q] = a.n_groups;
      blocks += a.n_groups * (a.J / 16);
      lds = std::max(lds, conv_ksplit_lds<1, 1>() + coef);
    }
  }
  G.dstart[n] = blocks;
  G.probe = conv_probe();
  if (b == 0) return 0;
  BMNAS_COUNT(F_BWD_GROUP);
  hipLaunchKernelGGL(conv_bwd_group_k, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, G);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv_family_calls(long* out, int n, int reset) {
  if (!out && n > 0) return BMNAS_E_ARG;
  for (int i = 0; i < n && i < F_COUNT; ++i) out[i] = g_family_calls[i];
  if (reset)
    for (int i = 0; i < F_COUNT; ++i) g_family_calls[i] = 0;
  return F_COUNT;
}

extern "C" const char* bmnas_conv_family_name(int i) {
  static const char* names[F_COUNT] = {"ksplit", "pipe_fwd", "pipe_bwd", "lds", "fwd_sdpa_pipe",
                                       "fwd_sdpa_ksplit", "bwd_all_pipe", "bwd_all_ksplit",
                                       "conv_w", "bwd_pair", "fwd_group", "bwd_group", "fwd_quads_group"};
  return (i >= 0 && i < F_COUNT) ? names[i] : nullptr;
}

extern "C" int bmnas_fold_weight(const float* W, float* Weff, int M, int C, void* stream) {
  if (!W || !Weff || M < 1 || C < 1) return BMNAS_E_ARG;
  if (C % 4) return BMNAS_E_SHAPE;
  const int64_t total = (int64_t)M * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(fold_weight_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, W, Weff, M, C);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_set_deterministic(int on) {
  g_conv_deterministic = on ? 1 : 0;
  return 0;
}

// tools/stamp_probe.py (timing builds): slot 6 = the data-gradient tiles of the pipelined backward GEMM
BMNAS_DEFINE_STAMP_SETTER(bmnas_debug_stamps_conv)
