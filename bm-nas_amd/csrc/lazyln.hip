// The step node's LayerNorm without a one-workgroup-per-sample kernel (node_multiplier == 1).
//
// NodeCell ends in `out += x; out = LayerNorm_[C,L](out)` (node_search.py:67-68).  A kernel that normalises has to see
// the whole sample, which made the launches around it grids of b workgroups (128 of 256 CUs busy at MM-IMDB b = 128,
// each pulling 100 KB through one CU).  Here the node output is produced and differentiated by STREAMING grids and
// the normalisation moves into its consumers:
//
//   forward   bmnas_node_mix_pre_fwd      gamma-mix (K2, node_operations.py:118-120) + residual -> pre, un-normalised,
//                                         plus per-part moment records (lazy_ln.hpp)
//             bmnas_mixsum_pair_fwd_lazy  the NEXT cell step's K1 pair sum (model_search.py:58, node_search.py:54) reads
//                                         `pre`, combines the records, normalises in registers, and is the launch that
//                                         writes the node output, its LayerNorm statistics and its per-sample sums
//             bmnas_head_fwd_lazy         (head.hip) the LAST step node's output is normalised in the classifier GEMM's
//                                         operand fetch and never written
//   backward  every producer of a piece of the node-output gradient gy (bmnas_head_bwd_lazy, bmnas_mixsum_pair_bwd_lazy)
//             also leaves per-workgroup partials of the two sums the LayerNorm backward needs —
//             S(gy w) and S(gy w xhat) are linear in gy — as plain stores;
//             bmnas_node_mix_lnp_bwd      sums the partials of its samples, applies the LayerNorm backward
//             elementwise and continues into the mix backward (the arithmetic of node_mix_bwd_k), streaming.
//
// No atomics on any of the new per-sample quantities: the records and partials are plain stores combined in a fixed
// order (run-to-run deterministic).
#include "common.hpp"
#include "../../include/bmnas_hip.h"
#include "bn_fin.hpp"
#include "lazy_ln.hpp"
#include "mix_common.hpp"

namespace {

// ------------------------------------------------------------------------------------------------ forward: producer
// grid = (P parts, b samples), 256 threads, one float4 per thread.
// pre = g0 (x + y) + g1 p1 + g2 drop(va sigmoid(vg)) + g3 drop(relu(vf)) + resid   (same expression order as
// node_mix_ln_fwd_k, so `pre` is bit-identical to that kernel's)
__global__ __launch_bounds__(256) void node_mix_pre_fwd_k(
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ p1,
    const float* __restrict__ U, float* __restrict__ chan, BnFin fin, const float* __restrict__ gamma,
    const float* __restrict__ resid, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
    float* __restrict__ pre, float* __restrict__ rec, float* __restrict__ prm, int b, int C, int L,
    DropCfg dglu, DropCfg dfc) {
  __shared__ float red[4];
  __shared__ float red5[2 * 4 * 5];
  extern __shared__ float fin_lds[];
  const int cl4 = C * L / 4, l4n = L / 4, M = 3 * C;
  const int part = blockIdx.x, smp = blockIdx.y, P = gridDim.x;
  STAMP(0, smp * P + part, 0);
  float* sc = fin_lds;
  float* sh = fin_lds + M;
  const int r0 = part * kLazyPart + threadIdx.x;
  const bool act = r0 < cl4;
  const int r = act ? r0 : cl4 - 1;                           // clamped address, no predicated loads
  const int c = r / l4n;
  const int64_t e = ((int64_t)smp * cl4 + r) * 4;
  const int64_t ub = ((int64_t)smp * M) * L + (int64_t)r * 4;
  // (every operand load first, then the BatchNorm finalisation — its own memory round trip + a barrier.  Issuing the
  // finalisation's loads FIRST, so that it completes while the operands are still travelling, was measured: 7.22 ->
  // 7.13 us, inside the noise; removed)
  const float4 lw = ld4(ln_w + (int64_t)r * 4), lb = ld4(ln_b + (int64_t)r * 4);
  const float4 ua = ld4(U + ub), ug = ld4(U + ub + (int64_t)C * L), uf = ld4(U + ub + (int64_t)2 * C * L);
  const float4 xv = ld4(x + e), yv = ld4(y + e), pv = ld4(p1 + e), rv = ld4(resid + e);
  // only now what hangs off a SECOND scalar round trip (the dropout step counters and gamma sit behind pointers of
  // the argument block): ahead of the operand loads each of them was a kernarg fetch -> wait -> dependent fetch ->
  // wait chain in front of the first vector load (in-kernel stamps: 1.5 us from entry to "loads issued")
  __builtin_amdgcn_sched_barrier(0);
  const float g0 = gamma[0], g1 = gamma[1], g2 = gamma[2], g3 = gamma[3];
  const DropRt rglu = drop_begin(dglu), rfc = drop_begin(dfc);
  STAMP(0, smp * P + part, 1);
  bn_fin_fill<256>(fin, chan, M, b * L, sc, sh, part == 0 && smp == 0);
  STAMP(0, smp * P + part, 2);
  const float4 va = affine4(ua, sc[c], sh[c]);
  const float4 vg = affine4(ug, sc[C + c], sh[C + c]);
  const float4 vf = affine4(uf, sc[2 * C + c], sh[2 * C + c]);
  const float4 m2 = drop_mult4(rglu, (uint64_t)e), m3 = drop_mult4(rfc, (uint64_t)e);
  float4 o;
  o.x = g0 * (xv.x + yv.x) + g1 * pv.x + g2 * (va.x * sigmoidf(vg.x) * m2.x) + g3 * (fmaxf(vf.x, 0.f) * m3.x);
  o.y = g0 * (xv.y + yv.y) + g1 * pv.y + g2 * (va.y * sigmoidf(vg.y) * m2.y) + g3 * (fmaxf(vf.y, 0.f) * m3.y);
  o.z = g0 * (xv.z + yv.z) + g1 * pv.z + g2 * (va.z * sigmoidf(vg.z) * m2.z) + g3 * (fmaxf(vf.z, 0.f) * m3.z);
  o.w = g0 * (xv.w + yv.w) + g1 * pv.w + g2 * (va.w * sigmoidf(vg.w) * m2.w) + g3 * (fmaxf(vf.w, 0.f) * m3.w);
  const float4 v = f4_add(o, rv);
  if (act) st4_w0<3>(pre + e, v);
  STAMP(0, smp * P + part, 3);
  const int n4 = cl4 - part * kLazyPart < kLazyPart ? cl4 - part * kLazyPart : kLazyPart;
  const float nk = (float)(4 * n4);
  // (each LDS array below is written once per launch: the reductions need no barrier in FRONT of their writes —
  // block_sum / block_sum_lead carry one for callers that reuse the array)
  float mk = wave_sum(act ? f4_hsum(v) : 0.f);
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  if (ln == 0) red[wv] = mk;
  __syncthreads();
  mk = (((red[0] + red[1]) + red[2]) + red[3]) / nk;
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (act) {
    const float4 cd = make_float4(v.x - mk, v.y - mk, v.z - mk, v.w - mk);
    const float4 cw = f4_mul(cd, lw);
    acc[0] = f4_dot(cd, cd);
    acc[1] = f4_hsum(cw);
    acc[2] = f4_dot(cw, cw);
    acc[3] = f4_dot(cw, lb);
    acc[4] = f4_dot(cw, lw);
  }
  // the five centred sums — and, for sample 0's workgroups, the affine parameters' own five — behind ONE barrier
  float pa[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (smp == 0 && act) {                                      // (smp: uniform)
    pa[0] = f4_dot(lw, lw);
    pa[1] = f4_hsum(lw);
    pa[2] = f4_dot(lw, lb);
    pa[3] = f4_hsum(lb);
    pa[4] = f4_dot(lb, lb);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = wave_sum(acc[i]);
  if (smp == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) pa[i] = wave_sum(pa[i]);
  }
  if (ln == 0) {
#pragma unroll
    for (int i = 0; i < 5; ++i) red5[i * 4 + wv] = acc[i];
    if (smp == 0) {
#pragma unroll
      for (int i = 0; i < 5; ++i) red5[20 + i * 4 + wv] = pa[i];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = ((red5[i * 4] + red5[i * 4 + 1]) + red5[i * 4 + 2]) + red5[i * 4 + 3];
    float* rr = rec + ((int64_t)smp * P + part) * kLazyRec;
    st4(rr, make_float4(mk, t[0], t[1], t[2]));
    st4(rr + 4, make_float4(t[3], t[4], 0.f, 0.f));
  }
  STAMP(0, smp * P + part, 4);
  if (smp == 0 && threadIdx.x == 64) {
    float t[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
      t[i] = ((red5[20 + i * 4] + red5[20 + i * 4 + 1]) + red5[20 + i * 4 + 2]) + red5[20 + i * 4 + 3];
    st4(prm + part * kLazyRec, make_float4(t[0], t[1], t[2], t[3]));
    st4(prm + part * kLazyRec + 4, make_float4(t[4], nk, 0.f, 0.f));
  }
}

// ------------------------------------------------------------------------------- forward: first consumer (K1 pair)
// h = sum_{j < NIN} w_j x_j + w_NIN n,  z = (w2_0 + w2_1) h,  n = LayerNorm(pre) formed here and written out.
// Same accumulation order as mixsum_pair_fwd_k with n as its last input.
template <int NIN>
__global__ __launch_bounds__(256) void mixsum_pair_fwd_lazy_k(
    PtrsIn xs, const float* __restrict__ w, int w_stride, const float* __restrict__ w2, int w2_stride,
    const float* __restrict__ pre, const float* __restrict__ rec, const float* __restrict__ prm,
    const float* __restrict__ ln_w, const float* __restrict__ ln_b, float* __restrict__ nout,
    float* __restrict__ stats, float* __restrict__ osum, float* __restrict__ out, float* __restrict__ out2,
    int cl4) {
  const int part = blockIdx.x, smp = blockIdx.y, P = gridDim.x;
  STAMP(1, smp * P + part, 0);
  const int r0 = part * kLazyPart + threadIdx.x;
  const bool act = r0 < cl4;
  const int r = act ? r0 : cl4 - 1;
  const int64_t i = (int64_t)smp * cl4 + r;
  float wj[NIN + 1];
#pragma unroll
  for (int j = 0; j <= NIN; ++j) wj[j] = w[j * w_stride];
  const float s2 = w2[0] + w2[w2_stride];
  float4 v[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) v[j] = reinterpret_cast<const float4*>(xs.p[j])[i];
  const float4 pv = reinterpret_cast<const float4*>(pre)[i];
  const float4 lw = ld4(ln_w + (int64_t)r * 4), lb = ld4(ln_b + (int64_t)r * 4);
  const LazyStats st = lazy_combine(rec, prm, P, smp);
  const float4 n = make_float4((pv.x - st.mean) * st.rstd * lw.x + lb.x, (pv.y - st.mean) * st.rstd * lw.y + lb.y,
                               (pv.z - st.mean) * st.rstd * lw.z + lb.z, (pv.w - st.mean) * st.rstd * lw.w + lb.w);
  float4 acc = f4_scale(v[0], wj[0]);
#pragma unroll
  for (int j = 1; j < NIN; ++j) {
    acc.x = fmaf(wj[j], v[j].x, acc.x);
    acc.y = fmaf(wj[j], v[j].y, acc.y);
    acc.z = fmaf(wj[j], v[j].z, acc.z);
    acc.w = fmaf(wj[j], v[j].w, acc.w);
  }
  acc.x = fmaf(wj[NIN], n.x, acc.x);
  acc.y = fmaf(wj[NIN], n.y, acc.y);
  acc.z = fmaf(wj[NIN], n.z, acc.z);
  acc.w = fmaf(wj[NIN], n.w, acc.w);
  if (act) {
    st4_w0<4>(nout + 4 * i, n);
    st4_w0<4>(out + 4 * i, acc);
    st4_w0<4>(out2 + 4 * i, f4_scale(acc, s2));
  }
  STAMP(1, smp * P + part, 1);
  if (part == 0 && threadIdx.x == 0) {
    stats[2 * smp] = st.mean;
    stats[2 * smp + 1] = st.rstd;
    if (osum != nullptr) {
      osum[2 * smp] = st.osum;
      osum[2 * smp + 1] = st.osq;
    }
  }
}

// ----------------------------------------------------------------------------------- backward: K1 pair, lazy inputs
// bmnas_mixsum_pair_bwd on the (part, sample) grid.  Its last NLZ inputs are step-node outputs whose LayerNorm
// backward runs in a streaming launch later: for each of them the workgroup stores its partial of
//   S(gy w),  S(gy w xhat)     with gy = w_j G the piece of the node-output gradient formed here.
// g_full (nullable): G itself is stored, and the gradients of the first `skip` inputs are NOT written — the first
// step's backward forms them once from every step's G (bmnas_mixsum_pair_bwd_x).
constexpr int kMaxLazyIn = 2;
struct LazyIn {
  const float* pre[kMaxLazyIn];
  const float* ln_w[kMaxLazyIn];
  const float* stats[kMaxLazyIn];
  float* lnpart[kMaxLazyIn];
  int stride[kMaxLazyIn];        // pairs per sample in lnpart[t] (>= P: several consumers share one buffer)
};

// DOTS = false (dw == dw2 == NULL at the C ABI): nobody differentiates the edge weights — the weight step of the search
// loop — so the NIN inputs and h, read only for the dot products, are not loaded at all (7 + 1 of 17 streams at MM-IMDB).
template <int NIN, int NLZ, bool DOTS = true>
__global__ __launch_bounds__(256) void mixsum_pair_bwd_lazy_k(
    PtrsIn xs, PtrsOut dxs, const float* __restrict__ w, int w_stride, const float* __restrict__ w2, int w2_stride,
    const float* __restrict__ h, const float* __restrict__ gh, const float* __restrict__ gz,
    const float* __restrict__ gz2, float* dw, float* dw2, int dw_shards, int64_t dw_shard_stride, uint32_t acc_mask,
    LazyIn Z, float* __restrict__ g_full, int cl4) {
  constexpr int NR = NIN + 1 + 2 * NLZ;
  __shared__ float red[4 * NR];
  const int part = blockIdx.x, smp = blockIdx.y, P = gridDim.x;
  STAMP(2, smp * P + part, 0);
  const int r0 = part * kLazyPart + threadIdx.x;
  const bool act = r0 < cl4;
  const int r = act ? r0 : cl4 - 1;
  const int64_t i = (int64_t)smp * cl4 + r;
  float wj[NIN];
#pragma unroll
  for (int j = 0; j < NIN; ++j) wj[j] = w[j * w_stride];
  const float s2 = w2[0] + w2[w2_stride];
  // EVERY load unconditional (an absent optional operand reads gz again and is masked out): a load under `if` is a
  // branch whose join waits for vmcnt(0) — gz, gz2 and gh were three dependent round trips in front of the operands
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
  float4 g4 = f4_add(f4_scale(z4, s2), gh != nullptr ? ghv : zero4);
  float4 pz[NLZ], lz[NLZ], oldn[NLZ];
  float mean[NLZ], rstd[NLZ];
#pragma unroll
  for (int t = 0; t < NLZ; ++t) {
    pz[t] = reinterpret_cast<const float4*>(Z.pre[t])[i];
    lz[t] = ld4(Z.ln_w[t] + (int64_t)r * 4);
    mean[t] = Z.stats[t][2 * smp];
    rstd[t] = Z.stats[t][2 * smp + 1];
    // the node-output gradients are read-modify-written (the head's launch wrote them first): their old values are
    // fetched HERE, with the operands — next to their store the load was a dependent round trip of its own at the
    // end of the kernel (load, s_waitcnt vmcnt(0), add, store)
    const float* od = dxs.p[NIN - NLZ + t];
    oldn[t] = reinterpret_cast<const float4*>(od != nullptr && (acc_mask & (1u << (NIN - NLZ + t))) ? od : gz)[i];
  }
  if (!act) {
    z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    g4 = z4;
  }
  STAMP(2, smp * P + part, 1);
  float part_[NR];
#pragma unroll
  for (int j = 0; j < NIN; ++j) part_[j] = f4_dot(g4, v[j]);
  part_[NIN] = f4_dot(z4, h4);
#pragma unroll
  for (int t = 0; t < NLZ; ++t) {
    const float4 gw = f4_mul(f4_scale(g4, wj[NIN - NLZ + t]), lz[t]);
    const float4 xh = make_float4((pz[t].x - mean[t]) * rstd[t], (pz[t].y - mean[t]) * rstd[t],
                                  (pz[t].z - mean[t]) * rstd[t], (pz[t].w - mean[t]) * rstd[t]);
    part_[NIN + 1 + 2 * t] = f4_hsum(gw);
    part_[NIN + 2 + 2 * t] = f4_dot(gw, xh);
  }
  if (act) {
    if (g_full != nullptr) st4_w0<9>(g_full + 4 * i, g4);
#pragma unroll
    for (int j = 0; j < NIN; ++j) {                           // in order: destinations may alias each other
      float* d = dxs.p[j];
      if (d == nullptr) continue;
      float4 rr = f4_scale(g4, wj[j]);
      if (acc_mask & (1u << j)) {
        // (the lazy inputs are pairwise distinct step-node outputs, host-checked: their old values came with the loads)
        rr = f4_add(rr, j >= NIN - NLZ ? oldn[j - (NIN - NLZ) < 0 ? 0 : j - (NIN - NLZ)] : reinterpret_cast<float4*>(d)[i]);
      }
      st4_w0<9>(d + 4 * i, rr);
    }
  }
  STAMP(2, smp * P + part, 2);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    if (!DOTS && q <= NIN) continue;                          // (the dot products: nobody reads them)
    const float sv = wave_sum(part_[q]);
    if (lane == 0) red[wave * NR + q] = sv;
  }
  __syncthreads();
  if ((int)threadIdx.x < NR + 1 && (DOTS || (int)threadIdx.x > NIN + 1)) {
    const int t = threadIdx.x;
    const int q = t <= NIN ? (t < NIN ? t : NIN) : t - 1;     // t = NIN and NIN + 1: the two dw2 adds
    const float val = red[q] + red[NR + q] + red[2 * NR + q] + red[3 * NR + q];
    const int64_t shd = (int64_t)((smp * P + part) % dw_shards) * dw_shard_stride;
    if (t < NIN) atomicAdd(dw + shd + t * w_stride, val);
    else if (t <= NIN + 1) atomicAdd(dw2 + shd + (t - NIN) * w2_stride, val);
    else {
      const int u = t - (NIN + 2);                            // 0 .. 2 NLZ - 1
      Z.lnpart[u >> 1][((int64_t)smp * Z.stride[u >> 1] + part) * 2 + (u & 1)] = val;
    }
  }
  STAMP(2, smp * P + part, 3);
}

// ------------------------------------------------------------------------------- backward: LayerNorm + mix, streaming
// grid = (ceil(cl4 / 64) slot groups, sample chunks); thread = (col: float4 slot of the (C, L) tile, sl: one of 4
// sample lanes = one wave each).  A wave first sums the partials of ITS sample (n0 pairs from lnp0, n1 from lnp1 —
// the head's and / or the later cell steps' K1 backward), then
//   g = rstd (gy w - m1 - xhat m2)          LayerNorm input gradient (= gradient of the mix output and of the residual)
// and the mix backward of node_mix_bwd_k.  BatchNorm reductions: registers over the chunk, LDS over the 4 sample
// lanes, one atomic pair per channel per workgroup.
__global__ __launch_bounds__(256) void node_mix_lnp_bwd_k(
    const float* __restrict__ gy, const float* __restrict__ pre, const float* __restrict__ ln_w,
    const float* __restrict__ stats, const float* __restrict__ lnp0, int n0, const float* __restrict__ lnp1, int n1,
    float* __restrict__ gbuf, float* dresid, int acc_resid, const float* __restrict__ x,
    const float* __restrict__ y, const float* __restrict__ p1, const float* __restrict__ U,
    const float* __restrict__ chan, const float* __restrict__ gamma, float* dgamma, int dg_shards, int64_t dg_stride,
    float* dx, float* dy, uint32_t acc_mask, float* __restrict__ dV, float* bn_grad, float* __restrict__ bn_part,
    int b, int C, int L, int chunk, DropCfg dglu, DropCfg dfc) {
  __shared__ float red16[16];
  __shared__ float csum[3][6][64];
  const int cl4 = C * L / 4, l4n = L / 4, M = 3 * C;
  STAMP(3, blockIdx.y * gridDim.x + blockIdx.x, 0);
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64 + col;
  const bool active = r0 < cl4;
  const int r = active ? r0 : cl4 - 1;
  const int c = r / l4n;
  const float g0 = gamma[0], g2 = gamma[2], g3 = gamma[3];
  const DropRt rglu = drop_begin(dglu), rfc = drop_begin(dfc);
  float sc[3], sh[3], mu[3], rs[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    mu[k] = chan[k * C + c];
    rs[k] = chan[M + k * C + c];
    sc[k] = chan[2 * M + k * C + c];
    sh[k] = chan[3 * M + k * C + c];
  }
  const float4 lw = ld4(ln_w + (int64_t)r * 4);
  const float inv_d = 1.f / (float)(cl4 * 4);
  float dgam[4] = {0.f, 0.f, 0.f, 0.f};
  float sw[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
  const int s_beg = blockIdx.y * chunk;
  int s_end = s_beg + chunk;
  if (s_end > b) s_end = b;
  const int np = n0 + n1;
  for (int s0 = s_beg; s0 < s_end; s0 += 4) {
    const int s_raw = s0 + sl;
    const bool on = active && s_raw < s_end;
    const int s = s_raw < s_end ? s_raw : s_end - 1;          // clamped (wave-uniform)
    // the sample's partial sums first, then every operand: one round trip
    float q1 = 0.f, q2 = 0.f;
    for (int t = col; t < np; t += 64) {
      const float2 pp = t < n0 ? reinterpret_cast<const float2*>(lnp0)[(int64_t)s * n0 + t]
                               : reinterpret_cast<const float2*>(lnp1)[(int64_t)s * n1 + (t - n0)];
      q1 += pp.x;
      q2 += pp.y;
    }
    const float mean = stats[2 * s], rstd = stats[2 * s + 1];
    const int64_t e = ((int64_t)s * cl4 + r) * 4;
    const int64_t ub = ((int64_t)s * M) * L + (int64_t)r * 4;
    const float4 gyv = ld4(gy + e), prv = ld4(pre + e);
    const float4 ua = ld4(U + ub), ug = ld4(U + ub + (int64_t)C * L), uf = ld4(U + ub + (int64_t)2 * C * L);
    const float4 xv = ld4(x + e), yv = ld4(y + e), pv = ld4(p1 + e);
    float4 oldr = make_float4(0.f, 0.f, 0.f, 0.f), oldx = oldr, oldy = oldr;
    if (acc_resid) oldr = ld4(dresid + e);
    if (dx != nullptr && (acc_mask & 1u)) oldx = ld4(dx + e);
    if (dy != nullptr && (acc_mask & 2u)) oldy = ld4(dy + e);
    const float m1 = wave_sum(q1) * inv_d, m2 = wave_sum(q2) * inv_d;
    STAMP(3, blockIdx.y * gridDim.x + blockIdx.x, 1);
    if (on) {
      const float4 xh = make_float4((prv.x - mean) * rstd, (prv.y - mean) * rstd, (prv.z - mean) * rstd,
                                    (prv.w - mean) * rstd);
      float4 gv;
      gv.x = rstd * (gyv.x * lw.x - m1 - xh.x * m2);
      gv.y = rstd * (gyv.y * lw.y - m1 - xh.y * m2);
      gv.z = rstd * (gyv.z * lw.z - m1 - xh.z * m2);
      gv.w = rstd * (gyv.w * lw.w - m1 - xh.w * m2);
      if (gbuf != nullptr) st4_w0<6>(gbuf + e, gv);
      if (dresid != nullptr) st4_w0<6>(dresid + e, f4_add(gv, oldr));
      const float4 m2d = drop_mult4(rglu, (uint64_t)e), m3d = drop_mult4(rfc, (uint64_t)e);
      const float gq[4] = {gv.x, gv.y, gv.z, gv.w};
      const float uaq[4] = {ua.x, ua.y, ua.z, ua.w}, ugq[4] = {ug.x, ug.y, ug.z, ug.w},
                  ufq[4] = {uf.x, uf.y, uf.z, uf.w};
      const float xq[4] = {xv.x + yv.x, xv.y + yv.y, xv.z + yv.z, xv.w + yv.w};
      const float pq[4] = {pv.x, pv.y, pv.z, pv.w};
      const float m2q[4] = {m2d.x, m2d.y, m2d.z, m2d.w}, m3q[4] = {m3d.x, m3d.y, m3d.z, m3d.w};
      float da[4], dg[4], df[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float va = fmaf(uaq[t], sc[0], sh[0]), vg = fmaf(ugq[t], sc[1], sh[1]), vf = fmaf(ufq[t], sc[2], sh[2]);
        const float sg = sigmoidf(vg);
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
      st4_w0<6>(dV + ub, make_float4(da[0], da[1], da[2], da[3]));
      st4_w0<6>(dV + ub + (int64_t)C * L, make_float4(dg[0], dg[1], dg[2], dg[3]));
      st4_w0<6>(dV + ub + (int64_t)2 * C * L, make_float4(df[0], df[1], df[2], df[3]));
      const float4 d0 = f4_scale(gv, g0);
      if (dx != nullptr) st4_w0<6>(dx + e, f4_add((dy == nullptr) ? f4_scale(d0, 2.f) : d0, oldx));
      if (dy != nullptr) st4_w0<6>(dy + e, f4_add(d0, oldy));
    }
  }
  STAMP(3, blockIdx.y * gridDim.x + blockIdx.x, 2);
  float cs[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    cs[k] = row_sum(sw[k], l4n);
    cs[3 + k] = row_sum(sb[k], l4n);
  }
  if (sl > 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) csum[sl - 1][k][col] = cs[k];
  }
  // dgamma's wave sums meet in LDS behind the SAME barrier (a block_sum_lead after the BatchNorm sums was two more)
#pragma unroll
  for (int q = 0; q < 4; ++q) dgam[q] = wave_sum(dgam[q]);
  if (col == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) red16[q * 4 + sl] = dgam[q];
  }
  __syncthreads();
  if (sl == 0 && active && (r % l4n) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) cs[k] += csum[0][k][col] + csum[1][k][col] + csum[2][k][col];
    if (bn_part != nullptr) {
      // deterministic mode: this sample chunk's row of partials, plain stores (the launcher sums the rows in order)
      float* row = bn_part + (int64_t)blockIdx.y * 2 * M;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        row[k * C + c] = cs[k];
        row[M + k * C + c] = cs[3 + k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        atomicAdd(bn_grad + k * C + c, cs[k]);
        atomicAdd(bn_grad + M + k * C + c, cs[3 + k]);
      }
    }
  }
  STAMP(3, blockIdx.y * gridDim.x + blockIdx.x, 3);
  if (threadIdx.x < 4 && dgamma != nullptr) {
    const int q = threadIdx.x;
    float* p = dgamma + (int64_t)((blockIdx.y * gridDim.x + blockIdx.x) % dg_shards) * dg_stride;
    atomicAdd(p + q, ((red16[q * 4] + red16[q * 4 + 1]) + red16[q * 4 + 2]) + red16[q * 4 + 3]);
  }
  STAMP(3, blockIdx.y * gridDim.x + blockIdx.x, 4);
}

}  // namespace

// tools/stamp_probe.py (timing builds): slot 0 node_mix_pre_fwd_k, 1 mixsum_pair_fwd_lazy_k, 2 mixsum_pair_bwd_lazy_k,
// 3 node_mix_lnp_bwd_k
BMNAS_DEFINE_STAMP_SETTER(bmnas_debug_stamps)

extern "C" int bmnas_lazy_ln_ok(int C, int L) {
  return C >= 1 && L >= 4 && L % 4 == 0 && L <= 16 && lazy_parts(C * L / 4) <= kLazyMaxParts;
}

extern "C" int bmnas_lazy_ln_parts(int C, int L) {
  if (!bmnas_lazy_ln_ok(C, L)) return BMNAS_E_LIMIT;
  return lazy_parts(C * L / 4);
}

extern "C" int bmnas_node_mix_pre_fwd(const float* x, const float* y, const float* p1, const float* U, float* chan,
                                      bmnas_bn_fin_t fin, const float* gamma, const float* resid, const float* ln_w,
                                      const float* ln_b, float* pre, float* rec, float* prm, int b, int C, int L,
                                      bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, void* stream) {
  if (!x || !y || !p1 || !U || !chan || !gamma || !resid || !ln_w || !ln_b || !pre || !rec || !prm || b < 0 || C < 1)
    return BMNAS_E_ARG;
  if (L % 4 || L > 16 || L < 4) return BMNAS_E_SHAPE;
  if (!bmnas_lazy_ln_ok(C, L)) return BMNAS_E_LIMIT;
  BnFin f;
  if (int e = to_fin(fin, &f)) return e;
  if (f.on && f.training && b * L < 2) return BMNAS_E_ARG;
  if (b == 0) return 0;
  const int P = lazy_parts(C * L / 4);
  hipLaunchKernelGGL(node_mix_pre_fwd_k, dim3(P, b), dim3(256), (size_t)6 * C * sizeof(float), (hipStream_t)stream,
                     x, y, p1, U, chan, f, gamma, resid, ln_w, ln_b, pre, rec, prm, b, C, L, to_cfg(drop_glu),
                     to_cfg(drop_fc));
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_mixsum_pair_fwd_lazy(const float* const* xs, int n_in, const float* w, int w_stride,
                                          const float* w2, int w2_stride, const bmnas_lazy_ln_t* last,
                                          float* last_out, float* last_sums, float* out, float* out2, int b, int C,
                                          int L, void* stream) {
  if (!xs || !w || !w2 || !last || !last_out || !out || !out2 || n_in < 1 || b < 0 || w_stride < 1 || w2_stride < 1)
    return BMNAS_E_ARG;
  if (!last->pre || !last->rec || !last->prm || !last->ln_w || !last->ln_b || !last->stats) return BMNAS_E_ARG;
  if (n_in > BMNAS_MAX_PTRS - 1) return BMNAS_E_LIMIT;
  if (!bmnas_lazy_ln_ok(C, L)) return BMNAS_E_LIMIT;
  if (b == 0) return 0;
  PtrsIn p{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
  }
  const int cl4 = C * L / 4;
  dim3 grid(lazy_parts(cl4), b);
  hipStream_t st = (hipStream_t)stream;
#define CALL(N)                                                                                              \
  hipLaunchKernelGGL(mixsum_pair_fwd_lazy_k<N>, grid, dim3(256), 0, st, p, w, w_stride, w2, w2_stride,       \
                     last->pre, last->rec, last->prm, last->ln_w, last->ln_b, last_out, last->stats,         \
                     last_sums, out, out2, cl4)
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

extern "C" int bmnas_mixsum_pair_bwd_lazy(const float* const* xs, float* const* dxs, int n_in, const float* w,
                                          int w_stride, const float* w2, int w2_stride, const float* h,
                                          const float* gh, const float* gz, const float* gz2, float* dw, float* dw2,
                                          int dw_shards, int64_t dw_shard_stride, uint32_t accumulate_mask,
                                          const bmnas_lazy_ln_t* lazy, float* const* lnpart,
                                          const int* lnpart_stride, int n_lazy, float* g_full, int b, int C, int L,
                                          void* stream) {
  if (!xs || !dxs || !w || !w2 || !h || !gz || !lazy || !lnpart || !lnpart_stride || n_in < 1 || b < 0 ||
      w_stride < 1 || w2_stride < 1 || dw_shards < 1)
    return BMNAS_E_ARG;
  if ((dw == nullptr) != (dw2 == nullptr)) return BMNAS_E_ARG;      // both (the dot products wanted) or neither
  const bool dots = dw != nullptr;
  if (n_lazy < 1 || n_lazy > kMaxLazyIn || n_lazy >= n_in) return BMNAS_E_LIMIT;
  if (n_in > BMNAS_MAX_PTRS - 1) return BMNAS_E_LIMIT;
  if (!bmnas_lazy_ln_ok(C, L)) return BMNAS_E_LIMIT;
  if (b == 0) return 0;
  PtrsIn p{};
  PtrsOut d{};
  for (int j = 0; j < n_in; ++j) {
    if (!xs[j]) return BMNAS_E_ARG;
    p.p[j] = xs[j];
    d.p[j] = dxs[j];
  }
  LazyIn Z{};
  for (int t = 0; t < n_lazy; ++t) {
    if (!lazy[t].pre || !lazy[t].ln_w || !lazy[t].stats || !lnpart[t]) return BMNAS_E_ARG;
    Z.pre[t] = lazy[t].pre; Z.ln_w[t] = lazy[t].ln_w; Z.stats[t] = lazy[t].stats; Z.lnpart[t] = lnpart[t];
    Z.stride[t] = lnpart_stride[t];
    if (Z.stride[t] < lazy_parts(C * L / 4)) return BMNAS_E_ARG;
    // the kernel fetches the old value of a lazy input's gradient with its operand loads: no other destination of
    // this launch may alias it
    const int jt = n_in - n_lazy + t;
    for (int j = 0; j < n_in; ++j)
      if (j != jt && dxs[jt] != nullptr && dxs[j] == dxs[jt]) return BMNAS_E_ARG;
  }
  const int cl4 = C * L / 4;
  dim3 grid(lazy_parts(cl4), b);
  hipStream_t st = (hipStream_t)stream;
#define CALL2(N, Z_)                                                                                                 \
  do {                                                                                                               \
    if (dots)                                                                                                        \
      hipLaunchKernelGGL((mixsum_pair_bwd_lazy_k<N, Z_, true>), grid, dim3(256), 0, st, p, d, w, w_stride, w2,       \
                         w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride, accumulate_mask, Z, g_full, \
                         cl4);                                                                                       \
    else                                                                                                             \
      hipLaunchKernelGGL((mixsum_pair_bwd_lazy_k<N, Z_, false>), grid, dim3(256), 0, st, p, d, w, w_stride, w2,      \
                         w2_stride, h, gh, gz, gz2, dw, dw2, dw_shards, dw_shard_stride, accumulate_mask, Z, g_full, \
                         cl4);                                                                                       \
  } while (0)
#define CALL(N)                            \
  do {                                     \
    if (n_lazy == 1) CALL2(N, 1);          \
    else CALL2(N, 2);                      \
  } while (0)
  switch (n_in) {
    case 2: CALL2(2, 1); break;
    case 3: CALL(3); break;   case 4: CALL(4); break;   case 5: CALL(5); break;   case 6: CALL(6); break;
    case 7: CALL(7); break;   case 8: CALL(8); break;   case 9: CALL(9); break;   case 10: CALL(10); break;
    case 11: CALL(11); break; case 12: CALL(12); break; case 13: CALL(13); break; case 14: CALL(14); break;
    case 15: CALL(15); break;
    default: return BMNAS_E_LIMIT;
  }
#undef CALL
#undef CALL2
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_node_mix_lnp_bwd(const float* g, const float* pre, const float* ln_w, const float* stats,
                                      const float* lnp0, int n0, const float* lnp1, int n1, float* g_in,
                                      float* dresid, int accumulate_resid, const float* x, const float* y,
                                      const float* p1, const float* U, const float* chan, const float* gamma,
                                      float* dgamma, int dgamma_shards, int64_t dgamma_shard_stride, float* dx,
                                      float* dy, uint32_t accumulate_mask, float* dV, float* bn_grad, int b, int C,
                                      int L, bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, float* bn_part,
                                      void* stream) {
  if (!g || !pre || !ln_w || !stats || !x || !y || !p1 || !U || !chan || !gamma || !dV || !bn_grad || b < 0 ||
      C < 1 || dgamma_shards < 1 || n0 < 0 || n1 < 0 || (n0 > 0 && !lnp0) || (n1 > 0 && !lnp1) || n0 + n1 < 1)
    return BMNAS_E_ARG;
  if (accumulate_resid && !dresid) return BMNAS_E_ARG;
  if (!(L == 4 || L == 8 || L == 16)) return BMNAS_E_SHAPE;
  if (b == 0) return 0;
  const int cl4 = C * L / 4;
  // one sample per thread while that still leaves the BatchNorm sums <= ~64 adds per address; longer walks above
  int chunk = 4;
  while ((b + chunk - 1) / chunk > 64) chunk += 4;
  dim3 grid((cl4 + 63) / 64, (b + chunk - 1) / chunk);
  hipLaunchKernelGGL(node_mix_lnp_bwd_k, grid, dim3(256), 0, (hipStream_t)stream, g, pre, ln_w, stats, lnp0, n0,
                     lnp1, n1, g_in, dresid, accumulate_resid, x, y, p1, U, chan, gamma, dgamma, dgamma_shards,
                     dgamma_shard_stride, dx, dy, accumulate_mask, dV, bn_grad, bn_part, b, C, L, chunk,
                     to_cfg(drop_glu), to_cfg(drop_fc));
  BMNAS_CHECK_LAUNCH();
  if (bn_part != nullptr)      // rows in order -> bn_grad (6 C floats per row; C % 16 == 0)
    return bmnas_sum_chunks(bn_part, bn_grad, (int)grid.y, (int64_t)6 * C, stream);
  return 0;
}

// rows of bn_part that bmnas_node_mix_lnp_bwd(bn_part != NULL) writes (each 6 C floats)
extern "C" int bmnas_node_mix_lnp_bwd_rows(int b) {
  if (b < 1) return BMNAS_E_ARG;
  int chunk = 4;
  while ((b + chunk - 1) / chunk > 64) chunk += 4;
  return (b + chunk - 1) / chunk;
}
