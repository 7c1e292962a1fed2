// Channel-owner kernels for the small per-GPU shards of the NTU / EgoGesture configurations (b L <= 64 columns).
//
// The column-tiled GEMM kernels of conv1x1.hip make the train-mode BatchNorm of LinearGLU / ConcatFC
// (reference models/search/darts/node_operations.py:34, :53) a GRID-wide dependency: every tile holds a few columns of
// every channel, the batch statistics of a channel need all columns, so an inner step of a NodeCell
// (node_search.py:52-57) is conv + attention | mix — two launches per direction — and at 8 / 6 samples per GPU a
// launch costs more than the work in it (25 / 33 launches of ~6 us ARE the step: profiles/r05_bench_ntu_b8.json).
// Here the ownership is flipped: a workgroup owns 16 output channels `cb` of the NodeMixedOp — rows {cb, C + cb,
// 2C + cb} of the stacked conv (GLU value, GLU gate, ConcatFC) — over ALL b L <= 64 columns.  Its BatchNorm statistics
// are sums inside the workgroup (two passes over values it holds: no E[d^2] - E[d]^2), GLU / ReLU / dropout / Sum are
// local, and the inner step becomes ONE launch:
//
//   forward  (bmnas_co_inner_fwd):   C / 16 channel-owner workgroups  +  b L / 16 attention workgroups (sdpa_body.hpp)
//       owner:      U = Weff z (+ bias) on v_mfma_f32_16x16x4_f32 straight from global loads in operand layout (wave w
//                   = column tile w, three accumulators = the three row tiles), accumulators transposed through LDS
//                   into the streaming layout (thread = channel x four adjacent l), statistics by DPP row sums,
//                   chan / running statistics / U written, s_own = g0 2z + g2 drop(glu) + g3 drop(relu);
//       attention:  p1 = LayerNorm(drop(softmax(z^T z / sqrt C) z^T)), unchanged;
//       both ADD their share of s (and of the next inner step's mixed sum z_next = sum_j w_j prev_j + w_n s) with fp32
//       atomics onto zero-filled buffers: exactly two addends per address, so the sum does not depend on their order.
//
//   backward (bmnas_co_inner_bwd):   (C / 16) x QS channel-owner workgroups  +  b L / 16 attention workgroups
//       every owner workgroup of a channel block first completes the incoming gradient for its channels
//       (g = g_in + w_n (gz + gz2): the backward of the NEXT inner step's mixed sum, node_search.py:54), runs the mix
//       backward on them, reduces the BatchNorm backward sums locally and forms dU (the BatchNorm INPUT gradient) in
//       LDS; then the QS workgroups of the block share the two contractions: dW rows of the block (= dU z^T, plain
//       stores: one writer per element, no atomics) and the block's share of dz = Weff^T dU (32 KB of partials added
//       with atomics onto a zero-filled buffer).  Workgroup q = 0 of a block also writes what must be written once
//       (g, d prev_j, the dot products for d beta / d gamma, the BatchNorm affine gradients, the Sum primitive's share
//       of dz).  The attention workgroups complete g the same way while loading it.
//
// Replaces bmnas_conv1x1_fwd_sdpa + bmnas_node_mix_fwd_next and bmnas_node_mix_bwd_next + bmnas_conv1x1_bwd_all_sdpa
// for the inner steps t < node_steps - 1 of a search-mode NodeCell (x is y); results equal theirs to fp32 round-off.
#include "sdpa_body.hpp"
#include <cstdlib>

namespace {

constexpr int kCoCols = 64;            // columns (b * L) a channel-owner workgroup covers
constexpr int kCoLdw = kCoCols + 4;    // LDS row stride of the 48 x 64 tile (floats)
constexpr int kCoPrev = 5;             // states the next inner step's sum can read besides s
constexpr float kEpsBn = 1e-5f, kMomBn = 0.1f;

__device__ __forceinline__ float co_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }

struct CoAttnF {
  const float* ln_w; const float* ln_b;
  float* p1; float* xhat; float* stats;
  SdpaGeom G;
  DropCfg drop;
};

struct CoFwdArgs {
  const float* z;          // (b, C, L): the inner mixed sum (NodeMixedOp(z, z))
  const float* Weff;       // (3C, C): W[:, :C] + W[:, C:]
  const float* bias;       // (3C), nullable
  const float* bn_w; const float* bn_b;
  float* rm; float* rv;    // running statistics (updated in training mode when present; read in eval mode)
  long long* nbt; int n_nbt; int training;
  const float* gamma;
  float* U;                // (b, 3C, L)
  float* chan;             // mean | rstd | scale | shift, 4 x 3C
  float* s;                // (b, C, L), zero-filled
  const float* prev[kCoPrev];
  const float* w;          // w[j * ws], j = 0 .. n_prev
  float* zn;               // (b, C, L), zero-filled; nullable
  int n_prev, ws;
  int b, C, L, Lb, NC;
  int probe;               // timing experiments only (BMNAS_CO_PROBE; results are wrong with any bit set)
  DropCfg dglu, dfc;
};

// BMNAS_CO_PROBE bits: 1 = attention workgroups return at once, 2 = owners store instead of atomics, 4 = no MFMA loop,
// 8 = no operand loads for it, 16 = no dropout draws, 32 = owners return before the streaming phase
inline int co_probe() {
  static const int v = [] { const char* e = getenv("BMNAS_CO_PROBE"); return e ? atoi(e) : 0; }();
  return v;
}

template <int KB>
__global__ __launch_bounds__(256) void co_inner_fwd_k(CoFwdArgs a, CoAttnF at, int n_attn) {
  extern __shared__ __attribute__((aligned(16))) char co_smem[];
  if ((int)blockIdx.x < n_attn) {
    if (a.probe & 1) return;
    SdpaMixAdd ma{a.s, a.zn, a.gamma, a.zn != nullptr ? a.w + (int64_t)a.n_prev * a.ws : nullptr};
    sdpa_fwd_body<(KB + 3) / 4>(blockIdx.x, a.z, a.z, at.ln_w, at.ln_b, at.p1, at.xhat, at.stats, at.G, at.drop,
                                co_smem, ma);
    return;
  }
  const int cb = blockIdx.x - n_attn;
  float* us = reinterpret_cast<float*>(co_smem);          // [48][kCoLdw]: d = Weff z of the block's rows
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lo = lane & 15, h = lane >> 4;
  const int C = a.C, L = a.L, NC = a.NC, M = 3 * C;
  // Every value the kernel needs from memory is requested by VECTOR loads in ONE batch, the two dropout step counters
  // first: scalar loads of the uniform values (gamma, the next sum's weights, the counters, num_batches_tracked)
  // compiled into ~9 dependent s_load -> s_waitcnt round trips in FRONT of the operand loads (2.7 us of an 8.8 us
  // launch by the probe table, profiles/r06_co_probe.txt).  `vz` is a zero the compiler cannot see through.
  int vz;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
  const uint64_t* dummy = reinterpret_cast<const uint64_t*>(a.Weff);
  const uint64_t st0 = (a.dglu.step != nullptr ? a.dglu.step : dummy)[vz];
  const uint64_t st1 = (a.dfc.step != nullptr ? a.dfc.step : dummy)[vz];
  const float4 gam = ld4(a.gamma + vz);

  // ---- this thread's element of the streaming phase: channel cb*16 + ec, columns col0 .. col0 + 3
  const int ec = tid >> 4, q4 = tid & 15, col0 = 4 * q4;
  const bool ev = col0 < NC;
  const int colc = ev ? col0 : 0;
  const int ch = cb * 16 + ec;
  const int64_t e = ((int64_t)(colc >> a.Lb) * C + ch) * L + (colc & (L - 1));
  const float4 zv = ld4(a.z + e);
  float4 pv[kCoPrev];
  float wj[kCoPrev + 1];
#pragma unroll
  for (int j = 0; j < kCoPrev; ++j) {
    const float* pp = (j < a.n_prev) ? a.prev[j] : a.z;
    pv[j] = ld4(pp + e);
  }
#pragma unroll
  for (int j = 0; j <= kCoPrev; ++j) wj[j] = a.w[(int64_t)(j <= a.n_prev ? j : 0) * a.ws + vz];
  // num_batches_tracked: read with everything else, stored at the end (a load -> add -> store loop at the end was two
  // more memory round trips on workgroup 0's path)
  long long nb[4];
  const bool bump = cb == 0 && tid == 0 && a.training && a.nbt != nullptr;
  {
    const long long* np = a.nbt != nullptr ? a.nbt : reinterpret_cast<const long long*>(a.Weff);
#pragma unroll
    for (int j = 0; j < 4; ++j) nb[j] = np[(j < a.n_nbt ? j : 0) + vz];
  }
  float bs[3], bw[3], bb[3], rmv[3], rvv[3];
  const bool has_run = a.rm != nullptr;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int row = k * C + ch;
    bs[k] = (a.bias != nullptr ? a.bias : a.bn_w)[row];
    bw[k] = a.bn_w[row];
    bb[k] = a.bn_b[row];
    rmv[k] = (has_run ? a.rm : a.bn_w)[row];
    rvv[k] = (has_run ? a.rv : a.bn_w)[row];
    if (a.bias == nullptr) bs[k] = 0.f;
  }
  const float g0 = gam.x, g2 = gam.z, g3 = gam.w;
  // ---- the block's rows of the stacked conv: wave = column tile, A = weight rows (float4 along k), B = z
  const int NT = (NC + 15) >> 4;
  float4 m2, m3;
  DropRt rglu{a.dglu.thr, a.dglu.scale, a.dglu.seed, a.dglu.offset}, rfc{a.dfc.thr, a.dfc.scale, a.dfc.seed, a.dfc.offset};
  if (a.probe & 16) rglu.thr = rfc.thr = 0u;
  if (wave >= NT) {        // (no operand loads in this wave: its dropout draws go here)
    rglu.off += a.dglu.step != nullptr ? st0 : 0ull;
    rfc.off += a.dfc.step != nullptr ? st1 : 0ull;
    m2 = drop_mult4(rglu, (uint64_t)e);
    m3 = drop_mult4(rfc, (uint64_t)e);
  }
  if (wave < NT) {                                       // wave-uniform
    const int col = wave * 16 + lo;
    const bool cv = col < NC;
    const int colr = cv ? col : NC - 1;
    const float* zb = a.z + ((int64_t)(colr >> a.Lb) * C + 4 * h) * L + (colr & (L - 1));
    float4 wa[3][KB];
    float zq[KB][4];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
      for (int rt = 0; rt < 3; ++rt)
        wa[rt][kb] = ld4(a.Weff + (int64_t)(rt * C + cb * 16 + lo) * C + kb * 16 + 4 * h);
#pragma unroll
      for (int r = 0; r < 4; ++r) zq[kb][r] = zb[(int64_t)(kb * 16 + r) * L];
    }
    if (a.probe & 8) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
        for (int rt = 0; rt < 3; ++rt) wa[rt][kb] = make_float4(1.f, 2.f, 3.f, (float)lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) zq[kb][r] = (float)(lane + r);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                  // (the loads above stay above the dropout arithmetic below)
    // the two dropout draws run while the operand loads are in flight (they wait for the two counters only)
    rglu.off += a.dglu.step != nullptr ? st0 : 0ull;
    rfc.off += a.dfc.step != nullptr ? st1 : 0ull;
    m2 = drop_mult4(rglu, (uint64_t)e);
    m3 = drop_mult4(rfc, (uint64_t)e);
    f32x4 acc[3];
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(a.probe & 4))
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float zr[4] = {cv ? zq[kb][0] : 0.f, cv ? zq[kb][1] : 0.f, cv ? zq[kb][2] : 0.f, cv ? zq[kb][3] : 0.f};
#pragma unroll
      for (int rt = 0; rt < 3; ++rt) {
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rt][kb].x, zr[0], acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rt][kb].y, zr[1], acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rt][kb].z, zr[2], acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rt][kb].w, zr[3], acc[rt], 0, 0, 0);
      }
    }
    // lane holds D[row = 4h + rr][col = lo]
#pragma unroll
    for (int rt = 0; rt < 3; ++rt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) us[(rt * 16 + 4 * h + rr) * kCoLdw + wave * 16 + lo] = acc[rt][rr];
    }
  }
  __syncthreads();
  if (a.probe & 32) return;

  // ---- BatchNorm statistics (a 16-lane DPP row = one channel over all columns), affine, mix
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) d[k] = ev ? ld4(us + (k * 16 + ec) * kCoLdw + col0) : zero4;
  const float invN = 1.f / (float)NC;
  float sc[3], sh[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float mean_u, rstd, var = 0.f;
    if (a.training) {
      const float md = row16_sum(f4_hsum(d[k])) * invN;
      const float4 c4 = ev ? make_float4(d[k].x - md, d[k].y - md, d[k].z - md, d[k].w - md) : zero4;
      var = row16_sum(f4_dot(c4, c4)) * invN;
      mean_u = md + bs[k];
      rstd = 1.f / sqrtf(var + kEpsBn);
    } else {
      mean_u = rmv[k];
      rstd = 1.f / sqrtf(rvv[k] + kEpsBn);
    }
    sc[k] = bw[k] * rstd;
    sh[k] = bb[k] - mean_u * sc[k];
    if (q4 == 0 && !(a.probe & 64)) {
      const int row = k * C + ch;
      a.chan[row] = mean_u;
      a.chan[M + row] = rstd;
      a.chan[2 * M + row] = sc[k];
      a.chan[3 * M + row] = sh[k];
      if (a.training && has_run) {
        a.rm[row] = (1.f - kMomBn) * rmv[k] + kMomBn * mean_u;
        a.rv[row] = (1.f - kMomBn) * rvv[k] + kMomBn * (var * (float)NC / (float)(NC - 1));
      }
    }
  }
  if (bump) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < a.n_nbt) a.nbt[j] = nb[j] + 1;
    }
  }
  if (!ev) return;
  const int64_t ub = ((int64_t)(col0 >> a.Lb) * M + ch) * L + (col0 & (L - 1));
  const float4 ua = make_float4(d[0].x + bs[0], d[0].y + bs[0], d[0].z + bs[0], d[0].w + bs[0]);
  const float4 ug = make_float4(d[1].x + bs[1], d[1].y + bs[1], d[1].z + bs[1], d[1].w + bs[1]);
  const float4 uf = make_float4(d[2].x + bs[2], d[2].y + bs[2], d[2].z + bs[2], d[2].w + bs[2]);
  if (!(a.probe & 64)) {
    st4(a.U + ub, ua);
    st4(a.U + ub + (int64_t)C * L, ug);
    st4(a.U + ub + (int64_t)2 * C * L, uf);
  }
  const float uaq[4] = {ua.x, ua.y, ua.z, ua.w}, ugq[4] = {ug.x, ug.y, ug.z, ug.w}, ufq[4] = {uf.x, uf.y, uf.z, uf.w};
  const float m2q[4] = {m2.x, m2.y, m2.z, m2.w}, m3q[4] = {m3.x, m3.y, m3.z, m3.w};
  const float zq4[4] = {zv.x, zv.y, zv.z, zv.w};
  float so[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float va = fmaf(uaq[t], sc[0], sh[0]), vg = fmaf(ugq[t], sc[1], sh[1]), vf = fmaf(ufq[t], sc[2], sh[2]);
    so[t] = g0 * (zq4[t] + zq4[t]) + g2 * (va * co_sigmoid(vg) * m2q[t]) + g3 * (fmaxf(vf, 0.f) * m3q[t]);
    if (a.probe & 128) { if (so[t] == 123.456f) a.s[e + t] = so[t]; }
    else if (a.probe & 2) a.s[e + t] = so[t];
    else atomicAdd(a.s + e + t, so[t]);
  }
  if (a.zn != nullptr) {
    float zz[4];
    const float wn = (a.n_prev == 0) ? wj[0] : (a.n_prev == 1) ? wj[1] : (a.n_prev == 2) ? wj[2]
                     : (a.n_prev == 3) ? wj[3] : (a.n_prev == 4) ? wj[4] : wj[5];
#pragma unroll
    for (int t = 0; t < 4; ++t) zz[t] = wn * so[t];
#pragma unroll
    for (int j = 0; j < kCoPrev; ++j) {
      if (j < a.n_prev) {
        zz[0] = fmaf(wj[j], pv[j].x, zz[0]);
        zz[1] = fmaf(wj[j], pv[j].y, zz[1]);
        zz[2] = fmaf(wj[j], pv[j].z, zz[2]);
        zz[3] = fmaf(wj[j], pv[j].w, zz[3]);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (a.probe & 128) { if (zz[t] == 123.456f) a.zn[e + t] = zz[t]; }
      else if (a.probe & 2) a.zn[e + t] = zz[t];
      else atomicAdd(a.zn + e + t, zz[t]);
    }
  }
}

inline int co_shape_ok(int b, int C, int L) {
  if (!(L == 4 || L == 8 || L == 16) || C % 16 != 0) return 0;
  const int kb = C / 16;
  if (!(kb == 4 || kb == 8 || kb == 12)) return 0;
  const int nc = b * L;
  return nc >= 2 && nc <= kCoCols;
}

}  // namespace

extern "C" int bmnas_co_inner_fwd_ok(int b, int C, int L) { return co_shape_ok(b, C, L); }

extern "C" int bmnas_co_inner_fwd(const float* z, const float* Weff, bmnas_bn_fin_t bn, const float* gamma,
                                  const float* ln_w, const float* ln_b, float* p1, float* xhat, float* stats1,
                                  float* U, float* chan, float* s, int b, int C, int L, bmnas_dropout_t drop_attn,
                                  bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, const float* const* prev,
                                  int n_prev, const float* w, int w_stride, float* z_next, void* stream) {
  if (!z || !Weff || !gamma || !ln_w || !ln_b || !p1 || !xhat || !stats1 || !U || !chan || !s || b < 1)
    return BMNAS_E_ARG;
  if (!bn.bn_w || !bn.bn_b || bn.n_nbt < 0) return BMNAS_E_ARG;
  if (!bn.training && (!bn.running_mean || !bn.running_var)) return BMNAS_E_ARG;
  if ((bn.running_mean == nullptr) != (bn.running_var == nullptr)) return BMNAS_E_ARG;
  if (n_prev < 0 || n_prev > kCoPrev) return BMNAS_E_LIMIT;
  if (z_next && (!prev || !w || w_stride < 1)) return BMNAS_E_ARG;
  if (!co_shape_ok(b, C, L)) return BMNAS_E_LIMIT;
  CoFwdArgs a{};
  a.z = z; a.Weff = Weff; a.bias = bn.conv_bias; a.bn_w = bn.bn_w; a.bn_b = bn.bn_b;
  a.rm = bn.running_mean; a.rv = bn.running_var;
  a.nbt = reinterpret_cast<long long*>(bn.num_batches_tracked); a.n_nbt = bn.n_nbt; a.training = bn.training ? 1 : 0;
  a.gamma = gamma; a.U = U; a.chan = chan; a.s = s;
  a.zn = z_next; a.w = z_next ? w : gamma; a.ws = z_next ? w_stride : 0; a.n_prev = z_next ? n_prev : 0;
  if (bn.n_nbt > 4) return BMNAS_E_LIMIT;
  for (int j = 0; j < kCoPrev; ++j) {
    a.prev[j] = (z_next && j < n_prev) ? prev[j] : z;
    if (!a.prev[j]) return BMNAS_E_ARG;
  }
  a.b = b; a.C = C; a.L = L; a.Lb = ilog2_exact(L); a.NC = b * L; a.probe = co_probe();
  a.dglu = to_cfg(drop_glu); a.dfc = to_cfg(drop_fc);
  CoAttnF at{};
  if (int e = geom(b, C, L, &at.G)) return e;
  at.ln_w = ln_w; at.ln_b = ln_b; at.p1 = p1; at.xhat = xhat; at.stats = stats1; at.drop = to_cfg(drop_attn);
  const int n_attn = (b + at.G.spw - 1) / at.G.spw;
  const size_t lds = std::max((size_t)kSdpaFwdLds, (size_t)48 * kCoLdw * sizeof(float));
  dim3 grid((unsigned)(n_attn + C / 16));
  hipStream_t st = (hipStream_t)stream;
  switch (C / 16) {
    case 4: hipLaunchKernelGGL(co_inner_fwd_k<4>, grid, dim3(256), lds, st, a, at, n_attn); break;
    case 8: hipLaunchKernelGGL(co_inner_fwd_k<8>, grid, dim3(256), lds, st, a, at, n_attn); break;
    default: hipLaunchKernelGGL(co_inner_fwd_k<12>, grid, dim3(256), lds, st, a, at, n_attn); break;
  }
  BMNAS_CHECK_LAUNCH();
  return 0;
}
