// K2 as the producer of out_conv's last operand (small grids, node_multiplier != 1).
//
// NodeCell.forward (node_search.py:55, 59-61): the last inner step's NodeMixedOp output s is the LAST of the
// node_multiplier states that out_conv (Conv1d 1x1 over their channel concat) reads, and nothing else reads it in
// the forward pass.  At 6-8 samples per GPU (NTU / EgoGesture, BASELINE configs 4 and 5) every launch of the step
// is at its floor, so the mix (bmnas_node_mix_fwd) and the GEMM (bmnas_conv1x1_fwd) as two launches cost two
// floors for 32 KB of data.  Here ONE launch: a workgroup owns a 16-column n-group x 16 output channels like the
// split-K GEMM (four waves split the contraction, partial tiles meet in LDS); all 256 threads first form the
// n-group's slice of s in the streaming layout of the mix kernel (float4 along l, one Philox draw per four
// elements, BatchNorm finalised in LDS from the producer GEMM's batch sums: bn_fin.hpp), park it in LDS as the
// MFMA A-operand of the contraction blocks that belong to it, and the workgroups of output tile 0 also write it
// out (the backward pass and later states read it).  The (C / 16)-fold recomputation of the slice across the
// output tiles is 2 K elements each — nothing against a launch.
#include "bn_fin.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr int kMixSrc = 3;                                      // node_multiplier - 1 <= 3 plain sources
constexpr int kLd = 20;                                         // LDS row of the parked slice: 16 columns + 4 (bank spread)
constexpr int kNv = 4;                                          // float4 of the slice per thread: C <= 256

struct MixConvArgs {
  const float* src[kMixSrc];
  const float *x, *y, *p1, *U, *gamma;
  float* chan;
  BnFin fin;
  float* mix_out;
  DropCfg dglu, dfc;
  const float *W, *bias;
  float *V, *stat;
  int stat_shards, ldw, nsrc, b, C, L, Lb, spw, n_groups;
};

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float4 aff4(float4 u, float sc, float sh) {
  return make_float4(fmaf(u.x, sc, sh), fmaf(u.y, sc, sh), fmaf(u.z, sc, sh), fmaf(u.w, sc, sh));
}

template <int KPW>
__global__ __launch_bounds__(256) void mix_conv_fwd_k(MixConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float mc_smem[];
  const int C = a.C, L = a.L, M3 = 3 * C;
  float* sc = mc_smem;
  float* sh = mc_smem + M3;
  float* Af = mc_smem + 2 * M3;                                 // [C][kLd]: s[c][n] of this n-group
  float4* part = reinterpret_cast<float4*>(Af + C * kLd);       // [4][64]
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lo = lane & 15, h = lane >> 4;
  const int bx = blockIdx.x % a.n_groups, by = blockIdx.x / a.n_groups;
  const DropRt rglu = drop_begin(a.dglu), rfc = drop_begin(a.dfc);
  const float g0 = a.gamma[0], g1 = a.gamma[1], g2 = a.gamma[2], g3 = a.gamma[3];

  // ---- every global load of the launch first: GEMM operands in MFMA layout ...
  int sm = bx * a.spw + (lo >> a.Lb);
  sm = sm < a.b ? sm : a.b - 1;                                 // clamped columns are never stored
  const int64_t abase = (int64_t)sm * C * L + (lo & (L - 1));
  const int jj = by * 16 + lo;
  const int cb = C / 16, nblk = (a.nsrc + 1) * cb;
  float av[KPW][4];
  float4 bv[KPW];
#pragma unroll
  for (int kb = 0; kb < KPW; ++kb) {
    const int blk = wave * KPW + kb;
    const int bc = blk < nblk ? blk : nblk - 1;
    const int q = bc / cb;
    const int qp = q < a.nsrc ? q : a.nsrc - 1;                 // blocks of s: a dummy (valid) read, replaced below
    const int ci = (bc - q * cb) * 16 + 4 * h;
    const float* src = pick_ptr(a.src, qp);
#pragma unroll
    for (int r = 0; r < 4; ++r) av[kb][r] = src[abase + (int64_t)(ci + r) * L];
    bv[kb] = ld4(a.W + (int64_t)jj * a.ldw + bc * 16 + 4 * h);
  }
  const float bj = a.bias[jj];
  // ... and the mix operands of this n-group's slice, float4 along l
  const int per = C * L / 4, l4n = L / 4, n4 = a.spw * per;
  float4 xv[kNv], yv[kNv], pv[kNv], ua[kNv], ug[kNv], uf[kNv];
#pragma unroll
  for (int k = 0; k < kNv; ++k) {
    const int idx = t + 256 * k;
    const int ic = idx < n4 ? idx : n4 - 1;                     // clamped, not predicated (a load under `if` is a wait)
    const int sl = ic / per, rem = ic - sl * per;
    int sg = bx * a.spw + sl;
    sg = sg < a.b ? sg : a.b - 1;
    const int64_t e = (int64_t)sg * C * L + (int64_t)rem * 4;
    const int64_t ub = (int64_t)sg * M3 * L + (int64_t)rem * 4;
    xv[k] = ld4(a.x + e);
    yv[k] = ld4(a.y + e);
    pv[k] = ld4(a.p1 + e);
    ua[k] = ld4(a.U + ub);
    ug[k] = ld4(a.U + ub + (int64_t)C * L);
    uf[k] = ld4(a.U + ub + (int64_t)2 * C * L);
  }
  bn_fin_fill<256>(a.fin, a.chan, M3, a.b * L, sc, sh, blockIdx.x == 0);

  // ---- s = g0 (x + y) + g1 p1 + g2 drop(glu) + g3 drop(relu(fc))  (the arithmetic of node_mix_fwd_k, bnmix.hip)
#pragma unroll
  for (int k = 0; k < kNv; ++k) {
    const int idx = t + 256 * k;
    if (idx < n4) {
      const int sl = idx / per, rem = idx - sl * per;
      const int c = rem / l4n, l4 = rem - c * l4n;
      const int sg = bx * a.spw + sl;
      const bool valid = sg < a.b;
      const int64_t e = (int64_t)sg * C * L + (int64_t)rem * 4;
      const float4 va = aff4(ua[k], sc[c], sh[c]);
      const float4 vg = aff4(ug[k], sc[C + c], sh[C + c]);
      const float4 vf = aff4(uf[k], sc[2 * C + c], sh[2 * C + c]);
      const float4 m2 = drop_mult4(rglu, (uint64_t)e), m3 = drop_mult4(rfc, (uint64_t)e);
      float4 o;
      o.x = g0 * (xv[k].x + yv[k].x) + g1 * pv[k].x + g2 * (va.x * sigm(vg.x) * m2.x) + g3 * (fmaxf(vf.x, 0.f) * m3.x);
      o.y = g0 * (xv[k].y + yv[k].y) + g1 * pv[k].y + g2 * (va.y * sigm(vg.y) * m2.y) + g3 * (fmaxf(vf.y, 0.f) * m3.y);
      o.z = g0 * (xv[k].z + yv[k].z) + g1 * pv[k].z + g2 * (va.z * sigm(vg.z) * m2.z) + g3 * (fmaxf(vf.z, 0.f) * m3.z);
      o.w = g0 * (xv[k].w + yv[k].w) + g1 * pv[k].w + g2 * (va.w * sigm(vg.w) * m2.w) + g3 * (fmaxf(vf.w, 0.f) * m3.w);
      st4(Af + c * kLd + sl * L + 4 * l4, valid ? o : make_float4(0.f, 0.f, 0.f, 0.f));
      if (by == 0 && valid) st4_wtg<4>(a.mix_out + e, o);
    }
  }
  __syncthreads();

  // ---- the contraction: blocks of s come from LDS, the rest is in registers already
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KPW; ++kb) {
    const int blk = wave * KPW + kb;
    const bool vb = blk < nblk;                                 // wave-uniform
    const int bc = vb ? blk : nblk - 1;
    const int q = bc / cb;
    if (q == a.nsrc) {
      const int ci = (bc - q * cb) * 16 + 4 * h;
#pragma unroll
      for (int r = 0; r < 4; ++r) av[kb][r] = Af[(ci + r) * kLd + lo];
    }
    if (!vb) av[kb][0] = av[kb][1] = av[kb][2] = av[kb][3] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][0], bv[kb].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][1], bv[kb].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][2], bv[kb].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kb][3], bv[kb].w, acc, 0, 0, 0);
  }
  part[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  __syncthreads();
  if (wave != 0) return;

  // acc[r] = OUT[n = 16 bx + 4h + r][j = jj]
  const float4 p0 = part[lane], p1 = part[64 + lane], p2 = part[128 + lane], p3 = part[192 + lane];
  const float4 d = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y),
                               (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
  const int so = bx * a.spw + ((4 * h) >> a.Lb), l0 = (4 * h) & (L - 1);
  const bool vo = so < a.b;
  if (vo) st4_wtg<4>(a.V + ((int64_t)so * C + jj) * L + l0, make_float4(d.x + bj, d.y + bj, d.z + bj, d.w + bj));
  if (a.stat != nullptr) {                                      // batch sums of d = v - bias (bn_tile_stats, conv1x1.hip)
    float sum = vo ? f4_hsum(d) : 0.f, sq = vo ? f4_dot(d, d) : 0.f;
    sum = xor16_sum(sum);
    sum = xor32_sum(sum);
    sq = xor16_sum(sq);
    sq = xor32_sum(sq);
    if (h == 0) {
      float* pp = a.stat + ((int64_t)(bx % a.stat_shards) * C + jj) * 2;
      atomicAdd(pp, sum);
      atomicAdd(pp + 1, sq);
    }
  }
}

inline DropCfg mc_cfg(const bmnas_dropout_t& d) {
  DropCfg c;
  c.thr = d.thr; c.scale = d.scale; c.seed = d.seed; c.offset = d.offset; c.step = d.step;
  return c;
}

}  // namespace

extern "C" int bmnas_node_mix_conv_fwd_ok(int b, int C, int L, int n_src) {
  if (b < 1 || n_src < 1 || n_src > kMixSrc) return 0;
  if (!(L == 4 || L == 8 || L == 16) || C % 64 || C > 64 * kNv) return 0;
  const int n_groups = (b * L + 15) / 16;
  // small grids only: every output tile recomputes its n-group's slice of s
  return n_groups * (C / 16) <= 256 && ((n_src + 1) * C / 16 + 3) / 4 <= 12;
}

extern "C" int bmnas_node_mix_conv_fwd(const float* x, const float* y, const float* p1, const float* U, float* chan,
                                       bmnas_bn_fin_t fin, const float* gamma, float* mix_out,
                                       bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, const float* const* srcs,
                                       int n_src, const float* W, int ldw, const float* bias, float* V, float* stat,
                                       int stat_shards, int b, int C, int L, void* stream) {
  if (!x || !y || !p1 || !U || !chan || !gamma || !mix_out || !srcs || !W || !bias || !V) return BMNAS_E_ARG;
  if (b < 0 || stat_shards < 0 || (stat_shards > 0) != (stat != nullptr)) return BMNAS_E_ARG;
  if (b == 0) return 0;
  if (!bmnas_node_mix_conv_fwd_ok(b, C, L, n_src)) return BMNAS_E_LIMIT;
  if (ldw < (n_src + 1) * C || ldw % 4) return BMNAS_E_SHAPE;
  MixConvArgs a{};
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.src[q] = srcs[q];
  }
  a.x = x; a.y = y; a.p1 = p1; a.U = U; a.gamma = gamma; a.chan = chan; a.mix_out = mix_out;
  // (the in-kernel finalisation only: a chan finalised by bmnas_bn_finalize goes through `on = 0`)
  {
    BnFin f{};
    f.on = fin.on ? 1 : 0;
    if (f.on) {
      if (!fin.bn_w || !fin.bn_b || fin.shards < 0 || fin.n_nbt < 0) return BMNAS_E_ARG;
      if (fin.shards > 4) return BMNAS_E_LIMIT;
      if (fin.training && (!fin.stat || fin.shards < 1)) return BMNAS_E_ARG;
      if (!fin.training && (!fin.running_mean || !fin.running_var)) return BMNAS_E_ARG;
      if ((fin.running_mean == nullptr) != (fin.running_var == nullptr)) return BMNAS_E_ARG;
      f.stat = fin.stat; f.conv_bias = fin.conv_bias; f.bn_w = fin.bn_w; f.bn_b = fin.bn_b;
      f.running_mean = fin.running_mean; f.running_var = fin.running_var;
      f.nbt = reinterpret_cast<long long*>(fin.num_batches_tracked);
      f.shards = fin.shards; f.n_nbt = fin.n_nbt; f.training = fin.training ? 1 : 0;
    }
    a.fin = f;
  }
  a.dglu = mc_cfg(drop_glu); a.dfc = mc_cfg(drop_fc);
  a.W = W; a.bias = bias; a.V = V; a.stat = stat; a.stat_shards = stat_shards; a.ldw = ldw; a.nsrc = n_src;
  a.b = b; a.C = C; a.L = L;
  a.Lb = L == 4 ? 2 : (L == 8 ? 3 : 4);
  a.spw = 16 / L;
  a.n_groups = (b + a.spw - 1) / a.spw;
  const int kpw = ((n_src + 1) * C / 16 + 3) / 4;
  const dim3 grid((unsigned)(a.n_groups * (C / 16)));
  const size_t lds = (size_t)(6 * C + C * kLd) * sizeof(float) + 4 * 64 * sizeof(float4);
  hipStream_t st = (hipStream_t)stream;
#define MC_CASE(K)                                                                  \
  if (kpw <= K) {                                                                   \
    hipLaunchKernelGGL(mix_conv_fwd_k<K>, grid, dim3(256), lds, st, a);             \
    BMNAS_CHECK_LAUNCH();                                                           \
    return 0;                                                                       \
  }
  MC_CASE(2) MC_CASE(3) MC_CASE(4) MC_CASE(6) MC_CASE(8) MC_CASE(12)
#undef MC_CASE
  return BMNAS_E_LIMIT;
}
