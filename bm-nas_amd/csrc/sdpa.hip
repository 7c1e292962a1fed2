// K3 — ScaledDotAttn: softmax(x^T y / sqrt(C)) applied to y^T, dropout, LayerNorm([C, L]).
// One wavefront owns one 16-row "tile group" = 16/L samples packed on the 16-wide MFMA
// dimension (block-diagonal mask keeps samples apart).  Both contractions run on
// v_mfma_f32_16x16x4_f32 straight from global loads in operand layout:
//   S^T[j][i]  = sum_c y[c][j] x[c][i]        (contraction over channels, dword loads)
//   O[i][c]    = sum_j P[i][j] y[c][j]        (contraction over l: y read as float4 along l,
//                                              the accumulator comes out as float4 along l)
// Row softmax: the lane holds 4 of the 16 scores of row i, the rest sit in lanes
// xor 16 / xor 32 -> two wavefront shuffles per reduction.  The per-sample LayerNorm
// reductions are shuffles over the lanes that share the sample.  LDS is only used as
// per-lane indexed storage for the (C x 16) tile and for the three small transposes the
// backward needs.  Memory-bound: 3*T bytes forward (x, y -> out).
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr float kEps = 1e-5f;

struct SdpaGeom {
  int b, C, L, Lb, spw;
};

// sum over the lanes that belong to the same sample as this lane
__device__ __forceinline__ float sample_sum(float v, int L) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  if (L >= 8) v += __shfl_xor(v, 16, 64);
  if (L >= 16) v += __shfl_xor(v, 32, 64);
  return v;
}

// Softmax probabilities P[i = lo][j = 4h + r] for this tile group (recomputed in backward).
__device__ __forceinline__ void attn_probs(const float* __restrict__ x, const float* __restrict__ y,
                                           const SdpaGeom& G, int g, int lo, int h, float p[4]) {
  const int s_lo = g * G.spw + (lo >> G.Lb);
  const bool v_lo = s_lo < G.b;
  const int64_t base = ((int64_t)s_lo * G.C) * G.L + (lo & (G.L - 1));
  const float* xb = x + base;
  const float* yb = y + base;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < G.C; c0 += 8) {
    const int64_t o0 = (int64_t)(c0 + h) * G.L, o1 = (int64_t)(c0 + 4 + h) * G.L;
    const float a0 = v_lo ? yb[o0] : 0.f, b0 = v_lo ? xb[o0] : 0.f;
    const float a1 = v_lo ? yb[o1] : 0.f, b1 = v_lo ? xb[o1] : 0.f;
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
  }
  // lane holds S[i = lo][j = 4h + r]; keep only j in the same sample as i
  const float inv = 1.f / sqrtf((float)G.C);
  const bool same = ((4 * h) >> G.Lb) == (lo >> G.Lb);
  float sc[4], mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = same ? (acc0[r] + acc1[r]) * inv : -INFINITY;
    mx = fmaxf(mx, sc[r]);
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float den = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] = same ? __expf(sc[r] - mx) : 0.f;
    den += p[r];
  }
  den += __shfl_xor(den, 16, 64);
  den += __shfl_xor(den, 32, 64);
  const float rden = 1.f / den;
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] *= rden;
}

__global__ __launch_bounds__(64) void sdpa_ln_fwd_k(const float* __restrict__ x,
                                                    const float* __restrict__ y,
                                                    const float* __restrict__ ln_w,
                                                    const float* __restrict__ ln_b,
                                                    float* __restrict__ out,
                                                    float* __restrict__ stats, SdpaGeom G,
                                                    DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* obuf = reinterpret_cast<float4*>(smem);          // [C/16][64]
  const int lane = threadIdx.x, lo = lane & 15, h = lane >> 4;
  const int g = blockIdx.x;
  float p[4];
  attn_probs(x, y, G, g, lo, h, p);

  const int sh = g * G.spw + ((4 * h) >> G.Lb);
  const int l0 = (4 * h) & (G.L - 1);
  const bool v_h = sh < G.b;
  const int nch = G.C / 16;
  float sum = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
    const float4 yv = v_h ? ld4(y + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0], yv.x, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[1], yv.y, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[2], yv.z, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[3], yv.w, o, 0, 0, 0);
    const float4 m = drop_mult4(drop, (uint64_t)e);
    const float4 od = make_float4(o[0] * m.x, o[1] * m.y, o[2] * m.z, o[3] * m.w);
    obuf[ch * 64 + lane] = od;
    sum += f4_hsum(od);
  }
  const float inv_d = 1.f / (float)(G.C * G.L);
  const float mean = sample_sum(sum, G.L) * inv_d;
  float sq = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const float4 od = obuf[ch * 64 + lane];
    const float4 c = make_float4(od.x - mean, od.y - mean, od.z - mean, od.w - mean);
    sq += f4_dot(c, c);
  }
  const float var = sample_sum(sq, G.L) * inv_d;
  const float rstd = 1.f / sqrtf(var + kEps);
  if (!v_h) return;
  if (lo == 0 && l0 == 0) {
    stats[2 * sh] = mean;
    stats[2 * sh + 1] = rstd;
  }
  for (int ch = 0; ch < nch; ++ch) {
    const int64_t pe = (int64_t)(ch * 16 + lo) * G.L + l0;
    const int64_t e = (int64_t)sh * G.C * G.L + pe;
    const float4 od = obuf[ch * 64 + lane];
    const float4 w = ld4(ln_w + pe), bb = ld4(ln_b + pe);
    st4(out + e, make_float4((od.x - mean) * rstd * w.x + bb.x, (od.y - mean) * rstd * w.y + bb.y,
                             (od.z - mean) * rstd * w.z + bb.z, (od.w - mean) * rstd * w.w + bb.w));
  }
}

__global__ __launch_bounds__(64) void sdpa_ln_bwd_k(
    const float* __restrict__ gout, const float* __restrict__ gscale, const float* __restrict__ x,
    const float* __restrict__ y, const float* __restrict__ ln_w, const float* __restrict__ stats,
    float* dx, float* dy, uint32_t acc_mask, float* dln_w, float* dln_b, SdpaGeom G, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = G.C / 16;
  float4* xh_buf = reinterpret_cast<float4*>(smem);              // [nch][64]  x_hat (layout V)
  float4* do_buf = xh_buf + nch * 64;                            // [nch][64]  dx_hat, then dO (layout V)
  float* dOt = reinterpret_cast<float*>(do_buf + nch * 64);      // [C][17]    dO as [c][i]
  float* tP = dOt + G.C * 17;                                    // [16][17]
  float* tS = tP + 16 * 17;                                      // [16][17]

  const int lane = threadIdx.x, lo = lane & 15, h = lane >> 4;
  const int g = blockIdx.x;
  float p[4];
  attn_probs(x, y, G, g, lo, h, p);                              // P[i = lo][j = 4h + r]
#pragma unroll
  for (int r = 0; r < 4; ++r) tP[lo * 17 + 4 * h + r] = p[r];
  __syncthreads();
  float pw[4];                                                   // P[i = 4h + r][j = lo]
#pragma unroll
  for (int r = 0; r < 4; ++r) pw[r] = tP[(4 * h + r) * 17 + lo];

  const int sh = g * G.spw + ((4 * h) >> G.Lb);
  const int l0 = (4 * h) & (G.L - 1);
  const bool v_h = sh < G.b;
  const float mean = v_h ? stats[2 * sh] : 0.f;
  const float rstd = v_h ? stats[2 * sh + 1] : 1.f;
  const float gs = (gscale != nullptr) ? gscale[0] : 1.f;

  // pass A: recompute O, x_hat; dx_hat = g*w; LayerNorm reductions; affine gradients
  float s1 = 0.f, s2 = 0.f;
  for (int ch = 0; ch < nch; ++ch) {
    const int64_t pe = (int64_t)(ch * 16 + lo) * G.L + l0;
    const int64_t e = (int64_t)sh * G.C * G.L + pe;
    float4 xh = make_float4(0.f, 0.f, 0.f, 0.f), dxh = xh;
    // MFMAs stay outside divergent control flow: every lane of the wave takes part,
    // padded samples feed zeros.
    const float4 yv = v_h ? ld4(y + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0], yv.x, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[1], yv.y, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[2], yv.z, o, 0, 0, 0);
    o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[3], yv.w, o, 0, 0, 0);
    if (v_h) {
      const float4 m = drop_mult4(drop, (uint64_t)e);
      xh = make_float4((o[0] * m.x - mean) * rstd, (o[1] * m.y - mean) * rstd,
                       (o[2] * m.z - mean) * rstd, (o[3] * m.w - mean) * rstd);
      const float4 gv = f4_scale(ld4(gout + e), gs);
      dxh = f4_mul(gv, ld4(ln_w + pe));
      if (dln_w != nullptr) {
        atomicAdd(dln_w + pe + 0, gv.x * xh.x); atomicAdd(dln_w + pe + 1, gv.y * xh.y);
        atomicAdd(dln_w + pe + 2, gv.z * xh.z); atomicAdd(dln_w + pe + 3, gv.w * xh.w);
        atomicAdd(dln_b + pe + 0, gv.x); atomicAdd(dln_b + pe + 1, gv.y);
        atomicAdd(dln_b + pe + 2, gv.z); atomicAdd(dln_b + pe + 3, gv.w);
      }
    }
    xh_buf[ch * 64 + lane] = xh;
    do_buf[ch * 64 + lane] = dxh;
    s1 += f4_hsum(dxh);
    s2 += f4_dot(dxh, xh);
  }
  const float inv_d = 1.f / (float)(G.C * G.L);
  const float m1 = sample_sum(s1, G.L) * inv_d;
  const float m2 = sample_sum(s2, G.L) * inv_d;

  // pass B: dO = rstd*(dx_hat - m1 - x_hat*m2) * dropout mask;  keep it in both layouts
  for (int ch = 0; ch < nch; ++ch) {
    const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
    const float4 xh = xh_buf[ch * 64 + lane], dxh = do_buf[ch * 64 + lane];
    const float4 m = v_h ? drop_mult4(drop, (uint64_t)e) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 d = make_float4(rstd * (dxh.x - m1 - xh.x * m2) * m.x, rstd * (dxh.y - m1 - xh.y * m2) * m.y,
                                 rstd * (dxh.z - m1 - xh.z * m2) * m.z, rstd * (dxh.w - m1 - xh.w * m2) * m.w);
    do_buf[ch * 64 + lane] = d;
    float* t = dOt + (ch * 16 + lo) * 17 + 4 * h;
    t[0] = d.x; t[1] = d.y; t[2] = d.z; t[3] = d.w;
  }
  __syncthreads();

  // dP[i = lo][j = 4h + r] = sum_c dO[c][i] * y[c][j]   (contraction over channels)
  float ds[4];
  {
    const int s_lo = g * G.spw + (lo >> G.Lb);
    const bool v_lo = s_lo < G.b;
    const float* yb = y + ((int64_t)s_lo * G.C) * G.L + (lo & (G.L - 1));
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < G.C; c0 += 8) {
      const float a0 = v_lo ? yb[(int64_t)(c0 + h) * G.L] : 0.f;
      const float a1 = v_lo ? yb[(int64_t)(c0 + 4 + h) * G.L] : 0.f;
      const float b0 = dOt[(c0 + h) * 17 + lo], b1 = dOt[(c0 + 4 + h) * 17 + lo];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    float rowdot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rowdot += (acc0[r] + acc1[r]) * p[r];
    rowdot += __shfl_xor(rowdot, 16, 64);
    rowdot += __shfl_xor(rowdot, 32, 64);
    const float inv = 1.f / sqrtf((float)G.C);
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[r] = p[r] * ((acc0[r] + acc1[r]) - rowdot) * inv;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tS[lo * 17 + 4 * h + r] = ds[r];
  __syncthreads();
  float dsw[4];                                                  // dS[i = 4h + r][j = lo]
#pragma unroll
  for (int r = 0; r < 4; ++r) dsw[r] = tS[(4 * h + r) * 17 + lo];

  // outputs, per 16-channel chunk, float4 along l
  for (int ch = 0; ch < nch; ++ch) {
    const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 yv = v_h ? ld4(y + e) : z4;
    const float4 xv = v_h ? ld4(x + e) : z4;
    const float4 dov = do_buf[ch * 64 + lane];
    f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ay = {0.f, 0.f, 0.f, 0.f};
    // dx[c][i]  = sum_j dS[i][j] y[c][j]
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[0], yv.x, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[1], yv.y, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[2], yv.z, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[3], yv.w, ax, 0, 0, 0);
    // dy[c][j]  = sum_i dS[i][j] x[c][i] + sum_i P[i][j] dO[c][i]
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[0], xv.x, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[1], xv.y, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[2], xv.z, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[3], xv.w, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[0], dov.x, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[1], dov.y, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[2], dov.z, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[3], dov.w, ay, 0, 0, 0);
    if (!v_h) continue;
    float4 rx = make_float4(ax[0], ax[1], ax[2], ax[3]);
    float4 ry = make_float4(ay[0], ay[1], ay[2], ay[3]);
    if (dy == nullptr) {
      rx = f4_add(rx, ry);
    } else {
      if (acc_mask & 2u) ry = f4_add(ry, ld4(dy + e));
      st4(dy + e, ry);
    }
    if (acc_mask & 1u) rx = f4_add(rx, ld4(dx + e));
    st4(dx + e, rx);
  }
}

inline int geom(int b, int C, int L, SdpaGeom* G) {
  if (!(L == 4 || L == 8 || L == 16) || C % 16 != 0 || C < 16) return BMNAS_E_SHAPE;
  G->b = b; G->C = C; G->L = L; G->Lb = ilog2_exact(L); G->spw = 16 / L;
  return 0;
}

inline DropCfg to_cfg(const bmnas_dropout_t& d) {
  DropCfg c;
  c.thr = d.thr; c.scale = d.scale; c.seed = d.seed; c.offset = d.offset; c.step = d.step;
  return c;
}

}  // namespace

extern "C" int bmnas_sdpa_ln_fwd(const float* x, const float* y, const float* ln_w,
                                 const float* ln_b, float* out, float* stats, int b, int C, int L,
                                 bmnas_dropout_t drop, void* stream) {
  if (!x || !y || !ln_w || !ln_b || !out || !stats || b < 0) return BMNAS_E_ARG;
  SdpaGeom G;
  if (int e = geom(b, C, L, &G)) return e;
  if (b == 0) return 0;
  const size_t lds = (size_t)C * 64;                 // [C/16][64] float4
  if (lds > 160 * 1024) return BMNAS_E_LIMIT;
  const int groups = (b + G.spw - 1) / G.spw;
  hipLaunchKernelGGL(sdpa_ln_fwd_k, dim3(groups), dim3(64), lds, (hipStream_t)stream, x, y, ln_w,
                     ln_b, out, stats, G, to_cfg(drop));
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_sdpa_ln_bwd(const float* g, const float* gscale, const float* x,
                                 const float* y, const float* ln_w, const float* stats, float* dx,
                                 float* dy, uint32_t accumulate_mask, float* dln_w, float* dln_b,
                                 int b, int C, int L, bmnas_dropout_t drop, void* stream) {
  if (!g || !x || !y || !ln_w || !stats || !dx || b < 0) return BMNAS_E_ARG;
  if ((dln_w == nullptr) != (dln_b == nullptr)) return BMNAS_E_ARG;
  SdpaGeom G;
  if (int e = geom(b, C, L, &G)) return e;
  if (b == 0) return 0;
  const size_t lds = (size_t)C * 64 * 2 + (size_t)C * 17 * 4 + 2 * 16 * 17 * 4;
  if (lds > 160 * 1024) return BMNAS_E_LIMIT;
  const int groups = (b + G.spw - 1) / G.spw;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sdpa_ln_bwd_k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(sdpa_ln_bwd_k, dim3(groups), dim3(64), lds, (hipStream_t)stream, g, gscale, x, y,
                     ln_w, stats, dx, dy, accumulate_mask, dln_w, dln_b, G, to_cfg(drop));
  BMNAS_CHECK_LAUNCH();
  return 0;
}
