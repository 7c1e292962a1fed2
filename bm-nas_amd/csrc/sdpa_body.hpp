// Device-side bodies of the K3 kernels (see sdpa.hip for the design notes).  Kept in a header so
// that the same code can run as its own launch (sdpa.hip) or share a launch with the conv GEMM
// that consumes the same input (conv1x1.hip: the 128 attention workgroups only fill half of the
// 256 CUs, the GEMM tiles fill the rest).
#pragma once
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr float kEps = 1e-5f;
constexpr int kSdpaFwdLds = 4096 + 64;                 // bytes of LDS sdpa_fwd_body needs
constexpr int kSdpaBwdFixedLds = 4096 + 2 * 4352 + 64;  // ... sdpa_bwd_body, before its [C][17] transpose buffer
constexpr int kMaxCh = 8;          // 16-channel chunks per wave: C <= 4 * 8 * 16 = 512
// The bodies are templated on KCH = chunks per wave actually needed (ceil(C / 64)): the operand
// prefetch arrays are sized by it, and a fixed size of 8 cost 212 VGPRs and 5 redundant (clamped)
// loads per array at C = 192.
inline int sdpa_kch(int C) { return (C / 16 + 3) / 4; }

struct SdpaGeom {
  int b, C, L, Lb, spw;
};

// sum over the lanes of this wave that belong to the same sample as this lane
__device__ __forceinline__ float sample_sum(float v, int L) {
  v = row16_sum(v);
  if (L >= 8) v = xor16_sum(v);
  if (L >= 16) v = xor32_sum(v);
  return v;
}

// per-sample sum across the workgroup: red is [4 waves][4 sample slots]
__device__ __forceinline__ float wg_sample_sum(float v, int L, int Lb, float (*red)[4], int wave, int lo,
                                               int h) {
  v = sample_sum(v, L);
  const int slot = (4 * h) >> Lb;
  __syncthreads();
  if (lo == 0 && ((4 * h) & (L - 1)) == 0) red[wave][slot] = v;
  __syncthreads();
  return red[0][slot] + red[1][slot] + red[2][slot] + red[3][slot];
}

// One MFMA k-step of the channel contraction: A = yb[c][.] (rows j), B = xb[c][.] (cols i)
#define SDPA_KSTEP(ACC, T)                                                         \
  do {                                                                             \
    const int64_t o_ = (int64_t)(4 * (T) + h) * G.L;                               \
    const float a_ = v_lo ? yb[o_] : 0.f, b_ = v_lo ? xb[o_] : 0.f;                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, ACC, 0, 0, 0);              \
  } while (0)

// Softmax probabilities P[i = lo][j = 4h + r] of this tile group, identical in all 4 waves.
__device__ __forceinline__ void attn_probs(const float* __restrict__ x, const float* __restrict__ y,
                                           const SdpaGeom& G, int g, int wave, int lane, float4* ldsS,
                                           float p[4]) {
  const int lo = lane & 15, h = lane >> 4;
  // padded samples read the last valid sample (clamped address, no predicated loads); the
  // block-diagonal mask keeps them away from real rows and nothing of theirs is stored
  int s_lo = g * G.spw + (lo >> G.Lb);
  s_lo = s_lo < G.b ? s_lo : G.b - 1;
  const bool v_lo = true;
  const int64_t base = ((int64_t)s_lo * G.C) * G.L + (lo & (G.L - 1));
  const float* xb = x + base;
  const float* yb = y + base;
  const int per = G.C / 16;                       // k-steps (of 4 channels) per wave
  const int t0 = wave * per, t1 = t0 + per;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  int t = t0;
  for (; t + 3 < t1; t += 4) {                    // 8 loads in flight, then 4 MFMAs
    const int64_t o0 = (int64_t)(4 * t + h) * G.L, st = (int64_t)4 * G.L;
    const float a0 = v_lo ? yb[o0] : 0.f, b0 = v_lo ? xb[o0] : 0.f;
    const float a1 = v_lo ? yb[o0 + st] : 0.f, b1 = v_lo ? xb[o0 + st] : 0.f;
    const float a2 = v_lo ? yb[o0 + 2 * st] : 0.f, b2 = v_lo ? xb[o0 + 2 * st] : 0.f;
    const float a3 = v_lo ? yb[o0 + 3 * st] : 0.f, b3 = v_lo ? xb[o0 + 3 * st] : 0.f;
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc1, 0, 0, 0);
  }
  for (; t < t1; ++t) SDPA_KSTEP(acc0, t);
  ldsS[wave * 64 + lane] = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2],
                                       acc0[3] + acc1[3]);
  __syncthreads();
  const float4 s0 = ldsS[lane], s1 = ldsS[64 + lane], s2 = ldsS[128 + lane], s3 = ldsS[192 + lane];
  const float raw[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y),
                        (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
  // lane holds S[i = lo][j = 4h + r]; keep only j in the same sample as i
  const float inv = 1.f / sqrtf((float)G.C);
  const bool same = ((4 * h) >> G.Lb) == (lo >> G.Lb);
  float sc[4], mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    sc[r] = same ? raw[r] * inv : -INFINITY;
    mx = fmaxf(mx, sc[r]);
  }
  mx = xor16_max(mx);
  mx = xor32_max(mx);
  float den = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    p[r] = same ? __expf(sc[r] - mx) : 0.f;
    den += p[r];
  }
  den = xor16_sum(den);
  den = xor32_sum(den);
  const float rden = 1.f / den;
#pragma unroll
  for (int r = 0; r < 4; ++r) p[r] *= rden;
}

// g = tile group (one workgroup of 256 threads each)
template <int KCH>
__device__ __forceinline__ void sdpa_fwd_body(const int g, const float* __restrict__ x,
                                              const float* __restrict__ y,
                                              const float* __restrict__ ln_w,
                                              const float* __restrict__ ln_b, float* __restrict__ out,
                                              float* __restrict__ xhat, float* __restrict__ stats,
                                              const SdpaGeom& G, const DropCfg& drop, char* lds) {
  const DropRt drop_rt = drop_begin(drop);          // the step counter's load goes out first
  // LDS comes from the caller (kSdpaFwdLds bytes, 16-byte aligned) so that a launch that merges
  // this body with others pays max(), not sum(), of their footprints
  float4* ldsS = reinterpret_cast<float4*>(lds);                 // [4 * 64]
  float (*red)[4] = reinterpret_cast<float (*)[4]>(lds + 4096);  // [4][4]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int sh = g * G.spw + ((4 * h) >> G.Lb);
  const int l0 = (4 * h) & (G.L - 1);
  const bool v_h = sh < G.b;
  const int shc = v_h ? sh : G.b - 1;                    // clamped: padded samples are never stored
  const int nch = G.C / 16;
  // Everything this wave will need later (its y chunks, the LayerNorm affine rows) is requested
  // BEFORE the score computation: one memory round trip for the whole kernel instead of three
  // dependent ones (the kernel is latency bound: 128 workgroups, ~40 KB each).
  float4 yv[KCH], od[KCH], lw[KCH], lb[KCH];
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    const int chc = ch < nch ? ch : nch - 1;
    const int64_t pe = (int64_t)(chc * 16 + lo) * G.L + l0;
    yv[k] = ld4(y + (int64_t)shc * G.C * G.L + pe);
    lw[k] = ld4(ln_w + pe);
    lb[k] = ld4(ln_b + pe);
  }
  float p[4];
  attn_probs(x, y, G, g, wave, lane, ldsS, p);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    od[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch < nch) {                                      // wave-uniform
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[0], yv[k].x, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[1], yv[k].y, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[2], yv[k].z, o, 0, 0, 0);
      o = __builtin_amdgcn_mfma_f32_16x16x4f32(p[3], yv[k].w, o, 0, 0, 0);
      const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
      const float4 m = v_h ? drop_mult4(drop_rt, (uint64_t)e) : make_float4(0.f, 0.f, 0.f, 0.f);
      od[k] = make_float4(o[0] * m.x, o[1] * m.y, o[2] * m.z, o[3] * m.w);
      sum += f4_hsum(od[k]);
    }
  }
  const float inv_d = 1.f / (float)(G.C * G.L);
  const float mean = wg_sample_sum(sum, G.L, G.Lb, red, wave, lo, h) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    if (wave + 4 * k < nch) {
      const float4 c = make_float4(od[k].x - mean, od[k].y - mean, od[k].z - mean, od[k].w - mean);
      sq += f4_dot(c, c);
    }
  }
  const float var = wg_sample_sum(sq, G.L, G.Lb, red, wave, lo, h) * inv_d;
  const float rstd = 1.f / sqrtf(var + kEps);
  if (!v_h) return;
  if (wave == 0 && lo == 0 && l0 == 0) {
    stats[2 * sh] = mean;
    stats[2 * sh + 1] = rstd;
  }
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    if (ch < nch) {
      const int64_t pe = (int64_t)(ch * 16 + lo) * G.L + l0;
      const int64_t e = (int64_t)sh * G.C * G.L + pe;
      const float4 w = lw[k], bb = lb[k];
      const float4 hh = make_float4((od[k].x - mean) * rstd, (od[k].y - mean) * rstd,
                                    (od[k].z - mean) * rstd, (od[k].w - mean) * rstd);
      st4_w0<2>(xhat + e, hh);
      st4_w0<2>(out + e, make_float4(hh.x * w.x + bb.x, hh.y * w.y + bb.y, hh.z * w.z + bb.z, hh.w * w.w + bb.w));
    }
  }
}

template <int KCH>
__device__ __forceinline__ void sdpa_bwd_body(
    const int g, const float* __restrict__ gout, const float* __restrict__ gscale,
    const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ ln_w,
    const float* __restrict__ xhat, const float* __restrict__ stats, float* dx, float* dy,
    uint32_t acc_mask, const SdpaGeom& G, const DropCfg& drop, char* lds) {
  const DropRt drop_rt = drop_begin(drop);          // the step counter's load goes out first
  // LDS from the caller: sdpa_bwd_lds(C) bytes, 16-byte aligned (see sdpa_fwd_body)
  float4* ldsS = reinterpret_cast<float4*>(lds);                                  // [4 * 64]
  float (*tP)[16 * 17] = reinterpret_cast<float (*)[16 * 17]>(lds + 4096);        // [4][272]
  float (*tS)[16 * 17] = reinterpret_cast<float (*)[16 * 17]>(lds + 4096 + 4352); // [4][272]
  float (*red)[4] = reinterpret_cast<float (*)[4]>(lds + 4096 + 2 * 4352);        // [4][4]
  float* dOt = reinterpret_cast<float*>(lds + kSdpaBwdFixedLds);                  // [C][17]: dO as [c][i]

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int nch = G.C / 16;
  const int sh = g * G.spw + ((4 * h) >> G.Lb);
  const int l0 = (4 * h) & (G.L - 1);
  const bool v_h = sh < G.b;
  const int shc = v_h ? sh : G.b - 1;
  // all of this wave's streaming operands are requested up front (one round trip, see forward)
  float4 xh[KCH], dv[KCH], yv[KCH], xv[KCH];
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    const int chc = ch < nch ? ch : nch - 1;
    const int64_t pe = (int64_t)(chc * 16 + lo) * G.L + l0;
    const int64_t e = (int64_t)shc * G.C * G.L + pe;
    xh[k] = ld4(xhat + e);
    dv[k] = f4_mul(ld4(gout + e), ld4(ln_w + pe));               // g * w (the gamma scale comes later)
    yv[k] = ld4(y + e);
    xv[k] = ld4(x + e);
  }
  float p[4];
  attn_probs(x, y, G, g, wave, lane, ldsS, p);                   // P[i = lo][j = 4h + r]
#pragma unroll
  for (int r = 0; r < 4; ++r) tP[wave][lo * 17 + 4 * h + r] = p[r];
  __syncthreads();
  float pw[4];                                                   // P[i = 4h + r][j = lo]
#pragma unroll
  for (int r = 0; r < 4; ++r) pw[r] = tP[wave][(4 * h + r) * 17 + lo];

  const float rstd = stats[2 * shc + 1];
  const float gs = (gscale != nullptr) ? gscale[0] : 1.f;

  // pass A: dx_hat = g * w and the two LayerNorm-backward reductions
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    if (ch < nch && v_h) {
      dv[k] = f4_scale(dv[k], gs);
      s1 += f4_hsum(dv[k]);
      s2 += f4_dot(dv[k], xh[k]);
    } else {
      xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      dv[k] = xh[k];
    }
  }
  const float inv_d = 1.f / (float)(G.C * G.L);
  const float m1 = wg_sample_sum(s1, G.L, G.Lb, red, wave, lo, h) * inv_d;
  const float m2 = wg_sample_sum(s2, G.L, G.Lb, red, wave, lo, h) * inv_d;

  // pass B: dO = rstd * (dx_hat - m1 - x_hat * m2) * dropout mask, kept in registers (lane =
  // channel, regs = rows i) and transposed into LDS for the contraction over channels
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    if (ch < nch) {
      const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
      const float4 m = v_h ? drop_mult4(drop_rt, (uint64_t)e) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 d = make_float4(rstd * (dv[k].x - m1 - xh[k].x * m2) * m.x,
                                   rstd * (dv[k].y - m1 - xh[k].y * m2) * m.y,
                                   rstd * (dv[k].z - m1 - xh[k].z * m2) * m.z,
                                   rstd * (dv[k].w - m1 - xh[k].w * m2) * m.w);
      dv[k] = d;
      float* t = dOt + (ch * 16 + lo) * 17 + 4 * h;
      t[0] = d.x; t[1] = d.y; t[2] = d.z; t[3] = d.w;
    }
  }
  __syncthreads();

  // dP[i = lo][j = 4h + r] = sum_c dO[c][i] * y[c][j]: each wave contracts its quarter of C
  float ds[4];
  {
    int s_lo = g * G.spw + (lo >> G.Lb);
    s_lo = s_lo < G.b ? s_lo : G.b - 1;                  // clamped; dO of padded samples is zero
    const bool v_lo = true;
    const float* yb = y + ((int64_t)s_lo * G.C) * G.L + (lo & (G.L - 1));
    const int per = G.C / 16;
    const int t0 = wave * per, t1 = t0 + per;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    int t = t0;
    for (; t + 1 < t1; t += 2) {
      const int c0 = 4 * t + h, c1 = c0 + 4;
      const float a0 = v_lo ? yb[(int64_t)c0 * G.L] : 0.f, a1 = v_lo ? yb[(int64_t)c1 * G.L] : 0.f;
      const float b0 = dOt[c0 * 17 + lo], b1 = dOt[c1 * 17 + lo];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc1, 0, 0, 0);
    }
    if (t < t1) {
      const int c0 = 4 * t + h;
      const float a0 = v_lo ? yb[(int64_t)c0 * G.L] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, dOt[c0 * 17 + lo], acc0, 0, 0, 0);
    }
    __syncthreads();                                   // ldsS is being reused
    ldsS[wave * 64 + lane] = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2],
                                         acc0[3] + acc1[3]);
    __syncthreads();
    const float4 q0 = ldsS[lane], q1 = ldsS[64 + lane], q2 = ldsS[128 + lane], q3 = ldsS[192 + lane];
    const float dP[4] = {(q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y),
                         (q0.z + q1.z) + (q2.z + q3.z), (q0.w + q1.w) + (q2.w + q3.w)};
    float rowdot = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rowdot += dP[r] * p[r];
    rowdot = xor16_sum(rowdot);
    rowdot = xor32_sum(rowdot);
    const float inv = 1.f / sqrtf((float)G.C);
#pragma unroll
    for (int r = 0; r < 4; ++r) ds[r] = p[r] * (dP[r] - rowdot) * inv;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) tS[wave][lo * 17 + 4 * h + r] = ds[r];
  __syncthreads();
  float dsw[4];                                                  // dS[i = 4h + r][j = lo]
#pragma unroll
  for (int r = 0; r < 4; ++r) dsw[r] = tS[wave][(4 * h + r) * 17 + lo];

  // outputs, per 16-channel chunk, float4 along l (operands were loaded at the top)
#pragma unroll
  for (int k = 0; k < KCH; ++k) {
    const int ch = wave + 4 * k;
    if (ch >= nch) continue;                                     // wave-uniform
    const int64_t e = ((int64_t)sh * G.C + ch * 16 + lo) * G.L + l0;
    f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ay = {0.f, 0.f, 0.f, 0.f};
    // dx[c][i]  = sum_j dS[i][j] y[c][j]
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[0], yv[k].x, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[1], yv[k].y, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[2], yv[k].z, ax, 0, 0, 0);
    ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[3], yv[k].w, ax, 0, 0, 0);
    // dy[c][j]  = sum_i dS[i][j] x[c][i] + sum_i P[i][j] dO[c][i]
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[0], xv[k].x, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[1], xv[k].y, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[2], xv[k].z, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(dsw[3], xv[k].w, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[0], dv[k].x, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[1], dv[k].y, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[2], dv[k].z, ay, 0, 0, 0);
    ay = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[3], dv[k].w, ay, 0, 0, 0);
    if (v_h) {
      float4 rx = make_float4(ax[0], ax[1], ax[2], ax[3]);
      float4 ry = make_float4(ay[0], ay[1], ay[2], ay[3]);
      if (dy == nullptr) {
        rx = f4_add(rx, ry);
      } else {
        if (acc_mask & 2u) ry = f4_add(ry, ld4(dy + e));
        st4_w0<8>(dy + e, ry);
      }
      if (acc_mask & 1u) rx = f4_add(rx, ld4(dx + e));
      st4_w0<8>(dx + e, rx);
    }
  }
}

inline int geom(int b, int C, int L, SdpaGeom* G) {
  if (!(L == 4 || L == 8 || L == 16) || C % 16 != 0 || C < 16) return BMNAS_E_SHAPE;
  if (C > 4 * kMaxCh * 16) return BMNAS_E_LIMIT;
  G->b = b; G->C = C; G->L = L; G->Lb = ilog2_exact(L); G->spw = 16 / L;
  return 0;
}

inline DropCfg to_cfg(const bmnas_dropout_t& d) {
  DropCfg c;
  c.thr = d.thr; c.scale = d.scale; c.seed = d.seed; c.offset = d.offset; c.step = d.step;
  return c;
}


inline size_t sdpa_bwd_lds(int C) { return kSdpaBwdFixedLds + (size_t)C * 17 * 4; }

}  // namespace
