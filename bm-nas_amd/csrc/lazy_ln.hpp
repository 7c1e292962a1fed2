// The step node's LayerNorm (NodeCell, node_search.py:67-68: out += x; out = ln(out)) applied by the CONSUMERS
// of the node output instead of by a one-workgroup-per-sample kernel.
//
// The producer (bmnas_node_mix_pre_fwd, a streaming grid) stores pre = mix + x un-normalised and, per PART of a
// sample (kLazyPart float4 = one workgroup's share), a record of moments CENTRED ON THE PART'S OWN MEAN:
//   rec[s][k] = { m_k, M2_k = S(c^2), A_k = S(c w), B_k = S(c^2 w^2), C_k = S(c w b), D_k = S(c w^2), -, - },  c = pre - m_k
//   prm[k]    = { E_k = S(w^2), F_k = S(w), G_k = S(w b), H_k = S(b), I_k = S(b^2), n_k, -, - }   (w, b: the affine)
// (plain stores, no atomics: run-to-run deterministic).  A consumer combines the P parts of a sample exactly
// (Chan et al.): mu = S(n_k m_k) / D,  d_k = m_k - mu,  M2 = S(M2_k + n_k d_k^2), and — for the K7 LayerNorm of
// the head, which needs the per-sample sums of the NORMALISED output o = (pre - mu) rstd w + b —
//   S(o)   = rstd S_k(A_k + d_k F_k) + S_k H_k
//   S(o^2) = rstd^2 S_k(B_k + 2 d_k D_k + d_k^2 E_k) + 2 rstd S_k(C_k + d_k G_k) + S_k I_k
// Everything stays centred (d_k is the distance between a part's mean and the sample's), so the precision is that of
// the two-pass form node_mix_ln_fwd_k uses.
#pragma once
#include "common.hpp"

constexpr int kLazyPart = 256;      // float4 per part (= threads of the producer's workgroup)
constexpr int kLazyMaxParts = 4;    // C * L <= 4096 (MM-IMDB: 3 parts; NTU / Ego: 1)
constexpr int kLazyRec = 8;         // floats per record
constexpr float kEpsLazy = 1e-5f;

struct LazyStats {
  float mean, rstd, osum, osq;
};

__host__ __device__ inline int lazy_parts(int cl4) { return (cl4 + kLazyPart - 1) / kLazyPart; }

// rec: (b, P, 8), prm: (P, 8).  Every load is unconditional and issued before the arithmetic.
__device__ __forceinline__ LazyStats lazy_combine(const float* __restrict__ rec, const float* __restrict__ prm,
                                                  const int P, const int s) {
  float4 ra[kLazyMaxParts], rb[kLazyMaxParts], pa[kLazyMaxParts], pb[kLazyMaxParts];
#pragma unroll
  for (int k = 0; k < kLazyMaxParts; ++k) {
    const int kc = k < P ? k : 0;                               // clamped: no predicated loads
    const float* r = rec + ((int64_t)s * P + kc) * kLazyRec;
    ra[k] = ld4(r);
    rb[k] = ld4(r + 4);
    pa[k] = ld4(prm + kc * kLazyRec);
    pb[k] = ld4(prm + kc * kLazyRec + 4);
  }
  float D = 0.f, sm = 0.f;
#pragma unroll
  for (int k = 0; k < kLazyMaxParts; ++k) {
    const float n = k < P ? pb[k].y : 0.f;
    D += n;
    sm += n * ra[k].x;
  }
  const float mu = sm / D;
  float M2 = 0.f, S1 = 0.f, S2 = 0.f, S3 = 0.f, H = 0.f, I = 0.f;
#pragma unroll
  for (int k = 0; k < kLazyMaxParts; ++k) {
    if (k < P) {
      const float d = ra[k].x - mu;
      M2 += ra[k].y + pb[k].y * d * d;
      S1 += ra[k].z + d * pa[k].y;
      S2 += ra[k].w + 2.f * d * rb[k].y + d * d * pa[k].x;
      S3 += rb[k].x + d * pa[k].z;
      H += pa[k].w;
      I += pb[k].x;
    }
  }
  LazyStats o;
  o.mean = mu;
  o.rstd = 1.f / sqrtf(M2 / D + kEpsLazy);
  o.osum = o.rstd * S1 + H;
  o.osq = o.rstd * o.rstd * S2 + 2.f * o.rstd * S3 + I;
  return o;
}
