// K6 / K7 — channel-concat (+ residual) + LayerNorm([D/L, L]) (+ ReLU), one workgroup per
// sample: the whole (D = n_src*C*L)-element sample lives in registers (VPT float4 per
// thread), two block reductions (mean, then centred second moment), one coalesced write.
// Algorithmic bytes fwd: (n_src [+1 resid] + n_src) * T (+ 2*D*4 params), T = b*C*L*4.
#include "common.hpp"
#include <cstdlib>
#include "../../include/bmnas_hip.h"
#include "arch_body.hpp"

namespace {

constexpr float kEps = 1e-5f;

struct LnSrc {
  const float* p[4];
};
struct LnDst {
  float* p[4];
};

// float4 index i (within the concatenated sample) -> pointer into source q
__device__ __forceinline__ const float* src_ptr(const LnSrc& s, int i4, int cl4, int sample) {
  const int q = i4 / cl4;
  const int off = i4 - q * cl4;
  return pick_ptr(s.p, q) + ((int64_t)sample * cl4 + off) * 4;      // (never s.p[q]: an indexed kernarg LOAD, common.hpp)
}

// BS = threads per workgroup: 512 when the batch leaves CUs idle (one workgroup per sample), so
// that a sample is spread over twice the waves and each thread's dependent chain is half as long.
template <int VPT, int BS>
__global__ __launch_bounds__(BS) void cat_ln_fwd_k(LnSrc srcs, const float* __restrict__ resid,
                                                    const float* __restrict__ ln_w,
                                                    const float* __restrict__ ln_b,
                                                    float* __restrict__ out,
                                                    float* __restrict__ stats, int cl4, int d4,
                                                    int relu, float* __restrict__ osum) {
  __shared__ float red[8];
  __shared__ float red6[8 * 6];
  const int s = blockIdx.x;
  float4 v[VPT], lw[VPT], lb[VPT];
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * BS;
    if (i < d4) {
      v[k] = ld4(src_ptr(srcs, i, cl4, s));
      if (resid != nullptr) v[k] = f4_add(v[k], ld4(resid + ((int64_t)s * d4 + i) * 4));
      lw[k] = ld4(ln_w + (int64_t)i * 4);          // requested with the sample: no second round
      lb[k] = ld4(ln_b + (int64_t)i * 4);          // trip after the two reductions
      sum += f4_hsum(v[k]);
    } else {
      v[k] = lw[k] = lb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float inv_d = 1.f / (float)(d4 * 4);
  const float mean = block_sum_fresh<BS / 64>(sum, red) * inv_d;     // (first use of red: no barrier in front)
  // second pass: the centred second moment and — same reduction round, when asked for — what the
  // per-sample sums of the OUTPUT need (see node_mix_ln_fwd_k; without ReLU only)
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool sums = osum != nullptr && !relu;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * BS;
    if (i < d4) {
      const float4 c = make_float4(v[k].x - mean, v[k].y - mean, v[k].z - mean, v[k].w - mean);
      acc[0] += f4_dot(c, c);
      if (sums) {
        const float4 cw = f4_mul(c, lw[k]);
        acc[1] += f4_hsum(cw);
        acc[2] += f4_dot(cw, cw);
        acc[3] += f4_dot(cw, lb[k]);
        acc[4] += f4_hsum(lb[k]);
        acc[5] += f4_dot(lb[k], lb[k]);
      }
    }
  }
  if (sums) {
    block_sum_lead_fresh<BS / 64, 6>(acc, red6);
  } else {
    acc[0] = block_sum_fresh<BS / 64>(acc[0], red6);         // (red6, not red: threads may still be reading the mean)
  }
  const float var = acc[0] * inv_d;
  const float rstd = 1.f / sqrtf(var + kEps);
  if (threadIdx.x == 0) {
    stats[2 * s] = mean;
    stats[2 * s + 1] = rstd;
    if (sums) {
      osum[2 * s] = rstd * acc[1] + acc[4];
      osum[2 * s + 1] = rstd * rstd * acc[2] + 2.f * rstd * acc[3] + acc[5];
    }
  }
  float os = 0.f, oq = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * BS;
    if (i < d4) {
      const float4 w = lw[k], bb = lb[k];
      float4 o;
      o.x = (v[k].x - mean) * rstd * w.x + bb.x;
      o.y = (v[k].y - mean) * rstd * w.y + bb.y;
      o.z = (v[k].z - mean) * rstd * w.z + bb.z;
      o.w = (v[k].w - mean) * rstd * w.w + bb.w;
      if (relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      st4_wtg<3>(out + ((int64_t)s * d4 + i) * 4, o);
      os += f4_hsum(o);
      oq += f4_dot(o, o);
    }
  }
  // with ReLU the sums cannot be derived from the moments: reduce the outputs themselves
  if (osum != nullptr && relu) {
    os = block_sum_fresh<BS / 64>(os, red);                  // (red's readers all passed the second reduction's barrier)
    oq = block_sum<BS / 64>(oq, red6);
    if (threadIdx.x == 0) {
      osum[2 * s] = os;
      osum[2 * s + 1] = oq;
    }
  }
}

template <int VPT, int BS>
__global__ __launch_bounds__(BS) void cat_ln_bwd_k(const float* __restrict__ g, LnSrc srcs,
                                                    const float* __restrict__ resid,
                                                    const float* __restrict__ ln_w,
                                                    const float* __restrict__ ln_b,
                                                    const float* __restrict__ stats, LnDst dsrcs,
                                                    float* dresid, uint32_t acc_mask, float* dln_w,
                                                    float* dln_b, int cl4, int d4, int relu,
                                                    float* __restrict__ scrub, int64_t scrub4) {
  __shared__ float red[2 * (BS / 64)];
  const int s = blockIdx.x;
  // side job: clear the caller's accumulation arena (saves a memset launch per backward)
  for (int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x; i < scrub4; i += (int64_t)gridDim.x * BS)
    st4_wtg<3>(scrub + 4 * i, make_float4(0.f, 0.f, 0.f, 0.f));
  const float mean = stats[2 * s], rstd = stats[2 * s + 1];
  float4 xh[VPT], dxh[VPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * BS;
    if (i < d4) {
      float4 x = ld4(src_ptr(srcs, i, cl4, s));
      if (resid != nullptr) x = f4_add(x, ld4(resid + ((int64_t)s * d4 + i) * 4));
      const float4 w = ld4(ln_w + (int64_t)i * 4);
      float4 gy = ld4(g + ((int64_t)s * d4 + i) * 4);
      float4 h;
      h.x = (x.x - mean) * rstd; h.y = (x.y - mean) * rstd;
      h.z = (x.z - mean) * rstd; h.w = (x.w - mean) * rstd;
      if (relu) {
        const float4 bb = ld4(ln_b + (int64_t)i * 4);
        if (h.x * w.x + bb.x <= 0.f) gy.x = 0.f;
        if (h.y * w.y + bb.y <= 0.f) gy.y = 0.f;
        if (h.z * w.z + bb.z <= 0.f) gy.z = 0.f;
        if (h.w * w.w + bb.w <= 0.f) gy.w = 0.f;
      }
      if (dln_w != nullptr) {
        float* pw = dln_w + (int64_t)i * 4;
        float* pb = dln_b + (int64_t)i * 4;
        atomicAdd(pw + 0, gy.x * h.x); atomicAdd(pw + 1, gy.y * h.y);
        atomicAdd(pw + 2, gy.z * h.z); atomicAdd(pw + 3, gy.w * h.w);
        atomicAdd(pb + 0, gy.x); atomicAdd(pb + 1, gy.y);
        atomicAdd(pb + 2, gy.z); atomicAdd(pb + 3, gy.w);
      }
      const float4 d = f4_mul(gy, w);
      xh[k] = h;
      dxh[k] = d;
      s1 += f4_hsum(d);
      s2 += f4_dot(d, h);
    } else {
      xh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      dxh[k] = xh[k];
    }
  }
  const float inv_d = 1.f / (float)(d4 * 4);
  // both sums behind ONE barrier (two block_sum calls were four; `red` is written once per launch)
  constexpr int NWV = BS / 64;
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s1;
    red[NWV + (threadIdx.x >> 6)] = s2;
  }
  __syncthreads();
  float m1 = red[0], m2 = red[NWV];
#pragma unroll
  for (int w = 1; w < NWV; ++w) {
    m1 += red[w];
    m2 += red[NWV + w];
  }
  m1 *= inv_d;
  m2 *= inv_d;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int i = threadIdx.x + k * BS;
    if (i < d4) {
      float4 dx;
      dx.x = rstd * (dxh[k].x - m1 - xh[k].x * m2);
      dx.y = rstd * (dxh[k].y - m1 - xh[k].y * m2);
      dx.z = rstd * (dxh[k].z - m1 - xh[k].z * m2);
      dx.w = rstd * (dxh[k].w - m1 - xh[k].w * m2);
      const int q = i / cl4;
      const int off = i - q * cl4;
      float* d = pick_ptr(dsrcs.p, q);
      if (d != nullptr) {
        float* a = d + ((int64_t)s * cl4 + off) * 4;
        st4_wtg<3>(a, (acc_mask & (1u << q)) ? f4_add(dx, ld4(a)) : dx);
      }
      if (dresid != nullptr) {
        float* a = dresid + ((int64_t)s * d4 + i) * 4;
        st4_wtg<3>(a, (acc_mask & (1u << 31)) ? f4_add(dx, ld4(a)) : dx);
      }
    }
  }
}


// LayerNorm affine gradients dln_w[e] = sum_s gy[s,e]*x_hat[s,e], dln_b[e] = sum_s gy[s,e]:
// a reduction over SAMPLES per element.  Workgroup = 64 float4 columns x 4 sample lanes
// (one wave per sample lane -> 1 KiB coalesced rows); each thread walks its share of the
// sample chunk, the four lanes are summed through LDS, one atomic per element per chunk
// (b/16 x fewer atomics than doing it per sample).  prenorm: srcs[0] already holds x_hat.
struct LnAffineProb {
  const float* g;
  const float* gscale;
  LnSrc srcs;
  const float* resid;
  const float* ln_w;
  const float* ln_b;
  const float* stats;
  float* dln_w;
  float* dln_b;
  int cl4, d4, relu, prenorm;
};

constexpr int kMaxLnProbs = 8;
struct LnAffineBatch {
  LnAffineProb p[kMaxLnProbs];
  int n, b, chunk;
};

__device__ __forceinline__ void ln_affine_body(const LnAffineProb& P, int b, int chunk, float4 (*red)[3][64]) {
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  const bool active = i < P.d4;
  const float gs = (P.gscale != nullptr) ? P.gscale[0] : 1.f;
  float4 aw = make_float4(0.f, 0.f, 0.f, 0.f), ab = aw;
  if (active) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f), bb = w;
    if (P.relu) {
      w = ld4(P.ln_w + (int64_t)i * 4);
      bb = ld4(P.ln_b + (int64_t)i * 4);
    }
    const int s_beg = blockIdx.y * chunk;
    int s_end = s_beg + chunk;
    if (s_end > b) s_end = b;
    for (int s = s_beg + sl; s < s_end; s += 4) {
      float4 x = ld4(src_ptr(P.srcs, i, P.cl4, s));
      float4 gy = f4_scale(ld4(P.g + ((int64_t)s * P.d4 + i) * 4), gs);
      float4 h = x;
      if (!P.prenorm) {
        if (P.resid != nullptr) x = f4_add(x, ld4(P.resid + ((int64_t)s * P.d4 + i) * 4));
        const float mean = P.stats[2 * s], rstd = P.stats[2 * s + 1];
        h = make_float4((x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd);
      }
      if (P.relu) {
        if (h.x * w.x + bb.x <= 0.f) gy.x = 0.f;
        if (h.y * w.y + bb.y <= 0.f) gy.y = 0.f;
        if (h.z * w.z + bb.z <= 0.f) gy.z = 0.f;
        if (h.w * w.w + bb.w <= 0.f) gy.w = 0.f;
      }
      aw = f4_add(aw, f4_mul(gy, h));
      ab = f4_add(ab, gy);
    }
  }
  if (sl > 0) {
    red[0][sl - 1][col] = aw;
    red[1][sl - 1][col] = ab;
  }
  __syncthreads();
  if (sl == 0 && active) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      aw = f4_add(aw, red[0][k][col]);
      ab = f4_add(ab, red[1][k][col]);
    }
    float* pw = P.dln_w + (int64_t)i * 4;
    float* pb = P.dln_b + (int64_t)i * 4;
    atomicAdd(pw + 0, aw.x); atomicAdd(pw + 1, aw.y); atomicAdd(pw + 2, aw.z); atomicAdd(pw + 3, aw.w);
    atomicAdd(pb + 0, ab.x); atomicAdd(pb + 1, ab.y); atomicAdd(pb + 2, ab.z); atomicAdd(pb + 3, ab.w);
  }
}

// blockIdx.z selects the problem: every LayerNorm of a backward pass (K7, the K6 of each step
// node, the attention LNs) in ONE launch at the end of the pass instead of five ~4.5 us ones.
__global__ __launch_bounds__(256) void ln_affine_bwd_k(LnAffineBatch B) {
  __shared__ float4 red[2][3][64];
  const LnAffineProb& P = B.p[blockIdx.z];
  if ((int)blockIdx.x * 64 >= P.d4) return;              // block-uniform: this problem is narrower
  ln_affine_body(P, B.b, B.chunk, red);
}

// Partial sums left by the head's backward (head.hip): out[e] = sum_c part[c][e], float4 stream.
constexpr int kMaxSums = 2;
struct SumPack {
  const float* part[kMaxSums];
  float* out[kMaxSums];
  long long n4[kMaxSums];
  int n_chunk[kMaxSums];
  int n;
  int reps;                      // z-slices (of gridDim.x * gridDim.y workgroups) per sum
};

// The two launches that end a fused cell's backward, as one: blockIdx.z < B.n are the LayerNorm
// affine problems above, the z = B.n slice walks the rows of the architecture tensors (four rows
// per workgroup, one wavefront each) for the row-softmax backward.  Independent work.
__global__ __launch_bounds__(256) void backward_epilogue_k(LnAffineBatch B, ArchPack A, int arch_rows, SumPack S) {
  __shared__ float4 red[2][3][64];
  if ((int)blockIdx.z > B.n) {                                   // z = B.n + 1 + i * reps + r: slice r of the i-th partial sum
    const int k = (int)blockIdx.z - B.n - 1;
    const int i = k / S.reps, rep = k - i * S.reps;
    const long long per = (long long)gridDim.x * gridDim.y;
    const long long n4 = S.n4[i], wg = rep * per + (long long)blockIdx.y * gridDim.x + blockIdx.x;
    const float* __restrict__ part = S.part[i];
    for (long long e = wg * 256 + threadIdx.x; e < n4; e += (long long)S.reps * per * 256) {
      float4 t = ld4(part + 4 * e);
      for (int c = 1; c < S.n_chunk[i]; ++c) t = f4_add(t, ld4(part + 4 * (e + (long long)c * n4)));
      st4_wtg<3>(S.out[i] + 4 * e, t);
    }
    return;
  }
  if ((int)blockIdx.z == B.n) {
    const int r = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * 4 + (int)(threadIdx.x >> 6);
    if (r < arch_rows) arch_softmax_bwd_row(A, r, threadIdx.x & 63);
    return;
  }
  const LnAffineProb& P = B.p[blockIdx.z];
  if ((int)blockIdx.x * 64 >= P.d4) return;
  ln_affine_body(P, B.b, B.chunk, red);
}

inline int pick_vpt(int d4, int bs) {
  const int need = (d4 + bs - 1) / bs;
  if (need <= 1) return 1;
  if (need <= 2) return 2;
  if (need <= 3) return 3;
  if (need <= 4) return 4;
  if (need <= 8) return 8;
  if (need <= 16) return 16;
  return -1;
}

}  // namespace

#define LN_DISPATCH(V, CALL)      \
  switch (V) {                    \
    case 1: CALL(1); break;       \
    case 2: CALL(2); break;       \
    case 3: CALL(3); break;       \
    case 4: CALL(4); break;       \
    case 8: CALL(8); break;       \
    case 16: CALL(16); break;     \
    default: return BMNAS_E_LIMIT; \
  }

extern "C" int bmnas_cat_ln_fwd(const float* const* srcs, int n_src, const float* resid,
                                const float* ln_w, const float* ln_b, float* out, float* stats,
                                int b, int C, int L, int relu, float* out_sums, void* stream) {
  if (!srcs || !ln_w || !ln_b || !out || !stats || n_src < 1 || b < 0 || C < 1 || L < 1)
    return BMNAS_E_ARG;
  if (n_src > 4) return BMNAS_E_LIMIT;
  if (resid && n_src != 1) return BMNAS_E_ARG;
  if ((C * L) % 4 != 0) return BMNAS_E_SHAPE;
  if (b == 0) return 0;
  LnSrc s{};
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    s.p[q] = srcs[q];
  }
  const int cl4 = C * L / 4, d4 = cl4 * n_src;
  const bool wide = b <= 256 && d4 >= 512;          // fewer samples than CUs: 8 waves per sample
  const int vpt = pick_vpt(d4, wide ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
#define CALL(V)                                                                                         \
  do {                                                                                                  \
    if (wide) hipLaunchKernelGGL((cat_ln_fwd_k<V, 512>), dim3(b), dim3(512), 0, st, s, resid, ln_w, ln_b, out, stats, cl4, d4, relu, out_sums); \
    else hipLaunchKernelGGL((cat_ln_fwd_k<V, 256>), dim3(b), dim3(256), 0, st, s, resid, ln_w, ln_b, out, stats, cl4, d4, relu, out_sums);      \
  } while (0)
  LN_DISPATCH(vpt, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_cat_ln_bwd(const float* g, const float* const* srcs, int n_src,
                                const float* resid, const float* ln_w, const float* ln_b,
                                const float* stats, float* const* dsrcs, float* dresid,
                                uint32_t accumulate_mask, float* dln_w, float* dln_b, int b, int C,
                                int L, int relu, float* scrub, int64_t scrub_n, void* stream) {
  if (!g || !srcs || !ln_w || !ln_b || !stats || !dsrcs || n_src < 1 || b < 0 || C < 1 || L < 1)
    return BMNAS_E_ARG;
  if (scrub_n < 0 || (scrub_n > 0 && !scrub) || scrub_n % 4) return BMNAS_E_ARG;
  if ((dln_w == nullptr) != (dln_b == nullptr)) return BMNAS_E_ARG;
  if (n_src > 4) return BMNAS_E_LIMIT;
  if (resid && n_src != 1) return BMNAS_E_ARG;
  if ((C * L) % 4 != 0) return BMNAS_E_SHAPE;
  if (b == 0) return 0;
  LnSrc s{};
  LnDst d{};
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    s.p[q] = srcs[q];
    d.p[q] = dsrcs[q];
  }
  const int cl4 = C * L / 4, d4 = cl4 * n_src;
  const bool wide = b <= 256 && d4 >= 512;
  const int vpt = pick_vpt(d4, wide ? 512 : 256);
  hipStream_t st = (hipStream_t)stream;
#define CALL(V)                                                                                         \
  do {                                                                                                  \
    if (wide) hipLaunchKernelGGL((cat_ln_bwd_k<V, 512>), dim3(b), dim3(512), 0, st, g, s, resid, ln_w, ln_b, stats, d, dresid, accumulate_mask, dln_w, dln_b, cl4, d4, relu, scrub, scrub_n / 4); \
    else hipLaunchKernelGGL((cat_ln_bwd_k<V, 256>), dim3(b), dim3(256), 0, st, g, s, resid, ln_w, ln_b, stats, d, dresid, accumulate_mask, dln_w, dln_b, cl4, d4, relu, scrub, scrub_n / 4);      \
  } while (0)
  LN_DISPATCH(vpt, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

namespace {
int fill_prob(LnAffineProb& P, const float* g, const float* gscale, const float* const* srcs, int n_src,
              const float* resid, const float* ln_w, const float* ln_b, const float* stats, float* dln_w,
              float* dln_b, int C, int L, int relu, int prenorm) {
  if (!g || !srcs || !dln_w || !dln_b || n_src < 1 || C < 1 || L < 1) return BMNAS_E_ARG;
  if (!prenorm && !stats) return BMNAS_E_ARG;
  if (relu && (!ln_w || !ln_b)) return BMNAS_E_ARG;
  if (n_src > 4) return BMNAS_E_LIMIT;
  if (resid && n_src != 1) return BMNAS_E_ARG;
  if ((C * L) % 4 != 0) return BMNAS_E_SHAPE;
  P = LnAffineProb{};
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    P.srcs.p[q] = srcs[q];
  }
  P.g = g; P.gscale = gscale; P.resid = resid; P.ln_w = ln_w; P.ln_b = ln_b; P.stats = stats;
  P.dln_w = dln_w; P.dln_b = dln_b;
  P.cl4 = C * L / 4; P.d4 = P.cl4 * n_src; P.relu = relu; P.prenorm = prenorm;
  return 0;
}

// samples per workgroup of the LayerNorm affine reductions: 16 up to 256 samples (smaller chunks = more
// atomics: measured slower, 6.5 vs 4.6 us at b = 128), then b / 16 — every chunk ends in one atomic per
// element, and at b = 1024 sixteen-sample chunks meant 1.5 M atomics per pass
// Deterministic mode (bmnas_set_deterministic): ONE chunk — every element receives a single add onto its zero-filled
// slot, so the result does not depend on the order workgroups finish in.
int g_ln_deterministic = 0;
inline int ln_affine_chunk(int b) {
  if (g_ln_deterministic) return b;
  return b <= 256 ? 16 : (b + 15) / 16;
}

int launch_ln_affine(LnAffineBatch& B, hipStream_t st) {
  B.chunk = ln_affine_chunk(B.b);
  int maxd4 = 0;
  for (int i = 0; i < B.n; ++i) maxd4 = B.p[i].d4 > maxd4 ? B.p[i].d4 : maxd4;
  dim3 grid((maxd4 + 63) / 64, (B.b + B.chunk - 1) / B.chunk, B.n);
  hipLaunchKernelGGL(ln_affine_bwd_k, grid, dim3(256), 0, st, B);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
}  // namespace

extern "C" int bmnas_ln_affine_bwd(const float* g, const float* gscale, const float* const* srcs,
                                   int n_src, const float* resid, const float* ln_w,
                                   const float* ln_b, const float* stats, float* dln_w, float* dln_b,
                                   int b, int C, int L, int relu, int prenorm, void* stream) {
  if (b < 0) return BMNAS_E_ARG;
  LnAffineBatch B{};
  if (int e = fill_prob(B.p[0], g, gscale, srcs, n_src, resid, ln_w, ln_b, stats, dln_w, dln_b, C, L, relu,
                        prenorm))
    return e;
  if (b == 0) return 0;
  B.n = 1;
  B.b = b;
  return launch_ln_affine(B, (hipStream_t)stream);
}

extern "C" int bmnas_ln_affine_bwd_multi(int n_prob, const float* const* g, const float* const* gscale,
                                         const float* const* const* srcs, const int* n_src,
                                         const float* const* resid, const float* const* ln_w,
                                         const float* const* ln_b, const float* const* stats,
                                         float* const* dln_w, float* const* dln_b, int b, const int* C,
                                         int L, const int* relu, const int* prenorm, void* stream) {
  if (n_prob < 1 || b < 0 || !g || !srcs || !n_src || !dln_w || !dln_b || !C || !relu || !prenorm)
    return BMNAS_E_ARG;
  if (n_prob > kMaxLnProbs) return BMNAS_E_LIMIT;
  LnAffineBatch B{};
  for (int i = 0; i < n_prob; ++i)
    if (int e = fill_prob(B.p[i], g[i], gscale ? gscale[i] : nullptr, srcs[i], n_src[i],
                          resid ? resid[i] : nullptr, ln_w ? ln_w[i] : nullptr, ln_b ? ln_b[i] : nullptr,
                          stats ? stats[i] : nullptr, dln_w[i], dln_b[i], C[i], L, relu[i], prenorm[i]))
      return e;
  if (b == 0) return 0;
  B.n = n_prob;
  B.b = b;
  return launch_ln_affine(B, (hipStream_t)stream);
}

extern "C" int bmnas_backward_epilogue(int n_prob, const float* const* g, const float* const* gscale,
                                       const float* const* const* srcs, const int* n_src,
                                       const float* const* resid, const float* const* ln_w,
                                       const float* const* ln_b, const float* const* stats,
                                       float* const* dln_w, float* const* dln_b, int b, const int* C,
                                       int L, const int* relu, const int* prenorm,
                                       const float* const* arch_w, const float* const* arch_dw,
                                       float* const* arch_out, const int* arch_rows, const int* arch_cols,
                                       int n_arch, int n_shards, int64_t shard_stride, int n_sums,
                                       const float* const* sum_part, float* const* sum_out,
                                       const int* sum_chunks, const int64_t* sum_n, void* stream) {
  if (n_prob < 1 || b < 1 || !g || !srcs || !n_src || !dln_w || !dln_b || !C || !relu || !prenorm)
    return BMNAS_E_ARG;
  if (n_prob > kMaxLnProbs || n_sums > kMaxSums) return BMNAS_E_LIMIT;
  if (n_sums < 0 || (n_sums > 0 && (!sum_part || !sum_out || !sum_chunks || !sum_n))) return BMNAS_E_ARG;
  SumPack S{};
  for (int i = 0; i < n_sums; ++i) {
    if (!sum_part[i] || !sum_out[i] || sum_chunks[i] < 1 || sum_n[i] < 0 || sum_n[i] % 4) return BMNAS_E_ARG;
    S.part[i] = sum_part[i]; S.out[i] = sum_out[i]; S.n_chunk[i] = sum_chunks[i]; S.n4[i] = sum_n[i] / 4;
  }
  S.n = n_sums;
  LnAffineBatch B{};
  for (int i = 0; i < n_prob; ++i)
    if (int e = fill_prob(B.p[i], g[i], gscale ? gscale[i] : nullptr, srcs[i], n_src[i],
                          resid ? resid[i] : nullptr, ln_w ? ln_w[i] : nullptr, ln_b ? ln_b[i] : nullptr,
                          stats ? stats[i] : nullptr, dln_w[i], dln_b[i], C[i], L, relu[i], prenorm[i]))
      return e;
  ArchPack A{};
  // (n_arch == 0: no architecture tensors — the per-op path's end-of-backward launch: affine reductions + chunk sums)
  const int total = n_arch > 0 ? fill_arch_pack(A, arch_w, arch_dw, arch_out, arch_rows, arch_cols, n_arch, 1, n_shards,
                                                shard_stride)
                               : 0;
  if (total < 0) return total;
  B.n = n_prob;
  B.b = b;
  B.chunk = ln_affine_chunk(B.b);
  int maxd4 = 0;
  for (int i = 0; i < B.n; ++i) maxd4 = B.p[i].d4 > maxd4 ? B.p[i].d4 : maxd4;
  dim3 grid((maxd4 + 63) / 64, (B.b + B.chunk - 1) / B.chunk, B.n + 1 + n_sums);
  // the arch slice needs one wavefront per row, four per workgroup (LayerNorm workgroups past a
  // problem's width return at once, so widening the grid for tiny shapes costs nothing)
  const unsigned need_x = (unsigned)((total + 4 * (int)grid.y - 1) / (4 * (int)grid.y));
  if (grid.x < need_x) grid.x = need_x;
  // ... and a partial sum gets one workgroup per 256 float4 of its output, up to 256 workgroups: on NTU / Ego's narrow
  // LayerNorm grids (4 column blocks x 3-4 sample chunks) the head's (O + 3) x D partials were walked by 12-16
  // workgroups, 8-14 rounds of n_chunk dependent loads each — the longest chain of the launch (Ego b48: 11.3 us)
  // (as S.reps z-slices per sum rather than a wider grid.x: that would multiply the LayerNorm slices' idle workgroups
  // too — measured +0.9 us at NTU b64)
  long long max_n4 = 0;
  for (int i = 0; i < n_sums; ++i) max_n4 = S.n4[i] > max_n4 ? S.n4[i] : max_n4;
  long long want_wg = (max_n4 + 255) / 256;
  if (want_wg > 256) want_wg = 256;
  const long long per = (long long)grid.x * grid.y;
  S.reps = (int)((want_wg + per - 1) / per);
  if (S.reps < 1) S.reps = 1;
  grid.z = (unsigned)(B.n + 1 + n_sums * S.reps);
  hipLaunchKernelGGL(backward_epilogue_k, grid, dim3(256), 0, (hipStream_t)stream, B, A, total, S);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

// BMNAS_DETERMINISTIC (bmnas.cell.DETERMINISTIC): run-to-run bit-identical results.  The host library keeps one
// flag per source that has a choice to make (no relocatable device code / shared globals in this build): here the
// LayerNorm-affine reductions take one sample chunk, in conv1x1.hip the weight-gradient tiles walk the whole batch.
extern "C" int bmnas_ln_set_deterministic(int on) {
  g_ln_deterministic = on ? 1 : 0;
  return 0;
}
