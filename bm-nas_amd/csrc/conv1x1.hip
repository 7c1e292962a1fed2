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
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

struct ConvArgs {
  PtrsIn act;            // n_act sources, each (b, Ci, L): contraction rows i = q*Ci + ci
  PtrsOut dst;           // n_dst destinations, each (b, Cj, L): output cols j = q*Cj + cj
  const float* W;
  const float* bias;     // per output col (fwd), nullable
  float* part;           // BN partial stats (fwd, training), nullable
  int ldw, Ci, Cj, I, J;
  int b, L, Lb, spw, n_groups, n_part;
  uint32_t acc_mask;
};

// OUT[n][j] = sum_i ACT[i][n] * MAT(i, j);  TRANS: MAT(i,j) = W[j*ldw + i] (forward conv),
// else MAT(i,j) = W[i*ldw + j] (data gradient).  Workgroup = 4 waves as 2 (n) x 2 (j);
// wave tile = 32 n x 32 j = 2x2 MFMA tiles.
template <bool TRANS>
__global__ __launch_bounds__(256) void conv_nj_k(ConvArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int wn = wave & 1, wj = wave >> 1;
  const int prow = blockIdx.x * 2 + wn;       // index of this wave's 32-column block
  const int g0 = prow * 2;
  const int j0 = (blockIdx.y * 2 + wj) * 32;
  if (g0 >= a.n_groups || j0 >= a.J) return;  // wave-uniform; no barriers below

  // A operand addressing: column n = lo of group g0+tn
  bool va[2];
  int64_t abase[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int g = g0 + tn;
    const int s = g * a.spw + (lo >> a.Lb);
    va[tn] = (g < a.n_groups) && (s < a.b);
    abase[tn] = ((int64_t)s * a.Ci) * a.L + (lo & (a.L - 1));
  }
  bool vj[2];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) vj[tj] = (j0 + 16 * tj) < a.J;   // J % 16 == 0

  f32x4 acc[2][2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) acc[tn][tj] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int i0 = 0; i0 < a.I; i0 += 16) {
    const int q = i0 / a.Ci;
    const int ci = i0 - q * a.Ci + 4 * h;
    const float* src = a.act.p[q];
    float av[2][4];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const float* p = src + abase[tn] + (int64_t)ci * a.L;
#pragma unroll
      for (int r = 0; r < 4; ++r) av[tn][r] = va[tn] ? p[(int64_t)r * a.L] : 0.f;
    }
    float bv[2][4];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int j = j0 + 16 * tj + lo;
      if (!vj[tj]) {
        bv[tj][0] = bv[tj][1] = bv[tj][2] = bv[tj][3] = 0.f;
      } else if (TRANS) {
        const float4 w4 = ld4(a.W + (int64_t)j * a.ldw + i0 + 4 * h);
        bv[tj][0] = w4.x; bv[tj][1] = w4.y; bv[tj][2] = w4.z; bv[tj][3] = w4.w;
      } else {
        const float* p = a.W + (int64_t)(i0 + 4 * h) * a.ldw + j;
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[tj][r] = p[(int64_t)r * a.ldw];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[tn][tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tn][r], bv[tj][r], acc[tn][tj], 0, 0, 0);
  }

  // epilogue.  acc[tn][tj][r] = OUT[n = 16*(g0+tn) + 4h + r][j = j0 + 16*tj + lo]
  bool vo[2];
  int so[2];
  const int l0 = (4 * h) & (a.L - 1);
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int g = g0 + tn;
    so[tn] = g * a.spw + ((4 * h) >> a.Lb);
    vo[tn] = (g < a.n_groups) && (so[tn] < a.b);
  }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    if (!vj[tj]) continue;
    const int j = j0 + 16 * tj + lo;
    const float bj = (a.bias != nullptr) ? a.bias[j] : 0.f;
    const int q = j / a.Cj;
    const int cj = j - q * a.Cj;
    float* d = a.dst.p[q];
    float4 o[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      o[tn] = make_float4(acc[tn][tj][0] + bj, acc[tn][tj][1] + bj, acc[tn][tj][2] + bj,
                          acc[tn][tj][3] + bj);
      if (vo[tn] && d != nullptr) {
        float* pp = d + ((int64_t)so[tn] * a.Cj + cj) * a.L + l0;
        st4(pp, (a.acc_mask & (1u << q)) ? f4_add(o[tn], ld4(pp)) : o[tn]);
      }
    }
    if (a.part != nullptr) {
      // per-channel partial statistics over this wave's (<= 32) valid columns
      float sum = (vo[0] ? f4_hsum(o[0]) : 0.f) + (vo[1] ? f4_hsum(o[1]) : 0.f);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      int cnt = a.b * a.L - 32 * prow;
      cnt = cnt > 32 ? 32 : cnt;
      const float mean = sum / (float)cnt;
      float m2 = 0.f;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        if (vo[tn]) {
          const float4 c = make_float4(o[tn].x - mean, o[tn].y - mean, o[tn].z - mean, o[tn].w - mean);
          m2 += f4_dot(c, c);
        }
      m2 += __shfl_xor(m2, 16, 64);
      m2 += __shfl_xor(m2, 32, 64);
      if (h == 0) {
        float* pp = a.part + ((int64_t)j * a.n_part + prow) * 2;
        pp[0] = sum;
        pp[1] = m2;
      }
    }
  }
}

struct ConvWArgs {
  const float* dU;       // (b, M, L)
  PtrsIn src;            // n_src sources (b, C_src, L); K = n_src * C_src
  float* dW;
  float* dbias;          // nullable
  int ldw, C_src, M, K, dup_cols;
  int b, L, Lb, spw, n_groups, groups_per_split;
  int use_atomic;
};

// dW[m][k] += sum_n dU[m][n] * X[k][n].  Workgroup = 8 waves on ONE 64(m) x 32(k) output
// tile, each wave striding over the n-groups of the workgroup's split; partial tiles are
// summed through LDS in wave order (deterministic inside the workgroup), then written
// with coalesced stores (single split) or fp32 atomics (several splits).
__global__ __launch_bounds__(512) void conv_w_k(ConvWArgs a) {
  __shared__ float tile[64 * 33];
  __shared__ float brow[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int m0 = blockIdx.x * 64, k0 = blockIdx.y * 32;
  const int gbeg = blockIdx.z * a.groups_per_split;
  int gend = gbeg + a.groups_per_split;
  if (gend > a.n_groups) gend = a.n_groups;
  const bool want_bias = (a.dbias != nullptr) && (blockIdx.y == 0);

  bool vm[4], vk[2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) vm[tm] = (m0 + 16 * tm) < a.M;
#pragma unroll
  for (int tk = 0; tk < 2; ++tk) vk[tk] = (k0 + 16 * tk) < a.K;
  // per-lane row offsets (in floats, without the sample term)
  int64_t aoff[4], boff[2];
  const float* bsrc[2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) aoff[tm] = (int64_t)(m0 + 16 * tm + lo) * a.L;
#pragma unroll
  for (int tk = 0; tk < 2; ++tk) {
    const int k = k0 + 16 * tk + lo;
    const int q = vk[tk] ? k / a.C_src : 0;
    bsrc[tk] = a.src.p[q];
    boff[tk] = (int64_t)(k - q * a.C_src) * a.L;
  }
  const int l0 = (4 * h) & (a.L - 1);
  const int sh = (4 * h) >> a.Lb;

  f32x4 acc[4][2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tk = 0; tk < 2; ++tk) acc[tm][tk] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};

  for (int g = gbeg + wave; g < gend; g += 8) {
    const int s = g * a.spw + sh;
    const bool vs = s < a.b;
    float4 av[4], bv[2];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
      av[tm] = (vs && vm[tm]) ? ld4(a.dU + (int64_t)s * a.M * a.L + aoff[tm] + l0)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tk = 0; tk < 2; ++tk)
      bv[tk] = (vs && vk[tk]) ? ld4(bsrc[tk] + (int64_t)s * a.C_src * a.L + boff[tk] + l0)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    if (want_bias) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) bsum[tm] += f4_hsum(av[tm]);
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].x, bv[tk].x, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].y, bv[tk].y, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].z, bv[tk].z, acc[tm][tk], 0, 0, 0);
        acc[tm][tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[tm].w, bv[tk].w, acc[tm][tk], 0, 0, 0);
      }
  }

  // cross-wave reduction: acc[tm][tk][r] = dW[m0 + 16tm + 4h + r][k0 + 16tk + lo]
  if (want_bias) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      bsum[tm] += __shfl_xor(bsum[tm], 16, 64);
      bsum[tm] += __shfl_xor(bsum[tm], 32, 64);
    }
  }
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tk = 0; tk < 2; ++tk)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* t = &tile[(16 * tm + 4 * h + r) * 33 + 16 * tk + lo];
            *t = (w == 0) ? acc[tm][tk][r] : (*t + acc[tm][tk][r]);
          }
      if (want_bias && h == 0) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
          float* t = &brow[16 * tm + lo];
          *t = (w == 0) ? bsum[tm] : (*t + bsum[tm]);
        }
      }
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 64 * 32; e += 512) {
    const int mm = e >> 5, kk = e & 31;
    const int m = m0 + mm, k = k0 + kk;
    if (m < a.M && k < a.K) {
      const float v = tile[mm * 33 + kk];
      float* p = a.dW + (int64_t)m * a.ldw + k;
      if (a.use_atomic) {
        atomicAdd(p, v);
        if (a.dup_cols > 0) atomicAdd(p + a.dup_cols, v);
      } else {
        *p += v;
        if (a.dup_cols > 0) p[a.dup_cols] += v;
      }
    }
  }
  if (want_bias && threadIdx.x < 64) {
    const int m = m0 + threadIdx.x;
    if (m < a.M) {
      if (a.use_atomic) atomicAdd(a.dbias + m, brow[threadIdx.x]);
      else a.dbias[m] += brow[threadIdx.x];
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
    st4(Weff + (int64_t)m * C + 4 * c4, f4_add(ld4(r), ld4(r + C)));
  }
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
  return (ng + 1) / 2;
}

extern "C" int bmnas_conv1x1_fwd(const float* const* srcs, int n_src, int C_src, const float* W,
                                 int ldw, const float* bias, float* U, float* part, int b, int L,
                                 int M, void* stream) {
  if (!srcs || !W || !U || n_src < 1 || C_src < 1 || b < 0 || M < 1) return BMNAS_E_ARG;
  if (n_src > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw % 4 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.act.p[q] = srcs[q];
  }
  a.dst.p[0] = U;
  a.W = W; a.bias = bias; a.part = part; a.ldw = ldw;
  a.Ci = C_src; a.I = n_src * C_src; a.Cj = M; a.J = M;
  a.b = b; a.L = L; a.acc_mask = 0; a.n_part = (a.n_groups + 1) / 2;
  dim3 grid((a.n_groups + 3) / 4, (M + 63) / 64);
  hipLaunchKernelGGL(conv_nj_k<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_bwd_data(const float* dU, const float* W, int ldw,
                                      float* const* dsrcs, int n_src, int C_src,
                                      uint32_t accumulate_mask, int b, int L, int M, void* stream) {
  if (!dU || !W || !dsrcs || n_src < 1 || C_src < 1 || b < 0 || M < 1) return BMNAS_E_ARG;
  if (n_src > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src) return BMNAS_E_SHAPE;
  ConvArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  a.act.p[0] = dU;
  for (int q = 0; q < n_src; ++q) a.dst.p[q] = dsrcs[q];
  a.W = W; a.bias = nullptr; a.part = nullptr; a.ldw = ldw;
  a.Ci = M; a.I = M; a.Cj = C_src; a.J = n_src * C_src;
  a.b = b; a.L = L; a.acc_mask = accumulate_mask;
  dim3 grid((a.n_groups + 3) / 4, (a.J + 63) / 64);
  hipLaunchKernelGGL(conv_nj_k<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_conv1x1_bwd_weight(const float* dU, const float* const* srcs, int n_src,
                                        int C_src, float* dW, int ldw, float* dbias, int dup_cols,
                                        int b, int L, int M, void* stream) {
  if (!dU || !srcs || !dW || n_src < 1 || C_src < 1 || b < 0 || M < 1 || dup_cols < 0)
    return BMNAS_E_ARG;
  if (n_src > BMNAS_MAX_PTRS) return BMNAS_E_LIMIT;
  if (C_src % 16 || M % 16 || ldw < n_src * C_src + dup_cols) return BMNAS_E_SHAPE;
  ConvWArgs a{};
  if (int e = check_shape(b, L, &a.Lb, &a.spw, &a.n_groups)) return e;
  if (b == 0) return 0;
  for (int q = 0; q < n_src; ++q) {
    if (!srcs[q]) return BMNAS_E_ARG;
    a.src.p[q] = srcs[q];
  }
  a.dU = dU; a.dW = dW; a.dbias = dbias; a.ldw = ldw; a.C_src = C_src; a.M = M;
  a.K = n_src * C_src; a.dup_cols = dup_cols; a.b = b; a.L = L;
  const int tiles = ((M + 63) / 64) * ((a.K + 31) / 32);
  // enough workgroups to cover the chip about twice, never fewer than 8 groups per split
  int splits = (512 + tiles - 1) / tiles;
  const int max_splits = (a.n_groups + 7) / 8;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.groups_per_split = (a.n_groups + splits - 1) / splits;
  splits = (a.n_groups + a.groups_per_split - 1) / a.groups_per_split;
  a.use_atomic = splits > 1;
  dim3 grid((M + 63) / 64, (a.K + 31) / 32, splits);
  hipLaunchKernelGGL(conv_w_k, grid, dim3(512), 0, (hipStream_t)stream, a);
  BMNAS_CHECK_LAUNCH();
  return 0;
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
