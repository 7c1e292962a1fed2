// central_classifier + loss epilogue: the skinny Linear(M*C*L -> classes)
// that follows the fusion cell (mmimdb_darts_searchable.py:82-83,114) and the criterion applied
// to it by the trainers (BCEWithLogitsLoss mmimdb_darts_searchable.py:22, CrossEntropyLoss
// ntu_darts_searchable.py:25 / ego_darts_searchable.py:24).
//   out[m][o] = bias[o] + sum_k feat[m][k] W[o][k]        b x O x K = 128 x 23 x 6144
// Vendor GEMMs pick poor tiles for O = 23..83 (15 us for the weight gradient); here all three
// products run on v_mfma_f32_16x16x4_f32 with both operands float4 along the contraction
// (k-permuted) in the forward, and the two backward products share one pass over W / feat.
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

constexpr int kMaxTJ = 8;            // classes padded to 16*TJ <= 128

// ---- forward: split-K.  grid = (16-sample row blocks) x (K splits); a workgroup's 4 waves take
// <= 4 blocks of 16 k each (all loads issued before the first MFMA), their partial 16 x 16TJ
// tiles are summed through LDS and added to `out` with atomics (out is zero-filled by the
// host wrapper; the split-0 workgroup also adds the bias).  Before: 8 workgroups walking K
// serially took 32 us; the work is 50 MFLOP.
template <int TJ>
__global__ __launch_bounds__(256) void linear_fwd_k(const float* __restrict__ feat,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ bias,
                                                    float* __restrict__ out, int b, int O, int K,
                                                    int blocks_per_wave) {
  __shared__ float4 part[4][TJ][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int m0 = blockIdx.x * 16;
  int mc = m0 + lo;
  mc = mc < b ? mc : b - 1;                                  // clamped rows are never stored
  const float* fa = feat + (int64_t)mc * K + 4 * h;
  const float* wb[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    int o = 16 * tj + lo;
    o = o < O ? o : O - 1;
    wb[tj] = W + (int64_t)o * K + 4 * h;
  }
  const int nblk = K / 16;
  const int blk0 = (blockIdx.y * 4 + wave) * blocks_per_wave;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 a4[4], b4[4][TJ];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = blk0 + i;
    const bool v = (i < blocks_per_wave) && (blk < nblk);    // wave-uniform
    const int bc = v ? blk : nblk - 1;
    const float4 t = ld4(fa + bc * 16);
    a4[i] = v ? t : z4;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) b4[i][tj] = ld4(wb[tj] + bc * 16);
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 acc[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) acc[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].x, b4[i][tj].x, acc[tj], 0, 0, 0);
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].y, b4[i][tj].y, acc[tj], 0, 0, 0);
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].z, b4[i][tj].z, acc[tj], 0, 0, 0);
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i].w, b4[i][tj].w, acc[tj], 0, 0, 0);
    }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
    part[wave][tj][lane] = make_float4(acc[tj][0], acc[tj][1], acc[tj][2], acc[tj][3]);
  __syncthreads();
  // acc[tj][r] = out[m0 + 4h + r][16 tj + lo]: wave w finishes tiles w, w + 4, ...
  for (int tj = wave; tj < TJ; tj += 4) {
    const float4 s = f4_add(f4_add(part[0][tj][lane], part[1][tj][lane]),
                            f4_add(part[2][tj][lane], part[3][tj][lane]));
    const int o = 16 * tj + lo;
    if (o < O) {
      const float bo = (blockIdx.y == 0) ? bias[o] : 0.f;
      const float v[4] = {s.x + bo, s.y + bo, s.z + bo, s.w + bo};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * h + r;
        if (m < b) atomicAdd(out + (int64_t)m * O + o, v[r]);
      }
    }
  }
}

// the zero fill in front of linear_fwd_k's atomics.  A kernel, not hipMemsetAsync: captured into a hipGraph
// (bmnas.graph.GraphedStep around a classifier that is not fused into the cell's tail) the memset node cleared `out`
// on the first replay only — later replays left 1e21-sized values under the atomics (ROCm 7.2, round 3;
// tools/memset_node_probe.py shows it with nothing but the memset and an add in the graph).
__global__ __launch_bounds__(256) void zero_fill_k(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// ---- backward.  Tiles are oriented with k on the accumulator rows so results leave as float4
// along k.  Both kernels are built for ONE memory round trip per wave (the first version
// walked 8 row blocks x 6 steps + 32 steps serially per wave: 22 us for 100 MFLOP).
//   dfeat[m][k] = gs * sum_o g[m][o] W[o][k]    one wave per (16 samples, 16 k): D[k][m]
__device__ __forceinline__ void linear_dfeat_body(const int bx, const int by, const float* __restrict__ g,
                                                  const float* __restrict__ gscale,
                                                  const float* __restrict__ W,
                                                  float* __restrict__ dfeat, int b, int O, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int kt = bx * 4 + wave;
  if (kt * 16 >= K) return;
  const int k0 = kt * 16, m0 = by * 16;
  int mc = m0 + lo;
  mc = mc < b ? mc : b - 1;
  const float gs = (gscale != nullptr) ? gscale[0] : 1.f;
  const int osteps = (O + 3) / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int t = 0; t < osteps; ++t) {            // A = W^T (lane k = lo, slot -> class), B = g^T
    const int o = 4 * t + h;
    const int oc = o < O ? o : O - 1;
    const float wv = W[(int64_t)oc * K + k0 + lo];           // clamped address + select (no branch)
    const float av = (o < O) ? wv : 0.f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, g[(int64_t)mc * O + oc], acc, 0, 0, 0);
  }
  if (m0 + lo < b)                                             // acc[r] = D[k = 4h + r][m = lo]
    st4_wtg<5>(dfeat + (int64_t)(m0 + lo) * K + k0 + 4 * h,
        make_float4(acc[0] * gs, acc[1] * gs, acc[2] * gs, acc[3] * gs));
}

//   dW[o][k] = gs * sum_m g[m][o] feat[m][k]    workgroup = one 16-k block, its 4 waves split
//   the samples, partial tiles are summed through LDS:  D[k][o]
template <int TJ>
__device__ __forceinline__ void linear_dw_body(const int bx, const float* __restrict__ g,
                                               const float* __restrict__ gscale,
                                               const float* __restrict__ feat,
                                               float* __restrict__ dW, int b, int O, int K) {
  __shared__ float4 part[4][TJ][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = lane & 15, h = lane >> 4;
  const int k0 = bx * 16;
  const float gs = (gscale != nullptr) ? gscale[0] : 1.f;
  const int msteps = (b + 3) / 4;
  const int per = (msteps + 3) / 4;
  const int t0 = wave * per;
  int t1 = t0 + per;
  if (t1 > msteps) t1 = msteps;
  int oc[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) {
    const int o = 16 * tj + lo;
    oc[tj] = o < O ? o : O - 1;
  }
  f32x4 acc[TJ];
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj) acc[tj] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int t = t0; t < t1; ++t) {               // A = feat^T (lane k = lo, slot -> sample), B = g
    const int m = 4 * t + h;
    const int mc = m < b ? m : b - 1;
    const float fv = feat[(int64_t)mc * K + k0 + lo];
    const float av = (m < b) ? fv : 0.f;
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
      acc[tj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, g[(int64_t)mc * O + oc[tj]], acc[tj], 0, 0, 0);
  }
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
    part[wave][tj][lane] = make_float4(acc[tj][0], acc[tj][1], acc[tj][2], acc[tj][3]);
  __syncthreads();
  for (int tj = wave; tj < TJ; tj += 4) {
    const float4 s4 = f4_add(f4_add(part[0][tj][lane], part[1][tj][lane]),
                             f4_add(part[2][tj][lane], part[3][tj][lane]));
    const int o = 16 * tj + lo;
    if (o < O) st4_wtg<5>(dW + (int64_t)o * K + k0 + 4 * h, f4_scale(s4, gs));   // D[k = 4h + r][o = lo]
  }
}

// dbias[o] = gs * sum_m g[m][o]
__device__ __forceinline__ void linear_dbias_body(const int bx, const float* __restrict__ g,
                                                  const float* __restrict__ gscale,
                                                  float* __restrict__ dbias, int b, int O) {
  const int o = bx * 4 + (threadIdx.x >> 6);                  // one wave per class
  if (o >= O) return;
  float s = 0.f;
  for (int m = threadIdx.x & 63; m < b; m += 64) s += g[(int64_t)m * O + o];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) dbias[o] = s * ((gscale != nullptr) ? gscale[0] : 1.f);
}

// the three backward products are independent: one grid carries them all (dfeat tiles first,
// then the dW blocks, then the bias rows); three launches of 3-4 us each took 12.6 us
template <int TJ>
__global__ __launch_bounds__(256) void linear_bwd_k(const float* __restrict__ g,
                                                    const float* __restrict__ gscale,
                                                    const float* __restrict__ feat,
                                                    const float* __restrict__ W, float* __restrict__ dfeat,
                                                    float* __restrict__ dW, float* __restrict__ dbias, int b,
                                                    int O, int K, int fx, int n_feat, int n_dw) {
  const int blk = blockIdx.x;
  if (blk < n_feat) {
    linear_dfeat_body(blk % fx, blk / fx, g, gscale, W, dfeat, b, O, K);
  } else if (blk < n_feat + n_dw) {
    linear_dw_body<TJ>(blk - n_feat, g, gscale, feat, dW, b, O, K);
  } else {
    linear_dbias_body(blk - n_feat - n_dw, g, gscale, dbias, b, O);
  }
}

// ---- losses (mean reduction) with the logits gradient produced in the same pass -----------
// BCEWithLogits: loss = mean(max(z,0) - z*y + log1p(exp(-|z|))), dz = (sigmoid(z) - y) / n
__global__ __launch_bounds__(1024) void bce_logits_k(const float* __restrict__ z,
                                                     const float* __restrict__ y,
                                                     float* __restrict__ loss, float* __restrict__ dz,
                                                     int n) {
  __shared__ float red[16];
  float s = 0.f;
  const float inv = 1.f / (float)n;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const float zi = z[i], yi = y[i];
    s += fmaxf(zi, 0.f) - zi * yi + log1pf(expf(-fabsf(zi)));
    if (dz != nullptr) dz[i] = (1.f / (1.f + expf(-zi)) - yi) * inv;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = t * inv;
  }
}

// CrossEntropy: loss = mean_m(logsumexp(z[m]) - z[m][label]), dz = (softmax - onehot) / b.
// One wave per sample.
__global__ __launch_bounds__(256) void ce_rows_k(const float* __restrict__ z,
                                                 const int64_t* __restrict__ label,
                                                 float* __restrict__ row_loss, float* __restrict__ dz,
                                                 int b, int O) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + wave;
  if (m >= b) return;
  const float* zr = z + (int64_t)m * O;
  float mx = -INFINITY;
  for (int o = lane; o < O; o += 64) mx = fmaxf(mx, zr[o]);
  mx = wave_max(mx);
  float den = 0.f;
  for (int o = lane; o < O; o += 64) den += expf(zr[o] - mx);
  den = wave_sum(den);
  const int lab = (int)label[m];
  if (lane == 0) row_loss[m] = (mx + logf(den)) - zr[lab];
  if (dz != nullptr) {
    const float inv = 1.f / (float)b;
    for (int o = lane; o < O; o += 64)
      dz[(int64_t)m * O + o] = (expf(zr[o] - mx) / den - (o == lab ? 1.f : 0.f)) * inv;
  }
}

// The same for batches of up to 256 samples as ONE workgroup: each wave walks its rows, the mean is formed behind one
// barrier — the found-stage / evaluation criterion as one launch instead of ce_rows_k + mean_k (round 5).
__global__ __launch_bounds__(256) void ce_small_k(const float* __restrict__ z, const int64_t* __restrict__ label,
                                                  float* __restrict__ row_loss, float* __restrict__ dz,
                                                  float* __restrict__ loss, int b, int O) {
  __shared__ float red[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float inv = 1.f / (float)b;
  float acc = 0.f;
  for (int m = wave; m < b; m += 4) {
    const float* zr = z + (int64_t)m * O;
    float mx = -INFINITY;
    for (int o = lane; o < O; o += 64) mx = fmaxf(mx, zr[o]);
    mx = wave_max(mx);
    float den = 0.f;
    for (int o = lane; o < O; o += 64) den += expf(zr[o] - mx);
    den = wave_sum(den);
    const int lab = (int)label[m];
    const float rl = (mx + logf(den)) - zr[lab];
    if (lane == 0) {
      row_loss[m] = rl;
      acc += rl;
    }
    if (dz != nullptr)
      for (int o = lane; o < O; o += 64)
        dz[(int64_t)m * O + o] = (expf(zr[o] - mx) / den - (o == lab ? 1.f : 0.f)) * inv;
  }
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv;
}

__global__ __launch_bounds__(256) void mean_k(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += v[i];
  const float t = block_sum256(s, red);
  if (threadIdx.x == 0) out[0] = t / (float)n;
}

}  // namespace

#define LIN_DISPATCH(TJ, CALL)      \
  switch (TJ) {                     \
    case 1: case 2: CALL(2); break; \
    case 3: case 4: CALL(4); break; \
    case 5: case 6: CALL(6); break; \
    case 7: case 8: CALL(8); break; \
    default: return BMNAS_E_LIMIT;  \
  }

extern "C" int bmnas_linear_fwd(const float* feat, const float* W, const float* bias, float* out,
                                int b, int O, int K, int out_is_zero, void* stream) {
  if (!feat || !W || !bias || !out || b < 0 || O < 1 || K < 1) return BMNAS_E_ARG;
  if (K % 16 != 0) return BMNAS_E_SHAPE;
  if (b == 0) return 0;
  const int tj = (O + 15) / 16;
  if (tj > kMaxTJ) return BMNAS_E_LIMIT;
  hipStream_t st = (hipStream_t)stream;
  if ((int64_t)b * O > (int64_t)1 << 30) return BMNAS_E_LIMIT;
  if (!out_is_zero) hipLaunchKernelGGL(zero_fill_k, dim3((b * O + 255) / 256), dim3(256), 0, st, out, b * O);
  const int nblk = K / 16;
  const int splits = (nblk + 15) / 16;                       // <= 4 blocks per wave, 4 waves
  const int bpw = (nblk + splits * 4 - 1) / (splits * 4);
  dim3 grid((b + 15) / 16, splits);
#define CALL(T) hipLaunchKernelGGL(linear_fwd_k<T>, grid, dim3(256), 0, st, feat, W, bias, out, b, O, K, bpw)
  LIN_DISPATCH(tj, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_linear_bwd(const float* g, const float* gscale, const float* feat,
                                const float* W, float* dfeat, float* dW, float* dbias, int b, int O,
                                int K, void* stream) {
  if (!g || !feat || !W || b < 0 || O < 1 || K < 1) return BMNAS_E_ARG;
  if (K % 16 != 0) return BMNAS_E_SHAPE;
  if (b == 0) return 0;
  const int tj = (O + 15) / 16;
  if (tj > kMaxTJ) return BMNAS_E_LIMIT;
  hipStream_t st = (hipStream_t)stream;
  const int kt = K / 16;
  const int fx = (kt + 3) / 4;
  const int n_feat = dfeat ? fx * ((b + 15) / 16) : 0;
  const int n_dw = dW ? kt : 0;
  const int n_db = dbias ? (O + 3) / 4 : 0;
  if (n_feat + n_dw + n_db == 0) return 0;
#define CALL(T)                                                                                         \
  hipLaunchKernelGGL(linear_bwd_k<T>, dim3(n_feat + n_dw + n_db), dim3(256), 0, st, g, gscale, feat, W, \
                     dfeat, dW, dbias, b, O, K, fx, n_feat, n_dw)
  LIN_DISPATCH(tj, CALL)
#undef CALL
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_bce_logits(const float* z, const float* y, float* loss, float* dz, int n,
                                void* stream) {
  if (!z || !y || !loss || n < 1) return BMNAS_E_ARG;
  hipLaunchKernelGGL(bce_logits_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, y, loss, dz, n);
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_cross_entropy(const float* z, const int64_t* label, float* loss, float* dz,
                                   float* row_loss, int b, int O, void* stream) {
  if (!z || !label || !loss || !row_loss || b < 1 || O < 1) return BMNAS_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (b <= 256) {
    hipLaunchKernelGGL(ce_small_k, dim3(1), dim3(256), 0, st, z, label, row_loss, dz, loss, b, O);
    BMNAS_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(ce_rows_k, dim3((b + 3) / 4), dim3(256), 0, st, z, label, row_loss, dz, b, O);
  BMNAS_CHECK_LAUNCH();
  hipLaunchKernelGGL(mean_k, dim3(1), dim3(256), 0, st, row_loss, loss, b);
  BMNAS_CHECK_LAUNCH();
  return 0;
}
