// K3 — ScaledDotAttn: softmax(x^T y / sqrt(C)) applied to y^T, dropout, LayerNorm([C, L]).
// One 4-wave workgroup owns one 16-row "tile group" = 16/L samples packed on the 16-wide
// MFMA dimension (a block-diagonal mask keeps samples apart); the channel dimension is
// split over the four waves.  Both contractions run on v_mfma_f32_16x16x4_f32 straight
// from global loads in operand layout:
//   S^T[j][i]  = sum_c y[c][j] x[c][i]   contraction over channels: each wave sums its quarter
//                                        of C (dword loads, 256-B segments), partial 16x16
//                                        tiles are added through LDS;
//   O[i][c]    = sum_j P[i][j] y[c][j]   contraction over l: y read as float4 along l, the
//                                        accumulator comes out as float4 along l.
// Row softmax: a lane holds 4 of the 16 scores of row i, the others sit in lanes xor 16 /
// xor 32 -> two wavefront shuffles per reduction.  The per-sample LayerNorm reductions are
// shuffles over the lanes that share the sample plus a 4-entry LDS exchange between waves.
// The forward also saves x_hat (the normalised, pre-affine output) so the backward needs no
// recompute of O.  Memory-bound: forward reads x, y and writes out, x_hat.
#include "sdpa_body.hpp"

namespace {

template <int KCH>
__global__ __launch_bounds__(256) void sdpa_ln_fwd_k(const float* __restrict__ x,
                                                     const float* __restrict__ y,
                                                     const float* __restrict__ ln_w,
                                                     const float* __restrict__ ln_b,
                                                     float* __restrict__ out,
                                                     float* __restrict__ xhat,
                                                     float* __restrict__ stats, SdpaGeom G,
                                                     DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char sdpa_smem[];
  sdpa_fwd_body<KCH>(blockIdx.x, x, y, ln_w, ln_b, out, xhat, stats, G, drop, sdpa_smem);
}

template <int KCH>
__global__ __launch_bounds__(256) void sdpa_ln_bwd_k(
    const float* __restrict__ gout, const float* __restrict__ gscale, const float* __restrict__ x,
    const float* __restrict__ y, const float* __restrict__ ln_w, const float* __restrict__ xhat,
    const float* __restrict__ stats, float* dx, float* dy, uint32_t acc_mask, SdpaGeom G, DropCfg drop) {
  extern __shared__ __attribute__((aligned(16))) char sdpa_smem[];
  sdpa_bwd_body<KCH>(blockIdx.x, gout, gscale, x, y, ln_w, xhat, stats, dx, dy, acc_mask, G, drop, sdpa_smem);
}

}  // namespace

#define SDPA_KCH_DISPATCH(KCH, CALL) \
  do {                               \
    const int k__ = (KCH);           \
    if (k__ <= 1) CALL(1);           \
    else if (k__ <= 2) CALL(2);      \
    else if (k__ <= 3) CALL(3);      \
    else if (k__ <= 4) CALL(4);      \
    else if (k__ <= 6) CALL(6);      \
    else CALL(8);                    \
  } while (0)

extern "C" int bmnas_sdpa_ln_fwd(const float* x, const float* y, const float* ln_w,
                                 const float* ln_b, float* out, float* xhat, float* stats, int b,
                                 int C, int L, bmnas_dropout_t drop, void* stream) {
  if (!x || !y || !ln_w || !ln_b || !out || !xhat || !stats || b < 0) return BMNAS_E_ARG;
  SdpaGeom G;
  if (int e = geom(b, C, L, &G)) return e;
  if (b == 0) return 0;
  const int groups = (b + G.spw - 1) / G.spw;
#define SDPA_F(K)                                                                                     \
  hipLaunchKernelGGL(sdpa_ln_fwd_k<K>, dim3(groups), dim3(256), kSdpaFwdLds, (hipStream_t)stream, x, y, ln_w, ln_b, \
                     out, xhat, stats, G, to_cfg(drop))
  SDPA_KCH_DISPATCH(sdpa_kch(C), SDPA_F);
#undef SDPA_F
  BMNAS_CHECK_LAUNCH();
  return 0;
}

extern "C" int bmnas_sdpa_ln_bwd(const float* g, const float* gscale, const float* x,
                                 const float* y, const float* ln_w, const float* xhat,
                                 const float* stats, float* dx, float* dy, uint32_t accumulate_mask,
                                 int b, int C, int L, bmnas_dropout_t drop, void* stream) {
  if (!g || !x || !y || !ln_w || !xhat || !stats || !dx || b < 0) return BMNAS_E_ARG;
  SdpaGeom G;
  if (int e = geom(b, C, L, &G)) return e;
  if (b == 0) return 0;
  const size_t lds = sdpa_bwd_lds(C);
  const int groups = (b + G.spw - 1) / G.spw;
#define SDPA_B(K)                                                                                      \
  hipLaunchKernelGGL(sdpa_ln_bwd_k<K>, dim3(groups), dim3(256), lds, (hipStream_t)stream, g, gscale, x, y, \
                     ln_w, xhat, stats, dx, dy, accumulate_mask, G, to_cfg(drop))
  SDPA_KCH_DISPATCH(sdpa_kch(C), SDPA_B);
#undef SDPA_B
  BMNAS_CHECK_LAUNCH();
  return 0;
}
