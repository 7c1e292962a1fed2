/* bmnas_hip.h — C ABI of libbmnas_hip.so: the MI355X (gfx950) kernels of the BM-NAS
 * fusion-search hot path.
 *
 * The reference (Somedaywilldo/BM-NAS) is pure Python/PyTorch: it has no FFI for this
 * path; its "interface" is the nn.Module surface of models/search/darts/{operations,node_operations,model_search,node_search}.py.  This
 * library sits UNDER that surface: the host-side mirror (bm-nas_amd/models/...) keeps the
 * reference's class names/signatures and calls these entry points through ctypes from
 * torch.autograd.Function bodies.  Each entry point below cites the reference code it
 * replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - every tensor pointer is a DEVICE pointer to contiguous fp32, caller-owned (allocated
 *     by PyTorch's caching allocator); feature tensors are (b, C, L) with L innermost;
 *   - `xs`/`dxs`-style arguments are HOST arrays of device pointers (<= BMNAS_MAX_PTRS);
 *   - `stream` is a hipStream_t; work is enqueued asynchronously, nothing synchronises,
 *     nothing allocates, no mutable global state (re-entrant per stream, capturable
 *     into a hipGraph);
 *   - return value: 0 = ok, > 0 = hipError_t from the launch, < 0 = argument error
 *     (BMNAS_E_*); no exception crosses the boundary;
 *   - shape limits: L in {4, 8, 16}; C % 16 == 0; at most 16 pointers per list.
 *   - "accumulate" flags: 0 -> destination is overwritten, 1 -> destination += result.
 *   - reductions over the batch (arch-weight dot products, LayerNorm / BatchNorm affine
 *     gradients, split-N weight gradients) use fp32 atomics into buffers the CALLER has
 *     zeroed (or that hold a running sum to add to).
 */
#ifndef BMNAS_HIP_H
#define BMNAS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMNAS_MAX_PTRS 16
#define BMNAS_E_ARG (-1)      /* null pointer / non-positive size                        */
#define BMNAS_E_SHAPE (-2)    /* shape outside the supported set (see limits above)      */
#define BMNAS_E_LIMIT (-3)    /* too many pointers / LDS budget exceeded                 */

/* Dropout descriptor: Philox4x32-10 counter RNG, element e of a tensor is kept iff
 * philox(seed, base + offset + e/4)[e%4] >= thr (thr = p * 2^32), and scaled by
 * `scale` = 1/(1-p); base = *step if step != NULL else 0.  `step` is a DEVICE counter so a
 * captured hipGraph draws fresh masks on every replay (the graph itself advances it).
 * thr == 0 means identity (eval mode, or p == 0).  The backward call passes the SAME
 * descriptor and regenerates the mask — no mask tensor is stored.  Replaces nn.Dropout
 * at node_operations.py:27,38 / :46,55 / :89,105 and node_search.py:42,64. */
typedef struct {
  uint32_t thr;
  float scale;
  uint64_t seed;
  uint64_t offset;
  const uint64_t* step;
} bmnas_dropout_t;

int bmnas_version(void);

/* The multipliers of ONE dropout site as a tensor: out[e] = (element e kept ? drop.scale : 0) for
 * e in [0, n_elem) — exactly what every kernel that takes `drop` applies to flat element e of that site's
 * (b, C, L) output (nn.Dropout at node_operations.py:38, :55, :105, node_search.py:64, aux_models.py:114).
 * Audit entry point: lets a CPU checker re-run a dropout-ON step under the SAME masks (torch's generator
 * cannot be reproduced bit for bit; this Philox stream can be exported).  With drop.step != NULL the device
 * counter is read when the launch executes, like in the kernels.  Nothing on the hypernet path calls it. */
int bmnas_dropout_mask(bmnas_dropout_t drop, int64_t n_elem, float* out, void* stream);

/* ---- K1: architecture-weighted mixed-edge sum ---------------------------------------
 * out[e] = sum_j w[j*w_stride] * xs[j][e]   over n_elem elements.
 * Replaces sum(FusionMixedOp_j(h_j, weights[offset+j])) at model_search.py:58 and
 * node_search.py:54 (FusionMixedOp.forward operations.py:104-105 with Zero :18-20 and
 * Identity :92-93).  w points at the 'skip' column of the softmaxed edge rows
 * (w_stride = 2).  The 'none' primitive contributes w0*(x*0) = 0 for finite x and is not
 * evaluated (differs from the reference only for non-finite inputs; the host mirror's debug switch
 * BMNAS_STRICT_ZERO=1 adds sum_j (x_j * 0.) of the cell inputs back onto every step's sum, so that NaN / Inf
 * inputs propagate element for element as in the reference — bmnas.cell.STRICT_ZERO). */
int bmnas_mixsum_fwd(const float* const* xs, int n_in, const float* w, int w_stride,
                     float* out, int64_t n_elem, void* stream);
/* dxs[j] (=|+=) w_j * g  (dxs[j] may be NULL to skip; bit j of accumulate_mask selects +=);
 * dw[shard*dw_shard_stride + j*w_stride] += <g, xs[j]> (atomic; dw may be NULL to skip the dot
 * products).  The adds of the workgroups are spread round-robin over dw_shards (>= 1) copies of
 * the buffer so that they do not serialise on n_in addresses; the consumer sums the copies
 * (bmnas_arch_softmax_multi does).  g2 (nullable): a second tensor added to g on load — the
 * gradient of the summed state arrives in two parts when bmnas_conv1x1_bwd_all_sdpa produced it. */
int bmnas_mixsum_bwd(const float* const* xs, float* const* dxs, int n_in, const float* w,
                     int w_stride, const float* g, const float* g2, float* dw, int dw_shards,
                     int64_t dw_shard_stride, uint32_t accumulate_mask, int64_t n_elem,
                     void* stream);

/* K1 pair (search mode): out = sum_j w_j xs[j] and, from the same registers, the first inner sum
 * of the step node that consumes it, out2 = (w2[0] + w2[w2_stride]) * out — FusionNode(h, h) at
 * model_search.py:59 makes NodeCell's states [h, h] (node_search.py:52-54). */
int bmnas_mixsum_pair_fwd(const float* const* xs, int n_in, const float* w, int w_stride,
                          const float* w2, int w2_stride, float* out, float* out2, int64_t n_elem,
                          void* stream);
/* Backward of the pair.  h = the saved `out`; gz = gradient of out2; gh = gradient `out` received
 * from its other consumers (NULL if none).  G = gh + (w2_0 + w2_1) gz;  dxs[j] (=|+=) w_j G;
 * dw[j] += <G, xs[j]>;  dw2[0], dw2[w2_stride] += <gz, h>  (n_in <= 15; shards as above, dw and dw2
 * share dw_shard_stride).  gz2 (nullable) is added to gz on load, as g2 above. */
int bmnas_mixsum_pair_bwd(const float* const* xs, float* const* dxs, int n_in, const float* w,
                          int w_stride, const float* w2, int w2_stride, const float* h,
                          const float* gh, const float* gz, const float* gz2, float* dw, float* dw2,
                          int dw_shards, int64_t dw_shard_stride, uint32_t accumulate_mask,
                          int64_t n_elem, void* stream);

/* ---- K6 / K7: channel-concat (+ residual) + LayerNorm (+ ReLU) -------------------------
 * x = cat(srcs[0..n_src), dim=1) (+ resid if non-NULL; n_src must be 1 then);
 * out = LayerNorm_[n_src*C, L](x; ln_w, ln_b) (eps 1e-5, biased variance), optional ReLU.
 * stats[s*2+{0,1}] = mean, rstd of sample s (saved for backward).
 * K7 = FusionCell.forward tail model_search.py:63-67 (n_src = multiplier, relu = 1);
 * K6 = NodeCell.forward tail node_search.py:67-68 (n_src = 1, resid = x, relu = 0). */
int bmnas_cat_ln_fwd(const float* const* srcs, int n_src, const float* resid, const float* ln_w,
                     const float* ln_b, float* out, float* stats, int b, int C, int L, int relu,
                     float* out_sums, void* stream);
/* g: gradient of `out`.  dsrcs[q] (NULL to skip) / dresid (NULL to skip) receive the input
 * gradient (acc bits: bit q for dsrcs[q], bit 31 for dresid); dln_w / dln_b += (atomic per
 * sample; pass NULL and use bmnas_ln_affine_bwd, which needs 16x fewer atomics).
 * scrub (nullable): scrub_n floats (multiple of 4) that the launch also zero-fills — the caller's
 * gradient-accumulation arena, cleared without a memset launch of its own. */
int bmnas_cat_ln_bwd(const float* g, const float* const* srcs, int n_src, const float* resid,
                     const float* ln_w, const float* ln_b, const float* stats,
                     float* const* dsrcs, float* dresid, uint32_t accumulate_mask,
                     float* dln_w, float* dln_b, int b, int C, int L, int relu, float* scrub,
                     int64_t scrub_n, void* stream);

/* LayerNorm affine gradients (a reduction over samples, kept out of the per-sample kernels):
 * dln_w[e] += sum_s gy*x_hat, dln_b[e] += sum_s gy with gy = g * (*gscale) * relu-mask.
 * prenorm != 0: srcs[0] already holds x_hat (the attention kernel saves it). */
int bmnas_ln_affine_bwd(const float* g, const float* gscale, const float* const* srcs, int n_src,
                        const float* resid, const float* ln_w, const float* ln_b,
                        const float* stats, float* dln_w, float* dln_b, int b, int C, int L,
                        int relu, int prenorm, void* stream);

/* The same for up to 8 LayerNorms at once (all with batch b and length L): arrays of n_prob
 * entries, C[i] = channels per source of problem i.  Used at the end of a backward pass: the
 * affine gradients feed nothing downstream, so five ~4.5 us launches collapse into one. */
int bmnas_ln_affine_bwd_multi(int n_prob, const float* const* g, const float* const* gscale,
                              const float* const* const* srcs, const int* n_src,
                              const float* const* resid, const float* const* ln_w,
                              const float* const* ln_b, const float* const* stats,
                              float* const* dln_w, float* const* dln_b, int b, const int* C, int L,
                              const int* relu, const int* prenorm, void* stream);

/* ---- K3: scaled-dot attention + dropout + LayerNorm -----------------------------------
 * ScaledDotAttn.forward node_operations.py:92-108: q = x^T, k = y, v = y^T,
 * scores = q@k / sqrt(C), softmax(-1), out = (attn@v)^T, Dropout, LayerNorm([C, L]).
 * One 4-wave workgroup per 16 rows (= 16/L samples), channels split over the waves; QK^T and
 * AV on v_mfma_f32_16x16x4_f32, row softmax by in-lane + cross-lane (xor 16/32) reductions.
 * xhat (b, C, L) receives the normalised pre-affine output (saved for backward);
 * stats as in cat_ln.  C <= 512. */
int bmnas_sdpa_ln_fwd(const float* x, const float* y, const float* ln_w, const float* ln_b,
                      float* out, float* xhat, float* stats, int b, int C, int L,
                      bmnas_dropout_t drop, void* stream);
/* g: gradient of out, multiplied in-kernel by *gscale if gscale != NULL (the gamma weight
 * of the mixed op).  dx / dy (=|+=, bit0 / bit1 of accumulate_mask); if dy == NULL the
 * y-gradient is added into dx (search mode, x is y).  The LayerNorm affine gradients come
 * from bmnas_ln_affine_bwd(g, gscale, {xhat}, prenorm = 1). */
int bmnas_sdpa_ln_bwd(const float* g, const float* gscale, const float* x, const float* y,
                      const float* ln_w, const float* xhat, const float* stats, float* dx,
                      float* dy, uint32_t accumulate_mask, int b, int C, int L,
                      bmnas_dropout_t drop, void* stream);

/* ---- K4 / K5 / out_conv: channel-concat + 1x1 Conv1d as an fp32-MFMA GEMM ----------------
 * U[s, m, l] = bias[m] + sum_k W[m*ldw + k] * cat(srcs)[s, k, l],  k < n_src*C_src, m < M.
 * Replaces torch.cat + nn.Conv1d(k=1) at node_operations.py:32-33, :51-52 and
 * node_search.py:59-61 (several convs sharing the same input may be stacked along M).
 * If part != NULL (train-mode BatchNorm follows): per-channel partial batch statistics
 * part[(m*P + p)*2 + {0,1}] = (sum, M2 about the partial's own mean) over the p-th block
 * of 16 (sample,l) columns; p < P = bmnas_conv1x1_num_partials(b, L). */
int bmnas_conv1x1_num_partials(int b, int L);
/* fold_cols > 0: the weight actually applied is W[m, k] + W[m, k + fold_cols] — the conv of
 * cat[z, z] (search mode, NodeMixedOp(z, z) at node_search.py:55) with n_src = 1.
 * stat_shards > 0 selects the other form of batch statistics: part is then a ZERO-FILLED buffer of
 * stat_shards * M * 2 floats into which the launch adds, with fp32 atomics, the per-channel sums of
 * d = U - bias and of d^2 (shard = column block % stat_shards).  The kernel that applies the
 * BatchNorm finalises them itself (bmnas_bn_fin_t below): no bmnas_bn_finalize launch. */
int bmnas_conv1x1_fwd(const float* const* srcs, int n_src, int C_src, const float* W, int ldw,
                      int fold_cols, const float* bias, float* U, float* part, int stat_shards, int b,
                      int L, int M, void* stream);
/* dsrcs[q][s, c, l] (=|+=) sum_m Weff[m, q*C_src + c] * dU[s, m, l]   (dsrcs[q] NULL: skip;
 * Weff as above) */
int bmnas_conv1x1_bwd_data(const float* dU, const float* W, int ldw, int fold_cols,
                           float* const* dsrcs, int n_src, int C_src, uint32_t accumulate_mask,
                           int b, int L, int M, void* stream);
/* bmnas_conv1x1_fwd and bmnas_sdpa_ln_fwd in ONE launch: the attention branch and the stacked
 * LinearGLU/ConcatFC conv of a NodeMixedOp (node_operations.py:118-120) read the same input and are
 * independent; K3's b*L/16 workgroups leave half the CUs idle, the GEMM tiles fill them.
 * Arguments = those of the two functions (same b, L). */
int bmnas_conv1x1_fwd_sdpa(const float* const* srcs, int n_src, int C_src, const float* W, int ldw,
                           int fold_cols, const float* bias, float* U, float* part, int stat_shards,
                           int b, int L, int M, const float* x, const float* y, const float* ln_w,
                           const float* ln_b, float* out, float* xhat, float* stats, int C,
                           bmnas_dropout_t drop, void* stream);
/* bmnas_conv1x1_bwd_data + bmnas_sdpa_ln_bwd + bmnas_conv1x1_bwd_weight in ONE launch: every
 * contraction of a NodeMixedOp's backward (search mode).  wsrcs: the conv's forward inputs
 * (n_src x (b, C_src, L)) for the weight gradient; dW / ldw_grad / dbias / dup_cols as in
 * bmnas_conv1x1_bwd_weight.  The three block classes run concurrently, so dx / dy must not be among dsrcs: the
 * attention gradient goes to its own buffer and the consumer adds the parts (g2 of bmnas_mixsum_bwd / gz2 of
 * bmnas_mixsum_pair_bwd).  Shapes outside the merged
 * kernels (M != 3C, C > 256) run as the three separate launches. 
 * bn_U != NULL folds the BatchNorm input gradient into the launch: dU then holds dV (the gradient
 * w.r.t. the BatchNorm OUTPUT, bn_grad already reduced) and the tile kernels form
 * scale * (dV - bn_grad[M+m]/N - u_hat * bn_grad[m]/N) (bmnas_bn_bwd_apply) while staging their operands;
 * shapes served by the other kernel families get the same result from a bmnas_bn_bwd_apply launch
 * issued first (dU is then overwritten in place). */
int bmnas_conv1x1_bwd_all_sdpa(const float* dU, const float* W, int ldw, int fold_cols,
                               float* const* dsrcs, int n_src, int C_src, uint32_t accumulate_mask,
                               int b, int L, int M, const float* const* wsrcs, float* dW,
                               int ldw_grad, float* dbias, int dup_cols, const float* g,
                               const float* gscale, const float* x, const float* y,
                               const float* ln_w, const float* xhat, const float* stats, float* dx,
                               float* dy, uint32_t sdpa_accumulate_mask, int C, bmnas_dropout_t drop,
                               const float* bn_U, const float* bn_chan, const float* bn_grad,
                               int bn_training, void* stream);
/* The backward of a conv + BatchNorm with no attention branch beside it (NodeCell's out_conv + bn,
 * reference models/search/darts/node_search.py:63-66): bmnas_bn_bwd_apply (when bn_U != NULL) +
 * bmnas_conv1x1_bwd_data + bmnas_conv1x1_bwd_weight.  Small grids (where the data gradient would take
 * the 1x1-tile split-K kernel) run as ONE launch with the BatchNorm input gradient applied while the
 * operands are staged; grids served by the pipelined tile kernel, and calls that want no data gradient
 * (every dsrcs[q] NULL: frozen backbones in front of the reshape layers), run as two / one launch(es) with
 * the same fold (dU is left untouched in all these cases); the remaining shapes run as the three launches,
 * dU overwritten in place by the first.  Arguments as in bmnas_conv1x1_bwd_all_sdpa. */
int bmnas_conv1x1_bwd_all(const float* dU, const float* W, int ldw, int fold_cols, float* const* dsrcs,
                          int n_src, int C_src, uint32_t accumulate_mask, int b, int L, int M,
                          const float* const* wsrcs, float* dW, int ldw_grad, float* dbias,
                          int dup_cols, const float* bn_U, const float* bn_chan, const float* bn_grad,
                          int bn_training, void* stream);
/* bmnas_conv1x1_bwd_all with the backward of the NodeMixedOp behind one of its sources as the epilogue of that
 * source's data-gradient tiles (small grids only: bmnas_conv1x1_bwd_all_mix_ok; BMNAS_E_LIMIT otherwise).
 * NodeCell's out_conv reads cat(states[-node_multiplier:]) (reference node_search.py:59-61); source `q` is the
 * output of the last inner step's NodeMixedOp (node_operations.py:118-120, x is y) and feeds nothing else, so its
 * data gradient — still written to dsrcs[q], the attention backward reads it — is the complete gradient g of
 * that output: the tile applies bmnas_node_mix_bwd(g, x, x, p1, U, chan, gamma, ...) to its elements (dx (=|+=)
 * 2 gamma[0] g, dV, bn_grad, dgamma shards as there).  One launch (~5.6 us at 6-8 samples per GPU) less per
 * cell step. */
typedef struct {
  const float* U;
  const float* chan;
  const float* x;
  const float* p1;
  const float* gamma;
  float* dgamma;
  int dgamma_shards;
  int64_t dgamma_shard_stride;
  float* dx;
  int accumulate_dx;
  float* dV;
  float* bn_grad;
  int q;
  bmnas_dropout_t drop_glu, drop_fc;
} bmnas_mix_ep_t;
int bmnas_conv1x1_bwd_all_mix_ok(int b, int L, int M, int n_src, int C_src);
int bmnas_conv1x1_bwd_all_mix(const float* dU, const float* W, int ldw, int fold_cols, float* const* dsrcs,
                              int n_src, int C_src, uint32_t accumulate_mask, int b, int L, int M,
                              const float* const* wsrcs, float* dW, int ldw_grad, float* dbias, int dup_cols,
                              const float* bn_U, const float* bn_chan, const float* bn_grad, int bn_training,
                              const bmnas_mix_ep_t* mix, void* stream);
/* dW[m*ldw + k] += sum_{s,l} dU[s,m,l] * cat(srcs)[s,k,l];  dbias[m] += sum_{s,l} dU[s,m,l]
 * (atomic adds: caller zeroes; dbias may be NULL).  If dup_cols > 0 the same value is also
 * added at column k + dup_cols (folded x-is-y weights, see bmnas_fold_weight). */
int bmnas_conv1x1_bwd_weight(const float* dU, const float* const* srcs, int n_src, int C_src,
                             float* dW, int ldw, float* dbias, int dup_cols, int b, int L, int M,
                             void* stream);
/* Weff[m*C + c] = W[m*2C + c] + W[m*2C + C + c]: the conv applied to cat[z, z] (search mode,
 * NodeMixedOp(z, z) at node_search.py:55) equals Weff applied to z. */
int bmnas_fold_weight(const float* W, float* Weff, int M, int C, void* stream);
/* Diagnostics (tests/test_dispatch_gpu.py): how many calls each GEMM kernel family has served since
 * the last reset — the conv entry points choose a family by shape (pipelined LDS tiles, split-K,
 * whole-K LDS, direct; merged with the attention / weight-gradient workgroups or not).  Copies
 * min(n, families) host counters to out, optionally resets them, returns the family count;
 * bmnas_conv_family_name(i) names family i.  Host-side bookkeeping only: no kernel reads it. */
int bmnas_conv_family_calls(long* out, int n, int reset);
const char* bmnas_conv_family_name(int i);

/* ---- BatchNorm1d bookkeeping ------------------------------------------------------------
 * Combines the GEMM's partial statistics (Chan's parallel variance), or uses the running
 * statistics in eval mode, into per-channel mean[m], rstd[m] and the fused affine
 * scale[m] = bn_w*rstd, shift[m] = bn_b - mean*scale; in training mode also updates
 * running_mean / running_var (momentum 0.1, unbiased variance) and the n_nbt consecutive
 * int64 num_batches_tracked counters (several stacked BatchNorms), like nn.BatchNorm1d (node_operations.py:26,34 / :45,53, node_search.py:40,62).
 * chan[4*M] = mean | rstd | scale | shift. */
int bmnas_bn_finalize(const float* part, int n_part, int b, int L, int M, const float* bn_w,
                      const float* bn_b, float* running_mean, float* running_var,
                      int64_t* num_batches_tracked, int n_nbt, int training, float* chan,
                      void* stream);

/* BatchNorm finalisation INSIDE the kernel that applies it (bmnas_node_mix_fwd, bmnas_node_mix_ln_fwd,
 * bmnas_bn_relu_fwd) instead of a bmnas_bn_finalize launch in front of it.  on = 0: `chan` already
 * holds mean | rstd | scale | shift.  on = 1: every workgroup derives scale / shift in LDS —
 * training: from `stat` (the sums a bmnas_conv1x1_fwd with stat_shards = shards accumulated; conv_bias
 * is the shift they were taken about), eval: from the running statistics — and workgroup 0 writes
 * `chan` (now an OUTPUT, for the backward kernels) and, in training mode, updates running_mean /
 * running_var / the n_nbt counters like nn.BatchNorm1d. */
typedef struct {
  const float* stat;
  const float* conv_bias;
  const float* bn_w;
  const float* bn_b;
  float* running_mean;
  float* running_var;
  int64_t* num_batches_tracked;
  int shards, n_nbt, training, on;
} bmnas_bn_fin_t;

/* ---- K2: the gamma-weighted NodeMixedOp combine -------------------------------------------
 * NodeMixedOp.forward node_operations.py:118-120 over [Sum, ScaleDotAttn, LinearGLU, ConcatFC]:
 *   s = g0*(x+y) + g1*p1 + g2*drop(glu(BN(U[:, 0:2C]))) + g3*drop(relu(BN(U[:, 2C:3C])))
 * gamma: 4 device floats (softmaxed row).  U: (b, 3C, L) stacked conv output [GLU | ConcatFC],
 * chan: its bn_finalize output (M = 3C).  p1 = attention branch output. */
int bmnas_node_mix_fwd(const float* x, const float* y, const float* p1, const float* U, float* chan,
                       bmnas_bn_fin_t fin, const float* gamma, float* out, int b, int C, int L,
                       bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, void* stream);
/* K2 + K6 in one launch (node_multiplier == 1, last inner step; node_search.py:55,67-68):
 * pre = NodeMixedOp(...) + resid (saved for backward), out = LayerNorm_[C, L](pre), stats as in
 * cat_ln.  The backward is bmnas_cat_ln_bwd(srcs = {pre}, resid = NULL) with its input gradient
 * routed to both the mix and the residual, then bmnas_node_mix_bwd. */
int bmnas_node_mix_ln_fwd(const float* x, const float* y, const float* p1, const float* U,
                          float* chan, bmnas_bn_fin_t fin, const float* gamma, const float* resid,
                          const float* ln_w, const float* ln_b, float* pre, float* out, float* stats,
                          int b, int C, int L, bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc,
                          float* out_sums, void* stream);
/* K2 as the PRODUCER of out_conv's last operand (small grids, node_multiplier != 1; reference
 * node_search.py:55 `out = node_op(x, y, gammas[i])` of the last inner step, then :59-61
 * `out = out_conv(torch.cat(states[-node_multiplier:], dim=1))`): ONE launch instead of bmnas_node_mix_fwd +
 * bmnas_conv1x1_fwd.  mix_out (b, C, L) receives s exactly as bmnas_node_mix_fwd would write it (the backward
 * pass and later states read it); V (b, C, L) = W cat(srcs[0..n_src), s) + bias with W (C, ldw) row-major,
 * the first (n_src + 1) * C columns used; `stat` (stat_shards, C, 2): zero-filled batch sums of V - bias as in
 * bmnas_conv1x1_fwd (NULL with stat_shards 0: no statistics).  x, y, p1, U, chan, fin, gamma, drop_*: as
 * bmnas_node_mix_fwd.  bmnas_node_mix_conv_fwd_ok: the shapes it takes (n_src <= 3, C % 64 == 0, C <= 256,
 * (b L / 16) (C / 16) <= 256 workgroups: every output tile recomputes its 16 columns of s); BMNAS_E_LIMIT
 * otherwise. */
int bmnas_node_mix_conv_fwd_ok(int b, int C, int L, int n_src);
int bmnas_node_mix_conv_fwd(const float* x, const float* y, const float* p1, const float* U, float* chan,
                            bmnas_bn_fin_t fin, const float* gamma, float* mix_out, bmnas_dropout_t drop_glu,
                            bmnas_dropout_t drop_fc, const float* const* srcs, int n_src, const float* W, int ldw,
                            const float* bias, float* V, float* stat, int stat_shards, int b, int C, int L,
                            void* stream);
/* Backward, phase A (elementwise + reductions):  g = grad of s.
 *   dgamma[shard*dgamma_shard_stride + q] += <g, p_q> (shards as in bmnas_mixsum_bwd);
 *   dx / dy (=|+=) g0*g (dy NULL: both into dx);
 *   dV[s, m, l] = gradient w.r.t. the BatchNorm OUTPUT (b, 3C, L);
 *   bn_grad[m] += sum dV*u_hat (= dBN.weight), bn_grad[3C + m] += sum dV (= dBN.bias). */
int bmnas_node_mix_bwd(const float* g, const float* x, const float* y, const float* p1,
                       const float* U, const float* chan, const float* gamma, float* dgamma,
                       int dgamma_shards, int64_t dgamma_shard_stride, float* dx, float* dy,
                       uint32_t accumulate_mask, float* dV, float* bn_grad, int b, int C, int L,
                       bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, void* stream);

/* K6 backward + K2 backward in one launch (node_multiplier == 1; reference node_search.py:55,67-68 run
 * backwards): g = grad of out = LayerNorm_[C, L](pre), pre / stats as saved by bmnas_node_mix_ln_fwd.
 * g_in (nullable) receives the LayerNorm input gradient (the attention backward reads it), dresid (=|+= by
 * accumulate_resid; nullable) the same values as the residual's gradient; everything else as
 * bmnas_node_mix_bwd with that gradient as its g.  Two workgroups per sample; one BatchNorm atomic pair per
 * channel per sample, hence b <= 128 and C*L <= 8192 (bmnas_node_mix_ln_bwd_ok; BMNAS_E_LIMIT otherwise).
 * Replaces bmnas_cat_ln_bwd + bmnas_node_mix_bwd. */
int bmnas_node_mix_ln_bwd_ok(int b, int C, int L);
int bmnas_node_mix_ln_bwd(const float* g, const float* pre, const float* ln_w, const float* stats,
                          float* g_in, float* dresid, int accumulate_resid, const float* x, const float* y,
                          const float* p1, const float* U, const float* chan, const float* gamma,
                          float* dgamma, int dgamma_shards, int64_t dgamma_shard_stride, float* dx, float* dy,
                          uint32_t accumulate_mask, float* dV, float* bn_grad, int b, int C, int L,
                          bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, void* stream);

/* K2 with the NEXT inner step's mixed sum riding along (NodeCell.forward, reference
 * models/search/darts/node_search.py:52-57: step t+1 starts with z = sum_j beta_j states[j], and
 * states[-1] is the s this launch produces):
 *   z_next = sum_{j < n_prev} w[j*w_stride] * prev[j] + w[n_prev*w_stride] * s      (n_prev <= 5)
 * one launch fewer per inner step; n_prev == 0: exactly bmnas_node_mix_fwd. */
int bmnas_node_mix_fwd_next(const float* x, const float* y, const float* p1, const float* U, float* chan,
                            bmnas_bn_fin_t fin, const float* gamma, float* out, int b, int C, int L,
                            bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, const float* const* prev,
                            int n_prev, const float* w, int w_stride, float* z_next, void* stream);
/* Backward of the same: bmnas_mixsum_bwd of the next step's sum, then bmnas_node_mix_bwd, one launch.
 * G = gz + gz2 (gz2 nullable) is the gradient of z_next; dprev[j] (=|+= by bit j of
 * prev_accumulate_mask, nullable, may alias each other: in-order read-modify-write) gets w_j * G;
 * dw[shard*dw_shard_stride + j*w_stride] += <G, prev_j> (j < n_prev) and <G, s> (j = n_prev);
 * g (nullable) is what other consumers of s accumulated so far, g_out (may be g) receives the complete
 * gradient g + w_n * G, which this launch then uses as bmnas_node_mix_bwd's g (and the attention
 * backward launched after it must read g_out). */
int bmnas_node_mix_bwd_next(const float* g, const float* x, const float* y, const float* p1, const float* U,
                            const float* chan, const float* gamma, float* dgamma, int dgamma_shards,
                            int64_t dgamma_shard_stride, float* dx, float* dy, uint32_t accumulate_mask,
                            float* dV, float* bn_grad, int b, int C, int L, bmnas_dropout_t drop_glu,
                            bmnas_dropout_t drop_fc, const float* const* prev, float* const* dprev,
                            int n_prev, uint32_t prev_accumulate_mask, const float* w, int w_stride,
                            float* dw, int dw_shards, int64_t dw_shard_stride, const float* s,
                            const float* gz, const float* gz2, float* g_out, void* stream);

/* ---- standalone LinearGLU tail (Found nets): out = drop(glu(BN(U))), U (b, 2C, L) --------
 * node_operations.py:34-38.  fin (as bmnas_bn_relu_fwd): when on, the BatchNorm batch sums that the conv's epilogue
 * accumulated are finalised by this launch (every workgroup forms scale / shift in LDS, workgroup 0 writes `chan` and the
 * running statistics) — no bmnas_bn_finalize launch in front; off: `chan` is read as given.
 * Backward phase A like bmnas_node_mix_bwd (M = 2C). */
int bmnas_bn_glu_fwd(const float* U, float* chan, bmnas_bn_fin_t fin, float* out, int b, int C, int L,
                     bmnas_dropout_t drop, void* stream);
int bmnas_bn_glu_bwd(const float* g, const float* U, const float* chan, float* dV, float* bn_grad,
                     int b, int C, int L, bmnas_dropout_t drop, void* stream);

/* ---- BN + ReLU + dropout: ConcatFC tail (node_operations.py:53-55) and the NodeCell
 * out_conv tail (node_search.py:60-64) ---------------------------------------------------- */
int bmnas_bn_relu_fwd(const float* U, float* chan, bmnas_bn_fin_t fin, float* out, int b, int M, int L,
                      bmnas_dropout_t drop, void* stream);
int bmnas_bn_relu_bwd(const float* g, const float* U, const float* chan, float* dV, float* bn_grad,
                      int b, int M, int L, bmnas_dropout_t drop, void* stream);

/* ---- the N reshape layers in front of the fusion cell as ONE launch per stage ----------
 * ReshapeInputLayer / ReshapeInputLayer_MMIMDB (aux_models.py:51-76, 87-115; built per modality at
 * mmimdb_darts_searchable.py:84-90, ntu_darts_searchable.py:102-108, ego_darts_searchable.py:102-108) are N
 * independent  Conv1d(C_in_i -> C, k = 1) -> BatchNorm1d(C) -> ReLU -> Dropout  stacks on N pooled feature
 * tensors (b, C_in_i, L) of one batch: the same four kernels N times over.  These entry points take all N
 * problems (n <= BMNAS_MAX_GROUP) and put every layer's tiles in ONE grid:
 *   forward :  bmnas_conv1x1_fwd_group (U_i = W_i x_i + bias_i, BatchNorm batch sums into stat_i)
 *              bmnas_bn_relu_fwd_group (x'_i = dropout(relu(bn(U_i))), statistics finalised in the launch)
 *   backward:  bmnas_bn_relu_bwd_group (dV_i, BatchNorm affine gradients)
 *              bmnas_conv1x1_bwd_group (dW_i, dbias_i and — dsrc_i != NULL — the gradient of the pooled
 *                                       input, with the BatchNorm input gradient applied while the operands
 *                                       are staged, as bmnas_conv1x1_bwd_all does for one conv)
 * Same arithmetic, operation by operation, as the single-conv entry points they stand for (the tile bodies are
 * shared); b, L, M = C and the BatchNorm mode are common to the group, C_in_i % 16 == 0, M <= 384. */
#define BMNAS_MAX_GROUP 8
typedef struct {
  const float* src;      /* (b, C_in, L) */
  const float* W;        /* (M, ldw) row-major, first C_in columns used */
  const float* bias;     /* (M), nullable */
  float* U;              /* (b, M, L) */
  float* stat;           /* stat_shards x M x 2 zero-filled sums (bmnas_conv1x1_fwd), or NULL with stat_shards 0 */
  int C_in, ldw;
} bmnas_conv_fwd_prob_t;
typedef struct {
  const float* U;
  float* chan;
  float* out;
  bmnas_bn_fin_t fin;
  bmnas_dropout_t drop;
} bmnas_bn_relu_fwd_prob_t;
typedef struct {
  const float* g;
  const float* U;
  const float* chan;
  float* dV;
  float* bn_grad;        /* [dBN.weight (M) | dBN.bias (M)], += (atomics: caller zeroes) */
  bmnas_dropout_t drop;
} bmnas_bn_relu_bwd_prob_t;
typedef struct {
  const float* dV;       /* (b, M, L): gradient w.r.t. the BatchNorm OUTPUT when bn_U != NULL, else dU */
  const float* W;
  const float* src;      /* the layer's input (b, C_in, L) */
  float* dsrc;           /* its gradient, NULL = not needed */
  float* dW;             /* (M, ldw_grad), += (atomics) */
  float* dbias;          /* (M), +=, nullable */
  const float* bn_U;     /* raw conv output, bn_chan, bn_grad: as bmnas_conv1x1_bwd_all */
  const float* bn_chan;
  const float* bn_grad;
  int C_in, ldw, ldw_grad, accumulate;   /* accumulate: dsrc += instead of = */
} bmnas_conv_bwd_prob_t;
/* The pooling in front of those convs, for the whole group in one launch per direction: AdaptiveMaxPool2d of
 * x_i viewed as (b, C_i, H_i, W_i) to (oh_i, ow_i), written as (b, C_i, oh_i * ow_i) — the GEMM's operand layout.
 * ReshapeInputLayer (aux_models.py:62-70): view (b, C_in, T, R), pool to (L, 1), then F.interpolate(., L) which is
 * an identity at that size; ReshapeInputLayer_MMIMDB (:101-108): view (b, C_in, H, W), pool to (sqrt L, sqrt L).
 * torch's rule: window of output row i = [floor(i H / oh), ceil((i + 1) H / oh)), first maximum in row-major order,
 * NaN counts as a maximum.  idx (int32, flat h * W + w, nullable in forward) feeds the backward, which writes EVERY
 * element of dx (gather form: no zero-fill, no atomics). */
typedef struct {
  const float* x;        /* forward: (b, C, H, W) */
  float* out;            /* forward: (b, C, oh * ow) */
  int* idx;              /* forward: written if not NULL; backward: read */
  const float* g;        /* backward: gradient of out */
  float* dx;             /* backward: (b, C, H, W), every element written */
  int C, H, W, oh, ow;
} bmnas_pool_prob_t;
int bmnas_adaptive_maxpool_fwd_group(const bmnas_pool_prob_t* probs, int n, int b, void* stream);
int bmnas_adaptive_maxpool_bwd_group(const bmnas_pool_prob_t* probs, int n, int b, void* stream);
int bmnas_conv1x1_group_ok(int n, const int* C_in, int b, int L, int M);
int bmnas_conv1x1_fwd_group(const bmnas_conv_fwd_prob_t* probs, int n, int stat_shards, int b, int L, int M,
                            void* stream);
int bmnas_bn_relu_fwd_group(const bmnas_bn_relu_fwd_prob_t* probs, int n, int b, int M, int L, void* stream);
int bmnas_bn_relu_bwd_group(const bmnas_bn_relu_bwd_prob_t* probs, int n, int b, int M, int L, void* stream);
int bmnas_conv1x1_bwd_group(const bmnas_conv_bwd_prob_t* probs, int n, int bn_training, int b, int L, int M,
                            void* stream);

/* NodeCell's whole tail for node_multiplier != 1 (reference node_search.py:64-69) as one launch per
 * direction:  o = dropout(relu(bn(U)));  out = LayerNorm_[C,L](o + resid).
 * fwd: U (b, C, L) raw out_conv output; chan / fin as in bmnas_bn_relu_fwd; o and out (b, C, L) written;
 * stats (b, 2) = per-sample (mean, rstd) of the LayerNorm; out_sums (b, 2) nullable = per-sample
 * (sum, sum of squares) of out (for bmnas_head_fwd).
 * bwd: g = grad of out.  dresid (=|+= by accumulate_resid; nullable) gets the LayerNorm input gradient,
 * dV (b, C, L) the gradient w.r.t. the BatchNorm output, bn_grad (2C, caller-zeroed) += (sum dV * u_hat |
 * sum dV) by atomics.  Replaces bmnas_cat_ln_bwd + bmnas_bn_relu_bwd. */
int bmnas_bn_relu_ln_fwd(const float* U, float* chan, bmnas_bn_fin_t fin, const float* resid,
                         const float* ln_w, const float* ln_b, float* o, float* out, float* stats, int b,
                         int C, int L, bmnas_dropout_t drop, float* out_sums, void* stream);
/* bmnas_bn_relu_ln_fwd with the NEXT cell step's K1 pair sum in the same launch (small batches:
 * bmnas_bn_relu_ln_fwd_pair_ok — b <= 128, C*L <= 1024, n_prev <= 15; BMNAS_E_LIMIT otherwise).  FusionCell.forward,
 * reference model_search.py:58 (+ node_search.py:54 at t = 0): the next step's inputs are n_prev earlier
 * states xs[j] (b, C, L) and the node output `out` this launch produces:
 *   h = sum_{j < n_prev} w[j*w_stride] xs[j] + w[n_prev*w_stride] out,   z = (w2[0] + w2[w2_stride]) h
 * (w, w2: SOFTMAXED weights, as bmnas_mixsum_pair_fwd takes them; same arithmetic order).  n_prev == 0: exactly
 * bmnas_bn_relu_ln_fwd. */
int bmnas_bn_relu_ln_fwd_pair_ok(int b, int C, int L, int n_prev);
int bmnas_bn_relu_ln_fwd_pair(const float* U, float* chan, bmnas_bn_fin_t fin, const float* resid,
                              const float* ln_w, const float* ln_b, float* o, float* out, float* stats, int b,
                              int C, int L, bmnas_dropout_t drop, float* out_sums, const float* const* xs,
                              int n_prev, const float* w, int w_stride, const float* w2, int w2_stride, float* h,
                              float* z, void* stream);
int bmnas_bn_relu_ln_bwd(const float* g, const float* o, const float* resid, const float* ln_w,
                         const float* stats, const float* U, const float* chan, float* dV, float* bn_grad,
                         float* dresid, int accumulate_resid, int b, int C, int L, bmnas_dropout_t drop,
                         void* stream);
/* bmnas_bn_relu_ln_bwd with the backward of the NEXT cell step's K1 pair sum first, in the same launch (the
 * counterpart of bmnas_bn_relu_ln_fwd_pair; same limits).  That sum's inputs are xs[0 .. n_prev-1] and this
 * node's forward output `out`:  G = gh + (w2[0] + w2[w2_stride]) (gz + gz2)  (gh, gz2 nullable);
 *   dxs[j] (=|+= by bit j of accumulate_mask; nullable; pairwise distinct) w[j*w_stride] G,
 *   dw[shard + j*w_stride] += <G, xs[j]> (j < n_prev), dw[shard + n_prev*w_stride] += <G, out>,
 *   dw2[shard], dw2[shard + w2_stride] += <gz + gz2, h>   (shards as in bmnas_mixsum_pair_bwd);
 * the node-output gradient the LayerNorm backward starts from is  g + w[n_prev*w_stride] G  (g nullable: what
 * other consumers accumulated), also written to g_full (b, C, L).  Replaces bmnas_mixsum_pair_bwd +
 * bmnas_bn_relu_ln_bwd (reference model_search.py:58 and node_search.py:64-69, backwards). */
int bmnas_bn_relu_ln_bwd_pair(const float* g, const float* o, const float* resid, const float* ln_w,
                              const float* stats, const float* U, const float* chan, float* dV, float* bn_grad,
                              float* dresid, int accumulate_resid, int b, int C, int L, bmnas_dropout_t drop,
                              const float* const* xs, float* const* dxs, int n_prev, uint32_t accumulate_mask,
                              const float* out, const float* w, int w_stride, const float* w2, int w2_stride,
                              const float* h, const float* gh, const float* gz, const float* gz2, float* dw,
                              float* dw2, int dw_shards, int64_t dw_shard_stride, float* g_full, void* stream);

/* Backward, phase B: BatchNorm input gradient, in place on dV (b, M, L):
 *   training: dU = scale*(dV - bn_grad[M+m]/N - u_hat*bn_grad[m]/N), N = b*L;  eval: dU = scale*dV. */
int bmnas_bn_bwd_apply(float* dV, const float* U, const float* chan, const float* bn_grad, int b,
                       int M, int L, int training, void* stream);

/* ---- architecture-parameter softmax (model_search.py:95, node_search.py:102-103) -----------
 * Row softmax of `rows` rows of `cols` (2 or 4) logits; backward:
 * dlogit[r,:] (=) w[r,:] * (dw[r,:] - sum_p w[r,p]*dw[r,p]). */
int bmnas_arch_softmax_fwd(const float* logits, float* w, int rows, int cols, void* stream);
int bmnas_arch_softmax_bwd(const float* w, const float* dw, float* dlogits, int rows, int cols,
                           void* stream);
/* All n (<= 16) architecture tensors in ONE launch.  backward == 0: out[t] = softmax(a[t]);
 * backward != 0: a[t] = softmax weights, dw[t] = their gradient (summed over n_shards copies
 * spaced shard_stride floats apart), out[t] = dlogits. */
int bmnas_arch_softmax_multi(const float* const* a, const float* const* dw, float* const* out,
                             const int* rows, const int* cols, int n, int backward, int n_shards,
                             int64_t shard_stride, void* stream);
/* The two launches that end a FusionCell's backward, as one: bmnas_ln_affine_bwd_multi (first 16
 * arguments) and bmnas_arch_softmax_multi with backward = 1 (arch_w = the softmaxed weights, arch_dw
 * their gradient shards, arch_out = the gradients of the logits).  Independent work sharing a grid. */
int bmnas_backward_epilogue(int n_prob, const float* const* g, const float* const* gscale,
                            const float* const* const* srcs, const int* n_src,
                            const float* const* resid, const float* const* ln_w,
                            const float* const* ln_b, const float* const* stats,
                            float* const* dln_w, float* const* dln_b, int b, const int* C, int L,
                            const int* relu, const int* prenorm, const float* const* arch_w,
                            const float* const* arch_dw, float* const* arch_out, const int* arch_rows,
                            const int* arch_cols, int n_arch, int n_shards, int64_t shard_stride, int n_sums,
                            const float* const* sum_part, float* const* sum_out, const int* sum_chunks,
                            const int64_t* sum_n, void* stream);
/* The forward prologue of a FusionCell in ONE launch: the n_arch row softmaxes of
 * bmnas_arch_softmax_multi (forward) and, for each of n_fold (<= 8) NodeMixedOps of the cell,
 * Weff[q] (M, C) = W[q][:, :C] + W[q][:, C:] as bmnas_fold_weight does (W[q] is (M, 2C)).
 * step_counter / step_span (both or neither): *step_counter += *step_span, once, before anything
 * else of the step reads the dropout step counter (bmnas_dropout_t.step) — how a hipGraph replay
 * moves on to fresh dropout masks without a launch of its own.
 * scrub (scrub_n floats, % 4 == 0, nullable): zero-filled by the same launch — the forward
 * accumulation buffers of the step (BatchNorm batch sums of bmnas_conv1x1_fwd(stat_shards > 0),
 * the head's logits) instead of a memset launch. */
int bmnas_cell_prologue(const float* const* a, float* const* out, const int* rows, const int* cols,
                        int n_arch, const float* const* W, float* const* Weff, int n_fold, int M,
                        int C, uint64_t* step_counter, const uint64_t* step_span, float* scrub,
                        int64_t scrub_n, void* stream);

/* bmnas_cell_prologue and the first data kernel of the cell in ONE launch: besides the prologue's jobs
 * the grid streams the first step's mixed-edge pair sum (bmnas_mixsum_pair_fwd)
 *   h = sum_j softmax(alpha_logits)[j, 1] xs[j],  z = (softmax(beta_logits)[0, 1] + softmax(beta_logits)[1, 1]) h
 * (model_search.py:58 + node_search.py:54 at t = 0) with the edge weights taken from the raw two-column
 * logits in registers — the prologue's outputs are only needed by the launches after this one.
 * alpha_logits: first of n_in consecutive rows of alphas_edges; beta_logits: rows 0, 1 of the node's betas. */
int bmnas_cell_prologue_pair(const float* const* a, float* const* out, const int* rows, const int* cols,
                             int n_arch, const float* const* W, float* const* Weff, int n_fold, int M,
                             int C, uint64_t* step_counter, const uint64_t* step_span, float* scrub,
                             int64_t scrub_n, const float* const* xs, int n_in, const float* alpha_logits,
                             const float* beta_logits, float* h, float* z, int64_t n_elem, void* stream);

/* ---- the head of a search step: K7 + central_classifier (+ criterion) in two launches ----------
 * Forward (model_search.py:63-67 + mmimdb_darts_searchable.py:114):
 *   logits = relu(LayerNorm_[M*C, L](cat(srcs))).view(b, -1) @ W^T + bias
 * without materialising the LayerNorm output: sums[q] = (b, 2) per-sample (sum, sum of squares) of
 * state q as left by the kernel that produced it (out_sums of bmnas_node_mix_ln_fwd / bmnas_cat_ln_fwd)
 * give the statistics, the normalisation happens in the operand fetch of the GEMM.
 *   hb: ZERO-FILLED [3][b][O] — logits | A | B (A, B: the two extra products the backward's
 *   LayerNorm reductions need, see csrc/head.hip); stats (b, 2) = mean | rstd (output).
 * O <= 128, (C*L) % 16 == 0, n_src <= 4. */
int bmnas_head_fwd(const float* const* srcs, const float* const* sums, int n_src, const float* ln_w,
                   const float* ln_b, const float* W, const float* bias, float* hb, float* stats, int b,
                   int C, int L, int O, void* stream);
/* Backward of the same, down to dsrcs[q] (=|+=, bit q of accumulate_mask; NULL: skip).
 *   mode 0: g = dlogits (b, O);  mode 1: BCEWithLogits(mean) of hb's logits against float labels
 *   (b, O), *loss += the mean loss;  mode 2: CrossEntropy(mean) against int64 labels (b) — the
 *   criterion (mmimdb_darts_searchable.py:22, ntu_darts_searchable.py:25) evaluated in the same launch.
 *   gscale (nullable): device scalar multiplying dlogits.
 * Batch reductions leave as per-sample-chunk partials (16 or 32 samples; n_chunk = bmnas_head_chunks(b)):
 *   part [n_chunk][O + 3][D]: rows 0..O-1 = dW, row O = dln_w, row O+1 = dln_b, row O+2 = dbias in its
 *   first O entries (the rest of that row is never written);
 * sum them with bmnas_backward_epilogue(n_sums ...) or bmnas_sum_chunks.
 * scrub: optional zero-fill side job (the caller's backward accumulation arena). */
int bmnas_head_chunks(int b);
int bmnas_head_bwd(const float* const* srcs, const float* const* sums, float* const* dsrcs, int n_src,
                   uint32_t accumulate_mask, const float* ln_w, const float* ln_b, const float* W,
                   const float* hb, const float* stats, int mode, const float* g, const float* gscale,
                   const void* labels, float* loss, float* part, int b, int C, int L, int O, float* scrub,
                   int64_t scrub_n, void* stream);
/* out[e] = sum_{c < n_chunk} part[c*n + e], e < n (n % 4 == 0) */
int bmnas_sum_chunks(const float* part, float* out, int n_chunk, int64_t n, void* stream);

/* ---- the step node's LayerNorm applied by its consumers (csrc/lazyln.hip, csrc/lazy_ln.hpp) ----------
 * NodeCell with node_multiplier == 1 ends in  out = mix(z);  out += x;  out = LayerNorm_[C,L](out)
 * (reference node_search.py:57-58, 67-68 with node_operations.py:118-120).  bmnas_node_mix_ln_fwd / _bwd do that as one
 * workgroup per sample; the entry points below do it with streaming grids over all compute units:
 *   bmnas_node_mix_pre_fwd       pre = mix + resid (un-normalised) and, per part of 1024 elements of a sample, a
 *                                record of moments centred on the part's own mean: rec (b, P, 8), prm (P, 8) with
 *                                P = bmnas_lazy_ln_parts(C, L) (layouts: csrc/lazy_ln.hpp).  Plain stores.
 *   bmnas_mixsum_pair_fwd_lazy   bmnas_mixsum_pair_fwd whose LAST input (weight w[n_in * w_stride]) is such a node
 *                                output: it combines the records, normalises in registers, and WRITES the node output
 *                                last_out (b, C, L), last->stats (b, 2) = mean | rstd and — last_sums nullable —
 *                                the per-sample (sum, sum of squares) of the output for bmnas_head_fwd.
 *   bmnas_head_fwd_lazy          bmnas_head_fwd whose source lazy_q is given un-normalised (srcs[lazy_q] == lazy->pre,
 *                                sums[lazy_q] ignored): normalised in the operand fetch, lazy->stats written.
 *   bmnas_head_bwd_lazy          bmnas_head_bwd with EVERY source described by lazy[q] (pre, ln_w, ln_b, stats); besides
 *                                dsrcs[q] (the gradient w.r.t. the node OUTPUT) it stores, per (sample, 64-k group),
 *                                the partials of the node LayerNorm backward's two sums  S(gy w), S(gy w xhat)  into
 *                                lnpart[q] (b, C*L/64, 2) (nullable).  (C*L) % 64 == 0.
 *   bmnas_mixsum_pair_bwd_lazy   bmnas_mixsum_pair_bwd whose last n_lazy (1 or 2) inputs are such node outputs (xs holds
 *                                their normalised values): additionally stores, per (sample, part), the partials of the
 *                                same two sums for the gradient piece w_j G it adds to dxs[j], into lnpart[t]
 *                                [(sample * lnpart_stride[t] + part) * 2 + {0, 1}] (lnpart_stride[t] >= P: the K1
 *                                launches of several later steps may share one buffer).  g_full (nullable): G =
 *                                gh + (w2_0 + w2_1)(gz + gz2) is stored as well.
 *   bmnas_node_mix_lnp_bwd       bmnas_node_mix_ln_bwd for any batch size: g = gradient of the node output, m1 / m2
 *                                of the LayerNorm backward from the partials lnp0 (b, n0, 2) and lnp1 (b, n1, 2)
 *                                (either may be absent: n = 0), then the mix backward of bmnas_node_mix_bwd.
 * bmnas_lazy_ln_ok: L in {4, 8, 16} and C*L <= 4096. */
typedef struct {
  const float* pre;    /* (b, C, L) mix + x, before the LayerNorm */
  const float* rec;    /* (b, P, 8) moment records (forward consumers) */
  const float* prm;    /* (P, 8) sums of the affine parameters (forward consumers) */
  const float* ln_w;   /* (C, L) */
  const float* ln_b;
  float* stats;        /* (b, 2) mean | rstd: written by the first forward consumer, read by the backward */
} bmnas_lazy_ln_t;
int bmnas_lazy_ln_ok(int C, int L);
int bmnas_lazy_ln_parts(int C, int L);
int bmnas_node_mix_pre_fwd(const float* x, const float* y, const float* p1, const float* U, float* chan,
                           bmnas_bn_fin_t fin, const float* gamma, const float* resid, const float* ln_w,
                           const float* ln_b, float* pre, float* rec, float* prm, int b, int C, int L,
                           bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc, void* stream);
int bmnas_mixsum_pair_fwd_lazy(const float* const* xs, int n_in, const float* w, int w_stride, const float* w2,
                               int w2_stride, const bmnas_lazy_ln_t* last, float* last_out, float* last_sums,
                               float* out, float* out2, int b, int C, int L, void* stream);
int bmnas_mixsum_pair_bwd_lazy(const float* const* xs, float* const* dxs, int n_in, const float* w, int w_stride,
                               const float* w2, int w2_stride, const float* h, const float* gh, const float* gz,
                               const float* gz2, float* dw, float* dw2, int dw_shards, int64_t dw_shard_stride,
                               uint32_t accumulate_mask, const bmnas_lazy_ln_t* lazy, float* const* lnpart,
                               const int* lnpart_stride, int n_lazy, float* g_full, int b, int C, int L,
                               void* stream);
/* bmnas_mixsum_pair_bwd for the FIRST cell step when the later steps' K1 backward launches stored their G_t instead
 * of read-modify-writing the N cell-input gradients (g_full of bmnas_mixsum_pair_bwd_lazy): every step's mixed sum
 * reads the same cell inputs (reference model_search.py:58), so
 *   dxs[j] (=|+=) w[j*w_stride] G + sum_{t < n_more} w_more[t][j*w_stride] g_more[t]        written ONCE.
 * n_more <= 2; n_more == 0 is bmnas_mixsum_pair_bwd.
 * dw == dw2 == NULL (both or neither; also bmnas_mixsum_pair_bwd and bmnas_mixsum_pair_bwd_lazy): nobody differentiates
 * the edge weights — the weight step of the search loop, whose optimizer holds the network weights only — and the launch
 * neither forms the dot products nor loads the n_in inputs and h, which it reads for nothing else. */
int bmnas_mixsum_pair_bwd_x(const float* const* xs, float* const* dxs, int n_in, const float* w, int w_stride,
                            const float* w2, int w2_stride, const float* h, const float* gh, const float* gz,
                            const float* gz2, float* dw, float* dw2, int dw_shards, int64_t dw_shard_stride,
                            uint32_t accumulate_mask, const float* const* g_more, const float* const* w_more,
                            int n_more, int64_t n_elem, void* stream);
int bmnas_head_fwd_lazy(const float* const* srcs, const float* const* sums, int n_src, int lazy_q,
                        const bmnas_lazy_ln_t* lazy, const float* ln_w, const float* ln_b, const float* W,
                        const float* bias, float* hb, float* stats, int b, int C, int L, int O, float* hb_part,
                        void* stream);
int bmnas_head_bwd_lazy(const bmnas_lazy_ln_t* lazy, float* const* lnpart, float* const* dsrcs, int n_src,
                        uint32_t accumulate_mask, const float* ln_w, const float* ln_b, const float* W,
                        const float* hb, const float* stats, int mode, const float* g, const float* gscale,
                        const void* labels, float* loss, float* part, int b, int C, int L, int O, float* scrub,
                        int64_t scrub_n, float* loss_part, void* stream);
int bmnas_node_mix_lnp_bwd(const float* g, const float* pre, const float* ln_w, const float* stats,
                           const float* lnp0, int n0, const float* lnp1, int n1, float* g_in, float* dresid,
                           int accumulate_resid, const float* x, const float* y, const float* p1, const float* U,
                           const float* chan, const float* gamma, float* dgamma, int dgamma_shards,
                           int64_t dgamma_shard_stride, float* dx, float* dy, uint32_t accumulate_mask, float* dV,
                           float* bn_grad, int b, int C, int L, bmnas_dropout_t drop_glu, bmnas_dropout_t drop_fc,
                           float* bn_part, void* stream);
/* ---- deterministic mode (BMNAS_DETERMINISTIC=1; bmnas.cell.DETERMINISTIC) --------------------------------------
 * Run-to-run bit-identical results for the search step with node_multiplier == 1 under the fused head.  The reference's
 * CPU path is deterministic; the default HIP path accumulates batch reductions with fp32 atomics, whose order varies.
 * In this mode every such reduction is a set of partials written with plain stores and summed in a fixed order:
 *   BatchNorm batch statistics     per-n-group partials + bmnas_bn_finalize (stat == NULL form of bmnas_conv1x1_fwd*)
 *   head logits | A | B            hb_part of bmnas_head_fwd_lazy: [slices][round_up(3 b O, 4)] floats
 *                                  (bmnas_head_fwd_part_floats bounds it); the launcher sums the slices into hb
 *   criterion                      loss_part of bmnas_head_bwd_lazy: [bmnas_head_chunks(b)] partial losses, summed by
 *                                  the caller in order
 *   BatchNorm affine gradients     bn_part of bmnas_node_mix_lnp_bwd: [bmnas_node_mix_lnp_bwd_rows(b)][6 C]; the
 *                                  launcher sums the rows into bn_grad
 *   dalpha / dbeta / dgamma        the caller passes as many shard copies as the launches have workgroups: one add
 *                                  per address, summed in shard order by the epilogue
 *   weight gradients, LayerNorm    bmnas_conv1x1_set_deterministic / bmnas_ln_set_deterministic: the weight-gradient
 *   affine gradients               tiles walk the whole batch (no batch splits), the affine reductions take one chunk
 * hb_part / loss_part / bn_part NULL: the default atomic forms. */
int bmnas_head_fwd_part_floats(int b, int C, int L, int n_src, int O);   /* < 0: BMNAS_E_* (BMNAS_E_LIMIT above 2^31) */
int bmnas_node_mix_lnp_bwd_rows(int b);
int bmnas_conv1x1_set_deterministic(int on);
int bmnas_ln_set_deterministic(int on);

/* Diagnostics (timing builds with -DBMNAS_BODY_PROBES=1 only; BMNAS_E_LIMIT otherwise): thread 0 of every workgroup
 * of the instrumented kernels records the shader clock at up to six points and the 100 MHz wall clock at entry / exit
 * into buf[kernel slot][workgroup][8] (uint64; slots workgroups per kernel, 8 kernel slots).  tools/stamp_probe.py. */
int bmnas_debug_stamps(void* buf, int slots);        /* csrc/lazyln.hip: slots 0-3 */
int bmnas_debug_stamps_head(void* buf, int slots);   /* csrc/head.hip: slots 4-5 */
int bmnas_debug_stamps_conv(void* buf, int slots);   /* csrc/conv1x1.hip: slot 6 (per-chunk phases of the data-gradient tiles) */

/* ---- central_classifier + criterion epilogue (the callers' side of the path) ---------------
 * out[m, o] = bias[o] + sum_k feat[m, k] * W[o, k]  — nn.Linear(M*C*L, classes) at
 * mmimdb_darts_searchable.py:82-83,114 (O <= 128, K % 16 == 0).  The k-slices ADD into `out`: out_is_zero != 0 says the
 * caller hands it over zero-filled (a captured step's arena, cleared by bmnas_copy_batch); 0: this call clears it. */
int bmnas_linear_fwd(const float* feat, const float* W, const float* bias, float* out, int b, int O,
                     int K, int out_is_zero, void* stream);
/* g: gradient of out, times *gscale if gscale != NULL.  dfeat (b, K), dW (O, K), dbias (O) are
 * OVERWRITTEN; any of them may be NULL. */
int bmnas_linear_bwd(const float* g, const float* gscale, const float* feat, const float* W,
                     float* dfeat, float* dW, float* dbias, int b, int O, int K, void* stream);
/* BCEWithLogitsLoss(reduction='mean') (mmimdb_darts_searchable.py:22) over n = b*O elements:
 * loss[0] and, if dz != NULL, dz = dloss/dz. */
int bmnas_bce_logits(const float* z, const float* y, float* loss, float* dz, int n, void* stream);
/* CrossEntropyLoss(reduction='mean') over int64 class labels (ntu_darts_searchable.py:25,
 * ego_darts_searchable.py:24); row_loss: (b) scratch. */
int bmnas_cross_entropy(const float* z, const int64_t* label, float* loss, float* dz,
                        float* row_loss, int b, int O, void* stream);

/* ---- multi-tensor Adam (row f2) ----------------------------------------------------------
 * One launch applies torch.optim.Adam's update (amsgrad off; L2 weight decay added to the
 * gradient) to every tensor of an optimizer — the w-step at train_searchable/mmimdb.py:101 and the
 * alpha-step at architect.py:24, optimizers built at mmimdb_darts_searchable.py:28-33.
 *   tensors: DEVICE array of descriptors (device pointers, fp32; any alignment, any numel);
 *   chunks:  DEVICE array of n_chunks (tensor index, chunk index) int32 pairs, one workgroup each,
 *            chunk c of a tensor covers elements [c*E, (c+1)*E), E = bmnas_adam_chunk_elems();
 *   hyp:     DEVICE array of 8-float rows, tensor -> row by hyp_row:
 *            { -(lr / (1 - beta1^t)), sqrt(1 - beta2^t), beta1, beta2, eps, weight_decay,
 *              1 - beta1, 1 - beta2 }  computed by the host in double (as torch does) and refreshed
 *            before each launch, so a captured graph replays with new rates / step counts. */
typedef struct {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  int32_t hyp_row;
  int32_t reserved;
} bmnas_adam_tensor_t;
int bmnas_adam_chunk_elems(void);
int bmnas_adam_multi(const bmnas_adam_tensor_t* tensors, const int32_t* chunks, int n_chunks,
                     const float* hyp, void* stream);

/* ---- a batch into a captured step's static tensors (bmnas.graph._copy_batch_in) -------------------
 * The reference hands every batch over with `.to(device)` per tensor (train_searchable/mmimdb.py:60-63,
 * ntu.py:60-66, ego.py:60-66); a replayed step reads its inputs from fixed addresses, so a batch that is
 * already on the device is copied there first — n <= bmnas_copy_batch_max() device-to-device copies of ANY
 * dtype (byte counts) as ONE launch, the (src, dst, bytes) triples by value in the kernel arguments.
 * 16-byte lanes where both addresses are 16-byte aligned, bytes otherwise.  Overlapping src / dst: undefined. */
int bmnas_copy_batch_max(void);
/* blob (optional, blob_bytes % 4 == 0, <= bmnas_copy_blob_max() bytes of HOST memory): copied by value into the kernel
 * arguments at the call and stored to the device address blob_dst by the same launch — the per-step scalars of a
 * captured optimizer step (bmnas_adam_multi's `hyp` rows: the learning rate the reference's scheduler sets per batch,
 * models/auxiliary/scheduler.py, and Adam's bias corrections) reach the device with the batch instead of through an
 * H2D copy node inside the step's graph.  n may be 0 (the blob alone).
 * srcs[i] == NULL: dsts[i] is ZERO-FILLED instead (the accumulators a captured per-op step adds into: cleared in front of
 * the replay by this launch, not by fill launches inside it).  add_dst (nullable): a 64-bit device counter advanced by
 * add_val by the same launch (the dropout step counter of a captured step that has no fused cell prologue to do it). */
int bmnas_copy_blob_max(void);
int bmnas_copy_batch(const void* const* srcs, void* const* dsts, const long long* bytes, int n,
                     void* blob_dst, const void* blob, int blob_bytes, unsigned long long* add_dst,
                     unsigned long long add_val, void* stream);

/* ---- data parallelism: RCCL behind the C ABI ---------------------------------------------------
 * One process per GPU; the data-path exchange of a search step is ONE in-place all-reduce of the flat
 * fp32 gradient bucket (w-grads, or the alpha/beta/gamma vector) over xGMI — what replaces
 * torch.nn.DataParallel at mmimdb_darts_searchable.py:36-37, ntu_darts_searchable.py:50-52,
 * ego_darts_searchable.py:51-53.  librccl is bound lazily (dlopen): without it these return -4 and
 * nothing else is affected.  Returns > 1000: RCCL's ncclResult_t + 1000.
 *   rank 0:  bmnas_comm_get_unique_id(id)  -> ship the bmnas_comm_unique_id_bytes() bytes to every rank
 *   all:     bmnas_comm_init_rank(&comm, world, rank, id)      (collective)
 *   step:    bmnas_allreduce_f32(bucket, n, average, comm, stream)   — asynchronous on `stream`, so it
 *            can be captured into the step's hipGraph between the backward and the Adam launch
 *   end:     bmnas_comm_destroy(comm) */
int bmnas_comm_available(void);
int bmnas_comm_unique_id_bytes(void);
int bmnas_comm_get_unique_id(void* id_out);
int bmnas_comm_init_rank(void** comm_out, int world, int rank, const void* id);
int bmnas_comm_destroy(void* comm);
int bmnas_allreduce_f32(float* buf, int64_t count, int average, void* comm, void* stream);
/* What the communicator itself reports — ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion —
 * each pointer nullable; -1 where librccl lacks the query.  bench.py prints them in its N > 1 line as
 * evidence that the collective spans the ranks the launcher started (the reference has no counterpart:
 * nn.DataParallel lives in one process). */
int bmnas_comm_info(void* comm, int* n_ranks, int* user_rank, int* hip_device, int* rccl_version);

/* ---- diagnostics: read-width calibration for rocprofv3's FETCH_SIZE (tools/calibrate_fetch.sh) ----
 * Reads p[0 .. n_floats) exactly once with `width` bytes per lane per load (4, 8, 16), or (width 64) as
 * 64-byte rows at row_stride floats — the operand pattern of the split-K GEMM kernels, which then reads
 * 64 bytes of every row_stride * 4.  Not part of the hypernet path. */
int bmnas_probe_read(const float* p, int64_t n_floats, int width, int row_stride, float* sink, void* stream);
/* Diagnostics (tools/barrier_probe.py; DESIGN.md, the persistent cell-step question): two dependent streaming phases
 * over n_floats (phase B reads what other workgroups wrote in phase A) as two launches (mode 0) or as ONE launch with a
 * grid-wide barrier between them (mode 1: agent-scope release -> counter -> poll -> acquire; `counter` zeroed once,
 * `round` = 1-based call number).  blocks <= 256 (all resident).  Not part of the hypernet path. */
int bmnas_probe_barrier(const float* in, float* tmp, float* out, int64_t n_floats, int blocks, int mode,
                        unsigned int* counter, int round, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BMNAS_HIP_H */
