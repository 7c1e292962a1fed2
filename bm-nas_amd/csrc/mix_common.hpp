// Helpers shared by the sources that apply / differentiate the NodeMixedOp mix (bnmix.hip, lazyln.hip).
#pragma once
#include "common.hpp"
#include "bn_fin.hpp"
#include "../../include/bmnas_hip.h"

namespace {

__device__ __forceinline__ float4 affine4(float4 u, float sc, float sh) {
  return make_float4(fmaf(u.x, sc, sh), fmaf(u.y, sc, sh), fmaf(u.z, sc, sh), fmaf(u.w, sc, sh));
}
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + __expf(-v)); }

// reduce v over the l4n adjacent lanes that share a channel row (l4n in {1, 2, 4})
__device__ __forceinline__ float row_sum(float v, int l4n) {
  if (l4n >= 2) v += lane_xor1(v);
  if (l4n >= 4) v += lane_xor2(v);
  return v;
}

inline DropCfg to_cfg(const bmnas_dropout_t& d) {
  DropCfg c;
  c.thr = d.thr; c.scale = d.scale; c.seed = d.seed; c.offset = d.offset; c.step = d.step;
  return c;
}

// bmnas_bn_fin_t -> BnFin; < 0 on a bad descriptor
inline int to_fin(const bmnas_bn_fin_t& f, BnFin* o) {
  o->on = f.on ? 1 : 0;
  if (!o->on) {
    *o = BnFin{};
    return 0;
  }
  if (!f.bn_w || !f.bn_b || f.shards < 0 || f.n_nbt < 0) return BMNAS_E_ARG;
  if (f.shards > 4) return BMNAS_E_LIMIT;
  if (f.training && (!f.stat || f.shards < 1)) return BMNAS_E_ARG;
  if (!f.training && (!f.running_mean || !f.running_var)) return BMNAS_E_ARG;
  if ((f.running_mean == nullptr) != (f.running_var == nullptr)) return BMNAS_E_ARG;
  o->stat = f.stat; o->conv_bias = f.conv_bias; o->bn_w = f.bn_w; o->bn_b = f.bn_b;
  o->running_mean = f.running_mean; o->running_var = f.running_var;
  o->nbt = reinterpret_cast<long long*>(f.num_batches_tracked);
  o->shards = f.shards; o->n_nbt = f.n_nbt; o->training = f.training ? 1 : 0;
  return 0;
}

}  // namespace
