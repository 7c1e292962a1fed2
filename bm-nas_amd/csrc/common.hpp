// Shared device helpers for the BM-NAS fusion-cell kernels (gfx950 / CDNA4 only).
// wave = 64 lanes everywhere; fp32 MFMA 16x16x4 is the only matrix instruction used.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BMNAS_MAX_PTRS 16
#define BMNAS_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;

struct PtrsIn { const float* p[BMNAS_MAX_PTRS]; };
struct PtrsOut { float* p[BMNAS_MAX_PTRS]; };

// Counter-based dropout: keep element e iff philox(seed, offset + e/4)[e%4] >= thr.
// thr == 0 -> identity (eval mode or p == 0).  scale = 1/(1-p).
struct DropCfg {
  uint32_t thr;
  float scale;
  uint64_t seed;
  uint64_t offset;
  const uint64_t* step;   // device counter added to offset (hipGraph replays), nullable
};

#define BMNAS_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return (int)e__;          \
  } while (0)

__device__ __forceinline__ uint4 philox4x32_10(uint64_t ctr, uint64_t seed) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x2545F491u, c3 = 0x9E3779B1u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// The same with the device step counter already read: kernels call drop_begin() FIRST, so that the counter's
// load goes out with their operand loads.  Read inside drop_mult4 (hipGraph replays: step != nullptr) it was a
// load -> s_waitcnt vmcnt(0) in the middle of the arithmetic, once per dropout site — dependent round trips
// on the critical path of every kernel with a dropout.
struct DropRt {
  uint32_t thr;
  float scale;
  uint64_t seed;
  uint64_t off;
};
__device__ __forceinline__ DropRt drop_begin(const DropCfg& d) {
  DropRt r{d.thr, d.scale, d.seed, d.offset};
  if (d.thr != 0u && d.step != nullptr) r.off += d.step[0];
  return r;
}
__device__ __forceinline__ float4 drop_mult4(const DropRt& d, uint64_t e) {
  if (d.thr == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  uint4 r = philox4x32_10(d.off + (e >> 2), d.seed);
  return make_float4(r.x >= d.thr ? d.scale : 0.f, r.y >= d.thr ? d.scale : 0.f,
                     r.z >= d.thr ? d.scale : 0.f, r.w >= d.thr ? d.scale : 0.f);
}

// mask*scale multipliers for the float4 starting at flat element index e (e % 4 == 0)
__device__ __forceinline__ float4 drop_mult4(const DropCfg& d, uint64_t e) {
  if (d.thr == 0u) return make_float4(1.f, 1.f, 1.f, 1.f);
  const uint64_t base = (d.step != nullptr) ? d.step[0] : 0ull;
  uint4 r = philox4x32_10(base + d.offset + (e >> 2), d.seed);
  return make_float4(r.x >= d.thr ? d.scale : 0.f, r.y >= d.thr ? d.scale : 0.f,
                     r.z >= d.thr ? d.scale : 0.f, r.w >= d.thr ? d.scale : 0.f);
}

// ---- in-kernel stamps (diagnostic builds only: -DBMNAS_BODY_PROBES=1; tools/stamp_probe.py) -------------------
// Thread 0 of every workgroup records the shader clock (s_memtime) at up to 6 points plus the 100 MHz wall clock
// (s_memrealtime) at entry and exit into a buffer of its own that nothing else reads (bmnas_debug_stamps).  In a
// production build the macros expand to nothing and no stamp executes.
#ifndef BMNAS_BODY_PROBES
#define BMNAS_BODY_PROBES 0
#endif
#if BMNAS_BODY_PROBES
// (no relocatable device code in this build: every translation unit that stamps has its OWN pointer, set by the
// setter it defines with BMNAS_DEFINE_STAMP_SETTER)
static __device__ unsigned long long* g_bmnas_stamps = nullptr;
static __device__ int g_bmnas_stamp_slots = 0;
#define BMNAS_DEFINE_STAMP_SETTER(name)                                                                   \
  extern "C" int name(void* buf, int slots) {                                                             \
    unsigned long long* p = reinterpret_cast<unsigned long long*>(buf);                                   \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bmnas_stamps), &p, sizeof(p)) != hipSuccess) return -1;            \
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bmnas_stamp_slots), &slots, sizeof(slots)) != hipSuccess) return -1; \
    return 0;                                                                                             \
  }
__device__ __forceinline__ unsigned long long stamp_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long stamp_wall() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// slot: which kernel of the probe (0 ... 7); wg: linear workgroup index; i: stamp index 0 ... 5 (6 = wall at entry, 7 = wall now)
#define STAMP(slot, wg, i)                                                                         \
  do {                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    if (threadIdx.x == 0 && g_bmnas_stamps != nullptr && (wg) < g_bmnas_stamp_slots) {             \
      unsigned long long* p__ = g_bmnas_stamps + ((size_t)(slot) * g_bmnas_stamp_slots + (wg)) * 8; \
      p__[i] = stamp_clock();                                                                      \
      if ((i) == 0) p__[6] = stamp_wall();                                                         \
      p__[7] = stamp_wall();                                                                       \
    }                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
  } while (0)
#else
#define STAMP(slot, wg, i) do { } while (0)
#define BMNAS_DEFINE_STAMP_SETTER(name) \
  extern "C" int name(void* buf, int slots) { (void)buf; (void)slots; return -3; /* BMNAS_E_LIMIT: no stamps here */ }
#endif

// ---- cross-lane reductions WITHOUT the LDS pipe ------------------------------------------------------------
// hipcc lowers every __shfl_xor to ds_bpermute_b32 + s_waitcnt lgkmcnt: an LDS round trip per step.  A block
// reduction of ten values was 60 of them back to back — in-kernel stamps put 2.8 us of an 8 us K1 backward launch
// there (tools/stamp_probe.py, profiles/r04_stamp_probe.txt).  These forms stay in the vector ALU:
//   inside a 16-lane row: DPP quad_perm (exact xor 1 / xor 2 partners) and row_ror:4 / row_ror:8;
//   across rows: gfx950's v_permlane16_swap / v_permlane32_swap — after swap(v, v) the two results hold, lane for
//   lane, the value of the lane itself and of its xor-16 (xor-32) partner, in one order or the other: a symmetric
//   op (+, max) needs no select.
// All of them expect the wave's 64 lanes active (as __shfl_xor did).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_xor1(float v) { return dpp_f<0xB1>(v); }     // quad_perm:[1,0,3,2]
__device__ __forceinline__ float lane_xor2(float v) { return dpp_f<0x4E>(v); }     // quad_perm:[2,3,0,1]
__device__ __forceinline__ float xor16_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor16_max(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_max(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum of the lanes {i, i^4, i^8, i^12} of a row (the lanes with the same lane & 3), in every one of them
__device__ __forceinline__ float row_stride4_sum(float v) {
  v += dpp_f<0x124>(v);                                                            // row_ror:4
  v += dpp_f<0x128>(v);                                                            // row_ror:8
  return v;
}
// sum / max over the 16 lanes of a row (lanes 16 r .. 16 r + 15), in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
  v += lane_xor1(v);
  v += lane_xor2(v);
  return row_stride4_sum(v);
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, lane_xor1(v));
  v = fmaxf(v, lane_xor2(v));
  v = fmaxf(v, dpp_f<0x124>(v));
  v = fmaxf(v, dpp_f<0x128>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(row16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(row16_max(v))); }

// Sum over the 256 threads of a block; every thread gets the result.
// red must hold >= 4 floats; two __syncthreads.
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// The same for a block of NW waves (red must hold >= NW floats).
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += red[i];
  return t;
}

// The same without the barrier in FRONT of the LDS writes, for an array that no thread can still be reading: one
// that this launch has not used yet, or whose last readers are separated from this call by a barrier.  (A kernel with
// two dependent reductions — mean, then centred moments — pays two barriers instead of four with two arrays.)
template <int NW>
__device__ __forceinline__ float block_sum_fresh(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) t += red[i];
  return t;
}

template <int NW, int N>
__device__ __forceinline__ void block_sum_lead_fresh(float (&v)[N], float* red) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[i * NW + w] = v[i];
  }
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) t += red[k];
  v[0] = t;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < N; ++i) {
      float u = red[i * NW];
#pragma unroll
      for (int k = 1; k < NW; ++k) u += red[i * NW + k];
      v[i] = u;
    }
  }
}

// N sums at once over a block of NW waves (red must hold >= NW * N floats); every thread gets all N.
template <int NW, int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float* red) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[w * N + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = red[i];
#pragma unroll
    for (int k = 1; k < NW; ++k) t += red[k * N + i];
    v[i] = t;
  }
}

// As block_sum_n, but only v[0] is handed to every thread; v[1..N-1] are complete on thread 0 only
// (the other threads skip their LDS reads).
template <int NW, int N>
__device__ __forceinline__ void block_sum_lead(float (&v)[N], float* red) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) red[i * NW + w] = v[i];
  }
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int k = 1; k < NW; ++k) t += red[k];
  v[0] = t;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < N; ++i) {
      float u = red[i * NW];
#pragma unroll
      for (int k = 1; k < NW; ++k) u += red[i * NW + k];
      v[i] = u;
    }
  }
}

// Picking one of a few kernel-argument pointers by a run-time (per-lane) index:  never `args.p[q]` — that
// is a MEMORY load of the pointer from the kernarg segment plus an s_waitcnt vmcnt(0) in front of the load
// it feeds (one more memory round trip, and every earlier load of the wave is waited for).  A chain of
// selects over the compile-time-indexed pointers keeps them in SGPRs; the empty asm keeps LLVM from folding
// select(load, load) back into load(select(address, address)).
template <typename T>
__device__ __forceinline__ T* sgpr_ptr(T* p) {
  asm volatile("" : "+s"(p));
  return p;
}
template <typename T, int N>
__device__ __forceinline__ T* pick_ptr(T* const (&arr)[N], int q) {
  T* p = sgpr_ptr(arr[0]);
#pragma unroll
  for (int qq = 1; qq < N; ++qq) {
    T* c = sgpr_ptr(arr[qq]);
    p = (q == qq) ? c : p;
  }
  return p;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Write-through (sc1) 16-byte store: the bytes go to memory when the store is issued instead of staying dirty in the
// XCD's L2 until the kernel's end-of-launch write-back (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s for B dirty
// bytes) — for tensors the NEXT launch reads, mostly from other XCDs anyway.  hipcc does not count an asm store in its
// waits: only for data this wave never loads again, and ONLY for pointers into global memory (an LDS or scratch address
// in a global_store faults).  (`s_nop 1`: the store reads its data registers after issue.)
__device__ __forceinline__ void st4_wt(float* p, float4 v) {
#if defined(BMNAS_NO_WT)
  *reinterpret_cast<float4*>(p) = v;
#else
  const f32x4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
#endif
}

// Store-policy groups (which launches write through is a measured choice, profiles/r06_write_through.txt): bit G of
// BMNAS_WT_MASK turns group G on.  0: the streaming / GEMM launches of the lazy-LayerNorm search path (st4_wt direct);
// 2: bnmix.hip's mix / BatchNorm-tail kernels; 3: layernorm.hip; 4: split-K conv epilogues + mixconv; 5: adam / linear.
// Default 0x29: groups 2 and 4 stay plain — NTU / Ego's per-sample workgroups read what the SAME XCD wrote one launch
// earlier (sample s -> workgroup s in producer and consumer), and a write-through store drops the line from that L2.
#ifndef BMNAS_WT_MASK
#define BMNAS_WT_MASK 0x29
#endif
template <int G>
__device__ __forceinline__ void st4_wtg(float* p, float4 v) {
  if constexpr ((BMNAS_WT_MASK >> G) & 1) st4_wt(p, v);
  else *reinterpret_cast<float4*>(p) = v;
}

// ... and inside group 0, by kernel (bit K of BMNAS_WT0_MASK; default: all write through): 0 prologue pair h / z, 1 conv
// forward U, 2 attention forward, 3 node_mix_pre_fwd `pre`, 4 mixsum_pair_fwd_lazy, 5 head_bwd state gradients, 6
// node_mix_lnp_bwd, 7 data-gradient tiles, 8 attention backward, 9 mixsum_pair_bwd_lazy, 10 mixsum_pair_bwd_x
#ifndef BMNAS_WT0_MASK
#define BMNAS_WT0_MASK 0x7FF
#endif
template <int K>
__device__ __forceinline__ void st4_w0(float* p, float4 v) {
  if constexpr ((BMNAS_WT0_MASK >> K) & 1) st4_wt(p, v);
  else *reinterpret_cast<float4*>(p) = v;
}

__device__ __forceinline__ void st16_wt(void* p, uint4 v) {     // the same for 16 raw bytes
#if defined(BMNAS_NO_WT) || !((BMNAS_WT_MASK >> 5) & 1)
  *reinterpret_cast<uint4*>(p) = v;
#else
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4_;
  const u32x4_ d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
#endif
}

__device__ __forceinline__ float4 f4_add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_mul(float4 a, float4 b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_scale(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float f4_hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }

static inline int ilog2_exact(int v) {   // host: log2 of a power of two, else -1
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}
