// Read-width calibration kernels for rocprofv3's FETCH_SIZE counter (tools/calibrate_fetch.sh).
// MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half the bytes of a 16-B-per-lane
// coalesced streaming read and other access widths are uncalibrated — these kernels read a buffer of
// known size once with 4 / 8 / 16 bytes per lane (and the 64-byte-row pattern of the GEMM operand
// loads), so the counter can be divided into the known byte count.  Diagnostics only: nothing on the
// hypernet path launches them.
#include "common.hpp"
#include "../../include/bmnas_hip.h"

namespace {

template <typename T>
__device__ __forceinline__ float lane_sum(T v);
template <>
__device__ __forceinline__ float lane_sum<float>(float v) { return v; }
template <>
__device__ __forceinline__ float lane_sum<float2>(float2 v) { return v.x + v.y; }
template <>
__device__ __forceinline__ float lane_sum<float4>(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// every element of p[0 .. n) read exactly once, sizeof(T) bytes per lane per load, wave-contiguous
template <typename T>
__global__ __launch_bounds__(256) void probe_read_k(const float* __restrict__ p, int64_t n_t, float* sink) {
  const T* q = reinterpret_cast<const T*>(p);
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_t; i += stride) acc += lane_sum<T>(q[i]);
  if (acc == 123456.789f) sink[0] = acc;               // keeps the loads alive
}

// the A-operand pattern of the split-K GEMM kernels: a wave reads 16 rows of 16 floats (64-byte rows,
// row stride `ld` floats), one dword per lane; rows of consecutive waves are adjacent
__global__ __launch_bounds__(256) void probe_rows_k(const float* __restrict__ p, int64_t n_rows, int ld,
                                                    float* sink) {
  const int lane = threadIdx.x & 63, lo = lane & 15, h = lane >> 4;
  float acc = 0.f;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave * 16; r0 + 16 <= n_rows; r0 += n_waves * 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += p[(r0 + 4 * h + r) * ld + lo];
  }
  if (acc == 123456.789f) sink[0] = acc;
}

}  // namespace

extern "C" int bmnas_probe_read(const float* p, int64_t n_floats, int width, int row_stride, float* sink,
                                void* stream) {
  if (!p || !sink || n_floats < 0) return BMNAS_E_ARG;
  if (n_floats % 4) return BMNAS_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(2048), block(256);
  switch (width) {
    case 4: hipLaunchKernelGGL(probe_read_k<float>, grid, block, 0, st, p, n_floats, sink); break;
    case 8: hipLaunchKernelGGL(probe_read_k<float2>, grid, block, 0, st, p, n_floats / 2, sink); break;
    case 16: hipLaunchKernelGGL(probe_read_k<float4>, grid, block, 0, st, p, n_floats / 4, sink); break;
    case 64:                                           // 64-byte rows at row_stride floats
      if (row_stride < 16) return BMNAS_E_ARG;
      hipLaunchKernelGGL(probe_rows_k, grid, block, 0, st, p, n_floats / row_stride, row_stride, sink);
      break;
    default: return BMNAS_E_ARG;
  }
  BMNAS_CHECK_LAUNCH();
  return 0;
}

// ---- diagnostics: what a grid-wide barrier costs INSIDE a launch against a launch boundary (tools/barrier_probe.py) ----
// DESIGN.md's persistent cell-step question: two streaming phases, the second reading what OTHER workgroups wrote in
// the first (a slice half the buffer away), either as two launches or as one launch with a grid barrier in between:
// lane 0 of every workgroup: agent-scope release fence -> s_waitcnt vmcnt(0) -> counter add -> relaxed sc1 poll with
// s_sleep -> agent-scope acquire fence; then the workgroup's barrier (MI355X_MICROARCH.md, Valid forms).  The grid must
// be resident at once (<= 256 workgroups of 256 threads here).  Not part of the hypernet path.
namespace {

__device__ __forceinline__ void probe_phase(const float* __restrict__ in, float* __restrict__ out, int64_t n4, int64_t shift4,
                                            float mul) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    int64_t j = i + shift4;
    j = j >= n4 ? j - n4 : j;
    const float4 v = ld4(in + 4 * j);
    st4(out + 4 * i, make_float4(v.x * mul + 1.f, v.y * mul + 1.f, v.z * mul + 1.f, v.w * mul + 1.f));
  }
}

__global__ __launch_bounds__(256) void probe_phase_k(const float* in, float* out, int64_t n4, int64_t shift4, float mul) {
  probe_phase(in, out, n4, shift4, mul);
}

__global__ __launch_bounds__(256) void probe_fused_k(const float* in, float* tmp, float* out, int64_t n4, int64_t shift4,
                                                     unsigned int* counter, unsigned int target) {
  probe_phase(in, tmp, n4, 0, 0.5f);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1 << 22))
      __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  probe_phase(tmp, out, n4, shift4, 2.f);
}

}  // namespace

// mode 0: phase A (in -> tmp) and phase B (tmp -> out, shifted) as TWO launches; mode 1: ONE launch with a grid barrier
// (counter must have been zeroed by the caller for the first call; `round` = 1-based call number: the counter is
// monotonic).  blocks <= 256.
extern "C" int bmnas_probe_barrier(const float* in, float* tmp, float* out, int64_t n_floats, int blocks, int mode,
                                   unsigned int* counter, int round, void* stream) {
  if (!in || !tmp || !out || n_floats < 4 || n_floats % 4 || blocks < 1 || blocks > 256 || round < 1) return BMNAS_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n4 = n_floats / 4, shift4 = n4 / 2;
  if (mode == 0) {
    hipLaunchKernelGGL(probe_phase_k, dim3(blocks), dim3(256), 0, st, in, tmp, n4, (int64_t)0, 0.5f);
    hipLaunchKernelGGL(probe_phase_k, dim3(blocks), dim3(256), 0, st, (const float*)tmp, out, n4, shift4, 2.f);
  } else {
    if (!counter) return BMNAS_E_ARG;
    hipLaunchKernelGGL(probe_fused_k, dim3(blocks), dim3(256), 0, st, in, tmp, out, n4, shift4, counter,
                       (unsigned int)(blocks * round));
  }
  BMNAS_CHECK_LAUNCH();
  return 0;
}
