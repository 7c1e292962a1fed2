// What does a launch-to-launch hand-off cost?  A hipGraph of 20 dependent launches; launch i reads the buffer launch
// i - 1 wrote and writes the next one (ping-pong), one float4 per thread, 256-thread workgroups.  Per size:
//   * reader mapping: workgroup w reads what workgroup (w + shift) wrote — shift 0 = the same XCD wrote it (workgroups
//     go round-robin over the 8 XCDs), shift 1 = the neighbouring XCD, shift 4 = across;
//   * store flavour: plain (stays dirty in the writer's L2 until the end-of-launch write-back) or write-through (sc1).
// -> microseconds per launch.  Answers: do the XCDs' L2s keep a predecessor's lines across a kernel boundary (shift 0
// faster than shift 1?), and from which size on does write-through pay.  (tools/micro/: measurement programs, not part of
// the library.)
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/micro/bin/handoff_chain tools/micro/handoff_chain.hip && tools/micro/bin/handoff_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool WT>
__global__ __launch_bounds__(256) void hand_k(const float* __restrict__ src, float* __restrict__ dst, int nwg, int shift) {
  const int w = ((int)blockIdx.x + shift) % nwg;
  const float4 v = reinterpret_cast<const float4*>(src)[(size_t)w * 256 + threadIdx.x];
  const float4 o = make_float4(v.x + 1.f, v.y, v.z, v.w);
  float* p = dst + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (WT) {
    const f32x4 d = {o.x, o.y, o.z, o.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(d) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = o;
  }
}

// second table: load flavour LF (0 plain, 1 nt, 2 sc1, 3 sc0 sc1) x store flavour SF (0 plain, 1 sc1, 2 nt, 3 sc0 sc1)
template <int LF, int SF>
__global__ __launch_bounds__(256) void hand2_k(const float* __restrict__ src, float* __restrict__ dst, int nwg, int shift) {
  const int w = ((int)blockIdx.x + shift) % nwg;
  const float* q = src + ((size_t)w * 256 + threadIdx.x) * 4;
  f32x4 v;
  if (LF == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
  else if (LF == 1) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
  else if (LF == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(q) : "memory");
  v.x += 1.f;
  float* p = dst + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (SF == 0) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if (SF == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else if (SF == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

template <int LF, int SF>
double run2(int nwg, int shift, float* a, float* b) {
  hipStream_t st;
  hipStreamCreate(&st);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 20; ++i)
    hipLaunchKernelGGL((hand2_k<LF, SF>), dim3(nwg), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, nwg, shift);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
  return us / (200 * 20);
}

template <bool WT>
double run(int nwg, int shift, float* a, float* b) {
  hipStream_t st;
  hipStreamCreate(&st);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 20; ++i)
    hipLaunchKernelGGL(hand_k<WT>, dim3(nwg), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, nwg, shift);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
  return us / (200 * 20);
}

int main() {
  const size_t maxb = (size_t)64 << 20;
  float *a, *b;
  hipMalloc(&a, maxb);
  hipMalloc(&b, maxb);
  hipMemset(a, 0, maxb);
  hipMemset(b, 0, maxb);
  printf("%-10s %-8s %-22s %-22s\n", "bytes", "wgs", "plain shift 0 / 1 / 4", "sc1 shift 0 / 1 / 4");
  const int wgs[] = {8, 64, 256, 512, 1024, 2048, 4096, 8192, 16384};
  for (int n : wgs) {
    const size_t bytes = (size_t)n * 4096;
    printf("%-10zu %-8d %6.2f %6.2f %6.2f   %6.2f %6.2f %6.2f\n", bytes, n, run<false>(n, 0, a, b), run<false>(n, 1, a, b),
           run<false>(n, 4, a, b), run<true>(n, 0, a, b), run<true>(n, 1, a, b), run<true>(n, 4, a, b));
    fflush(stdout);
  }
  // cross-XCD reader (shift 1), load flavour x store flavour, at the step's tensor sizes
  printf("\ncross-XCD reader (shift 1): rows = load plain / nt / sc1 / sc0 sc1, columns = store plain / sc1 / nt / sc0 sc1\n");
  const int wgs2[] = {384, 1152, 2304};      // 1.5 MB (T), 4.7 MB (3 T), 9.4 MB (6 T)
  for (int n : wgs2) {
    printf("%zu bytes\n", (size_t)n * 4096);
#define ROW(LF) printf("  %6.2f %6.2f %6.2f %6.2f\n", run2<LF, 0>(n, 1, a, b), run2<LF, 1>(n, 1, a, b), run2<LF, 2>(n, 1, a, b), run2<LF, 3>(n, 1, a, b)); fflush(stdout);
    ROW(0) ROW(1) ROW(2) ROW(3)
#undef ROW
  }
  return 0;
}
