// What does ONE more launch cost inside a captured step?  A hipGraph of 20 dependent launches of a kernel whose
// workgroups return at once, replayed 200 times: microseconds per launch as a function of grid size, workgroup size,
// dynamic LDS request and kernel-argument bytes.  (tools/micro/: measurement programs, not part of the library.)
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/micro/bin/launch_floor tools/micro/launch_floor.hip && tools/micro/bin/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

template <int BYTES>
struct Blob { char c[BYTES]; };

template <int BYTES>
__global__ void empty_k(Blob<BYTES> a, int* sink) {
  extern __shared__ char smem[];
  if (a.c[0] == 77 && sink) sink[0] = smem[threadIdx.x];      // never true: keeps the arguments and the LDS alive
}

template <int BYTES>
double run(int grid, int block, size_t lds, int* sink) {
  hipStream_t st;
  hipStreamCreate(&st);
  if (lds > 65536) hipFuncSetAttribute(reinterpret_cast<const void*>(empty_k<BYTES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  Blob<BYTES> a{};
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_k<BYTES>, dim3(grid), dim3(block), lds, st, a, sink);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(st);
  return us / (200 * 20);
}

int main() {
  int* sink;
  hipMalloc(&sink, 4096);
  printf("%-8s %-6s %-8s %-8s %s\n", "grid", "block", "lds", "args", "us/launch");
  const int grids[] = {1, 64, 256, 512, 1024, 2048};
  for (int g : grids) printf("%-8d %-6d %-8d %-8d %.2f\n", g, 256, 0, 64, run<64>(g, 256, 0, sink));
  const size_t ldss[] = {0, 16384, 38912, 65536, 81920};
  for (size_t l : ldss) printf("%-8d %-6d %-8zu %-8d %.2f\n", 512, 256, l, 64, run<64>(512, 256, l, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 512, 256, 0, 1024, run<1024>(512, 256, 0, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 512, 256, 0, 3072, run<3072>(512, 256, 0, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 512, 256, 65536, 1024, run<1024>(512, 256, 65536, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 256, 512, 0, 64, run<64>(256, 512, 0, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 128, 1024, 0, 64, run<64>(128, 1024, 0, sink));
  printf("%-8d %-6d %-8d %-8d %.2f\n", 64, 256, 65536, 1024, run<1024>(64, 256, 65536, sink));
  return 0;
}
