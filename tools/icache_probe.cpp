// Does a short kernel pay for cold instruction fetch on MI355X?  (the decode step's kernels run 2-7 us each)
//   hipcc --offload-arch=gfx950 -O2 tools/icache_probe.cpp -o tools/icache_probe && tools/icache_probe
// The same arithmetic (N dependent-chain-free fma's on 8 accumulators) as straight-line code (N x 8 bytes of instructions) and as
// a rolled loop (one cache line of code); a hipGraph of 64 launches of the SAME kernel; per launch thread 0 of every workgroup
// stamps s_memtime at entry and exit.  If the instruction cache survived a kernel boundary, launches 2.. of the straight-line kernel
// would run at the loop's speed.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int N, bool ROLLED>
__global__ void __launch_bounds__(256) body(float* out, unsigned long long* stamps, int launch, float b, float c) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (float)threadIdx.x + i;
  if (ROLLED) {
#pragma unroll 1
    for (int k = 0; k < N / 8; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N / 8; ++k) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  asm volatile("" ::"v"(s));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    stamps[((size_t)launch * gridDim.x + blockIdx.x) * 2] = t1 - t0;
  }
  if (s == 12345.678f) out[0] = s;
}

template <int N, bool ROLLED>
int run(float* out, unsigned long long* stamps, const char* what) {
  const int L = 64, G = 256;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < L; ++i) body<N, ROLLED><<<G, 256, 0, st>>>(out, stamps, i, 1.0001f, 0.5f);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(exec, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(exec, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)L * G * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double mean = 0, mx = 0, mn = 1e18;
  for (int i = 8; i < L; ++i)
    for (int b = 0; b < G; ++b) {
      const double c = (double)h[((size_t)i * G + b) * 2];
      mean += c; mx = std::max(mx, c); mn = std::min(mn, c);
    }
  mean /= (L - 8) * G;
  printf("%-34s N = %5d fma | period %.2f us | wave cycles entry->exit mean %.0f min %.0f max %.0f | %.2f cycles per fma\n", what, N, ms * 1e3 / (10 * L), mean, mn, mx, mean / N);
  CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
  return 0;
}

int main() {
  float* out;
  unsigned long long* stamps;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&stamps, (size_t)64 * 256 * 2 * 8));
  if (run<256, false>(out, stamps, "straight-line")) return 1;
  if (run<256, true>(out, stamps, "rolled loop")) return 1;
  if (run<1024, false>(out, stamps, "straight-line")) return 1;
  if (run<1024, true>(out, stamps, "rolled loop")) return 1;
  if (run<4096, false>(out, stamps, "straight-line")) return 1;
  if (run<4096, true>(out, stamps, "rolled loop")) return 1;
  return 0;
}
