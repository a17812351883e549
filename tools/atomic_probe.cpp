// Price of the split-K hand-off the round-6 decode step uses between the attention launch and the gate launch:
// 256 workgroups x 256 threads each add ONE int32 partial per thread (no-return device-scope atomics) into 2048 accumulators
// (32 workgroups per address, as 32 heads x 8 row ranges of o_proj), the NEXT kernel reads all 2048 sums on every CU and zeroes them.
// Measured as the period of {producer, consumer} pairs inside one hipGraph, against the same pair with plain stores (no reduction).
//   hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.cpp -o tools/atomic_probe && tools/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int slot(int n, int layout) {
  // layout 0: contiguous 8 KB; 1: 64-int (256 B) lines 4352 B apart; 2: 16-int (64 B) pieces 1088 B apart
  if (layout == 1) return (n & 63) + (n >> 6) * 1088;
  if (layout == 2) return (n & 15) + (n >> 4) * 272;
  return n;
}

// MODE 0: nothing; 1: no-return agent atomics; 2: returning atomics (value used); 3: plain store (no reduction: the last writer wins)
// MAP 0: row range = block % 8 (the 32 adders of one address sit on ONE XCD); 1: row range = block / 32 (spread over all XCDs)
template <int MODE, int MAP>
__global__ void __launch_bounds__(256) producer(int* acc, int layout, int* sink) {
  const int b = blockIdx.x, c = MAP == 0 ? (b & 7) : (b >> 5), h = MAP == 0 ? (b >> 3) : (b & 31);
  const int n = c * 256 + threadIdx.x;
  const int v = h + 1;
  if (MODE == 1) __hip_atomic_fetch_add(acc + slot(n, layout), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 2) {
    const int old = __hip_atomic_fetch_add(acc + slot(n, layout), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0x7fffffff) sink[0] = old;
  }
  if (MODE == 3) acc[slot(n, layout)] = v;
}

// every workgroup reads all 2048 sums (as the gate launch's prologue does), workgroup j checks and zeroes its 8
__global__ void __launch_bounds__(256) consumer(int* acc, int layout, int expect, int* bad, int* sink) {
  int s = 0;
  for (int i = threadIdx.x; i < 2048; i += 256) s += acc[slot(i, layout)];
  if (s == 0x12345678) sink[1] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    const int n = blockIdx.x * 8 + threadIdx.x;
    if (expect && acc[slot(n, layout)] != expect) atomicAdd(bad, 1);
  }
}
__global__ void zero(int* acc) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < 65536; i += gridDim.x * 256) acc[i] = 0;
}

template <int MODE, int MAP>
static float run(int* acc, int layout, int* bad, int* sink, const char* name) {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t g;
  hipGraphExec_t ge;
  const int pairs = 50;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < pairs; ++i) {
    zero<<<8, 256, 0, st>>>(acc);
    producer<MODE, MAP><<<256, 256, 0, st>>>(acc, layout, sink);
    consumer<<<256, 256, 0, st>>>(acc, layout, MODE == 1 || MODE == 2 ? 32 * 33 / 2 : 0, bad, sink);
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r && ms < best) best = ms;
  }
  int hbad = 0;
  CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
  printf("%-44s layout %d : %7.3f us per {zero, producer, consumer} triple   wrong sums %d\n", name, layout, best * 1e3f / pairs, hbad);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  CK(hipStreamDestroy(st));
  return best;
}

int main() {
  int *acc, *bad, *sink;
  CK(hipMalloc(&acc, 65536 * 4));
  CK(hipMalloc(&bad, 4));
  CK(hipMalloc(&sink, 16));
  CK(hipMemset(bad, 0, 4));
  for (int layout = 0; layout < 3; ++layout) {
    run<0, 0>(acc, layout, bad, sink, "no producer work");
    run<3, 0>(acc, layout, bad, sink, "plain stores");
    run<1, 0>(acc, layout, bad, sink, "no-return atomics, adders of an address on 1 XCD");
    run<1, 1>(acc, layout, bad, sink, "no-return atomics, adders spread over XCDs");
    run<2, 0>(acc, layout, bad, sink, "returning atomics, 1 XCD");
    run<2, 1>(acc, layout, bad, sink, "returning atomics, spread");
  }
  return 0;
}
