// Kernel-start memory latency on MI355X inside a chain of short dependent launches (the decode step's regime).
//   hipcc --offload-arch=gfx950 -O2 tools/latency_probe.cpp -o tools/latency_probe && tools/latency_probe
// Each launch: 256 workgroups x THREADS threads; thread 0 of each wave stamps s_memrealtime (100 MHz) at entry, issues its load(s),
// stamps again when the data has arrived.  A hipGraph of 64 launches is replayed; launch i reads
//   mode 0: the SAME 128-byte line in every launch and every workgroup            (hot line)
//   mode 1: one line per workgroup from a small buffer that every launch re-reads (L2 / MALL resident, TLB warm)
//   mode 2: one line per workgroup, 64 KiB apart, from 16 MiB no other launch of the graph touches (1 GiB per replay: HBM)
//   mode 3: mode 1, and then a 64 KiB stream per workgroup from fresh pages       (what the GEMV does)
//   mode 4: a line written by the PREVIOUS launch (other workgroup)               (the activation hand-off)
//   mode 5: mode 3, but launch 2k+1 re-reads what launch 2k read, same block -> same XCD           (L2-resident weight stream)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) probe(const char* __restrict__ small, const char* __restrict__ big, char* __restrict__ handoff,
                                              size_t big_off, unsigned long long* stamps, int launch, int* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int acc = 0;
  if (MODE == 0) acc = reinterpret_cast<const int*>(small)[lane & 31];
  if (MODE == 1 || MODE == 3 || MODE == 5) acc = reinterpret_cast<const int*>(small + (size_t)blockIdx.x * 128)[lane & 31];
  if (MODE == 2) acc = reinterpret_cast<const int*>(big + big_off + (size_t)blockIdx.x * (64u << 10))[lane & 31];
  if (MODE == 4) acc = reinterpret_cast<const int*>(handoff + (size_t)((blockIdx.x + 37) & 255) * 128)[lane & 31];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t2 = t1;
  if (MODE == 3 || MODE == 5) {
    v4i b[4];
    const char* p = big + big_off + (size_t)blockIdx.x * (64u << 10) + (size_t)wave * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) b[u] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(p + u * 1024) + lane);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += b[u][0] + b[u][3];
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc) : "memory");
    t2 = __builtin_amdgcn_s_memrealtime();
  }
  if (MODE == 4 && threadIdx.x < 32) reinterpret_cast<int*>(handoff + (size_t)blockIdx.x * 128)[threadIdx.x] = acc + launch;
  if (lane == 0) {
    unsigned long long* s = stamps + ((size_t)launch * gridDim.x + blockIdx.x) * 64 + wave * 4;
    s[0] = t0; s[1] = t1; s[2] = t2;
  }
  if (acc == 0x7fffffff) *sink = acc;
}

template <int MODE>
int run(int threads, const char* small, const char* big, char* handoff, unsigned long long* stamps, int* sink, const char* what) {
  const int L = 64, G = 256;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < L; ++i) {
    const size_t off = MODE == 5 ? (size_t)(i / 2) * (16u << 20) : (size_t)i * (16u << 20);   // mode 5: launch 2k+1 re-reads launch 2k's 16 MiB
    probe<MODE><<<G, threads, 0, st>>>(small, big, handoff, off, stamps, i, sink);
  }
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
  const int waves = threads / 64;
  std::vector<unsigned long long> h((size_t)L * G * 64);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double lat = 0, lat_max = 0, ramp = 0, gap = 0, span = 0, lat2 = 0;
  int n = 0;
  unsigned long long prev_end = 0;
  for (int i = 8; i < L; ++i) {
    unsigned long long first = ~0ull, last_start = 0, end = 0;
    double l = 0, lm = 0, l2 = 0;
    for (int b = 0; b < G; ++b)
      for (int w = 0; w < waves; ++w) {
        const unsigned long long* s = &h[((size_t)i * G + b) * 64 + w * 4];
        first = std::min(first, s[0]); last_start = std::max(last_start, s[0]); end = std::max(end, s[2]);
        l += (double)(s[1] - s[0]); lm = std::max(lm, (double)(s[1] - s[0])); l2 += (double)(s[2] - s[1]);
      }
    lat += l / (G * waves); lat_max += lm; lat2 += l2 / (G * waves); ramp += (double)(last_start - first); span += (double)(end - first);
    if (prev_end) gap += (double)(first - prev_end);
    prev_end = end;
    ++n;
  }
  if (MODE == 5) {
    double ev = 0, od = 0;
    for (int i = 8; i < L; ++i) {
      double l2 = 0;
      for (int b = 0; b < G; ++b)
        for (int w = 0; w < waves; ++w) { const unsigned long long* s = &h[((size_t)i * G + b) * 64 + w * 4]; l2 += (double)(s[2] - s[1]); }
      (i & 1 ? od : ev) += l2 / (G * waves);
    }
    printf("   mode 5: stream time first touch (HBM / MALL) %.2f us, re-read by the next launch with the same block mapping (L2) %.2f us\n", ev / ((L - 8) / 2) / 100, od / ((L - 8) / 2) / 100);
  }
  printf("%-58s threads %4d | period %.2f us | entry->data mean %.2f max %.2f us | stream after %.2f | start ramp %.2f | gap %.2f | span %.2f\n", what,
         threads, ms * 1e3 / (10 * L), lat / n / 100, lat_max / n / 100, lat2 / n / 100, ramp / n / 100, gap / (n - 1) / 100, span / n / 100);
  CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
  return 0;
}

int main() {
  char *small, *big, *handoff;
  unsigned long long* stamps;
  int* sink;
  const size_t BIG = (size_t)1536 << 20;
  CK(hipMalloc(&small, 1 << 20)); CK(hipMalloc(&big, BIG)); CK(hipMalloc(&handoff, 1 << 20));
  CK(hipMalloc(&stamps, (size_t)64 * 256 * 64 * 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(small, 1, 1 << 20)); CK(hipMemset(big, 1, BIG)); CK(hipMemset(handoff, 1, 1 << 20));
  CK(hipMemset(stamps, 0, (size_t)64 * 256 * 64 * 8));
  for (int threads : {256, 1024}) {
    if (run<0>(threads, small, big, handoff, stamps, sink, "0: same hot line, every launch / workgroup")) return 1;
    if (run<1>(threads, small, big, handoff, stamps, sink, "1: line per workgroup, small resident buffer")) return 1;
    if (run<2>(threads, small, big, handoff, stamps, sink, "2: line per workgroup from 16 MiB fresh per launch (HBM)")) return 1;
    if (run<3>(threads, small, big, handoff, stamps, sink, "3: resident line, then 64 KiB / workgroup fresh stream")) return 1;
    if (run<4>(threads, small, big, handoff, stamps, sink, "4: line written by the previous launch")) return 1;
    if (run<5>(threads, small, big, handoff, stamps, sink, "5: 64 KiB / workgroup stream, every second launch re-reads")) return 1;
  }
  return 0;
}
