// Cost of a software grid barrier on MI355X (one 1024-thread workgroup per CU, all resident): the question behind a persistent
// per-token decode kernel.  hipcc --offload-arch=gfx950 -O2 tools/barrier_probe.cpp -o tools/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) barrier_kernel(unsigned* counter, unsigned* flag, int rounds, float* data, long long* cycles) {
  const unsigned nb = gridDim.x;
  unsigned epoch = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    if (MODE >= 1) {                       // every block publishes a value the others read after the barrier
      if (threadIdx.x == 0) data[(r & 1) * 4096 + blockIdx.x] = (float)(r + blockIdx.x);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      epoch += nb;
      if (MODE >= 1) __threadfence();
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *flag = 1; break; }
      }
      if (MODE >= 1) __threadfence();
    }
    __syncthreads();
    if (MODE >= 1) {
      const float v = data[(r & 1) * 4096 + ((blockIdx.x + 1) % nb)];
      if (v != (float)(r + (blockIdx.x + 1) % nb) && threadIdx.x == 0) *flag = 2;
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = __builtin_readcyclecounter() - t0;
}

// Hierarchical variant: workgroup w lives on XCD w % 8 (round-robin dispatch).  Level 1: the 32 workgroups of an XCD meet on their own
// counter (own cache line); level 2: one leader per XCD meets the other 7 on a global counter and then releases its XCD through a
// per-XCD epoch word.  Same-address traffic drops from 256 to 32 + 8 requesters.
template <int MODE>
__global__ void __launch_bounds__(1024) hbarrier_kernel(unsigned* ctr /* [8][32] per-XCD arrive, [8][32] per-XCD release, [32] global */, unsigned* flag,
                                                        int rounds, float* data) {
  const unsigned nb = gridDim.x, xcd = blockIdx.x & 7, per = nb >> 3;
  unsigned* arrive = ctr + xcd * 32;
  unsigned* release = ctr + 8 * 32 + xcd * 32;
  unsigned* global = ctr + 16 * 32;
  for (int r = 0; r < rounds; ++r) {
    if (MODE >= 1 && threadIdx.x == 0) data[(r & 1) * 4096 + blockIdx.x] = (float)(r + blockIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE >= 1) __threadfence();
      const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      if (old == (unsigned)(r + 1) * per - 1) {                 // last arriver of this XCD: go to the global level
        __hip_atomic_fetch_add(global, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(global, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) * 8) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *flag = 1; break; }
        }
        __hip_atomic_store(release, (unsigned)(r + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(release, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 22)) { *flag = 1; break; }
        }
      }
      if (MODE >= 1) __threadfence();
    }
    __syncthreads();
    if (MODE >= 1) {
      const float v = data[(r & 1) * 4096 + ((blockIdx.x + 1) % nb)];
      if (v != (float)(r + (blockIdx.x + 1) % nb) && threadIdx.x == 0) *flag = 2;
    }
  }
}


// The recipe of MI355X_MICROARCH.md ("barrier-xcd"): the XCD is read from HW_REG_XCC_ID (not assumed from the block id), arrivals and
// polls are RELAXED agent-scope accesses, ordering comes from ONE release fence before the arrive and ONE acquire fence after the
// wait (an acquire load per poll iteration invalidates the caches every time round), the last arriver of an XCD goes to the top
// counter and then publishes the XCD's generation word.
template <int MODE>
__global__ void xbarrier_kernel(unsigned* ctr /* [8][32] arrive, [8][32] generation, [32] top */, unsigned* flag, int rounds, float* data) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const unsigned nb = gridDim.x, per = nb >> 3;
  unsigned* arrive = ctr + xcc * 32;
  unsigned* gen = ctr + 8 * 32 + xcc * 32;
  unsigned* top = ctr + 16 * 32;
  for (int r = 0; r < rounds; ++r) {
    if (MODE >= 1 && threadIdx.x == 0) data[(r & 1) * 4096 + blockIdx.x] = (float)(r + blockIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      if (old == (unsigned)(r + 1) * per - 1) {
        __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1) * 8) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { *flag = 1; return; }     // a wrong XCD population: give up instead of hanging
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(gen, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > (1 << 20)) { *flag = 1; return; }     // a wrong XCD population: give up instead of hanging
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
    if (MODE >= 1) {
      const float v = __hip_atomic_load(data + (r & 1) * 4096 + ((blockIdx.x + 1) % nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v != (float)(r + (blockIdx.x + 1) % nb) && threadIdx.x == 0) *flag = 2;
    }
  }
}

int main(int argc, char** argv) {
  int rounds = argc > 1 ? atoi(argv[1]) : 1000;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned *counter, *flag;
  float* data;
  long long* cyc;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&data, 2 * 4096 * 4)); CK(hipMalloc(&cyc, 8));
  for (int mode = 0; mode < 2; ++mode) {
    for (int blocks : {cus, cus / 2, 64}) {
      CK(hipMemset(counter, 0, 4)); CK(hipMemset(flag, 0, 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (mode == 0) barrier_kernel<0><<<blocks, 1024>>>(counter, flag, rounds, data, cyc);
      else barrier_kernel<1><<<blocks, 1024>>>(counter, flag, rounds, data, cyc);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned f;
      CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
      printf("mode %d (%s) blocks %3d x 1024 threads: %.3f us per barrier (flag %u)\n", mode, mode ? "publish + fences + check" : "counter only", blocks,
             ms * 1e3 / rounds, f);
    }
  }
  unsigned* hctr;
  CK(hipMalloc(&hctr, 17 * 32 * 4));
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(hctr, 0, 17 * 32 * 4)); CK(hipMemset(flag, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    if (mode == 0) hbarrier_kernel<0><<<cus, 1024>>>(hctr, flag, rounds, data);
    else hbarrier_kernel<1><<<cus, 1024>>>(hctr, flag, rounds, data);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f;
    CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
    printf("hierarchical (per-XCD, then global) mode %d (%s) blocks %3d x 1024 threads: %.3f us per barrier (flag %u)\n", mode,
           mode ? "publish + fences + check" : "counters only", cus, ms * 1e3 / rounds, f);
  }
  for (int threads : {1024, 512, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipMemset(hctr, 0, 17 * 32 * 4)); CK(hipMemset(flag, 0, 4));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      if (mode == 0) xbarrier_kernel<0><<<cus, threads>>>(hctr, flag, rounds, data);
      else xbarrier_kernel<1><<<cus, threads>>>(hctr, flag, rounds, data);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned f;
      CK(hipMemcpy(&f, flag, 4, hipMemcpyDeviceToHost));
      printf("guide recipe (HW_REG_XCC_ID, relaxed polls, one fence each side) mode %d (%s) blocks %3d x %4d threads: %.3f us per barrier (flag %u)\n", mode,
             mode ? "publish + check" : "counters only", cus, threads, ms * 1e3 / rounds, f);
    }
  }
  return 0;
}
