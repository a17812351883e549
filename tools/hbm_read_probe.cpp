// HBM read ceiling for a streaming kernel on this box: 537 MB of fp32 (pv_bmm's x1 at S = 2048) read once by grid-stride dwordx4 loads,
// U independent 16-byte requests per thread in flight, W workgroups of 256 threads per CU.  Prints TB/s per (U, W).
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_read_probe.cpp -o tools/hbm_read_probe && tools/hbm_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U>
__global__ void __launch_bounds__(256) rd(const v4f* __restrict__ x, long long n4, float* out) {
  const long long stride = (long long)gridDim.x * 256 * U;
  v4f acc = {0, 0, 0, 0};
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n4 ? __builtin_nontemporal_load(x + i + u * 256) : (v4f){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1.f;
}
template <int U>
__global__ void __launch_bounds__(256) wr(v4f* __restrict__ x, long long n4) {
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i + u * 256 < n4) x[i + u * 256] = (v4f){1.f, 2.f, 3.f, 4.f};
  }
}
int main() {
  const long long bytes = 32ll * 2048 * 2048 * 4, n4 = bytes / 16;
  v4f* x; float* o;
  hipMalloc(&x, bytes); hipMalloc(&o, 4); hipMemset(x, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    printf("%s %.1f us = %.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
  };
  char nm[64];
  for (int w : {2, 4, 8, 16}) {
    snprintf(nm, 64, "read  U=4  W=%2d", w); run(nm, [&] { rd<4><<<256 * w, 256>>>(x, n4, o); });
    snprintf(nm, 64, "read  U=8  W=%2d", w); run(nm, [&] { rd<8><<<256 * w, 256>>>(x, n4, o); });
    snprintf(nm, 64, "read  U=16 W=%2d", w); run(nm, [&] { rd<16><<<256 * w, 256>>>(x, n4, o); });
  }
  for (int w : {2, 4, 8, 16}) {
    snprintf(nm, 64, "write U=4  W=%2d", w); run(nm, [&] { wr<4><<<256 * w, 256>>>(x, n4); });
    snprintf(nm, 64, "write U=8  W=%2d", w); run(nm, [&] { wr<8><<<256 * w, 256>>>(x, n4); });
  }
  return 0;
}
