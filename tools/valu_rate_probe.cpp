// Issue cost of the VALU instructions of the attention kernel's quantizer chain on gfx950, in shader cycles per wave-instruction:
// v_fma_f32, v_pk_fma_f32 (two lanes' worth per instruction), v_exp_f32, v_med3_f32, v_add_f32, v_pk_add_f32 -- 16 independent
// chains per wave so that dependent-issue latency does not show, one or two waves per SIMD (256 / 512 threads per workgroup, one
// workgroup per CU).  Decides whether the packed fp32 forms pay in a VALU-bound loop (MI355X_MICROARCH.md calls them an anti-lever
// beside MFMAs).     hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.cpp -o /tmp/valu_rate_probe && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void probe(float* out, unsigned long long* cyc, int iters, float a, float b) {
  float x[16];
  v2f p[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = v2f{x[2 * i], x[2 * i + 1]};
  const v2f a2 = {a, a}, b2 = {b, b};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (OP == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      } else if (OP == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
      } else if (OP == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      } else if (OP == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
      } else if (OP == 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
      } else if (OP == 5) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
      } else if (OP == 6) {   // the chain's mix per score pair: pk_fma, 2 med3, pk_fma, 2 exp, pk_add
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
          asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(p[i].x) : "v"(a), "v"(b));
          asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(p[i].y) : "v"(a), "v"(b));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i + 2]) : "v"(a2), "v"(b2));
          asm volatile("v_exp_f32 %0, %0" : "+v"(p[i + 4].x));
          asm volatile("v_exp_f32 %0, %0" : "+v"(p[i + 4].y));
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i + 6]) : "v"(a2));
        }
      } else {                 // the same work in scalar form: 2 fma, 2 med3, 2 fma, 2 exp, 2 add
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i + 2]) : "v"(a), "v"(b));
          asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i + 4]) : "v"(a), "v"(b));
          asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i + 6]) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i + 8]) : "v"(a), "v"(b));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i + 10]) : "v"(a), "v"(b));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i + 12]));
          asm volatile("v_exp_f32 %0, %0" : "+v"(x[i + 14]));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i + 2]) : "v"(a));
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int insts_per_round, int threads) {
  const int blocks = 256, iters = 2000;
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(float) * blocks * threads));
  CK(hipMalloc(&cyc, 8 * blocks * (threads / 64)));
  for (int w = 0; w < 2; ++w) probe<OP><<<blocks, threads>>>(out, cyc, iters, 1.0001f, 0.5f);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * (threads / 64));
  CK(hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double per = (double)h[h.size() / 2] / ((double)iters * 4 * insts_per_round);
  printf("%-34s %d waves/SIMD: %.2f cycles per instruction and wave (%.2f per SIMD-issue)\n", name, threads / 256, per, per / (threads / 256));
  CK(hipFree(out));
  CK(hipFree(cyc));
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("v_fma_f32", 16, threads);
    run<1>("v_pk_fma_f32", 8, threads);
    run<2>("v_exp_f32", 16, threads);
    run<3>("v_med3_f32", 16, threads);
    run<4>("v_add_f32", 16, threads);
    run<5>("v_pk_add_f32", 8, threads);
    run<6>("chain, packed (14 inst / 4 scores)", 14, threads);
    run<7>("chain, scalar (20 inst / 4 scores)", 20, threads);
  }
  return 0;
}
