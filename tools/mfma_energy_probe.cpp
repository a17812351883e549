// Does the int8 matrix pipe hold a higher clock with v_mfma_i32_32x32x32_i8 than with v_mfma_i32_16x16x64_i8?
//
// the headline GEMM's MFMA floor is 13.3 us at the 1.65-1.69 GHz the chip holds under quantised-Gaussian
// operands (2.23 GHz with zeros).  A 32x32x32 MFMA reads half the operand registers per MAC (1 KiB A + 1 KiB B for 32 K MACs instead
// of 16 K); if register-operand traffic is what the power budget pays for, the same MACs would run at a higher clock.  Building that
// main loop means re-deriving the whole generated kernel (176 = 5.5 x 32 columns, every transposing epilogue), so the hypothesis is
// tested first, in isolation: the production wave layout (256 workgroups x 8 waves, two per SIMD), nothing but MFMAs on
// register-resident operands (or operands re-read from the LDS every k-step, as the production loop does), random int8 data vs zeros.
// Prints per variant: us per launch, TOPS, the shader clock from s_memtime / s_memrealtime.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_energy_probe.cpp -o /tmp/mfma_energy_probe && /tmp/mfma_energy_probe [ksteps] [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Stamp { unsigned long long cycles, ticks; };

__device__ inline unsigned long long memtime() { return __builtin_readcyclecounter(); }
__device__ inline unsigned long long realtime() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }


// LDS reads the compiler may neither hoist out of the k loop nor wait for early: an asm ds_read per fragment, and ONE counted wait that
// names every destination as an in-out operand (so the MFMAs that consume them cannot move above it).
__device__ inline v4i lds_read(unsigned addr) { v4i r; asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr)); return r; }
template <int N> __device__ inline void lds_wait(v4i (&f)[N]);
template <> __device__ inline void lds_wait<11>(v4i (&f)[11]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]));
}
template <> __device__ inline void lds_wait<10>(v4i (&f)[10]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]));
}

// One k-step of 64 over a 32 x 176 wave tile = 22 MFMAs of 16x16x64 (the production tile).  NB W fragments and 2 A fragments per
// k-step; POOL k-steps of distinct operands cycle through (so consecutive MFMAs see different data, as in a real loop).
// ROWS = A fragments per wave: 2 = the production layout (8 waves x 32 rows, two waves per SIMD); 4 = FOUR waves x 64 rows (one wave
// per SIMD, 176 accumulators): every W fragment read from the LDS then feeds four MFMAs instead of two -- half the LDS reads per MAC.
template <int POOL, bool LDS, int ROWS = 2>
__global__ __launch_bounds__(ROWS == 2 ? 512 : 256) void probe16(const v4i* __restrict__ src, int ksteps, int* __restrict__ sink, Stamp* __restrict__ stamps) {
  extern __shared__ v4i lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i a[POOL][ROWS], w[POOL][11];
  if (LDS) {       // W fragments live in the LDS (11 per pool slot, shared by the waves), A in registers
    for (int i = threadIdx.x; i < POOL * 11 * 64; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < POOL; ++p) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) a[p][i] = src[((p * ROWS + i) * 8 + wave) * 64 + lane + 4096];
    if (!LDS) {
#pragma unroll
      for (int j = 0; j < 11; ++j) w[p][j] = src[(p * 11 + j) * 64 + lane];
    }
  }
  v4i acc[ROWS][11];
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
#pragma unroll
    for (int j = 0; j < 11; ++j) acc[i][j] = (v4i){0, 0, 0, 0};
  const unsigned lbase = (unsigned)(lane * 16);
  v4i wf[2][11];
  if (LDS) {
#pragma unroll
    for (int j = 0; j < 11; ++j) wf[0][j] = lds_read(lbase + j * 1024);
    lds_wait<11>(wf[0]);
  }
  const unsigned long long c0 = memtime(), t0 = realtime();
  for (int k = 0; k < ksteps; k += POOL) {
#pragma unroll
    for (int p = 0; p < POOL; ++p) {
      if (LDS) {          // the NEXT k-step's W fragments are read while this one's MFMAs issue (the production loop's pipeline)
#pragma unroll
        for (int j = 0; j < 11; ++j) wf[(p + 1) & 1][j] = lds_read(lbase + (((p + 1) % POOL) * 11 + j) * 1024);
      } else {
#pragma unroll
        for (int j = 0; j < 11; ++j) wf[p & 1][j] = w[p][j];
      }
#pragma unroll
      for (int j = 0; j < 11; ++j)
#pragma unroll
        for (int i = 0; i < ROWS; ++i) acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[p & 1][j], a[p][i], acc[i][j], 0, 0, 0);
      if (LDS) lds_wait<11>(wf[(p + 1) & 1]);
    }
  }
  const unsigned long long c1 = memtime(), t1 = realtime();
  int x = 0;
#pragma unroll
  for (int i = 0; i < ROWS; ++i)
#pragma unroll
    for (int j = 0; j < 11; ++j) x ^= acc[i][j][0] ^ acc[i][j][1] ^ acc[i][j][2] ^ acc[i][j][3];
  if (x == 0x5a5a5a5a) sink[0] = x;
  if (lane == 0) {
    stamps[blockIdx.x * 8 + wave] = Stamp{c1 - c0, t1 - t0};
    if (ROWS == 4) stamps[blockIdx.x * 8 + wave + 4] = Stamp{c1 - c0, t1 - t0};       // (the host averages 8 slots per workgroup)
  }
}

// The same MAC count per k-step on 32x32x32: a 32 x 160 wave tile = 5 column blocks x 2 k-halves = 10 MFMAs of 32 K MACs (+ the
// sixth half block is left out: 327 680 MACs per k-step against 360 448 -- TOPS are normalised by the MACs actually issued).
template <int POOL, bool LDS>
__global__ __launch_bounds__(512) void probe32(const v4i* __restrict__ src, int ksteps, int* __restrict__ sink, Stamp* __restrict__ stamps) {
  extern __shared__ v4i lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4i a[POOL][2], w[POOL][10];
  if (LDS) {
    for (int i = threadIdx.x; i < POOL * 10 * 64; i += 512) lds[i] = src[i];
    __syncthreads();
  }
#pragma unroll
  for (int p = 0; p < POOL; ++p) {
#pragma unroll
    for (int i = 0; i < 2; ++i) a[p][i] = src[((p * 2 + i) * 8 + wave) * 64 + lane + 4096];
    if (!LDS) {
#pragma unroll
      for (int j = 0; j < 10; ++j) w[p][j] = src[(p * 10 + j) * 64 + lane];
    }
  }
  v16i acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0;
  const unsigned lbase = (unsigned)(lane * 16);
  v4i wf[2][10];
  if (LDS) {
#pragma unroll
    for (int j = 0; j < 10; ++j) wf[0][j] = lds_read(lbase + j * 1024);
    lds_wait<10>(wf[0]);
  }
  const unsigned long long c0 = memtime(), t0 = realtime();
  for (int k = 0; k < ksteps; k += POOL) {
#pragma unroll
    for (int p = 0; p < POOL; ++p) {
      if (LDS) {
#pragma unroll
        for (int j = 0; j < 10; ++j) wf[(p + 1) & 1][j] = lds_read(lbase + (((p + 1) % POOL) * 10 + j) * 1024);
      } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) wf[p & 1][j] = w[p][j];
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[p & 1][2 * j + h], a[p][h], acc[j], 0, 0, 0);
      if (LDS) lds_wait<10>(wf[(p + 1) & 1]);
    }
  }
  const unsigned long long c1 = memtime(), t1 = realtime();
  int x = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) x ^= acc[j][e];
  if (x == 0x5a5a5a5a) sink[0] = x;
  if (lane == 0) stamps[blockIdx.x * 8 + wave] = Stamp{c1 - c0, t1 - t0};
}

template <typename K>
static void run(const char* name, K kernel, double macs_per_kstep_wave, const v4i* src, int ksteps, int reps, int* sink, Stamp* stamps, size_t lds_bytes,
                int threads = 512) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), lds_bytes, 0, src, ksteps, sink, stamps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(256), dim3(threads), lds_bytes, 0, src, ksteps, sink, stamps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<Stamp> h(2048);
  CK(hipMemcpy(h.data(), stamps, sizeof(Stamp) * 2048, hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  double cyc = 0;
  for (auto& s : h) { mhz.push_back(double(s.cycles) / double(s.ticks) * 100.0); cyc += double(s.cycles); }
  std::sort(mhz.begin(), mhz.end());
  const double us = ms * 1e3 / reps;
  const double ops = 2.0 * macs_per_kstep_wave * ksteps * (threads / 64) * 256;
  const double loop_us = (cyc / 2048) / (mhz[1024] * 1e6) * 1e6;
  printf("%-42s %8.2f us/launch  %7.1f TOPS (launch)  %7.1f TOPS (mean wave window: the older wave of a SIMD gets the pipe first)  clock median %6.0f MHz (min %6.0f max %6.0f)  cycles/kstep %7.1f\n", name, us,
         ops / us * 1e-6, ops / loop_us * 1e-6, mhz[1024], mhz.front(), mhz.back(), cyc / 2048 / ksteps);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const int ksteps = argc > 1 ? atoi(argv[1]) : 256;       // multiples of POOL (4); 32 = the headline K = 2048
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  constexpr int POOL = 4, RPOOL = 2;      // register-resident operands: 2 x 13 fragments + 88 accumulators fit 256 registers
  const size_t n = 1 << 16;            // v4i elements
  std::vector<v4i> h(n);
  v4i *rnd, *gauss, *zero;
  int* sink;
  Stamp* stamps;
  CK(hipMalloc(&rnd, n * 16)); CK(hipMalloc(&gauss, n * 16)); CK(hipMalloc(&zero, n * 16)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, sizeof(Stamp) * 2048));
  srand(1);
  for (auto& v : h) for (int e = 0; e < 4; ++e) v[e] = (rand() & 0xffff) | (rand() << 16);
  CK(hipMemcpy(rnd, h.data(), n * 16, hipMemcpyHostToDevice));
  // quantised-Gaussian int8 (what the bench feeds: index - 128 of N(0,1) on a +-5 sigma grid -> stored values ~ N(0, 25))
  for (auto& v : h) for (int e = 0; e < 4; ++e) {
    unsigned word = 0;
    for (int b = 0; b < 4; ++b) {
      double s = 0;
      for (int i = 0; i < 12; ++i) s += rand() / (double)RAND_MAX;
      int q = (int)lrint((s - 6.0) * 25.0);
      q = q < -128 ? -128 : q > 127 ? 127 : q;
      word |= (unsigned)(q & 0xff) << (8 * b);
    }
    v[e] = (int)word;
  }
  CK(hipMemcpy(gauss, h.data(), n * 16, hipMemcpyHostToDevice));
  CK(hipMemset(zero, 0, n * 16));
  const size_t lds16 = POOL * 11 * 64 * 16, lds32 = POOL * 10 * 64 * 16;
  for (int round = 0; round < 2; ++round) {
    printf("-- round %d, %d k-steps of 64 per wave, 256 workgroups x 8 waves\n", round, ksteps);
    run("16x16x64 regs   uniform-random", probe16<RPOOL, false>, 22 * 16384.0, rnd, ksteps, reps, sink, stamps, 0);
    run("32x32x32 regs   uniform-random", probe32<RPOOL, false>, 10 * 32768.0, rnd, ksteps, reps, sink, stamps, 0);
    run("16x16x64 regs   gaussian", probe16<RPOOL, false>, 22 * 16384.0, gauss, ksteps, reps, sink, stamps, 0);
    run("32x32x32 regs   gaussian", probe32<RPOOL, false>, 10 * 32768.0, gauss, ksteps, reps, sink, stamps, 0);
    run("16x16x64 regs   zeros", probe16<RPOOL, false>, 22 * 16384.0, zero, ksteps, reps, sink, stamps, 0);
    run("32x32x32 regs   zeros", probe32<RPOOL, false>, 10 * 32768.0, zero, ksteps, reps, sink, stamps, 0);
    run("16x16x64 W from LDS gaussian", probe16<POOL, true>, 22 * 16384.0, gauss, ksteps, reps, sink, stamps, lds16);
    run("32x32x32 W from LDS gaussian", probe32<POOL, true>, 10 * 32768.0, gauss, ksteps, reps, sink, stamps, lds32);
    run("16x16x64 W from LDS zeros", probe16<POOL, true>, 22 * 16384.0, zero, ksteps, reps, sink, stamps, lds16);
    run("16x16x64 4 waves x 64 rows, LDS, gaussian", probe16<POOL, true, 4>, 44 * 16384.0, gauss, ksteps, reps, sink, stamps, lds16, 256);
    run("16x16x64 4 waves x 64 rows, LDS, zeros", probe16<POOL, true, 4>, 44 * 16384.0, zero, ksteps, reps, sink, stamps, lds16, 256);
    run("32x32x32 W from LDS zeros", probe32<POOL, true>, 10 * 32768.0, zero, ksteps, reps, sink, stamps, lds32);
  }
  return 0;
}
