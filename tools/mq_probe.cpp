// Standalone probe for libmobilequant_amd.so on a GPU box (no torch): checks the int8 GEMM against a
// host integer reference and times every tile variant with HIP events.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/mq_probe.cpp -Lmobilequant_amd/lib -lmobilequant_amd \
//         -Wl,-rpath,'$ORIGIN/../mobilequant_amd/lib' -o tools/mq_probe
//   tools/mq_probe [M N K] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>
#include <random>
#include <vector>

#include "mobilequant_amd.h"
#include "mobilequant_amd_tuning.h"
extern "C" int mq_gemm_set_debug_buffer_(void*) __attribute__((weak));   // ablation builds of the library only

#define HIPCHK(x)                                                                   \
  do {                                                                              \
    hipError_t e = (x);                                                             \
    if (e != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)
#define MQCHK(x)                                                                 \
  do {                                                                           \
    int r = (x);                                                                 \
    if (r != MQ_OK) {                                                            \
      fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, r, mq_last_error()); \
      exit(3);                                                                   \
    }                                                                            \
  } while (0)

template <typename T>
T* dmalloc(size_t n) {
  T* p;
  HIPCHK(hipMalloc(&p, n * sizeof(T)));
  return p;
}
template <typename T>
T* upload(const std::vector<T>& h) {
  T* p = dmalloc<T>(h.size());
  HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

struct Problem {
  int M, N, K;
  std::vector<int8_t> a, w;
  std::vector<int32_t> rs, zw, ct;
  std::vector<float> alpha, bias;
  int8_t *da, *dw, *dat = nullptr;   // dat: fragment-blocked copy of a (mq_quantize_tiled layout), made on demand
  int32_t *drs, *dzw, *dct;
  float *dalpha, *dbias, *dos, *doo;
  float os, oo;
};

static Problem make_problem(int M, int N, int K, unsigned seed, bool per_row) {
  Problem p;
  p.M = M; p.N = N; p.K = K;
  std::mt19937 g(seed);
  std::uniform_int_distribution<int> d8(-128, 127);
  p.a.resize((size_t)M * K);
  p.w.resize((size_t)N * K);
  // MQ_PROBE_FILL = zero | small: DVFS what-if (the chip clocks to its power budget, and operand toggling is power); results are
  // still checked against the host reference
  const char* fill = getenv("MQ_PROBE_FILL");
  std::uniform_int_distribution<int> d4(-8, 7);
  for (auto& v : p.a) v = (int8_t)(!fill ? d8(g) : !strcmp(fill, "zero") ? 0 : d4(g));
  for (auto& v : p.w) v = (int8_t)(!fill ? d8(g) : !strcmp(fill, "zero") ? 0 : d4(g));
  p.rs.resize(M);
  for (int m = 0; m < M; ++m) {
    int s = 0;
    for (int k = 0; k < K; ++k) s += p.a[(size_t)m * K + k];
    p.rs[m] = s;
  }
  const int za = 7;   // stored-domain activation zero point
  p.zw.resize(N); p.ct.resize(N); p.alpha.resize(N); p.bias.resize(N);
  std::uniform_real_distribution<float> uf(0.5f, 1.5f);
  for (int n = 0; n < N; ++n) {
    int cs = 0;
    for (int k = 0; k < K; ++k) cs += p.w[(size_t)n * K + k];
    p.zw[n] = per_row ? (n % 31) - 15 : -9;
    p.ct[n] = -za * cs + K * za * p.zw[n];
    p.alpha[n] = 1e-4f * (per_row ? uf(g) : 1.0f);
    p.bias[n] = 0.01f * (float)((n % 13) - 6);
  }
  p.da = upload(p.a); p.dw = upload(p.w); p.drs = upload(p.rs); p.dzw = upload(p.zw); p.dct = upload(p.ct);
  p.dalpha = upload(p.alpha); p.dbias = upload(p.bias);
  p.os = 0.05f; p.oo = 131.f;
  std::vector<float> s1{p.os}, o1{p.oo};
  p.dos = upload(s1); p.doo = upload(o1);
  return p;
}

// fragment-blocked layout of include/mobilequant_amd.h: 1-KiB blocks of 16 rows x 64 k ordered [row block][k block];
// inside a block byte offset 16 * ((row & 15) + 16 * ((k & 63) >> 4)) + (k & 15)
static void make_tiled(Problem& p) {
  if (p.dat) return;
  const int Mp = (p.M + 15) / 16 * 16, K = p.K;
  std::vector<int8_t> t((size_t)Mp * K, 0);
  for (int m = 0; m < p.M; ++m)
    for (int k = 0; k < K; ++k) {
      const size_t blk = ((size_t)(m >> 4) * (K >> 6) + (k >> 6)) * 1024;
      t[blk + 16 * ((m & 15) + 16 * ((k & 63) >> 4)) + (k & 15)] = p.a[(size_t)m * K + k];
    }
  p.dat = upload(t);
}

static int run_linear(Problem& p, int variant, const float* os, const float* oo, float qmax, void* dout, int out_dtype) {
  if (variant == 9 || variant == 11) {       // generated-ISA kernels: fragment-blocked activations through the tiled entry point
    make_tiled(p);
    return mq_w8a8_linear_tiled(p.dat, p.dw, p.M, p.N, p.K, p.drs, p.dalpha, p.dzw, p.dct, p.dbias, os, oo, 0.f, qmax, dout,
                                out_dtype, nullptr);
  }
  return mq_w8a8_linear(p.da, p.dw, p.M, p.N, p.K, p.drs, p.dalpha, p.dzw, p.dct, p.dbias, os, oo, 0.f, qmax, dout, out_dtype,
                        nullptr);
}

static void free_problem(Problem& p) {
  if (p.dat) hipFree(p.dat);
  hipFree(p.da); hipFree(p.dw); hipFree(p.drs); hipFree(p.dzw); hipFree(p.dct); hipFree(p.dalpha);
  hipFree(p.dbias); hipFree(p.dos); hipFree(p.doo);
}

// host reference for row m: returns pre-quant float values
static void ref_row(const Problem& p, int m, std::vector<float>& out, std::vector<int>& acc_out) {
  out.resize(p.N); acc_out.resize(p.N);
  for (int n = 0; n < p.N; ++n) {
    int64_t acc = 0;
    const int8_t* ar = &p.a[(size_t)m * p.K];
    const int8_t* wr = &p.w[(size_t)n * p.K];
    for (int k = 0; k < p.K; ++k) acc += (int)ar[k] * (int)wr[k];
    int t = (int)(acc - (int64_t)p.zw[n] * p.rs[m] + p.ct[n]);
    acc_out[n] = t;
    out[n] = (float)t * p.alpha[n] + p.bias[n];
  }
}

static int check(Problem& p, int variant, int out_dtype, bool outq) {
  const size_t esz = out_dtype == MQ_F32 ? 4 : (out_dtype == MQ_F16 || out_dtype == MQ_U16 || out_dtype == MQ_I16) ? 2 : 1;
  void* dout;
  HIPCHK(hipMalloc(&dout, (size_t)p.M * p.N * esz));
  HIPCHK(hipMemset(dout, 0xCD, (size_t)p.M * p.N * esz));
  mq_gemm_set_variant(variant);
  if ((variant == 9 || variant == 11) && !mq_gemm_tiled_supported(p.M, p.N, p.K)) { hipFree(dout); return 0; }   // shape not served
  MQCHK(run_linear(p, variant, outq ? p.dos : nullptr, outq ? p.doo : nullptr, out_dtype == MQ_U16 ? 65535.f : 255.f, dout, out_dtype));
  HIPCHK(hipDeviceSynchronize());
  std::vector<uint8_t> h((size_t)p.M * p.N * esz);
  HIPCHK(hipMemcpy(h.data(), dout, h.size(), hipMemcpyDeviceToHost));
  hipFree(dout);
  // rows to check: all if small, else a spread incl. first/last of tiles
  std::vector<int> rows;
  if (p.M <= 64) for (int m = 0; m < p.M; ++m) rows.push_back(m);
  else for (int i = 0; i < 24; ++i) rows.push_back((int)(((int64_t)i * 2654435761u + 17) % p.M));
  rows.push_back(0); rows.push_back(p.M - 1);
  int bad = 0;
  std::vector<float> ref; std::vector<int> racc;
  const float qmax = out_dtype == MQ_U16 ? 65535.f : 255.f;
  for (int m : rows) {
    ref_row(p, m, ref, racc);
    for (int n = 0; n < p.N; ++n) {
      float got, want = ref[n];
      size_t idx = (size_t)m * p.N + n;
      if (outq) {
        float q = rintf(want / p.os) + p.oo;
        q = std::min(std::max(q, 0.f), qmax);
        if (out_dtype == MQ_F32) { got = ((float*)h.data())[idx]; want = (q - p.oo) * p.os; }
        else if (out_dtype == MQ_U8) { got = (float)h[idx]; want = q; }
        else if (out_dtype == MQ_I8) { got = (float)((int8_t*)h.data())[idx] + 128.f; want = q; }
        else if (out_dtype == MQ_U16) { got = (float)((uint16_t*)h.data())[idx]; want = q; }
        else { got = want; }
        float tol = (out_dtype == MQ_F32) ? p.os * 1.001f : 1.0f;   // <= 1 LSB (recip-multiply rounding)
        if (fabsf(got - want) > tol) { if (bad < 5) fprintf(stderr, "  mismatch m=%d n=%d got=%g want=%g\n", m, n, got, want); ++bad; }
      } else {
        got = out_dtype == MQ_F32 ? ((float*)h.data())[idx] : 0.f;
        if (out_dtype == MQ_F32 && got != want) { if (bad < 5) fprintf(stderr, "  mismatch m=%d n=%d got=%g want=%g\n", m, n, got, want); ++bad; }
      }
    }
  }
  return bad;
}

static float time_variant(Problem& p, int variant, int out_dtype, bool outq, int iters) {
  const size_t esz = out_dtype == MQ_F32 ? 4 : (out_dtype == MQ_F16 || out_dtype == MQ_U16) ? 2 : 1;
  void* dout;
  HIPCHK(hipMalloc(&dout, (size_t)p.M * p.N * esz));
  mq_gemm_set_variant(variant);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  if ((variant == 9 || variant == 11) && !mq_gemm_tiled_supported(p.M, p.N, p.K)) return 0.f;
  auto run = [&]() { MQCHK(run_linear(p, variant, outq ? p.dos : nullptr, outq ? p.doo : nullptr, 255.f, dout, out_dtype)); };
  for (int i = 0; i < 5; ++i) run();
  HIPCHK(hipDeviceSynchronize());
  float best = 1e30f, tot = 0;
  const int reps = 5;
  for (int r = 0; r < reps; ++r) {
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) run();
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    best = std::min(best, ms);
    tot += ms;
  }
  hipFree(dout);
  hipEventDestroy(e0); hipEventDestroy(e1);
  (void)tot;
  return best;
}

int main(int argc, char** argv) {
  int cu = 0, khz = 0;
  char arch[64] = "";
  MQCHK(mq_device_info(&cu, &khz, arch, sizeof(arch)));
  printf("device: %s, %d CUs, %d kHz; lib version %d\n", arch, cu, khz, mq_version());
  const int nvar = mq_gemm_set_variant(-1);
  int iters = 50;
  if (argc >= 2 && !strcmp(argv[1], "prof")) {
    // mq_probe prof <variant> <dbg> <launches> [M N K] [out_dtype]: a few launches of one configuration for rocprofv3
    const int v = argc > 2 ? atoi(argv[2]) : 1, dbg = argc > 3 ? atoi(argv[3]) : 0, n = argc > 4 ? atoi(argv[4]) : 20;
    const int M = argc > 7 ? atoi(argv[5]) : 2048, N = argc > 7 ? atoi(argv[6]) : 5632, K = argc > 7 ? atoi(argv[7]) : 2048;
    const int od = argc > 8 ? atoi(argv[8]) : MQ_U8;
    Problem p = make_problem(M, N, K, 99u, false);
    const int nb0 = ((M + 255) / 256) * ((N + 175) / 176);
    unsigned long long* d0 = nullptr;
    if ((dbg & 16) && mq_gemm_set_debug_buffer_) {      // stamp builds write unconditionally: the buffer must exist before the first launch
      d0 = dmalloc<unsigned long long>((size_t)nb0 * 8 * 16);
      HIPCHK(hipMemset(d0, 0, (size_t)nb0 * 8 * 16 * 8));
      mq_gemm_set_debug_buffer_(d0);
    }
    if ((dbg & ~16) == 0) printf("prof check %s: bad=%d\n", mq_gemm_variant_name(v), check(p, v, od, od != MQ_F32 && od != MQ_F16));
    mq_gemm_set_debug(dbg);
    float t = time_variant(p, v, od, od != MQ_F32 && od != MQ_F16, n);
    printf("prof %s dbg=%d %dx%dx%d od=%d: %.2f us\n", mq_gemm_variant_name(v), dbg, M, N, K, od, t * 1e3);
    if ((dbg & 16) && mq_gemm_set_debug_buffer_) {   // s_memtime stamps: [block][wave][t0, loop start, loop end, end]
      const int nb = ((M + 255) / 256) * ((N + 175) / 176), nw = 8;
      unsigned long long* d = d0;
      time_variant(p, v, od, true, 3);
      std::vector<unsigned long long> h((size_t)nb * nw * 16);
      HIPCHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
      double pro = 0, loop = 0, epi = 0, tot = 0; unsigned long long gmin = ~0ull, gmax = 0;
      for (int i = 0; i < nb * nw; ++i) {
        const unsigned long long* q = &h[(size_t)i * 16];
        pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]); tot += (double)(q[3] - q[0]);
        gmin = std::min(gmin, q[0]); gmax = std::max(gmax, q[3]);
      }
      const double n2 = nb * nw;
      printf("stamps (cycles, mean over %d waves): prologue %.0f  loop %.0f (%.0f per K=128 stage)  epilogue %.0f  total %.0f; first start -> last end %llu\n",
             nb * nw, pro / n2, loop / n2, loop / n2 / (K / 128), epi / n2, tot / n2, gmax - gmin);
      if (v == 11) {   // free-running kernel: d[4], d[5] = s_memrealtime (100 MHz, chip-wide) at wave start / end; distribution of the segments
        std::vector<double> tot, pr, lp, ep;
        unsigned long long r0 = ~0ull, r1 = 0; double ticks = 0, rts = 0;
        for (int i = 0; i < nb * nw; ++i) {
          const unsigned long long* q = &h[(size_t)i * 16];
          tot.push_back((double)(q[3] - q[0])); pr.push_back((double)(q[1] - q[0])); lp.push_back((double)(q[2] - q[1])); ep.push_back((double)(q[3] - q[2]));
          r0 = std::min(r0, q[4]); r1 = std::max(r1, q[5]); ticks += (double)(q[3] - q[0]); rts += (double)(q[5] - q[4]);
        }
        auto pct = [](std::vector<double> x, double p) { std::sort(x.begin(), x.end()); return x[(size_t)(p * (x.size() - 1))]; };
        printf("realtime: waves alive from first start to last end %.2f us (one launch); shader clock = %.0f MHz (memtime ticks / realtime)\n",
               (double)(r1 - r0) / 100.0, ticks / rts * 100.0);
        for (auto& pr_ : {std::make_pair("total", &tot), std::make_pair("prologue", &pr), std::make_pair("loop", &lp), std::make_pair("epilogue", &ep)})
          printf("  %-9s min %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f\n", pr_.first, pct(*pr_.second, 0), pct(*pr_.second, 0.5), pct(*pr_.second, 0.9),
                 pct(*pr_.second, 0.99), pct(*pr_.second, 1.0));
        // start skew: when does each workgroup start relative to the first one (realtime)
        std::vector<double> st;
        for (int b = 0; b < nb; ++b) st.push_back((double)(h[(size_t)b * nw * 16 + 4] - r0) / 100.0);
        printf("  workgroup start after the first (us): p50 %.2f  p90 %.2f  max %.2f\n", pct(st, 0.5), pct(st, 0.9), pct(st, 1.0));
      }
      // per-group breakdown for block 0
      for (int b : {0, 100}) for (int w = 0; w < nw; ++w) { const unsigned long long* q = &h[((size_t)b * nw + w) * 16]; printf("  blk%d wave%d: pro %llu loop %llu epi %llu | stage8: %llu %llu %llu %llu\n", b, w, q[1]-q[0], q[2]-q[1], q[3]-q[2], q[5]-q[4], q[6]-q[5], q[7]-q[6], q[8]-q[7]); }
      mq_gemm_set_debug_buffer_(nullptr);
    }
    return 0;
  }
  // ---- correctness: small odd shapes on every variant, then the headline shape ---------------------
  int total_bad = 0;
  {
    struct S { int M, N, K; } shapes[] = {{64, 64, 128}, {100, 180, 256}, {300, 352, 384}, {257, 260, 128}};
    for (auto s : shapes) {
      Problem p = make_problem(s.M, s.N, s.K, 1234u + s.M, true);
      for (int v = 0; v < nvar; ++v) {
        int b0 = check(p, v, MQ_F32, false), b1 = check(p, v, MQ_U8, true), b2 = check(p, v, MQ_F32, true);
        int b3 = check(p, v, MQ_U16, true), b4 = check(p, v, MQ_I8, true);
        printf("check %4dx%4dx%4d %-16s f32:%d u8q:%d f32q:%d u16q:%d i8q:%d\n", s.M, s.N, s.K, mq_gemm_variant_name(v), b0, b1, b2, b3, b4);
        total_bad += b0 + b1 + b2 + b3 + b4;
      }
      free_problem(p);
    }
  }
  struct S { int M, N, K; };
  std::vector<S> shapes = {{2048, 5632, 2048}};
  if (argc >= 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3])}};
  if (argc >= 5) iters = atoi(argv[4]);
  for (auto s : shapes) {
    Problem p = make_problem(s.M, s.N, s.K, 99u, false);
    const double ops = 2.0 * s.M * (double)s.N * s.K;
    for (int v = 0; v < nvar; ++v) {
      int bad = check(p, v, MQ_F32, false) + check(p, v, MQ_U8, true);
      total_bad += bad;
      float t32 = time_variant(p, v, MQ_F32, false, iters);
      float t16 = time_variant(p, v, MQ_F16, false, iters);
      float tq8 = time_variant(p, v, MQ_U8, true, iters);
      float tqf = time_variant(p, v, MQ_F32, true, iters);
      printf("time %4dx%4dx%4d %-16s bad=%d  f32 %.2f us %.0f TOPS | f16 %.2f us %.0f | u8q %.2f us %.0f | f32q %.2f us %.0f\n", s.M, s.N, s.K,
             mq_gemm_variant_name(v), bad, t32 * 1e3, ops / t32 / 1e9, t16 * 1e3, ops / t16 / 1e9, tq8 * 1e3, ops / tq8 / 1e9, tqf * 1e3,
             ops / tqf / 1e9);
      fflush(stdout);
    }
    free_problem(p);
  }
  // ---- ablations on the headline shape (results invalid, timing only) ---------------------------------
  {
    Problem p = make_problem(2048, 5632, 2048, 99u, false);
    const double ops = 2.0 * 2048 * 5632.0 * 2048;
    const char* names[] = {"full", "noDMA", "noMFMA", "noDMA+noMFMA", "noEpi", "noDMA+noEpi", "noMFMA+noEpi", "none", "noREAD", "noREAD+noDMA"};
    for (int v = 0; v < nvar; ++v)
      for (int dbg = 0; dbg < 10; ++dbg) {
        if (dbg >= 8 && v != 7) continue;
        mq_gemm_set_debug(dbg);
        float t = time_variant(p, v, MQ_U8, true, iters);
        printf("ablate %-16s %-14s %.2f us (%.0f TOPS-equivalent)\n", mq_gemm_variant_name(v), names[dbg], t * 1e3, ops / t / 1e9);
      }
    mq_gemm_set_debug(0);
    free_problem(p);
    // fixed overhead: one K step only
    Problem p1 = make_problem(2048, 5632, 128, 5u, false);
    for (int v = 0; v < nvar; ++v) {
      float t = time_variant(p1, v, MQ_U8, true, iters);
      float t2 = time_variant(p1, v, MQ_F32, false, iters);
      printf("K=128 only %-16s u8q %.2f us  f32 %.2f us\n", mq_gemm_variant_name(v), t * 1e3, t2 * 1e3);
    }
    free_problem(p1);
  }
  printf("TOTAL_BAD=%d\n", total_bad);
  return total_bad ? 1 : 0;
}
