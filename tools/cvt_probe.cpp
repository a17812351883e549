// What does v_cvt_pk_u8_f32 do with ties, negatives and values above 255?  (decides the GEMM epilogue fast path)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
  const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.6f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.5f, -0.6f, -1.f, -300.f, 127.5f, 128.5f, 0.49999997f, 1.4999999f};
  const int n = sizeof(h) / sizeof(float);
  float* dx; unsigned* dout; unsigned r[64];
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dout, n);
  hipMemcpy(r, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%14.8g -> %u\n", h[i], r[i] & 0xff);
  return 0;
}
