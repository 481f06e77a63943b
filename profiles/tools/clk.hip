#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(int iters, double* sink, long long* cyc) {
  d4 acc = {0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  long long t1 = clock64();
  if (acc[0] == 12345.678) sink[0] = acc[0];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* sink; long long* cyc; (void)hipMalloc(&sink, 8); (void)hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {1, 1, 256, 256, 1}) {
    const int iters = 1000000;
    (void)hipEventRecord(e0); k<<<blocks, 256>>>(iters, sink, cyc); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("blocks=%d: %.2f ms, %lld ticks -> %.2f GHz (64 cycles per MFMA: %.2f GHz)\n", blocks, ms, c, c / ms / 1e6, iters * 64.0 / ms / 1e6);
  }
  return 0;
}
