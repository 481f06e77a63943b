#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
// single wave per SIMD: N16 independent 16x16x4 + N4 4x4x4 MFMAs per iteration
template <int N16, int N4, int ORDER>
__global__ void __launch_bounds__(256) k(int iters, double* sink, long long* cyc) {
  d4 acc[N16 > 0 ? N16 : 1];
  double r[N4 > 0 ? N4 : 1];
  for (int i = 0; i < N16; ++i) acc[i] = d4{0, 0, 0, 0};
  for (int i = 0; i < N4; ++i) r[i] = 0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < N16; ++i)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < N4; ++i)
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
    } else {
      // 4x4x4 in the middle
#pragma unroll
      for (int i = 0; i < N16 / 2; ++i)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < N4; ++i)
        asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = N16 / 2; i < N16; ++i)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < N16; ++i) s += acc[i][0];
  for (int i = 0; i < N4; ++i) s += r[i];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int N16, int N4, int ORDER>
void run() {
  double* sink; long long* cyc; hipMalloc(&sink, 8); hipMalloc(&cyc, 8);
  const int iters = 2000;
  k<N16, N4, ORDER><<<256, 256>>>(100, sink, cyc); hipDeviceSynchronize();
  k<N16, N4, ORDER><<<256, 256>>>(iters, sink, cyc); hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("N16=%d N4=%d order=%d: %.1f cycles/iter (ideal %d)\n", N16, N4, ORDER, (double)c / iters, 64 * N16 + 16 * N4);
}
int main() {
  run<12, 0, 0>(); run<12, 2, 0>(); run<12, 2, 1>(); run<0, 2, 0>(); run<0, 8, 0>(); run<4, 2, 0>(); run<4, 0, 0>(); run<2,2,0>(); run<1, 0, 0>(); run<2, 0, 0>();
  return 0;
}
