#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(1024) k(int iters, double* sink) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a + i, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0];
  if (s == 12345.678) sink[0] = s;
}
template <int NACC>
void run(int threads, int blocks_per_cu) {
  double* sink; hipMalloc(&sink, 8);
  const int iters = 20000;
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, threads>>>(100, sink); hipDeviceSynchronize();
  hipEventRecord(e0); k<NACC><<<blocks, threads>>>(iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
  printf("acc=%d threads=%d blocks/CU=%d waves/SIMD=%.1f: %.1f TF\n", NACC, threads, blocks_per_cu, threads / 64.0 * blocks_per_cu / 4, flops / ms / 1e9);
  hipFree(sink);
}
int main() {
  run<1>(256, 1); run<2>(256, 1); run<4>(256, 1); run<8>(256, 1);
  run<4>(512, 1); run<4>(1024, 1); run<2>(1024, 1); run<1>(1024, 2); run<4>(256, 2); run<4>(256, 4);
  return 0;
}
