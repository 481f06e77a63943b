// Sustained fp64 MFMA rate of the whole chip as a function of the operand data:
// constant operands vs random mantissas (switching power -> clock).
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NOPS, int AG, int NACC>
__global__ void __launch_bounds__(256) k(int iters, const double* src, double* sink, long long* cyc) {
  double a[NOPS], b[NOPS];
  for (int i = 0; i < NOPS; ++i) {
    a[i] = src[(threadIdx.x * NOPS + i) % 4096];
    b[i] = src[(threadIdx.x * NOPS + i + 2048) % 4096];
  }
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NOPS; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (AG) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[(i * 4 + j) % NACC]) : "v"(a[i]), "v"(b[(i + j) % NOPS]));
        else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[(i * 4 + j) % NACC]) : "v"(a[i]), "v"(b[(i + j) % NOPS]));
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int AG, int NACC>
void run(const double* src, double* sink, long long* cyc, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  k<8, AG, NACC><<<256, 256>>>(1000, src, sink, cyc); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<8, AG, NACC><<<256, 256>>>(iters, src, sink, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double flop = 256.0 * 4 * iters * 32 * 2048;
  printf("%s acc in %s, %d accumulators: %.2f ms  %.1f TFLOP/s  clock %.2f GHz  %.1f cycles/MFMA\n", name, AG ? "AGPR" : "VGPR", NACC, ms, flop / ms / 1e9, c / ms / 1e6, (double)c / iters / 32);
}
int main() {
  double* src; double* sink; long long* cyc;
  (void)hipMalloc(&src, 4096 * 8); (void)hipMalloc(&sink, 8); (void)hipMalloc(&cyc, 8);
  double h[4096];
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (int i = 0; i < 4096; ++i)
      h[i] = mode == 0 ? 0.0 : (mode == 1 ? 1.0 : (rand() / (double)RAND_MAX - 0.5) * 1e-3);
    (void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    const char* name = mode == 0 ? "zeros" : (mode == 1 ? "ones" : "random");
    run<1, 4>(src, sink, cyc, name); run<0, 4>(src, sink, cyc, name);
    run<1, 8>(src, sink, cyc, name); run<0, 8>(src, sink, cyc, name);
    run<1, 2>(src, sink, cyc, name); run<1, 1>(src, sink, cyc, name);
  }
  return 0;
}
