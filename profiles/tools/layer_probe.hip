// Isolated efficiency of the pipelined layer code (nb_mlp.h): one workgroup per
// CU, 4 wavefronts x 2 tiles, weights in LDS, no DMA, no barriers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nautilus_amd/csrc -I include profiles/tools/layer_probe.hip -o /tmp/layer_probe && /tmp/layer_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "nb_mlp.h"
void nb_set_error(const char*, ...) {}
#ifndef PD3
#define PD3 2
#endif
#ifndef PD4
#define PD4 5
#endif

template <int MODE>
__global__ void __launch_bounds__(256) probe(int iters, double* sink, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int i = threadIdx.x; i < 66 * NB_TILE; i += 256) lds[i] = 1e-3 * (i % 17);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  constexpr int T = 2;
  double tin[T][16], h1[T][28], h2[T][16], h3[T][8], o[T][4];
  for (int t = 0; t < T; ++t) for (int k = 0; k < 16; ++k) tin[t][k] = 1.0 + 1e-3 * (lane + k + t);
  int dummy = 0;
  auto tick = [&]() __attribute__((always_inline)) { if (MODE == 2) { if (dummy < iters - 1000000) { lds[lane] = 0; dummy += 4; } } };
  double acc = 0;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
      double a0[FlFirst<2, NB_HT1>::NA];
      fl_read_first<2, NB_HT1>(lds, lane, a0);
      fl_layer_from<T, 2, 16, 3, NB_HT1, false, 2, 1, 0>(lds, 13, tin, lane, h1, a0, []() __attribute__((always_inline)) {}, tick);
      fl_pad<T, NB_HT1, 25, 0>(h1, lane);
      for (int t = 0; t < T; ++t) for (int k = 0; k < 25; ++k) acc += h1[t][k];
    } else {
      for (int t = 0; t < T; ++t) for (int k = 0; k < 28; ++k) h1[t][k] = tin[t][k & 15] + it;
      for (int t = 0; t < T; ++t) for (int k = 0; k < 16; ++k) h2[t][k] = tin[t][k & 15] - it;
      for (int t = 0; t < T; ++t) for (int k = 0; k < 8; ++k) h3[t][k] = tin[t][k & 15] * it;
      const double* w2 = lds + 28 * NB_TILE;
      const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
      const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
      double a2[FlFirst<2, NB_HT2>::NA], a3[FlFirst<2, NB_HT3>::NA], a4[FlFirst<2, 1>::NA];
      if (MODE == 1) {
      fl_read_first<2, NB_HT2>(w2, lane, a2);
      fl_layer_from<T, 2, 26, 0, NB_HT2, true, 4, 1, 0>(w2, 26, h1, lane, h2, a2, [&]() __attribute__((always_inline)) { fl_read_first<2, NB_HT3>(w3, lane, a3); }, tick);
      fl_pad<T, NB_HT2, 12, 2>(h2, lane);
      fl_layer_from<T, 2, 13, 0, NB_HT3, true, 4, PD3, 0>(w3, 13, h2, lane, h3, a3, [&]() __attribute__((always_inline)) { fl_read_first<2, 1>(w4, lane, a4); }, tick);
      fl_pad<T, NB_HT3, 5, 0>(h3, lane);
      fl_layer_from<T, 2, 6, 0, 1, true, 4, PD4, 0>(w4, 6, h3, lane, o, a4, []() __attribute__((always_inline)) {}, tick);
      for (int t = 0; t < T; ++t) acc += o[t][0];
      } else if (MODE == 3) {
      fl_read_first<2, NB_HT2>(w2, lane, a2);
      fl_layer_from<T, 2, 26, 0, NB_HT2, true, 4, 1, 0>(w2, 26, h1, lane, h2, a2, []() __attribute__((always_inline)) {}, tick);
      for (int t = 0; t < T; ++t) for (int k = 0; k < 13; ++k) acc += h2[t][k];
      } else if (MODE == 4) {
      fl_read_first<2, NB_HT3>(w3, lane, a3);
      fl_layer_from<T, 2, 13, 0, NB_HT3, true, 4, PD3, 0>(w3, 13, h2, lane, h3, a3, []() __attribute__((always_inline)) {}, tick);
      for (int t = 0; t < T; ++t) for (int k = 0; k < 5; ++k) acc += h3[t][k];
      } else if (MODE == 5) {
      fl_read_first<2, 1>(w4, lane, a4);
      fl_layer_from<T, 2, 6, 0, 1, true, 4, PD4, 0>(w4, 6, h3, lane, o, a4, []() __attribute__((always_inline)) {}, tick);
      for (int t = 0; t < T; ++t) acc += o[t][0];
      }
    }
    for (int t = 0; t < T; ++t) for (int k = 0; k < 16; ++k) tin[t][k] += 1e-9 * acc;
  }
  long long t1 = clock64();
  if (acc == 12345.678) sink[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int ideal) {
  double* sink; long long* cyc; (void)hipMalloc(&sink, 8); (void)hipMalloc(&cyc, 8);
  (void)hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 2048);
  const int iters = 2000;
  for (int blocks : {1, 256}) {
    probe<MODE><<<blocks, 256, 66 * 2048>>>(100, sink, cyc); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 66 * 2048>>>(iters, sink, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s blocks=%d: %.0f ticks/iter, %.3f us/iter -> %.0f cycles at 2.4 GHz (ideal MFMA %d, eff %.2f)\n", name, blocks, (double)c / iters, ms * 1e3 / iters, ms * 1e3 / iters * 2400, ideal, ideal / (ms * 1e3 / iters * 2400));
  }
}
int main() {
  run<0>("layer 1 (13 k-steps)", 13 * 800);
  run<2>("layer 1 + tick branch", 13 * 800);
  run<1>("layers 2-4", 26 * 416 + 13 * 160 + 6 * 32);
  run<3>("layer 2", 26 * 416);
  run<4>("layer 3", 13 * 160);
  run<5>("layer 4", 6 * 32);
  return 0;
}
