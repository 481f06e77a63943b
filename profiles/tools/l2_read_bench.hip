// Micro-benchmark (development aid, not part of the library): how fast does
// ONE workgroup of four wavefronts pull an L2-resident block of weights, as
// the trainer's FB phase does -- 8-byte against 16-byte loads per lane, plain
// against device-scope (sc1) loads, 13 or 32 workgroups at once.
//   hipcc --offload-arch=gfx950 -O3 profiles/tools/l2_read_bench.hip -o /tmp/l2_read_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) double gd;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) d2 gd2;

template <int MODE>
__global__ void __launch_bounds__(256)
read_kernel(const double* buf, long long n_doubles, int iters, double* out,
            long long* cycles) {
  const gd* p = (const gd*)buf;
  const int tid = threadIdx.x;
  double acc = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // 8 bytes per lane, plain
      for (long long i = tid; i < n_doubles; i += 256 * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = p[i + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
    } else if (MODE == 1) {     // 8 bytes per lane, device scope (sc1)
      for (long long i = tid; i < n_doubles; i += 256 * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = __hip_atomic_load(p + i + j * 256, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
      }
    } else if (MODE == 2) {     // 16 bytes per lane, plain
      const gd2* q = (const gd2*)buf;
      for (long long i = tid; i < n_doubles / 2; i += 256 * 8) {
        d2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = q[i + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
      }
    } else {                    // 16 bytes per lane, sc1
      const gd2* q = (const gd2*)buf;
      for (long long i = tid; i < n_doubles / 2; i += 256 * 8) {
        d2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          asm volatile("global_load_dwordx4 %0, %1, off sc1"
                       : "=v"(v[j]) : "v"(q + i + j * 256) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x + v[j].y;
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + tid] = acc;
}

int main() {
  const long long n = 22016;             // 172 KB of doubles (multiple of 2048)
  double* buf; double* out; long long* cyc;
  hipMalloc(&buf, n * 8); hipMalloc(&out, 64 * 256 * 8);
  hipMalloc(&cyc, 64 * 8);
  hipMemset(buf, 0, n * 8);
  const int iters = 200;
  for (int blocks : {1, 13, 32}) {
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(read_kernel<0>, dim3(blocks), dim3(256), 0, 0, buf, n, iters, out, cyc); break;
          case 1: hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(256), 0, 0, buf, n, iters, out, cyc); break;
          case 2: hipLaunchKernelGGL(read_kernel<2>, dim3(blocks), dim3(256), 0, 0, buf, n, iters, out, cyc); break;
          default: hipLaunchKernelGGL(read_kernel<3>, dim3(blocks), dim3(256), 0, 0, buf, n, iters, out, cyc); break;
        }
        hipDeviceSynchronize();
      }
      std::vector<long long> h(blocks);
      hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
      long long mx = 0;
      for (long long c : h) mx = c > mx ? c : mx;
      // s_memtime counts at 100 MHz on this part; report both
      printf("blocks %2d mode %d (%s, %s): %lld ticks for %d x %lld KB -> %.1f B per tick and workgroup\n",
             blocks, mode, mode < 2 ? "8 B/lane" : "16 B/lane",
             (mode & 1) ? "sc1" : "plain", mx, iters, n * 8 / 1024,
             (double)iters * n * 8 / mx);
    }
  }
  return 0;
}
