// Device likelihoods of the BASELINE benchmark problems that are not
// quadratic forms: the Rosenbrock function of
// configuration C3 and the Neal funnel of configuration C5 (the D-dimensional
// extension of the reference's tests/test_sampler.py:311-314).  Both read the
// batch once (8 D bytes per point, HBM bound): 16 lanes share a point, every
// load instruction of a wavefront covers 4 points x 128 contiguous bytes, the
// per-point sum is a 4-step shuffle reduction.
#include "nb_common.h"

namespace {

__device__ __forceinline__ double group16_sum(double v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  return v;
}

// log L = -sum_i [ a (x_{i+1} - x_i^2)^2 + (1 - x_i)^2 ],  x = lo + (hi-lo) u
__global__ void __launch_bounds__(256)
nb_rosenbrock_kernel(const double* __restrict__ u, long long n, int d,
                     double lo, double width, double a,
                     double* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 4);
  for (long long p = (long long)blockIdx.x * (blockDim.x >> 4) +
                     (threadIdx.x >> 4);
       p < n; p += stride) {
    const double* row = u + p * d;
    double acc = 0.0;
    for (int i = sub; i < d - 1; i += 16) {
      const double xi = lo + width * row[i];
      const double xn = lo + width * row[i + 1];
      const double t = xn - xi * xi, s = 1.0 - xi;
      acc += a * t * t + s * s;
    }
    acc = group16_sum(acc);
    if (sub == 0) out[p] = -acc;
  }
}

// x_0 ~ N(mu, s0^2), x_i ~ N(mu, (exp(k (x_0 - mu)) / c)^2) for i > 0
__global__ void __launch_bounds__(256)
nb_funnel_kernel(const double* __restrict__ u, long long n, int d, double mu,
                 double s0, double k, double c, double* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const long long stride = (long long)gridDim.x * (blockDim.x >> 4);
  const double l2pi = 1.8378770664093453;
  for (long long p = (long long)blockIdx.x * (blockDim.x >> 4) +
                     (threadIdx.x >> 4);
       p < n; p += stride) {
    const double* row = u + p * d;
    const double x0 = row[0];
    const double ln_s = k * (x0 - mu) - log(c);
    double acc = 0.0;
    for (int i = 1 + sub; i < d; i += 16) {
      const double t = row[i] - mu;
      acc += t * t;
    }
    acc = group16_sum(acc);
    if (sub == 0) {
      const double z0 = (x0 - mu) / s0;
      out[p] = -0.5 * z0 * z0 - log(s0) - 0.5 * l2pi -
               0.5 * acc * exp(-2.0 * ln_s) - (d - 1) * ln_s -
               0.5 * (d - 1) * l2pi;
    }
  }
}

inline unsigned like_blocks(long long n) {
  long long b = (n + 15) / 16;
  return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

}  // namespace

int nb_launch_rosenbrock(const double* u, long long n, int d, double lo,
                         double hi, double a, double* out, hipStream_t stream) {
  if (n <= 0) return NB_OK;
  hipLaunchKernelGGL(nb_rosenbrock_kernel, dim3(like_blocks(n)), dim3(256), 0,
                     stream, u, n, d, lo, hi - lo, a, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_funnel(const double* u, long long n, int d, double mu, double s0,
                     double k, double c, double* out, hipStream_t stream) {
  if (n <= 0) return NB_OK;
  hipLaunchKernelGGL(nb_funnel_kernel, dim3(like_blocks(n)), dim3(256), 0,
                     stream, u, n, d, mu, s0, k, c, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
