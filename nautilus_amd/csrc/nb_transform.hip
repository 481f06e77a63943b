// Ellipsoid-frame coordinates and input standardisation for the emulator's
// training set (reference nautilus/bounds/basic.py:340 Ellipsoid.transform,
// nautilus/neural.py:74-77 mean / scale / (x - mean) / scale).  The transform
// runs through the same matrix-core tile code as contains() and the emulator
// evaluation (nb_tile.h), so the network is trained on exactly the inputs it
// later sees inside nb_eval_kernel.
#include "nb_tile.h"

namespace {

template <int DT>
__global__ void __launch_bounds__(256)
nb_transform_kernel(const double* __restrict__ blk, int n_dim,
                    const double* __restrict__ x, long long n,
                    double* __restrict__ y_out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lg = lane >> 4;
  const long long n_tiles = (n + 15) / 16;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles;
       tile += (long long)gridDim.x * 4) {
    long long pt[1] = {tile * 16 + (lane & 15)};
    bool valid[1] = {pt[0] < n};
    double xin[1][4 * DT], y[1][4 * DT], r2[1];
    bool box_bad[1];
    load_points<DT, 1>((const nb_gd*)x, pt, valid, n_dim, n, lane, xin);
    ell_eval<DT, 1>(blk, n_dim, xin, lane, y, box_bad, r2);
    if (valid[0]) {
#pragma unroll
      for (int j = 0; j < 4 * DT; ++j) {
        const int unit = 4 * j + lg;          // C/D layout of the MFMA
        if (unit < n_dim) y_out[pt[0] * n_dim + unit] = y[0][j];
      }
    }
  }
}

// per-column mean and population standard deviation (two passes, fixed
// reduction order), one workgroup per column
__global__ void __launch_bounds__(256)
nb_colstats_kernel(const double* __restrict__ x, long long n, int d,
                   double* __restrict__ mean, double* __restrict__ scale) {
  __shared__ double red[4];
  const int col = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double result[2];
  double centre = 0.0;
  for (int pass = 0; pass < 2; ++pass) {
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 256) {
      const double v = x[i * d + col] - centre;
      s += pass == 0 ? v : v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    result[pass] = (red[0] + red[1] + red[2] + red[3]) / (double)n;
    if (pass == 0) centre = result[0];
  }
  if (threadIdx.x == 0) {
    mean[col] = result[0];
    scale[col] = sqrt(result[1]);
  }
}

__global__ void __launch_bounds__(256)
nb_standardize_kernel(const double* __restrict__ x, long long total, int d,
                      const double* __restrict__ mean,
                      const double* __restrict__ scale,
                      double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
       e < total; e += stride) {
    const int col = (int)(e % d);
    out[e] = (x[e] - mean[col]) / scale[col];
  }
}

// Prior.unit_to_physical (reference nautilus/prior.py:85-120: x_j =
// dist_j.isf(1 - u_j)) for the two distribution families that cover flat and
// Gaussian priors, evaluated with scipy's formulas:
//   uniform(loc, scale).isf(q) = (1 - q) * scale + loc
//   norm(loc, scale).isf(q)    = -ndtri(q) * scale + loc
struct PriorArgs {
  double p0[16 * NB_MAX_DT], p1[16 * NB_MAX_DT];
  unsigned char kind[16 * NB_MAX_DT];     // 0 uniform, 1 normal
};

__global__ void __launch_bounds__(256)
nb_prior_kernel(const double* __restrict__ u, long long total, int d,
                PriorArgs a, double* __restrict__ out) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
       e < total; e += stride) {
    const int col = (int)(e % d);
    const double q = 1.0 - u[e];
    const double z = a.kind[col] ? -normcdfinv(q) : 1.0 - q;
    out[e] = z * a.p1[col] + a.p0[col];
  }
}

}  // namespace

int nb_launch_prior(const double* u, long long n, int d,
                    const unsigned char* kind, const double* loc,
                    const double* scale, double* out, hipStream_t stream) {
  if (n <= 0) return NB_OK;
  PriorArgs a;
  for (int j = 0; j < 16 * NB_MAX_DT; ++j) {
    a.kind[j] = j < d ? kind[j] : 0;
    a.p0[j] = j < d ? loc[j] : 0.0;
    a.p1[j] = j < d ? scale[j] : 1.0;
  }
  const long long total = n * d;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nb_prior_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     stream, u, total, d, a, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_transform(const double* ell_block, int dt, int n_dim,
                        const double* x, long long n, double* y,
                        hipStream_t stream) {
  if (n <= 0) return NB_OK;
  long long blocks = ((n + 15) / 16 + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  switch (dt) {
#define NB_CASE(DT_)                                                         \
    case DT_:                                                                \
      hipLaunchKernelGGL(nb_transform_kernel<DT_>, dim3((unsigned)blocks),   \
                         dim3(256), 0, stream, ell_block, n_dim, x, n, y);   \
      break;
    NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4)
    NB_CASE(5) NB_CASE(6) NB_CASE(7) NB_CASE(8)
#undef NB_CASE
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_standardize(const double* x, long long n, int d, double* mean,
                          double* scale, double* out, hipStream_t stream) {
  if (n <= 0 || d <= 0) return NB_OK;
  hipLaunchKernelGGL(nb_colstats_kernel, dim3(d), dim3(256), 0, stream, x, n,
                     d, mean, scale);
  if (out != nullptr) {
    const long long total = n * d;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(nb_standardize_kernel, dim3((unsigned)blocks),
                       dim3(256), 0, stream, x, total, d, mean, scale, out);
  }
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
