// Ellipsoid.contains as an HBM-streaming kernel (the north-star roofline
// kernel; reference nautilus/bounds/basic.py:340,360):
//     mask_i = | B_inv (x_i - c) |^2 < 1        (strict)
//
// Algorithmic traffic: 8*D bytes read + 1 byte written per point (SURVEY.md
// section 8d).  B_inv is lower triangular (it is inv(cholesky(A^-1)),
// basic.py:308-309), so a point costs D(D+1)/2 FMAs: at D = 50 that is 3.2
// flop/byte, below the fp64 ridge of the chip (78.6 TF / 8 TB/s), i.e. the
// kernel is HBM bound if the FMAs issue at >= ~50 % of peak.
//
// Design (CDNA4): one point per lane so the triangular mat-vec needs no
// cross-lane traffic and wastes no flops on padding; the 64 x D tile of a
// wavefront is fetched with coalesced 16-byte loads and transposed through LDS
// (odd row stride => conflict-free ds_read_b64); B_inv and c are wave-uniform
// and come in through scalar loads (SGPR operands of v_fma_f64), so the VALU
// only issues FMAs.  The next tile's global loads are issued before the
// current tile is computed.
#include "nb_common.h"

namespace {

template <int DT>
__global__ void __launch_bounds__(256)
nb_ell_stream_kernel(const double* __restrict__ cvec,
                     const double* __restrict__ binv,   // packed lower, row-major
                     int n_dim, const double* __restrict__ x, long long n,
                     unsigned char* __restrict__ mask) {
  constexpr int DP = 16 * DT;
  constexpr int NQ = DP / 2;          // 16-byte loads per lane per tile (max)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int S = n_dim | 1;            // odd row stride in doubles
  double* tile_lds = lds + (size_t)wave * 64 * S;

  // (row, col) of element e = 2*lane inside a tile, advanced by 128 per step
  const int row0 = (2 * lane) / n_dim;
  const int col0 = (2 * lane) - row0 * n_dim;
  const int drow = 128 / n_dim;
  const int dcol = 128 - drow * n_dim;
  const int nq = (64 * n_dim + 127) / 128;

  const long long n_tiles = (n + 63) >> 6;
  const long long n_rounds = (n_tiles + 4LL * gridDim.x - 1) / (4LL * gridDim.x);

  for (long long it = 0; it < n_rounds; ++it) {
    const long long tile = (it * gridDim.x + blockIdx.x) * 4 + wave;
    const long long p0 = tile * 64;
    long long cnt = (n - p0) * n_dim;              // elements left
    if (cnt > 64LL * n_dim) cnt = 64LL * n_dim;
    const double* src = x + p0 * n_dim;

    // coalesced 16-byte loads of the contiguous 64 x D block
    double2 stage[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const long long e = 2LL * (lane + 64 * q);
      stage[q] = make_double2(0.0, 0.0);
      if (q < nq && tile < n_tiles) {
        if (e + 1 < cnt) stage[q] = *(const double2*)(src + e);
        else if (e < cnt) stage[q].x = src[e];
      }
    }
    __syncthreads();                  // previous tile fully consumed
    {
      int row = row0, col = col0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q < nq) {
          if (row < 64) tile_lds[row * S + col] = stage[q].x;
          int rw2 = row, c2 = col + 1;
          if (c2 == n_dim) { c2 = 0; ++rw2; }
          if (rw2 < 64) tile_lds[rw2 * S + c2] = stage[q].y;
          row += drow; col += dcol;
          if (col >= n_dim) { col -= n_dim; ++row; }
        }
      }
    }
    __syncthreads();

    // one point per lane: d = x - c, y_i = sum_{j<=i} Binv[i][j] d_j
    double d[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j)
      d[j] = (j < n_dim) ? tile_lds[lane * S + j] - cvec[j] : 0.0;
    double r2 = 0.0;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
      if (i < n_dim) {
        const double* brow = binv + (i * (i + 1)) / 2;
        double y0 = 0.0, y1 = 0.0;
#pragma unroll
        for (int j = 0; j + 1 <= i; j += 2) {
          y0 = fma(brow[j], d[j], y0);
          y1 = fma(brow[j + 1], d[j + 1], y1);
        }
        if ((i & 1) == 0) y0 = fma(brow[i], d[i], y0);
        const double y = y0 + y1;
        r2 = fma(y, y, r2);
      }
    }
    const long long pt = p0 + lane;
    if (pt < n) mask[pt] = (r2 < 1.0) ? 1 : 0;
  }
}

template <int DT>
int launch(const double* cvec, const double* binv, int n_dim, const double* x,
           long long n, unsigned char* mask, hipStream_t stream) {
  const size_t lds = (size_t)4 * 64 * (n_dim | 1) * sizeof(double);
  static size_t lds_allowed = 0;
  if (lds > lds_allowed) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_ell_stream_kernel<DT>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    lds_allowed = lds;
  }
  (void)hipGetLastError();
  const long long n_tiles = (n + 63) >> 6;
  long long blocks = (n_tiles + 3) / 4;
  const long long per_cu = (160 * 1024) / (long long)lds;
  const long long cap = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nb_ell_stream_kernel<DT>, dim3((unsigned)blocks),
                     dim3(256), lds, stream, cvec, binv, n_dim, x, n, mask);
  return NB_OK;
}

}  // namespace

// cvec / binv point into the stream block of the blob (nb_common.h hdr[18]).
int nb_launch_ell_stream(const double* cvec, const double* binv, int n_dim,
                         const double* x, long long n, unsigned char* mask,
                         hipStream_t stream) {
  if (n <= 0) return NB_OK;
  const int dt = (n_dim + 15) / 16;
  switch (dt) {
    case 1: launch<1>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 2: launch<2>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 3: launch<3>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 4: launch<4>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 5: launch<5>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 6: launch<6>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 7: launch<7>(cvec, binv, n_dim, x, n, mask, stream); break;
    case 8: launch<8>(cvec, binv, n_dim, x, n, mask, stream); break;
    default:
      nb_set_error("n_dim > 128 unsupported");
      return NB_ERR_UNSUPPORTED;
  }
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
