// Minimum-volume enclosing ellipsoid: the batched Khachiyan iteration of
// minimum_volume_enclosing_ellipsoid (reference nautilus/bounds/basic.py:
// 175-232) for one point set, as ONE persistent workgroup.
//
//   q_i = (x_i, 1);  V = sum_i u_i q_i q_i^T;  g_i = q_i^T V^-1 q_i
//   per sweep (n_max = 100): the n_batch = 20 largest g_i, in descending
//   order, each update  a = (g - (D+1)) / ((D+1)(g - 1)),
//   V <- (1-a) V + a q_j q_j^T,  u <- (1-a) u + a e_j          (basic.py:217-231)
//
// Everything the loop touches except the points lives in LDS: V, its inverse
// and a work matrix ((D+1)^2 doubles each), the quadratic forms g.  Per sweep
//   1. V = L D L^T by Gaussian elimination on [V | I] (one barrier per
//      pivot), which yields L^-1 directly; R^-1 = D^-1/2 L^-1 and
//      V^-1 = L^-T D^-1 L^-1                                      (VALU, LDS)
//   2. g_i = |R^-1 q_i|^2 for all points on the matrix cores: the same
//      16-point-tile ellipsoid transform as Ellipsoid.contains (nb_tile.h),
//      with R^-1 scattered into the tile-major K-permuted operand layout
//   3. top-n_batch selection (block-wide arg-max rounds)
//   4. the sequential rank-one updates, two barriers each; V^-1 follows by
//      Sherman-Morrison (the reference re-factorises after every update,
//      basic.py:230 -- the same matrix up to rounding)
// The kernel returns the weights u; centre, covariance and the final scaling
// (basic.py:233-241) are a handful of (D x D) host operations on top of u.
#include "nb_tile.h"

namespace {

constexpr int MV_THREADS = 512;          // 8 wavefronts, 2 per SIMD
constexpr int MV_WAVES = MV_THREADS / 64;
constexpr int MV_EPT = 8;                // matrix elements per thread (m <= 64)
constexpr int MV_SEL = 64;               // upper limit of n_batch

__device__ __forceinline__ int mv_slot(int f) {      // slot_of_feature (nb_api)
  const int j = f >> 3, r = f & 7;
  return 4 * (2 * j + (r & 1)) + (r >> 1);
}

template <int DT>
__global__ void __launch_bounds__(MV_THREADS)
nb_mvee_kernel(const double* __restrict__ x, int n, int d, int n_max,
               int n_batch, volatile double* u, volatile double* g_glob,
               int g_in_lds) {
  constexpr int DP = 16 * DT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double wv[64], pr[64], piv[64], sel_g[MV_SEL], red_v[2 * MV_WAVES];
  __shared__ int sel_i[MV_SEL], red_i[2 * MV_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4;
  const int m = d + 1;
  const int mm = (m * m + 1) & ~1;
  double* V = lds;
  double* A = V + mm;          // elimination work matrix, then V^-1
  double* B = A + mm;          // L^-1 of V = L D L^T
  double* ell = B + mm;        // ell block: n_ell, pad, lo, hi, c, tiles
  double* tiles = ell + 2 + 3 * DP;
  double* qsel = ell + nb_ell_block_size(DT);       // [MV_SEL][DP] selected rows
  double* g_lds = qsel + (n_batch < MV_SEL ? n_batch : MV_SEL) * DP;
  const double inf = __builtin_huge_val();

  // the (row, column) pairs this thread owns in every element-wise pass
  int er[MV_EPT], ec[MV_EPT];
#pragma unroll
  for (int q = 0; q < MV_EPT; ++q) {
    const int e = tid + q * MV_THREADS;
    er[q] = (e < m * m) ? e / m : -1;
    ec[q] = (e < m * m) ? e - er[q] * m : 0;
  }

  // ---- initialisation ----------------------------------------------------
  for (int e = tid; e < nb_ell_block_size(DT); e += MV_THREADS) ell[e] = 0.0;
  __syncthreads();
  if (tid == 0) ((long long*)ell)[0] = m;
  for (int f = tid; f < DP; f += MV_THREADS) {
    ell[2 + f] = -inf;
    ell[2 + DP + f] = inf;
  }
  for (int i = tid; i < n; i += MV_THREADS) u[i] = 1.0 / (double)n;
  // V = sum_i u_i q_i q_i^T (basic.py:218), lower triangle then mirrored
#pragma unroll
  for (int q = 0; q < MV_EPT; ++q) {
    const int r = er[q], c = ec[q];
    if (r < 0 || c > r) continue;
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
      const double qr = (r < d) ? x[(long long)i * d + r] : 1.0;
      const double qc = (c < d) ? x[(long long)i * d + c] : 1.0;
      acc += qr * qc;
    }
    acc /= (double)n;
    V[r * m + c] = acc;
    V[c * m + r] = acc;
  }
  double scale = 1.0;          // true weights = scale * u (lazy (1-a) factors)
  __syncthreads();

  const int n_sel = n_batch < n ? (n_batch < MV_SEL ? n_batch : MV_SEL) : n;
  const int ks_one = mv_slot(d) >> 2, lg_one = mv_slot(d) & 3;

  for (int it = 0; it < n_max; ++it) {
    // ---- 1. V = L D L^T by elimination on [V | I]: A -> D L^T, B -> L^-1 ---
#pragma unroll
    for (int q = 0; q < MV_EPT; ++q)
      if (er[q] >= 0) {
        A[er[q] * m + ec[q]] = V[er[q] * m + ec[q]];
        B[er[q] * m + ec[q]] = (er[q] == ec[q]) ? 1.0 : 0.0;
      }
    __syncthreads();
    for (int k = 0; k < m - 1; ++k) {
      const double inv_d = 1.0 / A[k * m + k];
#pragma unroll
      for (int q = 0; q < MV_EPT; ++q) {
        const int i = er[q], j = ec[q];
        if (i > k) {
          const double f = A[i * m + k] * inv_d;
          if (j > k) A[i * m + j] -= f * A[k * m + j];
          else B[i * m + j] -= f * B[k * m + j];
        }
      }
      __syncthreads();
    }
    if (tid < m) piv[tid] = 1.0 / A[tid * m + tid];       // 1 / d_k
    __syncthreads();
    // R^-1 = D^-1/2 L^-1 -> operand tiles;  V^-1 = L^-T D^-1 L^-1 -> A
#pragma unroll
    for (int q = 0; q < MV_EPT; ++q) {
      const int r = er[q], c = ec[q];
      if (r < 0) continue;
      if (c <= r) {
        const int sl = mv_slot(c), ks = sl >> 2, lgk = sl & 3;
        tiles[((ks >> 2) * DT + (r >> 4)) * NB_TILE + (ks & 3) * 64 +
              lgk * 16 + (r & 15)] = B[r * m + c] * sqrt(piv[r]);
      }
    }
    __syncthreads();           // A (D L^T) is dead from here
#pragma unroll
    for (int q = 0; q < MV_EPT; ++q) {
      const int r = er[q], c = ec[q];
      if (r < 0) continue;
      double s = 0.0;
      for (int k = (r > c ? r : c); k < m; ++k)
        s += B[k * m + r] * B[k * m + c] * piv[k];
      A[r * m + c] = s;
    }

    // ---- 2. g_i = |R^-1 q_i|^2 on the matrix cores (basic.py:220) ---------
    for (int tile = wave; tile * 16 < n; tile += MV_WAVES) {
      long long pt[1] = {(long long)tile * 16 + (lane & 15)};
      bool valid[1] = {pt[0] < n};
      double xin[1][4 * DT], y[1][4 * DT], r2[1];
      bool box_bad[1];
      load_points<DT, 1>(x, pt, valid, d, (long long)n, lane, xin);
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks)
        if (ks == ks_one && lg == lg_one) xin[0][ks] = 1.0;
      ell_eval<DT, 1>(ell, m, xin, lane, y, box_bad, r2);
      if (valid[0] && lg == 0) {
        if (g_in_lds) g_lds[pt[0]] = r2[0];
        else g_glob[pt[0]] = r2[0];
      }
    }
    __threadfence_block();
    __syncthreads();

    // ---- 3. the n_batch largest g, descending (basic.py:221) --------------
    for (int t = 0; t < n_sel; ++t) {
      double bv = -inf;
      int bi = -1;
      for (int i = tid; i < n; i += MV_THREADS) {
        const double v = g_in_lds ? g_lds[i] : g_glob[i];
        if (v > bv || (v == bv && i > bi)) { bv = v; bi = i; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
      }
      // per-wave results are double buffered on the parity of t, so one
      // barrier per round is enough; every thread merges them identically
      double* rv = red_v + (t & 1) * MV_WAVES;
      int* ri = red_i + (t & 1) * MV_WAVES;
      if (lane == 0) { rv[wave] = bv; ri[wave] = bi; }
      __syncthreads();
      bv = rv[0]; bi = ri[0];
#pragma unroll
      for (int w = 1; w < MV_WAVES; ++w)
        if (rv[w] > bv || (rv[w] == bv && ri[w] > bi)) { bv = rv[w]; bi = ri[w]; }
      if (tid == 0) { sel_g[t] = bv; sel_i[t] = bi; }
      // the thread that scans element bi retires it before its next scan
      if ((bi % MV_THREADS) == tid) {
        if (g_in_lds) g_lds[bi] = -inf;
        else g_glob[bi] = -inf;
      }
    }
    __syncthreads();
    // the selected rows q_j = (x_j, 1), fetched in one go
    for (int e = tid; e < n_sel * m; e += MV_THREADS) {
      const int t = e / m, c = e - t * m;
      qsel[t * DP + c] = (c < d) ? x[(long long)sel_i[t] * d + c] : 1.0;
    }
    __syncthreads();

    // ---- 4. rank-one updates (basic.py:221-231) ---------------------------
    for (int t = 0; t < n_sel; ++t) {
      const double* qj = qsel + t * DP;
      {  // w = V^-1 q_j: 8 lanes per row, products q_j[r] w[r] for g
        const int r = tid >> 3, p = tid & 7;
        double s = 0.0;
        if (r < m)
          for (int c = p; c < m; c += 8) s += A[r * m + c] * qj[c];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (p == 0 && r < 64) {
          wv[r] = (r < m) ? s : 0.0;
          pr[r] = (r < m) ? s * qj[r] : 0.0;
        }
      }
      __syncthreads();
      double gq = pr[lane];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) gq += __shfl_xor(gq, o);
      if (t == 0) gq = sel_g[0];
      if (gq >= (double)m) {
        const double a = (gq - m) / ((double)m * (gq - 1.0));
        const double ratio = a / (1.0 - a);
        const double coef = ratio / (1.0 + ratio * gq);
        const double inv1a = 1.0 / (1.0 - a);
#pragma unroll
        for (int q = 0; q < MV_EPT; ++q) {
          const int r = er[q], c = ec[q];
          if (r < 0) continue;
          V[r * m + c] = V[r * m + c] * (1.0 - a) + a * (qj[r] * qj[c]);
          A[r * m + c] = (A[r * m + c] - wv[r] * wv[c] * coef) * inv1a;
        }
        scale *= (1.0 - a);
        if (tid == 0) u[sel_i[t]] += a / scale;
      }
      __syncthreads();
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int i = tid; i < n; i += MV_THREADS) u[i] *= scale;
}

template <int DT>
int launch_mvee(const double* x, int n, int d, int n_max, int n_batch,
                double* u, double* g, hipStream_t stream) {
  const int m = d + 1;
  const int mm = (m * m + 1) & ~1;
  size_t lds = ((size_t)3 * mm + nb_ell_block_size(DT) +
                (size_t)(n_batch < MV_SEL ? n_batch : MV_SEL) * 16 * DT) *
               sizeof(double);
  // the quadratic forms g stay in LDS when they fit next to the matrices
  const size_t room = (size_t)156 * 1024;
  const int g_in_lds = lds + (size_t)n * sizeof(double) <= room ? 1 : 0;
  if (g_in_lds) lds += (size_t)n * sizeof(double);
  static size_t allowed = 0;
  if (lds > allowed) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_mvee_kernel<DT>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    allowed = lds;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(nb_mvee_kernel<DT>, dim3(1), dim3(MV_THREADS), lds, stream,
                     x, n, d, n_max, n_batch, u, g, g_in_lds);
  return NB_OK;
}

}  // namespace

// host entry used by nb_api.hip.  n_dim + 1 <= 64 (the three matrices and the
// operand tiles must fit the 160 KB of LDS).
int nb_launch_mvee(const double* x, long long n, int n_dim, int n_max,
                   int n_batch, double* u, double* g, hipStream_t stream) {
  if (n_dim < 1 || n_dim + 1 > 64) {
    nb_set_error("device MVEE supports n_dim <= 63 (got %d)", n_dim);
    return NB_ERR_UNSUPPORTED;
  }
  if (n <= n_dim || n > 2147483647LL / (n_dim > 0 ? n_dim : 1)) {
    nb_set_error("device MVEE needs n_dim < n (n=%lld, n_dim=%d)", n, n_dim);
    return NB_ERR_ARG;
  }
  if (n_batch < 1 || n_batch > 64 || n_max < 0) {
    nb_set_error("device MVEE: n_batch must be in 1..64");
    return NB_ERR_ARG;
  }
  const int dt = (n_dim + 1 + 15) / 16;
  int rc = NB_OK;
  switch (dt) {
    case 1: rc = launch_mvee<1>(x, (int)n, n_dim, n_max, n_batch, u, g, stream); break;
    case 2: rc = launch_mvee<2>(x, (int)n, n_dim, n_max, n_batch, u, g, stream); break;
    case 3: rc = launch_mvee<3>(x, (int)n, n_dim, n_max, n_batch, u, g, stream); break;
    default: rc = launch_mvee<4>(x, (int)n, n_dim, n_max, n_batch, u, g, stream); break;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
