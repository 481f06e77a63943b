// Two-component full-covariance Gaussian mixture for Union.split (reference
// nautilus/bounds/union.py:185-187: scikit-learn GaussianMixture(n_components
// =2, n_init=10), the reference's third-party dependency for this step).  The
// algorithm restated here is scikit-learn 1.7's: k-means++ / Lloyd
// initialisation (mixture/_base.py:_initialize_parameters, cluster/_kmeans.py),
// then EM with tol = 1e-3 on the mean log-likelihood, reg_covar = 1e-6,
// max_iter = 100 (mixture/_base.py:fit_predict, _gaussian_mixture.py).
//
// One workgroup per restart, all restarts of a fit run concurrently.
//   E-step: Sigma_k = L D L^T by elimination in LDS, R^-1 = D^-1/2 L^-1 scattered
//           into MFMA operand tiles, |R^-1 (x - mu_k)|^2 for all points on the
//           matrix cores (the Ellipsoid.contains tile code, nb_tile.h)
//   M-step: ONE weighted second-moment product on the matrix cores over the
//           augmented rows q = (x, 1):  S0 = sum_i r_i0 q_i q_i^T  holds
//           sum r x x^T, sum r x and sum r at once; component 1 follows from
//           S_all - S0 (S_all = sum q q^T, computed once)
// Everything is deterministic (fixed reduction orders, Philox for the seeding).
#include "nb_tile.h"

namespace {

constexpr int GM_THREADS = 512;
constexpr int GM_WAVES = GM_THREADS / 64;
constexpr int GM_EPT = 8;                 // matrix elements per thread (<= 64^2)
constexpr unsigned GM_TAG = 3u;           // Philox tag of the seeding draws

struct GmmArgs {
  const double* x;
  int n, d, n_init;
  unsigned long long seed;
  double tol, reg;
  int max_iter;
  const int* init_labels;     // optional [n_init][n]: skip k-means (tests)
  double* out;                // [n_init][out_stride]
  double* scratch;            // [n_init][scratch_stride]
  long long out_stride, scratch_stride;
};

__device__ __forceinline__ int gm_slot(int f) {
  const int j = f >> 3, r = f & 7;
  return 4 * (2 * j + (r & 1)) + (r >> 1);
}

// deterministic block sum: shuffle tree inside a wave, fixed order across waves
__device__ __forceinline__ double block_sum(double v, double* red, int wave,
                                            int lane) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();                       // red free
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < GM_WAVES; ++w) s += red[w];
  return s;
}

// S = sum_p w_p q_p q_p^T, q = (x, 1) (m = d + 1), lower block triangle valid:
// read entry (r, c) as S[max(r,c) * m + min(r,c)]
template <int DT>
__device__ __forceinline__ void syrk_aug(const double* __restrict__ x, int n,
                                         int d, const volatile double* w,
                                         double* S, int wave, int lane) {
  const int m = d + 1;
  nb_d4 acc[DT * (DT + 1) / 2];
#pragma unroll
  for (int q = 0; q < DT * (DT + 1) / 2; ++q) acc[q] = nb_d4{0.0, 0.0, 0.0, 0.0};
  const int kp = lane >> 4, fi = lane & 15;
  for (int s = wave; 4 * s < n; s += GM_WAVES) {
    const int p = 4 * s + kp;
    const bool on = p < n;
    const double wp = on ? (w != nullptr ? w[p] : 1.0) : 0.0;
    double a[DT], aw[DT];
#pragma unroll
    for (int ft = 0; ft < DT; ++ft) {
      const int f = 16 * ft + fi;
      double v = 0.0;
      if (on && f < d) v = x[(long long)p * d + f];
      else if (on && f == d) v = 1.0;
      a[ft] = v;
      aw[ft] = v * wp;
    }
    int q = 0;
#pragma unroll
    for (int it = 0; it < DT; ++it)
#pragma unroll
      for (int jt = 0; jt <= it; ++jt) {
        acc[q] = MFMA(aw[it], a[jt], acc[q]);
        ++q;
      }
  }
  for (int w8 = 0; w8 < GM_WAVES; ++w8) {
    if (wave == w8) {
      int q = 0;
#pragma unroll
      for (int it = 0; it < DT; ++it)
#pragma unroll
        for (int jt = 0; jt <= it; ++jt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * it + (lane >> 4) + 4 * r;
            const int col = 16 * jt + (lane & 15);
            if (row < m && col < m) {
              double* dst = &S[row * m + col];
              *dst = (w8 == 0 ? 0.0 : *dst) + acc[q][r];
            }
          }
          ++q;
        }
    }
    __syncthreads();
  }
}

template <int DT>
__global__ void __launch_bounds__(GM_THREADS)
nb_gmm_kernel(GmmArgs a) {
  constexpr int DP = 16 * DT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[GM_WAVES], piv[64], cen[2][64], cpart[GM_WAVES][2][64];
  __shared__ double sh_val[4];
  __shared__ int sh_idx[2], sh_bad;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4;
  const int init = blockIdx.x;
  const double* __restrict__ x = a.x;
  const int n = a.n, d = a.d, m = d + 1;
  const int mm = (m * m + 1) & ~1;
  double* S = lds;             // weighted second moments (m x m)
  double* A = S + mm;          // covariance being factorised (d x d)
  double* B = A + mm;          // L^-1
  double* ell = B + mm;        // ell block for the matrix-core pass
  double* tiles = ell + 2 + 3 * DP;
  const double inf = __builtin_huge_val();

  double* out = a.out + (long long)init * a.out_stride;
  double* scr = a.scratch + (long long)init * a.scratch_stride;
  volatile double* sall = scr;                       // [m*m]
  volatile double* lp0 = scr + mm;                   // [n]
  volatile double* lp1 = lp0 + n;
  volatile double* r0 = lp1 + n;
  volatile double* d2 = r0 + n;
  volatile int* lab = (volatile int*)(d2 + n);       // [n] ints
  double* o_mean = out + 6;                          // [2][d]
  double* o_cov = o_mean + 2 * d;                    // [2][d*d]

  int er[GM_EPT], ec[GM_EPT];
#pragma unroll
  for (int q = 0; q < GM_EPT; ++q) {
    const int e = tid + q * GM_THREADS;
    er[q] = (e < d * d) ? e / d : -1;
    ec[q] = (e < d * d) ? e - er[q] * d : 0;
  }

  for (int e = tid; e < nb_ell_block_size(DT); e += GM_THREADS) ell[e] = 0.0;
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  if (tid == 0) ((long long*)ell)[0] = d;
  for (int f = tid; f < DP; f += GM_THREADS) {
    ell[2 + f] = -inf;
    ell[2 + DP + f] = inf;
  }
  __syncthreads();

  // S_all = sum q q^T, kept in global scratch
  syrk_aug<DT>(x, n, d, nullptr, S, wave, lane);
  for (int e = tid; e < m * m; e += GM_THREADS) sall[e] = S[e];
  // mean feature variance (tolerance scale of k-means, cluster/_kmeans.py:_tolerance)
  double mean_var = 0.0;
  for (int f = 0; f < d; ++f) {
    const double mu = S[d * m + f] / n;
    mean_var += S[f * m + f] / n - mu * mu;
  }
  mean_var /= d;
  __syncthreads();

  // ---- initial hard assignment ---------------------------------------------
  if (a.init_labels != nullptr) {
    for (int i = tid; i < n; i += GM_THREADS)
      r0[i] = a.init_labels[(long long)init * n + i] == 0 ? 1.0 : 0.0;
  } else {
    double u0, u1, u2, u3;
    nb_uniform_pair(a.seed, (unsigned long long)init, 0u, GM_TAG, u0, u1);
    nb_uniform_pair(a.seed, (unsigned long long)init, 1u, GM_TAG, u2, u3);
    int first = (int)(u0 * n);
    if (first > n - 1) first = n - 1;
    if (tid < d) cen[0][tid] = x[(long long)first * d + tid];
    __syncthreads();
    // squared distances to the first centre, k-means++ potential
    double part = 0.0;
    for (int i = tid; i < n; i += GM_THREADS) {
      double s = 0.0;
      for (int f = 0; f < d; ++f) {
        const double t = x[(long long)i * d + f] - cen[0][f];
        s += t * t;
      }
      d2[i] = s;
      part += s;
    }
    __threadfence_block();
    const double pot = block_sum(part, red, wave, lane);
    // two candidates drawn with probability ~ d2 (greedy k-means++,
    // cluster/_kmeans.py:_kmeans_plusplus with n_local_trials = 2)
    const int chunk = (n + GM_THREADS - 1) / GM_THREADS;
    const int lo = tid * chunk;
    const int hi = (lo + chunk < n) ? lo + chunk : n;
    double csum = 0.0;
    for (int i = lo; i < hi; ++i) csum += d2[i];
    // inclusive scan of the chunk sums: inside the wave, then across waves
    double inc = csum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double t = __shfl_up(inc, o);
      if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) red[wave] = inc;
    if (tid < 2) sh_idx[tid] = n - 1;
    __syncthreads();
    double before = inc - csum;
    for (int w = 0; w < wave; ++w) before += red[w];
    for (int t = 0; t < 2; ++t) {
      const double target = (t == 0 ? u1 : u2) * pot;
      if (target >= before && target < before + csum) {
        double run = before;
        int pick = hi - 1;
        for (int i = lo; i < hi; ++i) {
          run += d2[i];
          if (run > target) { pick = i; break; }
        }
        sh_idx[t] = pick;
      }
    }
    __syncthreads();
    double cand_pot[2];
    for (int t = 0; t < 2; ++t) {
      const int ci = sh_idx[t];
      double pp = 0.0;
      for (int i = tid; i < n; i += GM_THREADS) {
        double s = 0.0;
        for (int f = 0; f < d; ++f) {
          const double tt = x[(long long)i * d + f] - x[(long long)ci * d + f];
          s += tt * tt;
        }
        const double old = d2[i];
        pp += s < old ? s : old;
      }
      cand_pot[t] = block_sum(pp, red, wave, lane);
    }
    const int second = sh_idx[cand_pot[1] < cand_pot[0] ? 1 : 0];
    __syncthreads();
    if (tid < d) cen[1][tid] = x[(long long)second * d + tid];
    for (int i = tid; i < n; i += GM_THREADS) lab[i] = -1;
    __threadfence_block();
    __syncthreads();

    // Lloyd iterations (cluster/_kmeans.py:_kmeans_single_lloyd)
    for (int it = 0; it < 300; ++it) {
      double changed = 0.0;
      for (int i = tid; i < n; i += GM_THREADS) {
        double s0 = 0.0, s1 = 0.0;
        for (int f = 0; f < d; ++f) {
          const double xv = x[(long long)i * d + f];
          const double t0 = xv - cen[0][f], t1 = xv - cen[1][f];
          s0 += t0 * t0;
          s1 += t1 * t1;
        }
        const int l = s1 < s0 ? 1 : 0;
        if (lab[i] != l) changed += 1.0;
        lab[i] = l;
      }
      __threadfence_block();
      changed = block_sum(changed, red, wave, lane);
      __syncthreads();
      // new centres: (feature, chunk of points) decomposition
      {
        const int f = lane, ch = wave;
        const int per = (n + GM_WAVES - 1) / GM_WAVES;
        const int i0 = ch * per, i1 = (i0 + per < n) ? i0 + per : n;
        double s0 = 0.0, s1 = 0.0, c1 = 0.0;
        for (int i = i0; i < i1; ++i) {
          const int l = lab[i];
          const double xv = (f < d) ? x[(long long)i * d + f] : 0.0;
          if (l) { s1 += xv; c1 += 1.0; } else s0 += xv;
        }
        cpart[ch][0][f] = s0;
        cpart[ch][1][f] = s1;
        if (f == 0) red[ch] = c1;
      }
      __syncthreads();
      double n1 = 0.0;
      for (int w = 0; w < GM_WAVES; ++w) n1 += red[w];
      const double n0 = n - n1;
      if (n0 < 1.0 || n1 < 1.0) {            // an empty cluster: give up
        if (tid == 0) sh_bad = 1;
        __syncthreads();
        break;
      }
      double shift = 0.0;
      if (tid < 2 * 64) {
        const int k = tid >> 6, f = tid & 63;
        double s = 0.0;
        for (int w = 0; w < GM_WAVES; ++w) s += cpart[w][k][f];
        const double c_new = (f < d) ? s / (k ? n1 : n0) : 0.0;
        const double dlt = c_new - ((f < d) ? cen[k][f] : 0.0);
        shift = dlt * dlt;
        cen[k][f] = c_new;     // only this thread reads or writes cen[k][f] here
      }
      shift = block_sum(shift, red, wave, lane);
      __syncthreads();
      if (changed == 0.0 || shift <= 1e-4 * mean_var) break;
    }
    __syncthreads();
    for (int i = tid; i < n; i += GM_THREADS) r0[i] = lab[i] == 0 ? 1.0 : 0.0;
  }
  __threadfence_block();
  __syncthreads();

  // ---- EM ------------------------------------------------------------------
  const double eps10 = 10.0 * 2.220446049250313e-16;
  double lower = -inf;
  int n_iter = 0, converged = 0;
  double pi0 = 0.5, pi1 = 0.5;
  bool failed = sh_bad != 0;
  for (int it = 0; it <= a.max_iter && !failed; ++it) {
    // M-step (mixture/_gaussian_mixture.py:_estimate_gaussian_parameters)
    syrk_aug<DT>(x, n, d, r0, S, wave, lane);
    const double nk0 = S[d * m + d] + eps10;
    const double nk1 = ((double)n - S[d * m + d]) + eps10;
    pi0 = nk0 / (nk0 + nk1);      // _m_step: weights_ /= weights_.sum()
    pi1 = nk1 / (nk0 + nk1);
    if (tid < d) {
      o_mean[tid] = S[d * m + tid] / nk0;
      o_mean[d + tid] = (sall[d * m + tid] - S[d * m + tid]) / nk1;
    }
    __threadfence_block();
    __syncthreads();
#pragma unroll
    for (int q = 0; q < GM_EPT; ++q) {
      const int r = er[q], c = ec[q];
      if (r < 0) continue;
      const int hi_ = r > c ? r : c, lo_ = r > c ? c : r;
      const double s0 = S[hi_ * m + lo_];
      const double s1 = sall[hi_ * m + lo_] - s0;
      const double reg = (r == c) ? a.reg : 0.0;
      const double m0r = ((volatile double*)o_mean)[r], m0c = ((volatile double*)o_mean)[c];
      const double m1r = ((volatile double*)o_mean)[d + r], m1c = ((volatile double*)o_mean)[d + c];
      o_cov[r * d + c] = s0 / nk0 - m0r * m0c + reg;
      o_cov[d * d + r * d + c] = s1 / nk1 - m1r * m1c + reg;
    }
    __threadfence_block();
    __syncthreads();
    if (it == a.max_iter || converged) break;

    // E-step (mixture/_base.py:_estimate_log_prob_resp)
    for (int k = 0; k < 2; ++k) {
      const volatile double* cov = (volatile double*)o_cov + k * d * d;
      const volatile double* mean = (volatile double*)o_mean + k * d;
#pragma unroll
      for (int q = 0; q < GM_EPT; ++q)
        if (er[q] >= 0) {
          A[er[q] * d + ec[q]] = cov[er[q] * d + ec[q]];
          B[er[q] * d + ec[q]] = (er[q] == ec[q]) ? 1.0 : 0.0;
        }
      __syncthreads();
      for (int p = 0; p < d - 1; ++p) {
        const double inv_d = 1.0 / A[p * d + p];
#pragma unroll
        for (int q = 0; q < GM_EPT; ++q) {
          const int i = er[q], j = ec[q];
          if (i > p) {
            const double f = A[i * d + p] * inv_d;
            if (j > p) A[i * d + j] -= f * A[p * d + j];
            else B[i * d + j] -= f * B[p * d + j];
          }
        }
        __syncthreads();
      }
      if (tid < 64) {
        const double dv = (tid < d) ? A[tid * d + tid] : 1.0;
        if (!(dv > 0.0)) sh_bad = 1;           // not positive definite
        piv[tid] = 1.0 / dv;
        double ld = (tid < d) ? log(dv) : 0.0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ld += __shfl_xor(ld, o);
        if (tid == 0) sh_val[k] = ld;
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < GM_EPT; ++q) {
        const int r = er[q], c = ec[q];
        if (r < 0 || c > r) continue;
        const int sl = gm_slot(c), ks = sl >> 2, lgk = sl & 3;
        tiles[((ks >> 2) * DT + (r >> 4)) * NB_TILE + (ks & 3) * 64 + lgk * 16 +
              (r & 15)] = B[r * d + c] * sqrt(piv[r]);
      }
      if (tid < DP) ell[2 + 2 * DP + gm_slot(tid)] = (tid < d) ? mean[tid] : 0.0;
      __syncthreads();
      const double konst = log(k == 0 ? pi0 : pi1) -
                           0.5 * (d * 1.8378770664093453 + sh_val[k]);
      volatile double* lp = k == 0 ? lp0 : lp1;
      for (int tile = wave; tile * 16 < n; tile += GM_WAVES) {
        long long pt[1] = {(long long)tile * 16 + (lane & 15)};
        bool valid[1] = {pt[0] < n};
        double xin[1][4 * DT], y[1][4 * DT], r2[1];
        bool box_bad[1];
        load_points<DT, 1>(x, pt, valid, d, (long long)n, lane, xin);
        ell_eval<DT, 1>(ell, d, xin, lane, y, box_bad, r2);
        if (valid[0] && lg == 0) lp[pt[0]] = konst - 0.5 * r2[0];
      }
      __threadfence_block();
      __syncthreads();
    }
    failed = sh_bad != 0;
    double part = 0.0;
    for (int i = tid; i < n; i += GM_THREADS) {
      const double l0 = lp0[i], l1 = lp1[i];
      const double mx = l0 > l1 ? l0 : l1;
      const double lse = mx + log(exp(l0 - mx) + exp(l1 - mx));
      r0[i] = exp(l0 - lse);
      part += lse;
    }
    __threadfence_block();
    const double lb = block_sum(part, red, wave, lane) / n;
    __syncthreads();
    n_iter = it + 1;
    if (fabs(lb - lower) < a.tol) converged = 1;
    lower = lb;
  }
  if (tid == 0) {
    out[0] = failed ? -inf : lower;
    out[1] = n_iter;
    out[2] = converged;
    out[3] = failed ? 1.0 : 0.0;
    out[4] = pi0;
    out[5] = pi1;
  }
}

template <int DT>
int launch_gmm(const GmmArgs& a, hipStream_t stream) {
  const int m = a.d + 1;
  const int mm = (m * m + 1) & ~1;
  const size_t lds = ((size_t)3 * mm + nb_ell_block_size(DT)) * sizeof(double);
  static size_t allowed = 0;
  if (lds > allowed) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_gmm_kernel<DT>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    allowed = lds;
  }
  (void)hipGetLastError();
  hipLaunchKernelGGL(nb_gmm_kernel<DT>, dim3(a.n_init), dim3(GM_THREADS), lds,
                     stream, a);
  return NB_OK;
}

}  // namespace

long long nb_gmm_out_stride_impl(int d) { return 6 + 2LL * d + 2LL * d * d; }
long long nb_gmm_scratch_stride_impl(long long n, int d) {
  const int m = d + 1;
  const long long mm = (m * m + 1) & ~1;
  return mm + 5 * n + 2;
}

int nb_launch_gmm(const double* x, long long n, int d, int n_init,
                  unsigned long long seed, double tol, double reg, int max_iter,
                  const int* init_labels, double* out, double* scratch,
                  hipStream_t stream) {
  if (d < 1 || d + 1 > 64) {
    nb_set_error("device mixture fit supports n_dim <= 63 (got %d)", d);
    return NB_ERR_UNSUPPORTED;
  }
  if (n < 2 || n > 100000000LL || n_init < 1 || max_iter < 0) {
    nb_set_error("bad mixture fit arguments");
    return NB_ERR_ARG;
  }
  GmmArgs a;
  a.x = x; a.n = (int)n; a.d = d; a.n_init = n_init; a.seed = seed;
  a.tol = tol; a.reg = reg; a.max_iter = max_iter; a.init_labels = init_labels;
  a.out = out; a.scratch = scratch;
  a.out_stride = nb_gmm_out_stride_impl(d);
  a.scratch_stride = nb_gmm_scratch_stride_impl(n, d);
  const int dt = (d + 1 + 15) / 16;
  int rc = NB_OK;
  switch (dt) {
    case 1: rc = launch_gmm<1>(a, stream); break;
    case 2: rc = launch_gmm<2>(a, stream); break;
    case 3: rc = launch_gmm<3>(a, stream); break;
    default: rc = launch_gmm<4>(a, stream); break;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
