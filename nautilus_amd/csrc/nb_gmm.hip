// Two-component full-covariance Gaussian mixture for Union.split (reference
// nautilus/bounds/union.py:185-187: scikit-learn GaussianMixture(n_components
// =2, n_init=10), the reference's third-party dependency for this step).  The
// algorithm restated here is scikit-learn 1.7's: k-means++ / Lloyd
// initialisation (mixture/_base.py:_initialize_parameters, cluster/_kmeans.py),
// then EM with tol = 1e-3 on the mean log-likelihood, reg_covar = 1e-6,
// max_iter = 100 (mixture/_base.py:fit_predict, _gaussian_mixture.py).
//
// One workgroup per restart, all restarts of a fit run concurrently;
// n_dim <= 128.
//   M-step: ONE weighted second-moment product on the matrix cores over the
//           augmented rows q = (x, 1):  S0 = sum_i r_i0 q_i q_i^T  holds
//           sum r x x^T, sum r x and sum r at once (nb_sym.h, sy_moments);
//           component 1 follows from S_all - S0 (S_all computed once)
//   E-step: Sigma_k is built in LDS as lower-triangular operand tiles and
//           inverted in place with the symmetric sweep operator (the pivots
//           give log det Sigma_k); (x - mu_k)^T Sigma_k^-1 (x - mu_k) for all
//           points on the matrix cores (sy_quadform)
// Everything is deterministic (fixed reduction orders, Philox for the seeding).
#include "nb_sym.h"

namespace {

constexpr int GM_THREADS = 512;
constexpr int GM_WAVES = GM_THREADS / 64;
constexpr unsigned GM_TAG = 3u;           // Philox tag of the seeding draws
// both components' operand tiles side by side in LDS (2 x DT (DT + 1) / 2
// tiles of 2 KB): up to DT = 7 (112 KB)
__host__ __device__ constexpr bool gm_both(int dt) { return dt <= 7; }

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

// DT = ceil((d + 1) / 16): tiles of the augmented rows
template <int DT>
__global__ void __launch_bounds__(GM_THREADS)
nb_gmm_kernel(GmmArgs a) {
  constexpr int DP = 16 * DT;
  constexpr int NT = DT * (DT + 1) / 2;
  constexpr int GW = (DT + 1) / 2;          // waves per moment sub-group
  constexpr int SG = GM_WAVES / GW;
  constexpr bool BOTH = gm_both(DT);        // two sets of operand tiles in LDS
  constexpr int NKT = BOTH ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[GM_WAVES], sh_val[4];
  __shared__ int sh_idx[2], sh_bad;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lj = lane & 15;
  const int init = blockIdx.x;
  const double* __restrict__ x = a.x;
  const int n = a.n, d = a.d, m = d + 1;
  const int mm = (m * m + 1) & ~1;
  double* T = lds;                         // NKT x NT tiles: Sigma_k -> -Sigma_k^-1 -> T
  double* colk = T + NKT * NT * NB_TILE;   // [NKT][DP] pivot columns of the sweeps
  double* mus = colk + NKT * DP;           // [NKT][DP] mu_k in slot order
  double* cen = mus + NKT * DP;            // [2][DP] k-means centres
  unsigned short* swt = (unsigned short*)(cen + 2 * DP);   // [d (d + 1) / 2]
  double* cpart = T;                       // [GM_WAVES][2][DP] (seeding only)
  const double inf = __builtin_huge_val();

  double* out = a.out + (long long)init * a.out_stride;
  double* scr = a.scratch + (long long)init * a.scratch_stride;
  volatile double* sall = scr;                       // [m*m]
  volatile double* lp0 = scr + mm;                   // [n]
  volatile double* lp1 = lp0 + n;
  volatile double* r0 = lp1 + n;
  volatile double* d2 = r0 + n;
  volatile int* lab = (volatile int*)(d2 + n);       // [n] ints
  double* part = scr + mm + 5LL * n + 2;             // [SG][NT][256]
  double* o_mean = out + 6;                          // [2][d]
  double* o_cov = o_mean + 2 * d;                    // [2][d*d]

  if (tid == 0) sh_bad = 0;
  __syncthreads();
#ifdef NB_GMM_TIMING
  // cycle counts per phase of restart 0 (debug build, make debug
  // DEFS=-DNB_GMM_TIMING): S_all, seeding + Lloyd, then per EM phase
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = wall_clock64();
#define GM_STAMP(i) do { const long long t_now = wall_clock64(); \
    tk[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define GM_STAMP(i) do {} while (0)
#endif

  // weighted second moments of all points into `part` (every wave of a
  // sub-group owns two tile rows)
  auto moments = [&](const volatile double* w) {
    const int sg = wave / GW, wv = wave - sg * GW;
    if (sg < SG)
      // (a shorter ring than the stand-alone moment kernel's: this kernel
      // keeps more alive around it)
      sy_moments<DT, (DT <= 2 ? 8 : (DT <= 4 ? 4 : 2))>(x, w, d, 0, n, sg, SG, wv, lane,
                     part + (size_t)sg * NT * NB_TILE);
    __threadfence_block();
    __syncthreads();
  };
  auto moment = [&](int r, int c) {        // entry (r, c), r >= c
    return mom_element((const volatile double*)part, SG, NT, r, c);
  };

  // S_all = sum q q^T, kept in global scratch (lower triangle + mirror)
  moments(nullptr);
  for (int e = tid; e < m * m; e += GM_THREADS) {
    const int r = e / m, c = e - r * m;
    sall[e] = moment(r > c ? r : c, r > c ? c : r);
  }
  __threadfence_block();
  __syncthreads();
  // mean feature variance (tolerance scale of k-means, cluster/_kmeans.py:_tolerance)
  double mean_var = 0.0;
  for (int f = 0; f < d; ++f) {
    const double mu = sall[d * m + f] / n;
    mean_var += sall[f * m + f] / n - mu * mu;
  }
  mean_var /= d;
  GM_STAMP(0);

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
    for (int f = tid; f < d; f += GM_THREADS)
      cen[f] = x[(long long)first * d + f];
    __syncthreads();
    // squared distances to the first centre, k-means++ potential
    double pot_part = 0.0;
    for (int i = tid; i < n; i += GM_THREADS) {
      double s = 0.0;
      for (int f = 0; f < d; ++f) {
        const double t = x[(long long)i * d + f] - cen[f];
        s += t * t;
      }
      d2[i] = s;
      pot_part += s;
    }
    __threadfence_block();
    const double pot = block_sum(pot_part, red, wave, lane);
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
    for (int f = tid; f < d; f += GM_THREADS)
      cen[DP + f] = x[(long long)second * d + f];
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
          const double t0 = xv - cen[f], t1 = xv - cen[DP + f];
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
      // new centres: (feature, chunk of points) decomposition, the features
      // in passes of 64
      {
        const int ch = wave;
        const int per = (n + GM_WAVES - 1) / GM_WAVES;
        const int i0 = ch * per, i1 = (i0 + per < n) ? i0 + per : n;
        double c1 = 0.0;
        for (int fb = 0; fb < DP; fb += 64) {
          const int f = fb + lane;
          double s0 = 0.0, s1 = 0.0;
          c1 = 0.0;
          for (int i = i0; i < i1; ++i) {
            const int l = lab[i];
            const double xv = (f < d) ? x[(long long)i * d + f] : 0.0;
            if (l) { s1 += xv; c1 += 1.0; } else s0 += xv;
          }
          if (f < DP) {
            cpart[(ch * 2 + 0) * DP + f] = s0;
            cpart[(ch * 2 + 1) * DP + f] = s1;
          }
        }
        if (lane == 0) red[ch] = c1;
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
      if (tid < 2 * DP) {
        const int k = tid / DP, f = tid - k * DP;
        double s = 0.0;
        for (int w = 0; w < GM_WAVES; ++w) s += cpart[(w * 2 + k) * DP + f];
        const double c_new = (f < d) ? s / (k ? n1 : n0) : 0.0;
        const double dlt = c_new - ((f < d) ? cen[k * DP + f] : 0.0);
        shift = dlt * dlt;
        cen[k * DP + f] = c_new;   // only this thread touches cen[k][f] here
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
  GM_STAMP(1);

  // the entries (r >= c) of the lower triangle, packed (r << 8 | c), in LDS:
  // thread t owns the entries t, t + 512, ... of every sweep (a table in
  // registers -- 17 slots of three ints for n_dim 128 -- was live across the
  // whole EM loop and evaluated in full at every pivot whatever n_dim)
  const int n_low = d * (d + 1) / 2;
  for (int e = tid; e < n_low; e += GM_THREADS) {
    int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > e) --r;
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    swt[e] = (unsigned short)((r << 8) | (e - r * (r + 1) / 2));
  }
  __syncthreads();

  // ---- EM ------------------------------------------------------------------
  const double eps10 = 10.0 * 2.220446049250313e-16;
  double lower = -inf;
  int n_iter = 0, converged = 0;
  double pi0 = 0.5, pi1 = 0.5;
  bool failed = sh_bad != 0;
  const int ks_max = 4 * DT;
  // weighted log probabilities of both components for all points with the
  // current parameters (mixture/_base.py:_estimate_weighted_log_prob).  Both
  // components go through the pivots TOGETHER where two sets of operand
  // tiles fit the LDS (BOTH: n_dim <= 111): one pair of barriers per pivot
  // for both sweeps, and the quadratic forms of both read every point once.
  constexpr int NK = BOTH ? 2 : 1;
  auto log_prob = [&]() {
    for (int k0 = 0; k0 < 2; k0 += NK) {
      for (int e = tid; e < NK * NT * NB_TILE; e += GM_THREADS) T[e] = 0.0;
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        const int k = k0 + kk;
        const volatile double* cov = (volatile double*)o_cov + k * d * d;
        const volatile double* mean = (volatile double*)o_mean + k * d;
        double* Tk = T + kk * NT * NB_TILE;
        for (int e = tid; e < n_low; e += GM_THREADS) {
          const int r = swt[e] >> 8, c = swt[e] & 255;
          Tk[sy_pos(r, c)] = cov[r * d + c];
        }
        for (int f = tid; f < DP; f += GM_THREADS)
          mus[kk * DP + mv_slot(f)] = (f < d) ? mean[f] : 0.0;
      }
      __syncthreads();
      GM_STAMP(3);
      // symmetric sweep operator over all pivots: T <- -Sigma^-1, the pivots
      // are those of the L D L^T factorisation (log det = sum log d_p)
      double logdet[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) logdet[kk] = 0.0;
      for (int p = 0; p < d; ++p) {
        for (int i = tid; i < NK * d; i += GM_THREADS) {
          const int kk = i >= d ? 1 : 0, ii = i - kk * d;
          colk[kk * DP + ii] = T[kk * NT * NB_TILE +
                                 (ii >= p ? sy_pos(ii, p) : sy_pos(p, ii))];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          double* Tk = T + kk * NT * NB_TILE;
          const double* ck = colk + kk * DP;
          const double dp = ck[p];
          if (!(dp > 0.0)) {                   // not positive definite
            if (tid == 0) sh_bad = 1;
          }
          const double inv_d = 1.0 / dp;
          logdet[kk] += log(dp);
          for (int e = tid; e < n_low; e += GM_THREADS) {
            const int r = swt[e] >> 8, c = swt[e] & 255;
            const int at = sy_pos(r, c);
            double v;
            if (r == p && c == p) v = -inv_d;
            else if (r == p) v = ck[c] * inv_d;
            else if (c == p) v = ck[r] * inv_d;
            else v = Tk[at] - ck[r] * ck[c] * inv_d;
            Tk[at] = v;
          }
        }
        __syncthreads();
      }
      GM_STAMP(4);
      // -> T form: -(...) and doubled off-diagonal entries
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
        for (int e = tid; e < n_low; e += GM_THREADS) {
          const int r = swt[e] >> 8, c = swt[e] & 255;
          T[kk * NT * NB_TILE + sy_pos(r, c)] *= (r == c) ? -1.0 : -2.0;
        }
      __syncthreads();
      double konst[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
        konst[kk] = log(k0 + kk == 0 ? pi0 : pi1) -
                    0.5 * (d * 1.8378770664093453 + logdet[kk]);
      for (int tile = wave; tile * 16 < n; tile += GM_WAVES) {
        long long pt[1] = {(long long)tile * 16 + lj};
        bool valid[1] = {pt[0] < n};
        double xin[1][4 * DT];
        load_points<DT, 1>((const nb_gd*)x, pt, valid, d, (long long)n, lane, xin);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          double dx[4 * DT];
#pragma unroll
          for (int ks = 0; ks < ks_max; ++ks)
            dx[ks] = valid[0] ? xin[0][ks] - mus[kk * DP + 4 * ks + lg] : 0.0;
          const double g = sy_quadform<DT>(T + kk * NT * NB_TILE, dx, d, lane);
          volatile double* lp = (k0 + kk) == 0 ? lp0 : lp1;
          if (valid[0] && lg == 0) lp[pt[0]] = konst[kk] - 0.5 * g;
        }
      }
      __threadfence_block();
      __syncthreads();
      GM_STAMP(5);
    }
  };
  for (int it = 0; it <= a.max_iter && !failed; ++it) {
    // M-step (mixture/_gaussian_mixture.py:_estimate_gaussian_parameters)
    moments(r0);
    GM_STAMP(2);
    const double s00 = moment(d, d);
    const double nk0 = s00 + eps10;
    const double nk1 = ((double)n - s00) + eps10;
    pi0 = nk0 / (nk0 + nk1);      // _m_step: weights_ /= weights_.sum()
    pi1 = nk1 / (nk0 + nk1);
    for (int f = tid; f < d; f += GM_THREADS) {
      const double s0 = moment(d, f);
      o_mean[f] = s0 / nk0;
      o_mean[d + f] = (sall[d * m + f] - s0) / nk1;
    }
    __threadfence_block();
    __syncthreads();
    for (int e = tid; e < d * d; e += GM_THREADS) {
      const int r = e / d, c = e - r * d;
      if (c > r) continue;
      const double s0 = moment(r, c);
      const double s1 = sall[r * m + c] - s0;
      const double reg = (r == c) ? a.reg : 0.0;
      const double m0r = ((volatile double*)o_mean)[r], m0c = ((volatile double*)o_mean)[c];
      const double m1r = ((volatile double*)o_mean)[d + r], m1c = ((volatile double*)o_mean)[d + c];
      const double c0 = s0 / nk0 - m0r * m0c + reg;
      const double c1 = s1 / nk1 - m1r * m1c + reg;
      o_cov[r * d + c] = c0;
      o_cov[c * d + r] = c0;
      o_cov[d * d + r * d + c] = c1;
      o_cov[d * d + c * d + r] = c1;
    }
    __threadfence_block();
    __syncthreads();
    GM_STAMP(3);
    if (it == a.max_iter || converged) break;

    // E-step (mixture/_base.py:_estimate_log_prob_resp)
    log_prob();
    failed = sh_bad != 0;
    double lse_part = 0.0;
    for (int i = tid; i < n; i += GM_THREADS) {
      const double l0 = lp0[i], l1 = lp1[i];
      const double mx = l0 > l1 ? l0 : l1;
      const double lse = mx + log(exp(l0 - mx) + exp(l1 - mx));
      r0[i] = exp(l0 - lse);
      lse_part += lse;
    }
    __threadfence_block();
    const double lb = block_sum(lse_part, red, wave, lane) / n;
    __syncthreads();
    GM_STAMP(6);
    n_iter = it + 1;
    if (fabs(lb - lower) < a.tol) converged = 1;
    lower = lb;
  }
  // the log probabilities under the FINAL parameters (those of the last
  // M-step) stay in the scratch of this restart: Union.split assigns every
  // point to its more probable component (bounds/union.py:188-197)
  if (!failed) {
    log_prob();
    failed = sh_bad != 0;
  }
#ifdef NB_GMM_TIMING
  if (tid == 0 && init == 0)
    printf("[gmm] n=%d d=%d iters=%d  ticks(100MHz): s_all %lld seed+lloyd %lld "
           "moments %lld mstep/build %lld sweeps %lld quadform %lld estep %lld\n",
           n, d, n_iter, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6]);
#endif
  if (tid == 0) {
    out[0] = failed ? -inf : lower;
    out[1] = n_iter;
    out[2] = converged;
    out[3] = failed ? 1.0 : 0.0;
    out[4] = pi0;
    out[5] = pi1;
  }
}

inline int gm_dt(int d) { return (d + 1 + 15) / 16; }
inline size_t gm_lds_doubles(int dt) {
  const size_t nk = gm_both(dt) ? 2 : 1;
  const size_t tiles = nk * dt * (dt + 1) / 2 * NB_TILE;
  const size_t seed = (size_t)GM_WAVES * 2 * 16 * dt;      // cpart aliases T
  // ... pivot columns, means, k-means centres, table of the triangle's entries
  return (tiles > seed ? tiles : seed) + (2 * nk + 2) * 16 * dt +
         ((size_t)16 * dt * (16 * dt + 1) / 2 + 3) / 4;
}

template <int DT>
int launch_gmm(const GmmArgs& a, hipStream_t stream) {
  const size_t lds = gm_lds_doubles(DT) * sizeof(double);
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
  const int m = d + 1, dt = gm_dt(d);
  const long long mm = (m * m + 1) & ~1;
  const long long part = (long long)(GM_WAVES / ((dt + 1) / 2)) *
                         (dt * (dt + 1) / 2) * NB_TILE;
  return mm + 5 * n + 2 + part;
}

int nb_launch_gmm(const double* x, long long n, int d, int n_init,
                  unsigned long long seed, double tol, double reg, int max_iter,
                  const int* init_labels, double* out, double* scratch,
                  hipStream_t stream) {
  if (d < 1 || d > 128) {
    nb_set_error("device mixture fit supports n_dim <= 128 (got %d)", d);
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
  int rc = NB_OK;
  switch (gm_dt(d)) {
    case 1: rc = launch_gmm<1>(a, stream); break;
    case 2: rc = launch_gmm<2>(a, stream); break;
    case 3: rc = launch_gmm<3>(a, stream); break;
    case 4: rc = launch_gmm<4>(a, stream); break;
    case 5: rc = launch_gmm<5>(a, stream); break;
    case 6: rc = launch_gmm<6>(a, stream); break;
    case 7: rc = launch_gmm<7>(a, stream); break;
    case 8: rc = launch_gmm<8>(a, stream); break;
    default: rc = launch_gmm<9>(a, stream); break;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
