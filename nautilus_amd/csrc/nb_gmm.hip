// Two-component full-covariance Gaussian mixture for Union.split (reference
// nautilus/bounds/union.py:185-187: scikit-learn GaussianMixture(n_components
// =2, n_init=10), the reference's third-party dependency for this step).  The
// algorithm restated here is scikit-learn 1.7's: k-means++ / Lloyd
// initialisation (mixture/_base.py:_initialize_parameters, cluster/_kmeans.py),
// then EM with tol = 1e-3 on the mean log-likelihood, reg_covar = 1e-6,
// max_iter = 100 (mixture/_base.py:fit_predict, _gaussian_mixture.py).
//
// All restarts of a fit run concurrently, and every restart on W WORKGROUPS
// (grid = (W, n_init), W <= 8; n_dim <= 128): an fp64 MFMA takes 64 cycles on
// this part, so ONE compute unit needs ~40 us for the moment product and
// ~75 us for the two quadratic forms of 2000 points at n_dim 50 -- per EM
// iteration, of which a unimodal cloud takes up to 100.  The rows are dealt
// out over the workgroups of a restart; what depends on all rows crosses
// them once per iteration:
//   M-step: every workgroup forms the weighted second moments of ITS rows on
//           the matrix cores over the augmented rows q = (x, 1):
//           S0 = sum_i r_i0 q_i q_i^T holds sum r x x^T, sum r x and sum r at
//           once (nb_sym.h, sy_moments); the per-workgroup sums and the sum
//           of log-likelihoods of the E-step before go to global memory, ONE
//           barrier over the restart's workgroups (a counter in the L2),
//           then every workgroup adds the W partial sums in the same order
//           and derives the same parameters (component 1 from S_all - S0,
//           S_all computed once the same way) -- redundantly, bit-identical
//   E-step: every workgroup builds Sigma_k in its LDS as lower-triangular
//           operand tiles and inverts it in place with the symmetric sweep
//           operator (the pivots give log det Sigma_k; both components go
//           through the pivots together where two tile sets fit the LDS);
//           (x - mu_k)^T Sigma_k^-1 (x - mu_k) for ITS rows on the matrix
//           cores (sy_quadform), responsibilities and its share of the
//           log-likelihood sum
// The seeding (k-means++ and Lloyd, 0.3 ms of a fit) runs on workgroup 0 of
// the restart.  Everything is deterministic (fixed reduction orders, Philox
// for the seeding).
#include "nb_sym.h"

#include <atomic>
#include <cstdlib>

namespace {

constexpr int GM_THREADS = 512;
constexpr int GM_WAVES = GM_THREADS / 64;
constexpr unsigned GM_TAG = 3u;           // Philox tag of the seeding draws
constexpr int GM_MAXW = 8;                // workgroups per restart
constexpr int GM_SYNC_INTS = 32;          // counter block of a restart (128 B)
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
  int* sync;                  // [n_init][GM_SYNC_INTS], zero at launch
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

// scratch of one restart (doubles), sized for GM_MAXW workgroups
struct GmLayout {
  long long sall, lp0, lp1, r0, d2, lab, part_sg, part_wg, aux, kaux, total;
};
__host__ __device__ inline GmLayout gm_layout(long long n, int d) {
  const int m = d + 1, dt = (m + 15) / 16;
  const long long mm = (m * m + 1) & ~1;
  const long long nt = (long long)dt * (dt + 1) / 2 * NB_TILE;
  const long long sgs = GM_WAVES / ((dt + 1) / 2);
  const long long n2 = (n + 1) & ~1LL;
  GmLayout L;
  L.sall = 0;                                  // [GM_MAXW][mm] private copies
  L.lp0 = L.sall + GM_MAXW * mm;               // [n]
  L.lp1 = L.lp0 + n;                           // (contiguous: [2][n])
  L.r0 = (L.lp1 + n + 1) & ~1LL;
  L.d2 = L.r0 + n2;
  L.lab = L.d2 + n2;                           // [n] ints
  L.part_sg = L.lab + n2 / 2 + 2;              // [GM_MAXW][SG][NT][256]
  L.part_wg = L.part_sg + GM_MAXW * sgs * nt;  // [2][GM_MAXW][NT][256]
  L.aux = L.part_wg + 2 * GM_MAXW * nt;        // [2][GM_MAXW][8]
  L.kaux = L.aux + 2 * GM_MAXW * 8;            // [2][GM_MAXW][2 DP + 8]
  L.total = L.kaux + 2 * GM_MAXW * (2LL * 16 * dt + 8);
  return L;
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
  // entries of the lower triangle per thread (n_dim <= 16 DT - 1)
  constexpr int EPT_N = (DP - 1) * DP / 2 < 8256 ? (DP - 1) * DP / 2 : 8256;
  constexpr int EPT = (EPT_N + GM_THREADS - 1) / GM_THREADS;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ double red[GM_WAVES], sh_val[4];
  __shared__ int sh_idx[2], sh_bad;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lj = lane & 15;
  const int wg = blockIdx.x, W = gridDim.x;
  const int init = blockIdx.y;
  const double* __restrict__ x = a.x;
  const int n = a.n, d = a.d, m = d + 1;
  const int mm = (m * m + 1) & ~1;
  double* T = lds;                         // NKT x NT tiles: Sigma_k -> -Sigma_k^-1 -> T
  double* colk = T + NKT * NT * NB_TILE;   // [2][NKT][DP] pivot columns of the sweeps
  double* mus = colk + 2 * NKT * DP;       // [NKT][DP] mu_k in slot order
  double* cen = mus + NKT * DP;            // [2][DP] k-means centres
  double* mean_l = cen + 2 * DP;           // [2][DP] mu_k in feature order
  unsigned int* swt = (unsigned int*)(mean_l + 2 * DP);   // [d (d + 1) / 2]
  double* piv = (double*)(swt + ((DP * (DP + 1) / 2 + 1) & ~1));   // [NKT][DP] pivots
  double* cpart = T;                       // [GM_WAVES][2][DP] (seeding only)
  const double inf = __builtin_huge_val();

  // the rows of this workgroup (whole 16-point tiles)
  const int rows_per = (((n + W - 1) / W) + 15) / 16 * 16;
  const int row0 = wg * rows_per < n ? wg * rows_per : n;
  const int row1 = row0 + rows_per < n ? row0 + rows_per : n;

  double* out = a.out + (long long)init * a.out_stride;
  double* scr = a.scratch + (long long)init * a.scratch_stride;
  const GmLayout L = gm_layout(n, d);
  volatile double* sall = scr + L.sall + (long long)wg * mm;   // [m*m], private
  volatile double* lp0 = scr + L.lp0;                // [n]
  volatile double* lp1 = scr + L.lp1;
  volatile double* r0 = scr + L.r0;
  volatile double* d2 = scr + L.d2;
  volatile int* lab = (volatile int*)(scr + L.lab);  // [n] ints
  double* part_sg = scr + L.part_sg + (long long)wg * SG * NT * NB_TILE;
  volatile double* part_wg = scr + L.part_wg;        // [2][GM_MAXW][NT][256]
  volatile double* aux = scr + L.aux;                // [2][GM_MAXW][8]
  double* o_mean = out + 6;                          // [2][d]
  double* o_cov = o_mean + 2 * d;                    // [2][d*d]
  int* cnt = a.sync + init * GM_SYNC_INTS;

  if (tid == 0) sh_bad = 0;
  __syncthreads();
#ifdef NB_GMM_TIMING
  // cycle counts per phase of restart 0, workgroup 0 (debug build, make debug
  // DEFS=-DNB_GMM_TIMING)
  long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t_last = wall_clock64();
#define GM_STAMP(i) do { const long long t_now = wall_clock64(); \
    tk[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define GM_STAMP(i) do {} while (0)
#endif

  // barrier over the workgroups of this restart: the b-th one is open once
  // the counter has reached (b + 1) W.  What a workgroup wrote before it is
  // visible to the others' volatile (device-coherent) loads behind it.
  int n_bar = 0;
  auto restart_barrier = [&]() {
    __syncthreads();
    if (W > 1) {
      if (tid == 0) {
        __threadfence();
        atomicAdd(cnt, 1);
        const int want = (n_bar + 1) * W;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE,
                                 __HIP_MEMORY_SCOPE_AGENT) < want)
          __builtin_amdgcn_s_sleep(2);
        __threadfence();
      }
      __syncthreads();
    }
    ++n_bar;
  };

  // weighted second moments of this workgroup's rows: the sub-groups' partial
  // tiles (every wave of a sub-group owns two tile rows), summed over the
  // sub-groups into ONE partial set per workgroup, part_wg[buf][wg]
  auto moments_wg = [&](const volatile double* w, int buf) {
    const int sg = wave / GW, wv = wave - sg * GW;
    if (sg < SG)
      // (a shorter ring than the stand-alone moment kernel's: this kernel
      // keeps more alive around it)
      sy_moments<DT, (DT <= 2 ? 8 : (DT <= 4 ? 4 : 2))>(
          x, w, d, row0, row1, sg, SG, wv, lane,
          part_sg + (size_t)sg * NT * NB_TILE);
    __threadfence_block();
    __syncthreads();
    volatile double* dst = part_wg + ((size_t)buf * GM_MAXW + wg) * NT * NB_TILE;
    const volatile double* src = part_sg;
    for (int e = tid; e < NT * NB_TILE; e += GM_THREADS) {
      double v[SG];
#pragma unroll
      for (int s = 0; s < SG; ++s) v[s] = src[(size_t)s * NT * NB_TILE + e];
      double acc = v[0];
#pragma unroll
      for (int s = 1; s < SG; ++s) acc += v[s];
      dst[e] = acc;
    }
  };
  // entry (r, c), r >= c, of the moment matrix of ALL rows: the workgroups'
  // partial sums in workgroup order (loads first, then the sum)
  auto moment_all = [&](int buf, int r, int c) {
    const int it = r >> 4, jt = c >> 4, i = r & 15, j = c & 15;
    const int off = mv_tri(it, jt) * NB_TILE + (i >> 2) * 64 + (i & 3) * 16 + j;
    const volatile double* src =
        part_wg + (size_t)buf * GM_MAXW * NT * NB_TILE + off;
    double v[GM_MAXW];
#pragma unroll
    for (int u = 0; u < GM_MAXW; ++u)
      v[u] = src[(size_t)(u < W ? u : 0) * NT * NB_TILE];
    double s = v[0];
#pragma unroll
    for (int u = 1; u < GM_MAXW; ++u) s += (u < W) ? v[u] : 0.0;
    return s;
  };

  // S_all = sum q q^T over all rows: a private copy per workgroup (lower
  // triangle + mirror)
  moments_wg(nullptr, 0);
  restart_barrier();
  for (int e = tid; e < m * m; e += GM_THREADS) {
    const int r = e / m, c = e - r * m;
    sall[e] = moment_all(0, r > c ? r : c, r > c ? c : r);
  }
  __threadfence_block();
  __syncthreads();
  GM_STAMP(0);

  // ---- seeding ---------------------------------------------------------
  // mean feature variance (tolerance scale of k-means, cluster/_kmeans.py:
  // _tolerance)
  double mean_var;
  {
    double term = 0.0;
    if (tid < d) {
      const double mu = sall[d * m + tid] / n;
      term = sall[tid * m + tid] / n - mu * mu;
    }
    mean_var = block_sum(term, red, wave, lane) / d;
    __syncthreads();
  }
  // partial results of a Lloyd iteration per workgroup, two buffers in turn:
  // [2][DP] centre sums, points of cluster 1, changed labels; slot 2 of the
  // first buffer's tail: the seeding's verdict
  const long long kst = 2 * DP + 8;
  volatile double* kaux = scr + L.kaux;              // [2][GM_MAXW][kst]

  // ---- initial hard assignment ---------------------------------------------
  if (a.init_labels != nullptr) {
    for (int i = row0 + tid; i < row1; i += GM_THREADS)
      r0[i] = a.init_labels[(long long)init * n + i] == 0 ? 1.0 : 0.0;
  } else {
    // k-means++ (two centres) on workgroup 0, all rows; the centres go to
    // the others through global memory
    if (wg == 0) {
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
    for (int f = tid; f < DP; f += GM_THREADS) {
      kaux[f] = f < d ? cen[f] : 0.0;
      kaux[DP + f] = f < d ? cen[DP + f] : 0.0;
    }
    }
    restart_barrier();
    for (int f = tid; f < 2 * DP; f += GM_THREADS) cen[f] = kaux[f];
    for (int i = row0 + tid; i < row1; i += GM_THREADS) lab[i] = -1;
    __threadfence_block();
    restart_barrier();      // (kaux buffer 0 is reused by the first iteration)

    // Lloyd iterations (cluster/_kmeans.py:_kmeans_single_lloyd), the rows
    // dealt out over the workgroups: labels and partial centre sums of the
    // own rows, one barrier, then every workgroup adds the partial sums in
    // the same order and takes the same decisions
    for (int it = 0; it < 300; ++it) {
      volatile double* mine = kaux + ((size_t)(it & 1) * GM_MAXW + wg) * kst;
      double changed = 0.0;
      for (int i = row0 + tid; i < row1; i += GM_THREADS) {
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
      // centre sums of the own rows: (feature, chunk of points)
      // decomposition, the features in passes of 64
      {
        const int ch = wave;
        const int per = (row1 - row0 + GM_WAVES - 1) / GM_WAVES;
        const int i0 = row0 + ch * per;
        const int i1 = (i0 + per < row1) ? i0 + per : row1;
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
      if (tid < 2 * DP) {
        double sum = 0.0;
        for (int w = 0; w < GM_WAVES; ++w) sum += cpart[w * 2 * DP + tid];
        mine[tid] = sum;
      }
      if (tid == 0) {
        double n1w = 0.0;
        for (int w = 0; w < GM_WAVES; ++w) n1w += red[w];
        mine[2 * DP] = n1w;
        mine[2 * DP + 1] = changed;
      }
      restart_barrier();
      const volatile double* all = kaux + (size_t)(it & 1) * GM_MAXW * kst;
      double n1 = 0.0, changed_all = 0.0;
      for (int u = 0; u < W; ++u) {
        n1 += all[u * kst + 2 * DP];
        changed_all += all[u * kst + 2 * DP + 1];
      }
      const double n0 = n - n1;
      if (n0 < 1.0 || n1 < 1.0) {            // an empty cluster: give up
        if (tid == 0) sh_bad = 1;
        __syncthreads();
        break;
      }
      double shift = 0.0;
      if (tid < 2 * DP) {
        const int k = tid / DP, f = tid - k * DP;
        double v[GM_MAXW];
#pragma unroll
        for (int u = 0; u < GM_MAXW; ++u)
          v[u] = all[(u < W ? u : 0) * kst + tid];
        double sum = v[0];
#pragma unroll
        for (int u = 1; u < GM_MAXW; ++u) sum += (u < W) ? v[u] : 0.0;
        const double c_new = (f < d) ? sum / (k ? n1 : n0) : 0.0;
        const double dlt = c_new - ((f < d) ? cen[k * DP + f] : 0.0);
        shift = dlt * dlt;
        cen[k * DP + f] = c_new;   // only this thread touches cen[k][f] here
      }
      shift = block_sum(shift, red, wave, lane);
      __syncthreads();
      if (changed_all == 0.0 || shift <= 1e-4 * mean_var) break;
    }
    __syncthreads();
    for (int i = row0 + tid; i < row1; i += GM_THREADS)
      r0[i] = lab[i] == 0 ? 1.0 : 0.0;
  }
  __threadfence_block();
  __syncthreads();
  GM_STAMP(1);

  // the entries (r >= c) of the lower triangle, packed (position in the
  // operand tiles << 14 | r << 7 | c), in LDS:
  // thread t owns the entries t, t + 512, ... of every sweep (a table in
  // registers -- 17 slots of three ints for n_dim 128 -- was live across the
  // whole EM loop and evaluated in full at every pivot whatever n_dim)
  const int n_low = d * (d + 1) / 2;
  for (int e = tid; e < n_low; e += GM_THREADS) {
    int r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > e) --r;
    while ((r + 1) * (r + 2) / 2 <= e) ++r;
    const int c = e - r * (r + 1) / 2;
    swt[e] = ((unsigned)sy_pos(r, c) << 14) | (unsigned)(r << 7) | (unsigned)c;
  }
  __syncthreads();

  // ---- EM ------------------------------------------------------------------
  const double eps10 = 10.0 * 2.220446049250313e-16;
  double lower = -inf;
  int n_iter = 0, converged = 0;
  double pi0 = 0.5, pi1 = 0.5, nk0 = 1.0, nk1 = 1.0;
  bool failed = sh_bad != 0;
  const int ks_max = 4 * DT;
  int pb = 1;                 // partial-sum buffer of the current M-step
  // Parameters of the M-step behind the partial sums in buffer pb
  // (mixture/_gaussian_mixture.py:_estimate_gaussian_parameters): weights and
  // means here, the covariances where they are needed -- as operand tiles in
  // LDS (log_prob) and, at the end, in the output record
  auto m_step = [&]() {
    const double s00 = moment_all(pb, d, d);
    nk0 = s00 + eps10;
    nk1 = ((double)n - s00) + eps10;
    pi0 = nk0 / (nk0 + nk1);      // _m_step: weights_ /= weights_.sum()
    pi1 = nk1 / (nk0 + nk1);
    __syncthreads();              // mean_l free
    for (int f = tid; f < DP; f += GM_THREADS) {
      double m0 = 0.0, m1 = 0.0;
      if (f < d) {
        const double s0 = moment_all(pb, d, f);
        m0 = s0 / nk0;
        m1 = (sall[d * m + f] - s0) / nk1;
      }
      mean_l[f] = m0;
      mean_l[DP + f] = m1;
    }
    __syncthreads();
  };
  auto cov_entry = [&](int r, int c, double& c0, double& c1) {
    const double s0 = moment_all(pb, r, c);
    const double s1 = sall[r * m + c] - s0;
    const double reg = (r == c) ? a.reg : 0.0;
    c0 = s0 / nk0 - mean_l[r] * mean_l[c] + reg;
    c1 = s1 / nk1 - mean_l[DP + r] * mean_l[DP + c] + reg;
  };
  // weighted log probabilities of both components for this workgroup's rows
  // with the current parameters (mixture/_base.py:
  // _estimate_weighted_log_prob).  Both components go through the pivots
  // TOGETHER where two sets of operand tiles fit the LDS (BOTH: n_dim <=
  // 111): one pair of barriers per pivot for both sweeps, and the quadratic
  // forms of both read every point once.
  constexpr int NK = BOTH ? 2 : 1;
  auto log_prob = [&]() {
    for (int k0 = 0; k0 < 2; k0 += NK) {
      for (int e = tid; e < NK * NT * NB_TILE; e += GM_THREADS) T[e] = 0.0;
      __syncthreads();
      for (int e = tid; e < n_low; e += GM_THREADS) {
        const unsigned w = swt[e];
        double c0, c1;
        cov_entry((w >> 7) & 127, w & 127, c0, c1);
        if constexpr (BOTH) {
          T[w >> 14] = c0;
          T[NT * NB_TILE + (w >> 14)] = c1;
        } else {
          T[w >> 14] = k0 == 0 ? c0 : c1;
        }
      }
      for (int f = tid; f < NK * DP; f += GM_THREADS) {
        const int kk = f >= DP ? 1 : 0, ff = f - kk * DP;
        mus[kk * DP + mv_slot(ff)] = mean_l[(k0 + kk) * DP + ff];
      }
      __syncthreads();
      GM_STAMP(3);
      // symmetric sweep operator over all pivots: T <- -Sigma^-1, the pivots
      // are those of the L D L^T factorisation (log det = sum log d_p).
      // Every thread keeps ITS entries of the triangle (e = tid, tid + 512,
      // ...) in registers through all pivots; per pivot the owners of column
      // p publish it (LDS, two buffers in turn), ONE barrier, and every entry
      // is updated from two LDS reads -- the matrices themselves are read
      // and written once per sweep.  (With the matrices in LDS every pivot
      // cost 2.2 us: ~34 dependent LDS operations per thread and two
      // barriers; 62 % of an EM iteration.)
      int er[EPT], ec[EPT], epos[EPT];
      double tv[NK][EPT];
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        const int e = tid + u * GM_THREADS;
        const unsigned w = swt[e < n_low ? e : 0];
        er[u] = e < n_low ? (int)((w >> 7) & 127) : 255;
        ec[u] = (int)(w & 127);
        epos[u] = (int)(w >> 14);
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
          tv[kk][u] = T[kk * NT * NB_TILE + epos[u]];
      }
      for (int p = 0; p < d; ++p) {
        double* ck = colk + (p & 1) * NK * DP;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          if (er[u] == 255) continue;
          const int at = ec[u] == p ? er[u] : (er[u] == p ? ec[u] : -1);
          if (at >= 0) {
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) ck[kk * DP + at] = tv[kk][u];
          }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
          const double* c = ck + kk * DP;
          const double dp = c[p];
          if (!(dp > 0.0)) {                   // not positive definite
            if (tid == 0) sh_bad = 1;
          }
          if (tid == 0) piv[kk * DP + p] = dp;
          const double inv_d = 1.0 / dp;
          double cr[EPT], cc[EPT];
#pragma unroll
          for (int u = 0; u < EPT; ++u) {
            cr[u] = c[er[u] & 127];
            cc[u] = c[ec[u]];
          }
#pragma unroll
          for (int u = 0; u < EPT; ++u) {
            const int r = er[u], cidx = ec[u];
            double v;
            if (r == p && cidx == p) v = -inv_d;
            else if (r == p) v = cc[u] * inv_d;
            else if (cidx == p) v = cr[u] * inv_d;
            else v = tv[kk][u] - cr[u] * cc[u] * inv_d;
            tv[kk][u] = v;
          }
        }
      }
      // -> T form: -(...) and doubled off-diagonal entries, back to LDS
#pragma unroll
      for (int u = 0; u < EPT; ++u) {
        if (er[u] == 255) continue;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk)
          T[kk * NT * NB_TILE + epos[u]] =
              tv[kk][u] * (er[u] == ec[u] ? -1.0 : -2.0);
      }
      __syncthreads();
      // log det = sum of the logs of the pivots, in pivot order (every
      // thread the same sum: the logs in parallel, then added in order)
      for (int i = tid; i < NK * DP; i += GM_THREADS)
        if (i % DP < d) piv[i] = log(piv[i]);
      __syncthreads();
      double logdet[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk) {
        double acc = 0.0;
        for (int pp = 0; pp < d; ++pp) acc += piv[kk * DP + pp];
        logdet[kk] = acc;
      }
      GM_STAMP(4);
      double konst[NK];
#pragma unroll
      for (int kk = 0; kk < NK; ++kk)
        konst[kk] = log(k0 + kk == 0 ? pi0 : pi1) -
                    0.5 * (d * 1.8378770664093453 + logdet[kk]);
      // (the points of a wavefront's next tile are loaded in front of the
      // products of the current one where the registers allow it)
      constexpr bool PRE = DT <= 5;
      const int tile0 = row0 / 16;
      double xnext[1][4 * DT];
      if constexpr (PRE) {
        long long pn[1] = {(long long)(tile0 + wave) * 16 + lj};
        bool vn[1] = {pn[0] < row1};
        load_points<DT, 1>((const nb_gd*)x, pn, vn, d, (long long)n, lane,
                           xnext);
      }
      for (int tile = tile0 + wave; tile * 16 < row1; tile += GM_WAVES) {
        long long pt[1] = {(long long)tile * 16 + lj};
        bool valid[1] = {pt[0] < row1};
        double xin[1][4 * DT];
        if constexpr (PRE) {
#pragma unroll
          for (int ks = 0; ks < 4 * DT; ++ks) xin[0][ks] = xnext[0][ks];
          long long pn[1] = {(long long)(tile + GM_WAVES) * 16 + lj};
          bool vn[1] = {pn[0] < row1};
          load_points<DT, 1>((const nb_gd*)x, pn, vn, d, (long long)n, lane,
                             xnext);
        } else {
          load_points<DT, 1>((const nb_gd*)x, pt, valid, d, (long long)n, lane,
                             xin);
        }
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
  double lse_wg = 0.0;        // this workgroup's share of the last E-step's sum
  bool have_params = false;
  for (int it = 0; it <= a.max_iter && !failed; ++it) {
    // this workgroup's moments under the current responsibilities and its
    // share of the log-likelihood sum of the E-step before, then the barrier
    moments_wg(r0, pb);
    if (tid == 0) aux[((size_t)pb * GM_MAXW + wg) * 8] = lse_wg;
    restart_barrier();
    GM_STAMP(2);
    if (it > 0) {
      double lsum = 0.0;
      for (int u = 0; u < W; ++u) lsum += aux[((size_t)pb * GM_MAXW + u) * 8];
      const double lb = lsum / n;
      n_iter = it;
      if (fabs(lb - lower) < a.tol) converged = 1;
      lower = lb;
    }
    m_step();
    have_params = true;
    if (it == a.max_iter || converged) break;

    // E-step (mixture/_base.py:_estimate_log_prob_resp)
    log_prob();
    failed = sh_bad != 0;
    double lse_part = 0.0;
    for (int i = row0 + tid; i < row1; i += GM_THREADS) {
      const double l0 = lp0[i], l1 = lp1[i];
      const double mx = l0 > l1 ? l0 : l1;
      const double lse = mx + log(exp(l0 - mx) + exp(l1 - mx));
      r0[i] = exp(l0 - lse);
      lse_part += lse;
    }
    __threadfence_block();
    lse_wg = block_sum(lse_part, red, wave, lane);
    __syncthreads();
    GM_STAMP(6);
    pb ^= 1;
  }
  // the log probabilities under the FINAL parameters (those of the last
  // M-step) stay in the scratch of this restart: Union.split assigns every
  // point to its more probable component (bounds/union.py:188-197)
  if (!failed) {
    log_prob();
    failed = sh_bad != 0;
  }
#ifdef NB_GMM_TIMING
  if (tid == 0 && init == 0 && wg == 0)
    printf("[gmm] n=%d d=%d W=%d iters=%d  ticks(100MHz): s_all %lld "
           "seed+lloyd %lld moments+barrier %lld build %lld sweeps %lld "
           "quadform %lld estep %lld\n",
           n, d, W, n_iter, tk[0], tk[1], tk[2], tk[3], tk[4], tk[5], tk[6]);
#endif
  if (wg == 0) {
    // the record of the restart: final parameters of the last M-step
    if (have_params) {
      for (int f = tid; f < d; f += GM_THREADS) {
        o_mean[f] = mean_l[f];
        o_mean[d + f] = mean_l[DP + f];
      }
      for (int e = tid; e < n_low; e += GM_THREADS) {
        const unsigned w = swt[e];
        const int r = (w >> 7) & 127, c = w & 127;
        double c0, c1;
        cov_entry(r, c, c0, c1);
        o_cov[r * d + c] = c0;
        o_cov[c * d + r] = c0;
        o_cov[d * d + r * d + c] = c1;
        o_cov[d * d + c * d + r] = c1;
      }
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
}

inline int gm_dt(int d) { return (d + 1 + 15) / 16; }
inline size_t gm_lds_doubles(int dt) {
  const size_t nk = gm_both(dt) ? 2 : 1;
  const size_t tiles = nk * dt * (dt + 1) / 2 * NB_TILE;
  const size_t seed = (size_t)GM_WAVES * 2 * 16 * dt;      // cpart aliases T
  // ... pivot columns, means (slot and feature order), pivots, k-means
  // centres, table of the triangle's entries
  return (tiles > seed ? tiles : seed) + (4 * nk + 4) * 16 * dt +
         ((size_t)16 * dt * (16 * dt + 1) / 2 + 1) / 2 + 2;
}
// workgroups per restart: a tile row of 256 points or more each, all
// workgroups of the launch resident at once (they wait for each other)
static std::atomic<int> g_gmm_cap{0};   // 0 = not set through the C ABI

inline int gm_wgs(long long n, int n_init) {
  // The workgroups of a restart wait for each other inside an ordinary
  // launch, so all of a launch must be resident at once.  Three limits:
  //  * NB_GMM_MAX_WGS / nb_gmm_set_max_wgs(): processes that SHARE a GPU
  //    (several ranks on one device) and fit at the same time cannot count
  //    on residency -- they set the cap to 1 (parallel.py does when it finds
  //    two ranks on one device);
  //  * half of the device's CUs for the whole launch (one workgroup per CU
  //    at this kernel's LDS size; a partitioned device has 32-128 CUs, not
  //    256) -- the other half stays free for whatever else is in flight;
  //  * GM_MAXW.
  static const int env_cap = []() {
    const char* e = getenv("NB_GMM_MAX_WGS");
    const int v = e != nullptr ? atoi(e) : GM_MAXW;
    return v < 1 ? 1 : (v > GM_MAXW ? GM_MAXW : v);
  }();
  static const int half_cus = []() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount,
                              dev) != hipSuccess || cus <= 0)
      cus = 64;
    return cus / 2;
  }();
  const int set_cap = g_gmm_cap.load(std::memory_order_relaxed);
  const int cap = set_cap > 0 && set_cap < env_cap ? set_cap : env_cap;
  long long w = n / 256;
  if (w > cap) w = cap;
  if (w * n_init > half_cus) w = half_cus / n_init;
  return (int)(w < 1 ? 1 : w);
}

template <int DT>
int launch_gmm(const GmmArgs& a, int wgs, hipStream_t stream) {
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
  hipLaunchKernelGGL(nb_gmm_kernel<DT>, dim3(wgs, a.n_init), dim3(GM_THREADS),
                     lds, stream, a);
  return NB_OK;
}

}  // namespace

long long nb_gmm_out_stride_impl(int d) { return 6 + 2LL * d + 2LL * d * d; }
long long nb_gmm_scratch_stride_impl(long long n, int d) {
  return gm_layout(n, d).total;
}
// [n_init restarts][counter blocks of the restarts]
long long nb_gmm_work_doubles_impl(long long n, int d, int n_init) {
  return (long long)n_init * (nb_gmm_scratch_stride_impl(n, d) +
                              GM_SYNC_INTS / 2) + 16;
}
long long nb_gmm_logp_offset_impl(long long n, int d) {
  return gm_layout(n, d).lp0;
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
  a.sync = (int*)(scratch + (long long)n_init * a.scratch_stride);
  NB_HIP_CHECK(hipMemsetAsync(a.sync, 0,
                              (size_t)n_init * GM_SYNC_INTS * sizeof(int),
                              stream));
  const int wgs = gm_wgs(n, n_init);
  int rc = NB_OK;
  switch (gm_dt(d)) {
    case 1: rc = launch_gmm<1>(a, wgs, stream); break;
    case 2: rc = launch_gmm<2>(a, wgs, stream); break;
    case 3: rc = launch_gmm<3>(a, wgs, stream); break;
    case 4: rc = launch_gmm<4>(a, wgs, stream); break;
    case 5: rc = launch_gmm<5>(a, wgs, stream); break;
    case 6: rc = launch_gmm<6>(a, wgs, stream); break;
    case 7: rc = launch_gmm<7>(a, wgs, stream); break;
    case 8: rc = launch_gmm<8>(a, wgs, stream); break;
    default: rc = launch_gmm<9>(a, wgs, stream); break;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_gmm_set_cap_impl(int max_wgs) {
  g_gmm_cap.store(max_wgs < 0 ? 0 : max_wgs, std::memory_order_relaxed);
  return NB_OK;
}
