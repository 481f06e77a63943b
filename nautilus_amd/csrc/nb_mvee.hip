// Minimum-volume enclosing ellipsoid: the batched Khachiyan iteration of
// minimum_volume_enclosing_ellipsoid (reference nautilus/bounds/basic.py:
// 175-232) for a BATCH of independent point sets, n_dim <= 128, spread over
// many workgroups.
//
//   q_i = (x_i, 1) (m = D + 1);  V = sum_i u_i q_i q_i^T;  g_i = q_i^T V^-1 q_i
//   per sweep (n_max = 100): the n_batch = 20 largest g_i, in descending
//   order, each update  a = (g - m) / (m (g - 1)),
//   V <- (1-a) V + a q_j q_j^T,  u <- (1-a) u + a e_j          (basic.py:217-231)
//
// What the reference does with one LAPACK inversion per update is done here
// with P = V^-1 carried along (the updates only ever ADD positive rank-one
// terms, the benign direction of Sherman-Morrison), on standardised points
// (the iteration is affine invariant; centring removes the cancellation the
// homogeneous coordinate otherwise causes):
//
//   nb_moments_kernel   S = sum_i w_i q_i q_i^T on the matrix cores, partial
//                       sums per workgroup (fixed reduction order)
//   nb_spd_inverse_kernel  P_0 = (S / n)^-1, in-place Gauss-Jordan in LDS
//   nb_mvee_sweep_kernel   launched n_max + 1 times; grid = (workgroups per
//                       problem, problems).  Call k, every workgroup:
//     A. reads the candidates of call k-1 (the n_batch largest g of every
//        workgroup), merges them into the global top n_batch, and replays the
//        sequential rank-one updates in the n_batch-dimensional space spanned
//        by the selected points: with W = P Q_sel^T and G = Q_sel P Q_sel^T
//        every update is a rank-one update of the (n_batch x n_batch) Gram
//        matrix -- one wavefront, registers and readlane only, no barriers.
//        The result is P_new = s (P - sum_k kappa_k (W z_k)(W z_k)^T), a
//        rank-n_batch product on the matrix cores.  All workgroups do this
//        redundantly (bit-identical), workgroup 0 publishes P_new and u.
//     B. g_i = q_i^T P_new q_i for its own points on the matrix cores
//        (y = T q with T = lower triangle of P_new, off-diagonal doubled, so
//        that g = q . y needs the lower-triangular tiles only), then its own
//        n_batch largest -> candidates of call k+1.
//   The kernel boundary is the grid synchronisation (1.5-2 us, cheaper than
//   any in-kernel grid barrier on this part).
//   nb_mvee_finish_kernel  u <- scale * u (the (1-a) factors are applied
//                       lazily)
//
// Centre, covariance and the final scaling (basic.py:233-241) follow from u
// with one more nb_moments launch and one "B only" call (nb_quadform_max).
#include "nb_sym.h"

#include <cstring>

namespace {

constexpr int MV_THREADS = 512;
constexpr int MV_WAVES = MV_THREADS / 64;
constexpr int MV_MAXB = 16;        // problems per launch
constexpr int MV_MAXW = 32;        // workgroups (candidate lists) per problem
constexpr int MV_MAXPPW = 8192;    // points per workgroup (their g values live in LDS:
                                   // launch_sweep refuses what does not fit at the n_dim)
constexpr int MV_GJ_EPT = 33;      // ceil(129^2 / 512)

struct MvProb {
  const double* xs;      // standardised points [n][d]
  double* u;             // weights [n] (lazy scale until the finish kernel)
  double* P;             // 2 x (m*m) row-major, ping-pong
  double* cand_g;        // 2 x W x NSC
  int* cand_i;           // 2 x W x (NSC+1), last = entries of the list
  double* state;         // [0] lazy scale of u  [1] accepted updates
  int n, W, ppw, pad;
};
struct MvBatch { MvProb p[MV_MAXB]; };

struct MomProb {
  const double* x;       // [n][d]
  const double* w;       // [n] or null (unit weights)
  double* partial;       // [VW][NT][256]
  double* out;           // [m*m] (reduce kernel) or null
  int n, pad;
};
struct MomBatch { MomProb p[MV_MAXB]; };

typedef double mv_c2 __attribute__((ext_vector_type(2)));   // (g, index bits)

__device__ __forceinline__ mv_c2 mv_cand(double g, int i) {
  return mv_c2{g, __longlong_as_double((long long)i)};
}
__device__ __forceinline__ int mv_idx(mv_c2 c) {
  return (int)__double_as_longlong(c.y);
}
__device__ __forceinline__ bool mv_before(double yg, int yi, double xg, int xi) {
  return yg > xg || (yg == xg && yi > xi);
}

// Number of entries of NL lists (sorted in "before" order, list b at
// base + b * stride with len[b] <= 32 entries) that rank before x.  The
// bisections of all lists advance in lock step: per round NL independent
// 16-byte LDS reads, one wait.  (Written as nested ifs the compiler emits one
// branch + wait per probe, ~100 dependent LDS round trips per candidate.)
template <int NL>
__device__ __forceinline__ int mv_count_lists(const mv_c2* base, int stride,
                                              const int (&len)[NL], double xg,
                                              int xi) {
  int pos[NL];
#pragma unroll
  for (int b = 0; b < NL; ++b) pos[b] = 0;
#pragma unroll
  for (int step = 32; step > 0; step >>= 1) {
    mv_c2 v[NL];
#pragma unroll
    for (int b = 0; b < NL; ++b) {
      const int cand = pos[b] + step;
      const int at = (cand <= len[b] ? cand : 1) - 1;
      v[b] = base[b * stride + at];
    }
    // all loads issued; pin them (otherwise they sink into branches again)
#pragma unroll
    for (int b = 0; b < NL; ++b)
      asm volatile("" : "+v"(v[b].x), "+v"(v[b].y));
#pragma unroll
    for (int b = 0; b < NL; ++b) {
      const int cand = pos[b] + step;
      const bool ok = (cand <= len[b]) &
                      mv_before(v[b].x, mv_idx(v[b]), xg, xi);
      pos[b] = ok ? cand : pos[b];
    }
  }
  int total = 0;
#pragma unroll
  for (int b = 0; b < NL; ++b) total += pos[b];
  return total;
}

__device__ __forceinline__ double mv_readlane(double v, int l) {
  const unsigned long long b = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// LDS layout (doubles) shared by host and device
struct MvLds {
  int T, Wt, Qs, G0, Zs, misc, gwg, L1c, L1n, total;
};
__host__ __device__ inline MvLds mv_layout(int dt, int nsc, int ppw_max) {
  MvLds l;
  const int ldw = 16 * dt + 1;
  int off = 0;
  l.T = off; off += dt * (dt + 1) / 2 * NB_TILE;
  int wt = nsc * ldw;
  const int stage = 2 * MV_MAXW * nsc + MV_MAXW / 2 + 2;
  if (wt < stage) wt = stage;            // candidate staging aliases Wt
  l.Wt = off; off += (wt + 1) & ~1;
  l.Qs = off; off += (nsc * ldw + 1) & ~1;
  l.G0 = off; off += 2 * nsc * nsc;      // two K halves
  l.Zs = off; off += nsc * nsc;
  l.misc = off; off += 4 * nsc + 16;     // sel_g, kap, sel_i (ints), scalars
  off = (off + 3) & ~3;
  l.gwg = off; off += (ppw_max + 3) & ~3;
  l.L1c = off; off += 2 * MV_WAVES * nsc;
  l.L1n = off; off += MV_WAVES / 2 + 2;
  l.total = off;
  return l;
}

// second level of a selection: MV_WAVES sorted lists -> the k best overall,
// written through `emit(rank, g, i)`
template <typename Emit>
__device__ __forceinline__ void mv_select_l2(const mv_c2* L1c, const int* L1n,
                                             int nsc, int k, int tid,
                                             Emit emit) {
  if (tid < MV_WAVES * nsc) {
    const int a = tid / nsc, p = tid - a * nsc;
    if (p < L1n[a]) {
      const mv_c2 x = L1c[a * nsc + p];
      int len[MV_WAVES];
#pragma unroll
      for (int b = 0; b < MV_WAVES; ++b) len[b] = b != a ? L1n[b] : 0;
      const int rank =
          p + mv_count_lists<MV_WAVES>(L1c, nsc, len, x.x, mv_idx(x));
      if (rank < k) emit(rank, x.x, mv_idx(x));
    }
  }
}

// ---------------------------------------------------------------------------
// the sweep kernel
//   mode 0: Khachiyan call `call` of n_calls
//   mode 1: phase B only with the matrix in P[0] and k = 1 (largest g)
// ---------------------------------------------------------------------------
#ifdef NB_MVEE_TIMING
#define MV_STAMP(k)                                                           \
  do {                                                                        \
    if (wg == 0 && tid == 0 && call == 5)                                     \
      pb.state[8 + (k)] = (double)__builtin_amdgcn_s_memtime();               \
  } while (0)
#else
#define MV_STAMP(k)
#endif

template <int DT, int NSC>
__global__ void __launch_bounds__(MV_THREADS)
nb_mvee_sweep_kernel(MvBatch batch, int d, int n_batch, int call, int n_calls,
                     int mode, int ppw_max) {
  constexpr int DP = 16 * DT;
  constexpr int LDW = DP + 1;
  constexpr int NT = DT * (DT + 1) / 2;
  constexpr int TPWV = (NT + MV_WAVES - 1) / MV_WAVES;   // T tiles per wave
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const MvProb pb = batch.p[blockIdx.y];
  const int wg = blockIdx.x;
  if (wg >= pb.W) return;
  const bool last = (mode == 0 && call > 0 && call == n_calls - 1);
  if (last && wg != 0) return;

  const MvLds L = mv_layout(DT, NSC, ppw_max);
  double* T = lds + L.T;
  double* Wt = lds + L.Wt;
  double* Qs = lds + L.Qs;           // later Yt
  double* G0 = lds + L.G0;           // two partial sums (K halves)
  double* Zs = lds + L.Zs;
  double* sel_g = lds + L.misc;
  double* kap = sel_g + NSC;
  int* sel_i = (int*)(kap + NSC);
  double* scal = kap + NSC + NSC;    // [0] s_fin [1] n_acc
  double* gwg = lds + L.gwg;
  mv_c2* L1c = (mv_c2*)(lds + L.L1c);
  int* L1n = (int*)(lds + L.L1n);
  // candidate staging (merge) aliases Wt
  mv_c2* cc = (mv_c2*)Wt;
  int* cn = (int*)(Wt + 2 * MV_MAXW * NSC);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lj = lane & 15;
  const int n = pb.n, m = d + 1, W = pb.W;
  const double inf = __builtin_huge_val();
  const int K = mode == 1 ? 1 : (n_batch < n ? n_batch : n);
  const int cur = call & 1, prev = cur ^ 1;
  const double* P_old = pb.P + (size_t)(mode == 1 ? 0 : cur) * m * m;
  double* P_new = pb.P + (size_t)(cur ^ 1) * m * m;
  const bool phase_a = (mode == 0 && call > 0);

  int n_acc = 0;
  double s_fin = 1.0;

  if (mode == 0 && call == 0) {
    // u = 1/n (basic.py:216), the workgroup's own slice
    const int base = wg * pb.ppw;
    for (int i = tid; i < pb.ppw && base + i < n; i += MV_THREADS)
      pb.u[base + i] = 1.0 / (double)n;
    if (wg == 0 && tid == 0) { pb.state[0] = 1.0; pb.state[1] = 0.0; }
  }

  // ---- prefetch (global loads that depend on nothing computed here) --------
  // the P tiles this wave rewrites in A8
  double pold[TPWV][4];
#pragma unroll
  for (int jq = 0; jq < TPWV; ++jq) {
    const int q = wave + MV_WAVES * jq;
    int ht = 0;
    while (mv_tri(ht + 1, 0) <= q) ++ht;
    const int kt = q - mv_tri(ht, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int R = 16 * ht + lg + 4 * r, Cc = 16 * kt + lj;
      pold[jq][r] = (q < NT && !last && R < m && Cc < m)
                        ? P_old[(size_t)R * m + Cc] : 0.0;
    }
  }
  // the rows of P for this wave's first (row tile, point tile) pair of A4
  double apre[4 * DT];
  {
    const int row = 16 * (wave >> 1) + lj;
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks) {
      const int col = 4 * ks + lg;
      apre[ks] = (phase_a && wave < 2 * DT && row < m && col < m)
                     ? P_old[(size_t)row * m + col] : 0.0;
    }
  }

  MV_STAMP(0);
  if (phase_a) {
    // ---- A0: stage the candidate lists of the previous call ----------------
    const double* pg = pb.cand_g + (size_t)prev * W * NSC;
    const int* pi = pb.cand_i + (size_t)prev * W * (NSC + 1);
    for (int e = tid; e < W * NSC; e += MV_THREADS) {
      const int w = e / NSC, q = e - w * NSC;
      const int cnt = pi[w * (NSC + 1) + NSC];
      cc[e] = q < cnt ? mv_cand(pg[e], pi[w * (NSC + 1) + q]) : mv_cand(-inf, -1);
      if (q == 0) cn[w] = cnt;
    }
    for (int e = tid; e < MV_WAVES * NSC; e += MV_THREADS)
      L1c[e] = mv_cand(-inf, -1);
    __syncthreads();
    MV_STAMP(1);
    // ---- A1: first level, wave w merges the lists w, w+8, ... --------------
    {
      int total = 0;
      for (int w = wave; w < W; w += MV_WAVES) total += cn[w];
      const int nl = (W - wave + MV_WAVES - 1) / MV_WAVES;     // lists here
      for (int c = lane; c < nl * NSC; c += 64) {
        const int li = c / NSC, p = c - li * NSC;
        const int w = wave + MV_WAVES * li;
        if (p < cn[w]) {
          const mv_c2 x = cc[w * NSC + p];
          int len[MV_MAXW / MV_WAVES];
#pragma unroll
          for (int bl = 0; bl < MV_MAXW / MV_WAVES; ++bl) {
            const int b = wave + MV_WAVES * bl;
            len[bl] = (b < W && b != w) ? cn[b] : 0;
          }
          const int rank = p + mv_count_lists<MV_MAXW / MV_WAVES>(
                                   cc + wave * NSC, MV_WAVES * NSC, len, x.x,
                                   mv_idx(x));
          if (rank < K) L1c[wave * NSC + rank] = x;
        }
      }
      if (lane == 0) L1n[wave] = total < K ? total : K;
    }
    __syncthreads();
    MV_STAMP(2);
    // ---- A2: second level ---------------------------------------------------
    mv_select_l2(L1c, L1n, NSC, K, tid,
                 [&](int rank, double g, int i) { sel_g[rank] = g; sel_i[rank] = i; });
    __syncthreads();
    int n_sel = 0;
#pragma unroll
    for (int w = 0; w < MV_WAVES; ++w) n_sel += L1n[w];
    if (n_sel > K) n_sel = K;

    MV_STAMP(3);
    // ---- A3: the selected rows q_t = (x_t, 1) ------------------------------
    for (int e = tid; e < n_sel * DP; e += MV_THREADS) {
      const int t = e / DP, c = e - t * DP;
      double v = 0.0;
      if (c < d) v = pb.xs[(size_t)sel_i[t] * d + c];
      else if (c == d) v = 1.0;
      Qs[t * LDW + c] = v;
    }
    __syncthreads();
    MV_STAMP(4);

    // ---- A4: W = P Q_sel^T on the matrix cores -> Wt[t][feature] ----------
    // one (row tile, point tile) pair per wave and round
    for (int pair = wave; pair < 2 * DT; pair += MV_WAVES) {
      const int ht = pair >> 1, pt = pair & 1;
      if (16 * pt >= n_sel) continue;
      double a[4 * DT];
      if (pair == wave) {
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) a[ks] = apre[ks];
      } else {
        const int row = 16 * ht + lj;
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) {
          const int col = 4 * ks + lg;
          a[ks] = (row < m && col < m) ? P_old[(size_t)row * m + col] : 0.0;
        }
      }
      const int t = 16 * pt + lj;
      nb_d4 acc = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        const double b = t < n_sel ? Qs[t * LDW + 4 * ks + lg] : 0.0;
        acc = MFMA(a[ks], b, acc);
      }
      if (t < n_sel) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Wt[t * LDW + 16 * ht + lg + 4 * r] = acc[r];
      }
    }
    __syncthreads();
    MV_STAMP(5);

    // ---- A5: G0 = Q_sel W on the matrix cores, K split over two halves ------
    {
      const int si = wave & 1, ti = (wave >> 1) & 1, half = wave >> 2;
      const int ksn = (m + 3) >> 2;
      const int k0 = half == 0 ? 0 : ksn >> 1;
      const int k1 = half == 0 ? ksn >> 1 : ksn;
      if (16 * si < n_sel && 16 * ti < n_sel) {
        const int srow = 16 * si + lj, trow = 16 * ti + lj;
        nb_d4 acc = nb_d4{0.0, 0.0, 0.0, 0.0};
        for (int ks = k0; ks < k1; ++ks) {
          const int c = 4 * ks + lg;
          const double a = srow < n_sel ? Qs[srow * LDW + c] : 0.0;
          const double b = trow < n_sel ? Wt[trow * LDW + c] : 0.0;
          acc = MFMA(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int sr = 16 * si + lg + 4 * r;
          if (sr < n_sel && trow < n_sel)
            G0[half * NSC * NSC + sr * NSC + trow] = acc[r];
        }
      }
    }
    __syncthreads();
    MV_STAMP(6);

    // ---- A6: the sequential updates in the Gram space (one wavefront) ------
    if (wave == 0) {
      // lane l holds row l of the current Gram matrix Gk (q_l^T P_k q_r) and
      // of H = C G0 (z_t = e_t - H[:, t]); the vectors z of the accepted
      // updates go to Zs
      double Gk[NSC], H[NSC];
#pragma unroll
      for (int r = 0; r < NSC; ++r) {
        Gk[r] = (lane < n_sel && r < n_sel)
                    ? G0[lane * NSC + r] + G0[NSC * NSC + lane * NSC + r] : 0.0;
        H[r] = 0.0;
      }
      double s = 1.0, uinc = 0.0;
      double us = pb.state[0];
      double inv_us = 1.0 / us;
      const double md = (double)m;
      int acc_n = 0;
#pragma unroll
      for (int t = 0; t < NSC; ++t) {
        if (t < n_sel) {
          double g = mv_readlane(Gk[t], t);
          if (t == 0) g = mv_readlane(sel_g[0], 0);   // basic.py:222-223
          if (g >= md) {
            // a = (g - m) / (m (g - 1));  with it  1 - a = g (m-1) / (m (g-1))
            // and coef = ratio / (1 + ratio g) = (g - m) / (g (g - 1))  (ratio
            // = a / (1 - a)): ONE division per update instead of a chain of
            // five (the wavefront executes them in order)
            const double gm1 = g - 1.0;
            const double rr = 1.0 / (g * gm1 * md * (md - 1.0));
            const double a = (g - md) * g * (md - 1.0) * rr;
            const double coef = (g - md) * md * (md - 1.0) * rr;
            const double inv1a = md * md * gm1 * gm1 * rr;
            const double z = (lane == t ? 1.0 : 0.0) - H[t];
            const double gt = Gk[t];              // Gk[l][t]
#pragma unroll
            for (int r = 0; r < NSC; ++r) {
              if (r > t) {
                const double v = mv_readlane(Gk[r], t);      // Gk[t][r]
                Gk[r] = (Gk[r] - coef * gt * v) * inv1a;
                H[r] += coef * z * v;
              }
            }
            // record z (column acc_n of Zs) and kappa = coef * s
            if (lane < NSC) Zs[lane * NSC + acc_n] = z;
            us *= (1.0 - a);
            inv_us *= inv1a;
            if (lane == t) uinc = a * inv_us;     // u[sel_i[t]] += a / scale
            if (lane == 0) kap[acc_n] = coef * s;
            s *= inv1a;
            ++acc_n;
          }
        }
      }
      if (wg == 0 && lane < n_sel && uinc != 0.0) pb.u[sel_i[lane]] += uinc;
      if (lane == 0) {
        scal[0] = s;
        scal[1] = (double)acc_n;
        if (wg == 0) {
          pb.state[0] = us;
          pb.state[1] += (double)acc_n;
        }
      }
    }
    __syncthreads();
    s_fin = scal[0];
    n_acc = (int)scal[1];
    if (last) return;
    MV_STAMP(7);

    // ---- A7: Y = W Z on the matrix cores (Yt[k][feature], aliases Qs) -------
    double* Yt = Qs;
    for (int item = wave; item < 2 * DT; item += MV_WAVES) {
      const int ft = item >> 1, kt = item & 1;
      if (16 * kt >= n_acc) continue;
      nb_d4 acc = nb_d4{0.0, 0.0, 0.0, 0.0};
      for (int kb = 0; 4 * kb < n_sel; ++kb) {
        const int b = 4 * kb + lg;
        const double a = b < n_sel ? Wt[b * LDW + 16 * ft + lj] : 0.0;
        const double bz = (b < n_sel && 16 * kt + lj < n_acc)
                              ? Zs[b * NSC + 16 * kt + lj] : 0.0;
        acc = MFMA(a, bz, acc);
      }
      const int k = 16 * kt + lj;
      if (k < n_acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Yt[k * LDW + 16 * ft + lg + 4 * r] = acc[r];
      }
    }
    __syncthreads();
  }

  MV_STAMP(8);
  // ---- A8: P_new = s (P - sum_k kappa_k y_k y_k^T) -> operand tiles T ------
  {
    const double* Yt = Qs;
#pragma unroll
    for (int jq = 0; jq < TPWV; ++jq) {
      const int q = wave + MV_WAVES * jq;
      if (q >= NT) continue;
      int ht = 0;
      while (mv_tri(ht + 1, 0) <= q) ++ht;
      const int kt = q - mv_tri(ht, 0);
      nb_d4 acc = nb_d4{0.0, 0.0, 0.0, 0.0};
      for (int ka = 0; 4 * ka < n_acc; ++ka) {
        const int k = 4 * ka + lg;
        const double a = k < n_acc ? kap[k] * Yt[k * LDW + 16 * ht + lj] : 0.0;
        const double b = k < n_acc ? Yt[k * LDW + 16 * kt + lj] : 0.0;
        acc = MFMA(a, b, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int R = 16 * ht + lg + 4 * r, Cc = 16 * kt + lj;
        const bool in = R < m && Cc < m;
        const double v = s_fin * (pold[jq][r] - acc[r]);
        if (mode == 0 && wg == 0 && in && Cc <= R) {
          P_new[(size_t)R * m + Cc] = v;
          P_new[(size_t)Cc * m + R] = v;
        }
        double tv = 0.0;
        if (in) tv = Cc < R ? 2.0 * v : (Cc == R ? v : 0.0);
        T[q * NB_TILE + mv_kpos(Cc & 15) + mv_hpos(R & 15)] = tv;
      }
    }
  }
  __syncthreads();
  MV_STAMP(9);

  // ---- B1: g_i = q_i^T P q_i for the workgroup's points ---------------------
  const int base = wg * pb.ppw;
  const int cnt = (n - base) < pb.ppw ? (n - base) : pb.ppw;
  const int ks_one = mv_slot(d) >> 2, lg_one = mv_slot(d) & 3;
  for (int tile = wave; tile * 16 < cnt; tile += MV_WAVES) {
    long long pt[1] = {(long long)base + tile * 16 + lj};
    bool valid[1] = {tile * 16 + lj < cnt};
    double xin[1][4 * DT];
    load_points<DT, 1>((const nb_gd*)pb.xs, pt, valid, d, (long long)n, lane, xin);
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks)
      if (ks == ks_one && lg == lg_one) xin[0][ks] = valid[0] ? 1.0 : 0.0;
    const double g = sy_quadform<DT>(T, xin[0], m, lane);
    if (valid[0] && lg == 0) gwg[tile * 16 + lj] = g;
  }
  for (int e = tid; e < MV_WAVES * NSC; e += MV_THREADS)
    L1c[e] = mv_cand(-inf, -1);
  __syncthreads();
  MV_STAMP(10);

  // (One level for workgroups whose points have a thread each -- thread i
  // ranking g_i among all of the workgroup's values -- was measured: 3.26
  // against 3.18-3.26 ms per fit at n = 2000, n_dim 50; the loop over the
  // values waits for LDS as long as the two levels do.  profiles/r06/
  // mvee_phases_start.txt has the stamps of the two-level form.)
  // ---- B2: first level, wave w ranks its slice of the g values --------------
  {
    const int slice = ((cnt + MV_WAVES - 1) / MV_WAVES + 3) & ~3;
    const int s0 = wave * slice;
    const int s1 = (s0 + slice) < cnt ? (s0 + slice) : cnt;
    for (int i = s0 + lane; i < s1; i += 64) {
      const double xg = gwg[i];
      int rank = 0;
      for (int j = s0; j < s1; j += 4) {          // s0 is a multiple of 4
        const nb_d4 v = *(const nb_d4*)(gwg + j);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rank += (j + q < s1 && mv_before(v[q], j + q, xg, i)) ? 1 : 0;
      }
      if (rank < K) L1c[wave * NSC + rank] = mv_cand(xg, base + i);
    }
    const int have = s1 > s0 ? s1 - s0 : 0;
    if (lane == 0) L1n[wave] = have < K ? have : K;
  }
  __syncthreads();
  MV_STAMP(11);
  // ---- B3: second level -> candidates of the next call ----------------------
  {
    double* og = pb.cand_g + ((size_t)cur * W + wg) * NSC;
    int* oi = pb.cand_i + ((size_t)cur * W + wg) * (NSC + 1);
    mv_select_l2(L1c, L1n, NSC, K, tid,
                 [&](int rank, double g, int i) { og[rank] = g; oi[rank] = i; });
    if (tid == 0) oi[NSC] = cnt < K ? cnt : K;
  }
  MV_STAMP(12);
}

// u <- scale u (mode 0) / out = largest g over the lists (mode 1)
__global__ void __launch_bounds__(256)
nb_mvee_finish_kernel(MvBatch batch, int nsc, int mode, double* out) {
  const MvProb pb = batch.p[blockIdx.y];
  if (mode == 0) {
    const double s = pb.state[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pb.n;
         i += gridDim.x * blockDim.x)
      pb.u[i] *= s;
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    double best = -__builtin_huge_val();
    for (int w = 0; w < pb.W; ++w) {
      const double v = pb.cand_g[(size_t)w * nsc];
      if (v > best) best = v;
    }
    out[blockIdx.y] = best;
  }
}

// ---------------------------------------------------------------------------
// S = sum_i w_i q_i q_i^T, q = (x, 1): lower block triangle of 16x16 tiles in
// the accumulator layout, one partial sum per sub-group of wavefronts.  Wave w
// of a sub-group owns the tile rows w and DT-1-w (balanced triangle).
// ---------------------------------------------------------------------------
template <int DT>
__global__ void __launch_bounds__(MV_THREADS)
nb_moments_kernel(MomBatch batch, int d, int pts_per_wg) {
  constexpr int GW = (DT + 1) / 2;            // waves per sub-group
  constexpr int SG = MV_WAVES / GW;           // sub-groups per workgroup
  constexpr int NT = DT * (DT + 1) / 2;
  const MomProb pb = batch.p[blockIdx.y];
  const int p0 = blockIdx.x * pts_per_wg;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sg = wave / GW, w = wave - sg * GW;
  if (sg >= SG) return;
  const int p1 = (p0 + pts_per_wg) < pb.n ? (p0 + pts_per_wg) : pb.n;
  sy_moments<DT>(pb.x, pb.w, d, p0, p1, sg, SG, w, lane,
                 pb.partial + ((size_t)blockIdx.x * SG + sg) * NT * NB_TILE);
}

// out[r][c] = S[r][c] * scale, full symmetric row-major (m x m)
__global__ void __launch_bounds__(256)
nb_moments_reduce_kernel(MomBatch batch, int d, int vw, int nt, double scale) {
  const MomProb pb = batch.p[blockIdx.y];
  const int m = d + 1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < m * m;
       e += gridDim.x * blockDim.x) {
    const int r = e / m, c = e - r * m;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    pb.out[e] = mom_element(pb.partial, vw, nt, hi, lo) * scale;
  }
}

// P_0 = (S / n)^-1: in-place Gauss-Jordan of a symmetric positive definite
// matrix held in LDS, one workgroup per problem
__global__ void __launch_bounds__(MV_THREADS)
nb_spd_inverse_kernel(MomBatch batch, int d, int vw, int nt) {
  extern __shared__ __attribute__((aligned(16))) double A[];
  const MomProb pb = batch.p[blockIdx.y];
  const int m = d + 1, tid = threadIdx.x;
  const double inv_n = 1.0 / (double)pb.n;
  int er[MV_GJ_EPT], ec[MV_GJ_EPT];
#pragma unroll
  for (int q = 0; q < MV_GJ_EPT; ++q) {
    const int e = tid + q * MV_THREADS;
    er[q] = e < m * m ? e / m : -1;
    ec[q] = e < m * m ? e - er[q] * m : 0;
    if (er[q] >= 0) {
      const int hi = er[q] > ec[q] ? er[q] : ec[q];
      const int lo = er[q] > ec[q] ? ec[q] : er[q];
      A[e] = mom_element(pb.partial, vw, nt, hi, lo) * inv_n;
    }
  }
  __syncthreads();
  for (int k = 0; k < m; ++k) {
    const double p = 1.0 / A[k * m + k];
#pragma unroll
    for (int q = 0; q < MV_GJ_EPT; ++q) {
      const int i = er[q], j = ec[q];
      if (i >= 0 && i != k && j != k)
        A[i * m + j] -= A[i * m + k] * A[k * m + j] * p;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MV_GJ_EPT; ++q) {
      const int i = er[q], j = ec[q];
      if (i < 0) continue;
      if (i == k && j == k) A[k * m + k] = p;
      else if (i == k) A[k * m + j] *= p;
      else if (j == k) A[i * m + k] *= -p;
    }
    __syncthreads();
  }
  // symmetrise on the way out (the two triangles agree to rounding)
#pragma unroll
  for (int q = 0; q < MV_GJ_EPT; ++q) {
    const int i = er[q], j = ec[q];
    if (i < 0) continue;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    pb.out[i * m + j] = A[hi * m + lo];
  }
}

// Whitening factor of standardised points: C = S / n (the d x d block of the
// moments) = L D L^T by elimination on the packed lower triangles of [C | I]
// in LDS (one barrier per pivot), W = D^-1/2 L^-1.  Written twice: row-major
// (for the host's back transformation) and as the operand block of
// nb_transform_kernel (nb_common.h "ell block", centre 0), which maps the
// points to unit covariance.  Near-singular pivots are clamped.
constexpr int MV_WH_EPT = 17;          // ceil(128 * 129 / 2 / 512)

__device__ __forceinline__ int pk(int r, int c) { return r * (r + 1) / 2 + c; }

__global__ void __launch_bounds__(MV_THREADS)
nb_whiten_factor_kernel(MomBatch batch, int d, int vw, int nt, double* w_out,
                        double* ell_out) {
  extern __shared__ __attribute__((aligned(16))) double wl[];
  const MomProb pb = batch.p[0];
  const int tid = threadIdx.x;
  const int nl = d * (d + 1) / 2;
  double* A = wl;
  double* B = wl + nl;
  const double inv_n = 1.0 / (double)pb.n;
  int er[MV_WH_EPT], ec[MV_WH_EPT];
#pragma unroll
  for (int q = 0; q < MV_WH_EPT; ++q) {
    const int e = tid + q * MV_THREADS;
    int r = -1, c = 0;
    if (e < nl) {
      r = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
      while (r * (r + 1) / 2 > e) --r;
      while ((r + 1) * (r + 2) / 2 <= e) ++r;
      c = e - r * (r + 1) / 2;
      A[e] = mom_element(pb.partial, vw, nt, r, c) * inv_n;
      B[e] = r == c ? 1.0 : 0.0;
    }
    er[q] = r;
    ec[q] = c;
  }
  __syncthreads();
  for (int p = 0; p + 1 < d; ++p) {
    double dp = A[pk(p, p)];
    if (!(dp > 1e-15)) dp = 1e-15;
    const double inv = 1.0 / dp;
#pragma unroll
    for (int q = 0; q < MV_WH_EPT; ++q) {
      const int i = er[q], j = ec[q];
      if (i > p) {
        const double f = A[pk(i, p)] * inv;
        if (j > p) A[pk(i, j)] -= f * A[pk(j, p)];
        else B[pk(i, j)] -= f * B[pk(p, j)];
      }
    }
    __syncthreads();
  }
  const int dt = (d + 15) / 16, dp16 = 16 * dt;
  const int blk = nb_ell_block_size(dt);
  for (int e = tid; e < blk; e += MV_THREADS) ell_out[e] = 0.0;
  for (int e = tid; e < d * d; e += MV_THREADS) w_out[e] = 0.0;
  __threadfence_block();
  __syncthreads();
  if (tid == 0) ((long long*)ell_out)[0] = d;
  const double inf = __builtin_huge_val();
  for (int f = tid; f < dp16; f += MV_THREADS) {
    ell_out[2 + f] = -inf;
    ell_out[2 + dp16 + f] = inf;
  }
  double* tiles = ell_out + 2 + 3 * dp16;
#pragma unroll
  for (int q = 0; q < MV_WH_EPT; ++q) {
    const int r = er[q], c = ec[q];
    if (r < 0) continue;
    double dr = A[pk(r, r)];
    if (!(dr > 1e-15)) dr = 1e-15;
    const double v = B[pk(r, c)] / sqrt(dr);
    w_out[r * d + c] = v;
    const int sl = mv_slot(c), ks = sl >> 2, lgk = sl & 3;
    tiles[((ks >> 2) * dt + (r >> 4)) * NB_TILE + (ks & 3) * 64 + lgk * 16 +
          (r & 15)] = v;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct MvPlan {
  int W, ppw;
};

inline bool mv_plan(long long n, MvPlan* pl) {
  long long ppw = 128;
  long long W = (n + ppw - 1) / ppw;
  if (W > MV_MAXW) {
    W = MV_MAXW;
    ppw = ((n + W - 1) / W + 15) / 16 * 16;
    W = (n + ppw - 1) / ppw;
  }
  if (ppw > MV_MAXPPW) return false;
  pl->W = (int)W;
  pl->ppw = (int)ppw;
  return true;
}

inline int mom_pts_per_wg(long long n_max) {
  long long p = 512;
  while ((n_max + p - 1) / p > 64) p *= 2;
  return (int)p;
}
inline int mom_sg(int dt) { return MV_WAVES / ((dt + 1) / 2); }

template <typename F>
int set_lds(F* fn, size_t bytes, size_t* allowed) {
  if (bytes > *allowed) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", bytes,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    *allowed = bytes;
  }
  return NB_OK;
}

template <int DT, int NSC>
int launch_sweep(const MvBatch& b, int nb, int wmax, int d, int n_batch,
                 int call, int n_calls, int mode, int ppw_max,
                 hipStream_t stream) {
  static size_t allowed = 0;
  const size_t bytes = (size_t)mv_layout(DT, NSC, ppw_max).total * sizeof(double);
  if (bytes > 160 * 1024) {
    nb_set_error("device MVEE: %zu bytes of LDS needed (n_batch too large for "
                 "this n_dim)", bytes);
    return NB_ERR_UNSUPPORTED;
  }
  const int rc = set_lds(nb_mvee_sweep_kernel<DT, NSC>, bytes, &allowed);
  if (rc != NB_OK) return rc;
  const bool last = mode == 0 && call > 0 && call == n_calls - 1;
  hipLaunchKernelGGL((nb_mvee_sweep_kernel<DT, NSC>), dim3(last ? 1 : wmax, nb),
                     dim3(MV_THREADS), bytes, stream, b, d, n_batch, call,
                     n_calls, mode, ppw_max);
  return NB_OK;
}

template <int NSC>
int dispatch_sweep(int dt, const MvBatch& b, int nb, int wmax, int d,
                   int n_batch, int call, int n_calls, int mode, int ppw_max,
                   hipStream_t stream) {
  switch (dt) {
#define NB_CASE(DT_)                                                          \
    case DT_:                                                                 \
      return launch_sweep<DT_, NSC>(b, nb, wmax, d, n_batch, call, n_calls,   \
                                    mode, ppw_max, stream);
    NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5)
    NB_CASE(6) NB_CASE(7) NB_CASE(8) NB_CASE(9)
#undef NB_CASE
  }
  nb_set_error("device MVEE supports n_dim <= 128");
  return NB_ERR_UNSUPPORTED;
}

int launch_moments(int dt, const MomBatch& b, int nb, int d, int wg, int ppw,
                   hipStream_t stream) {
  switch (dt) {
#define NB_CASE(DT_)                                                          \
    case DT_:                                                                 \
      hipLaunchKernelGGL(nb_moments_kernel<DT_>, dim3(wg, nb),                \
                         dim3(MV_THREADS), 0, stream, b, d, ppw);             \
      return NB_OK;
    NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5)
    NB_CASE(6) NB_CASE(7) NB_CASE(8) NB_CASE(9)
#undef NB_CASE
  }
  nb_set_error("device moments support n_dim <= 128");
  return NB_ERR_UNSUPPORTED;
}

inline long long mom_partial_doubles(long long n_max, int dt) {
  const int ppw = mom_pts_per_wg(n_max);
  const long long wg = (n_max + ppw - 1) / ppw;
  return wg * mom_sg(dt) * (long long)(dt * (dt + 1) / 2) * NB_TILE;
}

inline int nsc_of(int n_batch) { return n_batch <= 20 ? 20 : 32; }

inline long long mv_prob_doubles(long long n_max, int d, int nsc) {
  const int m = d + 1, dt = (m + 15) / 16;
  long long t = 2LL * m * m + 2;                          // P
  t += 2LL * MV_MAXW * nsc;                               // cand_g
  t += (2LL * MV_MAXW * (nsc + 1) + 1) / 2 + 1;           // cand_i
  t += 32;                                                // state
  t += mom_partial_doubles(n_max, dt);
  return (t + 1) & ~1LL;
}

}  // namespace

long long nb_mvee_work_doubles_impl(int n_problems, long long n_max, int d,
                                    int n_batch) {
  return (long long)n_problems * mv_prob_doubles(n_max, d, nsc_of(n_batch)) + 64;
}

long long nb_moments_work_doubles_impl(long long n, int d) {
  const int dt = (d + 1 + 15) / 16;
  return mom_partial_doubles(n, dt) + 16;
}

// S = sum_i w_i q_i q_i^T (w = null: unit weights), scaled, into out[m*m]
int nb_launch_moments(const double* x, const double* w, long long n, int d,
                      double scale, double* out, double* work,
                      hipStream_t stream) {
  if (d < 1 || d > 128 || n < 1 || n > 2147483647LL / (d + 1)) {
    nb_set_error("device moments: bad shape (n=%lld, n_dim=%d)", n, d);
    return NB_ERR_ARG;
  }
  const int dt = (d + 1 + 15) / 16;
  const int ppw = mom_pts_per_wg(n);
  const int wg = (int)((n + ppw - 1) / ppw);
  MomBatch b;
  memset(&b, 0, sizeof(b));
  b.p[0].x = x; b.p[0].w = w; b.p[0].n = (int)n;
  b.p[0].partial = work; b.p[0].out = out;
  int rc = launch_moments(dt, b, 1, d, wg, ppw, stream);
  if (rc != NB_OK) return rc;
  const int m = d + 1;
  hipLaunchKernelGGL(nb_moments_reduce_kernel, dim3((m * m + 255) / 256, 1),
                     dim3(256), 0, stream, b, d, wg * mom_sg(dt),
                     dt * (dt + 1) / 2, scale);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

// The Khachiyan iteration for n_problems point sets (standardised points).
int nb_launch_mvee_batch(int n_problems, const double* const* xs,
                         const long long* n, int d, int n_max, int n_batch,
                         double* const* u, double* work, hipStream_t stream) {
  if (d < 1 || d > 128) {
    nb_set_error("device MVEE supports n_dim <= 128 (got %d)", d);
    return NB_ERR_UNSUPPORTED;
  }
  if (n_batch < 1 || n_batch > 32 || n_max < 0) {
    nb_set_error("device MVEE: n_batch must be in 1..32");
    return NB_ERR_ARG;
  }
  const int m = d + 1, dt = (m + 15) / 16, nsc = nsc_of(n_batch);
  long long n_hi = 0;
  for (int b = 0; b < n_problems; ++b) {
    if (n[b] <= d || n[b] > 2147483647LL / m) {
      nb_set_error("device MVEE needs n_dim < n (n=%lld, n_dim=%d)", n[b], d);
      return NB_ERR_ARG;
    }
    if (n[b] > n_hi) n_hi = n[b];
  }
  const long long stride = mv_prob_doubles(n_hi, d, nsc);
  const int mom_ppw = mom_pts_per_wg(n_hi);
  const int nt = dt * (dt + 1) / 2;

  for (int b0 = 0; b0 < n_problems; b0 += MV_MAXB) {
    const int nb = (n_problems - b0) < MV_MAXB ? (n_problems - b0) : MV_MAXB;
    MvBatch mb;
    MomBatch qb;
    memset(&mb, 0, sizeof(mb));
    memset(&qb, 0, sizeof(qb));
    int wmax = 0, ppw_max = 0, mom_wg = 0;
    for (int b = 0; b < nb; ++b) {
      MvPlan pl;
      if (!mv_plan(n[b0 + b], &pl)) {
        nb_set_error("device MVEE: too many points (n=%lld)", n[b0 + b]);
        return NB_ERR_UNSUPPORTED;
      }
      double* base = work + (size_t)(b0 + b) * stride;
      MvProb& p = mb.p[b];
      p.xs = xs[b0 + b];
      p.u = u[b0 + b];
      p.n = (int)n[b0 + b];
      p.W = pl.W;
      p.ppw = pl.ppw;
      p.P = base; base += 2LL * m * m + 2;
      p.cand_g = base; base += 2LL * MV_MAXW * nsc;
      p.cand_i = (int*)base; base += (2LL * MV_MAXW * (nsc + 1) + 1) / 2 + 1;
      p.state = base; base += 32;
      MomProb& q = qb.p[b];
      q.x = xs[b0 + b];
      q.w = nullptr;
      q.n = p.n;
      q.partial = base;
      q.out = p.P;                 // P[0] = (S / n)^-1
      if (pl.W > wmax) wmax = pl.W;
      if (pl.ppw > ppw_max) ppw_max = pl.ppw;
      const int wg = (p.n + mom_ppw - 1) / mom_ppw;
      if (wg > mom_wg) mom_wg = wg;
    }
    // every problem uses mom_wg workgroups (idle ones write zeros)
    int rc = launch_moments(dt, qb, nb, d, mom_wg, mom_ppw, stream);
    if (rc != NB_OK) return rc;
    {
      static size_t allowed = 0;
      const size_t bytes = (size_t)m * m * sizeof(double);
      rc = set_lds(nb_spd_inverse_kernel, bytes, &allowed);
      if (rc != NB_OK) return rc;
      hipLaunchKernelGGL(nb_spd_inverse_kernel, dim3(1, nb), dim3(MV_THREADS),
                         bytes, stream, qb, d, mom_wg * mom_sg(dt), nt);
    }
    const int n_calls = n_max + 1;
    for (int call = 0; call < n_calls; ++call) {
      rc = nsc == 20
               ? dispatch_sweep<20>(dt, mb, nb, wmax, d, n_batch, call, n_calls,
                                    0, ppw_max, stream)
               : dispatch_sweep<32>(dt, mb, nb, wmax, d, n_batch, call, n_calls,
                                    0, ppw_max, stream);
      if (rc != NB_OK) return rc;
    }
    hipLaunchKernelGGL(nb_mvee_finish_kernel, dim3(8, nb), dim3(256), 0, stream,
                       mb, nsc, 0, (double*)nullptr);
  }
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

// out[0] = max_i q_i^T P q_i, q = (x, 1), P (m x m) symmetric in p_dev
int nb_launch_quadform_max(const double* x, long long n, int d,
                           const double* p_dev, double* out, double* work,
                           hipStream_t stream) {
  if (d < 1 || d > 128 || n < 1 || n > 2147483647LL / (d + 1)) {
    nb_set_error("quadratic form: bad shape (n=%lld, n_dim=%d)", n, d);
    return NB_ERR_ARG;
  }
  const int m = d + 1, dt = (m + 15) / 16;
  MvPlan pl;
  if (!mv_plan(n, &pl)) {
    nb_set_error("quadratic form: too many points (n=%lld)", n);
    return NB_ERR_UNSUPPORTED;
  }
  MvBatch mb;
  memset(&mb, 0, sizeof(mb));
  MvProb& p = mb.p[0];
  p.xs = x;
  p.n = (int)n;
  p.W = pl.W;
  p.ppw = pl.ppw;
  p.P = const_cast<double*>(p_dev);
  p.cand_g = work;
  p.cand_i = (int*)(work + 2LL * MV_MAXW * 20);
  p.state = work;                       // unused in mode 1
  int rc = dispatch_sweep<20>(dt, mb, 1, pl.W, d, 1, 0, 1, 1, pl.ppw, stream);
  if (rc != NB_OK) return rc;
  hipLaunchKernelGGL(nb_mvee_finish_kernel, dim3(1, 1), dim3(256), 0, stream,
                     mb, 20, 1, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

long long nb_quadform_work_doubles_impl() {
  return 2LL * MV_MAXW * 20 + (2LL * MV_MAXW * 21 + 1) / 2 + 8;
}

long long nb_whiten_work_doubles_impl(long long n, int d) {
  const int dt = (d + 15) / 16;
  return ((n * d + 1) & ~1LL) + mom_partial_doubles(n, (d + 1 + 15) / 16) +
         nb_ell_block_size(dt) + 16;
}

int nb_launch_standardize(const double* x, long long n, int d, double* mean,
                          double* scale, double* out, hipStream_t stream);
int nb_launch_transform(const double* ell_block, int dt, int n_dim,
                        const double* x, long long n, double* y,
                        hipStream_t stream);

// xw = W ((x - mean) / sd): zero mean, unit covariance.  mean[d], sd[d],
// w[d*d] (lower triangular, row-major) describe the map.
int nb_launch_whiten(const double* x, long long n, int d, double* xw,
                     double* mean, double* sd, double* w, double* work,
                     hipStream_t stream) {
  if (d < 1 || d > 128 || n <= d || n > 2147483647LL / (d + 1)) {
    nb_set_error("whitening needs n_dim < n, n_dim <= 128 (n=%lld, n_dim=%d)",
                 n, d);
    return NB_ERR_ARG;
  }
  const int dtm = (d + 1 + 15) / 16, dt = (d + 15) / 16;
  double* xs = work;
  double* partial = xs + ((n * d + 1) & ~1LL);
  double* ell = partial + mom_partial_doubles(n, dtm);
  int rc = nb_launch_standardize(x, n, d, mean, sd, xs, stream);
  if (rc != NB_OK) return rc;
  MomBatch b;
  memset(&b, 0, sizeof(b));
  b.p[0].x = xs; b.p[0].w = nullptr; b.p[0].n = (int)n;
  b.p[0].partial = partial;
  const int ppw = mom_pts_per_wg(n);
  const int wg = (int)((n + ppw - 1) / ppw);
  rc = launch_moments(dtm, b, 1, d, wg, ppw, stream);
  if (rc != NB_OK) return rc;
  {
    static size_t allowed = 0;
    const size_t bytes = (size_t)d * (d + 1) * sizeof(double);
    rc = set_lds(nb_whiten_factor_kernel, bytes, &allowed);
    if (rc != NB_OK) return rc;
    hipLaunchKernelGGL(nb_whiten_factor_kernel, dim3(1), dim3(MV_THREADS), bytes,
                       stream, b, d, wg * mom_sg(dtm), dtm * (dtm + 1) / 2, w,
                       ell);
  }
  NB_HIP_CHECK(hipGetLastError());
  return nb_launch_transform(ell, dt, d, xs, n, xw, stream);
}
