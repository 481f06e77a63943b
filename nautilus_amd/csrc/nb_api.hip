// C ABI of the hot path (include/nautilus_hip.h): host-side packing of bound
// descriptions into HBM blobs and thin launch wrappers.  No torch types, no
// exceptions across the boundary.
#include "nb_common.h"
#include "../../include/nautilus_hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <csignal>
#include <cstring>
#include <ctime>
#include <execinfo.h>
#include <limits>
#include <unistd.h>
#include <string>
#include <vector>

// launchers implemented in the kernel translation units
bool nb_eval_fast_eligible(int n_dim, int K, int M, int E, bool sample);
int nb_launch_eval_fast(const double* blob_dev, int n_dim, bool sample, int m,
                        int recentre, const double* x, const long long* idx,
                        long long n,
                        unsigned char* out_u8, double* out_f64,
                        unsigned long long seed, unsigned long long offset,
                        hipStream_t stream);
int nb_launch_cand(int dt, int n_dim, const double* const* blobs_dev,
                   const int* group_base_dev, int nb, int n_groups, int b_off,
                   int g_off, int accumulate, int mode, const double* x,
                   long long n, unsigned char* st, int* first, int* work,
                   unsigned long long seed, unsigned long long offset,
                   int** dense_out, int** totals_out, long long* n_pad_out,
                   hipStream_t stream);
long long nb_cand_work_bytes(int dt, int mode, long long n, int n_groups);
int nb_launch_eval_fast_batch(const double* blob0_dev, int n_dim, int recentre,
                              const double* x, const void* groups_dev,
                              int n_groups, const int* totals_dev,
                              const int* dense_dev, long long n_pad,
                              long long n_upper, int out_mode,
                              unsigned char* st, int* first,
                              hipStream_t stream);
int nb_launch_eval(int dt, const double* const* blobs_dev, int nb, int mode,
                   const double* x, long long n, unsigned char* out_u8,
                   int* out_i32, double* out_f64, unsigned long long seed,
                   unsigned long long offset, hipStream_t stream);
void nb_eval_set_counters(unsigned long long* dev);
int nb_launch_draw(const double* blob_dev, int n_dim, unsigned long long seed,
                   unsigned long long offset, long long n, double* x_out,
                   hipStream_t stream);
long long nb_compact_chunks(long long n);
int nb_launch_compact(const double* x, const unsigned char* flags,
                      unsigned char mask, unsigned char flip, long long n,
                      int n_dim, double* out,
                      long long* src_idx, long long* counts,
                      long long* chunk_counts, hipStream_t stream);
int nb_lse_blocks(long long n);
int nb_launch_shell_stats(const double* log_l, long long n, double threshold,
                          double* out, double* partial, hipStream_t stream);
int nb_launch_philox(unsigned long long seed, unsigned long long offset,
                     unsigned block, unsigned tag, long long n, double* u,
                     hipStream_t stream);
int nb_run_mfma_peak(int iters, double* tflops);
int nb_launch_mvee_batch(int n_problems, const double* const* xs,
                         const long long* n, int d, int n_max, int n_batch,
                         double* const* u, double* work, hipStream_t stream);
long long nb_mvee_work_doubles_impl(int n_problems, long long n_max, int d,
                                    int n_batch);
int nb_launch_moments(const double* x, const double* w, long long n, int d,
                      double scale, double* out, double* work,
                      hipStream_t stream);
long long nb_moments_work_doubles_impl(long long n, int d);
int nb_launch_quadform_max(const double* x, long long n, int d,
                           const double* p_dev, double* out, double* work,
                           hipStream_t stream);
long long nb_quadform_work_doubles_impl();
int nb_launch_live_append(const double* ll, long long n, const double* thr,
                          double* pool, int* pool_n, int cap, int* overflow,
                          hipStream_t stream);
int nb_launch_live_select(const double* pool, const int* pool_n, int cap,
                          int k, double* out, int* out_n, double* thr,
                          double* stats, hipStream_t stream);
int nb_launch_live_stats(const double* ll, long long n, const double* thr,
                         double* out, hipStream_t stream);
int nb_launch_rosenbrock(const double* u, long long n, int d, double lo,
                         double hi, double a, double* out, hipStream_t stream);
int nb_launch_funnel(const double* u, long long n, int d, double mu, double s0,
                     double k, double c, double* out, hipStream_t stream);
long long nb_whiten_work_doubles_impl(long long n, int d);
int nb_launch_whiten(const double* x, long long n, int d, double* xw,
                     double* mean, double* sd, double* w, double* work,
                     hipStream_t stream);
int nb_launch_transform(const double* ell_block, int dt, int n_dim,
                        const double* x, long long n, double* y,
                        hipStream_t stream);
int nb_launch_standardize(const double* x, long long n, int d, double* mean,
                          double* scale, double* out, hipStream_t stream);
int nb_launch_prior(const double* u, long long n, int d,
                    const unsigned char* kind, const double* loc,
                    const double* scale, double* out, hipStream_t stream);
long long nb_gmm_out_stride_impl(int d);
long long nb_gmm_scratch_stride_impl(long long n, int d);
long long nb_gmm_work_doubles_impl(long long n, int d, int n_init);
long long nb_gmm_logp_offset_impl(long long n, int d);
int nb_gmm_set_cap_impl(int max_wgs);
int nb_launch_gmm(const double* x, long long n, int d, int n_init,
                  unsigned long long seed, double tol, double reg, int max_iter,
                  const int* init_labels, double* out, double* scratch,
                  hipStream_t stream);
int nb_launch_phase_shift(double* x, long long n, int n_dim, const double* s,
                          const unsigned char* on, int inverse,
                          hipStream_t stream);
int nb_launch_ell_stream(const double* cvec, const double* binv, int n_dim,
                         const double* x, long long n, unsigned char* mask,
                         hipStream_t stream);

static thread_local std::string g_error;

void nb_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
}

struct nb_bound {
  double* blob_dev = nullptr;
  const double** self_list_dev = nullptr;   // device array {blob_dev}
  // two-stage queries: the bound's (neural bound) groups, {0} as group base
  FastGroup* groups_dev = nullptr;
  int* group_base_dev = nullptr;
  int n_groups = 0;
  int64_t off_shift = 0, neural_stride = 0;
  int64_t n_doubles = 0;
  int n_dim = 0, dt = 0, K = 0, M = 0, E = 0;
  bool single_full_ellipsoid = false;
  int64_t off_stream = 0;
  int64_t off_members = 0, off_neural = 0;
};

static inline int64_t nb_hdr_host_off_members(const nb_bound* b) {
  return b->off_members;
}

struct nb_boundlist {
  const double** ptrs_dev = nullptr;
  int n = 0, dt = 0, n_dim = 0;
  int blocks_single = 0;   // n == 1: ellipsoid blocks (K + M) of the bound
  // two-stage queries: groups of all bounds in list order, first group of
  // every bound (n + 1 entries; host copy for splitting long lists)
  FastGroup* groups_dev = nullptr;
  int* group_base_dev = nullptr;
  std::vector<int> group_base;
  int n_groups = 0;
  const double* blob0 = nullptr;   // any blob of the list (n_dim, strides)
};

// groups of one bound, appended to `out`
static void nb_append_groups(const nb_bound* b, int pos,
                             std::vector<FastGroup>& out) {
  if (b->E < 1) return;
  for (int m = 0; m < b->M; ++m) {
    FastGroup g;
    g.nb = b->blob_dev + b->off_neural + (int64_t)m * b->neural_stride;
    g.shift = b->off_shift != 0 ? b->blob_dev + b->off_shift : nullptr;
    g.E = b->E;
    g.b = pos;
    out.push_back(g);
  }
}

namespace {

const double INF = std::numeric_limits<double>::infinity();

inline void put_i64(std::vector<double>& buf, size_t at, int64_t v) {
  std::memcpy(&buf[at], &v, sizeof v);
}

// K permutation of the ellipsoid stage (nb_eval.hip / nb_stream.hip): slot
// (ks, lg) <-> feature 8*(ks>>1) + 2*lg + (ks&1).
inline int slot_of_feature(int f) {
  const int j = f >> 3, r = f & 7;
  const int lg = r >> 1, o = r & 1;
  return 4 * (2 * j + o) + lg;
}

// fills one ell block at buf[at ...]; returns false on invalid input
bool fill_ell_block(std::vector<double>& buf, size_t at, int n_dim, int dt,
                    const nb_member_desc& m, bool is_neural) {
  const int dp = 16 * dt;
  if (m.n_ell < 0 || m.n_ell > n_dim) {
    nb_set_error("member n_ell=%d out of range for n_dim=%d", m.n_ell, n_dim);
    return false;
  }
  put_i64(buf, at, m.n_ell);
  double* lo = &buf[at + 2];
  double* hi = lo + dp;
  double* c = hi + dp;
  double* tiles = c + dp;
  std::vector<int> idx(m.n_ell);
  std::vector<char> is_ell(n_dim, 0);
  for (int i = 0; i < m.n_ell; ++i) {
    idx[i] = m.idx_ell ? m.idx_ell[i] : i;
    if (idx[i] < 0 || idx[i] >= n_dim || (i > 0 && idx[i] <= idx[i - 1])) {
      nb_set_error("idx_ell must be strictly increasing within [0, n_dim)");
      return false;
    }
    is_ell[idx[i]] = 1;
  }
  bool any_box = false;
  for (int f = 0; f < dp; ++f) {
    const bool boxed = f < n_dim && !is_ell[f] && !m.free_dims && !is_neural;
    const int sl = slot_of_feature(f);
    lo[sl] = boxed ? 0.0 : -INF;
    hi[sl] = boxed ? 1.0 : INF;
    c[sl] = 0.0;
    any_box |= boxed;
  }
  // [1]: members: "has finite box limits" (nb_cand.hip skips the box test
  // otherwise); neural blocks: the squared bounding radius, set by the caller
  put_i64(buf, at + 1, any_box ? 1 : 0);
  for (int i = 0; i < m.n_ell; ++i) c[slot_of_feature(idx[i])] = m.c[i];
  for (int i = 0; i < m.n_ell; ++i) {
    for (int j = 0; j < m.n_ell; ++j) {
      const double v = m.B_inv[(size_t)i * m.n_ell + j];
      if (j > i) {
        if (v != 0.0) {
          nb_set_error("B_inv must be lower triangular (basic.py:308-309)");
          return false;
        }
        continue;
      }
      // W0[k][h] = B_inv[h][k]; the K index of the tile is stored in slot
      // order: tile (kt, ht), k-step s, lane (li, lg) at s*64 + lg*16 + li
      const int h = idx[i], k = idx[j];
      const int sl = slot_of_feature(k);          // = 4*ks + lg
      const int ks = sl >> 2, lgk = sl & 3;
      tiles[((size_t)(ks >> 2) * dt + (h >> 4)) * NB_TILE + (ks & 3) * 64 +
            lgk * 16 + (h & 15)] = v;
    }
  }
  return true;
}

void put_weight(double* tiles, int ht_n, int k, int h, double v) {
  tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE + (k & 15) * 16 +
        (h & 15)] = v;
}

void fill_net(double* net, int n_dim, int kt1, const double* const* coefs,
              const double* const* intercepts) {
  double* w1 = net;
  double* w2 = w1 + (size_t)kt1 * NB_HT1 * NB_TILE;
  double* w3 = w2 + (size_t)NB_HT1 * NB_HT2 * NB_TILE;
  double* w4 = w3 + (size_t)NB_HT2 * NB_HT3 * NB_TILE;
  for (int k = 0; k < n_dim; ++k)
    for (int h = 0; h < NB_H1; ++h)
      put_weight(w1, NB_HT1, k, h, coefs[0][(size_t)k * NB_H1 + h]);
  for (int h = 0; h < NB_H1; ++h)
    put_weight(w1, NB_HT1, n_dim, h, intercepts[0][h]);
  for (int k = 0; k < NB_H1; ++k)
    for (int h = 0; h < NB_H2; ++h)
      put_weight(w2, NB_HT2, k, h, coefs[1][(size_t)k * NB_H2 + h]);
  for (int h = 0; h < NB_H2; ++h)
    put_weight(w2, NB_HT2, NB_H1, h, intercepts[1][h]);
  for (int k = 0; k < NB_H2; ++k)
    for (int h = 0; h < NB_H3; ++h)
      put_weight(w3, NB_HT3, k, h, coefs[2][(size_t)k * NB_H3 + h]);
  for (int h = 0; h < NB_H3; ++h)
    put_weight(w3, NB_HT3, NB_H2, h, intercepts[2][h]);
  for (int k = 0; k < NB_H3; ++k) put_weight(w4, 1, k, 0, coefs[3][k]);
  put_weight(w4, 1, NB_H3, 0, intercepts[3][0]);
}

inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

}  // namespace

extern "C" {

int nb_abi_version(void) { return NB_ABI_VERSION; }

const char* nb_last_error(void) { return g_error.c_str(); }

int nb_bound_create(const nb_bound_desc* d, nb_bound** out) {
  if (d == nullptr || out == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  const int n_dim = d->n_dim;
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT) {
    nb_set_error("n_dim=%d unsupported (1..%d)", n_dim, 16 * NB_MAX_DT);
    return NB_ERR_UNSUPPORTED;
  }
  const int dt = (n_dim + 15) / 16, dp = 16 * dt;
  const int K = d->n_members, M = d->n_neural;
  if (K < 0 || M < 0 || (K > 0 && d->members == nullptr) ||
      (M > 0 && d->neural == nullptr)) {
    nb_set_error("inconsistent member / neural counts");
    return NB_ERR_ARG;
  }
  int E = 0;
  for (int m = 0; m < M; ++m) {
    const int e = d->neural[m].mlp ? d->neural[m].mlp->n_networks : 0;
    if (m > 0 && e != E) {
      nb_set_error("all neural bounds must have the same number of networks");
      return NB_ERR_ARG;
    }
    E = e;
    if (d->neural[m].ellipsoid.n_ell != n_dim) {
      nb_set_error("NeuralBound ellipsoid must span all dimensions");
      return NB_ERR_ARG;
    }
  }
  const int kt1 = (n_dim + 1 + 15) / 16;
  const int64_t ell_size = nb_ell_block_size(dt);
  const int64_t net_stride = (int64_t)nb_net_tiles(kt1) * NB_TILE;
  const int64_t neural_stride = ell_size + 2 + 2 * dp + (int64_t)E * net_stride;
  const int64_t draw_stride = 2 + 4 * dp + (int64_t)dp * (dp + 1) / 2 + 16;

  int64_t off = NB_HDR;
  const int64_t off_cdf = off; off += ((K > 0 ? K : 1) + 1) / 2 * 2;
  const int64_t off_ulo = off; off += dp;
  const int64_t off_uhi = off; off += dp;
  const int64_t off_members = off; off += (int64_t)K * ell_size;
  const int64_t off_neural = off; off += (int64_t)M * neural_stride;
  const int64_t off_draw = off; off += (int64_t)K * draw_stride;
  const bool single_full = (K == 1 && M == 0 && !d->unit_cube &&
                            d->members[0].n_ell == n_dim);
  const int64_t off_stream = off;
  if (single_full) off += dp + (int64_t)dt * (dt + 1) / 2 * NB_TILE;
  if (d->n_periodic < 0 || d->n_periodic > n_dim ||
      (d->n_periodic > 0 && (d->periodic == nullptr || d->centers == nullptr))) {
    nb_set_error("bad periodic description");
    return NB_ERR_ARG;
  }
  const int64_t off_shift = off;
  if (d->n_periodic > 0) off += 2 * dp;
  const int64_t total = off;

  std::vector<double> buf((size_t)total, 0.0);
  put_i64(buf, NB_H_NDIM, n_dim);
  put_i64(buf, NB_H_DT, dt);
  put_i64(buf, NB_H_K, K);
  put_i64(buf, NB_H_USECUBE, d->unit_cube ? 1 : 0);
  put_i64(buf, NB_H_M, M);
  put_i64(buf, NB_H_E, E);
  put_i64(buf, NB_H_OFF_CDF, off_cdf);
  put_i64(buf, NB_H_OFF_ULO, off_ulo);
  put_i64(buf, NB_H_OFF_UHI, off_uhi);
  put_i64(buf, NB_H_OFF_MEMBERS, off_members);
  put_i64(buf, NB_H_ELL_STRIDE, ell_size);
  put_i64(buf, NB_H_OFF_NEURAL, off_neural);
  put_i64(buf, NB_H_NEURAL_STRIDE, neural_stride);
  put_i64(buf, NB_H_OFF_DRAW, off_draw);
  put_i64(buf, NB_H_DRAW_STRIDE, draw_stride);
  put_i64(buf, NB_H_NET_STRIDE, net_stride);
  put_i64(buf, NB_H_KT1, kt1);
  put_i64(buf, NB_H_TOTAL, total);
  put_i64(buf, NB_H_OFF_STREAM, single_full ? off_stream : 0);
  put_i64(buf, NB_H_OFF_SHIFT, d->n_periodic > 0 ? off_shift : 0);
  for (int i = 0; i < d->n_periodic; ++i) {
    const int f = d->periodic[i];
    if (f < 0 || f >= n_dim) {
      nb_set_error("periodic index %d out of range", f);
      return NB_ERR_ARG;
    }
    // periodic.py:69-71: the forward shift adds (-center + 0.5)
    buf[off_shift + slot_of_feature(f)] = -d->centers[i] + 0.5;
    buf[off_shift + dp + slot_of_feature(f)] = 1.0;
  }
  if (single_full) {
    // stream block: c, then lower-triangular tiles with the K permutation of
    // nb_stream.hip (slot 4kt+s of lane group lg <-> feature
    // 16kt + 8(s>>1) + 2lg + (s&1))
    const nb_member_desc& md = d->members[0];
    for (int i = 0; i < n_dim; ++i) buf[off_stream + i] = md.c[i];
    double* st = &buf[off_stream + dp];
    for (int ht = 0; ht < dt; ++ht)
      for (int kt = 0; kt <= ht; ++kt)
        for (int s4 = 0; s4 < 4; ++s4)
          for (int lg = 0; lg < 4; ++lg)
            for (int li = 0; li < 16; ++li) {
              const int k = 16 * kt + 8 * (s4 >> 1) + 2 * lg + (s4 & 1);
              const int h = 16 * ht + li;
              double v = 0.0;
              if (h < n_dim && k < n_dim && k <= h)
                v = md.B_inv[(size_t)h * n_dim + k];
              st[((size_t)(ht * (ht + 1)) / 2 + kt) * NB_TILE + s4 * 64 +
                 lg * 16 + li] = v;
            }
  }

  // member CDF over softmax(log_v_all), union.py:308
  if (K > 0) {
    double mx = -INF;
    for (int m = 0; m < K; ++m)
      mx = std::fmax(mx, d->log_v_all ? d->log_v_all[m] : 0.0);
    std::vector<double> p(K);
    double sum = 0.0;
    for (int m = 0; m < K; ++m) {
      p[m] = std::exp((d->log_v_all ? d->log_v_all[m] : 0.0) - mx);
      sum += p[m];
    }
    double run = 0.0;
    for (int m = 0; m < K; ++m) {
      run += p[m] / sum;
      buf[off_cdf + m] = run;
    }
    buf[off_cdf + K - 1] = 1.0;
  }
  for (int f = 0; f < dp; ++f) {
    const bool boxed = d->unit_cube && f < n_dim;
    buf[off_ulo + slot_of_feature(f)] = boxed ? 0.0 : -INF;
    buf[off_uhi + slot_of_feature(f)] = boxed ? 1.0 : INF;
  }

  for (int m = 0; m < K; ++m) {
    const nb_member_desc& md = d->members[m];
    if (!fill_ell_block(buf, off_members + m * ell_size, n_dim, dt, md, false))
      return NB_ERR_ARG;
    // compact draw block
    const size_t at = off_draw + m * draw_stride;
    std::vector<char> is_ell(n_dim, 0);
    for (int i = 0; i < md.n_ell; ++i)
      is_ell[md.idx_ell ? md.idx_ell[i] : i] = 1;
    int nc = 0;
    for (int f = 0; f < n_dim; ++f) {
      if (!is_ell[f] && !md.free_dims) {
        put_i64(buf, at + 2 + dp + nc, f);
        ++nc;
      }
    }
    put_i64(buf, at, md.n_ell);
    put_i64(buf, at + 1, nc);
    // slot of every column: ellipsoid dims first, then the cube dims
    {
      int cube_slot = md.n_ell;
      for (int f = 0; f < n_dim; ++f)
        if (!is_ell[f]) put_i64(buf, at + 2 + 2 * dp + f, cube_slot++);
    }
    for (int i = 0; i < md.n_ell; ++i) {
      const int f = md.idx_ell ? md.idx_ell[i] : i;
      put_i64(buf, at + 2 + i, f);
      put_i64(buf, at + 2 + 2 * dp + f, i);
      buf[at + 2 + 3 * dp + i] = md.c[i];
      for (int j = 0; j <= i; ++j)
        buf[at + 2 + 4 * dp + (size_t)i * (i + 1) / 2 + j] =
            md.B[(size_t)i * md.n_ell + j];
    }
  }

  for (int m = 0; m < M; ++m) {
    const nb_neural_desc& nd = d->neural[m];
    const size_t at = off_neural + m * neural_stride;
    if (!fill_ell_block(buf, at, n_dim, dt, nd.ellipsoid, true))
      return NB_ERR_ARG;
    buf[at + ell_size] = nd.score_predict_min - 1e-9;   // bounds/neural.py:125
    {
      // bounding sphere of the ellipsoid: x inside => |x - c|^2 <= radius2
      double r2 = nd.radius2;
      if (!(r2 > 0.0)) {
        // conservative fallback: lambda_max(B B^T) <= min(Gershgorin, ||B||_F^2)
        const int n = nd.ellipsoid.n_ell;
        const double* B = nd.ellipsoid.B;
        double fro = 0.0, ger = 0.0;
        for (int i = 0; i < n * n; ++i) fro += B[i] * B[i];
        for (int i = 0; i < n; ++i) {
          double rowsum = 0.0;
          for (int j = 0; j < n; ++j) {
            double aij = 0.0;
            for (int k = 0; k < n; ++k) aij += B[i * n + k] * B[j * n + k];
            rowsum += std::fabs(aij);
          }
          ger = std::fmax(ger, rowsum);
        }
        r2 = std::fmin(fro, ger) * (1.0 + 1e-9);
      }
      buf[at + 1] = r2;
    }
    double* mean = &buf[at + ell_size + 2];
    double* scale = mean + dp;
    for (int f = 0; f < dp; ++f) { mean[f] = 0.0; scale[f] = 1.0; }
    if (nd.mlp != nullptr) {
      for (int f = 0; f < n_dim; ++f) {
        mean[f] = nd.mlp->mean[f];
        scale[f] = 1.0 / nd.mlp->scale[f];   // the kernel multiplies
      }
      for (int e = 0; e < E; ++e)
        fill_net(scale + dp + (size_t)e * net_stride, n_dim, kt1,
                 nd.mlp->coefs + 4 * e, nd.mlp->intercepts + 4 * e);
    }
  }

  nb_bound* b = new nb_bound();
  b->n_doubles = total;
  b->n_dim = n_dim; b->dt = dt; b->K = K; b->M = M; b->E = E;
  b->single_full_ellipsoid = single_full;
  b->off_stream = off_stream;
  b->off_members = off_members;
  b->off_neural = off_neural;
  b->off_shift = nb_hdr(buf.data(), NB_H_OFF_SHIFT);
  b->neural_stride = nb_hdr(buf.data(), NB_H_NEURAL_STRIDE);
  // (the kernels stage parts of the blob in whole 1 KB pieces -- dma_weights,
  // the ellipsoid block of nb_eval_fast.hip -- so the last piece of the last
  // block may read up to 1 KB past its end: keep that inside the allocation)
  constexpr size_t SLACK = 512;
  hipError_t e = hipMalloc((void**)&b->blob_dev,
                           ((size_t)total + SLACK) * sizeof(double));
  if (e == hipSuccess)
    e = hipMemset(b->blob_dev + total, 0, SLACK * sizeof(double));
  if (e == hipSuccess)
    e = hipMemcpy(b->blob_dev, buf.data(), (size_t)total * sizeof(double),
                  hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc((void**)&b->self_list_dev, sizeof(double*));
  if (e == hipSuccess)
    e = hipMemcpy(b->self_list_dev, &b->blob_dev, sizeof(double*),
                  hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    std::vector<FastGroup> groups;
    nb_append_groups(b, 0, groups);
    b->n_groups = (int)groups.size();
    const int base[2] = {0, b->n_groups};
    e = hipMalloc((void**)&b->group_base_dev, sizeof base);
    if (e == hipSuccess)
      e = hipMemcpy(b->group_base_dev, base, sizeof base,
                    hipMemcpyHostToDevice);
    if (e == hipSuccess && b->n_groups > 0) {
      e = hipMalloc((void**)&b->groups_dev, groups.size() * sizeof(FastGroup));
      if (e == hipSuccess)
        e = hipMemcpy(b->groups_dev, groups.data(),
                      groups.size() * sizeof(FastGroup), hipMemcpyHostToDevice);
    }
  }
  if (e != hipSuccess) {
    nb_set_error("bound upload failed: %s", hipGetErrorString(e));
    nb_bound_destroy(b);
    return NB_ERR_HIP;
  }
  *out = b;
  return NB_OK;
}

int nb_bound_destroy(nb_bound* b) {
  if (b == nullptr) return NB_OK;
  if (b->blob_dev) (void)hipFree(b->blob_dev);
  if (b->self_list_dev) (void)hipFree(b->self_list_dev);
  if (b->groups_dev) (void)hipFree(b->groups_dev);
  if (b->group_base_dev) (void)hipFree(b->group_base_dev);
  delete b;
  return NB_OK;
}

int64_t nb_bound_nbytes(const nb_bound* b) {
  return b ? b->n_doubles * (int64_t)sizeof(double) : 0;
}

int nb_boundlist_create(nb_bound* const* bounds, int32_t n,
                        nb_boundlist** out) {
  if (n < 0 || (n > 0 && bounds == nullptr) || out == nullptr) {
    nb_set_error("bad bound list");
    return NB_ERR_ARG;
  }
  nb_boundlist* l = new nb_boundlist();
  l->n = n;
  std::vector<const double*> ptrs(n);
  for (int i = 0; i < n; ++i) {
    if (i > 0 && bounds[i]->n_dim != bounds[0]->n_dim) {
      nb_set_error("bounds of a list must share n_dim");
      delete l;
      return NB_ERR_ARG;
    }
    ptrs[i] = bounds[i]->blob_dev;
  }
  if (n > 0) {
    l->dt = bounds[0]->dt;
    l->n_dim = bounds[0]->n_dim;
    l->blob0 = bounds[0]->blob_dev;
    if (n == 1) l->blocks_single = bounds[0]->K + bounds[0]->M;
    hipError_t e = hipMalloc((void**)&l->ptrs_dev, n * sizeof(double*));
    if (e == hipSuccess)
      e = hipMemcpy(l->ptrs_dev, ptrs.data(), n * sizeof(double*),
                    hipMemcpyHostToDevice);
    std::vector<FastGroup> groups;
    l->group_base.assign(1, 0);
    for (int i = 0; i < n; ++i) {
      nb_append_groups(bounds[i], i, groups);
      l->group_base.push_back((int)groups.size());
    }
    l->n_groups = (int)groups.size();
    if (e == hipSuccess)
      e = hipMalloc((void**)&l->group_base_dev, (n + 1) * sizeof(int));
    if (e == hipSuccess)
      e = hipMemcpy(l->group_base_dev, l->group_base.data(),
                    (n + 1) * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess && l->n_groups > 0) {
      e = hipMalloc((void**)&l->groups_dev, groups.size() * sizeof(FastGroup));
      if (e == hipSuccess)
        e = hipMemcpy(l->groups_dev, groups.data(),
                      groups.size() * sizeof(FastGroup), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
      nb_set_error("bound list upload failed: %s", hipGetErrorString(e));
      nb_boundlist_destroy(l);
      return NB_ERR_HIP;
    }
  }
  *out = l;
  return NB_OK;
}

int nb_boundlist_destroy(nb_boundlist* l) {
  if (l == nullptr) return NB_OK;
  if (l->ptrs_dev) (void)hipFree(l->ptrs_dev);
  if (l->groups_dev) (void)hipFree(l->groups_dev);
  if (l->group_base_dev) (void)hipFree(l->group_base_dev);
  delete l;
  return NB_OK;
}

int nb_contains(const nb_bound* b, const double* x, int64_t n, uint8_t* mask,
                void* stream) {
  return nb_launch_eval(b->dt, b->self_list_dev, 1, 0, x, n, mask, nullptr,
                        nullptr, 0, 0, as_stream(stream));
}

int nb_contains_any(const nb_boundlist* l, const double* x, int64_t n,
                    uint8_t* mask, void* stream) {
  if (l->n == 0) {
    NB_HIP_CHECK(hipMemsetAsync(mask, 0, (size_t)n, as_stream(stream)));
    return NB_OK;
  }
  return nb_launch_eval(l->dt, l->ptrs_dev, l->n, 0, x, n, mask, nullptr,
                        nullptr, 0, 0, as_stream(stream));
}

int nb_first_containing(const nb_boundlist* l, const double* x, int64_t n,
                        int32_t* idx, void* stream) {
  if (l->n == 0) {
    NB_HIP_CHECK(hipMemsetAsync(idx, 0xFF, (size_t)n * 4, as_stream(stream)));
    return NB_OK;
  }
  return nb_launch_eval(l->dt, l->ptrs_dev, l->n, 1, x, n, nullptr, idx,
                        nullptr, 0, 0, as_stream(stream));
}

int nb_member_count(const nb_bound* b, const double* x, int64_t n,
                    uint8_t* count, void* stream) {
  return nb_launch_eval(b->dt, b->self_list_dev, 1, 3, x, n, count, nullptr,
                        nullptr, 0, 0, as_stream(stream));
}

int nb_neural_score(const nb_bound* b, const double* x, int64_t n, double* out,
                    void* stream) {
  if (b->M < 1) {
    nb_set_error("bound has no neural bound");
    return NB_ERR_ARG;
  }
#ifndef NB_NO_FAST_EVAL
  if (nb_eval_fast_eligible(b->n_dim, b->K, b->M, b->E, false))
    return nb_launch_eval_fast(b->blob_dev, b->n_dim, false, 0, 0, x, nullptr, n,
                               nullptr, out, 0, 0, as_stream(stream));
#endif
  return nb_launch_eval(b->dt, b->self_list_dev, 1, 4, x, n, nullptr, nullptr,
                        out, 0, 0, as_stream(stream));
}

// debugging aid: native backtrace on SIGABRT / SIGSEGV (NB_ABORT_TRACE=1)
static void nb_abort_trace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);
  const char msg[] = "[nautilus_hip] fatal signal, native backtrace:\n";
  (void)!write(2, msg, sizeof msg - 1);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

// (not an export: installed when the library is loaded with NB_ABORT_TRACE set)
__attribute__((constructor)) static void nb_install_abort_trace(void) {
  if (getenv("NB_ABORT_TRACE") == nullptr) return;
  signal(SIGABRT, nb_abort_trace);
  signal(SIGSEGV, nb_abort_trace);
}

// ---- two-stage queries (nb_cand.hip + nb_eval_fast.hip BATCH) --------------
// groups per slice of a long list: the second stage's pass table holds 1024
// (NB_LIST_SLICE_GROUPS lowers it: the test of the slicing itself)
static int nb_slice_groups() {
  const char* e = getenv("NB_LIST_SLICE_GROUPS");
  const int v = e != nullptr ? atoi(e) : 1024;
  return v < 1 ? 1 : (v > 1024 ? 1024 : v);
}
#define NB_SLICE_GROUPS nb_slice_groups()

int64_t nb_list_eval_work_bytes(const nb_boundlist* l, int64_t n) {
  const int g = l->n_groups < NB_SLICE_GROUPS ? l->n_groups : NB_SLICE_GROUPS;
  // (a slice never splits a bound: allow for the largest bound's groups)
  int most = 0;
  for (int i = 0; i < l->n; ++i) {
    const int k = l->group_base[i + 1] - l->group_base[i];
    most = k > most ? k : most;
  }
  return nb_cand_work_bytes(l->dt, 0, n, (g > most ? g : most) + 1);
}

int64_t nb_accept_staged_work_bytes(const nb_bound* b, int64_t n) {
  return nb_cand_work_bytes(b->dt, 2, n, b->n_groups + 1);
}

static int nb_stage_two(const double* blob0, int n_dim, int recentre,
                        const double* x, const FastGroup* groups, int n_groups,
                        int* totals, int* dense, long long n_pad, long long n,
                        int mode, uint8_t* st, int32_t* first,
                        hipStream_t stream) {
  if (n_groups == 0) return NB_OK;
  // every row is a candidate of every group at most once
  return nb_launch_eval_fast_batch(blob0, n_dim, recentre, x, groups, n_groups,
                                   totals, dense, n_pad,
                                   (long long)n * (n_groups < 8 ? n_groups : 8),
                                   mode, st, first, stream);
}

int nb_list_eval(const nb_boundlist* l, int32_t mode, const double* x,
                 int64_t n, uint8_t* st, int32_t* first, void* work,
                 int64_t work_bytes, void* stream) {
  if (mode != 0 && mode != 1) {
    nb_set_error("nb_list_eval: mode must be 0 (any) or 1 (first)");
    return NB_ERR_ARG;
  }
  if (mode == 1 && first == nullptr) {
    nb_set_error("nb_list_eval: mode 1 needs the `first` output");
    return NB_ERR_ARG;
  }
  if (n <= 0) return NB_OK;
  if (l->n == 0) {
    NB_HIP_CHECK(hipMemsetAsync(st, 0, (size_t)n, as_stream(stream)));
    if (first != nullptr)
      NB_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)first, 0x7fffffff,
                                     (size_t)n, as_stream(stream)));
    return NB_OK;
  }
  if (work_bytes < nb_list_eval_work_bytes(l, n)) {
    nb_set_error("nb_list_eval: work space of %lld bytes, need %lld",
                 (long long)work_bytes,
                 (long long)nb_list_eval_work_bytes(l, n));
    return NB_ERR_ARG;
  }
  // slices of the list with at most NB_SLICE_GROUPS groups each
  int b0 = 0;
  while (b0 < l->n) {
    int b1 = b0 + 1;
    while (b1 < l->n && l->group_base[b1 + 1] - l->group_base[b0] <=
                            NB_SLICE_GROUPS)
      ++b1;
    const int g0 = l->group_base[b0], ng = l->group_base[b1] - g0;
    int *dense = nullptr, *totals = nullptr;
    long long n_pad = 0;
    int rc = nb_launch_cand(l->dt, l->n_dim, l->ptrs_dev + b0, l->group_base_dev + b0,
                            b1 - b0, ng, b0, g0, b0 > 0 ? 1 : 0, mode, x, n,
                            st, first, (int*)work, 0, 0, &dense, &totals,
                            &n_pad, as_stream(stream));
    if (rc != NB_OK) return rc;
    rc = nb_stage_two(l->blob0, l->n_dim, 1, x, l->groups_dev + g0, ng, totals,
                      dense, n_pad, n, mode, st, first, as_stream(stream));
    if (rc != NB_OK) return rc;
    b0 = b1;
  }
  return NB_OK;
}

int nb_accept_staged(const nb_bound* b, uint64_t seed, uint64_t offset,
                     const double* x, int64_t n, uint8_t* flags, void* work,
                     int64_t work_bytes, int64_t* totals_offset, void* stream) {
  if (n <= 0) return NB_OK;
  if (work_bytes < nb_accept_staged_work_bytes(b, n)) {
    nb_set_error("nb_accept_staged: work space of %lld bytes, need %lld",
                 (long long)work_bytes,
                 (long long)nb_accept_staged_work_bytes(b, n));
    return NB_ERR_ARG;
  }
  int *dense = nullptr, *totals = nullptr;
  long long n_pad = 0;
  // NB_STAGE_TIMING=1 (debugging aid): HIP events around the two stages; the
  // times of a call are printed by the NEXT call, so that nothing waits for
  // the device inside the measured sequence
  static const bool timing = getenv("NB_STAGE_TIMING") != nullptr;
  static hipEvent_t ev[3][3];
  static double host_us[3][4];
  static int ev_calls = 0;
  hipEvent_t* e = nullptr;
  double* hu = nullptr;
  auto now_us = []() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
  };
  if (timing) {
    if (ev_calls == 0)
      for (int i = 0; i < 9; ++i) (void)hipEventCreate(&ev[i / 3][i % 3]);
    if (ev_calls > 1) {
      // call k - 2 and the gap to call k - 1 (both finished or in flight
      // behind nothing this call waits for but their own events)
      hipEvent_t* p = ev[(ev_calls - 2) % 3];
      hipEvent_t* q = ev[(ev_calls - 1) % 3];
      const double* h = host_us[(ev_calls - 2) % 3];
      const double* hq = host_us[(ev_calls - 1) % 3];
      float t_geo = 0.f, t_emu = 0.f, t_gap = 0.f;
      (void)hipEventSynchronize(q[0]);
      (void)hipEventElapsedTime(&t_geo, p[0], p[1]);
      (void)hipEventElapsedTime(&t_emu, p[1], p[2]);
      (void)hipEventElapsedTime(&t_gap, p[2], q[0]);
      fprintf(stderr, "[stage] call %d: device: candidates + compaction %.3f "
              "ms, batched scores %.3f ms, then idle until the next call's "
              "first event %.3f ms; host: stage-one launches %.0f us, "
              "stage-two launch %.0f us, until the next call %.0f us\n",
              ev_calls - 2, t_geo, t_emu, t_gap, h[1] - h[0], h[2] - h[1],
              hq[0] - h[2]);
    }
    e = ev[ev_calls % 3];
    hu = host_us[ev_calls % 3];
    ++ev_calls;
    hu[0] = now_us();
    (void)hipEventRecord(e[0], as_stream(stream));
  }
  int rc = nb_launch_cand(b->dt, b->n_dim, b->self_list_dev, b->group_base_dev, 1,
                          b->n_groups, 0, 0, 0, 2, x, n, flags, nullptr,
                          (int*)work, seed, offset, &dense, &totals, &n_pad,
                          as_stream(stream));
  if (rc != NB_OK) return rc;
  if (e != nullptr) {
    (void)hipEventRecord(e[1], as_stream(stream));
    hu[1] = now_us();
  }
  if (totals_offset != nullptr)
    *totals_offset = (int64_t)((char*)totals - (char*)work);
  rc = nb_stage_two(b->blob_dev, b->n_dim, 0, x, b->groups_dev, b->n_groups,
                    totals, dense, n_pad, n, 2, flags, nullptr,
                    as_stream(stream));
  if (e != nullptr) {
    (void)hipEventRecord(e[2], as_stream(stream));
    hu[2] = now_us();
  }
  return rc;
}

int nb_neural_score_rows(const nb_bound* b, int32_t m, int32_t recentre,
                         const double* x, const int64_t* idx, int64_t n,
                         double* out, void* stream) {
  if (m < 0 || m >= b->M || b->E < 1) {
    nb_set_error("nb_neural_score_rows: neural bound %d of %d (E = %d)", m,
                 b->M, b->E);
    return NB_ERR_ARG;
  }
  return nb_launch_eval_fast(b->blob_dev, b->n_dim, false, m, recentre, x,
                             (const long long*)idx, n, nullptr, out, 0, 0,
                             as_stream(stream));
}

int nb_propose(const nb_bound* b, uint64_t seed, uint64_t offset, int64_t n,
               double* x, void* stream) {
  if (b->K < 1) {
    nb_set_error("bound has no outer members to draw from");
    return NB_ERR_ARG;
  }
  return nb_launch_draw(b->blob_dev, b->n_dim, seed, offset, n, x,
                        as_stream(stream));
}

int nb_accept(const nb_bound* b, uint64_t seed, uint64_t offset,
              const double* x, int64_t n, uint8_t* flags, void* stream) {
#ifndef NB_NO_FAST_EVAL
  // one neural bound, at most one outer member: the pipelined kernel
  if (nb_eval_fast_eligible(b->n_dim, b->K, b->M, b->E, true))
    return nb_launch_eval_fast(b->blob_dev, b->n_dim, true, 0, 0, x, nullptr, n,
                               flags, nullptr, seed, offset,
                               as_stream(stream));
#endif
  return nb_launch_eval(b->dt, b->self_list_dev, 1, 2, x, n, flags, nullptr,
                        nullptr, seed, offset, as_stream(stream));
}

int64_t nb_compact_scratch_bytes(int64_t n) {
  return (nb_compact_chunks(n) * 2 + 2) * (int64_t)sizeof(long long);
}

int nb_compact_rows(const double* x, const uint8_t* flags, uint8_t mask,
                    uint8_t flip, int64_t n, int32_t n_dim, double* out,
                    int64_t* src_idx, int64_t* counts, void* scratch,
                    void* stream) {
  return nb_launch_compact(x, flags, mask, flip, n, n_dim, out,
                           (long long*)src_idx, (long long*)counts,
                           (long long*)scratch, as_stream(stream));
}

int64_t nb_shell_stats_scratch_bytes(int64_t n) {
  return (int64_t)nb_lse_blocks(n) * 4 * sizeof(double);
}

int nb_shell_stats(const double* log_l, int64_t n, double threshold,
                   double* out, void* scratch, void* stream) {
  return nb_launch_shell_stats(log_l, n, threshold, out, (double*)scratch,
                               as_stream(stream));
}

int nb_philox_uniform(uint64_t seed, uint64_t offset, uint32_t block,
                      uint32_t tag, int64_t n, double* u, void* stream) {
  return nb_launch_philox(seed, offset, block, tag, n, u, as_stream(stream));
}

int nb_set_eval_counters(uint64_t* counters_dev) {
  nb_eval_set_counters((unsigned long long*)counters_dev);
  return NB_OK;
}

int nb_mfma_f64_peak(int32_t iters, double* tflops) {
  return nb_run_mfma_peak(iters, tflops);
}

int64_t nb_mvee_work_doubles(int32_t n_problems, int64_t n_points_max,
                             int32_t n_dim, int32_t n_batch) {
  return nb_mvee_work_doubles_impl(n_problems, n_points_max, n_dim, n_batch);
}

int nb_mvee_khachiyan(int32_t n_problems, const double* const* xs,
                      const int64_t* n_points, int32_t n_dim, int32_t n_max,
                      int32_t n_batch, double* const* u, double* work,
                      void* stream) {
  if (n_problems < 1 || xs == nullptr || n_points == nullptr || u == nullptr ||
      work == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  std::vector<long long> n(n_points, n_points + n_problems);
  return nb_launch_mvee_batch(n_problems, xs, n.data(), n_dim, n_max, n_batch,
                              u, work, as_stream(stream));
}

int64_t nb_whiten_work_doubles(int64_t n, int32_t n_dim) {
  return nb_whiten_work_doubles_impl(n, n_dim);
}

int nb_whiten(const double* x, int64_t n, int32_t n_dim, double* xw,
              double* mean, double* sd, double* w, double* work,
              void* stream) {
  if (x == nullptr || xw == nullptr || mean == nullptr || sd == nullptr ||
      w == nullptr || work == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_whiten(x, n, n_dim, xw, mean, sd, w, work,
                          as_stream(stream));
}

int64_t nb_mvee_weights_work_doubles(int64_t n, int32_t n_dim,
                                     int32_t n_batch) {
  return 2 * 16 * NB_MAX_DT + (int64_t)n_dim * n_dim + n * n_dim + 4 +
         nb_whiten_work_doubles_impl(n, n_dim) +
         nb_mvee_work_doubles_impl(1, n, n_dim, n_batch);
}

int nb_mvee_weights(const double* x, int64_t n, int32_t n_dim, int32_t n_max,
                    int32_t n_batch, double* u, double* work, void* stream) {
  if (x == nullptr || u == nullptr || work == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT) {
    nb_set_error("device MVEE supports n_dim <= 128 (got %d)", n_dim);
    return NB_ERR_UNSUPPORTED;
  }
  // whitened copy of the points (the iteration is affine invariant)
  double* mean = work;
  double* scale = work + 16 * NB_MAX_DT;
  double* wmat = scale + 16 * NB_MAX_DT;
  double* xw = wmat + (((int64_t)n_dim * n_dim + 1) & ~(int64_t)1);
  double* wh = xw + ((n * n_dim + 1) & ~(int64_t)1);
  double* rest = wh + nb_whiten_work_doubles_impl(n, n_dim);
  int rc = nb_launch_whiten(x, n, n_dim, xw, mean, scale, wmat, wh,
                            as_stream(stream));
  if (rc != NB_OK) return rc;
  const double* xw_c = xw;
  long long nn = n;
  return nb_launch_mvee_batch(1, &xw_c, &nn, n_dim, n_max, n_batch, &u, rest,
                              as_stream(stream));
}

int64_t nb_moments_work_doubles(int64_t n, int32_t n_dim) {
  return nb_moments_work_doubles_impl(n, n_dim);
}

int nb_weighted_moments(const double* x, const double* w, int64_t n,
                        int32_t n_dim, double scale, double* out, double* work,
                        void* stream) {
  if (x == nullptr || out == nullptr || work == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_moments(x, w, n, n_dim, scale, out, work,
                           as_stream(stream));
}

int64_t nb_quadform_work_doubles(void) {
  return nb_quadform_work_doubles_impl();
}

int nb_quadform_max(const double* x, int64_t n, int32_t n_dim,
                    const double* p_dev, double* out, double* work,
                    void* stream) {
  if (x == nullptr || p_dev == nullptr || out == nullptr || work == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_quadform_max(x, n, n_dim, p_dev, out, work,
                                as_stream(stream));
}

int nb_ellipsoid_transform(const nb_bound* b, const double* x, int64_t n,
                           double* y, void* stream) {
  const double* blk = nullptr;
  if (b->single_full_ellipsoid)
    blk = b->blob_dev + nb_hdr_host_off_members(b);
  else if (b->K == 0 && b->M >= 1)
    blk = b->blob_dev + b->off_neural;
  if (blk == nullptr) {
    nb_set_error("nb_ellipsoid_transform needs an Ellipsoid or a NeuralBound");
    return NB_ERR_ARG;
  }
  return nb_launch_transform(blk, b->dt, b->n_dim, x, n, y, as_stream(stream));
}

int nb_standardize(const double* x, int64_t n, int32_t n_dim, double* mean,
                   double* scale, double* out, void* stream) {
  if (x == nullptr || mean == nullptr || scale == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_standardize(x, n, n_dim, mean, scale, out,
                               as_stream(stream));
}

int nb_prior_transform(const double* u, int64_t n, int32_t n_dim,
                       const uint8_t* kind, const double* loc,
                       const double* scale, double* out, void* stream) {
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT || kind == nullptr ||
      loc == nullptr || scale == nullptr || (n > 0 && (!u || !out))) {
    nb_set_error("bad prior transform arguments");
    return NB_ERR_ARG;
  }
  for (int j = 0; j < n_dim; ++j)
    if (kind[j] > 1) {
      nb_set_error("prior kind %d of parameter %d is not supported", kind[j], j);
      return NB_ERR_UNSUPPORTED;
    }
  return nb_launch_prior(u, n, n_dim, kind, loc, scale, out, as_stream(stream));
}

int nb_loglike_rosenbrock(const double* u, int64_t n, int32_t n_dim, double lo,
                          double hi, double a, double* out, void* stream) {
  if (n_dim < 2 || (n > 0 && (u == nullptr || out == nullptr))) {
    nb_set_error("bad Rosenbrock likelihood arguments");
    return NB_ERR_ARG;
  }
  return nb_launch_rosenbrock(u, n, n_dim, lo, hi, a, out, as_stream(stream));
}

int nb_loglike_funnel(const double* u, int64_t n, int32_t n_dim, double mu,
                      double sigma0, double k, double c, double* out,
                      void* stream) {
  if (n_dim < 2 || !(sigma0 > 0.0) || !(c > 0.0) ||
      (n > 0 && (u == nullptr || out == nullptr))) {
    nb_set_error("bad funnel likelihood arguments");
    return NB_ERR_ARG;
  }
  return nb_launch_funnel(u, n, n_dim, mu, sigma0, k, c, out,
                          as_stream(stream));
}

int nb_live_append(const double* log_l, int64_t n, const double* thr,
                   double* pool, int32_t* pool_n, int32_t capacity,
                   int32_t* overflow, void* stream) {
  if (capacity < 1 || thr == nullptr || pool == nullptr || pool_n == nullptr ||
      overflow == nullptr || (n > 0 && log_l == nullptr)) {
    nb_set_error("bad live-pool arguments");
    return NB_ERR_ARG;
  }
  return nb_launch_live_append(log_l, n, thr, pool, pool_n, capacity, overflow,
                               as_stream(stream));
}

int nb_live_select(const double* pool, const int32_t* pool_n, int32_t capacity,
                   int32_t k, double* pool_out, int32_t* pool_out_n,
                   double* thr, double* stats, void* stream) {
  if (capacity < 1 || k < 1 || pool == nullptr || pool_n == nullptr ||
      pool_out == nullptr || pool_out_n == nullptr || thr == nullptr ||
      stats == nullptr) {
    nb_set_error("bad live-pool arguments");
    return NB_ERR_ARG;
  }
  return nb_launch_live_select(pool, pool_n, capacity, k, pool_out, pool_out_n,
                               thr, stats, as_stream(stream));
}

int nb_live_stats(const double* log_l, int64_t n, const double* thr,
                  double* out, void* stream) {
  if (thr == nullptr || out == nullptr || (n > 0 && log_l == nullptr)) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_live_stats(log_l, n, thr, out, as_stream(stream));
}

int64_t nb_gmm_out_doubles(int32_t n_dim) {
  return nb_gmm_out_stride_impl(n_dim);
}

int64_t nb_gmm_scratch_doubles(int64_t n, int32_t n_dim) {
  return nb_gmm_scratch_stride_impl(n, n_dim);
}

int64_t nb_gmm_work_doubles(int64_t n, int32_t n_dim, int32_t n_init) {
  return nb_gmm_work_doubles_impl(n, n_dim, n_init);
}

int nb_gmm_set_max_wgs(int32_t max_wgs) { return nb_gmm_set_cap_impl(max_wgs); }
int64_t nb_gmm_logp_offset(int32_t n_dim) {
  return nb_gmm_logp_offset_impl(2, n_dim);     // (does not depend on n)
}

int nb_gmm_fit(const double* x, int64_t n, int32_t n_dim, int32_t n_init,
               uint64_t seed, double tol, double reg_covar, int32_t max_iter,
               const int32_t* init_labels, double* out, double* scratch,
               void* stream) {
  if (x == nullptr || out == nullptr || scratch == nullptr) {
    nb_set_error("null argument");
    return NB_ERR_ARG;
  }
  return nb_launch_gmm(x, n, n_dim, n_init, seed, tol, reg_covar, max_iter,
                       init_labels, out, scratch, as_stream(stream));
}

int nb_phase_shift(double* x, int64_t n, int32_t n_dim, int32_t n_periodic,
                   const int32_t* periodic, const double* centers,
                   int32_t inverse, void* stream) {
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT || n_periodic < 0 ||
      (n_periodic > 0 && (periodic == nullptr || centers == nullptr))) {
    nb_set_error("bad phase shift arguments");
    return NB_ERR_ARG;
  }
  double s[16 * NB_MAX_DT] = {0.0};
  unsigned char on[16 * NB_MAX_DT] = {0};
  for (int i = 0; i < n_periodic; ++i) {
    if (periodic[i] < 0 || periodic[i] >= n_dim) {
      nb_set_error("periodic index %d out of range", periodic[i]);
      return NB_ERR_ARG;
    }
    s[periodic[i]] = -centers[i] + 0.5;
    on[periodic[i]] = 1;
  }
  if (n <= 0 || n_periodic == 0) return NB_OK;
  return nb_launch_phase_shift(x, n, n_dim, s, on, inverse, as_stream(stream));
}

int nb_ellipsoid_contains_stream(const nb_bound* b, const double* x, int64_t n,
                                 uint8_t* mask, void* stream) {
  if (!b->single_full_ellipsoid) {
    nb_set_error("nb_ellipsoid_contains_stream needs a single full ellipsoid");
    return NB_ERR_ARG;
  }
  const double* cvec = b->blob_dev + b->off_stream;
  return nb_launch_ell_stream(cvec, cvec + 16 * b->dt, b->n_dim, x, n, mask,
                              as_stream(stream));
}

}  // extern "C"
