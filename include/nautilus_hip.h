/* nautilus_hip.h -- C ABI of the MI355X (gfx950) shell-filling hot path.
 *
 * The reference (johannesulf/nautilus v1.0.6) is pure Python and has no FFI;
 * its seam for this path is the duck-typed bound protocol
 *     Class.compute / .contains(points) / .sample(n) / .log_v / .reset(rng)
 * used by nautilus/sampler.py:791-798, 932, 1002, 1023-1035, 1069, 1218 and
 * the emulator protocol NeuralNetworkEmulator.train / .predict
 * (nautilus/neural.py:50, 100).  Every entry point below states the reference
 * call it replaces (paths relative to /root/reference).  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (NB_OK) on success, nonzero otherwise;
 *     nb_last_error() gives the message; nothing throws across the boundary;
 *   - "dev" pointers are device (HBM) pointers, "host" pointers are host
 *     memory; points are float64, C-contiguous, row-major (n, n_dim) exactly
 *     like the reference's numpy arrays;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued asynchronously on it, the caller synchronises;
 *   - n_dim <= 128; the emulator architecture is the reference default
 *     (100, 50, 20) ReLU MLP (nautilus/neural.py:79-81).
 */
#ifndef NAUTILUS_HIP_H
#define NAUTILUS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NB_ABI_VERSION 6

typedef struct nb_bound nb_bound;          /* opaque bound living in HBM     */
typedef struct nb_boundlist nb_boundlist;  /* device array of bound pointers */

/* One member of a union: an ellipsoid over the dims `idx_ell`, the unit
 * interval [0,1) over all other dims.
 *   n_ell == n_dim -> Ellipsoid                  (bounds/basic.py:244)
 *   0 < n_ell < n_dim -> UnitCubeEllipsoidMixture (bounds/basic.py:452)
 *   n_ell == 0 -> UnitCube                        (bounds/basic.py:9)       */
typedef struct {
  int32_t n_ell;
  const int32_t* idx_ell; /* [n_ell] increasing; NULL means 0..n_ell-1       */
  const double* c;        /* [n_ell]        Ellipsoid.c                      */
  const double* B;        /* [n_ell*n_ell]  Ellipsoid.B, lower triangular    */
  const double* B_inv;    /* [n_ell*n_ell]  Ellipsoid.B_inv, lower triangular*/
  int32_t free_dims;      /* 1: the non-ellipsoid dims are unbounded (a plain
                             lower-dimensional Ellipsoid), 0: they are [0,1) */
} nb_member_desc;

/* sklearn MLPRegressor weights of the emulator (neural.py:139-143 layout).  */
typedef struct {
  int32_t n_networks;
  const double* mean;              /* [n_dim] NeuralNetworkEmulator.mean     */
  const double* scale;             /* [n_dim] NeuralNetworkEmulator.scale    */
  const double* const* coefs;      /* [n_networks*4] (fan_in, fan_out) arrays*/
  const double* const* intercepts; /* [n_networks*4]                         */
} nb_mlp_desc;

/* NeuralBound (bounds/neural.py:10): ellipsoid AND emulator threshold.      */
typedef struct {
  nb_member_desc ellipsoid;     /* NeuralBound.outer_bound (n_ell == n_dim)  */
  const nb_mlp_desc* mlp;       /* NULL when n_networks == 0                 */
  double score_predict_min;     /* NeuralBound.score_predict_min             */
  double radius2;               /* >= largest squared semi-axis of the
                                   ellipsoid (sigma_max(B)^2), used to skip
                                   bounds whose ellipsoids cannot contain any
                                   point of a workgroup; <= 0: derive a
                                   conservative value from B                 */
} nb_neural_desc;

/* Any bound of the reference as one description:
 *   UnitCube        n_members=1 (n_ell=0), unit_cube=1, n_neural=0
 *   Ellipsoid / Mixture   n_members=1, unit_cube=0
 *   Union           n_members=K, unit_cube = (Union.cube is not None)
 *   NeuralBound     n_members=0, n_neural=1
 *   NautilusBound   outer_bound -> members, neural_bounds -> neural         */
typedef struct {
  int32_t n_dim;
  int32_t n_members;
  const nb_member_desc* members;
  const double* log_v_all;      /* [n_members] Union.log_v_all               */
  int32_t unit_cube;
  int32_t n_neural;
  const nb_neural_desc* neural;
  /* NautilusBound.shift (bounds/periodic.py:6-19, nautilus.py:91-96): the
   * periodic dimensions and their centres; contains() recentres the points
   * first (nautilus.py:162-163).  n_periodic = 0: no shift.                 */
  int32_t n_periodic;
  const int32_t* periodic;      /* [n_periodic] PhaseShift.periodic          */
  const double* centers;        /* [n_periodic] PhaseShift.centers           */
} nb_bound_desc;

int nb_abi_version(void);
const char* nb_last_error(void);

/* Upload a bound (host description -> packed HBM blob).  Replaces the
 * in-memory state built by *.compute (basic.py:265-316, union.py:78-151,
 * bounds/neural.py:58-97, nautilus.py:88-144) once the host logic of the
 * construction has decided the shape; the numerical work of the construction
 * is further down: nb_mvee_*, nb_gmm_fit, nb_trainer_* (SURVEY.md section 8
 * rows f1/f2).                                                              */
int nb_bound_create(const nb_bound_desc* desc, nb_bound** out);
int nb_bound_destroy(nb_bound* bound);
int64_t nb_bound_nbytes(const nb_bound* bound);

/* Device array of bounds for multi-bound queries.                           */
int nb_boundlist_create(nb_bound* const* bounds, int32_t n, nb_boundlist** out);
int nb_boundlist_destroy(nb_boundlist* list);

/* contains(points) of any bound: basic.py:51-67, 344-360, 594-617,
 * union.py:269-289, bounds/neural.py:99-126, nautilus.py:146-169.
 * mask_dev[i] = 1 if point i is inside.                                     */
int nb_contains(const nb_bound* bound, const double* x_dev, int64_t n,
                uint8_t* mask_dev, void* stream);

/* Shell exclusion (sampler.py:796-798): mask_dev[i] = 1 if ANY bound of the
 * list contains point i (bounds are tested in list order, a tile of points
 * stops as soon as all of its points are decided).                          */
int nb_contains_any(const nb_boundlist* list, const double* x_dev, int64_t n,
                    uint8_t* mask_dev, void* stream);

/* Shell association (sampler.py:1192-1221): idx_dev[i] = position in the list
 * of the FIRST bound that contains point i, -1 if none (pass the bounds from
 * the highest index down to get the reference's association).               */
int nb_first_containing(const nb_boundlist* list, const double* x_dev,
                        int64_t n, int32_t* idx_dev, void* stream);

/* Overlap count k_i = #members containing x_i (union.py:316-317).           */
int nb_member_count(const nb_bound* bound, const double* x_dev, int64_t n,
                    uint8_t* count_dev, void* stream);

/* Ellipsoid-frame radius and emulator score of neural bound 0 of `bound`:
 * out_dev[2i] = |B_inv (x_i - c)|^2 (basic.py:340,360), out_dev[2i+1] =
 * NeuralNetworkEmulator.predict(transform(x_i)) (neural.py:100-116).        */
int nb_neural_score(const nb_bound* bound, const double* x_dev, int64_t n,
                    double* out_dev, void* stream);

/* Raw proposals of Union.sample / UnitCube.sample / Ellipsoid.sample
 * (union.py:305-312, basic.py:85, 376-381, 633-640) for the global proposal
 * indices offset .. offset+n-1 of Philox stream `seed` (DESIGN.md "RNG
 * contract"): member by inverse CDF on softmax(log_v_all), uniform-in-
 * ellipsoid map, uniform cube columns.  x_dev receives n rows.              */
int nb_propose(const nb_bound* bound, uint64_t seed, uint64_t offset,
               int64_t n, double* x_dev, void* stream);

/* Acceptance of the proposals written by nb_propose with the same (seed,
 * offset): flags_dev[i] bit0 = kept by the outer union (unit-cube clip,
 * union.py:313-314, and u > 1 - 1/k, union.py:316-319), bit1 = also inside
 * any NeuralBound (nautilus.py:217-219).                                    */
int nb_accept(const nb_bound* bound, uint64_t seed, uint64_t offset,
              const double* x_dev, int64_t n, uint8_t* flags_dev,
              void* stream);

/* Stable stream compaction (the reference's boolean indexing
 * `points[in_bound]`, `points[in_shell]`): rows with ((flags ^ flip) & mask)
 * != 0 are copied to out_dev in input order (flip = 1, mask = 1 keeps the
 * rows whose flag is 0: the points NOT inside a later bound,
 * sampler.py:796-799).  counts_dev[0] = rows with bit0 of flags ^ flip set,
 * counts_dev[1] = rows copied.  src_idx_dev (optional) receives the source
 * row of every output row -- src_idx_dev[k-1] + 1 rows were examined for
 * the first k survivors, the "trials until the k-th success" of
 * Sampler.sample_shell.  scratch_dev must hold nb_compact_scratch_bytes(n)
 * bytes.                                                                    */
int64_t nb_compact_scratch_bytes(int64_t n);
int nb_compact_rows(const double* x_dev, const uint8_t* flags_dev,
                    uint8_t mask, uint8_t flip, int64_t n, int32_t n_dim,
                    double* out_dev, int64_t* src_idx_dev, int64_t* counts_dev,
                    void* scratch_dev, void* stream);

/* Per-shell evidence statistics (sampler.py:927-943): out_dev[0] =
 * logsumexp(log_l), out_dev[1] = logsumexp(2 log_l), out_dev[2] = max,
 * out_dev[3] = number of elements >= threshold (sampler.py:1144).           */
int nb_shell_stats(const double* log_l_dev, int64_t n, double threshold,
                   double* out_dev, void* scratch_dev, void* stream);
int64_t nb_shell_stats_scratch_bytes(int64_t n);

/* The live set of the exploration phase on the device (Sampler.f_live /
 * log_v_live / the threshold of add_bound, sampler.py:1147-1190, 1004-1010:
 * the reference sorts every stored log L on every iteration).  A pool in HBM
 * holds every log L at or above the current threshold (thr_dev[0], start at
 * -inf):
 *   nb_live_append   appends the values >= threshold of a new batch
 *                    (pool_n_dev[0] entries so far; overflow_dev[0] is set if
 *                    the capacity does not suffice -- rebuild with a larger
 *                    pool);
 *   nb_live_select   the exact k-th largest value of the pool (radix
 *                    selection, one workgroup) becomes the new threshold, the
 *                    values >= it are copied to pool_out (pool_out_n_dev[0]
 *                    entries); stats_dev[0..2] = threshold, #values above,
 *                    #values equal (ties at the threshold);
 *   nb_live_stats    for one shell's log L: out_dev[0] = #(l > threshold),
 *                    out_dev[1] = logsumexp(l | l > threshold), out_dev[2] =
 *                    #(l == threshold) -- the shell's share of the live
 *                    weight (shuffle reductions).                            */
int nb_live_append(const double* log_l_dev, int64_t n, const double* thr_dev,
                   double* pool_dev, int32_t* pool_n_dev, int32_t capacity,
                   int32_t* overflow_dev, void* stream);
int nb_live_select(const double* pool_dev, const int32_t* pool_n_dev,
                   int32_t capacity, int32_t k, double* pool_out_dev,
                   int32_t* pool_out_n_dev, double* thr_dev, double* stats_dev,
                   void* stream);
int nb_live_stats(const double* log_l_dev, int64_t n, const double* thr_dev,
                  double* out_dev, void* stream);

/* Emulator training, NeuralNetworkEmulator.train -> MLPRegressor.fit
 * (neural.py:50-98; sklearn/_multilayer_perceptron.py:620-760): Adam,
 * minibatch 200, squared loss, stop after 11 stale epochs.  A resident kernel
 * trains up to 16 networks at a time, every network on the 32 CUs of one XCD
 * (two networks per XCD beyond 8).  See nb_mlp_train.hip for the state
 * layout.                                                                    */
typedef struct nb_trainer nb_trainer;
int nb_trainer_create(int32_t n_dim, int32_t n_networks, int64_t n_rows,
                      const double* x_dev, const double* y_dev,
                      const double* const* coefs_host,
                      const double* const* intercepts_host,
                      nb_trainer** out);
/* A fleet: the networks of SEVERAL ensembles -- the neural bounds of a
 * multi-modal NautilusBound, nautilus.py:107-114 -- in one trainer, network i
 * with the training set (x_dev_of[i], y_dev_of[i], n_rows_of[i]) of its
 * ensemble; all of them train in the same resident launches.               */
int nb_trainer_create_fleet(int32_t n_dim, int32_t n_networks,
                            const int64_t* n_rows_of,
                            const double* const* x_dev_of,
                            const double* const* y_dev_of,
                            const double* const* coefs_host,
                            const double* const* intercepts_host,
                            nb_trainer** out);
/* Optional: MLPRegressor hyper-parameters (defaults are the reference's,
 * neural.py:79-81: lr 1e-2, betas 0.9/0.999, eps 1e-8, batch 200, max_iter
 * 10000, n_iter_no_change 10, tol 0).                                       */
int nb_trainer_set_hparams(nb_trainer* t, double lr, double beta1,
                           double beta2, double epsilon, int32_t batch,
                           int32_t max_iter, int32_t n_iter_no_change,
                           double tol);
/* Run up to n_epochs epochs; perm_dev holds n_networks*n_epochs*n_rows int32
 * row orders (network-major) and must stay alive until the work has finished.
 * status_host[e] receives n_iter so far, or -n_iter when network e has
 * stopped; with status_host == NULL the call only enqueues (asynchronous).   */
int nb_trainer_run(nb_trainer* t, const int32_t* perm_dev, int32_t n_epochs,
                   int32_t* status_host, void* stream);
/* ... of a fleet: perm_dev_of[i] = n_epochs * n_rows_of[i] row orders of
 * network i.                                                                 */
int nb_trainer_run_fleet(nb_trainer* t, const int32_t* const* perm_dev_of,
                         int32_t n_epochs, int32_t* status_host, void* stream);
/* Wait for the launches enqueued by nb_trainer_run(..., status_host = NULL)
 * and read the per-network status (same encoding).                          */
int nb_trainer_status(nb_trainer* t, int32_t* status_host, void* stream);
/* Pipelined form of the two above: enqueue a chunk of epochs (a fleet's
 * arguments; a plain trainer passes the same training set n_networks times)
 * together with an asynchronous copy of its status words to pinned host
 * memory, and collect the status of ticket k later -- while chunk k + 1 is
 * already queued behind it, so the GPU never waits for the host between
 * chunks.  A network that stops inside chunk k skips chunk k + 1 on the
 * device.  At most 4 tickets may be outstanding.                             */
int nb_trainer_run_async(nb_trainer* t, const int32_t* const* perm_dev_of,
                         int32_t n_epochs, void* stream, int64_t* ticket);
int nb_trainer_wait(nb_trainer* t, int64_t ticket, int32_t* status_host);
/* Host helper (no device work): the minibatch orders scikit-learn draws
 * before every epoch -- sklearn.utils.shuffle = numpy's legacy
 * RandomState.shuffle on MT19937, composed epoch after epoch
 * (_multilayer_perceptron.py:700-704; reached from neural.py:79-98) -- for
 * n_streams independent generator states (key_of[s]: the 624 words and pos[s]
 * the position of RandomState.get_state(), both advanced in place), one host
 * thread per stream.  order_of[s] (n_of[s] entries, in/out) is the current
 * order, out_of[s] receives n_epochs x n_of[s] orders; streams with a NULL
 * out_of[s] are skipped.                                                     */
int nb_host_shuffle_epochs(int32_t n_streams, uint32_t* const* key_of,
                           int32_t* pos, const int64_t* n_of, int32_t n_epochs,
                           int32_t* const* order_of, int32_t* const* out_of);
int nb_trainer_loss_curve(nb_trainer* t, int32_t net, double* out_host,
                          int32_t max_len);
int nb_trainer_weights(nb_trainer* t, int32_t net, double* const* coefs_host,
                       double* const* intercepts_host);
int nb_trainer_destroy(nb_trainer* t);

/* Philox helper for tests: u_dev[2i], u_dev[2i+1] = the two uniforms of
 * (seed, offset+i, block, tag).                                             */
int nb_philox_uniform(uint64_t seed, uint64_t offset, uint32_t block,
                      uint32_t tag, int64_t n, double* u_dev, void* stream);

/* Measurement hook for bench.py: when non-NULL, every bound-evaluation launch
 * adds to counters_dev[0..2] the number of point evaluations it performed
 * (outer-member tests, neural-ellipsoid transforms, emulator forward passes x
 * networks) -- the algorithmic work of the roofline.  */
int nb_set_eval_counters(uint64_t* counters_dev);

/* fp64 MFMA issue-rate microbenchmark (peak calibration for bench.py):
 * returns achieved TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64.          */
int nb_mfma_f64_peak(int32_t iters, double* tflops_host);

/* The iteration of minimum_volume_enclosing_ellipsoid (bounds/basic.py:
 * 175-232; called by Ellipsoid.compute :295 and, through it, by
 * UnitCubeEllipsoidMixture.compute and Union.compute/split): n_max sweeps of
 * up to n_batch Khachiyan updates over the n points x_dev (n > n_dim,
 * n_dim <= 128, n_batch <= 32).  u_dev[n] receives the weights u of
 * basic.py:231.  The points are whitened internally (the iteration is
 * affine invariant) and the work is spread over up to 32 workgroups.
 * work_dev: nb_mvee_weights_work_doubles(n, n_dim, n_batch) doubles.  The
 * reference's defaults are n_max = 100, n_batch = 20.                       */
int64_t nb_mvee_weights_work_doubles(int64_t n, int32_t n_dim,
                                     int32_t n_batch);
int nb_mvee_weights(const double* x_dev, int64_t n, int32_t n_dim,
                    int32_t n_max, int32_t n_batch, double* u_dev,
                    double* work_dev, void* stream);

/* Whitening of a point set: xw_dev[n][n_dim] = W ((x - mean) / sd) has zero
 * mean and unit covariance (mean_dev, sd_dev: [n_dim]; w_dev: [n_dim^2], lower
 * triangular, row-major; per-column statistics, second moments on the matrix
 * cores, L D L^T of the correlation matrix in LDS, triangular product on the
 * matrix cores).  The Khachiyan iteration is invariant under affine maps;
 * on whitened points its matrices stay well conditioned however correlated
 * or badly scaled the input is.  work_dev: nb_whiten_work_doubles(n, n_dim).  */
int64_t nb_whiten_work_doubles(int64_t n, int32_t n_dim);
int nb_whiten(const double* x_dev, int64_t n, int32_t n_dim, double* xw_dev,
              double* mean_dev, double* sd_dev, double* w_dev,
              double* work_dev, void* stream);

/* The same iteration for a batch of independent point sets of one dimension
 * (the two children of Union.split, union.py:198-202; the neural-bound
 * ellipsoids of one NautilusBound, nautilus.py:107-114), advanced side by
 * side by the same launches.  xs_dev[b] are WHITENED points (nb_whiten);
 * u_dev[b] receives the weights of set b.  work_dev: nb_mvee_work_doubles(n_problems, max n,
 * n_dim, n_batch) doubles.                                                   */
int64_t nb_mvee_work_doubles(int32_t n_problems, int64_t n_points_max,
                             int32_t n_dim, int32_t n_batch);
int nb_mvee_khachiyan(int32_t n_problems, const double* const* xs_dev,
                      const int64_t* n_points, int32_t n_dim, int32_t n_max,
                      int32_t n_batch, double* const* u_dev, double* work_dev,
                      void* stream);

/* Weighted second moments of the augmented rows q_i = (x_i, 1):
 * out_dev[(n_dim+1)^2] = scale * sum_i w_i q_i q_i^T (row-major, symmetric;
 * w_dev == NULL: unit weights).  With the MVEE weights this is the centre and
 * covariance of basic.py:233-234 in one pass on the matrix cores (the last
 * row holds sum w x, the corner sum w).  work_dev:
 * nb_moments_work_doubles(n, n_dim) doubles.                                 */
int64_t nb_moments_work_doubles(int64_t n, int32_t n_dim);
int nb_weighted_moments(const double* x_dev, const double* w_dev, int64_t n,
                        int32_t n_dim, double scale, double* out_dev,
                        double* work_dev, void* stream);

/* out_dev[0] = max_i q_i^T P q_i, q_i = (x_i, 1), for a symmetric
 * (n_dim+1)^2 matrix p_dev -- the scaling step of the MVEE (basic.py:236:
 * with P the inverse of the second-moment matrix the largest squared
 * Mahalanobis distance is out - 1).  work_dev: nb_quadform_work_doubles().   */
int64_t nb_quadform_work_doubles(void);
int nb_quadform_max(const double* x_dev, int64_t n, int32_t n_dim,
                    const double* p_dev, double* out_dev, double* work_dev,
                    void* stream);

/* Ellipsoid.transform (bounds/basic.py:318-342, forward direction):
 * y_dev[i] = B_inv (x_i - c) for an Ellipsoid bound (or the ellipsoid of a
 * NeuralBound) -- the emulator's training inputs (bounds/neural.py:90).      */
int nb_ellipsoid_transform(const nb_bound* bound, const double* x_dev,
                           int64_t n, double* y_dev, void* stream);

/* Input standardisation of NeuralNetworkEmulator.train (neural.py:74-77):
 * mean_dev[j] = mean(x[:, j]), scale_dev[j] = std(x[:, j]) and, when out_dev
 * is not NULL, out_dev = (x - mean) / scale.                                 */
int nb_standardize(const double* x_dev, int64_t n, int32_t n_dim,
                   double* mean_dev, double* scale_dev, double* out_dev,
                   void* stream);

/* Prior.unit_to_physical (prior.py:85-120), x = dist.isf(1 - u), for
 * parameters with uniform (kind 0) or normal (kind 1) distributions;
 * kind / loc / scale are host arrays of length n_dim (scipy's loc / scale).
 * Keeps the physical points of a device likelihood on the GPU.               */
int nb_prior_transform(const double* u_dev, int64_t n, int32_t n_dim,
                       const uint8_t* kind, const double* loc,
                       const double* scale, double* out_dev, void* stream);

/* Device likelihoods of the benchmark problems (the user-side callable of
 * sampler.py:863-873 for the BASELINE configurations C3 and C5; Gaussians go
 * through nb_neural_score): out_dev[i] = log L of row i of u_dev (unit-cube
 * points, n x n_dim).
 *   Rosenbrock: x = lo + (hi - lo) u,
 *               log L = -sum_i [a (x_{i+1} - x_i^2)^2 + (1 - x_i)^2]
 *   Neal funnel (tests/test_sampler.py:311-314 in n_dim dimensions):
 *               x_0 ~ N(mu, sigma0^2), x_i ~ N(mu, (exp(k (x_0 - mu)) / c)^2) */
int nb_loglike_rosenbrock(const double* u_dev, int64_t n, int32_t n_dim,
                          double lo, double hi, double a, double* out_dev,
                          void* stream);
int nb_loglike_funnel(const double* u_dev, int64_t n, int32_t n_dim, double mu,
                      double sigma0, double k, double c, double* out_dev,
                      void* stream);


/* Two-stage evaluation of bounds with several outer members, several neural
 * bounds, or of lists of bounds (bounds/union.py:285-289, 316-319;
 * bounds/nautilus.py:162-169, 212-222; sampler.py:797-798, 1213-1219),
 * entirely on the device -- no host round trip between the stages:
 *
 * Stage 1 (nb_cand.hip): everything that needs no emulator -- periodic
 * recentring, unit-cube clip, overlap count of the outer members, acceptance
 * draw, the ellipsoids of the neural bounds -- for every point against every
 * bound of the list; a point becomes a CANDIDATE of every (bound, neural
 * bound) whose ellipsoid contains it (the reference's contains / sample are
 * disjunctions over the neural bounds, so nothing waits for anything).
 * Stage 2 (nb_eval_fast.hip): the emulators of ALL groups on their candidate
 * rows in ONE launch of the pipelined kernel (dense 128-point passes, row
 * counts read from device memory); a candidate whose score exceeds
 * score_predict_min - 1e-9 (bounds/neural.py:125) sets its row's result.
 *
 * nb_list_eval, mode 0 (shell exclusion, sampler.py:797-798): st_dev[i] = 2
 * if any bound of the list contains point i, else 0.  mode 1 (shell
 * association, sampler.py:1213-1219): additionally first_dev[i] = position
 * of the first bound that contains point i, INT32_MAX if none.  Points are in
 * the sampler's frame (periodic bounds recentre them, nautilus.py:162-163).
 * nb_accept_staged: the flags of nb_accept (bit 0: kept by the outer union's
 * acceptance draw, bit 1: accepted) for proposals x_dev of stream position
 * `offset`, for any number of outer members and neural bounds.  If
 * totals_offset is not NULL it receives the byte offset inside work_dev of
 * the int32 candidate counts per neural bound (valid in stream order).
 * work_dev: device scratch of at least nb_*_work_bytes(.., n) bytes.        */
int64_t nb_list_eval_work_bytes(const nb_boundlist* list, int64_t n);
int nb_list_eval(const nb_boundlist* list, int32_t mode, const double* x_dev,
                 int64_t n, uint8_t* st_dev, int32_t* first_dev, void* work_dev,
                 int64_t work_bytes, void* stream);
int64_t nb_accept_staged_work_bytes(const nb_bound* bound, int64_t n);
int nb_accept_staged(const nb_bound* bound, uint64_t seed, uint64_t offset,
                     const double* x_dev, int64_t n, uint8_t* flags_dev,
                     void* work_dev, int64_t work_bytes, int64_t* totals_offset,
                     void* stream);
/* (r2, score) of neural bound m of `bound` for the rows idx_dev[0..n) of x,
 * densely packed (out_dev[2 i], [2 i + 1]) -- the pipelined emulator kernel
 * on full tiles.  recentre != 0: the rows are in the sampler's frame and the
 * bound's periodic shift is applied first, as contains() does
 * (nautilus.py:162-163); 0 for proposals, which live in the bound's frame.  */
int nb_neural_score_rows(const nb_bound* bound, int32_t m, int32_t recentre,
                         const double* x_dev, const int64_t* idx_dev,
                         int64_t n, double* out_dev, void* stream);

/* The mixture fit of Union.split (bounds/union.py:185-187): scikit-learn's
 * GaussianMixture(n_components=2, n_init=n_init, covariance_type='full')
 * restated on the device -- k-means++ / Lloyd initialisation, EM until the
 * mean log-likelihood changes by less than tol (scikit-learn defaults: tol
 * 1e-3, reg_covar 1e-6, max_iter 100).  All restarts run concurrently, each
 * on up to eight workgroups that share its rows.
 * out_dev: n_init records of nb_gmm_out_doubles(n_dim) doubles
 *   [lower_bound, n_iter, converged, failed, weight0, weight1,
 *    mean0[D], mean1[D], cov0[D*D], cov1[D*D]]
 * (the caller keeps the record with the largest lower bound, as
 * mixture/_base.py:fit_predict does).  scratch_dev:
 * nb_gmm_work_doubles(n, n_dim, n_init) doubles; restart r works in the
 * nb_gmm_scratch_doubles(n, n_dim) doubles from r * that many on (the rest
 * holds the restarts' barrier counters).  init_labels_dev (optional,
 * [n_init][n] int32 in {0,1}) replaces the k-means initialisation.
 * n_dim <= 128.  After the call the scratch of restart r holds, from double
 * nb_gmm_logp_offset(n_dim) on, log(w_k N(x_i; mu_k, Sigma_k)) of all points
 * under the returned parameters, [n] for component 0 then [n] for component 1
 * -- the hard assignment of union.py:188-197 without a second pass over the
 * points on the host.                                                         */
int64_t nb_gmm_out_doubles(int32_t n_dim);
int64_t nb_gmm_scratch_doubles(int64_t n, int32_t n_dim);
int64_t nb_gmm_work_doubles(int64_t n, int32_t n_dim, int32_t n_init);
int64_t nb_gmm_logp_offset(int32_t n_dim);
int nb_gmm_fit(const double* x_dev, int64_t n, int32_t n_dim, int32_t n_init,
               uint64_t seed, double tol, double reg_covar, int32_t max_iter,
               const int32_t* init_labels_dev, double* out_dev,
               double* scratch_dev, void* stream);
/* The workgroups of a restart wait for each other inside an ordinary launch:
 * they must all be resident.  The library limits a fit to half of the
 * device's CUs on its own; a host whose processes SHARE a device (several
 * ranks on one GPU) sets max_wgs = 1 (one workgroup per restart, nothing
 * waits); 0 = back to the default (8, or the NB_GMM_MAX_WGS variable).     */
int nb_gmm_set_max_wgs(int32_t max_wgs);

/* PhaseShift.transform (bounds/periodic.py:50-72), in place on device rows:
 * x[:, periodic[i]] = (x[:, periodic[i]] -/+ (0.5 - centers[i])) mod 1
 * (inverse != 0 undoes the shift, nautilus.py:241-243).  `periodic` and
 * `centers` are host arrays.                                                */
int nb_phase_shift(double* x_dev, int64_t n, int32_t n_dim, int32_t n_periodic,
                   const int32_t* periodic, const double* centers,
                   int32_t inverse, void* stream);

/* Multi-GPU exchange over RCCL / xGMI for a binding that does not bring its
 * own collectives (the product's Python layer uses torch.distributed, whose
 * "nccl" backend is the same RCCL): the reference's parallel pattern on this
 * path -- replicate the bound, draw independent streams, concatenate the
 * accepted points, add the counters (bounds/nautilus.py:223-237) -- is one
 * all-gather of equal per-rank blocks of doubles and one all-reduce of
 * int64 counters per batch.  Rank 0 creates the 128-byte id and hands it to
 * the other processes out of band (one process per GPU); nb_comm_rank_key
 * gives rank r its own Philox key.  RCCL is loaded at run time.            */
#define NB_COMM_ID_BYTES 128
typedef struct nb_comm nb_comm;
int nb_comm_unique_id(uint8_t* id_out /* [NB_COMM_ID_BYTES] */);
int nb_comm_init(int32_t rank, int32_t n_ranks, const uint8_t* id,
                 nb_comm** out);
int nb_comm_destroy(nb_comm* comm);
uint64_t nb_comm_rank_key(uint64_t seed, int32_t rank);
int nb_comm_allgather_f64(nb_comm* comm, const double* send_dev, int64_t count,
                          double* recv_dev /* [n_ranks * count] */,
                          void* stream);
int nb_comm_allreduce_i64(nb_comm* comm, int64_t* buf_dev /* in place, sum */,
                          int64_t count, void* stream);

/* The roofline kernel: single-ellipsoid contains (basic.py:344-360) with the
 * points streamed once from HBM.  Same result as nb_contains.               */
int nb_ellipsoid_contains_stream(const nb_bound* bound, const double* x_dev,
                                 int64_t n, uint8_t* mask_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NAUTILUS_HIP_H */
