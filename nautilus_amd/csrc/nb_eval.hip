// Bound evaluation on the matrix cores: contains() of UnitCube / Ellipsoid /
// UnitCubeEllipsoidMixture / Union / NeuralBound / NautilusBound, the overlap
// count + acceptance of Union.sample and the NeuralBound filter of
// NautilusBound.sample, for tiles of 16 points per wavefront.
//
// Reference semantics (file:line relative to /root/reference/nautilus):
//   bounds/basic.py:340,360   y = B_inv (x - c),  sum(y^2) < 1  (strict)
//   bounds/basic.py:67        unit cube: 0 <= x < 1
//   bounds/basic.py:610-617   mixture: cube columns AND ellipsoid columns
//   bounds/union.py:285-289   union: any member AND unit cube
//   bounds/union.py:316-319   k = #members containing x, keep if u > 1 - 1/k
//   bounds/neural.py:115-126  ellipsoid AND emulator(y) > score_min - 1e-9
//   neural.py:114-116         emulator = mean over nets of MLP((y-mean)/scale)
//   bounds/nautilus.py:162-169  outer union AND any neural bound
//   sampler.py:797-798        shell exclusion: any later bound contains
//   sampler.py:1213-1219      shell association: highest-index containing bound
//
// Mapping to v_mfma_f64_16x16x4_f64: rows i = output units, cols j = the 16
// points of the tile, K = input features.  Lane l holds for point (l & 15) the
// features 4*ks + (l >> 4), ks = 0,1,...  The C/D layout (col = l & 15,
// row = (l >> 4) + 4*reg) is exactly the B-operand layout of the next layer,
// so activations never leave the registers between layers.  Weights are read
// as A operands from 16x16 tile-major storage: one contiguous 512-byte row of
// the tile per MFMA.
#include "nb_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

struct EvalArgs {
  const double* const* blobs;   // device array of nb blobs
  int nb;
  int mode;
  const double* x;
  long long n;
  unsigned char* out_u8;
  int* out_i32;
  double* out_f64;
  unsigned long long seed;
  unsigned long long offset;
  unsigned long long* counters;   // optional: [0] outer-member point evals,
                                  // [1] neural-ellipsoid point evals,
                                  // [2] emulator point evals (x E networks)
};

enum { MODE_ANY = 0, MODE_ASSOC = 1, MODE_SAMPLE = 2, MODE_COUNT = 3,
       MODE_SCORE = 4 };

__device__ inline double lane_group_sum(double v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// true for every lane of a point if any of its four lanes has `flag`
__device__ inline bool point_any(bool flag, int lane) {
  const unsigned long long b = __ballot(flag);
  return ((b >> (lane & 15)) & 0x0001000100010001ull) != 0ull;
}

// y = B_inv (x - c) on the matrix cores plus the per-dimension box test.
// Returns r2 = |y|^2 (replicated over the 4 lanes of a point); box_bad is set
// if any coordinate violates the member's [lo, hi) limits.
template <int DT>
__device__ __forceinline__ double ell_eval(const double* __restrict__ blk, int n_dim,
                                  const double (&xin)[4 * DT], int lane,
                                  double (&y)[4 * DT], bool& box_bad,
                                  bool want_y) {
  constexpr int DP = 16 * DT;
  const double* lo = blk + 1;
  const double* hi = lo + DP;
  const double* c = hi + DP;
  const double* tiles = c + DP;
  const long long n_ell = ((const long long*)blk)[0];
  const int lg = lane >> 4;

  double d[4 * DT];
  bool bad = false;
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const int f = 4 * ks + lg;
    const double xv = xin[ks];
    bad |= !(xv >= lo[f] && xv < hi[f]);
    d[ks] = xv - c[f];
  }
  box_bad = point_any(bad, lane);

  double part = 0.0;
  if (n_ell > 0) {
#pragma unroll
    for (int ht = 0; ht < DT; ++ht) {
      if (16 * ht < n_dim) {
        nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4 * (ht + 1); ++ks) {   // lower-triangular
          const int kt = ks >> 2, s = ks & 3;
          const double a = tiles[(kt * DT + ht) * NB_TILE + s * 64 + lane];
          acc = MFMA(a, d[ks], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[4 * ht + r] = acc[r];
          part += acc[r] * acc[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) y[4 * ht + r] = 0.0;
      }
    }
  } else if (want_y) {
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks) y[ks] = 0.0;
  }
  return lane_group_sum(part);
}

// one dense layer on the matrix cores: out[h] = act(sum_k in[k] W[k][h]),
// bias folded in as row k = K (the input carries a constant 1 there).
template <int KSMAX, int HT, bool RELU>
__device__ __forceinline__ void mlp_layer(const double* __restrict__ w,
                                          int ks_n, const double* in, int lane,
                                          double* out) {
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
      if (ks < ks_n) {
        const int kt = ks >> 2, s = ks & 3;
        const double a = w[(kt * HT + ht) * NB_TILE + s * 64 + lane];
        acc = MFMA(a, in[ks], acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      out[4 * ht + r] = RELU ? fmax(acc[r], 0.0) : acc[r];
  }
}

// Emulator score for the 16 points of the tile; y is the ellipsoid-frame
// coordinate block.  Result valid in every lane (replicated per point).
template <int DT>
__device__ __forceinline__ double mlp_score(const double* __restrict__ nblk, int n_dim,
                                   int n_net, int kt1, long long net_stride,
                                   const double (&y)[4 * DT], int lane) {
  constexpr int DP = 16 * DT;
  constexpr int KS1MAX = 4 * DT + 1;
  const double* mean = nblk + nb_ell_block_size(DT) + 1;
  const double* scale = mean + DP;
  const double* nets = scale + DP;
  const int lg = lane >> 4;
  const int ks1 = (n_dim + 1 + 3) >> 2;

  double t[KS1MAX];
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const int f = 4 * ks + lg;
    t[ks] = (f < n_dim) ? (y[ks] - mean[f]) / scale[f]
                        : ((f == n_dim) ? 1.0 : 0.0);
  }
  t[4 * DT] = (4 * (4 * DT) + lg == n_dim) ? 1.0 : 0.0;

  double total = 0.0;
  for (int e = 0; e < n_net; ++e) {
    const double* w1 = nets + e * net_stride;
    const double* w2 = w1 + kt1 * NB_HT1 * NB_TILE;
    const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
    const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
    double h1[4 * NB_HT1], h2[4 * NB_HT2], h3[4 * NB_HT3], o[4];
    mlp_layer<KS1MAX, NB_HT1, true>(w1, ks1, t, lane, h1);
    if (lg == 0) h1[25] = 1.0;                       // bias unit 100
    mlp_layer<26, NB_HT2, true>(w2, 26, h1, lane, h2);
    if (lg == 2) h2[12] = 1.0;                       // bias unit 50
    mlp_layer<13, NB_HT3, true>(w3, 13, h2, lane, h3);
    if (lg == 0) h3[5] = 1.0;                        // bias unit 20
    mlp_layer<6, 1, false>(w4, 6, h3, lane, o);
    total += o[0];                                   // unit 0 lives in lg == 0
  }
  total = __shfl(total, lane & 15);                  // broadcast from lg == 0
  return total / (double)n_net;
}

template <int DT>
__global__ void __launch_bounds__(256)
nb_eval_kernel(EvalArgs a) {
  constexpr int DP = 16 * DT;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const long long n_tiles = (a.n + 15) >> 4;
  const long long stride = (long long)gridDim.x * 4;
  unsigned long long cnt_outer = 0, cnt_ell = 0, cnt_mlp = 0;

  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles;
       tile += stride) {
    const long long pt = tile * 16 + (lane & 15);
    const bool valid = pt < a.n;

    // geometry of the first blob gives n_dim (all blobs of a list agree)
    const double* blob0 = a.blobs[0];
    const int n_dim = (int)nb_hdr(blob0, NB_H_NDIM);

    double xin[4 * DT];
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks) {
      const int f = 4 * ks + lg;
      xin[ks] = (valid && f < n_dim) ? a.x[pt * n_dim + f] : 0.0;
    }

    bool hit = false;       // MODE_ANY / ASSOC: some bound of the list contains
    int hit_idx = -1;
    unsigned char flags = 0;
    int count_out = 0;
    double r2_out = 0.0, score_out = 0.0;

    for (int b = 0; b < a.nb; ++b) {
      const double* blob = a.blobs[b];
      const int K = (int)nb_hdr(blob, NB_H_K);
      const int M = (int)nb_hdr(blob, NB_H_M);
      const int E = (int)nb_hdr(blob, NB_H_E);
      const double* ulo = blob + nb_hdr(blob, NB_H_OFF_ULO);
      const double* uhi = blob + nb_hdr(blob, NB_H_OFF_UHI);
      const long long ell_stride = nb_hdr(blob, NB_H_ELL_STRIDE);
      const long long neural_stride = nb_hdr(blob, NB_H_NEURAL_STRIDE);

      // unit-cube clip of the union (union.py:287-288 / 313-314)
      bool cbad = false;
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        const int f = 4 * ks + lg;
        cbad |= !(xin[ks] >= ulo[f] && xin[ks] < uhi[f]);
      }
      const bool in_cube = !point_any(cbad, lane);

      const bool active = valid && !hit;

      // ---- outer union: overlap count --------------------------------
      int k_cnt = 0;
      const double* mblk = blob + nb_hdr(blob, NB_H_OFF_MEMBERS);
      if (a.mode == MODE_SAMPLE && K == 1) {
        k_cnt = 1;            // drawn from the only member
      } else {
        for (int m = 0; m < K; ++m) {
          double y[4 * DT];
          bool box_bad;
          const double r2 = ell_eval<DT>(mblk + m * ell_stride, n_dim, xin,
                                         lane, y, box_bad, false);
          k_cnt += (!box_bad && r2 < 1.0) ? 1 : 0;
        }
        cnt_outer += (unsigned long long)K *
                     __popcll(__ballot(active && lg == 0));
      }
      const bool outer_ok = in_cube && (K == 0 || k_cnt > 0);

      // acceptance of the overlap-corrected union draw (union.py:318-319)
      bool acc_outer = false;
      if (a.mode == MODE_SAMPLE) {
        double u0, u_acc;
        nb_uniform_pair(a.seed, a.offset + (unsigned long long)pt, 0u,
                        NB_TAG_CTRL, u0, u_acc);
        acc_outer = in_cube && (u_acc > 1.0 - 1.0 / (double)k_cnt);
      }

      // ---- neural bounds ----------------------------------------------
      bool neural_ok = (M == 0);
      bool want;
      if (a.mode == MODE_SAMPLE) want = valid && acc_outer;
      else if (a.mode == MODE_SCORE) want = valid;
      else if (a.mode == MODE_COUNT) want = false;
      else want = active && outer_ok;
      if (M > 0 && __any(want)) {
        const double* nblk = blob + nb_hdr(blob, NB_H_OFF_NEURAL);
        const int kt1 = (int)nb_hdr(blob, NB_H_KT1);
        const long long net_stride = nb_hdr(blob, NB_H_NET_STRIDE);
        for (int m = 0; m < M; ++m) {
          const double* nb_m = nblk + m * neural_stride;
          double y[4 * DT];
          bool box_bad;
          const double r2 = ell_eval<DT>(nb_m, n_dim, xin, lane, y, box_bad,
                                         true);
          const bool inside_e = !box_bad && r2 < 1.0;
          bool ok = inside_e;
          const bool need = want && inside_e && !neural_ok;
          cnt_ell += __popcll(__ballot(want && lg == 0));
          cnt_mlp += (unsigned long long)E *
                     __popcll(__ballot((need || (a.mode == MODE_SCORE && valid))
                                       && lg == 0));
          if (E > 0 && (a.mode == MODE_SCORE || __any(need))) {
            const double thr = nb_m[nb_ell_block_size(DT)];
            const double score = mlp_score<DT>(nb_m, n_dim, E, kt1,
                                               net_stride, y, lane);
            ok = inside_e && (score > thr);
            if (m == 0) score_out = score;
          }
          if (m == 0) r2_out = r2;
          neural_ok |= ok;
        }
      }

      const bool contained = outer_ok && neural_ok;
      if (a.mode == MODE_ANY || a.mode == MODE_ASSOC) {
        if (active && contained) { hit = true; hit_idx = b; }
        if (__all(hit || !valid)) break;
      } else if (a.mode == MODE_SAMPLE) {
        flags = (acc_outer ? 1 : 0) | ((acc_outer && neural_ok) ? 2 : 0);
      } else if (a.mode == MODE_COUNT) {
        count_out = k_cnt;
      }
    }

    if (valid && lg == 0) {
      if (a.mode == MODE_ANY) a.out_u8[pt] = hit ? 1 : 0;
      else if (a.mode == MODE_ASSOC) a.out_i32[pt] = hit_idx;
      else if (a.mode == MODE_SAMPLE) a.out_u8[pt] = flags;
      else if (a.mode == MODE_COUNT) a.out_u8[pt] = (unsigned char)count_out;
      else if (a.mode == MODE_SCORE) {
        a.out_f64[2 * pt] = r2_out;
        a.out_f64[2 * pt + 1] = score_out;
      }
    }
  }
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[0], cnt_outer);
    atomicAdd(&a.counters[1], cnt_ell);
    atomicAdd(&a.counters[2], cnt_mlp);
  }
}

template <int DT>
int launch_eval(const EvalArgs& a, hipStream_t stream) {
  const long long n_tiles = (a.n + 15) >> 4;
  long long blocks = (n_tiles + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(nb_eval_kernel<DT>, dim3((unsigned)blocks), dim3(256), 0,
                     stream, a);
  return NB_OK;
}

}  // namespace

static unsigned long long* g_eval_counters = nullptr;
void nb_eval_set_counters(unsigned long long* dev) { g_eval_counters = dev; }

// host entry used by nb_api.cpp
int nb_launch_eval(int dt, const double* const* blobs_dev, int nb, int mode,
                   const double* x, long long n, unsigned char* out_u8,
                   int* out_i32, double* out_f64, unsigned long long seed,
                   unsigned long long offset, hipStream_t stream) {
  EvalArgs a;
  a.blobs = blobs_dev; a.nb = nb; a.mode = mode; a.x = x; a.n = n;
  a.out_u8 = out_u8; a.out_i32 = out_i32; a.out_f64 = out_f64;
  a.seed = seed; a.offset = offset;
  a.counters = g_eval_counters;
  if (n <= 0 || nb <= 0) return NB_OK;
  switch (dt) {
    case 1: launch_eval<1>(a, stream); break;
    case 2: launch_eval<2>(a, stream); break;
    case 3: launch_eval<3>(a, stream); break;
    case 4: launch_eval<4>(a, stream); break;
    case 5: launch_eval<5>(a, stream); break;
    case 6: launch_eval<6>(a, stream); break;
    case 7: launch_eval<7>(a, stream); break;
    case 8: launch_eval<8>(a, stream); break;
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
