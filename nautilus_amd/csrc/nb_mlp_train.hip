// Emulator training on the matrix cores: NeuralNetworkEmulator.train ->
// MLPRegressor.fit restated for gfx950 (reference nautilus/neural.py:50-98;
// algorithm: sklearn/neural_network/_multilayer_perceptron.py:620-760
// (_fit_stochastic), :297-389 (_backprop), _stochastic_optimizers.py:255-287
// (Adam), _base.py:187-189 (squared loss)).
//
// All networks of an emulator train concurrently.  Every Adam step is two
// kernel launches on one stream (the kernel boundary is the grid-wide sync):
//
//  FB  one workgroup of four wavefronts per 16-row tile of the minibatch:
//      forward through the four layers (the output tiles of a layer are split
//      over the wavefronts, activations / deltas are exchanged through LDS in
//      [unit][row] layout), output delta, backward deltas through W^T read
//      from the same 16x16 tile-major weights; every weight operand is
//      loaded before the first barrier; activations and deltas go row-major
//      to a stash in global memory (L2 resident, ~0.8 MB per network).
//  G   one workgroup of four wavefronts per 16x16 weight tile: dW = act^T delta
//      over the rows of the minibatch, wavefront q contracting k-step q of
//      every 16-row tile; the four partial tiles meet in LDS, are added in a
//      fixed order (deterministic, no atomics) and every wavefront applies
//      Adam to a quarter of the tile in place.  The bias is row K of the
//      weight matrix (the activations carry a constant 1 in column K).
//
// Minibatch order comes from the host (numpy RandomState shuffles identical to
// sklearn's), so the device sees exactly the reference's data order.
#include "nb_common.h"
#include "../../include/nautilus_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

constexpr int MAXB = 208;        // minibatch rows padded to 16 (batch <= 200)
constexpr int LD1 = 112, LD2 = 64, LD3 = 32, LD4 = 16;
constexpr int NWAVE = 8;

struct NetState {
  double* W; double* M; double* V;   // tile-major weights and Adam moments
  double* stash;                     // A0 A1 A2 A3 D1 D2 D3 D4
  double* loss_curve;
  double* scal;   // [0] adam t  [1] best loss  [2] stale  [3] n_iter  [4] done
};

struct TrainArgs {
  const NetState* nets;
  const double* X;       // (n, D) standardised inputs
  const double* y;       // (n)
  const int* perm;       // (E, n_epochs, n)
  long long n;
  int n_dim, kt1, n_epochs, max_iter, n_iter_no_change, batch;
  double tol, lr, b1, b2, eps;
};

// out[h] = sum_k in[k] W[k][h] for one layer, tiles [kt][HT]
template <int KSMAX, int HT>
__device__ __forceinline__ void fwd_layer(const double* __restrict__ w,
                                          int ks_n, const double* in, int lane,
                                          double* out, bool relu) {
#pragma unroll
  for (int ht = 0; ht < HT; ++ht) {
    nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KSMAX; ++ks) {
      if (ks < ks_n) {
        const double a = w[((ks >> 2) * HT + ht) * NB_TILE + (ks & 3) * 64 + lane];
        acc = MFMA(a, in[ks], acc);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      out[4 * ht + r] = relu ? fmax(acc[r], 0.0) : acc[r];
  }
}

// din[k] = sum_h dout[h] W[k][h]  (k over KT tiles, h over HS k-steps of 4)
template <int KT, int HT, int HS>
__device__ __forceinline__ void bwd_layer(const double* __restrict__ w,
                                          const double* dout, int lane,
                                          double* din) {
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int hs = 0; hs < HS; ++hs) {
      const int h0 = 4 * hs;
      const double a = w[(kt * HT + (h0 >> 4)) * NB_TILE + li * 16 +
                         (h0 & 15) + lg];
      acc = MFMA(a, dout[hs], acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) din[4 * kt + r] = acc[r];
  }
}

// write a register block (unit = 4*j + lg of point li) to stash[pt][ld]
template <int NREG>
__device__ __forceinline__ void put_stash(double* __restrict__ base, int ld,
                                          int pt, int lane, const double* v) {
  const int lg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NREG; ++j) base[(long long)pt * ld + 4 * j + lg] = v[j];
}

// ---------------------------------------------------------------------------
// Step = two launches (kernel boundaries give grid-wide ordering and
// visibility for ~1.5 us each on MI355X, far cheaper than an in-kernel grid
// barrier across XCDs):
//   nb_train_fb_kernel  grid (row tiles of the minibatch, networks), 1 wave
//   nb_train_g_kernel   grid (weight tiles, networks), 1 wave
// and one nb_train_epoch_kernel per epoch for the stopping rule.
// ---------------------------------------------------------------------------
// ---- FB: forward + backward deltas of one 16-row tile ---------------------
// Four wavefronts share the tile: every layer's output tiles are split over
// the wavefronts and the activations / deltas are exchanged through LDS in
// [unit][row] layout (the B operand of the next layer is then one contiguous
// 512-byte read), which cuts the dependent MFMA chain of a tile from ~340 to
// ~100 instructions.
constexpr int LS = 17;   // LDS row stride (odd: conflict-free both ways)

// workgroup barrier that orders LDS traffic only: global loads already issued
// (the weight operands of later layers) stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int NREG>
__device__ __forceinline__ void lds_operand(const double* act, int lane,
                                            double* in) {
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < NREG; ++ks) in[ks] = act[(4 * ks + lg) * LS + li];
}

// cooperative LDS [unit][row] -> global stash [row][unit] (coalesced rows)
// (n_unit is a multiple of 16: thread -> row tid / 16, 16 consecutive units)
__device__ __forceinline__ void flush_stash(const double* act, double* dst,
                                            int ld, int n_unit, int tile,
                                            int tid) {
  const int r = tid >> 4, c = tid & 15;
  double* row = dst + (long long)(tile * 16 + r) * ld;
  for (int u = c; u < n_unit; u += 16) row[u] = act[u * LS + r];
}

// A operands of one forward output tile, loaded up front
template <int KSMAX>
__device__ __forceinline__ void load_fwd(const double* __restrict__ w, int ht_n,
                                         int ht, int ks_n, int lane,
                                         double* wr) {
#pragma unroll
  for (int ks = 0; ks < KSMAX; ++ks)
    wr[ks] = (ks < ks_n)
        ? w[((ks >> 2) * ht_n + ht) * NB_TILE + (ks & 3) * 64 + lane] : 0.0;
}

// A operands of one backward output tile (transposed access)
template <int HS>
__device__ __forceinline__ void load_bwd(const double* __restrict__ w, int ht_n,
                                         int kt, int lane, double* wr) {
  const int li = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int hs = 0; hs < HS; ++hs) {
    const int h0 = 4 * hs;
    wr[hs] = w[(kt * ht_n + (h0 >> 4)) * NB_TILE + li * 16 + (h0 & 15) + lg];
  }
}

template <int N>
__device__ __forceinline__ nb_d4 mma(const double* wr, const double* in) {
  nb_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
  for (int k = 0; k + 1 < N; k += 2) {
    acc0 = MFMA(wr[k], in[k], acc0);
    acc1 = MFMA(wr[k + 1], in[k + 1], acc1);
  }
  if (N & 1) acc0 = MFMA(wr[N - 1], in[N - 1], acc0);
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
  return acc0;
}

// forward output tile `ht`: acc = sum_ks W[ks][ht] * in[ks]
template <int KSMAX>
__device__ __forceinline__ nb_d4 fwd_tile(const double* __restrict__ w,
                                          int ht_n, int ht, int ks_n,
                                          const double* in, int lane) {
  nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < KSMAX; ++ks) {
    if (ks < ks_n) {
      const double a = w[((ks >> 2) * ht_n + ht) * NB_TILE + (ks & 3) * 64 + lane];
      acc = MFMA(a, in[ks], acc);
    }
  }
  return acc;
}

// backward output tile `kt`: acc = sum_hs W[kt][hs]^T * dout[hs]
template <int HS>
__device__ __forceinline__ nb_d4 bwd_tile(const double* __restrict__ w,
                                          int ht_n, int kt, const double* dout,
                                          int lane) {
  const int li = lane & 15, lg = lane >> 4;
  nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int hs = 0; hs < HS; ++hs) {
    const int h0 = 4 * hs;
    const double a = w[(kt * ht_n + (h0 >> 4)) * NB_TILE + li * 16 +
                       (h0 & 15) + lg];
    acc = MFMA(a, dout[hs], acc);
  }
  return acc;
}

// The rows of a tile's minibatch slice: permutation entry, the k-steps of the
// input block this wavefront fills (ks % 4 == wave) and the target.  Read-only
// data, so the resident kernel fetches the NEXT step's rows while it waits at
// the barrier that ends the current one (two dependent global latencies off
// the critical path).
template <int DT>
struct FbRows {
  long long row;
  double x[DT + 1];
  double yv;
};

template <int DT>
__device__ __forceinline__ void fb_gather(const TrainArgs& a, int net, int tile,
                                          int ep, long long start, int nb,
                                          FbRows<DT>& in) {
  constexpr int KS1MAX = 4 * DT + 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int D = a.n_dim, ld0 = 16 * a.kt1;
  const int* perm = a.perm + ((long long)net * a.n_epochs + ep) * a.n;
  const int pt = tile * 16 + li;
  const bool valid = pt < nb;
  in.row = valid ? perm[start + pt] : 0;
#pragma unroll
  for (int j = 0; j < DT + 1; ++j) {
    const int ks = 4 * j + wave;
    const int f = 4 * ks + lg;
    double v = 0.0;
    if (ks < KS1MAX && 4 * ks < ld0)
      v = (f < D) ? (valid ? a.X[in.row * D + f] : 0.0)
                  : ((f == D) ? 1.0 : 0.0);
    in.x[j] = v;
  }
  in.yv = (wave == 0 && lg == 0 && valid) ? a.y[in.row] : 0.0;
}

#ifdef NB_TRAIN_TIMING
__device__ long long g_train_ticks[64];
#define FB_STAMP(i)                                                           \
  do {                                                                        \
    if (net == 0 && tile == 0 && threadIdx.x == 0) {                          \
      const long long t_now = (long long)__builtin_amdgcn_s_memtime();        \
      g_train_ticks[i] += t_now - fb_prev;                                    \
      fb_prev = t_now;                                                        \
    }                                                                         \
  } while (0)
#else
#define FB_STAMP(i)
#endif

// LDS of a workgroup: the activation / delta blocks of FB in [unit][row]
// layout; the G phase reuses everything behind the input block (whose zero
// padding has to survive the step) for its partial tiles.
// weight tiles per workgroup and step in the resident kernel (32 workgroups)
template <int DT>
struct GTiles {
  static constexpr int N = (NB_HT1 * (DT + 1) + 38 + 31) / 32;
};

template <int DT>
struct FbLds {
  static constexpr int LD0MAX = 16 * (DT + 1);
  static constexpr int A0 = 0;
  static constexpr int A1 = A0 + LD0MAX * LS;
  static constexpr int A2 = A1 + LD1 * LS;
  static constexpr int A3 = A2 + LD2 * LS;
  static constexpr int D4 = A3 + LD3 * LS;
  static constexpr int D3 = D4 + LD4 * LS;
  static constexpr int D2 = D3 + LD3 * LS;
  static constexpr int TOTAL = D2 + LD2 * LS;
  static constexpr int G_RED = A1;             // 1024 doubles per tile
};

template <int DT, bool CHECK_DONE>
__device__ __forceinline__ void fb_body(const TrainArgs& a, const NetState& st,
                                        int net, int tile, int ep,
                                        long long start, int nb,
                                        const FbRows<DT>& rows,
                                        bool zero_input, double* lds) {
  constexpr int KS1MAX = 4 * DT + 1;
  constexpr int LD0MAX = 16 * (DT + 1);
  // activations / deltas of the tile in [unit][row] layout
  double* sA0 = lds + FbLds<DT>::A0;
  double* sA1 = lds + FbLds<DT>::A1;
  double* sA2 = lds + FbLds<DT>::A2;
  double* sA3 = lds + FbLds<DT>::A3;
  double* sD4 = lds + FbLds<DT>::D4;
  double* sD3 = lds + FbLds<DT>::D3;
  double* sD2 = lds + FbLds<DT>::D2;

  if (CHECK_DONE) {
    if (st.scal[4] != 0.0) return;               // network already stopped
  }
  // (opaque to the optimiser: the per-lane addresses of this function are
  // recomputed every step -- a few dozen VALU instructions -- instead of being
  // hoisted out of the resident kernel's step loop and spilled)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int D = a.n_dim, kt1 = a.kt1;
  const int ld0 = 16 * kt1;
  const int ks1 = (D + 1 + 3) >> 2;

  double* W1 = st.W;
  double* W2 = W1 + kt1 * NB_HT1 * NB_TILE;
  double* W3 = W2 + NB_HT1 * NB_HT2 * NB_TILE;
  double* W4 = W3 + NB_HT2 * NB_HT3 * NB_TILE;
  double* A0 = st.stash;
  double* A1 = A0 + MAXB * ld0;
  double* A2 = A1 + MAXB * LD1;
  double* A3 = A2 + MAXB * LD2;
  double* D1 = A3 + MAXB * LD3;
  double* D2 = D1 + MAXB * LD1;
  double* D3 = D2 + MAXB * LD2;
  double* D4 = D3 + MAXB * LD3;

  const int pt = tile * 16 + li;
  const bool valid = pt < nb;

  // ---- weight operands: every wavefront loads the A operands of ITS output
  // tiles straight into registers, one layer ahead of their use.  A wavefront
  // can only keep ~64 vector-memory instructions in flight: issuing all ~110
  // loads of the step up front (as an earlier version did) stalls it until
  // half of them have returned -- 4.5 us of the step.  Now layer 1 (+ layer 2
  // for n_dim <= 64) goes out first, the rest after the layer-1 products
  // (the barriers between the layers are compiler barriers for memory
  // operations, so the loads stay where they are written; they only wait
  // for LDS, so operands stay in flight across them). ----------------------
#ifdef NB_TRAIN_TIMING
  long long fb_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
  double w1r[2][KS1MAX], w2r[26], w3r[13], w4r[6], b4r[1], b3r[5], b2r[2][13];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int ht = (wave + 4 * rep < NB_HT1) ? wave + 4 * rep : wave;
    load_fwd<KS1MAX>(W1, NB_HT1, ht, ks1, lane, w1r[rep]);
  }
  if constexpr (DT <= 4) load_fwd<26>(W2, NB_HT2, wave, 26, lane, w2r);

  // ---- input block: k-step ks is handled by wavefront ks % 4 -------------
  // (all of it: rows >= ld0 are multiplied by zero weights and must not
  // hold NaN bit patterns)
  // (the padding only has to be cleared once per launch: every step rewrites
  // exactly the rows below ld0)
  if (zero_input) {
    for (int i = tid; i < LD0MAX * LS; i += 256) sA0[i] = 0.0;
  }
  lds_barrier();
  FB_STAMP(10);
#pragma unroll
  for (int j = 0; j < DT + 1; ++j) {
    const int ks = 4 * j + wave;
    if (ks < KS1MAX && 4 * ks < ld0) sA0[(4 * ks + lg) * LS + li] = rows.x[j];
  }
  lds_barrier();
  FB_STAMP(11);
  if constexpr (DT > 4) load_fwd<26>(W2, NB_HT2, wave, 26, lane, w2r);
  // the stash for the G phase is written as soon as a block is complete
  // (coalesced, fire and forget: the stores overlap the next layer)
  flush_stash(sA0, A0, ld0, ld0, tile, tid);

  // ---- layer 1: output tiles wave, wave + 4 ------------------------------
  {
    double in[KS1MAX];
    lds_operand<KS1MAX>(sA0, lane, in);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int ht = wave + 4 * rep;
      if (ht < NB_HT1) {
        const nb_d4 acc = mma<KS1MAX>(w1r[rep], in);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double v = fmax(acc[r], 0.0);
          const int unit = 16 * ht + 4 * r + lg;
          if (unit == NB_H1) v = 1.0;                    // bias unit
          sA1[unit * LS + li] = v;
        }
      }
    }
  }
  lds_barrier();
  FB_STAMP(12);
  // the remaining operands, in the order of use (in flight during layer 2)
  load_fwd<13>(W3, NB_HT3, wave & 1, 13, lane, w3r);
  load_fwd<6>(W4, 1, 0, 6, lane, w4r);
  load_bwd<1>(W4, 1, wave & 1, lane, b4r);
  load_bwd<5>(W3, NB_HT3, wave, lane, b3r);
  flush_stash(sA1, A1, LD1, LD1, tile, tid);

  // ---- layer 2: output tile = wave ----------------------------------------
  {
    double in[26];
    lds_operand<26>(sA1, lane, in);
    const nb_d4 acc = mma<26>(w2r, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = fmax(acc[r], 0.0);
      const int unit = 16 * wave + 4 * r + lg;
      if (unit == NB_H2) v = 1.0;
      sA2[unit * LS + li] = v;
    }
  }
  lds_barrier();
  FB_STAMP(13);
  // (the operands of the last backward product take the registers layer 2's
  // have left; five stages until they are needed)
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int ht = (wave + 4 * rep < NB_HT1) ? wave + 4 * rep : wave;
    load_bwd<13>(W2, NB_HT2, ht, lane, b2r[rep]);
  }
  flush_stash(sA2, A2, LD2, LD2, tile, tid);

  // ---- layer 3: two output tiles ------------------------------------------
  if (wave < NB_HT3) {
    double in[13];
    lds_operand<13>(sA2, lane, in);
    const nb_d4 acc = mma<13>(w3r, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = fmax(acc[r], 0.0);
      const int unit = 16 * wave + 4 * r + lg;
      if (unit == NB_H3) v = 1.0;
      sA3[unit * LS + li] = v;
    }
  }
  lds_barrier();
  FB_STAMP(14);
  flush_stash(sA3, A3, LD3, LD3, tile, tid);

  // ---- output layer, delta 4, loss partial (wavefront 0) -------------------
  if (wave == 0) {
    double in[6];
    lds_operand<6>(sA3, lane, in);
    const nb_d4 acc = mma<6>(w4r, in);
    double d40 = 0.0;
    if (lg == 0 && valid) d40 = acc[0] - rows.yv;    // sklearn :365
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 4 * r + lg;
      const double v = (unit == 0) ? d40 : 0.0;
      sD4[unit * LS + li] = v;
    }
    double lp = 0.5 * d40 * d40;
    for (int s = 8; s >= 1; s >>= 1) lp += __shfl_xor(lp, s);
    if (lane == 0) st.scal[8 + tile] = lp;
  }
  lds_barrier();
  FB_STAMP(15);
  flush_stash(sD4, D4, LD4, LD4, tile, tid);

  // ---- delta 3 (ReLU mask = activation == 0; bias unit carries none) ------
  if (wave < NB_HT3) {
    double dout[1];
    lds_operand<1>(sD4, lane, dout);
    const nb_d4 acc = mma<1>(b4r, dout);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 16 * wave + 4 * r + lg;
      double v = acc[r];
      if (sA3[unit * LS + li] == 0.0 || unit == NB_H3) v = 0.0;
      sD3[unit * LS + li] = v;
    }
  }
  lds_barrier();
  FB_STAMP(16);
  flush_stash(sD3, D3, LD3, LD3, tile, tid);

  // ---- delta 2 --------------------------------------------------------------
  {
    double dout[5];
    lds_operand<5>(sD3, lane, dout);
    const nb_d4 acc = mma<5>(b3r, dout);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 16 * wave + 4 * r + lg;
      double v = acc[r];
      if (sA2[unit * LS + li] == 0.0 || unit == NB_H2) v = 0.0;
      sD2[unit * LS + li] = v;
    }
  }
  lds_barrier();
  FB_STAMP(17);
  flush_stash(sD2, D2, LD2, LD2, tile, tid);

  // ---- delta 1 --------------------------------------------------------------
  {
    double dout[13];
    lds_operand<13>(sD2, lane, dout);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int kt = wave + 4 * rep;
      if (kt < NB_HT1) {
        const nb_d4 acc = mma<13>(b2r[rep], dout);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = 16 * kt + 4 * r + lg;
          double v = acc[r];
          if (sA1[unit * LS + li] == 0.0 || unit == NB_H1) v = 0.0;
          D1[(long long)pt * LD1 + unit] = v;
        }
      }
    }
  }

  FB_STAMP(18);
}

template <int DT>
__global__ void __launch_bounds__(256)
nb_train_fb_kernel(TrainArgs a, int ep, long long start, int nb) {
  __shared__ __attribute__((aligned(16))) double lds[FbLds<DT>::TOTAL];
  const NetState st = a.nets[blockIdx.y];
  FbRows<DT> rows;
  fb_gather<DT>(a, (int)blockIdx.y, (int)blockIdx.x, ep, start, nb, rows);
  fb_body<DT, true>(a, st, (int)blockIdx.y, (int)blockIdx.x, ep, start, nb,
                    rows, true, lds);
}

// ---- G: dW of 16x16 weight tiles over the minibatch + Adam ------------------
// the step's loss partials folded into the epoch sum, in tile order
// (deterministic); one wavefront, partial i in lane i
__device__ __forceinline__ void loss_fold(const NetState& st, int nb,
                                          int lane) {
  const int n_tiles = (nb + 15) >> 4;
  const double p = (lane < n_tiles) ? st.scal[8 + lane] : 0.0;
  double acc = st.scal[5];
  for (int i = 0; i < n_tiles; ++i) acc += __shfl(p, i);
  if (lane == 0) st.scal[5] = acc;
}

// step size of Adam step t (sklearn _stochastic_optimizers.py:276-279)
__device__ __forceinline__ double adam_lr(const TrainArgs& a, long long t_adam) {
  return a.lr * sqrt(1.0 - pow(a.b2, (double)t_adam)) /
         (1.0 - pow(a.b1, (double)t_adam));
}

constexpr int G_ROWT = MAXB / 16;   // 16-row tiles of a minibatch

struct GTile {
  const double* as;   // activations, column block of the tile (wave-uniform)
  const double* bs;   // deltas, column block of the tile (wave-uniform)
  int lda, ldb;       // row strides
  long long woff;     // offset of the weight tile
};

// (everything here is wave-uniform: scalar registers)
__device__ __forceinline__ GTile g_decode(const TrainArgs& a,
                                          const NetState& st, int gt) {
  const int kt1 = a.kt1;
  const int ld0 = 16 * kt1;
  const int n_gt1 = kt1 * NB_HT1, n_gt2 = NB_HT1 * NB_HT2,
            n_gt3 = NB_HT2 * NB_HT3;
  const double* A0 = st.stash;
  const double* A1 = A0 + MAXB * ld0;
  const double* A2 = A1 + MAXB * LD1;
  const double* A3 = A2 + MAXB * LD2;
  const double* D1 = A3 + MAXB * LD3;
  const double* D2 = D1 + MAXB * LD1;
  const double* D3 = D2 + MAXB * LD2;
  const double* D4 = D3 + MAXB * LD3;
  int kt, ht;
  GTile t;
  if (gt < n_gt1) {
    kt = gt / NB_HT1; ht = gt % NB_HT1; t.as = A0; t.lda = ld0; t.bs = D1;
    t.ldb = LD1;
    t.woff = (long long)(kt * NB_HT1 + ht) * NB_TILE;
  } else if (gt < n_gt1 + n_gt2) {
    const int g = gt - n_gt1;
    kt = g / NB_HT2; ht = g % NB_HT2; t.as = A1; t.lda = LD1; t.bs = D2;
    t.ldb = LD2;
    t.woff = (long long)(n_gt1 + kt * NB_HT2 + ht) * NB_TILE;
  } else if (gt < n_gt1 + n_gt2 + n_gt3) {
    const int g = gt - n_gt1 - n_gt2;
    kt = g / NB_HT3; ht = g % NB_HT3; t.as = A2; t.lda = LD2; t.bs = D3;
    t.ldb = LD3;
    t.woff = (long long)(n_gt1 + n_gt2 + kt * NB_HT3 + ht) * NB_TILE;
  } else {
    const int g = gt - n_gt1 - n_gt2 - n_gt3;
    kt = g; ht = 0; t.as = A3; t.lda = LD3; t.bs = D4; t.ldb = LD4;
    t.woff = (long long)(n_gt1 + n_gt2 + n_gt3 + kt) * NB_TILE;
  }
  t.as += 16 * kt;
  t.bs += 16 * ht;
  return t;
}

// operands of quarter `wave` of a tile: rows 16 rt + 4 wave + lg
__device__ __forceinline__ void g_load(const GTile& t, int n_rt, int wave,
                                       int lane, double* av, double* bv) {
  const int li = lane & 15, lg = lane >> 4;
  const int oa = (4 * wave + lg) * t.lda + li;
  const int ob = (4 * wave + lg) * t.ldb + li;
#pragma unroll
  for (int rt = 0; rt < G_ROWT; ++rt) {
    const bool on = rt < n_rt;
    av[rt] = on ? t.as[oa + rt * 16 * t.lda] : 0.0;
    bv[rt] = on ? t.bs[ob + rt * 16 * t.ldb] : 0.0;
  }
}

#ifdef NB_TRAIN_TIMING
#define G_STAMP(i)                                                            \
  do {                                                                        \
    if (MAXT > 1 && first == 0 && threadIdx.x == 0) {                         \
      const long long t_now = (long long)__builtin_amdgcn_s_memtime();        \
      g_train_ticks[i] += t_now - g_prev;                                     \
      g_prev = t_now;                                                         \
    }                                                                         \
  } while (0)
#else
#define G_STAMP(i)
#endif

// Tiles first, first + stride, ... (< n_gt, at most MAXT of them) of one
// network, by a workgroup of four wavefronts.  Wavefront q contracts the rows
// 16 rt + 4 q + lg of the minibatch (k-step q of every 16-row tile, in the
// order of rt), so a tile is four independent chains of at most 13 MFMAs on
// four SIMDs; all operand loads of a tile are issued before the chain of the
// previous one.  The partial tiles go through LDS; wavefront r then owns rows
// lg + 4 r of every tile: gradient = ((p0 + p1) + p2) + p3, Adam (sklearn
// _stochastic_optimizers.py:255-287) in place.
template <int MAXT>
__device__ __forceinline__ void g_phase(const TrainArgs& a, const NetState& st,
                                        int first, int stride, int nb,
                                        double lr_t, double* red) {
  int lane = threadIdx.x & 63;
  // (opaque to the optimiser: per-lane addresses derived from it are
  // recomputed every step instead of being kept -- and spilled -- across the
  // forward / backward pass)
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int n_gt = nb_net_tiles(a.kt1);
  const int n_rt = (nb + 15) >> 4;
#ifdef NB_TRAIN_TIMING
  long long g_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
  GTile tl[MAXT];
  double av[2][G_ROWT], bv[2][G_ROWT];
  double w_old[MAXT], m_old[MAXT], v_old[MAXT];
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    const int gt = first + i * stride;
    if (gt < n_gt) tl[i] = g_decode(a, st, gt);
  }
  if (first < n_gt) g_load(tl[0], n_rt, wave, lane, av[0], bv[0]);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    if (first + i * stride < n_gt) {
      const long long idx = tl[i].woff + (lg + 4 * wave) * 16 + li;
      w_old[i] = st.W[idx]; m_old[i] = st.M[idx]; v_old[i] = st.V[idx];
    }
  }
  G_STAMP(30);
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    if (first + i * stride < n_gt) {
      if (i + 1 < MAXT) {
        if (first + (i + 1) * stride < n_gt)
          g_load(tl[i + 1], n_rt, wave, lane, av[(i + 1) & 1], bv[(i + 1) & 1]);
      }
      nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int rt = 0; rt < G_ROWT; ++rt)
        acc = MFMA(av[i & 1][rt], bv[i & 1][rt], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        red[((i * 4 + wave) * 4 + r) * 64 + lane] = acc[r];
    }
  }
  G_STAMP(31);
  lds_barrier();
  const double inv_nb = 1.0 / (double)nb;
#pragma unroll
  for (int i = 0; i < MAXT; ++i) {
    if (first + i * stride < n_gt) {
      const double* p = red + (i * 16 + wave) * 64 + lane;
      const double sum = ((p[0] + p[4 * 64]) + p[8 * 64]) + p[12 * 64];
      const long long idx = tl[i].woff + (lg + 4 * wave) * 16 + li;
      const double g = sum * inv_nb;
      const double m = a.b1 * m_old[i] + (1.0 - a.b1) * g;
      const double v = a.b2 * v_old[i] + (1.0 - a.b2) * (g * g);
      st.M[idx] = m;
      st.V[idx] = v;
      st.W[idx] = w_old[i] + -lr_t * m / (sqrt(v) + a.eps);
    }
  }
  G_STAMP(32);
}

__global__ void __launch_bounds__(256)
nb_train_g_kernel(TrainArgs a, int nb, long long t_adam) {
  __shared__ __attribute__((aligned(16))) double red[1024];
  const NetState st = a.nets[blockIdx.y];
  if (st.scal[4] != 0.0) return;                 // network already stopped
  // the first workgroup also folds the step's loss (the resident kernel gives
  // that to its least loaded workgroup)
  if (blockIdx.x == 0 && threadIdx.x < 64) loss_fold(st, nb, (int)threadIdx.x);
  g_phase<1>(a, st, (int)blockIdx.x, 1 << 30, nb, adam_lr(a, t_adam), red);
}

// end of epoch: loss curve and the stopping rule of _fit_stochastic
// (sklearn/_multilayer_perceptron.py:730-760, 819-822); one thread
__device__ __forceinline__ void epoch_body(const TrainArgs& a,
                                           const NetState& st,
                                           long long t_adam) {
  if (st.scal[4] != 0.0) return;
  const double loss = st.scal[5] / (double)a.n;
  int n_iter = (int)st.scal[3];
  double best = st.scal[1];
  int stale = (int)st.scal[2];
  st.loss_curve[n_iter] = loss;
  n_iter += 1;
  if (loss > best - a.tol) stale += 1; else stale = 0;
  if (loss < best) best = loss;
  st.scal[0] = (double)t_adam;
  st.scal[1] = best;
  st.scal[2] = (double)stale;
  st.scal[3] = (double)n_iter;
  st.scal[5] = 0.0;
  if (stale > a.n_iter_no_change || n_iter >= a.max_iter) st.scal[4] = 1.0;
}

__global__ void nb_train_epoch_kernel(TrainArgs a, long long t_adam) {
  const NetState st = a.nets[blockIdx.x];
  if (threadIdx.x == 0) epoch_body(a, st, t_adam);
}

// ---------------------------------------------------------------------------
// One launch per chunk of epochs, one XCD per network.
//
// Two launches per Adam step cost about a third of the step in dispatch and
// drain (the end-of-kernel release writes the L2 of every XCD back so that the
// next kernel's workgroups, anywhere on the chip, see the data).  A resident
// kernel with an agent-scope barrier pays the same write-back inside the
// kernel (measured: slower).  What does work: consecutive workgroup ids go
// round-robin over the 8 XCDs (workgroup i -> XCD i mod 8; checked at run
// time through HW_REG_XCC_ID), so the 17 workgroups {net, net + 8, ...} of a
// network share one L2.  Within one L2 a producer only has to wait for its
// stores (s_waitcnt) and a consumer to drop its CU's L1 (buffer_inv): no L2
// write-back, and the barrier is one atomic in that L2.
// ---------------------------------------------------------------------------
constexpr int XCD_COUNT = 8;
constexpr int XCD_SLOTS = 32;            // workgroups per network: one per CU
constexpr int SYNC_WORDS = 4;            // counter, error, (unused), ticket
constexpr int SYNC_LIMIT = 1 << 23;

// (split into arrive / wait so that read-only prefetches can be issued in
// between: after the workgroup has signalled, before it starts polling)
__device__ __forceinline__ void xcd_arrive(int* counter) {
  __syncthreads();
  if (threadIdx.x == 0) {
    // stores of this workgroup are in L2 once the counters drain
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void xcd_wait(int* counter, int* err, int& phase,
                                         int n_wg) {
  if (threadIdx.x == 0) {
    const int target = (++phase) * n_wg;
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > SYNC_LIMIT) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    // invalidate this CU's vector L1: later loads come from the shared L2
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ void xcd_barrier(int* counter, int* err, int& phase,
                                            int n_wg) {
  xcd_arrive(counter);
  xcd_wait(counter, err, phase, n_wg);
}

struct XcdMap {
  int n_nets;
  int net[XCD_COUNT];
};

__global__ void nb_xcc_probe_kernel(int* out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = (int)(xcc & 15);
  }
}

#ifdef NB_TRAIN_TIMING
#define TR_STAMP(i)                                                           \
  do {                                                                        \
    if (net == 0 && slot == 0 && threadIdx.x == 0) {                          \
      const long long t_now = (long long)__builtin_amdgcn_s_memtime();        \
      g_train_ticks[i] += t_now - t_prev;                                     \
      t_prev = t_now;                                                         \
    }                                                                         \
  } while (0)
#else
#define TR_STAMP(i)
#endif

// (two workgroups per CU: the register budget of 256 leaves every CU of an
// owned XCD a free slot, through which the workgroups of OTHER grids -- a
// concurrent trainer's, which leave at once here, or any other kernel's --
// pass while this one is resident)
template <int DT>
__global__ void __launch_bounds__(256, 2)
nb_train_xcd_kernel(TrainArgs a, XcdMap map, long long t_adam0, int* sync) {
  // concurrent trainers (the neural bounds of a multi-modal NautilusBound)
  // own disjoint XCDs; map.net[x] = network of XCD x or -1
  // The workgroup asks the hardware which XCD it runs on and takes a ticket
  // there: the first XCD_SLOTS arrivals on an XCD this trainer owns are the
  // network's workgroups, everybody else leaves.  (The dispatcher deals the
  // workgroups of a grid out round-robin over the XCDs, 17 each for this
  // grid, but not necessarily starting at XCD 0 when several queues are
  // active -- two concurrent trainers that both assumed blockIdx % 8 could
  // end up with 34 resident workgroups on the 32 CUs of one XCD and wait for
  // each other forever.)
  __shared__ int sh_slot;
  __shared__ __attribute__((aligned(16))) double lds[FbLds<DT>::TOTAL];
  static_assert(FbLds<DT>::TOTAL - FbLds<DT>::G_RED >= GTiles<DT>::N * 1024,
                "the partial tiles of G fit behind the input block");
  static_assert(XCD_SLOTS == 32, "GTiles assumes 32 workgroups");
  const int n_nets = map.n_nets;
  unsigned xcc_id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
  const int net = map.net[xcc_id & (XCD_COUNT - 1)];
  if (net < 0 || net >= n_nets) return;
  int* counter = sync + SYNC_WORDS * net;
  int* err = counter + 1;
  int* ticket = counter + 3;
  if (threadIdx.x == 0) sh_slot = atomicAdd(ticket, 1);
  __syncthreads();
  const int slot = __builtin_amdgcn_readfirstlane(sh_slot);
  if (slot >= XCD_SLOTS) return;
  const NetState st = a.nets[net];
  int phase = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n = a.n;
  const int steps = (int)((n + a.batch - 1) / a.batch);
  xcd_barrier(counter, err, phase, XCD_SLOTS);
  long long t_adam = t_adam0;
  FbRows<DT> rows;
  bool have_rows = false;        // rows = the slice of the step about to run
  bool zero_input = true;        // the padding of the input block, once
#ifdef NB_TRAIN_TIMING
  long long t_prev = (long long)__builtin_amdgcn_s_memtime();
#endif
  for (int ep = 0; ep < a.n_epochs; ++ep) {
    // uniform over the network's workgroups: the flag only changes in
    // epoch_body, which is followed by a barrier
    const bool done = ((volatile double*)st.scal)[4] != 0.0;
    for (int sidx = 0; sidx < steps; ++sidx) {
      const long long start = (long long)sidx * a.batch;
      const int nb = (int)((n - start < a.batch) ? (n - start) : a.batch);
      t_adam += 1;
      if (done) continue;
      TR_STAMP(0);
      if (slot * 16 < nb) {
        if (!have_rows) fb_gather<DT>(a, net, slot, ep, start, nb, rows);
        fb_body<DT, false>(a, st, net, slot, ep, start, nb, rows, zero_input,
                           lds);
        zero_input = false;
      }
      TR_STAMP(1);
      // (the step size -- two pow() -- is computed while waiting)
      xcd_arrive(counter);
      const double lr_t = adam_lr(a, t_adam);
      xcd_wait(counter, err, phase, XCD_SLOTS);
      TR_STAMP(2);
      // weight tiles slot, slot + 32, ... on the four wavefronts of this
      // workgroup (all 32 CUs of the XCD take part, also the ones without a
      // row tile in FB); the last workgroup has the fewest tiles and folds
      // the loss
      if (slot == XCD_SLOTS - 1 && wave == 3) loss_fold(st, nb, lane);
      g_phase<GTiles<DT>::N>(a, st, slot, XCD_SLOTS, nb, lr_t,
                      lds + FbLds<DT>::G_RED);
      TR_STAMP(3);
      xcd_arrive(counter);
      {
        // rows of the next step (next epoch's permutation after the last one)
        const bool last = sidx + 1 == steps;
        const int ep2 = last ? ep + 1 : ep;
        const long long start2 = last ? 0 : start + a.batch;
        const int nb2 = (int)((n - start2 < a.batch) ? (n - start2) : a.batch);
        have_rows = ep2 < a.n_epochs && slot * 16 < nb2;
        if (have_rows) fb_gather<DT>(a, net, slot, ep2, start2, nb2, rows);
      }
      xcd_wait(counter, err, phase, XCD_SLOTS);
      TR_STAMP(4);
      if (__hip_atomic_load(err, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT) != 0)
        return;
    }
    if (done) continue;
    if (slot == 0 && threadIdx.x == 0) epoch_body(a, st, t_adam);
    xcd_barrier(counter, err, phase, XCD_SLOTS);
  }
}

void put_w(double* tiles, int ht_n, int k, int h, double v) {
  tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE + (k & 15) * 16 +
        (h & 15)] = v;
}
double get_w(const double* tiles, int ht_n, int k, int h) {
  return tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE + (k & 15) * 16 +
               (h & 15)];
}

}  // namespace

// One-time check of the placement the resident kernel relies on: workgroups
// i, i + 8, i + 16, ... of a 1-D grid run on the same XCD.
static bool xcd_pinning_available() {
  static int cached = -1;
  if (cached >= 0) return cached == 1;
  cached = 0;
  const int n = XCD_COUNT * XCD_SLOTS;
  int* dev = nullptr;
  if (hipMalloc((void**)&dev, n * sizeof(int)) != hipSuccess) return false;
  hipLaunchKernelGGL(nb_xcc_probe_kernel, dim3(n), dim3(64), 0, 0, dev);
  int host[XCD_COUNT * XCD_SLOTS];
  const hipError_t e = hipMemcpy(host, dev, sizeof host, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  bool ok = true;
  for (int i = 0; i < n; ++i) ok = ok && host[i] == host[i % XCD_COUNT];
  cached = ok ? 1 : 0;
  return ok;
}

static unsigned g_xcd_in_use = 0;   // XCDs owned by live resident trainers

struct nb_trainer {
  int n_dim = 0, E = 0, kt1 = 0, dt = 0;
  long long n = 0;
  long long n_w = 0;
  const double* X = nullptr;
  const double* y = nullptr;
  std::vector<NetState> nets_host;
  NetState* nets_dev = nullptr;
  double* pool = nullptr;          // one allocation for all per-net buffers
  int max_iter = 10000, n_iter_no_change = 10, batch = 200;
  double tol = 0.0, lr = 1e-2, b1 = 0.9, b2 = 0.999, eps = 1e-8;
  long long t_adam = 0;
  int* sync_dev = nullptr;         // per network: counter, error, xcc mask
  bool two_launch = false;         // fall back to two launches per step
  XcdMap xcd_map;                  // XCDs owned by this trainer's networks
  unsigned xcd_owned = 0;
};

extern "C" {

int nb_trainer_create(int32_t n_dim, int32_t n_networks, int64_t n_rows,
                      const double* x_dev, const double* y_dev,
                      const double* const* coefs, const double* const* icpts,
                      nb_trainer** out) {
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT || n_networks < 1 || n_rows < 1) {
    nb_set_error("bad trainer shape (n_dim=%d, n_networks=%d, n_rows=%lld)",
                 n_dim, n_networks, (long long)n_rows);
    return NB_ERR_ARG;
  }
  nb_trainer* t = new nb_trainer();
  t->n_dim = n_dim; t->E = n_networks; t->n = n_rows;
  t->dt = (n_dim + 15) / 16;
  t->kt1 = (n_dim + 1 + 15) / 16;
  t->X = x_dev; t->y = y_dev;
  t->n_w = (long long)nb_net_tiles(t->kt1) * NB_TILE;
  const long long stash = (long long)MAXB * (16 * t->kt1 + 2 * (LD1 + LD2 + LD3) + LD4);
  const long long curve = t->max_iter;
  const long long per_net = 3 * t->n_w + stash + curve + 32;
  const size_t bytes = (size_t)per_net * n_networks * sizeof(double);
  hipError_t e = hipMalloc((void**)&t->pool, bytes);
  if (e == hipSuccess) e = hipMemset(t->pool, 0, bytes);
  if (e == hipSuccess)
    e = hipMalloc((void**)&t->nets_dev, n_networks * sizeof(NetState));
  if (e == hipSuccess)
    e = hipMalloc((void**)&t->sync_dev,
                  SYNC_WORDS * XCD_COUNT * sizeof(int));
  t->two_launch = n_networks > XCD_COUNT ||
                  getenv("NB_TRAIN_TWO_LAUNCH") != nullptr ||
                  !xcd_pinning_available();
  // A resident workgroup takes a whole CU; two resident kernels competing for
  // the CUs of one XCD could each end up partially dispatched and wait for
  // each other.  So every trainer owns its XCDs exclusively; a trainer that
  // finds too few free ones trains with two launches per step instead.
  t->xcd_map.n_nets = n_networks;
  for (int x = 0; x < XCD_COUNT; ++x) t->xcd_map.net[x] = -1;
  if (!t->two_launch) {
    int assigned = 0;
    for (int x = 0; x < XCD_COUNT && assigned < n_networks; ++x)
      if (!(g_xcd_in_use & (1u << x))) {
        t->xcd_map.net[x] = assigned++;
        t->xcd_owned |= 1u << x;
      }
    if (assigned < n_networks) {
      t->xcd_owned = 0;
      t->two_launch = true;
    } else {
      g_xcd_in_use |= t->xcd_owned;
    }
  }
  if (getenv("NB_TRAIN_DEBUG") != nullptr)
    fprintf(stderr, "[trainer] nets=%d n=%lld two_launch=%d owned=%02x in_use=%02x\n",
            n_networks, (long long)n_rows, (int)t->two_launch, t->xcd_owned,
            g_xcd_in_use);
  if (e != hipSuccess) {
    nb_set_error("trainer allocation failed: %s", hipGetErrorString(e));
    nb_trainer_destroy(t);
    return NB_ERR_HIP;
  }
  std::vector<double> w((size_t)t->n_w);
  for (int i = 0; i < n_networks; ++i) {
    NetState s;
    double* base = t->pool + (size_t)i * per_net;
    s.W = base; s.M = s.W + t->n_w; s.V = s.M + t->n_w;
    s.stash = s.V + t->n_w;
    s.loss_curve = s.stash + stash;
    s.scal = s.loss_curve + curve;
    t->nets_host.push_back(s);
    std::fill(w.begin(), w.end(), 0.0);
    double* w1 = w.data();
    double* w2 = w1 + (size_t)t->kt1 * NB_HT1 * NB_TILE;
    double* w3 = w2 + (size_t)NB_HT1 * NB_HT2 * NB_TILE;
    double* w4 = w3 + (size_t)NB_HT2 * NB_HT3 * NB_TILE;
    const double* const* c = coefs + 4 * i;
    const double* const* b = icpts + 4 * i;
    for (int k = 0; k < n_dim; ++k)
      for (int h = 0; h < NB_H1; ++h) put_w(w1, NB_HT1, k, h, c[0][(size_t)k * NB_H1 + h]);
    for (int h = 0; h < NB_H1; ++h) put_w(w1, NB_HT1, n_dim, h, b[0][h]);
    for (int k = 0; k < NB_H1; ++k)
      for (int h = 0; h < NB_H2; ++h) put_w(w2, NB_HT2, k, h, c[1][(size_t)k * NB_H2 + h]);
    for (int h = 0; h < NB_H2; ++h) put_w(w2, NB_HT2, NB_H1, h, b[1][h]);
    for (int k = 0; k < NB_H2; ++k)
      for (int h = 0; h < NB_H3; ++h) put_w(w3, NB_HT3, k, h, c[2][(size_t)k * NB_H3 + h]);
    for (int h = 0; h < NB_H3; ++h) put_w(w3, NB_HT3, NB_H2, h, b[2][h]);
    for (int k = 0; k < NB_H3; ++k) put_w(w4, 1, k, 0, c[3][k]);
    put_w(w4, 1, NB_H3, 0, b[3][0]);
    e = hipMemcpy(s.W, w.data(), (size_t)t->n_w * sizeof(double),
                  hipMemcpyHostToDevice);
    const double scal0[8] = {0.0, INFINITY, 0.0, 0.0, 0.0, 0.0, 0, 0};
    if (e == hipSuccess)
      e = hipMemcpy(s.scal, scal0, sizeof scal0, hipMemcpyHostToDevice);
    if (e != hipSuccess) break;
  }
  if (e == hipSuccess)
    e = hipMemcpy(t->nets_dev, t->nets_host.data(),
                  n_networks * sizeof(NetState), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    nb_set_error("trainer upload failed: %s", hipGetErrorString(e));
    nb_trainer_destroy(t);
    return NB_ERR_HIP;
  }
  *out = t;
  return NB_OK;
}

int nb_trainer_set_hparams(nb_trainer* t, double lr, double beta1,
                           double beta2, double epsilon, int32_t batch,
                           int32_t max_iter, int32_t n_iter_no_change,
                           double tol) {
  if (batch < 1 || batch > 200 || max_iter < 1 || max_iter > 10000) {
    nb_set_error("trainer: batch must be 1..200 and max_iter 1..10000");
    return NB_ERR_UNSUPPORTED;
  }
  t->lr = lr; t->b1 = beta1; t->b2 = beta2; t->eps = epsilon;
  t->batch = batch; t->max_iter = max_iter;
  t->n_iter_no_change = n_iter_no_change; t->tol = tol;
  return NB_OK;
}

int nb_trainer_run(nb_trainer* t, const int32_t* perm_dev, int32_t n_epochs,
                   int32_t* status_host, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  TrainArgs a;
  a.nets = t->nets_dev; a.X = t->X; a.y = t->y; a.perm = perm_dev;
  a.n = t->n; a.n_dim = t->n_dim; a.kt1 = t->kt1; a.n_epochs = n_epochs;
  a.max_iter = t->max_iter; a.n_iter_no_change = t->n_iter_no_change;
  a.batch = (int)((t->n < t->batch) ? t->n : t->batch);
  a.tol = t->tol; a.lr = t->lr; a.b1 = t->b1; a.b2 = t->b2; a.eps = t->eps;
  // Adam step counter continues across calls (host mirror of scal[0])
  const long long n = t->n;
  const int steps_per_epoch = (int)((n + a.batch - 1) / a.batch);
  const int n_gt = nb_net_tiles(t->kt1);
  if (!t->two_launch) {
    NB_HIP_CHECK(hipMemsetAsync(t->sync_dev, 0,
                                SYNC_WORDS * XCD_COUNT * sizeof(int), s));
    const dim3 grid(XCD_COUNT * XCD_SLOTS), blk(256);
    switch (t->dt) {
#define NB_CASE(DT_)                                                       \
      case DT_:                                                            \
        hipLaunchKernelGGL(nb_train_xcd_kernel<DT_>, grid, blk, 0, s, a,   \
                           t->xcd_map, t->t_adam, t->sync_dev);            \
        break;
      NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4)
      NB_CASE(5) NB_CASE(6) NB_CASE(7) NB_CASE(8)
#undef NB_CASE
      default: nb_set_error("n_dim unsupported"); return NB_ERR_UNSUPPORTED;
    }
    t->t_adam += (long long)n_epochs * steps_per_epoch;
    NB_HIP_CHECK(hipGetLastError());
    if (status_host != nullptr)
      return nb_trainer_status(t, status_host, stream);
    return NB_OK;
  }
  for (int ep = 0; ep < n_epochs; ++ep) {
    for (int sidx = 0; sidx < steps_per_epoch; ++sidx) {
      const long long start = (long long)sidx * a.batch;
      const int nb = (int)((n - start < a.batch) ? (n - start) : a.batch);
      const dim3 gfb((nb + 15) / 16, t->E), gg(n_gt, t->E), blk(256), blk_fb(256);
      t->t_adam += 1;
      switch (t->dt) {
        case 1: hipLaunchKernelGGL(nb_train_fb_kernel<1>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 2: hipLaunchKernelGGL(nb_train_fb_kernel<2>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 3: hipLaunchKernelGGL(nb_train_fb_kernel<3>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 4: hipLaunchKernelGGL(nb_train_fb_kernel<4>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 5: hipLaunchKernelGGL(nb_train_fb_kernel<5>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 6: hipLaunchKernelGGL(nb_train_fb_kernel<6>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 7: hipLaunchKernelGGL(nb_train_fb_kernel<7>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        case 8: hipLaunchKernelGGL(nb_train_fb_kernel<8>, gfb, blk_fb, 0, s, a, ep, start, nb); break;
        default: nb_set_error("n_dim unsupported"); return NB_ERR_UNSUPPORTED;
      }
      hipLaunchKernelGGL(nb_train_g_kernel, gg, blk, 0, s, a, nb, t->t_adam);
    }
    hipLaunchKernelGGL(nb_train_epoch_kernel, dim3(t->E), dim3(64), 0, s, a,
                       t->t_adam);
  }
  NB_HIP_CHECK(hipGetLastError());
  if (status_host != nullptr) return nb_trainer_status(t, status_host, stream);
  return NB_OK;
}

int nb_trainer_status(nb_trainer* t, int32_t* status_host, void* stream) {
  NB_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (!t->two_launch) {
    int sync[SYNC_WORDS * XCD_COUNT];
    NB_HIP_CHECK(hipMemcpy(sync, t->sync_dev, sizeof sync,
                           hipMemcpyDeviceToHost));
    for (int i = 0; i < t->E; ++i)
      if (sync[SYNC_WORDS * i + 1] != 0) {
        nb_set_error("resident training kernel failed for network %d (%s); "
                     "set NB_TRAIN_TWO_LAUNCH=1 to train with two launches "
                     "per step", i,
                     sync[SYNC_WORDS * i + 1] == 2
                         ? "its workgroups do not share an XCD"
                         : "barrier timeout");
        return NB_ERR_HIP;
      }
  }
  for (int i = 0; i < t->E; ++i) {
    double scal[8];
    NB_HIP_CHECK(hipMemcpy(scal, t->nets_host[i].scal, sizeof scal,
                           hipMemcpyDeviceToHost));
    const int n_iter = (int)scal[3];
    status_host[i] = (scal[4] != 0.0) ? -n_iter : n_iter;
  }
  return NB_OK;
}

int nb_trainer_loss_curve(nb_trainer* t, int32_t net, double* out_host,
                          int32_t max_len) {
  if (net < 0 || net >= t->E) { nb_set_error("bad net index"); return NB_ERR_ARG; }
  const int len = max_len < t->max_iter ? max_len : t->max_iter;
  NB_HIP_CHECK(hipMemcpy(out_host, t->nets_host[net].loss_curve,
                         (size_t)len * sizeof(double), hipMemcpyDeviceToHost));
  return NB_OK;
}

int nb_trainer_weights(nb_trainer* t, int32_t net, double* const* coefs,
                       double* const* icpts) {
  if (net < 0 || net >= t->E) { nb_set_error("bad net index"); return NB_ERR_ARG; }
  std::vector<double> w((size_t)t->n_w);
  NB_HIP_CHECK(hipMemcpy(w.data(), t->nets_host[net].W,
                         (size_t)t->n_w * sizeof(double), hipMemcpyDeviceToHost));
  const int D = t->n_dim;
  const double* w1 = w.data();
  const double* w2 = w1 + (size_t)t->kt1 * NB_HT1 * NB_TILE;
  const double* w3 = w2 + (size_t)NB_HT1 * NB_HT2 * NB_TILE;
  const double* w4 = w3 + (size_t)NB_HT2 * NB_HT3 * NB_TILE;
  for (int k = 0; k < D; ++k)
    for (int h = 0; h < NB_H1; ++h) coefs[0][(size_t)k * NB_H1 + h] = get_w(w1, NB_HT1, k, h);
  for (int h = 0; h < NB_H1; ++h) icpts[0][h] = get_w(w1, NB_HT1, D, h);
  for (int k = 0; k < NB_H1; ++k)
    for (int h = 0; h < NB_H2; ++h) coefs[1][(size_t)k * NB_H2 + h] = get_w(w2, NB_HT2, k, h);
  for (int h = 0; h < NB_H2; ++h) icpts[1][h] = get_w(w2, NB_HT2, NB_H1, h);
  for (int k = 0; k < NB_H2; ++k)
    for (int h = 0; h < NB_H3; ++h) coefs[2][(size_t)k * NB_H3 + h] = get_w(w3, NB_HT3, k, h);
  for (int h = 0; h < NB_H3; ++h) icpts[2][h] = get_w(w3, NB_HT3, NB_H2, h);
  for (int k = 0; k < NB_H3; ++k) coefs[3][k] = get_w(w4, 1, k, 0);
  icpts[3][0] = get_w(w4, 1, NB_H3, 0);
  return NB_OK;
}

#ifdef NB_TRAIN_TIMING
int nb_dbg_train_times(long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_train_ticks), 64 * sizeof(long long));
  long long zero[64] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_train_ticks), zero, sizeof zero);
  return 0;
}
#endif

int nb_trainer_destroy(nb_trainer* t) {
  if (t == nullptr) return NB_OK;
  if (t->pool) (void)hipFree(t->pool);
  g_xcd_in_use &= ~t->xcd_owned;
  if (t->nets_dev) (void)hipFree(t->nets_dev);
  if (t->sync_dev) (void)hipFree(t->sync_dev);
  delete t;
  return NB_OK;
}

}  // extern "C"
