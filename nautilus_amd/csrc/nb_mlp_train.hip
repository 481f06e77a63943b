// Emulator training on the matrix cores: NeuralNetworkEmulator.train ->
// MLPRegressor.fit restated for gfx950 (reference nautilus/neural.py:50-98;
// algorithm: sklearn/neural_network/_multilayer_perceptron.py:620-760
// (_fit_stochastic), :297-389 (_backprop), _stochastic_optimizers.py:255-287
// (Adam), _base.py:187-189 (squared loss)).
//
// All networks of an emulator train concurrently.  An Adam step has two
// phases, separated by a grid-wide synchronisation (barriers in one XCD's L2
// in the resident kernel, kernel boundaries in the two-launch form):
//
//  FB  one workgroup of four wavefronts per 16-row tile of the minibatch:
//      forward through the four layers (the output tiles of a layer are split
//      over the wavefronts, activations / deltas are exchanged through LDS in
//      [unit][row] layout), output delta, backward deltas through W^T (read
//      from transposed copies of the tiles); every weight operand is loaded a
//      layer or more before its use.  Then, with all activations and deltas
//      of the tile still in LDS, the tile's own share of every gradient:
//      dW[k][h] = sum over its 16 rows of act[row][k] delta[row][h], four
//      MFMAs per 16x16 weight tile, written as a PARTIAL tile (16-byte stores)
//      to part[row tile][weight tile].
//  R   all workgroups of the network: every thread owns a pair of weight
//      elements, adds the partials of the row tiles in row-tile order (fixed:
//      deterministic, no atomics; 16-byte loads, perfectly coalesced and
//      evenly spread over the CUs) and applies Adam in place (plus the
//      transposed copy the next backward pass reads).  The bias is row K of
//      the weight matrix (the activations carry a constant 1 in column K).
//
// (Rounds 1-3 wrote activations and deltas to a stash and contracted them
// over the minibatch in a second phase of 2 x 2-tile jobs: 80-106 KB of
// 8-byte, four-lines-per-instruction operand loads per job out of the L2 --
// the phase was bound by exactly those bytes, 13.9 k of a step's 35 k cycles
// at n_dim = 50, and by its slowest job.)
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

typedef double nb_d2 __attribute__((ext_vector_type(2)));

constexpr int MAXB = 208;        // minibatch rows padded to 16 (batch <= 200)
constexpr int LD1 = 112, LD2 = 64, LD3 = 32, LD4 = 16;
constexpr int G_ROWT = MAXB / 16;   // 16-row tiles of a minibatch
// transposed copies of the weight tiles of layers 2-4 (operands of the
// backward products): [ht][kt] tiles, element (hh, kk) = W[16 kt + kk][16 ht + hh]
constexpr int WT2 = 0;
constexpr int WT3 = WT2 + NB_HT2 * NB_HT1 * NB_TILE;
constexpr int WT4 = WT3 + NB_HT3 * NB_HT2 * NB_TILE;
constexpr int WT_DOUBLES = WT4 + 1 * NB_HT3 * NB_TILE;

// Data written by one CU of an XCD and read by another in the same step
// (weights, stash, loss partials, flags) is read with device-scope loads
// (sc1): they never hit the reading CU's L1 and are served by the XCD's L2,
// where the writer's write-through stores are.  (The alternative, dropping
// the L1 behind every barrier with buffer_inv sc1, also invalidates the L2's
// clean lines on this part -- measured: 4.6 us per step.)
__device__ __forceinline__ double ld_xcd(const nb_gd* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Weight tiles (W: [kt][ht], contraction index c = input unit; WT: [ht][kt],
// c = output unit) are stored so that a lane finds the operands of two
// consecutive k-steps side by side: k-step 2 P + j of the tile row covers
// c = 8 P + 2 lg + j (lg = lane / 16), and element (c, r) of a tile lives at
// ((c / 8) * 64 + ((c / 2) % 4) * 16 + r) * 2 + c % 2 -- one 16-byte load per
// lane and PAIR of k-steps, 1 KB contiguous per wavefront.  A CU's texture
// path moves ~26 B / clk with 8-byte and ~53 B / clk with 16-byte loads per
// lane (profiles/tools/l2_read_bench.hip), and the forward / backward phase is made of
// waiting for exactly these operands.
__host__ __device__ constexpr int tile_index(int c, int r) {
  return ((c >> 3) * 64 + ((c >> 1) & 3) * 16 + r) * 2 + (c & 1);
}

// 16-byte device-scope load (buffer_load_dwordx4 ... sc1; the global-pointer
// form of the atomic load builtin stops at 8 bytes): wave-uniform byte offset
// `soff` + per-lane byte offset `voff` into the buffer behind `rsrc`
__device__ __forceinline__ nb_d2 ld_xcd2(__amdgpu_buffer_rsrc_t rsrc,
                                         unsigned voff, unsigned soff) {
  return __builtin_bit_cast(
      nb_d2, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 16));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const nb_gd* p) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff,
                                           0x00020000);
}

struct NetState {
  nb_gd* W; nb_gd* M; nb_gd* V;      // tile-major weights and Adam moments
  nb_gd* WT;                         // transposed tiles of layers 2-4
  nb_gd* part;                       // gradient partials [row tile][weight tile]
  nb_gd* loss_curve;
  nb_gd* scal;    // [0] adam t  [1] best loss  [2] stale  [3] n_iter  [4] done
};

// the training set of one network and its shuffles for the epochs of a launch
struct NetData {
  const nb_gd* X;       // (n, D) standardised inputs
  const nb_gd* y;       // (n)
  const nb_gi* perm;    // (n_epochs, n)
  long long n;
  int batch;            // min(batch size, n)
};

constexpr int MAX_RESIDENT = 16;   // networks of one resident launch
struct FleetData { NetData d[MAX_RESIDENT]; };

struct TrainArgs {
  const NetState* nets;
  // two-launch form: all networks share one training set
  const nb_gd* X;
  const nb_gd* y;
  const nb_gi* perm;    // (E, n_epochs, n)
  long long n;
  int n_dim, kt1, n_epochs, max_iter, n_iter_no_change, batch;
  double tol, lr, b1, b2, eps;
};

__device__ __forceinline__ NetData shared_data(const TrainArgs& a, int net) {
  NetData d;
  d.X = a.X; d.y = a.y;
  d.perm = a.perm + (long long)net * a.n_epochs * a.n;
  d.n = a.n; d.batch = a.batch;
  return d;
}

// ---------------------------------------------------------------------------
// Step = FB then G (the two-launch form runs them as two kernels, the kernel
// boundary being the grid-wide synchronisation; the resident form separates
// them by barriers in the L2 of one XCD).
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
  // k-step ks covers the units 16 (ks / 4) + 8 ((ks / 2) % 2) + 2 lg + ks % 2
  // (tile_index)
#pragma unroll
  for (int ks = 0; ks < NREG; ++ks)
    in[ks] = act[(16 * (ks >> 2) + 8 * ((ks >> 1) & 1) + 2 * lg + (ks & 1)) *
                     LS + li];
}

// A operands of one 16x16 output tile, k-steps 0 .. N-1 (N even): `tile0` is
// the DOUBLE offset of the tile of k-tile 0 in the buffer behind `rsrc`
// (wave-uniform), consecutive k-tiles are TSTRIDE tiles apart; the operands
// of a pair of k-steps are one contiguous 1 KB block (16 bytes per lane).
// The same form serves the forward products (tiles [kt][ht] of W) and,
// through the transposed copy, the backward ones (tiles [ht][kt] of WT).
template <int N, int TSTRIDE>
__device__ __forceinline__ void load_ops(__amdgpu_buffer_rsrc_t rsrc,
                                         unsigned tile0, unsigned lane,
                                         double* wr) {
  static_assert(N % 2 == 0, "operands come in pairs of k-steps");
#pragma unroll
  for (int p = 0; p < N / 2; ++p) {
    const nb_d2 v = ld_xcd2(
        rsrc, lane * 16,
        (tile0 + (unsigned)((p >> 1) * TSTRIDE * NB_TILE + (p & 1) * 128)) * 8);
    wr[2 * p] = v.x;
    wr[2 * p + 1] = v.y;
  }
}

// ... of layer 1: only the pairs of the last k-tile depend on n_dim
template <int KT1>
__device__ __forceinline__ void load_ops_l1(__amdgpu_buffer_rsrc_t rsrc,
                                            unsigned tile0, int ks1,
                                            unsigned lane, double* wr) {
  load_ops<4 * (KT1 - 1), NB_HT1>(rsrc, tile0, lane, wr);
#pragma unroll
  for (int p = 2 * (KT1 - 1); p < 2 * KT1; ++p) {
    nb_d2 v = {0.0, 0.0};
    if (2 * p < ks1)
      v = ld_xcd2(rsrc, lane * 16,
                  (tile0 + (unsigned)((p >> 1) * NB_HT1 * NB_TILE +
                                      (p & 1) * 128)) * 8);
    wr[2 * p] = v.x;
    wr[2 * p + 1] = v.y;
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

// ... with a runtime number of k-steps in the last k-tile (same order of
// summation: even k-steps into one accumulator, odd ones into the other)
template <int N>
__device__ __forceinline__ nb_d4 mma_l1(const double* wr, const double* in,
                                        int ks_n) {
  nb_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
  for (int k = 0; k < N - 4; k += 2) {
    acc0 = MFMA(wr[k], in[k], acc0);
    acc1 = MFMA(wr[k + 1], in[k + 1], acc1);
  }
#pragma unroll
  for (int k = N - 4; k < N; ++k) {
    if (k < ks_n) {
      if (k & 1) acc1 = MFMA(wr[k], in[k], acc1);
      else acc0 = MFMA(wr[k], in[k], acc0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
  return acc0;
}

// The rows of a tile's minibatch slice: permutation entry, the k-steps of the
// input block this wavefront fills (ks % 4 == wave) and the target.  Read-only
// data, so the resident kernel fetches the NEXT step's rows while it waits at
// the barrier that ends the current one (two dependent global latencies off
// the critical path).
template <int KT1>
struct FbRows {
  double x[KT1];
  double yv;
};

// row index of this lane's point in the minibatch slice starting at `start`
// of epoch `ep` (rows past the end of the slice read entry 0 and are masked
// later)
__device__ __forceinline__ int fb_row_index(const NetData& nd, int tile,
                                            int ep, long long start, int nb) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const nb_gi* perm = nd.perm + (long long)ep * nd.n + start;
  const int pt = tile * 16 + (lane & 15);
  return perm[pt < nb ? pt : 0];
}

template <int KT1>
__device__ __forceinline__ void fb_gather(const NetData& nd, int D, int tile,
                                          int nb, int row, FbRows<KT1>& in) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const bool valid = tile * 16 + li < nb;
  const nb_gd* xr = nd.X + (long long)row * D;
#pragma unroll
  for (int j = 0; j < KT1; ++j) {
    const int f = 4 * (4 * j + wave) + lg;
    const double v = xr[f < D ? f : D - 1];
    in.x[j] = (f < D) ? (valid ? v : 0.0) : ((f == D) ? 1.0 : 0.0);
  }
  const double yv = nd.y[row];
  in.yv = (wave == 0 && lg == 0 && valid) ? yv : 0.0;
}

#ifdef NB_TRAIN_TIMING
// In-kernel time stamps (debug build only).  A stamp is s_memtime into LDS --
// no vector-memory instruction, so it does not wait for the loads in flight
// (an earlier form accumulated into global memory at every stamp and thereby
// charged the whole latency of prefetched operands to the stage it sat in);
// workgroup 0 of network 0 folds the differences once per step.
__device__ long long g_train_ticks[64];
__shared__ long long s_ts[48];
#define NB_STAMP(cond, i)                                                     \
  do {                                                                        \
    if ((cond) && threadIdx.x == 0)                                           \
      s_ts[i] = (long long)__builtin_amdgcn_s_memtime();                      \
  } while (0)
#define FB_STAMP(i) NB_STAMP(net == 0 && tile == 0, i)
#else
#define FB_STAMP(i)
#endif

// LDS of a workgroup: the activation / delta blocks of FB in [unit][row]
// layout.
template <int KT1>
struct FbLds {
  static constexpr int LD0 = 16 * KT1;
  static constexpr int A0 = 0;
  static constexpr int A1 = A0 + LD0 * LS;
  static constexpr int A2 = A1 + LD1 * LS;
  static constexpr int A3 = A2 + LD2 * LS;
  static constexpr int D4 = A3 + LD3 * LS;
  static constexpr int D3 = D4 + LD4 * LS;
  static constexpr int D2 = D3 + LD3 * LS;
  static constexpr int D1 = D2 + LD2 * LS;
  static constexpr int TOTAL = D1 + LD1 * LS;
};

// weight tiles of a network in the order of W: layer 1 [kt][ht], layer 2, ...
template <int KT1>
struct TileMap {
  static constexpr int N1 = KT1 * NB_HT1, N2 = NB_HT1 * NB_HT2,
                       N3 = NB_HT2 * NB_HT3, N4 = NB_HT3;
  static constexpr int NT = N1 + N2 + N3 + N4;
};

// One gradient partial: the 16 x 16 tile act_block^T delta_block over the 16
// rows of the workgroup's row tile.  Both blocks sit in LDS as [unit][row]:
// the MFMA operand of k-step s is element (unit = lane % 16, row = 4 s +
// lane / 16) of either.  The tile leaves as two 16-byte stores per lane,
// registers (0, 1) then (2, 3): element (2 h + e) of lane l at
// (h * 64 + l) * 2 + e.
__device__ __forceinline__ void grad_operand(const double* blk, int block,
                                             unsigned lane, double* v) {
  const unsigned li = lane & 15, lg = lane >> 4;
  const double* p = blk + (16 * block + li) * LS + lg;
#pragma unroll
  for (int s = 0; s < 4; ++s) v[s] = p[4 * s];
}
__device__ __forceinline__ void grad_tile(const double* a4, const double* b4,
                                          nb_gd* dst, unsigned lane) {
  nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = MFMA(a4[s], b4[s], acc);
  const nb_d2 lo = {acc[0], acc[1]}, hi = {acc[2], acc[3]};
  *(NB_G nb_d2*)(dst + 2 * lane) = lo;
  *(NB_G nb_d2*)(dst + 128 + 2 * lane) = hi;
}

template <int KT1, bool CHECK_DONE>
__device__ __forceinline__ void fb_body(const TrainArgs& a, const NetState& st,
                                        int net, int tile, int nb,
                                        const FbRows<KT1>& rows, double* lds) {
  constexpr int KS1 = 4 * KT1;
  constexpr int LD0 = 16 * KT1;
  // activations / deltas of the tile in [unit][row] layout
  double* sA0 = lds + FbLds<KT1>::A0;
  double* sA1 = lds + FbLds<KT1>::A1;
  double* sA2 = lds + FbLds<KT1>::A2;
  double* sA3 = lds + FbLds<KT1>::A3;
  double* sD4 = lds + FbLds<KT1>::D4;
  double* sD3 = lds + FbLds<KT1>::D3;
  double* sD2 = lds + FbLds<KT1>::D2;
  double* sD1 = lds + FbLds<KT1>::D1;

  if (CHECK_DONE) {
    if (st.scal[4] != 0.0) return;               // network already stopped
  }
  // (opaque to the optimiser: the per-lane offsets of this function are
  // recomputed every step -- a few VALU instructions -- instead of being
  // hoisted out of the resident kernel's step loop and spilled)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const unsigned lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  // k-steps of layer 1 in use: pairs, 8 input units (bias included) each
  const int ks1 = 2 * ((a.n_dim + 1 + 7) >> 3);

  // tile offsets (doubles) into the weights / their transposed copies
  const __amdgpu_buffer_rsrc_t rW = tile_rsrc(st.W);
  const __amdgpu_buffer_rsrc_t rT = tile_rsrc(st.WT);
  constexpr unsigned W1 = 0;
  constexpr unsigned W2 = W1 + KT1 * NB_HT1 * NB_TILE;
  constexpr unsigned W3 = W2 + NB_HT1 * NB_HT2 * NB_TILE;
  constexpr unsigned W4 = W3 + NB_HT2 * NB_HT3 * NB_TILE;
  constexpr unsigned T2 = WT2, T3 = WT3, T4 = WT4;

  const int pt = tile * 16 + li;
  const bool valid = pt < nb;

  // ---- weight operands: every wavefront loads the A operands of ITS output
  // tiles straight into registers (wave-uniform tile address + lane offset,
  // one contiguous 512-byte row block per operand), a layer or more ahead of
  // their use; nothing else sits in the memory queue in front of them -- the
  // stash stores of the step are issued at the very end. ---------------------
  double w1r[2][KS1], w2r[26], w3r[14], w4r[6], b4r[2], b3r[6], b2r[2][14];
  const int ht1b = (wave + 4 < NB_HT1) ? wave + 4 : wave;
  load_ops_l1<KT1>(rW, W1 + wave * NB_TILE, ks1, lane, w1r[0]);
  load_ops_l1<KT1>(rW, W1 + ht1b * NB_TILE, ks1, lane, w1r[1]);
  if constexpr (KT1 <= 4)
    load_ops<26, NB_HT2>(rW, W2 + wave * NB_TILE, lane, w2r);

  // ---- input block: k-step ks is handled by wavefront ks % 4 -------------
#pragma unroll
  for (int j = 0; j < KT1; ++j)
    sA0[(4 * (4 * j + wave) + lg) * LS + li] = rows.x[j];
  lds_barrier();
  FB_STAMP(11);
  if constexpr (KT1 > 4)
    load_ops<26, NB_HT2>(rW, W2 + wave * NB_TILE, lane, w2r);

  // ---- layer 1: output tiles wave, wave + 4 ------------------------------
  {
    double in[KS1];
    lds_operand<KS1>(sA0, lane, in);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int ht = wave + 4 * rep;
      if (ht < NB_HT1) {
        const nb_d4 acc = mma_l1<KS1>(w1r[rep], in, ks1);
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
  load_ops<14, NB_HT3>(rW, W3 + (wave & 1) * NB_TILE, lane, w3r);
  load_ops<6, 1>(rW, W4, lane, w4r);
  load_ops<2, NB_HT3>(rT, T4 + (wave & 1) * NB_TILE, lane, b4r);
  load_ops<6, NB_HT2>(rT, T3 + wave * NB_TILE, lane, b3r);

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
  load_ops<14, NB_HT1>(rT, T2 + wave * NB_TILE, lane, b2r[0]);
  load_ops<14, NB_HT1>(rT, T2 + ht1b * NB_TILE, lane, b2r[1]);

  // ---- layer 3: two output tiles ------------------------------------------
  if (wave < NB_HT3) {
    double in[14];
    lds_operand<14>(sA2, lane, in);
    const nb_d4 acc = mma<14>(w3r, in);
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

  // ---- output layer, delta 4, loss partial (wavefront 0) -------------------
  double lp = 0.0;
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
    lp = 0.5 * d40 * d40;
    for (int s = 8; s >= 1; s >>= 1) lp += __shfl_xor(lp, s);
  }
  lds_barrier();
  FB_STAMP(15);

  // ---- delta 3 (ReLU mask = activation == 0; bias unit carries none) ------
  if (wave < NB_HT3) {
    double dout[2];
    lds_operand<2>(sD4, lane, dout);
    const nb_d4 acc = mma<2>(b4r, dout);
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

  // ---- delta 2 --------------------------------------------------------------
  {
    double dout[6];
    lds_operand<6>(sD3, lane, dout);
    const nb_d4 acc = mma<6>(b3r, dout);
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

  // ---- delta 1 --------------------------------------------------------------
  {
    double dout[14];
    lds_operand<14>(sD2, lane, dout);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int kt = wave + 4 * rep;
      if (kt < NB_HT1) {
        const nb_d4 acc = mma<14>(b2r[rep], dout);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int unit = 16 * kt + 4 * r + lg;
          double v = acc[r];
          if (sA1[unit * LS + li] == 0.0 || unit == NB_H1) v = 0.0;
          sD1[unit * LS + li] = v;
        }
      }
    }
  }
  lds_barrier();
  FB_STAMP(18);
  // ---- the row tile's share of every gradient (see the file comment); the
  // tiles of a layer are dealt out over the wavefronts so that the block they
  // share is read once: layer 1 / 2 by output tile, layer 3 / 4 by input tile
  {
    using TM = TileMap<KT1>;
    nb_gd* part = st.part + (size_t)tile * TM::NT * NB_TILE;
    double a4[4], b4[4];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int ht = wave + 4 * rep;
      if (ht < NB_HT1) {
        grad_operand(sD1, ht, lane, b4);
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
          grad_operand(sA0, kt, lane, a4);
          grad_tile(a4, b4, part + (kt * NB_HT1 + ht) * NB_TILE, lane);
        }
      }
    }
    grad_operand(sD2, wave, lane, b4);
#pragma unroll
    for (int kt = 0; kt < NB_HT1; ++kt) {
      grad_operand(sA1, kt, lane, a4);
      grad_tile(a4, b4, part + (TM::N1 + kt * NB_HT2 + wave) * NB_TILE, lane);
    }
    grad_operand(sA2, wave, lane, a4);
#pragma unroll
    for (int ht = 0; ht < NB_HT3; ++ht) {
      grad_operand(sD3, ht, lane, b4);
      grad_tile(a4, b4,
                part + (TM::N1 + TM::N2 + wave * NB_HT3 + ht) * NB_TILE, lane);
    }
    if (wave < NB_HT3) {
      grad_operand(sA3, wave, lane, a4);
      grad_operand(sD4, 0, lane, b4);
      grad_tile(a4, b4, part + (TM::N1 + TM::N2 + TM::N3 + wave) * NB_TILE,
                lane);
    }
  }
  if (wave == 0 && lane == 0) st.scal[8 + tile] = lp;
  FB_STAMP(19);
}

template <int KT1>
__global__ void __launch_bounds__(256)
nb_train_fb_kernel(TrainArgs a, int ep, long long start, int nb) {
  __shared__ __attribute__((aligned(16))) double lds[FbLds<KT1>::TOTAL];
  const NetState st = a.nets[blockIdx.y];
  const int tile = (int)blockIdx.x;
  // (row tiles past the end of a short minibatch leave no partial: R adds the
  // partials of the tiles in use only)
  if (tile * 16 >= nb) return;
  for (int i = threadIdx.x; i < FbLds<KT1>::LD0 * LS; i += 256) lds[i] = 0.0;
  __syncthreads();
  FbRows<KT1> rows;
  const NetData nd = shared_data(a, (int)blockIdx.y);
  fb_gather<KT1>(nd, a.n_dim, tile, nb, fb_row_index(nd, tile, ep, start, nb),
                 rows);
  fb_body<KT1, true>(a, st, (int)blockIdx.y, tile, nb, rows, lds);
}

// ---- R: the partials of the row tiles added up, Adam ------------------------
// the step's loss partials folded into the epoch sum, in tile order
// (deterministic); one wavefront, partial i in lane i
__device__ __forceinline__ void loss_fold(const NetState& st, int nb,
                                          int lane) {
  const int n_tiles = (nb + 15) >> 4;
  const double p = (lane < n_tiles) ? ld_xcd(&st.scal[8 + lane]) : 0.0;
  double acc = ld_xcd(&st.scal[5]);
  for (int i = 0; i < n_tiles; ++i) acc += __shfl(p, i);
  if (lane == 0) st.scal[5] = acc;
}

// step size of Adam step t (sklearn _stochastic_optimizers.py:276-279)
__device__ __forceinline__ double adam_lr(const TrainArgs& a, long long t_adam) {
  return a.lr * sqrt(1.0 - pow(a.b2, (double)t_adam)) /
         (1.0 - pow(a.b1, (double)t_adam));
}

#ifdef NB_TRAIN_TIMING
#define G_STAMP(i) NB_STAMP(timed, i)
#else
#define G_STAMP(i)
#endif

// One item = the element pair (2 h, 2 h + 1) of lane l of weight tile tau:
// input units i = l / 16 + 4 (2 h + e), output unit j = l % 16 (the register
// layout of the MFMA result, see grad_tile).  The items of a network are
// dealt out over all threads of its `slots` workgroups; a thread adds the
// partials of its pair in row-tile order and applies Adam (sklearn
// _stochastic_optimizers.py:255-287) to W, M, V and, for the layers 2-4, to
// the transposed copy the next backward pass reads.  `after_loads` runs
// behind the loads of the first item (the resident kernel fetches the next
// step's input rows there).
template <int KT1, class Hook>
__device__ __forceinline__ void reduce_adam(const TrainArgs& a,
                                            const NetState& st, int slot,
                                            int slots, int nb, double lr_t,
                                            Hook&& after_loads,
                                            bool timed = false) {
  using TM = TileMap<KT1>;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));      // (nothing hoisted out of the step loop)
  const int n_rt = (nb + 15) >> 4;    // row tiles of this minibatch
  const double inv_nb = 1.0 / (double)nb;
  const __amdgpu_buffer_rsrc_t rP = tile_rsrc(st.part);
  constexpr int ITEMS = TM::NT * 128;
  constexpr unsigned RT_BYTES = (unsigned)TM::NT * NB_TILE * 8;
  bool hooked = false;
  G_STAMP(33);
  for (int it = slot * 256 + tid_; it < ITEMS; it += slots * 256) {
    const int tau = it >> 7, h = (it >> 6) & 1, ln = it & 63;
    const unsigned voff = (unsigned)(tau * NB_TILE + (h * 64 + ln) * 2) * 8;
    nb_d2 p[G_ROWT];
#pragma unroll
    for (int t = 0; t < G_ROWT; ++t) {
      p[t] = nb_d2{0.0, 0.0};
      if (t < n_rt) p[t] = ld_xcd2(rP, voff, (unsigned)t * RT_BYTES);
    }
    const int li = ln & 15, lg = ln >> 4;
    double w_old[2], m_old[2], v_old[2];
    const nb_gd* W = st.W + tau * NB_TILE;
    const nb_gd* M = st.M + tau * NB_TILE;
    const nb_gd* V = st.V + tau * NB_TILE;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = lg + 4 * (2 * h + e);
      w_old[e] = ld_xcd(&W[tile_index(i, li)]);
      m_old[e] = ld_xcd(&M[i * 16 + li]);
      v_old[e] = ld_xcd(&V[i * 16 + li]);
    }
    if (!hooked) { after_loads(); hooked = true; }
    G_STAMP(30);
    nb_d2 sum = p[0];
#pragma unroll
    for (int t = 1; t < G_ROWT; ++t) sum += p[t];      // (zeros past n_rt)
    G_STAMP(31);
    // transposed copy: tiles [ht][kt] of the layers 2-4
    int tbase = -1;
    if (tau >= TM::N1 + TM::N2 + TM::N3) {
      tbase = WT4 + (tau - (TM::N1 + TM::N2 + TM::N3)) * NB_TILE;
    } else if (tau >= TM::N1 + TM::N2) {
      const int q = tau - (TM::N1 + TM::N2);             // kt * NB_HT3 + ht
      tbase = WT3 + ((q % NB_HT3) * NB_HT2 + q / NB_HT3) * NB_TILE;
    } else if (tau >= TM::N1) {
      const int q = tau - TM::N1;                        // kt * NB_HT2 + ht
      tbase = WT2 + ((q % NB_HT2) * NB_HT1 + q / NB_HT2) * NB_TILE;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = lg + 4 * (2 * h + e);
      const double gr = sum[e] * inv_nb;
      const double m = a.b1 * m_old[e] + (1.0 - a.b1) * gr;
      const double v = a.b2 * v_old[e] + (1.0 - a.b2) * (gr * gr);
      const double w = w_old[e] + -lr_t * m / (sqrt(v) + a.eps);
      (st.M + tau * NB_TILE)[i * 16 + li] = m;
      (st.V + tau * NB_TILE)[i * 16 + li] = v;
      (st.W + tau * NB_TILE)[tile_index(i, li)] = w;
      if (tbase >= 0) (st.WT + tbase)[tile_index(li, i)] = w;
    }
  }
  if (!hooked) after_loads();
  G_STAMP(32);
}

template <int KT1>
__global__ void __launch_bounds__(256)
nb_train_g_kernel(TrainArgs a, int nb, long long t_adam) {
  const NetState st = a.nets[blockIdx.y];
  if (st.scal[4] != 0.0) return;                 // network already stopped
  // the first workgroup also folds the step's loss (the resident kernel gives
  // that to its last workgroup)
  if (blockIdx.x == 0 && threadIdx.x < 64) loss_fold(st, nb, (int)threadIdx.x);
  reduce_adam<KT1>(a, st, (int)blockIdx.x, (int)gridDim.x, nb,
                   adam_lr(a, t_adam), []() {});
}

// end of epoch: loss curve and the stopping rule of _fit_stochastic
// (sklearn/_multilayer_perceptron.py:730-760, 819-822); one thread
__device__ __forceinline__ void epoch_body(const TrainArgs& a,
                                           const NetState& st, long long n,
                                           long long t_adam) {
  if (ld_xcd(&st.scal[4]) != 0.0) return;
  const double loss = ld_xcd(&st.scal[5]) / (double)n;
  int n_iter = (int)ld_xcd(&st.scal[3]);
  double best = ld_xcd(&st.scal[1]);
  int stale = (int)ld_xcd(&st.scal[2]);
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
  if (threadIdx.x == 0) epoch_body(a, st, a.n, t_adam);
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
// time through HW_REG_XCC_ID), so 32 workgroups of a network -- one per CU --
// share one L2.  Within one L2 a producer only has to wait for its stores
// (s_waitcnt) and a consumer to drop its CU's L1 (buffer_inv): no L2
// write-back, and the barrier is one atomic in that L2.
// ---------------------------------------------------------------------------
// s_sleep between two polls of a barrier counter, in units of 64 clocks: 16
// when more than four networks train at once (the polls of eight busy XCDs
// get into each other's way: 3 us per step), 6 for up to four (half a
// microsecond less barrier latency per step)
#ifndef NB_POLL_SLEEP
#define NB_POLL_SLEEP 16
#endif
#ifndef NB_POLL_SLEEP_FEW
#define NB_POLL_SLEEP_FEW 6
#endif
constexpr int XCD_COUNT = 8;
constexpr int XCD_SLOTS = 32;            // workgroups per network: one per CU
constexpr int SYNC_WORDS = 4;            // per network: counter, error, -, -
// (+ one ticket counter per XCD behind the MAX_RESIDENT network records)
constexpr int SYNC_INTS = SYNC_WORDS * 16 + XCD_COUNT;
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

// (Nothing may be in flight in the polling wavefront's vector-memory queue:
// loads return in order, so a poll behind a prefetch would wait for the
// prefetch's whole latency -- the resident kernel issues its read-only
// prefetches inside the phases, not between arrive and wait.  A scalar poll,
// s_load glc, avoids the queue but was measured at several microseconds per
// round trip.)  The barrier behind the poll orders LDS traffic only; no cache
// is invalidated -- the data that crosses CUs is read with ld_xcd.
__device__ __forceinline__ void xcd_wait(int* counter, int* err, int& phase,
                                         int n_wg, bool few) {
  if (threadIdx.x == 0) {
    const int target = (++phase) * n_wg;
    int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (few) __builtin_amdgcn_s_sleep(NB_POLL_SLEEP_FEW);
      else __builtin_amdgcn_s_sleep(NB_POLL_SLEEP);
      if (++spins > SYNC_LIMIT) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  lds_barrier();
}

__device__ __forceinline__ void xcd_barrier(int* counter, int* err, int& phase,
                                            int n_wg, bool few) {
  xcd_arrive(counter);
  xcd_wait(counter, err, phase, n_wg, few);
}

// networks of the XCDs: up to two per XCD (two workgroups per CU), -1 = none
struct XcdMap {
  int n_nets;
  int net[XCD_COUNT][2];
};

__global__ void nb_xcc_probe_kernel(int* out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = (int)(xcc & 15);
  }
}

#ifdef NB_TRAIN_TIMING
#define TR_STAMP(i) NB_STAMP(net == 0 && slot == 0, i)
// order of the stamps within a step of workgroup 0
__device__ const int g_stamp_order[19] = {0, 11, 12, 13, 14, 15, 16, 17,
                                          18, 19, 1, 2, 33, 30, 31, 36, 32,
                                          3, 4};
#else
#define TR_STAMP(i)
#endif

// (two workgroups per CU: the register budget of 256 leaves every CU of an
// owned XCD a free slot, through which the workgroups of OTHER grids -- a
// concurrent trainer's, which leave at once here, or any other kernel's --
// pass while this one is resident)
template <int KT1>
__global__ void __launch_bounds__(256, 2)
nb_train_xcd_kernel(TrainArgs a, FleetData fleet, XcdMap map, int* sync) {
  // concurrent trainers (the neural bounds of a multi-modal NautilusBound)
  // own disjoint XCDs; map.net[x] = network of XCD x or -1
  // The workgroup asks the hardware which XCD it runs on and takes a ticket
  // there: the first XCD_SLOTS arrivals on an XCD this trainer owns are the
  // network's workgroups, everybody else leaves.  (The dispatcher deals the
  // workgroups of a grid out round-robin over the XCDs, 32 each for this
  // grid, but not necessarily starting at XCD 0 when several queues are
  // active.)
  __shared__ int sh_slot;
  __shared__ __attribute__((aligned(16))) double lds[FbLds<KT1>::TOTAL];
  const int n_nets = map.n_nets;
  unsigned xcc_id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
  const int xcd = xcc_id & (XCD_COUNT - 1);
  if (map.net[xcd][0] < 0) return;
  // One ticket counter per XCD.  An XCD with one network gives it all 32
  // arrivals (a workgroup on every CU: the shortest step); an XCD with two
  // networks gives each of them 16 -- on 16 CUs of their own, as long as the
  // dispatcher spreads the 32 workgroups of the grid over the 32 CUs: sharing
  // CUs was measured at twice the step time, i.e. no gain over training one
  // network after the other, because the phases are bound by what a CU gets
  // out of the L2 per clock.  (The CUs keep a free workgroup slot either way.)
  int* ticket = sync + SYNC_WORDS * MAX_RESIDENT + xcd;
  if (threadIdx.x == 0) sh_slot = atomicAdd(ticket, 1);
  __syncthreads();
  const int arrival = __builtin_amdgcn_readfirstlane(sh_slot);
  if (arrival >= XCD_SLOTS) return;
  const bool two = map.net[xcd][1] >= 0;
  const bool few = n_nets <= 4;          // few pollers: poll more often
  const int slots = two ? XCD_SLOTS / 2 : XCD_SLOTS;
  const int which = two ? arrival / slots : 0;
  const int net = map.net[xcd][which];
  if (net < 0 || net >= n_nets) return;
  const int slot = arrival - which * slots;
  int* counter = sync + SYNC_WORDS * net;
  int* err = counter + 1;
  const NetState st = a.nets[net];
  const NetData nd = fleet.d[net];
  int phase = 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long n = nd.n;
  const int batch = nd.batch;
  const int steps = (int)((n + batch - 1) / batch);
  // the zero padding of the input block (rows >= D + 1 of the last k-tile are
  // multiplied by zero weights and must not hold NaN bit patterns); every
  // step rewrites exactly the rows it fills
  for (int i = threadIdx.x; i < FbLds<KT1>::LD0 * LS; i += 256) lds[i] = 0.0;
  xcd_barrier(counter, err, phase, slots, few);
  // the Adam step counter lives with the network (epoch_body keeps it)
  long long t_adam = (long long)ld_xcd(&st.scal[0]);
  FbRows<KT1> rows;
  bool have_rows = false;        // rows = the slice of the step about to run
  int row_next = 0;              // ... and the row index of the step after it
#ifdef NB_TRAIN_TIMING
  bool have_stamps = false;
#endif
  for (int ep = 0; ep < a.n_epochs; ++ep) {
    // uniform over the network's workgroups: the flag only changes in
    // epoch_body, which is followed by a barrier
    const bool done = ld_xcd(&st.scal[4]) != 0.0;
    for (int sidx = 0; sidx < steps; ++sidx) {
      const long long start = (long long)sidx * batch;
      const int nb = (int)((n - start < batch) ? (n - start) : batch);
      t_adam += 1;
      if (done) continue;
#ifdef NB_TRAIN_TIMING
      if (net == 0 && slot == 0 && threadIdx.x == 0) {
        if (have_stamps) {
          long long prev = s_ts[0];
          for (int i = 1; i < 19; ++i) {
            const int k = g_stamp_order[i];
            g_train_ticks[k] += s_ts[k] - prev;
            prev = s_ts[k];
          }
          const long long now = (long long)__builtin_amdgcn_s_memtime();
          g_train_ticks[0] += now - prev;
        }
        have_stamps = true;
      }
#endif
      TR_STAMP(0);
      // the minibatch slice after this one (next epoch's permutation after
      // the last step of an epoch)
      const bool last = sidx + 1 == steps;
      const int ep2 = last ? ep + 1 : ep;
      const long long start2 = last ? 0 : start + batch;
      const int nb2 = (int)((n - start2 < batch) ? (n - start2) : batch);
      const bool next_rows = ep2 < a.n_epochs && slot * 16 < nb2;
      if (slot * 16 < nb) {
        if (!have_rows)
          fb_gather<KT1>(nd, a.n_dim, slot, nb,
                         fb_row_index(nd, slot, ep, start, nb), rows);
        // its row indices are fetched now and consumed after the G phase,
        // where the rows themselves are fetched behind the barrier signal:
        // neither of the two dependent loads is waited for where it is issued
        if (next_rows) row_next = fb_row_index(nd, slot, ep2, start2, nb2);
        fb_body<KT1, false>(a, st, net, slot, nb, rows, lds);
      }
      TR_STAMP(1);
      // (the step size -- two pow() -- is computed while waiting)
      xcd_arrive(counter);
      const double lr_t = adam_lr(a, t_adam);
      xcd_wait(counter, err, phase, slots, few);
      TR_STAMP(2);
      // R: the partials of the row tiles -> gradients -> Adam, the weight
      // elements dealt out over all workgroups of the network (also the ones
      // without a row tile in FB); the last workgroup folds the loss
      if (slot == slots - 1 && wave == 3) loss_fold(st, nb, lane);
      // the rows of the next step (read-only data) are fetched behind the
      // loads of this phase, back long before the barrier
      have_rows = next_rows && slot * 16 < nb;
      reduce_adam<KT1>(a, st, slot, slots, nb, lr_t,
                       [&]() __attribute__((always_inline)) {
                         if (have_rows)
                           fb_gather<KT1>(nd, a.n_dim, slot, nb2, row_next,
                                          rows);
                       },
                       net == 0 && slot == 0);
      TR_STAMP(3);
      xcd_barrier(counter, err, phase, slots, few);
      TR_STAMP(4);
      if (__hip_atomic_load(err, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT) != 0)
        return;
    }
    if (done) continue;
    if (slot == 0 && threadIdx.x == 0) epoch_body(a, st, n, t_adam);
    xcd_barrier(counter, err, phase, slots, few);
  }
}

void put_w(double* tiles, int ht_n, int k, int h, double v) {
  tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE +
        tile_index(k & 15, h & 15)] = v;
}
double get_w(const double* tiles, int ht_n, int k, int h) {
  return tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE +
               tile_index(k & 15, h & 15)];
}
// transposed copy: tiles [ht][kt], element (hh, kk)
void put_wt(double* tiles, int kt_n, int k, int h, double v) {
  tiles[((size_t)(h >> 4) * kt_n + (k >> 4)) * NB_TILE +
        tile_index(h & 15, k & 15)] = v;
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
  const double* X = nullptr;       // network 0's set (all networks', if shared)
  const double* y = nullptr;
  // per network: training set and rows (a fleet: the networks of several
  // ensembles, each with the set of its ensemble)
  std::vector<const double*> Xs, ys;
  std::vector<long long> ns;
  bool shared_set = true;
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
  // nb_trainer_run_async / nb_trainer_wait: the status words of a chunk of
  // epochs travel to pinned host memory behind that chunk, so that the next
  // chunk can be enqueued before the host has seen them
  static constexpr int RING = 4;
  long long per_net = 0;
  double* pin_scal = nullptr;      // RING x E x 8
  int* pin_sync = nullptr;         // RING x SYNC_INTS
  hipEvent_t ring_event[RING] = {};
  long long n_tickets = 0;
};

extern "C" {

int nb_trainer_create_fleet(int32_t n_dim, int32_t n_networks,
                            const int64_t* n_rows_of,
                            const double* const* x_dev_of,
                            const double* const* y_dev_of,
                            const double* const* coefs,
                            const double* const* icpts, nb_trainer** out);

int nb_trainer_create(int32_t n_dim, int32_t n_networks, int64_t n_rows,
                      const double* x_dev, const double* y_dev,
                      const double* const* coefs, const double* const* icpts,
                      nb_trainer** out) {
  if (n_networks < 1) {
    nb_set_error("bad trainer shape (n_networks=%d)", n_networks);
    return NB_ERR_ARG;
  }
  std::vector<int64_t> ns((size_t)n_networks, n_rows);
  std::vector<const double*> xs((size_t)n_networks, x_dev),
      ys((size_t)n_networks, y_dev);
  return nb_trainer_create_fleet(n_dim, n_networks, ns.data(), xs.data(),
                                 ys.data(), coefs, icpts, out);
}

int nb_trainer_create_fleet(int32_t n_dim, int32_t n_networks,
                            const int64_t* n_rows_of,
                            const double* const* x_dev_of,
                            const double* const* y_dev_of,
                            const double* const* coefs,
                            const double* const* icpts, nb_trainer** out) {
  if (n_dim < 1 || n_dim > 16 * NB_MAX_DT || n_networks < 1) {
    nb_set_error("bad trainer shape (n_dim=%d, n_networks=%d)", n_dim,
                 n_networks);
    return NB_ERR_ARG;
  }
  for (int i = 0; i < n_networks; ++i)
    if (n_rows_of[i] < 1) {
      nb_set_error("bad trainer shape (network %d has %lld rows)", i,
                   (long long)n_rows_of[i]);
      return NB_ERR_ARG;
    }
  const int64_t n_rows = n_rows_of[0];
  const double* x_dev = x_dev_of[0];
  const double* y_dev = y_dev_of[0];
  nb_trainer* t = new nb_trainer();
  t->n_dim = n_dim; t->E = n_networks; t->n = n_rows;
  t->dt = (n_dim + 15) / 16;
  t->kt1 = (n_dim + 1 + 15) / 16;
  t->X = x_dev; t->y = y_dev;
  for (int i = 0; i < n_networks; ++i) {
    t->Xs.push_back(x_dev_of[i]); t->ys.push_back(y_dev_of[i]);
    t->ns.push_back(n_rows_of[i]);
    if (x_dev_of[i] != x_dev || y_dev_of[i] != y_dev || n_rows_of[i] != n_rows)
      t->shared_set = false;
  }
  t->n_w = (long long)nb_net_tiles(t->kt1) * NB_TILE;
  // gradient partials: one tile per (row tile of the minibatch, weight tile)
  const long long part = (long long)G_ROWT * nb_net_tiles(t->kt1) * NB_TILE;
  const long long curve = t->max_iter;
  const long long per_net = 3 * t->n_w + WT_DOUBLES + part + curve + 32;
  t->per_net = per_net;
  const size_t bytes = (size_t)per_net * n_networks * sizeof(double);
  hipError_t e = hipMalloc((void**)&t->pool, bytes);
  if (e == hipSuccess) e = hipMemset(t->pool, 0, bytes);
  if (e == hipSuccess)
    e = hipMalloc((void**)&t->nets_dev, n_networks * sizeof(NetState));
  if (e == hipSuccess)
    e = hipMalloc((void**)&t->sync_dev, SYNC_INTS * sizeof(int));
  // (NB_TRAIN_NO_RESIDENT: the library-side switch only, for the test of the
  // host's fallback when the resident kernel is not to be had)
  t->two_launch = n_networks > MAX_RESIDENT ||
                  getenv("NB_TRAIN_TWO_LAUNCH") != nullptr ||
                  getenv("NB_TRAIN_NO_RESIDENT") != nullptr ||
                  !xcd_pinning_available();
  // A resident network takes one workgroup slot on every CU of an XCD (of two
  // per CU).  Every trainer owns its XCDs exclusively -- two resident kernels
  // that each hold a part of an XCD could wait for each other -- and puts
  // one network on each while they last, two beyond that (16 networks on the
  // 8 XCDs); a trainer that finds too few free XCDs trains with two launches
  // per step instead.
  t->xcd_map.n_nets = n_networks;
  for (int x = 0; x < XCD_COUNT; ++x)
    t->xcd_map.net[x][0] = t->xcd_map.net[x][1] = -1;
  if (!t->two_launch) {
    int n_free = 0;
    for (int x = 0; x < XCD_COUNT; ++x)
      if (!(g_xcd_in_use & (1u << x))) ++n_free;
    if (n_networks > 2 * n_free) {
      t->two_launch = true;
    } else {
      // as many XCDs as there are networks (up to the free ones), the
      // networks dealt out round-robin
      const int n_use = n_networks < n_free ? n_networks : n_free;
      int xs_used[XCD_COUNT], k = 0;
      for (int x = 0; x < XCD_COUNT && k < n_use; ++x)
        if (!(g_xcd_in_use & (1u << x))) xs_used[k++] = x;
      for (int i = 0; i < n_networks; ++i)
        t->xcd_map.net[xs_used[i % n_use]][i / n_use] = i;
      for (int j = 0; j < n_use; ++j) t->xcd_owned |= 1u << xs_used[j];
      g_xcd_in_use |= t->xcd_owned;
    }
  }
  if (t->two_launch && !t->shared_set) {
    nb_set_error("a trainer whose networks have different training sets "
                 "needs the resident kernel (at most %d networks, free XCDs, "
                 "NB_TRAIN_TWO_LAUNCH unset)", MAX_RESIDENT);
    nb_trainer_destroy(t);
    return NB_ERR_UNSUPPORTED;
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
    s.W = (nb_gd*)base; s.M = s.W + t->n_w; s.V = s.M + t->n_w;
    s.WT = s.V + t->n_w;
    s.part = s.WT + WT_DOUBLES;
    s.loss_curve = s.part + part;
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
    e = hipMemcpy((void*)s.W, w.data(), (size_t)t->n_w * sizeof(double),
                  hipMemcpyHostToDevice);
    {
      // transposed tiles of layers 2-4 (the backward pass's operands)
      std::vector<double> wt((size_t)WT_DOUBLES, 0.0);
      for (int k = 0; k <= NB_H1; ++k)
        for (int h = 0; h < NB_H2; ++h)
          put_wt(wt.data() + WT2, NB_HT1, k, h, get_w(w2, NB_HT2, k, h));
      for (int k = 0; k <= NB_H2; ++k)
        for (int h = 0; h < NB_H3; ++h)
          put_wt(wt.data() + WT3, NB_HT2, k, h, get_w(w3, NB_HT3, k, h));
      for (int k = 0; k <= NB_H3; ++k)
        put_wt(wt.data() + WT4, NB_HT3, k, 0, get_w(w4, 1, k, 0));
      if (e == hipSuccess)
        e = hipMemcpy((void*)s.WT, wt.data(), wt.size() * sizeof(double),
                      hipMemcpyHostToDevice);
    }
    const double scal0[8] = {0.0, INFINITY, 0.0, 0.0, 0.0, 0.0, 0, 0};
    if (e == hipSuccess)
      e = hipMemcpy((void*)s.scal, scal0, sizeof scal0, hipMemcpyHostToDevice);
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

int nb_trainer_run_fleet(nb_trainer* t, const int32_t* const* perm_dev_of,
                         int32_t n_epochs, int32_t* status_host, void* stream);

int nb_trainer_run(nb_trainer* t, const int32_t* perm_dev, int32_t n_epochs,
                   int32_t* status_host, void* stream) {
  if (!t->shared_set) {
    nb_set_error("nb_trainer_run: the networks have different training sets; "
                 "use nb_trainer_run_fleet");
    return NB_ERR_ARG;
  }
  std::vector<const int32_t*> perms((size_t)t->E);
  for (int i = 0; i < t->E; ++i)
    perms[i] = perm_dev + (size_t)i * n_epochs * t->n;
  return nb_trainer_run_fleet(t, perms.data(), n_epochs, status_host, stream);
}

int nb_trainer_run_fleet(nb_trainer* t, const int32_t* const* perm_dev_of,
                         int32_t n_epochs, int32_t* status_host,
                         void* stream) {
  hipStream_t s = (hipStream_t)stream;
  TrainArgs a;
  a.nets = t->nets_dev; a.X = (const nb_gd*)t->X; a.y = (const nb_gd*)t->y;
  a.perm = (const nb_gi*)perm_dev_of[0];
  a.n = t->n; a.n_dim = t->n_dim; a.kt1 = t->kt1; a.n_epochs = n_epochs;
  a.max_iter = t->max_iter; a.n_iter_no_change = t->n_iter_no_change;
  a.batch = (int)((t->n < t->batch) ? t->n : t->batch);
  a.tol = t->tol; a.lr = t->lr; a.b1 = t->b1; a.b2 = t->b2; a.eps = t->eps;
  const long long n = t->n;
  const int steps_per_epoch = (int)((n + a.batch - 1) / a.batch);
  if (!t->two_launch) {
    FleetData fleet;
    for (int i = 0; i < MAX_RESIDENT; ++i) {
      const int k = i < t->E ? i : 0;
      fleet.d[i].X = (const nb_gd*)t->Xs[k];
      fleet.d[i].y = (const nb_gd*)t->ys[k];
      fleet.d[i].perm = (const nb_gi*)perm_dev_of[k];
      fleet.d[i].n = t->ns[k];
      fleet.d[i].batch = (int)((t->ns[k] < t->batch) ? t->ns[k] : t->batch);
    }
    NB_HIP_CHECK(hipMemsetAsync(t->sync_dev, 0, SYNC_INTS * sizeof(int), s));
    // 32 workgroups per XCD: all of them for its network, or 16 for each of
    // its two
    const dim3 grid(XCD_COUNT * XCD_SLOTS), blk(256);
    switch (t->kt1) {
#define NB_CASE(KT1_)                                                      \
      case KT1_:                                                           \
        hipLaunchKernelGGL(nb_train_xcd_kernel<KT1_>, grid, blk, 0, s, a,  \
                           fleet, t->xcd_map, t->sync_dev);                \
        break;
      NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5)
      NB_CASE(6) NB_CASE(7) NB_CASE(8) NB_CASE(9)
#undef NB_CASE
      default: nb_set_error("n_dim unsupported"); return NB_ERR_UNSUPPORTED;
    }
    t->t_adam += (long long)n_epochs * steps_per_epoch;
    NB_HIP_CHECK(hipGetLastError());
    if (status_host != nullptr)
      return nb_trainer_status(t, status_host, stream);
    return NB_OK;
  }
  // two launches per step: one training set, contiguous shuffles
  for (int i = 1; i < t->E; ++i)
    if (perm_dev_of[i] != perm_dev_of[0] + (size_t)i * n_epochs * t->n) {
      nb_set_error("two-launch training needs the shuffles of all networks "
                   "in one (E, n_epochs, n) array");
      return NB_ERR_ARG;
    }
  for (int ep = 0; ep < n_epochs; ++ep) {
    for (int sidx = 0; sidx < steps_per_epoch; ++sidx) {
      const long long start = (long long)sidx * a.batch;
      const int nb = (int)((n - start < a.batch) ? (n - start) : a.batch);
      // (all G_ROWT row tiles: the ones past the end of a short minibatch
      // clear their delta rows)
      const dim3 gfb(G_ROWT, t->E), gg(XCD_SLOTS, t->E), blk(256);
      t->t_adam += 1;
      switch (t->kt1) {
#define NB_CASE(KT1_)                                                      \
        case KT1_:                                                         \
          hipLaunchKernelGGL(nb_train_fb_kernel<KT1_>, gfb, blk, 0, s, a,  \
                             ep, start, nb);                               \
          hipLaunchKernelGGL(nb_train_g_kernel<KT1_>, gg, blk, 0, s, a,    \
                             nb, t->t_adam);                               \
          break;
        NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5)
        NB_CASE(6) NB_CASE(7) NB_CASE(8) NB_CASE(9)
#undef NB_CASE
        default: nb_set_error("n_dim unsupported"); return NB_ERR_UNSUPPORTED;
      }
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
    int sync[SYNC_INTS];
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

int nb_trainer_run_async(nb_trainer* t, const int32_t* const* perm_dev_of,
                         int32_t n_epochs, void* stream, int64_t* ticket) {
  hipStream_t s = (hipStream_t)stream;
  if (t->pin_scal == nullptr) {
    NB_HIP_CHECK(hipHostMalloc((void**)&t->pin_scal,
                               (size_t)nb_trainer::RING * t->E * 8 *
                                   sizeof(double), hipHostMallocDefault));
    NB_HIP_CHECK(hipHostMalloc((void**)&t->pin_sync,
                               (size_t)nb_trainer::RING * SYNC_INTS *
                                   sizeof(int), hipHostMallocDefault));
    for (int i = 0; i < nb_trainer::RING; ++i)
      NB_HIP_CHECK(hipEventCreateWithFlags(&t->ring_event[i],
                                           hipEventDisableTiming));
  }
  const int rc = nb_trainer_run_fleet(t, perm_dev_of, n_epochs, nullptr, stream);
  if (rc != NB_OK) return rc;
  const int slot = (int)(t->n_tickets % nb_trainer::RING);
  // the scal blocks of all networks sit per_net doubles apart in the pool
  NB_HIP_CHECK(hipMemcpy2DAsync(
      t->pin_scal + (size_t)slot * t->E * 8, 8 * sizeof(double),
      (const void*)t->nets_host[0].scal, (size_t)t->per_net * sizeof(double),
      8 * sizeof(double), (size_t)t->E, hipMemcpyDeviceToHost, s));
  NB_HIP_CHECK(hipMemcpyAsync(t->pin_sync + (size_t)slot * SYNC_INTS,
                              t->sync_dev, SYNC_INTS * sizeof(int),
                              hipMemcpyDeviceToHost, s));
  NB_HIP_CHECK(hipEventRecord(t->ring_event[slot], s));
  *ticket = t->n_tickets++;
  return NB_OK;
}

int nb_trainer_wait(nb_trainer* t, int64_t ticket, int32_t* status_host) {
  // (a slot is reused RING tickets later)
  if (ticket < 0 || ticket >= t->n_tickets ||
      t->n_tickets > ticket + nb_trainer::RING) {
    nb_set_error("nb_trainer_wait: ticket %lld is not in flight (next %lld, "
                 "ring of %d)", (long long)ticket, (long long)t->n_tickets,
                 nb_trainer::RING);
    return NB_ERR_ARG;
  }
  const int slot = (int)(ticket % nb_trainer::RING);
  NB_HIP_CHECK(hipEventSynchronize(t->ring_event[slot]));
  if (!t->two_launch) {
    const int* sync = t->pin_sync + (size_t)slot * SYNC_INTS;
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
  const double* scal = t->pin_scal + (size_t)slot * t->E * 8;
  for (int i = 0; i < t->E; ++i) {
    const int n_iter = (int)scal[8 * i + 3];
    status_host[i] = (scal[8 * i + 4] != 0.0) ? -n_iter : n_iter;
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
  if (t->pin_scal) (void)hipHostFree(t->pin_scal);
  if (t->pin_sync) (void)hipHostFree(t->pin_sync);
  for (int i = 0; i < nb_trainer::RING; ++i)
    if (t->ring_event[i]) (void)hipEventDestroy(t->ring_event[i]);
  delete t;
  return NB_OK;
}

}  // extern "C"
