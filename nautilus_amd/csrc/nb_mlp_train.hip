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
//      from transposed copies of the tiles; weight operands go straight into
//      registers, issued from inside the MFMA chains of the stages before
//      their use (mma_hook); the small layers run on 4-unit sub-tiles
//      (v_mfma_f64_4x4x4_4b); activations and deltas go to a stash in global
//      memory (L2 resident, ~0.8 MB per network; stash_index).
//  G   one workgroup of four wavefronts per 16x16 weight tile: dW = act^T delta
//      over the rows of the minibatch, wavefront q contracting two k-steps of
//      every other 16-row tile; the four partial tiles meet in LDS, are added in a
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
#include <algorithm>
#include <vector>

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

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
  nb_gd* stash;                      // A0 A1 A2 A3 D1 D2 D3 D4
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

// A pointer every lane holds the same value of, moved to scalar registers:
// the record of a network is loaded through vector registers (its index is
// uniform but not provably so), and every address derived from it -- hoisted
// out of the step loop -- then lived in vector registers or, beyond 256 of
// them, in scratch, with a wait for the whole memory queue at every reload.
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const unsigned long long b = (unsigned long long)p;
  // (the builtin returns int: through unsigned, or the low word sign-extends)
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b);
  const unsigned hi =
      (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return (T*)((unsigned long long)lo | ((unsigned long long)hi << 32));
}

constexpr int MAX_RESIDENT = 16;   // networks of one resident launch
struct FleetData { NetData d[MAX_RESIDENT]; };

struct TrainArgs {
  const NetState* nets;
  // two-launch form: all networks share one training set
  const nb_gd* X;
  const nb_gd* y;
  const nb_gi* perm;    // (E, n_epochs, n)
  const nb_gi* jobs;    // G phase: n_jobs records of G_JOB_INTS ints
  const nb_gi* sched;   // resident kernel, 32 workgroups per network: the
                        // (early, late) job of every workgroup -- indices
                        // into sched_jobs -- or null
  const nb_gi* sched_jobs;
  int n_jobs;
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

// v of the lane n places further on in its row of 16 lanes (row_ror: a DPP
// operand modifier, no LDS traffic -- __shfl_xor is ds_bpermute and sits in
// front of the next workgroup barrier's counter wait)
template <int N>
__device__ __forceinline__ double row_ror(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
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
  // ALL reads leave before the stage's first MFMA.  (Left alone the scheduler
  // sinks every read to one MFMA ahead of its use to shorten live ranges: 64
  // cycles of cover for an LDS latency of 100-130, and the chains of the
  // layers 1, 2 and delta 1 -- 26-28 MFMAs -- ran at ~115 cycles per MFMA.)
  __builtin_amdgcn_sched_barrier(0);
}

// The stash of one matrix (activations or deltas of a layer, `ld` units padded
// to 16) holds the 16-row tile rt of the minibatch at rt * 16 * ld, and in it
// element (unit 16 ut + li, row 8 h + 2 lg + j) at
//   ((((ut * 2 + h) * 4 + lg) * 16 + li) * 2 + j
// -- the two rows a lane of the G phase feeds to two consecutive MFMAs of its
// chain side by side: one 16-byte load per lane and PAIR of k-steps, 1 KB
// contiguous per wavefront (what the weight operands of FB already do; with
// 8-byte loads a CU got 11-16 B / clk out of its L2 for these columns).
__host__ __device__ constexpr int stash_index(int unit, int row) {
  return (((((unit >> 4) * 2 + (row >> 3)) * 4 + ((row >> 1) & 3)) * 16 +
           (unit & 15)) * 2 + (row & 1));
}

// cooperative LDS [unit][row] -> global stash (16 bytes per thread, 1 KB
// contiguous per wavefront: wavefront w of iteration it writes half w % 2 of
// unit tile w / 2 + 2 it; n_unit is a multiple of 16, so half the wavefronts
// skip the last iteration where it is a multiple of 16 only)
__device__ __forceinline__ void flush_stash(const double* act, nb_gd* dst,
                                            int ld, int n_unit, int tile,
                                            int tid) {
  const unsigned li = tid & 15, lg = (tid >> 4) & 3, h = (tid >> 6) & 1;
  const unsigned ut = tid >> 7;
  nb_gd* row = dst + (tile * 16) * ld;               // wave-uniform
  const unsigned off = (((ut * 2 + h) * 4 + lg) * 16 + li) * 2;
  const unsigned src = (16 * ut + li) * LS + 8 * h + 2 * lg;
  for (int u = 0; u < n_unit; u += 32) {
    if (u + 16 * (int)ut < n_unit) {
      const nb_d2 v = {act[src + u * LS], act[src + u * LS + 1]};
      *(NB_G nb_d2*)(row + off + u * 16) = v;
    }
  }
}

// ... the same block by ONE wavefront (2 * N_UNIT / 16 stores of 1 KB): the
// blocks of the upper layers leave while their wavefront has nothing to
// compute (fb_body), so that the G jobs of the layers 2-4 can start before the
// backward pass has ended
template <int N_UNIT>
__device__ __forceinline__ void flush_wave(const double* act, nb_gd* dst,
                                           int ld, int tile, unsigned lane) {
  const unsigned li = lane & 15, lg = lane >> 4;
  nb_gd* row = dst + (tile * 16) * ld + lane * 2;    // stash_index
  const double* src = act + li * LS + 2 * lg;
  // the LDS reads of up to eight stores leave together (left to the
  // scheduler every store waited for its own read: one LDS latency per KB)
  constexpr int NC = N_UNIT / 8, CH = 8;
#pragma unroll
  for (int c0 = 0; c0 < NC; c0 += CH) {
    nb_d2 v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {                   // c = 2 ut + h
      const int c = c0 + i;
      if (c < NC)
        v[i] = nb_d2{src[(c >> 1) * 16 * LS + (c & 1) * 8],
                     src[(c >> 1) * 16 * LS + (c & 1) * 8 + 1]};
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int c = c0 + i;
      if (c < NC) *(NB_G nb_d2*)(row + c * 128) = v[i];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
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

// Pair p of the operands above, of the full tile (sub < 0) or of ONE 4-unit
// sub-tile of it (output rows 4 sub .. 4 sub + 3): the
// operand of v_mfma_f64_4x4x4_4b, whose four blocks are the four groups of
// four points of the 16-row tile and all take the same A -- replicated by the
// load (lane = point group * 4 + row, k = lane / 16; the instruction has no
// broadcast of its own for f64: profiles/tools/mfma_map.hip).  One such
// instruction takes 16 cycles, a 16x16x4 one 64, and its result is register
// `sub` of the tile's accumulator BIT FOR BIT (profiles/tools/
// mfma_subtile.hip), so the small layers -- 20 + 1 units in layer 3, one
// output -- run on the sub-tiles that exist instead of on padded tiles.
template <int TSTRIDE>
__device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t rsrc,
                                          unsigned tile0, unsigned lane,
                                          int sub, int p, double* wr) {
  const unsigned voff =
      sub < 0 ? lane * 16 : ((lane >> 4) * 16 + 4 * sub + (lane & 3)) * 16;
  const nb_d2 v = ld_xcd2(
      rsrc, voff,
      (tile0 + (unsigned)((p >> 1) * TSTRIDE * NB_TILE + (p & 1) * 128)) * 8);
  wr[2 * p] = v.x;
  wr[2 * p + 1] = v.y;
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

// ... with `hook(p)` behind the p-th pair of MFMAs, pinned there: the operand
// loads of LATER stages are issued from inside the chain, one or two per 128
// cycles of matrix work.  (In front of the chain their issue delays its
// first MFMA; behind it, the arrival at the stage's barrier; and wherever the
// source puts them the scheduler moves them to the end of the chain unless
// told otherwise.)
template <int N, class Hook>
__device__ __forceinline__ nb_d4 mma_hook(const double* wr, const double* in,
                                          Hook&& hook) {
  static_assert(N % 2 == 0, "pairs");
  nb_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
  for (int k = 0; k < N; k += 2) {
    acc0 = MFMA(wr[k], in[k], acc0);
    acc1 = MFMA(wr[k + 1], in[k + 1], acc1);
    __builtin_amdgcn_sched_barrier(0);
    hook(k >> 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) acc0[r] += acc1[r];
  return acc0;
}

// ... of a sub-tile (same order of summation)
template <int N>
__device__ __forceinline__ double mma4(const double* wr, const double* in) {
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int k = 0; k + 1 < N; k += 2) {
    acc0 = MFMA4(wr[k], in[k], acc0);
    acc1 = MFMA4(wr[k + 1], in[k + 1], acc1);
  }
  if (N & 1) acc0 = MFMA4(wr[N - 1], in[N - 1], acc0);
  return acc0 + acc1;
}

// ... of layer 1: a runtime number of k-steps in the last k-tile (same order
// of summation: even k-steps into one accumulator, odd ones into the other),
// a hook behind every pair (mma_hook)
template <int N, class Hook>
__device__ __forceinline__ nb_d4 mma_l1_hook(const double* wr, const double* in,
                                             int ks_n, bool run, Hook&& hook) {
  nb_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
  // (a wavefront without this tile only issues the loads: run == false)
  if (!run) ks_n = 0;
#pragma unroll
  for (int k = 0; k < N - 4; k += 2) {
    if (run) {
      acc0 = MFMA(wr[k], in[k], acc0);
      acc1 = MFMA(wr[k + 1], in[k + 1], acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
    hook(k >> 1);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = N - 4; k < N; ++k) {
    if (k < ks_n) {
      if (k & 1) acc1 = MFMA(wr[k], in[k], acc1);
      else acc0 = MFMA(wr[k], in[k], acc0);
    }
    if (k & 1) {
      __builtin_amdgcn_sched_barrier(0);
      hook(k >> 1);
      __builtin_amdgcn_sched_barrier(0);
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

// (The loads are unconditional -- clamped addresses -- and their values leave
// this function RAW: the masks (feature past n_dim, bias column, row past the
// minibatch) are applied by fb_body where the block goes to LDS.  A select on
// a loaded value here was turned into a branch around the load with a wait for
// the whole memory queue behind it: five dependent round trips per step, each
// behind all the operand loads of the G phase in which the prefetch sits.)
template <int KT1>
__device__ __forceinline__ void fb_gather(const NetData& nd, int D, int tile,
                                          int nb, int row, FbRows<KT1>& in) {
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const nb_gd* xr = nd.X + (long long)row * D;
#pragma unroll
  for (int j = 0; j < KT1; ++j) {
    const int f = 4 * (4 * j + wave) + lg;
    in.x[j] = xr[f < D ? f : D - 1];
  }
  in.yv = nd.y[row];
}

// the input block entry of k-tile j from the raw gathered value: feature f of
// a valid row, 1 in the bias column, 0 elsewhere (v * 1 + 0 and v * 0 + c are
// exact for the finite v of a standardised training set)
__device__ __forceinline__ double fb_input(double v, int f, int D, bool valid) {
  const double m = (f < D && valid) ? 1.0 : 0.0;
  const double c = (f == D) ? 1.0 : 0.0;
  return __builtin_fma(v, m, c);
}

#ifdef NB_TRAIN_TIMING
// In-kernel time stamps (debug build only).  A stamp is s_memtime into LDS --
// no vector-memory instruction, so it does not wait for the loads in flight
// (an earlier form accumulated into global memory at every stamp and thereby
// charged the whole latency of prefetched operands to the stage it sat in);
// workgroup 0 of network 0 folds the differences once per step.
__device__ long long g_train_ticks[64];
// per workgroup of network 0 (NB_TRAIN_SLOT_TIMING: every workgroup keeps four
// time stamps per step in LDS and adds its sums here when the launch ends):
// [0] last barrier open -> arrival at the barrier behind FB (FB, or the early
// job), [1] ... -> that barrier seen open, [2] ... -> arrival at the last
// barrier (late job), [3] ... -> seen open, [4] steps
__device__ long long g_slot_ticks[32 * 8];
__shared__ long long s_ts[48];
__shared__ long long s_slot[8];
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
// layout; the G phase reuses everything behind the input block for its
// partial tiles.
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
  static constexpr int G_RED = A1;             // 1024 doubles per tile
};

struct StashPtrs {
  nb_gd *A0, *A1, *A2, *A3, *D1, *D2, *D3, *D4;
};
__device__ __forceinline__ StashPtrs stash_ptrs(const NetState& st, int ld0) {
  StashPtrs p;
  p.A0 = st.stash;
  p.A1 = p.A0 + MAXB * ld0;
  p.A2 = p.A1 + MAXB * LD1;
  p.A3 = p.A2 + MAXB * LD2;
  p.D1 = p.A3 + MAXB * LD3;
  p.D2 = p.D1 + MAXB * LD1;
  p.D3 = p.D2 + MAXB * LD2;
  p.D4 = p.D3 + MAXB * LD3;
  return p;
}

// `upper_ready` is called by wavefront 3 once the activations of the layers
// 1-3 and the deltas of the layers 2-4 of this tile are in the L2 (all of the
// stash except the layer-1 deltas).
template <int KT1, bool CHECK_DONE, class Hook>
__device__ __forceinline__ void fb_body(const TrainArgs& a, const NetState& st,
                                        int net, int tile, int nb,
                                        const FbRows<KT1>& rows, double* lds,
                                        Hook&& upper_ready) {
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
  const StashPtrs sp = stash_ptrs(st, LD0);

  const int pt = tile * 16 + li;
  const bool valid = pt < nb;

  // ---- weight operands: every wavefront loads the A operands of ITS output
  // tiles straight into registers (wave-uniform tile address + lane offset,
  // one contiguous 512-byte row block per operand).  A CU takes one such 1 KB
  // load per ~16-20 cycles and the first data is ~1.5 k cycles away, so WHERE
  // the ~70 loads of a wavefront are issued is what the pass costs: up here
  // only those of layer 1's first tile (all 29 of layer 1 and 2, as until
  // round 5, took ~1.8 k cycles to issue with the input block waiting behind
  // them); everything else leaves from inside the MFMA chains (mma_hook), a
  // stage or more ahead of its use, and never behind a stage's products,
  // where the issue time went straight into the arrival at the barrier.
  // Unconditionally in every wavefront: behind a branch the compiler no
  // longer knows how many loads are in flight and the next wait is for all of
  // them.  The stash stores of the step go last. ------------------------------
  double w1r[2][KS1], w2r[26], w3r[2][14], w4r[6], b4r[2][2], b3r[6], b2r[2][14];
  // 4-unit sub-tiles of layer 3 (and of delta 3): NB_H3 units, the bias unit
  // behind them is a constant; wavefront w takes the sub-tiles w, w + 4
  constexpr int NSUB3 = (NB_H3 + 3) / 4;
  // unit tiles of the layer-1 activations that leave during layer 3 (the
  // others during the output layer)
  constexpr int L1_SPLIT = 3;
  static_assert(NSUB3 <= 8 && 4 * NSUB3 <= LD3, "two sub-tiles per wavefront");
  const int ht1b = (wave + 4 < NB_HT1) ? wave + 4 : wave;
  load_ops_l1<KT1>(rW, W1 + wave * NB_TILE, ks1, lane, w1r[0]);
  __builtin_amdgcn_sched_barrier(0);

  // ---- input block: k-step ks is handled by wavefront ks % 4 -------------
#pragma unroll
  for (int j = 0; j < KT1; ++j)
    sA0[(4 * (4 * j + wave) + lg) * LS + li] =
        fb_input(rows.x[j], 4 * (4 * j + wave) + lg, a.n_dim, valid);
  lds_barrier();
  FB_STAMP(11);

  // ---- layer 1: output tiles wave, wave + 4.  The second tile's operands
  // leave from the first half of the first chain, layer 2's from there on to
  // the end of the second (a wavefront without a second tile only issues
  // them) -- also the pairs past n_dim's last k-step, which no MFMA uses.
  // (Measured alternatives, profiles/r05/train_fb_schedule.txt: the second
  // tile's operands up front with the first's; layer 2's all in the first
  // chain: 1-2 % slower.) ------------------------------------------------------
  {
    double in[KS1];
    lds_operand<KS1>(sA0, lane, in);
#ifdef NB_W1B_BEHIND_INPUT
    // second tile's operands while the first chain waits for its own
#pragma unroll
    for (int i = 0; i < KS1 / 2; ++i)
      load_pair<NB_HT1>(rW, W1 + ht1b * NB_TILE, lane, -1, i, w1r[1]);
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int ht = wave + 4 * rep;
      const bool run = ht < NB_HT1;
#ifdef NB_W1B_BEHIND_INPUT
      constexpr int NS = KS1 / 2, H = 0, NW = 2 * NS - H;
#else
      constexpr int NS = KS1 / 2, H = (NS + 1) / 2, NW = 2 * NS - H;
#endif
      const nb_d4 acc = mma_l1_hook<KS1>(w1r[rep], in, ks1, run, [&](int p) {
        const int q = rep * NS + p;
        if (q < H) {
#pragma unroll
          for (int i = NS * q / (H ? H : 1); i < NS * (q + 1) / (H ? H : 1); ++i)
            load_pair<NB_HT1>(rW, W1 + ht1b * NB_TILE, lane, -1, i, w1r[1]);
        } else {
#pragma unroll
          for (int i = 13 * (q - H) / NW; i < 13 * (q - H + 1) / NW; ++i)
            load_pair<NB_HT2>(rW, W2 + wave * NB_TILE, lane, -1, i, w2r);
        }
      });
      if (run) {
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
  // ---- layer 2: output tile = wave.  The operands of the stages that
  // follow are loaded from inside its chain: layer 3's sub-tiles in the first
  // seven pairs, the output layer's and those of delta 3 / delta 2 behind
  // them, into registers the chain has left.  (UNCONDITIONAL loads -- a
  // wavefront without a second sub-tile loads the last one again: behind a
  // branch the compiler no longer knows how many loads are in flight and
  // makes the next MFMA wait for all of them.) ----------------------------------
  {
    const int sb0 = wave, sb1 = (wave + 4 < NSUB3) ? wave + 4 : NSUB3 - 1;
    double in[26];
    lds_operand<26>(sA1, lane, in);
    const nb_d4 acc = mma_hook<26>(w2r, in, [&](int p) {
      if (p < 7) {
        load_pair<NB_HT3>(rW, W3 + (sb0 >> 2) * NB_TILE, lane, sb0 & 3, p,
                          w3r[0]);
        load_pair<NB_HT3>(rW, W3 + (sb1 >> 2) * NB_TILE, lane, sb1 & 3, p,
                          w3r[1]);
      } else if (p < 10) {
        load_pair<1>(rW, W4, lane, 0, p - 7, w4r);
        load_pair<NB_HT2>(rT, T3 + wave * NB_TILE, lane, -1, p - 7, b3r);
      } else if (p == 10) {
        load_pair<NB_HT3>(rT, T4 + (sb0 >> 2) * NB_TILE, lane, sb0 & 3, 0,
                          b4r[0]);
        load_pair<NB_HT3>(rT, T4 + (sb1 >> 2) * NB_TILE, lane, sb1 & 3, 0,
                          b4r[1]);
      }
    });
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

  // ---- layer 3: two output tiles; the other two wavefronts send the blocks
  // that are complete to the stash (their weight operands are all older than
  // these stores in the in-order memory queue: nothing ever waits for them) ---
  {
    double in[14];
    lds_operand<14>(sA2, lane, in);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int sb = wave + 4 * rep;
      if (sb < NSUB3) {
        double v = fmax(mma4<14>(w3r[rep], in), 0.0);
        const int unit = 4 * sb + lg;
        if (unit == NB_H3) v = 1.0;
        sA3[unit * LS + li] = v;
      }
    }
    // the units behind the sub-tiles: the bias unit and the padding
    if (wave == 3) {
#pragma unroll
      for (int unit = 4 * NSUB3 + lg; unit < LD3; unit += 4)
        sA3[unit * LS + li] = (unit == NB_H3) ? 1.0 : 0.0;
    }
  }
  // (behind their sub-tile: the wavefronts with one sub-tile have the time
  // of wavefront 0's second one)
  if (wave == 1) flush_wave<LD0>(sA0, sp.A0, LD0, tile, lane);
  if (wave == 2) flush_wave<LD2>(sA2, sp.A2, LD2, tile, lane);
  if (wave == 3) flush_wave<L1_SPLIT * 16>(sA1, sp.A1, LD1, tile, lane);
  lds_barrier();
  FB_STAMP(14);

  // ---- output layer, delta 4, loss partial (wavefront 0) -------------------
  double lp = 0.0;
  if (wave == 1) flush_wave<LD3>(sA3, sp.A3, LD3, tile, lane);
  if (wave == 2)
    flush_wave<LD1 - L1_SPLIT * 16>(sA1 + L1_SPLIT * 16 * LS,
                                    sp.A1 + L1_SPLIT * 256, LD1, tile, lane);
  if (wave == 0) {
    double in[6];
    lds_operand<6>(sA3, lane, in);
    const double acc = mma4<6>(w4r, in);
    double d40 = 0.0;
    if (lg == 0 && valid) d40 = acc - rows.yv;       // sklearn :365
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int unit = 4 * r + lg;
      const double v = (unit == 0) ? d40 : 0.0;
      sD4[unit * LS + li] = v;
    }
    lp = 0.5 * d40 * d40;
    // sum over the 16 rows of the tile in the order of the xor butterfly
    // (8, 4, 2, 1): after the step of distance d the values have period d in
    // the row, so the lane d places further on holds what lane ^ d holds
    lp += row_ror<8>(lp);
    lp += row_ror<4>(lp);
    lp += row_ror<2>(lp);
    lp += row_ror<1>(lp);
  }
  // the operands of delta 1's first tile, three short stages ahead of their
  // use and behind this stage's work of every wavefront.  (A CU takes one 1 KB
  // load per ~20 cycles: the 14 loads of all four wavefronts behind layer 3,
  // where they were, made that stage wait for their issue.)
  __builtin_amdgcn_sched_barrier(0);
  load_ops<14, NB_HT1>(rT, T2 + wave * NB_TILE, lane, b2r[0]);
  __builtin_amdgcn_sched_barrier(0);
  lds_barrier();
  FB_STAMP(15);

  // ---- delta 3 (ReLU mask = activation == 0; bias unit carries none) ------
  {
    double dout[2];
    lds_operand<2>(sD4, lane, dout);
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int sb = wave + 4 * rep;
      if (sb < NSUB3) {
        const int unit = 4 * sb + lg;
        double v = mma4<2>(b4r[rep], dout);
        if (sA3[unit * LS + li] == 0.0 || unit == NB_H3) v = 0.0;
        sD3[unit * LS + li] = v;
      }
    }
    if (wave == 3) {
#pragma unroll
      for (int unit = 4 * NSUB3 + lg; unit < LD3; unit += 4)
        sD3[unit * LS + li] = 0.0;
    }
  }
  if (wave == 2) flush_wave<LD4>(sD4, sp.D4, LD4, tile, lane);
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
  // (every wavefront's stash stores so far have arrived: issued stages ago)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  FB_STAMP(17);

  // ---- delta 1; wavefront 3 -- one output tile here, the others two -- sends
  // the last two upper blocks first, waits for their acknowledgement and
  // reports the upper stash BEFORE its tile: with the counters near (~500
  // ticks per round trip) that fits into the time the others need for their
  // second tile, and the early jobs start ~1.2 k ticks sooner (with the
  // counters far away this order made wavefront 3 the last of the stage) ------
  // (the second tile's operands: its chain starts a tile's chain from here.
  // In front of the wait for the stores that wait would be for them; between
  // that wait and the barrier: measured no better)
  load_ops<14, NB_HT1>(rT, T2 + ht1b * NB_TILE, lane, b2r[1]);
  __builtin_amdgcn_sched_barrier(0);
  if (wave == 3) {
    flush_wave<LD3>(sD3, sp.D3, LD3, tile, lane);
    flush_wave<LD2>(sD2, sp.D2, LD2, tile, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    upper_ready();
  }
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
  // ---- the layer-1 deltas, the only block of the stash still to go.  (An
  // earlier form wrote every block as soon as it was complete from ALL
  // wavefronts: the stores then sat in front of weight operands in the in-order
  // memory queue and every stage waited for a store acknowledgement; the
  // wavefronts that flush now have issued their last operand load before.)
  flush_stash(sD1, sp.D1, LD1, LD1, tile, tid);
  if (wave == 0 && lane == 0) st.scal[8 + tile] = lp;
  FB_STAMP(19);
}

// A 16-row tile past the end of a short minibatch (the last step of an
// epoch): its delta rows are cleared, so that G can contract all G_ROWT row
// tiles of the stash without looking at the batch size (stale activations
// times zero deltas).
__device__ __forceinline__ void fb_clear_deltas(const NetState& st, int ld0,
                                                int tile) {
  const StashPtrs sp = stash_ptrs(st, ld0);
  const unsigned tid = threadIdx.x;
  nb_gd* d1 = sp.D1 + tile * 16 * LD1;
  nb_gd* d2 = sp.D2 + tile * 16 * LD2;
  nb_gd* d3 = sp.D3 + tile * 16 * LD3;
  nb_gd* d4 = sp.D4 + tile * 16 * LD4;
  for (unsigned i = tid; i < 16 * LD1; i += 256) d1[i] = 0.0;
  for (unsigned i = tid; i < 16 * LD2; i += 256) d2[i] = 0.0;
  for (unsigned i = tid; i < 16 * LD3; i += 256) d3[i] = 0.0;
  if (tid < 16 * LD4) d4[tid] = 0.0;
}

template <int KT1>
__global__ void __launch_bounds__(256)
nb_train_fb_kernel(TrainArgs a, int ep, long long start, int nb) {
  __shared__ __attribute__((aligned(16))) double lds[FbLds<KT1>::TOTAL];
  const NetState st = a.nets[blockIdx.y];
  const int tile = (int)blockIdx.x;
  if (tile * 16 >= nb) {
    if (st.scal[4] == 0.0) fb_clear_deltas(st, 16 * KT1, tile);
    return;
  }
  for (int i = threadIdx.x; i < FbLds<KT1>::LD0 * LS; i += 256) lds[i] = 0.0;
  __syncthreads();
  FbRows<KT1> rows;
  const NetData nd = shared_data(a, (int)blockIdx.y);
  fb_gather<KT1>(nd, a.n_dim, tile, nb, fb_row_index(nd, tile, ep, start, nb),
                 rows);
  fb_body<KT1, true>(a, st, (int)blockIdx.y, tile, nb, rows, lds, []() {});
}

// ---- G: dW of 16x16 weight tiles over the minibatch + Adam ------------------
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

// (hi + lo) *= b for an unevaluated sum hi + lo: the product hi * b exactly
// (fma), the rest in working precision, renormalised
__device__ __forceinline__ void dd_scale(double& hi, double& lo, double b) {
  const double p = hi * b;
  const double e = __builtin_fma(hi, b, -p) + lo * b;
  hi = p + e;
  lo = e - (hi - p);
}

// step size of Adam step t (sklearn _stochastic_optimizers.py:276-279)
__device__ __forceinline__ double adam_lr(const TrainArgs& a, long long t_adam) {
  return a.lr * sqrt(1.0 - pow(a.b2, (double)t_adam)) /
         (1.0 - pow(a.b1, (double)t_adam));
}

// A job of the G phase: a block of nk x nh (each 1 or 2, or 3 x 1) weight tiles
// of one layer -- k-tiles kt0 .. kt0 + nk - 1, output tiles ht0 .. ht0 + nh - 1 --
// whose gradients share their operand columns: the nk activation column
// blocks and the nh delta column blocks are read once for the nk * nh tiles.
// (A CU gets 26 bytes per clock out of the L2 with 8-byte loads per lane when
// everything misses its L1, as it does behind a barrier
// (profiles/tools/l2_read_bench.hip): the bytes a workgroup pulls per step are
// what the phase costs.  One tile per job reads two column blocks per tile, a
// 2 x 2 block one.)  The job list is built by the host (g_jobs) so that one
// round of 32 workgroups covers a network for every n_dim.
constexpr int G_JOB_INTS = 5;     // layer (0..3), kt0, nk, ht0, nh
constexpr int G_MAX_TILES = 4;

struct GLayer {
  const nb_gd* as;    // activations of the layer's input  (rows x lda)
  const nb_gd* bs;    // deltas of the layer's output      (rows x ldb)
  int lda, ldb;
  int ht_n, kt_n;     // tiles of the layer
  int wbase;          // offset of the layer's tiles in W / M / V
  int tbase;          // offset of its transposed tiles in WT, -1 for layer 1
};

// (wave-uniform: scalar registers)
__device__ __forceinline__ GLayer g_layer(const NetState& st, int kt1,
                                          int layer) {
  const StashPtrs sp = stash_ptrs(st, 16 * kt1);
  const int n_gt1 = kt1 * NB_HT1, n_gt2 = NB_HT1 * NB_HT2,
            n_gt3 = NB_HT2 * NB_HT3;
  GLayer g;
  if (layer == 0) {
    g.as = sp.A0; g.lda = 16 * kt1; g.bs = sp.D1; g.ldb = LD1;
    g.ht_n = NB_HT1; g.kt_n = kt1; g.wbase = 0; g.tbase = -1;
  } else if (layer == 1) {
    g.as = sp.A1; g.lda = LD1; g.bs = sp.D2; g.ldb = LD2;
    g.ht_n = NB_HT2; g.kt_n = NB_HT1; g.wbase = n_gt1 * NB_TILE;
    g.tbase = WT2;
  } else if (layer == 2) {
    g.as = sp.A2; g.lda = LD2; g.bs = sp.D3; g.ldb = LD3;
    g.ht_n = NB_HT3; g.kt_n = NB_HT2; g.wbase = (n_gt1 + n_gt2) * NB_TILE;
    g.tbase = WT3;
  } else {
    g.as = sp.A3; g.lda = LD3; g.bs = sp.D4; g.ldb = LD4;
    g.ht_n = 1; g.kt_n = NB_HT3;
    g.wbase = (n_gt1 + n_gt2 + n_gt3) * NB_TILE;
    g.tbase = WT4;
  }
  return g;
}

// one 16-unit column block of a stash matrix (stash_index): wavefront q takes
// half q % 2 (rows 8 (q % 2) + 2 lg + {0, 1}) of the row tiles q / 2, q / 2 + 2,
// ... -- two k-steps per row tile and 16-byte load, G_PAIRS loads per block
// (the last one of the wavefronts 2 and 3 would be row tile G_ROWT: zeros)
constexpr int G_PAIRS = (G_ROWT + 1) / 2;
__device__ __forceinline__ void g_load_col(const nb_gd* base, int ld, int col,
                                           int wave, unsigned lane,
                                           nb_d2* v) {
  // (the base is wave-uniform -- chosen by the job's layer -- but not provably
  // so: without the readfirstlane every load sits in a waterfall loop)
  const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(uniform_ptr(base));
  const unsigned voff = lane * 16;
  unsigned soff = ((unsigned)(wave >> 1) * 16 * ld +
                   (unsigned)(col * 2 + (wave & 1)) * 128) * 8;
#pragma unroll
  for (int it = 0; it < G_PAIRS; ++it) {
    if (2 * it + 1 < G_ROWT) {
      v[it] = ld_xcd2(rsrc, voff, soff);
    } else {
      // (G_ROWT is odd: the last pair of the wavefronts 2, 3 is past it.  They
      // load the row tile two before it again -- an unconditional load: a
      // branch or a select here put the wait for the whole memory queue in
      // front of the first MFMA chain -- and scale it to zero; the values are
      // finite, so 0 * v is 0.)
      const unsigned back = wave < 2 ? 0u : 2u * 16 * ld * 8;
      const double keep = wave < 2 ? 1.0 : 0.0;
      const nb_d2 w = ld_xcd2(rsrc, voff, soff - back);
      v[it] = nb_d2{w.x * keep, w.y * keep};
    }
    soff += 2 * 16 * ld * 8;
  }
}

#ifdef NB_TRAIN_TIMING
#define G_TIMED (net == 0 && slot == NB_TRAIN_TIMING_SLOT)
#define G_STAMP(i) NB_STAMP(timed, i)
#else
#define G_TIMED false
#define G_STAMP(i)
#endif

// One job by a workgroup of four wavefronts.  Wavefront q contracts the rows
// 8 (q % 2) + 2 lg + j (j = 0, 1: two MFMAs) of the row tiles q / 2, q / 2 + 2,
// ... in that order (all G_ROWT row tiles -- the delta rows past a short
// minibatch are zero), so a tile is four independent chains of 14 / 12 MFMAs
// on four SIMDs; all operand loads are issued before the first chain.  The partial tiles go
// through LDS; wavefront r then owns rows lg + 4 r of every tile: gradient =
// ((p0 + p1) + p2) + p3, Adam (sklearn _stochastic_optimizers.py:255-287) in
// place, and for the layers 2-4 the transposed copy the next backward pass
// reads.  `first_loads` runs in front of the operand loads (the resident
// kernel fetches the next step's input rows there).
struct GJob { int layer, kt0, nk, ht0, nh; };

__device__ __forceinline__ GJob g_job_record(const nb_gi* jobs, int job) {
  const nb_gi* rec = jobs + job * G_JOB_INTS;
  GJob j;
  j.layer = __builtin_amdgcn_readfirstlane(rec[0]);
  j.kt0 = __builtin_amdgcn_readfirstlane(rec[1]);
  j.nk = __builtin_amdgcn_readfirstlane(rec[2]);
  j.ht0 = __builtin_amdgcn_readfirstlane(rec[3]);
  j.nh = __builtin_amdgcn_readfirstlane(rec[4]);
  return j;
}

// (NK, NH = the job's shape at compile time: one straight-line body per shape,
// so that the wait in front of every MFMA chain is exactly for its operands --
// with the shape as run-time conditions around the loads the first chain
// waited for the whole memory queue)
template <int NK, int NH, class Hook>
__device__ __forceinline__ void g_job_shape(const TrainArgs& a,
                                            const NetState& st,
                                            const GJob& jb, int nb,
                                            double lr_t, double* red,
                                            Hook&& first_loads, bool timed) {
  int lane_ = threadIdx.x & 63;
  // (opaque to the optimiser: per-lane offsets derived from it are
  // recomputed every step instead of being kept -- and spilled -- across the
  // forward / backward pass)
  asm volatile("" : "+v"(lane_));
  const unsigned lane = lane_;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned li = lane & 15, lg = lane >> 4;
  const int layer = jb.layer, kt0 = jb.kt0, ht0 = jb.ht0;
  constexpr int nk = NK, nh = NH;
  const GLayer g = g_layer(st, a.kt1, layer);
  G_STAMP(33);
  // (first in the memory queue: vector memory returns in order, and every
  // load behind the operands is conditional -- the wait in front of the first
  // MFMA chain is therefore one for the whole queue; the rows come from HBM
  // while the operand columns stream out of the L2 behind them)
  first_loads();
  // Up to four operand column blocks: c0 / c2 = the activation blocks kt0,
  // kt0 + 1, c1 = the delta block ht0, and cx = the delta block ht0 + 1 (nh =
  // 2) or the activation block kt0 + 2 (nk = 3, nh = 1) -- one set of
  // registers for both.  Tiles of a job: t = 0: (kt0, ht0); 1: (kt0, ht0 + 1);
  // 2: (kt0 + 1, ht0); 3: (kt0 + 1, ht0 + 1), or (kt0 + 2, ht0) for nk = 3.
  constexpr bool three = nk == 3;
  nb_d2 c0[G_PAIRS], c1[G_PAIRS], c2[G_PAIRS], cx[G_PAIRS];
  g_load_col(g.as, g.lda, kt0, wave, lane, c0);
  g_load_col(g.bs, g.ldb, ht0, wave, lane, c1);
  if (nk > 1) g_load_col(g.as, g.lda, kt0 + 1, wave, lane, c2);
  if (three) g_load_col(g.as, g.lda, kt0 + 2, wave, lane, cx);
  else if (nh > 1) g_load_col(g.bs, g.ldb, ht0 + 1, wave, lane, cx);
  G_STAMP(30);
  // this lane's element of every tile of the job: row lg + 4 wave, column li
  const unsigned eoff = (lg + 4 * wave) * 16 + li;     // moments: row major
  const unsigned woff_e = tile_index(lg + 4 * wave, li);
  const unsigned toff_e = tile_index(li, lg + 4 * wave);
  // (wave-uniform) k-tile / output tile of tile t and whether the job has it
  constexpr bool has[G_MAX_TILES] = {true, nh > 1, nk > 1,
                                     three || (nk > 1 && nh > 1)};
  const int tk[G_MAX_TILES] = {kt0, kt0, kt0 + 1, three ? kt0 + 2 : kt0 + 1};
  const int th[G_MAX_TILES] = {ht0, ht0 + 1, ht0, three ? ht0 : ht0 + 1};
  double w_old[G_MAX_TILES], m_old[G_MAX_TILES], v_old[G_MAX_TILES];
#pragma unroll
  for (int t = 0; t < G_MAX_TILES; ++t)
    if (has[t]) {
      const int woff = g.wbase + (tk[t] * g.ht_n + th[t]) * NB_TILE;
      w_old[t] = ld_xcd(&(st.W + woff)[woff_e]);
      m_old[t] = ld_xcd(&(st.M + woff)[eoff]);
      v_old[t] = ld_xcd(&(st.V + woff)[eoff]);
    }
  auto chain = [&](const nb_d2* av, const nb_d2* bv, int t)
                   __attribute__((always_inline)) {
    nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int it = 0; it < G_PAIRS; ++it) {
      acc = MFMA(av[it].x, bv[it].x, acc);
      acc = MFMA(av[it].y, bv[it].y, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      red[((t * 4 + wave) * 4 + r) * 64 + lane] = acc[r];
  };
  chain(c0, c1, 0);
  if (has[1]) chain(c0, cx, 1);
  if (has[2]) chain(c2, c1, 2);
  if (has[3]) {
    if (three) chain(cx, c1, 3);
    else chain(c2, cx, 3);
  }
  G_STAMP(31);
  lds_barrier();
  G_STAMP(36);
  const double inv_nb = 1.0 / (double)nb;
#pragma unroll
  for (int t = 0; t < G_MAX_TILES; ++t)
    if (has[t]) {
      const double* p = red + (t * 16 + wave) * 64 + lane;
      const double sum = ((p[0] + p[4 * 64]) + p[8 * 64]) + p[12 * 64];
      const double gr = sum * inv_nb;
      const double m = a.b1 * m_old[t] + (1.0 - a.b1) * gr;
      const double v = a.b2 * v_old[t] + (1.0 - a.b2) * (gr * gr);
      const double w = w_old[t] + -lr_t * m / (sqrt(v) + a.eps);
      const int woff = g.wbase + (tk[t] * g.ht_n + th[t]) * NB_TILE;
      (st.M + woff)[eoff] = m;
      (st.V + woff)[eoff] = v;
      (st.W + woff)[woff_e] = w;
      if (g.tbase >= 0) {
        const int toff = g.tbase + (th[t] * g.kt_n + tk[t]) * NB_TILE;
        (st.WT + toff)[toff_e] = w;
      }
    }
  G_STAMP(32);
}

template <class Hook>
__device__ __forceinline__ void g_job(const TrainArgs& a, const NetState& st,
                                      const GJob& jb, int nb, double lr_t,
                                      double* red, Hook&& first_loads,
                                      bool timed = false) {
  const int shape = jb.nk * 4 + jb.nh;           // (wave-uniform)
  if (shape == 2 * 4 + 1)
    g_job_shape<2, 1>(a, st, jb, nb, lr_t, red, first_loads, timed);
  else if (shape == 1 * 4 + 2)
    g_job_shape<1, 2>(a, st, jb, nb, lr_t, red, first_loads, timed);
  else if (shape == 3 * 4 + 1)
    g_job_shape<3, 1>(a, st, jb, nb, lr_t, red, first_loads, timed);
  else if (shape == 2 * 4 + 2)
    g_job_shape<2, 2>(a, st, jb, nb, lr_t, red, first_loads, timed);
  else
    g_job_shape<1, 1>(a, st, jb, nb, lr_t, red, first_loads, timed);
}

__global__ void __launch_bounds__(256)
nb_train_g_kernel(TrainArgs a, int nb, long long t_adam) {
  __shared__ __attribute__((aligned(16))) double red[G_MAX_TILES * 1024];
  const NetState st = a.nets[blockIdx.y];
  if (st.scal[4] != 0.0) return;                 // network already stopped
  // the first workgroup also folds the step's loss (the resident kernel gives
  // that to its least loaded workgroup)
  if (blockIdx.x == 0 && threadIdx.x < 64) loss_fold(st, nb, (int)threadIdx.x);
  g_job(a, st, g_job_record(a.jobs, (int)blockIdx.x), nb, adam_lr(a, t_adam), red,
        []() {});
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
#define NB_POLL_SLEEP 2
#endif
#ifndef NB_POLL_SLEEP_FEW
#define NB_POLL_SLEEP_FEW 1
#endif
constexpr int XCD_COUNT = 8;
constexpr int XCD_SLOTS = 32;            // workgroups per network: one per CU
// per network 512 bytes: [0] counter of the barrier behind FB (and of those
// around the epochs), [1] error, [32] the count of reported upper stashes,
// [64] counter of the barrier that ends a step -- the counters of different
// networks (different XCDs) and the counters of one network on lines of
// their own.  (Sixteen
// bytes per network put the counters of all networks into one 128-byte line,
// on which the atomics and polls of four or eight XCDs met; which small
// allocation landed next to which then decided between 14.7 and 16.5-17 us
// per step from process to process, profiles/r04/second_session/
// train_mode_shift.txt.)
constexpr int SYNC_WORDS = 128;
constexpr int SYNC_UPPER = 32;
constexpr int SYNC_LAST = 64;            // the barrier that ends a step
// (+ one ticket counter per XCD behind the MAX_RESIDENT network records)
constexpr int SYNC_INTS = SYNC_WORDS * 16 + 32 * XCD_COUNT;
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

// wait until `counter` has reached `target` (the count of row tiles whose
// upper stash is in the L2: fb_body's upper_ready)
__device__ __forceinline__ void xcd_wait_for(int* counter, int* err, int target,
                                             bool few) {
  if (threadIdx.x == 0) {
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

// networks of the XCDs: up to two per XCD (two workgroups per CU), -1 = none
struct XcdMap {
  int n_nets;
  int net[XCD_COUNT][2];
};

// per network: its barrier record (SYNC_WORDS ints) in the arena
struct SyncTable { int* rec[MAX_RESIDENT]; };

// Where should the barrier record of an XCD live?  Device-scope atomics and
// the polls of a barrier are served at the memory side, and how far that is
// from an XCD depends on the address: with all records in one place the step
// time followed bit 13 of their address (12.95 against 14.2 us per step at
// n_dim 50 from one process to the next, and either way for half of the eight
// XCDs of an 8-network trainer; profiles/r04/second_session/
// train_sync_shift.txt).  The arena holds ARENA_PLACES candidate places 8 KB
// apart with a record for every (XCD, network of the XCD) in each; this
// kernel times a chain of dependent atomics on every place from every XCD
// (one wavefront per XCD, the first to arrive), the host takes the fastest
// place per XCD.
constexpr int ARENA_PLACES = 8;
constexpr int ARENA_PLACE_INTS = 2048;          // 8 KB
constexpr int ARENA_INTS = ARENA_PLACES * ARENA_PLACE_INTS;
__device__ __forceinline__ int* arena_record(int* arena, int place, int xcd,
                                             int which) {
  return arena + place * ARENA_PLACE_INTS + (xcd * 2 + which) * SYNC_WORDS;
}
__global__ void nb_sync_probe_kernel(int* arena, int* tickets,
                                     long long* ticks) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & (XCD_COUNT - 1));
  if (threadIdx.x != 0) return;
  if (atomicAdd(tickets + 32 * xcd, 1) != 0) return;
  for (int rep = 0; rep < 3; ++rep)
    for (int place = 0; place < ARENA_PLACES; ++place) {
      int* p = arena_record(arena, place, xcd, 0);
      int off = 0;
      const long long t0 = (long long)__builtin_amdgcn_s_memtime();
      for (int i = 0; i < 48; ++i) {
        int r = __hip_atomic_fetch_add(p + off, 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" : "+v"(r));
        off = (r >> 30) & 1;                     // (0: the chain is dependent)
      }
      const long long t1 = (long long)__builtin_amdgcn_s_memtime();
      if (rep > 0) {
        long long* out = ticks + xcd * ARENA_PLACES + place;
        if (rep == 1 || t1 - t0 < *out) *out = t1 - t0;
      }
    }
}

__global__ void nb_xcc_probe_kernel(int* out) {
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = (int)(xcc & 15);
  }
}

#ifdef NB_TRAIN_TIMING
// (NB_TRAIN_TIMING_SLOT: the stamped workgroup of network 0; one without a
// row tile -- 13 to 31 -- has the stamps of the second list)
#ifndef NB_TRAIN_TIMING_SLOT
#define NB_TRAIN_TIMING_SLOT 0
#endif
#ifdef NB_TRAIN_SLOT_TIMING
#define TR_TIMED false
#define SLOT_STAMP(i)                                                         \
  do {                                                                        \
    if (net == 0 && threadIdx.x == 0) {                                       \
      const long long now_ = (long long)__builtin_amdgcn_s_memtime();         \
      s_slot[i] += now_ - s_slot[5];                                          \
      s_slot[5] = now_;                                                       \
    }                                                                         \
  } while (0)
#else
#define TR_TIMED (net == 0 && slot == NB_TRAIN_TIMING_SLOT)
#define SLOT_STAMP(i)
#endif
#define TR_STAMP(i) NB_STAMP(TR_TIMED, i)
// order of the stamps within a step of the stamped workgroup
#if NB_TRAIN_TIMING_SLOT < 13
constexpr int N_STAMPS = 19;
__device__ const int g_stamp_order[N_STAMPS] = {0, 11, 12, 13, 14, 15, 16, 17,
                                                18, 19, 1, 2, 33, 30, 31, 36,
                                                32, 3, 4};
#else
// 5: the upper stash is there; 6: the early job is through; 2: the barrier
// behind FB has opened; 3: the late job (if any) is through (the stamps of
// g_job are those of the early job: the late one is not stamped)
constexpr int N_STAMPS = 12;
__device__ const int g_stamp_order[N_STAMPS] = {0, 1, 5, 33, 30, 31, 36, 32,
                                                6, 2, 3, 4};
#endif
#else
#define TR_STAMP(i)
#define SLOT_STAMP(i)
#endif

// (two workgroups per CU: the register budget of 256 leaves every CU of an
// owned XCD a free slot, through which the workgroups of OTHER grids -- a
// concurrent trainer's, which leave at once here, or any other kernel's --
// pass while this one is resident)
template <int KT1>
__global__ void __launch_bounds__(256, 2)
nb_train_xcd_kernel(TrainArgs a, FleetData fleet, XcdMap map, SyncTable tab,
                    int* sync) {
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
  static_assert(FbLds<KT1>::TOTAL - FbLds<KT1>::G_RED >= G_MAX_TILES * 1024,
                "the partial tiles of G fit behind the input block");
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
  int* ticket = sync + SYNC_WORDS * MAX_RESIDENT + 32 * xcd;
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
  // the network's barrier record: in the process-wide arena, at the place
  // this XCD reaches fastest (sync_arena); its error word is mirrored into
  // the trainer's own sync array when the workgroup leaves
  int* counter = tab.rec[net];
  int* err = counter + 1;
  int* err_mirror = sync + SYNC_WORDS * net + 1;
  NetState st = a.nets[net];
  st.W = uniform_ptr(st.W); st.M = uniform_ptr(st.M); st.V = uniform_ptr(st.V);
  st.WT = uniform_ptr(st.WT); st.stash = uniform_ptr(st.stash);
  st.loss_curve = uniform_ptr(st.loss_curve); st.scal = uniform_ptr(st.scal);
  NetData nd = fleet.d[net];
  nd.X = uniform_ptr(nd.X); nd.y = uniform_ptr(nd.y);
  nd.perm = uniform_ptr(nd.perm);
  // this workgroup's jobs of the G phase (the same in every step).  With the
  // host's schedule (a network on all 32 CUs of its XCD): an EARLY job, of the
  // layers 2-4, which starts as soon as every row tile has reported its upper
  // stash -- the workgroups without a row tile would otherwise idle through
  // the whole forward / backward pass, and the weights of these layers are not
  // read again in this step once every tile is past its delta-2 stage -- and a
  // LATE job, of layer 1, behind the barrier that ends FB.  Without it (two
  // networks per XCD): jobs slot, slot + slots, ... behind that barrier.
  const bool use_sched = a.sched != nullptr && !two;
  int early_i = -1, late_i = slot < a.n_jobs ? slot : -1;
  if (use_sched) {
    early_i = __builtin_amdgcn_readfirstlane(a.sched[2 * slot]);
    late_i = __builtin_amdgcn_readfirstlane(a.sched[2 * slot + 1]);
  }
  const nb_gi* job_list = use_sched ? a.sched_jobs : a.jobs;
  const GJob early_job = g_job_record(job_list, early_i >= 0 ? early_i : 0);
  const GJob late_job = g_job_record(job_list, late_i >= 0 ? late_i : 0);
  int* upper = counter + SYNC_UPPER;   // row tiles whose upper stash is in the L2
  // The barrier that ends a step has a counter of its own: a workgroup with an
  // early job and no late one then never has to look at the barrier behind FB
  // -- it arrives there when the step starts and goes from its job straight to
  // the end of the step (a look cost it a round trip to the L2 behind the
  // stores of its job, and the 3 x 1 early jobs were the last to arrive).
  int* end_ctr = counter + SYNC_LAST;
  int phase_last = 0;
  int ustep = 0;                 // steps run by this launch
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
  // beta_1^t and beta_2^t of the step size (adam_lr), carried from step to
  // step as unevaluated sums hi + lo (one exact product and a renormalisation
  // per step: ~20 instructions, the error of the pair grows by ~2^-104 per
  // step, so hi stays the correctly rounded power for any length of fit).  The
  // two pow() calls they replace were ~500 instructions per step, and wherever
  // they were put -- between the MFMA chains and Adam by the compiler, behind
  // the arrival at the last barrier by hand -- the workgroups with a row tile
  // had them on the critical path: they are never idle.
#ifdef NB_TRAIN_SLOT_TIMING
  if (threadIdx.x == 0) {
    for (int i = 0; i < 8; ++i) s_slot[i] = 0;
    s_slot[5] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif
  double p1h = pow(a.b1, (double)t_adam), p1l = 0.0;
  double p2h = pow(a.b2, (double)t_adam), p2l = 0.0;
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
      dd_scale(p1h, p1l, a.b1);
      dd_scale(p2h, p2l, a.b2);
      if (done) continue;
#ifdef NB_TRAIN_TIMING
      if (TR_TIMED && threadIdx.x == 0) {
        if (have_stamps) {
          long long prev = s_ts[0];
          for (int i = 1; i < N_STAMPS; ++i) {
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
        fb_body<KT1, false>(a, st, net, slot, nb, rows, lds,
                            [&]() __attribute__((always_inline)) {
                              if (lane == 0)
                                __hip_atomic_fetch_add(
                                    upper, 1, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
                            });
      } else if (slot < G_ROWT) {
        fb_clear_deltas(st, 16 * KT1, slot);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
          __hip_atomic_fetch_add(upper, 1, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
      }
      TR_STAMP(1);
      if (!(use_sched && early_i >= 0)) SLOT_STAMP(0);
      xcd_arrive(counter);
      const double lr_t =
          a.lr * sqrt(1.0 - (p2h + p2l)) / (1.0 - (p1h + p1l));
      ustep += 1;
      // One call site of g_job for the early job, the late job and the rounds
      // of the schedule-less form.  The rows of the next step (read-only
      // data) are fetched in front of the operand loads of the first job
      // behind the barrier, back long before the next one.
      have_rows = next_rows && slot * 16 < nb;
      {
        bool fetched = false, past_fb = false, red_used = false;
        for (int r = 0;; ++r) {
          int job;
          bool early = false;
          if (use_sched) {
            if (r >= 2) break;
            early = r == 0;
            job = early ? early_i : late_i;
          } else {
            job = slot + r * slots;
            if (job >= a.n_jobs) job = -1;
            if (job < 0 && past_fb) break;
          }
          if (early) {
            if (job >= 0) xcd_wait_for(upper, err, ustep * G_ROWT, few);
            TR_STAMP(5);
          } else if (!past_fb) {
            TR_STAMP(6);
            if (use_sched && early_i >= 0) SLOT_STAMP(0);
            if (use_sched && job < 0 && slot != slots - 1) {
              phase += 1;                  // (arrived; nothing to wait for)
            } else {
              xcd_wait(counter, err, phase, slots, few);
            }
            TR_STAMP(2);
            SLOT_STAMP(1);
            // the last workgroup folds the loss
            if (slot == slots - 1 && wave == 3) loss_fold(st, nb, lane);
            past_fb = true;
          }
          if (job >= 0) {
            if (red_used) lds_barrier();             // red is reused
            const GJob jb = use_sched ? (early ? early_job : late_job)
                                      : (r == 0 ? late_job
                                                : g_job_record(a.jobs, job));
            g_job(a, st, jb, nb, lr_t, lds + FbLds<KT1>::G_RED,
                  [&]() __attribute__((always_inline)) {
                    if (past_fb && have_rows && !fetched) {
                      fb_gather<KT1>(nd, a.n_dim, slot, nb2, row_next, rows);
                      fetched = true;
                    }
                  },
                  G_TIMED && !(use_sched && !early && early_i >= 0));
            red_used = true;
          }
        }
        if (have_rows && !fetched)
          fb_gather<KT1>(nd, a.n_dim, slot, nb2, row_next, rows);
      }
      TR_STAMP(3);
      SLOT_STAMP(2);
      xcd_barrier(end_ctr, err, phase_last, slots, few);
      TR_STAMP(4);
      SLOT_STAMP(3);
#ifdef NB_TRAIN_SLOT_TIMING
      if (net == 0 && threadIdx.x == 0) s_slot[4] += 1;
#endif
      if (__hip_atomic_load(err, __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT) != 0) {
        if (slot == 0 && threadIdx.x == 0) *err_mirror = 1;
        return;
      }
    }
    if (done) continue;
    if (slot == 0 && threadIdx.x == 0) epoch_body(a, st, n, t_adam);
    xcd_barrier(counter, err, phase, slots, few);
#ifdef NB_TRAIN_SLOT_TIMING
    if (net == 0 && threadIdx.x == 0)
      s_slot[5] = (long long)__builtin_amdgcn_s_memtime();
#endif
  }
  if (slot == 0 && threadIdx.x == 0)
    *err_mirror = __hip_atomic_load(err, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
#ifdef NB_TRAIN_SLOT_TIMING
  if (net == 0 && threadIdx.x == 0)
    for (int i = 0; i < 5; ++i)
      atomicAdd((unsigned long long*)&g_slot_ticks[slot * 8 + i],
                (unsigned long long)s_slot[i]);
#endif
}

void put_w(double* tiles, int ht_n, int k, int h, double v) {
  tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE +
        tile_index(k & 15, h & 15)] = v;
}
double get_w(const double* tiles, int ht_n, int k, int h) {
  return tiles[((size_t)(k >> 4) * ht_n + (h >> 4)) * NB_TILE +
               tile_index(k & 15, h & 15)];
}
// The job list of the forms without a schedule (two launches per step; two
// networks per XCD): blocks of up to 2 x 2 tiles per layer, shaped so that a
// network needs at most 32 jobs at every n_dim (two rounds of 16 workgroups).
std::vector<int> g_jobs_rounds(int kt1) {
  std::vector<int> jobs;
  auto blocks = [&](int layer, int kt_n, int ht_n, int bk, int bh) {
    for (int kt = 0; kt < kt_n; kt += bk)
      for (int ht = 0; ht < ht_n; ht += bh) {
        const int rec[G_JOB_INTS] = {layer, kt, kt + bk <= kt_n ? bk : 1, ht,
                                     ht + bh <= ht_n ? bh : 1};
        jobs.insert(jobs.end(), rec, rec + G_JOB_INTS);
      }
  };
  // layer 1 (kt1 x 7 tiles): pairs along k up to 64 dimensions, 2 x 2 beyond
  blocks(0, kt1, NB_HT1, 2, kt1 <= 4 ? 1 : 2);
  // layer 2 (7 x 4): pairs along h, 2 x 2 where layer 1 needs the workgroups
  blocks(1, NB_HT1, NB_HT2, kt1 >= 7 ? 2 : 1, 2);
  blocks(2, NB_HT2, NB_HT3, 2, 2);      // layer 3 (4 x 2)
  blocks(3, NB_HT3, 1, 2, 1);           // layer 4 (2 x 1)
  return jobs;
}

// The jobs of the G phase (see g_job) and their schedule on the 32 workgroups
// of a resident network.
//
// Jobs: blocks of nk x nh (1 or 2 each) weight tiles of one layer.  Measured
// on the resident kernel (profiles/r04/second_session/train_phases_*.txt) a
// job costs about 2.2 k ticks + 0.8 k per operand column block + 1.35 k per
// tile (MFMA chains, reduction, Adam, stores).
//
// Schedule: the G_ROWT workgroups with a row tile run FB; the other 19 start
// an EARLY job each -- the 38 tiles of the layers 2-4 -- G_HEAD ticks before
// the barrier that ends FB opens, and are free for a LATE job when they are
// through; the layer-1 tiles are all late.  Searched: the block shape of the
// late jobs (the job that ends last is split while that shortens the phase),
// and the number of workgroups without a row tile that are kept for a late
// job ONLY -- the early jobs then move together, pairs of them becoming 2 x 2
// blocks.  (n_dim 50: thirteen 2-tile late jobs on the FB workgroups, the
// fourteenth on a workgroup of its own, sixteen 2-tile early jobs and two of
// 3 x 1 tiles.)  G_HEAD is small: the report of the upper stash takes three
// round trips to the L2 (store acknowledgement, counter, poll) of ~500 ticks.
constexpr double G_HEAD = 3.0;
static double g_cost(int nk, int nh) {
  return 2.2 + 0.8 * (nk + nh) + 1.35 * nk * nh;
}
struct GRec { int layer, kt0, nk, ht0, nh; };

static void g_blocks(std::vector<GRec>& out, int layer, int kt_n, int ht_n,
                     int bk, int bh) {
  for (int kt = 0; kt < kt_n; kt += bk)
    for (int ht = 0; ht < ht_n; ht += bh)
      out.push_back({layer, kt, kt + bk <= kt_n ? bk : kt_n - kt, ht,
                     ht + bh <= ht_n ? bh : 1});
}

// the early jobs (the 38 tiles of the layers 2-4): 19 of two tiles, or fewer
// with 3 x 1 jobs among them -- level 1: layer 3 as (3, 3, 2) tiles instead of
// four pairs; levels 2, 3: the columns of one / both output-tile pairs of
// layer 2 as (3, 2, 2) along k instead of seven 1 x 2 jobs per pair
static std::vector<GRec> g_early(int level) {
  std::vector<GRec> out;
  if (level >= 1) {                              // layer 3 (4 x 2 tiles)
    out.push_back({2, 0, 3, 0, 1});
    out.push_back({2, 0, 3, 1, 1});
    out.push_back({2, 3, 1, 0, 2});
  } else {
    g_blocks(out, 2, NB_HT2, NB_HT3, 2, 1);
  }
  for (int ht = 0; ht < NB_HT2; ht += 2) {       // layer 2 (7 x 4 tiles)
    if (level >= 2 + ht / 2) {
      for (int h = ht; h < ht + 2; ++h) {
        out.push_back({1, 0, 3, h, 1});
        out.push_back({1, 3, 2, h, 1});
        out.push_back({1, 5, 2, h, 1});
      }
    } else {
      for (int kt = 0; kt < NB_HT1; ++kt) out.push_back({1, kt, 1, ht, 2});
    }
  }
  out.push_back({3, 0, 2, 0, 1});                // layer 4 (2 x 1 tiles)
  return out;
}

struct GPlan {
  std::vector<GRec> early, late;
  std::vector<int> early_slot, late_slot;
  double span = 1e30;
};

// the dearest late job to the workgroup that is free first, and so on
static double g_pair(const std::vector<GRec>& late,
                     const std::vector<std::pair<double, int>>& free_at,
                     std::vector<int>& slot_of, int& critical) {
  std::vector<int> idx(late.size());
  for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
  std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
    return g_cost(late[x].nk, late[x].nh) > g_cost(late[y].nk, late[y].nh);
  });
  slot_of.assign(late.size(), -1);
  double span = 0.0;
  critical = -1;
  for (size_t r = 0; r < idx.size(); ++r) {
    const int j = idx[r];
    const double end = free_at[r].first + g_cost(late[j].nk, late[j].nh);
    slot_of[j] = free_at[r].second;
    if (end > span) { span = end; critical = j; }
  }
  return span;
}

// jobs: records of G_JOB_INTS ints; sched: (early, late) job index per
// workgroup, -1 = none
void g_plan(int kt1, std::vector<int>& jobs, std::vector<int>& sched) {
  const int n_free = XCD_SLOTS - G_ROWT;
  GPlan best;
  const int shapes[5][2] = {{2, 1}, {1, 2}, {3, 1}, {2, 2}, {1, 1}};
  // (experiment switches: force the number of late-only workgroups / the
  // block shape of the late jobs instead of taking the cost model's choice)
  const char* e_lo = getenv("NB_TRAIN_LATE_ONLY");
  const char* e_sh = getenv("NB_TRAIN_LATE_SHAPE");
  const int f_lo = e_lo != nullptr ? atoi(e_lo) : -1;
  const int f_sh = e_sh != nullptr ? atoi(e_sh) : -1;
  for (int late_only = 0; late_only <= 3; ++late_only) {
    if (f_lo >= 0 && late_only != f_lo) continue;
    const std::vector<GRec> early = g_early(late_only);
    if ((int)early.size() + late_only != n_free) continue;
    // when the workgroups are free for a late job
    std::vector<std::pair<double, int>> free_at;
    std::vector<int> early_slot(early.size());
    double early_end = 0.0;
    for (int w = 0; w < G_ROWT + late_only; ++w) free_at.push_back({0.0, w});
    for (size_t i = 0; i < early.size(); ++i) {
      const int w = G_ROWT + late_only + (int)i;
      double c = g_cost(early[i].nk, early[i].nh) - G_HEAD;
      if (c < 0.0) c = 0.0;
      if (c > early_end) early_end = c;
      free_at.push_back({c, w});
      early_slot[i] = w;
    }
    std::stable_sort(free_at.begin(), free_at.end(),
                     [](const std::pair<double, int>& x,
                        const std::pair<double, int>& y) {
                       return x.first < y.first;
                     });
    for (int si = 0; si < 5; ++si) {
      if (f_sh >= 0 && si != f_sh) continue;
      const int* sh = shapes[si];
      std::vector<GRec> late;
      g_blocks(late, 0, kt1, NB_HT1, sh[0], sh[1]);
      while ((int)late.size() <= XCD_SLOTS) {
        std::vector<int> slot_of;
        int crit;
        double span = g_pair(late, free_at, slot_of, crit);
        if (early_end > span) span = early_end;
        if (span < best.span - 1e-9) {
          best.span = span; best.early = early; best.late = late;
          best.early_slot = early_slot; best.late_slot = slot_of;
        }
        GRec& c = late[crit];
        if (c.nk * c.nh == 1) break;
        GRec half = c;
        if (c.nk == 3) { c.nk = 2; half.nk = 1; half.kt0 += 2; }
        else if (c.nk == 2) { c.nk = 1; half.nk = 1; half.kt0 += 1; }
        else { c.nh = 1; half.nh = 1; half.ht0 += 1; }
        late.push_back(half);
      }
    }
  }
  jobs.clear();
  sched.clear();
  if (best.late.empty()) return;   // (the kernel then runs g_jobs_rounds)
  sched.assign(2 * XCD_SLOTS, -1);
  for (size_t i = 0; i < best.early.size(); ++i) {
    const GRec& r = best.early[i];
    const int rec[G_JOB_INTS] = {r.layer, r.kt0, r.nk, r.ht0, r.nh};
    sched[2 * best.early_slot[i]] = (int)jobs.size() / G_JOB_INTS;
    jobs.insert(jobs.end(), rec, rec + G_JOB_INTS);
  }
  for (size_t j = 0; j < best.late.size(); ++j) {
    const GRec& r = best.late[j];
    const int rec[G_JOB_INTS] = {r.layer, r.kt0, r.nk, r.ht0, r.nh};
    sched[2 * best.late_slot[j] + 1] = (int)jobs.size() / G_JOB_INTS;
    jobs.insert(jobs.end(), rec, rec + G_JOB_INTS);
  }
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

// the process-wide arena of barrier records and the place of every XCD in it
// (nb_sync_probe_kernel; NB_TRAIN_SYNC_PLACE=<p> forces place p for all)
static int* g_arena = nullptr;
static int g_place_of_xcd[XCD_COUNT];
static bool sync_arena() {
  static int state = 0;             // 1 = there, -1 = failed
  if (state != 0) return state == 1;
  state = -1;
  int* tickets = nullptr;
  long long* ticks = nullptr;
  if (hipMalloc((void**)&g_arena, ARENA_INTS * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&tickets, 32 * XCD_COUNT * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&ticks, XCD_COUNT * ARENA_PLACES * sizeof(long long)) !=
          hipSuccess)
    return false;
  (void)hipMemset(g_arena, 0, ARENA_INTS * sizeof(int));
  (void)hipMemset(tickets, 0, 32 * XCD_COUNT * sizeof(int));
  (void)hipMemset(ticks, 0, XCD_COUNT * ARENA_PLACES * sizeof(long long));
  hipLaunchKernelGGL(nb_sync_probe_kernel, dim3(XCD_COUNT * XCD_SLOTS),
                     dim3(64), 0, 0, g_arena, tickets, ticks);
  long long host[XCD_COUNT * ARENA_PLACES];
  const hipError_t e =
      hipMemcpy(host, ticks, sizeof host, hipMemcpyDeviceToHost);
  (void)hipFree(tickets);
  (void)hipFree(ticks);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }
  const char* forced = getenv("NB_TRAIN_SYNC_PLACE");
  for (int x = 0; x < XCD_COUNT; ++x) {
    int best = 0;
    for (int p = 1; p < ARENA_PLACES; ++p)
      if (host[x * ARENA_PLACES + p] < host[x * ARENA_PLACES + best]) best = p;
    g_place_of_xcd[x] = forced ? atoi(forced) % ARENA_PLACES : best;
    if (getenv("NB_TRAIN_DEBUG_BLOCK") != nullptr) {
      fprintf(stderr, "[trainer] XCD %d: ticks per 48 dependent atomics by "
              "place:", x);
      for (int p = 0; p < ARENA_PLACES; ++p)
        fprintf(stderr, " %lld", host[x * ARENA_PLACES + p]);
      fprintf(stderr, " -> place %d\n", g_place_of_xcd[x]);
    }
  }
  (void)hipMemset(g_arena, 0, ARENA_INTS * sizeof(int));
  state = 1;
  return true;
}

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
  char* block = nullptr;           // the one device allocation (see create)
  char* block_alloc = nullptr;
  int* sync_dev = nullptr;         // per network: counter, error, xcc mask
  int* jobs_dev = nullptr;         // job list of the G phase
  int* sched_dev = nullptr;        // (early, late) job per workgroup, or null
  int n_jobs = 0;
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
  const long long stash = (long long)MAXB * (16 * t->kt1 + 2 * (LD1 + LD2 + LD3) + LD4);
  const long long curve = (t->max_iter + 1) & ~1LL;   // 16-byte alignment
  // (a multiple of 4 KB: no line of one network's record next to another's)
  const long long per_net =
      (3 * t->n_w + WT_DOUBLES + stash + curve + 32 + 511) / 512 * 512;
  t->per_net = per_net;
  // ONE allocation for everything the kernels touch -- [barrier counters, 12
  // KB][network records, 1 KB][job lists, 1 + 2 KB][weights, moments, stash, loss
  // curve and scalars of every network] -- so that the small arrays lie the
  // same way relative to each other and to the pool in every process (as
  // separate allocations they landed wherever the allocator had a slot, and
  // the step time followed).
  const std::vector<int> jobs = g_jobs_rounds(t->kt1);
  t->n_jobs = (int)jobs.size() / G_JOB_INTS;
  // (NB_TRAIN_NO_SCHEDULE: all jobs behind the barrier, for comparison)
  std::vector<int> sjobs, sched;
  g_plan(t->kt1, sjobs, sched);
  const bool use_sched =
      !sched.empty() && getenv("NB_TRAIN_NO_SCHEDULE") == nullptr;
  // [sched (2 per workgroup)][its job records]
  sched.insert(sched.end(), sjobs.begin(), sjobs.end());
  constexpr size_t OFF_NETS = 12288, OFF_JOBS = 13312, OFF_SCHED = 14336,
                   OFF_SYNC2 = 16384, OFF_POOL = 32768;
  static_assert(SYNC_INTS * sizeof(int) <= OFF_NETS &&
                MAX_RESIDENT * sizeof(NetState) <= OFF_JOBS - OFF_NETS,
                "trainer block layout");
  // (with the NB_TRAIN_SYNC_SHIFT experiment the counters sit at OFF_SYNC2,
  // inside the schedule's slot: the schedule then has to end in front of them)
  const size_t sched_end =
      getenv("NB_TRAIN_SYNC_SHIFT") != nullptr ? OFF_SYNC2 : OFF_POOL;
  if (jobs.size() * sizeof(int) > OFF_SCHED - OFF_JOBS ||
      sched.size() * sizeof(int) > sched_end - OFF_SCHED) {
    nb_set_error("trainer: job lists exceed their slots");
    delete t;
    return NB_ERR_ARG;
  }
  // (whole 2 MB fragments: the mapping of a block that ends inside one was
  // seen to cost up to a microsecond per step)
  const size_t bytes =
      (OFF_POOL + (size_t)per_net * n_networks * sizeof(double) +
       (2u << 20) - 1) / (2u << 20) * (2u << 20);
  // (NB_TRAIN_BLOCK_SHIFT: experiment -- the block starts that many KB into
  // its allocation)
  const size_t shift = getenv("NB_TRAIN_BLOCK_SHIFT")
      ? (size_t)atol(getenv("NB_TRAIN_BLOCK_SHIFT")) * 1024 : 0;
  const size_t slack = getenv("NB_TRAIN_BLOCK_SHIFT") ? (2u << 20) : 0;
  hipError_t e = hipMalloc((void**)&t->block_alloc, bytes + slack);
  if (e == hipSuccess) e = hipMemset(t->block_alloc, 0, bytes + slack);
  if (e == hipSuccess) {
    t->block = t->block_alloc + shift;
    if (getenv("NB_TRAIN_DEBUG_BLOCK") != nullptr)
      fprintf(stderr, "[trainer] block at %p\n", (void*)t->block);
    // (NB_TRAIN_SYNC_SHIFT: experiment -- the counters 16 KB + that many KB
    // into the block instead of at its start)
    t->sync_dev = (int*)t->block;
    if (getenv("NB_TRAIN_SYNC_SHIFT"))
      t->sync_dev = (int*)(t->block + OFF_SYNC2 +
                           (size_t)atol(getenv("NB_TRAIN_SYNC_SHIFT")) * 1024);
    t->nets_dev = (NetState*)(t->block + OFF_NETS);
    t->jobs_dev = (int*)(t->block + OFF_JOBS);
    t->pool = (double*)(t->block + OFF_POOL);
    e = hipMemcpy(t->jobs_dev, jobs.data(), jobs.size() * sizeof(int),
                  hipMemcpyHostToDevice);
    if (e == hipSuccess && use_sched) {
      t->sched_dev = (int*)(t->block + OFF_SCHED);
      e = hipMemcpy(t->sched_dev, sched.data(), sched.size() * sizeof(int),
                    hipMemcpyHostToDevice);
    }
  }
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
  if (getenv("NB_TRAIN_DEBUG") != nullptr) {
    std::vector<int> sjobs, sched;
    g_plan(t->kt1, sjobs, sched);
    for (size_t w = 0; 2 * w + 1 < sched.size(); ++w)
      for (int r = 0; r < 2; ++r) {
        const int j = sched[2 * w + r];
        if (j >= 0)
          fprintf(stderr, "[trainer] workgroup %2d %s job: layer %d, k-tiles "
                  "%d+%d, h-tiles %d+%d\n", (int)w, r == 0 ? "early" : "late ",
                  sjobs[5 * j] + 1, sjobs[5 * j + 1], sjobs[5 * j + 2],
                  sjobs[5 * j + 3], sjobs[5 * j + 4]);
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
    s.W = (nb_gd*)base; s.M = s.W + t->n_w; s.V = s.M + t->n_w;
    s.WT = s.V + t->n_w;
    s.stash = s.WT + WT_DOUBLES;
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
  a.jobs = (const nb_gi*)t->jobs_dev; a.n_jobs = t->n_jobs;
  a.sched = (const nb_gi*)t->sched_dev;
  a.sched_jobs = a.sched ? a.sched + 2 * XCD_SLOTS : nullptr;
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
    SyncTable tab;
    for (int i = 0; i < MAX_RESIDENT; ++i) tab.rec[i] = t->sync_dev;
    for (int x = 0; x < XCD_COUNT; ++x)
      for (int w = 0; w < 2; ++w) {
        const int net = t->xcd_map.net[x][w];
        if (net < 0) continue;
        tab.rec[net] = sync_arena()
            ? g_arena + g_place_of_xcd[x] * ARENA_PLACE_INTS +
                  (x * 2 + w) * SYNC_WORDS
            : t->sync_dev + SYNC_WORDS * net;
        if (tab.rec[net] != t->sync_dev + SYNC_WORDS * net)
          NB_HIP_CHECK(hipMemsetAsync(tab.rec[net], 0,
                                      SYNC_WORDS * sizeof(int), s));
      }
    // 32 workgroups per XCD: all of them for its network, or 16 for each of
    // its two
    const dim3 grid(XCD_COUNT * XCD_SLOTS), blk(256);
    switch (t->kt1) {
#define NB_CASE(KT1_)                                                      \
      case KT1_:                                                           \
        hipLaunchKernelGGL(nb_train_xcd_kernel<KT1_>, grid, blk, 0, s, a,  \
                           fleet, t->xcd_map, tab, t->sync_dev);           \
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
      const dim3 gfb(G_ROWT, t->E), gg(t->n_jobs, t->E), blk(256);
      t->t_adam += 1;
      switch (t->kt1) {
#define NB_CASE(KT1_)                                                      \
        case KT1_:                                                         \
          hipLaunchKernelGGL(nb_train_fb_kernel<KT1_>, gfb, blk, 0, s, a,  \
                             ep, start, nb);                               \
          break;
        NB_CASE(1) NB_CASE(2) NB_CASE(3) NB_CASE(4) NB_CASE(5)
        NB_CASE(6) NB_CASE(7) NB_CASE(8) NB_CASE(9)
#undef NB_CASE
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
int nb_dbg_train_slot_times(long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slot_ticks), 256 * sizeof(long long));
  long long zero[256] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_slot_ticks), zero, sizeof zero);
  return 0;
}
int nb_dbg_train_times(long long* out) {
  hipMemcpyFromSymbol(out, HIP_SYMBOL(g_train_ticks), 64 * sizeof(long long));
  long long zero[64] = {0};
  hipMemcpyToSymbol(HIP_SYMBOL(g_train_ticks), zero, sizeof zero);
  return 0;
}
#endif

int nb_trainer_destroy(nb_trainer* t) {
  if (t == nullptr) return NB_OK;
  if (t->block_alloc) (void)hipFree(t->block_alloc);
  g_xcd_in_use &= ~t->xcd_owned;
  if (t->pin_scal) (void)hipHostFree(t->pin_scal);
  if (t->pin_sync) (void)hipHostFree(t->pin_sync);
  for (int i = 0; i < nb_trainer::RING; ++i)
    if (t->ring_event[i]) (void)hipEventDestroy(t->ring_event[i]);
  delete t;
  return NB_OK;
}

}  // extern "C"
