// Dense emulator layers on v_mfma_f64_16x16x4_f64 as ONE operand pipeline per
// stage (used by nb_eval_fast.hip): rows = output units, cols = the 16 points
// of a tile, K = input units; weights as A operands from tile-major LDS
// storage, activations as B operands in registers (the C/D layout of a layer
// is the B layout of the next one).
//
// Differences to the layer code of nb_eval.hip:
//  * the A operands of the first k-step of a block are read by the block
//    BEFORE it (during its last k-step), across block and layer boundaries,
//    so the LDS latency is exposed once per stage instead of once per block;
//  * ReLU is applied by the consumer (one v_max_f64 per B operand and k-step,
//    hidden behind the MFMAs) instead of by the producer at the end of a
//    block, where it had to wait for the matrix pipe to drain;
//  * the weight DMA of the next stage is sliced into the k-steps (`tick`)
//    instead of being issued in one burst at the start of the stage, where
//    the 64 B/clk fill path of the CU stalled the issuing wavefronts.
// Summation order per output unit is unchanged (k-steps ascending), so the
// results are bit-identical to nb_eval.hip's.
#pragma once
#include "nb_tile.h"

#define NB_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

namespace {

constexpr int fl_blk_end(int HT, int H0, int SPLIT) {
  return (HT - 1 - H0 > SPLIT) ? H0 + SPLIT : HT;
}

// output tiles [H0, H1) of a layer with HT tiles (+ its partial last tile if
// H1 == HT): NF full 16-unit tiles, NA operands per k-step
template <int HT, int H0, int H1>
struct FlShape {
  static constexpr int NFL = HT - 1;
  static constexpr bool REM = (H1 == HT);
  static constexpr int NF = (REM ? NFL : H1) - H0;
  static constexpr int NA = NF + (REM ? 1 : 0);
};

template <int HT, int H0, int H1>
__device__ __forceinline__ void fl_read_a(
    const double* w, int ks, int lane, double (&a)[FlShape<HT, H0, H1>::NA]) {
  using S = FlShape<HT, H0, H1>;
  const double* wk = w + (ks >> 2) * HT * NB_TILE + (ks & 3) * 64;
#pragma unroll
  for (int h = 0; h < S::NF; ++h) a[h] = wk[(H0 + h) * NB_TILE + lane];
  // partial tile on v_mfma_f64_4x4x4_4b: element (kk, hh = lane & 3)
  if constexpr (S::REM)
    a[S::NF] = wk[S::NFL * NB_TILE + (lane >> 4) * 16 + (lane & 3)];
}

// One block.  a0 = operands of k-step 0 (reads issued by the caller); `next`
// issues the reads of whatever follows this block, `tick` may issue one DMA
// instruction (called every TICK_P k-steps).  The last NGUARD k-steps depend
// on the runtime count ks_n and are not pipelined.  PD = k-steps the operand
// reads run ahead of the MFMAs (1 where the accumulators need the registers;
// more for the small last layers, whose k-steps are shorter than the LDS
// latency).
template <int T, int KSMAX, int NGUARD, int HT, int H0, int H1, bool RELU_IN,
          int TICK_P, int PD, int NIN, int NOUT, class Next, class Tick>
__device__ __forceinline__ void fl_block(
    const double* w, int ks_n, const double (&in)[T][NIN], int lane,
    double (&out)[T][NOUT], const double (&a0)[FlShape<HT, H0, H1>::NA],
    Next&& next, Tick&& tick) {
  using S = FlShape<HT, H0, H1>;
  constexpr int NF = S::NF, NA = S::NA, NFL = S::NFL;
  constexpr bool REM = S::REM;
  constexpr int KS_U = KSMAX - NGUARD;
  static_assert(KS_U >= 1 && KS_U >= PD && PD >= 1, "prefetch distance");
  constexpr int R = PD + 1;
  nb_d4 acc[T][NF > 0 ? NF : 1];
  double rem[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < (NF > 0 ? NF : 1); ++h)
      acc[t][h] = nb_d4{0.0, 0.0, 0.0, 0.0};
    rem[t] = 0.0;
  }
  auto step = [&](int ks, const double (&a)[NA]) __attribute__((always_inline)) {
    double b[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
      b[t] = RELU_IN ? fmax(in[t][ks], 0.0) : in[t][ks];
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t][h] = MFMA(a[h], b[t], acc[t][h]);
    if constexpr (REM) {
#pragma unroll
      for (int t = 0; t < T; ++t) rem[t] = NB_MFMA4(a[NF], b[t], rem[t]);
    }
  };
  // a use of the operands: the compiler's s_waitcnt lands here, before the
  // next reads are issued
  auto arrived = [&](const double (&a)[NA]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(a[i]));
  };
  double a[R][NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[0][i] = a0[i];
#pragma unroll
  for (int i = 1; i < PD; ++i) fl_read_a<HT, H0, H1>(w, i, lane, a[i]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < KS_U; ++ks) {
    arrived(a[ks % R]);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + PD < KS_U)
      fl_read_a<HT, H0, H1>(w, ks + PD, lane, a[(ks + PD) % R]);
    else if (ks + PD == KS_U) next();
    if (ks % TICK_P == 0) tick();
    __builtin_amdgcn_sched_barrier(0);
    step(ks, a[ks % R]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ks = KS_U; ks < KSMAX; ++ks) {
    if (ks < ks_n) {
      double ag[NA];
      fl_read_a<HT, H0, H1>(w, ks, lane, ag);
      step(ks, ag);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][4 * (H0 + h) + r] = acc[t][h][r];
    if constexpr (REM) out[t][4 * NFL] = rem[t];
  }
}

// K-chunked form of fl_block for a layer whose weights exceed an LDS region
// (layer 1 for n_dim > 64): k-steps [KS_LO, KS_HI) of all output tiles of the
// layer, `out` carries the pre-activations between the chunks, `w` is the
// chunk in LDS (k-tile index relative to KS_LO / 4).  The runtime-guarded
// k-steps are the last three of the layer.
template <int T, int KSMAX, int HT, int KS_LO, int KS_HI, int TICK_P, int NIN,
          int NOUT, class Tick>
__device__ __forceinline__ void fl_chunk(
    const double* w, int ks_n, const double (&in)[T][NIN], int lane,
    double (&out)[T][NOUT], Tick&& tick) {
  using S = FlShape<HT, 0, HT>;
  constexpr int NF = S::NF, NA = S::NA, NFL = S::NFL;
  static_assert(KS_LO % 4 == 0 && KS_LO < KS_HI && KS_HI <= KSMAX, "chunk");
  constexpr bool FIRST = (KS_LO == 0);
  constexpr int KS_U = (KS_HI < KSMAX - 3) ? KS_HI : KSMAX - 3;
  static_assert(KS_U > KS_LO, "at least one unguarded k-step per chunk");
  nb_d4 acc[T][NF];
  double rem[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[t][h][r] = FIRST ? 0.0 : out[t][4 * h + r];
    rem[t] = FIRST ? 0.0 : out[t][4 * NFL];
  }
  auto step = [&](int ks, const double (&a)[NA]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t)
        acc[t][h] = MFMA(a[h], in[t][ks], acc[t][h]);
#pragma unroll
    for (int t = 0; t < T; ++t) rem[t] = NB_MFMA4(a[NF], in[t][ks], rem[t]);
  };
  auto arrived = [&](const double (&a)[NA]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(a[i]));
  };
  double a[2][NA];
  fl_read_a<HT, 0, HT>(w, 0, lane, a[0]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = KS_LO; ks < KS_U; ++ks) {
    const int i = ks - KS_LO;
    arrived(a[i & 1]);
    __builtin_amdgcn_sched_barrier(0);
    if (ks + 1 < KS_U) fl_read_a<HT, 0, HT>(w, i + 1, lane, a[(i + 1) & 1]);
    if (i % TICK_P == 0) tick();
    __builtin_amdgcn_sched_barrier(0);
    step(ks, a[i & 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int ks = (KS_LO > KS_U ? KS_LO : KS_U); ks < KS_HI; ++ks) {
    if (ks < ks_n) {
      double ag[NA];
      fl_read_a<HT, 0, HT>(w, ks - KS_LO, lane, ag);
      step(ks, ag);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][4 * h + r] = acc[t][h][r];
    out[t][4 * NFL] = rem[t];
  }
}

// a layer = blocks of SPLIT output tiles chained through their first operands
template <int T, int SPLIT, int KSMAX, int NGUARD, int HT, bool RELU_IN,
          int TICK_P, int PD, int H0, int NIN, int NOUT, class Next, class Tick>
__device__ __forceinline__ void fl_layer_from(
    const double* w, int ks_n, const double (&in)[T][NIN], int lane,
    double (&out)[T][NOUT],
    const double (&a0)[FlShape<HT, H0, fl_blk_end(HT, H0, SPLIT)>::NA],
    Next&& next, Tick&& tick) {
  constexpr int H1 = fl_blk_end(HT, H0, SPLIT);
  if constexpr (H1 < HT) {
    constexpr int H2 = fl_blk_end(HT, H1, SPLIT);
    double a1[FlShape<HT, H1, H2>::NA];
    fl_block<T, KSMAX, NGUARD, HT, H0, H1, RELU_IN, TICK_P, PD>(
        w, ks_n, in, lane, out, a0,
        [&]() __attribute__((always_inline)) {
          fl_read_a<HT, H1, H2>(w, 0, lane, a1);
        },
        tick);
    fl_layer_from<T, SPLIT, KSMAX, NGUARD, HT, RELU_IN, TICK_P, PD, H1>(
        w, ks_n, in, lane, out, a1, next, tick);
  } else {
    fl_block<T, KSMAX, NGUARD, HT, H0, HT, RELU_IN, TICK_P, PD>(
        w, ks_n, in, lane, out, a0, next, tick);
  }
}

// operands of the first k-step of a layer (block 0)
template <int SPLIT, int HT>
struct FlFirst {
  static constexpr int H1 = fl_blk_end(HT, 0, SPLIT);
  static constexpr int NA = FlShape<HT, 0, H1>::NA;
};
template <int SPLIT, int HT>
__device__ __forceinline__ void fl_read_first(
    const double* w, int lane, double (&a)[FlFirst<SPLIT, HT>::NA]) {
  fl_read_a<HT, 0, FlFirst<SPLIT, HT>::H1>(w, 0, lane, a);
}

// registers 1..3 of a layer's partial last tile: zero padding, except the
// constant 1 that feeds the next layer's bias row (k-step ONE_KS, lane group
// ONE_LG; it may share the register of the partial tile, 50 = 48 + 2)
template <int T, int HT, int ONE_KS, int ONE_LG, int NOUT>
__device__ __forceinline__ void fl_pad(double (&out)[T][NOUT], int lane) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const int idx = 4 * (HT - 1) + r;
      out[t][idx] = (idx == ONE_KS && (lane >> 4) == ONE_LG) ? 1.0 : 0.0;
    }
  if constexpr (ONE_KS >= 0 && ONE_KS == 4 * (HT - 1)) {
    if ((lane >> 4) == ONE_LG) {
#pragma unroll
      for (int t = 0; t < T; ++t) out[t][ONE_KS] = 1.0;
    }
  }
}

}  // namespace
