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
//
// Workgroup structure: 4 wavefronts x 2 tiles = 128 points per workgroup pass
// (8 wavefronts x 1 tile for n_dim > 64), one workgroup per CU (the kernel
// uses the whole register file), grid-stride over the 128-point super tiles.
// The launches NautilusBound.sample spends its time in -- one neural bound, at
// most one outer member -- go through nb_eval_fast.hip instead; this kernel
// serves every other shape: bound lists (shell exclusion / association),
// unions with several members, several neural bounds, overlap counts.
// Emulator weights are the bulk of the operand traffic (132 KB per network
// at D = 50), so they stream through two LDS regions of 38 tiles: while one
// region feeds the matrix cores the other is refilled by global_load_lds DMA
// (layer 1, for n_dim > 64 in two K chunks, then layers 2-4; one barrier per
// stage).  All tiles of the workgroup read their A operands from LDS
// (conflict-free ds_read_b64) and each wavefront shares every A operand
// between its two tiles.  The partial last tile of every layer (4, 2, 4, 1
// units) runs on v_mfma_f64_4x4x4_4b_f64.  Ellipsoid blocks are staged in LDS
// per bound.  Measured on MI355X: 44 TFLOP/s
// algorithmic at D = 50 (0.56 of the fp64 MFMA peak), HBM traffic 1.02x
// algorithmic.
#include "nb_common.h"

#include <cstdlib>

#include "nb_tile.h"

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

// One dense layer on the matrix cores for the T tiles of a wavefront:
// out[h] = act(sum_k in[k] W[k][h]), bias folded in as row k = K (the input
// carries a constant 1 there).  w points to LDS.
//
// Loop order: k-step outer, output tile inner.  Every B operand (the
// activations, in registers) meets the HT A operands of its k-step; the
// 2 (HT - 1) accumulators are independent, so neither the MFMA latency nor the
// LDS latency of a single operand sits on the critical path, and only the
// last four k-steps depend on the runtime count ks_n (n_dim + 1 >
// 16 (DT - 1) + 1): the others form one basic block.  The accumulators are
// the output registers themselves.
//
// The hidden sizes (100, 50, 20, 1) leave 4, 2, 4 and 1 units in the last
// 16-wide tile.  Those are computed with v_mfma_f64_4x4x4_4b_f64 instead of a
// full 16x16x4 tile: four 4x4x4 blocks = the same 4 units for 4 x 4 points,
// 16 cycles instead of 64.  Register layout (measured on gfx950): A lane
// i + 4b + 16k, B lane p + 16k, D lane p + 16i with p = 4b + j -- i.e. the B
// operand is the one of the 16x16x4 instruction and D is its register 0, so
// the result is directly k-step 4*(HT-1) of the next layer.  The A operand is
// gathered from the unchanged tile-major weights (element (kk, hh) of the
// last tile, hh = lane & 3).
//
// A layer whose weights exceed an LDS region (layer 1 for n_dim > 64) runs in
// K chunks [KS_LO, KS_HI): `out` carries the pre-activations between them, w
// is the chunk in LDS (k-tile index relative to KS_LO / 4).
#ifndef NB_PRE_DT
#define NB_PRE_DT 3
#endif
#ifndef NB_SPLIT_LO
#define NB_SPLIT_LO 8
#endif
#ifndef NB_SPLIT_4
#define NB_SPLIT_4 2
#endif
#ifndef NB_SPLIT_HI
#define NB_SPLIT_HI 8
#endif
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// output tiles [H0, H1) of a layer (+ the partial last tile if H1 == HT)
template <int T, int KSMAX, int HT, bool RELU, int KS_LO, int KS_HI, int H0,
          int H1, int NIN, int NOUT>
__device__ __forceinline__ void mlp_block(const double* w, int ks_n,
                                          const double (&in)[T][NIN], int lane,
                                          double (&out)[T][NOUT]) {
  static_assert(KS_LO % 4 == 0, "chunks start at a k-tile boundary");
  constexpr bool FIRST = (KS_LO == 0), LAST = (KS_HI == KSMAX);
  constexpr int NFL = HT - 1;                 // full 16-unit tiles of the layer
  constexpr bool REM = (H1 == HT);
  constexpr int NF = (REM ? NFL : H1) - H0;   // full tiles of this block
  constexpr int NA = NF + (REM ? 1 : 0);      // A operands per k-step
  nb_d4 acc[T][NF > 0 ? NF : 1];
  double rem[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[t][h][r] = FIRST ? 0.0 : out[t][4 * (H0 + h) + r];
    rem[t] = (FIRST || !REM) ? 0.0 : out[t][4 * NFL];
  }
  const int roff = NFL * NB_TILE + (lane >> 4) * 16 + (lane & 3);
  auto read_a = [&](int ks, double (&a)[NA]) {
    const double* wk = w + ((ks - KS_LO) >> 2) * HT * NB_TILE + (ks & 3) * 64;
#pragma unroll
    for (int h = 0; h < NF; ++h) {
      a[h] = wk[(H0 + h) * NB_TILE + lane];
    }
    if constexpr (REM) a[NF] = wk[roff];
  };
  auto step = [&](int ks, const double (&a)[NA]) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t][h] = MFMA(a[h], in[t][ks], acc[t][h]);
    if constexpr (REM) {
#pragma unroll
      for (int t = 0; t < T; ++t) rem[t] = MFMA4(a[NF], in[t][ks], rem[t]);
    }
  };
  // Software pipeline over the unguarded k-steps: wait for the operands of
  // step ks (read one step ago), issue the reads of step ks + 1, then the
  // MFMAs of step ks.  The empty asm is a use of the operands, so that the
  // compiler's s_waitcnt lands BEFORE the next reads are issued (otherwise it
  // waits for those as well); sched_barrier pins the order.  (Prefetch
  // distances > 1 were measured: no gain, the extra operand registers spill.)
  auto arrived = [&](const double (&a)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) asm volatile("" ::"v"(a[i]));
  };
  constexpr int KS_U = (KS_HI < KSMAX - 4) ? KS_HI : KSMAX - 4;
  constexpr int PD = 1, R = PD + 1;
  if constexpr (KS_LO < KS_U) {
    double a[R][NA];
#pragma unroll
    for (int i = 0; i < PD; ++i)
      if (KS_LO + i < KS_U) read_a(KS_LO + i, a[i % R]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = KS_LO; ks < KS_U; ++ks) {
      arrived(a[(ks - KS_LO) % R]);
      __builtin_amdgcn_sched_barrier(0);
      if (ks + PD < KS_U) read_a(ks + PD, a[(ks - KS_LO + PD) % R]);
      __builtin_amdgcn_sched_barrier(0);
      step(ks, a[(ks - KS_LO) % R]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int ks = (KS_LO > KS_U ? KS_LO : KS_U); ks < KS_HI; ++ks) {
    if (ks < ks_n) {
      double a[NA];
      read_a(ks, a);
      step(ks, a);
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int h = 0; h < NF; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[t][4 * (H0 + h) + r] =
            (LAST && RELU) ? fmax(acc[t][h][r], 0.0) : acc[t][h][r];
    if constexpr (REM)
      out[t][4 * NFL] = (LAST && RELU) ? fmax(rem[t], 0.0) : rem[t];
  }
}

// a layer = blocks of SPLIT output tiles (fewer live accumulators per block,
// more operand reuse and independent accumulators within it)
template <int T, int SPLIT, int KSMAX, int HT, bool RELU, int KS_LO, int KS_HI,
          int H0 = 0, int NIN, int NOUT>
__device__ __forceinline__ void mlp_layer(const double* w, int ks_n,
                                          const double (&in)[T][NIN], int lane,
                                          double (&out)[T][NOUT]) {
  if constexpr (HT - 1 - H0 > SPLIT) {
    mlp_block<T, KSMAX, HT, RELU, KS_LO, KS_HI, H0, H0 + SPLIT>(
        w, ks_n, in, lane, out);
    mlp_layer<T, SPLIT, KSMAX, HT, RELU, KS_LO, KS_HI, H0 + SPLIT>(
        w, ks_n, in, lane, out);
  } else {
    mlp_block<T, KSMAX, HT, RELU, KS_LO, KS_HI, H0, HT>(w, ks_n, in, lane,
                                                        out);
  }
}

// the TPW tiles of a wavefront through one layer (+ the constant-1 unit that
// feeds the next layer's bias row: k-step ONE_KS, lane group ONE_LG); the
// gather may have left the wavefront a single tile
template <int TPW, int SPLIT, int KSMAX, int HT, bool RELU, int ONE_KS,
          int ONE_LG, int KS_LO = 0, int KS_HI = KSMAX, int NIN, int NOUT>
__device__ __forceinline__ void mlp_tiles(const double* w, int ks_n,
                                          const double (&in)[TPW][NIN],
                                          int lane, double (&out)[TPW][NOUT],
                                          bool two_tiles) {
  if constexpr (TPW == 2) {
    if (two_tiles) {
      mlp_layer<2, SPLIT, KSMAX, HT, RELU, KS_LO, KS_HI>(w, ks_n, in, lane,
                                                         out);
    } else {
      // branch-dependent element stores would turn the activation arrays into
      // scratch memory: compute into a copy, merge with selects
      double in1[1][NIN], out1[1][NOUT];
#pragma unroll
      for (int i = 0; i < NIN; ++i) in1[0][i] = in[0][i];
#pragma unroll
      for (int i = 0; i < NOUT; ++i) out1[0][i] = out[0][i];
      mlp_layer<1, SPLIT, KSMAX, HT, RELU, KS_LO, KS_HI>(w, ks_n, in1, lane,
                                                         out1);
#pragma unroll
      for (int i = 0; i <= 4 * (HT - 1); ++i) out[0][i] = out1[0][i];
    }
  } else {
    mlp_layer<1, SPLIT, KSMAX, HT, RELU, KS_LO, KS_HI>(w, ks_n, in, lane,
                                                       out);
  }
  if constexpr (KS_HI == KSMAX) {
    // registers 1..3 of the last tile: zero padding, except the constant 1
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int r = 1; r < 4; ++r) {
        const int idx = 4 * (HT - 1) + r;
        out[t][idx] = (idx == ONE_KS && (lane >> 4) == ONE_LG) ? 1.0 : 0.0;
      }
    if constexpr (ONE_KS >= 0 && ONE_KS == 4 * (HT - 1)) {
      // the constant shares the k-step of the partial tile (50 = 48 + 2)
      if ((lane >> 4) == ONE_LG) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) out[t][ONE_KS] = 1.0;
      }
    }
  }
}

// asynchronous global -> LDS copy (global_load_lds_dwordx4): every wavefront
// moves 1 KB chunks (LDS destination = wave-uniform base + lane * 16); the
// data is complete after s_waitcnt vmcnt(0) + workgroup barrier
typedef const void __attribute__((address_space(1))) * nb_gptr;
typedef void __attribute__((address_space(3))) * nb_lptr;
template <int NW>
__device__ __forceinline__ void dma_weights(const nb_gd* __restrict__ src,
                                            double* dst, int n_doubles,
                                            int wave, int lane) {
  for (int c = wave * 128; c < n_doubles; c += NW * 128)
    __builtin_amdgcn_global_load_lds((nb_gptr)(src + c + 2 * lane),
                                     (nb_lptr)(dst + c), 16, 0, 0);
}

// cooperative global -> LDS copy by the whole workgroup (16 bytes per lane)
template <int NW>
__device__ __forceinline__ void stage_weights(const nb_gd* __restrict__ src,
                                              double* dst, int n_doubles) {
  for (int i = 2 * threadIdx.x; i < n_doubles; i += 2 * 64 * NW) {
    const double2 v = *(const NB_G double2*)(src + i);
    *(double2*)(dst + i) = v;
  }
}

// COMPACT: the points of the workgroup that need the emulator are gathered
// into dense 16-point tiles through LDS before the MLP (shell exclusion and
// association only need it for a minority of the points; without the
// gather a wavefront evaluates all 32 of its points if one needs it).
// VARIANT 0 gather, 1 dense + DMA double buffer.  NW wavefronts x TPW tiles =
// 128 points per workgroup pass: 4 x 2 (every A operand feeds two tiles) or
// 8 x 1 (two wavefronts per SIMD hide each other's LDS and barrier waits).
template <int DT, int VARIANT, int TPW, int NW, bool SPARSE>
__global__ void __launch_bounds__(64 * NW)
nb_eval_kernel(EvalArgs a, int w_doubles) {
  // SPARSE: containment / association (MODE_ANY, MODE_ASSOC), where few
  // points reach an emulator; otherwise proposals, overlap counts and scores.
  // Compile-time so that neither instantiation carries the other's state.
  const bool m_any = SPARSE && a.mode == MODE_ANY;
  const bool m_assoc = SPARSE && a.mode == MODE_ASSOC;
  const bool m_sample = !SPARSE && a.mode == MODE_SAMPLE;
  const bool m_count = !SPARSE && a.mode == MODE_COUNT;
  const bool m_score = !SPARSE && a.mode == MODE_SCORE;
  constexpr bool COMPACT = (VARIANT == 0);
  constexpr bool DBUF = (VARIANT == 1);
  constexpr int DP = 16 * DT;
  constexpr int KS1MAX = 4 * DT + 1;
  // output tiles per block of the MLP layers (register budget, see mlp_layer)
  constexpr int SPLIT = DT <= 2 ? NB_SPLIT_LO : (DT <= 4 ? NB_SPLIT_4 : NB_SPLIT_HI);
  constexpr int TS = 4 * KS1MAX + 1;        // LDS row stride of a gathered point
  extern __shared__ __attribute__((aligned(16))) double wlds[];
  __shared__ int wcnt[NW];
  double* tlds = wlds + w_doubles;          // [128][TS] gathered inputs
  double* slds = tlds + 128 * TS;           // [128] scores
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const long long n_super = (a.n + 16 * NW * TPW - 1) / (16 * NW * TPW);
  unsigned long long cnt_outer = 0, cnt_ell = 0, cnt_mlp = 0;

  // (blob pointers typed as global memory: nb_common.h, NB_G)
  const nb_gd* const NB_G* blobs = (const nb_gd* const NB_G*)a.blobs;
  const double* blob0 = (const double*)blobs[0];
  const int n_dim = (int)nb_hdr(blob0, NB_H_NDIM);
  const int ks1 = (n_dim + 1 + 3) >> 2;

#ifdef NB_DBG_TIMING
  long long t_prev = clock64();
#define NB_TS(i) do { if (a.counters != nullptr && threadIdx.x == 0 && blockIdx.x == 0) { \
    const long long t_now = clock64(); a.counters[8 + (i)] += t_now - t_prev; t_prev = t_now; } } while (0)
#else
#define NB_TS(i)
#endif
  for (long long sup = blockIdx.x; sup < n_super; sup += gridDim.x) {
    long long pt[TPW];
    bool valid[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      pt[t] = ((sup * NW + wave) * TPW + t) * 16 + (lane & 15);
      valid[t] = pt[t] < a.n;
    }

    bool hit[TPW];
    int hit_idx[TPW], count_out[TPW];
    unsigned char flags[TPW];
    double r2_out[TPW], score_out[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      hit[t] = false; hit_idx[t] = -1; count_out[t] = 0; flags[t] = 0;
      r2_out[t] = 0.0; score_out[t] = 0.0;
    }

    for (int b = 0; b < a.nb; ++b) {
      const nb_gd* blob_g = blobs[b];
      const double* blob = (const double*)blob_g;
      const int K = (int)nb_hdr(blob, NB_H_K);
      const int M = (int)nb_hdr(blob, NB_H_M);
      const int E = (int)nb_hdr(blob, NB_H_E);
      const bool use_cube = nb_hdr(blob, NB_H_USECUBE) != 0;
      const long long ell_stride = nb_hdr(blob, NB_H_ELL_STRIDE);
      const long long neural_stride = nb_hdr(blob, NB_H_NEURAL_STRIDE);
      // contains() of a bound with periodic dimensions sees recentred points;
      // proposals (SAMPLE / COUNT / SCORE) already live in the shifted frame
      const long long off_shift = nb_hdr(blob, NB_H_OFF_SHIFT);
      const double* shift =
          (off_shift != 0 && SPARSE)
              ? blob + off_shift : nullptr;

      // Dense variant, proposals and scores (nearly every point reaches the
      // emulator): the ellipsoid block of the first neural bound goes to
      // region B and layer 1 of its first network to region A by DMA, issued
      // before the points are loaded so that all three overlap.  Only if no
      // outer member is staged in between, and only for n_dim <= 48 (beyond,
      // keeping the points in registers until the ellipsoid test spills:
      // measured 2 % slower at n_dim = 50).
      const nb_gd* nblk = blob_g + nb_hdr(blob, NB_H_OFF_NEURAL);
      const int kt1 = (int)nb_hdr(blob, NB_H_KT1);
      const int n_a = kt1 * NB_HT1 * NB_TILE;                   // layer 1
      constexpr int KA = (DT + 2) / 2;                  // k-tiles of chunk A
      const int n_a0 = (DT >= 5) ? KA * NB_HT1 * NB_TILE : n_a;
      // a neural bound's block is staged together with what follows it in
      // the blob: threshold, mean and inverse scale of the standardisation
      constexpr int NBLK = nb_ell_block_size(DT) + 2 + 2 * DP;
      // (whole 1 KB DMA chunks into region B, 38 tiles)
      const bool ell_dma = DBUF && ((NBLK + 127) / 128) * 128 <= 38 * NB_TILE;
      const bool early = ell_dma && M > 0 && E > 0 &&
                         (m_sample || m_score);
      const bool pre = DT <= NB_PRE_DT && early &&
                       (K == 0 || (m_sample && K == 1));
      if (pre) {
        __syncthreads();                               // LDS free
        dma_weights<NW>(nblk, tlds, NBLK, wave, lane);
        dma_weights<NW>(nblk + nb_ell_block_size(DT) + 2 + 2 * DP, wlds, n_a0,
                        wave, lane);
      }

      // unit-cube clip of the union (union.py:287-288 / 313-314)
      bool in_cube[TPW], active[TPW];
      int k_cnt[TPW];
      double xin[TPW][4 * DT];
      {
        load_points<DT, TPW>((const nb_gd*)a.x, pt, valid, n_dim, a.n, lane, xin, shift);

        // Bounding-sphere pre-test (shell exclusion / association): a point
        // of a bound with neural bounds lies inside one of their ellipsoids,
        // hence within sqrt(radius2) of that centre.  If no undecided point
        // of the workgroup passes this test the whole bound is skipped --
        // for nested bounds in high dimension all but the next few bounds.
        if (SPARSE && M > 0) {
          const nb_gd* nblk0 = blob_g + nb_hdr(blob, NB_H_OFF_NEURAL);
          bool maybe = false;
          for (int m = 0; m < M; ++m) {
            const nb_gd* nb_m = nblk0 + m * neural_stride;
            const double rad2 = nb_m[1];
            const nb_gd* cc = nb_m + 2 + 2 * DP;
            double d2[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) d2[t] = 0.0;
#pragma unroll
            for (int ks = 0; ks < 4 * DT; ++ks) {
              const double cv = cc[4 * ks + lg];
#pragma unroll
              for (int t = 0; t < TPW; ++t) {
                const double dv = xin[t][ks] - cv;
                d2[t] = fma(dv, dv, d2[t]);
              }
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
              maybe |= valid[t] && !hit[t] && lane_group_sum(d2[t]) <= rad2;
          }
          if (!__syncthreads_or(maybe ? 1 : 0)) continue;
        }

        // unit-cube limits of slot (ks, lg) = feature 8 (ks >> 1) + 2 lg +
        // (ks & 1): [0, 1) for the features of a bound clipped to the cube,
        // unbounded otherwise (what nb_api.hip stores at off_ulo / off_uhi;
        // computed here, the loads were serialised by the compiler)
        bool cbad[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) cbad[t] = false;
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) {
          const int f = 8 * (ks >> 1) + 2 * lg + (ks & 1);
          const bool boxed = use_cube && f < n_dim;
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            cbad[t] |= boxed && !(xin[t][ks] >= 0.0 && xin[t][ks] < 1.0);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          in_cube[t] = !point_any(cbad[t], lane);
          active[t] = valid[t] && !hit[t];
        }

      // ---- outer union: overlap count --------------------------------
#pragma unroll
      for (int t = 0; t < TPW; ++t) k_cnt[t] = 0;
      const nb_gd* mblk = blob_g + nb_hdr(blob, NB_H_OFF_MEMBERS);
      if (m_sample && K == 1) {
#pragma unroll
        for (int t = 0; t < TPW; ++t) k_cnt[t] = 1;   // drawn from the only member
      } else {
        for (int m = 0; m < K; ++m) {
          double y[TPW][4 * DT], r2[TPW];
          bool box_bad[TPW];
          // the member's limits, centre and B_inv tiles are staged in LDS and
          // shared by the 8 tiles of the workgroup
          __syncthreads();
          stage_weights<NW>(mblk + m * ell_stride, wlds, nb_ell_block_size(DT));
          __syncthreads();
          ell_eval<DT, TPW>(wlds, n_dim, xin, lane, y, box_bad, r2);
#pragma unroll
          for (int t = 0; t < TPW; ++t)
            k_cnt[t] += (!box_bad[t] && r2[t] < 1.0) ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
          cnt_outer += (unsigned long long)K *
                       __popcll(__ballot(active[t] && lg == 0));
      }
      }

      NB_TS(0);
      bool outer_ok[TPW], acc_outer[TPW], neural_ok[TPW], want[TPW];
      bool any_want = false;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        outer_ok[t] = in_cube[t] && (K == 0 || k_cnt[t] > 0);
        // acceptance of the overlap-corrected union draw (union.py:318-319)
        acc_outer[t] = false;
        if (m_sample) {
          double u0, u_acc;
          nb_uniform_pair(a.seed, a.offset + (unsigned long long)pt[t], 0u,
                          NB_TAG_CTRL, u0, u_acc);
          acc_outer[t] = in_cube[t] &&
                         (u_acc > 1.0 - 1.0 / (double)k_cnt[t]);
        }
        neural_ok[t] = (M == 0);
        if (m_sample) want[t] = valid[t] && acc_outer[t];
        else if (m_score) want[t] = valid[t];
        else if (m_count) want[t] = false;
        else want[t] = active[t] && outer_ok[t];
        any_want |= want[t];
      }

      // ---- neural bounds (workgroup-uniform control flow) --------------
      if (M > 0 && __syncthreads_or(any_want ? 1 : 0)) {
        const long long net_stride = nb_hdr(blob, NB_H_NET_STRIDE);
        const int n_b = (NB_HT1 * NB_HT2 + NB_HT2 * NB_HT3 + NB_HT3) * NB_TILE;
        for (int m = 0; m < M; ++m) {
          const nb_gd* nb_m = nblk + m * neural_stride;
          double y[TPW][4 * DT], r2[TPW];
          bool box_bad[TPW], inside_e[TPW], need[TPW];
          const nb_gd* nets =
              nb_m + nb_ell_block_size(DT) + 2 + 2 * DP;
          if (ell_dma) {
            // (m == 0 of a pre-issued bound: copies in flight, points loaded)
            if (!(pre && m == 0)) {
              __syncthreads();                         // LDS free
              dma_weights<NW>(nb_m, tlds, NBLK, wave, lane);
              if (early) dma_weights<NW>(nets, wlds, n_a0, wave, lane);
              load_points<DT, TPW>((const nb_gd*)a.x, pt, valid, n_dim, a.n, lane, xin,
                                   shift);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            ell_eval<DT, TPW>(tlds, n_dim, xin, lane, y, box_bad, r2);
          } else {
            load_points<DT, TPW>((const nb_gd*)a.x, pt, valid, n_dim, a.n, lane, xin, shift);
            __syncthreads();
            stage_weights<NW>(nb_m, wlds, NBLK);
            __syncthreads();
            ell_eval<DT, TPW>(wlds, n_dim, xin, lane, y, box_bad, r2);
          }
          // (in LDS until the first weight DMA of this neural bound lands)
          const double* blk_lds = ell_dma ? tlds : wlds;
          NB_TS(1);
          bool wave_need = false;
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            inside_e[t] = !box_bad[t] && r2[t] < 1.0;
            need[t] = want[t] && inside_e[t] && !neural_ok[t];
            if (m_score) need[t] = valid[t];
            cnt_ell += __popcll(__ballot(want[t] && lg == 0));
            cnt_mlp += (unsigned long long)E *
                       __popcll(__ballot(need[t] && lg == 0));
            wave_need |= need[t];
            if (m == 0) r2_out[t] = r2[t];
          }
          wave_need = __any(wave_need);
          bool ok[TPW];
#pragma unroll
          for (int t = 0; t < TPW; ++t) ok[t] = inside_e[t];

          // workgroup census of the points that need the emulator
          unsigned long long bal[TPW];
          int cnt_w = 0;
#pragma unroll
          for (int t = 0; t < TPW; ++t) {
            bal[t] = __ballot(need[t] && lg == 0);
            cnt_w += __popcll(bal[t]);
          }
          int base = 0, n_need = 0;
          if constexpr (COMPACT) {
            __syncthreads();
            if (lane == 0) wcnt[wave] = cnt_w;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < NW; ++w) {
              if (w < wave) base += wcnt[w];
              n_need += wcnt[w];
            }
          } else {
            // (also the barrier between the ellipsoid block's readers and the
            // first weight DMA into its region)
            n_need = __syncthreads_or(cnt_w > 0 ? 1 : 0);
          }

          if (E > 0 && n_need > 0) {
            const double thr = blk_lds[nb_ell_block_size(DT)];
            const double* mean = blk_lds + nb_ell_block_size(DT) + 2;
            const double* scale = mean + DP;
            // standardised input (neural.py:115), constant 1 at column D
            double tin[TPW][KS1MAX];
#pragma unroll
            for (int ks = 0; ks < 4 * DT; ++ks) {
              const int f = 4 * ks + lg;
              const double mv = mean[f], sv = scale[f];     // sv = 1 / scale
#pragma unroll
              for (int t = 0; t < TPW; ++t)
                tin[t][ks] = (f < n_dim) ? (y[t][ks] - mv) * sv
                                         : ((f == n_dim) ? 1.0 : 0.0);
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
              tin[t][4 * DT] = (16 * DT + lg == n_dim) ? 1.0 : 0.0;

            bool wave_mlp = wave_need;
            bool two_tiles = true;
            int cidx[TPW];
            if (COMPACT) {
              // gather: point -> dense slot base + rank
              int off = base;
#pragma unroll
              for (int t = 0; t < TPW; ++t) {
                const unsigned long long below =
                    bal[t] & ((1ull << (lane & 15)) - 1ull);
                cidx[t] = off + __popcll(below);
                off += __popcll(bal[t]);
                if (need[t]) {
#pragma unroll
                  for (int ks = 0; ks < KS1MAX; ++ks)
                    tlds[cidx[t] * TS + 4 * ks + lg] = tin[t][ks];
                }
              }
              __syncthreads();
              const int n_ct = (n_need + 15) >> 4;     // dense tiles
#pragma unroll
              for (int t = 0; t < TPW; ++t) {
                const int q = wave + NW * t;
                const int slot = 16 * q + (lane & 15);
                const bool on = q < n_ct && slot < n_need;
#pragma unroll
                for (int ks = 0; ks < KS1MAX; ++ks)
                  tin[t][ks] = on ? tlds[slot * TS + 4 * ks + lg] : 0.0;
              }
              wave_mlp = wave < n_ct;
              two_tiles = wave + NW < n_ct;
            }

            double total[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) total[t] = 0.0;
            if constexpr (DBUF) {
              // dense path: the weights stream through two LDS regions of 38
              // tiles; while one feeds the matrix cores the other is refilled
              // by DMA (one barrier per stage).  Stages of a network: layer 1
              // (two K chunks for n_dim > 64, where it exceeds a region), then
              // layers 2-4.
              constexpr bool TWO = (DT >= 5);
              constexpr int NST = TWO ? 3 : 2;
              double* reg[2] = {wlds, tlds};
              double h1[TPW][4 * NB_HT1];
              auto issue = [&](int e, int st, int q) {
                const nb_gd* w1 = nets + e * net_stride;
                if (st == 0) dma_weights<NW>(w1, reg[q & 1], n_a0, wave, lane);
                else if (TWO && st == 1)
                  dma_weights<NW>(w1 + n_a0, reg[q & 1], n_a - n_a0, wave, lane);
                else dma_weights<NW>(w1 + n_a, reg[q & 1], n_b, wave, lane);
              };
              if (!early) {
                __syncthreads();
                issue(0, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
              } else {
                // the standardisation vectors were read from region B, which
                // the first stage refills
                __syncthreads();
              }
              NB_TS(2);
              int q = 0;
              for (int e = 0; e < E; ++e) {
#pragma unroll
                for (int st = 0; st < NST; ++st, ++q) {
                  if (st + 1 < NST) issue(e, st + 1, q + 1);
                  else if (e + 1 < E) issue(e + 1, 0, q + 1);
                  const double* cur = reg[q & 1];
                  if (wave_mlp) {
                    if (st == 0) {
                      if constexpr (TWO)
                        mlp_tiles<TPW, SPLIT, KS1MAX, NB_HT1, true, 25, 0, 0,
                                  4 * KA>(cur, ks1, tin, lane, h1, true);
                      else
                        mlp_tiles<TPW, SPLIT, KS1MAX, NB_HT1, true, 25, 0>(
                            cur, ks1, tin, lane, h1, true);  // unit 100
                    } else if (TWO && st == 1) {
                      if constexpr (TWO)
                        mlp_tiles<TPW, SPLIT, KS1MAX, NB_HT1, true, 25, 0,
                                  4 * KA, KS1MAX>(cur, ks1, tin, lane, h1, true);
                    } else {
                      const double* w2 = cur;
                      const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
                      const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
                      double h2[TPW][4 * NB_HT2], h3[TPW][4 * NB_HT3],
                          o[TPW][4];
                      mlp_tiles<TPW, SPLIT, 26, NB_HT2, true, 12, 2>(
                          w2, 26, h1, lane, h2, true);     // unit 50
                      mlp_tiles<TPW, SPLIT, 13, NB_HT3, true, 5, 0>(
                          w3, 13, h2, lane, h3, true);     // unit 20
                      mlp_tiles<TPW, SPLIT, 6, 1, false, -1, 0>(
                          w4, 6, h3, lane, o, true);
#pragma unroll
                      for (int t = 0; t < TPW; ++t) total[t] += o[t][0];
                    }
                  }
                  NB_TS(3 + 2 * (q & 1));
                  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                  __syncthreads();
                  NB_TS(4 + 2 * (q & 1));
                }
              }
            } else {
              for (int e = 0; e < E; ++e) {
                const nb_gd* w1 = nets + e * net_stride;
                double h1[TPW][4 * NB_HT1];
                __syncthreads();                       // LDS free
                stage_weights<NW>(w1, wlds, n_a);
                __syncthreads();
                if (wave_mlp)
                  mlp_tiles<TPW, SPLIT, KS1MAX, NB_HT1, true, 25, 0>(
                      wlds, ks1, tin, lane, h1, two_tiles);
                __syncthreads();
                stage_weights<NW>(w1 + n_a, wlds, n_b);
                __syncthreads();
                if (wave_mlp) {
                  const double* w2 = wlds;
                  const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
                  const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
                  double h2[TPW][4 * NB_HT2], h3[TPW][4 * NB_HT3], o[TPW][4];
                  mlp_tiles<TPW, SPLIT, 26, NB_HT2, true, 12, 2>(
                      w2, 26, h1, lane, h2, two_tiles);
                  mlp_tiles<TPW, SPLIT, 13, NB_HT3, true, 5, 0>(
                      w3, 13, h2, lane, h3, two_tiles);
                  mlp_tiles<TPW, SPLIT, 6, 1, false, -1, 0>(
                      w4, 6, h3, lane, o, two_tiles);
                  total[0] += o[0][0];             // unit 0 lives in lg == 0
                  if constexpr (TPW == 2) {
                    if (two_tiles) total[1] += o[1][0];
                  }
                }
              }
            }
            if (COMPACT) {
              // scatter the scores back to the owners of the points
#pragma unroll
              for (int t = 0; t < TPW; ++t)
                if (lg == 0) slds[16 * (wave + NW * t) + (lane & 15)] = total[t];
              __syncthreads();
#pragma unroll
              for (int t = 0; t < TPW; ++t) {
                const double score =
                    need[t] ? slds[cidx[t]] / (double)E : 0.0;
                if (need[t]) ok[t] = inside_e[t] && (score > thr);
                if (m == 0) score_out[t] = score;
              }
            } else {
#pragma unroll
              for (int t = 0; t < TPW; ++t) {
                const double score =
                    __shfl(total[t], lane & 15) / (double)E;
                if (need[t]) ok[t] = inside_e[t] && (score > thr);
                if (m == 0) score_out[t] = score;
              }
            }
          }
#pragma unroll
          for (int t = 0; t < TPW; ++t) neural_ok[t] |= ok[t];
          // the points are re-read for the next member: end their live range
          // here so that they do not occupy registers during the MLP
#pragma unroll
          for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int ks = 0; ks < 4 * DT; ++ks)
              asm volatile("" : "=v"(xin[t][ks]));
        }
      }

      bool all_done = true;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const bool contained = outer_ok[t] && neural_ok[t];
        if (SPARSE) {
          if (active[t] && contained) { hit[t] = true; hit_idx[t] = b; }
          all_done &= (hit[t] || !valid[t]);
        } else if (m_sample) {
          flags[t] = (acc_outer[t] ? 1 : 0) |
                     ((acc_outer[t] && neural_ok[t]) ? 2 : 0);
        } else if (m_count) {
          count_out[t] = k_cnt[t];
        }
      }
      if (SPARSE) {
        if (__syncthreads_and(all_done ? 1 : 0)) break;
      }
    }

    NB_TS(7);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      if (valid[t] && lg == 0) {
        if (m_any) a.out_u8[pt[t]] = hit[t] ? 1 : 0;
        else if (m_assoc) a.out_i32[pt[t]] = hit_idx[t];
        else if (m_sample) a.out_u8[pt[t]] = flags[t];
        else if (m_count)
          a.out_u8[pt[t]] = (unsigned char)count_out[t];
        else if (m_score) {
          a.out_f64[2 * pt[t]] = r2_out[t];
          a.out_f64[2 * pt[t] + 1] = score_out[t];
        }
      }
    }
  }
  NB_TS(8);
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[0], cnt_outer);
    atomicAdd(&a.counters[1], cnt_ell);
    atomicAdd(&a.counters[2], cnt_mlp);
  }
}

template <int DT, int VARIANT, int TPW, int NW, bool SPARSE>
int launch_eval_impl(const EvalArgs& a, int lds_tiles, hipStream_t stream) {
  constexpr bool COMPACT = (VARIANT == 0);
  constexpr int TS = 4 * (4 * DT + 1) + 1;
  const int w_doubles = lds_tiles * NB_TILE;
  // gather variant: [weights stage][128 gathered inputs][128 scores];
  // dense variant:  [region A: layer 1 / ellipsoid block][region B: layers 2-4]
  const size_t lds = ((size_t)w_doubles +
                      (COMPACT ? 128 * TS + 128
                               : (VARIANT == 1 ? 38 * NB_TILE : 0))) *
                     sizeof(double);
  static size_t lds_allowed = 0;
  if (lds > lds_allowed) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_eval_kernel<DT, VARIANT, TPW, NW, SPARSE>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    lds_allowed = lds;
  }
  (void)hipGetLastError();
  const long long n_super = (a.n + 16 * NW * TPW - 1) / (16 * NW * TPW);
  // the kernel uses the whole register file (one wave per SIMD): one
  // workgroup per CU, grid-stride over the 128-point super tiles
  long long blocks = n_super;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((nb_eval_kernel<DT, VARIANT, TPW, NW, SPARSE>),
                     dim3((unsigned)blocks), dim3(64 * NW), lds, stream, a,
                     w_doubles);
  return NB_OK;
}

template <int DT>
int launch_eval(const EvalArgs& a, int kt1_max, hipStream_t stream) {
  // region A: a whole layer 1 (n_dim <= 64) or one of its two K chunks; the
  // ellipsoid block of large n_dim may extend into region B (it is only
  // needed while no weights are staged)
  int lds_tiles = 38;
  if (DT <= 4 && kt1_max * NB_HT1 > lds_tiles) lds_tiles = kt1_max * NB_HT1;
  constexpr int ELL_TILES = (2 + 48 * DT + 256 * DT * DT + NB_TILE - 1) / NB_TILE;
  static_assert(ELL_TILES <= 76, "ellipsoid block must fit the two regions");
  const int gather_tiles = (lds_tiles > ELL_TILES) ? lds_tiles : ELL_TILES;
  constexpr int TS = 4 * (4 * DT + 1) + 1;
  // Gathered variant (emulator inputs compacted through LDS): pays off when
  // only a small minority of a workgroup's points reaches an emulator.  In
  // the benchmark's shell exclusion the dense variant with its overlapped
  // weight streaming is 3 % faster end to end, so it is the default (the
  // gathered instantiation is only built with -DNB_EVAL_GATHER, `make debug`).
  const size_t need = ((size_t)gather_tiles * NB_TILE + 128 * TS + 128) * 8 + 64;
  const bool sparse = (a.mode == MODE_ANY || a.mode == MODE_ASSOC);
  // two tiles per wavefront up to n_dim = 64; beyond that the per-lane state
  // (y, standardised input, hidden activations of two tiles) no longer fits
  // the register file (one tile, see below)
  constexpr int TPW = 2;
#ifdef NB_EVAL_GATHER
  if constexpr (DT <= 4) {
    if (sparse && need <= 160 * 1024)
      return launch_eval_impl<DT, 0, TPW, 4, true>(a, gather_tiles, stream);
  }
#else
  (void)need;
#endif
  // (8 wavefronts x 1 tile was measured as well: +7 % at D = 20, -2 % at
  // D = 50, where the 256-register budget per wavefront forces ~100 spills)
  // n_dim > 64: one tile per wavefront, and there the weight stream decides
  // (174 KB of DMA per network against ~12 B/clk a CU sustains): eight
  // wavefronts share every staged weight byte between 128 points instead of
  // 64.  Measured +9 % at D = 100, +13 % at D = 80, +8 % at D = 65 over four
  // wavefronts although the 256-register budget spills 130-330 registers of
  // cold per-point state.
  if constexpr (DT >= 5) {
    if (sparse) return launch_eval_impl<DT, 1, 1, 8, true>(a, lds_tiles, stream);
    return launch_eval_impl<DT, 1, 1, 8, false>(a, lds_tiles, stream);
  } else {
    if (sparse)
      return launch_eval_impl<DT, 1, TPW, 4, true>(a, lds_tiles, stream);
    return launch_eval_impl<DT, 1, TPW, 4, false>(a, lds_tiles, stream);
  }
}

}  // namespace

static unsigned long long* g_eval_counters = nullptr;
void nb_eval_set_counters(unsigned long long* dev) { g_eval_counters = dev; }
unsigned long long* nb_eval_counters() { return g_eval_counters; }

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
  const int kt1 = dt + 1;     // upper limit of ceil((D + 1) / 16)
  int rc = NB_OK;
  switch (dt) {
    case 1: rc = launch_eval<1>(a, kt1, stream); break;
    case 2: rc = launch_eval<2>(a, kt1, stream); break;
    case 3: rc = launch_eval<3>(a, kt1, stream); break;
    case 4: rc = launch_eval<4>(a, kt1, stream); break;
    case 5: rc = launch_eval<5>(a, kt1, stream); break;
    case 6: rc = launch_eval<6>(a, kt1, stream); break;
    case 7: rc = launch_eval<7>(a, kt1, stream); break;
    case 8: rc = launch_eval<8>(a, kt1, stream); break;
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
