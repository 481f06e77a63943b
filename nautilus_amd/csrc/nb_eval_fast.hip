// Proposal acceptance / emulator scores of ONE bound with ONE neural bound --
// the case NautilusBound.sample spends its time in (nautilus.py:193-244: a
// proposal from the outer union is kept if the neural bound contains it,
// bounds/neural.py:115-126) -- as a software pipeline over the 128-point
// passes of a workgroup.  Same arithmetic as nb_eval.hip (bit-identical
// scores), different schedule:
//
//  * EIGHT wavefronts x one 16-point tile (two wavefronts per SIMD, 256
//    registers each, no spills up to n_dim = 63): while one wavefront of a
//    SIMD waits for LDS operands, issues a weight DMA or runs the per-pass
//    prologue, the other feeds the matrix pipe.  (Four wavefronts x two tiles
//    -- every A operand shared by two tiles, one wavefront per SIMD -- reached
//    0.66 of the fp64 MFMA peak at n_dim = 50; this shape reaches 0.74, and
//    0.70 at n_dim = 100 where the four-wavefront kernel of nb_eval.hip gets
//    0.37.)  The weight stream sets the pace -- 66 to 101 tiles per network
//    against ~12 B/clk of DMA a CU sustains -- so a staged weight byte has to
//    serve as many points as the registers allow: 128 per pass.
//  * two LDS regions of 38 tiles alternate between feeding the matrix cores
//    and being refilled by global_load_lds; the ellipsoid block (centre,
//    lower-triangular tiles, threshold, mean, inverse scale) travels through
//    them like a stage of its own: it is fetched during the last stage of
//    the previous pass, layer 1 of the first network under the per-pass
//    prologue.  Layer 1 runs in two K chunks where it exceeds a region
//    (n_dim >= 80).  Stages of a pass:
//      [ellipsoid + standardisation] ([L1] or [L1 a][L1 b], [L2-4]) x E
//  * the points of pass p + 1 are loaded raw into registers during the last
//    stage of pass p (the standardised input's registers are dead there) and
//    masked when the pass starts;
//  * the layers of a stage run as one operand pipeline with the weight DMA of
//    the next stage sliced into the k-steps (nb_mlp.h).
//
// Eligible: n_dim <= 128, one neural bound with E >= 1 networks, MODE_SAMPLE
// with at most one outer member or MODE_SCORE; everything else goes through
// nb_eval.hip.
#include "nb_common.h"

#include <type_traits>

#include "nb_mlp.h"

namespace {

constexpr int FAST_MAX_GROUPS = 1024;     // pass table in LDS: 4 KB

struct FastArgs {
  const double* blob;
  int sample;                     // 1: flags of nb_accept, 0: (r2, score)
  const nb_gd* x;
  const long long* idx;           // optional (scores): rows of x to evaluate
  int m;                          // neural bound of the blob to evaluate
  int recentre;                   // rows are in the sampler's frame: apply
                                  // the bound's periodic shift first
  long long n;
  unsigned char* out_u8;
  double* out_f64;
  unsigned long long seed;
  unsigned long long offset;
  unsigned long long* counters;   // optional, as in nb_eval.hip
  // BATCH (second stage of nb_cand.hip): the candidates of ALL (bound, neural
  // bound) groups of a query in one launch.  Group g owns the rows
  // dense[g * n_pad ... + totals[g]); a 128-point pass belongs to one group.
  const FastGroup* groups;
  const int* totals;
  const int* dense;
  long long n_pad;
  int n_groups;
  int out_mode;                   // 0: any, 1: first, 2: sample (nb_cand.hip)
  unsigned char* st;              // per row: status written for candidates
  int* first;                     //          whose score passes the threshold
};

constexpr int FAST_B_DOUBLES =
    (NB_HT1 * NB_HT2 + NB_HT2 * NB_HT3 + NB_HT3) * NB_TILE;

typedef const void __attribute__((address_space(1))) * nbf_gptr;
typedef void __attribute__((address_space(3))) * nbf_lptr;

// y = B_inv (x - c) from the staged block: centre + lower-triangular tiles
// (no box test: a neural bound's ellipsoid spans all dimensions)
template <int DT, int T>
__device__ __forceinline__ void ell_eval_centre(
    const double* c, const double* tiles, int n_dim,
    const double (&xin)[T][4 * DT], int lane, int lg, double (&y)[T][4 * DT],
    double (&r2)[T]) {
  double d[T][4 * DT];
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const double cv = c[4 * ks + lg];
#pragma unroll
    for (int t = 0; t < T; ++t) d[t][ks] = xin[t][ks] - cv;
  }
  double part[T];
#pragma unroll
  for (int t = 0; t < T; ++t) part[t] = 0.0;
  // A last row tile of at most four real rows (n_dim mod 16 in 1..4: two at
  // n_dim 50, four at 100) runs on v_mfma_f64_4x4x4_4b -- 16 instead of 64
  // cycles per k-step -- without its last two k-steps (zero padding only),
  // exactly as in nb_cand.hip's cand_inside: bit for bit register 0 of the
  // full tile, the other three are the zero rows.
  const int live = n_dim - 16 * (DT - 1);
  const bool small = DT > 1 && live >= 1 && live <= 4;
  const int lane4 = (lane >> 4) * 16 + (lane & 3);
#pragma unroll
  for (int ht = 0; ht < DT; ++ht) {
    if (ht == DT - 1 && small) {
      double r4[T];
#pragma unroll
      for (int t = 0; t < T; ++t) r4[t] = 0.0;
#pragma unroll
      for (int ks = 0; ks < 4 * (ht + 1) - 2; ++ks) {
        const int kt = ks >> 2, s = ks & 3;
        const double a =
            tiles[(ht * (ht + 1) / 2 + kt) * NB_TILE + s * 64 + lane4];
#pragma unroll
        for (int t = 0; t < T; ++t) r4[t] = NB_MFMA4(a, d[t][ks], r4[t]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        y[t][4 * ht] = r4[t];
        y[t][4 * ht + 1] = y[t][4 * ht + 2] = y[t][4 * ht + 3] = 0.0;
        part[t] += r4[t] * r4[t];
      }
    } else if (16 * ht < n_dim) {
      nb_d4 acc[T];
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4 * (ht + 1); ++ks) {
        const int kt = ks >> 2, s = ks & 3;
        const double a =
            tiles[(ht * (ht + 1) / 2 + kt) * NB_TILE + s * 64 + lane];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = MFMA(a, d[t][ks], acc[t]);
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[t][4 * ht + r] = acc[t][r];
          part[t] += acc[t][r] * acc[t][r];
        }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[t][4 * ht + r] = 0.0;
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) r2[t] = lane_group_sum(part[t]);
}

// The points of a pass as they come from memory (16-byte loads, the slot
// layout of nb_tile.h's load_points); masking happens when the pass starts,
// so that nothing waits for the loads where they are issued.
template <int DT, int T>
__device__ __forceinline__ void load_points_raw(
    const nb_gd* __restrict__ x, const long long (&row_of)[T], int n_dim,
    int lane, double2 (&raw)[T][2 * DT]) {
  // (lg opaque: the clamped column offsets below are loop invariants of the
  // pass loop; hoisted, they are 6 DT 64-bit registers that lived through the
  // whole pass -- the 300-500 bytes of scratch this kernel had beyond n_dim 64)
  int lg = lane >> 4;
  asm volatile("" : "+v"(lg));
  asm volatile("" : "+s"(x));
  if ((n_dim & 1) == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const nb_gd* row = x + row_of[t] * n_dim;
#pragma unroll
      for (int j = 0; j < 2 * DT; ++j) {
        const int f = 8 * j + 2 * lg;
        raw[t][j] = *(const NB_G double2*)(row + (f < n_dim ? f : n_dim - 2));
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const nb_gd* row = x + row_of[t] * n_dim;
#pragma unroll
      for (int j = 0; j < 2 * DT; ++j) {
        const int f = 8 * j + 2 * lg;
        raw[t][j].x = row[f < n_dim ? f : n_dim - 1];
        raw[t][j].y = row[f + 1 < n_dim ? f + 1 : n_dim - 1];
      }
    }
  }
}

template <int DT, int T>
__device__ __forceinline__ void points_from_raw(
    const double2 (&raw)[T][2 * DT], const bool (&valid)[T], int n_dim,
    int lg, double (&xin)[T][4 * DT]) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 2 * DT; ++j) {
      const int f = 8 * j + 2 * lg;
      xin[t][2 * j] = (valid[t] && f < n_dim) ? raw[t][j].x : 0.0;
      xin[t][2 * j + 1] = (valid[t] && f + 1 < n_dim) ? raw[t][j].y : 0.0;
    }
}

constexpr int FAST_REGION = 38 * NB_TILE;            // doubles per region

template <int DT, int KT1, bool BATCH>
__global__ void __launch_bounds__(512) nb_eval_fast_kernel(FastArgs a) {
  constexpr int T = 1, NW = 8, DP = 16 * DT;
  constexpr int KS1 = 4 * KT1;
  // layer 1 in one stage if it fits a region, else in two K chunks
  constexpr bool TWO = KT1 * NB_HT1 * NB_TILE > FAST_REGION;
  constexpr int KA = TWO ? (KT1 + 1) / 2 : KT1;    // k-tiles of chunk a
  constexpr int NT = DT * (DT + 1) / 2;            // lower-triangular tiles
  // the block in LDS: [c (one 1 KB piece)][NT tiles][thr, pad, mean, 1/scale].
  // (A neural bound's ellipsoid spans all dimensions -- nb_bound_create checks
  // it -- so its per-dimension box limits are infinite and stay behind.)
  constexpr int HEAD = 128;
  constexpr int TAIL = HEAD + NT * NB_TILE;
  constexpr int TC = (2 + 2 * DP + 127) / 128;     // 1 KB pieces of the tail
  constexpr int ELL_CHUNKS = 1 + 2 * NT + TC;
  static_assert(DP <= HEAD, "centre fits the head");
  static_assert(TAIL + TC * 128 <= FAST_REGION, "ellipsoid block fits a region");
  static_assert(KA * NB_HT1 * NB_TILE <= FAST_REGION, "layer-1 chunk fits");
  constexpr int NA_D = KA * NB_HT1 * NB_TILE;               // chunk a
  constexpr int NB1_D = (KT1 - KA) * NB_HT1 * NB_TILE;      // chunk b
  constexpr int NC_D = FAST_B_DOUBLES;                      // layers 2-4
  extern __shared__ __attribute__((aligned(16))) double lds[];
  auto reg = [&](int i) __attribute__((always_inline)) {
    return lds + (i & 1) * FAST_REGION;
  };

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const bool m_sample = a.sample != 0;
  const double* blob = a.blob;
  const int n_dim = (int)nb_hdr(blob, NB_H_NDIM);
  const int K = (int)nb_hdr(blob, NB_H_K);
  const bool use_cube = nb_hdr(blob, NB_H_USECUBE) != 0;
  const int ks1 = (n_dim + 1 + 3) >> 2;
  const long long net_stride = nb_hdr(blob, NB_H_NET_STRIDE);
  unsigned long long cnt_ell = 0, cnt_mlp = 0;
  long long sup = blockIdx.x;

  // ---- what a pass works on ------------------------------------------------
  // plain launches: one neural bound for all passes; BATCH: the group of the
  // pass, found in the table of pass offsets (LDS, behind the two regions)
  struct Pass {
    const double* nb;             // neural block
    const double* shift;
    int E, b;
    long long base;               // BATCH: first slot of the pass in `dense`
    int n_valid;                  // BATCH: slots of the pass that hold rows
  };
  int* pp = (int*)(lds + 2 * FAST_REGION);
  long long n_super;
  if constexpr (BATCH) {
    // pp[g] = passes of the groups before g (every workgroup, redundantly)
    if (wave == 0) {
      int run = 0;
      for (int g0 = 0; g0 < a.n_groups; g0 += 64) {
        const int g = g0 + lane;
        int v = g < a.n_groups ? (a.totals[g] + 127) >> 7 : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int u = __shfl_up(v, d);
          if (lane >= d) v += u;
        }
        if (g < a.n_groups) pp[g + 1] = run + v;
        run += __shfl(v, 63);
      }
      if (lane == 0) pp[0] = 0;
    }
    __syncthreads();
    n_super = pp[a.n_groups];
    if (sup >= n_super) return;
  } else {
    n_super = (a.n + 16 * NW * T - 1) / (16 * NW * T);
  }
  auto pass_of = [&](long long P) __attribute__((always_inline)) {
    Pass ps;
    if constexpr (BATCH) {
      const long long Pc = P < n_super ? P : n_super - 1;
      int lo = 0, hi = a.n_groups;          // largest g with pp[g] <= Pc
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pp[mid] <= (int)Pc) lo = mid; else hi = mid;
      }
      const int g = __builtin_amdgcn_readfirstlane(lo);
      const FastGroup gd = a.groups[g];
      ps.nb = gd.nb;
      ps.shift = a.recentre != 0 ? gd.shift : nullptr;
      ps.E = gd.E;
      ps.b = gd.b;
      const int s0 = ((int)Pc - __builtin_amdgcn_readfirstlane(pp[g])) << 7;
      ps.base = (long long)g * a.n_pad + s0;
      const int left = a.totals[g] - s0;
      ps.n_valid = P < n_super ? (left < 128 ? left : 128) : 0;
    } else {
      ps.nb = blob + nb_hdr(blob, NB_H_OFF_NEURAL) +
              a.m * nb_hdr(blob, NB_H_NEURAL_STRIDE);
      ps.shift = (a.recentre != 0 && nb_hdr(blob, NB_H_OFF_SHIFT) != 0)
                     ? blob + nb_hdr(blob, NB_H_OFF_SHIFT) : nullptr;
      ps.E = (int)nb_hdr(blob, NB_H_E);
      ps.b = 0;
      ps.base = 0;
      ps.n_valid = 0;
    }
    return ps;
  };
  Pass cur_p = pass_of(sup);

  // ---- DMA: a linear run of 1 KB pieces, or the packed ellipsoid block ---
  const double* dma_src = nullptr;
  double* dma_dst = nullptr;
  int dma_c = 0, dma_n = 0;
  bool dma_ell = false;
  auto dma_begin = [&](const double* src, double* dst, int n_doubles)
      __attribute__((always_inline)) {
    dma_src = src; dma_dst = dst; dma_c = wave; dma_n = n_doubles >> 7;
    dma_ell = false;
  };
  auto dma_begin_ell = [&](const double* nb_blk, double* dst)
      __attribute__((always_inline)) {
    dma_src = nb_blk; dma_dst = dst; dma_c = wave;
    dma_n = ELL_CHUNKS;
    dma_ell = true;
  };
  auto dma_one = [&]() __attribute__((always_inline)) {
    int s_off, d_off;
    if (!dma_ell) {
      s_off = d_off = dma_c * 128;
    } else if (dma_c < 1) {                         // centre
      s_off = 2 + 2 * DP;
      d_off = 0;
    } else if (dma_c < 1 + 2 * NT) {                // tile p = (ht, kt <= ht)
      const int p = (dma_c - 1) >> 1, half = (dma_c - 1) & 1;
      int ht = 0;
      while ((ht + 1) * (ht + 2) / 2 <= p) ++ht;
      const int kt = p - ht * (ht + 1) / 2;
      s_off = 2 + 3 * DP + (kt * DT + ht) * NB_TILE + half * 128;
      d_off = HEAD + p * NB_TILE + half * 128;
    } else {                                        // threshold, mean, 1/scale
      const int i = dma_c - 1 - 2 * NT;
      s_off = nb_ell_block_size(DT) + i * 128;
      d_off = TAIL + i * 128;
    }
    __builtin_amdgcn_global_load_lds(
        (nbf_gptr)(dma_src + s_off + 2 * lane),
        (nbf_lptr)(dma_dst + d_off), 16, 0, 0);
    dma_c += NW;
  };
  auto dma_tick = [&]() __attribute__((always_inline)) {
    if (dma_c < dma_n) dma_one();
  };
  auto dma_flush = [&]() __attribute__((always_inline)) {
    while (dma_c < dma_n) dma_one();
  };

  int q = 0;                         // reg(q): ellipsoid block of this pass
  dma_begin_ell(cur_p.nb, reg(0));
  dma_flush();
  long long pt[T];
  bool valid[T];
  double2 xraw[T][2 * DT];
  // row of x behind slot p: p itself, or idx[p] (gathered scores); slots past
  // the end read row 0 and are masked
  auto row_of = [&](long long p) __attribute__((always_inline)) {
    return p < a.n ? (a.idx != nullptr ? a.idx[p] : p) : 0ll;
  };
  // BATCH: row behind slot (wave, t, lane) of a pass
  auto row_in = [&](const Pass& ps, int t) __attribute__((always_inline)) {
    const int slot = (wave * T + t) * 16 + (lane & 15);
    return (long long)a.dense[ps.base + (slot < ps.n_valid ? slot : 0)];
  };
  long long nrow[T];             // rows of the pass after the current one
#pragma unroll
  for (int t = 0; t < T; ++t) {
    pt[t] = ((sup * NW + wave) * T + t) * 16 + (lane & 15);
    valid[t] = pt[t] < a.n;
    if constexpr (BATCH) nrow[t] = row_in(cur_p, t);
    else nrow[t] = row_of(pt[t]);
  }
  load_points_raw<DT, T>(a.x, nrow, n_dim, lane, xraw);

  for (; sup < n_super; sup += gridDim.x) {
    const double* nb_m = cur_p.nb;
    const double* shift = cur_p.shift;
    const int E = cur_p.E;
    const double* nets = nb_m + nb_ell_block_size(DT) + 2 + 2 * DP;
    // (BATCH: the group of the next pass, a pass ahead of its block and rows)
    const Pass next_p = pass_of(sup + gridDim.x);
    // lane group, opaque per pass: everything indexed by 4 ks + lg below (the
    // centre, mean and scale slots, the cube and padding predicates) is a loop
    // invariant of the pass loop, and hoisted out of it those are ~6 DT
    // registers and as many saved predicates held across the network stages
    int lgp = lg;
    asm volatile("" : "+v"(lgp));
    long long crow[T];             // BATCH: the rows this pass evaluates
    // layer 1 (chunk a) of the first network -> the other region, under the
    // prologue; the ellipsoid block and the points are waited for here
    dma_begin(nets, reg(q ^ 1), NA_D);
    dma_flush();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NA_D / 128 + NW - 1) / NW)
                 : "memory");
    __syncthreads();
    const double* blk = reg(q);
    const double thr = blk[TAIL];

    bool in_cube[T], acc_outer[T], want[T];
    double xin[T][4 * DT];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      pt[t] = ((sup * NW + wave) * T + t) * 16 + (lane & 15);
      in_cube[t] = true;
      if constexpr (BATCH) {
        valid[t] = (wave * T + t) * 16 + (lane & 15) < cur_p.n_valid;
        crow[t] = nrow[t];
        nrow[t] = row_in(next_p, t);
      } else {
        valid[t] = pt[t] < a.n;
        // (the index of the next pass's row, a pass ahead of its loads)
        nrow[t] = row_of((((sup + gridDim.x) * NW + wave) * T + t) * 16 +
                         (lane & 15));
      }
    }
    points_from_raw<DT, T>(xraw, valid, n_dim, lgp, xin);
    if (shift != nullptr) {
      // periodic dimensions are recentred before the test (nautilus.py:
      // 162-163, periodic.py:69-71): x <- (x + (0.5 - centre)) mod 1
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        const double sv = shift[4 * ks + lgp];
        const bool on = shift[DP + 4 * ks + lgp] != 0.0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const double v = xin[t][ks] + sv;
          xin[t][ks] = on ? v - floor(v) : xin[t][ks];
        }
      }
    }
    if (m_sample) {
      bool cbad[T];
#pragma unroll
      for (int t = 0; t < T; ++t) cbad[t] = false;
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        const int f = 8 * (ks >> 1) + 2 * lgp + (ks & 1);
        const bool boxed = use_cube && f < n_dim;
#pragma unroll
        for (int t = 0; t < T; ++t)
          cbad[t] |= boxed && !(xin[t][ks] >= 0.0 && xin[t][ks] < 1.0);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        in_cube[t] = !point_any(cbad[t], lane);
        // (at most one outer member here: the acceptance draw u > 1 - 1 / k
        // of union.py:318-319 is u > 0 -- true but for a 2^-53 event -- and
        // is not made: ten Philox rounds per pass)
        acc_outer[t] = in_cube[t];
        want[t] = valid[t] && acc_outer[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) { acc_outer[t] = false; want[t] = valid[t]; }
    }

    double y[T][4 * DT], r2[T];
    bool box_bad[T], inside_e[T], need[T];
    ell_eval_centre<DT, T>(blk, blk + HEAD, n_dim, xin, lane, lgp, y, r2);
#pragma unroll
    for (int t = 0; t < T; ++t) box_bad[t] = false;
    bool wave_mlp = false;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      inside_e[t] = !box_bad[t] && r2[t] < 1.0;
      need[t] = m_sample ? (want[t] && inside_e[t]) : valid[t];
      cnt_ell += __popcll(__ballot(want[t] && lg == 0));
      cnt_mlp += (unsigned long long)E * __popcll(__ballot(need[t] && lg == 0));
      wave_mlp |= need[t];
    }
    wave_mlp = __any(wave_mlp);

    double tin[T][KS1];
    {
      const double* r_mean = blk + TAIL + 2;
      const double* r_isc = r_mean + DP;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        const int f = 4 * ks + lgp;
        if (ks < 4 * DT) {
          // (slots up to 16 DT exist in the block; what the padding holds is
          // discarded by the select)
          const double mv = r_mean[f], sv = r_isc[f];
          const double pad = (f == n_dim) ? 1.0 : 0.0;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double v = (y[t][ks] - mv) * sv;
            tin[t][ks] = (f < n_dim) ? v : pad;
          }
        } else {
#pragma unroll
          for (int t = 0; t < T; ++t) tin[t][ks] = (f == n_dim) ? 1.0 : 0.0;
        }
      }
    }
    // the block has been read; layer 1 (chunk a) has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // The row indices of the next pass (index lists / candidate lists) were
    // loaded above and are consumed here, where nothing is in flight: left to
    // their first real use -- the point loads in the last stage of the pass --
    // the compiler's wait for them also waits for every weight DMA issued in
    // between (it cannot count those): 17 % of a gathered score launch.
#pragma unroll
    for (int t = 0; t < T; ++t) asm volatile("" : "+v"(nrow[t]));

    double total[T];
#pragma unroll
    for (int t = 0; t < T; ++t) total[t] = 0.0;
    int cur = q ^ 1;                 // region of the stage about to run

    auto stage_end = [&]() __attribute__((always_inline)) {
      dma_flush();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    };
    auto network = [&](int e, auto last_c) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(last_c)::value;
      const double* w_e = nets + e * net_stride;
      double h1[T][4 * NB_HT1];
      if constexpr (TWO) {
        // -- L1 chunk a; chunk b -> other region
        dma_begin(w_e + NA_D, reg(cur ^ 1), NB1_D);
        if (wave_mlp)
          fl_chunk<T, KS1, NB_HT1, 0, 4 * KA, 2>(reg(cur), ks1, tin, lane, h1,
                                                dma_tick);
        stage_end();
        // -- L1 chunk b; layers 2-4 -> other region
        dma_begin(w_e + NA_D + NB1_D, reg(cur ^ 1), NC_D);
        if (wave_mlp) {
          fl_chunk<T, KS1, NB_HT1, 4 * KA, KS1, 1>(reg(cur), ks1, tin, lane,
                                                  h1, dma_tick);
          fl_pad<T, NB_HT1, 25, 0>(h1, lane);
        }
        stage_end();
      } else {
        // -- layer 1; layers 2-4 -> other region
        dma_begin(w_e + NA_D, reg(cur ^ 1), NC_D);
        if (wave_mlp) {
          fl_chunk<T, KS1, NB_HT1, 0, KS1, 1>(reg(cur), ks1, tin, lane, h1,
                                             dma_tick);
          fl_pad<T, NB_HT1, 25, 0>(h1, lane);
        }
        stage_end();
      }
      // -- layers 2-4; next network's chunk a, or the ellipsoid block of the
      // next pass, -> other region
      if constexpr (LAST) {
        load_points_raw<DT, T>(a.x, nrow, n_dim, lane, xraw);
        dma_begin_ell(next_p.nb, reg(cur ^ 1));
      } else {
        dma_begin(nets + (e + 1) * net_stride, reg(cur ^ 1), NA_D);
      }
      if (wave_mlp) {
        const double* w2 = reg(cur);
        const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
        const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
        double h2[T][4 * NB_HT2], h3[T][4 * NB_HT3], o[T][4];
        double a2[FlFirst<8, NB_HT2>::NA], a3[FlFirst<8, NB_HT3>::NA],
            a4[FlFirst<8, 1>::NA];
        fl_read_first<8, NB_HT2>(w2, lane, a2);
        fl_layer_from<T, 8, 26, 0, NB_HT2, true, 2, 1, 0>(
            w2, 26, h1, lane, h2, a2,
            [&]() __attribute__((always_inline)) {
              fl_read_first<8, NB_HT3>(w3, lane, a3);
            },
            dma_tick);
        fl_pad<T, NB_HT2, 12, 2>(h2, lane);
        fl_layer_from<T, 8, 13, 0, NB_HT3, true, 2, 2, 0>(
            w3, 13, h2, lane, h3, a3,
            [&]() __attribute__((always_inline)) {
              fl_read_first<8, 1>(w4, lane, a4);
            },
            dma_tick);
        fl_pad<T, NB_HT3, 5, 0>(h3, lane);
        fl_layer_from<T, 8, 6, 0, 1, true, 2, 5, 0>(
            w4, 6, h3, lane, o, a4, []() __attribute__((always_inline)) {},
            dma_tick);
#pragma unroll
        for (int t = 0; t < T; ++t) total[t] += o[t][0];
      }
      stage_end();
    };
    for (int e = 0; e + 1 < E; ++e) network(e, std::false_type{});
    network(E - 1, std::true_type{});
    q = cur;                         // where the next pass's block landed

#pragma unroll
    for (int t = 0; t < T; ++t) {
      const double score = __shfl(total[t], lane & 15) / (double)E;
      if constexpr (BATCH) {
        // the geometric stage found the point inside this neural bound's
        // ellipsoid; the emulator decides (bounds/neural.py:121-126).  Several
        // groups may say yes to the same row: they store the same byte.
        if (valid[t] && lg == 0 && score > thr) {
          a.st[crow[t]] = a.out_mode == 2 ? (unsigned char)3 : (unsigned char)2;
          if (a.out_mode == 1) atomicMin(&a.first[crow[t]], cur_p.b);
        }
      } else if (m_sample) {
        bool ok = inside_e[t];
        if (need[t]) ok = inside_e[t] && (score > thr);
        const unsigned char flags =
            (acc_outer[t] ? 1 : 0) | ((acc_outer[t] && ok) ? 2 : 0);
        if (valid[t] && lg == 0) a.out_u8[pt[t]] = flags;
      } else if (valid[t] && lg == 0) {
        a.out_f64[2 * pt[t]] = r2[t];
        a.out_f64[2 * pt[t] + 1] = score;
      }
    }
    cur_p = next_p;
  }
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[1], cnt_ell);
    atomicAdd(&a.counters[2], cnt_mlp);
  }
}

template <int DT, int KT1, bool BATCH>
int launch_fast_impl(const FastArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)2 * FAST_REGION * sizeof(double) +
                     (BATCH ? (FAST_MAX_GROUPS + 1) * sizeof(int) : 0);
  static bool configured = false;
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_eval_fast_kernel<DT, KT1, BATCH>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    configured = true;
  }
  // BATCH: the number of passes is only known on the device; workgroups
  // without a pass leave at once
  const long long n_super = (a.n + 127) / 128;
  long long blocks = n_super < 256 ? n_super : 256;
  hipLaunchKernelGGL((nb_eval_fast_kernel<DT, KT1, BATCH>),
                     dim3((unsigned)blocks), dim3(512), lds, stream, a);
  return NB_OK;
}

template <int DT, int KT1>
int launch_fast(const FastArgs& a, hipStream_t stream) {
  return a.groups != nullptr ? launch_fast_impl<DT, KT1, true>(a, stream)
                             : launch_fast_impl<DT, KT1, false>(a, stream);
}

}  // namespace

unsigned long long* nb_eval_counters();

// one neural bound with networks, and for proposals at most one outer member
// (the draw then needs no overlap count)
bool nb_eval_fast_eligible(int n_dim, int K, int M, int E, bool sample) {
  if (n_dim > 128 || M < 1 || E < 1) return false;
  // proposals: the fused cube test / acceptance draw of this kernel covers
  // one neural bound and at most one outer member; scores: any neural bound
  return !sample || (K <= 1 && M == 1);
}

static int dispatch_fast(const FastArgs& a, int n_dim, hipStream_t stream) {
  const int dt = (n_dim + 15) / 16, kt1 = (n_dim + 1 + 15) / 16;
  int rc = NB_ERR_UNSUPPORTED;
  switch (4 * dt + (kt1 - dt)) {
    case 4: rc = launch_fast<1, 1>(a, stream); break;
    case 5: rc = launch_fast<1, 2>(a, stream); break;
    case 8: rc = launch_fast<2, 2>(a, stream); break;
    case 9: rc = launch_fast<2, 3>(a, stream); break;
    case 12: rc = launch_fast<3, 3>(a, stream); break;
    case 13: rc = launch_fast<3, 4>(a, stream); break;
    case 16: rc = launch_fast<4, 4>(a, stream); break;
    case 17: rc = launch_fast<4, 5>(a, stream); break;
    case 20: rc = launch_fast<5, 5>(a, stream); break;
    case 21: rc = launch_fast<5, 6>(a, stream); break;
    case 24: rc = launch_fast<6, 6>(a, stream); break;
    case 25: rc = launch_fast<6, 7>(a, stream); break;
    case 28: rc = launch_fast<7, 7>(a, stream); break;
    case 29: rc = launch_fast<7, 8>(a, stream); break;
    case 32: rc = launch_fast<8, 8>(a, stream); break;
    case 33: rc = launch_fast<8, 9>(a, stream); break;
    default:
      nb_set_error("nb_launch_eval_fast: n_dim=%d not eligible", n_dim);
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_eval_fast(const double* blob_dev, int n_dim, bool sample, int m,
                        int recentre, const double* x, const long long* idx,
                        long long n,
                        unsigned char* out_u8, double* out_f64,
                        unsigned long long seed, unsigned long long offset,
                        hipStream_t stream) {
  if (n <= 0) return NB_OK;
  FastArgs a = {};
  a.blob = blob_dev; a.sample = sample ? 1 : 0; a.x = (const nb_gd*)x; a.n = n;
  a.idx = idx; a.m = m; a.recentre = recentre;
  a.out_u8 = out_u8; a.out_f64 = out_f64; a.seed = seed; a.offset = offset;
  a.counters = nb_eval_counters();
  return dispatch_fast(a, n_dim, stream);
}

// Second stage of the two-stage bound evaluation (nb_cand.hip): the emulators
// of every (bound, neural bound) group on the candidate rows the geometric
// stage left for it -- ONE launch for all groups, the row counts read from
// device memory (totals_dev), the grid sized for n_upper candidates.
// `groups_dev`: FastGroup records (struct layout shared with nb_api.hip).
int nb_launch_eval_fast_batch(const double* blob0_dev, int n_dim, int recentre,
                              const double* x, const void* groups_dev,
                              int n_groups, const int* totals_dev,
                              const int* dense_dev, long long n_pad,
                              long long n_upper, int out_mode,
                              unsigned char* st, int* first,
                              hipStream_t stream) {
  if (n_upper <= 0 || n_groups <= 0) return NB_OK;
  if (n_groups > FAST_MAX_GROUPS) {
    nb_set_error("nb_launch_eval_fast_batch: %d groups (limit %d)", n_groups,
                 FAST_MAX_GROUPS);
    return NB_ERR_ARG;
  }
  FastArgs a = {};
  a.blob = blob0_dev; a.sample = 0; a.x = (const nb_gd*)x;
  // (grid: every group may end in a partial pass)
  a.n = n_upper + 128ll * n_groups;
  a.recentre = recentre;
  a.counters = nb_eval_counters();
  a.groups = (const FastGroup*)groups_dev; a.n_groups = n_groups;
  a.totals = totals_dev; a.dense = dense_dev; a.n_pad = n_pad;
  a.out_mode = out_mode; a.st = st; a.first = first;
  return dispatch_fast(a, n_dim, stream);
}
