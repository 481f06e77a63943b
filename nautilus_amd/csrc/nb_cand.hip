// First stage of the two-stage bound evaluation, entirely on the device:
// everything contains() / sample() decide WITHOUT an emulator -- periodic
// recentring, unit-cube clip, the ellipsoids of the outer union's members
// (overlap count, union.py:285-289, 316-319), the acceptance draw, the
// ellipsoids of the neural bounds (bounds/neural.py:115-120) -- for single
// bounds and for lists of bounds (shell exclusion, sampler.py:797-798; shell
// association, 1213-1219), and the CANDIDATE LISTS for the second stage.
//
// The reference evaluates `outer.contains(x) & any(nb.contains(x) for nb in
// neural_bounds)` (nautilus.py:162-169, 212-216): a disjunction over the
// neural bounds, and for a list of bounds a disjunction (exclusion) or the
// first hit (association) over the bounds.  So a point needs the emulator of
// EVERY (bound b, neural bound m) whose ellipsoid contains it, and nothing
// has to wait for anything: this kernel walks the whole list once per point
// and appends the row to the candidate list of each such group (b, m); the
// second stage (nb_eval_fast.hip, BATCH) scores all lists in ONE launch on
// dense 128-point passes and ORs the verdicts into the status bytes.  No
// rounds, no host round trip, no index sorting: three launches per query
// (candidates, list compaction, scores) with all counts read on the device.
//
// Candidate lists without atomics: wavefront w of the grid owns the points
// [w * chunk, (w + 1) * chunk) and, in every group's list, the segment of the
// same range; its fill counts live in LDS (one int per group and wavefront)
// and go to counts[g][w] at the end.  nb_cand_compact_kernel turns the
// segments into dense lists (prefix sums over the wavefronts, binary search
// per destination slot) -- rows stay in ascending order, so the lists, the
// passes built from them and every result are reproducible run to run.
//
// The geometric tests run on the matrix cores like every ellipsoid test of
// this library (nb_tile.h layout), but per WAVEFRONT: T tiles of 16 points
// share each A operand, which is read straight from the bound's blob (L2 /
// L1; the blocks of a list are far too many for LDS, and a workgroup-wide
// staging step would tie eight wavefronts to the slowest point), `PD` k-steps
// ahead of the MFMAs.  Bounds are skipped per wavefront (bounding-sphere
// pre-test, cube clip, nothing left to decide).
#include <limits.h>

#include "nb_common.h"

#include "nb_tile.h"

namespace {

constexpr int CD_WPB = 4;              // wavefronts per workgroup
constexpr int CD_MAX_WAVES = 4096;     // segments per group (compaction: LDS)

enum { CM_ANY = 0, CM_FIRST = 1, CM_SAMPLE = 2 };
// status byte of a row (SAMPLE: the flags of nb_accept)
enum : unsigned char { CS_OUTER = 1, CS_INSIDE = 2 };

struct CandArgs {
  const double* const* blobs;   // device array of nb blob pointers
  const int* group_base;        // device: first group of bound b
  int nb, n_groups, mode;
  int b_off, g_off;             // a slice of a longer list: its first bound /
                                // group (positions and groups are the list's)
  int accumulate;               // ... after earlier slices: rows already
                                // inside stay as they are
  const nb_gd* x;               // (n, n_dim)
  long long n;
  unsigned char* st;            // per row: status
  int* first;                   // CM_FIRST: bound index or INT_MAX
  int* seg;                     // [n_groups][n_pad] candidate rows
  int* counts;                  // [n_groups][n_waves]
  long long n_pad;
  int chunk, n_waves;
  unsigned long long seed, offset;
  unsigned long long* counters; // optional, as in nb_eval.hip
  // long lists over few rows: blockIdx.y owns the bounds [y * b_chunk,
  // (y + 1) * b_chunk) of the slice (0: one block row walks them all).  The
  // blocks of a row range then share its status bytes / first-bound words:
  // the launcher has initialised them, the kernel only ever ORs / MINs
  int b_chunk;
};

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// the k-steps of a lower-triangular ellipsoid transform in execution order.
// SMALL: the last row tile holds at most four real rows (n_dim mod 16 in
// 1..4: n_dim 50 has two, n_dim 100 four) -- its last two k-steps then hold
// zero padding only (the K permutation pairs k-steps: 2 j, 2 j + 1 cover the
// features 8 j .. 8 j + 7) and are left out, and the tile itself runs on
// v_mfma_f64_4x4x4_4b (cand_inside).
template <int DT, bool SMALL>
struct StepTable {
  static constexpr int N = 2 * DT * (DT + 1) - (SMALL ? 2 : 0);
  unsigned char ht[N], ks[N];
  constexpr StepTable() : ht(), ks() {
    int i = 0;
    for (int h = 0; h < DT; ++h)
      for (int k = 0; k < 4 * (h + 1) - ((SMALL && h == DT - 1) ? 2 : 0);
           ++k) {
        ht[i] = (unsigned char)h;
        ks[i] = (unsigned char)k;
        ++i;
      }
  }
};

// inside[t] = point of tile t passes the block's box limits and lies inside
// its ellipsoid.  X(t, ks) = coordinate slot ks of tile t (lane layout of
// nb_tile.h).  Operands (A tile rows, centre) come from global memory PD
// k-steps ahead; same summation order as ell_eval / ell_eval_centre, so r2
// is bit-identical to the other kernels'.  SMALL (see StepTable): the last
// row tile as 4 rows x 16 points on v_mfma_f64_4x4x4_4b -- 16 instead of 64
// cycles per k-step; the A operand (lane i + 4 b + 16 k: row i, replicated
// over the four blocks b) is gathered from the same tile storage, the B
// operand is the 16x16x4 one, and the result (lane p + 16 i) is bit for bit
// register 0 of the 16x16x4 accumulator: same k order per row, and the
// squares land in the lane groups where the full tile had them (rows 4..15 of
// that tile are zero padding and added +0.0).
//
// The centre goes through a wavefront-private strip of LDS (`cw`, DP doubles):
// read from the blob once per block (one or two loads per lane) and from
// there per k-step.  Straight from the blob it was a second vector-memory
// instruction per k-step next to the A operand -- and the CU's texture path
// takes ~16-20 cycles per 64-lane load whatever its addresses are: with eight
// wavefronts per CU that path, not the matrix pipe, set the pace (n_dim 50:
// 4 x 76 loads per 3.6 k cycles of matrix work and SIMD).
// (STRIP = false: two tiles at n_dim > 112 have no register left for it and
// read the centre from the blob as before.)
template <int DT, int T, int PD, bool SMALL, bool STRIP, class XF>
__device__ __forceinline__ void cand_inside(const nb_gd* blk, bool has_ell,
                                            bool has_box, XF&& X, int lane,
                                            int lg, double* cw,
                                            bool (&inside)[T]) {
  constexpr int DP = 16 * DT;
  constexpr StepTable<DT, SMALL> TAB{};
  constexpr int N = StepTable<DT, SMALL>::N;
  const int lane4 = (lane >> 4) * 16 + (lane & 3);
  const nb_gd* lo = blk + 2;
  const nb_gd* hi = lo + DP;
  const nb_gd* c = hi + DP;
  const nb_gd* tiles = c + DP;
  bool bad[T];
#pragma unroll
  for (int t = 0; t < T; ++t) bad[t] = false;
  if (has_box) {
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks) {
      const double lov = lo[4 * ks + lg], hiv = hi[4 * ks + lg];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double xv = X(t, ks);
        bad[t] |= !(xv >= lov && xv < hiv);
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) bad[t] = point_any(bad[t], lane);
  }
  double part[T];
#pragma unroll
  for (int t = 0; t < T; ++t) part[t] = 0.0;
  if (has_ell) {
    double a[PD], cv[PD];
    // (the strip is the wavefront's own: its LDS operations execute in
    // order, no barrier)
    if (STRIP) {
#pragma unroll
      for (int j = 0; j < DP; j += 64)
        if (j + 64 <= DP || lane < DP - j) cw[j + lane] = c[j + lane];
    }
    auto fetch = [&](int i, int slot) __attribute__((always_inline)) {
      const int ht = TAB.ht[i], ks = TAB.ks[i];
      a[slot] = tiles[((ks >> 2) * DT + ht) * NB_TILE + (ks & 3) * 64 +
                      ((SMALL && ht == DT - 1) ? lane4 : lane)];
      cv[slot] = STRIP ? cw[4 * ks + lg] : c[4 * ks + lg];
    };
#pragma unroll
    for (int i = 0; i < PD && i < N; ++i) fetch(i, i);
    nb_d4 acc[T];
    double r4[T];
    // (the order is pinned: left alone, the scheduler hoists the loads of all
    // k-steps to the top and spills)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int ht = TAB.ht[i], ks = TAB.ks[i];
      const bool small = SMALL && ht == DT - 1;
      if (ks == 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          acc[t] = nb_d4{0.0, 0.0, 0.0, 0.0};
          r4[t] = 0.0;
        }
      }
      const double av = a[i % PD];
      double cc = cv[i % PD];
      // the wait lands here; and the centre stays opaque: recognised as the
      // value of an earlier k-step it would keep every x - c of the block
      // alive next to x (twice the registers)
      asm volatile("" : "+v"(cc) : "v"(av));
      __builtin_amdgcn_sched_barrier(0);
      if (i + PD < N) fetch(i + PD, i % PD);
      __builtin_amdgcn_sched_barrier(0);
      if (small) {
#pragma unroll
        for (int t = 0; t < T; ++t) r4[t] = MFMA4(av, X(t, ks) - cc, r4[t]);
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = MFMA(av, X(t, ks) - cc, acc[t]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (small && ks == 4 * (ht + 1) - 3) {
#pragma unroll
        for (int t = 0; t < T; ++t) part[t] += r4[t] * r4[t];
      } else if (!small && ks == 4 * (ht + 1) - 1) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[t] += acc[t][r] * acc[t][r];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t)
    inside[t] = !bad[t] && lane_group_sum(part[t]) < 1.0;
}

// (register budget: OCC wavefronts per SIMD.  Without a limit the scheduler
// hoists every load of the unrolled k-steps and takes all 512 registers, one
// wavefront per SIMD; two per SIMD run without spills up to n_dim = 64 and
// were faster than three with spills: 2.81 against 3.10 ms per 2^20 proposals
// at n_dim = 50, K = M = 4)
template <int DT, int T, int OCC, bool SAMPLE, bool SMALL>
__global__ void __launch_bounds__(64 * CD_WPB)
__attribute__((amdgpu_waves_per_eu(OCC, OCC))) nb_cand_kernel(CandArgs a) {
  constexpr int DP = 16 * DT;
  // k-steps the operands run ahead (two tiles at n_dim > 112 have the
  // registers for six)
  // (measured at n_dim 50 / 100, K = M = 4: 7 k-steps 0.858 / 2.23 ms, 10:
  // 0.875 / 2.19, 14: 0.900 / 2.25, 20: 0.93 / 2.85; three tiles per
  // wavefront at n_dim 50: 0.927 -- the stage is not waiting for its
  // operands; profiles/r05/cand_prefetch_depth_and_three_tiles.txt)
  constexpr int PD = OCC >= 4 ? 6 : (OCC == 3 && DT == 4 ? (SMALL ? 6 : 4) : ((DT == 8 && T == 2) ? 6 : 10));
  extern __shared__ int cur[];           // [n_groups][CD_WPB] fill counts
  constexpr bool STRIP = !(DT == 8 && T == 2);
  __shared__ double centre_strip[STRIP ? CD_WPB : 1][DP];   // cand_inside: `cw`
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  for (int i = threadIdx.x; i < a.n_groups * CD_WPB; i += 64 * CD_WPB)
    cur[i] = 0;
  __syncthreads();
  // (instantiated per mode: the proposals' acceptance draw and the lists'
  // sphere pre-test / periodic shift do not share registers)
  constexpr bool m_sample = SAMPLE;
  const nb_gd* const NB_G* blobs = (const nb_gd* const NB_G*)a.blobs;
  const NB_G int* group_base = (const NB_G int*)a.group_base;
  const int n_dim = (int)nb_hdr((const double*)blobs[0], NB_H_NDIM);
  const long long w = (long long)blockIdx.x * CD_WPB + wave;
  // the bounds of this block row (see CandArgs::b_chunk)
  const bool chunked = !m_sample && a.b_chunk > 0;
  const int b_lo = chunked ? (int)blockIdx.y * a.b_chunk : 0;
  const int b_hi = chunked && b_lo + a.b_chunk < a.nb ? b_lo + a.b_chunk
                                                      : a.nb;
  const long long p_begin = w * a.chunk;
  const long long p_end = p_begin + a.chunk < a.n ? p_begin + a.chunk : a.n;
  unsigned long long cnt_outer = 0, cnt_ell = 0;

  // append the rows with `hit` to the wavefront's segment of group g
  auto emit = [&](int g, bool hit, long long row) __attribute__((always_inline)) {
    const unsigned long long bal = __ballot(hit && lg == 0);
    if (bal == 0ull) return;
    int base = 0;
    if (lane == 0) {
      base = cur[g * CD_WPB + wave];
      cur[g * CD_WPB + wave] = base + __popcll(bal);
    }
    base = __shfl(base, 0);
    if (hit && lg == 0)
      a.seg[(long long)g * a.n_pad + p_begin + base +
            __popcll(bal & ((1ull << lane) - 1ull))] = (int)row;
  };

  for (long long p0 = p_begin; p0 < p_end; p0 += 16 * T) {
    long long row[T];
    bool valid[T], active[T];
    unsigned char st[T], st_in[T];
    int first[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      row[t] = p0 + 16 * t + (lane & 15);
      valid[t] = row[t] < p_end;
      active[t] = valid[t];
      st[t] = 0;
      st_in[t] = 0;
      first[t] = INT_MAX;
      if (a.accumulate && valid[t]) {
        st[t] = st_in[t] = a.st[row[t]];
        if (a.mode == CM_FIRST) first[t] = a.first[row[t]];
        active[t] = !(st[t] & CS_INSIDE);
        // Block rows over the bounds run side by side: "inside" may have
        // been set a moment ago by the block row of LATER bounds, and the
        // association wants the FIRST bound of the list -- only a bound in
        // front of this block row's (earlier slices included) settles a row.
        // (Exclusion asks for any bound: whoever was first is enough.)
        if (chunked && a.mode == CM_FIRST)
          active[t] = !(first[t] < a.b_off + b_lo);
      }
    }
    double xin[T][4 * DT];
    load_points<DT, T>(a.x, row, valid, n_dim, a.n, lane, xin);

    bool reload = false;
    for (int b = b_lo; b < b_hi; ++b) {
      // lane group, opaque per bound: what is indexed by 4 ks + lg below
      // (offsets of the centre / limit / shift slots, padding predicates of
      // the cube clip) would otherwise be hoisted out of both loops and held
      // in registers for the whole kernel
      int lgb = lg;
      asm volatile("" : "+v"(lgb));
      if (reload) {
        load_points<DT, T>(a.x, row, valid, n_dim, a.n, lane, xin);
        reload = false;
      }
      bool any_active = false;
#pragma unroll
      for (int t = 0; t < T; ++t) any_active |= active[t];
      if (!__any(any_active)) break;
      const nb_gd* blob = blobs[b];
      const double* hdr = (const double*)blob;
      const int K = (int)nb_hdr(hdr, NB_H_K);
      const int M = (int)nb_hdr(hdr, NB_H_M);
      const int E = (int)nb_hdr(hdr, NB_H_E);
      const bool use_cube = nb_hdr(hdr, NB_H_USECUBE) != 0;
      const long long ell_stride = nb_hdr(hdr, NB_H_ELL_STRIDE);
      const long long neural_stride = nb_hdr(hdr, NB_H_NEURAL_STRIDE);
      const long long off_shift = nb_hdr(hdr, NB_H_OFF_SHIFT);
      const nb_gd* nblk0 = blob + nb_hdr(hdr, NB_H_OFF_NEURAL);
      // contains() of a bound with periodic dimensions sees recentred points
      // (nautilus.py:162-163, periodic.py:69-71: x <- (x + 0.5 - centre) mod
      // 1); proposals already live in the shifted frame.  Rare: the shift is
      // applied in place and the points are read again behind the bound.
      const bool shifted = off_shift != 0 && !m_sample;
      if (shifted) {
        reload = true;
        const nb_gd* shift = blob + off_shift;
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) {
          const double sv = shift[4 * ks + lgb];
          const bool on = shift[DP + 4 * ks + lgb] != 0.0;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double u = xin[t][ks] + sv;
            xin[t][ks] = on ? u - floor(u) : xin[t][ks];
          }
        }
      }
      auto X = [&](int t, int ks) __attribute__((always_inline)) {
        return xin[t][ks];
      };

      // Bounding-sphere pre-test (bound lists): a point of a bound with
      // neural bounds lies inside one of their ellipsoids, hence within
      // sqrt(radius2) of that centre -- for nested bounds in high dimension
      // all but the next few bounds end here.
      bool maybe[T];
#pragma unroll
      for (int t = 0; t < T; ++t) maybe[t] = active[t];
      if (!m_sample && M > 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) maybe[t] = false;
        for (int m = 0; m < M; ++m) {
          const nb_gd* nb_m = nblk0 + m * neural_stride;
          const double rad2 = nb_m[1];
          const nb_gd* cc = nb_m + 2 + 2 * DP;
          double d2[T];
#pragma unroll
          for (int t = 0; t < T; ++t) d2[t] = 0.0;
#pragma unroll
          for (int ks = 0; ks < 4 * DT; ++ks) {
            const double cv = cc[4 * ks + lgb];
#pragma unroll
            for (int t = 0; t < T; ++t) {
              const double dv = X(t, ks) - cv;
              d2[t] = fma(dv, dv, d2[t]);
            }
          }
#pragma unroll
          for (int t = 0; t < T; ++t)
            maybe[t] |= active[t] && lane_group_sum(d2[t]) <= rad2;
        }
        bool any_maybe = false;
#pragma unroll
        for (int t = 0; t < T; ++t) any_maybe |= maybe[t];
        if (!__any(any_maybe)) continue;
      }

      // unit-cube clip of the union (union.py:287-288 / 313-314)
      bool in_cube[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        bool cbad = false;
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) {
          const int f = 8 * (ks >> 1) + 2 * lgb + (ks & 1);
          const double xv = X(t, ks);
          cbad |= use_cube && f < n_dim && !(xv >= 0.0 && xv < 1.0);
        }
        in_cube[t] = !point_any(cbad, lane);
      }

      // ---- outer union: overlap count ------------------------------------
      int k_cnt[T];
      bool outer_need = false;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        k_cnt[t] = 0;
        outer_need |= maybe[t] && in_cube[t];
      }
      if (m_sample && K == 1) {
#pragma unroll
        for (int t = 0; t < T; ++t) k_cnt[t] = 1;   // drawn from the only member
      } else if (K > 0 && __any(outer_need)) {
        const nb_gd* mblk = blob + nb_hdr(hdr, NB_H_OFF_MEMBERS);
        for (int m = 0; m < K; ++m) {
          const nb_gd* blk = mblk + m * ell_stride;
          const bool has_ell = ((const NB_G long long*)blk)[0] > 0;
          const bool has_box = ((const NB_G long long*)blk)[1] != 0;
          bool ins[T];
          cand_inside<DT, T, PD, SMALL, STRIP>(blk, has_ell, has_box, X, lane, lgb,
                                       centre_strip[STRIP ? wave : 0], ins);
#pragma unroll
          for (int t = 0; t < T; ++t) k_cnt[t] += ins[t] ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
          cnt_outer += (unsigned long long)K *
                       __popcll(__ballot(maybe[t] && in_cube[t] && lg == 0));
      }
      bool want[T];
      if (m_sample) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
          if (valid[t] && K == 1) {
            // one member: u > 1 - 1 / 1 = 0, true but for a 2^-53 event --
            // the draw is not made
            if (in_cube[t]) st[t] |= CS_OUTER;
          } else if (valid[t]) {
            double u0, u_acc;
            nb_uniform_pair(a.seed, a.offset + (unsigned long long)row[t], 0u,
                            NB_TAG_CTRL, u0, u_acc);
            // (no member contains it: 1 - 1 / 0 = -inf, kept -- as in the
            // reference, union.py:318-319)
            if (in_cube[t] && (u_acc > 1.0 - 1.0 / (double)k_cnt[t]))
              st[t] |= CS_OUTER;
          }
          want[t] = valid[t] && (st[t] & CS_OUTER);
        }
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t)
          want[t] = maybe[t] && in_cube[t] && (K == 0 || k_cnt[t] > 0);
      }

      // ---- neural bounds: every one whose ellipsoid contains the point ----
      bool decided[T];
#pragma unroll
      for (int t = 0; t < T; ++t) decided[t] = (M == 0) && want[t];
      const int g0 = group_base[b] - a.g_off;
      for (int m = 0; m < M; ++m) {
        bool any_want = false;
#pragma unroll
        for (int t = 0; t < T; ++t) any_want |= want[t] && !decided[t];
        if (!__any(any_want)) break;
        const nb_gd* nb_m = nblk0 + m * neural_stride;
        bool ins[T];
        cand_inside<DT, T, PD, SMALL, STRIP>(nb_m, true, false, X, lane, lgb,
                                       centre_strip[STRIP ? wave : 0], ins);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const bool test = want[t] && !decided[t];
          cnt_ell += __popcll(__ballot(test && lg == 0));
          const bool hit = test && ins[t];
          if (E == 0) decided[t] |= hit;   // no emulator: the ellipsoid decides
          else emit(g0 + m, hit, row[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (decided[t]) {
          st[t] |= CS_INSIDE;
          first[t] = a.b_off + b;
          active[t] = false;               // nothing later can change it
        }
      if (m_sample) break;                 // a single bound
    }
    if (chunked) {
      // other block rows work on the same rows: set bits / lower the first
      // bound, never overwrite (the byte's 32-bit word takes the OR)
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (valid[t] && lg == 0) {
          const unsigned char fresh = st[t] & (unsigned char)~st_in[t];
          if (fresh != 0) {
            // (the slab of rows may start at any byte of its allocation)
            const unsigned long long at =
                (unsigned long long)(a.st + row[t]);
            atomicOr((unsigned*)(at & ~3ull),
                     (unsigned)fresh << (8 * (int)(at & 3ull)));
          }
          if (a.mode == CM_FIRST && first[t] != INT_MAX)
            atomicMin(a.first + row[t], first[t]);
        }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t)
        if (valid[t] && lg == 0) {
          a.st[row[t]] = st[t];
          if (a.mode == CM_FIRST) a.first[row[t]] = first[t];
        }
    }
  }
  __syncthreads();
  // fill counts of the groups this block row owns (all of them unless the
  // bounds are dealt out over block rows)
  const int g_lo = chunked ? group_base[b_lo] - a.g_off : 0;
  const int g_hi = chunked ? group_base[b_hi] - a.g_off : a.n_groups;
  if (w < a.n_waves)
    for (int g = g_lo + lane; g < g_hi; g += 64)
      a.counts[(long long)g * a.n_waves + w] = cur[g * CD_WPB + wave];
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[0], cnt_outer);
    atomicAdd(&a.counters[1], cnt_ell);
  }
}

// segments -> dense lists.  grid = (blocks over the destination slots, groups)
__global__ void __launch_bounds__(256)
nb_cand_compact_kernel(const int* __restrict__ counts,
                       const int* __restrict__ seg, int* __restrict__ dense,
                       int* __restrict__ totals, int n_waves, int chunk,
                       long long n_pad, int per_block) {
  __shared__ int pre[CD_MAX_WAVES + 1];
  __shared__ int part[256];
  const int g = blockIdx.y, t = threadIdx.x;
  const int* cg = counts + (long long)g * n_waves;
  const int per = (n_waves + 255) / 256;
  const int w_lo = t * per;
  int sum = 0;
  for (int i = 0; i < per; ++i)
    if (w_lo + i < n_waves) sum += cg[w_lo + i];
  part[t] = sum;
  __syncthreads();
  // exclusive scan of the 256 partial sums (one wavefront)
  if (t < 64) {
    int v[4], run = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = part[4 * t + j]; run += v[j]; }
    int incl = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d);
      if (t >= d) incl += u;
    }
    int excl = incl - run;
#pragma unroll
    for (int j = 0; j < 4; ++j) { part[4 * t + j] = excl; excl += v[j]; }
  }
  __syncthreads();
  int run = part[t];
  for (int i = 0; i < per; ++i)
    if (w_lo + i < n_waves) { pre[w_lo + i] = run; run += cg[w_lo + i]; }
  if (t == 255) pre[n_waves] = run;
  __syncthreads();
  const int total = pre[n_waves];
  if (blockIdx.x == 0 && t == 0) totals[g] = total;
  const long long j0 = (long long)blockIdx.x * per_block;
  const long long j1 = j0 + per_block < total ? j0 + per_block : total;
  for (long long j = j0 + t; j < j1; j += 256) {
    int lo = 0, hi = n_waves;            // largest w with pre[w] <= j
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= (int)j) lo = mid; else hi = mid;
    }
    dense[(long long)g * n_pad + j] =
        seg[(long long)g * n_pad + (long long)lo * chunk + ((int)j - pre[lo])];
  }
}

template <int DT, int T, int OCC, bool SAMPLE, bool SMALL>
void launch_cand_m(const CandArgs& a, hipStream_t stream) {
  const size_t lds = (size_t)a.n_groups * CD_WPB * sizeof(int);
  const int blocks = (a.n_waves + CD_WPB - 1) / CD_WPB;
  const int rows_y =
      !SAMPLE && a.b_chunk > 0 ? (a.nb + a.b_chunk - 1) / a.b_chunk : 1;
  hipLaunchKernelGGL((nb_cand_kernel<DT, T, OCC, SAMPLE, SMALL>),
                     dim3((unsigned)blocks, (unsigned)rows_y),
                     dim3(64 * CD_WPB), lds, stream, a);
}

// T_S / T_L: tiles per wavefront for proposals / lists (cand_tiles); `small`:
// the last row tile has at most four real rows (StepTable)
// OCC_S / OCC_L: wavefronts per SIMD (the register budget) for proposals /
// lists -- cand_occ() below is the host's copy of the table
template <int DT, int T_S, int T_L, int OCC_S, int OCC_L>
int launch_cand_t(const CandArgs& a, bool small, hipStream_t stream) {
  if (DT > 1 && small) {
    if (a.mode == CM_SAMPLE)
      launch_cand_m<DT, T_S, OCC_S, true, (DT > 1)>(a, stream);
    else
      launch_cand_m<DT, T_L, OCC_L, false, (DT > 1)>(a, stream);
  } else {
    if (a.mode == CM_SAMPLE)
      launch_cand_m<DT, T_S, OCC_S, true, false>(a, stream);
    else
      launch_cand_m<DT, T_L, OCC_L, false, false>(a, stream);
  }
  return NB_OK;
}

}  // namespace

unsigned long long* nb_eval_counters();

// Tiles per wavefront: two share every A operand.  Proposals: two at any
// n_dim; lists (first-hit bookkeeping, sphere pre-test, periodic shift next to
// the points): two up to n_dim 80, one beyond -- the shapes that need no
// scratch (profiles/tools/kernel_resources.sh: 0 bytes for every shipped
// instantiation).  Until round 5 the kernels held ~6 DT registers of
// loop-invariant slot offsets and padding predicates across the bound loop
// and spilled with two tiles beyond n_dim 64; with the lane group opaque per
// bound they are recomputed where they are used.
static int cand_tiles(int dt, int mode) {
  if (mode == CM_SAMPLE) return 2;
  return dt <= 5 ? 2 : 1;
}

// Geometry of the candidate lists for n points: points per wavefront, number
// of wavefronts, padded list length (all multiples the kernels rely on).
// wavefronts per SIMD the kernel's registers allow (the OCC template argument
// of the launch below): proposals at 49..64 dimensions run three per SIMD (165
// registers with six k-steps of operands in flight; 2.41 -> 2.39 ms per 2^20
// proposals at n_dim 50, K = M = 4), lists keep two (three would spill)
static int cand_occ(int dt, int mode) {
  if (dt == 1) return 4;
  if (dt == 2) return 3;
  return (dt == 4 && mode == CM_SAMPLE) ? 3 : 2;
}

void nb_cand_shape(int dt, int mode, long long n, int* chunk, int* n_waves,
                   long long* n_pad) {
  const int tile = 16 * cand_tiles(dt, mode);
  // wavefronts of the grid: what the chip holds at once at the kernel's
  // register budget (two per SIMD from n_dim 33 on: 2048; three: 3072), all
  // of CD_MAX_WAVES for the small kernels
  const int occ = cand_occ(dt, mode);
  const int max_waves = occ >= 3 && dt <= 2 ? CD_MAX_WAVES : 1024 * occ;
  const long long passes = (n + tile - 1) / tile;
  const long long per_wave = (passes + max_waves - 1) / max_waves;
  *chunk = (int)((per_wave < 1 ? 1 : per_wave) * tile);
  long long w = (n + *chunk - 1) / *chunk;
  w = (w + CD_WPB - 1) / CD_WPB * CD_WPB;
  *n_waves = (int)(w < CD_WPB ? CD_WPB : w);
  *n_pad = (long long)*n_waves * *chunk;
}

// work space (bytes) of one query over n points and n_groups groups:
// [seg][dense] int32 (n_groups x n_pad each), [counts] (n_groups x n_waves),
// [totals] (n_groups)
long long nb_cand_work_bytes(int dt, int mode, long long n, int n_groups) {
  int chunk, n_waves;
  long long n_pad;
  nb_cand_shape(dt, mode, n, &chunk, &n_waves, &n_pad);
  return ((long long)n_groups * (2 * n_pad + n_waves + 1) + 64) *
         (long long)sizeof(int);
}

// candidates + compaction.  On return (in stream order) totals[g] and
// dense[g * n_pad ...] describe the second stage's work.
int nb_launch_cand(int dt, int n_dim, const double* const* blobs_dev,
                   const int* group_base_dev, int nb, int n_groups, int b_off,
                   int g_off, int accumulate, int mode, const double* x, long long n, unsigned char* st, int* first,
                   int* work, unsigned long long seed,
                   unsigned long long offset, int** dense_out,
                   int** totals_out, long long* n_pad_out,
                   hipStream_t stream) {
  if (n <= 0 || nb <= 0) return NB_OK;
  if (n > 0x7fffffffll) {
    nb_set_error("nb_launch_cand: %lld rows (limit 2^31 - 1)", n);
    return NB_ERR_ARG;
  }
  if ((size_t)n_groups * CD_WPB * sizeof(int) > 60 * 1024) {
    nb_set_error("nb_launch_cand: %d groups exceed the fill-count table",
                 n_groups);
    return NB_ERR_ARG;
  }
  CandArgs a;
  a.blobs = blobs_dev; a.group_base = group_base_dev; a.nb = nb;
  a.n_groups = n_groups; a.mode = mode; a.x = (const nb_gd*)x; a.n = n;
  a.b_off = b_off; a.g_off = g_off; a.accumulate = accumulate;
  a.st = st; a.first = first; a.seed = seed; a.offset = offset;
  a.counters = nb_eval_counters();
  nb_cand_shape(dt, mode, n, &a.chunk, &a.n_waves, &a.n_pad);
  // A long list over few rows (the shell association and exclusion of a run
  // at the reference's batch size of 100: hundreds of nested bounds, a few
  // hundred rows) is a handful of wavefronts each walking the whole list --
  // ~1.5 us of dependent loads and a full ellipsoid test per bound, 0.5 ms
  // per call, 48 % of the kernel time of a 50-dimensional funnel run
  // (profiles/r06/funnel50_kernel_stats_before.csv).  The bounds are dealt
  // out over block rows until the grid fills the chip: every block row takes
  // b_chunk bounds for all rows, results meet in the status bytes (OR) and
  // first-bound words (MIN) -- order independent, so the answer is the
  // sequential walk's.
  a.b_chunk = 0;
  if (mode != CM_SAMPLE && nb >= 8 && a.n_waves <= 512) {
    int rows_y = 2048 / a.n_waves;
    if (rows_y > nb / 2) rows_y = nb / 2;
    if (rows_y > 1) {
      a.b_chunk = (nb + rows_y - 1) / rows_y;
      if (!accumulate) {
        // (the kernel ORs / MINs into what it finds)
        NB_HIP_CHECK(hipMemsetAsync(st, 0, (size_t)n, stream));
        if (first != nullptr)
          NB_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)first, 0x7fffffff,
                                         (size_t)n, stream));
        a.accumulate = 1;
      }
    }
  }
  const long long gp = (long long)n_groups * a.n_pad;
  a.seg = work;
  int* dense = work + gp;
  a.counts = dense + gp;
  int* totals = a.counts + (long long)n_groups * a.n_waves;
  int rc = NB_OK;
  const int live = n_dim - 16 * (dt - 1);        // real rows of the last tile
  const bool small = live >= 1 && live <= 4;
  switch (dt) {
    case 1: rc = launch_cand_t<1, 2, 2, 4, 4>(a, small, stream); break;
    case 2: rc = launch_cand_t<2, 2, 2, 3, 3>(a, small, stream); break;
    case 3: rc = launch_cand_t<3, 2, 2, 2, 2>(a, small, stream); break;
    case 4: rc = launch_cand_t<4, 2, 2, 3, 2>(a, small, stream); break;
    case 5: rc = launch_cand_t<5, 2, 2, 2, 2>(a, small, stream); break;
    case 6: rc = launch_cand_t<6, 2, 1, 2, 2>(a, small, stream); break;
    case 7: rc = launch_cand_t<7, 2, 1, 2, 2>(a, small, stream); break;
    case 8: rc = launch_cand_t<8, 2, 1, 2, 2>(a, small, stream); break;
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  if (n_groups > 0) {
    const int per_block = 4096;
    const dim3 grid((unsigned)((a.n_pad + per_block - 1) / per_block),
                    (unsigned)n_groups);
    hipLaunchKernelGGL(nb_cand_compact_kernel, grid, dim3(256), 0, stream,
                       a.counts, a.seg, dense, totals, a.n_waves, a.chunk,
                       a.n_pad, per_block);
    NB_HIP_CHECK(hipGetLastError());
  }
  *dense_out = dense;
  *totals_out = totals;
  *n_pad_out = a.n_pad;
  return NB_OK;
}
