// Helpers shared by the construction kernels (nb_mvee.hip, nb_gmm.hip):
// symmetric matrices as lower-triangular sets of 16x16 MFMA operand tiles,
// weighted second moments on the matrix cores, quadratic forms q^T P q.
#pragma once
#include "nb_tile.h"

namespace {

constexpr int SY_WAVES = 8;          // wavefronts of a construction workgroup

__host__ __device__ inline int mv_tri(int ht, int kt) {
  return ht * (ht + 1) / 2 + kt;
}
// position of feature offset o (0..15) of a k-tile in the operand tile:
// slot s = 2 (o >> 3) + (o & 1), lane group (o >> 1) & 3   (nb_tile.h, perm)
__device__ __forceinline__ int mv_kpos(int o) {
  return (2 * (o >> 3) + (o & 1)) * 64 + ((o >> 1) & 3) * 16;
}
// row position of feature offset o of an h-tile: the accumulator register r
// of lane group lg then holds feature 8 (r >> 1) + 2 lg + (r & 1), i.e. the
// feature the lane holds in input slot 4 ht + r
__device__ __forceinline__ int mv_hpos(int o) {
  return ((o >> 1) & 3) + 4 * (2 * (o >> 3) + (o & 1));
}
__device__ __forceinline__ int mv_slot(int f) {       // slot_of_feature
  const int j = f >> 3, r = f & 7;
  return 4 * (2 * j + (r & 1)) + (r >> 1);
}
// LDS position of entry (r, c), r >= c, of a symmetric matrix held as
// lower-triangular operand tiles
__device__ __forceinline__ int sy_pos(int r, int c) {
  return mv_tri(r >> 4, c >> 4) * NB_TILE + mv_kpos(c & 15) + mv_hpos(r & 15);
}

// g = q^T P q for the 16 points of a tile: `T` holds the lower triangle of the
// symmetric P as operand tiles with the off-diagonal entries doubled, so that
// y = T q needs the lower-triangular tiles only and g = q . y.  xin: the
// B-operand block of the points (nb_tile.h load_points), m: rows of P.
template <int DT>
__device__ __forceinline__ double sy_quadform(const double* T,
                                              const double (&xin)[4 * DT],
                                              int m, int lane) {
  double part = 0.0;
#pragma unroll
  for (int ht = 0; ht < DT; ++ht) {
    if (16 * ht < m) {
      nb_d4 acc = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4 * (ht + 1); ++ks) {
        const double a = T[mv_tri(ht, ks >> 2) * NB_TILE + (ks & 3) * 64 + lane];
        acc = MFMA(a, xin[ks], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part += acc[r] * xin[4 * ht + r];
    }
  }
  return lane_group_sum(part);
}

// Partial sums of S = sum_p w_p q_p q_p^T, q = (x, 1), over the points
// p0 + 4 (sg + nsg j) + {0..3} < p1: lower block triangle of 16x16 tiles in
// the accumulator layout, written to out[NT][256].  Wave `w` of a sub-group
// of (DT + 1) / 2 wavefronts owns the tile rows w and DT-1-w (balanced
// triangle).
template <int DT, int PF_ = 0, typename WPtr>
__device__ __forceinline__ void sy_moments(const double* __restrict__ x,
                                           WPtr wgt, int d, int p0, int p1,
                                           int sg, int nsg, int w, int lane,
                                           double* out) {
  const int kp = lane >> 4, fi = lane & 15;
  const int row_lo = w, row_hi = DT - 1 - w;
  const bool two = row_hi != row_lo;
  nb_d4 acc_lo[DT], acc_hi[DT];
#pragma unroll
  for (int j = 0; j < DT; ++j) {
    acc_lo[j] = nb_d4{0.0, 0.0, 0.0, 0.0};
    acc_hi[j] = nb_d4{0.0, 0.0, 0.0, 0.0};
  }
  // The rows of PF steps are in flight ahead of the MFMAs of a step (one
  // step = 4 points = DT + 1 MFMAs per wavefront, ~0.15 us; a load from L2 /
  // HBM takes 1-2 us: without the ring every step paid that latency -- 160 us
  // per call for 2000 points at n_dim 50, against ~20 us of matrix-core
  // work).  Loads are unconditional (clamped addresses) and masked by
  // multiplication, so that nothing branches around them; steps past the end
  // contribute exact zeros, the order of the sums is that of the plain loop.
  constexpr int PF =
      PF_ > 0 ? PF_ : (DT <= 4 ? 8 : (DT <= 6 ? 5 : (DT <= 7 ? 3 : 2)));
  const int step = 4 * nsg;
  // per-lane masks of the feature columns (1 for a feature, the constant 1 in
  // column d)
  double m_feat[DT], m_one[DT];
  int f_at[DT];
#pragma unroll
  for (int ft = 0; ft < DT; ++ft) {
    const int f = 16 * ft + fi;
    m_feat[ft] = f < d ? 1.0 : 0.0;
    m_one[ft] = f == d ? 1.0 : 0.0;
    f_at[ft] = f < d ? f : d - 1;
  }
  double rb[PF][DT], rw[PF];
  auto fetch = [&](int s, double (&b)[DT], double& wp)
      __attribute__((always_inline)) {
    const int p = s + kp;
    const int pc = p < p1 ? p : (p1 > 0 ? p1 - 1 : 0);
    const double on = p < p1 ? 1.0 : 0.0;
    const double* row = x + (size_t)pc * d;
#pragma unroll
    for (int ft = 0; ft < DT; ++ft) b[ft] = row[f_at[ft]];
    wp = wgt != nullptr ? wgt[pc] : 1.0;
    // (masks applied where the values are used)
    wp *= on;
  };
  const int s0 = p0 + 4 * sg;
#pragma unroll
  for (int j = 0; j < PF; ++j) fetch(s0 + j * step, rb[j], rw[j]);
  for (int s = s0; s < p1; s += PF * step) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      double b[DT];
      const double on = (s + j * step + kp) < p1 ? 1.0 : 0.0;
#pragma unroll
      for (int ft = 0; ft < DT; ++ft)
        b[ft] = (rb[j][ft] * m_feat[ft] + m_one[ft]) * on;
      const double wp = rw[j];
      // rows of the A operand: the same columns, weighted
      double a_lo = 0.0, a_hi = 0.0;
#pragma unroll
      for (int ft = 0; ft < DT; ++ft) {
        if (ft == row_lo) a_lo = b[ft] * wp;
        if (ft == row_hi) a_hi = b[ft] * wp;
      }
      fetch(s + (j + PF) * step, rb[j], rw[j]);
#pragma unroll
      for (int jt = 0; jt < DT; ++jt) {
        if (jt <= row_lo) acc_lo[jt] = MFMA(a_lo, b[jt], acc_lo[jt]);
        if (two && jt <= row_hi) acc_hi[jt] = MFMA(a_hi, b[jt], acc_hi[jt]);
      }
    }
  }
#pragma unroll
  for (int jt = 0; jt < DT; ++jt) {
    if (jt <= row_lo) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[mv_tri(row_lo, jt) * NB_TILE + r * 64 + lane] = acc_lo[jt][r];
    }
    if (two && jt <= row_hi) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[mv_tri(row_hi, jt) * NB_TILE + r * 64 + lane] = acc_hi[jt][r];
    }
  }
}

// entry (r, c), r >= c, of the moment matrix: the partial results summed in
// fixed order
template <typename Ptr>
__device__ __forceinline__ double mom_element(Ptr partial, int vw, int nt,
                                              int r, int c) {
  const int it = r >> 4, jt = c >> 4, i = r & 15, j = c & 15;
  const int off = mv_tri(it, jt) * NB_TILE + (i >> 2) * 64 + (i & 3) * 16 + j;
  double s = 0.0;
  for (int v = 0; v < vw; ++v) s += partial[(size_t)v * nt * NB_TILE + off];
  return s;
}

}  // namespace
