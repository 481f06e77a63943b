// MFMA tile helpers shared by the bound-evaluation kernel (nb_eval.hip) and
// the MVEE kernel (nb_mvee.hip): the B-operand block of 16-point tiles and the
// ellipsoid transform y = B_inv (x - c) on v_mfma_f64_16x16x4_f64.
#pragma once
#include "nb_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

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

// TPW = 16-point tiles per wavefront (template parameter): 2 shares every A
// operand between two tiles; 1 halves the register footprint (n_dim > 64).

// y = B_inv (x - c) on the matrix cores plus the per-dimension box test, for
// the TPW tiles of a wavefront (the A operand is shared).  r2 = |y|^2
// (replicated over the 4 lanes of a point); box_bad is set if any coordinate
// violates the member's [lo, hi) limits.
template <int DT, int TPW>
__device__ __forceinline__ void ell_eval(const double* blk,
                                         int n_dim,
                                         const double (&xin)[TPW][4 * DT],
                                         int lane, double (&y)[TPW][4 * DT],
                                         bool (&box_bad)[TPW],
                                         double (&r2)[TPW]) {
  constexpr int DP = 16 * DT;
  const double* lo = blk + 2;
  const double* hi = lo + DP;
  const double* c = hi + DP;
  const double* tiles = c + DP;
  const long long n_ell = ((const long long*)blk)[0];
  const int lg = lane >> 4;

  double d[TPW][4 * DT];
  bool bad[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) bad[t] = false;
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const int f = 4 * ks + lg;               // slot index (host permuted)
    const double lov = lo[f], hiv = hi[f], cv = c[f];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const double xv = xin[t][ks];
      bad[t] |= !(xv >= lov && xv < hiv);
      d[t][ks] = xv - cv;
    }
  }
  double part[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    box_bad[t] = point_any(bad[t], lane);
    part[t] = 0.0;
  }
  if (n_ell > 0) {
#pragma unroll
    for (int ht = 0; ht < DT; ++ht) {
      if (16 * ht < n_dim) {
        nb_d4 acc[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4 * (ht + 1); ++ks) {   // lower-triangular
          const int kt = ks >> 2, s = ks & 3;
          const double a = tiles[(kt * DT + ht) * NB_TILE + s * 64 + lane];
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t] = MFMA(a, d[t][ks], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            y[t][4 * ht + r] = acc[t][r];
            part[t] += acc[t][r] * acc[t][r];
          }
      } else {
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) y[t][4 * ht + r] = 0.0;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) y[t][ks] = 0.0;
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t) r2[t] = lane_group_sum(part[t]);
}

// B-operand block of the points: lane l holds feature 4*ks + (l >> 4) of
// point (l & 15).  Re-read (L1/L2 hits) wherever it is needed instead of being
// kept live across the emulator evaluation, which needs the registers.
// K permutation shared with nb_stream.hip: slot ks of lane group lg holds
// feature perm(ks, lg) = 8*(ks>>1) + 2*lg + (ks&1), so that a lane reads its
// slots (2j, 2j+1) with one 16-byte load and a point is covered by 64
// contiguous bytes per instruction.  The per-dimension vectors (lo, hi, c) and
// the K index of the ellipsoid tiles are stored in slot order by the host.
template <int DT, int TPW>
__device__ __forceinline__ void load_points(const nb_gd* __restrict__ x,
                                            const long long (&pt)[TPW],
                                            const bool (&valid)[TPW],
                                            int n_dim, long long n, int lane,
                                            double (&xin)[TPW][4 * DT],
                                            const double* shift = nullptr) {
  // (opaque: the clamped column offsets and padding predicates below are
  // invariants of the callers' loops; hoisted out of them they are 6 DT
  // registers held for the whole kernel)
  int lg = lane >> 4;
  asm volatile("" : "+v"(lg));
  const bool even = (n_dim & 1) == 0;
  // the loads are loop invariant across the bounds of a list; laundering the
  // base pointer keeps the compiler from hoisting them (and the registers
  // they occupy) out of the bound loop
  asm volatile("" : "+s"(x));
  if (even) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const nb_gd* row = x + (valid[t] ? pt[t] : n - 1) * n_dim;
#pragma unroll
      for (int j = 0; j < 2 * DT; ++j) {
        const int f = 8 * j + 2 * lg;
        const bool in = valid[t] && f < n_dim;
        const double2 v =
            *(const NB_G double2*)(row + (f < n_dim ? f : n_dim - 2));
        xin[t][2 * j] = in ? v.x : 0.0;
        xin[t][2 * j + 1] = in ? v.y : 0.0;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const nb_gd* row = x + (valid[t] ? pt[t] : n - 1) * n_dim;
#pragma unroll
      for (int j = 0; j < 2 * DT; ++j) {
        const int f = 8 * j + 2 * lg;
        const double v0 = row[f < n_dim ? f : n_dim - 1];
        const double v1 = row[f + 1 < n_dim ? f + 1 : n_dim - 1];
        xin[t][2 * j] = (valid[t] && f < n_dim) ? v0 : 0.0;
        xin[t][2 * j + 1] = (valid[t] && f + 1 < n_dim) ? v1 : 0.0;
      }
    }
  }
  // periodic dimensions are recentred before the test (nautilus.py:162-163,
  // periodic.py:69-71): x <- (x + (0.5 - centre)) mod 1
  if (shift != nullptr) {
    constexpr int DP = 16 * DT;
#pragma unroll
    for (int ks = 0; ks < 4 * DT; ++ks) {
      const double sv = shift[4 * ks + lg];
      const bool on = shift[DP + 4 * ks + lg] != 0.0;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const double v = xin[t][ks] + sv;
        xin[t][ks] = on ? v - floor(v) : xin[t][ks];
      }
    }
  }
}

}  // namespace
