// Ellipsoid.contains as an HBM-streaming kernel (the north-star roofline
// kernel; reference nautilus/bounds/basic.py:340,360):
//     mask_i = | B_inv (x_i - c) |^2 < 1        (strict)
//
// Algorithmic traffic: 8*D bytes read + 1 byte written per point (SURVEY.md
// section 8d); D(D+1) flop per point (B_inv is lower triangular: it is
// inv(cholesky(A^-1)), basic.py:308-309).
//
// Design (CDNA4).  Two VALU designs (one point per lane, B_inv through scalar
// loads or LDS broadcasts) stalled at 20-26 % of HBM peak: 1275 distinct
// matrix entries per 64 points cannot be fed to v_fma_f64 fast enough (SGPR
// spills / LDS issue).  The matrix cores take the matrix as a *vector*
// operand, so one 512-byte LDS read feeds 4 x 1024 FMAs:
//  * a wavefront owns 4 (D <= 64) or 2 tiles of 16 points; rows of y = B_inv (x - c) are the
//    MFMA rows, the points are the columns, K runs over the features;
//  * B_inv lives in LDS as lower-triangular 16x16 tiles (20 KB at D = 50),
//    loaded once per workgroup; every A operand is shared by the 4 tiles;
//  * x is read exactly once with 16-byte loads: lane (li, lg) reads the pairs
//    (8j + 2lg, 8j + 2lg + 1) of point li -- 64 contiguous bytes per point and
//    instruction -- and the K index of the MFMA is permuted to match (the
//    host packs the tiles with the same permutation), so no transpose is
//    needed anywhere.  Rows of an odd n_dim are 8-byte aligned only; the
//    same pair loads work there (global_load_dwordx4 needs dword alignment):
//    0.55 of the HBM peak at n_dim 49 against 0.62 at 50, where a kernel
//    that staged 16-point blocks in LDS (global_load_lds) and gathered the
//    operands with 8-byte LDS reads reached 0.47
//    (profiles/r05/fifth_session/stream_odd_ab.txt);
//  * r^2 is reduced over the 4 lanes of a point with two xor-shuffles.
#include "nb_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

namespace {

// |B_inv (x - c)|^2 contributions of one group of TPW tiles: d holds the
// centred inputs in the permuted K order.  Work is trimmed to the real
// dimension: k-steps beyond ceil(n_dim / 4) hold only zero padding and are
// skipped, and a last row tile with at most 4 real rows (n_dim mod 16 in
// 1..4, e.g. D = 50 or 20) runs on v_mfma_f64_4x4x4_4b_f64 -- 16 instead of 64
// cycles per k-step (A lane i + 4b + 16k, B lane p + 16k, D lane p + 16i; the
// A operand is gathered from the same tile storage).  At D = 50 that removes
// a third of the matrix cycles of a kernel that sits at the ridge between
// the HBM and the fp64 MFMA roof.
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// KL = number of k-steps that hold real features (the K permutation pairs
// k-steps: 2j, 2j+1 cover features 8j .. 8j+7, so KL = 2 ceil(n_dim / 8));
// SMALL = the last row tile has at most 4 real rows.  Both are compile-time
// so that the MFMA sequences stay branch free.
template <int DT, int TPW, int KL, bool SMALL>
__device__ __forceinline__ void stream_quadform(const double* wl, int n_dim,
                                                int lane,
                                                const double (&d)[TPW][4 * DT],
                                                double (&part)[TPW]) {
#pragma unroll
  for (int t = 0; t < TPW; ++t) part[t] = 0.0;
#pragma unroll
  for (int ht = 0; ht < DT; ++ht) {
    const int ks_n = (4 * (ht + 1) < KL) ? 4 * (ht + 1) : KL;   // lower-tri
    // (always true; the run-time test keeps the row tiles in separate basic
    // blocks -- merged, the scheduler hoists every operand read and the kernel
    // spills 220 registers)
    if (16 * ht >= n_dim) continue;
    if (SMALL && ht == DT - 1) {
      double r[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) r[t] = 0.0;
      const int roff = (lane >> 4) * 16 + (lane & 3);
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        if (ks < ks_n) {
          const int kt = ks >> 2, s = ks & 3;
          const double a =
              wl[((ht * (ht + 1)) / 2 + kt) * NB_TILE + s * 64 + roff];
#pragma unroll
          for (int t = 0; t < TPW; ++t) r[t] = MFMA4(a, d[t][ks], r[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) part[t] = fma(r[t], r[t], part[t]);
    } else {
      nb_d4 acc[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        if (ks < ks_n) {
          const int kt = ks >> 2, s = ks & 3;
          const double a =
              wl[((ht * (ht + 1)) / 2 + kt) * NB_TILE + s * 64 + lane];
#pragma unroll
          for (int t = 0; t < TPW; ++t) acc[t] = MFMA(a, d[t][ks], acc[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[t] = fma(acc[t][r], acc[t][r], part[t]);
    }
  }
}

// The same with the A operands read AHEAD of their MFMAs: left to the
// scheduler every LDS read sits directly in front of its first MFMA, and with
// one or two tiles per wavefront (n_dim > 64) an operand feeds 64-128 cycles
// of matrix work behind ~120 cycles of exposed LDS latency.  The k-steps go
// in chunks of four; the reads of a chunk are issued in front of the MFMAs of
// the chunk before it (256-512 cycles of cover, eight more registers) and
// pinned there by scheduling barriers.
template <int DT, int TPW, int KL, bool SMALL>
__device__ __forceinline__ void stream_quadform_ahead(
    const double* wl, int lane, const double (&d)[TPW][4 * DT],
    double (&part)[TPW]) {
  double a[DT][4 * DT];
  auto read_chunk = [&](int ht, int c) __attribute__((always_inline)) {
    const int ks_n = (4 * (ht + 1) < KL) ? 4 * (ht + 1) : KL;
    // (a last row tile of at most 4 rows: the A operand of the 4x4x4 tile)
    const int off = (SMALL && ht == DT - 1)
                        ? (lane >> 4) * 16 + (lane & 3) : lane;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (4 * c + s < ks_n)
        a[ht][4 * c + s] =
            wl[((ht * (ht + 1)) / 2 + c) * NB_TILE + s * 64 + off];
  };
#pragma unroll
  for (int t = 0; t < TPW; ++t) part[t] = 0.0;
  read_chunk(0, 0);
#pragma unroll
  for (int ht = 0; ht < DT; ++ht) {
    const int ks_n = (4 * (ht + 1) < KL) ? 4 * (ht + 1) : KL;
    const int n_c = (ks_n + 3) / 4;
    if (SMALL && ht == DT - 1) {
      // 4x4x4 tiles (16 instead of 64 cycles per k-step); even and odd
      // k-steps in accumulators of their own: with one or two tiles per
      // wavefront a single chain waits for its own results
      double r[TPW][2];
#pragma unroll
      for (int t = 0; t < TPW; ++t) r[t][0] = r[t][1] = 0.0;
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        if (c < n_c) {
          if (c + 1 < n_c) read_chunk(ht, c + 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (4 * c + s < ks_n) {
#pragma unroll
              for (int t = 0; t < TPW; ++t)
                r[t][s & 1] = MFMA4(a[ht][4 * c + s], d[t][4 * c + s],
                                    r[t][s & 1]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const double y = r[t][0] + r[t][1];
        part[t] = fma(y, y, part[t]);
      }
    } else {
      nb_d4 acc[TPW];
#pragma unroll
      for (int t = 0; t < TPW; ++t) acc[t] = nb_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        if (c < n_c) {
          if (c + 1 < n_c) read_chunk(ht, c + 1);
          else if (ht + 1 < DT) read_chunk(ht + 1, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (4 * c + s < ks_n) {
#pragma unroll
              for (int t = 0; t < TPW; ++t)
                acc[t] = MFMA(a[ht][4 * c + s], d[t][4 * c + s], acc[t]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[t] = fma(acc[t][r], acc[t][r], part[t]);
    }
  }
}

template <int DT, int TPW, int KL, bool SMALL>
__global__ void __launch_bounds__(256, 2)
nb_ell_stream_kernel(const double* __restrict__ cvec,
                     const double* __restrict__ tiles,   // permuted, lower-tri
                     int n_dim, const double* __restrict__ x, long long n,
                     unsigned char* __restrict__ mask) {
  constexpr int NT = DT * (DT + 1) / 2;
  __shared__ __attribute__((aligned(16))) double wl[NT * NB_TILE];
  for (int i = 2 * threadIdx.x; i < NT * NB_TILE; i += 512)
    *(double2*)(wl + i) = *(const double2*)(tiles + i);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const bool even = (n_dim & 1) == 0;

  // centre in the permuted K order: slot (2j + o) <-> feature 8j + 2lg + o
  double cper[4 * DT];
#pragma unroll
  for (int j = 0; j < 2 * DT; ++j) {
    const int f = 8 * j + 2 * lg;
    cper[2 * j] = cvec[f];              // cvec is zero padded to 16*DT
    cper[2 * j + 1] = cvec[f + 1];
  }

  const long long n_groups = (n + 16 * TPW - 1) / (16 * TPW);
  for (long long grp = (long long)blockIdx.x * 4 + wave; grp < n_groups;
       grp += (long long)gridDim.x * 4) {
    // branch-free loads: out-of-range rows / columns are clamped to a valid
    // address and zeroed by a select, so all loads of a group issue
    // back to back
    double d[TPW][4 * DT];
    if (even) {
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const long long pt = (grp * TPW + t) * 16 + li;
        const bool ok = pt < n;
        const double* row = x + (ok ? pt : n - 1) * n_dim;
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) {
          const int f = 8 * j + 2 * lg;
          const bool in = ok && f < n_dim;
          const double2 v = *(const double2*)(row + (f < n_dim ? f : n_dim - 2));
          d[t][2 * j] = (in ? v.x : 0.0) - cper[2 * j];
          d[t][2 * j + 1] = (in ? v.y : 0.0) - cper[2 * j + 1];
        }
      }
    } else if (n_dim >= 3) {
      // odd n_dim: the rows are 8-byte aligned only.  global_load_dwordx4
      // needs dword alignment, not 16 bytes, so the same pair loads work from
      // an 8-byte-aligned address (nb_d2u); the pair that holds the row's
      // last feature is read one element earlier -- (x[D-2], x[D-1]) -- so
      // that nothing behind the array is touched.
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const long long pt = (grp * TPW + t) * 16 + li;
        const bool ok = pt < n;
        const double* row = x + (ok ? pt : n - 1) * n_dim;
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) {
          const int f = 8 * j + 2 * lg;
          const bool full = f + 1 < n_dim, half = f + 1 == n_dim;
          const nb_d2u v =
              *(const nb_d2u*)(row + (full ? f : (half ? f - 1 : 0)));
          const double v0 = half ? v.y : v.x;
          d[t][2 * j] = ((ok && (full || half)) ? v0 : 0.0) - cper[2 * j];
          d[t][2 * j + 1] = ((ok && full) ? v.y : 0.0) - cper[2 * j + 1];
        }
      }
    } else {
      // n_dim = 1: one 8-byte load per point (a pair load would reach in
      // front of or behind the array)
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const long long pt = (grp * TPW + t) * 16 + li;
        const bool ok = pt < n;
        const double v0 = x[ok ? pt : n - 1];
#pragma unroll
        for (int j = 0; j < 2 * DT; ++j) {
          d[t][2 * j] = ((ok && j == 0 && lg == 0) ? v0 : 0.0) - cper[2 * j];
          d[t][2 * j + 1] = 0.0 - cper[2 * j + 1];
        }
      }
    }

    double part[TPW];
    if constexpr (DT >= 5)
      stream_quadform_ahead<DT, TPW, KL, SMALL>(wl, lane, d, part);
    else
      stream_quadform<DT, TPW, KL, SMALL>(wl, n_dim, lane, d, part);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      double r2 = part[t];
      r2 += __shfl_xor(r2, 16);
      r2 += __shfl_xor(r2, 32);
      const long long pt = (grp * TPW + t) * 16 + li;
      if (lg == 0 && pt < n) mask[pt] = (r2 < 1.0) ? 1 : 0;
    }
  }
}

// The same kernel software-pipelined for n_dim <= 64: a wavefront works on
// UNITS of two tiles (32 points) and the loads of the unit after the current
// one are in flight while the current one is multiplied (two register
// buffers, the loop unrolled by two).  With four tiles loaded and then
// multiplied the rate was 0.56 / 0.64 / 0.55 / 0.60 of the HBM peak at n_dim
// 49 / 50 / 63 / 64; pipelined 0.63 / 0.64 / 0.61 / 0.62
// (profiles/r05/fifth_session/stream_pipe_ab.txt).
// All loads are unconditional -- a unit past the end is the last unit again,
// recomputed and stored with the same values -- because the wait counts in
// front of the MFMA chains must be exact (behind a conditional load the
// compiler waits for the whole memory queue).
template <int DT, int UT>
struct StreamRaw { nb_d2u v[UT][2 * DT]; };

template <int DT, int UT>
__device__ __forceinline__ void stream_issue(const double* __restrict__ x,
                                             long long n, int n_dim,
                                             long long unit, int li, int lg,
                                             StreamRaw<DT, UT>& raw) {
  const bool even = (n_dim & 1) == 0;
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    const long long pt = (unit * UT + t) * 16 + li;
    const double* row = x + (pt < n ? pt : n - 1) * n_dim;
#pragma unroll
    for (int j = 0; j < 2 * DT; ++j) {
      const int f = 8 * j + 2 * lg;
      int at;
      if (even) {
        at = f < n_dim ? f : n_dim - 2;
      } else {
        const bool full = f + 1 < n_dim, half = f + 1 == n_dim;
        at = full ? f : (half ? f - 1 : 0);
      }
      raw.v[t][j] = *(const nb_d2u*)(row + at);
    }
  }
}

template <int DT, int UT, int KL, bool SMALL>
__device__ __forceinline__ void stream_consume(
    const double* wl, const double (&cper)[4 * DT], long long n, int n_dim,
    long long unit, int lane, const StreamRaw<DT, UT>& raw,
    unsigned char* __restrict__ mask) {
  const int li = lane & 15, lg = lane >> 4;
  const bool even = (n_dim & 1) == 0;
  // No masking of the loaded values: a slot past n_dim holds an element of
  // the point's own row (the clamped address) and meets exact zeros in the
  // tiles (nb_api.hip packs B_inv[h][k] = 0 for k >= n_dim), a lane past the
  // end of the array computes on the last row and stores nothing.  (The
  // selects were ~4 of the ~6 VALU instructions per loaded pair, on a kernel
  // whose matrix pipe and memory system are both ~64 % busy.)
  double d[UT][4 * DT];
#pragma unroll
  for (int t = 0; t < UT; ++t) {
#pragma unroll
    for (int j = 0; j < 2 * DT; ++j) {
      const int f = 8 * j + 2 * lg;
      const nb_d2u v = raw.v[t][j];
      const double v0 = (!even && f + 1 == n_dim) ? v.y : v.x;
      d[t][2 * j] = v0 - cper[2 * j];
      d[t][2 * j + 1] = v.y - cper[2 * j + 1];
    }
  }
  double part[UT];
  // (one tile per unit beyond 64 dimensions: operands read ahead)
  if constexpr (DT >= 5)
    stream_quadform_ahead<DT, UT, KL, SMALL>(wl, lane, d, part);
  else
    stream_quadform<DT, UT, KL, SMALL>(wl, n_dim, lane, d, part);
#pragma unroll
  for (int t = 0; t < UT; ++t) {
    double r2 = part[t];
    r2 += __shfl_xor(r2, 16);
    r2 += __shfl_xor(r2, 32);
    const long long pt = (unit * UT + t) * 16 + li;
    if (lg == 0 && pt < n) mask[pt] = (r2 < 1.0) ? 1 : 0;
  }
}

template <int DT, int UT, int KL, bool SMALL>
__global__ void __launch_bounds__(256, 2)
nb_ell_stream_pipe_kernel(const double* __restrict__ cvec,
                          const double* __restrict__ tiles, int n_dim,
                          const double* __restrict__ x, long long n,
                          unsigned char* __restrict__ mask) {
  constexpr int NT = DT * (DT + 1) / 2;
  __shared__ __attribute__((aligned(16))) double wl[NT * NB_TILE];
  for (int i = 2 * threadIdx.x; i < NT * NB_TILE; i += 512)
    *(double2*)(wl + i) = *(const double2*)(tiles + i);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  double cper[4 * DT];
#pragma unroll
  for (int j = 0; j < 2 * DT; ++j) {
    const int f = 8 * j + 2 * lg;
    cper[2 * j] = cvec[f];              // cvec is zero padded to 16*DT
    cper[2 * j + 1] = cvec[f + 1];
  }
  const long long n_units = (n + 16 * UT - 1) / (16 * UT), last = n_units - 1;
  const long long stride = (long long)gridDim.x * 4;
  long long u = (long long)blockIdx.x * 4 + wave;
  if (u >= n_units) return;
  StreamRaw<DT, UT> a, b;
  stream_issue<DT, UT>(x, n, n_dim, u, li, lg, a);
  for (; u < n_units; u += 2 * stride) {
    const long long u1 = u + stride < n_units ? u + stride : last;
    const long long u2 = u + 2 * stride < n_units ? u + 2 * stride : last;
    stream_issue<DT, UT>(x, n, n_dim, u1, li, lg, b);
    stream_consume<DT, UT, KL, SMALL>(wl, cper, n, n_dim, u, lane, a, mask);
    stream_issue<DT, UT>(x, n, n_dim, u2, li, lg, a);
    stream_consume<DT, UT, KL, SMALL>(wl, cper, n, n_dim, u1, lane, b, mask);
  }
}

template <int DT, int KL, bool SMALL>
int launch_variant(const double* cvec, const double* tiles, int n_dim,
                   const double* x, long long n, unsigned char* mask,
                   hipStream_t stream) {
  // 4 tiles per wavefront while the operands fit the register file, one
  // beyond 96 dimensions (two tiles of 28 slots spill 26-42 registers)
  constexpr int TPW = (DT <= 4) ? 4 : (DT <= 6 ? 2 : 1);
  // (n_dim <= 16: the plain kernel is 3 % ahead)
  if constexpr (DT >= 2 && DT <= 4) {
    {
      const long long n_units = (n + 31) / 32;
      long long blocks = (n_units + 3) / 4;
      if (blocks > 256 * 2 * 2) blocks = 256 * 2 * 2;
      hipLaunchKernelGGL((nb_ell_stream_pipe_kernel<DT, 2, KL, SMALL>),
                         dim3((unsigned)blocks), dim3(256), 0, stream, cvec,
                         tiles, n_dim, x, n, mask);
      return NB_OK;
    }
  }
  // 65 <= n_dim <= 112: units of ONE tile, operands read ahead (n_dim 96:
  // 0.99 -> 0.845 ms per 2^22 points against two tiles loaded and multiplied
  // in turn, 100: 0.955 -> 0.925; profiles/r05/fifth_session/
  // stream_pipe1_ab.txt); beyond that two register buffers do not fit
  if constexpr (DT >= 5 && DT <= 7) {
    if (n_dim >= 3) {
      const long long n_units = (n + 15) / 16;
      long long blocks = (n_units + 3) / 4;
      if (blocks > 256 * 2 * 2) blocks = 256 * 2 * 2;
      hipLaunchKernelGGL((nb_ell_stream_pipe_kernel<DT, 1, KL, SMALL>),
                         dim3((unsigned)blocks), dim3(256), 0, stream, cvec,
                         tiles, n_dim, x, n, mask);
      return NB_OK;
    }
  }
  const long long n_groups = (n + 16 * TPW - 1) / (16 * TPW);
  long long blocks = (n_groups + 3) / 4;
  if (blocks > 256 * 2 * 2) blocks = 256 * 2 * 2;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((nb_ell_stream_kernel<DT, TPW, KL, SMALL>),
                     dim3((unsigned)blocks), dim3(256), 0, stream, cvec, tiles,
                     n_dim, x, n, mask);
  return NB_OK;
}

template <int DT>
int launch(const double* cvec, const double* tiles, int n_dim, const double* x,
           long long n, unsigned char* mask, hipStream_t stream) {
  const int kl = 2 * ((n_dim + 7) >> 3);
  const int rem = n_dim & 15;
  if (kl == 4 * DT)
    return launch_variant<DT, 4 * DT, false>(cvec, tiles, n_dim, x, n, mask,
                                             stream);
  // (beyond 64 dimensions -- one or two tiles per wavefront -- the 4x4x4
  // tiles pay since the operands are read ahead: stream_quadform_ahead)
  if (rem >= 1 && rem <= 4)
    return launch_variant<DT, 4 * DT - 2, true>(cvec, tiles, n_dim, x, n, mask,
                                                stream);
  return launch_variant<DT, 4 * DT - 2, false>(cvec, tiles, n_dim, x, n, mask,
                                               stream);
}

}  // namespace

// cvec / tiles point into the stream block of the blob (nb_common.h hdr[18]):
// c zero padded to 16*DT, then DT(DT+1)/2 lower-triangular 16x16 tiles of
// W0[k][h] = B_inv[h][k] with the K permutation described above.
int nb_launch_ell_stream(const double* cvec, const double* tiles, int n_dim,
                         const double* x, long long n, unsigned char* mask,
                         hipStream_t stream) {
  if (n <= 0) return NB_OK;
  const int dt = (n_dim + 15) / 16;
  int rc = NB_OK;
  switch (dt) {
    case 1: rc = launch<1>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 2: rc = launch<2>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 3: rc = launch<3>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 4: rc = launch<4>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 5: rc = launch<5>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 6: rc = launch<6>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 7: rc = launch<7>(cvec, tiles, n_dim, x, n, mask, stream); break;
    case 8: rc = launch<8>(cvec, tiles, n_dim, x, n, mask, stream); break;
    default:
      nb_set_error("n_dim > 128 unsupported");
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
