// Geometric stage of the bound evaluation: everything contains() / sample()
// decide WITHOUT an emulator -- periodic recentring, unit-cube clip, the
// ellipsoids of the outer union's members (overlap count, union.py:285-289,
// 316-319), the acceptance draw, the ellipsoids of the neural bounds
// (bounds/neural.py:115-120) -- for single bounds and for lists of bounds
// (shell exclusion, sampler.py:797-798; shell association, 1213-1219).
//
// A point leaves this kernel decided (inside / outside / accepted /
// rejected) or PENDING on one emulator: (bound b, neural bound m), the first
// neural bound whose ellipsoid contains it.  The host gathers the pending
// points per (b, m) and scores them densely with the pipelined emulator kernel
// (nb_eval_fast.hip, index indirection); points an emulator turns down come
// back here and resume their walk behind (b, m).  So the matrix-core work of
// the emulators -- 99 % of the flops -- only ever runs on full tiles of
// points that need it, whatever the number of outer members, neural bounds or
// bounds in the list; the geometric stage itself is HBM bound (8 D bytes per
// point and pass, ellipsoid blocks from L2).
//
// A SINGLE bound (proposal acceptance, contains) whose ellipsoid blocks --
// outer members and neural bounds -- fit into LDS together (7 blocks at
// n_dim = 50) keeps them there for the whole launch: a workgroup stages them
// once and then only streams points.  Staged per 128-point pass, the blocks
// cost more L2 traffic than the points cost HBM traffic (22 KB per block and
// pass against 51 KB of points at n_dim = 50).
//
// Workgroup = 8 wavefronts x one 16-point tile; an ellipsoid block (limits,
// centre, lower-triangular 16x16 tiles of B_inv^T) is staged in LDS for the
// 128 points of a pass and evaluated on v_mfma_f64_16x16x4_f64 like every
// other ellipsoid test of this library (nb_tile.h layout: lane l holds slot
// 4 ks + (l >> 4) of point l & 15).
#include "nb_common.h"

#include "nb_tile.h"

namespace {

constexpr int GM_NW = 8;

// status byte of a point
enum : unsigned char {
  GS_OUTER = 1,     // SAMPLE: kept by the outer union's acceptance draw
  GS_INSIDE = 2,    // contained (ANY / ASSOC) or finally accepted (SAMPLE)
  GS_PENDING = 4,   // waits for the emulator of (pos >> 8, pos & 255)
  GS_DONE = 8       // decided
};

enum { GM_ANY = 0, GM_ASSOC = 1, GM_SAMPLE = 2 };

struct GeomArgs {
  const double* const* blobs;   // device array of nb blob pointers
  int nb;
  int mode;
  const nb_gd* x;               // (n_rows, n_dim)
  const long long* idx;         // optional: rows to process (n entries)
  long long n;                  // points to process
  long long n_rows;
  int* pos;                     // per ROW: resume position b * 256 + m
  unsigned char* st;            // per ROW: status
  unsigned long long seed, offset;
  unsigned long long* counters; // optional, as in nb_eval.hip
  int resident;                 // single bound: all its blocks stay in LDS
};

// packed block in LDS: lo, hi, c (slot order), lower-triangular tiles
template <int DT>
struct GeomLds {
  static constexpr int DP = 16 * DT;
  static constexpr int NT = DT * (DT + 1) / 2;
  static constexpr int TILES = 3 * DP;
  static constexpr int TOTAL = TILES + NT * NB_TILE;
};

// (row tile, k tile) of the p-th lower-triangular tile, p = ht (ht + 1) / 2 + kt
struct TriTable {
  unsigned char ht[36], kt[36];
  constexpr TriTable() : ht(), kt() {
    int p = 0;
    for (int h = 0; h < 8; ++h)
      for (int k = 0; k <= h; ++k) {
        ht[p] = (unsigned char)h;
        kt[p] = (unsigned char)k;
        ++p;
      }
  }
};
__device__ constexpr TriTable TRI{};

// global ell block -> packed LDS block, whole workgroup, 16 bytes per lane:
// a quarter of the workgroup (128 threads) copies one 2 KB tile; all loads of
// a round of 16 tiles are issued before the first store (the copy of a block
// used to spend its time in index arithmetic and one load latency per tile)
template <int DT>
__device__ __forceinline__ void geom_stage(const nb_gd* blk, double* lds) {
  constexpr int DP = 16 * DT, NT = DT * (DT + 1) / 2;
  const int tid = threadIdx.x;
  if (2 * tid < 3 * DP) {
    const double2 v = *(const NB_G double2*)(blk + 2 + 2 * tid);
    *(double2*)(lds + 2 * tid) = v;
  }
  const int g = tid >> 7, e = 2 * (tid & 127);
  const nb_gd* src = blk + 2 + 3 * DP + e;
  double* dst = lds + 3 * DP + e;
  constexpr int ROUND = 4;                 // tiles per thread and round
#pragma unroll
  for (int p0 = 0; p0 < NT; p0 += 4 * ROUND) {
    double2 v[ROUND];
#pragma unroll
    for (int j = 0; j < ROUND; ++j) {
      const int p = p0 + 4 * j + g;
      if (p < NT)
        v[j] = *(const NB_G double2*)(
            src + ((int)TRI.kt[p] * DT + (int)TRI.ht[p]) * NB_TILE);
    }
#pragma unroll
    for (int j = 0; j < ROUND; ++j) {
      const int p = p0 + 4 * j + g;
      if (p < NT) *(double2*)(dst + p * NB_TILE) = v[j];
    }
  }
}

// box test + |B_inv (x - c)|^2 from the packed block
template <int DT>
__device__ __forceinline__ bool geom_inside(const double* blk, bool has_ell,
                                            int n_dim,
                                            const double (&xin)[4 * DT],
                                            int lane) {
  constexpr int DP = 16 * DT;
  const double* lo = blk;
  const double* hi = lo + DP;
  const double* c = hi + DP;
  const double* tiles = c + DP;
  const int lg = lane >> 4;
  double d[4 * DT];
  bool bad = false;
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const int f = 4 * ks + lg;
    const double xv = xin[ks];
    bad |= !(xv >= lo[f] && xv < hi[f]);
    d[ks] = xv - c[f];
  }
  bad = point_any(bad, lane);
  double part = 0.0;
  if (has_ell) {
#pragma unroll
    for (int ht = 0; ht < DT; ++ht) {
      if (16 * ht < n_dim) {
        nb_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4 * (ht + 1); ++ks) {
          const double a = tiles[(ht * (ht + 1) / 2 + (ks >> 2)) * NB_TILE +
                                 (ks & 3) * 64 + lane];
          acc = MFMA(a, d[ks], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part += acc[r] * acc[r];
      }
    }
  }
  const double r2 = lane_group_sum(part);
  return !bad && r2 < 1.0;
}

template <int DT>
__global__ void __launch_bounds__(64 * GM_NW)
nb_geom_kernel(GeomArgs a) {
  constexpr int DP = 16 * DT;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  __shared__ int sh_min_b;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const bool m_sample = a.mode == GM_SAMPLE;
  const nb_gd* const NB_G* blobs = (const nb_gd* const NB_G*)a.blobs;
  const nb_gd* blob0 = blobs[0];
  const int n_dim = (int)nb_hdr((const double*)blob0, NB_H_NDIM);
  const long long n_pass = (a.n + 16 * GM_NW - 1) / (16 * GM_NW);
  unsigned long long cnt_outer = 0, cnt_ell = 0;
  constexpr int BLK = GeomLds<DT>::TOTAL;
  // resident blocks: [members (unless the proposals skip them)] [neural]
  int res_members = 0;
  if (a.resident) {
    const double* hdr = (const double*)blob0;
    const int K = (int)nb_hdr(hdr, NB_H_K);
    const int M = (int)nb_hdr(hdr, NB_H_M);
    res_members = (m_sample && K == 1) ? 0 : K;
    const nb_gd* mblk = blob0 + nb_hdr(hdr, NB_H_OFF_MEMBERS);
    const nb_gd* nblk = blob0 + nb_hdr(hdr, NB_H_OFF_NEURAL);
    const long long ell_stride = nb_hdr(hdr, NB_H_ELL_STRIDE);
    const long long neural_stride = nb_hdr(hdr, NB_H_NEURAL_STRIDE);
    for (int j = 0; j < res_members; ++j)
      geom_stage<DT>(mblk + j * ell_stride, lds + j * BLK);
    for (int j = 0; j < M; ++j)
      geom_stage<DT>(nblk + j * neural_stride, lds + (res_members + j) * BLK);
    __syncthreads();
  }

  for (long long pass = blockIdx.x; pass < n_pass; pass += gridDim.x) {
    const long long p = (pass * GM_NW + wave) * 16 + (lane & 15);
    const bool valid = p < a.n;
    const long long row = valid ? (a.idx != nullptr ? a.idx[p] : p) : 0;
    int pos = valid ? a.pos[row] : 0;
    unsigned char st = valid ? a.st[row] : (unsigned char)GS_DONE;
    // undecided points walk on; pending ones were sent back by an emulator
    // that turned them down: resume behind (b, m)
    bool active = valid && !(st & GS_DONE);
    if (active && (st & GS_PENDING)) {
      st &= (unsigned char)~GS_PENDING;
      pos += 1;
    }
    if (threadIdx.x == 0) sh_min_b = a.nb;
    __syncthreads();
    if (active) atomicMin(&sh_min_b, pos >> 8);
    __syncthreads();
    const int b_first = sh_min_b;

    for (int b = b_first; b < a.nb; ++b) {
      const nb_gd* blob = blobs[b];
      const double* hdr = (const double*)blob;
      const int K = (int)nb_hdr(hdr, NB_H_K);
      const int M = (int)nb_hdr(hdr, NB_H_M);
      const int E = (int)nb_hdr(hdr, NB_H_E);
      const bool use_cube = nb_hdr(hdr, NB_H_USECUBE) != 0;
      const long long ell_stride = nb_hdr(hdr, NB_H_ELL_STRIDE);
      const long long neural_stride = nb_hdr(hdr, NB_H_NEURAL_STRIDE);
      const long long off_shift = nb_hdr(hdr, NB_H_OFF_SHIFT);
      const bool mine = active && (pos >> 8) <= b;
      if (!__syncthreads_or(mine ? 1 : 0)) continue;
      // contains() of a bound with periodic dimensions sees recentred points;
      // proposals already live in the shifted frame
      const double* shift =
          (off_shift != 0 && !m_sample) ? (const double*)blob + off_shift
                                        : nullptr;
      const int m_start = ((pos >> 8) == b) ? (pos & 255) : 0;
      const bool resumed = m_start > 0;       // the outer union said yes

      double xin[4 * DT];
      {
        const long long pts[1] = {row};
        const bool vs[1] = {valid};
        double xt[1][4 * DT];
        load_points<DT, 1>(a.x, pts, vs, n_dim, a.n_rows, lane,
                           xt, shift);
#pragma unroll
        for (int ks = 0; ks < 4 * DT; ++ks) xin[ks] = xt[0][ks];
      }
      const nb_gd* nblk0 = blob + nb_hdr(hdr, NB_H_OFF_NEURAL);

      // Bounding-sphere pre-test (bound lists): a point of a bound with
      // neural bounds lies inside one of their ellipsoids, hence within
      // sqrt(radius2) of that centre.  If no point of the workgroup passes,
      // the whole bound is skipped -- for nested bounds in high dimension all
      // but the next few bounds.
      if (!m_sample && M > 0) {
        bool maybe = false;
        for (int m = 0; m < M; ++m) {
          const nb_gd* nb_m = nblk0 + m * neural_stride;
          const double rad2 = nb_m[1];
          const nb_gd* cc = nb_m + 2 + 2 * DP;
          double d2 = 0.0;
#pragma unroll
          for (int ks = 0; ks < 4 * DT; ++ks) {
            const double dv = xin[ks] - cc[4 * ks + lg];
            d2 = fma(dv, dv, d2);
          }
          maybe |= mine && lane_group_sum(d2) <= rad2;
        }
        if (!__syncthreads_or(maybe ? 1 : 0)) continue;
      }

      // unit-cube clip of the union (union.py:287-288 / 313-314)
      bool cbad = false;
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        const int f = 8 * (ks >> 1) + 2 * lg + (ks & 1);
        cbad |= use_cube && f < n_dim && !(xin[ks] >= 0.0 && xin[ks] < 1.0);
      }
      const bool in_cube = !point_any(cbad, lane);

      // ---- outer union: overlap count ------------------------------------
      int k_cnt = 0;
      if (m_sample && K == 1) {
        k_cnt = 1;                       // drawn from the only member
      } else if (!__syncthreads_and((!mine || resumed || !in_cube) ? 1 : 0)) {
        const nb_gd* mblk = blob + nb_hdr(hdr, NB_H_OFF_MEMBERS);
        for (int m = 0; m < K; ++m) {
          const nb_gd* blk = mblk + m * ell_stride;
          const double* sblk = lds;
          if (a.resident) {
            sblk = lds + m * BLK;
          } else {
            __syncthreads();
            geom_stage<DT>(blk, lds);
            __syncthreads();
          }
          const bool has_ell = ((const NB_G long long*)blk)[0] > 0;
          k_cnt += geom_inside<DT>(sblk, has_ell, n_dim, xin, lane) ? 1 : 0;
        }
        cnt_outer += (unsigned long long)K *
                     __popcll(__ballot(mine && lg == 0));
      }
      const bool outer_ok = resumed || (in_cube && (K == 0 || k_cnt > 0));
      bool want;
      if (m_sample) {
        if (!resumed && mine) {
          double u0, u_acc;
          nb_uniform_pair(a.seed, a.offset + (unsigned long long)row, 0u,
                          NB_TAG_CTRL, u0, u_acc);
          // (no member contains it: 1 - 1 / 0 = -inf, kept -- as in the
          // reference, union.py:318-319)
          const bool acc = in_cube && (u_acc > 1.0 - 1.0 / (double)k_cnt);
          if (acc) st |= GS_OUTER;
        }
        want = mine && (st & GS_OUTER);
      } else {
        want = mine && outer_ok;
      }

      // ---- neural bounds: the first whose ellipsoid contains the point ----
      bool found = false, found_final = false;
      int found_m = 0;
      if (M == 0) {
        found = want;
        found_final = want;
      } else {
        for (int m = 0; m < M; ++m) {
          const bool test = want && !found && m >= m_start;
          if (!__syncthreads_or(test ? 1 : 0)) continue;
          const nb_gd* nb_m = nblk0 + m * neural_stride;
          const double* sblk = lds;
          if (a.resident) {
            sblk = lds + (res_members + m) * BLK;
          } else {
            __syncthreads();
            geom_stage<DT>(nb_m, lds);
            __syncthreads();
          }
          const bool inside = geom_inside<DT>(sblk, true, n_dim, xin, lane);
          cnt_ell += __popcll(__ballot(test && lg == 0));
          if (test && inside) {
            found = true;
            found_m = m;
            found_final = (E == 0);      // no emulator: the ellipsoid decides
          }
        }
      }
      if (mine) {
        if (found && found_final) {
          st |= GS_INSIDE | GS_DONE;
          pos = b << 8;
          active = false;
        } else if (found) {
          st |= GS_PENDING;
          pos = (b << 8) | found_m;
          active = false;
        } else if (m_sample) {
          st |= GS_DONE;                 // rejected
          active = false;
        } else {
          pos = (b + 1) << 8;            // next bound of the list
        }
      }
      if (m_sample) break;               // a single bound
    }
    if (active) st |= GS_DONE;           // no bound of the list contains it
    if (valid && lg == 0) {
      a.st[row] = st;
      a.pos[row] = pos;
    }
  }
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[0], cnt_outer);
    atomicAdd(&a.counters[1], cnt_ell);
  }
}

constexpr size_t GM_LDS_MAX = 160 * 1024 - 64;   // sh_min_b lives there too

template <int DT>
int launch_geom(GeomArgs a, int n_blocks, hipStream_t stream) {
  const size_t one = (size_t)GeomLds<DT>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    const size_t most = GM_LDS_MAX / one * one;
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_geom_kernel<DT>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", most,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    configured = true;
  }
  const long long n_pass = (a.n + 16 * GM_NW - 1) / (16 * GM_NW);
  // Resident blocks pay when a workgroup sees several passes: one workgroup
  // per CU then, staging once.  (Few passes: stage per pass, more
  // workgroups.)
  a.resident = (a.nb == 1 && n_blocks >= 1 &&
                (size_t)n_blocks * one <= GM_LDS_MAX && n_pass >= 4 * 256)
                   ? 1 : 0;
  const size_t lds = a.resident ? (size_t)n_blocks * one : one;
  long long blocks = n_pass < 2048 ? n_pass : 2048;
  if (a.resident) blocks = lds > 80 * 1024 ? 256 : 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((nb_geom_kernel<DT>), dim3((unsigned)blocks),
                     dim3(64 * GM_NW), lds, stream, a);
  return NB_OK;
}

}  // namespace

unsigned long long* nb_eval_counters();

int nb_launch_geom(int dt, const double* const* blobs_dev, int nb, int mode,
                   int n_blocks, const double* x, long long n_rows,
                   const long long* idx, long long n, int* pos,
                   unsigned char* st, unsigned long long seed,
                   unsigned long long offset, hipStream_t stream) {
  if (n <= 0 || nb <= 0) return NB_OK;
  GeomArgs a;
  a.blobs = blobs_dev; a.nb = nb; a.mode = mode; a.x = (const nb_gd*)x;
  a.idx = idx; a.n = n; a.n_rows = n_rows; a.pos = pos; a.st = st;
  a.seed = seed; a.offset = offset; a.counters = nb_eval_counters();
  a.resident = 0;
  int rc = NB_OK;
  switch (dt) {
    case 1: rc = launch_geom<1>(a, n_blocks, stream); break;
    case 2: rc = launch_geom<2>(a, n_blocks, stream); break;
    case 3: rc = launch_geom<3>(a, n_blocks, stream); break;
    case 4: rc = launch_geom<4>(a, n_blocks, stream); break;
    case 5: rc = launch_geom<5>(a, n_blocks, stream); break;
    case 6: rc = launch_geom<6>(a, n_blocks, stream); break;
    case 7: rc = launch_geom<7>(a, n_blocks, stream); break;
    case 8: rc = launch_geom<8>(a, n_blocks, stream); break;
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
