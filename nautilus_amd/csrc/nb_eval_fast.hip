// Proposal acceptance / emulator scores of ONE bound with ONE neural bound --
// the case NautilusBound.sample spends its time in (nautilus.py:193-244: a
// proposal from the outer union is kept if the neural bound contains it,
// bounds/neural.py:115-126) -- as a software pipeline over the 128-point
// passes of a workgroup.  Same arithmetic as nb_eval.hip (bit-identical
// scores), different schedule:
//
//  * everything that does not change between passes stays in LDS for the whole
//    launch: the ellipsoid block (lower-triangular tiles only), the cube
//    limits, the standardisation vectors (nb_eval.hip re-fetches them per
//    pass, the small vectors with one dependent global load per k-step);
//  * the points of pass p + 1 are loaded into registers during the last MLP
//    stage of pass p (the standardised input's registers are dead there), and
//    layer 1 of the first network is fetched into its region during that stage
//    as well, so a pass starts with everything it needs on the CU;
//  * the MLP layers run as one operand pipeline per stage with the weight DMA
//    of the next stage sliced into the k-steps (nb_mlp.h).
//
// LDS: [resident block][region A: layer 1, KT1 x 7 tiles][region B: layers
// 2-4, 38 tiles]; 158 KB at n_dim = 50.  Eligible: n_dim <= 63 (region A),
// one neural bound with E >= 1 networks, MODE_SAMPLE with at most one outer
// member or MODE_SCORE; everything else goes through nb_eval.hip.
#include "nb_common.h"

#include <type_traits>

#include "nb_mlp.h"

// k-steps between two DMA instructions of a wavefront in stage 1 / stage 2,
// operand prefetch distance of layers 3 / 4
#ifndef NBF_P1
#define NBF_P1 2
#endif
#ifndef NBF_P2
#define NBF_P2 4
#endif
#ifndef NBF_TB
#define NBF_TB 1                  // DMA instructions per tick
#endif
#ifndef NBF_PD3
#define NBF_PD3 2
#endif
#ifndef NBF_PD4
#define NBF_PD4 5
#endif

namespace {

struct FastArgs {
  const double* blob;
  int sample;                     // 1: flags of nb_accept, 0: (r2, score)
  const double* x;
  long long n;
  unsigned char* out_u8;
  double* out_f64;
  unsigned long long seed;
  unsigned long long offset;
  unsigned long long* counters;   // optional, as in nb_eval.hip
};

constexpr int fast_resident_doubles(int dt) {
  return 2 + 5 * 16 * dt + dt * (dt + 1) / 2 * NB_TILE;
}
constexpr int FAST_B_DOUBLES =
    (NB_HT1 * NB_HT2 + NB_HT2 * NB_HT3 + NB_HT3) * NB_TILE;

typedef const void __attribute__((address_space(1))) * nbf_gptr;
typedef void __attribute__((address_space(3))) * nbf_lptr;

// y = B_inv (x - c) and the member's box test from the resident block
template <int DT, int T>
__device__ __forceinline__ void ell_eval_resident(
    const double* lo, const double* hi, const double* c, const double* tiles,
    int n_dim, const double (&xin)[T][4 * DT], int lane,
    double (&y)[T][4 * DT], bool (&box_bad)[T], double (&r2)[T]) {
  const int lg = lane >> 4;
  double d[T][4 * DT];
  bool bad[T];
#pragma unroll
  for (int t = 0; t < T; ++t) bad[t] = false;
#pragma unroll
  for (int ks = 0; ks < 4 * DT; ++ks) {
    const int f = 4 * ks + lg;
    const double lov = lo[f], hiv = hi[f], cv = c[f];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const double xv = xin[t][ks];
      bad[t] |= !(xv >= lov && xv < hiv);
      d[t][ks] = xv - cv;
    }
  }
  double part[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    box_bad[t] = point_any(bad[t], lane);
    part[t] = 0.0;
  }
#pragma unroll
  for (int ht = 0; ht < DT; ++ht) {
    if (16 * ht < n_dim) {
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
    const double* __restrict__ x, const long long (&pt)[T],
    const bool (&valid)[T], int n_dim, long long n, int lane,
    double2 (&raw)[T][2 * DT]) {
  const int lg = lane >> 4;
  asm volatile("" : "+s"(x));
  if ((n_dim & 1) == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const double* row = x + (valid[t] ? pt[t] : n - 1) * n_dim;
#pragma unroll
      for (int j = 0; j < 2 * DT; ++j) {
        const int f = 8 * j + 2 * lg;
        raw[t][j] = *(const double2*)(row + (f < n_dim ? f : n_dim - 2));
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const double* row = x + (valid[t] ? pt[t] : n - 1) * n_dim;
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
    int lane, double (&xin)[T][4 * DT]) {
  const int lg = lane >> 4;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 2 * DT; ++j) {
      const int f = 8 * j + 2 * lg;
      xin[t][2 * j] = (valid[t] && f < n_dim) ? raw[t][j].x : 0.0;
      xin[t][2 * j + 1] = (valid[t] && f + 1 < n_dim) ? raw[t][j].y : 0.0;
    }
}

template <int DT, int KT1>
__global__ void __launch_bounds__(256) nb_eval_fast_kernel(FastArgs a) {
  constexpr int T = 2, NW = 4, DP = 16 * DT;
  constexpr int KS1 = 4 * KT1;               // k-steps of layer 1 (padded)
  constexpr int SPLIT = DT <= 2 ? 8 : 2;     // output tiles per block
  constexpr int RES = fast_resident_doubles(DT);
  constexpr int NA_D = KT1 * NB_HT1 * NB_TILE;
  constexpr int NB_D = FAST_B_DOUBLES;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* res = lds;
  double* reg_a = lds + RES;
  double* reg_b = reg_a + NA_D;
  // resident block: [n_ell, thr][lo][hi][c][mean][1/scale][tiles]
  const double* r_lo = res + 2;
  const double* r_hi = r_lo + DP;
  const double* r_c = r_hi + DP;
  const double* r_mean = r_c + DP;
  const double* r_isc = r_mean + DP;
  const double* r_tiles = r_isc + DP;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lg = lane >> 4;
  const bool m_sample = a.sample != 0;
  const double* blob = a.blob;
  const int n_dim = (int)nb_hdr(blob, NB_H_NDIM);
  const int K = (int)nb_hdr(blob, NB_H_K);
  const bool use_cube = nb_hdr(blob, NB_H_USECUBE) != 0;
  const int E = (int)nb_hdr(blob, NB_H_E);
  const int ks1 = (n_dim + 1 + 3) >> 2;
  const double* nb_m = blob + nb_hdr(blob, NB_H_OFF_NEURAL);
  const long long net_stride = nb_hdr(blob, NB_H_NET_STRIDE);
  const double* nets = nb_m + nb_ell_block_size(DT) + 2 + 2 * DP;
  const long long n_super = (a.n + 16 * NW * T - 1) / (16 * NW * T);
  unsigned long long cnt_ell = 0, cnt_mlp = 0;

  long long sup = blockIdx.x;
  // ---- weight DMA (global_load_lds_dwordx4, 1 KB per instruction) --------
  const double* dma_src = nullptr;
  double* dma_dst = nullptr;
  int dma_c = 0, dma_n = 0;                 // chunk index of this wavefront
  auto dma_begin = [&](const double* src, double* dst, int n_doubles)
      __attribute__((always_inline)) {
    dma_src = src; dma_dst = dst; dma_c = wave; dma_n = n_doubles >> 7;
  };
  auto dma_tick = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBF_TB; ++i)
      if (dma_c < dma_n) {
        __builtin_amdgcn_global_load_lds(
            (nbf_gptr)(dma_src + dma_c * 128 + 2 * lane),
            (nbf_lptr)(dma_dst + dma_c * 128), 16, 0, 0);
        dma_c += NW;
      }
  };
  auto dma_flush = [&]() __attribute__((always_inline)) {
    while (dma_c < dma_n) {
      __builtin_amdgcn_global_load_lds(
          (nbf_gptr)(dma_src + dma_c * 128 + 2 * lane),
          (nbf_lptr)(dma_dst + dma_c * 128), 16, 0, 0);
      dma_c += NW;
    }
  };

  // ---- resident block ----------------------------------------------------
  {
    const double* mean = nb_m + nb_ell_block_size(DT) + 2;
    for (int i = threadIdx.x; i < 2 + 3 * DP; i += 64 * NW)
      res[i] = (i == 1) ? nb_m[nb_ell_block_size(DT)] : nb_m[i];
    for (int i = threadIdx.x; i < DP; i += 64 * NW) {
      res[2 + 3 * DP + i] = mean[i];
      res[2 + 4 * DP + i] = mean[DP + i];
    }
    const double* tsrc = nb_m + 2 + 3 * DP;
    double* tdst = res + 2 + 5 * DP;
#pragma unroll
    for (int ht = 0; ht < DT; ++ht)
#pragma unroll
      for (int kt = 0; kt <= ht; ++kt)
        for (int i = threadIdx.x; i < NB_TILE; i += 64 * NW)
          tdst[(ht * (ht + 1) / 2 + kt) * NB_TILE + i] =
              tsrc[(kt * DT + ht) * NB_TILE + i];
    dma_begin(nets, reg_a, NA_D);
    dma_flush();
  }

#ifdef NB_DBG_TIMING
  // cycle stamps of wavefront 0 of workgroup 0, accumulated in registers
  // (profiles/tools/fast_ts.py)
  long long t_prev = clock64();
  long long ts_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define NBF_TS(i) do { const long long t_now = clock64(); ts_acc[i] += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define NBF_TS(i)
#endif
  long long pt[T];
  bool valid[T];
  double2 xraw[T][2 * DT];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    pt[t] = ((sup * NW + wave) * T + t) * 16 + (lane & 15);
    valid[t] = pt[t] < a.n;
  }
  load_points_raw<DT, T>(a.x, pt, valid, n_dim, a.n, lane, xraw);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const double thr = res[1];

  for (; sup < n_super; sup += gridDim.x) {
    // ---- this pass's points have arrived in xraw --------------------------
    bool in_cube[T], acc_outer[T], want[T];
    double xin[T][4 * DT];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      pt[t] = ((sup * NW + wave) * T + t) * 16 + (lane & 15);
      valid[t] = pt[t] < a.n;
      in_cube[t] = true;
    }
    points_from_raw<DT, T>(xraw, valid, n_dim, lane, xin);
    if (m_sample) {
      // unit-cube clip of the union (union.py:313-314) and the acceptance of
      // the overlap-corrected draw (union.py:318-319) with k = K
      bool cbad[T];
#pragma unroll
      for (int t = 0; t < T; ++t) cbad[t] = false;
#pragma unroll
      for (int ks = 0; ks < 4 * DT; ++ks) {
        // slot (ks, lg) holds feature 8 (ks >> 1) + 2 lg + (ks & 1)
        const int f = 8 * (ks >> 1) + 2 * lg + (ks & 1);
        const bool boxed = use_cube && f < n_dim;
#pragma unroll
        for (int t = 0; t < T; ++t)
          cbad[t] |= boxed && !(xin[t][ks] >= 0.0 && xin[t][ks] < 1.0);
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        in_cube[t] = !point_any(cbad[t], lane);
        double u0, u_acc;
        nb_uniform_pair(a.seed, a.offset + (unsigned long long)pt[t], 0u,
                        NB_TAG_CTRL, u0, u_acc);
        acc_outer[t] = in_cube[t] && (u_acc > 1.0 - 1.0 / (double)K);
        want[t] = valid[t] && acc_outer[t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < T; ++t) { acc_outer[t] = false; want[t] = valid[t]; }
    }

    double y[T][4 * DT], r2[T];
    bool box_bad[T], inside_e[T], need[T];
    ell_eval_resident<DT, T>(r_lo, r_hi, r_c, r_tiles, n_dim, xin, lane, y,
                             box_bad, r2);
    NBF_TS(0);
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

    // standardised input (neural.py:115), constant 1 at column n_dim
    double tin[T][KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const int f = 4 * ks + lg;
      if (ks < 4 * DT) {
        const double mv = r_mean[f], sv = r_isc[f];
#pragma unroll
        for (int t = 0; t < T; ++t)
          tin[t][ks] = (f < n_dim) ? (y[t][ks] - mv) * sv
                                   : ((f == n_dim) ? 1.0 : 0.0);
      } else {
#pragma unroll
        for (int t = 0; t < T; ++t) tin[t][ks] = (f == n_dim) ? 1.0 : 0.0;
      }
    }

    double total[T];
#pragma unroll
    for (int t = 0; t < T; ++t) total[t] = 0.0;
    NBF_TS(1);

    // one network = two stages; region A holds its layer 1 on entry
    auto network = [&](int e, auto last_c) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(last_c)::value;
      const double* w_e = nets + e * net_stride;
      double h1[T][4 * NB_HT1];
      // -- stage 1: layer 1 from region A, layers 2-4 -> region B ----------
      dma_begin(w_e + NA_D, reg_b, NB_D);
      if (wave_mlp) {
        double a0[FlFirst<SPLIT, NB_HT1>::NA];
        fl_read_first<SPLIT, NB_HT1>(reg_a, lane, a0);
        fl_layer_from<T, SPLIT, KS1, 3, NB_HT1, false, NBF_P1, 1, 0>(
            reg_a, ks1, tin, lane, h1, a0,
            []() __attribute__((always_inline)) {}, dma_tick);
        fl_pad<T, NB_HT1, 25, 0>(h1, lane);                  // unit 100
      }
      NBF_TS(2);
      dma_flush();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      NBF_TS(3);
      // -- stage 2: layers 2-4 from region B, next layer 1 -> region A -----
      if constexpr (LAST) {
        // the next pass's points (tin is dead from here on)
        const long long nsup = sup + gridDim.x;
        long long npt[T];
        bool nvalid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          npt[t] = ((nsup * NW + wave) * T + t) * 16 + (lane & 15);
          nvalid[t] = npt[t] < a.n;
        }
        load_points_raw<DT, T>(a.x, npt, nvalid, n_dim, a.n, lane, xraw);
      }
      dma_begin(nets + (LAST ? 0 : (e + 1) * net_stride), reg_a, NA_D);
      if (wave_mlp) {
        const double* w2 = reg_b;
        const double* w3 = w2 + NB_HT1 * NB_HT2 * NB_TILE;
        const double* w4 = w3 + NB_HT2 * NB_HT3 * NB_TILE;
        double h2[T][4 * NB_HT2], h3[T][4 * NB_HT3], o[T][4];
        double a2[FlFirst<SPLIT, NB_HT2>::NA], a3[FlFirst<SPLIT, NB_HT3>::NA],
            a4[FlFirst<SPLIT, 1>::NA];
        fl_read_first<SPLIT, NB_HT2>(w2, lane, a2);
        fl_layer_from<T, SPLIT, 26, 0, NB_HT2, true, NBF_P2, 1, 0>(
            w2, 26, h1, lane, h2, a2,
            [&]() __attribute__((always_inline)) {
              fl_read_first<SPLIT, NB_HT3>(w3, lane, a3);
            },
            dma_tick);
        fl_pad<T, NB_HT2, 12, 2>(h2, lane);                  // unit 50
        fl_layer_from<T, SPLIT, 13, 0, NB_HT3, true, NBF_P2, NBF_PD3, 0>(
            w3, 13, h2, lane, h3, a3,
            [&]() __attribute__((always_inline)) {
              fl_read_first<SPLIT, 1>(w4, lane, a4);
            },
            dma_tick);
        fl_pad<T, NB_HT3, 5, 0>(h3, lane);                   // unit 20
        fl_layer_from<T, SPLIT, 6, 0, 1, true, NBF_P2, NBF_PD4, 0>(
            w4, 6, h3, lane, o, a4, []() __attribute__((always_inline)) {},
            dma_tick);
#pragma unroll
        for (int t = 0; t < T; ++t) total[t] += o[t][0];
      }
      NBF_TS(4);
      dma_flush();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      NBF_TS(5);
    };
    for (int e = 0; e + 1 < E; ++e) network(e, std::false_type{});
    network(E - 1, std::true_type{});

    // ---- epilogue ---------------------------------------------------------
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const double score = __shfl(total[t], lane & 15) / (double)E;
      if (m_sample) {
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
    NBF_TS(6);
  }
#ifdef NB_DBG_TIMING
  if (a.counters != nullptr && threadIdx.x == 0 && blockIdx.x == 0)
    for (int i = 0; i < 7; ++i) a.counters[8 + i] += ts_acc[i];
#endif
  if (a.counters != nullptr && lane == 0) {
    atomicAdd(&a.counters[1], cnt_ell);
    atomicAdd(&a.counters[2], cnt_mlp);
  }
}

template <int DT, int KT1>
int launch_fast(const FastArgs& a, hipStream_t stream) {
  const size_t lds = ((size_t)fast_resident_doubles(DT) +
                      (size_t)KT1 * NB_HT1 * NB_TILE + FAST_B_DOUBLES) *
                     sizeof(double);
  static bool configured = false;
  if (!configured) {
    const hipError_t e = hipFuncSetAttribute(
        (const void*)nb_eval_fast_kernel<DT, KT1>,
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      nb_set_error("hipFuncSetAttribute(%zu bytes LDS) failed: %s", lds,
                   hipGetErrorString(e));
      return NB_ERR_HIP;
    }
    configured = true;
  }
  const long long n_super = (a.n + 127) / 128;
  long long blocks = n_super < 256 ? n_super : 256;   // one workgroup per CU
  hipLaunchKernelGGL((nb_eval_fast_kernel<DT, KT1>), dim3((unsigned)blocks),
                     dim3(256), lds, stream, a);
  return NB_OK;
}

}  // namespace

unsigned long long* nb_eval_counters();

// n_dim <= 63, one neural bound with networks, and for proposals at most one
// outer member (the draw then needs no overlap count)
bool nb_eval_fast_eligible(int n_dim, int K, int M, int E, bool sample) {
  if (n_dim > 63 || M != 1 || E < 1) return false;
  return !sample || K <= 1;
}

int nb_launch_eval_fast(const double* blob_dev, int n_dim, bool sample,
                        const double* x, long long n, unsigned char* out_u8,
                        double* out_f64, unsigned long long seed,
                        unsigned long long offset, hipStream_t stream) {
  if (n <= 0) return NB_OK;
  FastArgs a;
  a.blob = blob_dev; a.sample = sample ? 1 : 0; a.x = x; a.n = n;
  a.out_u8 = out_u8; a.out_f64 = out_f64; a.seed = seed; a.offset = offset;
  a.counters = nb_eval_counters();
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
    default:
      nb_set_error("nb_launch_eval_fast: n_dim=%d not eligible", n_dim);
      return NB_ERR_UNSUPPORTED;
  }
  if (rc != NB_OK) return rc;
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
