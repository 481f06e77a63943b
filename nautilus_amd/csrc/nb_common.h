// Shared device/host definitions for the nautilus hot path on gfx950 (MI355X).
//
// Everything is fp64, like the reference (all arrays are
// float64 C-contiguous (N, D)).  Wavefront = 64 lanes, hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define NB_WAVE 64
#define NB_MAX_DT 8          // n_dim <= 128
#define NB_H1 100            // hidden sizes of the emulator, nautilus/neural.py:80
#define NB_H2 50
#define NB_H3 20
#define NB_HT1 7             // 16-wide tiles covering the hidden layers
#define NB_HT2 4
#define NB_HT3 2
#define NB_TILE 256          // doubles in one 16x16 tile

typedef double nb_d4 __attribute__((ext_vector_type(4)));
// a pair of doubles at an 8-byte-aligned address: one 16-byte load
// (global_load_dwordx4 needs dword alignment only) -- the rows of an odd n_dim
typedef double nb_d2u __attribute__((ext_vector_type(2), aligned(8)));

// Pointers to global memory, typed as such for device code.  A pointer the
// compiler cannot trace back to a kernel argument -- loaded from a descriptor
// in memory, or laundered through an empty asm -- is a generic ("flat")
// pointer to it: every access becomes flat_load / flat_store, which count
// against the LDS counter as well (each wait for an LDS read then also waits
// for every global load in flight) and cannot use scalar-base addressing.
// (Casting back and forth does not help: the optimiser folds the round trip;
// the pointer has to keep the address space in its type.)
#if defined(__HIP_DEVICE_COMPILE__)
#define NB_G __attribute__((address_space(1)))
#else
#define NB_G
#endif
typedef NB_G double nb_gd;
typedef NB_G int nb_gi;

// ---------------------------------------------------------------------------
// Device "bound blob": one contiguous array of doubles per bound, built on the
// host by nb_api.cpp (BlobBuilder) and uploaded once.  The first NB_HDR
// doubles hold integers (stored as int64 bit patterns) describing the layout;
// all offsets are in doubles from the start of the blob.
//
//   hdr[0] n_dim D          hdr[1] DT = ceil(D/16)     hdr[2] K outer members
//   hdr[3] use_cube         hdr[4] M neural bounds     hdr[5] E nets per bound
//   hdr[6] off_cdf (K doubles)                          hdr[7] off_ulo (DP)
//   hdr[8] off_uhi (DP)     hdr[9] off_members (K ell blocks, stride ell_stride)
//   hdr[10] ell_stride      hdr[11] off_neural (M neural blocks)
//   hdr[12] neural_stride   hdr[13] off_draw (K draw blocks)  hdr[14] draw_stride
//   hdr[15] mlp_net_stride  hdr[16] KT1 (k-tiles of MLP layer 1 incl. bias row)
//   hdr[17] total doubles   hdr[18] off_stream (single full ellipsoid only:
//                           c[DP], then DT(DT+1)/2 K-permuted 16x16 tiles of
//                           B_inv^T, see nb_stream.hip)
//   hdr[19] off_shift (0 = no periodic dimensions): shift[DP], on[DP] in slot
//                           order; contains() tests frac(x + shift) where on
//                           (bounds/periodic.py:50-72)
//
// Ell block (member of the outer union, or ellipsoid of a neural bound):
//   [0]            n_ell (as int64 bits; 0 => pure cube member, no MFMA work)
//   [1]            padding (keeps every tile 16-byte aligned)
//   [2 .. 2+DP)    lo   per-dimension lower limit (0 for cube dims, -inf else)
//   [..+DP)        hi   per-dimension upper limit (1 for cube dims, +inf else)
//   [..+DP)        c    centre embedded in full-D order (0 for cube dims)
//   [.. DT*DT tiles)  W0[k][h] = B_inv[h][k] embedded in full-D order, stored as
//                  16x16 tiles [kt][ht], tile element (kk, hh) at kk*16+hh
// Neural block = ell block, then:
//   thr, pad       score_predict_min - 1e-9 (bounds/neural.py:125)
//   mean[DP], inv_scale[DP]
//   E nets, each: L1 tiles [KT1][7], L2 [7][4], L3 [4][2], L4 [2][1]
//   (weights W_l[k][h] with the bias stored as row k = K_l; zero padded)
// Draw block (compact, for the per-proposal VALU draw kernel):
//   n_ell, n_cube, idx_ell[DP], idx_cube[DP], slot_of_column[DP] (as int64
//   bits), c[n_ell->DP], B packed lower-triangular row-major [DP*(DP+1)/2]
// ---------------------------------------------------------------------------
#define NB_HDR 32

enum {
  NB_H_NDIM = 0, NB_H_DT, NB_H_K, NB_H_USECUBE, NB_H_M, NB_H_E, NB_H_OFF_CDF,
  NB_H_OFF_ULO, NB_H_OFF_UHI, NB_H_OFF_MEMBERS, NB_H_ELL_STRIDE,
  NB_H_OFF_NEURAL, NB_H_NEURAL_STRIDE, NB_H_OFF_DRAW, NB_H_DRAW_STRIDE,
  NB_H_NET_STRIDE, NB_H_KT1, NB_H_TOTAL, NB_H_OFF_STREAM, NB_H_OFF_SHIFT
};

__host__ __device__ inline int64_t nb_hdr(const double* blob, int i) {
  return ((const int64_t*)blob)[i];
}

__host__ __device__ constexpr int nb_ell_block_size(int dt) {
  return 2 + 3 * dt * 16 + dt * dt * NB_TILE;   // even => 16-byte aligned tiles
}
// tiles of one network given KT1
__host__ __device__ inline int nb_net_tiles(int kt1) {
  return kt1 * NB_HT1 + NB_HT1 * NB_HT2 + NB_HT2 * NB_HT3 + NB_HT3 * 1;
}

// One (bound, neural bound) group of a two-stage query (nb_cand.hip ->
// nb_eval_fast.hip, BATCH): built on the host when a bound / a bound list is
// created, read by the second stage per 128-point pass.
struct FastGroup {
  const double* nb;               // neural block inside the bound's blob
  const double* shift;            // the bound's periodic shift block or null
  int E;                          // networks
  int b;                          // position of the bound in its list
};

// ---------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Stream layout: DESIGN.md "RNG
// contract" / oracle/philox.py.
// ---------------------------------------------------------------------------
#define NB_TAG_CTRL 0u
#define NB_TAG_NORMAL 1u
#define NB_TAG_CUBE 2u

struct nb_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline nb_u4 nb_philox(uint32_t c0, uint32_t c1, uint32_t c2,
                                           uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  nb_u4 o = {c0, c1, c2, c3};
  return o;
}

__host__ __device__ inline double nb_unit(uint32_t hi, uint32_t lo) {
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) *
         (1.0 / 9007199254740992.0);
}

// uniform in (0, 1) from ONE word: (w + 1/2) / 2^32 (the normals of the
// proposal draw: 2^-32 is far below anything a direction on the sphere can
// resolve statistically, and a Philox call then feeds two Box-Muller pairs)
__host__ __device__ inline double nb_unit32(uint32_t w) {
  return ((double)w + 0.5) * (1.0 / 4294967296.0);
}

// two uniforms in [0,1) for proposal g, block, tag
__host__ __device__ inline void nb_uniform_pair(uint64_t seed, uint64_t g,
                                                uint32_t block, uint32_t tag,
                                                double& u0, double& u1) {
  nb_u4 w = nb_philox((uint32_t)g, (uint32_t)(g >> 32), block, tag,
                      (uint32_t)seed, (uint32_t)(seed >> 32));
  u0 = nb_unit(w.x, w.y);
  u1 = nb_unit(w.z, w.w);
}

// ---------------------------------------------------------------------------
// error handling for the C ABI
// ---------------------------------------------------------------------------
#define NB_OK 0
#define NB_ERR_ARG 1
#define NB_ERR_HIP 2
#define NB_ERR_UNSUPPORTED 3

void nb_set_error(const char* fmt, ...);

#define NB_HIP_CHECK(expr)                                                   \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      nb_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                   __FILE__, __LINE__);                                      \
      return NB_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)
