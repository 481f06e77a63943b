// The live set on the device (reference nautilus/sampler.py:1147-1190:
// Sampler.f_live / log_v_live sort ALL stored log-likelihoods on every
// exploration iteration to find the n_live largest).  Here the candidates --
// every log L at or above the current threshold -- live in a small pool in
// HBM: a new batch appends its values above the threshold, one workgroup
// finds the exact n_live-th largest by radix selection (8 passes over 8 key
// bits, histograms in LDS) and drops what fell below it.  The threshold only
// ever rises, so the pool stays at n_live + one batch.  The per-shell sums
// over the live points are wavefront-shuffle reductions (nb_live_stats).
#include "nb_common.h"

namespace {

constexpr int LV_THREADS = 1024;

__device__ __forceinline__ unsigned long long lv_key(double v) {
  // order-preserving map double -> uint64 (ascending)
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void __launch_bounds__(256)
nb_live_append_kernel(const double* __restrict__ ll, long long n,
                      const double* __restrict__ thr, double* pool,
                      int* pool_n, int cap, int* overflow) {
  const double t = thr[0];
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    const double v = ll[i];
    if (v >= t) {
      const int slot = atomicAdd(pool_n, 1);
      if (slot < cap) pool[slot] = v;
      else *overflow = 1;
    }
  }
}

// stats[0] = threshold (the k-th largest value; -inf if fewer than k),
// stats[1] = #values > threshold, stats[2] = #values == threshold
__global__ void __launch_bounds__(LV_THREADS)
nb_live_select_kernel(const double* __restrict__ pool, const int* pool_n,
                      int cap, int k, double* out, int* out_n, double* thr,
                      double* stats) {
  __shared__ unsigned hist[256];
  __shared__ unsigned long long sh_prefix;
  __shared__ int sh_k, sh_cnt[2];
  const int tid = threadIdx.x;
  int p = *pool_n;
  if (p > cap) p = cap;
  double t = -__builtin_huge_val();
  if (p >= k) {
    if (tid == 0) { sh_prefix = 0ull; sh_k = k; }
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned long long prefix = sh_prefix;
      for (int i = tid; i < p; i += LV_THREADS) {
        const unsigned long long key = lv_key(pool[i]);
        if (pass == 0 || (key >> (shift + 8)) == prefix)
          atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int need = sh_k;
        int bin = 255;
        for (; bin > 0; --bin) {
          if ((int)hist[bin] >= need) break;
          need -= (int)hist[bin];
        }
        sh_prefix = (prefix << 8) | (unsigned long long)bin;
        sh_k = need;
      }
      __syncthreads();
    }
    const unsigned long long key = sh_prefix;
    const unsigned long long bits =
        (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    t = __longlong_as_double((long long)bits);
  }
  if (tid < 2) sh_cnt[tid] = 0;
  if (tid == 0) *out_n = 0;
  __syncthreads();
  int c_gt = 0, c_eq = 0;
  for (int i = tid; i < p; i += LV_THREADS) {
    const double v = pool[i];
    if (v >= t) {
      out[atomicAdd(out_n, 1)] = v;
      if (v > t) ++c_gt; else ++c_eq;
    }
  }
  atomicAdd(&sh_cnt[0], c_gt);
  atomicAdd(&sh_cnt[1], c_eq);
  __syncthreads();
  if (tid == 0) {
    thr[0] = t;
    stats[0] = t;
    stats[1] = (double)sh_cnt[0];
    stats[2] = (double)sh_cnt[1];
  }
}

struct LvAcc { double m, s, gt, eq; };

__device__ __forceinline__ LvAcc lv_merge(LvAcc a, LvAcc b) {
  LvAcc o;
  o.m = fmax(a.m, b.m);
  o.gt = a.gt + b.gt;
  o.eq = a.eq + b.eq;
  if (o.m == -__builtin_huge_val()) { o.s = 0.0; return o; }
  o.s = (a.m == -__builtin_huge_val() ? 0.0 : a.s * exp(a.m - o.m)) +
        (b.m == -__builtin_huge_val() ? 0.0 : b.s * exp(b.m - o.m));
  return o;
}

__device__ __forceinline__ LvAcc lv_wave(LvAcc v) {
  for (int d = 32; d >= 1; d >>= 1) {
    LvAcc o;
    o.m = __shfl_xor(v.m, d);
    o.s = __shfl_xor(v.s, d);
    o.gt = __shfl_xor(v.gt, d);
    o.eq = __shfl_xor(v.eq, d);
    v = lv_merge(v, o);
  }
  return v;
}

// per shell: out[0] = #(l > thr), out[1] = logsumexp(l | l > thr),
// out[2] = #(l == thr); one workgroup
__global__ void __launch_bounds__(1024)
nb_live_stats_kernel(const double* __restrict__ ll, long long n,
                     const double* __restrict__ thr, double* out) {
  __shared__ LvAcc sh[16];
  const double t = thr[0];
  const double ninf = -__builtin_huge_val();
  LvAcc acc = {ninf, 0.0, 0.0, 0.0};
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const double v = ll[i];
    LvAcc e = {ninf, 0.0, 0.0, 0.0};
    if (v > t) { e.m = v; e.s = 1.0; e.gt = 1.0; }
    else if (v == t) e.eq = 1.0;
    acc = lv_merge(acc, e);
  }
  acc = lv_wave(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    LvAcc tot = sh[0];
    for (int w = 1; w < 16; ++w) tot = lv_merge(tot, sh[w]);
    out[0] = tot.gt;
    out[1] = tot.m == ninf ? ninf : tot.m + log(tot.s);
    out[2] = tot.eq;
  }
}

}  // namespace

int nb_launch_live_append(const double* ll, long long n, const double* thr,
                          double* pool, int* pool_n, int cap, int* overflow,
                          hipStream_t stream) {
  if (n <= 0) return NB_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(nb_live_append_kernel, dim3((unsigned)blocks), dim3(256),
                     0, stream, ll, n, thr, pool, pool_n, cap, overflow);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_live_select(const double* pool, const int* pool_n, int cap,
                          int k, double* out, int* out_n, double* thr,
                          double* stats, hipStream_t stream) {
  hipLaunchKernelGGL(nb_live_select_kernel, dim3(1), dim3(LV_THREADS), 0,
                     stream, pool, pool_n, cap, k, out, out_n, thr, stats);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_live_stats(const double* ll, long long n, const double* thr,
                         double* out, hipStream_t stream) {
  hipLaunchKernelGGL(nb_live_stats_kernel, dim3(1), dim3(1024), 0, stream, ll,
                     n, thr, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
