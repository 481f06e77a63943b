// Proposal draw, stream compaction, shell statistics and small helpers.
#include "nb_common.h"

#include "nb_draw.h"

namespace {

// ---------------------------------------------------------------------------
// Proposal draw: 64 proposals per workgroup of four wavefronts; lane = proposal
// in every wavefront, the wavefronts split the work of a proposal:
//   union.py:308-312  member ~ softmax(log_v_all)  (inverse CDF per proposal,
//                     same distribution as multinomial + shuffle)
//   basic.py:376-381  z ~ N(0,I); z /= |z|; z *= u^(1/D); x = B z + c
//   basic.py:85, 633-640  cube columns ~ U[0,1)
// Wavefront w draws the quadruples of normals q = w, w+4, ... -- one Philox
// call feeds TWO Box-Muller pairs (four 32-bit uniforms (w + 1/2) / 2^32: the
// ten Philox rounds are quarter-rate integer multiplies and cost more than
// the fp64 log / sqrt / sincospi of a pair; with a 53-bit uniform per word
// pair they were half of the kernel) --, then rows r = w, w+4, ... of the
// triangular product.  The row index is wave uniform, so for a
// single-member bound the B operands arrive through scalar loads.  z lives in
// LDS as z[slot][proposal]; x = B z + c overwrites it four rows at a time from
// the last row up (row r only needs z_0..z_r), and the 64 x D block is then
// written with fully coalesced stores.  Splitting a proposal over wavefronts
// instead of giving each wavefront its own 64 proposals keeps the LDS
// footprint per wavefront at a quarter: 24 wavefronts per CU instead of 6
// hide the latency of the dependent fp64 chains.
// ---------------------------------------------------------------------------
constexpr int ZS = 65;     // LDS row stride (odd: transposed reads conflict-free)
constexpr int DW = 4;      // wavefronts per workgroup
constexpr int RB = 8;      // terms of a row product per trip

// sum_{j <= r} B[r][j] z_j for the proposal of this lane, in the fixed order
// j = 0..r, RB terms per trip; the last trip is padded with zero weights
// (reads stay inside the row-packed B, which the host pads by RB doubles, and
// inside the z slots 0..r).  With a wave-uniform B the operands are scalar
// loads.
__device__ __forceinline__ double draw_row(const double* __restrict__ B,
                                           const double* zs, int r, int lane) {
  const double* brow = B + (long long)r * (r + 1) / 2;
  double acc = 0.0;
  for (int j = 0; j <= r; j += RB) {
    double bv[RB], zv[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const int jj = j + u;
      const double b = brow[jj];           // unconditional: B is padded
      bv[u] = (jj <= r) ? b : 0.0;
      zv[u] = zs[(jj <= r ? jj : r) * ZS + lane];
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) acc += bv[u] * zv[u];
  }
  return acc;
}

template <int DT>
__global__ void __launch_bounds__(64 * DW)
nb_draw_kernel(const double* __restrict__ blob, unsigned long long seed,
               unsigned long long offset, long long n,
               double* __restrict__ x_out) {
  extern __shared__ __attribute__((aligned(16))) double zs[];   // [slot][ZS]
  __shared__ int member_of[64];
  __shared__ double norm_part[DW][64];
  __shared__ double radius_of[64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long i0 = (long long)blockIdx.x * 64;
  const long long i = i0 + lane;
  const unsigned long long g = offset + (unsigned long long)i;

  const int n_dim = (int)nb_hdr(blob, NB_H_NDIM);
  const int dp = 16 * (int)nb_hdr(blob, NB_H_DT);
  const int K = (int)nb_hdr(blob, NB_H_K);
  const double* cdf = blob + nb_hdr(blob, NB_H_OFF_CDF);
  const double* draw = blob + nb_hdr(blob, NB_H_OFF_DRAW);
  const long long draw_stride = nb_hdr(blob, NB_H_DRAW_STRIDE);

  // every wavefront needs the member of its proposals
  int m = 0;
  if (K > 1) {
    double u_member, u_accept;
    nb_uniform_pair(seed, g, 0u, NB_TAG_CTRL, u_member, u_accept);
    for (int j = 0; j < K; ++j) m += (cdf[j] <= u_member) ? 1 : 0;
    if (m > K - 1) m = K - 1;
  }
  if (wave == 0) member_of[lane] = m;

  // K == 1: the member block is wave uniform (scalar loads feed the FMAs)
  const double* blk = (K == 1) ? draw : draw + m * draw_stride;
  const long long* iblk = (const long long*)blk;
  const int ne = (int)iblk[0];
  const int nc = (int)iblk[1];
  const double* c = blk + 2 + 3 * dp;
  const double* B = c + dp;

  if (wave == DW - 1) {
    // radius: the last wavefront has the fewest normals to draw
    double u_radius, u_spare;
    nb_uniform_pair(seed, g, 1u, NB_TAG_CTRL, u_radius, u_spare);
    radius_of[lane] = (ne > 0) ? pow(u_radius, 1.0 / (double)ne) : 0.0;
  }

  // normals: slots 0..ne-1 (basic.py:376-381); block q of the normal stream
  // gives the normals 4 q .. 4 q + 3
  double part = 0.0;
  for (int q = wave; 4 * q < ne; q += DW) {
    const nb_u4 w = nb_philox((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)q,
                              NB_TAG_NORMAL, (uint32_t)seed,
                              (uint32_t)(seed >> 32));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = 2 * q + h;                   // Box-Muller pair
      if (2 * j < ne) {
        double z0, z1;
        draw_normal_pair(h == 0 ? w.x : w.z, h == 0 ? w.y : w.w, z0, z1);
        zs[(2 * j) * ZS + lane] = z0;
        part += z0 * z0;
        if (2 * j + 1 < ne) {
          zs[(2 * j + 1) * ZS + lane] = z1;
          part += z1 * z1;
        }
      }
    }
  }
  norm_part[wave][lane] = part;
  // cube part: slots ne..ne+nc-1 (basic.py:85, 633-640)
  for (int j = wave; 2 * j < nc; j += DW) {
    double u0, u1;
    nb_uniform_pair(seed, g, (unsigned)j, NB_TAG_CUBE, u0, u1);
    zs[(ne + 2 * j) * ZS + lane] = u0;
    if (2 * j + 1 < nc) zs[(ne + 2 * j + 1) * ZS + lane] = u1;
  }
  __syncthreads();

  // |z| -> u^(1/D): every wavefront rescales the slots it wrote
  double norm2 = norm_part[0][lane];
#pragma unroll
  for (int w = 1; w < DW; ++w) norm2 += norm_part[w][lane];
  const double scale = radius_of[lane] / sqrt(norm2);
  for (int j = wave; 2 * j < ne; j += DW) {
    zs[(2 * j) * ZS + lane] *= scale;
    if (2 * j + 1 < ne) zs[(2 * j + 1) * ZS + lane] *= scale;
  }
  __syncthreads();

  if (K == 1) {
    // One member: B is wave uniform.  Every wavefront takes the whole z of
    // its 64 proposals into registers (one LDS read per slot instead of one
    // per term of every row) and computes the rows r = 4 i + wave with scalar
    // B operands; same accumulation order as draw_row (j ascending, the up to
    // three terms beyond the diagonal carry a zero weight; B is padded).
    double z[16 * DT];
#pragma unroll
    for (int j = 0; j < 16 * DT; ++j) z[j] = (j < ne) ? zs[j * ZS + lane] : 0.0;
    __syncthreads();                    // all of z is read before it is replaced
    const double* Bp = draw + 2 + 4 * dp;
#pragma unroll
    for (int i = 0; i < 4 * DT; ++i) {
      const int r = 4 * i + wave;
      if (r < ne) {
        const double* brow = Bp + (long long)r * (r + 1) / 2;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 4 * i + 4; ++j) {
          const double b = brow[j];
          acc += ((j <= r) ? b : 0.0) * z[j];
        }
        zs[r * ZS + lane] = acc + c[r];
      }
    }
    __syncthreads();
  } else {
  // x = B z + c in place, four rows per step from the last row up; the rows
  // of a step are read completely before any of them is overwritten
  int ne_max = ne;
  if (K > 1) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int other = __shfl_xor(ne_max, o);
      ne_max = other > ne_max ? other : ne_max;
    }
  }
  ne_max = __builtin_amdgcn_readfirstlane(ne_max);
  for (int r0 = ((ne_max + DW - 1) / DW - 1) * DW; r0 >= 0; r0 -= DW) {
    const int r = r0 + wave;
    double acc = 0.0;
    if (r < ne) {
      acc = (K == 1) ? draw_row(draw + 2 + 4 * dp, zs, r, lane)
                     : draw_row(B, zs, r, lane);
      acc += c[r];
    }
    __syncthreads();
    if (r < ne) zs[r * ZS + lane] = acc;
  }
  __syncthreads();
  }

  // coalesced store of the workgroup's contiguous 64 x D block; every row
  // maps its columns to slots through its member's table
  long long rows = n - i0;
  if (rows > 64) rows = 64;
  const int total = (int)rows * n_dim;
  const int nt = 64 * DW;
  int row = (int)threadIdx.x / n_dim, col = (int)threadIdx.x - row * n_dim;
  const int drow = nt / n_dim, dcol = nt - drow * n_dim;
  double* dst = x_out + i0 * n_dim;
  for (int e = threadIdx.x; e < total; e += nt) {
    const long long* sl = (const long long*)(draw + member_of[row] *
                                             draw_stride) + 2 + 2 * dp;
    dst[e] = zs[(int)sl[col] * ZS + row];
    row += drow; col += dcol;
    if (col >= n_dim) { col -= n_dim; ++row; }
  }
}

// ---------------------------------------------------------------------------
// Stable stream compaction, three small kernels:
//   count  : per chunk of CHUNK flags, number of rows with bit0 / with mask
//   scan   : exclusive scan of the chunk counts (single workgroup)
//   scatter: wavefront ballot + popcount prefix inside the chunk, survivors
//            copied row-wise (coalesced) in input order
// ---------------------------------------------------------------------------
constexpr int CHUNK = 2048;   // flags per workgroup (256 threads x 8)

__global__ void __launch_bounds__(256)
nb_count_kernel(const unsigned char* __restrict__ flags, unsigned char mask,
                unsigned char flip, long long n,
                long long* __restrict__ chunk_counts) {
  __shared__ int s0[4], s1[4];
  const long long base = (long long)blockIdx.x * CHUNK;
  int c0 = 0, c1 = 0;
  for (int k = 0; k < CHUNK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    const unsigned char f = (i < n) ? (flags[i] ^ flip) : 0;
    c0 += __popcll(__ballot((f & 1) != 0));
    c1 += __popcll(__ballot((f & mask) != 0));
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s0[wave] = c0; s1[wave] = c1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    chunk_counts[2 * blockIdx.x] = s0[0] + s0[1] + s0[2] + s0[3];
    chunk_counts[2 * blockIdx.x + 1] = s1[0] + s1[1] + s1[2] + s1[3];
  }
}

__global__ void __launch_bounds__(256)
nb_scan_kernel(long long* __restrict__ chunk_counts, long long n_chunks,
               long long* __restrict__ totals) {
  // chunk_counts[2c+1] -> exclusive offset (in place); totals = {sum0, sum1}
  __shared__ long long wsum0[4], wsum1[4];
  __shared__ long long carry0, carry1;
  if (threadIdx.x == 0) { carry0 = 0; carry1 = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long long base = 0; base < n_chunks; base += 256) {
    const long long c = base + threadIdx.x;
    long long v0 = (c < n_chunks) ? chunk_counts[2 * c] : 0;
    long long v1 = (c < n_chunks) ? chunk_counts[2 * c + 1] : 0;
    long long i0 = v0, i1 = v1;
    for (int d = 1; d < 64; d <<= 1) {       // inclusive wave scan
      const long long t0 = __shfl_up(i0, d), t1 = __shfl_up(i1, d);
      if (lane >= d) { i0 += t0; i1 += t1; }
    }
    if (lane == 63) { wsum0[wave] = i0; wsum1[wave] = i1; }
    __syncthreads();
    long long w0 = 0, w1 = 0;
    for (int w = 0; w < wave; ++w) { w0 += wsum0[w]; w1 += wsum1[w]; }
    const long long ex1 = carry1 + w1 + i1 - v1;
    if (c < n_chunks) chunk_counts[2 * c + 1] = ex1;
    __syncthreads();
    if (threadIdx.x == 255) { carry0 += w0 + i0; carry1 += w1 + i1; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { totals[0] = carry0; totals[1] = carry1; }
}

__global__ void __launch_bounds__(256)
nb_scatter_kernel(const double* __restrict__ x,
                  const unsigned char* __restrict__ flags, unsigned char mask,
                  unsigned char flip, long long n, int n_dim,
                  const long long* __restrict__ chunk_counts,
                  double* __restrict__ out, long long* __restrict__ src_idx) {
  // Per round of 256 flags: the survivors' rows (chunk-relative) are listed
  // in LDS in order, then ALL threads copy the elements of those rows --
  // thread t the elements t, t + 256, ... of the contiguous destination
  // block, every load independent of the others.  (One row at a time per
  // wavefront -- a dependent load / store pair per survivor -- took 171 us
  // for the 65 536 survivors of a shell-exclusion batch at n_dim 50: 150 GB/s.)
  __shared__ int wcount[4];
  __shared__ int srcs[256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.x * CHUNK;
  long long dst = chunk_counts[2 * blockIdx.x + 1];
  const int r0 = (int)threadIdx.x / n_dim, c0 = (int)threadIdx.x - r0 * n_dim;
  const int dr = 256 / n_dim, dc = 256 - dr * n_dim;
  for (int k = 0; k < CHUNK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    const bool keep = (i < n) && (((flags[i] ^ flip) & mask) != 0);
    const unsigned long long b = __ballot(keep);
    if (lane == 0) wcount[wave] = __popcll(b);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wcount[w];
    const int total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    if (keep) {
      const int at = woff + __popcll(b & ((1ull << lane) - 1ull));
      srcs[at] = k * 256 + (int)threadIdx.x;
      if (src_idx != nullptr) src_idx[dst + at] = i;
    }
    __syncthreads();
    const double* src = x + base * n_dim;
    double* drow = out + dst * n_dim;
    int r = r0, c = c0;
    const int n_el = total * n_dim;
#pragma unroll 4
    for (int e = threadIdx.x; e < n_el; e += 256) {
      drow[e] = src[(long long)srcs[r] * n_dim + c];
      r += dr;
      c += dc;
      if (c >= n_dim) { c -= n_dim; ++r; }
    }
    __syncthreads();
    dst += total;
  }
}

// ---------------------------------------------------------------------------
// Shell statistics (sampler.py:934-937, 1144): streaming (max, sum exp,
// sum exp^2) with wavefront-shuffle merges.
// ---------------------------------------------------------------------------
struct Lse { double m, s1, s2, cnt; };

__device__ inline Lse lse_merge(Lse a, Lse b) {
  Lse o;
  o.m = fmax(a.m, b.m);
  o.cnt = a.cnt + b.cnt;
  if (o.m == -INFINITY) { o.s1 = 0.0; o.s2 = 0.0; return o; }
  const double ea = (a.m == -INFINITY) ? 0.0 : exp(a.m - o.m);
  const double eb = (b.m == -INFINITY) ? 0.0 : exp(b.m - o.m);
  o.s1 = a.s1 * ea + b.s1 * eb;
  o.s2 = a.s2 * ea * ea + b.s2 * eb * eb;
  return o;
}

__device__ inline Lse lse_wave(Lse v) {
  for (int d = 32; d >= 1; d >>= 1) {
    Lse o;
    o.m = __shfl_xor(v.m, d);
    o.s1 = __shfl_xor(v.s1, d);
    o.s2 = __shfl_xor(v.s2, d);
    o.cnt = __shfl_xor(v.cnt, d);
    v = lse_merge(v, o);
  }
  return v;
}

__global__ void __launch_bounds__(256)
nb_lse_partial_kernel(const double* __restrict__ log_l, long long n,
                      double threshold, double* __restrict__ partial) {
  __shared__ Lse sh[4];
  Lse acc = {-INFINITY, 0.0, 0.0, 0.0};
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (long long)gridDim.x * 256) {
    const double v = log_l[i];
    Lse e = {v, (v == -INFINITY) ? 0.0 : 1.0, (v == -INFINITY) ? 0.0 : 1.0,
             (v >= threshold) ? 1.0 : 0.0};
    acc = lse_merge(acc, e);
  }
  acc = lse_wave(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    Lse t = lse_merge(lse_merge(sh[0], sh[1]), lse_merge(sh[2], sh[3]));
    partial[4 * blockIdx.x + 0] = t.m;
    partial[4 * blockIdx.x + 1] = t.s1;
    partial[4 * blockIdx.x + 2] = t.s2;
    partial[4 * blockIdx.x + 3] = t.cnt;
  }
}

__global__ void __launch_bounds__(64)
nb_lse_final_kernel(const double* __restrict__ partial, int n_part,
                    double* __restrict__ out) {
  Lse acc = {-INFINITY, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < n_part; i += 64) {
    Lse e = {partial[4 * i], partial[4 * i + 1], partial[4 * i + 2],
             partial[4 * i + 3]};
    acc = lse_merge(acc, e);
  }
  acc = lse_wave(acc);
  if (threadIdx.x == 0) {
    out[0] = (acc.m == -INFINITY) ? -INFINITY : acc.m + log(acc.s1);
    out[1] = (acc.m == -INFINITY) ? -INFINITY : 2.0 * acc.m + log(acc.s2);
    out[2] = acc.m;
    out[3] = acc.cnt;
  }
}

__global__ void nb_philox_kernel(unsigned long long seed,
                                 unsigned long long offset, unsigned block,
                                 unsigned tag, long long n, double* u) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double u0, u1;
  nb_uniform_pair(seed, offset + (unsigned long long)i, block, tag, u0, u1);
  u[2 * i] = u0;
  u[2 * i + 1] = u1;
}

// back-to-back fp64 MFMA issue-rate probe (4 independent accumulators).  The
// register budget of a 1024-thread block keeps the accumulators in ordinary
// VGPRs; with a larger budget the compiler moves them to AGPRs and copies all
// of them in and out on every loop trip, which measures the copies (47
// instead of 77 TFLOP/s).
__global__ void __launch_bounds__(1024)
nb_mfma_peak_kernel(int iters, double* sink) {
  nb_d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a3, 0, 0, 0);
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.678)
    sink[0] = a0[0];
}

}  // namespace

int nb_launch_draw(const double* blob_dev, int n_dim, unsigned long long seed,
                   unsigned long long offset, long long n, double* x_out,
                   hipStream_t stream) {
  if (n <= 0) return NB_OK;
  const long long blocks = (n + 63) / 64;
  const size_t lds = (size_t)n_dim * 65 * sizeof(double);
#define NB_DRAW_CASE(D_T)                                                     \
  case D_T:                                                                  \
    hipLaunchKernelGGL(nb_draw_kernel<D_T>, dim3((unsigned)blocks),          \
                       dim3(64 * DW), lds, stream, blob_dev, seed, offset, n, \
                       x_out);                                               \
    break;
  switch ((n_dim + 15) / 16) {
    NB_DRAW_CASE(1) NB_DRAW_CASE(2) NB_DRAW_CASE(3) NB_DRAW_CASE(4)
    NB_DRAW_CASE(5) NB_DRAW_CASE(6) NB_DRAW_CASE(7) NB_DRAW_CASE(8)
    default:
      nb_set_error("n_dim > 128 is not supported by the device kernels");
      return NB_ERR_UNSUPPORTED;
  }
#undef NB_DRAW_CASE
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

long long nb_compact_chunks(long long n) { return (n + CHUNK - 1) / CHUNK; }

int nb_launch_compact(const double* x, const unsigned char* flags,
                      unsigned char mask, unsigned char flip, long long n,
                      int n_dim, double* out,
                      long long* src_idx, long long* counts,
                      long long* chunk_counts, hipStream_t stream) {
  const long long n_chunks = nb_compact_chunks(n);
  if (n_chunks == 0) {
    NB_HIP_CHECK(hipMemsetAsync(counts, 0, 2 * sizeof(long long), stream));
    return NB_OK;
  }
  hipLaunchKernelGGL(nb_count_kernel, dim3((unsigned)n_chunks), dim3(256), 0,
                     stream, flags, mask, flip, n, chunk_counts);
  hipLaunchKernelGGL(nb_scan_kernel, dim3(1), dim3(256), 0, stream,
                     chunk_counts, n_chunks, counts);
  if (out != nullptr)
    hipLaunchKernelGGL(nb_scatter_kernel, dim3((unsigned)n_chunks), dim3(256),
                       0, stream, x, flags, mask, flip, n, n_dim, chunk_counts,
                       out,
                       src_idx);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_lse_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

int nb_launch_shell_stats(const double* log_l, long long n, double threshold,
                          double* out, double* partial, hipStream_t stream) {
  const int blocks = nb_lse_blocks(n);
  hipLaunchKernelGGL(nb_lse_partial_kernel, dim3(blocks), dim3(256), 0,
                     stream, log_l, n, threshold, partial);
  hipLaunchKernelGGL(nb_lse_final_kernel, dim3(1), dim3(64), 0, stream,
                     partial, blocks, out);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_launch_philox(unsigned long long seed, unsigned long long offset,
                     unsigned block, unsigned tag, long long n, double* u,
                     hipStream_t stream) {
  if (n <= 0) return NB_OK;
  hipLaunchKernelGGL(nb_philox_kernel, dim3((unsigned)((n + 255) / 256)),
                     dim3(256), 0, stream, seed, offset, block, tag, n, u);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}

int nb_run_mfma_peak(int iters, double* tflops) {
  double* sink = nullptr;
  NB_HIP_CHECK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  NB_HIP_CHECK(hipEventCreate(&e0));
  NB_HIP_CHECK(hipEventCreate(&e1));
  const int blocks = 256 * 4;   // 4 workgroups of 4 waves per CU
  // warm-up long enough for the clocks to leave their idle state (a short
  // probe right after an idle period reads ~48 instead of ~77 TFLOP/s)
  hipLaunchKernelGGL(nb_mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0,
                     iters > 20000 ? iters : 20000, sink);
  NB_HIP_CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(nb_mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, iters,
                     sink);
  NB_HIP_CHECK(hipEventRecord(e1, 0));
  NB_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  NB_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * iters * 4.0 * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return NB_OK;
}

// ---------------------------------------------------------------------------
// PhaseShift.transform (bounds/periodic.py:50-72), in place.  numpy's
// ``v % 1`` for doubles is fmod(v, 1), plus 1 if that is negative; for the
// values that occur (|v| < 2) this equals v - floor(v) bit for bit.
// ---------------------------------------------------------------------------
namespace {
struct ShiftArgs {
  double s[16 * NB_MAX_DT];
  unsigned char on[16 * NB_MAX_DT];
};

__global__ void __launch_bounds__(256)
nb_phase_shift_kernel(double* __restrict__ x, long long total, int n_dim,
                      ShiftArgs a, int inverse) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
       e < total; e += stride) {
    const int col = (int)(e % n_dim);
    if (a.on[col]) {
      const double v = inverse ? x[e] - a.s[col] : x[e] + a.s[col];
      x[e] = v - floor(v);
    }
  }
}
}  // namespace

int nb_launch_phase_shift(double* x, long long n, int n_dim, const double* s,
                          const unsigned char* on, int inverse,
                          hipStream_t stream) {
  ShiftArgs a;
  for (int i = 0; i < 16 * NB_MAX_DT; ++i) { a.s[i] = s[i]; a.on[i] = on[i]; }
  const long long total = n * n_dim;
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(nb_phase_shift_kernel, dim3((unsigned)blocks), dim3(256),
                     0, stream, x, total, n_dim, a, inverse);
  NB_HIP_CHECK(hipGetLastError());
  return NB_OK;
}
