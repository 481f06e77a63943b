// Does v_mfma_f64_4x4x4_4b with CBSZ = 2 / ABID = s (block s of the A operand
// broadcast to all four blocks) reproduce register s of v_mfma_f64_16x16x4 bit
// for bit?  (A tile operand loaded once, the 4-unit sub-tiles of its 16
// outputs issued one by one: 16 cycles each instead of 64 for the tile.)
//   hipcc --offload-arch=gfx950 -O2 profiles/tools/mfma_subtile.hip -o /tmp/mfma_subtile && /tmp/mfma_subtile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));

// sub-tile S of the tile operand: outputs 4 S .. 4 S + 3, replicated over the
// four blocks (= groups of four points) by the load
template <int S>
__device__ double chain4(const double* A, const double* b, int n) {
  const int lane = threadIdx.x;
  double c0 = 0.0, c1 = 0.0;
  for (int k = 0; k + 1 < n; k += 2) {
    const double a0 = A[k * 64 + (lane >> 4) * 16 + 4 * S + (lane & 3)];
    const double a1 = A[(k + 1) * 64 + (lane >> 4) * 16 + 4 * S + (lane & 3)];
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b[k], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b[k + 1], c1, 0, 0, 0);
  }
  return c0 + c1;
}

__global__ void k(const double* A, const double* B, double* out16, double* out4,
                  long long* ticks) {
  const int lane = threadIdx.x;
  double a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = A[i * 64 + lane]; b[i] = B[i * 64 + lane]; }
  d4 c0 = {0, 0, 0, 0}, c1 = c0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 8; i += 2) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i + 1], b[i + 1], c1, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) c0[r] += c1[r];
  asm volatile("" ::"v"(c0));
  long long t1 = __builtin_amdgcn_s_memtime();
  double s0 = chain4<0>(A, b, 8);
  asm volatile("" ::"v"(s0));
  long long t2 = __builtin_amdgcn_s_memtime();
  double s1 = chain4<1>(A, b, 8), s2 = chain4<2>(A, b, 8), s3 = chain4<3>(A, b, 8);
  for (int r = 0; r < 4; ++r) out16[r * 64 + lane] = c0[r];
  out4[0 * 64 + lane] = s0; out4[1 * 64 + lane] = s1;
  out4[2 * 64 + lane] = s2; out4[3 * 64 + lane] = s3;
  if (lane == 0) { ticks[0] = t1 - t0; ticks[1] = t2 - t1; }
}

int main() {
  double hA[512], hB[512], h16[256], h4[256];
  srand(1);
  for (int i = 0; i < 512; ++i) {
    hA[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0;
    hB[i] = (rand() / (double)RAND_MAX - 0.5) * 3.0;
  }
  double *A, *B, *o16, *o4; long long* t;
  hipMalloc(&A, sizeof hA); hipMalloc(&B, sizeof hB);
  hipMalloc(&o16, sizeof h16); hipMalloc(&o4, sizeof h4); hipMalloc(&t, 16);
  hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, o16, o4, t);
  hipMemcpy(h16, o16, sizeof h16, hipMemcpyDeviceToHost);
  hipMemcpy(h4, o4, sizeof h4, hipMemcpyDeviceToHost);
  long long ht[2]; hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
  // host models: lane = x + 4 blk + 16 y
  auto A_ = [&](int ks, int l) { return hA[ks * 64 + l]; };
  auto B_ = [&](int ks, int l) { return hB[ks * 64 + l]; };
  for (int model = 0; model < 3; ++model) {
    // 0: broadcast of block s (cbsz 2, abid s); 1: no broadcast (own block);
    // 2: explicit check of the 16x16x4 tile against the host
    for (int s = 0; s < 4; ++s) {
    double worst = 0; int exact = 0;
      for (int l = 0; l < 64; ++l) {
        const int j = l & 3, blk = (l >> 2) & 3, i = l >> 4;
        double e = 0, o = 0;
        for (int ks = 0; ks < 8; ++ks) {
          double acc = 0;
          for (int kk = 0; kk < 4; ++kk) {
            double av, bv;
            if (model == 0) { av = A_(ks, i + 4 * s + 16 * kk); bv = B_(ks, j + 4 * blk + 16 * kk); }
            else if (model == 1) { av = A_(ks, i + 4 * blk + 16 * kk); bv = B_(ks, j + 4 * blk + 16 * kk); }
            else { av = A_(ks, (4 * s + i) + 16 * kk); bv = B_(ks, (l & 15) + 16 * kk); }
            acc = __builtin_fma(av, bv, acc);
          }
          if (ks & 1) o += acc; else e += acc;
        }
        const double got = (model == 2) ? h16[s * 64 + l] : h4[s * 64 + l];
        double d = got - (e + o); if (d < 0) d = -d;
        if (d > worst) worst = d;
        if (d == 0) ++exact;
      }
    printf("model %d s %d: largest deviation %g, %d of 64 exact\n", model, s, worst, exact);
    }
  }
  int bad = 0; double worst = 0;
  for (int i = 0; i < 256; ++i) {
    if (memcmp(&h16[i], &h4[i], 8)) { ++bad; double d = h16[i] - h4[i]; if (d < 0) d = -d; if (d > worst) worst = d; }
  }
  printf("16x16x4 chain of 8: %lld ticks; 4x4x4_4b (cbsz 2) chain of 8: %lld ticks\n", ht[0], ht[1]);
  printf("sub-tile s == register s of the tile: %d of 256 values differ (largest %g)\n", bad, worst);
  return 0;
}
