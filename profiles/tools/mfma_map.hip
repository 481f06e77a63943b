// Lane maps of v_mfma_f64_4x4x4_4b found with one-hot operands (CBSZ / ABID as
// template arguments).  hipcc --offload-arch=gfx950 -O2 ... && run
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CBSZ, int ABID>
__global__ void k(int la, int lb, double* out) {
  const int lane = threadIdx.x;
  double a = (la < 0) ? 1.0 : (lane == la ? 1.0 : 0.0);
  double b = (lb < 0) ? 1.0 : (lane == lb ? 1.0 : 0.0);
  double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
  out[lane] = c;
}
template <int CBSZ, int ABID>
void run(const char* name) {
  double* o; hipMalloc(&o, 512); double h[64];
  printf("%s\n", name);
  for (int which = 0; which < 2; ++which) {
    for (int l = 0; l < 64; ++l) {
      if (which == 0) k<CBSZ, ABID><<<1, 64>>>(l, -1, o); else k<CBSZ, ABID><<<1, 64>>>(-1, l, o);
      hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
      printf(" %s lane %2d -> D lanes:", which ? "B" : "A", l);
      for (int d = 0; d < 64; ++d) if (h[d] != 0.0) printf(" %d", d);
      printf("\n");
    }
  }
}
int main() { run<0, 0>("cbsz 0"); run<2, 1>("cbsz 2 abid 1"); return 0; }
