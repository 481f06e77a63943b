#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  if (threadIdx.x == 0) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x] = (int)v;
  }
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    const int n = 136;
    k<<<n, 256>>>(d);
    int h[4096]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if ((h[i] & 15) != (h[i % 8] & 15)) ++bad;
    printf("rep %d: raw first 16:", rep);
    for (int i = 0; i < 16; ++i) printf(" %x", h[i]);
    printf("  | WGs whose xcc differs from WG (i mod 8): %d\n", bad);
  }
  return 0;
}
