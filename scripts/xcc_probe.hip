// Development probe: which XCD (XCC) does a workgroup run on?  Prints the histogram of HW_REG_XCC_ID against blockIdx.x % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned *out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
  const int n = 4096;
  unsigned *d, h[n];
  hipMalloc(&d, n * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[16][8] = {{0}};
  unsigned ormask = 0;
  for (int i = 0; i < n; i++) { hist[h[i] & 15][i % 8]++; ormask |= h[i]; }
  printf("raw OR of register values: 0x%x\n", ormask);
  for (int x = 0; x < 16; x++) {
    int tot = 0; for (int m = 0; m < 8; m++) tot += hist[x][m];
    if (!tot) continue;
    printf("xcc_id(low 4 bits) %2d: ", x);
    for (int m = 0; m < 8; m++) printf("%5d", hist[x][m]);
    printf("\n");
  }
  return 0;
}
