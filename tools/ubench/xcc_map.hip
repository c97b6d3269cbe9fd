// Which XCD does workgroup b land on?  (placement is not a contract; we only use it for speed)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void k(int *out, int lds_words) {
  extern __shared__ int lds[];
  if (threadIdx.x < (unsigned)lds_words) lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf) + (lds[0] & 0);
  // keep the block alive a little so that all blocks of a 256-grid are co-resident
  for (int i = 0; i < 2000; ++i) asm volatile("s_sleep 10");
}
int main() {
  for (int grid : {256, 600, 4800}) {
    int *d; hipMalloc(&d, grid * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    k<<<grid, 512, 128 * 1024>>>(d, 64);
    std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
    int match = 0; for (int b = 0; b < grid; ++b) match += (h[b] == (b % 8));
    printf("grid %d: xcc == b%%8 for %d/%d blocks; first 24:", grid, match, grid);
    for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
    printf("\n");
    hipFree(d);
  }
}
