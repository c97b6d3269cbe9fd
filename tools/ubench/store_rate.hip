// Micro-benchmark: how fast can a CU STORE a GEMM output tile?  One 512-thread workgroup per CU writes
// fp16 [rows][ldc] tiles (320 x 256 per workgroup and iteration, as the tower's GEMM epilogue does) with
// 16-byte stores in four lane->address patterns:
//   0  32 rows x 32 B per instruction   (MFMA layout after v_permlane32_swap)
//   1  16 rows x 64 B per instruction   (LDS-staged line-assembling epilogue)
//   2   8 rows x 128 B per instruction  (full 128-byte lines)
//   3   1 KiB contiguous per instruction (ldc = 256: the tile is one contiguous block)
// Prints bytes / cycle / CU (s_memtime) and GB/s.  usage: store_rate [iters=200] [workgroups=256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned char *C, size_t ldc_bytes, int iters,
                                                    unsigned long long *cycles) {
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // workgroup tile: rows [320 * (blockIdx + gridDim * it) ...), 256 fp16 columns = 512 bytes per row
    unsigned char *tile = C + (size_t)((blockIdx.x + (size_t)gridDim.x * (it & 3)) * 320) * ldc_bytes;
    unsigned char *wave = tile + (size_t)(wr * 160) * ldc_bytes + wc * 128;   // 160 rows x 128 B per wave
#pragma unroll
    for (int k = 0; k < 20; ++k) {   // 20 x 1 KiB per wave
      size_t off;
      if (MODE == 0) {          // 32 rows x 32 B: instruction k covers rows 32 (k / 4) .. +31, byte 32 (k % 4)
        off = (size_t)(32 * (k >> 2) + (lane & 31)) * ldc_bytes + 32 * (k & 3) + 16 * (lane >> 5);
      } else if (MODE == 1) {   // 16 rows x 64 B
        off = (size_t)(16 * (k >> 1) + (lane >> 2)) * ldc_bytes + 64 * (k & 1) + 16 * (lane & 3);
      } else if (MODE == 2) {   // 8 rows x 128 B
        off = (size_t)(8 * k + (lane >> 3)) * ldc_bytes + 16 * (lane & 7);
      } else {                  // contiguous KiB
        off = (size_t)k * 1024 + lane * 16 + (size_t)(wr * 4 + wc) * 20480 - (size_t)(wr * 160) * ldc_bytes - wc * 128;
      }
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(wave + off), "v"(v) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200, wgs = argc > 2 ? atoi(argv[2]) : 256;
  const size_t ldc = 2304 * 2;   // QKV output rows
  const size_t bytes = (size_t)wgs * 4 * 320 * ldc + (1 << 20);
  unsigned char *C;
  unsigned long long *cyc;
  hipMalloc(&C, bytes);
  hipMalloc(&cyc, wgs * sizeof(unsigned long long));
  hipMemset(C, 0, bytes);
  for (int mode = 0; mode < 4; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int n) {
      switch (mode) {
        case 0: store_kernel<0><<<wgs, 512>>>(C, ldc, n, cyc); break;
        case 1: store_kernel<1><<<wgs, 512>>>(C, ldc, n, cyc); break;
        case 2: store_kernel<2><<<wgs, 512>>>(C, ldc, n, cyc); break;
        default: store_kernel<3><<<wgs, 512>>>(C, ldc, n, cyc); break;
      }
    };
    launch(10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < wgs; ++i) mean += (double)h[i] / wgs;
    const double per_wg = (double)iters * 8 * 20 * 1024;
    printf("mode %d (%s): %.1f B/cycle/CU, %.0f cycles per 160 KiB tile, %.2f TB/s over %d workgroups\n", mode,
           mode == 0 ? "32 rows x 32 B" : mode == 1 ? "16 rows x 64 B" : mode == 2 ? "8 rows x 128 B" : "1 KiB contiguous",
           per_wg / mean, mean / iters, per_wg * wgs / (ms * 1e-3) / 1e12, wgs);
  }
  return 0;
}
