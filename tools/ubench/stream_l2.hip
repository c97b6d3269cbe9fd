// Micro-benchmark: how fast can one workgroup per CU pull L2-resident data (a) into LDS by
// LDS-DMA, (b) into VGPRs by global_load_dwordx4 ?  Sets the ceiling for GEMM operand staging.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

constexpr int ROWVEC = 96;  // 768 halfs = 1536 B rows
template <int MODE, int THREADS, int INFLIGHT, int SHARE = 1>
__global__ __launch_bounds__(THREADS) void stream_kernel(const uint4 *src, size_t n_vec_mask, int iters,
                                                         uint4 *sink) {
  __shared__ __attribute__((aligned(16))) uint4 lds[INFLIGHT * THREADS];
  const int tid = threadIdx.x, wid = tid >> 6;
  // every block walks its own window of the (L2-sized) buffer
  size_t base = ((size_t)(SHARE > 1 ? ((blockIdx.x & 7) * 64 + ((blockIdx.x >> 3) / SHARE)) : blockIdx.x) * 7919u * THREADS) & n_vec_mask;
  uint4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2 || MODE == 3) {
      // GEMM-like: thread -> row (tid>>3) + 64k of a row-major [rows][ROWVEC] matrix, chunk tid&7
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) {
        const int row = (tid >> 3) + (THREADS / 8) * k;
        int ch = tid & 7;
        if (MODE == 3) ch ^= (row >> 1) & 7;
        const size_t idx = (base + (size_t)row * ROWVEC + (size_t)(it % (ROWVEC / 8)) * 8 + ch) & n_vec_mask;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + idx), (lptr_t)(lds + k * THREADS + wid * 64), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) {
        const uint4 *p = src + ((base + (size_t)k * THREADS + tid) & n_vec_mask);
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(lds + k * THREADS + wid * 64), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      uint4 v[INFLIGHT];
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) v[k] = src[(base + (size_t)k * THREADS + tid) & n_vec_mask];
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) { acc.x ^= v[k].x; acc.y ^= v[k].y; acc.z ^= v[k].z; acc.w ^= v[k].w; }
    }
    if (MODE >= 2) { if ((it % (ROWVEC / 8)) == ROWVEC / 8 - 1) base = (base + (size_t)INFLIGHT * (THREADS / 8) * ROWVEC) & n_vec_mask; }
    else base = (base + (size_t)INFLIGHT * THREADS) & n_vec_mask;
  }
  if (MODE != 1) { __syncthreads(); acc = lds[tid]; }
  if (acc.x == 0x12345678u) sink[tid] = acc;
}

template <int MODE, int THREADS, int INFLIGHT, int SHARE = 1>
void run(const char *name, const uint4 *d, size_t nvec, uint4 *sink, int blocks) {
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  stream_kernel<MODE, THREADS, INFLIGHT, SHARE><<<blocks, THREADS>>>(d, nvec - 1, 50, sink);
  hipEventRecord(a);
  stream_kernel<MODE, THREADS, INFLIGHT, SHARE><<<blocks, THREADS>>>(d, nvec - 1, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)blocks * iters * INFLIGHT * THREADS * 16.0;
  printf("%-34s blocks=%4d thr=%4d inflight=%2d KB/iter: %8.2f TB/s  (%.1f B/clk/CU @2.1GHz)\n", name, blocks, THREADS,
         INFLIGHT * THREADS * 16 / 1024, bytes / ms / 1e9, bytes / ms / 1e9 * 1e12 / 256 / 2.1e9 / 1e0 / 1e0 * 1e-0 / 1.0 / 1.0 * 1.0 / 1.0 / 1.0 * 1e-0 / 1e0 / 1.0);
}

int main() {
  for (size_t mb : {2, 32, 512}) {
    size_t bytes = mb << 20, nvec = bytes / 16;
    uint4 *d, *sink; hipMalloc(&d, bytes); hipMalloc(&sink, 1 << 16);
    hipMemset(d, 1, bytes);
    printf("== buffer %zu MiB\n", mb);
    run<0, 512, 6>("glds 512thr x6 (48KB in flight)", d, nvec, sink, 256);
    run<0, 512, 12>("glds 512thr x12 (96KB in flight)", d, nvec, sink, 256);
    run<0, 256, 8>("glds 256thr x8, 2 blocks/CU", d, nvec, sink, 512);
    run<0, 64, 16>("glds 1 wave x16", d, nvec, sink, 256);
    run<2, 512, 6>("glds rows stride1536 noswz x6", d, nvec, sink, 256);
    run<3, 512, 6>("glds rows stride1536 xorswz x6", d, nvec, sink, 256);
    run<3, 512, 6, 4>("glds rows, 4 blocks/XCD share", d, nvec, sink, 256);
    run<3, 512, 6, 8>("glds rows, 8 blocks/XCD share", d, nvec, sink, 256);
    run<3, 512, 6, 32>("glds rows, 32 blocks/XCD share", d, nvec, sink, 256);
    run<1, 512, 6>("vgpr 512thr x6", d, nvec, sink, 256);
    run<1, 512, 12>("vgpr 512thr x12", d, nvec, sink, 256);
    run<1, 256, 8>("vgpr 256thr x8, 2 blocks/CU", d, nvec, sink, 512);
    hipFree(d); hipFree(sink);
  }
  return 0;
}
