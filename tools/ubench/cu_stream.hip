// Micro-benchmark: what ONE CU can stream through its vector-memory path, and what that depends on.
//
// Every wave issues 64-lane x 16-byte instructions back to back, WINDOW at a time (s_waitcnt vmcnt(0) between windows):
//   op      0 = LDS-DMA load (global_load_lds_dwordx4), 1 = load to VGPRs (global_load_dwordx4), 2 = store (global_store_dwordx4)
//   source  "L2":  every wave re-reads / re-writes its own 16 KiB (the data stays in the XCD's L2)
//           "HBM": every wave walks fresh memory (2 GiB buffer: each byte touched once per launch)
//   waves per CU 4 (one per SIMD: csrc/bottleneck_fused.hip) or 8 (the GEMMs), workgroups 32 (1/8 of the chip: nothing shared is
//   saturated) or 256 (every CU).
// Prints GB/s per CU and shader clocks per instruction and CU.  If a CU's rate from HBM is the same at 32 as at 256 workgroups,
// grows with the window and is several times higher out of L2, it is bounded by (bytes a CU can have in flight) / (latency),
// not by the chip's memory system: the "vector-memory wall" of DESIGN.md 5.2 / 5.6.
// usage: cu_stream [iters=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int OP, int THREADS, int WINDOW, bool HBM>
__global__ __launch_bounds__(THREADS) void stream_kernel(unsigned char *buf, size_t wave_span, int iters, unsigned long long *clocks,
                                                         u32x4 *sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[THREADS * 16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const size_t wave_id = (size_t)blockIdx.x * (THREADS / 64) + wid;
  unsigned char *base = buf + wave_id * wave_span + lane * 16;
  const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)lds + wid * 1024);
  const u32x4 val = {(unsigned)tid, 1u, 2u, 3u};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    // HBM: window `it` of this wave's span; L2: the same 16 KiB again and again
    unsigned char *p = base + (HBM ? (size_t)it * WINDOW * 1024 : 0);
    u32x4 got[OP == 1 ? WINDOW : 1];   // (destinations stay allocated until the wait: a register reused while its load flies is clobbered)
#pragma unroll
    for (int k = 0; k < WINDOW; ++k) {
      unsigned char *q = p + (HBM ? k * 1024 : ((k & 15) * 1024));
      if (OP == 0) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(q), "s"(lds_dst) : "memory");
      } else if (OP == 1) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(got[k]) : "v"(q) : "memory");
      } else {
        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(q), "v"(val) : "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (OP == 1) {
#pragma unroll
      for (int k = 0; k < WINDOW; ++k) asm volatile("" ::"v"(got[k]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) clocks[wave_id] = t1 - t0;
  if (t1 == 0x12345678u) sink[0] = val;
}

template <int OP, int THREADS, int WINDOW, bool HBM>
void run(const char *name, unsigned char *buf, size_t bytes, int iters, int blocks, unsigned long long *clocks, u32x4 *sink) {
  const int waves = THREADS / 64;
  const size_t span = HBM ? (size_t)iters * WINDOW * 1024 : 16384;
  if ((size_t)blocks * waves * span > bytes) { printf("%-44s skipped (buffer too small)\n", name); return; }
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  stream_kernel<OP, THREADS, WINDOW, HBM><<<blocks, THREADS>>>(buf, span, 2, clocks, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  stream_kernel<OP, THREADS, WINDOW, HBM><<<blocks, THREADS>>>(buf, span, iters, clocks, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  unsigned long long *h = (unsigned long long *)malloc((size_t)blocks * waves * 8);
  hipMemcpy(h, clocks, (size_t)blocks * waves * 8, hipMemcpyDeviceToHost);
  double cmax = 0;
  for (int i = 0; i < blocks * waves; ++i) cmax = h[i] > cmax ? (double)h[i] : cmax;
  free(h);
  const double instr_per_cu = (double)waves * iters * WINDOW;
  const double gbs_cu = instr_per_cu * 1024.0 / (ms * 1e-3) / 1e9;
  printf("%-44s wgs=%3d waves/CU=%d window=%2d: %7.1f GB/s per CU  %6.1f clocks per instruction and CU  (%.2f TB/s in all)\n", name, blocks,
         waves, WINDOW, gbs_cu, cmax / instr_per_cu, gbs_cu * blocks / 1e3);
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 64;
  const int only = argc > 2 ? atoi(argv[2]) : -1;   // one op only
  setvbuf(stdout, nullptr, _IONBF, 0);
  const size_t bytes = (size_t)2 << 30;
  unsigned char *buf;
  unsigned long long *clocks;
  u32x4 *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMalloc(&clocks, 256 * 8 * 8);
  hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes);
#define ROW(OP, NAME)                                                                                       \
  for (int blocks : {32, 256}) {                                                                            \
    run<OP, 256, 8, true>(NAME " from / to HBM", buf, bytes, iters, blocks, clocks, sink);                  \
    run<OP, 256, 16, true>(NAME " from / to HBM", buf, bytes, iters, blocks, clocks, sink);                 \
    run<OP, 256, 32, true>(NAME " from / to HBM", buf, bytes, iters, blocks, clocks, sink);                 \
    run<OP, 512, 16, true>(NAME " from / to HBM", buf, bytes, iters, blocks, clocks, sink);                 \
    run<OP, 256, 16, false>(NAME " in L2", buf, bytes, iters * 4, blocks, clocks, sink);                    \
    run<OP, 512, 16, false>(NAME " in L2", buf, bytes, iters * 4, blocks, clocks, sink);                    \
  }
  if (only < 0 || only == 0) { ROW(0, "LDS-DMA load") }
  if (only < 0 || only == 1) { ROW(1, "load to VGPRs") }
  if (only < 0 || only == 2) { ROW(2, "store") }
  return 0;
}
