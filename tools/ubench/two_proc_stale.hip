// Reproducer for the 1-rank != 2-rank record of tests/test_gpu_configs.py: do a kernel's
// plain stores reach the NEXT kernel of the same stream when a SECOND PROCESS is busy on the same GPU?
//
//   producer<<<G>>>: buf[i] = tag(iter, i), plain 16-byte stores, block b writes chunk b
//   consumer<<<G>>>: block b reads chunk (b + SHIFT) % G -- SHIFT = 1: another XCD than the one that wrote it (block b
//                    runs on XCD b % 8), SHIFT = 8: the same XCD, another CU -- with plain / `sc1` / `sc0 sc1` loads and
//                    counts the 16-byte words that do not carry this iteration's tag (= stale: the previous iteration's)
// Same stream, nothing between the two launches but the kernel boundary.  One process: the boundary's release /
// acquire must make every count 0.  Two processes (`two_proc_stale 2`, fork() before the first HIP call): the same
// loop in both, concurrently.  Prints per process and load kind: launches, stale words, launches with a stale word.
//
//   make -C tools/ubench two_proc_stale && tools/ubench/two_proc_stale 1 400 && tools/ubench/two_proc_stale 2 400
//   two_proc_stale <processes> <iterations> [blocks = 2048] [burst = 1: pairs between two host syncs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>

constexpr int kThreads = 256, kPerThread = 8;   // a block = 32 KiB; blocks = argv[3] (default 2048 = 64 MiB; 256 = 8 MiB stays in the L2s)
static int kGrid = 2048, burst = 1;

__device__ __forceinline__ unsigned tag(unsigned iter, size_t i) { return iter * 2654435761u + (unsigned)i; }

__global__ void producer(uint4 *buf, unsigned iter) {
  const size_t base = ((size_t)blockIdx.x * kPerThread) * kThreads + threadIdx.x;
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)k * kThreads;
    const unsigned t = tag(iter, i);
    buf[i] = make_uint4(t, t ^ 1u, t ^ 2u, t ^ 3u);
  }
}

template <int KIND>   // 0 plain, 1 sc1, 2 sc0 sc1
__global__ void consumer(const uint4 *buf, unsigned iter, int shift, unsigned long long *stale) {
  const int blk = (blockIdx.x + shift) % gridDim.x;
  const size_t base = ((size_t)blk * kPerThread) * kThreads + threadIdx.x;
  unsigned bad = 0;
  for (int k = 0; k < kPerThread; ++k) {
    const size_t i = base + (size_t)k * kThreads;
    uint4 v;
    if (KIND == 0) v = buf[i];
    else if (KIND == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + i) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(buf + i) : "memory");
    const unsigned t = tag(iter, i);
    bad += (v.x != t) | (v.y != (t ^ 1u)) | (v.z != (t ^ 2u)) | (v.w != (t ^ 3u));
  }
  if (bad) atomicAdd(stale, (unsigned long long)bad);
}

static int run(int who, int iters) {
  const size_t kWords = (size_t)kGrid * kThreads * kPerThread;     // uint4 words
  uint4 *buf;
  unsigned long long *stale, host = 0;
  if (hipMalloc(&buf, kWords * sizeof(uint4)) != hipSuccess || hipMalloc(&stale, 8) != hipSuccess) return 1;
  hipStream_t st;
  hipStreamCreate(&st);
  const char *names[3] = {"plain", "sc1", "sc0 sc1"};
  for (int shift : {1, 8})
    for (int kind = 0; kind < 3; ++kind) {
      unsigned long long total = 0;
      int launches_bad = 0;
      // `burst` producer / consumer pairs back to back between two host syncs (the tower queues ~90 kernels per pass
      // without one); the stale counter accumulates on the device
      for (int it = 1; it <= iters; it += burst) {
        hipMemsetAsync(stale, 0, 8, st);
        for (int k = 0; k < burst; ++k) {
          const unsigned tg = (unsigned)(it + k + 1000 * kind + 100000 * shift);
          producer<<<kGrid, kThreads, 0, st>>>(buf, tg);
          if (kind == 0) consumer<0><<<kGrid, kThreads, 0, st>>>(buf, tg, shift, stale);
          else if (kind == 1) consumer<1><<<kGrid, kThreads, 0, st>>>(buf, tg, shift, stale);
          else consumer<2><<<kGrid, kThreads, 0, st>>>(buf, tg, shift, stale);
        }
        hipMemcpyAsync(&host, stale, 8, hipMemcpyDeviceToHost, st);
        hipStreamSynchronize(st);
        total += host;
        launches_bad += host != 0;
      }
      printf("blocks %d  process %d  shift %d (%s XCD)  %-7s loads: %d launches, %llu stale 16-byte words in %d bursts of %d\n", kGrid, who, shift,
             shift % 8 ? "other" : "same", names[kind], iters, total, launches_bad, burst);
      fflush(stdout);
    }
  return 0;
}

int main(int argc, char **argv) {
  const int procs = argc > 1 ? atoi(argv[1]) : 1, iters = argc > 2 ? atoi(argv[2]) : 400;
  if (argc > 3) kGrid = atoi(argv[3]);
  if (argc > 4) burst = atoi(argv[4]);
  int who = 0;
  for (int p = 1; p < procs; ++p)
    if (fork() == 0) { who = p; break; }      // (before the first HIP call: every process gets its own context and queues)
  const int rc = run(who, iters);
  if (who == 0) while (wait(nullptr) > 0) {}
  return rc;
}
