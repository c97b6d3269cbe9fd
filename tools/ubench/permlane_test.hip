// What v_permlane32_swap / DPP mirrors / readlane return, lane by lane (input: v = lane id).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  const unsigned lane = threadIdx.x;
  unsigned u = lane, w = lane;
  asm volatile("" : "+v"(w));
  const auto r = __builtin_amdgcn_permlane32_swap(u, w, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  out[128 + lane] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lane, 0xB1, 0xf, 0xf, true);
  out[192 + lane] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lane, 0x4E, 0xf, 0xf, true);
  out[256 + lane] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lane, 0x141, 0xf, 0xf, true);
  out[320 + lane] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lane, 0x140, 0xf, 0xf, true);
}
int main() {
  unsigned *d, h[384];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char *names[6] = {"swap r[0]", "swap r[1]", "quad 0xB1", "quad 0x4E", "half_mirror", "row_mirror"};
  for (int t = 0; t < 6; ++t) {
    printf("%-12s", names[t]);
    for (int i = 0; i < 64; ++i) printf(" %u", h[64 * t + i]);
    printf("\n");
  }
  return 0;
}
