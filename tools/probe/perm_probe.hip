#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  int a = lane, b = 100 + lane;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  out[lane] = a; out[64 + lane] = b;
  int c = lane, d = 100 + lane;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  out[128 + lane] = c; out[192 + lane] = d;
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[4] = {"swap32 a", "swap32 b", "swap16 a", "swap16 b"};
  for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; i += 8) printf(" %d", h[64 * r + i]); printf("\n"); }
  return 0;
}
