#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ inline __amdgpu_buffer_rsrc_t mk(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__global__ void k(const float* src, float* dst, int n_valid) {
  __amdgpu_buffer_rsrc_t rs = mk(src, n_valid * 4), rd = mk(dst, n_valid * 4);
  int off = threadIdx.x * 4;
  float a = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
  float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 256, 0, 0));
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a + b), rd, off, 0, 0);
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, a - b), rd, off + 256, 0, 0);
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = i;
  float *s, *d; hipMalloc(&s, 1024); hipMalloc(&d, 1024); hipMemcpy(s, h, 1024, hipMemcpyHostToDevice); hipMemset(d, 0xff, 1024);
  k<<<1, 64>>>(s, d, 100);
  float o[256]; hipMemcpy(o, d, 1024, hipMemcpyDeviceToHost);
  printf("%g %g %g %g | %g %g %g\n", o[0], o[35], o[36], o[63], o[64], o[99], o[100]);
  return 0;
}
