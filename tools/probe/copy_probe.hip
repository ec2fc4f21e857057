// dev probe: device-to-device copy variants, 512 MB, GB/s (read + write).  hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float4_ __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_gs(const float4_* __restrict__ s, float4_* __restrict__ d, size_t n) {   // grid-stride, U in flight
  const size_t st = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * st < n; i += U * st) {
    float4_ v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * st) : s[i + u * st];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * st); else d[i + u * st] = v[u]; }
  }
  for (; i < n; i += st) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_blk(const float4_* __restrict__ s, float4_* __restrict__ d, size_t n) {   // block-contiguous U*4 KB per step
  const size_t st = (size_t)gridDim.x * 256 * U;
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  for (; i + (U - 1) * 256 < n; i += st) {
    float4_ v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
  }
}
template <int U>
__global__ __launch_bounds__(256) void k_once(const float4_* __restrict__ s, float4_* __restrict__ d, size_t n) {   // one shot, no loop
  const size_t i = ((size_t)blockIdx.x * 256 * U) + threadIdx.x;
  float4_ v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = s[i + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) d[i + u * 256] = v[u];
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
  const size_t bytes = 512ull << 20, n = bytes / 16;
  float4_ *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes);
  auto rep = [&](const char* nm, float ms) { printf("%-28s %7.1f GB/s\n", nm, 2.0 * bytes / ms / 1e6); };
  for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, 64, "gs U4 grid %d", g); rep(nm, timeit([&] { k_gs<4, false><<<g, 256>>>(s, d, n); }));
    snprintf(nm, 64, "gs U8 grid %d", g); rep(nm, timeit([&] { k_gs<8, false><<<g, 256>>>(s, d, n); }));
    snprintf(nm, 64, "gs U4 nt grid %d", g); rep(nm, timeit([&] { k_gs<4, true><<<g, 256>>>(s, d, n); }));
    snprintf(nm, 64, "blk U4 grid %d", g); rep(nm, timeit([&] { k_blk<4, false><<<g, 256>>>(s, d, n); }));
    snprintf(nm, 64, "blk U8 grid %d", g); rep(nm, timeit([&] { k_blk<8, false><<<g, 256>>>(s, d, n); }));
    snprintf(nm, 64, "blk U4 nt grid %d", g); rep(nm, timeit([&] { k_blk<4, true><<<g, 256>>>(s, d, n); }));
  }
  rep("once U1", timeit([&] { k_once<1><<<(unsigned)(n / 256), 256>>>(s, d, n); }));
  rep("once U2", timeit([&] { k_once<2><<<(unsigned)(n / 512), 256>>>(s, d, n); }));
  rep("once U4", timeit([&] { k_once<4><<<(unsigned)(n / 1024), 256>>>(s, d, n); }));
  rep("once U8", timeit([&] { k_once<8><<<(unsigned)(n / 2048), 256>>>(s, d, n); }));
  rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }));
  return 0;
}
