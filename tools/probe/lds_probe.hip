// dev tool: time LDS read patterns (ds_read_b128 / ds_read_b64_tr_b16 / ds_read_b64) given per-lane byte offsets.
// usage: lds_probe  (patterns are built in main()); prints cycles per wave-instruction with 4 and 8 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>
#include <functional>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 x4v;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))

template <int KIND>
__global__ __launch_bounds__(512) void probe(const int* offs, int nrep, long long* out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = i;
  __syncthreads();
  const int o = offs[lane];
  uint32_t acc = 0;
  long long t0 = clock64();
  for (int r = 0; r < nrep; ++r) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const char* p = smem + o + u * 2048;     // 16 independent reads, same bank pattern (2048 B = 8 x 256)
      if (KIND == 0) { u32x4 v; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(v) : "v"((uint32_t)(uintptr_t)LDSP(char, p))); acc += v[0] ^ v[3]; }
      else if (KIND == 3) { u32x4 v = {acc, acc, acc, acc}; asm volatile("ds_write_b128 %0, %1" :: "v"((uint32_t)(uintptr_t)LDSP(char, p)), "v"(v) : "memory"); }
      else if (KIND == 4) { u32x2 v = {acc, acc}; asm volatile("ds_write_b64 %0, %1" :: "v"((uint32_t)(uintptr_t)LDSP(char, p)), "v"(v) : "memory"); }
      else if (KIND == 1) { u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDSP(x4v, p))); acc += v[0] ^ v[1]; }
      else { u32x2 v; asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(v) : "v"((uint32_t)(uintptr_t)LDSP(char, p))); acc += v[0] ^ v[1]; }
    }
  }
  long long t1 = clock64();
  if (lane == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

struct Pat { std::string name; int kind; std::function<int(int g, int li)> f; };

int main() {
  std::vector<Pat> pats;
  auto swz = [](int row, int chunk, int phi) { return row * 128 + ((chunk ^ phi) << 4); };
  // A: b128 row operand, row = li, chunk = 2g (ks = 0); phi variants
  pats.push_back({"b128 rowop phi=r&7 (current)", 0, [=](int g, int li) { return swz(li, 2 * g, li & 7); }});
  pats.push_back({"b128 rowop no swizzle", 0, [=](int g, int li) { return swz(li, 2 * g, 0); }});
  pats.push_back({"b128 rowop phi=(r>>1)&7", 0, [=](int g, int li) { return swz(li, 2 * g, (li >> 1) & 7); }});
  pats.push_back({"b128 rowop phi=linear-like ((r>>1)&3)<<1|((r>>3)&1)", 0, [=](int g, int li) { return swz(li, 2 * g, (((li >> 1) & 3) << 1) | ((li >> 3) & 1)); }});
  // B old: tr read, row = 4g + (li>>2), byte col = 32 (li&3) + 8 dt (dt = 0), phi = r&7
  pats.push_back({"tr old (stride-32 segs) phi=r&7 (current)", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 32 * (li & 3); return r * 128 + (((colb >> 4) ^ (r & 7)) << 4) + (colb & 15); }});
  pats.push_back({"tr old no swizzle", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 32 * (li & 3); return r * 128 + colb; }});
  // B new: tr read, 4 lanes of a row read 32 contiguous bytes
  pats.push_back({"tr new (contiguous 32B) phi=r&7", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 8 * (li & 3); return r * 128 + (((colb >> 4) ^ (r & 7)) << 4) + (colb & 15); }});
  pats.push_back({"tr new no swizzle", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 8 * (li & 3); return r * 128 + colb; }});
  pats.push_back({"tr new phi=(r>>1)&7", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 8 * (li & 3); return r * 128 + (((colb >> 4) ^ ((r >> 1) & 7)) << 4) + (colb & 15); }});
  // 8-byte-granular swizzle for the OLD tr pattern: unit8 ^ ((r>>1)&3)
  pats.push_back({"tr old, 8B-granular xor ((r>>1)&3)", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int u = 4 * (li & 3); u ^= (r >> 1) & 3; return r * 128 + u * 8; }});
  auto phis = [](int r) { return (((r >> 1) & 3) << 1) | ((r ^ (r >> 3)) & 1); };
  for (int ks = 0; ks < 2; ++ks)
    pats.push_back({std::string("b128 rowop phi* ks=") + char('0' + ks), 0, [=](int g, int li) { return swz(li, 2 * g + ks, phis(li)); }});
  for (int dt = 0; dt < 4; dt += 3)
    pats.push_back({std::string("tr new phi* dt=") + char('0' + dt), 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 32 * dt + 8 * (li & 3); return r * 128 + (((colb >> 4) ^ phis(r)) << 4) + (colb & 15); }});
  pats.push_back({"write b128 rowop (row=li, chunk 2g) phi=r&7", 3, [=](int g, int li) { return swz(li, 2 * g, li & 7); }});
  pats.push_back({"write b128 rowop phi*", 3, [=](int g, int li) { return swz(li, 2 * g, phis(li)); }});
  pats.push_back({"write b64 slab (row=li, byte col 8g) phi=r&7", 4, [=](int g, int li) { return li * 128 + ((((8 * g) >> 4) ^ (li & 7)) << 4) + ((8 * g) & 15); }});
  pats.push_back({"write b64 slab phi*", 4, [=](int g, int li) { return li * 128 + ((((8 * g) >> 4) ^ phis(li)) << 4) + ((8 * g) & 15); }});
  // slab tr read (P4): rows 4g+(li>>2), 32 contiguous bytes, phi=r&7 / phi*
  pats.push_back({"tr slab contiguous phi=r&7 (P4 current)", 1, [=](int g, int li) { int r = 4 * g + (li >> 2); int colb = 8 * (li & 3); return r * 128 + (((colb >> 4) ^ (r & 7)) << 4) + (colb & 15); }});
  // plain b64 reads, all lanes distinct consecutive (reference)
  pats.push_back({"b64 linear (reference)", 2, [=](int g, int li) { return (16 * g + li) * 8; }});
  pats.push_back({"b128 linear (reference)", 0, [=](int g, int li) { return (16 * g + li) * 16; }});
  // X kernel's transposed copy reads: b64 at 136*drow + 8g
  pats.push_back({"b64 MT_LDB=136 drow=16(li>>2)+(li&3)", 2, [=](int g, int li) { int drow = 16 * (li >> 2) + (li & 3); return drow * 136 + 8 * g; }});

  int* d_off; long long* d_out; uint32_t* d_sink;
  hipMalloc(&d_off, 64 * sizeof(int)); hipMalloc(&d_out, 4096 * sizeof(long long)); hipMalloc(&d_sink, 256 * 512 * 4);
  const int nrep = 200;
  for (auto& p : pats) {
    int h[64];
    for (int l = 0; l < 64; ++l) h[l] = p.f(l >> 4, l & 15);
    hipMemcpy(d_off, h, sizeof(h), hipMemcpyHostToDevice);
    for (int nt : {256, 512}) {
      for (int it = 0; it < 2; ++it) {
        if (p.kind == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(nt), 65536, 0, d_off, nrep, d_out, d_sink);
        else if (p.kind == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(nt), 65536, 0, d_off, nrep, d_out, d_sink);
        else if (p.kind == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(nt), 65536, 0, d_off, nrep, d_out, d_sink);
        else if (p.kind == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(nt), 65536, 0, d_off, nrep, d_out, d_sink);
        else hipLaunchKernelGGL(probe<2>, dim3(256), dim3(nt), 65536, 0, d_off, nrep, d_out, d_sink);
      }
      hipDeviceSynchronize();
      std::vector<long long> o(256 * nt / 64);
      hipMemcpy(o.data(), d_out, o.size() * sizeof(long long), hipMemcpyDeviceToHost);
      double s = 0; for (auto v : o) s += (double)v;
      s /= o.size();
      // clock64 = 100 MHz wall clock? report raw ticks per wave-instruction and per-CU instruction
      printf("%-58s waves/CU %d: %.3f ticks / wave-instr, %.3f ticks / CU-instr\n", p.name.c_str(), nt / 64, s / (nrep * 16.0), s / (nrep * 16.0) / (nt / 64));
    }
  }
  return 0;
}
