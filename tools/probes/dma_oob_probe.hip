// dma_oob_probe.hip — does `buffer_load_dwordx4 ... offen lds` write ZEROS into LDS for lanes whose offset fails the range check, is the
// instruction's scalar offset left out of that check, and does M0 take a full LDS byte address (beyond 64 KiB)?  (conv_trunk.hip relies on all three)
//   hipcc -O3 --offload-arch=gfx950 tools/probes/dma_oob_probe.hip -o tools/probes/dma_oob_probe && tools/probes/dma_oob_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k(const unsigned char* src, unsigned bytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  const int lane = threadIdx.x;
  for (int i = lane; i < 140 * 256; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xDEADBEEFu;       // 140 KiB of poison
  __syncthreads();
  const unsigned long long u = reinterpret_cast<unsigned long long>(src);
  u32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((unsigned)u);
  rs[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) & 0xFFFFu;
  rs[2] = bytes;
  rs[3] = 0x00020000u;
  const unsigned voff = (lane & 1) ? 0xFFFFFFFFu : (unsigned)lane * 16u;      // odd lanes out of range
  unsigned keep;
  const unsigned dst1 = lds0 + 1024u, dst2 = lds0 + 100u * 1024u, soff = 4096u;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst1), "s"(0u) : "memory");
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst2), "s"(soff) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int j = 0; j < 4; ++j) {
    out[lane * 4 + j] = reinterpret_cast<unsigned*>(smem + 1024)[lane * 4 + j];
    out[256 + lane * 4 + j] = reinterpret_cast<unsigned*>(smem + 100 * 1024)[lane * 4 + j];
  }
  out[512 + lane] = reinterpret_cast<unsigned*>(smem)[lane];          // untouched neighbourhood
}
int main() {
  const unsigned bytes = 1024;                                         // range: offsets >= 1024 are out of range; the second DMA's soffset (4096) must NOT count
  unsigned char* src; unsigned* out;
  hipMalloc(&src, 8192); hipMalloc(&out, 4 * 1024);
  std::vector<unsigned> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = 0x1000u + i;
  hipMemcpy(src, h.data(), 8192, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 140 * 1024, 0, src, bytes, out);
  std::vector<unsigned> o(1024);
  hipError_t e = hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
  printf("status %s\n", hipGetErrorString(e));
  int ok_in = 0, ok_zero = 0, poison = 0, ok2 = 0, zero2 = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const unsigned v = o[l * 4 + j], v2 = o[256 + l * 4 + j];
      if (!(l & 1)) { ok_in += v == 0x1000u + l * 4 + j; ok2 += v2 == 0x1000u + 1024 + l * 4 + j; }
      else { ok_zero += v == 0; poison += v == 0xDEADBEEFu; zero2 += v2 == 0; }
    }
  printf("DMA 1 (M0 = 1 KiB, soffset 0): in-range lanes correct %d/128, out-of-range lanes zero %d/128 (left as poison %d)\n", ok_in, ok_zero, poison);
  printf("DMA 2 (M0 = 100 KiB, soffset 4096 > num_records): in-range lanes read base + soffset + voffset correctly %d/128, out-of-range lanes zero %d/128\n", ok2, zero2);
  printf("first words of LDS untouched: %s\n", o[512] == 0xDEADBEEFu ? "yes" : "NO");
  return 0;
}
