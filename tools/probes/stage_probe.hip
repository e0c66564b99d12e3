// stage_probe.hip — what bounds the global -> LDS staging of the Real-ESRGAN body kernels: bytes per clock and CU for 1 KiB wave-pieces
// (64 lanes x 16 B) by ACCESS PATTERN, load form and cache policy. Round-3's phase probe shows conv_rows16_kernel's staging alone takes
// longer than its MFMAs alone; this probe asks whether the 64-byte pixel segments of the NHWC layout (half of a 128-byte line per pixel
// and 32-channel chunk) are what it pays for.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stage_probe.hip -o /tmp/stage_probe && /tmp/stage_probe
// patterns: 0 contiguous 1 KiB pieces swept through a large buffer; 1 pixel segments: 16 pixels x 64 B at a 384-byte pixel stride (NHWC,
// 192 channels fp16, one 32-channel chunk); 2 the same, both 64-byte halves of a line requested back to back (chunks c, c+1); 3 contiguous
// pieces out of a 256 KiB table every workgroup re-reads (weights: L2 hits); 4 pixel segments at a 128-byte pixel stride (64-channel tensor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int FORM, int AUX, int DEPTH>      // FORM 0: buffer load -> VGPR -> ds_write_b128; 1: LDS-DMA (global_load_lds_dwordx4)
__global__ void __launch_bounds__(1024, 1) stage_kernel(const unsigned char* __restrict__ src, unsigned long long bytes, int pattern, int pieces_per_wave,
                                                        unsigned* __restrict__ sink, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int nw = (int)blockDim.x >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // a wave's stream of pieces: global piece index gp = ((block * nw + wave) * pieces_per_wave + i)
  const unsigned long long wbase = ((unsigned long long)blockIdx.x * nw + wave) * (unsigned long long)pieces_per_wave;
  auto addr = [&](int i) -> const unsigned char* {
    unsigned long long gp = wbase + (unsigned long long)i;
    unsigned long long off;
    if (pattern == 0) {
      off = ((gp * 1024ull) & (bytes - 1)) + (unsigned long long)lane * 16ull;
    } else if (pattern == 1) {                    // 16 consecutive pixels, one 64-byte segment each; consecutive pieces = consecutive pixel runs, chunk fixed per 18 pieces
      const unsigned long long run = gp, c = (gp >> 4) & 3ull;
      off = (((run * 16ull + (unsigned long long)(lane >> 2)) * 384ull) & (bytes - 1)) + c * 64ull + (unsigned long long)(lane & 3) * 16ull;
    } else if (pattern == 2) {                    // pieces 2k, 2k+1: the same 16 pixels, segments c and c+1 (one whole 128-byte line between them)
      const unsigned long long run = gp >> 1, c = (gp & 1ull) + 2ull * ((gp >> 5) & 1ull);
      off = (((run * 16ull + (unsigned long long)(lane >> 2)) * 384ull) & (bytes - 1)) + c * 64ull + (unsigned long long)(lane & 3) * 16ull;
    } else if (pattern == 3) {
      off = ((gp * 1024ull) & (256ull * 1024ull - 1)) + (unsigned long long)lane * 16ull;
    } else {
      const unsigned long long run = gp, c = (gp >> 4) & 1ull;
      off = (((run * 16ull + (unsigned long long)(lane >> 2)) * 128ull) & (bytes - 1)) + c * 64ull + (unsigned long long)(lane & 3) * 16ull;
    }
    return src + off;
  };
  unsigned char* my = smem + wave * (DEPTH * 1024);
  unsigned acc = 0;
  if constexpr (FORM == 0) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 0x7FFFFFF0, 0x00020000);
    u32x4 r[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(addr(d) - src), 0, AUX);
    for (int i = DEPTH; i < pieces_per_wave + DEPTH; i += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        *reinterpret_cast<u32x4*>(my + d * 1024 + lane * 16) = r[d];
        if (i + d < pieces_per_wave) r[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(addr(i + d) - src), 0, AUX);
      }
    }
    acc = *reinterpret_cast<unsigned*>(my + lane * 4);
  } else {
    // DEPTH pieces in flight: issue piece i, then wait until at most DEPTH - 1 are outstanding
    for (int i = 0; i < pieces_per_wave; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)addr(i),
                                       (__attribute__((address_space(3))) void*)(my + (i % DEPTH) * 1024), 16, 0, AUX);
      if (i >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc = *reinterpret_cast<unsigned*>(my + lane * 4);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0x12345678u) sink[0] = acc;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int FORM, int AUX, int DEPTH>
static void run(const char* name, const unsigned char* src, unsigned long long bytes, int pattern, int nwaves, unsigned* sink, unsigned long long* stamps) {
  const int grid = 256, ppw = 2048 * 8 / nwaves / 4;           // 4 MiB per workgroup           // 16 MiB per workgroup, 4 GiB of requests per launch
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<FORM, AUX, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * DEPTH * 1024 > 160 * 1024 ? 160 * 1024 : 16 * DEPTH * 1024));
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stage_kernel<FORM, AUX, DEPTH>), dim3(grid), dim3(nwaves * 64), nwaves * DEPTH * 1024, 0, src, bytes, pattern, ppw, sink, stamps);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  std::vector<unsigned long long> h(grid * 2);
  CK(hipMemcpy(h.data(), stamps, grid * 16, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double ghz = cyc / rt * 0.1;                       // s_memrealtime ticks at 100 MHz
  const double total = (double)grid * nwaves * ppw * 1024.0;
  printf("%-34s pattern %d waves %d depth %d : %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  (%.2f GHz)\n", name, pattern, nwaves, DEPTH, best * 1e3, total / (best * 1e-3) / 1e12,
         total / 256.0 / (cyc / grid), ghz);
}

int main() {
  const unsigned long long bytes = 512ull << 20;           // larger than the Infinity Cache: HBM-served; the body tensors of a 10-frame batch are 150-230 MB
  unsigned char* src; unsigned* sink; unsigned long long* stamps;
  CK(hipMalloc(&src, bytes + 4096)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stamps, 256 * 16));
  CK(hipMemset(src, 1, bytes + 4096));
  const unsigned long long small = 128ull << 20;           // Infinity-Cache resident once touched
  const bool second = getenv("STAGE_PROBE_2") != nullptr;
  if (second) {
    // round 2 of the probe: the per-CU rate by NUMBER OF WAVES issuing (4..16) and by pieces in flight per wave (2..8), LDS-DMA only
    for (int p : {3, 1}) {
      for (int nw : {2, 4, 8, 12, 16}) {
        run<1, 0, 2>("lds-dma plain", src, small, p, nw, sink, stamps);
        run<1, 0, 4>("lds-dma plain", src, small, p, nw, sink, stamps);
        run<1, 0, 8>("lds-dma plain", src, small, p, nw, sink, stamps);
      }
    }
    return 0;
  }
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned long long B = pass == 0 ? small : bytes;
    printf("---- tensor of %llu MiB\n", B >> 20);
    for (int nw = 4; nw <= 8; nw += 4) {
      for (int p = 0; p <= 4; ++p) {
        run<0, 0, 8>("vgpr+ds_write plain", src, B, p, nw, sink, stamps);
        run<1, 0, 8>("lds-dma plain", src, B, p, nw, sink, stamps);
        run<1, 0, 16>("lds-dma plain", src, B, p, nw, sink, stamps);
        run<0, 16, 8>("vgpr+ds_write sc1", src, B, p, nw, sink, stamps);      // aux 16 = sc1
        run<1, 16, 8>("lds-dma sc1", src, B, p, nw, sink, stamps);
        run<1, 2, 8>("lds-dma nt", src, B, p, nw, sink, stamps);
      }
    }
  }
  return 0;
}
