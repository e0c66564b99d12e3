// mfma_peak.hip — what the matrix pipe of THIS chip sustains on random operands (MI355X_MICROARCH.md, DVFS give-back): bare MFMA loops,
// one or two waves per SIMD, with and without LDS operand reads, clock stamped in the kernel (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// MODE 0: 16x16x32 f16, operands in registers; 1: 32x32x16 f16, registers; 2: 16x16x32 with LDS reads per MFMA (RD16 sixteenths of a
// ds_read_b128 per MFMA); 3: 32x32x16 with LDS reads
template <int MODE, int RD16>
__global__ void __launch_bounds__(256, 2) peak_kernel(const uint4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  uint4* l = reinterpret_cast<uint4*>(smem);
  for (int i = threadIdx.x; i < 2048; i += 256) l[i] = src[(blockIdx.x * 2048 + i) & 0xFFFF];
  __syncthreads();
  union U { uint4 u; f16x8 h; };
  U a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i].u = src[(threadIdx.x * 4 + i) & 0xFFFF]; b[i].u = src[(threadIdx.x * 4 + i + 1024) & 0xFFFF]; }
  f32x4 c4[16];
  f32x16 c16[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) c4[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned off = lane * 16;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 4) {                    // the CDNA3 form, K = 16: is it still full rate per FLOP on gfx950?
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const f16x4 a4 = {a[i & 3].h[0], a[i & 3].h[1], a[i & 3].h[2], a[i & 3].h[3]}, b4 = {b[(i >> 2) & 3].h[0], b[(i >> 2) & 3].h[1], b[(i >> 2) & 3].h[2], b[(i >> 2) & 3].h[3]};
        c4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c4[i], 0, 0, 0);
      }
    } else if constexpr (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if constexpr (MODE == 2) {
          if ((i * RD16) / 16 != ((i + 1) * RD16) / 16) { b[i & 3].u = l[(off >> 4) + ((i * 64) & 1023)]; }
        }
        c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3].h, b[(i >> 2) & 3].h, c4[i], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (MODE == 3) {
          if ((i * RD16) / 8 != ((i + 1) * RD16) / 8) { b[i & 3].u = l[(off >> 4) + ((i * 64) & 1023)]; }
        }
        c16[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3].h, b[(i >> 1) & 3].h, c16[i & 3], 0, 0, 0);
      }
    }
    off = (off + 1024) & 0x7FFF;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c4[i][0] + c4[i][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c16[i][0] + c16[i][15];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE, int RD16> void run(const char* name, const uint4* src, float* out, unsigned long long* st, int wg_per_cu) {
  const int iters = 40000, grid = 256 * wg_per_cu;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&peak_kernel<MODE, RD16>), hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 60; ++w) hipLaunchKernelGGL((peak_kernel<MODE, RD16>), dim3(grid), dim3(256), 32768, 0, src, out, st, iters);      // ~ a second of load first: DVFS settles
  hipEventRecord(e0);
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((peak_kernel<MODE, RD16>), dim3(grid), dim3(256), 32768, 0, src, out, st, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * 2);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> clk;
  for (int i = 0; i < grid; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);     // GHz: memrealtime ticks at 100 MHz
  std::sort(clk.begin(), clk.end());
  const double mfma_per_it = (MODE == 0 || MODE == 2 || MODE == 4) ? 16 : 8;
  const double flop_per = MODE == 4 ? 16.0 * 16 * 16 * 2 : (MODE == 0 || MODE == 2) ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2;
  const double flops = (double)grid * 4 * iters * mfma_per_it * flop_per * reps;
  printf("%-44s %d wg/CU: %7.1f TFLOP/s (%.3f of 2500)  in-kernel clock median %.2f GHz\n", name, wg_per_cu, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15, clk[clk.size() / 2]);
}

int main() {
  uint4* src; float* out; unsigned long long* st;
  hipMalloc(&src, 65536 * 16); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&st, 4096 * 16);
  std::vector<_Float16> h(65536 * 8);
  srand(1);
  for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 4.f);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int wg = 1; wg <= 2; ++wg) {
    run<0, 0>("16x16x32 f16, operands in registers", src, out, st, wg);
    run<1, 0>("32x32x16 f16, operands in registers", src, out, st, wg);
    run<4, 0>("16x16x16 f16 (CDNA3 form), operands in registers", src, out, st, wg);
    run<2, 8>("16x16x32 f16 + 0.5 ds_read_b128 per MFMA", src, out, st, wg);
    run<2, 5>("16x16x32 f16 + 0.31 ds_read_b128 per MFMA", src, out, st, wg);
    run<3, 8>("32x32x16 f16 + 1 ds_read_b128 per MFMA", src, out, st, wg);
    run<3, 5>("32x32x16 f16 + 0.62 ds_read_b128 per MFMA", src, out, st, wg);
  }
  return 0;
}
