// stream_overlap_probe.hip — do two chains of small dependent kernels on two HIP streams run side by side on this system?
// Each kernel: `wgs` workgroups of 256 threads spinning for `us` microseconds (s_memrealtime, 100 MHz). A chain = n launches on one stream.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/stream_overlap_probe.hip -o /tmp/stream_overlap_probe && /tmp/stream_overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void spin_lds(unsigned ticks, unsigned* sink) {          // the same with dynamic LDS (launch parameter): how many workgroups fit a CU
  extern __shared__ unsigned lds_[];
  lds_[threadIdx.x] = ticks;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (sink && lds_[threadIdx.x ^ 1] == 0xFFFFFFFFu) *sink = 1;
}

// chains with LDS: stream a = na kernels of ua us on wa workgroups, stream b = nb x ub on wb, every workgroup holding `lds` bytes
static double chain_lds(hipStream_t a, hipStream_t b, int na, int ua, int wa, int nb, int ub, int wb, int lds, double* tb_done) {
  CK(hipDeviceSynchronize());
  hipEvent_t eb;
  CK(hipEventCreate(&eb));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < (na > nb ? na : nb); ++i) {
    if (a && i < na) hipLaunchKernelGGL(spin_lds, dim3(wa), dim3(256), lds, a, (unsigned)(ua * 100), nullptr);
    if (b && i < nb) hipLaunchKernelGGL(spin_lds, dim3(wb), dim3(256), lds, b, (unsigned)(ub * 100), nullptr);
  }
  if (b) { CK(hipEventRecord(eb, b)); CK(hipEventSynchronize(eb)); }
  *tb_done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CK(hipDeviceSynchronize());
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

__global__ void spin(unsigned ticks, unsigned* sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}

// unequal chains: stream a runs na kernels of ua microseconds, stream b nb kernels of ub; returns wall ms and (through tb_done) when b's chain finished
static double chain2(hipStream_t a, hipStream_t b, int na, int ua, int nb, int ub, int wgs, double* tb_done) {
  CK(hipDeviceSynchronize());
  hipEvent_t eb;
  CK(hipEventCreate(&eb));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < (na > nb ? na : nb); ++i) {
    if (a && i < na) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, a, (unsigned)(ua * 100), nullptr);
    if (b && i < nb) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, b, (unsigned)(ub * 100), nullptr);
  }
  if (b) { CK(hipEventRecord(eb, b)); CK(hipEventSynchronize(eb)); }
  *tb_done = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  CK(hipDeviceSynchronize());
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

static double chain(hipStream_t a, hipStream_t b, int n, int wgs, int us, bool graph) {
  CK(hipDeviceSynchronize());
  hipGraphExec_t ga = nullptr, gb = nullptr;
  if (graph) {
    for (int k = 0; k < 2; ++k) {
      hipStream_t s = k ? b : a;
      if (!s) continue;
      hipGraph_t g;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, s, (unsigned)(us * 100), nullptr);
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(k ? &gb : &ga, g, nullptr, nullptr, 0));
    }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (graph) {
    if (a) CK(hipGraphLaunch(ga, a));
    if (b) CK(hipGraphLaunch(gb, b));
  } else {
    for (int i = 0; i < n; ++i) {
      if (a) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, a, (unsigned)(us * 100), nullptr);
      if (b) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, b, (unsigned)(us * 100), nullptr);
    }
  }
  CK(hipDeviceSynchronize());
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main() {
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  for (int graph = 0; graph < 2; ++graph)
    for (int wgs : {64, 256, 1024})
      for (int us : {5, 50}) {
        const int n = us == 5 ? 400 : 100;
        chain(a, b, 20, wgs, us, graph);
        const double ta = chain(a, nullptr, n, wgs, us, graph), tb = chain(nullptr, b, n, wgs, us, graph), tab = chain(a, b, n, wgs, us, graph);
        printf("%s  %4d workgroups x %2d us x %3d launches per stream: A alone %7.2f ms  B alone %7.2f ms  both %7.2f ms  (ideal overlap %.2f, serial %.2f)\n",
               graph ? "graph" : "eager", wgs, us, n, ta, tb, tab, ta > tb ? ta : tb, ta + tb);
      }
  for (int wgs : {64, 256}) {
    double tb = 0, tb2 = 0;
    chain2(a, b, 20, 50, 20, 5, wgs, &tb);
    const double ta = chain2(a, nullptr, 100, 50, 0, 0, wgs, &tb);
    chain2(nullptr, b, 0, 0, 400, 5, wgs, &tb);
    const double tab = chain2(a, b, 100, 50, 400, 5, wgs, &tb2);
    printf("unequal %4d workgroups: A = 100 x 50 us alone %.2f ms, B = 400 x 5 us alone %.2f ms; together: all done %.2f ms, B's chain done after %.2f ms\n", wgs, ta, tb, tab, tb2);
  }
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int lds : {80 * 1024, 40 * 1024, 16 * 1024}) {
    double tb = 0, tb2 = 0;
    chain_lds(a, b, 10, 50, 512, 10, 5, 200, lds, &tb);
    const double ta = chain_lds(a, nullptr, 100, 50, 512, 0, 0, 0, lds, &tb);
    chain_lds(nullptr, b, 0, 0, 0, 400, 5, 200, lds, &tb);
    const double tab = chain_lds(a, b, 100, 50, 512, 400, 5, 200, lds, &tb2);
    printf("LDS %3d KiB per workgroup: A = 100 x 50 us x 512 workgroups alone %.2f ms, B = 400 x 5 us x 200 workgroups alone %.2f ms; together: all done %.2f ms, B's chain done after %.2f ms\n", lds / 1024, ta, tb, tab, tb2);
  }
  return 0;
}
