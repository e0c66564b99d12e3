// Cost of a software grid-wide barrier on MI355X (tuning probe for a persistent multi-layer SR kernel; not product code).
// Every workgroup: L rounds of { agent-scope atomic add on a counter; spin (bounded) until all G workgroups have arrived }.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier_probe.hip -o gpurun_out/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_loop(unsigned* counter, int rounds, unsigned* fail, int payload_bytes, float* sink) {
  extern __shared__ float lds[];
  const unsigned G = gridDim.x;
  float acc = 0.f;
  for (int r = 1; r <= rounds; ++r) {
    // a little "work": touch LDS so the kernel is not empty
    for (int i = threadIdx.x; i < payload_bytes / 4; i += blockDim.x) lds[i] = (float)(i + r);
    __syncthreads();
    acc += lds[threadIdx.x % (payload_bytes / 4 > 0 ? payload_bytes / 4 : 1)];
    __threadfence();                                   // release this workgroup's global writes (none here) at agent scope
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = G * (unsigned)r;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > 2000000) { atomicAdd(fail, 1u); break; }   // bounded: never hangs
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  }
  if (acc == -1.f) sink[0] = acc;
}

int main() {
  unsigned *counter, *fail;
  float* sink;
  hipMalloc(&counter, 4); hipMalloc(&fail, 4); hipMalloc(&sink, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int G : {256, 512}) {
    for (int rounds : {1, 101}) {
      float best = 1e9f;
      unsigned hf = 0;
      for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(barrier_loop, dim3(G), dim3(256), 16384, 0, counter, rounds, fail, 4096, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
      }
      printf("G=%d rounds=%d: %.2f us total (timeouts: %u)\n", G, rounds, best * 1e3f, hf);
    }
  }
  return 0;
}
