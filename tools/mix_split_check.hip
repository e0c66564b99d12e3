// Bit-level check of the two ways to cut x * t (t a power of two) into fp16 hi + fp16 lo on gfx950:
//   A: what hipcc makes of  hi = (f16)(x*t); lo = (f16)fma(x, t, -(float)hi)      (v_pk_mul / v_cvt_pk / v_cvt_f32 / v_pk_fma / v_cvt_pk)
//   B: v_fma_mixlo/mixhi_f16 (fp32 fma, one rounding to fp16, fp16 third operand read in place): 2 instructions per value
// over random bit patterns, every exponent, fp16 subnormal results and values that round up to the next binade.
//   hipcc --offload-arch=gfx950 tools/mix_split_check.hip -o /tmp/mix_split_check && /tmp/mix_split_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__global__ void split_a(const float* x, float t, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  const _Float16 h0 = (_Float16)(a * t), h1 = (_Float16)(b * t);
  const _Float16 l0 = (_Float16)__builtin_fmaf(a, t, -(float)h0), l1 = (_Float16)__builtin_fmaf(b, t, -(float)h1);
  f16x2 H = {h0, h1}, L = {l0, l1};
  o[2 * i] = *reinterpret_cast<unsigned*>(&H);
  o[2 * i + 1] = *reinterpret_cast<unsigned*>(&L);
}
__global__ void split_b(const float* x, float t, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  unsigned hi = 0, lo = 0;
  const float z = 0.f;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3" : "+v"(hi) : "v"(a), "v"(t), "v"(z));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(hi) : "v"(b), "v"(t), "v"(z));
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(a), "v"(t), "v"(hi));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(t), "v"(hi));
  o[2 * i] = hi;
  o[2 * i + 1] = lo;
}

int main() {
  const int n = 1 << 22;
  std::vector<float> hx(2 * n);
  unsigned s = 12345u;
  for (int i = 0; i < 2 * n; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned bits = s;
    if ((i & 3) == 1) bits = (bits & 0x807FFFFFu) | ((100u + (s >> 9) % 40u) << 23);      // around the split's working range
    if ((i & 3) == 2) bits = (bits & 0x80000000u) | ((127u + (i >> 2) % 16u) << 23) | 0x7FF000u | (s & 0xFFFu);   // rounds up to the next binade
    union { unsigned u; float f; } c; c.u = bits;
    if (c.f != c.f) c.f = 1.5f;
    hx[i] = c.f;
  }
  float* dx; unsigned *da, *db;
  hipMalloc(&dx, 2 * n * 4); hipMalloc(&da, 2 * n * 4); hipMalloc(&db, 2 * n * 4);
  hipMemcpy(dx, hx.data(), 2 * n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> a(2 * n), b(2 * n);
  long long bad = 0;
  const float ts[] = {1.f, 0.5f, 1024.f, 1.f / 65536.f, 16384.f, 1.f / (1 << 20)};
  for (float t : ts) {
    split_a<<<n / 256, 256>>>(dx, t, da, n);
    split_b<<<n / 256, 256>>>(dx, t, db, n);
    hipMemcpy(a.data(), da, 2 * n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, 2 * n * 4, hipMemcpyDeviceToHost);
    long long d = 0;
    for (int i = 0; i < 2 * n; ++i)
      if (a[i] != b[i]) { if (d < 5) printf("t=%g i=%d x=(%a,%a) A=%08x B=%08x\n", t, i, hx[(i & ~1)], hx[(i & ~1) + 1], a[i], b[i]); ++d; }
    printf("t=%g: %lld of %d words differ\n", t, d, 2 * n);
    bad += d;
  }
  printf(bad ? "MISMATCH\n" : "IDENTICAL\n");
  return bad ? 1 : 0;
}
