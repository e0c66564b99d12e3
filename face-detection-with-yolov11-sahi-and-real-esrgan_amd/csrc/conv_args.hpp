// conv_args.hpp — kernel argument block and device helpers shared by the MFMA convolution kernels (conv_mfma.hip, conv_rows.hip).
#pragma once
#include "ops.hpp"

namespace ffp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
  const void* in;
  const void* wpk;
  const float* bias;
  void* out;
  const void* res1;
  const void* res2;
  const int4* in_tab;
  const int4* out_tab;
  const int4* tiles;
  long long total_px;   // KS == 1: flat pixel count
  int in_cs, in_coff, cin, cin_pad;
  int out_cs, out_coff, cout;
  int r1_cs, r1_coff, r2_cs, r2_coff;
  float s1, s2;
  int act, out_f32, up, n_nblk, ncg, ntiles32, vec_ok, fast_out;
  int force_shape;   // tuning only: -1 auto
  int dbg;   // tuning only (ffp_op_conv2d_time): 1 skip stores, 2 skip MFMAs, 4 skip chunk refetch, 8 skip LDS stash
  const void* up_src;  // 1x1 only: channels [0, up_c) come from this coarser view through up_map (see ConvOp::has_up2)
  const int* up_map;
  int up_c, up_cs;
  const void* zeros;   // 256 zero bytes in device memory (padding source of the LDS-DMA loader in conv_rows.hip)
  // scaled split (FFP_PREC_F32X3): the loader multiplies the activations by 2^(13 - e), e = exponent of max(*amax_in, *amax_in2),
  // before cutting them into fp16 hi + lo parts; the weights were packed with a per-output-channel power of two (oscale = its
  // inverse); the epilogue multiplies the sums back and raises *amax_out to the largest |output| it stores
  const unsigned* amax_in;
  const unsigned* amax_in2;
  unsigned* amax_out;
  const float* oscale;
  // per-image exponent slots (TView::amax_n > 1): slot of image i = amax_*[i]; amax_img = 0: one slot per buffer (index 0 for every image).
  // 3x3 kernels know the image of a tile; the 1x1 kernels look the image of a 32-pixel fragment of the flat pixel array up in frag_img
  // (levels aligned to 32 pixels: no fragment straddles two images)
  int amax_img;
  const int* frag_img;                 // [fragment] x {image, end of the image's real pixels}
  int ntiles_host;          // tile-loop kernels (conv_rows16.hip): tiles of an exact-mode launch (the grid no longer says)
  const int* n_tiles_dev;   // capacity-mode levels (Level::reserve): the batch's tile count lives in device memory and the grid is
                            // sized for the capacity — workgroups past n_tiles * n_nblk exit; nullptr: the grid is exact
  // stem-fused loader (conv_mfma_kernel<..., STEM = true>, launch_stem_conv): the conv's input is never stored — every workgroup
  // computes its halo tile of the image-input conv (model.0: 3x3 stride 2 + SiLU over the letterboxed u8 frame) on the matrix
  // cores and writes it straight into the LDS stage as split fp16 records
  const unsigned char* st_frame;     // HxWx3 u8
  const LetterboxImg* st_imgs;       // per item
  const int4* st_tab;                // level table of the letterboxed network images (.y rows, .z columns)
  const void* st_w;                  // stem weights as MFMA A fragments [chunk][hi|lo][64 lanes] x 16 B (stem_pack_kernel)
  const float* st_bias;
  int st_W;                          // frame width
  long long st_bytes;                // frame size in bytes, rounded up to whole dwords (range check of the loader's dword reads)
  float st_scale;                    // 1 / (power of two the packed stem weights were multiplied by)
};

// (logical workgroup id, number of live workgroups) for a tile kernel; live == 0: this workgroup has nothing to do
__device__ __forceinline__ int live_workgroups(const ConvArgs& a) {
  return a.n_tiles_dev ? __builtin_amdgcn_readfirstlane(*a.n_tiles_dev) * a.n_nblk : (int)gridDim.x;
}

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  // blocks b and b+8 share an XCD (observed round-robin dispatch): give each XCD a contiguous run of logical ids.
  const int xcd = b & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// wave-uniform 32-bit load through the scalar cache: read-only tables and max-|value| slots written by EARLIER kernels. (Inside a
// loop with stores hipcc turns a uniform global load into a vector load + readfirstlane, i.e. a full memory round trip on the
// critical path; the constant address space keeps it an s_load whose result nothing waits for until it is used.)
__device__ __forceinline__ int sload(const void* p, int idx) {
  typedef const __attribute__((address_space(4))) int* cptr;
  return reinterpret_cast<cptr>(reinterpret_cast<unsigned long long>(p))[idx];
}

// wave-wide maximum of non-negative, non-NaN floats as a wave-uniform bit pattern: four DPP steps inside the rows of 16 lanes,
// then the four rows by v_readlane (no LDS round trips, unlike __shfl_xor)
__device__ __forceinline__ unsigned wave_max_bits(float mx) {
  int v = __float_as_int(mx);
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));    // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));    // row_mirror
  const int a = max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16));
  const int b = max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48));
  return (unsigned)max(a, b);
}

// raise a max-|value| slot (see TView::amax): wave-wide maximum first, and the atomic only when it would change the slot.
// `old` = a value the slot held at some earlier time (slots only grow, so a stale value costs at most a redundant atomic): kernels that
// raise a slot per work item load it early so that nothing waits for the round trip here.
__device__ __forceinline__ unsigned slot_peek(const unsigned* slot) {
  return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void raise_amax(unsigned* slot, float mx, unsigned old) {
  const unsigned b = wave_max_bits(mx);
  if ((threadIdx.x & 63) == 0 && b > old) atomicMax(slot, b);
}
__device__ __forceinline__ void raise_amax(unsigned* slot, float mx) {
  const unsigned b = wave_max_bits(mx);
  if ((threadIdx.x & 63) == 0 && b > slot_peek(slot)) atomicMax(slot, b);
}

// bit pattern of max |value| of the conv's input for image `img` (both sources of a two-source 1x1)
__device__ __forceinline__ unsigned amax_in_bits(const ConvArgs& a, int img) {
  if (!a.amax_in) return 0u;
  const int i = a.amax_img ? img : 0;
  unsigned m = a.amax_in[i];
  if (a.amax_in2) m = max(m, a.amax_in2[i]);
  return m;
}
// the same through the scalar cache (img wave-uniform)
__device__ __forceinline__ unsigned amax_in_bits_s(const ConvArgs& a, int img) {
  if (!a.amax_in) return 0u;
  const int i = a.amax_img ? img : 0;
  unsigned m = (unsigned)sload(a.amax_in, i);
  if (a.amax_in2) m = max(m, (unsigned)sload(a.amax_in2, i));
  return m;
}

// activation scale of the split and its inverse from a max-|value| bit pattern: 2^(13 - e) puts the largest value in [2^13, 2^14)
__device__ __forceinline__ void split_scales(unsigned amax_bits, float* t, float* tinv) {
  const unsigned eb = (amax_bits >> 23) & 0xFFu;
  const bool ok = eb >= 14u && eb < 255u;         // zero / denormal-tiny / inf / nan tensors: unscaled
  *t = ok ? __uint_as_float((267u - eb) << 23) : 1.f;
  *tinv = ok ? __uint_as_float((eb - 13u) << 23) : 1.f;
}

// (x0, x1) * t -> packed fp16 pair hi = RN(x * t) and packed fp16 pair lo = RN(x * t - hi), t a power of two: the operand split of
// FFP_PREC_F32X3. v_fma_mix{lo,hi}_f16 compute an fp32 fma and round once to fp16, reading the fp16 third operand in place: two
// instructions per value, against four for multiply / convert / convert back / subtract / convert. Bit-identical to
//   hi = (_Float16)(x * t); lo = (_Float16)fmaf(x, t, -(float)hi)
// on every input (tools/mix_split_check.hip compares 50 M values over all exponents, fp16 overflow and subnormal results).
__device__ __forceinline__ void split_pair(float x0, float x1, float t, unsigned& hi, unsigned& lo) {
  unsigned h, l;                        // mixlo leaves the upper half of its destination as it was; mixhi then writes it
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(t));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(t));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(t), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(t), "v"(h));
  hi = h;
  lo = l;
}

__device__ __forceinline__ float apply_act(float v, int act) {
  // SiLU with the hardware exp2 / rcp (1 ulp each): ~6 VALU ops instead of ~25 for expf + IEEE divide; the epilogue of a
  // 256 px x 64 ch block otherwise spends ~2.7 us in the activation alone
  if (act == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  if (act == ACT_LRELU) return v >= 0.f ? v : v * 0.2f;
  return v;
}

// host side (conv_mfma.hip)
ConvArgs make_conv_args(const ConvOp& op);
// k3 s1 fp16 row-reuse kernel (conv_rows.hip)
bool conv_rows_eligible(const ConvOp& op, const ConvArgs& a);
void launch_conv_rows(ConvArgs& a, Level* out_lvl, hipStream_t st);
void conv_rows_init();
// k3 s1 fp16, second generation on v_mfma_f32_16x16x32_f16 (conv_rows16.hip)
bool conv_rows16_eligible(const ConvOp& op, const ConvArgs& a);
void launch_conv_rows16(ConvArgs& a, const PackedConv& pc, Level* out_lvl, hipStream_t st);
void conv_rows16_init();
bool conv_rows16_enabled();     // FFP_ROWS16=0 keeps the first-generation kernel (A/B aid)
// the same convs with producer / consumer waves (conv_rows16pc.hip): OPT-IN (force_shape 24 or FFP_ROWS16_PC=1; measured 0.73-0.94x); bit-identical results
bool conv_rows16pc_selected(const ConvArgs& a);
void launch_conv_rows16pc(ConvArgs& a, const PackedConv& pc, Level* out_lvl, hipStream_t st);
void conv_rows16pc_init();
// 1x1 fp32-split convs without LDS staging of the activations (conv_pw.hip): bit s of the mask = force_shape s (10..16) can run this op
unsigned conv_pw_mask(const ConvOp& op, const ConvArgs& a);
void launch_conv_pw(ConvArgs& a, int shape, hipStream_t st);
void conv_pw_init();
bool conv_pw_enabled();         // FFP_PW=0 keeps the tuner to the generic shapes (A/B aid)

// 3x3 fp32-split convs, stride 1 and 2, weights straight from L2 into MFMA operand registers (conv_k3d.hip): bit s of the mask =
// force_shape s (17..21) can run this op; results are bit-identical to the generic kernel's
unsigned conv_k3d_mask(const ConvOp& op, const ConvArgs& a);
void launch_conv_k3d(ConvArgs& a, int shape, int stride, Level* out_lvl, hipStream_t st);
void conv_k3d_init();
bool conv_k3d_enabled();        // FFP_K3D=0 keeps the tuner to the generic shapes (A/B aid)

}  // namespace ffp
