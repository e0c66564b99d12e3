// conv_mfma.hip — dense convolution as an im2col-free implicit GEMM on the gfx950 matrix cores.
//
// One launch convolves a whole ragged batch (images of different sizes stored back to back, NHWC).
//   D[out channel][pixel] += W[out channel][tap, c] * X[tap-shifted pixel][c]
// The MFMA A operand is a pre-packed weight fragment (32 out channels x one k-group, weights.cpp), the B operand a
// pixel fragment (32 pixels = 2 tile rows x 16 columns, k contiguous) read from an LDS-staged halo tile, so the
// accumulator holds, per lane, ONE pixel and 16 out channels in groups of 4 consecutive channels -> 16-byte
// (fp32) / 8-byte (fp16) epilogue stores.
//   fp16: v_mfma_f32_32x32x16_f16, k-group = 16 channels      (FFP_PREC_F16)
//   fp32: 4 x v_mfma_f32_32x32x2_f32 per 8-channel k-group      (FFP_PREC_F32, exact-f32 fmaf chain)
// Workgroup = 256 threads = 4 waves laid out WM x WN; a wave owns MI pixel fragments x NIW 32-channel tiles.
// blockIdx is remapped so that the n-blocks of one pixel tile and neighbouring tiles share an XCD (L2 reuse of the
// input tile); the mapping only affects speed.
#include <cstdlib>
#include <type_traits>

#include "conv_args.hpp"
#include "trunk.hpp"
#include "letterbox.hpp"

#ifndef FFP_SINGLE_STAGE
#define FFP_SINGLE_STAGE 1     // tuning switch: 0 = always double-buffer the LDS stage
#endif
#ifndef FFP_OCC3
#define FFP_OCC3 1           // tuning switch: 0 = small-accumulator split shapes stay at two workgroups per CU with two LDS stages
#endif
#ifndef FFP_DEEP_OCC1
#define FFP_DEEP_OCC1 0      // tuning switch: 1 = shapes whose LDS allows one workgroup per CU get 512 registers and a deeper prefetch ring (measured: no gain)
#endif

namespace ffp {

// fp32 storage with operands split into fp16 hi + lo parts: a*b ~= ah*bh + ah*bl + al*bh (three fp16 MFMAs, fp32
// accumulate; the dropped al*bl term is 2^-22 relative) — fp32-grade products at 3/16 of the exact-fp32 MFMA time.
struct X3 {};

// arithmetic traits: GT = element type in HBM, KG = input channels per MFMA k-group, LDS_EB = LDS bytes per element,
// WFRAG = bytes of one packed weight fragment (32 out channels x one k-group)
template <typename T> struct Ar;
template <> struct Ar<float> { using GT = float; static constexpr int KG = 8, LDS_EB = 4, WFRAG = 1024; };
template <> struct Ar<_Float16> { using GT = _Float16; static constexpr int KG = 16, LDS_EB = 2, WFRAG = 1024; };
template <> struct Ar<X3> { using GT = float; static constexpr int KG = 16, LDS_EB = 4, WFRAG = 2048; };

template <typename T> struct MM;
template <> struct MM<float> {
  static constexpr int KG = 8;
  static __device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};
template <> struct MM<_Float16> {
  static constexpr int KG = 16;
  static __device__ __forceinline__ void mma(f32x16& acc, const uint4& a, const uint4& b) {
    union { uint4 u; f16x8 h; } ua, ub;
    ua.u = a; ub.u = b;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc, 0, 0, 0);
  }
};

template <> struct MM<X3> {
  static constexpr int KG = 16;
  static __device__ __forceinline__ void mma3(f32x16& acc, const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl) {
    union { uint4 u; f16x8 h; } uah, ual, ubh, ubl;
    uah.u = ah; ual.u = al; ubh.u = bh; ubl.u = bl;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ual.h, ubh.h, acc, 0, 0, 0);     // small terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uah.h, ubl.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(uah.h, ubh.h, acc, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&r)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&r)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
}
template <> __device__ __forceinline__ void load4<_Float16>(const _Float16* p, float (&r)[4]) {
  union { uint2 u; _Float16 h[4]; } v;
  v.u = *reinterpret_cast<const uint2*>(p);
  r[0] = (float)v.h[0]; r[1] = (float)v.h[1]; r[2] = (float)v.h[2]; r[3] = (float)v.h[3];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&r)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&r)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
}
template <> __device__ __forceinline__ void store4<_Float16>(_Float16* p, const float (&r)[4]) {
  union { uint2 u; _Float16 h[4]; } v;
  v.h[0] = (_Float16)r[0]; v.h[1] = (_Float16)r[1]; v.h[2] = (_Float16)r[2]; v.h[3] = (_Float16)r[3];
  *reinterpret_cast<uint2*>(p) = v.u;
}

// compile-time geometry shared by the kernel and its launcher
template <typename T, int KS, int STRIDE, int WM, int WN, int MI, int NIW, int KC> struct Geo {
  using GT = typename Ar<T>::GT;
  static constexpr int ES = sizeof(GT);                      // bytes per element in HBM
  static constexpr int KG = Ar<T>::KG;
  static constexpr int WFRAG = Ar<T>::WFRAG;
  static constexpr int KCG = KC / KG;                       // k-groups per chunk
  static constexpr int FR = WM * MI;                        // pixel fragments per workgroup (32 px each)
  static constexpr int TH = FR * 2;                         // output tile rows (16 columns)
  static constexpr int HH = KS == 1 ? 1 : (TH - 1) * STRIDE + KS;
  static constexpr int HW = KS == 1 ? FR * 32 : 15 * STRIDE + KS;
  static constexpr int NPIX = HH * HW;                      // staged input pixels
  static constexpr int PS = KC * Ar<T>::LDS_EB + 16;        // LDS bytes per pixel record (+1 b128 pad against bank conflicts)
  static constexpr int VPP = KC * ES / 16;                  // 16-byte global vectors per pixel per chunk
  static constexpr int EPV = 16 / ES;                       // elements per vector
  static constexpr int TAPS = KS * KS;
  static constexpr int NTB = WN * NIW;                      // 32-channel tiles per workgroup
  static constexpr int IN_BYTES = (NPIX * PS + 15) / 16 * 16;
  static constexpr int W_FRAGS = NTB * TAPS * KCG;          // weight fragments per chunk
  static constexpr int VPF = WFRAG / 16;                    // 16-byte vectors per weight fragment
  static constexpr int BUF = IN_BYTES + W_FRAGS * WFRAG;    // one LDS stage: input tile chunk + its weights
  // Two stages (chunk c+1 is written while chunk c is multiplied) where both fit 80 KiB, i.e. two workgroups per CU. A
  // shape whose single stage still fits 80 KiB runs ONE stage (two barriers per chunk, no overlap inside the workgroup)
  // and lets the second resident workgroup fill the CU instead: with one workgroup per CU its fetch, MFMA and store
  // phases simply add up (measured on the fp32-split 3x3 stride-2 layers).
  // Small-accumulator shapes (one 32-channel tile per wave) whose single stage fits a third of the LDS run THREE workgroups per CU
  // on one stage each: their fetch, MFMA and store phases do not overlap inside a workgroup (measured: the phase times add up,
  // profiles/r02_k3_phase_probe.txt), so residency is what overlaps them.
  static constexpr bool OCC3 = FFP_OCC3 && NIW == 1 && MI <= 2 && 3 * BUF <= 160 * 1024 && std::is_same<T, X3>::value;
  static constexpr int STAGES = OCC3 ? 1 : (2 * BUF <= 80 * 1024 || BUF > 80 * 1024 || !FFP_SINGLE_STAGE) ? 2 : 1;
  static constexpr int LDS = STAGES * BUF;
  static constexpr int NVI = NPIX * VPP, NVW = W_FRAGS * VPF;
  static constexpr int RI = (NVI + 255) / 256, RW = (NVW + 255) / 256;   // prefetch registers (16 B each) per thread and chunk
  // chunks kept in flight global->registers ahead of the one being multiplied: as many (<= 4) as fit a 2-waves-per-SIMD
  // register budget next to the accumulators. Short chunks (0.25 us of MFMA at KC = 16) need several L2 round trips in flight.
  // LDS->register fragment ring of the MFMA loop: a step = one (k-group, tap) = FPS 16-byte fragments; PD steps of lookahead
  static constexpr int FPS = (MI + NIW) * (std::is_same<T, X3>::value ? 2 : 1);
  static constexpr int NS = KCG * TAPS;
  static constexpr int PD_RAW = NIW * MI >= 8 ? 0 : 12 / FPS - 1;     // 128 accumulator registers leave no room for a ring
  static constexpr int PD_CL = PD_RAW < 0 ? 0 : PD_RAW > 4 ? 4 : PD_RAW;
  static constexpr int PD = PD_CL > NS - 1 ? NS - 1 : PD_CL;
  static constexpr int FRAG_REGS = 4 * (PD + 1) * FPS;
  // a shape whose LDS footprint allows one workgroup per CU anyway gets the whole 512-register file (launch bound 1 wave
  // per SIMD) and spends it on prefetch depth: with D = 1 every chunk of an HBM-fed layer exposes a full memory round trip
  static constexpr int OCC = OCC3 ? 3 : (LDS <= 80 * 1024 || !FFP_DEEP_OCC1) ? 2 : 1;
  static constexpr int REG_BUDGET = OCC == 3 ? 168 : OCC == 2 ? 256 : 448;
  static constexpr int DEPTH_RAW = (REG_BUDGET - NIW * MI * 16 - 56 - FRAG_REGS - 3 * (RI + RW)) / (4 * (RI + RW));   // accumulators, fragment ring, ~56 misc, address slots
  static constexpr int DEPTH = DEPTH_RAW < 1 ? 1 : DEPTH_RAW > 4 ? 4 : DEPTH_RAW;
  static_assert(KC % KG == 0 && WM * WN == 4, "geometry");
};

// Main loop: per chunk of KC input channels the input halo tile AND the chunk's weight fragments live in one LDS stage;
// chunk c+1 is fetched global->registers while chunk c is multiplied out of LDS (no global access inside the MFMA
// loop), then written to the other stage: one barrier per chunk.
// STEM = true (one instantiation, launch_stem_conv): the input tensor does not exist — the loader computes the halo tile of the
// image-input conv from the u8 frame (see ConvArgs::st_frame) and writes the split records itself.
template <typename T, int KS, int STRIDE, int WM, int WN, int MI, int NIW, int KC, bool STEM = false>
__global__ void __launch_bounds__(256, (Geo<T, KS, STRIDE, WM, WN, MI, NIW, KC>::OCC)) conv_mfma_kernel(const ConvArgs a) {
  using G = Geo<T, KS, STRIDE, WM, WN, MI, NIW, KC>;
  using GT = typename G::GT;
  constexpr bool SPLIT = std::is_same<T, X3>::value;
  constexpr int ES = G::ES, KG = G::KG, KCG = G::KCG, HW = G::HW, NPIX = G::NPIX, PS = G::PS, VPP = G::VPP, EPV = G::EPV;
  constexpr int TAPS = G::TAPS, NTB = G::NTB, RI = G::RI, RW = G::RW, WFRAG = G::WFRAG, VPF = G::VPF;
  constexpr int LKB = KG * Ar<T>::LDS_EB / (SPLIT ? 2 : 1);   // LDS bytes one k-group spans inside the (hi) block of a pixel record
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 31, hh = lane >> 5;

  const int nwg = KS == 3 ? live_workgroups(a) : (int)gridDim.x;
  if ((int)blockIdx.x >= nwg) return;                  // capacity-sized grid, smaller batch
  const int L = xcd_remap(blockIdx.x, nwg);
  const int nblk = L % a.n_nblk, tile = L / a.n_nblk;

  int oy0 = 0, ox0 = 0, Ho = 1, Wo = 0, Hi = 0, Wi = 0;
  long long in_base, out_base;
  if (KS == 3) {
    const int4 t = a.tiles[tile];
    const int4 it = a.in_tab[t.x], ot = a.out_tab[t.x];
    oy0 = t.y; ox0 = t.z;
    in_base = it.x; Hi = it.y; Wi = it.z;
    out_base = ot.x; Ho = ot.y; Wo = ot.z;
  } else {
    in_base = out_base = (long long)tile * NPIX;
  }
  const int iy0 = oy0 * STRIDE - (KS / 2), ix0 = ox0 * STRIDE - (KS / 2);
  const int Hv = Hi << a.up, Wv = Wi << a.up;      // input size seen through the optional nearest x2

  int boff[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    const int pix = KS == 1 ? f * 32 + p : ((2 * f + (p >> 4)) * STRIDE) * HW + (p & 15) * STRIDE;
    boff[mi] = pix * PS + hh * 16;
  }
  const int ntile0 = (nblk * WN + wn) * NIW;
  const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in);
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wpk);

  // Global operands are read with raw BUFFER loads: the resource base is wave-uniform (this workgroup's image / pixel run,
  // the packed weights), the per-lane part a 32-bit byte offset, and an out-of-image / past-the-end slot uses offset
  // 0xFFFFFFFF, which the hardware range check turns into zeros — conv zero padding without a branch or a select, so the
  // loads stay straight-line code and hipcc retires them with counted s_waitcnt vmcnt(N).
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto uniform_ptr = [](const unsigned char* q) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(inb) + ((long long)in_base * a.in_cs + a.in_coff) * ES), 0, 0x7FFFFFF0, 0x00020000);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(wb)), 0, 0x7FFFFFF0, 0x00020000);
  auto bload = [](decltype(rs_in) rs, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };

  // scaled split: power of two of the activations (max |value| of the input buffer(s) -> [2^13, 2^14) in fp16), per buffer or — with
  // per-image exponent slots — per image: a 3x3 workgroup works on ONE image (its tile's), a 1x1 workgroup on FR 32-pixel fragments
  // of the flat pixel array, each inside one image (levels aligned to 32 pixels), so the scale is a property of the fragment
  float tscale = 1.f, tinv = 1.f;
  int img3 = 0;                                         // KS == 3: the tile's image
  if constexpr (KS == 3) img3 = __builtin_amdgcn_readfirstlane(a.tiles[tile].x);
  // KS == 1: {image, end of its real pixels, scale, inverse scale} of the FR fragments of this workgroup's pixel block, wave-uniform, read
  // through the scalar cache (the table entries of a block are consecutive)
  int f_img[G::FR], f_end[G::FR];
  float f_ts[G::FR], f_ti[G::FR];
#pragma unroll
  for (int f = 0; f < G::FR; ++f) { f_img[f] = 0; f_end[f] = 0; f_ts[f] = 1.f; f_ti[f] = 1.f; }
  if constexpr (SPLIT) {
    if (a.amax_in && KS == 3) split_scales(amax_in_bits_s(a, img3), &tscale, &tinv);
    if (a.amax_in && KS == 1 && !a.amax_img) split_scales(amax_in_bits_s(a, 0), &tscale, &tinv);
    if constexpr (KS == 1) {
      if (a.amax_img) {
        const int nf = (int)(a.total_px >> 5), fb = __builtin_amdgcn_readfirstlane((int)(in_base >> 5));
#pragma unroll
        for (int f = 0; f < G::FR; ++f) {
          const int e = 2 * min(fb + f, nf - 1);
          f_img[f] = sload(a.frag_img, e);
          f_end[f] = sload(a.frag_img, e + 1);
        }
#pragma unroll
        for (int f = 0; f < G::FR; ++f)
          if (a.amax_in) split_scales(amax_in_bits_s(a, f_img[f]), &f_ts[f], &f_ti[f]);
      }
    }
  }

  // per-thread staging slots: which vector each of this thread's RI + RW registers carries (chunk independent part)
  unsigned isrc[RI];           // byte offset of the pixel record relative to the resource base (OOB: zero fill)
  int ivec[RI];                // vector index inside the chunk
  float tsc[RI];               // split scale of the slot's pixel (per-image exponent slots: the scale of its fragment's image)
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int idx = tid + i * 256;
    isrc[i] = OOB; ivec[i] = 0; tsc[i] = tscale;
    if (!STEM && idx < G::NVI) {
      const int hp = idx / VPP;
      if constexpr (SPLIT && KS == 1) {
        if (a.amax_img && a.amax_in) {
#pragma unroll
          for (int f = 0; f < G::FR; ++f) if ((hp >> 5) == f) tsc[i] = f_ts[f];
        }
      }
      ivec[i] = idx % VPP;
      long long rel;
      bool ok;
      if (KS == 1) {
        rel = hp;
        ok = in_base + hp < a.total_px;
      } else {
        const int hy = hp / HW, hx = hp - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
        rel = (long long)(iy >> a.up) * Wi + (ix >> a.up);
      }
      if (ok) isrc[i] = (unsigned)(rel * a.in_cs * ES);
    }
  }
  // virtual concat [nearest_x2(coarser view) | rest] (1x1 only): for such a conv isrc[] carries the offsets into the coarser
  // view (through the x2 pixel map) and the offsets into `in` are recomputed per load (one multiply), so that no shape
  // pays registers for the second source
  const unsigned char* upb = a.up_c > 0 ? reinterpret_cast<const unsigned char*>(a.up_src) : wb;
  const auto rs_up = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(upb)), 0, 0x7FFFFFF0, 0x00020000);
  if constexpr (KS == 1) {
    if (a.up_c > 0) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
        if (isrc[i] != OOB) isrc[i] = (unsigned)a.up_map[in_base + (tid + i * 256) / VPP] * (unsigned)(a.up_cs * ES);
    }
  }
  unsigned wsrc[RW];           // byte offset of the fragment row start for k-group 0 (OOB: no such slot)
  int wkg[RW];
#pragma unroll
  for (int i = 0; i < RW; ++i) {
    const int idx = tid + i * 256;
    wsrc[i] = OOB; wkg[i] = 0;
    if (idx < G::NVW) {
      const int fl = idx / VPF, l = idx % VPF;
      const int kg = fl % KCG, tap = (fl / KCG) % TAPS, ntl = fl / (KCG * TAPS);
      const int nt = min(nblk * NTB + ntl, a.ntiles32 - 1);
      wkg[i] = kg;
      wsrc[i] = (unsigned)(((long long)(nt * TAPS + tap) * a.ncg) * WFRAG + l * 16);
    }
  }

  constexpr int D = G::DEPTH;
  uint4 ri[D][RI], rw[D][RW];                      // D chunks in flight (static indexing: every loop over D is unrolled)
  auto fetch = [&](int c0, uint4 (&qi)[RI], uint4 (&qw)[RW]) {
    const bool from_up = KS == 1 && c0 < a.up_c;          // chunk-uniform: up_c is a multiple of every chunk size
#pragma unroll
    for (int i = 0; i < (STEM ? 0 : RI); ++i) {
      const int c = c0 + ivec[i] * EPV;
      if constexpr (KS == 1) {
        unsigned so = isrc[i];
        if (a.up_c > 0 && !from_up && so != OOB) so = (unsigned)(((tid + i * 256) / VPP) * a.in_cs * ES);
        qi[i] = bload(from_up ? rs_up : rs_in, (so != OOB && c < a.cin) ? so + (unsigned)(c * ES) : OOB);
      } else {
        qi[i] = bload(rs_in, (isrc[i] != OOB && c < a.cin) ? isrc[i] + (unsigned)(c * ES) : OOB);
      }
    }
    const int cg0 = c0 / KG;
#pragma unroll
    for (int i = 0; i < RW; ++i)
      qw[i] = bload(rs_w, (wsrc[i] != OOB && cg0 + wkg[i] < a.ncg) ? wsrc[i] + (unsigned)((cg0 + wkg[i]) * WFRAG) : OOB);
  };
  auto stash = [&](unsigned char* buf, const uint4 (&qi)[RI], const uint4 (&qw)[RW]) {
#pragma unroll
    for (int i = 0; i < (STEM ? 0 : RI); ++i) {
      const int idx = tid + i * 256;
      if (idx < G::NVI) {
        if (SPLIT) {   // 4 floats -> 4 fp16 hi parts + 4 fp16 residuals, stored in the hi / lo halves of the pixel record
          uint2 hi, lo;
          split_pair(__uint_as_float(qi[i].x), __uint_as_float(qi[i].y), tsc[i], hi.x, lo.x);
          split_pair(__uint_as_float(qi[i].z), __uint_as_float(qi[i].w), tsc[i], hi.y, lo.y);
          unsigned char* rec = buf + (idx / VPP) * PS + (idx % VPP) * 8;
          *reinterpret_cast<uint2*>(rec) = hi;
          *reinterpret_cast<uint2*>(rec + KC * 2) = lo;
        } else {
          *reinterpret_cast<uint4*>(buf + (idx / VPP) * PS + (idx % VPP) * 16) = qi[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int idx = tid + i * 256;
      if (idx < G::NVW) *reinterpret_cast<uint4*>(buf + G::IN_BYTES + idx * 16) = qw[i];
    }
  };

  // ---- stem-fused loader. The halo tile's NPIX pixels are taken in groups of 16 (SGW groups per wave); a group is ONE B operand
  // of v_mfma_f32_16x16x32_f16: lane (n = lane % 16, kb = lane / 16) holds 8 of the 27 (+5 zero) taps x channels of the stem's 3x3
  // stride-2 window around pixel n — kb < 3: window row kb, bytes 0..7 of its 9; kb = 3: byte 8 of the three rows — as fp16, which
  // holds a u8 exactly. The / 255 sits in the weights (packed hi + lo fp16, stem_pack_kernel), so two MFMAs per 16 channels give
  // the fp32-grade sum; lane (n, g = kb) ends with channels 4g..4g+3 of pixel n: bias, SiLU, zero outside the stem map (the conv's
  // own zero padding), then the same split records the fp32 loader writes. Chunk 0 goes to the stage, chunk 1 waits in registers.
  constexpr int SG = (NPIX + 15) / 16, SGW = (SG + 3) / 4;
  constexpr int SCH = NIW;                       // 16-channel chunks of the stem output = input chunks of this conv (YOLO11s: 2 with 64 outputs, YOLO11n: 1 with 32; launch_stem_conv checks)
  uint2 s1h[STEM && SCH == 2 ? SGW : 1], s1l[STEM && SCH == 2 ? SGW : 1];
  auto stem_fill = [&](unsigned char* buf) {
    if constexpr (STEM) {
      static_assert(!STEM || (KC == 16 && SPLIT && KS == 3 && STRIDE == 2), "stem-fused loader: split 3x3 stride 2, 16-channel chunks");
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      const int n = lane & 15, kb = lane >> 4;
      const int item = a.tiles[tile].x;
      const LetterboxImg Lb = a.st_imgs[item];
      const int4 nt = a.st_tab[item];
      const int Hn = nt.y, Wn = nt.z;
      const uint4* swp = reinterpret_cast<const uint4*>(a.st_w) + lane;
      union FU { uint4 u; f16x8 h; };
      static_assert(!STEM || SCH == 1 || SCH == 2, "stem-fused loader: one or two 16-channel chunks");
      FU afr[SCH][2];
      float4 sb[SCH];
#pragma unroll
      for (int c = 0; c < SCH; ++c) {
#pragma unroll
        for (int part = 0; part < 2; ++part) afr[c][part].u = swp[(c * 2 + part) * 64];
        sb[c] = *reinterpret_cast<const float4*>(a.st_bias + 16 * c + 4 * kb);
      }
      // B operands of all groups first (their loads in flight together), then the arithmetic
      uint4 bq[SGW];
      const f16x2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
      auto to_half = [&](unsigned two_bytes_spread) {       // (b0 | b1 << 16) -> two fp16: 0x6400 | v is 1024 + v, exactly
        union { unsigned u; f16x2 h; } t;
        t.u = 0x64006400u | two_bytes_spread;
        t.h = t.h - k1024;
        return t.u;
      };
      auto geom = [&](int j, int& q, int& sy, int& sx) {
        q = (wave * SGW + j) * 16 + n;
        const int hy = q / HW, hx = q - hy * HW;
        sy = iy0 + hy; sx = ix0 + hx;
        return q < NPIX && (unsigned)sy < (unsigned)Hi && (unsigned)sx < (unsigned)Wi;
      };
      // item-uniform: the network image IS the source crop (no resize, no padding) — SAHI slices at their native size
      const bool plain = __builtin_amdgcn_readfirstlane((int)(Lb.new_w == Lb.sw && Lb.new_h == Lb.sh && Lb.top == 0 && Lb.left == 0)) != 0;
      if (plain) {
        // lane (n, kb) reads the 9 bytes of window row min(kb, 2) as three aligned dwords through a range-checked resource (a dword
        // wholly outside the frame reads zero); kb = 3 takes byte 8 of the three rows from its neighbours' registers
        // (the resource starts at the item's first frame row, so a stack of frames of any size stays within its 32-bit offsets)
        const long long row_b = (long long)Lb.y0 * a.st_W * 3;
        const long long row0 = row_b & ~3ll;                 // dword-aligned like the frame itself
        const int skew = (int)(row_b - row0);
        const long long left = a.st_bytes - row0;
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(a.st_frame) + row0), 0,
                                                            (int)(left < 0x7FFFFFFCll ? left : 0x7FFFFFFCll), 0x00020000);
        const int r = kb < 3 ? kb : 2;
        unsigned w0[SGW], w1[SGW], w2[SGW];
#pragma unroll
        for (int j = 0; j < SGW; ++j) {
          int q, sy, sx;
          const bool inmap = geom(j, q, sy, sx);
          const int ny = 2 * sy - 1 + r;
          const int A = (ny * a.st_W + Lb.x0 + 2 * sx - 1) * 3 + skew;
          const unsigned base = (inmap && (unsigned)ny < (unsigned)Hn) ? (unsigned)(A & ~3) : OOB;
          w0[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, base, 0, 0);
          w1[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, base == OOB ? OOB : base + 4u, 0, 0);
          w2[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_f, base == OOB ? OOB : base + 8u, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < SGW; ++j) {
          int q, sy, sx;
          const bool inmap = geom(j, q, sy, sx);
          const int ny = 2 * sy - 1 + r;
          const bool rowok = inmap && (unsigned)ny < (unsigned)Hn;
          const unsigned al = (unsigned)((ny * a.st_W + Lb.x0 + 2 * sx - 1) * 3 + skew) & 3u;
          unsigned d0 = __builtin_amdgcn_alignbyte(w1[j], w0[j], al), d1 = __builtin_amdgcn_alignbyte(w2[j], w1[j], al);
          unsigned b8 = __builtin_amdgcn_alignbyte(0u, w2[j], al) & 0xFFu;
          if (sx == 0) d0 &= 0xFF000000u;                    // window column -1: the stem's own zero padding
          if (!rowok) { d0 = 0u; d1 = 0u; b8 = 0u; }
          const unsigned t0 = (unsigned)__shfl((int)b8, n), t1 = (unsigned)__shfl((int)b8, n + 16), t2 = (unsigned)__shfl((int)b8, n + 32);
          if (kb == 3) { d0 = t0 | (t1 << 8) | (t2 << 16); d1 = 0u; }
          bq[j] = make_uint4(to_half(__builtin_amdgcn_perm(0u, d0, 0x0c010c00u)), to_half(__builtin_amdgcn_perm(0u, d0, 0x0c030c02u)),
                             to_half(__builtin_amdgcn_perm(0u, d1, 0x0c010c00u)), to_half(__builtin_amdgcn_perm(0u, d1, 0x0c030c02u)));
        }
      } else {
#pragma unroll 1
        for (int j = 0; j < SGW; ++j) {
          int q, sy, sx;
          const bool inmap = geom(j, q, sy, sx);
          int v[3][3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int rr = kb < 3 ? kb : t, c = kb < 3 ? t : 2;
            const int ny = 2 * sy - 1 + rr, nx = 2 * sx - 1 + c;
            v[t][0] = v[t][1] = v[t][2] = 0;
            if (inmap && (unsigned)ny < (unsigned)Hn && (unsigned)nx < (unsigned)Wn) letterbox_sample(a.st_frame, a.st_W, Lb, ny, nx, v[t]);
          }
          const bool row = kb < 3;
          const unsigned e0 = row ? v[0][0] : v[0][2], e1 = row ? v[0][1] : v[1][2], e2 = row ? v[0][2] : v[2][2];
          const unsigned e3 = row ? v[1][0] : 0, e4 = row ? v[1][1] : 0, e5 = row ? v[1][2] : 0, e6 = row ? v[2][0] : 0, e7 = row ? v[2][1] : 0;
          const uint4 bj = make_uint4(to_half(e0 | (e1 << 16)), to_half(e2 | (e3 << 16)), to_half(e4 | (e5 << 16)), to_half(e6 | (e7 << 16)));
          // static register indexing only: bq[] lives in registers
#pragma unroll
          for (int jj = 0; jj < SGW; ++jj) if (jj == j) bq[jj] = bj;
        }
      }
#pragma unroll
      for (int j = 0; j < SGW; ++j) {
        int q, sy, sx;
        const bool inmap = geom(j, q, sy, sx);
        FU bfr;
        bfr.u = bq[j];
#pragma unroll
        for (int c = 0; c < SCH; ++c) {
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[c][0].h, bfr.h, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_f16(afr[c][1].h, bfr.h, d, 0, 0, 0);
          float o0 = apply_act(fmaf(d[0], a.st_scale, sb[c].x), ACT_SILU), o1 = apply_act(fmaf(d[1], a.st_scale, sb[c].y), ACT_SILU);
          float o2 = apply_act(fmaf(d[2], a.st_scale, sb[c].z), ACT_SILU), o3 = apply_act(fmaf(d[3], a.st_scale, sb[c].w), ACT_SILU);
          if (!inmap) o0 = o1 = o2 = o3 = 0.f;
          uint2 hi, lo;
          split_pair(o0, o1, tscale, hi.x, lo.x);
          split_pair(o2, o3, tscale, hi.y, lo.y);
          if (c == 0) {
            if (q < NPIX) {
              unsigned char* rec = buf + q * PS + kb * 8;
              *reinterpret_cast<uint2*>(rec) = hi;
              *reinterpret_cast<uint2*>(rec + KC * 2) = lo;
            }
          } else if constexpr (SCH == 2) {
            s1h[j] = hi; s1l[j] = lo;
          }
        }
      }
    }
  };
  auto stem_flush = [&](unsigned char* buf) {
    if constexpr (STEM && SCH == 2) {
      const int n = lane & 15, kb = lane >> 4;
#pragma unroll
      for (int j = 0; j < SGW; ++j) {
        const int q = (wave * SGW + j) * 16 + n;
        if (q < NPIX) {
          unsigned char* rec = buf + q * PS + kb * 8;
          *reinterpret_cast<uint2*>(rec) = s1h[j];
          *reinterpret_cast<uint2*>(rec + KC * 2) = s1l[j];
        }
      }
    }
  };

  f32x16 acc[NIW][MI];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

  // bias (and, for the split, the per-channel output scale) in the ROW order of epilogue A — lane % LPP picks CPL consecutive
  // channels of each 32-channel tile: fetched now, consumed after the LDS round trip (no exposed L2 round trip there)
  constexpr int CPL_A = 16 / ES, LPP_A = 32 / CPL_A;
  float bias_l[NIW][CPL_A], osc_l[NIW][CPL_A];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni) {
    const int chb = min(ntile0 + ni, a.ntiles32 - 1) * 32 + (lane % LPP_A) * CPL_A;
#pragma unroll
    for (int q = 0; q < CPL_A; q += 4) {
      const float4 bv = *reinterpret_cast<const float4*>(a.bias + chb + q);
      bias_l[ni][q] = bv.x; bias_l[ni][q + 1] = bv.y; bias_l[ni][q + 2] = bv.z; bias_l[ni][q + 3] = bv.w;
      float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
      if constexpr (SPLIT) { if (a.oscale) sv = *reinterpret_cast<const float4*>(a.oscale + chb + q); }
      osc_l[ni][q] = sv.x; osc_l[ni][q + 1] = sv.y; osc_l[ni][q + 2] = sv.z; osc_l[ni][q + 3] = sv.w;
    }
  }
  // per pixel fragment of this wave: inverse activation scale, image (slot index) and the end of the image's real pixels (padding
  // pixels behind an image are computed like any pixel but neither stored nor counted in the max-|value| slot)
  float tinv_f[MI], amax_run[MI];
  int img_f[MI];
  long long real_end[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    tinv_f[mi] = tinv; amax_run[mi] = 0.f; img_f[mi] = KS == 3 ? img3 : 0; real_end[mi] = a.total_px;
    if constexpr (SPLIT && KS == 1) {
      if (a.amax_img) {
        const int fw = __builtin_amdgcn_readfirstlane(wm * MI + mi);
#pragma unroll
        for (int f = 0; f < G::FR; ++f)
          if (fw == f) { img_f[mi] = f_img[f]; real_end[mi] = f_end[f]; tinv_f[mi] = f_ti[f]; }
      }
    }
  }

  // One step = one (k-group, tap): MI pixel fragments + NIW weight fragments -> NIW x MI MFMAs. The fragments of step
  // s + PD are requested from LDS before the MFMAs of step s issue (explicit register ring, everything unrolled): with one
  // or two waves per SIMD nothing else hides the ~100+ cycle LDS latency, and hipcc on its own schedules each read right
  // in front of its MFMA (measured: the MFMA phase of the SR body convs ran 3x longer than its MFMA cycles).
  // A partial last chunk is multiplied in full: its missing channels / k-groups were zero-filled by the loader.
  constexpr int NS = G::NS, PD = G::PD;
  constexpr int FB = MI * (SPLIT ? 2 : 1), FA = NIW * (SPLIT ? 2 : 1);
  auto compute = [&](const unsigned char* sb) {
    const unsigned char* sw = sb + G::IN_BYTES + ((wn * NIW) * TAPS * KCG) * WFRAG + lane * 16;
    uint4 bq[PD + 1][FB], aq[PD + 1][FA];
    auto ld = [&](int s, int q) {
      const int ks = s / TAPS, tap = s % TAPS, ky = tap / KS, kx = tap % KS;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const unsigned char* bp = sb + boff[mi] + (ky * HW + kx) * PS + ks * LKB;
        bq[q][mi] = *reinterpret_cast<const uint4*>(bp);
        if constexpr (SPLIT) bq[q][MI + mi] = *reinterpret_cast<const uint4*>(bp + KC * 2);
      }
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const unsigned char* ap = sw + ((ni * TAPS + tap) * KCG + ks) * WFRAG;
        aq[q][ni] = *reinterpret_cast<const uint4*>(ap);
        if constexpr (SPLIT) aq[q][NIW + ni] = *reinterpret_cast<const uint4*>(ap + 1024);
      }
    };
#pragma unroll
    for (int s = 0; s < PD; ++s) { ld(s, s); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + PD < NS) ld(s + PD, (s + PD) % (PD + 1));
      if constexpr (PD > 0) __builtin_amdgcn_sched_barrier(0);        // keep the reads above ahead of this step's MFMAs (the scheduler otherwise sinks them)
      const int q = s % (PD + 1);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if constexpr (SPLIT) MM<X3>::mma3(acc[ni][mi], aq[q][ni], aq[q][NIW + ni], bq[q][mi], bq[q][MI + mi]);
          else MM<T>::mma(acc[ni][mi], aq[q][ni], bq[q][mi]);
        }
      if constexpr (PD > 0) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // prologue: D chunks requested back to back, the first one moved into LDS stage 0
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d * KC < a.cin && (d == 0 || !(a.dbg & 4))) fetch(d * KC, ri[d], rw[d]);
  stem_fill(smem);
  stash(smem, ri[0], rw[0]);
  __syncthreads();
  int cur = 0;
  for (int cb = 0; cb < a.cin; cb += D * KC) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int c0 = cb + d * KC;
      if (c0 < a.cin) {                                              // block-uniform
        // register set d carried chunk c0, which already sits in LDS stage `cur`: refill it D chunks ahead
        if (c0 + D * KC < a.cin && !(a.dbg & 4)) fetch(c0 + D * KC, ri[d], rw[d]);
        if (!(a.dbg & 2)) compute(smem + cur * G::BUF);
        // chunk c0 + KC was requested D - 1 iterations ago: move it into the other stage (last read one barrier ago)
        if constexpr (G::STAGES == 1) {
          __syncthreads();                                             // everyone is done reading the only stage
          if (c0 + KC < a.cin && !(a.dbg & 8)) { stem_flush(smem); stash(smem, ri[(d + 1) % D], rw[(d + 1) % D]); }
          __syncthreads();
        } else {
          if (c0 + KC < a.cin && !(a.dbg & 8)) { stem_flush(smem + (cur ^ 1) * G::BUF); stash(smem + (cur ^ 1) * G::BUF, ri[(d + 1) % D], rw[(d + 1) % D]); }
          __syncthreads();
          cur ^= 1;
        }
      }
    }
  }

  // ---- epilogue A (aligned outputs): transpose each 32 px x 32 ch accumulator tile through a wave-private LDS tile so
  // that global stores (and residual loads) are whole pixel rows — 8 (fp32) / 4 (fp16) consecutive lanes cover one
  // pixel's 32 channels with 16-byte accesses. The direct form (epilogue B) writes 16 scattered bytes per lane and
  // measured 40-65 % of the kernel time on store-heavy layers.
  if (a.fast_out && !(a.dbg & 1)) {
    constexpr int EROW = 32 * 4 + 16;                 // fp32 staging row per pixel (+16 B pad)
    constexpr int CPL = 16 / ES;                      // output channels per 16-byte chunk (4 fp32 / 8 fp16)
    constexpr int LPP = 32 / CPL;                     // lanes per pixel row (8 / 4)
    constexpr int PPI = 64 / LPP;                     // pixels per store instruction (8 / 16)
    unsigned char* et = smem + wave * (32 * EROW);
    const unsigned char* r1b = a.res1 ? reinterpret_cast<const unsigned char*>(a.res1) + ((long long)out_base * a.r1_cs + a.r1_coff) * ES : wb;
    const unsigned char* r2b = a.res2 ? reinterpret_cast<const unsigned char*>(a.res2) + ((long long)out_base * a.r2_cs + a.r2_coff) * ES : wb;
    const auto rs_r1 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(r1b)), 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_r2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<unsigned char*>(r2b)), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int f = wm * MI + mi;
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const int nt = ntile0 + ni;
        if (nt >= a.ntiles32) continue;
        // this lane's share of the tile in the transposed (row) order: pixel index + validity + residual chunks, issued
        // BEFORE the LDS round trip so that their latency overlaps it
        long long gps[32 / PPI];
        bool oks[32 / PPI];
        uint4 r1v[32 / PPI], r2v[32 / PPI];
#pragma unroll
        for (int it = 0; it < 32 / PPI; ++it) {
          const int pp = it * PPI + lane / LPP, ch0 = (lane % LPP) * CPL;
          if (KS == 1) {
            gps[it] = out_base + f * 32 + pp;
            oks[it] = gps[it] < real_end[mi];
          } else {
            const int oy = oy0 + 2 * f + (pp >> 4), ox = ox0 + (pp & 15);
            oks[it] = oy < Ho && ox < Wo;
            gps[it] = out_base + (long long)oy * Wo + ox;
          }
          oks[it] = oks[it] && nt * 32 + ch0 < a.cout;
          r1v[it] = make_uint4(0u, 0u, 0u, 0u); r2v[it] = r1v[it];
          const unsigned rel_px = (unsigned)(gps[it] - out_base);           // pixel index inside this workgroup's image / run
          if (a.res1) r1v[it] = bload(rs_r1, oks[it] ? (rel_px * a.r1_cs + nt * 32 + ch0) * ES : OOB);      // uniform branch, branch-free lanes
          if (a.res2) r2v[it] = bload(rs_r2, oks[it] ? (rel_px * a.r2_cs + nt * 32 + ch0) * ES : OOB);
        }
        float oscf[CPL];                                        // output scale of this lane's channels x the fragment's inverse activation scale
#pragma unroll
        for (int q = 0; q < CPL; ++q) oscf[q] = osc_l[ni][q] * tinv_f[mi];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // previous tile's staging reads have landed in registers
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = 8 * g + 4 * hh;
          *reinterpret_cast<float4*>(et + p * EROW + ch * 4) =
              make_float4(acc[ni][mi][4 * g + 0], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS is in order per wave; make the tile visible to all lanes
#pragma unroll
        for (int it = 0; it < 32 / PPI; ++it) {
          const int pp = it * PPI + lane / LPP, ch0 = (lane % LPP) * CPL;
          if (!oks[it]) continue;
          float v[CPL];
#pragma unroll
          for (int q = 0; q < CPL / 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(et + pp * EROW + (ch0 + 4 * q) * 4);
            v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
          }
#pragma unroll
          for (int q = 0; q < CPL; ++q) {
            if constexpr (SPLIT) v[q] = apply_act(fmaf(v[q], oscf[q], bias_l[ni][q]), a.act);
            else v[q] = apply_act(v[q] + bias_l[ni][q], a.act);
          }
          if (a.res1) {
            const GT* r = reinterpret_cast<const GT*>(&r1v[it]);
#pragma unroll
            for (int q = 0; q < CPL; ++q) v[q] = v[q] * a.s1 + (float)r[q];
          }
          if (a.res2) {
            const GT* r = reinterpret_cast<const GT*>(&r2v[it]);
#pragma unroll
            for (int q = 0; q < CPL; ++q) v[q] = v[q] * a.s2 + (float)r[q];
          }
          if constexpr (SPLIT) {
#pragma unroll
            for (int q = 0; q < CPL; ++q) amax_run[mi] = fmaxf(amax_run[mi], fabsf(v[q]));
          }
          uint4 ov;
          GT* o = reinterpret_cast<GT*>(&ov);
#pragma unroll
          for (int q = 0; q < CPL; ++q) o[q] = (GT)v[q];
          *reinterpret_cast<uint4*>(reinterpret_cast<GT*>(a.out) + (size_t)gps[it] * a.out_cs + a.out_coff + nt * 32 + ch0) = ov;
        }
      }
    }
    if constexpr (SPLIT) {
      if (a.amax_out) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {                 // fragments of one image (always, for a 3x3 tile) raise its slot once
          bool same = false;
          if (mi + 1 < MI) same = !a.amax_img || img_f[mi + 1] == img_f[mi];
          if (same) amax_run[mi + 1 < MI ? mi + 1 : mi] = fmaxf(amax_run[mi + 1 < MI ? mi + 1 : mi], amax_run[mi]);
          else raise_amax(a.amax_out + (a.amax_img ? img_f[mi] : 0), amax_run[mi]);
        }
      }
    }
    return;
  }

  // ---- epilogue B (any alignment / channel count): lane = one pixel; regs 4g..4g+3 = channels 8g+4hh..+3.
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int f = wm * MI + mi;
    long long gp;
    bool ok;
    if (KS == 1) {
      gp = out_base + f * 32 + p;
      ok = gp < real_end[mi];
    } else {
      const int oy = oy0 + 2 * f + (p >> 4), ox = ox0 + (p & 15);
      ok = oy < Ho && ox < Wo;
      gp = out_base + (long long)oy * Wo + ox;
    }
    if (!ok || (a.dbg & 1)) continue;
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      const int nt = ntile0 + ni;
      if (nt >= a.ntiles32) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = nt * 32 + 8 * g + 4 * hh;
        if (ch >= a.cout) continue;
        float v[4];
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + ch);      // bias is padded to 32 channels per tile
        if constexpr (SPLIT) {
          float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
          if (a.oscale) sv = *reinterpret_cast<const float4*>(a.oscale + ch);
          v[0] = fmaf(acc[ni][mi][4 * g + 0], sv.x * tinv_f[mi], bv.x);
          v[1] = fmaf(acc[ni][mi][4 * g + 1], sv.y * tinv_f[mi], bv.y);
          v[2] = fmaf(acc[ni][mi][4 * g + 2], sv.z * tinv_f[mi], bv.z);
          v[3] = fmaf(acc[ni][mi][4 * g + 3], sv.w * tinv_f[mi], bv.w);
        } else {
          v[0] = acc[ni][mi][4 * g + 0] + bv.x;
          v[1] = acc[ni][mi][4 * g + 1] + bv.y;
          v[2] = acc[ni][mi][4 * g + 2] + bv.z;
          v[3] = acc[ni][mi][4 * g + 3] + bv.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], a.act);
        const bool vec = a.vec_ok && ch + 3 < a.cout;
        if (a.res1) {
          const GT* rp = reinterpret_cast<const GT*>(a.res1) + (size_t)gp * a.r1_cs + a.r1_coff + ch;
          float r[4] = {0.f, 0.f, 0.f, 0.f};
          if (vec) load4<GT>(rp, r);
          else for (int j = 0; j < 4; ++j) if (ch + j < a.cout) r[j] = (float)rp[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] * a.s1 + r[j];
        }
        if (a.res2) {
          const GT* rp = reinterpret_cast<const GT*>(a.res2) + (size_t)gp * a.r2_cs + a.r2_coff + ch;
          float r[4] = {0.f, 0.f, 0.f, 0.f};
          if (vec) load4<GT>(rp, r);
          else for (int j = 0; j < 4; ++j) if (ch + j < a.cout) r[j] = (float)rp[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] * a.s2 + r[j];
        }
        if constexpr (SPLIT) {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (ch + j < a.cout) amax_run[mi] = fmaxf(amax_run[mi], fabsf(v[j]));
        }
        const size_t oidx = (size_t)gp * a.out_cs + a.out_coff + ch;
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + oidx;
          if (vec) store4<float>(op, v);
          else for (int j = 0; j < 4; ++j) if (ch + j < a.cout) op[j] = v[j];
        } else {
          GT* op = reinterpret_cast<GT*>(a.out) + oidx;
          if (vec) store4<GT>(op, v);
          else for (int j = 0; j < 4; ++j) if (ch + j < a.cout) op[j] = (GT)v[j];
        }
      }
    }
  }
  if constexpr (SPLIT) {
    if (a.amax_out) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bool same = false;
        if (mi + 1 < MI) same = !a.amax_img || img_f[mi + 1] == img_f[mi];
        if (same) amax_run[mi + 1 < MI ? mi + 1 : mi] = fmaxf(amax_run[mi + 1 < MI ? mi + 1 : mi], amax_run[mi]);
        else raise_amax(a.amax_out + (a.amax_img ? img_f[mi] : 0), amax_run[mi]);
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------
namespace {

template <typename T, int KS, int STRIDE, int WM, int WN, int MI, int NIW, int KC> struct Cfg {
  using G = Geo<T, KS, STRIDE, WM, WN, MI, NIW, KC>;
  static constexpr int LDS = G::LDS;
  static constexpr bool OK = LDS <= 160 * 1024;            // shapes that do not fit the LDS are simply not offered
  static constexpr int BLOCK_PX = G::FR * 32;
  static constexpr int BLOCK_N = G::NTB * 32;
  static void init() {
    if constexpr (OK)
      FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<T, KS, STRIDE, WM, WN, MI, NIW, KC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  static void launch(ConvArgs& a, Level* out_lvl, hipStream_t st) {
    if constexpr (OK) {
      int n_tiles;
      if (KS == 1) {
        n_tiles = (int)((a.total_px + BLOCK_PX - 1) / BLOCK_PX);
      } else {
        a.tiles = out_lvl->tile_table(G::TH, &n_tiles, &a.n_tiles_dev, st);
      }
      a.n_nblk = (a.ntiles32 * 32 + BLOCK_N - 1) / BLOCK_N;
      const int grid = n_tiles * a.n_nblk;
      if (grid == 0) return;
      hipLaunchKernelGGL((conv_mfma_kernel<T, KS, STRIDE, WM, WN, MI, NIW, KC>), dim3(grid), dim3(256), LDS, st, a);
    } else {
      fail(FFP_ERR_STATE, "conv: shape does not fit the LDS");
    }
  }
};

// Chunk sizes (input channels per LDS stage) per shape {0 wide, 1 wideH, 2 narrow2, 3 narrow2H, 4 narrow1, 5 narrow1H}: the largest
// that keeps 2 x (input tile + weight fragments) <= 80 KiB, i.e. TWO workgroups resident per CU, so that one
// workgroup's prologue / epilogue overlaps the other's MFMA phase (measured: an occupancy-2 32-channel block beats an
// occupancy-1 64-channel block by 1.3x on conv_hr). Where even the minimum chunk does not fit 80 KiB the minimum is used.
template <typename T, int KS, int STRIDE, int SHAPE> struct KCof;
#define FFP_KC(T, KS, S, a, b, c, d, e, f)                                           \
  template <> struct KCof<T, KS, S, 0> { static constexpr int v = a; };              \
  template <> struct KCof<T, KS, S, 1> { static constexpr int v = b; };              \
  template <> struct KCof<T, KS, S, 2> { static constexpr int v = c; };              \
  template <> struct KCof<T, KS, S, 3> { static constexpr int v = d; };              \
  template <> struct KCof<T, KS, S, 4> { static constexpr int v = e; };              \
  template <> struct KCof<T, KS, S, 5> { static constexpr int v = f; };
//                  wide wideH  n2  n2H  n1  n1H
FFP_KC(float, 1, 1,   16,  32,  16,  32, 16,  32)
FFP_KC(float, 3, 1,    8,   8,   8,   8,  8,  16)
FFP_KC(float, 3, 2,    8,   8,   8,   8,  8,   8)
FFP_KC(_Float16, 1, 1, 32,  64,  32,  64, 32,  64)
FFP_KC(_Float16, 3, 1, 16,  16,  16,  16, 16,  32)
FFP_KC(_Float16, 3, 2, 16,  16,  16,  16, 16,  16)
FFP_KC(X3, 1, 1,      16,  32,  16,  32, 16,  32)
FFP_KC(X3, 3, 1,      16,  16,  16,  16, 16,  16)
FFP_KC(X3, 3, 2,      16,  16,  16,  16, 16,  16)
#undef FFP_KC

// Shape selection: six workgroup shapes per (dtype, k, stride) — {wide 128ch, narrow2 64ch, narrow1 32ch} x {full, half
// pixel tile}. The largest block (most operand reuse) that still yields >= 2 workgroups per CU wins; layers with few
// pixels (stride-32 maps, face crops) fall through to smaller blocks so that the whole chip is busy.
template <typename T, int KS, int STRIDE> struct Family {
  static constexpr int MIW = STRIDE == 2 ? 2 : 4;   // wide: 2x2 waves
  static constexpr int MIN = STRIDE == 2 ? 1 : 2;   // narrow: 4x1 waves
  static constexpr int MIWH = MIW / 2;
  static constexpr int MINH = MIN > 1 ? MIN / 2 : 1;
  using Wide = Cfg<T, KS, STRIDE, 2, 2, MIW, 2, KCof<T, KS, STRIDE, 0>::v>;
  using WideH = Cfg<T, KS, STRIDE, 2, 2, MIWH, 2, KCof<T, KS, STRIDE, 1>::v>;
  using Narrow2 = Cfg<T, KS, STRIDE, 4, 1, MIN, 2, KCof<T, KS, STRIDE, 2>::v>;
  using Narrow2H = Cfg<T, KS, STRIDE, 4, 1, MINH, 2, KCof<T, KS, STRIDE, 3>::v>;
  using Narrow1 = Cfg<T, KS, STRIDE, 4, 1, MIN, 1, KCof<T, KS, STRIDE, 4>::v>;
  using Narrow1H = Cfg<T, KS, STRIDE, 4, 1, MINH, 1, KCof<T, KS, STRIDE, 5>::v>;
  static constexpr bool HAS_NH = MIN > 1;
  static void init() {
    Wide::init(); Narrow2::init(); Narrow1::init(); WideH::init();
    if (HAS_NH) { Narrow2H::init(); Narrow1H::init(); }
  }
  template <class C> static long long wgs(const ConvArgs& a, Level* out_lvl) {
    const long long tiles = KS == 1 ? (a.total_px + C::BLOCK_PX - 1) / C::BLOCK_PX : out_lvl->count_tiles(C::G::TH);
    return tiles * ((a.ntiles32 * 32 + C::BLOCK_N - 1) / C::BLOCK_N);
  }
  static unsigned valid_mask(const ConvArgs& a) {
    const bool wide_ok = a.ntiles32 >= 3 && (KS == 1 || STRIDE == 2);
    unsigned m = 0;
    if (Wide::OK && wide_ok) m |= 1u << 0;
    if (WideH::OK && wide_ok) m |= 1u << 1;
    if (Narrow2::OK && a.ntiles32 >= 2) m |= 1u << 2;
    if (HAS_NH && Narrow2H::OK && a.ntiles32 >= 2) m |= 1u << 3;
    if (Narrow1::OK) m |= 1u << 4;
    if (HAS_NH && Narrow1H::OK) m |= 1u << 5;
    return m;
  }
  static int choose(const ConvArgs& a, Level* out_lvl) {
    // candidates in decreasing block size; first pass: shapes that allow two resident workgroups per CU (LDS <= 80 KiB) and fill the chip; second pass: any shape with enough work; else the shape with the most workgroups.
    constexpr long long ENOUGH = 192;     // >= 3/4 of the 256 CUs get a workgroup: measured better than insisting on two for the SR body convs
    struct Cand { int id; long long n; bool valid; bool occ2; };
    const bool wide_ok = a.ntiles32 >= 3 && (KS == 1 || STRIDE == 2);
    const Cand c[6] = {
        {0, wgs<Wide>(a, out_lvl), Wide::OK && wide_ok, Wide::LDS <= 80 * 1024},
        {1, wgs<WideH>(a, out_lvl), WideH::OK && wide_ok, WideH::LDS <= 80 * 1024},
        {2, wgs<Narrow2>(a, out_lvl), Narrow2::OK && a.ntiles32 >= 2, Narrow2::LDS <= 80 * 1024},
        {3, wgs<Narrow2H>(a, out_lvl), HAS_NH && Narrow2H::OK && a.ntiles32 >= 2, Narrow2H::LDS <= 80 * 1024},
        {4, wgs<Narrow1>(a, out_lvl), Narrow1::OK, Narrow1::LDS <= 80 * 1024},
        {5, wgs<Narrow1H>(a, out_lvl), HAS_NH && Narrow1H::OK, Narrow1H::LDS <= 80 * 1024}};
    for (const Cand& k : c) if (k.valid && k.occ2 && k.n >= ENOUGH) return k.id;
    for (const Cand& k : c) if (k.valid && k.n >= ENOUGH) return k.id;
    int best = -1;
    long long best_n = -1;
    for (const Cand& k : c) if (k.valid && k.n > best_n) { best = k.id; best_n = k.n; }
    if (best < 0) fail(FFP_ERR_STATE, "conv: no workgroup shape fits");
    return best;
  }
  static void launch(ConvArgs& a, Level* out_lvl, hipStream_t st) {
    switch (a.force_shape >= 0 ? a.force_shape : choose(a, out_lvl)) {
      case 0: Wide::launch(a, out_lvl, st); break;
      case 1: WideH::launch(a, out_lvl, st); break;
      case 2: Narrow2::launch(a, out_lvl, st); break;
      case 3: Narrow2H::launch(a, out_lvl, st); break;
      case 4: Narrow1::launch(a, out_lvl, st); break;
      default: Narrow1H::launch(a, out_lvl, st); break;
    }
  }
};

static const char* kShapeNames[6] = {"wide", "wideH", "narrow2", "narrow2H", "narrow1", "narrow1H"};

template <typename T> int choose_t(const ConvArgs& a, int k, int stride, Level* out_lvl) {
  if (k == 1 && stride == 1) return Family<T, 1, 1>::choose(a, out_lvl);
  if (k == 3 && stride == 1) return Family<T, 3, 1>::choose(a, out_lvl);
  if (k == 3 && stride == 2) return Family<T, 3, 2>::choose(a, out_lvl);
  return 4;
}

template <typename T> unsigned valid_t(const ConvArgs& a, int k, int stride) {
  if (k == 1 && stride == 1) return Family<T, 1, 1>::valid_mask(a);
  if (k == 3 && stride == 1) return Family<T, 3, 1>::valid_mask(a);
  if (k == 3 && stride == 2) return Family<T, 3, 2>::valid_mask(a);
  return 0;
}

template <typename T> void launch_t(ConvArgs& a, int k, int stride, Level* out_lvl, hipStream_t st) {
  if (k == 1 && stride == 1) Family<T, 1, 1>::launch(a, out_lvl, st);
  else if (k == 3 && stride == 1) Family<T, 3, 1>::launch(a, out_lvl, st);
  else if (k == 3 && stride == 2) Family<T, 3, 2>::launch(a, out_lvl, st);
  else fail(FFP_ERR_ARG, "conv: k=%d stride=%d has no kernel", k, stride);
}

}  // namespace

// 256 zero bytes per device: where the LDS-DMA loader of conv_rows.hip reads conv zero padding from
static const void* zero_block() {
  static std::map<int, DevBuf> blocks;
  int dev = 0;
  FFP_HIP(hipGetDevice(&dev));
  auto it = blocks.find(dev);
  if (it == blocks.end()) {
    DevBuf b(256);
    FFP_HIP(hipMemset(b.p, 0, 256));
    it = blocks.emplace(dev, std::move(b)).first;
  }
  return it->second.p;
}

// ---- stem fused into the first MFMA conv (conv_mfma_kernel<X3, 3, 2, 4, 1, 1, 2, 16, true>) ------------------------------------
namespace {
using StemG = Geo<X3, 3, 2, 4, 1, 1, 2, 16>;       // YOLO11s: 32 stem channels -> 64
using StemGn = Geo<X3, 3, 2, 4, 1, 1, 1, 16>;      // YOLO11n: 16 -> 32
constexpr float kStemWScale = 4096.f;      // the packed stem weights are w / 255 * 2^12: fp16 normal range for every weight that matters

// fp32 stem weights [tap][net channel][cout] -> A fragments of v_mfma_f32_16x16x32_f16 in the loader's k order (see stem_fill):
// out[((chunk * 2 + part) * 64 + lane) * 8 + i], part 0 = hi, 1 = lo; lane = (channel % 16) + 16 * kb
__global__ void stem_pack_kernel(const float* __restrict__ w, int cout, int flip, _Float16* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * 64 * 8) return;
  const int i = idx & 7, lane = (idx >> 3) & 63, chunk = idx >> 9;
  const int ch = chunk * 16 + (lane & 15), kb = lane >> 4;
  int r = -1, col = 0, sc = 0;                      // window row / column / source channel of k slot (kb, i); r < 0: zero slot
  if (kb < 3) { r = kb; col = i / 3; sc = i % 3; }
  else if (i < 3) { r = i; col = 2; sc = 2; }
  float v = 0.f;
  if (r >= 0 && ch < cout) v = w[((r * 3 + col) * 3 + (flip ? 2 - sc : sc)) * cout + ch] / 255.0f * kStemWScale;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  out[((chunk * 2 + 0) * 64 + lane) * 8 + i] = hi;
  out[((chunk * 2 + 1) * 64 + lane) * 8 + i] = lo;
}
}  // namespace

bool stem_conv_eligible(const ConvOp& stem, const ConvOp& conv) {
  const PackedConv& ps = *stem.pc;
  const PackedConv& pc = *conv.pc;
  static const bool off = [] { const char* e = getenv("FFP_NO_STEM_FUSE"); return e && e[0] == '1'; }();
  return !off && conv_direct_eligible(stem) && ps.dt == F32 && stem.stride == 2 && stem.act == ACT_SILU && (ps.cout == 32 || ps.cout == 16) && ps.w_direct.p != nullptr &&
         pc.split && pc.dt == F32 && pc.k == 3 && conv.stride == 2 && pc.cin == ps.cout && pc.cout_pad == 2 * ps.cout && !conv.up && !conv.has_res1 && !conv.has_res2 &&
         !conv.has_up2 && conv.in.lvl == stem.out.lvl && conv.in.amax != nullptr && !conv.out.lvl->capacity();
}

void stem_conv_pack(const ConvOp& stem, int flip, DevBuf& out, hipStream_t st) {
  out.alloc(2 * 2 * 64 * 16);
  hipLaunchKernelGGL(stem_pack_kernel, dim3(4), dim3(256), 0, st, stem.pc->w_direct.as<float>(), stem.pc->cout, flip, out.as<_Float16>());
  FFP_HIP(hipGetLastError());
}

void launch_stem_conv(const uint8_t* d_frame, int H, int W, const DevBuf& d_imgs, const DevBuf& stem_w, const ConvOp& stem, const ConvOp& conv, hipStream_t st) {
  FFP_CHECK(stem_conv_eligible(stem, conv), FFP_ERR_ARG, "conv %s: not eligible for the stem-fused loader", conv.pc->name.c_str());
  ConvArgs a = make_conv_args(conv);
  a.in = nullptr;                                  // never read
  a.st_frame = d_frame; a.st_W = W;
  a.st_bytes = ((long long)H * W * 3 + 3) & ~3ll;    // whole dwords: the last pixel's dword may end up to 3 bytes past the frame (inside the allocation granule)
  a.st_imgs = d_imgs.as<LetterboxImg>();
  a.st_tab = stem.in.lvl->d_tab.as<int4>();
  a.st_w = stem_w.p;
  a.st_bias = stem.pc->bias.as<float>();
  a.st_scale = 1.0f / kStemWScale;
  int n_tiles = 0;
  static_assert(StemG::TH == StemGn::TH, "one tile table for both");
  a.tiles = conv.out.lvl->tile_table(StemG::TH, &n_tiles, &a.n_tiles_dev, st);
  a.n_nblk = 1;
  if (n_tiles == 0) return;
  if (stem.pc->cout == 32) hipLaunchKernelGGL((conv_mfma_kernel<X3, 3, 2, 4, 1, 1, 2, 16, true>), dim3(n_tiles), dim3(256), StemG::LDS, st, a);
  else hipLaunchKernelGGL((conv_mfma_kernel<X3, 3, 2, 4, 1, 1, 1, 16, true>), dim3(n_tiles), dim3(256), StemGn::LDS, st, a);
  FFP_HIP(hipGetLastError());
}

void conv_kernels_init() {
  (void)zero_block();
  static bool done = false;
  if (done) return;
  conv_rows_init();
  conv_rows16_init();
  conv_rows16pc_init();
  conv_trunk_init();
  conv_pw_init();
  conv_k3d_init();
  Family<float, 1, 1>::init(); Family<float, 3, 1>::init(); Family<float, 3, 2>::init();
  Family<_Float16, 1, 1>::init(); Family<_Float16, 3, 1>::init(); Family<_Float16, 3, 2>::init();
  Family<X3, 1, 1>::init(); Family<X3, 3, 1>::init(); Family<X3, 3, 2>::init();
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<X3, 3, 2, 4, 1, 1, 2, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, StemG::LDS));
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<X3, 3, 2, 4, 1, 1, 1, 16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, StemGn::LDS));
  done = true;
}

ConvArgs make_conv_args(const ConvOp& op) {
  const PackedConv& pc = *op.pc;
  FFP_CHECK(!pc.depthwise(), FFP_ERR_ARG, "conv %s: depthwise goes through launch_dwconv", pc.name.c_str());
  FFP_CHECK(op.in.dt == pc.dt, FFP_ERR_ARG, "conv %s: input dtype differs from packed weights", pc.name.c_str());
  FFP_CHECK(op.in.C == pc.cin && op.out.C == pc.cout, FFP_ERR_ARG, "conv %s: view channels (%d->%d) != weights (%d->%d)",
            pc.name.c_str(), op.in.C, op.out.C, pc.cin, pc.cout);
  const int epv = 16 / dsize(pc.dt);
  FFP_CHECK(op.in.cs % epv == 0 && op.in.coff % epv == 0 && pc.cin % epv == 0, FFP_ERR_ARG,
            "conv %s: input view not 16-byte aligned (cs=%d coff=%d cin=%d)", pc.name.c_str(), op.in.cs, op.in.coff, pc.cin);
  FFP_CHECK(!(op.up && (pc.k != 3 || op.stride != 1)), FFP_ERR_ARG, "conv %s: upsampled input needs k3 s1", pc.name.c_str());
  ConvArgs a{};
  a.in = op.in.ptr; a.wpk = pc.w.p; a.bias = pc.bias.as<float>(); a.out = op.out.ptr;
  a.res1 = op.has_res1 ? op.res1.ptr : nullptr;
  a.res2 = op.has_res2 ? op.res2.ptr : nullptr;
  a.in_tab = op.in.lvl->d_tab.as<int4>();
  a.out_tab = op.out.lvl->d_tab.as<int4>();
  a.tiles = nullptr;
  a.n_tiles_dev = nullptr;
  a.total_px = op.out.lvl->total_px;
  a.in_cs = op.in.cs; a.in_coff = op.in.coff; a.cin = pc.cin; a.cin_pad = pc.cin_pad;
  a.out_cs = op.out.cs; a.out_coff = op.out.coff; a.cout = pc.cout;
  a.r1_cs = op.res1.cs; a.r1_coff = op.res1.coff; a.r2_cs = op.res2.cs; a.r2_coff = op.res2.coff;
  a.s1 = op.s1; a.s2 = op.s2;
  a.act = op.act; a.out_f32 = (op.out.dt == F32) ? 1 : 0; a.up = op.up;
  a.ncg = pc.ncg; a.ntiles32 = pc.cout_pad / 32;
  bool vec = pc.cout % 4 == 0 && op.out.cs % 4 == 0 && op.out.coff % 4 == 0;
  if (op.has_res1) {
    FFP_CHECK(op.res1.dt == pc.dt, FFP_ERR_ARG, "conv %s: residual dtype", pc.name.c_str());
    vec = vec && op.res1.cs % 4 == 0 && op.res1.coff % 4 == 0;
  }
  if (op.has_res2) {
    FFP_CHECK(op.res2.dt == pc.dt, FFP_ERR_ARG, "conv %s: residual dtype", pc.name.c_str());
    vec = vec && op.res2.cs % 4 == 0 && op.res2.coff % 4 == 0;
  }
  a.vec_ok = vec ? 1 : 0;
  {   // epilogue A needs 16-byte chunks of the OUTPUT element type everywhere (out + residuals) and out dtype == activation dtype
    const int cpl = 16 / dsize(pc.dt);
    bool f = op.out.dt == pc.dt && pc.cout % cpl == 0 && op.out.cs % cpl == 0 && op.out.coff % cpl == 0;
    if (op.has_res1) f = f && op.res1.cs % cpl == 0 && op.res1.coff % cpl == 0;
    if (op.has_res2) f = f && op.res2.cs % cpl == 0 && op.res2.coff % cpl == 0;
    a.fast_out = f ? 1 : 0;
  }
  a.up_src = nullptr; a.up_map = nullptr; a.up_c = 0; a.up_cs = 0;
  if (op.has_up2) {
    FFP_CHECK(pc.k == 1 && op.stride == 1 && op.up2.dt == pc.dt && op.up2_c > 0 && op.up2_c % 64 == 0 && op.up2_c <= pc.cin && op.up2.C == op.up2_c,
              FFP_ERR_ARG, "conv %s: x2-source needs a 1x1 conv and a channel count that is a multiple of 64", pc.name.c_str());
    FFP_CHECK(op.up2.cs % epv == 0 && op.up2.coff % epv == 0, FFP_ERR_ARG, "conv %s: x2-source view not 16-byte aligned", pc.name.c_str());
    a.up_src = reinterpret_cast<const unsigned char*>(op.up2.ptr) + (size_t)op.up2.coff * dsize(pc.dt);
    a.up_map = op.up2_map;
    a.up_c = op.up2_c; a.up_cs = op.up2.cs;
    FFP_CHECK(a.up_map != nullptr, FFP_ERR_STATE, "conv %s: x2-source map missing", pc.name.c_str());
  }
  a.amax_in = a.amax_in2 = nullptr; a.amax_out = nullptr; a.oscale = nullptr;
  a.amax_img = 0; a.frag_img = nullptr;
  if (pc.split && op.in.amax && op.in.amax_n > 1) {          // a slot per image (Plan::per_image_amax)
    FFP_CHECK(op.in.amax_n == op.in.lvl->n && (!op.out.amax || op.out.amax_n == op.in.amax_n) && (!op.has_up2 || !op.up2.amax || op.up2.amax_n == op.in.amax_n), FFP_ERR_STATE,
              "conv %s: per-image exponent slots of input / output / second source disagree", pc.name.c_str());
    a.amax_img = 1;
    if (pc.k == 1) a.frag_img = op.out.lvl->frag_img();
  }
  if (pc.split) {
    a.oscale = pc.oscale.as<float>();
    a.amax_in = op.in.amax;
    a.amax_in2 = op.has_up2 ? op.up2.amax : nullptr;
    a.amax_out = op.out.amax;
    FFP_CHECK(!op.has_up2 || (op.in.amax != nullptr) == (op.up2.amax != nullptr), FFP_ERR_STATE, "conv %s: both sources need a max-|value| slot", pc.name.c_str());
  }
  a.ntiles_host = 0;
  a.dbg = op.dbg;
  a.force_shape = op.force_shape;
  a.zeros = zero_block();
  if (pc.k == 1) FFP_CHECK(!op.out.lvl->capacity(), FFP_ERR_ARG, "conv %s: 1x1 convs are not built for capacity-mode levels", pc.name.c_str());
  if (pc.k == 1) FFP_CHECK(op.in.lvl->total_px == op.out.lvl->total_px, FFP_ERR_ARG, "conv %s: 1x1 levels differ", pc.name.c_str());
  return a;
}

bool conv_rows16_enabled() {
  static const bool on = [] { const char* e = getenv("FFP_ROWS16"); return !(e && e[0] == '0'); }();
  return on;
}

bool conv_pw_enabled() {
  static const bool on = [] { const char* e = getenv("FFP_PW"); return !(e && e[0] == '0'); }();
  return on;
}

static bool use_rows16(const ConvOp& op, const ConvArgs& a) {
  if (!conv_rows16_eligible(op, a)) return false;
  return a.force_shape == 9 || a.force_shape == 23 || a.force_shape == 24 || (a.force_shape < 0 && conv_rows16_enabled());
}

void launch_conv(const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  if (conv_direct_eligible(op)) { launch_conv_direct(op, st); return; }
  ConvArgs a = make_conv_args(op);
  if (a.force_shape == 25) {                     // the fused-body kernel on ONE layer (tests, probes): a throw-away single-layer plan
    FFP_CHECK(conv_trunk_layer_ok(op), FFP_ERR_ARG, "conv %s: not a layer conv_trunk_kernel can run", pc.name.c_str());
    TrunkPlan tp(std::vector<ConvOp>{op});
    tp.launch(st, op.dbg);
    FFP_HIP(hipStreamSynchronize(st));           // the plan's tables die with it
    return;
  }
  if (use_rows16(op, a)) {
    if (conv_rows16pc_selected(a)) launch_conv_rows16pc(a, pc, op.out.lvl, st);
    else launch_conv_rows16(a, pc, op.out.lvl, st);
    FFP_HIP(hipGetLastError());
    return;
  }
  if (conv_rows_eligible(op, a)) { launch_conv_rows(a, op.out.lvl, st); FFP_HIP(hipGetLastError()); return; }
  if ((a.force_shape >= 10 && a.force_shape <= 16) || a.force_shape == 22) {      // pointwise kernels (conv_pw.hip): only ever picked by measurement (conv_tune) or by hand
    FFP_CHECK(conv_pw_mask(op, a) & (1u << a.force_shape), FFP_ERR_ARG, "conv %s: pointwise shape %d cannot run this op", pc.name.c_str(), a.force_shape);
    launch_conv_pw(a, a.force_shape, st);
    FFP_HIP(hipGetLastError());
    return;
  }
  if (a.force_shape >= 17 && a.force_shape <= 21) {      // direct-weight 3x3 kernels (conv_k3d.hip): picked by measurement (conv_tune) or by hand
    FFP_CHECK(conv_k3d_mask(op, a) & (1u << a.force_shape), FFP_ERR_ARG, "conv %s: k3d shape %d cannot run this op", pc.name.c_str(), a.force_shape);
    launch_conv_k3d(a, a.force_shape, op.stride, op.out.lvl, st);
    FFP_HIP(hipGetLastError());
    return;
  }
  if (pc.dt == F16) launch_t<_Float16>(a, pc.k, op.stride, op.out.lvl, st);
  else if (pc.split) launch_t<X3>(a, pc.k, op.stride, op.out.lvl, st);
  else launch_t<float>(a, pc.k, op.stride, op.out.lvl, st);
  FFP_HIP(hipGetLastError());
}

std::string conv_variant(const ConvOp& op);

int conv_tune(const ConvOp& op, hipStream_t st) {
  const PackedConv& pc = *op.pc;
  if (conv_direct_eligible(op)) return -1;
  const ConvArgs a = make_conv_args(op);
  if (use_rows16(op, a) || conv_rows_eligible(op, a)) return -1;
  const unsigned mask = (pc.dt == F16 ? valid_t<_Float16>(a, pc.k, op.stride) : pc.split ? valid_t<X3>(a, pc.k, op.stride) : valid_t<float>(a, pc.k, op.stride)) |
                        (conv_pw_enabled() ? conv_pw_mask(op, a) : 0u) | (conv_k3d_enabled() ? conv_k3d_mask(op, a) : 0u);
  if (__builtin_popcount(mask) < 2) return -1;
  const int heur = pc.dt == F16 ? choose_t<_Float16>(a, pc.k, op.stride, op.out.lvl)
                   : pc.split ? choose_t<X3>(a, pc.k, op.stride, op.out.lvl) : choose_t<float>(a, pc.k, op.stride, op.out.lvl);
  hipEvent_t e0, e1;
  FFP_HIP(hipEventCreate(&e0));
  FFP_HIP(hipEventCreate(&e1));
  auto time_shape = [&](int shape, int iters) {
    ConvOp o = op;
    o.force_shape = shape;
    FFP_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_conv(o, st);
    FFP_HIP(hipEventRecord(e1, st));
    FFP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    FFP_HIP(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
  };
  float best_t = 0.f, heur_t = 0.f;
  int best = -1;
  for (int shape = 0; shape < 23; ++shape) {
    if (!(mask & (1u << shape))) continue;
    const float t1 = time_shape(shape, 2);                              // also builds the shape's tile table
    const int iters = std::min(24, std::max(3, (int)(300.f / std::max(t1, 1.f))));
    const float t = time_shape(shape, iters);
    if (shape == heur) heur_t = t;
    if (best < 0 || t < best_t) { best = shape; best_t = t; }
    static const bool log = [] { const char* e = getenv("FFP_TUNE_LOG"); return e && e[0] == '1'; }();       // every candidate's time, one line per (layer, shape)
    if (log) {
      ConvOp o = op;
      o.force_shape = shape;
      fprintf(stderr, "[tune] %-28s %4d->%4d k%d s%d px %9lld  shape %2d %-22s %9.1f us%s\n", pc.name.c_str(), pc.cin_real, pc.cout, pc.k, op.stride,
              (long long)op.out.lvl->actual_px(), shape, conv_variant(o).c_str(), t, shape == heur ? "  (heuristic)" : "");
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  // keep the heuristic's choice unless another shape is clearly (> 3 %) faster: timing noise must not flip plans
  if (heur_t > 0.f && best_t > 0.97f * heur_t) return heur;
  return best;
}

std::string conv_variant(const ConvOp& op) {
  const PackedConv& pc = *op.pc;
  if (conv_direct_eligible(op)) return std::string(pc.dt == F32 ? "f32" : "f16") + "_k3_direct";
  const ConvArgs a = make_conv_args(op);
  if (use_rows16(op, a)) return conv_rows16pc_selected(a) ? "f16_k3s1_rows16pc" : "f16_k3s1_rows16";
  if (conv_rows_eligible(op, a)) return "f16_k3s1_rows";
  if (op.force_shape == 22) return "f32x3_k1s1_pw2x2s";
  if (op.force_shape >= 10 && op.force_shape <= 16) {
    static const char* pw[7] = {"f32x3_k1s1_pw1x4", "f32x3_k1s1_pw2x2", "f32x3_k1s1_pw2x1", "f32x3_k1s1_pw1x4w", "f32x3_k1s1_pw2x2w", "f32x3_k1s1_pw2x1w", "f32x3_k1s1_pw1x4s"};
    return pw[op.force_shape - 10];
  }
  if (op.force_shape >= 17 && op.force_shape <= 21) {
    static const char* k3[5] = {"d128", "d64", "d32", "d128x256", "d64x256"};
    char b[64];
    snprintf(b, sizeof(b), "f32x3_k3s%d_%s", op.stride, k3[op.force_shape - 17]);
    return b;
  }
  const int shape = (op.force_shape >= 0 && op.force_shape < 6) ? op.force_shape
                    : pc.dt == F16 ? choose_t<_Float16>(a, pc.k, op.stride, op.out.lvl)
                    : pc.split ? choose_t<X3>(a, pc.k, op.stride, op.out.lvl) : choose_t<float>(a, pc.k, op.stride, op.out.lvl);
  char buf[64];
  snprintf(buf, sizeof(buf), "%s_k%ds%d_%s", pc.dt == F16 ? "f16" : pc.split ? "f32x3" : "f32", pc.k, op.stride, kShapeNames[shape < 0 ? 4 : shape]);
  return buf;
}

}  // namespace ffp
