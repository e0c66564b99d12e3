// conv_rows.hip — 3x3 stride-1 fp16 convolution with few output channels (32 / 64): the Real-ESRGAN dense-block convs.
//
// Why a second kernel: with 32 output channels an MFMA B operand (pixel fragment) read from LDS feeds ONE MFMA, so the
// generic kernel (conv_mfma.hip) moves 1.5 KiB LDS -> VGPR per 32x32x16 MFMA plus a ds_write pass per chunk and is
// LDS-bound at ~1/3 of its MFMA cycles. Here
//   * a pixel fragment = 16 columns of tile row r and 16 columns of row r+4, so the fragments of the three vertical
//     taps of neighbouring output rows COINCIDE: a wave that owns 8 output rows x 16 columns reads 6 input fragments per
//     horizontal tap and uses each for up to 3 MFMAs (0.75 LDS reads per MFMA at 32 channels, 0.5 at 64);
//   * the two k-groups of a 32-channel chunk go to two wave pairs (split K inside the workgroup, summed through LDS at
//     the end), which keeps the 16x16-pixel workgroup tile of the generic kernel (crops are ~40 px wide);
//   * global -> LDS staging is LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write pass; the LDS image is
//     lane-linear, bank conflicts are removed by XOR-swizzling the SOURCE channel slot and by rotating the second row's
//     columns by 8 (row pitch 18: 4 rows = 72 px = 8 mod 16); zero padding reads a 16-byte zero block.
// Workgroup = 4 waves = (spatial half sh) x (k-group kh); LDS = NST stages of {18x18 px x 32 ch | 18*NIW weight fragments}.
#include "conv_args.hpp"

namespace ffp {

namespace {

template <int NIW> struct RowsGeo {
  static constexpr int IN_PIECES = 21;                        // 324 px x 4 slots x 16 B = 1296 vectors = 20.25 KiB, issued as 1-KiB pieces
  static constexpr int IN_BYTES = IN_PIECES * 1024;
  static constexpr int W_FRAGS = NIW * 18;                    // 9 taps x 2 k-groups x NIW 32-channel tiles
  static constexpr int DUMMY = IN_BYTES + W_FRAGS * 1024;     // 1 KiB that absorbs the padding pieces (keeps every wave's DMA count equal)
  static constexpr int STAGE = DUMMY + 1024;
  static constexpr int IN_PER_WAVE = 6;                       // pieces wave w issues: w, w+4, ... (>= 21: dummy)
  static constexpr int W_PER_WAVE = (W_FRAGS + 3) / 4;
  static constexpr int CPW = IN_PER_WAVE + W_PER_WAVE;        // LDS-DMA instructions per wave and chunk
  static constexpr int XBUF = 4 * 2 * NIW * 4096;             // split-K exchange: per wave 2*NIW accumulator fragments
  static constexpr int EROW = 32 * 4 + 16;
  static constexpr int EBUF = 4 * 32 * EROW;
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  // LDS[M0 + lane*16 .. +16) <- global[gsrc .. +16) per lane; completion is counted on vmcnt by hand (hipcc does not see it)
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int NIW, int NST>
__global__ void __launch_bounds__(256, NIW == 1 ? 2 : 1) conv_rows_kernel(const ConvArgs a) {
  using G = RowsGeo<NIW>;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sh = wave & 1, kh = wave >> 1;
  const int p = lane & 31, hh = lane >> 5;

  // the channel blocks of one pixel tile are neighbours in the (XCD-contiguous) logical order: they share the input tile in L2
  const int nwg = live_workgroups(a);
  if ((int)blockIdx.x >= nwg) return;                  // capacity-sized grid, smaller batch
  const int L = xcd_remap(blockIdx.x, nwg);
  const int nb = L % a.n_nblk, tile = L / a.n_nblk;
  const int nt0 = nb * NIW;                     // first 32-channel tile of this workgroup
  const int4 t = a.tiles[tile];
  const int4 it = a.in_tab[t.x], ot = a.out_tab[t.x];
  const int oy0 = t.y, ox0 = t.z;
  const int Hi = it.y, Wi = it.z, Ho = ot.y, Wo = ot.z;
  const long long in_base = it.x, out_base = ot.x;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int Hv = Hi << a.up, Wv = Wi << a.up;

  // bias first: the oldest loads retire first, every later vmcnt wait is about the DMA pieces only
  float4 bias_r[NIW][4];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias_r[ni][g] = *reinterpret_cast<const float4*>(a.bias + (nt0 + ni) * 32 + 8 * g + 4 * hh);

  // ---- LDS-DMA sources: chunk-independent part per lane -------------------------------------------------------
  const unsigned char* img = reinterpret_cast<const unsigned char*>(a.in) + ((long long)in_base * a.in_cs + a.in_coff) * 2;
  const unsigned char* zeros = reinterpret_cast<const unsigned char*>(a.zeros);
  unsigned isrc[G::IN_PER_WAVE];          // byte offset of (pixel, swizzled slot) from `img`; 0xFFFFFFFF: zero block
#pragma unroll
  for (int i = 0; i < G::IN_PER_WAVE; ++i) {
    const int q = wave + 4 * i;
    const int L = q * 64 + lane, px = L >> 2, slot = (L & 3) ^ ((px >> 2) & 3);
    const int hy = px / 18, hx = px - hy * 18;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = q < G::IN_PIECES && px < 324 && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
    isrc[i] = ok ? (unsigned)(((long long)(iy >> a.up) * Wi + (ix >> a.up)) * a.in_cs * 2 + slot * 16) : 0xFFFFFFFFu;
  }
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wpk) + lane * 16;

  auto issue = [&](int c, int stage) {
    const unsigned sb = lds0 + stage * G::STAGE;
#pragma unroll
    for (int i = 0; i < G::IN_PER_WAVE; ++i) {
      const int q = wave + 4 * i;
      const unsigned char* src = isrc[i] != 0xFFFFFFFFu ? img + isrc[i] + c * 64 : zeros;
      glds16(src, sb + (q < G::IN_PIECES ? q * 1024 : G::DUMMY));
    }
#pragma unroll
    for (int i = 0; i < G::W_PER_WAVE; ++i) {
      const int f = wave + 4 * i;                    // LDS fragment slot: (ni*9 + tap)*2 + kg
      if (f < G::W_FRAGS) {
        const int kg = f & 1, nt_tap = f >> 1;       // nt_tap = ni*9 + tap
        glds16(wsrc + ((long long)(nt0 * 9 + nt_tap) * a.ncg + (2 * c + kg)) * 1024, sb + G::IN_BYTES + f * 1024);
      } else {
        glds16(zeros, sb + G::DUMMY);
      }
    }
  };

  // ---- LDS read offsets of this lane's 18 pixel fragments (input fragment j = rows 8sh+j and 8sh+4+j, tap column kx) ----
  unsigned boff[6][3];
  {
    const int col = (p & 15) ^ ((p >> 4) << 3), row0 = 8 * sh + 4 * (p >> 4), slot = 2 * kh + hh;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int px = (row0 + j) * 18 + col + kx;
        boff[j][kx] = px * 64 + ((slot ^ ((px >> 2) & 3)) << 4);
      }
  }
  const unsigned aoff = G::IN_BYTES + kh * 1024 + lane * 16;

  f32x16 acc[NIW][4];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][i][r] = 0.f;

  auto compute = [&](const unsigned char* sb) {
    uint4 bq[2][6], aq[2][3 * NIW];
    auto ld = [&](int kx, int q) {
#pragma unroll
      for (int j = 0; j < 6; ++j) bq[q][j] = *reinterpret_cast<const uint4*>(sb + boff[j][kx]);
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
          aq[q][ni * 3 + ky] = *reinterpret_cast<const uint4*>(sb + aoff + ((ni * 9 + ky * 3 + kx) * 2) * 1024);
    };
    ld(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int q = kx & 1;
      if (kx < 2) ld(kx + 1, q ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ni = 0; ni < NIW; ++ni) {
            union { uint4 u; f16x8 h; } ua, ub;
            ua.u = aq[q][ni * 3 + ky]; ub.u = bq[q][i + ky];
            acc[ni][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ua.h, ub.h, acc[ni][i], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop: chunk c lives in stage c % NST; NST-1 chunks are in flight ahead of the one being multiplied ----
  const int NC = a.cin >> 5;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < NC) issue(s, s);
  int stage = 0;
  for (int c = 0; c < NC; ++c) {
    // my pieces of chunk c have landed once at most the pieces of the (NST-2) younger chunks are outstanding
    if (NST > 2 && c + NST - 2 < NC) wait_vm<(NST > 2 ? (NST - 2) : 0) * G::CPW>();
    else if (NST > 3 && c + NST - 3 < NC) wait_vm<(NST > 3 ? (NST - 3) : 0) * G::CPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();          // everyone's pieces of chunk c are in LDS; everyone is done reading chunk c-1's stage
    if (c + NST - 1 < NC && !(a.dbg & 4)) {
      int ns = stage + NST - 1;
      if (ns >= NST) ns -= NST;
      issue(c + NST - 1, ns);
    }
    if (!(a.dbg & 2)) compute(smem + stage * G::STAGE);
    if (++stage == NST) stage = 0;
  }
  __syncthreads();

  // ---- split-K: each wave hands two of its four fragments (per channel tile) to its partner and sums the other two ----
  // kh = 0 keeps fragments {0,1}, kh = 1 keeps {2,3}
  unsigned char* xb = smem;
  {
    unsigned char* mine = xb + wave * (2 * NIW * 4096) + lane * 16;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          float4 v;
          if (kh == 0) v = make_float4(acc[ni][2 + g][4 * r4], acc[ni][2 + g][4 * r4 + 1], acc[ni][2 + g][4 * r4 + 2], acc[ni][2 + g][4 * r4 + 3]);
          else v = make_float4(acc[ni][g][4 * r4], acc[ni][g][4 * r4 + 1], acc[ni][g][4 * r4 + 2], acc[ni][g][4 * r4 + 3]);
          *reinterpret_cast<float4*>(mine + ((g * NIW + ni) * 4 + r4) * 1024) = v;
        }
  }
  __syncthreads();
  f32x16 fin[NIW][2];
  {
    const unsigned char* theirs = xb + (wave ^ 2) * (2 * NIW * 4096) + lane * 16;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const float4 v = *reinterpret_cast<const float4*>(theirs + ((g * NIW + ni) * 4 + r4) * 1024);
          const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            // sum in k order: the kh = 0 partial first
            if (kh == 0) fin[ni][g][4 * r4 + e] = acc[ni][g][4 * r4 + e] + o[e];
            else fin[ni][g][4 * r4 + e] = o[e] + acc[ni][2 + g][4 * r4 + e];
          }
        }
  }

  if (a.dbg & 1) return;

  // ---- epilogue: bias + activation, transposed through a wave-private LDS tile into whole-pixel 16-byte stores -------
  constexpr int EROW = G::EROW, CPL = 8, LPP = 4, PPI = 16;
  unsigned char* et = smem + G::XBUF + wave * (32 * EROW);
  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto uniform_ptr = [](const unsigned char* q) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wpk);
  const unsigned char* r1b = a.res1 ? reinterpret_cast<const unsigned char*>(a.res1) + ((long long)out_base * a.r1_cs + a.r1_coff) * 2 : wb;
  const unsigned char* r2b = a.res2 ? reinterpret_cast<const unsigned char*>(a.res2) + ((long long)out_base * a.r2_cs + a.r2_coff) * 2 : wb;
  const auto rs_r1 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r1b), 0, 0x7FFFFFF0, 0x00020000);
  const auto rs_r2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r2b), 0, 0x7FFFFFF0, 0x00020000);
  auto bload = [](decltype(rs_r1) rs, unsigned off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int fi = 2 * kh + g;                       // fragment index inside the wave's 8-row half
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      long long gps[2];
      bool oks[2];
      uint4 r1v[2], r2v[2];
#pragma unroll
      for (int itr = 0; itr < 2; ++itr) {
        const int pp = itr * PPI + lane / LPP, ch0 = (lane % LPP) * CPL;
        const int oy = oy0 + 8 * sh + fi + 4 * (pp >> 4), ox = ox0 + ((pp & 15) ^ ((pp >> 4) << 3));
        oks[itr] = oy < Ho && ox < Wo;
        gps[itr] = out_base + (long long)oy * Wo + ox;
        r1v[itr] = make_uint4(0u, 0u, 0u, 0u); r2v[itr] = r1v[itr];
        const unsigned rel_px = (unsigned)(gps[itr] - out_base);
        if (a.res1) r1v[itr] = bload(rs_r1, oks[itr] ? (rel_px * a.r1_cs + (nt0 + ni) * 32 + ch0) * 2 : OOB);
        if (a.res2) r2v[itr] = bload(rs_r2, oks[itr] ? (rel_px * a.r2_cs + (nt0 + ni) * 32 + ch0) * 2 : OOB);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bv = bias_r[ni][q];
        float4 v;
        v.x = apply_act(fin[ni][g][4 * q + 0] + bv.x, a.act);
        v.y = apply_act(fin[ni][g][4 * q + 1] + bv.y, a.act);
        v.z = apply_act(fin[ni][g][4 * q + 2] + bv.z, a.act);
        v.w = apply_act(fin[ni][g][4 * q + 3] + bv.w, a.act);
        *reinterpret_cast<float4*>(et + p * EROW + (8 * q + 4 * hh) * 4) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int itr = 0; itr < 2; ++itr) {
        const int pp = itr * PPI + lane / LPP, ch0 = (lane % LPP) * CPL;
        if (!oks[itr]) continue;
        float v[CPL];
#pragma unroll
        for (int q = 0; q < CPL / 4; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(et + pp * EROW + (ch0 + 4 * q) * 4);
          v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
        }
        if (a.res1) {
          const _Float16* r = reinterpret_cast<const _Float16*>(&r1v[itr]);
#pragma unroll
          for (int q = 0; q < CPL; ++q) v[q] = v[q] * a.s1 + (float)r[q];
        }
        if (a.res2) {
          const _Float16* r = reinterpret_cast<const _Float16*>(&r2v[itr]);
#pragma unroll
          for (int q = 0; q < CPL; ++q) v[q] = v[q] * a.s2 + (float)r[q];
        }
        uint4 ov;
        _Float16* o = reinterpret_cast<_Float16*>(&ov);
#pragma unroll
        for (int q = 0; q < CPL; ++q) o[q] = (_Float16)v[q];
        *reinterpret_cast<uint4*>(reinterpret_cast<_Float16*>(a.out) + (size_t)gps[itr] * a.out_cs + a.out_coff + (nt0 + ni) * 32 + ch0) = ov;
      }
    }
  }
}

template <int NIW, int NST> struct RowsCfg {
  using G = RowsGeo<NIW>;
  static constexpr int LDS_MAIN = NST * G::STAGE, LDS_EPI = G::XBUF + G::EBUF;
  static constexpr int LDS = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(LDS <= 160 * 1024, "rows kernel: LDS");
  static void init() {
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_kernel<NIW, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  static void launch(ConvArgs& a, Level* out_lvl, hipStream_t st) {
    int n_tiles = 0;
    a.tiles = out_lvl->tile_table(16, &n_tiles, &a.n_tiles_dev, st);
    if (n_tiles == 0) return;
    a.n_nblk = a.ntiles32 / NIW;
    hipLaunchKernelGGL((conv_rows_kernel<NIW, NST>), dim3(n_tiles * a.n_nblk), dim3(256), LDS, st, a);
  }
};

}  // namespace

void conv_rows_init() {
  RowsCfg<1, 2>::init(); RowsCfg<1, 3>::init(); RowsCfg<2, 2>::init();
}

bool conv_rows_eligible(const ConvOp& op, const ConvArgs& a) {
  const PackedConv& pc = *op.pc;
  if (a.force_shape >= 0 && a.force_shape < 6) return false;        // tuning: a generic shape was asked for
  return pc.dt == F16 && pc.k == 3 && op.stride == 1 && pc.cin % 32 == 0 && pc.cin >= 64 && pc.cin == pc.cin_pad &&
         pc.cout % 32 == 0 && pc.cout <= 128 && a.fast_out && a.zeros != nullptr;
}

void launch_conv_rows(ConvArgs& a, Level* out_lvl, hipStream_t st) {
  // 32 channels per workgroup, two stages = 80 KiB: two workgroups per CU. (A 64-channel workgroup halves the LDS
  // reads per MFMA again but needs 128 accumulator registers + 116 KiB, i.e. one workgroup per CU: measured 1.5x slower
  // on the 192->64 dense-block conv, as is a third stage at 32 channels. force_shape 7 / 8 keep them reachable for tuning.)
  if (a.force_shape == 7) RowsCfg<1, 3>::launch(a, out_lvl, st);
  else if (a.force_shape == 8 && a.ntiles32 % 2 == 0) RowsCfg<2, 2>::launch(a, out_lvl, st);
  else RowsCfg<1, 2>::launch(a, out_lvl, st);
}

}  // namespace ffp
