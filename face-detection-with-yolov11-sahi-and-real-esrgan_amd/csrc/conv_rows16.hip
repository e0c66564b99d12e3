// conv_rows16.hip — 3x3 stride-1 fp16 convolution with 32 / 64 output channels on v_mfma_f32_16x16x32_f16: the
// Real-ESRGAN dense-block convs (every RRDBNet conv except conv_first / conv_last), second generation.
//
// What bounded the first row-reuse kernel (conv_rows.hip, 32x32x16 MFMA, split K inside the workgroup, one pixel tile per
// workgroup), measured with its phase switches and with SQ counters: per CU it finished a tile every 6.3 us against 2.4 us
// of MFMA time — the waves spent two thirds of their life parked or issue-stalled: first-load latency and store drain of
// every one-tile workgroup, a split-K exchange + transposition through LDS in the epilogue, LDS-DMA pieces at 60-185 issue
// cycles each, staging / MFMA / epilogue as separate phases of every wave. This kernel:
//   * MFMA 16x16x32: K = 32 is one whole input-channel chunk, so NO split K; a pixel fragment is 16 columns of ONE tile
//     row, so the fragment of input row r serves the three vertical taps of output rows r-1, r, r+1: a wave owns 4 output
//     rows x 16 columns and reads 6 input-row fragments + 6 weight fragments per horizontal tap for 24 MFMAs;
//   * the accumulator layout (lane = pixel column x channel group g, 4 registers = 4 channels) together with a
//     permutation of the weight rows at pack time (M-tile m, row 4g+r -> output channel 8g+4m+r) leaves each lane with
//     8 CONSECUTIVE output channels of one pixel: bias, activation, residuals and a 16-byte buffer store straight from
//     registers — no LDS in the epilogue, no branch (out-of-image lanes store to the out-of-range offset);
//   * the grid is persistent-ish (8 x Ws workgroups walk all pixel tiles of the launch) and the chunks of consecutive
//     tiles form ONE software pipeline: staging is global -> registers (raw buffer loads, out-of-image lanes read zeros)
//     -> ds_write_b128, three chunks ahead of the MFMAs, and its instructions are issued BETWEEN the MFMAs of the chunk
//     being multiplied (one piece per MFMA step), not in a phase of their own; stores are never waited for.
// LDS image of a chunk: [18 x 18 (+12 dummy) halo pixels][4 slots of 8 channels], slot index XOR ((column >> 1) & 2): the
// fragment reads (ds_read_b128, lane = column x slot) are bank-conflict free for every row and horizontal tap
// (SQ_LDS_BANK_CONFLICT = 0 measured); weight fragments are lane-linear. Two stages of 39 KiB: two workgroups per CU.
#include <algorithm>
#include <cstdlib>

#include "conv_args.hpp"

#ifndef FFP_R16_DBG
#define FFP_R16_DBG 0          // 1: also build the phase-skip instantiations of the kernel (tools/rows16_phase_probe.py); compile-time masks, so
#endif                         //    that what is left keeps the production kernel's instruction schedule

#ifndef FFP_R16_PRIO
#define FFP_R16_PRIO 0         // 1: s_setprio 1 over the MFMA stream of a chunk (A/B)
#endif
#ifndef FFP_R16_STASH
#define FFP_R16_STASH 0        // experiment: 1 = the next chunk's ten pieces are written to LDS and re-requested BEFORE the chunk's MFMA stream, 2 = AFTER it
#endif                         //             (0: one piece per MFMA step, the shipped schedule)
#ifndef FFP_R16_STORE_AUX
#define FFP_R16_STORE_AUX 0   // cache policy of the output stores (experiment): 16 = sc1 (write-through: nothing dirty left in L2 for the kernel-end write-back), 17 = sc0 sc1, 2 = nt
#endif
#ifndef FFP_R16_STAMP
#define FFP_R16_STAMP 0        // 1: diagnostic build — s_memtime stamps around the phases of every chunk, sums printed by the first workgroups
#endif                         //    (MI355X guide, "In-kernel stamps"); never in a shipped build, the stamps cost ~10 % of the kernel

namespace ffp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct R16Geo {
  static constexpr int IN_PX = 336;                           // 18 x 18 halo pixels + 12 dummies: 21 whole wave-pieces of 64 vectors
  static constexpr int IN_BYTES = IN_PX * 64;                 // 21504
  static constexpr int W_BYTES = 9 * 2 * 1024;                // [tap][M-tile][lane] : 18 wave-pieces
  static constexpr int STAGE = IN_BYTES + W_BYTES;            // 39936
  static constexpr int NP = 10;                               // wave-pieces per wave and chunk: 5 input, 4 weight, 1 mixed (see piece_fetch)
  static constexpr int TCAP = 40;                             // work items one workgroup walks (descriptor table in LDS)
  static constexpr int DESC = 2 * STAGE;                      // int4 x 3 per item: tile {img, y0, x0}, input {base, h, w}, output {base, h, w}
  static constexpr int LDS = 2 * STAGE + TCAP * 48;           // 81792 <= 80 KiB
  // weight-resident form (RES): [input stage 0][input stage 1][all chunks' weight fragments of the workgroup's channel block][descriptors];
  // one workgroup per CU, the weights are fetched ONCE per workgroup instead of once per tile
  static constexpr int NPR = 6;                               // staging slots per wave and chunk: 5 input pieces + input piece 20 (every wave, same bytes)
  static constexpr int RES_W = 2 * IN_BYTES;                  // offset of the resident weights
  static constexpr int RES_MAXC = 6;                          // chunks (cin <= 192)
  static constexpr int res_lds(int nc) { return 2 * IN_BYTES + nc * W_BYTES + TCAP * 48; }
};

// Items are dealt so that each XCD (workgroups b, b + 8, ... share one) gets a contiguous run of logical ids (pixel tile x
// 32-channel block) and the workgroups of an XCD take consecutive ids at the same time: the channel blocks of a pixel tile
// and neighbouring tiles meet in that XCD's L2. The mapping only affects speed.
template <int DBG, bool RES = false>             // phase-skip bits: 1 epilogue, 2 MFMA, 4 staging requests, 8 staging LDS writes, 16 fragment reads, 32 barriers
__global__ void __launch_bounds__(256, (RES ? 1 : 2)) conv_rows16_kernel(const ConvArgs a) {
  using G = R16Geo;
  constexpr int NP = RES ? G::NPR : G::NP;
  constexpr int STG = RES ? G::IN_BYTES : G::STAGE;            // bytes of one LDS stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pc = lane & 15, g = lane >> 4;             // pixel column inside the tile, channel group / k slot

  // ---- this workgroup's items: logical ids first + j * Ws, j < J --------------------------------------------------------------
  const int n_items = (a.n_tiles_dev ? __builtin_amdgcn_readfirstlane(*a.n_tiles_dev) : a.ntiles_host) * a.n_nblk;
  const int Ws = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int per_xcd = (n_items + 7) >> 3;
  per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  const int first = xcd * per_xcd + slot;
  const int last = min(n_items, (xcd + 1) * per_xcd);
  const int J = first < last ? min((last - first + Ws - 1) / Ws, G::TCAP) : 0;      // the launcher sizes Ws so that J <= TCAP
  if (J == 0) return;
  const int nt0 = first % a.n_nblk;                    // Ws and per_xcd are multiples of n_nblk: ONE channel block per workgroup

  const int NC = a.cin >> 5;
  int4* desc = reinterpret_cast<int4*>(smem + (RES ? G::RES_W + NC * G::W_BYTES : G::DESC));
  if (tid < J) {
    const int4 t = a.tiles[(first + tid * Ws) / a.n_nblk];
    desc[tid * 3] = t;
    desc[tid * 3 + 1] = a.in_tab[t.x];
    desc[tid * 3 + 2] = a.out_tab[t.x];
  }
  __syncthreads();

  constexpr unsigned OOB = 0xFFFFFFFFu;
  auto uniform_ptr = [](const unsigned char* q) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo);
  };
  auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wpk) + (long long)nt0 * NC * G::W_BYTES;
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wb), 0, 0x7FFFFFF0, 0x00020000);
  auto rs_in = rs_w;                                   // rebuilt per item (setup_pf)
  auto rs_wp = rs_w;                                   // the prefetcher's view of the weights (emptied when the stream ends)
  auto bload = [](decltype(rs_w) rs, unsigned off, int soff = 0) {      // soff: wave-uniform byte offset (an SGPR operand, no VALU)
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };

  // ---- staging: a chunk = 21 input + 18 weight wave-pieces (64 lanes x 16 bytes); wave w stages, in its NP slots,
  //   slots 0..4: input pieces w, w+4, .., w+16; slots 5..8: weight pieces w, w+4, w+8, w+12;
  //   slot 9: wave 0 input piece 20, wave 1 / 2 weight pieces 16 / 17, wave 3 weight piece 17 again (same bytes to the same place)
  // so every slot is a whole wave-instruction for every wave: no divergence, no validity tests in the pipeline.
  const bool s9_in = wave == 0;
  const int s9_piece = wave == 0 ? 20 : wave == 1 ? 16 : 17;
  unsigned isrc[6];        // byte offset of (pixel, slot) of the item being PREFETCHED from its image base; OOB: zeros
  unsigned idst[6];        // LDS byte offset of that vector inside the stage (item independent)
  unsigned wrel[5];        // weight pieces: byte offset inside a chunk's 18 KiB, also the LDS offset behind IN_BYTES
  unsigned hyx[6];         // the slot's halo pixel and vector slot (slot << 16 | row << 8 | column): item independent, so that an item's set-up is adds and compares
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int idx = (i < 5 ? wave + 4 * i : 20) * 64 + lane;
    const int px = idx >> 2, s = idx & 3;
    const int hy = px / 18, hx = px - hy * 18;
    idst[i] = (unsigned)(px * 64 + ((s ^ ((hx >> 1) & 2)) << 4));
    hyx[i] = px < 324 ? (unsigned)((s << 16) | (hy << 8) | hx) : (unsigned)((s << 16) | 0xFFFF);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) wrel[i] = (unsigned)(((i < 4 ? wave + 4 * i : s9_piece) * 64 + lane) * 16);
  // weights in global memory: [32-channel tile][chunk][tap][M-tile][lane] — a chunk's 18 KiB are laid out like the LDS stage, so one
  // per-lane offset serves the load (chunk offset as the instruction's scalar operand) and the LDS write

  int pf_item = 0, pf_c = 0;                           // prefetch cursor: three chunks ahead of the MFMAs
  int cj = 0, cc = 0;                                  // compute cursor: item, chunk
#if FFP_R16_STAMP
  unsigned long long st_t1 = 0, st_sum[4] = {0, 0, 0, 0};     // exposed first fragment reads, MFMA stream, epilogue + bookkeeping, barrier
#endif
  auto setup_pf = [&](int j) {
    const int4 t = desc[j * 3], it = desc[j * 3 + 1];
    const int oy0 = sgpr(t.y), ox0 = sgpr(t.z), Hi = sgpr(it.y), Wi = sgpr(it.z);
    const int Hv = Hi << a.up, Wv = Wi << a.up;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + ((long long)sgpr(it.x) * a.in_cs + a.in_coff) * 2;
    rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(inb), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 6; ++i) {                      // hyx[i]: the slot's halo pixel (row << 8 | column) and vector slot, fixed per lane; 0xFFFF: padding pixel
      const int hy = (int)(hyx[i] >> 8) & 0xFF, hx = (int)hyx[i] & 0xFF, sl = (int)(hyx[i] >> 16) & 3;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = (hyx[i] & 0xFFFFu) != 0xFFFFu && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
      isrc[i] = ok ? (unsigned)(((iy >> a.up) * Wi + (ix >> a.up)) * a.in_cs * 2 + sl * 16) : OOB;
    }
  };
  uint4 ra[NP] = {}, rb[NP] = {};                      // two register sets
  auto piece_fetch = [&](int p, uint4& r) {            // slot p of the prefetch cursor's chunk -> r
    if (DBG & 4) return;
    if constexpr (RES) {                               // input pieces only: slots 0..4 as below, slot 5 = input piece 20 for every wave
      const unsigned o = isrc[p];
      r = bload(rs_in, o != OOB ? o + (unsigned)(pf_c * 64) : OOB);
      return;
    }
    if (p < 5) {
      r = bload(rs_in, isrc[p] != OOB ? isrc[p] + (unsigned)(pf_c * 64) : OOB);
    } else if (p < 9) {
      r = bload(rs_wp, wrel[p - 5], pf_c * G::W_BYTES);
    } else {                                           // the mixed slot: wave-uniform choice, no branch
      const unsigned oi = isrc[5] != OOB ? isrc[5] + (unsigned)(pf_c * 64) : OOB;
      r = bload(s9_in ? rs_in : rs_wp, s9_in ? oi : wrel[4], s9_in ? 0 : pf_c * G::W_BYTES);
    }
  };
  auto piece_stash = [&](unsigned char* sb, int p, const uint4& r) {
    if (DBG & 8) { asm volatile("" :: "v"(r.x), "v"(r.y), "v"(r.z), "v"(r.w)); return; }      // the requests stay alive without the writes
    if constexpr (RES) { *reinterpret_cast<uint4*>(sb + idst[p]) = r; return; }
    if (p < 5) *reinterpret_cast<uint4*>(sb + idst[p]) = r;
    else if (p < 9) *reinterpret_cast<uint4*>(sb + G::IN_BYTES + wrel[p - 5]) = r;
    else *reinterpret_cast<uint4*>(sb + (s9_in ? idst[5] : G::IN_BYTES + wrel[4])) = r;
  };
  auto advance_pf = [&]() {                            // after the last piece of a chunk has been requested
    if (pf_item >= J) return;
    if (++pf_c == NC) {
      pf_c = 0;
      if (++pf_item < J) {
        setup_pf(pf_item);
      } else {                                         // stream exhausted: every later request is out of range (zeros, no traffic)
        rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wb), 0, 0, 0x00020000);
        rs_wp = rs_in;
      }
    }
  };
  auto fetch_all = [&](uint4 (&q)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) piece_fetch(p, q[p]);
    advance_pf();
  };
  auto stash_all = [&](unsigned char* sb, const uint4 (&q)[NP]) {
#pragma unroll
    for (int p = 0; p < NP; ++p) piece_stash(sb, p, q[p]);
  };

  // bias of this lane's 8 channels (consumed in the epilogues)
  const float4 bias0 = *reinterpret_cast<const float4*>(a.bias + nt0 * 32 + 8 * g);
  const float4 bias1 = *reinterpret_cast<const float4*>(a.bias + nt0 * 32 + 8 * g + 4);

  // ---- fragment read offsets: input rows 4*wave + j (j = 0..5: halo rows), column pc + kx, slot g ------------------------------
  unsigned boff[3];        // per kx, for halo row 4*wave; rows are 18 * 64 = 1152 bytes apart
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = pc + kx;
    boff[kx] = (unsigned)(((4 * wave) * 18 + hx) * 64 + ((g ^ ((hx >> 1) & 2)) << 4));
  }
  const unsigned aoff = G::IN_BYTES + lane * 16;

  f32x4 acc[4][2];
  auto zero_acc = [&]() {                              // accumulation starts from the bias: nothing to add in the epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i][0] = f32x4{bias0.x, bias0.y, bias0.z, bias0.w};
      acc[i][1] = f32x4{bias1.x, bias1.y, bias1.z, bias1.w};
    }
  };
  zero_acc();

  // One step = one tap (kx, ky): 2 weight fragments (M-tiles) x 4 output rows = 8 MFMAs. The weight fragments of step s+2 and
  // (at ky = 0) the 6 input-row fragments of the next kx are requested from LDS ahead of the MFMAs that use them. The staging of
  // the stream rides in the same instruction stream: in step s, slot s (and slot 9 in step 8) of the NEXT chunk (register set
  // `set`, requested two chunks ago) is written to the other LDS stage and the same register is re-requested for the chunk three
  // ahead — 10 ds_write + 10 buffer loads per chunk, issued between MFMAs instead of in a phase of their own.
  auto chunk = [&](const unsigned char* sb, unsigned char* sbn, uint4 (&set)[NP]) {
    uint4 bq[2][6] = {}, aq[3][2] = {};
    const unsigned char* wres = smem + (G::RES_W - G::IN_BYTES) + cc * G::W_BYTES;       // RES: this chunk's resident weights (aoff carries IN_BYTES)
    auto ldB = [&](int kx, int q) {
      if (DBG & 16) {                                          // opaque operands instead of fragment reads
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(bq[q][j].x), "+v"(bq[q][j].y), "+v"(bq[q][j].z), "+v"(bq[q][j].w));
        return;
      }
#pragma unroll
      for (int j = 0; j < 6; ++j) bq[q][j] = *reinterpret_cast<const uint4*>(sb + boff[kx] + j * 1152);
    };
    auto ldA = [&](int s, int q) {                             // step s = (kx, ky) = (s / 3, s % 3); packed tap index ky * 3 + kx
      const int tap = (s % 3) * 3 + s / 3;
      if (DBG & 16) {
#pragma unroll
        for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(aq[q][m].x), "+v"(aq[q][m].y), "+v"(aq[q][m].z), "+v"(aq[q][m].w));
        return;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) aq[q][m] = *reinterpret_cast<const uint4*>((RES ? wres : sb) + aoff + ((tap * 2 + m) << 10));
    };
#if FFP_R16_STASH == 1
#pragma unroll
    for (int p = 0; p < NP; ++p) { piece_stash(sbn, p, set[p]); piece_fetch(p, set[p]); }
    __builtin_amdgcn_sched_barrier(0);
#endif
    ldB(0, 0);
    ldA(0, 0);
    ldA(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#if FFP_R16_PRIO
    __builtin_amdgcn_s_setprio(1);                             // the MFMA stream outranks the partner wave's epilogue / set-up VALU work
#endif
#if FFP_R16_STAMP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st_t1 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int kx = s / 3, ky = s - 3 * kx;
      if (s + 2 < 9) ldA(s + 2, (s + 2) % 3);
      if (ky == 0 && kx < 2) ldB(kx + 1, (kx + 1) & 1);
      // staging slots per step: {0} {1} {2} {3} {4} {5} {6} {7} {8, 9}
#if FFP_R16_STASH == 0
#pragma unroll
      for (int p = s; p < (RES ? (s < NP ? s + 1 : s) : (s == 8 ? 10 : s + 1)); ++p) {
        piece_stash(sbn, p, set[p]);
        piece_fetch(p, set[p]);
      }
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if (DBG & 2) continue;
          union { uint4 u; f16x8 h; } ua, ub;
          ua.u = aq[s % 3][m]; ub.u = bq[kx & 1][i + ky];
          acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, acc[i][m], 0, 0, 0);
        }
      // one MFMA, then up to two of the other instructions, eight times: the LDS / memory / address work sits in the MFMAs' issue shadow
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x7F6, 2, 0);     // anything else (VALU, SALU, VMEM, DS)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#if FFP_R16_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if FFP_R16_STASH == 2
#pragma unroll
    for (int p = 0; p < NP; ++p) { piece_stash(sbn, p, set[p]); piece_fetch(p, set[p]); }
#endif
    advance_pf();
  };

  // ---- epilogue of item j: lane (pc, g) holds channels 8g..8g+7 of pixel (row 4*wave + i, column pc).
  // Stores are not waited for: they drain while the next item is multiplied. --------------------------------------------------------
  auto epilogue = [&](int j) {
    const int4 t = desc[j * 3], ot = desc[j * 3 + 2];
    const int oy0 = sgpr(t.y), ox0 = sgpr(t.z), Ho = sgpr(ot.y), Wo = sgpr(ot.z);
    const long long out_base = sgpr(ot.x);
    const unsigned char* r1b = a.res1 ? reinterpret_cast<const unsigned char*>(a.res1) + (out_base * a.r1_cs + a.r1_coff) * 2 : wb;
    const unsigned char* r2b = a.res2 ? reinterpret_cast<const unsigned char*>(a.res2) + (out_base * a.r2_cs + a.r2_coff) * 2 : wb;
    unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + (out_base * a.out_cs + a.out_coff) * 2;
    const auto rs_r1 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r1b), 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_r2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r2b), 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ob), 0, 0x7FFFFFF0, 0x00020000);
    const int ox = ox0 + pc;
    const int ch0 = nt0 * 32 + 8 * g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = oy0 + 4 * wave + i;
      const bool ok = oy < Ho && ox < Wo;
      const unsigned rel_px = (unsigned)(oy * Wo + ox);
      uint4 r1v = make_uint4(0u, 0u, 0u, 0u), r2v = r1v;
      if (a.res1) r1v = bload(rs_r1, ok ? (rel_px * a.r1_cs + ch0) * 2 : OOB);
      if (a.res2) r2v = bload(rs_r2, ok ? (rel_px * a.r2_cs + ch0) * 2 : OOB);
      float v[8];
      v[0] = acc[i][0][0]; v[1] = acc[i][0][1]; v[2] = acc[i][0][2]; v[3] = acc[i][0][3];
      v[4] = acc[i][1][0]; v[5] = acc[i][1][1]; v[6] = acc[i][1][2]; v[7] = acc[i][1][3];
      if (a.act == ACT_LRELU) {                        // max(x, 0.2 x): two VALU ops per value
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], v[q] * 0.2f);
      } else if (a.act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = apply_act(v[q], ACT_SILU);
      }
      if (a.res1) {
        const _Float16* r = reinterpret_cast<const _Float16*>(&r1v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] * a.s1 + (float)r[q];
      }
      if (a.res2) {
        const _Float16* r = reinterpret_cast<const _Float16*>(&r2v);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = v[q] * a.s2 + (float)r[q];
      }
      union { u32x4 u; _Float16 h[8]; } ov;
#pragma unroll
      for (int q = 0; q < 8; ++q) ov.h[q] = (_Float16)v[q];
      __builtin_amdgcn_raw_buffer_store_b128(ov.u, rs_o, ok ? (rel_px * a.out_cs + ch0) * 2 : OOB, 0, FFP_R16_STORE_AUX);
    }
  };

  // ---- the chunk stream: chunk q (of all items, back to back) is multiplied out of stage q & 1 while chunk q+1 moves from its
  // register set into the other stage and chunks q+2, q+3 are in flight -----------------------------------------------------------------
  const int Q = J * NC;
  auto finish_chunk = [&]() {
    if (++cc == NC) {
      if (!(DBG & 1)) {
        epilogue(cj);
      } else {                                                   // keep the sums alive without the epilogue
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(acc[i][0]), "v"(acc[i][1]));
      }
      zero_acc();
      cc = 0; ++cj;
    }
  };
  if constexpr (RES) {
    // the workgroup's weights, all chunks, once: NC x 18 wave-pieces, eight in flight per wave; visible to all after the barrier below
    const int n_wp = NC * 18;
    for (int p0 = wave; p0 < n_wp; p0 += 32) {
      uint4 t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (p0 + 4 * k < n_wp) t[k] = bload(rs_w, (unsigned)lane * 16u, sgpr((p0 + 4 * k) << 10));
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (p0 + 4 * k < n_wp) *reinterpret_cast<uint4*>(smem + G::RES_W + ((p0 + 4 * k) << 10) + lane * 16) = t[k];
    }
  }
  setup_pf(0);
  fetch_all(ra);
  fetch_all(rb);
  stash_all(smem, ra);
  fetch_all(ra);
  __syncthreads();
  for (int q = 0; q < Q; q += 2) {
    // stage 0 holds chunk q; rb holds chunk q+1 (-> stage 1), its registers are re-requested for chunk q+3
#if FFP_R16_STAMP
#define R16_PHASES(SB, SBN, SET)                                                      \
    {                                                                                 \
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();                     \
      chunk(SB, SBN, SET);                                                            \
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();                     \
      finish_chunk();                                                                 \
      const unsigned long long t3 = __builtin_amdgcn_s_memtime();                     \
      __syncthreads();                                                                \
      const unsigned long long t4 = __builtin_amdgcn_s_memtime();                     \
      st_sum[0] += st_t1 - t0; st_sum[1] += t2 - st_t1; st_sum[2] += t3 - t2; st_sum[3] += t4 - t3; \
    }
    R16_PHASES(smem, smem + STG, rb)
    if (q + 1 >= Q) break;
    R16_PHASES(smem + STG, smem, ra)
#else
    chunk(smem, smem + STG, rb);
    finish_chunk();
    if (!(DBG & 32)) __syncthreads();
    if (q + 1 >= Q) break;
    chunk(smem + STG, smem, ra);
    finish_chunk();
    if (!(DBG & 32)) __syncthreads();
#endif
  }
#if FFP_R16_STAMP
  if (lane == 0 && (blockIdx.x % 37) == 0 && blockIdx.x < 512)
    printf("r16stamp wg %d wave %d chunks %d items %d : first_frag_reads %llu stream %llu (ideal %d) epilogue+setup %llu barrier %llu cycles per chunk\n", (int)blockIdx.x, wave, Q, J,
           st_sum[0] / Q, st_sum[1] / Q, 72 * 16, st_sum[2] / Q, st_sum[3] / Q);
#endif
}

}  // namespace

namespace {
#if FFP_R16_DBG
constexpr int kR16Masks[] = {0, 1, 2, 12, 16, 32, 13, 29, 61, 19, 31, 63};
#else
constexpr int kR16Masks[] = {0, 1};
#endif
template <int I = 0> void r16_for_mask(int mask, bool res, bool init, const ConvArgs* a, unsigned grid, int lds, hipStream_t st) {
  if constexpr (I < (int)(sizeof(kR16Masks) / sizeof(int))) {
    if (init) {
      FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows16_kernel<kR16Masks[I], false>), hipFuncAttributeMaxDynamicSharedMemorySize, R16Geo::LDS));
      FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows16_kernel<kR16Masks[I], true>), hipFuncAttributeMaxDynamicSharedMemorySize, R16Geo::res_lds(R16Geo::RES_MAXC)));
      r16_for_mask<I + 1>(mask, res, init, a, grid, lds, st);
    } else if (mask == kR16Masks[I]) {
      if (res) hipLaunchKernelGGL((conv_rows16_kernel<kR16Masks[I], true>), dim3(grid), dim3(256), lds, st, *a);
      else hipLaunchKernelGGL((conv_rows16_kernel<kR16Masks[I], false>), dim3(grid), dim3(256), lds, st, *a);
    } else {
      r16_for_mask<I + 1>(mask, res, init, a, grid, lds, st);
    }
  } else if (!init) {
    fail(FFP_ERR_ARG, "rows16: phase-skip mask %d is not built (FFP_R16_DBG)", mask);
  }
}
}  // namespace

void conv_rows16_init() { r16_for_mask<>(0, false, true, nullptr, 0, 0, nullptr); }

// the weight-resident form: force_shape 23, or every eligible layer with FFP_ROWS16_RES=1 (A/B aid)
static bool rows16_resident(const ConvArgs& a) {
  static const bool env_on = [] { const char* e = getenv("FFP_ROWS16_RES"); return e && e[0] == '1'; }();
  return (a.force_shape == 23 || (a.force_shape < 0 && env_on)) && (a.cin >> 5) <= R16Geo::RES_MAXC;
}

bool conv_rows16_eligible(const ConvOp& op, const ConvArgs& a) {
  const PackedConv& pc = *op.pc;
  if (a.force_shape >= 0 && a.force_shape != 9 && a.force_shape != 23 && a.force_shape != 24) return false;   // tuning: another kernel was asked for
  return pc.w16.p != nullptr && pc.dt == F16 && pc.k == 3 && op.stride == 1 && pc.cin % 32 == 0 && pc.cin >= 64 && pc.cout % 32 == 0 &&
         pc.cout <= 128 && a.fast_out && op.out.cs % 8 == 0 && op.out.coff % 8 == 0;
}

void launch_conv_rows16(ConvArgs& a, const PackedConv& pc, Level* out_lvl, hipStream_t st) {
  using G = R16Geo;
  static_assert(G::LDS <= 80 * 1024, "rows16 kernel: two workgroups per CU");
  a.wpk = pc.w16.p;
  int n_tiles = 0;
  a.tiles = out_lvl->tile_table(16, &n_tiles, &a.n_tiles_dev, st);     // n_tiles: the launch extent (capacity-mode levels: the capacity)
  if (n_tiles == 0) return;
  a.ntiles_host = n_tiles;
  a.n_nblk = a.ntiles32;
  // 8 x Ws workgroups, Ws a multiple of the channel blocks per tile (a workgroup then keeps ONE channel block: weights and bias
  // are item independent) and large enough that no workgroup walks more than TCAP items; Ws = 64 is two workgroups per CU
  const long long items = (long long)n_tiles * a.n_nblk;
  long long per_xcd = (items + 7) / 8;
  per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  const bool res = rows16_resident(a);
  static const int ws_env = [] { const char* e = getenv("FFP_ROWS16_WS"); return e ? atoi(e) : 0; }();      // experiment: workgroups per XCD (64 = two per CU)
  long long ws = std::max<long long>(ws_env > 0 ? ws_env : (res ? 32 : 64), (per_xcd + G::TCAP - 1) / G::TCAP);          // resident weights: one workgroup per CU
  ws = (ws + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  FFP_CHECK(8 * ws < (1ll << 31) && (per_xcd + ws - 1) / ws <= G::TCAP, FFP_ERR_STATE, "rows16: launch geometry");
  r16_for_mask<>(a.dbg, res, false, &a, (unsigned)(8 * ws), res ? G::res_lds(a.cin >> 5) : G::LDS, st);
}

}  // namespace ffp
