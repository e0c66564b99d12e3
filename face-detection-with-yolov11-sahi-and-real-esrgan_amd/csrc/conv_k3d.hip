// conv_k3d.hip — 3x3 convolutions (stride 1 and 2) of the detector in the split arithmetic (FFP_PREC_F32X3), second generation.
//
// What bounded the generic kernel (conv_mfma.hip) on these layers (profiles/r02_k3_phase_probe.txt, r02_pmc_util.json): a 128 px x
// 64 ch workgroup stages 36 KiB of weights + a 36 KiB halo tile through LDS per 16-channel chunk — as many LDS bytes written as the
// chunk's 54 MFMAs per wave can cover, plus 6 fragment reads per 6 MFMAs: the LDS is saturated at about half the matrix rate; one
// tile per workgroup pays first-load latency, LDS transposition and store drain with its slots idle; fetch / stash / MFMA / store
// are separate phases of every wave. 0.19-0.22 of the MFMA peak, 1.5-1.7x the algorithmic HBM traffic on the stride-2 layers.
// This kernel:
//   * weights never touch LDS: a wave's A operands (its NIW 32-channel tiles x hi / lo) are raw buffer loads straight from the packed
//     fragments in L2 into a register ring, at least 24 MFMAs ahead of their use (a 2 x 2 wave grid: a fragment is fetched by two
//     waves instead of being written to LDS once and read by four); LDS holds only the halo tile of the input, 64 bytes per pixel
//     and chunk, no padding: half the LDS traffic per MFMA, and 128 output channels per workgroup fit (the input is read once);
//   * LDS image [halo row][pixel][4 slots of 16 B: hi.k0-7, hi.k8-15, lo.k0-7, lo.k8-15], slot index XOR-swizzled by pixel column
//     and row; stride 2 stores the even and the odd columns of a row as two runs, so that the 16 pixels of a fragment row are
//     consecutive records for every horizontal tap: ds_read_b128 fragment reads are bank-conflict free for every tap at both
//     strides (tools/lds_layout_search.py searches the layouts exhaustively);
//   * persistent workgroups walk (pixel tile, channel block) items, each XCD a contiguous run (conv_rows16.hip's scheme), and the
//     16-channel chunks of consecutive items are ONE software pipeline: chunk q is multiplied out of stage q & 1 while the pieces of
//     chunk q + 1 (requested a whole chunk earlier) are split into fp16 hi + lo and written to the other stage, and their registers
//     re-requested for chunk q + 2 — one piece per MFMA step, between the MFMAs (sched_group_barrier), not in a phase of its own;
//     out-of-image pixels read zeros through the buffer range check (no branch);
//   * the epilogue (LDS transposition to whole 128-byte pixel rows, bias, scale, SiLU, residuals, max-|value| slot) of an item runs
//     while the next item's first chunks are already in registers / in the other stage; where a stage is large enough the
//     transposition reuses the stage just consumed instead of LDS of its own (two workgroups per CU at stride 2).
// The arithmetic is the generic kernel's, instruction for instruction: v_mfma_f32_32x32x16_f16 on the same packed fragments, k order
// = 16-channel group, then tap, then al*bh, ah*bl, ah*bh — results are bit-identical to conv_mfma_kernel<X3, 3, ...>, so the plan
// tuner may pick either per layer without changing one bit of the output.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_args.hpp"

#ifndef FFP_K3D_DBG
#define FFP_K3D_DBG 0          // 1: honour ConvArgs::dbg phase-skip bits (tools/k3d_phase_probe.py); the tests cost ~15 % of the kernel's speed
#endif

namespace ffp {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int S, int WM, int WN, int MI, int NIW> struct K3Geo {
  static_assert(WM * WN == 4 && (S == 1 || S == 2), "geometry");
  static constexpr int TH = 2 * WM * MI;                     // output tile rows (16 columns)
  static constexpr int HH = (TH - 1) * S + 3, HWD = 15 * S + 3;
  static constexpr int RP = HWD;                              // records per halo row
  static constexpr int PP = 17;                               // stride 2: offset of the odd-column run inside a row
  static constexpr int NPX = HH * HWD;
  static constexpr int NPIECE = ((NPX * 4 + 63) / 64 + 3) / 4 * 4;      // wave-pieces (64 lanes x 16 B of fp32) per chunk, a multiple of the 4 waves
  static constexpr int NPW = NPIECE / 4;
  static constexpr int STAGE = NPIECE * 16 * 64;
  static constexpr int EROW = 144;                            // transposition row: 32 fp32 + 16 B
  static constexpr int SCR = 4 * 32 * EROW;                   // one 32 px x 32 ch tile per wave
  static constexpr bool SCR_IN_STAGE = STAGE >= SCR;
  static constexpr int TCAP = 32;
  static constexpr int DESC = 2 * STAGE + (SCR_IN_STAGE ? 0 : SCR);
  static constexpr int LDS = DESC + TCAP * 48;
  static constexpr int NTB = WN * NIW;                        // 32-channel tiles per workgroup
  static constexpr int OCC = (MI * NIW * 16 > 64 || LDS > 80 * 1024) ? 1 : 2;
  // steps the A fragments are requested ahead of their MFMAs (L2 latency: at least 24 MFMAs of this wave); the ring has AD + 1 slots and
  // 18 % (AD + 1) == 0, so that the slot of a step is a compile-time constant in the two unrolled chunk bodies
  // staging register sets: 2 = the pieces requested at the end of chunk q are written during chunk q + 2 (a whole chunk + two steps of
  // latency allowance); 1 = during chunk q + 1 (two steps), where 256 registers do not hold a second set
  static constexpr int NSET = ((OCC == 2 && MI * NIW * 16 + 2 * NPW * 4 > 120) || NPW > 9) ? 1 : 2;
  static constexpr int MPS = 3 * NIW * MI;                    // MFMAs per step
  static constexpr int AD = MPS >= 12 ? 2 : MPS >= 6 ? 5 : 8;
};

template <int S, int WM, int WN, int MI, int NIW>
__global__ void __launch_bounds__(256, (K3Geo<S, WM, WN, MI, NIW>::OCC)) conv_k3d_kernel(const ConvArgs a) {
  using G = K3Geo<S, WM, WN, MI, NIW>;
  constexpr int NPW = G::NPW, RP = G::RP, HWD = G::HWD, NPX = G::NPX;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int DBG = FFP_K3D_DBG ? a.dbg : 0;             // phase-skip bits: 1 epilogue, 2 MFMA, 4 piece requests, 8 split + LDS writes, 16 A requests, 32 B reads, 64 barriers
  const int wm = wave % WM, wn = wave / WM;
  const int p = lane & 31, hh = lane >> 5;

  // ---- this workgroup's items: logical ids first + j * Ws, j < J (one channel block per workgroup) ---------------------------------
  const int n_items = (a.n_tiles_dev ? __builtin_amdgcn_readfirstlane(*a.n_tiles_dev) : a.ntiles_host) * a.n_nblk;
  const int Ws = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int per_xcd = (n_items + 7) >> 3;
  per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  const int first = xcd * per_xcd + slot;
  const int last = min(n_items, (xcd + 1) * per_xcd);
  const int J = first < last ? min((last - first + Ws - 1) / Ws, G::TCAP) : 0;
  if (J == 0) return;
  const int nb0 = first % a.n_nblk;

  int4* desc = reinterpret_cast<int4*>(smem + G::DESC);
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
  const int NC = a.cin >> 4;                           // 16-channel chunks (the launcher requires cin % 16 == 0)
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wpk);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wb), 0, 0x7FFFFFF0, 0x00020000);
  auto rs_in = rs_w;                                   // rebuilt per item (setup_pf)
  auto bload = [](decltype(rs_w) rs, unsigned off, int soff = 0) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };

  // activation scale of the split: per buffer, or per image (= per item: an item is one tile of one image). The prefetch cursor's
  // scale travels with the staging set its pieces were requested into (set_ts); the epilogue looks its item's inverse scale up again.
  float pf_ts = 1.f;
  auto item_scales = [&](int img, float* ts, float* ti) {
    *ts = 1.f; *ti = 1.f;
    if (a.amax_in) split_scales(amax_in_bits_s(a, img), ts, ti);                    // scalar cache: nothing waits for it before its first use
  };

  // ---- LDS image: record index of halo pixel (hy, hx) and the XOR term of its slot index ------------------------------------------------
  auto rec_of = [](int hy, int hx) { return S == 1 ? hy * RP + hx : hy * RP + (hx & 1) * G::PP + (hx >> 1); };
  auto swz_of = [](int hy, int hx) { return S == 1 ? (((hx >> 1) & 2) ^ (hy & 1)) : ((((hx >> 1) >> 1) & 2) ^ ((hy >> 1) & 1)); };

  // ---- staging: a chunk = NPIECE wave-pieces; wave w stages pieces w, w + 4, ...: lane -> (halo pixel, 4-channel quarter) ----------------
  unsigned isrc[NPW];      // byte offset of the lane's vector of the item being PREFETCHED from its image base (chunk 0); OOB: zeros
  unsigned idst[NPW];      // LDS byte offset of the lane's hi half inside a stage (item independent); the lo half sits at idst ^ 32
  unsigned hyx[NPW];       // the piece's halo pixel and quarter (quarter << 16 | row << 8 | column), 0xFFFF: padding pixel
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int idx = (wave + 4 * i) * 64 + lane;
    const int px = idx >> 2, q = idx & 3;
    const int hy = px / HWD, hx = px - hy * HWD;
    const int rec = px < NPX ? rec_of(hy, hx) : px;                    // padding pixels: records of their own past the image
    const int x = px < NPX ? swz_of(hy, hx) : 0;
    idst[i] = (unsigned)(rec * 64 + (((q >> 1) ^ x) << 4) + ((q & 1) << 3));
    hyx[i] = px < NPX ? (unsigned)((q << 16) | (hy << 8) | hx) : 0xFFFFu;      // item independent: an item's set-up is adds and compares, no division
  }
  int pf_item = 0, pf_c = 0;                           // prefetch cursor: two chunks ahead of the MFMAs
  auto setup_pf = [&](int j) {
    const int4 t = desc[j * 3], it = desc[j * 3 + 1];
    const int oy0 = sgpr(t.y), ox0 = sgpr(t.z), Hi = sgpr(it.y), Wi = sgpr(it.z);
    float ti_unused;
    item_scales(sgpr(t.x), &pf_ts, &ti_unused);
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + ((long long)sgpr(it.x) * a.in_cs + a.in_coff) * 4;
    rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(inb), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int hy = (int)(hyx[i] >> 8) & 0xFF, hx = (int)hyx[i] & 0xFF, q = (int)(hyx[i] >> 16);
      const int iy = oy0 * S - 1 + hy, ix = ox0 * S - 1 + hx;
      const bool ok = (hyx[i] & 0xFFFFu) != 0xFFFFu && (unsigned)iy < (unsigned)Hi && (unsigned)ix < (unsigned)Wi;
      isrc[i] = ok ? (unsigned)((iy * Wi + ix) * a.in_cs * 4 + q * 16) : OOB;
    }
  };
  constexpr int NSET = G::NSET;
  uint4 set[NSET][NPW];
  float set_ts[NSET];                                  // split scale of the item each set's pieces belong to
  // `after`: a value the request must wait for in program order — the split results of the piece whose registers it refills. Without it
  // hipcc hoists the load above the split and gives it registers of its own: two staging sets live instead of one (36 VGPRs at stride 2).
  auto piece_fetch = [&](int k, int i, unsigned after = 0u) {
    unsigned off = isrc[i] != OOB ? isrc[i] + (unsigned)(pf_c * 64) : OOB;
    asm("" : "+v"(off) : "v"(after));
    if (!(DBG & 4)) set[k][i] = bload(rs_in, off);
  };
  auto piece_stash = [&](unsigned char* sb, int k, int i) {
    uint2 hi, lo;
    split_pair(__uint_as_float(set[k][i].x), __uint_as_float(set[k][i].y), set_ts[k], hi.x, lo.x);
    split_pair(__uint_as_float(set[k][i].z), __uint_as_float(set[k][i].w), set_ts[k], hi.y, lo.y);
    unsigned d = idst[i];
    asm volatile("" : "+v"(d));                        // keeps d ^ 32 inside the loop: hoisted, the lo addresses cost a register per piece
    *reinterpret_cast<uint2*>(sb + idst[i]) = hi;
    *reinterpret_cast<uint2*>(sb + (d ^ 32u)) = lo;
    return lo.x ^ lo.y;
  };
  auto advance_pf = [&]() {                            // after the last piece of a chunk has been requested
    if (pf_item >= J) return;
    if (++pf_c == NC) {
      pf_c = 0;
      if (++pf_item < J) setup_pf(pf_item);
      else rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wb), 0, 0, 0x00020000);     // stream exhausted: zeros, no traffic
    }
  };

  // ---- A operands: this wave's NIW tiles, straight from the packed fragments [tile][tap][k-group][hi 1 KiB | lo 1 KiB] ---------------------
  int wbase[NIW];
#pragma unroll
  for (int ni = 0; ni < NIW; ++ni) wbase[ni] = min(nb0 * G::NTB + wn * NIW + ni, a.ntiles32 - 1) * 9 * a.ncg * 2048;
  const unsigned wlane = (unsigned)lane * 16u;
  constexpr int AD = G::AD, AR = AD + 1;            // ring depth
  static_assert(18 % AR == 0 && AD < 9, "A ring");
  uint4 aq[AR][2 * NIW];
  auto ldA = [&](int tap, int cg, int r) {
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni) {
      const int so = __builtin_amdgcn_readfirstlane(wbase[ni] + (tap * a.ncg + cg) * 2048);      // wave-uniform: an SGPR operand of the load
      if (DBG & 16) continue;
      aq[r][ni] = bload(rs_w, wlane, so);
      aq[r][NIW + ni] = bload(rs_w, wlane, so + 1024);
    }
  };

  // ---- B operands: fragment f = wm * MI + mi covers tile rows 2f, 2f + 1; lane (p, hh) reads pixel (row p >> 4, column p & 15) ----------------
  // boff[kx][cls]: the address for mi = 0 and the first ky of parity class cls; (mi, ky) add whole rows (immediates)
  unsigned boff[3][2];
  {
    const int r0 = p >> 4, c0 = p & 15;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int cls = 0; cls < 2; ++cls) {
        const int hy = S == 1 ? 2 * wm * MI + r0 + cls : 2 * (2 * wm * MI + r0) + 2 * cls;
        const int hx = c0 * S + kx;
        boff[kx][cls] = (unsigned)(rec_of(hy, hx) * 64 + ((hh ^ swz_of(hy, hx)) << 4));
      }
  }

  f32x16 acc[NIW][MI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
  };
  zero_acc();

  float amax_run = 0.f;

  // One step = one tap (ky, kx) in the generic kernel's order: NIW x MI x 3 MFMAs. The B fragments of step s + 1 are requested from
  // LDS and the A fragments of step s + 2 from L2 ahead of the MFMAs of step s; the staging of the stream rides in the same
  // instruction stream: in step s the pieces [s * NPW / 9, (s + 1) * NPW / 9) of the NEXT chunk (requested one chunk ago) are split,
  // written to the other LDS stage and their registers re-requested for the chunk after it.
  auto chunk = [&](auto phase_tag, const unsigned char* sb, unsigned char* sbn, int cg, int cg_next) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase_tag)::value * 9;           // ring slot of step s: (PH + s) % AR
    constexpr int KS = NSET == 2 ? decltype(phase_tag)::value : 0;   // staging set of this chunk
    uint4 bq[2][2 * MI];
    unsigned tok[NPW];
    auto ldB = [&](int s, int r) {
      const int ky = s / 3, kx = s - 3 * ky;
      const int cls = S == 1 ? (ky & 1) : (ky >> 1);
      const int kyo = S == 1 ? ky - cls : ky - 2 * cls;
      unsigned ol = boff[kx][cls];
      asm volatile("" : "+v"(ol));                               // the lo address is one XOR per step instead of six more address registers
      ol ^= 32u;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if (DBG & 32) continue;
        const unsigned ro = (unsigned)(((S == 1 ? 2 * mi : 4 * mi) + kyo) * RP * 64);      // whole rows: a multiple of 64, commutes with the XOR
        bq[r][mi] = *reinterpret_cast<const uint4*>(sb + boff[kx][cls] + ro);
        bq[r][MI + mi] = *reinterpret_cast<const uint4*>(sb + ol + ro);
      }
    };
    ldB(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      if (s + 1 < 9) ldB(s + 1, (s + 1) & 1);
      if (s + AD < 9) ldA(s + AD, cg, (PH + s + AD) % AR);
      else ldA(s + AD - 9, cg_next, (PH + s + AD) % AR);
      // staging: the pieces of the next chunk are split and written in steps 2..8; ALL of them are re-requested in one burst at the end
      // of step 8. vmcnt retires in order: a wait for A fragments also waits for every older request, so a piece request (HBM latency)
      // issued between two A requests stalls the A wait two steps later — once per piece with one request per step (measured: 0.35 of the
      // MFMA rate whatever the tile size), once per CHUNK with the burst.
      if (s >= 2) {
#pragma unroll
        for (int i = (s - 2) * NPW / 7; i < (s - 1) * NPW / 7; ++i) tok[i] = (DBG & 8) ? 0u : piece_stash(sbn, KS, i);
      }
      if (s == 8) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) piece_fetch(KS, i, tok[i]);
        set_ts[KS] = pf_ts;                                        // the cursor's item (advance_pf comes after the burst)
      }
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          union { uint4 u; f16x8 h; } ah, al, bh, bl;
          if (DBG & 2) continue;
          ah.u = aq[(PH + s) % AR][ni]; al.u = aq[(PH + s) % AR][NIW + ni]; bh.u = bq[s & 1][mi]; bl.u = bq[s & 1][MI + mi];
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh.h, acc[ni][mi], 0, 0, 0);     // small terms first
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl.h, acc[ni][mi], 0, 0, 0);
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh.h, acc[ni][mi], 0, 0, 0);
        }
      // one MFMA, then up to three of the other instructions: LDS / memory / split work sits in the MFMAs' issue shadow
#pragma unroll
      for (int r = 0; r < 3 * NIW * MI; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x7F6, 3, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    advance_pf();
  };

  // ---- epilogue of item j: each 32 px x 32 ch accumulator tile goes through a wave-private LDS tile so that stores and residual
  // loads are whole 128-byte pixel rows (8 consecutive lanes x 16 B) -------------------------------------------------------------------
  auto epilogue = [&](int j, unsigned char* scr) __attribute__((always_inline)) {
    const int4 t = desc[j * 3], ot = desc[j * 3 + 2];
    const int oy0 = sgpr(t.y), ox0 = sgpr(t.z), Ho = sgpr(ot.y), Wo = sgpr(ot.z);
    const int img = sgpr(t.x);
    float ts_unused, tinv;
    item_scales(img, &ts_unused, &tinv);
    float am = 0.f;                                    // largest |value| of this item's outputs
    unsigned slot_now = 0u;                            // what the item's slot holds (read here, compared after the stores)
    if (a.amax_img && a.amax_out) slot_now = slot_peek(a.amax_out + img);
    const long long out_base = sgpr(ot.x);
    const unsigned char* r1b = a.res1 ? reinterpret_cast<const unsigned char*>(a.res1) + (out_base * a.r1_cs + a.r1_coff) * 4 : wb;
    const unsigned char* r2b = a.res2 ? reinterpret_cast<const unsigned char*>(a.res2) + (out_base * a.r2_cs + a.r2_coff) * 4 : wb;
    unsigned char* ob = reinterpret_cast<unsigned char*>(a.out) + (out_base * a.out_cs + a.out_coff) * 4;
    const auto rs_r1 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r1b), 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_r2 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(r2b), 0, 0x7FFFFFF0, 0x00020000);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ob), 0, 0x7FFFFFF0, 0x00020000);
    unsigned char* et = scr + wave * (32 * G::EROW);
    const int ch0 = (lane & 7) * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int f = wm * MI + mi;
#pragma unroll
      for (int ni = 0; ni < NIW; ++ni) {
        const int nt = nb0 * G::NTB + wn * NIW + ni;
        if (nt >= a.ntiles32) continue;                          // wave-uniform
        // bias and output scale of this lane's 4 channels of the tile (L2-resident: fetched per tile instead of held across the main loop)
        const int chb = nt * 32 + ch0;
        const float4 bv = *reinterpret_cast<const float4*>(a.bias + chb);
        const float4 sv = a.oscale ? *reinterpret_cast<const float4*>(a.oscale + chb) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float bias_t[4] = {bv.x, bv.y, bv.z, bv.w};
        const float osc_t[4] = {sv.x * tinv, sv.y * tinv, sv.z * tinv, sv.w * tinv};
        unsigned rel[4];
        bool oks[4];
        uint4 r1v[4], r2v[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pp = it * 8 + (lane >> 3);
          const int oy = oy0 + 2 * f + (pp >> 4), ox = ox0 + (pp & 15);
          oks[it] = oy < Ho && ox < Wo && nt * 32 + ch0 < a.cout;
          rel[it] = (unsigned)(oy * Wo + ox);
          r1v[it] = make_uint4(0u, 0u, 0u, 0u); r2v[it] = r1v[it];
          if (a.res1) r1v[it] = bload(rs_r1, oks[it] ? (rel[it] * a.r1_cs + nt * 32 + ch0) * 4 : OOB);
          if (a.res2) r2v[it] = bload(rs_r2, oks[it] ? (rel[it] * a.r2_cs + nt * 32 + ch0) * 4 : OOB);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous tile's staging reads have landed in registers
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(et + p * G::EROW + (8 * g + 4 * hh) * 4) =
              make_float4(acc[ni][mi][4 * g + 0], acc[ni][mi][4 * g + 1], acc[ni][mi][4 * g + 2], acc[ni][mi][4 * g + 3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // LDS is in order per wave: the tile is visible to all its lanes
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int pp = it * 8 + (lane >> 3);
          const float4 t4 = *reinterpret_cast<const float4*>(et + pp * G::EROW + ch0 * 4);
          float v[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = apply_act(fmaf(v[q], osc_t[q], bias_t[q]), a.act);
          if (a.res1) {
            const float* r = reinterpret_cast<const float*>(&r1v[it]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] * a.s1 + r[q];
          }
          if (a.res2) {
            const float* r = reinterpret_cast<const float*>(&r2v[it]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] * a.s2 + r[q];
          }
          if (oks[it]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) am = fmaxf(am, fabsf(v[q]));
          }
          u32x4 ov = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(ov, rs_o, oks[it] ? (rel[it] * a.out_cs + nt * 32 + ch0) * 4 : OOB, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (a.amax_img) { if (a.amax_out) raise_amax(a.amax_out + img, am, slot_now); }
    else amax_run = fmaxf(amax_run, am);
  };

  // ---- the chunk stream ----------------------------------------------------------------------------------------------------------------
  const int Q = J * NC;
  int cj = 0, cc = 0;
  setup_pf(0);
#pragma unroll
  for (int i = 0; i < NPW; ++i) piece_fetch(0, i);
  set_ts[0] = pf_ts;
  advance_pf();
#pragma unroll
  for (int t = 0; t < AD; ++t) ldA(t, 0, t);
#pragma unroll
  for (int i = 0; i < NPW; ++i) piece_fetch(0, i, piece_stash(smem, 0, i));
  set_ts[0] = pf_ts;
  advance_pf();
  if constexpr (NSET == 2) {
#pragma unroll
    for (int i = 0; i < NPW; ++i) piece_fetch(1, i);
    set_ts[1] = pf_ts;
    advance_pf();
  }
  __syncthreads();
  // two chunks per iteration so that the stage bases are compile-time offsets of the LDS instructions (no address arithmetic, no address registers)
  auto step_chunk = [&](auto phase_tag, unsigned char* sb, unsigned char* sbn) __attribute__((always_inline)) {
    const int cn = cc + 1 == NC ? 0 : cc + 1;
    chunk(phase_tag, sb, sbn, cc, cn);
    if (++cc == NC) {
      if constexpr (G::SCR_IN_STAGE) __syncthreads();            // every wave is done reading the stage that now serves as scratch
      if (!(DBG & 1)) epilogue(cj, G::SCR_IN_STAGE ? sb : smem + 2 * G::STAGE);
      zero_acc();
      cc = 0; ++cj;
    }
    if (!(DBG & 64)) __syncthreads();
  };
  for (int q = 0; q < Q; q += 2) {
    step_chunk(std::integral_constant<int, 0>{}, smem, smem + G::STAGE);
    if (q + 1 >= Q) break;
    step_chunk(std::integral_constant<int, 1>{}, smem + G::STAGE, smem);
  }
  if (a.amax_out && !a.amax_img) raise_amax(a.amax_out, amax_run);
}

template <int S, int WM, int WN, int MI, int NIW> struct K3Cfg {
  using G = K3Geo<S, WM, WN, MI, NIW>;
  static_assert(G::LDS <= 160 * 1024, "conv_k3d: LDS");
  static void init() {
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_k3d_kernel<S, WM, WN, MI, NIW>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS));
  }
  static void launch(ConvArgs& a, Level* out_lvl, hipStream_t st) {
    int n_tiles = 0;
    a.tiles = out_lvl->tile_table(G::TH, &n_tiles, &a.n_tiles_dev, st);
    if (n_tiles == 0) return;
    a.ntiles_host = n_tiles;
    a.n_nblk = (a.ntiles32 + G::NTB - 1) / G::NTB;
    const long long items = (long long)n_tiles * a.n_nblk;
    long long per_xcd = (items + 7) / 8;
    per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
    static const int ws_env = [] { const char* e = getenv("FFP_K3D_WS"); return e ? atoi(e) : 0; }();      // experiment: workgroups per XCD (32 = one per CU: half of every CU's LDS stays free)
    long long ws = std::max<long long>(ws_env > 0 ? ws_env : 32 * G::OCC, (per_xcd + G::TCAP - 1) / G::TCAP);
    ws = (ws + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
    FFP_CHECK(8 * ws < (1ll << 31) && (per_xcd + ws - 1) / ws <= G::TCAP, FFP_ERR_STATE, "conv_k3d: launch geometry");
    hipLaunchKernelGGL((conv_k3d_kernel<S, WM, WN, MI, NIW>), dim3((unsigned)(8 * ws)), dim3(256), G::LDS, st, a);
  }
};

// force_shape 17..21, per stride: {128 px x 128 ch, 128 px x 64 ch, 128 px x 32 ch, 256 px x 128 ch (one workgroup per CU, 512 registers), 256 px x 64 ch}
template <int S> struct K3Family {
  using V0 = K3Cfg<S, 2, 2, 2, 2>;
  using V1 = K3Cfg<S, 2, 2, 2, 1>;
  using V2 = K3Cfg<S, 4, 1, 1, 1>;
  using V3 = K3Cfg<S, 2, 2, 4, 2>;
  using V4 = K3Cfg<S, 2, 2, 4, 1>;
  static void init() { V0::init(); V1::init(); V2::init(); V3::init(); V4::init(); }
  static void launch(ConvArgs& a, int shape, Level* out_lvl, hipStream_t st) {
    switch (shape) {
      case 17: V0::launch(a, out_lvl, st); break;
      case 18: V1::launch(a, out_lvl, st); break;
      case 19: V2::launch(a, out_lvl, st); break;
      case 20: V3::launch(a, out_lvl, st); break;
      default: V4::launch(a, out_lvl, st); break;
    }
  }
};

}  // namespace

void conv_k3d_init() { K3Family<1>::init(); K3Family<2>::init(); }

bool conv_k3d_enabled() {
  static const bool on = [] { const char* e = getenv("FFP_K3D"); return !(e && e[0] == '0'); }();
  return on;
}

// what the kernel does not do: fp16 / exact-fp32 arithmetic, upsampled inputs, unaligned outputs, input channel counts that are not
// whole 16-channel groups, images whose byte size does not fit the 32-bit offsets of one buffer resource
unsigned conv_k3d_mask(const ConvOp& op, const ConvArgs& a) {
  const PackedConv& pc = *op.pc;
  if (!(pc.split && pc.dt == F32 && pc.k == 3 && (op.stride == 1 || op.stride == 2) && !op.up && !op.has_up2 && a.fast_out && a.out_f32 &&
        pc.cin % 16 == 0 && pc.cin_pad == pc.cin && op.in.lvl->n > 0))
    return 0;
  long long max_px = 0;
  for (int i = 0; i < op.in.lvl->n; ++i) max_px = std::max(max_px, (long long)op.in.lvl->h[i] * op.in.lvl->w[i]);
  long long max_opx = 0;
  for (int i = 0; i < op.out.lvl->n; ++i) max_opx = std::max(max_opx, (long long)op.out.lvl->h[i] * op.out.lvl->w[i]);
  const long long cs_max = std::max<long long>(op.out.cs, std::max(op.has_res1 ? op.res1.cs : 0, op.has_res2 ? op.res2.cs : 0));
  if (max_px * op.in.cs * 4 >= 0x7FFFFFF0ll || max_opx * cs_max * 4 >= 0x7FFFFFF0ll) return 0;
  unsigned m = 1u << 19;
  if (a.ntiles32 >= 2) m |= (1u << 18) | (1u << 21);
  if (a.ntiles32 >= 3) m |= (1u << 17) | (1u << 20);
  return m;
}

void launch_conv_k3d(ConvArgs& a, int shape, int stride, Level* out_lvl, hipStream_t st) {
  if (stride == 1) K3Family<1>::launch(a, shape, out_lvl, st);
  else K3Family<2>::launch(a, shape, out_lvl, st);
}

}  // namespace ffp
