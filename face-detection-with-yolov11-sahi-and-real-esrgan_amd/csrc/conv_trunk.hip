// conv_trunk.hip — the Real-ESRGAN body (every 3x3 fp16 conv of the 69 residual dense blocks: 345 layers) as ONE persistent launch.
//
// What round 4 measured first (tools/probes/stage_probe.hip, profiles/r04_staging_rate_probe.txt): the global -> LDS staging rate of a CU
// scales with the NUMBER OF WAVES that issue loads (~4 B/clk per wave) up to ~32 B/clk per CU on L2 hits and ~23 B/clk out of the Infinity
// Cache, whatever the prefetch depth, access pattern or load form. conv_rows16_kernel stages 39 KiB per 288 MFMAs = 34 B/clk per CU at
// the full matrix rate: it is STAGING-BOUND (its phase probe: staging alone 26 us, MFMAs alone 23.6 us, together 43.8 us), and a role
// split with four producer waves (conv_rows16pc) delivers 12-14 B/clk. So this kernel
//   * stages fewer bytes per MFMA: one 8-wave workgroup per CU owns a 32 x 16 pixel tile (wave w: rows 4w..4w+3), so a chunk's weight
//     fragments serve twice the pixels (25 B/clk per CU at the full matrix rate for 32 output channels), and the 64-channel layers (conv5 of
//     every dense block: 46 % of the body's FLOPs) run BOTH 32-channel blocks over one staged halo tile (16.5 B/clk);
//   * stages with LDS-DMA (buffer_load_dwordx4 ... lds) issued by ALL EIGHT waves: no staging VGPRs, no ds_write_b128 traffic on the
//     VGPR -> LDS path, the LDS image is written lane-linear and the bank-conflict-free XOR swizzle is applied to the SOURCE channel slot;
//     out-of-image lanes use an out-of-range buffer offset (the range check returns zeros: the conv's zero padding);
//   * walks (layer, tile) items of ALL the layers it is given out of one global queue (an atomic counter, layer-major order): no launch
//     boundary, no per-layer tail, no per-launch skeleton. Layer l + 1 of a tile needs layer l of the tile and of its eight neighbours in the
//     same image: `done[tile]` = layers completed (written by one lane after every wave of the workgroup has drained its stores),
//     polled ahead of time by wave 0 — one small control step per chunk: item ids are fetched three items ahead, their tile entries one
//     step later, the nine counters of every fetched item every step — so that a dependency never stalls the MFMA stream unless it is
//     really late (small batches), and then only this workgroup. Items are taken in queue order and depend only on EARLIER items, so
//     the grid cannot deadlock whatever part of it is resident (a second persistent launch on the device, the detector's kernels).
//     Activations written inside the launch are stored write-through (sc1) and read with sc1 loads (DMA and residuals): per-CU L1s are
//     never refreshed and per-XCD L2s are not coherent (MI355X guide, inter-workgroup visibility); weights and tables are read-only.
// Arithmetic is conv_rows16_kernel's instruction for instruction (v_mfma_f32_16x16x32_f16, accumulators start from the bias, chunks
// ascending, taps kx-major inside a chunk, register epilogue): results are BIT-IDENTICAL, which is the kernel's parity oracle
// (tests/test_gpu_trunk.py) on top of the usual one.
// LDS: two stages of [40 pieces of halo pixels | 36 pieces of weights] = 2 x 77,824 B + descriptors: one workgroup per CU.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_args.hpp"
#include "trunk.hpp"

#ifndef FFP_TRUNK_DBG
#define FFP_TRUNK_DBG 0        // 1: diagnostic build — s_memtime stamps per phase of every iteration, sums of workgroup 0 behind the queue words (FFP_TRUNK_DUMP=1 prints them)
#endif
#ifndef FFP_TRUNK_SKIP
#define FFP_TRUNK_SKIP 0       // diagnostic builds: COMPILE-TIME phase-skip bits (what is left keeps the production schedule): 1 epilogue, 2 MFMA, 4 DMA, 16 fragment reads
#endif

namespace ffp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct TG {
  static constexpr int TH = 32;                               // tile rows (tile columns: 16)
  static constexpr int HR = TH + 2, HC = 18;                  // halo rows / columns
  static constexpr int PIX_PIECES = 39;                       // 34 x 18 = 612 halo pixels x 64 B = 38.25 KiB -> 39 pieces of 1 KiB (12 dummy pixels)
  static constexpr int PIX = 40 * 1024;                       // pixel area of a stage (one spare piece keeps the weight area 1 KiB aligned)
  static constexpr int WB = 18 * 1024;                        // weight fragments of one 32-channel block and chunk: [tap][M-tile][lane] x 16 B
  static constexpr int STAGE = PIX + 2 * WB;                  // 77,824
  static constexpr int MISC = 2 * STAGE;                      // descriptor ring + control words
  static constexpr int LDS = MISC + 4 * 64 + 64;              // 155,968 B
  static constexpr int ROWB = HC * 64;                        // bytes of a halo row
};

enum : int { D_EMPTY = 0, D_KNOWN = 1, D_READY = 2, D_END = 3 };
// descriptor slot (16 ints in LDS): 0 state, 1 queue id, 2 layer, 3 tile index, 4 first pixel of the image, 5 y0 | x0 << 16, 6 h | w << 16,
// 7 tile columns | tile rows << 16 of the image (4..7 = the packed tile entry, Level::tile_table_packed)

__device__ __forceinline__ unsigned rfl(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// a raw buffer descriptor built by hand (the DMA instruction is issued from inline asm and takes it as four SGPRs)
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(base);
  u32x4 r;
  r[0] = rfl((unsigned)u);
  r[1] = rfl((unsigned)(u >> 32)) & 0xFFFFu;                  // stride 0: raw buffer
  r[2] = bytes;
  r[3] = 0x00020000u;
  return r;
}

// LDS[lds + lane * 16 .. + 16) <- buffer[voff + soff .. + 16) per lane, straight into LDS (no VGPR): completion is counted on vmcnt BY HAND
// (hipcc does not see the instruction; a visible LDS-DMA would make it drain vmcnt before every LDS read). voff = 0xFFFFFFFF: out of
// range, the lane's 16 bytes are zeros. M0 carries the LDS address and is restored (compiler-reserved).
template <bool SC1> __device__ __forceinline__ void dma16(u32x4 rs, unsigned voff, unsigned soff, unsigned lds) {
  unsigned keep;
  if constexpr (SC1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
}

struct Item {            // wave-uniform description of a work item (SGPRs)
  int layer, tile, y0, x0;
  int NC, NT;            // 32-channel input chunks, 32-channel output blocks (1 or 2)
  int H, W;              // image size
  int px0;               // first pixel of the image in the level
};

template <bool COH>      // COH: several layers in one launch — sc1 loads / stores of activations and the done[] protocol
__global__ void __launch_bounds__(512, 2) conv_trunk_kernel(const TrunkArgs a) {
  using G = TG;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  int* const desc = reinterpret_cast<int*>(smem + G::MISC);   // 4 slots x 16 ints
  int* const snap = desc + 64;                                  // [iteration parity]: is the item after the current one READY? (see the main loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl(tid >> 6);
  const int pc = lane & 15, g = lane >> 4;
  typedef const __attribute__((address_space(4))) TrunkLayer* LPtr;
  const LPtr layers = reinterpret_cast<LPtr>(reinterpret_cast<unsigned long long>(a.layers));

  const int n_tiles = a.n_tiles_dev ? sload(a.n_tiles_dev, 0) : a.ntiles_host;
  const unsigned total = (unsigned)n_tiles * (unsigned)a.n_layers;
  if (total == 0) return;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr int AUXC = COH ? 16 : 0;                          // sc1 on activation loads / stores of a multi-layer launch

  unsigned isrc[5];        // byte offset of the lane's 16 bytes of chunk 0 from the image's first pixel record (+ in_coff), or OOB

  // fragment read offsets: halo rows 4 * wave + j (j = 0..5), column pc + kx, slot g
  unsigned boff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = pc + kx;
    boff[kx] = (unsigned)(((4 * wave) * G::HC + hx) * 64 + ((g ^ ((hx >> 1) & 2)) << 4));
  }
  const unsigned aoff = G::PIX + lane * 16;

  // ---- control (wave 0): the queue, the tile entries and the dependency counters of the next three items, one step per chunk -------------
  int ck = -1;             // sequence number (within this workgroup) of the CURRENT item (-1: none yet); slot = seq & 3
  int fk = 0, fstate = 0;  // fetcher: next sequence number to fetch; 0 idle, 1 queue id asked, 2 id known, 3 tile entry asked, 4 queue exhausted
  int f_id = 0;            // pending atomic result (lane 0)
  u32x4 f_tile = {0u, 0u, 0u, 0u};
  int f_layer = 0, f_t = 0;
  unsigned f_qid = 0;
  // wave 0's own copy of the four descriptor slots (scalar registers; LDS holds what the other waves read): state, layer, tile index,
  // tile position and grid packed as tx | ty << 8 | nx << 16 | ny << 24 — a control step reads no LDS
  int s_st[4] = {0, 0, 0, 0}, s_ly[4] = {0, 0, 0, 0}, s_tt[4] = {0, 0, 0, 0}, s_geo[4] = {0, 0, 0, 0};
  auto get4 = [](const int (&v)[4], int k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3]; };
  auto put4 = [](int (&v)[4], int k, int x) { v[0] = k == 0 ? x : v[0]; v[1] = k == 1 ? x : v[1]; v[2] = k == 2 ? x : v[2]; v[3] = k == 3 ? x : v[3]; };
  int p_base = -1;         // polls in flight: sequence number of lane group 0 (-1: none)
  unsigned p_val = 0;
  int p_need = 0;          // per lane: the layer its polled counter must have reached
  bool p_act = false;      // per lane: it polled a counter
  const auto rs_sync = __builtin_amdgcn_make_buffer_rsrc(a.queue, 0, 64 + 4 * n_tiles, 0x00020000);          // [queue | done[]]
  const auto rs_tiles = __builtin_amdgcn_make_buffer_rsrc(const_cast<int4*>(a.tiles), 0, 16 * n_tiles, 0x00020000);
  // "nothing to ask" offset of the control step's dword / atomic operations: aligned and beyond any num_records (0xFFFFFFFF + 4 wraps in
  // a 32-bit range check; a buffer ATOMIC at that offset faulted with a memory aperture violation on gfx950, 16-byte loads do not)
  constexpr unsigned OOBA = 0x80000000u;
  // A control step has two halves. control_issue() — at the top of an iteration — asks: the queue for the next id (at most three items ahead),
  // or the tile table for the id it got, and (multi-layer launches) the dependency counters of every fetched, not yet ready item.
  // control_consume() — at the END of the iteration, right after the wave's full vmcnt(0) wait — reads the answers. Every request is a buffer
  // operation issued unconditionally inside the step (an out-of-range offset where there is nothing to ask): behind exec-masked loads, or for
  // results carried over the loop's back edge, hipcc waits for vmcnt(0) at the point of use, and that wait would also cover the epilogue's
  // stores (1-2 us: measured 1,500 cycles per iteration with the one-step form). A last-chunk iteration (stores still in flight at its end)
  // skips the consume; nothing new is asked until the answers have been read.
  unsigned f_tile_off = OOBA;
  bool ctl_pending = false;
  auto control_issue = [&]() {
    if (ctl_pending) return;
    const bool want_id = fstate == 0 && fk <= ck + 3;
    const bool want_tile = fstate == 2;
    const bool want_poll = COH && (s_st[0] == D_KNOWN || s_st[1] == D_KNOWN || s_st[2] == D_KNOWN || s_st[3] == D_KNOWN);
    if (!want_id && !want_tile && !want_poll) return;          // the steady state of a large batch: three items fetched and ready
    f_id = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rs_sync, (want_id && lane == 0) ? 0u : OOBA, 0, 0);
    f_tile = __builtin_amdgcn_raw_buffer_load_b128(rs_tiles, want_tile ? f_tile_off : OOBA, 0, 0);
    if (want_id) fstate = 1;
    if (want_tile) fstate = 3;
    if (COH) {              // tile t and its neighbours t + dy * nx + dx inside the image's tile grid: lane group gi asks for sequence ck + 1 + gi
      const int gi = lane >> 4, j = lane & 15;
      const int sq = ck + 1 + gi, k = sq & 3;
      const int stt = get4(s_st, k), geo = get4(s_geo, k), t = get4(s_tt, k);
      p_need = get4(s_ly, k);
      p_act = gi < 3 && j < 9 && sq < fk && stt == D_KNOWN;
      const int dy = j / 3 - 1, dx = j - (j / 3) * 3 - 1;
      const int tx = geo & 0xFF, ty = (geo >> 8) & 0xFF, nx = (geo >> 16) & 0xFF, ny = (geo >> 24) & 0xFF;
      const bool in = (unsigned)(tx + dx) < (unsigned)nx && (unsigned)(ty + dy) < (unsigned)ny;
      const unsigned off = p_act ? 64u + 4u * (unsigned)(in ? t + dy * nx + dx : t) : OOBA;
      p_val = __builtin_amdgcn_raw_buffer_load_b32(rs_sync, off, 0, 16);      // sc1: the counters are written by other workgroups
      p_base = __builtin_amdgcn_ballot_w64(p_act) != 0 ? ck + 1 : -1;
    }
    ctl_pending = true;
  };
  auto control_consume = [&]() {
    if (!ctl_pending) return;
    ctl_pending = false;
    if (fstate == 1) {
      const unsigned id = (unsigned)rfl(f_id);
      if (id >= total) {
        put4(s_st, fk & 3, D_END);
        if (lane == 0) desc[(fk & 3) * 16] = D_END;
        fstate = 4;
      } else {
        f_qid = id;
        f_layer = (int)(id / (unsigned)n_tiles);
        f_t = (int)(id - (unsigned)f_layer * (unsigned)n_tiles);
        f_tile_off = (unsigned)f_t * 16u;
        fstate = 2;
      }
    } else if (fstate == 3) {
      const int e0 = rfl((int)f_tile[0]), e1 = rfl((int)f_tile[1]), e2 = rfl((int)f_tile[2]), txy = rfl((int)f_tile[3]);
      const int k = fk & 3;
      const int state = (COH && f_layer > 0) ? D_KNOWN : D_READY;       // the first layer of a launch depends on earlier launches only
      put4(s_st, k, state); put4(s_ly, k, f_layer); put4(s_tt, k, f_t);
      put4(s_geo, k, ((e1 >> 16) >> 4) | (((e1 & 0xFFFF) / G::TH) << 8) | ((txy & 0xFF) << 16) | (((txy >> 16) & 0xFF) << 24));
      if (lane == 0) {
        int* d = desc + k * 16;
        d[1] = (int)f_qid; d[2] = f_layer; d[3] = f_t; d[4] = e0; d[5] = e1; d[6] = e2; d[7] = txy;
        d[0] = state;
      }
      ++fk;
      fstate = 0;
    }
    if (COH && p_base >= 0) {
      const unsigned long long okm = __builtin_amdgcn_ballot_w64(!p_act || (int)p_val >= p_need);
      const unsigned long long actm = __builtin_amdgcn_ballot_w64(p_act);
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const unsigned long long gm = 0x1FFull << (16 * gi);
        if ((actm & gm) != 0 && (okm & gm) == gm) {
          const int k = (p_base + gi) & 3;
          put4(s_st, k, D_READY);
          if (lane == 0) desc[k * 16] = D_READY;
        }
      }
      p_base = -1;
    }
  };
  // blocking form (prologue, late dependencies): until sequence `sq` is READY or the queue has ended. Bounded: a spin that never ends
  // would hang the device — it gives up, raises the error word and ends this workgroup's walk instead
  auto control_wait = [&](int sq) {
    for (int spin = 0;; ++spin) {
      const int st = get4(s_st, sq & 3);
      if (st == D_READY || st == D_END) break;
      if (spin > (1 << 22)) {
        if (lane == 0) { atomicAdd(a.queue + 1, 1u); desc[(sq & 3) * 16] = D_END; }
        put4(s_st, sq & 3, D_END);
        break;
      }
      control_issue();
      control_consume();
      if (spin > 4) __builtin_amdgcn_s_sleep(4);
    }
  };

  if (tid < 80) desc[tid] = 0;
  __syncthreads();

  // ---- item set-up: descriptor slot -> SGPRs, per-lane source offsets of the halo pixels ------------------------------------------------
  auto load_item = [&](int sq) {
    const int4 d0 = *reinterpret_cast<const int4*>(desc + (sq & 3) * 16), d1 = *reinterpret_cast<const int4*>(desc + (sq & 3) * 16 + 4);
    Item it;
    it.layer = rfl(d0.z); it.tile = rfl(d0.w);
    it.px0 = rfl(d1.x);
    const int yx = rfl(d1.y), hw = rfl(d1.z);
    it.y0 = yx & 0xFFFF; it.x0 = (int)((unsigned)yx >> 16);
    it.H = hw & 0xFFFF; it.W = (int)((unsigned)hw >> 16);
    it.NC = layers[it.layer].cin >> 5;
    it.NT = layers[it.layer].cout >> 5;
    return it;
  };
  u32x4 rs_in, rs_w, rs_b;
  auto setup_dma = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    const int cs = L->in_cs;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(L->in) + ((long long)it.px0 * cs + L->in_coff) * 2;
    rs_in = make_rsrc(inb, 0x7FFFFFF0u);
    rs_w = make_rsrc(L->wpk, (unsigned)(it.NT * it.NC * G::WB));
    rs_b = make_rsrc(L->bias, (unsigned)(it.NT * 128));
#pragma unroll
    for (int i = 0; i < 5; ++i) {                    // pixel piece wave + 8 i: lane -> halo pixel q = piece * 16 + lane / 4, LDS slot position lane % 4
      const int p = wave + 8 * i;
      const int q = p * 16 + (lane >> 2);
      const int hy = (q * 3641) >> 16, hx = q - hy * G::HC;          // q / 18 for q < 640
      const int sl = (lane & 3) ^ ((hx >> 1) & 2);                   // which holds SOURCE slot (lane % 4) ^ ((hx >> 1) & 2): conflict-free fragment reads
      const int iy = it.y0 - 1 + hy, ix = it.x0 - 1 + hx;
      const bool ok = p < G::PIX_PIECES && q < G::HR * G::HC && (unsigned)iy < (unsigned)it.H && (unsigned)ix < (unsigned)it.W;
      isrc[i] = ok ? (unsigned)((iy * it.W + ix) * cs * 2 + sl * 16) : OOB;
    }
  };
  // chunk c of the item -> stage at LDS byte address `st`: <= 5 pixel pieces + <= 3 (5) weight pieces per wave, issued as ten STEPS (one per
  // MFMA step of the chunk being multiplied, so that the instructions' issue time — 60-185 cycles each while the address unit is busy —
  // sits in the matrix instructions' shadow instead of in front of them). The chunk's 64 bytes per pixel ride in the scalar offset.
  auto dma_piece = [&](const Item& it, int c, unsigned st, int step) {
    if ((FFP_TRUNK_SKIP & 4)) return;
    if (step < 5) {
      if (wave + 8 * step < G::PIX_PIECES) dma16<COH>(rs_in, isrc[step], (unsigned)(c * 64), st + (unsigned)((wave + 8 * step) << 10));
      else if (c == 0) dma16<false>(rs_b, (unsigned)lane * 16u, 0u, st + (unsigned)(G::PIX_PIECES << 10));   // the layer's bias: piece 39 of chunk 0's stage (wave 7)
    } else {
      const int q = wave + 8 * (step - 5);
      if (q < it.NT * 18) {
        const int nt = q >= 18 ? 1 : 0, pq = q - nt * 18;
        dma16<false>(rs_w, (unsigned)lane * 16u, (unsigned)(((nt * it.NC + c) * 18 + pq) << 10), st + (unsigned)(G::PIX + (q << 10)));
      }
    }
  };
  auto issue_dma = [&](const Item& it, int c, unsigned st) {
#pragma unroll
    for (int step = 0; step < 10; ++step) dma_piece(it, c, st, step);
  };

  // ---- arithmetic: conv_rows16_kernel's chunk, for NM = 2 NT M-tiles -----------------------------------------------------------------------
  f32x4 acc[4][4];
  auto init_acc = [&](const Item& it, const unsigned char* sb) {      // the bias came with chunk 0's stage (piece 39): no global load at an item's start
    const float* b = reinterpret_cast<const float*>(sb + (G::PIX_PIECES << 10));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (nt < it.NT) {
        const float4 b0 = *reinterpret_cast<const float4*>(b + nt * 32 + 8 * g);
        const float4 b1 = *reinterpret_cast<const float4*>(b + nt * 32 + 8 * g + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][nt * 2] = f32x4{b0.x, b0.y, b0.z, b0.w};
          acc[i][nt * 2 + 1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        }
      }
    }
  };
  auto chunk = [&](const unsigned char* sb, auto nm_tag, auto&& between) {
    constexpr int NM = decltype(nm_tag)::value;
    // Input-row fragments in EIGHT rolling slots instead of two sets of six: fragment (kx, row r) lives in slot (r + 6 kx) % 8 and is read
    // from LDS at least one step before its first MFMA, into a slot whose previous row is dead (row r of a kx serves the steps ky = r - 3 .. r):
    //   step 0: (1,0) (1,1) | 1: (1,2) | 2: (1,3) | 3: (1,4) (1,5) | 4: (2,0) (2,1) (2,2) | 5: (2,3) | 6: (2,4) (2,5)
    // 16 registers fewer than the double set; the weight fragments are one step ahead (a step is 8 or 16 MFMAs: 128 / 256 cycles of cover).
    uint4 bq[8] = {}, aq[2][NM] = {};
    auto ldB1 = [&](int kx, int r) {
      if ((FFP_TRUNK_SKIP & 16)) { asm volatile("" : "+v"(bq[(r + 6 * kx) & 7].x), "+v"(bq[(r + 6 * kx) & 7].y), "+v"(bq[(r + 6 * kx) & 7].z), "+v"(bq[(r + 6 * kx) & 7].w)); return; }
      bq[(r + 6 * kx) & 7] = *reinterpret_cast<const uint4*>(sb + boff[kx] + r * G::ROWB);
    };
    auto ldA = [&](int s, int q) {
      const int tap = (s % 3) * 3 + s / 3;
      if ((FFP_TRUNK_SKIP & 16)) {
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) asm volatile("" : "+v"(aq[q][mm].x), "+v"(aq[q][mm].y), "+v"(aq[q][mm].z), "+v"(aq[q][mm].w));
        return;
      }
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) aq[q][mm] = *reinterpret_cast<const uint4*>(sb + aoff + (((mm >> 1) * 18 + tap * 2 + (mm & 1)) << 10));
    };
#pragma unroll
    for (int r = 0; r < 6; ++r) ldB1(0, r);
    ldA(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int kx = s / 3, ky = s - 3 * kx;
      if (s + 1 < 9) ldA(s + 1, (s + 1) & 1);
      if (s == 0) { ldB1(1, 0); ldB1(1, 1); }
      if (s == 1) ldB1(1, 2);
      if (s == 2) ldB1(1, 3);
      if (s == 3) { ldB1(1, 4); ldB1(1, 5); }
      if (s == 4) { ldB1(2, 0); ldB1(2, 1); ldB1(2, 2); }
      if (s == 5) ldB1(2, 3);
      if (s == 6) { ldB1(2, 4); ldB1(2, 5); }
      between(s);
      if (s == 8) between(9);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) {
          if ((FFP_TRUNK_SKIP & 2)) continue;
          union { uint4 u; f16x8 h; } ua, ub;
          ua.u = aq[s & 1][mm]; ub.u = bq[(i + ky + 6 * kx) & 7];
          acc[i][mm] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, acc[i][mm], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 4 * NM; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x7F6, 1, 0);     // one other instruction
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue: lane (pc, g) holds channels 32 nt + 8 g .. + 7 of pixel (row 4 * wave + i, column pc); 4 NT stores per wave -----------------
  struct Epi {             // the epilogue's layer parameters (scalar loads issued BEFORE the last chunk's MFMAs: their latency is off the item's tail)
    const unsigned char *ob, *r1b, *r2b;
    int act, o_cs, r1_cs, r2_cs;
    float s1, s2;
    bool has1, has2;
  };
  auto load_epi = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    Epi e;
    e.act = L->act; e.s1 = L->s1; e.s2 = L->s2;
    e.o_cs = L->out_cs; e.r1_cs = L->r1_cs; e.r2_cs = L->r2_cs;
    e.has1 = L->res1 != nullptr; e.has2 = L->res2 != nullptr;
    e.ob = reinterpret_cast<const unsigned char*>(L->out) + ((long long)it.px0 * e.o_cs + L->out_coff) * 2;
    e.r1b = e.has1 ? reinterpret_cast<const unsigned char*>(L->res1) + ((long long)it.px0 * e.r1_cs + L->r1_coff) * 2 : e.ob;
    e.r2b = e.has2 ? reinterpret_cast<const unsigned char*>(L->res2) + ((long long)it.px0 * e.r2_cs + L->r2_coff) * 2 : e.ob;
    return e;
  };
  auto epilogue = [&](const Item& it, const Epi& e) {
    const int act = e.act;
    const float s1 = e.s1, s2 = e.s2;
    const int o_cs = e.o_cs, r1_cs = e.r1_cs, r2_cs = e.r2_cs;
    const bool has1 = e.has1, has2 = e.has2;
    auto mk = [](const unsigned char* q) {
      const unsigned long long u = reinterpret_cast<unsigned long long>(q);
      const unsigned lo = rfl((unsigned)u), hi = rfl((unsigned)(u >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFF0, 0x00020000);
    };
    const auto rs_o = mk(e.ob), rs_r1 = mk(e.r1b), rs_r2 = mk(e.r2b);
    const int ox = it.x0 + pc;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      if (nt >= it.NT) break;
      const int ch0 = nt * 32 + 8 * g;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = it.y0 + 4 * wave + i;
        const bool ok = oy < it.H && ox < it.W;
        const unsigned rel = (unsigned)(oy * it.W + ox);
        u32x4 r1v = {0u, 0u, 0u, 0u}, r2v = r1v;
        if (has1) r1v = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, ok ? (rel * r1_cs + ch0) * 2 : OOB, 0, AUXC);
        if (has2) r2v = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, ok ? (rel * r2_cs + ch0) * 2 : OOB, 0, AUXC);
        float v[8];
        v[0] = acc[i][nt * 2][0]; v[1] = acc[i][nt * 2][1]; v[2] = acc[i][nt * 2][2]; v[3] = acc[i][nt * 2][3];
        v[4] = acc[i][nt * 2 + 1][0]; v[5] = acc[i][nt * 2 + 1][1]; v[6] = acc[i][nt * 2 + 1][2]; v[7] = acc[i][nt * 2 + 1][3];
        if (act == ACT_LRELU) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], v[q] * 0.2f);
        } else if (act == ACT_SILU) {
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = apply_act(v[q], ACT_SILU);
        }
        if (has1) {
          const _Float16* r = reinterpret_cast<const _Float16*>(&r1v);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = v[q] * s1 + (float)r[q];
        }
        if (has2) {
          const _Float16* r = reinterpret_cast<const _Float16*>(&r2v);
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = v[q] * s2 + (float)r[q];
        }
        union { u32x4 u; _Float16 h[8]; } ov;
#pragma unroll
        for (int q = 0; q < 8; ++q) ov.h[q] = (_Float16)v[q];
        __builtin_amdgcn_raw_buffer_store_b128(ov.u, rs_o, ok ? (rel * o_cs + ch0) * 2 : OOB, 0, AUXC);
      }
    }
  };
  auto publish = [&](int tile, int layer) {        // every wave has drained its stores and the workgroup has met at a barrier since
    if ((FFP_TRUNK_SKIP & 64)) return;
    if (COH && tid == 0) __hip_atomic_store(a.done + tile, (unsigned)(layer + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- prologue: the first item --------------------------------------------------------------------------------------------------------------
  if (wave == 0) control_wait(0);
  __syncthreads();
  if (rfl(desc[0]) != D_READY) return;
  ck = 0;
  Item cur = load_item(0);
  setup_dma(cur);
  issue_dma(cur, 0, lds0);
  wait_all();
  __syncthreads();
  init_acc(cur, smem);
  int c = 0;
  unsigned stage = 0;
  int pub_tile = -1, pub_layer = 0;
  // One iteration = one chunk. Whether the NEXT item can be prefetched is decided from a snapshot that wave 0 wrote during the PREVIOUS
  // iteration (snap[parity]): every wave of the workgroup takes the same branch, whatever wave 0's control step is doing meanwhile.
#if FFP_TRUNK_DBG
  unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();      // control, next-item set-up, chunk, epilogue, wait, barrier, slow path
  unsigned n_iter = 0, n_slow = 0;
#define TSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[k] += t_ - tprev; tprev = t_; }
#define TDUMP() if (blockIdx.x == 0 && lane == 0 && wave < 2) { unsigned* o = a.queue + 2 + wave * 7; for (int k_ = 0; k_ < 7; ++k_) o[k_] = (unsigned)(tsum[k_] / (n_iter ? n_iter : 1)); \
                  if (wave == 0) { a.queue[1] = 0; } if (wave == 1) { o[6] = n_iter | (n_slow << 16); } }
#else
#define TSTAMP(k)
#define TDUMP()
#endif
  bool have_nxt = false;
  Item nxt = cur;
  for (unsigned itn = 0;; ++itn) {
    const bool last = c == cur.NC - 1;
    const bool pre = last && have_nxt;
    // is the item after this one READY? Decided from a snapshot wave 0 wrote during the PREVIOUS iteration: every wave takes the same branch
    const bool nxt_ready = c == cur.NC - 2 && rfl(snap[itn & 1u]) != 0;
    if (wave == 0) control_issue();
    TSTAMP(0)
    Epi ep = {};
    if (last) ep = load_epi(cur);
    TSTAMP(1)
    // the chunk that is staged while this one is multiplied: the item's next chunk, or chunk 0 of the next item, or nothing
    const bool stg = !last || pre;
    const int sc = last ? 0 : c + 1;
    const unsigned sst = lds0 + (stage ^ 1u) * G::STAGE;
    const unsigned char* sb = smem + stage * G::STAGE;
    auto between = [&](int step) { if (stg) dma_piece(last ? nxt : cur, sc, sst, step); };
    if (cur.NT == 2) chunk(sb, std::integral_constant<int, 4>{}, between);
    else chunk(sb, std::integral_constant<int, 2>{}, between);
    TSTAMP(2)
    if (nxt_ready) {                               // the item's last chunk has been requested: the staging registers now describe the NEXT item
      nxt = load_item(ck + 1);                     // (its descriptor reads and scalar loads overlap the wait below instead of leading the last iteration)
      setup_dma(nxt);
      have_nxt = true;
    }
    if (last) {
      if (!(FFP_TRUNK_SKIP & 1)) epilogue(cur, ep);
      TSTAMP(3)
      // this wave's DMA pieces are older than the epilogue's loads and stores: all but the 4 NT stores must have completed
      if (cur.NT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      wait_all();
      if (wave == 0) control_consume();            // everything this wave has asked for is back (the full wait above)
    }
    if (tid == 0) {
      const int nseq = pre ? ck + 2 : ck + 1;      // the item that will be "next" in the following iteration
      snap[(itn + 1u) & 1u] = get4(s_st, nseq & 3) == D_READY ? 1 : 0;
    }
    TSTAMP(4)
    __syncthreads();
    TSTAMP(5)
#if FFP_TRUNK_DBG
    ++n_iter;
#endif
    if (pub_tile >= 0) {                           // the previous item: its stores were drained by this (non-final) chunk's full wait
      publish(pub_tile, pub_layer);
      pub_tile = -1;
    }
    if (!last) { ++c; stage ^= 1u; continue; }
    if (tid == 0) desc[(ck & 3) * 16] = D_EMPTY;   // the finished item's slot: sequence ck + 4 will be fetched into it
    if (wave == 0) put4(s_st, ck & 3, D_EMPTY);
    have_nxt = false;
    if (pre) {
      pub_tile = cur.tile; pub_layer = cur.layer;
      cur = nxt; ++ck; c = 0; stage ^= 1u;
      init_acc(cur, smem + stage * G::STAGE);
      continue;
    }
    // the next item is not ready (or there is none): finish this one for good, then wait for it
    wait_all();
    __syncthreads();
    publish(cur.tile, cur.layer);
    if (wave == 0) control_wait(ck + 1);
    __syncthreads();
    if (rfl(desc[((ck + 1) & 3) * 16]) != D_READY) { TDUMP() return; }
    ++ck;
    cur = load_item(ck);
    setup_dma(cur);
    stage ^= 1u;
    issue_dma(cur, 0, lds0 + stage * G::STAGE);
    if (tid == 0) { snap[0] = 0; snap[1] = 0; }
    wait_all();
    __syncthreads();
    init_acc(cur, smem + stage * G::STAGE);
    c = 0;
#if FFP_TRUNK_DBG
    ++n_slow;
#endif
    TSTAMP(6)
  }
}

}  // namespace

void conv_trunk_init() {
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS));
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS));
}

bool conv_trunk_enabled() {
  static const bool on = [] { const char* e = getenv("FFP_TRUNK"); return !(e && e[0] == '0'); }();
  return on;
}

bool conv_trunk_layer_ok(const ConvOp& op) {
  const PackedConv& pc = *op.pc;
  const int cpl = 8;
  // one image table serves input, output and residuals: the same Level, or (single-operator hooks) exact-mode levels of identical geometry
  auto same = [](const Level* x, const Level* y) { return x == y || (x && y && !x->capacity() && !y->capacity() && x->n == y->n && x->h == y->h && x->w == y->w && x->off == y->off); };
  bool ok = pc.w16.p != nullptr && pc.dt == F16 && pc.k == 3 && op.stride == 1 && op.up == 0 && !op.has_up2 && pc.cin % 32 == 0 && pc.cin >= 64 &&
            (pc.cout == 32 || pc.cout == 64) && op.in.dt == F16 && op.out.dt == F16 && same(op.in.lvl, op.out.lvl) && op.in.cs % cpl == 0 && op.in.coff % cpl == 0 &&
            op.out.cs % cpl == 0 && op.out.coff % cpl == 0;
  if (op.has_res1) ok = ok && op.res1.dt == F16 && same(op.res1.lvl, op.out.lvl) && op.res1.cs % cpl == 0 && op.res1.coff % cpl == 0;
  if (op.has_res2) ok = ok && op.res2.dt == F16 && same(op.res2.lvl, op.out.lvl) && op.res2.cs % cpl == 0 && op.res2.coff % cpl == 0;
  return ok;
}

TrunkPlan::TrunkPlan(const std::vector<ConvOp>& ops) {
  FFP_CHECK(!ops.empty(), FFP_ERR_ARG, "trunk: no layers");
  lvl = ops[0].out.lvl;
  std::vector<TrunkLayer> h(ops.size());
  for (size_t i = 0; i < ops.size(); ++i) {
    const ConvOp& op = ops[i];
    FFP_CHECK(conv_trunk_layer_ok(op) && op.out.lvl == lvl, FFP_ERR_ARG, "trunk: layer %s cannot run in the fused launch", op.pc->name.c_str());
    TrunkLayer& L = h[i];
    std::memset(&L, 0, sizeof(L));
    L.in = op.in.ptr; L.out = op.out.ptr; L.res1 = op.has_res1 ? op.res1.ptr : nullptr; L.res2 = op.has_res2 ? op.res2.ptr : nullptr;
    L.wpk = op.pc->w16.p; L.bias = op.pc->bias.as<float>();
    L.in_cs = op.in.cs; L.in_coff = op.in.coff; L.cin = op.pc->cin;
    L.out_cs = op.out.cs; L.out_coff = op.out.coff; L.cout = op.pc->cout;
    L.r1_cs = op.res1.cs; L.r1_coff = op.res1.coff; L.r2_cs = op.res2.cs; L.r2_coff = op.res2.coff;
    L.s1 = op.s1; L.s2 = op.s2; L.act = op.act;
  }
  n_layers = (int)ops.size();
  d_layers.alloc(sizeof(TrunkLayer) * h.size());
  FFP_HIP(hipMemcpy(d_layers.p, h.data(), sizeof(TrunkLayer) * h.size(), hipMemcpyHostToDevice));
}

void TrunkPlan::launch(hipStream_t st, int dbg) {
  using G = TG;
  TrunkArgs a{};
  int n_tiles = 0;
  a.tiles = lvl->tile_table_packed(G::TH, &n_tiles, &a.n_tiles_dev, st);
  if (n_tiles == 0) return;
  a.ntiles_host = n_tiles;
  a.layers = d_layers.as<TrunkLayer>();
  a.n_layers = n_layers;
  // [queue head, error word, padding to 64 B | done[tiles]]: zeroed before EVERY launch (a memset node under graph replay)
  const size_t need = 64 + sizeof(unsigned) * (size_t)n_tiles;
  const size_t nb = (need + 63) / 64 * 64;
  if (sync.n < nb) sync.alloc(nb);
  FFP_HIP(hipMemsetAsync(sync.p, 0, nb, st));
  a.queue = sync.as<unsigned>();
  a.done = sync.as<unsigned>() + 16;
  static const int env_dbg = [] { const char* e = getenv("FFP_TRUNK_DBGMASK"); return e ? atoi(e) : 0; }();      // diagnostic builds only (FFP_TRUNK_DBG)
  a.dbg = dbg | env_dbg;
  static const bool dump = [] { const char* e = getenv("FFP_TRUNK_DUMP"); return e && e[0] == '1'; }();
  // one workgroup per CU; capacity-mode levels launch for the capacity (workgroups beyond the batch's items find the queue empty)
  const long long items = (long long)n_tiles * n_layers;
  const unsigned grid = (unsigned)std::min<long long>(256, items);
  if (n_layers > 1) hipLaunchKernelGGL(conv_trunk_kernel<true>, dim3(grid), dim3(512), G::LDS, st, a);
  else hipLaunchKernelGGL(conv_trunk_kernel<false>, dim3(grid), dim3(512), G::LDS, st, a);
  FFP_HIP(hipGetLastError());
  if (dump) {                                          // diagnostic: the queue head, the error word and the debug words behind them
    unsigned h[16];
    FFP_HIP(hipMemcpyAsync(h, sync.p, sizeof(h), hipMemcpyDeviceToHost, st));
    FFP_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "trunk launch: tiles %d layers %d grid %u | head %u err %u | per iteration, wave 0 then wave 1: control, next-item set-up, chunk, epilogue, wait, barrier, slow path (per iteration); last word: iterations | slow paths << 16 |", n_tiles, n_layers, grid, h[0], h[1]);
    for (int i = 2; i < 16; ++i) fprintf(stderr, " %u", h[i]);
    fprintf(stderr, "\n");
  }
}

unsigned TrunkPlan::errors(hipStream_t st) {
  if (!sync.p) return 0;
  unsigned e = 0;
  FFP_HIP(hipMemcpyAsync(&e, sync.as<unsigned>() + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  FFP_HIP(hipStreamSynchronize(st));
  return e;
}

}  // namespace ffp
