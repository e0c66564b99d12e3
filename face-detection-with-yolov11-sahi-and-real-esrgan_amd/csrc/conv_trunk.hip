// conv_trunk.hip — the Real-ESRGAN body (every 3x3 fp16 conv of the 69 residual dense blocks: 345 layers) as ONE persistent launch.
//
// What round 4 measured first (tools/probes/stage_probe.hip, profiles/r04_staging_rate_probe.txt; in-kernel stamps, profiles/r04_trunk_stamps.txt):
//   * the global -> LDS staging rate of a CU scales with the NUMBER OF WAVES that issue loads (~4 B/clk per wave) up to ~32 B/clk per CU on
//     L2 hits and ~23 B/clk out of the Infinity Cache, whatever the prefetch depth, access pattern or load form. conv_rows16_kernel stages
//     39 KiB per 288 MFMAs = 34 B/clk per CU at the full matrix rate: it is STAGING-BOUND (phase probe: staging alone 26 us, MFMAs alone
//     23.6 us, together 43.8 us), and a role split with four producer waves (conv_rows16pc) delivers 12-14 B/clk;
//   * an LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB) takes ~230 cycles to ISSUE while the CU's address unit is at that cap, and an
//     in-order wave's MFMAs wait behind it: a 72-MFMA step took 1,589 cycles without the DMA instructions in its stream and 3,404 with them.
// So this kernel
//   * stages fewer bytes per MFMA: one workgroup per CU owns a 32 x 16 pixel tile, so a chunk's weight fragments serve twice the pixels
//     of conv_rows16's tile: 57 KiB per 576 MFMAs = 25 B/clk per CU at the full matrix rate;
//   * splits the workgroup's TWELVE waves by role: four COMPUTE waves, one per SIMD, each owning 8 output rows x 16 columns (64 accumulator
//     registers, 0.46 KiB of fragment reads per MFMA, conv_rows16: 0.5) whose stream is fragment reads + 144 MFMAs per step and nothing
//     else, and EIGHT LOADER waves that issue every LDS-DMA piece (eight, because the staging rate is per issuing wave) — their issue stalls
//     stall nobody's MFMAs. (Sixteen waves — eight compute waves of 4 rows — have 128 registers each: hipcc spilled 46-142 of them, and
//     scratch traffic inside the MFMA stream is one more stalled vector-memory instruction: 179 us against conv_rows16's 96.) No staging VGPRs, no ds_write traffic; the LDS image is written lane-linear and the bank-conflict-free XOR swizzle
//     is applied to the SOURCE channel slot; out-of-image lanes use an out-of-range buffer offset (zeros: the conv's zero padding);
//   * is ONE STREAM OF STEPS per workgroup — step = (item, 32-channel input chunk) — over three pixel stages and two weight stages: while
//     step q multiplies, the loaders request step q + 1's weights and the pixel chunk of step q + 2, across item boundaries;
//   * walks (layer, tile, 32-channel output block) items of ALL the layers it is given out of one global queue (an atomic counter,
//     layer-major order): no launch boundary, no per-layer tail, no per-launch skeleton. A 64-channel layer (conv5 of every dense block) is
//     two items per tile, neighbours in the queue. Layer l + 1 of a tile needs layer l of the tile and of its eight neighbours in the same
//     image: `done[tile]` counts finished (layer, block) items (one lane's atomic add after every compute wave has drained its stores),
//     polled ahead of time by wave 0 — ids are fetched three items ahead, their tile entries one step later, the nine counters of every
//     fetched item every step, each request consumed one step after it was issued — so that a dependency never stalls the MFMA stream
//     unless it is really late (small batches), and then only this workgroup. Items are taken in queue order and depend only on
//     EARLIER items: the grid cannot deadlock whatever part of it is resident (a second persistent launch on the device, the detector's
//     kernels). Activations written inside the launch are stored write-through (sc1) and read with sc1 loads (DMA and residuals):
//     per-CU L1s are never refreshed and per-XCD L2s are not coherent (MI355X guide); weights and tables are read-only.
// Arithmetic is conv_rows16_kernel's instruction for instruction (v_mfma_f32_16x16x32_f16, accumulators start from the bias, chunks
// ascending, taps kx-major inside a chunk, register epilogue): results are BIT-IDENTICAL, which is the kernel's parity oracle
// (tests/test_gpu_trunk.py) on top of the usual one.
// Hardware facts this file relies on, each probed (tools/probes/dma_oob_probe.hip): out-of-range lanes of an LDS-DMA write zeros; M0 takes a
// full LDS byte address; the scalar offset IS part of the range check; a buffer ATOMIC at offset 0xFFFFFFFF faults (0x80000000 does not).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "conv_args.hpp"
#include "trunk.hpp"

#ifndef FFP_TRUNK_DBG
#define FFP_TRUNK_DBG 0        // 1: diagnostic build — s_memtime stamps per phase of every step, sums of workgroup 0 behind the queue words (FFP_TRUNK_DUMP=1 prints them)
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
  static constexpr int PIX = PIX_PIECES * 1024;               // one pixel stage (a 32-channel chunk of the halo tile)
  static constexpr int WB = 18 * 1024;                        // one weight stage: the fragments of one 32-channel block and chunk, [tap][M-tile][lane] x 16 B
  static constexpr int WOFF = 3 * PIX;                        // LDS: [3 pixel stages][2 weight stages][descriptor ring, snapshot words][layer table][2 bias slots]
  static constexpr int MISC = WOFF + 2 * WB;                  // 156,672
  static constexpr int CUM = MISC + 4 * 64 + 64;              // blocks before layer l, l = 0 .. n_layers (MAXL + 1 ints)
  static constexpr int MAXL = 511;
  static constexpr int BIAS = CUM + (MAXL + 1) * 4;           // 2 bias slots, 1 KiB apart: an LDS-DMA piece always writes 64 lanes x 16 B (zeros beyond the 128 B)
  static constexpr int LDS = BIAS + 2 * 1024;                 // 161,088 B: one workgroup per CU
  static constexpr int ROWB = HC * 64;                        // bytes of a halo row
};

enum : int { D_EMPTY = 0, D_KNOWN = 1, D_READY = 2, D_END = 3 };
// descriptor slot (16 ints in LDS): 0 state, 1 output block, 2 layer, 3 tile index, 4 first pixel of the image, 5 y0 | x0 << 16, 6 h | w << 16,
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
  // the descriptor, the scalar offset and the LDS address are wave-uniform by construction; where hipcc's uniformity analysis cannot see it
  // (values merged over the walk's loop) it would hand the asm statement vector registers: say so once more (folds away where it is known)
  rs[0] = rfl(rs[0]); rs[1] = rfl(rs[1]); rs[2] = rfl(rs[2]); rs[3] = rfl(rs[3]);
  soff = rfl(soff); lds = rfl(lds);
  if constexpr (SC1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen sc1 lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds), "s"(soff) : "memory");
}

struct Item {            // wave-uniform description of a work item (SGPRs)
  int layer, tile, nb;   // nb: the 32-channel output block
  int y0, x0;
  int NC;                // 32-channel input chunks = steps of the item
  int H, W;              // image size
  int px0;               // first pixel of the image in the level
  int xchg;              // the layer has no residual inputs: its finished tile can go to the loaders through LDS (see the walk)
};

template <bool COH>      // COH: several layers in one launch — sc1 loads / stores of activations and the done[] protocol
__global__ void __launch_bounds__(768, 3) conv_trunk_kernel(const TrunkArgs a) {
  using G = TG;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem;
  int* const desc = reinterpret_cast<int*>(smem + G::MISC);   // 4 slots x 16 ints
  int* const snap = desc + 64;                                  // [iteration parity]: is the item after the ones this workgroup holds READY?
  int* const cumtab = reinterpret_cast<int*>(smem + G::CUM);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = rfl(tid >> 6);
  const bool loader = wave >= 4;                               // waves 4..11 stage, waves 0..3 multiply — one per SIMD (wave 0 also runs the control steps)
  const int w8 = (wave - 4) & 7;                               // loader index 0..7
  const int pc = lane & 15, g = lane >> 4;
  typedef const __attribute__((address_space(4))) TrunkLayer* LPtr;
  const LPtr layers = reinterpret_cast<LPtr>(reinterpret_cast<unsigned long long>(a.layers));

  const int n_tiles = a.n_tiles_dev ? sload(a.n_tiles_dev, 0) : a.ntiles_host;
  const unsigned total = (unsigned)n_tiles * (unsigned)a.n_blocks;
  if (total == 0) return;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr int AUXC = COH ? 16 : 0;                          // sc1 on activation loads / stores of a multi-layer launch

  // fragment read offsets (compute wave w: output rows 8w..8w+7): halo rows 8 * wave + r (r = 0..9), column pc + kx, slot g
  unsigned boff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = pc + kx;
    boff[kx] = (unsigned)(((8 * (wave & 3)) * G::HC + hx) * 64 + ((g ^ ((hx >> 1) & 2)) << 4));
  }
  unsigned isrc[6];        // loader waves 1..7: byte offset of the lane's 16 bytes of chunk 0 from the image's first pixel record (+ in_coff), or OOB

  // ---- control (wave 0): the queue, the tile entries and the dependency counters of the next three items -----------------------------------------
  int ck = -1;             // sequence number (within this workgroup) of the CURRENT item (-1: none yet); slot = seq & 3
  int fk = 0, fstate = 0;  // fetcher: next sequence number to fetch; 0 idle, 1 queue id asked, 2 id known, 3 tile entry asked, 4 queue exhausted
  int f_id = 0;            // pending atomic result (lane 0)
  u32x4 f_tile = {0u, 0u, 0u, 0u};
  int f_layer = 0, f_t = 0, f_nb = 0, f_need = 0;
  int f_ly = 0;            // layer cursor of the fetcher: queue ids only grow, so the layer of an id is found by stepping this forward
  // wave 0's own copy of the four descriptor slots (scalar registers; LDS holds what the other waves read): state, finished blocks the item's
  // neighbourhood must show, tile index, tile position and grid packed as tx | ty << 8 | nx << 16 | ny << 24 — a control step reads no slot from LDS
  int s_st[4] = {0, 0, 0, 0}, s_need[4] = {0, 0, 0, 0}, s_tt[4] = {0, 0, 0, 0}, s_geo[4] = {0, 0, 0, 0};
  auto get4 = [](const int (&v)[4], int k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3]; };
  auto put4 = [](int (&v)[4], int k, int x) { v[0] = k == 0 ? x : v[0]; v[1] = k == 1 ? x : v[1]; v[2] = k == 2 ? x : v[2]; v[3] = k == 3 ? x : v[3]; };
  int p_base = -1;         // polls in flight: sequence number of lane group 0 (-1: none)
  unsigned p_val = 0;
  int p_need = 0;          // per lane: the count its polled counter must have reached
  bool p_act = false;      // per lane: it polled a counter
  const auto rs_sync = __builtin_amdgcn_make_buffer_rsrc(a.queue, 0, 64 + 4 * n_tiles, 0x00020000);          // [queue | done[]]
  const auto rs_tiles = __builtin_amdgcn_make_buffer_rsrc(const_cast<int4*>(a.tiles), 0, 16 * n_tiles, 0x00020000);
  // "nothing to ask" offset of the control step's dword / atomic operations: aligned and beyond any num_records (0xFFFFFFFF + 4 wraps in
  // a 32-bit range check; a buffer ATOMIC at that offset faulted with a memory aperture violation on gfx950, 16-byte loads do not)
  constexpr unsigned OOBA = 0x80000000u;
  // A control step has two halves. control_issue() — at the top of a step — asks: the queue for the next id (at most three items ahead),
  // or the tile table for the id it got, and (multi-layer launches) the dependency counters of every fetched, not yet ready item.
  // control_consume() — at the END of the step, after wave 0's vmcnt(0) — reads the answers. Every request is a buffer operation issued
  // unconditionally inside the step (an out-of-range offset where there is nothing to ask): behind exec-masked loads, or for results carried
  // over the loop's back edge, hipcc waits for vmcnt(0) at the point of use, and that wait would also cover the epilogue's stores (1-2 us:
  // measured 1,500 cycles per step with a one-piece form). An item's last step (stores still in flight at its end) skips the consume;
  // nothing new is asked until the answers have been read.
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
      p_need = get4(s_need, k);
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
        // the layer of a queue id: ids [n_tiles * cum[l], n_tiles * cum[l + 1]) belong to layer l, (tile, block) = divmod(rest, blocks of the layer)
        // (LDS reads are per-lane values to hipcc: without the readfirstlanes the layer cursor — and every piece of control state derived from
        // it — counts as divergent and moves from scalar to vector registers, in all twelve waves)
        while (id >= (unsigned)n_tiles * (unsigned)rfl(cumtab[f_ly + 1])) ++f_ly;
        const int c0 = rfl(cumtab[f_ly]), nblk = rfl(cumtab[f_ly + 1]) - c0;
        const unsigned r = id - (unsigned)n_tiles * (unsigned)c0;
        f_layer = f_ly;
        f_need = c0;
        f_t = nblk == 2 ? (int)(r >> 1) : (int)r;
        f_nb = nblk == 2 ? (int)(r & 1u) : 0;
        f_tile_off = (unsigned)f_t * 16u;
        fstate = 2;
      }
    } else if (fstate == 3) {
      const int e0 = rfl((int)f_tile[0]), e1 = rfl((int)f_tile[1]), e2 = rfl((int)f_tile[2]), txy = rfl((int)f_tile[3]);
      const int k = fk & 3;
      const int state = (COH && f_need > 0) ? D_KNOWN : D_READY;        // the first layer of a launch depends on earlier launches only
      put4(s_st, k, state); put4(s_need, k, f_need); put4(s_tt, k, f_t);
      put4(s_geo, k, ((e1 >> 16) >> 4) | (((e1 & 0xFFFF) / G::TH) << 8) | ((txy & 0xFF) << 16) | (((txy >> 16) & 0xFF) << 24));
      if (lane == 0) {
        int* d = desc + k * 16;
        d[1] = f_nb; d[2] = f_layer; d[3] = f_t; d[4] = e0; d[5] = e1; d[6] = e2; d[7] = txy;
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
  // blocking form (the walk's first item, late dependencies): until sequence `sq` is READY or the queue has ended. Bounded: a spin that never
  // ends would hang the device — it gives up, raises the error word and ends this workgroup's walk instead
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
  for (int i = tid; i <= a.n_layers; i += 768) cumtab[i] = i < a.n_layers ? a.layers[i].cum : a.n_blocks;
  __syncthreads();

  // ---- item set-up: descriptor slot -> SGPRs ------------------------------------------------------------------------------------------------
  auto load_item = [&](int sq) {
    const int4 d0 = *reinterpret_cast<const int4*>(desc + (sq & 3) * 16), d1 = *reinterpret_cast<const int4*>(desc + (sq & 3) * 16 + 4);
    Item it;
    it.nb = rfl(d0.y); it.layer = rfl(d0.z); it.tile = rfl(d0.w);
    it.px0 = rfl(d1.x);
    const int yx = rfl(d1.y), hw = rfl(d1.z);
    it.y0 = yx & 0xFFFF; it.x0 = (int)((unsigned)yx >> 16);
    it.H = hw & 0xFFFF; it.W = (int)((unsigned)hw >> 16);
    it.NC = layers[it.layer].cin >> 5;
    it.xchg = (layers[it.layer].res1 == nullptr && layers[it.layer].res2 == nullptr) ? 1 : 0;
    return it;
  };
  // ---- staging (loader waves 8..15). The workgroup's work is ONE STREAM OF STEPS: step q = (item, 32-channel input chunk c), 72 MFMAs per compute
  // wave. Step q multiplies the chunk's halo tile (pixel stage q % 3) by the item's weight fragments of that chunk (weight stage q % 2). While it
  // runs, the loaders request step q + 1's weights and the pixel chunk of step q + 2 — across items: the stream runs into the next item as soon
  // as that item is known and its dependencies are met.
  u32x4 rs_in, rs_w, rs_b;                         // pixel source (the item `pix_item` describes), weight / bias source (the item of the weight cursor)
  int pix_item = -1;                               // sequence number of the item isrc[] / rs_in describe (-1: none)
  int w_soff0 = 0;                                 // weight cursor: byte offset of (block nb, chunk 0) in the layer's packed fragments
  auto setup_pix = [&](const Item& it, int seq) {
    const LPtr L = layers + it.layer;
    const int cs = L->in_cs;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(L->in) + ((long long)it.px0 * cs + L->in_coff) * 2;
    rs_in = make_rsrc(inb, 0x7FFFFFF0u);
#pragma unroll
    for (int i = 0; i < 6; ++i) {                    // pixel piece (w8 - 1) + 7 i of the 39 (loaders 1..7; loader 0 runs the control steps instead: see dma_pix)
      const int p = w8 - 1 + 7 * i;
      const int q = p * 16 + (lane >> 2);            // lane -> halo pixel q = piece * 16 + lane / 4, LDS slot position lane % 4
      const int hy = (q * 3641) >> 16, hx = q - hy * G::HC;          // q / 18 for q < 640
      const int sl = (lane & 3) ^ ((hx >> 1) & 2);                   // which holds SOURCE slot (lane % 4) ^ ((hx >> 1) & 2): conflict-free fragment reads
      const int iy = it.y0 - 1 + hy, ix = it.x0 - 1 + hx;
      const bool ok = w8 > 0 && p < G::PIX_PIECES && q < G::HR * G::HC && (unsigned)iy < (unsigned)it.H && (unsigned)ix < (unsigned)it.W;
      isrc[i] = ok ? (unsigned)((iy * it.W + ix) * cs * 2 + sl * 16) : OOB;
    }
    pix_item = seq;
  };
  auto setup_w = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    const int nblk = L->cout >> 5;
    rs_w = make_rsrc(L->wpk, (unsigned)(nblk * it.NC * G::WB));
    rs_b = make_rsrc(L->bias + it.nb * 32, 128u);
    w_soff0 = it.nb * it.NC * G::WB;
  };
  // one wave-piece (64 lanes x 16 B) per call. Pixel chunk c of the item isrc[] describes -> pixel stage at LDS address `st`: pieces (w8 - 1) + 7 i.
  // Loader 0 stages NO pixel pieces: it runs the control steps, and at the end of a step its vector-memory queue must hold nothing that is meant
  // to stay in flight — reading the control answers is a vmcnt(0) as far as hipcc can tell.
  auto dma_pix = [&](int c, unsigned st, int i) {
    if ((FFP_TRUNK_SKIP & 4)) return;
    const int p = w8 - 1 + 7 * i;
    if (w8 > 0 && p < G::PIX_PIECES) dma16<COH>(rs_in, isrc[i], (unsigned)(c * 64), st + (unsigned)(p << 10));
  };
  const int pix_pieces = w8 == 0 ? 0 : (w8 <= 4 ? 6 : 5);      // 39 = 4 x 6 + 3 x 5
  // weight fragments of chunk c of the weight cursor's block -> weight stage at `st`: pieces w8 + 8 j < 18; with first = true also the block's
  // bias (128 B; loader 7's third slot) -> bias slot at `bst`
  auto dma_w = [&](int c, unsigned st, int j, bool first, unsigned bst) {
    if ((FFP_TRUNK_SKIP & 4)) return;
    const int q = w8 + 8 * j;
    if (q < 18) dma16<false>(rs_w, (unsigned)lane * 16u, (unsigned)(w_soff0 + ((c * 18 + q) << 10)), st + (unsigned)(q << 10));
    else if (first && q == 23) dma16<false>(rs_b, lane < 8 ? (unsigned)lane * 16u : OOB, 0u, bst);
  };
  // wait until at most `keep` of this wave's vector-memory operations are outstanding (the youngest ones: a step's pixel pieces are requested
  // last and belong to the step after next — they are NOT waited for at the end of the step that requests them)
  auto wait_keep = [&](int keep) {
    switch (keep) {
      case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
  };
  auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- arithmetic (compute waves 0..3): conv_rows16_kernel's sums for one 32-channel output block, EIGHT output rows per wave --------------------
  f32x4 acc[8][2];
  auto init_acc = [&](const unsigned char* bias) {      // the bias came by DMA with the item's first weight stage: no global load at an item's start
    const float* b = reinterpret_cast<const float*>(bias);
    const float4 b0 = *reinterpret_cast<const float4*>(b + 8 * g);
    const float4 b1 = *reinterpret_cast<const float4*>(b + 8 * g + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i][0] = f32x4{b0.x, b0.y, b0.z, b0.w};
      acc[i][1] = f32x4{b1.x, b1.y, b1.z, b1.w};
    }
  };
  // A chunk is 18 tap steps of 8 MFMAs: for kx: for row group grp (rows 4 grp .. + 3): for ky — every accumulator still meets its taps in
  // conv_rows16's order (kx-major), so the sums are the same bits. Tap step (kx, grp, ky) multiplies input rows 4 grp + ky + i (i = 0..3) of
  // the kx-shifted halo rows 0..9. Row fragments live in EIGHT rolling slots, slot = (10 kx + row) % 8, each read from LDS one tap step before
  // its first use, into a slot whose previous row is dead (first uses within a kx, by t = 3 grp + ky: t0 rows 0-3, t1 row 4, t2 row 5,
  // t3 rows 6-7, t4 row 8, t5 row 9): 10 row reads per kx for 48 MFMAs (0.21 KiB per MFMA, conv_rows16: 0.25) and the chunk's 18 weight
  // fragments twice (once per row group; 0.25 KiB per MFMA), one tap step ahead.
  auto chunk = [&](const unsigned char* sp, const unsigned char* sw) {
    uint4 bq[8] = {}, aq[2][2] = {};
    auto ldB1 = [&](int kx, int r) {
      if ((FFP_TRUNK_SKIP & 16)) { asm volatile("" : "+v"(bq[(10 * kx + r) & 7].x), "+v"(bq[(10 * kx + r) & 7].y), "+v"(bq[(10 * kx + r) & 7].z), "+v"(bq[(10 * kx + r) & 7].w)); return; }
      bq[(10 * kx + r) & 7] = *reinterpret_cast<const uint4*>(sp + boff[kx] + r * G::ROWB);
    };
    auto ldA = [&](int T, int q) {                 // tap step T = 6 kx + 3 grp + ky: the packed tap index is ky * 3 + kx
      const int kx = T / 6, ky = T % 3, tap = ky * 3 + kx;
      if ((FFP_TRUNK_SKIP & 16)) {
#pragma unroll
        for (int m = 0; m < 2; ++m) asm volatile("" : "+v"(aq[q][m].x), "+v"(aq[q][m].y), "+v"(aq[q][m].z), "+v"(aq[q][m].w));
        return;
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) aq[q][m] = *reinterpret_cast<const uint4*>(sw + lane * 16 + ((tap * 2 + m) << 10));
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) ldB1(0, r);
    ldA(0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int T = 0; T < 18; ++T) {
      const int kx = T / 6, t = T % 6, grp = t / 3, ky = t % 3;
      if (T + 1 < 18) ldA(T + 1, (T + 1) & 1);
      if (t == 0) ldB1(kx, 4);
      if (t == 1) ldB1(kx, 5);
      if (t == 2) { ldB1(kx, 6); ldB1(kx, 7); }
      if (t == 3) ldB1(kx, 8);
      if (t == 4) ldB1(kx, 9);
      if (t == 5 && kx < 2) { ldB1(kx + 1, 0); ldB1(kx + 1, 1); ldB1(kx + 1, 2); ldB1(kx + 1, 3); }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          if ((FFP_TRUNK_SKIP & 2)) continue;
          union { uint4 u; f16x8 h; } ua, ub;
          ua.u = aq[T & 1][m]; ub.u = bq[(10 * kx + 4 * grp + ky + i) & 7];
          acc[4 * grp + i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, acc[4 * grp + i][m], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        __builtin_amdgcn_sched_group_barrier(0x7F6, 1, 0);     // one other instruction
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- epilogue: lane (pc, g) holds channels 32 nb + 8 g .. + 7 of pixel (row 8 * wave + i, column pc); 8 stores per compute wave ---------------
  struct Epi {             // the epilogue's layer parameters (scalar loads issued BEFORE the last step's MFMAs: their latency is off the item's tail)
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
    const int ch0 = it.nb * 32 + 8 * g;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i == 4) __builtin_amdgcn_sched_barrier(0);             // four rows at a time: hipcc otherwise hoists every row's residual loads (64 registers)
      const int oy = it.y0 + 8 * (wave & 3) + i;
      const bool ok = oy < it.H && ox < it.W;
      const unsigned rel = (unsigned)(oy * it.W + ox);
      u32x4 r1v = {0u, 0u, 0u, 0u}, r2v = r1v;
      if (has1) r1v = __builtin_amdgcn_raw_buffer_load_b128(rs_r1, ok ? (rel * r1_cs + ch0) * 2 : OOB, 0, AUXC);
      if (has2) r2v = __builtin_amdgcn_raw_buffer_load_b128(rs_r2, ok ? (rel * r2_cs + ch0) * 2 : OOB, 0, AUXC);
      float v[8];
      v[0] = acc[i][0][0]; v[1] = acc[i][0][1]; v[2] = acc[i][0][2]; v[3] = acc[i][0][3];
      v[4] = acc[i][1][0]; v[5] = acc[i][1][1]; v[6] = acc[i][1][2]; v[7] = acc[i][1][3];
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
  };
  // ---- the epilogue's other form. Global stores issue at ~14 B/clk per CU: an item's 32 KiB cost its compute waves ~4 us with the matrix pipe
  // idle (76 us of a 108 us launch without the epilogue). So where the layer has no residual input (conv1..conv4 of every dense block) a compute
  // wave only applies the activation, rounds to fp16 — the same values the register epilogue stores — and writes its eight rows into the
  // pixel stage its item has just finished with (row R of the tile = KiB R of the stage); two steps later loader w reads rows w, 8 + w, 16 + w,
  // 24 + w — exactly the KiBs its own next DMA pieces overwrite — and issues the stores beside the next item's MFMAs.
  auto xchg_write = [&](unsigned char* xs, int act) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v[8];
      v[0] = acc[i][0][0]; v[1] = acc[i][0][1]; v[2] = acc[i][0][2]; v[3] = acc[i][0][3];
      v[4] = acc[i][1][0]; v[5] = acc[i][1][1]; v[6] = acc[i][1][2]; v[7] = acc[i][1][3];
      if (act == ACT_LRELU) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], v[q] * 0.2f);
      } else if (act == ACT_SILU) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = apply_act(v[q], ACT_SILU);
      }
      union { uint4 u; _Float16 h[8]; } ov;
#pragma unroll
      for (int q = 0; q < 8; ++q) ov.h[q] = (_Float16)v[q];
      *reinterpret_cast<uint4*>(xs + ((8 * (wave & 3) + i) << 10) + lane * 16) = ov.u;
    }
  };
  auto xchg_store = [&](const unsigned char* xs, const Item& it) {
    const LPtr L = layers + it.layer;
    const int o_cs = L->out_cs;
    unsigned char* ob = reinterpret_cast<unsigned char*>(L->out) + ((long long)it.px0 * o_cs + L->out_coff) * 2;
    const unsigned long long u = reinterpret_cast<unsigned long long>(ob);
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(((unsigned long long)rfl((unsigned)(u >> 32)) << 32) | rfl((unsigned)u)), 0, 0x7FFFFFF0, 0x00020000);
    const int ox = it.x0 + pc, ch0 = it.nb * 32 + 8 * g;
    u32x4 d[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) d[w] = *reinterpret_cast<const u32x4*>(xs + ((8 * w + w8) << 10) + lane * 16);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int oy = it.y0 + 8 * w + w8;
      const bool ok = oy < it.H && ox < it.W;
      __builtin_amdgcn_raw_buffer_store_b128(d[w], rs_o, ok ? ((unsigned)(oy * it.W + ox) * o_cs + ch0) * 2 : OOB, 0, AUXC);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the rows are in registers before this wave's next DMA pieces land on them
  };
  auto publish = [&](int tile) {                   // every compute wave has drained its stores and the workgroup has met at a barrier since
    if ((FFP_TRUNK_SKIP & 64)) return;
    if (COH && tid == 256) __hip_atomic_fetch_add(a.done + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // ---- the walk: ONE body, instantiated per role — a compute wave's loop then carries no staging state and a loader's loop no accumulators (with
  // one loop for both roles hipcc spilled 65-190 registers at the 168 a 12-wave workgroup leaves each wave). Both instances take the same
  // decisions from the same words in LDS and their own copies of the wave-uniform cursors: they meet at the same barriers. ------------------------------
  auto walk = [&](auto role_tag) {
  constexpr bool LD = decltype(role_tag)::value;
  if (rfl(desc[0]) != D_READY) return;
  ck = 0;
  Item cur = load_item(0), nxt = cur;
  bool have_nxt = false;
  int c = 0;               // chunk of the current item = its step
  unsigned q = 0;          // step index in the stream: weight stage q & 1, pixel stage q % 3
  unsigned q0 = 0;         // stream index of the current item's chunk 0
  unsigned pix_next = 0;   // stream index of the next pixel chunk to request (pix_next > q: this step's chunk is in LDS or on its way)
  bool w_have = false;     // this step's weights have been requested (in the previous step)
  int pub_tile = -1;
  unsigned pub_after = 0;  // ... once the barrier of stream step pub_after has been passed
  bool pub_by_loader = false;
  int x_age = 0;           // a finished tile sits in a pixel stage for the loaders: 1 in the step after its item's last, 2 in the step the loaders store it
  unsigned x_q = 0;        // ... the stage of stream step x_q
  Item x_item = cur;
  auto pstage = [&](unsigned n) { return lds0 + (n % 3u) * G::PIX; };
  // which (item, chunk) is step n of the stream? 0: the current item, 1: the next one, -1: not known yet
  auto step_of = [&](unsigned n, int& cc) {
    const int o = (int)(n - q0);
    if (o < cur.NC) { cc = o; return 0; }
    if (have_nxt && o - cur.NC < nxt.NC) { cc = o - cur.NC; return 1; }
    cc = 0;
    return -1;
  };
#if FFP_TRUNK_DBG
  unsigned long long tsum[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();      // top + control, set-up, step (MFMAs / DMA issue), epilogue + next item, wait, barrier, synchronous staging
  unsigned n_iter = 0, n_slow = 0;
#define TSTAMP(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsum[k] += t_ - tprev; tprev = t_; }
#ifndef FFP_TRUNK_STAMP_WAVE
#define FFP_TRUNK_STAMP_WAVE 4       // which loader wave's sums are printed beside compute wave 0's (4 = loader 0, the control wave; 5..11 stage pixel pieces)
#endif
#define TDUMP() if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == FFP_TRUNK_STAMP_WAVE)) { unsigned* o = a.queue + 2 + (wave ? 1 : 0) * 7; for (int k_ = 0; k_ < 7; ++k_) o[k_] = (unsigned)(tsum[k_] / (n_iter ? n_iter : 1)); \
                  if (wave) { o[6] = n_iter | (n_slow << 16); } }
#else
#define TSTAMP(k)
#define TDUMP()
#endif
  for (unsigned itn = 0;; ++itn) {
    const int xa = x_age;                          // 1: the compute waves have just written a finished tile into stage x_q % 3 — hands off; 2: store it now
    if (x_age) x_age = x_age == 2 ? 0 : 2;
    if (LD && xa == 2) xchg_store(smem + (x_q % 3u) * G::PIX, x_item);
    // ---- operands of this step not requested yet (the walk's first step, after a late dependency, after a gap in the look-ahead): the loaders
    // request them now — and as much of the look-ahead as is known — and wait. Every wave takes the same branch (the cursors are wave-uniform).
    if (pix_next <= q || !w_have) {
      while (pix_next <= q + 2) {
        int cc;
        const int which = step_of(pix_next, cc);
        if (which < 0) break;
        if (xa == 1 && pix_next % 3u == x_q % 3u) break;       // that stage holds a finished tile the loaders have not read yet
        if constexpr (LD) {
          const int seq = ck + which;
          if (pix_item != seq) setup_pix(which ? nxt : cur, seq);
#pragma unroll
          for (int i = 0; i < 6; ++i) dma_pix(cc, pstage(pix_next), i);
        }
        ++pix_next;
      }
      if (!w_have) {
        if constexpr (LD) {
          setup_w(cur);
#pragma unroll
          for (int j = 0; j < 3; ++j) dma_w(c, lds0 + G::WOFF + (q & 1u) * G::WB, j, c == 0, lds0 + G::BIAS + (unsigned)(ck & 1) * 1024u);
        }
        w_have = true;
      }
      wait_all();
      __syncthreads();
      if (c == 0 && !LD) init_acc(smem + G::BIAS + (ck & 1) * 1024);
#if FFP_TRUNK_DBG
      ++n_slow;
#endif
      TSTAMP(6)
    }
    const bool last = c == cur.NC - 1;                           // the item's last step
    const bool nxt_known = rfl(snap[itn & 1u]) != 0;             // is the item after the ones held READY? (wave 0's snapshot of the PREVIOUS iteration: uniform)
    if (LD && wave == 4) control_issue();
    TSTAMP(0)
    // ---- what is requested while this step multiplies: the next step's weights (this item's next chunk, or the next item's first) and ONE
    // pixel chunk, up to two steps ahead
    int wc = c + 1, wwhich = 0;
    if (wc == cur.NC) { wc = 0; wwhich = have_nxt ? 1 : -1; }
    // (up to two when the look-ahead is behind: after a tile exchange kept a stage busy, after a late item)
    const bool use_x = last && cur.xchg && wwhich == 1;          // the finished tile goes to the loaders through this step's pixel stage
    int npix = 0;
    unsigned p_last = q;                                         // stream index of the last pixel chunk requested in this step
    if constexpr (LD) {
      if (wwhich == 1) setup_w(nxt);
      TSTAMP(1)
      const unsigned wst = lds0 + G::WOFF + ((q + 1u) & 1u) * G::WB, bst = lds0 + G::BIAS + (unsigned)((ck + 1) & 1) * 1024u;
      if (wwhich >= 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) dma_w(wc, wst, j, wwhich == 1, bst);
      }
    }
    for (int n_req = 0; n_req < 2 && pix_next <= q + 2; ++n_req) {
      int pcc;
      const int pwhich = step_of(pix_next, pcc);
      if (pwhich < 0) break;
      if (xa == 1 && pix_next % 3u == x_q % 3u) break;           // that stage holds a finished tile the loaders have not read yet
      if constexpr (LD) {
        if (pix_item != ck + pwhich) setup_pix(pwhich ? nxt : cur, ck + pwhich);
#pragma unroll
        for (int i = 0; i < 6; ++i) dma_pix(pcc, pstage(pix_next), i);
      }
      p_last = pix_next;
      ++pix_next;
    }
    if constexpr (LD) {
      // the pixel pieces of a chunk two steps ahead are the youngest requests: they stay in flight over this step's end (an older chunk requested
      // in the same step, and this step's stores, complete under the same wait)
      if (p_last == q + 2u) npix = pix_pieces;
      TSTAMP(2)
    } else {
      Epi ep = {};
      if (last && !use_x) ep = load_epi(cur);
      TSTAMP(1)
      chunk(smem + (q % 3u) * G::PIX, smem + G::WOFF + (q & 1u) * G::WB);
      TSTAMP(2)
      if (last && !use_x) {
        if (!(FFP_TRUNK_SKIP & 1)) {
          epilogue(cur, ep);
        } else {                                   // diagnostic build: keep the sums alive without the epilogue
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(acc[i][0]), "v"(acc[i][1]));
        }
      }
    }
    const bool w_next_have = wwhich >= 0;
    if (!have_nxt && nxt_known) {                  // the next item's descriptor (LDS reads + scalar loads: they overlap the wait below)
      nxt = load_item(ck + 1);
      have_nxt = true;
    }
    TSTAMP(3)
    // ---- the end of the step. Loaders: the next step's weights (requested first) must be in LDS, this step's pixel pieces stay in flight.
    // Compute waves: on every step but an item's last, everything they have asked for is back — the previous item's stores (what the deferred
    // publish below relies on) and wave 0's control requests; an item's last step leaves its four stores in flight.
    if constexpr (LD) {
      wait_keep(npix);
      if (wave == 4) control_consume();            // loader 0 stages no pixel pieces (npix = 0): everything it has asked for is back
    } else {
      if (!last) wait_all();
    }
    const bool switching = last && w_next_have;    // the stream runs on into the next item
    if (LD && tid == 256) {                        // is the item after the ones this workgroup will hold in the next iteration READY (and not held yet)?
      const bool will_hold_next = have_nxt && !switching;
      snap[(itn + 1u) & 1u] = !will_hold_next && get4(s_st, (switching ? ck + 2 : ck + 1) & 3) == D_READY ? 1 : 0;
    }
    TSTAMP(4)
    __syncthreads();
    TSTAMP(5)
#if FFP_TRUNK_DBG
    ++n_iter;
#endif
    // the previous item's tile becomes visible to its dependents once its stores have been drained: the compute waves' own stores by their full wait
    // of a non-final step, the loaders' (tile exchange) by the counted wait of the step in which they issued them
    if (pub_tile >= 0 && q >= pub_after && (pub_by_loader || !last)) {
      if (LD) publish(pub_tile);
      pub_tile = -1;
    }
    ++q;
    w_have = w_next_have;
    if (!last) { ++c; continue; }
    // ---- the item is finished
    if (LD && tid == 256) desc[(ck & 3) * 16] = D_EMPTY;  // its slot: sequence ck + 4 will be fetched into it
    if (LD && wave == 4) put4(s_st, ck & 3, D_EMPTY);
    q0 += (unsigned)cur.NC;
    c = 0;
    if (switching) {
      if (pub_tile >= 0) {                         // cannot happen (every item has at least two steps); never lose a tile's publication
        wait_all();
        __syncthreads();
        if (LD) publish(pub_tile);
      }
      pub_tile = cur.tile; pub_by_loader = use_x; pub_after = use_x ? q + 1u : q;       // q is already the next step's index
      if (use_x) {
        x_age = 1; x_q = q - 1u; x_item = cur;
        if (!LD) xchg_write(smem + (x_q % 3u) * G::PIX, layers[cur.layer].act);
      }
      cur = nxt; have_nxt = false; ++ck;
      if (!LD) init_acc(smem + G::BIAS + (ck & 1) * 1024);
      continue;
    }
    // the next item is not known or not ready (or there is none): finish this one for good, then wait for it
    if (LD && x_age == 2) { xchg_store(smem + (x_q % 3u) * G::PIX, x_item); }          // (a tile still in a stage: store it before the walk stops or pauses)
    x_age = 0;
    wait_all();
    __syncthreads();
    if (LD && pub_tile >= 0) publish(pub_tile);
    pub_tile = -1;
    if (LD) publish(cur.tile);
    if (LD && wave == 4) control_wait(ck + 1);
    __syncthreads();
    if (rfl(desc[((ck + 1) & 3) * 16]) != D_READY) { TDUMP() return; }
    ++ck;
    cur = load_item(ck);
    have_nxt = false;
    pix_next = q;                                  // nothing of the new item has been requested: the top of the loop does it and waits
    w_have = false;
    if (LD && tid == 256) { snap[0] = 0; snap[1] = 0; }
    __syncthreads();
  }
  };
  if (wave == 4) control_wait(0);
  __syncthreads();
  if (loader) walk(std::true_type{});
  else walk(std::false_type{});
}

}  // namespace

void conv_trunk_init() {
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS));
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_trunk_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TG::LDS));
}

bool conv_trunk_enabled() {
  // opt-in (FFP_TRUNK=1 or ffp_sr_set_fused_body): measured on MI355X it is bit-identical to, and 6 % slower than, the per-layer launches at 320 crops
  // and 2x slower at 32 (profiles/r04_sr_batch_sweep.txt) — the shipped default stays the faster path
  static const bool on = [] { const char* e = getenv("FFP_TRUNK"); return e && e[0] == '1'; }();
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
  FFP_CHECK(!ops.empty() && (int)ops.size() <= TG::MAXL, FFP_ERR_ARG, "trunk: %d layers (1..%d)", (int)ops.size(), TG::MAXL);
  lvl = ops[0].out.lvl;
  std::vector<TrunkLayer> h(ops.size());
  int cum = 0;
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
    L.cum = cum;
    cum += op.pc->cout / 32;
  }
  n_layers = (int)ops.size();
  n_blocks = cum;
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
  a.n_blocks = n_blocks;
  // [queue head, error word, padding to 64 B | done[tiles]]: zeroed before EVERY launch (a memset node under graph replay)
  const size_t need = 64 + sizeof(unsigned) * (size_t)n_tiles;
  const size_t nb = (need + 63) / 64 * 64;
  if (sync.n < nb) sync.alloc(nb);
  FFP_HIP(hipMemsetAsync(sync.p, 0, nb, st));
  a.queue = sync.as<unsigned>();
  a.done = sync.as<unsigned>() + 16;
  a.dbg = dbg;
  static const bool dump = [] { const char* e = getenv("FFP_TRUNK_DUMP"); return e && e[0] == '1'; }();
  // one workgroup per CU; capacity-mode levels launch for the capacity (workgroups beyond the batch's items find the queue empty)
  const long long items = (long long)n_tiles * n_blocks;
  const unsigned grid = (unsigned)std::min<long long>(256, items);
  if (n_layers > 1) hipLaunchKernelGGL(conv_trunk_kernel<true>, dim3(grid), dim3(768), G::LDS, st, a);
  else hipLaunchKernelGGL(conv_trunk_kernel<false>, dim3(grid), dim3(768), G::LDS, st, a);
  FFP_HIP(hipGetLastError());
  if (dump) {                                          // diagnostic: the queue head, the error word and the debug words behind them
    unsigned h[16];
    FFP_HIP(hipMemcpyAsync(h, sync.p, sizeof(h), hipMemcpyDeviceToHost, st));
    FFP_HIP(hipStreamSynchronize(st));
    fprintf(stderr, "trunk launch: tiles %d layers %d blocks %d grid %u | head %u err %u | cycles per step, compute wave 0 then loader wave 4: top + control, set-up, step, epilogue + next item, wait, barrier, synchronous staging; last word: steps | synchronous stagings << 16 |", n_tiles, n_layers, n_blocks, grid, h[0], h[1]);
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
