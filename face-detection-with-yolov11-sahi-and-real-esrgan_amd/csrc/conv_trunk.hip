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
#define FFP_TRUNK_DBG 0        // 1: diagnostic build (phase-skip bits in TrunkArgs::dbg: 1 epilogue, 2 MFMA, 4 DMA, 16 fragment reads)
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
// descriptor slot (16 ints in LDS): 0 state, 1 queue id, 2 layer, 3 tile index, 4 image, 5 y0, 6 x0, 7 tiles_x | tiles_y << 16

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
  int layer, tile, img, y0, x0;
  int NC, NT;            // 32-channel input chunks, 32-channel output blocks (1 or 2)
  int H, W;              // image size
  long long px0;         // first pixel of the image in the level
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

  // ---- per-lane constants of the staging: pixel pieces wave + 8 i (i < 5; piece < 39), lane -> halo pixel q = piece * 16 + lane / 4 and
  // LDS slot position lane % 4, which holds SOURCE slot (lane % 4) ^ ((hx >> 1) & 2): the fragment reads below are then conflict free
  unsigned hyx[5];         // (source slot << 16) | (hy << 8) | hx, or 0xFFFFFFFF for a dummy pixel / a piece this wave does not have
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int p = wave + 8 * i;
    const int q = p * 16 + (lane >> 2);
    const int hy = q / G::HC, hx = q - hy * G::HC;
    const int s = (lane & 3) ^ ((hx >> 1) & 2);
    hyx[i] = (p < G::PIX_PIECES && q < G::HR * G::HC) ? (unsigned)((s << 16) | (hy << 8) | hx) : OOB;
  }
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
  int fk = 0, fstate = 0;  // fetcher: next sequence number to fetch; 0 idle, 1 queue id pending, 2 tile entry pending, 3 queue exhausted
  int f_id = 0;            // pending atomic result (lane 0)
  u32x4 f_tile = {0u, 0u, 0u, 0u};
  int f_layer = 0, f_t = 0;
  unsigned f_qid = 0;
  int p_base = -1;         // polls in flight: sequence number of lane group 0 (-1: none)
  unsigned p_val = 0;
  // Every memory operation of a control step is issued UNCONDITIONALLY — through buffer descriptors, with an out-of-range offset where the
  // step has nothing to ask (the range check drops the request) — and consumed one step later: behind an exec-masked or conditional
  // load hipcc waits for vmcnt(0) at once, which would park wave 0 for a memory round trip (an atomic: 1-2 us) in front of its MFMAs.
  const auto rs_sync = __builtin_amdgcn_make_buffer_rsrc(a.queue, 0, 64 + 4 * n_tiles, 0x00020000);          // [queue | done[]]
  const auto rs_tiles = __builtin_amdgcn_make_buffer_rsrc(const_cast<int4*>(a.tiles), 0, 16 * n_tiles, 0x00020000);
  // "nothing to ask" offset of the control step's dword / atomic operations: aligned and beyond any num_records (0xFFFFFFFF + 4 wraps in
  // a 32-bit range check; a buffer ATOMIC at that offset faulted with a memory aperture violation on gfx950, 16-byte loads do not)
  constexpr unsigned OOBA = 0x80000000u;
  auto control_step = [&]() {
    // 1. consume the fetch issued one step ago
    unsigned tile_off = OOBA;
    if (fstate == 1) {
      const unsigned id = (unsigned)rfl(f_id);
      if (id >= total) {
        if (lane == 0) desc[(fk & 3) * 16] = D_END;
        fstate = 3;
      } else {
        f_qid = id;
        f_layer = (int)(id / (unsigned)n_tiles);
        f_t = (int)(id - (unsigned)f_layer * (unsigned)n_tiles);
        tile_off = (unsigned)f_t * 16u;
        fstate = 2;
      }
    } else if (fstate == 2) {
      const int img = rfl((int)f_tile[0]), y0 = rfl((int)f_tile[1]), x0 = rfl((int)f_tile[2]), txy = rfl((int)f_tile[3]);
      if (lane == 0) {
        int* d = desc + (fk & 3) * 16;
        d[1] = (int)f_qid; d[2] = f_layer; d[3] = f_t; d[4] = img; d[5] = y0; d[6] = x0; d[7] = txy;
        d[0] = (COH && f_layer > 0) ? D_KNOWN : D_READY;       // the first layer of a launch depends on earlier launches only
      }
      ++fk;
      fstate = 0;
    }
    // 2. consume the polls issued one step ago: group gi (lanes 16 gi .. 16 gi + 8) -> sequence p_base + gi
    if (COH && p_base >= 0) {
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int sq = p_base + gi;
        int* d = desc + (sq & 3) * 16;
        if (sq > ck && sq < fk && d[0] == D_KNOWN) {
          const int need = d[2];
          const bool mine = (lane >> 4) == gi && (lane & 15) < 9;
          const bool ok = !mine || (int)p_val >= need;
          if (__builtin_amdgcn_ballot_w64(ok) == ~0ull && lane == 0) d[0] = D_READY;
        }
      }
      p_base = -1;
    }
    // 3. the next fetch step: ids are taken at most three items ahead of the current one
    bool take = false;
    if (fstate == 0 && fk <= ck + 3) { take = true; fstate = 1; }
    f_id = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, rs_sync, (take && lane == 0) ? 0u : OOBA, 0, 0);
    f_tile = __builtin_amdgcn_raw_buffer_load_b128(rs_tiles, tile_off, 0, 0);
    // 4. poll the counters of every fetched, not yet ready item: tile t and its neighbours t + dy * tiles_x + dx inside the image
    if (COH) {
      bool any = false;
      unsigned off = OOBA;
#pragma unroll
      for (int gi = 0; gi < 3; ++gi) {
        const int sq = ck + 1 + gi;
        const int* d = desc + (sq & 3) * 16;
        if (sq < fk && d[0] == D_KNOWN) {
          any = true;
          const int j = lane & 15, dy = j / 3 - 1, dx = j - (j / 3) * 3 - 1;
          const int t = d[3], tx = d[6] >> 4, ty = d[5] / G::TH, nx = d[7] & 0xFFFF, ny = d[7] >> 16;
          const bool in = (unsigned)(tx + dx) < (unsigned)nx && (unsigned)(ty + dy) < (unsigned)ny;
          if ((lane >> 4) == gi && j < 9) off = 64u + 4u * (unsigned)(in ? t + dy * nx + dx : t);
        }
      }
      p_val = __builtin_amdgcn_raw_buffer_load_b32(rs_sync, off, 0, 16);      // sc1: the counters are written by other workgroups
      if (any) p_base = ck + 1;
    }
  };
  // blocking form (prologue, late dependencies): until the item after `ck_prev` is READY or the queue has ended. Bounded: a spin that
  // never ends would hang the device — it gives up, raises the error word and ends this workgroup's walk instead
  auto control_wait = [&](int sq) {
    int* d = desc + (sq & 3) * 16;
    for (int spin = 0;; ++spin) {
      const int s = d[0];
      if (s == D_READY || s == D_END) break;
      if (spin > (1 << 22)) {
        if (lane == 0) { atomicAdd(a.queue + 1, 1u); d[0] = D_END; }
        break;
      }
      control_step();
      if (spin > 4) __builtin_amdgcn_s_sleep(4);
    }
  };

  if (tid < 80) desc[tid] = 0;
  __syncthreads();

  // ---- item set-up: descriptor slot -> SGPRs, per-lane source offsets of the halo pixels ------------------------------------------------
  auto load_item = [&](int sq) {
    const int* d = desc + (sq & 3) * 16;
    Item it;
    it.layer = rfl(d[2]); it.tile = rfl(d[3]); it.img = rfl(d[4]); it.y0 = rfl(d[5]); it.x0 = rfl(d[6]);
    const int4 im = a.img_tab[it.img];
    it.px0 = rfl(im.x); it.H = rfl(im.y); it.W = rfl(im.z);
    it.NC = layers[it.layer].cin >> 5;
    it.NT = layers[it.layer].cout >> 5;
    return it;
  };
  u32x4 rs_in, rs_w;
  auto setup_dma = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    const int cs = L->in_cs;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(L->in) + (it.px0 * cs + L->in_coff) * 2;
    rs_in = make_rsrc(inb, 0x7FFFFFF0u);
    rs_w = make_rsrc(L->wpk, (unsigned)(it.NT * it.NC * G::WB));
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hy = (int)(hyx[i] >> 8) & 0xFF, hx = (int)hyx[i] & 0xFF, sl = (int)(hyx[i] >> 16) & 3;
      const int iy = it.y0 - 1 + hy, ix = it.x0 - 1 + hx;
      const bool ok = hyx[i] != OOB && (unsigned)iy < (unsigned)it.H && (unsigned)ix < (unsigned)it.W;
      isrc[i] = ok ? (unsigned)((iy * it.W + ix) * cs * 2 + sl * 16) : OOB;
    }
  };
  // chunk c of the item -> stage at LDS byte address `st`: <= 5 pixel pieces + <= 3 (5) weight pieces per wave; the chunk's 64 bytes
  // per pixel ride in the instruction's scalar offset (not part of the range check: OOB lanes stay out of range)
  auto issue_dma = [&](const Item& it, int c, unsigned st) {
    if (FFP_TRUNK_DBG && (a.dbg & 4)) return;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (wave + 8 * i < G::PIX_PIECES) dma16<COH>(rs_in, isrc[i], (unsigned)(c * 64), st + (unsigned)((wave + 8 * i) << 10));
    const int nwp = it.NT * 18;
    for (int q = wave; q < nwp; q += 8) {
      const int nt = q >= 18 ? 1 : 0, pq = q - nt * 18;
      dma16<false>(rs_w, (unsigned)lane * 16u, (unsigned)(((nt * it.NC + c) * 18 + pq) << 10), st + (unsigned)(G::PIX + (q << 10)));
    }
  };

  // ---- arithmetic: conv_rows16_kernel's chunk, for NM = 2 NT M-tiles -----------------------------------------------------------------------
  f32x4 acc[4][4];
  auto init_acc = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    const float* b = L->bias;
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
  auto chunk = [&](const unsigned char* sb, auto nm_tag) {
    constexpr int NM = decltype(nm_tag)::value;
    uint4 bq[2][6], aq[3][NM];
    auto ldB = [&](int kx, int q) {
#pragma unroll
      for (int j = 0; j < 6; ++j) bq[q][j] = *reinterpret_cast<const uint4*>(sb + boff[kx] + j * G::ROWB);
    };
    auto ldA = [&](int s, int q) {
      const int tap = (s % 3) * 3 + s / 3;
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) aq[q][mm] = *reinterpret_cast<const uint4*>(sb + aoff + (((mm >> 1) * 18 + tap * 2 + (mm & 1)) << 10));
    };
    ldB(0, 0);
    ldA(0, 0);
    ldA(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int kx = s / 3, ky = s - 3 * kx;
      if (s + 2 < 9) ldA(s + 2, (s + 2) % 3);
      if (ky == 0 && kx < 2) ldB(kx + 1, (kx + 1) & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mm = 0; mm < NM; ++mm) {
          if (FFP_TRUNK_DBG && (a.dbg & 2)) continue;
          union { uint4 u; f16x8 h; } ua, ub;
          ua.u = aq[s % 3][mm]; ub.u = bq[kx & 1][i + ky];
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
  auto epilogue = [&](const Item& it) {
    const LPtr L = layers + it.layer;
    const int act = L->act;
    const float s1 = L->s1, s2 = L->s2;
    const int o_cs = L->out_cs, r1_cs = L->r1_cs, r2_cs = L->r2_cs;
    const bool has1 = L->res1 != nullptr, has2 = L->res2 != nullptr;
    unsigned char* ob = reinterpret_cast<unsigned char*>(L->out) + (it.px0 * o_cs + L->out_coff) * 2;
    const unsigned char* r1b = has1 ? reinterpret_cast<const unsigned char*>(L->res1) + (it.px0 * r1_cs + L->r1_coff) * 2 : ob;
    const unsigned char* r2b = has2 ? reinterpret_cast<const unsigned char*>(L->res2) + (it.px0 * r2_cs + L->r2_coff) * 2 : ob;
    auto mk = [](const unsigned char* q) {
      const unsigned long long u = reinterpret_cast<unsigned long long>(q);
      const unsigned lo = rfl((unsigned)u), hi = rfl((unsigned)(u >> 32));
      return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<unsigned char*>(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFF0, 0x00020000);
    };
    const auto rs_o = mk(ob), rs_r1 = mk(r1b), rs_r2 = mk(r2b);
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
    if (FFP_TRUNK_DBG && (a.dbg & 64)) return;
    if (COH && tid == 0) __hip_atomic_store(a.done + tile, (unsigned)(layer + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto wait_all = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  // ---- prologue: the first item --------------------------------------------------------------------------------------------------------------
  if (wave == 0) control_wait(0);
  __syncthreads();
  if (rfl(desc[0]) != D_READY) return;
  ck = 0;
  Item cur = load_item(0);
  if (FFP_TRUNK_DBG && (a.dbg & 512)) {              // diagnostic: what the first item of workgroup 0 looks like, then every workgroup ends
    if (blockIdx.x == 0 && tid == 0) {
      a.queue[2] = (unsigned)cur.layer; a.queue[3] = (unsigned)cur.tile; a.queue[4] = (unsigned)cur.img; a.queue[5] = (unsigned)cur.y0; a.queue[6] = (unsigned)cur.x0;
      a.queue[7] = (unsigned)cur.NC; a.queue[8] = (unsigned)cur.NT; a.queue[9] = (unsigned)cur.H; a.queue[10] = (unsigned)cur.W; a.queue[11] = (unsigned)cur.px0;
      a.queue[12] = (unsigned)n_tiles; a.queue[13] = total; a.queue[14] = (unsigned)desc[7];
    }
    return;
  }
  setup_dma(cur);
  issue_dma(cur, 0, lds0);
  init_acc(cur);
  if (FFP_TRUNK_DBG && (a.dbg & 1024)) return;
  wait_all();
  __syncthreads();
  int c = 0;
  unsigned stage = 0;
  int pub_tile = -1, pub_layer = 0;
  // One iteration = one chunk. Whether the NEXT item can be prefetched is decided from a snapshot that wave 0 wrote during the PREVIOUS
  // iteration (snap[parity]): every wave of the workgroup takes the same branch, whatever wave 0's control step is doing meanwhile.
  for (unsigned itn = 0;; ++itn) {
    const bool last = c == cur.NC - 1;
    const bool pre = last && rfl(snap[itn & 1u]) != 0;
    if (wave == 0) {
      control_step();
      const int nseq = pre ? ck + 2 : ck + 1;                    // the item that will be "next" in the following iteration
      if (lane == 0) snap[(itn + 1u) & 1u] = desc[(nseq & 3) * 16] == D_READY ? 1 : 0;
    }
    Item nxt = cur;
    if (!last) {
      issue_dma(cur, c + 1, lds0 + (stage ^ 1u) * G::STAGE);
    } else if (pre) {
      nxt = load_item(ck + 1);
      setup_dma(nxt);
      issue_dma(nxt, 0, lds0 + (stage ^ 1u) * G::STAGE);
    }
    const unsigned char* sb = smem + stage * G::STAGE;
    if (cur.NT == 2) chunk(sb, std::integral_constant<int, 4>{});
    else chunk(sb, std::integral_constant<int, 2>{});
    if (last) {
      if (!(FFP_TRUNK_DBG && (a.dbg & 1))) epilogue(cur);
      // this wave's DMA pieces are older than the epilogue's loads and stores: all but the 4 NT stores must have completed
      if (cur.NT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      wait_all();
    }
    __syncthreads();
    if (pub_tile >= 0) {                           // the previous item: its stores were drained by this (non-final) chunk's full wait
      publish(pub_tile, pub_layer);
      pub_tile = -1;
    }
    if (!last) { ++c; stage ^= 1u; continue; }
    if (tid == 0) desc[(ck & 3) * 16] = D_EMPTY;   // the finished item's slot: sequence ck + 4 will be fetched into it
    if (pre) {
      pub_tile = cur.tile; pub_layer = cur.layer;
      cur = nxt; ++ck; c = 0; stage ^= 1u;
      init_acc(cur);
      continue;
    }
    // the next item is not ready (or there is none): finish this one for good, then wait for it
    wait_all();
    __syncthreads();
    publish(cur.tile, cur.layer);
    if (wave == 0) control_wait(ck + 1);
    __syncthreads();
    if (rfl(desc[((ck + 1) & 3) * 16]) != D_READY) return;
    ++ck;
    cur = load_item(ck);
    setup_dma(cur);
    stage ^= 1u;
    issue_dma(cur, 0, lds0 + stage * G::STAGE);
    init_acc(cur);
    if (tid == 0) { snap[0] = 0; snap[1] = 0; }
    wait_all();
    __syncthreads();
    c = 0;
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
  a.tiles = lvl->tile_table(G::TH, &n_tiles, &a.n_tiles_dev, st);
  if (n_tiles == 0) return;
  a.ntiles_host = n_tiles;
  a.layers = d_layers.as<TrunkLayer>();
  a.n_layers = n_layers;
  a.img_tab = lvl->d_tab.as<int4>();
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
    fprintf(stderr, "trunk launch: tiles %d layers %d grid %u | head %u err %u |", n_tiles, n_layers, grid, h[0], h[1]);
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
