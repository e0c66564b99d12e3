// conv_rows16pc.hip — conv_rows16_kernel with the work of a workgroup split by ROLE: four producer waves own the global -> LDS
// staging and the epilogues, four consumer waves only read fragments and issue MFMAs. One 8-wave workgroup per CU.
//
// Why (round-3 probes of conv_rows16_kernel, profiles/r03_rows16_phase_probe.txt, r03_rows16_in_kernel_stamps.txt): with everything in
// one instruction stream per wave, MFMA-only takes 23 us, staging-only 26 us and the full kernel 44 us — the two do not overlap; a chunk
// of 1,152 MFMA cycles costs a wave ~215 cycles of exposed first fragment reads, ~2,000 of MFMA stream (its own staging waits and the
// partner wave's MFMAs inside), 480-690 of epilogue + item set-up and ~150 of barrier. Neither more prefetch distance, nor fewer bytes
// (weights resident: 0.76-0.89x), nor priorities moved it: one in-order wave cannot keep the matrix pipe fed beside its own staging,
// epilogue and set-up. Here every SIMD hosts ONE consumer wave, whose stream is fragment reads + 72 MFMAs per chunk and nothing else,
// and ONE producer wave with a load counter of its own, whose VALU / LDS / memory work fills the issue slots the MFMAs leave:
//   * producers: the staging of conv_rows16_kernel unchanged (21 input + 18 weight wave-pieces per chunk over four waves, raw buffer
//     loads, out-of-image lanes read zeros, two register sets = three chunks of prefetch distance), written into the other LDS stage
//     while the consumers multiply the current one;
//   * consumers: the MFMA stream of conv_rows16_kernel unchanged (same fragments, same order: results are bit-identical); at the end
//     of an item a consumer wave writes its 32 accumulator registers to an 8 KiB exchange block in LDS (8 ds_write_b128) and starts
//     the next item from the bias;
//   * the producer wave paired with it picks the block up after the chunk's barrier and runs the register epilogue (activation,
//     residuals, fp16 conversion, 16-byte stores) while the consumers are already multiplying the next item;
//   * one s_barrier per chunk for all eight waves, as before: the roles advance in lockstep, chunk time = the consumers' 1,152 MFMA
//     cycles + first fragment reads + barrier as long as the producers keep up (they have ~300 issue cycles of work per chunk).
// LDS: two 39 KiB stages + 32 KiB of exchange blocks + descriptors = 112 KiB.
// MEASURED (round 3, profiles/r03_rows16_role_split_probe.txt): bit-identical, and 0.73-0.94x the speed of conv_rows16_kernel — the
// hypothesis above is wrong, or incomplete: with the LDS bytes per MFMA unchanged (0.5 KiB of fragment reads + 0.135 KiB of staging
// writes) a CU with ONE tile in flight is slower than a CU with two whatever the wave roles are. What remains untested is the other
// half of the idea: consumers with 8 rows each (0.33 KiB of reads per MFMA) fed by LDS-DMA producers (no ds_write_b128 on the shared
// VGPR -> LDS path). Kept behind force_shape 24 / FFP_ROWS16_PC=1 as the starting point for that.
#include <algorithm>
#include <cstdlib>

#include "conv_args.hpp"

namespace ffp {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PCGeo {
  static constexpr int IN_PX = 336;
  static constexpr int IN_BYTES = IN_PX * 64;                 // 21504
  static constexpr int W_BYTES = 9 * 2 * 1024;                // 18432
  static constexpr int STAGE = IN_BYTES + W_BYTES;            // 39936
  static constexpr int NP = 10;
  static constexpr int TCAP = 40;
  static constexpr int DUMP = 2 * STAGE;                      // four exchange blocks of 8 KiB: [consumer wave][row i][M-tile m][lane] x 16 B
  static constexpr int DUMP_W = 8192;
  static constexpr int DESC = DUMP + 4 * DUMP_W;
  static constexpr int LDS = DESC + TCAP * 48;                // 114560
};

template <int NS>              // staging register sets of a producer wave = chunks in flight ahead of the one being written (even)
__global__ void __launch_bounds__(512, 2) conv_rows16pc_kernel(const ConvArgs a) {
  using G = PCGeo;
  constexpr int NP = G::NP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave8 >= 4;                    // waves w and w + 4 share a SIMD (dispatch order 0 -> 2 -> 1 -> 3, twice)
  const int wave = wave8 & 3;
  const int pc = lane & 15, g = lane >> 4;

  // ---- this workgroup's items (conv_rows16_kernel's dealing) -------------------------------------------------------------------------
  const int n_items = (a.n_tiles_dev ? __builtin_amdgcn_readfirstlane(*a.n_tiles_dev) : a.ntiles_host) * a.n_nblk;
  const int Ws = (int)gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  int per_xcd = (n_items + 7) >> 3;
  per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  const int first = xcd * per_xcd + slot;
  const int last = min(n_items, (xcd + 1) * per_xcd);
  const int J = first < last ? min((last - first + Ws - 1) / Ws, G::TCAP) : 0;
  if (J == 0) return;                                  // whole workgroup: no barrier is skipped by part of it
  const int nt0 = first % a.n_nblk;

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
  const int NC = a.cin >> 5;
  const int Q = J * NC;                                // chunks of this workgroup: both roles run exactly Q iterations, one barrier each
  const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.wpk) + (long long)nt0 * NC * G::W_BYTES;
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(wb), 0, 0x7FFFFFF0, 0x00020000);
  auto bload = [](decltype(rs_w) rs, unsigned off, int soff = 0) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
  unsigned char* dump = smem + G::DUMP + wave * G::DUMP_W + lane * 16;       // this wave pair's exchange block

  if (producer) {
    // ================================================= producer: staging + epilogues ==================================================
    auto rs_in = rs_w;
    auto rs_wp = rs_w;
    const bool s9_in = wave == 0;
    const int s9_piece = wave == 0 ? 20 : wave == 1 ? 16 : 17;
    unsigned isrc[6], idst[6], wrel[5], hyx[6];
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

    int pf_item = 0, pf_c = 0;
    auto setup_pf = [&](int j) {
      const int4 t = desc[j * 3], it = desc[j * 3 + 1];
      const int oy0 = sgpr(t.y), ox0 = sgpr(t.z), Hi = sgpr(it.y), Wi = sgpr(it.z);
      const int Hv = Hi << a.up, Wv = Wi << a.up;
      const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.in) + ((long long)sgpr(it.x) * a.in_cs + a.in_coff) * 2;
      rs_in = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(inb), 0, 0x7FFFFFF0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int hy = (int)(hyx[i] >> 8) & 0xFF, hx = (int)hyx[i] & 0xFF, sl = (int)(hyx[i] >> 16) & 3;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = (hyx[i] & 0xFFFFu) != 0xFFFFu && (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
        isrc[i] = ok ? (unsigned)(((iy >> a.up) * Wi + (ix >> a.up)) * a.in_cs * 2 + sl * 16) : OOB;
      }
    };
    uint4 sets[NS][NP];
    auto piece_fetch = [&](int p, uint4& r) {
      if (p < 5) {
        r = bload(rs_in, isrc[p] != OOB ? isrc[p] + (unsigned)(pf_c * 64) : OOB);
      } else if (p < 9) {
        r = bload(rs_wp, wrel[p - 5], pf_c * G::W_BYTES);
      } else {
        const unsigned oi = isrc[5] != OOB ? isrc[5] + (unsigned)(pf_c * 64) : OOB;
        r = bload(s9_in ? rs_in : rs_wp, s9_in ? oi : wrel[4], s9_in ? 0 : pf_c * G::W_BYTES);
      }
    };
    auto piece_stash = [&](unsigned char* sb, int p, const uint4& r) {
      if (p < 5) *reinterpret_cast<uint4*>(sb + idst[p]) = r;
      else if (p < 9) *reinterpret_cast<uint4*>(sb + G::IN_BYTES + wrel[p - 5]) = r;
      else *reinterpret_cast<uint4*>(sb + (s9_in ? idst[5] : G::IN_BYTES + wrel[4])) = r;
    };
    auto advance_pf = [&]() {
      if (pf_item >= J) return;
      if (++pf_c == NC) {
        pf_c = 0;
        if (++pf_item < J) {
          setup_pf(pf_item);
        } else {
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
    // the chunk in `set` goes to stage sb; each register is re-requested for the chunk two further on right after its write
    auto restage = [&](unsigned char* sb, uint4 (&set)[NP]) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        piece_stash(sb, p, set[p]);
        piece_fetch(p, set[p]);
      }
      advance_pf();
    };

    // ---- epilogue of item j from the exchange block: lane (pc, g) holds channels 8g..8g+7 of pixel (row 4*wave + i, column pc) ------------
    auto epilogue = [&](int j) {
      f32x4 acc[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[i][m] = *reinterpret_cast<const f32x4*>(dump + ((i * 2 + m) << 10));
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
        if (a.act == ACT_LRELU) {
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
        __builtin_amdgcn_raw_buffer_store_b128(ov.u, rs_o, ok ? (rel_px * a.out_cs + ch0) * 2 : OOB, 0, 0);
      }
    };

    // ---- prologue: chunks 0 .. NS-1 requested (set c holds chunk c mod NS), chunk 0 into stage 0, its set re-requested for chunk NS -------
    setup_pf(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) fetch_all(sets[k]);
#pragma unroll
    for (int p = 0; p < NP; ++p) piece_stash(smem, p, sets[0][p]);
    fetch_all(sets[0]);
    __syncthreads();
    int cc = 0, cj = 0;                                // the chunk the consumers multiply in this iteration
    for (int q0 = 0; q0 < Q; q0 += NS) {
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (q0 + k >= Q) break;
        // iteration q0 + k: stage k & 1 is being multiplied; set (k + 1) % NS holds chunk q0 + k + 1 -> the other stage; the item that
        // ended at the last barrier is stored
        restage(smem + ((k + 1) & 1) * G::STAGE, sets[(k + 1) % NS]);
        if (cc == 0 && cj > 0) epilogue(cj - 1);
        if (++cc == NC) { cc = 0; ++cj; }
        __syncthreads();
      }
    }
    epilogue(J - 1);                                   // the last item's sums were written before the last barrier
    return;
  }

  // ===================================================== consumer: fragment reads + MFMAs =====================================================
  const float4 bias0 = *reinterpret_cast<const float4*>(a.bias + nt0 * 32 + 8 * g);
  const float4 bias1 = *reinterpret_cast<const float4*>(a.bias + nt0 * 32 + 8 * g + 4);
  unsigned boff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = pc + kx;
    boff[kx] = (unsigned)(((4 * wave) * 18 + hx) * 64 + ((g ^ ((hx >> 1) & 2)) << 4));
  }
  const unsigned aoff = G::IN_BYTES + lane * 16;
  f32x4 acc[4][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i][0] = f32x4{bias0.x, bias0.y, bias0.z, bias0.w};
      acc[i][1] = f32x4{bias1.x, bias1.y, bias1.z, bias1.w};
    }
  };
  zero_acc();
  // one step = one tap: 2 weight fragments x 4 output rows = 8 MFMAs; weight fragments two steps, input-row fragments one kx ahead
  auto chunk = [&](const unsigned char* sb) {
    uint4 bq[2][6], aq[3][2];
    auto ldB = [&](int kx, int q) {
#pragma unroll
      for (int j = 0; j < 6; ++j) bq[q][j] = *reinterpret_cast<const uint4*>(sb + boff[kx] + j * 1152);
    };
    auto ldA = [&](int s, int q) {
      const int tap = (s % 3) * 3 + s / 3;
#pragma unroll
      for (int m = 0; m < 2; ++m) aq[q][m] = *reinterpret_cast<const uint4*>(sb + aoff + ((tap * 2 + m) << 10));
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
        for (int m = 0; m < 2; ++m) {
          union { uint4 u; f16x8 h; } ua, ub;
          ua.u = aq[s % 3][m]; ub.u = bq[kx & 1][i + ky];
          acc[i][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ua.h, ub.h, acc[i][m], 0, 0, 0);
        }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x7F6, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto hand_over = [&]() {                             // the item's sums to the paired producer wave, the next item starts from the bias
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m) *reinterpret_cast<f32x4*>(dump + ((i * 2 + m) << 10)) = acc[i][m];
    zero_acc();
  };
  __syncthreads();                                     // the producers' prologue: chunk 0 sits in stage 0
  int cc = 0;
  for (int q = 0; q < Q; q += 2) {
    chunk(smem);
    if (++cc == NC) { cc = 0; hand_over(); }
    __syncthreads();
    if (q + 1 >= Q) break;
    chunk(smem + G::STAGE);
    if (++cc == NC) { cc = 0; hand_over(); }
    __syncthreads();
  }
}

}  // namespace

void conv_rows16pc_init() {
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows16pc_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PCGeo::LDS));
  FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows16pc_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, PCGeo::LDS));
}

// force_shape 24, or every eligible layer with FFP_ROWS16_PC=1 (A/B aid). NOT the default: measured 0.73-0.94x of conv_rows16_kernel
// (profiles/r03_rows16_role_split_probe.txt) — with the same LDS traffic per MFMA the role split only halves the tiles in flight per CU.
bool conv_rows16pc_selected(const ConvArgs& a) {
  static const bool env_on = [] { const char* e = getenv("FFP_ROWS16_PC"); return e && e[0] == '1'; }();
  return a.force_shape == 24 || (a.force_shape < 0 && env_on && a.dbg == 0);
}

void launch_conv_rows16pc(ConvArgs& a, const PackedConv& pc, Level* out_lvl, hipStream_t st) {
  using G = PCGeo;
  a.wpk = pc.w16.p;
  int n_tiles = 0;
  a.tiles = out_lvl->tile_table(16, &n_tiles, &a.n_tiles_dev, st);
  if (n_tiles == 0) return;
  a.ntiles_host = n_tiles;
  a.n_nblk = a.ntiles32;
  const long long items = (long long)n_tiles * a.n_nblk;
  long long per_xcd = (items + 7) / 8;
  per_xcd = (per_xcd + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  long long ws = std::max<long long>(32, (per_xcd + G::TCAP - 1) / G::TCAP);          // 8 x 32 workgroups: one per CU
  ws = (ws + a.n_nblk - 1) / a.n_nblk * a.n_nblk;
  FFP_CHECK(8 * ws < (1ll << 31) && (per_xcd + ws - 1) / ws <= G::TCAP, FFP_ERR_STATE, "rows16pc: launch geometry");
  static const int ns = [] { const char* e = getenv("FFP_ROWS16_PC_SETS"); return e ? atoi(e) : 2; }();       // 4 sets (five chunks ahead) measured SLOWER than 2: 0.68-0.85x vs 0.78-0.95x of conv_rows16_kernel
  if (ns == 2) hipLaunchKernelGGL(conv_rows16pc_kernel<2>, dim3((unsigned)(8 * ws)), dim3(512), G::LDS, st, a);
  else hipLaunchKernelGGL(conv_rows16pc_kernel<4>, dim3((unsigned)(8 * ws)), dim3(512), G::LDS, st, a);
}

}  // namespace ffp
