// conv_pw.hip — 1x1 convolution in the fp32-grade split arithmetic (FFP_PREC_F32X3), the detector's HBM-bound layers.
//
// A 1x1 conv is a GEMM in which every activation is needed by exactly ONE pixel row of the product. The generic kernel
// (conv_mfma.hip) nevertheless stages the activations through LDS like a 3x3 halo tile: global -> registers -> hi/lo split
// -> LDS -> fragments, one barrier per 16-channel chunk, and with 128 accumulator registers only ONE chunk in flight, so
// every chunk exposes a memory round trip (measured 2.2-3.5 TB/s on layers that move 0.9 KB and multiply 18 MFMA cycles per
// pixel). Here:
//   * the MFMA operands are swapped: A = pixel fragment, B = packed weight fragment (same bytes as the generic kernel's A
//     operand: the two layouts are mirror images), so a lane of the accumulator holds ONE out channel x 16 pixels;
//   * a lane's share of the A operand — 8 consecutive input channels of its pixel — is 32 contiguous bytes of the NHWC
//     tensor: raw buffer loads put it straight into registers, the hi/lo split happens there, and the activations never
//     touch LDS. The four loads that consume one 128-byte line of a pixel are issued back to back;
//   * a wave owns MI x 32 pixels for all of the workgroup's NT x 32 out channels and runs alone: no barrier after the
//     weights are in. A ring of RP line-loads per pixel (RP x 128 B x MI x 32 px per wave, 8 waves per CU) stays in
//     flight, across work items too — the next pixel block's first lines are requested before this block's epilogue;
//   * the weights of the workgroup's channel block are LDS-resident for the life of the (persistent) workgroup: a block
//     walks a CONTIGUOUS run of pixel blocks with the same weights (contiguous: a wave then meets every image once, and the
//     per-image max-|value| slots cost one atomic per image and wave instead of one per block);
//   * epilogue: lanes 0..31 of a store are 32 consecutive channels of one pixel — whole 128-byte lines from dword
//     stores, no LDS transposition; pixel / channel tails are cut by the buffer range check, not by branches.
#include <algorithm>

#include "conv_args.hpp"

#ifndef FFP_PW_STAMP
#define FFP_PW_STAMP 0         // 1: diagnostic build — s_memtime stamps per stage of the streamed-weight form, printed by a few workgroups
#endif

namespace ffp {

namespace {

constexpr unsigned PW_OOB = 0x80000000u;      // beyond any num_records we set (<= 0x7FFFFFF0), also after adding small offsets
// 2 KiB weight fragments (hi + lo) a workgroup keeps resident: half a CU's LDS for the 4-wave shapes (two workgroups per CU), all
// of it for the 8-wave shapes (one workgroup per CU, twice the input channels for the same register budget per wave)
constexpr int pw_lds_frags(int nw) { return nw == 8 ? 80 : 40; }

// ST (deep inputs whose weights do not fit the LDS): one EXTRA wave per workgroup streams the weights — RP line-groups per stage, two
// stages — while the NW consumer waves multiply out of the other stage; one barrier per RP line-groups. The producer has its own
// in-order load counter: a consumer that waited for weights it had loaded itself would also wait for every activation load issued
// before them (measured: that variant lost 1.5x), and the consumers' activation ring stays as deep as in the resident form.
template <int NW, int MI, int NT, int RP, bool UP, bool ST = false>
__global__ void __launch_bounds__((NW + (ST ? 1 : 0)) * 64, 2) conv_pw_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BPX = NW * 32 * MI;            // pixels per work item: NW waves x MI fragments of 32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, hh = lane >> 5;

  const int G = (int)gridDim.x;
  const int L = xcd_remap(blockIdx.x, G);       // the channel blocks of a pixel block and neighbouring pixel blocks share an XCD
  const int nblk = L % a.n_nblk, slot = L / a.n_nblk, nslots = G / a.n_nblk;
  const int ntile0 = nblk * NT;
  const int ncg = a.ncg, npairs = ncg >> 1;
  const int PB = (int)((a.total_px + BPX - 1) / BPX);
  const int pb_q = PB / nslots, pb_r = PB % nslots;                      // this slot's run of pixel blocks: [pb_begin, pb_end)
  const int pb_begin = slot * pb_q + min(slot, pb_r), pb_end = pb_begin + pb_q + (slot < pb_r ? 1 : 0);
  auto after = [&](int pb, int k) { return pb + k < pb_end ? pb + k : PB; };      // k blocks on, or PB = nothing (empty span)

  auto sgpr = [](unsigned long long u) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
  };
  // (base, bytes) of a resource over the pixel records of block pb and everything after it (up to 2 GiB): record stride cs, view
  // offset coff. Past the last block: empty (loads return zeros without memory traffic, stores are dropped)
  struct Span { unsigned long long base; unsigned num; };
  auto block_span = [&](const void* base, int pb, int cs, int coff) {
    const long long px0 = (long long)pb * BPX;
    const long long rem = pb < PB ? (a.total_px - px0) * cs * 4 - (long long)coff * 4 : 0;
    Span s;
    s.num = (unsigned)__builtin_amdgcn_readfirstlane((int)(rem > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : rem));
    s.base = sgpr(reinterpret_cast<unsigned long long>(base) + (unsigned long long)((pb < PB ? px0 : 0) * cs + coff) * 4ull);
    return s;
  };
  auto rsrc = [](const Span& s) { return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(s.base), 0, s.num, 0x00020000); };
  auto bl = [](decltype(rsrc(Span{})) rs, unsigned voff, int soff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
  };

  // ---- weights of this channel block -> LDS, [nt][k-group] x {hi 1 KiB, lo 1 KiB}: one contiguous run of the packed array
  constexpr int CKG = 2 * RP;                  // ST: k-groups per stage; a stage is [nt][CKG] x 2 KiB
  constexpr int STAGE = NT * CKG * 2048;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wpk) + (size_t)ntile0 * ncg * 2048;
  if constexpr (ST) {
    if (wave == NW) {
      // ---- the producer wave: chunk 0 before the first barrier, then chunk i + 1 into the idle stage during chunk i, in 1 KiB pieces
      // (64 lanes x 16 B) with 8 loads in flight; it takes part in every barrier of the consumers' schedule and does nothing else
      const int nchunks = npairs / RP;
      constexpr int PIECES = NT * CKG * 2;
      // LDS-DMA (global_load_lds_dwordx4: global -> LDS without a register in between): the whole stage is requested at once, 64 KiB in
      // flight instead of the 8 KiB a register batch allowed — the single producer wave was latency-bound at ~8 KiB per memory round trip
      // and the seven consumer waves waited for it (deep layers: 27-30 % of the MFMA peak). The __syncthreads() behind every call waits
      // for the loads (hipcc drains vmcnt at a barrier with LDS-DMA in flight) — exactly what the hand-over of the stage needs.
      auto copy_chunk = [&](int c, unsigned char* stage) {
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
          const int nt = pc / (2 * CKG), off = (pc % (2 * CKG)) * 1024;
          const unsigned char* g = wsrc + ((size_t)nt * ncg + (size_t)c * CKG) * 2048 + off + lane * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                           (__attribute__((address_space(3))) void*)(stage + nt * (CKG * 2048) + off), 16, 0, 0);
        }
      };
      // the hand-over of a stage: the producer's LDS-DMA must have LANDED before the barrier releases the consumers. Stated here, not left
      // to the barrier's lowering (a workgroup-scope release does not require vmcnt(0) for global operations)
      copy_chunk(0, smem);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int it = 0;
      for (int pbp = pb_begin; pbp < pb_end; ++pbp)
        for (int c = 0; c < nchunks; ++c, ++it) {
          copy_chunk(c + 1 < nchunks ? c + 1 : 0, smem + ((it + 1) & 1) * STAGE);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      return;
    }
  } else {
    const uint4* src = reinterpret_cast<const uint4*>(wsrc);
    const int nv = NT * ncg * 128;
    for (int i = tid; i < nv; i += NW * 64) reinterpret_cast<uint4*>(smem)[i] = src[i];
  }

  // activation scale of the split: per buffer, or (per-image exponent slots) per 32-pixel fragment = per image, looked up for every
  // pixel block this wave walks (in the block loop below)
  float tscale = 1.f, tinv = 1.f;
  if (a.amax_in && !a.amax_img) split_scales((unsigned)__builtin_amdgcn_readfirstlane((int)amax_in_bits(a, 0)), &tscale, &tinv);
  float bias_n[NT], osc_n[NT];
  unsigned cbyte[NT];                          // byte offset of this lane's channel inside a pixel record (OOB: padded channel)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (ntile0 + nt) * 32 + p;
    bias_n[nt] = a.bias[c];
    osc_n[nt] = a.oscale ? a.oscale[c] : 1.f;
    cbyte[nt] = c < a.cout ? (unsigned)c * 4u : PW_OOB;
  }

  unsigned voff[MI];                           // this lane's pixel inside the block (item independent), + its k half
  unsigned vout[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    voff[mi] = (unsigned)((wave * MI + mi) * 32 + p) * (unsigned)(a.in_cs * 4) + hh * 32;
    vout[mi] = (unsigned)((wave * MI + mi) * 32 + 4 * hh) * (unsigned)(a.out_cs * 4);
  }
  const unsigned ocs4 = (unsigned)a.out_cs * 4u;

  // ring slot = one 128-byte line per pixel = k-groups 2*pr and 2*pr + 1: this lane's 2 x 32 bytes of it
  uint4 raw[RP][MI][4];
  auto issue = [&](uint4 (&q)[MI][4], const Span& sp, const unsigned (&vo)[MI], int pr) {
    const auto rs = rsrc(sp);
    const int so = pr * 128;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      q[mi][0] = bl(rs, vo[mi], so);
      q[mi][1] = bl(rs, vo[mi] + 16, so);
      q[mi][2] = bl(rs, vo[mi] + 64, so);
      q[mi][3] = bl(rs, vo[mi] + 80, so);
    }
  };
  // virtual concat [nearest_x2(coarser view) | rest] (ConvOp::has_up2): input channels [0, up_c) of a pixel live in the coarser
  // view at pixel up_map[px]. A line-load is wholly on one side (up_c is a multiple of 64); the map entries of a pixel block are
  // fetched one block ahead, so nothing waits for them except the very first load of the kernel.
  Span sp_up{0ull, 0u};
  unsigned vup_cur[MI], vup_next[MI];
  auto load_vup = [&](int pbx, unsigned (&v)[MI]) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long long px = (long long)pbx * BPX + (wave * MI + mi) * 32 + p;
      v[mi] = (pbx < PB && px < a.total_px) ? (unsigned)a.up_map[px] * (unsigned)(a.up_cs * 4) + hh * 32 : PW_OOB;
    }
  };
  // line-load `pr` of a block whose own span / map offsets are (sp_in, vu): pick the side
  auto fetch = [&](uint4 (&q)[MI][4], const Span& sp_in, const unsigned (&vu)[MI], int pr) {
    if constexpr (UP) {
      const bool up = pr * 32 < a.up_c;
      Span sp;
      sp.base = up ? sp_up.base : sp_in.base;
      sp.num = up ? sp_up.num : sp_in.num;
      unsigned vo[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) vo[mi] = up ? vu[mi] : voff[mi];
      issue(q, sp, vo, pr);
    } else {
      issue(q, sp_in, voff, pr);
    }
  };
  auto split8 = [&](const uint4& r0, const uint4& r1, float ts, f16x8& hi, f16x8& lo) {
    union { uint4 u; f16x8 h; } H, Lo;
    split_pair(__uint_as_float(r0.x), __uint_as_float(r0.y), ts, H.u.x, Lo.u.x);
    split_pair(__uint_as_float(r0.z), __uint_as_float(r0.w), ts, H.u.y, Lo.u.y);
    split_pair(__uint_as_float(r1.x), __uint_as_float(r1.y), ts, H.u.z, Lo.u.z);
    split_pair(__uint_as_float(r1.z), __uint_as_float(r1.w), ts, H.u.w, Lo.u.w);
    hi = H.h;
    lo = Lo.h;
  };

  int pb = pb_begin;
  Span rs_cur = block_span(a.in, pb, a.in_cs, a.in_coff);
  if constexpr (UP) {
    sp_up.base = sgpr(reinterpret_cast<unsigned long long>(a.up_src));
    sp_up.num = 0x7FFFFFF0u;                   // the launcher checked that the coarser tensor is smaller than that
    load_vup(pb, vup_cur);
  }
#pragma unroll
  for (int r = 0; r < RP; ++r) fetch(raw[r], rs_cur, vup_cur, r);
  __syncthreads();                             // weights visible; from here on the waves run on their own

  // per-image exponent slots: {image, end of its real pixels} of this wave's fragments and the input's max-|value| bits, fetched through
  // the scalar cache TWO blocks (table) / ONE block (slot) ahead, so that the chain table -> slot never sits on the critical path
  const int nfrag = (int)(a.total_px >> 5);
  auto frag_of = [&](int pbx, int mi) { return 2 * min((int)(((long long)pbx * BPX) >> 5) + wave * MI + mi, nfrag - 1); };
  int im_n[MI], re_n[MI], im_nn[MI], re_nn[MI];
  unsigned am_n[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    im_n[mi] = im_nn[mi] = 0; re_n[mi] = re_nn[mi] = 0; am_n[mi] = 0u;
    if (a.amax_img) {
      im_n[mi] = sload(a.frag_img, frag_of(pb, mi));
      re_n[mi] = sload(a.frag_img, frag_of(pb, mi) + 1);
      im_nn[mi] = sload(a.frag_img, frag_of(after(pb, 1), mi));
      re_nn[mi] = sload(a.frag_img, frag_of(after(pb, 1), mi) + 1);
      am_n[mi] = amax_in_bits_s(a, im_n[mi]);
    }
  }
  const unsigned char* wl0 = smem + lane * 16;
  uint4 wq[2][2];                              // weight fragment ring {hi, lo} x 2 steps
  wq[0][0] = *reinterpret_cast<const uint4*>(wl0);
  wq[0][1] = *reinterpret_cast<const uint4*>(wl0 + 1024);
  int it = 0;                                  // ST: chunks done so far; the current stage is it & 1
#if FFP_PW_STAMP
  unsigned long long stp_stream = 0, stp_barrier = 0, stp_epi = 0, stp_n = 0, stp_blk = 0, stp_nb = 0;
  const unsigned long long stp_k0 = __builtin_amdgcn_s_memtime();
#endif
  int run_img = -1;                            // the image this wave is collecting max |value| for, and the per-lane maximum so far
  float run_max = 0.f;
  for (; pb < pb_end; ++pb) {
#if FFP_PW_STAMP
    const unsigned long long stp_b0 = __builtin_amdgcn_s_memtime();
#endif
    // this block's fragments: scale, inverse scale, image (slot index) and the end of the image's real pixels
    float ts_f[MI], ti_f[MI];
    int im_f[MI];
    long long rend_f[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ts_f[mi] = tscale; ti_f[mi] = tinv; im_f[mi] = 0; rend_f[mi] = a.total_px;
      if (a.amax_img) {
        im_f[mi] = im_n[mi];
        rend_f[mi] = re_n[mi];
        if (a.amax_in) split_scales(am_n[mi], &ts_f[mi], &ti_f[mi]);
        im_n[mi] = im_nn[mi];
        re_n[mi] = re_nn[mi];
        am_n[mi] = amax_in_bits_s(a, im_n[mi]);
        im_nn[mi] = sload(a.frag_img, frag_of(after(pb, 2), mi));
        re_nn[mi] = sload(a.frag_img, frag_of(after(pb, 2), mi) + 1);
      }
    }
    const Span rs_next = block_span(a.in, after(pb, 1), a.in_cs, a.in_coff);
    if constexpr (UP) load_vup(after(pb, 1), vup_next);
    f32x16 acc[MI][NT];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][nt][r] = 0.f;

    for (int kb = 0; kb < npairs; kb += RP) {
#if FFP_PW_STAMP
      const unsigned long long stp_t0 = __builtin_amdgcn_s_memtime();
#endif
      const bool more = kb + RP < npairs;
      Span rs_f;
      rs_f.base = more ? rs_cur.base : rs_next.base;
      rs_f.num = more ? rs_cur.num : rs_next.num;
      const int prb = more ? kb + RP : 0;
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        // 2 * NT steps (k-group q = j / NT, channel tile nt = j % NT) of 3 * MI MFMAs; the weight fragments of step j + 1 are
        // requested from LDS before the MFMAs of step j issue (two-deep register ring carried over slots, blocks and items)
        const int pr_nx = r + 1 < RP ? kb + r + 1 : prb;              // line-group whose first fragments the last step requests
        const unsigned char* wbase = ST ? wl0 + (it & 1) * STAGE : wl0;
        const unsigned char* wp = wbase + 2 * (ST ? r : kb + r) * 2048;
        const unsigned char* wp_nx = ST ? wp + 4096 : wl0 + 2 * pr_nx * 2048;      // ST: never read for the last line-group of a stage
        const int wstride = ST ? CKG * 2048 : ncg * 2048;
        f16x8 ph[MI], pl[MI];
#pragma unroll
        for (int j = 0; j < 2 * NT; ++j) {
          const int q = j / NT, nt = j % NT;
          if (nt == 0) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) split8(raw[r][mi][2 * q], raw[r][mi][2 * q + 1], ts_f[mi], ph[mi], pl[mi]);
            if (q == 1) {                                            // both halves of the line are in hi/lo form: refill the slot HERE
              __builtin_amdgcn_sched_barrier(0);                     // (left alone the scheduler sinks every refill to the end of the loop
              if constexpr (UP) {                                    // body, where the next iteration waits for the first one at once)
                unsigned vu[MI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) vu[mi] = more ? vup_cur[mi] : vup_next[mi];
                fetch(raw[r], rs_f, vu, prb + r);
              } else {
                fetch(raw[r], rs_f, vup_cur, prb + r);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          const unsigned char* nx = j + 1 < 2 * NT ? wp + ((j + 1) / NT) * 2048 + ((j + 1) % NT) * wstride : wp_nx;
          if (!(ST && r == RP - 1 && j + 1 == 2 * NT)) {             // (the next stage is complete only after the barrier below)
            wq[(j + 1) & 1][0] = *reinterpret_cast<const uint4*>(nx);
            wq[(j + 1) & 1][1] = *reinterpret_cast<const uint4*>(nx + 1024);
          }
          union { uint4 u; f16x8 h; } wh, wl;
          wh.u = wq[j & 1][0];
          wl.u = wq[j & 1][1];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[mi], wl.h, acc[mi][nt], 0, 0, 0);     // small terms first
            acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl[mi], wh.h, acc[mi][nt], 0, 0, 0);
            acc[mi][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph[mi], wh.h, acc[mi][nt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         // the two reads for step j + 1, then this step's MFMAs
          __builtin_amdgcn_sched_group_barrier(0x008, 3 * MI, 0);
        }
      }
#if FFP_PW_STAMP
      asm volatile("s_nop 0" ::: "memory");
      const unsigned long long stp_t1 = __builtin_amdgcn_s_memtime();
#endif
      if constexpr (ST) {                                            // the producer has filled the other stage; everyone is done with this one
        __syncthreads();
#if FFP_PW_STAMP
        const unsigned long long stp_t2 = __builtin_amdgcn_s_memtime();
        stp_stream += stp_t1 - stp_t0; stp_barrier += stp_t2 - stp_t1; ++stp_n;
#endif
        ++it;
        wq[0][0] = *reinterpret_cast<const uint4*>(wl0 + (it & 1) * STAGE);
        wq[0][1] = *reinterpret_cast<const uint4*>(wl0 + (it & 1) * STAGE + 1024);
      }
    }

#if FFP_PW_STAMP
    const unsigned long long stp_e0 = __builtin_amdgcn_s_memtime();
#endif
    // ---- epilogue: register i of a tile = pixel row 8*(i>>2) + 4*hh + (i&3), this lane's channel
    const auto rs_out = rsrc(block_span(a.out, pb, a.out_cs, a.out_coff));
    const bool silu = a.act == ACT_SILU;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float am = 0.f;
      const long long px0 = (long long)pb * BPX + (wave * MI + mi) * 32 + 4 * hh;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned vo = vout[mi] + cbyte[nt];
        const float osc = osc_n[nt] * ti_f[mi];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = fmaf(acc[mi][nt][i], osc, bias_n[nt]);
          if (silu) v = v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
          if (px0 + 8 * (i >> 2) + (i & 3) < rend_f[mi] && cbyte[nt] != PW_OOB) am = fmaxf(am, fabsf(v));      // padding pixels / channels do not count
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, vo + (unsigned)(8 * (i >> 2) + (i & 3)) * ocs4, 0, 0);
        }
      }
      // the wave's fragments come in pixel order, so an image is one run of them: its slot is raised when the run ends
      if (im_f[mi] != run_img) {
        if (run_img >= 0 && a.amax_out) raise_amax(a.amax_out + run_img, run_max);
        run_img = im_f[mi];
        run_max = 0.f;
      }
      run_max = fmaxf(run_max, am);
    }
#if FFP_PW_STAMP
    {
      const unsigned long long stp_e1 = __builtin_amdgcn_s_memtime();
      stp_epi += stp_e1 - stp_e0; stp_blk += stp_e1 - stp_b0; ++stp_nb;
    }
#endif
    rs_cur = rs_next;
    if constexpr (UP) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) vup_cur[mi] = vup_next[mi];
    }
  }
  if (a.amax_out && run_img >= 0) raise_amax(a.amax_out + run_img, run_max);
#if FFP_PW_STAMP
  if (ST && lane == 0 && (blockIdx.x % 61) == 0 && stp_n)
    printf("pwstamp wg %d wave %d stages %llu : stream %llu barrier %llu per stage (MFMA %d) | blocks %llu: %llu per block, epilogue %llu | kernel %llu\n", (int)blockIdx.x, wave, stp_n,
           stp_stream / stp_n, stp_barrier / stp_n, RP * 2 * NT * 3 * MI * 32, stp_nb, stp_blk / (stp_nb ? stp_nb : 1), stp_epi / (stp_nb ? stp_nb : 1), __builtin_amdgcn_s_memtime() - stp_k0);
#endif
}

template <int NW, int MI, int NT, int RP> struct PwCfg {
  static void init() {
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<NW, MI, NT, RP, false>), hipFuncAttributeMaxDynamicSharedMemorySize, pw_lds_frags(NW) * 2048));
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<NW, MI, NT, RP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pw_lds_frags(NW) * 2048));
  }
  static void launch(ConvArgs& a, hipStream_t st) {
    constexpr int BPX = NW * 32 * MI;
    const int pbn = (int)((a.total_px + BPX - 1) / BPX);
    a.n_nblk = a.ntiles32 / NT;
    const int nslots = std::min(pbn, std::max(1, (NW == 8 ? 256 : 512) / a.n_nblk));      // every workgroup resident at once
    if (nslots == 0) return;
    const int lds = NT * a.ncg * 2048;
    if (a.up_c > 0) hipLaunchKernelGGL((conv_pw_kernel<NW, MI, NT, RP, true>), dim3(nslots * a.n_nblk), dim3(NW * 64), lds, st, a);
    else hipLaunchKernelGGL((conv_pw_kernel<NW, MI, NT, RP, false>), dim3(nslots * a.n_nblk), dim3(NW * 64), lds, st, a);
  }
};

template <int NW, int MI, int NT, int RPMAX> struct PwShape {
  static bool fits(const ConvArgs& a) { return a.ntiles32 % NT == 0 && NT * a.ncg <= pw_lds_frags(NW); }
  static int ring(const ConvArgs& a) {
    const int np = a.ncg / 2;
    for (int r = RPMAX; r > 1; --r) if (np % r == 0) return r;
    return 1;
  }
  static void init() {
    PwCfg<NW, MI, NT, 1>::init(); PwCfg<NW, MI, NT, 2>::init();
    if constexpr (RPMAX >= 3) PwCfg<NW, MI, NT, 3>::init();
    if constexpr (RPMAX >= 4) PwCfg<NW, MI, NT, 4>::init();
  }
  static void launch(ConvArgs& a, hipStream_t st) {
    const int r = ring(a);
    if constexpr (RPMAX >= 4) { if (r == 4) { PwCfg<NW, MI, NT, 4>::launch(a, st); return; } }
    if constexpr (RPMAX >= 3) { if (r == 3) { PwCfg<NW, MI, NT, 3>::launch(a, st); return; } }
    if (r == 2) PwCfg<NW, MI, NT, 2>::launch(a, st);
    else PwCfg<NW, MI, NT, 1>::launch(a, st);
  }
};

// streamed weights: 7 consumer waves + the weight producer. 16: 32 px x 128 channels per wave, any K that is a multiple of 128;
// 22: 64 px x 64 channels per wave (half the LDS fragment reads per MFMA: the deep 16^2 / 32^2 layers), any K that is a multiple of 64
template <int MI_, int NT_, int RP_> struct PwSt {
  static constexpr int NWC = 7, MI = MI_, RP = RP_, NT = NT_, LDS = 2 * NT * 2 * RP * 2048;
  static bool fits(const ConvArgs& a) { return a.ntiles32 % NT == 0 && (a.ncg / 2) % RP == 0; }
  static void init() {
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<NWC, MI, NT, RP, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    FFP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pw_kernel<NWC, MI, NT, RP, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  }
  static void launch(ConvArgs& a, hipStream_t st) {
    constexpr int BPX = NWC * 32 * MI;
    const int pbn = (int)((a.total_px + BPX - 1) / BPX);
    a.n_nblk = a.ntiles32 / NT;
    const int nslots = std::min(pbn, std::max(1, 256 / a.n_nblk));
    if (nslots == 0) return;
    if (a.up_c > 0) hipLaunchKernelGGL((conv_pw_kernel<NWC, MI, NT, RP, true, true>), dim3(nslots * a.n_nblk), dim3((NWC + 1) * 64), LDS, st, a);
    else hipLaunchKernelGGL((conv_pw_kernel<NWC, MI, NT, RP, false, true>), dim3(nslots * a.n_nblk), dim3((NWC + 1) * 64), LDS, st, a);
  }
};
using Pw14s = PwSt<1, 4, 4>;
using Pw22s = PwSt<2, 2, 2>;

using Pw14 = PwShape<4, 1, 4, 4>;      // force_shape 10: 128 px x 128 ch per workgroup, K <= 160
using Pw22 = PwShape<4, 2, 2, 2>;      // 11: 256 px x 64 ch, K <= 320
using Pw21 = PwShape<4, 2, 1, 4>;      // 12: 256 px x 32 ch, K <= 640
using Pw14w = PwShape<8, 1, 4, 4>;     // 13..15: the same per-wave tiles in 8-wave workgroups, twice the K
using Pw22w = PwShape<8, 2, 2, 2>;
using Pw21w = PwShape<8, 2, 1, 4>;

}  // namespace

void conv_pw_init() { Pw14::init(); Pw22::init(); Pw21::init(); Pw14w::init(); Pw22w::init(); Pw21w::init(); Pw14s::init(); Pw22s::init(); }

// what the kernel does not do: residual inputs, fp16 outputs, input channel counts that are not whole 128-byte lines
static bool pw_common(const ConvOp& op, const ConvArgs& a) {
  const PackedConv& pc = *op.pc;
  if (op.has_up2 && (long long)op.up2.lvl->total_px * op.up2.cs * 4 >= 0x7FFFFFF0ll) return false;      // one buffer resource spans the coarser tensor
  return pc.split && pc.k == 1 && op.stride == 1 && !op.has_res1 && !op.has_res2 && a.out_f32 && pc.cin % 32 == 0 &&
         pc.cin_pad == pc.cin && !op.out.lvl->capacity() && a.total_px > 0;
}

unsigned conv_pw_mask(const ConvOp& op, const ConvArgs& a) {
  if (!pw_common(op, a)) return 0;
  unsigned m = 0;
  if (Pw14::fits(a)) m |= 1u << 10;
  if (Pw22::fits(a)) m |= 1u << 11;
  if (Pw21::fits(a)) m |= 1u << 12;
  if (Pw14w::fits(a)) m |= 1u << 13;
  if (Pw22w::fits(a)) m |= 1u << 14;
  if (Pw21w::fits(a)) m |= 1u << 15;
  if (Pw14s::fits(a)) m |= 1u << 16;
  if (Pw22s::fits(a)) m |= 1u << 22;
  return m;
}

void launch_conv_pw(ConvArgs& a, int shape, hipStream_t st) {
  switch (shape) {
    case 10: Pw14::launch(a, st); break;
    case 11: Pw22::launch(a, st); break;
    case 12: Pw21::launch(a, st); break;
    case 13: Pw14w::launch(a, st); break;
    case 14: Pw22w::launch(a, st); break;
    case 15: Pw21w::launch(a, st); break;
    case 22: Pw22s::launch(a, st); break;
    default: Pw14s::launch(a, st); break;
  }
}

}  // namespace ffp
