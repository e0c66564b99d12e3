// rrdb.cpp — Real-ESRGAN: RRDBNet as a plan of MFMA convolutions over a ragged batch of tiles.
//
// Network: basicsr RRDBNet(3, 3, 64, num_block, 32, scale) as the reference builds it at utils/enhancer.py:99-128
// (SURVEY.md Appendix D.1). Dense-block concatenations are virtual: each RDB works inside one [pixel][192] buffer
// (x | x1 | x2 | x3 | x4), conv5 writes the next RDB's x slot with the `*0.2 + x` residual(s) fused in its epilogue;
// nearest x2 upsampling is folded into the loader of conv_up1/conv_up2.
// Tiling: RealESRGANer.tile_process (Appendix D.2) — every padded tile of every image becomes one image of the ragged batch.
#include "sr_ops.hpp"

#include <cmath>

namespace ffp {

SrEngine::SrEngine(const void* weights, size_t nbytes, int scale, int num_block, int device, int half) {
  FFP_CHECK(scale == 4 || scale == 2, FFP_ERR_ARG, "scale must be 4 or 2");
  FFP_CHECK(num_block >= 1 && num_block <= 64, FFP_ERR_ARG, "num_block");
  int ndev = 0;
  FFP_CHECK(hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0, FFP_ERR_HIP, "no HIP device available (the enhancer has no CPU path)");
  FFP_CHECK(device >= 0 && device < ndev, FFP_ERR_ARG, "device %d of %d", device, ndev);
  scale_ = scale; num_block_ = num_block; device_ = device; dt_ = half ? F16 : F32;
  FFP_HIP(hipSetDevice(device_));
  st_ = create_engine_stream("FFP_SR_CU_MASK");
  for (auto& e : ev_) FFP_HIP(hipEventCreate(&e));
  conv_kernels_init();
  WeightFile wf;
  wf.parse(weights, nbytes);
  auto add = [&](const std::string& n) { pack_conv(convs_[n], wf, n, 3, 1, dt_, st_); };
  add("conv_first");
  for (int b = 0; b < num_block_; ++b)
    for (int r = 1; r <= 3; ++r)
      for (int c = 1; c <= 5; ++c) add("body." + std::to_string(b) + ".rdb" + std::to_string(r) + ".conv" + std::to_string(c));
  add("conv_body"); add("conv_up1"); add("conv_up2"); add("conv_hr"); add("conv_last");
  const int cin_first = 3 * (scale_ == 2 ? 4 : 1);
  FFP_CHECK(conv("conv_first")->cin_real == cin_first && conv("conv_first")->cout == 64, FFP_ERR_WEIGHTS,
            "conv_first is %d->%d, scale %d expects %d->64", conv("conv_first")->cin_real, conv("conv_first")->cout, scale_, cin_first);
  fused_body_ = dt_ == F16 && conv_trunk_enabled();
  for (const auto& kv : convs_) weight_bytes_ += kv.second.w.n + kv.second.w16.n + kv.second.bias.n + kv.second.w_direct.n;
}

void SrEngine::set_fused_body(bool on) {
  on = on && dt_ == F16;
  if (on == fused_body_) return;
  FFP_HIP(hipSetDevice(device_));
  if (pending_) wait_done();
  FFP_HIP(hipStreamSynchronize(st_));
  plans_.clear();
  fused_body_ = on;
}

void SrEngine::drop_plans() {
  FFP_HIP(hipSetDevice(device_));
  if (pending_) wait_done();
  FFP_HIP(hipStreamSynchronize(st_));
  plans_.clear();
}

size_t SrEngine::plan_bytes() const {
  size_t b = 0;
  for (const auto& kv : plans_) b += kv.second->bytes;
  return b;
}

SrEngine::~SrEngine() {
  (void)hipSetDevice(device_);
  if (st_) (void)hipStreamSynchronize(st_);
  plans_.clear();
  convs_.clear();
  for (auto& e : ev_) (void)hipEventDestroy(e);
  if (st_) (void)hipStreamDestroy(st_);
}

const PackedConv* SrEngine::conv(const std::string& name) const {
  auto it = convs_.find(name);
  FFP_CHECK(it != convs_.end(), FFP_ERR_WEIGHTS, "conv '%s' missing from the weight container", name.c_str());
  return &it->second;
}

// capacity buckets: 64 * {1, 1.5, 2, 3, 4, 6, ...} 16x16 tiles at the body level (at most 1.5x the batch)
static int capacity_bucket(long long t16) {
  long long b = 64;
  for (;;) {
    if (t16 <= b) return (int)b;
    if (t16 <= b + b / 2) return (int)(b + b / 2);
    b *= 2;
    FFP_CHECK(b < (1ll << 22), FFP_ERR_ARG, "enhance: batch of %lld tiles is too large", t16);
  }
}

void SrEngine::build_plan(SrPlan& P, int cap_t16) {
  // every image has at least one tile and a tile at most 256 pixels; x2 / x4 levels scale by 4 / 16
  P.cap_t16 = cap_t16;
  P.Lb = P.add_level_capacity(cap_t16, 256ll * cap_t16, cap_t16, st_);
  P.L1 = P.add_level_capacity(cap_t16, 1024ll * cap_t16, 4 * cap_t16, st_);
  P.L2 = P.add_level_capacity(cap_t16, 4096ll * cap_t16, 16 * cap_t16, st_);
  const DType T = dt_;
  const int cin = conv("conv_first")->cin;     // padded to the vector width
  P.input = P.alloc(P.Lb, cin, T);
  TView feat = P.alloc(P.Lb, 64, T);
  TView ring[4];
  for (auto& r : ring) r = P.alloc(P.Lb, 192, T);

  // the body's convs are collected and added as ONE fused step (conv_trunk.hip) when the engine runs the fused body
  std::vector<ConvOp> body;
  bool collect = false;
  auto cv = [&](const std::string& name, const TView& in, const TView& out, int act, int up = 0, const TView* r1 = nullptr, float s1 = 1.f,
                const TView* r2 = nullptr, float s2 = 1.f) {
    ConvOp o;
    o.pc = conv(name); o.in = in; o.out = out; o.stride = 1; o.act = act; o.up = up;
    if (r1) { o.has_res1 = true; o.res1 = *r1; o.s1 = s1; }
    if (r2) { o.has_res2 = true; o.res2 = *r2; o.s2 = s2; }
    FFP_CHECK(o.pc->cin == in.C && o.pc->cout == out.C, FFP_ERR_WEIGHTS, "%s: weights are %d->%d, graph expects %d->%d", name.c_str(),
              o.pc->cin, o.pc->cout, in.C, out.C);
    if (collect) {
      FFP_CHECK(conv_trunk_layer_ok(o), FFP_ERR_STATE, "%s: not a layer the fused body launch can run", name.c_str());
      body.push_back(o);
    } else {
      P.add_conv(o);
    }
  };

  cv("conv_first", P.input, feat, ACT_NONE);
  cv("conv_first", P.input, ring[0].slice(0, 64), ACT_NONE);   // second copy inside the first dense buffer (3->64: negligible)
  int cur = 0;
  collect = fused_body_;
  for (int b = 0; b < num_block_; ++b) {
    const TView rrdb_in = ring[cur].slice(0, 64);
    for (int r = 1; r <= 3; ++r) {
      const std::string p = "body." + std::to_string(b) + ".rdb" + std::to_string(r);
      const TView& buf = ring[cur];
      const TView x = buf.slice(0, 64);
      for (int c = 1; c <= 4; ++c)
        cv(p + ".conv" + std::to_string(c), buf.slice(0, 64 + 32 * (c - 1)), buf.slice(64 + 32 * (c - 1), 32), ACT_LRELU);
      const int nxt = (cur + 1) & 3;
      const TView y = ring[nxt].slice(0, 64);
      if (r < 3) cv(p + ".conv5", buf, y, ACT_NONE, 0, &x, 0.2f);                       // x5*0.2 + x
      else cv(p + ".conv5", buf, y, ACT_NONE, 0, &x, 0.2f, &rrdb_in, 0.2f);             // (x5*0.2 + x)*0.2 + rrdb_in
      cur = nxt;
    }
    // ring[cur] now holds the RRDB output; the RRDB input slot (3 steps back == (cur+1)&3) is free again
  }
  collect = false;
  if (!body.empty()) P.add_trunk(body);
  TView feat2 = P.alloc(P.Lb, 64, T);
  cv("conv_body", ring[cur].slice(0, 64), feat2, ACT_NONE, 0, &feat, 1.0f);            // feat + conv_body(body)
  TView u1 = P.alloc(P.L1, 64, T);
  cv("conv_up1", feat2, u1, ACT_LRELU, 1);
  TView u2 = P.alloc(P.L2, 64, T);
  cv("conv_up2", u1, u2, ACT_LRELU, 1);
  TView hr = P.alloc(P.L2, 64, T);
  cv("conv_hr", u2, hr, ACT_LRELU);
  P.out = P.alloc(P.L2, 4, F32);
  cv("conv_last", hr, P.out.slice(0, 3), ACT_NONE);
  const size_t n = (size_t)cap_t16;
  P.d_srcs.alloc(sizeof(SrSrc) * n);
  P.d_dsts.alloc(sizeof(SrDst) * n);
  P.d_core_off.alloc(sizeof(long long) * (n + 1));
  P.stage_bytes = (sizeof(SrSrc) + sizeof(SrDst) + sizeof(long long)) * (n + 1);
  FFP_HIP(hipHostMalloc(&P.stage, P.stage_bytes, hipHostMallocDefault));
}

SrPlan::~SrPlan() {
  if (stage) (void)hipHostFree(stage);
}

SrPlan& SrEngine::plan_for(long long t16) {
  // smallest resident plan that holds the batch; else a new one at the next capacity bucket (least recently used one evicted)
  SrPlan* best = nullptr;
  for (auto& kv : plans_)
    if (kv.first >= t16 && (!best || kv.first < best->cap_t16)) best = kv.second.get();
  if (!best) {
    const int cap = capacity_bucket(t16);
    if (plans_.size() >= 4) {
      auto victim = plans_.begin();
      for (auto it = plans_.begin(); it != plans_.end(); ++it)
        if (it->second->last_use < victim->second->last_use) victim = it;
      plans_.erase(victim);
    }
    std::unique_ptr<SrPlan> p(new SrPlan());
    build_plan(*p, cap);
    ++plans_built;
    best = p.get();
    plans_.emplace(cap, std::move(p));
  }
  best->last_use = ++use_clock_;
  return *best;
}

void SrEngine::wait_done() {
  FFP_HIP(hipSetDevice(device_));
  FFP_HIP(hipStreamSynchronize(st_));
  if (pending_) {
    if (prof.enabled) prof.collect();
    FFP_HIP(hipEventElapsedTime(&last_ms, ev_[0], ev_[1]));
    pending_ = false;
  }
}

void SrEngine::enhance_dev(const uint8_t* d_in, uint8_t* d_out, const std::vector<SrImage>& imgs, int tile, int tile_pad, int pre_pad, bool wait) {
  FFP_HIP(hipSetDevice(device_));
  if (pending_) wait_done();
  FFP_CHECK(!imgs.empty(), FFP_ERR_ARG, "enhance: empty batch");
  FFP_CHECK(tile_pad >= 0 && pre_pad >= 0, FFP_ERR_ARG, "enhance: negative padding");
  const int s = scale_;
  const int shuf = s == 2 ? 2 : 1;
  std::vector<TileDesc> tiles;
  for (const SrImage& im : imgs) {
    FFP_CHECK(im.h >= 1 && im.w >= 1, FFP_ERR_ARG, "enhance: empty image");
    // pre_pad (reflect, bottom/right), then mod pad to a multiple of 2 for the x2 model (RealESRGANer.pre_process)
    FFP_CHECK(pre_pad < im.h && pre_pad < im.w, FFP_ERR_ARG, "enhance: pre_pad %d >= image size (reflect padding undefined)", pre_pad);
    int H = im.h + pre_pad, W = im.w + pre_pad;
    int mph = 0, mpw = 0;
    if (s == 2) { if (H % 2) mph = 2 - H % 2; if (W % 2) mpw = 2 - W % 2; }
    FFP_CHECK((mph == 0 || H >= 2) && (mpw == 0 || W >= 2), FFP_ERR_ARG, "enhance: image too small for reflect padding");
    H += mph; W += mpw;
    const int out_h = im.h * s, out_w = im.w * s;     // pads are cropped off again (post_process)
    int tsz = tile > 0 ? tile : std::max(H, W);
    const int txn = (W + tsz - 1) / tsz, tyn = (H + tsz - 1) / tsz;
    for (int ty = 0; ty < tyn; ++ty)
      for (int tx = 0; tx < txn; ++tx) {
        const int sx = tx * tsz, ex = std::min(sx + tsz, W), sy = ty * tsz, ey = std::min(sy + tsz, H);
        int sxp = sx, exp_ = ex, syp = sy, eyp = ey;
        if (tile > 0) { sxp = std::max(sx - tile_pad, 0); exp_ = std::min(ex + tile_pad, W); syp = std::max(sy - tile_pad, 0); eyp = std::min(ey + tile_pad, H); }
        TileDesc t;
        t.h = eyp - syp; t.w = exp_ - sxp;
        FFP_CHECK(shuf == 1 || (t.h % 2 == 0 && t.w % 2 == 0 && sxp % 2 == 0 && syp % 2 == 0), FFP_ERR_ARG,
                  "enhance: x2 model needs even tile geometry (tile %d, pad %d)", tile, tile_pad);
        t.src.src_off = im.in_off; t.src.src_stride = im.in_stride; t.src.src_h = im.h; t.src.src_w = im.w; t.src.pre_h = im.h + pre_pad; t.src.pre_w = im.w + pre_pad;
        t.src.x0 = sxp; t.src.y0 = syp;
        // core region in output pixels, clipped to the un-padded output
        const int ox = sx * s, oy = sy * s;
        const int cw = std::min(ex * s, out_w) - ox, chh = std::min(ey * s, out_h) - oy;
        t.dst.dst_off = im.out_off; t.dst.dst_stride = im.out_stride;
        t.dst.ox = ox; t.dst.oy = oy; t.dst.cw = std::max(cw, 0); t.dst.ch = std::max(chh, 0);
        t.dst.tx = (sx - sxp) * s; t.dst.ty = (sy - syp) * s;
        tiles.push_back(t);
      }
  }
  const int n = (int)tiles.size();
  std::vector<int> hs(n), ws(n), h1(n), w1(n), h2(n), w2(n);
  long long t16 = 0;
  for (int i = 0; i < n; ++i) {
    hs[i] = tiles[i].h / shuf; ws[i] = tiles[i].w / shuf;
    h1[i] = hs[i] * 2; w1[i] = ws[i] * 2; h2[i] = hs[i] * 4; w2[i] = ws[i] * 4;
    t16 += (long long)((hs[i] + 15) / 16) * ((ws[i] + 15) / 16);
  }
  // The plan depends on a CAPACITY, not on the sizes: a new multiset of crop sizes (every frame of a real stream) costs table
  // uploads only — no allocation, no tuning, no graph capture (the reference enhances arbitrary crops back to back,
  // utils/enhancer.py:344-391).
  SrPlan& P = plan_for(t16);
  P.Lb->assign(hs, ws, st_);
  P.L1->assign(h1, w1, st_);
  P.L2->assign(h2, w2, st_);
  const int cap = P.cap_t16;
  SrSrc* srcs = reinterpret_cast<SrSrc*>(P.stage);
  SrDst* dsts = reinterpret_cast<SrDst*>(srcs + cap);
  long long* coff = reinterpret_cast<long long*>(dsts + cap);
  long long tot = 0;
  for (int i = 0; i < n; ++i) { srcs[i] = tiles[i].src; dsts[i] = tiles[i].dst; coff[i] = tot; tot += (long long)dsts[i].cw * dsts[i].ch; }
  for (int i = n; i < cap; ++i) { dsts[i] = SrDst{}; coff[i] = tot; }       // padding entries: empty cores behind the last real one
  coff[cap] = tot;
  FFP_HIP(hipMemcpyAsync(P.d_srcs.p, srcs, sizeof(SrSrc) * n, hipMemcpyHostToDevice, st_));
  FFP_HIP(hipMemcpyAsync(P.d_dsts.p, dsts, sizeof(SrDst) * cap, hipMemcpyHostToDevice, st_));
  FFP_HIP(hipMemcpyAsync(P.d_core_off.p, coff, sizeof(long long) * (cap + 1), hipMemcpyHostToDevice, st_));
  FFP_HIP(hipEventRecord(ev_[0], st_));
  launch_sr_pre(d_in, P.d_srcs.as<SrSrc>(), P.input, shuf, st_);
  if (prof.enabled) prof.begin();
  P.execute(st_, &prof);
  launch_sr_post(P.out, P.d_dsts.as<SrDst>(), P.d_core_off.as<long long>(), P.L2->total_px, d_out, st_);
  FFP_HIP(hipEventRecord(ev_[1], st_));
  last_conv_flops = P.actual_conv_flops();
  last_graph = P.gexec != nullptr;
  last_conv_launches = P.conv_launches;
  pending_ = true;
  if (wait) wait_done();
}

}  // namespace ffp
