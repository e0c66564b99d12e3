// engine.cpp — plan bookkeeping and conv-kernel profiling.
#include "engine.hpp"

#include <mutex>

#include <shared_mutex>

#include <cstdlib>

namespace ffp {

ConvProfile::~ConvProfile() {
  for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
}

void ConvProfile::begin() {
  table.clear();
  pending.clear();
  detail.clear();
}

int ConvProfile::open(hipStream_t st) {
  const int slot = (int)pending.size();
  if (slot >= (int)ev.size()) {
    hipEvent_t a, b;
    FFP_HIP(hipEventCreate(&a));
    FFP_HIP(hipEventCreate(&b));
    ev.emplace_back(a, b);
  }
  FFP_HIP(hipEventRecord(ev[slot].first, st));
  return slot;
}

void ConvProfile::close(int slot, hipStream_t st, const std::string& variant, double flops, const std::string& name, double bytes) {
  FFP_HIP(hipEventRecord(ev[slot].second, st));
  pending.push_back({slot, variant, flops, name, bytes});
}

void ConvProfile::collect() {
  for (const Pending& p : pending) {
    float ms = 0.f;
    FFP_HIP(hipEventElapsedTime(&ms, ev[p.ev].first, ev[p.ev].second));
    Entry& e = table[p.variant];
    e.variant = p.variant;
    e.flops += p.flops;
    e.bytes += p.bytes;
    e.ms += ms;
    e.launches += 1;
    Entry d;
    d.variant = p.variant + " " + p.name; d.flops = p.flops; d.ms = ms; d.launches = 1; d.bytes = p.bytes;
    detail.push_back(d);
  }
  pending.clear();
}

Level* Plan::add_level(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st) {
  levels.emplace_back(new Level());
  levels.back()->build(hs, ws, st, px_align);
  return levels.back().get();
}

Level* Plan::add_level_capacity(int cap_n, int64_t cap_px, int cap_t16, hipStream_t st) {
  levels.emplace_back(new Level());
  levels.back()->reserve(cap_n, cap_px, cap_t16, st);
  return levels.back().get();
}

double Step::actual_flops() const {
  if (!is_conv) return 0;
  if (!trunk) return conv_flops_of(*conv->pc, conv->out.lvl->actual_px());
  double f = 0;
  for (const auto& c : fused) f += conv_flops_of(*c->pc, c->out.lvl->actual_px());
  return f;
}

static double conv_bytes_of(const ConvOp& c) {
  const PackedConv& pc = *c.pc;
  const double opx = (double)c.out.lvl->actual_px(), ipx = (double)c.in.lvl->actual_px();
  const double ie = (double)dsize(c.in.dt), oe = (double)dsize(c.out.dt);
  double b = opx * pc.cout * oe;
  if (c.has_up2) b += (double)c.up2.lvl->actual_px() * c.up2_c * (double)dsize(c.up2.dt) + ipx * (pc.cin_real - c.up2_c) * ie;
  else b += ipx * pc.cin_real * ie;
  if (c.has_res1) b += opx * pc.cout * (double)dsize(c.res1.dt);
  if (c.has_res2) b += opx * pc.cout * (double)dsize(c.res2.dt);
  b += (double)pc.cout * (pc.cin_real / pc.groups) * pc.k * pc.k * (double)dsize(pc.dt) * (pc.split ? 1.0 : 1.0) + 4.0 * pc.cout;
  return b;
}

double Step::actual_bytes() const {
  if (!is_conv) return 0;
  if (!trunk) return conv_bytes_of(*conv);
  double b = 0;
  for (const auto& c : fused) b += conv_bytes_of(*c);
  return b;
}

namespace {
std::mutex g_totals_mu;
bool g_totals_on = false;
std::map<std::string, ConvTotals> g_totals;
}  // namespace

void conv_totals_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_totals_mu);
  g_totals_on = on;
  if (on) g_totals.clear();
}

std::vector<ConvTotals> conv_totals() {
  std::lock_guard<std::mutex> lk(g_totals_mu);
  std::vector<ConvTotals> v;
  for (auto& kv : g_totals) v.push_back(kv.second);
  return v;
}

static void totals_add(const std::vector<Step>& steps) {
  std::lock_guard<std::mutex> lk(g_totals_mu);
  if (!g_totals_on) return;
  for (const Step& s : steps) {
    if (s.kind != 0 || !s.is_conv) continue;
    ConvTotals& t = g_totals[s.variant];
    t.variant = s.variant;
    t.flops += s.actual_flops();
    t.bytes += s.actual_bytes();
    t.launches += 1;
  }
}

double Plan::actual_conv_flops() const {
  double f = 0;
  for (const Step& s : steps) f += s.actual_flops();
  return f;
}

TView Plan::alloc(Level* l, int C, DType dt) {
  TView v = alloc_virtual(l, C, dt);
  materialize(v);
  return v;
}

void Plan::materialize(TView& v) {
  if (v.ptr) return;
  const size_t nb = (size_t)v.lvl->total_px * v.cs * dsize(v.dt) + 256;   // slack: vector loads of the last record stay in bounds
  bufs.emplace_back(nb);
  bytes += nb;
  v.ptr = bufs.back().p;
}

TView Plan::alloc_virtual(Level* l, int C, DType dt) {
  TView v;
  v.ptr = nullptr; v.dt = dt; v.cs = C; v.coff = 0; v.C = C; v.lvl = l;
  const int ns = per_image_amax ? l->n : 1;               // slots of this buffer
  if (!amax_slots.p) {
    amax_cap_slots = (size_t)AMAX_CAP * ns;
    amax_slots.alloc(sizeof(unsigned) * amax_cap_slots);
  }
  FFP_CHECK(amax_init.size() + ns <= amax_cap_slots, FFP_ERR_STATE, "plan: more than %d buffers", AMAX_CAP);
  FFP_CHECK(!per_image_amax || l->px_align % 32 == 0, FFP_ERR_STATE, "plan: per-image exponent slots need levels aligned to 32 pixels");
  v.amax = amax_slots.as<unsigned>() + amax_init.size();
  v.amax_n = ns;
  amax_init.insert(amax_init.end(), ns, 0u);
  return v;
}

void Plan::set_amax_bound(const TView& v, float bound) {
  FFP_CHECK(v.amax && amax_slots.p, FFP_ERR_STATE, "plan: view has no max-|value| slot");
  unsigned b;
  std::memcpy(&b, &bound, 4);
  for (int i = 0; i < v.amax_n; ++i) amax_init[(v.amax - amax_slots.as<unsigned>()) + i] = b;
}

void Plan::add_amax_reset(hipStream_t st) {
  if (amax_init.empty()) return;
  amax_init_dev.alloc(sizeof(unsigned) * amax_init.size());
  FFP_HIP(hipMemcpyAsync(amax_init_dev.p, amax_init.data(), sizeof(unsigned) * amax_init.size(), hipMemcpyHostToDevice, st));
  FFP_HIP(hipStreamSynchronize(st));
  unsigned* slots = amax_slots.as<unsigned>();
  const unsigned* init = amax_init_dev.as<unsigned>();
  const int n = (int)amax_init.size();
  if (reset_outside) {
    reset_fn = [slots, init, n](hipStream_t q) { launch_amax_init(slots, init, n, q); };
    return;
  }
  Step s;
  s.run = [slots, init, n](hipStream_t q) { launch_amax_init(slots, init, n, q); };
  steps.insert(steps.begin(), std::move(s));
}

void Plan::add_conv(const ConvOp& op) {
  Step s;
  ConvOp o = op;
  o.flops = conv_flops_of(*op.pc, op.out.lvl->actual_px());
  s.is_conv = true;
  s.variant = conv_variant(o);
  s.name = op.pc->name;
  s.flops = o.flops;
  s.conv = std::make_shared<ConvOp>(o);
  std::shared_ptr<ConvOp> cp = s.conv;
  s.lane = cur_lane;
  s.run = [cp](hipStream_t st) { launch_conv(*cp, st); };
  conv_flops += o.flops;
  conv_launches += 1;
  steps.push_back(std::move(s));
}

void Plan::add_trunk(const std::vector<ConvOp>& ops) {
  Step s;
  s.is_conv = true;
  s.trunk = std::make_shared<TrunkPlan>(ops);
  for (const ConvOp& op : ops) {
    ConvOp o = op;
    o.flops = conv_flops_of(*op.pc, op.out.lvl->actual_px());
    o.force_shape = 25;                                   // nothing to tune: the fused launch has one shape
    s.fused.push_back(std::make_shared<ConvOp>(o));
    s.flops += o.flops;
  }
  s.conv = s.fused.front();
  s.variant = "f16_k3s1_trunk";
  s.name = ops.front().pc->name + ".." + ops.back().pc->name;
  s.lane = cur_lane;
  std::shared_ptr<TrunkPlan> tp = s.trunk;
  s.run = [tp](hipStream_t st) { tp->launch(st); };
  conv_flops += s.flops;
  conv_launches += 1;
  steps.push_back(std::move(s));
}

namespace {
std::shared_mutex g_gate;
thread_local int tl_shared_depth = 0;
}  // namespace

ApiShared::ApiShared() {
  if (tl_shared_depth++ == 0) g_gate.lock_shared();
}
ApiShared::~ApiShared() {
  if (--tl_shared_depth == 0) g_gate.unlock_shared();
}
CaptureExclusive::CaptureExclusive() {
  if (tl_shared_depth > 0) g_gate.unlock_shared();
  g_gate.lock();
}
CaptureExclusive::~CaptureExclusive() {
  g_gate.unlock();
  if (tl_shared_depth > 0) g_gate.lock_shared();
}

Plan::~Plan() {
  if (gexec) (void)hipGraphExecDestroy(gexec);
  if (graph) (void)hipGraphDestroy(graph);
  for (Step& s : steps) if (s.ev) (void)hipEventDestroy(s.ev);
  for (hipStream_t q : lane_streams) (void)hipStreamDestroy(q);
}

int Plan::lanes_env() {
  static const int m = [] { const char* e = getenv("FFP_LANES"); return e ? atoi(e) : -1; }();
  return m;
}

void Plan::fork(int lane) {
  if (!lanes_enabled() || lane <= 0) return;
  while ((int)lane_streams.size() < lane) {
    hipStream_t q;
    FFP_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    lane_streams.push_back(q);
  }
  Step s;
  s.kind = 1; s.lane = lane;
  FFP_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  steps.push_back(std::move(s));
}

void Plan::join(int lane) {
  if (!lanes_enabled() || lane <= 0) return;
  Step s;
  s.kind = 2; s.lane = lane;
  FFP_HIP(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  steps.push_back(std::move(s));
}

// use_lanes = false: every step on `st` in issue order (a valid serialisation: forks / joins only relax the order)
void Plan::run_steps(hipStream_t st, bool use_lanes) {
  for (Step& s : steps) {
    hipStream_t q = (use_lanes && s.lane > 0) ? lane_streams[s.lane - 1] : st;
    if (s.kind == 1) {
      if (use_lanes) { FFP_HIP(hipEventRecord(s.ev, st)); FFP_HIP(hipStreamWaitEvent(q, s.ev, 0)); }
    } else if (s.kind == 2) {
      if (use_lanes) { FFP_HIP(hipEventRecord(s.ev, q)); FFP_HIP(hipStreamWaitEvent(st, s.ev, 0)); }
    } else {
      s.run(q);
    }
  }
}

static bool graphs_enabled() {
  static const bool on = [] { const char* e = getenv("FFP_NO_GRAPH"); return !(e && e[0] == '1'); }();
  return on;
}

void Plan::tune(hipStream_t st) {
  tuned = true;
  static const bool off = [] { const char* e = getenv("FFP_NO_TUNE"); return e && e[0] == '1'; }();
  if (off) return;
  for (Step& s : steps) {
    if (!s.is_conv || s.trunk || s.conv->force_shape >= 0) continue;
    const int best = conv_tune(*s.conv, st);
    if (best >= 0) {
      s.conv->force_shape = best;
      s.variant = conv_variant(*s.conv);
    }
  }
}

void Plan::execute(hipStream_t st, ConvProfile* prof) {
  if (!tuned) tune(st);
  if (g_totals_on) totals_add(steps);
  const bool p = prof && prof->enabled;
  if (p) {                                   // per-launch events: always eager
    for (Step& s : steps) {
      if (s.kind != 0) continue;
      if (s.is_conv) {
        const int slot = prof->open(st);
        s.run(st);
        prof->close(slot, st, s.variant, s.actual_flops(), s.name, s.actual_bytes());
      } else {
        s.run(st);
      }
    }
    return;
  }
  if (runs > 0 && graph_ok && graphs_enabled()) {
    if (!gexec) {
      // first eager run has built every lazily created table; capture the identical sequence now
      CaptureExclusive only_me;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        bool ok = true;
        try {
          run_steps(st, true);
        } catch (...) {
          ok = false;
        }
        hipGraph_t g = nullptr;
        if (hipStreamEndCapture(st, &g) != hipSuccess || !ok || !g) {
          graph_ok = ++capture_failures < 3;          // an invalidated capture is retried on a later run, then given up
          if (!graph_ok) fprintf(stderr, "libffp: hipGraph capture failed %d times, this plan keeps launching eagerly (ffp_*_graph_status reports -1)\n", capture_failures);
          if (g) (void)hipGraphDestroy(g);
          hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
          if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) { g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); }
          (void)hipGetLastError();
        } else if (hipGraphInstantiate(&gexec, g, nullptr, nullptr, 0) != hipSuccess) {
          graph_ok = false;
          gexec = nullptr;
          fprintf(stderr, "libffp: hipGraphInstantiate failed, this plan keeps launching eagerly (ffp_*_graph_status reports -1)\n");
          (void)hipGraphDestroy(g);
          (void)hipGetLastError();
        } else {
          graph = g;
        }
      } else {
        graph_ok = false;
        (void)hipGetLastError();
      }
    }
    if (gexec) {
      FFP_HIP(hipGraphLaunch(gexec, st));
      ++runs;
      return;
    }
  }
  run_steps(st, runs > 0);      // the first run builds the lazily created tables: one stream
  ++runs;
}

}  // namespace ffp
