// engine.cpp — plan bookkeeping and conv-kernel profiling.
#include "engine.hpp"

namespace ffp {

ConvProfile::~ConvProfile() {
  for (auto& e : ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
}

void ConvProfile::begin() {
  table.clear();
  pending.clear();
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

void ConvProfile::close(int slot, hipStream_t st, const std::string& variant, double flops) {
  FFP_HIP(hipEventRecord(ev[slot].second, st));
  pending.push_back({slot, variant, flops});
}

void ConvProfile::collect() {
  for (const Pending& p : pending) {
    float ms = 0.f;
    FFP_HIP(hipEventElapsedTime(&ms, ev[p.ev].first, ev[p.ev].second));
    Entry& e = table[p.variant];
    e.variant = p.variant;
    e.flops += p.flops;
    e.ms += ms;
    e.launches += 1;
  }
  pending.clear();
}

std::string conv_variant(const ConvOp& op) {
  const PackedConv& pc = *op.pc;
  const int nt = pc.cout_pad / 32;
  char buf[64];
  snprintf(buf, sizeof(buf), "%s_k%ds%d_%s", pc.dt == F32 ? "f32" : "f16", pc.k, op.stride,
           nt >= 3 ? "wide" : nt == 2 ? "narrow2" : "narrow1");
  return buf;
}

Level* Plan::add_level(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st) {
  levels.emplace_back(new Level());
  levels.back()->build(hs, ws, st);
  return levels.back().get();
}

TView Plan::alloc(Level* l, int C, DType dt) {
  const size_t nb = (size_t)l->total_px * C * dsize(dt) + 256;   // slack: vector loads of the last record stay in bounds
  bufs.emplace_back(nb);
  bytes += nb;
  TView v;
  v.ptr = bufs.back().p; v.dt = dt; v.cs = C; v.coff = 0; v.C = C; v.lvl = l;
  return v;
}

void Plan::add_conv(const ConvOp& op) {
  Step s;
  ConvOp o = op;
  o.flops = conv_flops_of(*op.pc, op.out.lvl->total_px);
  s.is_conv = true;
  s.variant = conv_variant(o);
  s.flops = o.flops;
  s.run = [o](hipStream_t st) { launch_conv(o, st); };
  conv_flops += o.flops;
  conv_launches += 1;
  steps.push_back(std::move(s));
}

void Plan::execute(hipStream_t st, ConvProfile* prof) {
  const bool p = prof && prof->enabled;
  for (Step& s : steps) {
    if (p && s.is_conv) {
      const int slot = prof->open(st);
      s.run(st);
      prof->close(slot, st, s.variant, s.flops);
    } else {
      s.run(st);
    }
  }
}

}  // namespace ffp
