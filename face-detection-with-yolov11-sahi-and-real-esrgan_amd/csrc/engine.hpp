// engine.hpp — plans (a network laid out for one ragged batch shape), step lists and per-kernel profiling.
#pragma once
#include <functional>

#include "common.hpp"
#include "det_post.hpp"
#include "ops.hpp"
#include "trunk.hpp"
#include "weights.hpp"

namespace ffp {

// Per-launch HIP-event timing of convolution kernels, aggregated by kernel variant ("f32 k3s1 wide", ...).
struct ConvProfile {
  struct Entry { std::string variant; double flops = 0; double ms = 0; int launches = 0; double bytes = 0; };
  bool enabled = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;   // recycled pool
  struct Pending { int ev; std::string variant; double flops; std::string name; double bytes; };
  std::vector<Entry> detail;                      // one entry per launch of the last call (variant = "<variant> <layer name>")
  std::vector<Pending> pending;
  std::map<std::string, Entry> table;
  ~ConvProfile();
  void begin();                                  // clear the table, start collecting
  int open(hipStream_t st);                      // record a start event, returns slot
  void close(int slot, hipStream_t st, const std::string& variant, double flops, const std::string& name, double bytes = 0);
  void collect();                                // after the stream has drained
};

std::string conv_variant(const ConvOp& op);

// One step of a plan. Kept as a closure: the op descriptors are fixed at plan build time.
struct Step {
  std::function<void(hipStream_t)> run;
  bool is_conv = false;
  std::shared_ptr<ConvOp> conv;      // is_conv: the descriptor `run` launches (its workgroup shape can be tuned after the plan is laid out)
  // a fused run of convs (conv_trunk.hip): ONE launch; `fused` lists the layers (FLOP accounting), `conv` is the first of them
  std::shared_ptr<TrunkPlan> trunk;
  std::vector<std::shared_ptr<ConvOp>> fused;
  double actual_flops() const;       // algorithmic FLOPs of the batch the step's level currently describes
  // algorithmic bytes of the same batch: every input, residual and output element once (the real channels at the tensors' element size) plus the
  // weights and the bias once — what a launch must move if nothing is read twice; the PMC traffic of the launch is compared with this
  double actual_bytes() const;
  std::string variant, name;
  double flops = 0;
  // lanes: steps of different lanes have no data dependency between a fork and the matching join and may overlap on the device
  // (independent branches of the captured graph). kind 1 = fork: `lane` starts after everything lane 0 has issued so far;
  // kind 2 = join: lane 0 continues after everything `lane` has issued.
  int lane = 0, kind = 0;
  hipEvent_t ev = nullptr;
};

struct Plan {
  std::vector<std::unique_ptr<Level>> levels;
  std::vector<DevBuf> bufs;
  std::vector<Step> steps;
  double conv_flops = 0;
  int conv_launches = 0;
  size_t bytes = 0;
  // max-|value| slots, one per allocated buffer (TView::amax): reset to `amax_init` by the first step of the plan
  static constexpr int AMAX_CAP = 512;
  // per_image_amax: a slot per (buffer, image) instead of one per buffer — an image's exponent of the scaled split then depends on that
  // image alone, so its results do not depend on its batch mates (sharding-independent bits in FFP_PREC_F32X3). Needs px_align % 32 == 0.
  bool per_image_amax = false;
  int px_align = 1;                    // pixel alignment of the images of every level this plan adds (Level::build)
  size_t amax_cap_slots = 0;
  DevBuf amax_slots, amax_init_dev;
  std::vector<unsigned> amax_init;
  void set_amax_bound(const TView& v, float bound);      // a bound known without looking at data (image inputs, stem output)
  void add_amax_reset(hipStream_t st);                   // uploads the initial values and appends the reset step
  // reset_outside: the owner launches `reset_fn` itself, ahead of work it runs before execute() (the detector's stem-fused first
  // conv raises a max-|value| slot and needs the frame pointer, so it cannot be one of the captured steps)
  bool reset_outside = false;
  std::function<void(hipStream_t)> reset_fn;
  // The launch sequence of a plan is fixed (every pointer is plan-owned), so after one eager run it is captured into a
  // hipGraph and replayed: ~110 (detector) / ~355 (SR) launches per frame become one graph launch.
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  int runs = 0;
  bool graph_ok = true;
  int capture_failures = 0;
  ~Plan();

  Level* add_level(const std::vector<int>& hs, const std::vector<int>& ws, hipStream_t st);
  Level* add_level_capacity(int cap_n, int64_t cap_px, int cap_t16, hipStream_t st);     // Level::reserve
  double actual_conv_flops() const;     // algorithmic FLOPs of the batch the levels currently describe (capacity-mode plans)
  int graph_state() const { return gexec ? 1 : graph_ok ? 0 : -1; }    // 1 replaying a captured graph, 0 not captured yet, -1 capture failed: eager
  TView alloc(Level* l, int C, DType dt);
  TView alloc_virtual(Level* l, int C, DType dt);        // a view with a max-|value| slot but no storage yet: a tensor that may be fused away
  void materialize(TView& v);                            // ... and its storage, once some kernel turns out to need it
  void add_conv(const ConvOp& op);
  void add_trunk(const std::vector<ConvOp>& ops);        // the ops as ONE persistent launch (every op: conv_trunk_layer_ok)
  void add(std::function<void(hipStream_t)> f) { Step s; s.run = std::move(f); s.lane = cur_lane; steps.push_back(std::move(s)); }
  // side lanes (FFP_LANES=0: everything stays on lane 0). New steps go to `cur_lane`.
  int cur_lane = 0;
  std::vector<hipStream_t> lane_streams;                 // lane k > 0 -> lane_streams[k - 1]
  // lanes: 0 off (default: beside the enhancer's stream more overlap costs more than it gains), 1 every branch its
  // own lane (detector-only deployments), 2 one lane per head level, 3 head level 0 only. The owner sets it before building;
  // env FFP_LANES overrides it for A/B runs.
  int lanes = 0;
  static int lanes_env();                                // FFP_LANES or -1
  int lane_mode() const { return lanes_env() >= 0 ? lanes_env() : lanes; }
  bool lanes_enabled() const { return lane_mode() != 0; }
  void fork(int lane);                                   // no-op when lanes are disabled (the steps then run in issue order on lane 0)
  void join(int lane);
  void run_steps(hipStream_t st, bool use_lanes);
  void execute(hipStream_t st, ConvProfile* prof);
  // time every valid workgroup shape of every dense conv on the plan's own buffers and keep the fastest (all shapes give
  // bit-identical results: the accumulation order over k does not depend on the shape). FFP_NO_TUNE=1 keeps the heuristic.
  void tune(hipStream_t st);
  bool tuned = false;
};

// Process-wide gate between hipGraph capture and everything else the library does on other host threads: an allocation,
// a blocking copy or a synchronisation issued by another thread while a stream is being captured can invalidate the
// capture on ROCm (seen with one host thread per engine). Every C-ABI entry holds the gate shared; a capture trades its
// share for exclusive ownership (~1 ms, once per plan).
struct ApiShared {
  ApiShared();
  ~ApiShared();
};
struct CaptureExclusive {
  CaptureExclusive();
  ~CaptureExclusive();
};

// Lifetime totals per kernel variant over every plan execution of the process (eager, profiled or graph replay): launches, algorithmic FLOPs
// and algorithmic bytes. Off by default (a map walk per execution); bench.py switches it on for the run a PMC pass measures, so that the
// counters' per-launch traffic and the per-launch algorithmic figures describe ONE population of launches (ffp_conv_totals_*).
struct ConvTotals { std::string variant; double flops = 0, bytes = 0; long long launches = 0; };
void conv_totals_enable(bool on);
std::vector<ConvTotals> conv_totals();

inline double conv_flops_of(const PackedConv& pc, int64_t out_px) {
  return 2.0 * (double)(pc.cin_real / pc.groups) * pc.k * pc.k * (double)pc.cout * (double)out_px;
}

}  // namespace ffp
