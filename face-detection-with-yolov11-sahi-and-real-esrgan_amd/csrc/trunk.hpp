// trunk.hpp — the fused multi-layer launch of conv_trunk.hip: a run of 3x3 stride-1 fp16 convs over ONE level (the Real-ESRGAN body) walked by a
// persistent grid with per-tile dependency counters instead of one launch per layer.
#pragma once
#include "ops.hpp"

namespace ffp {

struct TrunkLayer {          // one conv of the run (device table, read through the scalar cache)
  const void* in;
  void* out;
  const void* res1;
  const void* res2;
  const void* wpk;           // PackedConv::w16
  const float* bias;
  int in_cs, in_coff, cin;
  int out_cs, out_coff, cout;
  int r1_cs, r1_coff, r2_cs, r2_coff;
  float s1, s2;
  int act;
  int cum;                   // 32-channel output blocks of all earlier layers of the run (an item = one block of one tile: queue ids, dependency counts)
};

struct TrunkArgs {
  const TrunkLayer* layers;
  int n_layers;
  int n_blocks;              // blocks of all layers: items = tiles x n_blocks
  int ntiles_host;
  const int* n_tiles_dev;    // capacity-mode levels: the batch's tile count (see ConvArgs)
  const int4* tiles;         // Level::tile_table_packed(32): {first pixel of the image, y0 | x0 << 16, h | w << 16, tile columns | rows << 16} per 32 x 16 tile,
                             // the tiles of an image row-major and consecutive
  unsigned* queue;           // [0] next item, [1] error word (a dependency wait that gave up)
  unsigned* done;            // [tile] finished (layer, block) items of the tile
  int dbg;
};

// a run of convs laid out for one persistent launch; owns the device layer table and the queue / counter block (zeroed by every launch)
struct TrunkPlan {
  explicit TrunkPlan(const std::vector<ConvOp>& ops);
  void launch(hipStream_t st, int dbg = 0);
  unsigned errors(hipStream_t st);      // dependency waits that gave up during the launches so far (0 always, unless the device is wedged)
  Level* lvl = nullptr;
  int n_layers = 0, n_blocks = 0;
  DevBuf d_layers, sync;
};

bool conv_trunk_layer_ok(const ConvOp& op);     // the conv can be a layer of a fused run (3x3 s1 fp16, cin % 32 == 0, cout 32 or 64, one level)
bool conv_trunk_enabled();                      // FFP_TRUNK=0: one launch per layer (A/B aid, and the fused launch's parity oracle)
void conv_trunk_init();

}  // namespace ffp
