// weights.hpp — FFPW container parsing and repacking of OIHW fp32 weights into the MFMA fragment layout.
#pragma once
#include "common.hpp"

namespace ffp {

struct HostTensor {
  const float* data = nullptr;
  std::vector<int> dims;
  size_t numel() const { size_t n = 1; for (int d : dims) n *= (size_t)d; return n; }
};

// Parsed FFPW container (weights_io.py). Does not own the bytes.
struct WeightFile {
  std::map<std::string, HostTensor> t;
  void parse(const void* bytes, size_t n);
  const HostTensor& get(const std::string& name) const;
  bool has(const std::string& name) const { return t.count(name) != 0; }
};

// One convolution's parameters on the device.
//  groups == 1 : `w` holds MFMA A-operand fragments, [ntile32][tap][cgroup][lane 0..63][16 bytes]
//                 fp16: cgroup = 16 input channels, lane l element j -> (n = 32*ntile + (l&31), c = 16*cg + 8*(l>>5) + j)
//                 fp32: cgroup =  8 input channels, lane l element j -> (n = 32*ntile + (l&31), c =  8*cg + 4*(l>>5) + j)
//                 zero padded in both n and c.
//  depthwise   : `w` holds fp32 [tap][C].
//  bias: fp32, padded to cout_pad.
struct PackedConv {
  std::string name;
  int cin = 0, cout = 0, k = 1, groups = 1;   // cin is padded up to a 16-byte multiple of the element type
  int cin_real = 0;                            // channels that carry weights (FLOP accounting)
  int cin_pad = 0, cout_pad = 0, ncg = 0;
  DType dt = F32;
  bool split = false;   // dt == F32 only: fragments hold fp16 hi + fp16 lo parts (2 KiB each, k-group = 16) for the X3 kernels
  DevBuf w, bias;
  DevBuf oscale;     // split only: fp32 [cout_pad] = 1 / s_n, the inverse of the power of two each output channel's weights were scaled by
  float out_bound = 0.f;   // image-input convs (3 real channels in [0, 1]): max over channels of sum |w| + |b| >= any |output| (SiLU / none)
  DevBuf w_direct;   // k3 convs with 3 real input channels (stem, conv_first): fp32 [tap][3][cout] for the direct kernel
  // fp16 k3 convs with cin % 32 == 0 and cout % 32 == 0 (Real-ESRGAN body): A fragments of v_mfma_f32_16x16x32_f16 for
  // conv_rows16.hip, [32-channel tile][32-channel chunk][tap][M-tile m][lane l][8 halfs]:
  //   lane l, element j -> (n = 32*tile + 8*((l&15)>>2) + 4*m + (l&3), c = 32*chunk + 8*(l>>4) + j)
  // (the row permutation leaves each accumulator lane with 8 consecutive output channels)
  DevBuf w16;
  bool depthwise() const { return groups > 1; }
};

// w: (cout, cin/groups, k, k) fp32 OIHW, b: (cout) or null (zeros)
void pack_conv(PackedConv& pc, const std::string& name, const float* w, const float* b, int cout, int cin, int k,
               int groups, DType dt, hipStream_t st, bool split = false);
void pack_conv(PackedConv& pc, const WeightFile& wf, const std::string& name, int k, int groups, DType dt,
               hipStream_t st, bool split = false);

}  // namespace ffp
