// weights.cpp — FFPW parsing + MFMA-fragment weight packing (host side).
#include "weights.hpp"

#include <algorithm>
#include <cmath>

namespace ffp {

namespace {
template <class T> T rd(const uint8_t*& p, const uint8_t* end) {
  FFP_CHECK(p + sizeof(T) <= end, FFP_ERR_WEIGHTS, "FFPW: truncated header");
  T v;
  std::memcpy(&v, p, sizeof(T));
  p += sizeof(T);
  return v;
}
}  // namespace

void WeightFile::parse(const void* bytes, size_t n) {
  const uint8_t* base = static_cast<const uint8_t*>(bytes);
  const uint8_t* p = base;
  const uint8_t* end = base + n;
  FFP_CHECK(bytes && n >= 20 && std::memcmp(p, "FFPW", 4) == 0, FFP_ERR_WEIGHTS, "not an FFPW container");
  p += 4;
  uint32_t ver = rd<uint32_t>(p, end), cnt = rd<uint32_t>(p, end);
  uint64_t data_off = rd<uint64_t>(p, end);
  FFP_CHECK(ver == 1, FFP_ERR_WEIGHTS, "FFPW: unsupported version %u", ver);
  FFP_CHECK(data_off <= n, FFP_ERR_WEIGHTS, "FFPW: bad data offset");
  for (uint32_t i = 0; i < cnt; ++i) {
    uint16_t ln = rd<uint16_t>(p, end);
    FFP_CHECK(p + ln <= end, FFP_ERR_WEIGHTS, "FFPW: truncated name");
    std::string name(reinterpret_cast<const char*>(p), ln);
    p += ln;
    uint8_t dt = rd<uint8_t>(p, end), nd = rd<uint8_t>(p, end);
    FFP_CHECK(dt == 0, FFP_ERR_WEIGHTS, "FFPW: tensor %s has unsupported dtype %d", name.c_str(), dt);
    HostTensor ht;
    for (int d = 0; d < nd; ++d) ht.dims.push_back((int)rd<uint32_t>(p, end));
    uint64_t off = rd<uint64_t>(p, end), nb = rd<uint64_t>(p, end);
    // no sum of file-provided 64-bit fields: each is checked against what is left, so nothing can wrap
    FFP_CHECK(off <= n - data_off && nb <= n - data_off - off && nb / 4 == ht.numel() && nb % 4 == 0, FFP_ERR_WEIGHTS,
              "FFPW: tensor %s out of bounds", name.c_str());
    FFP_CHECK(reinterpret_cast<uintptr_t>(base + data_off + off) % alignof(float) == 0, FFP_ERR_WEIGHTS, "FFPW: tensor %s is not 4-byte aligned", name.c_str());
    ht.data = reinterpret_cast<const float*>(base + data_off + off);
    t[name] = ht;
  }
}

const HostTensor& WeightFile::get(const std::string& name) const {
  auto it = t.find(name);
  FFP_CHECK(it != t.end(), FFP_ERR_WEIGHTS, "weight tensor '%s' missing from container", name.c_str());
  return it->second;
}

static inline uint16_t f32_to_f16_bits(float f) {
  _Float16 h = (_Float16)f;  // round-to-nearest-even, host compiler supports _Float16 on x86-64
  uint16_t b;
  std::memcpy(&b, &h, 2);
  return b;
}

void pack_conv(PackedConv& pc, const std::string& name, const float* w, const float* b, int cout, int cin, int k,
               int groups, DType dt, hipStream_t st, bool split) {
  FFP_CHECK(k == 1 || k == 3, FFP_ERR_ARG, "conv %s: kernel size %d unsupported", name.c_str(), k);
  FFP_CHECK(!split || dt == F32, FFP_ERR_ARG, "conv %s: split packing is an fp32-storage mode", name.c_str());
  pc.split = split && groups == 1;
  FFP_CHECK(groups == 1 || (groups == cin && cin == cout), FFP_ERR_ARG, "conv %s: only dense or depthwise groups", name.c_str());
  const int taps = k * k;
  std::vector<float> wpad;
  const int cin_real = cin;
  if (groups == 1) {   // activations are addressed in 16-byte vectors: pad the input channels with zero weights
    const int epv = dt == F16 ? 8 : 4;
    const int cin_al = (cin + epv - 1) / epv * epv;
    if (cin_al != cin) {
      wpad.assign((size_t)cout * cin_al * taps, 0.f);
      for (int n = 0; n < cout; ++n)
        for (int c = 0; c < cin; ++c)
          for (int t = 0; t < taps; ++t) wpad[((size_t)n * cin_al + c) * taps + t] = w[((size_t)n * cin + c) * taps + t];
      w = wpad.data();
      cin = cin_al;
    }
  }
  pc.name = name; pc.cin = cin; pc.cin_real = cin_real; pc.cout = cout; pc.k = k; pc.groups = groups; pc.dt = dt;
  pc.cout_pad = (cout + 31) / 32 * 32;
  std::vector<float> hb(pc.cout_pad, 0.f);
  if (b) std::memcpy(hb.data(), b, sizeof(float) * cout);
  pc.bias.alloc(hb.size() * 4);
  FFP_HIP(hipMemcpyAsync(pc.bias.p, hb.data(), hb.size() * 4, hipMemcpyHostToDevice, st));
  if (groups > 1) {
    std::vector<float> hw((size_t)taps * cout);
    for (int c = 0; c < cout; ++c)
      for (int t = 0; t < taps; ++t) hw[(size_t)t * cout + c] = w[(size_t)c * taps + t];
    pc.cin_pad = cin; pc.ncg = 0;
    pc.w.alloc(hw.size() * 4);
    FFP_HIP(hipMemcpyAsync(pc.w.p, hw.data(), hw.size() * 4, hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
    return;
  }
  if (cin_real == 3 && k == 3) {         // image-input convs also get a direct (VALU) form: [tap][ci][cout_pad]
    std::vector<float> hd((size_t)taps * 3 * cout, 0.f);    // [tap][ci][cout], unpadded (the direct kernel is instantiated per cout)
    for (int n = 0; n < cout; ++n)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < taps; ++t) hd[((size_t)t * 3 + c) * cout + n] = w[((size_t)n * cin + c) * taps + t];
    pc.w_direct.alloc(hd.size() * 4);
    FFP_HIP(hipMemcpyAsync(pc.w_direct.p, hd.data(), hd.size() * 4, hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
  }
  if (dt == F16 && k == 3 && cin % 32 == 0 && cout % 32 == 0) {
    const int nt32 = cout / 32, nch = cin / 32;
    std::vector<uint16_t> h16((size_t)nt32 * taps * nch * 2 * 64 * 8);
    size_t o = 0;
    for (int nt = 0; nt < nt32; ++nt)
      for (int ch = 0; ch < nch; ++ch)
        for (int t = 0; t < taps; ++t)
          for (int m = 0; m < 2; ++m)
            for (int l = 0; l < 64; ++l) {
              const int n = nt * 32 + 8 * ((l & 15) >> 2) + 4 * m + (l & 3);
              for (int j = 0; j < 8; ++j) {
                const int c = ch * 32 + 8 * (l >> 4) + j;
                h16[o++] = f32_to_f16_bits(w[((size_t)n * cin + c) * taps + t]);
              }
            }
    pc.w16.alloc(h16.size() * 2);
    FFP_HIP(hipMemcpyAsync(pc.w16.p, h16.data(), h16.size() * 2, hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
  }
  std::vector<float> wsc(pc.cout_pad, 1.f);           // per-output-channel power of two of the split packing
  if (pc.split) {
    // fp16 hi + lo keep ~22 bits of a value only while the lo part is a normal fp16 number, i.e. for |value| >= 2^-3, and fp16
    // overflows at 65504: scale each output channel's weights so that its largest one sits in [2^13, 2^14) (an exact power of
    // two, undone in the epilogue). Smaller weights of the channel then carry an ABSOLUTE error <= 2^-25 (fp16 subnormal
    // spacing), 2^-38 of the channel's largest weight — below fp32's own rounding of the sum.
    std::vector<float> inv(pc.cout_pad, 1.f);
    for (int n = 0; n < cout; ++n) {
      float m = 0.f;
      for (size_t i = 0; i < (size_t)cin * taps; ++i) m = std::max(m, std::fabs(w[(size_t)n * cin * taps + i]));
      if (m > 0.f && std::isfinite(m)) {
        int e;
        std::frexp(m, &e);                            // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(14 - e) in [2^13, 2^14)
        wsc[n] = std::ldexp(1.f, 14 - e);
        inv[n] = std::ldexp(1.f, e - 14);
      }
    }
    pc.oscale.alloc(inv.size() * 4);
    FFP_HIP(hipMemcpyAsync(pc.oscale.p, inv.data(), inv.size() * 4, hipMemcpyHostToDevice, st));
    FFP_HIP(hipStreamSynchronize(st));
  }
  if (cin_real == 3 && k == 3) {
    float bound = 0.f;
    for (int n = 0; n < cout; ++n) {
      float sum = b ? std::fabs(b[n]) : 0.f;
      for (size_t i = 0; i < (size_t)cin * taps; ++i) sum += std::fabs(w[(size_t)n * cin * taps + i]);
      bound = std::max(bound, sum);
    }
    pc.out_bound = bound;
  }
  const int KG = (dt == F16 || pc.split) ? 16 : 8;    // input channels per fragment group
  const int EH = KG / 2;                // elements per lane (8 halfs / 4 floats = 16 bytes)
  pc.cin_pad = (cin + KG - 1) / KG * KG;
  pc.ncg = pc.cin_pad / KG;
  const int ntile = pc.cout_pad / 32;
  const size_t nfrag = (size_t)ntile * taps * pc.ncg;
  const size_t fbytes = pc.split ? 2048 : 1024;
  std::vector<uint8_t> hw(nfrag * fbytes, 0);
  for (int nt = 0; nt < ntile; ++nt)
    for (int t = 0; t < taps; ++t)
      for (int cg = 0; cg < pc.ncg; ++cg) {
        uint8_t* frag = hw.data() + (((size_t)nt * taps + t) * pc.ncg + cg) * fbytes;
        for (int l = 0; l < 64; ++l) {
          const int n = nt * 32 + (l & 31);
          for (int j = 0; j < EH; ++j) {
            const int c = cg * KG + EH * (l >> 5) + j;
            float v = (n < cout && c < cin) ? w[((size_t)n * cin + c) * taps + t] : 0.f;
            if (pc.split) {          // hi = fp16(v * s_n), lo = fp16(v * s_n - hi): hi fragment then lo fragment
              v *= wsc[std::min(n, pc.cout_pad - 1)];
              const _Float16 hi = (_Float16)v;
              const _Float16 lo = (_Float16)(v - (float)hi);
              std::memcpy(frag + l * 16 + j * 2, &hi, 2);
              std::memcpy(frag + 1024 + l * 16 + j * 2, &lo, 2);
            } else if (dt == F16) {
              uint16_t hbits = f32_to_f16_bits(v);
              std::memcpy(frag + l * 16 + j * 2, &hbits, 2);
            } else {
              std::memcpy(frag + l * 16 + j * 4, &v, 4);
            }
          }
        }
      }
  pc.w.alloc(hw.size());
  FFP_HIP(hipMemcpyAsync(pc.w.p, hw.data(), hw.size(), hipMemcpyHostToDevice, st));
  FFP_HIP(hipStreamSynchronize(st));
}

void pack_conv(PackedConv& pc, const WeightFile& wf, const std::string& name, int k, int groups, DType dt,
               hipStream_t st, bool split) {
  const HostTensor& w = wf.get(name + ".weight");
  const HostTensor& b = wf.get(name + ".bias");
  FFP_CHECK(w.dims.size() == 4 && w.dims[2] == k && w.dims[3] == k, FFP_ERR_WEIGHTS, "%s.weight: expected (co,ci,%d,%d)", name.c_str(), k, k);
  const int cout = w.dims[0];
  const int cin = w.dims[1] * groups;
  FFP_CHECK(groups == 1 || w.dims[1] == 1, FFP_ERR_WEIGHTS, "%s.weight: depthwise expects (c,1,k,k)", name.c_str());
  FFP_CHECK((int)b.numel() == cout, FFP_ERR_WEIGHTS, "%s.bias: expected %d values", name.c_str(), cout);
  pack_conv(pc, name, w.data, b.data, cout, cin, k, groups, dt, st, split);
}

}  // namespace ffp
