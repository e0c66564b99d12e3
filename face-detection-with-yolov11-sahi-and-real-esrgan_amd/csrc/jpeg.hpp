// jpeg.hpp — baseline JPEG encoding of device-resident images (jpeg.hip): SURVEY.md §8 row f2.
#pragma once
#include <vector>

#include "common.hpp"

namespace ffp {

// SOI .. SOS of the file libjpeg writes for an h x w three-component 4:2:0 baseline image at `quality`
std::vector<unsigned char> jpeg_header(int h, int w, int quality);

// One launch sequence for a whole batch of device-resident images (e.g. the enhanced crops of a frame in the SR output buffer):
// files[i] = the JFIF file of src[i]. Two stream synchronisations per batch, whatever its size.
struct JpegSrc { const unsigned char* d_img; int h, w; long long stride; };
void jpeg_encode_batch_device(const JpegSrc* src, int n_img, int bgr, int quality, std::vector<std::vector<unsigned char>>& files, hipStream_t st);

// Encodes the h x w x 3 uint8 image at d_img (row pitch `stride` bytes, channel order RGB or BGR) into `out` (host, `cap` bytes):
// the whole JFIF file. Returns its size, or minus the size needed when out is null or too small. Synchronises `st`.
long long jpeg_encode_device(const unsigned char* d_img, int h, int w, long long stride, int bgr, int quality, unsigned char* out, long long cap, hipStream_t st);

// ---- decoding: host entropy decoding (jpeg_dec.cpp) + device reconstruction (jpeg.hip) ------------------------------------------------
struct JpegComp { int hs = 1, vs = 1, tq = 0, td = 0, ta = 0, blocks_x = 0, blocks_y = 0; };
struct JpegScan {
  int h = 0, w = 0, ncomp = 0, hmax = 1, vmax = 1;
  JpegComp comp[3];
  unsigned short qt[4][64];                     // natural order
  short* coef[3] = {nullptr, nullptr, nullptr}; // [blocks_y][blocks_x][64] natural order, quantised: caller's (pinned) memory, zeroed by the decoder
};
// staging of one decode in flight: pinned host coefficient planes (the entropy decoder writes them, the copies to the device run at
// full PCIe rate) and their device twins + sample planes; kept in a pool so that concurrent decodes neither allocate nor free
struct JpegHuffWs;                              // buffers of the device entropy decoder (jpeg_huff.hip)
void jpeg_huff_ws_delete(JpegHuffWs* w);
struct JpegDecodeWs {
  HostPinned host[3];                           // host-decoder path only (allocated on first use)
  DevBuf coef_all, plane[3], qt;                // the three coefficient planes are one allocation (one memset clears them)
  short* dev[3] = {nullptr, nullptr, nullptr};  // component c's plane inside coef_all
  size_t cap[3] = {0, 0, 0}, coef_bytes = 0;    // coef_bytes: what the current scan uses of coef_all
  int device = 0;                               // the pool hands a workspace out only on the device it was made on
  JpegHuffWs* huff = nullptr;
  JpegDecodeWs() = default;
  JpegDecodeWs(const JpegDecodeWs&) = delete;
  JpegDecodeWs& operator=(const JpegDecodeWs&) = delete;
  ~JpegDecodeWs() { if (huff) jpeg_huff_ws_delete(huff); }
  void ensure(const JpegScan& s);               // sizes the device planes for the scan
  void ensure_host(const JpegScan& s);          // ... and the pinned host planes, and points s.coef at them
};
JpegDecodeWs* jpeg_ws_acquire();
void jpeg_ws_release(JpegDecodeWs* ws);
// what the device entropy decoder (jpeg_huff.hip) needs from the markers: the raw Huffman tables, the restart interval and where
// the entropy-coded segment starts
struct JpegHuffSpec { bool present = false; int n = 0; unsigned char bits[16]; unsigned char vals[256]; };
struct JpegHead { JpegHuffSpec dc[4], ac[4]; int dri = 0; long long data_off = 0; };
// markers + Huffman decoding of a baseline / extended-sequential 8-bit JFIF file (one interleaved scan, 1 or 3 components, chroma
// at full, half-width or half-width-half-height resolution; restart intervals). Throws ffp::Error on anything else.
void jpeg_entropy_decode(const unsigned char* data, long long n, JpegScan& out, bool header_only, JpegHead* head = nullptr);
// the same Huffman decoding on the device (jpeg_huff.hip): queues everything on `st` without synchronising; after the caller's
// synchronisation jpeg_huff_finish returns 1 (coefficient planes ws.dev[] are right), 2 (right after further rounds: run the
// reconstruction again) or 0 (the stream needs the host decoder: damaged data, or something libjpeg treats specially)
// returns false, with nothing queued, for a file outside this decoder's limits (a scan of 256 MiB or more, more than six blocks per MCU)
bool jpeg_huff_decode_async(const unsigned char* data, long long n, const JpegScan& s, const JpegHead& head, JpegDecodeWs& ws, hipStream_t st);
int jpeg_huff_finish(const JpegScan& s, JpegDecodeWs& ws, hipStream_t st);
void jpeg_huff_stats(long long* device_decodes, long long* host_fallbacks, long long* extra_sync_rounds);
void jpeg_huff_note_fallback();
// dequantisation + integer IDCT + upsampling + colour conversion into d_out (h x w x 3 uint8, row pitch `stride`, RGB or BGR);
// upload: the coefficients are in s.coef (host decoder) and go to the device first. Queues on `st`, no synchronisation.
void jpeg_reconstruct_device(const JpegScan& s, JpegDecodeWs& ws, unsigned char* d_out, long long stride, int bgr, hipStream_t st, bool upload);
// header -> workspace -> entropy decode -> reconstruction, for a stream of `n` bytes (the three entry points of api.cpp share it)
void jpeg_decode_to_device(const unsigned char* data, long long n, unsigned char* d_out, long long stride, long long cap, int bgr, hipStream_t st, int* out_h, int* out_w);

}  // namespace ffp
