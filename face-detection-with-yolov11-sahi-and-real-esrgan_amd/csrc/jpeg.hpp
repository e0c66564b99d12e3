// jpeg.hpp — baseline JPEG encoding of device-resident images (jpeg.hip): SURVEY.md §8 row f2.
#pragma once
#include <vector>

#include "common.hpp"

namespace ffp {

// SOI .. SOS of the file libjpeg writes for an h x w three-component 4:2:0 baseline image at `quality`
std::vector<unsigned char> jpeg_header(int h, int w, int quality);

// Encodes the h x w x 3 uint8 image at d_img (row pitch `stride` bytes, channel order RGB or BGR) into `out` (host, `cap` bytes):
// the whole JFIF file. Returns its size, or minus the size needed when out is null or too small. Synchronises `st`.
long long jpeg_encode_device(const unsigned char* d_img, int h, int w, long long stride, int bgr, int quality, unsigned char* out, long long cap, hipStream_t st);

}  // namespace ffp
