// pislam/Bilinear.h — drop-in for reference include/Bilinear.h:42 (bilinear7_8<vstep>) and
// Bilinear.h:165 (bilinear13_16<vstep>).
#ifndef PISLAM_BILINEAR_H_
#define PISLAM_BILINEAR_H_

#include <cstdint>

#include "detail/Runtime.h"

namespace pislam {

/// Reduce by 7/8 (image padded to a multiple of 8; output dimensions round down); img may equal out.
template <int vstep>
void bilinear7_8(const int width, const int height, uint8_t img[][vstep], uint8_t out[][vstep]) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  detail::check(r, pislam_bilinear7_8(r.ctx, vstep, width, height, &img[0][0], &out[0][0]), "bilinear7_8");
}

/// Reduce by 13/16 (image padded to a multiple of 16; output dimensions round down); img may equal out.
template <int vstep>
void bilinear13_16(const int width, const int height, uint8_t img[][vstep], uint8_t out[][vstep]) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  detail::check(r, pislam_bilinear13_16(r.ctx, vstep, width, height, &img[0][0], &out[0][0]), "bilinear13_16");
}

}  // namespace pislam
#endif
