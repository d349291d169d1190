// pislam/Gaussian.h — drop-in for reference include/Gaussian.h:48 (gaussian5x5<vstep>).
#ifndef PISLAM_GAUSSIAN_H_
#define PISLAM_GAUSSIAN_H_

#include <cstdint>

#include "detail/Runtime.h"

namespace pislam {

/// 5x5 binomial blur (rounding-halving-add tree, reflect-101 borders); img may equal out.
template <int vstep>
void gaussian5x5(const int width, const int height, uint8_t img[][vstep], uint8_t out[][vstep]) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  detail::check(r, pislam_gaussian5x5(r.ctx, vstep, width, height, &img[0][0], &out[0][0]), "gaussian5x5");
}

}  // namespace pislam
#endif
