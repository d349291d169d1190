// pislam/Harris.h — drop-in for reference include/Harris.h:80 (harrisScoreSobel).
// harrisEval (Harris.h:37) takes NEON vector types and is an internal helper of the
// reference; its arithmetic lives in the HIP kernel (pislam_dev.h harris_eval).
#ifndef PISLAM_HARRIS_H_
#define PISLAM_HARRIS_H_

#include <cstdint>

#include "Util.h"
#include "detail/Runtime.h"

namespace pislam {

template <int vstep>
uint8_t harrisScoreSobel(uint8_t img[][vstep], int x, int y, int32_t threshold) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  // the library stages the byte hull around (x,y); pass the array base like the reference does
  const uint32_t point = encodeFast(0, (uint32_t)x, (uint32_t)y);
  uint8_t score = 0;
  detail::check(r, pislam_harris_score_points(r.ctx, vstep, &img[0][0], &point, 1, threshold, &score),
                "harrisScoreSobel");
  return score;
}

}  // namespace pislam
#endif
