// pislam/Fast.h — drop-in for reference include/Fast.h: same names, template
// parameter lists and argument order; bodies forward to the MI355X library.
#ifndef PISLAM_FAST_H_
#define PISLAM_FAST_H_

#include <cstddef>
#include <cstdint>
#include <vector>

#include "Harris.h"
#include "Util.h"
#include "detail/Runtime.h"

namespace pislam {

/// reference Fast.h:54 — FAST-9 map, 0xff / 0x00 into `out`.
template <int vstep, int border>
void fastDetect(const int width, const int height, uint8_t img[][vstep], uint8_t out[][vstep],
                int threshold) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  detail::check(r, pislam_fast_detect(r.ctx, vstep, border, width, height, &img[0][0], &out[0][0],
                                      threshold), "fastDetect");
}

/// reference Fast.h:166 — replace non-zero bytes of `out` by the 8-bit Harris score.
template <int vstep, int border>
void fastScoreHarris(int width, int height, uint8_t img[][vstep], int32_t threshold,
                     uint8_t out[][vstep]) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  detail::check(r, pislam_fast_score_harris(r.ctx, vstep, border, width, height, &img[0][0], threshold,
                                            &out[0][0]), "fastScoreHarris");
}

/// reference Fast.h:196 — NMS (+ optional buckets); appends to `results` and returns a copy of it.
template <int vstep, int border, int logBucketSize = 0, int bucketLimit = 5>
std::vector<uint32_t> fastExtract(const int width, const int height, uint8_t out[][vstep],
                                  std::vector<uint32_t> &results) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  const size_t old = results.size();
  const size_t cap = (size_t)((width + 1) / 2) * (size_t)((height + 1) / 2);   // one per 2x2 block
  results.resize(old + cap);
  size_t n = 0;
  const int rc = pislam_fast_extract(r.ctx, vstep, border, logBucketSize, bucketLimit, width, height,
                                     &out[0][0], results.data() + old, cap, &n);
  results.resize(old + (rc == PISLAM_OK && n < cap ? n : (rc == PISLAM_OK ? cap : 0)));
  detail::check(r, rc, "fastExtract");
  return results;
}

}  // namespace pislam
#endif
