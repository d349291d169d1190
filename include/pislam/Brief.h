// pislam/Brief.h — drop-in for reference include/Brief.h:57 (briefDescribeRot) and
// Brief.h:637 (briefDescribe).  The rotated sampling table lives in the library.
#ifndef PISLAM_BRIEF_H_
#define PISLAM_BRIEF_H_

#include <cstdint>

#include "Util.h"
#include "detail/Runtime.h"

namespace pislam {

template <int vstep, int words>
void briefDescribe(uint8_t img[][vstep], int x, int y, int rot, uint32_t descriptor[words]) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  const uint32_t point = encodeFast(0, (uint32_t)x, (uint32_t)y);
  const uint8_t rot8 = (rot >= 0 && rot < 30) ? (uint8_t)rot : (uint8_t)255;   // out of range: no write
  detail::check(r, pislam_brief_describe(r.ctx, vstep, words, &img[0][0], &point, &rot8, 1, descriptor),
                "briefDescribe");
}

template <int vstep, int rot, int words>
void briefDescribeRot(uint8_t img[][vstep], int x, int y, uint32_t descriptor[words]) {
  static_assert(rot >= 0 && rot < 30, "rot is discretised to [0..30)");
  briefDescribe<vstep, words>(img, x, y, rot, descriptor);
}

}  // namespace pislam
#endif
