// pislam/Util.h — keypoint word codec, drop-in for reference include/Util.h:27-45.
// Layout: bits 31..24 score, 23..12 x, 11..0 y.
#ifndef PISLAM_UTIL_H_
#define PISLAM_UTIL_H_

#include <cstdint>

namespace pislam {

static inline uint32_t encodeFast(uint32_t score, uint32_t x, uint32_t y) {
  return y | (x << 12) | (score << 24);
}
static inline uint32_t rencodeFastScore(uint32_t score, uint32_t encoded) {
  return (encoded & 0x00ffffffu) | (score << 24);
}
static inline uint32_t decodeFastX(uint32_t encoded) { return (encoded >> 12) & 0xfffu; }
static inline uint32_t decodeFastY(uint32_t encoded) { return encoded & 0xfffu; }
static inline uint32_t decodeFastScore(uint32_t encoded) { return encoded >> 24; }

}  // namespace pislam
#endif
