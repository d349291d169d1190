// pislam/Orb.h — drop-in for reference include/Orb.h: orbCentroids (Orb.h:80),
// atan2 (Orb.h:310; `inline` here — the reference's non-inline definition breaks
// multi-TU linking) and orbCompute (Orb.h:396).
#ifndef PISLAM_ORB_H_
#define PISLAM_ORB_H_

#include <cstdint>
#include <vector>

#include "Brief.h"
#include "Util.h"
#include "detail/Runtime.h"

namespace pislam {

template <int vstep>
std::vector<int32_t> orbCentroids(uint8_t img[][vstep], const std::vector<uint32_t> &points) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  std::vector<int32_t> centroids(pislam_centroids_size(points.size()));
  detail::check(r, pislam_orb_centroids(r.ctx, vstep, &img[0][0], points.data(), points.size(),
                                        centroids.data()), "orbCentroids");
  return centroids;
}

inline std::vector<uint8_t> atan2(const std::vector<int32_t> &xys) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  const size_t n8 = xys.size() & ~(size_t)7;       // the reference consumes whole groups of 8
  std::vector<uint8_t> angles(n8 / 2);
  detail::check(r, pislam_orb_angles(r.ctx, xys.data(), n8, angles.data()), "atan2");
  return angles;
}

template <int vstep, int words>
void orbCompute(uint8_t img[][vstep], const std::vector<uint32_t> &points,
                std::vector<uint32_t> &descriptors) {
  detail::Runtime &r = detail::runtime();
  std::lock_guard<std::mutex> g(r.lock);
  const size_t old = descriptors.size();
  descriptors.resize(old + points.size() * words);
  detail::check(r, pislam_orb_compute(r.ctx, vstep, words, &img[0][0], points.data(), points.size(),
                                      descriptors.data() + old), "orbCompute");
}

}  // namespace pislam
#endif
