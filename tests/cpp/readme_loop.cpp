// tests/cpp/readme_loop.cpp — the reference's README.md:59-82 pyramid loop, compiled UNCHANGED in
// structure against include/pislam/*.h (the drop-in headers) and linked with libpislam_hip.so.
// usage: readme_loop <raw 640x2210 grey pyramid> <out.bin> [logBucketSize-variant: 0|1]
// out.bin: uint32 n_keypoints, uint32 n_desc_words, keypoints[n], descriptors[n*8],
//          then uint32 n_centroids, centroids (int32), uint32 n_angles, angles (uint8)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pislam/Fast.h"
#include "pislam/Orb.h"

struct Level { int width, height; };
static const Level pyramidLevels[8] = {{640, 480}, {533, 400}, {444, 333}, {370, 278},
                                       {309, 231}, {257, 193}, {214, 161}, {179, 134}};
static uint8_t img[2210][640];
static uint8_t out[2210][640];

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(img, 1, sizeof(img), f) != sizeof(img)) return 3;
  fclose(f);
  const bool buckets = argc > 3 && atoi(argv[3]) != 0;

  std::vector<uint32_t> keypoints;
  std::vector<uint32_t> descriptors;
  int y = 0;
  for (int level = 0; level < 8; level += 1) {
    int oldSize = keypoints.size();
    int width = pyramidLevels[level].width;
    int height = pyramidLevels[level].height;
    pislam::fastDetect<640, 16>(width, height, &img[y], &out[y], 20);
    pislam::fastScoreHarris<640, 16>(width, height, &img[y], 1 << 15, &out[y]);
    if (buckets)
      pislam::fastExtract<640, 16, 4, 3>(width, height, &out[y], keypoints);
    else
      pislam::fastExtract<640, 16>(width, height, &out[y], keypoints);
    for (auto it = keypoints.begin() + oldSize; it < keypoints.end(); ++it) (*it) += y;
    y += height;
  }
  pislam::orbCompute<640, 8>(img, keypoints, descriptors);
  std::vector<int32_t> centroids = pislam::orbCentroids<640>(img, keypoints);
  std::vector<uint8_t> angles = pislam::atan2(centroids);

  FILE *o = fopen(argv[2], "wb");
  if (!o) return 4;
  uint32_t n = keypoints.size(), m = descriptors.size(), nc = centroids.size(), na = angles.size();
  fwrite(&n, 4, 1, o);
  fwrite(&m, 4, 1, o);
  fwrite(keypoints.data(), 4, n, o);
  fwrite(descriptors.data(), 4, m, o);
  fwrite(&nc, 4, 1, o);
  fwrite(centroids.data(), 4, nc, o);
  fwrite(&na, 4, 1, o);
  fwrite(angles.data(), 1, na, o);
  fclose(o);
  printf("%u features\n", n);
  return 0;
}
