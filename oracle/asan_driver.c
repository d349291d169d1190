/*
 * oracle/asan_driver.c — TEST INFRASTRUCTURE (SURVEY §5 "sanitizers"): a stand-alone program over
 * oracle/pislam_oracle.c, built by `make -C oracle _asan/orc_asan` with -fsanitize=address,undefined
 * (no recovery), so that tests/test_sanitizers.py can run the oracle's whole path — and its Gaussian /
 * bilinear restatements — on the reference's demo pyramid and on adversarial levels under the sanitizers,
 * and compare the printed checksum with what the un-instrumented liborc.so returns for the same input.
 *
 * usage: orc_asan pyramid <file.raw> <vstep> <rows> <border> <thr> <hthr> <lbs> <limit> <words> <cap> <nlev> {w h row0 col0}*
 *        orc_asan prep <gaussian|b78|b1316> <file.raw> <vstep> <rows> <w> <h>
 * The buffers are EXACTLY rows * vstep bytes (heap): a read or write past the caller's pyramid is an ASan report.
 * prints: n=<keypoints> fnv=<FNV-1a 64 over the kept keypoint words, then the descriptor words>   (pyramid)
 *         fnv=<FNV-1a 64 over the buffer after the in-place call>                                  (prep)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

size_t orc_pyramid4(int vstep, int border, int fast_threshold, int32_t harris_threshold, int logBucketSize,
                    int bucketLimit, int words, const uint8_t *img, uint8_t *score, const int32_t *levels4,
                    int nlevels, uint32_t *kp, uint32_t *desc, size_t cap, uint32_t *level_counts);
void orc_gaussian5x5(int vstep, int width, int height, uint8_t *m);
void orc_bilinear7_8(int vstep, int width, int height, uint8_t *m);
void orc_bilinear13_16(int vstep, int width, int height, uint8_t *m);

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
  const uint8_t *b = (const uint8_t *)p;
  for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001b3ull;
  return h;
}

static uint8_t *read_exact(const char *path, size_t bytes) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  uint8_t *buf = (uint8_t *)malloc(bytes);
  if (fread(buf, 1, bytes, f) != bytes) { fprintf(stderr, "%s: short read\n", path); exit(2); }
  fclose(f);
  return buf;
}

int main(int argc, char **argv) {
  if (argc >= 8 && !strcmp(argv[1], "prep")) {
    const int vstep = atoi(argv[4]), rows = atoi(argv[5]), w = atoi(argv[6]), h = atoi(argv[7]);
    uint8_t *m = read_exact(argv[3], (size_t)vstep * rows);
    if (!strcmp(argv[2], "gaussian")) orc_gaussian5x5(vstep, w, h, m);
    else if (!strcmp(argv[2], "b78")) orc_bilinear7_8(vstep, w, h, m);
    else orc_bilinear13_16(vstep, w, h, m);
    printf("fnv=%016llx\n", (unsigned long long)fnv(0xcbf29ce484222325ull, m, (size_t)vstep * rows));
    free(m);
    return 0;
  }
  if (argc < 14 || strcmp(argv[1], "pyramid")) { fprintf(stderr, "usage: see oracle/asan_driver.c\n"); return 2; }
  const int vstep = atoi(argv[3]), rows = atoi(argv[4]), border = atoi(argv[5]), thr = atoi(argv[6]);
  const int32_t hthr = (int32_t)atol(argv[7]);
  const int lbs = atoi(argv[8]), limit = atoi(argv[9]), words = atoi(argv[10]);
  const size_t cap = (size_t)atol(argv[11]);
  const int nlev = atoi(argv[12]);
  if (argc != 13 + 4 * nlev) { fprintf(stderr, "level table: 4 numbers per level\n"); return 2; }
  int32_t *lv = (int32_t *)malloc(sizeof(int32_t) * 4 * (size_t)nlev);
  for (int i = 0; i < 4 * nlev; i++) lv[i] = atoi(argv[13 + i]);
  const size_t bytes = (size_t)vstep * rows;
  uint8_t *img = read_exact(argv[2], bytes);
  uint8_t *score = (uint8_t *)calloc(bytes, 1);                     /* Fast.h:42-44: `out` starts as zeros */
  uint32_t *kp = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
  uint32_t *desc = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1) * (size_t)words);
  const size_t n = orc_pyramid4(vstep, border, thr, hthr, lbs, limit, words, img, score, lv, nlev, kp, desc, cap, NULL);
  const size_t kept = n < cap ? n : cap;
  uint64_t h = fnv(0xcbf29ce484222325ull, kp, sizeof(uint32_t) * kept);
  h = fnv(h, desc, sizeof(uint32_t) * kept * (size_t)words);
  printf("n=%zu fnv=%016llx\n", n, (unsigned long long)h);
  free(desc); free(kp); free(score); free(img); free(lv);
  return 0;
}
