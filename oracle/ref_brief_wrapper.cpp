// oracle/ref_brief_wrapper.cpp — TEST INFRASTRUCTURE.
// Thin extern "C" shim around the REAL reference header Brief.h (compiled from
// /root/reference/include where it lies; it has no NEON in it and builds with
// plain g++).  Used to pin the oracle's BRIEF table/descriptors and to generate
// brief_pattern_base.inc by single-pixel probing.  Built only in the dev
// container by oracle/Makefile into oracle/_ref/ (git-ignored).
#include <cstdint>
#include <cstddef>
#include "Brief.h"

extern "C" {
// Brief.h:637 briefDescribe<vstep,words>(img, x, y, rot, descriptor)
void ref_brief_describe_64(uint8_t *img, int x, int y, int rot, uint32_t *out8) {
  pislam::briefDescribe<64, 8>((uint8_t (*)[64])img, x, y, rot, out8);
}
void ref_brief_describe_640(uint8_t *img, int x, int y, int rot, uint32_t *out8) {
  pislam::briefDescribe<640, 8>((uint8_t (*)[640])img, x, y, rot, out8);
}
}
