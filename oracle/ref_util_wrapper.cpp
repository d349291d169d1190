// oracle/ref_util_wrapper.cpp — TEST INFRASTRUCTURE.
// extern "C" shim around the REAL reference header include/Util.h (keypoint word codec, no NEON in it),
// compiled from /root/reference/include where it lies.  Pins the codec the product re-implements
// (include/pislam/Util.h, pislam_amd/frontend.py, pdev::encode_fast).  Built only in the dev container
// by oracle/Makefile into oracle/_ref/.
#include <cstdint>
#include "Util.h"

extern "C" {
uint32_t ref_encodeFast(uint32_t score, uint32_t x, uint32_t y) { return pislam::encodeFast(score, x, y); }        // Util.h:27
uint32_t ref_rencodeFastScore(uint32_t score, uint32_t e) { return pislam::rencodeFastScore(score, e); }             // Util.h:31
uint32_t ref_decodeFastX(uint32_t e) { return pislam::decodeFastX(e); }                                              // Util.h:35
uint32_t ref_decodeFastY(uint32_t e) { return pislam::decodeFastY(e); }                                              // Util.h:39
uint32_t ref_decodeFastScore(uint32_t e) { return pislam::decodeFastScore(e); }                                      // Util.h:43
}
