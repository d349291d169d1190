// oracle/ref_testutil_wrapper.cpp — TEST INFRASTRUCTURE.
// extern "C" shim around the REAL reference test utility test/TestUtil.cpp (plain C++, compiled from
// /root/reference/test where it lies): fill_spiral is the fixture of the reference's GaussianTest and
// BilinearTest; the oracle's orc_fill_spiral restates it and is pinned against this build
// (tests/test_oracle.py).  Built only in the dev container by oracle/Makefile into oracle/_ref/.
#include <cstdint>
#include "TestUtil.h"

extern "C" void ref_fill_spiral(int vstep, int width, int height, int cx, int cy, uint8_t *buffer) {
  test_util::fill_spiral(vstep, width, height, cx, cy, buffer);   // TestUtil.cpp:27
}

extern "C" void ref_fill_random(int vstep, int width, int height, uint8_t *buffer) {
  test_util::fill_random(vstep, width, height, buffer);           // TestUtil.cpp:57
}
