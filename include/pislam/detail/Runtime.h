// pislam/detail/Runtime.h — process-wide pislam_ctx used by the drop-in templates.
//
// The reference's functions are stateless free functions, so the wrappers keep
// one lazily created context (current HIP device, null stream) and serialise
// calls on it.  Errors from the C ABI become std::runtime_error (the reference
// itself never reports errors; there is deliberately no CPU fallback).
#ifndef PISLAM_DETAIL_RUNTIME_H_
#define PISLAM_DETAIL_RUNTIME_H_

#include <mutex>
#include <stdexcept>
#include <string>

#include "../../pislam_hip.h"

namespace pislam {
namespace detail {

struct Runtime {
  pislam_ctx *ctx;
  std::mutex lock;
  Runtime() : ctx(nullptr) {
    const int rc = pislam_ctx_create(-1, &ctx);
    if (rc != PISLAM_OK)
      throw std::runtime_error("pislam: no usable MI355X/HIP device (pislam_ctx_create = " +
                               std::to_string(rc) + ")");
  }
  ~Runtime() {
    if (ctx) pislam_ctx_destroy(ctx);
  }
  Runtime(const Runtime &) = delete;
  Runtime &operator=(const Runtime &) = delete;
};

inline Runtime &runtime() {
  static Runtime r;
  return r;
}

inline void check(Runtime &r, int rc, const char *what) {
  if (rc != PISLAM_OK)
    throw std::runtime_error(std::string("pislam: ") + what + " failed (" + std::to_string(rc) +
                             "): " + pislam_last_error(r.ctx));
}

}  // namespace detail
}  // namespace pislam
#endif
