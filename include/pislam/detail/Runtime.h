// pislam/detail/Runtime.h — process-wide pislam_ctx used by the drop-in templates.
//
// The reference's functions are stateless, re-entrant free functions, so the
// wrappers keep one lazily created context PER THREAD (current HIP device, a
// stream of its own): calls from different threads run on different contexts
// and streams and do not serialise each other; calls of one thread are ordered.
// Errors from the C ABI become std::runtime_error (the reference itself never
// reports errors; there is deliberately no CPU fallback).
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
    (void)pislam_ctx_set_option(ctx, "own_stream", 1);   // not the (process-wide, synchronising) null stream
  }
  ~Runtime() {
    if (ctx) pislam_ctx_destroy(ctx);
  }
  Runtime(const Runtime &) = delete;
  Runtime &operator=(const Runtime &) = delete;
};

inline Runtime &runtime() {
  static thread_local Runtime r;
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
