// emu_ctx.cpp -- the context / error plumbing of api.hip for the host-emulated bundle-adjustment library (test infrastructure only:
// tests/native/build_emu.py links it with ba.hip and ba_general.hip compiled against tests/native/hipemu)
#include <cstdarg>
#include <cstdio>
#include <new>

#include "osfm_internal.h"

static thread_local char g_err[1024] = "";
thread_local unsigned osfm_error_epoch = 0;

void osfm_set_error(const char *fmt, ...) {
  ++osfm_error_epoch;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *osfm_last_error(void) { return g_err; }
extern "C" int osfm_ctx_create(int device, osfm_ctx **out) {
  osfm_ctx *c = new (std::nothrow) osfm_ctx();
  if (!c) return OSFM_E_NOMEM;
  c->device = device;
  c->num_cus = 256;
  (void)hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  for (int i = 0; i < 8; ++i) (void)hipEventCreate(&c->ev[i]);
  *out = c;
  return OSFM_OK;
}
extern "C" void osfm_ctx_destroy(osfm_ctx *c) {
  if (!c) return;
  for (int i = 0; i < 8; ++i) (void)hipEventDestroy(c->ev[i]);
  (void)hipStreamDestroy(c->stream);
  if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (int i = 0; i < 2; ++i)
    if (c->ev_side[i]) (void)hipEventDestroy(c->ev_side[i]);
  for (auto &b : c->pool) (void)hipFree(b.p);
  delete c;
}
extern "C" int64_t osfm_ctx_trim_pool(osfm_ctx *) { return 0; }
hipError_t osfm_malloc_retry(osfm_ctx *, void **p, size_t bytes) { return hipMalloc(p, bytes ? bytes : 16); }
extern "C" long hipemu_launch_count(void) { return hipemu::g_launches; }
extern "C" int osfm_ctx_device(const osfm_ctx *c) { return c ? c->device : -1; }
extern "C" int osfm_ctx_num_cus(const osfm_ctx *c) { return c ? c->num_cus : 0; }
extern "C" const char *osfm_version(void) { return "osfm-mi355 host emulation (tests only)"; }
