// join.cpp — placeholder, replaced below in this round
#include "common.hpp"
using namespace qhip;
extern "C" int qhip_hash_join_execute(qhip_ctx* ctx, const qhip_table*, const qhip_table*, int32_t, const qhip_expr*, int32_t, const qhip_expr*, int32_t,
                                      const int32_t*, const int32_t*, int32_t, const qhip_expr*, int32_t, int32_t, const int32_t*, const int32_t*, int32_t,
                                      qhip_table** out) {
  if (out) *out = nullptr;
  return guarded(ctx, [&] { fail(QHIP_UNSUPPORTED, "qhip_hash_join_execute: not built yet"); });
}
