// exchange.cpp — placeholder, replaced below in this round
#include "common.hpp"
using namespace qhip;
extern "C" {
int qhip_partition_by_key(qhip_ctx* ctx, const qhip_table*, const qhip_expr*, int32_t, const int32_t*, int32_t, int32_t, qhip_table**) {
  return guarded(ctx, [&] { fail(QHIP_UNSUPPORTED, "qhip_partition_by_key: not built yet"); });
}
int qhip_table_concat(qhip_ctx* ctx, const qhip_table* const*, int32_t, qhip_table** out) {
  if (out) *out = nullptr;
  return guarded(ctx, [&] { fail(QHIP_UNSUPPORTED, "qhip_table_concat: not built yet"); });
}
int qhip_table_column_buffer(const qhip_table*, int64_t, int32_t, void**, int64_t*) { return QHIP_UNSUPPORTED; }
int qhip_table_from_device(qhip_ctx* ctx, const char* const*, const qhip_device_column*, int32_t, int64_t, qhip_table** out) {
  if (out) *out = nullptr;
  return guarded(ctx, [&] { fail(QHIP_UNSUPPORTED, "qhip_table_from_device: not built yet"); });
}
}
