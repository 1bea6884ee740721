// bench_stream.hip — the plain streaming-read yardstick of bench.py (SURVEY §8d "achievable-copy ceiling"), part of
// libqhip_bench.so (include/qhip_bench.h), not of the product library: reads a scratch HBM buffer with 16-byte-per-lane
// non-temporal loads (the fastest plain reader found on MI355X, tools/stream_sweep.py) and reports GB/s.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "../../include/qhip_bench.h"

namespace {
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;

template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_stream_read(const v4u* p, uint64_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t stride = (uint64_t)gridDim.x * kBlock;
  uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  for (; i + (U - 1) * stride < n16; i += U * stride) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(&p[i + u * stride]) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { const v4u v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u) *sink = acc;   // keeps the loads alive; practically never taken
}

void launch(const void* p, uint64_t bytes, uint32_t* sink, unsigned blocks, hipStream_t s) {
  const char* uv = getenv("QHIP_STREAM_UNROLL");
  const char* nv = getenv("QHIP_STREAM_NT");
  const int U = uv ? atoi(uv) : 1;
  const bool nt = !(nv && *nv == '0');
#define LAUNCH(UU, NN) hipLaunchKernelGGL((k_stream_read<UU, NN>), dim3(blocks), dim3(kBlock), 0, s, (const v4u*)p, (uint64_t)(bytes / 16), sink)
  if (U >= 8) { if (nt) LAUNCH(8, true); else LAUNCH(8, false); }
  else if (U >= 4) { if (nt) LAUNCH(4, true); else LAUNCH(4, false); }
  else if (U >= 2) { if (nt) LAUNCH(2, true); else LAUNCH(2, false); }
  else { if (nt) LAUNCH(1, true); else LAUNCH(1, false); }
#undef LAUNCH
}
}  // namespace

extern "C" int qhip_bench_stream_read(int32_t device, int64_t bytes, int32_t iters, double* gb_per_s) {
  if (!gb_per_s || bytes < (1 << 20) || iters < 1) return 1;
  void* buf = nullptr;
  uint32_t* sink = nullptr;
  hipStream_t s = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 3;
  hipDeviceProp_t prop;
  do {
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) break;
    if (hipMalloc(&buf, (size_t)bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) break;
    if (hipStreamCreate(&s) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
    if (hipMemsetAsync(buf, 1, (size_t)bytes, s) != hipSuccess) break;
    const char* bp = getenv("QHIP_STREAM_BLOCKS_PER_CU");
    const unsigned blocks = (unsigned)prop.multiProcessorCount * (unsigned)std::max(1, bp ? atoi(bp) : 8);
    launch(buf, (uint64_t)bytes, sink, blocks, s);
    if (hipEventRecord(e0, s) != hipSuccess) break;
    for (int k = 0; k < iters; ++k) launch(buf, (uint64_t)bytes, sink, blocks, s);
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) break;
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0) break;
    *gb_per_s = (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    rc = 0;
  } while (false);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (s) (void)hipStreamDestroy(s);
  if (buf) (void)hipFree(buf);
  if (sink) (void)hipFree(sink);
  return rc;
}
