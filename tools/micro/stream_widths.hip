// Micro-benchmark (MI355X): read bandwidth of a streaming kernel as a function of the bytes a lane loads per instruction
// (4 / 8 / 16), the loads in flight per wave (U) and WHO reads what: grid-stride (the chip sweeps the buffer front to
// back) vs chunked (every wavefront owns a contiguous chunk, like the join probe's tile chunks).
//   hipcc --offload-arch=gfx950 -O3 -o stream_widths tools/micro/stream_widths.hip && ./stream_widths
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T> __device__ __forceinline__ u32 fold(T v);
template <> __device__ __forceinline__ u32 fold<u32>(u32 v) { return v; }
template <> __device__ __forceinline__ u32 fold<v2u>(v2u v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ u32 fold<v4u>(v4u v) { return v.x ^ v.y ^ v.z ^ v.w; }

// grid-stride: element i of type T by global thread i, then + total threads; U loads issued before any is used
template <class T, int U>
__global__ __launch_bounds__(256) void k_grid(const T* p, u64 n, u32* sink) {
  u32 acc = 0;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&p[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= fold<T>(v[u]);
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
// chunked: wavefront w owns elements [w * chunk, (w + 1) * chunk); per trip it loads U * 64 consecutive elements
template <class T, int U>
__global__ __launch_bounds__(256) void k_chunk(const T* p, u64 n, u64 chunk, u32* sink) {
  u32 acc = 0;
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 first = wave * chunk, last = first + chunk < n ? first + chunk : n;
  for (u64 i = first; i + U * 64 <= last; i += U * 64) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&p[i + u * 64 + lane]);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= fold<T>(v[u]);
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
// two streams like the probe's (8-byte key column + 4-byte date column), chunked, U rows per lane and trip
template <int U>
__global__ __launch_bounds__(256) void k_two(const v2u* a, const u32* b, u64 n, u64 chunk, u32* sink) {
  u32 acc = 0;
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 first = wave * chunk, last = first + chunk < n ? first + chunk : n;
  for (u64 i = first; i + U * 64 <= last; i += U * 64) {
    v2u x[U]; u32 y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = __builtin_nontemporal_load(&a[i + u * 64 + lane]); y[u] = __builtin_nontemporal_load(&b[i + u * 64 + lane]); }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ y[u];
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_two_grid(const v2u* a, const u32* b, u64 n, u32* sink) {
  u32 acc = 0;
  const u64 stride = (u64)gridDim.x * 256;
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
    v2u x[U]; u32 y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = __builtin_nontemporal_load(&a[i + u * stride]); y[u] = __builtin_nontemporal_load(&b[i + u * stride]); }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ y[u];
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

// ... plus X extra dword loads per 64 rows from a small (L2 / L1 resident) table: how many vector-memory INSTRUCTIONS per row
// can a CU issue? (the probe kernel's bitmap / row_of lookups and its stores are such instructions)
template <int U, int X>
__global__ __launch_bounds__(256) void k_two_extra(const v2u* a, const u32* b, const u32* small, u64 n, u64 chunk, u32* sink) {
  u32 acc = 0;
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 first = wave * chunk, last = first + chunk < n ? first + chunk : n;
  u32 prev = lane;
  for (u64 i = first; i + U * 64 <= last; i += U * 64) {
    v2u x[U]; u32 y[U]; u32 z[U * X];
#pragma unroll
    for (int u = 0; u < U * X; ++u) z[u] = small[(prev * 2654435761u + u * 97u) & 2047u];   // addresses from the PREVIOUS trip's data
#pragma unroll
    for (int u = 0; u < U; ++u) { x[u] = __builtin_nontemporal_load(&a[i + u * 64 + lane]); y[u] = __builtin_nontemporal_load(&b[i + u * 64 + lane]); }
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= x[u].x ^ x[u].y ^ y[u];
#pragma unroll
    for (int u = 0; u < U * X; ++u) acc ^= z[u];
    prev = acc;
  }
  if (acc == 0x9e3779b9u) *sink = acc;
}

template <class F> double time_ms(F&& launch, int iters = 5) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();
  CHECK(hipEventRecord(e0));
  for (int k = 0; k < iters; ++k) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  const u64 bytes = 1ull << 30;   // 1 GiB per stream: far beyond the 256 MiB Infinity Cache
  void *buf, *buf2; u32* sink;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&buf2, bytes)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMemset(buf2, 2, bytes));
  int cus = 256; hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0)); cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; 1 GiB per stream, nt loads; GB/s\n", prop.gcnArchName, cus);
#define GRID(T, U, BPC) { double ms = time_ms([&] { hipLaunchKernelGGL((k_grid<T, U>), dim3(cus * BPC), dim3(256), 0, 0, (const T*)buf, bytes / sizeof(T), sink); }); \
    printf("grid-stride  %2zu B/lane  U=%d  %d wg/CU : %7.0f\n", sizeof(T), U, BPC, bytes / ms / 1e6); }
  GRID(u32, 1, 8) GRID(u32, 4, 8) GRID(u32, 8, 8) GRID(v2u, 1, 8) GRID(v2u, 4, 8) GRID(v2u, 8, 8) GRID(v4u, 1, 8) GRID(v4u, 2, 8) GRID(v4u, 4, 8)
  GRID(u32, 4, 4) GRID(v2u, 4, 4) GRID(v4u, 4, 4)
#define CHUNK(T, U, CH) { const u64 n = bytes / sizeof(T); const u64 waves = (n + (CH) - 1) / (CH); const unsigned grid = (unsigned)((waves + 3) / 4); \
    double ms = time_ms([&] { hipLaunchKernelGGL((k_chunk<T, U>), dim3(grid), dim3(256), 0, 0, (const T*)buf, n, (u64)(CH), sink); }); \
    printf("chunked      %2zu B/lane  U=%d  chunk %6d rows (%u wgs): %7.0f\n", sizeof(T), U, (int)(CH), grid, bytes / ms / 1e6); }
  CHUNK(u32, 4, 3072) CHUNK(v2u, 4, 3072) CHUNK(v4u, 4, 3072) CHUNK(u32, 8, 3072) CHUNK(v2u, 8, 3072) CHUNK(v2u, 4, 12288) CHUNK(v2u, 4, 49152) CHUNK(v4u, 4, 49152)
#define TWO(U, CH) { const u64 n = bytes / 8; const u64 waves = (n + (CH) - 1) / (CH); const unsigned grid = (unsigned)((waves + 3) / 4); \
    double ms = time_ms([&] { hipLaunchKernelGGL((k_two<U>), dim3(grid), dim3(256), 0, 0, (const v2u*)buf, (const u32*)buf2, n, (u64)(CH), sink); }); \
    printf("two streams 8+4 B, chunked, U=%d chunk %6d rows (%u wgs): %7.0f\n", U, (int)(CH), grid, n * 12 / ms / 1e6); }
  TWO(4, 3072) TWO(8, 3072) TWO(4, 12288) TWO(12, 3072) TWO(16, 4096)
#define TWOG(U, BPC) { const u64 n = bytes / 8; double ms = time_ms([&] { hipLaunchKernelGGL((k_two_grid<U>), dim3(cus * BPC), dim3(256), 0, 0, (const v2u*)buf, (const u32*)buf2, n, sink); }); \
    printf("two streams 8+4 B, grid-stride, U=%d %d wg/CU: %7.0f\n", U, BPC, n * 12 / ms / 1e6); }
  TWOG(1, 8) TWOG(4, 8) TWOG(8, 8) TWOG(4, 4) TWOG(8, 4)
#define TWOX(U, X, CH) { const u64 n = bytes / 8; const u64 waves = (n + (CH) - 1) / (CH); const unsigned grid = (unsigned)((waves + 3) / 4); \
    double ms = time_ms([&] { hipLaunchKernelGGL((k_two_extra<U, X>), dim3(grid), dim3(256), 0, 0, (const v2u*)buf, (const u32*)buf2, (const u32*)sink2, n, (u64)(CH), sink); }); \
    printf("two streams 8+4 B chunked U=%d + %d small-table dword loads per 64 rows (%d vmem instr per 64 rows): %7.0f\n", U, X, 2 + X, n * 12 / ms / 1e6); }
  u32* sink2; CHECK(hipMalloc((void**)&sink2, 8192)); CHECK(hipMemset(sink2, 0, 8192));
  TWOX(4, 1, 3072) TWOX(4, 2, 3072) TWOX(4, 3, 3072) TWOX(4, 4, 3072) TWOX(4, 6, 3072)
  return 0;
}
