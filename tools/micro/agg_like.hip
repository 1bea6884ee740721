// Micro-benchmark (MI355X): WHO reads WHICH rows in a persistent scan-aggregate kernel — the access patterns the fused
// filter + aggregate kernel could use over TPC-H Q1's columns (4 x 16 B decimals + a 4 B date + 2 x 1 B flags = 70 B per row),
// with next to no arithmetic, so that only the pattern is measured.
//   wgtile   workgroup g takes tiles g, g + G, g + 2G, ... of 256 R rows; thread tid reads rows r * 256 + tid of its tile
//            (what qh_filter_agg_body did up to round 3)
//   wavetile the same, but a tile belongs to a WAVEFRONT (64 R rows; rows r * 64 + lane), tiles strided over all wavefronts
//   wavechunk wavefront w owns `chunk` consecutive rows at a time (chunks c = w, w + W, ...), walks a chunk front to back in
//            trips of 64 R rows
// each with R rows per lane in flight, B workgroups per CU (persistent), and with the next trip's loads issued before the
// current trip's values are consumed (PIPE).
//   hipcc --offload-arch=gfx950 -O3 -o agg_like tools/micro/agg_like.hip && ./agg_like
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;
typedef unsigned char u8;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Cols { const v4u *a, *b, *c, *d; const u32* date; const u8 *f1, *f2; };
struct Raw { v4u a, b, c, d; u32 date; u8 f1, f2; };
template <int NCOL> __device__ __forceinline__ void load(const Cols& t, i64 row, Raw& w) {
  w.a = __builtin_nontemporal_load(&t.a[row]);
  if (NCOL > 1) {
    w.b = __builtin_nontemporal_load(&t.b[row]); w.c = __builtin_nontemporal_load(&t.c[row]); w.d = __builtin_nontemporal_load(&t.d[row]);
    w.date = __builtin_nontemporal_load(&t.date[row]); w.f1 = __builtin_nontemporal_load(&t.f1[row]); w.f2 = __builtin_nontemporal_load(&t.f2[row]);
  }
}
template <int NCOL> __device__ __forceinline__ u32 fold(const Raw& w) {
  u32 x = w.a.x ^ w.a.y ^ w.a.z ^ w.a.w;
  if (NCOL > 1) x ^= w.b.x ^ w.b.y ^ w.b.z ^ w.b.w ^ w.c.x ^ w.c.y ^ w.c.z ^ w.c.w ^ w.d.x ^ w.d.y ^ w.d.z ^ w.d.w ^ w.date ^ (u32)w.f1 ^ ((u32)w.f2 << 8);
  return x;
}

// MODE 0 wgtile, 1 wavetile, 2 wavechunk
template <int MODE, int R, int NCOL, bool PIPE>
__global__ __launch_bounds__(256) void k_scan(Cols t, i64 nrows, i64 chunk, u32* sink) {
  u32 acc = 0;
  const int tid = threadIdx.x, lane = tid & 63;
  const i64 wave = (i64)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = (i64)gridDim.x * 4;
  // the next trip of this thread -> first row of its R-row group and the row stride inside the group; false behind the end
  i64 k = 0, c = wave, j = 0;
  const i64 trips_per_chunk = MODE == 2 ? chunk / (64 * R) : 1;
  auto where = [&](i64& base, int& stride) -> bool {
    if (MODE == 0) { base = ((i64)blockIdx.x + k * gridDim.x) * (256 * R) + tid; stride = 256; }
    else if (MODE == 1) { base = (wave + k * nwaves) * (64 * R) + lane; stride = 64; }
    else {
      base = c * chunk + j * (64 * R) + lane; stride = 64;
      if (++j == trips_per_chunk) { j = 0; c += nwaves; }
    }
    ++k;
    return base + (i64)(R - 1) * stride < nrows;
  };
  if (!PIPE) {
    for (;;) {
      i64 base; int stride;
      if (!where(base, stride)) break;
      Raw w[R];
#pragma unroll
      for (int r = 0; r < R; ++r) load<NCOL>(t, base + (i64)r * stride, w[r]);
#pragma unroll
      for (int r = 0; r < R; ++r) acc ^= fold<NCOL>(w[r]);
    }
  } else {
    Raw A[R], B[R];
    i64 base; int stride;
    bool liveA = where(base, stride), liveB;
#pragma unroll
    for (int r = 0; r < R; ++r) load<NCOL>(t, liveA ? base + (i64)r * stride : 0, A[r]);
    while (liveA) {
      liveB = where(base, stride);
#pragma unroll
      for (int r = 0; r < R; ++r) load<NCOL>(t, liveB ? base + (i64)r * stride : 0, B[r]);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc ^= fold<NCOL>(A[r]);
      if (!liveB) break;
      liveA = where(base, stride);
#pragma unroll
      for (int r = 0; r < R; ++r) load<NCOL>(t, liveA ? base + (i64)r * stride : 0, A[r]);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < R; ++r) acc ^= fold<NCOL>(B[r]);
    }
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
  const i64 nrows = 60000000 / 12288 * 12288;   // whole chunks / tiles for every shape below
  Cols t; u32* sink;
  void* p[7]; const size_t w[7] = {16, 16, 16, 16, 4, 1, 1};
  for (int k = 0; k < 7; ++k) { CHECK(hipMalloc(&p[k], (size_t)nrows * w[k] + 4096)); CHECK(hipMemset(p[k], k + 1, (size_t)nrows * w[k])); }
  t.a = (const v4u*)p[0]; t.b = (const v4u*)p[1]; t.c = (const v4u*)p[2]; t.d = (const v4u*)p[3]; t.date = (const u32*)p[4]; t.f1 = (const u8*)p[5]; t.f2 = (const u8*)p[6];
  CHECK(hipMalloc(&sink, 4));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; %lld rows; GB/s\n", prop.gcnArchName, cus, (long long)nrows);
#define RUN(MODE, R, NCOL, PIPE, BPC, CHUNK) { const double bytes = (double)nrows * (NCOL > 1 ? 70 : 16); \
    double ms = time_ms([&] { hipLaunchKernelGGL((k_scan<MODE, R, NCOL, PIPE>), dim3(cus * BPC), dim3(256), 0, 0, t, nrows, (i64)(CHUNK), sink); }); \
    printf("%-9s %s R=%d pipe=%d %d wg/CU chunk %6d : %7.0f\n", MODE == 0 ? "wgtile" : MODE == 1 ? "wavetile" : "wavechunk", NCOL > 1 ? "7 cols 70 B" : "1 col  16 B", R, (int)PIPE, BPC, (int)(CHUNK), bytes / ms / 1e6); }
  // one 16-byte column
  RUN(0, 4, 1, false, 4, 0) RUN(0, 4, 1, false, 8, 0) RUN(0, 1, 1, false, 8, 0) RUN(0, 2, 1, false, 8, 0) RUN(0, 4, 1, true, 4, 0)
  RUN(1, 4, 1, false, 4, 0) RUN(1, 4, 1, false, 8, 0) RUN(1, 1, 1, false, 8, 0)
  RUN(2, 4, 1, false, 4, 3072) RUN(2, 4, 1, false, 8, 3072) RUN(2, 4, 1, false, 4, 12288) RUN(2, 4, 1, true, 4, 3072) RUN(2, 2, 1, true, 4, 3072) RUN(2, 2, 1, false, 8, 3072)
  // Q1's seven columns
  RUN(0, 3, 7, false, 4, 0) RUN(0, 2, 7, false, 4, 0) RUN(0, 1, 7, false, 4, 0) RUN(0, 1, 7, false, 8, 0) RUN(0, 2, 7, false, 8, 0)
  RUN(0, 3, 7, true, 4, 0) RUN(0, 2, 7, true, 4, 0) RUN(0, 1, 7, true, 4, 0) RUN(0, 1, 7, true, 8, 0)
  RUN(1, 3, 7, false, 4, 0) RUN(1, 2, 7, false, 4, 0) RUN(1, 1, 7, false, 8, 0) RUN(1, 2, 7, true, 4, 0) RUN(1, 1, 7, true, 4, 0)
  RUN(2, 3, 7, false, 4, 3072) RUN(2, 2, 7, false, 4, 3072) RUN(2, 1, 7, false, 4, 3072) RUN(2, 1, 7, false, 8, 3072) RUN(2, 2, 7, false, 4, 12288)
  RUN(2, 3, 7, true, 4, 3072) RUN(2, 2, 7, true, 4, 3072) RUN(2, 1, 7, true, 4, 3072) RUN(2, 1, 7, true, 8, 3072) RUN(2, 2, 7, true, 4, 12288) RUN(2, 2, 7, true, 3, 3072)
  return 0;
}
