// Micro-benchmark (MI355X): the dense join probe reduced to its memory behaviour, WITHOUT software pipelining — a wavefront
// owns a chunk of tiles; per tile: R x (8-byte key + 4-byte date) nt loads -> predicate + idx -> masked bitmap-word loads
// -> bit test -> masked row_of loads -> ballot-ranked entry stores. Clustered keys (TPC-H lineitem order), 60 M rows.
// Question: what does the plain dependent chain reach at 8 waves per SIMD, against the pipelined kernel's 4.7 TB/s?
//   hipcc --offload-arch=gfx950 -O3 -o probe_like tools/micro/probe_like.hip && ./probe_like
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef long long i64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(u64* key, u32* date, u64 n) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
    const u64 o = i >> 2;
    key[i] = (o >> 3) * 32 + (o & 7) + 1;
    u64 h = i * 0x9E3779B97F4A7C15ULL; h ^= h >> 29;
    date[i] = (u32)(h & 255u);
  }
}
__global__ void k_build(u32* bits, u32* row_of, u64 norders) {
  for (u64 o = (u64)blockIdx.x * 256 + threadIdx.x; o < norders; o += (u64)gridDim.x * 256) {
    if (o % 10 == 0) {
      const u64 idx = (o >> 3) * 32 + (o & 7) + 1;
      atomicOr(&bits[idx >> 5], 1u << (idx & 31));
      row_of[idx] = (u32)o;
    }
  }
}

template <int R, int MODE>   // MODE 0: full; 1: no row_of loads / stores (count only); 2: columns + predicate only
__global__ __launch_bounds__(256) void k_probe(const u64* __restrict__ key, const u32* __restrict__ date, const u32* __restrict__ bits,
                                               const u32* __restrict__ row_of, u64 n, u32 tiles_per_wave, u32 dense_n, u32* ent_slot, u32* ent_row,
                                               u32* chunk_nent) {
  constexpr int TILE = 64 * R;
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 ntiles = n / TILE;
  const u64 first = wave * tiles_per_wave;
  u32 nent = 0;
  for (u64 t = first; t < first + tiles_per_wave && t < ntiles; ++t) {
    const u64 tb = t * TILE;
    u64 k[R]; u32 d[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { k[r] = __builtin_nontemporal_load(&key[tb + r * 64 + lane]); d[r] = __builtin_nontemporal_load(&date[tb + r * 64 + lane]); }
    bool ok[R]; u32 idx[R], bw[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u64 i = k[r] - 1;
      ok[r] = d[r] > 117u && i < (u64)dense_n;
      idx[r] = ok[r] ? (u32)i : 0u;
    }
    if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < R; ++r) nent += (u32)__builtin_popcountll(__ballot(ok[r]));
      continue;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) bw[r] = bits[idx[r] >> 5];
#pragma unroll
    for (int r = 0; r < R; ++r) ok[r] = ok[r] && ((bw[r] >> (idx[r] & 31)) & 1u);
    if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) nent += (u32)__builtin_popcountll(__ballot(ok[r]));
      continue;
    }
    u32 row[R];
    if (MODE == 4) {
#pragma unroll
      for (int r = 0; r < R; ++r) row[r] = idx[r];                       // no row_of loads, stores as in the full kernel
    } else if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < R; ++r) { row[r] = 0; if (ok[r]) row[r] = row_of[idx[r]]; }   // loads only by the lanes with a hit (EXEC-masked)
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) row[r] = row_of[ok[r] ? idx[r] : 0u];
    }
    if (MODE == 3) {   // row_of loads, no stores
#pragma unroll
      for (int r = 0; r < R; ++r) nent += (u32)__builtin_popcountll(__ballot(ok[r] && row[r] != 0xFFFFFFFFu));
      continue;
    }
    if (MODE == 6) {   // ONE 8-byte store per entry into an interleaved (slot, row) array
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const u64 m = __ballot(ok[r]);
        if (ok[r]) {
          const u64 pos = first * TILE + nent + (u32)__builtin_popcountll(m & ((1ULL << lane) - 1));
          ((u64*)ent_slot)[pos] = ((u64)(u32)(tb + r * 64 + lane) << 32) | row[r];
        }
        nent += (u32)__builtin_popcountll(m);
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const u64 m = __ballot(ok[r]);
      if (ok[r]) {
        const u64 pos = first * TILE + nent + (u32)__builtin_popcountll(m & ((1ULL << lane) - 1));
        ent_slot[pos] = row[r];
        ent_row[pos] = (u32)(tb + r * 64 + lane);
      }
      nent += (u32)__builtin_popcountll(m);
    }
  }
  if (lane == 0) chunk_nent[wave] = nent;
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
  const u64 n = 59986052ull / 256 * 256;
  const u64 norders = n / 4 + 1;
  const u32 dense_n = (u32)((norders >> 3) * 32 + 40);
  u64* key; u32 *date, *bits, *row_of, *ent_slot, *ent_row, *cn;
  CHECK(hipMalloc(&key, n * 8)); CHECK(hipMalloc(&date, n * 4)); CHECK(hipMalloc(&bits, (size_t)dense_n / 8 + 64)); CHECK(hipMalloc(&row_of, (size_t)dense_n * 4 + 64));
  CHECK(hipMalloc(&ent_slot, n * 8)); CHECK(hipMalloc(&ent_row, n * 4)); CHECK(hipMalloc(&cn, 4 << 20));
  CHECK(hipMemset(bits, 0, (size_t)dense_n / 8 + 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, key, date, n);
  hipLaunchKernelGGL(k_build, dim3(2048), dim3(256), 0, 0, bits, row_of, norders);
  CHECK(hipDeviceSynchronize());
  printf("%llu rows, key range %u (bitmap %.1f MB), 12 B/row; GB/s of column bytes\n", (unsigned long long)n, dense_n, dense_n / 8e6);
#define RUN(R, MODE, TPW) { const u64 ntiles = n / (64 * R); const u64 waves = (ntiles + (TPW) - 1) / (TPW); const unsigned grid = (unsigned)((waves + 3) / 4); \
    double ms = time_ms([&] { hipLaunchKernelGGL((k_probe<R, MODE>), dim3(grid), dim3(256), 0, 0, key, date, bits, row_of, n, (u32)(TPW), dense_n, ent_slot, ent_row, cn); }); \
    printf("R=%d mode %d (%s) tiles/wave %3d, %6u wgs: %7.1f us  %6.0f GB/s\n", R, MODE, MODE == 0 ? "full" : MODE == 1 ? "no row_of / stores" : MODE == 2 ? "columns + predicate" : MODE == 3 ? "row_of loads, no stores" : MODE == 4 ? "stores, no row_of loads" : MODE == 5 ? "full, row_of loads EXEC-masked" : "full, one 8-byte store per entry", (int)(TPW), grid, ms * 1e3, n * 12 / ms / 1e6); }
  RUN(4, 3, 12) RUN(4, 4, 12) RUN(4, 5, 12) RUN(4, 6, 12)
  RUN(4, 2, 12) RUN(4, 1, 12) RUN(4, 0, 12) RUN(4, 0, 29) RUN(4, 0, 6) RUN(2, 0, 24) RUN(2, 0, 58) RUN(8, 0, 6) RUN(8, 0, 15) RUN(1, 0, 48) RUN(4, 2, 29) RUN(8, 2, 15)
  u32 h[4]; CHECK(hipMemcpy(h, cn, 16, hipMemcpyDeviceToHost));
  printf("(first chunk counts %u %u)\n", h[0], h[1]);
  return 0;
}
