// Micro-benchmark (MI355X): what the chip delivers for the traffic MIX of the exchange's pass 2 — read N x 16-byte values,
// write a fraction f of them (here: every lane writes its value when (i * 0x9E3779B1) >> 32-bit hash < f, compacted per
// wavefront with a ballot so the writes are dense) — against pure read, pure write and a plain copy, with plain and
// non-temporal stores. The ceiling the partition kernel's roofline fraction should be read against.
//   hipcc --offload-arch=gfx950 -O3 -o copy_mix tools/micro/copy_mix.hip && ./copy_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 v4u __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// wavefront w owns elements [w * chunk, (w + 1) * chunk): reads them all (U x 64 per trip), keeps those whose hash passes
// (keep_per_256 of 256), writes the kept ones densely from out + w * chunk on (MODE 0 plain stores, 1 non-temporal)
template <int U, int MODE, bool READ, bool WRITE>
__global__ __launch_bounds__(256) void k_mix(const v4u* in, v4u* out, u64 n, u64 chunk, u32 keep_per_256, u32* sink) {
  const u64 wave = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6;
  const u32 lane = threadIdx.x & 63;
  const u64 first = wave * chunk, last = first + chunk < n ? first + chunk : n;
  u64 at = first;
  u32 acc = 0;
  for (u64 i = first; i + U * 64 <= last; i += U * 64) {
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (READ) v[u] = __builtin_nontemporal_load(&in[i + u * 64 + lane]);
      else { v[u].x = (u32)i; v[u].y = lane; v[u].z = u; v[u].w = 7; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u64 row = i + u * 64 + lane;
      const bool keep = (((u32)row * 0x9E3779B1u) >> 24) < keep_per_256;
      const u64 m = __ballot(keep);
      if (WRITE) {
        const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0));
        if (keep) { if (MODE == 1) __builtin_nontemporal_store(v[u], &out[at + rank]); else out[at + rank] = v[u]; }
        at += __builtin_popcountll(m);
      } else acc ^= v[u].x ^ v[u].w;
    }
  }
  if (!WRITE && acc == 0x9e3779b9u) *sink = acc;
}

template <int U, int MODE, bool READ, bool WRITE>
static void run(const char* name, const v4u* in, v4u* out, u64 n, u32 keep, u64 chunk, u32* sink) {
  const u64 waves = (n + chunk - 1) / chunk;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL((k_mix<U, MODE, READ, WRITE>), dim3(grid), dim3(256), 0, 0, in, out, n, chunk, keep, sink);
  CHECK(hipEventRecord(a));
  const int it = 10;
  for (int k = 0; k < it; ++k) hipLaunchKernelGGL((k_mix<U, MODE, READ, WRITE>), dim3(grid), dim3(256), 0, 0, in, out, n, chunk, keep, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= it;
  const double rd = READ ? (double)n * 16 : 0, wr = WRITE ? (double)n * 16 * keep / 256.0 : 0;
  printf("%-44s chunk %7llu rows  %.3f ms  read %.2f GB + written %.2f GB -> %.2f TB/s\n", name, (unsigned long long)chunk, ms, rd / 1e9, wr / 1e9,
         (rd + wr) / (ms * 1e-3) / 1e12);
}

int main() {
  const u64 n = 150ull << 20;   // 157 M x 16 B = 2.5 GB
  v4u *in, *out;
  u32* sink;
  CHECK(hipMalloc(&in, n * 16)); CHECK(hipMalloc(&out, n * 16)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(in, 1, n * 16)); CHECK(hipMemset(out, 0, n * 16));
  for (u64 chunk : {4096ull, 16384ull, 65536ull, 262144ull}) {
    run<4, 0, true, false>("read only", in, out, n, 0, chunk, sink);
    run<4, 0, false, true>("write only, all rows", in, out, n, 256, chunk, sink);
    run<4, 1, false, true>("write only, all rows, non-temporal", in, out, n, 256, chunk, sink);
    run<4, 0, true, true>("copy (keep all)", in, out, n, 256, chunk, sink);
    run<4, 1, true, true>("copy (keep all), non-temporal stores", in, out, n, 256, chunk, sink);
    run<4, 0, true, true>("read all, write 54 %", in, out, n, 138, chunk, sink);
    run<4, 1, true, true>("read all, write 54 %, non-temporal stores", in, out, n, 138, chunk, sink);
    run<8, 0, true, true>("read all, write 54 %, U = 8", in, out, n, 138, chunk, sink);
  }
  return 0;
}
