// kernels.hip — plan-independent gfx950 kernels compiled ahead of time with hipcc into libqhip.so.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device/qhip_status.h"
#include "device/qhip_device.hpp"
#include "kernels.hpp"

namespace qhip {

// ---------------------------------------------------------------- group table -> dense slots
// GroupAccumulator::output (physical/plan/aggregate/hash.rs:89-107): one output row per group. Group
// order = slot order, i.e. unspecified, like the reference's HashMap iteration order (hash.rs:98).
// Every wavefront owns a contiguous range of slots: it counts the ready ones (state words only), reserves its share of
// the output with ONE atomic, then copies. (Atomics on one address cost ~10 ns each across the 8 XCDs: one per group
// or even one per 64 slots would dominate the kernel.)
// out_host (optional): page-locked HOST memory that receives the first cap_host dense slots as well — a result of few groups
// lands where the host reads it without a device-to-host copy behind the kernel (a 26 KB copy goes through the DMA engine:
// ~25 us of stream time with its hand-over gaps, on the critical path of TPC-H Q1's step)
__global__ __launch_bounds__(QH_BLOCK) void k_compact_slots(const u64* table, u32 nslots, int slot_words, u64* out, u32* counter,
                                                            u32 out_capacity, u64* out_host, u32 cap_host) {
  const int lane = qh_lane();
  const u32 nwaves = gridDim.x * (QH_BLOCK / 64), wave = blockIdx.x * (QH_BLOCK / 64) + (threadIdx.x >> 6);
  const u32 per_wave = ((nslots + nwaves - 1) / nwaves + 63) / 64 * 64;
  const u64 lo = (u64)wave * per_wave, hi = lo + per_wave < (u64)nslots ? lo + per_wave : (u64)nslots;
  // ONE atomic per workgroup: the wavefronts' totals meet in LDS, wavefront 0 reserves the workgroup's share and every
  // wavefront starts behind its predecessors' (a 262 k-slot table: 256 same-address atomics instead of 1 024)
  __shared__ u32 wtot[QH_BLOCK / 64];
  __shared__ u32 wg_base;
  const u32 w = threadIdx.x >> 6;
  u32 mine = 0;
  if (lo < hi) for (u64 s = lo + lane; s < hi; s += 64) mine += table[(size_t)s * slot_words] == QH_READY ? 1u : 0u;
  const u32 total = (u32)qh_wave_sum_u64(mine);
  if (lane == 0) wtot[w] = total;
  __syncthreads();
  if (threadIdx.x == 0) {
    u32 all = 0;
    for (int k = 0; k < QH_BLOCK / 64; ++k) all += wtot[k];
    wg_base = all ? atomicAdd(counter, all) : 0u;
  }
  __syncthreads();
  if (lo >= hi || !total) return;
  u32 base = wg_base;
  for (u32 k = 0; k < w; ++k) base += wtot[k];
  for (u64 s0 = lo; s0 < hi; s0 += 64) {
    const u64 s = s0 + lane;
    const u64* slot = table + (size_t)s * slot_words;
    const bool ready = s < hi && slot[0] == QH_READY;
    const u64 m = qh_ballot(ready);
    if (ready) {
      const u32 idx = base + (u32)__popcll(m & ((1ULL << lane) - 1));
      if (idx < out_capacity) {
        u64* o = out + (size_t)idx * slot_words;
        for (int k = 0; k < slot_words; ++k) o[k] = slot[k];
      }
      if (idx < cap_host) {
        u64* o = out_host + (size_t)idx * slot_words;
        for (int k = 0; k < slot_words; ++k) o[k] = slot[k];
      }
    }
    base += (u32)__popcll(m);
  }
}

static inline unsigned grid_for(uint64_t n, unsigned cap = 2048) {
  uint64_t g = (n + QH_BLOCK - 1) / QH_BLOCK;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

void launch_compact_slots(const uint64_t* table, uint32_t nslots, int slot_words, uint64_t* out, uint32_t* counter,
                          uint32_t out_capacity, hipStream_t s, uint64_t* out_host, uint32_t cap_host) {
  // few, long-lived wavefronts: 64 .. 2048 of them, each with >= 256 slots
  const unsigned blocks = (unsigned)std::max<uint64_t>(16, std::min<uint64_t>(512, ((uint64_t)nslots + 1023) / 1024));
  hipLaunchKernelGGL(k_compact_slots, dim3(blocks), dim3(QH_BLOCK), 0, s, (const u64*)table, nslots, slot_words, (u64*)out, counter,
                     out_capacity, (u64*)out_host, out_host ? cap_host : 0u);
}

}  // namespace qhip
