// kernels.hip — plan-independent gfx950 kernels compiled ahead of time with hipcc into libqhip.so.
#include <hip/hip_runtime.h>

#include "device/qhip_status.h"
#include "device/qhip_device.hpp"
#include "kernels.hpp"

namespace qhip {

// ---------------------------------------------------------------- group table -> dense slots
// GroupAccumulator::output (physical/plan/aggregate/hash.rs:89-107): one output row per group. Group
// order = slot order, i.e. unspecified, like the reference's HashMap iteration order (hash.rs:98).
__global__ __launch_bounds__(QH_BLOCK) void k_count_ready(const u64* table, u32 nslots, int slot_words, u32* counter) {
  u32 local = 0;
  for (u32 s = blockIdx.x * QH_BLOCK + threadIdx.x; s < nslots; s += gridDim.x * QH_BLOCK)
    local += table[(size_t)s * slot_words] == QH_READY ? 1u : 0u;
  u64 total = qh_wave_sum_u64(local);
  if (qh_lane() == 0 && total) atomicAdd(counter, (u32)total);
}

__global__ __launch_bounds__(QH_BLOCK) void k_compact_slots(const u64* table, u32 nslots, int slot_words, u64* out, u32* counter,
                                                            u32 out_capacity) {
  for (u32 s = blockIdx.x * QH_BLOCK + threadIdx.x; s < nslots; s += gridDim.x * QH_BLOCK) {
    const u64* slot = table + (size_t)s * slot_words;
    if (slot[0] == QH_READY) {
      const u32 idx = atomicAdd(counter, 1u);
      if (idx < out_capacity) {
        u64* o = out + (size_t)idx * slot_words;
        for (int k = 0; k < slot_words; ++k) o[k] = slot[k];
      }
    }
  }
}

static inline unsigned grid_for(uint64_t n, unsigned cap = 2048) {
  uint64_t g = (n + QH_BLOCK - 1) / QH_BLOCK;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

void launch_count_ready(const uint64_t* table, uint32_t nslots, int slot_words, uint32_t* counter, hipStream_t s) {
  hipLaunchKernelGGL(k_count_ready, dim3(grid_for(nslots)), dim3(QH_BLOCK), 0, s, (const u64*)table, nslots, slot_words, counter);
}
void launch_compact_slots(const uint64_t* table, uint32_t nslots, int slot_words, uint64_t* out, uint32_t* counter,
                          uint32_t out_capacity, hipStream_t s) {
  hipLaunchKernelGGL(k_compact_slots, dim3(grid_for(nslots)), dim3(QH_BLOCK), 0, s, (const u64*)table, nslots, slot_words, (u64*)out,
                     counter, out_capacity);
}

}  // namespace qhip
