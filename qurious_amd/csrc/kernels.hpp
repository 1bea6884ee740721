// kernels.hpp — host launchers of the plan-independent (ahead-of-time compiled) kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

namespace qhip {

// group table -> dense slot array
void launch_count_ready(const uint64_t* table, uint32_t nslots, int slot_words, uint32_t* counter, hipStream_t s);
void launch_compact_slots(const uint64_t* table, uint32_t nslots, int slot_words, uint64_t* out, uint32_t* counter,
                          uint32_t out_capacity, hipStream_t s);

}  // namespace qhip
