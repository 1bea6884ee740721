// kargs_host.hpp — host mirrors of the kernel-argument structs declared in device/qhip_device.hpp.
// Layouts must match byte for byte (both sides are LP64 with natural alignment); static_asserts below
// pin the sizes the device header produces.
#pragma once
#include <cstdint>

namespace qhip {

constexpr int kMaxCols = 24;   // QH_MAXC
constexpr int kMaxLits = 24;   // QH_MAXL

struct HKCol {
  const void* v;
  const uint8_t* n;
  const uint8_t* d;
};
struct HKArgs {
  HKCol c[kMaxCols];
  uint64_t lit_lo[kMaxLits];
  int64_t lit_hi[kMaxLits];
  const uint8_t* strlit;
  int stroff[kMaxLits + 1];
  int64_t nrows;
  const uint32_t* nrows_dev;
};
static_assert(sizeof(HKCol) == 24, "KCol layout");
static_assert(sizeof(HKArgs) == 24 * 24 + 8 * 24 + 8 * 24 + 8 + 104 + 8 + 8, "KArgs layout");

struct HAggLaunch {
  uint64_t* gtable;
  uint32_t g_nslots;
  uint32_t l_nslots;
  uint32_t* status;
  uint32_t replicas;
  uint32_t collect_stats;
  const uint32_t* part_runs = nullptr;   // PARTS form (qk_filter_agg_parts)
  uint64_t* dense_out = nullptr;
  uint32_t* dense_counter = nullptr;
  uint32_t part_stride = 0, dense_cap = 0;
  uint32_t n_parts = 0, part_max = 0;
};
static_assert(sizeof(HAggLaunch) == 72, "AggLaunch layout");

// host mirrors of PartLaunch / ReduceLaunch (the aggregate's partitioned path)
struct HPartLaunch {
  uint32_t* hist;
  uint64_t* records;
  uint32_t* status;
  uint32_t n_bins;
  uint32_t rows_per_wg;
};
static_assert(sizeof(HPartLaunch) == 32, "PartLaunch layout");
struct HReduceLaunch {
  const uint64_t* records;
  const uint32_t* item_first;
  uint32_t n_items;
  uint32_t slices = 1;
  const uint32_t* hist = nullptr;   // device-side work items (item_first == nullptr)
  uint32_t g1 = 0, n_bins = 0, min_slice = 0, pad_ = 0;
};
static_assert(sizeof(HReduceLaunch) == 48, "ReduceLaunch layout");

struct HRunsLaunch {
  uint64_t* dense_out;
  uint32_t* counter;
  uint32_t* status;
  uint32_t* flags;
  uint32_t cap, max_run;
  uint32_t lds_bytes = 0, pad_ = 0;
};
static_assert(sizeof(HRunsLaunch) == 48, "RunsLaunch layout");

struct HProjOut {
  void* v[kMaxCols];
  uint64_t* n[kMaxCols];
  uint8_t* d[kMaxCols];
};
static_assert(sizeof(HProjOut) == 3 * 8 * 24, "ProjOut layout");

struct HProbeLaunch {
  const uint64_t* table;
  const uint64_t* bloom;
  const uint32_t* count;
  const uint32_t* start;
  const uint32_t* rows;
  uint32_t* ent_slot;
  uint32_t* ent_row;
  uint32_t* tile_nent;
  uint32_t* tile_total;
  uint32_t* visited;
  uint32_t* status;
  uint32_t nslots, bloom_mask;
  uint32_t n_regions = 0, slot_bits = 0, bword_bits = 0, stage_cap = 0;   // region layout of the LDS-staged build (0 = legacy)
  uint32_t tiles_per_wave = 1, lds_words = 0;
  uint64_t dense_min = 0;                        // dense (direct-address) layout: bloom = exact bitmap, table = u32 row_of[]
  uint32_t dense_n = 0, dense_words = 0;
};
static_assert(sizeof(HProbeLaunch) == 11 * 8 + 8 + 16 + 8 + 16, "ProbeLaunch layout");

struct HDenseBuildLaunch {
  uint32_t* bits;
  uint32_t* row_of;
  uint32_t* status;
  uint64_t kmin;
  uint32_t n;
  uint32_t gen = 0;
  uint8_t* bytes = nullptr;
  uint32_t* counters = nullptr;
};
static_assert(sizeof(HDenseBuildLaunch) == 56, "DenseBuildLaunch layout");

struct HPartIdsLaunch {
  uint8_t* ids;
  uint32_t* hist;
  uint32_t* status;
  uint32_t n_units;
  uint32_t rows_per_unit;
  uint32_t wg_units;
  uint32_t pad_;
  const int64_t* bounds = nullptr;
};
static_assert(sizeof(HPartIdsLaunch) == 48, "PartIdsLaunch layout");

constexpr int kPartMaxCols = 8;   // QH_PART_MAXC
struct HPartScatterLaunch {
  const uint8_t* ids;
  const uint32_t* runs;
  int64_t nrows;
  const uint32_t* nrows_dev;
  uint32_t n_units, rows_per_unit, n_parts, sub;
  void* trash;
  const void* src[kPartMaxCols];
  const uint32_t* idx[kPartMaxCols];
  void* out[kPartMaxCols];
};
static_assert(sizeof(HPartScatterLaunch) == 32 + 16 + 8 + 3 * 8 * 8, "PartScatterLaunch layout");

struct HScatterLaunch {
  uint64_t* entries;
  uint32_t* first;
  uint32_t* status;
  uint32_t n_regions;
  uint32_t rows_per_wg;
};
static_assert(sizeof(HScatterLaunch) == 32, "ScatterLaunch layout");

}  // namespace qhip
