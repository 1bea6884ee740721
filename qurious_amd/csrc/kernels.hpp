// kernels.hpp — host launchers of the plan-independent (ahead-of-time compiled) kernels in kernels.hip / kernels_rel.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>

namespace qhip {

constexpr uint32_t kNullIdx = 0xFFFFFFFFu;   // NULL row index in u32 index vectors (tables hold < 2^32 - 1 rows)

// group table -> dense slot array (kernels.hip)
void launch_compact_slots(const uint64_t* table, uint32_t nslots, int slot_words, uint64_t* out, uint32_t* counter,
                          uint32_t out_capacity, hipStream_t s, uint64_t* out_host = nullptr, uint32_t cap_host = 0);   // out_host: page-locked host memory for the first cap_host slots

// scan / selection / gather (kernels_rel.hip)
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* total_dev, hipStream_t s);
void launch_select_indices(const uint64_t* mask, const uint32_t* wave_offset, uint64_t nrows, uint32_t* sel, hipStream_t s);
void launch_mask_prefix_at(const uint64_t* mask, const uint32_t* wave_offset, const uint64_t* rows, uint32_t n, uint64_t nrows, uint32_t total,
                           uint32_t* out, hipStream_t s);
void launch_mask_from_bits(const uint32_t* bits, uint64_t nrows, int want_set, uint64_t* mask, uint32_t* wave_count, hipStream_t s);
void launch_gather_fixed(const void* in, const uint32_t* idx, void* out, uint64_t m, int width, hipStream_t s);
// out[wave_offset[j] + rank] = in[64 j + lane] for the set bits of mask word j (mask-driven compaction, row order kept)
void launch_compact_fixed(const void* in, const uint64_t* mask, const uint32_t* wave_offset, void* out, uint64_t nrows, int width, hipStream_t s);
void launch_gather_bits(const uint8_t* bitmap, const uint32_t* idx, uint64_t m, uint64_t* out_words, uint32_t* set_count, hipStream_t s);
void launch_gather_utf8_lengths(const int32_t* offsets, const uint32_t* idx, uint64_t m, uint32_t* out_len, hipStream_t s);
void launch_gather_utf8_bytes(const int32_t* offsets, const uint8_t* data, const uint32_t* idx, uint64_t m, const uint32_t* out_off,
                              uint8_t* out_data, hipStream_t s);
// out[0] = max |v| of the 63-bit values of an Int64 (words 1) / Decimal128 (words 2) column, out[1] != 0: some value is wider
// (narrow32: when set, value i's low 4 bytes are written there in the same pass — the speculative narrow copy)
void launch_value_maxabs(const void* values, uint64_t n, int words, uint64_t* out, hipStream_t s, uint32_t* narrow32 = nullptr);
// out[i] = the low 4 / 8 bytes of the 16-byte value i (the narrow copy of a Decimal128 column whose values fit, DevColumn::narrow)
// (src_words = 1: the source is an Int64 column, bytes = 4)
void launch_narrow_decimal(const void* values, uint64_t n, int bytes, void* out, hipStream_t s, int src_words = 2);
void launch_pack_field(const void* values, uint64_t n, int width, void* records, uint32_t stride, uint32_t offset, hipStream_t s);
// out[0] / out[1] (zero-filled u64 words): order-preserving images of the max / the complement of the min of an integer column
void launch_value_range(const void* values, uint64_t n, int width, bool is_signed, uint64_t* out, hipStream_t s);
void launch_store_u32(uint32_t* p, uint32_t v, hipStream_t s);
void launch_utf8_max_len(const int32_t* offsets, uint64_t n, uint32_t* out, hipStream_t s);
void launch_iota_u32(uint32_t* out, uint64_t n, hipStream_t s, uint32_t first = 0);
void launch_iota_stride_u32(uint32_t* out, uint64_t n, uint32_t stride, hipStream_t s);   // out[i] = i * stride
void launch_fill_u32(uint32_t* out, uint64_t n, uint32_t v, hipStream_t s);
void launch_lookup_u32(const uint32_t* off, const uint64_t* rows, uint32_t n, uint64_t nrows, uint32_t total, uint32_t* out, hipStream_t s);

// hash join (kernels_rel.hip)
void launch_join_build_insert(int W, const uint64_t* keys, const uint64_t* keyvalid, uint64_t n, uint64_t* table, uint32_t nslots,
                              uint32_t* row_slot, uint32_t* extra, uint64_t* bloom, uint32_t bloom_mask, uint32_t* status, hipStream_t s);
// LDS-staged build, step 2: one workgroup per region collects the region's entries of every step-1 workgroup and stores the
// region's open-addressing image + filter slice
void launch_join_region_build(int W, const uint64_t* entries, const uint32_t* first, uint32_t n_wgs, uint32_t rows_per_wg, uint64_t* table,
                              uint64_t* bloom, uint32_t n_regions, uint32_t slot_bits, uint32_t bword_bits, uint32_t* status, hipStream_t s);
void launch_join_full_counts(int W, const uint64_t* table, uint32_t nslots, uint32_t* count, hipStream_t s);
void launch_add_i32(int32_t* p, uint64_t n, int32_t delta, hipStream_t s);
// p[i] += shifts[b] for the batch b with starts[b] <= i < starts[b + 1] (starts: nb + 1 ascending row numbers)
void launch_add_i32_batched(int32_t* p, uint64_t n, const uint64_t* starts, const int32_t* shifts, uint32_t nb, hipStream_t s);
// dst (zeroed) bits [dst_pos, dst_pos + nbits) |= src bits [0, nbits); src == nullptr appends ones
void launch_bits_append(uint32_t* dst, uint64_t dst_pos, const uint8_t* src, uint64_t nbits, hipStream_t s);
void launch_pair_indices(uint32_t* minor, uint32_t* major, uint64_t n, uint32_t n_minor, uint32_t minor0, uint32_t major0, int unused, hipStream_t s);
void launch_count_bits(const uint64_t* words, uint64_t nbits, uint32_t* out, hipStream_t s);
void launch_sort_gather_img(const uint64_t* img, const uint32_t* idx, uint64_t n, uint64_t flip, uint64_t* out, hipStream_t s);
void launch_sort_gather_valid(const uint64_t* validwords, const uint32_t* idx, uint64_t n, bool nulls_first, uint64_t* out, hipStream_t s);
void launch_sort_utf8_chunk(const int32_t* offsets, const uint8_t* data, const uint8_t* validity, const uint32_t* idx, uint64_t n, int chunk,
                            uint64_t flip, uint64_t* out, hipStream_t s);
void stable_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint64_t n, int bits,
                           hipStream_t s);
// several fixed-width gathers (out[k] = in[idx[k]], NULL index -> zero) in one launch
struct GatherDesc { const void* in; const uint32_t* idx; void* out; uint64_t m; uint32_t width; uint32_t null_ones; };   // null_ones: a NULL index yields all-one bytes (index composition: NULL stays NULL), else zeros
constexpr int kGatherBatch = 8;
struct GatherBatch { GatherDesc d[kGatherBatch]; };
void launch_gather_multi(const GatherBatch& b, int n, hipStream_t s);
void launch_gather_u32_nullable(const uint32_t* inner, const uint32_t* idx, uint32_t* out, uint64_t m, hipStream_t s);
void launch_lower_bound_u32(const uint32_t* a, uint64_t m, const uint32_t* m_dev, const uint64_t* bound, uint32_t nb, uint32_t* pos, hipStream_t s);
void launch_bytes_to_bits(const uint8_t* bytes, uint32_t gen, uint32_t* bits, uint32_t nwords, uint32_t* counters, uint32_t* status, hipStream_t s);
void launch_join_emit(const uint32_t* ent_slot, const uint32_t* ent_row, const uint32_t* chunk_nent, const uint32_t* chunk_off, const uint32_t* count,
                      const uint32_t* start, const uint32_t* rows, const uint32_t* row_of, uint64_t nchunks, uint64_t chunk_rows, uint32_t* b_idx, uint32_t* p_idx,
                      uint32_t* pair_off, uint32_t* cnt_out, uint32_t* visited, uint32_t cap, const uint32_t* stat_block, uint32_t* publish,
                      uint32_t* rows_out, hipStream_t s, int64_t* key_out = nullptr, uint64_t key_min = 0);
void launch_join_mark(const uint32_t* b_idx, const uint32_t* p_idx, uint64_t m, uint32_t* visited_bits, uint32_t* cnt_per_probe, hipStream_t s);
void launch_join_out_counts(const uint32_t* cnt, uint64_t np, uint32_t* out_cnt, hipStream_t s);
void launch_join_adjust_right(const uint32_t* b_in, const uint32_t* cnt, const uint32_t* in_off, const uint32_t* out_off, uint64_t np,
                              uint32_t* b_out, uint32_t* p_out, hipStream_t s);
// exchange, pass 2 (kernels_rel.hip k_part_scatter): every column of `cols` is moved from row order into per-part runs
struct PartCol {
  const void* src;       // the column's values (kind 0)
  const uint32_t* idx;   // ... read through this index vector when set (a deferred gather that was never materialised)
  void* out;             // ONE buffer over all parts: part p = positions [runs[p * n_units], runs[(p + 1) * n_units])
  uint32_t width;        // bytes per value: 1, 2, 4, 8, 16
  uint32_t kind;         // 0 = values; 1 = the row number itself (u32: the selection vector of the parts, for the columns that
                         //     need a gather of their own — validity bits, strings, Booleans)
};
constexpr int kPartCols = 12;
struct PartScatterArgs {
  const uint8_t* ids;        // pass 1: part of every row, 0xFF = dropped
  const uint32_t* runs;      // exclusive scan of pass 1's hist[part][unit] (part-major)
  uint64_t nrows;
  const uint32_t* nrows_dev; // the input's row count lives on the device (a join output of deferred size), nrows = capacity
  uint32_t n_units, rows_per_unit, n_parts, n_cols;
  PartCol cols[kPartCols];
};
void launch_part_scatter(const PartScatterArgs& a, hipStream_t s);
struct PendingSlots { const uint32_t* slot[64]; uint64_t cap[64]; uint32_t n; uint32_t pad_; };   // (kSizeSlots joins of deferred size in flight at most)
void launch_shuffle_meta(const uint32_t* starts, uint32_t n_parts, int64_t* rows_out, hipStream_t s);
void launch_pending_flags(const PendingSlots& p, int64_t* flag, hipStream_t s);
void launch_gather_stride_u32(const uint32_t* in, uint32_t stride, uint32_t n, uint32_t* out, hipStream_t s);
void launch_partition_ids(int W, const uint64_t* keys, uint64_t n, uint32_t nparts, uint32_t* part, uint32_t* hist, hipStream_t s);
void stable_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint64_t n, int bits,
                           hipStream_t s);

}  // namespace qhip

namespace qhip {
// ---- device-side assembly of aggregate output columns from dense group slots (kernels_rel.hip)
enum FinKind { F_KEY_FIXED = 0, F_KEY_DEC, F_KEY_UTF8_LEN, F_SUM64, F_SUM128, F_COUNT, F_AVG_F64, F_AVG_DEC, F_MM_INT, F_MM_F64, F_MM_F32, F_MM_DEC };
struct FinCol {
  int32_t kind;
  int32_t src_word;     // word offset of the value inside the slot (key word / value cell)
  int32_t cnt_word;     // word offset of the non-null count cell, -1 = always valid
  int32_t width;        // bytes per output value
  int32_t key_index;    // key columns: bit in the null-mask word, -1 = not nullable
  int32_t is_min;       // MIN cells store the complement of the order image
  int32_t is_signed;    // F_MM_INT
  int32_t pad;
  uint64_t mul_lo, mul_hi;   // F_AVG_DEC: 10^(s_out - s)
  uint64_t lim_lo, lim_hi;   // F_AVG_DEC: 10^p_out
  void* out_values;     // device
  uint64_t* out_valid;  // device, one ballot word per 64 groups
};
constexpr int kFinColsByValue = 24;   // column descriptors passed as a kernel argument (24 x 80 bytes); more go through cols_dev
void launch_agg_finalize(const uint64_t* dense, uint32_t G_cap, const uint32_t* g_dev, int slot_words, int null_mask_word, const FinCol* cols_host,
                         const FinCol* cols_dev, int ncols, uint32_t* null_counts, uint32_t* status, hipStream_t s);
void launch_agg_utf8_key_bytes(const uint64_t* dense, uint32_t G, int slot_words, int src_word, const uint32_t* offsets, uint8_t* data,
                               hipStream_t s);
}  // namespace qhip
