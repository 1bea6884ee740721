// codegen.hpp — turns typed expression trees into the policy structs the hand-written kernel
// templates of device/qhip_device.hpp are instantiated with.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "expr.hpp"

namespace qhip {

// what a generated kernel needs bound at launch: which table column sits in KArgs.c[slot], literal values
struct KernelBindings {
  std::vector<int> cols;          // slot -> input column index
  std::vector<char> indirect;     // slot -> read through the deferred gather's index vector (InputCol::indirect)
  std::vector<char> narrow;       // slot -> 0, or the width in bytes of the narrow copy the kernel reads (InputCol::narrow_bytes)
  std::vector<char> rec;          // slot -> 0, or the stride of the record copy an indirect column is read from (InputCol::rec_stride)
  std::vector<uint64_t> lit_lo;
  std::vector<int64_t> lit_hi;
  std::string strlits;            // concatenated Utf8 literals
  std::vector<int> stroff;        // per literal slot: offset into strlits (size = nlits + 1)
};

class ExprGen {
 public:
  ExprGen(const ExprSet& es, const std::vector<InputCol>& in);
  // append the statements that compute node k (and whatever it needs that was not emitted yet)
  void emit(int k, std::string& out);
  std::string val(int k) const { return "v" + std::to_string(k); }
  std::string ok(int k) const;                       // validity expression ("true" when never null)
  std::string ptr(int k) const { return "p" + std::to_string(k); }
  std::string len(int k) const { return "l" + std::to_string(k); }
  KernelBindings bind;
  // how a column element is addressed in the generated code: ((T*)col + base)[idx]; `row` is the absolute row (bitmaps)
  void set_indexing(const std::string& base, const std::string& idx, const std::string& row) { base_ = base; idx_ = idx; row_ = row; }
  // raw mode: column loads are collected into `load_code` (branch-free, writes fields of `Raw w`) and the expression code
  // only aliases those fields, so a kernel can issue the loads of several rows before any row's (branchy) computation
  void set_raw_mode(bool on) { raw_ = on; }
  // fixed-width column values are read with streaming (non-temporal) loads: they are used once and should not push a
  // kernel's random-access working set (join table, hash filter) out of L2
  void set_streaming_loads(bool on) { nt_ = on; }
  void mark_utf8_key(int node, int words) { utf8_key_words_[node] = words; }
  bool raw_key_prefetched(int node) const { return raw_ && utf8_key_words_.count(node) > 0; }
  std::string raw_fields, load_code;
  static std::string ctype(const DType& t);
  static std::string i128_const(i128 v);

 private:
  int col_slot(int table_col);
  int lit_slot(const ENode& n);
  int str_slot(const std::string& bytes);   // a literal slot holding only string bytes (LIKE patterns)
  const ExprSet& es_;
  const std::vector<InputCol>& in_;
  std::vector<bool> done_;
  std::string base_ = "", idx_ = "i", row_ = "i";
  bool raw_ = false;
  bool nt_ = false;
  std::map<int, int> utf8_key_words_;
};

// ---------------------------------------------------------------- aggregate plan
enum CellKind { CELL_ROWS = 0, CELL_SUM_I128, CELL_SUM_U64, CELL_SUM_F64, CELL_CNT, CELL_MAXORD64, CELL_MAXORD128 };

struct KeyDesc {
  int root; DType type; bool nullable;
  int word_off;   // first key word (after the optional null-mask word)
  int words;
};
struct ArgDesc { int root; DType type; bool nullable; };
struct CellDesc {
  int kind; int arg;   // arg = index into args (-1 for CELL_ROWS)
  bool is_min;         // MAXORD cells: true stores ~ord(v) so that the running maximum is the minimum
  int off;             // word offset inside the slot's cell area
  int words;
};
struct AggDesc {
  int kind; DType ret; int arg;
  int value_cell;      // SUM/AVG: sum cell, MIN/MAX: maxord cell, COUNT: -1
  int count_cell;      // cell counting the non-null argument values (the CELL_ROWS cell when the argument is never null)
};
struct AggPlan {
  int W = 0;                   // key words (incl. null-mask word)
  bool null_mask_word = false;
  int R = 4;
  int KC = 0;                  // wave-resident hot keys
  int slot_words = 0;          // 1 + W + cell words
  int part_pr = 0;             // partitioned path: rows per thread of the LDS-staged scatter (0: records too wide for it)
  bool has_runs = false;       // the source has the sorted-run entry point qk_agg_runs (grouped, no scan filter)
  mutable bool not_sorted = false;   // ... and an execution found this plan's input not to be of that kind (never tried again)
  int RC = 0;                  // rows per lane of the consecutive-rows form (qk_filter_agg_cons); 0 = the plan has no such entry point
  std::vector<KeyDesc> keys;
  std::vector<ArgDesc> args;
  std::vector<CellDesc> cells;
  std::vector<AggDesc> aggs;
  KernelBindings bind;
  std::string source;          // policy struct + extern "C" kernel, to be appended to the device header
  std::string kernel_name;
  mutable uint32_t last_groups = 0;   // groups the plan produced the last time it ran (sizes the first table attempt)
  mutable uint32_t last_dense = 0;    // ... and the occupied slots over all table replicas (sizes the one read-back)
  mutable uint64_t learnt_at = 0;     // when the two were last set (a per-process tick): twin plans take over the fresher pair
  // the small replicated first-attempt table of a plan that keeps producing few groups is kept between calls, zeroed at
  // the END of a call (off the critical path): [status words | dense counter | table | dense slots]
  mutable std::shared_ptr<void> arena;
  mutable size_t arena_bytes = 0;
  mutable bool arena_clean = false;
  mutable std::shared_ptr<void> module;   // the loaded main kernel (jit.hpp Module), looked up once per cached plan
  mutable std::shared_ptr<void> strlit;   // DevBuf: the plan's string literals, uploaded once
};

// group_roots / aggs refer to nodes of `es`; predicate_root < 0 = no filter
void plan_aggregate(const ExprSet& es, const std::vector<InputCol>& input, int predicate_root, const int32_t* group_roots,
                    int n_groups, const qhip_agg* aggs, int n_aggs, int rows_per_thread, AggPlan& out, bool dev_rows = false);
// (dev_rows: the input's row count lives on the device — KArgs::nrows_dev, a join output of deferred size; a separate
// instantiation of the kernel bodies, so that kernels over ordinary tables keep their row count a plain kernel argument)

struct MaskPlan { KernelBindings bind; std::string source; std::string kernel_name; int mask_r = 4; /* mask words per tile (P::MASK_R) */ };
void plan_predicate_mask(const ExprSet& es, const std::vector<InputCol>& input, int predicate_root, MaskPlan& out);

struct KeysPlan {
  int W = 0; bool null_mask_word = false;   // join keys never carry a null mask (NULL keys never match); partition keys do not need one either
  std::vector<KeyDesc> keys;
  KernelBindings bind; std::string source; std::string kernel_name;
  int probe_r = 4;   // KEYS_KERNEL_PROBE: probe rows per thread and tile (P::PROBE_R)
};
// which kernel wraps the generated key policy: qk_eval_keys (key words -> arrays), qk_join_probe (fused filter + key + lookup,
// qh_join_probe_body) or qk_join_scatter (build rows -> region entries of the LDS-staged join build, qh_join_scatter_body)
// ... or the two kernels of the dense (direct-address) join layout: qk_join_dense_build (qh_join_dense_build_body) and
// qk_join_probe_dense / qk_join_probe_dense_lds (qh_join_probe_dense_body)
// ... or pass 1 of the exchange's fused filter + partition: qk_part_ids (qh_part_ids_body; n_parts is a compile-time constant of it)
enum { KEYS_KERNEL_EVAL = 0, KEYS_KERNEL_PROBE = 1, KEYS_KERNEL_SCATTER = 2, KEYS_KERNEL_DENSE_BUILD = 3, KEYS_KERNEL_DENSE_PROBE = 4, KEYS_KERNEL_PARTITION = 5 };
void plan_keys(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, KeysPlan& out, int predicate_root = -1,
               int kernel = KEYS_KERNEL_EVAL, bool dev_rows = false, int n_parts = 0);   // (dev_rows: the build kernels and the partition kernel)

// ---------------------------------------------------------------- exchange, pass 2 (qh_part_scatter_body)
// widths[c]: bytes per value of column c (1, 2, 4, 8, 16), 0 = the row number itself as u32 (the parts' selection vector);
// indirect[c]: the column is read through an index vector; n_parts picks the ranking variant (<= 8, <= 16, more)
struct PartScatterPlan { std::string source; std::string kernel_name; int rows_per_lane = 4; int wg_threads = 512; /* qk_part_scatter_wg's workgroup */ };
// unstable: the order of the rows inside a part is free (ranks from returning DS atomics: cheap for many parts)
void plan_part_scatter(const std::vector<int>& widths, const std::vector<char>& indirect, int n_parts, bool dev_rows, PartScatterPlan& out, bool unstable = false);

// ---------------------------------------------------------------- projection (physical/plan/projection.rs:27-46)
struct ProjOutDesc { int root; DType type; bool nullable; };
struct ProjectionPlan {
  bool has_utf8 = false;             // some output is a computed Utf8 value: the source also has qk_project_copy (the second pass)
  std::vector<ProjOutDesc> outs;     // the computed outputs, in kernel slot order
  KernelBindings bind; std::string source; std::string kernel_name;
};
// roots: the expressions to compute (plain columns are shared by the operator and never get here)
void plan_projection(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, ProjectionPlan& out);

// ---------------------------------------------------------------- sort keys (physical/plan/sort.rs:48-82)
// Every sort key becomes an ORDER-PRESERVING unsigned image (ascending unsigned order of the image = ascending order of
// the value under arrow's lexsort: integers biased, floats in IEEE total order, Decimal128 as two words) so that the
// lexicographic sort is a sequence of stable LSD radix passes. Utf8 keys must be plain columns; their images are cut
// from the column's bytes chunk by chunk at sort time (words = 0 here).
struct SortKeyDesc {
  int root; DType type; bool nullable;
  int word_off, words;   // image words [word_off, word_off + words), least significant first
  int top_bits;          // significant bits of the most significant word (radix passes needed)
  int column;            // Utf8: the table column
};
struct SortKeysPlan {
  int NW = 0;
  std::vector<SortKeyDesc> keys;
  KernelBindings bind; std::string source; std::string kernel_name;
};
void plan_sort_keys(const ExprSet& es, const std::vector<InputCol>& input, const int32_t* roots, int n, SortKeysPlan& out);

}  // namespace qhip
