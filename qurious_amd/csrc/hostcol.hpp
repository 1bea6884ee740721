// hostcol.hpp — small host-side column builder used when an operator assembles its (small) result on
// the host before it becomes a device table again (aggregate output: G rows).
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace qhip {

struct HostColumn {
  DType type;
  int64_t length = 0;
  int64_t null_count = 0;
  std::vector<uint8_t> values;     // fixed-width values / bit-packed bools
  std::vector<uint8_t> validity;   // bitmap; empty when null_count == 0
  std::vector<int32_t> offsets;    // utf8
  std::vector<uint8_t> data;       // utf8
  void init_fixed(const DType& t, int64_t n);
  void set_null(int64_t i);
  template <class T> T* as() { return reinterpret_cast<T*>(values.data()); }
};

// builds a single-batch (or zero-batch) device table from host columns
qhip_table* table_from_host(Ctx* ctx, const std::vector<std::string>& names, const std::vector<bool>& nullable,
                            std::vector<HostColumn>& cols, int64_t nrows, bool zero_batches);

}  // namespace qhip
