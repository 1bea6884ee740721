// jit.hpp — kernel module cache (hiprtc) of libqhip.
#pragma once
#include <hip/hip_runtime_api.h>

#include <memory>
#include <string>
#include <vector>

#include "common.hpp"

namespace qhip {

struct Module {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  int wgs_per_cu = 0;   // occupancy of the kernel at the launch shape its operator uses (0 = not asked yet)
  size_t wgs_dyn_lds = 0;   // ... and the dynamic LDS size it was asked for
  ~Module();
};

const char* device_source();
std::string full_source_for(const std::string& policy_source);
std::string cache_path_for(const std::string& dir, const std::string& full_source);
std::vector<char> compile_to_code_object(const std::string& full_source, std::string* log_out);
// returns the loaded module for (device header + policy_source); compiles / reads the disk cache on first use
std::shared_ptr<Module> get_module(Ctx* ctx, const std::string& policy_source, const std::string& kernel_name);

}  // namespace qhip
