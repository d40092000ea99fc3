// Host-only part of the library: routines that never touch the GPU -- the error string behind svt_last_error(), parameter intake by
// key (svt_*_load_param), configuration validation (svt_encoder_create) and the frames -> notes scan (svt_frames_to_notes).  Plain
// C++17 with no HIP header, so the same translation unit is compiled twice: by hipcc into libsvt_mi355*.so, and by g++ with
// -fsanitize=address,undefined into the CPU test binary of `make san` (tests/test_host_sanitizers.py drives it with the reference's
// frame2note fixture and hostile arguments).  Sanitizers run on the CPU build only.
#pragma once
#include "../../include/svt_mi355.h"

#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace svt {

void set_error(const std::string& msg);

struct Param {
  std::vector<float> v;
  std::vector<int64_t> shape;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};
typedef std::map<std::string, Param> ParamMap;

int load_param_into(ParamMap& m, const char* key, const void* data, int dtype, const int64_t* shape, int ndim);
const Param* find(const ParamMap& m, const std::string& k);
int need(const ParamMap& m, const std::string& k, std::vector<int64_t> shape, const Param** out);
int validate_cfg(const svt_encoder_config& c);

#ifdef SVT_OPERAND_F16
// the IEEE-half build serves the exact fp32 mode and the 16-bit throughput mode; the split-operand engines live in the bf16 build
static inline bool valid_precision(int p) { return p == SVT_PREC_FP32 || p == SVT_PREC_BF16; }
#else
static inline bool valid_precision(int p) { return p >= SVT_PREC_FP32 && p <= SVT_PREC_FP16X3; }
#endif

}  // namespace svt
