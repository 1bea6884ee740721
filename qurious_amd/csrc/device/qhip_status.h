// qhip_status.h — device -> host status words shared by the kernels and the host runtime.
// (Prepended to qhip_device.hpp when the device source is embedded for hiprtc.)
#pragma once
enum {
  QS_OVERFLOW = 0,       // HBM group/join table too small: host retries with a larger one
  QS_KEY_TOO_LONG = 1,   // Utf8 key longer than the packed-key limit
  QS_DIV_ZERO = 2,       // integer division by zero on a valid row (arrow: DivideByZero)
  QS_CAST_OVERFLOW = 3,  // cast with safe=false overflowed (cast.rs:15-18)
  QS_ARITH_OVERFLOW = 4, // checked arithmetic overflowed (integer MIN / -1)
  QS_LDS_SPILL = 5,      // informational: some keys bypassed the LDS-staged table
  QS_LDS_USED = 6,       // value, not a flag: occupied LDS-table slots summed over the workgroups (statistics runs only)
  QS_MAXCOUNT = 7,       // value, not a flag: largest number of build rows sharing one join key
  QS_WORDS = 8
};
