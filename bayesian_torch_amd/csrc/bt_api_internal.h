// Host-side helpers shared by the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bt_hip.h"
#include "bt_device.h"

namespace bt {

int set_error(int code, const char* msg);  // records msg for bt_last_error_string(); returns code
void note_kernel(const char* name);         // records the kernel instance a fused launch chose (bt_last_kernel_name)

inline int check_launch(const char* who) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return BT_OK;
  char buf[256];
  snprintf(buf, sizeof(buf), "%s: launch failed: %s", who, hipGetErrorString(e));
  return set_error(BT_ERR_HIP_BASE - (int)e, buf);
}

inline RngKey make_key(const bt_rng& r, uint32_t tensor) {
  RngKey k;
  k.seed_lo = (uint32_t)r.seed;
  k.seed_hi = (uint32_t)(r.seed >> 32);
  k.call = r.call;
  k.layer_tensor = layer_tensor_word(r.layer_id, tensor);
  return k;
}

}  // namespace bt
