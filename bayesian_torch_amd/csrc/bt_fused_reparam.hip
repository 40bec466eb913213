#include "bt_fused_dispatch.h"
namespace bt {
int launch_split(FwdArgs& a, hipStream_t stream);  // bt_fused_split.hip: 0 taken, 1 not applicable, < 0 error
int launch_reparam(bool linear, FwdArgs& a, hipStream_t stream) {
  {   // (a Linear layer is a 1x1 convolution over 1x1 images: the same memory layout)
    FwdArgs b = a;
    const int rc = launch_split(b, stream);
    if (rc <= 0) return rc;
  }
  return launch_flavour<false, false>(linear, a, stream);
}
}  // namespace bt
