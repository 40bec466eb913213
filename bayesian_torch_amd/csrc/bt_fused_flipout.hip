#include "bt_fused_dispatch.h"
namespace bt {
int launch_split_flip(FwdArgs& a, hipStream_t stream);  // bt_fused_split_flip.hip: 0 taken, 1 not applicable, < 0 error
int launch_flipout(bool linear, FwdArgs& a, hipStream_t stream) {
  {
    FwdArgs b = a;
    const int rc = launch_split_flip(b, stream);
    if (rc <= 0) return rc;
    if (a.pixel_major) {  // pixel-major tiles are not built for the split Flipout: tiles of whole images instead
      b = a;
      b.pixel_major = 0;
      b.out_vec4 = 0;
      const int rc2 = launch_split_flip(b, stream);
      if (rc2 <= 0) return rc2;
    }
  }
  return launch_flavour<true, false>(linear, a, stream);
}
}  // namespace bt
