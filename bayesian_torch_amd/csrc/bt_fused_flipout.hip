#include "bt_fused_dispatch.h"
namespace bt {
int launch_flipout(bool linear, FwdArgs& a, hipStream_t stream) { return launch_flavour<true, false>(linear, a, stream); }
}  // namespace bt
