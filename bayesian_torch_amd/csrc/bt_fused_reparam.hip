#include "bt_fused_dispatch.h"
namespace bt {
int launch_reparam(bool linear, FwdArgs& a, hipStream_t stream) { return launch_flavour<false, false>(linear, a, stream); }
}  // namespace bt
