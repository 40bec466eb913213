#include "bt_fused_dispatch.h"
namespace bt {
int launch_reparam_inj(bool linear, FwdArgs& a, hipStream_t stream) { return launch_flavour<false, true>(linear, a, stream); }
}  // namespace bt
