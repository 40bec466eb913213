#include "bt_fused_dispatch.h"
namespace bt {
int launch_flipout_inj(bool linear, FwdArgs& a, hipStream_t stream) { return launch_flavour<true, true>(linear, a, stream); }
}  // namespace bt
