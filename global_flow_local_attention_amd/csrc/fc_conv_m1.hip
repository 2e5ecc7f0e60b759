// fc_conv_kernel<1, *, *>: the FC-layer convolutions in arithmetic mode 1 (fc_gemm.h), see fc_conv_impl.h.
#include "fc_conv_impl.h"

namespace gfla {
GFLA_DEFINE_FC_CONV_MODE(1)
}  // namespace gfla
