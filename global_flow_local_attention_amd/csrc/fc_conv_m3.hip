// fc_conv_kernel<3, *, *>: the FC-layer convolutions in arithmetic mode 3 (fc_gemm.h), see fc_conv_impl.h.
#include "fc_conv_impl.h"

namespace gfla {
GFLA_DEFINE_FC_CONV_MODE(3)
}  // namespace gfla
