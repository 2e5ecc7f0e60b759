// Matrix-core scatter paths (patch_mfma.hip).  Both return GFLA_ERR_UNSUPPORTED without launching anything when the
// shape is outside the path; the target is accumulated into (+=) when accumulate != 0, overwritten otherwise.
#pragma once

#include "gfla_common.h"

namespace gfla {

// adaptive != 0: the kernels decide ON THE DEVICE (from how far the flow spreads the patches) whether they run;
// *skip_stat / *skip_limit receive the predicate the caller's LDS-atomic kernel has to be launched with right behind
// (it returns at once when *skip_stat <= skip_limit, i.e. when the matrix-core path did the work).
int agg_source_bwd_mfma(const float *flow, const float *attn, const float *gout, float *gsrc, void *workspace,
                        int64_t B, int64_t C, int64_t Hs, int64_t Ws, int64_t H, int64_t W, int k, int accumulate,
                        int adaptive, const unsigned **skip_stat, unsigned *skip_limit, hipStream_t stream);
int rs_input1_bwd_mfma(const float *in2, const float *gout, float *gin1, void *workspace, int64_t B, int64_t C,
                       int64_t Hi, int64_t Wi, int64_t H, int64_t W, int k, int trunc, int accumulate, int adaptive,
                       const unsigned **skip_stat, unsigned *skip_limit, hipStream_t stream);
// scratch of one op invocation: [patch table + tile rows + dispatch statistic | resample2d's tap records (resample2d.hip)]
constexpr int kRsTapRecBytes = 48;
int64_t pm_table_bytes(int64_t B, int64_t H, int64_t W, int entries);     // the first part = offset of the second
int64_t pm_workspace_bytes(int64_t B, int64_t H, int64_t W, int entries);

}  // namespace gfla
