// workspace_layout.hpp -- indices of the per-problem solver scalars kept in the workspace between kernels; shared by
// the kernels (mpc_kernels.hpp) and by the host code that sizes the workspace (cpmpc_api.hip).
#pragma once

namespace cpmpc {

// per-problem real scalars kept in the workspace (index into `sc`)
// SC_TRIAL: 1 when SC_F_LAST / SC_CN_LAST are the merit pieces of the CURRENT iterate as the line search evaluated
// them (the accepted trial point IS the new iterate, bit for bit), 0 when no trial has been accepted yet
enum { SC_LAMBDA = 0, SC_MU, SC_F_LAST, SC_CN_LAST, SC_UPREV, SC_ALPHA, SC_TRIAL, SC_COUNT };
// per-problem int scalars (index into `ist`)
enum { IS_STATUS = 0, IS_ITERS, IS_LS_EVALS, IS_FAILED, IS_COUNT };
// bins of the histogram of SQP iterations per problem that finalize_kernel leaves for the host (the last bin collects
// every larger count) -- what the host plans the next step's stages of the fused pipeline from
constexpr int kFbBins = 16;
// at most this many workgroups of finalize_kernel report (every fb_stride-th one: a sample spread evenly over the batch)
constexpr int kFbReporters = 256;

}  // namespace cpmpc
