// njode_gen_host.h -- entry points of the shape-generic kernel family (njode_gen.hip) for the
// C ABI unit (njode_api.hip).  Same contracts as the njode_*_f32 functions of include/njode_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/njode_hip.h"

namespace njode {

// Budget of the per-(tile, Euler step) TRAINING RECORDS of the masked lockstep kernels, which grow
// up to 16x when a tile holds fewer than 16 paths (q4_paths_per_tile, gen_paths_per_tile): the
// smaller of NJODE_REC_BUDGET_GB (default 16) and 1/8 of the device's TOTAL memory.  Total, not
// free: the layout is computed twice per call (njode_workspace_bytes, then the call itself, after
// the caller has allocated the workspace) and must come out the same both times.  Defined in
// njode_api.hip.
double record_budget_bytes();

namespace gen {

bool gen_supported(const NjodeDims* dims);
size_t gen_param_count(const NjodeDims* dims);
int gen_workspace_bytes(const NjodeDims* dims, int B, int n_obs, int nt, int K, int call_flags,
                        size_t* out, bool plan_only);
int gen_plan(const NjodeDims* dims, const NjodeBatch* b, const NjodeSchedule* s, int call_flags,
             void* plan, size_t plan_bytes, hipStream_t st);
int gen_forward(const NjodeDims* dims, const float* params, const NjodeBatch* b,
                const NjodeSchedule* s, int call_flags, float weight, float dropout_p,
                uint64_t seed, float* hT, float* loss, float* path_h, float* path_y, void* ws,
                size_t ws_bytes, hipStream_t st);
int gen_backward(const NjodeDims* dims, const float* params, const NjodeBatch* b,
                 const NjodeSchedule* s, int call_flags, float weight, float dropout_p,
                 uint64_t seed, const float* grad_loss, float* grad_params, void* ws,
                 size_t ws_bytes, hipStream_t st);

}  // namespace gen
}  // namespace njode
