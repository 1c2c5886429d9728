// association.h -- device-side joins behind sfmba_find_2d3d_matches / sfmba_merge_candidates (association.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfmba {

// return values besides 0 (ok) and positive hipError_t codes
enum { ASSOC_ERR_CAPACITY = -1, ASSOC_ERR_TOO_LARGE = -2 };

// Host pointers in and out; see include/sfmba.h for the meaning of the arrays.
int assoc_find_2d3d(hipStream_t s, int device, int n_views, const unsigned char* view_done, int n_pt, const int64_t* view_ptr,
                    const int32_t* view_idx, const int32_t* feat_idx, int n_pairs, const int32_t* pair_left, const int32_t* pair_right,
                    const int64_t* pair_ptr, const int32_t* query_idx, const int32_t* train_idx, int64_t* out_ptr, int32_t* out_point,
                    int32_t* out_feature, int64_t cap, int64_t* total);

int assoc_radius_candidates(hipStream_t s, int device, int n_exist, const float* exist_xyz, int n_new, const float* new_xyz, float max_dist,
                            int64_t* cand_ptr, int32_t* cand_idx, int64_t cap, int64_t* total);

}  // namespace sfmba
