// ge_rank_dev.h -- what the two ranking kernels (ge_rank.hip: any embedding_dim % 8 == 0; ge_rank_pipe.hip:
// the software-pipelined sweep for embedding_dim % 40, % 32 or % 24 == 0) share.
#pragma once
#include "ge_common.h"

namespace ge {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kRB = 128;          // test rows per workgroup, candidates per tile

// sigmoid for the ranking epilogue: 4 VALU instructions (v_exp_f32, v_rcp_f32; ~2 ulp), used for EVERY loss the
// ranking kernels form -- candidates and true entities alike -- so comparisons are self-consistent; within 1e-6 of
// sigmoidf_dev, well inside the 1e-5 score bar.
__device__ __forceinline__ float rank_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}

// ge_rank_pipe.hip.  spec: the table is a spectral HolE table (ge_hole_to_spectral).  Returns GE_ENOTSUP when embedding_dim has no pipelined instantiation (the caller then uses
// the generic kernel), 0 / an error otherwise.
int rank_pipe_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                     const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                     const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                     float* scores_out, int spec, const void* planes_ws, hipStream_t st, int vs_loss = 0);

// ge_rank_f16.hip: the split-precision sweep (embedding_dim % 8 == 0 in 56 ... 288, max_norm <= 8), ranks or scores.
// planes_ws: the candidates' fp16 planes + entity -> position map (rank_planes_launch into rank_planes_bytes bytes, 256-byte
// aligned) for the same (table, cand, max_norm, spec); NULL: built inside, in a stream-ordered allocation.
int64_t rank_planes_bytes(int64_t N, int32_t d, int64_t K);
int rank_planes_launch(const float* table, int64_t N, int32_t d, const int32_t* cand, int64_t K, float max_norm, int spec,
                       void* planes_ws, hipStream_t st);
int sweep_f16_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                     const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                     const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                     float* scores_out, int spec, int scores_only, int sweep_flags, const void* planes_ws, hipStream_t st);

// ge_complex_score_1vK on the pipelined sweep (same GE_ENOTSUP convention)
int score_pipe_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* cand,
                      int64_t K, float max_norm, int apply_sigmoid, int cand_is_head, float* out, hipStream_t st);

}  // namespace ge
