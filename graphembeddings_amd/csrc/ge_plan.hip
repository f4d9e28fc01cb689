// ge_plan.hip -- the local stages of the row-sharded step's exchange planner (graphembeddings_amd/sharded.py,
// ShardedTrainer.plan_chunk).  Negatives never depend on the table, so for a chunk of S steps the
// planner works out, ahead of time and off the training stream: which distinct rows each step needs and from which
// owner, where each of the step's 6B gradient slots finds its row in the staging buffer, and the work items of
// the two atomic-free reductions (gradient slots -> staged row, received gradient sums -> shard row).
//
// The planner is a radix sort of step-tagged keys (device library sort, on the host side) plus the index
// arithmetic around it.  Written with tensor ops that arithmetic is ~200 launches of 5-10 us per chunk -- 155 us of
// device time per step that competes with the training kernels for the same HBM; here it is six kernels, every
// one a single pass over its input.  HBM-bound integer work: no LDS tiling, coalesced streams, scattered 4-byte
// writes only where the permutation requires them.
#include "ge_common.h"

namespace ge {
namespace {

constexpr int kThreads = 256;

// element (s, j) of cat([pos, neg], 1).reshape(S, 6B): pos[s] flattened, then neg[s] flattened
__device__ __forceinline__ int32_t slot_id(const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int64_t B,
                                           int64_t s, int64_t j) {
  const int64_t B3 = 3 * B;
  return j < B3 ? pos[s * B3 + j] : neg[s * B3 + (j - B3)];
}

// key = ((step * G + id % G) * N + id): sorts as (step, owner, id).  Ids outside [0, N) alias row 0 (their slots
// stay empty: plan_scatter gives them remap -1).
template <typename K>
__global__ __launch_bounds__(kThreads) void plan_keys_kernel(const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ neg, int64_t S, int64_t B,
                                                             int64_t N, int64_t G, K* __restrict__ key) {
  const int64_t n = S * 6 * B;
  for (int64_t f = (int64_t)blockIdx.x * kThreads + threadIdx.x; f < n; f += (int64_t)gridDim.x * kThreads) {
    const int64_t s = f / (6 * B), j = f - s * 6 * B;
    int64_t id = slot_id(pos, neg, B, s, j);
    if (id < 0 || id >= N) id = 0;
    key[f] = (K)((s * G + id % G) * N + id);
  }
}

// flag[i] = 1 where a run of equal keys starts (its inclusive scan - 1 is the run index of every element)
template <typename K>
__global__ __launch_bounds__(kThreads) void plan_flags_kernel(const K* __restrict__ key, int64_t n,
                                                              int32_t* __restrict__ flag) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads)
    flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// per run u (incl = inclusive scan of flag): its first sorted position, key / div and key % div; first_pos[U] = n
template <typename K>
__global__ __launch_bounds__(kThreads) void plan_heads_kernel(const K* __restrict__ key, const int32_t* __restrict__ incl,
                                                              int64_t n, int64_t div, int32_t* __restrict__ first_pos,
                                                              int32_t* __restrict__ quot, int32_t* __restrict__ rem) {
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int32_t u = incl[i] - 1;
    if (i == 0 || incl[i - 1] != incl[i]) {
      const int64_t k = (int64_t)key[i];
      first_pos[u] = (int32_t)i;
      quot[u] = (int32_t)(k / div);
      rem[u] = (int32_t)(k - (k / div) * div);
    }
    if (i == n - 1) first_pos[u + 1] = (int32_t)n;
  }
}

// per sorted position i (slot = perm[i]): remap[slot] = staged row of the slot inside its step (-1: invalid id),
// order[i] = the slot's position in ge_hinge_grad's output (h+,t+,r+,h-,t-,r- per pair)
__global__ __launch_bounds__(kThreads) void plan_scatter_kernel(const int64_t* __restrict__ perm,
                                                                const int32_t* __restrict__ incl,
                                                                const int64_t* __restrict__ step_start,
                                                                const int32_t* __restrict__ pos,
                                                                const int32_t* __restrict__ neg, int64_t S, int64_t B,
                                                                int64_t N, int32_t* __restrict__ remap,
                                                                int32_t* __restrict__ order) {
  const int64_t M6 = 6 * B, n = S * M6;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    const int64_t slot = perm[i];
    const int64_t s = slot / M6, j = slot - s * M6;
    const int32_t id = slot_id(pos, neg, B, s, j);
    remap[slot] = (id >= 0 && id < N) ? (int32_t)(incl[i] - 1 - step_start[s]) : -1;
    const int64_t tr = j / 3, X = j - tr * 3;
    order[i] = (int32_t)((tr % B) * 6 + (tr / B) * 3 + X);
  }
}

// per run u < U: work items needed (<= max_item sources each) and whether the run is split over several
__global__ __launch_bounds__(kThreads) void plan_item_counts_kernel(const int32_t* __restrict__ first_pos,
                                                                    const int32_t* __restrict__ n_runs, int64_t cap,
                                                                    int32_t max_item, int32_t* __restrict__ n_it,
                                                                    int32_t* __restrict__ split) {
  const int64_t U = *n_runs;
  for (int64_t u = (int64_t)blockIdx.x * kThreads + threadIdx.x; u < cap; u += (int64_t)gridDim.x * kThreads) {
    int32_t k = 0;
    if (u < U) k = (first_pos[u + 1] - first_pos[u] + max_item - 1) / max_item;
    n_it[u] = k;
    split[u] = k > 1 ? 1 : 0;
  }
}

// per run u: its items (begin/length into the sorted order, destination row; ~row when the run is split) at
// it_incl[u] - n_it[u] .., and the run's row in the list of split rows
__global__ __launch_bounds__(kThreads) void plan_items_kernel(const int32_t* __restrict__ first_pos,
                                                              const int32_t* __restrict__ n_runs,
                                                              const int64_t* __restrict__ it_incl,
                                                              const int64_t* __restrict__ sp_incl,
                                                              const int32_t* __restrict__ row_of, /* nullable */
                                                              const int32_t* __restrict__ bucket,
                                                              const int64_t* __restrict__ step_start, int64_t G,
                                                              int32_t max_item, int32_t* __restrict__ begin,
                                                              int32_t* __restrict__ length, int32_t* __restrict__ target,
                                                              int64_t* __restrict__ split_rows) {
  const int64_t U = *n_runs;
  for (int64_t u = (int64_t)blockIdx.x * kThreads + threadIdx.x; u < U; u += (int64_t)gridDim.x * kThreads) {
    const int32_t p0 = first_pos[u], cnt = first_pos[u + 1] - p0;
    const int32_t k = (cnt + max_item - 1) / max_item;
    // destination row: given (owner apply: the shard row) or the run's index inside its step (pre-reduction)
    const int32_t row = row_of ? row_of[u] : (int32_t)(u - step_start[bucket[u] / G]);
    const int64_t it0 = it_incl[u] - k;
    for (int32_t r = 0; r < k; ++r) {
      begin[it0 + r] = p0 + r * max_item;
      length[it0 + r] = min(max_item, cnt - r * max_item);
      target[it0 + r] = k > 1 ? -row - 1 : row;
    }
    if (k > 1) split_rows[sp_incl[u] - 1] = row;
  }
}

inline int grid_1d(int64_t n) {
  const int64_t g = (n + kThreads - 1) / kThreads;
  return (int)std::max<int64_t>(1, std::min<int64_t>(g, 256 * 32));
}

}  // namespace

int plan_keys_launch(const int32_t* pos, const int32_t* neg, int64_t S, int64_t B, int64_t N, int64_t G, void* key,
                     int key64, hipStream_t st) {
  const int64_t n = S * 6 * B;
  if (n == 0) return 0;
  if (key64) hipLaunchKernelGGL(plan_keys_kernel<int64_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, pos, neg, S, B, N, G, (int64_t*)key);
  else hipLaunchKernelGGL(plan_keys_kernel<int32_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, pos, neg, S, B, N, G, (int32_t*)key);
  return launch_status();
}

int plan_flags_launch(const void* key, int64_t n, int key64, int32_t* flag, hipStream_t st) {
  if (n == 0) return 0;
  if (key64) hipLaunchKernelGGL(plan_flags_kernel<int64_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, (const int64_t*)key, n, flag);
  else hipLaunchKernelGGL(plan_flags_kernel<int32_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, (const int32_t*)key, n, flag);
  return launch_status();
}

int plan_heads_launch(const void* key, const int32_t* incl, int64_t n, int key64, int64_t div, int32_t* first_pos,
                      int32_t* quot, int32_t* rem, hipStream_t st) {
  if (n == 0) return 0;
  if (key64) hipLaunchKernelGGL(plan_heads_kernel<int64_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, (const int64_t*)key, incl, n, div, first_pos, quot, rem);
  else hipLaunchKernelGGL(plan_heads_kernel<int32_t>, dim3(grid_1d(n)), dim3(kThreads), 0, st, (const int32_t*)key, incl, n, div, first_pos, quot, rem);
  return launch_status();
}

int plan_scatter_launch(const int64_t* perm, const int32_t* incl, const int64_t* step_start, const int32_t* pos,
                        const int32_t* neg, int64_t S, int64_t B, int64_t N, int32_t* remap, int32_t* order,
                        hipStream_t st) {
  const int64_t n = S * 6 * B;
  if (n == 0) return 0;
  hipLaunchKernelGGL(plan_scatter_kernel, dim3(grid_1d(n)), dim3(kThreads), 0, st, perm, incl, step_start, pos, neg, S, B, N, remap, order);
  return launch_status();
}

int plan_item_counts_launch(const int32_t* first_pos, const int32_t* n_runs, int64_t cap, int32_t max_item,
                            int32_t* n_it, int32_t* split, hipStream_t st) {
  if (cap == 0) return 0;
  hipLaunchKernelGGL(plan_item_counts_kernel, dim3(grid_1d(cap)), dim3(kThreads), 0, st, first_pos, n_runs, cap, max_item, n_it, split);
  return launch_status();
}

int plan_items_launch(const int32_t* first_pos, const int32_t* n_runs, int64_t cap, const int64_t* it_incl,
                      const int64_t* sp_incl, const int32_t* row_of, const int32_t* bucket, const int64_t* step_start,
                      int64_t G, int32_t max_item, int32_t* begin, int32_t* length, int32_t* target, int64_t* split_rows,
                      hipStream_t st) {
  if (cap == 0) return 0;
  hipLaunchKernelGGL(plan_items_kernel, dim3(grid_1d(cap)), dim3(kThreads), 0, st, first_pos, n_runs, it_incl, sp_incl, row_of, bucket, step_start, G, max_item, begin, length, target, split_rows);
  return launch_status();
}

}  // namespace ge
