// ge_rows.hip -- row scatter-add (the ScatterSub of holE.py:296), row gather, and the type-safe
// corruption sampler (holE.py:97-140 fused with the host resample of holE.py:343-347).
#include "ge_common.h"

namespace ge {

// table[idx[i]] += val[i]  -- one wavefront per row, one dword per lane per atomic instruction,
// 256 contiguous bytes per wave-instruction (the shape the memory-side float atomics run at full
// rate for).  Duplicated indices all land (ScatterSub semantics, graph.pbtxt:47850-48001).
__global__ __launch_bounds__(kBlock) void scatter_add_rows_kernel(
    float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ idx,
    const float* __restrict__ val, int64_t R) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t r = wave; r < R; r += nwaves) {
    const int32_t id = idx[r];
    if (id < 0 || id >= N) continue;
    float* dst = table + (int64_t)id * d;
    const float* src = val + r * d;
    for (int c = lane; c < d; c += kWave) atomic_add_f32(dst + c, src[c]);
  }
}

// out[i] = table[idx[i]]  (zeros for idx < 0) -- one wavefront per row.
template <int VEC>
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ idx, int64_t R,
    float* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int nvec = d / VEC;
  for (int64_t r = wave; r < R; r += nwaves) {
    const int32_t id = idx[r];
    const bool ok = id >= 0 && id < N;
    const float* src = table + (int64_t)(ok ? id : 0) * d;
    float* dst = out + r * d;
    for (int j = lane; j < nvec; j += kWave) {
      float v[VEC];
      if (ok) load_vec<VEC>(src + j * VEC, v);
      else {
#pragma unroll
        for (int q = 0; q < VEC; ++q) v[q] = 0.f;
      }
      store_vec<VEC>(dst + j * VEC, v);
    }
  }
}

__global__ __launch_bounds__(kBlock) void corrupt_batch_kernel(
    const int32_t* __restrict__ pos, int64_t B, const int32_t* __restrict__ id_to_type, int64_t N,
    const int64_t* __restrict__ type_offsets, int32_t n_types, const int32_t* __restrict__ type_ids,
    uint64_t seed, uint64_t step, int32_t padded_size, int32_t mode, int32_t* __restrict__ neg) {
  const bool batch_heads = (mode == GE_CORRUPT_BATCH_COIN) ? batch_coin_heads(seed, step) : false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t t[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    int col;
    const int32_t repl = corrupt_one(t, i, batch_heads, id_to_type, N, type_offsets, n_types, type_ids,
                                     seed, step, padded_size, mode, col);
    t[col] = repl;
    neg[3 * i] = t[0]; neg[3 * i + 1] = t[1]; neg[3 * i + 2] = t[2];
  }
}

// ---------------------------------------------------------------- Bernoulli filtered sampler
// GPU form of the reference's native sampler init.so (init.cpp:159-246): the side to corrupt is
// chosen per triple with P(tail) = hpt/(hpt+tph) of its relation (init.cpp:226-228) and the
// replacement is drawn uniformly among the entities that do NOT complete a known triple, by mapping
// tmp in [0, E - cnt) past the sorted known entities of the (fixed entity, relation) key with a
// binary search (init.cpp:177-188).  One thread per triple, per-row Philox draws instead of the
// reference's single global LCG (init.cpp:145-150); the two init.cpp defects (sizeof(pointer) memset,
// `j < rig` loop bound) live in the host-side statistics and are fixed there.
__device__ __forceinline__ int64_t lower_bound64(const int64_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] < key) lo = mid + 1; else hi = mid; }
  return lo;
}
__device__ __forceinline__ int64_t upper_bound64(const int64_t* __restrict__ a, int64_t n, int64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= key) lo = mid + 1; else hi = mid; }
  return lo;
}

#define GE_TAG_BSIDE 0x62736964u
#define GE_TAG_BPICK 0x62706963u

__global__ __launch_bounds__(kBlock) void bernoulli_corrupt_kernel(
    const int32_t* __restrict__ pos, int64_t B, const int64_t* __restrict__ bh_key,
    const int32_t* __restrict__ bh_ent, const int64_t* __restrict__ bt_key,
    const int32_t* __restrict__ bt_ent, int64_t n_known, const uint32_t* __restrict__ tail_threshold,
    int32_t n_rel, int32_t ent_lo, int32_t n_ent, uint64_t seed, uint64_t step, int32_t* __restrict__ neg) {
  const uint32_t slo = (uint32_t)step, shi = (uint32_t)(step >> 32);
  const uint32_t klo = (uint32_t)seed, khi = (uint32_t)(seed >> 32);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < B;
       i += (int64_t)gridDim.x * blockDim.x) {
    int32_t t[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    const int32_t r = t[2];
    if (r < 0 || r >= n_rel) { neg[3 * i] = -1; neg[3 * i + 1] = -1; neg[3 * i + 2] = -1; continue; }
    const uint32_t ilo = (uint32_t)i, ihi = (uint32_t)((uint64_t)i >> 32);
    const uint32_t w_side = philox_w0(slo, shi, ilo, ihi, klo ^ GE_TAG_BSIDE, khi);
    const uint32_t w_pick = philox_w0(slo, shi, ilo, ihi, klo ^ GE_TAG_BPICK, khi);
    const bool tail_side = w_side < tail_threshold[r];
    const int64_t* key_arr = tail_side ? bh_key : bt_key;
    const int32_t* ent_arr = tail_side ? bh_ent : bt_ent;
    const int64_t key = (int64_t)(tail_side ? t[0] : t[1]) * n_rel + r;
    const int col = tail_side ? 1 : 0;
    // the known completions of (fixed entity, relation): a sorted run [first, last] of ent_arr
    const int64_t first = lower_bound64(key_arr, n_known, key);
    const int64_t last = upper_bound64(key_arr, n_known, key) - 1;
    const int64_t cnt = last >= first ? last - first + 1 : 0;
    const int64_t free_n = (int64_t)n_ent - cnt;
    int32_t repl = -1;
    if (free_n > 0) {
      // the draw-th entity that is NOT a known completion (init.cpp:159-190: skip the known ones by bisection on
      // "free entities below the p-th known one" = ent[p] - ent_lo - (p - first))
      const int64_t draw = (int64_t)(((uint64_t)w_pick * (uint64_t)free_n) >> 32);
      auto free_below = [&](int64_t p) { return (int64_t)ent_arr[p] - ent_lo - (p - first); };
      int64_t j;
      if (cnt == 0 || draw < free_below(first)) j = draw;
      else if (draw >= free_below(last)) j = draw + cnt;
      else {
        int64_t below = first, above = last + 1;        // free_below(below) <= draw < free_below(above)
        while (below + 1 < above) {
          const int64_t probe = (below + above) >> 1;
          if (free_below(probe) <= draw) below = probe; else above = probe;
        }
        j = draw + (below - first + 1);
      }
      repl = ent_lo + (int32_t)j;
    }
    t[col] = repl;
    neg[3 * i] = t[0]; neg[3 * i + 1] = t[1]; neg[3 * i + 2] = t[2];
  }
}

int bernoulli_corrupt_launch(const int32_t* pos, int64_t B, const int64_t* bh_key, const int32_t* bh_ent,
                             const int64_t* bt_key, const int32_t* bt_ent, int64_t n_known,
                             const uint32_t* tail_threshold, int32_t n_rel, int32_t ent_lo, int32_t n_ent,
                             uint64_t seed, uint64_t step, int32_t* neg, hipStream_t st) {
  if (n_known < 0 || n_rel <= 0 || n_ent <= 0) return GE_EINVAL;
  if (B == 0) return 0;
  const int grid = grid_for(B, kBlock);
  hipLaunchKernelGGL(bernoulli_corrupt_kernel, dim3(grid), dim3(kBlock), 0, st, pos, B, bh_key, bh_ent, bt_key,
                     bt_ent, n_known, tail_threshold, n_rel, ent_lo, n_ent, seed, step, neg);
  return launch_status();
}

int scatter_add_rows_launch(float* table, int64_t N, int32_t d, const int32_t* idx, const float* val,
                            int64_t R, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  if (d <= 0) return GE_EINVAL;
  if (R == 0) return 0;
  const int grid = grid_for(R, kBlock / kWave);
  hipExtLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, N, d, idx, val, R);
  return launch_status();
}

int gather_rows_launch(const float* table, int64_t N, int32_t d, const int32_t* idx, int64_t R,
                       float* out, hipStream_t st) {
  if (d <= 0) return GE_EINVAL;
  if (R == 0) return 0;
  const int grid = grid_for(R, kBlock / kWave);
  const uintptr_t a = reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out);
  if (d % 4 == 0 && a % 16 == 0)
    hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(grid), dim3(kBlock), 0, st, table, N, d, idx, R, out);
  else if (d % 2 == 0 && a % 8 == 0)
    hipLaunchKernelGGL(gather_rows_kernel<2>, dim3(grid), dim3(kBlock), 0, st, table, N, d, idx, R, out);
  else
    hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(grid), dim3(kBlock), 0, st, table, N, d, idx, R, out);
  return launch_status();
}

int corrupt_batch_launch(const int32_t* pos, int64_t B, const int32_t* id_to_type, int64_t N,
                         const int64_t* type_offsets, int32_t n_types, const int32_t* type_ids,
                         uint64_t seed, uint64_t step, int32_t padded_size, int32_t mode, int32_t* neg,
                         hipStream_t st) {
  if (mode < 0 || mode > 3 || padded_size < 0 || n_types < 0) return GE_EINVAL;
  if (B == 0) return 0;
  const int grid = grid_for(B, kBlock);
  hipLaunchKernelGGL(corrupt_batch_kernel, dim3(grid), dim3(kBlock), 0, st, pos, B, id_to_type, N,
                     type_offsets, n_types, type_ids, seed, step, padded_size, mode, neg);
  return launch_status();
}

// ---------------------------------------------------------------- validation tick (holE.py:299-304, 351-360)
#define GE_TAG_VSEL 0x7673656Cu

// a batch of validation triples drawn uniformly with replacement (the reference takes them from a shuffle queue
// over the validation split, holE.py:300): out[i] = valid[philox(seed, counter, i) % V]
__global__ __launch_bounds__(kBlock) void select_rows_kernel(const int32_t* __restrict__ valid, int64_t V, int64_t B,
                                                             uint64_t seed, uint64_t counter, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B) return;
  const uint32_t w0 = philox_w0((uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)i, (uint32_t)(i >> 32),
                                (uint32_t)seed ^ GE_TAG_VSEL, (uint32_t)(seed >> 32));
  const uint32_t w1 = philox_w0((uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)i, (uint32_t)(i >> 32) ^ 0x80000000u,
                                (uint32_t)seed ^ GE_TAG_VSEL, (uint32_t)(seed >> 32));
  const uint64_t r = (((uint64_t)w1 << 32) | w0) % (uint64_t)V;
  out[3 * i] = valid[3 * r];
  out[3 * i + 1] = valid[3 * r + 1];
  out[3 * i + 2] = valid[3 * r + 2];
}

// one workgroup: mean of the loss vector (fixed summation order: bitwise reproducible) -> *mean_out; the pocket
// bookkeeping of holE.py:357-360 on the device: improved = mean < *best, *best = min, *flag = improved
__global__ __launch_bounds__(1024) void mean_pocket_kernel(const float* __restrict__ loss, int64_t B,
                                                           float* __restrict__ mean_out, float* __restrict__ best,
                                                           int32_t* __restrict__ flag) {
  __shared__ double part[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < B; i += 1024) s += (double)loss[i];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += part[w];
    const float mean = (float)(t / (double)B);
    *mean_out = mean;
    const bool improved = mean < *best;     // a NaN loss never improves
    if (improved) *best = mean;
    *flag = improved ? 1 : 0;
  }
}

// pocket <- table iff *flag (the flag is read on the device: no host decision, no synchronisation)
__global__ __launch_bounds__(kBlock) void copy_if_kernel(const float4* __restrict__ src, float4* __restrict__ dst,
                                                         int64_t n4, const float* __restrict__ src_tail,
                                                         float* __restrict__ dst_tail, int n_tail,
                                                         const int32_t* __restrict__ flag) {
  if (*flag == 0) return;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

int select_rows_launch(const int32_t* valid, int64_t V, int64_t B, uint64_t seed, uint64_t counter, int32_t* out,
                       hipStream_t st) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(select_rows_kernel, dim3(grid_for(B, kBlock)), dim3(kBlock), 0, st, valid, V, B, seed, counter, out);
  return launch_status();
}

int mean_pocket_launch(const float* loss, int64_t B, float* mean_out, float* best, int32_t* flag, hipStream_t st) {
  hipLaunchKernelGGL(mean_pocket_kernel, dim3(1), dim3(1024), 0, st, loss, B, mean_out, best, flag);
  return launch_status();
}

int copy_if_launch(const float* src, float* dst, int64_t n, const int32_t* flag, hipStream_t st) {
  if (n == 0) return 0;
  const bool vec = reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  const int64_t rest = n - 4 * n4;
  if (rest > kBlock) {   // unaligned buffers: plain float copy through the tail path is not enough; fall back to scalars
    return GE_EINVAL;
  }
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n4 + kBlock - 1) / kBlock, 256 * 8));
  hipLaunchKernelGGL(copy_if_kernel, dim3(grid), dim3(kBlock), 0, st, reinterpret_cast<const float4*>(src),
                     reinterpret_cast<float4*>(dst), n4, src + 4 * n4, dst + 4 * n4, (int)rest, flag);
  return launch_status();
}

}  // namespace ge
