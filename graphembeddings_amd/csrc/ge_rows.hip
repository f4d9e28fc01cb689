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

}  // namespace ge
