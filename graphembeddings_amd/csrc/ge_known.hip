// ge_known.hip -- the known-true cells of a ranking sweep (the filter of holE.py:454-463) as the per-tile lists the rank
// kernels take: for test row i = (fixed entity, relation), every entity e with (fixed, e, relation) known true that is a
// candidate contributes the cell (i, position of e).  The host keeps the known triples as a sorted index (key = fixed *
// n_rows + relation, evaluate.KnownIndex); three launches replace the tensor-op pipeline (two searchsorted,
// repeat_interleave, gather, sort by tile, searchsorted) that had become half of a filtered evaluation's time:
//   count   one wave per test row: binary search of its key range, one atomic per cell into its tile's counter
//   scan    exclusive prefix sum over the tiles (one workgroup)
//   fill    the same walk, cells written at their tile's cursor (order inside a tile is irrelevant to the counts)
#include "ge_common.h"

namespace ge {
namespace {

constexpr int kTile = 128;

__device__ __forceinline__ int64_t lower_bound_i64(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// FILL = false: tile_cnt[tile] += 1 per cell.  FILL = true: rc[off[tile] + cursor[tile]++] = (row % 128) << 7 | (col % 128).
// One WAVE per test row, its key range walked 64 entries at a time: a popular (entity, relation) has thousands of known
// completions, and with a thread per row that one thread was the whole kernel's duration.
template <bool FILL>
__global__ __launch_bounds__(256) void known_cells_kernel(const int64_t* __restrict__ key, const int64_t* __restrict__ ent,
                                                          int64_t M, const int64_t* __restrict__ fixed,
                                                          const int64_t* __restrict__ rel, int64_t B,
                                                          const int64_t* __restrict__ pos_of, int64_t n_rows, int64_t n_ct,
                                                          int32_t* __restrict__ tile_cnt, const int32_t* __restrict__ off,
                                                          uint16_t* __restrict__ rc) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int64_t f = fixed[i], r = rel[i];
  if (f < 0 || f >= n_rows || r < 0 || r >= n_rows) return;
  const int64_t q = f * n_rows + r;
  const int64_t lo = lower_bound_i64(key, M, q), hi = lower_bound_i64(key, M, q + 1);
  for (int64_t e = lo + lane; e < hi; e += 64) {
    const int64_t x = ent[e];
    const int64_t col = (x >= 0 && x < n_rows) ? pos_of[x] : -1;
    if (col < 0) continue;
    const int64_t tile = (i / kTile) * n_ct + col / kTile;
    const int slot = atomicAdd(&tile_cnt[tile], 1);
    if (FILL) rc[(int64_t)off[tile] + slot] = (uint16_t)(((i % kTile) << 7) | (col % kTile));
  }
}

// off[0 .. n] = exclusive prefix sum of cnt[0 .. n - 1] (off[n] = total); cnt is zeroed for the fill pass's cursors.
// One workgroup walks the array in pieces of 4096: four consecutive counters per thread (coalesced), a wave scan by
// shuffles, the sixteen wave totals through LDS, the running total in a register.
__global__ __launch_bounds__(1024) void known_scan_kernel(int32_t* __restrict__ cnt, int32_t* __restrict__ off, int64_t n) {
  __shared__ int32_t wave_tot[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int32_t carry = 0;
  for (int64_t base = 0; base < n; base += 4096) {
    const int64_t i0 = base + 4 * t;
    int32_t v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = i0 + j < n ? cnt[i0 + j] : 0;
    const int32_t mine = v[0] + v[1] + v[2] + v[3];
    int32_t incl = mine;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const int32_t up = __shfl_up(incl, s, kWave);
      if (lane >= s) incl += up;
    }
    if (lane == 63) wave_tot[w] = incl;
    __syncthreads();
    int32_t before = carry, all = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) { const int32_t x = wave_tot[k]; if (k < w) before += x; all += x; }
    int32_t run = before + incl - mine;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < n) { off[i0 + j] = run; run += v[j]; cnt[i0 + j] = 0; }
    carry += all;
    __syncthreads();
  }
  if (t == 0) off[n] = carry;
}

}  // namespace

// pass 0: tile_cnt (zeroed here) <- cells per tile, off <- their exclusive prefix sum (off[n_tiles] = total).
// pass 1 (after the caller has read the total and sized rc): rc filled; tile_cnt is scratch.
int known_cells_launch(int pass, const int64_t* key, const int64_t* ent, int64_t M, const int64_t* fixed, const int64_t* rel,
                       int64_t B, const int64_t* pos_of, int64_t n_rows, int64_t n_cand, int32_t* tile_cnt, int32_t* off,
                       uint16_t* rc, hipStream_t st) {
  const int64_t n_ct = (n_cand + kTile - 1) / kTile, n_tiles = ((B + kTile - 1) / kTile) * n_ct;
  if (n_tiles >= INT32_MAX) return GE_ENOTSUP;
  const unsigned grid = (unsigned)((B + 3) / 4);                 // four waves = four test rows a workgroup
  if (pass == 0) {
    hipError_t e = hipMemsetAsync(tile_cnt, 0, sizeof(int32_t) * (size_t)n_tiles, st);
    if (e != hipSuccess) return (int)e;
    if (B > 0 && M > 0)
      hipLaunchKernelGGL(known_cells_kernel<false>, dim3(grid), dim3(256), 0, st, key, ent, M, fixed, rel, B, pos_of, n_rows,
                         n_ct, tile_cnt, off, rc);
    hipLaunchKernelGGL(known_scan_kernel, dim3(1), dim3(1024), 0, st, tile_cnt, off, n_tiles);
  } else if (B > 0 && M > 0) {
    hipLaunchKernelGGL(known_cells_kernel<true>, dim3(grid), dim3(256), 0, st, key, ent, M, fixed, rel, B, pos_of, n_rows,
                       n_ct, tile_cnt, off, rc);
  }
  return launch_status();
}

}  // namespace ge
