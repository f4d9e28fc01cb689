// ge_shard.hip -- the row-sharded training step's planner (graphembeddings_amd/sharded.py; SURVEY.md 8e,
// BASELINE config 4): the table is mod-sharded over G ranks (owner = id % G, local row = id / G); per step a
// rank evaluates holE.py:287-296 on its B pairs with every gradient taken against the pre-step table.
//
// Negatives never depend on the table, so everything about a step except its floating-point work is planned
// ahead for a chunk of S steps, natively, with the multi-workgroup radix sort of ge_prep_big.hip:
//
// REQUESTER side (ge_shard_plan).  A pair's <= 4 gradient slots are keyed by a VIRTUAL row
//     own rows:    id / G                          (< R = ceil(N / G))
//     other rows:  R * (1 + id % G) + id / G       (sorted behind the own rows, by owner, then row)
// and sorted per step.  Own rows are read IN PLACE from the shard and updated exactly like in the one-GPU loop
// (sole-slot rows by the producing pair, the rest by ONE read-modify-write per distinct row: the same work items).
// The runs of other owners' rows, numbered u = 0, 1, .. in sorted order, ARE the step's staging order: row u of
// the fetched-rows buffer, row u of the gradient-sum buffer that goes back, entry u of the request list (grouped
// by owner: the all-to-all split sizes are the per-owner run counts).  Work items of those rows carry R + u.
//
// OWNER side (ge_shard_owner_plan).  The request lists received from the peers, keyed (row, position in the
// receive buffer) and sorted per step, give the work items that add the received gradient sums to the shard --
// again one read-modify-write per distinct row, in a fixed order.
#include "ge_prep.h"

namespace ge {

int shard_hinge_grad_launch(float*, int32_t, const float*, const int32_t*, const int32_t*, const int32_t*, int32_t, int64_t,
                            float, float, float, float*, int32_t*, float*, float*, int, hipStream_t, hipEvent_t, hipEvent_t,
                            const int32_t*, const float* const*, int);
int apply_items_launch(float*, int, const TileGeom&, const int32_t*, const int32_t*, const float*, int, float*, hipStream_t,
                       hipEvent_t, hipEvent_t, int det);
size_t sort_scratch_bytes(int64_t n, int64_t n_sub, int64_t P);
unsigned long long* sort_scratch_keys(void* scratch);
const unsigned long long* sort_tiles_launch(void* scratch, int64_t n, int64_t n_sub, int64_t P, int64_t n_rows, hipStream_t st,
                                            const unsigned* limit = nullptr);
int items_launch(const unsigned long long* sorted, int64_t n, const TileGeom& G, int direct, int32_t* out, const ShardOut* so,
                 hipStream_t st);
int relation_order_launch(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t s0, int64_t n, int64_t N,
                          int32_t* out, int64_t stride, int64_t off_order, void* scratch, hipStream_t st);

// ---------------------------------------------------------------- requester: keys
// grid (S, tiles).  neg must differ from pos in at most one of (head, tail) -- what ge_corrupt_batch produces;
// any other pair is treated like one with an invalid id (NaN loss, no update).
__global__ __launch_bounds__(kPrepThreads) void shard_keys_kernel(
    const int32_t* __restrict__ pos, const int32_t* __restrict__ neg, int64_t B, int64_t N, int32_t G, int32_t rank,
    int32_t R, PrepLayout L, int32_t* __restrict__ prep, int32_t* __restrict__ pos_src, int32_t* __restrict__ neg_src,
    unsigned long long* __restrict__ keys_out) {
  const int tid = threadIdx.x, sub = blockIdx.y;
  const int64_t s = blockIdx.x;
  int32_t* slot_item = prep + s * L.stride + L.off_slot;
  const int64_t i0 = (int64_t)sub * kSub;
  const int S = (int)((B - i0) < kSub ? (B - i0) : kSub);
  unsigned long long* out = keys_out + (s * L.n_sub + sub) * L.P;
  for (int i = tid; i < 6 * S; i += kPrepThreads) slot_item[6 * i0 + i] = -1;
  auto vrow = [&](int32_t id) -> unsigned long long {
    const int32_t o = id % G, l = id / G;
    return (unsigned long long)(uint32_t)(o == rank ? l : R * (1 + o) + l);
  };
  for (int il = tid; il < (int)L.S; il += kPrepThreads) {
    unsigned long long k[4] = {kInvalidKey, kInvalidKey, kInvalidKey, kInvalidKey};
    if (il < S) {
      const int64_t i = i0 + il;
      const int32_t* p = pos + (s * B + i) * 3;
      const int32_t* n = neg + (s * B + i) * 3;
      const int32_t p0 = p[0], p1 = p[1], p2 = p[2], n0 = n[0], n1 = n[1], n2 = n[2];
      bool bad = p0 < 0 || p1 < 0 || p2 < 0 || p0 >= N || p1 >= N || p2 >= N || n0 < 0 || n1 < 0 || n0 >= N || n1 >= N;
      bad = bad || n2 != p2 || (n0 != p0 && n1 != p1);
      pos_src[(s * B + i) * 3] = -1; pos_src[(s * B + i) * 3 + 1] = -1; pos_src[(s * B + i) * 3 + 2] = -1;
      neg_src[s * B + i] = -1;
      if (!bad) {
        k[0] = (vrow(p0) << 32) | (uint32_t)(6 * i);
        k[1] = (vrow(p1) << 32) | (uint32_t)(6 * i + 1);
        k[2] = (vrow(p2) << 32) | (uint32_t)(6 * i + 2);
        if (n0 != p0) k[3] = (vrow(n0) << 32) | (uint32_t)(6 * i + 3);
        else if (n1 != p1) k[3] = (vrow(n1) << 32) | (uint32_t)(6 * i + 4);
      }
    }
#pragma unroll
    for (int X = 0; X < 4; ++X) out[4 * il + X] = k[X];
  }
}

// grid (S, tiles) over the SORTED keys: distinct remote rows that start in each tile, and how many of them each
// owner is asked for (counts [S][G], zeroed before; column `rank` counts the distinct own rows)
__global__ __launch_bounds__(kPrepThreads) void shard_heads_kernel(
    const unsigned long long* __restrict__ sorted, int P, int n_sub, int32_t R, int32_t G, int32_t rank,
    int32_t* __restrict__ tile_heads, int32_t* __restrict__ counts) {
  __shared__ int own[64 + 1];
  const int tid = threadIdx.x, tile = blockIdx.y;
  const unsigned long long* g = sorted + (int64_t)blockIdx.x * n_sub * P;
  const int base = tile * P;
  if (tid <= 64) own[tid] = 0;
  __syncthreads();
  for (int i = tid; i < P; i += kPrepThreads) {
    const unsigned long long kk = g[base + i];
    if (kk == kInvalidKey) continue;
    const uint32_t row = (uint32_t)(kk >> 32);
    const bool head = (base + i == 0) || (uint32_t)(g[base + i - 1] >> 32) != row;
    if (!head) continue;
    if (row < (uint32_t)R) { atomicAdd(&own[rank], 1); continue; }     // column `rank`: distinct own rows (statistics)
    atomicAdd(&own[row / (uint32_t)R - 1], 1);
    atomicAdd(&own[64], 1);
  }
  __syncthreads();
  if (tid < G && own[tid]) atomicAdd(&counts[(int64_t)blockIdx.x * G + tid], own[tid]);
  if (tid == 0) tile_heads[(int64_t)blockIdx.x * n_sub + tile] = own[64];
}

// ---------------------------------------------------------------- owner: keys of the received request lists
// req_all: the chunk's requests in (step, peer) order; step s owns [req_start[s], req_start[s+1]).  Key = (row,
// position inside the step's list) = (row, row of the receive buffer that will hold its gradient sum).
__global__ __launch_bounds__(kPrepThreads) void owner_keys_kernel(
    const int32_t* __restrict__ req_all, const int64_t* __restrict__ req_start, int P, int n_sub, int32_t rows_local,
    unsigned long long* __restrict__ keys_out) {
  const int64_t s = blockIdx.x;
  const int64_t b = req_start[s], n = req_start[s + 1] - b;
  unsigned long long* out = keys_out + (s * n_sub + blockIdx.y) * P;
  for (int i = threadIdx.x; i < P; i += kPrepThreads) {
    const int64_t j = (int64_t)blockIdx.y * P + i;
    unsigned long long k = kInvalidKey;
    if (j < n) {
      const int32_t row = req_all[b + j];
      if (row >= 0 && row < rows_local) k = ((unsigned long long)(uint32_t)row << 32) | (uint32_t)j;
    }
    out[i] = k;
  }
}

static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------- host
size_t shard_plan_scratch_bytes(int64_t B, int64_t S) {
  const PrepLayout L = prep_layout(B);
  return sort_scratch_bytes(S, L.n_sub, L.P) + align_up_sz(sizeof(int32_t) * (size_t)S * (size_t)L.n_sub, 256);
}

int shard_plan_launch(const int32_t* pos, const int32_t* neg, int64_t S, int64_t B, int64_t N, int32_t G, int32_t rank,
                      int32_t* records, int32_t* pos_src, int32_t* neg_src, int32_t* req_row, int32_t* counts,
                      void* scratch, int peer, hipStream_t st) {
  const PrepLayout L = prep_layout(B);
  const int32_t R = (int32_t)((N + G - 1) / G);
  const dim3 grid((unsigned)S, (unsigned)L.n_sub), block(kPrepThreads);
  int32_t* tile_heads = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(scratch) + sort_scratch_bytes(S, L.n_sub, L.P));
  hipError_t e = hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)S * (size_t)G, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(shard_keys_kernel, grid, block, 0, st, pos, neg, B, N, G, rank, R, L, records, pos_src, neg_src,
                     sort_scratch_keys(scratch));
  const unsigned long long* sorted = sort_tiles_launch(scratch, S, L.n_sub, L.P, (int64_t)R * (G + 1), st);
  hipLaunchKernelGGL(shard_heads_kernel, grid, block, 0, st, sorted, (int)L.P, (int)L.n_sub, R, G, rank, tile_heads, counts);
  const ShardOut so{R, B, pos_src, neg_src, req_row, tile_heads, peer ? 1 : 0};
  const int rc = items_launch(sorted, S, geom_of(L), /*direct=*/1, records, &so, st);
  if (rc || L.off_order < 0) return rc;
  return relation_order_launch(pos, 0, 0, B, 0, S, N, records, L.stride, L.off_order, scratch, st);
}

int shard_grad_launch(float* shard, int32_t d, const float* staged, const int32_t* pos_src, const int32_t* neg_src,
                      const int32_t* record, int32_t R, int64_t B, float margin, float lr, float max_norm, int spectral,
                      float* loss, int32_t* gidx, float* gval, float* gsum, const float* const* peers, int n_peers,
                      hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  const PrepLayout L = prep_layout(B);
  return shard_hinge_grad_launch(shard, d, staged, pos_src, neg_src, record + L.off_slot, R, B, margin, lr, max_norm, loss,
                                 gidx, gval, gsum, spectral, st, e0, e1, L.off_order >= 0 ? record + L.off_order : nullptr,
                                 peers, n_peers);
}

int shard_apply_launch(float* shard, int32_t d, const int32_t* record, int64_t B, const int32_t* gidx, const float* gval,
                       int32_t R, float* gsum, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
  return apply_items_launch(shard, d, geom_of(prep_layout(B)), record, gidx, gval, R, gsum, st, e0, e1, /*det=*/0);
}

// owner records: tiles of kOwnerP keys; cap = the longest per-step request list of the chunk
constexpr int kOwnerP = 16384;
static inline int owner_tiles(int64_t cap) { return (int)((cap + kOwnerP - 1) / kOwnerP); }
int64_t shard_owner_record_words(int64_t cap) { return cap <= 0 ? 0 : geom_plain(kOwnerP, owner_tiles(cap)).stride; }
size_t shard_owner_scratch_bytes(int64_t cap, int64_t S) { return cap <= 0 ? 0 : sort_scratch_bytes(S, owner_tiles(cap), kOwnerP); }

int shard_owner_plan_launch(const int32_t* req_all, const int64_t* req_start, int64_t S, int64_t cap, int32_t rows_local,
                            int32_t* records, void* scratch, hipStream_t st) {
  if (cap <= 0 || S <= 0) return 0;
  const int n_sub = owner_tiles(cap);
  const dim3 grid((unsigned)S, (unsigned)n_sub), block(kPrepThreads);
  hipLaunchKernelGGL(owner_keys_kernel, grid, block, 0, st, req_all, req_start, kOwnerP, n_sub, rows_local,
                     sort_scratch_keys(scratch));
  const unsigned long long* sorted = sort_tiles_launch(scratch, S, n_sub, kOwnerP, rows_local, st);
  return items_launch(sorted, S, geom_plain(kOwnerP, n_sub), /*direct=*/0, records, nullptr, st);
}

int shard_owner_apply_launch(float* shard, int32_t d, const int32_t* record, int64_t cap, const float* recv, hipStream_t st) {
  if (cap <= 0) return 0;
  return apply_items_launch(shard, d, geom_plain(kOwnerP, owner_tiles(cap)), record, nullptr, recv, 0x7FFFFFFF, nullptr, st,
                            nullptr, nullptr, /*det=*/0);
}

}  // namespace ge
