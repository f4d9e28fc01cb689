// ge_prep_big.hip -- the prepare stage of ge_train_steps / ge_train_steps_logloss for steps of MORE than
// 4096 units (holE.py:340-362 at large batch_size): the step's (row, slot) keys no longer fit one
// workgroup's LDS, so the stable LSD radix sort runs across workgroups:
//
//   prep_big_keys_kernel     grid (steps, tiles): negatives (holE.py:97-140, 343-347) + the <= 4 sort keys
//                            of every unit, in slot order, to a global key array of tiles x P positions
//   per radix pass           prep_big_hist_kernel     digit counts of every tile
//                            prep_big_scatter_kernel  ranks inside the tile by the wave-level multisplit of
//                                                     train_prepare_kernel (ballot ranks + per-wave digit
//                                                     counters), global position = digits below + same digit
//                                                     in earlier tiles + earlier waves of this tile + rank
//   prep_big_items_kernel    grid (steps, tiles): the sorted tile back into LDS, cut into work items of
//                            <= 16 slots of one row exactly like the one-tile kernel; runs that cross a tile
//                            boundary are followed through global memory (an item belongs to the tile it
//                            starts in), so the update stays ONE plain read-modify-write per distinct row
//                            at any batch size -- no float atomics between tiles.
//
// Everything is integer work off the training stream's critical path (the look-ahead pipeline runs it on
// the side stream one chunk of steps ahead).
#include "ge_prep.h"
#include <algorithm>

namespace ge {

// ---------------------------------------------------------------- keys
__global__ __launch_bounds__(kPrepThreads) void prep_big_keys_kernel(
    const int32_t* __restrict__ triples, int64_t T, int64_t first_row, int64_t B, int64_t s0,
    const int32_t* __restrict__ id_to_type, int64_t N, const int64_t* __restrict__ type_offsets,
    int32_t n_types, const int32_t* __restrict__ type_ids, uint64_t seed, uint64_t global_step0,
    int32_t padded_size, int32_t mode, int direct, int negs, int32_t* __restrict__ prep,
    unsigned long long* __restrict__ keys_out) {
  const PrepLayout L = prep_layout(B, negs);
  const int tid = threadIdx.x, sub = blockIdx.y;
  const int64_t s = s0 + blockIdx.x;
  int32_t* rec = prep + (int64_t)blockIdx.x * L.stride;
  int32_t* slot_item = rec + L.off_slot;
  const int64_t i0 = (int64_t)sub * kSub;
  const int S = (int)((L.units - i0) < kSub ? (L.units - i0) : kSub);     // units of this tile
  StepSource src{triples + 3 * step_row(first_row, T, B, s), B, id_to_type, N, type_offsets, n_types, type_ids, seed,
                 global_step0 + (uint64_t)s, padded_size, mode, negs};
  const bool batch_heads = (mode == GE_CORRUPT_BATCH_COIN) ? batch_coin_heads(seed, src.step) : false;
  unsigned long long* out = keys_out + ((int64_t)blockIdx.x * L.n_sub + sub) * L.P;
  if (direct)
    for (int i = tid; i < 6 * S; i += kPrepThreads) slot_item[6 * i0 + i] = -1;
  for (int il = tid; il < (int)L.S; il += kPrepThreads) {
    unsigned long long k[4] = {kInvalidKey, kInvalidKey, kInvalidKey, kInvalidKey};
    if (il < S) unit_keys(src, i0 + il, batch_heads, rec, k);
#pragma unroll
    for (int X = 0; X < 4; ++X)
      if (X < L.epu) out[L.epu * il + X] = k[X];
  }
}

// ---------------------------------------------------------------- one radix pass
// hist[(step * radix + digit) * n_sub + tile]
// `limit` (nullable): the largest row field among the keys, written by the kernel that made them.  A pass whose digits
// are all zero (shift at or beyond the limit's bit length) is the identity permutation: the count kernel returns and the
// scatter kernel copies -- the relation-order sort runs over relation ids (18 of them in a Diffbot-like graph, 1,345
// in FB15k) with the pass count of the TABLE's row range, two of its three passes sorted nothing.
__device__ __forceinline__ bool pass_is_identity(const unsigned* limit, int shift) {
  return limit && ((unsigned long long)(*limit) >> (shift - 32)) == 0ull;
}

__global__ __launch_bounds__(kPrepThreads) void prep_big_hist_kernel(
    const unsigned long long* __restrict__ src, int P, int n_sub, int shift, int bits, unsigned* __restrict__ hist,
    const unsigned* __restrict__ limit) {
  __shared__ unsigned cnt[kMaxRadix];
  if (pass_is_identity(limit, shift)) return;
  const int tid = threadIdx.x, tile = blockIdx.y;
  const int radix = 1 << bits;
  const unsigned dmask = (unsigned)radix - 1u;
  if (tid < radix) cnt[tid] = 0;
  __syncthreads();
  const unsigned long long* k = src + ((int64_t)blockIdx.x * n_sub + tile) * P;
  for (int i = tid; i < P; i += kPrepThreads) atomicAdd(&cnt[(unsigned)(k[i] >> shift) & dmask], 1u);
  __syncthreads();
  if (tid < radix) hist[((int64_t)blockIdx.x * radix + tid) * n_sub + tile] = cnt[tid];
}

__global__ __launch_bounds__(kPrepThreads) void prep_big_scatter_kernel(
    const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst,
    const unsigned* __restrict__ hist, int P, int n_sub, int shift, int bits, const unsigned* __restrict__ limit) {
  __shared__ unsigned whist[kMaxRadix * kPrepWaves];   // (digit, wave) counters of this tile
  __shared__ unsigned gbase[kMaxRadix];                // first global position of (digit, this tile)
  __shared__ unsigned dstart[kMaxRadix];               // first position of the digit inside the tile once sorted
  __shared__ unsigned wsum[4], wsum2[4];
  extern __shared__ __attribute__((aligned(16))) unsigned long long lkeys[];   // the tile, reordered by digit
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, tile = blockIdx.y;
  if (pass_is_identity(limit, shift)) {                // (block-uniform)
    const int64_t o = ((int64_t)blockIdx.x * n_sub + tile) * P;
    for (int i = tid; i < P; i += kPrepThreads) dst[o + i] = src[o + i];
    return;
  }
  const int radix = 1 << bits, R = P / kPrepThreads, seg = kWave * R;
  const unsigned dmask = (unsigned)radix - 1u;
  const unsigned* h = hist + (int64_t)blockIdx.x * radix * n_sub;
  // keys with a smaller digit anywhere + the same digit in earlier tiles
  unsigned tot = 0, before = 0, mine = 0;
  if (tid < radix)
    for (int q = 0; q < n_sub; ++q) { const unsigned v = h[tid * n_sub + q]; if (q < tile) before += v; if (q == tile) mine = v; tot += v; }
  int incl = 0, incl2 = 0;
  if (tid < kMaxRadix) {
    incl = wave_incl_add((int)tot, lane);
    incl2 = wave_incl_add((int)mine, lane);
    if (lane == kWave - 1) { wsum[wave] = (unsigned)incl; wsum2[wave] = (unsigned)incl2; }
  }
  for (int i = tid; i < radix * kPrepWaves; i += kPrepThreads) whist[i] = 0;
  __syncthreads();
  if (tid < radix) {
    unsigned run = (unsigned)incl - tot, run2 = (unsigned)incl2 - mine;
    for (int w = 0; w < wave; ++w) { run += wsum[w]; run2 += wsum2[w]; }
    gbase[tid] = run + before;
    dstart[tid] = run2;
  }
  // ranks inside the tile: wave w owns positions [w*seg, (w+1)*seg), round r holds w*seg + r*64 + lane
  const unsigned long long* kin = src + ((int64_t)blockIdx.x * n_sub + tile) * P;
  unsigned long long k[16];
  unsigned off[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (r < R) k[r] = kin[wave * seg + r * kWave + lane];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < R) {
      const unsigned dg = (unsigned)(k[r] >> shift) & dmask;
      unsigned long long peers = ~0ull;
      for (int b = 0; b < bits; ++b) {
        const bool bit = (dg >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
      }
      const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
      const unsigned cnt = (unsigned)__popcll(peers);
      unsigned* hp = whist + dg * kPrepWaves + wave;
      const unsigned base = *hp;                       // same value for every peer ...
      __builtin_amdgcn_wave_barrier();
      if (rank == 0) *hp = base + cnt;                 // ... then the lowest peer advances the counter
      __builtin_amdgcn_wave_barrier();
      off[r] = base + rank;
    }
  }
  __syncthreads();
  if (tid < radix) {                                   // same digit in earlier waves of this tile
    unsigned run = 0;
    for (int w = 0; w < kPrepWaves; ++w) { const unsigned c = whist[tid * kPrepWaves + w]; whist[tid * kPrepWaves + w] = run; run += c; }
  }
  __syncthreads();
  // The tile is first put in digit order in LDS, then written out in that order: consecutive lanes then write
  // consecutive keys of one digit's run (about P / radix = 128 keys = 1 KiB per run) instead of 64 scattered 8-byte
  // stores per wave instruction -- the scattered form moved 0.3 TB/s and was most of the sort's time.
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < R) {
      const unsigned dg = (unsigned)(k[r] >> shift) & dmask;
      lkeys[dstart[dg] + whist[dg * kPrepWaves + wave] + off[r]] = k[r];
    }
  }
  __syncthreads();
  unsigned long long* out = dst + (int64_t)blockIdx.x * n_sub * P;
  for (int i = tid; i < P; i += kPrepThreads) {
    const unsigned long long kk = lkeys[i];
    const unsigned dg = (unsigned)(kk >> shift) & dmask;
    out[gbase[dg] + ((unsigned)i - dstart[dg])] = kk;
  }
}

// ---------------------------------------------------------------- work items of one sorted tile
// LDS: keys[P] | wtot[32].  Positions are GLOBAL (tile * P + i) so that runs continue across tiles.
// SHARD (the row-sharded step, ge_shard.hip): rows >= so.R are remote rows in staging order.  Their runs are
// numbered u = 0, 1, .. across the step (remote heads in earlier tiles come from so.tile_heads); every key
// tells its slot where the row is read from (so.pos_src / so.neg_src), a remote head adds its row to the
// request list, a sole remote slot is tagged -3 - u (the producing pair stores its gradient row straight into
// the send buffer) and items of remote rows carry R + u as their row.
template <bool SHARD>
__global__ __launch_bounds__(kPrepThreads) void prep_big_items_kernel(
    const unsigned long long* __restrict__ sorted, TileGeom G, int direct, int32_t* __restrict__ prep, ShardOut so) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int P = G.P, R = P / kPrepThreads, seg = kWave * R, n_sub = G.n_sub;
  int* wtot = reinterpret_cast<int*>(keys + P);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6, tile = blockIdx.y;
  const unsigned long long* g = sorted + (int64_t)blockIdx.x * n_sub * P;
  const int total = n_sub * P, base = tile * P;
  int32_t* rec = prep + (int64_t)blockIdx.x * G.stride;
  int32_t* slot_item = G.off_slot >= 0 ? rec + G.off_slot : nullptr;
  int32_t* subrec = rec + G.off_sub + tile * G.sub_stride;
  int32_t* items = subrec + G.off_items;
  int32_t* islots = subrec + G.off_islots;
  for (int i = tid; i < P; i += kPrepThreads) keys[i] = g[base + i];
  if (tid == 0) {
    // a run that began in an earlier tile: its first position, by binary search over the sorted sequence
    int rs = -1;
    if (tile > 0) {
      const unsigned long long k0 = g[base], kp = g[base - 1];
      const uint32_t r0 = (uint32_t)(k0 >> 32);
      if (k0 != kInvalidKey && (uint32_t)(kp >> 32) == r0) {
        int lo = 0, hi = base - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(g[mid] >> 32) < r0) lo = mid + 1; else hi = mid; }
        rs = lo;
      }
    }
    wtot[31] = rs;
    int before = 0;                                   // remote rows that start in earlier tiles
    if (SHARD)
      for (int t = 0; t < tile; ++t) before += so.tile_heads[(int64_t)blockIdx.x * n_sub + t];
    wtot[30] = before;
  }
  __syncthreads();
  const int rs_in = wtot[31], u_before = wtot[30];
  // i is tile-relative and may leave [0, P): the neighbours come from global memory
  auto key_at = [&](int i) -> unsigned long long {
    if (i >= 0 && i < P) return keys[i];
    const int gi = base + i;
    return (gi >= 0 && gi < total) ? g[gi] : kInvalidKey;
  };
  // (a) run start (global position) of every position: max-scan of head positions
  {
    int carry = -1;
    for (int r = 0; r < R; ++r) {
      const int i = wave * seg + r * kWave + lane;
      const unsigned long long kk = keys[i], kp = key_at(i - 1);
      const bool head = kk != kInvalidKey && (base + i == 0 || (uint32_t)(kp >> 32) != (uint32_t)(kk >> 32));
      const int m = max(wave_incl_max(head ? base + i : -1, lane), carry);
      carry = __shfl(m, kWave - 1, kWave);
    }
    if (lane == 0) wtot[wave] = carry;
  }
  __syncthreads();
  int prev_max = rs_in;
  for (int w = 0; w < wave; ++w) prev_max = max(prev_max, wtot[w]);
  __syncthreads();
  // (b) item starts and remote heads, counted together: low 16 bits items, high 16 bits remote heads (<= P each)
  unsigned start_mask = 0, sole_mask = 0, rhead_mask = 0;
  int inc_loc[16], rs_loc[16];
  {
    int carry = 0, mcarry = prev_max;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        const int i = wave * seg + r * kWave + lane;
        const unsigned long long kk = keys[i], kp = key_at(i - 1), kn = key_at(i + 1);
        const bool valid = kk != kInvalidKey;
        const uint32_t row = (uint32_t)(kk >> 32);
        const bool head = valid && (base + i == 0 || (uint32_t)(kp >> 32) != row);
        // (peer-mapped shards: an own row may be read by another rank's gradient kernel at this very moment, so
        // it is never updated by the producing pair; it waits for the apply kernel behind the cross-rank barrier)
        const bool sole = direct && head && !(kn != kInvalidKey && (uint32_t)(kn >> 32) == row) &&
                          !(SHARD && so.peer && row < (uint32_t)so.R);
        const bool rhead = SHARD && head && row >= (uint32_t)so.R;
        const int rs = max(wave_incl_max(head ? base + i : -1, lane), mcarry);
        mcarry = __shfl(rs, kWave - 1, kWave);
        rs_loc[r] = rs;
        const bool start = valid && ((base + i - rs) % kItemCap) == 0 && !sole;
        if (start) start_mask |= 1u << r;
        if (sole) sole_mask |= 1u << r;
        if (rhead) rhead_mask |= 1u << r;
        const int incl = wave_incl_add((start ? 1 : 0) | (rhead ? 1 << 16 : 0), lane) + carry;
        inc_loc[r] = incl;
        carry = __shfl(incl, kWave - 1, kWave);
      }
    }
    if (lane == 0) wtot[wave] = carry;
  }
  __syncthreads();
  int prev = 0, all = 0;
  for (int w = 0; w < kPrepWaves; ++w) { if (w < wave) prev += wtot[w]; all += wtot[w]; }
  if (tid == 0) subrec[0] = all & 0xFFFF;
  const int prev_items = prev & 0xFFFF, u0 = u_before + (prev >> 16) - 1;   // u = u0 + remote heads at or before
  // (c) emit
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < R) {
      const int i = wave * seg + r * kWave + lane;
      const unsigned long long kk = keys[i];
      const bool valid = kk != kInvalidKey;
      uint32_t row = (uint32_t)(kk >> 32);
      const uint32_t slot = (uint32_t)kk;
      const bool sole = (sole_mask >> r) & 1u;
      int tag = kSlotDirect;
      if (SHARD && valid) {
        const bool remote = row >= (uint32_t)so.R;
        const int u = u0 + (inc_loc[r] >> 16);
        uint32_t src = row;                             // where the pair reads this row from
        if (remote) {
          if ((rhead_mask >> r) & 1u) so.req_row[(int64_t)blockIdx.x * total + u] = (int32_t)(row % (uint32_t)so.R);
          if (!so.peer) src = (uint32_t)so.R + (uint32_t)u;
          row = (uint32_t)so.R + (uint32_t)u;
          tag = -3 - u;
        }
        const uint32_t pair = slot / 6u, X = slot - pair * 6u;
        if (X < 3) so.pos_src[((int64_t)blockIdx.x * so.B + pair) * 3 + X] = (int32_t)src;
        else so.neg_src[(int64_t)blockIdx.x * so.B + pair] = (int32_t)((src << 1) | (X - 3u));
      }
      if (sole) slot_item[slot] = tag;
      // Items that start in this round, FOUR per wave instruction, sixteen lanes an item: lane j of a group looks at
      // position start + j (is it the item's row?), the group's 16 flags give the count, the slot list leaves as ONE
      // 64-byte store.  (One lane an item -- a 16-step walk over its positions, then 16 dword stores per lane -- was 140
      // of this kernel's 200 us: measured by ablation in round 4.)
      const bool is_start = (start_mask >> r) & 1u;
      unsigned long long sm = __ballot(is_start);
      const int idx_me = prev_items + (inc_loc[r] & 0xFFFF) - 1;
      const int g16 = lane >> 4, j16 = lane & 15;
      while (sm) {                                     // (wave-uniform)
        int srcl = -1;                                 // the start lane this 16-lane group serves
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int bpos = sm ? __builtin_ctzll(sm) : -1;
          if (sm) sm &= sm - 1;
          if (g == g16) srcl = bpos;
        }
        const int sl = srcl < 0 ? 0 : srcl;
        const int i_s = wave * seg + r * kWave + sl;
        const int idx_s = __shfl(idx_me, sl, kWave);
        const int row_s = __shfl((int)row, sl, kWave);   // (as stored in the item: a remote row's staging id R + u)
        const int rs_s = __shfl(rs_loc[r], sl, kWave);
        const unsigned long long k0 = key_at(i_s);
        const uint32_t krow = (uint32_t)(k0 >> 32);
        const unsigned long long kj = key_at(i_s + j16);
        const bool same = srcl >= 0 && kj != kInvalidKey && (uint32_t)(kj >> 32) == krow;
        const unsigned m16 = (unsigned)((__ballot(same) >> (16 * g16)) & 0xFFFFull);
        const int cnt = m16 == 0xFFFFu ? kItemCap : __builtin_ctz(~m16);   // the item's positions are consecutive: a prefix
        if (srcl >= 0) {
          islots[idx_s * kItemCap + j16] = j16 < cnt ? (int32_t)(uint32_t)kj : -1;
          if (j16 == 0) {
            bool more = false;
            if (cnt == kItemCap) { const unsigned long long ke = key_at(i_s + kItemCap); more = ke != kInvalidKey && (uint32_t)(ke >> 32) == krow; }
            const bool multi = (base + i_s != rs_s) || more;
            items[2 * idx_s] = row_s;
            items[2 * idx_s + 1] = cnt | (multi ? (1 << 30) : 0);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- pairs in relation order
// order[B] of a hinge step: the pairs sorted by their relation row (stable).  The gradient kernel walks the pairs
// in this order, a contiguous run per wave, so that consecutive pairs of a wave share their relation and the
// relation row's gradient can be summed in registers (complex_hinge_grad_kernel).  Keys (relation << 32 | pair)
// in tiles of kOrderP, sorted by the same passes as the slot keys.  T == 0: positives given as [steps][B][3].
constexpr int kOrderP = 16384;
__global__ __launch_bounds__(kPrepThreads) void order_keys_kernel(
    const int32_t* __restrict__ triples, int64_t T, int64_t first_row, int64_t B, int64_t s0, int64_t N, int n_o,
    unsigned long long* __restrict__ keys_out, unsigned* __restrict__ limit) {
  const int64_t s = blockIdx.x;
  unsigned rmax = 0;
  const int32_t* pos = T > 0 ? triples + 3 * step_row(first_row, T, B, s0 + s) : triples + s * 3 * B;
  unsigned long long* out = keys_out + (s * n_o + blockIdx.y) * kOrderP;
  for (int i = threadIdx.x; i < kOrderP; i += kPrepThreads) {
    const int64_t j = (int64_t)blockIdx.y * kOrderP + i;
    unsigned long long k = kInvalidKey;
    if (j < B) {
      int32_t r = pos[3 * j + 2];
      if (r < 0 || r >= N) r = (int32_t)N;            // invalid pairs last (they contribute nothing anyway)
      k = ((unsigned long long)(uint32_t)r << 32) | (uint32_t)j;
      rmax = max(rmax, (unsigned)r);
    }
    out[i] = k;
  }
  // the largest relation id of the chunk: what the sort's passes have to cover (one atomic per wave)
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) rmax = max(rmax, (unsigned)__shfl_xor((int)rmax, m, kWave));
  if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(limit, rmax);
}

__global__ __launch_bounds__(kPrepThreads) void order_write_kernel(
    const unsigned long long* __restrict__ sorted, int n_o, int64_t B, int64_t stride, int64_t off_order,
    int32_t* __restrict__ prep) {
  const unsigned long long* g = sorted + (int64_t)blockIdx.x * n_o * kOrderP;
  int32_t* order = prep + (int64_t)blockIdx.x * stride + off_order;
  for (int64_t i = (int64_t)blockIdx.y * kPrepThreads + threadIdx.x; i < B; i += (int64_t)gridDim.y * kPrepThreads)
    order[i] = (int32_t)(uint32_t)g[i];
}

// ---------------------------------------------------------------- host
static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

// scratch of one sort over n steps of n_sub tiles of P keys: two key arrays (ping-pong) + the digit counts of one pass
size_t sort_scratch_bytes(int64_t n, int64_t n_sub, int64_t P) {
  const size_t keys = align_up_sz(sizeof(unsigned long long) * (size_t)n * (size_t)n_sub * (size_t)P, 256);
  const size_t hist = align_up_sz(sizeof(unsigned) * (size_t)n * kMaxRadix * (size_t)n_sub, 256);
  return 2 * keys + hist + 256;        // + the `limit` word of a sort whose key kernel reports its largest row field
}
static unsigned* sort_scratch_limit(void* scratch, int64_t n, int64_t n_sub, int64_t P) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(scratch) + sort_scratch_bytes(n, n_sub, P) - 256);
}
unsigned long long* sort_scratch_keys(void* scratch) { return reinterpret_cast<unsigned long long*>(scratch); }

// stable LSD radix sort of the keys in scratch (as written by a key kernel) on their row field (< n_rows);
// returns the array that holds the result
const unsigned long long* sort_tiles_launch(void* scratch, int64_t n, int64_t n_sub, int64_t P, int64_t n_rows, hipStream_t st,
                                            const unsigned* limit) {
  const SortBits sb = sort_bits_for(n_rows);
  const size_t keys_bytes = align_up_sz(sizeof(unsigned long long) * (size_t)n * (size_t)n_sub * (size_t)P, 256);
  unsigned long long* ka = reinterpret_cast<unsigned long long*>(scratch);
  unsigned long long* kb = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + keys_bytes);
  unsigned* hist = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(scratch) + 2 * keys_bytes);
  const dim3 grid((unsigned)n, (unsigned)n_sub), block(kPrepThreads);
  const size_t lds = sizeof(unsigned long long) * (size_t)P;      // the tile in digit order (128 KiB at P = 16,384)
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(prep_big_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          140 * 1024) != hipSuccess) return nullptr;
  for (int pass = 0; pass < sb.n_pass; ++pass) {
    const int shift = 32 + pass * sb.bits;
    hipLaunchKernelGGL(prep_big_hist_kernel, grid, block, 0, st, ka, (int)P, (int)n_sub, shift, sb.bits, hist, limit);
    hipLaunchKernelGGL(prep_big_scatter_kernel, grid, block, lds, st, ka, kb, hist, (int)P, (int)n_sub, shift, sb.bits, limit);
    unsigned long long* t = ka; ka = kb; kb = t;
  }
  return ka;
}

int items_launch(const unsigned long long* sorted, int64_t n, const TileGeom& G, int direct, int32_t* out, const ShardOut* so,
                 hipStream_t st) {
  const size_t lds = sizeof(unsigned long long) * (size_t)G.P + sizeof(int) * 32;
  const dim3 grid((unsigned)n, (unsigned)G.n_sub), block(kPrepThreads);
  if (so) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prep_big_items_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(prep_big_items_kernel<true>, grid, block, lds, st, sorted, G, direct, out, *so);
  } else {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(prep_big_items_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(prep_big_items_kernel<false>, grid, block, lds, st, sorted, G, direct, out, ShardOut{});
  }
  return launch_status();
}

// order[B] of n steps into their records (scratch: that of the slot-key sort, which is done with it by now)
int relation_order_launch(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t s0, int64_t n, int64_t N,
                          int32_t* out, int64_t stride, int64_t off_order, void* scratch, hipStream_t st) {
  const int n_o = (int)((B + kOrderP - 1) / kOrderP);
  // (the scratch is the slot-key sort's, sized for more tiles: the limit word sits where THIS sort's geometry puts it)
  unsigned* limit = sort_scratch_limit(scratch, n, n_o, kOrderP);
  if (hipMemsetAsync(limit, 0, sizeof(unsigned), st) != hipSuccess) return launch_status();
  hipLaunchKernelGGL(order_keys_kernel, dim3((unsigned)n, (unsigned)n_o), dim3(kPrepThreads), 0, st, triples, T, first_row, B,
                     s0, N, n_o, sort_scratch_keys(scratch), limit);
  const unsigned long long* sorted = sort_tiles_launch(scratch, n, n_o, kOrderP, N + 1, st, limit);
  const int gy = (int)std::min<int64_t>((B + kPrepThreads - 1) / kPrepThreads, 64);
  hipLaunchKernelGGL(order_write_kernel, dim3((unsigned)n, (unsigned)gy), dim3(kPrepThreads), 0, st, sorted, n_o, B, stride,
                     off_order, out);
  return launch_status();
}

size_t prep_big_scratch_bytes(int64_t B, int64_t negs, int64_t n) {
  const PrepLayout L = prep_layout(B, negs);
  return L.n_sub <= 1 ? 0 : sort_scratch_bytes(n, L.n_sub, L.P);
}

int prepare_big_launch(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t s0, int64_t n,
                       const int32_t* id_to_type, int64_t N, const int64_t* type_offsets, int32_t n_types,
                       const int32_t* type_ids, uint64_t seed, uint64_t global_step0, int32_t padded_size,
                       int32_t mode, int direct, int32_t* out, void* scratch, hipStream_t st, int negs) {
  const PrepLayout L = prep_layout(B, negs);
  const dim3 grid((unsigned)n, (unsigned)L.n_sub), block(kPrepThreads);
  hipLaunchKernelGGL(prep_big_keys_kernel, grid, block, 0, st, triples, T, first_row, B, s0, id_to_type, N, type_offsets,
                     n_types, type_ids, seed, global_step0, padded_size, mode, direct, negs, out, sort_scratch_keys(scratch));
  const unsigned long long* sorted = sort_tiles_launch(scratch, n, L.n_sub, L.P, N, st, nullptr);
  const int rc = items_launch(sorted, n, geom_of(L), direct, out, nullptr, st);
  if (rc || L.off_order < 0) return rc;
  return relation_order_launch(triples, T, first_row, B, s0, n, N, out, L.stride, L.off_order, scratch, st);
}

}  // namespace ge
