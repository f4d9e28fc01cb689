// ge_train.hip -- the native inner training loop (holE.py:340-362 minus validation).
//
// Negatives depend only on (seed, step, positives) -- never on the table -- so everything about a
// step except its floating-point work can be prepared ahead, in bulk, for a chunk of steps:
//
//   train_prepare_kernel   one 1024-thread workgroup per step of <= 4096 units: draws the
//                          negatives (holE.py:97-140, 343-347), builds the (row, IndexedSlices slot)
//                          list of the <= 4 gradient rows per pair, sorts it by row with a stable LSD
//                          radix sort held entirely in LDS (wave-level multisplit: ballot ranks +
//                          per-wave digit counters, 2 passes for FB15k's 16,296 rows, 3 for 1.2 M rows)
//                          and cuts it into work items of <= 16 slots of one row.  Larger steps: the
//                          same sort across workgroups, ge_prep_big.hip.
//   (per step) hinge_grad  fused gather -> clip -> score -> sigmoid -> hinge -> gradient rows
//   (per step) apply_sorted_kernel   one wavefront per work item sums its gradient rows and
//                          updates the table row ONCE with a plain read-modify-write.
//
// This replaces the float-atomic ScatterSub (memory-side atomics ~1.3 TB/s chip-wide and an order
// of magnitude slower when many waves hit one hot row -- Zipfian heads, a handful of relations)
// by coalesced row reads plus one write per distinct row, and makes the update order fixed:
// rows with <= 16 occurrences in a step (the vast majority) are bitwise reproducible at any batch size.
// Only rows split over several work items combine their partial sums with atomics.
//
// The prepared records live in the caller's workspace, two chunks of steps (double buffer).  With a
// pipeline handle (ge_train_pipeline_create) the prepare launches run on the handle's side stream one
// chunk AHEAD of the steps that consume them -- across ge_train_steps calls too: a call that continues
// where the previous one stopped finds its records already built.  Without a handle the launches go
// to the caller's stream (one ~20 us launch per chunk of steps) and the library keeps no state at all.
#include "ge_prep.h"
#include <cmath>

namespace ge {

int complex_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr, const int32_t* slot_item = nullptr, float* table_rw = nullptr, int spectral = 0, const int32_t* order = nullptr);
int hole_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int scatter_add_rows_launch(float*, int64_t, int32_t, const int32_t*, const float*, int64_t, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int corrupt_batch_launch(const int32_t*, int64_t, const int32_t*, int64_t, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int32_t*, hipStream_t);
int hole_spectral_launch(float*, int64_t, int32_t, int, hipStream_t);
size_t prep_big_scratch_bytes(int64_t B, int64_t negs, int64_t n);
int prepare_big_launch(const int32_t*, int64_t, int64_t, int64_t, int64_t, int64_t, const int32_t*, int64_t, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int, int32_t*, void*, hipStream_t, int);

// grid (steps in the chunk, n_sub).  LDS: keys[P] (8 B) | hist[16 waves x 256 digits] | wtot[32].
__global__ __launch_bounds__(kPrepThreads) void train_prepare_kernel(
    const int32_t* __restrict__ triples, int64_t T, int64_t first_row, int64_t B, int64_t s0,
    const int32_t* __restrict__ id_to_type, int64_t N, const int64_t* __restrict__ type_offsets,
    int32_t n_types, const int32_t* __restrict__ type_ids, uint64_t seed, uint64_t global_step0,
    int32_t padded_size, int32_t mode, int direct, int n_pass, int bits, int negs, int32_t* __restrict__ prep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const PrepLayout L = prep_layout(B, negs);
  const int P = (int)L.P, R = P / kPrepThreads;   // R = keys per lane per wave segment (1..16)
  unsigned* hist = reinterpret_cast<unsigned*>(keys + P);
  int* wtot = reinterpret_cast<int*>(hist + kPrepWaves * kMaxRadix);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const int sub = blockIdx.y;
  const int64_t s = s0 + blockIdx.x;
  const int32_t* pos = triples + 3 * step_row(first_row, T, B, s);
  int32_t* rec = prep + (int64_t)blockIdx.x * L.stride;
  int32_t* neg = rec;
  int32_t* slot_item = rec + L.off_slot;
  int32_t* subrec = rec + L.off_sub + sub * L.sub_stride;
  int32_t* items = subrec + L.off_items;
  int32_t* islots = subrec + L.off_islots;
  const int64_t i0 = (int64_t)sub * kSub;
  const int S = (int)((L.units - i0) < kSub ? (L.units - i0) : kSub);     // units of this sub-batch
  const uint64_t step = global_step0 + (uint64_t)s;
  const bool batch_heads = (mode == GE_CORRUPT_BATCH_COIN) ? batch_coin_heads(seed, step) : false;
  constexpr unsigned long long kInvalid = ~0ull;

  if (direct)
    for (int i = tid; i < 6 * S; i += kPrepThreads) slot_item[6 * i0 + i] = -1;

  // ---- phase 1: negatives + sort keys, in slot order (key index epu*unit + k  <->  ascending slot id)
  {
    const StepSource src{pos, B, id_to_type, N, type_offsets, n_types, type_ids, seed, step, padded_size, mode, negs};
    for (int il = tid; il < (int)L.S; il += kPrepThreads) {
      unsigned long long k4[4] = {kInvalid, kInvalid, kInvalid, kInvalid};
      if (il < S) unit_keys(src, i0 + il, batch_heads, neg, k4);
      if (negs > 0) {
#pragma unroll
        for (int X = 0; X < 3; ++X) keys[3 * il + X] = k4[X];
      } else {
#pragma unroll
        for (int X = 0; X < 4; ++X) keys[4 * il + X] = k4[X];
      }
    }
  }
  __syncthreads();

  // ---- phase 2: stable LSD radix sort by row.  Wave w owns the contiguous segment [w*64R, (w+1)*64R);
  // in round r its lanes hold elements w*64R + r*64 + lane.  Rank inside (wave, digit) = the wave's
  // running digit counter + the number of lower lanes with the same digit in this round (ballots).
  const int radix = 1 << bits;
  const unsigned dmask = (unsigned)radix - 1u;
  const int seg = kWave * R;
  for (int pass = 0; pass < n_pass; ++pass) {
    const int shift = 32 + pass * bits;
    for (int i = tid; i < radix * kPrepWaves; i += kPrepThreads) hist[i] = 0;
    __syncthreads();
    unsigned long long k[16];
    unsigned off[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        k[r] = keys[wave * seg + r * kWave + lane];
        const unsigned dg = (unsigned)(k[r] >> shift) & dmask;
        unsigned long long peers = ~0ull;
        for (int b = 0; b < bits; ++b) {
          const bool bit = (dg >> b) & 1u;
          const unsigned long long bal = __ballot(bit);
          peers &= bit ? bal : ~bal;
        }
        const unsigned rank = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        const unsigned cnt = (unsigned)__popcll(peers);
        unsigned* hp = hist + dg * kPrepWaves + wave;
        const unsigned base = *hp;                       // same value for every peer ...
        __builtin_amdgcn_wave_barrier();
        if (rank == 0) *hp = base + cnt;                 // ... then the lowest peer advances the counter
        __builtin_amdgcn_wave_barrier();
        off[r] = base + rank;
      }
    }
    __syncthreads();
    {  // exclusive scan of hist in (digit, wave) order
      const int q = radix * kPrepWaves / kPrepThreads;   // 1, 2 or 4 consecutive counters per thread
      unsigned loc[4];
      int sum = 0;
      for (int u = 0; u < q; ++u) { loc[u] = hist[tid * q + u]; sum += (int)loc[u]; }
      const int incl = wave_incl_add(sum, lane);
      if (lane == kWave - 1) wtot[wave] = incl;
      __syncthreads();
      int run = incl - sum;
      for (int w = 0; w < wave; ++w) run += wtot[w];
      for (int u = 0; u < q; ++u) { hist[tid * q + u] = (unsigned)run; run += (int)loc[u]; }
    }
    __syncthreads();   // every key is in registers: the scatter may overwrite the one LDS buffer
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        const unsigned dg = (unsigned)(k[r] >> shift) & dmask;
        keys[hist[dg * kPrepWaves + wave] + off[r]] = k[r];
      }
    }
    __syncthreads();
  }

  // ---- phase 3: work items.  Every run of equal rows is cut into pieces of <= kItemCap entries counted
  // from the start of the run (a row with <= kItemCap occurrences is always exactly one item).  With
  // `direct`, a row with exactly ONE gradient slot in the step has one reader and one writer -- the same
  // pair: it is not queued, its slot is tagged kSlotDirect and the producing pair updates the table row.
  auto key_at = [&](int i) -> unsigned long long { return (i >= 0 && i < P) ? keys[i] : kInvalid; };
  int rs_loc[16];          // (a) run start of every position: max-scan of head positions
  {
    int carry = -1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        const int i = wave * seg + r * kWave + lane;
        const unsigned long long kk = keys[i], kp = key_at(i - 1);
        const bool head = kk != kInvalid && (i == 0 || (uint32_t)(kp >> 32) != (uint32_t)(kk >> 32));
        const int m = max(wave_incl_max(head ? i : -1, lane), carry);
        rs_loc[r] = m;
        carry = __shfl(m, kWave - 1, kWave);
      }
    }
    if (lane == 0) wtot[wave] = carry;
  }
  __syncthreads();
  int prev_max = -1;
  for (int w = 0; w < wave; ++w) prev_max = max(prev_max, wtot[w]);
  __syncthreads();
  unsigned start_mask = 0;   // (b) item starts, counted
  int idx_loc[16];
  {
    int carry = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < R) {
        const int i = wave * seg + r * kWave + lane;
        const unsigned long long kk = keys[i], kp = key_at(i - 1), kn = key_at(i + 1);
        const bool valid = kk != kInvalid;
        const uint32_t row = (uint32_t)(kk >> 32);
        const bool head = valid && (i == 0 || (uint32_t)(kp >> 32) != row);
        const bool sole = direct && head && !(kn != kInvalid && (uint32_t)(kn >> 32) == row);
        const int rs = max(rs_loc[r], prev_max);
        rs_loc[r] = rs;
        const bool start = valid && ((i - rs) % kItemCap) == 0 && !sole;
        if (start) start_mask |= 1u << r;
        if (sole) slot_item[(uint32_t)kk] = kSlotDirect;
        const int incl = wave_incl_add(start ? 1 : 0, lane) + carry;
        idx_loc[r] = incl - (start ? 1 : 0);
        carry = __shfl(incl, kWave - 1, kWave);
      }
    }
    if (lane == 0) wtot[wave] = carry;
  }
  __syncthreads();
  int prev_items = 0, total = 0;
  for (int w = 0; w < kPrepWaves; ++w) { if (w < wave) prev_items += wtot[w]; total += wtot[w]; }
  if (tid == 0) subrec[0] = total;
  // (c) emit
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (r < R && ((start_mask >> r) & 1u)) {
      const int i = wave * seg + r * kWave + lane;
      const uint32_t row = (uint32_t)(keys[i] >> 32);
      const int idx = prev_items + idx_loc[r];
      int e = i + 1;
      while (e < P && (e - i) < kItemCap && keys[e] != kInvalid && (uint32_t)(keys[e] >> 32) == row) ++e;
      const bool more = (e < P && keys[e] != kInvalid && (uint32_t)(keys[e] >> 32) == row);
      const bool multi = (i != rs_loc[r]) || more;
      items[2 * idx] = (int32_t)row;
      items[2 * idx + 1] = (e - i) | (multi ? (1 << 30) : 0);
#pragma unroll
      for (int j = 0; j < kItemCap; ++j)
        islots[idx * kItemCap + j] = (i + j < e) ? (int32_t)(uint32_t)keys[i + j] : -1;
    }
  }
}

// One wavefront per work item: sum the item's gradient rows (skipping slots whose pair was
// hinge-inactive: grad_idx < 0), then table[row] += sum -- plain RMW when the row has a single
// item, atomics when it was split over several (> 16 slots).  blockIdx.y = tile of the step's sorted key
// sequence; tiles never share a single-item row (the sort is over the whole step, ge_prep_big.hip).
// Lane l owns columns l, l+64, ... (256 contiguous bytes per wave instruction for loads, stores and
// atomics alike).
// COND (steps of several tiles): a gradient row is requested only if its slot is live -- there most relation slots are
// empty (their sum sits in the run's last slot, complex_hinge_grad_kernel) and bandwidth, not the length of the
// dependent-load chain, is what the kernel runs out of; one-tile steps request every listed row together with the flags.
// DET (opt-in, GE_STEP_DETERMINISTIC): an item of a row that is split over several items does not ADD its partial sum to
// the table (float atomics: the order of the items' adds is whatever the scheduler makes it) but STORES it over the gradient
// row of its first slot; hot_rows_kernel then sums a row's partials in item order and does the one read-modify-write.
template <int NJ, bool COND, bool DET = false>
__global__ __launch_bounds__(kBlock) void apply_sorted_kernel(
    float* __restrict__ table, int d, const int32_t* __restrict__ sub0, int64_t sub_stride, int off_items,
    int off_islots, const int32_t* __restrict__ grad_idx, const float* grad_val, int split,
    float* __restrict__ out2) {
  // grad_idx == null: every listed slot is live (the owner side of the row-sharded step sums received rows).
  // Rows >= split (row-sharded step only) are rows of ANOTHER owner: their sum is stored to row - split of the
  // send buffer out2 (zeroed beforehand: split rows add atomically) instead of being added to the table.
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int32_t* subrec = sub0 + (int64_t)blockIdx.y * sub_stride;
  const int32_t* items = subrec + off_items;
  const int32_t* islots = subrec + off_islots;
  const int n_items = subrec[0];
  for (int w = wave; w < n_items; w += nwaves) {
    // two independent loads first: the item header and its inline slot list (lane o < 16 -> slot o)
    const int row = items[2 * w], cm = items[2 * w + 1];
    int slot_v = (lane < kItemCap) ? islots[w * kItemCap + lane] : -1;
    const int cnt = cm & 0x3FFFFFFF;
    const bool multi = (cm >> 30) & 1;
    const bool away = row >= split;
    float* dst = away ? out2 + (int64_t)(row - split) * d : table + (int64_t)row * d;
    // the table row is fetched now, under the gradient-row loads, not after them
    float base[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + kWave * j;
      base[j] = (!multi && !away && c < d) ? dst[c] : 0.f;
    }
    const bool act_v = slot_v >= 0 && (!grad_idx || grad_idx[slot_v] >= 0);   // pair was hinge-active
    if (slot_v < 0) slot_v = 0;
    const unsigned long long live = __ballot(act_v);
    const bool park = DET && multi && !away;           // this item's sum is parked for the ordered pass (zeros included)
    if (live == 0ull && !(away && !multi) && !park) continue;  // wave-uniform (a remote row is sent whatever it sums to)
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    for (int o = 0; o < cnt; o += 4) {
      int sl[4];
      bool on[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int src_lane = (o + q < cnt) ? (o + q) : o;
        sl[q] = __shfl(slot_v, src_lane, kWave);
        on[q] = (o + q < cnt) && ((live >> (o + q)) & 1ull);
      }
      float v[4][NJ];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* src = grad_val + (int64_t)sl[q] * d;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int c = lane + kWave * j;
          v[q][j] = (c < d && (!COND || on[q])) ? src[c] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] += on[q] ? v[q][j] : 0.f;
    }
    float* parked = const_cast<float*>(grad_val) + (int64_t)__shfl(slot_v, 0, kWave) * d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + kWave * j;
      if (c < d) {
        if (park) parked[c] = acc[j];
        else if (multi) atomic_add_f32(dst + c, acc[j]);
        else dst[c] = base[j] + acc[j];
      }
    }
  }
}

// The ordered pass of GE_STEP_DETERMINISTIC: every row that is split over several items (> 16 gradient slots in the step;
// its items are consecutive in the sorted item lists, possibly across a tile boundary) is reduced by ONE workgroup in a
// fixed order and gets its single read-modify-write: a row of fewer than 64 items by ONE wave (its parked partials in item
// order), a longer one by the eight waves of the workgroup (wave j: items j, j + 8, j + 16, ... in that order; the wave sums
// added in wave order).  Workgroup (b, t) looks at items [64 b, 64 b + 64)
// of tile t for HEADS (a multi item whose predecessor belongs to another row) and reduces the rows that start there.
// A wave looks its items up 64 at a time (lane l: the wave's l-th item -- header and first slot, two parallel loads) and
// then streams their parked rows eight at a time: per item a fraction of a memory round trip, not three.
constexpr int kHotBlock = 512;      // (1024 threads a workgroup -- 16 waves a hot row -- was slower: 39.6 vs 20 us at 65,536 pairs: the launch of the
                                    // mostly idle big workgroups, not the reduction, is what the pass costs)
template <int NJ>
__global__ __launch_bounds__(kHotBlock) void hot_rows_kernel(
    float* __restrict__ table, int d, const int32_t* __restrict__ sub0, int64_t sub_stride, int off_items, int off_islots,
    int n_sub, const float* __restrict__ grad_val) {
  constexpr int kScan = kWave;                        // items a workgroup looks at for heads
  constexpr int kWv = kHotBlock / kWave;
  __shared__ int heads[kScan];
  __shared__ int n_heads;
  __shared__ float part[kWv - 1][NJ * kWave];
  const int t = threadIdx.x, lane = t & (kWave - 1), wv = t >> 6, tile = blockIdx.y;
  auto sub_of = [&](int tl) { return sub0 + (int64_t)tl * sub_stride; };
  const int32_t* subrec = sub_of(tile);
  const int n_items = subrec[0];
  for (int win = blockIdx.x; win * kScan < n_items; win += gridDim.x) {    // (block-uniform trip count)
  if (t == 0) n_heads = 0;
  __syncthreads();
  const int k = win * kScan + t;
  if (t < kScan && k < n_items) {
    const int32_t* items = subrec + off_items;
    const int row = items[2 * k], cm = items[2 * k + 1];
    if ((cm >> 30) & 1) {
      int prev_row = -1;
      if (k > 0) prev_row = items[2 * (k - 1)];
      else if (tile > 0) { const int32_t* ps = sub_of(tile - 1); const int pn = ps[0]; if (pn > 0) prev_row = ps[off_items + 2 * (pn - 1)]; }
      if (prev_row != row) heads[atomicAdd(&n_heads, 1)] = k;   // (any order: every head is reduced independently)
    }
  }
  __syncthreads();
  const int nh = n_heads;
  // the item at flattened position k0 + r of the row that starts at item k0 of this tile: (belongs to the row?, first slot)
  auto item_at = [&](int k0, int r, int row, int& first_slot) -> bool {
    int idx = k0 + r, tl = tile;
    const int32_t* sr = subrec;
    int ni = n_items;
    while (idx >= ni) {
      idx -= ni; ++tl;
      if (tl >= n_sub) return false;
      sr = sub_of(tl); ni = sr[0];
    }
    first_slot = sr[off_islots + idx * kItemCap];
    return sr[off_items + 2 * idx] == row;             // the row's items are consecutive: valid r form a prefix
  };
  // n parked rows (first slots in lanes 0 .. n-1 of `fs`) added to acc in lane order, kInFlight requests in flight
  constexpr int kInFlight = NJ <= 4 ? 16 : 8;
  auto add_rows = [&](int fs, int n, float (&acc)[NJ]) {
    for (int o = 0; o < n; o += kInFlight) {
      float v[kInFlight][NJ];
#pragma unroll
      for (int q = 0; q < kInFlight; ++q)
        if (o + q < n) {
          const float* src = grad_val + (int64_t)__shfl(fs, o + q, kWave) * d;
#pragma unroll
          for (int j = 0; j < NJ; ++j) { const int c = lane + kWave * j; v[q][j] = src[c < d ? c : 0]; }
        }
#pragma unroll
      for (int q = 0; q < kInFlight; ++q)
        if (o + q < n) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[j] += v[q][j];
        }
    }
  };
  // (1) rows of fewer than 64 items (nearly all): one WAVE a row, the window's heads spread over the waves
  for (int h = wv; h < nh; h += kWv) {
    const int k0 = heads[h];
    const int row = (subrec + off_items)[2 * k0];
    int fs = 0;
    const bool mine = item_at(k0, lane, row, fs);
    const int n = __popcll(__ballot(mine));
    if (n == kWave) continue;                          // a long row: step (2)
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    add_rows(fs, n, acc);
    float* dst = table + (int64_t)row * d;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int c = lane + kWave * j; if (c < d) dst[c] = dst[c] + acc[j]; }
    if (lane == 0) heads[h] = -1;                       // done
  }
  __syncthreads();
  // (2) rows of 64 items and more (the relation rows of a large step): the whole workgroup, wave j taking items j, j + 8, ...
  for (int h = 0; h < nh; ++h) {
    const int k0 = heads[h];
    if (k0 < 0) continue;                              // (block-uniform: heads[] is read after the barrier)
    const int row = (subrec + off_items)[2 * k0];
    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
    for (int r0 = wv; ; r0 += kWv * kWave) {
      int fs = 0;
      const bool mine = item_at(k0, r0 + kWv * lane, row, fs);
      const int n = __popcll(__ballot(mine));
      add_rows(fs, n, acc);
      if (n < kWave) break;                            // wave-uniform
    }
    if (wv > 0) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) part[wv - 1][j * kWave + lane] = acc[j];
    }
    __syncthreads();
    if (wv == 0) {
      float* dst = table + (int64_t)row * d;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = lane + kWave * j;
        float tot = acc[j];
#pragma unroll
        for (int w2 = 0; w2 < kWv - 1; ++w2) tot += part[w2][j * kWave + lane];
        if (c < d) dst[c] = dst[c] + tot;
      }
    }
    __syncthreads();
  }
  __syncthreads();
  }
}

// The same work item for steps of several tiles (large batches, the row-sharded step), where the kernel is a few hundred
// thousand short waves whose time is their chain of dependent memory round trips, not their bytes:
//  * the item header comes on the scalar path (s_load: one item per wave, so it is wave-uniform);
//  * a row is ONE request per wave -- lane l holds VW consecutive columns (d = 200: 50 lanes x 16 bytes) -- instead of
//    four 256-byte ones, addressed from scalar registers (v_readlane of the slot + scalar arithmetic);
//  * the rows of live slots are requested behind wave-uniform branches on the liveness mask, four at a time.
// Measured at 65,536 pairs on the 960 MB table (round 4, alternated on one box): 47 -> 41 us.  What did NOT help: 8 or
// 16 rows in flight per wave (45 / 51 us: fewer waves per SIMD), requesting every listed row before the liveness flags
// are back (one round trip less, dead rows fetched for nothing: 80 -> 91 us in the 16-row form), a quarter or a
// sixteenth of the workgroups with several items a wave (38-44 us, inside the noise).  Rows split over several items
// keep the column-per-lane layout: float atomics run at full rate only on 256 contiguous bytes per instruction
// (16-byte lanes there: 80 us).
// Summation order and arithmetic are the old kernel's: 0 + g_0 + g_1 + ... in slot order, then row + sum.
template <int VW, int NJ, bool DET = false>
__global__ __launch_bounds__(kBlock) void apply_rows_kernel(
    float* __restrict__ table, int d, const int32_t* __restrict__ sub0, int64_t sub_stride, int off_items,
    int off_islots, const int32_t* __restrict__ grad_idx, const float* grad_val, int split,
    float* __restrict__ out2) {
  typedef const __attribute__((address_space(4))) int32_t* kptr_t;
  // row registers a lane: 16 (4 rows of 200 columns in flight per wave; 8 and 16 rows in flight were slower -- 45 / 51 us
  // against 41: more waves per SIMD hide more than more rows per wave)
  constexpr int RG0 = 16 / (VW * NJ);
  constexpr int RG = RG0 > kItemCap ? kItemCap : (RG0 < 2 ? 2 : RG0);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane((int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  const int nwaves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int32_t* subrec = sub0 + (int64_t)blockIdx.y * sub_stride;
  const kptr_t items = (kptr_t)(uintptr_t)(subrec + off_items);
  const int32_t* islots = subrec + off_islots;
  const int n_items = ((kptr_t)(uintptr_t)subrec)[0];
  int col[NJ];
  bool ok[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { col[j] = VW * (lane + kWave * j); ok[j] = col[j] < d; if (!ok[j]) col[j] = 0; }
  for (int w = wave; w < n_items; w += nwaves) {
    const int row = items[2 * w], cm = items[2 * w + 1];
    int slot_v = (lane < kItemCap) ? islots[w * kItemCap + lane] : -1;
    const int cnt = cm & 0x3FFFFFFF;
    const bool multi = (cm >> 30) & 1;
    const bool away = row >= split;
    float* dst = away ? out2 + (int64_t)(row - split) * d : table + (int64_t)row * d;
    if (multi) {
      // a hot row (> 16 slots: several items): its partial sum is ADDED with float atomics, and those run at full rate
      // only on 256 contiguous bytes per wave instruction (MI355X_MICROARCH.md): column-per-lane layout, as the
      // one-tile kernel's
      constexpr int ND = VW * NJ;
      const bool act_v = slot_v >= 0 && (!grad_idx || grad_idx[slot_v] >= 0);
      const unsigned long long lv = __ballot(act_v);
      const bool park = DET && !away;                  // (see apply_sorted_kernel: the sum is parked for hot_rows_kernel)
      if (lv == 0ull && !park) continue;
      float a[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) a[j] = 0.f;
      for (int o = 0; o < cnt; o += 4) {
        float v[4][ND];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (o + q < cnt && ((lv >> (o + q)) & 1ull)) {
            const float* src = grad_val + (int64_t)__builtin_amdgcn_readlane(slot_v, o + q) * d;
#pragma unroll
            for (int j = 0; j < ND; ++j) { const int c = lane + kWave * j; v[q][j] = src[c < d ? c : 0]; }
          }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (o + q < cnt && ((lv >> (o + q)) & 1ull)) {
#pragma unroll
            for (int j = 0; j < ND; ++j) a[j] += v[q][j];
          }
      }
      float* parked = const_cast<float*>(grad_val) + (int64_t)__builtin_amdgcn_readlane(slot_v, 0) * d;
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        const int c = lane + kWave * j;
        if (c < d) { if (park) parked[c] = a[j]; else atomic_add_f32(dst + c, a[j]); }
      }
      continue;
    }
    // the table row is fetched now, under the gradient-row loads, not after them
    float base[NJ][VW];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!away) load_vec<VW>(dst + col[j], base[j]);
      else {
#pragma unroll
        for (int e = 0; e < VW; ++e) base[j][e] = 0.f;
      }
    }
    float acc[NJ][VW];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) acc[j][e] = 0.f;
    unsigned long long live = 0ull;
    bool have_live = false;
    for (int o = 0; o < cnt; o += RG) {
      float v[RG][NJ][VW];
      if (!have_live) {
        const bool act_v = slot_v >= 0 && (!grad_idx || grad_idx[slot_v] >= 0);   // pair was hinge-active
        live = __ballot(act_v);
        have_live = true;
      }
#pragma unroll
      for (int q = 0; q < RG; ++q)
        if (o + q < cnt && ((live >> (o + q)) & 1ull)) {
          const int sl = __builtin_amdgcn_readlane(slot_v, o + q);
          const float* src = grad_val + (int64_t)sl * d;
#pragma unroll
          for (int j = 0; j < NJ; ++j) load_vec<VW>(src + col[j], v[q][j]);
        }
#pragma unroll
      for (int q = 0; q < RG; ++q)
        if (o + q < cnt && ((live >> (o + q)) & 1ull)) {
#pragma unroll
          for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[j][e] += v[q][j][e];
        }
    }
    if (live == 0ull && !away) continue;              // wave-uniform (a remote row is sent whatever it sums to)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (!ok[j]) continue;
      float o_[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) o_[e] = base[j][e] + acc[j][e];
      store_vec<VW>(dst + col[j], o_);
    }
  }
}

// det (GE_STEP_DETERMINISTIC; rows of this table only, no `split` rows): items of rows split over several items park their
// sums, then hot_rows_kernel reduces every such row in item order (one more launch)
int apply_items_launch(float* table, int d, const TileGeom& G, const int32_t* step_rec, const int32_t* gidx,
                       const float* gval, int split, float* out2, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop,
                       int det) {
  const int grid = grid_for(G.P, kBlock / kWave);  // at most P items per tile
  const int nj = (d + kWave - 1) / kWave;
  const dim3 g((unsigned)grid, (unsigned)G.n_sub);   // (one tile: 1024 workgroups instead of 2048 measured the same, 512 slower)
  // the ordered pass behind either update kernel (det): heads are looked for 64 items a workgroup
  auto hot = [&]() -> int {
    const dim3 gh((unsigned)((G.P + kWave - 1) / kWave), (unsigned)G.n_sub);   // one 64-item window a workgroup (windows past the tile's items return at once)
#define LH(NJ) hipLaunchKernelGGL((hot_rows_kernel<NJ>), gh, dim3(kHotBlock), 0, st, table, d, step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, G.n_sub, gval)
    if (nj <= 1) LH(1); else if (nj <= 2) LH(2); else if (nj <= 4) LH(4); else if (nj <= 8) LH(8); else LH(16);
#undef LH
    return launch_status();
  };
  if (det && nj > 16) return GE_ENOTSUP;
#define LA(NJ)                                                                                                             \
  {                                                                                                                        \
    if (G.n_sub > 1) hipExtLaunchKernelGGL((apply_sorted_kernel<NJ, true>), g, dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, gidx, gval, split, out2); \
    else if (det) { hipExtLaunchKernelGGL((apply_sorted_kernel<NJ, false, true>), g, dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, gidx, gval, split, out2); return hot(); } \
    else hipExtLaunchKernelGGL((apply_sorted_kernel<NJ, false>), g, dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, gidx, gval, split, out2); \
  }
  if (gidx != nullptr || G.n_sub > 1) {
    // whole-row requests (see apply_rows_kernel): steps of several tiles, and -- measured late in round 4, 8.9 -> 7.3-7.7 us at
    // config 2, alternated on one box -- one-tile steps too; the column-per-lane kernel keeps the owner side of the
    // row-sharded step at small request lists (no liveness flags: every listed row is requested with the header)
    auto al = [](const void* q, int a) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % a == 0; };
    const int vw = (d % 4 == 0 && al(table, 16) && al(gval, 16) && al(out2, 16)) ? 4 : (d % 2 == 0 && al(table, 8) && al(gval, 8) && al(out2, 8)) ? 2 : 1;
    const int njr = (d / vw + kWave - 1) / kWave;
    // steps of several tiles: a tile of P keys holds far fewer than P items (65,536 pairs: ~1,250 items of 2-4 rows in each of
    // 16 tiles of 16,384 keys), and waves are launched at about one per clock and XCD -- a grid of one wave per KEY was 131,000
    // waves of which 110,000 found nothing to do.  One wave per 8 keys; a tile with more items than waves takes another round
    // of the item loop.  Update kernel 35.3-36.0 -> 33.4-33.7 us, 15 -> 11 us with no pair hinge-active (alternated on one
    // box, `profiles/r04_ab_gridx.log`; 1024 workgroups a tile: 34.8-34.9, 384: 33.3-33.5).
    const dim3 g((unsigned)(G.n_sub > 1 ? grid_for((G.P + 7) / 8, kBlock / kWave) : grid), (unsigned)G.n_sub);
#define LR(VW, NJ)                                                                                                       \
    {                                                                                                                    \
      if (det) {                                                                                                         \
        hipExtLaunchKernelGGL((apply_rows_kernel<VW, NJ, true>), g, dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, \
                              step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, gidx, gval, split, out2);   \
        return hot();                                                                                                    \
      }                                                                                                                  \
      hipExtLaunchKernelGGL((apply_rows_kernel<VW, NJ>), g, dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, \
                            step_rec + G.off_sub, G.sub_stride, G.off_items, G.off_islots, gidx, gval, split, out2);     \
      return launch_status();                                                                                            \
    }
    if (vw == 4) { if (njr <= 1) LR(4, 1) else if (njr <= 2) LR(4, 2) else if (njr <= 4) LR(4, 4) }
    else if (vw == 2) { if (njr <= 1) LR(2, 1) else if (njr <= 2) LR(2, 2) else if (njr <= 4) LR(2, 4) else if (njr <= 8) LR(2, 8) }
    else { if (njr <= 1) LR(1, 1) else if (njr <= 2) LR(1, 2) else if (njr <= 4) LR(1, 4) else if (njr <= 8) LR(1, 8) else if (njr <= 16) LR(1, 16) }
#undef LR
  }
  if (nj <= 1) LA(1) else if (nj <= 2) LA(2) else if (nj <= 4) LA(4) else if (nj <= 8) LA(8) else if (nj <= 16) LA(16)
  else return GE_ENOTSUP;
#undef LA
  return launch_status();
}

static int apply_sorted_launch(float* table, int d, const PrepLayout& L, const int32_t* step_rec, const int32_t* gidx,
                               const float* gval, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, int det = 0) {
  return apply_items_launch(table, d, geom_of(L), step_rec, gidx, gval, 0x7FFFFFFF, nullptr, st, ev_start, ev_stop, det);
}

static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t hinge_ws_bytes(int64_t B, int32_t d) {
  return align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256) + align_up_sz(sizeof(float) * 6 * (size_t)B * (size_t)d, 256);
}

bool train_fast_ok(int64_t B, int32_t d) { return B >= 1 && B <= (int64_t)1 << 24 && d <= 1024; }

// steps prepared per launch: 32 at B <= 4096 (one record is ~1.2 MB there), fewer for large batches
static int64_t prep_chunk_steps(int64_t B, int64_t negs = 0) {
  const int64_t u = negs > 0 ? (1 + negs) * B : B;
  int64_t c = (32 * kSub) / (u < kSub ? kSub : u);
  // the multi-tile sorts (slot keys, relation order) are ~16 short launches whose latency is the same for 2 steps
  // or 16: a chunk holds enough steps that the sequence (~0.3-0.5 ms) hides behind them -- 16 up to 32,768 units
  // (42 us steps at 16,384), 8 up to 262,144 (19 MB of records each at B = 65,536), 2 beyond
  const int64_t floor_steps = u <= 32768 ? 16 : u <= 4 * 65536 ? 8 : 2;   // (16 and 32 measured at 65,536: no difference)
  return c < floor_steps ? floor_steps : c;
}

// training workspace: [gidx 6B][gval RING x (6B x d)][prep chunk buffer 0][prep chunk buffer 1].
// The gradient rows go to a RING of regions, one per step in turn: a region is written by the grad kernel
// on one XCD and read by the apply kernel on another, and rewriting lines that still sit in another XCD's
// L2 costs ~3.7 us per 13 MB (tools/probes/xcd_locality_probe.hip: 9.1 vs 5.4 us); by the time a region comes
// round again (normally 4 steps later, > the 32 MB of L2 in between) its lines have been evicted and the
// stores take the fast path.
static size_t grad_region_bytes(int64_t B, int32_t d) {
  return align_up_sz(sizeof(float) * 6 * (size_t)B * (size_t)d, 256);
}
static int grad_ring(int64_t B, int32_t d) {
  const size_t reg = grad_region_bytes(B, d);
  // measured at B=4096, d=200 (20 MB regions): 1 -> 22.4, 2 -> 21.3, 4 -> 21.0, 8 -> 21.0, 16 -> 21.8, 32 -> 22.8 us/step:
  // long enough to outlive the L2s, short enough to stay inside the 256 MB Infinity Cache
  size_t r = ((size_t)64 << 20) / reg + 1;
  if (r < 2) r = 2;
  if (r > 16) r = 16;
  if (r < 4 && 4 * reg <= ((size_t)160 << 20)) r = 4;
  return (int)r;
}
static size_t train_grad_bytes(int64_t B, int32_t d) {
  return align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256) + (size_t)grad_ring(B, d) * grad_region_bytes(B, d);
}
static size_t prep_chunk_bytes(int64_t B, int64_t negs = 0) {
  return align_up_sz(sizeof(int32_t) * (size_t)prep_chunk_steps(B, negs) * (size_t)prep_layout(B, negs).stride, 256);
}
size_t train_ws_bytes(int64_t B, int32_t d) {
  if (!train_fast_ok(B, d)) return hinge_ws_bytes(B, d);
  // two chunk buffers + (B > 4096) the key arrays of the multi-tile sort of ONE prepare sequence
  return train_grad_bytes(B, d) + 2 * prep_chunk_bytes(B) + prep_big_scratch_bytes(B, 0, prep_chunk_steps(B));
}

static size_t prep_lds_bytes(const PrepLayout& L) {
  return sizeof(unsigned long long) * (size_t)L.P + sizeof(unsigned) * kPrepWaves * kMaxRadix + sizeof(int) * 32;
}

static int prepare_launch(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t s0, int64_t n,
                          const int32_t* id_to_type, int64_t N, const int64_t* type_offsets, int32_t n_types,
                          const int32_t* type_ids, uint64_t seed, uint64_t global_step0, int32_t padded_size,
                          int32_t mode, int direct, int32_t* out, void* scratch, hipStream_t st, int negs = 0) {
  const PrepLayout L = prep_layout(B, negs);
  if (L.n_sub > 1)   // more keys than one workgroup's LDS holds: the sort runs across workgroups
    return prepare_big_launch(triples, T, first_row, B, s0, n, id_to_type, N, type_offsets, n_types, type_ids, seed,
                              global_step0, padded_size, mode, direct, out, scratch, st, negs);
  const SortBits sb = sort_bits_for(N);
  const int n_pass = sb.n_pass, bits = sb.bits;
  const size_t lds = prep_lds_bytes(L);
  // per device, idempotent and cheap: no cached flag, so no state and nothing to get stale on a second device
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_prepare_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(train_prepare_kernel, dim3((unsigned)n, (unsigned)L.n_sub), dim3(kPrepThreads), lds, st, triples,
                     T, first_row, B, s0, id_to_type, N, type_offsets, n_types, type_ids, seed, global_step0,
                     padded_size, mode, direct, n_pass, bits, negs, out);
  return launch_status();
}

#define GE_HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------ the pipeline handle (host object)
// Identity of the sequence of prepared steps + which chunk each of the two workspace buffers holds.
struct Pipeline {
  int device = -1;
  hipStream_t side = nullptr;
  hipEvent_t prep_done[2] = {nullptr, nullptr};
  hipEvent_t buf_free[2] = {nullptr, nullptr};
  bool live = false;
  const int32_t* triples = nullptr; const int32_t* id_to_type = nullptr; const int64_t* type_offsets = nullptr;
  const int32_t* type_ids = nullptr; void* workspace = nullptr;
  int64_t T = 0, B = 0, N = 0, origin_row = 0, next_abs = 0, resident[2] = {-1, -1};
  uint64_t seed = 0, origin_gs = 0;
  int32_t n_types = 0, padded_size = 0, mode = 0, d = 0;
  int direct = 0, negs = 0;
};

int pipeline_create(void** out) {
  Pipeline* p = new Pipeline();
  hipError_t e = hipGetDevice(&p->device);
  // (default priority: the highest one was measured in round 4 and changed nothing -- 126.6 vs 124.7 us per step at
  // 65,536 pairs, alternated on one box -- and neither did the lowest: 119.2 / 117.1 vs 119.4 / 119.8; un-profiled, the host is 8x ahead of the device and the chunk boundary costs no
  // more than an ordinary kernel boundary)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipEventCreateWithFlags(&p->prep_done[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&p->buf_free[i], hipEventDisableTiming);
  }
  if (e != hipSuccess) { delete p; return (int)e; }
  *out = p;
  return 0;
}
int pipeline_reset(void* h) {
  if (!h) return GE_EINVAL;
  static_cast<Pipeline*>(h)->live = false;
  return 0;
}
int pipeline_destroy(void* h) {
  if (!h) return GE_EINVAL;
  Pipeline* p = static_cast<Pipeline*>(h);
  if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); }
  for (int i = 0; i < 2; ++i) {
    if (p->prep_done[i]) (void)hipEventDestroy(p->prep_done[i]);
    if (p->buf_free[i]) (void)hipEventDestroy(p->buf_free[i]);
  }
  delete p;
  return 0;
}

static inline int64_t norm_row(int64_t first_row, int64_t T, int64_t B) {
  int64_t f = first_row % T;
  if (f + B > T) f = 0;
  return f;
}

// ------------------------------------------------------------------ walking the prepared records
// Shared by the hinge and the log-loss loops: which chunk of steps each of the two workspace buffers holds,
// when a prepare launch is due (on the handle's side stream, one chunk ahead, or on the caller's stream
// without a handle) and where step s finds its record.
struct StepIdentity {
  const int32_t* triples; int64_t T, first_row, B; const int32_t* id_to_type; int64_t N;
  const int64_t* type_offsets; int32_t n_types; const int32_t* type_ids; uint64_t seed, global_step0;
  int32_t padded_size, mode, d; int direct, negs; void* workspace;
};

class PrepCursor {
 public:
  PrepCursor(const StepIdentity& id, Pipeline* handle, int32_t* prep_base, hipStream_t st)
      : id_(id), L_(prep_layout(id.B, id.negs)), K_(prep_chunk_steps(id.B, id.negs)),
        buf_ints_((int64_t)(prep_chunk_bytes(id.B, id.negs) / sizeof(int32_t))), base_(prep_base), st_(st),
        pipe_(handle ? handle : &local_), own_side_(handle != nullptr) {}

  int begin() {
    if (own_side_) {
      int dev = -1;
      GE_HIP_TRY(hipGetDevice(&dev));
      if (dev != pipe_->device) return GE_EINVAL;
      Pipeline* p = pipe_;
      const bool cont = p->live && p->triples == id_.triples && p->id_to_type == id_.id_to_type &&
                        p->type_offsets == id_.type_offsets && p->type_ids == id_.type_ids &&
                        p->workspace == id_.workspace && p->T == id_.T && p->B == id_.B && p->N == id_.N &&
                        p->seed == id_.seed && p->n_types == id_.n_types && p->padded_size == id_.padded_size &&
                        p->mode == id_.mode && p->d == id_.d && p->direct == id_.direct && p->negs == id_.negs &&
                        id_.global_step0 == p->origin_gs + (uint64_t)p->next_abs &&
                        norm_row(id_.first_row, id_.T, id_.B) == step_row(p->origin_row, id_.T, id_.B, p->next_abs);
      if (!cont) {
        p->live = true;
        p->triples = id_.triples; p->id_to_type = id_.id_to_type; p->type_offsets = id_.type_offsets;
        p->type_ids = id_.type_ids; p->workspace = id_.workspace; p->T = id_.T; p->B = id_.B; p->N = id_.N;
        p->seed = id_.seed; p->n_types = id_.n_types; p->padded_size = id_.padded_size; p->mode = id_.mode;
        p->d = id_.d; p->direct = id_.direct; p->negs = id_.negs;
        p->origin_row = norm_row(id_.first_row, id_.T, id_.B); p->origin_gs = id_.global_step0; p->next_abs = 0;
        p->resident[0] = p->resident[1] = -1;
      }
      base_abs_ = p->next_abs;
    } else {
      pipe_->origin_row = norm_row(id_.first_row, id_.T, id_.B);
      pipe_->origin_gs = id_.global_step0;
      pipe_->side = st_;
    }
    return 0;
  }

  // record of step s of this call (s ascending); launches / waits for prepare work as chunks are entered
  int step(int64_t s, const int32_t** rec) {
    const int64_t abs_s = base_abs_ + s, chunk = abs_s / K_, in_chunk = abs_s % K_;
    if (s == 0 || in_chunk == 0) {
      int rc = ensure(chunk);
      if (rc) return rc;
      // the next chunk is prepared on the side stream while this one trains -- also when it lies beyond
      // this call's last step: the call that continues from there finds it done
      if (own_side_) {
        rc = ensure(chunk + 1);
        if (rc) return rc;
        GE_HIP_TRY(hipStreamWaitEvent(st_, pipe_->prep_done[chunk & 1], 0));
      }
    }
    *rec = base_ + (chunk & 1) * buf_ints_ + in_chunk * L_.stride;
    return 0;
  }

  // after the last step: the last record (for the caller's neg_ws) and the end-of-call ordering
  int finish(int64_t n_steps, const int32_t** last_rec) {
    const int64_t la = base_abs_ + n_steps - 1;
    *last_rec = base_ + ((la / K_) & 1) * buf_ints_ + (la % K_) * L_.stride;
    if (own_side_) {
      // the look-ahead launch writes into the caller's workspace after this call returns: order it before
      // whatever the caller enqueues on `st` next (e.g. a stream-ordered free of the workspace)
      GE_HIP_TRY(hipStreamWaitEvent(st_, pipe_->prep_done[((la / K_) + 1) & 1], 0));
      pipe_->next_abs = base_abs_ + n_steps;
    }
    return 0;
  }

  const PrepLayout& layout() const { return L_; }

 private:
  // chunk c of the sequence -> buffer c & 1.  Before a prepare launch overwrites a buffer, the side
  // stream waits for everything enqueued on `st` so far: that includes every step of the chunk that
  // held the buffer before, and (first launch) earlier work that may still be writing `triples`.
  int ensure(int64_t c) {
    const int b = (int)(c & 1);
    if (pipe_->resident[b] == c) return 0;
    if (own_side_) {
      GE_HIP_TRY(hipEventRecord(pipe_->buf_free[b], st_));
      GE_HIP_TRY(hipStreamWaitEvent(pipe_->side, pipe_->buf_free[b], 0));
    }
    int rc = prepare_launch(id_.triples, id_.T, pipe_->origin_row, id_.B, c * K_, K_, id_.id_to_type, id_.N,
                            id_.type_offsets, id_.n_types, id_.type_ids, id_.seed, pipe_->origin_gs, id_.padded_size,
                            id_.mode, id_.direct, base_ + b * buf_ints_, base_ + 2 * buf_ints_, pipe_->side, id_.negs);
    if (rc) return rc;
    if (own_side_) GE_HIP_TRY(hipEventRecord(pipe_->prep_done[b], pipe_->side));
    pipe_->resident[b] = c;
    return 0;
  }

  StepIdentity id_;
  PrepLayout L_;
  int64_t K_, buf_ints_;
  int32_t* base_;
  hipStream_t st_;
  Pipeline local_;          // no handle: prepare on the caller's stream, nothing survives the call
  Pipeline* pipe_;
  bool own_side_;
  int64_t base_abs_ = 0;
};

int train_steps_run(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                    int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                    int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                    int32_t padded_size, int32_t mode, float margin, float lr0, float decay_steps,
                    float decay_rate, float max_norm, int model, float* loss, int keep_all_losses,
                    int32_t* neg_ws, void* workspace, size_t workspace_bytes, void** ev_pairs, int ev_kernel,
                    void* pipe_handle, hipStream_t st) {
  if (n_steps <= 0) return 0;
  const int deterministic = (model & GE_STEP_DETERMINISTIC) ? 1 : 0;   // rows with > 16 slots reduced in a fixed order (prepared path)
  model &= ~GE_STEP_DETERMINISTIC;
  // HolE (model 1, the caller's real-valued table) is carried in the frequency domain for the duration
  // of the call when d is even: the DFT is linear and norm-preserving, so the clip and SGD commute with
  // it, and README.md:42's r . ifft(conj(fft h) fft t) is the ComplEx-shaped trilinear form on the half
  // spectrum (ge_complex_dev.h, SPEC).  Model 2: the table already IS spectral (ge_hole_to_spectral).
  // Model 3 (and odd d): the direct-correlation kernels of ge_hole.hip on the real table.
  const bool transform = model == GE_MODEL_HOLE && !(d & 1);
  const bool spectral = transform || model == GE_MODEL_HOLE_SPECTRAL;
  const bool hole_direct = (model == GE_MODEL_HOLE && (d & 1)) || model == GE_MODEL_HOLE_DIRECT;
  if (transform) { int rc = hole_spectral_launch(table, N, d, /*inverse=*/0, st); if (rc) return rc; }
  int32_t* gidx = reinterpret_cast<int32_t*>(workspace);
  float* gval0 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256));
  const bool fast = train_fast_ok(B, d) && workspace_bytes >= train_ws_bytes(B, d);
  const int ring = fast ? grad_ring(B, d) : 1;
  const size_t region_floats = grad_region_bytes(B, d) / sizeof(float);
  const int direct = hole_direct ? 0 : 1;   // sole-slot rows updated by the producing pair
  const StepIdentity ident{triples, T, first_row, B, id_to_type, N, type_offsets, n_types, type_ids, seed, global_step0,
                           padded_size, mode, d, direct, 0, workspace};
  PrepCursor cur(ident, fast ? static_cast<Pipeline*>(pipe_handle) : nullptr,
                 reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + train_grad_bytes(B, d)), st);
  const PrepLayout& L = cur.layout();
  if (fast) { int rc = cur.begin(); if (rc) return rc; }
  auto lr_at = [&](uint64_t gs) {
    return decay_steps > 0.f ? lr0 / (1.0f + decay_rate * ((float)gs / decay_steps)) : lr0;
  };
  for (int64_t s = 0; s < n_steps; ++s) {
    const uint64_t gs = global_step0 + (uint64_t)s;
    const float lr = lr_at(gs);
    const int32_t* pos = triples + 3 * step_row(first_row, T, B, s);
    float* loss_s = keep_all_losses ? loss + s * B : loss;
    hipEvent_t e0 = ev_pairs ? (hipEvent_t)ev_pairs[2 * s] : nullptr;
    hipEvent_t e1 = ev_pairs ? (hipEvent_t)ev_pairs[2 * s + 1] : nullptr;
    int rc;
    const int32_t* neg;
    const int32_t* step_rec = nullptr;
    if (e0 && ev_kernel == 0) (void)hipEventRecord(e0, st);
    if (fast) {
      rc = cur.step(s, &step_rec);
      if (rc) return rc;
      neg = step_rec;
    } else {
      rc = corrupt_batch_launch(pos, B, id_to_type, N, type_offsets, n_types, type_ids, seed, gs, padded_size,
                                mode, neg_ws, st);
      if (rc) return rc;
      neg = neg_ws;
    }
    if (e1 && ev_kernel == 0) (void)hipEventRecord(e1, st);
    float* gval = gval0 + (size_t)(gs % (uint64_t)ring) * region_floats;   // this step's region of the ring
    // the timing events ride on the dispatch packet itself (hipExtLaunchKernel start/stop events):
    // they report the kernel's own begin/end timestamps, like rocprofv3's kernel trace
    hipEvent_t g0 = ev_kernel == 1 ? e0 : nullptr, g1 = ev_kernel == 1 ? e1 : nullptr;
    hipEvent_t a0 = ev_kernel == 2 ? e0 : nullptr, a1 = ev_kernel == 2 ? e1 : nullptr;
    const bool dir = fast && direct;
    rc = hole_direct ? hole_hinge_grad_launch(table, N, d, pos, neg, B, margin, lr, max_norm, loss_s, gidx, gval, st, g0, g1)
                     : complex_hinge_grad_launch(table, N, d, pos, neg, B, margin, lr, max_norm, loss_s, gidx, gval, st, g0, g1,
                                                 dir ? step_rec + L.off_slot : nullptr, dir ? table : nullptr, spectral ? 1 : 0,
                                                 (fast && L.off_order >= 0) ? step_rec + L.off_order : nullptr);
    if (rc) return rc;
    if (fast) rc = apply_sorted_launch(table, d, L, step_rec, gidx, gval, st, a0, a1, deterministic);
    else rc = scatter_add_rows_launch(table, N, d, gidx, gval, 6 * B, st, a0, a1);
    if (rc) return rc;
  }
  if (fast) {
    // keep the caller's neg_ws meaningful: the last step's negatives
    const int32_t* last = nullptr;
    int rc = cur.finish(n_steps, &last);
    if (rc) return rc;
    GE_HIP_TRY(hipMemcpyAsync(neg_ws, last, sizeof(int32_t) * 3 * (size_t)B, hipMemcpyDeviceToDevice, st));
  }
  if (transform) { int rc = hole_spectral_launch(table, N, d, /*inverse=*/1, st); if (rc) return rc; }
  return 0;
}

// ------------------------------------------------------------------ the --log_loss loop (holE.py:194-196, 206-220, 296)
// Per step: M = (1+K) B triples -- the positives (label +1) and K corrupted batches (label -1, the k-th drawn
// with Philox step key global_step * K + k) -- loss_i = log(1 + exp(-y_i s_i)) + l2 * l2_loss(table), and
// minimize() of the SUM:   table <- table * (1 - lr M l2) - lr * sparse gradient.
// The dense factor f_s = 1 - lr_s M l2 touches every row every step (two passes over a 960 MB table in the
// reference's formulation).  Here the table is held as  actual = g * stored  with ONE scalar g: the dense decay
// is g <- g f_s on the host, the kernels read stored rows times g, and the sparse update, which is in actual
// units, is added to the stored rows divided by the NEW scale.  The table is materialised (stored *= g, one
// pass) at the end of the call, or earlier if |g| leaves [2^-40, 2^40] or a factor is exactly 0.
// l2_loss(table) enters only the reported loss values: it is summed (one read pass) for the steps whose loss
// vector the caller keeps (all of them with keep_all_losses, else the last).
size_t train_logloss_ws_bytes(int64_t B, int32_t negs, int32_t d) {
  const size_t M = (size_t)(1 + negs) * (size_t)B;
  const size_t region = align_up_sz(sizeof(float) * 3 * M * (size_t)d, 256);
  size_t ring = ((size_t)64 << 20) / region + 1;
  if (ring < 2) ring = 2;
  if (ring > 8) ring = 8;
  return 256 + align_up_sz(sizeof(int32_t) * 3 * M, 256) + ring * region + 2 * prep_chunk_bytes(B, negs) +
         prep_big_scratch_bytes(B, negs, prep_chunk_steps(B, negs));
}

int complex_logloss_grad_launch(const float*, int64_t, int32_t, const int32_t*, const float*, int64_t, float, float, float, const float*, float*, int32_t*, float*, hipStream_t, const int32_t*, int64_t, float, float, hipEvent_t, hipEvent_t);
int table_sumsq_launch(const float*, int64_t, float*, hipStream_t);
int table_scale_launch(float*, int64_t, float, hipStream_t);

int train_logloss_run(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                      int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                      int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                      int32_t padded_size, int32_t mode, int32_t negs, float l2, float lr0, float decay_steps,
                      float decay_rate, float max_norm, float* loss, int keep_all_losses, int32_t* neg_ws,
                      void* workspace, size_t workspace_bytes, void* pipe_handle, hipStream_t st) {
  if (n_steps <= 0) return 0;
  if (workspace_bytes < train_logloss_ws_bytes(B, negs, d)) return GE_ENOMEM;
  const size_t M = (size_t)(1 + negs) * (size_t)B;
  const size_t region = align_up_sz(sizeof(float) * 3 * M * (size_t)d, 256);
  size_t ring = ((size_t)64 << 20) / region + 1;
  if (ring < 2) ring = 2;
  if (ring > 8) ring = 8;
  char* w = reinterpret_cast<char*>(workspace);
  float* sumsq = reinterpret_cast<float*>(w);
  int32_t* gidx = reinterpret_cast<int32_t*>(w + 256);
  char* gval0 = w + 256 + align_up_sz(sizeof(int32_t) * 3 * M, 256);
  int32_t* prep_base = reinterpret_cast<int32_t*>(gval0 + ring * region);
  const StepIdentity ident{triples, T, first_row, B, id_to_type, N, type_offsets, n_types, type_ids, seed, global_step0,
                           padded_size, mode, d, 0, negs, workspace};
  PrepCursor cur(ident, static_cast<Pipeline*>(pipe_handle), prep_base, st);
  const PrepLayout& L = cur.layout();
  int rc = cur.begin();
  if (rc) return rc;
  double g = 1.0;   // actual table = g * stored table
  auto materialise = [&]() -> int {
    if (g == 1.0) return 0;
    int r2 = table_scale_launch(table, N * (int64_t)d, (float)g, st);
    g = 1.0;
    return r2;
  };
  for (int64_t s = 0; s < n_steps; ++s) {
    const uint64_t gs = global_step0 + (uint64_t)s;
    const float lr = decay_steps > 0.f ? lr0 / (1.0f + decay_rate * ((float)gs / decay_steps)) : lr0;
    const int32_t* pos = triples + 3 * step_row(first_row, T, B, s);
    const int32_t* step_rec = nullptr;
    rc = cur.step(s, &step_rec);
    if (rc) return rc;
    const double f = 1.0 - (double)lr * (double)M * (double)l2;      // this step's dense factor
    double g_new = g * f;
    if (g_new == 0.0 || std::fabs(g_new) < 9.1e-13 || std::fabs(g_new) > 1.1e12) {
      // the scalar would leave the comfortable range (or the factor is exactly 0: the reference's own defaults
      // give f = 1 - 0.1 * 1024 * 0.1 < 0): apply the pending scale now and take this step's factor densely too
      rc = materialise();
      if (rc) return rc;
      g_new = 1.0;
    }
    const bool want_loss = keep_all_losses || s == n_steps - 1;
    if (want_loss) { rc = table_sumsq_launch(table, N * (int64_t)d, sumsq, st); if (rc) return rc; }
    float* loss_s = keep_all_losses ? loss + s * (int64_t)M : loss;
    float* gval = reinterpret_cast<float*>(gval0 + (size_t)(gs % (uint64_t)ring) * region);
    const bool dense_now = (g_new == 1.0 && g * f != 1.0);            // this step's factor is applied by a dense pass
    // sparse part in actual units is -lr * grad; stored rows take it divided by the scale they will carry
    const float neg_lr_eff = (float)(-(double)lr / (dense_now ? 1.0 : g_new));
    rc = complex_logloss_grad_launch(table, N, d, pos, nullptr, (int64_t)M, lr, max_norm, l2, want_loss ? sumsq : nullptr,
                                     loss_s, gidx, gval, st, step_rec, B, (float)g, neg_lr_eff, nullptr, nullptr);
    if (rc) return rc;
    if (dense_now) {   // g == 1 here (materialised above): stored = actual; new = actual * f + sparse
      rc = table_scale_launch(table, N * (int64_t)d, (float)f, st);
      if (rc) return rc;
    }
    rc = apply_sorted_launch(table, d, L, step_rec, gidx, gval, st, nullptr, nullptr);
    if (rc) return rc;
    g = g_new;
  }
  rc = materialise();
  if (rc) return rc;
  const int32_t* last = nullptr;
  rc = cur.finish(n_steps, &last);
  if (rc) return rc;
  GE_HIP_TRY(hipMemcpyAsync(neg_ws, last, sizeof(int32_t) * 3 * (size_t)negs * (size_t)B, hipMemcpyDeviceToDevice, st));
  return 0;
}

// prepared records of `n_steps` consecutive steps into a caller buffer (tests, tools): the same launch
// ge_train_steps uses.  For B > 4096 the buffer also holds the multi-tile sort's scratch, behind the records.
size_t train_prepare_bytes(int64_t B, int64_t n_steps) {
  return align_up_sz(sizeof(int32_t) * (size_t)n_steps * (size_t)prep_layout(B).stride, 256) + prep_big_scratch_bytes(B, 0, n_steps);
}
int train_prepare_run(const int32_t* triples, int64_t T, int64_t first_row, int64_t B, int64_t n_steps,
                      const int32_t* id_to_type, int64_t N, const int64_t* type_offsets, int32_t n_types,
                      const int32_t* type_ids, uint64_t seed, uint64_t global_step0, int32_t padded_size,
                      int32_t mode, int direct, int32_t* out, hipStream_t st) {
  if (n_steps == 0) return 0;
  void* scratch = reinterpret_cast<char*>(out) + align_up_sz(sizeof(int32_t) * (size_t)n_steps * (size_t)prep_layout(B).stride, 256);
  return prepare_launch(triples, T, norm_row(first_row, T, B), B, 0, n_steps, id_to_type, N, type_offsets, n_types,
                        type_ids, seed, global_step0, padded_size, mode, direct ? 1 : 0, out, scratch, st);
}

void train_prepared_layout(int64_t B, int64_t* out) {
  const PrepLayout L = prep_layout(B);
  out[0] = L.stride; out[1] = L.n_sub; out[2] = L.S; out[3] = L.off_slot; out[4] = L.off_sub;
  out[5] = L.sub_stride; out[6] = L.off_items; out[7] = L.off_islots;
}

}  // namespace ge
