// ge_prep.h -- the prepared-step record shared by the prepare kernels (ge_train.hip: one workgroup sorts a
// whole step in LDS; ge_prep_big.hip: steps of more than 4096 units, sorted across workgroups) and by the
// kernels that consume it (apply_sorted_kernel, the grad kernels' "direct" tags).
#pragma once
#include "ge_common.h"

namespace ge {

constexpr int kSlotDirect = -2;  // slot_item code: sole contributor of its row, applied by the producer
constexpr int kPrepThreads = 1024;
constexpr int kPrepWaves = kPrepThreads / kWave;
constexpr int kItemCap = 16;       // C: max gradient rows summed by one wavefront
constexpr int64_t kSub = 4096;     // units per sort tile: 4 x 4096 keys of 8 B = 128 KiB of the CU's 160 KiB
constexpr int kMaxRadix = 256;
constexpr unsigned long long kInvalidKey = ~0ull;

__host__ __device__ inline int64_t step_row(int64_t first_row, int64_t T, int64_t B, int64_t s) {
  // batch s starts at first_row + s*B, wrapping to row 0 whenever a batch would run past T
  // (no short batches, holE.py:283)
  int64_t first = first_row % T;
  if (first + B > T) first = 0;
  const int64_t n0 = (T - first) / B;  // steps before the first wrap
  if (s < n0) return first + s * B;
  const int64_t per = T / B;
  return ((s - n0) % per) * B;
}

// ------------------------------------------------------------------ prepared-step record (int32 words)
// hinge (negs = 0):   neg[3B] | slot_item[6B] | (n_sub > 1: order[B]) | pad to 64 | n_sub x { n_items, pad to 64 | items[P][2] | islots[P][16] }
//   a UNIT is a (pos,neg) pair with 4 sort keys; slot = 6*pair + {0 h+,1 t+,2 r+,3 h-,4 t-,5 r-}.
// log-loss (negs = K): neg[K][B][3] | pad to 64 | n_sub x { ... }
//   a UNIT is one of the M = (1+K)B triples (positives first, then the K corrupted batches, holE.py:206-220)
//   with 3 sort keys; slot = 3*triple + {0 h, 1 t, 2 r}.
// S = units per tile (<= 4096), P = keys per tile (a multiple of 1024).  items[k] = {table row,
// count | multi << 30}; islots[k][0..16) = the slots the item sums, -1 padded.
// The step's keys are sorted by (row, slot) as ONE sequence of n_sub * P positions (invalid keys last);
// sub-record t holds the items that START in positions [t*P, (t+1)*P) -- an item may run up to 15
// positions into the next tile.  With one tile the whole sort happens in LDS (train_prepare_kernel).
struct PrepLayout {
  int64_t B, negs, units, n_sub, S, P, off_slot, off_order, off_sub, sub_stride, off_items, off_islots, stride;
  int epu;   // sort keys per unit
};
__host__ __device__ inline PrepLayout prep_layout(int64_t B, int64_t negs = 0) {
  PrepLayout L;
  L.B = B;
  L.negs = negs;
  L.epu = negs > 0 ? 3 : 4;
  L.units = negs > 0 ? (1 + negs) * B : B;
  L.n_sub = (L.units + kSub - 1) / kSub;
  const int64_t s = L.units < kSub ? L.units : kSub;
  const int64_t gran = negs > 0 ? 1024 : 256;          // P = epu * S must be a multiple of 1024
  L.S = (s + gran - 1) / gran * gran;
  L.P = L.epu * L.S;
  L.off_slot = 3 * B;
  // hinge steps of more than one tile also carry order[B]: the pairs sorted by relation row (stable), the order
  // the gradient kernel walks them in so that a wave meets runs of one relation (see complex_hinge_grad_kernel)
  L.off_order = (negs == 0 && L.n_sub > 1) ? 9 * B : -1;
  L.off_sub = negs > 0 ? (3 * negs * B + 63) / 64 * 64 : ((L.off_order >= 0 ? 10 : 9) * B + 63) / 64 * 64;
  L.off_items = 64;
  L.off_islots = 64 + 2 * L.P;
  L.sub_stride = 64 + 2 * L.P + kItemCap * L.P;
  L.stride = L.off_sub + L.n_sub * L.sub_stride;
  return L;
}

// what the multi-tile kernels (ge_prep_big.hip) need to know about a record: the native loops derive it from a
// PrepLayout, the row-sharded step's owner side (ge_shard.hip) from the longest request list of the chunk
struct TileGeom {
  int P, n_sub;                 // keys per tile, tiles per step
  int64_t stride;               // int32 words per step record
  int64_t off_slot;             // slot_item inside a record (-1: the record has none)
  int64_t off_sub, sub_stride;  // tile 0, words per tile
  int off_items, off_islots;
};
inline TileGeom geom_of(const PrepLayout& L) {
  return TileGeom{(int)L.P, (int)L.n_sub, L.stride, L.off_slot, L.off_sub, L.sub_stride, (int)L.off_items, (int)L.off_islots};
}
// a record that is nothing but tiles { n_items, pad to 64 | items[P][2] | islots[P][16] }
inline TileGeom geom_plain(int P, int n_sub) {
  const int64_t sub = 64 + 2 * (int64_t)P + kItemCap * (int64_t)P;
  return TileGeom{P, n_sub, n_sub * sub, -1, 0, sub, 64, 64 + 2 * P};
}

// row-sharded step (ge_shard.hip): what the items kernel adds per step when rows >= R are REMOTE rows (sorted by
// owner, then row): their index u in the step's staging order, where every slot reads its row from, the request list
struct ShardOut {
  int32_t R;                    // rows < R: this rank's shard rows; R + u: staged row u of the step
  int64_t B;
  int32_t* pos_src;             // [S][3B]  per pair and column: shard row, or R + u; -1 = invalid pair
  int32_t* neg_src;             // [S][B]   (source << 1 | column) of the corrupted entity, -1 = none of its own
  int32_t* req_row;             // [S][n_sub * P]  row u's index in its owner's shard
  const int32_t* tile_heads;    // [S][n_sub] distinct remote rows that START in each tile
  int32_t peer;                 // 1: pos_src / neg_src name another owner's row by its VIRTUAL row R (1 + owner) + local row
                                //    (peer-mapped shards are read directly), not by its staging index R + u
};

// radix-sort geometry for row ids < N: n_pass passes of `bits` bits (>= 6: one counter per thread in the scan)
struct SortBits { int n_pass, bits; };
inline SortBits sort_bits_for(int64_t N) {
  int nbits = 1;
  while (((int64_t)1 << nbits) <= N) ++nbits;          // the value N itself (and above) is free for invalid keys
  SortBits s;
  s.n_pass = (nbits + 7) / 8;
  s.bits = (nbits + s.n_pass - 1) / s.n_pass;
  if (s.bits < 6) s.bits = 6;
  return s;
}

__device__ __forceinline__ int wave_incl_add(int v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_up(v, o, kWave); if (lane >= o) v += t; }
  return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) { const int t = __shfl_up(v, o, kWave); if (lane >= o) v = max(v, t); }
  return v;
}

// the sort keys of unit `j` of a step (phase 1 of both prepare kernels): draws the negatives, writes them to
// the record, returns up to 4 keys (row << 32 | slot), kInvalidKey where a unit has no such slot.
struct StepSource {
  const int32_t* pos; int64_t B; const int32_t* id_to_type; int64_t N; const int64_t* type_offsets;
  int32_t n_types; const int32_t* type_ids; uint64_t seed, step; int32_t padded_size, mode, negs;
};

__device__ __forceinline__ void unit_keys(const StepSource& S, int64_t j, bool batch_heads, int32_t* __restrict__ neg,
                                          unsigned long long (&k)[4]) {
  if (S.negs > 0) {
    // log-loss: unit j < B is positive j; unit B + k*B + i is positive i corrupted by the k-th corrupt_batch
    // call of the step, whose Philox step key is global_step * K + k (its own coin, its own subsample)
    const int64_t B = S.B;
    const int64_t i = j < B ? j : (j - B) % B;
    int32_t t3[3] = {S.pos[3 * i], S.pos[3 * i + 1], S.pos[3 * i + 2]};
    if (j >= B) {
      const int64_t kk = (j - B) / B;
      const uint64_t stepk = S.step * (uint64_t)S.negs + (uint64_t)kk;
      const bool heads_k = (S.mode == GE_CORRUPT_BATCH_COIN) ? batch_coin_heads(S.seed, stepk) : false;
      int col;
      const int32_t repl = corrupt_one(t3, i, heads_k, S.id_to_type, S.N, S.type_offsets, S.n_types, S.type_ids, S.seed,
                                       stepk, S.padded_size, S.mode, col);
      t3[col] = repl;
      int32_t* o = neg + 3 * (kk * B + i);
      o[0] = t3[0]; o[1] = t3[1]; o[2] = t3[2];
    }
    const bool bad = t3[0] < 0 || t3[1] < 0 || t3[2] < 0 || t3[0] >= S.N || t3[1] >= S.N || t3[2] >= S.N;
#pragma unroll
    for (int X = 0; X < 3; ++X)
      k[X] = bad ? kInvalidKey : (((unsigned long long)(uint32_t)t3[X] << 32) | (uint32_t)(3 * j + X));
    k[3] = kInvalidKey;
    return;
  }
  const int64_t i = j;
  int32_t p[3] = {S.pos[3 * i], S.pos[3 * i + 1], S.pos[3 * i + 2]};
  int col;
  const int32_t repl = corrupt_one(p, i, batch_heads, S.id_to_type, S.N, S.type_offsets, S.n_types, S.type_ids, S.seed,
                                   S.step, S.padded_size, S.mode, col);
  int32_t n[3] = {p[0], p[1], p[2]};
  n[col] = repl;
  neg[3 * i] = n[0]; neg[3 * i + 1] = n[1]; neg[3 * i + 2] = n[2];
  const bool bad = p[0] < 0 || p[1] < 0 || p[2] < 0 || p[0] >= S.N || p[1] >= S.N || p[2] >= S.N || repl < 0 || repl >= S.N;
  // IndexedSlices slots of pair i (ge_hip.h): h+ 0, t+ 1, r+ 2, h- 3, t- 4, r- 5; a negative-side
  // slot exists only where the row differs from the positive one.
#pragma unroll
  for (int X = 0; X < 3; ++X)
    k[X] = bad ? kInvalidKey : (((unsigned long long)(uint32_t)p[X] << 32) | (uint32_t)(6 * i + X));
  k[3] = (bad || repl == p[col]) ? kInvalidKey : (((unsigned long long)(uint32_t)repl << 32) | (uint32_t)(6 * i + 3 + col));
}

}  // namespace ge
