// ge_train.hip -- the native inner training loop (holE.py:340-362 minus validation).
//
// Negatives depend only on (seed, step, positives) -- never on the table -- so everything about a
// step except its floating-point work can be prepared ahead, in bulk, for a chunk of steps:
//
//   train_prepare_kernel   one 1024-thread workgroup per step: draws the step's negatives
//                          (holE.py:97-140, 343-347), builds the (row, IndexedSlices slot) list of the
//                          <= 4B gradient rows the step will emit, sorts it by row with a bitonic
//                          sort held entirely in LDS (4B x 8 B = 128 KiB of the CU's 160 KiB at
//                          B=4096) and cuts it into work items of <= C entries of one row.
//   (per step) hinge_grad  fused gather -> clip -> score -> sigmoid -> hinge -> gradient rows
//   (per step) apply_sorted_kernel   one wavefront per work item sums its gradient rows and
//                          updates the table row ONCE with a plain read-modify-write.
//
// This replaces the float-atomic ScatterSub (memory-side atomics ~1.3 TB/s chip-wide and an order
// of magnitude slower when many waves hit one hot row -- Zipfian heads, a handful of relations)
// by coalesced row reads plus one write per distinct row, and makes the update order fixed:
// rows with <= C occurrences (the vast majority) are bitwise reproducible.  Only rows split
// over several work items combine their partial sums with atomics.
#include "ge_common.h"
#include <cstdlib>

namespace ge {

int complex_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr, const int32_t* slot_item = nullptr);
int hole_hinge_grad_launch(const float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float, float*, int32_t*, float*, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int scatter_add_rows_launch(float*, int64_t, int32_t, const int32_t*, const float*, int64_t, hipStream_t, hipEvent_t = nullptr, hipEvent_t = nullptr);
int corrupt_batch_launch(const int32_t*, int64_t, const int32_t*, int64_t, const int64_t*, int32_t, const int32_t*, uint64_t, uint64_t, int32_t, int32_t, int32_t*, hipStream_t);

struct FusedArgs {
  const int32_t* slot_item; const int32_t* items; const int32_t* occ;
  int32_t* item_cnt; int32_t* row_cnt; float* partials; int32_t* gidx; float* gval; int gstride; int debug;
};
bool fused_shape_ok(int d, const void* table, int& lpt, int& niter);
int fused_gstride(int d);
int complex_fused_step_launch(float*, int64_t, int32_t, const int32_t*, const int32_t*, int64_t, float, float, float,
                              float*, const FusedArgs&, hipStream_t, hipEvent_t, hipEvent_t);

static int g_fused_enabled = -1;  // -1: read GE_FUSED_STEP on first use
int set_fused_step(int on) {
  const int prev = g_fused_enabled;
  g_fused_enabled = on ? 1 : 0;
  return prev;
}
static bool fused_enabled() {
  if (g_fused_enabled < 0) {
    const char* e = getenv("GE_FUSED_STEP");
    g_fused_enabled = (e && e[0] == '1') ? 1 : 0;   // opt-in: measured slower than the two-launch step
  }
  return g_fused_enabled == 1;
}

constexpr int kSlotDirect = -2;  // slot_item code: sole contributor of its row, applied by the producer
constexpr int kPrepThreads = 1024;
constexpr int kItemCap = 16;       // C: max gradient rows summed by one wavefront
constexpr int kPrepChunk = 32;     // steps prepared per launch (two buffers of 32 x ~1.7 MB at B=4096)
constexpr int64_t kFastMaxB = 4096;  // 4B sort keys of 8 B must fit the CU's LDS

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

// per-step int32 layout of the prepared data:
//   neg[3B] occ[4B] items[4B*5] n_items[64] slot_item[6B] item_cnt[4B] row_cnt[4B] islots[4B*16]
__host__ __device__ inline int64_t prep_stride(int64_t B) { return 105 * B + 64; }
__host__ __device__ inline int64_t off_occ(int64_t B) { return 3 * B; }
__host__ __device__ inline int64_t off_items(int64_t B) { return 7 * B; }
__host__ __device__ inline int64_t off_nitems(int64_t B) { return 27 * B; }
__host__ __device__ inline int64_t off_slot_item(int64_t B) { return 27 * B + 64; }
__host__ __device__ inline int64_t off_item_cnt(int64_t B) { return 33 * B + 64; }
__host__ __device__ inline int64_t off_row_cnt(int64_t B) { return 37 * B + 64; }
__host__ __device__ inline int64_t off_islots(int64_t B) { return 41 * B + 64; }

__global__ __launch_bounds__(kPrepThreads) void train_prepare_kernel(
    const int32_t* __restrict__ triples, int64_t T, int64_t first_row, int64_t B, int64_t s0,
    const int32_t* __restrict__ id_to_type, int64_t N, const int64_t* __restrict__ type_offsets,
    int32_t n_types, const int32_t* __restrict__ type_ids, uint64_t seed, uint64_t global_step0,
    int32_t padded_size, int32_t mode, int P, int direct, int32_t* __restrict__ prep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  int* scan = reinterpret_cast<int*>(keys + P);  // kPrepThreads + 1 ints
  const int tid = threadIdx.x;
  const int64_t s = s0 + blockIdx.x;
  const int32_t* pos = triples + 3 * step_row(first_row, T, B, s);
  int32_t* neg = prep + (int64_t)blockIdx.x * prep_stride(B);
  int32_t* occ = neg + off_occ(B);
  int32_t* items = neg + off_items(B);
  int32_t* n_items = neg + off_nitems(B);
  int32_t* slot_item = neg + off_slot_item(B);
  int32_t* islots = neg + off_islots(B);          // per item: its <= 16 slot ids inline (-1 padded)
  int32_t* item_cnt = neg + off_item_cnt(B);   // followed by row_cnt: 8B counters zeroed here, used once
  for (int i = tid; i < 6 * B; i += kPrepThreads) slot_item[i] = -1;
  for (int i = tid; i < 8 * B; i += kPrepThreads) item_cnt[i] = 0;
  const uint64_t step = global_step0 + (uint64_t)s;
  const bool batch_heads = (mode == GE_CORRUPT_BATCH_COIN) ? batch_coin_heads(seed, step) : false;
  constexpr unsigned long long kInvalid = ~0ull;

  for (int i = tid; i < B; i += kPrepThreads) {
    int32_t p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
    int col;
    const int32_t repl = corrupt_one(p, i, batch_heads, id_to_type, N, type_offsets, n_types, type_ids,
                                     seed, step, padded_size, mode, col);
    int32_t n[3] = {p[0], p[1], p[2]};
    n[col] = repl;
    neg[3 * i] = n[0]; neg[3 * i + 1] = n[1]; neg[3 * i + 2] = n[2];
    const bool bad = p[0] < 0 || p[1] < 0 || p[2] < 0 || p[0] >= N || p[1] >= N || p[2] >= N ||
                     repl < 0 || repl >= N;
    // IndexedSlices slots of pair i (ge_hip.h): h+ 0, t+ 1, r+ 2, h- 3, t- 4, r- 5; a negative-side
    // slot exists only where the row differs from the positive one.
#pragma unroll
    for (int X = 0; X < 3; ++X)
      keys[4 * i + X] = bad ? kInvalid : (((unsigned long long)(uint32_t)p[X] << 32) | (uint32_t)(6 * i + X));
    keys[4 * i + 3] = (bad || repl == p[col]) ? kInvalid
                                              : (((unsigned long long)(uint32_t)repl << 32) | (uint32_t)(6 * i + 3 + col));
  }
  for (int i = 4 * (int)B + tid; i < P; i += kPrepThreads) keys[i] = kInvalid;
  __syncthreads();

  // bitonic sort, ascending by (row, slot)
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += kPrepThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const unsigned long long a = keys[i], b = keys[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[l] = a; }
      }
      __syncthreads();
    }
  }

  // work items: cut every run of equal rows into pieces of <= kItemCap entries, counted from the
  // start of the run (so a row with <= kItemCap occurrences is always exactly one item).
  const int per = P / kPrepThreads;  // P >= kPrepThreads
  const int base = tid * per;
  auto rowof = [&](int i) -> uint32_t { return (uint32_t)(keys[i] >> 32); };
  auto valid = [&](int i) -> bool { return keys[i] != kInvalid; };
  auto is_head = [&](int i) -> bool { return valid(i) && (i == 0 || rowof(i) != rowof(i - 1)); };
  // a row with exactly one gradient slot in the step has one reader and one writer -- the same pair:
  // with `direct` it is not queued as an item, its slot is tagged kSlotDirect and the producing pair
  // updates the table row itself (no gradient-row round trip, no counter)
  auto is_sole = [&](int i) -> bool {
    return direct && is_head(i) && !(i + 1 < P && valid(i + 1) && rowof(i + 1) == rowof(i));
  };
  // (a) run start of every position: block-wide max-scan of the last head position per thread chunk
  int last_head = -1;
  for (int i = base; i < base + per; ++i) if (is_head(i)) last_head = i;
  scan[tid] = last_head;
  __syncthreads();
  for (int off = 1; off < kPrepThreads; off <<= 1) {
    const int v = (tid >= off) ? scan[tid - off] : -1;
    __syncthreads();
    scan[tid] = max(scan[tid], v);
    __syncthreads();
  }
  const int carry = (tid == 0) ? -1 : scan[tid - 1];
  __syncthreads();
  // (b) count item starts in this chunk, exclusive-scan the counts
  int cnt = 0;
  {
    int rs = carry;
    for (int i = base; i < base + per; ++i) {
      if (is_head(i)) rs = i;
      if (valid(i) && ((i - rs) % kItemCap) == 0 && !is_sole(i)) ++cnt;
    }
  }
  scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < kPrepThreads; off <<= 1) {  // inclusive Hillis-Steele scan
    const int v = (tid >= off) ? scan[tid - off] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int idx = scan[tid] - cnt;  // exclusive prefix
  if (tid == kPrepThreads - 1) n_items[0] = scan[tid];
  // (c) emit items and the slot list
  int rs = carry;
  for (int i = base; i < base + per; ++i) {
    if (i < 4 * B) occ[i] = valid(i) ? (int32_t)(uint32_t)keys[i] : -1;
    if (!valid(i)) continue;
    if (is_head(i)) rs = i;
    if (((i - rs) % kItemCap) != 0) continue;
    if (is_sole(i)) { slot_item[(uint32_t)keys[i]] = kSlotDirect; continue; }
    const uint32_t r = rowof(i);
    int e = i + 1;
    while (e < P && (e - i) < kItemCap && valid(e) && rowof(e) == r) ++e;
    const bool more = (e < P && valid(e) && rowof(e) == r);
    const bool multi = (i != rs) || more;
    const int ordinal = (i - rs) / kItemCap;     // items of one row are consecutive
    items[5 * idx] = (int32_t)r;
    items[5 * idx + 1] = i;
    items[5 * idx + 2] = (e - i) | (multi ? (1 << 30) : 0);
    items[5 * idx + 3] = idx - ordinal;          // the row's first item
    if (!more) items[5 * (idx - ordinal) + 4] = ordinal + 1;  // the row's item count, kept at its first item
    for (int j = i; j < e; ++j) slot_item[(uint32_t)keys[j]] = idx;
    for (int j = 0; j < kItemCap; ++j) islots[idx * kItemCap + j] = (i + j < e) ? (int32_t)(uint32_t)keys[i + j] : -1;
    ++idx;
  }
}

// One wavefront per work item: sum the item's gradient rows (skipping slots whose pair was
// hinge-inactive: grad_idx < 0), then table[row] += sum -- plain RMW when the row has a single
// item, atomics when it was split.  Lane l owns columns l, l+64, ... (256 contiguous bytes per
// wave instruction for loads, stores and atomics alike).
template <int NJ>
__global__ __launch_bounds__(kBlock) void apply_sorted_kernel(
    float* __restrict__ table, int d, const int32_t* __restrict__ items,
    const int32_t* __restrict__ n_items_ptr, const int32_t* __restrict__ islots,
    const int32_t* __restrict__ grad_idx, const float* __restrict__ grad_val) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int n_items = n_items_ptr[0];
  for (int w = wave; w < n_items; w += nwaves) {
    // two independent loads first: the item header and its inline slot list (lane o < 16 -> slot o)
    const int row = items[5 * w], cm = items[5 * w + 2];
    int slot_v = (lane < kItemCap) ? islots[w * kItemCap + lane] : -1;
    const int cnt = cm & 0x3FFFFFFF;
    const bool multi = (cm >> 30) & 1;
    float* dst = table + (int64_t)row * d;
    // the table row is fetched now, under the gradient-row loads, not after them
    float base[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + kWave * j;
      base[j] = (!multi && c < d) ? dst[c] : 0.f;
    }
    const bool act_v = slot_v >= 0 && grad_idx[slot_v] >= 0;   // pair was hinge-active
    if (slot_v < 0) slot_v = 0;
    const unsigned long long live = __ballot(act_v);
    if (live == 0ull) continue;  // wave-uniform
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
          v[q][j] = (c < d) ? src[c] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] += on[q] ? v[q][j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + kWave * j;
      if (c < d) {
        if (multi) atomic_add_f32(dst + c, acc[j]);
        else dst[c] = base[j] + acc[j];
      }
    }
  }
}

static int apply_sorted_launch(float* table, int d, int64_t B, const int32_t* items, const int32_t* n_items,
                               const int32_t* islots, const int32_t* gidx, const float* gval, hipStream_t st,
                               hipEvent_t ev_start, hipEvent_t ev_stop) {
  const int grid = grid_for(4 * B, kBlock / kWave);  // at most 4B items
  const int nj = (d + kWave - 1) / kWave;
#define LA(NJ) hipExtLaunchKernelGGL(apply_sorted_kernel<NJ>, dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, d, items, n_items, islots, gidx, gval)
  if (nj <= 1) LA(1); else if (nj <= 2) LA(2); else if (nj <= 4) LA(4); else if (nj <= 8) LA(8); else if (nj <= 16) LA(16);
  else return GE_ENOTSUP;
#undef LA
  return launch_status();
}

static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t hinge_ws_bytes(int64_t B, int32_t d) {
  return align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256) + align_up_sz(sizeof(float) * 6 * (size_t)B * (size_t)d, 256);
}

bool train_fast_ok(int64_t B, int32_t d) { return B >= 1 && B <= kFastMaxB && d <= 1024; }

// training workspace: [gidx 6B][gval RING x (6B x gs)][partials 4B x gs][prep buffer 0][prep buffer 1],
// gs = gradient-row stride padded to whole 128-byte lines (the fused kernel's publish unit).
// The gradient rows go to a RING of regions, one per step in turn: a region is written by the grad kernel
// on one XCD and read by the apply kernel on another, and rewriting lines that still sit in another XCD's
// L2 costs ~3.7 us per 13 MB (tools/xcd_locality_probe.hip: 9.1 vs 5.4 us); by the time a region comes
// round again (normally 4 steps later, > the 32 MB of L2 in between) its lines have been evicted and the
// stores take the fast path.
static size_t grad_region_bytes(int64_t B, int32_t d) {
  return align_up_sz(sizeof(float) * 6 * (size_t)B * (size_t)fused_gstride(d), 256);
}
static int grad_ring(int64_t B, int32_t d) {
  const size_t reg = grad_region_bytes(B, d);
  if (const char* e = getenv("GE_GRAD_RING")) { const int v = atoi(e); if (v >= 1 && v <= 64) return v; }   // tuning
  // measured at B=4096, d=200 (22 MB regions): 1 -> 22.4, 2 -> 21.3, 4 -> 21.0, 8 -> 21.0, 16 -> 21.8, 32 -> 22.8 us/step:
  // long enough to outlive the L2s, short enough to stay inside the 256 MB Infinity Cache
  size_t r = ((size_t)64 << 20) / reg + 1;
  if (r < 2) r = 2;
  if (r > 16) r = 16;
  if (r < 4 && 4 * reg <= ((size_t)160 << 20)) r = 4;
  return (int)r;
}
static size_t train_grad_bytes(int64_t B, int32_t d) {
  const size_t gs = (size_t)fused_gstride(d);
  return align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256) + (size_t)grad_ring(B, d) * grad_region_bytes(B, d) +
         align_up_sz(sizeof(float) * 4 * (size_t)B * gs, 256);
}
size_t train_ws_bytes(int64_t B, int32_t d) {
  if (!train_fast_ok(B, d)) return hinge_ws_bytes(B, d);
  return train_grad_bytes(B, d) + 2 * sizeof(int32_t) * (size_t)kPrepChunk * (size_t)prep_stride(B);  // double-buffered
}

static int prep_pow2(int64_t B) {
  int P = kPrepThreads;
  while (P < 4 * B) P <<= 1;
  return P;
}

// The prepare launches never touch the table, so they run on a side stream, one chunk ahead of the
// steps that consume them (double-buffered), and disappear behind the training kernels.  The side
// stream and its 4 events are the only state the library keeps; they are created on first use, one
// set per device.
struct AuxState {
  hipStream_t stream = nullptr;
  hipEvent_t prep_done[2] = {nullptr, nullptr};
  hipEvent_t buf_free[2] = {nullptr, nullptr};
  hipEvent_t entry = nullptr;
  bool attr_set = false;
};
static AuxState g_aux[16];

static int aux_for_current_device(AuxState** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= 16) return GE_ENOTSUP;
  AuxState& a = g_aux[dev];
  if (!a.stream) {
    if ((e = hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking)) != hipSuccess) return (int)e;
    for (int i = 0; i < 2; ++i) {
      if ((e = hipEventCreateWithFlags(&a.prep_done[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
      if ((e = hipEventCreateWithFlags(&a.buf_free[i], hipEventDisableTiming)) != hipSuccess) return (int)e;
    }
    if ((e = hipEventCreateWithFlags(&a.entry, hipEventDisableTiming)) != hipSuccess) return (int)e;
  }
  if (!a.attr_set) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(train_prepare_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    a.attr_set = true;
  }
  *out = &a;
  return 0;
}

#define GE_HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

int train_steps_run(float* table, int64_t N, int32_t d, const int32_t* triples, int64_t T, int64_t first_row,
                    int64_t B, int64_t n_steps, const int32_t* id_to_type, const int64_t* type_offsets,
                    int32_t n_types, const int32_t* type_ids, uint64_t seed, uint64_t global_step0,
                    int32_t padded_size, int32_t mode, float margin, float lr0, float decay_steps,
                    float decay_rate, float max_norm, int model, float* loss, int keep_all_losses,
                    int32_t* neg_ws, void* workspace, size_t workspace_bytes, void** ev_pairs, int ev_kernel,
                    hipStream_t st) {
  int32_t* gidx = reinterpret_cast<int32_t*>(workspace);
  float* gval0 = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align_up_sz(sizeof(int32_t) * 6 * (size_t)B, 256));
  const bool fast = train_fast_ok(B, d) && workspace_bytes >= train_ws_bytes(B, d);
  const size_t grow = (size_t)fused_gstride(d);  // floats per gradient row in the fused layout
  const int ring = fast ? grad_ring(B, d) : 1;
  const size_t region_floats = grad_region_bytes(B, d) / sizeof(float);
  float* partials = reinterpret_cast<float*>(reinterpret_cast<char*>(gval0) + (size_t)ring * grad_region_bytes(B, d));
  int32_t* prep_base = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + train_grad_bytes(B, d));
  int lpt_, niter_;
  const bool fused = fast && model == 0 && fused_enabled() && fused_shape_ok(d, table, lpt_, niter_);
  const int64_t buf_ints = (int64_t)kPrepChunk * prep_stride(B);
  const int P = prep_pow2(B);
  const size_t lds = sizeof(unsigned long long) * (size_t)P + sizeof(int) * (kPrepThreads + 1);
  AuxState* aux = nullptr;
  if (fast && n_steps > 0) {
    int rc = aux_for_current_device(&aux);
    if (rc) return rc;
    // earlier work on `st` may still be reading the prep buffers / writing the triples
    GE_HIP_TRY(hipEventRecord(aux->entry, st));
    GE_HIP_TRY(hipStreamWaitEvent(aux->stream, aux->entry, 0));
  }
  auto launch_prepare = [&](int64_t chunk) -> int {
    const int64_t s0 = chunk * kPrepChunk;
    const int64_t todo = (n_steps - s0) < kPrepChunk ? (n_steps - s0) : kPrepChunk;
    int32_t* buf = prep_base + (chunk & 1) * buf_ints;
    hipLaunchKernelGGL(train_prepare_kernel, dim3((unsigned)todo), dim3(kPrepThreads), lds, aux->stream, triples, T,
                       first_row, B, s0, id_to_type, N, type_offsets, n_types, type_ids, seed, global_step0,
                       padded_size, mode, P, model == 0 ? 1 : 0, buf);
    int rc = launch_status();
    if (rc) return rc;
    GE_HIP_TRY(hipEventRecord(aux->prep_done[chunk & 1], aux->stream));
    return 0;
  };
  if (fast && n_steps > 0) {
    int rc = launch_prepare(0);
    if (rc) return rc;
  }
  auto lr_at = [&](uint64_t gs) {
    return decay_steps > 0.f ? lr0 / (1.0f + decay_rate * ((float)gs / decay_steps)) : lr0;
  };
  const int64_t n_chunks = (n_steps + kPrepChunk - 1) / kPrepChunk;
  for (int64_t s = 0; s < n_steps; ++s) {
    const uint64_t gs = global_step0 + (uint64_t)s;
    const float lr = lr_at(gs);
    const int32_t* pos = triples + 3 * step_row(first_row, T, B, s);
    float* loss_s = keep_all_losses ? loss + s * B : loss;
    hipEvent_t e0 = ev_pairs ? (hipEvent_t)ev_pairs[2 * s] : nullptr;
    hipEvent_t e1 = ev_pairs ? (hipEvent_t)ev_pairs[2 * s + 1] : nullptr;
    int rc;
    const int32_t* neg;
    const int32_t* step_prep = nullptr;
    if (e0 && ev_kernel == 0) (void)hipEventRecord(e0, st);
    if (fast) {
      const int64_t chunk = s / kPrepChunk, in_chunk = s % kPrepChunk;
      if (in_chunk == 0) {
        // the next chunk is prepared on the side stream while this one trains; its buffer was last
        // read by chunk-1, whose completion `buf_free` marks
        if (chunk + 1 < n_chunks) {
          if (chunk >= 1) GE_HIP_TRY(hipStreamWaitEvent(aux->stream, aux->buf_free[(chunk + 1) & 1], 0));
          rc = launch_prepare(chunk + 1);
          if (rc) return rc;
        }
        GE_HIP_TRY(hipStreamWaitEvent(st, aux->prep_done[chunk & 1], 0));
      }
      step_prep = prep_base + (chunk & 1) * buf_ints + in_chunk * prep_stride(B);
      neg = step_prep;
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
    if (fused) {
      // one launch per step: gradients + sparse update ("last arriver executes", ge_fused.hip)
      int32_t* sp = const_cast<int32_t*>(step_prep);
      FusedArgs fa{sp + off_slot_item(B), sp + off_items(B), sp + off_occ(B), sp + off_item_cnt(B),
                   sp + off_row_cnt(B), partials, gidx, gval, (int)grow, 0};
      rc = complex_fused_step_launch(table, N, d, pos, neg, B, margin, lr, max_norm, loss_s, fa, st, g0, g1);
      if (rc) return rc;
      if (a0) { (void)hipEventRecord(a0, st); (void)hipEventRecord(a1, st); }
      if ((s % kPrepChunk) == kPrepChunk - 1 || s == n_steps - 1)
        GE_HIP_TRY(hipEventRecord(aux->buf_free[(s / kPrepChunk) & 1], st));
      continue;
    }
    rc = model == 0 ? complex_hinge_grad_launch(table, N, d, pos, neg, B, margin, lr, max_norm, loss_s, gidx, gval, st, g0, g1,
                                                fast ? step_prep + off_slot_item(B) : nullptr)
                    : hole_hinge_grad_launch(table, N, d, pos, neg, B, margin, lr, max_norm, loss_s, gidx, gval, st, g0, g1);
    if (rc) return rc;
    if (fast) rc = apply_sorted_launch(table, d, B, step_prep + off_items(B), step_prep + off_nitems(B), step_prep + off_islots(B), gidx, gval, st, a0, a1);
    else rc = scatter_add_rows_launch(table, N, d, gidx, gval, 6 * B, st, a0, a1);
    if (rc) return rc;
    if (fast && ((s % kPrepChunk) == kPrepChunk - 1 || s == n_steps - 1))
      GE_HIP_TRY(hipEventRecord(aux->buf_free[(s / kPrepChunk) & 1], st));
  }
  // keep the caller's neg_ws meaningful: the last step's negatives
  if (fast && n_steps > 0) {
    const int64_t ls = n_steps - 1;
    const int32_t* last = prep_base + ((ls / kPrepChunk) & 1) * buf_ints + (ls % kPrepChunk) * prep_stride(B);
    GE_HIP_TRY(hipMemcpyAsync(neg_ws, last, sizeof(int32_t) * 3 * (size_t)B, hipMemcpyDeviceToDevice, st));
  }
  return 0;
}

}  // namespace ge
