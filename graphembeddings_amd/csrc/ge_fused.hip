// ge_fused.hip -- one launch per training step: gradients AND the sparse update, with no grid
// barrier and no second kernel ("last arriver executes").  OPT-IN (ge_set_fused_step(1) /
// GE_FUSED_STEP=1): correct and bitwise-validated against the two-launch step, but measured slower
// on MI355X at B=4096 (36 us vs 22 us per step, FB15k-shaped batch) -- the dependency chain through
// the hottest rows (16-slot items -> partial sums -> row total) is serial, whereas the two-launch
// step spreads the same work over ~3 k independent wavefronts.  Kept as the basis for a dataflow
// variant that hands hot rows to dedicated waves.
//
// The two-kernel step (hinge_grad, then apply_sorted) pays two launches of ~10 us each although both
// are latency-bound at B=4096.  Fusing them needs one property: a table row may be updated as soon
// as every pair that READS it has finished computing -- and the pairs that read row X in a step are
// exactly the pairs that contribute a gradient slot to X.  The prepared index (ge_train.hip) already
// groups the slots of a step by row into items of <= 16 slots, so:
//
//   every pair group: gather -> clip -> score -> hinge -> gradient rows (as complex_hinge_grad_kernel),
//       written WRITE-THROUGH (sc1) to the gradient buffer, drained (s_waitcnt vmcnt(0)), then one
//       agent-scope atomic add on the counter of every item it feeds;
//   the group whose add completes an item (old+1 == item size) sums the item's gradient rows (sc1
//       loads, fixed slot order -> bitwise reproducible) and applies them to the table row with a plain
//       read-modify-write;
//   rows split over several items (> 16 occurrences in the step) go through one more level: each
//       completed item publishes its partial sum (sc1) and bumps the row's counter; the group that
//       completes the row adds the partials in item order and applies.  (The two-kernel path combines
//       these with float atomics; here even hot rows are reproducible.)
//
// Nobody ever waits: no spin, no co-residency requirement, no deadlock.  Visibility follows the
// sc1/drain/atomic publish form of cdna_hip_programming.md Guideline 16 (R1): every handed-off byte
// (gradient rows, slot flags, partial sums) is stored sc1 and loaded sc1; gradient rows are padded to
// 128-byte lines so that no line is shared between two publishers.  The counters are zeroed by the
// prepare kernel (off the critical path), once per use.
#include "ge_complex_dev.h"

namespace ge {

using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
constexpr int kSlotDirect = -2;  // slot_item code written by train_prepare_kernel
constexpr int kAuxSC1 = 16;  // aux bit 4 = sc1 (agent-coherent write-through / L1-bypassing load)

struct FusedArgs {
  const int32_t* slot_item;  // [6B]  item of each IndexedSlices slot (-1 none)
  const int32_t* items;      // [n_items,5] row, start, cnt|multi<<30, row_first_item, row_n_items (at first item)
  const int32_t* occ;        // [4B]  slot ids sorted by row
  int32_t* item_cnt;         // [4B]  arrivals per item   (zeroed by prepare)
  int32_t* row_cnt;          // [4B]  completed items per split row, indexed by its first item
  float* partials;           // [4B, gstride] partial sums of split rows
  int32_t* gidx;             // [6B]
  float* gval;               // [6B, gstride]
  int gstride;               // floats per gradient row (multiple of 32 = one 128-B line)
  int debug;
};

__device__ __forceinline__ void st_sc1_v4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, const float (&v)[4]) {
  u32x4 u = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
  __builtin_amdgcn_raw_buffer_store_b128(u, rsrc, byte_off, 0, kAuxSC1);
}
__device__ __forceinline__ void ld_sc1_v4(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, float (&v)[4]) {
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, kAuxSC1);
  v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
}

// VEC is fixed to 4 (d % 8 == 0): every lane moves 16-byte pieces.
template <int LPT, int NITER>
__global__ __launch_bounds__(kBlock) void complex_fused_step_kernel(
    float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float lr, float max_norm,
    float* __restrict__ loss, FusedArgs fa) {
  constexpr int VEC = 4;
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  const float neg_lr = -lr;
  const uint32_t row_bytes = (uint32_t)fa.gstride * 4u;
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(fa.gval, 0, (int)(6 * B * row_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(fa.partials, 0, (int)(4 * B * row_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t i_rsrc = __builtin_amdgcn_make_buffer_rsrc(fa.gidx, 0, (int)(6 * B * 4), 0x00020000);

  for (int64_t base = wave * GPW; base < B; base += nwaves * GPW) {
    const int64_t g = base + grp;
    const bool live = g < B;
    int32_t p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = pos[3 * g + c]; n[c] = neg[3 * g + c]; }
    }
    const bool bad = bad3(N, p[0], p[1], p[2]) || bad3(N, n[0], n[1], n[2]);
    if (bad) { p[0] = p[1] = p[2] = n[0] = n[1] = n[2] = 0; }
    // ------------------------------------------------------------------ gradients (publish)
    {
      Row<VEC, NITER> xp[3], xn[3];
#pragma unroll
      for (int X = 0; X < 3; ++X) load_row<VEC, LPT, NITER>(table, p[X], d, k, nvec, sub, xp[X]);
#pragma unroll
      for (int X = 0; X < 3; ++X) load_row<VEC, LPT, NITER>(table, n[X], d, k, nvec, sub, xn[X]);
      const SideFwd fp = side_forward<VEC, LPT, NITER>(xp[0], xp[1], xp[2], max_norm);
      const SideFwd fn = side_forward<VEC, LPT, NITER>(xn[0], xn[1], xn[2], max_norm);
      const float pre = fp.sig - fn.sig + margin;
      const bool on = live && !bad && (pre >= 0.f);  // MaximumGrad: x >= y
      if (live && sub == 0) loss[g] = bad ? __builtin_nanf("") : fmaxf(pre, 0.f);
      const float cp = fp.sig * (1.f - fp.sig), cn = -fn.sig * (1.f - fn.sig);
#pragma unroll
      for (int X = 0; X < 3; ++X) {
        const bool same = p[X] == n[X];
        const uint32_t rowP = (uint32_t)(g * 6 + X), rowN = (uint32_t)(g * 6 + 3 + X);
        // sole contributor of its row (tagged by the prepare kernel): update the table row right here
        const bool dirP = live && fa.slot_item[rowP] == kSlotDirect;
        const bool dirN = live && !same && fa.slot_item[rowN] == kSlotDirect;
        if (live && sub == 0) {
          if (!dirP) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(on ? p[X] : -1), i_rsrc, rowP * 4u, 0, kAuxSC1);
          if (!same && !dirN) __builtin_amdgcn_raw_buffer_store_b32((unsigned)(on ? n[X] : -1), i_rsrc, rowN * 4u, 0, kAuxSC1);
        }
        if (!on) continue;
        const RowCoef kp = row_coef(cp, fp, X, max_norm, neg_lr);
        const RowCoef kn = row_coef(cn, fn, X, max_norm, neg_lr);
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
          const int j = sub + it * LPT;
          if (j >= nvec) continue;
          float pre_[VEC], pim_[VEC], nre_[VEC], nim_[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            float gre, gim;
            graw<VEC, NITER>(X, xp[0], xp[1], xp[2], it, v, gre, gim);
            pre_[v] = kp.alpha * gre + kp.beta * xp[X].re[it][v];
            pim_[v] = kp.alpha * gim + kp.beta * xp[X].im[it][v];
            graw<VEC, NITER>(X, xn[0], xn[1], xn[2], it, v, gre, gim);
            nre_[v] = kn.alpha * gre + kn.beta * xn[X].re[it][v];
            nim_[v] = kn.alpha * gim + kn.beta * xn[X].im[it][v];
          }
          if (same) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { pre_[v] += nre_[v]; pim_[v] += nim_[v]; }
          } else if (dirN) {
            float* dn = table + (int64_t)n[X] * d;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { nre_[v] += xn[X].re[it][v]; nim_[v] += xn[X].im[it][v]; }
            store_vec<VEC>(dn + j * VEC, nre_);
            store_vec<VEC>(dn + k + j * VEC, nim_);
          } else {
            st_sc1_v4(g_rsrc, rowN * row_bytes + (uint32_t)(j * VEC) * 4u, nre_);
            st_sc1_v4(g_rsrc, rowN * row_bytes + (uint32_t)(k + j * VEC) * 4u, nim_);
          }
          if (dirP) {
            float* dp = table + (int64_t)p[X] * d;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { pre_[v] += xp[X].re[it][v]; pim_[v] += xp[X].im[it][v]; }
            store_vec<VEC>(dp + j * VEC, pre_);
            store_vec<VEC>(dp + k + j * VEC, pim_);
          } else {
            st_sc1_v4(g_rsrc, rowP * row_bytes + (uint32_t)(j * VEC) * 4u, pre_);
            st_sc1_v4(g_rsrc, rowP * row_bytes + (uint32_t)(k + j * VEC) * 4u, pim_);
          }
        }
      }
    }
    // every lane of this wave has drained its write-through stores before anyone signals
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ------------------------------------------------------------------ signal: one lane per slot
    // lane sub = 0,1,2 -> positive slots h+,t+,r+; lane sub = 3 -> the (single) negative-side slot
    int my_item = -1;
    bool my_last = false;
    if (live && !bad && sub < 4) {
      int slot = -1;
      if (sub < 3) slot = (int)(g * 6 + sub);
      else {
        const int X = (p[0] != n[0]) ? 0 : (p[1] != n[1]) ? 1 : (p[2] != n[2]) ? 2 : -1;
        if (X >= 0) slot = (int)(g * 6 + 3 + X);
      }
      if (slot >= 0) {
        const int it = fa.slot_item[slot];
        if (it >= 0) {
          const int old = __hip_atomic_fetch_add(fa.item_cnt + it, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          my_item = it;
          my_last = (old + 1) == (fa.items[5 * it + 2] & 0x3FFFFFFF);
        }
      }
    }
    // no instruction: keeps the compiler from moving the sc1 loads below above the counter adds
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ------------------------------------------------------------------ last arriver executes
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int it = __shfl(my_item, grp * LPT + q, kWave);
      const bool do_it = __shfl(my_last ? 1 : 0, grp * LPT + q, kWave) != 0;
      if (!do_it) continue;  // uniform within the group
      const int row = fa.items[5 * it], start = fa.items[5 * it + 1], cm = fa.items[5 * it + 2];
      const int cnt = cm & 0x3FFFFFFF;
      const bool multi = (cm >> 30) & 1;
      const int row_first = fa.items[5 * it + 3];
      Row<VEC, NITER> acc;
#pragma unroll
      for (int i2 = 0; i2 < NITER; ++i2)
#pragma unroll
        for (int v = 0; v < VEC; ++v) { acc.re[i2][v] = 0.f; acc.im[i2][v] = 0.f; }
      // lane o (< cnt <= 16 <= LPT) fetches slot o and whether its pair was hinge-active, so the
      // dependent chain is one round trip for the whole item instead of one per slot
      int slot_v = 0;
      bool act_v = false;
      if (sub < cnt) {
        slot_v = fa.occ[start + sub];
        act_v = (int)__builtin_amdgcn_raw_buffer_load_b32(i_rsrc, (uint32_t)slot_v * 4u, 0, kAuxSC1) >= 0;
      }
      const unsigned live_mask = (unsigned)((__ballot(act_v) >> (grp * LPT)) & 0xFFFFull);
      const bool any = live_mask != 0u;
      if (any) {
        for (int o = 0; o < cnt; o += 4) {
          int sl[4];
          bool on4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int src = (o + u < cnt) ? (o + u) : o;
            sl[u] = __shfl(slot_v, grp * LPT + src, kWave);
            on4[u] = (o + u < cnt) && ((live_mask >> (o + u)) & 1u);
          }
#pragma unroll
          for (int i2 = 0; i2 < NITER; ++i2) {
            const int j = sub + i2 * LPT;
            if (j >= nvec) continue;
            float a[4][4], b[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // 8 independent 16-byte loads in flight
              ld_sc1_v4(g_rsrc, (uint32_t)sl[u] * row_bytes + (uint32_t)(j * VEC) * 4u, a[u]);
              ld_sc1_v4(g_rsrc, (uint32_t)sl[u] * row_bytes + (uint32_t)(k + j * VEC) * 4u, b[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)    // summed in slot order: same bits as the two-kernel path
#pragma unroll
              for (int v = 0; v < VEC; ++v) {
                acc.re[i2][v] += on4[u] ? a[u][v] : 0.f;
                acc.im[i2][v] += on4[u] ? b[u][v] : 0.f;
              }
          }
        }
      }
      bool apply = !multi && any;
      int n_row_items = 1;
      if (multi) {
        // publish this item's partial sum, then count it on the row
#pragma unroll
        for (int i2 = 0; i2 < NITER; ++i2) {
          const int j = sub + i2 * LPT;
          if (j >= nvec) continue;
          st_sc1_v4(p_rsrc, (uint32_t)it * row_bytes + (uint32_t)(j * VEC) * 4u, acc.re[i2]);
          st_sc1_v4(p_rsrc, (uint32_t)it * row_bytes + (uint32_t)(k + j * VEC) * 4u, acc.im[i2]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        n_row_items = fa.items[5 * row_first + 4];
        int old = 0;
        if (sub == 0) old = __hip_atomic_fetch_add(fa.row_cnt + row_first, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        old = __shfl(old, grp * LPT, kWave);
        apply = (old + 1) == n_row_items;
        if (apply) {
          // sum the row's partials in item order (fixed order: reproducible)
#pragma unroll
          for (int i2 = 0; i2 < NITER; ++i2)
#pragma unroll
            for (int v = 0; v < VEC; ++v) { acc.re[i2][v] = 0.f; acc.im[i2][v] = 0.f; }
          for (int m = 0; m < n_row_items; m += 4) {
#pragma unroll
            for (int i2 = 0; i2 < NITER; ++i2) {
              const int j = sub + i2 * LPT;
              if (j >= nvec) continue;
              float a[4][4], b[4][4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const int mm = (m + u < n_row_items) ? (m + u) : m;
                ld_sc1_v4(p_rsrc, (uint32_t)(row_first + mm) * row_bytes + (uint32_t)(j * VEC) * 4u, a[u]);
                ld_sc1_v4(p_rsrc, (uint32_t)(row_first + mm) * row_bytes + (uint32_t)(k + j * VEC) * 4u, b[u]);
              }
#pragma unroll
              for (int u = 0; u < 4; ++u)   // item order: reproducible
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                  acc.re[i2][v] += (m + u < n_row_items) ? a[u][v] : 0.f;
                  acc.im[i2][v] += (m + u < n_row_items) ? b[u][v] : 0.f;
                }
            }
          }
        }
      }
      if (apply) {
        float* dst = table + (int64_t)row * d;
#pragma unroll
        for (int i2 = 0; i2 < NITER; ++i2) {
          const int j = sub + i2 * LPT;
          if (j >= nvec) continue;
          float tr[4], ti[4];
          load_vec<VEC>(dst + j * VEC, tr);
          load_vec<VEC>(dst + k + j * VEC, ti);
#pragma unroll
          for (int v = 0; v < VEC; ++v) { tr[v] += acc.re[i2][v]; ti[v] += acc.im[i2][v]; }
          store_vec<VEC>(dst + j * VEC, tr);
          store_vec<VEC>(dst + k + j * VEC, ti);
        }
      }
    }
  }
}

bool fused_shape_ok(int d, const void* table, int& lpt, int& niter) {
  if (d <= 0 || (d % 8) != 0 || (reinterpret_cast<uintptr_t>(table) % 16) != 0) return false;
  const int nvec = d / 8;
  lpt = 16;
  while (lpt < 64 && lpt < nvec) lpt <<= 1;
  niter = (nvec + lpt - 1) / lpt;
  return niter <= 2;
}

int fused_gstride(int d) { return (d + 31) / 32 * 32; }

int complex_fused_step_launch(float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg, int64_t B,
                              float margin, float lr, float max_norm, float* loss, const FusedArgs& fa,
                              hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
  int lpt, niter;
  if (!fused_shape_ok(d, table, lpt, niter)) return GE_ENOTSUP;
  if (B == 0) return 0;
  const int gpb = (kBlock / kWave) * (kWave / lpt);
  const int grid = grid_for(B, gpb);
#define LF(L, NI) hipExtLaunchKernelGGL((complex_fused_step_kernel<L, NI>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, table, N, d, pos, neg, B, margin, lr, max_norm, loss, fa)
  const int key = lpt * 10 + niter;
  switch (key) {
    case 161: LF(16, 1); break;
    case 321: LF(32, 1); break;
    case 641: LF(64, 1); break;
    case 642: LF(64, 2); break;
    default: return GE_ENOTSUP;
  }
#undef LF
  return launch_status();
}

}  // namespace ge
