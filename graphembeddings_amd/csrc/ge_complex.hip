// ge_complex.hip -- fused gather -> clip -> ComplEx score -> sigmoid -> hinge -> row-gradient kernels.
//
// Reference path replaced (all of it stock TF ops on CPU in the reference):
//   get_embedding      holE.py:161-168   (Gather, clip_by_norm chain graph.pbtxt:3108-3596, Complex)
//   evaluate_triples   holE.py:179-198   (Mul, Conj, Real, Sum, Sigmoid)
//   evaluate_batch     holE.py:222-234   (Maximum)
//   minimize()         holE.py:296       (autodiff of SUM(loss) -> IndexedSlices -> ScatterSub)
//
// Work decomposition (CDNA4): a group of LPT lanes (16/32/64, a power-of-two slice of one 64-lane
// wavefront) owns one triple / one (pos,neg) pair.  Lane `sub` of the group holds VEC consecutive
// real parts and the VEC matching imaginary parts of every row (row layout [Re(d/2) | Im(d/2)],
// holE.py:164-166), so each row is read with two fully coalesced 16-byte-per-lane loads
// (d=200: 25 lanes x 16 B = 400 B per half row) and the complex arithmetic is lane-local.
// Norms and the score are reduced across the group with wave shuffles; nothing goes through LDS.
// The clip is a per-row scalar and the score is trilinear, so s = sh*st*sr*s_raw and, by Euler's
// identity for a trilinear form, (d s/d y_X) . x_X = s / scale_X: the backward pass through the
// clip needs no reduction beyond the forward ones.
#include "ge_complex_dev.h"

namespace ge {

// ---------------------------------------------------------------- evaluate_triples
template <bool SPEC, int VEC, int LPT, int NITER>
__global__ __launch_bounds__(kBlock) void complex_score_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ triples, int64_t B,
    float max_norm, int apply_sigmoid, float* __restrict__ out, float label, float l2,
    const float* __restrict__ table_sumsq, int64_t ld) {
  // ld: floats between consecutive rows (= d for the reference's dense [N,d] table; ge_complex_score_strided
  // measures padded layouts, profiles/r03_padded_rows.txt)
  // apply_sigmoid: 0 raw score, 1 sigmoid (holE.py:198), 2 the --log_loss branch of evaluate_triples
  // (holE.py:194-196): log(1 + exp(-label * score)) + l2 * l2_loss(whole table)
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  const float wscale = 1.0f / (float)d;   // SPEC only: Parseval / correlation-theorem factor
  for (int64_t base = wave * GPW; base < B; base += nwaves * GPW) {
    const int64_t g = base + grp;
    const bool live = g < B;
    int32_t hi = 0, ti = 0, ri = 0;
    if (live) { hi = triples[3 * g]; ti = triples[3 * g + 1]; ri = triples[3 * g + 2]; }
    const bool bad = bad3(N, hi, ti, ri);
    if (bad) { hi = ti = ri = 0; }
    Row<VEC, NITER> h, t, r;
    load_row_at<VEC, LPT, NITER>(table + (int64_t)hi * ld, k, nvec, sub, h);
    load_row_at<VEC, LPT, NITER>(table + (int64_t)ti * ld, k, nvec, sub, t);
    load_row_at<VEC, LPT, NITER>(table + (int64_t)ri * ld, k, nvec, sub, r);
    const SideFwd f = side_forward<SPEC, VEC, LPT, NITER>(h, t, r, max_norm, sub == 0, wscale);
    if (live && sub == 0) {
      float v = apply_sigmoid == 1 ? f.sig : f.s;
      if (apply_sigmoid == 2) {
        const float z = -label * f.s;
        v = (z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z))) + l2 * 0.5f * table_sumsq[0];
      }
      out[g] = bad ? __builtin_nanf("") : v;
    }
  }
}

// ---------------------------------------------------------------- evaluate_batch (forward only)
template <bool SPEC, int VEC, int LPT, int NITER>
__global__ __launch_bounds__(kBlock) void complex_hinge_loss_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float max_norm,
    float* __restrict__ loss, float* __restrict__ sig_out) {
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  const float wscale = 1.0f / (float)d;   // SPEC only: Parseval / correlation-theorem factor
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
    Row<VEC, NITER> h, t, r;
    load_row<VEC, LPT, NITER>(table, p[0], d, k, nvec, sub, h);
    load_row<VEC, LPT, NITER>(table, p[1], d, k, nvec, sub, t);
    load_row<VEC, LPT, NITER>(table, p[2], d, k, nvec, sub, r);
    const SideFwd fp = side_forward<SPEC, VEC, LPT, NITER>(h, t, r, max_norm, sub == 0, wscale);
    load_row<VEC, LPT, NITER>(table, n[0], d, k, nvec, sub, h);
    load_row<VEC, LPT, NITER>(table, n[1], d, k, nvec, sub, t);
    load_row<VEC, LPT, NITER>(table, n[2], d, k, nvec, sub, r);
    const SideFwd fn = side_forward<SPEC, VEC, LPT, NITER>(h, t, r, max_norm, sub == 0, wscale);
    if (live && sub == 0) {
      const float nanv = __builtin_nanf("");
      loss[g] = bad ? nanv : fmaxf(fp.sig - fn.sig + margin, 0.f);  // holE.py:231
      if (sig_out) { sig_out[g] = bad ? nanv : fp.sig; sig_out[B + g] = bad ? nanv : fn.sig; }
    }
  }
}

// ---------------------------------------------------------------- hinge + row gradients
// d(sum_i L_i)/d(raw rows), pre-multiplied by -lr, as IndexedSlices (6 slots per pair:
// h+, t+, r+, h-, t-, r-).  Through the clip (MinimumGrad routes to the rsqrt branch iff
// rsqrt(ss) <= 1/c):  gx = c*(gy*inv - x*(gy.x)*inv^3)  with  gy.x = coef*P_X*s_raw.
// Written per row X as  gx = alpha_X * Graw_X + beta_X * x_X.
template <bool SPEC, int VEC, int LPT, int NITER>
__global__ __launch_bounds__(kBlock) void complex_hinge_grad_kernel(
    const float* rows, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float lr, float max_norm,
    float* __restrict__ loss, int32_t* __restrict__ grad_idx, float* __restrict__ grad_val,
    const int32_t* __restrict__ slot_item, float* table_rw) {
  // slot_item + table_rw (training loop only, else null; table_rw aliases `rows`, which is therefore
  // NOT declared __restrict__ const in that instantiation): a slot tagged kSlotDirect is the ONLY
  // gradient slot of its table row in this step -- this pair is the row's only reader and writer -- so
  // the update is applied right here (rows + (-lr*g)) and no gradient row is written for it.
  constexpr int kSlotDirect = -2;
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  const float wscale = 1.0f / (float)d;   // SPEC only: Parseval / correlation-theorem factor
  const float neg_lr = -lr;
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
    // the six "applied by the producer" tags of this pair, requested together with its ids (not after the forward)
    int32_t tag[6] = {0, 0, 0, 0, 0, 0};
    if (slot_item && live) {
#pragma unroll
      for (int c = 0; c < 6; ++c) tag[c] = slot_item[g * 6 + c];
    }
    Row<VEC, NITER> xp[3], xn[3];
#pragma unroll
    for (int X = 0; X < 3; ++X) load_row<VEC, LPT, NITER>(rows, p[X], d, k, nvec, sub, xp[X]);
#pragma unroll
    for (int X = 0; X < 3; ++X) load_row<VEC, LPT, NITER>(rows, n[X], d, k, nvec, sub, xn[X]);
    const SideFwd fp = side_forward<SPEC, VEC, LPT, NITER>(xp[0], xp[1], xp[2], max_norm, sub == 0, wscale);
    const SideFwd fn = side_forward<SPEC, VEC, LPT, NITER>(xn[0], xn[1], xn[2], max_norm, sub == 0, wscale);
    const float pre = fp.sig - fn.sig + margin;
    const bool on = live && !bad && (pre >= 0.f);  // MaximumGrad: x >= y
    if (live && sub == 0) loss[g] = bad ? __builtin_nanf("") : fmaxf(pre, 0.f);
    const float cp = fp.sig * (1.f - fp.sig), cn = -fn.sig * (1.f - fn.sig);
#pragma unroll
    for (int X = 0; X < 3; ++X) {
      const bool same = p[X] == n[X];
      const int64_t rowP = g * 6 + X, rowN = g * 6 + 3 + X;
      const bool dirP = slot_item && live && tag[X] == kSlotDirect;
      const bool dirN = slot_item && live && !same && tag[3 + X] == kSlotDirect;
      if (live && sub == 0) {
        grad_idx[rowP] = (on && !dirP) ? p[X] : -1;
        grad_idx[rowN] = (on && !same && !dirN) ? n[X] : -1;
      }
      if (!on) continue;
      const RowCoef kp = row_coef(cp, fp, X, max_norm, neg_lr);
      const RowCoef kn = row_coef(cn, fn, X, max_norm, neg_lr);
      float* gp = dirP ? table_rw + (int64_t)p[X] * d : grad_val + rowP * d;
      float* gn = dirN ? table_rw + (int64_t)n[X] * d : grad_val + rowN * d;
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float pre_[VEC], pim_[VEC], nre_[VEC], nim_[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float gre, gim;
          graw<SPEC, VEC, NITER>(X, xp[0], xp[1], xp[2], it, v, sub == 0, gre, gim);
          pre_[v] = kp.alpha * gre + kp.beta * xp[X].re[it][v];
          pim_[v] = kp.alpha * gim + kp.beta * xp[X].im[it][v];
          graw<SPEC, VEC, NITER>(X, xn[0], xn[1], xn[2], it, v, sub == 0, gre, gim);
          nre_[v] = kn.alpha * gre + kn.beta * xn[X].re[it][v];
          nim_[v] = kn.alpha * gim + kn.beta * xn[X].im[it][v];
        }
        if (same) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) { pre_[v] += nre_[v]; pim_[v] += nim_[v]; }
        } else {
          if (dirN) {   // x' = x + (-lr g): same bits as the apply kernel's 0 + g then x + sum
#pragma unroll
            for (int v = 0; v < VEC; ++v) { nre_[v] += xn[X].re[it][v]; nim_[v] += xn[X].im[it][v]; }
          }
          store_vec<VEC>(gn + j * VEC, nre_);
          store_vec<VEC>(gn + k + j * VEC, nim_);
        }
        if (dirP) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) { pre_[v] += xp[X].re[it][v]; pim_[v] += xp[X].im[it][v]; }
        }
        store_vec<VEC>(gp + j * VEC, pre_);
        store_vec<VEC>(gp + k + j * VEC, pim_);
      }
    }
  }
}

// ---------------------------------------------------------------- hinge + row gradients, planned steps
// The gradient kernel of the steps that come with a plan beyond the direct tags: SHARD (two row stores, the row-sharded
// step) and / or ORD (pairs walked in relation order, large batches).  The one-tile training step and the single-step
// API keep the kernel above unchanged: its code is tuned to the 4096-pair step, where every instruction between
// the id loads and the row loads shows.
// d(sum_i L_i)/d(raw rows), pre-multiplied by -lr, as IndexedSlices (6 slots per pair: h+, t+, r+, h-, t-, r-; the two
// negative-side slots of the rows the negative shares with the positive never exist).  Through the clip (MinimumGrad
// routes to the rsqrt branch iff rsqrt(ss) <= 1/c):  gx = c*(gy*inv - x*(gy.x)*inv^3)  with  gy.x = coef*P_X*s_raw,
// written per row X as  gx = alpha_X * Graw_X + beta_X * x_X.
//
// One (pos, neg) pair = FOUR distinct rows, not six: the negative is the positive with ONE entity replaced
// (holE.py:104-112, 137-140; the prepared record holds exactly these four sort keys, ge_prep.h).  With F the entity
// both sides share, V / C the positive's and the negative's other entity and r~ = (Re r, sigma Im r), sigma = +1 when
// the tail is the replaced column and -1 when the head is,
//     score(v) = sum Re(w conj v),   w = F r~          (Re<h,r,conj t> with h = F resp. t = F)
//     d score / d v = w,   d score / d F = conj(r~) v,   d score / d r = (Re, sigma Im) of conj(F) v
// so both sides share w, the shared rows' loads, norms and clip scales, and the F and r gradients are ONE complex
// product each of the coefficient-weighted sum of V and C: 32 row registers a lane instead of 48, 6 group reductions
// instead of 8 (on DPP adds, not LDS permutes), a third fewer vector instructions -- four waves per SIMD (round 3: three).
// All row loads are unconditional (lanes past the row re-read its last vector and are masked out of the sums): a
// predicated load puts every row behind its own branch and wait.

// sum over a group of LPT consecutive lanes on the DPP path: xor 1, xor 2 (quad_perm), row_half_mirror, row_mirror give
// every lane of a 16-lane row the row's sum in four v_add_f32_dpp; the two rows of a 32-lane group meet through
// ds_swizzle (no address register), the two halves of the wave through one permute
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int LPT>
__device__ __forceinline__ float group_sum_dpp(float v) {
  static_assert(LPT == 16 || LPT == 32 || LPT == 64, "lane group");
  v = dpp_add<0xB1>(v);     // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);     // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);    // row_half_mirror
  v = dpp_add<0x140>(v);    // row_mirror
  if (LPT >= 32) v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));   // lane ^ 16
  if (LPT >= 64) v += __shfl_xor(v, 32, kWave);
  return v;
}

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ void load_row_clamped(const float* __restrict__ p, int k, int nvec, int sub, Row<VEC, NITER>& R) {
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int j = min(sub + it * LPT, nvec - 1);
    load_vec<VEC>(p + j * VEC, R.re[it]);
    load_vec<VEC>(p + k + j * VEC, R.im[it]);
  }
}

template <bool SPEC, bool SHARD, bool ORD, int VEC, int LPT, int NITER, bool PEER = false>
// (four waves per SIMD: five -- 96 registers -- spill 13 and run a third slower; 1024 workgroups of longer runs instead of
// 2048 were 7 % slower)
__global__ __launch_bounds__(kBlock, (VEC * NITER <= 4 ? 4 : 2)) void complex_hinge_grad_plan_kernel(
    const float* rows, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float lr, float max_norm,
    float* __restrict__ loss, int32_t* __restrict__ grad_idx, float* __restrict__ grad_val,
    const int32_t* __restrict__ slot_item, float* table_rw, ShardGrad sg, const int32_t* __restrict__ order) {
  // slot_item + table_rw (training loop only, else null; table_rw aliases `rows`, which is therefore
  // NOT declared __restrict__ const in that instantiation): a slot tagged kSlotDirect is the ONLY
  // gradient slot of its table row in this step -- this pair is the row's only reader and writer -- so
  // the update is applied right here (rows + (-lr*g)) and no gradient row is written for it.
  // SHARD: pos / neg are null; the pair's rows are named by sg.pos_src / sg.neg_src (see ShardGrad).
  // order (large batches, else null): the pairs sorted by relation row.  Each wave then walks a CONTIGUOUS run of that
  // order and sums the relation row's gradient of consecutive pairs in registers: one gradient row is written per run
  // of equal relations (into the slot of the run's last contributing pair; the others' relation slots stay empty)
  // instead of one per pair -- a quarter of all gradient-row bytes belong to a handful of relation rows.
  constexpr int kSlotDirect = -2;
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int wave = (int)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int nwaves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
  const int nB = (int)B;                              // (B <= 2^24 pairs a step, train_fast_ok)
  const int k = d >> 1, nvec = k / VEC;
  const float wscale = 1.0f / (float)d;   // SPEC only: Parseval / correlation-theorem factor
  const float neg_lr = -lr;
  // which of a lane's vectors are part of the row (the others hold copies of the last one, masked out of every sum)
  float act[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) act[it] = (sub + it * LPT < nvec) ? 1.f : 0.f;
  // ORD: this wave's contiguous run [first, last) of the relation order
  const int per = ORD ? ((nB + nwaves - 1) / nwaves + GPW - 1) / GPW * GPW : 0;
  float racc_re[NITER][VEC], racc_im[NITER][VEC];   // running relation-row gradient of this lane group's run
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) { racc_re[it][v] = 0.f; racc_im[it][v] = 0.f; }
  int32_t cur_rel = -1;
  int cur_slot = -1;                                  // >= 0: the slot the running sum will be written to
  auto flush = [&]() {
    float* o = grad_val + (int64_t)cur_slot * d;
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int j = sub + it * LPT;
      if (j < nvec) {
        store_vec<VEC>(o + j * VEC, racc_re[it]);
        store_vec<VEC>(o + k + j * VEC, racc_im[it]);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) { racc_re[it][v] = 0.f; racc_im[it][v] = 0.f; }
    }
    if (sub == 0) grad_idx[cur_slot] = cur_rel;
    cur_slot = -1;
  };
  // The id words of pair g (5: the positive triple + the negative's head and tail -- its relation is the positive's;
  // SHARD: 3 row sources + the corrupted entity's) and the tags of its slots 0 .. 4 (slot 5 never exists).  A lane group
  // works on ONE pair, so everything that names the pair is wave-uniform per group: it is read on the SCALAR path
  // (s_load through the constant address space: the arrays were written by earlier kernels) into SGPRs -- no vector
  // registers held for the pair ahead, no place in the vector-memory queue, whose wait counter retires loads and
  // stores in issue order (ids requested before a pair's stores and used after them would make the next pair's row
  // requests wait for those stores).
  typedef const __attribute__((address_space(4))) int32_t* kptr_t;
  auto kld = [](const int32_t* p, int i) -> int32_t { return ((kptr_t)(uintptr_t)p)[i]; };
  auto load_ids = [&](int g, int32_t (&raw)[5], int32_t (&tag)[5]) {     // g wave-uniform
    if (SHARD) {
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[c] = kld(sg.pos_src, 3 * g + c);
      raw[3] = kld(sg.neg_src, g); raw[4] = 0;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[c] = kld(pos, 3 * g + c);
      raw[3] = kld(neg, 3 * g); raw[4] = kld(neg, 3 * g + 1);
    }
#pragma unroll
    for (int c = 0; c < 5; ++c) tag[c] = slot_item ? kld(slot_item, g * 6 + c) : 0;
  };
  // lane group j's value of a per-group scalar
  auto pick = [&](const int32_t (&v)[GPW]) -> int32_t {
    int32_t r = v[0];
#pragma unroll
    for (int j = 1; j < GPW; ++j) r = (grp == j) ? v[j] : r;
    return r;
  };
  auto body = [&](const int g, const bool live, const int32_t (&raw)[5], const int32_t (&tag)[5], auto&& after_loads) {
    int32_t p0 = 0, p1 = 0, p2 = 0, c = 0;
    bool hs = false, bad;               // hs: the head is the replaced column
    if (SHARD) {
      int32_t ns = -1;
      if (live) { p0 = raw[0]; p1 = raw[1]; p2 = raw[2]; ns = raw[3]; }
      bad = p0 < 0 || p1 < 0 || p2 < 0;
      hs = ns >= 0 && (ns & 1) == 0;
      c = ns >= 0 ? (ns >> 1) : p1;     // ns < 0: the negative IS the positive (hs = false: V = the tail)
    } else {
      int32_t n0 = 0, n1 = 0;
      if (live) { p0 = raw[0]; p1 = raw[1]; p2 = raw[2]; n0 = raw[3]; n1 = raw[4]; }
      bad = bad3(N, p0, p1, p2) || n0 < 0 || n1 < 0 || n0 >= N || n1 >= N;
      hs = n0 != p0;
      c = hs ? n0 : n1;
    }
    if (bad) { p0 = p1 = p2 = c = 0; hs = false; }
    const int32_t idF = hs ? p1 : p0, idV = hs ? p0 : p1;
    const float sigma = hs ? -1.f : 1.f;
    // Staged rows (the all-to-all schedule) are addressed WITHOUT a branch: with one, the rows' requests each sat
    // behind their own branch and `s_waitcnt vmcnt(0)` -- one memory round trip per row instead of one per pair (found in
    // the ISA in round 3).  The peer-mapped experiment (PEER: the owner's shard read in place, an integer division per
    // row) is a separate instantiation.
    auto row_ptr = [&](int32_t id) -> const float* {
      if constexpr (SHARD && PEER) {
        if (id >= sg.R) {
          const int32_t o = id / sg.R - 1;
          return sg.peer[o & (kMaxPeers - 1)] + (int64_t)(id - sg.R * (o + 1)) * d;
        }
        return rows + (int64_t)id * d;
      } else if constexpr (SHARD) {
        const bool rem = id >= sg.R;
        const float* base = rem ? sg.staged : rows;
        return base + (int64_t)(rem ? id - sg.R : id) * d;
      } else {
        return rows + (int64_t)id * d;
      }
    };
    Row<VEC, NITER> xF, xV, xC, xR;
    load_row_clamped<VEC, LPT, NITER>(row_ptr(idF), k, nvec, sub, xF);
    load_row_clamped<VEC, LPT, NITER>(row_ptr(idV), k, nvec, sub, xV);
    load_row_clamped<VEC, LPT, NITER>(row_ptr(c), k, nvec, sub, xC);
    load_row_clamped<VEC, LPT, NITER>(row_ptr(p2), k, nvec, sub, xR);
    __builtin_amdgcn_sched_barrier(0);
    after_loads();                      // the next pairs' id words are requested behind this pair's rows
    __builtin_amdgcn_sched_barrier(0);
    const bool first_el = sub == 0;
    // forward: w = F r~ once, both scores against it
    float wre[NITER][VEC], wim[NITER][VEC];
    float accP = 0.f, accN = 0.f, qF = 0.f, qV = 0.f, qC = 0.f, qR = 0.f;
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      float tP = 0.f, tN = 0.f;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float a = xF.re[it][v], b = xF.im[it][v], cr = xR.re[it][v], di = xR.im[it][v];
        const float ds = sigma * di;
        float wr = a * cr - b * ds, wi = a * ds + b * cr;
        if (SPEC && it == 0 && v == 0 && first_el) { wr = a * cr; wi = b * di; }   // (X_0, X_k): two real dimensions
        wre[it][v] = wr; wim[it][v] = wi;
        const float tp = wr * xV.re[it][v] + wi * xV.im[it][v], tn = wr * xC.re[it][v] + wi * xC.im[it][v];
        if (SPEC && !(it == 0 && v == 0 && first_el)) { tP += 2.f * tp; tN += 2.f * tn; }
        else { tP += tp; tN += tn; }
      }
      accP += act[it] * tP; accN += act[it] * tN;
    }
    {
      // row_sumsq per vector, masked (a lane's clamped copies must not count)
      auto ssq = [&](const Row<VEC, NITER>& R) {
        float s = 0.f;
#pragma unroll
        for (int it = 0; it < NITER; ++it) {
          float t = 0.f;
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float q = R.re[it][v] * R.re[it][v] + R.im[it][v] * R.im[it][v];
            if (SPEC) t += (it == 0 && v == 0 && first_el) ? q : 2.f * q;
            else t += q;
          }
          s += act[it] * t;
        }
        return s;
      };
      qF = ssq(xF); qV = ssq(xV); qC = ssq(xC); qR = ssq(xR);
    }
    float ssF = group_sum_dpp<LPT>(qF), ssV = group_sum_dpp<LPT>(qV);
    float ssC = group_sum_dpp<LPT>(qC), ssR = group_sum_dpp<LPT>(qR);
    float srP = group_sum_dpp<LPT>(accP), srN = group_sum_dpp<LPT>(accN);
    if (SPEC) { ssF *= wscale; ssV *= wscale; ssC *= wscale; ssR *= wscale; srP *= wscale; srN *= wscale; }
    float invF, invV, invC, invR;
    const float scF = clip_scale(ssF, max_norm, invF), scV = clip_scale(ssV, max_norm, invV);
    const float scC = clip_scale(ssC, max_norm, invC), scR = clip_scale(ssR, max_norm, invR);
    const float sigP = sigmoidf_dev(srP * (hs ? scV : scF) * (hs ? scF : scV) * scR);     // s_raw * sc_h * sc_t * sc_r
    const float sigN = sigmoidf_dev(srN * (hs ? scC : scF) * (hs ? scF : scC) * scR);
    const float pre = sigP - sigN + margin;
    const bool on = live && !bad && (pre >= 0.f);  // MaximumGrad: x >= y
    if (live && sub == 0) loss[g] = bad ? __builtin_nanf("") : fmaxf(pre, 0.f);
    const bool same = c == idV;                    // the corrupted entity is the original: one merged row for V
    const int slot0 = g * 6;
    const int32_t tagF = hs ? tag[1] : tag[0], tagV = hs ? tag[0] : tag[1], tagR = tag[2], tagC = hs ? tag[3] : tag[4];
    // local direct (kSlotDirect): row + gradient, applied here; remote direct (SHARD, tag <= -3): the gradient row
    // itself into the send buffer
    const bool dirF = slot_item && live && tagF == kSlotDirect, dirV = slot_item && live && tagV == kSlotDirect;
    const bool dirR = slot_item && live && tagR == kSlotDirect, dirC = slot_item && live && !same && tagC == kSlotDirect;
    const bool sndF = SHARD && slot_item && live && tagF <= -3, sndV = SHARD && slot_item && live && tagV <= -3;
    const bool sndR = SHARD && slot_item && live && tagR <= -3, sndC = SHARD && slot_item && live && !same && tagC <= -3;
    // the relation row of a pair walked in relation order: summed over the run instead of stored per pair
    const bool accum = ORD && !dirR && !sndR;
    if (accum && live && cur_slot >= 0 && p2 != cur_rel) flush();
    if (live && sub == 0) {
      // (selects, not an array indexed by the column: a dynamically indexed array lives in scratch memory)
      const int32_t giF = (on && !dirF && !sndF) ? idF : -1, giV = (on && !dirV && !sndV) ? idV : -1;
      const int32_t giC = (on && !same && !dirC && !sndC) ? c : -1;
      grad_idx[slot0] = hs ? giV : giF;
      grad_idx[slot0 + 1] = hs ? giF : giV;
      grad_idx[slot0 + 2] = (on && !dirR && !sndR && !accum) ? p2 : -1;
      grad_idx[slot0 + 3] = hs ? giC : -1;
      grad_idx[slot0 + 4] = hs ? -1 : giC;
      grad_idx[slot0 + 5] = -1;
    }
    if (!on) {
      // a remote row's only gradient slot is inactive: the owner still receives a row for it -- zeros
      if (SHARD && (sndF || sndV || sndR || sndC)) {
        float z[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) z[v] = 0.f;
        auto zero_row = [&](bool doit, int32_t tg) {
          if (!doit) return;
          float* o = sg.gsum + (int64_t)(-3 - tg) * d;
#pragma unroll
          for (int it = 0; it < NITER; ++it) {
            const int j = sub + it * LPT;
            if (j < nvec) { store_vec<VEC>(o + j * VEC, z); store_vec<VEC>(o + k + j * VEC, z); }
          }
        };
        zero_row(sndF, tagF); zero_row(sndV, tagV); zero_row(sndR, tagR); zero_row(sndC, tagC);
      }
      return;
    }
    // coefficients through the clip (row_coef): g_x = alpha * G_raw + beta * x per side
    const float cp = sigP * (1.f - sigP), cn = -sigN * (1.f - sigN);
    auto coef = [&](float cf, float P, float inv, float s_raw, float& alpha, float& beta) {
      const float A = cf * P;
      const bool active = inv <= 1.0f / max_norm;
      alpha = neg_lr * (active ? max_norm * A * inv : A);
      beta = active ? neg_lr * (-max_norm * A * s_raw * inv * inv * inv) : 0.f;
    };
    float aFp, bFp, aFn, bFn, aRp, bRp, aRn, bRn, aV, bV, aC, bC;
    coef(cp, scV * scR, invF, srP, aFp, bFp);
    coef(cn, scC * scR, invF, srN, aFn, bFn);
    coef(cp, scF * scV, invR, srP, aRp, bRp);
    coef(cn, scF * scC, invR, srN, aRn, bRn);
    coef(cp, scF * scR, invV, srP, aV, bV);
    coef(cn, scF * scR, invC, srN, aC, bC);
    const float bF = bFp + bFn, bR = bRp + bRn;
    if (same) { aV += aC; bV += bC; }
    const int hsi = hs ? 1 : 0;
    auto out_ptr = [&](bool dir, bool snd, int32_t tg, int32_t id, int slot) -> float* {
      return dir ? table_rw + (int64_t)id * d : (SHARD && snd) ? sg.gsum + (int64_t)(-3 - tg) * d : grad_val + (int64_t)slot * d;
    };
    // rows V and C: alpha w + beta x
    {
      float* o = out_ptr(dirV, sndV, tagV, idV, slot0 + 1 - hsi);
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float ore[VEC], oim[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          ore[v] = aV * wre[it][v] + bV * xV.re[it][v];
          oim[v] = aV * wim[it][v] + bV * xV.im[it][v];
          // a direct row: x' = x + (-lr g), the same bits as the apply kernel's 0 + g, then x + sum
          if (dirV) { ore[v] += xV.re[it][v]; oim[v] += xV.im[it][v]; }
        }
        store_vec<VEC>(o + j * VEC, ore);
        store_vec<VEC>(o + k + j * VEC, oim);
      }
    }
    if (!same) {
      float* o = out_ptr(dirC, sndC, tagC, c, slot0 + 4 - hsi);
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float ore[VEC], oim[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          ore[v] = aC * wre[it][v] + bC * xC.re[it][v];
          oim[v] = aC * wim[it][v] + bC * xC.im[it][v];
          if (dirC) { ore[v] += xC.re[it][v]; oim[v] += xC.im[it][v]; }
        }
        store_vec<VEC>(o + j * VEC, ore);
        store_vec<VEC>(o + k + j * VEC, oim);
      }
    }
    // row F: conj(r~) (aFp V + aFn C) + bF F
    {
      float* o = out_ptr(dirF, sndF, tagF, idF, slot0 + hsi);
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float ore[VEC], oim[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float ur = aFp * xV.re[it][v] + aFn * xC.re[it][v], ui = aFp * xV.im[it][v] + aFn * xC.im[it][v];
          const float cr = xR.re[it][v], di = xR.im[it][v], ds = sigma * di;
          float gr = cr * ur + ds * ui, gim = cr * ui - ds * ur;
          if (SPEC && it == 0 && v == 0 && first_el) { gr = cr * ur; gim = di * ui; }
          ore[v] = gr + bF * xF.re[it][v];
          oim[v] = gim + bF * xF.im[it][v];
          if (dirF) { ore[v] += xF.re[it][v]; oim[v] += xF.im[it][v]; }
        }
        store_vec<VEC>(o + j * VEC, ore);
        store_vec<VEC>(o + k + j * VEC, oim);
      }
    }
    // row r: (Re, sigma Im) of conj(F) (aRp V + aRn C), + bR r
    {
      float* o = out_ptr(dirR, sndR, tagR, p2, slot0 + 2);
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float ore[VEC], oim[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          const float ur = aRp * xV.re[it][v] + aRn * xC.re[it][v], ui = aRp * xV.im[it][v] + aRn * xC.im[it][v];
          const float a = xF.re[it][v], b = xF.im[it][v];
          float gr = a * ur + b * ui, gim = sigma * (a * ui - b * ur);
          if (SPEC && it == 0 && v == 0 && first_el) { gr = a * ur; gim = b * ui; }
          ore[v] = gr + bR * xR.re[it][v];
          oim[v] = gim + bR * xR.im[it][v];
          if (dirR) { ore[v] += xR.re[it][v]; oim[v] += xR.im[it][v]; }
        }
        if (accum) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) { racc_re[it][v] += ore[v]; racc_im[it][v] += oim[v]; }
        } else {
          store_vec<VEC>(o + j * VEC, ore);
          store_vec<VEC>(o + k + j * VEC, oim);
        }
      }
      if (accum) { cur_rel = p2; cur_slot = slot0 + 2; }
    }
  };
  // Three stages deep, all but the rows on the scalar path: while pair i's rows are in flight, the ids of pair i + 1
  // (whose index arrived an iteration ago) and the relation-order entry of pair i + 2 are requested; per iteration a wave
  // waits for one memory round trip (the rows) -- at 65,536 pairs a wave walks 4-8 pairs and the chain, not bandwidth,
  // set the kernel's time.  Pairs past the run re-read pair 0 and are processed as not live.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int start = ORD ? wave_u * per : wave_u * GPW, stop = ORD ? min(start + per, nB) : nB;
  const int stride = ORD ? GPW : nwaves * GPW;
  auto pair_at = [&](int i) -> int32_t {              // the pair at position i of this wave's walk (i wave-uniform)
    if (i >= stop) return 0;
    return ORD ? kld(order, i) : i;
  };
  int32_t gC[GPW], gN[GPW], rawS[GPW][5], tagS[GPW][5];
#pragma unroll
  for (int j = 0; j < GPW; ++j) { gC[j] = pair_at(start + j); gN[j] = pair_at(start + stride + j); }
#pragma unroll
  for (int j = 0; j < GPW; ++j) load_ids(gC[j], rawS[j], tagS[j]);
  for (int base = start; base < stop; base += stride) {
    const bool live = base + grp < stop;
    const int g = pick(gC);
    int32_t raw[5], tag[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      int32_t rv[GPW], tv[GPW];
#pragma unroll
      for (int j = 0; j < GPW; ++j) { rv[j] = rawS[j][c]; tv[j] = tagS[j][c]; }
      raw[c] = pick(rv); tag[c] = pick(tv);
    }
    body(g, live, raw, tag, [&]() {
#pragma unroll
      for (int j = 0; j < GPW; ++j) {
        gC[j] = gN[j];
        load_ids(gC[j], rawS[j], tagS[j]);
        gN[j] = pair_at(base + 2 * stride + j);
      }
    });
  }
  if constexpr (ORD) {
    // What is still in the running sums when the runs end: the workgroup's lane groups hold consecutive runs of the relation
    // order, nearly always of ONE relation (3,640 pairs a relation at 65,536 pairs and 18 relations, 8-16 pairs a wave), so
    // their sums meet in LDS and leave as one gradient row per relation and workgroup -- the last group's slot takes it, the
    // others' slots stay empty -- instead of one per lane group: an eighth of the relation rows written here and read,
    // flagged and atomically added by the update kernel.  Fixed order (group index), so nothing about it depends on timing.
    constexpr int NG = (kBlock / kWave) * GPW, NF = NITER * VEC * LPT;
    __shared__ float part[NG][2 * NF];
    __shared__ int meta_rel[NG], meta_slot[NG];
    const int gi = (int)(threadIdx.x >> 6) * GPW + grp;
#pragma unroll
    for (int it = 0; it < NITER; ++it)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        part[gi][(it * VEC + v) * LPT + sub] = racc_re[it][v];
        part[gi][NF + (it * VEC + v) * LPT + sub] = racc_im[it][v];
      }
    if (sub == 0) { meta_rel[gi] = cur_rel; meta_slot[gi] = cur_slot; }
    __syncthreads();
    if (cur_slot >= 0) {
      bool leader = true;
      for (int g2 = gi + 1; g2 < NG; ++g2) leader = leader && !(meta_slot[g2] >= 0 && meta_rel[g2] == cur_rel);
      if (leader) {
#pragma unroll
        for (int it = 0; it < NITER; ++it)
#pragma unroll
          for (int v = 0; v < VEC; ++v) { racc_re[it][v] = 0.f; racc_im[it][v] = 0.f; }
        for (int g2 = 0; g2 <= gi; ++g2) {
          if (!(meta_slot[g2] >= 0 && meta_rel[g2] == cur_rel)) continue;
#pragma unroll
          for (int it = 0; it < NITER; ++it)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              racc_re[it][v] += part[g2][(it * VEC + v) * LPT + sub];
              racc_im[it][v] += part[g2][NF + (it * VEC + v) * LPT + sub];
            }
        }
        flush();
      }
    }
  }
}

// ---------------------------------------------------------------- logistic loss (--log_loss)
// holE.py:194-196 with labels +1 / -1 (holE.py:209, 215): per triple
//     loss = log(1 + exp(-y s)) + l2 * l2_loss(table),   d loss/d s = -y sigma(-y s).
// One group per triple; emits 3 IndexedSlices rows (h, t, r) already multiplied by -lr.  The dense
// L2 term is handled by the caller (a whole-table scale, see ge_complex_logloss_step).
// Two ways to name the M triples: (triples [M,3], labels [M]) -- the single-step entry point -- or, inside the
// native loop (negs != null), positives `triples` [B,3] followed by negs [K*B,3] with labels +1 / -1 by position.
// row_scale: the table holds rows / row_scale (the dense L2 decay of the step is kept as ONE scalar, see
// train_logloss_run); loaded rows are multiplied by it and `neg_lr_eff` already carries the 1 / new-scale.
template <int VEC, int LPT, int NITER>
__global__ __launch_bounds__(kBlock) void complex_logloss_grad_kernel(
    const float* __restrict__ rows, int64_t N, int d, const int32_t* __restrict__ triples,
    const float* __restrict__ labels, const int32_t* __restrict__ negs, int64_t B, int64_t M, float neg_lr_eff,
    float max_norm, float l2, float row_scale, const float* __restrict__ table_sumsq, float* __restrict__ loss,
    int32_t* __restrict__ grad_idx, float* __restrict__ grad_val) {
  constexpr int GPW = kWave / LPT;
  const int lane = threadIdx.x & (kWave - 1), sub = lane % LPT, grp = lane / LPT;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  // l2 * tf.nn.l2_loss(embeddings), same for every row of the loss vector (null: not asked for this step)
  const float reg = table_sumsq ? l2 * 0.5f * row_scale * row_scale * table_sumsq[0] : 0.f;
  for (int64_t base = wave * GPW; base < M; base += nwaves * GPW) {
    const int64_t g = base + grp;
    const bool live = g < M;
    int32_t p[3] = {0, 0, 0};
    float y = 1.f;
    if (live) {
      const int32_t* t3 = (negs && g >= B) ? negs + 3 * (g - B) : triples + 3 * g;
      p[0] = t3[0]; p[1] = t3[1]; p[2] = t3[2];
      y = negs ? (g < B ? 1.f : -1.f) : labels[g];
    }
    const bool bad = bad3(N, p[0], p[1], p[2]);
    if (bad) { p[0] = p[1] = p[2] = 0; }
    Row<VEC, NITER> x[3];
#pragma unroll
    for (int X = 0; X < 3; ++X) {
      load_row<VEC, LPT, NITER>(rows, p[X], d, k, nvec, sub, x[X]);
#pragma unroll
      for (int it = 0; it < NITER; ++it)
#pragma unroll
        for (int v = 0; v < VEC; ++v) { x[X].re[it][v] *= row_scale; x[X].im[it][v] *= row_scale; }
    }
    const SideFwd f = side_forward<false, VEC, LPT, NITER>(x[0], x[1], x[2], max_norm, false, 1.f);
    const float z = -y * f.s;                                            // holE.py:195
    const float softplus = z > 0.f ? z + log1pf(expf(-z)) : log1pf(expf(z));
    const float coef = -y * sigmoidf_dev(z);                             // d/ds log(1+exp(-y s))
    if (live && sub == 0) loss[g] = bad ? __builtin_nanf("") : softplus + reg;
#pragma unroll
    for (int X = 0; X < 3; ++X) {
      const int64_t slot = g * 3 + X;
      if (live && sub == 0) grad_idx[slot] = bad ? -1 : p[X];
      if (!live || bad) continue;
      const RowCoef kc = row_coef(coef, f, X, max_norm, neg_lr_eff);
      float* gp = grad_val + slot * d;
#pragma unroll
      for (int it = 0; it < NITER; ++it) {
        const int j = sub + it * LPT;
        if (j >= nvec) continue;
        float re_[VEC], im_[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          float gre, gim;
          graw<false, VEC, NITER>(X, x[0], x[1], x[2], it, v, false, gre, gim);
          re_[v] = kc.alpha * gre + kc.beta * x[X].re[it][v];
          im_[v] = kc.alpha * gim + kc.beta * x[X].im[it][v];
        }
        store_vec<VEC>(gp + j * VEC, re_);
        store_vec<VEC>(gp + k + j * VEC, im_);
      }
    }
  }
}

// sum of squares of the whole table (tf.nn.l2_loss * 2) -> out[0]; out must be zeroed before.
__global__ __launch_bounds__(kBlock) void table_sumsq_kernel(const float* __restrict__ t, int64_t n,
                                                              float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = t[i];
    acc += v * v;
  }
  acc = group_sum<kWave>(acc);
  __shared__ float part[kBlock / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) s += part[w];
    atomic_add_f32(out, s);
  }
}

// table *= factor : the dense part of the SGD step, (1 - lr * M * l2)
__global__ __launch_bounds__(kBlock) void table_scale_kernel(float* __restrict__ t, int64_t n, float factor) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    t[i] *= factor;
}

// ---------------------------------------------------------------- dispatch
struct Shape { int vec, lpt, niter; };

static bool pick_shape(int d, const void* base, int max_niter, Shape& s) {
  if (d <= 0 || (d & 1)) return false;
  const int k = d / 2;
  int vec = (k % 4 == 0) ? 4 : (k % 2 == 0) ? 2 : 1;
  const uintptr_t a = reinterpret_cast<uintptr_t>(base);
  while (vec > 1 && (a % (vec * 4)) != 0) vec >>= 1;
  const int nvec = k / vec;
  int lpt = 16;
  while (lpt < 64 && lpt < nvec) lpt <<= 1;
  int niter = (nvec + lpt - 1) / lpt;
  if (niter > 2) niter = 4;
  if (niter > max_niter) return false;
  s = {vec, lpt, niter};
  return true;
}

#define GE_DISPATCH_SHAPE(S, MAXN, CALL)                                               \
  do {                                                                                 \
    const int key_ = (S).vec * 1000 + (S).lpt * 10 + (S).niter;                        \
    switch (key_) {                                                                    \
      case 1161: { CALL(1, 16, 1); } break;                                            \
      case 1321: { CALL(1, 32, 1); } break;                                            \
      case 1641: { CALL(1, 64, 1); } break;                                            \
      case 1642: { CALL(1, 64, 2); } break;                                            \
      case 2161: { CALL(2, 16, 1); } break;                                            \
      case 2321: { CALL(2, 32, 1); } break;                                            \
      case 2641: { CALL(2, 64, 1); } break;                                            \
      case 2642: { CALL(2, 64, 2); } break;                                            \
      case 4161: { CALL(4, 16, 1); } break;                                            \
      case 4162: { CALL(4, 16, 2); } break;                                            \
      case 4322: { CALL(4, 32, 2); } break;                                            \
      case 4321: { CALL(4, 32, 1); } break;                                            \
      case 4641: { CALL(4, 64, 1); } break;                                            \
      case 4642: { CALL(4, 64, 2); } break;                                            \
      default: return GE_ENOTSUP;                                                      \
    }                                                                                  \
  } while (0)

// SP (the SPEC template argument) is bound per branch; CALL mentions it
#define GE_DISPATCH_SPEC(FLAG, S, CALL)                                                \
  do {                                                                                 \
    if (FLAG) { constexpr bool SP = true; GE_DISPATCH_SHAPE(S, 2, CALL); }             \
    else { constexpr bool SP = false; GE_DISPATCH_SHAPE(S, 2, CALL); }                 \
  } while (0)

int complex_score_launch(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                         float max_norm, int apply_sigmoid, float* out, hipStream_t st, int spectral, float label,
                         float l2, const float* table_sumsq, int64_t ld) {
  Shape s;
  if (ld <= 0) ld = d;
  if (!pick_shape(d, table, 2, s)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  if (ld < d || (s.vec > 1 && ld % s.vec != 0)) return GE_EINVAL;
  if (B == 0) return 0;
  const int gpb = (kBlock / kWave) * (kWave / s.lpt);
  const int grid = grid_for(B, gpb);
#define CALL(V, L, NI) \
  hipLaunchKernelGGL((complex_score_kernel<SP, V, L, NI>), dim3(grid), dim3(kBlock), 0, st, table, N, d, triples, B, max_norm, apply_sigmoid, out, label, l2, table_sumsq, ld)
  GE_DISPATCH_SPEC(spectral, s, CALL);
#undef CALL
  return launch_status();
}

int complex_hinge_loss_launch(const float* table, int64_t N, int32_t d, const int32_t* pos,
                              const int32_t* neg, int64_t B, float margin, float max_norm, float* loss,
                              float* sig_out, hipStream_t st, int spectral) {
  Shape s;
  if (!pick_shape(d, table, 2, s)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  if (B == 0) return 0;
  const int gpb = (kBlock / kWave) * (kWave / s.lpt);
  const int grid = grid_for(B, gpb);
#define CALL(V, L, NI) \
  hipLaunchKernelGGL((complex_hinge_loss_kernel<SP, V, L, NI>), dim3(grid), dim3(kBlock), 0, st, table, N, d, pos, neg, B, margin, max_norm, loss, sig_out)
  GE_DISPATCH_SPEC(spectral, s, CALL);
#undef CALL
  return launch_status();
}

int complex_hinge_grad_launch(const float* rows, int64_t N, int32_t d, const int32_t* pos,
                              const int32_t* neg, int64_t B, float margin, float lr, float max_norm,
                              float* loss, int32_t* grad_idx, float* grad_val, hipStream_t st,
                              hipEvent_t ev_start, hipEvent_t ev_stop, const int32_t* slot_item, float* table_rw,
                              int spectral, const int32_t* order) {
  Shape s;
  if (!pick_shape(d, rows, 2, s)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  // the gradient rows are written with the same vector width: grad_val must be as aligned as rows
  while (s.vec > 1 && (reinterpret_cast<uintptr_t>(grad_val) % (s.vec * 4)) != 0) return GE_EINVAL;
  if ((slot_item != nullptr) != (table_rw != nullptr)) return GE_EINVAL;
  if (B == 0) return 0;
  const int gpb = (kBlock / kWave) * (kWave / s.lpt);
  const int grid = grid_for(B, gpb);
#define CALL(V, L, NI) \
  hipExtLaunchKernelGGL((complex_hinge_grad_kernel<SP, V, L, NI>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val, slot_item, table_rw)
#define CALLO(V, L, NI) \
  hipExtLaunchKernelGGL((complex_hinge_grad_plan_kernel<SP, false, true, V, L, NI>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val, slot_item, table_rw, ShardGrad{}, order)
#define CALLP(V, L, NI) \
  hipExtLaunchKernelGGL((complex_hinge_grad_plan_kernel<SP, false, false, V, L, NI>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val, slot_item, table_rw, ShardGrad{}, nullptr)
  // The training loops' steps (slot_item given: negatives from the sampler, one column replaced) run the four-row kernel at
  // every batch size -- at 4096 pairs 7.85 us against 8.5-8.8 for the six-row kernel, alternated on one box (round 4) --
  // the single-step API, whose negatives are arbitrary triples, keeps the six-row one.  (One pair a wave -- 2 complex
  // elements a lane, twice the waves -- was slower at 4096 pairs: 8.56 vs 7.82 us.)
  if (order) { GE_DISPATCH_SPEC(spectral, s, CALLO); }
  else if (slot_item) { GE_DISPATCH_SPEC(spectral, s, CALLP); }
  else { GE_DISPATCH_SPEC(spectral, s, CALL); }
#undef CALLP
#undef CALLO
#undef CALL
  return launch_status();
}

// the row-sharded step's form: rows from the shard (in place) and from the staging buffer of fetched rows
int shard_hinge_grad_launch(float* shard, int32_t d, const float* staged, const int32_t* pos_src, const int32_t* neg_src,
                            const int32_t* slot_item, int32_t R, int64_t B, float margin, float lr, float max_norm,
                            float* loss, int32_t* grad_idx, float* grad_val, float* gsum, int spectral, hipStream_t st,
                            hipEvent_t ev_start, hipEvent_t ev_stop, const int32_t* order, const float* const* peers, int n_peers) {
  Shape s;
  if (!pick_shape(d, shard, 2, s)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  for (const void* q : {(const void*)grad_val, (const void*)staged, (const void*)gsum})
    if (q && (reinterpret_cast<uintptr_t>(q) % (s.vec * 4)) != 0) return GE_EINVAL;
  if (B == 0) return 0;
  const int gpb = (kBlock / kWave) * (kWave / s.lpt);
  const int grid = grid_for(B, gpb);
  ShardGrad sg{staged, R, pos_src, neg_src, gsum, {}};
  if (peers) {
    if (staged || n_peers > kMaxPeers) return GE_EINVAL;
    for (int i = 0; i < kMaxPeers; ++i) sg.peer[i] = i < n_peers ? peers[i] : nullptr;
  }
#define CALL(V, L, NI) \
  hipExtLaunchKernelGGL((complex_hinge_grad_plan_kernel<SP, true, ORDF, V, L, NI, PEERF>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, shard, (int64_t)R, d, nullptr, nullptr, B, margin, lr, max_norm, loss, grad_idx, grad_val, slot_item, shard, sg, order)
  if (peers) {
    constexpr bool PEERF = true;
    if (order) { constexpr bool ORDF = true; GE_DISPATCH_SPEC(spectral, s, CALL); }
    else { constexpr bool ORDF = false; GE_DISPATCH_SPEC(spectral, s, CALL); }
  } else {
    constexpr bool PEERF = false;
    if (order) { constexpr bool ORDF = true; GE_DISPATCH_SPEC(spectral, s, CALL); }
    else { constexpr bool ORDF = false; GE_DISPATCH_SPEC(spectral, s, CALL); }
  }
#undef CALL
  return launch_status();
}

int complex_logloss_grad_launch(const float* rows, int64_t N, int32_t d, const int32_t* triples,
                                const float* labels, int64_t M, float lr, float max_norm, float l2,
                                const float* table_sumsq, float* loss, int32_t* grad_idx, float* grad_val,
                                hipStream_t st, const int32_t* negs, int64_t B, float row_scale, float neg_lr_eff,
                                hipEvent_t ev_start, hipEvent_t ev_stop) {
  Shape s;
  if (!pick_shape(d, rows, 2, s)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  if ((reinterpret_cast<uintptr_t>(grad_val) % (s.vec * 4)) != 0) return GE_EINVAL;
  if (M == 0) return 0;
  if (!negs) neg_lr_eff = -lr;
  const int gpb = (kBlock / kWave) * (kWave / s.lpt);
  const int grid = grid_for(M, gpb);
#define CALL(V, L, NI) \
  hipExtLaunchKernelGGL((complex_logloss_grad_kernel<V, L, NI>), dim3(grid), dim3(kBlock), 0, st, ev_start, ev_stop, 0, rows, N, d, triples, labels, negs, B, M, neg_lr_eff, max_norm, l2, row_scale, table_sumsq, loss, grad_idx, grad_val)
  GE_DISPATCH_SHAPE(s, 2, CALL);
#undef CALL
  return launch_status();
}

int table_sumsq_launch(const float* table, int64_t n, float* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(table_sumsq_kernel, dim3(grid_for(n, kBlock * 8)), dim3(kBlock), 0, st, table, n, out);
  return launch_status();
}

int table_scale_launch(float* table, int64_t n, float factor, hipStream_t st) {
  hipLaunchKernelGGL(table_scale_kernel, dim3(grid_for(n, kBlock * 8)), dim3(kBlock), 0, st, table, n, factor);
  return launch_status();
}

int complex_max_dim() { return 4 * 64 * 2 * 2; }  // VEC*LPT*NITER complex dims * 2

}  // namespace ge
