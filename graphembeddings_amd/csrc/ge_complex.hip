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
// The same kernel for the steps that come with a plan beyond the direct tags: SHARD (two row stores, the row-sharded
// step) and / or ORD (pairs walked in relation order, large batches).  The one-tile training step and the single-step
// API keep the kernel above unchanged: its code is tuned to the 4096-pair step, where every instruction between
// the id loads and the row loads shows.
// d(sum_i L_i)/d(raw rows), pre-multiplied by -lr, as IndexedSlices (6 slots per pair:
// h+, t+, r+, h-, t-, r-).  Through the clip (MinimumGrad routes to the rsqrt branch iff
// rsqrt(ss) <= 1/c):  gx = c*(gy*inv - x*(gy.x)*inv^3)  with  gy.x = coef*P_X*s_raw.
// Written per row X as  gx = alpha_X * Graw_X + beta_X * x_X.
template <bool SPEC, bool SHARD, bool ORD, int VEC, int LPT, int NITER, bool PEER = false>
__global__ __launch_bounds__(kBlock) void complex_hinge_grad_plan_kernel(
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
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int k = d >> 1, nvec = k / VEC;
  const float wscale = 1.0f / (float)d;   // SPEC only: Parseval / correlation-theorem factor
  const float neg_lr = -lr;
  // ORD: this wave's contiguous run [first, last) of the relation order
  const int64_t per = ORD ? ((B + nwaves - 1) / nwaves + GPW - 1) / GPW * GPW : 0;
  const int64_t first = wave * per, last = first + per < B ? first + per : B;
  float racc_re[NITER][VEC], racc_im[NITER][VEC];   // running relation-row gradient of this lane group's run
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) { racc_re[it][v] = 0.f; racc_im[it][v] = 0.f; }
  int32_t cur_rel = -1;
  int64_t cur_slot = -1;                              // >= 0: the slot the running sum will be written to
  auto flush = [&]() {
    float* o = grad_val + cur_slot * d;
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
  // the id words of pair g (6: pos + neg triples; SHARD: 3 row sources + the corrupted entity's) and its six tags
  auto load_ids = [&](int64_t g, int32_t (&raw)[6], int32_t (&tag)[6]) {
    if (SHARD) {
#pragma unroll
      for (int c = 0; c < 3; ++c) raw[c] = sg.pos_src[3 * g + c];
      raw[3] = sg.neg_src[g]; raw[4] = 0; raw[5] = 0;
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { raw[c] = pos[3 * g + c]; raw[3 + c] = neg[3 * g + c]; }
    }
    if (slot_item) {       // one test around all six: per-element tests make the compiler wait after every load
#pragma unroll
      for (int c = 0; c < 6; ++c) tag[c] = slot_item[g * 6 + c];
    } else {
#pragma unroll
      for (int c = 0; c < 6; ++c) tag[c] = 0;
    }
  };
  auto body = [&](const int64_t g, const bool live, const int32_t (&raw)[6], const int32_t (&tag)[6]) {
    int32_t p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    bool bad;
    if (SHARD) {
      int32_t ns = -1;
      if (live) { p[0] = raw[0]; p[1] = raw[1]; p[2] = raw[2]; ns = raw[3]; }
      bad = p[0] < 0 || p[1] < 0 || p[2] < 0;
      n[0] = (ns >= 0 && (ns & 1) == 0) ? (ns >> 1) : p[0];
      n[1] = (ns >= 0 && (ns & 1) == 1) ? (ns >> 1) : p[1];
      n[2] = p[2];
    } else {
      if (live) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { p[c] = raw[c]; n[c] = raw[3 + c]; }
      }
      bad = bad3(N, p[0], p[1], p[2]) || bad3(N, n[0], n[1], n[2]);
    }
    if (bad) { p[0] = p[1] = p[2] = n[0] = n[1] = n[2] = 0; }
    // Staged rows (the all-to-all schedule) are addressed WITHOUT a branch: with one, the six rows' requests each sat
    // behind their own branch and `s_waitcnt vmcnt(0)` -- six memory round trips a pair instead of one, 104 us instead of
    // 87 for the same step (found in the ISA).  The peer-mapped experiment (PEER: the owner's shard read in place, an
    // integer division per row) is a separate instantiation.
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
    Row<VEC, NITER> xp[3], xn[3];
#pragma unroll
    for (int X = 0; X < 3; ++X) load_row_at<VEC, LPT, NITER>(row_ptr(p[X]), k, nvec, sub, xp[X]);
#pragma unroll
    for (int X = 0; X < 3; ++X) load_row_at<VEC, LPT, NITER>(row_ptr(n[X]), k, nvec, sub, xn[X]);
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
      // local direct: row + gradient; remote direct (SHARD, tag <= -3): the gradient row itself into the send buffer
      const bool dirP = slot_item && live && tag[X] == kSlotDirect;
      const bool dirN = slot_item && live && !same && tag[3 + X] == kSlotDirect;
      const bool sndP = SHARD && slot_item && live && tag[X] <= -3;
      const bool sndN = SHARD && slot_item && live && !same && tag[3 + X] <= -3;
      // the relation row of a pair walked in relation order: summed over the run instead of stored per pair
      const bool accum = ORD && X == 2 && same && !dirP && !sndP;
      if (accum && live && cur_slot >= 0 && p[2] != cur_rel) flush();
      if (live && sub == 0) {
        grad_idx[rowP] = (on && !dirP && !sndP && !accum) ? p[X] : -1;
        grad_idx[rowN] = (on && !same && !dirN && !sndN) ? n[X] : -1;
      }
      if (!on) {
        // a remote row's only gradient slot is inactive: the owner still receives a row for it -- zeros
        if (SHARD && (sndP || sndN)) {
          float z[VEC];
#pragma unroll
          for (int v = 0; v < VEC; ++v) z[v] = 0.f;
#pragma unroll
          for (int it = 0; it < NITER; ++it) {
            const int j = sub + it * LPT;
            if (j >= nvec) continue;
            if (sndP) { float* o = sg.gsum + (int64_t)(-3 - tag[X]) * d; store_vec<VEC>(o + j * VEC, z); store_vec<VEC>(o + k + j * VEC, z); }
            if (sndN) { float* o = sg.gsum + (int64_t)(-3 - tag[3 + X]) * d; store_vec<VEC>(o + j * VEC, z); store_vec<VEC>(o + k + j * VEC, z); }
          }
        }
        continue;
      }
      const RowCoef kp = row_coef(cp, fp, X, max_norm, neg_lr);
      const RowCoef kn = row_coef(cn, fn, X, max_norm, neg_lr);
      float* gp = dirP ? table_rw + (int64_t)p[X] * d : sndP ? sg.gsum + (int64_t)(-3 - tag[X]) * d : grad_val + rowP * d;
      float* gn = dirN ? table_rw + (int64_t)n[X] * d : sndN ? sg.gsum + (int64_t)(-3 - tag[3 + X]) * d : grad_val + rowN * d;
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
          if (accum) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) { racc_re[it][v] += pre_[v]; racc_im[it][v] += pim_[v]; }
            continue;
          }
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
      if (accum) { cur_rel = p[2]; cur_slot = rowP; }   // (reached only for hinge-active pairs)
    }
  };
  if constexpr (ORD) {
    // The run's pair indices come with ONE load per 64 pairs (lane l holds order[first + l]) and every pair's id words
    // and tags are requested an iteration ahead, while the current pair's rows are in flight: per iteration a wave then
    // waits for one memory round trip (the rows) instead of three (order -> ids -> rows) -- at 65,536 pairs a wave walks
    // 4-8 pairs and the chain, not bandwidth, set the kernel's time.  All loads are unconditional (a pair past the run
    // re-reads pair 0 and is processed as not live).
    int32_t ord_reg = order[(first + lane < B) ? first + lane : B - 1];
    int32_t raw[6], tag[6];
    bool live = first + grp < last;
    int64_t g = live ? (int64_t)__shfl(ord_reg, grp, kWave) : 0;
    load_ids(g, raw, tag);
    for (int64_t base = first; base < last; base += GPW) {
      const int64_t nb = base + GPW;
      const int off = (int)((nb - first) & 63);
      if (off == 0 && nb < last) ord_reg = order[(nb + lane < B) ? nb + lane : B - 1];
      const int32_t gs = __shfl(ord_reg, off + grp, kWave);
      const bool live_n = nb + grp < last;
      const int64_t gn = live_n ? (int64_t)gs : 0;
      int32_t rawn[6], tagn[6];
      load_ids(gn, rawn, tagn);
      body(g, live, raw, tag);
      g = gn; live = live_n;
#pragma unroll
      for (int c = 0; c < 6; ++c) { raw[c] = rawn[c]; tag[c] = tagn[c]; }
    }
    if (cur_slot >= 0) flush();
  } else {
    for (int64_t base = wave * GPW; base < B; base += nwaves * GPW) {
      const int64_t g = base + grp;
      const bool live = g < B;
      int32_t raw[6] = {0, 0, 0, 0, 0, 0}, tag[6] = {0, 0, 0, 0, 0, 0};
      if (SHARD) raw[3] = -1;
      if (live) load_ids(g, raw, tag);     // ids and tags requested together (not after the forward)
      body(g, live, raw, tag);
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
  if (order) { GE_DISPATCH_SPEC(spectral, s, CALLO); }
  else { GE_DISPATCH_SPEC(spectral, s, CALL); }
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
