// ge_complex_dev.h -- device helpers of the ComplEx-shaped kernels (ge_complex.hip):
// lane-sliced rows, clip/score forward of one side, closed-form row-gradient coefficients.
//
// SPEC = true is the same arithmetic on a SPECTRAL HolE table (ge_hole_to_spectral): a row holds the
// half spectrum of a real d-vector x, packed in d floats as [Re X_0 .. Re X_{k-1} | Re X_k, Im X_1 ..
// Im X_{k-1}], k = d/2 (X_0 and the Nyquist bin X_k are real; X_k sits in the unused Im X_0 slot).
// With Hermitian weights w_0 = w_k = 1, w_f = 2:
//   |x|^2        = (1/d) sum_f w_f |X_f|^2                                   (Parseval)
//   r . (h * t)  = (1/d) sum_f w_f Re(H_f R_f conj(T_f))                      (README.md:42, correlation theorem)
// i.e. HolE IS the ComplEx trilinear form on the half spectrum; the DFT is linear, so clip (a per-row
// scalar) and SGD commute with it, and the DFT of the real-domain row gradient is the UNWEIGHTED
// complex gradient of that form (the factor d/w_f of the change of variables cancels the weight).
// Lane sub = 0, element 0 holds the packed pair (X_0, X_k): two independent real dimensions.
#pragma once
#include "ge_common.h"

namespace ge {

template <int VEC, int NITER>
struct Row {
  float re[NITER][VEC];
  float im[NITER][VEC];
};

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ void load_row_at(const float* __restrict__ p, int k, int nvec, int sub, Row<VEC, NITER>& R);

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ void load_row(const float* __restrict__ rows, int32_t id, int d, int k,
                                         int nvec, int sub, Row<VEC, NITER>& R) {
  load_row_at<VEC, LPT, NITER>(rows + (int64_t)id * d, k, nvec, sub, R);
}

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ void load_row_at(const float* __restrict__ p, int k, int nvec, int sub, Row<VEC, NITER>& R) {
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int j = sub + it * LPT;
    if (j < nvec) {
      load_vec<VEC>(p + j * VEC, R.re[it]);
      load_vec<VEC>(p + k + j * VEC, R.im[it]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) { R.re[it][v] = 0.f; R.im[it][v] = 0.f; }
    }
  }
}

// `first` = this lane holds element 0 of the row (sub == 0): only meaningful for SPEC
template <bool SPEC, int VEC, int NITER>
__device__ __forceinline__ float row_sumsq(const Row<VEC, NITER>& R, bool first) {
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float q = R.re[it][v] * R.re[it][v] + R.im[it][v] * R.im[it][v];
      if (SPEC) ss += (it == 0 && v == 0 && first) ? q : 2.f * q;
      else ss += q;
    }
  return ss;
}

// lane-partial of sum_k Re(h_k r_k conj(t_k)) = a(ce+df) + b(cf-de)   (holE.py:191-192)
template <bool SPEC, int VEC, int NITER>
__device__ __forceinline__ float raw_score(const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                           const Row<VEC, NITER>& r, bool first) {
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float a = h.re[it][v], b = h.im[it][v], e = t.re[it][v], f = t.im[it][v];
      const float c = r.re[it][v], dd = r.im[it][v];
      if (SPEC) s += (it == 0 && v == 0 && first) ? (a * c * e + b * dd * f) : 2.f * (a * (c * e + dd * f) + b * (c * f - dd * e));
      else s += a * (c * e + dd * f) + b * (c * f - dd * e);
    }
  return s;
}

struct SideFwd {
  float s_raw;             // score of the un-clipped rows
  float sc[3];             // clip scales h,t,r
  float inv[3];            // rsqrt(sum x^2) h,t,r
  float sig;               // sigma(s)
  float s;                 // clipped score
};

// wscale = 1 (ComplEx) or 1/d (SPEC: the Parseval / correlation-theorem factor)
template <bool SPEC, int VEC, int LPT, int NITER>
__device__ __forceinline__ SideFwd side_forward(const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                                const Row<VEC, NITER>& r, float max_norm, bool first,
                                                float wscale) {
  SideFwd o;
  float ssh = group_sum<LPT>(row_sumsq<SPEC>(h, first));
  float sst = group_sum<LPT>(row_sumsq<SPEC>(t, first));
  float ssr = group_sum<LPT>(row_sumsq<SPEC>(r, first));
  o.s_raw = group_sum<LPT>(raw_score<SPEC>(h, t, r, first));
  if (SPEC) { ssh *= wscale; sst *= wscale; ssr *= wscale; o.s_raw *= wscale; }
  o.sc[0] = clip_scale(ssh, max_norm, o.inv[0]);
  o.sc[1] = clip_scale(sst, max_norm, o.inv[1]);
  o.sc[2] = clip_scale(ssr, max_norm, o.inv[2]);
  o.s = o.s_raw * o.sc[0] * o.sc[1] * o.sc[2];
  o.sig = sigmoidf_dev(o.s);
  return o;
}

// The row-sharded step (ge_shard.hip, sharded.py): a pair's rows come from two stores -- ids below R are rows of
// this rank's shard (read in place), R + u is row u of the staging buffer that holds the rows fetched from
// the other owners -- and a gradient row with a tag <= -3 is the only one of its (remote) row in this step: it
// is stored straight into row -3 - tag of the send buffer `gsum` instead of being queued for the reduction.
// Peer-mapped variant (an experiment): staged == null and the other owners' shards are mapped
// into this process (IPC); a source >= R is then the VIRTUAL row R (1 + owner) + local row and is read from
// peer[owner] directly.  Gradient sums still travel through gsum.
constexpr int kMaxPeers = 8;
struct ShardGrad {
  const float* staged;
  int32_t R;
  const int32_t* pos_src;   // [B][3]
  const int32_t* neg_src;   // [B]
  float* gsum;
  const float* peer[kMaxPeers];
};

__device__ __forceinline__ bool bad3(int64_t N, int32_t a, int32_t b, int32_t c) {
  return a < 0 || b < 0 || c < 0 || a >= N || b >= N || c >= N;
}

struct RowCoef { float alpha, beta; };

__device__ __forceinline__ RowCoef row_coef(float coef, const SideFwd& f, int X, float max_norm,
                                            float neg_lr) {
  const float P = (X == 0 ? f.sc[1] * f.sc[2] : X == 1 ? f.sc[0] * f.sc[2] : f.sc[0] * f.sc[1]);
  const float A = coef * P;
  const float inv = f.inv[X];
  const bool active = inv <= 1.0f / max_norm;
  RowCoef c;
  c.alpha = neg_lr * (active ? max_norm * A * inv : A);
  c.beta = active ? neg_lr * (-max_norm * A * f.s_raw * inv * inv * inv) : 0.f;
  return c;
}

// raw bilinear gradients of s_raw wrt X for one lane slice (SPEC: the DFT of the real-domain gradient;
// the packed element 0 is two real dimensions, each the plain product of the other two)
template <bool SPEC, int VEC, int NITER>
__device__ __forceinline__ void graw(int X, const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                     const Row<VEC, NITER>& r, int it, int v, bool first, float& gre, float& gim) {
  const float a = h.re[it][v], b = h.im[it][v], e = t.re[it][v], f = t.im[it][v];
  const float c = r.re[it][v], dd = r.im[it][v];
  if (SPEC && it == 0 && v == 0 && first) {
    if (X == 0) { gre = c * e; gim = dd * f; }
    else if (X == 1) { gre = a * c; gim = b * dd; }
    else { gre = a * e; gim = b * f; }
    return;
  }
  if (X == 0) { gre = c * e + dd * f; gim = c * f - dd * e; }        // d/dh
  else if (X == 1) { gre = a * c - b * dd; gim = a * dd + b * c; }   // d/dt
  else { gre = a * e + b * f; gim = a * f - b * e; }                 // d/dr
}

}  // namespace ge
