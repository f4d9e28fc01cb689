// ge_complex_dev.h -- device helpers shared by the ComplEx kernels (ge_complex.hip, ge_fused.hip):
// lane-sliced rows, clip/score forward of one side, closed-form row-gradient coefficients.
#pragma once
#include "ge_common.h"

namespace ge {

template <int VEC, int NITER>
struct Row {
  float re[NITER][VEC];
  float im[NITER][VEC];
};

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ void load_row(const float* __restrict__ rows, int32_t id, int d, int k,
                                         int nvec, int sub, Row<VEC, NITER>& R) {
  const float* p = rows + (int64_t)id * d;
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

template <int VEC, int NITER>
__device__ __forceinline__ float row_sumsq(const Row<VEC, NITER>& R) {
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) ss += R.re[it][v] * R.re[it][v] + R.im[it][v] * R.im[it][v];
  return ss;
}

// lane-partial of sum_k Re(h_k r_k conj(t_k)) = a(ce+df) + b(cf-de)   (holE.py:191-192)
template <int VEC, int NITER>
__device__ __forceinline__ float raw_score(const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                           const Row<VEC, NITER>& r) {
  float s = 0.f;
#pragma unroll
  for (int it = 0; it < NITER; ++it)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float a = h.re[it][v], b = h.im[it][v], e = t.re[it][v], f = t.im[it][v];
      const float c = r.re[it][v], dd = r.im[it][v];
      s += a * (c * e + dd * f) + b * (c * f - dd * e);
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

template <int VEC, int LPT, int NITER>
__device__ __forceinline__ SideFwd side_forward(const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                                const Row<VEC, NITER>& r, float max_norm) {
  SideFwd o;
  const float ssh = group_sum<LPT>(row_sumsq(h));
  const float sst = group_sum<LPT>(row_sumsq(t));
  const float ssr = group_sum<LPT>(row_sumsq(r));
  o.s_raw = group_sum<LPT>(raw_score(h, t, r));
  o.sc[0] = clip_scale(ssh, max_norm, o.inv[0]);
  o.sc[1] = clip_scale(sst, max_norm, o.inv[1]);
  o.sc[2] = clip_scale(ssr, max_norm, o.inv[2]);
  o.s = o.s_raw * o.sc[0] * o.sc[1] * o.sc[2];
  o.sig = sigmoidf_dev(o.s);
  return o;
}

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

// raw bilinear gradients of s_raw wrt X for one lane slice
template <int VEC, int NITER>
__device__ __forceinline__ void graw(int X, const Row<VEC, NITER>& h, const Row<VEC, NITER>& t,
                                     const Row<VEC, NITER>& r, int it, int v, float& gre, float& gim) {
  const float a = h.re[it][v], b = h.im[it][v], e = t.re[it][v], f = t.im[it][v];
  const float c = r.re[it][v], dd = r.im[it][v];
  if (X == 0) { gre = c * e + dd * f; gim = c * f - dd * e; }        // d/dh
  else if (X == 1) { gre = a * c - b * dd; gim = a * dd + b * c; }   // d/dt
  else { gre = a * e + b * f; gim = a * f - b * e; }                 // d/dr
}

}  // namespace ge
