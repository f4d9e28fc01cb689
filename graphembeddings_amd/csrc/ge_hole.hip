// ge_hole.hip -- HolE (holographic embeddings) score / hinge / row gradients.
//
// Score of README.md:42:  E(h,t,r) = sigmoid( r . (h star t) ),
//   (h star t)_k = sum_i h_i t_{(i+k) mod d}  ( = ifft(conj(fft(h)) fft(t)) ).
// The current holE.py computes ComplEx (holE.py:191-192); its circular_correlation helper
// (holE.py:175-176) is dead code, so the README formula is the contract.  Rows are clipped as in
// get_embedding (holE.py:162) and the score is squashed by the same sigmoid (holE.py:198).
//
// Kernel design (CDNA4): one 64-lane wavefront per triple.  Rows are staged in LDS; the "b" operand
// of a correlation is stored doubled (b[j] = x[j mod d]) so no modulo is needed in the inner loop.
// Lane l owns the 4 consecutive lags k = 256*c + 4*l .. +3 and walks i in steps of 4: per step one
// broadcast ds_read_b128 of a[i..i+3] and two conflict-free ds_read_b128 of b[i+k .. i+k+7] feed
// 16 FMAs (3 LDS instructions per 16 FMAs, register-blocked 4x4).  d=200 is not a power of two
// and needs no padding beyond rounding to a multiple of 4.
// Where two correlations share the b operand (h star t and r star t), the a operands are stored
// interleaved (h_i, r_i) and the pair (c_hr[q]) is accumulated with ONE v_pk_fma_f32 per (i, q):
// (c1,c2) += (h_i, r_i) * b_{i+k+q} -- the packed halves are the two correlations, so every LDS
// read stays a 16-byte aligned ds_read_b128 (4 reads per 16 packed FMAs = LDS 256 B/clk balanced
// against the packed-fp32 rate).  This file is compiled with -fno-slp-vectorize (build.py): the SLP
// vectoriser pairs the scalar form over adjacent lags instead, whose b pairs are unaligned for odd
// i and get re-read with 4-way bank-conflicting ds_read2_b32 (measured 2x slower).
// The gradients are three more correlations of the same shape:
//   ds/dr_m = (h star t)_m     ds/dh_m = (r star t)_m     ds/dt_m = (rev(r) star h)_m,
// with rev(r)_i = r_{(-i) mod d}  (so that sum_k r_k h_{(m-k)} becomes a correlation).
#include "ge_common.h"

namespace ge {

struct HoleFwd {
  float s_raw, s, sig;
  float sc[3], inv[3];
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// One 16-byte aligned LDS read that stays ONE ds_read_b128: a volatile access is neither split into
// dwords nor merged with the overlapping window of the next loop iteration (the optimiser otherwise
// carries half of b across iterations and fetches the rest as bank-conflicting ds_read2_b32).
__device__ __forceinline__ f4 lds_read16(const float* p) {
  typedef const volatile __attribute__((address_space(3))) f4* lds_f4_ptr;
  return *(lds_f4_ptr)p;   // explicit LDS address space: a volatile generic access would become a flat load
}

// 4 lags per lane per chunk: c[q] += sum_i a[i] * b[i + k0 + q]
__device__ __forceinline__ void corr4(const float* __restrict__ a, const float* __restrict__ b, int d4,
                                      int k0, float (&c)[4]) {
  c[0] = c[1] = c[2] = c[3] = 0.f;
  const float* bk = b + k0;
  int i = 0;
  for (; i + 8 <= d4; i += 8) {   // 8 i per step: 5 LDS reads per 32 FMAs (the 11-float window of b is 3 reads)
    const f4 a0 = lds_read16(a + i), a1 = lds_read16(a + i + 4);
    const f4 b0 = lds_read16(bk + i), b1 = lds_read16(bk + i + 4), b2 = lds_read16(bk + i + 8);
    const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
    const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] += aa[u] * bb[u + q];
  }
  for (; i < d4; i += 4) {
    const f4 av = lds_read16(a + i);
    const f4 b0 = lds_read16(bk + i);
    const f4 b1 = lds_read16(bk + i + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const float aa[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] += aa[u] * bb[u + q];
  }
}

// Two correlations against the same doubled operand b in one pass:
// c[q] = (c1[q], c2[q]),  c1[q] += sum_i a1[i] b[i+k0+q],  c2[q] += sum_i a2[i] b[i+k0+q],
// with a12 = interleaved (a1[i], a2[i]) pairs -- 4 aligned 16-byte LDS reads per 16 packed FMAs.
__device__ __forceinline__ void corr4x2(const float* __restrict__ a12, const float* __restrict__ b, int d4,
                                        int k0, float (&c1)[4], float (&c2)[4]) {
  f2 c[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = f2{0.f, 0.f};
  const float* bk = b + k0;
  int i = 0;
  for (; i + 8 <= d4; i += 8) {   // 8 i per step: 7 LDS reads per 32 packed FMAs
    const f4 p0 = lds_read16(a12 + 2 * i), p1 = lds_read16(a12 + 2 * i + 4);
    const f4 p2 = lds_read16(a12 + 2 * i + 8), p3 = lds_read16(a12 + 2 * i + 12);
    const f4 b0 = lds_read16(bk + i), b1 = lds_read16(bk + i + 4), b2 = lds_read16(bk + i + 8);
    const float bb[12] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w};
    const f2 aa[8] = {f2{p0.x, p0.y}, f2{p0.z, p0.w}, f2{p1.x, p1.y}, f2{p1.z, p1.w},
                      f2{p2.x, p2.y}, f2{p2.z, p2.w}, f2{p3.x, p3.y}, f2{p3.z, p3.w}};
#pragma unroll
    for (int w = 0; w < 8; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = __builtin_elementwise_fma(aa[w], f2{bb[w + q], bb[w + q]}, c[q]);
  }
  for (; i < d4; i += 4) {
    const f4 p0 = lds_read16(a12 + 2 * i);
    const f4 p1 = lds_read16(a12 + 2 * i + 4);
    const f4 b0 = lds_read16(bk + i);
    const f4 b1 = lds_read16(bk + i + 4);
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    const f2 aa[4] = {f2{p0.x, p0.y}, f2{p0.z, p0.w}, f2{p1.x, p1.y}, f2{p1.z, p1.w}};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
      for (int q = 0; q < 4; ++q) c[q] = __builtin_elementwise_fma(aa[w], f2{bb[w + q], bb[w + q]}, c[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) { c1[q] = c[q].x; c2[q] = c[q].y; }
}

// A wavefront's LDS slice is private to it, so the staging -> correlation hand-off needs ordering only
// inside the wave: its LDS operations execute in program order; the fence stops the compiler from
// moving accesses across the hand-off.  (No workgroup barrier: waves whose pair is hinge-inactive
// leave early.)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS slice of one wavefront (floats):  A[2*d4]  B[LB],  LB = d4 + NCH*256.
//   forward  (score, and phase 1 of the gradient): A = h and r (GRAD: interleaved (h_i, r_i)), B = t doubled
//   backward (phase 2 of the gradient, active pairs only): A = rev(r) [d4], B = h doubled
__device__ __forceinline__ int hole_slice_floats(int d4, int nch) { return 2 * d4 + d4 + nch * 256; }

// Phase 1: stage h, r, t; (h star t) [and (r star t) when GRAD]; raw score, clip scales, sigmoid.
template <int NCH, bool GRAD>
__device__ __forceinline__ void hole_forward(const float* __restrict__ rows, int d, int d4,
                                             const int32_t (&id)[3], float* __restrict__ lds, int lane,
                                             float max_norm, HoleFwd& f, float (&Gr)[NCH][4],
                                             float (&Gh)[NCH][4]) {
  const int LB = d4 + NCH * 256;
  float* a_h = lds;               // !GRAD: h[d4] r[d4];   GRAD: interleaved (h_i, r_i)[2*d4]
  float* a_r = a_h + d4;
  float* b_t = a_h + 2 * d4;
  const float* xh = rows + (int64_t)id[0] * d;
  const float* xt = rows + (int64_t)id[1] * d;
  const float* xr = rows + (int64_t)id[2] * d;
  float ssh = 0.f, sst = 0.f, ssr = 0.f;
  if ((d & 3) == 0) {
    // 16-byte path (d == d4): one float4 per lane and row; the doubled operand needs exactly two copies
    // (the largest index read is k0 + i + 7 <= 2d - 1).
    for (int c4 = lane; c4 < (d >> 2); c4 += kWave) {
      const float4 h4 = *reinterpret_cast<const float4*>(xh + 4 * c4);
      const float4 t4 = *reinterpret_cast<const float4*>(xt + 4 * c4);
      const float4 r4 = *reinterpret_cast<const float4*>(xr + 4 * c4);
      ssh += h4.x * h4.x + h4.y * h4.y + h4.z * h4.z + h4.w * h4.w;
      sst += t4.x * t4.x + t4.y * t4.y + t4.z * t4.z + t4.w * t4.w;
      ssr += r4.x * r4.x + r4.y * r4.y + r4.z * r4.z + r4.w * r4.w;
      if (GRAD) {
        *reinterpret_cast<float4*>(a_h + 8 * c4) = make_float4(h4.x, r4.x, h4.y, r4.y);
        *reinterpret_cast<float4*>(a_h + 8 * c4 + 4) = make_float4(h4.z, r4.z, h4.w, r4.w);
      } else {
        *reinterpret_cast<float4*>(a_h + 4 * c4) = h4;
        *reinterpret_cast<float4*>(a_r + 4 * c4) = r4;
      }
      *reinterpret_cast<float4*>(b_t + 4 * c4) = t4;
      *reinterpret_cast<float4*>(b_t + d + 4 * c4) = t4;
    }
  } else {
    for (int j = lane; j < d4; j += kWave) {
      const bool in = j < d;
      const float vh = in ? xh[j] : 0.f, vr = in ? xr[j] : 0.f, vt = in ? xt[j] : 0.f;
      if (GRAD) {
        a_h[2 * j] = vh;
        a_h[2 * j + 1] = vr;
      } else {
        a_h[j] = vh;
        a_r[j] = vr;
      }
      ssh += vh * vh; ssr += vr * vr; sst += vt * vt;
    }
    for (int j = lane; j < LB; j += kWave) {
      int jm = j;
      while (jm >= d) jm -= d;
      b_t[j] = xt[jm];
    }
  }
  wave_lds_sync();
  float part = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k0 = c * 256 + 4 * lane;
    if (k0 < d4) {
      float rv[4];
      if (GRAD) {
        corr4x2(a_h, b_t, d4, k0, Gr[c], Gh[c]);       // (h star t) = ds/dr and (r star t) = ds/dh
        const float4 u = *reinterpret_cast<const float4*>(a_h + 2 * k0);
        const float4 v = *reinterpret_cast<const float4*>(a_h + 2 * k0 + 4);
        rv[0] = u.y; rv[1] = u.w; rv[2] = v.y; rv[3] = v.w;
      } else {
        corr4(a_h, b_t, d4, k0, Gr[c]);                // (h star t): the score
        const float4 u = *reinterpret_cast<const float4*>(a_r + k0);
        rv[0] = u.x; rv[1] = u.y; rv[2] = u.z; rv[3] = u.w;
      }
      part += rv[0] * Gr[c][0] + rv[1] * Gr[c][1] + rv[2] * Gr[c][2] + rv[3] * Gr[c][3];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { Gr[c][q] = 0.f; Gh[c][q] = 0.f; }
    }
  }
  f.s_raw = group_sum<kWave>(part);
  ssh = group_sum<kWave>(ssh); sst = group_sum<kWave>(sst); ssr = group_sum<kWave>(ssr);
  f.sc[0] = clip_scale(ssh, max_norm, f.inv[0]);
  f.sc[1] = clip_scale(sst, max_norm, f.inv[1]);
  f.sc[2] = clip_scale(ssr, max_norm, f.inv[2]);
  f.s = f.s_raw * f.sc[0] * f.sc[1] * f.sc[2];
  f.sig = sigmoidf_dev(f.s);
  wave_lds_sync();  // the slice is reused by the next side / phase / triple
}

// Phase 2 (hinge-active pairs only): stage rev(r) and h doubled; (rev r star h) = ds/dt.
template <int NCH>
__device__ __forceinline__ void hole_backward_t(const float* __restrict__ rows, int d, int d4,
                                                const int32_t (&id)[3], float* __restrict__ lds, int lane,
                                                float (&Gt)[NCH][4]) {
  const int LB = d4 + NCH * 256;
  float* a_rr = lds;
  float* b_h = lds + 2 * d4;
  const float* xh = rows + (int64_t)id[0] * d;
  const float* xr = rows + (int64_t)id[2] * d;
  if ((d & 3) == 0) {
    for (int c4 = lane; c4 < (d >> 2); c4 += kWave) {
      const float4 h4 = *reinterpret_cast<const float4*>(xh + 4 * c4);
      const float* rr = xr + d - 4 * c4;   // rev(r)_j = r_{(d - j) mod d}, j = 4*c4 + m
      *reinterpret_cast<float4*>(a_rr + 4 * c4) = make_float4(c4 == 0 ? xr[0] : rr[0], rr[-1], rr[-2], rr[-3]);
      *reinterpret_cast<float4*>(b_h + 4 * c4) = h4;
      *reinterpret_cast<float4*>(b_h + d + 4 * c4) = h4;
    }
  } else {
    for (int j = lane; j < d4; j += kWave) a_rr[j] = j < d ? xr[j == 0 ? 0 : d - j] : 0.f;
    for (int j = lane; j < LB; j += kWave) {
      int jm = j;
      while (jm >= d) jm -= d;
      b_h[j] = xh[jm];
    }
  }
  wave_lds_sync();
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k0 = c * 256 + 4 * lane;
    if (k0 < d4) corr4(a_rr, b_h, d4, k0, Gt[c]);
    else {
#pragma unroll
      for (int q = 0; q < 4; ++q) Gt[c][q] = 0.f;
    }
  }
  wave_lds_sync();
}

__device__ __forceinline__ bool hbad3(int64_t N, const int32_t (&t)[3]) {
  return t[0] < 0 || t[1] < 0 || t[2] < 0 || t[0] >= N || t[1] >= N || t[2] >= N;
}

template <int NCH>
__global__ __launch_bounds__(kBlock) void hole_score_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ triples, int64_t B,
    float max_norm, int apply_sigmoid, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d4 = (d + 3) & ~3;
  const int per_wave = hole_slice_floats(d4, NCH);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  float* lds = smem + w * per_wave;
  constexpr int WPB = kBlock / kWave;
  for (int64_t base = (int64_t)blockIdx.x * WPB; base < B; base += (int64_t)gridDim.x * WPB) {
    const int64_t g = base + w;
    const bool live = g < B;
    int32_t id[3] = {0, 0, 0};
    if (live) { id[0] = triples[3 * g]; id[1] = triples[3 * g + 1]; id[2] = triples[3 * g + 2]; }
    const bool bad = hbad3(N, id);
    if (bad) { id[0] = id[1] = id[2] = 0; }
    HoleFwd f;
    float Gr[NCH][4], Gh[NCH][4];
    hole_forward<NCH, false>(table, d, d4, id, lds, lane, max_norm, f, Gr, Gh);
    if (live && lane == 0) out[g] = bad ? __builtin_nanf("") : (apply_sigmoid ? f.sig : f.s);
  }
}

template <int NCH>
__global__ __launch_bounds__(kBlock) void hole_hinge_loss_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float max_norm,
    float* __restrict__ loss, float* __restrict__ sig_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d4 = (d + 3) & ~3;
  const int per_wave = hole_slice_floats(d4, NCH);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  float* lds = smem + w * per_wave;
  constexpr int WPB = kBlock / kWave;
  for (int64_t base = (int64_t)blockIdx.x * WPB; base < B; base += (int64_t)gridDim.x * WPB) {
    const int64_t g = base + w;
    const bool live = g < B;
    int32_t p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = pos[3 * g + c]; n[c] = neg[3 * g + c]; }
    }
    const bool bad = hbad3(N, p) || hbad3(N, n);
    if (bad) { p[0] = p[1] = p[2] = n[0] = n[1] = n[2] = 0; }
    HoleFwd fp, fn;
    float Gr[NCH][4], Gh[NCH][4];
    hole_forward<NCH, false>(table, d, d4, p, lds, lane, max_norm, fp, Gr, Gh);
    hole_forward<NCH, false>(table, d, d4, n, lds, lane, max_norm, fn, Gr, Gh);
    if (live && lane == 0) {
      const float nanv = __builtin_nanf("");
      loss[g] = bad ? nanv : fmaxf(fp.sig - fn.sig + margin, 0.f);
      if (sig_out) { sig_out[g] = bad ? nanv : fp.sig; sig_out[B + g] = bad ? nanv : fn.sig; }
    }
  }
}

struct HCoef { float alpha, beta; };
__device__ __forceinline__ HCoef hole_coef(float coef, const HoleFwd& f, int X, float max_norm, float neg_lr) {
  const float P = (X == 0 ? f.sc[1] * f.sc[2] : X == 1 ? f.sc[0] * f.sc[2] : f.sc[0] * f.sc[1]);
  const float A = coef * P;
  const float inv = f.inv[X];
  const bool active = inv <= 1.0f / max_norm;  // MinimumGrad routes to the rsqrt branch
  HCoef c;
  c.alpha = neg_lr * (active ? max_norm * A * inv : A);
  c.beta = active ? neg_lr * (-max_norm * A * f.s_raw * inv * inv * inv) : 0.f;
  return c;
}

// IndexedSlices of d(sum_i L_i)/d(rows) * (-lr); slot order h+, t+, r+, h-, t-, r- (see ge_hip.h).
template <int NCH>
__global__ __launch_bounds__(kBlock, NCH == 1 ? 4 : 2) void hole_hinge_grad_kernel(
    const float* __restrict__ rows, int64_t N, int d, const int32_t* __restrict__ pos,
    const int32_t* __restrict__ neg, int64_t B, float margin, float lr, float max_norm,
    float* __restrict__ loss, int32_t* __restrict__ grad_idx, float* __restrict__ grad_val) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d4 = (d + 3) & ~3;
  const int per_wave = hole_slice_floats(d4, NCH);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x >> 6;
  float* lds = smem + w * per_wave;
  constexpr int WPB = kBlock / kWave;
  const float neg_lr = -lr;
  for (int64_t base = (int64_t)blockIdx.x * WPB; base < B; base += (int64_t)gridDim.x * WPB) {
    const int64_t g = base + w;
    const bool live = g < B;
    int32_t p[3] = {0, 0, 0}, n[3] = {0, 0, 0};
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { p[c] = pos[3 * g + c]; n[c] = neg[3 * g + c]; }
    }
    const bool bad = hbad3(N, p) || hbad3(N, n);
    if (bad) { p[0] = p[1] = p[2] = n[0] = n[1] = n[2] = 0; }
    HoleFwd fp, fn;
    // G[side][X]: X = 0 h, 1 t, 2 r
    float GP[3][NCH][4], GN[3][NCH][4];
    // one pair per wave: make the ids scalar so the sharing tests below are scalar branches
#pragma unroll
    for (int c = 0; c < 3; ++c) { p[c] = __builtin_amdgcn_readfirstlane(p[c]); n[c] = __builtin_amdgcn_readfirstlane(n[c]); }
    hole_forward<NCH, true>(rows, d, d4, p, lds, lane, max_norm, fp, GP[2], GP[0]);
    // pos and neg differ in one entity, so one of the six correlations is shared (bitwise the same sum):
    // head corrupted -> (r star t) is common, tail corrupted -> (rev r star h) is common.
    if (n[1] == p[1] && n[2] == p[2]) {
      hole_forward<NCH, false>(rows, d, d4, n, lds, lane, max_norm, fn, GN[2], GN[0]);   // only (h' star t)
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) GN[0][c][q] = GP[0][c][q];
    } else {
      hole_forward<NCH, true>(rows, d, d4, n, lds, lane, max_norm, fn, GN[2], GN[0]);
    }
    const float pre = fp.sig - fn.sig + margin;
    const bool on = live && !bad && (pre >= 0.f);
    if (live && lane == 0) loss[g] = bad ? __builtin_nanf("") : fmaxf(pre, 0.f);
    if (!on) {   // wave-uniform: the pair's gradient is exactly zero -- no third correlation, empty slots
      if (live && lane < 6) grad_idx[g * 6 + lane] = -1;
      continue;
    }
    hole_backward_t<NCH>(rows, d, d4, p, lds, lane, GP[1]);
    if (n[0] == p[0] && n[2] == p[2]) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) GN[1][c][q] = GP[1][c][q];
    } else {
      hole_backward_t<NCH>(rows, d, d4, n, lds, lane, GN[1]);
    }
    const float cp = fp.sig * (1.f - fp.sig), cn = -fn.sig * (1.f - fn.sig);
#pragma unroll
    for (int X = 0; X < 3; ++X) {
      const bool same = p[X] == n[X];
      const int64_t rowP = g * 6 + X, rowN = g * 6 + 3 + X;
      if (live && lane == 0) {
        grad_idx[rowP] = on ? p[X] : -1;
        grad_idx[rowN] = (on && !same) ? n[X] : -1;
      }
      if (!on) continue;
      const HCoef kp = hole_coef(cp, fp, X, max_norm, neg_lr);
      const HCoef kn = hole_coef(cn, fn, X, max_norm, neg_lr);
      const float* xp = rows + (int64_t)p[X] * d;
      const float* xn = rows + (int64_t)n[X] * d;
      float* gp = grad_val + rowP * d;
      float* gn = grad_val + rowN * d;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int k0 = c * 256 + 4 * lane;
        if ((d & 3) == 0) {                       // 16-byte path: the lane's 4 lags are one float4
          if (k0 >= d) continue;
          const float4 p4 = *reinterpret_cast<const float4*>(xp + k0);
          const float4 n4 = *reinterpret_cast<const float4*>(xn + k0);
          const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, nv[4] = {n4.x, n4.y, n4.z, n4.w};
          float vp[4], vn[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vp[q] = kp.alpha * GP[X][c][q] + kp.beta * pv[q];
            vn[q] = kn.alpha * GN[X][c][q] + kn.beta * nv[q];
          }
          if (same) {
            *reinterpret_cast<float4*>(gp + k0) = make_float4(vp[0] + vn[0], vp[1] + vn[1], vp[2] + vn[2], vp[3] + vn[3]);
          } else {
            *reinterpret_cast<float4*>(gp + k0) = make_float4(vp[0], vp[1], vp[2], vp[3]);
            *reinterpret_cast<float4*>(gn + k0) = make_float4(vn[0], vn[1], vn[2], vn[3]);
          }
          continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kk = k0 + q;
          if (kk >= d) continue;
          const float vp = kp.alpha * GP[X][c][q] + kp.beta * xp[kk];
          const float vn = kn.alpha * GN[X][c][q] + kn.beta * xn[kk];
          if (same) gp[kk] = vp + vn;
          else { gp[kk] = vp; gn[kk] = vn; }
        }
      }
    }
  }
}

static inline size_t hole_lds_bytes(int d, int nch) {
  const int d4 = (d + 3) & ~3;
  return sizeof(float) * (size_t)(kBlock / kWave) * (size_t)(2 * d4 + d4 + nch * 256);
}

int hole_max_dim() { return 512; }

int hole_score_launch(const float* table, int64_t N, int32_t d, const int32_t* triples, int64_t B,
                      float max_norm, int apply_sigmoid, float* out, hipStream_t st) {
  if (d <= 0) return GE_EINVAL;
  if (d > hole_max_dim()) return GE_ENOTSUP;
  if (B == 0) return 0;
  const int nch = d <= 256 ? 1 : 2;
  const int grid = grid_for(B, kBlock / kWave);
  const size_t lds = hole_lds_bytes(d, nch);
  if (nch == 1)
    hipLaunchKernelGGL(hole_score_kernel<1>, dim3(grid), dim3(kBlock), lds, st, table, N, d, triples, B, max_norm, apply_sigmoid, out);
  else
    hipLaunchKernelGGL(hole_score_kernel<2>, dim3(grid), dim3(kBlock), lds, st, table, N, d, triples, B, max_norm, apply_sigmoid, out);
  return launch_status();
}

int hole_hinge_loss_launch(const float* table, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                           int64_t B, float margin, float max_norm, float* loss, float* sig_out,
                           hipStream_t st) {
  if (d <= 0) return GE_EINVAL;
  if (d > hole_max_dim()) return GE_ENOTSUP;
  if (B == 0) return 0;
  const int nch = d <= 256 ? 1 : 2;
  const int grid = grid_for(B, kBlock / kWave);
  const size_t lds = hole_lds_bytes(d, nch);
  if (nch == 1)
    hipLaunchKernelGGL(hole_hinge_loss_kernel<1>, dim3(grid), dim3(kBlock), lds, st, table, N, d, pos, neg, B, margin, max_norm, loss, sig_out);
  else
    hipLaunchKernelGGL(hole_hinge_loss_kernel<2>, dim3(grid), dim3(kBlock), lds, st, table, N, d, pos, neg, B, margin, max_norm, loss, sig_out);
  return launch_status();
}

int hole_hinge_grad_launch(const float* rows, int64_t N, int32_t d, const int32_t* pos, const int32_t* neg,
                           int64_t B, float margin, float lr, float max_norm, float* loss,
                           int32_t* grad_idx, float* grad_val, hipStream_t st, hipEvent_t ev_start,
                           hipEvent_t ev_stop) {
  if (d <= 0) return GE_EINVAL;
  if (d > hole_max_dim()) return GE_ENOTSUP;
  if (B == 0) return 0;
  const int nch = d <= 256 ? 1 : 2;
  const int grid = grid_for(B, kBlock / kWave);
  const size_t lds = hole_lds_bytes(d, nch);
  if (nch == 1)
    hipExtLaunchKernelGGL(hole_hinge_grad_kernel<1>, dim3(grid), dim3(kBlock), lds, st, ev_start, ev_stop, 0, rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val);
  else
    hipExtLaunchKernelGGL(hole_hinge_grad_kernel<2>, dim3(grid), dim3(kBlock), lds, st, ev_start, ev_stop, 0, rows, N, d, pos, neg, B, margin, lr, max_norm, loss, grad_idx, grad_val);
  return launch_status();
}

}  // namespace ge
