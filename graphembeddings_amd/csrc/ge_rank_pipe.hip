// ge_rank_pipe.hip -- the link-prediction ranking sweep (holE.py:427-472, 564-575; semantics in ge_rank.hip) on the fp32
// MFMA, software-pipelined for one wave per SIMD.  It serves what the split-precision sweep (ge_rank_f16.hip, tried
// first by sweep_pipe_launch) does not: embedding_dim % 8 == 0 below 56, and max_norm > 8.
//
// What the microbenchmark (tools/probes/mfma_probe.hip) says about v_mfma_f32_32x32x2_f32 on gfx950 with one
// wave per SIMD (the Q operand fills the LDS, so a CU holds one workgroup):
//   * LDS reads, LDS writes and global loads issued BETWEEN matrix instructions are free (2-3 cycles each);
//   * operand reads issued up front are not (+8 % per 64-MFMA chunk), a workgroup barrier costs ~5 %;
//   * VALU instructions never hide behind the matrix pipe: 2.3 (bursts) to 4.9 (spread out) cycles of MFMA time
//     each in the probe, more in the kernel when they chain through scalar registers.
// So the kernel is organised around the MFMA stream and nothing else may take VALU slots:
//   * the candidate chunk width CW divides embedding_dim (40 for d = 200: 5 chunks, no tail, no masking);
//   * operands travel LDS -> registers in groups of 4 k-pairs (16 MFMAs); group g+1 is read between the MFMAs
//     of group g, into the other of two 16-register sets, across chunk boundaries too;
//   * candidate rows travel global -> registers (chunk q+2) -> LDS (chunk q+1) between the MFMAs of chunk q;
//     addresses advance by immediates, requests are never predicated (a branch makes the waitcnt pass serialise
//     them), the one barrier per chunk sits before the chunk's last operand group; the row norm is two FMA chains;
//   * the epilogue compares RAW scores: per test row two thresholds lo/hi bracket the true candidate's score by
//     more than the sigmoid's rounding, x < lo means "pops before", x > hi "pops after".  Per score: a multiply,
//     two compares, two v_writelane (the "before" bit into the row-major mask) and two v_addc (both outcomes
//     into per-lane bitmaps).  Where the bitmaps differ the score sits inside a bracket: only those lanes
//     evaluate the exact fp32 sigmoid comparison with the id tie-break and set the bit in LDS.  The outcome
//     equals comparing the sigmoids everywhere;
//   * the grid is two workgroups per CU, each with an equal share of the (row block x candidate tile) list;
//   * model: ComplEx, or HolE on a table held in the frequency domain (same GEMM; Hermitian weights go into Q,
//     the candidate norm is Parseval-weighted).
#include <algorithm>
#include <type_traits>

#include "ge_rank_dev.h"

#ifndef GE_PIPE_GRID_M
#define GE_PIPE_GRID_M 2   // workgroups per CU (each CU holds one at a time): equal shares, two rounds
#endif

namespace ge {
namespace {

// compile-time loop: f(integral_constant<int, 0>{}) ... f(integral_constant<int, N-1>{})
template <int I0, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, N>(f);
  }
}

template <int CW>
struct Cfg {
  static constexpr int kGS = 4;                 // k-pairs per operand group (16 MFMAs)
  static constexpr int kNG = CW / 2 / kGS;      // operand groups per chunk
  static constexpr int kNV = CW / 8;            // 16-byte requests per staging thread per chunk
  static constexpr int kLdb = CW + 1;           // odd LDS row stride
  static_assert(CW % 8 == 0 && kNG >= 2 && 4 * (kNG - 1) >= 2 * kNV, "chunk width");
};

typedef float f2 __attribute__((ext_vector_type(2)));

struct Ops { float a0[4], a1[4], b0[4], b1[4]; };   // 4 k-pairs of this wave's 64 x 64 block: 2 A and 2 B fragments

struct PipeLds {
  float* A;        // [kRB][lda]  q = fixed o relation, not yet scaled by the rows' clip scales
  float* Bs;       // [2][kRB][CW+1]
  float* sA;       // [kRB] product of the fixed and relation rows' clip scales (NaN: bad id / beyond B)
  float* sB;       // [kRB] candidate clip scale (NaN: bad id / beyond K)
  float* eT;       // [kRB] loss of the true candidate
  float2* lohi;    // [kRB] raw-score bracket of the true candidate
  unsigned* bm;    // [kRB][4] `pops before` bits of the current tile
  int* skip;       // [kRB] known-true candidates ranked before the target
  int* tI;         // [kRB] entity id of the true candidate (-1 beyond B)
};

template <int CW>
size_t pipe_lds_bytes(int d) {
  return sizeof(float) * ((size_t)kRB * (d + 1) + 2 * kRB * Cfg<CW>::kLdb + 3 * kRB) + sizeof(float2) * kRB +
         sizeof(unsigned) * kRB * 4 + sizeof(int) * 2 * kRB;
}

// chunk `c` (clamped to the row) of candidate row `cid`: thread t requests its CW/2 reals of row t>>1
template <int CW>
__device__ __forceinline__ void pipe_fetch(const float* __restrict__ table, int64_t N, int d, int32_t cid, int c,
                                           float4 (&r)[Cfg<CW>::kNV]) {
  const int half = threadIdx.x & 1;
  const bool bad = cid < 0 || cid >= N;
  const float* src = table + (int64_t)(bad ? 0 : cid) * d + half * (CW / 2) + min(c, d / CW - 1) * CW;
#pragma unroll
  for (int v = 0; v < Cfg<CW>::kNV; ++v) r[v] = *reinterpret_cast<const float4*>(src + 4 * v);
}

// piece i (0..7) of an operand group: two k-pairs of one fragment
__device__ __forceinline__ void ops_piece(Ops& o, const float* __restrict__ ap, const float* __restrict__ bp, int lda,
                                          int ldb, int g, int i) {
  const int s0 = 2 * (i & 1), k2 = 2 * (4 * g + s0);
  switch (i >> 1) {
    case 0: o.a0[s0] = ap[k2]; o.a0[s0 + 1] = ap[k2 + 2]; break;
    case 1: o.b0[s0] = bp[k2]; o.b0[s0 + 1] = bp[k2 + 2]; break;
    case 2: o.b1[s0] = bp[32 * ldb + k2]; o.b1[s0 + 1] = bp[32 * ldb + k2 + 2]; break;
    default: o.a1[s0] = ap[32 * lda + k2]; o.a1[s0 + 1] = ap[32 * lda + k2 + 2]; break;
  }
}

// One 128 x 128 tile: acc = Q . T^T for the candidate rows `cid` (this thread stages row t>>1, half t&1),
// candidate clip scales to lds.sB.  rA / rB hold chunks 0 and 1 of the row on entry.
template <int CW, int NCH>
__device__ __forceinline__ void pipe_tile(const float* __restrict__ table, int64_t N, int d, int lda, int32_t cid,
                                          float max_norm, int spec, const PipeLds& lds,
                                          float4 (&rA)[Cfg<CW>::kNV], float4 (&rB)[Cfg<CW>::kNV],
                                          f32x16 (&acc)[2][2]) {
  using C = Cfg<CW>;
  constexpr int NG = C::kNG, NV = C::kNV, LDB = C::kLdb;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int srow = t >> 1, half = t & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool bad = cid < 0 || cid >= N;
  const float* crow = table + (int64_t)(bad ? 0 : cid) * d + half * (CW / 2);
  // spectral HolE rows (ge_complex_dev.h): |x|^2 = (2 sum - X_0^2 - X_k^2) / d.  Both reals are requested here,
  // unconditionally, and used at the end of the tile.
  const float x_dc = crow[-half * (CW / 2)], x_ny = crow[-half * (CW / 2) + (d >> 1)];
  const int n_chunks = NCH ? NCH : d / CW;                          // NCH: chunks per row known at compile time (0: loop)
  const float* ap0 = lds.A + (wm * 64 + li) * lda + lh;             // this lane's A fragment rows, chunk 0
  const float* bp0 = lds.Bs + (wn * 64 + li) * LDB + lh;            // this lane's B fragment rows, buffer 0
  float* st0 = lds.Bs + srow * LDB + half * (CW / 2);               // this thread's staging slice, buffer 0
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  f2 ss2 = {0.f, 0.f};                                              // the row's sum of squares, one v_pk_fma_f32 per 2 reals
#pragma unroll
  for (int v = 0; v < NV; ++v) {                                    // chunk 0 -> buffer 0
    st0[4 * v] = rA[v].x; st0[4 * v + 1] = rA[v].y; st0[4 * v + 2] = rA[v].z; st0[4 * v + 3] = rA[v].w;
    ss2 = __builtin_elementwise_fma(f2{rA[v].x, rA[v].y}, f2{rA[v].x, rA[v].y}, ss2);
    ss2 = __builtin_elementwise_fma(f2{rA[v].z, rA[v].w}, f2{rA[v].z, rA[v].w}, ss2);
  }
  __syncthreads();
  Ops ops[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) ops_piece(ops[0], ap0, bp0, lda, LDB, 0, i);

  // chunk q from LDS buffer BUF (= q & 1).  rs holds chunk q+1 (stored to the other buffer here), rf receives q+2.
  auto chunk = [&](auto buf_c, auto last_c, int q, float4 (&rs)[NV], float4 (&rf)[NV]) {
    constexpr int BUF = decltype(buf_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    constexpr int P0 = (BUF & NG) & 1;                              // operand set holding group 0 of this chunk
    const float* apq = ap0 + q * CW;
    const float* bpq = bp0 + BUF * kRB * LDB;
    const float* bpn = bp0 + (BUF ^ 1) * kRB * LDB;
    float* dst = st0 + (BUF ^ 1) * kRB * LDB;
    const float* gsrc = crow + min(q + 2, n_chunks - 1) * CW;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!LAST && g == NG - 1) __syncthreads();                    // chunk q+1 is in LDS; chunk q-1's buffer is free
      Ops& cur = ops[(P0 + g) & 1];
      Ops& nxt = ops[(P0 + g + 1) & 1];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int s = j >> 2;
        switch (j & 3) {
          case 0: acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[s], cur.b0[s], acc[0][0], 0, 0, 0); break;
          case 1: acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[s], cur.b1[s], acc[0][1], 0, 0, 0); break;
          case 2: acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[s], cur.b0[s], acc[1][0], 0, 0, 0); break;
          default: acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[s], cur.b1[s], acc[1][1], 0, 0, 0); break;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (j < 8) {                                                // the next operand group
          if (g + 1 < NG) ops_piece(nxt, apq, bpq, lda, LDB, g + 1, j);
          else if (!LAST) ops_piece(nxt, apq + CW, bpn, lda, LDB, 0, j);
        } else if (j < 12) {                                        // half a float4 of chunk q+1 to LDS
          const int sp = 4 * g + (j - 8), v = sp >> 1;
          if (!LAST && g < NG - 1 && sp < 2 * NV) {
            if (sp & 1) {
              dst[4 * v + 2] = rs[v].z; dst[4 * v + 3] = rs[v].w;
              ss2 = __builtin_elementwise_fma(f2{rs[v].z, rs[v].w}, f2{rs[v].z, rs[v].w}, ss2);
            } else {
              dst[4 * v] = rs[v].x; dst[4 * v + 1] = rs[v].y;
              ss2 = __builtin_elementwise_fma(f2{rs[v].x, rs[v].y}, f2{rs[v].x, rs[v].y}, ss2);
            }
          }
        } else {                                                    // one 16-byte request of chunk q+2
          const int fp = 4 * g + (j - 12);
          if (!LAST && fp < NV) rf[fp] = *reinterpret_cast<const float4*>(gsrc + 4 * fp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  if constexpr (NCH > 0) {                                          // straight-line: no accumulator copies at joins
    static_for<0, NCH>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (q & 1) chunk(B1{}, std::bool_constant<q == NCH - 1>{}, q, rA, rB);
      else chunk(B0{}, std::bool_constant<q == NCH - 1>{}, q, rB, rA);
    });
  } else {
    int q = 0;
    for (; q + 2 < n_chunks; q += 2) {                              // register sets swap roles every chunk
      chunk(B0{}, std::false_type{}, q, rB, rA);
      chunk(B1{}, std::false_type{}, q + 1, rA, rB);
    }
    if (n_chunks - q == 2) {
      chunk(B0{}, std::false_type{}, q, rB, rA);
      chunk(B1{}, std::true_type{}, q + 1, rA, rB);
    } else {
      chunk(B0{}, std::true_type{}, q, rB, rA);
    }
  }
  float ss = ss2.x + ss2.y;
  ss += __shfl_xor(ss, 1, kWave);
  if (spec) ss = (2.f * ss - x_dc * x_dc - x_ny * x_ny) / (float)d;
  if (half == 0) {
    float inv;
    lds.sB[srow] = bad ? __builtin_nanf("") : clip_scale(ss, max_norm, inv);
  }
  __syncthreads();
}

// v_writelane_b32 with a constant lane (this clang has no builtin for it): lane `LANE` of m = the wave-uniform v.
template <int LANE>
__device__ __forceinline__ void set_lane(int& m, unsigned v) {
  // (s_nop 1: v may come straight out of a VALU compare, and on gfx950 a VALU read of a VALU-written SGPR needs two wait
  // states, which the compiler's hazard recognizer cannot insert for an instruction inside inline asm)
  asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(v), "n"(LANE));
}

// m = 2 * m + (this lane's bit of the wave mask): one v_addc_co_u32 with the mask as carry-in
__device__ __forceinline__ void shift_in(unsigned& m, unsigned long long mask) {
  unsigned long long carry_out;
  asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(m), "=s"(carry_out) : "s"(mask));
}

// The exact comparison for one accumulator register of both column halves: fp32 sigmoid of the scaled score, ties
// by entity id (the reference's heap pops equal losses in triple order, holE.py:427-472).
template <bool SCORES>
__device__ __forceinline__ void exact_masks(const PipeLds& lds, float x0, float x1, int rl, int32_t c0, int32_t c1,
                                            unsigned long long& m0, unsigned long long& m1, float* scores_out,
                                            int64_t row, int64_t B, int64_t K, int64_t col0, int64_t col1) {
  const float sa = lds.sA[rl], et = lds.eT[rl];
  const int ti = lds.tI[rl];
  const float e0 = rank_sigmoid(x0 * sa), e1 = rank_sigmoid(x1 * sa);
  m0 = __ballot(e0 < et) | __ballot(e0 == et && c0 < ti);
  m1 = __ballot(e1 < et) | __ballot(e1 == et && c1 < ti);
  if (SCORES && row < B) {
    if (col0 < K) scores_out[row * K + col0] = e0;
    if (col1 < K) scores_out[row * K + col1] = e1;
  }
}

// MODE 0: ranks.  1: ranks, every loss computed exactly and stored too (tests).  2: no ranking at all -- the sweep
// writes scores_out[B,K] (raw score, or its sigmoid when `sweep_flags` & 1): ge_complex_score_1vK on this pipeline.
template <int CW, int NCH, int MODE>
__global__ __launch_bounds__(kBlock) void rank_pipe_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ true_id, const int32_t* __restrict__ cand, int64_t K, float max_norm,
    int cand_is_head, const int32_t* __restrict__ known_off, const uint16_t* __restrict__ known_rc,
    int32_t* __restrict__ raw_cnt, int32_t* __restrict__ skip_cnt, float* true_loss,
    float* __restrict__ scores_out, int n_ct, int64_t n_tiles, int spec, int sweep_flags) {
  constexpr bool SCORES = MODE == 1;
  using C = Cfg<CW>;
  constexpr int NV = C::kNV;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lda = d + 1;
  PipeLds lds;
  lds.A = smem;
  lds.Bs = lds.A + kRB * lda;
  lds.sA = lds.Bs + 2 * kRB * C::kLdb;
  lds.sB = lds.sA + kRB;
  lds.eT = lds.sB + kRB;
  lds.lohi = reinterpret_cast<float2*>(lds.eT + kRB);             // an even number of floats in: 8-byte aligned
  lds.bm = reinterpret_cast<unsigned*>(lds.lohi + kRB);
  lds.skip = reinterpret_cast<int*>(lds.bm + kRB * 4);
  lds.tI = lds.skip + kRB;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = t >> 1, half = t & 1;
  const int k = d >> 1;

  // this workgroup's share of the (row block, candidate tile) list, row-block major
  int64_t idx = n_tiles * blockIdx.x / gridDim.x;
  const int64_t idx_end = n_tiles * (blockIdx.x + 1) / gridDim.x;
  while (idx < idx_end) {
    const int rb = (int)(idx / n_ct);
    const int ct0 = (int)(idx - (int64_t)rb * n_ct);
    const int ct1 = (int)min((int64_t)n_ct, ct0 + (idx_end - idx));
    const int64_t m0 = (int64_t)rb * kRB;
    idx += ct1 - ct0;
    __syncthreads();                                             // the previous row block's LDS is done with

    // ---- Q = fixed o relation for the block's 128 rows
    {   // whole k range in fp32; the clip scales stay a per-row factor
      const int64_t r = m0 + srow;
      int32_t fid = -1, rid = -1;
      if (r < B) { fid = hr[2 * r]; rid = hr[2 * r + 1]; }
      const bool bad = fid < 0 || fid >= N || rid < 0 || rid >= N;
      const float* frow = table + (int64_t)(bad ? 0 : fid) * d;
      const float* rrow = table + (int64_t)(bad ? 0 : rid) * d;
      float ssf = 0.f, ssr = 0.f;
      float* arow = lds.A + srow * lda;
#pragma unroll 4
      for (int j = half; j < (k >> 2); j += 2) {                 // 4 complex dims per step, the row's two threads interleaved
        const float4 fre = *reinterpret_cast<const float4*>(frow + 4 * j), fim = *reinterpret_cast<const float4*>(frow + k + 4 * j);
        const float4 rre = *reinterpret_cast<const float4*>(rrow + 4 * j), rim = *reinterpret_cast<const float4*>(rrow + k + 4 * j);
        const float fr[4] = {fre.x, fre.y, fre.z, fre.w}, fi[4] = {fim.x, fim.y, fim.z, fim.w};
        const float rr[4] = {rre.x, rre.y, rre.z, rre.w}, ri[4] = {rim.x, rim.y, rim.z, rim.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // spectral HolE: Hermitian weight 2 on every bin but element 0, which packs the two REAL bins X_0 | X_k
          const bool packed = spec && j == 0 && i == 0;
          const float wgt = (spec && !packed) ? 2.f : 1.f;
          ssf += wgt * (fr[i] * fr[i] + fi[i] * fi[i]);
          ssr += wgt * (rr[i] * rr[i] + ri[i] * ri[i]);
          float qre, qim;
          if (packed) {          // two independent real dimensions: products of the re slots and of the im slots
            qre = fr[i] * rr[i];
            qim = fi[i] * ri[i];
          } else if (!cand_is_head) {   // q = h * r ; score = Re(q conj t)
            qre = fr[i] * rr[i] - fi[i] * ri[i];
            qim = fr[i] * ri[i] + fi[i] * rr[i];
          } else {               // Re(h r conj t) with h the candidate: Q = [Re(r conj t) | -Im(r conj t)]
            qre = rr[i] * fr[i] + ri[i] * fi[i];
            qim = -(ri[i] * fr[i] - rr[i] * fi[i]);
          }
          arow[4 * j + i] = qre * wgt;
          arow[k + 4 * j + i] = qim * wgt;
        }
      }
      ssf += __shfl_xor(ssf, 1, kWave);
      ssr += __shfl_xor(ssr, 1, kWave);
      if (half == 0) {
        float i0, i1;
        const float inv_d = spec ? 1.0f / (float)d : 1.0f;   // Parseval / correlation-theorem factor of the spectral form
        lds.sA[srow] = (bad || r >= B) ? __builtin_nanf("")
                                       : clip_scale(ssf * inv_d, max_norm, i0) * clip_scale(ssr * inv_d, max_norm, i1) * inv_d;
        lds.skip[srow] = 0;
        lds.tI[srow] = (MODE != 2 && r < B) ? true_id[r] : -1;
      }
    }
    __syncthreads();

    auto cand_of = [&](int ct) -> int32_t {
      const int64_t c = (int64_t)ct * kRB + srow;
      return (ct < ct1 && c < K) ? cand[c] : -1;
    };
    auto known_of = [&](int ct, int32_t& k0, int32_t& k1) {
      k0 = k1 = 0;
      if (known_off && ct < ct1) {
        const int64_t tile = (int64_t)rb * n_ct + ct;
        k0 = known_off[tile]; k1 = known_off[tile + 1];
      }
    };
    f32x16 acc[2][2];
    float4 rA[NV], rB[NV];
    // ---- the true candidates: a tile whose candidate rows are this block's 128 true entities
    if constexpr (MODE != 2) {
      const int32_t tid = lds.tI[srow];
      pipe_fetch<CW>(table, N, d, tid, 0, rA);
      pipe_fetch<CW>(table, N, d, tid, 1, rB);
      pipe_tile<CW, NCH>(table, N, d, lda, tid, max_norm, spec, lds, rA, rB, acc);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            const int cl = wn * 64 + tn * 32 + li;
            if (rl == cl) lds.eT[rl] = acc[tm][tn][q] * lds.sB[cl];     // raw score, row scale still to come
          }
        }
      __syncthreads();
      if (t < kRB) {
        // Bracket of the true candidate's raw score.  With g = e (1 - e) the sigmoid's slope at the true score and
        // w = 1e-6 / g <= 0.1, the slope anywhere inside [xs - w, xs + w] is >= g exp(-w) (the sigmoid is concave on one
        // side: a first-order bound alone is not enough), so a candidate whose scaled score lies outside has a loss that
        // differs by >= 0.9e-6, three times what the roundings of x * sA and of the 4-instruction sigmoid
        // (< 1.5e-7 each side) can move: outside the bracket the order of the losses is the order of the raw scores.
        // Near saturation (g < 1e-5, |score| > 11.5) no finite bracket gives that margin: it is infinite there and
        // every candidate of the row takes the exact comparison.
        float xp = lds.eT[t];
        const float sa = lds.sA[t];
        float xs = xp * sa, e = rank_sigmoid(xs);
        if ((sweep_flags & 2) && m0 + t < B) {
          // ranking against a GIVEN loss (ge_rank_1vK_vs_loss; the tile above ran on whatever row the tie-break id names):
          // bracket centred on the loss's logit, exact comparisons against the given value (see ge_rank_f16.hip)
          e = true_loss[m0 + t];
          const float ec = fminf(fmaxf(e, 1e-30f), 0.99999994f);
          xs = (e == e) ? logf(ec / (1.0f - ec)) : e;
          xp = xs / sa;
        }
        const float gs = e * (1.0f - e);
        const float wx = !(gs >= 1e-5f) ? __builtin_inff() : 1e-6f / gs + 4e-7f * fabsf(xs);
        const float wq = wx / sa;
        lds.lohi[t] = make_float2(xp - wq, xp + wq);
        lds.eT[t] = e;
        if (true_loss && !(sweep_flags & 2) && ct0 == 0 && m0 + t < B) true_loss[m0 + t] = e;
      }
      __syncthreads();
    }
    int raw_reg = 0;

    // ---- the sweep over this share's candidate tiles of the row block
    // candidate ids and known-cell ranges are requested one tile ahead of their use: nothing ever waits on them
    int32_t cid = cand_of(ct0), cid_next = cand_of(ct0 + 1), kn0, kn1, kn0_next, kn1_next;
    known_of(ct0, kn0_next, kn1_next);
    pipe_fetch<CW>(table, N, d, cid, 0, rA);
    pipe_fetch<CW>(table, N, d, cid, 1, rB);
    for (int ct = ct0; ct < ct1; ++ct) {
      const int64_t n0 = (int64_t)ct * kRB;
      pipe_tile<CW, NCH>(table, N, d, lda, cid, max_norm, spec, lds, rA, rB, acc);
      cid = cid_next; kn0 = kn0_next; kn1 = kn1_next;
      pipe_fetch<CW>(table, N, d, cid, 0, rA);                   // land while the epilogue below runs
      pipe_fetch<CW>(table, N, d, cid, 1, rB);
      cid_next = cand_of(ct + 2);
      known_of(ct + 1, kn0_next, kn1_next);
      // epilogue: C layout of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
      // A candidate beyond K or with a bad id has a NaN clip scale, a row beyond B a NaN bracket: no bit is set.
      const int cl0 = wn * 64 + li, cl1 = cl0 + 32;
      const float sb0 = lds.sB[cl0], sb1 = lds.sB[cl1];
      if constexpr (MODE == 2) {                                 // scores only: 32 consecutive floats of a row per half-wave
        const int64_t col0 = n0 + cl0, col1 = n0 + cl1;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            const int64_t row = m0 + rl;
            const float sa = lds.sA[rl];
            float v0 = acc[tm][0][q] * sb0 * sa, v1 = acc[tm][1][q] * sb1 * sa;
            if (sweep_flags & 1) { v0 = rank_sigmoid(v0); v1 = rank_sigmoid(v1); }   // 4 VALU, within 3e-7 of expf's
            if (row < B) {
              if (col0 < K) scores_out[row * K + col0] = v0;
              if (col1 < K) scores_out[row * K + col1] = v1;
            }
          }
        continue;
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        int M0 = 0, M1 = 0;                                      // lane r: the 32 column bits of row r of the 32 x 32 block
        const int64_t col0 = n0 + cl0, col1 = n0 + cl1;
        unsigned* mrow = lds.bm + (wm * 64 + tm * 32) * 4 + wn * 2;
        if constexpr (SCORES) {                                  // tests: every loss exactly, and stored
          const int32_t c0 = col0 < K ? cand[col0] : -1, c1 = col1 < K ? cand[col1] : -1;
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            const int rl = wm * 64 + tm * 32 + R32 + 4 * lh;
            unsigned long long e0, e1;
            exact_masks<true>(lds, acc[tm][0][q] * sb0, acc[tm][1][q] * sb1, rl, c0, c1, e0, e1, scores_out, m0 + rl, B, K,
                              col0, col1);
            set_lane<R32>(M0, (unsigned)e0);
            set_lane<R32 + 4>(M0, (unsigned)(e0 >> 32));
            set_lane<R32>(M1, (unsigned)e1);
            set_lane<R32 + 4>(M1, (unsigned)(e1 >> 32));
          });
          if (lane < 32) { mrow[lane * 4] = (unsigned)M0; mrow[lane * 4 + 1] = (unsigned)M1; }
        } else {
          // Per score: a multiply, "x < lo" (the bit, as a wave mask -> two v_writelane) and "x <= hi"; the scores
          // inside the bracket (le and not lt: one scalar and-not) are shifted into a per-lane bitmap (one v_addc).
          // Longer scalar chains on compare results (compare / select / or per score) stall the wave: measured.
          unsigned I0 = 0, I1 = 0;                               // per-lane bitmaps of "inside the bracket"
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            const float2 br = lds.lohi[wm * 64 + tm * 32 + R32 + 4 * lh];
            const float x0 = acc[tm][0][q] * sb0, x1 = acc[tm][1][q] * sb1;
            const unsigned long long lt0 = __ballot(x0 < br.x), lt1 = __ballot(x1 < br.x);
            shift_in(I0, __ballot(x0 <= br.y) & ~lt0);            // one scalar and-not per score, no chain
            shift_in(I1, __ballot(x1 <= br.y) & ~lt1);
            set_lane<R32>(M0, (unsigned)lt0);
            set_lane<R32 + 4>(M0, (unsigned)(lt0 >> 32));
            set_lane<R32>(M1, (unsigned)lt1);
            set_lane<R32 + 4>(M1, (unsigned)(lt1 >> 32));
          });
          if (lane < 32) { mrow[lane * 4] = (unsigned)M0; mrow[lane * 4 + 1] = (unsigned)M1; }
          const unsigned in0 = I0, in1 = I1;                     // score (q, tn) of this lane: bit 15 - q of in<tn>
          if (in0 | in1) {                                       // lanes owning a score inside a bracket: the exact
            static_for<0, 4>([&](auto gc) {                      // comparison, bit set in LDS; 8 scores per outer test
              constexpr int g4 = decltype(gc)::value;
              if ((in0 | in1) & (0xf000u >> (4 * g4))) {
                static_for<0, 8>([&](auto kc) {
                  constexpr int kk = decltype(kc)::value, q = 4 * g4 + (kk >> 1), tn = kk & 1, R32 = (q & 3) + 8 * (q >> 2);
                  if ((tn ? in1 : in0) & (0x8000u >> q)) {
                    const int rl = wm * 64 + tm * 32 + R32 + 4 * lh;
                    const float e = rank_sigmoid(acc[tm][tn][q] * (tn ? sb1 : sb0) * lds.sA[rl]), et = lds.eT[rl];
                    bool before = e < et;
                    if (e == et) {                               // equal losses pop in id order
                      const int64_t col = tn ? col1 : col0;
                      before = (col < K ? cand[col] : -1) < lds.tI[rl];
                    }
                    if (before) atomicOr(mrow + (R32 + 4 * lh) * 4 + tn, 1u << li);
                  }
                });
              }
            });
          }
        }
      }
      __syncthreads();
      if (t < kRB) {
        const unsigned* m = lds.bm + t * 4;
        raw_reg += __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
      }
      if (known_off) {
        for (int32_t e = kn0 + t; e < kn1; e += kBlock) {
          const unsigned rc = known_rc[e];
          const int rl = rc >> 7, cl = rc & 127;
          if ((lds.bm[rl * 4 + (cl >> 5)] >> (cl & 31)) & 1u) atomicAdd(&lds.skip[rl], 1);
        }
      }
      // no barrier: the next tile's first write to bm / sB comes after its own barriers
    }
    __syncthreads();
    if (MODE != 2 && t < kRB && m0 + t < B) {
      if (raw_reg) atomicAdd(&raw_cnt[m0 + t], raw_reg);
      if (lds.skip[t]) atomicAdd(&skip_cnt[m0 + t], lds.skip[t]);
    }
  }
}

int pipe_cu_count() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  return cus;
}

template <int CW>
int pipe_launch_cw(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                   const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                   const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss, float* scores_out,
                   int spec, int scores_only, int sweep_flags, hipStream_t st) {
  const size_t lds = pipe_lds_bytes<CW>(d);
  if (lds > 160 * 1024) return GE_ENOTSUP;
  const int64_t n_rb = (B + kRB - 1) / kRB, n_ct = (K + kRB - 1) / kRB;
  if (n_ct > INT32_MAX / 2 || n_rb > INT32_MAX / 2) return GE_ENOTSUP;
  const int64_t n_tiles = n_rb * n_ct;
  const int64_t grid = std::min<int64_t>(n_tiles, GE_PIPE_GRID_M * (int64_t)pipe_cu_count());
  auto go = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlock), lds, st, table, N, d, hr, B, true_id, cand, K, max_norm,
                       cand_is_head, known_off, known_rc, raw_cnt, skip_cnt, true_loss, scores_out, (int)n_ct, n_tiles, spec, sweep_flags);
    return launch_status();
  };
  if (scores_only) return (CW == 40 && d == 200) ? go(rank_pipe_kernel<CW, (CW == 40 ? 5 : 0), 2>) : go(rank_pipe_kernel<CW, 0, 2>);
  if (scores_out) return go(rank_pipe_kernel<CW, 0, 1>);
  if (CW == 40 && d == 200) return go(rank_pipe_kernel<CW, (CW == 40 ? 5 : 0), 0>);   // the FB15k configuration, unrolled
  return go(rank_pipe_kernel<CW, 0, 0>);
}

}  // namespace

int sweep_pipe_launch(const float*, int64_t, int32_t, const int32_t*, int64_t, const int32_t*, const int32_t*, int64_t, float,
                      int, const int32_t*, const uint16_t*, int32_t*, int32_t*, float*, float*, int, int, int, const void*, hipStream_t);

int rank_pipe_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                     const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                     const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                     float* scores_out, int spec, const void* planes_ws, hipStream_t st, int vs_loss) {
  return sweep_pipe_launch(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc, raw_cnt,
                           skip_cnt, true_loss, scores_out, spec, 0, vs_loss ? 2 : 0, planes_ws, st);
}

// ge_complex_score_1vK on the same pipeline: out [B,K] = score (sigmoid when apply_sigmoid)
int score_pipe_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* cand,
                      int64_t K, float max_norm, int apply_sigmoid, int cand_is_head, float* out, hipStream_t st) {
  return sweep_pipe_launch(table, N, d, hr, B, nullptr, cand, K, max_norm, cand_is_head, nullptr, nullptr, nullptr, nullptr,
                           nullptr, out, 0, 1, apply_sigmoid ? 1 : 0, nullptr, st);
}

int sweep_pipe_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                      const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                      const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                      float* scores_out, int spec, int scores_only, int sweep_flags, const void* planes_ws, hipStream_t st) {
  {   // embedding_dim % 8 == 0 in 56 ... 288, max_norm <= 8: the split-precision sweep (ge_rank_f16.hip)
    const int rc = sweep_f16_launch(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc, raw_cnt,
                                    skip_cnt, true_loss, scores_out, spec, scores_only, sweep_flags, planes_ws, st);
    if (rc != GE_ENOTSUP) return rc;
  }
#define GE_PIPE(CW)                                                                                                  \
  return pipe_launch_cw<CW>(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc, raw_cnt, \
                            skip_cnt, true_loss, scores_out, spec, scores_only, sweep_flags, st)
  if (d % 40 == 0) GE_PIPE(40);
  if (d % 32 == 0) GE_PIPE(32);
  if (d % 24 == 0) GE_PIPE(24);
#undef GE_PIPE
  return GE_ENOTSUP;
}

}  // namespace ge
