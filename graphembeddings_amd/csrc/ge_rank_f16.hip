// ge_rank_f16.hip -- the split-precision link-prediction sweep (holE.py:427-472, 564-575; semantics in ge_rank.hip)
// with TWO waves per SIMD.
//
// tools/probes/mfma_gap_probe.hip (profiles/r03_mfma_gap_probe.txt) says why: with one wave per SIMD the shadow of a
// v_mfma_f32_32x32x16_f16 hides four or five INDEPENDENT VALU instructions and next to nothing of the rank epilogue's
// compare -> scalar -> v_addc / v_writelane chains (12 such instructions cost 58-68 cycles wherever they are put,
// between the MFMAs or behind them), so cutting the epilogue into the gaps of the next tile gained nothing (measured).
// A second wave on the SIMD runs its MFMAs under the first one's epilogue and staging: the same instruction mix goes
// through 1.45-1.5 x faster.  So: 512 threads per workgroup, eight waves of 64 x 32 scores each on a 128 x 128 tile,
// <= 256 registers a wave.
//
// x * 2^8 = hi + mid with two fp16 values (round toward zero, so mid has hi's sign) is exact to 22 bits, and
//     q . t  =  2^-16 (qh.th + qh.tm + qm.th)  +  O(2^-22) per product
// accumulated in fp32: three f16 MFMAs per 16-wide k block.  Q (pre-multiplied by the rows' clip scales) sits in LDS
// as two fp16 planes for the whole row block; a candidate row (held whole in registers a tile ahead, so its clip scale
// is known before anything is stored) is split chunk by chunk (32 columns) as it is stored to LDS.
// The kernel is compiled per number of k blocks (embedding_dim 64 ... 208, any multiple of 8); embedding_dim itself is
// a run-time value (the clamp of the row's last requests, the zero fill behind the row).
// Epilogue: as ge_rank_pipe.hip -- raw scores against a bracket of the true candidate's raw score, bits into a
// row-major bitmap by v_writelane, the exact fp32 comparison only for scores inside the bracket.
#include <algorithm>
#include <type_traits>

#include "ge_rank_dev.h"

#ifndef GE_PIPE_GRID_M
#define GE_PIPE_GRID_M 2   // workgroups per CU (each CU holds one at a time): equal shares, two rounds
#endif

namespace ge {
namespace {

#ifdef GE_RANK_STAMPS   // diagnostic build only (tools/probes/rank_phase_probe.py): cycles per phase, summed over waves
__device__ unsigned long long g_rank_stamps[16];
#define GE_STAMP(i, t_prev) do { const long long now_ = __builtin_readcyclecounter(); st_[i] += now_ - (t_prev); (t_prev) = now_; } while (0)
#else
#define GE_STAMP(i, t_prev) do { } while (0)
#endif

constexpr int kBlk = 512;               // 8 waves: wm = w >> 2 (64 rows), wn = w & 3 (32 candidates)

template <int I0, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, N>(f);
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KKB>
struct HCfg {
  static constexpr int kKB = KKB;                   // k blocks of 16 (the last zero padded behind embedding_dim)
  static constexpr int kChunks = (KKB + 1) / 2;     // staged chunks of two k blocks = register slots of 32 reals
  static constexpr int kSA = 16 * KKB + 8;          // halves per Q row (+16 bytes: ds_read_b128 of 32 rows hits 32 bank groups)
  static_assert(KKB >= 4 && KKB <= 13, "embedding_dim 64 ... 208 (LDS: Q planes + two chunk buffers)");
};
constexpr int kSB = 32 + 8;             // halves per candidate chunk row
constexpr float kQScale = 256.f;        // both operands: |q|, |t * clip| <= max_norm^2 resp. max_norm sqrt(d/2)

struct HLds {
  _Float16* Ah;    // [kRB][kSA] high halves of Q * 2^8 ...
  _Float16* Am;    //   ... and the remainders (Q * 2^8 = Ah + Am to 22 bits)
  _Float16* Bp;    // [2 buffers][2 planes][kRB][kSB] candidate chunk * clip * 2^8, high halves | remainders
  float* sA;       // [kRB] 2^-16 (NaN: bad id / beyond B)
  float* eT;       // [kRB] loss of the true candidate
  float2* lohi;    // [kRB] raw-score bracket of the true candidate
  unsigned* bm;    // [kRB][4] `pops before` bits of the current tile
  int* skip;       // [kRB] known-true candidates ranked before the target
  int* tI;         // [kRB] entity id of the true candidate (-1 beyond B)
};

template <int KKB>
constexpr size_t h_lds_bytes() {
  return sizeof(_Float16) * ((size_t)2 * kRB * HCfg<KKB>::kSA + 2 * 2 * kRB * kSB) + sizeof(float) * 2 * kRB +
         sizeof(float2) * kRB + sizeof(unsigned) * kRB * 4 + sizeof(int) * 2 * kRB;
}

struct HOps { h8 ah[2], am[2], bh, bm; };           // one k block of this wave's 64 x 32 block

__device__ __forceinline__ void h_split(float x0, float x1, h2& hi, h2& mid) {
  typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
  const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(h2, h);
  const fp16x2 m = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi.x, x1 - (float)hi.y);
  mid = __builtin_bit_cast(h2, m);
}

// v_writelane_b32 with a constant lane: lane `LANE` of m = the wave-uniform v
template <int LANE>
__device__ __forceinline__ void set_lane(int& m, unsigned v) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(v), "n"(LANE));
}
// m = 2 * m + (this lane's bit of the wave mask): one v_addc_co_u32 with the mask as carry-in
__device__ __forceinline__ void shift_in(unsigned& m, unsigned long long mask) {
  unsigned long long carry_out;
  asm("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(m), "=s"(carry_out) : "s"(mask));
}

// One score of the bracket epilogue, as ONE instruction sequence (the compiler's hazard recognizer does not look
// inside inline asm: on gfx950 a VALU read of an SGPR that a VALU wrote needs two instructions in between, which the
// order below provides -- the v_writelane read vcc two instructions after the compare that wrote it):
//   inside = (x <= hi) & ~(x < lo) shifted into the per-lane bitmap I; the wave mask of x < lo into lanes R32 / R32 + 4 of M
template <int R32>
__device__ __forceinline__ void bracket_item(float x, float2 br, int& M, unsigned& I) {
  unsigned long long tmp;
  asm("v_cmp_le_f32_e64 %2, %3, %5\n\t"
      "v_cmp_lt_f32_e32 vcc, %3, %4\n\t"
      "s_andn2_b64 %2, %2, vcc\n\t"
      "v_addc_co_u32_e64 %1, %2, %1, %1, %2\n\t"
      "v_writelane_b32 %0, vcc_lo, %6\n\t"
      "v_writelane_b32 %0, vcc_hi, %7"
      : "+v"(M), "+v"(I), "=&s"(tmp)
      : "v"(x), "v"(br.x), "v"(br.y), "n"(R32), "n"(R32 + 4)
      : "vcc");
}

// A candidate row in flight: slot c = columns 32 c + 8 qt ... + 7 of the row (this thread's quarter of chunk c), and
// the two real bins of a spectral row.  Requests are clamped into the row, never predicated.
template <int KKB>
struct HRow {
  float4 r[HCfg<KKB>::kChunks][2];
  float x_dc, x_ny;
};

template <int KKB>
__device__ __forceinline__ void h_fetch_slot(const float* __restrict__ row, int d, int qt, int c, float4 (&r)[2]) {
  const int col = min(c * 32 + qt * 8, d - 8);
  r[0] = *reinterpret_cast<const float4*>(row + col);
  r[1] = *reinterpret_cast<const float4*>(row + col + 4);
}

// this thread's 8 reals of slot C, times `scale` (0 behind the row), split to the two planes of LDS buffer `buf`
template <int KKB, int C>
__device__ __forceinline__ void h_stash(const HLds& lds, int d, int srow, int qt, int buf, const float4 (&r)[2], float scale) {
  constexpr bool may_end = C * 32 + 32 > 16 * (KKB - 1);          // the row may end inside this slot
  float x[8] = {r[0].x, r[0].y, r[0].z, r[0].w, r[1].x, r[1].y, r[1].z, r[1].w};
  const bool in = !may_end || C * 32 + qt * 8 < d;                // (embedding_dim % 8 == 0: all eight or none)
  h8 hi, mid;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h2 a, b;
    h_split(in ? x[2 * i] * scale : 0.f, in ? x[2 * i + 1] * scale : 0.f, a, b);
    hi[2 * i] = a.x; hi[2 * i + 1] = a.y; mid[2 * i] = b.x; mid[2 * i + 1] = b.y;
  }
  *reinterpret_cast<h8*>(lds.Bp + ((buf * 2 + 0) * kRB + srow) * kSB + qt * 8) = hi;
  *reinterpret_cast<h8*>(lds.Bp + ((buf * 2 + 1) * kRB + srow) * kSB + qt * 8) = mid;
}

// piece i (0..5) of the operands of k block `kb`: ah0 ah1 am0 am1 bh bm
template <int KKB>
__device__ __forceinline__ void h_ops_piece(HOps& o, const HLds& lds, int wm, int wn, int li, int lh, int kb, int i) {
  constexpr int kSA = HCfg<KKB>::kSA;
  if (i < 4) {
    const _Float16* ap = (i & 2 ? lds.Am : lds.Ah) + (wm * 64 + (i & 1) * 32 + li) * kSA + kb * 16 + lh * 8;
    if (i & 2) o.am[i & 1] = *reinterpret_cast<const h8*>(ap); else o.ah[i & 1] = *reinterpret_cast<const h8*>(ap);
  } else {
    const int buf = (kb >> 1) & 1, within = kb & 1;
    const _Float16* bp = lds.Bp + ((buf * 2 + (i & 1)) * kRB + wn * 32 + li) * kSB + within * 16 + lh * 8;
    if (i & 1) o.bm = *reinterpret_cast<const h8*>(bp); else o.bh = *reinterpret_cast<const h8*>(bp);
  }
}

// One 128 x 128 tile.  R holds this tile's candidate row on entry and the NEXT tile's row (next_row) on exit: a slot
// is refilled right after it has been stored to LDS, a whole tile ahead of its use.
template <int KKB>
__device__ __forceinline__ void h_tile(const float* __restrict__ next_row, int d, bool bad, float max_norm, int spec,
                                       const HLds& lds, HRow<KKB>& R, f32x16 (&acc)[2], long long (&st_)[8], long long& tp_) {
  constexpr int kKB = KKB, kChunks = HCfg<KKB>::kChunks;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int srow = t >> 2, qt = t & 3;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  // The whole row is in registers, so its clip scale is known BEFORE anything is stored: the planes hold
  // t * clip(t) * 2^8 (|t clip| <= max_norm, or max_norm sqrt(d/2) for one bin of a spectral row: no fp16 overflow
  // for max_norm <= 8 whatever the table holds) and the epilogue needs no column scale.
  f2 ss2 = {0.f, 0.f};
  static_for<0, kChunks>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr bool may_end = c * 32 + 32 > 16 * (KKB - 1);
    const bool in = !may_end || c * 32 + qt * 8 < d;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const f2 xy = in ? f2{R.r[c][v].x, R.r[c][v].y} : f2{0.f, 0.f}, zw = in ? f2{R.r[c][v].z, R.r[c][v].w} : f2{0.f, 0.f};
      ss2 = __builtin_elementwise_fma(xy, xy, ss2);
      ss2 = __builtin_elementwise_fma(zw, zw, ss2);
    }
  });
  float ss = ss2.x + ss2.y;
  ss += __shfl_xor(ss, 1, kWave);
  ss += __shfl_xor(ss, 2, kWave);
  // spectral HolE rows: |x|^2 = (2 sum - X_0^2 - X_k^2) / d; the two real bins sit at columns 0 and d/2
  if (spec) ss = (2.f * ss - R.x_dc * R.x_dc - R.x_ny * R.x_ny) / (float)d;
  float inv;
  const float scale = bad ? __builtin_nanf("") : clip_scale(ss, max_norm, inv) * kQScale;
  h_stash<KKB, 0>(lds, d, srow, qt, 0, R.r[0], scale);
  R.x_dc = next_row[0];
  R.x_ny = next_row[d >> 1];
  GE_STAMP(0, tp_);
  __syncthreads();
  GE_STAMP(1, tp_);
  HOps ops[2];
#pragma unroll
  for (int i = 0; i < 6; ++i) h_ops_piece<KKB>(ops[0], lds, wm, wn, li, lh, 0, i);
  static_for<0, kKB>([&](auto kbc) {
    constexpr int kb = decltype(kbc)::value, qc = kb >> 1, within = kb & 1;
    constexpr bool last_of_chunk = within == 1 || kb == kKB - 1;
    constexpr int nslot = qc + 1 < kChunks ? qc + 1 : 0;          // the slot stored / refilled beside chunk qc
    if (last_of_chunk && qc + 1 < kChunks) __syncthreads();       // chunk qc+1 is in LDS; chunk qc-1's buffer is free
    HOps& cur = ops[kb & 1];
    HOps& nxt = ops[(kb + 1) & 1];
    static_for<0, 6>([&](auto pc) {
      constexpr int p = decltype(pc)::value, ty = p >> 1, tm = p & 1;   // consecutive MFMAs hit different accumulators
      const h8 a = ty == 2 ? cur.am[tm] : cur.ah[tm];
      const h8 b = ty == 1 ? cur.bm : cur.bh;
      acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[tm], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (kb + 1 < kKB) h_ops_piece<KKB>(nxt, lds, wm, wn, li, lh, kb + 1, p);
      if constexpr (!last_of_chunk) {                             // first k block of a chunk: store chunk qc+1
        if constexpr (p == 3 && qc + 1 < kChunks) h_stash<KKB, nslot>(lds, d, srow, qt, (qc + 1) & 1, R.r[nslot], scale);
      } else if constexpr (p >= 4) {                              // last k block: refill the slot just stored
        R.r[nslot][p - 4] = *reinterpret_cast<const float4*>(next_row + min(nslot * 32 + qt * 8, d - 8) + 4 * (p - 4));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  GE_STAMP(2, tp_);
  __syncthreads();                              // the bitmask / bracket arrays of the epilogue are free again
  GE_STAMP(3, tp_);
}

// MODE 0: ranks.  1: ranks, every loss computed exactly and stored too (tests).  2: no ranking at all -- the sweep
// writes scores_out[B,K] (raw score, or its sigmoid when `sweep_flags` & 1): ge_complex_score_1vK on this pipeline.
template <int KKB, int MODE>
__global__ __launch_bounds__(kBlk) void rank_f16_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ true_id, const int32_t* __restrict__ cand, int64_t K, float max_norm,
    int cand_is_head, const int32_t* __restrict__ known_off, const uint16_t* __restrict__ known_rc,
    int32_t* __restrict__ raw_cnt, int32_t* __restrict__ skip_cnt, float* __restrict__ true_loss,
    float* __restrict__ scores_out, int n_ct, int64_t n_tiles, int spec, int sweep_flags) {
  constexpr bool SCORES = MODE == 1;
  constexpr int kChunks = HCfg<KKB>::kChunks, kSA = HCfg<KKB>::kSA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  HLds lds;
  lds.Ah = reinterpret_cast<_Float16*>(smem);
  lds.Am = lds.Ah + kRB * kSA;
  lds.Bp = lds.Am + kRB * kSA;
  lds.sA = reinterpret_cast<float*>(lds.Bp + 2 * 2 * kRB * kSB);
  lds.eT = lds.sA + kRB;
  lds.lohi = reinterpret_cast<float2*>(lds.eT + kRB);             // an even number of floats in: 8-byte aligned
  lds.bm = reinterpret_cast<unsigned*>(lds.lohi + kRB);
  lds.skip = reinterpret_cast<int*>(lds.bm + kRB * 4);
  lds.tI = lds.skip + kRB;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = t >> 2, qt = t & 3;
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

    // ---- Q = fixed o relation for the block's 128 rows (four threads a row), scaled by the rows' clip scales and
    // 2^8, split into two fp16 planes
    {
      const int64_t r = m0 + srow;
      int32_t fid = -1, rid = -1;
      if (r < B) { fid = hr[2 * r]; rid = hr[2 * r + 1]; }
      const bool bad = fid < 0 || fid >= N || rid < 0 || rid >= N;
      const float* frow = table + (int64_t)(bad ? 0 : fid) * d;
      const float* rrow = table + (int64_t)(bad ? 0 : rid) * d;
      float ssf = 0.f, ssr = 0.f;
      // spectral HolE (ge_complex_dev.h): Hermitian weight 2 on every bin but element 0, which packs the two REAL
      // bins X_0 | X_k; norms and score carry the Parseval factor 1/d
      for (int j = qt; j < (k >> 2); j += 4) {                   // pass 1: the two clip norms
        const float4 fre = *reinterpret_cast<const float4*>(frow + 4 * j), fim = *reinterpret_cast<const float4*>(frow + k + 4 * j);
        const float4 rre = *reinterpret_cast<const float4*>(rrow + 4 * j), rim = *reinterpret_cast<const float4*>(rrow + k + 4 * j);
        const float w0 = (spec && j != 0) ? 2.f : 1.f, w1 = spec ? 2.f : 1.f;     // element 0 of the row / the others
        ssf += w0 * (fre.x * fre.x + fim.x * fim.x) + w1 * (fre.y * fre.y + fre.z * fre.z + fre.w * fre.w + fim.y * fim.y + fim.z * fim.z + fim.w * fim.w);
        ssr += w0 * (rre.x * rre.x + rim.x * rim.x) + w1 * (rre.y * rre.y + rre.z * rre.z + rre.w * rre.w + rim.y * rim.y + rim.z * rim.z + rim.w * rim.w);
      }
      ssf += __shfl_xor(ssf, 1, kWave); ssf += __shfl_xor(ssf, 2, kWave);
      ssr += __shfl_xor(ssr, 1, kWave); ssr += __shfl_xor(ssr, 2, kWave);
      float i0, i1;
      const float inv_d = spec ? 1.0f / (float)d : 1.0f;
      // The planes hold q * (clip scales) * (1/d for a spectral table) * 2^8.  ComplEx: |q sa| <= 2 max_norm^2.  A spectral
      // row's clip bounds its Parseval-weighted norm, so ONE bin may reach max_norm sqrt(d/2) and a Hermitian-weighted
      // product d max_norm^2: the 1/d of the correlation theorem is folded in BEFORE the split (|q sa / d| <= max_norm^2),
      // which keeps every plane entry below 2^8 * 64 for max_norm <= 8 whatever the table holds.
      const float sa = clip_scale(ssf * inv_d, max_norm, i0) * clip_scale(ssr * inv_d, max_norm, i1) * inv_d * kQScale;
      _Float16* ah = lds.Ah + srow * kSA;
      _Float16* am = lds.Am + srow * kSA;
      for (int j = qt; j < (k >> 2); j += 4) {                   // pass 2: q * sa * 2^8 -> high halves and remainders
        const float4 fre = *reinterpret_cast<const float4*>(frow + 4 * j), fim = *reinterpret_cast<const float4*>(frow + k + 4 * j);
        const float4 rre = *reinterpret_cast<const float4*>(rrow + 4 * j), rim = *reinterpret_cast<const float4*>(rrow + k + 4 * j);
        const float fr[4] = {fre.x, fre.y, fre.z, fre.w}, fi[4] = {fim.x, fim.y, fim.z, fim.w};
        const float rr[4] = {rre.x, rre.y, rre.z, rre.w}, ri[4] = {rim.x, rim.y, rim.z, rim.w};
        float qre[4], qim[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool packed = spec && j == 0 && i == 0;
          if (packed) {          // two independent real dimensions: products of the re slots and of the im slots
            qre[i] = fr[i] * rr[i];
            qim[i] = fi[i] * ri[i];
          } else if (!cand_is_head) {   // q = h * r ; score = Re(q conj t)
            qre[i] = fr[i] * rr[i] - fi[i] * ri[i];
            qim[i] = fr[i] * ri[i] + fi[i] * rr[i];
          } else {               // Re(h r conj t) with h the candidate: Q = [Re(r conj t) | -Im(r conj t)]
            qre[i] = rr[i] * fr[i] + ri[i] * fi[i];
            qim[i] = -(ri[i] * fr[i] - rr[i] * fi[i]);
          }
          if (spec && !packed) { qre[i] *= 2.f; qim[i] *= 2.f; }   // Hermitian weight
        }
        h4 rh, rm, ih, im;
#pragma unroll
        for (int i = 0; i < 4; i += 2) {
          h2 a, b;
          h_split(qre[i] * sa, qre[i + 1] * sa, a, b);
          rh[i] = a.x; rh[i + 1] = a.y; rm[i] = b.x; rm[i + 1] = b.y;
          h_split(qim[i] * sa, qim[i + 1] * sa, a, b);
          ih[i] = a.x; ih[i + 1] = a.y; im[i] = b.x; im[i + 1] = b.y;
        }
        *reinterpret_cast<h4*>(ah + 4 * j) = rh; *reinterpret_cast<h4*>(am + 4 * j) = rm;          // (row stride, k: multiples of 4)
        *reinterpret_cast<h4*>(ah + k + 4 * j) = ih; *reinterpret_cast<h4*>(am + k + 4 * j) = im;
      }
      if (qt == 0) {
        for (int c = d; c < 16 * KKB; ++c) { ah[c] = (_Float16)0.f; am[c] = (_Float16)0.f; }       // k padding
        lds.sA[srow] = (bad || r >= B) ? __builtin_nanf("") : 1.0f / (kQScale * kQScale);
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
    auto row_of = [&](int32_t id) -> const float* {               // (bad ids read row 0; their clip scale is NaN)
      return table + (int64_t)((id < 0 || id >= N) ? 0 : id) * d;
    };
    auto fetch_row = [&](const float* row, HRow<KKB>& R) {
#pragma unroll
      for (int c = 0; c < kChunks; ++c) h_fetch_slot<KKB>(row, d, qt, c, R.r[c]);
      R.x_dc = row[0];
      R.x_ny = row[d >> 1];
    };
    f32x16 acc[2];
    HRow<KKB> R;
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp_ = 0;
    (void)st_; (void)tp_;
    // ---- the true candidates: a tile whose candidate rows are this block's 128 true entities
    if constexpr (MODE != 2) {
      const int32_t tid = lds.tI[srow];
      fetch_row(row_of(tid), R);
      h_tile<KKB>(row_of(cand_of(ct0)), d, tid < 0 || tid >= N, max_norm, spec, lds, R, acc, st_, tp_);   // leaves the first tile's row in R
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
          if (rl == wn * 32 + li) lds.eT[rl] = acc[tm][q];        // raw score, row scale still to come
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
        const float xp = lds.eT[t], sa = lds.sA[t];
        const float xs = xp * sa, e = rank_sigmoid(xs), gs = e * (1.0f - e);
        const float wx = gs < 1e-5f ? __builtin_inff() : 1e-6f / gs + 4e-7f * fabsf(xs);
        const float wq = wx / sa;
        lds.lohi[t] = make_float2(xp - wq, xp + wq);
        lds.eT[t] = e;
        if (true_loss && ct0 == 0 && m0 + t < B) true_loss[m0 + t] = e;
      }
      __syncthreads();
    }
    int raw_reg = 0;

    // ---- the sweep over this share's candidate tiles of the row block
    // candidate ids and known-cell ranges are requested one tile ahead of their use: nothing ever waits on them
    int32_t cid = cand_of(ct0), cid_next = cand_of(ct0 + 1), kn0, kn1, kn0_next, kn1_next;
    known_of(ct0, kn0_next, kn1_next);
    if constexpr (MODE == 2) fetch_row(row_of(cid), R);          // no diagonal tile ran: the first row is not in R yet
#ifdef GE_RANK_STAMPS
    for (int i = 0; i < 8; ++i) st_[i] = 0;
    tp_ = __builtin_readcyclecounter();
#endif
    for (int ct = ct0; ct < ct1; ++ct) {
      const int64_t n0 = (int64_t)ct * kRB;
      h_tile<KKB>(row_of(cid_next), d, cid < 0 || cid >= N, max_norm, spec, lds, R, acc, st_, tp_);
      cid = cid_next; kn0 = kn0_next; kn1 = kn1_next;
      cid_next = cand_of(ct + 2);
      known_of(ct + 1, kn0_next, kn1_next);
      // epilogue: C layout of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
      // A candidate beyond K or with a bad id has a NaN clip scale, a row beyond B a NaN bracket: no bit is set.
      const int cl = wn * 32 + li;
      const int64_t col = n0 + cl;
      if constexpr (MODE == 2) {                                 // scores only: 32 consecutive floats of a row per half-wave
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            const int64_t row = m0 + rl;
            float v = acc[tm][q] * lds.sA[rl];
            if (sweep_flags & 1) v = rank_sigmoid(v);            // 4 VALU, within 3e-7 of expf's
            if (row < B && col < K) scores_out[row * K + col] = v;
          }
        continue;
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        int M = 0;                                               // lane r: the 32 column bits of row r of the 32 x 32 block
        unsigned* mrow = lds.bm + (wm * 64 + tm * 32) * 4 + wn;
        if constexpr (SCORES) {                                  // tests: every loss exactly, and stored
          const int32_t c0 = col < K ? cand[col] : -1;
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            const int rl = wm * 64 + tm * 32 + R32 + 4 * lh;
            const float et = lds.eT[rl];
            const float e0 = rank_sigmoid(acc[tm][q] * lds.sA[rl]);
            const unsigned long long mk = __ballot(e0 < et) | __ballot(e0 == et && c0 < lds.tI[rl]);
            if (m0 + rl < B && col < K) scores_out[(m0 + rl) * K + col] = e0;
            set_lane<R32>(M, (unsigned)mk);
            set_lane<R32 + 4>(M, (unsigned)(mk >> 32));
          });
          if (lane < 32) mrow[lane * 4] = (unsigned)M;
        } else {
          // Per score: "x < lo" (the bit, as a wave mask -> two v_writelane) and "x <= hi"; the scores inside the bracket
          // (le and not lt: one scalar and-not) are shifted into a per-lane bitmap (one v_addc): bracket_item.  Longer
          // scalar chains on compare results (compare / select / or per score) stall the wave: measured.
          unsigned I = 0;                                        // per-lane bitmap of "inside the bracket"
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            bracket_item<R32>(acc[tm][q], lds.lohi[wm * 64 + tm * 32 + R32 + 4 * lh], M, I);   // (the planes carry the clip scale)
          });
          if (lane < 32) mrow[lane * 4] = (unsigned)M;
          if (I) {                                               // lanes owning a score inside a bracket: the exact
            static_for<0, 4>([&](auto gc) {                      // comparison, bit set in LDS; 4 scores per outer test
              constexpr int g4 = decltype(gc)::value;
              if (I & (0xf000u >> (4 * g4))) {
                static_for<0, 4>([&](auto kc) {
                  constexpr int q = 4 * g4 + decltype(kc)::value, R32 = (q & 3) + 8 * (q >> 2);
                  if (I & (0x8000u >> q)) {
                    const int rl = wm * 64 + tm * 32 + R32 + 4 * lh;
                    const float e = rank_sigmoid(acc[tm][q] * lds.sA[rl]), et = lds.eT[rl];
                    bool before = e < et;
                    if (e == et) before = (col < K ? cand[col] : -1) < lds.tI[rl];   // equal losses pop in id order
                    if (before) atomicOr(mrow + (R32 + 4 * lh) * 4, 1u << li);
                  }
                });
              }
            });
          }
        }
      }
      GE_STAMP(4, tp_);
      __syncthreads();
      GE_STAMP(5, tp_);
      // the tile's bitmap is complete in LDS: rows count their bits, known cells that rank before the target are tallied
      if (t < kRB) {
        const unsigned* m = lds.bm + t * 4;
        raw_reg += __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
      }
      if (known_off) {
        for (int32_t e = kn0 + t; e < kn1; e += kBlk) {
          const unsigned rc = known_rc[e];
          const int rl = rc >> 7, cl2 = rc & 127;
          if ((lds.bm[rl * 4 + (cl2 >> 5)] >> (cl2 & 31)) & 1u) atomicAdd(&lds.skip[rl], 1);
        }
      }
      GE_STAMP(6, tp_);
      // no barrier: the next tile's first write to bm comes after its own barriers
    }
#ifdef GE_RANK_STAMPS
    if (lane == 0) {
      for (int i = 0; i < 7; ++i) atomicAdd(&g_rank_stamps[i], (unsigned long long)st_[i]);
      atomicAdd(&g_rank_stamps[7], (unsigned long long)(ct1 - ct0));
    }
#endif
    __syncthreads();
    if (MODE != 2 && t < kRB && m0 + t < B) {
      if (raw_reg) atomicAdd(&raw_cnt[m0 + t], raw_reg);
      if (lds.skip[t]) atomicAdd(&skip_cnt[m0 + t], lds.skip[t]);
    }
  }
}

int f16_cu_count() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  return cus;
}

template <int KKB>
int f16_launch_kkb(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                   const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                   const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss, float* scores_out,
                   int spec, int scores_only, int sweep_flags, hipStream_t st) {
  const int64_t n_rb = (B + kRB - 1) / kRB, n_ct = (K + kRB - 1) / kRB;
  if (n_ct > INT32_MAX / 2 || n_rb > INT32_MAX / 2) return GE_ENOTSUP;
  const int64_t n_tiles = n_rb * n_ct;
  const int64_t grid = std::min<int64_t>(n_tiles, GE_PIPE_GRID_M * (int64_t)f16_cu_count());
  auto go = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlk), h_lds_bytes<KKB>(), st, table, N, d, hr, B, true_id, cand, K,
                       max_norm, cand_is_head, known_off, known_rc, raw_cnt, skip_cnt, true_loss, scores_out, (int)n_ct,
                       n_tiles, spec, sweep_flags);
    return launch_status();
  };
  if (scores_only) return go(rank_f16_kernel<KKB, 2>);
  if (scores_out) return go(rank_f16_kernel<KKB, 1>);
  return go(rank_f16_kernel<KKB, 0>);
}

}  // namespace

#ifdef GE_RANK_STAMPS
extern "C" int ge_debug_rank_stamps(unsigned long long* out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rank_stamps), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) {
    unsigned long long z[16] = {0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_rank_stamps), z, sizeof(z));
  }
  return (int)e;
}
#endif

// The split-precision sweep: embedding_dim % 8 == 0 in 56 ... 208 (k blocks 4 ... 13), max_norm <= 8
// (|q sa (1/d)| <= 2 max_norm^2, |t clip| <= max_norm sqrt(d/2): x 2^8 inside fp16).  GE_ENOTSUP otherwise.
int sweep_f16_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                     const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                     const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                     float* scores_out, int spec, int scores_only, int sweep_flags, hipStream_t st) {
  if (d % 8 != 0 || d < 56 || d > 208 || !(max_norm <= 8.f)) return GE_ENOTSUP;
  static_assert(h_lds_bytes<13>() <= 160 * 1024, "LDS of the largest instantiation");
#define GE_KKB(KKB)                                                                                                  \
  case KKB:                                                                                                          \
    return f16_launch_kkb<KKB>(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc,   \
                               raw_cnt, skip_cnt, true_loss, scores_out, spec, scores_only, sweep_flags, st)
  switch ((d + 15) / 16) {
    GE_KKB(4); GE_KKB(5); GE_KKB(6); GE_KKB(7); GE_KKB(8); GE_KKB(9); GE_KKB(10); GE_KKB(11); GE_KKB(12); GE_KKB(13);
    default: return GE_ENOTSUP;
  }
#undef GE_KKB
}

}  // namespace ge
