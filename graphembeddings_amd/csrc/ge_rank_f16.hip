// ge_rank_f16.hip -- the split-precision link-prediction sweep (holE.py:427-472, 564-575; semantics in ge_rank.hip):
// f16 MFMAs on pre-split candidate planes, eight free-running waves per workgroup (two per SIMD).
//
// Why this shape (tools/probes/mfma_gap_probe.hip -> profiles/r03_mfma_gap_probe.txt; s_memtime stamps per phase and
// ablated builds, recorded in DESIGN.md section 9):
//   * with one wave per SIMD the shadow of a v_mfma_f32_32x32x16_f16 hides four or five INDEPENDENT VALU instructions
//     and next to nothing of the rank epilogue's compare -> scalar -> v_addc / v_writelane chains: cutting the epilogue
//     into the gaps of the next tile's MFMAs gained nothing (measured).  A second wave on the SIMD hides it -- if it
//     is in the OTHER phase;
//   * eight waves in one barrier domain (all in the MFMA loop, then all in the epilogue) gained 9 %; two groups of
//     four waves half a barrier cycle apart 15 %: with twelve workgroup barriers per tile the MFMA pipe still idled
//     half the time (an ablated loop with nothing but MFMAs and barriers ran at 53 % of the pipe);
//   * so nothing in the sweep is shared between waves any more.  Each wave owns a 64 x 32 block of the 128 x 128 tile:
//     its candidate operands come straight from global memory (the planes are L2-resident: every CU walks the candidate
//     tiles in the same order at the same pace) into registers in MFMA layout, three k blocks ahead; its bits go to a
//     bitmap of its own; it counts its own rows and looks up its own known cells.  No barrier between the row
//     block's set-up and its end: the two waves of a SIMD drift apart and one's epilogue runs under the other's MFMAs;
//   * splitting a candidate row into fp16 planes (norm, clip scale, 2 x cvt_pkrtz per pair) was half of the sweep's
//     VALU work and was repeated for every block of 128 test rows: it is a pre-pass (rank_planes_launch) whose output
//     -- `planes`, 1 KiB per (32 candidates, 16 columns, plane), exactly one operand fetch of one wave -- the sweep
//     only loads.
//
// x * 2^8 = hi + mid with two fp16 values (round toward zero, so mid has hi's sign) is exact to 22 bits, and
//     q . t  =  2^-16 (qh.th + qh.tm + qm.th)  +  O(2^-22) per product
// accumulated in fp32: three f16 MFMAs per 16-wide k block.  Q (pre-multiplied by the rows' clip scales) sits in LDS
// as two fp16 planes for the whole row block.  The kernel is compiled per number of k blocks (embedding_dim 56 ... 288,
// any multiple of 8); embedding_dim itself is a run-time value.
// Epilogue: raw scores against a bracket of the true candidate's raw score, bits into a row-major bitmap by
// v_writelane, the exact fp32 comparison (id tie-break) only for scores inside the bracket -- the outcome equals
// comparing the sigmoids everywhere.
#include <algorithm>
#include <type_traits>

#include "ge_rank_dev.h"

#ifndef GE_PIPE_GRID_M
#define GE_PIPE_GRID_M 2   // workgroups per CU (each CU holds one at a time): equal shares, two rounds
#endif

namespace ge {
namespace {

constexpr int kBlk = 512;               // eight waves: wm = w >> 2 (64 rows), wn = w & 3 (32 candidates of the 128-wide tile)
constexpr int kSL = 32;                 // candidates per slice of `planes` = one wave's columns

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
  static constexpr int kChunks = (KKB + 1) / 2;     // 32-column pieces of a row (the pre-pass's unit)
  static constexpr int kSA = 16 * KKB + 8;          // halves per Q row (+16 bytes: ds_read_b128 of 32 rows hits 32 bank groups)
  static_assert(KKB >= 4 && KKB <= 18, "embedding_dim 56 ... 288 (LDS: the Q planes, 152 KB at 18 k blocks)");
};
constexpr float kQScale = 256.f;        // both operands: |q|, |t * clip| <= max_norm^2 resp. max_norm sqrt(d/2)
constexpr int kOpHalves = kSL * 16;     // one operand fetch of one wave in `planes`: [32 candidates][16 columns], 1 KiB
constexpr int kAhead = 3;               // k blocks between a candidate operand's request and its first MFMA

struct HLds {
  _Float16* Ah;    // [kRB][kSA] high halves of Q * 2^8 ...
  _Float16* Am;    //   ... and the remainders (Q * 2^8 = Ah + Am to 22 bits)
  float* sA;       // [kRB] 2^-16 (NaN: bad id / beyond B)
  float* eT;       // [kRB] loss of the true candidate
  float2* lohi;    // [kRB] raw-score bracket of the true candidate
  unsigned* bm;    // this wave's [64] rows x 32 `pops before` bits of its current block
  int* skip;       // [kRB] known-true candidates ranked before the target
  int* extra;      // [kRB] candidates inside the bracket that the exact comparison put before the target
  int* tI;         // [kRB] entity id of the true candidate (-1 beyond B)
  int* tP;         // [kRB] its position among the candidates (-1: not a candidate)
  int* next;       // [4] per candidate slice wn: the next (tile, row half) block of the sweep not yet taken by a wave
};

template <int KKB>
constexpr size_t h_lds_bytes() {
  return sizeof(_Float16) * ((size_t)2 * kRB * HCfg<KKB>::kSA) + sizeof(float) * 2 * kRB + sizeof(float2) * kRB +
         sizeof(unsigned) * 8 * 64 + sizeof(int) * (4 * kRB + 4);
}

struct HA { h8 ah[2], am[2]; };          // the Q operands of one k block of this wave's 64 rows
struct HB { h8 bh, bm; };                // the candidate operands of one k block of this wave's 32 columns

__device__ __forceinline__ void h_split(float x0, float x1, h2& hi, h2& mid) {
  typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
  const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(h2, h);
  const fp16x2 m = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi.x, x1 - (float)hi.y);
  mid = __builtin_bit_cast(h2, m);
}

// v_writelane_b32 with a constant lane: lane `LANE` of m = the wave-uniform v (v must not come straight out of a
// VALU compare: see bracket_item)
template <int LANE>
__device__ __forceinline__ void set_lane(int& m, unsigned v) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(v), "n"(LANE));
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

// the candidate operands of k block kb; off = byte offset in `planes` of this lane's 16 bytes of the slice's k block 0, high
// plane -- a wave-uniform base and a 32-bit lane offset: the loads take the scalar-base form and the stride over the k
// blocks costs one 32-bit add per load, not a 64-bit add with its carry chain
__device__ __forceinline__ void h_loadB(HB& b, const _Float16* __restrict__ planes, unsigned off, int kb) {
  const char* base = reinterpret_cast<const char*>(planes);
  b.bh = *reinterpret_cast<const h8*>(base + (off + (unsigned)(kb * 2 * kOpHalves * 2)));
  b.bm = *reinterpret_cast<const h8*>(base + (off + (unsigned)((kb * 2 + 1) * kOpHalves * 2)));
}

// piece i (0..3) of the Q operands of k block `kb`, in the order the MFMAs of that k block first need them: ah0 ah1 am0 am1
template <int KKB>
__device__ __forceinline__ void h_opsA(HA& o, const HLds& lds, int wm, int li, int lh, int kb, int i) {
  constexpr int kSA = HCfg<KKB>::kSA;
  const int tm = i & 1, mid = i >> 1;
  const _Float16* ap = (mid ? lds.Am : lds.Ah) + (wm * 64 + tm * 32 + li) * kSA + kb * 16 + lh * 8;
  if (mid) o.am[tm] = *reinterpret_cast<const h8*>(ap); else o.ah[tm] = *reinterpret_cast<const h8*>(ap);
}

// The MFMA loop of one 64 x 32 block.  B[0 .. kAhead - 1] hold the candidate operands of k blocks 0 .. kAhead - 1 of the
// slice `cur` on entry and of the slice `nxt` on exit (a ring of kAhead + 1 register sets; a k block's operands are
// requested kAhead k blocks -- 18 MFMAs -- before its first MFMA, across the block boundary too).  No barrier.
template <int KKB>
__device__ __forceinline__ void h_mfma_loop(const HLds& lds, const _Float16* __restrict__ planes, unsigned cur, unsigned nxt,
                                            HB (&B)[kAhead + 1], f32x16 (&acc)[2], int wm, int li, int lh) {
  constexpr int kKB = KKB, kR = kAhead + 1;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
  HA ops[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) h_opsA<KKB>(ops[0], lds, wm, li, lh, 0, i);
  static_for<0, kKB>([&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    HA& ca = ops[kb & 1];
    HA& na = ops[(kb + 1) & 1];
    HB& cb = B[kb % kR];
    static_for<0, 6>([&](auto pc) {
      constexpr int p = decltype(pc)::value, ty = p >> 1, tm = p & 1;   // consecutive MFMAs hit different accumulators
      const h8 a = ty == 2 ? ca.am[tm] : ca.ah[tm];
      const h8 b = ty == 1 ? cb.bm : cb.bh;
      acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[tm], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (p < 4) {
        if constexpr (kb + 1 < kKB) h_opsA<KKB>(na, lds, wm, li, lh, kb + 1, p);
      } else if constexpr (p == 4) {                              // the ring slot of k block kb - 1 is free: k block kb + kAhead
        constexpr int kn = kb + kAhead;
        if constexpr (kn < kKB) h_loadB(B[kn % kR], planes, cur, kn);
        else h_loadB(B[kn % kR], planes, nxt, kn - kKB);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  // the next block's k blocks 0 .. kAhead - 1 sit in ring slots (kKB + j) % kR: move them to slots j (register renaming
  // at the loop's back edge; a few v_mov at most)
  HB t[kAhead];
#pragma unroll
  for (int j = 0; j < kAhead; ++j) t[j] = B[(kKB + j) % kR];
#pragma unroll
  for (int j = 0; j < kAhead; ++j) B[j] = t[j];
}

// MODE 0: ranks.  1: ranks, every loss computed exactly and stored too (tests).  2: no ranking at all -- the sweep
// writes scores_out[B,K] (raw score, or its sigmoid when `sweep_flags` & 1): ge_complex_score_1vK on this pipeline.
template <int KKB, int MODE>
__global__ __launch_bounds__(kBlk) void rank_f16_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ true_id, const int32_t* __restrict__ cand, int64_t K, float max_norm,
    int cand_is_head, const int32_t* __restrict__ known_off, const uint16_t* __restrict__ known_rc,
    int32_t* __restrict__ raw_cnt, int32_t* __restrict__ skip_cnt, float* true_loss,
    float* __restrict__ scores_out, int n_ct, int64_t n_tiles, int spec, int sweep_flags,
    const int32_t* __restrict__ pos_of, const _Float16* __restrict__ planes) {
  constexpr bool SCORES = MODE == 1;
  constexpr int kSA = HCfg<KKB>::kSA;
  constexpr int64_t kSliceHalves = (int64_t)KKB * 2 * kOpHalves;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int qt = t & 3;
  const int n_sl = 4 * n_ct;                                     // slices in `planes` (rows behind K: NaN)
  HLds lds;
  lds.Ah = reinterpret_cast<_Float16*>(smem);
  lds.Am = lds.Ah + kRB * kSA;
  lds.sA = reinterpret_cast<float*>(lds.Am + kRB * kSA);
  lds.eT = lds.sA + kRB;
  lds.lohi = reinterpret_cast<float2*>(lds.eT + kRB);             // an even number of floats in: 8-byte aligned
  lds.bm = reinterpret_cast<unsigned*>(lds.lohi + kRB) + w * 64;
  lds.skip = reinterpret_cast<int*>(reinterpret_cast<unsigned*>(lds.lohi + kRB) + 8 * 64);
  lds.extra = lds.skip + kRB;
  lds.tI = lds.extra + kRB;
  lds.tP = lds.tI + kRB;
  lds.next = lds.tP + kRB;
  const int k = d >> 1;

  // This workgroup's share of the (row block, 128-candidate tile) list, row-block major -- walked so that every
  // workgroup of the chip starts at candidate tile 0 and sweeps upwards at the same pace: the share's FIRST row block
  // (entered at some tile ct_a > 0) is taken last.  The CUs of an XCD then read the same tiles of `planes` within a
  // few tiles of each other and the XCD's 4 MiB L2 serves all but the first of them; walked in list order the 32 CUs
  // sat at 32 different places of the candidate ring, the planes (13 MB) streamed through every L2 and 80 % of the
  // reads missed it (TCC_HIT / TCC_MISS: 20 % -> 93 % hits).
  const int64_t share0 = n_tiles * blockIdx.x / gridDim.x, share1 = n_tiles * (blockIdx.x + 1) / gridDim.x;
  const int64_t first_end = min(share1, (share0 / n_ct + 1) * n_ct);
  for (int pass = 0; pass < 2; ++pass) {
  int64_t idx = pass ? share0 : first_end;
  const int64_t idx_end = pass ? first_end : share1;
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
      const int qrow = t >> 2;
      const int64_t r = m0 + qrow;
      int32_t fid = -1, rid = -1;
      if (r < B) { fid = hr[2 * r]; rid = hr[2 * r + 1]; }
      const bool bad = fid < 0 || fid >= N || rid < 0 || rid >= N;
      const float* frow = table + (int64_t)(bad ? 0 : fid) * d;
      const float* rrow = table + (int64_t)(bad ? 0 : rid) * d;
      // Two compact loops (not unrolled: straight-line code that runs once per row block is fetched cold -- about 300
      // cycles per 64 bytes of instructions, measured on the unrolled form of this staging and on a second copy of the
      // MFMA loop), each requesting the next iteration's four float4 before this iteration's arithmetic.
      const int nj = k >> 2;
      auto ld4 = [&](const float* row, int j) -> float4 { return *reinterpret_cast<const float4*>(row + 4 * min(j, nj - 1)); };
      float ssf = 0.f, ssr = 0.f;
      // spectral HolE (ge_complex_dev.h): Hermitian weight 2 on every bin but element 0, which packs the two REAL
      // bins X_0 | X_k; norms and score carry the Parseval factor 1/d
      {
        float4 nfre = ld4(frow, qt), nfim = ld4(frow + k, qt), nrre = ld4(rrow, qt), nrim = ld4(rrow + k, qt);
#pragma unroll 1
        for (int j = qt; j < nj; j += 4) {                       // pass 1: the two clip norms
          const float4 fre = nfre, fim = nfim, rre = nrre, rim = nrim;
          nfre = ld4(frow, j + 4); nfim = ld4(frow + k, j + 4); nrre = ld4(rrow, j + 4); nrim = ld4(rrow + k, j + 4);
          const float w0 = (spec && j != 0) ? 2.f : 1.f, w1 = spec ? 2.f : 1.f;   // element 0 of the row / the others
          ssf += w0 * (fre.x * fre.x + fim.x * fim.x) + w1 * (fre.y * fre.y + fre.z * fre.z + fre.w * fre.w + fim.y * fim.y + fim.z * fim.z + fim.w * fim.w);
          ssr += w0 * (rre.x * rre.x + rim.x * rim.x) + w1 * (rre.y * rre.y + rre.z * rre.z + rre.w * rre.w + rim.y * rim.y + rim.z * rim.z + rim.w * rim.w);
        }
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
      _Float16* ah = lds.Ah + qrow * kSA;
      _Float16* am = lds.Am + qrow * kSA;
      {
        float4 nfre = ld4(frow, qt), nfim = ld4(frow + k, qt), nrre = ld4(rrow, qt), nrim = ld4(rrow + k, qt);
#pragma unroll 1
        for (int j = qt; j < nj; j += 4) {                       // pass 2: q * sa * 2^8 -> high halves and remainders
          const float4 fre = nfre, fim = nfim, rre = nrre, rim = nrim;
          nfre = ld4(frow, j + 4); nfim = ld4(frow + k, j + 4); nrre = ld4(rrow, j + 4); nrim = ld4(rrow + k, j + 4);
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
      }
      if (qt == 0) {
        for (int c = d; c < 16 * KKB; ++c) { ah[c] = (_Float16)0.f; am[c] = (_Float16)0.f; }       // k padding
        lds.sA[qrow] = (bad || r >= B) ? __builtin_nanf("") : 1.0f / (kQScale * kQScale);
        lds.skip[qrow] = 0;
        lds.extra[qrow] = 0;
        const int32_t tid = (MODE != 2 && r < B) ? true_id[r] : -1;
        lds.tI[qrow] = tid;
        // (sweep_flags & 2, ranking against GIVEN losses: the "true candidate" pass still runs -- the loop below has one
        // copy of the MFMA code -- on candidate 0's planes, and its result is replaced by the given loss)
        lds.tP[qrow] = (sweep_flags & 2) ? (r < B ? 0 : -1) : (tid >= 0 && tid < N) ? pos_of[tid] : -1;
      }
      if (t < 4) lds.next[t] = 2;                                // (blocks 0 and 1 of a slice go to its two waves up front)
    }
    __syncthreads();

    // this wave's slices of `planes`: 32 candidates each, slice 4 ct + wn of the share's 128-candidate tiles
    auto slice_src = [&](int s) -> unsigned {                     // byte offset of this lane's 16 bytes of slice s's k block 0 (clamped)
      return (unsigned)min(s, n_sl - 1) * (unsigned)(kSliceHalves * 2) + (unsigned)(li * 32 + lh * 16);
    };
    auto known_of = [&](int ct, int32_t& k0, int32_t& k1) {
      k0 = k1 = 0;
      if (known_off && ct < ct1) {
        const int64_t tile = (int64_t)rb * n_ct + ct;
        k0 = known_off[tile]; k1 = known_off[tile + 1];
      }
    };
    f32x16 acc[2];
    HB Bq[kAhead + 1];
    // The sweep's blocks of candidate slice wn -- (tile, row half) = (ct0 + (i >> 1), i & 1), i < 2 (ct1 - ct0) -- are handed
    // out from a counter to the two waves that own the slice (w = wn and wn + 4: the two waves of one SIMD).  With a fixed
    // row half each, the older wave of the SIMD won every issue arbitration, finished 14 tiles early and waited 11 % of
    // the kernel at the closing barrier while its partner ran alone, MFMA loop and epilogue back to back (measured;
    // alternating s_setprio did not change it).  A wave holds two blocks: the one it computes and the one it prefetches
    // (the first is block wm, so a row block with a single tile still keeps both waves busy).
    const int n_items = 2 * (ct1 - ct0);
    auto take = [&]() -> int {
      int v = 0;
      if (lane == 0) v = atomicAdd(&lds.next[wn], 1);
      return __builtin_amdgcn_readfirstlane(v);
    };
    int item = wm, item_next = take();
    const int s0 = 4 * (ct0 + (item >> 1)) + wn;
    // ---- the first pass of the loop below (ranks): the true candidates -- a tile whose candidate rows are the block's 128
    // true entities (this wave: 32 of them, gathered by position), through the SAME copy of the MFMA loop as the sweep's
    // blocks (a second copy, fetched cold once per row block, took six tiles' time)
    bool diag = MODE != 2;
    unsigned cur = slice_src(s0);
    if constexpr (MODE != 2) {
      const int pos = lds.tP[wn * 32 + li];
      const int pc = pos < 0 ? 0 : pos;
      cur = (unsigned)(pc >> 5) * (unsigned)(kSliceHalves * 2) + (unsigned)((pc & 31) * 32 + lh * 16);
    }
#pragma unroll
    for (int j = 0; j < kAhead; ++j) h_loadB(Bq[j], planes, cur, j);
    int raw_reg[2][2] = {{0, 0}, {0, 0}};                         // lane r < 32: bits counted for row half*64 + tm*32 + r

    // ---- the sweep: behind the true-candidate pass no barrier until the row block is done
    int32_t kn0 = 0, kn1 = 0, kn0_next, kn1_next;
    known_of(ct0 + (item >> 1), kn0_next, kn1_next);
    while (diag || item < n_items) {
      const int ct = ct0 + (item >> 1), wmi = diag ? wm : (item & 1);
      const int64_t col = (int64_t)(4 * ct + wn) * kSL + li;     // this lane's candidate
      const unsigned nxt = slice_src(4 * (ct0 + ((diag ? item : item_next) >> 1)) + wn);
      if (!diag) {
        // (the next block's known-cell range is requested BEFORE the MFMA loop: it is a scalar load, and the wait in front
        // of the epilogue -- for the brackets -- waits for everything on that counter)
        kn0 = kn0_next; kn1 = kn1_next;
        known_of(ct0 + (item_next >> 1), kn0_next, kn1_next);
      }
      h_mfma_loop<KKB>(lds, planes, cur, nxt, Bq, acc, wmi, li, lh);     // leaves the next block's leading operands in Bq
      cur = nxt;
      if (diag) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (rl == wn * 32 + li) lds.eT[rl] = acc[tm][q];      // raw score, row scale still to come
          }
        __syncthreads();
        if (t < kRB) {
          // Bracket of the true candidate's raw score.  With g = e (1 - e) the sigmoid's slope at the true score and
          // w = 1e-6 / g <= 0.1, the slope anywhere inside [xs - w, xs + w] is >= g exp(-w) (the sigmoid is concave on one
          // side: a first-order bound alone is not enough), so a candidate whose scaled score lies outside has a loss that
          // differs by >= 0.9e-6, three times what the roundings of x * sA and of the 4-instruction sigmoid
          // (< 1.5e-7 each side) can move: outside the bracket the order of the losses is the order of the raw scores.
          // Near saturation (g < 1e-5, |score| > 11.5) no finite bracket gives that margin: it is infinite there and
          // every candidate of the row takes the exact comparison.  A true entity that is not among the candidates has
          // no rank: NaN bracket, NaN loss, no bit is ever set.
          float xp = lds.tP[t] < 0 ? __builtin_nanf("") : lds.eT[t];
          const float sa = lds.sA[t];
          float xs = xp * sa, e = rank_sigmoid(xs);
          if ((sweep_flags & 2) && lds.tP[t] >= 0) {
            // ranking against a GIVEN loss (ge_rank_1vK_vs_loss: the candidate it belongs to need not be in this list):
            // the bracket is centred on its logit -- rounded, but by orders of magnitude less than the bracket's
            // 1e-6 in loss units -- and the exact comparison inside the bracket is against the given value itself
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
        diag = false;
        continue;
      }
      item = item_next;
      item_next = take();
      // the brackets of this lane's 32 rows, requested together (read score by score -- a wait on the LDS queue in front
      // of every compare sequence -- the epilogue took 200 cycles per score; held in registers across the MFMA loop
      // they are spilled)
      float2 br[2][16];
      if constexpr (MODE == 0) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) br[tm][q] = lds.lohi[wmi * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh];
      }
      // epilogue: C layout of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  A candidate
      // beyond K or with a bad id has NaN planes, a row beyond B a NaN bracket: no bit is set.
      if constexpr (MODE == 2) {                                 // scores only: 32 consecutive floats of a row per half-wave
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int rl = wmi * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            const int64_t row = m0 + rl;
            float v = acc[tm][q] * lds.sA[rl];
            if (sweep_flags & 1) v = rank_sigmoid(v);            // 4 VALU, within 3e-7 of expf's
            if (row < B && col < K) scores_out[row * K + col] = v;
          }
        continue;
      }
      int32_t c0 = -1;
      if constexpr (SCORES) c0 = col < K ? cand[col] : -1;
      (void)c0;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        int M = 0;                                               // lane r: the 32 column bits of row r of the 32 x 32 block
        unsigned* mrow = lds.bm + tm * 32;
        if constexpr (SCORES) {                                  // tests: every loss exactly, and stored
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            const int rl = wmi * 64 + tm * 32 + R32 + 4 * lh;
            const float et = lds.eT[rl];
            const float e0 = rank_sigmoid(acc[tm][q] * lds.sA[rl]);
            const unsigned long long mk = __ballot(e0 < et) | __ballot(e0 == et && c0 < lds.tI[rl]);
            if (m0 + rl < B && col < K) scores_out[(m0 + rl) * K + col] = e0;
            set_lane<R32>(M, (unsigned)mk);                      // (mk comes out of a scalar OR: no VALU -> VALU SGPR hazard)
            set_lane<R32 + 4>(M, (unsigned)(mk >> 32));
          });
          raw_reg[0][tm] += wmi ? 0 : __popc((unsigned)M);
          raw_reg[1][tm] += wmi ? __popc((unsigned)M) : 0;
          if (lane < 32) mrow[lane] = (unsigned)M;
        } else {
          // Per score: "x < lo" (the bit, as a wave mask -> two v_writelane) and "x <= hi"; the scores inside the bracket
          // (le and not lt: one scalar and-not) are shifted into a per-lane bitmap (one v_addc): bracket_item.  Longer
          // scalar chains on compare results (compare / select / or per score) stall the wave: measured.
          unsigned I = 0;                                        // per-lane bitmap of "inside the bracket"
          static_for<0, 16>([&](auto qc) {
            constexpr int q = decltype(qc)::value, R32 = (q & 3) + 8 * (q >> 2);
            bracket_item<R32>(acc[tm][q], br[tm][q], M, I);      // (the planes carry the clip scale)
          });
          raw_reg[0][tm] += wmi ? 0 : __popc((unsigned)M);       // (lanes 32 .. 63 of M stay 0; no dynamic register index)
          raw_reg[1][tm] += wmi ? __popc((unsigned)M) : 0;
          if (lane < 32) mrow[lane] = (unsigned)M;
          if (I) {                                               // lanes owning a score inside a bracket: the exact
            static_for<0, 4>([&](auto gc) {                      // comparison, bit set in LDS; 4 scores per outer test
              constexpr int g4 = decltype(gc)::value;
              if (I & (0xf000u >> (4 * g4))) {
                static_for<0, 4>([&](auto kc) {
                  constexpr int q = 4 * g4 + decltype(kc)::value, R32 = (q & 3) + 8 * (q >> 2);
                  if (I & (0x8000u >> q)) {
                    const int rl = wmi * 64 + tm * 32 + R32 + 4 * lh;
                    // (a score inside a bracket belongs to a row WITH a bracket: its sA is the constant, not NaN -- no LDS read)
                    const float e = rank_sigmoid(acc[tm][q] * (1.0f / (kQScale * kQScale))), et = lds.eT[rl];
                    bool before = e < et;
                    if (e == et) before = (col < K ? cand[col] : -1) < lds.tI[rl];   // equal losses pop in id order
                    if (before) { atomicOr(mrow + R32 + 4 * lh, 1u << li); atomicAdd(&lds.extra[rl], 1); }
                  }
                });
              }
            });
          }
        }
      }
      // this wave's bitmap is complete (its own LDS writes, in order): known cells of its 64 x 32 block that rank before
      // the target are tallied (every wave scans the tile's few cells and keeps its own)
      if (known_off) {
        for (int32_t e = kn0 + lane; e < kn1; e += kWave) {
          const unsigned rc = known_rc[e];
          const int rl = rc >> 7, cl = rc & 127;
          if ((rl >> 6) == wmi && (cl >> 5) == wn && ((lds.bm[rl & 63] >> (cl & 31)) & 1u)) atomicAdd(&lds.skip[rl], 1);
        }
      }
    }
    __syncthreads();
    if constexpr (MODE != 2) {
      if (lane < 32) {
#pragma unroll
        for (int hm = 0; hm < 2; ++hm)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm) {
            const int64_t row = m0 + hm * 64 + tm * 32 + lane;
            if (row < B && raw_reg[hm][tm]) atomicAdd(&raw_cnt[row], raw_reg[hm][tm]);
          }
      }
      if (t < kRB && m0 + t < B) {
        if (lds.extra[t]) atomicAdd(&raw_cnt[m0 + t], lds.extra[t]);
        if (lds.skip[t]) atomicAdd(&skip_cnt[m0 + t], lds.skip[t]);
      }
    }
  }
  }
}

// ---- the pre-pass: candidate planes.  Slice S = 32 candidates, k block kb = 16 columns:
//   planes[((S * kKB + kb) * 2 + plane) * 32 * 16 + row * 16 + column]   fp16
// = high halves / remainders of cand row * clip scale * 2^8 (0 behind embedding_dim; NaN for a bad id or a row behind K):
// 1 KiB per (slice, k block, plane) = one operand fetch of one wave, lane (row, half) reading its 16 bytes in place.
template <int KKB>
__global__ __launch_bounds__(256) void rank_planes_kernel(const float* __restrict__ table, int64_t N, int d,
                                                          const int32_t* __restrict__ cand, int64_t K, float max_norm,
                                                          int spec, _Float16* __restrict__ planes) {
  constexpr int kChunks = HCfg<KKB>::kChunks;
  const int srow = threadIdx.x >> 2, qt = threadIdx.x & 3;       // 64 candidates a workgroup, four threads a row
  const int64_t pos = (int64_t)blockIdx.x * 64 + srow;
  const int32_t id = pos < K ? cand[pos] : -1;
  const bool bad = id < 0 || id >= N;
  const float* row = table + (int64_t)(bad ? 0 : id) * d;
  float4 r[kChunks][2];
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int col = min(c * 32 + qt * 8, d - 8);                 // clamped into the row, zeroed below
    r[c][0] = *reinterpret_cast<const float4*>(row + col);
    r[c][1] = *reinterpret_cast<const float4*>(row + col + 4);
  }
  f2 ss2 = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const bool in = c * 32 + qt * 8 < d;                         // (embedding_dim % 8 == 0: all eight or none)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const f2 xy = in ? f2{r[c][v].x, r[c][v].y} : f2{0.f, 0.f}, zw = in ? f2{r[c][v].z, r[c][v].w} : f2{0.f, 0.f};
      ss2 = __builtin_elementwise_fma(xy, xy, ss2);
      ss2 = __builtin_elementwise_fma(zw, zw, ss2);
    }
  }
  float ss = ss2.x + ss2.y;
  ss += __shfl_xor(ss, 1, kWave);
  ss += __shfl_xor(ss, 2, kWave);
  // spectral HolE rows: |x|^2 = (2 sum - X_0^2 - X_k^2) / d; the two real bins sit at columns 0 and d/2
  if (spec) {
    const float x_dc = row[0], x_ny = row[d >> 1];
    ss = (2.f * ss - x_dc * x_dc - x_ny * x_ny) / (float)d;
  }
  float inv;
  // t * clip(t) * 2^8: |t clip| <= max_norm, or max_norm sqrt(d/2) for one bin of a spectral row -- no fp16 overflow
  // for max_norm <= 8 whatever the table holds
  const float scale = bad ? __builtin_nanf("") : clip_scale(ss, max_norm, inv) * kQScale;
  _Float16* dst = planes + (pos >> 5) * (int64_t)KKB * 2 * kOpHalves + (pos & 31) * 16 + (qt & 1) * 8;
#pragma unroll
  for (int c = 0; c < kChunks; ++c) {
    const int kb = 2 * c + (qt >> 1);                            // this thread's eight columns: half of k block kb
    if (kb >= KKB) continue;
    const bool in = bad || c * 32 + qt * 8 < d;                  // (a bad row is NaN everywhere)
    const float x[8] = {r[c][0].x, r[c][0].y, r[c][0].z, r[c][0].w, r[c][1].x, r[c][1].y, r[c][1].z, r[c][1].w};
    h8 hi, mid;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h2 a, b;
      h_split(in ? x[2 * i] * scale : 0.f, in ? x[2 * i + 1] * scale : 0.f, a, b);
      hi[2 * i] = a.x; hi[2 * i + 1] = a.y; mid[2 * i] = b.x; mid[2 * i + 1] = b.y;
    }
    *reinterpret_cast<h8*>(dst + kb * 2 * kOpHalves) = hi;
    *reinterpret_cast<h8*>(dst + kb * 2 * kOpHalves + kOpHalves) = mid;
  }
}

// pos_of[entity] = its position in `cand` (-1, from the memset before: none)
__global__ void rank_pos_kernel(const int32_t* __restrict__ cand, int64_t K, int64_t N, int32_t* __restrict__ pos_of) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < K) {
    const int32_t id = cand[i];
    if (id >= 0 && id < N) pos_of[id] = (int32_t)i;
  }
}

int f16_cu_count() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 256;
  return cus;
}

inline bool f16_dim_ok(int32_t d, float max_norm) { return d % 8 == 0 && d >= 56 && d <= 288 && max_norm <= 8.f; }
inline int64_t pos_bytes(int64_t N) { return (N * (int64_t)sizeof(int32_t) + 255) / 256 * 256; }
inline int64_t planes_slices(int64_t K) { return 4 * ((K + kRB - 1) / kRB); }   // whole 128-candidate tiles

template <int KKB>
int f16_launch_kkb(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                   const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                   const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss, float* scores_out,
                   int spec, int scores_only, int sweep_flags, const void* planes_ws, hipStream_t st) {
  const int64_t n_rb = (B + kRB - 1) / kRB, n_ct = (K + kRB - 1) / kRB;
  if (n_ct > INT32_MAX / 8 || n_rb > INT32_MAX / 8) return GE_ENOTSUP;
  if (4 * n_ct * (int64_t)KKB * 2 * kOpHalves * 2 >= ((int64_t)1 << 32)) return GE_ENOTSUP;   // (32-bit byte offsets into the planes)
  const int64_t n_tiles = n_rb * n_ct;
  const int64_t grid = std::min<int64_t>(n_tiles, GE_PIPE_GRID_M * (int64_t)f16_cu_count());
  const int32_t* pos_of = reinterpret_cast<const int32_t*>(planes_ws);
  const _Float16* planes = reinterpret_cast<const _Float16*>(reinterpret_cast<const char*>(planes_ws) + pos_bytes(N));
  auto go = [&](auto kern) -> int {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBlk), h_lds_bytes<KKB>(), st, table, N, d, hr, B, true_id, cand, K,
                       max_norm, cand_is_head, known_off, known_rc, raw_cnt, skip_cnt, true_loss, scores_out, (int)n_ct,
                       n_tiles, spec, sweep_flags, pos_of, planes);
    return launch_status();
  };
  if (scores_only) return go(rank_f16_kernel<KKB, 2>);
  if (scores_out) return go(rank_f16_kernel<KKB, 1>);
  return go(rank_f16_kernel<KKB, 0>);
}

}  // namespace

#define GE_KKB_SWITCH(d, CALL)                                                                        \
  switch (((d) + 15) / 16) {                                                                          \
    case 4: CALL(4); case 5: CALL(5); case 6: CALL(6); case 7: CALL(7); case 8: CALL(8);              \
    case 9: CALL(9); case 10: CALL(10); case 11: CALL(11); case 12: CALL(12); case 13: CALL(13);      \
    case 14: CALL(14); case 15: CALL(15); case 16: CALL(16); case 17: CALL(17); case 18: CALL(18);    \
    default: return GE_ENOTSUP;                                                                       \
  }

// bytes of the candidate planes of a K-candidate sweep over an N-row table (0: embedding_dim has no split-precision sweep)
int64_t rank_planes_bytes(int64_t N, int32_t d, int64_t K) {
  if (d % 8 != 0 || d < 56 || d > 288 || N <= 0 || K <= 0) return 0;
  const int64_t kkb = (d + 15) / 16;
  const int64_t plane_bytes = planes_slices(K) * kkb * 2 * kOpHalves * (int64_t)sizeof(_Float16);
  // the sweep addresses the planes with 32-bit byte offsets (about 5.1 M candidates at d = 200): beyond that there is no
  // split-precision sweep -- said HERE, so that nobody allocates and fills 4 GiB of planes the sweep then refuses
  if (plane_bytes >= ((int64_t)1 << 32)) return 0;
  return pos_bytes(N) + plane_bytes;
}

// planes_ws (rank_planes_bytes, 256-byte aligned) <- the entity -> position map, then the candidates' fp16 planes
int rank_planes_launch(const float* table, int64_t N, int32_t d, const int32_t* cand, int64_t K, float max_norm, int spec,
                       void* planes_ws, hipStream_t st) {
  if (!f16_dim_ok(d, max_norm) || (K > 0 && N > 0 && rank_planes_bytes(N, d, K) == 0)) return GE_ENOTSUP;
  if (reinterpret_cast<uintptr_t>(planes_ws) % 256 != 0 || reinterpret_cast<uintptr_t>(table) % 16 != 0) return GE_EINVAL;
  if (K <= 0 || N <= 0) return 0;
  int32_t* pos_of = reinterpret_cast<int32_t*>(planes_ws);
  _Float16* planes = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(planes_ws) + pos_bytes(N));
  hipError_t e = hipMemsetAsync(pos_of, 0xff, (size_t)N * sizeof(int32_t), st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(rank_pos_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, cand, K, N, pos_of);
  const int64_t n_blocks = planes_slices(K) / 2;                // 64 candidates a workgroup
  if (n_blocks > INT32_MAX) return GE_ENOTSUP;
#define GE_CALL(KKB)                                                                                                   \
  hipLaunchKernelGGL(rank_planes_kernel<KKB>, dim3((unsigned)n_blocks), dim3(256), 0, st, table, N, d, cand, K, max_norm, \
                     spec, planes);                                                                                    \
  return launch_status()
  GE_KKB_SWITCH(d, GE_CALL)
#undef GE_CALL
}

// The split-precision sweep: embedding_dim % 8 == 0 in 56 ... 288 (k blocks 4 ... 18), max_norm <= 8
// (|q sa (1/d)| <= 2 max_norm^2, |t clip| <= max_norm sqrt(d/2): x 2^8 inside fp16).  GE_ENOTSUP otherwise.
// planes_ws: the candidates' planes from rank_planes_launch for the same (table, cand, max_norm, spec), or NULL -- then
// they are built here in a stream-ordered allocation (one more pass over the K candidate rows).
int sweep_f16_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B, const int32_t* true_id,
                     const int32_t* cand, int64_t K, float max_norm, int cand_is_head, const int32_t* known_off,
                     const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt, float* true_loss,
                     float* scores_out, int spec, int scores_only, int sweep_flags, const void* planes_ws, hipStream_t st) {
  if (!f16_dim_ok(d, max_norm) || rank_planes_bytes(N, d, K) == 0) return GE_ENOTSUP;   // (incl. planes beyond 32-bit offsets)
  static_assert(h_lds_bytes<18>() <= 160 * 1024, "LDS of the largest instantiation");
  void* own = nullptr;
  if (!planes_ws) {
    hipError_t e = hipMallocAsync(&own, (size_t)rank_planes_bytes(N, d, K), st);
    if (e != hipSuccess) return (int)e;
    const int rc = rank_planes_launch(table, N, d, cand, K, max_norm, spec, own, st);
    if (rc != 0) { (void)hipFreeAsync(own, st); return rc; }
    planes_ws = own;
  }
  auto run = [&]() -> int {
#define GE_CALL(KKB)                                                                                                 \
  return f16_launch_kkb<KKB>(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc,       \
                             raw_cnt, skip_cnt, true_loss, scores_out, spec, scores_only, sweep_flags, planes_ws, st)
    GE_KKB_SWITCH(d, GE_CALL)
#undef GE_CALL
  };
  int rc = run();
  if (own) {
    const hipError_t e = hipFreeAsync(own, st);
    if (rc == 0 && e != hipSuccess) rc = (int)e;
  }
  return rc;
}

}  // namespace ge
