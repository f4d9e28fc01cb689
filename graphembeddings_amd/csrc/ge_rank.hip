// ge_rank.hip -- link-prediction ranks straight out of the candidate sweep (holE.py:427-472, 564-575).
//
// The reference scores one (head, relation) against every candidate tail, pushes (loss, triple) on a
// heap and pops it in ascending order: raw rank = pops until the true tail, filtered rank = the same
// without known-true candidates.  Equivalently, for test row i with true candidate c*:
//     raw_i      = 1 + #{c : (E_ic, id_c) < (E_ic*, id_c*)}        (lexicographic: loss, then entity id)
//     filtered_i = raw_i - #{known-true c != c* counted above}.
// ge_complex_score_1vK + host code materialises all B x K losses to do this (883 M floats per side for
// the FB15k test set).  Here the counting is the GEMM's epilogue and no score is ever stored.
//
// Structure (CDNA4): one 256-thread workgroup per (row block of 128 test rows, column split); it keeps
// its Q operand -- q = clip-free fixed o relation, the whole k range, [Re q | Im q] -- in LDS for its
// entire life (128 x 201 floats = 103 KB of the CU's 160 KB) and sweeps its share of the 128-wide
// candidate tiles.  Per tile only the candidate rows move: 32-float chunks of the raw table rows
// (score = [Re q | Im q] . [Re t | Im t], the row exactly as stored), double-buffered through LDS while
// v_mfma_f32_32x32x2_f32 (exact fp32) runs on the previous chunk; d = 200 is 6 chunks + 8 floats, no k
// padding.  Row strides 201 / 33 floats are odd: the 32 rows a half-wave reads hit 32 banks.
// The true candidate's loss comes from the SAME tile code (a "diagonal" tile whose candidate rows are the
// 128 true entities), so equal losses are bitwise equal and ties break by id exactly as the heap does.
// Epilogue per tile: sigmoid, compare, one ballot per accumulator register -> a 128 x 128 bit mask in
// LDS; a thread per row pop-counts it (raw), and the tile's (row, col) list of known-true candidates,
// prepared by the host, is looked up in the same mask (filtered).
#include "ge_rank_dev.h"

namespace ge {

constexpr int kChunk = 32;        // reals per staged B chunk
constexpr int kLdb = kChunk + 1;  // odd LDS stride

struct RankLds {
  float* A;        // [kRB][lda]
  float* Bs;       // [2][kRB][kLdb]
  float* sA;       // [kRB] (unused: the fixed x relation clip scale is folded into A)
  float* sB;       // [kRB] candidate clip scale (NaN: bad id)
  float* eT;       // [kRB] loss of the true candidate
  unsigned* bm;    // [kRB][4] `before` bits of the current tile
  int* skip;       // [kRB] known-true candidates ranked before the target
  int* tI;         // [kRB] entity id of the true candidate (-1 beyond B)
};

// STEPS k-pairs of one staged chunk, fully unrolled: with one wave per SIMD (the Q operand fills the LDS, so a CU
// holds one workgroup) nothing else hides the LDS latency -- all 4*STEPS operand reads are visible to the
// scheduler at once and run ahead of the MFMAs that consume them.
template <int STEPS>
__device__ __forceinline__ void rank_mma(const float* __restrict__ ap, const float* __restrict__ bp, int lda,
                                         f32x16 (&acc)[2][2]) {
  float a0[STEPS], a1[STEPS], b0[STEPS], b1[STEPS];
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    a0[s] = ap[2 * s]; a1[s] = ap[32 * lda + 2 * s];
    b0[s] = bp[2 * s]; b1[s] = bp[32 * kLdb + 2 * s];
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead: the scheduler otherwise sinks them next to their MFMA
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b0[s], acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], b1[s], acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b0[s], acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], b1[s], acc[1][1], 0, 0, 0);
  }
}

// Chunk `ch` (32 reals) of candidate row `cid` into registers: thread t stages row t>>1, half t&1.
__device__ __forceinline__ void rank_fetch(const float* __restrict__ table, int64_t N, int d, int32_t cid, int ch,
                                           float4 (&r)[4]) {
  const int half = threadIdx.x & 1;
  const bool bad = cid < 0 || cid >= N;
  const float* crow = table + (int64_t)(bad ? 0 : cid) * d;
  const int c0 = ch * kChunk + half * 16;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const int c = c0 + 4 * v;
    r[v] = *reinterpret_cast<const float4*>(crow + min(c, d - 4));   // clamped, never predicated; zeroed when stored
  }
}

// One 128 x 128 tile: acc = Q . T^T for the candidate rows `cid` (this thread stages row t>>1, half t&1),
// candidate clip scales to lds.sB.  rA / rB hold chunks 0 and 1 of the row on entry (the caller requested them
// under the previous tile's epilogue).  Identical instruction sequence for every tile, diagonal tile included.
//
// A chunk iteration is ONE instruction stream per wave (one wave per SIMD): everything that is not an MFMA is
// cut into slices of a few instructions and placed BETWEEN the MFMAs, where it issues in the shadow of the
// 64-cycle matrix instruction before it: iteration ch stores chunk ch+1 (registers, requested an iteration
// ago) to the free LDS buffer in 8 slices, then requests chunk ch+2 in 4 slices.  sched_barriers pin the order.
__device__ __forceinline__ void rank_tile(const float* __restrict__ table, int64_t N, int d, int lda,
                                          int32_t cid, float max_norm, int spec, const RankLds& lds, float4 (&rA)[4],
                                          float4 (&rB)[4], f32x16 (&acc)[2][2]) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int srow = t >> 1, half = t & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool bad = cid < 0 || cid >= N;
  const float* crow = table + (int64_t)(bad ? 0 : cid) * d;
  const float x_dc = crow[0], x_ny = crow[d >> 1];   // spectral HolE rows: |x|^2 = (2 sum - X_0^2 - X_k^2) / d
  const int n_chunks = (d + kChunk - 1) / kChunk;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  float ss = 0.f;
  {  // chunk 0 -> buffer 0
    float* dst = lds.Bs + srow * kLdb + half * 16;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const bool ok = !bad && half * 16 + 4 * v + 3 < d;
      const float x = ok ? rA[v].x : 0.f, y = ok ? rA[v].y : 0.f, z = ok ? rA[v].z : 0.f, w2 = ok ? rA[v].w : 0.f;
      dst[4 * v] = x; dst[4 * v + 1] = y; dst[4 * v + 2] = z; dst[4 * v + 3] = w2;
      ss += x * x + y * y + z * z + w2 * w2;
    }
  }
  __syncthreads();
  // body(ch, rs, rf): rs holds chunk ch+1, rf is free and receives chunk ch+2
  auto body = [&](int ch, float4 (&rs)[4], float4 (&rf)[4]) {
    const int buf = ch & 1;
    const float* ap = lds.A + (wm * 64 + li) * lda + ch * kChunk + lh;
    const float* bp = lds.Bs + (buf * kRB + wn * 64 + li) * kLdb + lh;
    float* dst = lds.Bs + ((buf ^ 1) * kRB + srow) * kLdb + half * 16;
    const int c2 = (ch + 2) * kChunk + half * 16;
    float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      a0[s2] = ap[2 * s2]; a1[s2] = ap[32 * lda + 2 * s2];
      b0[s2] = bp[2 * s2]; b1[s2] = bp[32 * kLdb + 2 * s2];
    }
    __builtin_amdgcn_sched_barrier(0);
    // No control flow in here: a branch makes the waitcnt pass serialise the requests (vmcnt(0) before each).
    // Requests past the row's end are clamped to its last 16 bytes and zeroed when they are stored.
    auto piece = [&](int pc) {
      if (pc < 4) {                                   // one 16-byte request of chunk ch+2
        rf[pc] = *reinterpret_cast<const float4*>(crow + min(c2 + 4 * pc, d - 4));
      } else if (pc >= 8 && pc < 40 && (pc & 3) == 0) {   // half a float4 of chunk ch+1 to LDS, its squares to the norm
        const int i = (pc - 8) >> 2, v = i >> 1;
        const bool ok = !bad && c2 - kChunk + 4 * v + 3 < d;
        if (i & 1) {
          const float z = ok ? rs[v].z : 0.f, w2 = ok ? rs[v].w : 0.f;
          dst[4 * v + 2] = z; dst[4 * v + 3] = w2; ss += z * z + w2 * w2;
        } else {
          const float x = ok ? rs[v].x : 0.f, y = ok ? rs[v].y : 0.f;
          dst[4 * v] = x; dst[4 * v + 1] = y; ss += x * x + y * y;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s2], b0[s2], acc[0][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(4 * s2);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s2], b1[s2], acc[0][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(4 * s2 + 1);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s2], b0[s2], acc[1][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(4 * s2 + 2);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s2], b1[s2], acc[1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      piece(4 * s2 + 3);
    }
    __syncthreads();
  };
  const int n_full = d / kChunk;
  int ch = 0;
  for (; ch + 1 < n_full; ch += 2) {      // register sets swap roles every iteration
    body(ch, rB, rA);
    body(ch + 1, rA, rB);
  }
  if (ch < n_full) { body(ch, rB, rA); ++ch; }
  if (n_full < n_chunks) {                // the 8/16/24-real tail chunk (already in LDS)
    const float* ap = lds.A + (wm * 64 + li) * lda + n_full * kChunk + lh;
    const float* bp = lds.Bs + ((n_full & 1) * kRB + wn * 64 + li) * kLdb + lh;
    for (int kk = 0; kk < d - n_full * kChunk; kk += 8) rank_mma<4>(ap + kk, bp + kk, lda, acc);
    __syncthreads();
  }
  ss += __shfl_xor(ss, 1, kWave);
  if (spec) ss = (2.f * ss - x_dc * x_dc - x_ny * x_ny) / (float)d;
  if (half == 0) {
    float inv;
    lds.sB[srow] = bad ? __builtin_nanf("") : clip_scale(ss, max_norm, inv);
  }
  __syncthreads();
}

// grid (column splits, row blocks).  known_off [n_row_blocks * n_col_tiles + 1], known_rc: (row_local << 7 |
// col_local) of the known-true candidates of each (row block, column tile), may be null.
__global__ __launch_bounds__(kBlock) void rank_1vK_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ true_id, const int32_t* __restrict__ cand, int64_t K, float max_norm,
    int cand_is_head, const int32_t* __restrict__ known_off, const uint16_t* __restrict__ known_rc,
    int32_t* __restrict__ raw_cnt, int32_t* __restrict__ skip_cnt, float* true_loss,
    float* __restrict__ scores_out, int lda, int spec, int vs_loss) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  RankLds lds;
  lds.A = smem;
  lds.Bs = lds.A + kRB * lda;
  lds.sA = lds.Bs + 2 * kRB * kLdb;
  lds.sB = lds.sA + kRB;
  lds.eT = lds.sB + kRB;
  lds.bm = reinterpret_cast<unsigned*>(lds.eT + kRB);
  lds.skip = reinterpret_cast<int*>(lds.bm + kRB * 4);
  lds.tI = lds.skip + kRB;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int srow = t >> 1, half = t & 1;
  const int k = d >> 1;
  const int64_t m0 = (int64_t)blockIdx.y * kRB;
  const int n_ct = (int)((K + kRB - 1) / kRB);

  // ---- Q = fixed o relation for this block's rows, whole k range, kept in LDS for the sweep
  {
    const int64_t r = m0 + srow;
    int32_t fid = -1, rid = -1;
    if (r < B) { fid = hr[2 * r]; rid = hr[2 * r + 1]; }
    const bool bad = fid < 0 || fid >= N || rid < 0 || rid >= N;
    const float* frow = table + (int64_t)(bad ? 0 : fid) * d;
    const float* rrow = table + (int64_t)(bad ? 0 : rid) * d;
    float ssf = 0.f, ssr = 0.f;
    float* arow = lds.A + srow * lda;
    const int kh = (k + 1) / 2;                                  // complex dims per staging thread
    const int c_lo = half * kh, c_hi = min(k, (half + 1) * kh);
    // spectral HolE (ge_complex_dev.h): Hermitian weight 2 on every bin but element 0, which packs the two REAL
    // bins X_0 | X_k; norms and score carry the Parseval factor 1/d
    for (int c = c_lo; c < c_hi; ++c) {                          // pass 1: the two clip norms
      const float fre = bad ? 0.f : frow[c], fim = bad ? 0.f : frow[k + c];
      const float rre = bad ? 0.f : rrow[c], rim = bad ? 0.f : rrow[k + c];
      const float wgt = (spec && c != 0) ? 2.f : 1.f;
      ssf += wgt * (fre * fre + fim * fim);
      ssr += wgt * (rre * rre + rim * rim);
    }
    ssf += __shfl_xor(ssf, 1, kWave);
    ssr += __shfl_xor(ssr, 1, kWave);
    const float inv_d = spec ? 1.0f / (float)d : 1.0f;
    float i0, i1;
    // the product of the fixed row's and the relation row's clip scales is folded into Q (NaN: bad ids / beyond B,
    // which makes every loss of the row NaN and every comparison false)
    const float sa = (bad || r >= B) ? __builtin_nanf("")
                                     : clip_scale(ssf * inv_d, max_norm, i0) * clip_scale(ssr * inv_d, max_norm, i1) * inv_d;
    for (int c = c_lo; c < c_hi; ++c) {                          // pass 2: q = fixed o relation, scaled
      const float fre = bad ? 0.f : frow[c], fim = bad ? 0.f : frow[k + c];
      const float rre = bad ? 0.f : rrow[c], rim = bad ? 0.f : rrow[k + c];
      float qre, qim;
      if (spec && c == 0) {  // two independent real dimensions
        qre = fre * rre;
        qim = fim * rim;
      } else if (!cand_is_head) {   // q = h * r ; score = Re(q conj t)
        qre = fre * rre - fim * rim;
        qim = fre * rim + fim * rre;
      } else {               // Re(h r conj t) with h the candidate: Q = [Re(r conj t) | -Im(r conj t)]
        qre = rre * fre + rim * fim;
        qim = -(rim * fre - rre * fim);
      }
      const float wgt = (spec && c != 0) ? 2.f : 1.f;
      arow[c] = qre * sa * wgt;
      arow[k + c] = qim * sa * wgt;
    }
    if (half == 0) {
      lds.skip[srow] = 0;
      lds.tI[srow] = r < B ? true_id[r] : -1;
    }
  }
  __syncthreads();

  f32x16 acc[2][2];
  // ---- the true candidates' losses: a tile whose candidate rows are this block's 128 true entities
  float et[2][16];     // E of the true candidate of each of this lane's 32 accumulator rows, for the whole sweep
  float4 rA[4], rB[4];  // chunks 0 and 1 of the NEXT tile's candidate row, requested before the current tile's epilogue
  {
    rank_fetch(table, N, d, lds.tI[srow], 0, rA);
    rank_fetch(table, N, d, lds.tI[srow], 1, rB);
    rank_tile(table, N, d, lda, lds.tI[srow], max_norm, spec, lds, rA, rB, acc);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          const int cl = wn * 64 + tn * 32 + li;
          if (rl == cl) lds.eT[rl] = rank_sigmoid(acc[tm][tn][q] * lds.sB[cl]);
        }
      }
    __syncthreads();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int q = 0; q < 16; ++q) et[tm][q] = lds.eT[wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh];
    if (vs_loss) {
      // ranking against GIVEN losses (ge_rank_1vK_vs_loss): the tile above ran on whatever rows true_id names (the
      // tie-break ids; any row, or none) and is overruled
      __syncthreads();
      if (t < kRB) lds.eT[t] = m0 + t < B ? true_loss[m0 + t] : __builtin_nanf("");
      __syncthreads();
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) et[tm][q] = lds.eT[wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh];
    } else if (true_loss && blockIdx.x == 0 && t < kRB && m0 + t < B) true_loss[m0 + t] = lds.eT[t];
  }
  int raw_reg = 0;

  // ---- the sweep over this split's candidate tiles
  auto cand_of = [&](int ct) -> int32_t {
    const int64_t c = (int64_t)ct * kRB + srow;
    return (ct < n_ct && c < K) ? cand[c] : -1;
  };
  int32_t cid = cand_of(blockIdx.x);
  rank_fetch(table, N, d, cid, 0, rA);
  rank_fetch(table, N, d, cid, 1, rB);
  for (int ct = blockIdx.x; ct < n_ct; ct += gridDim.x) {
    const int64_t n0 = (int64_t)ct * kRB;
    rank_tile(table, N, d, lda, cid, max_norm, spec, lds, rA, rB, acc);
    cid = cand_of(ct + gridDim.x);
    rank_fetch(table, N, d, cid, 0, rA);                         // land while the epilogue below runs
    rank_fetch(table, N, d, cid, 1, rB);
    // epilogue: C layout of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // A candidate beyond K or with a bad id has a NaN clip scale, a row beyond B a NaN Q: every comparison false.
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
      const int cl = wn * 64 + tn * 32 + li;
      const int64_t col = n0 + cl;
      const float sb = lds.sB[cl];
      const int32_t cand_c = col < K ? cand[col] : -1;
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float e = rank_sigmoid(acc[tm][tn][q] * sb);
          unsigned long long mask = __ballot(e < et[tm][q]);
          const unsigned long long ties = __ballot(e == et[tm][q]);
          const int rl = wm * 64 + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
          if (ties) mask |= __ballot(e == et[tm][q] && cand_c < lds.tI[rl]);     // equal losses pop in id order
          if (li == 0) lds.bm[rl * 4 + wn * 2 + tn] = (unsigned)(mask >> (32 * lh));
          if (scores_out && col < K && m0 + rl < B) scores_out[(m0 + rl) * K + col] = e;
        }
    }
    __syncthreads();
    if (t < kRB) {
      const unsigned* m = lds.bm + t * 4;
      raw_reg += __popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]);
    }
    if (known_off) {
      const int64_t tile = (int64_t)blockIdx.y * n_ct + ct;
      const int32_t e0 = known_off[tile], e1 = known_off[tile + 1];
      for (int32_t e = e0 + t; e < e1; e += kBlock) {
        const unsigned rc = known_rc[e];
        const int rl = rc >> 7, cl = rc & 127;
        if ((lds.bm[rl * 4 + (cl >> 5)] >> (cl & 31)) & 1u) atomicAdd(&lds.skip[rl], 1);
      }
    }
    __syncthreads();
  }
  if (t < kRB && m0 + t < B) {
    atomicAdd(&raw_cnt[m0 + t], raw_reg);
    if (lds.skip[t]) atomicAdd(&skip_cnt[m0 + t], lds.skip[t]);
  }
}

static size_t rank_lds_bytes(int d) {
  const int lda = d + 1;
  return sizeof(float) * ((size_t)kRB * lda + 2 * kRB * kLdb + 3 * kRB) + sizeof(unsigned) * kRB * 4 + sizeof(int) * 2 * kRB;
}

int rank_max_dim() { return 288; }   // the split-precision sweep's Q planes fill the LDS at 18 k blocks (max_norm <= 8);
constexpr int kRankMaxDimF32 = 232;  // the fp32 kernels: Q (128 x (d+1) floats) + two candidate chunks must fit the CU's 160 KiB

int complex_rank_1vK_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B,
                            const int32_t* true_id, const int32_t* cand, int64_t K, float max_norm, int cand_is_head,
                            const int32_t* known_off, const uint16_t* known_rc, int32_t* raw_cnt, int32_t* skip_cnt,
                            float* true_loss, float* scores_out, int spec, const void* planes_ws, hipStream_t st, int vs_loss) {
  // vs_loss: true_loss is an INPUT -- the loss every candidate of row i is ranked against -- and true_id the tie-break id
  if (d <= 0 || (d & 7)) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;   // 16-byte candidate loads, 8-float tail
  if (d > rank_max_dim()) return GE_ENOTSUP;
  if (reinterpret_cast<uintptr_t>(table) % 16 != 0) return GE_EINVAL;
  if (B == 0 || K == 0) return 0;
  {  // embedding_dim a multiple of 40, 32 or 24: the software-pipelined kernel (ge_rank_pipe.hip)
    const int rc = rank_pipe_launch(table, N, d, hr, B, true_id, cand, K, max_norm, cand_is_head, known_off, known_rc,
                                    raw_cnt, skip_cnt, true_loss, scores_out, spec, planes_ws, st, vs_loss);
    if (rc != GE_ENOTSUP) return rc;
  }
  if (d > kRankMaxDimF32) return GE_ENOTSUP;                    // (233 ... 288 with max_norm > 8)
  const int64_t n_rb = (B + kRB - 1) / kRB, n_ct = (K + kRB - 1) / kRB;
  if (n_rb > 65535) return GE_ENOTSUP;
  // column splits: enough workgroups for ~4 waves of the 256 CUs, never more than column tiles
  int64_t splits = (4 * 256 + n_rb - 1) / n_rb;
  if (splits > n_ct) splits = n_ct;
  if (splits < 1) splits = 1;
  const size_t lds = rank_lds_bytes(d);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rank_1vK_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(rank_1vK_kernel, dim3((unsigned)splits, (unsigned)n_rb), dim3(kBlock), lds, st, table, N, d, hr, B,
                     true_id, cand, K, max_norm, cand_is_head, known_off, known_rc, raw_cnt, skip_cnt, true_loss,
                     scores_out, d + 1, spec, vs_loss);
  return launch_status();
}

}  // namespace ge
