// ge_1vk.hip -- 1-vs-K candidate scoring as an fp32-MFMA GEMM.
//
// Replaces the inference loop of holE.py:564-569 (itertools.product([head], tail_candidates,
// relations) -> evaluate_triples per candidate) and serves K shared negatives per positive
// (BASELINE config 5).  With q = clip(h) o clip(r) (complex product),
//     score(h, r, t_j) = sum_k Re(q_k conj(t_jk)) = [Re q | Im q] . [Re t_j | Im t_j],
// so S[B,K] = Q[B,d] . T[K,d]^T.  Every clip is a per-row scalar: the GEMM runs on the RAW rows
// (Q_raw = fixed o rel formed on the fly while staging) and the product of the three clip scales is
// applied in the epilogue, fused with the sigmoid (holE.py:198).
//
// MFMA: v_mfma_f32_32x32x2_f32 -- exact fp32 (a k-ordered fmaf chain), which keeps the result
// within 1e-5 of the fp32 reference arithmetic; bf16 MFMA would not.  Block = 4 waves as 2x2, each
// wave TM x TN tiles of 32x32; k is walked in chunks of 16 complex dims (= 32 real k) staged in LDS
// with a row stride of 33 floats so the 32 rows a half-wave reads per operand fall in 32 banks.
#include "ge_common.h"
#include "ge_rank_dev.h"

namespace ge {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int KC = 16;   // complex dims per chunk
constexpr int LDA = 33;  // LDS row stride (floats): 2*KC + 1

template <bool V4>
__device__ __forceinline__ void load4(const float* __restrict__ row, int c0, int k, bool im, float (&v)[4]) {
  // 4 consecutive complex-dim values c0..c0+3 of the real (im=false) or imaginary half
  const float* p = row + (im ? k : 0) + c0;
  if (V4) {
    if (c0 + 3 < k) {
      const float4 x = *reinterpret_cast<const float4*>(p);
      v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
      return;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = (c0 + q < k) ? p[q] : 0.f;
}

template <int TM, int TN, bool V4>
__global__ __launch_bounds__(kBlock) void score_1vK_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ cand, int64_t K, float max_norm, int apply_sigmoid, int cand_is_head,
    float* __restrict__ out) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  __shared__ float As[BM * LDA];
  __shared__ float Bs[BN * LDA];
  __shared__ float sA[BM];
  __shared__ float sB[BN];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int lrow = t >> 2, lj = t & 3;
  const int k = d >> 1;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  int32_t fid[TM], rid[TM], cid[TN];
  bool abad[TM], bbad[TN];
#pragma unroll
  for (int p = 0; p < TM; ++p) {
    const int64_t r = m0 + lrow + 64 * p;
    fid[p] = -1; rid[p] = -1; abad[p] = false;
    if (r < B) {
      fid[p] = hr[2 * r]; rid[p] = hr[2 * r + 1];
      abad[p] = fid[p] < 0 || fid[p] >= N || rid[p] < 0 || rid[p] >= N;
      if (abad[p]) { fid[p] = -1; rid[p] = -1; }
    }
  }
#pragma unroll
  for (int p = 0; p < TN; ++p) {
    const int64_t c = n0 + lrow + 64 * p;
    cid[p] = -1; bbad[p] = false;
    if (c < K) {
      cid[p] = cand[c];
      bbad[p] = cid[p] < 0 || cid[p] >= N;
      if (bbad[p]) cid[p] = -1;
    }
  }
  float ssf[TM], ssr[TM], ssc[TN];
#pragma unroll
  for (int p = 0; p < TM; ++p) { ssf[p] = 0.f; ssr[p] = 0.f; }
#pragma unroll
  for (int p = 0; p < TN; ++p) ssc[p] = 0.f;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

  for (int kc = 0; kc < k; kc += KC) {
    const int c0 = kc + 4 * lj;
#pragma unroll
    for (int p = 0; p < TM; ++p) {
      float qre[4] = {0.f, 0.f, 0.f, 0.f}, qim[4] = {0.f, 0.f, 0.f, 0.f};
      if (fid[p] >= 0 && c0 < k) {
        float fre[4], fim[4], rre[4], rim[4];
        const float* frow = table + (int64_t)fid[p] * d;
        const float* rrow = table + (int64_t)rid[p] * d;
        load4<V4>(frow, c0, k, false, fre); load4<V4>(frow, c0, k, true, fim);
        load4<V4>(rrow, c0, k, false, rre); load4<V4>(rrow, c0, k, true, rim);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ssf[p] += fre[q] * fre[q] + fim[q] * fim[q];
          ssr[p] += rre[q] * rre[q] + rim[q] * rim[q];
          if (!cand_is_head) {  // q = h * r
            qre[q] = fre[q] * rre[q] - fim[q] * rim[q];
            qim[q] = fre[q] * rim[q] + fim[q] * rre[q];
          } else {              // Re(h * r * conj(t)): Q = [Re(r conj t) | -Im(r conj t)]
            qre[q] = rre[q] * fre[q] + rim[q] * fim[q];
            qim[q] = -(rim[q] * fre[q] - rre[q] * fim[q]);
          }
        }
      }
      float* dst = As + (lrow + 64 * p) * LDA + 4 * lj;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dst[q] = qre[q]; dst[KC + q] = qim[q]; }
    }
#pragma unroll
    for (int p = 0; p < TN; ++p) {
      float cre[4] = {0.f, 0.f, 0.f, 0.f}, cim[4] = {0.f, 0.f, 0.f, 0.f};
      if (cid[p] >= 0 && c0 < k) {
        const float* crow = table + (int64_t)cid[p] * d;
        load4<V4>(crow, c0, k, false, cre); load4<V4>(crow, c0, k, true, cim);
#pragma unroll
        for (int q = 0; q < 4; ++q) ssc[p] += cre[q] * cre[q] + cim[q] * cim[q];
      }
      float* dst = Bs + (lrow + 64 * p) * LDA + 4 * lj;
#pragma unroll
      for (int q = 0; q < 4; ++q) { dst[q] = cre[q]; dst[KC + q] = cim[q]; }
    }
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2 * KC; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) a[tm] = As[(wm * 32 * TM + tm * 32 + li) * LDA + kk + lh];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[(wn * 32 * TN + tn * 32 + li) * LDA + kk + lh];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
    __syncthreads();
  }

  // clip scales (get_embedding, holE.py:162) -> LDS
  const float nanv = __builtin_nanf("");
#pragma unroll
  for (int p = 0; p < TM; ++p) {
    float a = ssf[p], b = ssr[p];
    a += __shfl_xor(a, 1, kWave); a += __shfl_xor(a, 2, kWave);
    b += __shfl_xor(b, 1, kWave); b += __shfl_xor(b, 2, kWave);
    if (lj == 0) {
      float i0, i1;
      sA[lrow + 64 * p] = abad[p] ? nanv : clip_scale(a, max_norm, i0) * clip_scale(b, max_norm, i1);
    }
  }
#pragma unroll
  for (int p = 0; p < TN; ++p) {
    float c = ssc[p];
    c += __shfl_xor(c, 1, kWave); c += __shfl_xor(c, 2, kWave);
    if (lj == 0) {
      float i0;
      sB[lrow + 64 * p] = bbad[p] ? nanv : clip_scale(c, max_norm, i0);
    }
  }
  __syncthreads();

  // epilogue: C layout of the 32x32 f32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int cl = wn * 32 * TN + tn * 32 + li;
      const int64_t col = n0 + cl;
      const float sb = sB[cl];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rl = wm * 32 * TM + tm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const int64_t row = m0 + rl;
        if (row < B && col < K) {
          const float s = acc[tm][tn][q] * sA[rl] * sb;
          out[row * K + col] = apply_sigmoid ? sigmoidf_dev(s) : s;
        }
      }
    }
}

// Small problems (a few hundred 64x64 tiles, e.g. B=4096 positives x K=256 shared negatives) are
// latency-bound in the chunked kernel: 13 load->barrier->MFMA->barrier rounds at one block per CU.
// This variant stages the WHOLE k range once (64 x (2*KP+1) floats per operand, ~100 KiB of LDS at
// d=200), keeps every global load of the block in flight together, and then runs the MFMAs
// back-to-back.  Row stride 2*KP+1 is odd, so the 32 rows a half-wave reads hit 32 distinct banks.
template <bool V4>
__global__ __launch_bounds__(kBlock) void score_1vK_fullk_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ cand, int64_t K, float max_norm, int apply_sigmoid, int cand_is_head,
    float* __restrict__ out, int KP) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lda = 2 * KP + 1;
  float* As = smem;
  float* Bs = As + 64 * lda;
  float* sA = Bs + 64 * lda;
  float* sB = sA + 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int lrow = t >> 2, lj = t & 3;
  const int k = d >> 1;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  int32_t fid = -1, rid = -1, cid = -1;
  bool abad = false, bbad = false;
  {
    const int64_t r = m0 + lrow;
    if (r < B) {
      fid = hr[2 * r]; rid = hr[2 * r + 1];
      abad = fid < 0 || fid >= N || rid < 0 || rid >= N;
      if (abad) { fid = -1; rid = -1; }
    }
    const int64_t c = n0 + lrow;
    if (c < K) {
      cid = cand[c];
      bbad = cid < 0 || cid >= N;
      if (bbad) cid = -1;
    }
  }
  float ssf = 0.f, ssr = 0.f, ssc = 0.f;
  const float* frow = table + (int64_t)(fid >= 0 ? fid : 0) * d;
  const float* rrow = table + (int64_t)(rid >= 0 ? rid : 0) * d;
  const float* crow = table + (int64_t)(cid >= 0 ? cid : 0) * d;
  float* arow = As + lrow * lda;
  float* brow = Bs + lrow * lda;
  for (int c0 = 4 * lj; c0 < KP; c0 += 32) {   // two 4-wide column groups per pass: 12 float4 loads in flight
    float fre[2][4], fim[2][4], rre[2][4], rim[2][4], cre[2][4], cim[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cc = c0 + 16 * u;
      const bool in = cc < KP;
#pragma unroll
      for (int q = 0; q < 4; ++q) { fre[u][q] = fim[u][q] = rre[u][q] = rim[u][q] = cre[u][q] = cim[u][q] = 0.f; }
      if (in && fid >= 0) {
        load4<V4>(frow, cc, k, false, fre[u]); load4<V4>(frow, cc, k, true, fim[u]);
        load4<V4>(rrow, cc, k, false, rre[u]); load4<V4>(rrow, cc, k, true, rim[u]);
      }
      if (in && cid >= 0) { load4<V4>(crow, cc, k, false, cre[u]); load4<V4>(crow, cc, k, true, cim[u]); }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cc = c0 + 16 * u;
      if (cc >= KP) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ssf += fre[u][q] * fre[u][q] + fim[u][q] * fim[u][q];
        ssr += rre[u][q] * rre[u][q] + rim[u][q] * rim[u][q];
        ssc += cre[u][q] * cre[u][q] + cim[u][q] * cim[u][q];
        float qre, qim;
        if (!cand_is_head) {
          qre = fre[u][q] * rre[u][q] - fim[u][q] * rim[u][q];
          qim = fre[u][q] * rim[u][q] + fim[u][q] * rre[u][q];
        } else {
          qre = rre[u][q] * fre[u][q] + rim[u][q] * fim[u][q];
          qim = -(rim[u][q] * fre[u][q] - rre[u][q] * fim[u][q]);
        }
        arow[cc + q] = qre; arow[KP + cc + q] = qim;
        brow[cc + q] = cre[u][q]; brow[KP + cc + q] = cim[u][q];
      }
    }
  }
  // clip scales
  ssf += __shfl_xor(ssf, 1, kWave); ssf += __shfl_xor(ssf, 2, kWave);
  ssr += __shfl_xor(ssr, 1, kWave); ssr += __shfl_xor(ssr, 2, kWave);
  ssc += __shfl_xor(ssc, 1, kWave); ssc += __shfl_xor(ssc, 2, kWave);
  if (lj == 0) {
    const float nanv = __builtin_nanf("");
    float i0, i1;
    sA[lrow] = abad ? nanv : clip_scale(ssf, max_norm, i0) * clip_scale(ssr, max_norm, i1);
    sB[lrow] = bbad ? nanv : clip_scale(ssc, max_norm, i0);
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int li = lane & 31, lh = lane >> 5;
  const float* ap = As + (wm * 32 + li) * lda + lh;
  const float* bp = Bs + (wn * 32 + li) * lda + lh;
#pragma unroll 4
  for (int kk = 0; kk < 2 * KP; kk += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
  const int cl = wn * 32 + li;
  const int64_t col = n0 + cl;
  const float sb = sB[cl];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rl = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
    const int64_t row = m0 + rl;
    if (row < B && col < K) {
      const float sv = acc[q] * sA[rl] * sb;
      out[row * K + col] = apply_sigmoid ? sigmoidf_dev(sv) : sv;
    }
  }
}

// BASELINE config 5 at its own shape (B = 4096 positives x K = 256 shared negatives x d = 200: 256 tiles of 64 x 64,
// one per CU) is bounded by how fast ONE workgroup gets its 192 rows out of L2 and through 100 fp32 MFMAs per wave,
// not by throughput.  This variant keeps the whole tile's global loads in flight in TWO batches (complex column
// groups [0, 16) and [16, ..)), so the MFMAs of the first batch run while the second is still landing, and lays the
// operands out for 16-byte LDS traffic both ways: Xs[g][row] = 4 consecutive k of a row (g = k / 4; real parts first,
// then imaginary parts), written with ds_write_b128 by a lane mapping that puts 4 lanes on 64 contiguous bytes of a
// table row (coalesced) and read back with ONE ds_read_b128 per operand per two MFMAs -- the half-wave lh = 1 takes
// k = 2, 3 of the group where lh = 0 takes 0, 1 (any k order gives the same dot product as long as both operands
// agree; the summation order differs from the other kernels' in the last bits only).
template <int NITA, int NITB>
__global__ __launch_bounds__(kBlock) void score_1vK_tile_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ cand, int64_t K, float max_norm, int apply_sigmoid, int cand_is_head,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = d >> 1, CG = k >> 2;                 // complex column groups of 4 (d % 8 == 0)
  float4* As = reinterpret_cast<float4*>(smem);      // [2 CG][64]
  float4* Bs = As + 2 * CG * 64;
  // the clip scales sit behind BOTH the operands and the 64 x 65-float output tile that later aliases them (d < 40:
  // the tile is the larger of the two), so the epilogue's Cs stores never reach sA / sB while other waves read them
  float* sA = smem + max(4 * 4 * CG * 64, 64 * 65);
  float* sB = sA + 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int srow = 16 * w + (lane & 15), gq = lane >> 4;     // staging: 4 lanes (gq) on 64 contiguous bytes of a row
  // Workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md), each with its own L2: the tiles that share
  // a row block (its 128 rows of Q operands) are renumbered onto ONE XCD, so those rows leave the Infinity Cache once
  // per row block instead of once per tile (39 MB -> ~10 MB of fabric traffic at 4096 x 256 x 200).  Speed only.
  const int gx = gridDim.x, T = gx * (int)gridDim.y;
  const int L = (int)blockIdx.y * gx + (int)blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int V = xcd * (T >> 3) + min(xcd, T & 7) + slot;
  const int64_t m0 = (int64_t)(V / gx) * 64, n0 = (int64_t)(V % gx) * 64;
  int32_t fid = -1, rid = -1, cid = -1;
  bool abad = false, bbad = false;
  {
    const int64_t r = m0 + srow;
    if (r < B) {
      fid = hr[2 * r]; rid = hr[2 * r + 1];
      abad = fid < 0 || fid >= N || rid < 0 || rid >= N;
    }
    const int64_t c = n0 + srow;
    if (c < K) { cid = cand[c]; bbad = cid < 0 || cid >= N; }
  }
  const bool aok = fid >= 0 && !abad, bok = cid >= 0 && !bbad;
  const float* frow = table + (int64_t)(aok ? fid : 0) * d;
  const float* rrow = table + (int64_t)(aok ? rid : 0) * d;
  const float* crow = table + (int64_t)(bok ? cid : 0) * d;
  float4 va[NITA][6], vb[NITB][6];
  // every load is unconditional (a predicated load makes the compiler serialise the whole batch behind waitcnts):
  // column groups past the row re-read the last one and are dropped in stash(); rows of bad ids / past B or K read
  // row 0 and are neutralised by their NaN clip scale or by the bounds check of the store
  auto fetch = [&](int c, float4 (&v)[6]) {
    const int cc = 4 * min(c, CG - 1);
    v[0] = *reinterpret_cast<const float4*>(frow + cc);
    v[1] = *reinterpret_cast<const float4*>(frow + k + cc);
    v[2] = *reinterpret_cast<const float4*>(rrow + cc);
    v[3] = *reinterpret_cast<const float4*>(rrow + k + cc);
    v[4] = *reinterpret_cast<const float4*>(crow + cc);
    v[5] = *reinterpret_cast<const float4*>(crow + k + cc);
  };
#pragma unroll
  for (int i = 0; i < NITA; ++i) fetch(4 * i + gq, va[i]);
#pragma unroll
  for (int i = 0; i < NITB; ++i) fetch(4 * (NITA + i) + gq, vb[i]);
  float ssf = 0.f, ssr = 0.f, ssc = 0.f;
  auto dot4 = [](const float4& a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; };
  auto stash = [&](int c, const float4 (&v)[6]) {
    if (c >= CG) return;
    ssf += dot4(v[0]) + dot4(v[1]);
    ssr += dot4(v[2]) + dot4(v[3]);
    ssc += dot4(v[4]) + dot4(v[5]);
    const float fr[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, fi[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
    const float rr[4] = {v[2].x, v[2].y, v[2].z, v[2].w}, ri[4] = {v[3].x, v[3].y, v[3].z, v[3].w};
    float qre[4], qim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!cand_is_head) {  // q = h * r
        qre[q] = fr[q] * rr[q] - fi[q] * ri[q];
        qim[q] = fr[q] * ri[q] + fi[q] * rr[q];
      } else {              // Re(h * r * conj(t)) with h the candidate: Q = [Re(r conj t) | -Im(r conj t)]
        qre[q] = rr[q] * fr[q] + ri[q] * fi[q];
        qim[q] = -(ri[q] * fr[q] - rr[q] * fi[q]);
      }
    }
    As[c * 64 + srow] = make_float4(qre[0], qre[1], qre[2], qre[3]);
    As[(CG + c) * 64 + srow] = make_float4(qim[0], qim[1], qim[2], qim[3]);
    Bs[c * 64 + srow] = v[4];
    Bs[(CG + c) * 64 + srow] = v[5];
  };
  f32x16 acc, acc2;                                  // two independent MFMA chains: real-part groups, imaginary-part groups
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
  const int li = lane & 31, lh = lane >> 5;
  // lh = 0 takes k = 0, 1 of a group, lh = 1 takes k = 2, 3: one ds_read_b64 per operand per two MFMAs
  const float2* ap = reinterpret_cast<const float2*>(As + wm * 32 + li) + lh;
  const float2* bp = reinterpret_cast<const float2*>(Bs + wn * 32 + li) + lh;
  // groups [c0, c1) of both halves (real parts g = c, imaginary parts g = CG + c).  The operands of group c + 1 are
  // requested BEFORE the four MFMAs of group c and consumed after them (scheduling barriers keep the compiler from
  // sinking the reads to their use, where their latency would sit between the MFMAs); two register sets, no copies.
  auto ldg = [&](int c, float2 (&o)[4]) {
    o[0] = ap[c * 128]; o[1] = bp[c * 128]; o[2] = ap[(CG + c) * 128]; o[3] = bp[(CG + c) * 128];
  };
  auto mm4 = [&](const float2 (&o)[4]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o[0].x, o[1].x, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(o[2].x, o[3].x, acc2, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o[0].y, o[1].y, acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(o[2].y, o[3].y, acc2, 0, 0, 0);
  };
  auto mma_range = [&](int c0, int c1) {
    if (c0 >= c1) return;
    float2 p0[4], p1[4];
    ldg(c0, p0);
    for (int c = c0; c < c1; c += 2) {
      ldg(min(c + 1, c1 - 1), p1);
      __builtin_amdgcn_sched_barrier(0);
      mm4(p0);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 < c1) {
        ldg(min(c + 2, c1 - 1), p0);
        __builtin_amdgcn_sched_barrier(0);
        mm4(p1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // batch A lands first: stash it and run its MFMAs while batch B is still in flight
#pragma unroll
  for (int i = 0; i < NITA; ++i) stash(4 * i + gq, va[i]);
  __syncthreads();
  const int ga = min(4 * NITA, CG);
  mma_range(0, ga);
#pragma unroll
  for (int i = 0; i < NITB; ++i) stash(4 * (NITA + i) + gq, vb[i]);
  ssf += __shfl_xor(ssf, 16, kWave); ssf += __shfl_xor(ssf, 32, kWave);
  ssr += __shfl_xor(ssr, 16, kWave); ssr += __shfl_xor(ssr, 32, kWave);
  ssc += __shfl_xor(ssc, 16, kWave); ssc += __shfl_xor(ssc, 32, kWave);
  if (gq == 0) {
    const float nanv = __builtin_nanf("");
    float i0, i1;
    sA[srow] = (abad || fid < 0) ? nanv : clip_scale(ssf, max_norm, i0) * clip_scale(ssr, max_norm, i1);
    sB[srow] = (bbad || cid < 0) ? nanv : clip_scale(ssc, max_norm, i0);
  }
  __syncthreads();
  mma_range(ga, CG);
  __syncthreads();                                   // every wave is done reading the operands: As becomes the output tile
  // epilogue through LDS: C layout (col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 lh) -> Cs[row][65-float stride],
  // then each lane stores 16 contiguous bytes of a row (4 store instructions per lane instead of 16)
  float* Cs = smem;
  const int cl = wn * 32 + li;
  const float sb = sB[cl];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rl = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
    const float sv = (acc[q] + acc2[q]) * sA[rl] * sb;
    Cs[rl * 65 + cl] = apply_sigmoid ? rank_sigmoid(sv) : sv;
  }
  __syncthreads();
  const bool k4 = (K & 3) == 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int rl = 16 * p + (t >> 4), c4 = 4 * (t & 15);
    const int64_t row = m0 + rl, col = n0 + c4;
    if (row >= B) continue;
    const float* src = Cs + rl * 65 + c4;
    if (k4 && col + 3 < K) *reinterpret_cast<float4*>(out + row * K + col) = make_float4(src[0], src[1], src[2], src[3]);
    else
      for (int j = 0; j < 4; ++j)
        if (col + j < K) out[row * K + col + j] = src[j];
  }
}

// The same tile on the f16 MFMA (phase stamps of the fp32 form, 4096 x 256 x 200: row loads 43 % -- the CU's 64 B/clk
// vector-memory path moving 154 KB --, the 100 fp32 MFMAs per wave 35 %, stashes 17 %, epilogue 12 %).  x * 2^8 = hi + mid
// with two fp16 values (round toward zero) is exact to 22 bits and q . t = 2^-16 (qh.th + qh.tm + qm.th) + O(2^-22) per
// product: 3 x 13 f16 MFMAs of 32 cycles instead of 100 fp32 MFMAs of 64.  Both operands are multiplied by their rows'
// clip scales BEFORE the split (|q sa| <= 2 max_norm^2, |t sb| <= max_norm: inside fp16 for max_norm <= 8 whatever the
// table holds), so all of a row's loads must have landed before its first stash -- the two-batch overlap of the fp32
// form is given up for a third of its MFMA time.  LDS: four planes [64 rows][16 KKB + 8 halves].
typedef _Float16 v1h2 __attribute__((ext_vector_type(2)));
typedef _Float16 v1h4 __attribute__((ext_vector_type(4)));
typedef _Float16 v1h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void v1_split(float x0, float x1, v1h2& hi, v1h2& mid) {
  typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
  const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(x0, x1);
  hi = __builtin_bit_cast(v1h2, h);
  const fp16x2 m = __builtin_amdgcn_cvt_pkrtz(x0 - (float)hi.x, x1 - (float)hi.y);
  mid = __builtin_bit_cast(v1h2, m);
}

template <int NIT>
__global__ __launch_bounds__(kBlock) void score_1vK_f16_kernel(
    const float* __restrict__ table, int64_t N, int d, const int32_t* __restrict__ hr, int64_t B,
    const int32_t* __restrict__ cand, int64_t K, float max_norm, int apply_sigmoid, int cand_is_head,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = d >> 1, CG = k >> 2;                 // complex column groups of 4 (d % 8 == 0)
  const int KKB = (d + 15) >> 4, SA = 16 * KKB + 8;  // k blocks of 16 columns; halves per plane row
  _Float16* Ah = reinterpret_cast<_Float16*>(smem);
  _Float16* Am = Ah + 64 * SA;
  _Float16* Bh = Am + 64 * SA;
  _Float16* Bm = Bh + 64 * SA;
  float* sA = reinterpret_cast<float*>(Bm + 64 * SA);
  float* sB = sA + 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  // staging: 4 ADJACENT lanes (gq) on 64 contiguous bytes of a row.  (With a row's four lanes 16 lanes apart every lane's 16
  // bytes were a request of their own: 11.8 -> 10.55 us per call at 4096 x 256 x 200, alternated on one box, round 4.  8 or 16
  // adjacent lanes a row -- whole 128-byte lines per request, two or four row sets a wave -- 11.9 / 13.2 us: more idle
  // lanes in the last column-group iteration, more LDS bank conflicts in the stash.)
  const int srow = 16 * w + (lane >> 2), gq = lane & 3;
  // tiles that share a row block are renumbered onto ONE XCD (score_1vK_tile_kernel)
  const int gx = gridDim.x, T = gx * (int)gridDim.y;
  const int L = (int)blockIdx.y * gx + (int)blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int V = xcd * (T >> 3) + min(xcd, T & 7) + slot;
  const int64_t m0 = (int64_t)(V / gx) * 64, n0 = (int64_t)(V % gx) * 64;
  int32_t fid = -1, rid = -1, cid = -1;
  bool abad = false, bbad = false;
  {
    const int64_t r = m0 + srow;
    if (r < B) {
      fid = hr[2 * r]; rid = hr[2 * r + 1];
      abad = fid < 0 || fid >= N || rid < 0 || rid >= N;
    }
    const int64_t c = n0 + srow;
    if (c < K) { cid = cand[c]; bbad = cid < 0 || cid >= N; }
  }
  const bool aok = fid >= 0 && !abad, bok = cid >= 0 && !bbad;
  const float* frow = table + (int64_t)(aok ? fid : 0) * d;
  const float* rrow = table + (int64_t)(aok ? rid : 0) * d;
  const float* crow = table + (int64_t)(bok ? cid : 0) * d;
  float4 v[NIT][6];
  // every load is unconditional (a predicated load makes the compiler serialise the whole batch behind waitcnts):
  // column groups past the row re-read the last one and are dropped below
  // (NIT = ceil(CG / 4) exactly, so only the LAST iteration can run past the row: the others address with immediates)
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int cc = i + 1 < NIT ? 4 * (4 * i + gq) : 4 * min(4 * i + gq, CG - 1);
    v[i][0] = *reinterpret_cast<const float4*>(frow + cc);
    v[i][1] = *reinterpret_cast<const float4*>(frow + k + cc);
    v[i][2] = *reinterpret_cast<const float4*>(rrow + cc);
    v[i][3] = *reinterpret_cast<const float4*>(rrow + k + cc);
    v[i][4] = *reinterpret_cast<const float4*>(crow + cc);
    v[i][5] = *reinterpret_cast<const float4*>(crow + k + cc);
  }
  float ssf = 0.f, ssr = 0.f, ssc = 0.f;
  auto dot4 = [](const float4& a) { return a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w; };
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    if (4 * i + gq >= CG) continue;
    ssf += dot4(v[i][0]) + dot4(v[i][1]);
    ssr += dot4(v[i][2]) + dot4(v[i][3]);
    ssc += dot4(v[i][4]) + dot4(v[i][5]);
  }
  ssf += __shfl_xor(ssf, 1, kWave); ssf += __shfl_xor(ssf, 2, kWave);
  ssr += __shfl_xor(ssr, 1, kWave); ssr += __shfl_xor(ssr, 2, kWave);
  ssc += __shfl_xor(ssc, 1, kWave); ssc += __shfl_xor(ssc, 2, kWave);
  float i0, i1;
  const float sa = clip_scale(ssf, max_norm, i0) * clip_scale(ssr, max_norm, i1) * 256.f;
  const float sb = clip_scale(ssc, max_norm, i0) * 256.f;
  _Float16* ah = Ah + srow * SA;
  _Float16* am = Am + srow * SA;
  _Float16* bh = Bh + srow * SA;
  _Float16* bm = Bm + srow * SA;
  auto put4 = [](_Float16* hi_p, _Float16* mid_p, float x0, float x1, float x2, float x3) {
    v1h2 a, b, c, e;
    v1_split(x0, x1, a, b);
    v1_split(x2, x3, c, e);
    *reinterpret_cast<v1h4*>(hi_p) = v1h4{a.x, a.y, c.x, c.y};
    *reinterpret_cast<v1h4*>(mid_p) = v1h4{b.x, b.y, e.x, e.y};
  };
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = 4 * i + gq;
    if (c >= CG) continue;
    const float fr[4] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w}, fi[4] = {v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w};
    const float rr[4] = {v[i][2].x, v[i][2].y, v[i][2].z, v[i][2].w}, ri[4] = {v[i][3].x, v[i][3].y, v[i][3].z, v[i][3].w};
    float qre[4], qim[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!cand_is_head) {  // q = h * r
        qre[q] = fr[q] * rr[q] - fi[q] * ri[q];
        qim[q] = fr[q] * ri[q] + fi[q] * rr[q];
      } else {              // Re(h * r * conj(t)) with h the candidate: Q = [Re(r conj t) | -Im(r conj t)]
        qre[q] = rr[q] * fr[q] + ri[q] * fi[q];
        qim[q] = -(ri[q] * fr[q] - rr[q] * fi[q]);
      }
    }
    put4(ah + 4 * c, am + 4 * c, qre[0] * sa, qre[1] * sa, qre[2] * sa, qre[3] * sa);
    put4(ah + k + 4 * c, am + k + 4 * c, qim[0] * sa, qim[1] * sa, qim[2] * sa, qim[3] * sa);
    put4(bh + 4 * c, bm + 4 * c, v[i][4].x * sb, v[i][4].y * sb, v[i][4].z * sb, v[i][4].w * sb);
    put4(bh + k + 4 * c, bm + k + 4 * c, v[i][5].x * sb, v[i][5].y * sb, v[i][5].z * sb, v[i][5].w * sb);
  }
  if (gq == 0) {
    for (int c = d; c < 16 * KKB; ++c) { ah[c] = (_Float16)0.f; am[c] = (_Float16)0.f; bh[c] = (_Float16)0.f; bm[c] = (_Float16)0.f; }
    const float nanv = __builtin_nanf("");
    sA[srow] = (abad || fid < 0) ? nanv : 1.0f / 65536.f;       // the two 2^8 scales; NaN: bad id / row past B
    sB[srow] = (bbad || cid < 0) ? nanv : 1.0f;
  }
  __syncthreads();
  f32x16 acc, acc2;                                  // two MFMA chains: the high x high products, the two cross terms
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc2[q] = 0.f; }
  const int li = lane & 31, lh = lane >> 5;
  const _Float16* pah = Ah + (wm * 32 + li) * SA + lh * 8;
  const _Float16* pam = Am + (wm * 32 + li) * SA + lh * 8;
  const _Float16* pbh = Bh + (wn * 32 + li) * SA + lh * 8;
  const _Float16* pbm = Bm + (wn * 32 + li) * SA + lh * 8;
  auto ldo = [&](int kb, v1h8 (&o)[4]) {
    o[0] = *reinterpret_cast<const v1h8*>(pah + 16 * kb); o[1] = *reinterpret_cast<const v1h8*>(pbh + 16 * kb);
    o[2] = *reinterpret_cast<const v1h8*>(pam + 16 * kb); o[3] = *reinterpret_cast<const v1h8*>(pbm + 16 * kb);
  };
  auto mm3 = [&](const v1h8 (&o)[4]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[0], o[1], acc, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[0], o[3], acc2, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[2], o[1], acc2, 0, 0, 0);
  };
  {
    // the operands of k block kb + 1 are requested BEFORE the MFMAs of kb and consumed after them (two register sets)
    v1h8 p0[4], p1[4];
    ldo(0, p0);
    for (int kb = 0; kb < KKB; kb += 2) {
      ldo(min(kb + 1, KKB - 1), p1);
      __builtin_amdgcn_sched_barrier(0);
      mm3(p0);
      __builtin_amdgcn_sched_barrier(0);
      if (kb + 1 < KKB) {
        ldo(min(kb + 2, KKB - 1), p0);
        __builtin_amdgcn_sched_barrier(0);
        mm3(p1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();                                   // every wave is done reading the operands: the planes become the output tile
  // epilogue through LDS: C layout (col = lane & 31, row = (q & 3) + 8 (q >> 2) + 4 lh) -> Cs[row][65-float stride],
  // then each lane stores 16 contiguous bytes of a row
  const int cl = wn * 32 + li;
  const float sbv = sB[cl];                          // (sA / sB sit behind the planes: Cs does not reach them)
  float* Cs = smem;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int rl = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
    const float sv = (acc[q] + acc2[q]) * sA[rl] * sbv;
    Cs[rl * 65 + cl] = apply_sigmoid ? rank_sigmoid(sv) : sv;
  }
  __syncthreads();
  const bool k4 = (K & 3) == 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int rl = 16 * p + (t >> 4), c4 = 4 * (t & 15);
    const int64_t row = m0 + rl, col = n0 + c4;
    if (row >= B) continue;
    const float* src = Cs + rl * 65 + c4;
    if (k4 && col + 3 < K) *reinterpret_cast<float4*>(out + row * K + col) = make_float4(src[0], src[1], src[2], src[3]);
    else
      for (int j = 0; j < 4; ++j)
        if (col + j < K) out[row * K + col + j] = src[j];
  }
}

int complex_score_1vK_launch(const float* table, int64_t N, int32_t d, const int32_t* hr, int64_t B,
                             const int32_t* cand, int64_t K, float max_norm, int apply_sigmoid,
                             int cand_is_head, float* out, hipStream_t st) {
  if (d <= 0 || (d & 1)) return GE_EINVAL;
  if (B == 0 || K == 0) return 0;
  // large sweeps with embedding_dim a multiple of 40 / 32 / 24: the software-pipelined kernel of ge_rank_pipe.hip
  // (Q resident in LDS, operands and candidate rows issued between the MFMAs) writing scores instead of counting
  if (((B + 127) / 128) * ((K + 127) / 128) >= 512 && reinterpret_cast<uintptr_t>(table) % 16 == 0) {
    const int rc = score_pipe_launch(table, N, d, hr, B, cand, K, max_norm, apply_sigmoid, cand_is_head, out, st);
    if (rc != GE_ENOTSUP) return rc;
  }
  const int k = d / 2;
  const bool v4 = (k % 4 == 0) && (reinterpret_cast<uintptr_t>(table) % 16 == 0);
  const int64_t big_blocks = ((B + 127) / 128) * ((K + 127) / 128);
  const bool big = big_blocks >= 512;
  const int bm = big ? 128 : 64;
  const int64_t gy = (B + bm - 1) / bm, gx = (K + bm - 1) / bm;
  if (gy > 65535 || gx > 2147483647LL) return GE_ENOTSUP;
  dim3 grid((unsigned)gx, (unsigned)gy);
  if (!big && d % 8 == 0 && d >= 56 && d <= 224 && max_norm <= 8.f && reinterpret_cast<uintptr_t>(table) % 16 == 0) {
    // up to a few hundred 64 x 64 tiles, split precision: all of the tile's loads in flight, f16 MFMAs
    const int kkb = (d + 15) / 16, cgq = (k / 4 + 3) / 4;        // column groups per staging lane
    const size_t lds16 = sizeof(_Float16) * (size_t)(4 * 64 * (16 * kkb + 8)) + sizeof(float) * 128;
#define LH(NIT_)                                                                                                          \
    {                                                                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_1vK_f16_kernel<NIT_>),                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      if (e != hipSuccess) return (int)e;                                                                                 \
      hipLaunchKernelGGL((score_1vK_f16_kernel<NIT_>), grid, dim3(kBlock), lds16, st, table, N, d, hr, B, cand, K,        \
                         max_norm, apply_sigmoid, cand_is_head, out);                                                     \
      return launch_status();                                                                                             \
    }
    switch (cgq) {                                               // (56 <= d <= 224: 2 ... 7 groups a lane, compiled exactly)
      case 2: LH(2) case 3: LH(3) case 4: LH(4) case 5: LH(5) case 6: LH(6) default: LH(7)
    }
#undef LH
  }
  if (!big && d % 8 == 0 && d <= 256 && reinterpret_cast<uintptr_t>(table) % 16 == 0) {
    // up to a few hundred 64 x 64 tiles: the whole tile's loads in flight in two batches, 16-byte LDS traffic
    const size_t lds = sizeof(float) * std::max<size_t>((size_t)16 * (k / 4) * 64, 64 * 65) + sizeof(float) * 128;
    const int cg = k / 4;
#define LT(NA, NB)                                                                                                        \
    {                                                                                                                     \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_1vK_tile_kernel<NA, NB>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
      if (e != hipSuccess) return (int)e;                                                                                 \
      hipLaunchKernelGGL((score_1vK_tile_kernel<NA, NB>), grid, dim3(kBlock), lds, st, table, N, d, hr, B, cand, K,       \
                         max_norm, apply_sigmoid, cand_is_head, out);                                                     \
      return launch_status();                                                                                             \
    }
    if (cg <= 8) LT(1, 1) else if (cg <= 16) LT(2, 2) else if (cg <= 28) LT(4, 3) else LT(4, 4)
#undef LT
  }
  const int KP = (k + 3) & ~3;
  const size_t fullk_lds = sizeof(float) * (size_t)(2 * 64 * (2 * KP + 1) + 128);
  if (!big && fullk_lds <= 150 * 1024) {
    // the opt-in is a per-device property: set it on every launch (idempotent, cheap) rather than cache a
    // per-process flag that would be wrong on a second device
    hipError_t e = v4 ? hipFuncSetAttribute(reinterpret_cast<const void*>(score_1vK_fullk_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                      : hipFuncSetAttribute(reinterpret_cast<const void*>(score_1vK_fullk_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (v4)
      hipLaunchKernelGGL(score_1vK_fullk_kernel<true>, grid, dim3(kBlock), fullk_lds, st, table, N, d, hr, B, cand, K, max_norm, apply_sigmoid, cand_is_head, out, KP);
    else
      hipLaunchKernelGGL(score_1vK_fullk_kernel<false>, grid, dim3(kBlock), fullk_lds, st, table, N, d, hr, B, cand, K, max_norm, apply_sigmoid, cand_is_head, out, KP);
    return launch_status();
  }
#define L1VK(TM, TN, V) \
  hipLaunchKernelGGL((score_1vK_kernel<TM, TN, V>), grid, dim3(kBlock), 0, st, table, N, d, hr, B, cand, K, max_norm, apply_sigmoid, cand_is_head, out)
  if (big) { if (v4) L1VK(2, 2, true); else L1VK(2, 2, false); }
  else { if (v4) L1VK(1, 1, true); else L1VK(1, 1, false); }
#undef L1VK
  return launch_status();
}

}  // namespace ge
