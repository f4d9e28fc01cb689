// ge_spectral.hip -- row-wise real DFT of a HolE table, in place (ge_hole_to_spectral / _from_spectral).
//
// README.md:42 scores HolE as r . ifft(conj(fft h) fft t).  ge_train_steps carries the table in the
// frequency domain instead of transforming three rows per triple per step: a row x (d reals, d even,
// k = d/2) becomes its half spectrum X_f = sum_n x_n e^{-2 pi i f n / d}, f = 0..k, packed into the same d
// floats as [Re X_0 .. Re X_{k-1} | Re X_k, Im X_1 .. Im X_{k-1}] (X_0 and X_k are real).  On that
// layout the HolE score, the max-norm clip and the SGD update are the ComplEx-shaped kernels of
// ge_complex.hip with Hermitian weights (ge_complex_dev.h, SPEC = true): zero transforms per step.
//
// One wavefront per row: the row is staged in the wave's LDS slice, lane l produces outputs l, l+64, ...
// as a direct O(d) sum against a d-entry (cos, sin) table in LDS with the phase index advanced mod d
// (exact: no accumulated angle error); twiddles come from sincospi in double and the sums run in fp64.  O(d^2) per row is
// 40 k FMA at d = 200 -- the whole FB15k table is ~0.7 GFLOP, a few tens of microseconds, paid once
// per ge_train_steps call (model 1) or once per training run (model 2, table kept spectral).
#include "ge_common.h"

namespace ge {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS: tw[d] double2 (cos, sin) | per wave: x[d] floats.  Sums run in fp64 against fp64 twiddles and are
// rounded to fp32 once: the transform pair adds no more than one rounding per element to a training run
// (fp64 FMA is half the fp32 rate on gfx950; the transforms are O(N d^2) once per call / per run).
// A lane carries two outputs through one pass over the row (one broadcast read of x_n feeds both).
template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void hole_dft_rows_kernel(float* __restrict__ table, int64_t N, int d) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  double2* tw = reinterpret_cast<double2*>(smem_d);
  const int lane = threadIdx.x & (kWave - 1), wave_in_block = threadIdx.x >> 6;
  float* xw = reinterpret_cast<float*>(smem_d + 2 * d) + wave_in_block * d;
  const int k = d >> 1;
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)d, &sn, &cs);
    tw[j] = make_double2(cs, sn);
  }
  __syncthreads();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const double invd = 1.0 / (double)d;
  for (int64_t row = wave; row < N; row += nwaves) {
    float* p = table + row * d;
    for (int c = lane; c < d; c += kWave) xw[c] = p[c];
    wave_lds_sync();
    if (!INVERSE) {
      // X_f = sum_n x_n (cos - i sin)(2 pi f n / d), outputs f0 = base + lane and f1 = f0 + 64, both <= k
      for (int base = 0; base <= k; base += 2 * kWave) {
        const int f0 = base + lane, f1 = f0 + kWave;
        const int s0 = f0 <= k ? f0 : 0, s1 = f1 <= k ? f1 : 0;     // idle halves walk f = 0 (no divergence)
        double re0 = 0.0, im0 = 0.0, re1 = 0.0, im1 = 0.0;
        int i0 = 0, i1 = 0;
#pragma unroll 4
        for (int n = 0; n < d; ++n) {
          const double x = (double)xw[n];
          const double2 w0 = tw[i0], w1 = tw[i1];
          re0 = fma(x, w0.x, re0); im0 = fma(-x, w0.y, im0);
          re1 = fma(x, w1.x, re1); im1 = fma(-x, w1.y, im1);
          i0 += s0; if (i0 >= d) i0 -= d;
          i1 += s1; if (i1 >= d) i1 -= d;
        }
        if (f0 < k) { p[f0] = (float)re0; if (f0 > 0) p[k + f0] = (float)im0; }
        else if (f0 == k) p[k] = (float)re0;                        // the Nyquist bin, real, in the Im X_0 slot
        if (f1 < k) { p[f1] = (float)re1; p[k + f1] = (float)im1; }
        else if (f1 == k) p[k] = (float)re1;
      }
    } else {
      // x_n = (1/d) [A_0 + (-1)^n A_k + 2 sum_{f=1}^{k-1} (A_f cos - B_f sin)(2 pi f n / d)]
      for (int base = 0; base < d; base += 2 * kWave) {
        const int n0 = base + lane, n1 = n0 + kWave;
        const int s0 = n0 < d ? n0 : 0, s1 = n1 < d ? n1 : 0;
        double acc0 = 0.0, acc1 = 0.0;
        int i0 = s0, i1 = s1;                                        // phase index of f = 1
#pragma unroll 4
        for (int f = 1; f < k; ++f) {
          const double a = (double)xw[f], b = (double)xw[k + f];
          const double2 w0 = tw[i0], w1 = tw[i1];
          acc0 = fma(a, w0.x, acc0); acc0 = fma(-b, w0.y, acc0);
          acc1 = fma(a, w1.x, acc1); acc1 = fma(-b, w1.y, acc1);
          i0 += s0; if (i0 >= d) i0 -= d;
          i1 += s1; if (i1 >= d) i1 -= d;
        }
        const double a0 = (double)xw[0], ak = (double)xw[k];
        if (n0 < d) p[n0] = (float)((a0 + ((n0 & 1) ? -ak : ak) + 2.0 * acc0) * invd);
        if (n1 < d) p[n1] = (float)((a0 + ((n1 & 1) ? -ak : ak) + 2.0 * acc1) * invd);
      }
    }
    wave_lds_sync();   // the next row overwrites this wave's LDS slice
  }
}

// ---------------------------------------------------------------- the same transforms on the fp64 matrix cores
// A row block of 16 rows times the d x d real-DFT matrix is a GEMM; v_mfma_f64_16x16x4_f64 keeps the arithmetic of
// the kernel above (fp64 products and sums against fp64 twiddles, one rounding to fp32 per output) at the matrix
// pipe's rate instead of two fp64 VALU FMAs per term: 11 + 12 ms -> 4.3 + 5.7 ms for the 1.2 M x 200 table, 150 + 165 us
// -> 105 + 70 us for FB15k's (tools/spectral_probe.py; a quarter of the fp64 matrix peak: every MFMA takes one twiddle
// from LDS at a data-dependent index).
//   forward:  out j <  = k : sum_n x_n  cos(2 pi j n / d)        (j = k is the Nyquist bin, in the Im X_0 slot)
//             out j >    k : sum_n x_n -sin(2 pi (j-k) n / d)
//   inverse:  out n = sum_{f=0..k} (c_f A_f) cos(2 pi f n / d) + sum_{f=1..k-1} (2/d B_f) -sin(2 pi f n / d),
//             c_0 = c_k = 1/d, else 2/d (the coefficient rides on the A operand)
// i.e. one or two SEGMENTS of a reduction index r that advances by one while the phase index (r * g) mod d advances
// by a per-output constant g (the output's frequency, or its sample number) -- exact integer phases, no angle error.
// Lane l: A[row l & 15][r = 4 step + (l >> 4)], B[r][out = 16 tile + (l & 15)] looked up in a d-entry fp64 table in
// LDS, C: col = l & 15, row = (l >> 4) + 4 reg (cdna_hip_programming.md: the f64 MFMA's own C/D map).
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int kDftTiles = 13;      // output tiles of 16 held in accumulators at once (13 x 16 = 208 >= d = 200)

template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void hole_dft_mfma_kernel(float* __restrict__ table, int64_t N, int d) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  double* cosT = smem_d;                 // cos(2 pi m / d)
  double* nsinT = smem_d + d;            // -sin(2 pi m / d)
  const int lane = threadIdx.x & (kWave - 1), wave_in_block = threadIdx.x >> 6;
  float* xw = reinterpret_cast<float*>(smem_d + 2 * d) + wave_in_block * 16 * d;     // this wave's 16 rows
  const int k = d >> 1;
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)d, &sn, &cs);
    cosT[j] = cs; nsinT[j] = -sn;
  }
  __syncthreads();
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int n_tiles = (d + 15) >> 4;
  const double invd = 1.0 / (double)d;
  for (int64_t rb = wave; rb * 16 < N; rb += nwaves) {
    const int64_t row0 = rb * 16;
    const int nrows = (int)((N - row0) < 16 ? (N - row0) : 16);
    for (int c = lane; c < 16 * d; c += kWave) xw[c] = (c / d < nrows) ? table[row0 * d + c] : 0.f;
    wave_lds_sync();
    for (int t0 = 0; t0 < n_tiles; t0 += kDftTiles) {
      f64x4 acc[kDftTiles];
#pragma unroll
      for (int t = 0; t < kDftTiles; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
      // segment 0: forward n = 0..d-1 (per-output table); inverse f = 0..k against cos.  segment 1 (inverse only):
      // f = 1..k-1 against -sin, input element k + f
      for (int seg = 0; seg < (INVERSE ? 2 : 1); ++seg) {
        const int r_lo = (INVERSE && seg == 1) ? 1 : 0;
        const int r_hi = INVERSE ? (seg == 0 ? k + 1 : k) : d;        // exclusive
        int m[kDftTiles], inc[kDftTiles];
        const double* tbl[kDftTiles];
#pragma unroll
        for (int t = 0; t < kDftTiles; ++t) {
          const int out = 16 * (t0 + t) + col;
          int g = 0;                                                   // phase advance per unit of r
          const double* tb = cosT;
          if (out < d) {
            if (INVERSE) { g = out; tb = seg == 0 ? cosT : nsinT; }
            else { g = out <= k ? out : out - k; tb = out <= k ? cosT : nsinT; }
          }
          tbl[t] = tb;
          m[t] = (int)(((int64_t)(r_lo + q) * g) % d);
          inc[t] = (4 * g) % d;
        }
        for (int r = r_lo + q; r - q < r_hi; r += 4) {
          double a = 0.0;
          if (r < r_hi) {
            if (INVERSE) {
              const float v = seg == 0 ? xw[col * d + r] : xw[col * d + k + r];
              a = (double)v * ((seg == 0 && (r == 0 || r == k)) ? invd : 2.0 * invd);
            } else {
              a = (double)xw[col * d + r];
            }
          }
#pragma unroll
          for (int t = 0; t < kDftTiles; ++t) {
            if (t0 + t < n_tiles) {
              const double b = tbl[t][m[t]];
              acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
              m[t] += inc[t];
              if (m[t] >= d) m[t] -= d;
            }
          }
        }
      }
      // C: col = lane & 15 (output), row = q + 4 reg
#pragma unroll
      for (int t = 0; t < kDftTiles; ++t) {
        const int out = 16 * (t0 + t) + col;
        if (t0 + t < n_tiles && out < d) {
          // forward: slot k + 0 does not exist (Im X_0 = 0): out = k holds the Nyquist bin, handled by the table choice
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int row = q + 4 * reg;
            if (row < nrows) table[(row0 + row) * d + out] = (float)acc[t][reg];
          }
        }
      }
    }
    wave_lds_sync();   // the next row block overwrites this wave's LDS slice
  }
}

int hole_spectral_launch(float* table, int64_t N, int32_t d, int inverse, hipStream_t st) {
  if (d <= 0 || (d & 1) || d > 1024) return (d <= 0 || (d & 1)) ? GE_EINVAL : GE_ENOTSUP;
  if (N == 0) return 0;
  // 16 rows per wave on the fp64 matrix cores; every block first builds the two d-entry fp64 twiddle tables
  const size_t lds = sizeof(double) * 2 * (size_t)d + sizeof(float) * (size_t)((kBlock / kWave) * 16 * d);
  if (lds <= 150 * 1024) {
    int64_t g = ((N + 15) / 16 + (kBlock / kWave) - 1) / (kBlock / kWave);
    if (g > 1024) g = 1024;
    hipError_t e = inverse ? hipFuncSetAttribute(reinterpret_cast<const void*>(hole_dft_mfma_kernel<true>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(hole_dft_mfma_kernel<false>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    if (inverse) hipLaunchKernelGGL(hole_dft_mfma_kernel<true>, dim3((unsigned)g), dim3(kBlock), lds, st, table, N, d);
    else hipLaunchKernelGGL(hole_dft_mfma_kernel<false>, dim3((unsigned)g), dim3(kBlock), lds, st, table, N, d);
    return launch_status();
  }
  // very wide rows (16 x d floats per wave do not fit LDS): the wave-per-row kernel
  const size_t lds1 = sizeof(double) * 2 * (size_t)d + sizeof(float) * (size_t)((kBlock / kWave) * d);
  int64_t g = (N + (kBlock / kWave) - 1) / (kBlock / kWave);
  if (g > 1024) g = 1024;
  const int grid = (int)g;
  if (inverse) hipLaunchKernelGGL(hole_dft_rows_kernel<true>, dim3(grid), dim3(kBlock), lds1, st, table, N, d);
  else hipLaunchKernelGGL(hole_dft_rows_kernel<false>, dim3(grid), dim3(kBlock), lds1, st, table, N, d);
  return launch_status();
}

}  // namespace ge
