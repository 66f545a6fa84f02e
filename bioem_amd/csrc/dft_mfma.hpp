// dft_mfma.hpp -- r2c of the projections / particle maps on the fp64 matrix cores (v_mfma_f64_16x16x4_f64)
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
//
// Same arithmetic as k_dft_rows / k_dft_cols of prep_kernels.hpp (one Cooley-Tukey split N = A*B, both sub-transforms as
// plain sums in double, table twiddles), with the sums issued as 16 x 16 x 4 matrix products:
//   M = 16 outputs of one sub-transform, N = 16 image rows (row pass) or 16 spectrum columns (column pass), K = its
//   inputs.
// A block keeps its 16 rows / columns and the twiddle table in LDS; the first pass stays in registers until every wave is
// done with the input, then Y takes the input's place (272 N and 288 N bytes: two blocks per CU at 224 and 256 pixels).
// Images beyond kDftMfmaMaxN pixels take the vector kernels.
// The row pass leaves its result transposed, spec[b][k][i], so that both passes read and write 128-byte runs.
#ifndef BIOEM_DFT_MFMA_HPP
#define BIOEM_DFT_MFMA_HPP

namespace
{

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kDftMfmaMaxN = 304;
constexpr int kDftMfmaThreads = 512; // 8 waves; the index arithmetic below relies on it
constexpr int kDftMfmaWaves = kDftMfmaThreads / 64;
// first-pass tiles a wave keeps in registers (template T = 2 or 5): A * ceil(B / 16) <= 40 up to 304 pixels, <= 16 for
// the usual 16 x 14, 16 x 16 splits
constexpr int kDftMfmaTiles = 5;
// LDS: the twiddle table, then the 16 rows (float, row stride 17) / 16 columns (two double planes, stride 17), which the
// intermediate Y (two double planes, stride 16) overwrites once the first pass sits in registers
inline size_t dft_mfma_rows_lds(int N) { return (size_t) N * (sizeof(double2) + 2 * 16 * sizeof(double)); }
inline size_t dft_mfma_cols_lds(int N) { return (size_t) N * (sizeof(double2) + 2 * 17 * sizeof(double)); }
inline bool dft_mfma_fits(int N, int A, int B)
{
  return N <= kDftMfmaMaxN && A * ((B + 15) / 16) <= kDftMfmaTiles * kDftMfmaWaves;
}

// D += conj(w) * y over `cnt` inputs: w = tw[(idx + ks * step4) mod N] gathered per lane (A operand: output row lane&15,
// input 4 ks + lane>>4), y = (Pr, Pi)[(base + kk * kstride) * ps + lane&15]
__device__ inline void dft_tile_complex(const double2 *tw, int N, int idx, int step4, bool okm, int cnt,
                                        const double *Pr, const double *Pi, int base, int kstride, int ps, int q, int n,
                                        double4_t &Dr, double4_t &Di)
{
  for (int k0 = 0; k0 < cnt; k0 += 4)
  {
    const int kk = k0 + q;
    const bool okk = kk < cnt;
    const double2 w = tw[idx];
    const double wr = (okm && okk) ? w.x : 0., wi = (okm && okk) ? w.y : 0.;
    const int e = okk ? (base + kk * kstride) * ps + n : n;
    double yr = Pr[e], yi = Pi[e];
    yr = okk ? yr : 0.;
    yi = okk ? yi : 0.;
    Dr = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, yr, Dr, 0, 0, 0);
    Di = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, yi, Di, 0, 0, 0);
    Dr = __builtin_amdgcn_mfma_f64_16x16x4f64(wi, yi, Dr, 0, 0, 0);
    Di = __builtin_amdgcn_mfma_f64_16x16x4f64(-wi, yr, Di, 0, 0, 0);
    idx += step4;
    idx = idx >= N ? idx - N : idx;
  }
}

// first-pass tile t = (j1, kt) -> Y[j1][16 kt + ...] rows of the accumulator
__device__ inline void dft_store_y(double *Yr, double *Yi, int B, int j1, int kt, int q, int n, const double4_t &Dr,
                                   const double4_t &Di)
{
  for (int r = 0; r < 4; r++)
  {
    const int kbo = kt * 16 + q + 4 * r;
    if (kbo < B)
    {
      Yr[(j1 * B + kbo) * 16 + n] = Dr[r];
      Yi[(j1 * B + kbo) * 16 + n] = Di[r];
    }
  }
}

// rows i0 .. i0+15 of image b: x -> Y[j1][kb] -> spec[b][k][i], k < H
template <int T>
__global__ __launch_bounds__(kDftMfmaThreads, T <= 2 ? 4 : 2) void
k_dft_rows_mfma(const double *__restrict__ srcD, const float *__restrict__ srcF, const double *__restrict__ tempden,
                float NormDen, int N, int H, int A, int B, int nImg, const double2 *__restrict__ twD,
                double2 *__restrict__ spec)
{
  extern __shared__ double sm[];
  double2 *tw = reinterpret_cast<double2 *>(sm); // [N]
  float *xs = reinterpret_cast<float *>(sm + 2 * N); // [N][17]
  double *Yr = sm + 2 * N;                            // [A*B][16], over xs
  double *Yi = Yr + 16 * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  for (int j = threadIdx.x; j < N; j += kDftMfmaThreads)
    tw[j] = twD[j];
  const int nG = (N + 15) >> 4;
  // a resident grid walks the (image, 16 rows) units: the block launch and the twiddle table are paid once
  for (int unit = blockIdx.x; unit < nImg * nG; unit += gridDim.x)
  {
    const int b = unit / nG, i0 = (unit - b * nG) * 16;
    __syncthreads(); // Y of the previous unit is consumed
    {
      const int c = threadIdx.x >> 5, i = i0 + c;
      float ratio = 1.f;
      if (srcD)
        ratio = NormDen / (float) tempden[b]; // bioem.cpp:1808-1818
      for (int j = threadIdx.x & 31; j < N; j += 32)
      {
        float v = 0.f;
        if (i < N)
        {
          if (srcD)
          {
            v = (float) srcD[((size_t) b * N + i) * N + j];
            v = v * ratio;
          }
          else
            v = srcF[((size_t) b * N + i) * N + j];
        }
        xs[j * 17 + c] = v;
      }
    }
    __syncthreads();
    // Y[j1][kb] = sum_j2 x[A j2 + j1] conj(w_B^(j2 kb)),  w_B^(j2 kb) = tw[(A kb j2) mod N]
    const int nKT = (B + 15) >> 4;
    double4_t Dr[T], Di[T];
#pragma unroll
    for (int s = 0; s < T; s++)
    {
      Dr[s] = double4_t{0., 0., 0., 0.};
      Di[s] = double4_t{0., 0., 0., 0.};
      const int t = wave + kDftMfmaWaves * s;
      if (t < A * nKT)
      {
        const int j1 = t / nKT, kt = t - j1 * nKT;
        const int kb = kt * 16 + n;
        const bool okm = kb < B;
        const int step = okm ? A * kb : 0; // < N
        int idx = (step * q) % N;
        const int step4 = (4 * step) % N;
        for (int k0 = 0; k0 < B; k0 += 4)
        {
          const int j2 = k0 + q;
          const bool okk = j2 < B;
          const double2 w = tw[idx];
          const double wr = (okm && okk) ? w.x : 0., wi = (okm && okk) ? -w.y : 0.;
          float xf = xs[okk ? (A * j2 + j1) * 17 + n : n];
          const double x = okk ? (double) xf : 0.;
          Dr[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(wr, x, Dr[s], 0, 0, 0);
          Di[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(wi, x, Di[s], 0, 0, 0);
          idx += step4;
          idx = idx >= N ? idx - N : idx;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < T; s++)
    {
      const int t = wave + kDftMfmaWaves * s;
      if (t < A * nKT)
        dft_store_y(Yr, Yi, B, t / nKT, t % nKT, q, n, Dr[s], Di[s]);
    }
    __syncthreads();
    // X[kb + B m] = sum_j1 Y[j1][kb] conj(w_N^(j1 (kb + B m)))
    const int nMT = (A + 15) >> 4;
    for (int t = wave; t < B * nMT; t += kDftMfmaWaves)
    {
      const int kb = t / nMT, mt = t - kb * nMT;
      if (kb + B * (mt * 16) >= H)
        continue;
      const int m = mt * 16 + n;
      const bool okm = m < A;
      const int k = okm ? kb + B * m : 0; // < N
      double4_t Er = {0., 0., 0., 0.}, Ei = {0., 0., 0., 0.};
      dft_tile_complex(tw, N, (k * q) % N, (4 * k) % N, okm, A, Yr, Yi, kb, B, 16, q, n, Er, Ei);
      for (int r = 0; r < 4; r++)
      {
        const int mo = mt * 16 + q + 4 * r, ko = kb + B * mo;
        if (mo < A && ko < H && i0 + n < N)
          spec[((size_t) b * H + ko) * N + i0 + n] = make_double2(Er[r], Ei[r]);
      }
    }
  }
}

// spectrum columns k0 .. k0+15 of image b: spec[b][k][.] -> out[b][u][k] (reference layout, float)
template <int T>
__global__ __launch_bounds__(kDftMfmaThreads, T <= 2 ? 4 : 2) void
k_dft_cols_mfma(const double2 *__restrict__ spec, int N, int H, int A, int B, int nImg,
                const double2 *__restrict__ twD, float2 *__restrict__ out)
{
  extern __shared__ double sm[];
  double2 *tw = reinterpret_cast<double2 *>(sm); // [N]
  double *Cr = sm + 2 * N;                       // [N][17]
  double *Ci = Cr + 17 * N;
  double *Yr = sm + 2 * N;                       // [A*B][16], over C
  double *Yi = Yr + 16 * N;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  for (int j = threadIdx.x; j < N; j += kDftMfmaThreads)
    tw[j] = twD[j];
  const int nG = (H + 15) >> 4;
  for (int unit = blockIdx.x; unit < nImg * nG; unit += gridDim.x)
  {
    const int b = unit / nG, k0 = (unit - b * nG) * 16;
    __syncthreads(); // Y of the previous unit is consumed
    {
      const int c = threadIdx.x >> 5;
      for (int i = threadIdx.x & 31; i < N; i += 32)
      {
        double2 v = make_double2(0., 0.);
        if (k0 + c < H)
          v = spec[((size_t) b * H + k0 + c) * N + i];
        Cr[i * 17 + c] = v.x;
        Ci[i * 17 + c] = v.y;
      }
    }
    __syncthreads();
    const int nKT = (B + 15) >> 4;
    double4_t Dr[T], Di[T];
#pragma unroll
    for (int s = 0; s < T; s++)
    {
      Dr[s] = double4_t{0., 0., 0., 0.};
      Di[s] = double4_t{0., 0., 0., 0.};
      const int t = wave + kDftMfmaWaves * s;
      if (t < A * nKT)
      {
        const int j1 = t / nKT, kt = t - j1 * nKT;
        const int kb = kt * 16 + n;
        const bool okm = kb < B;
        const int step = okm ? A * kb : 0;
        dft_tile_complex(tw, N, (step * q) % N, (4 * step) % N, okm, B, Cr, Ci, j1, A, 17, q, n, Dr[s], Di[s]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < T; s++)
    {
      const int t = wave + kDftMfmaWaves * s;
      if (t < A * nKT)
        dft_store_y(Yr, Yi, B, t / nKT, t % nKT, q, n, Dr[s], Di[s]);
    }
    __syncthreads();
    const int nMT = (A + 15) >> 4;
    for (int t = wave; t < B * nMT; t += kDftMfmaWaves)
    {
      const int kb = t / nMT, mt = t - kb * nMT;
      const int m = mt * 16 + n;
      const bool okm = m < A;
      const int u = okm ? kb + B * m : 0;
      double4_t Er = {0., 0., 0., 0.}, Ei = {0., 0., 0., 0.};
      dft_tile_complex(tw, N, (u * q) % N, (4 * u) % N, okm, A, Yr, Yi, kb, B, 16, q, n, Er, Ei);
      for (int r = 0; r < 4; r++)
      {
        const int mo = mt * 16 + q + 4 * r, uo = kb + B * mo;
        if (mo < A && k0 + n < H)
          out[(size_t) b * N * H + (size_t) uo * H + k0 + n] = make_float2((float) Er[r], (float) Ei[r]);
      }
    }
  }
}

} // namespace

#endif
