// r2c_fft.hpp -- r2c of the projections / particle maps as a fast transform in double (round 4)
// Part of libbioem_hip.so; included by kernels_r2c.hip only (anonymous namespace).
//
// What the reference takes from FFTW (fftwf_plan_dft_r2c_2d, bioem.cpp:1848, map.cpp:585) and rounds 1-3 computed as
// an exact DFT (prep_kernels.hpp, dft_mfma.hpp: N (A + B) products per transform) is here one Cooley-Tukey split
// N = A * B whose two sub-transforms are register-resident mixed-radix FFTs, one per thread:
//   stage 1   thread (g, j1):  Y[j1][kb] = w_N^(j1 kb) * FFT_B over j2 of z[A j2 + j1]
//   LDS       Y, one exchange per transform
//   stage 2   thread (g, kb):  Z[kb + B m] = FFT_A over j1 of Y[j1][kb]
// Row pass: two image rows per complex transform (z = row_i + i row_(i+1)), separated on the way out
// (X_i[k] = (Z[k] + conj Z[N-k]) / 2, X_(i+1)[k] = (Z[k] - conj Z[N-k]) / 2i); its result is left transposed,
// spec[b][k][i], as the matrix kernels leave it (i over the rows that are not zero: projections arrive as the box of
// pixels the model can reach, a third of the map's side at BASELINE's sizes, and everything outside it is skipped --
// rows not transformed, zeros not moved).  Column pass: one spectrum column per transform, result rounded to
// float in the reference layout out[b][u][k].  Everything between the float input and the float output is double, so
// the result is the correctly rounded spectrum but for ~1e-16 relative (the DFT kernels: the same); the two paths agree
// to the last float bit in all but a few values per million.
// Sub-transform lengths 2...20 are compiled (one switch per stage, uniform over the launch): every N = A * B with
// B <= 20, i.e. all sizes up to 400 pixels whose largest prime factor is at most 19.
#ifndef BIOEM_R2C_FFT_HPP
#define BIOEM_R2C_FFT_HPP

namespace
{

constexpr int kR2cMaxLen = 20;
constexpr int kR2cThreads = 256;
constexpr int kR2cLdsBudget = 53 * 1024; // three blocks per CU

// ---- compile-time roots of unity, reduced to the first octant before the series ----
struct CisD
{
  double c, s;
};
constexpr double r2c_pi = 3.14159265358979323846264338327950288;
constexpr CisD r2c_cis(int k, int L) // exp(+2 pi i k / L)
{
  int n = 8 * (((k % L) + L) % L);
  const int T = 8 * L;
  bool negS = false, negC = false, swp = false;
  if (n > T / 2)
  {
    n = T - n;
    negS = true;
  }
  if (n > T / 4)
  {
    n = T / 2 - n;
    negC = true;
  }
  if (n > T / 8)
  {
    n = T / 4 - n;
    swp = true;
  }
  const double x = 2.0 * r2c_pi * (double) n / (double) T; // <= pi / 4
  double c = 1.0, s = x, tc = 1.0, ts = x;
  for (int j = 1; j < 14; j++)
  {
    tc *= -x * x / (double) ((2 * j - 1) * (2 * j));
    ts *= -x * x / (double) ((2 * j) * (2 * j + 1));
    c += tc;
    s += ts;
  }
  if (n == 0)
  {
    c = 1.0;
    s = 0.0;
  }
  if (swp)
  {
    const double t = c;
    c = s;
    s = t;
  }
  return CisD{negC ? -c : c, negS ? -s : s};
}
template <int L>
struct CisTable
{
  double c[L], s[L];
  constexpr CisTable() : c(), s()
  {
    for (int k = 0; k < L; k++)
    {
      const CisD w = r2c_cis(k, L);
      c[k] = w.c;
      s[k] = w.s;
    }
  }
};
template <int L>
__device__ constexpr CisTable<L> CIS = CisTable<L>();

constexpr int r2c_radix(int L)
{
  if (L % 4 == 0)
    return 4;
  for (int p = 2; p < L; p++)
    if (L % p == 0)
      return p;
  return L;
}

// forward DFT of P points in place (P = 2, 4 or odd)
template <int P>
__device__ __forceinline__ void r2c_butterfly(double (&tr)[P], double (&ti)[P])
{
  if constexpr (P == 2)
  {
    const double ar = tr[0], ai = ti[0];
    tr[0] = ar + tr[1];
    ti[0] = ai + ti[1];
    tr[1] = ar - tr[1];
    ti[1] = ai - ti[1];
  }
  else if constexpr (P == 4)
  {
    const double ar = tr[0] + tr[2], ai = ti[0] + ti[2], br = tr[0] - tr[2], bi = ti[0] - ti[2];
    const double cr = tr[1] + tr[3], ci = ti[1] + ti[3], dr = tr[1] - tr[3], di = ti[1] - ti[3];
    tr[0] = ar + cr;
    ti[0] = ai + ci;
    tr[2] = ar - cr;
    ti[2] = ai - ci;
    tr[1] = br + di; // b - i d
    ti[1] = bi - dr;
    tr[3] = br - di; // b + i d
    ti[3] = bi + dr;
  }
  else
  {
    constexpr int Q = (P - 1) / 2;
    double ar[Q], ai[Q], dr[Q], di[Q];
#pragma unroll
    for (int q = 0; q < Q; q++)
    {
      ar[q] = tr[q + 1] + tr[P - 1 - q];
      ai[q] = ti[q + 1] + ti[P - 1 - q];
      dr[q] = tr[q + 1] - tr[P - 1 - q];
      di[q] = ti[q + 1] - ti[P - 1 - q];
    }
    const double x0r = tr[0], x0i = ti[0];
    double sr = x0r, si = x0i;
#pragma unroll
    for (int q = 0; q < Q; q++)
    {
      sr += ar[q];
      si += ai[q];
    }
    tr[0] = sr;
    ti[0] = si;
#pragma unroll
    for (int r = 1; r <= Q; r++)
    {
      double er = x0r, ei = x0i, fr = 0., fi = 0.;
#pragma unroll
      for (int q = 0; q < Q; q++)
      {
        const int t = ((q + 1) * r) % P;
        er = __builtin_fma(CIS<P>.c[t], ar[q], er);
        ei = __builtin_fma(CIS<P>.c[t], ai[q], ei);
        fr = __builtin_fma(CIS<P>.s[t], dr[q], fr);
        fi = __builtin_fma(CIS<P>.s[t], di[q], fi);
      }
      tr[r] = er + fi; // e - i f
      ti[r] = ei - fr;
      tr[P - r] = er - fi; // e + i f
      ti[P - r] = ei + fr;
    }
  }
}

// forward FFT of L points in registers, natural order in and out (decimation in time, radix 4 / 2 / odd primes)
template <int L>
__device__ __forceinline__ void r2c_fft_fwd(double (&re)[L], double (&im)[L])
{
  if constexpr (L > 1)
  {
    constexpr int P = r2c_radix(L), M = L / P;
    if constexpr (M == 1)
      r2c_butterfly<P>(re, im);
    else
    {
      double sr[P][M], si[P][M];
#pragma unroll
      for (int q = 0; q < P; q++)
      {
#pragma unroll
        for (int j = 0; j < M; j++)
        {
          sr[q][j] = re[P * j + q];
          si[q][j] = im[P * j + q];
        }
        r2c_fft_fwd<M>(sr[q], si[q]);
      }
#pragma unroll
      for (int k = 0; k < M; k++)
      {
        double tr[P], ti[P];
#pragma unroll
        for (int q = 0; q < P; q++)
        {
          const int t = (q * k) % L; // times exp(-2 pi i t / L)
          const double vr = sr[q][k], vi = si[q][k];
          if (t == 0)
          {
            tr[q] = vr;
            ti[q] = vi;
          }
          else if (4 * t == L)
          { // -i
            tr[q] = vi;
            ti[q] = -vr;
          }
          else if (2 * t == L)
          {
            tr[q] = -vr;
            ti[q] = -vi;
          }
          else if (4 * t == 3 * L)
          { // +i
            tr[q] = -vi;
            ti[q] = vr;
          }
          else
          {
            const double c = CIS<L>.c[t], s = CIS<L>.s[t];
            tr[q] = __builtin_fma(s, vi, c * vr);
            ti[q] = __builtin_fma(-s, vr, c * vi);
          }
        }
        r2c_butterfly<P>(tr, ti);
#pragma unroll
        for (int r = 0; r < P; r++)
        {
          re[k + M * r] = tr[r];
          im[k + M * r] = ti[r];
        }
      }
    }
  }
}

struct R2cArgs
{
  const double *srcD;    // row pass: double maps (projections), scaled by NormDen / tempden[b] in float ...
  const float *srcF;     // ... or float maps (particles)
  const double *tempden;
  float NormDen;
  const double2 *specIn; // column pass: the row pass' result [b][k][i]
  double2 *specOut;      // row pass
  float2 *out;           // column pass: reference layout [b][u][k]
  const double2 *twD;    // exp(+2 pi i j / N), j < N
  int N, H, A, B, nImg;
  // the input is zero outside the square [lo, lo + side)^2 and srcD holds that square only, [b][side][side] (the box of
  // k_project_box; lo = 0, side = N: whole maps): the row pass transforms the `side` rows that are not zero, the
  // intermediate is [b][k][side], the column pass reads zeros for the other rows
  int lo, side;
  int G;                 // transforms per unit (block iteration)
  int YG;                // doubles per transform in a Y plane, = B (mod 32): stage 2 reads without bank conflicts
  int Gp;                // odd row length of the transposed Z planes
  int plane;             // doubles per LDS plane, max(G * YG, N * Gp)
};

// stage 1 of transform `item` (block-local g), sub-sequence j1.  (A group's transforms beyond the last item take the
// last item again: their results are never stored.)
template <int B, bool ROWS>
__device__ __forceinline__ void r2c_stage1(const R2cArgs &a, const double2 *tw, double *Yr, double *Yi, int g, int j1,
                                           long item, long nItems)
{
  double re[B], im[B];
  const int N = a.N, A = a.A;
  item = item < nItems ? item : nItems - 1;
  const int lo = a.lo, side = a.side;
  if constexpr (ROWS)
  {
    const int NP = (side + 1) >> 1; // pairs of rows lo + 2 r, lo + 2 r + 1
    const int b = (int) (item / NP), i = 2 * (int) (item - (long) b * NP);
    const int second = i + 1 < side ? side : 0; // odd: the last row is paired with itself, the copy is dropped on the way out
    const size_t r0 = ((size_t) b * side + i) * side;
    if (a.srcD)
    {
      const float ratio = a.NormDen / (float) a.tempden[b]; // bioem.cpp:1808-1818
#pragma unroll
      for (int j2 = 0; j2 < B; j2++)
      {
        const int j = A * j2 + j1 - lo;
        const bool in = (unsigned) j < (unsigned) side;
        const size_t e = r0 + (in ? j : 0);
        const float v0 = (float) a.srcD[e] * ratio, v1 = (float) a.srcD[e + second] * ratio;
        re[j2] = in ? (double) v0 : 0.;
        im[j2] = in ? (double) v1 : 0.;
      }
    }
    else
    {
#pragma unroll
      for (int j2 = 0; j2 < B; j2++)
      {
        const int j = A * j2 + j1 - lo;
        const bool in = (unsigned) j < (unsigned) side;
        const size_t e = r0 + (in ? j : 0);
        re[j2] = in ? (double) a.srcF[e] : 0.;
        im[j2] = in ? (double) a.srcF[e + second] : 0.;
      }
    }
  }
  else
  {
    const size_t r0 = (size_t) item * side; // item = b * H + k
#pragma unroll
    for (int j2 = 0; j2 < B; j2++)
    {
      const int i = A * j2 + j1 - lo;
      const bool in = (unsigned) i < (unsigned) side;
      const double2 v = a.specIn[r0 + (in ? i : 0)];
      re[j2] = in ? v.x : 0.;
      im[j2] = in ? v.y : 0.;
    }
  }
  r2c_fft_fwd<B>(re, im);
  double *yr = Yr + g * a.YG + j1 * (B + 1), *yi = Yi + g * a.YG + j1 * (B + 1);
  yr[0] = re[0];
  yi[0] = im[0];
#pragma unroll
  for (int kb = 1; kb < B; kb++)
  {
    const double2 w = tw[j1 * kb]; // < N
    yr[kb] = __builtin_fma(w.y, im[kb], w.x * re[kb]); // times conj(w)
    yi[kb] = __builtin_fma(-w.y, re[kb], w.x * im[kb]);
  }
}

// stage 2 of block-local transform g, residue kb; Z is written over Y once every thread of the block has read its inputs
template <int A>
__device__ __forceinline__ void r2c_stage2(const R2cArgs &a, double *Pr, double *Pi, int g, int kb, bool active)
{
  double re[A], im[A];
  const int B = a.B;
  if (active)
  {
#pragma unroll
    for (int j1 = 0; j1 < A; j1++)
    {
      re[j1] = Pr[g * a.YG + j1 * (B + 1) + kb];
      im[j1] = Pi[g * a.YG + j1 * (B + 1) + kb];
    }
    r2c_fft_fwd<A>(re, im);
  }
  __syncthreads();
  if (active)
  {
#pragma unroll
    for (int m = 0; m < A; m++)
    {
      Pr[(kb + B * m) * a.Gp + g] = re[m];
      Pi[(kb + B * m) * a.Gp + g] = im[m];
    }
  }
}

#define R2C_LENGTHS(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20)

// (Tried: the stage-1 inputs of the next unit fetched while this one goes through stage 2 and the stores -- 64 more
// registers.  Alone, 384 images of 224^2: rows 81 -> 65 us, columns 61 -> 71 us; beside a comparison kernel the 191
// registers do not fit the slot a retiring comparison block leaves, see kernels_r2c.hip.)
template <bool ROWS, int LMAX, int NBLK>
__global__ __launch_bounds__(kR2cThreads, NBLK) void k_r2c_fft(const R2cArgs a)
{
  extern __shared__ double sm[];
  double2 *tw = reinterpret_cast<double2 *>(sm); // [N]
  double *Pr = sm + 2 * a.N, *Pi = Pr + a.plane;
  const int t = threadIdx.x;
  const int N = a.N, H = a.H, A = a.A, B = a.B, G = a.G;
  for (int j = t; j < N; j += kR2cThreads)
    tw[j] = a.twD[j];
  const int perImg = ROWS ? (a.side + 1) >> 1 : H;
  const long nItems = (long) a.nImg * perImg;
  const int nUnits = (int) ((nItems + G - 1) / G);
  const int g1h = t / A, j1h = t - g1h * A;
  const int g2h = t / B, kbh = t - g2h * B;
  // a resident grid walks the units: block launch and twiddle table are paid once
  for (int unit = blockIdx.x; unit < nUnits; unit += gridDim.x)
  {
    const long item0 = (long) unit * G;
    // opaque copies: the address arithmetic of every length (19 cases per stage) must not be hoisted out of this loop
    int g1 = g1h, j1 = j1h, g2 = g2h, kb = kbh;
    asm volatile("" : "+v"(g1), "+v"(j1), "+v"(g2), "+v"(kb));
    __syncthreads(); // Z of the previous unit is consumed (first unit: the twiddle table is complete)
    if (g1 < G)
    {
      switch (B)
      {
#define X(L)                                                                                                            \
  case L:                                                                                                               \
    if constexpr (L <= LMAX)                                                                                            \
      r2c_stage1<L, ROWS>(a, tw, Pr, Pi, g1, j1, item0 + g1, nItems);                                                                 \
    break;
        R2C_LENGTHS(X)
#undef X
      }
    }
    __syncthreads();
    switch (A)
    {
#define X(L)                                                                                                            \
  case L:                                                                                                               \
    if constexpr (L <= LMAX)                                                                                            \
      r2c_stage2<L>(a, Pr, Pi, g2, kb, g2 < G);                                                                         \
    break;
      R2C_LENGTHS(X)
#undef X
    }
    __syncthreads();
    if constexpr (ROWS)
    {
      const int NP = (a.side + 1) >> 1;
      for (int idx = t; idx < G * H; idx += kR2cThreads)
      {
        const int k = idx / G, g = idx - k * G;
        const long item = item0 + g;
        if (item >= nItems)
          continue;
        const int b = (int) (item / NP), i = 2 * (int) (item - (long) b * NP);
        const int k2 = k ? N - k : 0;
        const double zr = Pr[k * a.Gp + g], zi = Pi[k * a.Gp + g];
        const double yr = Pr[k2 * a.Gp + g], yi = Pi[k2 * a.Gp + g];
        double2 *dst = a.specOut + ((size_t) b * H + k) * a.side + i;
        dst[0] = make_double2(0.5 * (zr + yr), 0.5 * (zi - yi));
        if (i + 1 < a.side)
          dst[1] = make_double2(0.5 * (zi + yi), 0.5 * (yr - zr));
      }
    }
    else
    {
      for (int idx = t; idx < G * N; idx += kR2cThreads)
      {
        const int u = idx / G, g = idx - u * G;
        const long item = item0 + g;
        if (item >= nItems)
          continue;
        const int b = (int) (item / H), k = (int) (item - (long) b * H);
        a.out[((size_t) b * N + u) * H + k] = make_float2((float) Pr[u * a.Gp + g], (float) Pi[u * a.Gp + g]);
      }
    }
  }
}

} // namespace

#endif
