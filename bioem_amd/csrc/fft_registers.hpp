// fft_registers.hpp -- register-resident inverse FFTs (radix 2 for 2..32, mixed radix 2/3/5 for 6..30) and load helpers
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_FFT_REGISTERS_HPP
#define BIOEM_FFT_REGISTERS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// exp(+2 pi i t / 32), t = 0..15: twiddles of the radix-2 register FFTs
// ------------------------------------------------------------------------------------------------
__device__ constexpr float COS32[16] = {1.0f,
                                        0.98078528040323044913f,
                                        0.92387953251128675613f,
                                        0.83146961230254523708f,
                                        0.70710678118654752440f,
                                        0.55557023301960222474f,
                                        0.38268343236508977173f,
                                        0.19509032201612826785f,
                                        0.0f,
                                        -0.19509032201612826785f,
                                        -0.38268343236508977173f,
                                        -0.55557023301960222474f,
                                        -0.70710678118654752440f,
                                        -0.83146961230254523708f,
                                        -0.92387953251128675613f,
                                        -0.98078528040323044913f};
__device__ constexpr float SIN32[16] = {0.0f,
                                        0.19509032201612826785f,
                                        0.38268343236508977173f,
                                        0.55557023301960222474f,
                                        0.70710678118654752440f,
                                        0.83146961230254523708f,
                                        0.92387953251128675613f,
                                        0.98078528040323044913f,
                                        1.0f,
                                        0.98078528040323044913f,
                                        0.92387953251128675613f,
                                        0.83146961230254523708f,
                                        0.70710678118654752440f,
                                        0.55557023301960222474f,
                                        0.38268343236508977173f,
                                        0.19509032201612826785f};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 as_float2(u32x2 v) { return make_float2(__uint_as_float(v.x), __uint_as_float(v.y)); }

__device__ __forceinline__ float4 as_float4(u32x4 v)
{
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// R-point inverse FFT (R = 8, 16, 32), radix-2 decimation in time: input must be stored at the bit-reversed
// position (element k2 at index bitrev<R>(k2)), output in natural order.  Twiddled butterflies use
//   out0 = a + w*b  (4 FMAs),  out1 = 2a - out0  (2 FMAs)
// i.e. 6 instead of 8 operations.
template <int R>
__host__ __device__ constexpr int bitrevR(int n)
{
  int r = 0;
  for (int b = 1, c = R >> 1; b < R; b <<= 1, c >>= 1)
    if (n & b)
      r |= c;
  return r;
}

template <int R>
__device__ __forceinline__ void fft_inverse_dit(float (&xr)[R], float (&xi)[R])
{
  constexpr int LOG2R = (R == 32) ? 5 : (R == 16) ? 4 : (R == 8) ? 3 : (R == 4) ? 2 : 1;
#pragma unroll
  for (int s = 0; s < LOG2R; s++)
  {
    const int m = 1 << s;
#pragma unroll
    for (int b = 0; b < R; b += 2 * m)
    {
#pragma unroll
      for (int j = 0; j < m; j++)
      {
        const int i0 = b + j, i1 = b + j + m;
        const int t = j * (16 >> s); // w_32^t = exp(+2 pi i j / (2m))
        const float ar = xr[i0], ai = xi[i0], br = xr[i1], bi = xi[i1];
        if (t == 0)
        {
          xr[i0] = ar + br;
          xi[i0] = ai + bi;
          xr[i1] = ar - br;
          xi[i1] = ai - bi;
        }
        else if (t == 8)
        { // w = i: w*b = (-bi, br)
          xr[i0] = ar - bi;
          xi[i0] = ai + br;
          xr[i1] = ar + bi;
          xi[i1] = ai - br;
        }
        else
        {
          const float c = COS32[t], sn = SIN32[t];
          const float o0r = fmaf(-sn, bi, fmaf(c, br, ar));
          const float o0i = fmaf(sn, br, fmaf(c, bi, ai));
          xr[i0] = o0r;
          xi[i0] = o0i;
          xr[i1] = fmaf(2.0f, ar, -o0r);
          xi[i1] = fmaf(2.0f, ai, -o0i);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Mixed-radix register FFTs for lengths R = 2^a 3^b 5^c <= 32 (6, 10, 12, 18, 20, 30): decimation in time,
// in place, stages in the order 5.., 3.., 2.. (small spans first, so the odd radices need few twiddles).  Input
// element k2 must be stored at digitrevR<R>(k2); output in natural order.  Sign + (inverse), unnormalised.
// ------------------------------------------------------------------------------------------------
constexpr double cx_pi = 3.14159265358979323846264338327950288;

constexpr double cx_sin_taylor(double x)
{ // |x| <= pi
  double term = x, sum = x;
  for (int n = 1; n < 20; n++)
  {
    term *= -x * x / (double) ((2 * n) * (2 * n + 1));
    sum += term;
  }
  return sum;
}

constexpr double cx_cos_taylor(double x)
{
  double term = 1.0, sum = 1.0;
  for (int n = 1; n < 20; n++)
  {
    term *= -x * x / (double) ((2 * n - 1) * (2 * n));
    sum += term;
  }
  return sum;
}

// exp(+2 pi i k / R), k = 0..R-1, as compile-time float tables
template <int R>
struct TwiddleTable
{
  float c[R], s[R];
  constexpr TwiddleTable() : c(), s()
  {
    for (int k = 0; k < R; k++)
    {
      // exact values on the axes and octant symmetry keep the table as accurate as the literals above
      const int k8 = 8 * k;
      if (k8 % R == 0 && (k8 / R) % 2 == 0)
      { // multiple of 90 degrees
        const int q = (k8 / R) / 2;
        c[k] = q == 0 ? 1.f : q == 2 ? -1.f : 0.f;
        s[k] = q == 1 ? 1.f : q == 3 ? -1.f : 0.f;
      }
      else
      {
        double x = 2.0 * cx_pi * (double) k / (double) R;
        if (x > cx_pi)
          x -= 2.0 * cx_pi;
        c[k] = (float) cx_cos_taylor(x);
        s[k] = (float) cx_sin_taylor(x);
      }
    }
  }
};
template <int R>
__device__ constexpr TwiddleTable<R> TW = TwiddleTable<R>();

// radix of the stage whose input span is M (stages 5.., 3.., 2..)
template <int R>
__host__ __device__ constexpr int stage_radix(int M)
{
  return ((R / M) % 5 == 0) ? 5 : ((R / M) % 3 == 0) ? 3 : 2;
}

template <int R>
__host__ __device__ constexpr int digitrevR(int n)
{
  // radices first..last
  int rad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int k = 0;
  for (int M = 1; M < R;)
  {
    rad[k] = stage_radix<R>(M);
    M *= rad[k];
    k++;
  }
  int pos = 0, m = R, rem = n;
  for (int st = k - 1; st >= 0; st--)
  {
    m /= rad[st];
    pos += (rem % rad[st]) * m;
    rem /= rad[st];
  }
  return pos;
}

// digit-reversal as a compile-time table (indexing a constexpr table with an unrolled loop counter folds to a
// constant register index; calling the function above with it does not)
template <int R>
struct DigitRevTable
{
  int pos[R];
  constexpr DigitRevTable() : pos()
  {
    for (int k = 0; k < R; k++)
      pos[k] = digitrevR<R>(k);
  }
};
template <int R>
__device__ constexpr DigitRevTable<R> DIGITREV = DigitRevTable<R>();

template <int R, int P, int M>
__device__ __forceinline__ void mixed_stage(float (&xr)[R], float (&xi)[R])
{
  constexpr int STRIDE = R / (M * P); // w_{M*P}^t = w_R^(t*STRIDE)
#pragma unroll
  for (int b = 0; b < R; b += M * P)
  {
#pragma unroll
    for (int j = 0; j < M; j++)
    {
      float ar[P], ai[P];
#pragma unroll
      for (int q = 0; q < P; q++)
      {
        const int idx = b + j + q * M;
        const int t = (j * q * STRIDE) % R;
        const float vr = xr[idx], vi = xi[idx];
        if (t == 0)
        {
          ar[q] = vr;
          ai[q] = vi;
        }
        else
        {
          const float c = TW<R>.c[t], sn = TW<R>.s[t];
          ar[q] = fmaf(-sn, vi, c * vr);
          ai[q] = fmaf(sn, vr, c * vi);
        }
      }
      if (P == 2)
      {
        xr[b + j] = ar[0] + ar[1];
        xi[b + j] = ai[0] + ai[1];
        xr[b + j + M] = ar[0] - ar[1];
        xi[b + j + M] = ai[0] - ai[1];
      }
      else if (P == 3)
      {
        constexpr float S3 = 0.86602540378443864676f; // sin(2 pi / 3)
        const float t1r = ar[1] + ar[2], t1i = ai[1] + ai[2];
        const float t2r = fmaf(-0.5f, t1r, ar[0]), t2i = fmaf(-0.5f, t1i, ai[0]);
        const float dr = ar[1] - ar[2], di = ai[1] - ai[2];
        xr[b + j] = ar[0] + t1r;
        xi[b + j] = ai[0] + t1i;
        // y1 = t2 + i*S3*d,  y2 = t2 - i*S3*d,  i*d = (-di, dr)
        xr[b + j + M] = fmaf(-S3, di, t2r);
        xi[b + j + M] = fmaf(S3, dr, t2i);
        xr[b + j + 2 * M] = fmaf(S3, di, t2r);
        xi[b + j + 2 * M] = fmaf(-S3, dr, t2i);
      }
      else
      { // P == 5
        constexpr float C1 = 0.30901699437494742410f, C2 = -0.80901699437494742410f;
        constexpr float S1 = 0.95105651629515357212f, S2 = 0.58778525229247312917f;
        const float t1r = ar[1] + ar[4], t1i = ai[1] + ai[4];
        const float t2r = ar[2] + ar[3], t2i = ai[2] + ai[3];
        const float t3r = ar[1] - ar[4], t3i = ai[1] - ai[4];
        const float t4r = ar[2] - ar[3], t4i = ai[2] - ai[3];
        xr[b + j] = ar[0] + t1r + t2r;
        xi[b + j] = ai[0] + t1i + t2i;
        const float m1r = fmaf(C2, t2r, fmaf(C1, t1r, ar[0])), m1i = fmaf(C2, t2i, fmaf(C1, t1i, ai[0]));
        const float m2r = fmaf(C1, t2r, fmaf(C2, t1r, ar[0])), m2i = fmaf(C1, t2i, fmaf(C2, t1i, ai[0]));
        const float n1r = fmaf(S2, t4r, S1 * t3r), n1i = fmaf(S2, t4i, S1 * t3i);
        const float n2r = fmaf(-S1, t4r, S2 * t3r), n2i = fmaf(-S1, t4i, S2 * t3i);
        // y1 = m1 + i n1, y4 = m1 - i n1, y2 = m2 + i n2, y3 = m2 - i n2,   i n = (-n.i, n.r)
        xr[b + j + M] = m1r - n1i;
        xi[b + j + M] = m1i + n1r;
        xr[b + j + 4 * M] = m1r + n1i;
        xi[b + j + 4 * M] = m1i - n1r;
        xr[b + j + 2 * M] = m2r - n2i;
        xi[b + j + 2 * M] = m2i + n2r;
        xr[b + j + 3 * M] = m2r + n2i;
        xi[b + j + 3 * M] = m2i - n2r;
      }
      // keep the butterflies in program order: interleaving them (the scheduler's preference) multiplies the
      // live temporaries of the odd radices and pushes the comparison kernel into scratch
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int R, int M>
__device__ __forceinline__ void mixed_stages(float (&xr)[R], float (&xi)[R])
{
  if constexpr (M < R)
  {
    constexpr int P = stage_radix<R>(M);
    mixed_stage<R, P, M>(xr, xi);
    mixed_stages<R, M * P>(xr, xi);
  }
}

template <int R>
__device__ __forceinline__ void fft_inverse_mixed(float (&xr)[R], float (&xi)[R])
{
  mixed_stages<R, 1>(xr, xi);
}

constexpr bool is_pow2(int r) { return (r & (r - 1)) == 0; }

} // namespace

#endif
