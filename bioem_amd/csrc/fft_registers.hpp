// fft_registers.hpp -- register-resident inverse FFTs of length 2..32 and 128-bit load helpers
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_FFT_REGISTERS_HPP
#define BIOEM_FFT_REGISTERS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// 32-point inverse FFT in registers: radix-2 decimation in frequency, sign +, output bit-reversed.
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

__host__ __device__ constexpr int bitrev5(int n)
{
  return ((n & 1) << 4) | ((n & 2) << 2) | (n & 4) | ((n & 8) >> 2) | ((n & 16) >> 4);
}

__device__ __forceinline__ void fft32_inverse(float (&xr)[32], float (&xi)[32])
{
#pragma unroll
  for (int s = 0; s < 5; s++)
  {
    const int m = 16 >> s;
#pragma unroll
    for (int b = 0; b < 32; b += 2 * m)
    {
#pragma unroll
      for (int j = 0; j < m; j++)
      {
        const int i0 = b + j, i1 = b + j + m;
        const int t = j << s;
        const float ar = xr[i0], ai = xi[i0], br = xr[i1], bi = xi[i1];
        xr[i0] = ar + br;
        xi[i0] = ai + bi;
        const float dr = ar - br, di = ai - bi;
        if (t == 0)
        {
          xr[i1] = dr;
          xi[i1] = di;
        }
        else if (t == 8)
        {
          xr[i1] = -di;
          xi[i1] = dr;
        }
        else
        {
          const float c = COS32[t], sn = SIN32[t];
          xr[i1] = fmaf(dr, c, -(di * sn));
          xi[i1] = fmaf(dr, sn, di * c);
        }
      }
    }
  }
}

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

} // namespace

#endif
