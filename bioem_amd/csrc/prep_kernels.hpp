// prep_kernels.hpp -- comparison layout, projection, exact-DFT r2c, particle sums, CTF convolution
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_PREP_KERNELS_HPP
#define BIOEM_PREP_KERNELS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// layout of a half spectrum used by the comparison kernels
//   fast   : N = N1*R (R = 32, 16, 8, 4 or 2), kx = N1*k2 + k1  ->  float2 index ((k1*R/2 + (k2>>1))*H + ky)*2 + (k2&1)
//            (lane = ky reads 16 B = two k2 of one k1 -> fully coalesced dwordx4, and the R inputs
//             of one register FFT arrive as R/2 such loads).  The `fast` argument carries R/2 (0 = generic).
//   generic: reference layout kx*H + ky
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t layout_index(int fast, int N1, int H, int kx, int ky)
{
  if (!fast)
    return (size_t) kx * H + ky;
  const int k1 = kx % N1, k2 = kx / N1;
  return ((size_t) (k1 * fast + (k2 >> 1)) * H + ky) * 2 + (k2 & 1);
}

__global__ void k_reorder(const float2 *__restrict__ src, float2 *__restrict__ dst, int nImg, int N, int H, int fast,
                          int N1)
{
  const size_t M = (size_t) N * H;
  const size_t total = M * nImg;
  for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t) gridDim.x * blockDim.x)
  {
    const size_t img = e / M;
    const int r = (int) (e - img * M);
    const int kx = r / H, ky = r - kx * H;
    dst[img * M + layout_index(fast, N1, H, kx, ky)] = src[e];
  }
}

__global__ void k_unreorder(const float2 *__restrict__ src, float2 *__restrict__ dst, int nImg, int N, int H,
                            int fast, int N1)
{
  const size_t M = (size_t) N * H;
  const size_t total = M * nImg;
  for (size_t e = (size_t) blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t) gridDim.x * blockDim.x)
  {
    const size_t img = e / M;
    const int r = (int) (e - img * M);
    const int kx = r / H, ky = r - kx * H;
    dst[e] = src[img * M + layout_index(fast, N1, H, kx, ky)];
  }
}

// ------------------------------------------------------------------------------------------------
// projection: bioem.cpp:1604-1818 (rotation, point / sphere splat, tempden)
// one thread per model point, blockIdx.y = orientation inside the batch
// ------------------------------------------------------------------------------------------------
__global__ void k_project(const bioem_hip_model_point *__restrict__ pts, int nPts, const float4 *__restrict__ angles,
                          int o0, int isQuat, int N, float pixelSize, int shiftX, int shiftY,
                          double *__restrict__ proj, double *__restrict__ tempden)
{
  const int ob = blockIdx.y;
  const float4 a = angles[o0 + ob];
  float rotmat[3][3];
  if (isQuat)
  {
    const float q0 = a.x, q1 = a.y, q2 = a.z, q3 = a.w; // bioem.cpp:1632-1646
    rotmat[0][0] = 1 - 2 * q1 * q1 - 2 * q2 * q2;
    rotmat[1][0] = 2 * (q0 * q1 - q2 * q3);
    rotmat[2][0] = 2 * (q0 * q2 + q1 * q3);
    rotmat[0][1] = 2 * (q0 * q1 + q2 * q3);
    rotmat[1][1] = 1 - 2 * q0 * q0 - 2 * q2 * q2;
    rotmat[2][1] = 2 * (q1 * q2 - q0 * q3);
    rotmat[0][2] = 2 * (q0 * q2 - q1 * q3);
    rotmat[1][2] = 2 * (q1 * q2 + q0 * q3);
    rotmat[2][2] = 1 - 2 * q0 * q0 - 2 * q1 * q1;
  }
  else
  {
    const float alpha = a.x, beta = a.y, gam = a.z; // bioem.cpp:1653-1672
    const float ca = cosf(alpha), sa = sinf(alpha), cb = cosf(beta), sb = sinf(beta), cg = cosf(gam), sg = sinf(gam);
    rotmat[0][0] = cg * ca - cb * sa * sg;
    rotmat[0][1] = cg * sa + cb * ca * sg;
    rotmat[0][2] = sg * sb;
    rotmat[1][0] = -sg * ca - cb * sa * cg;
    rotmat[1][1] = -sg * sa + cb * ca * cg;
    rotmat[1][2] = cg * sb;
    rotmat[2][0] = sb * sa;
    rotmat[2][1] = -sb * ca;
    rotmat[2][2] = cb;
  }
  double td = 0.;
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  double *map = proj + (size_t) ob * N * N;
  if (n < nPts)
  {
    const bioem_hip_model_point p = pts[n];
    float rp[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < 3; j++)
        rp[k] += rotmat[k][j] * p.pos[j];
    const float radius = p.radius, density = p.density;
    if (radius <= pixelSize)
    {
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f);
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f);
      if (!(i < 0 || j < 0 || i >= N || j >= N))
      {
        atomicAdd(&map[i * N + j], (double) density);
        td += (double) density;
      }
    }
    else
    {
      const int i = (int) floorf(rp[0] / pixelSize + (float) N / 2.0f + 0.5f) - shiftX;
      const int j = (int) floorf(rp[1] / pixelSize + (float) N / 2.0f + 0.5f) - shiftY;
      const int irad = (int) (radius / pixelSize) + 1;
      const float rad2 = radius * radius;
      if (!(i < irad || j < irad || i >= N - irad || j >= N - irad))
      {
        for (int ii = i - irad; ii < i + irad + 1; ii++)
          for (int jj = j - irad; jj < j + irad + 1; jj++)
          {
            const float dist = ((float) (ii - i) * (ii - i) + (jj - j) * (jj - j)) * pixelSize * pixelSize;
            if (dist < rad2)
            {
              const double w = (double) (pixelSize * pixelSize * 2 * sqrtf(rad2 - dist) * density * 3) /
                               (4 * M_PI * radius * rad2);
              atomicAdd(&map[ii * N + jj], w);
              td += w;
            }
          }
      }
    }
  }
  // block reduction of tempden
  __shared__ double red[256];
  red[threadIdx.x] = td;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1)
  {
    if ((int) threadIdx.x < s)
      red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && red[0] != 0.)
    atomicAdd(&tempden[ob], red[0]);
}

// ------------------------------------------------------------------------------------------------
// r2c as two exact-DFT passes with double accumulation (FFTW forward convention, unnormalised).
// rows: src is either the double projection map scaled by NormDen/tempden in float (bioem.cpp:1808-1818)
//       or float particle maps.
// ------------------------------------------------------------------------------------------------
// Both passes use one Cooley-Tukey split N = A*B (A the largest divisor <= sqrt(N); A = 1 for prime N):
//   Y[j1][kb] = sum_{j2<B} x[A*j2 + j1] * w_B^(j2*kb),   X[k] = sum_{j1<A} w_N^(j1*k) * Y[j1][k mod B]
// i.e. N*(A+B) instead of N*N terms per 1-D transform, still exact-DFT arithmetic in double.
__global__ void k_dft_rows(const double *__restrict__ srcD, const float *__restrict__ srcF,
                           const double *__restrict__ tempden, float NormDen, int N, int H, int A, int B,
                           const double2 *__restrict__ twD, double2 *__restrict__ rowspec)
{
  extern __shared__ double srow[];          // N doubles, then N double2
  double2 *Y = reinterpret_cast<double2 *>(srow + N + (N & 1));
  const int i = blockIdx.x, b = blockIdx.y;
  float ratio = 1.f;
  if (srcD)
    ratio = NormDen / (float) tempden[b];
  for (int j = threadIdx.x; j < N; j += blockDim.x)
  {
    float v;
    if (srcD)
    {
      v = (float) srcD[((size_t) b * N + i) * N + j];
      v = v * ratio;
    }
    else
      v = srcF[((size_t) b * N + i) * N + j];
    srow[j] = (double) v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < N; e += blockDim.x)
  {
    const int j1 = e / B, kb = e - j1 * B;
    double ar = 0., ai = 0.;
    int idx = 0;
    const int step = (A * kb) % N;
    for (int j2 = 0; j2 < B; j2++)
    {
      const double2 w = twD[idx];
      const double x = srow[A * j2 + j1];
      ar = fma(x, w.x, ar);
      ai = fma(-x, w.y, ai); // forward: e^{-i}
      idx += step;
      if (idx >= N)
        idx -= N;
    }
    Y[e] = make_double2(ar, ai);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < H; k += blockDim.x)
  {
    const int kb = k % B;
    double ar = 0., ai = 0.;
    int idx = 0;
    for (int j1 = 0; j1 < A; j1++)
    {
      const double2 w = twD[idx]; // multiply by conj(w)
      const double2 y = Y[j1 * B + kb];
      ar = fma(y.x, w.x, ar);
      ar = fma(y.y, w.y, ar);
      ai = fma(y.y, w.x, ai);
      ai = fma(-y.x, w.y, ai);
      idx += k;
      if (idx >= N)
        idx -= N;
    }
    rowspec[((size_t) b * N + i) * H + k] = make_double2(ar, ai);
  }
}

__global__ void k_dft_cols(const double2 *__restrict__ rowspec, int N, int H, int A, int B,
                           const double2 *__restrict__ twD, float2 *__restrict__ out)
{
  extern __shared__ double srow[];
  double2 *col = reinterpret_cast<double2 *>(srow);
  double2 *Y = col + N;
  const int k = blockIdx.x, b = blockIdx.y;
  for (int i = threadIdx.x; i < N; i += blockDim.x)
    col[i] = rowspec[((size_t) b * N + i) * H + k];
  __syncthreads();
  for (int e = threadIdx.x; e < N; e += blockDim.x)
  {
    const int j1 = e / B, kb = e - j1 * B;
    double ar = 0., ai = 0.;
    int idx = 0;
    const int step = (A * kb) % N;
    for (int j2 = 0; j2 < B; j2++)
    {
      const double2 w = twD[idx];
      const double2 x = col[A * j2 + j1];
      ar = fma(x.x, w.x, ar);
      ar = fma(x.y, w.y, ar);
      ai = fma(x.y, w.x, ai);
      ai = fma(-x.x, w.y, ai);
      idx += step;
      if (idx >= N)
        idx -= N;
    }
    Y[e] = make_double2(ar, ai);
  }
  __syncthreads();
  for (int u = threadIdx.x; u < N; u += blockDim.x)
  {
    const int kb = u % B;
    double ar = 0., ai = 0.;
    int idx = 0;
    for (int j1 = 0; j1 < A; j1++)
    {
      const double2 w = twD[idx];
      const double2 y = Y[j1 * B + kb];
      ar = fma(y.x, w.x, ar);
      ar = fma(y.y, w.y, ar);
      ai = fma(y.y, w.x, ai);
      ai = fma(-y.x, w.y, ai);
      idx += u;
      if (idx >= N)
        idx -= N;
    }
    out[(size_t) b * N * H + (size_t) u * H + k] = make_float2((float) ar, (float) ai);
  }
}

// particle sums, bioem.cpp:2087-2107: sequential float accumulation in row-major order.
__global__ void k_map_sums(const float *__restrict__ maps, int NN, float *__restrict__ sum, float *__restrict__ sumsq)
{
  __shared__ float buf[4096];
  const float *m = maps + (size_t) blockIdx.x * NN;
  float s = 0.0f, s2 = 0.0f;
  for (int base = 0; base < NN; base += 4096)
  {
    const int cnt = min(4096, NN - base);
    for (int t = threadIdx.x; t < cnt; t += blockDim.x)
      buf[t] = m[base + t];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int t = 0; t < cnt; t++)
      {
        s += buf[t];
        s2 += buf[t] * buf[t];
      }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    sum[blockIdx.x] = s;
    sumsq[blockIdx.x] = s2;
  }
}

// ------------------------------------------------------------------------------------------------
// convolution: bioem.cpp:1855-1923.  grid (CTFs of the launch, nOrientInBatch); CTF index = c0 + blockIdx.x.
// sumsquareC is accumulated sequentially in float in the reference's order (rows; inside a row the
// interior columns doubled, then column 0, then column N/2 for even N): the terms are produced in
// parallel into `scratch` in that order and summed by one lane.
// ------------------------------------------------------------------------------------------------
__global__ void k_convolve(const float2 *__restrict__ proj, const float2 *__restrict__ ctf,
                           const float *__restrict__ ctfParam, int N, int H, int fast, int N1, int c0,
                           float2 *__restrict__ conv, float *__restrict__ scratch,
                           bioem_hip_param5 *__restrict__ params)
{
  __shared__ __align__(16) float buf[2][4096];
  const int c = c0 + blockIdx.x, ob = blockIdx.y;
  const int oc = ob * gridDim.x + blockIdx.x;
  const int M = N * H;
  const float2 *P = proj + (size_t) ob * M;
  const float2 *K = ctf + (size_t) c * M;
  float2 *O = conv + (size_t) oc * M;
  float *S = scratch + (size_t) oc * M;
  const int even = ((N & 1) == 0);
  const int jend = even ? H - 1 : H;
  float sumC = 0.f;
  for (int e = threadIdx.x; e < M; e += blockDim.x)
  {
    const int i = e / H, j = e - i * H;
    const float2 p = P[e], k = K[e];
    float2 o;
    o.x = (p.x * k.x + p.y * k.y);
    o.y = (p.y * k.x - p.x * k.y);
    O[layout_index(fast, N1, H, i, j)] = o;
    const float t = o.x * o.x + o.y * o.y;
    // position of this term in the reference's summation order
    int pos;
    if (j >= 1 && j < jend)
      pos = i * H + (j - 1);
    else if (j == 0)
      pos = i * H + (jend - 1);
    else
      pos = i * H + jend; // j == H-1, even N
    S[pos] = (j >= 1 && j < jend) ? t * 2 : t;
    if (e == 0)
      sumC = o.x;
  }
  __syncthreads();
  __threadfence_block();
  // one lane adds the terms in order (the float rounding of the reference's loop); waves 1..3 stage the next 4 096
  // terms into the other half of `buf` meanwhile, and the adding lane reads four terms per LDS access
  float ss = 0.f;
  const int wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < min(4096, M); t += blockDim.x)
    buf[0][t] = S[t];
  __syncthreads();
  for (int base = 0, b = 0; base < M; base += 4096, b ^= 1)
  {
    const int cnt = min(4096, M - base);
    if (wave != 0)
    {
      const int nb = base + 4096;
      if (nb < M)
      {
        const int cntn = min(4096, M - nb);
        for (int t = threadIdx.x - 64; t < cntn; t += blockDim.x - 64)
          buf[b ^ 1][t] = S[nb + t];
      }
    }
    else if (threadIdx.x == 0)
    {
      const float4 *q = reinterpret_cast<const float4 *>(buf[b]);
      const int n4 = cnt >> 2;
#pragma unroll 8
      for (int t = 0; t < n4; t++)
      {
        const float4 v = q[t];
        ss += v.x;
        ss += v.y;
        ss += v.z;
        ss += v.w;
      }
      for (int t = n4 << 2; t < cnt; t++)
        ss += buf[b][t];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    bioem_hip_param5 r;
    r.amp = ctfParam[3 * c + 0];
    r.pha = ctfParam[3 * c + 1];
    r.env = ctfParam[3 * c + 2];
    r.sumC = sumC;
    const float norm2 = (float) (N * N);
    r.sumsquareC = ss / norm2;
    params[oc] = r;
  }
}

// sums only (compat entry supplies conv spectra but we never trust host params blindly: they are used as given)

} // namespace

#endif
