// bioem_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the BioEM compare path behind the
// C ABI of include/bioem_hip.h.  Hand-written HIP, wave64, no vendor FFT/BLAS on the hot path.
//
// Hot path (replaces bioem_cuda::compareRefMaps + cuFFT, /root/reference/bioem_cuda.cu:527-684, and the
// host-side createProjection / createConvolutedProjectionMap, /root/reference/bioem.cpp:1604-1923):
//
//   k_project        model points -> real-space projection (double atomics), one launch per batch
//   k_dft_rows/cols  r2c of the projections (exact DFT, double accumulation; not on the critical path)
//   k_convolve       proj * conj(CTF) -> conv spectra in the comparison layout, sumC, sumsquareC
//   k_compare_fast   one WAVE per (particle, orientation*CTF) comparison:
//                      spectrum product -> pruned inverse 2-D transform -> displacement-window log posterior
//                      -> wave log-sum-exp/arg-max partial.  The length-N inverse along kx is split as
//                      N = N1*32: N1 register-resident 32-point FFTs per frequency column (lane = column),
//                      recombined only for the 2*maxD+1 displacements that are consumed (output pruning),
//                      so no radix-7 butterfly is ever needed for N = 224.  The transform along ky is a
//                      pruned real DFT evaluated from LDS for the displacement window only.
//   k_compare_generic  same maths for any N / maxD (direct pruned DFT), correctness path
//   k_fold           folds the per-(particle, orientation*CTF) partials into the probability block in the
//                      reference's (orientation, CTF) order (bioem_algorithm.h:96-123, bioem.cpp:1527-1600)
//
// Numerics: float expressions that the reference evaluates in float are written in the same order and the
// file is compiled with -ffp-contract=off (FMAs only where fmaf() is spelled out).  Sums that the reference
// accumulates sequentially in float (sumsquareC, particle sums) are accumulated in the same order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bioem_hip.h"

#define MIN_PROB (-999999.)

namespace
{

struct Partial
{
  double sumExp;
  float best;
  int id;
  float value;
  int pad;
};

typedef bioem_hip_param_device PD;

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
struct HipErr
{
  std::string msg;
};

#define HIP_CHECK(h, expr)                                                                                         \
  do                                                                                                               \
  {                                                                                                                \
    hipError_t e_ = (expr);                                                                                        \
    if (e_ != hipSuccess)                                                                                          \
    {                                                                                                              \
      char buf_[512];                                                                                              \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);     \
      (h)->err = buf_;                                                                                             \
      return 1;                                                                                                    \
    }                                                                                                              \
  } while (0)

} // namespace

struct bioem_hip_ctx
{
  int device = 0;
  hipStream_t stream = nullptr;
  PD pd;
  int nMaps = 0, nAngles = 0, nCTF = 0, algo = 1;
  int N = 0, H = 0, M = 0;
  int fast = 0, N1 = 0, winD = 0; // winD = template window half width used by the fast kernel
  int pchunk = 128;               // particle chunk of the fast kernel's block order (0 = all particles); measured:
                                  // 1 000 particles 6.73 -> 6.56 ms, 10 000 particles (2 GB, beyond the Infinity
                                  // Cache) 78.5 -> 63.2 ms per launch
  int gs = 1;                     // pixels per window row of the fast kernel (gcd of the displacement offsets)
  bool nyq = false;               // Nyquist column handled outside the 64-column blocks (N/2 a multiple of 64)
  int nd = 0;                     // displacements per axis
  std::vector<int> disp;
  int OB = 0;     // orientations per batch of the native path
  int maxOC = 0;  // capacity of conv/param/partial buffers in (orientation*CTF) units
  int chunkB = 0; // images per DFT chunk

  float2 *dRef = nullptr;
  float *dSumRef = nullptr, *dSumsqRef = nullptr;
  float2 *dCTF = nullptr;
  float *dCtfParam = nullptr;
  bioem_hip_model_point *dPts = nullptr;
  int nPts = 0;
  float NormDen = 0, pixelSize = 0;
  int shiftX = 0, shiftY = 0;
  float4 *dAngles = nullptr;
  int nAnglesUp = 0, isQuat = 1;
  float2 *dTw = nullptr;   // N+1 entries exp(+2 pi i k/N), float
  double2 *dTwD = nullptr; // N entries, double
  int *dDisp = nullptr;
  double2 *dLtab = nullptr;
  float2 *dTwk = nullptr;
  float *dTnyq = nullptr; // [nMaps][maxOC][2*winD+1] Nyquist-column rows of the current launch (nyq only)

  double *dProjReal = nullptr; // [chunkB][N*N]
  double *dTempDen = nullptr;  // [chunkB]
  double2 *dRowSpec = nullptr; // [chunkB][N][H]
  float2 *dSpecRef = nullptr;  // [chunkB][M] reference layout
  float *dScratch = nullptr;   // [maxOC][M] ordered |X|^2 terms for sumsquareC
  float2 *dConv = nullptr;     // [maxOC][M] comparison layout
  bioem_hip_param5 *dParams = nullptr;
  Partial *dPartials = nullptr; // [nMaps][maxOC]
  unsigned char *dProb = nullptr;
  size_t probBytes = 0;

  // second buffer set + stream: projection/convolution of batch k+1 overlap the comparison of batch k
  hipStream_t prepStream = nullptr;
  double *dProjReal2 = nullptr;
  double *dTempDen2 = nullptr;
  double2 *dRowSpec2 = nullptr;
  float2 *dSpecRef2 = nullptr;
  float *dScratch2 = nullptr;
  float2 *dConv2 = nullptr;
  bioem_hip_param5 *dParams2 = nullptr;
  hipEvent_t prepDone[2] = {nullptr, nullptr};
  hipEvent_t cmpDone[2] = {nullptr, nullptr};
  bool cmpPending[2] = {false, false};

  // compat entry staging
  float2 *hStage = nullptr;
  float2 *dStage = nullptr;
  bioem_hip_param5 *hStageP = nullptr;
  int stageConv = 0;
  hipEvent_t slotEvent[2] = {nullptr, nullptr};
  bool slotPending[2] = {false, false};

  // timing
  std::vector<hipEvent_t> evPool;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> evPending;
  double compareMs = 0;
  long long launches = 0, comparisons = 0;

  std::string err;
};

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
// convolution: bioem.cpp:1855-1923.  grid (nCTF, nOrientInBatch).
// sumsquareC is accumulated sequentially in float in the reference's order (rows; inside a row the
// interior columns doubled, then column 0, then column N/2 for even N): the terms are produced in
// parallel into `scratch` in that order and summed by one lane.
// ------------------------------------------------------------------------------------------------
__global__ void k_convolve(const float2 *__restrict__ proj, const float2 *__restrict__ ctf,
                           const float *__restrict__ ctfParam, int N, int H, int fast, int N1, int nCTF,
                           float2 *__restrict__ conv, float *__restrict__ scratch,
                           bioem_hip_param5 *__restrict__ params)
{
  __shared__ float buf[4096];
  const int c = blockIdx.x, ob = blockIdx.y;
  const int oc = ob * nCTF + c;
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
  float ss = 0.f;
  for (int base = 0; base < M; base += 4096)
  {
    const int cnt = min(4096, M - base);
    for (int t = threadIdx.x; t < cnt; t += blockDim.x)
      buf[t] = S[base + t];
    __syncthreads();
    if (threadIdx.x == 0)
      for (int t = 0; t < cnt; t++)
        ss += buf[t];
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

// ------------------------------------------------------------------------------------------------
// log posterior, bioem_algorithm.h:18-70.  constPart = second log term, priorPart = Gaussian priors:
// both depend on the (orientation, CTF) pair only and are hoisted; the summation order
// (t1 + t2) - prior of the reference is kept.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void logpro_consts(const PD &pd, const bioem_hip_param5 &q, double &t2, double &prior)
{
  const float Np = pd.Ntotpi;
  const double ForLogProb = (double) (q.sumsquareC * Np - q.sumC * q.sumC);
  t2 = ((double) Np * 0.5 - 2) * log((double) (Np - 2) * ForLogProb);
  const float amp = q.amp, pha = q.pha, env = q.env;
  if (!pd.tousepsf)
  {
    prior = (double) (env * env) / 2. / (double) pd.sigmaPriorbctf / (double) pd.sigmaPriorbctf -
            (double) ((pha - pd.Priordefcent) * (pha - pd.Priordefcent)) / 2. / (double) pd.sigmaPriordefo /
                (double) pd.sigmaPriordefo -
            (double) ((amp - pd.Priorampcent) * (amp - pd.Priorampcent)) / 2. / (double) pd.sigmaPrioramp /
                (double) pd.sigmaPrioramp;
  }
  else
  {
    const double envF = 4. * M_PI * M_PI * (double) env / (double) (env * env + pha * pha);
    const double phaF = 4. * M_PI * M_PI * (double) pha / (double) (env * env + pha * pha);
    const double dp = phaF - (double) pd.Priordefcent;
    prior = envF * envF / 2. / (double) pd.sigmaPriorbctf / (double) pd.sigmaPriorbctf -
            dp * dp / 2. / (double) pd.sigmaPriordefo / (double) pd.sigmaPriordefo -
            (double) ((amp - pd.Priorampcent) * (amp - pd.Priorampcent)) / 2. / (double) pd.sigmaPrioramp /
                (double) pd.sigmaPrioramp;
  }
}

__device__ __forceinline__ double logpro_eval(const PD &pd, const bioem_hip_param5 &q, float cc, float sumref,
                                              float sumsqref, double t2, double prior)
{
  const float Np = pd.Ntotpi;
  const float sum = q.sumC, sumsq = q.sumsquareC;
  const float firstele_f = Np * (sumsqref * sumsq - cc * cc) + 2 * sumref * sum * cc - sumsqref * sum * sum -
                           sumref * sumref * sumsq;
  double logpro = (double) (3 - Np) * 0.5 * log((double) firstele_f) + t2;
  logpro -= prior;
  return logpro;
}

// online log-sum-exp state of one lane / wave
struct Lse
{
  float m;   // best logpro (narrowed to float as the reference does)
  double s;  // sum exp(logpro - m)
  int id;    // rank of the best displacement in the reference's visiting order
  float val; // cross-correlation value at the best displacement
};

__device__ __forceinline__ void lse_init(Lse &L)
{
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
}

// algo 1: logpro narrowed to float before use (bioem_algorithm.h:84); algo 2: double in the exponent,
// float for the running best (bioem.cpp:1470,1500-1507)
__device__ __forceinline__ void lse_push(Lse &L, double lp, int id, float val, int algo)
{
  const float lpf = (float) lp;
  const double lpe = (algo == 1) ? (double) lpf : lp;
  if (L.m < lpf)
  {
    L.s = (L.m == -INFINITY) ? 0. : L.s * exp((double) L.m - (double) lpf);
    L.m = lpf;
    L.id = id;
    L.val = val;
  }
  L.s += exp(lpe - (double) L.m);
}

__device__ __forceinline__ void lse_merge(Lse &L, float m2, double s2, int id2, float val2)
{
  if (m2 > L.m || (m2 == L.m && id2 < L.id))
  {
    const double sc = (L.m == -INFINITY) ? 0. : L.s * exp((double) L.m - (double) m2);
    L.s = sc + s2;
    L.m = m2;
    L.id = id2;
    L.val = val2;
  }
  else
  {
    const double sc = (m2 == -INFINITY) ? 0. : s2 * exp((double) m2 - (double) L.m);
    L.s += sc;
  }
}

__device__ __forceinline__ void lse_wave_reduce(Lse &L)
{
  for (int off = 32; off > 0; off >>= 1)
  {
    const float m2 = __shfl_xor(L.m, off);
    const double s2 = __shfl_xor(L.s, off);
    const int id2 = __shfl_xor(L.id, off);
    const float v2 = __shfl_xor(L.val, off);
    lse_merge(L, m2, s2, id2, v2);
  }
}

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

struct CompareArgs
{
  const float2 *ref;  // [nMaps][M] comparison layout
  const float2 *conv; // [nOC][M]
  const bioem_hip_param5 *params;
  const float *sumRef, *sumsqRef;
  const float2 *tw; // N+1
  const int *disp;  // nd
  const double2 *ltab; // 64 x {c, -log c}
  const float2 *twk;   // [N1][2*WD+1] recombination twiddles exp(2 pi i d k1 / N), d = -WD..WD
  float *tnyq;         // [nMaps][ldPart][2*WD+1] Nyquist-column rows (fast path with the Nyquist split only)
  Partial *partials; // [nMaps][ldPart]
  int ldPart;
  int N, H, N1, nd, maxD, nOC, nMaps, algo;
  int pchunk; // particles per block-order chunk of the fast kernel
  int gs;     // pixels per window row of the fast kernel (template GS)
  PD pd;
};

// ------------------------------------------------------------------------------------------------
// log of a positive float in double precision, cheap: f = m * 2^e, m in [1,2); c ~ 1/m from a 64-entry
// table, r = m*c - 1 exactly rounded by one fma (|r| <= 2^-7), log f = e ln2 - log c + log1p(r) with a
// degree-6 Taylor polynomial (truncation 2^-49/7).  Absolute error ~1e-15, i.e. < 3e-11 after the
// (3-Np)/2 amplification -- far below the float narrowing the reference applies to logpro.
// Table entry = {c, -log(c)} in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __noinline__ double log_slow_path(float f) { return log((double) f); }

__device__ __forceinline__ double log_of_float(float f, const double2 *ltab)
{
  const unsigned int bits = __float_as_uint(f);
  if (!(f > 1.1754944e-38f) || bits >= 0x7f800000u) // zero, negative, subnormal, inf, nan: exact slow path
    return log_slow_path(f);
  const int e = (int) (bits >> 23) - 127;
  const float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
  const double2 t = ltab[(bits >> 17) & 63];
  const double r = fma((double) m, t.x, -1.0);
  double p = fma(r, -1.0 / 6.0, 1.0 / 5.0);
  p = fma(r, p, -1.0 / 4.0);
  p = fma(r, p, 1.0 / 3.0);
  p = fma(r, p, -1.0 / 2.0);
  p = fma(r * r, p, r);
  return fma((double) e, 0.693147180559945309417232, t.y + p);
}

// exp of a non-positive double difference through the hardware exp2 (relative error ~2e-7 per term; the
// terms are summed in double, so log(Total) moves by < 1e-6)
__device__ __forceinline__ double exp_fast_nonpos(double x) { return (double) __expf((float) x); }

struct LseF
{
  float m;
  double s;
  int id;
  float val;
};

__device__ __forceinline__ void lsef_push(LseF &L, double lp, int id, float val, int algo)
{
  const float lpf = (float) lp;
  const double lpe = (algo == 1) ? (double) lpf : lp;
  if (L.m < lpf)
  {
    L.s = (L.m == -INFINITY) ? 0. : L.s * exp_fast_nonpos((double) L.m - (double) lpf);
    L.m = lpf;
    L.id = id;
    L.val = val;
  }
  else if (L.m == lpf && id < L.id)
  { // equal maxima: the first VISITED displacement wins (ids are visiting ranks; a lane may push out of order)
    L.id = id;
    L.val = val;
  }
  L.s += exp_fast_nonpos(lpe - (double) L.m);
}

__device__ __forceinline__ void lsef_wave_reduce(LseF &L)
{
  for (int off = 32; off > 0; off >>= 1)
  {
    const float m2 = __shfl_xor(L.m, off);
    const double s2 = __shfl_xor(L.s, off);
    const int id2 = __shfl_xor(L.id, off);
    const float v2 = __shfl_xor(L.val, off);
    if (m2 > L.m || (m2 == L.m && id2 < L.id))
    {
      const double sc = (L.m == -INFINITY) ? 0. : L.s * exp_fast_nonpos((double) L.m - (double) m2);
      L.s = sc + s2;
      L.m = m2;
      L.id = id2;
      L.val = v2;
    }
    else
    {
      const double sc = (m2 == -INFINITY) ? 0. : s2 * exp_fast_nonpos((double) m2 - (double) L.m);
      L.s += sc;
    }
  }
}

// Window accumulation over one block of 64 frequency columns held in LDS as Tl[row = dx + WD][64] float2
// (already weighted by 1 or 2 per column; zero beyond H).  lane = (iy, group); a group owns `nr` consecutive
// displacement rows so that each LDS twiddle read E[ky*dy] feeds nr accumulators; T is read two columns
// at a time (ds_read_b128).  STATIC: nr == NR known at compile time (the +-10 px, grid 1 case).
// NP = number of column pairs: 32 for a block, 1 for the Nyquist column parked in the pad columns 64/65.
template <int NR, bool STATIC, int NP, int TS>
__device__ __forceinline__ void window_accumulate(const float2 *Tl, const float2 *twl, int N, int step, int idx0,
                                                  const int (&rowoff)[NR], int nr, float (&acc)[NR])
{
  int idx = idx0;
#pragma unroll 2
  for (int kp = 0; kp < NP; kp++)
  {
    const float2 w0 = twl[idx];
    idx += step;
    if (idx >= N)
      idx -= N;
    const float2 w1 = twl[idx];
    idx += step;
    if (idx >= N)
      idx -= N;
#pragma unroll
    for (int r = 0; r < NR; r++)
    {
      if (STATIC || r < nr)
      {
        // STATIC: the nr rows of a lane are consecutive (unit grid), so one base + compile-time offsets
        const float4 t = *reinterpret_cast<const float4 *>(&Tl[(STATIC ? rowoff[0] + r * TS : rowoff[r]) + 2 * kp]);
        float v = acc[r];
        v = fmaf(t.x, w0.x, v);
        v = fmaf(-t.y, w0.y, v);
        v = fmaf(t.z, w1.x, v);
        v = fmaf(-t.w, w1.y, v);
        acc[r] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fast comparison kernel: N = R*N1 with R = 32, 16, 8, 4 or 2 (the largest of those dividing N, i.e. any even N),
// 2*maxD+1 <= 2*WD+1 <= 31.
// block = 4 waves = 4 consecutive (orientation*CTF) indices of ONE particle (the particle columns are then
// served to waves 1..3 from L1); blockIdx.x = ocGroup * nMaps + particle, so concurrently resident blocks
// share the same 4 conv spectra in L2.  Columns are processed in blocks of 64 (lane = column): register
// FFTs -> T block in LDS -> window accumulation, so the LDS footprint per wave is (2*WD+1)*64*8 bytes
// (10.5 KiB for +-10 px => 3 blocks per CU, matching the VGPR-limited 3 waves per SIMD).
// ------------------------------------------------------------------------------------------------
// FFT flavour: decimation in time with 6-op butterflies (default) or the decimation-in-frequency original
// (R = 32 only)
#ifndef BIOEM_FFT_DIF
#define BIOEM_FFT_DIF 0
#endif
#if BIOEM_FFT_DIF
#define FFT_IN(k) (k)
#define FFT_OUT(n) bitrev5(n)
#define FFT_RUN(xr, xi) fft32_inverse(xr, xi)
#else
#define FFT_IN(k) bitrevR<R>(k)
#define FFT_OUT(n) (n)
#define FFT_RUN(xr, xi) fft_inverse_dit<R>(xr, xi)
#endif
#ifndef BIOEM_BLOCK_BARRIER
#define BIOEM_BLOCK_BARRIER 1
#endif
#if BIOEM_BLOCK_BARRIER
#define WAVE_OR_BLOCK_SYNC() __syncthreads()
#else
#define WAVE_OR_BLOCK_SYNC()                                                                                       \
  do                                                                                                               \
  {                                                                                                                \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                         \
    __builtin_amdgcn_wave_barrier();                                                                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                         \
  } while (0)
#endif
#ifndef BIOEM_FAST_WAVES_PER_SIMD
#define BIOEM_FAST_WAVES_PER_SIMD 3
#endif
// NYQ (N/2 a multiple of 64, e.g. 128 and 256): the half spectrum has N/2 + 1 columns, one more than fills the
// 64-lane column blocks, and a whole extra block pass for that single Nyquist column would cost 1/2 (128) or 1/3
// (256) of the kernel.  Instead k_nyquist_rows (below) forms the 2*WD+1 column-transform outputs of that column
// for every comparison of the launch by direct summation, and this kernel adds (-1)^dy * Re T[dx][N/2] to its
// window sums (FFTW c2r convention: weight 1, real part only).  The tail is deliberately tiny: anything larger
// (an inlined or called summation) pushes the register allocation of the main loop into scratch.
// GS (1..4): row stride of the window in pixels.  T row m (-WD..WD) holds displacement dx = m*GS, so a coarse
// DISPLACE_CENTER grid whose offsets are all multiples of GS reaches +-15*GS pixels with the same 2*WD+1 rows.
template <int WD, int R, bool NYQ, int GS>
__global__ __launch_bounds__(256, (WD <= 10 ? BIOEM_FAST_WAVES_PER_SIMD : 2)) void k_compare_fast(const CompareArgs a)
{
  constexpr int NW = 2 * WD + 1;
  constexpr int R2 = R / 2;            // rows (k2 pairs) per k1 step
  constexpr int RD = R2 < 4 ? R2 : 4;  // depth of the operand ring
  constexpr int NR = (WD <= 5) ? 3 : (WD <= 10) ? 7 : 16; // accumulators (window rows) per lane
  constexpr int TS = 66; // T row stride in float2 (64 columns + 2 pad: row groups land on different banks)
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1;
  float2 *twl = reinterpret_cast<float2 *>(smem);                            // N+1 (+pad)
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8); // nd ints (256 B reserved)
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256); // 64 entries
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256 + 1024);
  // wave index made provably uniform (SGPR) so that per-wave base pointers use scalar addressing
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float2 *Tl = Tall + (size_t) wave * NW * TS;

  for (int t = threadIdx.x; t <= N; t += blockDim.x)
    twl[t] = a.tw[t];
  int *dinv = displ + 32; // visiting rank of window row m (displacement m*GS), index m + mD
  const int mD = a.maxD / GS;
  for (int t = threadIdx.x; t < a.nd; t += blockDim.x)
  {
    const int dv = a.disp[t];
    displ[t] = dv;
    const int m = dv / GS + mD;
    if (m >= 0 && m < 32)
      dinv[m] = t;
  }
  for (int t = threadIdx.x; t < 64; t += blockDim.x)
    ltab[t] = a.ltab[t];
  __syncthreads();

  // block -> (particle, group of 4 orientation*CTF): particle chunks of a.pchunk; inside a chunk the particle index
  // runs fastest, then the group.  Workgroups go round-robin over the 8 XCDs, so with a chunk size that is a
  // multiple of 8 a particle always lands on the same XCD, and the ~96 blocks resident per XCD cover
  // (pchunk/8 particles) x (a few groups): every particle line is then shared through that XCD's L2 by several
  // groups and every conv line by pchunk/8 particles, instead of each particle line being fetched from Infinity
  // Cache/HBM once per group.
  int p, ocg;
  {
    const int ocGroups = (a.nOC + 3) >> 2;
    const int per = a.pchunk * ocGroups;
    int c = blockIdx.x / per;
    const int nch = (a.nMaps + a.pchunk - 1) / a.pchunk;
    c = min(c, nch - 1);
    const int rem = blockIdx.x - c * per;
    const int pc = min(a.pchunk, a.nMaps - c * a.pchunk);
    ocg = rem / pc;
    p = c * a.pchunk + (rem - ocg * pc);
  }
  const int oc_raw = ocg * 4 + wave;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const size_t M = (size_t) N * H;
  // buffer descriptors built from wave-uniform values only (blockIdx / readfirstlane'd wave id)
  // timing-only ablation builds (never shipped): a zero-record descriptor drops the loads of one operand while
  // the instruction stream and waits stay (cdna_hip_programming.md, profiling: pricing one buffer's traffic)
#ifndef BIOEM_ABLATE_F
#define BIOEM_ABLATE_F 0
#endif
#ifndef BIOEM_ABLATE_C
#define BIOEM_ABLATE_C 0
#endif
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.ref + (size_t) p * M), 0,
                                                       BIOEM_ABLATE_F ? 0 : (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.conv + (size_t) oc * M), 0,
                                                       BIOEM_ABLATE_C ? 0 : (int) (M * sizeof(float2)), 0x00020000);

  // window lanes
  const int nd = a.nd;
  const int G = 64 / nd;
  const int nr = (nd + G - 1) / G;
  const int iy = lane % nd, grp = lane / nd;
  const bool wactive = grp < G;
  const int dy = displ[iy];
  const int step = dy < 0 ? dy + N : dy;
  // static window (the +-10 px, grid 1 case): the window rows are -mD..mD and every lane group owns exactly
  // NR CONSECUTIVE rows of it in sorted order, whatever the visiting order of the algorithm (ALGO 1 visits
  // 0..maxD, -maxD..-1); dinv[] translates back to visiting ranks for the arg-max bookkeeping
  const bool is_static = (nr == NR) && (nd == G * NR) && (nd == 2 * mD + 1);
  float acc[NR];
#pragma unroll
  for (int r = 0; r < NR; r++)
    acc[r] = 0.f;
  // T row (in float2 units) of accumulator r of this lane; idle lanes (grp >= G) read rows 0.. and are dropped
  // later.  Only the first is kept live across the column loop: the static window uses base + r*TS, the general
  // one re-reads its rows from the displacement list per block.
  auto row_of = [&](int r) -> int {
    int ix = wactive ? grp * nr + r : r;
    if (ix >= nd)
      ix = nd - 1;
    return (displ[ix] / GS + WD) * TS;
  };
  const int rowbase = is_static ? ((wactive ? grp : 0) * NR - mD + WD) * TS : row_of(0);

  const int nblk = NYQ ? (H - 1) / 64 : (H + 63) / 64;
  // Operand stream (software pipelined across k1 iterations AND column blocks): the (k1, k2-pair) loads of a
  // lane walk t = k1*16 + k2p with a constant stride of H float4; a 4-deep ring of (F, C) pairs keeps 8 dwordx4
  // loads (8 KiB per wave) in flight, re-issued as soon as a slot is consumed.  The ring runs on into the first
  // rows of the NEXT column block, so those loads fly during the T exchange / window phase of this block.
  // Addressing: buffer loads -- 128-bit descriptor (SGPRs), one 32-bit lane offset (VGPR), row offset in an SGPR.
  const int ttotal = R2 * N1;
  const unsigned rowbytes = (unsigned) H * 16u;
  u32x4 rf[RD], rc[RD];
  {
    const unsigned lo0 = (unsigned) (lane < H ? lane : H - 1) * 16u;
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      rf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, lo0, (unsigned) t * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, lo0, (unsigned) t * rowbytes, 0);
    }
  }
  for (int blk = 0; blk < nblk; blk++)
  {
    const int ky = blk * 64 + lane;
    const int kyc = ky < H ? ky : H - 1;
    const unsigned laneoff = (unsigned) kyc * 16u;
    const int kyn = ky + 64 < H ? ky + 64 : H - 1;
    const unsigned laneoff_next = (unsigned) kyn * 16u;
    const bool has_next = blk + 1 < nblk;
    float Tr[NW], Ti[NW];
#pragma unroll
    for (int d = 0; d < NW; d++)
    {
      Tr[d] = 0.f;
      Ti[d] = 0.f;
    }
    for (int k1 = 0; k1 < N1; k1++)
    {
      float xr[R], xi[R];
      // the 2*WD+1 recombination twiddles of this k1 are contiguous: a few wide scalar loads, issued early
      float2 wk[NW];
      const float2 *twk = a.twk + (size_t) k1 * NW;
#pragma unroll
      for (int d = 0; d < NW; d++)
        wk[d] = twk[d];
#pragma unroll
      for (int k2p = 0; k2p < R2; k2p++)
      {
        const float4 f = as_float4(rf[k2p % RD]);
        const float4 c = as_float4(rc[k2p % RD]);
        // X = conv * conj(ref)   (bioem.cpp:1452-1455)
        xr[FFT_IN(2 * k2p)] = fmaf(c.x, f.x, c.y * f.y);
        xi[FFT_IN(2 * k2p)] = fmaf(c.y, f.x, -(c.x * f.y));
        xr[FFT_IN(2 * k2p + 1)] = fmaf(c.z, f.z, c.w * f.w);
        xi[FFT_IN(2 * k2p + 1)] = fmaf(c.w, f.z, -(c.z * f.w));
        int tn = k1 * R2 + k2p + RD;
        unsigned vo = laneoff;
        if (tn >= ttotal)
        { // last steps of this block: run on into the next block (or re-read the last row at the very end)
          tn = has_next ? tn - ttotal : ttotal - 1;
          vo = has_next ? laneoff_next : laneoff;
        }
        rf[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, vo, (unsigned) tn * rowbytes, 0);
        rc[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, vo, (unsigned) tn * rowbytes, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      FFT_RUN(xr, xi);
      // recombination of the N1 sub-transforms for the displacement window only:
      //   T[dx] += w_N^(dx*k1) * y_k1[dx mod 32]
#pragma unroll
      for (int d = -WD; d <= WD; d++)
      {
        const int pos = FFT_OUT((d * GS) & (R - 1));
        const float2 w = wk[d + WD];
        float tr = Tr[d + WD], ti = Ti[d + WD];
        tr = fmaf(xr[pos], w.x, tr);
        tr = fmaf(-xi[pos], w.y, tr);
        ti = fmaf(xr[pos], w.y, ti);
        ti = fmaf(xi[pos], w.x, ti);
        Tr[d + WD] = tr;
        Ti[d + WD] = ti;
      }
    }
    // FFTW c2r convention: columns 0 and N/2 enter once (real part only after the ky pass), others twice
    float wgt = 2.f;
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H)
      wgt = 0.f;
    // T block of THIS wave only: LDS operations of one wave execute in order, so a wave-level fence (no
    // s_barrier) is enough; BIOEM_BLOCK_BARRIER=1 restores block barriers (keeps the 4 waves in lock-step)
    WAVE_OR_BLOCK_SYNC(); // previous block's window reads are done
#pragma unroll
    for (int d = 0; d < NW; d++)
      Tl[d * TS + lane] = make_float2(Tr[d] * wgt, Ti[d] * wgt);
    WAVE_OR_BLOCK_SYNC();
    const int idx0 = (int) (((long long) blk * 64 * step) % N);
    if (is_static)
    {
      const int rowoff[NR] = {rowbase};
      window_accumulate<NR, true, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc);
    }
    else
    {
      int rowoff[NR];
#pragma unroll
      for (int r = 0; r < NR; r++)
        rowoff[r] = row_of(r);
      window_accumulate<NR, false, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc);
    }
  }
  if (NYQ)
  {
    const float *tq = a.tnyq + ((size_t) p * a.ldPart + oc) * NW;
    const float sg = (dy & 1) ? -1.f : 1.f;
#pragma unroll
    for (int r = 0; r < NR; r++)
      acc[r] = fmaf(sg, tq[is_static ? rowbase / TS + r : row_of(r) / TS], acc[r]);
  }

  const bioem_hip_param5 q = a.params[oc];
  const float sumref = a.sumRef[p], sumsqref = a.sumsqRef[p];
  double t2, prior;
  logpro_consts(a.pd, q, t2, prior);
  const float Np = a.pd.Ntotpi;
  const double A = (double) (3 - Np) * 0.5;
  const float nn = (float) (N * N);
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
#pragma unroll
  for (int r = 0; r < NR; r++)
  {
    const int ixs = grp * nr + r; // position in the lane-group order; ix = visiting rank of that displacement
    if (r < nr && wactive && ixs < nd)
    {
      const int ix = is_static ? dinv[ixs] : ixs;
      const float cc = acc[r] / nn;
      // bioem_algorithm.h:32-36, float expression in the reference's order
      const float firstele = Np * (sumsqref * q.sumsquareC - cc * cc) + 2 * sumref * q.sumC * cc -
                             sumsqref * q.sumC * q.sumC - sumref * sumref * q.sumsquareC;
      double lp = A * log_of_float(firstele, ltab) + t2;
      lp -= prior;
      lsef_push(L, lp, ix * nd + iy, cc, a.algo);
    }
  }
  lsef_wave_reduce(L);
  if (lane == 0 && oc_valid)
  {
    Partial r;
    r.sumExp = L.s;
    r.best = L.m;
    r.id = L.id;
    r.value = L.val;
    r.pad = 0;
    a.partials[(size_t) p * a.ldPart + oc] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// Nyquist-column rows for the fast kernel's NYQ mode: thread = one (particle, orientation*CTF) pair, tile of
// 16 x 16 pairs per block (each operand line is shared by 16 threads).
//   tnyq[p][oc][m + WD] = Re sum_kx conv[oc][kx][N/2] * conj(ref[p][kx][N/2]) * w_N^(kx m gs),  m = -WD..WD
// The twiddle index is uniform over the block (LDS broadcast reads).
// ------------------------------------------------------------------------------------------------
template <int WD>
__global__ __launch_bounds__(256) void k_nyquist_rows(const CompareArgs a)
{
  constexpr int NW = 2 * WD + 1;
  __shared__ float2 twl[1024];
  const int N = a.N, H = a.H, N1 = a.N1;
  const int R2 = N / (2 * N1);
  for (int t = threadIdx.x; t < N; t += blockDim.x)
    twl[t] = a.tw[t];
  __syncthreads();
  const int tilesOC = (a.nOC + 15) / 16;
  const int tp = blockIdx.x / tilesOC, to = blockIdx.x - tp * tilesOC;
  const int p = tp * 16 + (threadIdx.x >> 4), oc = to * 16 + (threadIdx.x & 15);
  const bool valid = p < a.nMaps && oc < a.nOC;
  const size_t M = (size_t) N * H;
  const float2 *F = a.ref + (size_t) (valid ? p : 0) * M;
  const float2 *C = a.conv + (size_t) (valid ? oc : 0) * M;
  float acc[NW];
#pragma unroll
  for (int d = 0; d < NW; d++)
    acc[d] = 0.f;
  for (int k1 = 0; k1 < N1; k1++)
    for (int k2p = 0; k2p < R2; k2p++)
    {
      // the two k2 of a pair are adjacent in the comparison layout: one 16-byte load per operand
      const size_t li = ((size_t) (k1 * R2 + k2p) * H + N / 2) * 2;
      const float4 c = *reinterpret_cast<const float4 *>(C + li);
      const float4 f = *reinterpret_cast<const float4 *>(F + li);
      // X = conv * conj(ref)   (bioem.cpp:1452-1455)
      const float x0r = fmaf(c.x, f.x, c.y * f.y), x0i = fmaf(c.y, f.x, -(c.x * f.y));
      const float x1r = fmaf(c.z, f.z, c.w * f.w), x1i = fmaf(c.w, f.z, -(c.z * f.w));
      const int kx0 = N1 * (2 * k2p) + k1, kx1 = kx0 + N1;
      // w^(kx * dx) for dx = -WD*gs, then dx -> dx + gs
      const int s0 = (int) (((long long) kx0 * a.gs) % N), s1 = (int) (((long long) kx1 * a.gs) % N);
      int i0 = (int) ((N - ((long long) s0 * WD) % N) % N), i1 = (int) ((N - ((long long) s1 * WD) % N) % N);
#pragma unroll
      for (int d = 0; d < NW; d++)
      {
        const float2 w0 = twl[i0], w1 = twl[i1];
        float v = acc[d];
        v = fmaf(x0r, w0.x, v);
        v = fmaf(-x0i, w0.y, v);
        v = fmaf(x1r, w1.x, v);
        v = fmaf(-x1i, w1.y, v);
        acc[d] = v;
        i0 += s0;
        if (i0 >= N)
          i0 -= N;
        i1 += s1;
        if (i1 >= N)
          i1 -= N;
      }
    }
  if (valid)
  {
    float *o = a.tnyq + ((size_t) p * a.ldPart + oc) * NW;
#pragma unroll
    for (int d = 0; d < NW; d++)
      o[d] = acc[d];
  }
}

// ------------------------------------------------------------------------------------------------
// generic comparison kernel: any N, any maxD.  Reference layout.  One wave per comparison;
// T[dx][ky] = sum_kx X[kx][ky] w^(kx dx) by direct summation.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compare_generic(const CompareArgs a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H;
  const int Hs = (H + 1) & ~1;
  const int NW = 2 * a.maxD + 1;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float2 *Tl = Tall + (size_t) wave * NW * Hs;
  for (int t = threadIdx.x; t <= N; t += blockDim.x)
    twl[t] = a.tw[t];
  for (int t = threadIdx.x; t < a.nd; t += blockDim.x)
    displ[t] = a.disp[t];
  __syncthreads();

  const int p = blockIdx.x % a.nMaps;
  const int ocg = blockIdx.x / a.nMaps;
  const int oc_raw = ocg * 4 + wave;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const size_t M = (size_t) N * H;
  const float2 *F = a.ref + (size_t) p * M;
  const float2 *C = a.conv + (size_t) oc * M;

  for (int e = lane; e < NW * Hs; e += 64)
  {
    const int dxi = e / Hs, ky = e - dxi * Hs;
    float tr = 0.f, ti = 0.f;
    if (ky < H)
    {
      const int dx = dxi - a.maxD;
      const int step = dx < 0 ? dx + N : dx;
      int idx = 0;
      for (int kx = 0; kx < N; kx++)
      {
        const float2 c = C[(size_t) kx * H + ky], f = F[(size_t) kx * H + ky];
        const float xr = fmaf(c.x, f.x, c.y * f.y);
        const float xi = fmaf(c.y, f.x, -(c.x * f.y));
        const float2 w = twl[idx];
        tr = fmaf(xr, w.x, tr);
        tr = fmaf(-xi, w.y, tr);
        ti = fmaf(xr, w.y, ti);
        ti = fmaf(xi, w.x, ti);
        idx += step;
        if (idx >= N)
          idx -= N;
      }
      float wgt = 2.f;
      if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
        wgt = 1.f;
      tr *= wgt;
      ti *= wgt;
    }
    Tl[dxi * Hs + ky] = make_float2(tr, ti);
  }
  __syncthreads();

  const bioem_hip_param5 q = a.params[oc];
  double t2, prior;
  logpro_consts(a.pd, q, t2, prior);
  const float sumref = a.sumRef[p], sumsqref = a.sumsqRef[p];
  const float nn = (float) (N * N);
  Lse L;
  lse_init(L);
  const int nd = a.nd;
  for (int e = lane; e < nd * nd; e += 64)
  {
    const int ix = e / nd, iy = e - ix * nd;
    const int dy = displ[iy];
    const int step = dy < 0 ? dy + N : dy;
    const float2 *row = Tl + (size_t) (displ[ix] + a.maxD) * Hs;
    float acc = 0.f;
    int idx = 0;
    for (int ky = 0; ky < H; ky++)
    {
      const float2 t = row[ky], w = twl[idx];
      acc = fmaf(t.x, w.x, acc);
      acc = fmaf(-t.y, w.y, acc);
      idx += step;
      if (idx >= N)
        idx -= N;
    }
    const float value = acc / nn;
    const double lp = logpro_eval(a.pd, q, value, sumref, sumsqref, t2, prior);
    lse_push(L, lp, e, value, a.algo);
  }
  lse_wave_reduce(L);
  if (lane == 0 && oc_valid)
  {
    Partial r;
    r.sumExp = L.s;
    r.best = L.m;
    r.id = L.id;
    r.value = L.val;
    r.pad = 0;
    a.partials[(size_t) p * a.ldPart + oc] = r;
  }
}

// ------------------------------------------------------------------------------------------------
// fold: one thread per particle walks its partials in (orientation, CTF) order.
// bioem_algorithm.h:94-141 / bioem.cpp:1527-1600.
// ------------------------------------------------------------------------------------------------
__global__ void k_fold(const Partial *__restrict__ partials, int ldPart, int nOC, int nMaps,
                       const bioem_hip_param5 *__restrict__ params, const float *__restrict__ sumRef,
                       const int *__restrict__ disp, int nd, PD pd, int orient0, int conv0, int convPerOrient,
                       bioem_hip_prob_map *__restrict__ pmap, bioem_hip_prob_angle *__restrict__ pang)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nMaps)
    return;
  bioem_hip_prob_map pm = pmap[p];
  const float sumref = sumRef[p];
  const Partial *P = partials + (size_t) p * ldPart;
  for (int oc = 0; oc < nOC; oc++)
  {
    const Partial r = P[oc];
    const int iOrient = orient0 + oc / convPerOrient;
    const int iConv = conv0 + oc % convPerOrient;
    const double lp = (double) r.best;
    if (pm.Constoadd < lp)
    {
      pm.Total *= exp(-lp + pm.Constoadd);
      pm.Constoadd = lp;
      const int ix = r.id / nd, iy = r.id - ix * nd;
      pm.max_prob_cent_x = -disp[ix];
      pm.max_prob_cent_y = -disp[iy];
      pm.max_prob_orient = iOrient;
      pm.max_prob_conv = iConv;
      const bioem_hip_param5 q = params[oc];
      const float value = r.value;
      pm.max_prob_norm = -(-q.sumC * sumref + pd.Ntotpi * value) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
      pm.max_prob_mu = -(-q.sumC * value + q.sumsquareC * sumref) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
    }
    pm.Total += r.sumExp * exp(lp - pm.Constoadd);
    if (pd.writeAngles)
    {
      bioem_hip_prob_angle pa = pang[(size_t) iOrient * nMaps + p];
      if (pa.ConstAngle < lp)
      {
        pa.forAngles *= exp(-lp + pa.ConstAngle);
        pa.ConstAngle = lp;
      }
      pa.forAngles += r.sumExp * exp(lp - pa.ConstAngle);
      pang[(size_t) iOrient * nMaps + p] = pa;
    }
  }
  pmap[p] = pm;
}

// ------------------------------------------------------------------------------------------------
// wave-parallel fold (no WRITE_PROB_ANGLES): one wave per particle; lane l folds a contiguous chunk of
// (orientation, CTF) partials in order, the 64 chunk results are merged by a shuffle reduction that keeps
// the FIRST maximum (lowest index), then combined with the running state exactly like the sequential fold.
// The log-sum-exp merge is associative, so the result equals k_fold's up to double rounding (1e-16).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fold_wave(const Partial *__restrict__ partials, int ldPart, int nOC,
                                                   int nMaps, const bioem_hip_param5 *__restrict__ params,
                                                   const float *__restrict__ sumRef, const int *__restrict__ disp,
                                                   int nd, PD pd, int orient0, int conv0, int convPerOrient,
                                                   bioem_hip_prob_map *__restrict__ pmap)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wave;
  if (p >= nMaps)
    return;
  const Partial *P = partials + (size_t) p * ldPart;
  const int chunk = (nOC + 63) / 64;
  const int b = lane * chunk, e = min(nOC, b + chunk);
  double m = -INFINITY, sacc = 0.;
  int idx = 0x7fffffff;
  for (int oc = b; oc < e; oc++)
  {
    const Partial r = P[oc];
    const double lp = (double) r.best;
    if (m < lp)
    {
      sacc = (m == -INFINITY) ? 0. : sacc * exp(m - lp);
      m = lp;
      idx = oc;
    }
    sacc += r.sumExp * exp(lp - m);
  }
  for (int off = 32; off > 0; off >>= 1)
  {
    const double m2 = __shfl_xor(m, off);
    const double s2 = __shfl_xor(sacc, off);
    const int i2 = __shfl_xor(idx, off);
    if (m2 > m || (m2 == m && i2 < idx))
    {
      sacc = ((m == -INFINITY) ? 0. : sacc * exp(m - m2)) + s2;
      m = m2;
      idx = i2;
    }
    else
      sacc += (m2 == -INFINITY) ? 0. : s2 * exp(m2 - m);
  }
  if (lane == 0 && idx != 0x7fffffff)
  {
    bioem_hip_prob_map pm = pmap[p];
    if (pm.Constoadd < m)
    {
      pm.Total *= exp(-m + pm.Constoadd);
      pm.Constoadd = m;
      const Partial r = P[idx];
      const int ix = r.id / nd, iy = r.id - ix * nd;
      pm.max_prob_cent_x = -disp[ix];
      pm.max_prob_cent_y = -disp[iy];
      pm.max_prob_orient = orient0 + idx / convPerOrient;
      pm.max_prob_conv = conv0 + idx % convPerOrient;
      const bioem_hip_param5 q = params[idx];
      const float sumref = sumRef[p];
      const float value = r.value;
      pm.max_prob_norm = -(-q.sumC * sumref + pd.Ntotpi * value) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
      pm.max_prob_mu = -(-q.sumC * value + q.sumsquareC * sumref) / (q.sumC * q.sumC - q.sumsquareC * pd.Ntotpi);
    }
    pm.Total += sacc * exp(m - pm.Constoadd);
    pmap[p] = pm;
  }
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
size_t compare_lds_bytes(int N, int H, int NW, int waves)
{ // generic kernel: tables + per-wave T [NW][Hs]
  const int Hs = (H + 1) & ~1;
  return (size_t) ((N + 2) & ~1) * 8 + 256 + (size_t) waves * NW * Hs * 8;
}

size_t fast_lds_bytes(int N, int NW, int waves)
{ // fast kernel: twiddles + displacement list + log table + per-wave T block [NW][66]
  return (size_t) ((N + 2) & ~1) * 8 + 256 + 1024 + (size_t) waves * NW * 66 * 8;
}

hipEvent_t get_event(bioem_hip_ctx *h)
{
  if (!h->evPool.empty())
  {
    hipEvent_t e = h->evPool.back();
    h->evPool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess)
    return nullptr;
  return e;
}

void drain_events(bioem_hip_ctx *h)
{
  for (auto &pr : h->evPending)
  {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
      h->compareMs += (double) ms;
    h->evPool.push_back(pr.first);
    h->evPool.push_back(pr.second);
  }
  h->evPending.clear();
}

struct BatchBuf
{
  double *projReal;
  double *tempDen;
  double2 *rowSpec;
  float2 *specRef;
  float *scratch;
  float2 *conv;
  bioem_hip_param5 *params;
};

BatchBuf batch_buf(bioem_hip_ctx *h, int which)
{
  BatchBuf b;
  if (which == 0)
    b = {h->dProjReal, h->dTempDen, h->dRowSpec, h->dSpecRef, h->dScratch, h->dConv, h->dParams};
  else
    b = {h->dProjReal2, h->dTempDen2, h->dRowSpec2, h->dSpecRef2, h->dScratch2, h->dConv2, h->dParams2};
  return b;
}

// the fast-kernel instantiation for a window half width (10 or 15) and register-FFT length (32, 16, 8)
typedef void (*fast_kernel_t)(const CompareArgs);
template <int WD, int GS>
fast_kernel_t fast_kernel_r(int R, bool nyq)
{
  if (nyq) // N/2 a multiple of 64 implies R = 32
    return k_compare_fast<WD, 32, true, GS>;
  return R == 32   ? k_compare_fast<WD, 32, false, GS>
         : R == 16 ? k_compare_fast<WD, 16, false, GS>
         : R == 8  ? k_compare_fast<WD, 8, false, GS>
         : R == 4  ? k_compare_fast<WD, 4, false, GS>
                   : k_compare_fast<WD, 2, false, GS>;
}

template <int WD>
fast_kernel_t fast_kernel_g(int R, bool nyq, int gs)
{
  return gs == 1 ? fast_kernel_r<WD, 1>(R, nyq) : gs == 2 ? fast_kernel_r<WD, 2>(R, nyq)
         : gs == 3 ? fast_kernel_r<WD, 3>(R, nyq) : fast_kernel_r<WD, 4>(R, nyq);
}

fast_kernel_t fast_kernel(int winD, int R, bool nyq, int gs)
{
  return winD == 5 ? fast_kernel_g<5>(R, nyq, gs) : winD == 10 ? fast_kernel_g<10>(R, nyq, gs) : fast_kernel_g<15>(R, nyq, gs);
}

int launch_compare_fold(bioem_hip_ctx *h, const BatchBuf &bb, int nOC, int orient0, int conv0, int convPerOrient)
{
  CompareArgs a;
  a.ref = h->dRef;
  a.conv = bb.conv;
  a.params = bb.params;
  a.sumRef = h->dSumRef;
  a.sumsqRef = h->dSumsqRef;
  a.tw = h->dTw;
  a.disp = h->dDisp;
  a.ltab = h->dLtab;
  a.twk = h->dTwk;
  a.tnyq = h->dTnyq;
  a.partials = h->dPartials;
  a.ldPart = h->maxOC;
  a.N = h->N;
  a.H = h->H;
  a.N1 = h->N1;
  a.nd = h->nd;
  a.maxD = h->pd.maxDisplaceCenter;
  a.nOC = nOC;
  a.nMaps = h->nMaps;
  a.algo = h->algo;
  a.pd = h->pd;
  a.gs = h->gs;
  a.pchunk = h->pchunk > 0 ? std::min(h->pchunk, h->nMaps) : h->nMaps;
  const int ocGroups = (nOC + 3) / 4;
  const dim3 grid((unsigned) ((size_t) ocGroups * h->nMaps));
  hipEvent_t e0 = get_event(h), e1 = get_event(h);
  if (!e0 || !e1)
  {
    h->err = "hipEventCreate failed";
    return 1;
  }
  HIP_CHECK(h, hipEventRecord(e0, h->stream));
  if (h->fast)
  {
    const int NW = 2 * h->winD + 1;
    const size_t lds = fast_lds_bytes(h->N, NW, 4);
    if (h->nyq)
    {
      const dim3 gridq((unsigned) (((size_t) (h->nMaps + 15) / 16) * ((nOC + 15) / 16)));
      if (h->winD == 5)
        hipLaunchKernelGGL(k_nyquist_rows<5>, gridq, dim3(256), 0, h->stream, a);
      else if (h->winD == 10)
        hipLaunchKernelGGL(k_nyquist_rows<10>, gridq, dim3(256), 0, h->stream, a);
      else
        hipLaunchKernelGGL(k_nyquist_rows<15>, gridq, dim3(256), 0, h->stream, a);
    }
    hipLaunchKernelGGL(fast_kernel(h->winD, 2 * h->fast, h->nyq, h->gs), grid, dim3(256), lds, h->stream, a);
  }
  else
  {
    const size_t lds = compare_lds_bytes(h->N, h->H, 2 * h->pd.maxDisplaceCenter + 1, 4);
    hipLaunchKernelGGL(k_compare_generic, grid, dim3(256), lds, h->stream, a);
  }
  HIP_CHECK(h, hipGetLastError());
  HIP_CHECK(h, hipEventRecord(e1, h->stream));
  h->evPending.push_back({e0, e1});
  h->launches++;
  h->comparisons += (long long) nOC * h->nMaps;
  bioem_hip_prob_map *pmap = reinterpret_cast<bioem_hip_prob_map *>(h->dProb);
  bioem_hip_prob_angle *pang = reinterpret_cast<bioem_hip_prob_angle *>(h->dProb + sizeof(bioem_hip_prob_map) * h->nMaps);
  if (h->pd.writeAngles)
    hipLaunchKernelGGL(k_fold, dim3((h->nMaps + 127) / 128), dim3(128), 0, h->stream, h->dPartials, h->maxOC, nOC,
                       h->nMaps, bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, pmap,
                       pang);
  else
    hipLaunchKernelGGL(k_fold_wave, dim3((h->nMaps + 3) / 4), dim3(256), 0, h->stream, h->dPartials, h->maxOC, nOC,
                       h->nMaps, bb.params, h->dSumRef, h->dDisp, h->nd, h->pd, orient0, conv0, convPerOrient, pmap);
  HIP_CHECK(h, hipGetLastError());
  if (h->evPending.size() > 512)
    drain_events(h);
  return 0;
}

void dft_split(int N, int &A, int &B)
{
  A = 1;
  for (int d = 1; d * d <= N; d++)
    if (N % d == 0)
      A = d;
  B = N / A;
}

// r2c of nImg images (either double projection maps with tempden scaling, or float maps) into bb.specRef
int run_r2c(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, const double *srcD, const float *srcF, int nImg)
{
  const int N = h->N, H = h->H;
  int A, B;
  dft_split(N, A, B);
  hipLaunchKernelGGL(k_dft_rows, dim3(N, nImg), dim3(128), sizeof(double) * (3 * N + 2), st, srcD, srcF, bb.tempDen,
                     h->NormDen, N, H, A, B, h->dTwD, bb.rowSpec);
  HIP_CHECK(h, hipGetLastError());
  hipLaunchKernelGGL(k_dft_cols, dim3(H, nImg), dim3(256), sizeof(double2) * 2 * N, st, bb.rowSpec, N, H, A, B,
                     h->dTwD, bb.specRef);
  HIP_CHECK(h, hipGetLastError());
  return 0;
}

int project_batch(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, int o0, int nO)
{
  const int N = h->N;
  HIP_CHECK(h, hipMemsetAsync(bb.projReal, 0, sizeof(double) * (size_t) nO * N * N, st));
  HIP_CHECK(h, hipMemsetAsync(bb.tempDen, 0, sizeof(double) * nO, st));
  hipLaunchKernelGGL(k_project, dim3((h->nPts + 255) / 256, nO), dim3(256), 0, st, h->dPts, h->nPts, h->dAngles, o0,
                     h->isQuat, N, h->pixelSize, h->shiftX, h->shiftY, bb.projReal, bb.tempDen);
  HIP_CHECK(h, hipGetLastError());
  return run_r2c(h, bb, st, bb.projReal, nullptr, nO);
}

int convolve_batch(bioem_hip_ctx *h, const BatchBuf &bb, hipStream_t st, int nO)
{
  hipLaunchKernelGGL(k_convolve, dim3(h->nCTF, nO), dim3(256), 0, st, bb.specRef, h->dCTF, h->dCtfParam, h->N, h->H,
                     h->fast, h->N1, h->nCTF, bb.conv, bb.scratch, bb.params);
  HIP_CHECK(h, hipGetLastError());
  return 0;
}

} // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int bioem_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

size_t bioem_hip_prob_size(int nMaps, int nAngles, int writeAngles)
{
  size_t size = sizeof(bioem_hip_prob_map);
  if (writeAngles)
    size += (size_t) nAngles * sizeof(bioem_hip_prob_angle);
  return (size_t) nMaps * size;
}

const char *bioem_hip_last_error(bioem_hip_handle h) { return h ? h->err.c_str() : "null handle"; }

int bioem_hip_create(bioem_hip_handle *out, int device, const bioem_hip_param_device *pd, int nMaps, int nAngles,
                     int nCTF, int algo)
{
  if (!out || !pd)
    return 2;
  bioem_hip_ctx *h = new bioem_hip_ctx;
  *out = h;
  h->device = device;
  h->pd = *pd;
  h->nMaps = nMaps;
  h->nAngles = nAngles;
  h->nCTF = nCTF;
  h->algo = algo;
  const int N = pd->NumberPixels;
  h->N = N;
  h->H = N / 2 + 1;
  h->M = N * h->H;
  if (N < 2 || nMaps < 1 || nAngles < 1 || nCTF < 1 || pd->maxDisplaceCenter < 0 || pd->GridSpaceCenter < 1 ||
      pd->maxDisplaceCenter >= N / 2)
  {
    h->err = "invalid configuration (need N>=2, nMaps,nAngles,nCTF>=1, 0<=maxD<N/2, grid>=1)";
    return 2;
  }
  HIP_CHECK(h, hipSetDevice(device));
  {
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prHigh));
  }

  // displacement list per axis in the reference's visiting order
  const int maxD = pd->maxDisplaceCenter, g = pd->GridSpaceCenter;
  if (algo == 1)
  { // bioem_algorithm.h:156-197
    for (int c = 0; c <= maxD; c += g)
      h->disp.push_back(c);
    for (int c = N - maxD; c < N; c += g)
      h->disp.push_back(c - N);
  }
  else
  { // bioem.cpp:1477-1485
    const int NxDisp = 2 * (maxD / g) + 1;
    for (int m = 0; m < NxDisp; m++)
      h->disp.push_back(m * g - maxD);
  }
  h->nd = (int) h->disp.size();

  // fast path: N = N1 * R with R the largest of 32/16/8/4/2 dividing N; h->fast holds R/2 (rows per k1 step)
  // window rows: row m holds displacement m * gs, gs = gcd of all offsets (1..4 are instantiated), so a coarse grid
  // with maxD a multiple of the spacing reaches +-15*gs pixels
  {
    int gg = 0;
    for (int d : h->disp)
    {
      int x = d < 0 ? -d : d, y = gg;
      while (y)
      {
        const int t = x % y;
        x = y;
        y = t;
      }
      gg = x;
    }
    h->gs = (gg >= 1 && gg <= 4) ? gg : 1;
  }
  h->fast = 0;
  if (N % 2 == 0 && N >= 8 && maxD / h->gs <= 15 && h->nd <= 31)
    h->fast = (N % 32 == 0) ? 16 : (N % 16 == 0) ? 8 : (N % 8 == 0) ? 4 : (N % 4 == 0) ? 2 : 1;
  h->N1 = h->fast ? N / (2 * h->fast) : 0;
#ifndef BIOEM_NYQUIST_SPLIT
#define BIOEM_NYQUIST_SPLIT 1
#endif
  h->nyq = BIOEM_NYQUIST_SPLIT && h->fast && (N / 2) % 64 == 0;
  // window template: 2*winD+1 rows, nd <= rows (ALGO 1 with maxD % grid != 0 visits up to 2*(maxD/grid)+2 offsets)
  {
    const int mD = maxD / h->gs;
    h->winD = (mD <= 5 && h->nd <= 11) ? 5 : (mD <= 10 && h->nd <= 21) ? 10 : 15;
  }
  // LDS budget check
  {
    const size_t lds = h->fast ? fast_lds_bytes(N, 2 * h->winD + 1, 4) : compare_lds_bytes(N, h->H, 2 * maxD + 1, 4);
    if (lds > 160 * 1024)
    {
      h->err = "configuration exceeds the 160 KiB LDS budget of the comparison kernel";
      return 2;
    }
    if (h->fast)
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(fast_kernel(h->winD, 2 * h->fast, h->nyq, h->gs)),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    else
    {
      const size_t ldsg = compare_lds_bytes(N, h->H, 2 * maxD + 1, 4);
      HIP_CHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_compare_generic),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsg));
    }
  }

  // batch sizing: conv buffer <= ~96 MiB, partial buffer <= ~128 MiB
  const size_t M = (size_t) h->M;
  size_t ocCap = (96u << 20) / (M * sizeof(float2));
  const size_t partCap = (128u << 20) / ((size_t) nMaps * sizeof(Partial));
  if (ocCap > partCap)
    ocCap = partCap;
  int OB = (int) (ocCap / (size_t) nCTF);
  if (OB < 1)
    OB = 1;
  if (OB > 64)
    OB = 64;
  if (getenv("BIOEM_PCHUNK")) // tuning knob: particle chunk of the comparison kernel's block order
    h->pchunk = atoi(getenv("BIOEM_PCHUNK"));
  if (getenv("BIOEM_BATCH_ORIENTATIONS")) // tuning knob: orientations per batch (conv buffer = OB*nCTF spectra)
    OB = std::max(1, std::min(OB, atoi(getenv("BIOEM_BATCH_ORIENTATIONS"))));
  if (OB > nAngles)
    OB = nAngles;
  h->OB = OB;
  h->maxOC = OB * nCTF;
  h->chunkB = OB > 32 ? OB : 32;

  HIP_CHECK(h, hipMalloc(&h->dRef, sizeof(float2) * M * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dSumRef, sizeof(float) * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dSumsqRef, sizeof(float) * nMaps));
  HIP_CHECK(h, hipMalloc(&h->dCTF, sizeof(float2) * M * nCTF));
  HIP_CHECK(h, hipMalloc(&h->dCtfParam, sizeof(float) * 3 * nCTF));
  HIP_CHECK(h, hipMalloc(&h->dAngles, sizeof(float4) * nAngles));
  HIP_CHECK(h, hipMalloc(&h->dTw, sizeof(float2) * (N + 1)));
  HIP_CHECK(h, hipMalloc(&h->dTwD, sizeof(double2) * N));
  HIP_CHECK(h, hipMalloc(&h->dDisp, sizeof(int) * h->nd));
  HIP_CHECK(h, hipMalloc(&h->dLtab, sizeof(double2) * 64));
  HIP_CHECK(h, hipMalloc(&h->dProjReal, sizeof(double) * (size_t) h->chunkB * N * N));
  HIP_CHECK(h, hipMalloc(&h->dTempDen, sizeof(double) * h->chunkB));
  HIP_CHECK(h, hipMalloc(&h->dRowSpec, sizeof(double2) * (size_t) h->chunkB * M));
  HIP_CHECK(h, hipMalloc(&h->dSpecRef, sizeof(float2) * (size_t) h->chunkB * M));
  HIP_CHECK(h, hipMalloc(&h->dScratch, sizeof(float) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dConv, sizeof(float2) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dParams, sizeof(bioem_hip_param5) * h->maxOC));
  HIP_CHECK(h, hipMalloc(&h->dPartials, sizeof(Partial) * (size_t) nMaps * h->maxOC));
  if (h->nyq)
    HIP_CHECK(h, hipMalloc(&h->dTnyq, sizeof(float) * (size_t) nMaps * h->maxOC * (2 * h->winD + 1)));
  h->probBytes = bioem_hip_prob_size(nMaps, nAngles, pd->writeAngles);
  HIP_CHECK(h, hipMalloc(&h->dProb, h->probBytes));
  {
    // projection/convolution are filler work: lowest priority so that comparison blocks win the CUs
    int prLow = 0, prHigh = 0;
    HIP_CHECK(h, hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
    HIP_CHECK(h, hipStreamCreateWithPriority(&h->prepStream, hipStreamNonBlocking, prLow));
  }
  HIP_CHECK(h, hipMalloc(&h->dProjReal2, sizeof(double) * (size_t) h->OB * N * N));
  HIP_CHECK(h, hipMalloc(&h->dTempDen2, sizeof(double) * h->OB));
  HIP_CHECK(h, hipMalloc(&h->dRowSpec2, sizeof(double2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dSpecRef2, sizeof(float2) * (size_t) h->OB * M));
  HIP_CHECK(h, hipMalloc(&h->dScratch2, sizeof(float) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dConv2, sizeof(float2) * (size_t) h->maxOC * M));
  HIP_CHECK(h, hipMalloc(&h->dParams2, sizeof(bioem_hip_param5) * h->maxOC));
  for (int i = 0; i < 2; i++)
  {
    HIP_CHECK(h, hipEventCreateWithFlags(&h->prepDone[i], hipEventDisableTiming));
    HIP_CHECK(h, hipEventCreateWithFlags(&h->cmpDone[i], hipEventDisableTiming));
  }

  std::vector<float2> tw(N + 1);
  std::vector<double2> twd(N);
  for (int k = 0; k <= N; k++)
  {
    const double ang = 2.0 * M_PI * (double) (k % N) / (double) N;
    tw[k] = make_float2((float) cos(ang), (float) sin(ang));
    if (k < N)
      twd[k] = make_double2(cos(ang), sin(ang));
  }
  HIP_CHECK(h, hipMemcpy(h->dTw, tw.data(), sizeof(float2) * (N + 1), hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dTwD, twd.data(), sizeof(double2) * N, hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dDisp, h->disp.data(), sizeof(int) * h->nd, hipMemcpyHostToDevice));
  {
    // log table: bin i of the mantissa interval [1,2): c = 1/centre, entry {c, -log(c)}
    std::vector<double2> lt(64);
    for (int i = 0; i < 64; i++)
    {
      const double c = 1.0 / (1.0 + ((double) i + 0.5) / 64.0);
      lt[i] = make_double2(c, -log(c));
    }
    HIP_CHECK(h, hipMemcpy(h->dLtab, lt.data(), sizeof(double2) * 64, hipMemcpyHostToDevice));
  }
  if (h->fast)
  {
    const int NW = 2 * h->winD + 1;
    std::vector<float2> twk((size_t) h->N1 * NW);
    for (int k1 = 0; k1 < h->N1; k1++)
      for (int d = -h->winD; d <= h->winD; d++)
      {
        const double ang = 2.0 * M_PI * (double) ((((long long) d * h->gs * k1) % N + N) % N) / (double) N;
        twk[(size_t) k1 * NW + d + h->winD] = make_float2((float) cos(ang), (float) sin(ang));
      }
    HIP_CHECK(h, hipMalloc(&h->dTwk, sizeof(float2) * twk.size()));
    HIP_CHECK(h, hipMemcpy(h->dTwk, twk.data(), sizeof(float2) * twk.size(), hipMemcpyHostToDevice));
  }
  HIP_CHECK(h, hipEventCreateWithFlags(&h->slotEvent[0], hipEventDisableTiming));
  HIP_CHECK(h, hipEventCreateWithFlags(&h->slotEvent[1], hipEventDisableTiming));
  return 0;
}

int bioem_hip_destroy(bioem_hip_handle h)
{
  if (!h)
    return 0;
  hipSetDevice(h->device);
  if (h->stream)
    hipStreamSynchronize(h->stream);
  drain_events(h);
  for (hipEvent_t e : h->evPool)
    hipEventDestroy(e);
  void *ptrs[] = {h->dRef,     h->dSumRef,  h->dSumsqRef, h->dCTF,     h->dCtfParam, h->dPts,   h->dAngles,
                  h->dTw,      h->dTwD,     h->dDisp,     h->dLtab,    h->dTwk,     h->dProjReal, h->dTempDen,  h->dRowSpec, h->dSpecRef,
                  h->dScratch, h->dConv,    h->dParams,   h->dPartials, h->dProb,     h->dStage,
                  h->dProjReal2, h->dTempDen2, h->dRowSpec2, h->dSpecRef2, h->dScratch2, h->dConv2, h->dParams2,
                  h->dTnyq};
  for (void *p : ptrs)
    if (p)
      hipFree(p);
  if (h->hStage)
    hipHostFree(h->hStage);
  if (h->hStageP)
    hipHostFree(h->hStageP);
  for (int i = 0; i < 2; i++)
  {
    if (h->slotEvent[i])
      hipEventDestroy(h->slotEvent[i]);
    if (h->prepDone[i])
      hipEventDestroy(h->prepDone[i]);
    if (h->cmpDone[i])
      hipEventDestroy(h->cmpDone[i]);
  }
  if (h->prepStream)
  {
    hipStreamSynchronize(h->prepStream);
    hipStreamDestroy(h->prepStream);
  }
  if (h->stream)
    hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int bioem_hip_upload_particles(bioem_hip_handle h, const float *refFFT, const float *sum, const float *sumsq)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  HIP_CHECK(h, hipMemcpyAsync(h->dSumRef, sum, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(h->dSumsqRef, sumsq, sizeof(float) * h->nMaps, hipMemcpyHostToDevice, h->stream));
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    HIP_CHECK(h, hipMemcpyAsync(h->dSpecRef, refFFT + 2 * M * (size_t) b, sizeof(float2) * M * n,
                                hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + M * (size_t) b, n, h->N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  return 0;
}

int bioem_hip_upload_particle_maps(bioem_hip_handle h, const float *maps)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  const int N = h->N;
  float *dMaps = nullptr;
  HIP_CHECK(h, hipMalloc(&dMaps, sizeof(float) * (size_t) h->chunkB * N * N));
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    HIP_CHECK(h, hipMemcpyAsync(dMaps, maps + (size_t) b * N * N, sizeof(float) * (size_t) n * N * N,
                                hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_map_sums, dim3(n), dim3(256), 0, h->stream, dMaps, N * N, h->dSumRef + b, h->dSumsqRef + b);
    HIP_CHECK(h, hipGetLastError());
    if (run_r2c(h, batch_buf(h, 0), h->stream, nullptr, dMaps, n))
    {
      hipFree(dMaps);
      return 1;
    }
    hipLaunchKernelGGL(k_reorder, dim3(1024), dim3(256), 0, h->stream, h->dSpecRef, h->dRef + M * (size_t) b, n, N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  hipFree(dMaps);
  return 0;
}

int bioem_hip_upload_ctf(bioem_hip_handle h, const float *refCTF, const float *ctfParam3)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipMemcpy(h->dCTF, refCTF, sizeof(float2) * (size_t) h->M * h->nCTF, hipMemcpyHostToDevice));
  HIP_CHECK(h, hipMemcpy(h->dCtfParam, ctfParam3, sizeof(float) * 3 * h->nCTF, hipMemcpyHostToDevice));
  return 0;
}

int bioem_hip_upload_model(bioem_hip_handle h, const bioem_hip_model_point *pts, int nPts, float NormDen,
                           float pixelSize, int shiftX, int shiftY)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (h->dPts)
    hipFree(h->dPts);
  h->dPts = nullptr;
  HIP_CHECK(h, hipMalloc(&h->dPts, sizeof(bioem_hip_model_point) * (size_t) nPts));
  HIP_CHECK(h, hipMemcpy(h->dPts, pts, sizeof(bioem_hip_model_point) * (size_t) nPts, hipMemcpyHostToDevice));
  h->nPts = nPts;
  h->NormDen = NormDen;
  h->pixelSize = pixelSize;
  h->shiftX = shiftX;
  h->shiftY = shiftY;
  return 0;
}

int bioem_hip_upload_orientations(bioem_hip_handle h, const float *angles4, int n, int isQuat)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (n > h->nAngles)
  {
    h->err = "more orientations than the handle was created for";
    return 2;
  }
  HIP_CHECK(h, hipMemcpy(h->dAngles, angles4, sizeof(float4) * (size_t) n, hipMemcpyHostToDevice));
  h->nAnglesUp = n;
  h->isQuat = isQuat;
  return 0;
}

void *bioem_hip_host_alloc(size_t size)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess)
    return nullptr;
  return p;
}

void bioem_hip_host_free(void *ptr)
{
  if (ptr)
    hipHostFree(ptr);
}

int bioem_hip_start_run(bioem_hip_handle h, const void *pProb_host)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipMemcpyAsync(h->dProb, pProb_host, h->probBytes, hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_compare(bioem_hip_handle h, int iPipeline, int iOrient, int iConvStart, int maxParallelConv,
                      int nTotParallelConv, const float *conv_mapsFFT, const bioem_hip_param5 *comp_params)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  if (maxParallelConv < 1 || maxParallelConv > nTotParallelConv || maxParallelConv > h->maxOC)
  {
    h->err = "bioem_hip_compare: maxParallelConv out of range";
    return 2;
  }
  if (h->stageConv < nTotParallelConv)
  {
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
    if (h->hStage)
      hipHostFree(h->hStage);
    if (h->hStageP)
      hipHostFree(h->hStageP);
    if (h->dStage)
      hipFree(h->dStage);
    HIP_CHECK(h, hipHostMalloc(&h->hStage, sizeof(float2) * M * 2 * nTotParallelConv, hipHostMallocDefault));
    HIP_CHECK(h, hipHostMalloc(&h->hStageP, sizeof(bioem_hip_param5) * 2 * nTotParallelConv, hipHostMallocDefault));
    HIP_CHECK(h, hipMalloc(&h->dStage, sizeof(float2) * M * nTotParallelConv));
    h->stageConv = nTotParallelConv;
  }
  const int par = iPipeline & 1;
  const int k = par * nTotParallelConv; // bioem.cpp:1388
  if (h->slotPending[par])
  {
    HIP_CHECK(h, hipEventSynchronize(h->slotEvent[par])); // bioem_cuda.cu:539
    h->slotPending[par] = false;
  }
  memcpy(h->hStage + M * k, conv_mapsFFT + 2 * M * k, sizeof(float2) * M * maxParallelConv);
  memcpy(h->hStageP + k, comp_params + k, sizeof(bioem_hip_param5) * maxParallelConv);
  HIP_CHECK(h, hipMemcpyAsync(h->dStage, h->hStage + M * k, sizeof(float2) * M * maxParallelConv,
                              hipMemcpyHostToDevice, h->stream));
  HIP_CHECK(h, hipMemcpyAsync(h->dParams, h->hStageP + k, sizeof(bioem_hip_param5) * maxParallelConv,
                              hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_reorder, dim3(256), dim3(256), 0, h->stream, h->dStage, h->dConv, maxParallelConv, h->N, h->H,
                     h->fast, h->N1);
  HIP_CHECK(h, hipGetLastError());
  if (launch_compare_fold(h, batch_buf(h, 0), maxParallelConv, iOrient, iConvStart, maxParallelConv))
    return 1;
  HIP_CHECK(h, hipEventRecord(h->slotEvent[par], h->stream));
  h->slotPending[par] = true;
  return 0;
}

int bioem_hip_project_convolve_compare(bioem_hip_handle h, int iOrientBegin, int iOrientEnd)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (!h->dPts || iOrientBegin < 0 || iOrientEnd > h->nAnglesUp || iOrientBegin > iOrientEnd)
  {
    h->err = "project_convolve_compare: model/orientations not uploaded or range invalid";
    return 2;
  }
  // two-slot pipeline: projection + convolution of batch b+1 run on prepStream while batch b is compared
  const int nb = (iOrientEnd - iOrientBegin + h->OB - 1) / h->OB;
  auto prep = [&](int b) -> int {
    const int slot = b & 1;
    const int o0 = iOrientBegin + b * h->OB;
    const int nO = std::min(h->OB, iOrientEnd - o0);
    const BatchBuf bb = batch_buf(h, slot);
    if (h->cmpPending[slot])
    {
      HIP_CHECK(h, hipStreamWaitEvent(h->prepStream, h->cmpDone[slot], 0));
      h->cmpPending[slot] = false;
    }
    if (project_batch(h, bb, h->prepStream, o0, nO))
      return 1;
    if (convolve_batch(h, bb, h->prepStream, nO))
      return 1;
    HIP_CHECK(h, hipEventRecord(h->prepDone[slot], h->prepStream));
    return 0;
  };
  // anything still queued on the main stream that uses slot 0/1 buffers (debug hooks, compat entry) goes first
  HIP_CHECK(h, hipEventRecord(h->cmpDone[0], h->stream));
  HIP_CHECK(h, hipEventRecord(h->cmpDone[1], h->stream));
  h->cmpPending[0] = h->cmpPending[1] = true;
  if (nb > 0 && prep(0))
    return 1;
  for (int b = 0; b < nb; b++)
  {
    const int slot = b & 1;
    const int o0 = iOrientBegin + b * h->OB;
    const int nO = std::min(h->OB, iOrientEnd - o0);
    if (b + 1 < nb && prep(b + 1))
      return 1;
    HIP_CHECK(h, hipStreamWaitEvent(h->stream, h->prepDone[slot], 0));
    if (launch_compare_fold(h, batch_buf(h, slot), nO * h->nCTF, o0, 0, h->nCTF))
      return 1;
    HIP_CHECK(h, hipEventRecord(h->cmpDone[slot], h->stream));
    h->cmpPending[slot] = true;
  }
  // later main-stream work (finish_run, debug hooks) must also see prepStream drained: it is, through prepDone
  return 0;
}

int bioem_hip_finish_run(bioem_hip_handle h, void *pProb_host)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipMemcpyAsync(pProb_host, h->dProb, h->probBytes, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  h->slotPending[0] = h->slotPending[1] = false;
  drain_events(h);
  return 0;
}

int bioem_hip_synchronize(bioem_hip_handle h)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_merge_host(int nShards, int nMaps, int nAngles, int writeAngles, const void *const *shards, void *out)
{
  if (nShards < 1)
    return 2;
  bioem_hip_prob_map *om = reinterpret_cast<bioem_hip_prob_map *>(out);
  bioem_hip_prob_angle *oa = reinterpret_cast<bioem_hip_prob_angle *>(om + nMaps);
  for (int i = 0; i < nMaps; i++)
  {
    int who = 0;
    double cmax = reinterpret_cast<const bioem_hip_prob_map *>(shards[0])[i].Constoadd;
    for (int s = 1; s < nShards; s++)
    {
      const double c = reinterpret_cast<const bioem_hip_prob_map *>(shards[s])[i].Constoadd;
      if (c > cmax)
      {
        cmax = c;
        who = s;
      }
    }
    double tot = 0.;
    for (int s = 0; s < nShards; s++)
    {
      const bioem_hip_prob_map &m = reinterpret_cast<const bioem_hip_prob_map *>(shards[s])[i];
      tot += m.Total * exp(m.Constoadd - cmax);
    }
    om[i] = reinterpret_cast<const bioem_hip_prob_map *>(shards[who])[i];
    om[i].Total = tot;
    om[i].Constoadd = cmax;
  }
  if (writeAngles)
  {
    const size_t cnt = (size_t) nMaps * nAngles;
    for (size_t e = 0; e < cnt; e++)
    {
      double cmax = MIN_PROB;
      for (int s = 0; s < nShards; s++)
      {
        const bioem_hip_prob_angle *a =
            reinterpret_cast<const bioem_hip_prob_angle *>(reinterpret_cast<const bioem_hip_prob_map *>(shards[s]) + nMaps);
        if (a[e].ConstAngle > cmax)
          cmax = a[e].ConstAngle;
      }
      double tot = 0.;
      for (int s = 0; s < nShards; s++)
      {
        const bioem_hip_prob_angle *a =
            reinterpret_cast<const bioem_hip_prob_angle *>(reinterpret_cast<const bioem_hip_prob_map *>(shards[s]) + nMaps);
        tot += a[e].forAngles * exp(a[e].ConstAngle - cmax);
      }
      oa[e].forAngles = tot;
      oa[e].ConstAngle = cmax;
    }
  }
  return 0;
}

int bioem_hip_debug_projection(bioem_hip_handle h, int iOrient, float *spec_out)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (project_batch(h, batch_buf(h, 0), h->stream, iOrient, 1))
    return 1;
  HIP_CHECK(h, hipMemcpyAsync(spec_out, h->dSpecRef, sizeof(float2) * (size_t) h->M, hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int bioem_hip_debug_convolution(bioem_hip_handle h, int iOrient, int iConv, float *spec_out, float *sumC,
                                float *sumsquareC)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  if (project_batch(h, batch_buf(h, 0), h->stream, iOrient, 1))
    return 1;
  if (convolve_batch(h, batch_buf(h, 0), h->stream, 1))
    return 1;
  const size_t M = (size_t) h->M;
  float2 *tmp = h->dSpecRef + M; // chunkB >= 32 slots; slot 0 holds the projection spectrum
  hipLaunchKernelGGL(k_unreorder, dim3(256), dim3(256), 0, h->stream, h->dConv + M * (size_t) iConv, tmp, 1, h->N,
                     h->H, h->fast, h->N1);
  HIP_CHECK(h, hipGetLastError());
  HIP_CHECK(h, hipMemcpyAsync(spec_out, tmp, sizeof(float2) * M, hipMemcpyDeviceToHost, h->stream));
  bioem_hip_param5 q;
  HIP_CHECK(h, hipMemcpyAsync(&q, h->dParams + iConv, sizeof(q), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  *sumC = q.sumC;
  *sumsquareC = q.sumsquareC;
  return 0;
}

int bioem_hip_debug_particles(bioem_hip_handle h, float *refFFT_out, float *sum_out, float *sumsq_out)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  const size_t M = (size_t) h->M;
  for (int b = 0; b < h->nMaps; b += h->chunkB)
  {
    const int n = std::min(h->chunkB, h->nMaps - b);
    hipLaunchKernelGGL(k_unreorder, dim3(1024), dim3(256), 0, h->stream, h->dRef + M * (size_t) b, h->dSpecRef, n, h->N,
                       h->H, h->fast, h->N1);
    HIP_CHECK(h, hipGetLastError());
    HIP_CHECK(h, hipMemcpyAsync(refFFT_out + 2 * M * (size_t) b, h->dSpecRef, sizeof(float2) * M * n,
                                hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(h, hipStreamSynchronize(h->stream));
  }
  HIP_CHECK(h, hipMemcpy(sum_out, h->dSumRef, sizeof(float) * h->nMaps, hipMemcpyDeviceToHost));
  HIP_CHECK(h, hipMemcpy(sumsq_out, h->dSumsqRef, sizeof(float) * h->nMaps, hipMemcpyDeviceToHost));
  return 0;
}

int bioem_hip_kernel_stats(bioem_hip_handle h, double *compare_ms, long long *launches, long long *comparisons)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  drain_events(h);
  if (compare_ms)
    *compare_ms = h->compareMs;
  if (launches)
    *launches = h->launches;
  if (comparisons)
    *comparisons = h->comparisons;
  return 0;
}

int bioem_hip_reset_kernel_stats(bioem_hip_handle h)
{
  HIP_CHECK(h, hipSetDevice(h->device));
  HIP_CHECK(h, hipStreamSynchronize(h->stream));
  drain_events(h);
  h->compareMs = 0;
  h->launches = 0;
  h->comparisons = 0;
  return 0;
}

int bioem_hip_uses_fast_path(bioem_hip_handle h) { return h && h->fast ? 1 : 0; }

int bioem_hip_r2c(int device, int N, int nImg, const float *in, float *out)
{
  if (N < 1 || nImg < 1 || hipSetDevice(device) != hipSuccess)
    return 1;
  const int H = N / 2 + 1;
  const size_t M = (size_t) N * H;
  std::vector<double2> twd(N);
  for (int k = 0; k < N; k++)
  {
    const double ang = 2.0 * M_PI * (double) k / (double) N;
    twd[k] = make_double2(cos(ang), sin(ang));
  }
  double2 *dTw = nullptr, *dRow = nullptr;
  float *dIn = nullptr;
  float2 *dOut = nullptr;
  int rc = 1;
  if (hipMalloc(&dTw, sizeof(double2) * N) == hipSuccess && hipMalloc(&dRow, sizeof(double2) * M * nImg) == hipSuccess &&
      hipMalloc(&dIn, sizeof(float) * (size_t) N * N * nImg) == hipSuccess &&
      hipMalloc(&dOut, sizeof(float2) * M * nImg) == hipSuccess &&
      hipMemcpy(dTw, twd.data(), sizeof(double2) * N, hipMemcpyHostToDevice) == hipSuccess &&
      hipMemcpy(dIn, in, sizeof(float) * (size_t) N * N * nImg, hipMemcpyHostToDevice) == hipSuccess)
  {
    int A, B;
    dft_split(N, A, B);
    hipLaunchKernelGGL(k_dft_rows, dim3(N, nImg), dim3(128), sizeof(double) * (3 * N + 2), 0, nullptr, dIn, nullptr,
                       1.f, N, H, A, B, dTw, dRow);
    hipLaunchKernelGGL(k_dft_cols, dim3(H, nImg), dim3(256), sizeof(double2) * 2 * N, 0, dRow, N, H, A, B, dTw, dOut);
    if (hipGetLastError() == hipSuccess &&
        hipMemcpy(out, dOut, sizeof(float2) * M * nImg, hipMemcpyDeviceToHost) == hipSuccess)
      rc = 0;
  }
  hipFree(dTw);
  hipFree(dRow);
  hipFree(dIn);
  hipFree(dOut);
  return rc;
}

} // extern "C"
