// compare_direct.hpp -- BASELINE config 4: the cross-correlation WITHOUT a transform, as a sliding window in real space
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
//
// The reference always correlates through the FFT (doc/index.rst:1658-1663, bioem.cpp:1435-1459); the values it feeds
// calc_logpro are
//     cc[dx][dy] = sum_{x,y} conv[(x + dx) % N][(y + dy) % N] * img[x][y],   conv = c2r(C) / N^2, img = the particle,
// and this file evaluates that sum directly (BIOEM_CC_DIRECT=1), for comparison with the transform path:
//   k_c2r_cols / k_c2r_rows   the conv spectrum of every (orientation, CTF) row back to real space once per batch,
//                             unnormalised (so that posterior_batch divides by N^2 exactly as on the transform path):
//                             FFTW's rdft2 c2r convention, exact DFT in double (inverse along x for the H stored
//                             columns, then per row the half-complex inverse along y that ignores Im of column 0 and N/2)
//   k_compare_direct<NDXW, NWV>  one block of NWV waves per (conv map, 32 particles): the conv map sits in LDS, the
//                             particles' rows stream through it.  For a window row offset dx the 2-D sum is a matrix product per image
//                             row x:  out[dy][p] += A_x+dx[dy][y] * B_x[y][p],  A = the Toeplitz matrix of the conv row
//                             (read from LDS with the column offset of the lane), B = row x of 32 particles.
//                             v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulation): M = 32 window columns,
//                             N = 32 particles, K = 2 pixels per instruction.  Each wave owns NDXW window rows (its
//                             accumulators), all share B.  Then the log posterior of every displacement
//                             (posterior_batch) and the block's log-sum-exp partial per particle.
// Cost: N^2 (2 maxD / gs + 1) * 32 / 2 matrix instructions per 32 comparisons = 21 x 16 384 FMAs per comparison at 128^2
// +-10 px with the 21 -> 32 padding of M on top: the matrix pipe bounds it at ~7 M comparisons/s, a twentieth of the
// transform path.  Images up to 160 pixels (the conv map must fit LDS), windows up to 24 offsets per axis (eight waves
// of three window rows each: 3.5 M comparisons/s at 128^2 +-10 px, 56 % of the matrix-pipe rate).
#ifndef BIOEM_COMPARE_DIRECT_HPP
#define BIOEM_COMPARE_DIRECT_HPP

namespace
{

constexpr int kDirectMaxN = 160;

// Z[oc][x][ky] = sum_kx C[oc][kx][ky] e^{+2 pi i kx x / N}; block = (ky, oc), thread = x
__global__ void k_c2r_cols(const float2 *__restrict__ conv, int N, int H, int fast, int N1,
                           const double2 *__restrict__ twD, double2 *__restrict__ Z)
{
  extern __shared__ double2 colS[]; // N
  const int ky = blockIdx.x, oc = blockIdx.y;
  const size_t M = (size_t) N * H;
  for (int kx = threadIdx.x; kx < N; kx += blockDim.x)
  {
    const float2 c = conv[(size_t) oc * M + layout_index(fast, N1, H, kx, ky)];
    colS[kx] = make_double2((double) c.x, (double) c.y);
  }
  __syncthreads();
  for (int x = threadIdx.x; x < N; x += blockDim.x)
  {
    double zr = 0., zi = 0.;
    int idx = 0;
    for (int kx = 0; kx < N; kx++)
    {
      const double2 w = twD[idx], c = colS[kx];
      zr = fma(c.x, w.x, zr);
      zr = fma(-c.y, w.y, zr);
      zi = fma(c.x, w.y, zi);
      zi = fma(c.y, w.x, zi);
      idx += x;
      idx = idx >= N ? idx - N : idx;
    }
    Z[((size_t) oc * N + x) * H + ky] = make_double2(zr, zi);
  }
}

// real[oc][x][y] = Re Z[x][0] + (-1)^y Re Z[x][N/2] (even N)
//                   + 2 sum_{0 < ky < N/2 (odd N: <= (N-1)/2)} Re(Z[x][ky] e^{+2 pi i ky y/N});  block = (x, oc), thread = y
__global__ void k_c2r_rows(const double2 *__restrict__ Z, int N, int H, const double2 *__restrict__ twD,
                           float *__restrict__ real)
{
  extern __shared__ double2 rowS[]; // H
  const int x = blockIdx.x, oc = blockIdx.y;
  for (int ky = threadIdx.x; ky < H; ky += blockDim.x)
    rowS[ky] = Z[((size_t) oc * N + x) * H + ky];
  __syncthreads();
  const bool even = (N & 1) == 0;
  const int kend = even ? H - 1 : H; // interior columns 1 .. kend-1
  for (int y = threadIdx.x; y < N; y += blockDim.x)
  {
    double s = rowS[0].x;
    if (even)
      s += (y & 1) ? -rowS[H - 1].x : rowS[H - 1].x;
    double t = 0.;
    int idx = y; // ky * y mod N at ky = 1
    for (int ky = 1; ky < kend; ky++)
    {
      const double2 w = twD[idx], z = rowS[ky];
      t = fma(z.x, w.x, t);
      t = fma(-z.y, w.y, t);
      idx += y;
      idx = idx >= N ? idx - N : idx;
    }
    real[((size_t) oc * N + x) * N + y] = (float) (s + 2. * t);
  }
}

// merge of two log-sum-exp partials (the rule of lsef_wave_reduce)
__device__ __forceinline__ void lsef_merge(LseF &L, float m2, double s2, int id2, float v2)
{
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

inline size_t direct_lds_bytes(int N)
{
  return sizeof(float) * ((size_t) N * N + 2 * 32 * (size_t) (N + 1)) + 64 * sizeof(double2) + 32 * sizeof(int) +
         8 * 32 * (sizeof(double) + 3 * sizeof(float)) + 64;
}

template <int NDXW, int NWV>
__global__ __launch_bounds__(64 * NWV, NWV / 4) void
k_compare_direct(const CompareArgs a, const float *__restrict__ convReal, const float *__restrict__ maps)
{
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N;
  const int BS = N + 1; // row stride of the particle tile: the 32 lanes of a read hit 32 banks
  float *convS = reinterpret_cast<float *>(smem);                     // [N][N]
  float *Bt = convS + (size_t) N * N;                                 // [2][32][BS]
  const int padF = (4 - ((2 * 32 * BS + N * N) & 3)) & 3; // floats up to the next 16-byte boundary
  double2 *ltab = reinterpret_cast<double2 *>(Bt + 2 * 32 * BS + padF); // 64
  int *rankW = reinterpret_cast<int *>(ltab + 64);                    // 32
  double *mS = reinterpret_cast<double *>(rankW + 32);                // [NWV][32] partial sums
  float *mM = reinterpret_cast<float *>(mS + NWV * 32);               // [NWV][32] maxima
  int *mI = reinterpret_cast<int *>(mM + NWV * 32);                   // [NWV][32] ids
  float *mV = reinterpret_cast<float *>(mI + NWV * 32);               // [NWV][32] values
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int j = lane & 31, kh = lane >> 5;
  const int nPG = (a.nMaps + 31) >> 5;
  const int oc = blockIdx.x / nPG, p0 = (blockIdx.x - oc * nPG) * 32;
  const int gs = a.gs, mD = a.maxD / gs, nd = a.nd;

  const float *cr = convReal + (size_t) oc * N * N;
  for (int e = threadIdx.x; e < N * N; e += blockDim.x)
    convS[e] = cr[e];
  if (threadIdx.x < 32)
    rankW[threadIdx.x] = -1;
  for (int t = threadIdx.x; t < 64; t += blockDim.x)
    ltab[t] = a.ltab[t];
  __syncthreads();
  for (int t = threadIdx.x; t < nd; t += blockDim.x)
  {
    const int m = a.disp[t] / gs + mD;
    if (m >= 0 && m < 32)
      rankW[m] = t;
  }
  // row x of the 32 particles -> registers (fetch) -> Bt[buf][particle][y] (stash): the loads of row x + 1 fly while row
  // x is multiplied
  constexpr int PRE = (32 * kDirectMaxN + 64 * NWV - 1) / (64 * NWV);
  float pre[PRE];
  auto fetch = [&](int x) {
#pragma unroll
    for (int u = 0; u < PRE; u++)
    {
      const int e = (int) threadIdx.x + u * 64 * NWV;
      const int jj = e / N, y = e - jj * N;
      const int p = p0 + jj;
      pre[u] = (jj < 32 && p < a.nMaps) ? maps[((size_t) p * N + x) * N + y] : 0.f;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < PRE; u++)
    {
      const int e = (int) threadIdx.x + u * 64 * NWV;
      const int jj = e / N, y = e - jj * N;
      if (jj < 32)
        Bt[(buf * 32 + jj) * BS + y] = pre[u];
    }
  };
  fetch(0);
  stash(0);

  floatx16 D[NDXW];
#pragma unroll
  for (int q = 0; q < NDXW; q++)
#pragma unroll
    for (int i = 0; i < 16; i++)
      D[q][i] = 0.f;
  // the A operand of this lane: window column m = j (offset (j - mD) gs; rows beyond the window read valid memory and
  // are masked later), pixel kh of the instruction's pair
  int baseA = (kh + (j - mD) * gs) % N;
  baseA = baseA < 0 ? baseA + N : baseA;
  const int steps = (N + 1) >> 1; // pixel pairs per row
  constexpr int KC = 32; // pairs whose B operands sit in registers at a time

  for (int x = 0; x < N; x++)
  {
    __syncthreads(); // Bt[x & 1] is complete, Bt[~x & 1] is free
    if (x + 1 < N)
      fetch(x + 1);
    const float *Bx = Bt + ((x & 1) * 32 + j) * BS;
    for (int s0 = 0; s0 < steps; s0 += KC)
    {
      float b[KC];
#pragma unroll
      for (int s = 0; s < KC; s++)
      {
        const int y = 2 * (s0 + s) + kh;
        b[s] = y < N ? Bx[min(y, N - 1)] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < NDXW; q++)
      {
        const int mx = wave * NDXW + q; // window row of this wave's accumulator q
        if (mx >= nd)
          continue;
        int xr = (x + (mx - mD) * gs) % N;
        xr = xr < 0 ? xr + N : xr;
        const float *Ar = convS + xr * N;
        int idx = baseA + 2 * s0;
        idx = idx >= N ? idx - N : idx;
        idx = idx >= N ? idx - N : idx;
#pragma unroll
        for (int s = 0; s < KC; s++)
        {
          const float av = Ar[idx];
          D[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[s], D[q], 0, 0, 0);
          idx += 2;
          idx = idx >= N ? idx - N : idx;
        }
      }
    }
    if (x + 1 < N)
      stash((x + 1) & 1);
  }

  // D[q][i]: window row mx = wave NDXW + q, window column m = 8 (i / 4) + 4 kh + i % 4, particle p0 + j
  const int p = p0 + j;
  const bool pvalid = p < a.nMaps;
  const int pc = pvalid ? p : a.nMaps - 1;
  const PostW pw = post_consts(a.pd.Ntotpi, N, a.params[oc], a.sumRef[pc], a.sumsqRef[pc], a.postc[oc].x, a.postc[oc].y);
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
#pragma unroll
  for (int q = 0; q < NDXW; q++)
  {
    const int mx = wave * NDXW + q;
    const int rkx = rankW[mx < 32 ? mx : 0];
    const bool okx = pvalid && mx < nd && mx < 32 && rkx >= 0;
#pragma unroll
    for (int bb = 0; bb < 2; bb++)
    {
      float accv[8];
      int idv[8];
      bool okv[8];
#pragma unroll
      for (int t = 0; t < 8; t++)
      {
        const int i = 8 * bb + t;
        const int m = 8 * (i / 4) + 4 * kh + (i % 4);
        const int rky = rankW[m];
        accv[t] = D[q][i];
        okv[t] = okx && m < nd && rky >= 0;
        idv[t] = rkx * nd + rky;
      }
      posterior_batch<8>(L, accv, idv, okv, pw, ltab, a.algo);
    }
  }
  // the two lanes of a particle, then the four waves (fixed order)
  lsef_merge(L, __shfl_xor(L.m, 32), __shfl_xor(L.s, 32), __shfl_xor(L.id, 32), __shfl_xor(L.val, 32));
  if (lane < 32)
  {
    mS[wave * 32 + j] = L.s;
    mM[wave * 32 + j] = L.m;
    mI[wave * 32 + j] = L.id;
    mV[wave * 32 + j] = L.val;
  }
  __syncthreads();
  if (wave == 0 && lane < 32 && pvalid)
  {
    for (int w = 1; w < NWV; w++)
      lsef_merge(L, mM[w * 32 + j], mS[w * 32 + j], mI[w * 32 + j], mV[w * 32 + j]);
    Partial r;
    r.sumExp = L.s;
    r.best = L.m;
    r.id = L.id;
    r.value = L.val;
    r.pad = 0;
    a.partials[(size_t) p * a.ldPart + oc] = r;
  }
}

} // namespace

#endif
