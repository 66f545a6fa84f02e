// compare_generic.hpp -- generic comparison kernel (direct pruned DFT; odd sizes, very wide windows)
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_GENERIC_HPP
#define BIOEM_COMPARE_GENERIC_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// generic comparison kernel: any N, any window (odd N; more than 31 offsets per axis).  Reference layout.
// One wave per comparison.  Column transform T[j][ky] = sum_kx X[kx][ky] w^(kx dx_j) by direct summation for the
// nd displacement rows only: lane = frequency column, rows in chunks of 16 register accumulators, so X is formed
// once per (kx, chunk) and each twiddle (uniform over the wave: one LDS broadcast read) feeds 64 columns.
// The rows go through LDS in groups of a.ts (a multiple of the chunk; all nd where they fit): column transform of a
// group, its posteriors, the next group -- the same work whatever the group size, and no window is too wide for the LDS.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_compare_generic(const CompareArgs a)
{
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int CH = 16; // rows per register chunk
  const int N = a.N, H = a.H;
  const int Hs = (H + 1) & ~1;
  const int nd = a.nd;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);
  const size_t dispBytes = ((size_t) nd * 4 + 255) & ~(size_t) 255; // displacement list, 256-byte granules
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + dispBytes);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int RG = a.ts > 0 && a.ts < nd ? a.ts : nd; // rows per group
  float2 *Tl = Tall + (size_t) wave * RG * Hs;
  for (int t = threadIdx.x; t <= N; t += blockDim.x)
    twl[t] = a.tw[t];
  for (int t = threadIdx.x; t < nd; t += blockDim.x)
    displ[t] = a.disp[t];
  __syncthreads();

  const int p = blockIdx.x % a.nMaps;
  const int ocg = blockIdx.x / a.nMaps;
  const int nWaves = (int) (blockDim.x >> 6); // 4, or fewer when the per-wave T block is large (wide windows)
  const int oc_raw = ocg * nWaves + wave;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const size_t M = (size_t) N * H;
  const float2 *F = a.ref + (size_t) p * M;
  const float2 *C = a.conv + (size_t) oc * M;

  const bioem_hip_param5 q = a.params[oc];
  const double2 pc = a.postc[oc];
  const double t2 = pc.x, prior = pc.y;
  const float sumref = a.sumRef[p], sumsqref = a.sumsqRef[p];
  const float nn = (float) (N * N);
  Lse L;
  lse_init(L);
  for (int g0 = 0; g0 < nd; g0 += RG)
  {
  const int gEnd = min(nd, g0 + RG);
  for (int ky0 = 0; ky0 < H; ky0 += 64)
  {
    const int ky = ky0 + lane;
    const int kyc = ky < H ? ky : H - 1;
    float wgt = 2.f;
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H)
      wgt = 0.f;
    for (int j0 = g0; j0 < gEnd; j0 += CH)
    {
      float tr[CH], ti[CH];
      int step[CH], idx[CH]; // wave-uniform (SGPRs): twiddle index of row j at the current kx
#pragma unroll
      for (int j = 0; j < CH; j++)
      {
        tr[j] = 0.f;
        ti[j] = 0.f;
        const int dx = a.disp[min(j0 + j, nd - 1)]; // uniform scalar load
        step[j] = dx < 0 ? dx + N : dx;
        idx[j] = 0;
      }
      for (int kx = 0; kx < N; kx++)
      {
        const float2 c = C[(size_t) kx * H + kyc], f = F[(size_t) kx * H + kyc];
        // X = conv * conj(ref)   (bioem.cpp:1452-1455)
        const float xr = fmaf(c.x, f.x, c.y * f.y);
        const float xi = fmaf(c.y, f.x, -(c.x * f.y));
#pragma unroll
        for (int j = 0; j < CH; j++)
        {
          const float2 w = twl[idx[j]];
          tr[j] = fmaf(xr, w.x, tr[j]);
          tr[j] = fmaf(-xi, w.y, tr[j]);
          ti[j] = fmaf(xr, w.y, ti[j]);
          ti[j] = fmaf(xi, w.x, ti[j]);
          idx[j] += step[j];
          if (idx[j] >= N)
            idx[j] -= N;
        }
      }
      if (ky < Hs)
      {
#pragma unroll
        for (int j = 0; j < CH; j++)
          if (j0 + j < gEnd)
            Tl[(size_t) (j0 + j - g0) * Hs + ky] = make_float2(tr[j] * wgt, ti[j] * wgt);
      }
    }
  }
  __syncthreads();

  for (int el = lane; el < (gEnd - g0) * nd; el += 64)
  {
    const int e = g0 * nd + el; // visiting rank of the displacement
    const int ix = e / nd, iy = e - ix * nd;
    const int dy = displ[iy];
    const int step = dy < 0 ? dy + N : dy;
    const float2 *row = Tl + (size_t) (ix - g0) * Hs;
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
  __syncthreads(); // (the next group overwrites T)
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

} // namespace

#endif
