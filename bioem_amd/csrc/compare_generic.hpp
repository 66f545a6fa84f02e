// compare_generic.hpp -- generic comparison kernel (direct pruned DFT; odd sizes, very wide windows)
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_GENERIC_HPP
#define BIOEM_COMPARE_GENERIC_HPP

namespace
{

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

} // namespace

#endif
