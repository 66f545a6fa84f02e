// window_tiles.hpp -- wide translation windows as tiles of the fast kernel's 21- or 31-row window
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_WINDOW_TILES_HPP
#define BIOEM_WINDOW_TILES_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// A window of more than 31 offsets per axis (the reference's tutorial suggests DISPLACE_CENTER 40 1 for production
// runs) is covered by T x T tiles of the fast kernel's window.  A tile centred at (sx, sy) pixels is the ordinary
// window of the phase-shifted spectrum  conv'[kx][ky] = conv[kx][ky] * exp(+2 pi i (kx sx + ky sy) / N):
//   cc'[dx][dy] = cc[dx + sx][dy + sy]   exactly (circular shift theorem; the FFTW c2r weights are unaffected).
// Every tile runs the unmodified comparison kernel on the shifted conv spectra with a LOCAL sorted displacement
// list; ndx / ndy mask the rows / lanes of the last tiles that stick out of the window.  k_merge_tiles then folds
// the T*T partials of a comparison into one partial whose arg-max id is the GLOBAL visiting rank, so that the fold
// kernels and everything downstream see exactly what one big window would have produced.
// ------------------------------------------------------------------------------------------------
__global__ void k_phase_shift(const float2 *__restrict__ src, float2 *__restrict__ dst, size_t total, int N, int H,
                              int R2, int N1, int sx, int sy, const float2 *__restrict__ tw)
{
  const size_t M = (size_t) N * H;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t) gridDim.x * blockDim.x)
  {
    const size_t e = i % M;
    int ky;
    long long kx;
    if (R2 == 0)
    { // reference layout kx*H + ky (k_compare_rows)
      ky = (int) (e % H);
      kx = (long long) (e / H);
    }
    else
    { // comparison layout: ((k1*R2 + k2p)*H + ky)*2 + (k2 & 1),  kx = N1*k2 + k1
      const int par = (int) (e & 1);
      const size_t q = e >> 1;
      ky = (int) (q % H);
      const int rp = (int) (q / H);
      const int k1 = rp / R2, k2 = 2 * (rp % R2) + par;
      kx = (long long) N1 * k2 + k1;
    }
    const int t = (int) (((kx * sx + (long long) ky * sy) % N + N) % N);
    const float2 w = tw[t], c = src[i];
    dst[i] = make_float2(fmaf(c.x, w.x, -(c.y * w.y)), fmaf(c.x, w.y, c.y * w.x));
  }
}

// one thread per (particle, orientation*CTF): merge the partials of all tiles
__global__ void k_merge_tiles(const Partial *__restrict__ tiles, int nTiles, size_t tileStride, int ldPart, int nOC,
                              int nMaps, int tileT, int tilesPerAxis, const int *__restrict__ tileCenter, int mD,
                              int ndG, const int *__restrict__ rankOfRow, Partial *__restrict__ out)
{
  const long long t = (long long) blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long) nOC * nMaps)
    return;
  const int p = (int) (t / nOC), oc = (int) (t - (long long) p * nOC);
  const int tW = (tileT - 1) / 2;
  double m = -INFINITY, s = 0.;
  int idBest = 0x7fffffff;
  float valBest = 0.f;
  for (int k = 0; k < nTiles; k++)
  {
    const Partial r = tiles[(size_t) k * tileStride + (size_t) p * ldPart + oc];
    if (r.id == 0x7fffffff)
      continue;
    const int ix = r.id / tileT, iy = r.id - ix * tileT;
    const int mx = ix - tW + tileCenter[k / tilesPerAxis], my = iy - tW + tileCenter[k % tilesPerAxis];
    const int gid = rankOfRow[mx + mD] * ndG + rankOfRow[my + mD]; // visiting rank in the reference's order
    const double lp = (double) r.best;
    if (lp > m || (lp == m && gid < idBest))
    {
      s = ((m == -INFINITY) ? 0. : s * exp(m - lp)) + r.sumExp;
      m = lp;
      idBest = gid;
      valBest = r.value;
    }
    else
      s += r.sumExp * exp(lp - m);
  }
  Partial o;
  o.sumExp = s;
  o.best = (float) m;
  o.id = idBest;
  o.value = valBest;
  o.pad = 0;
  out[(size_t) p * ldPart + oc] = o;
}

} // namespace

#endif
