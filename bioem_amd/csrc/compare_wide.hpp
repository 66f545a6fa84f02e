// compare_wide.hpp -- wide-window variant of the fast comparison kernel: one comparison per 2 or 4 waves
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_WIDE_HPP
#define BIOEM_COMPARE_WIDE_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// k_compare_wide<R, GS, WPC>: the tiles of a wide window (window_tiles.hpp) that share an x-tile also share the
// column transforms T[row][ky] -- only the dy offsets of their y-tiles differ.  Here WPC (2 or 4) waves work on ONE
// comparison: the column transforms are split between them (by column block, then by k1 range), the partial T
// blocks are combined in LDS in a fixed wave order (deterministic), and then every wave runs the window pass and the
// posterior of ITS y-tile on the shared T.  Per comparison and x-tile the operand stream and the register FFTs are
// paid once instead of once per y-tile.  21-row tiles, any register-FFT length of the fast kernel, as many
// 64-column blocks as waves per comparison.
// Output: one partial per (x-tile, y-tile) with tile-local ids, exactly what the tile-per-launch path writes, so
// k_merge_tiles is unchanged.
// ------------------------------------------------------------------------------------------------
template <int R, int GS, int WPC, bool NYQ>
__global__ __launch_bounds__(256, 3) void k_compare_wide(const CompareArgs a)
{
  constexpr int WD = 10, NW = 2 * WD + 1, NR = 7, TS = 66;
  constexpr int R2 = R / 2;
  // ring depth: divides R2 (mixed-radix lengths: R2 = 3, 5, 6, 9, 10, 15)
  constexpr int RD = (R2 % 4 == 0) ? 4 : (R2 % 5 == 0) ? 5 : (R2 % 3 == 0) ? 3 : (R2 % 2 == 0) ? 2 : 1;
  constexpr int CPB = 4 / WPC; // comparisons per block
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256);
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256 + 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int cmp = wave / WPC, sub = wave % WPC;

  for (int t = threadIdx.x; t <= N; t += blockDim.x)
    twl[t] = a.tw[t];
  int *dinv = displ + 32;
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

  int p, ocg;
  { // block order as in k_compare_fast, groups of CPB comparisons
    const int ocGroups = (a.nOC + CPB - 1) / CPB;
    const int per = a.pchunk * ocGroups;
    int c = blockIdx.x / per;
    const int nch = (a.nMaps + a.pchunk - 1) / a.pchunk;
    c = min(c, nch - 1);
    const int rem = blockIdx.x - c * per;
    const int pc = min(a.pchunk, a.nMaps - c * a.pchunk);
    ocg = rem / pc;
    p = c * a.pchunk + (rem - ocg * pc);
  }
  const int oc_raw = ocg * CPB + cmp;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const size_t M = (size_t) N * H;
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.ref + (size_t) p * M), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(a.conv + (size_t) oc * M), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);

  // window lanes; this wave's y-tile
  const int nd = a.nd;
  const int G = 64 / nd;
  const int nr = (nd + G - 1) / G;
  const int iy = lane % nd, grp = lane / nd;
  const bool wactive = grp < G;
  const int ty = a.yTile0 + sub;
  const bool tyValid = ty < a.nTiles;
  const int tyc = tyValid ? ty : a.nTiles - 1;
  const int ndy = tyValid ? a.tileValid[tyc] : 0;
  int dy = displ[iy] + GS * a.tileCenter[tyc]; // pixels
  dy %= N;
  const int step = dy < 0 ? dy + N : dy;
  const bool is_static = (nr == NR) && (nd == G * NR) && (nd == 2 * mD + 1);
  float acc[NR];
#pragma unroll
  for (int r = 0; r < NR; r++)
    acc[r] = 0.f;
  auto row_of = [&](int r) -> int {
    int ix = wactive ? grp * nr + r : r;
    if (ix >= nd)
      ix = nd - 1;
    return (displ[ix] / GS + WD) * TS;
  };
  const int rowbase = is_static ? ((wactive ? grp : 0) * NR - mD + WD) * TS : row_of(0);

  // ---- column transforms: this wave's share = k1 range [k1a, k1b) of column block myblk ----
  // nblk <= WPC column blocks (host); NYQ: the Nyquist column comes from k_nyquist_rows (compare_fast.hpp).  If nblk
  // divides WPC, C = WPC / nblk waves share a block by k1 range; otherwise (3 blocks on 4 waves) every wave takes
  // one whole block and the last wave sits the column phase out
  const int nblk = NYQ ? (H - 1) / 64 : (H + 63) / 64;
  const bool split = (WPC % nblk) == 0;
  const int C = split ? WPC / nblk : 1; // waves per column block
  const int myblk = split ? sub / C : sub, part = split ? sub - myblk * C : 0;
  const bool owns = myblk < nblk;
  const int k1a = owns ? part * N1 / C : 0, k1b = owns ? (part + 1) * N1 / C : 0;
  const unsigned rowbytes = (unsigned) H * 16u;
  const int ky = myblk * 64 + lane;
  const int kyc = ky < H ? ky : H - 1;
  const unsigned laneoff = (unsigned) kyc * 16u;
  const int tend = k1b * R2; // one past the last row of this wave
  float Tr[NW], Ti[NW];
#pragma unroll
  for (int d = 0; d < NW; d++)
  {
    Tr[d] = 0.f;
    Ti[d] = 0.f;
  }
  if (k1a < k1b)
  {
    u32x4 rf[RD], rc[RD];
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      const int tn = min(k1a * R2 + t, tend - 1);
      rf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, laneoff, (unsigned) tn * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, laneoff, (unsigned) tn * rowbytes, 0);
    }
    if (ky < H)
      for (int k1 = k1a; k1 < k1b; k1++)
      {
        float xr[R], xi[R];
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
          const int tn = min(k1 * R2 + k2p + RD, tend - 1); // past the end: re-read the last row (unused)
          rf[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, laneoff, (unsigned) tn * rowbytes, 0);
          rc[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, laneoff, (unsigned) tn * rowbytes, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        FFT_RUN(xr, xi);
#pragma unroll
        for (int d = -WD; d <= WD; d++)
        {
          const int pos = FFT_OUT((((d * GS) % R) + R) % R);
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
  }
  // ---- combine the partial T blocks in LDS: the C waves of a column block add in wave order ----
  {
    float wgt = 2.f; // FFTW c2r convention: columns 0 and N/2 enter once, others twice
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H)
      wgt = 0.f;
    float2 *Tb = Tall + (size_t) ((cmp * nblk + myblk) * NW) * TS;
    for (int turn = 0; turn < C; turn++)
    {
      if (part == turn && owns)
      {
#pragma unroll
        for (int d = 0; d < NW; d++)
        {
          float2 v = make_float2(Tr[d] * wgt, Ti[d] * wgt);
          if (turn > 0)
          {
            const float2 o = Tb[d * TS + lane];
            v.x += o.x;
            v.y += o.y;
          }
          Tb[d * TS + lane] = v;
        }
      }
      __syncthreads();
    }
  }
  // ---- window pass of this wave's y-tile over the shared T ----
  for (int blk = 0; blk < nblk; blk++)
  {
    const float2 *Tl = Tall + (size_t) ((cmp * nblk + blk) * NW) * TS;
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
  const double2 pc = a.postc[oc];
  const double t2 = pc.x, prior = pc.y;
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
    const int ixs = grp * nr + r;
    if (r < nr && wactive && ixs < a.ndx && iy < ndy)
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
  if (lane == 0 && oc_valid && tyValid)
  {
    Partial r;
    r.sumExp = L.s;
    r.best = L.m;
    r.id = L.id;
    r.value = L.val;
    r.pad = 0;
    a.partials[(size_t) ty * a.tileStride + (size_t) p * a.ldPart + oc] = r;
  }
}

} // namespace

#endif
