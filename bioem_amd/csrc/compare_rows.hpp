// compare_rows.hpp -- comparison kernel for odd image sizes: direct column sums in registers, fast-kernel back half
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_ROWS_HPP
#define BIOEM_COMPARE_ROWS_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// k_compare_rows<WD, GS>: an odd N has no even factor for the paired comparison layout and no register FFT, so the
// column transform is the plain sum  T[m][ky] = sum_kx X[kx][ky] w_N^(kx m GS)  over the reference layout
// [kx][ky] -- i.e. the fast kernel with a "register FFT" of length 1: lane = frequency column, the 2*WD+1 window
// rows accumulate in registers with wave-uniform twiddles (table twk[kx][m], wide scalar loads), operands stream
// through a 4-deep ring of 8-byte buffer loads.  Everything behind the column transform (T exchange through LDS per
// 64-column block, window pass, posterior, partial) is the fast kernel's.  ~3.5x the instructions of the FFT path
// at a similar size, ~3.5x fewer than the generic kernel needs.
// ------------------------------------------------------------------------------------------------
template <int WD, int GS>
__global__ __launch_bounds__(256, (WD <= 10 ? 3 : 2)) void k_compare_rows(const CompareArgs a)
{
  constexpr int NW = 2 * WD + 1;
  constexpr int RD = 4;
  constexpr int NR = (WD <= 5) ? 3 : (WD <= 10) ? 7 : 16;
  constexpr int TS = 66;
  constexpr bool NYQ = false;
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256);
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256 + 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float2 *Tl = Tall + (size_t) wave * NW * TS;

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
  { // block order: see k_compare_fast
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
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.ref + (size_t) p * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.conv + (size_t) oc * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);

  // window lanes (as in k_compare_fast)
  const int nd = a.nd;
  const int G = 64 / nd;
  const int nr = (nd + G - 1) / G;
  const int iy = lane % nd, grp = lane / nd;
  const bool wactive = grp < G;
  const int dy = displ[iy];
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

  const int nblk = (H + 63) / 64;
  const unsigned rowbytes = (unsigned) H * 8u; // reference layout: one float2 per (kx, ky)
  u32x2 rf[RD], rc[RD];
  for (int blk = 0; blk < nblk; blk++)
  {
    const int ky = blk * 64 + lane;
    const int kyc = ky < H ? ky : H - 1;
    const unsigned laneoff = (unsigned) kyc * 8u;
    // the ring is refilled per column block: slot = row mod RD only holds inside a block (N is odd)
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      rf[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrcF, laneoff, (unsigned) min(t, N - 1) * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrcC, laneoff, (unsigned) min(t, N - 1) * rowbytes, 0);
    }
    float Tr[NW], Ti[NW];
#pragma unroll
    for (int d = 0; d < NW; d++)
    {
      Tr[d] = 0.f;
      Ti[d] = 0.f;
    }
    if (ky < H)
      for (int kb = 0; kb < N; kb += RD)
      {
#pragma unroll
        for (int u = 0; u < RD; u++)
        {
          const int kx = kb + u;
          if (kx < N) // wave-uniform
          {
            const float2 f = as_float2(rf[u]);
            const float2 c = as_float2(rc[u]);
            // X = conv * conj(ref)   (bioem.cpp:1452-1455)
            const float xr = fmaf(c.x, f.x, c.y * f.y);
            const float xi = fmaf(c.y, f.x, -(c.x * f.y));
            const int tn = min(kx + RD, N - 1); // past the end: re-read the last row (unused)
            rf[u] = __builtin_amdgcn_raw_buffer_load_b64(rsrcF, laneoff, (unsigned) tn * rowbytes, 0);
            rc[u] = __builtin_amdgcn_raw_buffer_load_b64(rsrcC, laneoff, (unsigned) tn * rowbytes, 0);
            const float2 *twk = a.twk + (size_t) kx * NW; // w_N^(kx m GS), m = -WD..WD: wave-uniform
#pragma unroll
            for (int d = 0; d < NW; d++)
            {
              const float2 w = twk[d];
              float tr = Tr[d], ti = Ti[d];
              tr = fmaf(xr, w.x, tr);
              tr = fmaf(-xi, w.y, tr);
              ti = fmaf(xr, w.y, ti);
              ti = fmaf(xi, w.x, ti);
              Tr[d] = tr;
              Ti[d] = ti;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    float wgt = 2.f; // FFTW c2r convention: column 0 (and N/2 for even N) enters once, others twice
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H)
      wgt = 0.f;
    WAVE_OR_BLOCK_SYNC(); // previous window reads are done
#pragma unroll
    for (int d = 0; d < NW; d++)
      Tl[d * TS + lane] = make_float2(Tr[d] * wgt, Ti[d] * wgt);
    WAVE_OR_BLOCK_SYNC();
    const int idx0 = (int) (((long long) blk * 64 * step) % N);
    const int npairs = min(32, (H - blk * 64 + 1) >> 1); // columns of this block that exist, in pairs
    if (is_static)
    {
      const int rowoff[NR] = {rowbase};
      window_accumulate<NR, true, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
    }
    else
    {
      int rowoff[NR];
#pragma unroll
      for (int r = 0; r < NR; r++)
        rowoff[r] = row_of(r);
      window_accumulate<NR, false, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
    }
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
    const int ixs = grp * nr + r; // position in the lane-group order; ix = visiting rank of that displacement
    if (r < nr && wactive && ixs < a.ndx && iy < a.ndy)
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
// k_compare_oddfft<WD, R>: odd N with a factor R in {3, 5, 9, 15, 25}: N = N1 * R, kx = N1*k2 + k1 as in the fast
// kernel, but over the REFERENCE layout -- input k2 of the register FFT of step k1 is simply row N1*k2 + k1 -- with
// 8-byte loads and the mixed-radix register FFT (fft_registers.hpp).  No paired layout, so no even factor is needed.
// Unit row stride only.
// ------------------------------------------------------------------------------------------------
template <int WD, int R>
__global__ __launch_bounds__(256, (WD <= 10 ? 3 : 2)) void k_compare_oddfft(const CompareArgs a)
{
  constexpr int GS = 1;
  constexpr int NW = 2 * WD + 1;
  constexpr int RD = (R % 5 == 0) ? 5 : 3; // ring depth: divides R (3, 5, 9, 15, 25)
  constexpr int NR = (WD <= 5) ? 3 : (WD <= 10) ? 7 : 16;
  constexpr int TS = 66;
  constexpr bool NYQ = false;
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *displ = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256);
  float2 *Tall = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 256 + 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float2 *Tl = Tall + (size_t) wave * NW * TS;

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
  { // block order: see k_compare_fast
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
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.ref + (size_t) p * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.conv + (size_t) oc * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);

  // window lanes (as in k_compare_fast)
  const int nd = a.nd;
  const int G = 64 / nd;
  const int nr = (nd + G - 1) / G;
  const int iy = lane % nd, grp = lane / nd;
  const bool wactive = grp < G;
  const int dy = displ[iy];
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

  const int nblk = (H + 63) / 64;
  const unsigned rowbytes = (unsigned) H * 8u; // reference layout: one float2 per (kx, ky)
  u32x2 rf[RD], rc[RD];
  for (int blk = 0; blk < nblk; blk++)
  {
    const int ky = blk * 64 + lane;
    const int kyc = ky < H ? ky : H - 1;
    const unsigned laneoff = (unsigned) kyc * 8u;
    // the ring is refilled per column block (inputs 0..RD-1 of k1 = 0)
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      rf[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrcF, laneoff, (unsigned) (N1 * t) * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b64(rsrcC, laneoff, (unsigned) (N1 * t) * rowbytes, 0);
    }
    float Tr[NW], Ti[NW];
#pragma unroll
    for (int d = 0; d < NW; d++)
    {
      Tr[d] = 0.f;
      Ti[d] = 0.f;
    }
    if (ky < H)
      for (int k1 = 0; k1 < N1; k1++)
      {
        float xr[R], xi[R];
        float2 wk[NW];
        const float2 *twk = a.twk + (size_t) k1 * NW;
#pragma unroll
        for (int d = 0; d < NW; d++)
          wk[d] = twk[d];
#pragma unroll
        for (int k2 = 0; k2 < R; k2++)
        {
          const float2 f = as_float2(rf[k2 % RD]);
          const float2 c = as_float2(rc[k2 % RD]);
          // X = conv * conj(ref)   (bioem.cpp:1452-1455); input k2 of the register FFT sits in row N1*k2 + k1
          xr[DIGITREV<R>.pos[k2]] = fmaf(c.x, f.x, c.y * f.y);
          xi[DIGITREV<R>.pos[k2]] = fmaf(c.y, f.x, -(c.x * f.y));
          // the load RD inputs ahead: same k1 while k2 + RD < R, else the first inputs of the next k1
          int k2n = k2 + RD, k1n = k1;
          if (k2n >= R)
          {
            k2n -= R;
            k1n = min(k1 + 1, N1 - 1); // past the end: re-read (unused)
          }
          const unsigned so = (unsigned) (N1 * k2n + k1n) * rowbytes;
          rf[k2 % RD] = __builtin_amdgcn_raw_buffer_load_b64(rsrcF, laneoff, so, 0);
          rc[k2 % RD] = __builtin_amdgcn_raw_buffer_load_b64(rsrcC, laneoff, so, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        fft_inverse_mixed<R>(xr, xi);
#pragma unroll
        for (int d = -WD; d <= WD; d++)
        {
          const int pos = ((d % R) + R) % R;
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
    float wgt = 2.f; // FFTW c2r convention: column 0 (and N/2 for even N) enters once, others twice
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H)
      wgt = 0.f;
    WAVE_OR_BLOCK_SYNC(); // previous window reads are done
#pragma unroll
    for (int d = 0; d < NW; d++)
      Tl[d * TS + lane] = make_float2(Tr[d] * wgt, Ti[d] * wgt);
    WAVE_OR_BLOCK_SYNC();
    const int idx0 = (int) (((long long) blk * 64 * step) % N);
    const int npairs = min(32, (H - blk * 64 + 1) >> 1); // columns of this block that exist, in pairs
    if (is_static)
    {
      const int rowoff[NR] = {rowbase};
      window_accumulate<NR, true, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
    }
    else
    {
      int rowoff[NR];
#pragma unroll
      for (int r = 0; r < NR; r++)
        rowoff[r] = row_of(r);
      window_accumulate<NR, false, 32, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
    }
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
    const int ixs = grp * nr + r; // position in the lane-group order; ix = visiting rank of that displacement
    if (r < nr && wactive && ixs < a.ndx && iy < a.ndy)
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

} // namespace

#endif
