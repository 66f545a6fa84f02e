// compare_fastm2.hpp -- 33..47-row translation windows (+-16 ... +-23 px): one wave per comparison, window rows split
// over the two halves of the wave, window pass as 3 x 3 tiles of v_mfma_f32_16x16x4_f32
// Part of libbioem_hip.so; included by kernels_fastm2.hip (and by bioem_hip.hip in experiment builds).
#ifndef BIOEM_COMPARE_FASTM2_HPP
#define BIOEM_COMPARE_FASTM2_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// Round 3 ran windows of more than 31 rows on k_compare_wide2 (four waves share one comparison through LDS slots and
// block barriers): 224^2 +-20 px 24.7 M/s against 42.7 at +-15 px, a 1.7x step for 1.3x the work.  What keeps the
// one-wave-per-comparison kernels from wider windows is registers: k_compare_fast(m) holds every window row of a
// frequency column in one lane (2 x rows accumulators), and 41 rows + a 16-point FFT + the operand ring do not fit the
// 168 registers of three waves per SIMD.
//
// This kernel halves the accumulators: a column pass covers 32 frequency columns, lane l and lane l + 32 work on the
// SAME column.  In a step the low half forms spectrum product and register FFT of k1 = 2 s, the high half of
// k1 = 2 s + 1 -- no lane idles --, then
//     v_permlane32_swap (y[p], y[p + 8])       p = (j - WD) mod 16, j = 0..7        (gfx950; 16 per step; R = 16 shown)
// leaves in register y[p] the k1 = 2 s outputs and in y[p + 8] the k1 = 2 s + 1 outputs -- residue p in the low half,
// residue p + 8 in the high half.  The low half owns the window rows m = 16 g + j, the high half the rows
// m = 16 g + 8 + j (g = 0..2, j = 0..7): rows 8 apart have residues 8 apart, so ONE register serves accumulator (g, j)
// of both halves.  24 complex accumulators per lane instead of 47.
//   T[dx] += w_N^(dx k1) y_k1[dx mod 16]:  the high rows are the low rows + 8, w_N^((dx + 8) k1) = w_N^(dx k1) w_N^(8 k1):
//   every lane multiplies the eight registers that end up in the high half by w_N^(8 k1) of its OWN k1 before the swap
//   (32 instructions), after which the recombination twiddles are those of the low rows for every lane: wave-uniform,
//   wide scalar loads, no vector registers (as in k_compare_fast).
// Window pass (cc[dx][dy] = sum_ky Re T cos - Im T sin) on the matrix cores, exact f32: 48 x 48 outputs = 3 x 3 tiles of
// v_mfma_f32_16x16x4_f32 (K = 2 columns x {re, im}), 36 accumulator registers that never leave the register file; T
// reaches the A-operand planes in LDS 16 columns at a time (6.4 KiB per wave).  One 32 x 32 x 2 tile pair per axis would
// cost 64 accumulators and 1.8x the matrix-pipe time.
// Rows m = 0..46 hold dx = (m - 23) GS for EVERY window of the family (+-16 ... +-23 rows at stride GS = 1..4 pixels): rows
// outside the displacement list
// get zero twiddles and a rank of -1 -- the cost does not depend on the width inside the family (48 rows either way).
// R = 16, or 12 / 10 / 8 where 16 does not divide N (rows 8 / 6 / 5 / 4 apart pair up; 3 / 4 / 5 / 6 groups of R rows,
// 24 or 25 accumulators); an odd N1 = N / R leaves the high half of the last step reading beyond the buffer
// descriptor's range, which returns zeros: its contribution vanishes without a branch.
// ------------------------------------------------------------------------------------------------
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));

constexpr int kFm2WD = 23;        // window rows m = 0..2 WD hold dx = m - WD
constexpr int kFm2Rows = 48;      // three 16-row tiles
constexpr int kFm2PS = 33;        // plane row stride in floats (32 columns + 1: the 16 rows of a tile hit 16 banks)
constexpr int kFm2PlaneRows = 47; // row 47 holds no window row: neither written nor, where it is read as an operand, used
// LDS floats per wave: the two A-operand planes (12.1 KiB); between matrix passes the same space holds the 36 tile
// accumulators of every lane
constexpr int kFm2WaveFloats = 36 * 64 > 2 * kFm2PlaneRows * kFm2PS ? 36 * 64 : 2 * kFm2PlaneRows * kFm2PS;
// Rows OFF apart pair up: OFF = the smallest row distance whose pixel distance OFF GS is R / 2 mod R (8 / 6 / 5 / 4 rows for
// R = 16 / 12 / 10 / 8 at unit stride; 4, 8, 2 rows for R = 16 at stride 2, 3, 4).  The low half of the wave owns the
// rows m = 2 OFF g + j (j < OFF), the high half the rows OFF further; groups and complex accumulators per lane:
__host__ __device__ constexpr int fastm2_off(int R, int GS)
{
  for (int o = 1; o <= R; o++)
    if ((o * GS) % R == R / 2)
      return o;
  return 0;
}
__host__ __device__ constexpr int fastm2_groups(int R, int GS) { return (2 * kFm2WD + 2 * fastm2_off(R, GS)) / (2 * fastm2_off(R, GS)); }
__host__ __device__ constexpr int fastm2_acc(int R, int GS) { return fastm2_groups(R, GS) * fastm2_off(R, GS); }
// the B operand of the matrix pass -- (lane / 16 odd ? sin : cos)(2 pi ky dy / N), ky = 2 K + lane / 32, dy = 16 ct + lane
// % 16 - 23 -- is the same matrix for every comparison: tabulated on the host, [column pass][16 k-steps][3 column
// tiles][64 lanes] floats (index arithmetic on the LDS twiddle table cost 30 vector instructions per k-step and a
// 64-bit modulo per exchange: 19.0 k -> 17.0 k vector instructions per comparison at 224^2)
__host__ __device__ constexpr int Hlim0(int H, bool nyq) { return nyq ? H - 1 : H; }
__host__ __device__ constexpr size_t fastm2_btab_floats(int H, bool nyq) { return (size_t) ((Hlim0(H, nyq) + 31) / 32) * 16 * 3 * 64; }

// recombination twiddles of the low rows, [k1 pair s][accumulator a = OFF g + j] = {w^(dx 2s), w^(dx (2s+1))},
// dx = (2 OFF g + j - WD) GS
typedef const float4 __attribute__((address_space(4))) *const_float4_ptr;

template <int R, bool NYQ, int GS>
__global__ __launch_bounds__(256, 3) void k_compare_fastm2(const CompareArgs a)
{
  static_assert(R == 16 || R == 12 || R == 10 || R == 8, "register FFT of 16, 12, 10 or 8 points");
  constexpr int OFF = fastm2_off(R, GS);    // rows between the halves' rows
  static_assert(OFF > 0, "no row distance pairs up at this stride");
  constexpr int WD = kFm2WD, NWR = 2 * WD + 1;
  constexpr int R2 = R / 2;                 // row pairs of a step
  constexpr int RD = (R2 % 4 == 0) ? 4 : (R2 % 3 == 0) ? 3 : R2; // operand ring depth (divides R2)
  constexpr int NACC = fastm2_acc(R, GS);   // 24 (25 for R = 10)
  constexpr int PS = kFm2PS, PLANE = kFm2PlaneRows * PS;
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1;
  int *rankW = reinterpret_cast<int *>(smem);                // 48 ints (256 B reserved)
  double2 *ltab = reinterpret_cast<double2 *>(smem + 256);     // 64 entries
  float *Pall = reinterpret_cast<float *>(smem + 256 + 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float *Pl = Pall + (size_t) wave * kFm2WaveFloats;
  float *Dpark = Pl + lane;

  if (threadIdx.x < kFm2Rows)
    rankW[threadIdx.x] = -1;
  for (int t = threadIdx.x; t < 64; t += blockDim.x)
    ltab[t] = a.ltab[t];
  __syncthreads();
  for (int t = threadIdx.x; t < a.nd; t += blockDim.x)
  {
    const int m = a.disp[t] / GS + WD;
    if (m >= 0 && m < NWR)
      rankW[m] = t;
  }
  __syncthreads();

  int p, ocg;
  if (!fast_block_pair(a, p, ocg)) // the XCD-aware block order of k_compare_fast
    return;
  const int oc_raw = ocg * 4 + wave;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const int Hp = a.Hp; // row-pair pitch in 16-byte words (H, or H + 15: comparison_pitch in bioem_hip.hip)
  const size_t M = (size_t) N * Hp;
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.ref + (size_t) p * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.conv + (size_t) oc * M)), 0,
                                                       (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float *>(a.btab)), 0,
                                                       (int) (fastm2_btab_floats(a.H, NYQ) * sizeof(float)), 0x00020000);
  const int hh = lane >> 5, c32 = lane & 31;
  // matrix-pass lane constants: tile row / column jn = lane % 16, k index kq = lane / 16 = (column of the pair, re | im)
  const int jn = lane & 15, kq = lane >> 4;
  const float *Arow = Pl + (kq & 1) * PLANE + jn * PS + (kq >> 1); // A[row 16 rt + jn][k]: Arow[rt * 16 * PS + 2 ks]
#pragma unroll
  for (int i = 0; i < 36; i++)
    Dpark[i * 64] = 0.f;

  const int Hlim = NYQ ? H - 1 : H; // columns of the passes (the Nyquist column of 128^2 / 256^2 comes from k_nyquist_rows)
  const int npass = (Hlim + 31) / 32;
  const int nS = (N1 + 1) >> 1; // steps: k1 pairs
  const unsigned rowbytes = (unsigned) Hp * 16u;
  const unsigned halfoff = (unsigned) hh * (unsigned) R2 * rowbytes; // the high half reads k1 = 2 s + 1: R2 row pairs on
  u32x4 rf[RD], rc[RD];
  {
    const unsigned lo0 = (unsigned) (c32 < H ? c32 : H - 1) * 16u + halfoff;
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      rf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, lo0, (unsigned) t * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, lo0, (unsigned) t * rowbytes, 0);
    }
  }
  for (int cp = 0; cp < npass; cp++)
  {
    const int ky = cp * 32 + c32;
    const int kyc = ky < H ? ky : H - 1;
    const unsigned laneoff = (unsigned) kyc * 16u + halfoff;
    const int kyn = ky + 32 < H ? ky + 32 : H - 1;
    const unsigned laneoff_next = (unsigned) kyn * 16u + halfoff;
    const bool has_next = cp + 1 < npass;
    float Tr[NACC], Ti[NACC];
#pragma unroll
    for (int d = 0; d < NACC; d++)
    {
      Tr[d] = 0.f;
      Ti[d] = 0.f;
    }
    if (ky < Hlim) // (both lanes of a column share the test: the swap below never meets a masked partner)
    for (int s = 0; s < nS; s++)
    {
      float xr[R], xi[R];
      // w_N^(OFF GS k1) of this lane's k1 = 2 s + hh: the high half's rows are OFF GS pixels further
      const float2 rot = a.tw[(OFF * GS * (2 * s + hh)) % N];
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
        // next request of the ring: row pair k2p + RD of this step, or of the next step / the next pass
        int sn = s, kn = k2p + RD;
        if (kn >= R2)
        {
          kn -= R2;
          sn = s + 1;
        }
        unsigned vo = laneoff;
        if (k2p + RD >= R2 && sn >= nS)
        { // runs on into the next pass (or re-reads the last row pair at the very end)
          sn = has_next ? 0 : nS - 1;
          kn = has_next ? kn : R2 - 1;
          vo = has_next ? laneoff_next : laneoff;
        }
        const unsigned so = (unsigned) (sn * 2 * R2 + kn) * rowbytes;
        rf[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, vo, so, 0);
        rc[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, vo, so, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      FFT_RUN(xr, xi);
      // the OFF outputs that the HIGH half will fold: times w_N^(OFF GS k1); then the exchange between the halves
#pragma unroll
      for (int j = 0; j < OFF; j++)
      {
        const int pp = ((((j - WD) * GS) % R) + R) % R, qq = (pp + R2) % R; // residues of row j and of row j + OFF
        const float yr = xr[FFT_OUT(qq)], yi = xi[FFT_OUT(qq)];
        const float zr = fmaf(-yi, rot.y, yr * rot.x);
        const float zi = fmaf(yi, rot.x, yr * rot.y);
        const u32x2s sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(xr[FFT_OUT(pp)]), __float_as_uint(zr), false, false);
        const u32x2s si = __builtin_amdgcn_permlane32_swap(__float_as_uint(xi[FFT_OUT(pp)]), __float_as_uint(zi), false, false);
        xr[FFT_OUT(pp)] = __uint_as_float(sr.x); // k1 = 2 s:     residue pp (low half) | pp + R / 2 (high half)
        xr[FFT_OUT(qq)] = __uint_as_float(sr.y); // k1 = 2 s + 1
        xi[FFT_OUT(pp)] = __uint_as_float(si.x);
        xi[FFT_OUT(qq)] = __uint_as_float(si.y);
      }
      // recombination: T[a] += wE[a] E_j + wO[a] O_j, twiddles of the LOW rows (wave-uniform), four accumulators at a time
      const const_float4_ptr twk = (const_float4_ptr) (unsigned long long) (reinterpret_cast<const float4 *>(a.twk) + (size_t) s * NACC);
#pragma unroll
      for (int a0 = 0; a0 < NACC; a0 += 4)
      {
        float4 wk[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
          const int ae = a0 + e < NACC ? a0 + e : NACC - 1;
          wk[e] = make_float4(twk[ae].x, twk[ae].y, twk[ae].z, twk[ae].w);
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
          if (a0 + e < NACC)
          {
            const int j = (a0 + e) % OFF;
            const int pp = ((((j - WD) * GS) % R) + R) % R, qq = (pp + R2) % R;
            float tr = Tr[a0 + e], ti = Ti[a0 + e];
            tr = fmaf(xr[FFT_OUT(pp)], wk[e].x, tr);
            tr = fmaf(-xi[FFT_OUT(pp)], wk[e].y, tr);
            tr = fmaf(xr[FFT_OUT(qq)], wk[e].z, tr);
            tr = fmaf(-xi[FFT_OUT(qq)], wk[e].w, tr);
            ti = fmaf(xr[FFT_OUT(pp)], wk[e].y, ti);
            ti = fmaf(xi[FFT_OUT(pp)], wk[e].x, ti);
            ti = fmaf(xr[FFT_OUT(qq)], wk[e].w, ti);
            ti = fmaf(xi[FFT_OUT(qq)], wk[e].z, ti);
            Tr[a0 + e] = tr;
            Ti[a0 + e] = ti;
          }
        }
      }
    }
    // FFTW c2r convention: columns 0 and N/2 enter once (real part only after the ky pass), others twice
    float wgt = 2.f;
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= Hlim)
      wgt = 0.f;
    floatx4 D[3][3];
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int ct = 0; ct < 3; ct++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          D[rt][ct][i] = Dpark[((rt * 3 + ct) * 4 + i) * 64];
    {
      // B operand of k-step K = cp 16 + ks (columns 2 K, 2 K + 1), column tile ct, as this lane supplies it: fetched
      // through a ring PF k-steps ahead -- the requests of the first PF k-steps fly while T moves into the planes
      constexpr int PF = 4;
      const unsigned soff0 = (unsigned) cp * 16u * 768u;
      float bq[PF][3];
      const int nks = min(16, (Hlim - cp * 32 + 1) >> 1);
#pragma unroll
      for (int q = 0; q < PF; q++)
#pragma unroll
        for (int ct = 0; ct < 3; ct++)
          bq[q][ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcB, (unsigned) lane * 4u, soff0 + (unsigned) (q * 3 + ct) * 256u, 0));
      WAVE_OR_BLOCK_SYNC(); // D is out of LDS
#pragma unroll
      for (int d = 0; d < NACC; d++)
      {
        const int m = (d / OFF) * 2 * OFF + (d % OFF); // low-half row; the high half's row is OFF further
        if (m < kFm2PlaneRows)
        {
          if (m + OFF < kFm2PlaneRows || hh == 0)
          {
            Pl[(m + OFF * hh) * PS + c32] = Tr[d] * wgt;
            Pl[PLANE + (m + OFF * hh) * PS + c32] = -(Ti[d] * wgt);
          }
        }
      }
      WAVE_OR_BLOCK_SYNC();
#pragma unroll
      for (int k4 = 0; k4 < 16; k4 += PF)
      {
        if (k4 < nks) // (k-steps beyond the last column pair of a chunk of PF multiply zeros)
        {
#pragma unroll
          for (int q = 0; q < PF; q++)
          {
            const int ks = k4 + q;
            float av[3], bv[3];
#pragma unroll
            for (int rt = 0; rt < 3; rt++)
              av[rt] = Arow[rt * 16 * PS + 2 * ks];
#pragma unroll
            for (int ct = 0; ct < 3; ct++)
              bv[ct] = bq[q][ct];
            if (ks + PF < 16)
            {
#pragma unroll
              for (int ct = 0; ct < 3; ct++)
                bq[q][ct] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrcB, (unsigned) lane * 4u, soff0 + (unsigned) ((ks + PF) * 3 + ct) * 256u, 0));
            }
#pragma unroll
            for (int rt = 0; rt < 3; rt++)
#pragma unroll
              for (int ct = 0; ct < 3; ct++)
                D[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], D[rt][ct], 0, 0, 0);
          }
        }
      }
    }
    WAVE_OR_BLOCK_SYNC(); // the planes are read: their space takes the accumulators until the next matrix pass
#pragma unroll
    for (int rt = 0; rt < 3; rt++)
#pragma unroll
      for (int ct = 0; ct < 3; ct++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          Dpark[((rt * 3 + ct) * 4 + i) * 64] = D[rt][ct][i];
  }

  // tile element (rt, ct, i) of this lane: window row m = 16 rt + 4 kq + i, window column n = 16 ct + jn
  const int mD = a.maxD / GS;
  const double2 pc = a.postc[oc];
  const PostW pw = post_consts(a.pd.Ntotpi, N, a.params[oc], a.sumRef[p], a.sumsqRef[p], pc.x, pc.y);
  const float *tq = NYQ ? a.tnyq + ((size_t) p * a.ldPart + oc) * (2 * a.nyqWD + 1) + a.nyqWD - WD : nullptr;
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
  int rankn[3];
#pragma unroll
  for (int ct = 0; ct < 3; ct++)
    rankn[ct] = rankW[16 * ct + jn];
  // posterior: one row tile (12 values of a lane) per iteration of a ROLLED loop -- the accumulators are fetched from
  // their parking place with a runtime offset; unrolled over all 36 values the epilogue spilled 22 registers
#pragma unroll 1
  for (int rt = 0; rt < 3; rt++)
  {
    float accv[12];
    int idv[12];
    bool okv[12];
#pragma unroll
    for (int i = 0; i < 4; i++)
    { // element (i, ct): window row m = 16 rt + 4 kq + i, window column n = 16 ct + jn
      const int m = 16 * rt + 4 * kq + i;
      const int rk = rankW[m];
      float nq = 0.f;
      if (NYQ) // (-1)^dy Re T[dx][N/2]: weight 1, real part only
        nq = tq[min(max(m, WD - mD), WD + mD)];
#pragma unroll
      for (int ct = 0; ct < 3; ct++)
      {
        float v = Dpark[(rt * 12 + ct * 4 + i) * 64];
        if (NYQ)
          v = fmaf((((16 * ct + jn - WD) * GS) & 1) ? -1.f : 1.f, nq, v);
        accv[i * 3 + ct] = v;
        okv[i * 3 + ct] = rk >= 0 && rankn[ct] >= 0;
        idv[i * 3 + ct] = rk * a.nd + rankn[ct];
      }
    }
    posterior_batch<12>(L, accv, idv, okv, pw, ltab, a.algo);
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
