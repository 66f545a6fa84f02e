// compare_fastm.hpp -- 23..31-row translation windows: k_compare_fast's column pass, window pass on the matrix cores
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_FASTM_HPP
#define BIOEM_COMPARE_FASTM_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// Windows of 23..31 rows (+-11 ... +-15 px at unit stride).  One wave = one comparison, lane = frequency column, as in
// k_compare_fast; what differs is the window (row) pass
//     cc[dx][dy] = sum_ky  Re T[dx][ky] cos(2 pi ky dy / N) - Im T[dx][ky] sin(2 pi ky dy / N)      (weights folded into T)
// which costs the vector pipe 4 W^2 H / 64 instructions per comparison (3.6 k of 16.7 k at 224^2 +-13 px, 4.9 k at +-15 px)
// and 16 live accumulators per lane.  A window of up to 32 x 32 displacements is exactly one 32 x 32 MFMA tile:
//     D[m][n] += A[m][k] B[k][n],   m = window row (dx), n = window column (dy), k = (ky, re | im)
// with v_mfma_f32_32x32x2_f32 -- f32 operands, f32 accumulation, bit-exact products (no reduced precision anywhere): one
// instruction per frequency column, issued on the otherwise idle matrix pipe while the other waves of the SIMD run their
// register FFTs on the vector pipe.  27 rows fill 71 % of the tile, 31 rows 94 % (the 21-row window of the headline
// shape fills 43 % and stays on the vector pipe, where it measured faster).
//   A operand: lane l supplies A[l % 32][l / 32] = (l < 32 ? Re T : -Im T)[row l % 32][ky] from two float planes in LDS,
//     row stride 33 floats: the 32 rows of a lane group hit 32 different banks (ds_read_b32, no conflicts); T reaches
//     the planes 32 columns at a time (half-width exchange: 8.25 KiB per wave, three blocks per CU)
//   B operand: lane l supplies (l < 32 ? cos : sin)(2 pi ky dy_n / N), n = l % 32, from cos / sin planes of the twiddle
//     table stored with one pad word per 32 entries (index i -> i + i / 32): the arithmetic progressions ky dy_n mod N
//     of a lane group spread over the banks instead of piling on one (stride 32 would otherwise be a 7-way conflict)
//   D: 16 registers per lane = window column n = l % 32, rows m = 8 (i / 4) + 4 (l / 32) + i % 4, i = 0..15: the
//     posterior takes them in two batches of 8 (posterior_batch)
// Window rows m = 0..2 WD hold dx = (m - WD) GS; rankW[m] is the visiting rank of that offset in the reference's order
// (bioem_algorithm.h:156-197) or -1 where the displacement list has no such offset (ALGO 1 with maxD % grid != 0).
// ------------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

__host__ __device__ constexpr int fastm_table_floats(int N) { return (N + (N >> 5) + 4) & ~3; }

template <int WD, int R, bool NYQ, int GS>
__global__ __launch_bounds__(256, 3) void k_compare_fastm(const CompareArgs a)
{
  static_assert(WD <= 15, "windows of at most 31 rows");
  constexpr int NW = 2 * WD + 1;
  constexpr int R2 = R / 2;
  constexpr int RD = (R2 % 4 == 0) ? 4 : (R2 % 5 == 0) ? 5 : (R2 % 3 == 0) ? 3 : (R2 % 2 == 0) ? 2 : 1;
  constexpr int PS = 33;         // plane row stride in floats
  constexpr int PLANE = 32 * PS; // floats per plane (32 tile rows)
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1;
  const int NT = fastm_table_floats(N);
  float *tcos = reinterpret_cast<float *>(smem);
  int *rankW = reinterpret_cast<int *>(smem + (size_t) 2 * NT * 4);                  // 32 ints
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) 2 * NT * 4 + 128);       // 64 entries
  float *Pall = reinterpret_cast<float *>(smem + (size_t) 2 * NT * 4 + 128 + 1024);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  float *Pl = Pall + (size_t) wave * 2 * PLANE;
  // the 16 tile accumulators of a lane rest here while the next column block is transformed (they would otherwise be
  // live across the whole column pass: 16 registers the register FFT needs)
  float *Dpark = Pall + (size_t) 4 * 2 * PLANE + (size_t) wave * 16 * 64 + lane;

  for (int t = threadIdx.x; t < N; t += blockDim.x)
  {
    const float2 w = a.tw[t];
    tcos[t + (t >> 5)] = w.x;
    tcos[NT + t + (t >> 5)] = w.y;
  }
  if (threadIdx.x < 32)
    rankW[threadIdx.x] = -1;
  for (int t = threadIdx.x; t < 64; t += blockDim.x)
    ltab[t] = a.ltab[t];
  // tile rows beyond the window never receive T: zero them once (their outputs are masked, this only keeps them finite)
  for (int t = lane; t < (32 - NW) * PS; t += 64)
  {
    Pl[NW * PS + t] = 0.f;
    Pl[PLANE + NW * PS + t] = 0.f;
  }
  __syncthreads();
  const int mD = a.maxD / GS;
  for (int t = threadIdx.x; t < a.nd; t += blockDim.x)
  {
    const int m = a.disp[t] / GS + WD;
    if (m >= 0 && m < 32)
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

  // matrix-pass lane constants: tile column n = lane % 32 (dy), operand half kh = lane / 32 (re | im, cos | sin)
  const int n = lane & 31, kh = lane >> 5;
  const int dyn = n < NW ? (n - WD) * GS : 0;
  const int stepn = dyn < 0 ? dyn + N : dyn;
  const float *Arow = Pl + kh * PLANE + n * PS; // A[m = lane % 32][kh] of column kk: Arow[kk]
  const float *Btab = tcos + kh * NT;
  floatx16 Dfin;

  const int nblk = NYQ ? (H - 1) / 64 : (H + 63) / 64;
  const unsigned rowbytes = (unsigned) Hp * 16u;
  // split last block (a.split, as in k_compare_fast: at most 32 columns, shared by the half-waves -- the low half the k1
  // steps 0 .. sHalf - 1, the high half the rest, its sums turned by w^(dx sHalf) and added after the loop)
  // (the 31-row kernels with 10- and 8-point FFTs have no registers left for it: they keep the whole pass)
  constexpr bool SPLIT_OK = !NYQ && !(WD == 15 && R <= 10);
  const int sHalf = (N1 + 1) >> 1;
  const int hsel = lane >> 5;
  const unsigned halfoff = (unsigned) (hsel * sHalf * R2) * rowbytes;
  auto lane_offset = [&](int b) -> unsigned {
    const bool sp = SPLIT_OK && a.split && b == nblk - 1;
    const int kyb = b * 64 + (sp ? (lane & 31) : lane);
    return (unsigned) (kyb < H ? kyb : H - 1) * 16u + (sp ? halfoff : 0u);
  };
  u32x4 rf[RD], rc[RD];
  {
    const unsigned lo0 = lane_offset(0);
#pragma unroll
    for (int t = 0; t < RD; t++)
    {
      rf[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, lo0, (unsigned) t * rowbytes, 0);
      rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, lo0, (unsigned) t * rowbytes, 0);
    }
  }
  for (int blk = 0; blk < nblk; blk++)
  {
    const bool split = SPLIT_OK && a.split && blk == nblk - 1;
    const int n1 = split ? sHalf : N1;
    const int ttotal = R2 * n1;
    const int ky = blk * 64 + (split ? (lane & 31) : lane);
    const unsigned laneoff = lane_offset(blk);
    const unsigned laneoff_next = lane_offset(min(blk + 1, nblk - 1));
    const bool has_next = blk + 1 < nblk;
    float Tr[NW], Ti[NW];
#pragma unroll
    for (int d = 0; d < NW; d++)
    {
      Tr[d] = 0.f;
      Ti[d] = 0.f;
    }
    if (ky < H)
    for (int k1 = 0; k1 < n1; k1++)
    {
      float xr[R], xi[R];
      const const_float2_ptr twk = as_constant(a.twk) + (size_t) k1 * NW;
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
        // (only the last RD requests of a k1 step can leave the block: the first test folds at compile time)
        if (k2p + RD >= R2 && tn >= ttotal)
        {
          tn = has_next ? tn - ttotal : ttotal - 1;
          vo = has_next ? laneoff_next : laneoff;
        }
        rf[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, vo, (unsigned) tn * rowbytes, 0);
        rc[k2p % RD] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, vo, (unsigned) tn * rowbytes, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      FFT_RUN(xr, xi);
      // recombination for the window rows only:  T[dx] += w_N^(dx k1) y_k1[dx mod R]; the twiddles of a k1 arrive in
      // chunks of 8 rows (wide scalar loads) so that no more than 16 scalar registers hold them at a time
#pragma unroll
      for (int d0 = 0; d0 < NW; d0 += 8)
      {
        float2 wk[8];
#pragma unroll
        for (int e = 0; e < 8; e++)
        {
          const int d = d0 + e < NW ? d0 + e : NW - 1;
          wk[e] = make_float2(twk[d].x, twk[d].y);
        }
#pragma unroll
        for (int e = 0; e < 8; e++)
        {
          const int d = d0 + e;
          if (d < NW)
          {
            const int pos = FFT_OUT(((((d - WD) * GS) % R) + R) % R);
            float tr = Tr[d], ti = Ti[d];
            tr = fmaf(xr[pos], wk[e].x, tr);
            tr = fmaf(-xi[pos], wk[e].y, tr);
            ti = fmaf(xr[pos], wk[e].y, ti);
            ti = fmaf(xi[pos], wk[e].x, ti);
            Tr[d] = tr;
            Ti[d] = ti;
          }
        }
      }
    }
    if (split)
    {
      const const_float2_ptr ws = as_constant(a.twk) + (size_t) sHalf * NW; // w^(dx sHalf), the table's row sHalf
#pragma unroll
      for (int d = 0; d < NW; d++)
      {
        const float wx = hsel ? ws[d].x : 1.f, wy = hsel ? ws[d].y : 0.f;
        const float tr = fmaf(-Ti[d], wy, Tr[d] * wx), ti = fmaf(Ti[d], wx, Tr[d] * wy);
        const u32x2 sr = __builtin_amdgcn_permlane32_swap(__float_as_uint(tr), __float_as_uint(tr), false, false);
        const u32x2 si = __builtin_amdgcn_permlane32_swap(__float_as_uint(ti), __float_as_uint(ti), false, false);
        Tr[d] = __uint_as_float(sr.x) + __uint_as_float(sr.y); // low half + high half, in every lane
        Ti[d] = __uint_as_float(si.x) + __uint_as_float(si.y);
      }
    }
    // FFTW c2r convention: columns 0 and N/2 enter once (real part only after the ky pass), others twice
    float wgt = 2.f;
    if (ky == 0 || (((N & 1) == 0) && ky == N / 2))
      wgt = 1.f;
    if (ky >= H || (split && hsel))
      wgt = 0.f;
    floatx16 D;
    if (blk == 0)
    {
#pragma unroll
      for (int i = 0; i < 16; i++)
        D[i] = 0.f;
    }
    else
    {
#pragma unroll
      for (int i = 0; i < 16; i++)
        D[i] = Dpark[i * 64];
    }
#pragma unroll
    for (int hh = 0; hh < 2; hh++)
    {
      WAVE_OR_BLOCK_SYNC(); // the matrix pass of the previous half has read its operands
      if ((lane >> 5) == hh)
      {
#pragma unroll
        for (int d = 0; d < NW; d++)
        {
          Pl[d * PS + (lane & 31)] = Tr[d] * wgt;
          Pl[PLANE + d * PS + (lane & 31)] = -(Ti[d] * wgt);
        }
      }
      WAVE_OR_BLOCK_SYNC();
      const int ky0 = blk * 64 + hh * 32;
      if (ky0 < (NYQ ? H - 1 : H))
      {
        unsigned idx = (unsigned) (((long long) ky0 * stepn) % N);
        // columns of this half that exist (the rest hold zeros: their products leave D unchanged).  (A separate fully
        // unrolled stream for full halves is another 1 % faster at 200^2 and spills two registers in <15, 10>.)
        const int nk = min(32, (NYQ ? H - 1 : H) - ky0);
#pragma unroll 8
        for (int kk = 0; kk < nk; kk++)
        {
          const float av = Arow[kk];
          const float bv = Btab[idx + (idx >> 5)];
          D = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, D, 0, 0, 0);
          idx += (unsigned) stepn;
          idx = min(idx, idx - (unsigned) N); // idx < 2N: the wrapped difference is huge unless idx >= N
        }
      }
    }
    if (has_next)
    {
#pragma unroll
      for (int i = 0; i < 16; i++)
        Dpark[i * 64] = D[i];
    }
    Dfin = D;
  }
  floatx16 D = Dfin;

  // tile element i of this lane: window row m = 8 (i / 4) + 4 kh + i % 4, window column n
  // (window tiles: only the first ndx rows / ndy columns of the tile's sorted list count -- edge tiles stick out of the
  // window; an untiled launch has ndx = ndy = nd and the displacement list alone decides)
  const int sn = n - WD + mD; // position of the column in the sorted window of a tile launch
  const bool okn = n < NW && rankW[n < NW ? n : 0] >= 0 && (a.ndy >= a.nd || sn < a.ndy);
  const int rankn = rankW[n < NW ? n : 0];
  if (NYQ)
  {
    const float *tq = a.tnyq + ((size_t) p * a.ldPart + oc) * NW;
    const float sg = (dyn & 1) ? -1.f : 1.f;
#pragma unroll
    for (int i = 0; i < 16; i++)
    {
      const int m = 8 * (i / 4) + 4 * kh + (i % 4);
      D[i] = fmaf(sg, tq[m < NW ? m : NW - 1], D[i]);
    }
  }
  const PostW pw = post_consts(a.pd.Ntotpi, N, a.params[oc], a.sumRef[p], a.sumsqRef[p], a.postc[oc].x, a.postc[oc].y);
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
#pragma unroll
  for (int b = 0; b < 2; b++)
  {
    float accv[8];
    int idv[8];
    bool okv[8];
#pragma unroll
    for (int j = 0; j < 8; j++)
    {
      const int i = 8 * b + j;
      const int m = 8 * (i / 4) + 4 * kh + (i % 4);
      const int rk = rankW[m < NW ? m : 0];
      accv[j] = D[i];
      okv[j] = okn && m < NW && rk >= 0 && (a.ndx >= a.nd || (m - WD + mD) < a.ndx);
      idv[j] = rk * a.nd + rankn;
    }
    posterior_batch<8>(L, accv, idv, okv, pw, ltab, a.algo);
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
