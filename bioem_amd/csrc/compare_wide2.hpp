// compare_wide2.hpp -- wide translation windows (more than 31 offsets per axis) with a row FFT: one comparison per block
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_WIDE2_HPP
#define BIOEM_COMPARE_WIDE2_HPP

namespace
{

// ------------------------------------------------------------------------------------------------
// The reference reads any window out of one full c2r (bioem.cpp:1458, doc/index.rst:1355-1360: cost "almost
// independent" of the displacement count).  The tiled kernels pay the column transforms once per 21-row x-tile and a
// direct O(W^2 H) row pass; this kernel pays every transform once and runs the row pass as an FFT, too:
//
//   column pass  T[dx][ky] = sum_k1 w_N^(dx k1) y_k1[dx mod R][ky],  y_k1 = IFFT_R over k2 of X[N1 k2 + k1][ky]
//     The four waves of the block work on ONE comparison.  In rounds of four (k1, column block) steps every wave
//     forms the spectrum product and the R-point register FFT of one step (lane = ky, as in k_compare_fast) and parks
//     the R outputs in an LDS slot; after a barrier every wave folds all four slots into ITS quarter of the window
//     rows (<= NRW rows: 2 NRW NBLK accumulators in registers).  Product and FFT are paid once per comparison, not
//     once per x-tile.
//   T -> LDS     [rows][TS] float2, unweighted (the c2r weights are folded into the Hermitian extension below); the
//     slots are dead by then and share the space.
//   row pass     cc[dx][dy] = Re sum_ky w_ky T[dx][ky] e^(2 pi i ky dy / N)  (FFTW c2r: columns 0 and N/2 enter once,
//     real part only).  Two rows a, b at a time as ONE complex length-N inverse transform of U = G_a + i G_b, with G
//     the Hermitian extension of the half spectrum (G[N - ky] = conj(T[ky]), Im G[0] = Im G[N/2] = 0): the result is
//     cc_a + i cc_b.  Same N = N1 R split: lane = (row pair, k1) runs one R-point register FFT over k2 (for k2 < R/2
//     the inputs are direct, beyond mirrored -- a compile-time property of k2), writes y[n][k1] over the pair's own
//     rows, then lanes = dy recombine  S[dy] = sum_k1 w_N^(dy k1) y[dy mod R][k1]  and run the posterior for
//     (a, dy) = Re S and (b, dy) = Im S.  Pairs are private to a wave: no block barrier inside the row pass.
//   one Partial per comparison (the four waves' log-sum-exp states merged in wave order), ids = global visiting
//     ranks: the fold kernels see what k_compare_fast would have produced for this window.
// Window rows m = 0..nd-1 hold dx = (m - mD) gs (sorted); dinv[m] is the visiting rank of that offset in the
// reference's order (bioem_algorithm.h:156-197 / bioem.cpp:1477-1485).
// LDS: tables + max(4 R 512 B, rows2 TS 8 B): 76 KiB at 224^2 +-40 px -> two blocks per CU.
// ------------------------------------------------------------------------------------------------
// diagnostic build only (-DBIOEM_W2_STAMPS, never shipped): shader-clock cycles per phase of wave 0, summed over blocks
#ifdef BIOEM_W2_STAMPS
__device__ unsigned long long g_w2_stamps[16];
#define W2_STAMP(k)                                                                                                \
  do                                                                                                               \
  {                                                                                                                \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
    if (threadIdx.x == 0)                                                                                          \
      atomicAdd(&g_w2_stamps[k], now_ - stamp_);                                                                   \
    stamp_ = now_;                                                                                                 \
  } while (0)
#else
#define W2_STAMP(k)
#endif

// HALVES = 2: the T block goes through LDS in two halves of the window rows (row pass + posterior per half), for the
// sizes whose whole T block would leave one block per CU (240^2 ... 256^2 at +-35 ... +-42 px)
// NW = 8: eight waves per comparison (512-thread blocks) -- half the window rows per wave, so that the 16-point
// instantiation with 11 rows per wave (110 registers) covers windows of up to 88 rows at FOUR waves per SIMD with two
// blocks per CU, and larger images / windows keep a register-sized share per wave
#ifndef BIOEM_W2_HALVES_WAVES
#define BIOEM_W2_HALVES_WAVES 2 // (experiment builds: 3 = the halves kernels under the three-wave register bound)
#endif
template <int R, int NRW, int NBLK, bool NYQ, int HALVES = 1, int NW = 4>
// waves per SIMD the registers must allow: the T block in halves means its LDS footprint holds a CU to two blocks anyway
__global__ __launch_bounds__(64 * NW, NW == 8 ? ((NRW * NBLK <= 26 && R <= 16) ? 4 : 2)
                                      : HALVES == 2 ? BIOEM_W2_HALVES_WAVES
                                                    : ((NRW * NBLK <= 42 && R <= 16) ? 3 : 2)) void k_compare_wide2(const CompareArgs a)
{
  constexpr int R2 = R / 2;
  // depth of the operand ring (divides R2): the first RD row pairs of a wave's next step are issued before the
  // round's barriers.  With only two waves per SIMD the ring is what hides the L2 latency: as deep as registers allow
#ifndef BIOEM_W2_RING
#define BIOEM_W2_RING 4
#endif
  constexpr int RD = (R2 % BIOEM_W2_RING == 0) ? BIOEM_W2_RING
                     : (R2 % 4 == 0)           ? 4
                     : (R2 % 5 == 0)           ? 5
                     : (R2 % 3 == 0)           ? 3
                     : (R2 % 2 == 0)           ? 2
                                               : 1;
  struct PostConst
  {
    double t2, prior;
    bioem_hip_param5 q;
    float sumref, sumsqref;
  };
  extern __shared__ __align__(16) unsigned char smem[];
  const int N = a.N, H = a.H, N1 = a.N1, nd = a.nd, TS = a.ts;
  float2 *twl = reinterpret_cast<float2 *>(smem);
  int *dinv = reinterpret_cast<int *>(smem + (size_t) ((N + 2) & ~1) * 8);                 // nd ints (512 B reserved)
  double2 *ltab = reinterpret_cast<double2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 512);    // 64 entries
  // wave results (NW x 24 B) and posterior constants (48 B) share 256 B
  LseF *lsew = reinterpret_cast<LseF *>(smem + (size_t) ((N + 2) & ~1) * 8 + 512 + 1024);
  PostConst *cst = reinterpret_cast<PostConst *>(smem + (size_t) ((N + 2) & ~1) * 8 + 512 + 1024 + 192);
  float2 *U = reinterpret_cast<float2 *>(smem + (size_t) ((N + 2) & ~1) * 8 + 512 + 1024 + 256);
  const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int gs = a.gs, mD = a.maxD / gs;

  for (int t = threadIdx.x; t <= N; t += blockDim.x)
    twl[t] = a.tw[t];
  for (int t = threadIdx.x; t < nd; t += blockDim.x)
    dinv[a.disp[t] / gs + mD] = t; // a.disp = offsets in visiting order
  for (int t = threadIdx.x; t < 64; t += blockDim.x)
    ltab[t] = a.ltab[t];

  int p, oc;
  { // block order of k_compare_fast with one comparison per block: particle chunks, particle index fastest
    const int per = a.pchunk * a.nOC;
    int c = blockIdx.x / per;
    const int nch = (a.nMaps + a.pchunk - 1) / a.pchunk;
    c = min(c, nch - 1);
    const int rem = blockIdx.x - c * per;
    const int pc = min(a.pchunk, a.nMaps - c * a.pchunk);
    oc = rem / pc;
    p = c * a.pchunk + (rem - oc * pc);
  }
  // provably wave-uniform (SGPRs): the buffer descriptors built from them must not end up in vector registers -- every
  // buffer load would then sit in a waterfall loop (4 v_readfirstlane + 2 v_cmp + exec juggling per load)
  p = __builtin_amdgcn_readfirstlane(p);
  oc = __builtin_amdgcn_readfirstlane(oc);
  // constants of the comparison's posterior: fetched by wave 0 while the tables load (read where the posterior
  // starts, their L2 round trips were exposed in all four waves)
  if (wave == 0)
  {
    const bioem_hip_param5 q = a.params[oc];
    const float sumref = a.sumRef[p], sumsqref = a.sumsqRef[p];
    const double2 pc = a.postc[oc];
    const double t2 = pc.x, prior = pc.y;
    if (lane == 0)
    {
      cst->t2 = t2;
      cst->prior = prior;
      cst->q = q;
      cst->sumref = sumref;
      cst->sumsqref = sumsqref;
    }
  }
  const size_t M = (size_t) N * H;
  // timing-only ablation builds (never shipped): zero-record descriptors drop the operand traffic, the instruction
  // stream and waits stay
#ifndef BIOEM_W2_ABLATE
#define BIOEM_W2_ABLATE 0
#endif
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.ref + (size_t) p * M)), 0,
                                                       BIOEM_W2_ABLATE ? 0 : (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.conv + (size_t) oc * M)), 0,
                                                       BIOEM_W2_ABLATE ? 0 : (int) (M * sizeof(float2)), 0x00020000);

#ifdef BIOEM_W2_STAMPS
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
  // ---------------- column pass ----------------
  const int nblk = NYQ ? (H - 1) / 64 : (H + 63) / 64;
  const int rpw = (nd + NW - 1) / NW; // window rows per wave
  if (nblk > NBLK || rpw > NRW)
  { // launched on a shape this instantiation does not cover (a dispatch error): poison the result instead of
    // dropping columns or rows silently
    if (threadIdx.x == 0)
    {
      Partial r;
      r.sumExp = __builtin_nan("");
      r.best = __builtin_nanf("");
      r.id = 0;
      r.value = 0.f;
      r.pad = 0;
      a.partials[(size_t) p * a.ldPart + oc] = r;
    }
    return;
  }
  const int r0 = wave * rpw;
  const int nrows = max(0, min(rpw, nd - r0));
  const unsigned rowbytes = (unsigned) H * 16u;
  float Tr[NBLK][NRW], Ti[NBLK][NRW];
#pragma unroll
  for (int b = 0; b < NBLK; b++)
#pragma unroll
    for (int d = 0; d < NRW; d++)
    {
      Tr[b][d] = 0.f;
      Ti[b][d] = 0.f;
    }
  __syncthreads();
  // LDS position of window row d's residue row inside a slot: lane addresses held in registers over the column pass,
  // except in the 8-point instantiations, which form the wave-uniform part on the scalar side where it is used
  // (measured: registers +1 % for R >= 16, scalar +8 % for R = 8 at 120^2 +-25 px)
  #ifndef BIOEM_W2_YOFF
#define BIOEM_W2_YOFF 1
#endif
  constexpr bool YOFF_REGS = BIOEM_W2_YOFF && R > 8 && !(HALVES == 2 && NBLK >= 3) && !(HALVES == 2 && NRW > 21);
  const int dx0 = (r0 - mD) * gs;
  const int res0 = ((dx0 % R) + R) % R; // residue of the wave's first row; row d: (res0 + d gs) mod R
  int yoff[YOFF_REGS ? NRW : 1];
  if (YOFF_REGS)
  {
#pragma unroll
    for (int d = 0; d < NRW; d++)
      yoff[YOFF_REGS ? d : 0] = ((res0 + d * gs) % R) * 64 + lane;
  }
  // particle row pairs of this wave's NEXT step may be requested as soon as the current step's outputs are parked
  // (BIOEM_W2_FNEXT; measured: -1.6 %, and the kernel without ANY operand traffic -- zero-record descriptors -- is only
  // 3 % faster: operand delivery is not what the time goes to, so the registers are spent elsewhere)
#ifndef BIOEM_W2_FNEXT
#define BIOEM_W2_FNEXT 0
#endif
  // particle row pairs requested at once: all R/2, or a ring of 8 where three column blocks of accumulators leave
  // no room for 16 (k_compare_wide2<32, 21, 3>)
  constexpr int RF = (NBLK >= 3 && R2 == 16) ? 8 : R2;
  u32x4 fx[RF];
  bool fready = false;
  auto request_f = [&](int k1n, unsigned laneoffn) {
#pragma unroll
    for (int t = 0; t < RF; t++)
      fx[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, laneoffn, (unsigned) (k1n * R2 + t) * rowbytes, 0);
  };
#pragma unroll
  for (int blk = 0; blk < NBLK; blk++)
  {
    if (blk < nblk)
    {
      const int ky = blk * 64 + lane;
      const unsigned laneoff = (unsigned) (ky < H ? ky : H - 1) * 16u;
      const int kyn = ky + 64;
      const unsigned laneoff_next = (unsigned) (kyn < H ? kyn : H - 1) * 16u;
      for (int base = 0; base < N1; base += NW)
      {
        const int k1 = base + wave;
        if (k1 < N1)
        { // this wave's step of the round: product + register FFT, outputs to slot `wave`
          float xr[R], xi[R];
          // operands of the step: ALL R/2 particle row pairs are requested at once -- they land in the registers the
          // FFT inputs take over (a row pair of F is consumed exactly when its two products are formed), the conv
          // row pairs follow through a ring
#ifndef BIOEM_W2_CRING
#define BIOEM_W2_CRING 4
#endif
          constexpr int RC = (R2 % BIOEM_W2_CRING == 0) ? BIOEM_W2_CRING : RD;
          u32x4 rc[RC];
          if (!fready)
            request_f(k1, laneoff);
#pragma unroll
          for (int t = 0; t < RC; t++)
            rc[t] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, laneoff, (unsigned) (k1 * R2 + t) * rowbytes, 0);
#pragma unroll
          for (int k2p = 0; k2p < R2; k2p++)
          {
            const float4 f = as_float4(fx[k2p % RF]);
            const float4 c = as_float4(rc[k2p % RC]);
            // X = conv * conj(ref)   (bioem.cpp:1452-1455)
            xr[FFT_IN(2 * k2p)] = fmaf(c.x, f.x, c.y * f.y);
            xi[FFT_IN(2 * k2p)] = fmaf(c.y, f.x, -(c.x * f.y));
            xr[FFT_IN(2 * k2p + 1)] = fmaf(c.z, f.z, c.w * f.w);
            xi[FFT_IN(2 * k2p + 1)] = fmaf(c.w, f.z, -(c.z * f.w));
            if (k2p + RF < R2)
              fx[k2p % RF] = __builtin_amdgcn_raw_buffer_load_b128(rsrcF, laneoff,
                                                                  (unsigned) (k1 * R2 + k2p + RF) * rowbytes, 0);
            if (k2p + RC < R2)
              rc[k2p % RC] = __builtin_amdgcn_raw_buffer_load_b128(rsrcC, laneoff,
                                                                  (unsigned) (k1 * R2 + k2p + RC) * rowbytes, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          W2_STAMP(8);
          FFT_RUN(xr, xi);
          W2_STAMP(9);
          float2 *slot = U + (size_t) wave * R * 64 + lane;
#pragma unroll
          for (int n = 0; n < R; n++)
            slot[n * 64] = make_float2(xr[FFT_OUT(n)], xi[FFT_OUT(n)]);
          // next step of this wave: same block four k1 further, or the first one of the next column block
          fready = false;
          if (BIOEM_W2_FNEXT && k1 + NW < N1)
          {
            request_f(k1 + NW, laneoff);
            fready = true;
          }
          else if (BIOEM_W2_FNEXT && blk + 1 < nblk && wave < N1)
          {
            request_f(wave, laneoff_next);
            fready = true;
          }
          W2_STAMP(10);
        }
#ifndef BIOEM_W2_TIMING_NOBARRIER
        __syncthreads();
#endif
        W2_STAMP(5);
        // fold the round's slots into this wave's rows:  T[dx] += w_N^(dx k1) * y_k1[dx mod R]
        // (all NRW accumulators; rows beyond this wave's share fold zeros and are never stored)
#pragma unroll
        for (int s = 0; s < NW; s++)
        {
          const int k1s = base + s;
#ifdef BIOEM_W2_TIMING_NOFOLD
          if (k1s < N1 && a.nd > 1000)
#else
          if (k1s < N1)
#endif
          {
            // this wave's NRW twiddles of step k1s are contiguous: a few wide scalar loads
            const const_float2_ptr twk = as_constant(a.twk) + ((size_t) k1s * NW + wave) * NRW;
            const float2 *ys = U + (size_t) s * R * 64;
#ifndef BIOEM_W2_FOLD_CHUNK
#define BIOEM_W2_FOLD_CHUNK 7
#endif
            // rows in chunks: a chunk's LDS reads are issued together, then its FMAs (the chunk size bounds the
            // registers the reads occupy)
            constexpr int FC = BIOEM_W2_FOLD_CHUNK;
#pragma unroll
            for (int d0 = 0; d0 < NRW; d0 += FC)
            {
              float2 w[FC], y[FC];
#pragma unroll
              for (int e = 0; e < FC; e++)
              {
                const int d = d0 + e < NRW ? d0 + e : NRW - 1;
#ifdef BIOEM_W2_TIMING_ONETW
                w[e] = make_float2(twk[0].x, twk[0].y); // timing only: one scalar load per slot instead of NRW
#else
                w[e] = make_float2(twk[d].x, twk[d].y);
#endif
#ifdef BIOEM_W2_TIMING_ONEY
                y[e] = ys[yoff[0]];                     // timing only: one LDS read per slot
#else
                y[e] = YOFF_REGS ? ys[yoff[YOFF_REGS ? d : 0]] : ys[((res0 + d * gs) % R) * 64 + lane];
#endif
              }
#pragma unroll
              for (int e = 0; e < FC; e++)
              {
                const int d = d0 + e;
                if (d < NRW)
                {
                  float tr = Tr[blk][d], ti = Ti[blk][d];
                  tr = fmaf(y[e].x, w[e].x, tr);
                  tr = fmaf(-y[e].y, w[e].y, tr);
                  ti = fmaf(y[e].x, w[e].y, ti);
                  ti = fmaf(y[e].y, w[e].x, ti);
                  Tr[blk][d] = tr;
                  Ti[blk][d] = ti;
                }
              }
            }
          }
        }
#ifndef BIOEM_W2_TIMING_NOBARRIER
        __syncthreads();
#endif
        W2_STAMP(6);
      }
    }
  }
  W2_STAMP(0);
  const PostW pw = post_consts(a.pd.Ntotpi, N, cst->q, cst->sumref, cst->sumsqref, cst->t2, cst->prior);
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
  const int rows2 = 2 * ((nd + 1) >> 1);
  const int hrows = HALVES == 1 ? rows2 : ((rows2 / 2 + 1) & ~1);
#pragma unroll
  for (int hf = 0; hf < HALVES; hf++)
  {
  // ---------------- T -> LDS (the slots are dead) ----------------
  float2 *Tl = U;
  const int h0 = hf * hrows, h1 = min(rows2, h0 + hrows); // window rows of this half (all of them for HALVES = 1)
#pragma unroll
  for (int blk = 0; blk < NBLK; blk++)
  {
    const int ky = blk * 64 + lane;
    if (blk < nblk && ky < (NYQ ? H - 1 : H))
    {
#pragma unroll
      for (int d = 0; d < NRW; d++)
        if (d < nrows && (HALVES == 1 || (r0 + d >= h0 && r0 + d < h1)))
          Tl[(size_t) (r0 + d - h0) * TS + ky] = make_float2(Tr[blk][d], Ti[blk][d]);
    }
  }
  if (NYQ && lane < nrows && r0 + lane >= h0 && r0 + lane < h1)
  { // Nyquist column from k_nyquist_rows: rows -nyqWD..nyqWD of Re T[.][N/2] (the imaginary part never enters)
    const int NWQ = 2 * a.nyqWD + 1;
    const float *tq = a.tnyq + ((size_t) p * a.ldPart + oc) * NWQ;
    Tl[(size_t) (r0 + lane - h0) * TS + N / 2] = make_float2(tq[r0 + lane - mD + a.nyqWD], 0.f);
  }
  if ((nd & 1) && wave == NW - 1 && nd >= h0 && nd < h1)
    for (int c = lane; c < H; c += 64) // odd row count: the last pair's second row is empty
      Tl[(size_t) (nd - h0) * TS + c] = make_float2(0.f, 0.f);
  __syncthreads();

  W2_STAMP(1);
  // ---------------- row pass: pairs of rows, private to a wave ----------------
  const int npairs = (h1 - h0) >> 1;
  const int ppw = (npairs + NW - 1) / NW;
  const int pj0 = wave * ppw;
  const int npw = max(0, min(ppw, npairs - pj0));
  const int PPP = 64 / N1; // pairs per pass
  for (int pass0 = 0; pass0 < npw; pass0 += PPP)
  {
    const int pl = lane / N1, k1 = lane - pl * N1;
    const bool act = pl < PPP && pass0 + pl < npw;
    const int pair = pj0 + (act ? pass0 + pl : 0);
    const float2 *ra = Tl + (size_t) (2 * pair) * TS;
    const float2 *rb = ra + TS;
    const float s0 = k1 == 0 ? 0.f : 1.f; // columns 0 and N/2: imaginary parts do not enter (FFTW c2r)
    float xr[R], xi[R];
#pragma unroll
    for (int k2 = 0; k2 < R; k2++)
    {
      float2 ta, tb;
      float s;
      if (k2 < R2)
      { // ky = N1 k2 + k1 <= N/2: the stored half
        ta = ra[N1 * k2 + k1];
        tb = rb[N1 * k2 + k1];
        s = k2 == 0 ? s0 : 1.f;
      }
      else
      { // ky > N/2 (or ky = N/2 for k2 = R/2, k1 = 0): G[ky] = conj(T[N - ky]),  N - ky = N1 (R - k2) - k1
        ta = ra[N1 * (R - k2) - k1];
        tb = rb[N1 * (R - k2) - k1];
        s = k2 == R2 ? -s0 : -1.f;
      }
      // U = G_a + i G_b with Im G = s Im T
      xr[FFT_IN(k2)] = fmaf(-s, tb.y, ta.x);
      xi[FFT_IN(k2)] = fmaf(s, ta.y, tb.x);
    }
    FFT_RUN(xr, xi);
    if (act)
    {
      float2 *yp = const_cast<float2 *>(ra) + k1; // y[n][k1] over the pair's own two rows (R N1 <= 2 TS entries)
#pragma unroll
      for (int n = 0; n < R; n++)
        yp[n * N1] = make_float2(xr[FFT_OUT(n)], xi[FFT_OUT(n)]);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  W2_STAMP(2);

  // ---------------- recombination over k1 + posterior: lanes = dy ----------------
  for (int c0 = 0; c0 < nd; c0 += 64)
  {
    const int wc = min(64, nd - c0);
    const int G = 64 / wc;
    const int g = lane / wc, dyl = lane - g * wc;
    const bool act = g < G;
    const int dyi = c0 + dyl;          // sorted index of this lane's dy
    const int dy = (dyi - mD) * gs;    // pixels
    const int res = ((dy % R) + R) % R;
    const int step = dy < 0 ? dy + N : dy;
    const int iyr = dinv[dyi];         // visiting rank
    // two row pairs per step: their recombination sums and the four log posteriors are independent chains that
    // overlap each other's LDS and double-precision latencies (two waves per SIMD hide little of them)
    // (measured: +4 % at +-40 px for the two-wave instantiations, -2..-4 % for the three-wave ones, which keep one pair)
#ifndef BIOEM_W2_PPS
#define BIOEM_W2_PPS 0
#endif
    constexpr int PPS = BIOEM_W2_PPS ? BIOEM_W2_PPS : (NRW * NBLK <= 26 && R <= 16) ? 1 : 2;
    // the lane's N1 - 1 recombination twiddles depend on dy only: for N1 <= 8 they are fetched once per chunk and
    // stay in registers over all row pairs (entries beyond N1 are zero and multiply a clamped, finite y)
#ifndef BIOEM_W2_K1H
#define BIOEM_W2_K1H 8
#endif
    constexpr int K1H = BIOEM_W2_K1H;
    const bool hoist = N1 <= K1H;
    float2 wk[K1H];
    if (hoist)
    {
      int idx = 0;
#pragma unroll
      for (int k1 = 1; k1 < K1H; k1++)
      {
        idx += step;
        if (idx >= N)
          idx -= N;
        wk[k1] = k1 < N1 ? twl[idx] : make_float2(0.f, 0.f);
      }
    }
    for (int pj = act ? g : npw; pj < npw; pj += PPS * G)
    {
      const float2 *yp[PPS];
      float sr[PPS], si[PPS];
      int pairs[PPS];
      bool pv[PPS];
#pragma unroll
      for (int u = 0; u < PPS; u++)
      {
        pv[u] = pj + u * G < npw;
        pairs[u] = pj0 + (pv[u] ? pj + u * G : pj);
        yp[u] = Tl + (size_t) (2 * pairs[u]) * TS + res * N1;
        const float2 y = yp[u][0];
        sr[u] = y.x; // k1 = 0: twiddle 1
        si[u] = y.y;
      }
      if (hoist)
      {
        float2 yv[PPS][K1H];
#pragma unroll
        for (int k1 = 1; k1 < K1H; k1++)
#pragma unroll
          for (int u = 0; u < PPS; u++)
            yv[u][k1] = yp[u][min(k1, N1 - 1)];
#pragma unroll
        for (int k1 = 1; k1 < K1H; k1++)
#pragma unroll
          for (int u = 0; u < PPS; u++)
          {
            sr[u] = fmaf(yv[u][k1].x, wk[k1].x, sr[u]);
            sr[u] = fmaf(-yv[u][k1].y, wk[k1].y, sr[u]);
            si[u] = fmaf(yv[u][k1].x, wk[k1].y, si[u]);
            si[u] = fmaf(yv[u][k1].y, wk[k1].x, si[u]);
          }
      }
      else
      {
        int idx = 0;
#pragma unroll 2
        for (int k1 = 1; k1 < N1; k1++)
        {
          idx += step;
          if (idx >= N)
            idx -= N;
          const float2 w = twl[idx];
#pragma unroll
          for (int u = 0; u < PPS; u++)
          {
            const float2 y = yp[u][k1];
            sr[u] = fmaf(y.x, w.x, sr[u]);
            sr[u] = fmaf(-y.y, w.y, sr[u]);
            si[u] = fmaf(y.x, w.y, si[u]);
            si[u] = fmaf(y.y, w.x, si[u]);
          }
        }
      }
      W2_STAMP(11);
      float accv[2 * PPS];
      int idv[2 * PPS];
      bool okv[2 * PPS];
#pragma unroll
      for (int v = 0; v < 2 * PPS; v++)
      {
        accv[v] = (v & 1) ? si[v >> 1] : sr[v >> 1];
        const int m = h0 + 2 * pairs[v >> 1] + (v & 1);
        okv[v] = pv[v >> 1] && m < nd;
        idv[v] = dinv[min(m, nd - 1)] * nd + iyr;
      }
      posterior_batch<2 * PPS>(L, accv, idv, okv, pw, ltab, a.algo);
      W2_STAMP(12);
    }
  }
    if (hf + 1 < HALVES)
      __syncthreads(); // every wave is done with this half's rows before the next half overwrites them
  }
  W2_STAMP(3);
  lsef_wave_reduce(L);
  if (lane == 0)
    lsew[wave] = L;
  __syncthreads();
  W2_STAMP(4);
  if (threadIdx.x == 0)
  {
    LseF Z = lsew[0];
    for (int w = 1; w < NW; w++)
    {
      const LseF o = lsew[w];
      if (o.m > Z.m || (o.m == Z.m && o.id < Z.id))
      {
        const double sc = (Z.m == -INFINITY) ? 0. : Z.s * exp_fast_nonpos((double) Z.m - (double) o.m);
        Z.s = sc + o.s;
        Z.m = o.m;
        Z.id = o.id;
        Z.val = o.val;
      }
      else
      {
        const double sc = (o.m == -INFINITY) ? 0. : o.s * exp_fast_nonpos((double) o.m - (double) Z.m);
        Z.s += sc;
      }
    }
    Partial r;
    r.sumExp = Z.s;
    r.best = Z.m;
    r.id = Z.id;
    r.value = Z.val;
    r.pad = 0;
    a.partials[(size_t) p * a.ldPart + oc] = r;
  }
}

} // namespace

#endif
