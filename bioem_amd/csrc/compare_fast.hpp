// compare_fast.hpp -- fast comparison kernel (output-pruned 2-D transform fused with the posterior) + Nyquist rows
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
#ifndef BIOEM_COMPARE_FAST_HPP
#define BIOEM_COMPARE_FAST_HPP

namespace
{

// wave-uniform table reads through the constant address space: always scalar loads (the uniform global loads of this
// kernel were otherwise emitted as vector loads, one L2 round trip each)
typedef const float2 __attribute__((address_space(4))) *const_float2_ptr;
__device__ __forceinline__ const_float2_ptr as_constant(const float2 *p)
{
  return (const_float2_ptr) (unsigned long long) p;
}

// A wave-uniform pointer, provably so for the compiler: the 64-bit address arithmetic behind it (base + index * size)
// is done on the vector unit, and a buffer descriptor built from a value the compiler keeps in vector registers makes
// EVERY buffer load a waterfall loop (4 v_readfirstlane + 2 v_cmp + exec juggling per load; found in all
// k_compare_wide2 instantiations of round 2).  Two v_readfirstlane here put it into scalar registers once.
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
  const unsigned long long v = (unsigned long long) p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
  return (T *) (((unsigned long long) hi << 32) | lo);
}

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

// ------------------------------------------------------------------------------------------------
// Posterior of a BATCH of B displacements of one lane (bioem_algorithm.h:18-123), written so that the B evaluations are
// independent straight-line chains the scheduler can interleave (one basic block):
//   * cc = acc / N^2 exactly as the reference divides (bioem_algorithm.h:163-164), but as q = a*r, q' = q + r*(a - q*b)
//     with r = RN(1/b): correctly rounded for a correctly rounded reciprocal (Markstein), 3 instructions instead of the
//     11 of the generic IEEE division sequence;
//   * the log-table lookups of all B values are issued together; the exact slow path (non-positive / non-finite
//     argument) is ONE wave-uniform branch per batch instead of a call site per value;
//   * log-sum-exp per batch: the batch maximum first (first visited displacement wins among equal float maxima, as in
//     lsef_push), ONE rescale of the running sum, then B independent exponentials.
// ------------------------------------------------------------------------------------------------
struct PostW
{
  float Np, nn, rnn; // Ntotpi, N^2 and RN(1 / N^2)
  float k0, k1, k2, k3; // sumsqref*sumsquareC, 2*sumref*sumC, sumsqref*sumC*sumC, sumref*sumref*sumsquareC
  double A, t2, prior;
};

__device__ __forceinline__ PostW post_consts(float Np, int N, const bioem_hip_param5 &q, float sumref, float sumsqref,
                                             double t2, double prior)
{
  PostW w;
  w.Np = Np;
  w.nn = (float) (N * N);
  w.rnn = 1.0f / w.nn;
  // sub-expressions of bioem_algorithm.h:32-36 in the reference's association order
  w.k0 = sumsqref * q.sumsquareC;
  w.k1 = 2 * sumref * q.sumC;
  w.k2 = sumsqref * q.sumC * q.sumC;
  w.k3 = sumref * sumref * q.sumsquareC;
  w.A = (double) (3 - Np) * 0.5;
  w.t2 = t2;
  w.prior = prior;
  return w;
}

__device__ __forceinline__ float div_by_nn(float a, const PostW &w)
{
  const float q = a * w.rnn;
  const float r = fmaf(-q, w.nn, a);
  return fmaf(r, w.rnn, q);
}

template <int B>
__device__ __forceinline__ void posterior_batch(LseF &L, const float (&acc)[B], const int (&id)[B], const bool (&ok)[B],
                                                const PostW &w, const double2 *ltab, int algo)
{
  float cc[B], fe[B];
  bool bad = false;
#pragma unroll
  for (int v = 0; v < B; v++)
  {
    cc[v] = div_by_nn(acc[v], w);
    // bioem_algorithm.h:32-36, float expression in the reference's order
    const float f = w.Np * (w.k0 - cc[v] * cc[v]) + w.k1 * cc[v] - w.k2 - w.k3;
    fe[v] = ok[v] ? f : 1.f;
    bad = bad || !(fe[v] > 1.1754944e-38f) || __float_as_uint(fe[v]) >= 0x7f800000u;
  }
  double lg[B];
  if (__builtin_expect(__any(bad), 0))
  { // some lane of the wave holds a non-positive / non-finite argument: the exact per-value path
#pragma unroll
    for (int v = 0; v < B; v++)
      lg[v] = log_of_float(fe[v], ltab);
  }
  else
  {
    double2 t[B];
#pragma unroll
    for (int v = 0; v < B; v++)
      t[v] = ltab[(__float_as_uint(fe[v]) >> 17) & 63];
#pragma unroll
    for (int v = 0; v < B; v++)
    { // same arithmetic as log_of_float
      const unsigned int bits = __float_as_uint(fe[v]);
      const int e = (int) (bits >> 23) - 127;
      const float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
      const double r = fma((double) m, t[v].x, -1.0);
      double p = fma(r, -1.0 / 6.0, 1.0 / 5.0);
      p = fma(r, p, -1.0 / 4.0);
      p = fma(r, p, 1.0 / 3.0);
      p = fma(r, p, -1.0 / 2.0);
      p = fma(r * r, p, r);
      lg[v] = fma((double) e, 0.693147180559945309417232, t[v].y + p);
    }
  }
  float lpf[B];
  double lpe[B];
#pragma unroll
  for (int v = 0; v < B; v++)
  {
    double lp = w.A * lg[v] + w.t2;
    lp -= w.prior;
    lpf[v] = ok[v] ? (float) lp : -INFINITY;
    lpe[v] = (algo == 1) ? (double) lpf[v] : lp;
  }
  // batch maximum; among equal maxima the first VISITED displacement (smallest id) wins
  float mb = lpf[0], vb = cc[0];
  int ib = ok[0] ? id[0] : 0x7fffffff;
#pragma unroll
  for (int v = 1; v < B; v++)
  {
    const bool t = ok[v] && (lpf[v] > mb || (lpf[v] == mb && id[v] < ib));
    mb = t ? lpf[v] : mb;
    ib = t ? id[v] : ib;
    vb = t ? cc[v] : vb;
  }
  const bool take = mb > L.m || (mb == L.m && ib < L.id);
  const float newm = fmaxf(L.m, mb);
  // running sum rescaled once (exp(-inf) = 0 restarts an empty sum; equal maxima: factor 1 without an exponential)
  double sum = (L.m == newm) ? L.s : L.s * exp_fast_nonpos((double) L.m - (double) newm);
#pragma unroll
  for (int v = 0; v < B; v++)
  {
    const double e = exp_fast_nonpos(lpe[v] - (double) newm);
    sum += ok[v] ? e : 0.;
  }
  L.s = sum;
  L.m = newm;
  L.id = take ? ib : L.id;
  L.val = take ? vb : L.val;
}

// block -> (particle, group of 4 orientation*CTF).  Workgroups go round-robin over the 8 XCDs (each with its own L2).
//   a.pchunk > 0: particle chunks of a.pchunk; inside a chunk the particle index runs fastest, then the group.  With a
//     chunk size that is a multiple of 8 a particle always lands on the same XCD, and the ~96 blocks resident per XCD
//     cover (pchunk/8 particles) x (a few groups): every particle line is then shared through that XCD's L2 by several
//     groups and every conv line by pchunk/8 particles, instead of each particle line being fetched from Infinity
//     Cache/HBM once per group.
//   a.pchunk < 0 (few particles): group g goes to XCD g % 8 with ALL its particles (grid padded to a multiple of 8
//     groups).  With 20 particles the chunk order would spread the 20 blocks of a group over all 8 XCDs, i.e. fetch
//     every conv line into 8 L2s for 2.5 blocks each (measured: 8 GB per launch from the fabric, 33 M/s); the particle
//     spectra are few enough to sit in every L2.
__device__ __forceinline__ bool fast_block_pair(const CompareArgs &a, int &p, int &ocg)
{
  const int ocGroups = (a.nOC + 3) >> 2;
  if (a.pchunk < 0)
  {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    ocg = (q / a.nMaps) * 8 + x;
    p = q % a.nMaps;
    return ocg < ocGroups;
  }
  const int per = a.pchunk * ocGroups;
  int c = blockIdx.x / per;
  const int nch = (a.nMaps + a.pchunk - 1) / a.pchunk;
  c = min(c, nch - 1);
  const int rem = blockIdx.x - c * per;
  const int pc = min(a.pchunk, a.nMaps - c * a.pchunk);
  ocg = rem / pc;
  p = c * a.pchunk + (rem - ocg * pc);
  return true;
}

// Window accumulation over one block of 64 frequency columns held in LDS as Tl[row = dx + WD][64] float2
// (already weighted by 1 or 2 per column; zero beyond H).  lane = (iy, group); a group owns `nr` consecutive
// displacement rows so that each LDS twiddle read E[ky*dy] feeds nr accumulators; T is read two columns
// at a time (ds_read_b128).  STATIC: nr == NR known at compile time (the +-10 px, grid 1 case).
// NP = number of column pairs: 32 for a block, 1 for the Nyquist column parked in the pad columns 64/65.
// npairs <= NP: column pairs that hold columns of the spectrum (the last block of a size like 224, 160 is not full; the
// columns beyond are zeros whose terms leave every accumulator unchanged, so skipping them is bit-identical).
template <int NR, bool STATIC, int NP, int TS>
__device__ __forceinline__ void window_accumulate(const float2 *Tl, const float2 *twl, int N, int step, int idx0,
                                                  const int (&rowoff)[NR], int nr, float (&acc)[NR], int npairs = NP)
{
  int idx = idx0;
#pragma unroll 2
  for (int kp = 0; kp < npairs; kp++)
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
// register FFT: radix-2 decimation in time with 6-op butterflies for power-of-two lengths, mixed radix otherwise;
// inputs go to the (bit- or digit-) reversed position, outputs come out in natural order
#define FFT_IN(k) (is_pow2(R) ? bitrevR<R>(k) : DIGITREV<R>.pos[k])
#define FFT_OUT(n) (n)
#define FFT_RUN(xr, xi)                                                                                            \
  do                                                                                                               \
  {                                                                                                                \
    if constexpr (is_pow2(R))                                                                                      \
      fft_inverse_dit<R>(xr, xi);                                                                                  \
    else                                                                                                           \
      fft_inverse_mixed<R>(xr, xi);                                                                                \
  } while (0)
#ifndef BIOEM_MASK_IDLE_COLUMNS
#define BIOEM_MASK_IDLE_COLUMNS 1
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
#ifndef BIOEM_FAST_HALVES
#define BIOEM_FAST_HALVES 0
#endif
// NYQ (N/2 a multiple of 64, e.g. 128 and 256): the half spectrum has N/2 + 1 columns, one more than fills the
// 64-lane column blocks, and a whole extra block pass for that single Nyquist column would cost 1/2 (128) or 1/3
// (256) of the kernel.  Instead k_nyquist_rows (below) forms the 2*WD+1 column-transform outputs of that column
// for every comparison of the launch by direct summation, and this kernel adds (-1)^dy * Re T[dx][N/2] to its
// window sums (FFTW c2r convention: weight 1, real part only).  The tail is deliberately tiny: anything larger
// (an inlined or called summation) pushes the register allocation of the main loop into scratch.
// GS (1..4): row stride of the window in pixels.  T row m (-WD..WD) holds displacement dx = m*GS, so a coarse
// DISPLACE_CENTER grid whose offsets are all multiples of GS reaches +-15*GS pixels with the same 2*WD+1 rows.
// (27- and 31-row windows: k_compare_fastm, compare_fastm.hpp)
template <int WD, int R, bool NYQ, int GS>
__global__ __launch_bounds__(256, BIOEM_FAST_WAVES_PER_SIMD) void k_compare_fast(const CompareArgs a)
{
  static_assert(WD <= 10, "windows of at most 21 rows; 27 / 31 rows: k_compare_fastm");
  constexpr int NW = 2 * WD + 1;
  constexpr int R2 = R / 2;            // rows (k2 pairs) per k1 step
  // depth of the operand ring: must divide R2 so that a ring slot is a compile-time function of the k2 pair
  // (R = 30: a ring of 3, not 5 -- 16 registers the 21-row window of that length needs)
#ifndef BIOEM_FAST_RING
#define BIOEM_FAST_RING 0
#endif
  constexpr int RD = BIOEM_FAST_RING ? BIOEM_FAST_RING : (R2 % 4 == 0) ? 4 : (R2 == 15) ? 3 : (R2 % 5 == 0) ? 5 : (R2 % 3 == 0) ? 3 : (R2 % 2 == 0) ? 2 : 1;
  constexpr int NR = (WD <= 5) ? 3 : 7; // accumulators (window rows) per lane
  // T row stride in float2 (64 or 32 columns + 2 pad: row groups land on different banks)
  constexpr int TS = BIOEM_FAST_HALVES ? 34 : 66;
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
  if (!fast_block_pair(a, p, ocg))
    return;
  const int oc_raw = ocg * 4 + wave;
  const bool oc_valid = oc_raw < a.nOC;
  const int oc = oc_valid ? oc_raw : a.nOC - 1;
  const int Hp = a.Hp; // row-pair pitch in 16-byte words (H, or H + 15: comparison_pitch in bioem_hip.hip)
  const size_t M = (size_t) N * Hp;
  // buffer descriptors built from wave-uniform values only (blockIdx / readfirstlane'd wave id)
  // timing-only ablation builds (never shipped): a zero-record descriptor drops the loads of one operand while
  // the instruction stream and waits stay (cdna_hip_programming.md, profiling: pricing one buffer's traffic)
#ifndef BIOEM_ABLATE_F
#define BIOEM_ABLATE_F 0
#endif
#ifndef BIOEM_ABLATE_C
#define BIOEM_ABLATE_C 0
#endif
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.ref + (size_t) p * M)), 0,
                                                       BIOEM_ABLATE_F ? 0 : (int) (M * sizeof(float2)), 0x00020000);
  const auto rsrcC = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<float2 *>(a.conv + (size_t) oc * M)), 0,
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

  const int nblk = NYQ ? (H - 1 + 63) / 64 : (H + 63) / 64; // (Nyquist split: whole blocks, or a last one of 32 columns)
  // Operand stream (software pipelined across k1 iterations AND column blocks): the (k1, k2-pair) loads of a
  // lane walk t = k1*16 + k2p with a constant stride of H float4; a 4-deep ring of (F, C) pairs keeps 8 dwordx4
  // loads (8 KiB per wave) in flight, re-issued as soon as a slot is consumed.  The ring runs on into the first
  // rows of the NEXT column block, so those loads fly during the T exchange / window phase of this block.
  // Addressing: buffer loads -- 128-bit descriptor (SGPRs), one 32-bit lane offset (VGPR), row offset in an SGPR.
  const unsigned rowbytes = (unsigned) Hp * 16u;
  // Split last block (a.split: it holds at most 32 columns, e.g. 17 of the 81 at 160^2): lane l and lane l + 32 take the
  // SAME column, the low half the k1 steps 0 .. sHalf - 1, the high half sHalf .. N1 - 1 (one past the end with an odd
  // N1: those rows lie beyond the buffer and read as zeros).  Inside the loop both halves use the low half's
  // recombination twiddles -- wave-uniform as everywhere --, w^(dx (k1 + sHalf)) = w^(dx k1) w^(dx sHalf): the high half's
  // sums are turned by w^(dx sHalf) once, after the loop, and the halves added with v_permlane32_swap.  The pass then
  // costs sHalf instead of N1 steps: 1.5 instead of 2 passes at 160^2.
  constexpr bool SPLIT_OK = !(NYQ && R == 32); // (the 32-point Nyquist kernels have no registers left for it)
  const int sHalf = (N1 + 1) >> 1;
  const int hsel = lane >> 5;
  const unsigned halfoff = (unsigned) (hsel * sHalf * R2) * rowbytes;
  auto lane_offset = [&](int b) -> unsigned { // byte offset of this lane's column (and half) in column block b
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
#if BIOEM_MASK_IDLE_COLUMNS
    // lanes beyond the last column of the last block sit out the whole transform (EXEC masked once): their
    // loads would only burn vector-memory cycles, and their T columns stay zero
    if (ky < H)
#endif
    for (int k1 = 0; k1 < n1; k1++)
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
        // (only the last RD requests of a k1 step can leave the block: the first test folds at compile time)
        if (k2p + RD >= R2 && tn >= ttotal)
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
    if (split)
    {
      const float2 *ws = a.twk + (size_t) sHalf * NW; // w^(dx sHalf), the table's row sHalf
#pragma unroll
      for (int d = 0; d < NW; d++)
      {
        const float2 w = ws[d];
        const float wx = hsel ? w.x : 1.f, wy = hsel ? w.y : 0.f;
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
    // T block of THIS wave only: LDS operations of one wave execute in order, so a wave-level fence (no
    // s_barrier) is enough; BIOEM_BLOCK_BARRIER=1 restores block barriers (keeps the 4 waves in lock-step)
#if BIOEM_FAST_HALVES
    for (int hx = 0; hx < 2; hx++)
    {
      const int npairs = min(16, ((H - blk * 64 + 1) >> 1) - 16 * hx);
      if (npairs <= 0)
        break;
      WAVE_OR_BLOCK_SYNC(); // previous window reads are done
      if (hsel == hx)
      {
#pragma unroll
        for (int d = 0; d < NW; d++)
          Tl[d * TS + (lane & 31)] = make_float2(Tr[d] * wgt, Ti[d] * wgt);
      }
      WAVE_OR_BLOCK_SYNC();
      const int idx0 = (int) (((long long) (blk * 64 + 32 * hx) * step) % N);
      if (is_static)
      {
        const int rowoff[NR] = {rowbase};
        window_accumulate<NR, true, 16, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
      }
      else
      {
        int rowoff[NR];
#pragma unroll
        for (int r = 0; r < NR; r++)
          rowoff[r] = row_of(r);
        window_accumulate<NR, false, 16, TS>(Tl, twl, N, step, idx0, rowoff, nr, acc, npairs);
      }
    }
#else
    {
      WAVE_OR_BLOCK_SYNC(); // previous window reads are done
#pragma unroll
      for (int d = 0; d < NW; d++)
        Tl[d * TS + lane] = make_float2(Tr[d] * wgt, Ti[d] * wgt);
      WAVE_OR_BLOCK_SYNC();
      const int idx0 = (int) (((long long) (blk * 64) * step) % N);
#if BIOEM_MASK_IDLE_COLUMNS
      const int npairs = min(32, (H - blk * 64 + 1) >> 1); // columns of this block that exist, in pairs
#else
      const int npairs = 32;
#endif
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
#endif
  }
  if (NYQ)
  {
    const float *tq = a.tnyq + ((size_t) p * a.ldPart + oc) * NW;
    const float sg = (dy & 1) ? -1.f : 1.f;
#pragma unroll
    for (int r = 0; r < NR; r++)
      acc[r] = fmaf(sg, tq[is_static ? rowbase / TS + r : row_of(r) / TS], acc[r]);
  }

  const double2 pc = a.postc[oc];
  const PostW pw = post_consts(a.pd.Ntotpi, N, a.params[oc], a.sumRef[p], a.sumsqRef[p], pc.x, pc.y);
  LseF L;
  L.m = -INFINITY;
  L.s = 0.;
  L.id = 0x7fffffff;
  L.val = 0.f;
  {
    // the NR displacements of this lane as one batch (posterior_batch: exact division by N^2 in three instructions,
    // the log-table reads issued together, one log-sum-exp rescale)
    constexpr int PB = NR > 8 ? 8 : NR;
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += PB)
    {
      float accv[PB];
      int idv[PB];
      bool okv[PB];
#pragma unroll
      for (int j = 0; j < PB; j++)
      {
        const int r = r0 + j < NR ? r0 + j : NR - 1;
        const int ixs = grp * nr + r; // position in the lane-group order; ix = visiting rank of that displacement
        okv[j] = r0 + j < NR && r < nr && wactive && ixs < a.ndx && iy < a.ndy;
        const int ix = is_static ? dinv[min(ixs, 31)] : ixs;
        accv[j] = acc[r];
        idv[j] = ix * nd + iy;
      }
      posterior_batch<PB>(L, accv, idv, okv, pw, ltab, a.algo);
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
// The twiddles are uniform over the block and tabulated per row pair (twnyq): wide scalar loads.
// ------------------------------------------------------------------------------------------------
// Q = 4 (few particles: 16 x 16 pairs per block are 192 blocks for 10 particles x 3 072 spectra, every thread a chain of
// N/2 row pairs): the four waves of a block share its 4 x 16 pairs, wave w the w-th quarter of the row pairs in order
// (the twiddles stay wave-uniform), the quarters added through LDS as (q0 + q1) + (q2 + q3).
template <int WD, int Q>
__global__ __launch_bounds__(256) void k_nyquist_rows(const CompareArgs a)
{
  static_assert(Q == 1 || Q == 4, "one thread per pair, or one per pair and wave");
  constexpr int NW = 2 * WD + 1;
  __shared__ float part[Q == 4 ? 2 * 64 * NW : 1];
  const int N = a.N, H = a.H, N1 = a.N1;
  const int R2 = N / (2 * N1);
  const int tilesOC = (a.nOC + 15) / 16;
  const int tp = blockIdx.x / tilesOC, to = blockIdx.x - tp * tilesOC;
  const int q = Q == 1 ? 0 : __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
  const int t = Q == 1 ? threadIdx.x : threadIdx.x & 63;
  const int p = tp * (16 / Q) + (t >> 4), oc = to * 16 + (t & 15);
  const bool valid = p < a.nMaps && oc < a.nOC;
  const size_t M = (size_t) N * a.Hp; // an image of the comparison layout (Hp >= H: its row-pair pitch)
  const float2 *F = a.ref + (size_t) (valid ? p : 0) * M;
  const float2 *C = a.conv + (size_t) (valid ? oc : 0) * M;
  float acc[NW];
#pragma unroll
  for (int d = 0; d < NW; d++)
    acc[d] = 0.f;
  const int nRP = N1 * R2 / Q; // row pairs of this thread: [q nRP, (q + 1) nRP)
  // row pair = (k1, k2 pair): kx0 = N1*(2 k2p) + k1, kx1 = kx0 + N1.  Eight pairs' operands are fetched at once: every
  // load of a thread is a cache line of its own (one particle, one conv spectrum), and with one pair in flight the
  // kernel was a chain of N/2 memory latencies (58 us for 30 720 comparisons at 128^2); the sums keep their order.
  constexpr int NB = WD > 31 ? 4 : 8; // N/2 is a multiple of 64, a quarter of it of 16
  for (int rp0 = q * nRP; rp0 < (q + 1) * nRP; rp0 += NB)
  {
    float4 cb[NB], fb[NB];
#pragma unroll
    for (int u = 0; u < NB; u++)
    {
      // the two k2 of a pair are adjacent in the comparison layout: one 16-byte load per operand
      const size_t li = ((size_t) (rp0 + u) * a.Hp + N / 2) * 2;
      cb[u] = *reinterpret_cast<const float4 *>(C + li);
      fb[u] = *reinterpret_cast<const float4 *>(F + li);
    }
#pragma unroll
    for (int u = 0; u < NB; u++)
    {
      const float4 c = cb[u], f = fb[u];
      // X = conv * conj(ref)   (bioem.cpp:1452-1455)
      const float x0r = fmaf(c.x, f.x, c.y * f.y), x0i = fmaf(c.y, f.x, -(c.x * f.y));
      const float x1r = fmaf(c.z, f.z, c.w * f.w), x1i = fmaf(c.w, f.z, -(c.z * f.w));
      // twiddles of this row pair: block-uniform, contiguous -> a few wide scalar loads
      const float4 *tw = reinterpret_cast<const float4 *>(a.twnyq + (size_t) (rp0 + u) * NW * 2);
#pragma unroll
      for (int d = 0; d < NW; d++)
      {
        const float4 w = tw[d]; // (w0.re, w0.im, w1.re, w1.im)
        float v = acc[d];
        v = fmaf(x0r, w.x, v);
        v = fmaf(-x0i, w.y, v);
        v = fmaf(x1r, w.z, v);
        v = fmaf(-x1i, w.w, v);
        acc[d] = v;
      }
    }
  }
  if (Q == 4)
  {
    // (q0 + q1) in wave 0, (q2 + q3) in wave 2, then their sum in wave 0
    float *mine = part + ((q >> 1) * 64 + t) * NW;
    if (q & 1)
    {
#pragma unroll
      for (int d = 0; d < NW; d++)
        mine[d] = acc[d];
    }
    __syncthreads();
    if (!(q & 1))
    {
#pragma unroll
      for (int d = 0; d < NW; d++)
        acc[d] += mine[d];
    }
    __syncthreads();
    if (q == 2)
    {
#pragma unroll
      for (int d = 0; d < NW; d++)
        mine[d] = acc[d];
    }
    __syncthreads();
    if (q == 0)
    {
#pragma unroll
      for (int d = 0; d < NW; d++)
        acc[d] += part[(64 + t) * NW + d];
    }
  }
  if (valid && q == 0)
  {
    float *o = a.tnyq + ((size_t) p * a.ldPart + oc) * NW;
#pragma unroll
    for (int d = 0; d < NW; d++)
      o[d] = acc[d];
  }
}

} // namespace

#endif
