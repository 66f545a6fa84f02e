// kernel_select.hpp -- which comparison kernel runs a given (image size, displacement set): ONE place
// Part of libbioem_hip.so; included by bioem_hip.hip only (one translation unit, anonymous namespace).
//
//   kernel_table.inc   the instantiations that exist (one line each; scripts/make_kernel_table.py writes it from a
//                      selection snapshot, scripts/check_kernel_coverage.py holds it against what the GPU tests ran)
//   find_kernel        registry lookup; a combination that is not in the table is simply not available and the
//                      planner moves on to its next candidate -- removing a line can cost speed, never correctness
//   kWide2Rules        the k_compare_wide2 variants in order of preference, each with the measurement that put it there
//   plan_kernels       pure function (no HIP call): displacement list -> KernelPlan; bioem_hip_plan exposes it, so the
//                      selection is testable without a GPU (tests/test_selection_table.py)
//
// Environment knobs of the selection (experiments; none is needed for production runs):
//   BIOEM_NO_WIDE2        wide windows on tiles of the 21/27/31-row kernels instead of k_compare_wide2
//   BIOEM_FORCE_WIDE2     k_compare_wide2 from 21 rows on (tests: every instantiation on small shapes)
//   BIOEM_W2_R=<len>      register-FFT length of k_compare_wide2 where it divides N
//   BIOEM_NO_TILES        no window tiles: what does not fit a kernel runs k_compare_generic
//   BIOEM_TILE_ROWS=<t>   tile size of the tiled path (21, 27, 31)
//   BIOEM_NO_FASTM2       33..47-row windows on k_compare_wide2 (k_compare_fastm2 off)
//   BIOEM_NO_FASTM        27/31-row windows on tiles of the 21-row kernel (k_compare_fastm off)
//   BIOEM_NO_ROWS_KERNEL  odd sizes on k_compare_generic
//   BIOEM_NO_ODD_FFT      odd sizes on k_compare_rows (direct column sums) even where an odd register FFT divides N
//   BIOEM_POW2_FFT        power-of-two register FFTs only
//   BIOEM_LONGEST_FFT     k_compare_fast with the longest register FFT at every size (round 3's rule)
//   BIOEM_FAST_R=<len>    k_compare_fast with this register-FFT length only (sweeps)
//   BIOEM_KEEP_WD5        11-row windows keep the 11-row template at every size
// (run-time knobs outside the selection: BIOEM_PCHUNK, BIOEM_BATCH_ORIENTATIONS, BIOEM_FIXED_BATCH, BIOEM_NO_GROUP_XCD,
//  BIOEM_COMPAT_RING, BIOEM_SERIAL_FOLD, BIOEM_SIGNATURE_LOG, BIOEM_HIP_LIBRARY (Python loader), BIOEM_NO_SPLIT_LAST (no
//  half-wave split of a narrow last column block), BIOEM_NO_PITCH_PAD / BIOEM_PITCH_PAD=<words> (row-pair pitch of the
//  comparison layout), BIOEM_MIN_BATCH_PAIRS, BIOEM_CONVOLVE_FUSED)
#ifndef BIOEM_KERNEL_SELECT_HPP
#define BIOEM_KERNEL_SELECT_HPP

#ifndef BIOEM_NYQUIST_SPLIT
#define BIOEM_NYQUIST_SPLIT 1
#endif

namespace
{

typedef void (*fast_kernel_t)(const CompareArgs);

// ------------------------------------------------------------------------------------------------
// registry: the family tables live in their own translation units (kernels_*.hip, engine_types.hpp); an experiment
// build (BIOEM_SLIM, scripts/slim_build.sh) is this one file with the instantiations named on the command line
// ------------------------------------------------------------------------------------------------
typedef BioemKernelEntry KernelEntry;

#ifdef BIOEM_SLIM
const KernelEntry kSlimTable[] = {
#ifdef BIOEM_SLIM_FAST
    {KF_FAST, {BIOEM_SLIM_FAST, 0, 0}, reinterpret_cast<const void *>(k_compare_fast<BIOEM_SLIM_FAST>)},
#endif
#ifdef BIOEM_SLIM_FASTM
    {KF_FASTM, {BIOEM_SLIM_FASTM, 0, 0}, reinterpret_cast<const void *>(k_compare_fastm<BIOEM_SLIM_FASTM>)},
#endif
#ifdef BIOEM_SLIM_FASTM2
    {KF_FASTM2, {BIOEM_SLIM_FASTM2, 0, 0, 0}, reinterpret_cast<const void *>(k_compare_fastm2<BIOEM_SLIM_FASTM2>)},
#endif
#ifdef BIOEM_SLIM_W2
    {KF_WIDE2, {BIOEM_SLIM_W2}, reinterpret_cast<const void *>(k_compare_wide2<BIOEM_SLIM_W2>)},
#endif
    {KF_GENERIC, {0, 0, 0, 0, 0, 0}, reinterpret_cast<const void *>(k_compare_generic)}};
#endif

fast_kernel_t find_kernel(int family, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0, int a4 = 0, int a5 = 0)
{
  if (family == KF_GENERIC)
    return k_compare_generic;
  const KernelEntry *tabs[8];
  int cnt[8], nt = 0;
#ifdef BIOEM_SLIM
  tabs[nt] = kSlimTable;
  cnt[nt++] = (int) (sizeof(kSlimTable) / sizeof(kSlimTable[0]));
#else
  switch (family)
  {
  case KF_FAST: tabs[nt] = bioem_kernels_fast(&cnt[nt]); nt++; break;
  case KF_FASTM: tabs[nt] = bioem_kernels_fastm(&cnt[nt]); nt++; break;
  case KF_FASTM2: tabs[nt] = bioem_kernels_fastm2(&cnt[nt]); nt++; break;
  case KF_WIDE2:
    tabs[nt] = bioem_kernels_wide2_short(&cnt[nt]); nt++;
    tabs[nt] = bioem_kernels_wide2_16(&cnt[nt]); nt++;
    tabs[nt] = bioem_kernels_wide2_long(&cnt[nt]); nt++;
    break;
  default: tabs[nt] = bioem_kernels_odd(&cnt[nt]); nt++; break;
  }
#endif
  for (int t = 0; t < nt; t++)
    for (int i = 0; i < cnt[t]; i++)
    {
      const KernelEntry &e = tabs[t][i];
      if (e.fn && e.family == family && e.a[0] == a0 && e.a[1] == a1 && e.a[2] == a2 && e.a[3] == a3 && e.a[4] == a4 &&
          e.a[5] == a5)
        return reinterpret_cast<fast_kernel_t>(const_cast<void *>(e.fn));
    }
  return nullptr;
}

// ------------------------------------------------------------------------------------------------
// LDS budgets (bytes of dynamic shared memory per block)
// ------------------------------------------------------------------------------------------------
size_t compare_lds_bytes(int N, int H, int NW, int waves, int rows = 0)
{ // generic kernel: tables + per-wave T [rows of a group][Hs] (rows = 0: all NW)
  const int Hs = (H + 1) & ~1;
  const size_t dispBytes = ((size_t) NW * 4 + 255) & ~(size_t) 255;
  return (size_t) ((N + 2) & ~1) * 8 + dispBytes + (size_t) waves * (rows ? rows : NW) * Hs * 8;
}
#ifndef BIOEM_FAST_HALVES
#define BIOEM_FAST_HALVES 0
#endif
size_t fast_lds_bytes(int N, int NW, int waves, bool half)
{ // fast / rows kernels: twiddles + displacement list + log table + per-wave T block [NW][66] ([NW][34] half exchange)
  return (size_t) ((N + 2) & ~1) * 8 + 256 + 1024 + (size_t) waves * NW * (half ? 34 : 66) * 8;
}
size_t fastm_lds_bytes(int N)
{ // cos / sin planes of the twiddle table (padded), window ranks, log table, per wave two 32 x 33 float planes and the
  // resting place of the 16 tile accumulators
  return (size_t) 2 * fastm_table_floats(N) * 4 + 128 + 1024 + (size_t) 4 * 2 * 32 * 33 * 4 + (size_t) 4 * 16 * 64 * 4;
}
size_t fastm2_lds_bytes()
{ // window ranks, log table, per wave the two 47 x 33 float planes / the resting place of the 36 tile accumulators
  // (+ 256 B: the operand read of plane row 47, which holds no window row, may run past the last wave's planes)
  return (size_t) 256 + 1024 + (size_t) 4 * kFm2WaveFloats * 4 + 256;
}
size_t wide2_lds_bytes(int N, int R, int rows2, int ts, int nw = 4)
{ // tables (twiddles, visiting ranks, log table, wave results, posterior constants) + max(one FFT-output slot per wave, T block)
  const size_t slots = (size_t) nw * R * 64 * 8, tblock = (size_t) rows2 * ts * 8;
  return (size_t) ((N + 2) & ~1) * 8 + 512 + 1024 + 256 + std::max(slots, tblock);
}
constexpr size_t kLdsCU = 160 * 1024; // LDS of a CU

// ------------------------------------------------------------------------------------------------
// the plan
// ------------------------------------------------------------------------------------------------
struct KernelPlan
{
  std::vector<int> disp; // displacement list per axis in the reference's visiting order
  int nd = 0, gs = 1, winD = 0;
  int family = KF_GENERIC;
  int fast = 0, N1 = 0, oddR = 0; // fast = register-FFT length / 2 (0: no register FFT)
  bool nyq = false, fastm = false, fastm2 = false, rowsK = false, wide2 = false;
  int tileT = 0, tilesPerAxis = 1;
  std::vector<int> tileCenter, tileValid;
  int w2NRW = 0, w2NBLK = 0, w2TS = 0, w2Rows2 = 0, nyqWD = 0, w2Halves = 1, w2NW = 4;
  int genericWaves = 4;
  int genericRows = 0; // k_compare_generic: window rows per pass through the LDS (0 = all)
  size_t ldsBytes = 0;
  fast_kernel_t fn = nullptr;
  const char *err = nullptr;
};

// register-FFT lengths: the power-of-two part of N up to `cap`, or -- where that part is only 2 or 4 (or `preferMixed`)
// -- the longest 2/3/5-smooth even divisor up to `cap` (mixed-radix register FFT; measured: 250^2 20 -> 40 M/s,
// 180^2 47 -> 53, 100^2 129 -> 143; with a power-of-two part of 8 it does not pay in k_compare_fast: 200^2, 120^2)
void fft_lengths(int N, int cap, bool allowMixed, std::vector<int> &out)
{
  int p2 = 2;
  for (int r : {32, 16, 8, 4})
    if (r <= cap && N % r == 0)
    {
      p2 = r;
      break;
    }
  if (allowMixed && p2 < 8)
    for (int r : {30, 20, 18, 12, 10, 6})
      if (r <= cap && N % r == 0 && r > p2)
        out.push_back(r);
  out.push_back(p2);
  for (int r : {16, 8, 4, 2}) // shorter power-of-two lengths as further fall-backs
    if (r < p2 && N % r == 0)
      out.push_back(r);
}

// ------------------------------------------------------------------------------------------------
// k_compare_wide2 variants, in order of preference.  A rule applies when the shape meets its limits AND the
// instantiation is in the table; the first that applies wins.  Measurements: 1 000 particles, 5 CTFs, MI355X, M/s.
// ------------------------------------------------------------------------------------------------
enum W2Length
{
  W2_LEN_THREE_WAVE, // the longest of 16 / 12 / 10 / 8 dividing N (16 only with the Nyquist split): 3 waves per SIMD
  W2_LEN_SIXTEEN,    // 16 points where 32 divide N (half the slot space, 146 registers)
  W2_LEN_EIGHT_WAVE, // 16 or 10 points, no Nyquist split
  W2_LEN_LONGEST     // 32, or the longest mixed length where the power-of-two part is 8 or less
};
struct Wide2Rule
{
  const char *name;
  int nblkLo, nblkHi; // 64-column blocks of the half spectrum
  int nw;             // waves per comparison
  int nrw;            // window rows per wave (template NRW; 0: 32 for one block, 24 / 21 / 11 by shape)
  W2Length len;
  int minRows;        // window rows from which the variant pays (against the tiled kernels)
  int blocksPerCU;    // its LDS must let this many blocks share a CU
  bool halves;        // may pass the T block through LDS in two halves to keep blocksPerCU
};
const Wide2Rule kWide2Rules[] = {
    // one column block (N <= 126, 128 with the Nyquist split), <= 21 rows per wave, 42 accumulators + a short FFT:
    // 128^2 +-40 px 21.1 -> 28.4, +-30 px 27.8 -> 45.8, 120^2 +-25 px 29.2 -> 38.6, 96^2 +-20 px 47.3 -> 56.1
    {"one block, three waves per SIMD", 1, 1, 4, 21, W2_LEN_THREE_WAVE, 32, 3, false},
    // two blocks, 32..52 rows: 224^2 +-16 px 29.8, +-20 px 20.7 -> 24.3, +-24 px 17.7 -> 22.7; 256^2 +-16 px 17.4 -> 23.2
    {"two blocks, 11 rows per wave", 2, 2, 4, 11, W2_LEN_THREE_WAVE, 32, 3, false},
    {"two blocks, 13 rows per wave", 2, 2, 4, 13, W2_LEN_THREE_WAVE, 32, 3, false},
    // 53 rows at 224^2 (+-26 px): 17.6 -> 19.8; one row more and only two blocks fit a CU, where it loses (14.3 vs 17.6)
    {"two blocks, 21 rows per wave, 16 points", 2, 2, 4, 21, W2_LEN_SIXTEEN, 43, 3, false},
    // eight waves, 110 registers, four waves per SIMD at two blocks per CU: 208^2 +-30 px 14.5 -> 17.7, 240^2 13.4 -> 16.0,
    // 176^2 +-40 px 13.4 -> 15.2, 250^2 +-30 px 10.2 -> 12.1 (10 points); loses where three four-wave blocks fit
    {"two blocks, eight waves", 2, 2, 8, 11, W2_LEN_EIGHT_WAVE, 43, 2, false},
    // the two-wave kernels: 224^2 +-40 px 7.2 (tiled) -> 14.5, 256^2 +-40 px 6.1 -> 11.1 (T block in halves), 320^2 +-40 px
    // 4.1 -> 6.8 (three blocks), 208^2 +-42 px 3.2 -> 10.3 (24 rows per wave), 512^2 +-20 px 3.3 -> 4.7 (four blocks)
    // (round 4, measured and not kept: six waves per comparison with four rotating producer slots -- 163 registers, three
    //  waves per SIMD on paper -- 224^2 +-40 px 8.7 against 15.3 M/s: a 6-wave block lands 2/1/2/1 on the four SIMDs and a
    //  second block of the same shape does not fit beside it at three waves per SIMD, so one block per CU ran)
    {"one block, 32 rows per wave", 1, 1, 4, 32, W2_LEN_LONGEST, 43, 1, false},
    {"two blocks, 21 rows per wave", 2, 2, 4, 21, W2_LEN_LONGEST, 43, 2, true},
    {"two blocks, 24 rows per wave", 2, 2, 4, 24, W2_LEN_LONGEST, 43, 2, true},
    {"three blocks, 21 rows per wave", 3, 3, 4, 21, W2_LEN_LONGEST, 32, 2, true},
    {"four blocks, 11 rows per wave", 4, 4, 4, 11, W2_LEN_LONGEST, 32, 2, true},
    // four blocks, 45..88 rows, one 512-thread block per CU: 512^2 +-40 px 1.67 -> 2.28, 448^2 +-40 px 2.48 -> 3.71
    {"four blocks, eight waves", 4, 4, 8, 11, W2_LEN_LONGEST, 45, 1, true},
    // (last resort inside the family: the two-wave kernels with ONE block per CU)
    {"two blocks, 21 rows per wave, one block per CU", 2, 2, 4, 21, W2_LEN_LONGEST, 43, 1, false},
    {"two blocks, 24 rows per wave, one block per CU", 2, 2, 4, 24, W2_LEN_LONGEST, 43, 1, false},
    {"three blocks, 21 rows per wave, one block per CU", 3, 3, 4, 21, W2_LEN_LONGEST, 32, 1, false},
    {"four blocks, 11 rows per wave, one block per CU", 4, 4, 4, 11, W2_LEN_LONGEST, 32, 1, false},
};

bool plan_wide2(KernelPlan &P, int N, int H, int mD)
{
  const int nd = P.nd;
  const bool nyq = BIOEM_NYQUIST_SPLIT && (N / 2) % 64 == 0;
  const int nblk = nyq ? (H - 1) / 64 : (H + 63) / 64;
  const int rows2 = 2 * ((nd + 1) / 2), hrows = (rows2 / 2 + 1) & ~1;
  int ts = H; // row stride = 4 mod 16 float2: the (row pair, k1) lanes of the row pass spread over the banks
  while (ts % 16 != 4)
    ts++;
  const bool force = getenv("BIOEM_FORCE_WIDE2") != nullptr;
  const int forcedR = getenv("BIOEM_W2_R") ? atoi(getenv("BIOEM_W2_R")) : 0;
  if (nd > 128 || (nyq && mD > 42))
    return false;
  for (const Wide2Rule &r : kWide2Rules)
  {
    if (nblk < r.nblkLo || nblk > r.nblkHi)
      continue;
    if (nd < r.minRows && !(force && nd >= 21))
      continue;
    if ((nd + r.nw - 1) / r.nw > r.nrw)
      continue;
    std::vector<int> lens;
    switch (r.len)
    {
    case W2_LEN_THREE_WAVE:
      for (int l : {16, 12, 10, 8})
        if (N % l == 0 && (l == 16 || !nyq))
          lens.push_back(l);
      break;
    case W2_LEN_SIXTEEN:
      if (N % 32 == 0)
        lens.push_back(16);
      break;
    case W2_LEN_EIGHT_WAVE:
      if (!nyq && N % 16 == 0 && N % 32 != 0)
        lens.push_back(16);
      else if (!nyq && N % 10 == 0 && N % 8 != 0 && N % 20 != 0 && N % 30 != 0)
        lens.push_back(10);
      break;
    case W2_LEN_LONGEST:
      if (N % 32 == 0)
      {
        lens.push_back(32);
        lens.push_back(16); // (where the 32-point instantiation of a rule is not in the table)
      }
      else
      { // the two-wave kernels run faster on the longest mixed length where the power-of-two part is 8 or less
        // (200^2 +-30 px 11.2 -> 15.3 M/s with 20 points), else on the power-of-two part
        if (!nyq && !getenv("BIOEM_POW2_FFT"))
          for (int l : {30, 20, 18, 12, 10})
            if (N % l == 0 && N % 16 != 0)
              lens.push_back(l);
        for (int l : {16, 8, 4, 2})
          if (N % l == 0)
          {
            lens.push_back(l);
            break;
          }
        if (!nyq && !getenv("BIOEM_POW2_FFT") && N % 6 == 0 && N % 4 != 0)
          lens.insert(lens.begin(), 6);
      }
      break;
    }
    if (forcedR)
    {
      lens.clear();
      if (N % forcedR == 0)
        lens.push_back(forcedR);
    }
    for (int R : lens)
    {
      if (N / R > 32)
        continue;
      const size_t budget = kLdsCU / r.blocksPerCU;
      int halves = 0;
      if (wide2_lds_bytes(N, R, rows2, ts, r.nw) <= budget)
        halves = 1;
      else if (r.halves && wide2_lds_bytes(N, R, hrows, ts, r.nw) <= budget)
        halves = 2;
      if (!halves)
        continue;
      // the eight-wave kernel wins only where three four-wave blocks of that length would NOT fit a CU
      if (r.nw == 8 && r.nblkLo == 2 && wide2_lds_bytes(N, R, rows2, ts, 4) <= kLdsCU / 3)
        continue;
      const fast_kernel_t fn = find_kernel(KF_WIDE2, R, r.nrw, nblk, nyq, halves, r.nw);
      if (!fn)
        continue;
      P.wide2 = true;
      P.family = KF_WIDE2;
      P.fn = fn;
      P.fast = R / 2;
      P.N1 = N / R;
      P.nyq = nyq;
      P.w2NBLK = nblk;
      P.w2NRW = r.nrw;
      P.w2NW = r.nw;
      P.w2Halves = halves;
      P.w2TS = ts;
      P.w2Rows2 = halves == 2 ? hrows : rows2;
      P.nyqWD = mD <= 20 ? 20 : mD <= 31 ? 31 : 42;
      if (nyq)
        P.winD = P.nyqWD; // sizes the Nyquist pre-kernel's tables
      P.ldsBytes = wide2_lds_bytes(N, R, P.w2Rows2, ts, r.nw);
      return true;
    }
  }
  return false;
}

// base kernel of a window of at most 2 winD + 1 rows (also the tile kernel of the tiled path)
bool plan_window_kernel(KernelPlan &P, int N, int H, int winD, bool untiled)
{
  const bool nyq = BIOEM_NYQUIST_SPLIT && N % 2 == 0 && (N / 2) % 64 == 0;
  if (N % 2 == 0 && N >= 8)
  {
    if (winD > 10)
    { // 27 / 31 rows: k_compare_fastm, register FFT of at most 16 points (three waves per SIMD); Nyquist split: 16
      if (getenv("BIOEM_NO_FASTM"))
        return false;
      std::vector<int> lens;
      if (nyq)
        lens.push_back(16);
      else
        fft_lengths(N, 16, P.gs == 1 && !getenv("BIOEM_POW2_FFT"), lens);
      for (int R : lens)
      {
        if (getenv("BIOEM_FAST_R") && R != atoi(getenv("BIOEM_FAST_R")))
          continue;
        if (const fast_kernel_t fn = find_kernel(KF_FASTM, winD, R, nyq, P.gs))
        {
          P.family = KF_FASTM;
          P.fastm = true;
          P.fn = fn;
          P.fast = R / 2;
          P.N1 = N / R;
          P.nyq = nyq;
          P.winD = winD;
          P.ldsBytes = fastm_lds_bytes(N);
          return true;
        }
      }
      return false;
    }
    // 64^2 and 192^2 (N / 2 = 32 mod 64): the Nyquist column by direct summation as for 128^2 / 256^2, and the 32 columns
    // that remain beyond the whole blocks as a split block (compare_fast.hpp) -- 1.5 passes instead of 2 at 192^2: +-10 px
    // 62.3 -> 63.8 M/s, +-5 px 72.4 -> 77.2.  The 32-point Nyquist kernels have no registers left for the split: 16
    // points (21 rows) / 8 points (11 rows) only, which pays at 320^2 with 21 rows still (25.5 -> 26.7 M/s) and no longer
    // with 11 rows (-3 %) or at 448^2 (-3 / -30 %).
    const bool nyq32 = BIOEM_NYQUIST_SPLIT && !nyq && (N / 2) % 64 == 32 && (N <= 256 || (N == 320 && winD == 10)) &&
                       !getenv("BIOEM_NO_SPLIT_LAST");
    if (nyq32)
    {
      const int R = winD == 5 ? 8 : 16;
      if (!(getenv("BIOEM_FAST_R") && R != atoi(getenv("BIOEM_FAST_R"))))
        if (const fast_kernel_t fn = find_kernel(KF_FAST, winD, R, true, P.gs))
        {
          P.family = KF_FAST;
          P.fn = fn;
          P.fast = R / 2;
          P.N1 = N / R;
          P.nyq = true;
          P.winD = winD;
          P.ldsBytes = fast_lds_bytes(N, 2 * winD + 1, 4, false);
          return true;
        }
    }
    std::vector<int> lens;
    if (nyq)
    { // 128^2, 256^2, ...: 32 points, and the shorter lengths that the sweep below found faster (128^2 +-5 px: 155 -> 173
      // M/s with 8 points; 128^2 / 256^2 +-10 px: +1.4 / +1.9 % with 16; 256^2 +-5 px: 16 level, 8 slower)
      lens.push_back(32);
      if (winD != 5)
        lens.push_back(16);
      else if (N <= 160)
        lens.push_back(8);
    }
    else
      fft_lengths(N, 32, P.gs == 1 && !getenv("BIOEM_POW2_FFT"), lens);
    {
      // up to 256 pixels the longest length is not the fastest (round 4, sweep of every length in the table over 64...240
      // pixels, 1 000 particles): 21-row windows 16 > 12 > 20 > 18 > 10 > 32 > 30 > 8 (224^2: 51.0 / 50.0 / 47.5 M/s
      // with 16 / 32 / 8 points, 180^2: 61.0 with 12 against 59.5 with 30), 11-row windows 8 first up to 160 pixels
      // (64^2: 271 / 259 / 228 M/s with 8 / 16 / 32), 16 first above (224^2: 57.1 / 55.8 / 55.7); from 288 pixels on
      // 32 and 16 points are level and the order stays
      if (N <= 256 && !getenv("BIOEM_LONGEST_FFT"))
      {
        static const int pref21[] = {16, 12, 20, 18, 10, 32, 30, 8, 6, 4, 2};
        static const int pref11s[] = {8, 16, 12, 10, 18, 20, 32, 30, 6, 4, 2};
        static const int pref11l[] = {16, 12, 18, 10, 8, 20, 32, 30, 6, 4, 2};
        const int *pref = winD == 5 ? (N <= 160 ? pref11s : pref11l) : pref21;
        auto rank = [&](int r) {
          for (int k = 0; k < 11; k++)
            if (pref[k] == r)
              return k;
          return 11;
        };
        std::stable_sort(lens.begin(), lens.end(), [&](int a, int b) { return rank(a) < rank(b); });
      }
    }
    for (int R : lens)
    {
      if (getenv("BIOEM_FAST_R") && R != atoi(getenv("BIOEM_FAST_R"))) // timing experiments: another register-FFT length
        continue;
      int wd = winD;
      // 11-row windows: with four column blocks and a length <= 16 (or 40+ column steps) the 21-row template is the
      // faster one (+-5 px: 432^2 11.0 -> 13.0 M/s, 360^2 16.7 -> 19.2, 400^2 13.5 -> 14.7)
      const int nblkF = nyq ? (H - 1) / 64 : (H + 63) / 64;
      if (wd == 5 && untiled && ((nblkF >= 4 && R <= 16) || N / R >= 40) && !getenv("BIOEM_KEEP_WD5") &&
          find_kernel(KF_FAST, 10, R, nyq, P.gs))
        wd = 10;
      if (const fast_kernel_t fn = find_kernel(KF_FAST, wd, R, nyq, P.gs))
      {
        P.family = KF_FAST;
        P.fn = fn;
        P.fast = R / 2;
        P.N1 = N / R;
        P.nyq = nyq;
        P.winD = wd;
        P.ldsBytes = fast_lds_bytes(N, 2 * wd + 1, 4, BIOEM_FAST_HALVES);
        return true;
      }
    }
    return false;
  }
  if (N >= 8 && !getenv("BIOEM_NO_ROWS_KERNEL"))
  { // odd N: register FFT of odd length over the reference layout where 25 / 15 / 9 / 5 / 3 divides N, else direct sums
    if (P.gs == 1 && !getenv("BIOEM_NO_ODD_FFT"))
      for (int r : {25, 15, 9, 5, 3})
        if (N % r == 0)
          if (const fast_kernel_t fn = find_kernel(KF_ODDFFT, winD, r))
          {
            P.family = KF_ODDFFT;
            P.rowsK = true;
            P.oddR = r;
            P.N1 = N / r;
            P.fn = fn;
            P.winD = winD;
            P.ldsBytes = fast_lds_bytes(N, 2 * winD + 1, 4, false);
            return true;
          }
    if (const fast_kernel_t fn = find_kernel(KF_ROWS, winD, P.gs))
    {
      P.family = KF_ROWS;
      P.rowsK = true;
      P.fn = fn;
      P.winD = winD;
      P.ldsBytes = fast_lds_bytes(N, 2 * winD + 1, 4, false);
      return true;
    }
  }
  return false;
}

KernelPlan plan_kernels(int N, int maxD, int grid, int algo)
{
  KernelPlan P;
  const int H = N / 2 + 1;
  // displacement list per axis in the reference's visiting order
  if (algo == 1)
  { // bioem_algorithm.h:156-197
    for (int c = 0; c <= maxD; c += grid)
      P.disp.push_back(c);
    for (int c = N - maxD; c < N; c += grid)
      P.disp.push_back(c - N);
  }
  else
  { // bioem.cpp:1477-1485
    const int NxDisp = 2 * (maxD / grid) + 1;
    for (int m = 0; m < NxDisp; m++)
      P.disp.push_back(m * grid - maxD);
  }
  P.nd = (int) P.disp.size();
  // window rows: row m holds displacement m * gs, gs = gcd of all offsets (1..4 are instantiated), so a coarse grid
  // with maxD a multiple of the spacing reaches +-15 gs pixels with the 31-row window
  {
    int gg = 0;
    for (int d : P.disp)
    {
      int x = d < 0 ? -d : d, y = gg;
      while (y)
      {
        const int t = x % y;
        x = y;
        y = t;
      }
      gg = x;
    }
    P.gs = (gg >= 1 && gg <= 4) ? gg : 1;
  }
  const int mD = maxD / P.gs;
  const bool symmetric = P.nd == 2 * mD + 1; // the plain set {gs m, |m| <= mD}
  // window template: 2 winD + 1 rows, nd <= rows (ALGO 1 with maxD % grid != 0 visits up to 2 (maxD / grid) + 2 offsets)
  const int winD = (mD <= 5 && P.nd <= 11) ? 5 : (mD <= 10 && P.nd <= 21) ? 10 : (mD <= 13 && P.nd <= 27) ? 13 : 15;
  const bool fitsWindow = mD <= 15 && P.nd <= 31;
  P.winD = winD;

  // 1. one window kernel
  if (fitsWindow && plan_window_kernel(P, N, H, winD, true))
    return P;
  // 2a. 33..47 rows at unit stride, N a multiple of 16 / 12 / 10 / 8: k_compare_fastm2 (one wave per comparison, rows split
  //     over the half-waves; round 4: 224^2 +-20 px 24.8 -> 30.9 M/s against k_compare_wide2<16, 11, 2>).  The
  //     longest length that divides N: the recombination costs 8 fused multiply-adds per accumulator and step whatever
  //     the length, so fewer, longer steps win (per column and unit of N: 14 / 16 / 18 / 19 instructions at 16 / 12 / 10 / 8)
  //     Row strides 2..4 (a coarse DISPLACE_CENTER grid) with the 16-point kernel: rows 4 / 8 / 2 apart pair up.
  if (N >= 64 && symmetric && P.nd >= 33 && P.nd <= 2 * kFm2WD + 1 && !getenv("BIOEM_NO_FASTM2") &&
      !getenv("BIOEM_FORCE_WIDE2"))
  {
    const bool nyq = BIOEM_NYQUIST_SPLIT && (N / 2) % 64 == 0;
    for (int R : {16, 12, 10, 8})
    {
      if (N % R != 0)
        continue;
      const fast_kernel_t fn = find_kernel(KF_FASTM2, R, nyq, P.gs);
      if (!fn)
        continue;
      P.family = KF_FASTM2;
      P.fastm2 = true;
      P.fn = fn;
      P.fast = R / 2;
      P.N1 = N / R;
      P.nyq = nyq;
      P.nyqWD = mD <= 20 ? 20 : 31;
      if (nyq)
        P.winD = P.nyqWD; // sizes the Nyquist pre-kernel's tables
      P.ldsBytes = fastm2_lds_bytes();
      return P;
    }
  }
  // 2. wide symmetric windows on even sizes: k_compare_wide2
  if (N % 2 == 0 && N >= 8 && symmetric && (P.nd > 31 || (getenv("BIOEM_FORCE_WIDE2") && P.nd >= 21)) &&
      !getenv("BIOEM_NO_WIDE2") && plan_wide2(P, N, H, mD))
    return P;
  // 3. tiles of a window kernel on phase-shifted conv spectra (window_tiles.hpp): launches^2 x the measured cost of one
  //    launch of the 21- / 27- / 31-row kernel (ms at 224^2: k_compare_fast 6.05, k_compare_fastm 6.5 / 6.7)
  //    (also a window that fits one kernel whose instantiation is not in the table -- 31 rows at stride 4: it must not
  //    drop to the generic kernel)
  if (N >= 8 && symmetric && P.nd > 11 && !getenv("BIOEM_NO_TILES"))
  {
    static const int tileRows[3] = {21, 27, 31};
    static const double tileCost[3] = {6.05, 6.5, 6.7};
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int x, int y) {
      const int nx = (P.nd + tileRows[x] - 1) / tileRows[x], ny = (P.nd + tileRows[y] - 1) / tileRows[y];
      return nx * nx * tileCost[x] < ny * ny * tileCost[y];
    });
    for (int k : order)
    {
      int t = tileRows[k];
      if (getenv("BIOEM_TILE_ROWS"))
        t = atoi(getenv("BIOEM_TILE_ROWS")) == 31 ? 31 : atoi(getenv("BIOEM_TILE_ROWS")) == 27 ? 27 : 21;
      KernelPlan Q = P;
      if (!plan_window_kernel(Q, N, H, (t - 1) / 2, false))
        continue;
      Q.tileT = t;
      Q.tilesPerAxis = (P.nd + t - 1) / t;
      for (int i = 0; i < Q.tilesPerAxis; i++)
      {
        Q.tileCenter.push_back(-mD + i * t + Q.winD);       // centre row of tile i
        Q.tileValid.push_back(std::min(t, P.nd - i * t));   // rows of tile i inside the window
      }
      return Q;
    }
  }
  // 4. the direct pruned DFT for what is left (irregular displacement sets beyond 31 rows, N < 8)
  P.family = KF_GENERIC;
  P.fn = find_kernel(KF_GENERIC);
  P.fast = 0;
  P.N1 = 0;
  P.nyq = false;
  // the window rows pass through the LDS in groups of whole register chunks (16 rows) where they do not fit at once --
  // the same work, so the groups shrink before the block does (four waves share the particle's rows through L1)
  P.genericWaves = 4;
  P.genericRows = 0;
  while (compare_lds_bytes(N, H, P.nd, P.genericWaves, P.genericRows) > kLdsCU)
  {
    const int rows = P.genericRows ? P.genericRows : P.nd;
    if (rows > 16)
      P.genericRows = std::max(16, (rows / 2 + 15) / 16 * 16);
    else if (P.genericWaves > 1)
      P.genericWaves >>= 1;
    else
      break;
  }
  P.ldsBytes = compare_lds_bytes(N, H, P.nd, P.genericWaves, P.genericRows);
  if (P.ldsBytes > kLdsCU)
    P.err = "configuration exceeds the 160 KiB LDS budget of the comparison kernel";
  return P;
}

void plan_signature(const KernelPlan &P, char *buf, size_t cap)
{
  const char *nq = P.nyq ? "true" : "false";
  switch (P.family)
  {
  case KF_WIDE2:
    if (P.w2Halves == 2)
      snprintf(buf, cap, P.w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 2, 8>" : "k_compare_wide2<%d, %d, %d, %s, 2>",
               2 * P.fast, P.w2NRW, P.w2NBLK, nq);
    else
      snprintf(buf, cap, P.w2NW == 8 ? "k_compare_wide2<%d, %d, %d, %s, 1, 8>" : "k_compare_wide2<%d, %d, %d, %s>",
               2 * P.fast, P.w2NRW, P.w2NBLK, nq);
    break;
  case KF_FASTM: snprintf(buf, cap, "k_compare_fastm<%d, %d, %s, %d>", P.winD, 2 * P.fast, nq, P.gs); break;
  case KF_FASTM2: snprintf(buf, cap, "k_compare_fastm2<%d, %s, %d>", 2 * P.fast, nq, P.gs); break;
  case KF_FAST: snprintf(buf, cap, "k_compare_fast<%d, %d, %s, %d>", P.winD, 2 * P.fast, nq, P.gs); break;
  case KF_ODDFFT: snprintf(buf, cap, "k_compare_oddfft<%d, %d>", P.winD, P.oddR); break;
  case KF_ROWS: snprintf(buf, cap, "k_compare_rows<%d, %d>", P.winD, P.gs); break;
  default: snprintf(buf, cap, "k_compare_generic"); break;
  }
}

} // namespace

#endif
